import sys, os, numpy as np, torch
sys.path.insert(0, os.getcwd())
from oracle import net2d
from oracle.params import det_tensor
from mopa_amd import dense2d
from mopa_amd.config import default_cfg
from mopa_amd.models.build import build_model_2d
g = dict(np.load("tests/golden/g1_net2dseg_pad_train.npz"))
model, _ = build_model_2d(default_cfg(5, True))
model.load_state_dict({k: det_tensor(k, v.shape) for k, v in model.state_dict().items()})
model.net_2d.dropout.p = 0.0
model = model.cuda().train()
dense2d.DEBUG = {}
out = model({"img": torch.from_numpy(g["img"]), "img_indices": [g["idx0"], g["idx1"]]})
sum((out[k] * torch.from_numpy(g["gin_" + k]).cuda()).sum() for k in out).backward()
dt = torch.float64
P = {k: (det_tensor(k, v).to(dt) if "num_batches" not in k else det_tensor(k, v)) for k, v in net2d.param_shapes(5, True).items()}
for k, v in P.items():
    if v.dtype.is_floating_point and "running" not in k:
        v.requires_grad_(True)
taps = {}
ref = net2d.net2dseg_forward(P, torch.from_numpy(g["img"]).to(dt), [g["idx0"], g["idx1"]], training=True, dropout_p=0.0, taps=taps)
sum((ref[k] * torch.from_numpy(g["gin_" + k]).to(dt)).sum() for k in ref).backward()
for stage, lvl in (("5", 3), ("4", 2), ("3", 1), ("2", 0)):
    t = taps["join" + stage].grad  # (B, 2C, H, W)
    B, C2, H, W = t.shape
    got = dense2d.DEBUG[f"dJ{lvl}"].cpu().double().reshape(B, H, W, C2).permute(0, 3, 1, 2)
    e = (got - t).abs()
    h = C2 // 2
    tv = taps["join" + stage].detach()
    gotv = dense2d.DEBUG[f"J{lvl}"].cpu().double().reshape(B, H, W, C2).permute(0, 3, 1, 2)
    ev = (gotv - tv).abs()
    flips = ((gotv > 0) != (tv > 0))
    print(stage, "mask flips left/right", int(flips[:, :h].sum()), int(flips[:, h:].sum()), "min |y| at flips", float(tv.abs()[flips].min()) if flips.any() else None)
    print(stage, "FWD scale", float(tv.abs().max()), "left", float(ev[:, :h].max()), "right", float(ev[:, h:].max()))
    print(stage, "scale", float(t.abs().max()), "left err", float(e[:, :h].max()), "right err", float(e[:, h:].max()))
