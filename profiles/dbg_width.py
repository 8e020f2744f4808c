import sys, numpy as np, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import test_gpu_3d as T
from oracle import scn3d
m, num_planes, residual = int(sys.argv[1]), int(sys.argv[2]), bool(int(sys.argv[3]))
torch.manual_seed(0)
import os
c = T._cloud(int(os.environ.get("SEED", "7")), n=6000, size=48)
model = T._build_3d(num_planes, 1, residual=residual, m=m); model.train(True)
rng = np.random.Generator(np.random.PCG64(5))
feats = torch.from_numpy(rng.random((c.shape[0], 1), dtype=np.float32) + 0.5)
f_dev = feats.cuda().requires_grad_(True)
sd_before = {k: v.clone() for k, v in scn3d.fold_state_dict(model.state_dict()).items()}
out = model({"x": [torch.from_numpy(c), f_dev]})
gouts = {k: torch.from_numpy(rng.standard_normal(tuple(v.shape), dtype=np.float32)) for k, v in out.items()}
sum((out[k] * gouts[k].cuda()).sum() for k in out).backward()
ref_m = T._build_3d(num_planes, 1, residual=residual, m=m); ref_m.load_state_dict(sd_before)
P, f, ref = T._oracle_run(ref_m, c, feats, num_planes, True, gouts, residual=residual, m=m)
named = dict(model.named_parameters())
for k in ("feats", "seg_logit", "seg_logit2"):
    t = ref[k].detach().numpy(); print(k, float(np.abs(out[k].detach().cpu().numpy() - t).max() / np.abs(t).max()))
for k, p in P.items():
    if p.requires_grad:
        t = p.grad.numpy(); g = named[k].grad.cpu().numpy()
        rel = float(np.abs(g - t).max() / max(1e-9, np.abs(t).max()))
        worst = max(globals().get("worst", 0.0), rel)
        if os.environ.get("QUIET") != "1":
            print(f"{k:50s} {tuple(t.shape)} rel {rel:.2e}")
print("worst gradient rel error", worst)
for k in sys.argv[4:]:
    t = P[k].grad.numpy(); g = named[k].grad.cpu().numpy()
    print(k, "truth", np.round(t, 4).tolist()); print(k, "got - truth", np.round(g - t, 5).tolist())
