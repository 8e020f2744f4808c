#!/usr/bin/env python3
"""How much does the conv algorithm move the parameter gradients of Net2DSeg at the bench size (B images of 302x480)?
Truth = the fp64 CPU oracle.  Configs: direct kernels only, Winograd F(2x2) (forward + backward), F(4x4) in the backward
passes, F(4x4) in all three passes.  Prints median / 90th percentile / max of  max|g - truth| / max|truth|  over the
parameter tensors, and writes them to gpurun_out/f4_gradient_noise.json (copy it to profiles/<round>_f4_gradient_noise.json: bench.py
quotes it as config.gradient_error_vs_fp64).  Usage: python profiles/f4_gradient_noise.py [B=8]"""
import json, os, subprocess, sys, time
import numpy as np
import torch
sys.path.insert(0, os.getcwd())
from mopa_amd import dense2d
from mopa_amd.config import default_cfg
from mopa_amd.models.build import build_model_2d
from oracle import net2d
from oracle.params import det_tensor

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
H, W = 302, 480
rng = np.random.Generator(np.random.PCG64(3))
img = torch.from_numpy(rng.random((B, 3, H, W), dtype=np.float32))
idx = [np.stack([rng.integers(0, H, 4000), rng.integers(0, W, 4000)], 1) for _ in range(B)]
shapes = net2d.param_shapes(5, True)


def loss_of(out):
    return out["seg_logit"].square().mean() + out["seg_logit2"].square().mean() + out["seg_logit_all"].square().mean()


def run(roles, winograd=True):
    dense2d.F4_ROLES = roles
    dense2d.F4_FWD_MIN_PIXELS = 0
    os.environ["MOPA_WINOGRAD"] = "1" if winograd else "0"
    model = build_model_2d(default_cfg())[0]
    model.load_state_dict({k: det_tensor(k, v.shape) for k, v in model.state_dict().items()})
    model = model.cuda().train()
    model.net_2d.dropout.p = 0.0
    loss_of(model({"img": img, "img_indices": idx})).backward()
    return {k: p.grad.cpu() for k, p in model.named_parameters()}


t0 = time.time()
torch.set_num_threads(min(64, os.cpu_count() or 1))
P = {k: (det_tensor(k, v).double() if "num_batches" not in k else det_tensor(k, v)) for k, v in shapes.items()}
for k, v in P.items():
    if v.dtype.is_floating_point and "running" not in k:
        v.requires_grad_(True)
loss_of(net2d.net2dseg_forward(P, img.double(), idx, training=True, dropout_p=0.0)).backward()
print(f"fp64 oracle: {time.time() - t0:.0f} s", flush=True)
out = {"what": "max |g - g_fp64| / max |g_fp64| per parameter tensor of Net2DSeg (train mode, dropout off), median / p90 / max over the tensors; "
               "fp64 = the CPU oracle (oracle/net2d.py)", "images": f"{B} x {H} x {W}", "points_per_image": 4000,
       "commit": subprocess.run(("git", "rev-parse", "--short", "HEAD"), capture_output=True, text=True).stdout.strip() or None, "algorithms": {}}
for name, roles, wg in (("direct kernels", (), False), ("F(2x2) fwd+bwd", (), True), ("F(4x4) backward", ("dgrad", "wgrad"), True),
                        ("F(4x4) fwd+bwd", ("fwd", "dgrad", "wgrad"), True)):
    g = run(roles, wg)
    errs = []
    for k in g:
        truth = P[k].grad.float()
        scale = float(truth.abs().max())
        if scale > 1e-6:
            errs.append(float((g[k] - truth).abs().max()) / scale)
    errs = np.array(errs)
    print(f"{name:18s} median {np.median(errs):.2e}  p90 {np.percentile(errs, 90):.2e}  max {errs.max():.2e}  ({len(errs)} tensors)", flush=True)
    out["algorithms"][name] = {"median": float(f"{np.median(errs):.3e}"), "p90": float(f"{np.percentile(errs, 90):.3e}"), "max": float(f"{errs.max():.3e}"),
                               "tensors": len(errs), "shipped": name == "F(4x4) fwd+bwd"}
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/f4_gradient_noise.json", "w"), indent=1)
