"""A/B of the offset-major sparse convolution inside a whole network: run once per setting of MOPA_SPCONV_RUN (the switch is read
once per process), the second run compares with the first run's saved outputs and gradients.
Usage: MOPA_SPCONV_RUN=0 python profiles/dbg_run_ab.py save m planes residual; MOPA_SPCONV_RUN=1 python profiles/dbg_run_ab.py cmp m planes residual"""
import os, sys, numpy as np, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import test_gpu_3d as T
mode, m, num_planes, residual = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), bool(int(sys.argv[4]))
reps = int(os.environ.get("REPS", "1")); in_ch = int(os.environ.get("INCH", "1"))
torch.manual_seed(0)
c = T._cloud(int(os.environ.get("SEED", "7")), n=int(os.environ.get("NPTS", "6000")), size=120 if num_planes == 7 else 48)
model = T._build_3d(num_planes, in_ch, block_reps=reps, residual=residual, m=m); model.train(os.environ.get('EVAL', '0') != '1')
rng = np.random.Generator(np.random.PCG64(5))
feats = torch.from_numpy(rng.random((c.shape[0], in_ch), dtype=np.float32) + 0.5)
if os.environ.get("EVAL", "0") == "1" and os.environ.get("EVALFIX", "1") == "1":   # as tests/test_gpu_3d.py::_check_net3dseg: running statistics := batch statistics
    from oracle import scn3d
    P0 = {k: v.detach().cpu().double().clone() for k, v in scn3d.fold_state_dict(model.state_dict()).items()}
    old, scn3d.BN_MOMENTUM = scn3d.BN_MOMENTUM, 1.0
    scn3d.net3dseg_forward(P0, scn3d.Geometry(c, num_planes), feats.double(), training=True, num_planes=num_planes, block_reps=reps, residual_blocks=residual, m=m)
    scn3d.BN_MOMENTUM = old
    model.load_state_dict({k: v.float() for k, v in P0.items()})
f_dev = feats.cuda().requires_grad_(True)
out = model({"x": [torch.from_numpy(c), f_dev]})
gouts = {k: torch.from_numpy(rng.standard_normal(tuple(v.shape), dtype=np.float32)) for k, v in out.items()}
sum((out[k] * gouts[k].cuda()).sum() for k in out).backward()
res = {k: v.detach().cpu() for k, v in out.items()}
res.update({"grad/" + k: p.grad.cpu() for k, p in model.named_parameters()})
res["dfeats"] = f_dev.grad.cpu()
path = f"/tmp/dbg_run_ab_{m}_{num_planes}_{int(residual)}.pt"
if mode == "save":
    torch.save(res, path)
else:
    ref = torch.load(path)
    for k in res:
        d = float((res[k] - ref[k]).abs().max()); s = float(ref[k].abs().max()) + 1e-30
        nbad = int(((res[k] - ref[k]).abs() > 1e-4 * s).sum())
        print(f"{k:55s} rel {d / s:.2e}  elements off by > 1e-4 of scale: {nbad} of {res[k].numel()}")
