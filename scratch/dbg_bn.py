import sys, os, numpy as np, torch
import torch.nn.functional as F
sys.path.insert(0, os.getcwd())
from mopa_amd.dense2d import Img, new_img, bn_fwd, bn_bwd
rng = np.random.Generator(np.random.PCG64(1))
for rows, C in [(48, 256), (12, 512), (192, 128), (48, 64)]:
    x = torch.from_numpy(rng.standard_normal((rows, C), dtype=np.float32) * 2 + 0.5)
    gam = torch.from_numpy(rng.uniform(0.5, 1.5, C).astype(np.float32)); bet = torch.from_numpy(rng.standard_normal(C).astype(np.float32) * 0.1)
    gout = torch.from_numpy(rng.standard_normal((rows, C), dtype=np.float32))
    xr, gr, br = x.double().requires_grad_(True), gam.double().requires_grad_(True), bet.double().requires_grad_(True)
    ref = F.relu(F.batch_norm(xr, None, None, gr, br, True, 0.1, 1e-5))
    (ref * gout.double()).sum().backward()
    P = {"bn.weight": gam.cuda(), "bn.bias": bet.cuda(), "bn.running_mean": torch.zeros(C).cuda(), "bn.running_var": torch.ones(C).cuda()}
    xi = Img(x.cuda(), 1, rows, 1)
    J = torch.zeros(rows, 2 * C, device="cuda")
    y = Img(J, 1, rows, 1, C, C)
    stats = torch.empty(4, C, device="cuda")
    bn_fwd(xi, y, P, "bn", 1, None, True, stats)
    dJ = torch.zeros(rows, 2 * C, device="cuda"); dJ[:, C:] = gout.cuda()
    dy = Img(dJ, 1, rows, 1, C, C)
    dx = new_img(1, rows, 1, C, "cuda"); dg, db = torch.empty(C, device="cuda"), torch.empty(C, device="cuda")
    bn_bwd(dy, xi, dx, stats, 1, None, None, False, True, dg, db)
    print(rows, C, "y", float((y.dense().cpu() - ref.float()).abs().max()), "dx", float((dx.dense().cpu() - xr.grad.float()).abs().max()),
          "dg", float((dg.cpu() - gr.grad.float()).abs().max()), "db", float((db.cpu() - br.grad.float()).abs().max()))
