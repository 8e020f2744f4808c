import sys, os, numpy as np, torch
import torch.nn.functional as F
sys.path.insert(0, os.getcwd())
from mopa_amd.dense2d import ConvOp, ConvTOp, Img, new_img, bn_fwd, bn_bwd
rng = np.random.Generator(np.random.PCG64(1))
def nhwc(t):
    B, C, H, W = t.shape
    return Img(t.permute(0, 2, 3, 1).reshape(B * H * W, C).contiguous().cuda(), B, H, W)
def nchw(img):
    return img.dense().reshape(img.B, img.H, img.W, img.C).permute(0, 3, 1, 2).cpu()
for (cin, cout, H, W) in [(512, 256, 4, 6), (256, 128, 8, 12), (512, 512, 2, 3)]:
    B = 2
    x = torch.from_numpy(rng.standard_normal((B, cin, H, W), dtype=np.float32))
    w = torch.from_numpy(rng.standard_normal((cout, cin, 3, 3), dtype=np.float32) * 0.05)
    xr, wr = x.double().requires_grad_(True), w.double().requires_grad_(True)
    ref = F.conv2d(xr, wr, None, 1, 1)
    gout = torch.from_numpy(rng.standard_normal(tuple(ref.shape), dtype=np.float32))
    (ref * gout.double()).sum().backward()
    op = ConvOp(w.cuda(), None, 3, 1, 1)
    xi = nhwc(x); out = new_img(B, H, W, cout, "cuda")
    op.forward(xi, out)
    dx = new_img(B, H, W, cin, "cuda"); dw = torch.empty_like(op.w)
    op.backward(xi, nhwc(gout), dx, dw, None, False)
    e = (nchw(dx) - xr.grad.float()).abs()
    print(cin, cout, H, W, "fwd", float((nchw(out) - ref.float()).abs().max()), "dgrad err by 64-col block",
          [round(float(e[:, i:i + 64].max()), 5) for i in range(0, cin, 64)], "dw", float((dw.cpu() - wr.grad.float()).abs().max()))
# convT 512->256 at 2x3
B, cin, cout, H, W = 2, 512, 256, 2, 3
x = torch.from_numpy(rng.standard_normal((B, cin, H, W), dtype=np.float32))
w = torch.from_numpy(rng.standard_normal((cin, cout, 2, 2), dtype=np.float32) * 0.05)
b = torch.from_numpy(rng.standard_normal(cout, dtype=np.float32))
xr, wr, br = x.double().requires_grad_(True), w.double().requires_grad_(True), b.double().requires_grad_(True)
ref = F.conv_transpose2d(xr, wr, br, 2)
gout = torch.from_numpy(rng.standard_normal(tuple(ref.shape), dtype=np.float32))
(ref * gout.double()).sum().backward()
op = ConvTOp(w.cuda(), b.cuda())
out = new_img(B, 2 * H, 2 * W, cout, "cuda"); xi = nhwc(x)
op.forward(xi, out)
dx = new_img(B, H, W, cin, "cuda"); dw, db = torch.empty_like(op.w), torch.empty_like(op.b)
op.backward(xi, nhwc(gout), dx, dw, db)
print("convT fwd", float((nchw(out) - ref.float()).abs().max()), "dx", float((nchw(dx) - xr.grad.float()).abs().max()),
      "dw", float((dw.cpu() - wr.grad.float()).abs().max()), "db", float((db.cpu() - br.grad.float()).abs().max()))
