import sys, torch, numpy as np
sys.path.insert(0, ".")
import torch.nn.functional as F
from mopa_amd._lib import call, ptr, stream
torch.manual_seed(0)
for (B,H,W,cin,cout) in ((1,16,16,16,64),(1,16,32,16,64),(1,32,32,32,64),(2,40,64,64,128),(1,36,60,64,64)):
    x = torch.randn(B*H*W, cin, device="cuda")
    w = torch.randn(cout, cin, 3, 3, device="cuda")*0.1
    Uq = torch.empty(36, cin, cout, device="cuda")
    call("mopa_wino4_weight_q", ptr(w), cout, cin, 0, ptr(Uq), stream())
    o = torch.full((B*H*W, cout), float("nan"), device="cuda")
    call("mopa_wino4_conv9", ptr(x), cin, ptr(Uq), None, ptr(o), cout, B, H, W, cin, cout, 0, None, 1, 0, stream())
    ref = F.conv2d(x.reshape(B,H,W,cin).permute(0,3,1,2).double(), w.double(), None, padding=1).permute(0,2,3,1).reshape(B*H*W, cout)
    err = (o.double()-ref).abs()
    e = err.reshape(B,H,W,cout)
    print((B,H,W,cin,cout), "max err", float(err.max()), "scale", float(ref.abs().max()), "nan", int(torch.isnan(o).sum()))
    bad = (e > 1e-3).float()
    print("  bad frac by co block of 32:", [round(float(bad[...,k*32:(k+1)*32].mean()),3) for k in range(cout//32)])
    print("  bad frac by pixel row%4:", [round(float(bad[:, r::4].mean()),3) for r in range(4)], "col%4:", [round(float(bad[:,:,c::4].mean()),3) for c in range(4)])
    print("  bad frac by tile col:", [round(float(bad[:,:,4*k:4*k+4].mean()),2) for k in range(W//4)][:16])
print("---- large")
for (B,H,W,cin,cout) in ((4,152,240,64,64),(16,152,240,64,64),(16,76,120,128,128)):
    x = torch.randn(B*H*W, cin, device="cuda")
    w = torch.randn(cout, cin, 3, 3, device="cuda")*0.1
    Uq = torch.empty(36, cin, cout, device="cuda"); Uf = torch.empty(36, cin, cout, device="cuda")
    call("mopa_wino4_weight_q", ptr(w), cout, cin, 0, ptr(Uq), stream())
    call("mopa_wino4_weight_f", ptr(w), cout, cin, 0, ptr(Uf), stream())
    o = torch.full((B*H*W, cout), float("nan"), device="cuda"); o1 = torch.full((B*H*W, cout), float("nan"), device="cuda")
    call("mopa_wino4_conv9", ptr(x), cin, ptr(Uq), None, ptr(o), cout, B, H, W, cin, cout, 0, None, 1, 0, stream())
    call("mopa_wino4_conv", ptr(x), cin, ptr(Uf), None, ptr(o1), cout, B, H, W, cin, cout, 0, None, 1, 0, None, stream())
    ref = F.conv2d(x.reshape(B,H,W,cin).permute(0,3,1,2), w, None, padding=1).permute(0,2,3,1).reshape(B*H*W, cout)
    for nm, oo in (("conv9", o), ("conv", o1)):
        err = (oo-ref).abs(); e = err.reshape(B,H,W,cout); bad = (e > 1e-2).float()
        print((B,H,W,cin,cout), nm, "max err", float(err.max()), "nan", int(torch.isnan(oo).sum()), "bad frac", float(bad.mean()),
              "by image:", [round(float(bad[b].mean()),3) for b in range(min(B,6))], "by row block:", [round(float(bad[:, 16*k:16*k+16].mean()),2) for k in range(min(H//16,10))])
