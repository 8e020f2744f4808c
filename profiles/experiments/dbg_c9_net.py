import sys, torch
sys.path.insert(0, ".")
from mopa_amd._lib import call, ptr, stream
torch.manual_seed(0)
# (B,H,W,cin,cout,ld_in,col_in,ld_out,col_out,dgrad,acc)
for (B,H,W,cin,cout,ldi,ci0,ldo,co0,dg,acc) in ((8,304,480,64,128,64,0,128,0,1,0),(8,304,480,64,128,64,0,128,0,1,1),(8,152,240,64,128,64,0,128,0,1,0),
                                                (8,152,240,64,64,64,0,64,0,1,1),(8,152,240,64,64,128,64,64,0,0,0),(8,304,480,128,64,128,0,64,0,0,0),
                                                (8,152,240,64,64,64,0,128,64,1,1)):
    xw = torch.randn(B*H*W, ldi, device="cuda")
    O, I = (cin, cout) if dg else (cout, cin)
    w = torch.randn(O, I, 3, 3, device="cuda")*0.1
    Uq = torch.empty(36, cin, cout, device="cuda"); Uf = torch.empty(36, cin, cout, device="cuda")
    call("mopa_wino4_weight_q", ptr(w), O, I, dg, ptr(Uq), stream())
    call("mopa_wino4_weight_f", ptr(w), O, I, dg, ptr(Uf), stream())
    prev = torch.randn(B*H*W, ldo, device="cuda")
    o9, o1 = prev.clone(), prev.clone()
    call("mopa_wino4_conv9", ptr(xw, ci0), ldi, ptr(Uq), None, ptr(o9, co0), ldo, B, H, W, cin, cout, acc, None, 1, 0, stream())
    call("mopa_wino4_conv", ptr(xw, ci0), ldi, ptr(Uf), None, ptr(o1, co0), ldo, B, H, W, cin, cout, acc, None, 1, 0, None, stream())
    torch.cuda.synchronize()
    d = (o9 - o1).abs()
    print((B,H,W,cin,cout,ldi,ci0,ldo,co0,dg,acc), "max diff / scale", float(d.max() / o1.abs().max()), "outside slice equal", bool(torch.equal(o9[:, :co0], o1[:, :co0]) and torch.equal(o9[:, co0+cout:], o1[:, co0+cout:])))
