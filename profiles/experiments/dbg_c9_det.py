import sys, torch
sys.path.insert(0, ".")
from mopa_amd._lib import call, ptr, stream
torch.manual_seed(0)
for (B,H,W,cin,cout,acc) in ((16,152,240,64,64,0),(16,304,480,64,128,0),(16,304,480,128,64,0),(16,76,120,128,128,1),(16,152,240,128,64,1)):
    x = torch.randn(B*H*W, cin, device="cuda")
    w = torch.randn(cout, cin, 3, 3, device="cuda")*0.1
    Uq = torch.empty(36, cin, cout, device="cuda"); Uf = torch.empty(36, cin, cout, device="cuda")
    call("mopa_wino4_weight_q", ptr(w), cout, cin, 0, ptr(Uq), stream())
    call("mopa_wino4_weight_f", ptr(w), cout, cin, 0, ptr(Uf), stream())
    prev = torch.randn(B*H*W, cout, device="cuda")
    outs = []
    for r in range(6):
        o = prev.clone() if acc else torch.full((B*H*W, cout), float("nan"), device="cuda")
        call("mopa_wino4_conv9", ptr(x), cin, ptr(Uq), None, ptr(o), cout, B, H, W, cin, cout, acc, None, 1, 0, stream())
        outs.append(o)
    o1 = prev.clone() if acc else torch.empty(B*H*W, cout, device="cuda")
    call("mopa_wino4_conv", ptr(x), cin, ptr(Uf), None, ptr(o1), cout, B, H, W, cin, cout, acc, None, 1, 0, None, stream())
    torch.cuda.synchronize()
    nd = [int((outs[r] != outs[0]).sum()) for r in range(1, 6)]
    print((B,H,W,cin,cout,acc), "elements differing from run 0:", nd, "max |conv9 - conv| / scale", float((outs[0]-o1).abs().max()/o1.abs().max()))
