import sys, os, ctypes, torch
sys.path.insert(0, os.getcwd())
from mopa_amd import dense2d
from mopa_amd._lib import call, ptr, stream
B = 8
shapes = [("L1 64->64 152x240", 64, 64, 152, 240), ("L2 128->128 76x120", 128, 128, 76, 120), ("L3 256->256 38x60", 256, 256, 38, 60),
          ("L4 512->512 19x30", 512, 512, 19, 30), ("D4 512->256 38x60", 512, 256, 38, 60), ("D3 256->128 76x120", 256, 128, 76, 120),
          ("D2 128->64 152x240", 128, 64, 152, 240), ("D1 128->64 304x480", 128, 64, 304, 480), ("D1dgrad 64->128 304x480", 64, 128, 304, 480)]
for name, cin, cout, H, W in shapes:
    x = torch.randn(B * H * W, cin, device="cuda")
    w = torch.randn(3, 3, cin, cout, device="cuda") * 0.05
    out = torch.empty(B * H * W, cout, device="cuda")
    g = dense2d._geom(B=B, IH=H, IW=W, OHl=H, OWl=W, OHa=H, OWa=W, IY0=-1, IX0=-1, TH=3, TW=3, KWF=3, Cin=cin, Cout=cout, ld_in=cin, ld_out=cout)
    flops = 2.0 * B * H * W * cout * 9 * cin
    res = []
    for tile in (0, 1, 2, 3, -1):
        if tile >= 0 and cout % (64, 128, 64, 64)[tile]:
            res.append("   -  "); continue
        flags = ((tile + 1) << 8) if tile >= 0 else 0
        for _ in range(2):
            call("mopa_conv2d_igemm", ptr(x), ptr(w), None, ptr(out), ctypes.addressof(g), flags, stream())
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(5):
            call("mopa_conv2d_igemm", ptr(x), ptr(w), None, ptr(out), ctypes.addressof(g), flags, stream())
        e.record(); torch.cuda.synchronize()
        res.append("%6.1f" % (flops * 5 / (s.elapsed_time(e) * 1e-3) / 1e12))
    print("%-28s TF/s  256x64 %s | 128x128 %s | 128x64 %s | 64x64 %s | auto %s" % (name, *res))
