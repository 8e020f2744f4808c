import sys, torch, numpy as np
sys.path.insert(0, ".")
from mopa_amd._lib import call, ptr, stream
torch.manual_seed(0)
for (O, I, dgrad) in ((64, 64, 0), (64, 128, 1), (128, 64, 0), (128, 128, 1)):
    w = torch.randn(O, I, 3, 3, device="cuda")
    R, C = (O, I) if dgrad else (I, O)
    a = torch.empty(36, R, C, device="cuda"); b = torch.full((36, R, C), float("nan"), device="cuda")
    call("mopa_wino4_weight_q", ptr(w), O, I, dgrad, ptr(a), stream())
    desc = np.asarray([[w.data_ptr(), b.data_ptr(), O, I, 3, 3, 2, dgrad | 6]], dtype=np.int64)
    call("mopa_conv2d_weight_forms_batched", desc.ctypes.data, 1, stream())
    torch.cuda.synchronize()
    print((O, I, dgrad), "equal:", bool(torch.equal(a, b)), "nan in batched:", int(torch.isnan(b).sum()), "max diff", float((a - b).abs().nan_to_num(9).max()))
