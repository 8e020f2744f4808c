import torch, sys
sys.path.insert(0, "/root/repo")
x = torch.zeros(1 << 20, device="cuda")
torch.cuda.synchronize()
def bracket(fn, n=200):
    evs = []
    for _ in range(n):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record(); evs.append((s, e))
    torch.cuda.synchronize()
    v = sorted(s.elapsed_time(e) * 1e3 for s, e in evs)
    return v[len(v) // 2], sum(v) / len(v)
print("empty bracket (median, mean us):", bracket(lambda: None))
print("tiny kernel bracket:", bracket(lambda: x.add_(1.0)))
y = torch.zeros(1 << 26, device="cuda")
print("256 MB add bracket:", bracket(lambda: y.add_(1.0), 50))
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(50): y.add_(1.0)
e.record(); torch.cuda.synchronize()
print("256 MB add back-to-back us each:", s.elapsed_time(e) * 1e3 / 50)
