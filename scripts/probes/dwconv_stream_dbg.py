import os, sys, ctypes as C
sys.path.insert(0, "/root/repo")
import torch, v2a_amd
from v2a_amd import _lib as L
sys.path.insert(0, "/root/repo/scripts")
def timeit(fn, n=20):
    fn(); torch.cuda.synchronize()
    st = torch.cuda.Stream(); g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=st):
        for _ in range(n): fn()
    best = 1e9
    for _ in range(5):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); g.replay(); e.record(); torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) / n * 1e3)
    return best
B, N, d = 16, 782, 1024
x = torch.randn(B, N, d, device="cuda"); out = torch.empty_like(x); wt = torch.randn(31, d, device="cuda"); bias = torch.randn(d, device="cuda")
for dbg in (0, 1, 2, 3):
    t = L.Tuning(-1, 0, 1, 0, 0, 0, 0); t.reserved[0] = dbg
    L.check(L.lib().v2a_set_tuning(C.byref(t)))
    print("dbg", dbg, "(1 = no FMAs, 2 = no DMA after the prologue):", round(timeit(lambda: L.dwconv(x, out, wt, bias, B=B, N=N, d=d, ksize=31)), 2), "us")
