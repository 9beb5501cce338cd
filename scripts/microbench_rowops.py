"""Isolated timing of the row kernels (tuning aid): back-to-back launches bracketed by one event pair."""
import sys, os, math
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import v2a_amd
from v2a_amd import _lib as L

def timeit(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3

dev = "cuda"
for B, N, d in [(2, 782, 1024), (2, 782, 1280), (2, 782, 512), (2, 782, 256), (1, 782, 1024), (4, 782, 1024), (16, 782, 1024), (2, 200, 1024), (2, 3128, 1024)]:
    x = torch.randn(B, N, d, device=dev); out = torch.empty_like(x)
    wt = torch.randn(31, d, device=dev); bias = torch.randn(d, device=dev)
    t = timeit(lambda: L.dwconv(x, out, wt, bias, B=B, N=N, d=d, ksize=31))
    y = torch.empty(B * N, d, dtype=torch.bfloat16, device=dev); g = torch.ones(d, device=dev)
    tn = timeit(lambda: L.rmsnorm(x, y, rows=B * N, d=d, gamma=g))
    mb = B * N * d * 8 / 1e6
    print(f"B={B:2d} N={N:4d} d={d:4d}: dwconv {t:7.2f} us ({mb / t * 1e-3 * 1e3:6.1f} GB/s)   rmsnorm {tn:6.2f} us")
