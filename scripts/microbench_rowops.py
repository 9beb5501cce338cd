"""Isolated timing of the row kernels (tuning aid): 20 back-to-back launches inside a hipGraph, best of 5 replays."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import v2a_amd  # noqa: E402,F401
from v2a_amd import _lib as L  # noqa: E402


def timeit(fn, n=20):
    fn()
    torch.cuda.synchronize()
    st = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=st):
        for _ in range(n):
            fn()
    best = 1e9
    for _ in range(5):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        g.replay()
        e.record()
        torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) / n * 1e3)
    return best


dev = "cuda"
for B, N, d in [(2, 782, 1024), (2, 782, 1280), (2, 782, 512), (16, 782, 1024), (16, 782, 1280), (16, 782, 512)]:
    x = torch.randn(B, N, d, device=dev)
    out = torch.empty_like(x)
    wt = torch.randn(31, d, device=dev)
    bias = torch.randn(d, device=dev)
    y = torch.empty(B * N, d, dtype=torch.bfloat16, device=dev)
    g = torch.ones(d, device=dev)
    ssq = torch.zeros(B * N, (d // 32 + 3) // 4 * 4, device=dev)
    nk = dict(norm=dict(out_bf16=y, gamma=g, ssq=ssq))
    L.set_tuning(dwconv_rows_per_wave=-1)          # the per-wave kernel for every size (round 2)
    t6 = timeit(lambda: L.dwconv(x, out, wt, bias, B=B, N=N, d=d, ksize=31))
    t6n = timeit(lambda: L.dwconv(x, out, wt, bias, B=B, N=N, d=d, ksize=31, **nk))
    L.set_tuning()                                 # default: streaming kernel for chip-filling launches
    t = timeit(lambda: L.dwconv(x, out, wt, bias, B=B, N=N, d=d, ksize=31))
    tnm = timeit(lambda: L.dwconv(x, out, wt, bias, B=B, N=N, d=d, ksize=31, **nk))
    tn = timeit(lambda: L.rmsnorm(x, y, rows=B * N, d=d, gamma=g))
    mb = B * N * d * 8 / 1e6
    mbn = B * N * d * 6 / 1e6
    mb2 = B * N * d * 10 / 1e6
    print(f"B={B:2d} N={N:4d} d={d:4d}: dwconv {t:7.2f} us ({mb / t * 1e3:7.1f} GB/s; per-wave kernel {t6:7.2f} us)  dwconv+norm {tnm:7.2f} us "
          f"({mb2 / tnm * 1e3:7.1f} GB/s; per-wave kernel {t6n:7.2f} us)  rmsnorm {tn:6.2f} us ({mbn / tn * 1e3:7.1f} GB/s)", flush=True)
