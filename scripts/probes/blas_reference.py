#!/usr/bin/env python3
"""Calibration only (not a product path): what the vendor BLAS behind torch.mm reaches on the sampler's plain bf16 GEMM shapes,
next to this repo's kernels with their fused epilogues (hipGraph of 20 launches, best of 5)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, v2a_amd  # noqa
from v2a_amd import _lib as L

def timeit(fn, n=20):
    fn(); torch.cuda.synchronize()
    st = torch.cuda.Stream(); g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(st):
        with torch.cuda.graph(g, stream=st):
            for _ in range(n): fn()
    best = 1e9
    for _ in range(5):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); g.replay(); e.record(); torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) / n * 1e3)
    return best

for M, N, K in [(1564, 1024, 1024), (1564, 1024, 4096), (1564, 3088, 1024), (1564, 8192, 1024), (1564, 10240, 1280),
                (12512, 1024, 1024), (12512, 1024, 4096), (12512, 3088, 1024), (12512, 8192, 1024), (12512, 10240, 1280), (12512, 1280, 5120)]:
    a = (torch.randn(M, K, device="cuda") * 0.5).bfloat16()
    w = (torch.randn(N, K, device="cuda") * 0.05).bfloat16()
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    wt = w.t()
    tb = timeit(lambda: torch.mm(a, wt, out=out))
    tm = timeit(lambda: L.gemm([(a, K, K)], w, out, M=M, N=N, compute=L.BF16))
    fl = 2.0 * M * N * K
    print(f"{M:6d} x {N:6d} x {K:5d}: torch.mm (vendor BLAS) {tb:8.2f} us {fl / tb / 1e6:7.1f} TF/s | v2a_gemm (store bf16) {tm:8.2f} us {fl / tm / 1e6:7.1f} TF/s", flush=True)
