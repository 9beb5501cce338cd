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

# Round 5: the split (bf16x3) form.  v2a_gemm with hi | lo operand planes runs three bf16 products per fp32 product -- on the 8-phase kernel as
# three passes over K -- so the vendor's plain bf16 GEMM with the K extent TRIPLED issues the same MFMA flops from the same bytes per pass: the
# like-for-like calibration of the kernel the headline mode spends 75 % of its 8-clip time in.
def planes(x):
    hi = x.bfloat16()
    return torch.cat([hi, (x - hi.float()).bfloat16()], -1).contiguous()

print("split operands (three products): v2a_gemm M x N x K, fp32 store, against torch.mm M x N x 3K, bf16 store")
for M, N, K in [(1564, 8192, 1024), (1564, 3088, 1024), (12512, 8192, 1024), (12512, 10240, 1280), (12512, 3088, 1024), (12512, 1024, 4096), (12512, 1280, 5120)]:
    a3 = (torch.randn(M, 3 * K, device="cuda") * 0.5).bfloat16()
    w3 = (torch.randn(N, 3 * K, device="cuda") * 0.05).bfloat16()
    o3 = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    w3t = w3.t()
    tb = timeit(lambda: torch.mm(a3, w3t, out=o3))
    ap = planes(torch.randn(M, K) * 0.5).cuda()
    wp = planes(torch.randn(N, K) * 0.05).cuda()
    of = torch.empty(M, N, device="cuda")
    tm = timeit(lambda: L.gemm([(ap, 2 * K, K)], wp, of, M=M, N=N, compute=L.BF16, a_split=True, tile_hint=5))
    fl = 6.0 * M * N * K
    print(f"{M:6d} x {N:6d} x {K:5d}: torch.mm (K x 3) {tb:8.2f} us {fl / tb / 1e6:7.1f} TF/s | v2a_gemm split, 8-phase {tm:8.2f} us {fl / tm / 1e6:7.1f} TF/s issued  ({tb / tm:4.2f} of the vendor's rate)", flush=True)
