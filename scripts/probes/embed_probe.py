#!/usr/bin/env python3
"""Stand-alone timing of the embed launch (v2a_linear_small at the sampler's one-clip shape), 20 launches per hipGraph (tuning aid)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
import v2a_amd  # noqa: E402,F401
from v2a_amd import _lib as L  # noqa: E402

if len(sys.argv) > 2:
    L.LIB_PATH = os.path.abspath(sys.argv[2])
DEV = torch.device("cuda:0")
B, T, K, d, R = int(sys.argv[1]) if len(sys.argv) > 1 else 1, 750, 128, 1024, 32
N = T + R
g = torch.Generator().manual_seed(0)
y = torch.randn(B, T, K, generator=g).to(DEV)
wt = torch.randn(K, d, generator=g).to(DEV)
bias = torch.randn(d, generator=g).to(DEV)
pos = torch.randn(T, d, generator=g).to(DEV)
regs = torch.randn(R, d, generator=g).to(DEV)
out = torch.zeros(2 * B, N, d, device=DEV)
sh = torch.zeros(2 * B, N, d, device=DEV, dtype=torch.bfloat16)


def run():
    L.linear_small(y, wt, bias, pos, out, M=B * T, K=K, T=T, out_batch_stride=N * d, row_off=R, d=d, dup=B, regs=regs, out_bf16=sh)


run()
torch.cuda.synchronize()
st = torch.cuda.Stream()
gr = torch.cuda.CUDAGraph()
with torch.cuda.stream(st):
    with torch.cuda.graph(gr, stream=st):
        for _ in range(20):
            run()
best = 1e9
for _ in range(5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    gr.replay()
    e1.record()
    torch.cuda.synchronize()
    best = min(best, e0.elapsed_time(e1) * 1e3 / 20)
print(f"embed (linear_small {B * T}x{K} -> {d}, dup, registers, shadow): {best:.2f} us; checksum {float(out.double().sum()):.6e}")
