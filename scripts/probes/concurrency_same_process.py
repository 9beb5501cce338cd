#!/usr/bin/env python3
"""Does v2a_linear_small differ while the SAME process runs MFMA kernels on another stream?  (debug aid)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import v2a_amd  # noqa: E402,F401
from v2a_amd import _lib as L  # noqa: E402

DEV = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
R = lambda *s: torch.randn(*s, generator=g).to(DEV)
y = R(5, 120, 32)
wt_in, b_in, pos, regs = R(32, 256), R(256), R(120, 256), R(8, 256)


def run():
    out = torch.zeros(10, 128, 256, device=DEV)
    L.linear_small(y, wt_in, b_in, pos, out, M=5 * 120, K=32, T=120, out_batch_stride=128 * 256, row_off=8, d=256, dup=5, regs=regs)
    return out


ref = run().cpu()
a = torch.randn(4096, 4096, device=DEV, dtype=torch.bfloat16)
side = torch.cuda.Stream()
for kind in ("torch.mm (vendor BLAS)", "v2a_gemm (own MFMA kernel)"):
    bad = 0
    aw = torch.randn(2048, 1024, device=DEV).bfloat16()
    ww = torch.randn(8192, 1024, device=DEV).bfloat16()
    oo = torch.empty(2048, 4096, device=DEV, dtype=torch.bfloat16)
    for it in range(200):
        with torch.cuda.stream(side):
            for _ in range(4):
                if kind.startswith("torch"):
                    b = a @ a
                else:
                    L.gemm([(aw, 1024, 1024)], ww, oo, M=2048, N=8192, compute=L.BF16, epilogue=L.EPI_GEGLU, ldo=4096)
        o = run()
        torch.cuda.synchronize()
        bad += not torch.equal(o.cpu(), ref)
    print(f"same process, other stream busy with {kind}: {bad} of 200 launches differ", flush=True)
