#!/usr/bin/env python3
"""Time v2a_gemm on a list of MxNxK shapes (bf16, RESID epilogue, fp32 out) inside a hipGraph.

Tuning aid: the tile configuration is forced per process with V2A_GEMM_TILE (0: 128x256, 1: 128x128, 2: 128x64,
3: 64x64, 5: 256x256); `split_k` > 1 on a shape spec (MxNxKxS) exercises the split-K path.
usage: V2A_GEMM_TILE=1 python scripts/gemm_probe.py 1564x1024x1024 3128x1024x512 ...
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import v2a_amd  # noqa: E402,F401
from v2a_amd import _lib  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    reps = 40
    for spec in sys.argv[1:]:
        dims = [int(v) for v in spec.split("x")]
        M, N, K = dims[:3]
        S = dims[3] if len(dims) > 3 else 1
        g = torch.Generator(device="cpu").manual_seed(0)
        a = (torch.randn(M, K, generator=g) * 0.5).to(dev, torch.bfloat16)
        w = (torch.randn(N, K, generator=g) * 0.05).to(dev, torch.bfloat16)
        res = torch.randn(M, N, generator=g).to(dev)
        out = torch.empty(M, N, device=dev)
        kw = {}
        if S > 1:
            kw = dict(split_k=S, workspace=torch.zeros(S * M * N + 4096, device=dev))

        def run():
            _lib.gemm([(a, K, K)], w, out, M=M, N=N, compute=_lib.BF16, epilogue=_lib.EPI_RESID, resid=res, **kw)

        run()
        torch.cuda.synchronize()
        ref = res + a.float() @ w.float().t()
        err = (out - ref).abs().max().item()
        st = torch.cuda.Stream()
        with torch.cuda.stream(st):
            run()
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr, stream=st):
                for _ in range(reps):
                    run()
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            gr.replay()
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) * 1e3 / reps)
        print(f"tile={os.environ.get('V2A_GEMM_TILE', 'auto'):>4s} {spec:>20s}  {best:7.2f} us  {2.0 * M * N * K / best / 1e6:7.1f} TF/s  maxerr {err:.3e}",
              flush=True)


if __name__ == "__main__":
    main()
