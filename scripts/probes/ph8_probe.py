#!/usr/bin/env python3
"""Which pipe bounds the 8-phase 256x256 K loop?  (tuning aid)  Times v2a_gemm with the 8-phase kernel forced, against probe
libraries built with -DV2A_8PH_SKIP=bits (gemm_8phase.hip: 1 W fragment reads, 2 W DMAs, 4 A fragment reads, 8 A DMAs dropped
after the first K tile).  Only durations mean anything for bits != 0.
usage: python scripts/probes/ph8_probe.py <library.so> [MxNxK:epi ...]       (built by scripts/probes/ph8_probe_run.sh)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import v2a_amd  # noqa: E402,F401
from v2a_amd import _lib  # noqa: E402

MODE = None
if sys.argv[1].startswith("mode"):          # "mode1" / "mode2": the shipped library with v2a_tuning.gemm_8phase = 1 / 2
    MODE = int(sys.argv[1][4:])
else:
    _lib.LIB_PATH = os.path.abspath(sys.argv[1])
DEV = torch.device("cuda:0")
REPS = 20


def main():
    specs = sys.argv[2:] or ["12512x8192x1024:geglu", "12512x1024x4096:resid", "12512x3088x1024:store", "1564x8192x1024:geglu", "12512x8192x4096:geglu"]
    if MODE is not None:
        _lib.set_tuning(eight_phase=MODE)
    for spec in specs:
        shp, epi = spec.split(":")
        M, N, K = (int(v) for v in shp.split("x"))
        g = torch.Generator().manual_seed(0)
        a = (torch.randn(M, K, generator=g) * 0.5).to(DEV, torch.bfloat16)
        w = (torch.randn(N, K, generator=g) * 0.05).to(DEV, torch.bfloat16)
        if epi == "geglu":
            out = torch.empty(M, N // 2, device=DEV, dtype=torch.bfloat16)
            kw = dict(epilogue=_lib.EPI_GEGLU, ldo=N // 2)
        elif epi == "store":
            out = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
            kw = dict()
        else:
            out = torch.empty(M, N, device=DEV)
            kw = dict(epilogue=_lib.EPI_RESID, resid=torch.randn(M, N, generator=g).to(DEV))
        run = lambda: _lib.gemm([(a, K, K)], w, out, M=M, N=N, compute=_lib.BF16, tile_hint=7, **kw)
        run()
        torch.cuda.synchronize()
        st = torch.cuda.Stream()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.stream(st):
            with torch.cuda.graph(gr, stream=st):
                for _ in range(REPS):
                    run()
        best = 1e9
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            gr.replay()
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) * 1e3 / REPS)
        print(f"{os.path.basename(sys.argv[1]):28s} {spec:26s} {best:8.2f} us  {2.0 * M * N * K / best / 1e6:7.1f} TF/s", flush=True)


if __name__ == "__main__":
    main()
