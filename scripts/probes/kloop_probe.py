#!/usr/bin/env python3
"""What bounds the K loop of the LDS-DMA ring GEMM at one clip?  (tuning aid; needs the instrumented library built by
`bash video-to-audio-and-piano-rp_amd/csrc/build.sh --probe`)

Every launch is timed in a hipGraph of 20 back-to-back launches with parts of the K loop switched off (bits of
v2a_tuning.reserved[0], probe builds only):
   1  no LDS fragment reads, no MFMAs (the operand stream alone: DMA issue, counted wait, barrier)
   2  no DMA and no wait inside the K loop (fragment reads + MFMAs + barrier alone, on stale LDS)
   4 / 8  A / W operand DMA with the non-temporal cache policy
  16  no workgroup barrier
Results of the masked variants are wrong by construction; only the durations mean something.
usage: python scripts/probes/kloop_probe.py [MxNxK ...]
"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import v2a_amd  # noqa: E402,F401
from v2a_amd import _lib  # noqa: E402

_lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), "libv2a_cfm_probe.so")
DEV = torch.device("cuda:0")
REPS = 20


def tuning(force_tile, dbg):
    t = _lib.Tuning(force_tile, 0, 1, 0, 0, 0, 0)
    t.reserved[0] = dbg
    _lib.check(_lib.lib().v2a_set_tuning(C.byref(t)))


def main():
    shapes = sys.argv[1:] or ["1564x1024x4096", "1564x1024x1024", "1564x1280x5120", "1564x3088x1024", "12512x1024x4096"]
    tiles = [(3, "64x64"), (8, "64x64 6-deep"), (1, "128x128"), (0, "128x256")]
    dbgs = [(0, "full"), (1, "stream only"), (2, "compute only"), (18, "compute, no barrier"), (4, "A nt"), (8, "W nt"), (12, "A+W nt"), (5, "stream, A nt"), (9, "stream, W nt")]
    for spec in shapes:
        M, N, K = (int(v) for v in spec.split("x"))
        g = torch.Generator().manual_seed(0)
        a = (torch.randn(M, K, generator=g) * 0.5).to(DEV, torch.bfloat16)
        w = (torch.randn(N, K, generator=g) * 0.05).to(DEV, torch.bfloat16)
        res = torch.randn(M, N, generator=g).to(DEV)
        out = torch.empty(M, N, device=DEV)
        for tile, tname in tiles:
            graphs = {}
            for dbg, _ in dbgs:
                tuning(tile, dbg)
                _lib.gemm([(a, K, K)], w, out, M=M, N=N, compute=_lib.BF16, epilogue=_lib.EPI_RESID, resid=res)
                torch.cuda.synchronize()
                st = torch.cuda.Stream()
                gr = torch.cuda.CUDAGraph()
                with torch.cuda.stream(st):
                    with torch.cuda.graph(gr, stream=st):
                        for _ in range(REPS):
                            _lib.gemm([(a, K, K)], w, out, M=M, N=N, compute=_lib.BF16, epilogue=_lib.EPI_RESID, resid=res)
                graphs[dbg] = gr
            best = {d: 1e9 for d, _ in dbgs}
            for _ in range(5):
                for d, _n in dbgs:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    graphs[d].replay()
                    e1.record()
                    torch.cuda.synchronize()
                    best[d] = min(best[d], e0.elapsed_time(e1) * 1e3 / REPS)
            print(f"{spec:>16s} {tname:13s} | " + "  ".join(f"{n} {best[d]:6.2f}" for d, n in dbgs) + " us", flush=True)
    _lib.check(_lib.lib().v2a_set_tuning(None))


if __name__ == "__main__":
    main()
