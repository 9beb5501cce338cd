#!/usr/bin/env python3
"""What bounds the K loop of the split-operand ring GEMM (bf16x3 mode) at one clip?  Needs the instrumented library
(`bash video-to-audio-and-piano-rp_amd/csrc/build.sh --probe`).  Every variant inside a hipGraph of 20 back-to-back launches, best of 5,
with parts of the K loop switched off by bits of v2a_tuning.reserved[0] (probe builds only):
   1  no LDS fragment reads, no MFMAs: the operand stream alone (DMA issue, counted wait, barrier)
   2  no DMA and no wait inside the K loop: fragment reads + MFMAs + barrier alone, on whatever the prologue staged
  32  hi x hi products only (one MFMA per fragment pair instead of three; all four planes still staged and read)
Results of the masked variants are wrong by construction; only the durations mean something.
usage: python scripts/probes/kloop_split_probe.py [--tiles 4,1,7] [MxNxK ...]"""
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


def planes(x):
    hi = x.bfloat16()
    return torch.cat([hi, (x - hi.float()).bfloat16()], -1).contiguous()


def main():
    args = sys.argv[1:]
    tiles = ["4", "1", "7"]
    if args and args[0] == "--tiles":
        tiles, args = args[1].split(","), args[2:]
    shapes = args or ["1564x1024x4096", "1564x1024x1024", "1564x1280x5120"]
    dbgs = [(0, "full"), (1, "stream only"), (2, "compute only"), (32, "one product"), (34, "compute only, one product")]
    for spec in shapes:
        M, N, K = (int(v) for v in spec.split("x"))
        g = torch.Generator().manual_seed(0)
        a = planes(torch.randn(M, K, generator=g) * 0.5).to(DEV)
        w = planes(torch.randn(N, K, generator=g) * 0.05).to(DEV)
        res = torch.randn(M, N, generator=g).to(DEV)
        out = torch.empty(M, N, device=DEV)
        for t in tiles:
            graphs = {}
            for dbg, _ in dbgs:
                _lib.set_tuning(reserved=dbg)

                def call():
                    _lib.gemm([(a, 2 * K, K)], w, out, M=M, N=N, compute=_lib.BF16, a_split=True, tile_hint=int(t), epilogue=_lib.EPI_RESID, resid=res)
                call()
                torch.cuda.synchronize()
                gr = torch.cuda.CUDAGraph()
                with torch.cuda.graph(gr):
                    for _ in range(REPS):
                        call()
                graphs[dbg] = gr
            best = {d: 1e9 for d, _ in dbgs}
            for _ in range(5):
                for d, gr in graphs.items():
                    torch.cuda.synchronize()
                    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    s.record()
                    gr.replay()
                    e.record()
                    torch.cuda.synchronize()
                    best[d] = min(best[d], s.elapsed_time(e) / REPS * 1e3)
            print("%-18s tile %s  %s" % (spec, t, "  ".join("%s %.1f us" % (n, best[d]) for d, n in dbgs)), flush=True)
    _lib.set_tuning()


if __name__ == "__main__":
    main()
