#!/usr/bin/env python3
"""Time v2a_gemm with split (hi | lo plane) operands -- the bf16x3 mode's GEMMs -- on MxNxK shapes for several tile configurations in ONE
process: each configuration inside a hipGraph of back-to-back launches on random data, interleaved rounds, best of 5 (rule 24 of the
CDNA guide).  K may be a '+'-joined list of logical segments (cross-condition: 1024+1280+512).

tile_hint: 0 = by shape, 1 = 64x64, 2 = 128x64, 3 = 128x128 / 8 waves / 2 stages, 4 = 64x128 / 8 waves, 5 = the 256x256 8-phase kernel on
three passes, 6 = 128x256 / 8 waves (32-wide K stages), 7 = 128x128 / 8 waves (32-wide K stages).
(profiles/r05_split_probe.txt was taken with two more shapes in the build: there t7 / t9 = 64x128 / 4 waves on 32-wide stages with 3 / 4 stages, and t8 = today's 7.)
usage: python scripts/split_probe.py [--tiles 0,4,5,6,7] [--epi resid|store|geglu|gate] 1564x1024x4096 1564x1024x1024+1280+512 ...
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import v2a_amd  # noqa: E402,F401
from v2a_amd import _lib as L  # noqa: E402

ONE_CLIP = ["1564x3088x1024", "1564x3088x1280", "1564x1552x512", "1564x8192x1024", "1564x10240x1280", "1564x4096x512",
            "1564x1024x1024", "1564x1280x1280", "1564x512x512", "1564x1024x4096", "1564x1280x5120", "1564x512x2048",
            "1564x1024x1024+1280+512", "1564x1280x1024+1280", "1564x512x1024+512", "1564x1024x1024+1024", "782x1024x1024"]


def planes(x):
    hi = x.bfloat16()
    return torch.cat([hi, (x - hi.float()).bfloat16()], -1).contiguous()


def main():
    args = sys.argv[1:]
    tiles, epi = ["0", "4", "5", "6", "7"], "resid"
    while args and args[0].startswith("--"):
        if args[0] == "--tiles":
            tiles = args[1].split(",")
        elif args[0] == "--epi":
            epi = args[1]
        args = args[2:]
    dev = torch.device("cuda:0")
    reps = 20
    for spec in args or ONE_CLIP:
        M, N, ks = spec.split("x")
        M, N, ks = int(M), int(N), [int(k) for k in ks.split("+")]
        K = sum(ks)
        g = torch.Generator(device="cpu").manual_seed(0)
        a = [torch.randn(M, k, generator=g) * 0.5 for k in ks]
        w = torch.randn(N, K, generator=g) * 0.05
        segs = [(planes(x).to(dev), 2 * k, k) for x, k in zip(a, ks)]
        wd = planes(w).to(dev)
        res = torch.randn(M, N, generator=g).to(dev)
        ref = torch.cat(a, 1).double() @ w.double().t()
        if epi == "resid":
            out = torch.empty(M, N, device=dev)
            kw = dict(epilogue=L.EPI_RESID, resid=res)
            ref = res.cpu().double() + ref
        elif epi == "gate":
            out = res.clone()
            gate = torch.rand(N, device=dev)
            kw = dict(epilogue=L.EPI_GATE_RESID, resid=out, gate=gate)
            ref = None
        elif epi == "geglu":
            out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
            kw = dict(epilogue=L.EPI_GEGLU, ldo=N, out_split=True)
            ref = None
        else:
            out = torch.empty(M, N, device=dev)
            kw = {}
        graphs, errs = {}, {}
        for t in tiles:
            def call():
                L.gemm(segs, wd, out, M=M, N=N, compute=L.BF16, a_split=True, tile_hint=int(t), **kw)
            try:
                call()
                torch.cuda.synchronize()
            except L.V2AError as e:
                print("   tile %s: %s" % (t, str(e)[:100]))
                continue
            if ref is not None and epi in ("resid", "store"):
                errs[t] = float((out.cpu().double() - ref).abs().max())
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr):
                for _ in range(reps):
                    call()
            graphs[t] = gr
        best = {t: 1e9 for t in graphs}
        for _ in range(5):
            for t, gr in graphs.items():
                torch.cuda.synchronize()
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                gr.replay()
                e.record()
                torch.cuda.synchronize()
                best[t] = min(best[t], s.elapsed_time(e) / reps * 1e3)
        fl = 6.0 * M * N * K
        print("%-28s %s" % (spec + " " + epi, "  ".join("t%s %6.1f us %4.0f TF%s" % (t, us, fl / us / 1e6, (" e%.0e" % errs[t]) if t in errs else "") for t, us in best.items())), flush=True)


if __name__ == "__main__":
    main()
