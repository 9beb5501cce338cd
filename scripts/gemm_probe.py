#!/usr/bin/env python3
"""Time v2a_gemm on MxNxK shapes (bf16 operands) for several tile configurations in ONE process, each inside a hipGraph
of back-to-back launches on random data (interleaved rounds, best of 5: rule 24 of the CDNA guide).

Tuning aid.  Tile configurations are selected with v2a_set_tuning: 0 = 128x256, 1 = 128x128, 2 = 128x64, 3 = 64x64,
5 = 256x256 (2-deep ring), 6 = 256x256 8-phase (staggered), 7 = 256x256 8-phase (lock-step), 17 / 18 = 64x128 / 64x64 with a
6-deep ring, -1 = automatic.
usage: python scripts/gemm_probe.py [--tiles 0,1,3,6,7] [--epi resid|geglu|store] 1564x1024x1024 1564x8192x1024 ...
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import v2a_amd  # noqa: E402,F401
from v2a_amd import _lib  # noqa: E402

SAMPLER_SHAPES = ["1564x3088x1024", "1564x3088x1280", "1564x1552x512", "1564x8192x1024", "1564x10240x1280", "1564x4096x512",
                  "1564x1024x1024", "1564x1280x1024", "1564x512x512", "1564x1024x4096", "1564x1280x5120", "1564x512x2048",
                  "1564x1024x2816", "1564x1280x2304", "1564x512x1536", "1564x1024x2048", "782x1040x1024"]


def set_cfg(t):
    if t == 6:
        _lib.set_tuning(force_tile=6, eight_phase=1)
    elif t == 7:
        _lib.set_tuning(force_tile=6, eight_phase=2)
    elif t >= 17:                       # 17 / 18: 64x128 / 64x64 tiles with the 6-deep ring
        _lib.set_tuning(force_tile=t - 10)
    else:
        _lib.set_tuning(force_tile=t)


def main():
    args = sys.argv[1:]
    tiles, epi = [-1, 0, 1, 3, 6, 7], "resid"
    while args and args[0].startswith("--"):
        if args[0] == "--tiles":
            tiles = [int(v) for v in args[1].split(",")]
        elif args[0] == "--epi":
            epi = args[1]
        args = args[2:]
    specs = args or SAMPLER_SHAPES
    dev = torch.device("cuda:0")
    reps = 20
    for spec in specs:
        M, N, K = [int(v) for v in spec.split("x")[:3]]
        g = torch.Generator(device="cpu").manual_seed(0)
        a = (torch.randn(M, K, generator=g) * 0.5).to(dev, torch.bfloat16)
        w = (torch.randn(N, K, generator=g) * 0.05).to(dev, torch.bfloat16)
        res = torch.randn(M, N, generator=g).to(dev)
        if epi == "resid":
            out = torch.empty(M, N, device=dev)
            kw = dict(epilogue=_lib.EPI_RESID, resid=res)
            ref = res + a.float() @ w.float().t()
        elif epi in ("gate", "gate_prod"):             # GATE_RESID in place (+ folded-RMSNorm producer with a switch row)
            out = res.clone()
            gate = torch.rand(4, N, device=dev)
            step = torch.tensor([1], dtype=torch.int32, device=dev)
            kw = dict(epilogue=_lib.EPI_GATE_RESID, resid=out, gate=gate, step=step, gate_step_stride=N, rows_per_batch=782)
            if epi == "gate_prod":
                kw.update(out_bf16=torch.empty(M, N, device=dev, dtype=torch.bfloat16), norm_gamma=torch.ones(4, 2, N, device=dev),
                          norm_step_stride=2 * N, norm_switch_row=M // 2, norm_switch_offset=N, norm_ssq=torch.zeros(M, N // 32, device=dev))
            ref = None
        elif epi in ("resid_shadow", "resid_prod"):     # RESID + bf16 shadow / + folded-RMSNorm producer (gamma, sums of squares)
            out = torch.empty(M, N, device=dev)
            sh = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
            kw = dict(epilogue=_lib.EPI_RESID, resid=res, out_bf16=sh)
            if epi == "resid_prod":
                kw.update(norm_gamma=torch.ones(N, device=dev), norm_ssq=torch.zeros(M, N // 32, device=dev))
            ref = res + a.float() @ w.float().t()
        elif epi == "geglu":
            out = torch.empty(M, N // 2, device=dev, dtype=torch.bfloat16)
            kw = dict(epilogue=_lib.EPI_GEGLU, ldo=N // 2)
            ref = None
        else:
            out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
            kw = dict()
            ref = a.float() @ w.float().t()

        def run():
            _lib.gemm([(a, K, K)], w, out, M=M, N=N, compute=_lib.BF16, **kw)

        graphs, errs = {}, {}
        for t in tiles:
            set_cfg(t)
            out.zero_()
            run()
            torch.cuda.synchronize()
            errs[t] = float((out.float() - ref).abs().max()) if ref is not None else float("nan")
            st = torch.cuda.Stream()
            with torch.cuda.stream(st):
                gr = torch.cuda.CUDAGraph()
                with torch.cuda.graph(gr, stream=st):
                    for _ in range(reps):
                        run()
            graphs[t] = gr
        _lib.set_tuning()
        torch.cuda.synchronize()
        best = {t: 1e9 for t in tiles}
        for _ in range(5):
            for t in tiles:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                graphs[t].replay()
                e1.record()
                torch.cuda.synchronize()
                best[t] = min(best[t], e0.elapsed_time(e1) * 1e3 / reps)
        line = "  ".join("t%-2d %6.2f us %6.1f TF" % (t, best[t], 2.0 * M * N * K / best[t] / 1e6) for t in tiles)
        print(f"{spec:>18s} {epi:5s} | {line} | maxerr {max(e for e in errs.values() if e == e) if ref is not None else 0:.2e}", flush=True)


if __name__ == "__main__":
    main()
