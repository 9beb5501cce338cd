#!/usr/bin/env python3
"""[Round-5 note: grouped launches (ABI 7) left the library with ABI 8; this probe runs against commit 88e6ef7 (round 4) and is kept for its record under profiles/.]
Time the sampler's GROUPED launches (v2a_gemm_grouped / v2a_attention_grouped / v2a_dwconv_grouped) stand-alone, per group and per
tile shape, each inside a hipGraph of back-to-back launches on random data (interleaved rounds, best of 5).  One clip of the shipped
widths unless --clips says otherwise: M = 2 * clips * 782 rows, streams a / t / f = 1024 / 1280 / 512 wide.

usage: python scripts/group_probe.py [--clips 1] [--members atf|tf|af|a|t|f ...] [--split]
Output: one line per (op, members): time per tile_hint, and the sum of the members' single launches with the library's own choice.
"""
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import v2a_amd  # noqa: E402,F401
from v2a_amd import _lib as L  # noqa: E402

DEV = "cuda"
DIMS = {"a": (1024, 16), "t": (1280, 16), "f": (512, 8)}
REPS = 20


def bf(*shape, scale=1.0):
    return (torch.randn(*shape) * scale).to(torch.bfloat16).to(DEV)


def planes(x32):
    hi = x32.to(torch.bfloat16)
    lo = (x32 - hi.float()).to(torch.bfloat16)
    return torch.cat([hi, lo], -1).contiguous().to(DEV)


def operand(M, k, split):
    return planes(torch.randn(M, k) * 0.5) if split else bf(M, k, scale=0.5)


def build(op, s, M, split):
    """gemm_args of op for stream s with the sampler's shapes and epilogue features."""
    d, H = DIMS[s]
    inner = H * 64
    lda = lambda k: 2 * k if split else k
    kw = dict(M=M, compute=L.BF16)
    if split:
        kw["a_split"] = True
    if op == "cross":
        ks = {"a": (2048, 1280, 512), "t": (1024, 1280), "f": (1024, 512)}[s]        # audio: the fused cross-condition + skip GEMM (bf16 mode)
        if split and s == "a":
            ks = (1024, 1280, 512)
        N = d
        res = torch.randn(M, N, device=DEV)
        kw.update(N=N, epilogue=L.EPI_RESID, resid=res)
        out = res
    elif op == "qkv":
        ks, N = (d,), 3 * inner + 16
        out = torch.empty(M, N, device=DEV, dtype=torch.float32 if split else torch.bfloat16)
        ang = torch.arange(790).float()[:, None] * (1.0 / (10000 ** (torch.arange(0, 64, 2).float() / 64)))[None]
        tab = torch.stack((ang.cos(), ang.sin()), -1).contiguous().to(DEV)
        ssq = torch.full((M, 40), 20.0, device=DEV)
        kw.update(N=N, bias=torch.zeros(N, device=DEV), rope_table=tab, rope_cols=2 * inner, rows_per_batch=782, row_ssq=ssq, row_norm_dim=d)
    elif op == "out":
        ks, N = (inner,), d
        res = torch.randn(M, N, device=DEV)
        sh = torch.empty(M, 2 * N if split else N, device=DEV, dtype=torch.bfloat16)
        kw.update(N=N, epilogue=L.EPI_RESID, resid=res, out_bf16=sh, norm_gamma=torch.ones(N, device=DEV), norm_ssq=torch.zeros(M, 40, device=DEV))
        if split:
            kw.update(out_bf16_split=True, ld_out_bf16=2 * N)
        if s == "a":
            kw.update(epilogue=L.EPI_GATE_RESID, gate=torch.rand(N, device=DEV))
        out = res
    elif op == "ff1":
        ks, N = (d,), 8 * d
        out = torch.empty(M, N if split else N // 2, device=DEV, dtype=torch.bfloat16)
        kw.update(N=N, epilogue=L.EPI_GEGLU, bias=torch.zeros(N, device=DEV), ldo=out.stride(0), row_ssq=torch.full((M, 40), 20.0, device=DEV), row_norm_dim=d)
        if split:
            kw.update(out_split=True)
    else:   # ff2
        ks, N = (4 * d,), d
        res = torch.randn(M, N, device=DEV)
        sh = torch.empty(M, 2 * N if split else N, device=DEV, dtype=torch.bfloat16)
        kw.update(N=N, epilogue=L.EPI_RESID, resid=res, bias=torch.zeros(N, device=DEV), out_bf16=sh)
        if split:
            kw.update(out_bf16_split=True, ld_out_bf16=2 * N)
        if s == "a":
            kw.update(epilogue=L.EPI_GATE_RESID, gate=torch.rand(N, device=DEV))
        out = res
    K = sum(ks)
    segs = [(operand(M, k, split), lda(k), k) for k in ks]
    w = planes(torch.randn(kw["N"], K) / math.sqrt(K)) if split else bf(kw["N"], K, scale=1 / math.sqrt(K))
    flops = 2.0 * M * kw["N"] * K
    return (segs, w, out, kw), flops


def graph_of(fn):
    fn()
    torch.cuda.synchronize()
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st):
            for _ in range(REPS):
                fn()
    return g


def best_of(graphs):
    best = {k: 1e9 for k in graphs}
    for _ in range(5):
        for k, g in graphs.items():
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            g.replay()
            e1.record()
            torch.cuda.synchronize()
            best[k] = min(best[k], e0.elapsed_time(e1) * 1e3 / REPS)
    return best


def main():
    args = sys.argv[1:]
    clips, members, split = 1, ["atf", "tf", "af"], False
    i = 0
    while i < len(args):
        if args[i] == "--clips":
            clips = int(args[i + 1]); i += 2
        elif args[i] == "--members":
            members = args[i + 1].split(","); i += 2
        elif args[i] == "--split":
            split = True; i += 1
        else:
            raise SystemExit("unknown argument " + args[i])
    torch.manual_seed(0)
    M = 2 * clips * 782
    hints = [1, 2, 3, 4, 5] if split else [13, 15, 16, 2, 4, 1, 7]
    print("clips %d (M = %d rows), %s operands; tile hints: %s" % (clips, M, "split" if split else "bf16",
          "1 64x64, 2 128x64, 3 128x128/8w, 4 64x128/8w, 5 8-phase" if split else "13 128x128/8w, 15 128x64/8w, 16 64x128/8w, 2 128x128, 4 64x64, 1 128x256, 7 8-phase"))
    for op in ("cross", "qkv", "out", "ff1", "ff2"):
        built = {s: build(op, s, M, split) for s in "atf"}
        single = {}
        for s in "atf":
            (segs, w, out, kw), _ = built[s]
            single[s] = graph_of(lambda segs=segs, w=w, out=out, kw=kw: L.gemm(segs, w, out, **kw))
        tsingle = best_of(single)
        for mem in members:
            graphs = {}
            for h in hints:
                if split and h == 5 and op == "cross":
                    continue
                probs = [L.gemm_args(*built[s][0][:3], **built[s][0][3]) for s in mem]
                try:
                    graphs[h] = graph_of(lambda probs=probs, h=h: L.gemm_grouped(probs, tile_hint=h))
                except L.V2AError:
                    continue
            t = best_of(graphs)
            fl = sum(built[s][1] for s in mem) * (3 if split else 1)
            line = "  ".join("h%-2d %6.1f" % (h, t[h]) for h in t)
            bh = min(t, key=t.get)
            print("%-5s %-3s | %s | best h%d %.1f us = %.0f TF/s | singles %s = %.1f us" % (
                op, mem, line, bh, t[bh], fl / t[bh] / 1e6, "+".join("%.1f" % tsingle[s] for s in mem), sum(tsingle[s] for s in mem)), flush=True)
    # attention and convolution groups
    if not split:
        N = 782
        Bt = 2 * clips
        bufs = {s: bf(Bt * N, 3 * DIMS[s][1] * 64 + 16) for s in "atf"}
        aos = {s: torch.empty(Bt * N, DIMS[s][1] * 64, device=DEV, dtype=torch.bfloat16) for s in "atf"}

        def aargs(s):
            H = DIMS[s][1]
            inner, npad = H * 64, 3 * H * 64 + 16
            base = bufs[s].data_ptr()
            return L.attention_args(base, base + inner * 2, base + 2 * inner * 2, base + 3 * inner * 2, aos[s].data_ptr(),
                                    strides=(npad, npad, npad, npad, inner, N * npad, N * npad, N * npad, N * npad, N * inner), B=Bt, H=H, Nq=N, Nk=N,
                                    scale=0.125, softclamp=50.0, dtype=L.BF16)
        for mem in members + ["a", "t", "f"]:
            for og in (0, 1):
                L.set_tuning(attn_one_group_from=og)
                g = {og: graph_of(lambda mem=mem: L.attention_grouped([aargs(s) for s in mem]))}
                L.set_tuning()
                print("attn  %-3s one_group_from=%d | %.1f us" % (mem, og, best_of(g)[og]), flush=True)
        xs = {s: torch.randn(Bt, N, DIMS[s][0], device=DEV) for s in "atf"}
        for mem in members + ["a", "t", "f"]:
            items = []
            for s in mem:
                d = DIMS[s][0]
                items.append(dict(x=xs[s], out=torch.empty_like(xs[s]), wt=torch.randn(31, d, device=DEV) / 5, bias=torch.zeros(d, device=DEV), d=d,
                                  norm=dict(out_bf16=torch.empty(Bt * N, d, device=DEV, dtype=torch.bfloat16), ld_out_bf16=d, gamma=torch.ones(d, device=DEV),
                                            ssq=torch.zeros(Bt * N, 40, device=DEV))))
            g = {0: graph_of(lambda items=items: L.dwconv_grouped(items, B=Bt, N=N, ksize=31))}
            print("conv  %-3s | %.1f us" % (mem, best_of(g)[0]), flush=True)


if __name__ == "__main__":
    main()
