#!/usr/bin/env python3
"""A/B of the ring GEMM kernel's K loop: plain (wait -> barrier -> fragment reads -> MFMAs per step) against the software-pipelined form
(v2a_tuning.reserved bit 7: the reads of tile k+1 land under the MFMAs of tile k), per shape and tile shape, each inside a hipGraph of
back-to-back launches; results must be equal bit for bit (same K order per output element).
usage: python scripts/probes/ring_pipe_probe.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
import v2a_amd  # noqa: E402,F401
from v2a_amd import _lib as L  # noqa: E402

DEV = "cuda"
REPS = 20
SHAPES = [(1564, 1024, 1024), (1564, 1024, 4096), (1564, 1280, 5120), (1564, 1024, 3840), (1564, 512, 2048), (1564, 3088, 1024), (12512, 1024, 4096)]
HINTS = {13: "128x128/8w", 15: "128x64/8w", 16: "64x128/8w", 2: "128x128/4w", 4: "64x64/4w"}


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


def best(g):
    b = 1e9
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
        b = min(b, e0.elapsed_time(e1) * 1e3 / REPS)
    return b


torch.manual_seed(0)
for M, N, K in SHAPES:
    a = (torch.randn(M, K) * 0.5).to(torch.bfloat16).to(DEV)
    w = (torch.randn(N, K) * 0.03).to(torch.bfloat16).to(DEV)
    res = torch.randn(M, N, device=DEV)
    gate = torch.rand(N, device=DEV)
    line = []
    for h, name in HINTS.items():
        outs, times = [], []
        for pipe in (False, True):
            L.set_tuning(ring_pipe=pipe)
            out = torch.empty(M, N, device=DEV)
            fn = lambda: L.gemm([(a, K, K)], w, out, M=M, N=N, compute=L.BF16, epilogue=L.EPI_GATE_RESID, resid=res, gate=gate, tile_hint=h)
            g = graph_of(fn)          # captured while the tuning bit is set: the choice is made at launch (capture) time
            times.append(best(g))
            fn()
            torch.cuda.synchronize()
            outs.append(out.clone())
        L.set_tuning()
        eq = torch.equal(outs[0], outs[1])
        line.append("%s %6.1f -> %6.1f us (%+5.1f %%)%s" % (name, times[0], times[1], 100 * (times[0] / times[1] - 1), "" if eq else " MISMATCH"))
    print("%5dx%4dx%4d | %s" % (M, N, K, " | ".join(line)), flush=True)
