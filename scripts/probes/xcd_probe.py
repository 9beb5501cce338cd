#!/usr/bin/env python3
"""[Round-5 note: XCD-subset placement (v2a_gemm_args.xcd_mask, ABI 5-7) left the library with ABI 8; this probe runs against commit 88e6ef7 (round 4) and is kept for its record under profiles/.]
XCD-subset placement probe (tuning aid): time v2a_gemm launches alone and CONCURRENTLY on separate streams, each
launch optionally confined to a subset of the 8 XCDs (v2a_gemm_args.xcd_mask) with a chosen tile shape.

A group is a '+'-joined list of specs  MxNxK[:tile[:mask[:epi]]]  (tile = tile_hint - 1 of v2a_gemm: 0 128x256, 1 128x128,
2 128x64, 3 64x64, 6 8-phase 256x256, -1 library choice; mask = hex XCD mask, ff = default placement, 1ff = all eight XCDs
through the claim path; epi = resid | geglu | store).  Every member of a group runs `reps` back-to-back launches on its own
stream inside ONE hipGraph (fork / join on the capture stream); the group's wall time per rep is printed next to the members'
stand-alone times, so "3 GEMMs beside each other on disjoint XCDs" can be compared with "3 GEMMs spread over all XCDs".
usage: python scripts/probes/xcd_probe.py 1564x1024x4096:3:ff 1564x1024x4096:1:0f 1564x1024x4096:1:0f+1564x1280x5120:1:f0 ...
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
import v2a_amd  # noqa: E402,F401
from v2a_amd import _lib  # noqa: E402

DEV = torch.device("cuda:0")
REPS = 20


class Member:
    def __init__(self, spec, idx):
        f = spec.split(":")
        self.spec = spec
        self.M, self.N, self.K = (int(v) for v in f[0].split("x"))
        self.tile = int(f[1]) if len(f) > 1 and f[1] != "" else -1
        self.mask = int(f[2], 16) if len(f) > 2 and f[2] != "" else 0xFF
        self.epi = f[3] if len(f) > 3 else "resid"
        g = torch.Generator().manual_seed(idx)
        M, N, K = self.M, self.N, self.K
        self.split = self.epi.startswith("s3")          # split (hi | lo plane) operands: K is the LOGICAL K, three products per step
        if self.split:
            self.epi = self.epi[2:] or "resid"
        kk = 2 * K if self.split else K
        self.a = (torch.randn(M, kk, generator=g) * 0.5).to(DEV, torch.bfloat16)
        self.w = (torch.randn(N, kk, generator=g) * 0.05).to(DEV, torch.bfloat16)
        self.ctr = torch.zeros(16, dtype=torch.int32, device=DEV)
        if self.epi == "store" and self.split:
            self.out = torch.empty(M, N, device=DEV)
            self.kw = dict()
        elif self.epi == "geglu" and self.split:
            self.out = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
            self.kw = dict(epilogue=_lib.EPI_GEGLU, ldo=N, out_split=True)
        elif self.epi == "geglu":
            self.out = torch.empty(M, N // 2, device=DEV, dtype=torch.bfloat16)
            self.kw = dict(epilogue=_lib.EPI_GEGLU, ldo=N // 2)
        elif self.epi == "store":
            self.out = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
            self.kw = dict()
        else:
            self.res = torch.randn(M, N, generator=g).to(DEV)
            self.out = torch.empty(M, N, device=DEV)
            self.kw = dict(epilogue=_lib.EPI_RESID, resid=self.res)
        self.flops = (6.0 if self.split else 2.0) * M * N * K

    def run(self):
        if self.split:
            _lib.gemm([(self.a, 2 * self.K, self.K)], self.w, self.out, M=self.M, N=self.N, compute=_lib.BF16, tile_hint=self.tile + 1,
                      a_split=True, **self.kw)
            return
        _lib.gemm([(self.a, self.K, self.K)], self.w, self.out, M=self.M, N=self.N, compute=_lib.BF16, tile_hint=self.tile + 1,
                  xcd_mask=self.mask, tile_counters=self.ctr, **self.kw)

    def check(self):
        if self.split:
            return True
        self.out.zero_()
        self.run()
        torch.cuda.synchronize()
        got = self.out.float().clone()
        mask, tile = self.mask, self.tile
        self.mask, self.tile = 0xFF, -1
        self.out.zero_()
        self.run()
        torch.cuda.synchronize()
        self.mask, self.tile = mask, tile
        ok = torch.equal(got, self.out.float())
        assert bool((self.ctr == 0).all()), self.ctr
        return ok


def graph_of(members, streams):
    cap = torch.cuda.Stream()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.stream(cap):
        with torch.cuda.graph(gr, stream=cap):
            fork = torch.cuda.Event()
            fork.record(cap)
            joins = []
            for m, s in zip(members, streams):
                s.wait_event(fork)
                with torch.cuda.stream(s):
                    for _ in range(REPS):
                        m.run()
                    e = torch.cuda.Event()
                    e.record(s)
                    joins.append(e)
            for e in joins:
                cap.wait_event(e)
    return gr


def time_graph(gr, rounds=5):
    best = 1e9
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        gr.replay()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / REPS)
    return best


def main():
    groups = sys.argv[1:]
    streams = [torch.cuda.Stream() for _ in range(4)]
    for gi, gs in enumerate(groups):
        members = [Member(s, 100 * gi + i) for i, s in enumerate(gs.split("+"))]
        oks = [m.check() for m in members]
        alone = [time_graph(graph_of([m], streams[:1])) for m in members]
        if len(members) > 1:
            tg = time_graph(graph_of(members, streams))
            tf = sum(m.flops for m in members) / tg / 1e6
            print(f"GROUP {gs}: {tg:7.2f} us per round of {len(members)} ({tf:6.1f} TF/s together); alone: "
                  + "  ".join(f"{t:6.2f} us" for t in alone) + f" (sum {sum(alone):.2f}, max {max(alone):.2f})  bit-equal {oks}", flush=True)
        else:
            m = members[0]
            print(f"ALONE {gs:40s}: {alone[0]:7.2f} us  {m.flops / alone[0] / 1e6:6.1f} TF/s  bit-equal to default placement: {oks[0]}", flush=True)


if __name__ == "__main__":
    main()
