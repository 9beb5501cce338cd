#!/usr/bin/env python3
"""Where does v2a_linear_small differ beside another GPU process?  (debug aid)"""
import os
import subprocess
import sys
import time

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
variants = {"full": dict(dup=5, regs=regs, add=pos), "no dup": dict(dup=0, regs=regs, add=pos), "no regs": dict(dup=5, regs=None, add=pos),
            "no add": dict(dup=5, regs=regs, add=None)}


def run(v, fillv=0.0):
    out = torch.full((10, 128, 256), fillv, device=DEV)
    L.linear_small(y, wt_in, b_in, v["add"], out, M=5 * 120, K=32, T=120, out_batch_stride=128 * 256, row_off=8, d=256, dup=v["dup"], regs=v["regs"])
    return out.cpu()


refs = {k: run(v) for k, v in variants.items()}
for kind in ("idle", "matmul", "fill", "both"):
    child = subprocess.Popen([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "concurrency_probe.py"), "--load", "30", kind])
    time.sleep(5.0)
    bad, lanes = 0, set()
    for it in range(200):
        o = run(variants["full"])
        df = (o - refs["full"]).abs()
        if float(df.max()) != 0.0:
            bad += 1
            lanes |= set((idx // 4) for idx in (df > 0).nonzero()[:, 2].unique().tolist())
    print(f"other process: {kind:7s}: {bad} of 200 launches differ; threads with wrong columns: {sorted(lanes)[:4]}..{max(lanes) if lanes else None}", flush=True)
    child.kill()
    child.wait()
