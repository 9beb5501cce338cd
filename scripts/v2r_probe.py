#!/usr/bin/env python3
"""Time Video2RollEngine.encode_frames for a list of chunk sizes (tuning aid).  usage: python scripts/v2r_probe.py 25 50 126 251"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import v2a_amd  # noqa: E402,F401
from v2a_amd.synth import random_video2roll_state_dict, synthetic_piano_frames  # noqa: E402
from v2a_amd.video2roll import Video2RollEngine  # noqa: E402

sd = random_video2roll_state_dict(0)
x = synthetic_piano_frames(1, 251, seed=0).to("cuda")
for ch in [int(a) for a in sys.argv[1:]]:
    eng = Video2RollEngine(sd, "cuda", compute="bf16", chunk=ch)
    eng.encode_frames(x, 750)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        eng.encode_frames(x, 750)
    torch.cuda.synchronize()
    print(f"chunk {ch:4d}: {(time.perf_counter() - t0) / 3 * 1e3:7.2f} ms per 251-frame clip", flush=True)
    del eng
    torch.cuda.empty_cache()
