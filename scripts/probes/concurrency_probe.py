#!/usr/bin/env python3
"""Is a sample reproducible while ANOTHER process keeps the same GPU busy?  (debug aid for the two-rank rehearsal)  The parent computes a
reference sample on a quiet device, starts one child process that runs matmuls and copies in a loop on the same GPU, repeats the sample and
prints the largest difference; then does the same with engine switches toggled.
usage: python scripts/probes/concurrency_probe.py [mode] [repeats]"""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

if len(sys.argv) > 1 and sys.argv[1] == "--load":
    import torch
    kind = sys.argv[3] if len(sys.argv) > 3 else "both"        # both | matmul | fill | idle
    a = torch.randn(2048, 2048, device="cuda", dtype=torch.bfloat16)
    big = torch.empty(64 << 20, device="cuda")
    t_end = time.time() + float(sys.argv[2])
    ready = sys.argv[4] if len(sys.argv) > 4 else None         # file written once the first matmuls have completed on the device
    while time.time() < t_end:
        if kind == "idle":
            time.sleep(0.1)
            continue
        for _ in range(20):
            if kind in ("both", "matmul"):
                b = a @ a
            if kind in ("both", "fill"):
                big.fill_(1.0)
        torch.cuda.synchronize()
        if ready:
            open(ready, "w").write("ready\n")
            ready = None
    sys.exit(0)

import torch  # noqa: E402
import v2a_amd  # noqa: E402
from v2a_amd import _lib as L  # noqa: E402
from v2a_amd.synth import random_state_dict, synthetic_conditioning  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "bf16x3"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 12
full = len(sys.argv) > 3 and sys.argv[3] == "full"          # the shipped shape: depth 12, dims 1024 / 1280 / 512, one clip of 750 frames
dev = torch.device("cuda:0")
if full:
    cfg = v2a_amd.DiTConfig()
    T, NC, steps, n = 750, 16, 6, 1
else:
    cfg = v2a_amd.DiTConfig(dim=256, dim_text=320, dim_frames=128, depth=4, heads=4, frames_heads=2, num_registers=8, num_channels=32, max_seq_len=512)
    T, NC, steps, n = 120, 12, 8, 5
sd = random_state_dict(cfg, seed=0, device="cpu")
tk = {k: v for k, v in cfg.to_dict().items() if k not in ("num_channels", "notes", "cond_proj_in", "dim_context", "kernel_size", "ff_mult")}
y0, text, roll, ctx, cm = synthetic_conditioning(cfg, n, T, NC, seed=77, piano=True, device="cpu")
kw = dict(steps=steps, cfg_strength=2.0, remove_parallel_component=False, sway_sampling=True, return_raw_output=True)


def trial(name, graph=True, tuning=None, **sw):
    if tuning:
        L.set_tuning(**tuning)
    m = v2a_amd.E2TTS(transformer=dict(if_text_modules=True, if_cross_attn=True, if_audio_conv=True, if_text_conv=True, **tk),
                      num_channels=cfg.num_channels, sampling_rate=24000, if_cond_proj_in=False, compute_dtype=mode, device=dev, use_graph=graph)
    m.load_state_dict(sd, strict=False)
    for k, v in sw.items():
        setattr(m.engine(), k, v)
    run = lambda: m.sample(torch.zeros(n, T, cfg.num_channels), y0=y0, text_embed=text, context=ctx, context_mask=cm, frames_embed=roll, **kw).float().cpu()
    ref = run()
    quiet = max(float((run() - ref).abs().max()) for _ in range(3))
    child = subprocess.Popen([sys.executable, os.path.abspath(__file__), "--load", "60"])
    time.sleep(4.0)                       # the child's first import of torch
    worst, bad = 0.0, 0
    for _ in range(reps):
        d = float((run() - ref).abs().nan_to_num(9e9).max())
        worst = max(worst, d)
        bad += d != 0.0
    child.kill()
    child.wait()
    print(f"{name:34s} quiet device: {quiet:.3e}   beside another process: worst {worst:.3e}, {bad} of {reps} runs differ", flush=True)
    if tuning:
        L.set_tuning()


trial("default")
if not full:
    trial("no graph", graph=False)
    trial("single stream", multi_stream=False)
    trial("fold_norm off", fold_norm=False)
