#!/usr/bin/env python3
"""Soak: the full-shape sampler run many times, every result compared bit for bit with the first (a rare race in a hand-written kernel
-- a stage restaged too early, a missing wait -- shows up as a run that differs).  usage: python scripts/probes/soak_probe.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
import v2a_amd  # noqa: E402
from v2a_amd.synth import random_state_dict, synthetic_conditioning  # noqa: E402

dev = torch.device("cuda:0")
cfg = v2a_amd.DiTConfig()
sd = random_state_dict(cfg, seed=0, device="cpu")
tk = {k: v for k, v in cfg.to_dict().items() if k not in ("num_channels", "notes", "cond_proj_in", "dim_context", "kernel_size", "ff_mult")}
# (mode, clips, runs)
for mode, clips, runs in (("bf16x3", 1, 120), ("bf16x3", 8, 20), ("bf16x3", 2, 30), ("bf16", 1, 200), ("bf16", 2, 40), ("bf16", 8, 40), ("bf16", 3, 40), ("fp32", 1, 20)):
    m = v2a_amd.E2TTS(transformer=dict(if_text_modules=True, if_cross_attn=True, if_audio_conv=True, if_text_conv=True, **tk),
                      num_channels=cfg.num_channels, sampling_rate=24000, if_cond_proj_in=False, compute_dtype=mode, device=dev, use_graph=True)
    m.load_state_dict(sd, strict=False)
    y0, text, roll, ctx, cm = synthetic_conditioning(cfg, clips, 750, 16, seed=5, piano=False, device="cpu")
    kw = dict(steps=32, cfg_strength=2.0, remove_parallel_component=False, sway_sampling=True, return_raw_output=True)
    run = lambda: m.sample(torch.zeros(clips, 750, cfg.num_channels), y0=y0, text_embed=text, context=ctx, context_mask=cm, frames_embed=roll, **kw)
    ref = run().clone()
    t0 = time.time()
    bad = sum(not torch.equal(run(), ref) for _ in range(runs))
    print(f"{mode:7s} {clips} clip(s), 32-point grid: {bad} of {runs} runs differ from the first ({time.time() - t0:.0f} s); finite {bool(torch.isfinite(ref).all())}", flush=True)
    del m
