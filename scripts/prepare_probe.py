#!/usr/bin/env python3
"""Where does one sample() spend its time outside the 31 graph replays?  (tuning aid)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import v2a_amd  # noqa: E402
from v2a_amd.synth import random_state_dict, synthetic_conditioning  # noqa: E402

dev = torch.device("cuda:0")
cfg = v2a_amd.DiTConfig()
model = v2a_amd.E2TTS(transformer=dict(depth=cfg.depth, dim=cfg.dim, dim_text=cfg.dim_text, heads=cfg.heads, dim_head=cfg.dim_head,
                                       if_text_modules=True, if_cross_attn=True, if_audio_conv=True, if_text_conv=True),
                      num_channels=cfg.num_channels, sampling_rate=24000, if_cond_proj_in=False, compute_dtype="bf16", device=dev)
model.load_state_dict(random_state_dict(cfg, seed=0, device=dev), strict=False)
y0, text, roll, ctx, cm = synthetic_conditioning(cfg, 1, 750, 16, seed=1, device=dev)
cond = torch.empty(1, 750, cfg.num_channels, device=dev)
kw = dict(y0=y0, text_embed=text, context=ctx, context_mask=cm.cpu(), frames_embed=roll, steps=32, cfg_strength=2.0,
          remove_parallel_component=False, sway_sampling=True, return_raw_output=True)
for _ in range(2):
    model.sample(cond, **kw)
torch.cuda.synchronize()
eng = model.engine()
orig_prepare, orig_run = eng.prepare, model._run_steps
acc = {"prepare": 0.0, "run": 0.0}


def timed(name, fn):
    def w(*a, **k):
        torch.cuda.synchronize()
        t = time.perf_counter()
        r = fn(*a, **k)
        torch.cuda.synchronize()
        acc[name] += time.perf_counter() - t
        return r
    return w


eng.prepare = timed("prepare", orig_prepare)
model._run_steps = timed("run", orig_run)
N = 5
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(N):
    model.sample(cond, **kw)
torch.cuda.synchronize()
tot = (time.perf_counter() - t0) / N * 1e3
print(f"sample {tot:.2f} ms = prepare {acc['prepare'] / N * 1e3:.2f} + euler loop {acc['run'] / N * 1e3:.2f} + rest {tot - (acc['prepare'] + acc['run']) / N * 1e3:.2f}")

# ---- is the Euler loop bound by the host enqueueing the graph, or by the GPU executing it? ----
eng.prepare, model._run_steps = orig_prepare, orig_run
model.sample(cond, **kw)
torch.cuda.synchronize()
g = next(iter(model._graphs.values()))
for reps in (1, 31):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        g.replay()
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    print(f"{reps:2d} replay(s): host enqueue {t_host * 1e3 / reps:.3f} ms per replay, until the GPU is done {t_all * 1e3 / reps:.3f} ms per replay")
