#!/usr/bin/env python3
"""Does a clip's result depend on the batch it is sampled in?  (debug aid)  Samples clips [0,3) and [0,5) of one seeded batch on the small
rehearsal model and prints max |delta| of the common clips, with engine switches toggled one at a time.
usage: python scripts/probes/batch_indep_probe.py [mode]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
import v2a_amd  # noqa: E402
from v2a_amd import _lib as L  # noqa: E402
from v2a_amd.synth import random_state_dict, synthetic_conditioning  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "bf16x3"
dev = torch.device("cuda:0")
cfg = v2a_amd.DiTConfig(dim=256, dim_text=320, dim_frames=128, depth=4, heads=4, frames_heads=2, num_registers=8, num_channels=32, max_seq_len=512)
T, NC, steps, n = 120, 12, 8, 5
sd = random_state_dict(cfg, seed=0, device="cpu")
tk = {k: v for k, v in cfg.to_dict().items() if k not in ("num_channels", "notes", "cond_proj_in", "dim_context", "kernel_size", "ff_mult")}
y0, text, roll, ctx, cm = synthetic_conditioning(cfg, n, T, NC, seed=77, piano=True, device="cpu")
kw = dict(steps=steps, cfg_strength=2.0, remove_parallel_component=False, sway_sampling=True, return_raw_output=True)


def trial(name, graph=True, **sw):
    tun = sw.pop("tuning", None)
    if tun:
        L.set_tuning(**tun)
    m = v2a_amd.E2TTS(transformer=dict(if_text_modules=True, if_cross_attn=True, if_audio_conv=True, if_text_conv=True, **tk),
                      num_channels=cfg.num_channels, sampling_rate=24000, if_cond_proj_in=False, compute_dtype=mode, device=dev, use_graph=graph)
    m.load_state_dict(sd, strict=False)
    for k, v in sw.items():
        setattr(m.engine(), k, v)

    def run(lo, hi):
        return m.sample(torch.zeros(hi - lo, T, cfg.num_channels), y0=y0[lo:hi], text_embed=text[lo:hi], context=ctx[lo:hi], context_mask=cm[lo:hi],
                        frames_embed=roll[lo:hi], **kw).float().cpu()
    a, b, w = run(0, 3), run(3, 5), run(0, 5)
    a2 = run(0, 3)              # the same shape again: plan and graph cache hit
    w2 = run(0, 5)
    print(f"{name:40s} clips 0-2: {float((a - w[:3]).abs().max()):.3e}   clips 3-4: {float((b - w[3:]).abs().max()):.3e}   "
          f"second call of a shape: {float((a2 - a).abs().max()):.3e} / {float((w2 - w).abs().max()):.3e}", flush=True)
    if tun:
        L.set_tuning()


def poison(val):
    """fill the caching allocator's free memory with `val`: buffers from torch.empty then start out poisoned"""
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    t = torch.full((1 << 29,), val, device=dev)        # 2 GB, returned to the allocator's large pool: later allocations are carved from it
    del t


for val in (float("nan"), 1.0e4):
    for graph in (True, False):
        m = v2a_amd.E2TTS(transformer=dict(if_text_modules=True, if_cross_attn=True, if_audio_conv=True, if_text_conv=True, **tk),
                          num_channels=cfg.num_channels, sampling_rate=24000, if_cond_proj_in=False, compute_dtype=mode, device=dev, use_graph=graph)
        m.load_state_dict(sd, strict=False)
        m.engine()
        ref = m.sample(torch.zeros(5, T, cfg.num_channels), y0=y0, text_embed=text, context=ctx, context_mask=cm, frames_embed=roll, **kw).float().cpu()
        m2 = v2a_amd.E2TTS(transformer=dict(if_text_modules=True, if_cross_attn=True, if_audio_conv=True, if_text_conv=True, **tk),
                           num_channels=cfg.num_channels, sampling_rate=24000, if_cond_proj_in=False, compute_dtype=mode, device=dev, use_graph=graph)
        m2.load_state_dict(sd, strict=False)
        m2.engine()
        poison(val)
        got = m2.sample(torch.zeros(5, T, cfg.num_channels), y0=y0, text_embed=text, context=ctx, context_mask=cm, frames_embed=roll, **kw).float().cpu()
        print(f"poisoned allocator ({val}), graph {graph}: finite {bool(torch.isfinite(got).all())}, max |delta| vs clean run {float((got - ref).abs().nan_to_num(9e9).max()):.3e}", flush=True)
trial("default")
trial("no graph", graph=False)
trial("fold_norm off", fold_norm=False)
trial("fuse_xattn off", fuse_xattn=False)
trial("single stream", multi_stream=False)
trial("8-phase four phases", tuning=dict(eight_phase=1))
trial("no streaming dwconv", tuning=dict(dwconv_rows_per_wave=-1))
