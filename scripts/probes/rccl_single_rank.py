#!/usr/bin/env python3
"""RCCL sanity on the one-GPU box: a one-rank "nccl" process group, the sampler with its hipGraph on, then all_gather_into_tensor on the
device tensor (the collective of dist.gather_latents).  N > 1 ranks need N GPUs; this only shows that RCCL initialises and runs beside the
captured graphs in this environment.  usage: python scripts/probes/rccl_single_rank.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29577")
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
t0 = time.time()
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
import v2a_amd  # noqa: E402
from v2a_amd.synth import random_state_dict, synthetic_conditioning  # noqa: E402

cfg = v2a_amd.DiTConfig(dim=256, dim_text=320, dim_frames=128, depth=4, heads=4, frames_heads=2, num_registers=8, num_channels=32, max_seq_len=512)
sd = random_state_dict(cfg, seed=0, device="cpu")
tk = {k: v for k, v in cfg.to_dict().items() if k not in ("num_channels", "notes", "cond_proj_in", "dim_context", "kernel_size", "ff_mult")}
m = v2a_amd.E2TTS(transformer=dict(if_text_modules=True, if_cross_attn=True, if_audio_conv=True, if_text_conv=True, **tk),
                  num_channels=cfg.num_channels, sampling_rate=24000, if_cond_proj_in=False, compute_dtype="bf16", device=dev, use_graph=True)
m.load_state_dict(sd, strict=False)
y0, text, roll, ctx, cm = synthetic_conditioning(cfg, 3, 120, 12, seed=77, piano=True, device="cpu")
kw = dict(steps=8, cfg_strength=2.0, remove_parallel_component=False, sway_sampling=True, return_raw_output=True)
outs = []
for rep in range(3):
    mine = m.sample(torch.zeros(3, 120, cfg.num_channels), y0=y0, text_embed=text, context=ctx, context_mask=cm, frames_embed=roll, **kw).to(dev)
    got = torch.empty_like(mine)
    dist.all_gather_into_tensor(got, mine.contiguous())
    outs.append(got.cpu())
torch.cuda.synchronize()
assert torch.equal(outs[0], outs[1]) and torch.equal(outs[1], outs[2]) and bool(torch.isfinite(outs[0]).all())
dist.barrier()
dist.destroy_process_group()
print(f"RCCL one-rank group: init + 3 x (sample with hipGraph, all_gather_into_tensor) ok in {time.time() - t0:.1f} s; graph captures {m.graph_captures}")
