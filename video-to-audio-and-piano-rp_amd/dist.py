"""Clip-level sharding across the GPUs of one node.

Clips are independent (no cross-sample op anywhere in Transformer.forward, x3:941-1143), so
the path shards with NO per-step communication: rank r samples a contiguous slice of the batch
with replicated weights and ONE all-gather (RCCL over xGMI on GPUs; gloo in the CPU tests)
reassembles the (B, T, C) fp32 latents -- 0.38 MB per clip.  The reference has no equivalent
(manual start/end argv sharding, src/inference_v2a.py:7-8).
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def shard_range(n_clips: int, rank: int, world: int) -> tuple[int, int, int]:
    """Contiguous equal shards; the batch is padded up to a multiple of `world`.
    Returns (start, end, per_rank) with end clipped to n_clips (pad clips are dropped)."""
    per = (n_clips + world - 1) // world
    start = min(rank * per, n_clips)
    return start, min(start + per, n_clips), per


def gather_latents(local: torch.Tensor, n_clips: int, per_rank: int) -> torch.Tensor:
    """One all-gather of the per-rank latents -> (n_clips, T, C) on every rank."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return local[:n_clips]
    world = dist.get_world_size()
    T, C = local.shape[1:]
    dev = local.device
    if dist.get_backend() != "nccl" and local.is_cuda:
        # gloo rehearsal of the N-rank path on GPUs (CPU collectives): stage through the host explicitly.  Handing gloo a
        # CUDA tensor while hipGraph replays are still queued stalled for ~25 s per call when two ranks shared one device.
        local = local.cpu()
    buf = torch.zeros(per_rank, T, C, dtype=local.dtype, device=local.device)
    buf[: local.shape[0]] = local
    out = torch.empty(world * per_rank, T, C, dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, buf)
    return out[:n_clips].to(dev)
