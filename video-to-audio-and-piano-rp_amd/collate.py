"""Batched inference collator + clip sharder.

Counterpart of the reference's per-clip collation, which always builds a batch of ONE
(predict.py:210-237, app.py:211-238; dataset form trainer_multigpus_alldatas3.py:1122-1146,
1363, 1377-1378).  It produces the same 8-tuple
    [text, mel, video_paths, mel_len, video_drop_prompt, audio_drop_prompt, frames, midis]
for B clips at once and carries the precomputed conditioning tensors the accelerated
sampler consumes (CLIP features, T5 context, piano roll), padded to the longest clip.
Shapes: latent rate 24000 / 320 = 75 frames/s (torch_tools.py:32-40), so a 10 s clip is 750.
"""
from __future__ import annotations

from dataclasses import dataclass, field

import torch

NOTES = 51


@dataclass
class ClipRequest:
    video_path: str
    prompt: str                       # "" -> video_drop_prompt (predict.py:316-345, `drop = len(prompt) == 0`)
    n_frames: int                     # latent frames (mel_len); 750 for 10 s
    clip_embed: torch.Tensor          # (n_frames, dim_text) CLIP features resampled to the latent rate
    context: torch.Tensor             # (nc, ctx_dim) FLAN-T5 hidden states
    roll: torch.Tensor | None = None  # (n_frames, NOTES) piano roll (V2P) or None (V2A -> zeros, x3:2164-2165)


def collate_clips(clips: list[ClipRequest], num_channels: int = 128, generator: torch.Generator | None = None):
    """Returns (batch8, extras): `batch8` is the reference's 8-tuple; `extras` holds the padded
    conditioning tensors to pass as sample(text_embed=, context=, context_mask=, frames_embed=)."""
    assert clips, "empty batch"
    B = len(clips)
    n = max(c.n_frames for c in clips)
    nc = max(c.context.shape[0] for c in clips)
    dt, dc = clips[0].clip_embed.shape[-1], clips[0].context.shape[-1]
    text = [c.prompt for c in clips]
    video_paths = [c.video_path for c in clips]
    mel_len = torch.tensor([c.n_frames for c in clips], dtype=torch.int32)
    mel = torch.randn(B, n, num_channels, generator=generator)     # placeholder cond, as predict.py:261
    video_drop_prompt = [len(c.prompt) == 0 for c in clips]
    clip_embed = torch.zeros(B, n, dt)
    context = torch.zeros(B, nc, dc)
    context_mask = torch.zeros(B, nc, dtype=torch.bool)
    roll = torch.zeros(B, n, NOTES)
    any_roll = False
    for i, c in enumerate(clips):
        assert c.clip_embed.shape[0] == c.n_frames, "clip_embed must be resampled to n_frames (x3:1803-1805)"
        clip_embed[i, :c.n_frames] = c.clip_embed
        context[i, :c.context.shape[0]] = c.context
        context_mask[i, :c.context.shape[0]] = True
        if c.roll is not None:
            roll[i, :c.n_frames] = c.roll[:c.n_frames]
            any_roll = True
    batch8 = [text, mel, video_paths, mel_len, video_drop_prompt, None, None, roll if any_roll else None]
    extras = dict(text_embed=clip_embed, context=context, context_mask=context_mask, frames_embed=roll)
    return batch8, extras
