"""Cached CLIP features: on-disk format and frame-rate resampler.

Mirrors the cache half of `E2TTS.encode_video` (x3:1659-1827, `x3` =
/root/reference/src/e2_tts_pytorch/e2_tts_crossatt3.py): the CLIP image encoder runs once per
video and its output is kept next to the video as an `.npz` written by
`np.savez(feature_path, image_embeddings, duration)` (x3:1793), i.e.
    arr_0 : (n_video_frames, dim) float embeddings, one per decoded video frame
    arr_1 : 0-d float, clip duration in seconds
and every later call only resamples those rows to the 75 Hz latent rate by nearest video frame
(x3:1803-1813).  The encoder itself (ViT-bigG) is outside the accelerated path.
"""
from __future__ import annotations

import os

import numpy as np
import torch

_SUFFIX = {"clip_vit": ".generated.npz", "clip_vit2": ".generated.clip_vit2.npz",
           "clip_convnext": ".generated.clip_convnext.npz", "dinov2": ".generated.dinov2.npz", "mixed": ".generated.mixed.npz"}
_VGG_DIR = {"clip_vit": "/feature/", "clip_vit2": "/feature_clip_vit2/", "clip_convnext": "/feature_clip_convnext/",
            "dinov2": "/feature_dinov2/", "mixed": "/feature_mixed/"}
_VGG_ROOT = "/ailab-train2/speech/zhanghaomin/VGGSound/"


def feature_cache_path(video_path: str, video_encoder: str = "clip_vit") -> str:
    """x3:1678-1701: where encode_video looks for / writes the cache of one video."""
    if video_encoder not in _SUFFIX:
        raise ValueError("Invalid video_encoder " + video_encoder)
    if video_path.startswith(_VGG_ROOT):
        return video_path.replace("/video/", _VGG_DIR[video_encoder]).replace(".mp4", ".npz")
    return video_path.replace(".mp4", _SUFFIX[video_encoder])


def save_clip_cache(path: str, image_embeddings, duration: float) -> None:
    """x3:1793 `np.savez(feature_path, image_embeddings, duration)` -> keys arr_0, arr_1."""
    emb = image_embeddings.detach().cpu().numpy() if torch.is_tensor(image_embeddings) else np.asarray(image_embeddings)
    np.savez(path, emb, duration)


def load_clip_cache(path: str) -> tuple[torch.Tensor, float]:
    """x3:1796-1800."""
    data = np.load(path)
    return torch.from_numpy(data["arr_0"]), data["arr_1"].item()


def resample_indices(n_video_frames: int, duration: float, l: int, sampling_rate: int = 24000, frame_size: int = 320,
                     start_sample: int = 0, max_sample: int | None = None) -> list[int]:
    """x3:1801-1808: for every latent frame (hop `frame_size` samples) the index of the nearest video frame,
    at most `l` of them.  Python's round() (half to even) is part of the reference behaviour."""
    if max_sample is None:
        max_sample = int(duration * sampling_rate)
    out = []
    for i in range(start_sample, max_sample, frame_size):
        j = min(round((i + frame_size // 2) / sampling_rate / (duration / (n_video_frames - 1))), n_video_frames - 1)
        out.append(j)
        if len(out) >= l:
            break
    return out


def resample_clip_features(image_embeddings: torch.Tensor, duration: float, l: int, **kw) -> torch.Tensor:
    """(n_video_frames, dim) -> (l, dim): nearest-frame resampling, zero padded to l (x3:1815-1824)."""
    idx = resample_indices(image_embeddings.shape[0], duration, l, **kw)
    out = torch.zeros(l, image_embeddings.shape[1], dtype=image_embeddings.dtype)
    if idx:
        out[: len(idx)] = image_embeddings[torch.tensor(idx)]
    return out


def encode_video_cached(video_paths, l: int, dim: int = 1280, video_encoder: str = "clip_vit", sampling_rate: int = 24000,
                        frame_size: int = 320, encoder_fn=None) -> torch.Tensor:
    """Batch form of encode_video for cached features: (b, l, dim) float32 on the CPU.
    `None` paths give zero rows (x3:1669-1672); tuples are (path, start_sample, max_sample) (x3:1673-1674).
    A missing cache is produced by `encoder_fn(video_path) -> (embeddings, duration)` and saved, as the
    reference does with its CLIP model (x3:1706-1793); without encoder_fn it is an error."""
    rows = []
    for vp in video_paths:
        if vp is None:
            rows.append(torch.zeros(l, dim))
            continue
        start_sample, max_sample = 0, None
        if isinstance(vp, tuple):
            vp, start_sample, max_sample = vp
        fp = feature_cache_path(vp, video_encoder)
        if not os.path.exists(fp):
            if encoder_fn is None:
                raise FileNotFoundError(f"{fp}: no cached CLIP features for {vp} and no encoder_fn to create them")
            emb, duration = encoder_fn(vp)
            save_clip_cache(fp, emb, duration)
        emb, duration = load_clip_cache(fp)
        rows.append(resample_clip_features(emb.float(), duration, l, sampling_rate=sampling_rate, frame_size=frame_size,
                                           start_sample=start_sample, max_sample=max_sample))
    return torch.stack(rows, 0)


# ---- piano-frame cache of the V2P path (the `piano` branch of E2TTS.encode_video_frames, x3:1829-1991) ------------------
def piano_frames_cache_path(video_path: str) -> str:
    """x3:1876: the grey 100x900 frames of a video are cached next to it."""
    return video_path.replace(".mp4", ".generated_frames_raw.2.npz")


def save_piano_frames_cache(path: str, frames_raw, duration: float) -> None:
    """x3:1890-1891 `np.savez(frames_raw_path, frames_raw, duration)`: arr_0 = (n_video_frames, 100, 900, 1) float32 in
    [0, 1] (ToTensor of the 'L' image), arr_1 = duration in seconds."""
    fr = frames_raw.detach().cpu().numpy() if torch.is_tensor(frames_raw) else np.asarray(frames_raw)
    np.savez(path, fr.astype(np.float32), duration)


def piano_frame_indices(n_video_frames: int, duration: float, l: int, start_sample: int = 0, max_sample: int | None = None,
                        video_multi: float = 3.0, sampling_rate: int = 24000, frame_size: int = 320) -> list[int]:
    """x3:1903-1913: one video frame per `video_multi` latent frames (hop 3 * 320 samples), nearest by time, at most
    floor(l / video_multi) + 1 of them.  The loop runs one hop past max_sample, as the reference's does."""
    if max_sample is None:
        max_sample = int(duration * sampling_rate)
    hop = int(video_multi * frame_size)
    out = []
    for i in range(start_sample, max_sample + hop, hop):
        out.append(min(round(i / sampling_rate / (duration / n_video_frames)), n_video_frames - 1))
        if len(out) >= int(l // video_multi) + 1:
            break
    return out


def load_piano_frames(video_paths, l: int) -> torch.Tensor | None:
    """Batch form of the cached `piano` branch of encode_video_frames: (b, 1, t, 100, 900) float32 with
    t = max(floor(l / 3) + 1, longest clip) and zero frames as padding (x3:1935-1948); `None` paths give all-zero clips
    (x3:1939-1941).  Returns None when no path has frames (x3:1927-1928).  Tuples are (path, start_sample, max_sample)."""
    clips, lens = [], []
    for vp in video_paths:
        if vp is None:
            clips.append(None)
            lens.append(0)
            continue
        start_sample, max_sample = 0, None
        if isinstance(vp, tuple):
            vp, start_sample, max_sample = vp
        fp = piano_frames_cache_path(vp)
        if not os.path.exists(fp):
            raise FileNotFoundError(f"{fp}: no cached piano frames for {vp} (the reference decodes the video with moviepy here)")
        data = np.load(fp)
        raw = torch.from_numpy(data["arr_0"])
        idx = piano_frame_indices(raw.shape[0], data["arr_1"].item(), l, start_sample, max_sample)
        clips.append(raw[torch.tensor(idx)])
        lens.append(len(idx))
    if not any(c is not None for c in clips):
        return None
    H, W = next(c for c in clips if c is not None).shape[1:3]
    t = max(int(l // 3.0) + 1, max(lens))
    out = torch.zeros(len(clips), t, H, W, 1)
    for i, c in enumerate(clips):
        if c is not None:
            out[i, : c.shape[0]] = c
    return out.permute(0, 4, 1, 2, 3).contiguous()
