"""MI355X-native flow-matching V2A sampler (drop-in for the reference's E2TTS sampling path).

Import name: `v2a_amd` (the directory name contains hyphens; the repo-root shim `v2a_amd.py`
registers this package under that name).
"""
from .dit import DiTConfig, DiTEngine, PackedWeights, NOTES  # noqa: F401
from .e2tts import E2TTS, sway_grid, lens_to_mask, expected_state_dict_shapes  # noqa: F401
from .collate import collate_clips, ClipRequest  # noqa: F401
from .dist import shard_range, gather_latents  # noqa: F401
from .features import (feature_cache_path, save_clip_cache, load_clip_cache, resample_indices,  # noqa: F401
                       resample_clip_features, encode_video_cached, piano_frames_cache_path, save_piano_frames_cache,
                       piano_frame_indices, load_piano_frames)
from .video2roll import Video2RollEngine  # noqa: F401
from .encodec import EncodecDecoder  # noqa: F401
from . import _lib  # noqa: F401

__all__ = ["E2TTS", "DiTConfig", "DiTEngine", "PackedWeights", "collate_clips", "ClipRequest",
           "shard_range", "gather_latents", "sway_grid", "lens_to_mask", "expected_state_dict_shapes", "NOTES",
           "Video2RollEngine", "EncodecDecoder"]
