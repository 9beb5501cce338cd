"""CPU: CLIP feature cache format + resampler and the batched CLI's host logic (N4)."""
import os

import numpy as np
import pytest
import torch

import v2a_amd
from v2a_amd import cli


def _reference_loop(image_embeddings, duration, l, sampling_rate=24000, frame_size=320, start_sample=0, max_sample=None):
    """The resampling loop as written at e2_tts_crossatt3.py:1801-1813 (restated for the test)."""
    if max_sample is None:
        max_sample = int(duration * sampling_rate)
    interpolated = []
    for i in range(start_sample, max_sample, frame_size):
        j = min(round((i + frame_size // 2) / sampling_rate / (duration / (image_embeddings.shape[0] - 1))), image_embeddings.shape[0] - 1)
        interpolated.append(image_embeddings[j:j + 1])
        if len(interpolated) >= l:
            break
    return torch.cat(interpolated, dim=0)


@pytest.mark.parametrize("nf,duration,l", [(240, 10.0, 750), (251, 10.04, 750), (30, 1.3, 750), (300, 10.0, 400), (2, 0.5, 750)])
def test_resampler_matches_reference_loop(nf, duration, l):
    emb = torch.randn(nf, 16, generator=torch.Generator().manual_seed(nf))
    ref = _reference_loop(emb, duration, l)
    got = v2a_amd.resample_clip_features(emb, duration, l)
    assert got.shape == (l, 16)
    assert torch.equal(got[: ref.shape[0]], ref) and float(got[ref.shape[0]:].abs().max() if ref.shape[0] < l else 0) == 0
    idx = v2a_amd.resample_indices(nf, duration, l)
    assert idx == sorted(idx) and idx[0] == 0 and max(idx) <= nf - 1
    if nf == 240 and l == 750:         # 24 fps -> 75 Hz: runs of ~3 latent frames per video frame
        assert len(idx) == 750 and 230 < len(set(idx)) <= 240


def test_cache_roundtrip_and_paths(tmp_path):
    emb = torch.randn(24, 1280)
    vp = str(tmp_path / "clip.mp4")
    fp = v2a_amd.feature_cache_path(vp)
    assert fp.endswith("clip.generated.npz")
    assert v2a_amd.feature_cache_path("/ailab-train2/speech/zhanghaomin/VGGSound/video/a.mp4") == \
        "/ailab-train2/speech/zhanghaomin/VGGSound/feature/a.npz"
    v2a_amd.save_clip_cache(fp, emb, 1.0)
    data = np.load(fp)
    assert sorted(data.files) == ["arr_0", "arr_1"] and data["arr_1"].shape == ()        # np.savez positional keys
    e2, dur = v2a_amd.load_clip_cache(fp)
    assert torch.equal(e2, emb) and dur == 1.0
    out = v2a_amd.encode_video_cached([vp, None, (vp, 0, 12000)], 75)
    assert out.shape == (3, 75, 1280) and float(out[1].abs().max()) == 0
    assert float(out[2, 38:].abs().max()) == 0 and float(out[2, :37].abs().min()) >= 0     # max_sample = 0.5 s -> 38 frames
    with pytest.raises(FileNotFoundError):
        v2a_amd.encode_video_cached([str(tmp_path / "missing.mp4")], 75)
    made = v2a_amd.encode_video_cached([str(tmp_path / "new.mp4")], 75, encoder_fn=lambda p: (emb, 1.0))
    assert os.path.exists(str(tmp_path / "new.generated.npz")) and torch.equal(made[0], out[0])


def test_cli_scp_and_requests(tmp_path):
    scp = tmp_path / "test.scp"
    vids = [str(tmp_path / f"v{i}.mp4") for i in range(5)]
    scp.write_text("".join(f"{v}\tcaption {i}\n" for i, v in enumerate(vids)))
    for i, v in enumerate(vids):
        v2a_amd.save_clip_cache(v2a_amd.feature_cache_path(v), torch.randn(24 * (2 + i), 1280), 2.0 + i)
        np.savez(v.replace(".mp4", ".t5.npz"), np.random.randn(4 + i, 1024).astype(np.float32))
    items = cli.read_scp(str(scp), 1, 4)
    assert [c for _, c in items] == ["caption 1", "caption 2", "caption 3"]          # the reference's start/end slicing
    reqs = cli.build_requests(items, drop_prompt=False, n_frames=750)
    assert [r.n_frames for r in reqs] == [225, 300, 375] and reqs[0].clip_embed.shape == (225, 1280)
    batch8, extras = v2a_amd.collate_clips(reqs)
    assert extras["text_embed"].shape == (3, 375, 1280) and extras["context_mask"].sum(-1).tolist() == [5, 6, 7]
    dropped = cli.build_requests(items[:1], drop_prompt=True, n_frames=100)
    assert dropped[0].prompt == "" and dropped[0].n_frames == 100
    assert v2a_amd.collate_clips(dropped)[0][4] == [True]                              # video_drop_prompt
