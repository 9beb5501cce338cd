"""CPU: CLIP feature cache format + resampler and the batched CLI's host logic (N4)."""
import os

import numpy as np
import pytest
import torch

import v2a_amd
from v2a_amd import cli


@pytest.mark.parametrize("nf,duration,l", [(240, 10.0, 750), (251, 10.04, 750), (30, 1.3, 750), (300, 10.0, 400), (2, 0.5, 750)])
def test_resampler_properties(nf, duration, l):
    """Shape / monotonicity properties; the index values themselves are checked against the reference's own loop in
    tests/test_intree_golden.py::test_resample_indices_match_reference_loop (fixture produced by x3:1800-1808)."""
    emb = torch.randn(nf, 16, generator=torch.Generator().manual_seed(nf))
    got = v2a_amd.resample_clip_features(emb, duration, l)
    assert got.shape == (l, 16)
    idx = v2a_amd.resample_indices(nf, duration, l)
    assert idx == sorted(idx) and idx[0] == 0 and max(idx) <= nf - 1
    assert torch.equal(got[: len(idx)], emb[torch.tensor(idx)])
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


# ---- piano-frame cache (V2P, x3:1876-1948) ------------------------------------------------------------------------------
def test_piano_frame_indices_follow_the_reference_loop():
    import math
    import v2a_amd
    n, dur, l = 240, 10.0, 750                         # 24 fps video, 10 s, 750 latent frames
    idx = v2a_amd.piano_frame_indices(n, dur, l)
    # the reference loop, literally (x3:1905-1911)
    ref = []
    fsv = int(3.0 * 320)
    for i in range(0, int(dur * 24000) + fsv, fsv):
        ref.append(min(round(i / 24000 / (dur / (n - 0))), n - 1))
        if len(ref) >= math.floor(l / 3.0) + 1:
            break
    assert idx == ref and len(idx) == 251 and idx[0] == 0 and idx[-1] == n - 1
    assert len(v2a_amd.piano_frame_indices(n, dur, 40)) == 14          # floor(40 / 3) + 1


def test_load_piano_frames_batches_and_pads(tmp_path):
    import v2a_amd
    vids = [str(tmp_path / f"p{i}.mp4") for i in range(2)]
    g = torch.Generator().manual_seed(1)
    raws = [torch.rand(48, 100, 900, 1, generator=g), torch.rand(30, 100, 900, 1, generator=g)]
    for v, r, d in zip(vids, raws, (2.0, 1.2)):
        v2a_amd.save_piano_frames_cache(v2a_amd.piano_frames_cache_path(v), r, d)
    assert v2a_amd.piano_frames_cache_path(vids[0]).endswith("p0.generated_frames_raw.2.npz")
    fr = v2a_amd.load_piano_frames([vids[0], None, vids[1]], 150)
    assert fr.shape == (3, 1, 51, 100, 900) and fr.dtype == torch.float32
    assert torch.all(fr[1] == 0)                                        # missing video: zero frames
    i0 = v2a_amd.piano_frame_indices(48, 2.0, 150)
    assert torch.equal(fr[0, 0, : len(i0)], raws[0][i0, :, :, 0])
    i1 = v2a_amd.piano_frame_indices(30, 1.2, 150)
    assert len(i1) < 51 and torch.all(fr[2, 0, len(i1):] == 0)           # shorter clip: zero padding at the end
    assert v2a_amd.load_piano_frames([None, None], 150) is None
    with pytest.raises(FileNotFoundError):
        v2a_amd.load_piano_frames([str(tmp_path / "nope.mp4")], 150)
