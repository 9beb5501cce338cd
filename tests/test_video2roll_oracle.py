"""CPU: the Video2Roll restatement (oracle/video2roll_oracle.py) against vectors produced by the REFERENCE module
(oracle/make_golden_video2roll.py ran src/audeo/Video2RollNet.py itself) -- this path's parity is pinned."""
import os

import numpy as np
import pytest
import torch

import v2a_amd  # noqa: F401
from oracle import video2roll_oracle as VO
from v2a_amd.synth import random_video2roll_state_dict, synthetic_piano_frames
from v2a_amd.video2roll import expected_state_dict_shapes

GOLD = os.path.join(os.path.dirname(__file__), "golden")
PARAM_SEED, INPUT_SEED = 4321, 77


@pytest.fixture(scope="module")
def params():
    return random_video2roll_state_dict(PARAM_SEED)


def windows_0_3_6():
    frames = synthetic_piano_frames(1, 7, seed=INPUT_SEED)
    return VO.frame_windows(frames)[[0, 3, 6]]


def test_shapes_agree_between_oracle_and_product():
    a, b = VO.param_shapes(), expected_state_dict_shapes()
    assert a == b
    assert a["conv1.weight"] == (64, 5, 11, 11) and a["FRB4.fc1.weight"] == (128, 192) and a["fc.weight"] == (51, 128)


def test_seeded_weights_are_reproducible(params):
    again = random_video2roll_state_dict(PARAM_SEED)
    assert all(torch.equal(params[k], again[k]) for k in params)
    assert float(params["conv1.weight"].double().abs().sum()) == pytest.approx(float(again["conv1.weight"].double().abs().sum()))


def test_forward_matches_reference_vectors(params):
    g = np.load(os.path.join(GOLD, "video2roll_forward.npz"))
    taps = {}
    with torch.no_grad():
        logits = VO.resnet_forward(params, windows_0_3_6(), taps)
    # same torch ops as the reference module, evaluated functionally: agreement to fp32 rounding
    np.testing.assert_allclose(logits.numpy(), g["logits"], rtol=1e-4, atol=1e-4)
    for k in ("x1", "x2", "x3", "x4", "x5", "x2_", "x3_", "x4_"):
        a = taps[k].numpy()
        assert tuple(g[f"{k}_shape"]) == a.shape
        np.testing.assert_allclose(a[tuple(g[f"{k}_idx"].T)], g[f"{k}_val"], rtol=1e-4, atol=1e-5)
        st = g[f"{k}_stats"]
        assert a.mean(dtype=np.float64) == pytest.approx(st[0], rel=1e-4, abs=1e-6)
        assert np.abs(a).mean(dtype=np.float64) == pytest.approx(st[1], rel=1e-4)


@pytest.mark.parametrize("l", [10, 14])
def test_encode_frames_matches_reference_lines(params, l):
    g = np.load(os.path.join(GOLD, "video2roll_encode.npz"))
    x = synthetic_piano_frames(2, 4, seed=INPUT_SEED + 1)
    with torch.no_grad():
        roll = VO.encode_frames(params, x, l)
    assert roll.shape == (2, l, 51)
    np.testing.assert_allclose(roll.numpy(), g[f"roll_l{l}"], rtol=0, atol=2e-5)
    if l == 14:
        assert np.all(roll.numpy()[:, 12:] == 0)          # zero-padded tail (x3:1549-1550)
        assert np.array_equal(roll.numpy()[:, 0], roll.numpy()[:, 2])   # x3 temporal repeat (x3:1544)


def test_frame_windows_clamp_at_clip_edges():
    x = torch.arange(6, dtype=torch.float32).view(1, 1, 6, 1, 1).expand(1, 1, 6, 2, 3).contiguous()
    w = VO.frame_windows(x)
    assert w.shape == (6, 5, 2, 3)
    assert w[0, :, 0, 0].tolist() == [0, 0, 0, 1, 2]
    assert w[5, :, 0, 0].tolist() == [3, 4, 5, 5, 5]
    assert w[2, :, 0, 0].tolist() == [0, 1, 2, 3, 4]
