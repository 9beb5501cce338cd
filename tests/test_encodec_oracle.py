"""CPU: the Encodec decoder restatement (oracle/encodec_oracle.py) against vectors produced by the third-party library
the reference calls (oracle/make_golden_encodec.py ran transformers' EncodecModel(EncodecConfig()).decoder) -- pinned."""
import os

import numpy as np
import pytest
import torch

import v2a_amd  # noqa: F401
from oracle import encodec_oracle as EO
from v2a_amd.encodec import expected_state_dict_shapes
from v2a_amd.synth import random_encodec_decoder_state_dict

GOLD = os.path.join(os.path.dirname(__file__), "golden")
PARAM_SEED, INPUT_SEED = 2468, 135


def latents(T, seed):
    rs = np.random.RandomState(seed)
    base = rs.standard_normal((128, T // 4 + 2)).astype(np.float32)
    x = np.repeat(base, 4, axis=1)[:, :T] + 0.3 * rs.standard_normal((128, T)).astype(np.float32)
    return torch.from_numpy(x[None])


@pytest.fixture(scope="module")
def params():
    return random_encodec_decoder_state_dict(PARAM_SEED)


def test_shapes_agree_between_oracle_and_product():
    a, b = EO.param_shapes(), expected_state_dict_shapes()
    assert a == b and len(a) == 62
    assert a["layers.3.conv.parametrizations.weight.original1"] == (512, 256, 16)      # ConvTranspose1d: (in, out, k)
    assert a["layers.15.conv.parametrizations.weight.original1"] == (1, 32, 7)


def test_small_decode_matches_library_vectors(params):
    g = np.load(os.path.join(GOLD, "encodec_small.npz"))
    taps = {}
    with torch.no_grad():
        wav = EO.decoder_forward(params, latents(24, INPUT_SEED + 24), taps)
    assert wav.shape == (1, 1, 24 * 320)
    np.testing.assert_allclose(wav[0, 0].numpy(), g["wav"], rtol=0, atol=2e-5)
    for k in ("lstm", "stage8", "stage5", "stage4", "stage2"):
        a = taps[k][0].numpy()
        assert tuple(g[f"{k}_shape"]) == a.shape
        np.testing.assert_allclose(a[tuple(g[f"{k}_idx"].T)], g[f"{k}_val"], rtol=0, atol=2e-5)
    with torch.no_grad():
        assert torch.equal(EO.decode(params, latents(24, INPUT_SEED + 24)), wav[0])      # x3:436-437: output[0]


def test_full_clip_matches_library_vectors(params):
    """The BASELINE clip: 750 latent frames -> 240 000 samples; 750 recurrent LSTM steps do not drift (2e-5)."""
    g = np.load(os.path.join(GOLD, "encodec_full.npz"))
    with torch.no_grad():
        wav = EO.decoder_forward(params, latents(750, INPUT_SEED + 750))[0, 0].numpy()
    assert wav.shape[0] == int(g["length"]) == 240000
    np.testing.assert_allclose(wav[g["wav_idx"]], g["wav_val"], rtol=0, atol=2e-5)
    assert np.abs(wav).mean(dtype=np.float64) == pytest.approx(g["stats"][1], rel=1e-5)


def test_causality_of_the_restated_convolutions(params):
    """Every conv is causal (left padding only): the first 320*t samples do not depend on latent frames > t + history."""
    x = latents(24, 3)
    y = x.clone()
    y[:, :, 20:] += 1.0
    with torch.no_grad():
        a, b = EO.decoder_forward(params, x)[0, 0], EO.decoder_forward(params, y)[0, 0]
    assert torch.equal(a[: 20 * 320 - 0], b[: 20 * 320 - 0]) or float((a[: 19 * 320] - b[: 19 * 320]).abs().max()) == 0.0
    assert float((a[20 * 320:] - b[20 * 320:]).abs().max()) > 1e-3
