"""GPU parity of the Encodec decoder (vocoder) through the C ABI.

Checked against vectors produced by the third-party library the reference calls (tests/golden/encodec_*.npz, made by
oracle/make_golden_encodec.py running transformers' EncodecDecoder), the CPU restatement oracle/encodec_oracle.py, and
plain torch ops per kernel.  Everything is fp32: tolerance 1e-4 abs on O(1) waveforms (summation-order noise through
21 GEMMs and 1 500 recurrent LSTM steps), bit-level for the pad / ELU kernel."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import encodec_oracle as EO

pytestmark = pytest.mark.gpu
DEV = "cuda"
GOLD = os.path.join(os.path.dirname(__file__), "golden")
PARAM_SEED, INPUT_SEED = 2468, 135


def latents(T, seed):
    rs = np.random.RandomState(seed)
    base = rs.standard_normal((128, T // 4 + 2)).astype(np.float32)
    x = np.repeat(base, 4, axis=1)[:, :T] + 0.3 * rs.standard_normal((128, T)).astype(np.float32)
    return torch.from_numpy(x[None])


@pytest.fixture(scope="module")
def L():
    from v2a_amd import _lib
    _lib.lib()
    return _lib


@pytest.fixture(scope="module")
def params():
    from v2a_amd.synth import random_encodec_decoder_state_dict
    return random_encodec_decoder_state_dict(PARAM_SEED)


@pytest.fixture(scope="module")
def dec(params):
    from v2a_amd.encodec import EncodecDecoder
    return EncodecDecoder(params, DEV)


def _g(seed=0):
    return torch.Generator().manual_seed(seed)


@pytest.mark.parametrize("cfg", [(6, True, False), (2, True, True), (1, False, True), (0, True, True)])
def test_elu_pad_matches_torch(L, cfg):
    pad, reflect, act = cfg
    T, C = 37, 48
    x = torch.randn(T, C, generator=_g(pad))
    xa = F.elu(x) if act else x
    ref = F.pad(xa.t()[None], (pad, 0), mode="reflect" if reflect else "constant")[0].t() if pad else xa
    out = torch.full((T + pad, C), 9.0, device=DEV)
    L.elu_pad(x.to(DEV), out, T=T, C_=C, pad=pad, reflect=reflect, act=act)
    torch.testing.assert_close(out.cpu(), ref.contiguous(), atol=1e-6, rtol=1e-6)


@pytest.mark.parametrize("k,ci,co", [(7, 128, 512), (3, 32, 16), (1, 16, 32), (7, 32, 1)])
def test_causal_conv1d_is_one_gemm_over_overlapping_rows(L, k, ci, co):
    T = 100
    x = torch.randn(1, ci, T, generator=_g(k))
    w = torch.randn(co, ci, k, generator=_g(k + 1)) / (ci * k) ** 0.5
    b = torch.randn(co, generator=_g(k + 2))
    ref = F.conv1d(F.pad(x, (k - 1, 0), mode="reflect") if k > 1 else x, w, b)[0].t()
    a = torch.empty(T + k - 1, ci, device=DEV)
    L.elu_pad(x[0].t().contiguous().to(DEV), a, T=T, C_=ci, pad=k - 1, reflect=True, act=False)
    out = torch.empty(T, co, device=DEV)
    L.gemm([(a, ci, k * ci)], w.permute(0, 2, 1).reshape(co, k * ci).contiguous().to(DEV), out, M=T, N=co, compute=L.F32, bias=b.to(DEV), ldo=co)
    torch.testing.assert_close(out.cpu(), ref.contiguous(), atol=2e-5, rtol=2e-5)


@pytest.mark.parametrize("r,ci", [(8, 512), (5, 256), (2, 64)])
def test_conv_transpose1d_is_one_gemm(L, r, ci):
    """ConvTranspose1d(k = 2r, stride r) with the causal right trim: out[q*r + p] = x[q] w[..p] + x[q-1] w[..p+r]."""
    co, T = ci // 2, 50
    x = torch.randn(1, ci, T, generator=_g(r))
    w = torch.randn(ci, co, 2 * r, generator=_g(r + 1)) / (2 * ci) ** 0.5
    b = torch.randn(co, generator=_g(r + 2))
    ref = F.conv_transpose1d(x, w, b, stride=r)[0, :, : T * r].t()
    a = torch.empty(T + 1, ci, device=DEV)
    L.elu_pad(x[0].t().contiguous().to(DEV), a, T=T, C_=ci, pad=1, reflect=False, act=False)
    wp = torch.cat([w[:, :, r:].permute(2, 1, 0), w[:, :, :r].permute(2, 1, 0)], 2).reshape(r * co, 2 * ci).contiguous()
    out = torch.empty(T, r * co, device=DEV)
    L.gemm([(a, ci, 2 * ci)], wp.to(DEV), out, M=T, N=r * co, compute=L.F32, bias=b.repeat(r).to(DEV), ldo=r * co)
    torch.testing.assert_close(out.cpu().view(T * r, co), ref.contiguous(), atol=2e-5, rtol=2e-5)


def test_lstm_layer_matches_torch_lstm(L):
    """One nn.LSTM layer (H = 512) over 200 steps + the skip output; also re-run: the barrier workspace is re-armed."""
    T, H = 200, 512
    lstm = torch.nn.LSTM(H, H, 1)
    with torch.no_grad():
        for p in lstm.parameters():
            p.copy_(torch.rand(p.shape, generator=_g(p.numel())) * 2 - 1).mul_(1 / H ** 0.5)
    x = torch.randn(T, 1, H, generator=_g(1))
    with torch.no_grad():
        ref = lstm(x)[0][:, 0]
        gx = x[:, 0] @ lstm.weight_ih_l0.t() + lstm.bias_ih_l0 + lstm.bias_hh_l0
    ws = torch.zeros(4 * H + 2, dtype=torch.int32, device=DEV)
    for _ in range(2):
        h = torch.zeros(T, H, device=DEV)
        y = torch.zeros(T, H, device=DEV)
        L.lstm_layer(gx.to(DEV), lstm.weight_hh_l0.detach().to(DEV).contiguous(), h, ws, T=T, H=H, resid=x[:, 0].to(DEV).contiguous(), y=y)
        torch.cuda.synchronize()
        assert ws.tolist()[4 * H] == 0
        torch.testing.assert_close(h.cpu(), ref, atol=2e-5, rtol=2e-5)
        torch.testing.assert_close(y.cpu(), ref + x[:, 0], atol=2e-5, rtol=2e-5)
    with pytest.raises(L.V2AError, match="hidden size"):
        L.lstm_layer(gx.to(DEV), lstm.weight_hh_l0.detach().to(DEV), h, ws, T=T, H=256)


def test_small_decode_matches_library_vectors(dec):
    g = np.load(os.path.join(GOLD, "encodec_small.npz"))
    taps = {}
    wav = dec.decoder(latents(24, INPUT_SEED + 24), taps)
    assert wav.shape == (1, 1, 7680) and wav.is_cuda
    err = np.abs(wav[0, 0].cpu().numpy() - g["wav"]).max()
    print(f"\nencodec decoder (T=24) vs library vectors: max |d| = {err:.3e}")
    assert err < 1e-4
    for k in ("lstm", "stage8", "stage5", "stage4", "stage2"):
        a = taps[k].cpu().numpy()
        assert tuple(g[f"{k}_shape"]) == a.shape
        np.testing.assert_allclose(a[tuple(g[f"{k}_idx"].T)], g[f"{k}_val"], rtol=0, atol=1e-4)


def test_full_clip_matches_library_vectors(dec):
    """BASELINE clip: 750 latent frames -> 240 000 samples (10 s at 24 kHz)."""
    g = np.load(os.path.join(GOLD, "encodec_full.npz"))
    wav = dec.decode(latents(750, INPUT_SEED + 750))
    assert wav.shape == (1, 240000)
    w = wav[0].cpu().numpy()
    err = np.abs(w[g["wav_idx"]] - g["wav_val"]).max()
    print(f"\nencodec decoder (T=750) vs library vectors: max |d| = {err:.3e}")
    assert err < 1e-4
    assert np.abs(w).mean(dtype=np.float64) == pytest.approx(g["stats"][1], rel=1e-4)


def test_batch_and_legacy_weight_names(params, dec):
    """Two clips decode independently; the hub checkpoint's `decoder.`-prefixed weight_g / weight_v names load too."""
    from v2a_amd.encodec import EncodecDecoder
    emb = torch.cat([latents(30, 1), latents(30, 2)], 0)
    both = dec.decoder(emb)
    solo = dec.decoder(emb[1:])
    assert torch.equal(both[1], solo[0])
    with torch.no_grad():
        ref = EO.decoder_forward(params, emb)
    assert float((both.cpu() - ref).abs().max()) < 1e-4
    legacy = {}
    for k, v in params.items():
        k = "decoder." + k.replace("parametrizations.weight.original0", "weight_g").replace("parametrizations.weight.original1", "weight_v")
        legacy[k] = v
    legacy["encoder.layers.0.conv.bias"] = torch.zeros(3)            # ignored
    again = EncodecDecoder(legacy, DEV).decoder(emb)
    assert torch.equal(again, both)
    with pytest.raises(ValueError, match="7 latent frames"):
        dec.decoder(torch.zeros(1, 128, 5))


def test_shortest_clip_and_sample_level_causality(params, dec):
    """7 latent frames (the minimum the k=7 reflect padding allows); and causality end to end: changing latent frame 20
    leaves the first 19 * 320 samples bit-identical (every conv pads on the left only, the LSTM runs forward)."""
    emb = latents(7, 9)
    with torch.no_grad():
        ref = EO.decoder_forward(params, emb)
    got = dec.decoder(emb)
    assert got.shape == (1, 1, 7 * 320) and float((got.cpu() - ref).abs().max()) < 1e-4
    x = latents(24, 3)
    y = x.clone()
    y[:, :, 20:] += 1.0
    a, b = dec.decoder(x)[0, 0], dec.decoder(y)[0, 0]
    assert torch.equal(a[: 19 * 320], b[: 19 * 320])
    assert float((a[20 * 320:] - b[20 * 320:]).abs().max()) > 1e-3


def test_vocoder_next_to_other_work_on_the_gpu(dec, L):
    """The LSTM kernel is persistent (64 workgroups exchanging h_t through memory): decoding while another stream keeps the
    GPU busy with GEMMs must neither hang nor change a bit (workgroups that start late are simply waited for)."""
    emb = latents(200, 4)
    ref = dec.decoder(emb).clone()
    M, N, K = 4096, 4096, 1024
    a = torch.randn(M, K, device=DEV).to(torch.bfloat16)
    w = torch.randn(N, K, device=DEV).to(torch.bfloat16)
    out = torch.empty(M, N, device=DEV)
    side = torch.cuda.Stream()
    torch.cuda.synchronize()
    with torch.cuda.stream(side):
        for _ in range(300):                                    # ~20 ms of back-to-back chip-filling GEMMs
            L.gemm([(a, K, K)], w, out, M=M, N=N, compute=L.BF16)
    got = [dec.decoder(emb).clone() for _ in range(3)]          # on the current stream, concurrently
    torch.cuda.synchronize()
    for g in got:
        assert torch.equal(g, ref)


def test_lstm2_matches_torch_two_layer_lstm(L):
    """Both layers in one persistent kernel (layer 1 one step behind layer 0) == nn.LSTM(512, 512, 2) + skip, 300 steps."""
    T, H = 300, 512
    lstm = torch.nn.LSTM(H, H, 2)
    with torch.no_grad():
        for p in lstm.parameters():
            p.copy_(torch.rand(p.shape, generator=_g(p.numel() + 1)) * 2 - 1).mul_(1 / H ** 0.5)
    x = torch.randn(T, 1, H, generator=_g(2))
    with torch.no_grad():
        ref = lstm(x)[0][:, 0] + x[:, 0]
        gx = x[:, 0] @ lstm.weight_ih_l0.t() + lstm.bias_ih_l0 + lstm.bias_hh_l0
    d = lambda t: t.detach().to(DEV).contiguous()
    ws = torch.zeros(8 * H + 2, dtype=torch.int32, device=DEV)
    for _ in range(2):                                           # the second call re-arms the exchange tables
        y = torch.zeros(T, H, device=DEV)
        L.lstm2(d(gx), d(lstm.weight_hh_l0), d(lstm.weight_ih_l1), d(lstm.bias_ih_l1 + lstm.bias_hh_l1), d(lstm.weight_hh_l1), y, ws,
                T=T, H=H, resid=d(x[:, 0]))
        torch.cuda.synchronize()
        assert ws.tolist()[8 * H] == 0
        torch.testing.assert_close(y.cpu(), ref, atol=3e-5, rtol=3e-5)
