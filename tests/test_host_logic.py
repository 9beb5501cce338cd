"""CPU: host-side logic of the drop-in (no kernel launches): weight packing, state_dict mapping,
grid, collator, sharding arithmetic."""
import numpy as np
import pytest
import torch

import v2a_amd
from v2a_amd.dit import _FF, _Attn, PackedWeights, DiTConfig
from conftest import make_model


def test_state_dict_layout_matches_reference_key_scheme():
    shapes = v2a_amd.expected_state_dict_shapes(DiTConfig())
    # nested ModuleList indices of x3:824-933 for the shipped config
    assert shapes["transformer.layers.6.0.0.weight"] == (1024, 2048)            # skip_proj only in the later half
    assert "transformer.layers.5.0.0.weight" not in shapes
    assert shapes["transformer.layers.0.0.3.to_v_head_gate.weight"] == (16, 1024)
    assert shapes["transformer.layers.0.0.9.ff.0.proj.weight"] == (8192, 1024)
    assert shapes["transformer.layers.0.1.2.to_q.weight"] == (1024, 1280)       # text attention: 16 heads x 64 from dim 1280
    assert shapes["transformer.layers.0.2.2.to_q.weight"] == (512, 512)         # frames attention: 8 heads
    assert shapes["transformer.layers.3.1.5.text_frames_to_audio.weight"] == (1024, 2816)
    assert "transformer.layers.11.1.5.audio_to_text.weight" not in shapes       # cond_audio_to_text = not is_last
    assert shapes["transformer.time_cond_mlp.1.weight"] == (1024, 1025)
    assert shapes["proj_frames.weight"] == (512, 51)
    assert sum(int(np.prod(s)) for s in shapes.values()) == 776583456


def test_load_state_dict_reports_missing_and_unexpected(small):
    cfg, P = small["cfg"], small["P"]
    sd = dict(P)
    sd["vocos.decoder.weight"] = torch.zeros(3)
    sd["text_encoder2.shared.weight"] = torch.zeros(3)
    del sd["to_pred.bias"]
    m = v2a_amd.E2TTS(transformer=dict(dim=cfg.dim, dim_text=cfg.dim_text, dim_frames=cfg.dim_frames, depth=cfg.depth, heads=cfg.heads,
                                       frames_heads=cfg.frames_heads, num_registers=cfg.num_registers, max_seq_len=cfg.max_seq_len, if_text_conv=True),
                      num_channels=cfg.num_channels, device="cpu", if_cond_proj_in=False)
    res = m.load_state_dict(sd, strict=False)
    assert res.missing_keys == ["to_pred.bias"]
    # the constructor default if_cond_proj_in=True (x3:1312) registers the audio-prompt projection as well (x3:1365)
    m2 = v2a_amd.E2TTS(transformer=dict(dim=cfg.dim, dim_text=cfg.dim_text, dim_frames=cfg.dim_frames, depth=cfg.depth, heads=cfg.heads,
                                        frames_heads=cfg.frames_heads, num_registers=cfg.num_registers, max_seq_len=cfg.max_seq_len, if_text_conv=True),
                       num_channels=cfg.num_channels, device="cpu")
    assert m2.load_state_dict(dict(P), strict=False).missing_keys == ["cond_proj_in.weight", "cond_proj_in.bias"]
    assert sorted(res.unexpected_keys) == ["text_encoder2.shared.weight", "vocos.decoder.weight"]
    with pytest.raises(RuntimeError):
        m.load_state_dict(sd, strict=True)
    bad = dict(P)
    bad["to_pred.bias"] = torch.zeros(3)
    with pytest.raises(RuntimeError, match="size mismatch"):
        m.load_state_dict(bad, strict=False)


def test_unsupported_configs_are_rejected():
    with pytest.raises(NotImplementedError):
        v2a_amd.E2TTS(transformer=dict(dim=128, if_cross_attn=False, if_text_conv=True), num_channels=16, device="cpu")
    with pytest.raises(NotImplementedError):
        v2a_amd.E2TTS(transformer=dict(dim=128, if_text_conv=True), num_channels=16, odeint_kwargs=dict(method="midpoint"), device="cpu")
    with pytest.raises(TypeError):
        v2a_amd.E2TTS(transformer=None, num_channels=16, device="cpu")


def test_geglu_packing_roundtrip():
    d = 64
    g = torch.Generator().manual_seed(0)
    sd = {"p.ff.0.proj.weight": torch.randn(8 * d, d, generator=g), "p.ff.0.proj.bias": torch.randn(8 * d, generator=g),
          "p.ff.2.weight": torch.randn(d, 4 * d, generator=g), "p.ff.2.bias": torch.randn(d, generator=g)}
    F = _FF(sd, "p", d, torch.float32, "cpu")
    inner = 4 * d
    w = F.w1.reshape(inner // 16, 2, 16, d)
    assert torch.equal(w[:, 0].reshape(inner, d), sd["p.ff.0.proj.weight"][:inner])       # value rows
    assert torch.equal(w[:, 1].reshape(inner, d), sd["p.ff.0.proj.weight"][inner:])       # gate rows
    b = F.b1.reshape(inner // 16, 2, 16)
    assert torch.equal(b[:, 1].reshape(inner), sd["p.ff.0.proj.bias"][inner:])


def test_attention_packing(small):
    P, cfg = small["P"], small["cfg"]
    A = _Attn(P, "transformer.layers.0.0.3", cfg.dim, cfg.heads, 64, torch.float32, "cpu")
    inner = cfg.heads * 64
    assert A.n_pad % 16 == 0 and A.gate_col == 3 * inner
    assert torch.equal(A.w_in[inner:2 * inner], P["transformer.layers.0.0.3.to_k.weight"])
    assert torch.equal(A.w_in[A.gate_col:A.gate_col + cfg.heads], P["transformer.layers.0.0.3.to_v_head_gate.weight"])
    assert torch.equal(A.b_in[A.gate_col:A.gate_col + cfg.heads], P["transformer.layers.0.0.3.to_v_head_gate.bias"])
    assert float(A.b_in[:A.gate_col].abs().max()) == 0 and float(A.w_in[A.gate_col + cfg.heads:].abs().max()) == 0
    C = _Attn(P, "transformer.layers.0.0.6", cfg.dim, cfg.heads, 64, torch.float32, "cpu", cross=True)
    assert C.gate_col == inner and C.n_pad == inner + 16


def test_packed_weights_tables(small):
    P, cfg = small["P"], small["cfg"]
    c2 = DiTConfig(**cfg.to_dict())
    W = PackedWeights(c2, P, "cpu", torch.bfloat16)
    d, L = cfg.dim, cfg.depth
    assert W.norm_gamma_w.shape == (L * 3 * d, d) and W.norm_gamma_w.dtype == torch.float32     # tables stay fp32
    assert torch.equal(W.norm_gamma_w[(1 * 3 + 2) * d:(1 * 3 + 3) * d], P["transformer.layers.1.0.8.to_gamma.weight"])
    assert torch.equal(W.gate_b[(2 * 3 + 1) * d:(2 * 3 + 2) * d], P["transformer.layers.2.0.7.to_gamma.bias"])
    inner = cfg.heads * 64
    assert W.ctx_kv_w.shape == (2 * L * inner, cfg.dim) and W.ctx_kv_w.dtype == torch.bfloat16
    assert torch.equal(W.ctx_kv_w[(L + 3) * inner:(L + 4) * inner].float(), P["transformer.layers.3.0.6.to_v.weight"].bfloat16().float())
    assert W.layers[0]["a_conv"].wt.shape == (31, d)
    assert "skip" not in W.layers[0] and "skip" in W.layers[L // 2] and "x_at" not in W.layers[-1]


def test_sway_grid_matches_reference_formula():
    t = v2a_amd.sway_grid(32)
    lin = torch.linspace(0, 1, 32)
    ref = lin + -1.0 * (torch.cos(torch.pi / 2 * lin) - 1 + lin)                 # x3:2250-2252 verbatim arithmetic
    assert torch.equal(t, ref) and t.shape == (32,)
    assert torch.equal(v2a_amd.sway_grid(5, False), torch.linspace(0, 1, 5))


def test_collate_and_shards():
    g = torch.Generator().manual_seed(0)
    clips = [v2a_amd.ClipRequest(f"v{i}.mp4", "" if i == 1 else "a dog barks", n, torch.randn(n, 1280, generator=g),
                                 torch.randn(nc, 1024, generator=g), None if i else torch.rand(n, 51, generator=g))
             for i, (n, nc) in enumerate([(750, 7), (600, 4), (750, 16)])]
    batch8, ex = v2a_amd.collate_clips(clips, generator=g)
    text, mel, paths, mel_len, vdrop, adrop, frames, midis = batch8          # the reference's 8-tuple order (predict.py:237)
    assert mel.shape == (3, 750, 128) and mel_len.tolist() == [750, 600, 750] and vdrop == [False, True, False]
    assert ex["text_embed"].shape == (3, 750, 1280) and float(ex["text_embed"][1, 600:].abs().max()) == 0
    assert ex["context_mask"].sum(-1).tolist() == [7, 4, 16] and ex["frames_embed"].shape == (3, 750, 51)
    # clip sharding arithmetic: contiguous B/world clips, padded then dropped
    assert v2a_amd.shard_range(64, 3, 8) == (24, 32, 8)
    assert v2a_amd.shard_range(10, 3, 4) == (9, 10, 3) and v2a_amd.shard_range(2, 3, 4) == (2, 2, 1)
    covered = []
    for r in range(8):
        s, e, per = v2a_amd.shard_range(13, r, 8)
        covered += list(range(s, e))
    assert covered == list(range(13))


def test_tuned_tiles_table_names_existing_ab_logs():
    """dit.TUNED_TILES: one table of measured tile choices keyed by (mode, regime, widths); every entry names the A/B record that
    justifies it, and a record that is a file must exist under profiles/."""
    import os
    import re
    from v2a_amd.dit import TUNED_TILES, SHIPPED_WIDTHS, tuned_tiles
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    assert ("bf16", 0, SHIPPED_WIDTHS) in TUNED_TILES and ("bf16x3", 0, SHIPPED_WIDTHS) in TUNED_TILES
    for (mode, regime, widths), rows in TUNED_TILES.items():
        assert mode in ("bf16", "bf16x3") and regime in (0, 1, 2) and len(widths) == 4
        for (stream, op), (tile, log) in rows.items():
            assert stream in "atf" and op in ("x_tfa", "skip", "qkv", "out", "q2", "out2", "ff1", "ff2", "cross")
            assert 0 <= tile <= 15 and isinstance(log, str) and log
            for fn in re.findall(r"profiles/[\w.]+\.txt", log):
                assert os.path.isfile(os.path.join(root, fn)), fn
    assert tuned_tiles("bf16", 0, (64, 64, 64, 4)) == {}            # unmeasured widths: policy tiles only
    assert tuned_tiles("bf16", 0, SHIPPED_WIDTHS)[("t", "ff1")] == 6
