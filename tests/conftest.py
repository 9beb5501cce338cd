import json
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope="session")
def golden():
    out = {}
    for name in ("forward_small", "sample_small", "blocks_small"):
        out[name] = dict(np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False))
    out["meta"] = json.loads(str(out["forward_small"]["meta"]))
    with open(os.path.join(GOLDEN, "forward_full_stats.json")) as f:
        out["full_stats"] = json.load(f)
    return out


@pytest.fixture(scope="session")
def small(golden):
    """Small config + seeded oracle parameters + golden inputs as torch tensors."""
    from oracle import e2_cfm_oracle as O
    meta = golden["meta"]
    cfg = O.DiTConfig(**meta["cfg"])
    P = O.init_params(cfg, meta["param_seed"])
    g = golden["forward_small"]
    inp = dict(y0=torch.from_numpy(g["y0"]), text=torch.from_numpy(g["text"]), roll=torch.from_numpy(g["roll"]),
               ctx=torch.from_numpy(g["ctx"]), ctx_mask=torch.from_numpy(g["ctx_mask"]))
    return dict(cfg=cfg, P=P, inp=inp, meta=meta)


def make_model(cfg, P, compute="fp32", **kw):
    """Product-side E2TTS loaded from an oracle/reference-layout state_dict."""
    import v2a_amd
    m = v2a_amd.E2TTS(transformer=dict(dim=cfg.dim, dim_text=cfg.dim_text, dim_frames=cfg.dim_frames, depth=cfg.depth,
                                       heads=cfg.heads, dim_head=cfg.dim_head, frames_heads=cfg.frames_heads,
                                       num_registers=cfg.num_registers, max_seq_len=cfg.max_seq_len,
                                       if_text_modules=True, if_cross_attn=True, if_audio_conv=True, if_text_conv=True),
                      num_channels=cfg.num_channels, sampling_rate=24000, if_cond_proj_in=bool(getattr(cfg, "cond_proj_in", False)),
                      compute_dtype=compute, **kw)
    res = m.load_state_dict(P, strict=False)
    assert not res.missing_keys, res.missing_keys[:3]
    return m
