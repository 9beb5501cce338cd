"""CPU restatement of the Encodec (24 kHz) decoder used as the vocoder.   *** TEST INFRASTRUCTURE ***

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.

Reference call sites: `EncodecWrapper.decode` src/e2_tts_pytorch/e2_tts_crossatt3.py:434-437 (`self.model.decoder(emb)`,
`output[0]`), called as `e2tts.vocos.decode(outputs.transpose(-1,-2))` at predict.py:277-278.  The arithmetic lives in the
third-party `transformers.models.encodec.modeling_encodec.EncodecDecoder` (reference pin transformers==4.46.0,
requirements.txt:20; the image holds 5.15.0, whose decoder is the same SEANet stack).

PARITY PINNED against that library: oracle/make_golden_encodec.py instantiates the library's `EncodecModel(EncodecConfig())`
(the facebook/encodec_24khz architecture: the checkpoint itself is unreachable offline), loads seeded weights and commits
the decoder's outputs under tests/golden/encodec_*.npz; tests/test_encodec_oracle.py checks this file against them.

Published algorithm restated (SEANet decoder, causal, weight-normalised convolutions):
  Conv1d(128->512, k7) -> 2-layer LSTM(512) + skip -> for r in (8, 5, 4, 2): ELU, ConvTranspose1d(C->C/2, k=2r, stride r),
  ResBlock[ELU, Conv(k3, C/2->C/4), ELU, Conv(k1, C/4->C/2)] + Conv(k1) shortcut -> ELU -> Conv1d(32->1, k7).
  Every Conv1d is causal: left reflect-padding of (k-1), stride 1, so lengths are preserved; every ConvTranspose1d output is
  trimmed by (k - stride) samples on the right, so lengths multiply by r: 750 latent frames -> 240 000 samples.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

RATIOS = (8, 5, 4, 2)
HIDDEN, FILTERS = 128, 32


def param_shapes() -> dict[str, tuple]:
    """state_dict layout of `EncodecModel(EncodecConfig()).decoder` (weight_norm parametrisation: original0 = g, original1 = v)."""
    s: dict[str, tuple] = {}

    def conv(p, co, ci, k, transpose=False):
        s[f"{p}.conv.bias"] = (co,)
        s[f"{p}.conv.parametrizations.weight.original0"] = ((ci if transpose else co), 1, 1)
        s[f"{p}.conv.parametrizations.weight.original1"] = (ci, co, k) if transpose else (co, ci, k)

    c = FILTERS * 2 ** len(RATIOS)
    conv("layers.0", c, HIDDEN, 7)
    for l in range(2):
        s[f"layers.1.lstm.weight_ih_l{l}"] = (4 * c, c)
        s[f"layers.1.lstm.weight_hh_l{l}"] = (4 * c, c)
        s[f"layers.1.lstm.bias_ih_l{l}"] = (4 * c,)
        s[f"layers.1.lstm.bias_hh_l{l}"] = (4 * c,)
    idx = 3
    for r in RATIOS:
        conv(f"layers.{idx}", c // 2, c, 2 * r, transpose=True)
        c //= 2
        conv(f"layers.{idx + 1}.block.1", c // 2, c, 3)
        conv(f"layers.{idx + 1}.block.3", c, c // 2, 1)
        conv(f"layers.{idx + 1}.shortcut", c, c, 1)
        idx += 3
    conv(f"layers.{idx}", 1, FILTERS, 7)
    return s


def wn_weight(P, p):
    """torch weight_norm (dim 0): w = g * v / ||v|| with the norm over all dims but the first."""
    if f"{p}.conv.weight" in P:
        return P[f"{p}.conv.weight"]
    g, v = P[f"{p}.conv.parametrizations.weight.original0"], P[f"{p}.conv.parametrizations.weight.original1"]
    return g * v / v.norm(dim=(1, 2), keepdim=True)


def conv1d(P, p, x):
    """EncodecConv1d, causal, stride 1, dilation 1: reflect-pad (k-1) on the left."""
    w = wn_weight(P, p)
    k = w.shape[-1]
    if k > 1:
        x = F.pad(x, (k - 1, 0), mode="reflect")
    return F.conv1d(x, w, P[f"{p}.conv.bias"])


def conv_transpose1d(P, p, x):
    """EncodecConvTranspose1d, causal, trim_right_ratio 1: the (k - stride) extra samples are cut on the right."""
    w = wn_weight(P, p)
    k = w.shape[-1]
    stride = k // 2
    y = F.conv_transpose1d(x, w, P[f"{p}.conv.bias"], stride=stride)
    return y[..., : y.shape[-1] - (k - stride)]


def lstm(P, p, x):
    """EncodecLSTM: 2-layer nn.LSTM over time + skip.  x (b, C, T)."""
    seq = x.permute(2, 0, 1)
    inp = seq
    for l in range(2):
        wi, wh = P[f"{p}.lstm.weight_ih_l{l}"], P[f"{p}.lstm.weight_hh_l{l}"]
        b = P[f"{p}.lstm.bias_ih_l{l}"] + P[f"{p}.lstm.bias_hh_l{l}"]
        H = wh.shape[1]
        h = torch.zeros(inp.shape[1], H)
        c = torch.zeros(inp.shape[1], H)
        gx = inp @ wi.t() + b
        outs = []
        for t in range(inp.shape[0]):
            g = gx[t] + h @ wh.t()
            i, f, gg, o = g.split(H, dim=1)
            c = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(gg)
            h = torch.sigmoid(o) * torch.tanh(c)
            outs.append(h)
        inp = torch.stack(outs, 0)
    return (inp + seq).permute(1, 2, 0)


def resblock(P, p, x):
    h = conv1d(P, f"{p}.block.1", F.elu(x))
    h = conv1d(P, f"{p}.block.3", F.elu(h))
    return conv1d(P, f"{p}.shortcut", x) + h


def decoder_forward(P, emb, taps: dict | None = None):
    """emb (b, 128, T) -> waveform (b, 1, 320*T)."""
    x = conv1d(P, "layers.0", emb)
    x = lstm(P, "layers.1", x)
    if taps is not None:
        taps["lstm"] = x
    idx = 3
    for r in RATIOS:
        x = conv_transpose1d(P, f"layers.{idx}", F.elu(x))
        x = resblock(P, f"layers.{idx + 1}", x)
        if taps is not None:
            taps[f"stage{r}"] = x
        idx += 3
    return conv1d(P, f"layers.{idx}", F.elu(x))


def decode(P, emb):
    """EncodecWrapper.decode (x3:434-437): `self.model.decoder(emb)[0]` -- the first clip's (1, samples) waveform."""
    return decoder_forward(P, emb)[0]
