"""CPU restatement of the Video2Roll frame encoder.      *** TEST INFRASTRUCTURE ***

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the product
path (video-to-audio-and-piano-rp_amd/video2roll.py + libv2a_cfm.so) never does.

PARITY PINNED: unlike the sampler oracle, this path's reference file is importable in the build container
(`src/audeo/Video2RollNet.py` needs only torch + math), so `oracle/make_golden_video2roll.py` runs the
REFERENCE module on seeded weights / inputs and commits its outputs under tests/golden/video2roll_*.npz;
tests/test_video2roll_oracle.py checks this restatement against those vectors.

What is restated (plain torch fp32 functional ops, state-dict keys exactly as the reference module's):
  * `ResNet.forward`            src/audeo/Video2RollNet.py:195-251  (resnet18 layout, :254-258)
  * `BasicBlock.forward`        src/audeo/Video2RollNet.py:70-88
  * `FTB.forward`               src/audeo/Video2RollNet.py:24-36    (1x1 conv with padding=1, avg-pool 2x2/2 or 3x3/1)
  * `FRB.forward`               src/audeo/Video2RollNet.py:44-57    (squeeze-excite style channel gate)
  * `E2TTS.encode_frames`       src/e2_tts_pytorch/e2_tts_crossatt3.py:1525-1553 (5-frame clamped window, sigmoid,
                                 x3 temporal repeat, crop / zero-pad to the latent length)
BatchNorm runs in eval mode (running statistics), as under `E2TTS.sample` (`self.eval()`, x3:2153).
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

NOTES = 51
BN_EPS = 1e-5
LAYERS = (("layer1", 64, 1), ("layer2", 128, 2), ("layer3", 256, 2), ("layer4", 512, 2))


def param_shapes(num_classes: int = NOTES) -> dict[str, tuple]:
    """state_dict layout of `Video2RollNet.resnet18(num_classes=51)` (src/audeo/Video2RollNet.py:127-168, 254-258)."""
    s: dict[str, tuple] = {}

    def bn(p, c):
        for k in ("weight", "bias", "running_mean", "running_var"):
            s[f"{p}.{k}"] = (c,)
        s[f"{p}.num_batches_tracked"] = ()

    s["conv1.weight"] = (64, 5, 11, 11)
    bn("bn1", 64)
    inpl = 64
    for name, planes, stride in LAYERS:
        for b in range(2):
            st = stride if b == 0 else 1
            p = f"{name}.{b}"
            s[f"{p}.conv1.weight"] = (planes, inpl, 3, 3)
            bn(f"{p}.bn1", planes)
            s[f"{p}.conv2.weight"] = (planes, planes, 3, 3)
            bn(f"{p}.bn2", planes)
            if b == 0 and (st != 1 or inpl != planes):
                s[f"{p}.downsample.0.weight"] = (planes, inpl, 1, 1)
                bn(f"{p}.downsample.1", planes)
            inpl = planes
    for name, cin in (("FTB2_1", 128), ("FTB2_2", 128), ("FTB3", 256), ("FTB4", 512)):
        s[f"{name}.conv0.weight"] = (128, cin, 1, 1)
        s[f"{name}.conv1.weight"] = (128, 128, 3, 3)
        bn(f"{name}.bn1", 128)
        s[f"{name}.conv2.weight"] = (128, 128, 3, 3)
    for name, c1 in (("FRB2", 128), ("FRB3", 128), ("FRB4", 64)):
        s[f"{name}.fc1.weight"] = (128, c1 + 128)
        s[f"{name}.fc1.bias"] = (128,)
        s[f"{name}.fc2.weight"] = (128, 128)
        s[f"{name}.fc2.bias"] = (128,)
    s["toplayer.weight"] = (64, 512, 1, 1)
    s["toplayer.bias"] = (64,)
    bn("toplayer_bn", 64)
    s["conv2.weight"] = (128, 128, 1, 1)
    s["conv2.bias"] = (128,)
    s["fc.weight"] = (num_classes, 128)
    s["fc.bias"] = (num_classes,)
    return s


def _bn(P, p, x):
    return F.batch_norm(x, P[f"{p}.running_mean"], P[f"{p}.running_var"], P[f"{p}.weight"], P[f"{p}.bias"], False, 0.0, BN_EPS)


def basic_block(P, p, x, stride):
    """src/audeo/Video2RollNet.py:70-88."""
    out = F.relu(_bn(P, f"{p}.bn1", F.conv2d(x, P[f"{p}.conv1.weight"], None, stride, 1)))
    out = _bn(P, f"{p}.bn2", F.conv2d(out, P[f"{p}.conv2.weight"], None, 1, 1))
    res = x
    if f"{p}.downsample.0.weight" in P:
        res = _bn(P, f"{p}.downsample.1", F.conv2d(x, P[f"{p}.downsample.0.weight"], None, stride, 0))
    return F.relu(out + res)


def ftb(P, p, x, avg=True):
    """src/audeo/Video2RollNet.py:24-36: the 1x1 conv0 has padding=1, so it grows the map by a zero border."""
    x1 = F.conv2d(x, P[f"{p}.conv0.weight"], None, 1, 1)
    out = F.relu(_bn(P, f"{p}.bn1", F.conv2d(x1, P[f"{p}.conv1.weight"], None, 1, 1)))
    out = F.conv2d(out, P[f"{p}.conv2.weight"], None, 1, 1) + x1
    return F.avg_pool2d(out, 2, 2) if avg else F.avg_pool2d(out, 3, 1)


def frb(P, p, xl, xh):
    """src/audeo/Video2RollNet.py:44-57."""
    zc = torch.cat([xl, xh], 1).mean((2, 3))
    z = F.linear(F.relu(F.linear(zc, P[f"{p}.fc1.weight"], P[f"{p}.fc1.bias"])), P[f"{p}.fc2.weight"], P[f"{p}.fc2.bias"])
    return torch.sigmoid(z)[:, :, None, None] * xl


def resnet_forward(P, x, taps: dict | None = None):
    """src/audeo/Video2RollNet.py:195-251.  x (n, 5, H, W) -> logits (n, 51)."""
    h = F.relu(_bn(P, "bn1", F.conv2d(x, P["conv1.weight"], None, 2, 4)))
    h = F.max_pool2d(h, 3, 2, 1)
    feats = []
    for name, _, stride in LAYERS:
        h = basic_block(P, f"{name}.0", h, stride)
        h = basic_block(P, f"{name}.1", h, 1)
        feats.append(h)
    x1, x2, x3, x4 = feats
    x5 = F.relu(_bn(P, "toplayer_bn", F.conv2d(x4, P["toplayer.weight"], P["toplayer.bias"])))
    x2_ = ftb(P, "FTB2_2", ftb(P, "FTB2_1", x2))
    x3_ = ftb(P, "FTB3", x3)
    x4_ = ftb(P, "FTB4", x4, avg=False)
    p4 = frb(P, "FRB4", x4_, x5)
    p3 = frb(P, "FRB3", x3_, p4)
    p2 = frb(P, "FRB2", x2_, p3)
    out1 = p2 * p3
    a = F.softmax(out1.flatten(2), dim=2).view_as(out1)
    out = F.conv2d(a * p4, P["conv2.weight"], P["conv2.bias"]) + p4
    logits = F.linear(out.mean((2, 3)), P["fc.weight"], P["fc.bias"])
    if taps is not None:
        taps.update(x1=x1, x2=x2, x3=x3, x4=x4, x5=x5, x2_=x2_, x3_=x3_, x4_=x4_, p4=p4, p3=p3, p2=p2)
    return logits


def frame_windows(x):
    """x3:1531-1539: (b, 1, t, H, W) -> (b*t, 5, H, W); window i holds frames clamp(i-2 .. i+2, 0, t-1)."""
    b, c, t, H, W = x.shape
    assert c == 1
    idx = (torch.arange(t)[:, None] + torch.arange(-2, 3)[None, :]).clamp(0, t - 1)      # (t, 5)
    return x[:, 0][:, idx].reshape(b * t, 5, H, W)


def encode_frames(P, x, l: int):
    """x3:1525-1553: piano-roll probabilities at 3x the video frame rate, cropped / zero-padded to l rows."""
    b, _, t, _, _ = x.shape
    roll = torch.sigmoid(resnet_forward(P, frame_windows(x)))
    roll = roll.reshape(b, t, 1, NOTES).repeat(1, 1, 3, 1).reshape(b, t * 3, NOTES)
    d = roll.shape[1]
    if d > l:
        roll = roll[:, :l]
    elif d < l:
        roll = torch.cat((roll, torch.zeros(b, l - d, NOTES)), 1)
    return roll
