"""Batched V2A inference CLI.

Same positional arguments as the reference's `src/inference_v2a.py:3-11`
    ckpt  drop_prompt(0|1)  test_scp  start  end  out_dir
(`test_scp`: one `video_path<TAB>caption` per line, tests/vgg_test.scp) plus batching: clips [start, end)
are collated `--batch` at a time (the reference samples one clip per call, src/inference_v2a.py:157,183) and,
under torchrun, sharded contiguously over the ranks with ONE all-gather of the latents per batch.

What it needs next to each video: the cached CLIP features `<video>.generated.npz` (features.py) and a cached
FLAN-T5 context `<video>.t5.npz` (arr_0 = (nc, 1024) hidden states) unless `--t5 ./ckpts/flan-t5-large` points at
local T5 weights.  Output: `<out_dir>/<name>.latent.npy`, the (n, 128) Encodec latent that the reference feeds to
`vocos.decode` (src/inference_v2a.py / predict.py:277-278).

  --piano            V2P (src/inference_v2p.py): the cached grey frames `<video>.generated_frames_raw.2.npz` (features.py) go
                     through the HIP Video2Roll encoder; the checkpoint must hold `video2roll_net.*`
  --encodec STATE    torch-saved state dict of `EncodecModel.from_pretrained("facebook/encodec_24khz")` (or of its decoder):
                     each clip's valid frames are decoded by the HIP vocoder and written as `<name>.wav` (24 kHz float32),
                     what the reference does with `save_to_filename` / torchaudio.save (x3:2291-2303, predict.py:279-281).
The moviepy mux of audio and video stays outside (out of scope).
"""
from __future__ import annotations

import argparse
import os
import sys

import numpy as np
import torch

LATENT_RATE = 24000 / 320          # torch_tools.py:32-40


def read_scp(path: str, start: int, end: int, step: int = 1):
    out = []
    with open(path) as f:
        for ln in f.read().splitlines():
            if ln.strip():
                p, _, cap = ln.partition("\t")
                out.append((p, cap))
    return out[start:end:step]


def build_requests(items, drop_prompt: bool, n_frames: int, t5_encode=None):
    from .collate import ClipRequest
    from .features import feature_cache_path, load_clip_cache, resample_clip_features
    reqs = []
    for vp, cap in items:
        emb, duration = load_clip_cache(feature_cache_path(vp))
        n = min(n_frames, int(duration * 24000) // 320) if n_frames > 0 else int(duration * 24000) // 320
        clip = resample_clip_features(emb.float(), duration, n)
        prompt = "" if drop_prompt else cap
        t5p = vp.replace(".mp4", ".t5.npz")
        if t5_encode is not None:
            ctx = t5_encode(prompt if prompt else "the sound of X X")        # x3:2053-2056
        elif os.path.exists(t5p):
            ctx = torch.from_numpy(np.load(t5p)["arr_0"]).float()
        else:
            raise FileNotFoundError(f"{t5p}: no cached T5 context and no --t5 model")
        reqs.append(ClipRequest(vp, prompt, n, clip, ctx))
    return reqs


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("ckpt")
    ap.add_argument("drop_prompt", type=int)
    ap.add_argument("test_scp")
    ap.add_argument("start", type=int)
    ap.add_argument("end", type=int)
    ap.add_argument("out_dir")
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--steps", type=int, default=64)                 # src/inference_v2a.py:183
    ap.add_argument("--cfg-strength", type=float, default=2.0)
    ap.add_argument("--frames", type=int, default=750, help="latent frames per clip (10 s)")
    ap.add_argument("--dtype", default="bf16x3", choices=["bf16", "bf16x3", "fp32"],
                    help="bf16: fastest, |delta mel| ~ 5e-2 vs the fp32 reference arithmetic; bf16x3: split-bf16 products, < 1e-3 at ~0.4x the "
                         "bf16 speed; fp32: exact-fp32 MFMA, < 1e-3 at ~0.18x")
    ap.add_argument("--bucket-frames", type=int, default=64, help="pad plans to a multiple of this many latent frames (0 = exact shapes): "
                    "durations vary per clip, and every new shape costs a plan and a hipGraph capture")
    ap.add_argument("--bucket-ctx", type=int, default=16, help="pad the T5 context to a multiple of this many tokens (0 = exact)")
    ap.add_argument("--t5", default=None, help="local FLAN-T5 directory (reference: ./ckpts/flan-t5-large)")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--model-config", default=None, help="JSON dict of transformer kwargs (default: predict.py:120-134)")
    ap.add_argument("--piano", action="store_true", help="V2P: condition on the cached piano frames through the Video2Roll encoder")
    ap.add_argument("--encodec", default=None, help="state dict (.pt) of the Encodec model / decoder: also write <name>.wav")
    a = ap.parse_args(argv)

    import torch.distributed as dist
    from . import E2TTS, collate_clips, gather_latents, shard_range
    rank, world, local = (int(os.environ.get(k, d)) for k, d in (("RANK", 0), ("WORLD_SIZE", 1), ("LOCAL_RANK", 0)))
    if world > 1:
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    tk = dict(depth=12, dim=1024, dim_text=1280, heads=16, dim_head=64)
    if a.model_config:
        import json
        tk = json.loads(a.model_config)
    channels = tk.pop("num_channels", 128)
    model = E2TTS(transformer=dict(if_text_modules=True, if_cross_attn=True, if_audio_conv=True, if_text_conv=True, **tk),
                  num_channels=channels, sampling_rate=24000, if_cond_proj_in=False, tokenizer="phoneme_zh",
                  compute_dtype=a.dtype, device=torch.device("cuda", local), bucket_frames=a.bucket_frames, bucket_ctx=a.bucket_ctx)
    ck = torch.load(a.ckpt, map_location="cpu")
    res = model.load_state_dict(ck.get("model_state_dict", ck), strict=False)      # predict.py:161-168
    if res.missing_keys:
        raise RuntimeError(f"checkpoint lacks {len(res.missing_keys)} parameters of the sampled path, e.g. {res.missing_keys[0]}")
    t5_encode = None
    if a.t5:
        from transformers import AutoTokenizer, T5EncoderModel
        tok, enc = AutoTokenizer.from_pretrained(a.t5), T5EncoderModel.from_pretrained(a.t5).eval()
        def t5_encode(prompt):
            b = tok([prompt], max_length=tok.model_max_length, padding=True, truncation=True, return_tensors="pt")
            with torch.no_grad():
                return enc(input_ids=b.input_ids, attention_mask=b.attention_mask)[0][0]
    vocoder = None
    if a.encodec and rank == 0:
        from .encodec import EncodecDecoder
        vocoder = EncodecDecoder(torch.load(a.encodec, map_location="cpu"), torch.device("cuda", local))
    items = read_scp(a.test_scp, a.start, a.end)
    os.makedirs(a.out_dir, exist_ok=True)
    gen = torch.Generator().manual_seed(a.seed)
    written = []
    for b0 in range(0, len(items), a.batch):
        chunk = items[b0:b0 + a.batch]
        s, e, per = shard_range(len(chunk), rank, world)
        mine = chunk[s:e]
        if mine:
            batch8, extras = collate_clips(build_requests(mine, bool(a.drop_prompt), a.frames, t5_encode), channels, gen)
            frames = None
            if a.piano:
                from .features import load_piano_frames
                frames = load_piano_frames([vp for vp, _ in mine], int(batch8[3].max()))       # x3:1829, predict.py:231
            lat = model.sample(batch8[1], lens=batch8[3], duration=batch8[3], steps=a.steps, cfg_strength=a.cfg_strength,
                               remove_parallel_component=False, sway_sampling=True, video_drop_prompt=batch8[4],
                               return_raw_output=True, frames=frames, **extras).to(torch.device("cuda", local))
        else:
            lat = torch.zeros(0, a.frames, channels, device=torch.device("cuda", local))
        if lat.shape[1] < a.frames:
            lat = torch.nn.functional.pad(lat, (0, 0, 0, a.frames - lat.shape[1]))
        allat = gather_latents(lat, len(chunk), per)
        if rank == 0:
            for (vp, _), one in zip(chunk, allat):
                name = vp.rsplit("/", 1)[-1].rsplit(".", 1)[0]
                path = os.path.join(a.out_dir, name + ".latent.npy")
                np.save(path, one.float().cpu().numpy())
                written.append(path)
                if vocoder is not None:
                    from scipy.io import wavfile
                    from .features import load_clip_cache, feature_cache_path
                    n = min(a.frames, int(load_clip_cache(feature_cache_path(vp))[1] * 24000) // 320) if a.frames > 0 else one.shape[0]
                    wav = vocoder.decode(one[:n].t()[None].float())[0]                     # predict.py:277-278
                    wavfile.write(os.path.join(a.out_dir, name + ".wav"), 24000, wav.cpu().numpy())
    if world > 1:
        dist.destroy_process_group()
    return written


if __name__ == "__main__":
    main()
