"""Two-rank rehearsal of the REAL sampler under torch.distributed on one GPU box (RCCL itself needs an N-GPU node).

Started by the launcher, which touches no GPU itself:
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29531 scripts/dist_rehearsal.py
Every rank builds the same small model, takes its contiguous shard of a seeded batch (dist.shard_range), runs
`E2TTS.sample` with the hipGraph on (thread-local capture mode beside the collective library's threads), and
`gather_latents` reassembles the batch with ONE all-gather.  Each rank then samples the WHOLE batch in its own process and
asserts the gathered result equals it -- a clip's latents do not depend on the rank or the shard that produced them.

Backend: gloo when the ranks share a device (this pipeline's one-GPU box), nccl (= RCCL) when every rank has its own GPU.
Prints one line per rank and "REHEARSAL OK" from rank 0; any mismatch raises (non-zero exit of the launcher).
"""
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    rank, world, local = (int(os.environ.get(k, d)) for k, d in (("RANK", 0), ("WORLD_SIZE", 1), ("LOCAL_RANK", 0)))
    ndev = torch.cuda.device_count()
    assert ndev >= 1, "needs a GPU"
    shared = ndev < world                      # ranks share a device: collectives over gloo (host staging in gather_latents)
    dev = torch.device("cuda", local % ndev)
    torch.cuda.set_device(dev)
    if world > 1:
        if shared:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
    import v2a_amd
    from v2a_amd.synth import random_state_dict, synthetic_conditioning
    mode = sys.argv[1] if len(sys.argv) > 1 else "bf16"
    n_clips = int(sys.argv[2]) if len(sys.argv) > 2 else 5           # odd on purpose: ragged shards, one pad clip dropped
    cfg = v2a_amd.DiTConfig(dim=256, dim_text=320, dim_frames=128, depth=4, heads=4, frames_heads=2, num_registers=8, num_channels=32,
                            max_seq_len=512)
    T, NC, steps = 120, 12, 8
    sd = random_state_dict(cfg, seed=0, device="cpu")
    tk = {k: v for k, v in cfg.to_dict().items() if k not in ("num_channels", "notes", "cond_proj_in", "dim_context", "kernel_size", "ff_mult")}
    model = v2a_amd.E2TTS(transformer=dict(if_text_modules=True, if_cross_attn=True, if_audio_conv=True, if_text_conv=True, **tk),
                          num_channels=cfg.num_channels, sampling_rate=24000, if_cond_proj_in=False, compute_dtype=mode, device=dev,
                          use_graph=True)
    model.load_state_dict(sd, strict=False)
    y0, text, roll, ctx, cm = synthetic_conditioning(cfg, n_clips, T, NC, seed=77, piano=True, device="cpu")   # same on every rank
    kw = dict(steps=steps, cfg_strength=2.0, remove_parallel_component=False, sway_sampling=True, return_raw_output=True)

    def run(lo, hi):
        return model.sample(torch.zeros(hi - lo, T, cfg.num_channels), y0=y0[lo:hi], text_embed=text[lo:hi], context=ctx[lo:hi],
                            context_mask=cm[lo:hi], frames_embed=roll[lo:hi], **kw).to(dev)

    s, e, per = v2a_amd.shard_range(n_clips, rank, world)
    t0 = time.time()
    reps = []
    for rep in range(2):                       # second pass: plan + graph cache hit, replay next to a live process group
        mine = run(s, e) if e > s else torch.zeros(0, T, cfg.num_channels, device=dev)
        reps.append(mine.clone())
        got = v2a_amd.gather_latents(mine, n_clips, per)
    t1 = time.time()
    assert got.shape == (n_clips, T, cfg.num_channels) and bool(torch.isfinite(got).all())
    whole = run(0, n_clips)                    # the same clips as ONE single-process batch
    d = float((got - whole).abs().max())
    if d != 0.0:                               # say where: which clips, which pass, own shard or the other rank's
        per_clip = [float((got[i] - whole[i]).abs().max()) for i in range(n_clips)]
        print(f"rank {rank}: per-clip max |delta| gathered vs whole {per_clip}; pass 0 vs pass 1 of the own shard "
              f"{float((reps[0] - reps[1]).abs().max()) if e > s else 0.0:.3e}; own shard (pass 1) vs whole "
              f"{float((reps[1] - whole[s:e]).abs().max()) if e > s else 0.0:.3e}; (pass 0) {float((reps[0] - whole[s:e]).abs().max()) if e > s else 0.0:.3e}",
              flush=True)
    # batch-size independence: fp32 / bf16x3 results are bit-equal; bf16 GEMMs may pick another tile shape for another row
    # count, which keeps every element's K order -- equal as well
    print(f"rank {rank}/{world} [{mode}] device {dev} backend {dist.get_backend() if world > 1 else 'none'}: shard [{s},{e}) of {n_clips} clips, "
          f"gathered vs single-process batch max |delta| = {d:.3e}, captures {model.graph_captures}, {t1 - t0:.1f} s", flush=True)
    assert d == 0.0, d
    assert model.graph_captures == (2 if e > s else 1), model.graph_captures        # shard shape + whole-batch shape, one capture each
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print("REHEARSAL OK", flush=True)


if __name__ == "__main__":
    main()
