"""CPU: the N > 1 path (clip sharding + ONE all-gather) over gloo: world sizes 2 and 8, even and ragged batches, and bench.py's own
N-rank control path (`--stand-in-sampler`) launched the way the driver launches it."""
import json
import subprocess
import sys
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n_clips, q):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import v2a_amd
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    T, C = 6, 4
    s, e, per = v2a_amd.shard_range(n_clips, rank, world)
    # stand-in for sample(): each clip's latent is a deterministic function of its global index
    local = torch.stack([torch.full((T, C), float(i)) + torch.arange(C) for i in range(s, e)]) if e > s else torch.zeros(0, T, C)
    out = v2a_amd.gather_latents(local, n_clips, per)
    ok = out.shape == (n_clips, T, C) and all(torch.equal(out[i], torch.full((T, C), float(i)) + torch.arange(C)) for i in range(n_clips))
    q.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


def _run(n_clips, world=2):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, world, port, n_clips, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=240) for _ in ps)
    for p in ps:
        p.join(timeout=60)
    assert res == [(r, True) for r in range(world)], res


def test_gather_even_shards():
    _run(4)


def test_gather_ragged_shards_padded_and_dropped():
    _run(3)


def test_gather_world8_configs2_batch():
    """BASELINE configs[2]: 64 clips on 8 ranks, 8 per rank."""
    _run(64, world=8)


def test_gather_world8_ragged_batch_is_padded_and_dropped():
    """61 clips on 8 ranks: shards of 8 with the last rank holding 5 (pad to a multiple of the world size and drop)."""
    _run(61, world=8)


def test_shard_range_partitions_every_batch():
    import v2a_amd
    for world in (1, 2, 4, 8):
        for n in (1, 5, 8, 61, 64):
            spans = [v2a_amd.shard_range(n, r, world) for r in range(world)]
            per = spans[0][2]
            assert per * world >= n and all(s[2] == per for s in spans)
            covered = [i for lo, hi, _ in spans for i in range(lo, hi)]
            assert covered == list(range(n)), (world, n, spans)


def _bench_standin(world, clips=None, extra=()):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    if clips is not None:
        env["V2A_STANDIN_CLIPS"] = str(clips)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", str(world), "--steps", "3", "--warmup", "1",
           "--stand-in-sampler", *extra]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env, cwd=root)
    assert out.returncode == 0, out.stderr[-1500:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-500:]                   # rank 0 prints ONE JSON line
    return json.loads(lines[0])


def test_bench_two_rank_control_path_end_to_end():
    """`bench.py --gpus 2` as the driver launches it (torch.distributed.run, one process per rank), with the sampler replaced by a
    stand-in on CPU tensors: sharding, barriers, the timed loop, MAX over ranks, ONE all-gather and the like-for-like `scaling_reference`
    are bench.py's own code."""
    r = _bench_standin(2)
    assert r["n_gpus"] == 2 and r["steps"] == 3 and r["warmup"] == 1 and r["scaling"] == "weak"
    assert r["config"]["parallelism"] == "clip-sharded x2, 1 all-gather" and r["config"]["global_clips"] == 16 and r["config"]["clips_per_gpu"] == 8
    assert r["standin_gather_ok"] and r["standin_gather_ok_all_ranks"]
    sr = r["scaling_reference"]
    assert sr["clips_per_gpu"] == 8 and sr["n1_same_shape_mel_frames_per_s"] > 0
    assert abs(sr["efficiency"] - r["value"] / (2 * sr["n1_same_shape_mel_frames_per_s"])) < 1e-3
    assert "stand-in" in r["data"]


def test_bench_eight_rank_control_path_ragged_batch():
    """8 ranks, 61 clips: every rank ends with the full (61, T, C) tensor in clip order."""
    r = _bench_standin(8, clips=61)
    assert r["n_gpus"] == 8 and r["config"]["global_clips"] == 61 and r["config"]["clips_per_gpu"] == 8
    assert r["standin_gather_ok_all_ranks"] and r["config"]["parallelism"] == "clip-sharded x8, 1 all-gather"


def test_single_process_passthrough():
    import v2a_amd
    x = torch.randn(3, 5, 2)
    assert torch.equal(v2a_amd.gather_latents(x, 3, 3), x)
