"""CPU: the N > 1 path (clip sharding + ONE all-gather) with world_size 2 over gloo."""
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


def _run(n_clips):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, 2, port, n_clips, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=120) for _ in ps)
    for p in ps:
        p.join(timeout=60)
    assert res == [(0, True), (1, True)], res


def test_gather_even_shards():
    _run(4)


def test_gather_ragged_shards_padded_and_dropped():
    _run(3)


def test_single_process_passthrough():
    import v2a_amd
    x = torch.randn(3, 5, 2)
    assert torch.equal(v2a_amd.gather_latents(x, 3, 3), x)
