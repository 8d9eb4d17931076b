"""CPU, world_size 2 over gloo: the N>1 path of bench.py -- frame sharding without a data-path collective,
barrier + MAX-over-ranks timing -- gives the same per-frame results as one process."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import golden_cases as gc
from occdepth_amd import shard


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _frame(i):
    """A per-frame workload that is cheap on CPU: the oracle's Stereo-SFA on frame-specific features."""
    from oracle import occdepth_oracle as orc
    spec = dict(gc.SFA_CASES["kitti_v2"], C=8 + 0 * i)
    x2d, pix, fov = gc.sfa_inputs(spec)
    return orc.sfa(x2d * (i + 1), pix, fov, spec["scene"], spec["ps"], spec["dataset"])


def _worker(rank, world, port, n_frames, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mine = shard.frames_for_rank(n_frames, rank, world)
    shard.fence(dist)
    local = {i: _frame(i) for i in mine}
    shard.fence(dist)
    elapsed = shard.max_over_ranks(1.0 + rank, dist)          # rank 1 is "slower": MAX must win on both
    frames = shard.gather_frames(local, n_frames, dist)
    if rank == 0:
        q.put((mine, elapsed, [f.clone() for f in frames]))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharding_matches_single_process():
    n_frames, world = 5, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_frames, q)) for r in range(world)]
    for p in procs:
        p.start()
    mine, elapsed, frames = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert mine == [0, 2, 4]
    assert elapsed == 2.0
    for i in range(n_frames):
        assert torch.equal(frames[i], _frame(i))


def test_shards_partition_the_frames():
    for world in (1, 2, 3, 8):
        got = sorted(i for r in range(world) for i in shard.frames_for_rank(11, r, world))
        assert got == list(range(11))
    assert shard.max_over_ranks(3.5) == 3.5


# ---------------------------------------------------------------------------------------------------------------
def _toy():
    torch.manual_seed(4)
    m = torch.nn.Sequential(torch.nn.Linear(6, 16), torch.nn.ReLU(), torch.nn.Linear(16, 16), torch.nn.ReLU(),
                            torch.nn.Linear(16, 3))
    m.unused = torch.nn.Parameter(torch.ones(5))              # never touched by forward (cf. SURVEY 2.1 dead branches)
    return m


def _toy_batch(rank):
    g = torch.Generator().manual_seed(100 + rank)
    return torch.randn(7, 6, generator=g), torch.randn(7, 3, generator=g)


def _train_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    m = _toy()
    buckets = shard.GradBuckets(m.parameters(), dist, bucket_bytes=600)      # several small buckets
    out = []
    for step in range(2):                                                    # buckets are reusable
        m.zero_grad(set_to_none=True)
        x, y = _toy_batch(rank + 10 * step)
        ((m(x) - y) ** 2).mean().backward()
        buckets.finish()
        out.append({k: p.grad.clone() for k, p in m.named_parameters()})
    hist = torch.full((3, 3), rank + 1, dtype=torch.int64)
    shard.allreduce_confusion(hist, dist)
    q.put((rank, out, hist, len(buckets.buckets)))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gradient_buckets_average_like_one_process():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_train_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict()
    for _ in range(world):
        rank, out, hist, nb = q.get(timeout=120)
        got[rank] = (out, hist, nb)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert got[0][2] > 2                                                     # the toy really spans several buckets
    assert torch.equal(got[0][1], torch.full((3, 3), 3, dtype=torch.int64))
    for step in range(2):
        want = None
        for r in range(world):                                               # single-process mean of the per-rank grads
            m = _toy()
            x, y = _toy_batch(r + 10 * step)
            ((m(x) - y) ** 2).mean().backward()
            g = {k: (p.grad if p.grad is not None else torch.zeros_like(p)) for k, p in m.named_parameters()}
            want = g if want is None else {k: want[k] + g[k] for k in g}
        for r in range(world):
            for k, v in got[r][0][step].items():
                assert torch.allclose(v, want[k] / world, atol=1e-7), (step, r, k)
