"""CPU: which BatchNorm modules the fused training BatchNorm treats as synchronised, and that the per-layer exchange protocol
and the peer-memory set-up are decided by the GROUP, not by one rank.
The two-rank GPU twin is tests/test_syncbn_lightning_gpu.py."""
import os

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from occdepth_amd import bn as _bn
from occdepth_amd import shard
from test_shard_gloo import _free_port


def test_both_converters_produce_modules_bn_act_synchronises():
    """`torch.nn.SyncBatchNorm.convert_sync_batchnorm` is what Lightning's Trainer(sync_batchnorm=True) applies
    (reference scripts/train.py:179,195); `shard.convert_sync_batchnorm` is prepare_for_ddp's."""
    net = torch.nn.Sequential(torch.nn.Conv2d(3, 8, 1), torch.nn.BatchNorm2d(8), torch.nn.Conv3d(8, 8, 1), torch.nn.BatchNorm3d(8))
    assert not any(_bn.is_sync(m) for m in net.modules())
    a = torch.nn.SyncBatchNorm.convert_sync_batchnorm(net)
    assert sum(_bn.is_sync(m) for m in a.modules()) == 2
    assert all(type(m) is torch.nn.SyncBatchNorm for m in a.modules() if _bn.is_sync(m))
    b = shard.convert_sync_batchnorm(torch.nn.Sequential(torch.nn.BatchNorm2d(8), torch.nn.BatchNorm1d(4)))
    assert sum(_bn.is_sync(m) for m in b.modules()) == 2
    # no process group: synchronised module, nothing to exchange, no protocol record
    sync, group, proto = _bn._sync_protocol(torch.nn.SyncBatchNorm(8), torch.zeros(2, 8, 4, 4))
    assert sync and group is None and proto is None
    assert _bn._sync_protocol(torch.nn.BatchNorm2d(8), torch.zeros(2, 8, 4, 4)) == (False, None, None)


class _FakeExchange:
    def channel_args(self, C, device):
        return ()


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    out = {}
    # (1) the agreement primitive
    out["agree_mixed"] = shard.agree_flag(rank == 0)
    out["agree_all"] = shard.agree_flag(True)
    # (2) no GPU here: the lazy set-up declines on every rank alike and torch's SyncBatchNorm module still reports the group
    layer = torch.nn.SyncBatchNorm(64)
    sync, group, proto = _bn._sync_protocol(layer, torch.zeros(2, 64, 8, 8))
    out["cpu_proto"] = (sync, group is None, dict(proto))
    # (3) an explicit install must fail on EVERY rank with the same error (stage 0: no GPU), nobody left waiting in a collective
    try:
        shard.install_small_all_reduce(dist)
        out["install"] = "installed"
    except RuntimeError as e:
        out["install"] = str(e)
    shard._SMALL_TRIED.discard(None)
    # (4) per-layer protocol: rank 1's tensor does not fit the one-launch kernel -> BOTH ranks take the packed all-reduce;
    #     a layer that fits everywhere -> both take the one-launch exchange; a later oversized tensor on one rank is an error
    real = shard.ensure_small_all_reduce
    shard.ensure_small_all_reduce = lambda group=None, device=None: _FakeExchange()
    try:
        la, lb = torch.nn.SyncBatchNorm(64), torch.nn.SyncBatchNorm(64)
        small, big = torch.zeros(2, 64, 8, 8), torch.zeros(2, 64, 128, 128)
        out["proto_mixed"] = _bn._sync_protocol(la, small if rank == 0 else big)[2]["one_launch"]
        out["proto_small"] = _bn._sync_protocol(lb, small)[2]["one_launch"]
        out["proto_sticky"] = _bn._sync_protocol(lb, big)[2]["one_launch"]           # decided once per layer
        out["follow_packed"] = _bn._BNActFn._agreed_small(True, False, small)       # fits locally, group said packed
        try:
            _bn._BNActFn._agreed_small(False, True, big)
            out["outgrown"] = "no error"
        except RuntimeError as e:
            out["outgrown"] = "error"
    finally:
        shard.ensure_small_all_reduce = real
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_agree_on_protocol_and_on_set_up_failure():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in range(world):
        o = got[r]
        assert o["agree_mixed"] is False and o["agree_all"] is True
        assert o["cpu_proto"] == (True, True, {"one_launch": False, "installed": False})
        assert o["install"].startswith("SmallAllReduce: not every rank can take part"), o["install"]
        assert o["proto_mixed"] is False and o["proto_small"] is True and o["proto_sticky"] is True
        assert o["follow_packed"] is False and o["outgrown"] == "error"
    assert got[0]["install"] == got[1]["install"]


def test_poll_without_exchanges_is_a_no_op():
    shard.poll_exchanges()
