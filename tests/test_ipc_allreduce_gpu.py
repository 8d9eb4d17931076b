"""-m gpu: the peer-memory small-message all-reduce (occdepth_amd.shard.SmallAllReduce, csrc/ipc_allreduce.hip).
  * one process, world size 1: sequence numbers, both dtypes, argument checks, hipGraph replay;
  * TWO processes sharing the box's single GPU (IPC handles work where RCCL refuses duplicate devices): exact sums over a
    series of exchanges, captured-graph replays, SyncBatchNorm forward / backward equal to the gloo exchange.
Every wait is bounded (kernel budget + subprocess timeout): a protocol bug fails the test, it cannot hang the GPU."""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run_world(world, timeout=240):
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "ipc_worker.py")], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = []
    try:
        for p in procs:
            out, _ = p.communicate(timeout=timeout)
            outs.append((p.returncode, out))
    except subprocess.TimeoutExpired:
        for p in procs:
            p.kill()
        raise AssertionError("ipc workers timed out:\n" + "\n".join(o for _, o in outs))
    return outs


@pytest.mark.parametrize("world", [1, 2])
def test_small_all_reduce_processes_sharing_one_gpu(world):
    outs = _run_world(world)
    results = []
    for rc, out in outs:
        line = [l for l in out.splitlines() if l.startswith("IPC_RESULT ")]
        assert rc == 0 and line, out[-3000:]
        results.append(json.loads(line[-1][len("IPC_RESULT "):]))
    for r in results:
        assert r["world"] == world and r["series_max_abs_diff"] == 0.0 and r["graph_replays_exact"]
        assert r["syncbn_max_diff"] <= 2e-6
        tags = r["syncbn_tags"]
        assert "ipc_allreduce" in tags["48x9x20"] and "ipc_allreduce" in tags["96x80x80"], tags
        # the small layer: ONE launch per direction, the exchange inside it
        assert tags["96x9x20"] == ["bn_bwd_small_xchg", "bn_fwd_small_xchg"], tags
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", f"ipc_allreduce_world{world}.json"), "w") as f:
        json.dump(results, f)
    print("ipc all-reduce:", [(r["rank"], round(r["us_per_exchange"], 2)) for r in results], "us per exchange")


def test_small_all_reduce_argument_checks():
    from occdepth_amd import hip
    lib = hip.load()
    assert lib.occd_ipc_mailbox_bytes(0, 1024) < 0 and lib.occd_ipc_mailbox_bytes(17, 1024) < 0
    assert lib.occd_ipc_mailbox_bytes(2, 1024) == 256 + 2 * 2 * (2 * 1024)        # LL: 8-byte word per 32-bit half
    assert lib.occd_ipc_allreduce(None, None, 1, 0, None, 0, 1, 1024, 0, None, None) == -1
