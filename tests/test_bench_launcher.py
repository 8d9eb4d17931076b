"""GPU: `bench.py` under the driver's own launcher, one rank (first contact with a multi-GPU node must not
be the first time this command line runs).  The driver starts N > 1 as

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A gpurun box has one GPU, so N = 1 here -- but everything a rank does under the launcher is exercised: RANK / LOCAL_RANK /
WORLD_SIZE / MASTER_* from the environment, the `--gpus` == WORLD_SIZE assertion, exactly ONE JSON line on stdout, and with
OCCDEPTH_FORCE_DIST=1 a real (single-rank) RCCL process group: SyncBatchNorm's packed collectives + gradient buckets, the
step including its collectives captured into the whole-step hipGraph.  Reference: scripts/train.py:176-206.
Each leg is a child process (a NCCL watchdog thread and later hipGraph captures do not share a process, see
tests/test_rccl_single_rank.py)."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _launch(bench_args, extra_env=None, timeout=600):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.update(extra_env or {})
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "1"] + bench_args
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout, cwd=ROOT)
    tail = "\n".join((r.stdout[-1500:] + "\n--- stderr ---\n" + r.stderr[-2500:]).splitlines()[-40:])
    assert r.returncode == 0, tail
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    json_lines = [l for l in lines if l.startswith("{")]
    assert len(json_lines) == 1 and lines[-1] == json_lines[0], f"stdout must carry exactly one JSON line, last:\n{tail}"
    return json.loads(json_lines[0])


def test_forward_bench_under_the_launcher():
    t = _launch(["--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-extras"])
    assert t["n_gpus"] == 1 and t["steps"] == 2 and t["warmup"] == 1 and t["scaling"] == "weak"
    assert t["config"]["ranks"] == 1 and t["config"]["frames_per_step"] == 1
    assert t["config"]["graph_all"] is True and "graph_all_error" not in t["config"]
    assert t["value"] > 0 and abs(t["value"] - 1e3 / t["ms_per_step"]) < 1e-6 * t["value"]
    assert t["parity_rel_err"]["worst_of_all_outputs"] < 1e-3 and t["parity_rel_err"]["lift_kernels"] == ["sfa_lift_proj"]
    assert t["roofline"]["frac"] > 0 and t["cpu_baseline"] is None


@pytest.mark.parametrize("mode", ["bf16", "fp32"])
def test_training_bench_under_the_launcher_on_a_forced_rccl_group(mode):
    """`--train` through a single-rank RCCL group (OCCDEPTH_FORCE_DIST=1): SyncBatchNorm + buckets wrap the model, their
    collectives are captured into the whole-step hipGraph (OCCDEPTH_TRAIN_GRAPH_DDP=1) and the replayed step trains."""
    args = ["--train", "--steps", "2", "--warmup", "1"] + (["--bf16"] if mode == "bf16" else [])
    t = _launch(args, {"OCCDEPTH_FORCE_DIST": "1", "OCCDEPTH_TRAIN_GRAPH_DDP": "1"}, timeout=900)
    assert t["n_gpus"] == 1 and t["config"]["ranks"] == 1 and t["config"]["global_batch"] == 1
    assert "SyncBatchNorm" in t["config"]["parallelism"] and "gradient buckets" in t["config"]["parallelism"]
    assert not t["config"]["parallelism"].endswith("0 gradient buckets (none)")
    assert t["train_graph"] is True and t["train_graph_error"] is None, t["train_graph_error"]
    assert t["loss"] == t["loss"] and 0 < t["loss"] < 1e4            # finite
    assert t["ms_per_step"] > 0
