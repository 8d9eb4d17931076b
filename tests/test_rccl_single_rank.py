"""GPU, ONE rank on a real RCCL communicator (backend "nccl" on ROCm): the training-step exchange of SURVEY 8(e) --
SyncBatchNorm's packed all-reduces and the gradient buckets (all_reduce with ReduceOp.AVG, reduce-scatter + all-gather)
-- driven through the very calls an 8-GPU run issues, before a multi-GPU node ever sees them (a gpurun box has one GPU;
world-size-2 semantics are covered over gloo in test_shard_gloo.py).  Reference: scripts/train.py:176-206
(`accelerator="ddp"`, `sync_batchnorm=True`).

Process isolation: a NCCL process group brings watchdog / heartbeat threads into the process, and those threads and a later
hipGraph capture (tests/test_train_step.py) do not mix -- one full-suite run of round 3 died with a bare abort() from a
non-Python thread during the whole-step graph test, a few tests after this module had torn its group down.  So when pytest
collects this module in the main process it runs ONE test that re-invokes pytest on this file in a child process
(OCCD_RCCL_CHILD=1), where the tests below are the real ones.  `pytest tests/test_rccl_single_rank.py` keeps working."""
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"
IN_CHILD = os.environ.get("OCCD_RCCL_CHILD") == "1"

if not IN_CHILD:
    def test_rccl_single_rank_suite_in_child_process():
        env = dict(os.environ, OCCD_RCCL_CHILD="1", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-m", "gpu", "-s", "-p",
                            "no:cacheprovider"], env=env, capture_output=True, text=True, timeout=900,
                           cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        tail = "\n".join((r.stdout + r.stderr).splitlines()[-25:])
        print(tail)
        assert r.returncode == 0, tail
        assert " passed" in r.stdout and "failed" not in r.stdout.splitlines()[-1], tail




_real = pytest.mark.skipif(not IN_CHILD, reason="runs in the child process (see the module docstring)")


@pytest.fixture(scope="module")
def rccl():
    import torch.distributed as dist
    from occdepth_amd import shard
    if dist.is_initialized():
        pytest.skip("a process group already exists in this process")
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    saved = shard.FORCE_COLLECTIVES
    shard.FORCE_COLLECTIVES = True
    yield dist
    shard.FORCE_COLLECTIVES = saved
    dist.destroy_process_group()


@_real
def test_backend_is_rccl(rccl):
    assert rccl.get_backend() == "nccl" and rccl.get_world_size() == 1
    t = torch.arange(8, device=DEV, dtype=torch.float64)
    rccl.all_reduce(t)                                        # a real collective on the communicator
    assert torch.equal(t.cpu(), torch.arange(8, dtype=torch.float64))


@_real
@pytest.mark.parametrize("shape,dtype,cl", [((2, 16, 12, 10, 6), torch.float32, True), ((2, 16, 12, 10, 6), torch.float32, False),
                                            ((3, 24, 17, 33), torch.float32, False), ((2, 32, 9, 20), torch.bfloat16, True)])
def test_syncbn_on_rccl_matches_batchnorm(rccl, shape, dtype, cl):
    """forward, input / weight / bias gradients and running statistics of the converted layer (fused ATen passes + the
    float64 packed all-reduce on the device) against nn.BatchNorm in float64 on the CPU."""
    from occdepth_amd import shard
    torch.manual_seed(7)
    C = shape[1]
    Norm = torch.nn.BatchNorm3d if len(shape) == 5 else torch.nn.BatchNorm2d
    x = torch.randn(*shape) * 2.0 + 5.0
    g = torch.randn(*shape)
    ref = Norm(C).double().train()
    bn = shard.convert_sync_batchnorm(Norm(C)).to(DEV).train()
    assert isinstance(bn, shard.SyncBatchNorm)
    with torch.no_grad():
        for b in (bn, ref):
            b.weight.copy_(torch.linspace(0.5, 1.5, C))
            b.bias.copy_(torch.linspace(-1, 1, C))
    xs = x.to(DEV, dtype)
    if cl:
        xs = xs.contiguous(memory_format=torch.channels_last_3d if len(shape) == 5 else torch.channels_last)
    xs = xs.detach().requires_grad_(True)
    xr = xs.detach().cpu().double().requires_grad_(True)      # (the bf16-rounded values when dtype is bf16)
    ys = bn(xs)
    yr = ref(xr)
    assert ys.dtype == dtype
    ys.backward(g.to(DEV, dtype))
    yr.backward(g.to(dtype).double())
    tol = 3e-5 if dtype == torch.float32 else 2e-2           # bf16: the OUTPUT and gx are rounded to 8 bits
    assert float((ys.detach().cpu().double() - yr.detach()).abs().max() / yr.detach().abs().max()) < tol
    assert float((xs.grad.cpu().double() - xr.grad).abs().max() / xr.grad.abs().max()) < tol
    ptol = 3e-5 if dtype == torch.float32 else 8e-3          # bf16: ATen's fused backward-reduce rounds once to the input type
    assert float((bn.weight.grad.cpu().double() - ref.weight.grad).abs().max() / ref.weight.grad.abs().max()) < ptol
    assert float((bn.bias.grad.cpu().double() - ref.bias.grad).abs().max() / ref.bias.grad.abs().max()) < ptol
    assert float((bn.running_mean.cpu().double() - ref.running_mean).abs().max()) < ptol * 5
    assert float((bn.running_var.cpu().double() - ref.running_var).abs().max() / ref.running_var.abs().max()) < ptol
    assert int(bn.num_batches_tracked) == 1


def _toy():
    torch.manual_seed(4)
    m = torch.nn.Sequential(torch.nn.Linear(6, 64), torch.nn.ReLU(), torch.nn.Linear(64, 64), torch.nn.ReLU(),
                            torch.nn.Linear(64, 3))
    m.unused = torch.nn.Parameter(torch.ones(5))
    return m


@pytest.mark.parametrize("algo", ["all_reduce", "rs_ag"])
@_real
def test_grad_buckets_on_rccl(rccl, algo):
    """Bucketed exchange on the RCCL communicator (hooks, async launches in bucket order, AVG inside the collective,
    padded flat buffers for reduce-scatter + all-gather, no_sync accumulation, unused parameters): gradients equal the
    plain autograd ones on one rank."""
    from occdepth_amd import shard
    m, ref = _toy().to(DEV), _toy().to(DEV)
    b = shard.GradBuckets(m.parameters(), rccl, bucket_bytes=4096, algo=algo, force=True, check_used=True)
    assert b.active and b.avg_in_collective and len(b.buckets) > 2
    g = torch.Generator().manual_seed(5)
    for step in range(3):
        x, y = torch.randn(9, 6, generator=g).to(DEV), torch.randn(9, 3, generator=g).to(DEV)
        if step == 1:
            m.zero_grad(set_to_none=True)                    # detached gradients are re-adopted by the hooks
        else:
            b.zero_grad()
        ref.zero_grad(set_to_none=True)
        if step == 2:
            with b.no_sync():
                ((m(x) - y) ** 2).mean().backward()
            ((ref(x) - y) ** 2).mean().backward()
        ((m(x) - y) ** 2).mean().backward()
        ((ref(x) - y) ** 2).mean().backward()
        b.finish()
        torch.cuda.synchronize()
        assert m.unused.grad is None
        for (k, p), (_, q) in zip(m.named_parameters(), ref.named_parameters()):
            if q.grad is None:
                continue
            assert float((p.grad - q.grad).abs().max() / q.grad.abs().max().clamp_min(1e-30)) < 1e-6, (step, k)
    b.remove()


@_real
def test_prepare_for_ddp_forced_step_matches_plain_step(rccl):
    """The REAL reduced SemanticKITTI model, one training step (forward, all losses, backward) with
    prepare_for_ddp(force=True) -- every BatchNorm converted, every gradient through the buckets and RCCL -- against the
    same step of an identical un-converted model; also under bf16 autocast (statistics stay fp32)."""
    import golden_cases as gc
    from test_oracle_vs_golden import build_product
    from occdepth_amd import shard, synthetic
    outs = {}
    for forced in (False, True):
        torch.manual_seed(0)
        m, cfg, _ = build_product("kitti_small")
        m = m.to(DEV).train()
        batch = {k: ([t.to(DEV) for t in v] if isinstance(v, list) else v.to(DEV))
                 for k, v in gc.occdepth_batch("kitti_small").items()}
        synthetic.attach_training_targets(m, batch, cfg, seed=3)
        buckets = None
        if forced:
            m, buckets = shard.prepare_for_ddp(m, rccl, force=True)
            assert buckets is not None and any(isinstance(x, shard.SyncBatchNorm) for x in m.modules())
            buckets.zero_grad()
        loss = m.training_step(batch, 0)
        loss.backward()
        if buckets is not None:
            buckets.finish()
        torch.cuda.synchronize()
        outs[forced] = (float(loss.detach()), {k: p.grad.detach().clone() for k, p in m.named_parameters() if p.grad is not None})
        if buckets is not None:
            buckets.remove()
    (l0, g0), (l1, g1) = outs[False], outs[True]
    assert abs(l0 - l1) / abs(l0) < 1e-4, (l0, l1)
    assert set(g0) == set(g1)
    # one flat gradient vector per run: per-parameter ratios are meaningless on this random-init reduced network (a float32
    # round-off that lands on the other side of ONE ReLU kink moves percent-level gradient mass of small tensors, see
    # tests/test_stack3d_backward.py); the exchange itself is pinned exactly in test_grad_buckets_on_rccl
    a = torch.cat([g0[k].double().flatten() for k in sorted(g0)])
    b = torch.cat([g1[k].double().flatten() for k in sorted(g1)])
    cos = float(torch.dot(a, b) / (a.norm() * b.norm()))
    ratio = float(b.norm() / a.norm())
    print("forced-RCCL step vs plain step: loss", l0, l1, "gradient cosine", cos, "norm ratio", ratio)
    assert cos > 0.995 and abs(ratio - 1.0) < 2e-2, (cos, ratio)     # measured 0.9989 - 0.9996 (float32 kink flips between the two statistics paths)
