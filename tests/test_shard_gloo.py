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


def _train_worker(rank, world, port, q, comm_dtype=None):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    m = _toy()
    buckets = shard.GradBuckets(m.parameters(), dist, bucket_bytes=600, comm_dtype=comm_dtype)      # several small buckets
    out = []
    early = []
    for step in range(3):                                                    # buckets are reusable
        if step == 0:
            buckets.zero_grad()                                              # gradients stay views of the flat buffers
        else:
            m.zero_grad(set_to_none=True)                                    # detached: the hook re-adopts them
        x, y = _toy_batch(rank + 10 * step)
        if step == 2:                                                        # gradient accumulation: 2 micro-steps
            with buckets.no_sync():
                ((m(x) - y) ** 2).mean().backward()
        ((m(x) - y) ** 2).mean().backward()
        early.append(buckets._next)                                          # buckets whose collective the hooks launched
        buckets.finish()
        assert m.unused.grad is None                                         # as DDP leaves a globally unused parameter
        out.append({k: p.grad.numpy().copy() for k, p in m.named_parameters() if p.grad is not None})   # by value
        flat_ptrs = {b["flat"].data_ptr() for b in buckets.buckets}
        assert all(any(fp <= p.grad.data_ptr() < fp + 4 * b["flat"].numel() for fp, b in
                       zip(sorted(flat_ptrs), sorted(buckets.buckets, key=lambda b: b["flat"].data_ptr())))
                   for p in m.parameters() if p.grad is not None)            # zero-copy: grads live in the buckets
    hist = torch.full((3, 3), rank + 1, dtype=torch.int64)
    shard.allreduce_confusion(hist, dist)
    # a gradient for a parameter outside the learned set AFTER its bucket went out must not be dropped silently
    late_error = None
    m.zero_grad(set_to_none=True)
    x, y = _toy_batch(rank + 50)
    try:
        # (scaling the INPUT: this gradient is the last one of the backward, long after the bucket's other gradients)
        ((m(x * m.unused.mean()) - y) ** 2).mean().backward()
        buckets.finish()
    except RuntimeError as e:
        late_error = str(e)
    q.put((rank, out, hist.numpy().copy(), len(buckets.buckets), early, late_error))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("comm", [None, "bf16"])
def test_two_rank_gradient_buckets_average_like_one_process(comm):
    """comm = "bf16": the buckets travel as bfloat16 (half the bytes per xGMI link), the gradients stay float32."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    cd = torch.bfloat16 if comm == "bf16" else None
    procs = [ctx.Process(target=_train_worker, args=(r, world, port, q, cd)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict()
    for _ in range(world):
        rank, out, hist, nb, early, late_error = q.get(timeout=120)
        got[rank] = (out, hist, nb)
        # overlap despite the never-touched parameter: step 0 learns which parameters receive gradients
        # -- the bucket holding the unused one (and, buckets launching in order, every bucket behind it) waits for finish() --;
        # from step 1 on EVERY bucket is launched from the autograd hooks, i.e. while the backward is still running
        assert early[0] < nb and early[1] == nb and early[2] == nb, (early, nb)
        assert late_error is not None and "outside the learned set" in late_error
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert got[0][2] > 2                                                     # the toy really spans several buckets
    assert (got[0][1] == 3).all()
    tol = dict(atol=1e-6) if comm is None else dict(rtol=1e-2, atol=2e-3)    # bf16 on the wire: 2^-9 relative per rank's term
    for step in range(3):
        want = None
        for r in range(world):                                               # single-process mean of the per-rank grads
            m = _toy()
            x, y = _toy_batch(r + 10 * step)
            ((m(x) - y) ** 2).mean().backward()
            k_acc = 2.0 if step == 2 else 1.0                                # step 2 accumulated the same micro-batch twice
            g = {k: k_acc * p.grad for k, p in m.named_parameters() if p.grad is not None}
            want = g if want is None else {k: want[k] + g[k] for k in g}
        for r in range(world):
            assert set(got[r][0][step]) == set(want) and "unused" not in want
            for k, v in got[r][0][step].items():
                assert torch.allclose(torch.from_numpy(v), want[k] / world, **tol), (step, r, k)


# ---------------------------------------------------------------------------------------------------------------
# The REAL model: reduced SemanticKITTI config, training mode, one frame per rank, shard.prepare_for_ddp
# (SyncBatchNorm + gradient buckets) == one process running both frames as a batch of 2 (BatchNorm statistics over
# the batch, loss = mean of the per-frame losses -- what DDP's gradient averaging computes).
# Reference call sites: scripts/train.py:176-206 (accelerator="ddp", sync_batchnorm=True), models/OccDepth.py:378-537.
REAL_CFG = "kitti_small"


def _real_model():
    from test_oracle_vs_golden import build_product, gold
    m, cfg, _ = build_product(REAL_CFG)
    g = gold("train_step_small")
    over = {f[len(REAL_CFG) + 10:]: torch.from_numpy(g[f]) for f in g.files if f.startswith(REAL_CFG + ".override.")}
    m.load_state_dict(over, strict=False)              # down-scaled classifier convolutions: logits O(1)
    # float64: the network is piecewise linear and a float32 forward that lands on the other side of one ReLU kink
    # moves percent-level gradient mass (tests/test_stack3d_backward.py); in float64 the two formulations agree to
    # ~1e-12 before any kink matters, so this wiring test can be tight
    return m.double().train(), cfg


def _real_frame(i, m, cfg):
    """Frame i as a batch of one: frame-specific image, training targets shaped after the model's outputs."""
    from oracle import inputs
    b = inputs.kitti_batch(img_hw=(96, 320), scene=(64, 64, 16), project_scale=2, seed=gc.SEED + i, scale_k=320 / 1220)
    X, Y, Z = cfg.full_scene_size
    shapes = {"P_logits": (1, cfg.n_relations, (X // 16) * (Y // 16) * (Z // 16), (X // 8) * (Y // 8) * (Z // 8)),
              "depth_pred": (1,)}
    b.update(gc.train_extras(REAL_CFG + "x" * i, shapes, tuple(cfg.full_scene_size), cfg.n_classes, (96, 320)))
    b["img"] = b["img"].double()
    return b


def _collate(frames):
    out = {}
    for k in frames[0]:
        v = [f[k] for f in frames]
        out[k] = torch.cat(v) if torch.is_tensor(v[0]) else [t for f in v for t in f]
    return out


PICK = ("net_3d_decoder.ssc_head.conv0.weight", "net_3d_decoder.process_l1.0.main.0.bn2.weight",
        "net_3d_decoder.CP_mega_voxels.resize.0.weight", "net_rgb.decoder.up4._net.0.weight",
        "net_rgb.decoder.up4._net.1.bias", "net_rgb.encoder.original_model.blocks.1.0.bn1.weight",
        "flosp_depth.depth_net.0.depth_conv.1.bn1.weight")


def _real_worker(rank, world, port, q):
    import emu
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(4)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    m, cfg = _real_model()
    m, buckets = shard.prepare_for_ddp(m, dist, bucket_bytes=8 << 20)
    assert buckets is not None and len(buckets.buckets) > 2
    assert sum(isinstance(x, shard.SyncBatchNorm) for x in m.modules()) > 100
    from occdepth_amd.loss.sscMetrics import SSCMetrics
    with emu.patched():
        loss = m.step(_real_frame(rank, m, cfg), "train", SSCMetrics(cfg.n_classes, device="cpu"))
        loss.backward()
    buckets.finish()
    named = dict(m.named_parameters())
    sd = m.state_dict()
    q.put((rank, float(loss.detach()), {k: named[k].grad.numpy().copy() for k in PICK},      # numpy: pickled by value
           sorted(k for k, p in named.items() if p.grad is None),
           {k: sd[k].numpy().copy() for k in ("net_3d_decoder.ssc_head.bn1.0.running_var",
                                              "net_rgb.decoder.up8._net.1.running_mean")}))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_real_model_syncbn_and_buckets_match_batch_of_two():
    import emu
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_real_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    # meanwhile: the single-process statement of the same step (thread count restored below: other tests' float32
    # reduction order, hence their round-off, depends on it)
    n_threads = torch.get_num_threads()
    torch.set_num_threads(max(1, min(4, n_threads)))
    m, cfg = _real_model()
    frames = [_real_frame(i, m, cfg) for i in range(world)]
    from occdepth_amd.loss.sscMetrics import SSCMetrics
    with emu.patched():
        out = m(_collate(frames))                       # BatchNorm statistics over both frames
        real_forward, total, per_frame = m.forward, 0, []
        try:
            for i in range(world):
                m.forward = lambda b, i=i: {k: (v[i:i + 1] if torch.is_tensor(v) else v) for k, v in out.items()}
                li = m.step(frames[i], "train", SSCMetrics(cfg.n_classes, device="cpu"))
                per_frame.append(float(li.detach()))
                total = total + li / world
        finally:
            m.forward = real_forward
        total.backward()
    torch.set_num_threads(n_threads)
    want = dict(m.named_parameters())
    want_sd = m.state_dict()
    got = {}
    for _ in range(world):
        r = q.get(timeout=600)
        got[r[0]] = r[1:]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    none_ref = sorted(k for k, p in want.items() if p.grad is None)
    for r in range(world):
        loss_r, grads, none_keys, stats = got[r]
        assert loss_r == pytest.approx(per_frame[r], rel=2e-6), r   # the loss statistics take float32 logits
        assert none_keys == none_ref                     # the never-executed branches stay without gradient
        grads = {k: torch.from_numpy(v) for k, v in grads.items()}
        for k in PICK:
            ref = want[k].grad
            print(r, k, float((grads[k] - ref).norm() / ref.norm()), float(grads[k].norm() / ref.norm()) - 1)
        for k in PICK:
            ref = want[k].grad
            assert float((grads[k] - ref).norm() / ref.norm()) < 1e-6, (r, k)
        for k, v in stats.items():
            assert torch.allclose(torch.from_numpy(v), want_sd[k], rtol=1e-4, atol=1e-6), (r, k)
    for k in PICK:                                       # both ranks end up with the SAME averaged gradient
        assert (got[0][1][k] == got[1][1][k]).all(), k


# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("mean,std", [(0.0, 1.0), (50.0, 3.0), (-300.0, 0.5)])
def test_syncbn_single_process_matches_batchnorm_when_mean_is_far_from_zero(mean, std):
    """the statistics must not cancel when the batch mean is far from the running mean (first steps:
    running_mean = 0) or mean / std is large.  float32 activations on the CPU path against nn.BatchNorm3d in float64."""
    torch.manual_seed(3)
    x = (torch.randn(2, 5, 6, 7, 4) * std + mean)
    ref_bn = torch.nn.BatchNorm3d(5).double().train()
    bn = shard.convert_sync_batchnorm(torch.nn.BatchNorm3d(5)).train()
    with torch.no_grad():
        for b in (bn, ref_bn):
            b.weight.copy_(torch.linspace(0.5, 1.5, 5))
            b.bias.copy_(torch.linspace(-1, 1, 5))
    xr = x.double().requires_grad_(True)
    xs = x.clone().requires_grad_(True)
    yr, ys = ref_bn(xr), bn(xs)
    g = torch.randn_like(ys)
    yr.backward(g.double())
    ys.backward(g)
    # float32 input data carries eps * |mean| / std of relative noise per element whatever the algorithm does
    tol = 2e-6 * max(1.0, abs(mean) / std)
    err = float((ys.detach().double() - yr.detach()).abs().max())
    assert err < 3 * tol, err
    assert float((xs.grad.double() - xr.grad).abs().max() / xr.grad.abs().max()) < 3 * tol
    assert float((bn.weight.grad.double() - ref_bn.weight.grad).abs().max() / ref_bn.weight.grad.abs().max()) < 3 * tol
    assert float((bn.running_var.double() - ref_bn.running_var).abs().max() / ref_bn.running_var.abs().max()) < 3 * tol
    assert float((bn.running_mean.double() - ref_bn.running_mean).abs().max()) < 3 * tol * max(1.0, abs(mean))
