"""Worker of tests/test_syncbn_lightning_gpu.py: one rank of a 2-process gloo group in which BOTH ranks use the box's single GPU.

What the reference's UNMODIFIED multi-GPU entry does to the model (scripts/train.py:175-208 -> pytorch-lightning 1.4.9's
DDPPlugin.configure_ddp): `torch.nn.SyncBatchNorm.convert_sync_batchnorm(model)`, then
`DistributedDataParallel(LightningDistributedModule(model), device_ids=[local], find_unused_parameters=True)`, whose forward
calls `model.training_step(batch, batch_idx)` -- reproduced here line by line on the reduced SemanticKITTI model, one frame
per rank.  Compared on the same GPU, same kernels, fp32:
  (a) against this repo's own converter + gradient buckets (`shard.prepare_for_ddp`, pinned against a batch-of-two in
      float64 by tests/test_shard_gloo.py): the two routes must give the same loss, gradients and running statistics;
  (b) against ONE process running both frames as a batch of two (BatchNorm statistics over the batch, mean of the per-frame
      losses) -- what synchronised statistics + gradient averaging compute;
  (c) the defect of round 5 reproduced on purpose (torch's SyncBatchNorm treated as a per-rank BatchNorm): it must be FAR
      from (b), i.e. the comparison can tell.
Also driven: a set-up failure injected on ONE rank (both ranks must fall back to the process group together and still be
right), and the NaN poisoning of an exchange whose peer never arrives.  Prints one JSON line."""
import json
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

PICK = ("net_3d_decoder.ssc_head.conv0.weight", "net_3d_decoder.process_l1.0.main.0.bn2.weight",
        "net_3d_decoder.CP_mega_voxels.resize.0.weight", "net_rgb.decoder.up4._net.0.weight",
        "net_rgb.decoder.up4._net.1.bias", "net_rgb.encoder.original_model.blocks.1.0.bn1.weight",
        "net_rgb.encoder.original_model.blocks.5.0.bn2.weight", "flosp_depth.depth_net.0.depth_conv.1.bn1.weight")
STATS = ("net_3d_decoder.ssc_head.bn1.0.running_var", "net_rgb.decoder.up8._net.1.running_mean",
         "net_rgb.encoder.original_model.blocks.5.0.bn2.running_var")


class LightningDistributedModule(torch.nn.Module):
    """pytorch_lightning/overrides/base.py (1.4.9) `_LightningModuleWrapperBase.forward`, training branch."""

    def __init__(self, pl_module):
        super().__init__()
        self.module = pl_module

    def forward(self, *inputs, **kwargs):
        return self.module.training_step(*inputs, **kwargs)


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from occdepth_amd import bn as _bn
    from occdepth_amd import hip, shard
    from test_shard_gloo import _collate, _real_frame, _real_model
    hip.load()
    res = {"rank": rank, "world": world}

    def say(msg):
        print(f"[rank {rank}] {msg}", flush=True)
    say("process group up, library loaded")

    def model():
        m, cfg = _real_model()                      # seeded: the same weights on every rank and for every route
        return m.float().to(dev).train(), cfg

    def to_dev(b):
        return {k: ([t.to(dev) if torch.is_tensor(t) else t for t in v] if isinstance(v, (list, tuple))
                    else (v.to(dev) if torch.is_tensor(v) else v)) for k, v in b.items()}

    def collect(m, loss):
        named = dict(m.named_parameters())
        sd = m.state_dict()
        return {"loss": float(loss.detach()), "grads": {k: named[k].grad.detach().double().cpu() for k in PICK},
                "none": sorted(k for k, p in named.items() if p.grad is None),
                "stats": {k: sd[k].double().cpu() for k in STATS}}

    def lightning_route(expect_ipc):
        m, cfg = model()
        frame = to_dev({k: (v.float() if torch.is_tensor(v) and v.is_floating_point() else v)
                        for k, v in _real_frame(rank, m, cfg).items()})
        m = torch.nn.SyncBatchNorm.convert_sync_batchnorm(m)             # Trainer(sync_batchnorm=True)
        n_sync = sum(isinstance(x, torch.nn.SyncBatchNorm) for x in m.modules())
        ddp = torch.nn.parallel.DistributedDataParallel(LightningDistributedModule(m), device_ids=[0],
                                                        find_unused_parameters=True)
        with hip.profile() as prof:
            loss = ddp(frame, 0)
            loss.backward()
            torch.cuda.synchronize()
        m.on_train_batch_end(None, frame, 0, 0)
        tags = sorted({k.split(":")[0] for k in prof.rows if k.startswith(("bn_", "ipc_"))})
        assert (None in shard._SMALL) == expect_ipc, (expect_ipc, list(shard._SMALL))
        if None in shard._SMALL:
            shard._SMALL[None].check()
        out = collect(m, loss)
        out["n_sync"], out["tags"] = n_sync, tags
        del ddp
        return out

    # ---- (1) Lightning's route, the peer-memory exchange installed lazily by the first synchronised layer
    got = lightning_route(expect_ipc=True)
    res["n_sync_modules"], res["tags"] = got["n_sync"], got["tags"]
    say("(1) lightning route done")

    # ---- (2) this repo's converter + gradient buckets, same GPU, same frames
    m2, cfg = model()
    frame2 = to_dev({k: (v.float() if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in _real_frame(rank, m2, cfg).items()})
    m2, buckets = shard.prepare_for_ddp(m2, dist, bucket_bytes=8 << 20)
    loss2 = m2.training_step(frame2, 0)
    loss2.backward()
    buckets.finish()
    torch.cuda.synchronize()
    own = collect(m2, loss2)
    buckets.remove()
    say("(2) own route done")

    # ---- (3) one process, both frames as a batch of two (plain BatchNorm over the batch)
    m3, cfg = model()
    frames = [to_dev({k: (v.float() if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in _real_frame(i, m3, cfg).items()})
              for i in range(world)]
    out = m3(_collate(frames))
    real_forward, total, per_frame = m3.forward, 0, []
    from occdepth_amd.loss.sscMetrics import SSCMetrics
    for i in range(world):
        m3.forward = lambda b, i=i: {k: (v[i:i + 1] if torch.is_tensor(v) else v) for k, v in out.items()}
        li = m3.step(frames[i], "train", SSCMetrics(cfg.n_classes))
        per_frame.append(float(li.detach()))
        total = total + li / world
    m3.forward = real_forward
    total.backward()
    torch.cuda.synchronize()
    want = collect(m3, total)
    say("(3) batch of two done")

    def rel(a, b):
        return float((a - b).norm() / b.norm().clamp_min(1e-30))

    res["loss_vs_batch2"] = abs(got["loss"] - per_frame[rank]) / abs(per_frame[rank])
    res["loss_vs_own_route"] = abs(got["loss"] - own["loss"]) / abs(own["loss"])
    res["grad_vs_own_route"] = max(rel(got["grads"][k], own["grads"][k]) for k in PICK)
    res["grad_vs_batch2"] = {k: rel(got["grads"][k], want["grads"][k]) for k in PICK}
    res["stats_vs_batch2"] = max(rel(got["stats"][k], want["stats"][k]) for k in STATS)
    res["stats_vs_own_route"] = max(rel(got["stats"][k], own["stats"][k]) for k in STATS)
    res["none_keys_equal"] = got["none"] == want["none"] == own["none"]

    # ---- (4) round 5's defect on purpose: torch's SyncBatchNorm not recognised -> per-rank statistics
    real_is_sync = _bn.is_sync
    _bn.is_sync = lambda mod: isinstance(mod, shard.SyncBatchNorm)
    try:
        bad = lightning_route(expect_ipc=True)
    finally:
        _bn.is_sync = real_is_sync
    res["defect_stats_vs_batch2"] = max(rel(bad["stats"][k], want["stats"][k]) for k in STATS)
    res["defect_grad_vs_batch2"] = max(rel(bad["grads"][k], want["grads"][k]) for k in PICK)
    say("(4) defect route done")

    # ---- (5) a set-up failure on ONE rank: every rank falls back to the process group together, results unchanged
    shard.uninstall_small_all_reduce()
    dist.barrier()
    shard.SmallAllReduce._inject_failure = ("open", 1)
    import warnings
    with warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter("always")
        fb = lightning_route(expect_ipc=False)
    shard.SmallAllReduce._inject_failure = None
    res["fallback_warned"] = any("peer-memory all-reduce unavailable" in str(w.message) for w in caught)
    res["fallback_grad_vs_ipc"] = max(rel(fb["grads"][k], got["grads"][k]) for k in PICK)
    res["fallback_tags"] = fb["tags"]
    shard._SMALL_TRIED.discard(None)
    say("(5) fall-back route done")

    # ---- (6) a peer that never arrives: the result is NaN and the status raises (rank 0 calls alone, 300 ms budget)
    sm = shard.install_small_all_reduce(dist, timeout_ms=300)
    t = torch.ones(33, dtype=torch.float64, device=dev)
    if rank == 0:
        sm.all_reduce_(t)
        torch.cuda.synchronize()
        res["timeout_all_nan"] = bool(torch.isnan(t).all())
        try:
            sm.check()
            res["timeout_raised"] = False
        except RuntimeError:
            res["timeout_raised"] = True
    dist.barrier()
    if rank == 1:                                    # bring the sequence numbers back in step (rank 0's push is still there)
        sm.all_reduce_(t)
        torch.cuda.synchronize()
        res["late_peer_sum"] = float(t[0])
        sm.status.zero_()
    dist.barrier()
    # poll(): no synchronisation, raises at the call after the one that queued the copy of a set flag
    sm.status.fill_(1)
    sm.poll()
    torch.cuda.synchronize()
    try:
        sm.poll()
        res["poll_raised"] = False
    except RuntimeError:
        res["poll_raised"] = True
    say("(6) timeout / poll done")
    shard.uninstall_small_all_reduce()
    dist.barrier()
    dist.destroy_process_group()
    print("SYNCBN_RESULT " + json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
