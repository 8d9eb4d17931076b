"""Worker of tests/test_ipc_allreduce_gpu.py: one rank of a 2-process group (gloo rendezvous on 127.0.0.1) in which BOTH ranks
use the box's single GPU -- RCCL refuses duplicate devices, IPC handles do not, so this is how the peer-memory all-reduce
(occdepth_amd.shard.SmallAllReduce, csrc/ipc_allreduce.hip) is driven on more than one process with one MI355X.  Prints one JSON
line with what it measured; any failed check exits non-zero."""
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from occdepth_amd import hip, shard
    hip.load()
    sm = shard.install_small_all_reduce(dist, max_bytes=64 * 1024, timeout_ms=8000)
    if world == 1:
        shard.FORCE_COLLECTIVES = True               # a single rank still runs every exchange (against itself)
    res = {"rank": rank, "world": world}

    def vec(n, dtype, r, k):
        g = torch.Generator().manual_seed(1000 * k + r)
        return torch.randn(n, generator=g, dtype=torch.float64).to(dtype)

    # (1) a series of exchanges of every size class and both dtypes; expected = the sum over ranks in rank order, in the dtype
    worst = 0.0
    for k, (n, dtype) in enumerate([(1, torch.float32), (7, torch.float64), (257, torch.float32), (2 * 384 + 1, torch.float64),
                                    (2 * 3840 + 1, torch.float64), (16384, torch.float32), (8192, torch.float64), (3, torch.float32)] * 3):
        t = vec(n, dtype, rank, k).to(dev)
        sm.all_reduce_(t)
        want = vec(n, dtype, 0, k).clone()
        for r in range(1, world):
            want += vec(n, dtype, r, k)
        got = t.cpu()
        if not torch.equal(got, want):
            worst = max(worst, float((got.double() - want.double()).abs().max()))
    sm.check()
    res["series_max_abs_diff"] = worst

    # (2) the same exchange captured in a hipGraph and replayed: the sequence number lives on the device
    static = torch.zeros(513, dtype=torch.float64, device=dev)
    src = torch.zeros_like(static)
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        static.copy_(src)
        sm.all_reduce_(static)                       # warm-up on the side stream (advances the sequence on every rank alike)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        static.copy_(src)
        sm.all_reduce_(static)
    ok = True
    for it in range(4):
        src.copy_(vec(513, torch.float64, rank, 50 + it).to(dev))
        g.replay()
        torch.cuda.synchronize()
        want = sum(vec(513, torch.float64, r, 50 + it) for r in range(world))
        ok = ok and torch.equal(static.cpu(), want)
    sm.check()
    res["graph_replays_exact"] = bool(ok)

    # (3) SyncBatchNorm (K13 passes) forward + backward through the peer-memory exchange == through gloo's all_reduce:
    #     a layer that takes the separate passes + the vector exchange (48 channels), a SMALL layer whose exchange happens inside
    #     its single launch (96 channels x 360 pixels: occd_bn_*_small_xchg), and a large one (separate passes)
    from occdepth_amd import bn as _bn
    res["syncbn_used_kernels"] = bool(_bn.ENABLED)
    res["syncbn_max_diff"] = 0.0
    res["syncbn_tags"] = {}
    for ci, (C, H, W) in enumerate([(48, 9, 20), (96, 9, 20), (96, 80, 80)]):
        torch.manual_seed(5 + ci)
        layer = shard.SyncBatchNorm(C).to(dev).train()
        with torch.no_grad():
            layer.weight.uniform_(0.5, 1.5)
            layer.bias.normal_()
        gx = torch.Generator().manual_seed(77 + rank + 10 * ci)
        x0 = (torch.randn(2, C, H, W, generator=gx) * (1.0 + rank) + 0.3 * rank).to(dev)
        gy = torch.randn(2, C, H, W, generator=gx).to(dev)
        outs = {}
        for mode in ("ipc", "gloo"):
            if mode == "gloo":
                shard._SMALL.pop(None)               # (keep `sm` alive: re-installed below)
            lay = shard.SyncBatchNorm(C).to(dev).train()
            lay.load_state_dict(layer.state_dict())
            x = x0.clone().requires_grad_(True)
            with hip.profile() as prof:
                y = torch.relu(lay(x))
                y.backward(gy)
                torch.cuda.synchronize()
            outs[mode] = (y.detach().cpu(), x.grad.cpu(), lay.weight.grad.cpu(), lay.bias.grad.cpu(), lay.running_mean.cpu(),
                          lay.running_var.cpu())
            if mode == "gloo":
                shard._SMALL[None] = sm
            else:
                res["syncbn_tags"][f"{C}x{H}x{W}"] = sorted({k.split(":")[0] for k in prof.rows if k.startswith(("bn_", "ipc_"))})
        d = max(float((a.double() - b.double()).abs().max() / (b.double().abs().max() + 1e-30)) for a, b in zip(outs["ipc"], outs["gloo"]))
        res["syncbn_max_diff"] = max(res["syncbn_max_diff"], d)
    sm.check()

    # (4) latency of one exchange (2C + 1 doubles, C = 384), stream-timed over 200 back-to-back calls
    t = torch.zeros(2 * 384 + 1, dtype=torch.float64, device=dev)
    for _ in range(20):
        sm.all_reduce_(t)
    dist.barrier()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200):
        sm.all_reduce_(t)
    e1.record()
    torch.cuda.synchronize()
    sm.check()
    res["us_per_exchange"] = 1e3 * e0.elapsed_time(e1) / 200
    shard.uninstall_small_all_reduce()
    dist.barrier()
    dist.destroy_process_group()
    print("IPC_RESULT " + json.dumps(res), flush=True)
    bad = worst != 0.0 or not ok or res["syncbn_max_diff"] > 2e-6
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
