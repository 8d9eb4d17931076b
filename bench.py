#!/usr/bin/env python
"""Headline benchmark: frames/s of OccDepth.forward, SemanticKITTI stereo 1220x370 -> 256x256x32 voxels
(BASELINE.json configs[1]: EfficientNet-B7, feature 64, FLoSP-Depth + CRP + cascade head, batch 1 per GPU).

    python bench.py --gpus N --steps K --warmup W
    (N > 1: launched by torch.distributed.run, one rank per GPU, frames sharded over ranks -- the forward
     path has no data-path collective, so scaling is "weak": one frame per rank per step.)

A step = one forward of one synthetic stereo frame per rank, inputs already resident in HBM, random-init
weights of the named architecture, eval mode, fp32 (the 3-D stack runs on exact-fp32 MFMA).
Rank 0 prints ONE JSON line with the contract fields plus
  roofline     : dominant kernel (3x3x3 32->32 head convolution, 115.96 GFLOP per launch) timed live with HIP
                 events on the launch stream, against the fp32-MFMA peak (157.3 TF/s);
  cpu_baseline : the CPU oracle (oracle/occdepth_oracle.py, a port of the reference's PyTorch path) timed on
                 this box's host cores on ONE frame of the same workload (N=1 only).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

FP32_MFMA_PEAK_TFLOPS = 157.3          # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 64 FLOP/clk/SIMD
HEAD_CONV_TAG = ":32>32 k333 s111"   # 7 launches/frame, 2*256*256*32*27*32*32 = 115.96 GFLOP each
STACK3D_GFLOP = 1068.3                 # conv 1059.74 + CRP bmm 8.59
LIFT_MBYTES = 249.0                    # BASELINE.md / SURVEY.md 8(d) algorithmic HBM bytes of the lift


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def build_model(device):
    from occdepth_amd import configs
    from occdepth_amd.models.OccDepth import OccDepth
    cfg = configs.kitti_a100.clone()
    torch.manual_seed(0)
    import contextlib
    import io
    with contextlib.redirect_stdout(io.StringIO()):
        m = OccDepth(class_names=[str(i) for i in range(cfg.n_classes)], class_weights=torch.ones(cfg.n_classes),
                     class_weights_occ=torch.ones(2), full_scene_size=tuple(cfg.full_scene_size),
                     project_res=configs.PROJECT_RES, config=cfg)
    m.batch_views = os.environ.get("OCCDEPTH_BATCH_VIEWS", "1") == "1"
    m.graph_2d = m.batch_views and os.environ.get("OCCDEPTH_GRAPH_2D", "1") == "1"
    return m.to(device).eval(), cfg


def to_dev(batch, device):
    out = {}
    for k, v in batch.items():
        out[k] = [t.to(device) for t in v] if isinstance(v, list) else v.to(device)
    return out


def cpu_baseline(model, cfg, batch, seed, budget_s=45.0):
    """The CPU oracle on the host cores, same weights / inputs as the GPU run (reported, not a target).
    This leg is the only place bench.py touches oracle/: the oracle builds its own batch from the same seed
    (numpy restatement of the dataloader's vox2pix) and it must equal the GPU-projected one bit for bit."""
    import copy
    from oracle import inputs
    from oracle import occdepth_oracle as orc
    batch_cpu = inputs.kitti_batch(seed=seed)
    for k, v in batch_cpu.items():                     # (the product batch has one extra key: T_velo_2_cam_f64)
        got = batch[k]
        same = all(torch.equal(a.cpu(), b) for a, b in zip(got, v)) if isinstance(v, list) else torch.equal(got.cpu(), v)
        if not same:
            raise RuntimeError(f"synthetic batch entry {k!r} differs between the product and the oracle")
    threads = os.cpu_count() or 1
    torch.set_num_threads(min(threads, 64))          # torch's CPU pool stops scaling long before 256 threads
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    enc = copy.deepcopy(model.net_rgb.encoder.original_model).cpu().eval()
    ocfg = dict(cfg)
    ocfg["flosp_depth_conf"] = model.flosp_depth_conf
    times = []
    t_start = time.time()
    with torch.no_grad():
        for i in range(2):
            t0 = time.time()
            orc.occdepth_forward(sd, ocfg, batch_cpu, enc)
            times.append(time.time() - t0)
            if time.time() - t_start + times[-1] > budget_s:
                break
    best = min(times[1:]) if len(times) > 1 else times[0]
    return {"value": 1.0 / best, "unit": "frames/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{len(times)} full config-2 frame(s) through oracle/occdepth_oracle.py (first is warm-up), "
                      f"best {best:.2f} s/frame"}


def main():
    # stdout carries exactly ONE line (the JSON).  Native libraries write banners to fd 1 (RCCL prints its version
    # block at the first communicator, MIOpen its warnings): park fd 1 on stderr until the result is ready.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    try:
        _main(real_stdout)
    finally:
        sys.stdout.flush()
        os.dup2(real_stdout, 1)
        os.close(real_stdout)


def _main(real_stdout):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (the HIP kernels have no CPU path)")
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    dist = None
    if world > 1 or os.environ.get("OCCDEPTH_FORCE_DIST") == "1":     # (the env switch exercises the RCCL path on one GPU)
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", device_id=device)
    torch.backends.cudnn.allow_tf32 = False
    torch.backends.cuda.matmul.allow_tf32 = False
    # MIOpen exhaustive find costs minutes of untimed warm-up on a fresh box; opt in with OCCDEPTH_MIOPEN_FIND=1
    torch.backends.cudnn.benchmark = os.environ.get("OCCDEPTH_MIOPEN_FIND", "0") == "1"

    from occdepth_amd import build, hip
    if rank == 0:
        build.build(verbose=False)
    if dist is not None:
        dist.barrier()
    hip.load()

    from occdepth_amd import synthetic
    model, cfg = build_model(device)
    # one stereo frame per rank; the voxel->pixel tables come from the product's GPU projection (dataloader work,
    # done once, outside the timed region -- exactly what the reference's dataloader hands to the model)
    with torch.no_grad():
        batch = synthetic.attach_projection(model, synthetic.to_device(synthetic.kitti_frame(seed=rank), device))

    def step():
        with torch.no_grad():
            return model(batch)

    from occdepth_amd import shard

    def fence():
        shard.fence(dist)

    for _ in range(args.warmup):
        step()
    fence()
    with hip.profile() as prof:
        t0 = time.perf_counter()
        for _ in range(args.steps):
            out = step()
        fence()
        elapsed = time.perf_counter() - t0
    elapsed = shard.max_over_ranks(elapsed, dist, device)
    assert out["ssc_logit"].shape == (1, 20, 256, 256, 32)

    # untimed diagnostic pass: per-stage GPU time with events on the current stream
    stages = {}

    def timed(name, fn):
        def wrapper(*a, **k):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            r = fn(*a, **k)
            e1.record()
            stages[name] = (e0, e1)
            return r
        return wrapper

    orig = (model.process_rgbs, model._forward_2d_to_3d, model.net_3d_decoder.forward)
    model.process_rgbs = timed("net_rgb_2d_ms", orig[0])
    model._forward_2d_to_3d = timed("lift_2d_to_3d_ms", orig[1])
    model.net_3d_decoder.forward = timed("stack_3d_ms", orig[2])
    step()
    torch.cuda.synchronize()
    model.process_rgbs, model._forward_2d_to_3d, model.net_3d_decoder.forward = orig
    stages = {k: e0.elapsed_time(e1) for k, (e0, e1) in stages.items()}

    if rank == 0:
        fps = world * args.steps / elapsed
        head = [(k, v) for k, v in prof.rows.items() if HEAD_CONV_TAG in k and k.startswith("conv3d")]
        n_launch = sum(v["launches"] for _, v in head)
        ms = sum(v["ms"] for _, v in head)
        flops = sum(v["flops"] for _, v in head)
        ach = flops / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
        conv_ms = sum(v["ms"] for k, v in prof.rows.items() if k.startswith("conv3d_")) / args.steps
        lift_ms = sum(v["ms"] for k, v in prof.rows.items() if k.startswith("sfa_lift")) / args.steps
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "head_conv_hbm_bytes.json")
        if os.path.exists(tpath):
            with open(tpath) as f:
                traffic = json.load(f).get("bytes_per_launch")
        res = {
            "metric": "frames/sec forward, SemanticKITTI stereo->256x256x32 voxels",
            "value": fps, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: SemanticKITTI stereo 370x1220, tf_efficientnet_b7_ns, "
                                   "feature 64, flosp_depth + CRP + cascade head, 256x256x32 voxels, batch 1/GPU",
                       "frames_per_step": world, "parallelism": f"dp{world} (frames sharded, no collective)",
                       "batch_views": bool(model.batch_views), "graph_2d": bool(model.graph_2d)},
            "roofline": {"bound": "mfma", "kernel": "conv3d_c32_slide_kernel: 3x3x3 32->32 @256x256x32 (v_mfma_f32_32x32x2_f32)",
                         "achieved": ach, "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": ach / FP32_MFMA_PEAK_TFLOPS, "traffic": traffic,
                         "launches": int(n_launch), "avg_launch_ms": ms / max(n_launch, 1),
                         "gflop_per_launch": flops / max(n_launch, 1) / 1e9},
            "stack3d": {"ms_per_frame": conv_ms, "tflops": STACK3D_GFLOP / conv_ms if conv_ms else 0.0,
                        "frac_of_fp32_mfma_peak": STACK3D_GFLOP / conv_ms / FP32_MFMA_PEAK_TFLOPS if conv_ms else 0.0},
            "stages_ms": stages,
            "lift": {"ms_per_frame": lift_ms, "gbps": LIFT_MBYTES / lift_ms if lift_ms else 0.0,
                     "frac_of_8TBps": LIFT_MBYTES / lift_ms / 8000.0 if lift_ms else 0.0},
        }
        if world == 1 and not args.no_cpu_baseline:
            try:
                res["cpu_baseline"] = cpu_baseline(model, cfg, batch, rank)
            except Exception as e:  # the baseline is a report, never a reason to lose the measurement
                res["cpu_baseline"] = {"value": None, "unit": "frames/s", "cores": os.cpu_count(), "kind": "port",
                                       "sample": f"failed: {e!r}"}
        else:
            res["cpu_baseline"] = None
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(res) + "\n").encode())
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
