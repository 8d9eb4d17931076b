#!/usr/bin/env python
"""Headline benchmark: frames/s of OccDepth.forward, SemanticKITTI stereo 1220x370 -> 256x256x32 voxels
(BASELINE.json configs[1]: EfficientNet-B7, feature 64, FLoSP-Depth + CRP + cascade head, batch 1 per GPU).

    python bench.py --gpus N --steps K --warmup W [--train [--bf16]] [--config 5] [--no-extras]

N > 1: one rank per GPU over RCCL.  Under `torch.distributed.run` (WORLD_SIZE set) the ranks are the launcher's;
a bare `python bench.py --gpus N` re-executes itself through `torch.distributed.run` on 127.0.0.1, so the printed
`n_gpus` is always the number of ranks that really ran (asserted against --gpus).  Frames are sharded over the ranks:
the forward path has no data-path collective, so scaling is "weak" (one frame per rank per step).

Default (forward): a step = one forward of one synthetic stereo frame per rank, inputs already resident in HBM,
random-init weights of the named architecture, eval mode; float32 storage and accumulation everywhere, the matrix arithmetic
of the hot kernels (head convolutions K2s3, small-volume / transposed-convolution launches K2b, the 2-D network's GEMMs K16)
on the bf16 matrix pipe with both operands split into three bf16 terms (float32-level accuracy; OCCDEPTH_BF16X3=0
OCCDEPTH_GEMM_X3=0 = exact fp32 MFMA / library GEMMs everywhere, timed as `extras.exact_fp32`); the whole forward
is replayed from ONE hipGraph and the timed loop carries no events -- the roofline / stage numbers come from a separate
eager pass of the same model right after it.
`--config 5` (BASELINE configs[4]): UNet3D alone on a synthetic 512x512x64 grid (9268.2 GFLOP per frame).
`--train` (BASELINE configs[2]/[3]): a step = forward + losses + backward + gradient exchange (shard.prepare_for_ddp:
SyncBatchNorm + bucketed RCCL all-reduce, as the reference's DDP run) + AdamW on one frame per rank, replayed as one
hipGraph on one rank; `--bf16` = the bf16-MFMA mode (configs[3]: hand-written bf16 MFMA convolutions, fp32 master weights
and fp32 storage; `--autocast` adds torch autocast on top); OCCDEPTH_FORCE_DIST=1 drives SyncBatchNorm + buckets through a
single-rank RCCL group on one GPU.

Rank 0 prints ONE JSON line with the contract fields plus
  roofline       : dominant kernel (3x3x3 32->32 head convolution, 115.96 GFLOP per launch) timed live with HIP
                   events on the launch stream: issued bf16 MFMA flops (6 x algorithmic) against the dense bf16 peak
                   (2.5 PF) in the default mode, algorithmic flops against the fp32-MFMA peak (157.3 TF/s) in the exact
                   mode; `isolated_random_data` = the same kernel alone on N(0,1) data (the guide: low-toggle data clock higher);
  cpu_baseline   : the CPU oracle (oracle/occdepth_oracle.py, a port of the reference's PyTorch path) timed on
                   this box's host cores: 1 warm-up + 3 timed frames of the same workload, median (N=1 only);
  parity_rel_err : the SAME configuration as the timed model (whole-forward hipGraph, batched views, table-free lift, the
                   default convolution mode), run once untimed on the golden frame with the golden weights and compared
                   with the real reference's outputs (tests/golden/occdepth_kitti_a100.npz); max |delta| / max |ref| per
                   output, and `lift_kernels` = the lift launches an eager twin of that model makes;
  extras         : (N = 1, default run) config 5, the all-exact-fp32 frame and the fp32 / bf16 training step, each a short
                   child run of this script.
The oracle / golden machinery is used by the `cpu_baseline` and `parity_rel_err` legs only, as the checker.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

FP32_MFMA_PEAK_TFLOPS = 157.3          # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 64 FLOP/clk/SIMD
BF16_MFMA_PEAK_TFLOPS = 2500.0         # MI355X_MICROARCH.md: dense v_mfma_f32_32x32x16_bf16 (~2.5 PF, 16x the fp32 MFMA rate)
HEAD_CONV_TAG = ":32>32 k333 s111"   # 7 launches/frame, 2*256*256*32*27*32*32 = 115.96 GFLOP each
STACK3D_GFLOP = 1068.3                 # conv 1059.74 + CRP bmm 8.59
LIFT_MBYTES = 249.0                    # BASELINE.md / SURVEY.md 8(d) algorithmic HBM bytes of the lift
LIFT_PROJ_MBYTES = 240.4               # the fused lift (occd_lift_proj_fwd): no 8.9 MB of int64 tables / masks; 67.1 written
#                                        + 167.3 gathered + 6.0 depth volumes


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def model_flags():
    """(batch_views, graph_2d, graph_all): both views through the 2-D network as one batch; the whole forward replayed
    as ONE captured hipGraph (graph_all; graph_2d -- the 2-D network alone -- is what remains if that capture fails)."""
    bv = os.environ.get("OCCDEPTH_BATCH_VIEWS", "1") == "1"
    return (bv, bv and os.environ.get("OCCDEPTH_GRAPH_2D", "1") == "1",
            bv and os.environ.get("OCCDEPTH_GRAPH_ALL", "1") == "1")


def build_model(device, train=False):
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
    if train:
        return m.to(device).train(), cfg
    m.batch_views, m.graph_2d, m.graph_all = model_flags()
    m.clone_graph_outputs = clone_outputs()
    return m.to(device).eval(), cfg


def clone_outputs():
    """Fresh output tensors per forward (the reference's contract, `OccDepth.enable_fast_eval(clone_outputs=True)`, the
    default) or the graph's static buffers (OCCDEPTH_CLONE_OUTPUTS=0)."""
    return os.environ.get("OCCDEPTH_CLONE_OUTPUTS", "1") == "1"


def bench_batch(batch):
    """What the timed forward receives: the frame's images and calibration (float64 extrinsics included) and NO voxel->pixel
    tables -- the eval lift projects inside its kernel (`OccDepth.lift_in_kernel` "auto").  OCCDEPTH_LIFT_PROJ=0 keeps the
    tables (the table-path lift, for A/B)."""
    if os.environ.get("OCCDEPTH_LIFT_PROJ", "auto") == "0":
        return batch
    return {k: v for k, v in batch.items() if not (k.startswith("projected_pix") or k.startswith("fov_mask"))}


def parity_check(device):
    """The benched configuration against the real reference: golden weights + golden frame through a second model
    instance carrying the same flags as the timed one (untimed; rank 0 only)."""
    import contextlib
    import io
    import golden_cases as gc
    import numpy as np
    from test_oracle_vs_golden import build_product
    with contextlib.redirect_stdout(io.StringIO()):
        m, cfg, _ = build_product("kitti_a100")
    m = m.to(device).eval()
    m.batch_views, m.graph_2d, m.graph_all = model_flags()
    m.clone_graph_outputs = clone_outputs()
    from occdepth_amd import synthetic
    raw = dict(gc.occdepth_batch("kitti_a100"), T_velo_2_cam_f64=synthetic.kitti_frame(seed=gc.SEED)["T_velo_2_cam_f64"])
    batch = bench_batch({k: ([t.to(device) for t in v] if isinstance(v, list) else v.to(device)) for k, v in raw.items()})
    g = np.load(os.path.join(ROOT, "tests", "golden", "occdepth_kitti_a100.npz"))
    from occdepth_amd import hip
    with torch.no_grad():
        m(batch)                                           # capture pass
        out = m(batch)                                     # replay (what the timed loop runs)
        torch.cuda.synchronize()
        flags = (m.graph_2d, m.graph_all)
        m.graph_2d = m.graph_all = False                   # eager twin: which lift kernel does this configuration launch?
        with hip.profile() as prof:
            m(batch)
            torch.cuda.synchronize()
        m.graph_2d, m.graph_all = flags
    lift_tags = sorted({k.split(":")[0] for k in prof.rows if k.startswith(("sfa_lift", "flosp_sample"))})
    errs, extra = {}, {}
    for k, v in out.items():
        ref = torch.from_numpy(g[k])
        got = gc.subsample(v.cpu().contiguous())
        d = (got - ref).double()
        errs[k] = float(d.abs().max() / ref.abs().max())                  # max |delta| over the tensor's scale
        if k in ("ssc_logit", "occ_logit"):
            # finer-grained views of the same comparison: root-mean-square relative error, and the element-wise
            # relative error of every logit that is not near zero (|ref| >= 1 % of the tensor's scale)
            big = ref.abs() >= 0.01 * ref.abs().max()
            extra[k] = {"rms_rel": float(d.norm() / ref.double().norm()),
                        "max_elementwise_rel_where_ref_ge_1pct_of_scale": float((d.abs()[big] / ref.abs()[big].double()).max()),
                        "argmax_agreement": float((got.argmax(1) == ref.argmax(1)).double().mean())}
    return {"ssc_logit": errs["ssc_logit"], "occ_logit": errs["occ_logit"], "worst_of_all_outputs": max(errs.values()),
            "detail": extra,
            "batch_views": bool(m.batch_views), "graph_2d": bool(m.graph_2d), "graph_all": bool(m.graph_all),
            "fresh_outputs": bool(m.clone_graph_outputs), "lift_kernels": lift_tags,
            "metric": "max |delta| / max |ref| per output tensor -- the TENSOR-SCALE reading of north_star's '1e-3 rel' is the one "
                      "claimed and tested (< 1e-3 on every output); finer views (rms-relative, element-wise on logits >= 1 % of "
                      "scale, arg-max agreement) are reported under `detail`, not claimed at 1e-3",
            "reference": "tests/golden/occdepth_kitti_a100.npz (real reference, CPU fp32)", "bar": 1e-3}


def isolated_head(device, split, iters=10):
    """The dominant kernel ALONE on N(0,1) activations and weights (the frame's post-ReLU data of a
    random-init network toggle fewer bits and clock higher, so the in-frame time flatters the kernel a little): 32 -> 32 3x3x3
    @256x256x32 at dilation 1 / 2 / 3 without residual operands, stream events around `iters` back-to-back launches."""
    from occdepth_amd import hip
    dims = (256, 256, 32)
    g = torch.Generator(device="cpu").manual_seed(5)
    x = hip.Vox(torch.randn(1, *dims, 32, generator=g).to(device), 32)
    out = hip.Vox.empty(1, dims, 32, device)
    w = (torch.randn(32, 32, 3, 3, 3, generator=g) / (32 * 27) ** 0.5).to(device)
    bias = torch.randn(32, generator=g).to(device)
    wp = hip.pack_weights_bf16(w, split3=True) if split else hip.pack_weights(w)
    res = {}
    for d in (1, 2, 3):
        def run():
            if split:
                hip.conv3d_bf16(x, wp, bias, 32, (3, 3, 3), out, split3=True, dilation=(d,) * 3, padding=(d,) * 3)
            else:
                hip.conv3d(x, wp, bias, 32, (3, 3, 3), out, dilation=(d,) * 3, padding=(d,) * 3)
        for _ in range(10):                       # (clock ramp after the host-side pause in front of this pass)
            run()
        best = float("inf")
        for _ in range(3):                        # best of three rounds of `iters` back-to-back launches
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters):
                run()
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / iters)
        res[f"d{d}_ms"] = best
    mean = (3 * res["d1_ms"] + 2 * res["d2_ms"] + 2 * res["d3_ms"]) / 7       # the frame's mix of 32 -> 32 launches
    gflop = 2.0 * dims[0] * dims[1] * dims[2] * 27 * 32 * 32 / 1e9
    res.update({"mean_ms_frame_mix": mean, "algorithmic_tflops": gflop / mean,
                "frac": (6.0 * gflop / mean / BF16_MFMA_PEAK_TFLOPS) if split else gflop / mean / FP32_MFMA_PEAK_TFLOPS,
                "data": "N(0,1) activations and weights, no residual operands, best of 3 rounds of %d launches per dilation after "
                        "10 warm-up launches" % iters})
    return res


def cpu_baseline(model, cfg, batch, seed, timed_frames=3):
    """The CPU oracle on the host cores, same weights / inputs as the GPU run (reported, not a target; SURVEY 8(d):
    1 warm-up + >= 3 timed frames, median, thread count stated).  The oracle builds its own batch from the same seed
    (numpy restatement of the dataloader's vox2pix) and it must equal the GPU-projected one bit for bit.
    torch's CPU pool stops scaling long before a big box's hardware-thread count, so a short conv3d probe picks between
    min(64, os.cpu_count()) and os.cpu_count() threads; the frames run on the faster of the two and
    `cores` reports the threads actually used (the probe times are in `sample`)."""
    import copy
    import statistics
    from oracle import inputs
    from oracle import occdepth_oracle as orc
    batch_cpu = inputs.kitti_batch(seed=seed)
    for k, v in batch_cpu.items():                     # (the product batch has one extra key: T_velo_2_cam_f64)
        got = batch[k]
        same = all(torch.equal(a.cpu(), b) for a, b in zip(got, v)) if isinstance(v, list) else torch.equal(got.cpu(), v)
        if not same:
            raise RuntimeError(f"synthetic batch entry {k!r} differs between the product and the oracle")
    hw = os.cpu_count() or 1
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    enc = copy.deepcopy(model.net_rgb.encoder.original_model).cpu().eval()
    ocfg = dict(cfg)
    ocfg["flosp_depth_conf"] = model.flosp_depth_conf

    def frame(threads):
        torch.set_num_threads(threads)
        t0 = time.time()
        with torch.no_grad():
            orc.occdepth_forward(sd, ocfg, batch_cpu, enc)
        return time.time() - t0

    # thread count: torch's CPU pool stops scaling (and can collapse) long before a big box's hardware-thread count, and a full
    # frame at the wrong count costs minutes; a one-second 3-D convolution probe picks between min(64, hw) and hw instead
    def conv_probe(threads):
        torch.set_num_threads(threads)
        x = torch.randn(1, 32, 64, 64, 32)
        w = torch.randn(32, 32, 3, 3, 3)
        torch.nn.functional.conv3d(x, w, padding=1)
        t0 = time.time()
        for _ in range(3):
            torch.nn.functional.conv3d(x, w, padding=1)
        return (time.time() - t0) / 3

    first = min(hw, 64)
    probes = {first: conv_probe(first)}
    if hw != first:
        probes[hw] = conv_probe(hw)
    threads = min(probes, key=probes.get)
    log(f"cpu_baseline: conv3d probe {probes} -> {threads} threads")
    frame(threads)                                      # warm-up (allocator, oneDNN primitive caches)
    times = [frame(threads) for _ in range(timed_frames)]
    med = statistics.median(times)
    return {"value": 1.0 / med, "unit": "frames/s", "cores": threads, "kind": "port",
            "host_hw_threads": hw,
            "sample": f"{timed_frames} timed full config-2 frames through oracle/occdepth_oracle.py after 1 warm-up frame; thread "
                      f"count picked by a conv3d probe ({', '.join(f'{k} threads: {1e3 * v:.0f} ms' for k, v in probes.items())}); "
                      f"median {med:.2f} s/frame (min {min(times):.2f}, max {max(times):.2f}) on {threads} torch threads"}


def run_extras():
    """The auxiliary workloads next to the headline (numbers only the builder had seen), each as a
    short child run of this very script -- same code path as `python bench.py --config 5` / `--train [--bf16]`, its own
    process (no state shared with the headline measurement, which is finished by the time this runs).  Untimed for the
    headline; every leg reports its own ms/step.  OCCDEPTH_BENCH_EXTRAS=0 or --no-extras skips them."""
    legs = {"config5": ["--config", "5", "--steps", "5", "--warmup", "2"],
            "exact_fp32": ["--steps", "10", "--warmup", "3"],      # OCCDEPTH_BF16X3=0 OCCDEPTH_GEMM_X3=0
            "train_fp32": ["--train", "--steps", "3", "--warmup", "1"],
            "train_bf16": ["--train", "--bf16", "--steps", "3", "--warmup", "1"]}
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT",
                                                            "OCCDEPTH_FORCE_DIST")}
    out = {}
    for name, extra in legs.items():
        t0 = time.time()
        try:
            lenv, flags = env, ["--no-cpu-baseline", "--no-parity", "--no-extras"]
            if name == "exact_fp32":        # exact-fp32 MFMA convolutions + the library's fp32 GEMMs; its own golden parity check
                lenv = dict(env, OCCDEPTH_BF16X3="0", OCCDEPTH_GEMM_X3="0")
                flags = ["--no-cpu-baseline", "--no-extras"]
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--gpus", "1"] + flags + extra, env=lenv,
                               capture_output=True, text=True, timeout=420)
            line = [l for l in r.stdout.splitlines() if l.startswith('{"metric"')]
            if r.returncode != 0 or not line:
                out[name] = {"error": f"rc {r.returncode}: {r.stderr[-300:]}"}
                continue
            t = json.loads(line[-1])
            leg = {"ms_per_step": t["ms_per_step"], "value": t["value"], "unit": t["unit"], "steps": t["steps"],
                   "warmup": t["warmup"], "dtype": t["dtype"], "workload": t["config"]["workload"],
                   "wall_s_incl_startup": round(time.time() - t0, 1)}
            if name == "exact_fp32":
                pr = t.get("parity_rel_err") or {}
                leg.update({"stages_ms": t["stages_ms"], "bf16x3_split": t["config"].get("bf16x3_split"),
                            "gemm_x3": t["config"].get("gemm_x3"),
                            "head_conv": {k: t["roofline"][k] for k in ("kernel", "achieved", "peak", "frac", "avg_launch_ms")},
                            "parity_rel_err": {k: pr.get(k) for k in ("ssc_logit", "occ_logit", "worst_of_all_outputs", "error")
                                               if k in pr}})
            elif name == "config5":
                leg.update({"tflops": t["stack3d"]["tflops"], "frac_of_fp32_mfma_peak": t["stack3d"]["frac_of_fp32_mfma_peak"],
                            "gflop_per_frame": t["stack3d"]["gflop_per_frame"], "peak_hbm_GiB": t["peak_hbm_GiB"],
                            "head_conv": {k: t["roofline"][k] for k in ("kernel", "achieved", "peak", "frac", "avg_launch_ms")}})
            if name.startswith("train"):
                leg.update({"loss": t["loss"], "train_graph": t["train_graph"], "train_graph_error": t["train_graph_error"],
                            "max_mem_GiB": t["max_mem_GiB"], "parallelism": t["config"]["parallelism"],
                            "top_kernel_families_ms": dict(list(t["hip_kernel_families_ms_per_step"].items())[:6])})
                # the same step WITHOUT the hipGraph: what `trainer.fit` of the unmodified scripts/train.py runs (eager
                # training_step + backward + optimizer.step); OCCDEPTH_FAST_TRAIN=1 reaches the replayed one from there
                out[name + "_eager"] = {"ms_per_step": t.get("eager_ms_per_step"), "steps": t["steps"], "dtype": t["dtype"],
                                        "what": "training_step + backward + AdamW launched eagerly (no hipGraph), same process"}
            out[name] = leg
        except Exception as e:  # a report, never a reason to lose the measurement
            out[name] = {"error": repr(e)}
    return out


def respawn(args):
    """`python bench.py --gpus N` without a launcher: run N ranks through torch.distributed.run and relay their output
    (rank 0 prints the JSON line)."""
    have = torch.cuda.device_count()
    if have < args.gpus:
        raise SystemExit(f"bench.py --gpus {args.gpus}: only {have} GPU(s) visible")
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    raise SystemExit(subprocess.call(cmd, env=env))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--train", action="store_true", help="training step (configs[2]/[3]) instead of the forward")
    ap.add_argument("--bf16", action="store_true",
                    help="with --train (configs[3]): every convolution of the 3-D stack and of the 2-D decoder -- forward, data "
                         "gradient, weight gradient -- on the bf16 matrix pipe (K2b / K8b: fp32 accumulate, fp32 master weights, "
                         "fp32 activation storage)")
    ap.add_argument("--autocast", action="store_true",
                    help="with --train --bf16: additionally run the step under torch.autocast(bfloat16) (bf16 activation storage "
                         "where ATen / the libraries produce it)")
    ap.add_argument("--config", type=int, default=2, choices=(2, 5),
                    help="2: BASELINE configs[1] (the headline metric); 5: BASELINE configs[4], UNet3D alone on a synthetic "
                         "512x512x64 grid (auxiliary workload: 3-D-conv MFMA tiling + HBM footprint)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity", action="store_true")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the auxiliary legs of the default run (config 5, fp32 / bf16 training step: `extras` in the JSON line)")
    args = ap.parse_args()
    if args.steps is None:
        args.steps = 5 if args.train else 20
    if args.warmup is None:
        args.warmup = 2 if args.train else 5
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        respawn(args)
    # stdout carries exactly ONE line (the JSON).  Native libraries write banners to fd 1 (RCCL prints its version
    # block at the first communicator, MIOpen its warnings): park fd 1 on stderr until the result is ready.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    try:
        _main(args, real_stdout)
    finally:
        sys.stdout.flush()
        os.dup2(real_stdout, 1)
        os.close(real_stdout)


def _setup(args):
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started {world} rank(s)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (the HIP kernels have no CPU path)")
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    dist = None
    if world > 1 or os.environ.get("OCCDEPTH_FORCE_DIST") == "1":     # (the env switch exercises the RCCL path on one GPU)
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", device_id=device, rank=rank, world_size=world)
        assert dist.get_world_size() == args.gpus
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
    return world, rank, device, dist


def _main(args, real_stdout):
    world, rank, device, dist = _setup(args)
    if args.config == 5:
        res = _config5(args, world, rank, device, dist)
    else:
        res = _train(args, world, rank, device, dist) if args.train else _forward(args, world, rank, device, dist)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:                                   # the LAST thing on stdout (RCCL prints its banner at its first collective)
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(res) + "\n").encode())


def _forward(args, world, rank, device, dist):
    from occdepth_amd import hip, shard, synthetic
    model, cfg = build_model(device)
    # one stereo frame per rank; the voxel->pixel tables come from the product's GPU projection (dataloader work,
    # done once, outside the timed region -- exactly what the reference's dataloader hands to the model)
    with torch.no_grad():
        full_batch = synthetic.attach_projection(model, synthetic.to_device(synthetic.kitti_frame(seed=rank), device))
    batch = bench_batch(full_batch)     # (the tables stay in `full_batch` for the cpu_baseline's input cross-check)

    def step():
        with torch.no_grad():
            return model(batch)

    log(f"[bench {time.strftime('%H:%M:%S')}] model built, warm-up ({args.warmup} steps; the first captures the hipGraph)")
    for _ in range(args.warmup):
        step()
    log(f"[bench {time.strftime('%H:%M:%S')}] timed loop")
    # ---- the timed region: K steps, nothing but the forward (no per-launch HIP events, no host work besides the replay)
    shard.fence(dist)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    shard.fence(dist)
    elapsed = time.perf_counter() - t0
    elapsed = shard.max_over_ranks(elapsed, dist, device)
    assert out["ssc_logit"].shape == (1, 20, 256, 256, 32)
    graph_flags = {"batch_views": bool(model.batch_views), "graph_2d": bool(model.graph_2d),
                   "graph_all": bool(model.graph_all), "fresh_outputs": bool(model.clone_graph_outputs),
                   "voxel_tables_in_batch": any(k.startswith("projected_pix") for k in batch)}
    from occdepth_amd import fused as _fused
    graph_flags["bf16x3_split"] = _fused.BF16X3 or "off"    # "head" (default), "all" (experiment), "off" (exact-fp32 MFMA convolutions)
    graph_flags["gemm_x3"] = bool(hip.GEMM_X3)              # False: the 2-D network's GEMMs on the library's fp32 kernels
    # late round 5: short-K GEMMs on the panel-stationary kernel over pre-split weights; tap planes / expand results on a 128-byte pitch
    graph_flags["gemm_x3_panel"] = bool(hip.GEMM_X3 and hip.GEMM_X3_PANEL)
    graph_flags["padded_rows"] = bool(hip.PAD_ROWS and hip.GEMM_X3)     # (only K16 / K16p results are put on the padded pitch)
    graph_flags["lift_input"] = ("batch WITHOUT projected_pix_* / fov_mask_* (the kernel projects; INTEGRATION.md section 2: the "
                                 "one-line dataset edit) -- a batch from the reference's unmodified collate_fn carries the tables and "
                                 "takes the table lift (+ an 8.9 MB H2D copy, ~0.05 ms per frame; same parity)")

    # ---- untimed diagnostic passes (the same model, the same frame), eager so that every launch can be bracketed:
    # (1) per-launch HIP events on the launch stream (occd_prof_*) -> roofline.achieved of the head convolution;
    # (2) per-stage GPU time with events on the current stream.
    saved_flags = (model.graph_2d, model.graph_all)
    model.graph_all = False
    prof_steps = max(3, min(args.steps, 10))
    step()
    torch.cuda.synchronize()
    with hip.profile() as prof:
        for _ in range(prof_steps):
            step()
        torch.cuda.synchronize()
    # (3) the same eager frames with the 2-D network launched kernel by kernel too (graph_2d off): its families for roofline_2d
    saved_2d = model.graph_2d
    model.graph_2d = False
    step()
    torch.cuda.synchronize()
    with hip.profile() as prof2d:
        for _ in range(3):
            step()
        torch.cuda.synchronize()
    model.graph_2d = saved_2d
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
    model.graph_2d, model.graph_all = saved_flags
    if rank != 0:
        return None

    fps = world * args.steps / elapsed
    head = [(k, v) for k, v in prof.rows.items() if HEAD_CONV_TAG in k and k.startswith("conv3d")]
    n_launch = sum(v["launches"] for _, v in head)
    ms = sum(v["ms"] for _, v in head)
    flops = sum(v["flops"] for _, v in head)
    ach = flops / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
    conv_ms = sum(v["ms"] for k, v in prof.rows.items() if k.startswith(("conv3d_", "bottleneck3d", "rows_gemm"))) / prof_steps
    lift_ms = sum(v["ms"] for k, v in prof.rows.items() if k.startswith("sfa_lift")) / prof_steps
    lift_fused = any(k.startswith("sfa_lift_proj") for k in prof.rows)
    lift_mb = LIFT_PROJ_MBYTES if lift_fused else LIFT_MBYTES
    traffic, traffic_src = None, None
    for name, what in (("head_conv_hbm_bytes_inframe.json", "in-frame launches of this command (incl. the residual rows conv2.* "
                                                            "read): rocprofv3 --pmc FETCH_SIZE (x2, gfx950) and --pmc WRITE_SIZE, separate passes"),
                       ("head_conv_hbm_bytes.json", "ISOLATED launches without residual operands (tools/pmc_head.py): rocprofv3 "
                                                    "--pmc FETCH_SIZE (x2, gfx950) and --pmc WRITE_SIZE, separate passes")):
        tpath = os.path.join(ROOT, "profiles", name)
        if os.path.exists(tpath):
            with open(tpath) as f:
                traffic = json.load(f).get("bytes_per_launch")
            traffic_src = f"profiles/{name}: {what}"
            break
    res = {
        "metric": "frames/sec forward, SemanticKITTI stereo->256x256x32 voxels",
        "value": fps, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "BASELINE configs[1]: SemanticKITTI stereo 370x1220, tf_efficientnet_b7_ns, "
                               "feature 64, flosp_depth + CRP + cascade head, 256x256x32 voxels, batch 1/GPU",
                   "frames_per_step": world, "parallelism": f"dp{world} (frames sharded, no collective)",
                   "ranks": dist.get_world_size() if dist is not None else 1, **graph_flags},
        "roofline": {"bound": "mfma", "kernel": "conv3d_c32_slide_kernel: 3x3x3 32->32 @256x256x32 (v_mfma_f32_32x32x2_f32)",
                     "achieved": ach, "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                     "frac": ach / FP32_MFMA_PEAK_TFLOPS, "traffic": traffic, "traffic_source": traffic_src,
                     "launches": int(n_launch), "avg_launch_ms": ms / max(n_launch, 1),
                     "gflop_per_launch": flops / max(n_launch, 1) / 1e9,
                     "measured": f"HIP events on the launch stream around every launch of {prof_steps} eager frames of the same "
                                 "model right after the timed loop (the timed loop itself carries no events)"},
        "stack3d": {"ms_per_frame": conv_ms, "tflops": STACK3D_GFLOP / conv_ms if conv_ms else 0.0,
                    "frac_of_fp32_mfma_peak": STACK3D_GFLOP / conv_ms / FP32_MFMA_PEAK_TFLOPS if conv_ms else 0.0},
        "stack3d_launches_ms_per_frame": {k: [int(v["launches"] // prof_steps), round(v["ms"] / prof_steps, 3)] for k, v in
                                          sorted(((k, v) for k, v in prof.rows.items() if k.startswith(("conv3d", "bottleneck3d", "rows_gemm"))),
                                                 key=lambda kv: -kv[1]["ms"])[:24]},
        "stages_ms": stages,
        "stages_note": "eager pass with stream events around the three stages; ms_per_step is the graph-replayed frame",
        "lift": {"ms_per_frame": lift_ms, "gbps": lift_mb / lift_ms if lift_ms else 0.0,
                 "frac_of_8TBps": lift_mb / lift_ms / 8000.0 if lift_ms else 0.0, "algorithmic_mbytes": lift_mb,
                 "kernel": "lift_proj_kernel (projection + frustum sample + gather in one launch)" if lift_fused
                           else "lift_p1_kernel (tables from the batch) + flosp_sample_kernel"},
    }
    res["roofline_2d"] = roofline_2d(prof2d.rows, 3)
    split_head = any(k.startswith(("conv3d_c32x3", "conv3d_bf16x3")) for k, _ in head)
    if split_head:
        # the head convolutions run on the bf16 matrix pipe (3-way split, six bf16 MFMAs per algorithmic MAC): the roofline is
        # priced against the instruction that is issued; the fp32-equivalent figures stay next to it
        x3_slide = any(k.startswith("conv3d_c32x3") for k, _ in head)
        res["dtype"] = ("f32 storage and accumulate; matrix arithmetic as 3x bf16 split (hi + mid + lo of both operands, six "
                        "v_mfma_f32_32x32x16_bf16 per K step; error vs float64 tested within 1.5x of the exact-fp32 kernels', measured 0.8-1.1x) in the head convolutions "
                        "(K2s3), the small-volume 3x3x3 convolutions and transposed-convolution phases of the 3-D stack (K2b) and "
                        "the 2-D network's GEMMs (K16 / K16p); exact fp32 MFMA / VALU everywhere else (K2, K10, K11, K14, depthwise, lift). "
                        " OCCDEPTH_BF16X3=0 OCCDEPTH_GEMM_X3=0 restores exact fp32 everywhere")
        res["roofline"].update({
            "kernel": ("conv3d_c32_slide_x3_kernel" if x3_slide else "conv3d_bf16_kernel<SPLIT=3>") +
                      ": 3x3x3 32->32 @256x256x32 (6 x v_mfma_f32_32x32x16_bf16 per 16-channel K step)",
            "peak": BF16_MFMA_PEAK_TFLOPS, "achieved": 6.0 * ach, "frac": 6.0 * ach / BF16_MFMA_PEAK_TFLOPS,
            "algorithmic_tflops": ach, "fp32_mfma_equivalent_frac": ach / FP32_MFMA_PEAK_TFLOPS,
            "issued_gflop_per_launch": 6.0 * flops / max(n_launch, 1) / 1e9,
            "note": "achieved = ISSUED bf16 MFMA flops (6 x algorithmic) / time against the dense bf16 peak; algorithmic_tflops = "
                    "2*voxels*27*32*32 / time; fp32_mfma_equivalent_frac prices the same launch against the 157.3 TF/s exact-fp32 "
                    "instruction it replaced (> 1 is possible and is the point)"})
        res["stack3d"]["note"] = ("frac_of_fp32_mfma_peak is the fp32-EQUIVALENT rate of the whole stack (algorithmic flops / time / "
                                  "157.3 TF/s); the head's share of it runs on the bf16 pipe")
        xname = "head_conv_x3_hbm_bytes_inframe.json"
        xpath = os.path.join(ROOT, "profiles", xname)
        res["roofline"]["traffic"], res["roofline"]["traffic_source"] = None, None
        if os.path.exists(xpath):
            with open(xpath) as f:
                res["roofline"]["traffic"] = json.load(f).get("bytes_per_launch")
            res["roofline"]["traffic_source"] = (f"profiles/{xname}: in-frame launches of this command, rocprofv3 --pmc FETCH_SIZE "
                                                 "(x2, gfx950) and --pmc WRITE_SIZE, separate passes")
    if os.environ.get("OCCDEPTH_BENCH_ISOLATED", "1") == "1":     # (0 under rocprofv3: tools/summarize_trace.py counts head launches)
        try:
            res["roofline"]["isolated_random_data"] = isolated_head(device, split_head)
        except Exception as e:  # a report, never a reason to lose the measurement
            res["roofline"]["isolated_random_data"] = {"error": repr(e)}
    for attr in ("graph_2d_error", "graph_all_error"):
        if getattr(model, attr, None):
            res["config"][attr] = getattr(model, attr)
    if not args.no_parity:
        log(f"[bench {time.strftime('%H:%M:%S')}] parity check against the real-reference golden")
        try:
            res["parity_rel_err"] = parity_check(device)
        except Exception as e:  # a report, never a reason to lose the measurement
            res["parity_rel_err"] = {"error": repr(e)}
    if world == 1 and not args.no_extras and os.environ.get("OCCDEPTH_BENCH_EXTRAS", "1") == "1":
        log(f"[bench {time.strftime('%H:%M:%S')}] extras: config 5, training step fp32 / bf16 (child runs of this script)")
        del out
        torch.cuda.empty_cache()
        res["extras"] = run_extras()
    if world == 1 and not args.no_cpu_baseline:
        log(f"[bench {time.strftime('%H:%M:%S')}] CPU baseline (oracle, 1 warm-up + 3 timed frames)")
        try:
            res["cpu_baseline"] = cpu_baseline(model, cfg, full_batch, rank)
        except Exception as e:  # the baseline is a report, never a reason to lose the measurement
            res["cpu_baseline"] = {"value": None, "unit": "frames/s", "cores": os.cpu_count(), "kind": "port",
                                   "sample": f"failed: {e!r}"}
    else:
        res["cpu_baseline"] = None
    return res


def roofline_2d(rows, frames):
    """What limits the frame besides the head convolution: the 2-D network's kernel families from the SAME
    HIP-event pass as `roofline` -- launches and ms per frame, algorithmic rate (the flops / bytes every launch reports to the
    in-library profiler: 2 M N K for the GEMMs, the Winograd-domain MFMA work 2 * 16 * tiles * Cin * Cout for K10, bytes read +
    written for the memory-bound kernels) and its fraction of the pipe the family runs on.  Recomputable from
    profiles/r06_frame_per_launch.txt (same rows, same units)."""
    fams = {
        "K10_wino3x3": (("wino_conv3x3",), "mfma_f32", None),
        "K16_gemm_x3_large": (("gemm_f32x3:", "gemm_f32x3_preA"), "mfma_bf16x3", 20e9),      # launches of >= 20 GFLOP
        "K16_gemm_x3_small": (("gemm_f32x3:", "gemm_f32x3_preA"), "mfma_bf16x3", -20e9),     # the rest of the same kernels
        "K16p_gemm_x3_panel": (("gemm_f32x3_panel",), "mfma_bf16x3", None),
        "K11_pw_conv": (("pw_conv",), "mfma_f32", None),
        "K21_mbconv_project_splitk": (("gemm_f32x3_splitk",), "mfma_bf16x3", None),
        "depthwise": (("dwconv2d_nchw", "mbconv_dw"), "hbm", None),
        "K12_upconv_gather": (("upconv_gather",), "hbm", None),
        "squeeze_excite": (("se_gate", "se_fused"), "hbm", None),
    }
    out = {}
    for name, (prefixes, pipe, cut) in fams.items():
        sel = []
        for k, v in rows.items():
            if not k.startswith(prefixes):
                continue
            per_launch = v["flops"] / max(v["launches"], 1)
            if cut is not None and ((cut > 0 and per_launch < cut) or (cut < 0 and per_launch >= -cut)):
                continue
            sel.append(v)
        if not sel:
            continue
        ms = sum(v["ms"] for v in sel) / frames
        n = sum(v["launches"] for v in sel) / frames
        fl = sum(v["flops"] for v in sel) / frames
        by = sum(v["bytes"] for v in sel) / frames
        e = {"launches_per_frame": round(n, 1), "ms_per_frame": round(ms, 4)}
        if pipe == "hbm":
            gbps = by / (ms * 1e-3) / 1e9 if ms else 0.0
            e.update({"bound": "hbm", "algorithmic_GBps": round(gbps, 1), "peak": 8000.0, "frac": round(gbps / 8000.0, 4)})
        else:
            tf = fl / (ms * 1e-3) / 1e12 if ms else 0.0
            if pipe == "mfma_bf16x3":
                e.update({"bound": "mfma", "algorithmic_tflops": round(tf, 1), "issued_tflops": round(6 * tf, 1),
                          "peak": BF16_MFMA_PEAK_TFLOPS, "frac": round(6 * tf / BF16_MFMA_PEAK_TFLOPS, 4),
                          "pipe": "bf16 (six v_mfma_f32_32x32x16_bf16 per algorithmic 16-k MAC step)"})
            else:
                e.update({"bound": "mfma", "algorithmic_tflops": round(tf, 1), "peak": FP32_MFMA_PEAK_TFLOPS,
                          "frac": round(tf / FP32_MFMA_PEAK_TFLOPS, 4), "pipe": "fp32 (v_mfma_f32_32x32x2_f32)"})
        out[name] = e
    # (profiler scopes = operators: a squeeze-excite gate or a K21 product is ONE scope of two kernel launches; the kernel-launch
    #  count of a forward is in profiles/r06_steady_state_kernel_stats.csv)
    out["profiled_operators_per_frame"] = round(sum(v["launches"] for v in rows.values()) / frames, 1)
    return out


CONFIG5_GFLOP = 9268.2                 # 8718.5 conv + 549.8 CRP bmm


def _config5(args, world, rank, device, dist):
    """BASELINE configs[4]: `UNet3D(kitti)` alone, full_scene_size (512, 512, 64), project_scale 2, feature 64 ->
    x3d randn(1, 64, 256, 256, 32); CRP with N = 32768 voxels, M = 4096 mega voxels (reference: the `__main__` smoke
    shape of models/unet3d_kitti.py:129-171 scaled up).  Reported against the fp32-MFMA peak; eval mode, random-init
    weights, one frame per rank."""
    import torch.nn as nn
    from occdepth_amd import hip
    from occdepth_amd.models.unet3d_kitti import UNet3D
    from occdepth_amd import shard
    torch.manual_seed(0)
    m = UNet3D(20, nn.BatchNorm3d, (512, 512, 64), 64, 2, context_prior=True, cascade_cls=True).to(device).eval()
    x = hip.Vox.from_ncdhw(torch.randn(1, 64, 256, 256, 32, device=device))
    torch.cuda.reset_peak_memory_stats()

    def step():
        with torch.no_grad():
            return m({"x3d": x})

    for _ in range(max(args.warmup, 1)):
        out = step()
    del out
    shard.fence(dist)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    shard.fence(dist)
    elapsed = shard.max_over_ranks(time.perf_counter() - t0, dist, device)
    assert out["ssc_logit"].shape == (1, 20, 512, 512, 64)
    del out
    with hip.profile() as prof:
        step()
        torch.cuda.synchronize()
    if rank != 0:
        return None
    ms = 1e3 * elapsed / args.steps
    head = [(k, v) for k, v in prof.rows.items() if HEAD_CONV_TAG in k and k.startswith("conv3d")]
    hms, hfl, hn = sum(v["ms"] for _, v in head), sum(v["flops"] for _, v in head), sum(v["launches"] for _, v in head)
    ach = hfl / (hms * 1e-3) / 1e12 if hms > 0 else 0.0
    rows = sorted(((k, v) for k, v in prof.rows.items() if k.startswith(("conv3d", "bottleneck3d", "rows_gemm"))), key=lambda kv: -kv[1]["ms"])[:12]
    fams = sorted({k.split(":")[0] for k in prof.rows})
    x3 = [f for f in fams if f.startswith(("conv3d_c32x3", "conv3d_bf16x3"))]
    head_x3 = any(k.startswith(("conv3d_c32x3", "conv3d_bf16x3")) for k, _ in head)
    dtype = "f32" if not x3 else ("f32 storage and accumulate; 3x bf16 split (six v_mfma_f32_32x32x16_bf16 per K step, float32-level "
                                  "accuracy) in " + ", ".join(x3) + "; exact fp32 MFMA in the other launches")
    roof = {"bound": "mfma", "kernel": "head conv 3x3x3 32->32 @512x512x64 (927.7 GFLOP per launch)",
            "achieved": ach, "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ach / FP32_MFMA_PEAK_TFLOPS,
            "traffic": None, "launches": int(hn), "avg_launch_ms": hms / max(hn, 1),
            "head_kernel": sorted({k.split(":")[0] for k, _ in head})}
    by = {}
    for f in roof["head_kernel"]:
        sel = [v for k, v in head if k.split(":")[0] == f]
        fms, ffl, fn = sum(v["ms"] for v in sel), sum(v["flops"] for v in sel), sum(v["launches"] for v in sel)
        alg = ffl / (fms * 1e-3) / 1e12 if fms > 0 else 0.0
        isx3 = f.startswith(("conv3d_c32x3", "conv3d_bf16x3"))
        by[f] = {"launches": int(fn), "avg_launch_ms": fms / max(fn, 1), "algorithmic_tflops": alg,
                 "peak": BF16_MFMA_PEAK_TFLOPS if isx3 else FP32_MFMA_PEAK_TFLOPS,
                 "frac": (6.0 * alg / BF16_MFMA_PEAK_TFLOPS) if isx3 else alg / FP32_MFMA_PEAK_TFLOPS}
    roof["by_kernel"] = by
    if head_x3:       # the launches K2s3 takes are priced like the headline's (issued bf16 flops against the bf16 peak)
        f = [k for k in by if k.startswith(("conv3d_c32x3", "conv3d_bf16x3"))][0]
        roof.update({"algorithmic_tflops": by[f]["algorithmic_tflops"],
                     "fp32_mfma_equivalent_frac": by[f]["algorithmic_tflops"] / FP32_MFMA_PEAK_TFLOPS,
                     "achieved": 6.0 * by[f]["algorithmic_tflops"], "peak": BF16_MFMA_PEAK_TFLOPS, "frac": by[f]["frac"],
                     "launches": by[f]["launches"], "avg_launch_ms": by[f]["avg_launch_ms"],
                     "note": f"top-level figures = the {f} launches (issued bf16 flops, 6 x algorithmic, against the dense bf16 "
                             "peak); launches left on the exact-fp32 kernel are under by_kernel against the fp32-MFMA peak"})
    return {
        "metric": "frames/sec forward, UNet3D alone, synthetic 512x512x64 voxel grid (BASELINE configs[4])",
        "value": world * args.steps / elapsed, "unit": "frames/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": dtype, "data": "synthetic",
        "config": {"workload": "BASELINE configs[4]: UNet3D(kitti) alone, full_scene_size 512x512x64, project_scale 2, feature 64, "
                               "CRP + cascade head, x3d (1, 64, 256, 256, 32)", "frames_per_step": world},
        "stack3d": {"gflop_per_frame": CONFIG5_GFLOP, "tflops": CONFIG5_GFLOP / ms,
                    "frac_of_fp32_mfma_peak": CONFIG5_GFLOP / ms / FP32_MFMA_PEAK_TFLOPS},
        "roofline": roof,
        "peak_hbm_GiB": torch.cuda.max_memory_allocated() / 2 ** 30,
        "top_conv_launches_ms": {k: round(v["ms"], 3) for k, v in rows},
        "cpu_baseline": None,
    }


def _train(args, world, rank, device, dist):
    """BASELINE configs[2] (fp32) / configs[3] (--bf16): one frame per rank, the reference's full `training_step`
    (forward, every loss term, backward), the gradient exchange of its DDP run, AdamW."""
    from occdepth_amd import autograd3d, hip, shard, synthetic
    autograd3d.set_bf16_mfma(args.bf16)
    autocast = bool(args.bf16 and args.autocast)
    model, cfg = build_model(device, train=True)
    with torch.no_grad():
        batch = synthetic.attach_projection(model, synthetic.to_device(synthetic.kitti_frame(seed=rank), device))
    synthetic.attach_training_targets(model, batch, cfg, seed=1 + rank)
    forced = os.environ.get("OCCDEPTH_FORCE_DIST") == "1"    # one GPU: SyncBatchNorm + buckets on a single-rank RCCL group
    parts = os.environ.get("OCCDEPTH_FORCE_PARTS", "all")          # forced single-rank runs: which exchanges to drive
    # "bn": SyncBatchNorm's exchanges only (no gradient buckets); "buckets": gradient buckets only (local BatchNorm statistics)
    model, buckets = shard.prepare_for_ddp(model, dist, force=forced, grad_buckets=not (forced and parts == "bn"),
                                           sync_bn=not (forced and parts == "buckets"))
    opt = model.configure_optimizers()[0][0]
    # whole-step hipGraph: default on one rank without buckets; with buckets (collectives launched from autograd hooks
    # inside the capture) it is opt-in until it has run on a multi-GPU node: OCCDEPTH_TRAIN_GRAPH_DDP=1
    use_graph = os.environ.get("OCCDEPTH_TRAIN_GRAPH", "1") == "1" and \
        (buckets is None or os.environ.get("OCCDEPTH_TRAIN_GRAPH_DDP", "0") == "1")
    if use_graph:
        from occdepth_amd import train_graph
        train_graph.make_capturable(opt)                 # (before the optimizer's first step: device-side step counters)

    def step():
        if buckets is not None:
            buckets.zero_grad()
        else:
            opt.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=autocast):
            loss = model.training_step(batch, 0)
        loss.backward()
        if buckets is not None:
            buckets.finish()
        opt.step()
        return loss

    for _ in range(args.warmup):
        step()
    # whole-step hipGraph (occdepth_amd/train_graph.py): the eager step is ~9600 launches and host-bound.
    graphed, graph_error = None, None
    if use_graph:
        gs = train_graph.GraphedTrainStep(model, opt, batch, bf16=autocast, buckets=buckets, warmup=1)
        if gs.capture():
            graphed = gs
            gs()
        else:
            graph_error = gs.error
    shard.fence(dist)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = graphed() if graphed is not None else step()
    shard.fence(dist)
    elapsed = time.perf_counter() - t0
    elapsed = shard.max_over_ranks(elapsed, dist, device)
    loss_value = float(loss.detach())
    if None in shard._SMALL:
        shard._SMALL[None].check()                 # a peer-memory exchange that gave up waiting fails the run, loudly
    # the step a user of the UNMODIFIED scripts gets (Lightning's automatic optimisation: zero_grad, training_step, backward,
    # optimizer.step, all eager -- ~4 500 launches, host-bound), timed the same way right after the replayed one
    eager_ms = None
    if graphed is not None:
        step()
        shard.fence(dist)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        shard.fence(dist)
        eager_ms = 1e3 * shard.max_over_ranks(time.perf_counter() - t0, dist, device) / args.steps
    with hip.profile() as prof:                    # kernel table: ONE eager step after the timed loop (HIP events per launch)
        step()
        torch.cuda.synchronize()
    if rank != 0:
        return None
    rows = sorted(prof.rows.items(), key=lambda kv: -kv[1]["ms"])[:8]
    families = {}
    for k, v in prof.rows.items():
        fam = k.split(":")[0]
        families[fam] = families.get(fam, 0.0) + v["ms"]
    families = {k: round(v, 3) for k, v in sorted(families.items(), key=lambda kv: -kv[1])}
    return {
        "metric": "frames/sec trained (fwd + losses + bwd + gradient exchange + AdamW), SemanticKITTI stereo->256x256x32 voxels",
        "value": world * args.steps / elapsed, "unit": "frames/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None,
        "dtype": ("bf16 MFMA (v_mfma_f32_32x32x16_bf16, fp32 accumulate, fp32 master weights) for every convolution of the 3-D "
                  "stack and the 2-D decoder, forward + dgrad + wgrad; " +
                  ("torch.autocast(bf16) around the step" if autocast else "fp32 activation storage, EfficientNet encoder fp32"))
        if args.bf16 else "f32",
        "data": "synthetic",
        "config": {"workload": f"BASELINE configs[{3 if args.bf16 else 2}]: training step, SemanticKITTI stereo 370x1220, "
                               "tf_efficientnet_b7_ns, feature 64, flosp_depth + CRP + cascade head, batch 1/GPU",
                   "global_batch": world, "ranks": dist.get_world_size() if dist is not None else 1,
                   "parallelism": f"dp{world}: SyncBatchNorm (packed all-reduce per layer, " +
                                  ("peer-memory kernel csrc/ipc_allreduce.hip" if None in shard._SMALL else "process group") +
                                  f") + {len(buckets.buckets) if buckets else 0} gradient buckets ({buckets.algo if buckets else 'none'})"},
        "train_graph": graphed is not None, "train_graph_error": graph_error,
        "eager_ms_per_step": eager_ms if graphed is not None else 1e3 * elapsed / args.steps,
        "loss": loss_value, "max_mem_GiB": torch.cuda.max_memory_allocated() / 2 ** 30,
        "hip_kernels_ms_per_step": {k: v["ms"] for k, v in rows},
        "hip_kernel_families_ms_per_step": families,
    }


if __name__ == "__main__":
    main()
