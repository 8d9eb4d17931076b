"""Micro-benchmarks of the HIP kernels at BASELINE config-2 shapes (dev tool; needs the GPU).

    python tools/bench_kernels.py [head] [aspp] [lift] [stack]
Variants of one shape are timed interleaved over several rounds; the median is reported."""
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn as nn

from occdepth_amd import hip


def time_many(fns, rounds=5, iters=8):
    """fns: {name: callable}; returns {name: median ms per call} with rounds interleaved."""
    for f in fns.values():
        f()
    torch.cuda.synchronize()
    res = {k: [] for k in fns}
    for _ in range(rounds):
        for k, f in fns.items():
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters):
                f()
            e1.record()
            torch.cuda.synchronize()
            res[k].append(e0.elapsed_time(e1) / iters)
    return {k: statistics.median(v) for k, v in res.items()}


def conv_case(dims, cin, cout, d, hints, k=3):
    cs = hip.round_up(cin, 8)
    buf = torch.zeros(1, *dims, cs, device="cuda")
    buf[..., :cin] = torch.randn(1, *dims, cin, device="cuda")
    x = hip.Vox(buf, cin)
    w = torch.randn(cout, cin, k, k, k, device="cuda") * 0.05
    wpk = hip.pack_weights(w)
    out = hip.Vox.empty(1, dims, cout, "cuda")
    pad = (d * (k // 2),) * 3
    fns = {h: (lambda h=h: hip.conv3d(x, wpk, None, cout, (k, k, k), out, dilation=(d,) * 3, padding=pad,
                                      tile_hint=h)) for h in hints}
    ms = time_many(fns)
    fl = 2.0 * dims[0] * dims[1] * dims[2] * k ** 3 * cin * cout
    for h, t in ms.items():
        print(f"conv k{k} {cin}->{cout} @{dims} d={d} hint={h}: {t:.3f} ms {fl / t / 1e9:6.1f} TF/s "
              f"({fl / t / 1e9 / 157.3 * 100:.1f}%)", flush=True)


def bench_head():
    for d in (1, 2, 3):
        conv_case((256, 256, 32), 32, 32, d, [0, 1])
    conv_case((256, 256, 32), 32, 22, 1, [0])
    conv_case((256, 256, 32), 2, 20, 1, [0])


def bench_aspp():
    for d in (1, 3):
        conv_case((32, 32, 4), 256, 256, d, [3, 4, 6, 7])
    conv_case((128, 128, 16), 64, 64, 1, [0])


def bench_lift():
    from occdepth_amd import synthetic
    from occdepth_amd.models.SFA import voxel_layout
    b = synthetic.kitti_frame(seed=1)
    views = [hip.project_voxels(b["T_velo_2_cam"][0][v].double().numpy(), b["cam_k"][0][v].numpy(), (0.0, -25.6, -2.0),
                                0.4, (128, 128, 16), 1220, 370, device="cuda") for v in range(2)]
    pix = torch.stack([p for p, _ in views]).unsqueeze(0)
    fov = torch.stack([m for _, m in views]).unsqueeze(0)
    sizes = [(370, 1220), (185, 610), (93, 305), (47, 153)]
    rows = [[torch.randn(1, h, w, 64, device="cuda") for _ in range(2)] for h, w in sizes]
    n_dims, out_dims, strides = voxel_layout((256, 256, 32), 2, "kitti")
    out = hip.Vox.empty(1, out_dims, 64, "cuda")
    depth = torch.rand(1, 262144, device="cuda")
    ms = time_many({"lift": lambda: hip.lift(rows, [1, 2, 4, 8], pix, fov, n_dims, strides, out, depth_scale=depth)},
                   rounds=5, iters=20)["lift"]
    print(f"sfa_lift config2: {ms * 1e3:.1f} us -> {249.0 / ms / 1e3:.2f} TB/s of algorithmic bytes "
          f"({249.0 / ms / 1e3 / 8 * 100:.1f}% of 8 TB/s)")
    # the fused lift (projection + frustum sample in the kernel) against the table path + its frustum-sample launch
    from occdepth_amd.models.flosp_depth.flosp_depth import _grid_to_lidar
    cam_E = b["T_velo_2_cam_f64"][0].unsqueeze(0).cuda().contiguous()
    cam_k = b["cam_k"][0].unsqueeze(0).cuda().contiguous()
    dvol = torch.softmax(torch.randn(1, 2, 104, 47, 153, device="cuda"), 2).contiguous()
    g2l = _grid_to_lidar([0, -25.6, -2, 51.2, 25.6, 4.4], (128, 128, 16)).cuda()
    intr = torch.zeros(1, 2, 4, 4, device="cuda")
    intr[:, :, :3, :3] = b["cam_k"][0].float().cuda()
    intr[:, :, 3, 3] = 1
    fr = hip.Frustum(dvol, (b["T_velo_2_cam"][0].cuda().unsqueeze(0) @ g2l).contiguous(), intr[:, :, :3, :].contiguous(),
                     torch.eye(4, device="cuda").repeat(1, 2, 1, 1).contiguous(), (128, 128, 16), (370, 1220), 2.0, 54.0, True)
    fns = {"tables: flosp_sample + lift": lambda: hip.lift(rows, [1, 2, 4, 8], pix, fov, n_dims, strides, out,
                                                           depth_scale=fr.sample())}
    for mode in (0, 1, 2):
        fns[f"fused lift_proj xcd_mode={mode}"] = lambda mode=mode: hip.lift_proj(
            rows, [1, 2, 4, 8], cam_E, cam_k, (0.0, -25.6, -2.0), 0.4, (1220, 370), n_dims, strides, out, frustum=fr, xcd_mode=mode)
    for name, t in time_many(fns, rounds=5, iters=20).items():
        mb = 249.0 if name.startswith("tables") else 240.4
        print(f"{name:34s} (OCCD_LIFT_INFLIGHT={os.environ.get('OCCD_LIFT_INFLIGHT', '1')}): {t * 1e3:6.1f} us -> "
              f"{mb / t / 1e3:.2f} TB/s of {mb} MB algorithmic ({mb / t / 1e3 / 8 * 100:.1f}% of 8 TB/s)")
    x = torch.randn(2, 64, 370, 1220, device="cuda")
    ms = time_many({"t": lambda: hip.nchw_to_nhwc(x)}, rounds=3, iters=10)["t"]
    print(f"nchw_to_nhwc 2x64x370x1220: {ms * 1e3:.1f} us ({2 * x.numel() * 4 / ms / 1e9:.2f} TB/s)")


def bench_stack():
    from occdepth_amd.models.unet3d_kitti import UNet3D
    m = UNet3D(20, nn.BatchNorm3d, (256, 256, 32), 64, 2, context_prior=True, cascade_cls=True).cuda().eval()
    x = hip.Vox(torch.randn(1, 128, 128, 16, 64, device="cuda"), 64)
    with torch.no_grad():
        ms = time_many({"s": lambda: m({"x3d": x})}, rounds=3, iters=3)["s"]
        print(f"UNet3D kitti config2: {ms:.2f} ms/frame -> {1068.3 / ms:.1f} TF/s "
              f"({1068.3 / ms / 157.3 * 100:.1f}% of fp32 MFMA peak)")
        with hip.profile() as prof:
            m({"x3d": x})
        for k, v in sorted(prof.rows.items(), key=lambda kv: -kv[1]["ms"])[:24]:
            print(f"  {k:58s} n={v['launches']:3d} {v['ms']:7.3f} ms {v['flops'] / 1e9:8.2f} GF "
                  f"{v['flops'] / max(v['ms'], 1e-9) / 1e9:7.1f} TF/s")


def bench_dec2d():
    """2-D decoder 3x3 convolutions as (1, 3, 3) kernels over (X=1, Y=H, Z=W) channels-last volumes."""
    shapes = [((370, 1220), 163, 80), ((370, 1220), 80, 80), ((185, 610), 352, 160), ((93, 305), 688, 320),
              ((47, 153), 1360, 640), ((24, 77), 2784, 1280), ((24, 77), 1280, 1280)]
    for (H, W), cin, cout in shapes:
        dims = (1, H, W)
        cs = hip.round_up(cin, 8)
        buf = torch.zeros(2, *dims, cs, device="cuda")
        buf[..., :cin] = torch.randn(2, *dims, cin, device="cuda")
        x = hip.Vox(buf, cin)
        w = torch.randn(cout, cin, 1, 3, 3, device="cuda") * 0.02
        wpk = hip.pack_weights(w)
        out = hip.Vox.empty(2, dims, cout, "cuda")
        fns = {h: (lambda h=h: hip.conv3d(x, wpk, None, cout, (1, 3, 3), out, padding=(0, 1, 1), tile_hint=h))
               for h in (0, 2, 3)}
        ms = time_many(fns, rounds=3, iters=4)
        fl = 2.0 * 2 * H * W * 9 * cin * cout
        for h, t in ms.items():
            print(f"dec2d {cin}->{cout} @{H}x{W} B=2 hint={h}: {t:.3f} ms {fl / t / 1e9:6.1f} TF/s", flush=True)


def bench_wgrad():
    """Weight gradient K8 at the training step's dominant shapes."""
    shapes = [((256, 256, 32), 32, 32, 3, 1), ((256, 256, 32), 32, 32, 3, 3), ((256, 256, 32), 34, 20, 3, 1),
              ((128, 128, 16), 64, 64, 1, 1), ((32, 32, 4), 256, 256, 3, 1)]
    for dims, cin, cout, k, d in shapes:
        x = hip.Vox(torch.randn(1, *dims, hip.round_up(cin, 8), device="cuda"), cin)
        gy = hip.Vox(torch.randn(1, *dims, hip.round_up(cout, 8), device="cuda"), cout)
        pad = (d * (k // 2),) * 3
        ms = time_many({"wgrad": lambda: hip.conv3d_wgrad(x, gy, cin, cout, (k,) * 3, dilation=(d,) * 3, padding=pad)},
                       rounds=3, iters=4)["wgrad"]
        fl = 2.0 * dims[0] * dims[1] * dims[2] * k ** 3 * cin * cout
        print(f"wgrad k{k} {cin}->{cout} @{dims} d={d}: {ms:.3f} ms {fl / ms / 1e9:6.1f} TF/s "
              f"({fl / ms / 1e9 / 157.3 * 100:.1f}%)", flush=True)


def bench_loss():
    """One training step's scene-completion losses at config-2 size: the statistics path (K5 + K6) against the
    reference's formulation (a softmax per loss, 20-class loop, 64-frustum loop) written with the same torch ops."""
    import torch.nn.functional as F
    from occdepth_amd.loss import ssc_loss
    B, C, dims, NF = 1, 20, (256, 256, 32), 64
    logits = (torch.randn(B, C, *dims, device="cuda") * 2).requires_grad_(True)
    # scene-like inputs: labels in 16x16-voxel patches over mostly empty space, frustums = an 8x8 grid over (x, y)
    xs = torch.arange(dims[0], device="cuda").view(-1, 1, 1)
    ys = torch.arange(dims[1], device="cuda").view(1, -1, 1)
    zs = torch.arange(dims[2], device="cuda").view(1, 1, -1)
    target = (((xs // 16) + 3 * (ys // 16) + zs // 8) % C).expand(*dims).clone().unsqueeze(0)
    target[torch.rand(B, *dims, device="cuda") < 0.7] = 0
    target = target.to(torch.uint8)
    target[torch.rand(B, *dims, device="cuda") < 0.2] = 255
    fid = ((xs * 8 // dims[0]) * 8 + ys * 8 // dims[1] + 0 * zs).unsqueeze(0)
    fid = torch.where(zs.unsqueeze(0) < 4, torch.full_like(fid, NF + 1), fid)        # below the field of view
    masks = torch.stack([fid == f for f in range(NF)], 1)
    dists = torch.rand(B, NF, C, device="cuda")
    w = torch.rand(C, device="cuda") + 0.5

    def ours():
        logits.grad = None
        sum(ssc_loss.ssc_losses(logits, target, w, masks, dists).values()).backward()

    def reference_form():
        logits.grad = None
        t = target.long()
        loss = F.cross_entropy(logits, t, weight=w, ignore_index=255)
        mask = t != 255
        p = F.softmax(logits, 1)
        tm = t[mask]
        sem, cnt = 0, 0
        for i in range(C):                                     # ssc_loss.py:44-87
            pi = p[:, i][mask]
            ct = (tm == i).float()
            if ct.sum() > 0:
                cnt += 1
                nom = (pi * ct).sum()
                sem = sem - torch.log(nom / pi.sum()) - torch.log(nom / ct.sum()) \
                    - torch.log(((1 - pi) * (1 - ct)).sum() / (1 - ct).sum())
        loss = loss + sem / cnt
        p2 = F.softmax(logits, 1)                              # geo_scal: ssc_loss.py:17-41
        e = p2[:, 0][mask]
        nt = (tm != 0).float()
        inter = (nt * (1 - e)).sum()
        loss = loss - torch.log(inter / (1 - e).sum()) - torch.log(inter / nt.sum()) \
            - torch.log(((1 - nt) * e).sum() / (1 - nt).sum())
        p3 = F.softmax(logits, 1)                              # frustums: OccDepth.py:487-521
        cntf = dists.sum(0)
        fl, ne = 0, 0
        for f in range(NF):
            prob = (masks[:, f].unsqueeze(1).float() * p3).reshape(B, C, -1).permute(1, 0, 2).reshape(C, -1)
            cum = prob.sum(1)
            tp = prob.sum()
            if tp > 0 and cntf[f].sum() > 0:
                tgt = cntf[f] / cntf[f].sum()
                fl = fl + (tgt * (torch.log(tgt) - torch.log(cum / tp))).sum()
                ne += 1
        (loss + fl / ne).backward()

    ms = time_many({"hip statistics path (fwd+bwd)": ours, "reference formulation in torch (fwd+bwd)": reference_form},
                   rounds=3, iters=2)
    for k, t in ms.items():
        print(f"loss step @(1,20,256,256,32), 64 frustums: {k}: {t:.3f} ms", flush=True)
    with hip.profile() as prof:
        ours()
    for k, v in prof.rows.items():
        print(f"  {k:40s} n={v['launches']:3d} {v['ms']:7.3f} ms  {v['bytes'] / max(v['ms'], 1e-9) / 1e6:7.1f} GB/s")


if __name__ == "__main__":
    torch.manual_seed(0)
    what = sys.argv[1:] or ["head", "aspp", "lift", "stack"]
    for w in what:
        {"head": bench_head, "aspp": bench_aspp, "lift": bench_lift, "stack": bench_stack, "dec2d": bench_dec2d, "loss": bench_loss, "wgrad": bench_wgrad}[w]()
