"""Micro-benchmarks of the HIP kernels at BASELINE config-2 shapes (dev tool; needs the GPU)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn as nn
from occdepth_amd import hip


def timeit(fn, n=5, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(n):
        fn()
    t1.record(); torch.cuda.synchronize()
    return t0.elapsed_time(t1) / n


def head_conv(d, hint=0, dims=(256, 256, 32), cin=32, cout=32):
    cs = hip.round_up(cin, 8)
    buf = torch.zeros(1, *dims, cs, device="cuda")
    buf[..., :cin] = torch.randn(1, *dims, cin, device="cuda")
    x = hip.Vox(buf, cin)
    w = torch.randn(cout, cin, 3, 3, 3, device="cuda") * 0.05
    wpk = hip.pack_weights(w)
    out = hip.Vox.empty(1, dims, cout, "cuda")
    fn = lambda: hip.conv3d(x, wpk, None, cout, (3, 3, 3), out, dilation=(d,) * 3, padding=(d,) * 3, tile_hint=hint)
    ms = timeit(fn)
    fl = 2.0 * dims[0] * dims[1] * dims[2] * 27 * cin * cout
    print(f"conv3x3x3 {cin}->{cout} @{dims} d={d} hint={hint}: {ms:.3f} ms  {fl/ms/1e9:.1f} TF/s  ({fl/ms/1e9/157.3*100:.1f}% of 157.3)")


if __name__ == "__main__":
    torch.manual_seed(0)
    for d in (1, 2, 3):
        head_conv(d)
    for h in (1, 5, 6):
        for d in (1, 3):
            head_conv(d, hint=h)
    head_conv(1, cin=34, cout=20)
    head_conv(1, cin=32, cout=2)
    head_conv(1, dims=(32, 32, 4), cin=256, cout=256)
    head_conv(1, dims=(128, 128, 16), cin=64, cout=64)
    # whole 3-D stack at config 2
    from occdepth_amd.models.unet3d_kitti import UNet3D
    m = UNet3D(20, nn.BatchNorm3d, (256, 256, 32), 64, 2, context_prior=True, cascade_cls=True).cuda().eval()
    x = hip.Vox(torch.randn(1, 128, 128, 16, 64, device="cuda"), 64)
    with torch.no_grad():
        m({"x3d": x})
        ms = timeit(lambda: m({"x3d": x}), n=3, warm=1)
        print(f"UNet3D kitti config2: {ms:.2f} ms/frame -> {1068.3/ms:.1f} TF/s ({1068.3/ms/157.3*100:.1f}% of fp32 MFMA peak)")
        with hip.profile() as prof:
            m({"x3d": x})
        for k, v in sorted(prof.rows.items(), key=lambda kv: -kv[1]["ms"]):
            print(f"  {k:58s} n={v['launches']:3d} {v['ms']:7.3f} ms {v['flops']/1e9:8.2f} GF {v['flops']/max(v['ms'],1e-9)/1e9:7.1f} TF/s")
    print("peak mem GB", torch.cuda.max_memory_allocated() / 2**30)
