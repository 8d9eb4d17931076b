"""A/B of the full-resolution head convolution (32 -> 32, 3x3x3 @ 256x256x32, dilation 1 / 2 / 3) in isolation (dev tool; GPU):
  K2s   conv3d_c32_slide_kernel      exact fp32 MFMA (v_mfma_f32_32x32x2_f32)
  K2s3  conv3d_c32_slide_x3_kernel   3-way bf16 split in the sliding window (six v_mfma_f32_32x32x16_bf16 per K step)
  K2b3  conv3d_bf16_kernel<SPLIT=3>  the same split in the generic skeleton (round 3's experiment; forced by a tile hint)
with no / one / two residual operands, plus the error of each against ATen float64 on a 16-plane slab of the same data.

    python tools/bench_head_x3.py
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F

from occdepth_amd import hip
from bench_kernels import time_many

DIMS = (256, 256, 32)
GFLOP = 2.0 * DIMS[0] * DIMS[1] * DIMS[2] * 27 * 32 * 32 / 1e9


def main():
    torch.manual_seed(0)
    x = hip.Vox(torch.randn(1, *DIMS, 32, device="cuda"), 32)
    r1 = hip.Vox(torch.randn(1, *DIMS, 32, device="cuda"), 32)
    r2 = hip.Vox(torch.randn(1, *DIMS, 32, device="cuda"), 32)
    out = hip.Vox.empty(1, DIMS, 32, "cuda")
    w = torch.randn(32, 32, 3, 3, 3, device="cuda") / (32 * 27) ** 0.5
    bias = torch.randn(32, device="cuda")
    w32, w3 = hip.pack_weights(w), hip.pack_weights_bf16(w, split3=True)
    for d in (1, 2, 3):
        geo = dict(dilation=(d,) * 3, padding=(d,) * 3)
        for nres, kw in ((0, {}), (1, dict(res1=r1, act_out=hip.ACT_RELU)), (2, dict(res1=r1, res2=r2, act_in=hip.ACT_RELU,
                                                                                       act_out=hip.ACT_RELU))):
            fns = {
                "K2s  fp32": lambda: hip.conv3d(x, w32, bias, 32, (3, 3, 3), out, **geo, **kw),
                "K2s3 bf16x3 slide": lambda: hip.conv3d_bf16(x, w3, bias, 32, (3, 3, 3), out, split3=True, **geo, **kw),
            }
            if nres == 0:
                fns["K2b3 bf16x3 generic"] = lambda: hip.conv3d_bf16(x, w3, bias, 32, (3, 3, 3), out, split3=True, tile_hint=2,
                                                                      **geo, **kw)
            ms = time_many(fns, rounds=5, iters=6)
            for name, t in ms.items():
                issued = 6.0 if "x3" in name else 1.0
                peak = 2500.0 if "x3" in name else 157.3
                print(f"head d={d} nres={nres} {name:20s}: {t:7.4f} ms  {GFLOP / t:7.1f} TF/s algorithmic  "
                      f"{issued * GFLOP / t:7.1f} TF/s issued = {issued * GFLOP / t / peak * 100:5.1f} % of the {peak:.1f} TF/s pipe peak",
                      flush=True)
    # accuracy on a 16-plane slab of the same data (float64 on the CPU)
    xs = hip.Vox(x.buf[:, :16].contiguous(), 32)
    ref_in = xs.ncdhw().cpu().double()
    for d in (1, 2, 3):
        ref = F.conv3d(ref_in, w.cpu().double(), bias.cpu().double(), padding=d, dilation=d)
        o = hip.Vox.empty(1, (16, 256, 32), 32, "cuda")
        for name, fn in (("K2s  fp32", lambda: hip.conv3d(xs, w32, bias, 32, (3, 3, 3), o, dilation=(d,) * 3, padding=(d,) * 3)),
                         ("K2s3 bf16x3 slide", lambda: hip.conv3d_bf16(xs, w3, bias, 32, (3, 3, 3), o, dilation=(d,) * 3,
                                                                        padding=(d,) * 3, split3=True))):
            fn()
            got = o.ncdhw().cpu().double()
            print(f"error d={d} {name:20s}: max |delta| / max |ref| {float((got - ref).abs().max() / ref.abs().max()):.3e}   "
                  f"rms rel {float((got - ref).norm() / ref.norm()):.3e}", flush=True)


if __name__ == "__main__":
    main()
