"""Micro-benchmarks of the bf16-MFMA convolution kernels (K2b forward / data gradient, K8b weight gradient) next to their
fp32-MFMA counterparts at the shapes of the config-2 training step (dev tool; needs the GPU).

    python tools/bench_bf16.py [fwd] [wgrad]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from occdepth_amd import hip
from bench_kernels import time_many

BF16_PEAK = 2500.0


def vox(dims, c, dtype=torch.float32):
    cs = hip.round_up(c, 8)
    buf = torch.zeros(1, *dims, cs, device="cuda", dtype=dtype)
    buf[..., :c] = torch.randn(1, *dims, c, device="cuda").to(dtype)
    return hip.Vox(buf, c)


# (label, dims, cin, cout, kernel, dilation)
SHAPES = [
    ("head 3x3x3 d1", (256, 256, 32), 32, 32, (3, 3, 3), 1),
    ("head 3x3x3 d3", (256, 256, 32), 32, 32, (3, 3, 3), 3),
    ("l1 1x1x1 64>16", (128, 128, 16), 64, 16, (1, 1, 1), 1),
    ("l1 1x1x1 16>64", (128, 128, 16), 16, 64, (1, 1, 1), 1),
    ("aspp 256>256", (32, 32, 4), 256, 256, (3, 3, 3), 2),
    ("up 64>32 phase-like k222", (128, 128, 16), 64, 32, (2, 2, 2), 1),
    ("dec 1/1 163>80", (1, 740, 1220), 163, 80, (1, 3, 3), 1),        # both views stacked along H
    ("dec 1/1 80>80", (1, 740, 1220), 80, 80, (1, 3, 3), 1),
    ("dec 1/2 192>160", (1, 370, 610), 192, 160, (1, 3, 3), 1),
    ("dec 1/4 368>320", (1, 186, 305), 368, 320, (1, 3, 3), 1),
    ("dec 1/8 720>640", (1, 94, 153), 720, 640, (1, 3, 3), 1),
    ("dec 1/16 2784>1280", (1, 48, 77), 2784, 1280, (1, 3, 3), 1),
]


def bench_fwd():
    for label, dims, cin, cout, k, d in SHAPES:
        pad = tuple(d * (kk // 2) for kk in k)
        if k == (2, 2, 2):
            pad = (0, 0, 0)
        odims = tuple((n + 2 * p - d * (kk - 1) - 1) + 1 for n, kk, p in zip(dims, k, pad))
        w = torch.randn(cout, cin, *k, device="cuda") * 0.05
        fns = {}
        x32 = vox(dims, cin)
        o32 = hip.Vox.empty(1, odims, cout, "cuda")
        wb = hip.pack_weights_bf16(w)
        fns["bf16 mfma / fp32 store"] = lambda: hip.conv3d_bf16(x32, wb, None, cout, k, o32, dilation=(d,) * 3, padding=pad)
        xb = vox(dims, cin, torch.bfloat16)
        ob = hip.Vox.empty(1, odims, cout, "cuda", dtype=torch.bfloat16)
        fns["bf16 mfma / bf16 store"] = lambda: hip.conv3d_bf16(xb, wb, None, cout, k, ob, dilation=(d,) * 3, padding=pad)
        if cin * cout * k[0] * k[1] * k[2] <= 2784 * 320 * 9:
            wf = hip.pack_weights(w)
            fns["fp32 mfma (K2 / K2s)"] = lambda: hip.conv3d(x32, wf, None, cout, k, o32, dilation=(d,) * 3, padding=pad)
            w3 = hip.pack_weights_bf16(w, split3=True)
            fns["bf16x3 split / fp32 store"] = lambda: hip.conv3d_bf16(x32, w3, None, cout, k, o32, dilation=(d,) * 3,
                                                                        padding=pad, split3=True)
        ms = time_many(fns, rounds=3, iters=4)
        fl = 2.0 * odims[0] * odims[1] * odims[2] * k[0] * k[1] * k[2] * cin * cout
        vox_n = dims[0] * dims[1] * dims[2]
        for name, t in ms.items():
            esz = 2 if "bf16 store" in name else 4
            gb = esz * (vox_n * cin + odims[0] * odims[1] * odims[2] * cout) / 1e9
            print(f"fwd  {label:26s} {name:24s}: {t:8.3f} ms {fl / t / 1e9:7.1f} TF/s  {gb / t * 1e3:6.0f} GB/s algorithmic",
                  flush=True)


def bench_wgrad():
    for label, dims, cin, cout, k, d in SHAPES:
        if k == (1, 1, 1) or dims[2] < 16:
            continue
        pad = tuple(d * (kk // 2) for kk in k)
        if k == (2, 2, 2):
            pad = (0, 0, 0)
        odims = tuple((n + 2 * p - d * (kk - 1) - 1) + 1 for n, kk, p in zip(dims, k, pad))
        fns = {}
        x32, g32 = vox(dims, cin), vox(odims, cout)
        fns["bf16 mfma / fp32 store"] = lambda: hip.conv3d_wgrad_bf16(x32, g32, cin, cout, k, (1, 1, 1), (d,) * 3, pad)
        xb, gb_ = vox(dims, cin, torch.bfloat16), vox(odims, cout, torch.bfloat16)
        fns["bf16 mfma / bf16 store"] = lambda: hip.conv3d_wgrad_bf16(xb, gb_, cin, cout, k, (1, 1, 1), (d,) * 3, pad)
        if cin * cout <= 320 * 368:
            fns["fp32 mfma (K8)"] = lambda: hip.conv3d_wgrad(x32, g32, cin, cout, k, (1, 1, 1), (d,) * 3, pad)
        ms = time_many(fns, rounds=3, iters=3)
        fl = 2.0 * odims[0] * odims[1] * odims[2] * k[0] * k[1] * k[2] * cin * cout
        for name, t in ms.items():
            print(f"wgrad {label:26s} {name:24s}: {t:8.3f} ms {fl / t / 1e9:7.1f} TF/s", flush=True)


if __name__ == "__main__":
    what = sys.argv[1:] or ["fwd", "wgrad"]
    if "fwd" in what:
        bench_fwd()
    if "wgrad" in what:
        bench_wgrad()
