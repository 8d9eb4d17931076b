"""Per-layer gradient error of the bf16-MFMA training mode (autograd3d.BF16_MFMA, BASELINE configs[3]) against ATen
float64 on the UN-rounded operands, for the convolution geometries of the config-2 training step (channels and taps at
full size -- the error depends on the reduction length K = taps x cin and on nothing else --, spatial extents reduced so the
float64 CPU reference finishes in seconds).  VERDICT r2 item 1: "per-layer gradient error vs fp64 ATen reported in
profiles/ (expect ~2^-8-level, state the bound)".

    python tools/bf16_layer_errors.py > profiles/r03_bf16_layer_errors.txt        (needs the GPU)

Expected: bf16 rounding is 2^-9 relative per operand (round to nearest), i.e. 2^-8.5 ~ 2.8e-3 per product; a K-term dot
product of independent roundings gives a relative L2 error of about 2.8e-3 regardless of K (errors and signal both grow
like sqrt(K)).  Stated bound per tensor: ||g - g64|| / ||g64|| <= 6e-3 (2 x the expectation), fp32 mode <= 1e-5.
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F

from occdepth_amd import autograd3d

# (label, cin, cout, kernel, stride, padding, dilation, dims, transposed)
LAYERS = [
    ("head 3x3x3 32>32 d1 (modules.py:76)", 32, 32, (3, 3, 3), 1, 1, 1, (12, 20, 32), False),
    ("head 3x3x3 32>32 d2", 32, 32, (3, 3, 3), 1, 2, 2, (12, 20, 32), False),
    ("head 3x3x3 32>32 d3", 32, 32, (3, 3, 3), 1, 3, 3, (12, 20, 32), False),
    ("classes 3x3x3 34>20 (modules.py:118)", 34, 20, (3, 3, 3), 1, 1, 1, (10, 16, 32), False),
    ("bottleneck 1x1x1 64>16 (DDR.py:33)", 64, 16, (1, 1, 1), 1, 0, 1, (16, 16, 16), False),
    ("bottleneck (1,1,3) 16>16 d2 (DDR.py:38)", 16, 16, (1, 1, 3), 1, (0, 0, 2), (1, 1, 2), (16, 16, 16), False),
    ("bottleneck 1x1x1 16>64 (DDR.py:42)", 16, 64, (1, 1, 1), 1, 0, 1, (16, 16, 16), False),
    ("downsample 3x3x3 s2 64>128", 64, 128, (3, 3, 3), 2, 1, 1, (16, 16, 16), False),
    ("aspp 3x3x3 256>256 d2 (CRP3D.py)", 256, 256, (3, 3, 3), 1, 2, 2, (8, 8, 4), False),
    ("crp 1x1x1 512>512", 512, 512, (1, 1, 1), 1, 0, 1, (8, 8, 4), False),
    ("upsample convT 3x3x3 s2 64>32 (modules.py:192)", 64, 32, (3, 3, 3), 2, 1, 1, (8, 8, 8), True),
    ("decoder 1/1 3x3 163>80 (unet2d.py:24-46)", 163, 80, (1, 3, 3), 1, (0, 1, 1), 1, (1, 37, 61), False),
    ("decoder 1/2 3x3 352>160", 352, 160, (1, 3, 3), 1, (0, 1, 1), 1, (1, 24, 39), False),
    ("decoder 1/16 3x3 2784>1280", 2784, 1280, (1, 3, 3), 1, (0, 1, 1), 1, (1, 6, 10), False),
    ("decoder head 1x1 80>64 (unet2d.py:120-131)", 80, 64, (1, 1, 1), 1, 0, 1, (1, 37, 61), False),
]


def t3(v):
    return (v,) * 3 if isinstance(v, int) else tuple(v)


def run(layer, bf16):
    label, cin, cout, k, s, p, d, dims, transposed = layer
    g = torch.Generator().manual_seed(sum(map(ord, label)))
    x = torch.randn(1, cin, *dims, generator=g)
    fan = cin * k[0] * k[1] * k[2]
    w = torch.randn((cin, cout) + k if transposed else (cout, cin) + k, generator=g) / fan ** 0.5
    s, p, d = t3(s), t3(p), t3(d)
    xr = x.double().requires_grad_(True)
    wr = w.double().requires_grad_(True)
    if transposed:
        yr = F.conv_transpose3d(xr, wr, None, s, p, (1, 1, 1), 1, d)
    else:
        yr = F.conv3d(xr, wr, None, s, p, d)
    gy = torch.randn(yr.shape, generator=g)
    yr.backward(gy.double())
    old = autograd3d.set_bf16_mfma(bf16)
    try:
        xg = x.cuda().requires_grad_(True)
        wg = w.cuda().requires_grad_(True)
        if transposed:
            y = autograd3d._ConvTranspose3dFn.apply(xg, wg, None, s, p, (1, 1, 1), d)
        else:
            y = autograd3d._Conv3dFn.apply(xg, wg, None, s, p, d)
        y.backward(gy.cuda())
    finally:
        autograd3d.set_bf16_mfma(old)

    def rel(a, b):
        return float((a.detach().double().cpu() - b).norm() / b.norm())
    return rel(y, yr.detach()), rel(xg.grad, xr.grad), rel(wg.grad, wr.grad)


def main():
    print(f"{'layer':48s} {'K':>6s} | bf16 mode: fwd      dgrad    wgrad   | fp32 mode: fwd      dgrad    wgrad")
    worst = [0.0, 0.0]
    for layer in LAYERS:
        K = layer[1] * layer[3][0] * layer[3][1] * layer[3][2]
        b = run(layer, True)
        f = run(layer, False)
        worst = [max(worst[0], *b), max(worst[1], *f)]
        print(f"{layer[0]:48s} {K:6d} |           {b[0]:.2e} {b[1]:.2e} {b[2]:.2e} |            {f[0]:.2e} {f[1]:.2e} {f[2]:.2e}")
    print(f"worst relative L2 error: bf16 mode {worst[0]:.2e} (stated bound 6e-3), fp32 mode {worst[1]:.2e} (stated bound 1e-5)")
    assert worst[0] < 6e-3 and worst[1] < 1e-5


if __name__ == "__main__":
    main()
