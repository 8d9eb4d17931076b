"""Pointwise (1x1) convolutions of the EfficientNet-B7 encoder in the TRAINING step: forward + data gradient + weight gradient
on K16 / K16t (hip._PwConvFn) against ATen (MIOpen / rocBLAS) per layer shape (dev tool; GPU):  python tools/bench_pw_train.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F

from occdepth_amd import hip
from bench_kernels import time_many

# (cin, cout, H, W, how many such layers in B7)
SHAPES = [(64, 32, 185, 610, 1), (32, 32, 185, 610, 3), (32, 192, 185, 610, 1), (192, 48, 93, 305, 1), (48, 288, 93, 305, 7),
          (288, 48, 93, 305, 6), (288, 80, 47, 153, 1), (80, 480, 47, 153, 7), (480, 80, 47, 153, 6), (480, 160, 24, 77, 1),
          (160, 960, 24, 77, 10), (960, 160, 24, 77, 9), (960, 224, 24, 77, 1), (224, 1344, 24, 77, 10), (1344, 224, 24, 77, 9),
          (1344, 384, 12, 39, 1), (384, 2304, 12, 39, 13), (2304, 384, 12, 39, 12), (2304, 640, 12, 39, 1), (640, 3840, 12, 39, 3),
          (3840, 640, 12, 39, 3), (640, 2560, 12, 39, 1)]


def main():
    torch.manual_seed(0)
    tot = {"aten": 0.0, "k16": 0.0, "best": 0.0}
    for cin, cout, H, W, n in SHAPES:
        x = torch.randn(2, cin, H, W, device="cuda", requires_grad=True)
        w = (torch.randn(cout, cin, 1, 1, device="cuda") / cin ** 0.5).requires_grad_(True)
        gy = torch.randn(2, cout, H, W, device="cuda")

        def aten():
            y = F.conv2d(x, w)
            y.backward(gy)
            x.grad = w.grad = None

        def k16():
            y = hip.pw_conv_autograd(x, w)
            y.backward(gy)
            x.grad = w.grad = None

        ms = time_many({"aten": aten, "k16": k16}, rounds=3, iters=4)
        fl = 3 * 2.0 * 2 * cin * cout * H * W
        tot["aten"] += n * ms["aten"]
        tot["k16"] += n * ms["k16"]
        tot["best"] += n * min(ms.values())
        print(f"{cin:5d} -> {cout:5d} @{H}x{W} x{n:2d}: ATen {ms['aten']:7.3f} ms ({fl / ms['aten'] / 1e9:6.1f} TF/s)   "
              f"K16 {ms['k16']:7.3f} ms ({fl / ms['k16'] / 1e9:6.1f} TF/s)   pixels {2 * H * W}", flush=True)
    print("per step (fwd + dgrad + wgrad of all layers):", {k: round(v, 2) for k, v in tot.items()}, "ms")


if __name__ == "__main__":
    main()
