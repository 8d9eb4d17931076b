"""K16 (occd_gemm_f32x3) against torch.matmul / torch.bmm (hipBLASLt / rocBLAS float32) on the GEMMs of the config-2 frame
(dev tool; GPU):  python tools/bench_gemm_x3.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from occdepth_amd import hip
from bench_kernels import time_many

# (label, batch, M, N, K, A batched)
SHAPES = [
    ("tap 1/16  2560>9x1280 @14x41", 2, 11520, 574, 2560, False),
    ("tap 1/8   1280>9x640  @24x77", 2, 5760, 1848, 1280, False),
    ("tap 1/4   640>9x320   @47x153", 2, 2880, 7191, 640, False),
    ("tap 1/2   320>9x160   @93x305", 2, 1440, 28365, 320, False),
    ("tap 1/1   160>9x80    @185x610", 2, 720, 112850, 160, False),
    ("wino 1/16 1280>1280 T=936", 16, 936, 1280, 1280, True),
    ("wino 1/8  640>640   T=3696", 16, 3696, 640, 640, True),
    ("expand 1/16 224>1344 @24x77", 2, 1344, 1848, 224, False),
    ("expand 1/32 384>2304 @12x39", 2, 2304, 468, 384, False),
    ("expand 1/32 640>3840 @12x39", 2, 3840, 468, 640, False),
]


def main():
    torch.manual_seed(0)
    for label, batch, M, N, K, ab in SHAPES:
        a = torch.randn(*((batch, M, K) if ab else (M, K)), device="cuda") / K ** 0.5
        b = torch.randn(batch, K, N, device="cuda")
        out = torch.empty(batch, M, N, device="cuda")
        fns = {"torch.matmul fp32": lambda: torch.matmul(a, b, out=out)}
        for hint, nm in ((1, "K16 256x128"), (2, "K16 128x128"), (6, "K16w 256x128 wave-specialised"), (0, "K16 auto")):
            fns[nm] = lambda hint=hint: hip.gemm_x3(a, b, out=out, tile_hint=hint)
        ms = time_many(fns, rounds=3, iters=5)
        fl = 2.0 * M * N * K * batch
        line = "  ".join(f"{k}: {t:7.3f} ms {fl / t / 1e9:6.1f} TF/s" for k, t in ms.items())
        print(f"{label:34s} {fl / 1e9:6.1f} GFLOP  {line}", flush=True)


if __name__ == "__main__":
    main()
