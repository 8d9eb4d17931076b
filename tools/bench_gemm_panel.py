"""K16p (panel-stationary, pre-split A) against K16 on float32 operands on the short-K GEMMs of the config-2 frame -- the
expand convolutions (bias + swish epilogue) and the 1/1 tap GEMM (dev tool; GPU):  python tools/bench_gemm_panel.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from occdepth_amd import hip
from bench_kernels import time_many

# (label, batch, M, N, K, epilogue)
SHAPES = [
    ("tap 1/1    160>9x80  @185x610", 2, 720, 112850, 160, False),
    ("tap 1/2    320>9x160 @93x305", 2, 1440, 28365, 320, False),
    ("expand 1/32 384>2304 @12x39", 2, 2304, 468, 384, True),
    ("expand 1/32 640>3840 @12x39", 2, 3840, 468, 640, True),
    ("head 1/32   640>2560 @14x41", 2, 2560, 574, 640, True),
    ("expand 1/16 224>1344 @24x77", 2, 1344, 1848, 224, True),
    ("expand 1/16 160>960  @24x77", 2, 960, 1848, 160, True),
    ("expand 1/8  80>480   @47x153", 2, 480, 7191, 80, True),
    ("expand 1/4  48>288   @93x305", 2, 288, 28365, 48, True),
    ("expand 1/2  32>192   @185x610", 2, 192, 112850, 32, True),
]


def main():
    torch.manual_seed(0)
    for label, batch, M, N, K, epi in SHAPES:
        a = torch.randn(M, K, device="cuda") / K ** 0.5
        b = torch.randn(batch, K, N, device="cuda")
        bias = torch.randn(M, device="cuda") if epi else None
        act = "swish" if epi else None
        out = hip.padded_rows((batch, M, N), "cuda")
        pa = hip.GemmPacked(a, "a")
        fns = {"K16 auto (float32 A)": lambda: hip.gemm_x3(a, b, out=out, bias=bias, act=act),
               "K16 64x64 pre-split A": lambda: hip.gemm_x3(pa, b, out=out, bias=bias, act=act, tile_hint=4),
               "K16p": lambda: hip.gemm_x3(pa, b, out=out, bias=bias, act=act, tile_hint=8)}
        ms = time_many(fns, rounds=3, iters=10)
        fl = 2.0 * M * N * K * batch
        by = 4.0 * batch * N * (M + K)
        line = "  ".join(f"{k}: {t * 1e3:7.1f} us {fl / t / 1e9:6.1f} TF/s {by / t / 1e6:6.0f} GB/s" for k, t in ms.items())
        print(f"{label:32s} {line}", flush=True)


if __name__ == "__main__":
    main()
