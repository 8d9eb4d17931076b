"""dev: depthwise + SE-pooling kernel on the EfficientNet-B7 shapes of a config-2 frame (two views), GB/s against the
bytes it has to move (read x, write y)."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from occdepth_amd import hip
hip.load()
def t(fn, iters=10):
    fn(); torch.cuda.synchronize(); ts = []
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters): fn()
        e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / iters)
    return statistics.median(ts)
# (C, H, W, k, stride, count)
SHAPES = [(64, 185, 610, 3, 1, 1), (32, 185, 610, 3, 1, 3), (192, 185, 610, 3, 2, 1), (288, 93, 305, 3, 1, 6),
          (288, 93, 305, 5, 2, 1), (480, 47, 153, 5, 1, 6), (480, 47, 153, 3, 2, 1), (960, 24, 77, 3, 1, 9),
          (960, 24, 77, 5, 1, 1), (1344, 24, 77, 5, 1, 9), (1344, 24, 77, 5, 2, 1), (2304, 12, 39, 5, 1, 12),
          (2304, 12, 39, 3, 1, 1), (3840, 12, 39, 3, 1, 3)]
tot = 0.0
for C, H, W, k, s, n in SHAPES:
    x = torch.randn(2, C, H, W, device="cuda"); w = torch.randn(C, 1, k, k, device="cuda") * 0.2
    sc = torch.rand(C, device="cuda") + 0.5; sh = torch.randn(C, device="cuda") * 0.1
    ms = t(lambda: hip.dwconv2d_same_pool(x, w, sc, sh, s, "swish"))
    Ho, Wo = -(-H // s), -(-W // s)
    by = 4.0 * 2 * C * (H * W + Ho * Wo)
    tot += n * ms
    print(f"dw k{k} s{s} C={C:5d} @{H}x{W} x{n:2d}: {ms*1e3:7.1f} us  {by/ms/1e6:7.0f} GB/s", flush=True)
print(f"network total: {tot:.3f} ms")
