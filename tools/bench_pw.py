"""dev: K11 (pointwise GEMM) vs F.conv2d (rocBLAS / MIOpen) on the 1x1 convolution shapes of B7 + decoder heads, batch 2"""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from occdepth_amd import hip
hip.load()
torch.backends.cudnn.allow_tf32 = False
torch.backends.cuda.matmul.allow_tf32 = False
def t(fn, iters=5):
    fn(); torch.cuda.synchronize(); ts = []
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters): fn()
        e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / iters)
    return statistics.median(ts)
# (cin, cout, H, W, count in the network)
SHAPES = [(64, 32, 185, 610, 1), (32, 32, 185, 610, 3), (32, 192, 185, 610, 1), (192, 48, 93, 305, 1), (48, 288, 93, 305, 7),
          (288, 48, 93, 305, 6), (288, 80, 47, 153, 1), (80, 480, 47, 153, 7), (480, 80, 47, 153, 6), (480, 160, 24, 77, 1),
          (160, 960, 24, 77, 10), (960, 160, 24, 77, 9), (960, 224, 24, 77, 1), (224, 1344, 24, 77, 10), (1344, 224, 24, 77, 9),
          (1344, 384, 12, 39, 1), (384, 2304, 12, 39, 13), (2304, 384, 12, 39, 12), (2304, 640, 12, 39, 1),
          (640, 3840, 12, 39, 4), (3840, 640, 12, 39, 3), (640, 2560, 12, 39, 1),
          (1280, 64, 24, 77, 1), (640, 64, 47, 153, 1), (320, 64, 93, 305, 1), (160, 64, 185, 610, 1), (80, 64, 370, 1220, 1),
          (128, 104, 47, 153, 1)]
tot_ref = tot_k = 0.0
for cin, cout, H, W, n in SHAPES:
    x = torch.randn(2, cin, H, W, device="cuda"); w = torch.randn(cout, cin, 1, 1, device="cuda") * 0.05
    wpk = hip.pw_pack_weights(w); y = torch.empty(2, cout, H, W, device="cuda")
    tr = t(lambda: F.conv2d(x, w))
    best, bh = 1e9, 0
    row = []
    for h in range(0, 13):
        try:
            ms = t(lambda: hip.conv1x1(x, wpk, cout, tile_hint=h, out=y))
        except RuntimeError:
            continue
        row.append(f"{h}:{ms:.3f}")
        if h and ms < best: best, bh = ms, h
    auto = float(row[0].split(":")[1])
    fl = 2.0 * 2 * H * W * cin * cout
    by = 4.0 * 2 * H * W * (cin + cout)
    tot_ref += n * tr; tot_k += n * auto
    print(f"{cin:5d}->{cout:5d} @{H}x{W} x{n:2d}: torch {tr:.3f} ms | K11 auto {auto:.3f} (best hint {bh}: {best:.3f}; {fl/best/1e9:5.1f} TF/s, {by/best/1e6:6.0f} GB/s) | " + " ".join(row), flush=True)
print(f"network totals: torch {tot_ref:.2f} ms, K11 auto {tot_k:.2f} ms")
