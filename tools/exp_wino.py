"""dev: A/B of K10 variants (compute-only / DMA-only loops) on a few decoder levels"""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from occdepth_amd import hip
hip.load()
def t(fn, iters=5):
    fn(); torch.cuda.synchronize(); ts = []
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters): fn()
        e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / iters)
    return statistics.median(ts)
for (H, W), cin, cout in [((24, 77), 2784, 1280), ((93, 305), 688, 320), ((185, 610), 160, 160), ((370, 1220), 80, 80)]:
    x = torch.randn(2, cin, H, W, device="cuda"); w = torch.randn(cout, cin, 3, 3, device="cuda") * 0.02
    upk = hip.wino_pack_weights(w); y = torch.empty(2, cout, H, W, device="cuda")
    fl = 2.0 * 16 * 2 * ((H + 1) // 2) * ((W + 1) // 2) * ((cin + 7) // 8 * 8) * ((cout + 31) // 32 * 32)
    row = []
    for hint in (32, 16, 132, 232):
        ms = t(lambda: hip.conv2d_3x3_fused(x, upk, cout, None, "leaky", tile_hint=hint, out=y))
        row.append(f"{hint}: {ms:.3f} ms ({fl / ms / 1e9:5.1f} TF/s)")
    print(f"{cin}->{cout} @{H}x{W}: " + " | ".join(row), flush=True)
