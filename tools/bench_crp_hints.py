"""dev: generic 3-D igemm variants (tile_hint 1..9, 0 = automatic) on the CRP / ASPP / bottleneck shapes of config 2."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from occdepth_amd import hip
from occdepth_amd.hip import Vox
hip.load()
def t(fn, iters=10):
    fn(); torch.cuda.synchronize(); ts = []
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters): fn()
        e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / iters)
    return statistics.median(ts)
# (cin, cout, dims, kernel, stride, dilation, padding, count per frame)
SHAPES = [(512, 512, (32, 32, 4), (1, 1, 1), 1, 1, 0, 4), (256, 512, (32, 32, 4), (1, 1, 1), 1, 1, 0, 4),
          (2304, 256, (32, 32, 4), (1, 1, 1), 1, 1, 0, 1), (256, 256, (32, 32, 4), (3, 3, 3), 1, 1, 1, 2),
          (256, 256, (32, 32, 4), (3, 3, 3), 1, 2, 2, 2), (256, 256, (32, 32, 4), (3, 3, 3), 1, 3, 3, 2),
          (256, 512, (32, 32, 4), (3, 3, 3), 2, 1, 1, 1), (64, 16, (128, 128, 16), (1, 1, 1), 1, 1, 0, 4),
          (16, 64, (128, 128, 16), (1, 1, 1), 1, 1, 0, 3), (128, 32, (64, 64, 8), (1, 1, 1), 1, 1, 0, 4),
          (256, 64, (32, 32, 4), (1, 1, 1), 1, 1, 0, 2), (64, 256, (32, 32, 4), (1, 1, 1), 1, 1, 0, 2)]
for cin, cout, dims, k, s, d, p, n in SHAPES:
    x = Vox.empty(1, dims, cin, torch.device("cuda")); x.buf.normal_()
    w = torch.randn(cout, cin, *k, device="cuda") * 0.05
    wpk = hip.pack_weights(w)
    od = tuple((dims[i] + 2 * p - d * (k[i] - 1) - 1) // s + 1 for i in range(3))
    out = Vox.empty(1, od, cout, torch.device("cuda"))
    row = []
    for h in range(0, 10):
        try:
            ms = t(lambda: hip.conv3d(x, wpk, None, cout, k, out, stride=(s,) * 3, dilation=(d,) * 3, padding=(p,) * 3, tile_hint=h))
            row.append(f"{h}:{ms*1e3:.0f}")
        except RuntimeError:
            row.append(f"{h}:-")
    fl = 2.0 * od[0] * od[1] * od[2] * k[0] * k[1] * k[2] * cin * cout
    auto = float(row[0].split(":")[1])
    print(f"{cin:5d}->{cout:4d} k{k[0]}{k[1]}{k[2]} s{s} d{d} @{dims} x{n}: auto {auto:.0f} us ({fl/auto/1e6:.0f} TF/s) | " + " ".join(row), flush=True)
