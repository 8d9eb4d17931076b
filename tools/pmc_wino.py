"""dev: a few launches of K10 on one decoder level, for rocprofv3 --pmc runs (argv: H W cin cout hint...)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from occdepth_amd import hip
hip.load()
H, W, cin, cout = (int(v) for v in sys.argv[1:5])
hints = [int(v) for v in sys.argv[5:]] or [0]
x = torch.randn(2, cin, H, W, device="cuda"); w = torch.randn(cout, cin, 3, 3, device="cuda") * 0.02
upk = hip.wino_pack_weights(w); y = torch.empty(2, cout, H, W, device="cuda")
for h in hints:
    for _ in range(3):
        hip.conv2d_3x3_fused(x, upk, cout, None, "leaky", tile_hint=h, out=y)
torch.cuda.synchronize()
