"""One K16 launch shape under `rocprofv3 --att`: the tap GEMM of the 1/8 decoder level, 5760 x 1848 x 1280,
batch 2, weights as the pre-split fragment image (K16's PRE = 1 form) -- five launches after a warm-up."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from occdepth_amd import hip  # noqa: E402

hip.load()
g = torch.Generator().manual_seed(0)
w = (torch.randn(5760, 1280, generator=g) / 1280 ** 0.5).cuda()
x = torch.randn(2, 1280, 1848, generator=g).cuda()
pa = hip.GemmPacked(w, "a")
out = torch.empty(2, 5760, 1848, device="cuda")
for _ in range(2):
    hip.gemm_x3(pa, x, out=out)
torch.cuda.synchronize()
for _ in range(5):
    hip.gemm_x3(pa, x, out=out)
torch.cuda.synchronize()
print("done", float(out[0, 0, 0]))
