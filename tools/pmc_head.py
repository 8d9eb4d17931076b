"""Run the dominant kernel (head conv 32->32 3x3x3 @256x256x32) a few times, for rocprofv3 --pmc passes:
    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d out -- python tools/pmc_head.py
    rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d out -- python tools/pmc_head.py
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from occdepth_amd import hip

torch.manual_seed(0)
dims = (256, 256, 32)
for d in (1, 2, 3):
    x = hip.Vox(torch.randn(1, *dims, 32, device="cuda"), 32)
    w = torch.randn(32, 32, 3, 3, 3, device="cuda") * 0.05
    wpk = hip.pack_weights(w)
    out = hip.Vox.empty(1, dims, 32, "cuda")
    for _ in range(3):
        hip.conv3d(x, wpk, None, 32, (3, 3, 3), out, dilation=(d,) * 3, padding=(d,) * 3)
    torch.cuda.synchronize()
print("done")
