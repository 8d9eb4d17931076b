"""K16 / K16w / K2s3 a few launches each, for rocprofv3 --pmc passes (dev tool; GPU):
    rocprofv3 --kernel-trace --pmc <counters> --output-format csv -d out -- python tools/pmc_gemm.py
then  python tools/pmc_table.py out/*/*counter_collection.csv ..."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from occdepth_amd import hip

torch.manual_seed(0)
# the tap GEMM of the 1/8 decoder level and the Winograd-domain product of the 1/8 level
for (batch, M, N, K, ab) in ((2, 5760, 1848, 1280, False), (16, 3696, 640, 640, True)):
    a = torch.randn(*((batch, M, K) if ab else (M, K)), device="cuda") / K ** 0.5
    b = torch.randn(batch, K, N, device="cuda")
    out = torch.empty(batch, M, N, device="cuda")
    for hint in (1, 6):
        for _ in range(3):
            hip.gemm_x3(a, b, out=out, tile_hint=hint)
    for _ in range(3):
        torch.matmul(a, b, out=out)
dims = (256, 256, 32)
x = hip.Vox(torch.randn(1, *dims, 32, device="cuda"), 32)
w = torch.randn(32, 32, 3, 3, 3, device="cuda") * 0.05
w3, w32 = hip.pack_weights_bf16(w, split3=True), hip.pack_weights(w)
o = hip.Vox.empty(1, dims, 32, "cuda")
for d in (1, 3):
    for _ in range(3):
        hip.conv3d_bf16(x, w3, None, 32, (3, 3, 3), o, dilation=(d,) * 3, padding=(d,) * 3, split3=True)
        hip.conv3d(x, w32, None, 32, (3, 3, 3), o, dilation=(d,) * 3, padding=(d,) * 3)
torch.cuda.synchronize()
print("done")
