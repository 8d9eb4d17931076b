"""K16 (float32 operands) and K16p (pre-split weights, padded result rows) a few launches each on four GEMMs of the config-2 frame, for
rocprofv3 --pmc passes (dev tool; GPU):
    rocprofv3 --kernel-trace --pmc <counters> --output-format csv -d out -- python tools/pmc_gemm_panel.py
then  python tools/pmc_table.py out/*/*counter_collection.csv --match gemm_x3"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from occdepth_amd import hip

torch.manual_seed(0)
for (M, N, K) in ((720, 112850, 160), (1440, 28365, 320), (2304, 468, 384), (960, 1848, 160)):
    a = torch.randn(M, K, device="cuda") / K ** 0.5
    b = torch.randn(2, K, N, device="cuda")
    out = hip.padded_rows((2, M, N), "cuda")
    pa = hip.GemmPacked(a, "a")
    for _ in range(3):
        hip.gemm_x3(a, b, out=out)
    for _ in range(3):
        hip.gemm_x3(pa, b, out=out, tile_hint=8)
torch.cuda.synchronize()
print("done")
