"""K21 (occd_gemm_f32x3_splitk) against what the project convolutions of the 1/16 and 1/32 stages ran on in round 5: K16 with
the gate / skip epilogue (its in-workgroup split-K form where the host picks it) and K11s (exact fp32).  HIP events over 50
back-to-back launches, per problem and per (k16_per_z, nz, row_ranges) plan; prints a table (profiles/r06_gemm_splitk.txt)."""
import itertools
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from occdepth_amd import hip  # noqa: E402

DEV = "cuda"
CASES = [(2, 384, 468, 2304), (2, 640, 468, 3840), (2, 224, 1848, 1344), (2, 160, 1848, 960), (2, 640, 468, 2304)]


def timed(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / n


def main():
    hip.load()
    g = torch.Generator().manual_seed(0)
    for batch, M, N, K in CASES:
        w = (torch.randn(M, K, generator=g) / K ** 0.5).to(DEV)
        y = (torch.randn(batch, K, N, generator=g) * 2).to(DEV)
        gate = torch.sigmoid(torch.randn(batch, K, generator=g)).to(DEV)
        shift = torch.randn(M, generator=g).to(DEV)
        res = torch.randn(batch, M, N, generator=g).to(DEV)
        out = torch.empty(batch, M, N, device=DEV)
        pa = hip.GemmPacked(w, "a")
        gf = 2.0 * M * N * K * batch / 1e9
        t16 = timed(lambda: hip.gemm_x3(w, y, bias=shift, k_scale=gate, res=res, out=out))
        wpk = hip.pw_pack_weights(w.view(M, K, 1, 1))
        y4 = y.view(batch, K, 1, N)
        t11 = timed(lambda: hip.conv1x1(y4, wpk, M, shift, None, gate=gate, res=res.view(batch, M, 1, N)))
        print(f"{K}>{M} @{batch}x{N}: {gf:.2f} GFLOP | K16 (gate+skip) {t16:7.1f} us {gf / t16 * 1e3:6.1f} TF/s | K11s {t11:7.1f} us {gf / t11 * 1e3:6.1f} TF/s")
        k16 = (K + 15) // 16
        auto = hip.gemm_x3_splitk_plan(M, N, K, batch)[:3]
        plans = [auto]
        for nz, rr in itertools.product((2, 3, 4, 6, 8, 9, 12, 16, 18, 24), (1, 2, 3)):
            per = -(-k16 // nz)
            if per > 52 or (nz - 1) * per >= k16 or rr > (M + 31) // 32:
                continue
            plans.append((per, nz, rr))
        best = None
        for plan in plans:
            t = timed(lambda: hip.gemm_x3_splitk(pa, y, bias=shift, k_scale=gate, res=res, out=out, plan=plan))
            tag = " (auto)" if plan == auto and best is None else ""
            print(f"    K21 plan k16/z {plan[0]:3d} nz {plan[1]:3d} ranges {plan[2]}: {t:7.1f} us {gf / t * 1e3:6.1f} TF/s{tag}")
            if best is None or t < best[0]:
                best = (t, plan)
        print(f"    best {best[1]} {best[0]:.1f} us = {min(t16, t11) / best[0]:.2f}x the faster of K16 / K11s")


if __name__ == "__main__":
    main()
