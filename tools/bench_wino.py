"""Go / no-go numbers for an MFMA Winograd F(2x2,3x3) path of the 2-D decoder (dev tool, GPU):
MIOpen's 3x3 convolution vs the 16 batched GEMMs (T x Cin).(Cin x Cout) of the Winograd domain, per decoder level."""
import statistics
import sys
import os

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F

torch.backends.cudnn.allow_tf32 = False
torch.backends.cuda.matmul.allow_tf32 = False
LEVELS = [((24, 77), 2784, 1280), ((24, 77), 1280, 1280), ((47, 153), 1360, 640), ((47, 153), 640, 640),
          ((93, 305), 688, 320), ((93, 305), 320, 320), ((185, 610), 352, 160), ((185, 610), 160, 160),
          ((370, 1220), 163, 80), ((370, 1220), 80, 80)]


def t(fn, iters=5):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / iters)
    return statistics.median(ts)


tot_c = tot_g = 0.0
for (H, W), cin, cout in LEVELS:
    x = torch.randn(2, cin, H, W, device="cuda")
    w = torch.randn(cout, cin, 3, 3, device="cuda") * 0.02
    T = 2 * ((H + 1) // 2) * ((W + 1) // 2)
    V = torch.randn(16, T, cin, device="cuda")
    U = torch.randn(16, cin, cout, device="cuda")
    M = torch.empty(16, T, cout, device="cuda")
    tc = t(lambda: F.conv2d(x, w, padding=1))
    tg = t(lambda: torch.bmm(V, U, out=M))
    fl = 2.0 * 2 * H * W * 9 * cin * cout
    flg = 2.0 * 16 * T * cin * cout
    gb = 4.0 * (V.numel() + M.numel()) * 2 / 1e9            # transform traffic: V and M are each written once and read once
    tot_c += tc
    tot_g += tg
    print(f"{cin:5d}->{cout:4d} @{H}x{W}: MIOpen conv {tc:.3f} ms ({fl / tc / 1e9:5.1f} TF/s direct-equiv) | "
          f"bmm 16x({T}x{cin}x{cout}) {tg:.3f} ms ({flg / tg / 1e9:5.1f} TF/s) | transform traffic {gb:.2f} GB", flush=True)
print(f"total: MIOpen {tot_c:.2f} ms, batched GEMMs {tot_g:.2f} ms")

# ---- K10: the fused kernel per level (both wave shapes) against the two paths above
from occdepth_amd import hip  # noqa: E402

hip.load()
print("\nK10 fused Winograd kernel (csrc/wino_conv2d.hip), batch 2:")
tot = {"miopen": 0.0, "unfused": 0.0, "fused": 0.0}
for (H, W), cin, cout in LEVELS:
    x = torch.randn(2, cin, H, W, device="cuda")
    w = torch.randn(cout, cin, 3, 3, device="cuda") * 0.02
    sc, sh = torch.rand(cout, device="cuda") + 0.5, torch.randn(cout, device="cuda")
    upk = hip.wino_pack_weights(w, sc)
    U = hip.winograd_weights(w)
    y = torch.empty(2, cout, H, W, device="cuda")
    tm = t(lambda: hip.affine_act(F.conv2d(x, w, padding=1), sc, sh, "leaky"))
    tu = t(lambda: hip.conv2d_3x3_winograd(x, U, sc, sh, "leaky"))
    tf = {h: t(lambda: hip.conv2d_3x3_fused(x, upk, cout, sh, "leaky", tile_hint=h, out=y)) for h in (16, 32)}
    ref = hip.affine_act(F.conv2d(x, w, padding=1), sc, sh, "leaky")
    err = float((hip.conv2d_3x3_fused(x, upk, cout, sh, "leaky") - ref).abs().max() / ref.abs().max())
    fl = 2.0 * 2 * H * W * 9 * cin * cout
    best = min(tf.values())
    tot["miopen"] += tm
    tot["unfused"] += tu
    tot["fused"] += best
    print(f"{cin:5d}->{cout:4d} @{H}x{W}: MIOpen+BN/act {tm:.3f} ms | unfused wino {tu:.3f} ms | fused 2x16 {tf[16]:.3f} "
          f"1x32 {tf[32]:.3f} ms ({fl / best / 1e9:5.1f} TF/s direct-equiv, {fl / 2.25 / best / 1e9:5.1f} TF/s MFMA) "
          f"| vs MIOpen err {err:.1e}", flush=True)
print("totals (ms):", {k: round(v, 3) for k, v in tot.items()})
