"""dev: per-level timing of the decoder's eval path on the config-2 shapes (two views): tap GEMM, K12 gather, K10 over the
skip channels, second convolution (K10 or K9 + batched GEMMs)."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from occdepth_amd import hip
from occdepth_amd.models.unet2d import UpSampleBN
hip.load()
torch.backends.cuda.matmul.allow_tf32 = False
def t(fn, iters=10):
    fn(); torch.cuda.synchronize(); ts = []
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters): fn()
        e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / iters)
    return statistics.median(ts)
# (level, Cup, Cskip, Cout, (h, w), (H, W))
LEVELS = [(16, 2560, 224, 1280, (14, 41), (24, 77)), (8, 1280, 80, 640, (24, 77), (47, 153)), (4, 640, 48, 320, (47, 153), (93, 305)),
          (2, 320, 32, 160, (93, 305), (185, 610)), (1, 160, 3, 80, (185, 610), (370, 1220))]
tot = [0.0] * 4
with torch.no_grad():
    for lvl, cup, cs, cout, (h, w), (H, W) in LEVELS:
        m = UpSampleBN(cup + cs, cout).cuda().eval()
        x, skip = torch.randn(2, cup, h, w, device="cuda"), torch.randn(2, cs, H, W, device="cuda")
        n = m._net
        wpk9, w9, upk_skip, shift = m._upconv_operands(n[0], n[1], cup)
        z = torch.matmul(w9, x.view(2, cup, h * w)).view(2, 9 * cout, h, w)
        u = hip.upconv_gather(z, cout, (H, W))
        f = hip.conv2d_3x3_fused(skip, upk_skip, cout, shift, "leaky", 0.01, res=u, res_first=True)
        a = t(lambda: torch.matmul(w9, x.view(2, cup, h * w)))
        b = t(lambda: hip.upconv_gather(z, cout, (H, W)))
        c = t(lambda: hip.conv2d_3x3_fused(skip, upk_skip, cout, shift, "leaky", 0.01, res=u, res_first=True))
        d = t(lambda: m._conv_bn_act(f, n[3], n[4], n[5]))
        old = t(lambda: m._conv_bn_act(hip.upsample_bilinear_cat(x, skip), n[0], n[1], n[2]))
        for i, v in enumerate((a, b, c, d)): tot[i] += v
        gf = 2.0 * 2 * h * w * cup * 9 * cout / 1e9
        print(f"1/{lvl:<2d}: tap GEMM {a:.3f} ms ({gf / a:.0f} TF/s)  K12 {b:.3f}  K10 skip({cs}->{cout}) {c:.3f}  second conv {d:.3f}  | first conv, upsample+concat form: {old:.3f}", flush=True)
print("totals: tap GEMM %.2f  K12 %.2f  K10 skip %.2f  second %.2f  = %.2f ms" % (*tot, sum(tot)))
