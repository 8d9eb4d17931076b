"""VERDICT r5 item 3b -- "a measured prototype of the direct 3x3 on the split": the three large second convolutions of the 2-D
decoder levels (80 -> 80 @370x1220, 160 -> 160 @185x610, 320 -> 320 @93x305, batch 2 = both stereo views; ~104.5 GFLOP direct each)
  * on K10 (the shipped fused Winograd F(2x2,3x3) kernel on the fp32 matrix pipe, NCHW in / out, BN + LeakyReLU fused), and
  * as a DIRECT convolution on the bf16 pipe with the 3-way operand split -- K2b `conv3d_bf16_kernel<.., SPLIT=3>`, the repo's
    generic direct-convolution kernel with that arithmetic (weights streamed from L2 as pre-split fragments, activations
    staged + split per tile; the image is an X = 1 volume with a (1, 3, 3) kernel, channels-last rows) -- the form DESIGN
    section 10 argued a decoder-level direct kernel would land at;
  * plus the NCHW <-> channels-last transposes the decoder would pay around it if only this convolution moved.
Prints per level: time, direct-equivalent TF/s, max error against float64 (first 8 output channels of a crop)."""
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F

from occdepth_amd import hip  # noqa: E402
from occdepth_amd.fused import _pad_bias  # noqa: E402

LEVELS = [((370, 1220), 80), ((185, 610), 160), ((93, 305), 320)]


def t(fn, iters=10):
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


def main():
    hip.load()
    g = torch.Generator().manual_seed(0)
    for (H, W), c in LEVELS:
        x = torch.randn(2, c, H, W, generator=g).cuda()
        w = (torch.randn(c, c, 3, 3, generator=g) / (9 * c) ** 0.5).cuda()
        sc = (torch.rand(c, generator=g) + 0.5).cuda()
        sh = torch.randn(c, generator=g).cuda()
        fl = 2.0 * 2 * H * W * 9 * c * c
        # ---- K10
        upk = hip.wino_pack_weights(w, sc)
        y = torch.empty(2, c, H, W, device="cuda")
        t10 = min(t(lambda: hip.conv2d_3x3_fused(x, upk, c, sh, "leaky", tile_hint=h, out=y)) for h in (16, 32))
        y10 = hip.conv2d_3x3_fused(x, upk, c, sh, "leaky").clone()
        # ---- K2b SPLIT=3 on channels-last rows (X = 1 volume), BN scale folded into the weights, shift as bias, no activation
        tin = t(lambda: hip.nchw_to_nhwc(x))
        rows = hip.nchw_to_nhwc(x)                                    # (B, H, W, cs)
        vx = hip.Vox(rows.view(2, 1, H, W, rows.shape[-1]), c)
        w3 = (w * sc.view(-1, 1, 1, 1)).view(c, c, 1, 3, 3)
        wpk = hip.pack_weights_bf16(w3, split3=True)
        out = hip.Vox.empty(2, (1, H, W), c, "cuda")
        bias = _pad_bias(sh, c)
        with hip.profile() as prof:
            hip.conv3d_bf16(vx, wpk, bias, c, (1, 3, 3), out, padding=(0, 1, 1), split3=True)
        kinds = sorted({k.split(":")[0] for k in prof.rows})
        t2b = t(lambda: hip.conv3d_bf16(vx, wpk, bias, c, (1, 3, 3), out, padding=(0, 1, 1), split3=True))
        tout = t(lambda: hip.nhwc_to_nchw(out))
        # ---- error of both against float64 on a crop (first 8 couts)
        xs = x[:1, :, :40, :64].double().cpu()
        ref = F.conv2d(xs, (w * sc.view(-1, 1, 1, 1)).double().cpu()[:8], padding=1) + sh.double().cpu()[:8].view(1, -1, 1, 1)
        got2b = out.buf.view(2, H, W, out.cs)[0, :39, :63, :8].permute(2, 0, 1).double().cpu()
        e2b = float((got2b - ref[0, :, :39, :63]).abs().max() / ref.abs().max())
        ref10 = F.leaky_relu(ref, 0.01)
        e10 = float((y10[0, :8, :39, :63].double().cpu() - ref10[0, :, :39, :63]).abs().max() / ref10.abs().max())
        print(f"{c:4d}->{c:4d} @2x{H}x{W} ({fl / 1e9:.1f} GFLOP direct): K10 {t10:.3f} ms = {fl / t10 / 1e9:6.1f} TF/s direct-equiv "
              f"(err {e10:.1e}) | direct split ({'+'.join(kinds)}) {t2b:.3f} ms = {fl / t2b / 1e9:6.1f} TF/s (err {e2b:.1e}) "
              f"| NCHW->rows {tin:.3f} ms, rows->NCHW {tout:.3f} ms", flush=True)


if __name__ == "__main__":
    main()
