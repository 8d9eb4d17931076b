"""K11 (csrc/pw_gemm.hip): pointwise convolution + folded BatchNorm + activation (+ squeeze-excite gate on the input,
+ skip add) as one MFMA GEMM launch on NCHW maps.  Reference: float64 einsum on the CPU.
CPU: host-side wrapper logic through the test-only emulation; `-m gpu`: the kernel, every tile variant."""
import contextlib

import pytest
import torch

import emu

# (B, Cin, Cout, spatial, act, gate, residual)
CASES = [
    (2, 32, 192, (25, 41), "swish", False, False),      # expand conv of an MBConv block, ragged pixel count
    (1, 288, 48, (31, 30), None, True, True),           # project conv: SE gate on the input + skip add
    (2, 3840, 640, (12, 39), None, True, False),        # widest project conv of B7, few pixels
    (2, 640, 2560, (12, 39), None, False, False),       # conv_head
    (1, 83, 64, (50, 100), None, False, False),         # resize_output-like: ragged cin (tail chunk), bias via shift
    (3, 20, 10, (7,), "leaky", True, True),             # tiny everything, 1-D spatial
    (1, 128, 104, (47, 153), None, False, False),       # DepthNet.depth_pred
    (2, 2304, 384, (12, 39), None, True, True),         # stage-6 project conv (split-K variants by default)
    (2, 160, 960, (24, 77), "swish", False, False),     # stage-4 expand conv
    (1, 1344, 224, (5, 9), None, True, True),           # fewer pixels than one tile, K not a multiple of the wave split
]


def run(case, device, tol, hints=(0,)):
    from occdepth_amd import hip
    B, cin, cout, sp, act, with_gate, with_res = case
    g = torch.Generator().manual_seed(B * 100 + cin + cout)
    x = torch.randn(B, cin, *sp, generator=g)
    w = torch.randn(cout, cin, 1, 1, generator=g) * (1.0 / cin ** 0.5)
    scale, shift = torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g)
    gate = torch.rand(B, cin, 1, 1, generator=g) if with_gate else None
    res = torch.randn(B, cout, *sp, generator=g) if with_res else None
    xd = x.double() * (gate.double().reshape(B, cin, *([1] * len(sp))) if with_gate else 1.0)
    ref = torch.einsum("oc,bc...->bo...", w.double().reshape(cout, cin) * scale.double().view(-1, 1), xd)
    ref = ref + shift.double().view(1, -1, *([1] * len(sp)))
    ref = {"leaky": lambda t: torch.nn.functional.leaky_relu(t, 0.01), "swish": lambda t: t * torch.sigmoid(t),
           None: lambda t: t}[act](ref)
    if with_res:
        ref = ref + res.double()
    dev = torch.device(device)
    worst = 0.0
    with (emu.patched() if device == "cpu" else contextlib.nullcontext()):
        wpk = hip.pw_pack_weights(w.to(dev), scale.to(dev))
        for h in hints:
            y = hip.conv1x1(x.to(dev), wpk, cout, shift.to(dev), act, 0.01, gate.to(dev) if with_gate else None,
                            res.to(dev) if with_res else None, tile_hint=h)
            err = float((y.double().cpu() - ref).abs().max() / ref.abs().max())
            assert y.shape == ref.shape and err < tol, (case, h, err)
            worst = max(worst, err)
    return worst


@pytest.mark.parametrize("case", CASES)
def test_pointwise_conv_host_logic_cpu(case):
    run(case, "cpu", 2e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES)
def test_pointwise_conv_kernel_gpu(case, hip_lib):
    err = run(case, "cuda", 2e-5, hints=tuple(range(0, 13)))          # 1..6 = K11, 7..12 = K11s (split-K)
    print(case, f"worst rel err vs float64 over the tile variants {err:.2e}")


@pytest.mark.gpu
@pytest.mark.parametrize("case", [(2, 1280, 64, (24, 77)), (1, 83, 61, (50, 100)), (2, 160, 64, (9, 300))])
def test_pointwise_conv_pixel_major_gpu(case, hip_lib):
    """nhwc=True (what the decoder heads hand to the 2D->3D lift): every K11 / K11s variant writes pixel-major rows of
    ceil4(Cout) floats with a zero channel pad, and returns the logical (B, Cout, H, W) view of them."""
    from occdepth_amd import hip
    B, cin, cout, sp = case
    g = torch.Generator().manual_seed(cin + cout)
    x = torch.randn(B, cin, *sp, generator=g)
    w = torch.randn(cout, cin, 1, 1, generator=g) * (1.0 / cin ** 0.5)
    shift = torch.randn(cout, generator=g)
    ref = torch.einsum("oc,bchw->bohw", w.double().reshape(cout, cin), x.double()) + shift.double().view(1, -1, 1, 1)
    wpk = hip.pw_pack_weights(w.cuda())
    for h in range(0, 13):
        y = hip.conv1x1(x.cuda(), wpk, cout, shift.cuda(), nhwc=True, tile_hint=h)
        assert y.shape == ref.shape and y.stride(1) == 1, (h, y.shape, y.stride())
        err = float((y.double().cpu() - ref).abs().max() / ref.abs().max())
        assert err < 2e-5, (case, h, err)
        rows = y.permute(0, 2, 3, 1)                               # the storage: (B, H, W, ceil4(Cout)) rows
        base = torch.as_strided(rows, (B, sp[0], sp[1], (cout + 3) // 4 * 4), rows.stride())
        assert float(base[..., cout:].abs().max()) == 0.0 if cout % 4 else True


# ---------------------------------------------------------------------------------------------------------------
# depthwise + SE pooling, SE gate, and whole EfficientNet blocks on the fused path
@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(2, 24, 37, 53, 3, 1), (1, 40, 30, 41, 5, 2), (2, 7, 9, 300, 5, 1),
                                   # Cr % 4 == 0: the float4 expand kernel (3 chunks; 24 chunks > its 12-chunk preload x 4 lanes? no: 50 chunks)
                                   (2, 48, 13, 17, 3, 1), (2, 96, 6, 9, 3, 1), (1, 800, 5, 7, 5, 1)])
def test_dwconv_pool_and_se_gate_gpu(shape, hip_lib):
    from occdepth_amd import hip
    B, C, H, W, k, stride = shape
    g = torch.Generator().manual_seed(C + H)
    x = torch.randn(B, C, H, W, generator=g)
    w = torch.randn(C, 1, k, k, generator=g) * 0.3
    sc, sh = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.1
    y, part, plane = hip.dwconv2d_same_pool(x.cuda(), w.cuda(), sc.cuda(), sh.cuda(), stride, "swish")
    y_ref, part_ref, plane_ref = emu.dwconv2d_same_pool(x, w, sc, sh, stride, "swish")
    assert plane == plane_ref and torch.allclose(y.cpu(), y_ref, rtol=1e-5, atol=1e-5)
    y2 = hip.dwconv2d_same(x.cuda(), w.cuda(), sc.cuda(), sh.cuda(), stride, "swish")
    assert torch.equal(y, y2)                                   # same kernel, pooling on / off
    assert torch.allclose(part.double().sum(1).cpu(), part_ref.double().sum(1), rtol=1e-5, atol=1e-4)
    y3, part3, _ = hip.dwconv2d_same_pool(x.cuda(), w.cuda(), sc.cuda(), sh.cuda(), stride, "swish")
    assert torch.equal(part, part3)                             # fixed summation order: bit-reproducible
    # input planes on a 128-byte pitch (the view of an expand GEMM's hip.padded_rows result; NaN in the padding): the plane
    # stride travels to the kernel, no copy, bit-identical output and partials
    xp = hip.padded_rows((B, C, H * W), "cuda")
    torch.as_strided(xp, (B, C, xp.stride(1)), (xp.stride(0), xp.stride(1), 1)).fill_(float("nan"))
    xp.copy_(x.cuda().view(B, C, H * W))
    y4, part4, _ = hip.dwconv2d_same_pool(xp.view(B, C, H, W), w.cuda(), sc.cuda(), sh.cuda(), stride, "swish")
    assert torch.equal(y4, y) and torch.equal(part4, part)
    Cr = max(1, C // 4)
    wr, br = torch.randn(Cr, C, 1, 1, generator=g) * 0.3, torch.randn(Cr, generator=g) * 0.1
    we, be = torch.randn(C, Cr, 1, 1, generator=g) * 0.3, torch.randn(C, generator=g) * 0.1
    gate = hip.se_gate(part, plane, B, wr.cuda(), br.cuda(), we.cuda(), be.cuda())
    ref = emu.se_gate(part_ref, plane_ref, B, wr, br, we, be)
    assert gate.shape == (B, C) and torch.allclose(gate.cpu(), ref, rtol=1e-5, atol=1e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["ir_skip", "ir_stride2", "ds_skip"])
def test_efficientnet_block_fused_path_gpu(kind, hip_lib):
    """A whole MBConv / depthwise-separable block on the fused path (4 / 3 launches) against the module's own
    float64 CPU forward (the generic nn path)."""
    import copy
    from occdepth_amd.models import efficientnet as E
    torch.manual_seed(5)
    m = {"ir_skip": lambda: E.InvertedResidual(24, 24, 5, 1, 6), "ir_stride2": lambda: E.InvertedResidual(24, 40, 3, 2, 6),
         "ds_skip": lambda: E.DepthwiseSeparableConv(32, 32, 3, 1)}[kind]()
    for mod in m.modules():
        if isinstance(mod, torch.nn.BatchNorm2d):
            mod.running_mean.normal_(0, 0.2)
            mod.running_var.uniform_(0.5, 1.5)
            mod.weight.data.uniform_(0.5, 1.5)
            mod.bias.data.normal_(0, 0.2)
    m.eval()
    x = torch.randn(2, m.conv_dw.in_channels if kind == "ds_skip" else m.conv_pw.in_channels, 37, 61)
    saved = (E.PW_MIN_PIXELS, E.PW_EXPAND_LIB_BELOW, E.PW_PROJECT_K16_MIN_PIXELS, E.PW_PROJECT_K16_MIN_COUT)
    E.PW_MIN_PIXELS = 0                                        # (the product only takes this path on large maps)
    try:
        from occdepth_amd import hip
        with torch.no_grad():
            ref = copy.deepcopy(m).double()(x.double())
            assert E.PW_FUSED and E.pw_wins(x)
            mc = m.cuda()
            # expand / project convolutions on K16 (gate on the staged B rows, skip in the epilogue) ...
            E.PW_EXPAND_LIB_BELOW, E.PW_PROJECT_K16_MIN_PIXELS, E.PW_PROJECT_K16_MIN_COUT = 1 << 40, 0, 0
            with hip.profile() as prof16:
                got16 = mc(x.cuda())
            # ... and on K11 (the exact-fp32 pointwise GEMM with the same fusions)
            E.PW_EXPAND_LIB_BELOW, E.PW_PROJECT_K16_MIN_PIXELS = 0, 1 << 40
            with hip.profile() as prof11:
                got11 = mc(x.cuda())
        assert any(k.startswith("gemm_f32x3") for k in prof16.rows) and not any(k.startswith("pw_conv") for k in prof16.rows)
        assert any(k.startswith("pw_conv") for k in prof11.rows) and not any(k.startswith("gemm_f32x3") for k in prof11.rows)
        assert any(k.startswith("se_gate") for k in prof16.rows) and any(k.startswith("se_gate") for k in prof11.rows)
    finally:
        E.PW_MIN_PIXELS, E.PW_EXPAND_LIB_BELOW, E.PW_PROJECT_K16_MIN_PIXELS, E.PW_PROJECT_K16_MIN_COUT = saved
    for name, got in (("K16", got16), ("K11", got11)):
        err = float((got.double().cpu() - ref).abs().max() / ref.abs().max())
        assert got.shape == ref.shape and err < 2e-5, (kind, name, err)


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(2, 80, 64, 37, 61), (1, 163, 30, 9, 70)])
def test_pointwise_conv_nhwc_output_feeds_the_lift_without_a_transpose(shape, hip_lib):
    """K11's pixel-major mode: same numbers as the NCHW mode, returned as a channels-last view whose pixel rows
    `lift_scales` consumes in place."""
    from occdepth_amd import hip
    B, cin, cout, H, W = shape
    g = torch.Generator().manual_seed(cin)
    x = torch.randn(B, cin, H, W, generator=g).cuda()
    w = (torch.randn(cout, cin, 1, 1, generator=g) * 0.1).cuda()
    shift = torch.randn(cout, generator=g).cuda()
    wpk = hip.pw_pack_weights(w)
    for hint in (0, 1, 2, 6):
        a = hip.conv1x1(x, wpk, cout, shift, tile_hint=hint)
        b = hip.conv1x1(x, wpk, cout, shift, tile_hint=hint, nhwc=True)
        assert b.shape == a.shape and b.permute(0, 2, 3, 1).stride(3) == 1
        assert torch.allclose(a, b, rtol=1e-6, atol=1e-6)
        rows = b.permute(0, 2, 3, 1)
        cs = rows.stride(2)
        assert cs % 4 == 0 and cs >= cout
        if cs > cout:                                        # the channel pad of every pixel row is zero
            full = rows.as_strided((B, H, W, cs), rows.stride()[:3] + (1,))
            assert float(full[..., cout:].abs().max()) == 0.0


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(2, 104, 47, 153), (1, 7, 5, 9), (3, 128, 3, 70), (1, 130, 6, 11)])
def test_softmax_nchw_gpu(shape, hip_lib):
    """depth-bin softmax: the channel-split kernel (C <= 128) and the generic one."""
    from occdepth_amd import hip
    x = torch.randn(*shape, generator=torch.Generator().manual_seed(1)) * 5
    y = hip.softmax_nchw(x.cuda())
    ref = torch.softmax(x.double(), 1)
    assert float((y.double().cpu() - ref).abs().max()) < 5e-6      # float32 exp + a 104-term sum against float64


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(2, 24, 37, 53, 3, 1), (1, 40, 30, 41, 5, 2), (2, 7, 9, 300, 5, 1), (1, 16, 24, 77, 3, 2)])
def test_depthwise_autograd_gpu(shape, hip_lib):
    """Depthwise SAME convolution: HIP forward / data gradient / weight gradient against ATen float64 on the CPU."""
    import torch.nn.functional as F
    from occdepth_amd import hip
    B, C, H, W, k, stride = shape
    g = torch.Generator().manual_seed(C + W)
    x = torch.randn(B, C, H, W, generator=g)
    w = torch.randn(C, 1, k, k, generator=g) * 0.3
    Ho, Wo = -(-H // stride), -(-W // stride)
    R = torch.randn(B, C, Ho, Wo, generator=g)
    xd, wd = x.double().requires_grad_(True), w.double().requires_grad_(True)
    ph, pw_ = max((Ho - 1) * stride + k - H, 0), max((Wo - 1) * stride + k - W, 0)
    ref = F.conv2d(F.pad(xd, [pw_ // 2, pw_ - pw_ // 2, ph // 2, ph - ph // 2]), wd, None, stride, 0, 1, C)
    (ref * R.double()).sum().backward()
    xc, wc = x.cuda().requires_grad_(True), w.cuda().requires_grad_(True)
    y = hip.dwconv2d_same_autograd(xc, wc, stride)
    (y * R.cuda()).sum().backward()
    for got, want in ((y, ref), (xc.grad, xd.grad), (wc.grad, wd.grad)):
        assert got.shape == want.shape
        assert float((got.double().cpu() - want.detach()).abs().max() / want.detach().abs().max()) < 2e-5


@pytest.mark.gpu
def test_merged_head_matches_two_gemms_gpu(hip_lib):
    """Eval path: conv_head (encoder) folded into the decoder's conv2 -- one 1x1 convolution with W = W_conv2 . W_head,
    conv2's bias and its padding=1 frame -- against the unmerged module path of the same network."""
    from occdepth_amd.models.unet2d import UNet2D
    torch.manual_seed(3)
    m = UNet2D.build(out_feature=16, use_decoder=True, backbone_2d_name="tf_efficientnet_b3_ns", return_up_feats=1)
    for mod in m.modules():
        if isinstance(mod, torch.nn.BatchNorm2d):
            mod.running_mean.normal_(0, 0.1)
            mod.running_var.uniform_(0.7, 1.3)
    m = m.cuda().eval()
    x = torch.randn(2, 3, 70, 122, device="cuda")
    saved = UNet2D.MERGE_HEAD
    try:
        with torch.no_grad():
            UNet2D.MERGE_HEAD = False
            ref = m(x)
            UNet2D.MERGE_HEAD = True
            feats = m.encoder(x, skip_head=True)
            assert feats[11] is None and feats[12] is None and feats[10] is not None
            got = m(x)
    finally:
        UNet2D.MERGE_HEAD = saved
    assert set(got) == set(ref)
    for k in ref:
        err = float((got[k] - ref[k]).abs().max() / ref[k].abs().max())
        assert got[k].shape == ref[k].shape and err < 1e-4, (k, err)
