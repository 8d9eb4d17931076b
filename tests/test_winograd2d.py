"""Winograd F(2x2, 3x3) path of the 2-D decoder convolutions (N3): weight transform + tile conventions on the CPU
(test-only emulation of the two transform kernels), the HIP kernels themselves with `-m gpu`."""
import contextlib

import pytest
import torch
import torch.nn.functional as F

import emu

# (B, Cin, Cout, H, W, act, residual)
CASES = [(2, 40, 24, 12, 39, "leaky", False),      # the 1/16 level's odd width
         (1, 33, 70, 7, 5, None, False),           # ragged channels, odd sizes, fewer than one workgroup of tiles
         (2, 64, 64, 24, 77, "relu", True),        # BasicBlock: relu(bn(conv) + x)
         (1, 8, 8, 2, 130, "leaky", False)]        # more than one workgroup along x


def run(case, device, tol, strip_rows=None):
    from occdepth_amd import hip
    B, cin, cout, H, W, act, with_res = case
    g = torch.Generator().manual_seed(B * 1000 + cin + H)
    x = torch.randn(B, cin, H, W, generator=g)
    w = torch.randn(cout, cin, 3, 3, generator=g) * 0.1
    scale, shift = torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g)
    res = torch.randn(B, cout, H, W, generator=g) if with_res else None
    ref = F.conv2d(x.double(), w.double(), padding=1) * scale.double().view(1, -1, 1, 1) + shift.double().view(1, -1, 1, 1)
    if with_res:
        ref = ref + res.double()
    ref = F.leaky_relu(ref, 0.01) if act == "leaky" else (F.relu(ref) if act == "relu" else ref)
    dev = torch.device(device)
    with (emu.patched() if device == "cpu" else contextlib.nullcontext()):
        U = hip.winograd_weights(w.to(dev))
        y = hip.conv2d_3x3_winograd(x.to(dev), U, scale.to(dev), shift.to(dev), act, 0.01,
                                    res.to(dev) if with_res else None, res_first=True, strip_rows=strip_rows)
    err = float((y.double().cpu() - ref).abs().max() / ref.abs().max())
    assert y.shape == ref.shape and err < tol, (case, err)


@pytest.mark.parametrize("case", CASES)
def test_winograd_host_logic_cpu(case):
    run(case, "cpu", 2e-6)
    run(case, "cpu", 2e-6, strip_rows=3)          # strips of 3 tile rows (ragged last strip)


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES)
def test_winograd_kernels_gpu(case, hip_lib):
    torch.backends.cuda.matmul.allow_tf32 = False
    run(case, "cuda", 2e-5)
    run(case, "cuda", 2e-5, strip_rows=3)


@pytest.mark.gpu
def test_transform_kernels_match_emulation(hip_lib):
    from occdepth_amd import hip
    g = torch.Generator().manual_seed(9)
    x = torch.randn(2, 45, 13, 71, generator=g)
    V = hip.wino_input_transform(x.cuda())
    assert torch.allclose(V.cpu(), emu.wino_input_transform(x), rtol=1e-6, atol=1e-6)
    M = torch.randn(16, 2 * 7 * 36, 50, generator=g)
    res = torch.randn(2, 50, 13, 71, generator=g)
    sc, sh = torch.rand(50, generator=g), torch.randn(50, generator=g)
    for rf in (True, False):
        y = hip.wino_output_transform(M.cuda(), (2, 50, 13, 71), sc.cuda(), sh.cuda(), "leaky", 0.2, res.cuda(), rf)
        assert torch.allclose(y.cpu(), emu.wino_output_transform(M, (2, 50, 13, 71), sc, sh, "leaky", 0.2, res, rf),
                              rtol=1e-5, atol=1e-5)


# ---------------------------------------------------------------------------------------------------------------
# K10: the fused kernel (csrc/wino_conv2d.hip) -- V in registers, M in the MFMA accumulators, output transform in the
# epilogue.  Reference: ATen float64 conv2d + affine + activation (+ residual) on the CPU.
# (B, Cin, Cout, H, W, act, residual, res_first, tile_hint)
FUSED_CASES = [
    (2, 40, 24, 12, 39, "leaky", False, False, 0),       # odd width, cout < 32, cin = 5 chunks
    (1, 33, 70, 7, 5, None, False, False, 0),            # ragged channels (cin % 8 != 0, 3 cout blocks), tiny image
    (2, 64, 64, 24, 77, "relu", True, True, 16),         # BasicBlock: relu(bn(conv) + x), narrow waves
    (2, 64, 64, 24, 77, "relu", True, False, 32),        # residual after the activation, wide waves
    (1, 8, 8, 2, 130, "leaky", False, False, 0),         # several workgroups along x, one tile row
    (1, 19, 40, 37, 68, "leaky", False, False, 16),      # even width (paired stores), several workgroups along y
    (1, 19, 40, 37, 68, "swish", False, False, 32),
    (2, 163, 80, 50, 131, "leaky", False, False, 0),     # the up1 channel counts on a crop
]


def run_fused(case, device, tol):
    from occdepth_amd import hip
    B, cin, cout, H, W, act, with_res, res_first, hint = case
    g = torch.Generator().manual_seed(B * 1000 + cin + H + hint)
    x = torch.randn(B, cin, H, W, generator=g)
    w = torch.randn(cout, cin, 3, 3, generator=g) * 0.1
    scale, shift = torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g)
    res = torch.randn(B, cout, H, W, generator=g) if with_res else None
    ref = F.conv2d(x.double(), w.double(), padding=1) * scale.double().view(1, -1, 1, 1) + shift.double().view(1, -1, 1, 1)
    if with_res and res_first:
        ref = ref + res.double()
    ref = {"leaky": lambda t: F.leaky_relu(t, 0.01), "relu": F.relu, "swish": lambda t: t * torch.sigmoid(t),
           None: lambda t: t}[act](ref)
    if with_res and not res_first:
        ref = ref + res.double()
    dev = torch.device(device)
    with (emu.patched() if device == "cpu" else contextlib.nullcontext()):
        upk = hip.wino_pack_weights(w.to(dev), scale.to(dev))
        y = hip.conv2d_3x3_fused(x.to(dev), upk, cout, shift.to(dev), act, 0.01, res.to(dev) if with_res else None,
                                 res_first=res_first, tile_hint=hint)
    err = float((y.double().cpu() - ref).abs().max() / ref.abs().max())
    assert y.shape == ref.shape and err < tol, (case, err)
    return err


@pytest.mark.parametrize("case", FUSED_CASES)
def test_fused_winograd_host_logic_cpu(case):
    run_fused(case, "cpu", 2e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("case", FUSED_CASES)
def test_fused_winograd_kernel_gpu(case, hip_lib):
    err = run_fused(case, "cuda", 2e-5)
    print(case, f"rel err vs float64 {err:.2e}")


@pytest.mark.gpu
def test_fused_winograd_is_deterministic_and_ignores_garbage_outside(hip_lib):
    """Same launch twice = same bits; pixels outside the image never leak in (the staged patch is zero padded)."""
    from occdepth_amd import hip
    g = torch.Generator().manual_seed(4)
    x = torch.randn(1, 24, 9, 35, generator=g).cuda()
    w = (torch.randn(32, 24, 3, 3, generator=g) * 0.1).cuda()
    upk = hip.wino_pack_weights(w)
    a = hip.conv2d_3x3_fused(x, upk, 32)
    b = hip.conv2d_3x3_fused(x, upk, 32)
    assert torch.equal(a, b)
    big = torch.full((1, 24, 11, 37), float("nan"), device="cuda")     # the same image inside a NaN frame ...
    big[:, :, 1:10, 1:36] = x
    ref = F.conv2d(x.double(), w.double(), padding=1)
    assert float((a.double() - ref).abs().max() / ref.abs().max()) < 2e-5


# ---------------------------------------------------------------------------------------------------------------
# differentiable form (training step): forward and data gradient on K10, weight / bias gradient on the backend
GRAD_CASES = [(2, 19, 12, 13, 21, True), (1, 40, 35, 9, 30, False), (1, 8, 8, 4, 6, True)]      # B, Cin, Cout, H, W, bias


def _conv3x3_grads(case, device):
    from occdepth_amd import hip
    B, cin, cout, H, W, bias = case
    g = torch.Generator().manual_seed(cin * 100 + cout)
    x = torch.randn(B, cin, H, W, generator=g)
    w = torch.randn(cout, cin, 3, 3, generator=g) * (1.0 / (3.0 * cin ** 0.5))
    b = torch.randn(cout, generator=g) if bias else None
    gy = torch.randn(B, cout, H, W, generator=g)
    ref_in = [t.double().requires_grad_(True) for t in (x, w)] + ([b.double().requires_grad_(True)] if bias else [])
    ref = torch.nn.functional.conv2d(ref_in[0], ref_in[1], ref_in[2] if bias else None, padding=1)
    ref_g = torch.autograd.grad(ref, ref_in, gy.double())
    ins = [t.to(device).requires_grad_(True) for t in (x, w)] + ([b.to(device).requires_grad_(True)] if bias else [])
    y = hip.conv2d_3x3_autograd(ins[0], ins[1], ins[2] if bias else None)
    got_g = torch.autograd.grad(y, ins, gy.to(device))
    worst = float((y.detach().double().cpu() - ref.detach()).abs().max() / ref.detach().abs().max())
    for a, r in zip(got_g, ref_g):
        worst = max(worst, float((a.double().cpu() - r).abs().max() / r.abs().max()))
    return worst


@pytest.mark.parametrize("case", GRAD_CASES)
def test_conv3x3_autograd_host_logic_cpu(case):
    with emu.patched():
        assert _conv3x3_grads(case, "cpu") < 1e-5


@pytest.mark.gpu
@pytest.mark.parametrize("case", GRAD_CASES)
def test_conv3x3_autograd_gpu(case, hip_lib):
    err = _conv3x3_grads(case, "cuda")
    print(case, f"forward / dx / dw / db worst rel err vs float64 {err:.2e}")
    assert err < 2e-5


# ---------------------------------------------------------------------------------------------------------------
# conv3x3(bilinear_up(x)) as nine low-resolution tap GEMMs + upsample-shift-accumulate (upconv_gather_kernel)
def _upconv_reference(z, cout, size):
    """sum_t shift_t(bilinear_up(z_t)) with zero padding, float64, by ATen ops."""
    B = z.shape[0]
    H, W = size
    out = torch.zeros(B, cout, H, W, dtype=torch.float64)
    for ky in range(3):
        for kx in range(3):
            t = ky * 3 + kx
            up = F.interpolate(z[:, t * cout:(t + 1) * cout].double(), size=size, mode="bilinear", align_corners=True)
            out += F.pad(up, (1, 1, 1, 1))[:, :, ky:ky + H, kx:kx + W]           # out[p] += up[p + (ky - 1, kx - 1)]
    return out


@pytest.mark.gpu
@pytest.mark.parametrize("case", [(2, 5, 7, 11, 13, 22), (1, 3, 12, 39, 24, 77), (1, 2, 9, 8, 9, 8), (1, 4, 3, 5, 17, 130),
                                  (1, 2, 15, 20, 30, 40), (1, 2, 30, 40, 60, 80), (1, 1, 93, 305, 185, 610)])
def test_upconv_gather_kernel_gpu(case, hip_lib):
    from occdepth_amd import hip
    B, cout, h, w, H, W = case
    g = torch.Generator().manual_seed(h * 100 + W)
    z = torch.randn(B, 9 * cout, h, w, generator=g)
    got = hip.upconv_gather(z.cuda(), cout, (H, W)).cpu()
    ref = _upconv_reference(z, cout, (H, W))
    err = float((got.double() - ref).abs().max() / ref.abs().max())
    # the source coordinates are float32 products (as in ATen's float32 kernel): at column 600 one ulp of the coordinate
    # is 3e-5 of a pixel, which is the error of the interpolation weight against this float64 reference
    assert got.shape == ref.shape and err < (1e-5 if W < 200 else 6e-5), (case, err)
    # the tap planes on a 128-byte pitch (hip.padded_rows: how the decoder hands them over) and batch-inner: the strides travel,
    # the result is bit-identical and the padding is never read (NaN there)
    zp = hip.padded_rows((B, 9 * cout, h * w), "cuda")
    assert zp.stride(1) % 32 == 0 and zp.stride(2) == 1 and zp.stride(1) >= h * w
    torch.as_strided(zp, (B, 9 * cout, zp.stride(1)), (zp.stride(0), zp.stride(1), 1)).fill_(float("nan"))
    zp.copy_(z.cuda().view(B, 9 * cout, h * w))
    assert torch.equal(hip.upconv_gather(zp.view(B, 9 * cout, h, w), cout, (H, W)).cpu(), got)
    zi = hip.padded_rows((9 * cout, B * h * w), "cuda")
    torch.as_strided(zi, (9 * cout, zi.stride(0)), (zi.stride(0), 1)).fill_(float("nan"))
    zi.copy_(z.cuda().permute(1, 0, 2, 3).reshape(9 * cout, B * h * w))
    assert torch.equal(hip.upconv_gather(zi.view(9 * cout, B, h, w), cout, (H, W), batch_inner=True).cpu(), got)


@pytest.mark.gpu
@pytest.mark.parametrize("case", [(2, 40, 9, 16, 7, 9, 13, 18), (1, 24, 3, 8, 6, 20, 12, 40), (2, 16, 3, 8, 40, 150, 81, 301)])
def test_upsample_block_upconv_matches_module_gpu(case, hip_lib):
    """A whole decoder level (UpSampleBN, eval): the tap-GEMM form of its first convolution against the module's own
    float64 CPU forward (interpolate -> cat -> conv -> BN -> LeakyReLU, twice), and against the upsample+concat form."""
    import copy
    from occdepth_amd.models.unet2d import UpSampleBN
    B, cup, cs, cout, h, w, H, W = case
    torch.manual_seed(cup + cout)
    m = UpSampleBN(cup + cs, cout)
    for mod in m.modules():
        if isinstance(mod, torch.nn.BatchNorm2d):
            mod.running_mean.normal_(0, 0.2)
            mod.running_var.uniform_(0.5, 1.5)
            mod.weight.data.uniform_(0.5, 1.5)
            mod.bias.data.normal_(0, 0.2)
    m.eval()
    x, skip = torch.randn(B, cup, h, w), torch.randn(B, cs, H, W)
    saved = (UpSampleBN.UPCONV, UpSampleBN.FUSED_MIN_PIXELS, UpSampleBN.UPCONV_LIB_BELOW, UpSampleBN.UPCONV_FOLD_BELOW)
    try:
        with torch.no_grad():
            ref = copy.deepcopy(m).double()(x.double(), skip.double())
            mc = m.cuda()
            UpSampleBN.FUSED_MIN_PIXELS = 0
            outs = {}
            for name, on, lib_below, fold in (("upconv_k11", True, 0, 0), ("upconv_lib", True, 1 << 62, 0),
                                              ("upconv_fold", True, 1 << 62, 1 << 62), ("concat", False, 0, 0)):
                UpSampleBN.UPCONV, UpSampleBN.UPCONV_LIB_BELOW, UpSampleBN.UPCONV_FOLD_BELOW = on, lib_below, fold
                outs[name] = mc(x.cuda(), skip.cuda()).double().cpu()
    finally:
        UpSampleBN.UPCONV, UpSampleBN.FUSED_MIN_PIXELS, UpSampleBN.UPCONV_LIB_BELOW, UpSampleBN.UPCONV_FOLD_BELOW = saved
    for name, got in outs.items():
        err = float((got - ref).abs().max() / ref.abs().max())
        print(case, name, f"{err:.2e}")
        assert got.shape == ref.shape and err < 3e-5, (name, err)


@pytest.mark.gpu
def test_swish_and_upsample_cat_autograd_gpu(hip_lib):
    """Training-path Functions: fused swish (forward + one-pass backward) and upsample+concat (HIP forward, ATen backward
    on the gradient slice) against the autograd graphs they replace, in float64 on the CPU."""
    from occdepth_amd import hip
    g = torch.Generator().manual_seed(9)
    x = torch.randn(2, 7, 5, 9, generator=g) * 3
    gy = torch.randn(2, 7, 5, 9, generator=g)
    xr = x.double().requires_grad_(True)
    (xr * torch.sigmoid(xr)).backward(gy.double())
    xc = x.cuda().requires_grad_(True)
    y = hip.swish_autograd(xc)
    y.backward(gy.cuda())
    ref_y = (x.double() * torch.sigmoid(x.double()))
    assert float((y.detach().double().cpu() - ref_y).abs().max()) < 1e-5
    assert float((xc.grad.double().cpu() - xr.grad).abs().max() / xr.grad.abs().max()) < 1e-5
    a, s = torch.randn(2, 6, 4, 7, generator=g), torch.randn(2, 3, 9, 13, generator=g)
    go = torch.randn(2, 9, 9, 13, generator=g)
    ar, sr = a.double().requires_grad_(True), s.double().requires_grad_(True)
    ref = torch.cat([F.interpolate(ar, size=(9, 13), mode="bilinear", align_corners=True), sr], 1)
    ref.backward(go.double())
    ac, sc = a.cuda().requires_grad_(True), s.cuda().requires_grad_(True)
    out = hip.upsample_bilinear_cat_autograd(ac, sc)
    out.backward(go.cuda())
    assert float((out.detach().double().cpu() - ref.detach()).abs().max()) < 1e-5
    assert float((ac.grad.double().cpu() - ar.grad).abs().max()) < 1e-5
    assert torch.equal(sc.grad.cpu(), go[:, 6:])


# ---------------------------------------------------------------------------------------------------------------
# host algebra of the decoder's eval path on the CPU (kernels emulated): tap-GEMM split of the first convolution,
# BatchNorm folding, skip-channel convolution with the gathered part as residual; conv_head folded into conv2
@pytest.mark.parametrize("case", [(2, 24, 5, 8, 6, 9, 11, 17), (1, 16, 3, 12, 4, 5, 8, 10)])
def test_upconv_host_algebra_cpu(case):
    import copy
    from occdepth_amd.models.unet2d import UpSampleBN
    B, cup, cs, cout, h, w, H, W = case
    torch.manual_seed(cup * 7 + cout)
    m = UpSampleBN(cup + cs, cout)
    for mod in m.modules():
        if isinstance(mod, torch.nn.BatchNorm2d):
            mod.running_mean.normal_(0, 0.2)
            mod.running_var.uniform_(0.5, 1.5)
            mod.weight.data.uniform_(0.5, 1.5)
            mod.bias.data.normal_(0, 0.2)
    m.eval()
    x, skip = torch.randn(B, cup, h, w), torch.randn(B, cs, H, W)
    n = m._net
    with torch.no_grad():
        up = F.interpolate(x.double(), size=(H, W), mode="bilinear", align_corners=True)
        ref = copy.deepcopy(n[:3]).double()(torch.cat([up, skip.double()], 1))       # conv -> BN -> LeakyReLU
        saved = UpSampleBN.UPCONV_LIB_BELOW
        try:
            with emu.patched():
                for lib in (0, 1 << 62):                      # tap GEMM through the K11 wrapper / through torch.matmul
                    UpSampleBN.UPCONV_LIB_BELOW = lib
                    got = m._first_conv_upconv(x, skip, n[0], n[1], n[2])
                    err = float((got.double() - ref).abs().max() / ref.abs().max())
                    assert got.shape == ref.shape and err < 2e-5, (lib, err)
        finally:
            UpSampleBN.UPCONV_LIB_BELOW = saved


def test_merged_head_algebra_cpu():
    from occdepth_amd.models.unet2d import DecoderBN
    torch.manual_seed(4)
    dec = DecoderBN(num_features=64, bottleneck_features=64, out_feature=8, use_decoder=True,
                    backbone_2d_name="tf_efficientnet_b3_ns", return_up_feats=1)
    head = torch.nn.Conv2d(24, 64, 1, bias=False)
    f = torch.randn(2, 24, 5, 7)
    with torch.no_grad():
        ref = dec.conv2(head(f))
        got = dec._conv2_merged(f, head)
    assert got.shape == ref.shape == (2, 64, 7, 9)
    assert float((got - ref).abs().max() / ref.abs().max()) < 1e-5


@pytest.mark.gpu
def test_upconv_gather_shared_source_rows_variant_in_child_process():
    """K12's COOP form (round 6: the four waves of a workgroup share their source rows; measured slower and therefore opt-in,
    OCCD_UPCONV_COOP=1 -- the switch is read once per process) must stay correct: the upconv tests of this file again, in a
    child process with the switch on."""
    import os
    import subprocess
    import sys
    if os.environ.get("OCCD_UPCONV_COOP") == "1":
        pytest.skip("already inside the child")
    env = dict(os.environ, OCCD_UPCONV_COOP="1")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-m", "gpu", "-k", "upconv and not child",
                        "-p", "no:cacheprovider"], env=env, capture_output=True, text=True, timeout=600,
                       cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    tail = "\n".join((r.stdout + r.stderr).splitlines()[-15:])
    assert r.returncode == 0 and " passed" in tail, tail
