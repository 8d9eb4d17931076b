"""Differentiable 3-D convolutions (occdepth_amd/autograd3d.py): forward, data gradient (flipped / sub-pixel-phase
convolutions through the forward kernel) and weight gradient against ATen's float64 autograd on the CPU.
CPU tests run the host logic (phase decomposition, weight slicing, scatter geometry) through the test-only
emulation; the `-m gpu` tests run the HIP kernels."""
import pytest
import torch
import torch.nn.functional as F

import emu
from conv_grad_cases import CONV_CASES, CONVT_CASES, conv_tensors, convt_tensors
from occdepth_amd import autograd3d as ag


def reference_conv(name):
    cin, cout, k, s, p, d, dims, bias = CONV_CASES[name]
    x, w, b = conv_tensors(name)
    xd, wd = x.double().requires_grad_(True), w.double().requires_grad_(True)
    bd = b.double().requires_grad_(True) if b is not None else None
    y = F.conv3d(xd, wd, bd, stride=s, padding=p, dilation=d)
    gy = torch.randn(y.shape, generator=torch.Generator().manual_seed(7), dtype=torch.float64)
    y.backward(gy)
    return y.detach(), gy, xd.grad, wd.grad, (bd.grad if bd is not None else None)


def reference_convt(name):
    cin, cout, k, s, p, op, dims, bias = CONVT_CASES[name]
    x, w, b = convt_tensors(name)
    xd, wd = x.double().requires_grad_(True), w.double().requires_grad_(True)
    bd = b.double().requires_grad_(True) if b is not None else None
    y = F.conv_transpose3d(xd, wd, bd, stride=s, padding=p, output_padding=op)
    gy = torch.randn(y.shape, generator=torch.Generator().manual_seed(7), dtype=torch.float64)
    y.backward(gy)
    return y.detach(), gy, xd.grad, wd.grad, (bd.grad if bd is not None else None)


def close(got, want, tol, what):
    err = float((got.detach().double().cpu() - want).abs().max() / want.abs().max().clamp_min(1e-30))
    assert err < tol, (what, err)


def run_conv(name, device, tol):
    cin, cout, k, s, p, d, dims, bias = CONV_CASES[name]
    y_ref, gy, dx_ref, dw_ref, db_ref = reference_conv(name)
    x, w, b = conv_tensors(name)
    x = x.to(device).requires_grad_(True)
    w = w.to(device).requires_grad_(True)
    b = b.to(device).requires_grad_(True) if b is not None else None
    y = ag._Conv3dFn.apply(x, w, b, s, p, d)
    close(y, y_ref, tol, name + ".y")
    y.backward(gy.float().to(device))
    close(x.grad, dx_ref, tol, name + ".dx")
    close(w.grad, dw_ref, tol, name + ".dw")
    if b is not None:
        close(b.grad, db_ref, tol, name + ".db")


def run_convt(name, device, tol):
    cin, cout, k, s, p, op, dims, bias = CONVT_CASES[name]
    y_ref, gy, dx_ref, dw_ref, db_ref = reference_convt(name)
    x, w, b = convt_tensors(name)
    x = x.to(device).requires_grad_(True)
    w = w.to(device).requires_grad_(True)
    b = b.to(device).requires_grad_(True) if b is not None else None
    y = ag._ConvTranspose3dFn.apply(x, w, b, s, p, op, (1, 1, 1))
    close(y, y_ref, tol, name + ".y")
    y.backward(gy.float().to(device))
    close(x.grad, dx_ref, tol, name + ".dx")
    close(w.grad, dw_ref, tol, name + ".dw")
    if b is not None:
        close(b.grad, db_ref, tol, name + ".db")


@pytest.mark.parametrize("name", list(CONV_CASES))
def test_conv3d_host_logic_cpu(name):
    with emu.patched():
        run_conv(name, "cpu", 2e-6)


@pytest.mark.parametrize("name", list(CONVT_CASES))
def test_conv_transpose3d_host_logic_cpu(name):
    with emu.patched():
        run_convt(name, "cpu", 2e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(CONV_CASES))
def test_conv3d_kernels_gpu(name, hip_lib):
    run_conv(name, "cuda", 2e-5)


# bf16 mode (autograd3d.BF16_MFMA, BASELINE configs[3]): the same Function classes on K2b / K8b.  Against the float64
# reference of the UN-rounded operands the error is bf16's: every product carries two 2^-9 roundings, a sum of K of them
# ~ 2^-8 / sqrt(K) of the result's scale in the mean and a few times that in the maximum: bound 1.5e-2 of the tensor's
# scale (the kernels themselves are pinned at 2e-5 on bf16-rounded operands in test_bf16_conv.py).
@pytest.mark.parametrize("name", list(CONV_CASES))
def test_conv3d_bf16_mode_host_logic_cpu(name):
    old = ag.set_bf16_mfma(True)
    try:
        with emu.patched():
            run_conv(name, "cpu", 1.5e-2)
    finally:
        ag.set_bf16_mfma(old)


@pytest.mark.parametrize("name", list(CONVT_CASES))
def test_conv_transpose3d_bf16_mode_host_logic_cpu(name):
    old = ag.set_bf16_mfma(True)
    try:
        with emu.patched():
            run_convt(name, "cpu", 1.5e-2)
    finally:
        ag.set_bf16_mfma(old)


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(CONV_CASES))
def test_conv3d_bf16_mode_kernels_gpu(name, hip_lib):
    old = ag.set_bf16_mfma(True)
    try:
        run_conv(name, "cuda", 1.5e-2)
    finally:
        ag.set_bf16_mfma(old)


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(CONVT_CASES))
def test_conv_transpose3d_bf16_mode_kernels_gpu(name, hip_lib):
    old = ag.set_bf16_mfma(True)
    try:
        run_convt(name, "cuda", 1.5e-2)
    finally:
        ag.set_bf16_mfma(old)


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(CONVT_CASES))
def test_conv_transpose3d_kernels_gpu(name, hip_lib):
    run_convt(name, "cuda", 2e-5)


@pytest.mark.gpu
def test_modules_route_to_hip_and_match_aten_modules(hip_lib):
    """The nn.Module subclasses: same state_dict as nn.Conv3d / nn.ConvTranspose3d, HIP path on CUDA fp32,
    channels_last_3d outputs chained without a transpose, weight gradient deterministic."""
    from occdepth_amd import hip
    torch.manual_seed(0)
    a = ag.Conv3d(16, 32, 3, padding=2, dilation=2).cuda()
    t = ag.ConvTranspose3d(32, 16, 3, stride=2, padding=1, output_padding=1).cuda()
    ref_a, ref_t = torch.nn.Conv3d(16, 32, 3, padding=2, dilation=2), torch.nn.ConvTranspose3d(32, 16, 3, 2, 1, 1)
    ref_a.load_state_dict(a.state_dict())
    ref_t.load_state_dict(t.state_dict())
    x = torch.randn(1, 16, 8, 8, 8)
    with hip.profile() as prof:
        y = t(a(x.cuda()))
        y.square().sum().backward()
    kinds = {k.split(":")[0] for k in prof.rows}
    assert {"conv3d_igemm", "conv3d_wgrad"} <= kinds, kinds
    assert y.permute(0, 2, 3, 4, 1).is_contiguous()
    yr = ref_t.double()(ref_a.double()(x.double()))
    yr.square().sum().backward()
    close(y, yr.detach(), 2e-5, "y")
    close(a.weight.grad, ref_a.weight.grad, 2e-5, "dw conv")
    close(t.weight.grad, ref_t.weight.grad, 2e-5, "dw convT")
    close(a.bias.grad, ref_a.bias.grad, 2e-5, "db")
    g1 = a.weight.grad.clone()
    a.zero_grad()
    t.zero_grad()
    t(a(x.cuda())).square().sum().backward()
    assert torch.equal(a.weight.grad, g1)


@pytest.mark.gpu
def test_wgrad_full_size_head_properties(hip_lib):
    """256x256x32, 32 -> 32, 3x3x3: linearity in gy and agreement of a few taps with plain reductions."""
    from occdepth_amd import hip
    from occdepth_amd.hip import Vox
    g = torch.Generator(device="cuda").manual_seed(3)
    dims = (256, 256, 32)
    x = Vox(torch.randn(1, *dims, 32, device="cuda", generator=g), 32)
    gy = Vox(torch.randn(1, *dims, 32, device="cuda", generator=g), 32)
    dw = hip.conv3d_wgrad(x, gy, 32, 32, (3, 3, 3), padding=(1, 1, 1))
    assert torch.equal(dw, hip.conv3d_wgrad(x, gy, 32, 32, (3, 3, 3), padding=(1, 1, 1)))
    gy2 = Vox(gy.buf * 2.0, 32)
    assert torch.allclose(hip.conv3d_wgrad(x, gy2, 32, 32, (3, 3, 3), padding=(1, 1, 1)), 2 * dw, rtol=1e-6, atol=1e-3)
    xf, gf = x.buf[0].double(), gy.buf[0].double()
    centre = torch.einsum("xyzo,xyzi->oi", gf, xf)
    corner = torch.einsum("xyzo,xyzi->oi", gf[1:, 1:, 1:], xf[:-1, :-1, :-1])          # tap (0,0,0): x[v - 1]
    scale = centre.abs().max()
    assert float((dw[:, :, 1, 1, 1].double() - centre).abs().max() / scale) < 2e-5
    assert float((dw[:, :, 0, 0, 0].double() - corner).abs().max() / scale) < 2e-5


@pytest.mark.gpu
def test_autocast_and_bf16_mode_dtypes(hip_lib):
    """Under torch.autocast the HIP convolutions do no casting of their own.  Exact mode: a bf16 activation is widened and the
    result is the exact-fp32 convolution of those values; bf16 mode (BASELINE configs[3]): a bf16 activation stays bf16 in HBM
    and comes back bf16, a float32 one stays float32 (bf16 MFMA, fp32 storage); weights and their gradients are float32 in
    every case."""
    from occdepth_amd.loss import ssc_loss
    torch.manual_seed(1)
    conv = ag.Conv3d(16, 20, 3, padding=1).cuda()
    x = torch.randn(1, 16, 8, 8, 8, device="cuda", requires_grad=True)
    target = torch.randint(0, 20, (1, 8, 8, 8), device="cuda").to(torch.uint8)
    xb = x.detach().to(torch.bfloat16)
    ref = F.conv3d(xb.float(), conv.weight.detach(), conv.bias.detach(), padding=1)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y = conv(x.to(torch.bfloat16))
        assert y.dtype == torch.float32
        loss = ssc_loss.CE_ssc_loss(y.to(torch.bfloat16), target, torch.ones(20, device="cuda"))
    loss.backward()
    close(y, ref.double().cpu(), 2e-5, "autocast y")
    assert x.grad is not None and conv.weight.grad.dtype == torch.float32 and torch.isfinite(conv.weight.grad).all()
    old = ag.set_bf16_mfma(True)
    try:
        refb = F.conv3d(xb.float(), conv.weight.detach().to(torch.bfloat16).float(), conv.bias.detach(), padding=1)
        conv.zero_grad()
        yb = conv(xb.clone().requires_grad_(True))
        assert yb.dtype == torch.bfloat16
        close(yb.float(), refb.double().cpu(), 6e-3, "bf16 storage y")
        yf = conv(xb.float().requires_grad_(True))
        assert yf.dtype == torch.float32
        close(yf, refb.double().cpu(), 2e-5, "bf16 mfma / fp32 storage y")
        (yb.float().square().sum() + yf.square().sum()).backward()
        assert conv.weight.grad.dtype == torch.float32 and torch.isfinite(conv.weight.grad).all()
    finally:
        ag.set_bf16_mfma(old)


@pytest.mark.gpu
@pytest.mark.parametrize("d", [1, 2, 3])
def test_head_conv_forward_and_dgrad_take_k2s3_in_fp32_training_mode(d, hip_lib):
    """fp32 training mode: the full-resolution head convolutions run forward AND data gradient on K2s3 (3-way bf16 split,
    float32-level accuracy) when the eval default does (fused.BF16X3 = "head"); against ATen float64 on the CPU at the bound of
    the exact-fp32 kernels (2e-5), and the profile shows which kernel ran."""
    import torch.nn as nn
    from occdepth_amd import autograd3d, fused, hip
    assert fused.BF16X3 == "head" and not autograd3d.BF16_MFMA
    torch.manual_seed(d)
    conv = autograd3d.Conv3d(32, 32, 3, padding=d, dilation=d, bias=False).cuda()
    x = torch.randn(1, 32, 16, 256, 32, device="cuda").contiguous(memory_format=torch.channels_last_3d).requires_grad_(True)
    gy = torch.randn(1, 32, 16, 256, 32, device="cuda").contiguous(memory_format=torch.channels_last_3d)
    with hip.profile() as prof:
        y = conv(x)
        y.backward(gy)
        torch.cuda.synchronize()
    tags = sorted({k.split(":")[0] for k in prof.rows})
    assert sum(v["launches"] for k, v in prof.rows.items() if k.startswith("conv3d_c32x3")) == 2, tags     # forward + dgrad
    xr = x.detach().cpu().double().requires_grad_(True)
    wr = conv.weight.detach().cpu().double().requires_grad_(True)
    yr = torch.nn.functional.conv3d(xr, wr, padding=d, dilation=d)
    yr.backward(gy.cpu().double())
    for got, ref, what in ((y, yr, "y"), (x.grad, xr.grad, "dx"), (conv.weight.grad, wr.grad, "dw")):
        err = float((got.detach().cpu().double() - ref.detach()).abs().max() / ref.detach().abs().max())
        assert err < 2e-5, (d, what, err)


@pytest.mark.gpu
@pytest.mark.parametrize("bf16", [False, True])
@pytest.mark.parametrize("name", ["classes", "wide", "nyu_p5_s2", "nyu_p5_s2x", "nyu_20_5", "nyu_head", "k3_s2_p0"])
def test_forward_and_data_gradient_write_their_own_zero_pads_gpu(name, bf16, hip_lib):
    """The padded output rows of `_Conv3dFn.forward` and `conv3d_dgrad` carry ZERO pad lanes whatever the allocator handed back
    (buffers full of NaN here): consumers take such rows in place (`_padded_rows`), and NaN * 0 in a pad lane would poison the
    step.  (The kernels write zero pads themselves in all of these cases; the Functions memset as well -- round 6 dropped the
    memsets once and one run in six of the reduced model's five-step test produced a NaN parameter.)"""
    from occdepth_amd import hip
    cin, cout, k, s, p, d, dims, bias = CONV_CASES[name]
    x, w, b = conv_tensors(name)
    old = ag.set_bf16_mfma(bf16)
    try:
        for _ in range(2):                                         # (the caching allocator recycles these blocks for the outputs)
            junk = [torch.full((1 << 22,), float("nan"), device="cuda") for _ in range(4)]
            del junk
        y = ag._Conv3dFn.apply(x.cuda(), w.cuda(), b.cuda() if b is not None else None, s, p, d)
        rows = y._base if y._base is not None else y
        assert rows.shape[-1] == hip.round_up(cout, 8) and bool(torch.isfinite(rows).all())
        if rows.shape[-1] > cout:
            assert float(rows[..., cout:].abs().max()) == 0.0
        gy = torch.randn(y.shape, generator=torch.Generator().manual_seed(3)).cuda()
        junk = [torch.full((1 << 22,), float("nan"), device="cuda") for _ in range(4)]
        del junk
        dx = ag.conv3d_dgrad(ag._to_vox(gy), w.cuda(), tuple(dims), s, p, d)
        assert bool(torch.isfinite(dx.buf).all())
        if dx.cs > cin:
            assert float(dx.buf[..., cin:].abs().max()) == 0.0
    finally:
        ag.set_bf16_mfma(old)

