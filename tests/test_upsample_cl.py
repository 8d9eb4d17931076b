"""Channels-last upsample + concat of the bf16-mode decoder levels (hip._UpCatClFn: occd_upsample_bilinear_cat_nhwc forward,
occd_upsample_bilinear_nhwc_bwd gather backward) against ATen's F.interpolate(bilinear, align_corners=True) + torch.cat in
float64 (reference: occdepth/models/unet2d.py:38-46)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

# (B, C, Cskip, h, w, H, W)
CASES = [(1, 160, 3, 19, 31, 37, 61), (2, 24, 8, 6, 10, 12, 20), (1, 7, 5, 5, 4, 9, 8), (1, 16, 4, 1, 3, 4, 7),
         (1, 12, 4, 8, 9, 8, 9), (1, 320, 32, 24, 39, 47, 77)]


@pytest.mark.parametrize("case", CASES)
def test_upsample_cat_channels_last_forward_and_backward(case, hip_lib):
    from occdepth_amd import hip
    B, C, Cs, h, w, H, W = case
    g = torch.Generator().manual_seed(sum(case))
    x = torch.randn(B, C, h, w, generator=g)
    s = torch.randn(B, Cs, H, W, generator=g)
    go = torch.randn(B, C + Cs, H, W, generator=g)
    xr, sr = x.double().requires_grad_(True), s.double().requires_grad_(True)
    ref = torch.cat([F.interpolate(xr, size=(H, W), mode="bilinear", align_corners=True), sr], 1)
    ref.backward(go.double())
    xg = x.cuda().contiguous(memory_format=torch.channels_last).requires_grad_(True)
    sg = s.cuda().contiguous(memory_format=torch.channels_last).requires_grad_(True)
    out = hip.upsample_bilinear_cat_cl_autograd(xg, sg)
    assert out.shape == ref.shape and out.is_contiguous(memory_format=torch.channels_last)
    out.backward(go.cuda().contiguous(memory_format=torch.channels_last))
    assert float((out.detach().double().cpu() - ref.detach()).abs().max()) < 2e-6 * float(ref.abs().max())
    assert float((xg.grad.double().cpu() - xr.grad).abs().max()) < 2e-6 * float(xr.grad.abs().max())
    assert torch.equal(sg.grad.cpu(), go[:, C:])


@pytest.mark.parametrize("case", [(1, 160, 3, 19, 31, 37, 61), (2, 7, 5, 5, 4, 9, 8)])
def test_upsample_cat_rows_on_a_pitch(case, hip_lib):
    """occd_upsample_bilinear_cat_nhwc_rows (round 6): the same launch with output rows of ceil8(C + Cs) floats and ZERO pad lanes
    -- the layout a convolution takes in place.  Equal to the dense entry point on the real lanes, zero on the pads."""
    from occdepth_amd import hip
    B, C, Cs, h, w, H, W = case
    g = torch.Generator().manual_seed(sum(case) + 1)
    xr = torch.randn(B, h, w, C, generator=g).cuda()
    sr = torch.randn(B, H, W, Cs, generator=g).cuda()
    dense = torch.empty(B, H, W, C + Cs, device="cuda")
    lib = hip.load()
    hip._check(lib.occd_upsample_bilinear_cat_nhwc(xr.data_ptr(), sr.data_ptr(), dense.data_ptr(), B, C, Cs, h, w, H, W, None), "dense")
    cs = -(-(C + Cs) // 8) * 8
    rows = torch.full((B, H, W, cs), float("nan"), device="cuda")
    hip._check(lib.occd_upsample_bilinear_cat_nhwc_rows(xr.data_ptr(), sr.data_ptr(), rows.data_ptr(), B, C, Cs, h, w, H, W, cs, None),
               "rows")
    torch.cuda.synchronize()
    assert torch.equal(rows[..., :C + Cs], dense) and float(rows[..., C + Cs:].abs().max()) == 0.0
    assert lib.occd_upsample_bilinear_cat_nhwc_rows(xr.data_ptr(), sr.data_ptr(), rows.data_ptr(), B, C, Cs, h, w, H, W, C + Cs - 1,
                                                    None) == -1
