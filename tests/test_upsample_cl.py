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
    # round 6: channels-last pixel rows on a pitch of ceil8(C + Cs) floats with ZERO pads (what the level's first convolution takes
    # in place: autograd3d._padded_rows); a multiple-of-8 channel count is plain channels_last memory as before
    cs = -(-(C + Cs) // 8) * 8
    assert out.shape == ref.shape and out.stride() == (H * W * cs, 1, W * cs, cs) and out.storage_offset() == 0
    assert out.is_contiguous(memory_format=torch.channels_last) == (cs == C + Cs)
    rows = torch.as_strided(out.detach(), (B, H, W, cs), (H * W * cs, W * cs, cs, 1))
    assert cs == C + Cs or float(rows[..., C + Cs:].abs().max()) == 0.0
    from occdepth_amd import autograd3d
    # consumed in place as an X = 1 volume, the way the decoder hands it to the convolution Function (view, then detach)
    assert autograd3d._padded_rows(out.unsqueeze(2).detach()) is not None or (C + Cs) % 8 == 0
    # ... while a channel slice of a WIDER tensor in the same layout -- its "pads" are the neighbour's data -- is not
    # (cs - 4 channels out of cs: the strides, the offset and the storage size are exactly those of a padded row buffer)
    wide = torch.randn(B, H, W, cs, device="cuda").permute(0, 3, 1, 2)
    assert autograd3d._padded_rows(wide[:, :cs - 4].unsqueeze(2)) is None
    out.backward(go.cuda().contiguous(memory_format=torch.channels_last))
    assert float((out.detach().double().cpu() - ref.detach()).abs().max()) < 2e-6 * float(ref.abs().max())
    assert float((xg.grad.double().cpu() - xr.grad).abs().max()) < 2e-6 * float(xr.grad.abs().max())
    assert torch.equal(sg.grad.cpu(), go[:, C:])
    # the gradient as the convolution's data gradient delivers it: rows on the same pitch (pads hold anything) -- no copy, same result
    gpad = torch.full((B, H, W, cs), float("nan"), device="cuda")
    gpad[..., :C + Cs] = go.cuda().permute(0, 2, 3, 1)
    xg2 = x.cuda().contiguous(memory_format=torch.channels_last).requires_grad_(True)
    sg2 = s.cuda().contiguous(memory_format=torch.channels_last).requires_grad_(True)
    hip.upsample_bilinear_cat_cl_autograd(xg2, sg2).backward(gpad[..., :C + Cs].permute(0, 3, 1, 2))
    assert torch.equal(xg2.grad, xg.grad) and torch.equal(sg2.grad.cpu(), go[:, C:])
