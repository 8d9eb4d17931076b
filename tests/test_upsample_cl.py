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
