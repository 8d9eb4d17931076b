"""-m gpu: the differentiable HIP lift (occdepth_amd/lift_autograd.py: K1b forward + the one-launch backward
`occd_lift_bwd`) against the reference-structure ATen graph (per-sample, per-scale `SFA._forward_autograd`, then
`* depth * 100`, occdepth/models/OccDepth.py:266-298,339) evaluated in float64 on the CPU."""
import pytest
import torch

pytestmark = pytest.mark.gpu

# (B, V, C, scene (X, Y, Z) at project_scale 1, image (H, W), with depth volume)
CASES = [(1, 2, 16, (16, 12, 8), (23, 41), True), (2, 2, 64, (8, 8, 8), (30, 50), True),
         (1, 1, 32, (8, 8, 4), (17, 29), False), (1, 3, 8, (8, 8, 4), (20, 24), True)]
SCALES = [1, 2, 4, 8]


def make(case):
    B, V, C, scene, (H, W), with_depth = case
    g = torch.Generator().manual_seed(B * 7 + V * 3 + C)
    N = scene[0] * scene[1] * scene[2]
    feats = [[torch.randn(B, C, -(-H // s), -(-W // s), generator=g) for _ in range(V)] for s in SCALES]
    px = torch.randint(0, W, (B, V, N, 1, 1), generator=g)
    py = torch.randint(0, H, (B, V, N, 1, 1), generator=g)
    pix = torch.cat([px, py], -1)
    fov = torch.rand(B, V, N, 1, generator=g) < 0.7
    depth = torch.rand(B, 1, *scene, generator=g) if with_depth else None
    R = torch.randn(B, C, *scene, generator=g)
    return feats, pix, fov, depth, R


def reference(case, feats, pix, fov, depth, R):
    from occdepth_amd.models.SFA import SFA
    B, V, C, scene, _, with_depth = case
    feats = [[f.double().requires_grad_(True) for f in per] for per in feats]
    d = depth.double().requires_grad_(True) if with_depth else None
    sfa = SFA(scene, "kitti", 1)
    outs = []
    for b in range(B):
        x3d = 0
        for si, s in enumerate(SCALES):
            x2d = torch.stack([feats[si][v][b] for v in range(V)])
            x3d = x3d + sfa._forward_autograd(x2d, torch.div(pix[b], s, rounding_mode="floor"), fov[b])
        outs.append(x3d)
    out = torch.stack(outs)
    if with_depth:
        out = out * d * 100
    (out * R.double()).sum().backward()
    return out.detach(), [[f.grad for f in per] for per in feats], (d.grad if with_depth else None)


@pytest.mark.parametrize("case", CASES)
def test_fused_lift_forward_and_backward_match_the_aten_graph(case, hip_lib):
    from occdepth_amd import lift_autograd
    B, V, C, scene, _, with_depth = case
    feats, pix, fov, depth, R = make(case)
    ref_out, ref_gf, ref_gd = reference(case, feats, pix, fov, depth, R)
    dev_feats = [[f.cuda().requires_grad_(True) for f in per] for per in feats]
    dev_depth = depth.cuda().requires_grad_(True) if with_depth else None
    assert lift_autograd.usable(dev_feats, pix.cuda())
    out = lift_autograd.lift_scales_autograd(dev_feats, SCALES, pix.cuda(), fov.cuda(), scene, 1, "kitti",
                                             depth_scale=dev_depth, scale_const=100.0)
    assert out.shape == ref_out.shape
    assert float((out.double().cpu() - ref_out).abs().max() / ref_out.abs().max()) < 1e-5
    (out * R.cuda()).sum().backward()
    for si in range(len(SCALES)):
        for v in range(V):
            g, r = dev_feats[si][v].grad.double().cpu(), ref_gf[si][v]
            assert g.shape == r.shape
            assert float((g - r).abs().max() / r.abs().max()) < 1e-4, (si, v)
    if with_depth:
        g, r = dev_depth.grad.double().cpu(), ref_gd
        assert g.shape == r.shape and float((g - r).abs().max() / r.abs().max()) < 1e-4
