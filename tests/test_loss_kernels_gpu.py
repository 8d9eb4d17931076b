"""-m gpu: the round-5 training-step kernels of SURVEY 8(f) row N1 through the C ABI --
  * relation (context prior) BCE: occd_relation_bce_stats / _grad      (ref occdepth/loss/CRP_loss.py:4-24)
  * depth-distribution BCE:       occd_depth_bce_stats / _grad         (ref occdepth/loss/depth_loss.py:14-87)
  * frustum-sample transpose:     occd_flosp_sample_bwd                (ref occdepth/models/f2v/sampler.py:59-64 under autograd)
  * the scene-completion statistics / gradient / confusion passes on CHANNELS-LAST logits (no layout copies)
against the float64 emulation of the published semantics (tests/emu.py), against ATen's autograd where ATen has the op, and
through size-independent properties (determinism, layout independence) at BASELINE config-2 size.  The real-reference
goldens of the two losses and their gradients (tests/golden/losses.npz) are checked by tests/test_losses_gpu.py through the
very same kernels."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import emu

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(scope="module")
def hip():
    from occdepth_amd import hip as h
    h.load()
    return h


# ------------------------------------------------------------------------------------------------ relation BCE
def _relation_case(B, R, M, N, seed, label_dtype):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, R, M, N, generator=g) * 3.0
    y = torch.rand(B, R, N, M, generator=g) < torch.tensor([0.5, 0.1, 0.02, 0.3][:R]).view(1, R, 1, 1)
    return x, y.to(label_dtype)


def _reference_relation(x, labels):
    """CRP_loss.py:4-24 verbatim, float64."""
    bs, R = x.shape[:2]
    lg = torch.cat([x[i].permute(0, 2, 1).reshape(R, -1) for i in range(bs)], 1).T.double()
    lb = torch.cat([labels[i].reshape(R, -1) for i in range(bs)], 1).T
    pw = (lb == 0).sum(0) / lb.sum(0)
    return torch.nn.BCEWithLogitsLoss(pos_weight=pw.double())(lg, lb.double())


@pytest.mark.parametrize("layout", ["n_contiguous", "m_contiguous"])
@pytest.mark.parametrize("label_dtype", [torch.uint8, torch.float32, torch.bool])
@pytest.mark.parametrize("shape", [(2, 4, 24, 40), (1, 4, 70, 517), (1, 2, 513, 33)])
def test_relation_bce_kernels(hip, shape, label_dtype, layout):
    """Both logit layouts the model produces -- (B, R, M, N) contiguous from the autograd graph and the permuted view of
    (R, B, N, M) rows from the HIP forward -- give the reference's loss (float64, verbatim formula) and its gradient."""
    from occdepth_amd.loss.CRP_loss import compute_super_CP_multilabel_loss
    B, R, M, N = shape
    x, y = _relation_case(B, R, M, N, sum(shape), label_dtype)
    xd = x.double().requires_grad_(True)
    ref = _reference_relation(xd, y.to(torch.uint8))
    ref.backward()
    if layout == "n_contiguous":
        leaf = x.to(DEV).requires_grad_(True)
        lg = leaf
    else:                                      # the HIP forward's layout: rows (R, B, N, ceil8(M)) viewed as (B, R, M, N)
        m_cs = (M + 7) // 8 * 8
        leaf = torch.zeros(R, B, N, m_cs)
        leaf[..., :M] = x.permute(1, 0, 3, 2)
        leaf = leaf.to(DEV).requires_grad_(True)
        lg = leaf.view(R, B, N, m_cs)[..., :M].permute(1, 0, 3, 2)
        if m_cs != M:
            assert not hip.relation_bce_usable(lg, y.to(DEV))          # padded rows are not a dense permutation: ATen path
            lg = lg.contiguous()
    labels = [y[i].to(DEV) for i in range(B)]
    with hip.profile() as prof:
        loss = compute_super_CP_multilabel_loss(lg, labels)
        loss.backward()
    assert {k.split(":")[0] for k in prof.rows} >= {"relation_bce_stats", "relation_bce_grad"}, prof.rows.keys()
    assert float(loss.detach()) == pytest.approx(float(ref.detach()), rel=2e-6)
    got = leaf.grad.cpu().double()
    if layout == "m_contiguous":
        assert float(got[..., M:].abs().max() if got.shape[-1] > M else 0.0) == 0.0
        got = got[..., :M].permute(1, 0, 3, 2)
    err = float((got - xd.grad).abs().max() / xd.grad.abs().max())
    assert err < 2e-6, err
    # the statistics are integer sums: bit-identical from run to run, whatever the layout
    s1 = hip.relation_bce_stats(lg.detach(), torch.stack(labels))
    s2 = hip.relation_bce_stats(lg.detach().contiguous(), torch.stack(labels))
    assert torch.equal(s1, s2) and torch.equal(s1, hip.relation_bce_stats(lg.detach(), torch.stack(labels)))
    want = emu.relation_bce_stats(x, y)
    assert torch.equal(s1.cpu()[:, 0], want[:, 0])                                        # counts: exact
    assert torch.allclose(s1.cpu()[:, 1:].double(), want[:, 1:].double(), rtol=2e-6, atol=B * M * N * 0.5)


def test_relation_bce_config2_size_properties(hip):
    """(1, 4, 512, 4096) as in the config-2 step: loss = the ATen formulation on the GPU (float32) to 1e-5; gradient sums to
    what the closed form says (sum of d loss / d x over y = 0 elements of relation r = coef_neg * sum sigmoid(x))."""
    from occdepth_amd.loss.CRP_loss import compute_super_CP_multilabel_loss
    x, y = _relation_case(1, 4, 512, 4096, 7, torch.uint8)
    xg, yg = x.to(DEV).requires_grad_(True), y.to(DEV)
    loss = compute_super_CP_multilabel_loss(xg, [yg[0]])
    loss.backward()
    saved = hip.LOSS_KERNELS
    hip.LOSS_KERNELS = False
    try:
        xa = x.to(DEV).requires_grad_(True)
        la = compute_super_CP_multilabel_loss(xa, [yg[0]])
        la.backward()
    finally:
        hip.LOSS_KERNELS = saved
    assert float(loss) == pytest.approx(float(la), rel=1e-5)
    assert float((xg.grad - xa.grad).abs().max() / xa.grad.abs().max()) < 1e-5


# ------------------------------------------------------------------------------------------------ depth BCE
@pytest.mark.parametrize("geo", [(2, 24, 6, 10, 48, 80, 8), (2, 104, 47, 153, 370, 1220, 8), (3, 16, 5, 7, 23, 31, 4)])
def test_depth_bce_kernels(hip, geo):
    """Loss and gradient (a) against the float64 emulation of the published semantics (tests/emu.py: F.interpolate nearest,
    nearest depth per cell, LID bin, one-hot, clamped-log BCE, ATen's BCE backward) at 3e-6 and (b) against the product's own
    ATen formulation of depth_loss.py:54-87 run in float32 on the CPU (kernels switched off) at float32 round-off -- incl. a
    label map that is NOT cell x the prediction grid (370 x 1220 -> 376 x 1224: the nearest resample of :72-76), p = 0 / 1
    corners of the clamped logs, and a camera slice of a two-camera prediction tensor."""
    from occdepth_amd.loss.depth_loss import DepthClsLoss
    Bn, D, h, w, sh, sw, cell = geo
    g = torch.Generator().manual_seed(D * 7 + h)
    d_bound = [2.0, 2.0 + 0.5 * D, 0.5]
    d_off, d_step = d_bound[0] - d_bound[2], d_bound[2]
    gt = torch.rand(Bn, 1, sh, sw, generator=g) * (d_bound[1] + 4.0)
    gt = gt * (torch.rand(Bn, 1, sh, sw, generator=g) < 0.08)                       # sparse returns
    both = torch.softmax(torch.randn(Bn, 2, D, h, w, generator=g) * 2.0, 2)
    both[0, 0, 3, 1, 2] = 0.0                                                        # the clamped-log corners
    both[0, 0, 4, 1, 2] = 1.0
    fn = DepthClsLoss(cell, d_bound)
    saved = hip.LOSS_KERNELS
    hip.LOSS_KERNELS = False
    try:
        leaf32 = both.clone().requires_grad_(True)
        ref32 = fn.get_depth_loss(gt, leaf32[:, 0].unsqueeze(1))
        ref32.backward()
    finally:
        hip.LOSS_KERNELS = saved
    leaf = both.to(DEV).requires_grad_(True)
    with hip.profile() as prof:
        loss = fn.get_depth_loss(gt.to(DEV), leaf[:, 0].unsqueeze(1))
        loss.backward()
    assert {k.split(":")[0] for k in prof.rows} >= {"depth_bce_stats", "depth_bce_grad"}, prof.rows.keys()
    # (a) float64 emulation
    st64 = emu.depth_bce_stats(both[:, 0], gt[:, 0], cell, d_off, d_step).double()
    n_meas = max(1.0, float(st64[1]))
    assert float(st64[1]) > 0
    assert float(loss) == pytest.approx(float(st64[0]) / 16777216.0 / n_meas, rel=3e-6)
    g64 = emu.depth_bce_grad(both[:, 0].double(), gt[:, 0], cell, d_off, d_step, torch.tensor([1.0 / n_meas], dtype=torch.float64))
    got = leaf.grad[:, 0].cpu().double()
    small = g64.abs() < 1e6                                     # (p = 0 / 1 cells: 1 / max(p (1 - p), 1e-12) = 1e12 in both)
    assert float((got - g64.double())[small].abs().max() / g64[small].abs().max()) < 3e-6
    if (~small).any():
        assert float(((got - g64.double())[~small].abs() / g64.double()[~small].abs()).max()) < 1e-5
    assert torch.equal(leaf.grad[:, 1], torch.zeros_like(leaf.grad[:, 1]))          # the camera without ground truth
    # (b) the ATen formulation in float32
    assert float(loss) == pytest.approx(float(ref32), rel=2e-5)
    r32 = leaf32.grad[:, 0].double()
    assert float((got - r32)[small].abs().max() / r32[small].abs().max()) < 2e-4
    # statistics: integer sums (bit-identical from run to run), the measured-cell count exact
    prob = leaf.detach()[:, 0].contiguous()
    st = hip.depth_bce_stats(prob, gt[:, 0].to(DEV).contiguous(), cell, d_off, d_step)
    assert int(st[1]) == int(st64[1])
    assert torch.equal(st, hip.depth_bce_stats(prob, gt[:, 0].to(DEV).contiguous(), cell, d_off, d_step))


# ------------------------------------------------------------------------------------------------ frustum-sample transpose
def _frustum_case(hip, B, V, D, h, w, dims, seed, infer=False):
    """A FlospDepth-like geometry: cameras in front of the grid, LID bins; returns the hip.Frustum on the GPU + CPU copies."""
    g = torch.Generator().manual_seed(seed)
    depth = torch.softmax(torch.randn(B, V, D, h, w, generator=g), 2)
    A, Bd, C = dims
    final_dim = (h * 8, w * 8)
    # grid index -> "lidar" metres -> camera (x right, y down, z forward): a gentle perspective view of the whole grid
    trans = torch.zeros(B, V, 4, 4)
    for b in range(B):
        for v in range(V):
            sx = 40.0 / A
            trans[b, v] = torch.tensor([[0.0, -sx, 0.0, 20.0 + 0.6 * v], [0.0, 0.0, -sx * 0.5, 3.0],
                                        [sx, 0.0, 0.0, 1.0 + 0.2 * b], [0.0, 0.0, 0.0, 1.0]])
    f = 0.9 * final_dim[1]
    proj = torch.tensor([[f, 0.0, final_dim[1] / 2, 0.0], [0.0, f, final_dim[0] / 2, 0.0], [0.0, 0.0, 1.0, 0.0]]).expand(B, V, 3, 4).contiguous()
    ida = torch.eye(4).expand(B, V, 4, 4).contiguous()
    fr = hip.Frustum(depth.to(DEV), trans.to(DEV), proj.to(DEV), ida.to(DEV), dims, final_dim, 2.0, 58.0, True)
    return fr, depth, trans, proj, ida, final_dim


@pytest.mark.parametrize("geo", [(1, 2, 24, 6, 10, (16, 16, 4)), (2, 1, 16, 5, 9, (8, 12, 5)), (1, 2, 104, 47, 153, (32, 32, 8))])
def test_frustum_sample_backward_is_the_transpose_and_deterministic(hip, geo):
    """d depth from occd_flosp_sample_bwd == autograd through the float64 emulation of the forward (F.grid_sample x 2 per
    camera + the mean over cameras), i.e. the transpose of the very weights K1a applies; two launches agree bit for bit
    (fixed-point accumulation), and <out, gout> == <depth, gdepth> (adjoint identity) with the HIP forward."""
    B, V, D, h, w, dims = geo
    fr, depth, trans, proj, ida, final_dim = _frustum_case(hip, B, V, D, h, w, dims, sum(dims) + D)
    nvox = dims[0] * dims[1] * dims[2]
    g = torch.Generator().manual_seed(99)
    gout = torch.randn(B, nvox, generator=g) * torch.rand(B, nvox, generator=g)
    gout[0, :7] = 0.0
    got = hip.flosp_sample_bwd(fr, gout.to(DEV))
    again = hip.flosp_sample_bwd(fr, gout.to(DEV))
    assert torch.equal(got, again)
    cpu_fr = hip.Frustum(depth, trans, proj, ida, dims, final_dim, 2.0, 58.0, True)
    want = emu.flosp_sample_bwd(cpu_fr, gout)
    scale = want.abs().max()
    assert float(scale) > 0
    err = float((got.cpu() - want).abs().max() / scale)
    # the trilinear weights come from float32 sample coordinates (ulp ~1e-5 at ix ~ 150) evaluated by two different
    # instruction sequences (the kernel's fma contraction vs ATen's): bound = a few coordinate ulps, not the sum's round-off
    assert err < 3e-5, err
    out = fr.sample()
    lhs = float((out.double().cpu() * gout.double()).sum())
    rhs = float((depth.double() * got.double().cpu()).sum())
    assert lhs == pytest.approx(rhs, rel=1e-5, abs=1e-6 * float(gout.abs().sum()))
    # zero gradient in -> zero out (and no NaN from the scale of an all-zero maximum)
    z = hip.flosp_sample_bwd(fr, torch.zeros(B, nvox, device=DEV))
    assert torch.equal(z, torch.zeros_like(z))


def test_flosp_depth_training_path_uses_the_kernels(hip):
    """FlospDepth in training mode on the GPU: the sampled volume and d loss / d (DepthNet input) equal the ATen graph's
    (OCCDEPTH_LOSS_KERNELS=0) to float32 round-off, and the launch profile holds flosp_sample + flosp_sample_bwd."""
    from test_oracle_vs_golden import build_product
    import golden_cases as gc
    m, cfg, sd = build_product("kitti_small")
    fd = m.flosp_depth if hasattr(m, "flosp_depth") else None
    if fd is None:
        pytest.skip("config without flosp_depth")
    fd = fd.to(DEV).train()
    batch = gc.occdepth_batch("kitti_small")
    g = torch.Generator().manual_seed(1)
    c_in = fd.depth_net_conf["in_channels"]
    h, w = 6, 20
    feat = torch.randn(1, 2, c_in, h, w, generator=g).to(DEV)
    kw = dict(cam_k=[c.to(DEV) for c in batch["cam_k"]], T_velo_2_cam=[t.to(DEV) for t in batch["T_velo_2_cam"]],
              ida_mats=[t.to(DEV) for t in batch["ida_mats"]])
    res = {}
    for on in (True, False):
        saved = hip.LOSS_KERNELS
        hip.LOSS_KERNELS = on
        try:
            x = feat.clone().requires_grad_(True)
            torch.manual_seed(0)
            with hip.profile() as prof:
                out = fd(x, **kw)
                vox = out[0] if isinstance(out, tuple) else out
                (vox * torch.linspace(0.5, 1.5, vox.numel(), device=DEV).view_as(vox)).sum().backward()
            res[on] = (vox.detach(), x.grad.clone(), {k.split(":")[0] for k in prof.rows})
        finally:
            hip.LOSS_KERNELS = saved
    assert {"flosp_sample", "flosp_sample_bwd"} <= res[True][2] and "flosp_sample_bwd" not in res[False][2]
    assert float((res[True][0] - res[False][0]).abs().max()) <= 2e-6 * float(res[False][0].abs().max() + 1e-12)
    assert float((res[True][1] - res[False][1]).abs().max()) <= 1e-4 * float(res[False][1].abs().max() + 1e-12)


# ------------------------------------------------------------------------------------------------ channels-last ssc passes
@pytest.mark.parametrize("C,dims,B", [(20, (16, 16, 8), 2), (2, (16, 16, 8), 2), (12, (10, 6, 10), 1), (20, (64, 64, 32), 1)])
def test_ssc_passes_on_channels_last_logits(hip, C, dims, B):
    """The statistics (integer sums: bit-exact), the gradient and the confusion counts do not depend on whether the logits
    are (B, C, S) planes or the 3-D stack's channels-last rows; the channels-last gradient comes back as zero-padded rows
    that autograd3d._to_vox consumes without a copy."""
    from occdepth_amd import autograd3d
    g = torch.Generator().manual_seed(C + dims[0])
    S = dims[0] * dims[1] * dims[2]
    cs = (C + 7) // 8 * 8
    rows = torch.zeros(B, *dims, cs)
    rows[..., :C] = torch.randn(B, *dims, C, generator=g) * 2.0
    rows = rows.to(DEV)
    cl = rows[..., :C].permute(0, 4, 1, 2, 3)                       # (B, C, X, Y, Z) view, channels-last strides
    planes = cl.contiguous()
    assert hip._logit_layout(cl) == (S * cs, 1, cs) and hip._logit_layout(planes) == (C * S, S, 1)
    target = torch.randint(0, C + 1, (B, *dims), generator=g).to(torch.uint8)
    target[target == C] = 255
    target = target.to(DEV)
    Fm = 5
    masks = (torch.rand(B, Fm, *dims, generator=g) < 0.2).view(torch.uint8).to(DEV)
    w = (torch.rand(C, generator=g) + 0.5).to(DEV)
    for map_occ in ((False, True) if C == 2 else (False,)):
        s_cl = hip.ssc_loss_stats(cl, target, masks, w, map_occ)
        s_pl = hip.ssc_loss_stats(planes, target, masks, w, map_occ)
        assert torch.equal(s_cl, s_pl)
        gs = torch.randn(3 * C + 3 + Fm * C, generator=g).to(DEV)
        g_cl = hip.ssc_loss_grad(cl, target, masks, w, gs, map_occ)
        g_pl = hip.ssc_loss_grad(planes, target, masks, w, gs, map_occ)
        assert g_cl.shape == g_pl.shape and g_cl.stride(1) == 1 and g_pl.is_contiguous()
        assert float((g_cl - g_pl).abs().max()) <= 1e-6 * float(g_pl.abs().max())
        v = autograd3d._to_vox(g_cl)
        assert v.buf.data_ptr() == g_cl.data_ptr() and v.C == C and v.cs == cs                       # consumed in place
        if cs > C:
            assert float(v.buf[..., C:].abs().max()) == 0.0                                             # pads written as zeros
    h_cl = hip.ssc_confusion(torch.zeros(C, C, dtype=torch.int64, device=DEV), target, logits=cl)
    h_pl = hip.ssc_confusion(torch.zeros(C, C, dtype=torch.int64, device=DEV), target, logits=planes)
    assert torch.equal(h_cl, h_pl) and int(h_cl.sum()) == int((target != 255).sum())
    # a channel slice of a WIDER tensor is not mistaken for padded rows
    wide = torch.randn(B, *dims, cs + 8, device=DEV)
    sl = wide[..., :C].permute(0, 4, 1, 2, 3)
    assert autograd3d._padded_rows(sl) is None


def test_ssc_losses_autograd_on_channels_last_logits_matches_planes(hip):
    """ssc_loss.ssc_losses end to end (statistics -> float64 formulas -> gradient pass) on a channels-last leaf: same loss
    values and gradient as on the plane layout, with no layout copy of the logits in the launch profile."""
    from occdepth_amd.loss import ssc_loss
    import golden_cases as gc
    d = gc.loss_case("kitti_like")
    x = d["ssc_logit"]
    B, C = x.shape[:2]
    cs = (C + 7) // 8 * 8
    rows = torch.zeros(B, *x.shape[2:], cs)
    rows[..., :C] = x.permute(0, 2, 3, 4, 1)
    leaf_rows = rows.to(DEV).requires_grad_(True)
    cl = leaf_rows[..., :C].permute(0, 4, 1, 2, 3)
    leaf_pl = x.to(DEV).requires_grad_(True)
    args = (d["target"].to(DEV), d["class_weights"].to(DEV), [m.to(DEV) for m in d["frustums_masks"]],
            [f.to(DEV) for f in d["frustums_class_dists"]])
    tot = {}
    for name, t in (("cl", cl), ("planes", leaf_pl)):
        out = ssc_loss.ssc_losses(t, *args)
        tot[name] = sum(out.values())
        tot[name].backward()
    assert float(tot["cl"]) == pytest.approx(float(tot["planes"]), rel=1e-9)
    g_cl = leaf_rows.grad[..., :C].permute(0, 4, 1, 2, 3)
    assert float((g_cl - leaf_pl.grad).abs().max()) <= 1e-6 * float(leaf_pl.grad.abs().max())

