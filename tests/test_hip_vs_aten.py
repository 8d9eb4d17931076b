"""GPU numerics: every HIP kernel / fused eval module against plain PyTorch fp32 (ATen on the same
GPU) of the same op.  Tolerances: fp32 MFMA is an exact-fp32 fma chain, BatchNorm folding and a
different accumulation order move results by a few ulp -> 2e-4 relative to the tensor's scale.
"""
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DEV = "cuda"


def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")


def rel_err(a, b):
    a, b = a.float(), b.float()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-20)).item()


def randomize_bn(mod, seed=0):
    g = torch.Generator().manual_seed(seed)
    for m in mod.modules():
        if isinstance(m, (nn.BatchNorm2d, nn.BatchNorm3d)):
            n = m.num_features
            m.weight.data = torch.rand(n, generator=g) + 0.5
            m.bias.data = torch.randn(n, generator=g) * 0.2
            m.running_mean.data = torch.randn(n, generator=g) * 0.2
            m.running_var.data = torch.rand(n, generator=g) + 0.5
    return mod


@pytest.fixture(scope="module")
def hip(hip_lib):
    _need_gpu()
    from occdepth_amd import hip as h
    torch.backends.cudnn.allow_tf32 = False
    torch.backends.cuda.matmul.allow_tf32 = False
    return h


CONV_CASES = [
    # (B, Cin, Cout, dims, kernel, stride, dilation, padding)
    (1, 32, 32, (6, 20, 32), (3, 3, 3), (1, 1, 1), (1, 1, 1), (1, 1, 1)),
    (1, 32, 32, (9, 17, 32), (3, 3, 3), (1, 1, 1), (3, 3, 3), (3, 3, 3)),
    (2, 16, 16, (5, 12, 16), (1, 1, 3), (1, 1, 1), (1, 1, 2), (0, 0, 2)),
    (1, 16, 16, (6, 12, 16), (1, 3, 1), (1, 2, 1), (1, 1, 1), (0, 1, 0)),
    (1, 16, 16, (6, 12, 8), (3, 1, 1), (2, 1, 1), (1, 1, 1), (1, 0, 0)),
    (1, 64, 16, (4, 16, 16), (1, 1, 1), (1, 1, 1), (1, 1, 1), (0, 0, 0)),
    (1, 16, 128, (4, 8, 8), (1, 1, 1), (1, 1, 1), (1, 1, 1), (0, 0, 0)),
    (1, 64, 128, (8, 8, 4), (3, 3, 3), (2, 2, 2), (1, 1, 1), (1, 1, 1)),
    (1, 34, 20, (5, 9, 32), (3, 3, 3), (1, 1, 1), (1, 1, 1), (1, 1, 1)),
    (1, 32, 2, (5, 9, 32), (3, 3, 3), (1, 1, 1), (1, 1, 1), (1, 1, 1)),
    (1, 100, 25, (7, 9, 15), (1, 1, 1), (1, 1, 1), (1, 1, 1), (0, 0, 0)),
    (1, 25, 25, (7, 9, 15), (1, 3, 1), (1, 1, 1), (1, 2, 1), (0, 2, 0)),
    (1, 256, 256, (8, 8, 4), (3, 3, 3), (1, 1, 1), (2, 2, 2), (2, 2, 2)),
    (1, 96, 72, (3, 5, 60), (3, 3, 3), (1, 1, 1), (1, 1, 1), (1, 1, 1)),
    (1, 24, 40, (3, 3, 70), (3, 3, 3), (1, 1, 1), (1, 1, 1), (1, 1, 1)),
    (1, 64, 32, (4, 6, 8), (2, 2, 2), (2, 2, 2), (1, 1, 1), (0, 0, 0)),
]


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv3d_vs_aten(hip, case):
    B, cin, cout, dims, k, s, d, p = case
    torch.manual_seed(hash(case) % 1000)
    x = torch.randn(B, cin, *dims, device=DEV)
    w = torch.randn(cout, cin, *k, device=DEV) / (cin * k[0] * k[1] * k[2]) ** 0.5
    bias = torch.randn(cout, device=DEV)
    ref = F.conv3d(x, w, bias, stride=s, padding=p, dilation=d)
    r1 = torch.randn_like(ref)
    r2 = torch.randn_like(ref)
    from occdepth_amd.fused import _pad_bias
    vx = hip.Vox.from_ncdhw(x)
    wpk = hip.pack_weights(w)
    out = hip.Vox.empty(B, tuple(ref.shape[2:]), cout, DEV)
    # plain
    hip.conv3d(vx, wpk, _pad_bias(bias, cout), cout, k, out, stride=s, dilation=d, padding=p)
    torch.cuda.synchronize()
    assert rel_err(out.ncdhw(), ref) < 2e-5
    # relu in, two residuals, relu out
    out2 = hip.Vox.empty(B, tuple(ref.shape[2:]), cout, DEV)
    hip.conv3d(vx, wpk, _pad_bias(bias, cout), cout, k, out2, stride=s, dilation=d, padding=p,
               res1=hip.Vox.from_ncdhw(r1), res2=hip.Vox.from_ncdhw(r2), act_in=hip.ACT_RELU,
               act_out=hip.ACT_RELU)
    ref2 = F.relu(F.conv3d(F.relu(x), w, bias, stride=s, padding=p, dilation=d) + r1 + r2)
    assert rel_err(out2.ncdhw(), ref2) < 2e-5
    # channel pad of the output rows must be exactly zero
    if out.cs > cout:
        assert out.buf[..., cout:].abs().max().item() == 0.0
    # nhwc_to_nchw copy == view
    assert torch.equal(hip.nhwc_to_nchw(out), out.ncdhw().contiguous())


PERSIST_CASES = [
    # shapes that qualify for the persistent weights-stationary head kernel (Z == 32 k, 25..32 -> <=32 channels)
    (1, 32, 32, (16, 256, 32), 1),
    (1, 32, 32, (16, 256, 32), 3),
    (2, 30, 22, (9, 250, 32), 2),
    (1, 32, 2, (20, 208, 32), 1),
    # round 3: Z = 64 / 96 (BASELINE configs[4] is 512x512x64): two / three 32-column z tiles whose halo columns are the
    # neighbouring tile's data, zero only at the volume's ends
    (1, 32, 32, (16, 136, 64), 1),
    (1, 32, 32, (16, 136, 64), 3),
    (2, 28, 32, (8, 130, 64), 2),
    (1, 32, 20, (7, 200, 96), 2),
]


@pytest.mark.parametrize("case", PERSIST_CASES)
def test_conv3d_persistent_head_kernel(hip, case):
    B, cin, cout, dims, d = case
    from occdepth_amd.fused import _pad_bias
    torch.manual_seed(cin + cout + d)
    x = torch.randn(B, cin, *dims, device=DEV)
    w = torch.randn(cout, cin, 3, 3, 3, device=DEV) / (cin * 27) ** 0.5
    bias = torch.randn(cout, device=DEV)
    r1 = torch.randn(B, cout, *dims, device=DEV)
    r2 = torch.randn(B, cout, *dims, device=DEV)
    vx, wpk = hip.Vox.from_ncdhw(x), hip.pack_weights(w)
    out = hip.Vox.empty(B, dims, cout, DEV)
    with hip.profile() as prof:
        hip.conv3d(vx, wpk, _pad_bias(bias, cout), cout, (3, 3, 3), out, dilation=(d,) * 3, padding=(d,) * 3)
    assert any(k.startswith("conv3d_c32p") for k in prof.rows), prof.rows.keys()
    ref = F.conv3d(x, w, bias, padding=d, dilation=d)
    assert rel_err(out.ncdhw(), ref) < 2e-5
    if out.cs > cout:
        assert out.buf[..., cout:].abs().max().item() == 0.0
    out2 = hip.Vox.empty(B, dims, cout, DEV)
    hip.conv3d(vx, wpk, _pad_bias(bias, cout), cout, (3, 3, 3), out2, dilation=(d,) * 3, padding=(d,) * 3,
               res1=hip.Vox.from_ncdhw(r1), res2=hip.Vox.from_ncdhw(r2), act_in=hip.ACT_RELU, act_out=hip.ACT_RELU)
    ref2 = F.relu(F.conv3d(F.relu(x), w, bias, padding=d, dilation=d) + r1 + r2)
    assert rel_err(out2.ncdhw(), ref2) < 2e-5
    # the generic kernel (forced by a tile hint) gives the same numbers up to accumulation order
    out3 = hip.Vox.empty(B, dims, cout, DEV)
    hip.conv3d(vx, wpk, _pad_bias(bias, cout), cout, (3, 3, 3), out3, dilation=(d,) * 3, padding=(d,) * 3,
               tile_hint=1)
    assert rel_err(out3.ncdhw(), out.ncdhw()) < 1e-5


@pytest.mark.parametrize("hint", [1, 2, 3, 4, 5, 6, 7, 8, 9])
def test_conv3d_all_variants(hip, hint):
    torch.manual_seed(hint)
    B, cin, cout, dims = 1, 40, 136, (5, 7, 12)
    x = torch.randn(B, cin, *dims, device=DEV)
    w = torch.randn(cout, cin, 3, 3, 3, device=DEV) / (cin * 27) ** 0.5
    ref = F.conv3d(x, w, None, padding=2, dilation=2)
    out = hip.Vox.empty(B, dims, cout, DEV)
    hip.conv3d(hip.Vox.from_ncdhw(x), hip.pack_weights(w), None, cout, (3, 3, 3), out, dilation=(2, 2, 2),
               padding=(2, 2, 2), tile_hint=hint)
    assert rel_err(out.ncdhw(), ref) < 2e-5


def test_conv3d_sigmoid_gemm(hip):
    torch.manual_seed(3)
    from occdepth_amd.fused import gemm_rows
    a = torch.randn(1, 196, 5, 3, 7, device=DEV)
    bm = torch.randn(196, 72, device=DEV)
    out = hip.Vox.empty(1, (5, 3, 7), 72, DEV)
    gemm_rows(hip.Vox.from_ncdhw(a), bm, out, act_in=hip.ACT_SIGMOID)
    ref = torch.sigmoid(a.reshape(196, -1).t()) @ bm
    got = out.buf.reshape(-1, out.cs)[:, :72]
    assert rel_err(got, ref) < 2e-5


def test_crp_product_on_the_split_kernel(hip):
    """sigmoid(P_logits) @ mega at the config-2 size (4096 rows, K = N = 512): `gemm_rows` routes it to K2b's 3-way split with
    the in-workgroup split-K (input sigmoid applied while staging); float32-level against float64."""
    torch.manual_seed(4)
    from occdepth_amd import fused
    a = torch.randn(1, 512, 32, 32, 4, device=DEV) * 2
    bm = torch.randn(512, 512, device=DEV) / 512 ** 0.5
    ref = (torch.sigmoid(a.double().reshape(512, -1).t()) @ bm.double())
    outs = {}
    saved = (fused.BF16X3_SMALLVOL, fused.SMALLVOL_KERNELS)
    fused.SMALLVOL_KERNELS = ((3, 3, 3), (1, 1, 1))          # (opt-in route: OCCDEPTH_BF16X3_SMALLVOL_K1=1)
    try:
        for on in (True, False):
            fused.BF16X3_SMALLVOL = on
            out = hip.Vox.empty(1, (32, 32, 4), 512, DEV)
            with hip.profile() as prof:
                fused.gemm_rows(hip.Vox.from_ncdhw(a), bm, out, act_in=hip.ACT_SIGMOID)
            assert any(k.startswith("conv3d_bf16x3" if on and fused.BF16X3 else "conv3d_igemm") for k in prof.rows), prof.rows.keys()
            outs[on] = out.buf.reshape(-1, out.cs)[:, :512].double()
    finally:
        fused.BF16X3_SMALLVOL, fused.SMALLVOL_KERNELS = saved
    for on, got in outs.items():
        err = float((got - ref).abs().max() / ref.abs().max())
        assert err < 2e-6, (on, err)


def _module_cases():
    from occdepth_amd.models.CRP3D import CPMegaVoxels
    from occdepth_amd.models.DDR import Bottleneck3D
    from occdepth_amd.models.modules import (ASPP, Convblock3d, Downsample, Process, SegmentationHead,
                                             SegmentationHeadCascadeCLS, SegmentationHeadOccludedCLS, Upsample)
    bn = nn.BatchNorm3d
    return {
        "bottleneck_d2": (lambda: Bottleneck3D(32, 8, bn, dilation=[2, 2, 2]), (1, 32, 8, 12, 16)),
        "bottleneck_odd": (lambda: Bottleneck3D(100, 25, bn, dilation=[3, 3, 3]), (1, 100, 7, 9, 15)),
        "process": (lambda: Process(32, bn, 0.1), (2, 32, 8, 8, 8)),
        "downsample": (lambda: Downsample(32, bn, 0.1), (1, 32, 8, 12, 16)),
        "downsample_odd": (lambda: Downsample(24, bn, 0.1), (1, 24, 10, 6, 30)),
        "upsample": (lambda: Upsample(64, 32, bn, 0.1), (1, 64, 4, 6, 8)),
        "upsample_odd": (lambda: Upsample(40, 20, bn, 0.1), (1, 40, 5, 3, 5)),
        "convblock": (lambda: Convblock3d(32, 16, bn, 0.1), (1, 32, 6, 6, 8)),
        "aspp": (lambda: ASPP(32, [1, 2, 3]), (1, 32, 8, 8, 4)),
        "head": (lambda: SegmentationHead(16, 16, 12, [1, 2, 3]), (1, 16, 8, 12, 16)),
        "head_cascade": (lambda: SegmentationHeadCascadeCLS(8, 8, 20, [1, 2, 3]), (1, 8, 8, 8, 32)),
        "head_occluded": (lambda: SegmentationHeadOccludedCLS(16, 16, 20, [1, 2, 3]), (1, 16, 8, 8, 8)),
        "crp": (lambda: CPMegaVoxels(64, (8, 8, 2), bn_momentum=0.1), (1, 64, 8, 8, 2)),
        "crp_odd": (lambda: CPMegaVoxels(32, (5, 3, 5), n_relations=2, bn_momentum=0.1), (2, 32, 5, 3, 5)),
    }


def aten_reference(m, *args):
    """Run the module's ATen (autograd) graph with BatchNorm/Dropout in eval mode."""
    for sub in m.modules():
        sub.training = not isinstance(sub, (nn.BatchNorm2d, nn.BatchNorm3d, nn.Dropout))
    try:
        with torch.no_grad():
            return m(*args)
    finally:
        m.eval()


def compare(got, ref, tol, what=""):
    if isinstance(ref, dict):
        assert set(got) == set(ref)
        for k in ref:
            compare(got[k], ref[k], tol, f"{what}.{k}")
    elif isinstance(ref, (tuple, list)):
        for i, (g, r) in enumerate(zip(got, ref)):
            compare(g, r, tol, f"{what}[{i}]")
    else:
        assert got.shape == ref.shape, what
        assert rel_err(got, ref) < tol, (what, rel_err(got, ref))


@pytest.mark.parametrize("name", ["bottleneck_d2", "bottleneck_odd", "process", "downsample", "downsample_odd",
                                  "upsample", "upsample_odd", "convblock", "aspp", "head", "head_cascade",
                                  "head_occluded", "crp", "crp_odd"])
def test_module_hip_vs_aten(hip, name):
    make, shape = _module_cases()[name]
    torch.manual_seed(1)
    m = randomize_bn(make()).to(DEV).eval()
    x = torch.randn(*shape, device=DEV)
    with torch.no_grad():
        got = m(x)
    ref = aten_reference(m, x)
    compare(got, ref, 2e-4, name)


@pytest.mark.parametrize("case", [(2, 128, 64, (8, 8, 4)), (1, 64, 32, (16, 12, 8)), (1, 40, 20, (5, 3, 5)), (1, 256, 128, (8, 8, 2))])
def test_transposed_conv_phases_one_launch_equals_eight(hip, case):
    """occd_conv3d_fwd_phases (phase = low bits of blockIdx.y, one shared tiling) against the eight separate launches it
    replaces -- the same kernel and K order, so bit for bit unless the merged grid picks another tile variant -- and against
    ATen's ConvTranspose3d + BatchNorm (+ the residual and ReLU of the Upsample block)."""
    from occdepth_amd import fused
    B, cin, cout, dims = case
    torch.manual_seed(cin)
    convt = nn.ConvTranspose3d(cin, cout, 3, stride=2, padding=1, output_padding=1).to(DEV)
    bn = randomize_bn(nn.BatchNorm3d(cout)).to(DEV).eval()
    x = torch.randn(B, cin, *dims, device=DEV)
    res = torch.randn(B, cout, *(2 * n for n in dims), device=DEV)
    with torch.no_grad():
        ref = F.relu(bn(convt(x)) + res)
        plan = fused.ConvTransposePlan(convt, bn)
        vx, vr = hip.Vox.from_ncdhw(x), hip.Vox.from_ncdhw(res)
        saved = (fused.PHASES_ONE_LAUNCH, fused.PHASES_X3)
        try:
            fused.PHASES_ONE_LAUNCH, fused.PHASES_X3 = True, False
            with hip.profile() as prof:
                one = plan(vx, res1=vr, act_out=hip.ACT_RELU).ncdhw()
            convs = [k for k in prof.rows if k.startswith("conv3d")]          # (the first call also packs weights)
            assert len(convs) == 1 and convs[0].startswith("conv3d_igemm_phases"), convs
            fused.PHASES_ONE_LAUNCH = False
            eight = plan(vx, res1=vr, act_out=hip.ACT_RELU).ncdhw()
            # the same merged launch on K2b with the 3-way bf16 split (occd_conv3d_bf16_fwd_phases): float32-level accuracy
            fused.PHASES_ONE_LAUNCH, fused.PHASES_X3 = True, True
            with hip.profile() as prof:
                one_x3 = plan(vx, res1=vr, act_out=hip.ACT_RELU).ncdhw()
            if fused.BF16X3 and cin % 8 == 0:
                assert any(k.startswith("conv3d_bf16x3_phases") for k in prof.rows), list(prof.rows)
        finally:
            fused.PHASES_ONE_LAUNCH, fused.PHASES_X3 = saved
    ref64 = F.relu(bn.double()(F.conv_transpose3d(x.double(), convt.weight.double(), convt.bias.double(), stride=2, padding=1,
                                                  output_padding=1)) + res.double())
    bn.float()
    assert rel_err(one, ref) < 2e-5 and rel_err(eight, ref) < 2e-5
    assert rel_err(one, eight) < 1e-6
    e_x3, e_k2 = rel_err(one_x3.double(), ref64), rel_err(one.double(), ref64)
    print(case, f"vs float64: K2b split {e_x3:.2e}, K2 exact fp32 {e_k2:.2e}")
    assert e_x3 < max(2e-6, 1.5 * e_k2)


@pytest.mark.parametrize("cfg", ["kitti_ps2", "kitti_ps1", "nyu"])
def test_unet3d_hip_vs_aten(hip, cfg):
    torch.manual_seed(2)
    if cfg == "nyu":
        from occdepth_amd.models.unet3d_nyu import UNet3D
        m = UNet3D(12, nn.BatchNorm3d, feature=20, full_scene_size=(20, 12, 20), context_prior=True, n_relations=2)
        x = torch.randn(1, 20, 20, 12, 20, device=DEV)
    else:
        from occdepth_amd.models.unet3d_kitti import UNet3D
        ps = 2 if cfg == "kitti_ps2" else 1
        full = (64, 64, 16) if ps == 2 else (32, 32, 8)
        m = UNet3D(20, nn.BatchNorm3d, full, 16, ps, context_prior=True, cascade_cls=True, occluded_cls=(ps == 1))
        x = torch.randn(1, 16, 32, 32, 8, device=DEV)
    m = randomize_bn(m).to(DEV).eval()
    with torch.no_grad():
        got = m({"x3d": x})
    ref = aten_reference(m, {"x3d": x})
    compare(got, ref, 5e-4, cfg)


@pytest.mark.parametrize("dataset,C,V,P", [("kitti", 64, 2, 1), ("kitti", 32, 2, 5), ("NYU", 100, 2, 1),
                                           ("kitti", 24, 1, 1), ("kitti", 200, 3, 2)])
def test_sfa_hip_vs_aten(hip, dataset, C, V, P):
    from occdepth_amd.models.SFA import SFA
    torch.manual_seed(4)
    scene = (16, 12, 8) if dataset == "kitti" else (10, 6, 8)
    h, w = 23, 31
    N = scene[0] * scene[1] * scene[2]
    m = SFA(scene, dataset, 1).to(DEV).eval()
    x2d = torch.randn(V, C, h, w, device=DEV)
    px = torch.randint(0, w, (V, N, P, 1), device=DEV)
    py = torch.randint(0, h, (V, N, P, 1), device=DEV)
    pix = torch.cat([px, py], -1)
    fov = torch.rand(V, N, P, device=DEV) < 0.6
    with torch.no_grad():
        got = m(x2d, pix, fov)
        ref = m._forward_autograd(x2d, pix, fov)
    assert got.shape == ref.shape
    assert rel_err(got, ref) < 1e-5


def _flosp(dev, n_cams=2):
    from occdepth_amd.models.flosp_depth.flosp_depth import FlospDepth
    torch.manual_seed(5)
    m = FlospDepth(x_bound=[0, 12.8, 0.2], y_bound=[-6.4, 6.4, 0.2], z_bound=[-2, 1.2, 0.2],
                   d_bound=[2.0, 14.0, 0.5], final_dim=(96, 320), downsample_factor=8, output_channels=16,
                   depth_net_conf=dict(in_channels=16, mid_channels=32), scene_size=(64, 64, 16), project_scale=2,
                   return_depth=True)
    m = randomize_bn(m).to(dev).eval()
    feat = torch.randn(1, n_cams, 16, 12, 40, device=dev)
    k = torch.tensor([[180.0, 0, 160.0], [0, 180.0, 48.0], [0, 0, 1]], dtype=torch.float64, device=dev)
    tr = torch.tensor([[0, -1, 0, 0], [0, 0, -1, -0.08], [1, 0, 0, -0.27], [0, 0, 0, 1]], dtype=torch.float32,
                      device=dev)
    tr2 = tr.clone()
    tr2[0, 3] = -0.54
    cam_k = [torch.stack([k] * n_cams)]
    t_v2c = [torch.stack([tr, tr2][:n_cams])]
    ida = torch.eye(4, device=dev)
    ida_flip = ida.clone()
    ida_flip[0, 0] = -1
    ida_flip[0, 3] = 319
    idas = [torch.stack([ida, ida_flip][:n_cams])]
    return m, feat, cam_k, t_v2c, idas


@pytest.mark.parametrize("n_cams", [1, 2])
def test_flosp_depth_hip_vs_aten(hip, n_cams):
    m, feat, cam_k, t_v2c, idas = _flosp(DEV, n_cams)
    with torch.no_grad():
        vox, depth = m(feat, cam_k, t_v2c, idas)
    ref, depth_r = aten_reference(m, feat, cam_k, t_v2c, idas)
    assert vox.shape == ref.shape == (1, 1, 32, 32, 8)
    assert torch.allclose(depth, depth_r)
    assert ref.abs().max() > 1e-3
    assert (vox - ref).abs().max().item() < 2e-5 * ref.abs().max().item() + 1e-7


@pytest.mark.parametrize("act", [None, "relu", "swish", "leaky"])
def test_affine_act_nchw(hip, act):
    torch.manual_seed(7)
    x = torch.randn(2, 19, 13, 21, device=DEV)
    r = torch.randn_like(x)
    s, t = torch.rand(19, device=DEV) + 0.5, torch.randn(19, device=DEV)
    fn = {None: lambda v: v, "relu": F.relu, "swish": lambda v: v * torch.sigmoid(v),
          "leaky": lambda v: F.leaky_relu(v, 0.01)}[act]
    aff = x * s.view(1, -1, 1, 1) + t.view(1, -1, 1, 1)
    assert rel_err(hip.affine_act(x.clone(), s, t, act), fn(aff)) < 1e-6
    assert rel_err(hip.affine_act(x.clone(), s, t, act, res=r), fn(aff) + r) < 1e-6
    assert rel_err(hip.affine_act(x.clone(), s, t, act, res=r, res_first=True), fn(aff + r)) < 1e-6
    x4 = torch.randn(1, 8, 16, 16, device=DEV)                      # float4 path
    assert rel_err(hip.affine_act(x4.clone(), None, None, act), fn(x4)) < 1e-6


@pytest.mark.parametrize("k,stride,hw", [(3, 1, (37, 61)), (3, 2, (37, 61)), (5, 1, (24, 40)), (5, 2, (47, 153))])
def test_dwconv2d_same(hip, k, stride, hw):
    from occdepth_amd.models.efficientnet import Conv2dSame
    torch.manual_seed(k + stride)
    C = 24
    conv = Conv2dSame(C, C, k, stride=stride, groups=C, bias=False).to(DEV)
    x = torch.randn(2, C, *hw, device=DEV)
    s, t = torch.rand(C, device=DEV) + 0.5, torch.randn(C, device=DEV)
    with torch.no_grad():
        ref = conv(x) * s.view(1, -1, 1, 1) + t.view(1, -1, 1, 1)
        ref = ref * torch.sigmoid(ref)
        got = hip.dwconv2d_same(x, conv.weight, s, t, stride, "swish")
    assert got.shape == ref.shape
    assert rel_err(got, ref) < 1e-5


def test_upsample_bilinear_cat(hip):
    torch.manual_seed(9)
    x = torch.randn(2, 7, 12, 39, device=DEV)
    skip = torch.randn(2, 5, 24, 77, device=DEV)
    ref = torch.cat([F.interpolate(x, size=(24, 77), mode="bilinear", align_corners=True), skip], 1)
    got = hip.upsample_bilinear_cat(x, skip)
    assert got.shape == ref.shape
    assert rel_err(got, ref) < 1e-6


def test_unet2d_fused_eval_matches_plain_torch(hip):
    """The eval fast path of the 2-D UNet (fused BN+act, depthwise, upsample+cat kernels) against the
    same module run through plain torch ops (forced by module.training=True with BatchNorm in eval)."""
    from occdepth_amd.models.unet2d import UNet2D
    torch.manual_seed(11)
    m = UNet2D.build(out_feature=16, use_decoder=True, backbone_2d_name="tf_efficientnet_b3_ns", return_up_feats=1)
    m = randomize_bn(m).to(DEV).eval()
    x = torch.randn(1, 3, 74, 122, device=DEV)
    with torch.no_grad():
        got = m(x)
    ref = aten_reference(m, x)
    compare(got, ref, 2e-4, "unet2d")


def test_layout_roundtrip(hip):
    x = torch.randn(2, 37, 5, 7, 9, device=DEV)
    v = hip.Vox.from_ncdhw(x)
    assert v.cs == 40
    assert torch.equal(v.ncdhw(), x)
    assert v.buf[..., 37:].abs().max().item() == 0
    assert torch.equal(hip.nhwc_to_nchw(v), x)
    y = hip.nchw_to_nhwc(x[:, :, 0])
    assert torch.equal(y[..., :37].permute(0, 3, 1, 2), x[:, :, 0])


def test_errors_are_loud(hip):
    x = torch.randn(1, 8, 4, 4, 4)
    with pytest.raises(RuntimeError):
        hip.Vox.from_ncdhw(x)  # CPU tensor: there is no CPU path
    from occdepth_amd.models.DDR import Bottleneck3D
    m = Bottleneck3D(8, 2, nn.BatchNorm3d).eval()
    with pytest.raises(RuntimeError), torch.no_grad():          # the forward-only HIP eval path
        m(x)
