"""-m gpu: the two round-5 kernels that took the last library launches out of the eval trace --
  * occd_stem_conv3x3_nchw: conv_stem (3x3, TF-SAME, stride 2) + bn1 + swish of the EfficientNet encoder in one launch;
  * occd_depthnet_gate: DepthNet's Mlp(scaled pixel size) -> SELayer gate in one launch
against ATen float64 on the CPU, and DepthNet end to end in the fused form against its own library form."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(scope="module")
def hip():
    from occdepth_amd import hip as h
    h.load()
    return h


@pytest.mark.parametrize("shape", [(2, 370, 1220, 64, 2), (1, 96, 320, 32, 2), (3, 37, 45, 40, 2), (1, 33, 64, 24, 1),
                                   (2, 8, 7, 64, 2)])
@pytest.mark.parametrize("act", ["swish", "relu", None])
def test_stem_conv3x3(hip, shape, act):
    """Every SAME-padding parity (even / odd extents, stride 1 / 2), channel counts that are not a multiple of 16, partial
    64-pixel column blocks; against F.conv2d in float64 with the padding the TensorFlow rule prescribes."""
    B, H, W, cout, stride = shape
    g = torch.Generator().manual_seed(H * 3 + W)
    x = torch.randn(B, 3, H, W, generator=g)
    w = torch.randn(cout, 3, 3, 3, generator=g) * 0.3
    scale = torch.rand(cout, generator=g) + 0.5
    shift = torch.randn(cout, generator=g)
    Ho, Wo = -(-H // stride), -(-W // stride)
    ph, pw = max((Ho - 1) * stride + 3 - H, 0), max((Wo - 1) * stride + 3 - W, 0)
    ref = F.conv2d(F.pad(x.double(), [pw // 2, pw - pw // 2, ph // 2, ph - ph // 2]), w.double(), None, stride)
    ref = ref * scale.double().view(1, -1, 1, 1) + shift.double().view(1, -1, 1, 1)
    ref = ref * torch.sigmoid(ref) if act == "swish" else F.relu(ref) if act == "relu" else ref
    got = hip.stem_conv3x3(x.to(DEV), w.to(DEV), scale.to(DEV), shift.to(DEV), stride, act)
    assert got.shape == ref.shape
    err = float((got.cpu().double() - ref).abs().max() / ref.abs().max())
    assert err < (3e-6 if act == "swish" else 2e-6), err
    plain = hip.stem_conv3x3(x.to(DEV), w.to(DEV), None, None, stride, None)          # no affine
    ref0 = F.conv2d(F.pad(x.double(), [pw // 2, pw - pw // 2, ph // 2, ph - ph // 2]), w.double(), None, stride)
    assert float((plain.cpu().double() - ref0).abs().max() / ref0.abs().max()) < 2e-6


def test_encoder_uses_the_fused_stem(hip):
    """The eval fast path of the encoder launches `stem_conv3x3` (no MIOpen convolution / BatchNorm for the stem) and its
    tapped features equal the module-by-module form (OCCDEPTH_STEM_FUSED=0)."""
    from occdepth_amd.models import unet2d
    from occdepth_amd.models.efficientnet import EfficientNet
    torch.manual_seed(0)
    backend = EfficientNet("tf_efficientnet_b3_ns")
    backend.global_pool, backend.classifier = torch.nn.Identity(), torch.nn.Identity()      # (as unet2d.py:238-239 does)
    enc = unet2d.Encoder(backend).to(DEV).eval()
    for m in enc.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.normal_(0, 0.1)
            m.running_var.uniform_(0.5, 1.5)
    x = torch.randn(2, 3, 96, 160, device=DEV)
    with torch.no_grad():
        with hip.profile() as prof:
            f1 = enc(x)
        saved = unet2d.STEM_FUSED
        unet2d.STEM_FUSED = False
        try:
            f0 = enc(x)
        finally:
            unet2d.STEM_FUSED = saved
    assert any(k.startswith("stem_conv3x3") for k in prof.rows), prof.rows.keys()
    assert f1[1] is None and f1[2] is None and f0[1] is not None
    for i in (3, 4, 5, 6, 8, 11):
        assert float((f1[i] - f0[i]).abs().max()) <= 2e-4 * float(f0[i].abs().max()), i


@pytest.mark.parametrize("C,images", [(128, 2), (96, 1), (40, 5)])
def test_depthnet_gate(hip, C, images):
    from occdepth_amd.models.flosp_depth.flosp_depth import Mlp, SELayer
    torch.manual_seed(C)
    mlp, se = Mlp(1, C, C), SELayer(C)
    k = torch.zeros(images, 4, 4)
    k[:, 0, 0] = torch.rand(images) * 300 + 500
    k[:, 1, 1] = torch.rand(images) * 300 + 500
    k[:, 0, 2], k[:, 1, 2], k[:, 2, 2], k[:, 3, 3] = 610.0, 185.0, 1.0, 1.0
    sps = torch.sqrt((1 / k[:, 0, 0].double()) ** 2 + (1 / k[:, 1, 1].double()) ** 2) * 1000.0
    with torch.no_grad():
        m64, s64 = Mlp(1, C, C).double(), SELayer(C).double()
        m64.load_state_dict({n: v.double() for n, v in mlp.state_dict().items()})
        s64.load_state_dict({n: v.double() for n, v in se.state_dict().items()})
        x_se = m64(sps.reshape(-1, 1))[..., None, None]
        ref = s64.gate(s64.conv_expand(s64.act1(s64.conv_reduce(x_se)))).reshape(images, C)
    mlp, se = mlp.to(DEV), se.to(DEV)
    got_k = hip.depthnet_gate(mlp, se, images, intrins=k.to(DEV)).cpu().double()
    got_s = hip.depthnet_gate(mlp, se, images, sps=sps.float().to(DEV)).cpu().double()
    assert float((got_k - ref).abs().max()) < 2e-6 and float((got_s - ref).abs().max()) < 2e-6
    with pytest.raises(RuntimeError):
        hip.depthnet_gate(mlp, se, images)


def test_depthnet_fused_equals_library_form(hip):
    """DepthNet in eval on the GPU: the round-5 default (K10 convolutions + one-launch gate: `wino_conv3x3`, `depthnet_gate`
    in the profile, no ATen convolution) against the library form (OCCDEPTH_DEPTHNET_K10=0) on the same weights."""
    from occdepth_amd.models.flosp_depth import flosp_depth as fd
    torch.manual_seed(3)
    net = fd.DepthNet(128, 128, 64, 104).to(DEV).eval()
    for m in net.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.normal_(0, 0.1)
            m.running_var.uniform_(0.5, 1.5)
    x = torch.randn(2, 128, 47, 153, device=DEV)
    k = torch.eye(4).repeat(1, 2, 1, 1).to(DEV)
    k[0, :, 0, 0], k[0, :, 1, 1] = 707.09, 707.09
    res = {}
    for on in (True, False):
        saved = fd.DEPTHNET_K10
        fd.DEPTHNET_K10 = on
        try:
            with torch.no_grad(), hip.profile() as prof:
                y = net(x=x, sweep_intrins=k)
            res[on] = (y, {t.split(":")[0] for t in prof.rows})        # (the rows are collected when the profile closes)
        finally:
            fd.DEPTHNET_K10 = saved
    assert {"wino_conv3x3", "depthnet_gate"} <= res[True][1] and "depthnet_gate" not in res[False][1]
    a, b = res[True][0], res[False][0]
    assert a.shape == b.shape == (2, 104, 47, 153)
    assert float((a - b).abs().max()) <= 3e-4 * float(b.abs().max())


# (C, Cr, batch, pool partials per plane, plane size): the B7 stages (2304 / 96 at 1/32 ... 32 / 8 at 1/2) and ragged ones
SE_CASES = [(2304, 96, 2, 1, 468), (3840, 160, 2, 1, 468), (1344, 56, 2, 2, 1848), (960, 40, 2, 2, 1848), (480, 20, 2, 8, 7191),
            (288, 12, 2, 28, 28365), (192, 8, 2, 111, 112850), (32, 8, 2, 111, 112850), (70, 4, 3, 5, 100), (4096, 192, 1, 3, 77)]


@pytest.mark.parametrize("case", SE_CASES)
def test_se_gate_one_launch_equals_two_launches_and_float64(hip, case):
    """Round 6: the squeeze-excite gate as ONE launch (se_fused_kernel: reduce outputs published as self-validating words,
    polled by the expand workgroups) -- bit-identical to the reduce + expand launches it replaces, right against float64
    (geffnet SqueezeExcite: mean -> conv_reduce -> swish -> conv_expand -> sigmoid), and replayable from a hipGraph (the
    sequence number lives on the device).  Repeated 40 times back to back: every launch must see ITS OWN reduce results."""
    C, Cr, batch, nblk, S = case
    g = torch.Generator().manual_seed(C + Cr)
    wr, br = torch.randn(Cr, C, generator=g) / C ** 0.5, torch.randn(Cr, generator=g)
    we, be = torch.randn(C, Cr, generator=g) / Cr ** 0.5, torch.randn(C, generator=g)
    wr, br, we, be = (t.to(DEV) for t in (wr, br, we, be))
    lib = hip.load()

    def gates(fused, parts):
        old = lib.occd_se_gate_set_fused(1 if fused else 0)
        try:
            with hip.profile() as prof:
                out = [hip.se_gate(p, S, batch, wr, br, we, be) for p in parts]
                torch.cuda.synchronize()
        finally:
            lib.occd_se_gate_set_fused(old)
        return out, sum(v["launches"] for k, v in prof.rows.items() if k.startswith("se_gate"))

    parts = [(torch.randn(batch * C, nblk, generator=g) * (S / nblk) ** 0.5 + 0.1 * i).to(DEV) for i in range(40)]
    fused, n1 = gates(True, parts)
    plain, n2 = gates(False, parts)
    assert n1 == n2 == 40                                              # (one profiler scope per call in both forms)
    for i, (a, b) in enumerate(zip(fused, plain)):
        assert torch.equal(a, b), (case, i, float((a - b).abs().max()))
    # the squeezed activations r = swish(Wr mean + br), which the TRAINING backward reads (occd_se_bwd), come out of both forms
    rs = {}
    for mode in (1, 0):
        old = lib.occd_se_gate_set_fused(mode)
        r = torch.full((batch, Cr), float("nan"), device=DEV)
        gate = torch.empty((batch, C), device=DEV)
        hip._check(lib.occd_se_gate(parts[7].data_ptr(), wr.data_ptr(), br.data_ptr(), we.data_ptr(), be.data_ptr(), r.data_ptr(),
                                    gate.data_ptr(), batch, C, Cr, nblk, S, None), "occd_se_gate")
        torch.cuda.synchronize()
        lib.occd_se_gate_set_fused(old)
        rs[mode] = r
    assert torch.equal(rs[1], rs[0]) and bool(torch.isfinite(rs[1]).all())
    mean = parts[7].double().cpu().view(batch, C, nblk).sum(-1) / S
    r = mean @ wr.double().cpu().t() + br.double().cpu()
    r = r * torch.sigmoid(r)
    ref = torch.sigmoid(r @ we.double().cpu().t() + be.double().cpu())
    assert float((fused[7].cpu().double() - ref).abs().max()) < 2e-6
    # captured: replays with new partials in the same static buffer
    static = parts[0].clone()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        hip.se_gate(static, S, batch, wr, br, we, be)
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        out = hip.se_gate(static, S, batch, wr, br, we, be)
    for i in (3, 11, 12):
        static.copy_(parts[i])
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(out, plain[i]), (case, "replay", i)
