"""K16 (csrc/gemm_x3.hip, occd_gemm_f32x3): the row-major float32 GEMM with the 3-way bf16 split against float64 on the
UN-rounded operands -- float32 level (bound 2e-6 of the output maximum; torch.matmul's float32 result is measured beside it) --
on the shapes the 2-D network uses it for (tap GEMMs of the decoder levels, Winograd-domain products, expand 1x1
convolutions; reduced in the long dimension so the float64 reference stays cheap) and on the ragged ones: rows of odd
length (dword-aligned 16-byte loads, a last chunk that crosses the end of the row), M / N / K tails, strided views,
every tile variant, bias + activation epilogues."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(scope="module")
def hip():
    from occdepth_amd import hip as h
    h.load()
    return h


# name: (batch or None, M, N, K, A batched?)
SHAPES = {
    "tap_1_16": (2, 11520 // 8, 574, 2560, False),        # 9 Cout x Cup at 14 x 41 pixels (M cut to 1/8)
    "tap_1_4": (2, 2880, 7191 // 4, 640, False),          # N odd
    "tap_1_1": (2, 720, 225700 // 64, 160, False),        # short K, M = 2.8 tiles of 256
    "wino_1_16": (16, 936, 1280 // 4, 1280, True),        # V[xi] (T x Cin) . U[xi] (Cin x Cout)
    "expand_1_32": (2, 3840 // 4, 468, 640, False),
    "expand_k48": (2, 288, 1848, 48, False),              # K tail: 48 = 32 + 16
    "tiny": (None, 40, 7, 8, False),                      # single partial tile everywhere, N % 4 = 3
    "n_tail_1": (1, 70, 133, 24, False),                  # N % 4 = 1, K tail of 8
    "n_tail_2": (3, 33, 130, 104, True),                  # N % 4 = 2
}


def run(hip, a, b, **kw):
    return hip.gemm_x3(a, b, **kw)


@pytest.mark.parametrize("packed", [None, "a", "b"], ids=["split_on_the_fly", "A_presplit", "B_presplit"])
@pytest.mark.parametrize("name", list(SHAPES))
def test_gemm_x3_vs_float64(hip, name, packed):
    """packed: the static operand split once into its fragment image (hip.GemmPacked, occd_gemm_x3_pack) and read straight
    from L2 -- what the model does with its weights (A for the tap GEMMs / expand convolutions, B for the Winograd domain)."""
    batch, M, N, K, a_batched = SHAPES[name]
    g = torch.Generator().manual_seed(sum(map(ord, name)))
    bshape = (K, N) if batch is None else (batch, K, N)
    ashape = (batch, M, K) if a_batched else (M, K)
    a = torch.randn(*ashape, generator=g) * (1.0 / K ** 0.5)
    b = torch.randn(*bshape, generator=g) * 3.0
    ref = torch.matmul(a.double(), b.double())
    ad, bd = a.to(DEV), b.to(DEV)
    xa = hip.GemmPacked(ad, "a") if packed == "a" else ad
    xb = hip.GemmPacked(bd, "b") if packed == "b" else bd
    got = run(hip, xa, xb).cpu().double()
    assert got.shape == ref.shape
    if packed is None:                                     # the wave-specialised kernel on the same operands
        got_ws = run(hip, xa, xb, tile_hint=6).cpu().double()
        assert float((got_ws - ref).abs().max() / ref.abs().max()) < max(2e-6, 1.25 * float(
            (torch.matmul(ad, bd).cpu().double() - ref).abs().max() / ref.abs().max())), name
    scale = ref.abs().max()
    err = float((got - ref).abs().max() / scale)
    err32 = float((torch.matmul(ad, bd).cpu().double() - ref).abs().max() / scale)
    print(f"gemm_x3 {name} ({packed or 'no'} pre-split): max err {err:.2e} (torch.matmul fp32 {err32:.2e})")
    # float32 accumulation over K terms: the bound grows with sqrt(K) (measured 1.9e-6 at K = 2560, the library's 2.0e-6)
    assert err < max(2e-6, 1.25 * err32), (name, err, err32)


# 5 = 64 x 256 tile, 6 = the wave-specialised 256 x 128 kernel (K16w), 7 = 64 x 64 with the in-workgroup split-K (4 K groups)
@pytest.mark.parametrize("K", [72, 392])                   # 2 steps + a tail of 8 (split-K: two idle K groups); 12 steps + a tail
@pytest.mark.parametrize("hint", [1, 2, 3, 4, 5, 6, 7])
def test_gemm_x3_every_tile_variant_and_epilogue(hip, hint, K):
    g = torch.Generator().manual_seed(hint)
    batch, M, N = 2, 300, 333                              # M, N tails in every variant
    a = torch.randn(M, K, generator=g) / K ** 0.5
    b = torch.randn(batch, K, N, generator=g)
    bias = torch.randn(M, generator=g)
    lin = torch.matmul(a.double(), b.double()) + bias.double().view(1, -1, 1)
    for act, ref in ((None, lin), ("swish", lin * torch.sigmoid(lin)), ("leaky", torch.where(lin > 0, lin, lin * 0.2))):
        got = run(hip, a.to(DEV), b.to(DEV), bias=bias.to(DEV), act=act, slope=0.2, tile_hint=hint).cpu().double()
        err = float((got - ref).abs().max() / ref.abs().max())
        assert err < (3e-6 if act == "swish" else 2e-6), (hint, act, err)      # (swish: hardware exp2 / rcp, ~3 ulp)
    forms = [(a.to(DEV), b.to(DEV))]
    if hint < 6:                                           # (K16w and the split-K form take float32 operands only)
        forms += [(hip.GemmPacked(a.to(DEV), "a"), b.to(DEV)), (a.to(DEV), hip.GemmPacked(b.to(DEV), "b"))]
    for xa, xb in forms:
        got = run(hip, xa, xb, tile_hint=hint).cpu().double()
        assert float((got - (lin - bias.double().view(1, -1, 1))).abs().max() / lin.abs().max()) < 2e-6


# K16p (tile_hint 8): (batch, M, N, K, A batched?) -- row counts that give one and several rounds per wave, several row ranges (few
# column panels), idle sub-tiles and idle waves; K tails of 8 (K % 16 = 8) and the largest K that fits LDS; N tails of every
# residue and N < 64
PANEL_SHAPES = [(2, 720, 1100, 160, False), (2, 960, 1848, 160, False), (2, 1344, 460, 224, False), (2, 480, 7191 // 4, 80, False),
                (2, 288, 3000, 48, False), (1, 192, 5003, 32, False), (1, 130, 70, 352, False), (3, 257, 61, 72, True),
                (1, 40, 6, 8, False), (2, 2304 // 4, 468, 264, False),
                # 32-column panels: K beyond the 64-column form (<= 848), and the few-pixel launches hint 0 gives them to
                (2, 576, 468, 384, False), (2, 960, 468, 640, False), (1, 256, 203, 848, False), (2, 512, 1848, 224, False)]


@pytest.mark.parametrize("shape", PANEL_SHAPES, ids=lambda s: "x".join(map(str, s[:4])))
def test_gemm_x3_panel_vs_float64(hip, shape):
    """The panel-stationary kernel (pre-split A, the whole-K panel of 64 B columns resident in LDS) against float64, plain and
    with the bias + activation epilogues, into a strided output; hint 0 picks it for >= 256 rows (profile row name)."""
    batch, M, N, K, a_batched = shape
    g = torch.Generator().manual_seed(M + N + K)
    a = torch.randn(*((batch, M, K) if a_batched else (M, K)), generator=g) / K ** 0.5
    bbig = torch.randn(batch, K + 2, N + 5, generator=g) * 3.0
    b = bbig[:, 1:1 + K, 2:2 + N]                            # a view: ldb > N, dword-aligned rows
    bias = torch.randn(M, generator=g)
    lin = torch.matmul(a.double(), b.double())
    pa, bd = hip.GemmPacked(a.to(DEV), "a"), bbig.to(DEV)[:, 1:1 + K, 2:2 + N]
    err32 = float((torch.matmul(a.to(DEV), bd).cpu().double() - lin).abs().max() / lin.abs().max())
    with hip.profile() as prof:
        got = hip.gemm_x3(pa, bd, tile_hint=8)
    assert list(prof.rows)[0].startswith("gemm_f32x3_panel"), prof.rows.keys()
    assert float((got.cpu().double() - lin).abs().max() / lin.abs().max()) < max(2e-6, 1.25 * err32)
    with hip.profile() as prof:
        auto = hip.gemm_x3(pa, bd)
    assert list(prof.rows)[0].startswith("gemm_f32x3_panel" if M >= 256 else "gemm_f32x3_preA"), prof.rows.keys()
    assert float((auto.cpu().double() - lin).abs().max() / lin.abs().max()) < max(2e-6, 1.25 * err32)
    linb = lin + bias.double().view(1, -1, 1)
    obig = torch.full((batch, M + 1, N + 3), 7.0, device=DEV)
    for act, ref in ((None, linb), ("swish", linb * torch.sigmoid(linb)), ("leaky", torch.where(linb > 0, linb, linb * 0.2))):
        out = obig[:, 1:, 3:]
        hip.gemm_x3(pa, bd, bias=bias.to(DEV), act=act, slope=0.2, tile_hint=8, out=out)
        err = float((out.cpu().double() - ref).abs().max() / ref.abs().max())
        assert err < max(3e-6 if act == "swish" else 2e-6, 1.25 * err32), (act, err)
        assert float(obig[:, 0].min()) == 7.0 and float(obig[:, :, :3].min()) == 7.0 == float(obig[:, :, :3].max())
    # bit-identical to itself (no atomics, fixed order), and the same value whichever row range / wave owns a row
    assert torch.equal(hip.gemm_x3(pa, bd, tile_hint=8), got)


def test_gemm_x3_panel_limits(hip):
    a = torch.randn(256, 856, device=DEV)                    # K = 856: 864 k x 192 B > 160 KB of LDS
    b = torch.randn(2, 856, 100, device=DEV)
    with pytest.raises(RuntimeError):
        hip.gemm_x3(hip.GemmPacked(a, "a"), b, tile_hint=8)
    with hip.profile() as prof:
        y = hip.gemm_x3(hip.GemmPacked(a, "a"), b)           # hint 0: the barrier-phased PRE = 1 kernel
    assert list(prof.rows)[0].startswith("gemm_f32x3_preA")
    assert float((y - torch.matmul(a, b)).abs().max()) < 1e-3
    a2 = torch.randn(256, 64, device=DEV)
    b2 = torch.randn(2, 64, 100, device=DEV)
    with pytest.raises(RuntimeError):                        # float32 A / a residual: not this kernel
        hip.gemm_x3(a2, b2, tile_hint=8)
    with pytest.raises(RuntimeError):
        hip.gemm_x3(hip.GemmPacked(a2, "a"), b2, tile_hint=8, res=torch.zeros(2, 256, 100, device=DEV))
    # hip.matmul hands the image to K16 only where the panel kernel applies
    w = hip.matmul_operand(a2, "a")
    assert w[1] is not None and hip.matmul_operand(a2[:64].contiguous(), "a")[1] is None        # (fewer than 256 rows: never)
    wbig = hip.matmul_operand(a, "a")                        # K = 856: beyond K16p; the image exists for K16's pre-split form ...
    assert wbig[1] is not None
    with hip.profile() as prof:
        yb = hip.matmul(wbig, b)                             # ... which only the chip-filling launches take: this one stays float32
    assert [k.split(":")[0] for k in prof.rows] == ["gemm_f32x3"]
    assert float((yb - torch.matmul(a, b)).abs().max()) < 1e-3
    res = torch.randn(2, 256, 100, device=DEV)
    with hip.profile() as prof:
        y0 = hip.matmul(w, b2)
        y1 = hip.matmul(w, b2, res=res)
    names = sorted(k.split(":")[0] for k in prof.rows)
    assert names == ["gemm_f32x3", "gemm_f32x3_panel"], names
    assert float((y1 - res - y0).abs().max()) < 1e-5


def test_matmul_takes_the_presplit_form_on_large_long_k_launches(hip):
    """hip.matmul on a static left operand: K16's PRE = 1 form (weight fragments a full 32-k step ahead) where 256 x 128 tiles
    fill the chip and K >= 512 -- the tap GEMMs of the 1/4 ... 1/16 decoder levels (here: 1/8 with the pixel count cut) --
    against float64."""
    g = torch.Generator().manual_seed(5)
    M, N, K, batch = 5760, 1848, 1280, 2
    a = (torch.randn(M, K, generator=g) / K ** 0.5).to(DEV)
    b = torch.randn(batch, K, N, generator=g).to(DEV)
    w = hip.matmul_operand(a, "a")
    assert w[1] is not None and hip._presplit_launch(w[1], b) and not hip._panel_launch(w[1], b)
    with hip.profile() as prof:
        y = hip.matmul(w, b)
    assert [k.split(":")[0] for k in prof.rows] == ["gemm_f32x3_preA"], prof.rows.keys()
    rows = torch.arange(0, M, 37, device=DEV)                # (a float64 reference of every 37th row)
    ref = torch.matmul(a[rows].double(), b.double())
    assert float((y[:, rows].double() - ref).abs().max() / ref.abs().max()) < 2.5e-6
    assert torch.equal(hip.matmul(w, b), y)


def test_gemm_x3_strided_views_and_output(hip):
    """Operands as views (leading dimensions larger than the logical sizes) and a caller-provided strided output: the
    forms the decoder uses -- x[b] planes of an NCHW tensor as (K, N) with ldb = H W, weights as a slice of a wider matrix."""
    g = torch.Generator().manual_seed(11)
    M, N, K, batch = 96, 150, 64, 2
    abig = torch.randn(M + 5, K + 24, generator=g).to(DEV)
    bbig = torch.randn(batch, K + 3, N + 9, generator=g).to(DEV)
    a, b = abig[2:2 + M, 8:8 + K], bbig[:, 1:1 + K, 3:3 + N]
    assert not a.is_contiguous() and not b.is_contiguous() and hip.gemm_x3_supported(a, b)
    obig = torch.full((batch, M + 2, N + 6), 7.0, device=DEV)
    out = obig[:, 1:1 + M, 2:2 + N]
    hip.gemm_x3(a, b, out=out)
    ref = torch.matmul(a.double().cpu(), b.double().cpu())
    assert float((out.cpu().double() - ref).abs().max() / ref.abs().max()) < 2e-6
    border = obig.clone()
    border[:, 1:1 + M, 2:2 + N] = 7.0
    assert torch.equal(border, torch.full_like(obig, 7.0))          # nothing written outside the view
    assert not hip.gemm_x3_supported(a[:, :63], b[:, :63])          # K % 8
    assert not hip.gemm_x3_supported(abig[:, 1:1 + K], bbig[:, :K]) # rows of A not 16-byte aligned


def test_gemm_x3_sigmoid_on_a_batched_into_strided_slices(hip):
    """The CRP's relation products (occdepth/models/CRP3D.py:80: torch.bmm(sigmoid(P_logits_r), mega) for r = 0 .. 3) as ONE
    batched launch: A = the relations' logit rows (sigmoid applied while staged), B shared over the batch, C = channel
    slices of wider concat rows (stride_c = slice width, ldc = row width) -- against float64; nothing outside the slices is
    written."""
    g = torch.Generator().manual_seed(21)
    R, N, M, C2, C = 4, 520, 136, 72, 40                    # (N, M, C2 not multiples of the tiles)
    m_cs = 136
    logits = (torch.randn(R, N, m_cs, generator=g) * 2.0).to(DEV)
    mega = torch.randn(M, C2 + 8, generator=g).to(DEV)
    Ct = C + R * C2
    cat = torch.full((N, Ct), 7.0, device=DEV)
    a_op = torch.as_strided(logits, (R, N, M), (N * m_cs, m_cs, 1), 0)
    out = torch.as_strided(cat, (R, N, C2), (C2, Ct, 1), C)
    with hip.profile() as prof:
        hip.gemm_x3(a_op, mega[:, :C2], out=out, sigmoid_a=True)
    assert any(k.startswith("gemm_f32x3") for k in prof.rows)
    ref = torch.matmul(torch.sigmoid(logits[..., :M].double().cpu()), mega[:, :C2].double().cpu())        # (R, N, C2)
    got = cat.cpu().double()
    assert torch.equal(got[:, :C], torch.full((N, C), 7.0, dtype=torch.float64))
    for r in range(R):
        sl = got[:, C + r * C2:C + (r + 1) * C2]
        assert float((sl - ref[r]).abs().max() / ref[r].abs().max()) < 2e-6, r
    with pytest.raises(RuntimeError):
        hip.gemm_x3(hip.GemmPacked(logits[0, :, :M].contiguous(), "a"), mega[:, :C2], sigmoid_a=True)


def test_gemm_x3_rejects_bad_arguments(hip):
    import ctypes
    lib = hip.load()
    assert lib.occd_gemm_f32x3(None, None) == -1
    q = hip.GemmArgs()
    assert lib.occd_gemm_f32x3(ctypes.byref(q), None) == -1
    # an `out` / `res` whose batch extent does not match the operands' (the kernel trusts batch x stride_c)
    a = torch.randn(40, 16, device=DEV)
    b = torch.randn(3, 16, 24, device=DEV)
    for bad in (torch.empty(1, 40, 24, device=DEV), torch.empty(40, 24, device=DEV), torch.empty(2, 40, 24, device=DEV)):
        with pytest.raises(RuntimeError, match="out must be"):
            hip.gemm_x3(a, b, out=bad)
    with pytest.raises(RuntimeError, match="res must be"):
        hip.gemm_x3(a, b, res=torch.empty(1, 40, 24, device=DEV))
    ok = torch.empty(3, 40, 24, device=DEV)
    hip.gemm_x3(a, b, out=ok)
    assert float((ok.cpu().double() - torch.matmul(a.double().cpu(), b.double().cpu())).abs().max()) < 1e-4
    out2 = torch.empty(40, 24, device=DEV)                 # 2-D out: only when neither operand is batched
    hip.gemm_x3(a, b[0], out=out2)
    assert torch.equal(out2, ok[0])


NT_SHAPES = {
    "dw_project_1_32": (2, 384, 2304, 468),          # dW of 2304 -> 384 at 12 x 39 pixels: (Cout, HW) x (Cin, HW)
    "dw_expand_1_4": (2, 288, 48, 28365 // 8),       # K odd, rows of odd length
    "dw_tiny": (3, 40, 24, 7),                       # K < 8: the element-wise tail only
    "dw_tail_5": (1, 70, 130, 37),                   # K = 32 + 5
}


@pytest.mark.parametrize("name", list(NT_SHAPES))
@pytest.mark.parametrize("hint", [0, 1, 2])
def test_gemm_x3_nt_vs_float64(hip, name, hint):
    """K16t (occd_gemm_f32x3_nt): C = A . B^T with both operands k-contiguous, any alignment, any K (weight gradients of
    pointwise convolutions) against float64, incl. operands that are views into wider buffers (odd row strides)."""
    batch, M, N, K = NT_SHAPES[name]
    g = torch.Generator().manual_seed(sum(map(ord, name)))
    abig = torch.randn(batch, M, K + 3, generator=g).to(DEV)
    bbig = torch.randn(batch, N, K + 1, generator=g).to(DEV)
    a, b = abig[:, :, 2:2 + K], bbig[:, :, 1:1 + K]
    ref = torch.matmul(a.double().cpu(), b.double().cpu().transpose(1, 2))
    got = hip.gemm_x3_nt(a, b, tile_hint=hint, splits=1, reduce=False)[:, 0].cpu().double()
    err = float((got - ref).abs().max() / ref.abs().max())
    # split-K (the default: a small matrix reduced over many pixels) and the batch sum
    for splits in (None, 3):
        summed = hip.gemm_x3_nt(a, b, tile_hint=hint, splits=splits).cpu().double()
        assert float((summed - ref.sum(0)).abs().max() / ref.sum(0).abs().max()) < 3e-6, (name, hint, splits)
    err32 = float((torch.matmul(a, b.transpose(1, 2)).cpu().double() - ref).abs().max() / ref.abs().max())
    print(f"gemm_x3_nt {name} hint {hint}: max err {err:.2e} (torch.matmul fp32 {err32:.2e})")
    assert err < max(2e-6, 1.25 * err32), (name, hint, err, err32)


@pytest.mark.parametrize("shape", [(2, 48, 288, 23, 77), (2, 2304, 384, 12, 39), (1, 32, 32, 31, 45)])
def test_pointwise_conv_autograd_matches_aten(hip, shape):
    """hip._PwConvFn (training path of the MBConv 1x1 convolutions): forward, data gradient and weight gradient against
    ATen float64 on the CPU."""
    import torch.nn.functional as F
    B, cin, cout, H, W = shape
    g = torch.Generator().manual_seed(cin + cout)
    x = torch.randn(B, cin, H, W, generator=g)
    w = torch.randn(cout, cin, 1, 1, generator=g) / cin ** 0.5
    gy = torch.randn(B, cout, H, W, generator=g)
    xr, wr = x.double().requires_grad_(True), w.double().requires_grad_(True)
    F.conv2d(xr, wr).backward(gy.double())
    xd, wd = x.to(DEV).requires_grad_(True), w.to(DEV).requires_grad_(True)
    y = hip.pw_conv_autograd(xd, wd)
    y.backward(gy.to(DEV))
    for got, ref, what in ((y, F.conv2d(x.double(), w.double()), "y"), (xd.grad, xr.grad, "dx"), (wd.grad, wr.grad, "dw")):
        err = float((got.detach().cpu().double() - ref.detach()).abs().max() / ref.detach().abs().max())
        assert err < 3e-6, (shape, what, err)


def test_gemm_plain_bf16_mode_equals_float64_on_rounded_operands(hip):
    """`plain_bf16` (the bf16 training mode of K16 / K16t: operands rounded to ONE bf16 term, fp32 accumulate): against float64
    evaluated on the SAME bf16-rounded operands the products are exact, so the bound is float32 accumulation (2e-6), i.e. an
    index / layout bug cannot hide behind 2^-8."""
    g = torch.Generator().manual_seed(5)
    batch, M, N, K = 2, 200, 333, 104
    r = lambda t: t.to(torch.bfloat16).to(torch.float32)
    a = r(torch.randn(M, K, generator=g) / K ** 0.5)
    b = r(torch.randn(batch, K, N, generator=g))
    ref = torch.matmul(a.double(), b.double())
    for hint in (0, 2, 4):
        got = hip.gemm_x3(a.to(DEV), b.to(DEV), tile_hint=hint, plain_bf16=True).cpu().double()
        assert float((got - ref).abs().max() / ref.abs().max()) < 2e-6, hint
    a3 = r(torch.randn(batch, 72, 517, generator=g))
    b3 = r(torch.randn(batch, 40, 517, generator=g))
    ref = torch.matmul(a3.double(), b3.double().transpose(1, 2)).sum(0)
    got = hip.gemm_x3_nt(a3.to(DEV), b3.to(DEV), plain_bf16=True).cpu().double()
    assert float((got - ref).abs().max() / ref.abs().max()) < 2e-6


# project convolutions of the MBConv blocks on K16: squeeze-excite gate on the B rows (occd_gemm_args.scale_k), BatchNorm shift,
# the block's skip in the epilogue (.res) -- (batch, Cout = M, pixels = N, Cin = K, skip?)
@pytest.mark.parametrize("shape", [(2, 48, 2837, 288, True), (2, 384, 468, 2304, True), (2, 80, 1799, 480, False),
                                   (1, 224, 463, 1344, True), (2, 32, 2001, 32, True), (3, 70, 131, 104, True)])
@pytest.mark.parametrize("hint", [0, 4, 7])
def test_gemm_x3_gate_and_skip_epilogue(hip, shape, hint):
    batch, M, N, K, skip = shape
    g = torch.Generator().manual_seed(M + N + K)
    w = torch.randn(M, K, generator=g) / K ** 0.5
    y = torch.randn(batch, K, N, generator=g) * 2.0
    gate = torch.sigmoid(torch.randn(batch, K, generator=g))
    shift = torch.randn(M, generator=g)
    res = torch.randn(batch, M, N, generator=g) if skip else None
    gated = (y * gate.unsqueeze(-1))                      # float32 rounding of the product, as the reference's x * gate
    ref = torch.matmul(w.double(), gated.double()) + shift.double().view(-1, 1)
    if skip:
        ref = ref + res.double()
    got = hip.gemm_x3(w.to(DEV), y.to(DEV), bias=shift.to(DEV), k_scale=gate.to(DEV), res=res.to(DEV) if skip else None,
                      tile_hint=hint).cpu().double()
    err = float((got - ref).abs().max() / ref.abs().max())
    print(f"gemm_x3 gate+skip {shape} hint {hint}: {err:.2e}")
    assert got.shape == ref.shape and err < 2e-6, (shape, err)
    with pytest.raises(RuntimeError):                      # the wave-specialised kernel has no gate / skip path
        hip.gemm_x3(w.to(DEV), y.to(DEV), k_scale=gate.to(DEV), tile_hint=6)


# K21 (occd_gemm_f32x3_splitk, round 6): the skinny long-K project convolutions of the 1/16 and 1/32 stages with K cut over the
# grid -- (batch, Cout = M, pixels = N, Cin = K, skip?, plan or None = the library's)
SPLITK_CASES = [
    (2, 384, 468, 2304, True, None),            # 1/32 stage, the 13-fold launch
    (2, 640, 468, 3840, True, None),
    (2, 224, 1848, 1344, True, None),           # 1/16 stages
    (2, 160, 1848, 960, True, None),
    (2, 640, 468, 2304, False, None),           # first block of the last stage (no skip)
    (1, 200, 131, 1000, True, (7, 9, 2)),       # M, N, K tails (K = 62.5 steps of 16: the last chunk is short and partial), 2 row ranges
    (3, 40, 37, 256, False, (16, 1, 1)),        # one chunk: plain panel walk + the reduce launch; N % 4 = 1
    (2, 96, 70, 512, True, (2, 16, 3)),         # many 2-step chunks, N % 4 = 2, three row ranges for three row tiles
    (1, 384, 468, 2304, True, (26, 6, 1)),      # the largest chunk two workgroups per CU allow (416 k)
    (1, 64, 100, 2304, False, (53, 3, 1)),      # > 64 KB of LDS per workgroup (the big-LDS attribute path)
]


@pytest.mark.parametrize("case", SPLITK_CASES)
def test_gemm_x3_splitk_gate_skip_vs_float64(hip, case):
    batch, M, N, K, skip, plan = case
    g = torch.Generator().manual_seed(M + N + K)
    w = torch.randn(M, K, generator=g) / K ** 0.5
    y = torch.randn(batch, K, N, generator=g) * 2.0
    gate = torch.sigmoid(torch.randn(batch, K, generator=g))
    shift = torch.randn(M, generator=g)
    res = torch.randn(batch, M, N, generator=g) if skip else None
    gated = (y * gate.unsqueeze(-1))                      # float32 rounding of the product, as the reference's x * gate
    ref = torch.matmul(w.double(), gated.double()) + shift.double().view(-1, 1)
    if skip:
        ref = ref + res.double()
    pa = hip.GemmPacked(w.to(DEV), "a")
    yd, gd, sd, rd = y.to(DEV), gate.to(DEV), shift.to(DEV), (res.to(DEV) if skip else None)
    hip._SPLITK_WS.clear()
    with hip.profile() as prof:
        got = hip.gemm_x3_splitk(pa, yd, bias=sd, k_scale=gd, res=rd, plan=plan)
    assert any(k.startswith("gemm_f32x3_splitk") for k in prof.rows), list(prof.rows)
    # the workspace's pad columns / stale contents must not matter: poison it and run again -- bit-identical
    hip._SPLITK_WS[yd.device].fill_(float("nan"))
    again = hip.gemm_x3_splitk(pa, yd, bias=sd, k_scale=gd, res=rd, plan=plan)
    assert torch.equal(got, again)
    err = float((got.cpu().double() - ref).abs().max() / ref.abs().max())
    lib = torch.matmul(w.to(DEV), gated.to(DEV)) + sd.view(-1, 1) + (rd if skip else 0)
    err_lib = float((lib.cpu().double() - ref).abs().max() / ref.abs().max())
    used = plan or hip.gemm_x3_splitk_plan(M, N, K, batch)[:3]
    print(f"gemm_x3_splitk {case[:5]} plan {used}: {err:.2e} (library fp32 {err_lib:.2e})")
    assert got.shape == ref.shape and err < 2e-6, (case, err)
    # without gate / skip / bias, with an activation: the plain product
    plain = hip.gemm_x3_splitk(pa, yd, act="swish", plan=plan).cpu().double()
    r2 = torch.matmul(w.double(), y.double())
    r2 = r2 * torch.sigmoid(r2)
    assert float((plain - r2).abs().max() / r2.abs().max()) < 3e-6


def test_gemm_x3_splitk_plan_and_argument_checks(hip):
    for M, N, K, batch in ((384, 468, 2304, 2), (640, 468, 3840, 2), (224, 1848, 1344, 2), (160, 1848, 960, 2), (40, 37, 256, 3)):
        per, nz, rr, ws = hip.gemm_x3_splitk_plan(M, N, K, batch)
        k16 = (K + 15) // 16
        assert 1 <= per <= 52, (per, nz)                          # a chunk never exceeds the LDS (52 steps x 16 k x 192 B = 160 KB)
        assert (nz - 1) * per < k16 <= nz * per and rr >= 1 and ws == batch * nz * M * ((N + 31) // 32 * 32)
    w = torch.randn(64, 256, device=DEV)
    pa = hip.GemmPacked(w, "a")
    b = torch.randn(2, 256, 40, device=DEV)
    for bad in ((8, 3, 1), (4, 3, 1), (16, 1, 3), (900, 1, 1)):      # chunks past K / not covering K / more ranges than row tiles / LDS
        with pytest.raises(RuntimeError):
            hip.gemm_x3_splitk(pa, b, plan=bad)
    with pytest.raises(RuntimeError):
        hip.gemm_x3_splitk(w, b)                                   # float32 left operand: K16's business
    with pytest.raises(RuntimeError):
        hip.gemm_x3_splitk(pa, b, k_scale=torch.ones(2, 255, device=DEV))


@pytest.mark.parametrize("packed_b", [False, True])
@pytest.mark.parametrize("shape", [(4, 4096, 512, 256), (3, 70, 133, 104), (None, 40, 7, 8), (2, 300, 64, 1024)])
def test_gemm_x3_column_bias(hip, shape, packed_b):
    """occd_gemm_args.bias_n (ABI 13): one bias value per COLUMN, added ahead of the activation -- rows-times-weights products
    (rows = voxels, columns = output channels): the CRP's relation-logit convolutions as one batched launch with the left
    operand shared over the relations (first shape: config 2's R = 4, N = 4096 voxels, M = 512, C = 256)."""
    batch, M, N, K = shape
    g = torch.Generator().manual_seed(M + N + K)
    a = torch.randn(M, K, generator=g)                                   # shared over the batch
    b = torch.randn(*((K, N) if batch is None else (batch, K, N)), generator=g) / K ** 0.5
    bn = torch.randn(*((N,) if batch is None else (batch, N)), generator=g)
    ref = torch.matmul(a.double(), b.double()) + (bn.double() if batch is None else bn.double().unsqueeze(1))
    xb = hip.GemmPacked(b.to(DEV), "b") if packed_b else b.to(DEV)
    got = hip.gemm_x3(a.to(DEV), xb, bias_n=bn.to(DEV)).cpu().double()
    assert got.shape == ref.shape
    assert float((got - ref).abs().max() / ref.abs().max()) < 2e-6
    # with the row bias and an activation on top; a column bias shared over the batch
    rb = torch.randn(M, generator=g)
    shared = bn if batch is None else bn[0].contiguous()
    ref2 = F_leaky(torch.matmul(a.double(), b.double()) + shared.double() + rb.double().view(-1, 1))
    got2 = hip.gemm_x3(a.to(DEV), xb, bias=rb.to(DEV), bias_n=shared.to(DEV), act="leaky").cpu().double()
    assert float((got2 - ref2).abs().max() / ref2.abs().max()) < 2e-6
    with pytest.raises(RuntimeError):
        hip.gemm_x3(a.to(DEV), xb, bias_n=bn.to(DEV)[..., :-1].contiguous())
    if not packed_b:
        with pytest.raises(RuntimeError):                               # the wave-specialised kernel has no column bias
            hip.gemm_x3(a.to(DEV), xb, bias_n=bn.to(DEV), tile_hint=6)


def F_leaky(t, slope=0.01):
    return torch.where(t > 0, t, t * slope)
