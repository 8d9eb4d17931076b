"""K13 (csrc/bn.hip, occdepth_amd/bn.py): training-mode BatchNorm + activation + residual, forward and backward, against
torch.nn.BatchNorm in float64 on the CPU -- output, input / residual / weight / bias gradients, running statistics and
num_batches_tracked -- for both layouts (channels-last rows incl. ragged channel counts and bf16 storage, NCHW / NCDHW
planes), every fused activation and both residual positions."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda"


def reference(bn64, x, act, slope, res, res_first):
    y = bn64(x)
    if res is not None and res_first:
        y = y + res
    if act == "relu":
        y = F.relu(y)
    elif act == "leaky":
        y = F.leaky_relu(y, slope)
    elif act == "swish":
        y = y * torch.sigmoid(y)
    if res is not None and not res_first:
        y = y + res
    return y


CASES = [
    # (shape, channels_last, dtype, act, residual: None / "post" / "first", mean, std)
    ((2, 32, 6, 10, 32), True, torch.float32, "relu", None, 0.3, 1.0),
    ((1, 32, 5, 12, 32), True, torch.float32, None, "post", -0.5, 2.0),          # head: y += bn2(conv2(.))
    ((2, 64, 6, 8, 16), True, torch.float32, "relu", "first", 0.0, 1.0),         # bottleneck: relu(bn5(.) + skip)
    ((2, 20, 5, 6, 10), True, torch.float32, "relu", None, 1.0, 0.5),            # ragged channel count: rows of 24
    ((1, 34, 4, 6, 8), True, torch.float32, None, None, 0.0, 1.0),
    ((1, 576, 3, 4, 4), True, torch.float32, "relu", None, 0.0, 1.0),            # more than 256 channel quads... (C4 = 144)
    ((1, 2304, 2, 3, 4), True, torch.float32, None, None, 0.0, 1.0),             # C4 = 576 > 256: quad loop
    ((2, 80, 17, 33), True, torch.float32, "leaky", None, 0.2, 1.5),             # NHWC decoder level
    ((2, 80, 17, 33), True, torch.bfloat16, "leaky", None, 0.2, 1.5),
    ((2, 32, 6, 10, 32), True, torch.bfloat16, "relu", "first", 0.0, 1.0),
    ((2, 48, 19, 23), False, torch.float32, "swish", None, 0.1, 1.0),            # NCHW encoder: BN + swish
    ((2, 24, 11, 13), False, torch.float32, None, "post", 0.0, 1.0),             # MBConv bn3 + skip
    ((3, 16, 7, 9), False, torch.float32, "relu", "first", 0.0, 1.0),
    ((1, 640, 6, 20), False, torch.float32, "swish", None, 0.0, 1.0),            # small NCHW layers: ONE launch per direction
    ((2, 96, 40, 50), False, torch.float32, "swish", "post", 0.3, 1.2),
    ((1, 128, 47, 153), False, torch.float32, "relu", "first", 0.0, 1.0),
    ((2, 288, 31, 37), False, torch.float32, None, None, 20.0, 2.0),
    ((2, 16, 5, 6, 7), False, torch.float32, "relu", None, 0.0, 1.0),            # NCDHW planes
    ((2, 16, 9, 11), False, torch.float32, None, None, 50.0, 3.0),               # mean far from zero
    ((2, 16, 4, 9, 8), True, torch.float32, None, None, -300.0, 0.5),
]


@pytest.mark.parametrize("case", CASES, ids=[f"{i}" for i in range(len(CASES))])
def test_bn_act_matches_float64_reference(case):
    from occdepth_amd import bn as obn
    shape, cl, dtype, act, resmode, mean, std = case
    torch.manual_seed(len(shape) * 100 + shape[1])
    C = shape[1]
    Norm = torch.nn.BatchNorm3d if len(shape) == 5 else torch.nn.BatchNorm2d
    x = torch.randn(*shape) * std + mean
    g = torch.randn(*shape)
    res = torch.randn(*shape) if resmode else None
    ref = Norm(C, momentum=0.1).double().train()
    mod = Norm(C, momentum=0.1).to(DEV).train()
    with torch.no_grad():
        for b in (ref, mod):
            b.weight.copy_(torch.linspace(0.5, 1.5, C))
            b.bias.copy_(torch.linspace(-1, 1, C))
            b.running_mean.fill_(0.25)
            b.running_var.fill_(2.0)
    mf = torch.channels_last_3d if len(shape) == 5 else torch.channels_last

    def dev(t):
        t = t.to(DEV, dtype)
        if not cl:
            return t.contiguous()
        # channels-last: dense rows when C is a multiple of 4, else rows padded to ceil8(C) with zero pads -- the layout
        # the HIP convolutions hand over for ragged channel counts (an ATen channels_last tensor with C = 34 has rows of
        # 34 floats: not addressable in 16-byte quads, bn_act falls back to the backend for those)
        return t.contiguous(memory_format=mf) if C % 4 == 0 else obn._to_rows(t)

    xs = dev(x).detach().requires_grad_(True)
    rs = dev(res).detach().requires_grad_(True) if res is not None else None
    xr = xs.detach().cpu().double().requires_grad_(True)              # (the bf16-rounded values when dtype is bf16)
    rr = rs.detach().cpu().double().requires_grad_(True) if res is not None else None
    assert obn._Geom.supported(xs)
    ys = obn.bn_act(mod, xs, act=act, slope=0.02, res=rs, res_first=(resmode == "first"))
    assert ys.grad_fn is not None and type(ys.grad_fn).__name__.startswith("_BNActFn")
    yr = reference(ref, xr, act, 0.02, rr, resmode == "first")
    gs = dev(g)
    ys.backward(gs)
    yr.backward(gs.detach().cpu().double())
    bf = dtype == torch.bfloat16
    tol = 2e-2 if bf else 3e-5 * max(1.0, abs(mean) / std)

    def rel(a, b):
        return float((a.detach().cpu().double() - b.detach()).abs().max() / b.detach().abs().max().clamp_min(1e-30))

    assert rel(ys, yr) < tol, ("y", rel(ys, yr))
    assert rel(xs.grad, xr.grad) < (tol if bf else 2e-4 * max(1.0, abs(mean) / std)), ("gx", rel(xs.grad, xr.grad))
    if res is not None:
        assert rel(rs.grad, rr.grad) < tol, ("gres", rel(rs.grad, rr.grad))
    ptol = 2e-3 if bf else 1e-4 * max(1.0, abs(mean) / std)
    assert rel(mod.weight.grad, ref.weight.grad) < ptol, ("gw", rel(mod.weight.grad, ref.weight.grad))
    assert rel(mod.bias.grad, ref.bias.grad) < ptol, ("gb", rel(mod.bias.grad, ref.bias.grad))
    assert rel(mod.running_var, ref.running_var) < 1e-4 * max(1.0, abs(mean) / std)
    assert float((mod.running_mean.cpu().double() - ref.running_mean).abs().max()) < 1e-5 * max(1.0, abs(mean))
    assert int(mod.num_batches_tracked) == 1
    # channels-last outputs keep the layout (zero-copy into the next HIP convolution) and zero channel pads
    if cl:
        assert obn._rows_geometry(ys) is not None
        if C % 8:
            cs = obn._rows_geometry(ys)[1]
            flat = torch.as_strided(ys, (ys.numel() // C, cs), (cs, 1))
            assert float(flat[:, C:].float().abs().max()) == 0.0


def test_bn_act_is_plain_torch_in_eval_mode_and_when_disabled():
    from occdepth_amd import bn as obn
    torch.manual_seed(0)
    mod = torch.nn.BatchNorm2d(8).to(DEV)
    x = torch.randn(2, 8, 5, 6, device=DEV)
    mod.eval()
    y = obn.bn_act(mod, x, act="relu")
    assert torch.equal(y, F.relu(mod(x)))
    mod.train()
    obn.ENABLED = False
    try:
        y = obn.bn_act(mod, x.requires_grad_(True), act="relu")
        assert "BNAct" not in type(y.grad_fn).__name__
    finally:
        obn.ENABLED = True
