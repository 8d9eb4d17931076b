"""-m gpu: geffnet's SqueezeExcite in training on the HIP passes of csrc/se2d.hip (hip._SqueezeExciteFn: plane sums ->
occd_se_gate -> scale; backward: plane dots -> occd_se_bwd -> scale + mean gradient) against the module's own autograd graph
in float64 on the CPU -- output, input gradient and all four parameter gradients -- on the channel / squeeze / plane sizes of
the B7 encoder stages; determinism (fixed summation order)."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(scope="module")
def hip():
    from occdepth_amd import hip as h
    h.load()
    return h


# (images, channels, squeezed channels, H, W): stage shapes of tf_efficientnet_b7_ns at 370 x 1220 (planes cut where large)
CASES = [(2, 32, 8, 185, 610), (2, 288, 12, 93, 305), (2, 480, 20, 47, 153), (2, 1344, 56, 24, 77), (2, 2304, 96, 12, 39),
         (2, 3840, 160, 12, 39), (1, 40, 10, 7, 9), (5, 72, 3, 33, 17)]


@pytest.mark.parametrize("case", CASES)
def test_squeeze_excite_training_function(hip, case):
    from occdepth_amd.models.efficientnet import SqueezeExcite
    B, C, Cr, H, W = case
    torch.manual_seed(C + Cr)
    se = SqueezeExcite(C, Cr)
    with torch.no_grad():
        for p in se.parameters():
            p.mul_(3.0)                                           # gates away from 0.5
    x = torch.randn(B, C, H, W)
    gout = torch.randn(B, C, H, W)
    # float64 reference: the module's own graph (kernels off: CPU tensors never take them)
    se64 = SqueezeExcite(C, Cr).double()
    se64.load_state_dict({k: v.double() for k, v in se.state_dict().items()})
    x64 = x.double().requires_grad_(True)
    out64 = se64(x64)
    out64.backward(gout.double())
    se = se.to(DEV).train()
    xg = x.to(DEV).requires_grad_(True)
    assert hip.squeeze_excite_autograd_ok(se, xg)
    with hip.profile() as prof:
        out = se(xg)
        out.backward(gout.to(DEV))
    tags = {k.split(":")[0] for k in prof.rows}
    assert {"plane_sum", "plane_dot", "se_gate", "se_bwd"} <= tags, tags

    def rel(a, b):
        return float((a.detach().cpu().double() - b).abs().max() / b.abs().max())
    assert rel(out, out64.detach()) < 2e-6
    assert rel(xg.grad, x64.grad) < 2e-5
    for (n, p), (_, p64) in zip(se.named_parameters(), se64.named_parameters()):
        assert p.grad is not None and p.grad.shape == p.shape, n
        assert rel(p.grad, p64.grad) < 5e-5, (n, rel(p.grad, p64.grad))
    # fixed summation order: a second run is bit-identical
    g1 = [p.grad.clone() for p in se.parameters()] + [xg.grad.clone()]
    for p in se.parameters():
        p.grad = None
    xg.grad = None
    se(xg).backward(gout.to(DEV))
    for a, b in zip(g1, [p.grad for p in se.parameters()] + [xg.grad]):
        assert torch.equal(a, b)


def test_squeeze_excite_kernels_can_be_switched_off(hip):
    from occdepth_amd.models.efficientnet import SqueezeExcite
    se = SqueezeExcite(16, 4).to(DEV).train()
    x = torch.randn(2, 16, 5, 6, device=DEV, requires_grad=True)
    saved = hip.SE_TRAIN
    hip.SE_TRAIN = False
    try:
        with hip.profile() as prof:
            se(x).sum().backward()
        assert not any(k.startswith(("plane_", "se_bwd")) for k in prof.rows)
    finally:
        hip.SE_TRAIN = saved
