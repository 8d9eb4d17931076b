"""-m gpu: the statistics / gradient / confusion kernels of csrc/loss.hip through the C-ABI, against the reference
goldens (tests/golden/losses.npz), the oracle sums, and size-independent properties at BASELINE config-2 size."""
import os

import numpy as np
import pytest
import torch

import emu
import golden_cases as gc
from oracle import losses as L
from test_losses import GOLD, CASES, check_against_golden, run_product_step

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", CASES)
def test_step_losses_and_gradients_match_reference(name):
    out, total, grads = run_product_step(name, "cuda", use_emu=False)
    check_against_golden(name, out, total, grads)


@pytest.mark.parametrize("name", CASES)
def test_statistics_match_oracle_sums_and_are_deterministic(name):
    from occdepth_amd import hip
    d = gc.loss_case(name)
    C, F = d["n_classes"], d["frustums_masks"].shape[1]
    args = (d["ssc_logit"].cuda(), d["target"].to(torch.uint8).cuda(), d["frustums_masks"].view(torch.uint8).cuda(),
            d["class_weights"].cuda())
    raw = hip.ssc_loss_stats(*args)
    for _ in range(3):
        assert torch.equal(hip.ssc_loss_stats(*args), raw)            # integer accumulation: order-independent
    want = emu.ssc_loss_stats(d["ssc_logit"], d["target"].to(torch.uint8), d["frustums_masks"].view(torch.uint8),
                              d["class_weights"])
    sc = hip.ssc_stats_scale(C, F, "cpu")
    got_r, want_r = raw.cpu().double() * sc, want.double() * sc
    assert torch.equal(raw.cpu()[2 * C:3 * C + 1], want[2 * C:3 * C + 1])       # counts: exact
    assert torch.allclose(got_r, want_r, rtol=2e-6, atol=1e-6)
    # occupancy relabelling inside the kernel
    raw_occ = hip.ssc_loss_stats(d["occ_logit"].cuda(), args[1], None, d["class_weights_occ"].cuda(), map_occ=True)
    want_occ = emu.ssc_loss_stats(d["occ_logit"], d["target"].to(torch.uint8), None, d["class_weights_occ"], map_occ=True)
    assert torch.equal(raw_occ.cpu()[4:7], want_occ[4:7])
    assert torch.allclose(raw_occ.cpu().double() * hip.ssc_stats_scale(2, 0, "cpu"),
                          want_occ.double() * hip.ssc_stats_scale(2, 0, "cpu"), rtol=2e-6, atol=1e-6)


@pytest.mark.parametrize("name", CASES)
def test_gradient_kernel_matches_emulation(name):
    from occdepth_amd import hip
    d = gc.loss_case(name)
    C, F = d["n_classes"], d["frustums_masks"].shape[1]
    g = torch.randn(3 * C + 3 + F * C, generator=torch.Generator().manual_seed(5))
    t8, m8 = d["target"].to(torch.uint8), d["frustums_masks"].view(torch.uint8)
    got = hip.ssc_loss_grad(d["ssc_logit"].cuda(), t8.cuda(), m8.cuda(), d["class_weights"].cuda(), g.cuda())
    want = emu.ssc_loss_grad(d["ssc_logit"], t8, m8, d["class_weights"], g)
    assert (got.cpu() - want).abs().max() <= 2e-6 * want.abs().max()


@pytest.mark.parametrize("name", CASES)
def test_confusion_and_metrics_match_reference(name):
    from occdepth_amd.loss.sscMetrics import SSCMetrics
    d = gc.loss_case(name)
    m = SSCMetrics(d["n_classes"])
    m.add_batch_logits(d["ssc_logit"].cuda(), d["target"].cuda())
    m2 = SSCMetrics(d["n_classes"])
    m2.add_batch(d["ssc_logit"].argmax(1).numpy(), d["target"].numpy())
    assert torch.equal(m.hist, m2.hist)
    assert np.array_equal(m.hist.cpu().numpy(), L.confusion(d["ssc_logit"].argmax(1).numpy(), d["target"].numpy(),
                                                            d["n_classes"]))
    for k in ("tps", "fps", "fns"):
        assert np.array_equal(getattr(m, k), GOLD[f"{name}.metric.{k}"]), k
    st = m.get_stats()
    for k in ("precision", "recall", "iou", "iou_ssc_mean"):
        assert st[k] == pytest.approx(float(GOLD[f"{name}.metric.{k}"]), rel=1e-12)
    m.add_batch_logits(d["ssc_logit"].cuda(), d["target"].cuda())               # accumulates
    assert torch.equal(m.hist, 2 * m2.hist)


def test_full_size_properties_config2():
    """(1, 20, 256, 256, 32) logits + 64 frustum masks: conservation laws of the sums, zero-sum gradients."""
    from occdepth_amd import hip
    from occdepth_amd.loss import ssc_loss
    g = torch.Generator(device="cuda").manual_seed(11)
    B, C, dims, F = 1, 20, (256, 256, 32), 64
    logits = (torch.randn(B, C, *dims, device="cuda", generator=g) * 3).requires_grad_(True)
    target = torch.randint(0, C, (B, *dims), device="cuda", generator=g).to(torch.uint8)
    target[torch.rand(B, *dims, device="cuda", generator=g) < 0.2] = 255
    fid = torch.randint(0, F + 8, (B, *dims), device="cuda", generator=g)
    masks = torch.stack([fid == f for f in range(F)], 1)
    w = torch.rand(C, device="cuda", generator=g) + 0.5
    raw = hip.ssc_loss_stats(logits.detach(), target, masks.view(torch.uint8), w)
    assert torch.equal(raw, hip.ssc_loss_stats(logits.detach(), target, masks.view(torch.uint8), w))
    st = (raw.double() * hip.ssc_stats_scale(C, F, "cuda")).cpu()
    M = int((target != 255).sum())
    assert int(raw[3 * C]) == M and int(raw[2 * C:3 * C].sum()) == M
    assert float(st[:C].sum()) == pytest.approx(M, rel=1e-6)                       # sum_c P_c = #labelled
    per_f = st[3 * C + 3:].view(F, C).sum(1)
    counts = masks.view(B, F, -1).sum((0, 2)).cpu().double()
    assert torch.allclose(per_f, counts, rtol=1e-6)                                # sum_c F[f][c] = |frustum f|
    assert bool((st[C:2 * C] <= st[:C] + 1e-9).all())                              # N_c <= P_c
    dists = torch.rand(B, F, C, device="cuda", generator=g)
    out = ssc_loss.ssc_losses(logits, target, w, masks, dists)
    sum(out.values()).backward()
    gsum = logits.grad.sum(1).abs().max()
    assert float(gsum) <= 1e-5 * float(logits.grad.abs().max()) + 1e-12            # softmax Jacobian: zero row sums (fp32: C eps)
    assert torch.isfinite(logits.grad).all()
    # value check of one loss against plain torch at full size
    ce = torch.nn.functional.cross_entropy(logits.detach(), target.long(), weight=w, ignore_index=255)
    assert float(out["loss_ssc"]) == pytest.approx(float(ce), rel=1e-5)
