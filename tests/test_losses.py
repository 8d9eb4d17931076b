"""Training-step losses (N1) and SSC metrics (N4): oracle vs the reference goldens, and the product's host logic
(formulas on the statistics vector, autograd plumbing) through the TEST-ONLY emulation of the statistics kernels.
The `-m gpu` counterparts that run the real HIP kernels are in tests/test_losses_gpu.py."""
import os

import numpy as np
import pytest
import torch

import emu
import golden_cases as gc
from oracle import losses as L

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "losses.npz"))
CASES = list(gc.LOSS_CASES)


def _np(d, k):
    v = d[k]
    return [x.numpy() for x in v] if isinstance(v, list) else v.numpy()


@pytest.mark.parametrize("name", CASES)
def test_oracle_losses_match_reference(name):
    d = gc.loss_case(name)
    ssc, tgt = _np(d, "ssc_logit"), _np(d, "target")
    # the reference computes in float32: its own rounding is ~1e-6 relative
    assert L.ce_ssc_loss(ssc, tgt, _np(d, "class_weights")) == pytest.approx(float(GOLD[f"{name}.ce"]), rel=2e-6)
    assert L.sem_scal_loss(ssc, tgt) == pytest.approx(float(GOLD[f"{name}.sem_scal"]), rel=2e-6)
    assert L.geo_scal_loss(ssc, tgt) == pytest.approx(float(GOLD[f"{name}.geo_scal"]), rel=2e-6)
    assert L.relation_loss(_np(d, "P_logits"), _np(d, "CP_mega_matrices")) == \
        pytest.approx(float(GOLD[f"{name}.relation"]), rel=2e-6)
    assert L.depth_loss(_np(d, "gt_depth"), _np(d, "depth_pred"), d["depth_factor"], d["d_bound"]) == \
        pytest.approx(float(GOLD[f"{name}.depth"]), rel=2e-6)
    fr = L.frustum_proportion_loss(ssc, _np(d, "frustums_masks"), _np(d, "frustums_class_dists"))
    assert fr == pytest.approx(float(GOLD[f"{name}.step.train/loss_frustums"]), rel=2e-6)


@pytest.mark.parametrize("name", CASES)
def test_oracle_depth_onehot_and_metrics_match_reference(name):
    d = gc.loss_case(name)
    f = d["depth_factor"]
    h, w = d["depth_pred"].shape[3:]
    lab = _np(d, "gt_depth")[:, :, :h * f, :w * f]
    onehot = L.downsampled_gt_depth(lab.reshape(-1, h * f, w * f), f, d["d_bound"])
    assert np.array_equal(onehot, GOLD[f"{name}.depth_onehot"])
    pred = _np(d, "ssc_logit").argmax(1)
    m = L.metrics_from_confusion(L.confusion(pred, _np(d, "target"), d["n_classes"]))
    for k in ("tps", "fps", "fns", "completion"):
        assert np.array_equal(m[k], GOLD[f"{name}.metric.{k}"]), k
    for k in ("precision", "recall", "iou", "iou_ssc_mean"):
        assert m[k] == pytest.approx(float(GOLD[f"{name}.metric.{k}"]), rel=1e-12)
    assert np.allclose(m["iou_ssc"], GOLD[f"{name}.metric.iou_ssc"], rtol=1e-12)


def run_product_step(name, device, use_emu):
    """The product's loss functions on one case -> (dict of losses, dict of grads)."""
    import contextlib
    from occdepth_amd.loss import CRP_loss, depth_loss, ssc_loss
    d = gc.loss_case(name)
    dev = torch.device(device)
    leaves = {k: d[k].clone().to(dev).requires_grad_(True) for k in ("ssc_logit", "occ_logit", "P_logits", "depth_pred")}
    kitti = name == "kitti_like"
    with (emu.patched() if use_emu else contextlib.nullcontext()):
        out = ssc_loss.ssc_losses(leaves["ssc_logit"], d["target"].to(dev), d["class_weights"].to(dev),
                                  [m.to(dev) for m in d["frustums_masks"]],
                                  [x.to(dev) for x in d["frustums_class_dists"]])
        if kitti:
            out["loss_occ"] = ssc_loss.occ_ce_loss(leaves["occ_logit"], d["target"].to(dev),
                                                   d["class_weights_occ"].to(dev))
            out["loss_relation_ce_super"] = CRP_loss.compute_super_CP_multilabel_loss(
                leaves["P_logits"], [c.to(dev) for c in d["CP_mega_matrices"]])
            dl = depth_loss.DepthClsLoss(d["depth_factor"], d["d_bound"])
            out["loss_depth"] = dl.get_depth_loss(d["gt_depth"].to(dev), leaves["depth_pred"][:, 0].unsqueeze(1))
        total = sum(out.values())
        total.backward()
    grads = {k: v.grad for k, v in leaves.items() if v.grad is not None}
    return out, total, grads


def check_against_golden(name, out, total, grads, rel=2e-5):
    for k, v in out.items():
        assert float(v.detach()) == pytest.approx(float(GOLD[f"{name}.step.train/{k}"]), rel=rel), k
    assert float(total.detach()) == pytest.approx(float(GOLD[f"{name}.step.total"]), rel=rel)
    for k, g in grads.items():
        ref = GOLD[f"{name}.grad.{k}"]
        err = np.abs(g.detach().cpu().numpy() - ref).max() / np.abs(ref).max()
        assert err < 2e-5, (k, err)


@pytest.mark.parametrize("name", CASES)
def test_host_formulas_and_autograd_match_reference_step(name):
    out, total, grads = run_product_step(name, "cpu", use_emu=True)
    assert set(grads) == ({"ssc_logit", "occ_logit", "P_logits", "depth_pred"} if name == "kitti_like" else {"ssc_logit"})
    check_against_golden(name, out, total, grads)


@pytest.mark.parametrize("name", CASES)
def test_individual_loss_functions_keep_the_reference_signature(name):
    from occdepth_amd.loss import ssc_loss
    d = gc.loss_case(name)
    with emu.patched():
        ce = ssc_loss.CE_ssc_loss(d["ssc_logit"], d["target"], d["class_weights"])
        sem = ssc_loss.sem_scal_loss(d["ssc_logit"], d["target"])
        geo = ssc_loss.geo_scal_loss(d["ssc_logit"], d["target"])
        fr = ssc_loss.frustum_proportion_loss(d["ssc_logit"], d["frustums_masks"], d["frustums_class_dists"])
    assert float(ce) == pytest.approx(float(GOLD[f"{name}.ce"]), rel=2e-5)
    assert float(sem) == pytest.approx(float(GOLD[f"{name}.sem_scal"]), rel=2e-5)
    assert float(geo) == pytest.approx(float(GOLD[f"{name}.geo_scal"]), rel=2e-5)
    assert float(fr) == pytest.approx(float(GOLD[f"{name}.step.train/loss_frustums"]), rel=2e-5)
    p = torch.tensor([0.2, 0.3, 0.5])
    t = torch.tensor([0.0, 0.6, 0.4])
    assert float(ssc_loss.KL_sep(p, t)) == pytest.approx(0.6 * np.log(0.6 / 0.3) + 0.4 * np.log(0.4 / 0.5), rel=1e-6)


@pytest.mark.parametrize("name", CASES)
def test_metrics_host_logic_matches_reference(name):
    from occdepth_amd.loss.sscMetrics import SSCMetrics
    d = gc.loss_case(name)
    with emu.patched():
        m = SSCMetrics(d["n_classes"], device="cpu")
        m.add_batch(d["ssc_logit"].argmax(1).numpy(), d["target"].numpy())
        m2 = SSCMetrics(d["n_classes"], device="cpu")
        m2.add_batch_logits(d["ssc_logit"], d["target"])
    assert torch.equal(m.hist, m2.hist)
    for k in ("tps", "fps", "fns"):
        assert np.array_equal(getattr(m, k), GOLD[f"{name}.metric.{k}"]), k
    st = m.get_stats()
    for k in ("precision", "recall", "iou", "iou_ssc_mean"):
        assert st[k] == pytest.approx(float(GOLD[f"{name}.metric.{k}"]), rel=1e-12)
    assert np.allclose(st["iou_ssc"], GOLD[f"{name}.metric.iou_ssc"], rtol=1e-12)


def test_emulated_statistics_match_the_oracle_sums():
    """The emulation (and therefore the layout the kernels must produce) against independent numpy sums."""
    d = gc.loss_case("kitti_like")
    C = d["n_classes"]
    from occdepth_amd import hip
    raw = emu.ssc_loss_stats(d["ssc_logit"], d["target"].to(torch.uint8), d["frustums_masks"].view(torch.uint8),
                             d["class_weights"])
    st = raw.double() * hip.ssc_stats_scale(C, d["frustums_masks"].shape[1], "cpu")
    p = L.softmax(d["ssc_logit"].numpy(), 1)
    t = d["target"].numpy()
    lab = t != 255
    for c in (0, 3, C - 1):
        assert float(st[c]) == pytest.approx(p[:, c][lab].sum(), rel=1e-9)
        assert float(st[C + c]) == pytest.approx(p[:, c][t == c].sum(), rel=1e-9, abs=1e-9)
        assert int(raw[2 * C + c]) == int((t == c).sum())
    assert int(raw[3 * C]) == int(lab.sum())
    f = 5
    m = d["frustums_masks"].numpy()[:, f]
    assert float(st[3 * C + 3 + f * C + 2]) == pytest.approx(p[:, 2][m].sum(), rel=1e-9)
