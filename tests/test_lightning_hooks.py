"""The pytorch-lightning surface of occdepth.models.OccDepth (SURVEY 8(b) "must keep"): a minimal fake Trainer drives the
hooks in PL 1.4 order -- training_step / validation_step per batch, validation_epoch_end, then test_step /
test_epoch_end -- and checks what scripts/train.py and scripts/eval.py rely on:
  * `ModelCheckpoint(monitor="val/mIoU")` finds its key (scripts/train.py:152-168),
  * per-class `{prefix}_SemIoU/{class}`, `/mIoU`, `/IoU`, `/Precision`, `/Recall` are logged for train and val and the
    metrics are reset afterwards (reference models/OccDepth.py:542-557),
  * `trainer.test` prints the evaluation report in the reference's format (:562-580, scripts/eval.py:65-80).
CPU: the loss / confusion kernels run through the test-only emulation of the C ABI."""
import re

import numpy as np
import pytest
import torch

import emu
import golden_cases as gc
from test_train_step import CFGS  # noqa: F401
from test_oracle_vs_golden import build_product


class FakeTrainer:
    """Calls the LightningModule hooks in the order pytorch-lightning 1.4.9 does for fit + test."""

    def __init__(self, model):
        self.model = model
        self.logged = {}
        model.log = lambda key, value, **kw: self.logged.__setitem__(key, float(value))

    def fit(self, train_batches, val_batches):
        opt = self.model.configure_optimizers()
        for i, b in enumerate(train_batches):
            loss = self.model.training_step(b, i)
            assert torch.is_tensor(loss) and loss.requires_grad
        for i, b in enumerate(val_batches):
            with torch.no_grad():
                self.model.validation_step(b, i)
        self.model.validation_epoch_end([])
        return opt

    def test(self, batches):
        for i, b in enumerate(batches):
            with torch.no_grad():
                self.model.test_step(b, i)
        self.model.test_epoch_end([])


def _batch(cfg_name, m, cfg):
    batch = gc.occdepth_batch(cfg_name)
    with torch.no_grad(), emu.patched():
        out = m(batch)
    shapes = {k: tuple(v.shape) for k, v in out.items() if torch.is_tensor(v)}
    return dict(batch, **gc.train_extras(cfg_name, shapes, tuple(cfg.full_scene_size), cfg.n_classes,
                                         batch["img"].shape[-2:]))


def test_hooks_log_what_the_reference_scripts_monitor(capsys):
    cfg_name = "nyu_small"
    m, cfg, _ = build_product(cfg_name)
    m.eval()                                                   # running-statistics BatchNorm: deterministic, fast
    m.class_names = [f"cls{i}" for i in range(cfg.n_classes)]
    for attr in ("train_metrics", "val_metrics", "test_metrics"):
        assert getattr(m, attr).n_classes == cfg.n_classes     # reference :131-133
    b = _batch(cfg_name, m, cfg)
    tr = FakeTrainer(m)
    with emu.patched():
        opts, scheds = tr.fit([b], [b, b])
        want_val = m.val_metrics.get_stats()                   # (already reset: all zero)
    assert isinstance(opts[0], torch.optim.AdamW) and scheds[0].milestones == {18: 1, 24: 1} and scheds[0].gamma == 0.4
    assert m.cur_batch == 1
    for prefix in ("train", "val"):
        for key in ("mIoU", "IoU", "Precision", "Recall"):
            assert f"{prefix}/{key}" in tr.logged
        for c in m.class_names:
            assert f"{prefix}_SemIoU/{c}" in tr.logged
        assert f"{prefix}/loss" in tr.logged and f"{prefix}/loss_ssc" in tr.logged
    assert 0.0 <= tr.logged["val/mIoU"] <= 1.0
    # the logged numbers are the statistics of the accumulated counts (two identical val batches == one, as ratios)
    from occdepth_amd.loss.sscMetrics import SSCMetrics
    ref = SSCMetrics(cfg.n_classes, device="cpu")
    with torch.no_grad(), emu.patched():
        ref.add_batch_logits(m(b)["ssc_logit"], b["target"])
        st = ref.get_stats()
    import pytest                                              # (the 1e-5 in tp / (tp + fp + fn + 1e-5) is not scale-free)
    assert tr.logged["val/mIoU"] == pytest.approx(float(st["iou_ssc_mean"]), rel=1e-6)
    assert tr.logged["val/IoU"] == float(st["iou"])
    assert tr.logged["train/mIoU"] == float(st["iou_ssc_mean"])
    assert np.allclose([tr.logged[f"val_SemIoU/{c}"] for c in m.class_names], st["iou_ssc"], rtol=1e-6)
    assert int(m.val_metrics.hist.sum()) == 0 and int(m.train_metrics.hist.sum()) == 0 and want_val["iou"] == 0   # reset() happened (in place)

    capsys.readouterr()
    with emu.patched():
        tr.test([b])
    text = capsys.readouterr().out
    lines = text.strip().splitlines()
    assert lines[0] == "test======"
    assert re.fullmatch(r"Precision=\d+\.\d{4}, Recall=\d+\.\d{4}, IoU=\d+\.\d{4}", lines[1])
    assert lines[2] == "class IoU: {}, ".format(m.class_names)
    assert len(lines[3].split(",  ")) == cfg.n_classes and lines[3].endswith(", ")
    assert lines[4] == "mIoU={:.4f}".format(st["iou_ssc_mean"] * 100)
    assert lines[1] == "Precision={:.4f}, Recall={:.4f}, IoU={:.4f}".format(st["precision"] * 100, st["recall"] * 100,
                                                                             st["iou"] * 100)
    assert int(m.test_metrics.hist.sum()) == 0                # reset() zeroes the matrix in place
    assert "test/loss_frustums" not in tr.logged               # reference: no frustum loss in test


def test_configure_optimizers_by_dataset():
    m, cfg, _ = build_product("kitti_small")
    opts, scheds = m.configure_optimizers()
    assert scheds[0].milestones == {18: 1, 24: 1} and scheds[0].gamma == 0.4
    assert opts[0].defaults["lr"] == cfg.lr and opts[0].defaults["weight_decay"] == cfg.weight_decay
    m.dataset = "tartanair"
    assert m.configure_optimizers()[1][0].milestones == {20: 1}
    m.dataset = "other"
    try:
        m.configure_optimizers()
        raise AssertionError("unknown dataset must raise")
    except NotImplementedError:
        pass


def test_fast_eval_switches_are_off_by_default_and_read_from_the_environment(monkeypatch):
    """the benched eval configuration is reachable without a code edit -- OCCDEPTH_FAST_EVAL=1 (or the
    three individual variables) in the environment of an unmodified scripts/eval.py run, or `model.enable_fast_eval()`."""
    for k in ("OCCDEPTH_FAST_EVAL", "OCCDEPTH_BATCH_VIEWS", "OCCDEPTH_GRAPH_2D", "OCCDEPTH_GRAPH_ALL", "OCCDEPTH_CLONE_OUTPUTS"):
        monkeypatch.delenv(k, raising=False)
    m, _, _ = build_product("kitti_small")
    assert (m.batch_views, m.graph_2d, m.graph_all, m.clone_graph_outputs) == (False, False, False, True)
    monkeypatch.setenv("OCCDEPTH_FAST_EVAL", "1")
    m, _, _ = build_product("kitti_small")
    assert (m.batch_views, m.graph_2d, m.graph_all, m.clone_graph_outputs) == (True, True, True, True)
    monkeypatch.delenv("OCCDEPTH_FAST_EVAL")
    monkeypatch.setenv("OCCDEPTH_BATCH_VIEWS", "1")
    monkeypatch.setenv("OCCDEPTH_GRAPH_ALL", "1")
    monkeypatch.setenv("OCCDEPTH_CLONE_OUTPUTS", "0")
    m, _, _ = build_product("kitti_small")
    assert (m.batch_views, m.graph_2d, m.graph_all, m.clone_graph_outputs) == (True, False, True, False)
    monkeypatch.delenv("OCCDEPTH_BATCH_VIEWS")                     # the graphs need the batched views
    m, _, _ = build_product("kitti_small")
    assert (m.batch_views, m.graph_2d, m.graph_all) == (False, False, False)
    assert m.enable_fast_eval(clone_outputs=False) is m
    assert (m.batch_views, m.graph_2d, m.graph_all, m.clone_graph_outputs) == (True, True, True, False)


@pytest.mark.gpu
def test_eval_script_flow_reaches_the_fast_path_gpu(monkeypatch, capsys):
    """scripts/eval.py:65-80 (`trainer.test`) and scripts/generate_output.py:87-95 (a loop that KEEPS `pred`) against a model
    built with OCCDEPTH_FAST_EVAL=1 in the environment: the whole forward replays from one hipGraph, the printed report
    equals the plain model's, and predictions kept across forwards stay intact (fresh tensors by default)."""
    dev = "cuda"
    cfg_name = "kitti_small"

    def to_dev(b):
        return {k: ([t.to(dev) for t in v] if isinstance(v, list) and torch.is_tensor(v[0]) else
                    (v.to(dev) if torch.is_tensor(v) else v)) for k, v in b.items()}

    monkeypatch.delenv("OCCDEPTH_FAST_EVAL", raising=False)
    plain, cfg, _ = build_product(cfg_name)
    monkeypatch.setenv("OCCDEPTH_FAST_EVAL", "1")
    fast, _, _ = build_product(cfg_name)
    plain, fast = plain.to(dev).eval(), fast.to(dev).eval()
    assert fast.graph_all and not plain.graph_all
    base = gc.occdepth_batch(cfg_name)
    with torch.no_grad():
        shapes = {k: tuple(v.shape) for k, v in plain(to_dev(base)).items() if torch.is_tensor(v)}
    extras = gc.train_extras(cfg_name, shapes, tuple(cfg.full_scene_size), cfg.n_classes, base["img"].shape[-2:])
    g = torch.Generator().manual_seed(3)
    frames = [to_dev(dict(base, **extras, img=base["img"] + 0.3 * i * torch.randn(base["img"].shape, generator=g)))
              for i in range(3)]
    reports = {}
    for name, m in (("plain", plain), ("fast", fast)):
        m.class_names = [f"cls{i}" for i in range(cfg.n_classes)]
        capsys.readouterr()
        FakeTrainer(m).test(frames)
        reports[name] = capsys.readouterr().out.strip().splitlines()
    assert fast.graph_all, getattr(fast, "graph_all_error", None)
    assert [k[0] for k in fast._graphs] == ["all"]                 # one capture served the three frames
    assert reports["fast"][0] == "test======" and len(reports["fast"]) == len(reports["plain"]) == 5

    def numbers(line):
        return [float(x) for x in re.findall(r"\d+\.\d+", line)]
    for a, b in zip(reports["plain"], reports["fast"]):            # batched views: round-off may flip a few arg-max ties
        assert np.allclose(numbers(a), numbers(b), atol=0.05), (a, b)
    # generate_output.py keeps every `pred` it computed: with fresh outputs they must survive the following forwards
    with torch.no_grad():
        kept = [fast(f)["ssc_logit"] for f in frames]
        want = [plain(f)["ssc_logit"] for f in frames]
        fast.graph_2d = fast.graph_all = False                     # the same batched-views forward, eagerly
        eager = [fast(f)["ssc_logit"] for f in frames]
    assert len({k.data_ptr() for k in kept}) == 3
    for i, (k, e, w) in enumerate(zip(kept, eager, want)):
        scale = float(w.abs().max())
        d_graph, d_plain = float((k - e).abs().max()) / scale, float((k - w).abs().max()) / scale
        print(f"frame {i}: graph replay vs eager (same batched-views forward) {d_graph:.2e}; vs the per-view forward {d_plain:.2e}")
        # a replay IS the eager forward up to the libraries' algorithm choice inside / outside a capture (measured up to
        # 1.3e-4 on this reduced random-init net); a stale static buffer or a frame mix-up would show as O(1)
        assert d_graph < 1e-3, (i, d_graph)
        # per-view vs batched 2-D passes: the library picks other algorithms at batch 2, and this reduced random-init net
        # (logits ~1e6, frames outside its BatchNorm calibration) amplifies that round-off; config 2 pins the parity at 1e-3
        assert d_plain < 3e-2, (i, d_plain)
    assert float((kept[0] - kept[2]).abs().max()) > 0              # (the frames really differ)
