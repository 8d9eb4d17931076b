"""Generate the golden fixtures by running the REAL reference (/root/reference) on CPU.

Run in the build container only:   python tests/golden/make_golden.py [case ...]
Outputs tests/golden/*.npz (+ state_keys_*.json).  Inputs are rebuilt from seeds by
tests/golden_cases.py (shared with the tests); weights come from oracle/fill.py; only the
reference's OUTPUTS are stored (sub-sampled for the full-size config).
torch version is recorded in every file (CPU kernel semantics: grid_sample, cosine_similarity).
"""
import json
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle import ref_shims  # noqa: E402
from oracle.fill import bn_stats, calibrate_bn, fill_state_dict  # noqa: E402
import golden_cases as gc  # noqa: E402

ref_shims.install()
from occdepth_amd.models.efficientnet import EfficientNet  # noqa: E402  (the only non-reference piece: encoder)

ref_unet2d = ref_shims.patch_unet2d_build(lambda name: EfficientNet(name))

import occdepth.models.SFA as ref_sfa  # noqa: E402
import occdepth.models.DDR as ref_ddr  # noqa: E402
import occdepth.models.modules as ref_modules  # noqa: E402
import occdepth.models.CRP3D as ref_crp  # noqa: E402
import occdepth.models.unet3d_kitti as ref_u3k  # noqa: E402
import occdepth.models.unet3d_nyu as ref_u3n  # noqa: E402
import occdepth.models.flosp_depth.flosp_depth as ref_flosp  # noqa: E402
import occdepth.models.flosp_depth as ref_flosp_pkg  # noqa: E402
import occdepth.models.OccDepth as ref_occ  # noqa: E402

BN3 = torch.nn.BatchNorm3d


def _np(t):
    return t.detach().cpu().numpy().astype(np.float32)


def _save(name, arrays, meta=None):
    meta = dict(meta or {}, torch=torch.__version__, generated=time.strftime("%Y-%m-%d"))
    arrays = dict(arrays)
    arrays["__meta__"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrays)
    print(f"wrote {path} ({os.path.getsize(path) / 1e6:.2f} MB)")


def _load_filled(mod, seed, run=None, tag=None, arrays=None):
    """Seeded weights; when `run` is given, BatchNorm statistics are calibrated on that forward pass
    and stored in `arrays` as '<tag>::<key>' so the tests can rebuild the identical state_dict."""
    mod.load_state_dict(fill_state_dict(mod.state_dict(), seed), strict=True)
    mod.eval()
    if run is not None:
        calibrate_bn(mod, run)
        for k, v in bn_stats(mod.state_dict()).items():
            arrays[f"{tag}::{k}"] = v
    return mod.eval()


def _flat(out, prefix=""):
    res = {}
    if isinstance(out, dict):
        for k, v in out.items():
            res.update(_flat(v, f"{prefix}{k}"))
    elif isinstance(out, (tuple, list)):
        for i, v in enumerate(out):
            res.update(_flat(v, f"{prefix}{i}"))
    elif out is not None:
        res[prefix] = _np(out)
    return res


# --------------------------------------------------------------------------------------------- cases
def case_sfa():
    arrays = {}
    for name, spec in gc.SFA_CASES.items():
        x2d, pix, fov = gc.sfa_inputs(spec)
        m = ref_sfa.SFA(spec["scene"], spec["dataset"], spec["ps"])
        arrays[name] = _np(m(x2d, pix, fov))
    _save("sfa", arrays)


def case_blocks3d():
    arrays = {}
    makers = {
        "bottleneck_d2": lambda: ref_ddr.Bottleneck3D(32, 8, BN3, dilation=[2, 2, 2]),
        "bottleneck_odd": lambda: ref_ddr.Bottleneck3D(100, 25, BN3, dilation=[3, 3, 3]),
        "process": lambda: ref_modules.Process(32, BN3, 0.1),
        "downsample": lambda: ref_modules.Downsample(32, BN3, 0.1),
        "downsample_odd": lambda: ref_modules.Downsample(24, BN3, 0.1),
        "upsample": lambda: ref_modules.Upsample(64, 32, BN3, 0.1),
        "upsample_odd": lambda: ref_modules.Upsample(40, 20, BN3, 0.1),
        "convblock": lambda: ref_modules.Convblock3d(32, 16, BN3, 0.1),
        "aspp": lambda: ref_modules.ASPP(32, [1, 2, 3]),
        "head": lambda: ref_modules.SegmentationHead(16, 16, 12, [1, 2, 3]),
        "head_cascade": lambda: ref_modules.SegmentationHeadCascadeCLS(8, 8, 20, [1, 2, 3]),
        "head_occluded": lambda: ref_modules.SegmentationHeadOccludedCLS(16, 16, 20, [1, 2, 3]),
        "crp": lambda: ref_crp.CPMegaVoxels(64, (8, 8, 2), bn_momentum=0.1),
        "crp_odd": lambda: ref_crp.CPMegaVoxels(32, (5, 3, 5), n_relations=2, bn_momentum=0.1),
    }
    for name, shape in gc.BLOCK_CASES.items():
        x = gc.randn(shape, name)
        m = _load_filled(makers[name](), gc.SEED, lambda mm: mm(x), name, arrays)
        with torch.no_grad():
            out = m(x)
        for k, v in _flat(out).items():
            arrays[name + ("." + k if k else "")] = v
    _save("blocks3d", arrays)


def case_unet3d():
    arrays = {}
    for name, spec in gc.UNET3D_CASES.items():
        if spec["kind"] == "nyu":
            m = ref_u3n.UNet3D(spec["classes"], BN3, feature=spec["feature"], full_scene_size=spec["scene"],
                               context_prior=True, n_relations=spec["n_relations"])
        else:
            m = ref_u3k.UNet3D(spec["classes"], BN3, spec["scene"], spec["feature"], spec["ps"], context_prior=True,
                               cascade_cls=True, occluded_cls=spec["occluded"])
        x = gc.randn(spec["x"], name)
        m = _load_filled(m, gc.SEED, lambda mm: mm({"x3d": x}), name, arrays)
        with torch.no_grad():
            out = m({"x3d": x})
        for k, v in _flat(out).items():
            arrays[f"{name}.{k}"] = gc.maybe_subsample(v)
    _save("unet3d", arrays)


def case_unet3d_512():
    """BASELINE configs[4]: the reference UNet3D(kitti) alone on the synthetic 512x512x64 grid (lift volume
    256x256x32, feature 64; 9.3 TFLOP, ~1-2 min per forward on these host cores).  Stored: BatchNorm statistics
    calibrated on the same input, sub-sampled outputs, and each full output's |max| and sum."""
    spec = gc.UNET3D_512
    arrays = {}
    m = ref_u3k.UNet3D(spec["classes"], BN3, spec["scene"], spec["feature"], spec["ps"], context_prior=True,
                       cascade_cls=True)
    x = gc.randn(spec["x"], "unet3d_512")
    t0 = time.time()
    m = _load_filled(m, gc.SEED, lambda mm: mm({"x3d": x}), "unet3d_512", arrays)
    print(f"calibration forward {time.time() - t0:.0f} s")
    t0 = time.time()
    with torch.no_grad():
        out = m({"x3d": x})
    print(f"reference forward {time.time() - t0:.0f} s")
    for k, v in out.items():
        arrays[f"unet3d_512.{k}"] = _np(gc.subsample_512(v))
        arrays[f"unet3d_512.{k}.absmax"] = np.float64(v.abs().max())
        arrays[f"unet3d_512.{k}.sum"] = np.float64(v.double().sum())
        print(k, tuple(v.shape), float(v.abs().max()))
    _save("unet3d_512", arrays)


def case_flosp():
    arrays = {}
    for name, spec in gc.FLOSP_CASES.items():
        m = ref_flosp.FlospDepth(**spec["ctor"])
        feat, cam_k, t_v2c, idas = gc.flosp_inputs(spec)
        m = _load_filled(m, gc.SEED, lambda mm: mm(feat, cam_k, t_v2c, idas), name, arrays)
        with torch.no_grad():
            vox, depth = m(feat, cam_k, t_v2c, idas)
            ida = torch.stack(idas)
            tv = torch.stack(t_v2c).float()
            k3 = torch.stack(cam_k).float()
            intr = k3.new_zeros(1, k3.shape[1], 4, 4)
            intr[:, :, :3, :3] = k3
            intr[:, :, 3, 3] = 1
            shape = k3.new_zeros(1, 2)
            shape[:, 0:2] = torch.as_tensor(spec["ctor"]["final_dim"])
            grid = m.grid_generator(lidar_to_cam=tv[:, 0], cam_to_img=intr[:, 0, :3, :], ida_mats=ida[:, 0],
                                    image_shape=shape)
        arrays[name + ".vox"] = _np(vox)
        arrays[name + ".depth"] = _np(depth)
        arrays[name + ".grid0"] = _np(grid)
    _save("flosp", arrays)


def case_decoder2d():
    name = "tf_efficientnet_b3_ns"
    m = ref_unet2d.UNet2D.build(out_feature=16, use_decoder=True, backbone_2d_name=name, return_up_feats=1)
    arrays = {}
    x = gc.randn((1, 3, 74, 122), "decoder2d")
    m = _load_filled(m, gc.SEED, lambda mm: mm(x), "decoder2d", arrays)
    with torch.no_grad():
        out = m(x)
    arrays.update({k: _np(v) for k, v in out.items()})
    _save("decoder2d", arrays)


_PRISTINE_CONF = {}


def _build_ref_occdepth(cfg_name, arrays):
    cfg, conf_override = gc.occdepth_config(cfg_name)
    conf_map = ref_flosp_pkg.flosp_depth_conf_map
    if cfg.dataset not in _PRISTINE_CONF:
        _PRISTINE_CONF[cfg.dataset] = dict(conf_map[cfg.dataset])
    conf_map[cfg.dataset].clear()          # the reference mutates this module-level dict in place
    conf_map[cfg.dataset].update(_PRISTINE_CONF[cfg.dataset])
    if conf_override:
        conf_map[cfg.dataset].update(conf_override)
    from occdepth_amd.configs import PROJECT_RES
    m = ref_occ.OccDepth(class_names=[str(i) for i in range(cfg.n_classes)], class_weights=torch.ones(cfg.n_classes),
                         class_weights_occ=torch.ones(2), full_scene_size=tuple(cfg.full_scene_size),
                         project_res=PROJECT_RES, config=cfg)
    batch = gc.occdepth_batch(cfg_name)
    return _load_filled(m, gc.SEED, lambda mm: mm(batch), cfg_name, arrays), cfg, batch


def case_occdepth_small():
    arrays, keys = {}, {}
    for cfg_name in ("kitti_small", "nyu_small", "kitti_flosp_small"):
        m, cfg, batch = _build_ref_occdepth(cfg_name, arrays)
        with torch.no_grad():
            out = m(batch)
        for k, v in _flat(out).items():
            arrays[f"{cfg_name}.{k}"] = gc.maybe_subsample(v)
        keys[cfg_name] = {k: list(v.shape) for k, v in m.state_dict().items()}
    _save("occdepth_small", arrays)
    with open(os.path.join(HERE, "state_keys_small.json"), "w") as f:
        json.dump(keys, f, indent=0, sort_keys=True)


def _full_case(cfg_name, what):
    arrays = {}
    m, cfg, batch = _build_ref_occdepth(cfg_name, arrays)
    t0 = time.time()
    with torch.no_grad():
        out = m(batch)
    print("reference %s forward: %.1f s" % (what, time.time() - t0))
    for k, v in _flat(out).items():
        arrays[k] = gc.subsample(torch.from_numpy(v)).numpy()
        arrays[k + ".absmax"] = np.float32(np.abs(v).max())
        arrays[k + ".sum"] = np.float64(v.astype(np.float64).sum())
    _save("occdepth_" + cfg_name, arrays, meta={"subsample": "tests/golden_cases.py:subsample"})
    keys = {k: list(v.shape) for k, v in m.state_dict().items()}
    with open(os.path.join(HERE, "state_keys_%s.json" % cfg_name), "w") as f:
        json.dump(keys, f, indent=0, sort_keys=True)


def case_occdepth_full():
    """BASELINE config 2 (B7, 370x1220 stereo, 256x256x32): one reference frame, sub-sampled outputs."""
    _full_case("kitti_a100", "config-2")


def case_occdepth_nyu():
    """BASELINE config 1 (NYUv2 RGB-D frame, B4, feature 100, 60x36x60, virtual stereo), full size."""
    _full_case("nyu_2080ti", "config-1 (NYU)")


def case_losses():
    """Training-step losses (N1) and SSC metrics (N4): the reference's own functions, and its `OccDepth.step`
    with the forward pass replaced by a fixed out_dict, on the seeded inputs of golden_cases.loss_case."""
    import occdepth.loss.ssc_loss as ref_ssc
    import occdepth.loss.CRP_loss as ref_crp_loss
    import occdepth.loss.depth_loss as ref_depth
    import occdepth.loss.sscMetrics as ref_metrics
    arrays = {}
    for name, cfg_name in (("kitti_like", "kitti_small"), ("nyu_like", "nyu_small")):
        d = gc.loss_case(name)
        ssc, tgt = d["ssc_logit"], d["target"]
        arrays[f"{name}.ce"] = np.float64(ref_ssc.CE_ssc_loss(ssc, tgt, d["class_weights"]))
        arrays[f"{name}.sem_scal"] = np.float64(ref_ssc.sem_scal_loss(ssc, tgt))
        arrays[f"{name}.geo_scal"] = np.float64(ref_ssc.geo_scal_loss(ssc, tgt))
        arrays[f"{name}.relation"] = np.float64(ref_crp_loss.compute_super_CP_multilabel_loss(d["P_logits"],
                                                                                             d["CP_mega_matrices"]))
        dl = ref_depth.DepthClsLoss(d["depth_factor"], d["d_bound"])
        arrays[f"{name}.depth"] = np.float64(dl.get_depth_loss(d["gt_depth"], d["depth_pred"]))
        arrays[f"{name}.depth_onehot"] = _np(dl._get_downsampled_gt_depth(
            d["gt_depth"][:, :, : d["depth_pred"].shape[3] * d["depth_factor"], : d["depth_pred"].shape[4] * d["depth_factor"]]
            .contiguous()))
        # the step itself: every loss the shipped config switches on + the inline frustum loss + the metric update
        m, cfg, _ = _build_ref_occdepth(cfg_name, {})
        m.class_weights, m.class_weights_occ = d["class_weights"], d["class_weights_occ"]
        if hasattr(m, "depth_loss_fn"):
            m.depth_loss_fn = dl
        leaves = {k: d[k].clone().requires_grad_(True) for k in ("ssc_logit", "occ_logit", "P_logits", "depth_pred")}
        m.forward = lambda batch: dict(leaves)
        logged = {}
        m.log = lambda key, val, **kw: logged.__setitem__(key, float(val))
        bs = ssc.shape[0]
        batch = {"img": torch.zeros(bs, 1), "target": tgt, "CP_mega_matrices": d["CP_mega_matrices"],
                 "gt_depth": d["gt_depth"], "frustums_masks": list(d["frustums_masks"]),
                 "frustums_class_dists": list(d["frustums_class_dists"])}
        metric = ref_metrics.SSCMetrics(d["n_classes"])
        m.cur_batch = 7
        loss = m.step(batch, "train", metric)
        loss.backward()
        for k, v in logged.items():
            arrays[f"{name}.step.{k}"] = np.float64(v)
        arrays[f"{name}.step.total"] = np.float64(loss.detach())
        for k, t in leaves.items():
            if t.grad is not None:
                arrays[f"{name}.grad.{k}"] = _np(t.grad)
        st = metric.get_stats()
        for k in ("precision", "recall", "iou", "iou_ssc_mean"):
            arrays[f"{name}.metric.{k}"] = np.float64(st[k])
        arrays[f"{name}.metric.iou_ssc"] = np.asarray(st["iou_ssc"], np.float64)
        arrays[f"{name}.metric.tps"] = np.asarray(metric.tps, np.float64)
        arrays[f"{name}.metric.fps"] = np.asarray(metric.fps, np.float64)
        arrays[f"{name}.metric.fns"] = np.asarray(metric.fns, np.float64)
        arrays[f"{name}.metric.completion"] = np.array([metric.completion_tp, metric.completion_fp,
                                                         metric.completion_fn], np.float64)
        print(name, {k: round(v, 5) for k, v in logged.items()})
    _save("losses", arrays)


def case_train_step():
    """End-to-end step (N1): forward + the reference's own `step` + backward on the small configs, BatchNorm in
    eval mode (running statistics) so that the result is a deterministic function of the stored state."""
    import occdepth.loss.sscMetrics as ref_metrics
    arrays = {}
    for cfg_name in ("kitti_small", "nyu_small"):
        m, cfg, batch = _build_ref_occdepth(cfg_name, {})
        m.eval()
        # random-init classifier convolutions give logits of +-300 (softmax in float32 denormals): scale them so
        # that the step is conditioned like a real one; the scaled tensors travel in the fixture
        with torch.no_grad():
            for k, p_ in m.named_parameters():
                if gc.is_classifier_param(k):
                    p_.mul_(gc.CLASSIFIER_SCALE)
                    arrays[f"{cfg_name}.override.{k}"] = _np(p_)
            out = m(batch)
        print(cfg_name, "ssc_logit absmax", float(out["ssc_logit"].abs().max()))
        shapes = {k: tuple(v.shape) for k, v in out.items() if torch.is_tensor(v)}
        extras = gc.train_extras(cfg_name, shapes, tuple(cfg.full_scene_size), cfg.n_classes, batch["img"].shape[-2:])
        batch = dict(batch, **extras)
        logged = {}
        m.log = lambda key, val, **kw: logged.__setitem__(key, float(val))
        metric = ref_metrics.SSCMetrics(cfg.n_classes)
        m.cur_batch = 3
        m.zero_grad()
        loss = m.step(batch, "train", metric)
        loss.backward()
        grads = {k: p.grad for k, p in m.named_parameters()}
        for k, v in logged.items():
            arrays[f"{cfg_name}.{k}"] = np.float64(v)
        arrays[f"{cfg_name}.no_grad_keys"] = np.frombuffer(
            json.dumps(sorted(k for k, g in grads.items() if g is None)).encode(), dtype=np.uint8)
        for k in gc.pick_grad_keys(grads):
            arrays[f"{cfg_name}.grad.{k}"] = _np(grads[k]).reshape(-1)[:4096]
            arrays[f"{cfg_name}.gradnorm.{k}"] = np.float64(grads[k].double().norm())
        arrays[f"{cfg_name}.metric.tps"] = np.asarray(metric.tps, np.float64)
        print(cfg_name, {k: round(v, 5) for k, v in logged.items()}, "params without grad:",
              sum(g is None for g in grads.values()), "/", len(grads))
    _save("train_step_small", arrays)


def case_train_step_full():
    """The step bench.py TIMES (`--train`: BASELINE configs[2]), once through the real reference at full size: B7, 370x1220
    stereo -> 256x256x32, TRAINING mode (BatchNorm on batch statistics), the reference's `training_step` + `backward`.
    Stored: every logged loss term, the parameters left without gradient, and for 12 parameters spread over the model the
    gradient norm and its first 4096 elements (VERDICT r3 missing #6).  ~25 GB of autograd state, minutes of CPU time."""
    import occdepth.loss.sscMetrics as ref_metrics  # noqa: F401  (the reference's own numpy metric inside training_step)
    arrays = {}
    cfg_name = "kitti_a100"
    t0 = time.time()
    m, cfg, batch = _build_ref_occdepth(cfg_name, {})          # (BatchNorm statistics: the ones occdepth_kitti_a100.npz stores)
    with torch.no_grad():
        for k, p_ in m.named_parameters():
            if gc.is_classifier_param(k):
                p_.mul_(gc.CLASSIFIER_SCALE)
                arrays[f"override.{k}"] = _np(p_)
    shapes = {"P_logits": (1, 4, 512, 4096), "depth_pred": (1, 2, 104, 47, 153)}
    extras = gc.train_extras(cfg_name, shapes, tuple(cfg.full_scene_size), cfg.n_classes, batch["img"].shape[-2:])
    batch = dict(batch, **extras)
    logged = {}
    m.log = lambda key, val, **kw: logged.__setitem__(key, float(val))
    m.train()
    m.cur_batch = 0
    m.zero_grad()
    print("built + calibrated in %.0f s; training_step ..." % (time.time() - t0), flush=True)
    t0 = time.time()
    loss = m.training_step(batch, 0)
    print("forward + losses %.0f s; backward ..." % (time.time() - t0), flush=True)
    t0 = time.time()
    loss.backward()
    print("backward %.0f s" % (time.time() - t0), flush=True)
    grads = {k: p.grad for k, p in m.named_parameters()}
    for k, v in logged.items():
        arrays[k] = np.float64(v)
    arrays["no_grad_keys"] = np.frombuffer(json.dumps(sorted(k for k, g in grads.items() if g is None)).encode(), dtype=np.uint8)
    for k in gc.pick_grad_keys(grads):
        arrays[f"grad.{k}"] = _np(grads[k]).reshape(-1)[:4096]
        arrays[f"gradnorm.{k}"] = np.float64(grads[k].double().norm())
    # a few running statistics after the step (momentum update from the batch statistics of this very frame)
    sd = m.state_dict()
    bn_keys = sorted(k for k in sd if k.endswith("running_var"))
    for k in bn_keys[:: max(1, len(bn_keys) // 8)][:8]:
        arrays[f"running.{k}"] = _np(sd[k]).reshape(-1)[:256]
    print({k: round(v, 5) for k, v in logged.items()}, "params without grad:", sum(g is None for g in grads.values()), "/", len(grads))
    _save("train_step_full", arrays, meta={"mode": "train (batch statistics), cur_batch 0 -> 1, classifier convolutions scaled by %g"
                                                   % gc.CLASSIFIER_SCALE})


CASES = {"train_step_full": case_train_step_full, "sfa": case_sfa, "blocks3d": case_blocks3d, "unet3d": case_unet3d, "unet3d_512": case_unet3d_512,
         "flosp": case_flosp,
         "decoder2d": case_decoder2d, "occdepth_small": case_occdepth_small, "occdepth_full": case_occdepth_full,
         "occdepth_nyu": case_occdepth_nyu, "losses": case_losses, "train_step": case_train_step}

if __name__ == "__main__":
    torch.set_num_threads(os.cpu_count())
    for c in (sys.argv[1:] or list(CASES)):
        print("==", c)
        CASES[c]()
