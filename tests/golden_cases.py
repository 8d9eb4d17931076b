"""Seeded input / configuration definitions shared by tests/golden/make_golden.py (which feeds them
to the real reference) and the parity tests (which feed them to the oracle and to the HIP path).
Everything is rebuilt from numpy's frozen MT19937 stream so the GPU box sees bit-identical inputs."""
import zlib

import numpy as np
import torch

from occdepth_amd import configs

SEED = 3


def _rs(tag):
    return np.random.RandomState(zlib.crc32(str(tag).encode()) % (2 ** 32))


def randn(shape, tag):
    return torch.from_numpy(_rs(tag).standard_normal(tuple(shape)).astype(np.float32))


# ------------------------------------------------------------------------------------------ SFA
SFA_CASES = {
    "kitti_v2": dict(dataset="kitti", scene=(16, 12, 8), ps=1, C=8, V=2, P=1, hw=(23, 31)),
    "kitti_v2_pat5": dict(dataset="kitti", scene=(16, 12, 8), ps=2, C=32, V=2, P=5, hw=(19, 40)),
    "nyu_v2": dict(dataset="NYU", scene=(10, 6, 8), ps=1, C=12, V=2, P=1, hw=(24, 32)),
    "kitti_v1": dict(dataset="kitti", scene=(8, 8, 4), ps=1, C=24, V=1, P=1, hw=(11, 17)),
    "kitti_v3": dict(dataset="kitti", scene=(8, 8, 4), ps=1, C=16, V=3, P=2, hw=(11, 17)),
}


def sfa_inputs(spec):
    rs = _rs(("sfa", spec["dataset"], spec["C"], spec["V"], spec["P"]))
    h, w = spec["hw"]
    n = int(np.prod([s // spec["ps"] for s in spec["scene"]]))
    x2d = torch.from_numpy(rs.standard_normal((spec["V"], spec["C"], h, w)).astype(np.float32))
    px = rs.randint(0, w, (spec["V"], n, spec["P"], 1))
    py = rs.randint(0, h, (spec["V"], n, spec["P"], 1))
    pix = torch.from_numpy(np.concatenate([px, py], -1).astype(np.int64))
    fov = torch.from_numpy(rs.uniform(size=(spec["V"], n, spec["P"])) < 0.6)
    return x2d, pix, fov


# ------------------------------------------------------------------------------------------ 3-D blocks
BLOCK_CASES = {
    "bottleneck_d2": (1, 32, 8, 12, 16), "bottleneck_odd": (1, 100, 7, 9, 15), "process": (2, 32, 8, 8, 8),
    "downsample": (1, 32, 8, 12, 16), "downsample_odd": (1, 24, 10, 6, 30), "upsample": (1, 64, 4, 6, 8),
    "upsample_odd": (1, 40, 5, 3, 5), "convblock": (1, 32, 6, 6, 8), "aspp": (1, 32, 8, 8, 4),
    "head": (1, 16, 8, 12, 16), "head_cascade": (1, 8, 8, 8, 32), "head_occluded": (1, 16, 8, 8, 8),
    "crp": (1, 64, 8, 8, 2), "crp_odd": (2, 32, 5, 3, 5),
}

UNET3D_CASES = {
    "kitti_ps2": dict(kind="kitti", classes=20, scene=(64, 64, 16), feature=16, ps=2, occluded=False,
                      x=(1, 16, 32, 32, 8)),
    "kitti_ps1": dict(kind="kitti", classes=20, scene=(32, 32, 8), feature=16, ps=1, occluded=True,
                      x=(1, 16, 32, 32, 8)),
    "nyu_odd": dict(kind="nyu", classes=12, scene=(20, 12, 20), feature=20, n_relations=2, x=(1, 20, 20, 12, 20)),
}

# ------------------------------------------------------------------------------------------ FLoSP-Depth
_SMALL_FLOSP = dict(x_bound=[0, 12.8, 0.2], y_bound=[-6.4, 6.4, 0.2], z_bound=[-2, 1.2, 0.2],
                    d_bound=[2.0, 14.0, 0.5], final_dim=(96, 320), downsample_factor=8, output_channels=16,
                    depth_net_conf=dict(in_channels=16, mid_channels=32), scene_size=(64, 64, 16), project_scale=2,
                    return_depth=True)
FLOSP_CASES = {
    "stereo": dict(ctor=_SMALL_FLOSP, n_cams=2, flip=False),
    "stereo_flip": dict(ctor=_SMALL_FLOSP, n_cams=2, flip=True),
    "mono": dict(ctor=_SMALL_FLOSP, n_cams=1, flip=False),
}


def flosp_inputs(spec):
    n = spec["n_cams"]
    feat = randn((1, n, 16, 12, 40), ("flosp", n))
    k = torch.tensor([[185.4, 0, 157.9], [0, 185.4, 48.0], [0, 0, 1]], dtype=torch.float64)
    tr = torch.tensor([[0, -1, 0, 0], [0, 0, -1, -0.08], [1, 0, 0, -0.27], [0, 0, 0, 1]], dtype=torch.float32)
    tr2 = tr.clone()
    tr2[0, 3] = -0.54
    ida = torch.eye(4)
    ida2 = ida.clone()
    if spec["flip"]:
        ida2[0, 0] = -1
        ida2[0, 3] = 319
    return feat, [torch.stack([k] * n)], [torch.stack([tr, tr2][:n])], [torch.stack([ida, ida2][:n])]


# ------------------------------------------------------------------------------------------ whole model
def occdepth_config(name):
    """-> (config, flosp_depth_conf overrides).  Small variants shrink the scene / image / widths but
    keep every structural switch of the shipped yaml they derive from."""
    if name == "kitti_a100":
        return configs.kitti_a100.clone(), None
    if name == "nyu_2080ti":                      # BASELINE configs[0]: the reference's CPU-runnable NYUv2 case
        return configs.nyu_2080ti.clone(), None
    if name == "kitti_small":
        cfg = configs.kitti_a100.clone(full_scene_size=(64, 64, 16), feature=16, feature_2d_oc=16,
                                       backbone_2d_name="tf_efficientnet_b3_ns")
        conf = dict(x_bound=[0, 12.8, 0.2], y_bound=[-6.4, 6.4, 0.2], z_bound=[-2, 1.2, 0.2],
                    d_bound=[2.0, 14.0, 0.5], final_dim=(96, 320))
        return cfg, conf
    if name == "kitti_flosp_small":
        cfg = configs.kitti_flosp_a100.clone(full_scene_size=(64, 64, 16), feature=16, feature_2d_oc=16,
                                             backbone_2d_name="tf_efficientnet_b3_ns")
        return cfg, None
    if name == "nyu_small":
        cfg = configs.nyu_2080ti.clone(full_scene_size=(20, 12, 20), feature=20, feature_2d_oc=20,
                                       backbone_2d_name="tf_efficientnet_b3_ns")
        return cfg, None
    raise KeyError(name)


def occdepth_batch(name):
    from oracle import inputs
    if name == "kitti_a100":
        return inputs.kitti_batch(seed=SEED)
    if name in ("kitti_small", "kitti_flosp_small"):
        b = inputs.kitti_batch(img_hw=(96, 320), scene=(64, 64, 16), project_scale=2, seed=SEED, scale_k=320 / 1220)
        return b
    if name == "nyu_2080ti":
        return inputs.nyu_batch(seed=SEED)
    if name == "nyu_small":
        return inputs.nyu_batch(img_hw=(120, 160), scene=(20, 12, 20), seed=SEED, scale_k=0.25)
    raise KeyError(name)


def subsample(t):
    """Deterministic strided sample of a big output tensor (full-size golden fixtures)."""
    if t.dim() == 5 and t.shape[-1] * t.shape[-2] * t.shape[-3] > 1 << 16:
        return t[:, :, 3::7, 2::7, 1::5].contiguous()
    if t.dim() == 5:
        return t[:, :, ::2, ::2, ::2].contiguous()
    if t.dim() == 4 and t.numel() > 1 << 18:
        return t[:, :, 5::29, 3::23].contiguous()
    return t


UNET3D_512 = dict(classes=20, scene=(512, 512, 64), feature=64, ps=2, x=(1, 64, 256, 256, 32))   # BASELINE configs[4]


def subsample_512(t):
    """Coarser deterministic sample for the configs[4] fixture (outputs up to 2.1 GB each)."""
    if t.dim() == 5:
        return t[:, :, 3::13, 2::13, 1::9].contiguous() if t.shape[2] > 64 else t[:, :, 1::3, 2::3, 1::2].contiguous()
    if t.dim() == 4:
        return t[:, :, 5::61, 3::47].contiguous()
    return t


def maybe_subsample(a, limit=60000):
    """numpy / torch array -> itself when small, else subsample()d (same rule on both sides of a test)."""
    t = torch.as_tensor(a)
    if t.numel() <= limit:
        return a
    r = subsample(t)
    return r.numpy() if isinstance(a, np.ndarray) else r


# ---------------------------------------------------------------------------------------------------------------
# training-step inputs (N1): logits / targets / frustum masks / relation matrices / depth maps, all from seeds
LOSS_CASES = {
    # name: (bs, n_classes, (X, Y, Z), n_frustums per side, n_relations, (mega voxels, N), depth (ncam, D, h, w, factor))
    "kitti_like": (2, 20, (16, 16, 8), 4, 4, (24, 40), (1, 24, 6, 10, 4)),
    "nyu_like": (1, 12, (10, 6, 10), 2, 4, (15, 30), (2, 24, 5, 7, 4)),
}
LOSS_D_BOUND = [2.0, 14.0, 0.5]


def loss_case(name):
    """Seeded inputs of one training step.  Some classes are absent from the target, a band of voxels is
    unlabelled (255), one frustum is empty and one has no ground-truth counts: every branch of the reference runs."""
    import torch
    bs, c, dims, fside, nrel, (mega, nvox), (ncam, dbins, dh, dw, factor) = LOSS_CASES[name]
    g = torch.Generator().manual_seed(SEED + 17 + len(name))
    ssc = torch.randn(bs, c, *dims, generator=g) * 2.0
    occ = torch.randn(bs, 2, *dims, generator=g)
    present = [0, 1, 2, 3, 5, 8, c - 1]                           # the other classes never appear
    target = torch.tensor(present)[torch.randint(0, len(present), (bs, *dims), generator=g)]
    target[torch.rand(bs, *dims, generator=g) < 0.55] = 0         # mostly empty, like a real scene
    target[:, :, :, : dims[2] // 4][torch.rand(bs, dims[0], dims[1], dims[2] // 4, generator=g) < 0.7] = 255
    # frustums: fside x fside image-plane grid; a voxel belongs to at most one frustum
    nf = fside * fside
    fid = torch.randint(0, nf + 2, (bs, *dims), generator=g)      # ids nf, nf+1: outside the field of view
    masks = torch.stack([fid == f for f in range(nf)], 1)         # (bs, F, X, Y, Z) bool
    masks[:, 1] = False                                           # an empty frustum (total_prob == 0)
    dists = torch.zeros(bs, nf, c)
    for b in range(bs):
        for f in range(nf):
            sel = masks[b, f] & (target[b] != 255)
            dists[b, f] = torch.bincount(target[b][sel], minlength=c).float()
    dists[:, 2] = 0                                               # a frustum without any count (total_cnt == 0)
    p_logits = torch.randn(bs, nrel, mega, nvox, generator=g)
    cp = [(torch.rand(nrel, nvox, mega, generator=g) < 0.2).float() for _ in range(bs)]
    depth_pred = torch.softmax(torch.randn(bs, ncam, dbins, dh, dw, generator=g), 2)
    gt_depth = torch.rand(bs, ncam, dh * factor + 3, dw * factor + 5, generator=g) * 16.0
    gt_depth[torch.rand(gt_depth.shape, generator=g) < 0.6] = 0.0  # sparse lidar-like ground truth
    weights = 0.5 + torch.rand(c, generator=g) * 2.0
    weights_occ = torch.tensor([0.7, 1.9])
    return {"ssc_logit": ssc, "occ_logit": occ, "target": target, "frustums_masks": masks,
            "frustums_class_dists": dists, "P_logits": p_logits, "CP_mega_matrices": cp, "depth_pred": depth_pred,
            "gt_depth": gt_depth, "class_weights": weights, "class_weights_occ": weights_occ,
            "depth_factor": factor, "d_bound": LOSS_D_BOUND, "n_classes": c}


def train_extras(cfg_name, out_shapes, scene, n_classes, img_hw):
    """Seeded training-only batch entries for an end-to-end step of a small config: target, frustum masks and
    class distributions, relation matrices (shaped after the model's P_logits), sparse ground-truth depth."""
    import torch
    g = torch.Generator().manual_seed(SEED + 101 + len(cfg_name))
    bs = 1
    target = torch.randint(0, n_classes, (bs, *scene), generator=g)
    target[torch.rand(bs, *scene, generator=g) < 0.5] = 0
    target[torch.rand(bs, *scene, generator=g) < 0.15] = 255
    nf = 4
    fid = torch.randint(0, nf + 1, (bs, *scene), generator=g)
    masks = torch.stack([fid == f for f in range(nf)], 1)
    dists = torch.zeros(bs, nf, n_classes)
    for f in range(nf):
        sel = masks[0, f] & (target[0] != 255)
        dists[0, f] = torch.bincount(target[0][sel], minlength=n_classes).float()
    extras = {"target": target.to(torch.uint8), "frustums_masks": list(masks), "frustums_class_dists": list(dists)}
    if "P_logits" in out_shapes:
        _, r, a, b = out_shapes["P_logits"]
        extras["CP_mega_matrices"] = [(torch.rand(r, b, a, generator=g) < 0.3).float() for _ in range(bs)]
    if "depth_pred" in out_shapes:
        gt = torch.rand(bs, 1, *img_hw, generator=g) * 14.0 + 1.0
        gt[torch.rand(gt.shape, generator=g) < 0.7] = 0.0
        extras["gt_depth"] = gt
    return extras


TRAIN_GRAD_KEYS = 12      # parameters (sorted by name, evenly spaced) whose gradients are stored in the fixture


def pick_grad_keys(named_grads):
    keys = sorted(k for k, g in named_grads.items() if g is not None)
    step = max(1, len(keys) // TRAIN_GRAD_KEYS)
    return keys[::step][:TRAIN_GRAD_KEYS]


CLASSIFIER_SCALE = 0.02


def is_classifier_param(key):
    import re
    return re.search(r"(conv_classes|occ_classes)\.(weight|bias)$", key) is not None
