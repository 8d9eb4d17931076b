"""Synthetic SemanticKITTI-shaped frames for bench.py / tools (no dataset in the image).

Same collate schema as the reference dataloader (data/semantic_kitti/collate.py:62-83): `img` (B, 2, 3, H, W),
per-sample lists of `cam_k` (2, 3, 3) float64, `T_velo_2_cam` (2, 4, 4) float32, `ida_mats` (2, 4, 4).  The
calibration is sequence 00's P2 intrinsics and a rectified stereo pair 0.54 m apart.  The voxel->pixel tables
(`projected_pix_2`, `fov_mask_2`) are NOT generated here: `attach_projection` computes them on the GPU with the
product's own kernel (occd_project_voxels), which the parity tests pin bit-exactly to the numba
`vox2pix` semantics.
"""
import numpy as np
import torch

KITTI_K = np.array([[707.0912, 0, 601.8873], [0, 707.0912, 183.1104], [0, 0, 1]], dtype=np.float64)
KITTI_TR = np.array([[0, -1, 0, 0], [0, 0, -1, -0.08], [1, 0, 0, -0.27], [0, 0, 0, 1]], dtype=np.float64)
STEREO_BASELINE_M = 0.54


def kitti_frame(batch=1, img_hw=(370, 1220), seed=0):
    """CPU tensors of one stereo batch (images ~ N(0, 1), i.e. already normalised)."""
    g = torch.Generator().manual_seed(seed)
    H, W = img_hw
    tr_right = KITTI_TR.copy()
    tr_right[0, 3] = -STEREO_BASELINE_M
    return {
        "img": torch.randn(batch, 2, 3, H, W, generator=g),
        "cam_k": [torch.from_numpy(np.stack([KITTI_K, KITTI_K])) for _ in range(batch)],
        "T_velo_2_cam": [torch.from_numpy(np.stack([KITTI_TR, tr_right]).astype(np.float32)) for _ in range(batch)],
        "T_velo_2_cam_f64": [torch.from_numpy(np.stack([KITTI_TR, tr_right])) for _ in range(batch)],
        "ida_mats": [torch.eye(4).repeat(2, 1, 1) for _ in range(batch)],
    }


def to_device(batch, device):
    out = {}
    for k, v in batch.items():
        if isinstance(v, list):
            out[k] = [t.to(device) if torch.is_tensor(t) else t for t in v]
        else:
            out[k] = v.to(device) if torch.is_tensor(v) else v
    return out


def attach_projection(model, batch):
    """Add the dataloader's `projected_pix_{s}` / `fov_mask_{s}` entries, computed by the model's GPU projection."""
    s = model.project_scale
    pix, fov = model.project_voxels_on_gpu(batch, batch["img"])
    batch[f"projected_pix_{s}"] = [p for p in pix]
    batch[f"fov_mask_{s}"] = [m for m in fov]
    return batch


def attach_training_targets(model, batch, cfg, seed=1):
    """Training-only entries of the reference's collate schema for a synthetic frame already on the GPU
    (semantic_kitti/collate.py:62-83): `target` uint8 with 255 = unlabelled, `frustums_masks` / `frustums_class_dists`
    for frustum_size^2 image-plane frustums, `CP_mega_matrices` shaped after the model's relation logits, sparse
    `gt_depth`.  Values are random (there is no dataset here); shapes, dtypes and sparsity patterns are the real ones."""
    dev = batch["img"].device
    bs = batch["img"].shape[0]
    C = cfg.n_classes
    dims = tuple(cfg.full_scene_size)
    g = torch.Generator(device=dev).manual_seed(seed)
    target = torch.randint(0, C, (bs, *dims), device=dev, generator=g).to(torch.uint8)
    target[torch.rand(bs, *dims, device=dev, generator=g) < 0.5] = 0
    target[torch.rand(bs, *dims, device=dev, generator=g) < 0.2] = 255
    fs = cfg.frustum_size
    nf = fs * fs
    fid = (torch.arange(dims[0], device=dev).view(-1, 1, 1) * fs // dims[0]) * fs + \
        (torch.arange(dims[1], device=dev).view(1, -1, 1) * fs // dims[1]) + \
        torch.zeros(1, 1, dims[2], device=dev, dtype=torch.long)
    masks = torch.stack([fid == f for f in range(nf)], 0)
    batch["target"] = target
    batch["frustums_masks"] = [masks for _ in range(bs)]
    batch["frustums_class_dists"] = [torch.rand(nf, C, device=dev, generator=g) for _ in range(bs)]
    H, W = batch["img"].shape[-2:]
    gt = torch.rand(bs, 1, H, W, device=dev, generator=g) * 50.0
    gt[torch.rand(bs, 1, H, W, device=dev, generator=g) < 0.7] = 0.0
    batch["gt_depth"] = gt
    if cfg.context_prior:
        X, Y, Z = (d // 8 for d in dims)               # relation logits: (B, R, mega voxels at 1/16, voxels at 1/8)
        n, m = X * Y * Z, (X // 2) * (Y // 2) * (Z // 2)
        batch["CP_mega_matrices"] = [(torch.rand(cfg.n_relations, n, m, device=dev, generator=g) < 0.3).float()
                                     for _ in range(bs)]
    return batch
