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
