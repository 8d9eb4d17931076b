"""TEST INFRASTRUCTURE ONLY -- synthetic batches shaped like the reference dataloader's output
(collate schema: data/semantic_kitti/collate.py:62-83, data/NYU/collate.py:50-67) with the
voxel->pixel projection restated from data/utils/helpers.py:94-169 and data/utils/fusion.py:201-217,
234-343,518-522 (numba is absent here, so this numpy restatement is UNPINNED; it only produces
inputs that are fed identically to the reference, the oracle and the HIP path)."""
import numpy as np
import torch

KITTI_K = np.array([[707.0912, 0, 601.8873], [0, 707.0912, 183.1104], [0, 0, 1]], dtype=np.float64)
KITTI_TR = np.array([[0, -1, 0, 0], [0, 0, -1, -0.08], [1, 0, 0, -0.27], [0, 0, 0, 1]], dtype=np.float64)
NYU_K = np.array([[518.8579, 0, 320], [0, 518.8579, 240], [0, 0, 1]], dtype=np.float64)

_PATTERNS = {0: [[0, 0]], 1: [[0, 0], [0, -1], [-1, 0], [1, 0], [0, 1]],
             2: [[0, 0], [-1, -1], [1, 1], [-1, 1], [1, -1]]}


def vox2pix(cam_E, cam_k, vox_origin, voxel_size, img_W, img_H, scene_size, pattern_id=0):
    """helpers.py:94-169 -> projected_pix (N, P, 2) int64, fov_mask (N, P) bool, pix_z (N,)."""
    vox_origin = np.asarray(vox_origin, dtype=np.float64)
    dim = np.ceil(np.asarray(scene_size) / voxel_size).astype(int)
    xv, yv, zv = np.meshgrid(range(dim[0]), range(dim[1]), range(dim[2]), indexing="ij")
    coords = np.stack([xv.ravel(), yv.ravel(), zv.ravel()], 1).astype(np.float32)
    # fusion.py:203-217 vox2world: float32 origin, float64 arithmetic, float32 store
    pts = (vox_origin.astype(np.float32).astype(np.float64) + voxel_size * coords.astype(np.float64)
           + voxel_size * 0.5).astype(np.float32)
    pts_h = np.hstack([pts, np.ones((len(pts), 1), dtype=np.float32)])
    cam = np.dot(np.asarray(cam_E, dtype=np.float64), pts_h.T).T[:, :3]          # fusion.py:518-522
    k = np.asarray(cam_k).astype(np.float32)
    fx, fy, cx, cy = (np.float64(k[0, 0]), np.float64(k[1, 1]), np.float64(k[0, 2]), np.float64(k[1, 2]))
    with np.errstate(divide="ignore", invalid="ignore"):
        xc = np.round(cam[:, 0] * fx / cam[:, 2] + cx)                            # fusion.py:336-337
        yc = np.round(cam[:, 1] * fy / cam[:, 2] + cy)
    xc = np.nan_to_num(xc, nan=-1e9, posinf=1e9, neginf=-1e9).astype(np.int64)
    yc = np.nan_to_num(yc, nan=-1e9, posinf=1e9, neginf=-1e9).astype(np.int64)
    pat = np.asarray(_PATTERNS[pattern_id], dtype=np.int64)
    pix = np.stack([xc[:, None] + pat[None, :, 0], yc[:, None] + pat[None, :, 1]], -1)
    z = cam[:, 2]
    fov = (pix[..., 0] >= 0) & (pix[..., 0] < img_W) & (pix[..., 1] >= 0) & (pix[..., 1] < img_H) & (z[:, None] > 0)
    return pix, fov, z


def kitti_batch(batch=1, img_hw=(370, 1220), scene=(256, 256, 32), project_scale=2, seed=0, scale_k=1.0,
                pattern_id=0):
    """SURVEY.md 8(d) config 2/3 inputs.  `scale_k` shrinks the intrinsics with a reduced test image."""
    g = torch.Generator().manual_seed(seed)
    H, W = img_hw
    k = KITTI_K.copy()
    k[:2] *= scale_k
    tr2 = KITTI_TR.copy()
    tr2[0, 3] = -0.54
    voxel = 0.2 * project_scale
    scene_m = tuple(s * 0.2 for s in scene)
    pix, fov = [], []
    for T in (KITTI_TR, tr2):
        p, m, _ = vox2pix(T, k, (0, -scene_m[1] / 2, -2), voxel, W, H, scene_m, pattern_id)
        pix.append(p)
        fov.append(m)
    pix = torch.from_numpy(np.stack(pix))
    fov = torch.from_numpy(np.stack(fov))
    b = {
        "img": torch.randn(batch, 2, 3, H, W, generator=g),
        f"projected_pix_{project_scale}": [pix.clone() for _ in range(batch)],
        f"fov_mask_{project_scale}": [fov.clone() for _ in range(batch)],
        "cam_k": [torch.from_numpy(np.stack([k, k])) for _ in range(batch)],
        "T_velo_2_cam": [torch.from_numpy(np.stack([KITTI_TR, tr2]).astype(np.float32)) for _ in range(batch)],
        "ida_mats": [torch.eye(4).repeat(2, 1, 1) for _ in range(batch)],
    }
    return b


def nyu_batch(batch=1, img_hw=(480, 640), scene=(60, 36, 60), seed=0, scale_k=1.0):
    """SURVEY.md 8(d) config 1 inputs: one RGB-D view + virtual stereo (nyu_dataset.py:139-190)."""
    g = torch.Generator().manual_seed(seed)
    H, W = img_hw
    k = NYU_K.copy()
    k[:2] *= scale_k
    # camera looking along +y of the voxel volume from outside its near wall
    voxel = 0.08
    scene_m = (scene[0] * voxel, scene[2] * voxel, scene[1] * voxel)       # NYU volume is (x, z, y)-ordered in metres
    pose = np.array([[1, 0, 0, scene_m[0] / 2], [0, 0, 1, -0.5], [0, -1, 0, scene_m[2] / 2], [0, 0, 0, 1]],
                    dtype=np.float64)
    cam_E0 = np.linalg.inv(pose)
    shift = np.eye(4)
    shift[0, 3] = -0.1                                                     # T_cam0_2_cam1, nyu_dataset.py:170-176
    pix, fov = [], []
    for E in (cam_E0, shift @ cam_E0):
        p, m, _ = vox2pix(E, k, (0, 0, 0), voxel, W, H, scene_m, 0)
        pix.append(p)
        fov.append(m)
    pix = torch.from_numpy(np.stack(pix))
    fov = torch.from_numpy(np.stack(fov))
    return {
        "img": torch.randn(batch, 1, 3, H, W, generator=g),
        "gt_depth": torch.rand(batch, 1, H, W, generator=g) * 5.0 + 0.5,
        "virtual_bf": torch.full((batch,), 51.88579 * scale_k),
        "projected_pix_1": [pix.clone() for _ in range(batch)],
        "fov_mask_1": [fov.clone() for _ in range(batch)],
        "cam_k": [torch.from_numpy(k[None]) for _ in range(batch)],
        "T_velo_2_cam": [torch.from_numpy(cam_E0[None].astype(np.float32)) for _ in range(batch)],
        "ida_mats": [torch.eye(4)[None] for _ in range(batch)],
        "vox_origin": torch.zeros(batch, 3, dtype=torch.float64),
    }
