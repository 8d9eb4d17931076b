"""Stereo-SFA 2D->3D feature lift (mirror of occdepth/models/SFA.py:6-106).

Eval mode is the HIP gather kernel K1b (`occd_lift_fwd`): channels-last feature rows, LPV lanes
per voxel, cosine-similarity fusion of the views reduced with DPP inside the wavefront.
`lift_scales` is the fused multi-scale entry used by OccDepth.forward (one launch for all 2-D
scales and the `* depth * 100` of OccDepth.py:339); `SFA.forward` keeps the reference's
per-scale leaf signature.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import hip
from ..fused import needs_autograd
from ..hip import Vox


def voxel_layout(scene_size, project_scale, dataset):
    """(dims of the flat voxel index n, output (X, Y, Z), row strides of n's three digits)."""
    s = [int(v) // int(project_scale) for v in scene_size]
    if dataset == "NYU":
        # n enumerates (s0, s2, s1); the reference then permutes to (s0, s1, s2)   SFA.py:90-97
        X, Y, Z = s[0], s[1], s[2]
        return (s[0], s[2], s[1]), (X, Y, Z), (Y * Z, 1, Z)
    if dataset == "kitti":
        X, Y, Z = s
        return (X, Y, Z), (X, Y, Z), (Y * Z, Z, 1)
    raise NotImplementedError(f"SFA has no voxel layout for dataset {dataset!r}")


def pixel_rows(f):
    """(B, H, W, cs) pixel rows of a (B, C, H, W) feature map: zero-copy when every image already is channels-last with rows
    of >= C floats (what the decoder's 1x1 heads write through K11's NHWC mode or, in bf16-mode training, through K2b; any
    batch stride, e.g. a view of the two stereo views run as one batch), one transpose pass otherwise."""
    cl = f.permute(0, 2, 3, 1)
    cs = cl.stride(2)
    if (f.dtype == torch.float32 and f.is_cuda and cl.stride(3) == 1 and cs % 4 == 0 and cs >= f.shape[1]
            and cl.stride(1) == cs * f.shape[3] and cl.data_ptr() % 16 == 0
            and (f.shape[0] == 1 or cl.stride(0) % 4 == 0)):
        return cl.as_strided((f.shape[0], f.shape[2], f.shape[3], cs), (cl.stride(0), cl.stride(1), cs, 1))
    return hip.nchw_to_nhwc(f.float())


def lift_scales(feats, scale_divs, projected_pix, fov_mask, scene_size, project_scale, dataset,
                depth_scale=None, scale_const=100.0):
    """feats[s][v]: (B, C, h_s, w_s) feature maps (any layout) -> Vox (B, X, Y, Z, C).

    projected_pix (B, V, N, P, 2) int64 at full image resolution, fov_mask (B, V, N, P) bool."""
    n_dims, out_dims, strides = voxel_layout(scene_size, project_scale, dataset)
    rows = []
    for per_scale in feats:
        rows.append([pixel_rows(f) for f in per_scale])
    B, C = feats[0][0].shape[0], feats[0][0].shape[1]
    out = Vox.empty(B, out_dims, C, feats[0][0].device)
    return hip.lift(rows, scale_divs, projected_pix.contiguous(), fov_mask.contiguous(), n_dims, strides, out,
                    depth_scale=depth_scale, scale_const=scale_const)


def lift_scales_proj(feats, scale_divs, cam_E, cam_k, origin, voxel_size, img_wh, scene_size, project_scale, dataset,
                     frustum=None, scale_const=100.0):
    """`lift_scales` without the tables: the kernel projects the voxels from the calibration (hip.lift_proj) and, with a
    `hip.Frustum`, samples the depth frustum and applies `* depth * scale_const` itself."""
    n_dims, out_dims, strides = voxel_layout(scene_size, project_scale, dataset)
    rows = [[pixel_rows(f) for f in per_scale] for per_scale in feats]
    B, C = feats[0][0].shape[0], feats[0][0].shape[1]
    out = Vox.empty(B, out_dims, C, feats[0][0].device)
    return hip.lift_proj(rows, scale_divs, cam_E, cam_k, origin, voxel_size, img_wh, n_dims, strides, out,
                         frustum=frustum, scale_const=scale_const)


class SFA(nn.Module):
    def __init__(self, scene_size, dataset, project_scale):
        super().__init__()
        self.scene_size = scene_size
        self.dataset = dataset
        self.project_scale = project_scale

    def forward(self, x2d, projected_pix, fov_mask):
        if needs_autograd(self):
            return self._forward_autograd(x2d, projected_pix, fov_mask)
        feats = [[x2d[v:v + 1] for v in range(x2d.shape[0])]]
        vox = lift_scales(feats, [1], projected_pix.unsqueeze(0), fov_mask.unsqueeze(0), self.scene_size,
                          self.project_scale, self.dataset)
        return vox.ncdhw()[0]

    # differentiable ATen form used while training (autograd through the gather)
    def _forward_autograd(self, x2d, projected_pix, fov_mask):
        V, C, h, w = x2d.shape
        flat = F.pad(x2d.reshape(V, C, h * w), (0, 1))                     # extra all-zero pixel
        idx = projected_pix[..., 1] * w + projected_pix[..., 0]            # (V, N, P)
        idx = torch.where(fov_mask, idx, torch.full_like(idx, h * w))
        cnt = fov_mask.sum(-1)                                              # (V, N)
        feats, vis = [], []
        for v in range(V):
            # advanced indexing: its backward sorts the indices (1.8 ms per scale); index_select's atomic index_add_
            # was measured slower here (many voxels hit the same pixel column: +20 ms per step)
            g = flat[v][:, idx[v].reshape(-1)].reshape(C, idx.shape[1], idx.shape[2]).sum(-1)
            seen = cnt[v] > 0
            feats.append(torch.where(seen, g / cnt[v].clamp(min=1), torch.zeros_like(g)))
            vis.append(seen.to(x2d.dtype))
        if V == 1:
            fused = feats[0]
        else:
            fused = 0
            for i in range(V):
                for j in range(i + 1, V):
                    cos = F.cosine_similarity(feats[i], feats[j], 0) * vis[i] * vis[j]
                    fused = fused + (cos + (vis[i] > vis[j]).to(cos.dtype)) * feats[i] \
                        + (cos + (vis[j] > vis[i]).to(cos.dtype)) * feats[j]
            fused = fused / (V * (V - 1))
        n_dims, _, _ = voxel_layout(self.scene_size, self.project_scale, self.dataset)
        x3d = fused.reshape(C, *n_dims)
        return x3d.permute(0, 1, 3, 2) if self.dataset == "NYU" else x3d
