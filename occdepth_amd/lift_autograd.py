"""Differentiable 2D->3D lift on the HIP kernels (the training step).

In the reference, autograd walks B x 4 `SFA.forward` calls (gather, masked mean, cosine-similarity fusion: ~160 small
ATen kernels and ~10 (C, N) temporaries per scale) and the `x3ds * depth * 100` of occdepth/models/OccDepth.py:266-298,
339; its backward scatters with a sorted `index_put` (14.5 ms per step at config 2).  Here the forward is the same fused
kernel as in eval (K1b, `occd_lift_fwd`) and the backward is ONE launch (`occd_lift_bwd`) that recomputes the gathers and
the fusion weights and scatters the feature gradients with hardware float atomics.
Single-point patterns only (`pattern_id` 0, every shipped config); other patterns keep the ATen path.
"""
import torch

from . import hip
from .hip import Vox
from .models.SFA import pixel_rows, voxel_layout


class _LiftFn(torch.autograd.Function):
    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, pix, fov, depth_scale, meta, *feats_flat):
        scale_divs, n_views, scene_size, project_scale, dataset, scale_const = meta
        S = len(scale_divs)
        n_dims, out_dims, strides = voxel_layout(scene_size, project_scale, dataset)
        rows = [[pixel_rows(feats_flat[s * n_views + v].detach().float()) for v in range(n_views)] for s in range(S)]
        B, C = feats_flat[0].shape[0], feats_flat[0].shape[1]
        out = Vox.empty(B, out_dims, C, feats_flat[0].device)
        if out.cs != C:
            out.buf.zero_()
        ds = depth_scale.detach().float().reshape(B, -1).contiguous() if depth_scale is not None else None
        hip.lift(rows, scale_divs, pix, fov, n_dims, strides, out, depth_scale=ds, scale_const=scale_const)
        ctx.save_for_backward(pix, fov, ds, *[r for per in rows for r in per])
        ctx.meta = (scale_divs, n_views, n_dims, out_dims, strides, C, scale_const, depth_scale is not None,
                    [tuple(f.shape) for f in feats_flat], depth_scale.shape if depth_scale is not None else None)
        return out.ncdhw()

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, gy):
        scale_divs, n_views, n_dims, out_dims, strides, C, scale_const, has_depth, shapes, dshape = ctx.meta
        pix, fov, ds, *flat = ctx.saved_tensors
        S = len(scale_divs)
        rows = [[flat[s * n_views + v] for v in range(n_views)] for s in range(S)]
        B = gy.shape[0]
        g = gy.float().permute(0, 2, 3, 4, 1)
        cs = hip.round_up(C, 4)
        if not (g.is_contiguous() and cs == C):
            gp = torch.zeros((B,) + tuple(out_dims) + (cs,), device=gy.device, dtype=torch.float32)
            gp[..., :C] = g
            g = gp
        like = Vox(g, C)                         # (only its geometry is read)
        grads, gd = hip.lift_backward(rows, scale_divs, pix, fov, n_dims, strides, like, g, depth_scale=ds if has_depth else None,
                                      scale_const=scale_const)
        outs = []
        for s in range(S):
            for v in range(n_views):
                c = shapes[s * n_views + v][1]
                outs.append(grads[s][v][..., :c].permute(0, 3, 1, 2))      # logical (B, C, h, w), channels-last memory
        gdepth = gd.reshape(dshape) if has_depth else None
        return (None, None, gdepth, None) + tuple(outs)


def lift_scales_autograd(feats, scale_divs, projected_pix, fov_mask, scene_size, project_scale, dataset, depth_scale=None,
                         scale_const=100.0):
    """Differentiable twin of models.SFA.lift_scales: feats[s][v] (B, C, h_s, w_s) -> (B, C, X, Y, Z) tensor
    (channels-last memory), gradients to every feature map and to `depth_scale` ((B, 1, X, Y, Z) or (B, N))."""
    n_views = len(feats[0])
    meta = (tuple(int(d) for d in scale_divs), n_views, tuple(scene_size), project_scale, dataset, float(scale_const))
    flat = [f for per in feats for f in per]
    return _LiftFn.apply(projected_pix.contiguous(), fov_mask.contiguous(), depth_scale, meta, *flat)


def usable(feats, projected_pix):
    f = feats[0][0]
    return f.is_cuda and projected_pix.shape[3] == 1 and f.shape[1] % 4 == 0 and f.shape[1] <= 256
