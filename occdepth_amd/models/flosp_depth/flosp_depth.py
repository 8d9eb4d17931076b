"""FLoSP-Depth / occupancy-aware depth volume (mirror of the live part of
occdepth/models/flosp_depth/flosp_depth.py:160-257,324-608).

  DepthNet  : 3x3 conv+BN+ReLU -> camera-aware SE (MLP of the scaled pixel size) ->
              3 residual BasicBlocks -> 1x1 conv to D depth-bin logits        (torch / MIOpen)
  FlospDepth: softmax over the D bins, then for every voxel and camera the frustum
              coordinate (u, v, LID bin) is computed ON THE FLY inside the HIP kernel K1a
              (`occd_flosp_sample_fwd`) and the (D, h, w) probability frustum is sampled
              trilinearly; no (B, X, Y, Z, 3) grid tensor and no all-ones mask volume are
              ever materialised.

BasicBlock restates mmdet 2.20's resnet BasicBlock (conv1, bn1, conv2, bn2, identity skip);
mmdet itself is not a dependency.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from ... import hip
import os

from ... import fused as _fused
from ...fused import bn_affine_cached, needs_autograd, wino_fused_operands

# DepthNet's 3x3 convolutions on K10 (fused Winograd MFMA kernel, BatchNorm / ReLU / identity skip in its epilogue) instead
# of MIOpen + a BatchNorm pass, and its camera-aware gate (Mlp -> SELayer) as ONE launch (occd_depthnet_gate) instead of ~20
# ATen / rocBLAS ones.  Default since round 5 (no MIOpen / Cijk_* kernel left in the eval trace): a K10
# launch on 128 channels of a 47x153 map is 120 workgroups -- 62 us against MIOpen's 46 + 6 + a BatchNorm pass, which the
# ~25 launches that disappear pay back.  OCCDEPTH_DEPTHNET_K10=0 restores the library path for A/B.
DEPTHNET_K10 = os.environ.get("OCCDEPTH_DEPTHNET_K10", "1") == "1"


def _conv3x3_train(conv, x):
    """Training on the GPU: DepthNet's 3x3 / pad 1 convolutions forward and data gradient on K10 (hip.conv2d_3x3_autograd), as
    the decoder's; anything else (CPU, half precision, OCCDEPTH_DEPTHNET_K10=0) is the module itself."""
    if DEPTHNET_K10 and x.is_cuda and x.dtype == torch.float32 and conv.kernel_size == (3, 3) and conv.stride == (1, 1) and \
            conv.padding == (1, 1) and conv.dilation == (1, 1) and conv.groups == 1:
        return hip.conv2d_3x3_autograd(x, conv.weight, conv.bias)
    return conv(x)


class BasicBlock(nn.Module):
    def __init__(self, inplanes, planes):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 3, stride=1, padding=1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)

    def forward(self, x):
        if _fused.on_gpu(x) and not needs_autograd(self) and x.dtype == torch.float32:
            if DEPTHNET_K10:                                   # 2 launches per block
                upk, shift = wino_fused_operands(self, self.conv1, self.bn1)
                y = hip.conv2d_3x3_fused(x, upk, self.conv1.out_channels, shift, "relu")
                upk, shift = wino_fused_operands(self, self.conv2, self.bn2)
                return hip.conv2d_3x3_fused(y, upk, self.conv2.out_channels, shift, "relu", res=x, res_first=True)
            y = hip.affine_act(self.conv1(x), *bn_affine_cached(self.bn1), "relu")
            return hip.affine_act(self.conv2(y), *bn_affine_cached(self.bn2), "relu", res=x, res_first=True)
        out = self.bn2(_conv3x3_train(self.conv2, F.relu(self.bn1(_conv3x3_train(self.conv1, x)))))
        return F.relu(out + x)


class Mlp(nn.Module):
    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.ReLU, drop=0.0):
        super().__init__()
        out_features = out_features or in_features
        hidden_features = hidden_features or in_features
        self.fc1 = nn.Linear(in_features, hidden_features)
        self.act = act_layer()
        self.drop1 = nn.Dropout(drop)
        self.fc2 = nn.Linear(hidden_features, out_features)
        self.drop2 = nn.Dropout(drop)

    def forward(self, x):
        return self.drop2(self.fc2(self.drop1(self.act(self.fc1(x)))))


class SELayer(nn.Module):
    def __init__(self, channels, act_layer=nn.ReLU, gate_layer=nn.Sigmoid):
        super().__init__()
        self.conv_reduce = nn.Conv2d(channels, channels, 1, bias=True)
        self.act1 = act_layer()
        self.conv_expand = nn.Conv2d(channels, channels, 1, bias=True)
        self.gate = gate_layer()

    def forward(self, x, x_se):
        return x * self.gate(self.conv_expand(self.act1(self.conv_reduce(x_se))))


class DepthNet(nn.Module):
    def __init__(self, in_channels, mid_channels, context_channels, depth_channels, infer_mode=False):
        super().__init__()
        self.reduce_conv = nn.Sequential(
            nn.Conv2d(in_channels, mid_channels, kernel_size=3, stride=1, padding=1),
            nn.BatchNorm2d(mid_channels), nn.ReLU(inplace=True))
        self.mlp = Mlp(1, mid_channels, mid_channels)
        self.se = SELayer(mid_channels)
        self.depth_conv = nn.Sequential(*[BasicBlock(mid_channels, mid_channels) for _ in range(3)])
        self.depth_pred = nn.Conv2d(mid_channels, depth_channels, kernel_size=1, stride=1, padding=0)
        self.infer_mode = infer_mode

    @staticmethod
    def scaled_pixel_size(sweep_intrins, scale_depth_factor=1000.0, sync_free=False):
        if sync_free:
            # pinhole intrinsics are upper triangular, so diag(K^-1) = 1 / diag(K) exactly (what LU computes too);
            # torch.inverse on the GPU goes through rocSOLVER and synchronises the host, which drains the launch
            # queue in the middle of the forward pass
            d0, d1 = 1.0 / sweep_intrins[..., 0, 0], 1.0 / sweep_intrins[..., 1, 1]
        else:
            inv = torch.inverse(sweep_intrins)
            d0, d1 = inv[..., 0, 0], inv[..., 1, 1]
        size = torch.norm(torch.stack([d0, d1], dim=-1), dim=-1).reshape(-1, 1)
        return size * scale_depth_factor

    def forward(self, x=None, sweep_intrins=None, scaled_pixel_size=None, scale_depth_factor=1000.0):
        if DEPTHNET_K10 and _fused.on_gpu(x) and not needs_autograd(self) and x.dtype == torch.float32:
            upk, shift = wino_fused_operands(self, self.reduce_conv[0], self.reduce_conv[1])
            x = hip.conv2d_3x3_fused(x, upk, self.reduce_conv[0].out_channels, shift, "relu")
            n_img, c_mid = x.shape[:2]
            if self.infer_mode:
                gate = hip.depthnet_gate(self.mlp, self.se, n_img, sps=scaled_pixel_size)
            else:       # (the scaled pixel size is derived from the intrinsics inside the launch)
                gate = hip.depthnet_gate(self.mlp, self.se, n_img, intrins=sweep_intrins, factor=scale_depth_factor)
            x = hip.affine_act(x.view(1, n_img * c_mid, *x.shape[2:]), gate.view(-1), None).view(n_img, c_mid, *x.shape[2:])
        else:
            if not self.infer_mode:
                scaled_pixel_size = self.scaled_pixel_size(sweep_intrins, scale_depth_factor,
                                                           sync_free=sweep_intrins.is_cuda)
            x = self.reduce_conv[2](self.reduce_conv[1](_conv3x3_train(self.reduce_conv[0], x)))
            x = self.se(x, self.mlp(scaled_pixel_size.to(self.mlp.fc1.weight.dtype))[..., None, None])
        x = self.depth_conv(x)
        if _fused.on_gpu(x) and not needs_autograd(self) and x.dtype == torch.float32:
            from ..efficientnet import pw_operands, pw_wins
            if pw_wins(x):                                    # 1x1 convolution + bias on the MFMA GEMM (K11)
                wpk, shift = pw_operands(self, self.depth_pred)
                return hip.conv1x1(x, wpk, self.depth_pred.out_channels, shift)
        return self.depth_pred(x)


def _bounds_buffers(x_bound, y_bound, z_bound, project_scale):
    rows = [x_bound, y_bound, z_bound]
    size = torch.Tensor([r[2] * project_scale for r in rows])
    coord = torch.Tensor([r[0] + r[2] / 2.0 * project_scale for r in rows])
    num = torch.LongTensor([(r[1] - r[0]) / r[2] / project_scale for r in rows])
    return size, coord, num


def _grid_to_lidar(pc_range, grid_size):
    """4x4 voxel-index -> lidar matrix, built with the same fp32 tensor arithmetic as
    f2v/frustum_grid_generator.py:24-29,58-66."""
    pr = torch.as_tensor(pc_range).reshape(2, 3)
    pc_min, pc_max = pr[0], pr[1]
    vs = (pc_max - pc_min) / torch.as_tensor(grid_size)
    m = torch.eye(4, dtype=torch.float32)
    for i in range(3):
        m[i, i] = vs[i]
        m[i, 3] = pc_min[i]
    return m


class _FrustumSampleFn(torch.autograd.Function):
    """Training: out = frustum sample of the depth volume (K1a), d depth = its transpose (occd_flosp_sample_bwd).  The
    geometry (calibration matrices / grids) carries no gradient, as in the reference (batch inputs)."""

    @staticmethod
    def forward(ctx, dvol, fr):
        ctx.fr = fr
        return fr.sample()

    @staticmethod
    def backward(ctx, gout):
        return hip.flosp_sample_bwd(ctx.fr, gout.contiguous().float()), None


class FlospDepth(nn.Module):
    def __init__(self, x_bound, y_bound, z_bound, d_bound, final_dim, downsample_factor, output_channels,
                 depth_net_conf, scene_size, project_scale, return_depth, agg_voxel_mode="mean",
                 infer_mode=False, **kwargs):
        super().__init__()
        self.downsample_factor = downsample_factor
        self.d_bound = d_bound
        self.final_dim = final_dim
        self.output_channels = output_channels
        self.depth_channels = int((d_bound[1] - d_bound[0]) / d_bound[2])
        self.scene_size = scene_size
        self.project_scale = project_scale
        self.infer_mode = infer_mode
        self.depth_net_conf = depth_net_conf
        self.depth_net = nn.Sequential(DepthNet(depth_net_conf["in_channels"], depth_net_conf["mid_channels"],
                                                output_channels, self.depth_channels, infer_mode=infer_mode))
        size, coord, num = _bounds_buffers(x_bound, y_bound, z_bound, project_scale)
        self.register_buffer("voxel_size", size)
        self.register_buffer("voxel_coord", coord)
        self.register_buffer("voxel_num", num)
        self.use_quickcumsum = True
        self.return_depth = return_depth
        self.disc_cfg = {"mode": "LID", "num_bins": self.depth_channels, "depth_min": d_bound[0],
                         "depth_max": d_bound[1]}
        self.agg_voxel_mode = agg_voxel_mode
        self._pc_range = [x_bound[0], y_bound[0], z_bound[0], x_bound[1], y_bound[1], z_bound[1]]
        self._grid_dims = tuple(int(v) for v in num)

    def _rebound_nyu(self, vox_origin):
        # NYU scenes move their voxel origin per frame (reference :466-518)
        ext = (4.8, 4.8, 2.88)
        o = [float(vox_origin[0][i]) for i in range(3)]
        bounds = [[o[i], o[i] + ext[i], 0.08] for i in range(3)]
        size, coord, num = _bounds_buffers(*bounds, self.project_scale)
        dev = self.voxel_size.device
        self.voxel_size, self.voxel_coord, self.voxel_num = size.to(dev), coord.to(dev), num.to(dev)
        self._pc_range = [b[0] for b in bounds] + [b[1] for b in bounds]
        self._grid_dims = tuple(int(v) for v in num)

    def forward(self, img_feat, cam_k=None, T_velo_2_cam=None, ida_mats=None, vox_origin=None, grids=None,
                scaled_pixel_size=None, defer_sample=False):
        """defer_sample (eval on the GPU): return a `hip.Frustum` (the operands of the frustum sample) in place of the
        sampled volume -- the fused lift (hip.lift_proj) samples it per voxel itself."""
        if vox_origin is not None and not self.infer_mode:
            self._rebound_nyu(vox_origin)
        bs, n_cams, c, h, w = img_feat.shape
        feat = img_feat.reshape(bs * n_cams, c, h, w)
        if self.agg_voxel_mode not in ("mean", "sum"):
            raise NotImplementedError("agg_voxel_mode: {}".format(self.agg_voxel_mode))
        if self.infer_mode:
            logits = self.depth_net[0](x=feat, sweep_intrins=None, scaled_pixel_size=scaled_pixel_size)
        else:
            ida = torch.stack(ida_mats).to(torch.float32)
            t_v2c = torch.stack(T_velo_2_cam).to(torch.float32)
            k3 = torch.stack(cam_k).to(torch.float32)
            intrins = k3.new_zeros(bs, n_cams, 4, 4)
            intrins[:, :, :3, :3] = k3
            intrins[:, :, 3, 3] = 1
            logits = self.depth_net[0](x=feat, sweep_intrins=intrins, scaled_pixel_size=None)
        if _fused.on_gpu(logits) and not needs_autograd(self) and logits.dtype == torch.float32:
            depth = hip.softmax_nchw(logits).reshape(bs, n_cams, self.depth_channels, h, w)
        else:
            depth = logits.softmax(1).reshape(bs, n_cams, self.depth_channels, h, w)

        if needs_autograd(self) and hip.LOSS_KERNELS and _fused.on_gpu(depth) and depth.dtype == torch.float32:
            # training on the GPU (round 5): the frustum sample and its transpose are the HIP kernels (K1a and
            # occd_flosp_sample_bwd: deterministic) instead of F.grid_sample x 2 per camera and its atomics-based backward
            fr = self._frustum(depth.contiguous(), None if self.infer_mode else (t_v2c, intrins, ida), grids)
            vox = _FrustumSampleFn.apply(fr.depth, fr).view(bs, 1, *self._grid_dims)
        elif needs_autograd(self):
            vox = self._sample_autograd(depth, None if self.infer_mode else (t_v2c, intrins, ida), grids)
        else:
            fr = self._frustum(depth.float().contiguous(), None if self.infer_mode else (t_v2c, intrins, ida), grids)
            if defer_sample:
                return (fr, depth) if self.return_depth else fr
            vox = fr.sample().view(bs, 1, *self._grid_dims)
        if self.return_depth:
            return vox, depth
        return vox

    def _frustum(self, dvol, mats, grids):
        """The operands of the frustum sample as a `hip.Frustum` (mats = (T_velo_2_cam, intrinsics 4x4, ida), or the
        precomputed grids in infer_mode)."""
        if mats is None:
            g = torch.stack(list(grids)).float().contiguous()          # (n_cams, B, X, Y, Z, 3)
            return hip.Frustum(dvol, None, None, None, self._grid_dims, self.final_dim, self.d_bound[0],
                               self.d_bound[1], self.agg_voxel_mode == "mean", grids=g)
        t_v2c, intrins, ida = mats
        gkey = (tuple(self._pc_range), self._grid_dims, t_v2c.device)
        if getattr(self, "_g2l_key", None) != gkey:     # cached on the device: no per-frame H2D copy
            self._g2l_dev = _grid_to_lidar(self._pc_range, self._grid_dims).to(t_v2c.device)
            self._g2l_key = gkey
        # (B, V, 4, 4) @ (4, 4) as a broadcast multiply + sum: the last hipBLASLt launch of the eval frame was this 4 x 4 product
        trans = (t_v2c.unsqueeze(-1) * self._g2l_dev).sum(-2).contiguous()
        proj = intrins[:, :, :3, :].contiguous()
        return hip.Frustum(dvol, trans, proj, ida.contiguous(), self._grid_dims, self.final_dim,
                           self.d_bound[0], self.d_bound[1], self.agg_voxel_mode == "mean")

    # ---------------------------------------------------------------- ATen (training / autograd)
    def _device_const(self, key, dev, make):
        """Small host-built constants cached per device: an H2D copy per step is a host sync (and breaks hipGraph capture
        of the training step)."""
        cache = self.__dict__.setdefault("_dev_consts", {})
        k = (key, str(dev))
        if k not in cache:
            cache[k] = make().to(dev)
        return cache[k]

    def frustum_grid(self, t_v2c_cam, intrins_cam, ida_cam):
        """Normalised (B, X, Y, Z, 3) sampling grid of one camera, torch ops only."""
        A, Bd, C = self._grid_dims
        dev = t_v2c_cam.device
        ii = torch.stack(torch.meshgrid(torch.arange(A, device=dev), torch.arange(Bd, device=dev),
                                        torch.arange(C, device=dev), indexing="ij"), -1).float() + 0.5
        pts = torch.cat([ii, torch.ones_like(ii[..., :1])], -1).reshape(1, -1, 4)

        def dehomog(p):
            wv = p[..., -1:]
            scale = torch.where(wv.abs() > 1e-8, 1.0 / (wv + 1e-8), torch.ones_like(wv))
            return p[..., :-1] * scale

        trans = t_v2c_cam @ self._device_const("g2l", dev, lambda: _grid_to_lidar(self._pc_range, self._grid_dims))
        cam = dehomog(pts @ trans.transpose(1, 2))
        img = torch.cat([cam, torch.ones_like(cam[..., :1])], -1) @ intrins_cam[:, :3, :].transpose(1, 2)
        uv = dehomog(img)
        dep = img[..., 2] - intrins_cam[:, None, 2, 3]
        d0, d1, nb = self.d_bound[0], self.d_bound[1], self.depth_channels
        bin_size = 2 * (d1 - d0) / (nb * (1 + nb))
        idx = -0.5 + 0.5 * torch.sqrt(1 + 8 * (dep - d0) / bin_size)
        fr = torch.cat([uv, idx.unsqueeze(-1), torch.ones_like(idx).unsqueeze(-1)], -1)
        fr = dehomog(fr @ ida_cam.transpose(1, 2))
        shape = self._device_const(("shape", nb), dev, lambda: torch.tensor([self.final_dim[1], self.final_dim[0], nb],
                                                                             dtype=torch.float32))
        fr = fr / (shape - 1) * 2 - 1
        fr = torch.where(torch.isfinite(fr), fr, torch.full_like(fr, -2.0))
        return fr.reshape(-1, A, Bd, C, 3)

    def _sample_autograd(self, depth, mats, grids):
        bs, n_cams = depth.shape[:2]
        feats, masks = [], []
        for i in range(n_cams):
            grid = grids[i] if mats is None else self.frustum_grid(mats[0][:, i], mats[1][:, i], mats[2][:, i])
            vol = depth[:, i].unsqueeze(1)
            grid = grid.to(vol.dtype)              # geometry is float32 by definition; the volume may be bf16 / float64
            feats.append(F.grid_sample(vol, grid, mode="bilinear", padding_mode="zeros", align_corners=False))
            masks.append(F.grid_sample(torch.ones_like(vol), grid, mode="bilinear", padding_mode="zeros",
                                       align_corners=False))
        if n_cams == 1:
            return feats[0]
        total = sum(feats)
        if self.agg_voxel_mode == "mean":
            m = sum(masks)
            total = torch.where(m > 0, total / m.clamp(min=1e-30), total)
        return total
