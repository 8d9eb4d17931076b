"""TEST INFRASTRUCTURE ONLY -- CPU restatement (plain PyTorch fp32, functional, state-dict driven)
of the reference's forward hot path.  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this; the product (occdepth_amd/) never does.

Every function cites the reference lines it restates.  Parity of this oracle is PINNED by
tests/golden/*.npz: outputs of the real reference modules (imported from /root/reference through
oracle/ref_shims.py in the build container, generator tests/golden/make_golden.py) that
tests/test_oracle_vs_golden.py reproduces with this file.  Unpinned pieces are exactly the
third-party ones the reference itself does not contain (EfficientNet encoder, kornia helpers,
mmdet BasicBlock) -- see DESIGN.md.

`sd` is a flat dict name -> CPU float tensor with the reference's state_dict keys; `p` a key prefix.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

BN_EPS = 1e-5  # nn.BatchNorm{2,3}d default, used everywhere outside the EfficientNet encoder


# ------------------------------------------------------------------------------------------ primitives
def bn(sd, p, x, eps=BN_EPS):
    """nn.BatchNorm*d in eval mode (running statistics)."""
    return F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"], sd[p + ".weight"], sd[p + ".bias"],
                        False, 0.0, eps)


def conv3d(sd, p, x, stride=1, padding=0, dilation=1):
    return F.conv3d(x, sd[p + ".weight"], sd.get(p + ".bias"), stride=stride, padding=padding, dilation=dilation)


def conv2d(sd, p, x, stride=1, padding=0):
    return F.conv2d(x, sd[p + ".weight"], sd.get(p + ".bias"), stride=stride, padding=padding)


# ------------------------------------------------------------------------------------------ DDR.py
def bottleneck3d(sd, p, x, stride=1, dilation=(1, 1, 1), has_downsample=False):
    """occdepth/models/DDR.py:35-139 (Bottleneck3D.forward :111-139)."""
    s, d = stride, dilation
    out1 = F.relu(bn(sd, p + ".bn1", conv3d(sd, p + ".conv1", x)))
    out2 = bn(sd, p + ".bn2", conv3d(sd, p + ".conv2", out1, (1, 1, s), (0, 0, d[0]), (1, 1, d[0])))
    out3 = bn(sd, p + ".bn3", conv3d(sd, p + ".conv3", F.relu(out2), (1, s, 1), (0, d[1], 0), (1, d[1], 1)))
    if s != 1:  # downsample2 = AvgPool3d((1,s,1)) + 1x1x1 conv + BN   :95-99
        out2 = bn(sd, p + ".downsample2.2", conv3d(sd, p + ".downsample2.1", F.avg_pool3d(out2, (1, s, 1), (1, s, 1))))
    out3 = out3 + out2
    out4 = bn(sd, p + ".bn4", conv3d(sd, p + ".conv4", F.relu(out3), (s, 1, 1), (d[2], 0, 0), (d[2], 1, 1)))
    if s != 1:  # downsample3 / downsample4 = AvgPool3d((s,1,1)) + 1x1x1 conv + BN   :100-109
        out2 = bn(sd, p + ".downsample3.2", conv3d(sd, p + ".downsample3.1", F.avg_pool3d(out2, (s, 1, 1), (s, 1, 1))))
        out3 = bn(sd, p + ".downsample4.2", conv3d(sd, p + ".downsample4.1", F.avg_pool3d(out3, (s, 1, 1), (s, 1, 1))))
    out4 = out4 + out2 + out3
    out5 = bn(sd, p + ".bn5", conv3d(sd, p + ".conv5", F.relu(out4)))
    residual = x
    if has_downsample:  # modules.py:329-339: AvgPool3d(2) + 1x1x1 conv + BN
        residual = bn(sd, p + ".downsample.2", conv3d(sd, p + ".downsample.1", F.avg_pool3d(x, 2, 2)))
    return F.relu(out5 + residual)


# ------------------------------------------------------------------------------------------ modules.py
def process(sd, p, x, dilations=(1, 2, 3)):
    """modules.py:258-275: a Sequential of stride-1 bottlenecks, dilation [i, i, i]."""
    for i, d in enumerate(dilations):
        x = bottleneck3d(sd, f"{p}.main.{i}", x, 1, (d, d, d))
    return x


def downsample(sd, p, x):
    """modules.py:320-344: Bottleneck3D(stride=2, expansion=8) with the AvgPool residual branch."""
    return bottleneck3d(sd, p + ".main", x, 2, (1, 1, 1), has_downsample=True)


def upsample(sd, p, x, stride=2):
    """modules.py:278-317: ConvTranspose3d(k3, s, p1, output_padding=s-1) + BN + ReLU."""
    y = F.conv_transpose3d(x, sd[p + ".main.0.weight"], sd[p + ".main.0.bias"], stride=stride, padding=1,
                           output_padding=stride - 1)
    return F.relu(bn(sd, p + ".main.1", y))


def dilated_branches(sd, p, x, dilations=(1, 2, 3)):
    """modules.py:40-46 / :97-100 / :162-165: sum of conv-BN-ReLU-conv-BN branches, + x, ReLU."""
    y = None
    for i, d in enumerate(dilations):
        t = F.relu(bn(sd, f"{p}.bn1.{i}", conv3d(sd, f"{p}.conv1.{i}", x, 1, d, d)))
        t = bn(sd, f"{p}.bn2.{i}", conv3d(sd, f"{p}.conv2.{i}", t, 1, d, d))
        y = t if y is None else y + t
    return F.relu(y + x)


def seg_head(sd, p, x, dilations=(1, 2, 3), kind="plain"):
    """modules.py:51-235.  kind: plain -> logits; cascade -> (ssc, occ); occluded -> occ."""
    x = F.relu(conv3d(sd, p + ".conv0", x, 1, 1))
    x = dilated_branches(sd, p, x, dilations)
    if kind == "plain":
        return conv3d(sd, p + ".conv_classes", x, 1, 1)
    occ = conv3d(sd, p + ".occ_classes", x, 1, 1)
    if kind == "occluded":
        return occ
    x = torch.cat([x, F.softmax(occ, dim=1)], dim=1)  # :168-171
    return conv3d(sd, p + ".conv_classes", x, 1, 1), occ


# ------------------------------------------------------------------------------------------ CRP3D.py
def crp(sd, p, x, size, n_relations=4):
    """CRP3D.py:54-97."""
    bs = x.shape[0]
    n = size[0] * size[1] * size[2]
    m = (size[0] // 2) * (size[1] // 2) * (size[2] // 2)
    pad = tuple((s + 1) % 2 for s in size)
    x_agg = dilated_branches(sd, p + ".aspp", x)
    mega = conv3d(sd, p + ".mega_context.0", x_agg, 2, pad).reshape(bs, -1, m).permute(0, 2, 1)
    logits, rels = [], []
    for r in range(n_relations):
        lg = conv3d(sd, f"{p}.context_prior_logits.{r}.0", x_agg).reshape(bs, m, n)
        logits.append(lg.unsqueeze(1))
        rels.append(torch.bmm(torch.sigmoid(lg.permute(0, 2, 1)), mega))
    ctx = torch.cat(rels, dim=2).permute(0, 2, 1).reshape(bs, -1, *size)
    y = conv3d(sd, p + ".resize.0", torch.cat([x, ctx], dim=1))
    y = process(sd, p + ".resize.1", y, dilations=(1,))
    return {"P_logits": torch.cat(logits, dim=1), "x": y}


# ------------------------------------------------------------------------------------------ 3-D UNets
def unet3d_kitti(sd, x3d, full_scene_size, project_scale, context_prior=True, cascade_cls=True, occluded_cls=False,
                 infer_mode=False, p="net_3d_decoder"):
    """unet3d_kitti.py:89-126."""
    pre = p + "." if p else ""
    res = {}
    x1 = x3d
    x2 = downsample(sd, pre + "process_l1.1", process(sd, pre + "process_l1.0", x1))
    x3 = downsample(sd, pre + "process_l2.1", process(sd, pre + "process_l2.0", x2))
    if context_prior:
        size_l3 = tuple(int(s / project_scale) // 2 // 2 for s in full_scene_size)
        ret = crp(sd, pre + "CP_mega_voxels", x3, size_l3)
        x3 = ret["x"]
        res.update(ret)
    up2 = upsample(sd, pre + "up_13_l2", x3) + x2
    up1 = upsample(sd, pre + "up_12_l1", up2) + x1
    full = upsample(sd, pre + "up_l1_lfull", up1, stride=1 if project_scale == 1 else 2)
    if not infer_mode:
        res["x3d_l1"], res["x3d_l2"], res["x3d_l3"] = up1, up2, x3
    if cascade_cls:
        ssc, occ = seg_head(sd, pre + "ssc_head", full, kind="cascade")
        res["ssc_logit"] = ssc
        if not infer_mode:
            res["occ_logit"] = occ
    else:
        res["ssc_logit"] = seg_head(sd, pre + "ssc_head", full)
    if occluded_cls:
        occluded = seg_head(sd, pre + "occluded_head", full, kind="occluded")
        if not infer_mode:
            res["occluded_logit"] = occluded
    return res


def unet3d_nyu(sd, x3d, full_scene_size, context_prior=True, cascade_cls=False, n_relations=4, infer_mode=False,
               p="net_3d_decoder"):
    """unet3d_nyu.py:79-110."""
    pre = p + "." if p else ""
    res = {}
    x4 = x3d
    x8 = downsample(sd, pre + "process_1_4.1", process(sd, pre + "process_1_4.0", x4))
    x16 = downsample(sd, pre + "process_1_8.1", process(sd, pre + "process_1_8.0", x8))
    if context_prior:
        size = tuple(int(math.ceil(i / 4)) for i in full_scene_size)
        ret = crp(sd, pre + "CP_mega_voxels", x16, size, n_relations)
        x16 = ret["x"]
        res.update(ret)
    up8 = upsample(sd, pre + "up_1_16_1_8", x16) + x8
    up4 = upsample(sd, pre + "up_1_8_1_4", up8) + x4
    if not infer_mode:
        res["x3d_l1"], res["x3d_l2"], res["x3d_l3"] = up4, up8, x16
    if cascade_cls:
        ssc, occ = seg_head(sd, pre + "ssc_head_1_4", up4, kind="cascade")
        res["ssc_logit"] = ssc
        if not infer_mode:
            res["occ_logit"] = occ
    else:
        res["ssc_logit"] = seg_head(sd, pre + "ssc_head_1_4", up4)
    return res


# ------------------------------------------------------------------------------------------ SFA.py
def sfa(x2d, projected_pix, fov_mask, scene_size, project_scale, dataset):
    """SFA.py:12-106.  x2d (V, C, h, w); projected_pix (V, N, P, 2) int64 already divided by the 2-D
    scale; fov_mask (V, N, P) bool -> (C, X, Y, Z)."""
    V, C, h, w = x2d.shape
    N = projected_pix.shape[1]
    feats, vis = [], []
    for v in range(V):
        src = torch.cat([x2d[v].reshape(C, -1), torch.zeros(C, 1, dtype=x2d.dtype)], 1)   # :18-20 zero column
        idx = projected_pix[v, :, :, 1] * w + projected_pix[v, :, :, 0]                     # :21-22
        idx = torch.where(fov_mask[v], idx, torch.full_like(idx, h * w))                    # :26
        g = src[:, idx[:, 0]]
        for q in range(1, idx.shape[1]):                                                    # :28-30
            g = g + src[:, idx[:, q]]
        cnt = fov_mask[v].sum(1)                                                            # :31
        f = g / cnt                                                                         # :32 (0/0 -> NaN)
        f = torch.where(torch.isnan(f), torch.zeros_like(f), f)                             # :34-38
        m = (cnt / cnt)
        m = torch.where(torch.isnan(m), torch.zeros_like(m), m)                             # :33,39-41
        feats.append(f)
        vis.append(m.to(x2d.dtype))
    if V == 1:
        fused = feats[0]                                                                    # :88-89
    else:
        fused = torch.zeros(C, N, dtype=x2d.dtype)
        for i in range(V):
            for j in range(i + 1, V):
                both = vis[i] * vis[j]
                diff = vis[i] - vis[j]
                cos = torch.cosine_similarity(feats[i], feats[j], 0) * both                 # :66
                fused = fused + ((cos + (diff > 0).to(cos.dtype)) * feats[i]
                                 + (cos + (diff < 0).to(cos.dtype)) * feats[j])             # :67-80
        fused = fused / (V * (V - 1))                                                       # :84-86
    s = [int(v) // int(project_scale) for v in scene_size]
    if dataset == "NYU":                                                                    # :90-97
        return fused.reshape(C, s[0], s[2], s[1]).permute(0, 1, 3, 2)
    if dataset == "kitti":                                                                  # :98-104
        return fused.reshape(C, s[0], s[1], s[2])
    raise NotImplementedError(dataset)


def lift(x_rgb, project_res, projected_pix, fov_mask, scene_size, project_scale, dataset):
    """OccDepth.py:266-298: sum over 2-D scales of SFA(features, pix // scale), stacked over the batch.
    x_rgb: list over views of {"1_s": (B, C, h, w)}; projected_pix / fov_mask: per-sample lists."""
    out = []
    for i in range(len(projected_pix)):
        x3d = None
        for s in project_res:
            s = int(s)
            x2d = torch.stack([x_rgb[v]["1_" + str(s)][i] for v in range(len(x_rgb))])
            part = sfa(x2d, torch.div(projected_pix[i], s, rounding_mode="floor"), fov_mask[i], scene_size,
                       project_scale, dataset)
            x3d = part if x3d is None else x3d + part
        out.append(x3d)
    return torch.stack(out)


# ------------------------------------------------------------------------------------------ flosp_depth
def depth_net(sd, p, x, intrins, scaled_pixel_size=None):
    """flosp_depth.py:230-257 (DepthNet.forward) with mmdet BasicBlock x3 (:218-222)."""
    if scaled_pixel_size is None:
        inv = torch.inverse(intrins)
        size = torch.norm(torch.stack([inv[..., 0, 0], inv[..., 1, 1]], dim=-1), dim=-1).reshape(-1, 1)
        scaled_pixel_size = size * 1000.0
    x = F.relu(bn(sd, p + ".reduce_conv.1", conv2d(sd, p + ".reduce_conv.0", x, 1, 1)))
    se = F.linear(F.relu(F.linear(scaled_pixel_size.to(x.dtype), sd[p + ".mlp.fc1.weight"], sd[p + ".mlp.fc1.bias"])),
                  sd[p + ".mlp.fc2.weight"], sd[p + ".mlp.fc2.bias"])[..., None, None]
    gate = conv2d(sd, p + ".se.conv_expand", F.relu(conv2d(sd, p + ".se.conv_reduce", se)))
    x = x * torch.sigmoid(gate)
    for i in range(3):
        q = f"{p}.depth_conv.{i}"
        y = bn(sd, q + ".bn2", conv2d(sd, q + ".conv2", F.relu(bn(sd, q + ".bn1", conv2d(sd, q + ".conv1", x, 1, 1))),
                                      1, 1))
        x = F.relu(y + x)
    return conv2d(sd, p + ".depth_pred", x)


def _from_homogeneous(pts, eps=1e-8):
    z = pts[..., -1:]
    scale = torch.where(z.abs() > eps, 1.0 / (z + eps), torch.ones_like(z))
    return scale * pts[..., :-1]


def _transform_points(trans, pts):
    """kornia 0.5.0 transform_points on (B, N, 3) points with (B, 4, 4) transforms."""
    pts_h = F.pad(pts, [0, 1], "constant", 1.0)
    return _from_homogeneous(torch.bmm(pts_h, trans.permute(0, 2, 1)))


def frustum_grid(lidar_to_cam, cam_to_img, ida, voxel_num, pc_range, d_bound, num_bins, final_dim):
    """f2v/frustum_grid_generator.py:8-152 -> (B, X, Y, Z, 3) normalised sampling grid."""
    B = lidar_to_cam.shape[0]
    A, Bd, C = (int(v) for v in voxel_num)
    pr = torch.as_tensor(pc_range).reshape(2, 3)
    voxel_size = (pr[1] - pr[0]) / torch.as_tensor(voxel_num)                               # :24-29
    g2l = torch.eye(4, dtype=torch.float32)
    for i in range(3):
        g2l[i, i] = voxel_size[i]
        g2l[i, 3] = pr[0][i]                                                                 # :58-66
    idx = torch.stack(torch.meshgrid(torch.arange(A), torch.arange(Bd), torch.arange(C), indexing="ij"), -1)
    grid = (idx.float() + 0.5).reshape(1, -1, 3).repeat(B, 1, 1)                            # :32-42 voxel centres
    cam = _transform_points(lidar_to_cam @ g2l, grid)                                       # :89-96
    pts_t = (cam_to_img.reshape(B, 1, 3, 4) @ F.pad(cam, [0, 1], "constant", 1.0).unsqueeze(-1)).squeeze(-1)
    uv = _from_homogeneous(pts_t)                                                           # transform_utils.py:16-23
    depth = pts_t[..., -1] - cam_to_img[:, None, 2, 3]                                      # transform_utils.py:24
    bin_size = 2 * (d_bound[1] - d_bound[0]) / (num_bins * (1 + num_bins))                  # depth_utils.py:24-26
    bins = -0.5 + 0.5 * torch.sqrt(1 + 8 * (depth - d_bound[0]) / bin_size)
    fr = _transform_points(ida, torch.cat((uv, bins.unsqueeze(-1)), dim=-1))                # :113-114
    shape = torch.tensor([final_dim[1], final_dim[0], num_bins], dtype=torch.float32)       # :137-146 [W, H, D]
    fr = fr / (shape - 1) * 2 + -1                                                           # grid_utils.py:15-18
    fr[~torch.isfinite(fr)] = -2                                                            # :149-150
    return fr.reshape(B, A, Bd, C, 3)


def flosp_depth(sd, p, img_feat, cam_k, T_velo_2_cam, ida_mats, conf, voxel_num, pc_range, agg="mean"):
    """flosp_depth.py:456-608 (not infer_mode).  Returns (voxel volume (B,1,X,Y,Z), depth (B,V,D,h,w))."""
    bs, n_cams, c, h, w = img_feat.shape
    ida = torch.stack(ida_mats).float()
    t_v2c = torch.stack(T_velo_2_cam).to(torch.float32)
    k3 = torch.stack(cam_k).to(torch.float32)
    intr = k3.new_zeros(bs, n_cams, 4, 4)
    intr[:, :, :3, :3] = k3
    intr[:, :, 3, 3] = 1
    d_bound = conf["d_bound"]
    nb = int((d_bound[1] - d_bound[0]) / d_bound[2])
    logits = depth_net(sd, p + ".depth_net.0", img_feat.reshape(bs * n_cams, c, h, w), intr)
    depth = logits.softmax(1).reshape(bs, n_cams, nb, h, w)
    feats, masks = [], []
    for i in range(n_cams):
        grid = frustum_grid(t_v2c[:, i], intr[:, i, :3, :], ida[:, i], voxel_num, pc_range, d_bound, nb,
                            conf["final_dim"])
        vol = depth[:, i].unsqueeze(1)
        grid = grid.to(vol.dtype)                  # no-op in float32; lets tests run the NETWORK arithmetic in float64
        feats.append(F.grid_sample(vol, grid, mode="bilinear", padding_mode="zeros", align_corners=False))
        masks.append(F.grid_sample(torch.ones_like(vol), grid, mode="bilinear", padding_mode="zeros",
                                   align_corners=False))
    if n_cams == 1:
        out = feats[0]
    else:
        out = sum(feats)
        if agg == "mean":
            msum = sum(masks)
            out[msum > 0] = out[msum > 0] / msum[msum > 0]
    return out, depth


# ------------------------------------------------------------------------------------------ unet2d.py
def decoder_bn(sd, p, features, skip_taps=(8, 6, 5, 4, 0), return_up_feats=1):
    """unet2d.py:137-165 with UpSampleBN :38-46 (LeakyReLU slope 0.01, bilinear align_corners=True)."""
    x = conv2d(sd, p + ".conv2", features[11], 1, 1)  # 1x1 conv with padding=1 (sic)   :65-67,146
    res = {}
    for s, tap in zip((16, 8, 4, 2, 1), skip_taps):
        if return_up_feats > s:
            continue
        skip = features[tap]
        up = F.interpolate(x, size=skip.shape[2:], mode="bilinear", align_corners=True)
        x = torch.cat([up, skip], dim=1)
        q = f"{p}.up{s}._net"
        x = F.leaky_relu(bn(sd, q + ".1", conv2d(sd, q + ".0", x, 1, 1)))
        x = F.leaky_relu(bn(sd, q + ".4", conv2d(sd, q + ".3", x, 1, 1)))
        res[f"1_{s}"] = conv2d(sd, f"{p}.resize_output_1_{s}", x)
    return res


def encoder_features(backend, x):
    """unet2d.py:188-196: walk the backbone's _modules, blocks stage by stage, keep every output."""
    feats = [x]
    for name, mod in backend._modules.items():
        if name == "blocks":
            for stage in mod._modules.values():
                feats.append(stage(feats[-1]))
        else:
            feats.append(mod(feats[-1]))
    return feats


def generate_virtual_img(x, gt_depth, scale_2d, bf):
    """OccDepth.py:233-260 (uses sample 0's disparity for every batch item, :257)."""
    n, c, h, w = x.shape
    depth = F.interpolate(gt_depth, size=(h, w), mode="bilinear", align_corners=False)
    dx = torch.div(bf / int(scale_2d), depth).type_as(x)
    dx = torch.where(torch.isinf(dx), torch.zeros_like(dx), dx)
    hd = torch.arange(-1, 1, 2 / h)
    wd = torch.arange(-1, 1, 2 / w)
    my, mx = torch.meshgrid(hd, wd, indexing="ij")
    grid = torch.stack([torch.stack((mx, my), dim=2)] * n).type_as(dx)
    dx = dx * 2 / w
    grid[:, :, :, 0] = grid[:, :, :, 0] + dx[0, ...]
    return F.grid_sample(x, grid, mode="bilinear", padding_mode="border", align_corners=False)


# ------------------------------------------------------------------------------------------ OccDepth.py
def occdepth_forward(sd, cfg, batch, encoder, project_res=("1", "2", "4", "8"), infer_mode=False):
    """OccDepth.py:344-376 on CPU.  `encoder` is the (third-party, unpinned) EfficientNet module whose
    weights live in sd under net_rgb.encoder.original_model.*; everything downstream is restated here."""
    img = batch["img"]
    bs, n_views = img.shape[:2]
    with_depth_gt = cfg["use_stereo_depth_gt"] or cfg["use_lidar_depth_gt"] or cfg["use_depth_gt"]
    x_rgb = []
    for v in range(n_views):
        feats = encoder_features(encoder, img[:, v])
        x_rgb.append(decoder_bn(sd, "net_rgb.decoder", feats, return_up_feats=cfg["return_up_feats"]))
    if n_views == 1 and "gt_depth" in batch:                                               # :222-229
        bf = batch["virtual_bf"][0]
        x_rgb.append({"1_" + s: generate_virtual_img(x_rgb[0]["1_" + s], batch["gt_depth"], s, bf)
                      for s in project_res})
    ps = cfg["project_scale"]
    scene = cfg["full_scene_size"]
    x3ds = lift(x_rgb, project_res, batch[f"projected_pix_{ps}"], batch[f"fov_mask_{ps}"], scene, ps, cfg["dataset"])
    depth_pred = None
    if cfg["trans_2d_to_3d"] == "flosp_depth":                                            # :299-339
        conf = cfg["flosp_depth_conf"]
        nv = 1 if cfg["dataset"] == "NYU" else len(x_rgb)
        img_feat = torch.stack([x_rgb[j]["1_%d" % conf["downsample_factor"]] for j in range(nv)], 1)
        bounds = [conf["x_bound"], conf["y_bound"], conf["z_bound"]]
        if cfg["dataset"] == "NYU":                                                         # flosp_depth.py:466-518
            vo = batch["vox_origin"]
            bounds = [[float(vo[0][i]), float(vo[0][i]) + e, 0.08] for i, e in enumerate((4.8, 4.8, 2.88))]
        voxel_num = [int(v) for v in torch.LongTensor([(r[1] - r[0]) / r[2] / ps for r in bounds])]
        pc_range = [r[0] for r in bounds] + [r[1] for r in bounds]
        vol, depth_pred = flosp_depth(sd, "flosp_depth", img_feat, batch["cam_k"], batch["T_velo_2_cam"],
                                      batch["ida_mats"], conf, voxel_num, pc_range, conf.get("agg_voxel_mode", "mean"))
        if cfg["dataset"] == "NYU":
            vol = vol.permute(0, 1, 2, 4, 3).contiguous()
        x3ds = x3ds * vol * 100
    ctx = cfg["context_prior"] and not infer_mode
    if cfg["dataset"] == "kitti":
        out = unet3d_kitti(sd, x3ds, scene, ps, ctx, cfg["cascade_cls"], cfg["occluded_cls"], infer_mode)
    else:
        out = unet3d_nyu(sd, x3ds, scene, ctx, cfg["cascade_cls"], cfg["n_relations"], infer_mode)
    if with_depth_gt and cfg["trans_2d_to_3d"] == "flosp_depth":
        out["depth_pred"] = depth_pred
    return out
