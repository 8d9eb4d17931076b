"""TEST-ONLY CPU emulation of the C-ABI entry points (include/occdepth_amd.h) in plain torch.

`patched()` swaps the ctypes wrappers in occdepth_amd.hip for these functions so the host-side
orchestration of the eval path (plans, BN folding, transposed-conv phase slicing, concat-slice
bookkeeping, layout digits) can be executed and checked against the oracle without a GPU.
It is not importable from the product and is never used by the `-m gpu` tests.
"""
import contextlib

import torch
import torch.nn.functional as F

from occdepth_amd import fused, hip
from occdepth_amd.hip import ACT_RELU, ACT_RELU_PRE, ACT_SIGMOID, Vox, round_up


class _Packed:
    def __init__(self, w):
        self.w = w

    @property
    def device(self):
        return self.w.device


def pack_weights(w, scale=None, layout=0):
    w = w.detach().float()
    if layout == 2:
        w = w.t().reshape(w.shape[1], w.shape[0], 1, 1, 1)
    if scale is not None:
        w = w * scale.reshape(-1, 1, 1, 1, 1)
    return _Packed(w.contiguous())


def conv3d(x, wpk, bias, cout, kernel, out, stride=(1, 1, 1), dilation=(1, 1, 1), padding=(0, 0, 0), res1=None,
           res2=None, act_in=0, act_out=0, out_pos=None, o_stride=(1, 1, 1), o_off=(0, 0, 0), cin=None,
           tile_hint=0):
    cin = x.C if cin is None else cin
    assert x.coff % 4 == 0 and x.cs % 4 == 0 and x.coff + round_up(cin, 8) <= x.cs, "input row alignment"
    xin = x.buf[..., x.coff:x.coff + cin].permute(0, 4, 1, 2, 3)
    # the channel pad the kernel would read must be zero (or be multiplied by zero weights and finite)
    pad_ch = x.buf[..., x.coff + cin:x.coff + round_up(cin, 8)]
    assert torch.isfinite(pad_ch).all(), "non-finite values in a channel pad"
    if act_in == ACT_RELU:
        xin = F.relu(xin)
    elif act_in == ACT_SIGMOID:
        xin = torch.sigmoid(xin)
    if out_pos is None:
        out_pos = tuple((n + 2 * p - d * (k - 1) - 1) // s + 1
                        for n, p, d, k, s in zip(x.dims, padding, dilation, kernel, stride))
    pads = []
    for n, p, d, k, s, o in reversed(list(zip(x.dims, padding, dilation, kernel, stride, out_pos))):
        right = max(0, (o - 1) * s - p + (k - 1) * d - (n - 1))
        pads += [p, right]
    y = F.conv3d(F.pad(xin, pads), wpk.w[:, :cin], None, stride=stride, dilation=dilation)
    y = y[:, :, :out_pos[0], :out_pos[1], :out_pos[2]]
    if bias is not None:
        y = y + bias[:cout].reshape(1, -1, 1, 1, 1)
    if act_out == ACT_RELU_PRE:
        y = F.relu(y)
    store = min(round_up(cout, 8), out.cs - out.coff, round_up(cout, 32))
    sl = (slice(None),) + tuple(slice(off, off + st * (n - 1) + 1, st) for off, st, n in zip(o_off, o_stride, out_pos))
    yl = y.permute(0, 2, 3, 4, 1)
    full = torch.zeros(yl.shape[:-1] + (store,))
    full[..., :cout] = yl
    for r in (res1, res2):
        if r is not None:
            assert r.dims == out.dims
            full = full + r.buf[sl][..., r.coff:r.coff + store]
    if act_out == ACT_RELU:
        full = F.relu(full)
    out.buf[sl + (slice(out.coff, out.coff + store),)] = full
    return out


def _bf16(t):
    return t.to(torch.bfloat16).to(t.dtype)


def pack_weights_bf16(w, scale=None, layout=0, split3=False):
    """K2b's weight image: the float32 master weights rounded to bf16 (kept as a float32 tensor of bf16 values); the
    3-way split keeps float32 (hi + mid + lo carries 24 bits)."""
    pk = pack_weights(w, scale, layout)
    if not split3:
        pk.w = _bf16(pk.w)
    pk.bf16 = True
    pk.split3 = split3
    return pk


def conv3d_phases(x, phases, bias, cout, out, res1=None, act_out=0, out_pos=None, o_stride=(1, 1, 1), split3=False):
    for wpk, kernel, o_off in phases:
        conv3d(x, wpk, bias, cout, kernel, out, res1=res1, act_out=act_out, out_pos=out_pos, o_stride=o_stride, o_off=o_off)
    return out


def conv3d_bf16(x, wpk, bias, cout, kernel, out, **kw):
    """bf16 operands, exact products, float32 accumulation == the float32 emulation on bf16-rounded inputs."""
    assert getattr(wpk, "bf16", False), "conv3d_bf16 needs pack_weights_bf16's image"
    act_in = kw.get("act_in", 0)
    assert act_in in (0, ACT_RELU) or (act_in == ACT_SIGMOID and kw.get("split3", False))
    if kw.pop("split3", False):
        assert wpk.split3
        return conv3d(x, wpk, bias, cout, kernel, out, **kw)
    assert not wpk.split3
    xb = Vox(_bf16(x.buf.float()), x.C, x.coff)
    return conv3d(xb, wpk, bias, cout, kernel, out, **kw)


def pack_weights_gather(w, cout, cin, s_co, s_ci, tap_ofs, kernel=None, bf16=False):
    """The logical operator of occd_pack_weights*_gather, materialised with an index gather."""
    flat = w.detach().float().reshape(-1)
    co = torch.arange(cout).view(-1, 1, 1) * s_co
    ci = torch.arange(cin).view(1, -1, 1) * s_ci
    idx = co + ci + torch.tensor([int(o) for o in tap_ofs]).view(1, 1, -1)
    sub = flat[idx].reshape(cout, cin, *kernel)
    if bf16 == "x3":
        return pack_weights_bf16(sub, split3=True)
    return pack_weights_bf16(sub) if bf16 else pack_weights(sub)


def conv3d_wgrad_bf16(x, gy, cin, cout, kernel, stride=(1, 1, 1), dilation=(1, 1, 1), padding=(0, 0, 0)):
    return conv3d_wgrad(Vox(_bf16(x.buf.float()), x.C, x.coff), Vox(_bf16(gy.buf.float()), gy.C, gy.coff), cin, cout,
                        kernel, stride, dilation, padding)


def nchw_to_nhwc(x, cs=None):
    C = x.shape[1]
    cs = cs if cs is not None else round_up(C, 4)
    out = torch.zeros((x.shape[0],) + tuple(x.shape[2:]) + (cs,))
    out[..., :C] = x.permute(0, *range(2, x.dim()), 1)
    return out


def vox_from_ncdhw(x):
    C = x.shape[1]
    return Vox(nchw_to_nhwc(x.float(), round_up(C, 8)), C)


def softmax_channels(src, dst, n, dst_pad=0):
    dst.buf[..., dst.coff:dst.coff + n] = F.softmax(src.buf[..., src.coff:src.coff + n], dim=-1)
    dst.buf[..., dst.coff + n:dst.coff + n + dst_pad] = 0
    return dst


def flosp_sample(depth, trans, proj, ida, voxel_num, final_dim, d_min, d_max, mean_mode=True, grids=None):
    """K1a semantics from the published algorithm (oracle.frustum_grid is NOT used: trans already
    contains grid_to_lidar), one camera at a time."""
    from oracle.occdepth_oracle import _from_homogeneous
    B, V, D, h, w = depth.shape
    A, Bd, C = (int(v) for v in voxel_num)
    idx = torch.stack(torch.meshgrid(torch.arange(A), torch.arange(Bd), torch.arange(C), indexing="ij"), -1)
    pts = F.pad((idx.float() + 0.5).reshape(1, -1, 3), [0, 1], value=1.0).repeat(B, 1, 1)
    feat = mask = 0
    for v in range(V):
        if grids is not None:
            g = grids[v].reshape(B, -1, 3)
        else:
            cam = _from_homogeneous(torch.bmm(pts, trans[:, v].transpose(1, 2)))
            img = torch.bmm(F.pad(cam, [0, 1], value=1.0), proj[:, v].transpose(1, 2))
            uv = _from_homogeneous(img)
            dep = img[..., 2] - proj[:, v, None, 2, 3]
            bin_size = 2 * (d_max - d_min) / (D * (1 + D))
            bins = -0.5 + 0.5 * torch.sqrt(1 + 8 * (dep - d_min) / bin_size)
            fr = _from_homogeneous(torch.bmm(F.pad(torch.cat([uv, bins.unsqueeze(-1)], -1), [0, 1], value=1.0),
                                             ida[:, v].transpose(1, 2)))
            g = fr / (torch.tensor([final_dim[1], final_dim[0], D], dtype=torch.float32) - 1) * 2 - 1
            g = torch.where(torch.isfinite(g), g, torch.full_like(g, -2.0))
        g = g.reshape(B, A, Bd, C, 3)
        vol = depth[:, v].unsqueeze(1)
        feat = feat + F.grid_sample(vol, g, mode="bilinear", padding_mode="zeros", align_corners=False)
        mask = mask + F.grid_sample(torch.ones_like(vol), g, mode="bilinear", padding_mode="zeros",
                                    align_corners=False)
    if V > 1 and mean_mode:
        feat = torch.where(mask > 0, feat / mask.clamp_min(1e-30), feat)
    return feat.reshape(B, -1)


def lift(feats, scale_divs, pix, fov, n_dims, row_strides, out, depth_scale=None, scale_const=100.0, xcd_mode=None):
    from oracle.occdepth_oracle import sfa
    B = pix.shape[0]
    C = out.C
    dims = tuple(n_dims)
    for b in range(B):
        total = None
        for s, per_scale in enumerate(feats):
            x2d = torch.stack([f[b, ..., :C].permute(2, 0, 1) for f in per_scale])
            part = sfa(x2d, torch.div(pix[b], int(scale_divs[s]), rounding_mode="floor"), fov[b].bool(), dims, 1,
                       "kitti").reshape(C, -1)
            total = part if total is None else total + part
        if depth_scale is not None:
            total = total * depth_scale[b].reshape(1, -1) * scale_const
        n = torch.arange(total.shape[1])
        a = n // (dims[1] * dims[2])
        bb = (n // dims[2]) % dims[1]
        c = n % dims[2]
        rows = a * row_strides[0] + bb * row_strides[1] + c * row_strides[2]
        flat = out.buf[b].reshape(-1, out.cs)
        flat[rows, :C] = total.t()
        flat[rows, C:] = 0
    return out


def lift_proj(feats, scale_divs, cam_E, cam_k, origin, voxel_size, img_wh, n_dims, row_strides, out, frustum=None,
              scale_const=100.0, xcd_mode=None):
    """occd_lift_proj_fwd = the numpy vox2pix restatement per (sample, view) + the frustum sample + `lift`."""
    from oracle.inputs import vox2pix
    B, V = cam_E.shape[:2]
    scene = tuple(float(d) * voxel_size for d in n_dims)
    pix, fov = [], []
    for b in range(B):
        pv, fv = [], []
        for v in range(V):
            p, m, _ = vox2pix(cam_E[b, v].double().numpy(), cam_k[b, v].double().numpy(), origin, voxel_size, img_wh[0],
                              img_wh[1], scene, 0)
            pv.append(torch.from_numpy(p))
            fv.append(torch.from_numpy(m))
        pix.append(torch.stack(pv))
        fov.append(torch.stack(fv))
    ds = None
    if frustum is not None:
        ds = flosp_sample(frustum.depth, frustum.trans, frustum.proj, frustum.ida, frustum.voxel_num, frustum.final_dim,
                          frustum.d_min, frustum.d_max, frustum.mean_mode, frustum.grids)
    return lift(feats, scale_divs, torch.stack(pix), torch.stack(fov), n_dims, row_strides, out, depth_scale=ds,
                scale_const=scale_const)


def bottleneck3d(x, w, P, dilation, out=None):
    """K14 from its packed buffer (layout of occd_bottleneck3d_fwd), evaluated with ATen convolutions in float64."""
    C = x.C
    xin = x.buf[..., x.coff:x.coff + C].permute(0, 4, 1, 2, 3).double()
    w = w.double()
    off = [0]

    def take(n):
        t = w[off[0]:off[0] + n]
        off[0] += n
        return t

    def unfrag(flat, ntap, cin, cout):
        """fragment order (tap, t, m, g, i, e) -> (tap, cin, cout)"""
        f = flat.view(ntap, cin // 16, cout // 16, 4, 16, 4)
        return f.permute(0, 1, 3, 5, 2, 4).reshape(ntap, cin, cout)

    w1 = unfrag(take(C * P), 1, C, P)[0].t().reshape(P, C, 1, 1, 1)
    b1 = take(P)
    taps = []
    for axis in (2, 1, 0):                                       # conv2: Z, conv3: Y, conv4: X
        wk = unfrag(take(3 * P * P), 3, P, P).permute(2, 1, 0)  # (out, in, tap)
        shape = [1, 1, 1]
        shape[axis] = 3
        taps.append((wk.reshape(P, P, *shape), take(P), axis))
    w5 = unfrag(take(P * C), 1, P, C)[0].t().reshape(C, P, 1, 1, 1)
    b5 = take(C)
    assert off[0] == w.numel()

    def axis_conv(t, wk, bk, axis, d):
        pad, dil = [0, 0, 0], [1, 1, 1]
        pad[axis], dil[axis] = d, d
        return F.conv3d(t, wk, bk, padding=pad, dilation=dil)

    o1 = F.relu(F.conv3d(xin, w1, b1))
    o2 = axis_conv(o1, *taps[0], dilation[0])
    o3 = axis_conv(F.relu(o2), *taps[1], dilation[1]) + o2
    o4 = axis_conv(F.relu(o3), *taps[2], dilation[2]) + o2 + o3
    y = F.relu(F.conv3d(F.relu(o4), w5, b5) + xin)
    if out is None:
        out = Vox(torch.zeros(x.buf.shape[:-1] + (round_up(C, 8),)), C)
    out.buf[..., out.coff:out.coff + C] = y.permute(0, 2, 3, 4, 1).float()
    return out


def cascade_tail(part, occ_off, wn, nbr):
    soft = F.softmax(part.buf[..., occ_off:occ_off + 2], dim=-1).permute(0, 4, 1, 2, 3)
    y = F.conv3d(soft, wn.detach().float(), None, padding=1).permute(0, 2, 3, 4, 1) + part.buf[..., :nbr]
    out = Vox(torch.zeros(part.buf.shape[:-1] + (round_up(nbr, 4),)), nbr)
    out.buf[..., :nbr] = y
    return out


def conv3d_wgrad(x, gy, cin, cout, kernel, stride=(1, 1, 1), dilation=(1, 1, 1), padding=(0, 0, 0)):
    xin = x.buf[..., x.coff:x.coff + cin].permute(0, 4, 1, 2, 3).contiguous()
    g = gy.buf[..., gy.coff:gy.coff + cout].permute(0, 4, 1, 2, 3).contiguous()
    return torch.nn.grad.conv3d_weight(xin.double(), (cout, cin) + tuple(kernel), g.double(), stride=stride,
                                       padding=padding, dilation=dilation).float()


_BT = torch.tensor([[1., 0., -1., 0.], [0., 1., 1., 0.], [0., -1., 1., 0.], [0., 1., 0., -1.]], dtype=torch.float64)
_AT = torch.tensor([[1., 1., 1., 0.], [0., 1., -1., -1.]], dtype=torch.float64)


def wino_input_transform(x, ty0=0, ths=None):
    B, C, H, W = x.shape
    th, tw = (H + 1) // 2, (W + 1) // 2
    ths = th - ty0 if ths is None else ths
    xp = F.pad(x.double(), (1, 2 * tw + 1 - W, 1, 2 * th + 1 - H))
    d = F.unfold(xp, kernel_size=4, stride=2).reshape(B, C, 4, 4, th, tw)[:, :, :, :, ty0:ty0 + ths]   # patches at (2ty-1, 2tx-1)
    v = torch.einsum("ia,bcakyx,jk->ijbyxc", _BT, d, _BT)
    return v.reshape(16, B * ths * tw, C).float()


def wino_output_transform(M, shape, scale=None, shift=None, act=None, slope=0.01, res=None, res_first=False, out=None,
                          ty0=0, ths=None):
    B, C, H, W = shape
    th, tw = (H + 1) // 2, (W + 1) // 2
    ths = th - ty0 if ths is None else ths
    r0, r1 = 2 * ty0, min(2 * (ty0 + ths), H)
    m = M.double().reshape(4, 4, B, ths, tw, C)
    y = torch.einsum("ri,ijbyxc,sj->bcyrxs", _AT, m, _AT).reshape(B, C, 2 * ths, 2 * tw)[:, :, :r1 - r0, :W]
    if scale is not None:
        y = y * scale.double().view(1, -1, 1, 1)
    if shift is not None:
        y = y + shift.double().view(1, -1, 1, 1)
    rs = res.double()[:, :, r0:r1] if res is not None else None
    if rs is not None and res_first:
        y = y + rs
    if act == "relu":
        y = F.relu(y)
    elif act == "leaky":
        y = F.leaky_relu(y, slope)
    elif act == "swish":
        y = y * torch.sigmoid(y)
    if rs is not None and not res_first:
        y = y + rs
    if out is None:
        out = torch.empty(shape)
    out[:, :, r0:r1] = y.float()
    return out


class _PackedWino:
    def __init__(self, w):
        self.w = w

    def numel(self):
        return -1


def wino_pack_weights(w, scale=None):
    w = w.detach().double()
    return _PackedWino(w * scale.double().view(-1, 1, 1, 1) if scale is not None else w)


def conv2d_3x3_fused(x, upk, cout, shift=None, act=None, slope=0.01, res=None, res_first=False, tile_hint=0, out=None):
    """K10 semantics: act(conv3x3(x, g * scale, pad 1) + shift) (+ res)."""
    y = F.conv2d(x.double(), upk.w, None, padding=1)
    if shift is not None:
        y = y + shift.double().view(1, -1, 1, 1)
    if res is not None and res_first:
        y = y + res.double()
    if act == "relu":
        y = F.relu(y)
    elif act == "leaky":
        y = F.leaky_relu(y, slope)
    elif act == "swish":
        y = y * torch.sigmoid(y)
    if res is not None and not res_first:
        y = y + res.double()
    if out is not None:
        out.copy_(y.float())
        return out
    return y.float()


def dwconv2d_same_pool(x, w, scale, shift, stride, act=None):
    B, C, H, W = x.shape
    k = w.shape[-1]
    Ho, Wo = -(-H // stride), -(-W // stride)
    ph, pw_ = max((Ho - 1) * stride + k - H, 0), max((Wo - 1) * stride + k - W, 0)
    xp = F.pad(x.double(), [pw_ // 2, pw_ - pw_ // 2, ph // 2, ph - ph // 2])
    y = F.conv2d(xp, w.double(), None, stride, 0, 1, C)
    if scale is not None:
        y = y * scale.double().view(1, -1, 1, 1) + shift.double().view(1, -1, 1, 1)
    if act == "swish":
        y = y * torch.sigmoid(y)
    elif act == "relu":
        y = F.relu(y)
    y = y.float()
    return y, y.double().sum((2, 3)).reshape(B * C, 1).float(), Ho * Wo


def se_gate(part, plane_size, batch, w_reduce, b_reduce, w_expand, b_expand):
    C = part.shape[0] // batch
    mean = part.double().sum(1).reshape(batch, C) / plane_size
    r = mean @ w_reduce.double().reshape(-1, C).t() + b_reduce.double()
    r = r * torch.sigmoid(r)
    return torch.sigmoid(r @ w_expand.double().reshape(C, -1).t() + b_expand.double()).float()


def pw_pack_weights(w, scale=None):
    w = w.detach().double().reshape(w.shape[0], w.shape[1])
    return _PackedWino(w * scale.double().view(-1, 1) if scale is not None else w)


def conv1x1(x, wpk, cout, shift=None, act=None, slope=0.01, gate=None, res=None, tile_hint=0, out=None, nhwc=False):
    """K11 semantics: act(conv1x1(x * gate, w * scale) + shift) (+ res)."""
    xd = x.double()
    if gate is not None:
        xd = xd * gate.double().reshape(x.shape[0], x.shape[1], *([1] * (x.dim() - 2)))
    y = torch.einsum("oc,bc...->bo...", wpk.w, xd)
    if shift is not None:
        y = y + shift.double().view(1, -1, *([1] * (x.dim() - 2)))
    if act == "relu":
        y = F.relu(y)
    elif act == "leaky":
        y = F.leaky_relu(y, slope)
    elif act == "swish":
        y = y * torch.sigmoid(y)
    if res is not None:
        y = y + res.double()
    if out is not None:
        out.copy_(y.float())
        return out
    if nhwc:
        return y.float().contiguous(memory_format=torch.channels_last) if y.dim() == 4 else y.float()
    return y.float()


def _softmax_and_target(logits, target, map_occ):
    B, C = logits.shape[:2]
    p = F.softmax(logits.detach().double().reshape(B, C, -1), 1)        # (B, C, S)
    t = target.reshape(B, -1).long()
    if map_occ:
        t = torch.where((t != 0) & (t != 255), torch.ones_like(t), t)
    return p, t


def ssc_loss_stats(logits, target, masks=None, weights=None, map_occ=False):
    """Fixed-point statistics exactly as include/occdepth_amd.h lays them out (float64 math, then quantised)."""
    B, C = logits.shape[:2]
    p, t = _softmax_and_target(logits, target, map_occ)
    lab = (t != 255)
    w = torch.ones(C, dtype=torch.float64) if weights is None else weights.double()
    onehot = torch.zeros(B, C, t.shape[1], dtype=torch.float64)
    onehot.scatter_(1, t.clamp(max=C - 1).unsqueeze(1), 1.0)
    onehot = onehot * (lab & (t < C)).unsqueeze(1)
    P = (p * lab.unsqueeze(1)).sum((0, 2))
    N = (p * onehot).sum((0, 2))
    T = onehot.sum((0, 2))
    M = lab.sum().double()
    logp = torch.log_softmax(logits.detach().float().reshape(B, C, -1), 1).double()
    wt = (onehot * w.view(1, C, 1)).sum(1)
    num = -(logp * onehot).sum(1).mul(wt).sum()
    den = wt.sum()
    parts = [P * hip.SSC_Q32, N * hip.SSC_Q32, T, M.view(1), (num * hip.SSC_Q24).view(1), (den * hip.SSC_Q24).view(1)]
    if masks is not None:
        m = masks.reshape(B, masks.shape[1], -1).double()
        parts.append((torch.einsum("bfs,bcs->fc", m, p) * hip.SSC_Q32).reshape(-1))
    return torch.cat(parts).round().to(torch.int64)


def ssc_loss_grad(logits, target, masks, weights, gstats, map_occ=False):
    B, C = logits.shape[:2]
    p, t = _softmax_and_target(logits, target, map_occ)
    lab = (t != 255)
    g = gstats.double()
    onehot = torch.zeros_like(p)
    onehot.scatter_(1, t.clamp(max=C - 1).unsqueeze(1), 1.0)
    onehot = onehot * (lab & (t < C)).unsqueeze(1)
    gv = lab.unsqueeze(1) * g[:C].view(1, C, 1) + onehot * g[C:2 * C].view(1, C, 1)
    if masks is not None:
        m = masks.reshape(B, masks.shape[1], -1).double()
        gv = gv + torch.einsum("bfs,fc->bcs", m, g[3 * C + 3:].view(-1, C))
    dot = (p * gv).sum(1, keepdim=True)
    w = torch.ones(C, dtype=torch.float64) if weights is None else weights.double()
    wt = (onehot * w.view(1, C, 1)).sum(1, keepdim=True)
    grad = p * (gv - dot) + g[3 * C + 1] * wt * (p * onehot.sum(1, keepdim=True) - onehot)
    return grad.float().reshape(logits.shape)


def ssc_confusion(hist, target, logits=None, labels=None):
    C = hist.shape[0]
    t = target.reshape(-1).long()
    pred = labels.reshape(-1).long() if labels is not None else \
        logits.reshape(logits.shape[0], C, -1).argmax(1).reshape(-1)
    keep = (t != 255) & (t < C) & (pred < C)
    hist += torch.bincount(t[keep] * C + pred[keep], minlength=C * C).reshape(C, C)
    return hist


# ---- round 5: relation / depth BCE passes and the frustum-sample transpose (include/occdepth_amd.h semantics, float64) ----
def _relation_xy(logits, labels):
    x = logits.detach().double()                                       # (B, R, M, N)
    y = labels.permute(0, 1, 3, 2) != 0                                # labels (B, R, N, M)
    return x, y


def relation_bce_stats(logits, labels):
    x, y = _relation_xy(logits, labels)
    sp = (F.softplus(-x) * y).sum((0, 2, 3))
    sn = (F.softplus(x) * ~y).sum((0, 2, 3))
    return torch.stack([y.sum((0, 2, 3)).double(), (sp * hip.REL_Q24).round(), (sn * hip.REL_Q24).round()], 1).to(torch.int64)


def relation_bce_grad(logits, labels, coef):
    x, y = _relation_xy(logits, labels)
    c = coef.double()
    g = torch.where(y, -c[:, 0].view(1, -1, 1, 1) * torch.sigmoid(-x), c[:, 1].view(1, -1, 1, 1) * torch.sigmoid(x))
    return g.float()


def _depth_target_bin(gt, cell, d_off, d_step, D, h, w):
    """Reference depth_loss.py:14-52 + the nearest resample of :72-76 -> target bin per cell (-1 = no target)."""
    lab = F.interpolate(gt.float().unsqueeze(1), (h * cell, w * cell), mode="nearest")[:, 0]
    blocks = lab.reshape(-1, h, cell, w, cell)
    nearest = blocks.masked_fill(blocks == 0.0, 1e5).amin(dim=(2, 4))
    idx = (nearest - torch.tensor(d_off, dtype=torch.float32)) / torch.tensor(d_step, dtype=torch.float32)
    k = torch.where((idx < D + 1) & (idx >= 0.0), idx, torch.zeros_like(idx)).long()
    return k - 1                                                        # (Bn, h, w)


def depth_bce_stats(prob, gt, cell, d_off, d_step):
    Bn, D, h, w = prob.shape
    kb = _depth_target_bin(gt, cell, d_off, d_step, D, h, w)
    p = prob.detach().double()
    t = F.one_hot((kb + 1).clamp(min=0), D + 1)[..., 1:].permute(0, 3, 1, 2).double()
    bce = -(t * torch.log(p).clamp(min=-100.0) + (1 - t) * torch.log1p(-p).clamp(min=-100.0)).sum(1)
    meas = kb >= 0
    return torch.stack([((bce * meas).sum() * hip.REL_Q24).round(), meas.sum().double()]).to(torch.int64)


def depth_bce_grad(prob, gt, cell, d_off, d_step, gscale):
    Bn, D, h, w = prob.shape
    kb = _depth_target_bin(gt, cell, d_off, d_step, D, h, w)
    p = prob.detach().double()
    t = F.one_hot((kb + 1).clamp(min=0), D + 1)[..., 1:].permute(0, 3, 1, 2).double()
    g = (p - t) / ((1 - p) * p).clamp(min=1e-12) * gscale.double().reshape(())
    return (g * (kb >= 0).unsqueeze(1)).float()


def flosp_sample_bwd(fr, gout):
    # float32 geometry like the kernel (and like the reference's grid); the sample is linear in the volume, so autograd
    # through the emulated forward IS the transpose of the weights that forward applies
    d = fr.depth.detach().float().cpu().requires_grad_(True)
    cpu = lambda t: None if t is None else t.detach().float().cpu()
    out = flosp_sample(d, cpu(fr.trans), cpu(fr.proj), cpu(fr.ida), fr.voxel_num, fr.final_dim, fr.d_min, fr.d_max,
                       fr.mean_mode, cpu(fr.grids))
    return torch.autograd.grad(out, d, gout.detach().float().cpu().reshape(out.shape))[0]


def depthnet_gate(mlp, se, images, sps=None, intrins=None, factor=1000.0):
    if intrins is not None:
        k = intrins.detach().double().reshape(images, -1)
        sps = torch.sqrt((1.0 / k[:, 0]) ** 2 + (1.0 / k[:, 5]) ** 2) * factor
    s = sps.detach().double().reshape(images, 1)
    C = mlp.fc2.out_features
    d = lambda t: t.detach().double()
    h = F.relu(s * d(mlp.fc1.weight).reshape(1, C) + d(mlp.fc1.bias))
    m = h @ d(mlp.fc2.weight).t() + d(mlp.fc2.bias)
    r = F.relu(m @ d(se.conv_reduce.weight).reshape(C, C).t() + d(se.conv_reduce.bias))
    return torch.sigmoid(r @ d(se.conv_expand.weight).reshape(C, C).t() + d(se.conv_expand.bias)).float()


def stem_conv3x3(x, w, scale, shift, stride, act=None):
    H, W = x.shape[-2:]
    Ho, Wo = -(-H // stride), -(-W // stride)
    ph, pw = max((Ho - 1) * stride + 3 - H, 0), max((Wo - 1) * stride + 3 - W, 0)
    y = F.conv2d(F.pad(x.double(), [pw // 2, pw - pw // 2, ph // 2, ph - ph // 2]), w.detach().double(), None, stride)
    if scale is not None:
        y = y * scale.double().view(1, -1, 1, 1)
    if shift is not None:
        y = y + shift.double().view(1, -1, 1, 1)
    return _act2d(y, act, 0.0).float()


def upconv_gather(z, cout, size, batch_inner=False, skip=None, wskip=None, shift=None, slope=0.01):
    """K12 semantics: sum over the 9 taps of shift_t(bilinear_up(z_t, align_corners=True)), zero outside the grid."""
    if batch_inner:
        z = z.permute(1, 0, 2, 3)
    H, W = int(size[0]), int(size[1])
    out = torch.zeros(z.shape[0], cout, H, W, dtype=torch.float64)
    for ky in range(3):
        for kx in range(3):
            t = ky * 3 + kx
            up = F.interpolate(z[:, t * cout:(t + 1) * cout].double(), size=(H, W), mode="bilinear", align_corners=True)
            out += F.pad(up, (1, 1, 1, 1))[:, :, ky:ky + H, kx:kx + W]
    if skip is not None:
        out = F.leaky_relu(out + F.conv2d(skip.double(), wskip.double(), shift.double(), padding=1), slope)
    return out.float()


def _act2d(y, act, slope):
    if act == "relu":
        return F.relu(y)
    if act == "leaky":
        return F.leaky_relu(y, slope)
    if act == "swish":
        return y * torch.sigmoid(y)
    return y


def affine_act(x, scale, shift, act=None, slope=0.01, res=None, res_first=False, out=None):
    """occd_affine_act_nchw semantics: act(x * scale[c] + shift[c]) with the residual before or after the activation."""
    y = x.double()
    shape = (1, -1) + (1,) * (x.dim() - 2)
    if scale is not None:
        y = y * scale.double().view(shape)
    if shift is not None:
        y = y + shift.double().view(shape)
    if res is not None and res_first:
        y = y + res.double()
    y = _act2d(y, act, slope)
    if res is not None and not res_first:
        y = y + res.double()
    return y.float()


def dwconv2d_same(x, w, scale, shift, stride, act=None):
    return dwconv2d_same_pool(x, w, scale, shift, stride, act)[0]


def upsample_bilinear_cat(x, skip):
    up = F.interpolate(x.double(), size=skip.shape[2:], mode="bilinear", align_corners=True)
    return torch.cat([up, skip.double()], 1).float()


def softmax_nchw(x):
    return torch.softmax(x.double(), 1).float()


def gemm_x3_supported(a, b):
    """hip.gemm_x3_supported without the device check (the layout rules are the kernel's own)."""
    if a.dtype != torch.float32 or b.dtype != torch.float32:
        return False
    if a.dim() not in (2, 3) or b.dim() not in (2, 3) or a.stride(-1) != 1 or b.stride(-1) != 1:
        return False
    K, N = b.shape[-2], b.shape[-1]
    if a.shape[-1] != K or K % 8 or N < 4 or a.stride(-2) % 4 or a.stride(-2) < K or b.stride(-2) < N:
        return False
    return a.dim() == 2 or b.dim() == 2 or a.shape[0] == b.shape[0]


def gemm_x3(a, b, bias=None, act=None, slope=0.01, out=None, tile_hint=0, res=None, k_scale=None, sigmoid_a=False):
    """occd_gemm_f32x3: float32-level GEMM (B rows scaled by k_scale, rounded to float32 like the kernel's staging) + bias[:, None]
    + activation + res (evaluated in float64 here)."""
    if k_scale is not None:
        b = b * k_scale.unsqueeze(-1)
    y = torch.matmul(torch.sigmoid(a.double()) if sigmoid_a else a.double(), b.double())
    if bias is not None:
        y = y + bias.double().view(-1, 1)
    y = _act2d(y, act, slope)
    if res is not None:
        y = y + res.double()
    y = y.float()
    if out is not None:
        out.copy_(y if y.dim() == out.dim() else y.unsqueeze(0))
        return out
    return y if (a.dim() == 3 or b.dim() == 3) else y


@contextlib.contextmanager
def patched(fast2d=False):
    """fast2d: also route the 2-D eval fast paths (fused.on_gpu gates) through the emulation on CPU tensors."""
    saved = {k: getattr(hip, k) for k in ("affine_act", "dwconv2d_same", "upsample_bilinear_cat", "softmax_nchw", "pack_weights", "conv3d", "nchw_to_nhwc", "softmax_channels",
                                          "flosp_sample", "lift", "lift_proj", "bottleneck3d", "cascade_tail", "ssc_loss_stats", "ssc_loss_grad",
                                          "ssc_confusion", "conv3d_wgrad", "wino_input_transform", "wino_output_transform",
                                          "wino_pack_weights", "conv2d_3x3_fused", "pw_pack_weights", "conv1x1",
                                          "dwconv2d_same_pool", "se_gate", "upconv_gather", "pack_weights_bf16", "pack_weights_gather", "conv3d_bf16",
                                          "conv3d_wgrad_bf16", "gemm_x3", "gemm_x3_supported", "conv3d_phases",
                                          "relation_bce_stats", "relation_bce_grad", "relation_bce_usable", "depth_bce_stats",
                                          "depth_bce_grad", "depth_bce_usable", "flosp_sample_bwd", "stem_conv3x3", "depthnet_gate")}
    hip.stem_conv3x3, hip.depthnet_gate = stem_conv3x3, depthnet_gate
    hip.relation_bce_stats, hip.relation_bce_grad = relation_bce_stats, relation_bce_grad
    hip.depth_bce_stats, hip.depth_bce_grad, hip.flosp_sample_bwd = depth_bce_stats, depth_bce_grad, flosp_sample_bwd
    # the gates of the loss kernels without their is_cuda condition: the CPU suite drives the same autograd Functions
    hip.relation_bce_usable = lambda lg, lb: (hip.LOSS_KERNELS and lg.dim() == 4 and lg.dtype == torch.float32 and
                                              tuple(lb.shape) == (lg.shape[0], lg.shape[1], lg.shape[3], lg.shape[2]))
    hip.depth_bce_usable = lambda pr, lb: hip.LOSS_KERNELS and pr.dtype == torch.float32
    hip.conv3d_phases = conv3d_phases
    hip.pack_weights_bf16, hip.conv3d_bf16, hip.conv3d_wgrad_bf16 = pack_weights_bf16, conv3d_bf16, conv3d_wgrad_bf16
    hip.pack_weights_gather = pack_weights_gather
    hip.gemm_x3, hip.gemm_x3_supported = gemm_x3, gemm_x3_supported
    hip.upconv_gather = upconv_gather
    hip.affine_act, hip.dwconv2d_same = affine_act, dwconv2d_same
    hip.upsample_bilinear_cat, hip.softmax_nchw = upsample_bilinear_cat, softmax_nchw
    saved_on_gpu = fused.on_gpu
    if fast2d:
        fused.on_gpu = lambda t: True
    hip.dwconv2d_same_pool, hip.se_gate = dwconv2d_same_pool, se_gate
    hip.wino_pack_weights, hip.conv2d_3x3_fused = wino_pack_weights, conv2d_3x3_fused
    hip.pw_pack_weights, hip.conv1x1 = pw_pack_weights, conv1x1
    hip.conv3d_wgrad = conv3d_wgrad
    hip.wino_input_transform, hip.wino_output_transform = wino_input_transform, wino_output_transform
    hip.ssc_loss_stats, hip.ssc_loss_grad, hip.ssc_confusion = ssc_loss_stats, ssc_loss_grad, ssc_confusion
    saved_from = Vox.from_ncdhw
    saved_as_vox = fused.as_vox
    hip.pack_weights, hip.conv3d, hip.nchw_to_nhwc = pack_weights, conv3d, nchw_to_nhwc
    hip.softmax_channels, hip.flosp_sample, hip.lift = softmax_channels, flosp_sample, lift
    hip.lift_proj = lift_proj
    hip.bottleneck3d = bottleneck3d
    saved_sample = hip.Frustum.sample
    hip.Frustum.sample = lambda fr: flosp_sample(fr.depth, fr.trans, fr.proj, fr.ida, fr.voxel_num, fr.final_dim, fr.d_min,
                                                 fr.d_max, fr.mean_mode, fr.grids)
    hip.cascade_tail = cascade_tail
    Vox.from_ncdhw = staticmethod(vox_from_ncdhw)

    def as_vox_cpu(x):
        if isinstance(x, Vox):
            return x
        cl = x.float().permute(0, 2, 3, 4, 1)
        return Vox(cl, x.shape[1]) if cl.is_contiguous() and x.shape[1] % 8 == 0 else vox_from_ncdhw(x)

    import occdepth_amd.models.CRP3D as crp
    import occdepth_amd.models.DDR as ddr
    import occdepth_amd.models.modules as mods
    import occdepth_amd.models.unet3d_kitti as u3k
    import occdepth_amd.models.unet3d_nyu as u3n
    import occdepth_amd.models.unet3d_common as u3c
    users = [fused, crp, ddr, mods, u3k, u3n, u3c]
    old = [(m, m.as_vox) for m in users if hasattr(m, "as_vox")]
    for m, _ in old:
        m.as_vox = as_vox_cpu
    saved_prepare = u3c._VoxMode.prepare
    u3c._VoxMode.prepare = staticmethod(as_vox_cpu)
    try:
        yield
    finally:
        for k, v in saved.items():
            setattr(hip, k, v)
        fused.on_gpu = saved_on_gpu
        hip.Frustum.sample = saved_sample
        Vox.from_ncdhw = saved_from
        for m, f in old:
            m.as_vox = f
        u3c._VoxMode.prepare = staticmethod(saved_prepare)
