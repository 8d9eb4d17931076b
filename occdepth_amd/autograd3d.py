"""Differentiable 3-D convolutions on the HIP kernels (the training step).

In the reference, autograd sends every nn.Conv3d / nn.ConvTranspose3d of the 3-D stack through the backend's
convolution backward (occdepth/models/OccDepth.py:535-537 -> models/DDR.py, modules.py, CRP3D.py).  On ROCm that is
MIOpen's fp32 NCDHW path, which falls back to naive kernels for these shapes (measured: 3.8 s of a 4.0 s config-2
training step).  Here
    forward        = K2 / K2s implicit GEMM (csrc/conv3d_igemm.hip, conv3d_c32p.hip) on channels-last volumes,
    data gradient  = the SAME kernels on dL/dy with the flipped, channel-transposed weights; strided convolutions
                     decompose into the sub-pixel phases of the transposed convolution (output scatter),
    weight gradient = K8 (csrc/conv3d_wgrad.hip), deterministic,
and ConvTranspose3d is the same three pieces with forward and data-gradient swapped.
Tensors cross the boundary as (B, C, X, Y, Z) views with channels_last_3d strides, so chains of these layers
never transpose.
"""
import itertools
import os

import torch
import torch.nn as nn

from . import hip
from .hip import Vox

# BASELINE configs[3] (the bf16 training step): every convolution of this module -- forward, data gradient and weight
# gradient -- on the bf16 matrix pipe (K2b / K8b: v_mfma_f32_32x32x16_bf16, fp32 accumulate, fp32 master weights and
# fp32 gradients).  Activations keep the dtype they arrive in: float32 tensors stay float32 in HBM and are rounded to
# bf16 while a kernel stages them ("bf16 MFMA, fp32 storage"), bfloat16 tensors (an autocast region upstream) are
# consumed and produced as bfloat16.  Off (default): the exact-fp32 MFMA kernels K2 / K2s / K8.
BF16_MFMA = os.environ.get("OCCDEPTH_BF16_MFMA", "0") == "1"


def set_bf16_mfma(flag):
    """Switch the convolutions of the training path between the exact-fp32 and the bf16 matrix pipe; returns the old value."""
    global BF16_MFMA
    old, BF16_MFMA = BF16_MFMA, bool(flag)
    return old


def _to_vox(x):
    """(B, C, X, Y, Z) CUDA tensor -> Vox; zero-copy for channels_last_3d tensors with C % 8 == 0.  bfloat16 tensors stay
    bfloat16 in bf16 mode (anything else is float32)."""
    if x.dtype != torch.float32 and not (BF16_MFMA and x.dtype == torch.bfloat16):
        x = x.float()
    cl = x.permute(0, 2, 3, 4, 1)
    if x.shape[1] % 8 == 0 and cl.is_contiguous():
        return Vox(cl, x.shape[1])
    rows = _padded_rows(x)
    if rows is not None:
        return Vox(rows, x.shape[1])
    if x.dtype == torch.float32 and x.is_contiguous():
        return Vox.from_ncdhw(x)                       # NCDHW: one LDS-tiled transpose pass
    # channels-last with a ragged channel count (or a bf16 NCDHW tensor): one strided copy into zero-padded rows
    v = Vox(torch.zeros((x.shape[0],) + tuple(x.shape[2:]) + (hip.round_up(x.shape[1], 8),), device=x.device, dtype=x.dtype),
            x.shape[1])
    v.buf[..., :x.shape[1]].copy_(cl)
    return v


def _padded_rows(x):
    """Round 5: a (B, C, X, Y, Z) float32 view of a WHOLE buffer of zero-padded channels-last rows (B, X, Y, Z, ceil8(C)) --
    the layout `hip.ssc_loss_grad` writes the gradient of ragged-channel logits in (20 classes in rows of 24 floats, pads
    written as zeros) -- is consumed in place; anything else (a channel slice of a wider tensor, whose "pads" hold data)
    is not."""
    if x.dtype != torch.float32 or x.dim() != 5:
        return None
    B, C, X, Y, Z = x.shape
    st = x.stride()
    cs = hip.round_up(C, 8)
    if st != (X * Y * Z * cs, 1, Y * Z * cs, Z * cs, cs) or x.storage_offset() != 0:
        return None
    if x.untyped_storage().nbytes() != B * X * Y * Z * cs * 4:
        return None
    # the strides / offset / storage-size test alone also admits `wide[:, :C]` at offset 0 of a channels-last tensor
    # with cs logical channels (the first chunk torch.cat's backward narrows out of a 24-channel gradient): its "pad" lanes hold
    # the neighbour's DATA.  A view is taken in place only when its base IS the padded row buffer -- (B, X, Y, Z, cs) dense,
    # what hip.ssc_loss_grad allocates and fills, pads included -- not a wider (B, cs, X, Y, Z) tensor.
    base = x._base
    if base is not None and not (base.dim() == 5 and tuple(base.shape) == (B, X, Y, Z, cs) and base.is_contiguous()):
        return None
    return torch.as_strided(x, (B, X, Y, Z, cs), (X * Y * Z * cs, Y * Z * cs, Z * cs, cs, 1))


def _x3_head(x, cout, kernel, out, kw):
    """fp32 training mode: the full-resolution head convolutions (forward and data gradient) take K2s3 -- the 3-way bf16 split,
    float32-level accuracy, 0.51 against 0.90 ms per launch -- whenever the eval path's default does (fused.BF16X3)."""
    from . import fused
    return (not BF16_MFMA) and bool(fused.BF16X3) and x.buf.dtype == torch.float32 and hip.c32x3_eligible(x, cout, kernel, out, **kw)


def _conv(x, w, bias, cout, kernel, out, **kw):
    """One forward-kernel launch with the weights packed for the active matrix pipe."""
    if BF16_MFMA:
        return hip.conv3d_bf16(x, hip.pack_weights_bf16(w), bias, cout, kernel, out, **kw)
    if _x3_head(x, cout, kernel, out, kw):
        return hip.conv3d_bf16(x, hip.pack_weights_bf16(w, split3=True), bias, cout, kernel, out, split3=True, **kw)
    return hip.conv3d(x, hip.pack_weights(w), bias, cout, kernel, out, **kw)


def _conv_view(x, w, n_out, n_in, s_out, s_in, tap_ofs, kernel, out, **kw):
    """Forward-kernel launch whose operator is a VIEW of the dense weight `w` (hip.pack_weights_gather): no permute /
    index_select / contiguous temporaries."""
    if BF16_MFMA:
        return hip.conv3d_bf16(x, hip.pack_weights_gather(w, n_out, n_in, s_out, s_in, tap_ofs, kernel, bf16=True), None, n_out,
                               kernel, out, **kw)
    if _x3_head(x, n_out, kernel, out, kw):
        return hip.conv3d_bf16(x, hip.pack_weights_gather(w, n_out, n_in, s_out, s_in, tap_ofs, kernel, bf16="x3"), None, n_out,
                               kernel, out, split3=True, **kw)
    return hip.conv3d(x, hip.pack_weights_gather(w, n_out, n_in, s_out, s_in, tap_ofs, kernel), None, n_out, kernel, out, **kw)


def _wgrad(x, gy, cin, cout, K, stride, dilation, padding):
    """dW: K8b (bf16 MFMA through transposed LDS reads) for the shapes it is built for -- columns of >= 16 voxels and either
    many taps (3x3x3, 3x3, the 2x2x2 phases) or more than one cout tile --, the exact-fp32 K8 otherwise."""
    ntaps = K[0] * K[1] * K[2]
    if BF16_MFMA and gy.dims[2] >= 16 and (ntaps >= 8 or cout > 32) and ntaps <= 28 and x.buf.dtype == gy.buf.dtype:
        return hip.conv3d_wgrad_bf16(x, gy, cin, cout, K, stride, dilation, padding)
    if x.buf.dtype != torch.float32:
        x = Vox(x.buf.float(), x.C, x.coff)
    if gy.buf.dtype != torch.float32:
        gy = Vox(gy.buf.float(), gy.C, gy.coff)
    return hip.conv3d_wgrad(x, gy, cin, cout, K, stride, dilation, padding)


def _axis_phases(K, s, p, d):
    """Per-axis decomposition of dx[i] = sum_{o, k : o*s - p + k*d = i} w[k] gy[o] into phases r = i mod s.
    -> list of (r, taps (ascending source offset), sub-dilation, leading pad)."""
    out = []
    for r in range(s):
        ks = [k for k in range(K) if (r + p - k * d) % s == 0]
        if not ks:
            out.append((r, [], 1, 0))
            continue
        offs = sorted(((r + p - k * d) // s, k) for k in ks)          # gy index = j + offset
        cs = [c for c, _ in offs]
        step = cs[1] - cs[0] if len(cs) > 1 else 1
        if any(b - a != step for a, b in zip(cs, cs[1:])):
            raise NotImplementedError(f"transposed phase of K={K} s={s} p={p} d={d} is not a uniform convolution")
        # leading pad -cs[0]; negative when the convolution padded more than its kernel reaches (a 1x1 convolution with
        # padding 1, occdepth/models/unet2d.py:65-67): the data gradient then CROPS gy, which the kernels' coordinate
        # arithmetic (x_in = x_out * stride - pad + tap * dilation, bounds-checked) expresses as a negative pad
        out.append((r, [k for _, k in offs], step, -cs[0]))
    return out


def conv3d_dgrad(gy, w, in_dims, stride, padding, dilation):
    """dL/dx (Vox, (B, in_dims, cin)) of y = conv3d(x, w) given gy = dL/dy (Vox).  w: (cout, cin, kx, ky, kz)."""
    cout, cin = w.shape[:2]
    K = tuple(w.shape[2:])
    wd = w.detach()
    wd = wd if wd.dtype == torch.float32 and wd.is_contiguous() else wd.float().contiguous()
    ntap = K[0] * K[1] * K[2]
    out = Vox.empty(gy.batch, in_dims, cin, gy.buf.device, dtype=gy.buf.dtype)
    axes = [_axis_phases(K[a], stride[a], padding[a], dilation[a]) for a in range(3)]
    if any(not taps for ax in axes for _, taps, _, _ in ax) or out.cs != cin:
        out.buf.zero_()                                                    # empty phases / channel pad
    for (rx, tx, dx_, px), (ry, ty, dy_, py), (rz, tz, dz_, pz) in itertools.product(*axes):
        if not (tx and ty and tz):
            continue
        n_pos = tuple((in_dims[a] - r + stride[a] - 1) // stride[a] for a, r in enumerate((rx, ry, rz)))
        if min(n_pos) <= 0:
            continue
        # the transposed operator of this phase, W'[ci][co][a, b, c] = w[co][ci][tx[a], ty[b], tz[c]], packed straight
        # from w (one launch; the permute + 3 index_select + contiguous chain was 5 launches per phase)
        ofs = [(a * K[1] + b_) * K[2] + c for a in tx for b_ in ty for c in tz]
        _conv_view(gy, wd, cin, cout, ntap, cin * ntap, ofs, (len(tx), len(ty), len(tz)), out, dilation=(dx_, dy_, dz_),
                   padding=(px, py, pz), out_pos=n_pos, o_stride=tuple(stride), o_off=(rx, ry, rz), cin=cout)
    return out


def _out_dims(dims, K, stride, padding, dilation):
    return tuple((n + 2 * p - d * (k - 1) - 1) // s + 1 for n, k, s, p, d in zip(dims, K, stride, padding, dilation))


def _like(gy, ref_dtype):
    """The incoming gradient in the storage type of the saved activation (the two operands of a launch share it)."""
    return gy if gy.dtype == ref_dtype else gy.to(ref_dtype)


class _Conv3dFn(torch.autograd.Function):
    # No autocast casting here: float32 activations are consumed as they are (exact-fp32 MFMA, or -- BF16_MFMA -- rounded to
    # bf16 inside the kernels), bfloat16 activations are taken as bfloat16 in bf16 mode and widened otherwise.  The weights
    # are always the float32 master copies.
    @staticmethod
    def forward(ctx, x, w, b, stride, padding, dilation):
        with torch.autocast("cuda", enabled=False):
            xv = _to_vox(x.detach())
            cout, cin = w.shape[:2]
            K = tuple(w.shape[2:])
            out = Vox.empty(xv.batch, _out_dims(xv.dims, K, stride, padding, dilation), cout, x.device, dtype=xv.buf.dtype)
            if out.cs != cout:
                out.buf.zero_()
            bias = None
            if b is not None:
                bias = torch.zeros(hip.round_up(cout, 32), device=x.device)
                bias[:cout] = b.detach().float()
            _conv(xv, w.detach().float(), bias, cout, K, out, stride=stride, dilation=dilation, padding=padding, cin=cin)
        ctx.save_for_backward(xv.buf, w)
        ctx.geom = (xv.C, xv.coff, stride, padding, dilation, b is not None)
        return out.ncdhw()

    @staticmethod
    def backward(ctx, gy):
        xbuf, w = ctx.saved_tensors
        C, coff, stride, padding, dilation, has_bias = ctx.geom
        with torch.autocast("cuda", enabled=False):
            xv = Vox(xbuf, C, coff)
            gyv = _to_vox(_like(gy, xbuf.dtype))
            cout, cin = w.shape[:2]
            K = tuple(w.shape[2:])
            dx = dw = db = None
            if ctx.needs_input_grad[0]:
                dx = conv3d_dgrad(gyv, w.float(), xv.dims, stride, padding, dilation).ncdhw()
            if ctx.needs_input_grad[1]:
                dw = _wgrad(xv, gyv, cin, cout, K, stride, dilation, padding).to(w.dtype)
            if has_bias and ctx.needs_input_grad[2]:
                db = gyv.buf.reshape(-1, gyv.cs)[:, gyv.coff:gyv.coff + cout].sum(0, dtype=torch.float32)
        return dx, dw, db, None, None, None


class _ConvTranspose3dFn(torch.autograd.Function):
    """y = conv_transpose3d(x, w (cin, cout, k)) == the data gradient of the convolution whose weight is w."""

    @staticmethod
    def forward(ctx, x, w, b, stride, padding, output_padding, dilation):
        with torch.autocast("cuda", enabled=False):
            xv = _to_vox(x.detach())
            cin, cout = w.shape[:2]
            K = tuple(w.shape[2:])
            dims = tuple((n - 1) * s - 2 * p + d * (k - 1) + op + 1
                         for n, k, s, p, d, op in zip(xv.dims, K, stride, padding, dilation, output_padding))
            out = conv3d_dgrad(xv, w.detach().float(), dims, stride, padding, dilation)
            y = out.ncdhw()
            if b is not None:
                y = y + b.detach().to(y.dtype).view(1, -1, 1, 1, 1)
        ctx.save_for_backward(xv.buf, w)
        ctx.geom = (xv.C, xv.coff, stride, padding, dilation, b is not None)
        return y

    @staticmethod
    def backward(ctx, gy):
        xbuf, w = ctx.saved_tensors
        C, coff, stride, padding, dilation, has_bias = ctx.geom
        with torch.autocast("cuda", enabled=False):
            xv = Vox(xbuf, C, coff)
            gyv = _to_vox(_like(gy, xbuf.dtype))
            cin, cout = w.shape[:2]
            K = tuple(w.shape[2:])
            dx = dw = db = None
            if ctx.needs_input_grad[0]:                     # the convolution (weight w: "cout" = cin of the transpose)
                o = Vox.empty(xv.batch, xv.dims, cin, gy.device, dtype=gyv.buf.dtype)
                if o.cs != cin:
                    o.buf.zero_()
                _conv(gyv, w.detach().float(), None, cin, K, o, stride=stride, dilation=dilation,
                      padding=padding, out_pos=xv.dims, cin=cout)
                dx = o.ncdhw()
            if ctx.needs_input_grad[1]:                     # roles swapped: gy is the conv's input, x its output gradient
                dw = _wgrad(gyv, xv, cout, cin, K, stride, dilation, padding).to(w.dtype)
            if has_bias and ctx.needs_input_grad[2]:
                db = gyv.buf.reshape(-1, gyv.cs)[:, gyv.coff:gyv.coff + cout].sum(0, dtype=torch.float32)
        return dx, dw, db, None, None, None, None


def conv2d_cl(x, w, b=None, padding=0):
    """nn.Conv2d (stride 1) on the 3-D convolution kernels: the (B, C, H, W) image is the X = 1 volume (B, C, 1, H, W) of
    channels-last pixel rows, the (Cout, Cin, kh, kw) weight its (1, kh, kw) kernel.  Used by the 2-D decoder in bf16 mode
    (occdepth/models/unet2d.py:24-46, 65-67, 120-131): forward, data gradient and weight gradient then run on K2b / K8b.
    Returns a (B, Cout, H', W') tensor with channels-last memory."""
    p = (0, int(padding), int(padding))
    return _Conv3dFn.apply(x.unsqueeze(2), w.unsqueeze(2), b, (1, 1, 1), p, (1, 1, 1)).squeeze(2)


def _hip_ok(mod, x):
    return (x.is_cuda and x.dtype in (torch.float32, torch.bfloat16, torch.float16) and mod.weight.dtype == torch.float32
            and mod.groups == 1
            and mod.padding_mode == "zeros" and not isinstance(mod.padding, str))


class Conv3d(nn.Conv3d):
    """nn.Conv3d whose CUDA fp32 forward/backward run on the HIP kernels (same parameters, same state_dict)."""

    def forward(self, x):
        if _hip_ok(self, x):
            return _Conv3dFn.apply(x, self.weight, self.bias, tuple(self.stride), tuple(self.padding),
                                   tuple(self.dilation))
        return super().forward(x)


class ConvTranspose3d(nn.ConvTranspose3d):
    def forward(self, x, output_size=None):
        if _hip_ok(self, x) and output_size is None:
            return _ConvTranspose3dFn.apply(x, self.weight, self.bias, tuple(self.stride), tuple(self.padding),
                                            tuple(self.output_padding), tuple(self.dilation))
        return super().forward(x, output_size)
