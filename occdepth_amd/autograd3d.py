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

import torch
import torch.nn as nn

from . import hip
from .hip import Vox


def _to_vox(x):
    """(B, C, X, Y, Z) float32 CUDA tensor -> Vox; zero-copy for channels_last_3d tensors with C % 8 == 0."""
    cl = x.permute(0, 2, 3, 4, 1)
    if x.shape[1] % 8 == 0 and cl.is_contiguous():
        return Vox(cl, x.shape[1])
    return Vox.from_ncdhw(x)


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
        if any(b - a != step for a, b in zip(cs, cs[1:])) or cs[0] > 0:
            raise NotImplementedError(f"transposed phase of K={K} s={s} p={p} d={d} is not a uniform convolution")
        out.append((r, [k for _, k in offs], step, -cs[0]))
    return out


_TAPS = {}


def _taps(ks, device):
    """Tap index list -> cached device LongTensor (indexing with a Python list uploads it on every call: a host sync per
    phase per layer, and illegal inside hipGraph capture of the training step)."""
    key = (tuple(ks), str(device))
    t = _TAPS.get(key)
    if t is None:
        t = _TAPS[key] = torch.tensor(list(ks), dtype=torch.long, device=device)
    return t


def conv3d_dgrad(gy, w, in_dims, stride, padding, dilation):
    """dL/dx (Vox, (B, in_dims, cin)) of y = conv3d(x, w) given gy = dL/dy (Vox).  w: (cout, cin, kx, ky, kz)."""
    cout, cin = w.shape[:2]
    K = tuple(w.shape[2:])
    wt = w.detach().permute(1, 0, 2, 3, 4)                                  # (cin, cout, k): the transposed operator
    out = Vox.empty(gy.batch, in_dims, cin, gy.buf.device)
    axes = [_axis_phases(K[a], stride[a], padding[a], dilation[a]) for a in range(3)]
    if any(not taps for ax in axes for _, taps, _, _ in ax) or out.cs != cin:
        out.buf.zero_()                                                    # empty phases / channel pad
    for (rx, tx, dx_, px), (ry, ty, dy_, py), (rz, tz, dz_, pz) in itertools.product(*axes):
        if not (tx and ty and tz):
            continue
        sub = wt.index_select(2, _taps(tx, wt.device)).index_select(3, _taps(ty, wt.device)).index_select(4, _taps(tz, wt.device))
        n_pos = tuple((in_dims[a] - r + stride[a] - 1) // stride[a] for a, r in enumerate((rx, ry, rz)))
        if min(n_pos) <= 0:
            continue
        hip.conv3d(gy, hip.pack_weights(sub), None, cin, tuple(sub.shape[2:]), out, dilation=(dx_, dy_, dz_),
                   padding=(px, py, pz), out_pos=n_pos, o_stride=tuple(stride), o_off=(rx, ry, rz), cin=cout)
    return out


def _out_dims(dims, K, stride, padding, dilation):
    return tuple((n + 2 * p - d * (k - 1) - 1) // s + 1 for n, k, s, p, d in zip(dims, K, stride, padding, dilation))


class _Conv3dFn(torch.autograd.Function):
    # under torch.autocast the 3-D convolutions stay in float32 (exact-fp32 MFMA; the tensors are cast on entry)
    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, x, w, b, stride, padding, dilation):
        xv = _to_vox(x.detach())
        cout, cin = w.shape[:2]
        K = tuple(w.shape[2:])
        out = Vox.empty(xv.batch, _out_dims(xv.dims, K, stride, padding, dilation), cout, x.device)
        if out.cs != cout:
            out.buf.zero_()
        bias = None
        if b is not None:
            bias = torch.zeros(hip.round_up(cout, 32), device=x.device)
            bias[:cout] = b.detach()
        hip.conv3d(xv, hip.pack_weights(w.detach()), bias, cout, K, out, stride=stride, dilation=dilation,
                   padding=padding, cin=cin)
        ctx.save_for_backward(xv.buf, w)
        ctx.geom = (xv.C, xv.coff, stride, padding, dilation, b is not None)
        return out.ncdhw()

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, gy):
        xbuf, w = ctx.saved_tensors
        C, coff, stride, padding, dilation, has_bias = ctx.geom
        xv = Vox(xbuf, C, coff)
        gyv = _to_vox(gy.float())
        cout, cin = w.shape[:2]
        K = tuple(w.shape[2:])
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = conv3d_dgrad(gyv, w, xv.dims, stride, padding, dilation).ncdhw()
        if ctx.needs_input_grad[1]:
            dw = hip.conv3d_wgrad(xv, gyv, cin, cout, K, stride, dilation, padding)
        if has_bias and ctx.needs_input_grad[2]:
            db = gyv.buf.reshape(-1, gyv.cs)[:, gyv.coff:gyv.coff + cout].sum(0)
        return dx, dw, db, None, None, None


class _ConvTranspose3dFn(torch.autograd.Function):
    """y = conv_transpose3d(x, w (cin, cout, k)) == the data gradient of the convolution whose weight is w."""

    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, x, w, b, stride, padding, output_padding, dilation):
        xv = _to_vox(x.detach())
        cin, cout = w.shape[:2]
        K = tuple(w.shape[2:])
        dims = tuple((n - 1) * s - 2 * p + d * (k - 1) + op + 1
                     for n, k, s, p, d, op in zip(xv.dims, K, stride, padding, dilation, output_padding))
        out = conv3d_dgrad(xv, w, dims, stride, padding, dilation)
        y = out.ncdhw()
        if b is not None:
            y = y + b.detach().view(1, -1, 1, 1, 1)
        ctx.save_for_backward(xv.buf, w)
        ctx.geom = (xv.C, xv.coff, stride, padding, dilation, b is not None)
        return y

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, gy):
        xbuf, w = ctx.saved_tensors
        C, coff, stride, padding, dilation, has_bias = ctx.geom
        xv = Vox(xbuf, C, coff)
        gyv = _to_vox(gy.float())
        cin, cout = w.shape[:2]
        K = tuple(w.shape[2:])
        dx = dw = db = None
        if ctx.needs_input_grad[0]:                     # the convolution (weight w: "cout" = cin of the transpose)
            o = Vox.empty(xv.batch, xv.dims, cin, gy.device)
            if o.cs != cin:
                o.buf.zero_()
            hip.conv3d(gyv, hip.pack_weights(w.detach()), None, cin, K, o, stride=stride, dilation=dilation,
                       padding=padding, out_pos=xv.dims, cin=cout)
            dx = o.ncdhw()
        if ctx.needs_input_grad[1]:                     # roles swapped: gy is the conv's input, x its output gradient
            dw = hip.conv3d_wgrad(gyv, xv, cout, cin, K, stride, dilation, padding)
        if has_bias and ctx.needs_input_grad[2]:
            db = gyv.buf.reshape(-1, gyv.cs)[:, gyv.coff:gyv.coff + cout].sum(0)
        return dx, dw, db, None, None, None, None


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
