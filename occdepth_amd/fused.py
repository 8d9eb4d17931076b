"""Eval-mode execution plans: nn.Conv3d / ConvTranspose3d / AvgPool3d+Conv3d (+ BatchNorm3d)
folded into one packed-weight launch of the HIP implicit-GEMM kernel (K2).

A plan owns no parameters: it references the live nn modules, folds BatchNorm running
statistics into (weight scale, bias) and packs on first use; the packed image is rebuilt
whenever a source tensor is replaced or modified in place (load_state_dict, optimizer step),
detected through (data_ptr, _version).
"""
import os

import torch

from . import hip
from .hip import ACT_NONE, ACT_RELU, ACT_RELU_PRE, ACT_SIGMOID, Vox  # noqa: F401 (re-exported)


_warned_eval_with_grad = False


def needs_autograd(module):
    """True when the differentiable ATen graph must be built (training, or eval with gradients enabled);
    the HIP eval path is forward-only and is taken under torch.no_grad() / torch.inference_mode()."""
    if module.training:
        return True
    if torch.is_grad_enabled():
        global _warned_eval_with_grad
        if not _warned_eval_with_grad:
            _warned_eval_with_grad = True
            import warnings
            warnings.warn("occdepth_amd: eval-mode forward with autograd enabled runs the differentiable ATen path; "
                          "wrap inference in torch.no_grad() to take the HIP eval path", stacklevel=3)
        return True
    return False


def _triple(v):
    return tuple(v) if isinstance(v, (tuple, list)) else (int(v),) * 3


def on_gpu(t):
    """Gate of the 2-D eval fast paths (fused HIP kernels instead of ATen).  A function, not `t.is_cuda` inline, so that the
    CPU test-suite can drive the very same host logic through its torch emulation of the kernels (tests/emu.py)."""
    return t.is_cuda


def _stamp(*mods):
    key = []
    for m in mods:
        if m is None:
            continue
        for t in list(m.parameters(recurse=False)) + list(m.buffers(recurse=False)):
            key.append((t.data_ptr(), t._version, t.device))
    return tuple(key)


def _bn_affine(bn, cout, device):
    """BatchNorm (eval) as y = x * scale + shift."""
    if bn is None:
        return None, torch.zeros(cout, device=device)
    inv = torch.rsqrt(bn.running_var.float() + bn.eps)
    g = bn.weight.float() if bn.weight is not None else torch.ones_like(inv)
    b = bn.bias.float() if bn.bias is not None else torch.zeros_like(inv)
    scale = g * inv
    return scale, b - bn.running_mean.float() * scale


def bn_affine_cached(bn):
    """(scale, shift) of an eval-mode BatchNorm, cached on the module until its tensors change."""
    key = _stamp(bn)
    cached = getattr(bn, "_occd_affine", None)
    if cached is None or cached[0] != key:
        scale, shift = _bn_affine(bn, bn.num_features, bn.running_mean.device)
        cached = (key, scale.contiguous(), shift.contiguous())
        bn._occd_affine = cached
    return cached[1], cached[2]


def wino_fused_operands(owner, conv, bn):
    """(packed Winograd-domain weights of K10 with the BatchNorm scale folded in, shift) of a 3x3 convolution + BatchNorm,
    cached on `owner` until a source tensor changes ((data_ptr, _version) stamp)."""
    key = _stamp(conv, bn)
    cache = owner.__dict__.setdefault("_fused_cache", {})
    hit = cache.get(id(conv))
    if hit is None or hit[0] != key:
        scale, shift = bn_affine_cached(bn)
        if conv.bias is not None:
            shift = shift + scale * conv.bias.detach().float()
        hit = (key, hip.wino_pack_weights(conv.weight, scale), shift.contiguous())
        cache[id(conv)] = hit
    return hit[1:]


def _pad_bias(bias, cout):
    out = torch.zeros(hip.round_up(cout, 32), device=bias.device, dtype=torch.float32)
    out[:cout] = bias
    return out


# The 3-way bf16 split (x = hi + mid + lo, six bf16 MFMAs per K step, float32-level accuracy on the bf16 matrix pipe):
#   OCCDEPTH_BF16X3=head  (default since round 4) the full-resolution head convolutions (<= 32 -> <= 32 channels, 3x3x3,
#                         dilation 1 / 2 / 3, Z = 32) on K2s3, the sliding-window form of the split (csrc/conv3d_c32p.hip):
#                         0.51 ms per launch against 0.90 ms for the exact-fp32 K2s, error against float64 no larger than
#                         K2s' own (tests/test_bf16_conv.py::test_conv3d_slide_x3_head_kernel, profiles/r04_head_x3_ab.txt);
#                         also (see below) the long-K 3x3x3 convolutions of small volumes and the merged phase launches of the
#                         transposed convolutions on K2b's split form; everything else exact fp32
#   OCCDEPTH_BF16X3=0     exact-fp32 MFMA everywhere (v_mfma_f32_32x32x2_f32): the bit-for-bit round-3 path
#   OCCDEPTH_BF16X3=1     additionally every other large dilation-1 K2 launch on the generic K2b skeleton with the split
#                         (csrc/conv3d_bf16.hip; the round-3 experiment)
def _parse_bf16x3(v):
    v = (v or "0").strip().lower()
    if v in ("0", "", "off", "false"):
        return False
    if v in ("1", "all", "on", "true"):
        return "all"
    if v == "head":
        return "head"
    raise ValueError(f"OCCDEPTH_BF16X3={v!r}: expected 0, head or 1")


BF16X3 = _parse_bf16x3(os.environ.get("OCCDEPTH_BF16X3", "head"))


def set_bf16x3(on):
    """False / 'head' / 'all' (True = 'all'); plans re-pack their weights on the next call."""
    global BF16X3
    BF16X3 = "all" if on is True else _parse_bf16x3(str(on)) if on else False


class _DualW:
    """Both weight images of a plan while the split is on: the hi | mid | lo one for the launches the split kernels take,
    the exact-fp32 one for the rest."""

    def __init__(self, f32, x3):
        self.f32, self.x3 = f32, x3


BF16X3_DILATED = os.environ.get("OCCDEPTH_BF16X3_DILATED", "0") == "1"


def _head_weight(w, layout):
    return layout == 0 and w.dim() == 5 and tuple(w.shape[2:]) == (3, 3, 3) and w.shape[0] <= 32 and 24 < w.shape[1] <= 32


# Also part of the default ("head") mode since late round 4: the 3x3x3 convolutions with a long K on a SMALL volume (the
# ASPP branches and `mega_context` of the CRP at the 1/8 level: 256 -> 256 / 512 on 32x32x4 = 4096 voxels, K = 27 x 256).
# Too few output tiles to fill 256 CUs, so K2 runs them on its in-workgroup split-K variant at ~96 TF/s of the exact-fp32
# instruction; K2b's split form with the same 4-way split-K (variant 8: M32 x N128, 16 waves) does the MFMA work at 6/16 of
# that instruction's time.  OCCDEPTH_BF16X3_SMALLVOL=0 keeps them on K2.
BF16X3_SMALLVOL = os.environ.get("OCCDEPTH_BF16X3_SMALLVOL", "1") == "1"
SMALLVOL_MAX_VOXELS = 8192
# (1x1x1 convolutions and the CRP products sigmoid(P_logits) @ mega can take the same kernel -- input sigmoid applied while
#  staging -- but do not gain: 512 -> 512 on 4096 rows 78 us against K2's 61, 256 -> 512 27 against 28, 2304 -> 256 75 against
#  78: two MFMA steps per staged chunk.  OCCDEPTH_BF16X3_SMALLVOL_K1=1 routes them there for A/B.)
SMALLVOL_KERNELS = ((3, 3, 3), (1, 1, 1)) if os.environ.get("OCCDEPTH_BF16X3_SMALLVOL_K1", "0") == "1" else ((3, 3, 3),)


def _smallvol_weight(w, layout):
    if not BF16X3_SMALLVOL:
        return False
    if layout == 2:          # a dynamic (K, N) row-GEMM operand: the CRP products sigmoid(P_logits) @ mega
        return (1, 1, 1) in SMALLVOL_KERNELS and w.dim() == 2 and w.shape[0] >= 128 and w.shape[0] % 16 == 0 and w.shape[1] % 128 == 0
    return (layout == 0 and w.dim() == 5 and tuple(w.shape[2:]) in SMALLVOL_KERNELS and w.shape[1] >= 128
            and w.shape[1] % 16 == 0 and w.shape[0] % 128 == 0)


def _pack_w(w, scale=None, layout=0):
    if BF16X3 == "all" or (BF16X3 == "head" and (_head_weight(w, layout) or _smallvol_weight(w, layout))):
        return _DualW(hip.pack_weights(w, scale, layout), hip.pack_weights_bf16(w, scale, layout, split3=True))
    return hip.pack_weights(w, scale, layout)


def _smallvol_eligible(x, wpk, cout, kernel, out, **kw):
    if not BF16X3_SMALLVOL or tuple(kernel) not in SMALLVOL_KERNELS or kw.get("cin") not in (None, x.C) or x.C < 128 or cout % 128:
        return False
    if x.cs % 8 or x.coff % 8 or x.buf.dtype != torch.float32:
        return False
    pos = kw.get("out_pos")
    if pos is None:
        st, dl, pd = kw.get("stride", (1, 1, 1)), kw.get("dilation", (1, 1, 1)), kw.get("padding", (0, 0, 0))
        pos = tuple((n + 2 * p - d * (k - 1) - 1) // s + 1 for n, p, d, s, k in zip(x.dims, pd, dl, st, kernel))
    return x.batch * pos[0] * pos[1] * pos[2] <= SMALLVOL_MAX_VOXELS and kw.get("tile_hint", 0) == 0


def _conv3d(x, wpk, bias, cout, kernel, out, **kw):
    if isinstance(wpk, _DualW):
        if hip.c32x3_eligible(x, cout, kernel, out, **kw):        # K2s3 (all three dilations)
            return hip.conv3d_bf16(x, wpk.x3, bias, cout, kernel, out, split3=True, **kw)
        if _smallvol_eligible(x, wpk, cout, kernel, out, **kw):   # K2b split form, in-workgroup split-K
            return hip.conv3d_bf16(x, wpk.x3, bias, cout, kernel, out, split3=True, **kw)
        aligned = x.cs % 8 == 0 and x.coff % 8 == 0          # K2b stages 8 channels per 16-byte LDS chunk
        dil = tuple(kw.get("dilation", (1, 1, 1)))
        big = kernel[0] * kernel[1] * kernel[2] >= 8
        if BF16X3 == "all" and aligned and big and (dil == (1, 1, 1) or BF16X3_DILATED) and kw.get("cin") is None and \
                kw.get("act_in", ACT_NONE) in (ACT_NONE, ACT_RELU):
            return hip.conv3d_bf16(x, wpk.x3, bias, cout, kernel, out, split3=True, **kw)
        wpk = wpk.f32
    return hip.conv3d(x, wpk, bias, cout, kernel, out, **kw)


class ConvPlan:
    """conv (+ optional conv bias) + optional BatchNorm3d as one K2 launch."""

    def __init__(self, conv, bn=None, pool=None):
        self.conv, self.bn = conv, bn
        self.pool = _triple(pool) if pool is not None else None  # AvgPool3d(k=stride=pool) in front of a 1x1x1 conv
        self._key = None
        self._wpk = self._bias = None

    @property
    def cout(self):
        return self.conv.out_channels

    def _prepare(self):
        key = (_stamp(self.conv, self.bn), BF16X3)
        if key == self._key:
            return
        w = self.conv.weight.detach().float()
        dev = w.device
        cout = self.cout
        scale, shift = _bn_affine(self.bn, cout, dev)
        if self.conv.bias is not None:
            cb = self.conv.bias.detach().float()
            shift = shift + (cb * scale if scale is not None else cb)
        if self.pool is not None:
            # mean over the pooling window then 1x1x1 conv == conv with k = stride = window, w / |window|
            kx, ky, kz = self.pool
            w = (w / float(kx * ky * kz)).expand(-1, -1, kx, ky, kz).contiguous()
        self._wpk = _pack_w(w, scale)
        self._bias = _pad_bias(shift, cout)
        self._wrows = None
        if self.pool is None and tuple(w.shape[2:]) == (1, 1, 1) and _triple(self.conv.stride) == (1, 1, 1):
            w2 = w.reshape(cout, -1)
            if ROWS_GEMM and w2.is_cuda and w2.shape[1] % 16 == 0 and cout % 8 == 0:
                self._wrows = hip.RowsGemmWeights((w2 * scale.view(-1, 1) if scale is not None else w2).t().contiguous())
        self._key = key

    def geometry(self):
        if self.pool is not None:
            return self.pool, self.pool, (1, 1, 1), (0, 0, 0)
        c = self.conv
        return _triple(c.kernel_size), _triple(c.stride), _triple(c.dilation), _triple(c.padding)

    def out_dims(self, dims):
        k, s, d, p = self.geometry()
        return tuple((n + 2 * pp - dd * (kk - 1) - 1) // ss + 1 for n, kk, ss, dd, pp in zip(dims, k, s, d, p))

    def __call__(self, x, out=None, res1=None, res2=None, act_in=ACT_NONE, act_out=ACT_NONE):
        self._prepare()
        k, s, d, p = self.geometry()
        if out is None:
            out = Vox.empty(x.batch, self.out_dims(x.dims), self.cout, x.buf.device)
        if (ROWS_GEMM and self._wrows is not None and res2 is None and x.buf.is_cuda and _rows_gemm_wins(x, self._wrows.K)
                and hip.rows_gemm_supported(self._wrows.K, self.cout, x, out, res1)):
            return hip.rows_gemm(x, self._wrows, out, bias=self._bias, res=res1, act_in=act_in, act_out=act_out)
        return _conv3d(x, self._wpk, self._bias, self.cout, k, out, stride=s, dilation=d, padding=p,
                          res1=res1, res2=res2, act_in=act_in, act_out=act_out)


class DerivedConvPlan:
    """A K2 launch whose weights are a function of several modules' parameters (slices / concatenations),
    e.g. the cascade head's merged `conv_classes[:, :planes] | occ_classes` convolution.

    `derive()` returns (weight (Cout, Cin, kx, ky, kz), bias (Cout,) or None); geometry is fixed."""

    def __init__(self, modules, derive, padding=(1, 1, 1), dilation=(1, 1, 1), stride=(1, 1, 1)):
        self.modules, self.derive = list(modules), derive
        self.padding, self.dilation, self.stride = padding, dilation, stride
        self._key = None
        self._wpk = self._bias = None
        self.cout = self.kernel = None

    def _prepare(self):
        key = (_stamp(*self.modules), BF16X3)
        if key == self._key:
            return
        w, b = self.derive()
        w = w.detach().float().contiguous()
        self.cout, self.kernel = w.shape[0], tuple(w.shape[2:])
        self._wpk = _pack_w(w)
        self._bias = _pad_bias(b.detach().float(), self.cout) if b is not None else None
        self._key = key

    def __call__(self, x, out=None, res1=None, res2=None, act_in=ACT_NONE, act_out=ACT_NONE, out_cs=None):
        self._prepare()
        if out is None:
            dims = tuple((n + 2 * p - d * (k - 1) - 1) // s + 1
                         for n, k, s, d, p in zip(x.dims, self.kernel, self.stride, self.dilation, self.padding))
            out = Vox.empty(x.batch, dims, self.cout, x.buf.device, cs=out_cs)
        return _conv3d(x, self._wpk, self._bias, self.cout, self.kernel, out, stride=self.stride,
                          dilation=self.dilation, padding=self.padding, res1=res1, res2=res2, act_in=act_in,
                          act_out=act_out)


# the 8 sub-pixel phases of a stride-2 transposed convolution as one launch (occd_conv3d_fwd_phases); 0 = eight launches
PHASES_ONE_LAUNCH = os.environ.get("OCCDEPTH_PHASES_ONE_LAUNCH", "1") == "1"
# ... and that one launch on K2b with the 3-way bf16 split (occd_conv3d_bf16_fwd_phases) instead of the exact-fp32 K2, while the
# split is on at all (OCCDEPTH_BF16X3 != 0): float32-level accuracy, 6/16 of the fp32 instruction's MFMA time.  0 = K2.
PHASES_X3 = os.environ.get("OCCDEPTH_PHASES_X3", "1") == "1"


class ConvTransposePlan:
    """ConvTranspose3d(k=3, s=2, p=1, output_padding=1) (+BN) as 8 sub-pixel phase convolutions,
    or ConvTranspose3d(k=3, s=1, p=1) (+BN) as one flipped convolution."""

    def __init__(self, convt, bn=None):
        self.convt, self.bn = convt, bn
        ks, st = _triple(convt.kernel_size), _triple(convt.stride)
        pd, op = _triple(convt.padding), _triple(convt.output_padding)
        if ks != (3, 3, 3) or pd != (1, 1, 1) or _triple(convt.dilation) != (1, 1, 1):
            raise NotImplementedError("only the k3/p1 transposed convolutions of OccDepth are planned")
        if st == (2, 2, 2) and op == (1, 1, 1):
            self.up = 2
        elif st == (1, 1, 1) and op == (0, 0, 0):
            self.up = 1
        else:
            raise NotImplementedError(f"ConvTranspose3d stride {st} output_padding {op}")
        self._key = None
        self._phases = None
        self._bias = None

    @property
    def cout(self):
        return self.convt.out_channels

    def _prepare(self):
        key = (_stamp(self.convt, self.bn), BF16X3, PHASES_X3)
        if key == self._key:
            return
        x3 = bool(BF16X3) and PHASES_X3 and PHASES_ONE_LAUNCH

        def pack(w, scale):      # both images of a phase's tap subset when the merged launch may take the split kernel
            if x3:
                return _DualW(hip.pack_weights(w, scale), hip.pack_weights_bf16(w, scale, split3=True))
            return _pack_w(w, scale)
        wt = self.convt.weight.detach().float()  # (Cin, Cout, 3, 3, 3)
        dev = wt.device
        scale, shift = _bn_affine(self.bn, self.cout, dev)
        if self.convt.bias is not None:
            cb = self.convt.bias.detach().float()
            shift = shift + (cb * scale if scale is not None else cb)
        w = wt.permute(1, 0, 2, 3, 4)  # (Cout, Cin, k, k, k), k indexes the scatter offset o = s*i - 1 + k
        phases = []
        if self.up == 1:
            phases.append(((0, 0, 0), (3, 3, 3), (1, 1, 1), _pack_w(w.flip(2, 3, 4).contiguous(), scale)))
        else:
            # even outputs (o = 2j) see only k=1 at i=j; odd outputs (o = 2j+1) see k=2 at i=j and k=0 at i=j+1
            taps = ([1], [2, 0])
            for px in (0, 1):
                for py in (0, 1):
                    for pz in (0, 1):
                        sub = w[:, :, taps[px]][:, :, :, taps[py]][:, :, :, :, taps[pz]].contiguous()
                        phases.append(((px, py, pz), tuple(sub.shape[2:]), (0, 0, 0), pack(sub, scale)))
        self._phases = phases
        self._bias = _pad_bias(shift, self.cout)
        self._key = key

    def out_dims(self, dims):
        return tuple(n * self.up for n in dims)

    def __call__(self, x, out=None, res1=None, act_out=ACT_NONE):
        self._prepare()
        if out is None:
            out = Vox.empty(x.batch, self.out_dims(x.dims), self.cout, x.buf.device)
        def phase(off, kern, pad, wpk):
            return lambda: _conv3d(x, wpk, self._bias, self.cout, kern, out, padding=pad, res1=res1, act_out=act_out,
                                   out_pos=x.dims, o_stride=(self.up,) * 3, o_off=off)

        if PHASES_ONE_LAUNCH and len(self._phases) > 1 and BF16X3 and PHASES_X3 and x.buf.dtype == torch.float32 and \
                x.cs % 8 == 0 and x.coff % 8 == 0 and all(isinstance(ph[3], _DualW) for ph in self._phases):
            # one K2b launch for the 8 phases, 3-way bf16 split
            return hip.conv3d_phases(x, [(wpk.x3, kern, off) for off, kern, _, wpk in self._phases], self._bias, self.cout, out,
                                     res1=res1, act_out=act_out, out_pos=x.dims, o_stride=(self.up,) * 3, split3=True)
        if PHASES_ONE_LAUNCH and len(self._phases) > 1 and BF16X3 != "all":
            # one K2 launch for the 8 phases (phase = low bits of blockIdx.y, heaviest tap subset first); exact-fp32 images
            return hip.conv3d_phases(x, [(wpk.f32 if isinstance(wpk, _DualW) else wpk, kern, off) for off, kern, _, wpk in self._phases],
                                     self._bias, self.cout, out, res1=res1, act_out=act_out, out_pos=x.dims,
                                     o_stride=(self.up,) * 3)
        thunks = [phase(*ph) for ph in self._phases]
        if x.buf.is_cuda and len(thunks) > 1:
            run_parallel(thunks)                     # the phases write disjoint voxels of `out`
        else:
            for th in thunks:
                th()
        return out


# K15 (csrc/rows_gemm.hip): 1x1x1 convolutions and the CRP product as a row GEMM whose data operand never touches LDS.
# OFF by default -- measured round 3 (sessions 21-23, profiles/r03_rows_gemm_ab.txt): 512>512 on 4096 rows 55-57 us against
# K2's 63, 256>512 30 against 27, 2304>256 205-216 against 78; the config-2 frame 25.28-25.44 ms against 25.09-25.21.  With
# 256 workgroups on 256 CUs nothing hides the L2 latency of the per-stage weight fetch (3.5 us per 32-k stage for 0.85 us of
# MFMA work); K2's in-workgroup split-K (16 waves) does.  OCCDEPTH_ROWS_GEMM=1 enables it for small volumes / long K.
ROWS_GEMM = os.environ.get("OCCDEPTH_ROWS_GEMM", "0") == "1"


def _rows_gemm_wins(x, K):
    rows = x.batch * x.dims[0] * x.dims[1] * x.dims[2]
    return rows <= 65536 or K >= 256


def pack_rows(b_rows):
    """(K, N) row-major device matrix -> the B operand of `gemm_rows`: the matrix itself for K15, else K2's packed image
    (occd_pack_weights layout 2)."""
    if ROWS_GEMM and b_rows.is_cuda and b_rows.dtype == torch.float32 and b_rows.shape[0] % 16 == 0 and b_rows.shape[1] % 8 == 0:
        return hip.RowsGemmWeights(b_rows), tuple(b_rows.shape)
    return _pack_w(b_rows, layout=2), tuple(b_rows.shape)


def gemm_rows(a, b_rows, out, act_in=ACT_NONE):
    """out[row, :] = act_in(a[row, :K]) @ b_rows[:K, :N] for channels-last Vox a/out (CRP bmm).

    b_rows is a dense (K, N) row-major device matrix (packed on the fly) or the result of `pack_rows`."""
    wpk, (K, N) = b_rows if isinstance(b_rows, tuple) else pack_rows(b_rows)
    if isinstance(wpk, hip.RowsGemmWeights):
        if not hip.rows_gemm_supported(K, N, a, out):
            raise RuntimeError("gemm_rows: operands do not fit the row-GEMM kernel")
        return hip.rows_gemm(a, wpk, out, act_in=act_in)
    return _conv3d(a, wpk, None, N, (1, 1, 1), out, act_in=act_in, cin=K)


# Independent small launches (the 8 sub-pixel phases of a transposed convolution, the ASPP's dilation branches, the CRP's
# relation branches) CAN be issued round-robin on a few side streams forked from -- and joined back into -- the current
# stream (inside a hipGraph capture the fork / join become graph edges); the thunks must not allocate.  Measured round 3
# and NOT adopted: the config-2 frame went from 25.15 ms (one stream) to 25.25 / 26.04 / 25.73 ms with 2 / 3 / 4 streams --
# a branching graph costs more in cross-queue barriers than the overlapped tails give back.  Default 1 = sequential.
PARALLEL_STREAMS = int(os.environ.get("OCCDEPTH_PARALLEL_STREAMS", "1"))
_side_streams = {}


def run_parallel(thunks, width=None):
    width = PARALLEL_STREAMS if width is None else width
    if width < 2 or len(thunks) < 2 or not torch.cuda.is_available():
        for th in thunks:
            th()
        return
    cur = torch.cuda.current_stream()
    key = (cur.device, width)
    if key not in _side_streams:
        _side_streams[key] = [torch.cuda.Stream(device=cur.device) for _ in range(width)]
    pool = _side_streams[key][:min(width, len(thunks))]
    fork = torch.cuda.Event()
    fork.record(cur)
    for s in pool:
        s.wait_event(fork)
    for i, th in enumerate(thunks):
        with torch.cuda.stream(pool[i % len(pool)]):
            th()
    for s in pool:
        cur.wait_stream(s)


def as_vox(x):
    """(B, C, X, Y, Z) tensor -> Vox; zero-copy when it already is a full channels-last buffer of ours."""
    if isinstance(x, Vox):
        return x
    if x.dim() != 5:
        raise RuntimeError("expected a (B, C, X, Y, Z) tensor")
    if not x.is_cuda:
        raise RuntimeError("the HIP path needs GPU tensors (there is no CPU fallback)")
    x = x.float()
    cl = x.permute(0, 2, 3, 4, 1)
    if cl.is_contiguous() and x.shape[1] % 8 == 0:
        return Vox(cl, x.shape[1])
    return Vox.from_ncdhw(x)
