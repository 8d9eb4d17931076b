"""In-repo EfficientNet (tf_efficientnet_b{3,4,5,7}_ns topology) for the 2-D encoder.

The reference pulls `rwightman/gen-efficientnet-pytorch` through torch.hub
(occdepth/models/unet2d.py:238-240); that dependency is not vendored and there is no network,
so the architecture is restated here with geffnet's module / parameter names (conv_stem, bn1,
act1, blocks.{stage}.{block}.{conv_pw,bn1,conv_dw,bn2,se.conv_reduce,se.conv_expand,conv_pwl,bn3},
conv_head, bn2, act2, global_pool, classifier) so hub checkpoints load by key.  TF "SAME"
padding, swish, SE on the block input width, BN eps 1e-3.  Eval path on the GPU: every MBConv block is 4 launches of the
in-repo kernels (K11 / K11s pointwise GEMMs with BN / swish / SE gate / skip fused, LDS-staged depthwise + SE pooling, SE
gate); the expand convolutions of the 1/16 and 1/32 stages and the stem stay on the library (measured faster there).
Training path: depthwise convolutions and swish on HIP kernels with hand-written backward, the rest on ATen.
Parity of this file is UNPINNED (no reference copy of geffnet exists here).
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

import os

from .. import hip
from .. import fused as _fused
from ..bn import bn_act
from ..fused import _stamp, bn_affine_cached, needs_autograd

# K11 / SE-fusion path of the eval forward (OCCDEPTH_PW_FUSED=0 restores the MIOpen / rocBLAS 1x1 convolutions for A/B)
PW_FUSED = os.environ.get("OCCDEPTH_PW_FUSED", "1") == "1"
# Below this many pixels (B * H * W) the blocks fall back to rocBLAS / ATen glue.  0 = never: the library picks the
# streaming K11 variants where pixels are many and the split-K ones (K11s) on the 1/16 and 1/32 levels (K up to 3840,
# < 2000 pixels per view), so every MBConv block of the eval path is 4 launches.  Kept as an A/B switch.
PW_MIN_PIXELS = int(os.environ.get("OCCDEPTH_PW_MIN_PIXELS", "0"))          # B * H * W


def pw_wins(x):
    return PW_FUSED and x.shape[0] * x.shape[2] * x.shape[3] >= PW_MIN_PIXELS


# Expand convolutions (short K, up to 3840 couts, no gate) on maps of fewer pixels than this go to K16 (hip.matmul: the LDS-tiled
# GEMM with the 3-way bf16 split, BatchNorm shift + swish in its epilogue; round 2-3: the library GEMM + one BN / swish pass) --
# since round 4 at EVERY resolution: with 64 x 64 tiles for short K (csrc/gemm_x3.hip) K16 runs 48 -> 288 on 2 x 28365 pixels in
# 34 us against 61 for K11, 80 -> 480 on 2 x 7191 in 21 against 34, 32 -> 192 on 2 x 112850 in 55 against 104
# (profiles/r04_gemm_x3_v5_shortk.txt).  The project convolutions (long K, SE gate + BN + skip fused) stay on K11 / K11s.
# OCCDEPTH_PW_EXPAND_LIB_BELOW=14000 restores the round-3 split (K11 on the high-resolution stages).
PW_EXPAND_LIB_BELOW = int(os.environ.get("OCCDEPTH_PW_EXPAND_LIB_BELOW", str(1 << 40)))
# plain 1x1 convolutions outside the blocks (conv_head: 640 -> 2560 on the 1/32 map) below this many pixels: library GEMM
PW_PLAIN_LIB_BELOW = 14000


def expand_on_library(x):
    # (the exact-fp32 path, OCCDEPTH_GEMM_X3=0: K16 is off and hip.matmul is the library GEMM -- the round-3 split)
    below = PW_EXPAND_LIB_BELOW if hip.GEMM_X3 else min(PW_EXPAND_LIB_BELOW, PW_PLAIN_LIB_BELOW)
    return x.shape[1] % 8 == 0 and x.shape[0] * x.shape[2] * x.shape[3] < below


def pw_operands(owner, conv, bn=None):
    """(packed K11 weights with the BatchNorm scale folded in, shift) of a 1x1 convolution (+ BatchNorm), cached on
    `owner` until a source tensor changes (load_state_dict, optimizer step: (data_ptr, _version) stamp)."""
    key = _stamp(conv, bn)
    cache = owner.__dict__.setdefault("_pw_cache", {})
    hit = cache.get(id(conv))
    if hit is None or hit[0] != key:
        scale = shift = None
        if bn is not None:
            scale, shift = bn_affine_cached(bn)
        if conv.bias is not None:
            b = conv.bias.detach().float()
            shift = b * scale + shift if bn is not None else b
        hit = (key, hip.pw_pack_weights(conv.weight, scale), shift.contiguous() if shift is not None else None)
        cache[id(conv)] = hit
    return hit[1], hit[2]


def gemm_operands(owner, conv, bn):
    """(W * BatchNorm scale as a (Cout, Cin) matrix, shift) of a 1x1 convolution + BatchNorm for K16 (hip.matmul), cached on
    `owner` until a source tensor changes."""
    key = _stamp(conv, bn)
    cache = owner.__dict__.setdefault("_gemm_cache", {})
    hit = cache.get(id(conv))
    if hit is None or hit[0] != key:
        scale, shift = bn_affine_cached(bn)
        w = conv.weight.detach().float().flatten(1) * scale.view(-1, 1)
        if conv.bias is not None:
            shift = conv.bias.detach().float() * scale + shift
        hit = (key, hip.matmul_operand(w, "a"), shift.contiguous())
        cache[id(conv)] = hit
    return hit[1], hit[2]


def expand_gemm(owner, conv, bn, x, act, pad_out=False):
    """act(bn(conv1x1(x))) of the few-pixel stages: ONE K16 launch (BatchNorm folded, shift + swish in the epilogue) where
    the library path is a GEMM + an elementwise pass; falls back to that path when K16 does not apply."""
    B, C, H, W = x.shape
    if hip.GEMM_X3 and C % 8 == 0 and H * W >= 4 and x.is_contiguous():
        w, shift = gemm_operands(owner, conv, bn)
        # (the expanded planes on a 128-byte pitch: written 2x faster; the depthwise kernel that consumes them takes the stride)
        out = hip.padded_rows((B, w[0].shape[0], H * W), x.device) if pad_out else None
        return hip.matmul(w, x.view(B, C, H * W), bias=shift, act=act, out=out).view(B, -1, H, W)
    return hip.affine_act(F.conv2d(x, conv.weight), *bn_affine_cached(bn), act)


# Project convolutions (the squeeze-excite gate on the input channels, BatchNorm, the block's skip) on K16 as well where it
# wins: the gate is applied to the B rows while they are staged (occd_gemm_args.scale_k), the skip in the epilogue (.res); long K
# on few output tiles takes K16's in-workgroup split-K form.  Measured per launch, config 2, K16 against K11 / K11s
# (profiles/r04_gemm_x3_v6_project.txt): 288 -> 48 on 2 x 28365 pixels 30 against 54 us, 480 -> 80 on 2 x 7191 28 against 40,
# 960 -> 160 on 2 x 1848 28 against 35, 1344 -> 224 39 against 44; but 2304 -> 384 on 2 x 468 56 against 37 and 3840 -> 640 88
# against 94 (a 64 x 64 tile spends more on its split / LDS round trip per step than the exact-fp32 wave on its MFMAs), and
# 32 -> 32 on 2 x 112850 64 against 49 (half of a 64-row tile is padding).  So: maps of >= 3000 pixels, >= 48 output channels.
PW_PROJECT_K16 = os.environ.get("OCCDEPTH_PW_PROJECT_K16", "1") == "1"
PW_PROJECT_K16_MIN_PIXELS = int(os.environ.get("OCCDEPTH_PW_PROJECT_K16_MIN_PIXELS", "3000"))
PW_PROJECT_K16_MIN_COUT = int(os.environ.get("OCCDEPTH_PW_PROJECT_K16_MIN_COUT", "48"))


# Round 6: the project convolutions of the FEW-pixel stages (1/16, 1/32: K = 960 ... 3840 on 2 x 1848 / 2 x 468 pixels) on K21
# (hip.gemm_x3_splitk: K cut over the grid, pre-split weights, deterministic second launch) -- they ran at 37 ... 57 TF/s on
# K16's in-workgroup split-K form / K11s (90 ... 230 output tiles for 256 CUs).  OCCDEPTH_PW_PROJECT_SPLITK=0 restores them.
PW_PROJECT_SPLITK = os.environ.get("OCCDEPTH_PW_PROJECT_SPLITK", "1") == "1"
PW_PROJECT_SPLITK_MIN_K = int(os.environ.get("OCCDEPTH_PW_PROJECT_SPLITK_MIN_K", "768"))
PW_PROJECT_SPLITK_MAX_PIXELS = int(os.environ.get("OCCDEPTH_PW_PROJECT_SPLITK_MAX_PIXELS", "8000"))


def splitk_operands(owner, conv, bn):
    """(GemmPacked image of W * BatchNorm scale, shift) for K21, cached on `owner` until a source tensor changes."""
    key = _stamp(conv, bn)
    cache = owner.__dict__.setdefault("_splitk_cache", {})
    hit = cache.get(id(conv))
    if hit is None or hit[0] != key:
        scale, shift = bn_affine_cached(bn)
        w = (conv.weight.detach().float().flatten(1) * scale.view(-1, 1)).contiguous()
        if conv.bias is not None:
            shift = conv.bias.detach().float() * scale + shift
        hit = (key, hip.GemmPacked(w, "a"), shift.contiguous())
        cache[id(conv)] = hit
    return hit[1], hit[2]


def project_conv(owner, conv, bn, y, gate, res):
    """bn(conv1x1(y * gate)) (+ res): K21 on the few-pixel long-K stages (two launches), else one launch of K16 (hip.matmul)
    where it applies, else K11 / K11s."""
    B, C, H, W = y.shape
    cout = conv.out_channels
    if (PW_PROJECT_SPLITK and hip.GEMM_X3 and y.is_cuda and C % 8 == 0 and C >= PW_PROJECT_SPLITK_MIN_K and H * W >= 4
            and B * H * W <= PW_PROJECT_SPLITK_MAX_PIXELS and y.is_contiguous() and gate.is_contiguous()
            and tuple(gate.shape) == (B, C) and (res is None or res.is_contiguous())):
        pa, shift = splitk_operands(owner, conv, bn)
        out = hip.gemm_x3_splitk(pa, y.view(B, C, H * W), bias=shift, k_scale=gate,
                                 res=res.view(B, cout, H * W) if res is not None else None)
        return out.view(B, cout, H, W)
    if (PW_PROJECT_K16 and hip.GEMM_X3 and C % 8 == 0 and B * H * W >= PW_PROJECT_K16_MIN_PIXELS and y.is_contiguous()
            and cout >= PW_PROJECT_K16_MIN_COUT and gate.is_contiguous() and tuple(gate.shape) == (B, C) and (res is None or res.is_contiguous())):
        w, shift = gemm_operands(owner, conv, bn)
        out = hip.matmul(w, y.view(B, C, H * W), bias=shift, k_scale=gate,
                         res=res.view(B, cout, H * W) if res is not None else None)
        return out.view(B, cout, H, W)
    wpk, shift = pw_operands(owner, conv, bn)
    return hip.conv1x1(y, wpk, cout, shift, None, gate=gate, res=res)


def _fast(x, module):
    """eval-mode CUDA tensors take the fused HIP elementwise / depthwise kernels (occdepth_amd/csrc/nchw2d.hip)."""
    return _fused.on_gpu(x) and not needs_autograd(module) and x.dtype == torch.float32

# (block type, repeats, kernel, stride, expand, channels) of EfficientNet-B0
_B0_STAGES = (("ds", 1, 3, 1, 1, 16), ("ir", 2, 3, 2, 6, 24), ("ir", 2, 5, 2, 6, 40), ("ir", 3, 3, 2, 6, 80),
              ("ir", 3, 5, 1, 6, 112), ("ir", 4, 5, 2, 6, 192), ("ir", 1, 3, 1, 6, 320))
# name -> (channel multiplier, depth multiplier)
_SCALING = {"tf_efficientnet_b3_ns": (1.2, 1.4), "tf_efficientnet_b4_ns": (1.4, 1.8),
            "tf_efficientnet_b5_ns": (1.6, 2.2), "tf_efficientnet_b7_ns": (2.0, 3.1)}
_BN_EPS = 1e-3


def _round_channels(ch, mult, divisor=8):
    ch = ch * mult
    new = max(int(ch + divisor / 2) // divisor * divisor, divisor)
    return new + divisor if new < 0.9 * ch else new


class Conv2dSame(nn.Conv2d):
    """TensorFlow 'SAME' padding: output = ceil(input / stride), extra pad goes right/bottom."""

    def forward(self, x):
        if (self.kernel_size == (1, 1) and self.stride == (1, 1) and self.groups == 1 and _fast(x, self) and pw_wins(x)
                and x.shape[0] * x.shape[2] * x.shape[3] >= PW_PLAIN_LIB_BELOW):
            wpk, shift = pw_operands(self, self)            # e.g. conv_head: plain 1x1 convolution on the MFMA GEMM (K11)
            return hip.conv1x1(x, wpk, self.out_channels, shift)
        if (self.groups == self.in_channels == self.out_channels and self.groups > 1 and x.is_cuda and needs_autograd(self)
                and self.kernel_size in ((3, 3), (5, 5)) and self.stride in ((1, 1), (2, 2)) and self.bias is None
                and self.dilation == (1, 1) and x.dtype in (torch.float32, torch.bfloat16, torch.float16)
                and x.shape[0] * x.shape[1] <= 65535):
            # training: depthwise forward / data gradient / weight gradient on the HIP kernels (MIOpen's fp32 depthwise
            # path is the naive fallback: 17.5 ms of a config-2 step)
            return hip.dwconv2d_same_autograd(x, self.weight, self.stride[0])
        if needs_autograd(self) and hip.pw_conv_autograd_ok(self, x):
            # training: expand / project convolutions forward, data gradient and weight gradient on K16 / K16t
            return hip.pw_conv_autograd(x, self.weight)
        pads = []
        for size, k, s, d in zip(x.shape[-2:], self.kernel_size, self.stride, self.dilation):
            total = max((math.ceil(size / s) - 1) * s + (k - 1) * d + 1 - size, 0)
            pads = [total // 2, total - total // 2] + pads
        if any(pads):
            x = F.pad(x, pads)
        return F.conv2d(x, self.weight, self.bias, self.stride, (0, 0), self.dilation, self.groups)


# training on the GPU: swish as one forward + one backward launch (hip._SwishFn) instead of the six of the autograd graph
# of x * sigmoid(x); OCCDEPTH_TRAIN_FUSED_ACT=0 restores ATen for A/B
TRAIN_FUSED_ACT = os.environ.get("OCCDEPTH_TRAIN_FUSED_ACT", "1") == "1"


class Swish(nn.Module):
    def forward(self, x):
        if _fast(x, self):
            return hip.affine_act(x, None, None, "swish", out=torch.empty_like(x))
        if TRAIN_FUSED_ACT and x.is_cuda and x.dtype == torch.float32 and x.requires_grad and x.numel() >= 4:
            return hip.swish_autograd(x)
        return x * torch.sigmoid(x)


class SqueezeExcite(nn.Module):
    def __init__(self, chs, reduced):
        super().__init__()
        self.conv_reduce = nn.Conv2d(chs, reduced, 1, bias=True)
        self.act1 = Swish()
        self.conv_expand = nn.Conv2d(reduced, chs, 1, bias=True)

    def forward(self, x):
        if needs_autograd(self) and hip.squeeze_excite_autograd_ok(self, x):
            return hip.squeeze_excite_autograd(self, x)        # training on the GPU: 4 + 4 HIP launches (csrc/se2d.hip)
        g = self.conv_expand(self.act1(self.conv_reduce(x.mean((2, 3), keepdim=True))))
        return x * torch.sigmoid(g)

    def gate_from_pool(self, part, plane_size, batch):
        """The gate (B, C) from the pooling partials the depthwise kernel left behind (two small HIP launches)."""
        return hip.se_gate(part, plane_size, batch, self.conv_reduce.weight, self.conv_reduce.bias,
                           self.conv_expand.weight, self.conv_expand.bias)


def _bn(ch):
    return nn.BatchNorm2d(ch, eps=_BN_EPS)


class DepthwiseSeparableConv(nn.Module):
    def __init__(self, cin, cout, k, stride):
        super().__init__()
        self.skip = stride == 1 and cin == cout
        self.conv_dw = Conv2dSame(cin, cin, k, stride=stride, groups=cin, bias=False)
        self.bn1 = _bn(cin)
        self.act1 = Swish()
        self.se = SqueezeExcite(cin, max(1, int(cin * 0.25 + 0.5)))
        self.conv_pw = Conv2dSame(cin, cout, 1, bias=False)
        self.bn2 = _bn(cout)
        self.act2 = nn.Identity()

    def forward(self, x):
        if _fast(x, self) and pw_wins(x):
            # 3 launches: depthwise + BN + swish (+ SE pooling), SE gate, project GEMM with gate / BN / skip fused
            y, part, plane = hip.dwconv2d_same_pool(x, self.conv_dw.weight, *bn_affine_cached(self.bn1),
                                                    self.conv_dw.stride[0], "swish")
            gate = self.se.gate_from_pool(part, plane, x.shape[0])
            return project_conv(self, self.conv_pw, self.bn2, y, gate, x if self.skip else None)
        if _fast(x, self):
            y = hip.dwconv2d_same(x, self.conv_dw.weight, *bn_affine_cached(self.bn1), self.conv_dw.stride[0], "swish")
            y = F.conv2d(self.se(y), self.conv_pw.weight)
            return hip.affine_act(y, *bn_affine_cached(self.bn2), None, res=x if self.skip else None)
        # training: BatchNorm + swish / BatchNorm + skip as fused passes (bn.py; plain modules on the CPU)
        y = self.se(bn_act(self.bn1, self.conv_dw(x), "swish"))
        return bn_act(self.bn2, self.conv_pw(y), res=x if self.skip else None)


class InvertedResidual(nn.Module):
    def __init__(self, cin, cout, k, stride, expand):
        super().__init__()
        mid = cin * expand
        self.skip = stride == 1 and cin == cout
        self.conv_pw = Conv2dSame(cin, mid, 1, bias=False)
        self.bn1 = _bn(mid)
        self.act1 = Swish()
        self.conv_dw = Conv2dSame(mid, mid, k, stride=stride, groups=mid, bias=False)
        self.bn2 = _bn(mid)
        self.act2 = Swish()
        self.se = SqueezeExcite(mid, max(1, int(cin * 0.25 + 0.5)))
        self.conv_pwl = Conv2dSame(mid, cout, 1, bias=False)
        self.bn3 = _bn(cout)

    def forward(self, x):
        if _fast(x, self) and pw_wins(x):
            # 4 launches instead of 11: expand GEMM + BN + swish, depthwise + BN + swish + SE pooling, SE gate,
            # project GEMM with the gate on its input channels + BN + skip
            if expand_on_library(x):
                y = expand_gemm(self, self.conv_pw, self.bn1, x, "swish", pad_out=True)
            else:
                wpk, shift = pw_operands(self, self.conv_pw, self.bn1)
                y = hip.conv1x1(x, wpk, self.conv_pw.out_channels, shift, "swish")
            y, part, plane = hip.dwconv2d_same_pool(y, self.conv_dw.weight, *bn_affine_cached(self.bn2),
                                                    self.conv_dw.stride[0], "swish")
            gate = self.se.gate_from_pool(part, plane, x.shape[0])
            return project_conv(self, self.conv_pwl, self.bn3, y, gate, x if self.skip else None)
        if _fast(x, self):
            y = hip.affine_act(F.conv2d(x, self.conv_pw.weight), *bn_affine_cached(self.bn1), "swish")
            y = hip.dwconv2d_same(y, self.conv_dw.weight, *bn_affine_cached(self.bn2), self.conv_dw.stride[0], "swish")
            y = F.conv2d(self.se(y), self.conv_pwl.weight)
            return hip.affine_act(y, *bn_affine_cached(self.bn3), None, res=x if self.skip else None)
        # training: BatchNorm + swish / BatchNorm + skip as fused passes (bn.py; plain modules on the CPU)
        y = bn_act(self.bn1, self.conv_pw(x), "swish")
        y = self.se(bn_act(self.bn2, self.conv_dw(y), "swish"))
        return bn_act(self.bn3, self.conv_pwl(y), res=x if self.skip else None)


class EfficientNet(nn.Module):
    def __init__(self, name="tf_efficientnet_b7_ns", num_classes=1000):
        super().__init__()
        if name not in _SCALING:
            raise NotImplementedError(f"unknown backbone {name}")
        wmul, dmul = _SCALING[name]
        stem = _round_channels(32, wmul)
        self.conv_stem = Conv2dSame(3, stem, 3, stride=2, bias=False)
        self.bn1 = _bn(stem)
        self.act1 = Swish()
        stages, cin = [], stem
        for kind, rep, k, stride, expand, ch in _B0_STAGES:
            cout = _round_channels(ch, wmul)
            blocks = []
            for i in range(int(math.ceil(rep * dmul))):
                s = stride if i == 0 else 1
                blocks.append(DepthwiseSeparableConv(cin, cout, k, s) if kind == "ds"
                              else InvertedResidual(cin, cout, k, s, expand))
                cin = cout
            stages.append(nn.Sequential(*blocks))
        self.blocks = nn.Sequential(*stages)
        self.num_features = _round_channels(1280, wmul)
        self.conv_head = Conv2dSame(cin, self.num_features, 1, bias=False)
        self.bn2 = _bn(self.num_features)
        self.act2 = Swish()
        self.global_pool = nn.AdaptiveAvgPool2d(1)
        self.classifier = nn.Linear(self.num_features, num_classes)

    def forward(self, x):
        x = self.act1(self.bn1(self.conv_stem(x)))
        x = self.act2(self.bn2(self.conv_head(self.blocks(x))))
        return self.classifier(self.global_pool(x).flatten(1))
