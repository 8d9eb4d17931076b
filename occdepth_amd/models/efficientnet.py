"""In-repo EfficientNet (tf_efficientnet_b{3,4,5,7}_ns topology) for the 2-D encoder.

The reference pulls `rwightman/gen-efficientnet-pytorch` through torch.hub
(occdepth/models/unet2d.py:238-240); that dependency is not vendored and there is no network,
so the architecture is restated here with geffnet's module / parameter names (conv_stem, bn1,
act1, blocks.{stage}.{block}.{conv_pw,bn1,conv_dw,bn2,se.conv_reduce,se.conv_expand,conv_pwl,bn3},
conv_head, bn2, act2, global_pool, classifier) so hub checkpoints load by key.  TF "SAME"
padding, swish, SE on the block input width, BN eps 1e-3.  Runs on PyTorch-ROCm / MIOpen.
Parity of this file is UNPINNED (no reference copy of geffnet exists here).
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import hip
from ..fused import bn_affine_cached, needs_autograd


def _fast(x, module):
    """eval-mode CUDA tensors take the fused HIP elementwise / depthwise kernels (occdepth_amd/csrc/nchw2d.hip)."""
    return x.is_cuda and not needs_autograd(module) and x.dtype == torch.float32

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
        pads = []
        for size, k, s, d in zip(x.shape[-2:], self.kernel_size, self.stride, self.dilation):
            total = max((math.ceil(size / s) - 1) * s + (k - 1) * d + 1 - size, 0)
            pads = [total // 2, total - total // 2] + pads
        if any(pads):
            x = F.pad(x, pads)
        return F.conv2d(x, self.weight, self.bias, self.stride, (0, 0), self.dilation, self.groups)


class Swish(nn.Module):
    def forward(self, x):
        if _fast(x, self):
            return hip.affine_act(x, None, None, "swish", out=torch.empty_like(x))
        return x * torch.sigmoid(x)


class SqueezeExcite(nn.Module):
    def __init__(self, chs, reduced):
        super().__init__()
        self.conv_reduce = nn.Conv2d(chs, reduced, 1, bias=True)
        self.act1 = Swish()
        self.conv_expand = nn.Conv2d(reduced, chs, 1, bias=True)

    def forward(self, x):
        g = self.conv_expand(self.act1(self.conv_reduce(x.mean((2, 3), keepdim=True))))
        return x * torch.sigmoid(g)


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
        if _fast(x, self):
            y = hip.dwconv2d_same(x, self.conv_dw.weight, *bn_affine_cached(self.bn1), self.conv_dw.stride[0], "swish")
            y = F.conv2d(self.se(y), self.conv_pw.weight)
            return hip.affine_act(y, *bn_affine_cached(self.bn2), None, res=x if self.skip else None)
        y = self.se(self.act1(self.bn1(self.conv_dw(x))))
        y = self.act2(self.bn2(self.conv_pw(y)))
        return y + x if self.skip else y


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
        if _fast(x, self):
            y = hip.affine_act(F.conv2d(x, self.conv_pw.weight), *bn_affine_cached(self.bn1), "swish")
            y = hip.dwconv2d_same(y, self.conv_dw.weight, *bn_affine_cached(self.bn2), self.conv_dw.stride[0], "swish")
            y = F.conv2d(self.se(y), self.conv_pwl.weight)
            return hip.affine_act(y, *bn_affine_cached(self.bn3), None, res=x if self.skip else None)
        y = self.act1(self.bn1(self.conv_pw(x)))
        y = self.se(self.act2(self.bn2(self.conv_dw(y))))
        y = self.bn3(self.conv_pwl(y))
        return y + x if self.skip else y


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
