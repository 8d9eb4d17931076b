"""DDR-style factorised 3-D residual bottleneck (mirror of occdepth/models/DDR.py:35-139).

Parameter layout is the reference's (conv1..conv5, bn1..bn5, downsample, downsample2/3/4, the
last three always constructed even when stride == 1 so state_dicts interchange).  Eval-mode
forward is five K2 launches with BatchNorm folded and every ReLU / residual add fused:

    o1 = relu(c1(x))                 o2 = c2(o1)
    o3 = c3(relu(o2)) + p2(o2)       o4 = c4(relu(o3)) + p3(p2(o2)) + p4(o3)
    y  = relu(c5(relu(o4)) + skip(x))

where p2/p3/p4 are the AvgPool+1x1x1+BN side branches (identity when stride == 1), run as
k = stride = window convolutions.
"""
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import hip

from ..autograd3d import Conv3d, ConvTranspose3d  # noqa: F401  (nn.Conv3d subclasses: HIP forward/backward in training)
from ..bn import bn_act
from ..fused import ACT_RELU, ConvPlan, _bn_affine, _stamp, as_vox, needs_autograd


def _axis(value, axis, other):
    t = [other] * 3
    t[axis] = value
    return tuple(t)


class Bottleneck3D(nn.Module):
    def __init__(self, inplanes, planes, norm_layer, stride=1, dilation=[1, 1, 1], expansion=4,
                 downsample=None, fist_dilation=1, multi_grid=1, bn_momentum=0.0003):
        super().__init__()
        self.expansion = expansion
        self.stride = stride
        self.dilation = dilation
        self.conv1 = Conv3d(inplanes, planes, kernel_size=1, bias=False)
        self.bn1 = norm_layer(planes, momentum=bn_momentum)
        # conv2 / conv3 / conv4 filter along Z / Y / X respectively (tensor dims are X, Y, Z)
        for idx, axis in ((2, 2), (3, 1), (4, 0)):
            d = dilation[idx - 2]
            setattr(self, f"conv{idx}", Conv3d(
                planes, planes, kernel_size=_axis(3, axis, 1), stride=_axis(stride, axis, 1),
                dilation=_axis(d, axis, 1), padding=_axis(d, axis, 0), bias=False))
            setattr(self, f"bn{idx}", norm_layer(planes, momentum=bn_momentum))
        self.conv5 = Conv3d(planes, planes * expansion, kernel_size=1, bias=False)
        self.bn5 = norm_layer(planes * expansion, momentum=bn_momentum)
        self.relu = nn.ReLU(inplace=False)
        self.relu_inplace = nn.ReLU(inplace=True)
        self.downsample = downsample

        def side(window):
            return nn.Sequential(nn.AvgPool3d(kernel_size=window, stride=window),
                                 Conv3d(planes, planes, kernel_size=1, stride=1, bias=False),
                                 norm_layer(planes, momentum=bn_momentum))

        self.downsample2 = side((1, stride, 1))
        self.downsample3 = side((stride, 1, 1))
        self.downsample4 = side((stride, 1, 1))
        self._plans = None

    # ------------------------------------------------------------------ HIP (eval)
    def _build_plans(self):
        p = {i: ConvPlan(getattr(self, f"conv{i}"), getattr(self, f"bn{i}")) for i in range(1, 6)}
        if self.stride != 1:
            for i in (2, 3, 4):
                seq = getattr(self, f"downsample{i}")
                p[f"p{i}"] = ConvPlan(seq[1], seq[2], pool=seq[0].kernel_size)
        if self.downsample is not None:
            pool = self.downsample[0]
            p["skip"] = ConvPlan(self.downsample[1], self.downsample[2], pool=pool.kernel_size)
        return p

    # K14 (csrc/bneck3d.hip): the stride-1 block as two launches on the fp32 matrix pipe (P = 16 / 32 planes, Z = 4 / 8 / 16):
    # 67.5 us against 192 us for the five K2 launches at 128x128x16, -0.3 ms per config-2 frame (profiles/r03_k14_ab.txt).
    # OCCDEPTH_FUSED_BOTTLENECK=0 keeps the five launches for A/B.
    FUSED = os.environ.get("OCCDEPTH_FUSED_BOTTLENECK", "1") == "1"

    def _packed_block(self):
        """F(W1^T) | b1 | F(W2[k]) | b2 | F(W3[k]) | b3 | F(W4[k]) | b4 | F(W5^T) | b5 with the BatchNorm scales folded into the
        weights and every [cin][cout] matrix in MFMA fragment order (the layout occd_bottleneck3d_fwd documents), rebuilt
        when a source tensor changes."""
        mods = [getattr(self, f"{k}{i}") for i in range(1, 6) for k in ("conv", "bn")]
        key = _stamp(*mods)
        if getattr(self, "_k14_key", None) == key:
            return self._k14_w
        parts = []
        for i in range(1, 6):
            conv, bn = getattr(self, f"conv{i}"), getattr(self, f"bn{i}")
            w = conv.weight.detach().float()
            cout = w.shape[0]
            scale, shift = _bn_affine(bn, cout, w.device)
            w = w.reshape(cout, w.shape[1], -1) * scale.view(-1, 1, 1)       # (out, in, taps)
            cin = w.shape[1]
            m = w.permute(2, 1, 0)                                            # (tap, in, out)
            frag = m.reshape(-1, cin // 16, 4, 4, cout // 16, 16).permute(0, 1, 4, 2, 5, 3)   # (tap, t, m, g, i, e)
            parts += [frag.reshape(-1), shift.float().reshape(-1)]
        self._k14_w = torch.cat(parts).contiguous()
        self._k14_key = key
        return self._k14_w

    def forward_vox(self, x):
        if (self.FUSED and self.stride == 1 and self.downsample is None and x.buf.dtype == torch.float32
                and hip.bottleneck3d_supported(x.C, self.conv1.out_channels, x.dims)
                and self.conv5.out_channels == x.C):
            d = self.dilation
            return hip.bottleneck3d(x, self._packed_block(), self.conv1.out_channels, (d[0], d[1], d[2]))
        if self._plans is None:
            self._plans = self._build_plans()
        p = self._plans
        strided = self.stride != 1
        o1 = p[1](x, act_out=ACT_RELU)
        o2 = p[2](o1)
        o2s = p["p2"](o2) if strided else o2
        o3 = p[3](o2, res1=o2s, act_in=ACT_RELU)
        o2ss = p["p3"](o2s) if strided else o2s
        o3s = p["p4"](o3) if strided else o3
        o4 = p[4](o3, res1=o2ss, res2=o3s, act_in=ACT_RELU)
        skip = p["skip"](x) if self.downsample is not None else x
        return p[5](o4, res1=skip, act_in=ACT_RELU, act_out=ACT_RELU)

    # ------------------------------------------------------------------ ATen (training / autograd)
    @staticmethod
    def _side(seq, t):
        """AvgPool3d -> 1x1x1 conv -> BatchNorm side branch (`downsample*`)"""
        return bn_act(seq[2], seq[1](seq[0](t)))

    def _forward_autograd(self, x):
        """The reference's graph (DDR.py:111-139); every BatchNorm with its ReLU / residual add as one fused pass (bn.py)."""
        strided = self.stride != 1
        o1 = bn_act(self.bn1, self.conv1(x), "relu")
        o2 = bn_act(self.bn2, self.conv2(o1))
        o2s = self._side(self.downsample2, o2) if strided else o2
        o3 = bn_act(self.bn3, self.conv3(F.relu(o2)), res=o2s)                     # bn3(.) + o2
        o2ss = self._side(self.downsample3, o2s) if strided else o2s
        o3s = self._side(self.downsample4, o3) if strided else o3
        o4 = bn_act(self.bn4, self.conv4(F.relu(o3)), res=o2ss + o3s)              # bn4(.) + o2 + o3
        skip = x if self.downsample is None else self._side(self.downsample, x)
        return bn_act(self.bn5, self.conv5(F.relu(o4)), "relu", res=skip, res_first=True)   # relu(bn5(.) + skip)

    def forward(self, x):
        if needs_autograd(self):
            return self._forward_autograd(x)
        return self.forward_vox(as_vox(x)).ncdhw()
