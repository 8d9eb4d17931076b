"""3-D building blocks (mirror of occdepth/models/modules.py): ASPP, the three segmentation
heads, Process / Upsample / Convblock3d / Downsample.

State-dict names follow the reference (conv0, conv1.{i}, bn1.{i}, conv2.{i}, bn2.{i},
conv_classes, occ_classes, main.*).  Eval mode runs on the HIP implicit-GEMM kernel with
BatchNorm folded, ReLU / residual accumulation fused into the epilogues; the cascade head splits
`conv_classes` by linearity so its wide half shares one MFMA N-tile with `occ_classes`.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import hip
from ..autograd3d import Conv3d, ConvTranspose3d  # noqa: F401  (nn.Conv3d subclasses: HIP forward/backward in training)
from ..bn import bn_act
from ..fused import ACT_RELU, ACT_RELU_PRE, ConvPlan, ConvTransposePlan, DerivedConvPlan, Vox, as_vox, needs_autograd, run_parallel
from .DDR import Bottleneck3D


class _DilatedBranches(nn.Module):
    """sum_d BN2_d(conv2_d(relu(BN1_d(conv1_d(x))))) -> relu(sum + x)  (LMSCNet-style ASPP)."""

    def _make_branches(self, planes, dilations):
        self.conv_list = dilations

        def convs():
            return nn.ModuleList([Conv3d(planes, planes, kernel_size=3, padding=d, dilation=d, bias=False)
                                  for d in dilations])

        def norms():
            return nn.ModuleList([nn.BatchNorm3d(planes) for _ in dilations])

        self.conv1, self.bn1 = convs(), norms()
        self.conv2, self.bn2 = convs(), norms()
        self.relu = nn.ReLU()

    def _branch_plans(self):
        return [(ConvPlan(c1, b1), ConvPlan(c2, b2))
                for c1, b1, c2, b2 in zip(self.conv1, self.bn1, self.conv2, self.bn2)]

    # small volumes (the CRP's 32x32x4 ASPP: 256 workgroups per launch, 62 % MFMA-bound): the branches' first
    # convolutions are independent and run side by side on forked streams (fused.run_parallel)
    PARALLEL_BELOW = 65536

    def _branches_vox(self, plans, x, out=None):
        y = None
        last = len(plans) - 1
        if x.buf.is_cuda and len(plans) > 1 and x.batch * x.dims[0] * x.dims[1] * x.dims[2] <= self.PARALLEL_BELOW:
            for first, second in plans:
                first._prepare()
                second._prepare()
            ts = [Vox.empty(x.batch, first.out_dims(x.dims), first.cout, x.buf.device) for first, _ in plans]
            run_parallel([(lambda f=first, t=t: f(x, out=t, act_out=ACT_RELU)) for (first, _), t in zip(plans, ts)])
            for i, ((_, second), t) in enumerate(zip(plans, ts)):
                if i < last:
                    y = second(t, res1=y)
                else:
                    y = second(t, out=out, res1=y, res2=x, act_out=ACT_RELU)
            return y
        for i, (first, second) in enumerate(plans):
            t = first(x, act_out=ACT_RELU)
            if i < last:
                y = second(t, res1=y)
            else:
                y = second(t, out=out, res1=y, res2=x, act_out=ACT_RELU)
        return y

    def _branches_autograd(self, x):
        """relu(sum_d bn2_d(conv2_d(relu(bn1_d(conv1_d(x))))) + x) with every BatchNorm, its ReLU and the running sum as
        fused passes (bn.py); the last branch takes `y + x` as its pre-activation residual and applies the final ReLU."""
        y = None
        last = len(self.conv1) - 1
        for i, (c1, b1, c2, b2) in enumerate(zip(self.conv1, self.bn1, self.conv2, self.bn2)):
            t = c2(bn_act(b1, c1(x), "relu"))
            if i < last:
                y = bn_act(b2, t, res=y)
            else:
                y = bn_act(b2, t, "relu", res=x if y is None else y + x, res_first=True)
        return y


class ASPP(_DilatedBranches):
    def __init__(self, planes, dilations_conv_list):
        super().__init__()
        self._make_branches(planes, dilations_conv_list)
        self._plans = None

    def forward_vox(self, x):
        if self._plans is None:
            self._plans = self._branch_plans()
        return self._branches_vox(self._plans, x)

    def forward(self, x_in):
        if needs_autograd(self):
            return self._branches_autograd(x_in)
        return self.forward_vox(as_vox(x_in)).ncdhw()


class _HeadBase(_DilatedBranches):
    def _make_head(self, inplanes, planes, dilations):
        self.conv0 = Conv3d(inplanes, planes, kernel_size=3, padding=1, stride=1)
        self._make_branches(planes, dilations)
        self._plans = None

    def _trunk_vox(self, x, out=None):
        if self._plans is None:
            self._plans = {"conv0": ConvPlan(self.conv0), "branches": self._branch_plans()}
            self._extra_plans(self._plans)
        t0 = self._plans["conv0"](x, act_out=ACT_RELU)
        return self._branches_vox(self._plans["branches"], t0, out=out)

    def _extra_plans(self, plans):
        pass

    def _trunk_autograd(self, x):
        return self._branches_autograd(F.relu(self.conv0(x)))


class SegmentationHead(_HeadBase):
    def __init__(self, inplanes, planes, nbr_classes, dilations_conv_list):
        super().__init__()
        self._make_head(inplanes, planes, dilations_conv_list)
        self.conv_classes = Conv3d(planes, nbr_classes, kernel_size=3, padding=1, stride=1)

    def _extra_plans(self, plans):
        plans["cls"] = ConvPlan(self.conv_classes)

    def forward_vox(self, x):
        feat = self._trunk_vox(x)
        return self._plans["cls"](feat)

    def forward(self, x_in):
        if needs_autograd(self):
            return self.conv_classes(self._trunk_autograd(x_in))
        return self.forward_vox(as_vox(x_in)).ncdhw()


class SegmentationHeadCascadeCLS(_HeadBase):
    def __init__(self, inplanes, planes, nbr_classes, dilations_conv_list):
        super().__init__()
        self._make_head(inplanes, planes, dilations_conv_list)
        occ_classes = 2
        self.conv_classes = Conv3d(planes + occ_classes, nbr_classes, kernel_size=3, padding=1, stride=1)
        self.occ_classes = Conv3d(planes, occ_classes, kernel_size=3, padding=1, stride=1)
        self.softmax = nn.Softmax(dim=1)
        self.planes = planes

    def _extra_plans(self, plans):
        # conv_classes(cat[f, softmax(occ)]) = conv_classes[:, :planes](f) + conv_classes[:, planes:](softmax(occ))
        # (linearity), so the wide part shares ONE N=32 MFMA tile with occ_classes: cout = nbr + 2 <= 32.
        p, cls, occ = self.planes, self.conv_classes, self.occ_classes
        plans["wide"] = DerivedConvPlan(
            [cls, occ], lambda: (torch.cat([cls.weight[:, :p], occ.weight], 0), torch.cat([cls.bias, occ.bias], 0)))

    def forward_vox(self, x):
        """returns (ssc_logit Vox, occ_logit Vox)."""
        feat = self._trunk_vox(x)                       # relu(y + x_in), planes channels
        nbr = self.conv_classes.out_channels
        part = self._plans["wide"](feat)                # [0, nbr): partial class logits, [nbr, nbr+2): occ logits
        occ = Vox(part.buf, 2, nbr)
        # narrow half (K = 2 x 27) + softmax + concat: one VALU kernel, no softmax / concat buffer
        ssc = hip.cascade_tail(part, nbr, self.conv_classes.weight[:, self.planes:], nbr)
        return ssc, occ

    def forward(self, x_in):
        if needs_autograd(self):
            feat = self._trunk_autograd(x_in)
            x_occ = self.occ_classes(feat)
            return self.conv_classes(torch.cat([feat, self.softmax(x_occ)], dim=1)), x_occ
        ssc, occ = self.forward_vox(as_vox(x_in))
        return ssc.ncdhw(), occ.ncdhw()


class SegmentationHeadOccludedCLS(_HeadBase):
    def __init__(self, inplanes, planes, nbr_classes, dilations_conv_list):
        super().__init__()
        self._make_head(inplanes, planes, dilations_conv_list)
        self.occ_classes = Conv3d(planes, 2, kernel_size=3, padding=1, stride=1)

    def _extra_plans(self, plans):
        plans["occ"] = ConvPlan(self.occ_classes)

    def forward_vox(self, x):
        feat = self._trunk_vox(x)
        return self._plans["occ"](feat)

    def forward(self, x_in):
        if needs_autograd(self):
            return self.occ_classes(self._trunk_autograd(x_in))
        return self.forward_vox(as_vox(x_in)).ncdhw()


class Process(nn.Module):
    def __init__(self, feature, norm_layer, bn_momentum, dilations=[1, 2, 3]):
        super().__init__()
        self.main = nn.Sequential(*[
            Bottleneck3D(feature, feature // 4, bn_momentum=bn_momentum, norm_layer=norm_layer, dilation=[d, d, d])
            for d in dilations])

    def forward_vox(self, x):
        for block in self.main:
            x = block.forward_vox(x)
        return x

    def forward(self, x):
        if needs_autograd(self):
            return self.main(x)
        return self.forward_vox(as_vox(x)).ncdhw()


class _TransposedBlock(nn.Module):
    """ConvTranspose3d + norm + ReLU under `main.{0,1,2}`; eval: sub-pixel phase convs with the
    affine, the ReLU and an optional skip tensor (added AFTER the ReLU) fused."""

    def _make(self, in_channels, out_channels, norm_layer, bn_momentum, stride, output_padding):
        self.main = nn.Sequential(
            ConvTranspose3d(in_channels, out_channels, kernel_size=3, stride=stride, padding=1, dilation=1,
                               output_padding=output_padding),
            norm_layer(out_channels, momentum=bn_momentum),
            nn.ReLU())
        self._plan = None

    def forward_vox(self, x, skip=None):
        if self._plan is None:
            self._plan = ConvTransposePlan(self.main[0], self.main[1])
        if skip is None:
            return self._plan(x, act_out=ACT_RELU)
        return self._plan(x, res1=skip, act_out=ACT_RELU_PRE)

    def forward_train(self, x, skip=None):
        """ReLU(BN(ConvTranspose3d(x))) [+ skip]: BatchNorm, ReLU and the decoder's skip addition as one fused pass"""
        act = "relu" if isinstance(self.main[2], nn.ReLU) else None
        y = bn_act(self.main[1], self.main[0](x), act, res=skip)
        return y if act is not None or len(self.main) < 3 else self.main[2](y)

    def forward(self, x):
        if needs_autograd(self):
            return self.forward_train(x)
        return self.forward_vox(as_vox(x)).ncdhw()


class Upsample(_TransposedBlock):
    def __init__(self, in_channels, out_channels, norm_layer, bn_momentum):
        super().__init__()
        self._make(in_channels, out_channels, norm_layer, bn_momentum, stride=2, output_padding=1)


class Convblock3d(_TransposedBlock):
    def __init__(self, in_channels, out_channels, norm_layer, bn_momentum, stride=1):
        super().__init__()
        self._make(in_channels, out_channels, norm_layer, bn_momentum, stride=stride, output_padding=0)


class Downsample(nn.Module):
    def __init__(self, feature, norm_layer, bn_momentum, expansion=8):
        super().__init__()
        wide = int(feature * expansion / 4)
        self.main = Bottleneck3D(
            feature, feature // 4, bn_momentum=bn_momentum, expansion=expansion, stride=2,
            downsample=nn.Sequential(nn.AvgPool3d(kernel_size=2, stride=2),
                                     Conv3d(feature, wide, kernel_size=1, stride=1, bias=False),
                                     norm_layer(wide, momentum=bn_momentum)),
            norm_layer=norm_layer)

    def forward_vox(self, x):
        return self.main.forward_vox(x)

    def forward(self, x):
        if needs_autograd(self):
            return self.main(x)
        return self.forward_vox(as_vox(x)).ncdhw()
