"""SemanticKITTI 3-D UNet (mirror of occdepth/models/unet3d_kitti.py:14-126).

Eval forward keeps every activation channels-last on the GPU (`Vox`) from the lifted volume to
the logits: 2 x (3 dilated bottlenecks + strided bottleneck), the context-relation prior at
1/4 of the lift resolution, two transposed-conv upsamplings with the skip adds fused after the
ReLU, the x2 upsampling to full resolution and the (cascade) segmentation head.  Returned
tensors are logical (B, C, X, Y, Z) views with channels_last_3d strides.
"""
import torch.nn as nn

from ..fused import as_vox, needs_autograd
from .CRP3D import CPMegaVoxels
from .modules import (Convblock3d, Downsample, Process, SegmentationHead, SegmentationHeadCascadeCLS,
                      SegmentationHeadOccludedCLS, Upsample)


class UNet3D(nn.Module):
    def __init__(self, class_num, norm_layer, full_scene_size, feature, project_scale, context_prior=None,
                 bn_momentum=0.1, cascade_cls=False, occluded_cls=False, infer_mode=False):
        super().__init__()
        self.business_layer = []
        self.project_scale = project_scale
        self.full_scene_size = full_scene_size
        self.feature = feature
        self.cascade_cls = cascade_cls
        self.occluded_cls = occluded_cls
        self.infer_mode = infer_mode
        f = feature
        size_l1 = tuple(int(s / project_scale) for s in full_scene_size)
        size_l3 = tuple(s // 2 // 2 for s in size_l1)
        dilations = [1, 2, 3]

        self.process_l1 = nn.Sequential(Process(f, norm_layer, bn_momentum, dilations=[1, 2, 3]),
                                        Downsample(f, norm_layer, bn_momentum))
        self.process_l2 = nn.Sequential(Process(f * 2, norm_layer, bn_momentum, dilations=[1, 2, 3]),
                                        Downsample(f * 2, norm_layer, bn_momentum))
        self.up_13_l2 = Upsample(f * 4, f * 2, norm_layer, bn_momentum)
        self.up_12_l1 = Upsample(f * 2, f, norm_layer, bn_momentum)
        if project_scale == 1:
            self.up_l1_lfull = Convblock3d(f, f // 2, norm_layer, bn_momentum, stride=1)
        else:
            self.up_l1_lfull = Upsample(f, f // 2, norm_layer, bn_momentum)
        head = SegmentationHeadCascadeCLS if cascade_cls else SegmentationHead
        self.ssc_head = head(f // 2, f // 2, class_num, dilations)
        if occluded_cls:
            self.occluded_head = SegmentationHeadOccludedCLS(f // 2, f // 2, class_num, dilations)
        self.context_prior = context_prior
        if context_prior:
            self.CP_mega_voxels = CPMegaVoxels(f * 4, size_l3, bn_momentum=bn_momentum)

    def _forward_vox(self, x1):
        res = {}
        x2 = x1
        for m in self.process_l1:
            x2 = m.forward_vox(x2)
        x3 = x2
        for m in self.process_l2:
            x3 = m.forward_vox(x3)
        if self.context_prior:
            ret = self.CP_mega_voxels.forward_vox(x3)
            x3 = ret["x"]
            res["P_logits"] = ret["P_logits"]
            res["x"] = x3.ncdhw()
        up2 = self.up_13_l2.forward_vox(x3, skip=x2)
        up1 = self.up_12_l1.forward_vox(up2, skip=x1)
        full = self.up_l1_lfull.forward_vox(up1)
        if not self.infer_mode:
            res["x3d_l1"], res["x3d_l2"], res["x3d_l3"] = up1.ncdhw(), up2.ncdhw(), x3.ncdhw()
        if self.cascade_cls:
            ssc, occ = self.ssc_head.forward_vox(full)
            res["ssc_logit"] = ssc.ncdhw()
            if not self.infer_mode:
                res["occ_logit"] = occ.ncdhw()
        else:
            res["ssc_logit"] = self.ssc_head.forward_vox(full).ncdhw()
        if self.occluded_cls:
            occluded = self.occluded_head.forward_vox(full)
            if not self.infer_mode:
                res["occluded_logit"] = occluded.ncdhw()
        return res

    def _forward_autograd(self, x1):
        res = {}
        x2 = self.process_l1(x1)
        x3 = self.process_l2(x2)
        if self.context_prior:
            ret = self.CP_mega_voxels(x3)
            x3 = ret["x"]
            res.update(ret)
        up2 = self.up_13_l2(x3) + x2
        up1 = self.up_12_l1(up2) + x1
        full = self.up_l1_lfull(up1)
        if not self.infer_mode:
            res["x3d_l1"], res["x3d_l2"], res["x3d_l3"] = up1, up2, x3
        if self.cascade_cls:
            res["ssc_logit"], occ = self.ssc_head(full)
            if not self.infer_mode:
                res["occ_logit"] = occ
        else:
            res["ssc_logit"] = self.ssc_head(full)
        if self.occluded_cls:
            occluded = self.occluded_head(full)
            if not self.infer_mode:
                res["occluded_logit"] = occluded
        return res

    def forward(self, input_dict):
        x = input_dict["x3d"]
        if needs_autograd(self):
            return self._forward_autograd(x)
        return self._forward_vox(as_vox(x))
