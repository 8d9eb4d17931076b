"""NYUv2 3-D UNet (mirror of occdepth/models/unet3d_nyu.py:16-110): as the KITTI one without the
final x2 upsampling -- the head runs at the lift resolution -- and with `n_relations` exposed."""
import numpy as np
import torch.nn as nn

from ..fused import as_vox, needs_autograd
from .CRP3D import CPMegaVoxels
from .modules import Downsample, Process, SegmentationHead, SegmentationHeadCascadeCLS, Upsample


class UNet3D(nn.Module):
    def __init__(self, class_num, norm_layer, feature, full_scene_size, n_relations=4, project_res=[],
                 context_prior=True, bn_momentum=0.1, cascade_cls=False, infer_mode=False):
        super().__init__()
        self.business_layer = []
        self.project_res = project_res
        self.cascade_cls = cascade_cls
        self.infer_mode = infer_mode
        self.feature_1_4 = self.feature_1_4_dec = feature
        self.feature_1_8 = self.feature_1_8_dec = feature * 2
        self.feature_1_16 = self.feature_1_16_dec = feature * 4

        self.process_1_4 = nn.Sequential(Process(feature, norm_layer, bn_momentum, dilations=[1, 2, 3]),
                                         Downsample(feature, norm_layer, bn_momentum))
        self.process_1_8 = nn.Sequential(Process(feature * 2, norm_layer, bn_momentum, dilations=[1, 2, 3]),
                                         Downsample(feature * 2, norm_layer, bn_momentum))
        self.up_1_16_1_8 = Upsample(feature * 4, feature * 2, norm_layer, bn_momentum)
        self.up_1_8_1_4 = Upsample(feature * 2, feature, norm_layer, bn_momentum)
        head = SegmentationHeadCascadeCLS if cascade_cls else SegmentationHead
        self.ssc_head_1_4 = head(feature, feature, class_num, [1, 2, 3])
        self.context_prior = context_prior
        size_1_16 = tuple(int(np.ceil(i / 4)) for i in full_scene_size)
        if context_prior:
            self.CP_mega_voxels = CPMegaVoxels(self.feature_1_16, size_1_16, n_relations=n_relations,
                                               bn_momentum=bn_momentum)

    def _forward_vox(self, x4):
        res = {}
        x8 = x4
        for m in self.process_1_4:
            x8 = m.forward_vox(x8)
        x16 = x8
        for m in self.process_1_8:
            x16 = m.forward_vox(x16)
        if self.context_prior:
            ret = self.CP_mega_voxels.forward_vox(x16)
            x16 = ret["x"]
            res["P_logits"] = ret["P_logits"]
            res["x"] = x16.ncdhw()
        up8 = self.up_1_16_1_8.forward_vox(x16, skip=x8)
        up4 = self.up_1_8_1_4.forward_vox(up8, skip=x4)
        if not self.infer_mode:
            res["x3d_l1"], res["x3d_l2"], res["x3d_l3"] = up4.ncdhw(), up8.ncdhw(), x16.ncdhw()
        if self.cascade_cls:
            ssc, occ = self.ssc_head_1_4.forward_vox(up4)
            res["ssc_logit"] = ssc.ncdhw()
            if not self.infer_mode:
                res["occ_logit"] = occ.ncdhw()
        else:
            res["ssc_logit"] = self.ssc_head_1_4.forward_vox(up4).ncdhw()
        return res

    def _forward_autograd(self, x4):
        res = {}
        x8 = self.process_1_4(x4)
        x16 = self.process_1_8(x8)
        if self.context_prior:
            ret = self.CP_mega_voxels(x16)
            x16 = ret["x"]
            res.update(ret)
        up8 = self.up_1_16_1_8(x16) + x8
        up4 = self.up_1_8_1_4(up8) + x4
        if not self.infer_mode:
            res["x3d_l1"], res["x3d_l2"], res["x3d_l3"] = up4, up8, x16
        if self.cascade_cls:
            res["ssc_logit"], occ = self.ssc_head_1_4(up4)
            if not self.infer_mode:
                res["occ_logit"] = occ
        else:
            res["ssc_logit"] = self.ssc_head_1_4(up4)
        return res

    def forward(self, input_dict):
        x = input_dict["x3d"]
        if needs_autograd(self):
            return self._forward_autograd(x)
        return self._forward_vox(as_vox(x))
