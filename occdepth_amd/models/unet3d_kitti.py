"""SemanticKITTI 3-D UNet (same sub-module names and constructor as occdepth/models/unet3d_kitti.py:14-126).

The lift volume arrives at 1/project_scale of the scene; the decoder returns to that resolution and one more
transposed convolution (or a plain block when project_scale == 1) reaches the full 256 x 256 x 32 grid where the
(cascade) head runs.  The forward engine is unet3d_common.UNet3DBase.
"""
from .CRP3D import CPMegaVoxels
from .modules import (Convblock3d, Downsample, Process, SegmentationHead, SegmentationHeadCascadeCLS,
                      SegmentationHeadOccludedCLS, Upsample)
from .unet3d_common import UNet3DBase, nn

DILATIONS = (1, 2, 3)


class UNet3D(UNet3DBase):
    LAYOUT = ("process_l1", "process_l2", "up_13_l2", "up_12_l1", "up_l1_lfull", "ssc_head", "occluded_head")

    def __init__(self, class_num, norm_layer, full_scene_size, feature, project_scale, context_prior=None,
                 bn_momentum=0.1, cascade_cls=False, occluded_cls=False, infer_mode=False):
        super().__init__()
        self.business_layer = []
        self.project_scale, self.full_scene_size, self.feature = project_scale, full_scene_size, feature
        self.cascade_cls, self.occluded_cls, self.infer_mode = cascade_cls, occluded_cls, infer_mode
        self.context_prior = context_prior
        lift_size = [int(s / project_scale) for s in full_scene_size]
        coarse_size = tuple(s // 4 for s in lift_size)               # two stride-2 stages below the lift resolution
        bn = dict(norm_layer=norm_layer, bn_momentum=bn_momentum)

        def encoder(width):
            return nn.Sequential(Process(width, dilations=list(DILATIONS), **bn), Downsample(width, **bn))

        self.process_l1, self.process_l2 = encoder(feature), encoder(2 * feature)
        self.up_13_l2 = Upsample(4 * feature, 2 * feature, **bn)
        self.up_12_l1 = Upsample(2 * feature, feature, **bn)
        half = feature // 2
        self.up_l1_lfull = (Convblock3d(feature, half, stride=1, **bn) if project_scale == 1
                            else Upsample(feature, half, **bn))
        head_cls = SegmentationHeadCascadeCLS if cascade_cls else SegmentationHead
        self.ssc_head = head_cls(half, half, class_num, list(DILATIONS))
        if occluded_cls:
            self.occluded_head = SegmentationHeadOccludedCLS(half, half, class_num, list(DILATIONS))
        if context_prior:
            self.CP_mega_voxels = CPMegaVoxels(4 * feature, coarse_size, bn_momentum=bn_momentum)
