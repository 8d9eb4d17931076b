"""NYUv2 3-D UNet (same sub-module names and constructor as occdepth/models/unet3d_nyu.py:16-110): the KITTI network
without the last upsampling -- the head runs at the lift resolution -- and with `n_relations` exposed.
The forward engine is unet3d_common.UNet3DBase."""
import math

from .CRP3D import CPMegaVoxels
from .modules import Downsample, Process, SegmentationHead, SegmentationHeadCascadeCLS, Upsample
from .unet3d_common import UNet3DBase, nn

DILATIONS = (1, 2, 3)


class UNet3D(UNet3DBase):
    LAYOUT = ("process_1_4", "process_1_8", "up_1_16_1_8", "up_1_8_1_4", None, "ssc_head_1_4", None)

    def __init__(self, class_num, norm_layer, feature, full_scene_size, n_relations=4, project_res=[],
                 context_prior=True, bn_momentum=0.1, cascade_cls=False, infer_mode=False):
        super().__init__()
        self.business_layer = []
        self.project_res, self.cascade_cls, self.infer_mode = project_res, cascade_cls, infer_mode
        self.context_prior = context_prior
        for level, width in (("1_4", feature), ("1_8", 2 * feature), ("1_16", 4 * feature)):
            setattr(self, f"feature_{level}", width)
            setattr(self, f"feature_{level}_dec", width)
        bn = dict(norm_layer=norm_layer, bn_momentum=bn_momentum)

        def encoder(width):
            return nn.Sequential(Process(width, dilations=list(DILATIONS), **bn), Downsample(width, **bn))

        self.process_1_4, self.process_1_8 = encoder(feature), encoder(2 * feature)
        self.up_1_16_1_8 = Upsample(4 * feature, 2 * feature, **bn)
        self.up_1_8_1_4 = Upsample(2 * feature, feature, **bn)
        head_cls = SegmentationHeadCascadeCLS if cascade_cls else SegmentationHead
        self.ssc_head_1_4 = head_cls(feature, feature, class_num, list(DILATIONS))
        if context_prior:
            coarse_size = tuple(math.ceil(n / 4) for n in full_scene_size)
            self.CP_mega_voxels = CPMegaVoxels(4 * feature, coarse_size, n_relations=n_relations, bn_momentum=bn_momentum)
