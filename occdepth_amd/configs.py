"""The reference's shipped hydra configs as attribute dicts (values of
occdepth/config/{semantic_kitti,NYU}/*.yaml; only the keys OccDepth.__init__ reads plus the
shape-defining ones).  `kitti_a100` is BASELINE.json configs[1..3]; `nyu_2080ti` configs[0]."""


class Config(dict):
    __getattr__ = dict.__getitem__
    __setattr__ = dict.__setitem__

    def clone(self, **kw):
        c = Config(self)
        c.update(kw)
        return c


_COMMON = dict(
    n_relations=4, frustum_size=8, batch_size_per_gpu=1, n_gpus=1, lr=2e-4, weight_decay=1e-4, fp_loss=True,
    relation_loss=True, CE_ssc_loss=True, sem_scal_loss=True, geo_scal_loss=True, depth_loss_weight=1.0,
    sem_step_decay_loss=False, use_lidar_depth_gt=False, occluded_cls=False, return_up_feats=1, pattern_id=0,
    project_1_2=True, project_1_4=True, project_1_8=True,
)

# semantic_kitti/multicam_flospdepth_crp_stereodepth_cascadecls_a100.yaml
kitti_a100 = Config(_COMMON, dataset="kitti", full_scene_size=(256, 256, 32), project_scale=2, feature=64,
                    feature_2d_oc=64, n_classes=20, backbone_2d_name="tf_efficientnet_b7_ns", cascade_cls=True,
                    context_prior=True, trans_2d_to_3d="flosp_depth", multi_view_mode=True,
                    share_2d_backbone_gradient=False, use_stereo_depth_gt=True, use_depth_gt=False)
# semantic_kitti/multicam_flospdepth_crp_stereodepth_cascadecls_2080ti.yaml (released checkpoint)
kitti_2080ti = kitti_a100.clone(feature=32, feature_2d_oc=32, backbone_2d_name="tf_efficientnet_b3_ns")
# semantic_kitti/multicam_flosp_crp_cascadecls_a100.yaml
kitti_flosp_a100 = kitti_a100.clone(trans_2d_to_3d="flosp", use_stereo_depth_gt=False)
# NYU/multicam_flosp_crp_stereodepth_cascadecls_2080ti.yaml
nyu_2080ti = Config(_COMMON, dataset="NYU", full_scene_size=(60, 36, 60), project_scale=1, feature=100,
                    feature_2d_oc=100, n_classes=12, backbone_2d_name="tf_efficientnet_b4_ns", cascade_cls=False,
                    context_prior=False, trans_2d_to_3d="flosp", multi_view_mode=False,
                    share_2d_backbone_gradient=True, use_stereo_depth_gt=False, use_depth_gt=True)
# NYU/multicam_flospdepth_crp_stereodepth_cascadecls_v100.yaml
nyu_v100 = nyu_2080ti.clone(feature=200, feature_2d_oc=200, backbone_2d_name="tf_efficientnet_b7_ns",
                            context_prior=True)

PROJECT_RES = ["1", "2", "4", "8"]  # scripts/train.py:125-134 with project_1_{2,4,8} = true
