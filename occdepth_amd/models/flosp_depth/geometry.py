"""FLoSP-Depth frustum / voxel geometry per dataset (the numbers of the reference's
occdepth/models/flosp_depth/flosp_depth_conf_{kitti,nyu}.py, one table instead of two modules).

bounds are (min, max, step) in metres along the lidar/world x, y, z axes and along the camera depth axis."""


def make_conf(image_hw, x, y, z, depth, mid_channels, feature_channels=64, stride=8):
    return {
        "x_bound": list(x), "y_bound": list(y), "z_bound": list(z), "d_bound": list(depth),
        "final_dim": tuple(image_hw),
        "output_channels": feature_channels,
        "downsample_factor": stride,
        "depth_net_conf": {"in_channels": feature_channels, "mid_channels": mid_channels},
        "disc_cfg": {"mode": "LID"},          # linear-increasing depth discretisation
        "agg_voxel_mode": "mean",             # average of the cameras that see a voxel
    }
