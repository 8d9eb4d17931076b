"""SemanticKITTI: 51.2 m x 51.2 m x 6.4 m scene at 0.2 m, depth bins of 0.5 m from 2 m to 54 m, 370 x 1220 images."""
from .geometry import make_conf

flosp_depth_conf = make_conf(image_hw=(370, 1220), x=(0, 51.2, 0.2), y=(-25.6, 25.6, 0.2), z=(-2, 4.4, 0.2),
                             depth=(2.0, 54.0, 0.5), mid_channels=128)
final_dim = flosp_depth_conf["final_dim"]
