"""NYUv2: 4.8 m x 4.8 m x 2.88 m scene at 0.08 m, depth bins of 0.08 m up to 10 m, 480 x 640 images."""
from .geometry import make_conf

flosp_depth_conf = make_conf(image_hw=(480, 640), x=(0, 4.8, 0.08), y=(-2.4, 2.4, 0.08), z=(0, 2.88, 0.08),
                             depth=(0, 10, 0.08), mid_channels=128)
final_dim = flosp_depth_conf["final_dim"]
