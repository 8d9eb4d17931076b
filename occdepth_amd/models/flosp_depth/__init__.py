"""Dataset name -> FLoSP-Depth geometry table (same lookup the reference exposes as `flosp_depth_conf_map`)."""
from . import flosp_depth_conf_kitti as _kitti
from . import flosp_depth_conf_nyu as _nyu

flosp_depth_conf_map = {"kitti": _kitti.flosp_depth_conf, "NYU": _nyu.flosp_depth_conf}
