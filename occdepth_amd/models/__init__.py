"""MI355X-native mirror of the reference's `occdepth.models` nn.Module surface.

Same class names, constructor signatures, sub-module attribute names (hence
identical state_dict keys) and forward contracts as /root/reference/occdepth/models;
in eval mode every 3-D operator and the 2D->3D lift run as hand-written HIP
kernels from libocc_hip.so, in training mode (autograd) through ATen on ROCm.
"""
