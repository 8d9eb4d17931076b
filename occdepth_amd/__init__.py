"""occdepth_amd -- MI355X-native implementation of OccDepth's forward hot path.

`occdepth_amd.models` mirrors the reference's `occdepth.models` nn.Module surface;
`occdepth_amd.hip` is the ctypes binding of libocc_hip.so (include/occdepth_amd.h);
`occdepth_amd.build` compiles the library for gfx950.
"""
__version__ = "0.1.0"
