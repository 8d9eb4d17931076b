"""ctypes binding of libocc_hip.so (include/occdepth_amd.h).

torch is used here only as the owner of device memory and of the HIP stream:
every function takes torch CUDA(=HIP) tensors, passes raw pointers + sizes
through the C ABI and launches on torch's current stream.  There is NO fallback:
a missing / unloadable library or a CPU tensor raises.
"""
import ctypes
import os
from ctypes import POINTER, Structure, c_char, c_char_p, c_double, c_float, c_int32, c_int64, c_uint8, c_void_p

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libocc_hip.so")

MAX_VIEWS = 4
MAX_SCALES = 4
ACT_NONE, ACT_RELU, ACT_SIGMOID, ACT_RELU_PRE = 0, 1, 2, 3
ABI_VERSION = 14   # 14: occd_graph_replace_memsets (captured memset nodes -> fill kernels); 13: occd_gemm_args.bias_n / stride_bias_n (column bias: CRP relation-logit convolutions on K16), occd_gemm_f32x3_splitk (K21), occd_se_gate_set_fused; 11: occd_gemm_args.act_a (sigmoid on A: CRP products on K16), peer-memory exchanges (occd_ipc_*, occd_bn_*_small_xchg), occd_stem_conv3x3_nchw, occd_depthnet_gate, occd_plane_reduce / occd_se_bwd; 10: strided (channels-last) ssc loss / confusion passes, occd_relation_bce_*, occd_depth_bce_*, occd_flosp_sample_bwd (N1 kernels); 9: occd_gemm_args.res / scale_k (project convolutions on K16), occd_conv3d_fwd_phases; 8: occd_gemm_f32x3 (K16, row-major float32 GEMM with the 3-way bf16 split), K2s3 behind occd_conv3d_bf16_fwd dtype 2; 7: occd_lift_proj_fwd (fused projection + frustum sample + lift), occd_pack_weights_bf16x3 + split mode of occd_conv3d_bf16_fwd; 6: K2b / K8b bf16-MFMA convolution forward + weight gradient, BN kernels; 5: K11s split-K hints, occd_upconv_gather_nchw (K12); 4: K11 pointwise GEMM, SE gate, depthwise pool/backward, softmax, lift backward + xcd_mode/feat_bstride; 3: K10

_c_float_p = POINTER(c_float)


class Conv3dArgs(Structure):
    _fields_ = [
        ("inp", c_void_p), ("wpk", c_void_p), ("bias", c_void_p), ("res1", c_void_p), ("res2", c_void_p),
        ("out", c_void_p),
        ("batch", c_int32),
        ("X", c_int32), ("Y", c_int32), ("Z", c_int32),
        ("cin", c_int32), ("in_cs", c_int32), ("in_coff", c_int32),
        ("cout", c_int32), ("out_cs", c_int32), ("out_coff", c_int32),
        ("res1_cs", c_int32), ("res1_coff", c_int32), ("res2_cs", c_int32), ("res2_coff", c_int32),
        ("kx", c_int32), ("ky", c_int32), ("kz", c_int32),
        ("sx", c_int32), ("sy", c_int32), ("sz", c_int32),
        ("dx", c_int32), ("dy", c_int32), ("dz", c_int32),
        ("px", c_int32), ("py", c_int32), ("pz", c_int32),
        ("Xo", c_int32), ("Yo", c_int32), ("Zo", c_int32),
        ("OX", c_int32), ("OY", c_int32), ("OZ", c_int32),
        ("o_stride_x", c_int32), ("o_stride_y", c_int32), ("o_stride_z", c_int32),
        ("o_off_x", c_int32), ("o_off_y", c_int32), ("o_off_z", c_int32),
        ("act_in", c_int32), ("act_out", c_int32), ("cout_store", c_int32), ("tile_hint", c_int32),
    ]


class WgradArgs(Structure):
    _fields_ = [("x", c_void_p), ("gy", c_void_p), ("dw", c_void_p), ("workspace", c_void_p),
                ("workspace_floats", c_int64), ("batch", c_int32)] + \
        [(n, c_int32) for n in ("X", "Y", "Z", "cin", "x_cs", "x_coff", "Xo", "Yo", "Zo", "cout", "gy_cs", "gy_coff",
                                "kx", "ky", "kz", "sx", "sy", "sz", "dx", "dy", "dz", "px", "py", "pz")]


class FlospArgs(Structure):
    _fields_ = [
        ("depth", c_void_p), ("trans", c_void_p), ("proj", c_void_p), ("ida", c_void_p), ("grids", c_void_p),
        ("out", c_void_p),
        ("batch", c_int32), ("n_cams", c_int32),
        ("D", c_int32), ("h", c_int32), ("w", c_int32),
        ("A", c_int32), ("Bdim", c_int32), ("C", c_int32),
        ("img_w", c_float), ("img_h", c_float), ("depth_min", c_float), ("depth_max", c_float),
        ("mean_mode", c_int32),
    ]


class LiftArgs(Structure):
    _fields_ = [
        ("feat", (c_void_p * MAX_VIEWS) * MAX_SCALES),
        ("feat_h", c_int32 * MAX_SCALES), ("feat_w", c_int32 * MAX_SCALES),
        ("feat_cs", c_int32 * MAX_SCALES), ("scale_div", c_int32 * MAX_SCALES),
        ("n_scales", c_int32), ("n_views", c_int32), ("batch", c_int32), ("C", c_int32),
        ("pix", c_void_p), ("fov", c_void_p),
        ("N", c_int32), ("P", c_int32),
        ("depth_scale", c_void_p), ("scale_const", c_float),
        ("dimA", c_int32), ("dimB", c_int32), ("dimC", c_int32),
        ("row_a", c_int64), ("row_b", c_int64), ("row_c", c_int64),
        ("out", c_void_p), ("out_rows", c_int64), ("out_cs", c_int32),
        ("feat_bstride", (c_int64 * MAX_VIEWS) * MAX_SCALES), ("xcd_mode", c_int32),
    ]


class LiftProjArgs(Structure):
    _fields_ = [("lift", LiftArgs), ("cam_E", c_void_p), ("cam_k", c_void_p), ("voxel_size", ctypes.c_double),
                ("origin", c_float * 3),
                ("img_w", c_int32), ("img_h", c_int32), ("frustum", FlospArgs)]


class FlospBwdArgs(Structure):
    _fields_ = [("fwd", FlospArgs), ("gout", c_void_p), ("gdepth", c_void_p), ("workspace", c_void_p),
                ("workspace_bytes", c_int64)]


class BneckArgs(Structure):
    _fields_ = [("x", c_void_p), ("y", c_void_p), ("o2", c_void_p), ("w", c_void_p)] + \
        [(n, c_int32) for n in ("batch", "X", "Y", "Z", "C", "P", "x_cs", "x_coff", "y_cs", "y_coff", "d0", "d1", "d2")]


class RowsGemmArgs(Structure):
    _fields_ = [("a", c_void_p), ("w", c_void_p), ("bias", c_void_p), ("res", c_void_p), ("out", c_void_p), ("rows", c_int64)] + \
        [(n, c_int32) for n in ("K", "N", "a_cs", "a_coff", "out_cs", "out_coff", "res_cs", "res_coff", "w_stride", "act_in",
                                "act_out")]


class GemmArgs(Structure):
    _fields_ = [("A", c_void_p), ("B", c_void_p), ("C", c_void_p), ("bias", c_void_p)] + \
        [(n, c_int32) for n in ("M", "N", "K", "batch")] + \
        [(n, c_int64) for n in ("lda", "ldb", "ldc", "stride_a", "stride_b", "stride_c")] + \
        [("act", c_int32), ("slope", c_float), ("tile_hint", c_int32), ("pre", c_int32), ("res", c_void_p), ("scale_k", c_void_p),
         ("act_a", c_int32), ("bias_n", c_void_p), ("stride_bias_n", c_int64)]


class WinoArgs(Structure):
    _fields_ = [("x", c_void_p), ("upk", c_void_p), ("shift", c_void_p), ("res", c_void_p), ("y", c_void_p)] + \
        [(n, c_int32) for n in ("batch", "cin", "cout", "H", "W", "act", "res_first", "tile_hint")] + [("slope", c_float)]


class PwArgs(Structure):
    _fields_ = [("x", c_void_p), ("wpk", c_void_p), ("shift", c_void_p), ("gate", c_void_p), ("res", c_void_p),
                ("y", c_void_p), ("N", c_int64), ("batch", c_int32), ("cin", c_int32), ("cout", c_int32),
                ("act", c_int32), ("tile_hint", c_int32), ("slope", c_float), ("out_nhwc_cs", c_int32)]


class LiftBwdArgs(Structure):
    _fields_ = [("fwd", LiftArgs), ("gout", c_void_p), ("gfeat", (c_void_p * MAX_VIEWS) * MAX_SCALES), ("gdepth", c_void_p)]


class BnArgs(Structure):
    _fields_ = [(n, c_void_p) for n in ("x", "gy", "y", "res", "out", "out2", "a", "b", "mean", "invstd", "k1", "k2", "k3",
                                        "partial")] + \
        [("rows", c_int64), ("S", c_int64), ("batch", c_int32), ("C", c_int32), ("cw", c_int32), ("dtype", c_int32),
         ("layout", c_int32)] + \
        [(n, c_int32) for n in ("x_cs", "x_coff", "gy_cs", "gy_coff", "y_cs", "y_coff", "res_cs", "res_coff", "out_cs",
                                "out_coff", "out2_cs", "out2_coff", "act", "res_first")] + \
        [("slope", c_float), ("nblk", c_int32)]


class ProfRow(Structure):
    _fields_ = [("tag", c_char * 64), ("launches", c_int64), ("ms", c_double), ("flops", c_double),
                ("bytes", c_double)]


EXPORTS = {
    "occd_abi_version": (c_int32, []),
    "occd_strerror": (c_char_p, [c_int32]),
    "occd_conv3d_fwd": (c_int32, [POINTER(Conv3dArgs), c_void_p]),
    "occd_conv3d_fwd_phases": (c_int32, [POINTER(Conv3dArgs), c_int32, c_void_p]),
    "occd_conv3d_bf16_fwd_phases": (c_int32, [POINTER(Conv3dArgs), c_int32, c_int32, c_void_p]),
    "occd_packed_weight_floats": (c_int64, [c_int32, c_int32, c_int32]),
    "occd_pack_weights": (c_int32, [c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32,
                                    c_int32, c_void_p]),
    "occd_flosp_sample_fwd": (c_int32, [POINTER(FlospArgs), c_void_p]),
    "occd_rows_gemm_fwd": (c_int32, [POINTER(RowsGemmArgs), c_void_p]),
    "occd_gemm_f32x3": (c_int32, [POINTER(GemmArgs), c_void_p]),
    "occd_gemm_f32x3_nt": (c_int32, [POINTER(GemmArgs), c_void_p]),
    "occd_gemm_f32x3_nt_splits": (c_int32, [c_int32, c_int32, c_int32, c_int32]),
    "occd_gemm_x3_packed_elems": (c_int64, [c_int32, c_int32]),
    "occd_gemm_x3_pack": (c_int32, [c_void_p, c_void_p, c_int32, c_int32, c_int64, c_int32, c_int32, c_int64, c_void_p]),
    "occd_rows_gemm_packed_floats": (c_int64, [c_int32, c_int32]),
    "occd_rows_gemm_pack": (c_int32, [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_void_p]),
    "occd_bottleneck3d_weight_floats": (c_int64, [c_int32, c_int32]),
    "occd_bottleneck3d_fwd": (c_int32, [POINTER(BneckArgs), c_void_p]),
    "occd_lift_proj_fwd": (c_int32, [POINTER(LiftProjArgs), c_void_p]),
    "occd_lift_fwd": (c_int32, [POINTER(LiftArgs), c_void_p]),
    "occd_lift_bwd": (c_int32, [POINTER(LiftBwdArgs), c_void_p]),
    "occd_nchw_to_nhwc": (c_int32, [c_void_p, c_void_p, c_int32, c_int32, c_int64, c_int32, c_void_p]),
    "occd_nhwc_to_nchw": (c_int32, [c_void_p, c_void_p, c_int32, c_int32, c_int64, c_int32, c_int32, c_void_p]),
    "occd_softmax_channels": (c_int32, [c_void_p, c_void_p, c_int64, c_int32, c_int32, c_int32, c_int32, c_int32,
                                        c_int32, c_void_p]),
    "occd_affine_act_nchw": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int64,
                                       c_int32, c_float, c_int32, c_void_p]),
    "occd_dwconv2d_nchw": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p] + [c_int32] * 11 + [c_void_p]),
    "occd_upsample_bilinear_cat_nchw": (c_int32, [c_void_p, c_void_p, c_void_p] + [c_int32] * 7 + [c_void_p]),
    "occd_swish_bwd": (c_int32, [c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    "occd_upconv_gather_skip_nchw": (c_int32, [c_void_p] * 5 + [c_int32] * 7 + [c_int64, c_int64, ctypes.c_float, c_void_p]),
    "occd_upconv_gather_nchw": (c_int32, [c_void_p, c_void_p] + [c_int32] * 6 + [c_int64, c_int64, c_void_p]),
    "occd_wino_input_transform_nchw": (c_int32, [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32,
                                                 c_void_p]),
    "occd_wino_output_transform_nchw": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32,
                                                  c_int32, c_int32, c_int32, c_int32, c_int32, c_float, c_int32, c_void_p]),
    "occd_wino_packed_floats": (c_int64, [c_int32, c_int32]),
    "occd_wino_pack_weights": (c_int32, [c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_void_p]),
    "occd_wino_conv3x3_fwd": (c_int32, [POINTER(WinoArgs), c_void_p]),
    "occd_softmax_nchw": (c_int32, [c_void_p, c_void_p, c_int32, c_int32, c_int64, c_void_p]),
    "occd_dwconv2d_bwd_data_nchw": (c_int32, [c_void_p, c_void_p, c_void_p] + [c_int32] * 10 + [c_void_p]),
    "occd_dwconv2d_bwd_weight_workspace_floats": (c_int64, [c_int32] * 5),
    "occd_dwconv2d_bwd_weight_nchw": (c_int32, [c_void_p] * 4 + [c_int32] * 10 + [c_void_p]),
    "occd_dwconv2d_pool_blocks": (c_int32, [c_int32, c_int32]),
    "occd_dwconv2d_pool_nchw": (c_int32, [c_void_p] * 6 + [c_int32] * 11 + [c_int64, c_void_p]),
    "occd_se_gate": (c_int32, [c_void_p] * 7 + [c_int32, c_int32, c_int32, c_int32, c_int64, c_void_p]),
    "occd_pw_packed_floats": (c_int64, [c_int32, c_int32]),
    "occd_pw_pack_weights": (c_int32, [c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_void_p]),
    "occd_pw_conv_fwd": (c_int32, [POINTER(PwArgs), c_void_p]),
    "occd_project_voxels": (c_int32, [c_void_p, c_void_p, c_void_p, c_double] + [c_int32] * 5
                            + [c_void_p, c_void_p, c_void_p, c_void_p]),
    "occd_argmax_channels": (c_int32, [c_void_p, c_int64, c_int32, c_int32, c_int32, c_void_p, c_void_p, c_void_p]),
    "occd_cascade_tail_fwd": (c_int32, [c_void_p, c_void_p, c_void_p] + [c_int32] * 8 + [c_void_p]),
    "occd_conv3d_wgrad_workspace_floats": (c_int64, [POINTER(WgradArgs)]),
    "occd_conv3d_wgrad": (c_int32, [POINTER(WgradArgs), c_void_p]),
    "occd_packed_weight_bf16_elems": (c_int64, [c_int32, c_int32, c_int32]),
    "occd_pack_weights_bf16": (c_int32, [c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32,
                                         c_int32, c_void_p]),
    "occd_conv3d_bf16_fwd": (c_int32, [POINTER(Conv3dArgs), c_int32, c_void_p]),
    "occd_upsample_bilinear_cat_nhwc": (c_int32, [c_void_p, c_void_p, c_void_p] + [c_int32] * 7 + [c_void_p]),
    "occd_upsample_bilinear_cat_nhwc_rows": (c_int32, [c_void_p, c_void_p, c_void_p] + [c_int32] * 8 + [c_void_p]),
    "occd_upsample_bilinear_nhwc_bwd": (c_int32, [c_void_p, c_void_p] + [c_int32] * 7 + [c_void_p]),
    "occd_pack_weights_gather": (c_int32, [c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int64, c_int64,
                                           c_void_p, c_void_p]),
    "occd_pack_weights_bf16_gather": (c_int32, [c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int64, c_int64,
                                                c_void_p, c_void_p]),
    "occd_pack_weights_bf16x3_gather": (c_int32, [c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int64, c_int64,
                                                c_void_p, c_void_p]),
    "occd_pack_weights_bf16x3": (c_int32, [c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32,
                                           c_int32, c_void_p]),
    "occd_conv3d_wgrad_bf16_workspace_floats": (c_int64, [POINTER(WgradArgs), c_int32]),
    "occd_conv3d_wgrad_bf16": (c_int32, [POINTER(WgradArgs), c_int32, c_void_p]),
    "occd_bn_blocks": (c_int32, [POINTER(BnArgs)]),
    "occd_bn_stats": (c_int32, [POINTER(BnArgs), c_void_p]),
    "occd_bn_stats_combine": (c_int32, [POINTER(BnArgs), c_void_p, c_void_p]),
    "occd_bn_finish": (c_int32, [c_void_p, c_int32, c_float, c_float] + [c_void_p] * 9 + [c_void_p]),
    "occd_bn_apply": (c_int32, [POINTER(BnArgs), c_void_p]),
    "occd_bn_bwd_reduce": (c_int32, [POINTER(BnArgs), c_void_p]),
    "occd_bn_bwd_combine": (c_int32, [c_void_p, c_int32, c_int32, c_void_p, c_void_p]),
    "occd_bn_bwd_finish": (c_int32, [c_void_p, c_void_p, c_int32] + [c_void_p] * 9 + [c_void_p]),
    "occd_bn_bwd_apply": (c_int32, [POINTER(BnArgs), c_void_p]),
    "occd_bn_stats_finish": (c_int32, [POINTER(BnArgs), c_void_p, c_float, c_float] + [c_void_p] * 9 + [c_void_p]),
    "occd_bn_bwd_combine_finish": (c_int32, [c_void_p, c_int32, c_int32] + [c_void_p] * 9 + [c_void_p]),
    "occd_bn_small_ok": (c_int32, [POINTER(BnArgs)]),
    "occd_bn_fwd_small": (c_int32, [POINTER(BnArgs), c_void_p, c_float, c_float] + [c_void_p] * 9 + [c_void_p]),
    "occd_bn_bwd_small": (c_int32, [POINTER(BnArgs), c_void_p, c_void_p, c_void_p]),
    "occd_bn_xchg_mailbox_bytes": (c_int64, [c_int32, c_int32]),
    "occd_bn_fwd_small_xchg": (c_int32, [POINTER(BnArgs), c_void_p, c_float, c_float] + [c_void_p] * 9 +
                               [POINTER(c_void_p), c_int32, c_int32, c_int32, c_int32, c_void_p, c_void_p]),
    "occd_bn_bwd_small_xchg": (c_int32, [POINTER(BnArgs), c_void_p, c_void_p, c_void_p, POINTER(c_void_p), c_int32, c_int32,
                                         c_int32, c_int32, c_void_p, c_void_p]),
    "occd_ssc_stats_len": (c_int64, [c_int32, c_int32]),
    "occd_ssc_loss_stats_fwd": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_int64,
                                          c_int32, c_int32, c_void_p]),
    "occd_ssc_loss_stats_bwd": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int32,
                                          c_int64, c_int32, c_int32, c_void_p]),
    "occd_ssc_confusion": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_int64, c_void_p]),
    "occd_ssc_loss_stats_fwd_strided": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_int64,
                                                  c_int32, c_int32, c_int64, c_int64, c_int64, c_void_p]),
    "occd_ssc_loss_stats_bwd_strided": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int32,
                                                  c_int64, c_int32, c_int32] + [c_int64] * 6 + [c_int32, c_void_p]),
    "occd_ssc_confusion_strided": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_int64, c_int64,
                                             c_int64, c_int64, c_void_p]),
    "occd_relation_bce_stats": (c_int32, [c_void_p, c_void_p, c_int32, c_void_p, c_int64, c_int32] + [c_int64] * 6 + [c_void_p]),
    "occd_relation_bce_grad": (c_int32, [c_void_p, c_void_p, c_int32, c_void_p, c_void_p, c_int64, c_int32] + [c_int64] * 6 +
                               [c_void_p]),
    "occd_depth_bce_stats": (c_int32, [c_void_p, c_void_p, c_void_p, c_int64] + [c_int32] * 6 + [c_int64, c_float, c_float,
                                                                                               c_void_p]),
    "occd_depth_bce_grad": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64] + [c_int32] * 6 +
                            [c_int64, c_float, c_float, c_void_p]),
    "occd_flosp_sample_bwd": (c_int32, [POINTER(FlospBwdArgs), c_void_p]),
    "occd_stem_conv3x3_nchw": (c_int32, [c_void_p] * 5 + [c_int32] * 10 + [c_void_p]),
    "occd_ipc_mailbox_bytes": (c_int64, [c_int32, c_int64]),
    "occd_ipc_mailbox_create": (c_int32, [c_int64, POINTER(c_void_p), c_void_p]),
    "occd_ipc_mailbox_open": (c_int32, [c_void_p, POINTER(c_void_p)]),
    "occd_ipc_mailbox_close": (c_int32, [c_void_p]),
    "occd_ipc_mailbox_free": (c_int32, [c_void_p]),
    "occd_ipc_allreduce": (c_int32, [c_void_p, c_void_p, c_int64, c_int32, POINTER(c_void_p), c_int32, c_int32, c_int64,
                                     c_int32, c_void_p, c_void_p]),
    "occd_plane_reduce": (c_int32, [c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_void_p]),
    "occd_se_bwd": (c_int32, [c_void_p] * 13 + [c_int32, c_int32, c_int32, c_int64, c_void_p]),
    "occd_depthnet_gate": (c_int32, [c_void_p, c_void_p, c_int64, c_float] + [c_void_p] * 9 + [c_int32, c_int32, c_void_p]),
    "occd_prof_enable": (c_int32, [c_int32]),
    "occd_gemm_f32x3_splitk_plan": (c_int32, [c_int32, c_int32, c_int32, c_int32, POINTER(c_int32), POINTER(c_int32),
                                              POINTER(c_int32), POINTER(c_int64)]),
    "occd_gemm_f32x3_splitk": (c_int32, [POINTER(GemmArgs), c_int32, c_int32, c_int32, c_void_p, c_int64, c_void_p]),
    "occd_se_gate_set_fused": (c_int32, [c_int32]),
    "occd_graph_replace_memsets": (c_int32, [c_void_p]),
    "occd_prof_set_tag": (c_int32, [c_char_p]),
    "occd_prof_report": (c_int32, [POINTER(ProfRow), c_int32]),
}

_lib = None
_PROFILING = False


def load():
    """Load libocc_hip.so (raises if it has not been built: there is no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: build it with `python -m occdepth_amd.build` "
            "(the MI355X HIP kernels are the only implementation of this path)")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in EXPORTS.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    if lib.occd_abi_version() != ABI_VERSION:
        raise RuntimeError("libocc_hip.so ABI version mismatch; rebuild it")
    _lib = lib
    return lib


def _check(code, what):
    if code != 0:
        msg = load().occd_strerror(code).decode()
        raise RuntimeError(f"{what} failed: {msg} (code {code})")


def _ptr(t, name="tensor"):
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError(f"{name} must live on the GPU (the HIP kernels have no CPU path)")
    if not t.is_contiguous():
        raise RuntimeError(f"{name} must be contiguous")
    return t.data_ptr()


def _f32(t, name):
    if t.dtype != torch.float32:
        raise RuntimeError(f"{name} must be float32, got {t.dtype}")
    return _ptr(t, name)


def _stream():
    return torch.cuda.current_stream().cuda_stream


# ----------------------------------------------------------------------------- voxel tensors
class Vox:
    """Channels-last voxel tensor view: buf is (B, X, Y, Z, cs) float32 (bfloat16 for the bf16-storage variants of
    K2b / K8b); the logical tensor is buf[..., coff:coff+C].  Pads in [C, round8(C)) must be zero."""

    __slots__ = ("buf", "C", "coff")

    def __init__(self, buf, C, coff=0):
        assert buf.dim() == 5 and buf.dtype in (torch.float32, torch.bfloat16)
        self.buf, self.C, self.coff = buf, int(C), int(coff)

    @property
    def cs(self):
        return self.buf.shape[4]

    @property
    def dims(self):
        return tuple(self.buf.shape[1:4])

    @property
    def batch(self):
        return self.buf.shape[0]

    def ncdhw(self):
        """Logical (B, C, X, Y, Z) view (channels_last_3d strides, no copy)."""
        return self.buf[..., self.coff:self.coff + self.C].permute(0, 4, 1, 2, 3)

    @staticmethod
    def empty(batch, dims, C, device, cs=None, dtype=torch.float32):
        cs = cs if cs is not None else round_up(C, 8)
        return Vox(torch.empty((batch,) + tuple(dims) + (cs,), device=device, dtype=dtype), C)

    @staticmethod
    def from_ncdhw(x):
        """(B, C, X, Y, Z) float32 -> channels-last Vox with zeroed channel pad (HIP transpose)."""
        B, C = x.shape[0], x.shape[1]
        dims = tuple(x.shape[2:])
        v = Vox.empty(B, dims, C, x.device)
        xc = x.contiguous()
        S = dims[0] * dims[1] * dims[2]
        _check(load().occd_nchw_to_nhwc(_f32(xc, "x"), _f32(v.buf, "out"), B, C, S, v.cs, _stream()),
               "occd_nchw_to_nhwc")
        return v


def round_up(x, m):
    return (x + m - 1) // m * m


# ----------------------------------------------------------------------------- K2
def packed_weight_floats(cout, cin, taps):
    n = load().occd_packed_weight_floats(cout, cin, taps)
    if n <= 0:
        raise RuntimeError("occd_packed_weight_floats: bad shape")
    return n


def pack_weights(w, scale=None, layout=0):
    """w: (Cout, Cin, kx, ky, kz) [layout 0] / (K, N) row-major GEMM B operand [layout 2]."""
    if layout == 2:
        cin, cout = w.shape
        k = (1, 1, 1)
    else:
        cout, cin = w.shape[0], w.shape[1]
        k = tuple(w.shape[2:])
    w = w.contiguous()
    out = torch.empty(packed_weight_floats(cout, cin, k[0] * k[1] * k[2]), device=w.device, dtype=torch.float32)
    sc = scale.contiguous() if scale is not None else None
    _check(load().occd_pack_weights(_f32(w, "w"), _f32(sc, "scale") if sc is not None else None,
                                    _f32(out, "wpk"), cout, cin, k[0], k[1], k[2], layout, _stream()),
           "occd_pack_weights")
    return out


def pack_weights_gather(w, cout, cin, s_co, s_ci, tap_ofs, kernel=None, bf16=False):
    """Packed image of the operator W'[co][ci][tap] = w.flat[co * s_co + ci * s_ci + tap_ofs[tap]] (a transposed / flipped /
    tap-subset view of the dense float32 tensor `w`) for `conv3d` (bf16=False) or `conv3d_bf16`; one launch, no temporaries.
    `kernel` (the (kx, ky, kz) the taps enumerate) is only read by the CPU emulation."""
    if w.dtype != torch.float32 or not w.is_contiguous():
        raise RuntimeError("pack_weights_gather needs a contiguous float32 source tensor")
    n = len(tap_ofs)
    ofs = (c_int32 * n)(*[int(o) for o in tap_ofs])
    if max(ofs) + (cout - 1) * s_co + (cin - 1) * s_ci >= w.numel():
        raise RuntimeError("pack_weights_gather: view exceeds the source tensor")
    if bf16 == "x3":                       # hi | mid | lo images for conv3d_bf16(..., split3=True)
        out = torch.empty(3 * load().occd_packed_weight_bf16_elems(cout, cin, n), device=w.device, dtype=torch.bfloat16)
        _check(load().occd_pack_weights_bf16x3_gather(_f32(w, "w"), None, _ptr(out, "wpk"), cout, cin, n, s_co, s_ci,
                                                      ctypes.cast(ofs, c_void_p), _stream()), "occd_pack_weights_bf16x3_gather")
    elif bf16:
        out = torch.empty(load().occd_packed_weight_bf16_elems(cout, cin, n), device=w.device, dtype=torch.bfloat16)
        _check(load().occd_pack_weights_bf16_gather(_f32(w, "w"), None, _ptr(out, "wpk"), cout, cin, n, s_co, s_ci,
                                                    ctypes.cast(ofs, c_void_p), _stream()), "occd_pack_weights_bf16_gather")
    else:
        out = torch.empty(packed_weight_floats(cout, cin, n), device=w.device, dtype=torch.float32)
        _check(load().occd_pack_weights_gather(_f32(w, "w"), None, _f32(out, "wpk"), cout, cin, n, s_co, s_ci,
                                               ctypes.cast(ofs, c_void_p), _stream()), "occd_pack_weights_gather")
    return out


def _conv3d_args(x, wpk_ptr, bias, cout, kernel, out, stride, dilation, padding, res1, res2, act_in, act_out, out_pos,
                 o_stride, o_off, cin, tile_hint, ptr):
    a = Conv3dArgs()
    a.inp, a.wpk = ptr(x.buf, "x"), wpk_ptr
    a.bias = _f32(bias, "bias") if bias is not None else None
    a.res1 = ptr(res1.buf, "res1") if res1 is not None else None
    a.res2 = ptr(res2.buf, "res2") if res2 is not None else None
    a.out = ptr(out.buf, "out")
    a.batch = x.batch
    a.X, a.Y, a.Z = x.dims
    a.cin = x.C if cin is None else cin
    a.in_cs, a.in_coff = x.cs, x.coff
    a.cout, a.out_cs, a.out_coff = cout, out.cs, out.coff
    if res1 is not None:
        a.res1_cs, a.res1_coff = res1.cs, res1.coff
    if res2 is not None:
        a.res2_cs, a.res2_coff = res2.cs, res2.coff
    a.kx, a.ky, a.kz = kernel
    a.sx, a.sy, a.sz = stride
    a.dx, a.dy, a.dz = dilation
    a.px, a.py, a.pz = padding
    a.OX, a.OY, a.OZ = out.dims
    if out_pos is None:
        out_pos = tuple((n + 2 * p - d * (k - 1) - 1) // s + 1
                        for n, p, d, k, s in zip(x.dims, padding, dilation, kernel, stride))
    a.Xo, a.Yo, a.Zo = out_pos
    a.o_stride_x, a.o_stride_y, a.o_stride_z = o_stride
    a.o_off_x, a.o_off_y, a.o_off_z = o_off
    a.act_in, a.act_out = act_in, act_out
    a.cout_store = min(round_up(cout, 8), out.cs - out.coff, round_up(cout, 32))
    a.tile_hint = tile_hint
    if _PROFILING:
        set_tag("%d>%d k%d%d%d s%d%d%d d%d%d%d @%dx%dx%d" % ((a.cin, a.cout) + tuple(kernel) + tuple(stride)
                                                              + tuple(dilation) + tuple(x.dims)))
    return a


def conv3d(x, wpk, bias, cout, kernel, out, stride=(1, 1, 1), dilation=(1, 1, 1), padding=(0, 0, 0),
           res1=None, res2=None, act_in=ACT_NONE, act_out=ACT_NONE, out_pos=None, o_stride=(1, 1, 1),
           o_off=(0, 0, 0), cin=None, tile_hint=0):
    """out = act_out(conv(act_in(x)) + bias + res1 + res2) on Vox tensors (see include/occdepth_amd.h)."""
    a = _conv3d_args(x, _f32(wpk, "wpk"), bias, cout, kernel, out, stride, dilation, padding, res1, res2, act_in, act_out,
                     out_pos, o_stride, o_off, cin, tile_hint, _f32)
    _check(load().occd_conv3d_fwd(ctypes.byref(a), _stream()), "occd_conv3d_fwd")
    return out


def conv3d_phases(x, phases, bias, cout, out, res1=None, act_out=ACT_NONE, out_pos=None, o_stride=(1, 1, 1), split3=False):
    """The sub-pixel phases of one transposed convolution as ONE launch: K2 (occd_conv3d_fwd_phases) or, split3, K2b with the
    3-way bf16 split (occd_conv3d_bf16_fwd_phases).
    phases: 1 / 2 / 4 / 8 tuples (wpk, kernel, o_off) -- the packed weights of the phase's tap subset (pack_weights, or
    pack_weights_bf16(split3=True)), its extent and where its voxels land in `out`; padding 0, stride / dilation 1, everything
    else shared."""
    n = len(phases)
    arr = (Conv3dArgs * n)()
    for i, (wpk, kernel, o_off) in enumerate(phases):
        a = _conv3d_args(x, _ptr(wpk, "wpk") if split3 else _f32(wpk, "wpk"), bias, cout, kernel, out, (1, 1, 1), (1, 1, 1),
                         (0, 0, 0), res1, None, ACT_NONE, act_out, out_pos, o_stride, o_off, None, 0, _f32)
        ctypes.memmove(ctypes.byref(arr, i * ctypes.sizeof(Conv3dArgs)), ctypes.byref(a), ctypes.sizeof(Conv3dArgs))
    if _PROFILING:
        set_tag("%d>%d k333 s222 transposed: %d phases @%dx%dx%d" % ((x.C, cout, n) + tuple(x.dims)))
    if split3:
        _check(load().occd_conv3d_bf16_fwd_phases(arr, n, 2, _stream()), "occd_conv3d_bf16_fwd_phases")
    else:
        _check(load().occd_conv3d_fwd_phases(arr, n, _stream()), "occd_conv3d_fwd_phases")
    return out


# ----------------------------------------------------------------------------- K2b / K8b (bf16 MFMA)
def _act_ptr(dtype):
    def ptr(t, name):
        if t.dtype != dtype:
            raise RuntimeError(f"{name} must be {dtype} like the other activation tensors of the launch, got {t.dtype}")
        return _ptr(t, name)
    return ptr


def _storage_code(dtype):
    if dtype == torch.float32:
        return 0
    if dtype == torch.bfloat16:
        return 1
    raise RuntimeError(f"the bf16-MFMA kernels take float32 or bfloat16 activations, got {dtype}")


def pack_weights_bf16(w, scale=None, layout=0, split3=False):
    """fp32 master weights -> the bf16 fragment image of K2b (shapes / layouts as `pack_weights`); split3: the three images
    hi | mid | lo of the 3-way split experiment (conv3d_bf16(..., split3=True))."""
    if layout == 2:
        cin, cout = w.shape
        k = (1, 1, 1)
    else:
        cout, cin = w.shape[0], w.shape[1]
        k = tuple(w.shape[2:])
    w = w.detach().float().contiguous()
    n = load().occd_packed_weight_bf16_elems(cout, cin, k[0] * k[1] * k[2])
    if n <= 0:
        raise RuntimeError("occd_packed_weight_bf16_elems: bad shape")
    out = torch.empty(3 * n if split3 else n, device=w.device, dtype=torch.bfloat16)
    sc = scale.detach().float().contiguous() if scale is not None else None
    fn = load().occd_pack_weights_bf16x3 if split3 else load().occd_pack_weights_bf16
    _check(fn(_f32(w, "w"), _f32(sc, "scale") if sc is not None else None, _ptr(out, "wpk"),
              cout, cin, k[0], k[1], k[2], layout, _stream()), "occd_pack_weights_bf16")
    return out


def conv3d_bf16(x, wpk, bias, cout, kernel, out, stride=(1, 1, 1), dilation=(1, 1, 1), padding=(0, 0, 0),
                res1=None, res2=None, act_in=ACT_NONE, act_out=ACT_NONE, out_pos=None, o_stride=(1, 1, 1),
                o_off=(0, 0, 0), cin=None, tile_hint=0, split3=False):
    """K2b: `conv3d` on the bf16 matrix pipe (fp32 accumulate).  x / out / res1 / res2 are all float32 Vox tensors
    (converted to bf16 while staging) or all bfloat16 ones; wpk = pack_weights_bf16(w).
    split3 (float32 tensors, wpk = pack_weights_bf16(w, split3=True)): the 3-way split experiment, float32-level accuracy."""
    if wpk.dtype != torch.bfloat16:
        raise RuntimeError("conv3d_bf16 needs the bf16 weight image of pack_weights_bf16")
    if split3 and x.buf.dtype != torch.float32:
        raise RuntimeError("the 3-way split takes float32 tensors")
    ptr = _act_ptr(x.buf.dtype)
    a = _conv3d_args(x, _ptr(wpk, "wpk"), bias, cout, kernel, out, stride, dilation, padding, res1, res2, act_in, act_out,
                     out_pos, o_stride, o_off, cin, tile_hint, ptr)
    _check(load().occd_conv3d_bf16_fwd(ctypes.byref(a), 2 if split3 else _storage_code(x.buf.dtype), _stream()),
           "occd_conv3d_bf16_fwd")
    return out


GEMM_ACT = {None: 0, "none": 0, "swish": 1, "leaky": 2}


class GemmPacked:
    """A static GEMM operand (weights) split once into its three bf16 terms in MFMA fragment order (occd_gemm_x3_pack):
    role "a" = the left operand (M, K) / (batch, M, K), role "b" = the right operand (K, N) / (batch, K, N)."""

    def __init__(self, w, role):
        if role not in ("a", "b") or w.dim() not in (2, 3) or w.dtype != torch.float32 or not w.is_cuda:
            raise RuntimeError("GemmPacked: a 2-D / 3-D float32 GPU tensor and role 'a' or 'b'")
        w = w.detach().contiguous()
        self.role, self.batch = role, (w.shape[0] if w.dim() == 3 else None)
        if role == "a":
            self.rows, self.K = w.shape[-2], w.shape[-1]
        else:
            self.K, self.rows = w.shape[-2], w.shape[-1]
        self.shape = tuple(w.shape)
        self.per = load().occd_gemm_x3_packed_elems(self.rows, self.K)
        if self.per <= 0:
            raise RuntimeError("occd_gemm_x3_packed_elems: bad shape")
        nb = self.batch or 1
        self.buf = torch.empty(nb * self.per, device=w.device, dtype=torch.bfloat16)
        _check(load().occd_gemm_x3_pack(w.data_ptr(), self.buf.data_ptr(), self.rows, self.K, w.shape[-1], 0 if role == "a" else 1,
                                        nb, w.shape[-2] * w.shape[-1], _stream()), "occd_gemm_x3_pack")

    def dim(self):
        return len(self.shape)


def gemm_x3_supported(a, b):
    """Shapes / layouts K16 takes: a (M, K) or (batch, M, K), b (batch, K, N) or (K, N); float32, innermost stride 1,
    K % 8 == 0, 16-byte aligned rows of a, N >= 4.  Either operand may be a `GemmPacked` (pre-split weights)."""
    if isinstance(a, GemmPacked) and isinstance(b, GemmPacked):
        return False
    for t, role in ((a, "a"), (b, "b")):
        if isinstance(t, GemmPacked):
            if t.role != role:
                return False
        elif t.dtype != torch.float32 or not t.is_cuda or t.dim() not in (2, 3) or t.stride(-1) != 1:
            return False
    K = a.K if isinstance(a, GemmPacked) else a.shape[-1]
    Kb = b.K if isinstance(b, GemmPacked) else b.shape[-2]
    N = b.rows if isinstance(b, GemmPacked) else b.shape[-1]
    if K != Kb or K % 8 or N < 4:
        return False
    if not isinstance(a, GemmPacked):
        if a.stride(-2) % 4 or a.stride(-2) < K or a.data_ptr() % 16 or (a.dim() == 3 and a.stride(0) % 4):
            return False
    if not isinstance(b, GemmPacked) and b.stride(-2) < N:
        return False
    nb_a = (a.batch if isinstance(a, GemmPacked) else (a.shape[0] if a.dim() == 3 else None))
    nb_b = (b.batch if isinstance(b, GemmPacked) else (b.shape[0] if b.dim() == 3 else None))
    return nb_a is None or nb_b is None or nb_a == nb_b


def gemm_x3(a, b, bias=None, act=None, slope=0.01, out=None, tile_hint=0, plain_bf16=False, res=None, k_scale=None,
            sigmoid_a=False, bias_n=None):
    """K16 (occd_gemm_f32x3): out[i] = act(a[i] @ (b[i] * k_scale[i][:, None]) + bias[:, None]) + res[i] in float32-level
    accuracy on the bf16 matrix pipe (k_scale: (batch, K), the squeeze-excite gate of a project convolution; res: laid out
    like out, the block's skip connection).
    a: (M, K) shared over the batch, or (batch, M, K); b: (K, N) or (batch, K, N); out: (batch, M, N) ((M, N) when neither
    operand is batched).  Tensor operands may be strided views as long as the innermost stride is 1; a static operand may be
    given as `GemmPacked(w, "a" / "b")` (split once, read straight from L2).  plain_bf16: operands rounded to ONE bf16 term
    (the bf16 training mode: bf16 MFMA, fp32 storage and accumulate) instead of the split.  sigmoid_a: a -> sigmoid(a) while
    it is staged (the CRP's relation products)."""
    if not gemm_x3_supported(a, b):
        raise RuntimeError("gemm_x3: unsupported operands")
    pa, pb = isinstance(a, GemmPacked), isinstance(b, GemmPacked)
    nb_a = a.batch if pa else (a.shape[0] if a.dim() == 3 else None)
    nb_b = b.batch if pb else (b.shape[0] if b.dim() == 3 else None)
    batch = nb_a or nb_b or 1
    M = a.rows if pa else a.shape[-2]
    K = a.K if pa else a.shape[-1]
    N = b.rows if pb else b.shape[-1]
    squeeze = nb_a is None and nb_b is None
    dev = a.buf.device if pa else a.device
    if out is None:
        out = torch.empty((batch, M, N), device=dev, dtype=torch.float32)
    elif out.dtype != torch.float32 or out.stride(-1) != 1 or out.shape[-2:] != (M, N) or not out.is_cuda:
        raise RuntimeError("gemm_x3: bad output tensor")
    elif not ((out.dim() == 3 and out.shape[0] == batch) or (out.dim() == 2 and squeeze)):
        # the kernel trusts (batch, stride_c): a (1, M, N) or 2-D `out` under a batched operand would be written past its
        # end / once per batch item on top of itself
        raise RuntimeError("gemm_x3: out must be (batch, M, N) with batch = %d (2-D only when neither operand is batched)"
                           % batch)
    q = GemmArgs()
    q.A = a.buf.data_ptr() if pa else a.data_ptr()       # (strided views: the strides travel as lda / ldb / ldc)
    q.B = b.buf.data_ptr() if pb else b.data_ptr()
    q.C = out.data_ptr()
    q.bias = _f32(bias, "bias") if bias is not None else None
    if bias is not None and (bias.numel() != M or not bias.is_contiguous()):
        raise RuntimeError("gemm_x3: bias must be M contiguous floats")
    q.M, q.N, q.K, q.batch = M, N, K, batch
    q.lda = K if pa else a.stride(-2)
    q.ldb = N if pb else b.stride(-2)
    q.ldc = out.stride(-2)
    q.stride_a = (a.per if nb_a else 0) if pa else (a.stride(0) if a.dim() == 3 else 0)
    q.stride_b = (b.per if nb_b else 0) if pb else (b.stride(0) if b.dim() == 3 else 0)
    q.stride_c = out.stride(0) if out.dim() == 3 else 0
    q.act, q.slope, q.tile_hint = GEMM_ACT[act], slope, tile_hint
    q.pre = 1 if pa else 2 if pb else (3 if plain_bf16 else 0)
    if plain_bf16 and (pa or pb):
        raise RuntimeError("gemm_x3: plain_bf16 takes float32 tensor operands")
    if res is not None:
        if res.dtype != torch.float32 or res.shape != out.shape or res.stride() != out.stride():
            raise RuntimeError("gemm_x3: res must be laid out like out")
        q.res = res.data_ptr()
    if k_scale is not None:
        if pb or k_scale.dtype != torch.float32 or tuple(k_scale.shape) != (batch, K) or not k_scale.is_contiguous():
            raise RuntimeError("gemm_x3: k_scale must be (batch, K) contiguous floats and b a float32 tensor")
        q.scale_k = k_scale.data_ptr()
    if sigmoid_a:
        if pa or plain_bf16:
            raise RuntimeError("gemm_x3: sigmoid_a takes a float32 tensor A and the split arithmetic")
        q.act_a = 1
    if bias_n is not None:                                   # one value per COLUMN: (N,) shared or (batch, N)
        if bias_n.dtype != torch.float32 or not bias_n.is_cuda or not bias_n.is_contiguous() or \
                tuple(bias_n.shape) not in ((N,), (batch, N)):
            raise RuntimeError("gemm_x3: bias_n must be (N,) or (batch, N) contiguous floats")
        q.bias_n, q.stride_bias_n = bias_n.data_ptr(), (N if bias_n.dim() == 2 else 0)
    if _PROFILING:
        set_tag("%dx%dx%d b%d" % (M, N, K, batch))
    _check(load().occd_gemm_f32x3(ctypes.byref(q), _stream()), "occd_gemm_f32x3")
    return out[0] if squeeze and out.dim() == 3 else out


def gemm_x3_splitk_plan(M, N, K, batch):
    """(k16_per_z, nz, row_ranges, workspace_floats) the library proposes for K21 on this problem."""
    per, nz, rr, ws = c_int32(), c_int32(), c_int32(), c_int64()
    _check(load().occd_gemm_f32x3_splitk_plan(M, N, K, batch, ctypes.byref(per), ctypes.byref(nz), ctypes.byref(rr),
                                              ctypes.byref(ws)), "occd_gemm_f32x3_splitk_plan")
    return per.value, nz.value, rr.value, ws.value


_SPLITK_WS = {}        # device -> workspace tensor (grown on demand; stream-ordered reuse: the reduce launch follows its GEMM)


def gemm_x3_splitk(a, b, bias=None, act=None, slope=0.01, out=None, res=None, k_scale=None, plan=None):
    """K21 (occd_gemm_f32x3_splitk): out[i] = act(a @ (b[i] * k_scale[i][:, None]) + bias[:, None]) + res[i] for SKINNY long-K
    problems (the MBConv project convolutions of the 1/16 and 1/32 stages): K cut over the grid, float32 partial tiles in a
    workspace, deterministic second launch.  a: GemmPacked role "a" (shared over the batch); b: (batch, K, N) / (K, N)
    float32, innermost stride 1; plan: (k16_per_z, nz, row_ranges) or None for the library's proposal."""
    if not isinstance(a, GemmPacked) or a.role != "a" or a.batch is not None:
        raise RuntimeError("gemm_x3_splitk: a must be a shared GemmPacked left operand")
    if b.dtype != torch.float32 or not b.is_cuda or b.dim() not in (2, 3) or b.stride(-1) != 1:
        raise RuntimeError("gemm_x3_splitk: b must be a float32 GPU tensor with innermost stride 1")
    batch = b.shape[0] if b.dim() == 3 else 1
    M, K, N = a.rows, a.K, b.shape[-1]
    if b.shape[-2] != K:
        raise RuntimeError("gemm_x3_splitk: inner dimensions differ")
    squeeze = b.dim() == 2
    if out is None:
        out = torch.empty((batch, M, N), device=b.device, dtype=torch.float32)
    elif out.dtype != torch.float32 or out.stride(-1) != 1 or tuple(out.shape) != (batch, M, N) or not out.is_cuda:
        raise RuntimeError("gemm_x3_splitk: out must be a (batch, M, N) float32 GPU tensor")
    per, nz, rr, _ = gemm_x3_splitk_plan(M, N, K, batch) if plan is None else (tuple(plan) + (0,))
    need = batch * nz * M * round_up(N, 32)
    ws = _SPLITK_WS.get(b.device)
    if ws is None or ws.numel() < need:
        ws = torch.empty(max(need, 1 << 22), device=b.device, dtype=torch.float32)
        _SPLITK_WS[b.device] = ws
    q = GemmArgs()
    q.A, q.B, q.C = a.buf.data_ptr(), b.data_ptr(), out.data_ptr()
    if bias is not None:
        if bias.numel() != M or not bias.is_contiguous():
            raise RuntimeError("gemm_x3_splitk: bias must be M contiguous floats")
        q.bias = _f32(bias, "bias")
    q.M, q.N, q.K, q.batch = M, N, K, batch
    q.lda, q.ldb, q.ldc = K, b.stride(-2), out.stride(-2)
    q.stride_a, q.stride_b, q.stride_c = 0, (b.stride(0) if b.dim() == 3 else 0), out.stride(0)
    q.act, q.slope, q.tile_hint, q.pre = GEMM_ACT[act], slope, 0, 1
    if res is not None:
        if res.dtype != torch.float32 or res.shape != out.shape or res.stride() != out.stride():
            raise RuntimeError("gemm_x3_splitk: res must be laid out like out")
        q.res = res.data_ptr()
    if k_scale is not None:
        if k_scale.dtype != torch.float32 or tuple(k_scale.shape) != (batch, K) or not k_scale.is_contiguous():
            raise RuntimeError("gemm_x3_splitk: k_scale must be (batch, K) contiguous floats")
        q.scale_k = k_scale.data_ptr()
    if _PROFILING:
        set_tag("%dx%dx%d b%d z%d" % (M, N, K, batch, nz))
    _check(load().occd_gemm_f32x3_splitk(ctypes.byref(q), per, nz, rr, ws.data_ptr(), ws.numel(), _stream()),
           "occd_gemm_f32x3_splitk")
    return out[0] if squeeze else out


def gemm_x3_nt(a, b, tile_hint=0, splits=None, reduce=True, plain_bf16=False):
    """K16t (occd_gemm_f32x3_nt): a[i] @ b[i].T for a (batch, M, K), b (batch, N, K) -- both k-contiguous, any alignment /
    K: the weight gradient of a pointwise convolution (gy (Cout, HW) x (Cin, HW)).  The reduction dimension is split over
    `splits` workgroup groups (default: the library's proposal); reduce=True returns the (M, N) sum over batch and splits
    (what a weight gradient is), reduce=False the (batch, splits, M, N) partial products."""
    for t, nm in ((a, "a"), (b, "b")):
        if t.dim() != 3 or t.dtype != torch.float32 or not t.is_cuda or t.stride(-1) != 1:
            raise RuntimeError(f"gemm_x3_nt: {nm} must be a (batch, rows, K) float32 GPU tensor with unit innermost stride")
    batch, M, K = a.shape
    if b.shape[0] != batch or b.shape[2] != K:
        raise RuntimeError("gemm_x3_nt: operand shapes do not match")
    N = b.shape[1]
    if splits is None:
        splits = load().occd_gemm_f32x3_nt_splits(M, N, K, batch)
    steps = (K + 31) // 32
    per = (steps + splits - 1) // splits
    splits = (steps + per - 1) // per                 # every split owns at least one K step
    out = torch.empty((batch, splits, M, N), device=a.device, dtype=torch.float32)
    q = GemmArgs()
    q.A, q.B, q.C = a.data_ptr(), b.data_ptr(), out.data_ptr()
    q.M, q.N, q.K, q.batch = M, N, K, batch
    q.lda, q.ldb, q.ldc = a.stride(1), b.stride(1), N
    q.stride_a, q.stride_b, q.stride_c = a.stride(0), b.stride(0), M * N
    q.tile_hint, q.act, q.pre = tile_hint, splits, (3 if plain_bf16 else 0)
    if _PROFILING:
        set_tag("%dx%dx%d b%d z%d" % (M, N, K, batch, splits))
    _check(load().occd_gemm_f32x3_nt(ctypes.byref(q), _stream()), "occd_gemm_f32x3_nt")
    return out.sum((0, 1)) if reduce else out


class _PwConvFn(torch.autograd.Function):
    """Training: a pointwise (1x1, stride 1, no bias) convolution of an NCHW tensor on K16 / K16t -- forward W . x, data
    gradient W^T . gy (both the NN kernel, float32-level accuracy on the bf16 matrix pipe) and weight gradient gy . x^T (the
    NT kernel, on the tensors as they lie) -- instead of MIOpen / rocBLAS behind `aten::convolution_backward`: the 55 MBConv
    blocks' expand / project convolutions were 13 ms of the 106 ms bf16-mode step (profiles/r04_train_step_bf16_aten_ops.txt)."""

    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, x, w):
        from . import autograd3d
        B, C, H, W = x.shape
        xc = x if x.is_contiguous() else x.contiguous()
        ctx.save_for_backward(xc, w)
        ctx.plain = bool(autograd3d.BF16_MFMA) and PW_TRAIN == "bf16"      # (opt-in) one bf16 product instead of the 3-way split
        return gemm_x3(w.detach().reshape(w.shape[0], C), xc.view(B, C, H * W), plain_bf16=ctx.plain).view(B, w.shape[0], H, W)

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, gy):
        x, w = ctx.saved_tensors
        B, C, H, W = x.shape
        Co = w.shape[0]
        g = gy.float()
        g = (g if g.is_contiguous() else g.contiguous()).view(B, Co, H * W)
        gx = gw = None
        if ctx.needs_input_grad[0]:
            gx = gemm_x3(w.detach().reshape(Co, C).t().contiguous(), g, plain_bf16=ctx.plain).view(B, C, H, W)
        if ctx.needs_input_grad[1]:
            gw = gemm_x3_nt(g, x.view(B, C, H * W), plain_bf16=ctx.plain).view(Co, C, 1, 1)
        return gx, gw


# OCCDEPTH_TRAIN_PW_GEMM: 1 (default since round 5) = the encoder's pointwise (expand / project) convolutions run forward, data
# gradient and weight gradient on K16 / K16t with the 3-way split (float32-accurate) in training -- the same time as MIOpen /
# rocBLAS on these small shapes (round 4: 153.1 vs 152 ms fp32, 107.7 vs 106.5 ms bf16 mode; round 5: 104.1 / 105.0 vs 104.6 /
# 105.6 ms), but ~13 ms of library work (the largest non-repo share of the step) becomes in-repo and deterministic, and the full
# training-step parity test holds unchanged; 0 = ATen (A/B); bf16 = plain bf16 operands in the bf16-MFMA mode (105.5 vs 106.7 ms)
# -- NOT the default: through the 55 MBConv blocks of a random-init B7 the bf16 rounding of 110 chained pointwise convolutions
# turns the encoder's gradient directions to cosine 0.2 - 0.35 against the real reference (0.93 - 0.97 with the encoder in
# fp32; tests/test_train_step.py::test_train_step_full_config2_matches_reference_gpu).
PW_TRAIN = os.environ.get("OCCDEPTH_TRAIN_PW_GEMM", "1")


def pw_conv_autograd_ok(conv, x):
    """The pointwise convolutions `_PwConvFn` takes: 1x1, stride 1, one group, no bias, channel counts K16 accepts."""
    if PW_TRAIN not in ("1", "bf16"):
        return False
    # float32 inputs only, except under autocast (where _PwConvFn casts): a plain half-precision input falls back to ATen
    # instead of reaching gemm_x3 with operands it rejects
    dt_ok = x.dtype == torch.float32 or (torch.is_autocast_enabled() and x.dtype in (torch.bfloat16, torch.float16))
    return (GEMM_X3 and x.is_cuda and x.dim() == 4 and dt_ok
            and conv.kernel_size == (1, 1) and conv.stride == (1, 1) and conv.groups == 1 and conv.bias is None
            and conv.dilation == (1, 1) and conv.in_channels % 8 == 0 and conv.out_channels % 8 == 0
            and x.shape[2] * x.shape[3] >= 4 and x.shape[0] <= 65535)


def pw_conv_autograd(x, w):
    return _PwConvFn.apply(x, w)


# K16 instead of the library GEMMs in the eval path of the 2-D network (tap GEMMs, Winograd-domain products, expand 1x1
# convolutions, the merged conv_head o conv2).  OCCDEPTH_GEMM_X3=0 restores torch.matmul / torch.bmm / F.conv2d for A/B.
GEMM_X3 = os.environ.get("OCCDEPTH_GEMM_X3", "1") == "1"


# static operands pre-split once (GemmPacked) instead of split on the fly: measured NO gain (tap GEMMs +-3 %, Winograd domain
# 122 against 140 TF/s, frame 21.77 against 21.48 ms; profiles/r04_gemm_x3_v2_presplit.txt) -- the fragment stream from L2 costs
# what the split arithmetic saved -- so it stays an option (tested), off by default
GEMM_X3_PACK = os.environ.get("OCCDEPTH_GEMM_X3_PACK", "0") == "1"


# K16p (round 5): the SHORT-K left operands (expand convolutions, the 1/1 and 1/2 tap GEMMs: K <= 848, >= 256 rows) ARE pre-split, for
# the panel-stationary kernel that keeps a 64-column panel of B over the whole K in LDS (csrc/gemm_x3.hip).  OCCDEPTH_GEMM_X3_PANEL=0
# -> the barrier-phased K16 on float32 operands as in round 4.
GEMM_X3_PANEL = os.environ.get("OCCDEPTH_GEMM_X3_PANEL", "1") == "1"
PANEL_MAX_K, PANEL_MIN_ROWS = 848, 256


def _panel_operand(w, role):
    # (any K: beyond K16p's reach the image feeds K16's pre-split form on the large launches, `_presplit_launch`)
    return GEMM_X3_PANEL and role == "a" and w.dim() == 2 and w.shape[0] >= PANEL_MIN_ROWS


def _panel_launch(pa, b):
    """The host-side mirror of occd_gemm_f32x3's hint-0 rule for K16p: K <= 352, or (K <= 848) a few-pixel launch."""
    if pa.K <= 352:
        return True
    if pa.K > PANEL_MAX_K:
        return False
    n, batch = b.shape[-1], (b.shape[0] if b.dim() == 3 else 1)
    return -(-n // 64) * batch * -(-(-(-pa.rows // 32)) // 8) <= 512


def _presplit_launch(pa, b):
    """K16 on the pre-split weight image (PRE = 1, fragments requested a full 32-k step ahead since late round 5): the long-K
    GEMMs that fill the chip with 256 x 128 tiles -- the tap GEMMs of the 1/4, 1/8 and 1/16 levels: 425 -> 368, 397 -> 355, 438 ->
    422 us against float32 operands (profiles/r05_gemm_panel.txt); small launches and the Winograd-domain products do not gain."""
    n, batch = b.shape[-1], (b.shape[0] if b.dim() == 3 else 1)
    return pa.K >= 512 and -(-pa.rows // 256) * -(-n // 128) * batch >= 320


def matmul_operand(w, role):
    """A static operand (weights) in the form hip.matmul wants it: (the float32 tensor, its GemmPacked image or None)."""
    w = w.detach().float().contiguous()
    if GEMM_X3 and w.is_cuda and (GEMM_X3_PACK or _panel_operand(w, role)):
        return w, GemmPacked(w, role)
    return w, None


# Rows of a K16 result that start on 128-byte boundaries are WRITTEN 2.2x faster than rows of an odd pixel count (H * W floats:
# every 128-byte store segment straddles two cache lines that another workgroup completes later; tools/store_pattern.hip:
# 650 MB in 121 us against 262 - 290 us; profiles/r05_store_alignment.txt).  `padded_rows` hands out such a result buffer:
# (..., rows, n) as a view of (..., rows, n rounded up to 32 floats).  OCCDEPTH_PAD_ROWS=0 -> dense results as in round 4.
PAD_ROWS = os.environ.get("OCCDEPTH_PAD_ROWS", "1") == "1"


def padded_rows(shape, device):
    """An uninitialised float32 tensor of `shape` whose rows (last dimension) lie 128-byte aligned: a view of a buffer with the
    last dimension rounded up to 32 floats (stride(-1) == 1; the leading dimensions dense over the padded rows)."""
    n = int(shape[-1])
    pitch = round_up(n, 32) if PAD_ROWS else n
    return torch.empty(tuple(shape[:-1]) + (pitch,), device=device, dtype=torch.float32)[..., :n]


def matmul(a, b, bias=None, act=None, slope=0.01, res=None, k_scale=None, out=None):
    """act(a @ (b * k_scale[..., None]) + bias[:, None]) + res for the eval path: K16 when it applies, else the library + plain
    tensor ops.  a / b may be the (tensor, GemmPacked or None) pair of `matmul_operand`; out: an optional (batch, M, N) result
    tensor (rows may be padded: `padded_rows`)."""
    ta, pa = a if isinstance(a, tuple) else (a, None)
    tb, pb = b if isinstance(b, tuple) else (b, None)
    if GEMM_X3:
        # (a panel-only image serves the plain epilogue; with a residual / a k scale the float32 operand goes to K16)
        a_img = pa is not None and (GEMM_X3_PACK or (res is None and k_scale is None and
                                                     (_panel_launch(pa, tb) or _presplit_launch(pa, tb))))
        xa, xb = (pa if a_img else ta), (pb if pb is not None and k_scale is None else tb)
        if gemm_x3_supported(xa, xb):
            return gemm_x3(xa, xb, bias=bias, act=act, slope=slope, res=res, k_scale=k_scale, out=out)
        if gemm_x3_supported(ta, tb):
            return gemm_x3(ta, tb, bias=bias, act=act, slope=slope, res=res, k_scale=k_scale, out=out)
    if k_scale is not None:
        tb = tb * k_scale.unsqueeze(-1)
    y = torch.matmul(ta, tb)
    if bias is not None:
        y = y + bias.view(-1, 1)
    if act == "swish":
        y = y * torch.sigmoid(y)
    elif act == "leaky":
        y = torch.nn.functional.leaky_relu(y, slope)
    y = y + res if res is not None else y
    if out is not None:
        out.copy_(y)
        return out
    return y


def c32x3_eligible(x, cout, kernel, out, stride=(1, 1, 1), dilation=(1, 1, 1), padding=(0, 0, 0), res1=None, res2=None,
                   act_in=ACT_NONE, act_out=ACT_NONE, out_pos=None, o_stride=(1, 1, 1), o_off=(0, 0, 0), cin=None,
                   tile_hint=0):
    """True when `conv3d_bf16(..., split3=True)` takes the sliding-window split kernel K2s3 (the host-side mirror of
    `c32_geometry` + the Z / dilation condition of `try_conv3d_c32_slide_x3` in csrc/conv3d_c32p.hip): the full-resolution head convolutions."""
    d = tuple(dilation)
    if tuple(kernel) != (3, 3, 3) or tuple(stride) != (1, 1, 1) or d[0] != d[1] or d[0] != d[2] or not 1 <= d[0] <= 3:
        return False
    if tuple(padding) != d or (out_pos is not None and tuple(out_pos) != tuple(x.dims)) or tuple(o_stride) != (1, 1, 1) or \
            tuple(o_off) != (0, 0, 0):
        return False
    c = x.C if cin is None else cin
    if round_up(c, 8) != 32 or cout > 32 or x.coff + 32 > x.cs or act_in == ACT_SIGMOID or tile_hint != 0:
        return False
    if x.buf.dtype != torch.float32 or x.cs % 4 or x.coff % 4:
        return False
    X, Y, Z = x.dims
    # Z = 32, or (round 5) any multiple of 32: the z-halo form of the kernel (dilation 3 of a Z > 32 volume on six-row y tiles)
    if Z % 32 or tuple(out.dims) != (X, Y, Z):
        return False
    return x.batch * X * ((Y + 7) // 8) * (Z // 32) >= 512 and x.batch * X * Y * Z * x.cs < 2 ** 32


_wgrad_ws = {}


def conv3d_wgrad(x, gy, cin, cout, kernel, stride=(1, 1, 1), dilation=(1, 1, 1), padding=(0, 0, 0)):
    """dW (cout, cin, kx, ky, kz) of a convolution y = conv(x, W) from channels-last x and gy = dL/dy (K8)."""
    a = WgradArgs()
    a.x, a.gy = _f32(x.buf, "x"), _f32(gy.buf, "gy")
    a.batch = x.batch
    a.X, a.Y, a.Z = x.dims
    a.cin, a.x_cs, a.x_coff = cin, x.cs, x.coff
    a.Xo, a.Yo, a.Zo = gy.dims
    a.cout, a.gy_cs, a.gy_coff = cout, gy.cs, gy.coff
    a.kx, a.ky, a.kz = kernel
    a.sx, a.sy, a.sz = stride
    a.dx, a.dy, a.dz = dilation
    a.px, a.py, a.pz = padding
    need = load().occd_conv3d_wgrad_workspace_floats(ctypes.byref(a))
    if need <= 0:
        raise RuntimeError(f"occd_conv3d_wgrad_workspace_floats failed (code {need})")
    dev = x.buf.device
    ws = _wgrad_ws.get(dev)                      # one growing scratch buffer per device (stream-ordered reuse)
    if ws is None or ws.numel() < need:
        ws = torch.empty(need, dtype=torch.float32, device=dev)
        _wgrad_ws[dev] = ws
    dw = torch.empty((cout, cin) + tuple(kernel), dtype=torch.float32, device=dev)
    a.dw, a.workspace, a.workspace_floats = dw.data_ptr(), ws.data_ptr(), ws.numel()
    if _PROFILING:
        set_tag("%d>%d k%d%d%d s%d%d%d d%d%d%d @%dx%dx%d" % ((cin, cout) + tuple(kernel) + tuple(stride)
                                                              + tuple(dilation) + tuple(x.dims)))
    _check(load().occd_conv3d_wgrad(ctypes.byref(a), _stream()), "occd_conv3d_wgrad")
    return dw


def conv3d_wgrad_bf16(x, gy, cin, cout, kernel, stride=(1, 1, 1), dilation=(1, 1, 1), padding=(0, 0, 0)):
    """K8b: `conv3d_wgrad` on the bf16 matrix pipe (fp32 accumulate, fp32 result); x / gy both float32 or both bfloat16."""
    code = _storage_code(x.buf.dtype)
    ptr = _act_ptr(x.buf.dtype)
    a = WgradArgs()
    a.x, a.gy = ptr(x.buf, "x"), ptr(gy.buf, "gy")
    a.batch = x.batch
    a.X, a.Y, a.Z = x.dims
    a.cin, a.x_cs, a.x_coff = cin, x.cs, x.coff
    a.Xo, a.Yo, a.Zo = gy.dims
    a.cout, a.gy_cs, a.gy_coff = cout, gy.cs, gy.coff
    a.kx, a.ky, a.kz = kernel
    a.sx, a.sy, a.sz = stride
    a.dx, a.dy, a.dz = dilation
    a.px, a.py, a.pz = padding
    need = load().occd_conv3d_wgrad_bf16_workspace_floats(ctypes.byref(a), code)
    if need <= 0:
        raise RuntimeError(f"occd_conv3d_wgrad_bf16_workspace_floats failed (code {need})")
    dev = x.buf.device
    ws = _wgrad_ws.get(dev)
    if ws is None or ws.numel() < need:
        ws = torch.empty(need, dtype=torch.float32, device=dev)
        _wgrad_ws[dev] = ws
    dw = torch.empty((cout, cin) + tuple(kernel), dtype=torch.float32, device=dev)
    a.dw, a.workspace, a.workspace_floats = dw.data_ptr(), ws.data_ptr(), ws.numel()
    if _PROFILING:
        set_tag("%d>%d k%d%d%d s%d%d%d d%d%d%d @%dx%dx%d" % ((cin, cout) + tuple(kernel) + tuple(stride)
                                                              + tuple(dilation) + tuple(x.dims)))
    _check(load().occd_conv3d_wgrad_bf16(ctypes.byref(a), code, _stream()), "occd_conv3d_wgrad_bf16")
    return dw


# ----------------------------------------------------------------------------- K1
class Frustum:
    """The operands of the FLoSP-Depth frustum sample (occd_flosp_args without its output), so that the sample can run
    either on its own (`flosp_sample`) or inside the fused lift (`lift_proj`)."""

    def __init__(self, depth, trans, proj, ida, voxel_num, final_dim, d_min, d_max, mean_mode=True, grids=None):
        self.depth, self.trans, self.proj, self.ida, self.grids = depth, trans, proj, ida, grids
        self.voxel_num = tuple(int(v) for v in voxel_num)
        self.final_dim, self.d_min, self.d_max, self.mean_mode = final_dim, d_min, d_max, mean_mode

    def fill(self, a):
        B, V, D, h, w = self.depth.shape
        a.depth = _f32(self.depth, "depth")
        if self.grids is None:
            a.trans, a.proj, a.ida = _f32(self.trans, "trans"), _f32(self.proj, "proj"), _f32(self.ida, "ida")
        else:
            a.grids = _f32(self.grids, "grids")
        a.batch, a.n_cams, a.D, a.h, a.w = B, V, D, h, w
        a.A, a.Bdim, a.C = self.voxel_num
        a.img_h, a.img_w = float(self.final_dim[0]), float(self.final_dim[1])
        a.depth_min, a.depth_max = float(self.d_min), float(self.d_max)
        a.mean_mode = 1 if self.mean_mode else 0

    def sample(self):
        B = self.depth.shape[0]
        A, Bd, C = self.voxel_num
        out = torch.empty((B, A * Bd * C), device=self.depth.device, dtype=torch.float32)
        a = FlospArgs()
        self.fill(a)
        a.out = _f32(out, "out")
        _check(load().occd_flosp_sample_fwd(ctypes.byref(a), _stream()), "occd_flosp_sample_fwd")
        return out


def flosp_sample_bwd(fr, gout):
    """d loss / d depth (B, V, D, h, w) of `fr.sample()` given gout (B, A*B*C): the transpose of the frustum sample,
    deterministic (occd_flosp_sample_bwd)."""
    B, V, D, h, w = fr.depth.shape
    gdepth = torch.empty((B, V, D, h, w), device=gout.device, dtype=torch.float32)
    ws = torch.empty(B * V * D * h * w + 1, dtype=torch.int64, device=gout.device)
    q = FlospBwdArgs()
    fr.fill(q.fwd)
    q.gout = _f32(gout, "gout")
    q.gdepth = gdepth.data_ptr()
    q.workspace, q.workspace_bytes = ws.data_ptr(), ws.numel() * 8
    _check(load().occd_flosp_sample_bwd(ctypes.byref(q), _stream()), "occd_flosp_sample_bwd")
    return gdepth


def flosp_sample(depth, trans, proj, ida, voxel_num, final_dim, d_min, d_max, mean_mode=True, grids=None):
    """depth (B, V, D, h, w) -> (B, A*B*C) sampled voxel volume (see occd_flosp_sample_fwd)."""
    return Frustum(depth, trans, proj, ida, voxel_num, final_dim, d_min, d_max, mean_mode, grids).sample()


# workgroup -> XCD placement of the lift (profiles/r02_lift_xcd_modes.txt): 2 = y-blocks per XCD, the HBM fetch of the
# config-2 lift drops from 282 MB (dispatch order) to 174 MB = the distinct bytes
LIFT_XCD_MODE = int(os.environ.get("OCCDEPTH_LIFT_XCD", "2"))


def _lift_args(a, feats, scale_divs, pix, fov, n_dims, row_strides, out, depth_scale, scale_const, xcd_mode):
    S, V = len(feats), len(feats[0])
    B = feats[0][0].shape[0]
    for s in range(S):
        for v in range(V):
            f = feats[s][v]
            if f.dtype != torch.float32 or not f.is_cuda:
                raise RuntimeError("feature rows must be float32 GPU tensors")
            if not f[0].is_contiguous():
                raise RuntimeError("every image of a feature-row tensor must be dense (H, W, cs)")
            a.feat[s][v] = f.data_ptr()
            a.feat_bstride[s][v] = f.stride(0) if f.shape[0] > 1 else 0
        a.feat_h[s], a.feat_w[s] = feats[s][0].shape[1], feats[s][0].shape[2]
        a.feat_cs[s] = feats[s][0].shape[3]
        a.scale_div[s] = int(scale_divs[s])
    a.n_scales, a.n_views, a.batch = S, V, B
    a.C = out.C
    if pix is None:                    # occd_lift_proj_fwd: the kernel projects
        a.N, a.P = int(n_dims[0]) * int(n_dims[1]) * int(n_dims[2]), 1
    else:
        if pix.dtype != torch.int64:
            raise RuntimeError("projected_pix must be int64")
        a.pix = _ptr(pix, "projected_pix")
        fov8 = fov.view(torch.uint8) if fov.dtype == torch.bool else fov
        if fov8.dtype != torch.uint8:
            raise RuntimeError("fov_mask must be bool/uint8")
        a.fov = _ptr(fov8, "fov_mask")
        a.N, a.P = pix.shape[2], pix.shape[3]
    if depth_scale is not None:
        a.depth_scale = _f32(depth_scale, "depth_scale")
    a.scale_const = float(scale_const)
    a.dimA, a.dimB, a.dimC = n_dims
    a.row_a, a.row_b, a.row_c = row_strides
    a.out = _f32(out.buf, "out")
    a.out_rows = out.dims[0] * out.dims[1] * out.dims[2]
    a.out_cs = out.cs
    a.xcd_mode = LIFT_XCD_MODE if xcd_mode is None else int(xcd_mode)
    if out.coff != 0:
        raise RuntimeError("lift output must start at channel 0")


def lift(feats, scale_divs, pix, fov, n_dims, row_strides, out, depth_scale=None, scale_const=100.0, xcd_mode=None):
    """feats[s][v]: (B, H_s, W_s, cs) channels-last maps (dense per image; the batch stride is free); pix (B, V, N, P, 2)
    int64; fov (B, V, N, P) bool.

    Writes out.buf rows (channels-last voxel grid); n -> (a,b,c) over n_dims, row = a*ra + b*rb + c*rc."""
    a = LiftArgs()
    _lift_args(a, feats, scale_divs, pix, fov, n_dims, row_strides, out, depth_scale, scale_const, xcd_mode)
    _check(load().occd_lift_fwd(ctypes.byref(a), _stream()), "occd_lift_fwd")
    return out


class RowsGemmWeights:
    """The fragment-ordered image of a dense (K, N) float32 matrix for `rows_gemm` (occd_rows_gemm_pack)."""

    def __init__(self, w_rows):
        if w_rows.dtype != torch.float32 or w_rows.stride(1) != 1 or not w_rows.is_cuda:
            raise RuntimeError("rows_gemm: the weight matrix must be a float32 GPU tensor with unit column stride")
        self.K, self.N = int(w_rows.shape[0]), int(w_rows.shape[1])
        self.buf = torch.empty(load().occd_rows_gemm_packed_floats(self.K, self.N), device=w_rows.device, dtype=torch.float32)
        _check(load().occd_rows_gemm_pack(w_rows.data_ptr(), self.buf.data_ptr(), self.K, self.N, int(w_rows.stride(0)),
                                          _stream()), "occd_rows_gemm_pack")


def rows_gemm(a, w, out, bias=None, res=None, act_in=ACT_NONE, act_out=ACT_NONE):
    """K15: out[row, :N] = act_out(act_in(a[row, :K]) @ W[:K, :N] + bias (+ res[row, :N])) on float32 Vox rows;
    w: a dense (K, N) float32 device matrix (packed on the fly) or a `RowsGemmWeights`."""
    if not isinstance(w, RowsGemmWeights):
        w = RowsGemmWeights(w)
    K, N = w.K, w.N
    q = RowsGemmArgs()
    q.a, q.w, q.out = _f32(a.buf, "a"), w.buf.data_ptr(), _f32(out.buf, "out")
    q.bias = _f32(bias, "bias") if bias is not None else None
    q.res = _f32(res.buf, "res") if res is not None else None
    q.rows = a.batch * a.dims[0] * a.dims[1] * a.dims[2]
    q.K, q.N = K, N
    q.a_cs, q.a_coff, q.out_cs, q.out_coff = a.cs, a.coff, out.cs, out.coff
    if res is not None:
        q.res_cs, q.res_coff = res.cs, res.coff
    q.w_stride = N
    q.act_in, q.act_out = act_in, act_out
    if _PROFILING:
        set_tag("%d>%d @%d rows" % (K, N, q.rows))
    _check(load().occd_rows_gemm_fwd(ctypes.byref(q), _stream()), "occd_rows_gemm_fwd")
    return out


def rows_gemm_supported(K, N, a, out, res=None):
    """(N % 8: the kernel writes exactly N channels, and the consumers of a Vox read its rows in 8-channel groups)"""
    ok = K % 16 == 0 and N % 8 == 0 and a.cs % 4 == 0 and a.coff % 4 == 0 and out.cs % 4 == 0 and out.coff % 4 == 0
    ok = ok and a.buf.dtype == torch.float32 and a.coff + K <= a.cs and out.coff + N <= out.cs
    return ok and (res is None or (res.cs % 4 == 0 and res.coff % 4 == 0))


def bottleneck3d_supported(C, P, dims):
    """Geometries K14 (occd_bottleneck3d_fwd) is built for; everything else keeps the five-launch K2 form."""
    return P in (16, 32) and C % 16 == 0 and dims[2] in (4, 8, 16)


def bottleneck3d(x, w, P, dilation, out=None):
    """K14: y = one stride-1 DDR Bottleneck3D of the float32 Vox x (BatchNorm folded into `w`, packed as
    occd_bottleneck3d_fwd documents); dilation = (d_z, d_y, d_x) of conv2 / conv3 / conv4."""
    C = x.C
    if w.dtype != torch.float32 or w.numel() != load().occd_bottleneck3d_weight_floats(C, P):
        raise RuntimeError("bottleneck3d: packed weight buffer has the wrong size")
    if out is None:
        out = Vox.empty(x.batch, x.dims, C, x.buf.device)
    X, Y, Z = x.dims
    o2 = torch.empty(x.batch * X * Y * Z * P, device=x.buf.device, dtype=torch.float32)
    a = BneckArgs()
    a.x, a.y, a.o2, a.w = _f32(x.buf, "x"), _f32(out.buf, "y"), _f32(o2, "o2"), _f32(w, "w")
    a.batch, a.X, a.Y, a.Z, a.C, a.P = x.batch, X, Y, Z, C, P
    a.x_cs, a.x_coff, a.y_cs, a.y_coff = x.cs, x.coff, out.cs, out.coff
    a.d0, a.d1, a.d2 = (int(d) for d in dilation)
    if _PROFILING:
        set_tag("%d/%d d%d%d%d @%dx%dx%d" % ((C, P) + tuple(int(d) for d in dilation) + (X, Y, Z)))
    _check(load().occd_bottleneck3d_fwd(ctypes.byref(a), _stream()), "occd_bottleneck3d_fwd")
    return out


def lift_proj(feats, scale_divs, cam_E, cam_k, origin, voxel_size, img_wh, n_dims, row_strides, out, frustum=None,
              scale_const=100.0, xcd_mode=None):
    """The eval lift without its tables (occd_lift_proj_fwd): cam_E (B, V, 4, 4) / cam_k (B, V, 3, 3) float64 device
    tensors (the batch's extrinsics and intrinsics), the voxel grid n_dims = (X, Y, Z) of `voxel_size` metres from
    `origin`; frustum: a `Frustum` or None."""
    q = LiftProjArgs()
    _lift_args(q.lift, feats, scale_divs, None, None, n_dims, row_strides, out, None, scale_const, xcd_mode)
    for t, shape, what in ((cam_E, (4, 4), "cam_E"), (cam_k, (3, 3), "cam_k")):
        if t.dtype != torch.float64 or not t.is_cuda or not t.is_contiguous() or \
                tuple(t.shape) != (q.lift.batch, q.lift.n_views) + shape:
            raise RuntimeError(f"{what} must be a contiguous (batch, views, {shape[0]}, {shape[1]}) float64 GPU tensor")
    q.cam_E, q.cam_k = cam_E.data_ptr(), cam_k.data_ptr()
    q.voxel_size = float(voxel_size)
    for j in range(3):
        q.origin[j] = float(origin[j])
    q.img_w, q.img_h = int(img_wh[0]), int(img_wh[1])
    if frustum is not None:
        frustum.fill(q.frustum)
    if _PROFILING:
        set_tag("%d scales x %d views x %d ch > %d voxels" % (q.lift.n_scales, q.lift.n_views, q.lift.C, q.lift.N))
    _check(load().occd_lift_proj_fwd(ctypes.byref(q), _stream()), "occd_lift_proj_fwd")
    return out


def lift_backward(feats, scale_divs, pix, fov, n_dims, row_strides, out_like, gout, depth_scale=None, scale_const=100.0):
    """Backward of `lift` (single-point patterns): gout (B, X, Y, Z, cs) channels-last d loss / d out ->
    ([[d loss / d feats[s][v] (same shape, channels-last rows)]], d loss / d depth_scale (B, N) or None)."""
    q = LiftBwdArgs()
    _lift_args(q.fwd, feats, scale_divs, pix, fov, n_dims, row_strides, out_like, depth_scale, scale_const, 0)
    q.gout = _f32(gout, "gout")
    grads = []
    for s, per_scale in enumerate(feats):
        row = []
        for v, f in enumerate(per_scale):
            g = torch.zeros_like(f)
            if g.stride() != f.stride():
                raise RuntimeError("gradient map layout differs from the feature map's")
            q.gfeat[s][v] = g.data_ptr()
            row.append(g)
        grads.append(row)
    gd = None
    if depth_scale is not None:
        gd = torch.empty_like(depth_scale)
        q.gdepth = gd.data_ptr()
    _check(load().occd_lift_bwd(ctypes.byref(q), _stream()), "occd_lift_bwd")
    return grads, gd


# ----------------------------------------------------------------------------- helpers
def nchw_to_nhwc(x, cs=None):
    """(B, C, *spatial) -> (B, *spatial, cs) channels-last copy with zero pad."""
    B, C = x.shape[0], x.shape[1]
    sp = tuple(x.shape[2:])
    S = 1
    for d in sp:
        S *= d
    cs = cs if cs is not None else round_up(C, 4)
    out = torch.empty((B,) + sp + (cs,), device=x.device, dtype=torch.float32)
    xc = x.contiguous()
    _check(load().occd_nchw_to_nhwc(_f32(xc, "x"), _f32(out, "out"), B, C, S, cs, _stream()), "occd_nchw_to_nhwc")
    return out


def nhwc_to_nchw(vox):
    """Vox -> dense (B, C, X, Y, Z) tensor (a real copy; Vox.ncdhw() is the zero-copy view)."""
    B = vox.batch
    X, Y, Z = vox.dims
    out = torch.empty((B, vox.C, X, Y, Z), device=vox.buf.device, dtype=torch.float32)
    _check(load().occd_nhwc_to_nchw(_f32(vox.buf, "in"), _f32(out, "out"), B, vox.C, X * Y * Z, vox.cs, vox.coff,
                                    _stream()), "occd_nhwc_to_nchw")
    return out


def softmax_channels(src, dst, n, dst_pad=0):
    """softmax over src's n logical channels -> dst's n logical channels (+ dst_pad zeros) on the same grid."""
    rows = src.batch * src.dims[0] * src.dims[1] * src.dims[2]
    if _PROFILING:
        set_tag("%d @%d rows" % (n, rows))
    _check(load().occd_softmax_channels(_f32(src.buf, "src"), _f32(dst.buf, "dst"), rows, src.cs, src.coff, dst.cs,
                                        dst.coff, n, dst_pad, _stream()), "occd_softmax_channels")
    return dst


# ----------------------------------------------------------------------------- 2-D NCHW helpers
ACT2D = {None: 0, "none": 0, "relu": 1, "swish": 2, "leaky": 3}


def affine_act(x, scale, shift, act=None, slope=0.01, res=None, res_first=False, out=None):
    """y = act(x * scale[c] + shift[c]) (+ res) on (B, C, *spatial) float32; in place unless `out` is given."""
    if not x.is_contiguous():
        x = x.contiguous()
    B, C = x.shape[0], x.shape[1]
    S = x.numel() // (B * C)
    out = x if out is None else out
    if res is not None and not res.is_contiguous():
        res = res.contiguous()
    if _PROFILING:
        set_tag("%d @%dx%d" % (C, B, S))
    _check(load().occd_affine_act_nchw(_f32(x, "x"), _f32(res, "res") if res is not None else None, _f32(out, "out"),
                                       _f32(scale, "scale") if scale is not None else None,
                                       _f32(shift, "shift") if shift is not None else None, B, C, S, ACT2D[act],
                                       float(slope), 1 if res_first else 0, _stream()), "occd_affine_act_nchw")
    return out


_WINO_G = ((1.0, 0.0, 0.0), (0.5, 0.5, 0.5), (0.5, -0.5, 0.5), (0.0, 0.0, 1.0))


def winograd_weights(w):
    """(Cout, Cin, 3, 3) -> U (16, Cin, Cout) with U[4i+j][ci][co] = (G g G^T)[i][j] (float64 arithmetic, once per weight)."""
    G = torch.tensor(_WINO_G, dtype=torch.float64, device=w.device)
    u = torch.einsum("ia,ocab,jb->ijco", G, w.detach().double(), G)
    return u.reshape(16, w.shape[1], w.shape[0]).float().contiguous()


def wino_input_transform(x, ty0=0, ths=None):
    """x (B, C, H, W) -> V (16, B * ths * tw, C) for the tile rows [ty0, ty0 + ths) (default: all)."""
    B, C, H, W = x.shape
    ths = (H + 1) // 2 - ty0 if ths is None else ths
    T = B * ths * ((W + 1) // 2)
    V = torch.empty((16, T, C), device=x.device, dtype=torch.float32)
    xc = x if x.is_contiguous() else x.contiguous()
    if _PROFILING:
        set_tag("%d @%dx%dx%d" % (C, B, H, W))
    _check(load().occd_wino_input_transform_nchw(_f32(xc, "x"), V.data_ptr(), B, C, H, W, ty0, ths, _stream()),
           "occd_wino_input_transform_nchw")
    return V


def wino_output_transform(M, shape, scale=None, shift=None, act=None, slope=0.01, res=None, res_first=False, out=None,
                          ty0=0, ths=None):
    """M (16, B * ths * tw, C) -> rows [2 ty0, 2 (ty0 + ths)) of y (B, C, H, W) (allocated unless `out` is given)."""
    B, C, H, W = shape
    ths = (H + 1) // 2 - ty0 if ths is None else ths
    y = torch.empty(shape, device=M.device, dtype=torch.float32) if out is None else out
    if res is not None and not res.is_contiguous():
        res = res.contiguous()
    if _PROFILING:
        set_tag("%d @%dx%dx%d" % (C, B, H, W))
    _check(load().occd_wino_output_transform_nchw(_f32(M, "M"), _f32(scale, "scale") if scale is not None else None,
                                                  _f32(shift, "shift") if shift is not None else None,
                                                  _f32(res, "res") if res is not None else None, _f32(y, "y"), B, C, H, W,
                                                  ty0, ths, ACT2D[act], float(slope), 1 if res_first else 0, _stream()),
           "occd_wino_output_transform_nchw")
    return y


def conv2d_3x3_winograd(x, U, scale=None, shift=None, act=None, slope=0.01, res=None, res_first=False, strip_rows=None):
    """act(scale * conv3x3(x, g, pad 1) + shift) (+res) with U = winograd_weights(g) (or the `matmul_operand(U, "b")` pair):
    HIP transforms around 16 batched GEMMs (K16; the library with OCCDEPTH_GEMM_X3=0).  `strip_rows` tile rows per pass keep
    V / M cache-sized on big images."""
    B, Cin, H, W = x.shape
    cout = (U[0] if isinstance(U, tuple) else U).shape[2]
    th = (H + 1) // 2
    if strip_rows is None or strip_rows >= th:
        V = wino_input_transform(x)
        return wino_output_transform(matmul(V, U), (B, cout, H, W), scale, shift, act, slope, res, res_first)
    y = torch.empty((B, cout, H, W), device=x.device, dtype=torch.float32)
    xc = x if x.is_contiguous() else x.contiguous()
    for ty0 in range(0, th, strip_rows):
        ths = min(strip_rows, th - ty0)
        M = matmul(wino_input_transform(xc, ty0, ths), U)
        wino_output_transform(M, tuple(y.shape), scale, shift, act, slope, res, res_first, out=y, ty0=ty0, ths=ths)
    return y


def wino_pack_weights(w, scale=None):
    """(Cout, Cin, 3, 3) conv weight (+ per-cout scale, e.g. a folded BatchNorm) -> packed Winograd-domain operand of K10."""
    cout, cin = w.shape[0], w.shape[1]
    if tuple(w.shape[2:]) != (3, 3):
        raise RuntimeError("wino_pack_weights needs a (Cout, Cin, 3, 3) weight")
    n = load().occd_wino_packed_floats(cout, cin)
    if n <= 0:
        raise RuntimeError("occd_wino_packed_floats: bad shape")
    wc = w.detach().float().contiguous()
    upk = torch.empty(n, device=w.device, dtype=torch.float32)
    sc = scale.detach().float().contiguous() if scale is not None else None
    _check(load().occd_wino_pack_weights(_f32(wc, "w"), _f32(sc, "scale") if sc is not None else None,
                                         _f32(upk, "upk"), cout, cin, _stream()), "occd_wino_pack_weights")
    return upk


def conv2d_3x3_fused(x, upk, cout, shift=None, act=None, slope=0.01, res=None, res_first=False, tile_hint=0, out=None):
    """K10: act(conv3x3(x, g * scale, pad 1) + shift) (+ res) in one launch (upk = wino_pack_weights(g, scale))."""
    if not x.is_contiguous():
        x = x.contiguous()
    B, cin, H, W = x.shape
    y = torch.empty((B, cout, H, W), device=x.device, dtype=torch.float32) if out is None else out
    if res is not None and not res.is_contiguous():
        res = res.contiguous()
    a = WinoArgs()
    a.x, a.upk, a.y = _f32(x, "x"), _f32(upk, "upk"), _f32(y, "y")
    a.shift = _f32(shift, "shift") if shift is not None else None
    a.res = _f32(res, "res") if res is not None else None
    a.batch, a.cin, a.cout, a.H, a.W = B, cin, cout, H, W
    a.act, a.res_first, a.tile_hint, a.slope = ACT2D[act], 1 if res_first else 0, int(tile_hint), float(slope)
    if upk.numel() != load().occd_wino_packed_floats(cout, cin):
        raise RuntimeError("packed Winograd weights do not match (cout, cin)")
    if _PROFILING:
        set_tag("%d>%d @%dx%dx%d" % (cin, cout, B, H, W))
    _check(load().occd_wino_conv3x3_fwd(ctypes.byref(a), _stream()), "occd_wino_conv3x3_fwd")
    return y


def pw_pack_weights(w, scale=None):
    """(Cout, Cin[, 1, 1]) 1x1-conv weight (+ per-cout scale) -> packed A operand of K11."""
    cout, cin = w.shape[0], w.shape[1]
    n = load().occd_pw_packed_floats(cout, cin)
    if n <= 0 or w.numel() != cout * cin:
        raise RuntimeError("pw_pack_weights needs a (Cout, Cin[, 1, 1]) weight")
    wc = w.detach().float().reshape(cout, cin).contiguous()
    wpk = torch.empty(n, device=w.device, dtype=torch.float32)
    sc = scale.detach().float().contiguous() if scale is not None else None
    _check(load().occd_pw_pack_weights(_f32(wc, "w"), _f32(sc, "scale") if sc is not None else None, _f32(wpk, "wpk"),
                                       cout, cin, _stream()), "occd_pw_pack_weights")
    return wpk


def conv1x1(x, wpk, cout, shift=None, act=None, slope=0.01, gate=None, res=None, tile_hint=0, out=None, nhwc=False):
    """K11: act(conv1x1(x * gate, w * scale) + shift) (+ res) on (B, Cin, *spatial) float32, one launch.
    nhwc=True: the result is stored pixel-major (B, *spatial, ceil4(Cout)) and returned as its logical (B, Cout, *spatial)
    view (channels-last strides): what the 2D->3D lift gathers from."""
    if not x.is_contiguous():
        x = x.contiguous()
    B, cin = x.shape[0], x.shape[1]
    sp = tuple(x.shape[2:])
    N = 1
    for d in sp:
        N *= d
    ocs = round_up(cout, 4) if nhwc else 0
    if nhwc:
        y = torch.empty((B,) + sp + (ocs,), device=x.device, dtype=torch.float32)
    else:
        y = torch.empty((B, cout) + sp, device=x.device, dtype=torch.float32) if out is None else out
    if res is not None and not res.is_contiguous():
        res = res.contiguous()
    if gate is not None:
        gate = gate.reshape(B, cin)
        if not gate.is_contiguous():
            gate = gate.contiguous()
    if wpk.numel() != load().occd_pw_packed_floats(cout, cin):
        raise RuntimeError("packed 1x1 weights do not match (cout, cin)")
    a = PwArgs()
    a.x, a.wpk, a.y = _f32(x, "x"), _f32(wpk, "wpk"), _f32(y, "y")
    a.shift = _f32(shift, "shift") if shift is not None else None
    a.gate = _f32(gate, "gate") if gate is not None else None
    a.res = _f32(res, "res") if res is not None else None
    a.N, a.batch, a.cin, a.cout = N, B, cin, cout
    a.act, a.tile_hint, a.slope, a.out_nhwc_cs = ACT2D[act], int(tile_hint), float(slope), ocs
    if _PROFILING:
        set_tag("%d>%d @%dx%d" % (cin, cout, B, N))
    _check(load().occd_pw_conv_fwd(ctypes.byref(a), _stream()), "occd_pw_conv_fwd")
    if nhwc:
        nd = len(sp)
        return y[..., :cout].permute(0, nd + 1, *range(1, nd + 1))
    return y


def dwconv2d_same(x, w, scale, shift, stride, act=None):
    """Depthwise conv with TensorFlow SAME padding + per-channel affine + activation, (B, C, H, W) float32."""
    if not x.is_contiguous():
        x = x.contiguous()
    B, C, H, W = x.shape
    k = w.shape[-1]
    Ho, Wo = -(-H // stride), -(-W // stride)
    pad_h = max((Ho - 1) * stride + k - H, 0)
    pad_w = max((Wo - 1) * stride + k - W, 0)
    y = torch.empty((B, C, Ho, Wo), device=x.device, dtype=torch.float32)
    wc = w if w.is_contiguous() else w.contiguous()
    if _PROFILING:
        set_tag("%d k%d s%d @%dx%dx%d" % (C, k, stride, B, H, W))
    _check(load().occd_dwconv2d_nchw(_f32(x, "x"), _f32(wc, "w"), _f32(scale, "scale") if scale is not None else None,
                                     _f32(shift, "shift") if shift is not None else None, _f32(y, "y"), B, C, H, W, k,
                                     stride, pad_h // 2, pad_w // 2, Ho, Wo, ACT2D[act], _stream()),
           "occd_dwconv2d_nchw")
    return y


def depthnet_gate(mlp, se, images, sps=None, intrins=None, factor=1000.0):
    """(images, C) gate of DepthNet's camera-aware SE layer in one launch (occd_depthnet_gate): `mlp` = Mlp(1, C, C), `se` =
    SELayer(C); the scaled pixel size comes from `sps` (images,) or is derived from `intrins` (images, 4, 4) pinhole
    matrices (float32, dense)."""
    C = mlp.fc2.out_features

    def f(t):
        t = t.detach().float()
        return t if t.is_contiguous() else t.contiguous()
    gate = torch.empty((images, C), device=mlp.fc1.weight.device, dtype=torch.float32)
    if (sps is None) == (intrins is None):
        raise RuntimeError("depthnet_gate: give exactly one of sps / intrins")
    if intrins is not None:
        intr = f(intrins).reshape(images, -1)
        if intr.shape[1] < 6:
            raise RuntimeError("depthnet_gate: intrinsics need at least 6 floats per image")
        sp, ip, stride = None, _f32(intr, "intrins"), intr.shape[1]
    else:
        sv = f(sps).reshape(-1)
        if sv.numel() != images:
            raise RuntimeError("depthnet_gate: one scaled pixel size per image")
        sp, ip, stride = _f32(sv, "sps"), None, 0
    ws = [f(mlp.fc1.weight).reshape(-1), f(mlp.fc1.bias), f(mlp.fc2.weight), f(mlp.fc2.bias),
          f(se.conv_reduce.weight).reshape(C, C), f(se.conv_reduce.bias), f(se.conv_expand.weight).reshape(C, C),
          f(se.conv_expand.bias)]
    if _PROFILING:
        set_tag("%d @%d" % (C, images))
    _check(load().occd_depthnet_gate(sp, ip, stride, float(factor), *[_f32(w, "w") for w in ws], gate.data_ptr(), images, C,
                                     _stream()), "occd_depthnet_gate")
    return gate


def stem_conv3x3(x, w, scale, shift, stride, act=None):
    """conv3x3 (TensorFlow SAME padding, stride 1 / 2) of a 3-channel image + per-channel affine + activation in one launch
    (occd_stem_conv3x3_nchw): the EfficientNet stem."""
    if x.dim() != 4 or x.shape[1] != 3 or tuple(w.shape[1:]) != (3, 3, 3):
        raise RuntimeError("stem_conv3x3: x must be (B, 3, H, W) and w (cout, 3, 3, 3)")
    x = x if x.is_contiguous() else x.contiguous()
    B, _, H, W = x.shape
    cout = w.shape[0]
    Ho, Wo = -(-H // stride), -(-W // stride)
    pad_h = max((Ho - 1) * stride + 3 - H, 0)
    pad_w = max((Wo - 1) * stride + 3 - W, 0)
    y = torch.empty((B, cout, Ho, Wo), device=x.device, dtype=torch.float32)
    wc = w.detach().float()
    wc = wc if wc.is_contiguous() else wc.contiguous()
    if _PROFILING:
        set_tag("3>%d s%d @%dx%dx%d" % (cout, int(stride), B, H, W))
    _check(load().occd_stem_conv3x3_nchw(_f32(x, "x"), _f32(wc, "w"), _f32(scale, "scale") if scale is not None else None,
                                         _f32(shift, "shift") if shift is not None else None, y.data_ptr(), B, H, W, cout,
                                         int(stride), pad_h // 2, pad_w // 2, Ho, Wo, ACT2D[act], _stream()),
           "occd_stem_conv3x3_nchw")
    return y


def _same_geometry(H, W, k, stride):
    Ho, Wo = -(-H // stride), -(-W // stride)
    pad_h = max((Ho - 1) * stride + k - H, 0)
    pad_w = max((Wo - 1) * stride + k - W, 0)
    return Ho, Wo, pad_h // 2, pad_w // 2


class _DwConvSameFn(torch.autograd.Function):
    """Depthwise convolution with TensorFlow SAME padding, forward / data gradient / weight gradient on the HIP kernels
    (csrc/nchw2d.hip); under autocast it computes in float32 (HBM-bound either way)."""

    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, x, w, stride):
        x = x.contiguous()
        y = dwconv2d_same(x, w.detach(), None, None, stride, None)
        ctx.save_for_backward(x, w)
        ctx.stride = stride
        return y

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, gy):
        x, w = ctx.saved_tensors
        stride = ctx.stride
        B, C, H, W = x.shape
        k = w.shape[-1]
        Ho, Wo, pt, pl = _same_geometry(H, W, k, stride)
        gy = gy.float().contiguous()
        wc = w.detach().float().contiguous()
        dx = dw = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            _check(load().occd_dwconv2d_bwd_data_nchw(_f32(gy, "gy"), _f32(wc, "w"), _f32(dx, "dx"), B, C, H, W, k, stride,
                                                      pt, pl, Ho, Wo, _stream()), "occd_dwconv2d_bwd_data_nchw")
        if ctx.needs_input_grad[1]:
            n = load().occd_dwconv2d_bwd_weight_workspace_floats(B, C, k, Ho, Wo)
            ws = torch.empty(n, device=x.device, dtype=torch.float32)
            dw = torch.empty_like(wc)
            _check(load().occd_dwconv2d_bwd_weight_nchw(_f32(x, "x"), _f32(gy, "gy"), _f32(dw, "dw"), _f32(ws, "ws"), B, C,
                                                        H, W, k, stride, pt, pl, Ho, Wo, _stream()),
                   "occd_dwconv2d_bwd_weight_nchw")
        return dx, dw, None


def dwconv2d_same_autograd(x, w, stride):
    """Differentiable depthwise SAME convolution (training path of the EfficientNet blocks)."""
    return _DwConvSameFn.apply(x, w, int(stride))


class _SwishFn(torch.autograd.Function):
    """x * sigmoid(x) with a one-pass forward (affine_act) and a one-pass backward (occd_swish_bwd): 2 launches per
    site instead of the 6 of the autograd graph of `x * torch.sigmoid(x)` (training path of the EfficientNet blocks)."""

    @staticmethod
    def forward(ctx, x):
        xc = x.detach()
        if not xc.is_contiguous():
            xc = xc.contiguous()
        ctx.save_for_backward(xc)
        B, C = (xc.shape[0], xc.shape[1]) if xc.dim() >= 2 else (1, 1)
        return affine_act(xc.reshape(1, 1, -1) if B * C > 65535 or xc.dim() < 2 else xc, None, None, "swish",
                          out=torch.empty_like(xc)).reshape(x.shape)

    @staticmethod
    def backward(ctx, gy):
        (x,) = ctx.saved_tensors
        g = gy.float().contiguous()
        gx = torch.empty_like(x)
        _check(load().occd_swish_bwd(_f32(x, "x"), _f32(g, "gy"), _f32(gx, "gx"), x.numel(), _stream()), "occd_swish_bwd")
        return gx


def swish_autograd(x):
    """Differentiable swish on a float32 CUDA tensor (see _SwishFn)."""
    return _SwishFn.apply(x)


class _UpCatFn(torch.autograd.Function):
    """cat([F.interpolate(x, skip's size, bilinear, align_corners=True), skip], 1) with the one-pass HIP forward
    (upsample_cat_kernel) -- ATen needs an upsample pass (0.33 ms per call at the high-resolution levels) and a concat
    copy -- and ATen's upsample backward on the matching gradient slice."""

    @staticmethod
    def forward(ctx, x, skip):
        ctx.x_shape = tuple(x.shape)
        ctx.size = tuple(skip.shape[2:])
        return upsample_bilinear_cat(x.detach().float(), skip.detach().float())

    @staticmethod
    def backward(ctx, g):
        C = ctx.x_shape[1]
        gx = gs = None
        if ctx.needs_input_grad[0]:
            gx = torch.ops.aten.upsample_bilinear2d_backward(g[:, :C].contiguous(), list(ctx.size), list(ctx.x_shape), True,
                                                             None, None)
        if ctx.needs_input_grad[1]:
            gs = g[:, C:]
        return gx, gs


class _UpCatClFn(torch.autograd.Function):
    """Channels-last twin of _UpCatFn for the bf16-mode decoder: x, skip and the result are logical (B, C, H, W) tensors in
    channels_last memory; forward = one launch (occd_upsample_bilinear_cat_nhwc), backward of the resized part = one gather
    launch (occd_upsample_bilinear_nhwc_bwd; ATen's nhwc backward scatters with atomics), the skip gradient is a view."""

    @staticmethod
    def forward(ctx, x, skip):
        B, C, h, w = x.shape
        Cs, H, W = skip.shape[1], skip.shape[2], skip.shape[3]
        xr = x.detach().float().permute(0, 2, 3, 1).contiguous()
        sr = skip.detach().float().permute(0, 2, 3, 1).contiguous()
        out = torch.empty((B, H, W, C + Cs), device=x.device, dtype=torch.float32)
        _check(load().occd_upsample_bilinear_cat_nhwc(_f32(xr, "x"), _f32(sr, "skip"), _f32(out, "out"), B, C, Cs, h, w, H, W,
                                                      _stream()), "occd_upsample_bilinear_cat_nhwc")
        ctx.geom = (B, C, Cs, h, w, H, W)
        return out.permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, g):
        B, C, Cs, h, w, H, W = ctx.geom
        gr = g.float().permute(0, 2, 3, 1)
        if not gr.is_contiguous():
            gr = gr.contiguous()
        gx = gs = None
        if ctx.needs_input_grad[0]:
            gxr = torch.empty((B, h, w, C), device=g.device, dtype=torch.float32)
            _check(load().occd_upsample_bilinear_nhwc_bwd(_f32(gr, "gout"), _f32(gxr, "gx"), B, C, C + Cs, h, w, H, W,
                                                          _stream()), "occd_upsample_bilinear_nhwc_bwd")
            gx = gxr.permute(0, 3, 1, 2)
        if ctx.needs_input_grad[1]:
            gs = gr[..., C:].permute(0, 3, 1, 2)
        return gx, gs


def upsample_bilinear_cat_cl_autograd(x, skip):
    return _UpCatClFn.apply(x, skip)


def upsample_bilinear_cat_autograd(x, skip):
    return _UpCatFn.apply(x, skip)


class _Conv3x3Fn(torch.autograd.Function):
    """Differentiable nn.Conv2d(3x3, stride 1, padding 1) on K10 (SURVEY 8(f) rows N1 + N3: the 2-D decoder in the training
    step).  forward = K10; data gradient = K10 on dL/dy with the 180-degree rotated, channel-transposed kernel (a 3x3 / pad 1
    convolution is its own transpose up to that); weight / bias gradient = the backend's wgrad (MIOpen implicit GEMM).
    MIOpen's fp32 forward / data-gradient choice for these shapes is the VALU Winograd kernel: 32 ms of a config-2 step."""

    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, x, w, b):
        y = conv2d_3x3_fused(x, wino_pack_weights(w.detach()), w.shape[0], b.detach() if b is not None else None)
        ctx.save_for_backward(x, w)
        ctx.has_bias = b is not None
        return y

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, gy):
        x, w = ctx.saved_tensors
        gy = gy.float().contiguous()
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            wt = w.detach().flip(2, 3).transpose(0, 1).contiguous()          # (cin, cout, 3, 3), rotated
            dx = conv2d_3x3_fused(gy, wino_pack_weights(wt), w.shape[1])
        want_b = ctx.has_bias and ctx.needs_input_grad[2]
        if ctx.needs_input_grad[1] or want_b:
            _, dw, db = torch.ops.aten.convolution_backward(gy, x, w.detach(), [w.shape[0]] if want_b else None, [1, 1], [1, 1],
                                                            [1, 1], False, [0, 0], 1, [False, True, want_b])
            if not ctx.needs_input_grad[1]:
                dw = None
        return dx, dw, db


def conv2d_3x3_autograd(x, w, b=None):
    """Differentiable 3x3 / stride 1 / padding 1 convolution, (B, Cin, H, W) float32 CUDA tensors."""
    return _Conv3x3Fn.apply(x, w, b)


def softmax_nchw(x):
    """softmax over dim 1 of a contiguous float32 (B, C, *spatial) GPU tensor (one HIP launch)."""
    if not x.is_contiguous():
        x = x.contiguous()
    B, C = x.shape[0], x.shape[1]
    y = torch.empty_like(x)
    if _PROFILING:
        set_tag("%d @%dx%d" % (C, B, x.numel() // (B * C)))
    _check(load().occd_softmax_nchw(_f32(x, "x"), _f32(y, "y"), B, C, x.numel() // (B * C), _stream()), "occd_softmax_nchw")
    return y


def dwconv2d_same_pool(x, w, scale, shift, stride, act=None):
    """dwconv2d_same that also returns the squeeze-excite pooling partials (B*C, nblk) and the plane size Ho*Wo.
    x: dense, or (B, C, H, W) planes on a padded pitch (the view of an expand GEMM's `padded_rows` result)."""
    B, C, H, W = x.shape
    xps = x.stride(1)
    if not (x.stride(3) == 1 and x.stride(2) == W and xps >= H * W and (B == 1 or x.stride(0) == C * xps)):
        x = x.contiguous()
        xps = H * W
    k = w.shape[-1]
    Ho, Wo = -(-H // stride), -(-W // stride)
    pad_h = max((Ho - 1) * stride + k - H, 0)
    pad_w = max((Wo - 1) * stride + k - W, 0)
    y = torch.empty((B, C, Ho, Wo), device=x.device, dtype=torch.float32)
    nblk = load().occd_dwconv2d_pool_blocks(Ho, Wo)
    part = torch.empty((B * C, nblk), device=x.device, dtype=torch.float32)
    wc = w if w.is_contiguous() else w.contiguous()
    if x.dtype != torch.float32 or not x.is_cuda:
        raise RuntimeError("dwconv2d_same_pool: x must be a float32 GPU tensor")
    if _PROFILING:
        set_tag("%d k%d s%d @%dx%dx%d" % (C, k, stride, B, H, W))
    _check(load().occd_dwconv2d_pool_nchw(x.data_ptr(), _f32(wc, "w"), _f32(scale, "scale") if scale is not None else None,
                                          _f32(shift, "shift") if shift is not None else None, _f32(y, "y"),
                                          _f32(part, "pool_part"), B, C, H, W, k, stride, pad_h // 2, pad_w // 2, Ho, Wo,
                                          ACT2D[act], xps, _stream()), "occd_dwconv2d_pool_nchw")
    return y, part, Ho * Wo


def se_gate(part, plane_size, batch, w_reduce, b_reduce, w_expand, b_expand):
    """Squeeze-excite gate (B, C) from the pooling partials of dwconv2d_same_pool (two small launches)."""
    C = part.shape[0] // batch
    Cr = w_reduce.shape[0]
    wr = w_reduce.detach().reshape(Cr, C).contiguous()
    we = w_expand.detach().reshape(C, Cr).contiguous()
    r = torch.empty((batch, Cr), device=part.device, dtype=torch.float32)
    gate = torch.empty((batch, C), device=part.device, dtype=torch.float32)
    if _PROFILING:
        set_tag("%d>%d>%d @%dx%d" % (C, Cr, C, batch, int(plane_size)))
    _check(load().occd_se_gate(_f32(part, "pool_part"), _f32(wr, "w_reduce"), _f32(b_reduce.detach().contiguous(), "b_reduce"),
                               _f32(we, "w_expand"), _f32(b_expand.detach().contiguous(), "b_expand"), _f32(r, "r"),
                               _f32(gate, "gate"), batch, C, Cr, part.shape[1], int(plane_size), _stream()), "occd_se_gate")
    return gate


def plane_reduce(a, b=None):
    """(planes,) sums over the trailing plane of `a` ((..., H, W) dense), or of a * b when b is given (occd_plane_reduce)."""
    a = a if a.is_contiguous() else a.contiguous()
    S = a.shape[-2] * a.shape[-1]
    planes = a.numel() // S
    if b is not None:
        b = b if b.is_contiguous() else b.contiguous()
        if b.shape != a.shape:
            raise RuntimeError("plane_reduce: operands must have the same shape")
    out = torch.empty(planes, device=a.device, dtype=torch.float32)
    _check(load().occd_plane_reduce(_f32(a, "a"), _f32(b, "b") if b is not None else None, out.data_ptr(), planes, S, _stream()),
           "occd_plane_reduce")
    return out


class _SqueezeExciteFn(torch.autograd.Function):
    """Training: geffnet's SqueezeExcite -- out = x * sigmoid(We swish(Wr mean_hw(x) + br) + be) -- forward and backward on the
    HIP passes of csrc/se2d.hip (4 + 4 launches instead of the ~22 of the autograd graph; deterministic)."""

    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, x, w_reduce, b_reduce, w_expand, b_expand):
        x = x if x.is_contiguous() else x.contiguous()
        B, C, H, W = x.shape
        Cr = w_reduce.shape[0]
        sums = plane_reduce(x)
        wr = w_reduce.detach().reshape(Cr, C).contiguous()
        we = w_expand.detach().reshape(C, Cr).contiguous()
        br, be = b_reduce.detach().contiguous(), b_expand.detach().contiguous()
        r = torch.empty((B, Cr), device=x.device, dtype=torch.float32)
        gate = torch.empty((B, C), device=x.device, dtype=torch.float32)
        _check(load().occd_se_gate(sums.data_ptr(), _f32(wr, "w_reduce"), _f32(br, "b_reduce"), _f32(we, "w_expand"),
                                   _f32(be, "b_expand"), r.data_ptr(), gate.data_ptr(), B, C, Cr, 1, H * W, _stream()),
               "occd_se_gate")
        out = affine_act(x.view(1, B * C, H, W), gate.view(-1), None, out=torch.empty_like(x).view(1, B * C, H, W))
        ctx.save_for_backward(x, sums, gate, r, wr, br, we)
        return out.view(B, C, H, W)

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, gout):
        x, sums, gate, r, wr, br, we = ctx.saved_tensors
        B, C, H, W = x.shape
        Cr = wr.shape[0]
        gout = gout.float()
        gout = gout if gout.is_contiguous() else gout.contiguous()
        gg = plane_reduce(gout, x)
        dev = x.device
        dz = torch.empty((B, Cr), device=dev, dtype=torch.float32)
        gm = torch.empty((B, C), device=dev, dtype=torch.float32)
        gwr = torch.empty((Cr, C), device=dev, dtype=torch.float32)
        gbr = torch.empty(Cr, device=dev, dtype=torch.float32)
        gwe = torch.empty((C, Cr), device=dev, dtype=torch.float32)
        gbe = torch.empty(C, device=dev, dtype=torch.float32)
        _check(load().occd_se_bwd(gg.data_ptr(), gate.data_ptr(), sums.data_ptr(), r.data_ptr(), wr.data_ptr(), br.data_ptr(),
                                  we.data_ptr(), dz.data_ptr(), gm.data_ptr(), gwr.data_ptr(), gbr.data_ptr(), gwe.data_ptr(),
                                  gbe.data_ptr(), B, C, Cr, H * W, _stream()), "occd_se_bwd")
        gm.mul_(1.0 / (H * W))                                     # d loss / d x through the mean: + gm / S on every pixel
        gx = affine_act(gout.view(1, B * C, H, W), gate.view(-1), gm.view(-1), out=torch.empty_like(x).view(1, B * C, H, W))
        return gx.view(B, C, H, W), gwr.view(Cr, C, 1, 1), gbr, gwe.view(C, Cr, 1, 1), gbe


SE_TRAIN_MAX_IMAGES = 16
# OCCDEPTH_TRAIN_SE=0 restores the autograd graph of the SE blocks in training (A/B)
SE_TRAIN = os.environ.get("OCCDEPTH_TRAIN_SE", "1") == "1"


def squeeze_excite_autograd_ok(se, x):
    return (SE_TRAIN and x.is_cuda and x.dim() == 4 and x.shape[0] <= SE_TRAIN_MAX_IMAGES
            and (x.dtype == torch.float32 or (torch.is_autocast_enabled() and x.dtype in (torch.bfloat16, torch.float16)))
            and se.conv_reduce.bias is not None and se.conv_expand.bias is not None
            and 2 * x.shape[0] * se.conv_reduce.out_channels * 4 + x.shape[0] * 1024 <= 64 * 1024)


def squeeze_excite_autograd(se, x):
    return _SqueezeExciteFn.apply(x, se.conv_reduce.weight, se.conv_reduce.bias, se.conv_expand.weight, se.conv_expand.bias)


def upsample_bilinear_cat(x, skip):
    """cat([bilinear(x, skip's size, align_corners=True), skip], dim=1) in one pass."""
    if not x.is_contiguous():
        x = x.contiguous()
    if not skip.is_contiguous():
        skip = skip.contiguous()
    B, C, h, w = x.shape
    Cs, H, W = skip.shape[1], skip.shape[2], skip.shape[3]
    out = torch.empty((B, C + Cs, H, W), device=x.device, dtype=torch.float32)
    _check(load().occd_upsample_bilinear_cat_nchw(_f32(x, "x"), _f32(skip, "skip"), _f32(out, "out"), B, C, Cs, h, w,
                                                  H, W, _stream()), "occd_upsample_bilinear_cat_nchw")
    return out


def upconv_gather(z, cout, size, batch_inner=False, skip=None, wskip=None, shift=None, slope=0.01):
    """sum over the 9 taps of shift_t(bilinear_up(z_t)): z (B, 9 * cout, h, w) -> (B, cout, H, W), size = (H, W).
    With z = conv1x1(x, W9) this is conv3x3(pad 1)(bilinear_up(x, align_corners=True)) (see occd_upconv_gather_nchw).
    batch_inner: z is (9 * cout, B, h, w) -- the result of ONE GEMM over the pixels of all images."""
    # dense, or planes (the rows of the tap GEMM's result) on a padded pitch (`padded_rows`): the strides travel
    if not (z.stride(3) == 1 and z.stride(2) == z.shape[3] and z.stride(0) > 0 and z.stride(1) > 0 and z.dtype == torch.float32):
        z = z.contiguous()
    if batch_inner:
        c9, B, h, w = z.shape
        zcs, zbs = z.stride(0), z.stride(1)
        ok = zbs == h * w and zcs >= B * h * w
    else:
        B, c9, h, w = z.shape
        zcs, zbs = z.stride(1), z.stride(0)
        ok = zcs >= h * w and zbs >= c9 * zcs
    if not ok or not z.is_cuda:
        raise RuntimeError("upconv_gather: z must be a GPU tensor, dense or with planes on a padded pitch")
    zp = z.data_ptr()
    if c9 != 9 * cout:
        raise RuntimeError("upconv_gather: z must have 9 * cout channels")
    H, W = int(size[0]), int(size[1])
    out = torch.empty((B, cout, H, W), device=z.device, dtype=torch.float32)
    if _PROFILING:
        set_tag("%d taps9 @%dx%dx%d>%dx%d" % (cout, B, h, w, H, W))
    if skip is not None:
        # fused tail of the level: + conv3x3 over the (<= 4) skip channels + shift, LeakyReLU (occd_upconv_gather_skip_nchw)
        sk = skip if skip.is_contiguous() else skip.contiguous()
        _check(load().occd_upconv_gather_skip_nchw(zp, _f32(sk, "skip"), _f32(wskip, "wskip"), _f32(shift, "shift"),
                                                   _f32(out, "out"), B, cout, sk.shape[1], h, w, H, W, zcs, zbs, float(slope),
                                                   _stream()), "occd_upconv_gather_skip_nchw")
        return out
    _check(load().occd_upconv_gather_nchw(zp, _f32(out, "out"), B, cout, h, w, H, W, zcs, zbs, _stream()),
           "occd_upconv_gather_nchw")
    return out


def project_voxels(cam_E, cam_k, vox_origin, voxel_size, grid_dims, img_w, img_h, device="cuda", with_z=False):
    """GPU `vox2pix` (pattern 0): -> projected_pix (N, 1, 2) int64, fov_mask (N, 1) bool [, pix_z (N,) float32]."""
    import numpy as np
    e = np.ascontiguousarray(np.asarray(cam_E, dtype=np.float64).reshape(16))
    k = np.ascontiguousarray(np.asarray(cam_k, dtype=np.float64).reshape(9))
    o = np.ascontiguousarray(np.asarray(vox_origin, dtype=np.float64).reshape(3))
    X, Y, Z = (int(v) for v in grid_dims)
    n = X * Y * Z
    pix = torch.empty((n, 1, 2), dtype=torch.int64, device=device)
    fov = torch.empty((n, 1), dtype=torch.bool, device=device)
    z = torch.empty((n,), dtype=torch.float32, device=device) if with_z else None
    _check(load().occd_project_voxels(e.ctypes.data, k.ctypes.data, o.ctypes.data, float(voxel_size), X, Y, Z,
                                      int(img_w), int(img_h), _ptr(pix, "pix"), _ptr(fov.view(torch.uint8), "fov"),
                                      _ptr(z, "pix_z") if z is not None else None, _stream()), "occd_project_voxels")
    return (pix, fov, z) if with_z else (pix, fov)


def argmax_labels(logits, lut=None):
    """(B, C, X, Y, Z) float32 GPU logits -> (B, X, Y, Z) int32 class volume (argmax over C, first maximum wins),
    optionally mapped through `lut` (e.g. SemanticKITTI learning_map_inv).  Channels-last views returned by the
    eval path (rows of a possibly wider buffer) are consumed in place."""
    if logits.dtype != torch.float32 or not logits.is_cuda:
        raise RuntimeError("argmax_labels needs float32 GPU logits")
    cl = logits.permute(0, 2, 3, 4, 1)
    B, X, Y, Z, C = cl.shape
    cs = cl.stride(3)
    rows_ok = cl.stride(4) == 1 and cs >= C and cl.stride(2) == cs * Z and cl.stride(1) == cs * Z * Y and \
        cl.stride(0) == cs * Z * Y * X
    if not rows_ok:
        cl = cl.contiguous()
        cs = C
    rows = B * X * Y * Z
    out = torch.empty(rows, dtype=torch.int16, device=logits.device)
    lut_t = None
    if lut is not None:
        lut_t = torch.as_tensor(lut, dtype=torch.int32).to(torch.int16).to(logits.device).contiguous()
    _check(load().occd_argmax_channels(cl.data_ptr(), rows, cs, 0, C, lut_t.data_ptr() if lut_t is not None else None,
                                       out.data_ptr(), _stream()), "occd_argmax_channels")
    return (out.to(torch.int32) & 0xFFFF).view(B, X, Y, Z)


# ----------------------------------------------------------------------------- training-step statistics (N1 / N4)
SSC_Q32 = 4294967296.0
SSC_Q24 = 16777216.0


_ssc_scale_cache = {}


def ssc_stats_scale(C, F, device):
    """Multipliers that turn the fixed-point int64 statistics of occd_ssc_loss_stats_fwd into real sums
    (built once per (C, F, device): no host-to-device copy on the loss path after the first step)."""
    key = (int(C), int(F), str(device))
    sc = _ssc_scale_cache.get(key)
    if sc is None:
        sc = torch.ones(3 * C + 3 + F * C, dtype=torch.float64)
        sc[:2 * C] = 1.0 / SSC_Q32
        sc[3 * C + 1:3 * C + 3] = 1.0 / SSC_Q24
        sc[3 * C + 3:] = 1.0 / SSC_Q32
        sc = _ssc_scale_cache[key] = sc.to(device)
    return sc


def _logit_layout(logits):
    """(s_b, s_c, s_v) element strides of a (B, C, ...) logits tensor the loss kernels read in place: (B, C, S) planes
    (contiguous) or the 3-D stack's channels-last voxel rows (a permuted view of (B, ..., cs) rows, cs >= C); None when
    the tensor is neither (the caller makes it contiguous)."""
    B, C = logits.shape[:2]
    S = logits[0, 0].numel()
    if logits.is_contiguous():
        return C * S, S, 1
    st, sh = logits.stride(), logits.shape
    if st[1] != 1 or logits.dim() < 3:
        return None
    cs = st[-1]
    expect = cs
    for d in range(logits.dim() - 1, 1, -1):            # spatial dims must merge into one row index
        if sh[d] != 1 and st[d] != expect:
            return None
        expect *= sh[d]
    if cs < C or cs % 4 or (B > 1 and st[0] % 4) or logits.data_ptr() % 16:
        return None
    return (st[0] if B > 1 else S * cs), 1, cs


def _loss_operands(logits, target, masks, weights):
    if logits.dtype != torch.float32 or not logits.is_cuda or logits.dim() < 3:
        raise RuntimeError("ssc loss statistics need float32 GPU logits of shape (B, C, ...)")
    B, C = logits.shape[:2]
    S = logits[0, 0].numel()
    if target.dtype != torch.uint8 or tuple(target.shape) != (B,) + tuple(logits.shape[2:]):
        raise RuntimeError("target must be uint8 of shape (B, ...)")
    F = 0
    if masks is not None:
        if masks.dtype not in (torch.bool, torch.uint8) or masks.shape[0] != B or masks[0, 0].numel() != S:
            raise RuntimeError("frustum masks must be bool/uint8 of shape (B, F, ...)")
        F = masks.shape[1]
    if weights is not None and (weights.dtype != torch.float32 or weights.numel() != C):
        raise RuntimeError("class weights must be float32 of length C")
    return B, C, S, F


def ssc_loss_stats(logits, target, masks=None, weights=None, map_occ=False):
    """-> int64 (3C + 3 + F*C) fixed-point sums (see include/occdepth_amd.h); one pass over the logits, read in place
    whether they are (B, C, S) planes or channels-last voxel rows (`_logit_layout`)."""
    B, C, S, F = _loss_operands(logits, target, masks, weights)
    lay = _logit_layout(logits)
    if lay is None:
        logits = logits.contiguous()
        lay = (C * S, S, 1)
    stats = torch.empty(3 * C + 3 + F * C, dtype=torch.int64, device=logits.device)
    _check(load().occd_ssc_loss_stats_fwd_strided(logits.data_ptr(), _ptr(target, "target"), _ptr(masks, "masks"),
                                                  _ptr(weights, "weights"), stats.data_ptr(), B, C, S, F, int(bool(map_occ)),
                                                  lay[0], lay[1], lay[2], _stream()), "occd_ssc_loss_stats_fwd_strided")
    return stats


def ssc_loss_grad(logits, target, masks, weights, gstats, map_occ=False):
    """d loss / d logits from d loss / d sums (float32, the layout of ssc_loss_stats).  The gradient has the logits'
    layout: channels-last logits get a channels-last gradient -- a (B, C, ...) view of zero-padded (B, ..., cs) rows, which
    the convolution backward (autograd3d._to_vox) consumes without a transpose."""
    B, C, S, F = _loss_operands(logits, target, masks, weights)
    if gstats.dtype != torch.float32 or gstats.numel() != 3 * C + 3 + F * C:
        raise RuntimeError("gstats must be float32 of length 3C + 3 + F*C")
    lay = _logit_layout(logits)
    if lay is None:
        logits = logits.contiguous()
        lay = (C * S, S, 1)
    if lay[1] == 1:
        cs = round_up(C, 8)
        rows = torch.empty((B,) + tuple(logits.shape[2:]) + (cs,), dtype=torch.float32, device=logits.device)
        grad = rows[..., :C].permute(0, logits.dim() - 1, *range(1, logits.dim() - 1))
        glay, gpad = (S * cs, 1, cs), cs
    else:
        grad = torch.empty_like(logits)
        glay, gpad = lay, 0
    _check(load().occd_ssc_loss_stats_bwd_strided(logits.data_ptr(), _ptr(target, "target"), _ptr(masks, "masks"),
                                                  _ptr(weights, "weights"), _ptr(gstats, "gstats"), grad.data_ptr(), B, C, S, F,
                                                  int(bool(map_occ)), lay[0], lay[1], lay[2], glay[0], glay[1], glay[2], gpad,
                                                  _stream()), "occd_ssc_loss_stats_bwd_strided")
    return grad


def ssc_confusion(hist, target, logits=None, labels=None):
    """hist (C, C) int64 += confusion counts [target, prediction] over labelled voxels; the prediction is `labels`
    (uint8) or the arg-max of `logits` (B, C, ...; planes or channels-last rows, read in place)."""
    C = hist.shape[0]
    if hist.dtype != torch.int64 or hist.shape != (C, C) or target.dtype != torch.uint8:
        raise RuntimeError("hist must be int64 (C, C) and target uint8")
    if (logits is None) == (labels is None):
        raise RuntimeError("give exactly one of logits / labels")
    B = target.shape[0]
    S = target[0].numel()
    lay = (C * S, S, 1)
    if logits is not None:
        if logits.dtype != torch.float32 or logits.shape[:2] != (B, C) or logits[0, 0].numel() != S:
            raise RuntimeError("logits must be float32 (B, C, ...) matching target")
        lay = _logit_layout(logits)
        if lay is None:
            logits = logits.contiguous()
            lay = (C * S, S, 1)
    if labels is not None and (labels.dtype != torch.uint8 or labels.shape != target.shape):
        raise RuntimeError("labels must be uint8 with the shape of target")
    _check(load().occd_ssc_confusion_strided(logits.data_ptr() if logits is not None else None, _ptr(labels, "labels"),
                                             _ptr(target, "target"), _ptr(hist, "hist"), B, C, S, lay[0], lay[1], lay[2],
                                             _stream()), "occd_ssc_confusion_strided")
    return hist


# ---- relation (context prior) loss, occdepth/loss/CRP_loss.py:4-24 ------------------------------------------------
REL_Q24 = 16777216.0


def _relation_operands(logits, labels):
    if logits.dtype != torch.float32 or not logits.is_cuda or logits.dim() != 4:
        raise RuntimeError("relation logits must be a float32 GPU tensor (B, R, M, N)")
    B, R, M, N = logits.shape
    if labels.dtype not in (torch.uint8, torch.float32, torch.bool) or tuple(labels.shape) != (B, R, N, M) or \
            not labels.is_contiguous() or not labels.is_cuda:
        raise RuntimeError("relation labels must be a contiguous (B, R, N, M) uint8 / bool / float32 GPU tensor")
    st = logits.stride()
    if st[2] != 1 and st[3] != 1:
        raise RuntimeError("relation logits need unit stride along M or N")
    lab = labels.view(torch.uint8) if labels.dtype == torch.bool else labels
    return B, R, M, N, st, lab, (1 if lab.dtype == torch.float32 else 0)


# OCCDEPTH_LOSS_KERNELS=0 restores the ATen formulations of the relation / depth losses and of the frustum-sample backward (A/B)
LOSS_KERNELS = os.environ.get("OCCDEPTH_LOSS_KERNELS", "1") == "1"


def relation_bce_usable(logits, labels):
    """The relation loss runs on `relation_bce_stats` / `relation_bce_grad`: GPU tensors, float logits (B, R, M, N) that are a
    dense permutation with unit stride along M or N, labels (B, R, N, M) uint8 / bool / float32."""
    if not (LOSS_KERNELS and torch.is_tensor(logits) and logits.is_cuda and logits.dim() == 4 and labels.is_cuda):
        return False
    if not (logits.dtype == torch.float32 or (logits.dtype.is_floating_point and torch.is_autocast_enabled())) or \
            labels.dtype not in (torch.uint8, torch.bool, torch.float32):
        return False
    B, R, M, N = logits.shape
    st = logits.stride()
    dense = 1 + sum((n - 1) * s for n, s in zip(logits.shape, st)) == logits.numel()
    return tuple(labels.shape) == (B, R, N, M) and (st[2] == 1 or st[3] == 1) and dense and B * R <= 65535 and R <= 64


def relation_bce_stats(logits, labels):
    """-> int64 (R, 3): #positives, sum_{y=1} softplus(-x) and sum_{y=0} softplus(x) in Q24 (occd_relation_bce_stats)."""
    B, R, M, N, st, lab, ldt = _relation_operands(logits, labels)
    stats = torch.empty((R, 3), dtype=torch.int64, device=logits.device)
    _check(load().occd_relation_bce_stats(logits.data_ptr(), lab.data_ptr(), ldt, stats.data_ptr(), B, R, M, N, st[0], st[1],
                                          st[2], st[3], _stream()), "occd_relation_bce_stats")
    return stats


def relation_bce_grad(logits, labels, coef):
    """d loss / d logits with the logits' strides; coef (R, 2) float32 = g * (pos_weight_r, 1) / (R B M N)."""
    B, R, M, N, st, lab, ldt = _relation_operands(logits, labels)
    if coef.dtype != torch.float32 or tuple(coef.shape) != (R, 2) or not coef.is_contiguous():
        raise RuntimeError("coef must be contiguous float32 (R, 2)")
    dense = sorted(range(4), key=lambda d: -st[d])
    need = 1 + sum((logits.shape[d] - 1) * st[d] for d in range(4))
    if need != logits.numel():
        raise RuntimeError("relation logits must be a dense (permuted) tensor")
    grad = torch.empty_strided(tuple(logits.shape), st, dtype=torch.float32, device=logits.device)
    del dense
    _check(load().occd_relation_bce_grad(logits.data_ptr(), lab.data_ptr(), ldt, coef.data_ptr(), grad.data_ptr(), B, R, M, N,
                                         st[0], st[1], st[2], st[3], _stream()), "occd_relation_bce_grad")
    return grad


# ---- depth-distribution loss, occdepth/loss/depth_loss.py:14-87 ---------------------------------------------------
def depth_bce_usable(preds, labels):
    """The depth loss runs on `depth_bce_stats` / `depth_bce_grad`: float32 GPU predictions and GPU labels."""
    return LOSS_KERNELS and preds.is_cuda and labels.is_cuda and preds.dtype == torch.float32


def _depth_operands(prob, gt, cell):
    if prob.dtype != torch.float32 or not prob.is_cuda or prob.dim() != 4 or not prob[0].is_contiguous():
        raise RuntimeError("depth probabilities must be float32 GPU (Bn, D, h, w) with dense images")
    if gt.dtype != torch.float32 or gt.dim() != 3 or gt.shape[0] != prob.shape[0] or not gt.is_contiguous() or not gt.is_cuda:
        raise RuntimeError("depth labels must be contiguous float32 GPU (Bn, H, W)")
    Bn, D, h, w = prob.shape
    return Bn, D, h, w, gt.shape[1], gt.shape[2], (prob.stride(0) if Bn > 1 else D * h * w)


def depth_bce_stats(prob, gt, cell, d_off, d_step):
    """-> int64 [sum of per-cell BCE over measured cells (Q24), #measured cells] (occd_depth_bce_stats)."""
    Bn, D, h, w, sh, sw, pb = _depth_operands(prob, gt, cell)
    stats = torch.empty(2, dtype=torch.int64, device=prob.device)
    _check(load().occd_depth_bce_stats(prob.data_ptr(), gt.data_ptr(), stats.data_ptr(), Bn, D, h, w, sh, sw, int(cell), pb,
                                       float(d_off), float(d_step), _stream()), "occd_depth_bce_stats")
    return stats


def depth_bce_grad(prob, gt, cell, d_off, d_step, gscale):
    """dense (Bn, D, h, w) d loss / d prob; gscale: one-element float32 device tensor = g / max(1, #measured)."""
    Bn, D, h, w, sh, sw, pb = _depth_operands(prob, gt, cell)
    if gscale.dtype != torch.float32 or gscale.numel() != 1 or not gscale.is_cuda:
        raise RuntimeError("gscale must be a one-element float32 GPU tensor")
    grad = torch.empty((Bn, D, h, w), dtype=torch.float32, device=prob.device)
    _check(load().occd_depth_bce_grad(prob.data_ptr(), gt.data_ptr(), gscale.data_ptr(), grad.data_ptr(), Bn, D, h, w, sh, sw,
                                      int(cell), pb, float(d_off), float(d_step), _stream()), "occd_depth_bce_grad")
    return grad


def cascade_tail(part, occ_off, wn, nbr):
    """ssc = part[..., :nbr] + conv3x3x3(softmax(part[..., occ_off:occ_off+2]), wn); returns a Vox with cs = ceil4(nbr)."""
    out = Vox.empty(part.batch, part.dims, nbr, part.buf.device, cs=round_up(nbr, 4))
    X, Y, Z = part.dims
    wc = wn.detach().float().contiguous()
    if _PROFILING:
        set_tag("%d @%dx%dx%d" % (nbr, X, Y, Z))
    _check(load().occd_cascade_tail_fwd(_f32(part.buf, "part"), _f32(wc, "wn"), _f32(out.buf, "out"), part.batch, X, Y,
                                        Z, part.cs, occ_off, out.cs, nbr, _stream()), "occd_cascade_tail_fwd")
    return out


class profile:
    """Context manager: HIP-event timing of every kernel launched inside (see occd_prof_*)."""

    def __init__(self):
        self.rows = {}

    def __enter__(self):
        global _PROFILING
        load().occd_prof_report(None, 0)  # drop stale records
        load().occd_prof_enable(1)
        _PROFILING = True
        return self

    def __exit__(self, *exc):
        global _PROFILING
        lib = load()
        lib.occd_prof_enable(0)
        _PROFILING = False
        set_tag(None)
        buf = (ProfRow * 512)()
        n = lib.occd_prof_report(buf, 512)
        if n < 0:
            raise RuntimeError("occd_prof_report failed")
        for i in range(min(n, 512)):
            r = buf[i]
            self.rows[r.tag.decode()] = dict(launches=r.launches, ms=r.ms, flops=r.flops, bytes=r.bytes)
        return False


def set_tag(tag):
    load().occd_prof_set_tag(tag.encode() if tag else None)
