"""CPU: libocc_hip.so builds, loads, and exports exactly what include/occdepth_amd.h declares; the
ctypes structures in occdepth_amd/hip.py have the C layout (checked against gcc's sizeof/offsetof)."""
import ctypes
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "occdepth_amd.h")


def declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(occd_[a-z0-9_]+)\s*\(", src)))


def test_every_declared_symbol_is_exported_and_bound(hip_lib):
    from occdepth_amd import hip
    names = declared_functions()
    assert len(names) >= 12
    for n in names:
        assert hasattr(hip_lib, n), f"{n} declared in the header but not exported"
        assert n in hip.EXPORTS, f"{n} has no ctypes prototype"
    assert sorted(hip.EXPORTS) == names
    assert hip_lib.occd_abi_version() == hip.ABI_VERSION
    assert hip_lib.occd_strerror(-1).decode().startswith("invalid")


def test_argument_validation_without_gpu(hip_lib):
    """Pure host-side argument checks (no launch happens for invalid arguments)."""
    from occdepth_amd import hip
    assert hip_lib.occd_conv3d_fwd(None, None) == -1
    a = hip.Conv3dArgs()
    assert hip_lib.occd_conv3d_fwd(ctypes.byref(a), None) == -1
    assert hip_lib.occd_lift_fwd(None, None) == -1
    assert hip_lib.occd_flosp_sample_fwd(None, None) == -1
    assert hip_lib.occd_packed_weight_floats(32, 32, 27) == 27 * 4 * 1 * 256
    assert hip_lib.occd_packed_weight_floats(20, 34, 27) == 27 * 5 * 1 * 256
    assert hip_lib.occd_packed_weight_floats(0, 1, 1) < 0


def test_ctypes_structs_match_c_layout(tmp_path):
    from occdepth_amd import hip
    structs = {"occd_conv3d_args": hip.Conv3dArgs, "occd_flosp_args": hip.FlospArgs, "occd_lift_args": hip.LiftArgs,
               "occd_prof_row": hip.ProfRow, "occd_conv3d_wgrad_args": hip.WgradArgs, "occd_wino_args": hip.WinoArgs,
               "occd_pw_args": hip.PwArgs, "occd_lift_bwd_args": hip.LiftBwdArgs, "occd_bn_args": hip.BnArgs,
               "occd_lift_proj_args": hip.LiftProjArgs, "occd_bneck_args": hip.BneckArgs,
               "occd_rows_gemm_args": hip.RowsGemmArgs, "occd_gemm_args": hip.GemmArgs}
    rename = {"inp": "in"}
    lines = ['#include <stdio.h>', '#include <stddef.h>', f'#include "{HEADER}"', "int main(void){"]
    for cname, st in structs.items():
        lines.append(f'printf("{cname} %zu\\n", sizeof({cname}));')
        for fname, _ in st._fields_:
            lines.append(f'printf("{cname}.{fname} %zu\\n", offsetof({cname}, {rename.get(fname, fname)}));')
    lines += ["return 0;}"]
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-std=c11", str(src), "-o", str(exe)])
    out = dict(l.split() for l in subprocess.check_output([str(exe)]).decode().splitlines())
    for cname, st in structs.items():
        assert int(out[cname]) == ctypes.sizeof(st), cname
        for fname, _ in st._fields_:
            assert int(out[f"{cname}.{fname}"]) == getattr(st, fname).offset, f"{cname}.{fname}"


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from occdepth_amd import hip
    monkeypatch.setattr(hip, "_lib", None)
    monkeypatch.setattr(hip, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(RuntimeError, match="missing"):
        hip.load()


def test_new_entry_points_validate_arguments(hip_lib):
    """Host-side validation of the 2-D helper / projection / tail entry points (no launch for bad arguments)."""
    import ctypes
    assert hip_lib.occd_affine_act_nchw(None, None, None, None, None, 1, 1, 1, 0, 0.0, 0, None) == -1
    assert hip_lib.occd_dwconv2d_nchw(None, None, None, None, None, 1, 1, 1, 1, 3, 1, 0, 0, 1, 1, 0, None) == -1
    assert hip_lib.occd_upsample_bilinear_cat_nchw(None, None, None, 1, 1, 0, 1, 1, 1, 1, None) == -1
    assert hip_lib.occd_cascade_tail_fwd(None, None, None, 1, 1, 1, 1, 24, 20, 20, 20, None) == -1
    assert hip_lib.occd_argmax_channels(None, 1, 4, 0, 4, None, None, None) == -1
    e = (ctypes.c_double * 16)()
    assert hip_lib.occd_project_voxels(ctypes.addressof(e), None, None, 0.2, 1, 1, 1, 1, 1, None, None, None, None) == -1


def test_training_entry_points_validate_arguments(hip_lib):
    """Loss statistics / confusion / weight-gradient entry points reject bad arguments before any launch."""
    import ctypes
    from occdepth_amd import hip
    assert hip_lib.occd_ssc_stats_len(20, 64) == 3 * 20 + 3 + 64 * 20
    assert hip_lib.occd_ssc_loss_stats_fwd(None, None, None, None, None, 1, 20, 8, 0, 0, None) == -1
    one = ctypes.c_float(0.0)
    ptr = ctypes.addressof(one)
    assert hip_lib.occd_ssc_loss_stats_fwd(ptr, ptr, None, None, ptr, 1, 33, 8, 0, 0, None) == -1      # C > 32
    assert hip_lib.occd_ssc_loss_stats_fwd(ptr, ptr, None, None, ptr, 1, 20, 8, 4, 0, None) == -1       # masks missing
    assert hip_lib.occd_ssc_loss_stats_bwd(ptr, ptr, None, None, None, ptr, 1, 20, 8, 0, 0, None) == -1
    assert hip_lib.occd_ssc_confusion(ptr, ptr, ptr, ptr, 1, 20, 8, None) == -1                         # both predictions
    assert hip_lib.occd_ssc_confusion(None, None, ptr, ptr, 1, 20, 8, None) == -1                       # neither
    a = hip.WgradArgs()
    assert hip_lib.occd_conv3d_wgrad_workspace_floats(ctypes.byref(a)) == -1
    a.x = a.gy = ptr
    a.batch, a.X, a.Y, a.Z, a.cin, a.x_cs = 1, 4, 4, 8, 32, 32
    a.Xo, a.Yo, a.Zo, a.cout, a.gy_cs = 4, 4, 8, 32, 32
    a.kx = a.ky = a.kz = 3
    a.sx = a.sy = a.sz = a.dx = a.dy = a.dz = 1
    a.px = a.py = a.pz = 1
    need = hip_lib.occd_conv3d_wgrad_workspace_floats(ctypes.byref(a))
    assert need > 0 and need % (27 * 1024) == 0
    assert hip_lib.occd_conv3d_wgrad(ctypes.byref(a), None) == -1                                        # no dw / workspace
    a.kx = 6
    a.ky = 6
    assert hip_lib.occd_conv3d_wgrad_workspace_floats(ctypes.byref(a)) == -1                             # > 32 taps


def test_winograd_entry_points_validate_arguments(hip_lib):
    assert hip_lib.occd_wino_input_transform_nchw(None, None, 1, 8, 4, 4, 0, 2, None) == -1
    assert hip_lib.occd_wino_output_transform_nchw(None, None, None, None, None, 1, 8, 4, 4, 0, 2, 0, 0.0, 0, None) == -1
    import ctypes
    one = ctypes.c_float(0.0)
    ptr = ctypes.addressof(one)
    assert hip_lib.occd_wino_input_transform_nchw(ptr, ptr, 0, 8, 4, 4, 0, 2, None) == -1
    assert hip_lib.occd_wino_input_transform_nchw(ptr, ptr, 1, 8, 4, 4, 1, 2, None) == -1                            # strip past the image
    assert hip_lib.occd_wino_output_transform_nchw(ptr, None, None, None, ptr, 1, 8, 4, 4, 0, 2, 7, 0.0, 0, None) == -1     # act code


def test_round2_2d_entry_points_validate_arguments(hip_lib):
    """K10 / K11 / SE-gate / pooled depthwise entry points: argument checks happen before any launch."""
    import ctypes
    from occdepth_amd import hip
    assert hip_lib.occd_wino_packed_floats(80, 163) == 21 * 16 * 3 * 256
    assert hip_lib.occd_wino_packed_floats(0, 8) < 0
    assert hip_lib.occd_wino_conv3x3_fwd(None, None) == -1
    w = hip.WinoArgs()
    assert hip_lib.occd_wino_conv3x3_fwd(ctypes.byref(w), None) == -1
    one = ctypes.c_float(0.0)
    ptr = ctypes.addressof(one)
    w.x = w.upk = w.y = ptr
    w.batch, w.cin, w.cout, w.H, w.W, w.act = 1, 8, 8, 4, 4, 9
    assert hip_lib.occd_wino_conv3x3_fwd(ctypes.byref(w), None) == -1                  # act code
    assert hip_lib.occd_wino_pack_weights(None, None, None, 8, 8, None) == -1
    assert hip_lib.occd_pw_packed_floats(48, 288) == 36 * 2 * 256
    assert hip_lib.occd_pw_pack_weights(None, None, None, 8, 8, None) == -1
    a = hip.PwArgs()
    assert hip_lib.occd_pw_conv_fwd(ctypes.byref(a), None) == -1
    a.x = a.wpk = a.y = ptr
    a.batch, a.cin, a.cout, a.N = 1, 8, 8, 16
    a.tile_hint = 13
    assert hip_lib.occd_pw_conv_fwd(ctypes.byref(a), None) == -1                       # unknown variant (1..6 K11, 7..12 K11s)
    a.tile_hint, a.out_nhwc_cs = 0, 4
    assert hip_lib.occd_pw_conv_fwd(ctypes.byref(a), None) == -1                       # NHWC row shorter than Cout
    assert hip_lib.occd_dwconv2d_pool_blocks(185, 610) == (185 * 153 + 255) // 256
    assert hip_lib.occd_dwconv2d_pool_nchw(ptr, ptr, None, None, ptr, None, 1, 1, 4, 4, 3, 1, 1, 1, 4, 4, 0, 0, None) == -1
    assert hip_lib.occd_dwconv2d_pool_nchw(ptr, ptr, None, None, ptr, ptr, 1, 1, 4, 4, 3, 1, 1, 1, 4, 4, 0, 15, None) == -1   # plane stride < H * W
    assert hip_lib.occd_se_gate(None, None, None, None, None, None, None, 1, 8, 2, 1, 16, None) == -1
    assert hip_lib.occd_lift_bwd(None, None) == -1
    q = hip.LiftBwdArgs()
    assert hip_lib.occd_lift_bwd(ctypes.byref(q), None) == -1
    assert hip_lib.occd_softmax_nchw(None, None, 1, 4, 8, None) == -1


def test_gemm_entry_point_validates_kernel_hints(hip_lib):
    """occd_gemm_f32x3: host-side checks around the kernel hints (no launch for invalid arguments) -- hint 8 (K16p, the
    panel-stationary kernel) needs the pre-split A image, a plain epilogue and a K whose 32-column panel fits LDS."""
    from occdepth_amd import hip
    buf = (ctypes.c_float * 64)()
    ptr = (ctypes.addressof(buf) + 63) & ~63            # 64-byte aligned dummy address (never dereferenced: every call is rejected)
    q = hip.GemmArgs()
    q.A = q.B = q.C = ptr
    q.M, q.N, q.K, q.batch = 256, 64, 64, 1
    q.lda, q.ldb, q.ldc = 64, 64, 64
    for hint in (-1, 9):
        q.tile_hint = hint
        assert hip_lib.occd_gemm_f32x3(ctypes.byref(q), None) == -1
    q.tile_hint = 8                                      # float32 A: not the panel kernel's operand
    q.pre = 0
    assert hip_lib.occd_gemm_f32x3(ctypes.byref(q), None) == -1
    q.pre, q.K, q.lda = 1, 856, 856                      # 864 k x 192 B > 160 KB of LDS
    assert hip_lib.occd_gemm_f32x3(ctypes.byref(q), None) == -1
    q.K, q.lda, q.res = 64, 64, ptr                      # a residual operand: K16's epilogue, not K16p's
    assert hip_lib.occd_gemm_f32x3(ctypes.byref(q), None) == -1
    q.res, q.act_a = None, 1                             # sigmoid on A needs the float32 operand
    assert hip_lib.occd_gemm_f32x3(ctypes.byref(q), None) == -1
    q.act_a, q.K = 0, 60                                 # K % 8
    assert hip_lib.occd_gemm_f32x3(ctypes.byref(q), None) == -1
