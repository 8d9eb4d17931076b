"""CPU: the host-side orchestration of the eval path, executed with the C-ABI entry points replaced
by their plain-torch emulation (tests/emu.py), must reproduce the REAL reference outputs held in
tests/golden.  This checks BatchNorm folding, pooled-branch folding, transposed-conv phase slicing,
concat-slice bookkeeping, residual wiring and the voxel index digits without a GPU; the kernels
themselves are checked by the `-m gpu` tests.
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import emu
import golden_cases as gc
from test_oracle_vs_golden import (_block_module, _unet3d_module, build_product, close, flat, gold, sd_for)


def test_product_refuses_cpu_tensors_without_emulation():
    m = _block_module("bottleneck_d2").eval()
    with pytest.raises(RuntimeError), torch.no_grad():       # the forward-only HIP eval path: no CPU implementation
        m(torch.zeros(1, 32, 4, 4, 4))
    from occdepth_amd.loss import ssc_loss
    with pytest.raises(RuntimeError):                          # the loss statistics kernels likewise
        ssc_loss.CE_ssc_loss(torch.zeros(1, 3, 2, 2, 2), torch.zeros(1, 2, 2, 2, dtype=torch.uint8), torch.ones(3))


@pytest.mark.parametrize("name", list(gc.BLOCK_CASES))
def test_blocks_eval_path(name):
    m = _block_module(name)
    m.load_state_dict(sd_for(m, "blocks3d", name))
    m.eval()
    g = gold("blocks3d")
    with emu.patched(), torch.no_grad():
        out = m(gc.randn(gc.BLOCK_CASES[name], name))
    for k, v in flat(out).items():
        close(v, g[name + ("." + k if k else "")], tol=2e-5, what=f"{name}.{k}")


@pytest.mark.parametrize("name", list(gc.UNET3D_CASES))
def test_unet3d_eval_path(name):
    spec = gc.UNET3D_CASES[name]
    m = _unet3d_module(spec)
    m.load_state_dict(sd_for(m, "unet3d", name))
    m.eval()
    g = gold("unet3d")
    with emu.patched(), torch.no_grad():
        out = m({"x3d": gc.randn(spec["x"], name)})
    assert len(out) == len([k for k in g.files if k.startswith(name + ".")])
    for k, v in out.items():
        close(gc.maybe_subsample(v.contiguous()), g[f"{name}.{k}"], tol=3e-4, what=f"{name}.{k}")


@pytest.mark.parametrize("cfg_name", ["kitti_small", "nyu_small", "kitti_flosp_small"])
def test_occdepth_eval_path(cfg_name):
    m, cfg, sd = build_product(cfg_name)
    g = gold("occdepth_small")
    with emu.patched(), torch.no_grad():
        out = m(gc.occdepth_batch(cfg_name))
    assert len([v for v in out.values() if v is not None]) == len([k for k in g.files if k.startswith(cfg_name + ".")])
    for k, v in out.items():
        close(gc.maybe_subsample(v.contiguous()), g[f"{cfg_name}.{k}"], tol=3e-4, what=f"{cfg_name}.{k}")


def test_occdepth_eval_path_in_kernel_lift():
    """VERDICT r2 item 7 host logic: with the dataloader's float64 extrinsics in the batch the eval forward hands the
    calibration to the fused lift (hip.lift_proj; here its emulation = numpy vox2pix + frustum sample + SFA) instead of the
    tables -- same golden, same bar; without them (or with the switch off) it reads the tables."""
    from occdepth_amd import hip
    from oracle import inputs
    m, cfg, sd = build_product("kitti_small")
    g = gold("occdepth_small")
    batch = gc.occdepth_batch("kitti_small")
    tr2 = inputs.KITTI_TR.copy()
    tr2[0, 3] = -0.54
    b64 = dict(batch, T_velo_2_cam_f64=[torch.from_numpy(np.stack([inputs.KITTI_TR, tr2])) for _ in batch["cam_k"]])
    no_tables = {k: v for k, v in b64.items() if not (k.startswith("projected_pix") or k.startswith("fov_mask"))}
    calls = []
    with emu.patched(), torch.no_grad():
        real = hip.lift_proj
        hip.lift_proj = lambda *a, **k: (calls.append(("proj", tuple(a[4]))), real(*a, **k))[1]
        real_t = hip.lift
        hip.lift = lambda *a, **k: (calls.append(("table",)), real_t(*a, **k))[1]
        assert m.lift_in_kernel == "auto"
        out = m(no_tables)                                         # default: no tables in the batch -> projects in the kernel
        assert [c[0] for c in calls] == ["proj"]
        assert calls[0][1] == (0.0, -0.1 * cfg.full_scene_size[1], -2.0)      # kitti_dataset.py:82, scaled to the scene width
        for k, v in out.items():
            close(gc.maybe_subsample(v.contiguous()), g[f"kitti_small.{k}"], tol=3e-4, what=f"kitti_small.{k}")
        del calls[:]
        m(b64)                                                     # default + tables: the batch's own tables win
        m(batch)
        assert [c[0] for c in calls] == ["table", "table"]
        del calls[:]
        m.lift_in_kernel = True                                    # the caller vouches for the tables: float64 extrinsics needed
        out = m(b64)
        m(batch)                                                   # (tables, float32 extrinsics only: table path)
        assert [c[0] for c in calls] == ["proj", "table"]
        for k, v in out.items():
            close(gc.maybe_subsample(v.contiguous()), g[f"kitti_small.{k}"], tol=3e-4, what=f"kitti_small.{k}")
        del calls[:]
        m.lift_in_kernel = False
        m(b64)
        assert [c[0] for c in calls] == ["table"]
        # no tables and no float64 extrinsics: projects from the float32 ones; a batch-supplied vox_origin is honoured
        m.lift_in_kernel = "auto"
        del calls[:]
        m({k: v for k, v in no_tables.items() if k != "T_velo_2_cam_f64"})
        m(dict(no_tables, vox_origin=torch.tensor([[0.0, -3.2, -2.0]])))
        assert [c[0] for c in calls] == ["proj", "proj"] and calls[1][1] == pytest.approx((0.0, -3.2, -2.0))


@pytest.mark.parametrize("cfg_name", ["kitti_small", "nyu_small"])
def test_occdepth_eval_path_fast2d(cfg_name):
    """The same end-to-end run, with the 2-D eval FAST paths taken too (fused.on_gpu forced on, every HIP entry point
    emulated): 4-launch MBConv blocks with folded BatchNorm / SE gates, the tap-GEMM + K12 form of the decoder levels, the
    merged conv_head, pixel-major decoder heads, DepthNet on the fused passes -- the host algebra the GPU runs, on the CPU,
    against the real reference's golden."""
    from test_oracle_vs_golden import oracle_float64, rel_err
    m, cfg, sd = build_product(cfg_name)
    g = gold("occdepth_small")
    with emu.patched(fast2d=True), torch.no_grad():
        out = m(gc.occdepth_batch(cfg_name))
    # The emulation evaluates every fused op in float64, so this run sits closer to the float64 value of the network than
    # the reference's own float32 run does on these ill-conditioned random-init configs (test_reference_float32_roundoff_on_
    # small_configs): the bars are the GPU test's -- 1e-3 against float64, 1e-3 + (golden vs float64) against the golden.
    truth = oracle_float64(cfg_name)
    for k, v in out.items():
        ref = torch.from_numpy(g[f"{cfg_name}.{k}"])
        e64 = rel_err(v, truth[k])
        eg = rel_err(gc.maybe_subsample(v.contiguous()), ref)
        g64 = rel_err(ref, gc.maybe_subsample(truth[k]))
        assert e64 < 1e-3 and eg < 1e-3 + g64, (k, e64, eg, g64)


@pytest.mark.parametrize("cfg_name", ["kitti_small", "nyu_small"])
def test_training_graph_matches_eval_math(cfg_name):
    """The ATen (autograd) graph used in training mode computes the same function (BN in eval mode)."""
    import torch.nn as nn
    m, cfg, sd = build_product(cfg_name)
    g = gold("occdepth_small")
    for sub in m.modules():
        sub.training = not isinstance(sub, (nn.BatchNorm2d, nn.BatchNorm3d, nn.Dropout))
    with torch.no_grad():
        out = m(gc.occdepth_batch(cfg_name))
    for k in ("ssc_logit",):
        close(gc.maybe_subsample(out[k].contiguous()), g[f"{cfg_name}.{k}"], tol=3e-4, what=k)


def test_voxel_layout_digits():
    from occdepth_amd.models.SFA import voxel_layout
    n_dims, out_dims, strides = voxel_layout((60, 36, 60), 1, "NYU")
    assert n_dims == (60, 60, 36) and out_dims == (60, 36, 60) and strides == (36 * 60, 1, 60)
    n_dims, out_dims, strides = voxel_layout((256, 256, 32), 2, "kitti")
    assert n_dims == out_dims == (128, 128, 16) and strides == (128 * 16, 16, 1)
    with pytest.raises(NotImplementedError):
        voxel_layout((8, 8, 8), 1, "tartanair")


def test_fill_is_deterministic_and_name_keyed():
    from oracle.fill import fill_state_dict
    a = fill_state_dict({"x.weight": torch.zeros(4, 3, 1, 1, 1), "x.bias": torch.zeros(4)}, 3)
    b = fill_state_dict({"x.bias": torch.zeros(4), "x.weight": torch.zeros(4, 3, 1, 1, 1)}, 3)
    assert all(torch.equal(a[k], b[k]) for k in a)
    assert abs(float(a["x.bias"][0]) - 0.0) > 0


def test_synthetic_inputs_match_survey_statistics():
    from oracle import inputs
    b = inputs.kitti_batch()
    fov = b["fov_mask_2"][0].float().mean(dim=(1, 2))
    assert b["projected_pix_2"][0].shape == (2, 262144, 1, 2) and b["projected_pix_2"][0].dtype == torch.int64
    assert torch.allclose(fov, torch.tensor([0.68, 0.68]), atol=0.01)
    assert b["cam_k"][0].dtype == torch.float64 and b["T_velo_2_cam"][0].dtype == torch.float32


def test_padded_rows_and_the_panel_launch_rule(monkeypatch):
    """hip.padded_rows: results whose rows start on 128-byte boundaries (a view of a buffer with the last dimension rounded up
    to 32 floats; dense under OCCDEPTH_PAD_ROWS=0), and the host-side mirror of occd_gemm_f32x3's rule for the panel-stationary
    kernel: K <= 352 always, K <= 848 only on the few-pixel launches."""
    import torch
    from occdepth_amd import hip
    monkeypatch.setattr(hip, "PAD_ROWS", True)
    z = hip.padded_rows((2, 18, 35), "cpu")
    assert z.shape == (2, 18, 35) and z.stride() == (18 * 64, 64, 1) and not z.is_contiguous()
    assert z.view(2, 18, 5, 7).stride() == (18 * 64, 64, 7, 1)            # planes of (h, w) on the padded pitch: a view
    full = hip.padded_rows((3, 64), "cpu")
    assert full.is_contiguous() and full.stride() == (64, 1)               # already a multiple of 32: nothing to pad
    monkeypatch.setattr(hip, "PAD_ROWS", False)
    assert hip.padded_rows((2, 18, 35), "cpu").is_contiguous()

    class Img:                                                              # what _panel_launch reads of a GemmPacked
        def __init__(self, rows, K):
            self.rows, self.K = rows, K
    b = lambda batch, K, N: torch.empty(batch, K, N, device="meta")
    assert hip._panel_launch(Img(720, 160), b(2, 160, 112850))             # tap GEMM 1/1
    assert hip._panel_launch(Img(1440, 320), b(2, 320, 28365))             # tap GEMM 1/2
    assert hip._panel_launch(Img(2304, 384), b(2, 384, 468))               # expand convolution of the 1/32 stage
    assert hip._panel_launch(Img(3840, 640), b(2, 640, 468))
    assert not hip._panel_launch(Img(2880, 640), b(2, 640, 7191))          # tap GEMM 1/4: K16 is faster ...
    assert hip._presplit_launch(Img(2880, 640), b(2, 640, 7191))           # ... on its pre-split form (chip-filling, K >= 512)
    assert hip._presplit_launch(Img(11520, 2560), b(2, 2560, 574)) and not hip._panel_launch(Img(11520, 2560), b(2, 2560, 574))
    assert not hip._presplit_launch(Img(2304, 384), b(2, 384, 468)) and not hip._presplit_launch(Img(2560, 640), b(2, 640, 574))
    w = torch.empty(2880, 640)
    assert hip.matmul_operand(w, "a")[1] is None                            # (CPU tensors are never pre-split)
