"""The eval lift without its tables (occd_lift_proj_fwd): in-kernel voxel projection + in-kernel
frustum sample + SFA gather must reproduce the three-kernel path (occd_project_voxels -> int64 tables,
occd_flosp_sample_fwd -> depth vector, occd_lift_fwd) to float32 round-off (1e-6 of the output scale) -- the projection
is integer-exact by construction (explicitly rounded float64, tests/test_parity_gpu.py::test_project_voxels_bit_exact pins
it to the dataloader's numpy semantics), so every voxel gathers the same pixels, and the float32 arithmetic downstream is
the same source code (the compiler contracts a few multiply-adds differently in the two kernels: 3 ulp measured).  The table path itself is pinned to the real
reference by the SFA / flosp goldens of tests/test_parity_gpu.py."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"

KITTI_K = np.array([[707.0912, 0, 601.8873], [0, 707.0912, 183.1104], [0, 0, 1]], dtype=np.float64)
KITTI_TR = np.array([[0, -1, 0, 0], [0, 0, -1, -0.08], [1, 0, 0, -0.27], [0, 0, 0, 1]], dtype=np.float64)


def calibration(batch, views, jitter, img_w):
    g = np.random.default_rng(5)
    E, K = [], []
    for b in range(batch):
        eb, kb = [], []
        for v in range(views):
            e = KITTI_TR.copy()
            e[0, 3] = -0.54 * v
            e[:3, 3] += jitter * g.normal(size=3) * 0.05
            k = KITTI_K.copy()
            k[:2] *= img_w / 1220.0
            k[0, 0] *= 1 + jitter * 0.01 * b
            eb.append(e)
            kb.append(k)
        E.append(np.stack(eb))
        K.append(np.stack(kb))
    return np.stack(E), np.stack(K)


def build_case(hip, batch, views, C, scales, dims, img_hw, with_frustum, jitter=1.0, seed=0):
    from occdepth_amd.models.flosp_depth.flosp_depth import _grid_to_lidar
    torch.manual_seed(seed)
    H, W = img_hw
    E, K = calibration(batch, views, jitter, W)
    feats = []
    for s in scales:
        h, w = -(-H // s), -(-W // s)
        stacked = torch.randn(batch, views, h, w, C, device=DEV)
        feats.append([stacked[:, v] for v in range(views)])          # batch stride != h * w * C: exercised on purpose
    voxel = 51.2 / dims[0]
    pix, fov = [], []
    for b in range(batch):
        tabs = [hip.project_voxels(E[b, v], K[b, v], (0.0, -25.6, -2.0), voxel, dims, W, H) for v in range(views)]
        pix.append(torch.stack([p for p, _ in tabs]))
        fov.append(torch.stack([m for _, m in tabs]))
    pix, fov = torch.stack(pix), torch.stack(fov)
    cam = (torch.from_numpy(E).to(DEV).contiguous(), torch.from_numpy(K).to(DEV).contiguous())
    frustum = None
    if with_frustum:
        D, h, w = 104, -(-H // 8), -(-W // 8)
        depth = torch.softmax(torch.randn(batch, views, D, h, w, device=DEV), 2).contiguous()
        g2l = _grid_to_lidar([0, -25.6, -2, 51.2, 25.6, 4.4], dims).to(DEV)
        t = torch.from_numpy(E.astype(np.float32)).to(DEV)
        intr = torch.zeros(batch, views, 4, 4, device=DEV)
        intr[:, :, :3, :3] = torch.from_numpy(K.astype(np.float32)).to(DEV)
        intr[:, :, 3, 3] = 1
        ida = torch.eye(4, device=DEV).repeat(batch, views, 1, 1).contiguous()
        frustum = hip.Frustum(depth, (t @ g2l).contiguous(), intr[:, :, :3, :].contiguous(), ida, dims, (H, W), 2.0, 54.0,
                              True)
    return feats, pix, fov, cam, frustum, voxel


CASES = {
    # name: (batch, views, C, scales, grid, image, frustum)
    "config2": (1, 2, 64, (1, 2, 4, 8), (128, 128, 16), (370, 1220), True),
    "config2_no_depth": (1, 2, 64, (1, 2, 4, 8), (128, 128, 16), (370, 1220), False),
    "batch2_c32": (2, 2, 32, (1, 2), (64, 64, 8), (185, 610), True),
    "mono_c128": (1, 1, 128, (2, 4), (64, 64, 8), (370, 1220), True),
    "c24_pad": (2, 2, 24, (1, 4, 8), (32, 32, 4), (96, 320), False),
}


@pytest.mark.parametrize("name", list(CASES))
def test_lift_proj_matches_table_path(name):
    from occdepth_amd import hip
    from occdepth_amd.hip import Vox
    batch, views, C, scales, dims, img_hw, with_frustum = CASES[name]
    feats, pix, fov, cam, frustum, voxel = build_case(hip, batch, views, C, scales, dims, img_hw, with_frustum)
    strides = (dims[1] * dims[2], dims[2], 1)
    ref = Vox.empty(batch, dims, C, DEV)
    ds = frustum.sample() if frustum is not None else None
    hip.lift(feats, scales, pix, fov, dims, strides, ref, depth_scale=ds, scale_const=100.0)
    assert 0.2 < fov.float().mean().item() < 0.98
    for mode in (0, 1, 2):
        out = Vox.empty(batch, dims, C, DEV)
        out.buf.fill_(float("nan"))
        hip.lift_proj(feats, scales, cam[0], cam[1], (0.0, -25.6, -2.0), voxel, (img_hw[1], img_hw[0]), dims, strides, out,
                      frustum=frustum, scale_const=100.0, xcd_mode=mode)
        # same indices, same float32 formulas; the compiler contracts a few a * b + c differently in the two kernels
        # (measured 3 ulp): the bar is 1e-6 of the output scale, a moved pixel or a dropped view would show as O(1)
        err = float((out.buf - ref.buf).abs().max() / ref.buf.abs().max())
        assert err < 1e-6, (name, mode, err)
        assert bool(torch.isfinite(out.buf).all())


def test_lift_proj_rejects_what_it_cannot_do():
    from occdepth_amd import hip
    from occdepth_amd.hip import Vox
    feats, pix, fov, cam, frustum, voxel = build_case(hip, 1, 2, 32, (1, 2), (64, 64, 8), (185, 610), False)
    out = Vox.empty(1, (64, 64, 8), 32, DEV)
    with pytest.raises(RuntimeError):                       # non-power-of-two scale
        hip.lift_proj(feats, (1, 3), cam[0], cam[1], (0.0, -25.6, -2.0), voxel, (610, 185), (64, 64, 8), (512, 8, 1), out)
    with pytest.raises(RuntimeError):                       # float32 calibration
        hip.lift_proj(feats, (1, 2), cam[0].float(), cam[1], (0.0, -25.6, -2.0), voxel, (610, 185), (64, 64, 8),
                      (512, 8, 1), out)
    with pytest.raises(RuntimeError):                       # non-power-of-two (Y, Z) of the grid (X is free)
        hip.lift_proj(feats, (1, 2), cam[0], cam[1], (0.0, -25.6, -2.0), voxel, (610, 185), (64, 60, 8), (480, 8, 1),
                      Vox.empty(1, (64, 60, 8), 32, DEV))


def test_device_vox_origin_is_honoured_not_ignored():
    """Lightning's transfer_batch_to_device moves every batch tensor to the GPU, so a batch-supplied `vox_origin`
    arrives as a CUDA tensor: it is read back ONCE per model (cached host copy, a warning when it differs from the
    SemanticKITTI default), never silently replaced by the default; first seen during a graph capture it raises."""
    import warnings
    from test_oracle_vs_golden import build_product
    m, cfg, sd = build_product("kitti_small")
    default = (0.0, -0.1 * float(cfg.full_scene_size[1]), -2.0)
    assert m._kitti_origin({}) == default and m._kitti_origin(None) == default
    assert m._kitti_origin({"vox_origin": torch.tensor([[0.0, -3.2, -2.0]])}) == pytest.approx((0.0, -3.2, -2.0))
    dev_origin = torch.tensor([[0.5, -3.0, -1.5]], device="cuda")
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        got = m._kitti_origin({"vox_origin": dev_origin})
    assert got == pytest.approx((0.5, -3.0, -1.5)) and any("vox_origin" in str(x.message) for x in w)
    assert m._kitti_origin({"vox_origin": dev_origin}) == got          # cached: no second read-back
    m2, _, _ = build_product("kitti_small")
    real = torch.cuda.is_current_stream_capturing
    torch.cuda.is_current_stream_capturing = lambda: True        # (what the whole-forward capture sees; no real capture needed)
    try:
        with pytest.raises(RuntimeError, match="graph capture"):
            m2._kitti_origin({"vox_origin": dev_origin})
    finally:
        torch.cuda.is_current_stream_capturing = real
