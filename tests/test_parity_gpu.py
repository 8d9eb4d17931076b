"""GPU parity proper: the HIP eval path (through the C ABI) against
  (1) the committed golden fixtures = outputs of the REAL reference, and
  (2) the CPU oracle run live on the same seeded inputs,
plus size-independent properties at BASELINE config-2 size.

Tolerances (north_star: voxel logits within 1e-3 relative, fp32):
  * kernel-level (same inputs into the HIP kernels as into the oracle): 2e-4 of the tensor scale;
  * end-to-end including the MIOpen 2-D networks: 1e-3 of the tensor scale for the voxel logits.
"""
import os

import numpy as np
import pytest
import torch

import golden_cases as gc
from test_oracle_vs_golden import (_block_module, _unet3d_module, build_product, close, flat, gold, oracle_cfg, sd_for)

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(scope="module", autouse=True)
def _gpu(hip_lib):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    torch.backends.cudnn.allow_tf32 = False
    torch.backends.cuda.matmul.allow_tf32 = False


def to_dev(batch):
    out = {}
    for k, v in batch.items():
        if isinstance(v, list):
            out[k] = [t.to(DEV) if torch.is_tensor(t) else t for t in v]
        else:
            out[k] = v.to(DEV) if torch.is_tensor(v) else v
    return out


@pytest.mark.parametrize("name", list(gc.SFA_CASES))
def test_sfa_vs_golden(name):
    from occdepth_amd.models.SFA import SFA
    spec = gc.SFA_CASES[name]
    x2d, pix, fov = gc.sfa_inputs(spec)
    m = SFA(spec["scene"], spec["dataset"], spec["ps"]).to(DEV).eval()
    with torch.no_grad():
        got = m(x2d.to(DEV), pix.to(DEV), fov.to(DEV))
    close(got.cpu(), gold("sfa")[name], tol=1e-5, what=name)


@pytest.mark.parametrize("name", list(gc.BLOCK_CASES))
def test_blocks_vs_golden(name):
    m = _block_module(name)
    m.load_state_dict(sd_for(m, "blocks3d", name))
    m = m.to(DEV).eval()
    g = gold("blocks3d")
    with torch.no_grad():
        out = m(gc.randn(gc.BLOCK_CASES[name], name).to(DEV))
    for k, v in flat(out).items():
        close(v.cpu(), g[name + ("." + k if k else "")], tol=2e-4, what=f"{name}.{k}")


@pytest.mark.parametrize("name", list(gc.UNET3D_CASES))
def test_unet3d_vs_golden(name):
    spec = gc.UNET3D_CASES[name]
    m = _unet3d_module(spec)
    m.load_state_dict(sd_for(m, "unet3d", name))
    m = m.to(DEV).eval()
    g = gold("unet3d")
    with torch.no_grad():
        out = m({"x3d": gc.randn(spec["x"], name).to(DEV)})
    assert len(out) == len([k for k in g.files if k.startswith(name + ".")])
    for k, v in out.items():
        close(gc.maybe_subsample(v.cpu().contiguous()), g[f"{name}.{k}"], tol=3e-4, what=f"{name}.{k}")


@pytest.mark.parametrize("name", list(gc.FLOSP_CASES))
def test_flosp_vs_golden(name):
    from occdepth_amd.models.flosp_depth.flosp_depth import FlospDepth
    spec = gc.FLOSP_CASES[name]
    m = FlospDepth(**spec["ctor"])
    m.load_state_dict(sd_for(m, "flosp", name))
    m = m.to(DEV).eval()
    feat, cam_k, t_v2c, idas = gc.flosp_inputs(spec)
    with torch.no_grad():
        vox, depth = m(feat.to(DEV), [k.to(DEV) for k in cam_k], [t.to(DEV) for t in t_v2c],
                       [i.to(DEV) for i in idas])
    g = gold("flosp")
    close(depth.cpu(), g[name + ".depth"], tol=2e-4, what="depth")
    close(vox.cpu(), g[name + ".vox"], tol=2e-4, what="vox")


@pytest.mark.parametrize("name", list(gc.FLOSP_CASES))
def test_flosp_kernel_vs_oracle_same_depth(name):
    """K1a alone: feed the GOLDEN depth volume so only the frustum-sample kernel is under test."""
    from occdepth_amd import hip
    from occdepth_amd.models.flosp_depth.flosp_depth import _grid_to_lidar
    spec = gc.FLOSP_CASES[name]
    ctor = spec["ctor"]
    g = gold("flosp")
    feat, cam_k, t_v2c, idas = gc.flosp_inputs(spec)
    depth = torch.from_numpy(g[name + ".depth"]).to(DEV)
    bounds = [ctor["x_bound"], ctor["y_bound"], ctor["z_bound"]]
    vnum = [int(v) for v in torch.LongTensor([(r[1] - r[0]) / r[2] / ctor["project_scale"] for r in bounds])]
    pc_range = [r[0] for r in bounds] + [r[1] for r in bounds]
    tv = torch.stack(t_v2c).float()
    k3 = torch.stack(cam_k).float()
    intr = torch.zeros(1, k3.shape[1], 4, 4)
    intr[:, :, :3, :3] = k3
    intr[:, :, 3, 3] = 1
    trans = (tv @ _grid_to_lidar(pc_range, vnum)).contiguous().to(DEV)
    flat_vol = hip.flosp_sample(depth, trans, intr[:, :, :3, :].contiguous().to(DEV),
                                torch.stack(idas).float().contiguous().to(DEV), vnum, ctor["final_dim"],
                                ctor["d_bound"][0], ctor["d_bound"][1], True)
    close(flat_vol.cpu().reshape(g[name + ".vox"].shape), g[name + ".vox"], tol=2e-5, what="vox")


@pytest.mark.parametrize("cfg_name", ["kitti_small", "nyu_small", "kitti_flosp_small"])
def test_occdepth_small_vs_golden(cfg_name):
    """Reduced configs end to end.  Two references: the golden (the REAL reference's float32 CPU run) and the float64
    value of the same function (oracle with float64 network arithmetic, computed here).  These random-init reduced
    networks are ill-conditioned: the reference's own float32 result sits 1.5e-4 .. 9e-4 from the float64 value
    (tests/test_oracle_vs_golden.py::test_reference_float32_roundoff_on_small_configs).  So the bar is applied against
    the float64 value, and the bound against the golden is the derived one, bar + (golden vs float64).  The bar here is
    2e-3, not the 1e-3 of the real configurations (config 2 and NYU config 1 are held to 1e-3 on every output): any change of
    float32 summation order in the 2-D network moves these reduced nets by up to ~1e-3 -- measured 1.4e-4 .. 7.8e-4 with the
    library GEMMs of round 3 and 1.1e-3 (kitti_flosp_small occ_logit) with K16, whose own error against float64 equals
    torch.matmul's (tests/test_gemm_x3.py) -- i.e. the number measures the net's conditioning, not a kernel."""
    from test_oracle_vs_golden import oracle_float64, rel_err
    m, cfg, sd = build_product(cfg_name)
    m = m.to(DEV).eval()
    g = gold("occdepth_small")
    with torch.no_grad():
        out = m(to_dev(gc.occdepth_batch(cfg_name)))
    assert len([v for v in out.values() if v is not None]) == len([k for k in g.files if k.startswith(cfg_name + ".")])
    truth = oracle_float64(cfg_name)
    errs, errs64, gold64 = {}, {}, {}
    for k, v in out.items():
        ref = torch.from_numpy(g[f"{cfg_name}.{k}"])
        got = gc.maybe_subsample(v.cpu().contiguous())
        assert got.shape == ref.shape, k
        errs[k] = rel_err(got, ref)
        errs64[k] = rel_err(v.cpu(), truth[k])
        gold64[k] = rel_err(ref, gc.maybe_subsample(truth[k]))
    fmt = lambda d: {k: f"{e:.1e}" for k, e in d.items()}
    print(cfg_name, "HIP vs reference golden:", fmt(errs))
    print(cfg_name, "HIP vs float64 value   :", fmt(errs64))
    print(cfg_name, "golden vs float64 value:", fmt(gold64))
    for k in ("ssc_logit", "occ_logit"):
        if k in errs:
            assert errs64[k] < 2e-3, (k, errs64)
            assert errs[k] < 2e-3 + gold64[k], (k, errs, gold64)
    # intermediates downstream of the 2-D nets and the depth softmax are un-normalised sums (|x| ~ 4e3): sanity bound.
    # Their distance to the float64 value is set by the summation order of the squeeze-excite gates upstream, not by a kernel:
    # kitti_small `x` measured 2.0e-3 (exact-fp32 MFMA everywhere), 2.8e-3 (K16, serial SE expand), 5.2e-3 (K16, float4 SE
    # expand -- the kernel that is 1e-7 from its own float64 reference); the reference's float32 run itself sits at 1.8e-3.
    assert max(errs64.values()) < 1e-2, errs64


def test_lift_and_3d_stack_vs_oracle_same_features():
    """Hand-written path in isolation: identical 2-D feature maps go to the oracle (CPU) and to the HIP
    lift + 3-D stack (GPU).  This is the parity claim for the kernels this repo owns."""
    from oracle import occdepth_oracle as orc
    from occdepth_amd.models.SFA import lift_scales
    cfg_name = "kitti_small"
    m, cfg, sd = build_product(cfg_name)
    batch = gc.occdepth_batch(cfg_name)
    ocfg = oracle_cfg(m, cfg)
    with torch.no_grad():
        x_rgb = []
        for v in range(2):
            feats = orc.encoder_features(m.net_rgb.encoder.original_model, batch["img"][:, v])
            x_rgb.append(orc.decoder_bn(sd, "net_rgb.decoder", feats))
        ps = cfg.project_scale
        x3d = orc.lift(x_rgb, ["1", "2", "4", "8"], batch[f"projected_pix_{ps}"], batch[f"fov_mask_{ps}"],
                       cfg.full_scene_size, ps, cfg.dataset)
        ref = orc.unet3d_kitti(sd, x3d, cfg.full_scene_size, ps, True, True, False, False)
        m = m.to(DEV)
        feats_dev = [[x_rgb[v]["1_" + s].to(DEV) for v in range(2)] for s in ("1", "2", "4", "8")]
        pix = torch.stack(batch[f"projected_pix_{ps}"]).to(DEV)
        fov = torch.stack(batch[f"fov_mask_{ps}"]).to(DEV)
        vox = lift_scales(feats_dev, [1, 2, 4, 8], pix, fov, cfg.full_scene_size, ps, cfg.dataset)
        close(vox.ncdhw().cpu(), x3d, tol=1e-5, what="lift")
        got = m.net_3d_decoder({"x3d": vox})
    for k in ref:
        close(got[k].cpu(), ref[k], tol=2e-4, what=k)


# ----------------------------------------------------------------------------- BASELINE config 2
@pytest.fixture(scope="module")
def config2():
    m, cfg, sd = build_product("kitti_a100")
    m = m.to(DEV).eval()
    batch = to_dev(gc.occdepth_batch("kitti_a100"))
    with torch.no_grad():
        out = m(batch)
    return m, cfg, batch, out


def test_config2_vs_reference_golden(config2):
    """Full-size forward (B7, 370x1220 stereo -> 256x256x32) against the sub-sampled outputs of the real
    reference run in the build container (tests/golden/occdepth_kitti_a100.npz)."""
    m, cfg, batch, out = config2
    g = gold("occdepth_kitti_a100")
    assert out["ssc_logit"].shape == (1, 20, 256, 256, 32) and out["occ_logit"].shape == (1, 2, 256, 256, 32)
    assert out["P_logits"].shape == (1, 4, 512, 4096) and out["depth_pred"].shape == (1, 2, 104, 47, 153)
    worst = {}
    for k, v in out.items():
        ref = torch.from_numpy(g[k])
        got = gc.subsample(v.cpu().contiguous())
        assert got.shape == ref.shape, k
        worst[k] = ((got - ref).abs().max() / ref.abs().max()).item()
    print("config-2 relative errors vs reference:", {k: f"{e:.2e}" for k, e in worst.items()})
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/config2_parity.txt", "w") as f:
        f.write(repr(worst) + "\n")
    assert worst["ssc_logit"] < 1e-3 and worst["occ_logit"] < 1e-3, worst
    assert max(worst.values()) < 1e-3, worst          # every intermediate too (measured worst: P_logits 5.0e-4)


def config2_errors(out):
    g = gold("occdepth_kitti_a100")
    worst = {}
    for k, v in out.items():
        ref = torch.from_numpy(g[k])
        got = gc.subsample(v.cpu().contiguous())
        assert got.shape == ref.shape, k
        worst[k] = ((got - ref).abs().max() / ref.abs().max()).item()
    return worst


def test_config2_benched_flags_vs_reference_golden():
    """The 2-D half of the benched configuration (`batch_views=True`, `graph_2d=True`: both views through the 2-D
    network as one batch, replayed from a captured hipGraph; in-repo 2-D kernels on; the lift from the batch's own tables)
    against the real reference's config-2 golden, on the capture pass and on two replays with the input buffer refreshed in
    between.  EXACTLY what bench.py times is `test_config2_exactly_benched_configuration_vs_reference_golden` below."""
    m, cfg, sd = build_product("kitti_a100")
    m = m.to(DEV).eval()
    m.batch_views, m.graph_2d = True, True
    batch = to_dev(gc.occdepth_batch("kitti_a100"))
    other = dict(batch, img=torch.randn_like(batch["img"]))
    with torch.no_grad():
        runs = [m(batch)]                 # captures
        m(other)                          # replay on different pixels (the static input buffer must be refreshed)
        runs.append(m(batch))             # replay
        runs.append(m(batch))
    assert m.graph_2d, "hipGraph capture fell back to eager: the benched configuration did not run"
    assert len(m._graphs) == 1
    for i, out in enumerate(runs):
        worst = config2_errors(out)
        print(f"config-2, benched flags, pass {i}: relative errors vs reference:", {k: f"{e:.2e}" for k, e in worst.items()})
        assert worst["ssc_logit"] < 1e-3 and worst["occ_logit"] < 1e-3, (i, worst)
        assert max(worst.values()) < 1e-3, (i, worst)    # every intermediate too
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/config2_parity_benched_flags.txt", "w") as f:
        f.write(repr(config2_errors(runs[-1])) + "\n")


def benched_batch(device=DEV):
    """The golden config-2 frame in the form bench.py hands to the model: the dataloader's float64 extrinsics and NO
    voxel->pixel tables, so the eval forward projects and samples inside the lift kernel (`lift_in_kernel` "auto")."""
    from occdepth_amd import synthetic
    b = {k: v for k, v in gc.occdepth_batch("kitti_a100").items()
         if not (k.startswith("projected_pix") or k.startswith("fov_mask"))}
    b["T_velo_2_cam_f64"] = synthetic.kitti_frame(seed=gc.SEED)["T_velo_2_cam_f64"]
    return to_dev(b) if device == DEV else b


@pytest.mark.parametrize("mode", [("head", True), (False, True), (False, False)],
                         ids=["bf16x3_default", "exact_fp32_convs3d", "exact_fp32_everywhere"])
@pytest.mark.parametrize("clone", [True, False], ids=["fresh_outputs", "static_outputs"])
def test_config2_exactly_benched_configuration_vs_reference_golden(mode, clone):
    """EXACTLY the configuration `bench.py` times -- `enable_fast_eval()` (batch_views + graph_2d +
    graph_all: the whole forward replayed from ONE hipGraph), the table-free lift (float64 extrinsics, no tables in the
    batch) and the default convolution mode (head convolutions on the 3-way bf16 split; the exact-fp32 mode as the second
    parameter; the third is what the bench line's `dtype` string calls "exact fp32 everywhere": OCCDEPTH_BF16X3=0 AND
    OCCDEPTH_GEMM_X3=0, i.e. exact-fp32 MFMA convolutions and the library's fp32 GEMMs in the 2-D network -- VERDICT r4 weak
    #1) -- against the REAL reference's config-2 golden: every output < 1e-3 on the capture pass and on two replays
    with a different frame in between.  An eager twin of the same model proves through the launch profile which kernels the
    graph contains (`sfa_lift_proj`, neither `sfa_lift` nor `flosp_sample`; `conv3d_c32x3` iff the split is on)."""
    from occdepth_amd import fused, hip
    split, gemm_x3 = mode
    saved, saved_gx3 = fused.BF16X3, hip.GEMM_X3
    fused.set_bf16x3(split)
    hip.GEMM_X3 = gemm_x3
    try:
        m, cfg, sd = build_product("kitti_a100")
        m = m.to(DEV).eval()
        assert not (m.batch_views or m.graph_2d or m.graph_all)          # the fast path is opt-in
        batch = benched_batch()
        # ---- eager twin: which kernels does this configuration launch?
        m.batch_views = True
        with torch.no_grad(), hip.profile() as prof:
            m(batch)
            torch.cuda.synchronize()
        tags = {k.split(":")[0] for k in prof.rows}
        assert "sfa_lift_proj" in tags and "sfa_lift" not in tags and "flosp_sample" not in tags, sorted(tags)
        assert ("conv3d_c32x3" in tags) == (split == "head") and ("conv3d_c32p" in tags) == (split is False), sorted(tags)
        assert any(t.startswith("gemm_f32x3") for t in tags) == gemm_x3, sorted(tags)
        # ---- the benched path
        m.enable_fast_eval(clone_outputs=clone)
        assert m.batch_views and m.graph_2d and m.graph_all and m.clone_graph_outputs == clone
        other = dict(batch, img=torch.randn_like(batch["img"]))
        kept = None
        with torch.no_grad():
            for i in range(4):
                out = m(other if i == 1 else batch)       # 0: capture pass, 1: replay on a different frame, 2-3: replays
                if i == 1:
                    continue
                worst = config2_errors(out)               # (compared at once: static outputs are overwritten by the next forward)
                print(f"config-2, EXACTLY benched ({'split' if split else 'fp32' if gemm_x3 else 'fp32 everywhere'}, {'fresh' if clone else 'static'} outputs), "
                      f"pass {i}:", {k: f"{e:.2e}" for k, e in worst.items()})
                assert worst["ssc_logit"] < 1e-3 and worst["occ_logit"] < 1e-3, (i, worst)
                assert max(worst.values()) < 1e-3, (i, worst)
                if i == 0:
                    kept = out["ssc_logit"]
                    kept_copy = kept.clone()
        assert m.graph_all, f"the whole-forward capture fell back to eager: {getattr(m, 'graph_all_error', None)}"
        assert [k[0] for k in m._graphs] == ["all"]
        # fresh outputs survive later forwards; static ones are the graph's buffers (documented deviation, opt-in)
        if clone:
            assert torch.equal(kept, kept_copy) and kept.data_ptr() != out["ssc_logit"].data_ptr()
        os.makedirs("gpurun_out", exist_ok=True)
        with open(f"gpurun_out/config2_parity_exactly_benched_{'split' if split else 'fp32' if gemm_x3 else 'fp32_everywhere'}.txt", "w") as f:
            f.write(repr(worst) + "\n")
    finally:
        fused.set_bf16x3(saved)
        hip.GEMM_X3 = saved_gx3


def test_config2_properties(config2):
    """Size-independent properties at full size."""
    m, cfg, batch, out = config2
    # determinism of the hand-written path: the same lifted volume through the HIP 3-D stack twice is
    # bit-identical (no atomics, fixed reduction order).  The MIOpen 2-D nets may switch algorithm after
    # their first (find-mode) call, so whole-forward repeats are only required to agree to round-off.
    from occdepth_amd import hip
    vox = hip.Vox.from_ncdhw(torch.randn(1, 64, 128, 128, 16, device=DEV))
    with torch.no_grad():
        a = m.net_3d_decoder({"x3d": vox})
        b = m.net_3d_decoder({"x3d": vox})
        again = m(batch)
    for k in a:
        assert torch.equal(a[k], b[k]), k
    for k in out:
        assert ((out[k] - again[k]).abs().max() / out[k].abs().max()).item() < 1e-3, k
    # batch independence: frames in a batch of 2 equal their single-frame results (eval BN)
    b2 = {k: ([v[0], v[0]] if isinstance(v, list) else torch.cat([v, v.flip(1)])) for k, v in batch.items()}
    b2["projected_pix_2"] = [batch["projected_pix_2"][0], batch["projected_pix_2"][0].flip(0)]
    b2["fov_mask_2"] = [batch["fov_mask_2"][0], batch["fov_mask_2"][0].flip(0)]
    b2["cam_k"] = [batch["cam_k"][0], batch["cam_k"][0].flip(0)]
    b2["T_velo_2_cam"] = [batch["T_velo_2_cam"][0], batch["T_velo_2_cam"][0].flip(0)]
    with torch.no_grad():
        o2 = m(b2)
    assert o2["ssc_logit"].shape[0] == 2
    e = ((o2["ssc_logit"][0] - out["ssc_logit"][0]).abs().max() / out["ssc_logit"].abs().max()).item()
    assert e < 1e-3, e   # MIOpen picks other 2-D algorithms at batch 2; same class as its run-to-run noise
    # swapping the stereo views (features, projections, calibration) leaves the SFA fusion symmetric:
    # the lifted volume, hence every logit, is unchanged up to round-off
    e = ((o2["ssc_logit"][1] - out["ssc_logit"][0]).abs().max() / out["ssc_logit"].abs().max()).item()
    assert e < 1e-3, e


def test_nyu_config1_vs_reference_golden():
    """BASELINE configs[0] at full size on the GPU: NYUv2 RGB-D frame + virtual stereo view, B4, feature 100
    (channel pads 100 -> 104, 25 -> 32), odd volume sizes 60x36x60 -> 15x9x15."""
    m, cfg, sd = build_product("nyu_2080ti")
    m = m.to(DEV).eval()
    with torch.no_grad():
        out = m(to_dev(gc.occdepth_batch("nyu_2080ti")))
    g = gold("occdepth_nyu_2080ti")
    assert out["ssc_logit"].shape == (1, 12, 60, 36, 60)
    errs = {}
    for k, v in out.items():
        ref = torch.from_numpy(g[k])
        got = gc.subsample(v.cpu().contiguous())
        assert got.shape == ref.shape, k
        errs[k] = ((got - ref).abs().max() / ref.abs().max()).item()
    print("config-1 (NYU) relative errors vs reference:", {k: f"{e:.2e}" for k, e in errs.items()})
    assert max(errs.values()) < 1e-3, errs            # every output, intermediates included


def test_config5_synthetic_512_grid():
    """BASELINE configs[4]: UNet3D(kitti) alone on a 512x512x64 grid (lift volume 256x256x32, feature 64; CRP with
    N = 32768 voxels / M = 4096 mega voxels, P_logits 2.1 GB, 9.3 TFLOP) against the REAL reference's CPU run of the
    same seeded weights and input (tests/golden/unet3d_512.npz: sub-sampled outputs, |max| and sum of every full
    output), plus determinism and one full-resolution head convolution (Z = 64, generic kernel) against ATen."""
    import torch.nn as nn
    import torch.nn.functional as F
    from occdepth_amd import hip
    from occdepth_amd.models.unet3d_kitti import UNet3D
    from test_oracle_vs_golden import sd_for
    spec = gc.UNET3D_512
    m = UNet3D(spec["classes"], nn.BatchNorm3d, spec["scene"], spec["feature"], spec["ps"], context_prior=True,
               cascade_cls=True)
    m.load_state_dict(sd_for(m, "unet3d_512", "unet3d_512"))
    m = m.to(DEV).eval()
    g = gold("unet3d_512")
    x = hip.Vox.from_ncdhw(gc.randn(spec["x"], "unet3d_512").to(DEV))
    with torch.no_grad():
        out = m({"x3d": x})
        assert out["ssc_logit"].shape == (1, 20, 512, 512, 64) and out["P_logits"].shape == (1, 4, 4096, 32768)
        errs = {}
        for k, v in out.items():
            ref = torch.from_numpy(g[f"unet3d_512.{k}"]).to(DEV)
            got = gc.subsample_512(v)
            assert got.shape == ref.shape, k
            scale = float(g[f"unet3d_512.{k}.absmax"])
            errs[k] = ((got - ref).abs().max() / scale).item()
            assert float(v.abs().max()) == pytest.approx(scale, rel=3e-4), k
            total, want = float(v.double().sum()), float(g[f"unet3d_512.{k}.sum"])
            assert abs(total - want) <= 3e-4 * scale * v.numel() ** 0.5 + 1e-5 * abs(want), (k, total, want)
        print("config-5 (512x512x64) relative errors vs reference:", {k: f"{e:.2e}" for k, e in errs.items()})
        assert max(errs.values()) < 3e-4, errs
        first = out["ssc_logit"][0, :, ::37, ::41, ::7].clone()
        del out
        again = m({"x3d": x})["ssc_logit"][0, :, ::37, ::41, ::7]
        assert torch.equal(first, again)
        # one head convolution at this resolution vs ATen
        conv = m.ssc_head.conv1[1]
        xin = torch.randn(1, 32, 48, 512, 64, device=DEV)
        ref = F.conv3d(xin, conv.weight, None, padding=2, dilation=2)
        o = hip.Vox.empty(1, (48, 512, 64), 32, DEV)
        hip.conv3d(hip.Vox.from_ncdhw(xin), hip.pack_weights(conv.weight), None, 32, (3, 3, 3), o,
                   dilation=(2, 2, 2), padding=(2, 2, 2))
        assert ((o.ncdhw() - ref).abs().max() / ref.abs().max()).item() < 2e-5


@pytest.mark.parametrize("geom", ["kitti_ps2", "kitti_ps1_right", "nyu"])
def test_project_voxels_bit_exact(geom):
    """the GPU voxel->pixel projection reproduces the numpy restatement of the dataloader's
    numba vox2pix integer-exactly (int64 pixels, bool FOV mask), full-size grids."""
    import numpy as np
    from occdepth_amd import hip
    from oracle import inputs
    if geom.startswith("kitti"):
        ps = 2 if geom == "kitti_ps2" else 1
        E = inputs.KITTI_TR.copy()
        if geom.endswith("right"):
            E[0, 3] = -0.54
        k, origin, vs, dims, wh, scene_m = inputs.KITTI_K, (0, -25.6, -2), 0.2 * ps, (256 // ps, 256 // ps, 32 // ps), \
            (1220, 370), (51.2, 51.2, 6.4)
    else:
        pose = np.array([[1, 0, 0, 2.4], [0, 0, 1, -0.5], [0, -1, 0, 1.44], [0, 0, 0, 1]], dtype=np.float64)
        E, k, origin, vs, dims, wh, scene_m = np.linalg.inv(pose), inputs.NYU_K, (0.3, -0.2, 0.1), 0.08, \
            (60, 60, 36), (640, 480), (4.8, 4.8, 2.88)
    ref_pix, ref_fov, ref_z = inputs.vox2pix(E, k, origin, vs, wh[0], wh[1], scene_m, 0)
    pix, fov, z = hip.project_voxels(E, k, origin, vs, dims, wh[0], wh[1], with_z=True)
    assert pix.shape == (ref_pix.shape[0], 1, 2) and pix.dtype == torch.int64 and fov.dtype == torch.bool
    assert np.array_equal(fov.cpu().numpy(), ref_fov)
    inside = torch.from_numpy(ref_fov[:, 0])
    assert np.array_equal(pix.cpu().numpy()[ref_fov[:, 0]], ref_pix[ref_fov[:, 0]])
    # outside the FOV only the (masked) values of degenerate z == 0 projections may differ; none here
    assert np.array_equal(pix.cpu().numpy(), ref_pix)
    assert np.allclose(z.cpu().numpy(), ref_z.astype(np.float32), rtol=1e-6, atol=1e-6)
    assert 0.3 < inside.float().mean().item() < 0.95


def test_forward_without_projection_inputs(config2):
    """N2 end to end: dropping `projected_pix_2` / `fov_mask_2` from the batch makes the model project the voxels
    on the GPU from (cam_k, T_velo_2_cam).
    (1) With the dataloader's float64 extrinsics (`T_velo_2_cam_f64`) the tables are bit-identical, so the logits meet
        the same 1e-3 bar as any repeated forward.
    (2) The reference batch only carries a float32 copy of the extrinsics: a few pixels move by one in rounding ties
        (< 1e-4 of the table).  That is a different INPUT for those voxels, not round-off, so the check is that the
        change stays local: fewer than 1e-3 of the voxels see any logit move by more than 1e-3."""
    m, cfg, batch, out = config2
    b = {k: v for k, v in batch.items() if not (k.startswith("projected_pix") or k.startswith("fov_mask"))}
    ref_pix = torch.stack(batch["projected_pix_2"])
    ref_fov = torch.stack(batch["fov_mask_2"])
    scale = out["ssc_logit"].abs().max()
    from occdepth_amd import synthetic                  # same calibration constants, float64 extrinsics included
    b64 = dict(b, T_velo_2_cam_f64=[t.to(DEV) for t in synthetic.kitti_frame(seed=gc.SEED)["T_velo_2_cam_f64"]])
    with torch.no_grad():
        pix, fov = m.project_voxels_on_gpu(b64, batch["img"])
        o64 = m(b64)
    assert torch.equal(pix, ref_pix) and torch.equal(fov, ref_fov)
    assert ((o64["ssc_logit"] - out["ssc_logit"]).abs().max() / scale).item() < 1e-3
    with torch.no_grad():
        pix, fov = m.project_voxels_on_gpu(b, batch["img"])
        o = m(b)
    assert (fov != ref_fov).float().mean().item() < 1e-4
    both = fov & ref_fov
    assert ((pix != ref_pix).any(-1) & both).float().mean().item() < 1e-4
    moved = ((o["ssc_logit"] - out["ssc_logit"]).abs().amax(1) / scale) > 1e-3
    assert moved.float().mean().item() < 1e-3, moved.float().mean().item()


def test_in_kernel_lift_matches_table_path(config2):
    """with the dataloader's float64 extrinsics in the batch the eval forward takes the fused lift
    (projection + frustum sample inside the kernel, no tables read); switching it off (table path) must give the same
    logits to float32 round-off (the two lift kernels differ by 3 ulp, tests/test_lift_proj.py), and the fused path must not launch the standalone frustum sample / table lift."""
    from occdepth_amd import hip, synthetic
    m, cfg, batch, out = config2
    b64 = dict(batch, T_velo_2_cam_f64=[t.to(DEV) for t in synthetic.kitti_frame(seed=gc.SEED)["T_velo_2_cam_f64"]])
    saved = (m.lift_in_kernel, getattr(m, "graph_all", False))
    try:
        m.graph_all = False
        m.lift_in_kernel = True
        with torch.no_grad(), hip.profile() as prof:
            o_k = m(b64)
            torch.cuda.synchronize()
        tags = {k.split(":")[0] for k in prof.rows}
        assert "sfa_lift_proj" in tags and "sfa_lift" not in tags and "flosp_sample" not in tags, sorted(tags)
        m.lift_in_kernel = False
        with torch.no_grad():
            o_t = m(b64)
        scale = out["ssc_logit"].abs().max()
        assert ((o_k["ssc_logit"] - o_t["ssc_logit"]).abs().max() / scale).item() < 5e-4   # (3-ulp inputs through 60 layers)
        assert ((o_k["ssc_logit"] - out["ssc_logit"]).abs().max() / scale).item() < 1e-3
    finally:
        m.lift_in_kernel, m.graph_all = saved


def test_argmax_labels(config2):
    """N4: GPU arg-max over the channels-last logits == numpy argmax of the softmax (generate_output.py:94-95)."""
    import numpy as np
    from occdepth_amd import hip
    m, cfg, batch, out = config2
    labels = hip.argmax_labels(out["ssc_logit"])
    ref = np.argmax(torch.softmax(out["ssc_logit"], dim=1).cpu().numpy(), axis=1)
    assert labels.shape == (1, 256, 256, 32)
    assert (labels.cpu().numpy() != ref).mean() < 1e-6       # softmax rounding can only flip exact ties
    ref_raw = np.argmax(out["ssc_logit"].cpu().numpy(), axis=1)
    assert np.array_equal(labels.cpu().numpy(), ref_raw)
    lut = np.arange(20)[::-1].copy() * 3
    mapped = hip.argmax_labels(out["ssc_logit"], lut=lut)
    assert np.array_equal(mapped.cpu().numpy(), lut[ref_raw])
    occ = hip.argmax_labels(out["occ_logit"])               # a channel slice of a wider row buffer
    assert np.array_equal(occ.cpu().numpy(), np.argmax(out["occ_logit"].cpu().numpy(), axis=1))


def test_lift_properties_full_size():
    """K1b at config-2 size: linearity in the features, zero rows for voxels outside both FOVs."""
    from occdepth_amd.models.SFA import lift_scales
    from oracle import inputs
    b = inputs.kitti_batch(seed=1)
    pix = torch.stack(b["projected_pix_2"]).to(DEV)
    fov = torch.stack(b["fov_mask_2"]).to(DEV)
    g = torch.Generator(device=DEV).manual_seed(0)
    sizes = [(370, 1220), (185, 610), (93, 305), (47, 153)]

    def feats(scale):
        return [[scale * torch.randn(1, 64, h, w, device=DEV, generator=torch.Generator(device=DEV).manual_seed(
            97 * s + v)) for v in range(2)] for s, (h, w) in enumerate(sizes)]

    one = lift_scales(feats(1.0), [1, 2, 4, 8], pix, fov, (256, 256, 32), 2, "kitti").buf
    two = lift_scales(feats(2.0), [1, 2, 4, 8], pix, fov, (256, 256, 32), 2, "kitti").buf
    # cosine weights are scale invariant => the fused feature is homogeneous of degree 1
    assert ((two - 2 * one).abs().max() / one.abs().max()).item() < 1e-5
    outside = ~(fov[0, 0, :, 0] | fov[0, 1, :, 0])
    rows = one.reshape(-1, 64)
    assert rows[outside].abs().max().item() == 0.0
    assert rows[~outside].abs().max().item() > 0.0


def test_product_synthetic_frame_equals_oracle_batch(config2):
    """bench.py's frames (occdepth_amd/synthetic.py + the GPU projection kernel) are bit-identical to the oracle's
    numpy restatement of the dataloader (oracle/inputs.py): images, calibration, pixel tables, FOV masks."""
    from occdepth_amd import synthetic
    from oracle import inputs
    m = config2[0]
    with torch.no_grad():
        got = synthetic.attach_projection(m, synthetic.to_device(synthetic.kitti_frame(seed=5), DEV))
    want = inputs.kitti_batch(seed=5)
    for k, v in want.items():
        if isinstance(v, list):
            assert all(torch.equal(a.cpu(), b) and a.dtype == b.dtype for a, b in zip(got[k], v)), k
        else:
            assert torch.equal(got[k].cpu(), v), k


def test_graph_replay_of_2d_network_matches_eager(config2):
    """bench.py's `graph_2d` (hipGraph capture of the 2-D network, replayed per frame) reproduces the eager forward
    on fresh inputs -- the captured graph reads the new image, not the one it was captured with."""
    from occdepth_amd import synthetic
    m = config2[0]
    saved = (m.batch_views, m.graph_2d)
    try:
        m.batch_views = True
        frames = [synthetic.attach_projection(m, synthetic.to_device(synthetic.kitti_frame(seed=s), DEV)) for s in (11, 12)]
        with torch.no_grad():
            m.graph_2d = False
            eager = [{k: v.clone() for k, v in m(f).items()} for f in frames]
            m.graph_2d = True
            m._graphs = {}
            graphed = [{k: v.clone() for k, v in m(f).items()} for f in frames]        # first call captures
            again = {k: v.clone() for k, v in m(frames[0]).items()}                      # replay with frame 0 again
        assert m.graph_2d, "capture fell back to eager"
        for e, g in zip(eager, graphed):
            for k in e:
                err = ((e[k] - g[k]).abs().max() / e[k].abs().max()).item()
                assert err < 5e-4, (k, err)              # same kernels; MIOpen's Winograd differs run to run by ~2e-4
        assert not torch.equal(graphed[0]["ssc_logit"], graphed[1]["ssc_logit"])
        err = ((again["ssc_logit"] - graphed[0]["ssc_logit"]).abs().max() / again["ssc_logit"].abs().max()).item()
        assert err < 5e-4
    finally:
        m.batch_views, m.graph_2d = saved
        m._graphs = {}


def test_graph_capture_failure_falls_back_to_eager(monkeypatch):
    """`graph_2d` is an optimisation, never a requirement: when hipGraph capture raises, the model warns, records the
    reason (`graph_2d_error`, which bench.py copies into its JSON line), switches the flag off and returns the eager
    result of the same kernels."""
    m, cfg, sd = build_product("kitti_small")
    m = m.to(DEV).eval()
    batch = to_dev(gc.occdepth_batch("kitti_small"))
    m.batch_views = True
    with torch.no_grad():
        m.graph_2d = False
        eager = {k: v.clone() for k, v in m(batch).items() if v is not None}

    class Boom:
        def __init__(self, *a, **k):
            pass

        def __enter__(self):
            raise RuntimeError("capture refused (test)")

        def __exit__(self, *a):
            return False

    monkeypatch.setattr(torch.cuda, "graph", Boom)
    m.graph_2d = True
    m._graphs = {}
    with torch.no_grad(), pytest.warns(UserWarning, match="hipGraph capture"):
        out = m(batch)
    assert m.graph_2d is False and "capture refused" in m.graph_2d_error and not m._graphs
    for k, v in eager.items():
        err = ((out[k] - v).abs().max() / v.abs().max().clamp_min(1e-30)).item()
        assert err < 1e-3, (k, err)        # two eager runs of an ill-conditioned random-init net (library algorithm picks
        #                                    differ between calls): measured up to 5.0e-4
    with torch.no_grad():
        again = m(batch)                                    # stays eager, no second warning path
    assert torch.isfinite(again["ssc_logit"]).all()
