"""K14 (csrc/bneck3d.hip, occd_bottleneck3d_fwd): one stride-1 DDR Bottleneck3D as two launches, against the module's own
differentiable ATen graph (the reference's formulation, occdepth/models/DDR.py:111-139) evaluated in float64 on the CPU.
CPU: the packing host logic (BatchNorm folding, tap / channel order of the packed buffer) through the emulation.
GPU: the kernels at the three shapes of the config-2 stack (C = 64 / 128 at Z = 16 / 8 / 4; P = 64 and odd Z fall back to the five launches), ragged X / Y extents,
batch 2, dilations 1-3, and the five-launch K2 form of the same block on the same inputs."""
import copy

import pytest
import torch
import torch.nn as nn

import emu

# name: (C, P, dims, dilation, batch)
CASES = {
    "l1_d1": (64, 16, (9, 12, 16), 1, 1),
    "l1_d3": (64, 16, (13, 7, 16), 3, 2),
    "l1_d2_ragged": (64, 16, (5, 3, 16), 2, 1),
    "l2_d2": (128, 32, (10, 9, 8), 2, 1),
    "l2_d3": (128, 32, (7, 8, 8), 3, 2),
    "l2_z4": (128, 32, (6, 5, 4), 1, 1),
    "l1_z8_d2": (64, 16, (5, 9, 8), 2, 2),
    "crp_p64_fallback": (256, 64, (6, 5, 4), 1, 1),        # P = 64 and odd Z: the five-launch form (K14 declines)
    "z5_fallback": (64, 16, (6, 6, 5), 2, 1),
}


def make_block(C, P, d, seed):
    from occdepth_amd.models.DDR import Bottleneck3D
    torch.manual_seed(seed)
    m = Bottleneck3D(C, P, nn.BatchNorm3d, dilation=[d, d, d], expansion=C // P)
    for mod in m.modules():
        if isinstance(mod, nn.BatchNorm3d):
            mod.running_mean.normal_(0, 0.3)
            mod.running_var.uniform_(0.5, 1.5)
            mod.weight.data.uniform_(0.6, 1.4)
            mod.bias.data.normal_(0, 0.2)
    return m.eval()


def reference(m, x):
    with torch.no_grad():
        return copy.deepcopy(m).double()._forward_autograd(x.double())


@pytest.mark.parametrize("name", ["l1_d3", "l2_d2", "l1_z8_d2"])
def test_bottleneck_packing_host_logic_cpu(name, monkeypatch):
    C, P, dims, d, B = CASES[name]
    m = make_block(C, P, d, seed=len(name))
    x = torch.randn(B, C, *dims)
    from occdepth_amd.models.DDR import Bottleneck3D
    monkeypatch.setattr(Bottleneck3D, "FUSED", True)
    calls = []
    with emu.patched(), torch.no_grad():
        from occdepth_amd import hip
        real = hip.bottleneck3d
        hip.bottleneck3d = lambda *a, **k: (calls.append(1), real(*a, **k))[1]
        y = m(x)
    assert calls, "the fused block was not taken"
    ref = reference(m, x)
    assert float((y.double() - ref).abs().max() / ref.abs().max()) < 1e-5


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(CASES))
def test_bottleneck_fused_vs_float64(name, hip_lib):
    from occdepth_amd.models.DDR import Bottleneck3D
    C, P, dims, d, B = CASES[name]
    m = make_block(C, P, d, seed=len(name))
    x = torch.randn(B, C, *dims)
    ref = reference(m, x)
    mg = copy.deepcopy(m).cuda()
    saved = Bottleneck3D.FUSED
    try:
        with torch.no_grad():
            Bottleneck3D.FUSED = True
            y = mg(x.cuda())
            Bottleneck3D.FUSED = False
            y5 = mg(x.cuda())
    finally:
        Bottleneck3D.FUSED = saved
    err = float((y.double().cpu() - ref).abs().max() / ref.abs().max())
    err5 = float((y5.double().cpu() - ref).abs().max() / ref.abs().max())
    print(f"{name}: K14 {err:.2e}, five K2 launches {err5:.2e}")
    assert err < 2e-5 and err5 < 2e-5, (name, err, err5)


@pytest.mark.gpu
def test_bottleneck_fused_is_the_path_taken_and_timed(hip_lib):
    """Full-size l1 block (128 x 128 x 16, C = 64): the fused launch pair is what runs (profile tags), and its time next to
    the five-launch form is printed for the record."""
    from occdepth_amd import hip
    from occdepth_amd.models.DDR import Bottleneck3D
    m = make_block(64, 16, 3, seed=3).cuda()
    x = torch.randn(1, 64, 128, 128, 16, device="cuda")
    saved = Bottleneck3D.FUSED
    Bottleneck3D.FUSED = True
    with torch.no_grad():
        m(x)
        with hip.profile() as prof:
            for _ in range(5):
                m(x)
            torch.cuda.synchronize()
        tags = {k.split(":")[0]: v for k, v in prof.rows.items()}
        assert "bottleneck3d" in tags and not any(t.startswith("conv3d") for t in tags), sorted(tags)
        fused_ms = tags["bottleneck3d"]["ms"] / 5
        Bottleneck3D.FUSED = False
        try:
            m(x)
            with hip.profile() as prof5:
                for _ in range(5):
                    m(x)
                torch.cuda.synchronize()
        finally:
            Bottleneck3D.FUSED = saved
        five_ms = sum(v["ms"] for v in prof5.rows.values()) / 5
    print(f"Bottleneck3D 64/16 d3 @128x128x16: K14 {fused_ms * 1e3:.1f} us, five K2 launches {five_ms * 1e3:.1f} us")
    assert fused_ms < five_ms
