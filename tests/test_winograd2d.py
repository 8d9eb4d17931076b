"""Winograd F(2x2, 3x3) path of the 2-D decoder convolutions (N3): weight transform + tile conventions on the CPU
(test-only emulation of the two transform kernels), the HIP kernels themselves with `-m gpu`."""
import contextlib

import pytest
import torch
import torch.nn.functional as F

import emu

# (B, Cin, Cout, H, W, act, residual)
CASES = [(2, 40, 24, 12, 39, "leaky", False),      # the 1/16 level's odd width
         (1, 33, 70, 7, 5, None, False),           # ragged channels, odd sizes, fewer than one workgroup of tiles
         (2, 64, 64, 24, 77, "relu", True),        # BasicBlock: relu(bn(conv) + x)
         (1, 8, 8, 2, 130, "leaky", False)]        # more than one workgroup along x


def run(case, device, tol, strip_rows=None):
    from occdepth_amd import hip
    B, cin, cout, H, W, act, with_res = case
    g = torch.Generator().manual_seed(B * 1000 + cin + H)
    x = torch.randn(B, cin, H, W, generator=g)
    w = torch.randn(cout, cin, 3, 3, generator=g) * 0.1
    scale, shift = torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g)
    res = torch.randn(B, cout, H, W, generator=g) if with_res else None
    ref = F.conv2d(x.double(), w.double(), padding=1) * scale.double().view(1, -1, 1, 1) + shift.double().view(1, -1, 1, 1)
    if with_res:
        ref = ref + res.double()
    ref = F.leaky_relu(ref, 0.01) if act == "leaky" else (F.relu(ref) if act == "relu" else ref)
    dev = torch.device(device)
    with (emu.patched() if device == "cpu" else contextlib.nullcontext()):
        U = hip.winograd_weights(w.to(dev))
        y = hip.conv2d_3x3_winograd(x.to(dev), U, scale.to(dev), shift.to(dev), act, 0.01,
                                    res.to(dev) if with_res else None, res_first=True, strip_rows=strip_rows)
    err = float((y.double().cpu() - ref).abs().max() / ref.abs().max())
    assert y.shape == ref.shape and err < tol, (case, err)


@pytest.mark.parametrize("case", CASES)
def test_winograd_host_logic_cpu(case):
    run(case, "cpu", 2e-6)
    run(case, "cpu", 2e-6, strip_rows=3)          # strips of 3 tile rows (ragged last strip)


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES)
def test_winograd_kernels_gpu(case, hip_lib):
    torch.backends.cuda.matmul.allow_tf32 = False
    run(case, "cuda", 2e-5)
    run(case, "cuda", 2e-5, strip_rows=3)


@pytest.mark.gpu
def test_transform_kernels_match_emulation(hip_lib):
    from occdepth_amd import hip
    g = torch.Generator().manual_seed(9)
    x = torch.randn(2, 45, 13, 71, generator=g)
    V = hip.wino_input_transform(x.cuda())
    assert torch.allclose(V.cpu(), emu.wino_input_transform(x), rtol=1e-6, atol=1e-6)
    M = torch.randn(16, 2 * 7 * 36, 50, generator=g)
    res = torch.randn(2, 50, 13, 71, generator=g)
    sc, sh = torch.rand(50, generator=g), torch.randn(50, generator=g)
    for rf in (True, False):
        y = hip.wino_output_transform(M.cuda(), (2, 50, 13, 71), sc.cuda(), sh.cuda(), "leaky", 0.2, res.cuda(), rf)
        assert torch.allclose(y.cpu(), emu.wino_output_transform(M, (2, 50, 13, 71), sc, sh, "leaky", 0.2, res, rf),
                              rtol=1e-5, atol=1e-5)
