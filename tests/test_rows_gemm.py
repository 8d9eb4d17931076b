"""K15 (csrc/rows_gemm.hip, occd_rows_gemm_fwd): pointwise convolution / row GEMM on channels-last voxel rows against
float64 matmul, at the shapes of the CRP (occdepth/models/CRP3D.py:54-97: 256->512 logits, sigmoid(logits) @ mega rows into a
channel slice of the 2304-wide concat rows, 2304->256 resize) and of the strided bottlenecks' 1x1x1 convolutions, with bias,
residual, activations, ragged row counts and channel slices on both sides."""
import pytest
import torch

pytestmark = pytest.mark.gpu

# name: (rows as (B, X, Y, Z), K, N, a_coff, a_extra, out_coff, out_extra, act_in, act_out, bias, res)
CASES = {
    "crp_logits": ((1, 32, 32, 4), 256, 512, 0, 0, 0, 0, "none", "none", True, False),
    "crp_bmm_sigmoid_slice": ((1, 32, 32, 4), 512, 512, 0, 0, 768, 1024, "sigmoid", "none", False, False),
    "crp_resize": ((1, 8, 8, 4), 2304, 256, 0, 0, 0, 0, "none", "relu", True, False),
    "bneck_conv5_res": ((2, 5, 7, 8), 32, 128, 0, 0, 0, 0, "relu", "relu", True, True),
    "ragged_rows_n24": ((1, 3, 5, 7), 48, 24, 8, 16, 8, 8, "none", "relu_pre", True, True),
    "wide_k_small_n": ((1, 4, 4, 4), 640, 16, 0, 0, 0, 0, "none", "none", False, False),
    "n136": ((1, 6, 5, 4), 64, 136, 0, 0, 0, 0, "relu", "none", True, False),
}


@pytest.mark.parametrize("name", list(CASES))
def test_rows_gemm_vs_float64(name, hip_lib):
    from occdepth_amd import hip
    from occdepth_amd.hip import Vox
    dims4, K, N, a_coff, a_extra, o_coff, o_extra, act_in, act_out, has_bias, has_res = CASES[name]
    acts = {"none": hip.ACT_NONE, "relu": hip.ACT_RELU, "sigmoid": hip.ACT_SIGMOID, "relu_pre": hip.ACT_RELU_PRE}
    g = torch.Generator().manual_seed(sum(map(ord, name)))
    B, dims = dims4[0], dims4[1:]
    abuf = torch.randn(B, *dims, a_coff + K + a_extra, generator=g)
    w = torch.randn(K, N, generator=g) / K ** 0.5
    bias = torch.randn(N, generator=g) if has_bias else None
    rbuf = torch.randn(B, *dims, N + 8, generator=g) if has_res else None
    a64 = abuf[..., a_coff:a_coff + K].double()
    a64 = torch.relu(a64) if act_in == "relu" else torch.sigmoid(a64) if act_in == "sigmoid" else a64
    ref = a64 @ w.double()
    if bias is not None:
        ref = ref + bias.double()
    if act_out == "relu_pre":
        ref = torch.relu(ref)
    if rbuf is not None:
        ref = ref + rbuf[..., 4:4 + N].double()
    if act_out == "relu":
        ref = torch.relu(ref)
    obuf = torch.full((B, *dims, o_coff + N + o_extra), 7.0, device="cuda")
    bpad = None
    if bias is not None:
        bpad = torch.zeros(-(-N // 32) * 32, device="cuda")
        bpad[:N] = bias.cuda()
    a = Vox(abuf.cuda(), K, a_coff)
    out = Vox(obuf, N, o_coff)
    res = Vox(rbuf.cuda(), N, 4) if rbuf is not None else None
    assert hip.rows_gemm_supported(K, N, a, out, res)
    hip.rows_gemm(a, w.cuda(), out, bias=bpad, res=res, act_in=acts[act_in], act_out=acts[act_out])
    got = obuf[..., o_coff:o_coff + N].double().cpu()
    err = float((got - ref).abs().max() / ref.abs().max())
    assert err < 2e-5, (name, err)
    # nothing outside the [coff, coff + N) slice is touched
    assert float((obuf[..., :o_coff] - 7.0).abs().max()) == 0.0 if o_coff else True
    assert float((obuf[..., o_coff + N:] - 7.0).abs().max()) == 0.0 if o_extra else True


def test_rows_gemm_rejects_bad_geometry(hip_lib):
    from occdepth_amd import hip
    from occdepth_amd.hip import Vox
    a = Vox(torch.zeros(1, 2, 2, 2, 24, device="cuda"), 24)
    out = Vox(torch.zeros(1, 2, 2, 2, 16, device="cuda"), 16)
    with pytest.raises(RuntimeError):                                   # K not a multiple of 16
        hip.rows_gemm(a, torch.zeros(24, 16, device="cuda"), out)
    assert not hip.rows_gemm_supported(24, 16, a, out)
    assert not hip.rows_gemm_supported(16, 20, a, Vox(torch.zeros(1, 2, 2, 2, 24, device="cuda"), 20))
