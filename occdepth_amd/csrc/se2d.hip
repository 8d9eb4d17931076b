// Squeeze-excite gate of the EfficientNet blocks, from the pooled partial sums the depthwise
// kernel leaves behind (occd_dwconv2d_pool_nchw) -- replaces x.mean((2, 3)) -> conv_reduce -> Swish -> conv_expand ->
// sigmoid (5 ATen / rocBLAS launches per block and a full extra read of the activation) by two small launches; the
// gate itself is applied to the input channels of the following pointwise GEMM (K11), never to the tensor in memory.
//   se_reduce_kernel : mean[c] = sum_j part[b][c][j] / S (fixed order);  r[b][i] = swish(sum_c Wr[i][c] mean[c] + br[i])
//   se_expand_kernel : gate[b][c] = sigmoid(sum_i We[c][i] r[b][i] + be[c])
// Both are tiny GEMVs (C <= 3840, Cr <= 160) spread over enough workgroups that no single CU streams the weights.
// Reference: geffnet SqueezeExcite behind occdepth/models/unet2d.py:175-190.
#include "common.h"

namespace {

constexpr int kROut = 4;     // reduce outputs per workgroup (1 per wave)

// Latency, not bandwidth, decides these kernels: a wave that sums 60 dependent global loads one after the other waits
// 60 HBM round trips (the first version: 26 us on the 3840-channel stages).  So every loop below is unrolled into
// independent accumulators (8 loads in flight per lane), and the sums keep a fixed order (deterministic).
__global__ void __launch_bounds__(256) se_reduce_kernel(const float* __restrict__ part, const float* __restrict__ wr,
                                                        const float* __restrict__ br, float* __restrict__ r, int C,
                                                        int Cr, int nblk, float inv_s) {
    extern __shared__ float mean[];                       // C floats
    const int b = blockIdx.y;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const float* pb = part + (size_t)b * C * nblk;
    if (nblk <= 4) {
        for (int c = threadIdx.x; c < C; c += 256) {
            float s = 0.f;
            for (int j = 0; j < nblk; ++j) s += pb[(size_t)c * nblk + j];
            mean[c] = s * inv_s;
        }
    } else {
        // many partials per channel (high-resolution stages): a wave per channel, lanes across the partials
        for (int c = wave; c < C; c += 4) {
            const float* pc = pb + (size_t)c * nblk;
            float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
            int j = lane;
            for (; j + 192 < nblk; j += 256) {
                s0 += pc[j]; s1 += pc[j + 64]; s2 += pc[j + 128]; s3 += pc[j + 192];
            }
            for (; j < nblk; j += 64) s0 += pc[j];
            float s = (s0 + s1) + (s2 + s3);
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
            if (lane == 0) mean[c] = s * inv_s;
        }
    }
    __syncthreads();
    const int i = blockIdx.x * kROut + wave;
    if (i >= Cr) return;                                  // wave-uniform
    const float* w = wr + (size_t)i * C;
    float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int c = lane;
    for (; c + 7 * 64 < C; c += 8 * 64) {
#pragma unroll
        for (int u = 0; u < 8; ++u) a[u] += w[c + u * 64] * mean[c + u * 64];
    }
    for (; c < C; c += 64) a[0] += w[c] * mean[c];
    float s = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    if (lane == 0) {
        const float v = s + br[i];
        r[(size_t)b * Cr + i] = v / (1.f + expf(-v));
    }
}

__global__ void __launch_bounds__(256) se_expand_kernel(const float* __restrict__ r, const float* __restrict__ we,
                                                        const float* __restrict__ be, float* __restrict__ gate, int C,
                                                        int Cr) {
    extern __shared__ float rs[];                         // Cr floats
    const int b = blockIdx.y;
    for (int i = threadIdx.x; i < Cr; i += 256) rs[i] = r[(size_t)b * Cr + i];
    __syncthreads();
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    const float* w = we + (size_t)c * Cr;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    int i = 0;
    for (; i + 3 < Cr; i += 4) {
        a0 += w[i] * rs[i]; a1 += w[i + 1] * rs[i + 1]; a2 += w[i + 2] * rs[i + 2]; a3 += w[i + 3] * rs[i + 3];
    }
    for (; i < Cr; ++i) a0 += w[i] * rs[i];
    const float s = be[c] + ((a0 + a1) + (a2 + a3));
    gate[(size_t)b * C + c] = 1.f / (1.f + expf(-s));
}

}  // namespace

extern "C" int occd_se_gate(const float* pool_part, const float* w_reduce, const float* b_reduce, const float* w_expand,
                            const float* b_expand, float* r_scratch, float* gate, int32_t batch, int32_t C, int32_t Cr,
                            int32_t nblk, int64_t S, void* stream) {
    if (!pool_part || !w_reduce || !b_reduce || !w_expand || !b_expand || !r_scratch || !gate) return OCCD_EINVAL;
    if (batch < 1 || batch > 65535 || C < 1 || C > 16384 || Cr < 1 || Cr > 4096 || nblk < 1 || S < 1) return OCCD_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    occd::ProfScope prof("se_gate", st, 4.0 * batch * C * Cr, 8.0 * C * Cr + 4.0 * batch * C * nblk);
    hipLaunchKernelGGL(se_reduce_kernel, dim3((unsigned)((Cr + kROut - 1) / kROut), (unsigned)batch), dim3(256),
                       (size_t)C * sizeof(float), st, pool_part, w_reduce, b_reduce, r_scratch, C, Cr, nblk,
                       (float)(1.0 / (double)S));
    hipLaunchKernelGGL(se_expand_kernel, dim3((unsigned)((C + 255) / 256), (unsigned)batch), dim3(256),
                       (size_t)Cr * sizeof(float), st, r_scratch, w_expand, b_expand, gate, C, Cr);
    return occd::check_launch();
}
