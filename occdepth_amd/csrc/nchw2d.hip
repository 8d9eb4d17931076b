// 2-D (NCHW) memory-bound helpers around the MIOpen convolutions of the 2-D UNet (SURVEY.md 8f row N3,
// first step): every kernel replaces a chain of separate ATen elementwise / copy launches by ONE pass.
//
//   affine_act_kernel    y = act(x * scale[c] + shift[c]) (+ residual)   BatchNorm(eval)+Swish/LeakyReLU/ReLU
//                        (+ the MBConv skip add): 1 read + 1 write instead of 3-4 passes
//                        (efficientnet blocks bn1+act1 / bn2+act2 / bn3 (+skip); decoder conv+BN+LeakyReLU)
//   dwconv2d_kernel      depthwise k x k convolution with TensorFlow "SAME" padding computed in-kernel
//                        (no F.pad copy) and the following BatchNorm+Swish fused in the epilogue; replaces
//                        MIOpen's naive grouped-conv fallback
//   upsample_cat_kernel  out[:, :C] = bilinear(x, (H, W), align_corners=True), out[:, C:] = skip
//                        (F.interpolate + torch.cat of unet2d.py:38-46 in one pass)
//
// Reference semantics: occdepth/models/unet2d.py:24-46 and the geffnet EfficientNet blocks (third party).
#include <cstdlib>
#include "common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

namespace {

__device__ __forceinline__ float act_apply(float v, int act, float slope) {
    if (act == 1) return fmaxf(v, 0.f);                    // relu
    if (act == 2) return occd::swish_fast(v);              // swish / silu
    if (act == 3) return v > 0.f ? v : v * slope;          // leaky relu
    return v;
}

// one (b, c) plane per blockIdx.y, grid-stride over the plane; float4 when the plane size allows
__global__ void __launch_bounds__(256) affine_act_kernel(const float* __restrict__ x, const float* __restrict__ res,
                                                         float* __restrict__ y, const float* __restrict__ scale,
                                                         const float* __restrict__ shift, int C, long S, int act,
                                                         float slope, int res_first) {
    const int plane = blockIdx.y;
    const int c = plane % C;
    const float s = scale ? scale[c] : 1.f, t = shift ? shift[c] : 0.f;
    const size_t off = (size_t)plane * S;
    const long n4 = ((S & 3) == 0 && ((off & 3) == 0)) ? S >> 2 : 0;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        f32x4 v = *(const f32x4*)(x + off + i * 4);
        v = v * s + t;
        f32x4 r = {0.f, 0.f, 0.f, 0.f};
        if (res) r = *(const f32x4*)(res + off + i * 4);
        if (res_first) v += r;
        v.x = act_apply(v.x, act, slope); v.y = act_apply(v.y, act, slope);
        v.z = act_apply(v.z, act, slope); v.w = act_apply(v.w, act, slope);
        if (!res_first) v += r;
        *(f32x4*)(y + off + i * 4) = v;
    }
    for (long i = n4 * 4 + (long)blockIdx.x * 256 + threadIdx.x; i < S; i += (long)gridDim.x * 256) {
        float v = x[off + i] * s + t;
        const float r = res ? res[off + i] : 0.f;
        if (res_first) v += r;
        v = act_apply(v, act, slope);
        if (!res_first) v += r;
        y[off + i] = v;
    }
}

// each thread produces 4 horizontally adjacent outputs of one (b, c) plane: a row of the window is loaded once
// ((4-1)*stride + K values) and reused by the 4 outputs; the K*K weights live in registers.
template <int K, int STRIDE>
__global__ void __launch_bounds__(256) dwconv2d_direct_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                              const float* __restrict__ scale, const float* __restrict__ shift,
                                                              float* __restrict__ y, int C, int H, int W, int Ho, int Wo,
                                                              int pad_t, int pad_l, int act, float* __restrict__ pool_part,
                                                              long xps) {
    constexpr int NX = 4, SPAN = (NX - 1) * STRIDE + K;
    __shared__ float wsum[4];
    const int plane = blockIdx.y;
    const int c = plane % C;
    const int wq = (Wo + NX - 1) / NX;
    const int item_raw = blockIdx.x * 256 + threadIdx.x;
    const bool live = item_raw < Ho * wq;
    if (!live && pool_part == nullptr) return;
    const int item = live ? item_raw : 0;                  // (dead lanes of the last block still join the reduction)
    const int oy = item / wq, ox0 = (item - oy * wq) * NX;
    const float* xp = x + (size_t)plane * xps;            // xps: floats between input planes (>= H * W: rows of a padded GEMM result)
    float wr[K * K];
#pragma unroll
    for (int i = 0; i < K * K; ++i) wr[i] = w[(size_t)c * K * K + i];
    float acc[NX] = {0.f, 0.f, 0.f, 0.f};
    const int ix0 = ox0 * STRIDE - pad_l;
    // Every load is unconditional (clamped address) and the padding is a bit mask on the loaded value: a load guarded by
    // a runtime condition makes hipcc branch around it and wait for each one separately, which serialised the K * SPAN
    // loads of a thread (the kernel ran at a quarter of what its bytes allow).
    int cix[SPAN];
    uint32_t cm[SPAN];
#pragma unroll
    for (int j = 0; j < SPAN; ++j) {
        const int ix = ix0 + j;
        cm[j] = 0u - (uint32_t)((unsigned)ix < (unsigned)W);
        cix[j] = min(max(ix, 0), W - 1);
    }
    float v[K][SPAN];
#pragma unroll
    for (int ky = 0; ky < K; ++ky) {
        const int iy = oy * STRIDE - pad_t + ky;
        const uint32_t rm = 0u - (uint32_t)((unsigned)iy < (unsigned)H);
        const float* row = xp + (size_t)min(max(iy, 0), H - 1) * W;
#pragma unroll
        for (int j = 0; j < SPAN; ++j) v[ky][j] = __uint_as_float(__float_as_uint(row[cix[j]]) & (rm & cm[j]));
    }
#pragma unroll
    for (int ky = 0; ky < K; ++ky)
#pragma unroll
        for (int o = 0; o < NX; ++o)
#pragma unroll
            for (int kx = 0; kx < K; ++kx) acc[o] += v[ky][o * STRIDE + kx] * wr[ky * K + kx];
    const float s = scale ? scale[c] : 1.f, t = shift ? shift[c] : 0.f;
    float* yp = y + ((size_t)plane * Ho + oy) * Wo + ox0;
    float part = 0.f;
#pragma unroll
    for (int o = 0; o < NX; ++o)
        if (live && ox0 + o < Wo) {
            const float v = act_apply(acc[o] * s + t, act, 0.f);
            yp[o] = v;
            part += v;
        }
    if (pool_part != nullptr) {
        // squeeze-excite pooling: this workgroup's share of sum_{y,x} y[b][c] in a FIXED order (wave tree, then 4
        // partials), one float per (plane, workgroup); occd_se_gate sums the partials in index order: deterministic
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) part += __shfl_down(part, off, 64);
        if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = part;
        __syncthreads();
        if (threadIdx.x == 0) pool_part[(size_t)plane * gridDim.x + blockIdx.x] = (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]);
    }
}

// The forward kernel of the product.  The direct kernel above reads its window with 4-byte loads whose lanes sit 16
// bytes apart: every load instruction touches 8 cache lines for 256 useful bytes and the K * SPAN loads of a thread
// re-touch the same lines, so the kernel is bound by the texture-address / L1 path at ~1/4 of what its bytes allow
// (measured: 41 us for 40 MB at the 1/16 level).  Here a workgroup first stages the input rows of its 256 items in LDS
// with fully coalesced loads (lane = consecutive float; zero padding written into the tile, so the compute phase has no
// bounds checks), then every thread reads its window rows as aligned 16-byte LDS vectors (column j of the tile is image
// column j - pad_l, so the window of output quad q starts at j = 4 q STRIDE).
template <int K, int STRIDE>
__global__ void __launch_bounds__(256) dwconv2d_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                       const float* __restrict__ scale, const float* __restrict__ shift,
                                                       float* __restrict__ y, int C, int H, int W, int Ho, int Wo,
                                                       int pad_t, int pad_l, int act, float* __restrict__ pool_part,
                                                       int wp, occd::FastDiv wpd, long xps) {
    constexpr int NX = 4, SPAN = (NX - 1) * STRIDE + K, SPAN4 = (SPAN + 3) / 4;
    extern __shared__ __attribute__((aligned(16))) float tile[];          // [rows][wp]
    __shared__ float wsum[4];
    const int plane = blockIdx.y;
    const int c = plane % C;
    const int wq = (Wo + NX - 1) / NX;
    const int nitems = Ho * wq;
    const int item0 = blockIdx.x * 256;
    const int oy_first = item0 / wq;
    const int oy_last = min(item0 + 255, nitems - 1) / wq;
    const int iy_first = oy_first * STRIDE - pad_t;
    const int nrows = (oy_last - oy_first) * STRIDE + K;
    const float* xp = x + (size_t)plane * xps;            // xps: floats between input planes (>= H * W: rows of a padded GEMM result)
    // ---- stage: unconditional clamped loads, padding as a bit mask on the loaded value
    const int total = nrows * wp;
    for (int e = threadIdx.x; e < total; e += 256) {
        const int r = (int)occd_fastdiv((uint32_t)e, wpd), j = e - r * wp;      // (one mul-hi instead of ~25 instructions)
        const int iy = iy_first + r, ix = j - pad_l;
        const uint32_t ok = 0u - (uint32_t)(((unsigned)iy < (unsigned)H) & ((unsigned)ix < (unsigned)W));
        const float v = xp[(size_t)min(max(iy, 0), H - 1) * W + min(max(ix, 0), W - 1)];
        tile[e] = __uint_as_float(__float_as_uint(v) & ok);
    }
    float wr[K * K];
#pragma unroll
    for (int i = 0; i < K * K; ++i) wr[i] = w[(size_t)c * K * K + i];
    __syncthreads();
    // ---- compute
    const int item_raw = item0 + threadIdx.x;
    const bool live = item_raw < nitems;
    const int item = live ? item_raw : item0;               // (dead lanes of the last block still join the reduction)
    const int oy = item / wq, q = item - oy * wq, ox0 = q * NX;
    const float* trow = tile + (size_t)((oy - oy_first) * STRIDE) * wp + q * NX * STRIDE;
    float acc[NX] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ky = 0; ky < K; ++ky) {
        float v[SPAN4 * 4];
#pragma unroll
        for (int i = 0; i < SPAN4; ++i) {
            const f32x4 t = *(const f32x4*)(trow + ky * wp + i * 4);
            v[i * 4 + 0] = t.x; v[i * 4 + 1] = t.y; v[i * 4 + 2] = t.z; v[i * 4 + 3] = t.w;
        }
#pragma unroll
        for (int o = 0; o < NX; ++o)
#pragma unroll
            for (int kx = 0; kx < K; ++kx) acc[o] += v[o * STRIDE + kx] * wr[ky * K + kx];
    }
    const float s = scale ? scale[c] : 1.f, t = shift ? shift[c] : 0.f;
    float* yp = y + ((size_t)plane * Ho + oy) * Wo + ox0;
    float part = 0.f;
#pragma unroll
    for (int o = 0; o < NX; ++o)
        if (live && ox0 + o < Wo) {
            const float v = act_apply(acc[o] * s + t, act, 0.f);
            yp[o] = v;
            part += v;
        }
    if (pool_part != nullptr) {
        // squeeze-excite pooling: this workgroup's share of sum_{y,x} y[b][c] in a FIXED order (wave tree, then 4
        // partials), one float per (plane, workgroup); occd_se_gate sums the partials in index order: deterministic
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) part += __shfl_down(part, off, 64);
        if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = part;
        __syncthreads();
        if (threadIdx.x == 0) pool_part[(size_t)plane * gridDim.x + blockIdx.x] = (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]);
    }
}

// ---- depthwise convolution, backward (MIOpen falls back to naive kernels for fp32 depthwise
// training: 17.5 ms per config-2 step for forward + data + weight gradients).
// data gradient: dx[iy][ix] = sum_{ky,kx} gy[(iy + pad_t - ky) / s][(ix + pad_l - kx) / s] * w[ky][kx] over the taps whose
// source index is divisible by the stride; each thread produces 4 adjacent input pixels.
template <int K, int STRIDE>
__global__ void __launch_bounds__(256) dwconv2d_bwd_data_kernel(const float* __restrict__ gy, const float* __restrict__ w,
                                                                float* __restrict__ dx, int C, int H, int W, int Ho,
                                                                int Wo, int pad_t, int pad_l) {
    constexpr int NX = 4;
    const int plane = blockIdx.y;
    const int c = plane % C;
    const int wq = (W + NX - 1) / NX;
    const int item = blockIdx.x * 256 + threadIdx.x;
    if (item >= H * wq) return;
    const int iy = item / wq, ix0 = (item - iy * wq) * NX;
    const float* gp = gy + (size_t)plane * Ho * Wo;
    float wr[K * K];
#pragma unroll
    for (int i = 0; i < K * K; ++i) wr[i] = w[(size_t)c * K * K + i];
    float acc[NX] = {0.f, 0.f, 0.f, 0.f};
    if constexpr (STRIDE == 1) {
        // correlation with the flipped kernel: output o, tap kx reads gy column ix0 + o + pad_l - kx = j0 + (o + K - 1 - kx);
        // one span of NX + K - 1 values per row, all loads unconditional (clamped address, bit-mask zero fill)
        constexpr int SP = NX + K - 1;
        const int j0 = ix0 + pad_l - (K - 1);
        int cj[SP];
        uint32_t cm[SP];
#pragma unroll
        for (int j = 0; j < SP; ++j) {
            cm[j] = 0u - (uint32_t)((unsigned)(j0 + j) < (unsigned)Wo);
            cj[j] = min(max(j0 + j, 0), Wo - 1);
        }
        float v[K][SP];
#pragma unroll
        for (int ky = 0; ky < K; ++ky) {
            const int ty = iy + pad_t - ky;
            const uint32_t rm = 0u - (uint32_t)((unsigned)ty < (unsigned)Ho);
            const float* row = gp + (size_t)min(max(ty, 0), Ho - 1) * Wo;
#pragma unroll
            for (int j = 0; j < SP; ++j) v[ky][j] = __uint_as_float(__float_as_uint(row[cj[j]]) & (rm & cm[j]));
        }
#pragma unroll
        for (int ky = 0; ky < K; ++ky)
#pragma unroll
            for (int o = 0; o < NX; ++o)
#pragma unroll
                for (int kx = 0; kx < K; ++kx) acc[o] += v[ky][o + K - 1 - kx] * wr[ky * K + kx];
    } else {
#pragma unroll
        for (int ky = 0; ky < K; ++ky) {
            const int ty = iy + pad_t - ky;
            if (ty < 0 || ty % STRIDE != 0 || ty / STRIDE >= Ho) continue;
            const float* row = gp + (size_t)(ty / STRIDE) * Wo;
#pragma unroll
            for (int o = 0; o < NX; ++o)
#pragma unroll
                for (int kx = 0; kx < K; ++kx) {
                    const int tx = ix0 + o + pad_l - kx;
                    if (tx >= 0 && tx % STRIDE == 0 && tx / STRIDE < Wo) acc[o] += row[tx / STRIDE] * wr[ky * K + kx];
                }
        }
    }
    float* xp = dx + ((size_t)plane * H + iy) * W + ix0;
#pragma unroll
    for (int o = 0; o < NX; ++o)
        if (ix0 + o < W) xp[o] = acc[o];
}

// weight gradient: dw[c][ky][kx] = sum_{b,oy,ox} gy[b][c][oy][ox] * x[b][c][oy*s - pad_t + ky][ox*s - pad_l + kx].
// grid (chunks, C): a workgroup walks its share of the (b, oy, ox) positions of one channel with the K*K partial sums
// in registers, reduces them over the workgroup in a fixed order and writes one partial record; the second kernel sums
// the records in index order (deterministic, no float atomics).
template <int K, int STRIDE>
__global__ void __launch_bounds__(256) dwconv2d_bwd_weight_kernel(const float* __restrict__ x, const float* __restrict__ gy,
                                                                  float* __restrict__ part, int B, int C, int H, int W,
                                                                  int Ho, int Wo, int pad_t, int pad_l) {
    __shared__ float red[4][K * K];
    constexpr int NX = 4, SPAN = (NX - 1) * STRIDE + K;
    const int c = blockIdx.y;
    const int wq = (Wo + NX - 1) / NX;
    const long total = (long)B * Ho * wq;                    // items of NX horizontally adjacent outputs
    float acc[K * K];
#pragma unroll
    for (int i = 0; i < K * K; ++i) acc[i] = 0.f;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int b = (int)(i / ((long)Ho * wq));
        const int r = (int)(i - (long)b * Ho * wq);
        const int oy = r / wq, ox0 = (r - oy * wq) * NX;
        const float* gp = gy + (((size_t)b * C + c) * Ho + oy) * Wo;
        const float* xp = x + ((size_t)b * C + c) * H * W;
        // all loads unconditional (clamped address), padding / ragged tail as bit masks on the loaded values
        float g[NX];
#pragma unroll
        for (int o = 0; o < NX; ++o)
            g[o] = __uint_as_float(__float_as_uint(gp[min(ox0 + o, Wo - 1)]) & (0u - (uint32_t)(ox0 + o < Wo)));
        const int ix0 = ox0 * STRIDE - pad_l;
        int cix[SPAN];
        uint32_t cm[SPAN];
#pragma unroll
        for (int j = 0; j < SPAN; ++j) {
            cm[j] = 0u - (uint32_t)((unsigned)(ix0 + j) < (unsigned)W);
            cix[j] = min(max(ix0 + j, 0), W - 1);
        }
        float v[K][SPAN];
#pragma unroll
        for (int ky = 0; ky < K; ++ky) {
            const int iy = oy * STRIDE - pad_t + ky;
            const uint32_t rm = 0u - (uint32_t)((unsigned)iy < (unsigned)H);
            const float* row = xp + (size_t)min(max(iy, 0), H - 1) * W;
#pragma unroll
            for (int j = 0; j < SPAN; ++j) v[ky][j] = __uint_as_float(__float_as_uint(row[cix[j]]) & (rm & cm[j]));
        }
#pragma unroll
        for (int ky = 0; ky < K; ++ky)
#pragma unroll
            for (int kx = 0; kx < K; ++kx)
#pragma unroll
                for (int o = 0; o < NX; ++o) acc[ky * K + kx] += g[o] * v[ky][o * STRIDE + kx];
    }
#pragma unroll
    for (int i = 0; i < K * K; ++i) {
        float v = acc[i];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][i] = v;
    }
    __syncthreads();
    if (threadIdx.x < K * K)
        part[((size_t)c * gridDim.x + blockIdx.x) * (K * K) + threadIdx.x] =
            (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

__global__ void dwconv2d_bwd_weight_reduce_kernel(const float* __restrict__ part, float* __restrict__ dw, int C, int KK,
                                                  int chunks) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= C * KK) return;
    const int c = i / KK, t = i - c * KK;
    float s = 0.f;
    for (int j = 0; j < chunks; ++j) s += part[((size_t)c * chunks + j) * KK + t];
    dw[i] = s;
}

// bilinear upsampling (align_corners=True) of x (B, C, h, w) to (H, W) + concat with skip (B, Cs, H, W): each thread
// writes FOUR horizontally adjacent output pixels of one plane with one 16-byte store (the first version stored 4 bytes
// per lane: 2.1 TB/s on a pass that is almost pure writing); the vertical weights are shared by the four.
__global__ void __launch_bounds__(256) upsample_cat_kernel(const float* __restrict__ x, const float* __restrict__ skip,
                                                           float* __restrict__ out, int C, int Cs, int h, int w, int H,
                                                           int W, float rh, float rw) {
    const int plane = blockIdx.z;                 // b * (C + Cs) + channel
    const int ct = C + Cs;
    const int b = plane / ct, c = plane - b * ct;
    const int ox0 = (blockIdx.x * 64 + (threadIdx.x & 63)) * 4;
    const int oy = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (ox0 >= W || oy >= H) return;
    float v[4];
    if (c >= C) {
        const float* sp = skip + (((size_t)b * Cs + (c - C)) * H + oy) * W;
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = sp[min(ox0 + i, W - 1)];
    } else {
        // torch upsample_bilinear2d, align_corners=True: src = dst * (in - 1) / (out - 1)
        const float sy = rh * oy;
        const int y0 = (int)sy;
        const int y1 = y0 + (y0 < h - 1);
        const float ly = sy - y0, hy = 1.f - ly;
        const float* p0 = x + (((size_t)b * C + c) * h + y0) * w;
        const float* p1 = x + (((size_t)b * C + c) * h + y1) * w;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float sx = rw * min(ox0 + i, W - 1);
            const int x0 = (int)sx;
            const int x1 = x0 + (x0 < w - 1);
            const float lx = sx - x0, hx = 1.f - lx;
            v[i] = hy * (hx * p0[x0] + lx * p0[x1]) + ly * (hx * p1[x0] + lx * p1[x1]);
        }
    }
    float* op = out + ((size_t)plane * H + oy) * W + ox0;
    if (ox0 + 3 < W && ((reinterpret_cast<uintptr_t>(op) & 15) == 0)) {
        *(f32x4*)op = f32x4{v[0], v[1], v[2], v[3]};
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (ox0 + i < W) op[i] = v[i];
    }
}

// conv3x3(pad 1) applied to a bilinearly upsampled map, without ever forming the upsampled map (the
// first convolution of every decoder level, occdepth/models/unet2d.py:24-46: conv(cat[up(x), skip])).  Upsampling, the
// one-pixel tap shifts and the channel mixing are all linear and the channel mixing commutes with the two spatial ones:
//     conv(up(x))[co][p] = sum_t (up(z_t))[co][p + d_t],     z_t = W_t . x   (t = ky * 3 + kx, d_t = (ky - 1, kx - 1)),
// with up(z_t) := 0 outside the output grid (the convolution's zero padding).  The nine W_t . x are ONE pointwise GEMM at
// the LOW resolution (Cup -> 9 Cout, a quarter of the pixels: 9 / 16 of the multiplies of the Winograd form at the high
// resolution, and no (Cup + Cskip)-channel high-resolution tensor is written or re-read); this kernel is the rest:
//     out[b][co][oy][ox] = sum_{ky, kx} [inside(oy + ky - 1, ox + kx - 1)] bilinear(z[b][(ky * 3 + kx) * Cout + co], that pixel)
// with exactly upsample_cat_kernel's (= ATen's align_corners=True) source index arithmetic.  A thread owns 4 adjacent ox.
__global__ void __launch_bounds__(256) upconv_gather_direct_kernel(const float* __restrict__ z, float* __restrict__ out, int Cout,
                                                            int h, int w, int H, int W, float rh, float rw, long zcs,
                                                            long zbs) {
    const int plane = blockIdx.z;                 // b * Cout + co
    const int b = plane / Cout, co = plane - b * Cout;
    const int ox0 = (blockIdx.x * 64 + (threadIdx.x & 63)) * 4;
    const int oy = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (ox0 >= W || oy >= H) return;
    // the 6 columns ox0 - 1 .. ox0 + 4 the 3 horizontal taps of the 4 outputs look at
    int x0[6], x1[6];
    float lx[6], cm[6];
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        const int cx = ox0 - 1 + j;
        cm[j] = (unsigned)cx < (unsigned)W ? 1.f : 0.f;
        const float sx = rw * min(max(cx, 0), W - 1);
        x0[j] = (int)sx;
        x1[j] = x0[j] + (x0[j] < w - 1);
        lx[j] = sx - x0[j];
    }
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    const size_t tap_stride = (size_t)Cout * zcs;            // z[b][ch][y][x] at b * zbs + ch * zcs + y * w + x
    const float* zb = z + (size_t)b * zbs + (size_t)co * zcs;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
        const int ry = oy + ky - 1;
        if ((unsigned)ry >= (unsigned)H) continue;                   // (uniform over the wave: one output row per wave)
        const float sy = rh * ry;
        const int y0 = (int)sy;
        const int y1 = y0 + (y0 < h - 1);
        const float ly = sy - y0, hy = 1.f - ly;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const float* p0 = zb + (size_t)(ky * 3 + kx) * tap_stride + (size_t)y0 * w;
            const float* p1 = zb + (size_t)(ky * 3 + kx) * tap_stride + (size_t)y1 * w;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int j = i + kx;
                const float hx = 1.f - lx[j];
                const float v = hy * (hx * p0[x0[j]] + lx[j] * p0[x1[j]]) + ly * (hx * p1[x0[j]] + lx[j] * p1[x1[j]]);
                acc[i] += cm[j] * v;
            }
        }
    }
    float* op = out + ((size_t)plane * H + oy) * W + ox0;
    if (ox0 + 3 < W && ((reinterpret_cast<uintptr_t>(op) & 15) == 0)) {
        *(f32x4*)op = f32x4{acc[0], acc[1], acc[2], acc[3]};
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (ox0 + i < W) op[i] = acc[i];
    }
}

// Staged form of the kernel above (the product path).  The direct form issues 144 four-byte gathers per thread (9 taps x
// 4 outputs x 4 bilinear corners) and is bound by the texture-address path (0.58 ms per decoder level).  Here the
// workgroup (4 output rows x 256 output columns of one (b, co) plane) first builds, for every tap plane t and each of its
// 4 output rows, the VERTICALLY interpolated low-resolution row segment its
// columns can touch -- coalesced loads along the row, 2 per element, zero for rows outside the output grid -- in LDS:
// L[t][r][c] = hy z_t[y0][xlo + c] + ly z_t[y1][xlo + c] for the output row oy0 + r + ky - 1.  A thread then needs 2 LDS
// reads per (tap, output): 72 instead of 144 global ones.  Same arithmetic as the direct form up to the order of the two
// interpolations (vertical first here).
// Round 4: wave r stages exactly the 9 segments of ITS output row r (lane = column, 3 column chunks) through buffer loads
// (descriptor + scalar (tap, row) offset + a loop-invariant column offset: no 64-bit address registers), so all four waves
// load and every load of the tile is independent of the others: 54 loads in flight per lane instead of three dependent
// batches of 24 on 130 of the 256 threads -- the kernel was latency bound at 1.9 TB/s.  The 1/1 level with the fused skip
// convolution: 0.81 -> 0.55 ms.  (Measured and dropped: strips of 2 / 4 / 8 row groups per workgroup with the next
// group's loads issued before the arithmetic of the current one -- 184 VGPRs, 2 workgroups per CU, 0.64 ms.)
constexpr int kUpNC = 192;                                   // LDS row length: low-resolution columns a workgroup can touch
// SKIP: the convolution over the (few: <= kUpSkipC) skip channels, the BatchNorm shift and the LeakyReLU are applied here
// too, i.e. the kernel emits the finished first convolution of the level (the 1/1 level: 3 raw image channels).  K10 is
// the wrong tool for K = 3: its per-workgroup prologue / exchange / epilogue is fixed and it ran the 3 -> 80 convolution
// at 0.76 ms against the 0.2 ms its bytes cost, plus a write + re-read of the gathered tensor (2 x 289 MB).
struct UpSkipP {
    const float* skip;      // (B, Cs, H, W)
    const float* wskip;     // (Cout, Cs, 3, 3), BatchNorm scale folded in
    const float* shift;     // (Cout)
    int Cs;
    float slope;
};
constexpr int kUpSkipC = 4, kUpSkipW = 260;                  // skip tile: [Cs][6 rows][258 columns (+2 pad)]

// the tile's global loads
template <bool SKIP>
struct UpStage {
    float a0[9][3], a1[9][3];      // [tap][column chunk]: the two source rows of this wave's output row
    float ly[3], ok[3];            // per ky (wave-uniform)
    float s[SKIP ? kUpSkipC * 6 : 1];
    float sx;                      // the two extra tile columns (256, 257): one (channel, row, column) element on threads < 12 Cs
};

// A wave-uniform value that the vector ALU produced, as an SGPR.  A bare __builtin_amdgcn_readfirstlane is commuted by the
// optimiser to the operand side of the float -> int conversion (whose result is a VGPR again, and buffer descriptors /
// scalar offsets held in VGPRs are legalised with waterfall loops); the empty asm makes the converted value opaque first.
// (Not the instruction itself as inline asm: that form faulted on gfx950 -- the compiler cannot see a hazard inside asm.)
__device__ __forceinline__ int to_sgpr(int v) {
    asm volatile("" : "+v"(v));
    return __builtin_amdgcn_readfirstlane(v);
}

// COOP (round 6): the four waves of a workgroup share their source rows.  Output rows oy0 .. oy0 + 3 at vertical tap ky touch
// at most FOUR consecutive low-resolution rows (base(ky) .. base(ky) + 3 for any upsampling ratio >= 2: rh <= 0.5), and the
// non-COOP form loads EIGHT (every wave its own y0 / y1 pair): 216 instead of 108 row-segment loads per workgroup through the
// L1 / texture path, which -- not HBM (1.7 TB/s, traffic = algorithmic) -- is what bounds the kernel.  Here wave r loads row
// base(ky) + r of the three taps of every ky RAW into LDS (27 loads per lane instead of 54), and the vertical interpolation
// moves to the read side (four LDS reads per tap and output instead of two).
// MEASURED AND NOT ADOPTED (same box, four alternating bench runs): K12 family 1.040 / 1.039 ms per frame with COOP against
// 0.970 / 0.971 without (116 / 132 VGPRs against 106 / 95: one wave per SIMD fewer), frame 17.89 / 17.91 against 17.84 / 17.84 ms
// -- halving the loads through L1 does not help, so the load path is not what bounds the kernel either; what is left is the
// latency of one dependent load -> barrier -> compute -> store chain per workgroup at 4 workgroups per CU.  Opt-in:
// OCCD_UPCONV_COOP=1 (bit-identical results: tests/test_winograd2d.py runs both).
template <bool SKIP, bool COOP = false>
__global__ void __launch_bounds__(256) upconv_gather_kernel(const float* __restrict__ z, float* __restrict__ out, int Cout,
                                                            int h, int w, int H, int W, float rh, float rw, long zcs,
                                                            long zbs, const UpSkipP sk) {
    __shared__ float L[9 * 4 * kUpNC];
    __shared__ __attribute__((aligned(16))) float S[SKIP ? kUpSkipC * 6 * kUpSkipW : 4];
    // Workgroup -> tile mapping.  Vertically adjacent tiles share 2 of their ~4 low-resolution source rows; the
    // dispatcher deals consecutive workgroup ids round-robin to the 8 XCDs (each with its own L2), so in launch order
    // those rows came back from HBM twice (PMC: 2.2x the z tensor fetched).  XCD-aware bijective remap of the linear id,
    // then column-major tiles inside a plane: every XCD walks a contiguous run of vertically adjacent tiles.
    const uint32_t gx = gridDim.x, gy = gridDim.y, per_plane = gx * gy;
    uint32_t bid = blockIdx.x + gx * (blockIdx.y + gy * blockIdx.z);
    {
        const uint32_t nwg = per_plane * gridDim.z, q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int plane = bid / per_plane;            // b * Cout + co
    const uint32_t local = bid - plane * per_plane;
    const int bx = local / gy, by = local - bx * gy;
    const int b = plane / Cout, co = plane - b * Cout;
    const int X0 = bx * 256, oy0 = by * 4;
    // low-resolution column window of the hi-res columns X0 - 1 .. X0 + 256 (clamped): [xlo, xlo + nc)
    // (float-derived integers come out of the vector ALU; to_sgpr moves them, and with them the buffer descriptor and the
    //  scalar offsets built from them, into SGPRs -- otherwise every buffer load is wrapped in a waterfall loop)
    const int xlo = to_sgpr((int)(rw * max(X0 - 1, 0)));
    const int xe = to_sgpr((int)(rw * min(X0 + 256, W - 1)));
    const int nc = min(xe + 1, w - 1) - xlo + 1;              // (the host guarantees nc <= kUpNC)
    const size_t tap_stride = (size_t)Cout * zcs;            // z[b][ch][y][x] at b * zbs + ch * zcs + y * w + x
    const float* zb = z + (size_t)b * zbs + (size_t)co * zcs + xlo;
    const int lane = threadIdx.x & 63;
    const int r = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // output row of this wave inside a group
    const bool third = lane + 128 < nc;                       // (2x upsampling: nc = 131, three lanes)
    const int cA = min(lane, nc - 1), cB = min(lane + 64, nc - 1), cC = min(lane + 128, nc - 1);
    const float* sb = SKIP ? sk.skip + (size_t)b * sk.Cs * H * W : nullptr;
    const int cx0 = min(max(X0 - 1 + (int)threadIdx.x, 0), W - 1);
    const bool okx0 = (unsigned)(X0 - 1 + (int)threadIdx.x) < (unsigned)W;
    // (extra columns 256 / 257 of the skip tile: thread e -> channel e / 12, row (e % 12) / 2, column 256 + (e & 1))
    const int ec = threadIdx.x / 12, er = (threadIdx.x % 12) >> 1, ecol = 256 + (threadIdx.x & 1);
    const bool extra = SKIP && (int)threadIdx.x < 12 * sk.Cs;

    // buffer loads: one descriptor per tensor (SGPRs), the (tap, source row) part of the address as the scalar offset and the
    // column as a loop-invariant 32-bit vector offset -- no 64-bit address registers for the 54 + 19 loads of a row group
    const auto zr = __builtin_amdgcn_make_buffer_rsrc((void*)zb, 0, 0x7fffffff, 0x00020000);
    const auto sr = __builtin_amdgcn_make_buffer_rsrc((void*)(SKIP ? sb : z), 0, 0x7fffffff, 0x00020000);
    const unsigned vA = (unsigned)cA * 4u, vB = (unsigned)cB * 4u, vC = (unsigned)cC * 4u, vS = (unsigned)cx0 * 4u;
    const unsigned tap_bytes = (unsigned)tap_stride * 4u;
    auto ldz = [&](unsigned voff, unsigned soff) {
        return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(zr, voff, soff, 0));
    };
    auto issue = [&](UpStage<SKIP>& st) {
        const int oy = oy0 + r;
        unsigned o0[3], o1[3];
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int ry = oy + ky - 1;
            const bool ok = (unsigned)ry < (unsigned)H;
            const float sy = rh * (ok ? ry : 0);
            const int y0 = to_sgpr((int)sy);
            const int y1 = y0 + (y0 < h - 1);
            st.ly[ky] = sy - y0;                               // (may round a hair below 0 when contracted into an fma:
            st.ok[ky] = ok ? 1.f : 0.f;                        //  never use its sign as the "outside the grid" marker)
            o0[ky] = (unsigned)(y0 * w) * 4u;
            o1[ky] = (unsigned)(y1 * w) * 4u;
        }
        if (COOP) {
            // this wave's share of the group's source rows: row base(ky) + r (clamped) of the three taps of every ky
            unsigned oc[3];
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const int ryf = min(max(oy0 + ky - 1, 0), H - 1);
                const int base = to_sgpr((int)(rh * ryf));
                oc[ky] = (unsigned)(min(base + r, h - 1) * w) * 4u;
            }
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const unsigned s0 = (unsigned)t * tap_bytes + oc[t / 3];
                st.a0[t][0] = ldz(vA, s0);
                st.a0[t][1] = ldz(vB, s0);
            }
            if (third) {
#pragma unroll
                for (int t = 0; t < 9; ++t) st.a0[t][2] = ldz(vC, (unsigned)t * tap_bytes + oc[t / 3]);
            }
        } else {
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const unsigned s0 = (unsigned)t * tap_bytes + o0[t / 3], s1 = (unsigned)t * tap_bytes + o1[t / 3];
            st.a0[t][0] = ldz(vA, s0);
            st.a1[t][0] = ldz(vA, s1);
            st.a0[t][1] = ldz(vB, s0);
            st.a1[t][1] = ldz(vB, s1);
        }
        if (third) {
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                st.a0[t][2] = ldz(vC, (unsigned)t * tap_bytes + o0[t / 3]);
                st.a1[t][2] = ldz(vC, (unsigned)t * tap_bytes + o1[t / 3]);
            }
        }
        }
        if (SKIP) {
            // skip tile: rows oy0 - 1 .. oy0 + 4, columns X0 - 1 .. X0 + 256 of every skip channel, zero outside the image
            // (thread = tile column, channels x rows unrolled: no index divisions, all loads of a thread independent)
#pragma unroll
            for (int c = 0; c < kUpSkipC; ++c) {
                if (c < sk.Cs) {                                                 // (uniform)
#pragma unroll
                    for (int rr = 0; rr < 6; ++rr) {
                        const int iy = min(max(oy0 - 1 + rr, 0), H - 1);
                        st.s[SKIP ? c * 6 + rr : 0] = __builtin_bit_cast(
                            float, __builtin_amdgcn_raw_buffer_load_b32(sr, vS, (unsigned)((c * H + iy) * W) * 4u, 0));
                    }
                }
            }
            if (extra) {
                const int iy = min(max(oy0 - 1 + er, 0), H - 1), ix = min(X0 - 1 + ecol, W - 1);
                st.sx = sb[((size_t)ec * H + iy) * W + ix];
            }
        }
    };
    auto commit = [&](const UpStage<SKIP>& st) {
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const float ly = st.ly[t / 3], ok = st.ok[t / 3];
            float* row = L + (t * 4 + r) * kUpNC + lane;
            if (COOP) {                                        // raw source row base(ky) + r; interpolated by the readers
                row[0] = st.a0[t][0];
                row[64] = st.a0[t][1];
                row[128] = st.a0[t][2];
                continue;
            }
            row[0] = ok * ((1.f - ly) * st.a0[t][0] + ly * st.a1[t][0]);
            row[64] = ok * ((1.f - ly) * st.a0[t][1] + ly * st.a1[t][1]);
            row[128] = ok * ((1.f - ly) * st.a0[t][2] + ly * st.a1[t][2]);   // (columns >= nc: never read)
        }
        if (SKIP) {
#pragma unroll
            for (int c = 0; c < kUpSkipC; ++c) {
                if (c < sk.Cs) {
#pragma unroll
                    for (int rr = 0; rr < 6; ++rr) {
                        const bool oky = (unsigned)(oy0 - 1 + rr) < (unsigned)H;
                        S[(c * 6 + rr) * kUpSkipW + threadIdx.x] = oky && okx0 ? st.s[SKIP ? c * 6 + rr : 0] : 0.f;
                    }
                }
            }
            if (extra) {
                const bool oky = (unsigned)(oy0 - 1 + er) < (unsigned)H, okx = X0 - 1 + ecol < W;
                S[(ec * 6 + er) * kUpSkipW + ecol] = oky && okx ? st.sx : 0.f;
            }
        }
    };

    // per-thread column arithmetic of the 4 outputs
    const int ox0 = X0 + lane * 4;
    int x0[6], x1[6];
    float lx[6], cm[6];
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        const int cx = ox0 - 1 + j;
        cm[j] = (unsigned)cx < (unsigned)W ? 1.f : 0.f;
        const float sx = rw * min(max(cx, 0), W - 1);
        const int xa = (int)sx;
        x0[j] = xa - xlo;
        x1[j] = xa + (xa < w - 1) - xlo;
        lx[j] = sx - xa;
    }
    const float* wk = SKIP ? sk.wskip + (size_t)co * sk.Cs * 9 : nullptr;     // (uniform: scalar loads at the point of use)
    const float sh = SKIP ? sk.shift[co] : 0.f;

    UpStage<SKIP> st;
#pragma unroll
    for (int t = 0; t < 9; ++t) st.a0[t][2] = st.a1[t][2] = 0.f;  // (only lanes with a third column load them)
    st.sx = 0.f;
    issue(st);
    commit(st);
    __syncthreads();
    const int oy = oy0 + r;
    if (ox0 >= W || oy >= H) return;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    if (COOP) {
        // vertical part of this output row per ky: rows y0 / y1 relative to the group's base(ky) (the same arithmetic as the
        // loader's), weights (1 - ly, ly) * ok
        int ra[3], rb[3];
        float wa[3], wb[3];
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int ry = oy + ky - 1;
            const bool ok = (unsigned)ry < (unsigned)H;
            const float sy = rh * (ok ? ry : 0);
            const int y0 = (int)sy;
            const int y1 = y0 + (y0 < h - 1);
            const float ly = sy - y0;
            const int base = (int)(rh * min(max(oy0 + ky - 1, 0), H - 1));
            ra[ky] = min(max(min(y0, h - 1) - base, 0), 3);
            rb[ky] = min(max(min(y1, h - 1) - base, 0), 3);
            // (rows base + r are clamped to h - 1 by the loader: relative index of a clamped row = its distance, capped at 3)
            wa[ky] = ok ? 1.f - ly : 0.f;
            wb[ky] = ok ? ly : 0.f;
        }
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int ky = t / 3, kx = t % 3;
            const float* rowa = L + (t * 4 + ra[ky]) * kUpNC;
            const float* rowb = L + (t * 4 + rb[ky]) * kUpNC;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int j = i + kx;
                const float v0 = wa[ky] * rowa[x0[j]] + wb[ky] * rowb[x0[j]];
                const float v1 = wa[ky] * rowa[x1[j]] + wb[ky] * rowb[x1[j]];
                acc[i] += cm[j] * ((1.f - lx[j]) * v0 + lx[j] * v1);
            }
        }
    } else {
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const float* row = L + (t * 4 + r) * kUpNC;
        const int kx = t % 3;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int j = i + kx;
            acc[i] += cm[j] * ((1.f - lx[j]) * row[x0[j]] + lx[j] * row[x1[j]]);
        }
    }
    }
    if (SKIP) {
        const int lc = ox0 - X0;                              // tile column of output ox0 - 1 is lc (tile starts at X0 - 1)
#pragma unroll
        for (int c = 0; c < kUpSkipC; ++c) {
            if (c >= sk.Cs) break;                                           // (uniform)
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const float* srow = S + (c * 6 + r + ky) * kUpSkipW + lc;
                // (tile rows are 1040 bytes and lc = 4 * lane: one aligned 16-byte + one 8-byte LDS read per row)
                const f32x4 va = *(const f32x4*)srow;
                const f32x2 vb = *(const f32x2*)(srow + 4);
                const float v[6] = {va.x, va.y, va.z, va.w, vb.x, vb.y};
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const float wv = wk[(c * 3 + ky) * 3 + kx];
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc[i] += v[i + kx] * wv;
                }
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float t = acc[i] + sh;
            acc[i] = t > 0.f ? t : t * sk.slope;
        }
    }
    float* op = out + ((size_t)plane * H + oy) * W + ox0;
    if (ox0 + 3 < W && ((reinterpret_cast<uintptr_t>(op) & 15) == 0)) {
        *(f32x4*)op = f32x4{acc[0], acc[1], acc[2], acc[3]};
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (ox0 + i < W) op[i] = acc[i];
    }
}

}  // namespace

// softmax over the channel axis of an NCHW map: thread = pixel, channels strided by the plane (coalesced across the
// wave); three passes over C values that stay in L2.  Replaces ATen's SpatialSoftMax on the depth-bin logits
// (75 us -> a few us for (2, 104, 47, 153)).
__global__ void __launch_bounds__(256) softmax_nchw_kernel(const float* __restrict__ x, float* __restrict__ y, int C,
                                                           long S) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= S) return;
    const float* xp = x + (size_t)blockIdx.y * C * S + i;
    float* yp = y + (size_t)blockIdx.y * C * S + i;
    float mx = -INFINITY;
    for (int c = 0; c < C; ++c) mx = fmaxf(mx, xp[(size_t)c * S]);
    float sum = 0.f;
    for (int c = 0; c < C; ++c) sum += expf(xp[(size_t)c * S] - mx);
    const float inv = 1.f / sum;
    for (int c = 0; c < C; ++c) yp[(size_t)c * S] = expf(xp[(size_t)c * S] - mx) * inv;
}

// C <= 128 (the 104 depth bins): 64 pixels per workgroup, its 4 waves split the channels (wave w takes c = w, w + 4, ...:
// <= 32 values per thread, all loads independent and in registers), maxima and sums meet in LDS.  The kernel above runs
// the 3 x C loads of a pixel back to back in one thread on 58 workgroups: 64 us for 6 MB, pure latency.
__global__ void __launch_bounds__(256) softmax_nchw_c128_kernel(const float* __restrict__ x, float* __restrict__ y, int C,
                                                                long S) {
    __shared__ float red[2][4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long i = (long)blockIdx.x * 64 + lane;
    const bool ok = i < S;
    const float* xp = x + (size_t)blockIdx.y * C * S + (ok ? i : S - 1);
    float v[32];
    float mx = -INFINITY;
#pragma unroll
    for (int k = 0; k < 32; ++k) {
        const int c = wave + 4 * k;
        v[k] = c < C ? xp[(size_t)c * S] : -INFINITY;
        mx = fmaxf(mx, v[k]);
    }
    red[0][wave][lane] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0][0][lane], red[0][1][lane]), fmaxf(red[0][2][lane], red[0][3][lane]));
    float sum = 0.f;
#pragma unroll
    for (int k = 0; k < 32; ++k) {
        v[k] = expf(v[k] - mx);                         // (exp(-inf) = 0 for the channels past C)
        sum += v[k];
    }
    red[1][wave][lane] = sum;
    __syncthreads();
    const float inv = 1.f / ((red[1][0][lane] + red[1][1][lane]) + (red[1][2][lane] + red[1][3][lane]));
    if (!ok) return;
    float* yp = y + (size_t)blockIdx.y * C * S + i;
#pragma unroll
    for (int k = 0; k < 32; ++k) {
        const int c = wave + 4 * k;
        if (c < C) yp[(size_t)c * S] = v[k] * inv;
    }
}

extern "C" int occd_softmax_nchw(const float* x, float* y, int32_t batch, int32_t C, int64_t S, void* stream) {
    if (!x || !y || batch <= 0 || batch > 65535 || C <= 0 || S <= 0) return OCCD_EINVAL;
    occd::ProfScope prof("softmax_nchw", (hipStream_t)stream, 0.0, 8.0 * batch * C * (double)S);
    if (C <= 128) {
        hipLaunchKernelGGL(softmax_nchw_c128_kernel, dim3((unsigned)((S + 63) / 64), (unsigned)batch), dim3(256), 0,
                           (hipStream_t)stream, x, y, C, (long)S);
        return occd::check_launch();
    }
    hipLaunchKernelGGL(softmax_nchw_kernel, dim3((unsigned)((S + 255) / 256), (unsigned)batch), dim3(256), 0,
                       (hipStream_t)stream, x, y, C, (long)S);
    return occd::check_launch();
}

// d/dx [x sigmoid(x)] = s (1 + x (1 - s)): the backward of the EfficientNet swish in ONE pass.  The
// autograd graph of `x * torch.sigmoid(x)` is sigmoid + mul forward and sigmoid_backward + 2 mul + add backward: six
// launches and ten tensor passes per swish site, ~330 sites per training step.
__global__ void __launch_bounds__(256) swish_bwd_kernel(const float* __restrict__ x, const float* __restrict__ gy,
                                                        float* __restrict__ gx, long n) {
    const long n4 = n >> 2;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const f32x4 xv = ((const f32x4*)x)[i], g = ((const f32x4*)gy)[i];
        f32x4 o;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float sg = __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(xv[k] * -1.4426950408889634f));
            o[k] = g[k] * sg * (1.f + xv[k] * (1.f - sg));
        }
        ((f32x4*)gx)[i] = o;
    }
    for (long i = n4 * 4 + (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const float sg = __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(x[i] * -1.4426950408889634f));
        gx[i] = gy[i] * sg * (1.f + x[i] * (1.f - sg));
    }
}

extern "C" int occd_swish_bwd(const float* x, const float* gy, float* gx, int64_t n, void* stream) {
    if (!x || !gy || !gx || n <= 0) return OCCD_EINVAL;
    if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(gy) | reinterpret_cast<uintptr_t>(gx)) & 15) return OCCD_EINVAL;
    long blocks = (n / 4 + 255) / 256;
    if (blocks < 1) blocks = 1;
    if (blocks > 8192) blocks = 8192;
    occd::ProfScope prof("swish_bwd", (hipStream_t)stream, 0.0, 12.0 * (double)n);
    hipLaunchKernelGGL(swish_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, gy, gx, (long)n);
    return occd::check_launch();
}

extern "C" int occd_affine_act_nchw(const float* x, const float* res, float* y, const float* scale, const float* shift,
                                    int32_t batch, int32_t C, int64_t S, int32_t act, float slope, int32_t res_first,
                                    void* stream) {
    if (!x || !y || batch <= 0 || C <= 0 || S <= 0 || act < 0 || act > 3) return OCCD_EINVAL;
    const long planes = (long)batch * C;
    if (planes > 65535) return OCCD_EINVAL;
    long bx = (S / 4 + 255) / 256;
    if (bx < 1) bx = 1;
    if (bx > 64) bx = 64;
    occd::ProfScope prof("affine_act_nchw", (hipStream_t)stream, 0.0, 4.0 * planes * S * (2 + (res != nullptr)));
    hipLaunchKernelGGL(affine_act_kernel, dim3((unsigned)bx, (unsigned)planes), dim3(256), 0, (hipStream_t)stream, x,
                       res, y, scale, shift, C, (long)S, act, slope, res_first);
    return occd::check_launch();
}

static int dwconv_launch(const float* x, const float* w, const float* scale, const float* shift, float* y,
                         int32_t batch, int32_t C, int32_t H, int32_t W, int32_t k, int32_t stride, int32_t pad_top,
                         int32_t pad_left, int32_t Ho, int32_t Wo, int32_t act, float* pool_part, int64_t x_plane_stride,
                         void* stream) {
    if (!x || !w || !y || batch <= 0 || C <= 0 || H <= 0 || W <= 0 || Ho <= 0 || Wo <= 0 || stride <= 0)
        return OCCD_EINVAL;
    if (x_plane_stride != 0 && x_plane_stride < (int64_t)H * W) return OCCD_EINVAL;
    const long xps = x_plane_stride != 0 ? (long)x_plane_stride : (long)H * W;
    if ((k != 3 && k != 5) || act < 0 || act > 2 || (long)batch * C > 65535) return OCCD_EINVAL;
    if (stride != 1 && stride != 2) return OCCD_EINVAL;
    const int items = Ho * ((Wo + 3) / 4);
    const dim3 grid((unsigned)((items + 255) / 256), (unsigned)(batch * C));
    occd::ProfScope prof("dwconv2d_nchw", (hipStream_t)stream, 2.0 * batch * C * (double)Ho * Wo * k * k,
                         4.0 * batch * C * ((double)H * W + (double)Ho * Wo));
    hipStream_t st = (hipStream_t)stream;
    // LDS tile of the staged kernel: rows of one workgroup's 256 items x padded width (see dwconv2d_kernel)
    const int wq = (Wo + 3) / 4;
    const int span4 = ((3 * stride + k) + 3) / 4;
    const int wp = 4 * (wq - 1) * stride + 4 * span4;
    int dmax = (255 + wq - 1) / wq;
    if (dmax > Ho - 1) dmax = Ho - 1;
    const size_t lds = (size_t)(dmax * stride + k) * wp * sizeof(float);
    if (lds <= 48 * 1024) {
#define OCCD_DW(KK, SS)                                                                                             \
    hipLaunchKernelGGL((dwconv2d_kernel<KK, SS>), grid, dim3(256), lds, st, x, w, scale, shift, y, C, H, W, Ho, Wo, \
                       pad_top, pad_left, act, pool_part, wp, occd::make_fastdiv((uint32_t)wp), xps)
        if (k == 3 && stride == 1) OCCD_DW(3, 1);
        else if (k == 3) OCCD_DW(3, 2);
        else if (stride == 1) OCCD_DW(5, 1);
        else OCCD_DW(5, 2);
#undef OCCD_DW
        return occd::check_launch();
    }
#define OCCD_DW(KK, SS)                                                                                                 \
    hipLaunchKernelGGL((dwconv2d_direct_kernel<KK, SS>), grid, dim3(256), 0, st, x, w, scale, shift, y, C, H, W, Ho, Wo, \
                       pad_top, pad_left, act, pool_part, xps)
    if (k == 3 && stride == 1) OCCD_DW(3, 1);
    else if (k == 3) OCCD_DW(3, 2);
    else if (stride == 1) OCCD_DW(5, 1);
    else OCCD_DW(5, 2);
#undef OCCD_DW
    return occd::check_launch();
}

extern "C" int occd_dwconv2d_nchw(const float* x, const float* w, const float* scale, const float* shift, float* y,
                                  int32_t batch, int32_t C, int32_t H, int32_t W, int32_t k, int32_t stride,
                                  int32_t pad_top, int32_t pad_left, int32_t Ho, int32_t Wo, int32_t act,
                                  void* stream) {
    return dwconv_launch(x, w, scale, shift, y, batch, C, H, W, k, stride, pad_top, pad_left, Ho, Wo, act, nullptr, 0, stream);
}

// ---- encoder stem: 3x3 convolution of a FEW input channels (the RGB image), stride 1 / 2, TensorFlow SAME padding,
// + BatchNorm affine + activation in one pass (geffnet conv_stem + bn1 + act1 behind occdepth/models/unet2d.py:175-190;
// round 5: the last MIOpen convolution of the encoder -- miopenSp3AsmConv f3x2_stride2 56 us + BatchNormFwdInfer + swish).
// K = 27: nothing for the matrix pipe; the launch is bound by its output stream (58 MB at config 2).  A workgroup is 64
// output pixels of one row x 4 groups of 16 output channels; a lane keeps the 27 taps of its pixel in registers and walks
// the 16 channels of its group with the (scale-folded) weights read from LDS as broadcast float4s.
constexpr int kStemCin = 3;
constexpr int kStemRows = 8;
// (4 waves per SIMD requested: left alone, hipcc hoists all 108 broadcast weight reads of a lane above the multiplies --
//  256 VGPRs + 224 AGPRs, one wave per SIMD, 156 us for the config-2 launch)
template <int STRIDE>
__global__ void __launch_bounds__(256, 2) stem_conv3x3_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                           const float* __restrict__ scale, const float* __restrict__ shift,
                                                           float* __restrict__ y, int H, int W, int Cout, int pad_top,
                                                           int pad_left, int Ho, int Wo, int act) {
    extern __shared__ float wl[];                       // [27][Cout16]: tap-major, couts contiguous (zero padded to 16)
    const int Cout16 = (Cout + 15) & ~15;
    for (int i = threadIdx.x; i < 27 * Cout16; i += 256) {
        const int t = i / Cout16, co = i - t * Cout16;
        wl[i] = co < Cout ? w[co * 27 + t] * (scale != nullptr ? scale[co] : 1.f) : 0.f;     // w (Cout, 3, 3, 3): t = ci*9 + ky*3 + kx
    }
    __syncthreads();
    const int ox = blockIdx.x * 64 + (threadIdx.x & 63);
    const int b = blockIdx.z;
    const int grp = threadIdx.x >> 6;
    if (ox >= Wo) return;
    // kStemRows output rows per workgroup: the weight prologue (a gather + a barrier, ~2 us of latency) is paid once per eight
    // rows, and the whole launch is one round of resident workgroups (3700 one-row workgroups took 82 us)
    for (int oy = blockIdx.y * kStemRows; oy < min(Ho, (int)(blockIdx.y + 1) * kStemRows); ++oy) {
    float v[27];
    const float* xb = x + (size_t)b * kStemCin * H * W;
#pragma unroll
    for (int ci = 0; ci < kStemCin; ++ci)
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int iy = oy * STRIDE - pad_top + ky;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                // unconditional load from a clamped address + select: a load guarded by a run-time condition becomes a
                // branch with its own s_waitcnt (27 serialised HBM round trips per thread: 153 us for this launch)
                const int ix = ox * STRIDE - pad_left + kx;
                const bool ok = (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
                const float t = xb[((size_t)ci * H + min(max(iy, 0), H - 1)) * W + min(max(ix, 0), W - 1)];
                v[ci * 9 + ky * 3 + kx] = ok ? t : 0.f;
            }
        }
    for (int c0 = grp * 16; c0 < Cout16; c0 += 64) {
        // four output channels at a time (a rolled loop: the weight offset is an address, not a register index), the 27 taps
        // unrolled in three fenced groups so that at most nine broadcast weight vectors are live beside the 27 taps
#pragma unroll 1
        for (int q = 0; q < 4; ++q) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            const float* wq = wl + c0 + 4 * q;
#pragma unroll
            for (int t0 = 0; t0 < 27; t0 += 9) {
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int t = t0; t < t0 + 9; ++t) acc += *(const f32x4*)(wq + t * Cout16) * v[t];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int co = c0 + 4 * q + j;
                if (co < Cout) {
                    const float o = acc[j] + (shift != nullptr ? shift[co] : 0.f);
                    y[(((size_t)b * Cout + co) * Ho + oy) * Wo + ox] = act_apply(o, act, 0.f);
                }
            }
        }
    }
    }
}

extern "C" int occd_stem_conv3x3_nchw(const float* x, const float* w, const float* scale, const float* shift, float* y,
                                      int32_t batch, int32_t H, int32_t W, int32_t cout, int32_t stride, int32_t pad_top,
                                      int32_t pad_left, int32_t Ho, int32_t Wo, int32_t act, void* stream) {
    if (x == nullptr || w == nullptr || y == nullptr) return OCCD_EINVAL;
    if (batch < 1 || batch > 65535 || H < 1 || W < 1 || cout < 1 || cout > 512 || (stride != 1 && stride != 2)) return OCCD_EINVAL;
    if (Ho < 1 || Ho > 65535 || Wo < 1 || pad_top < 0 || pad_left < 0 || act < 0 || act > 2) return OCCD_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((unsigned)((Wo + 63) / 64), (unsigned)((Ho + kStemRows - 1) / kStemRows), (unsigned)batch);
    const size_t lds = (size_t)27 * ((cout + 15) & ~15) * sizeof(float);
    occd::ProfScope prof("stem_conv3x3", st, 2.0 * batch * (double)Ho * Wo * 27 * cout,
                         4.0 * batch * ((double)kStemCin * H * W + (double)cout * Ho * Wo));
    if (stride == 2)
        hipLaunchKernelGGL(stem_conv3x3_kernel<2>, grid, dim3(256), lds, st, x, w, scale, shift, y, H, W, cout, pad_top,
                           pad_left, Ho, Wo, act);
    else
        hipLaunchKernelGGL(stem_conv3x3_kernel<1>, grid, dim3(256), lds, st, x, w, scale, shift, y, H, W, cout, pad_top,
                           pad_left, Ho, Wo, act);
    return occd::check_launch();
}

extern "C" int32_t occd_dwconv2d_pool_blocks(int32_t Ho, int32_t Wo) { return (Ho * ((Wo + 3) / 4) + 255) / 256; }

extern "C" int occd_dwconv2d_pool_nchw(const float* x, const float* w, const float* scale, const float* shift, float* y,
                                       float* pool_part, int32_t batch, int32_t C, int32_t H, int32_t W, int32_t k,
                                       int32_t stride, int32_t pad_top, int32_t pad_left, int32_t Ho, int32_t Wo,
                                       int32_t act, int64_t x_plane_stride, void* stream) {
    if (pool_part == nullptr) return OCCD_EINVAL;
    return dwconv_launch(x, w, scale, shift, y, batch, C, H, W, k, stride, pad_top, pad_left, Ho, Wo, act, pool_part,
                         x_plane_stride, stream);
}

// ---- channels-last twins (bf16-mode training: the decoder levels live as (B, H, W, C) pixel rows) ------------------
// forward: out[b, Y, X, :C] = bilinear(x)(Y, X) (align_corners=True, the arithmetic of upsample_cat_kernel),
//          out[b, Y, X, C:] = skip[b, Y, X, :].   One thread per output element, channels fastest (coalesced rows; the
//          channel counts are not multiples of 4: 163 = 160 + 3 at the full-resolution level).
__global__ void __launch_bounds__(256) upsample_cat_nhwc_kernel(const float* __restrict__ x, const float* __restrict__ skip,
                                                                float* __restrict__ out, int C, int Cs, int h, int w, int H,
                                                                int W, float rh, float rw, long total, int ct) {
    // ct: floats per output row (>= C + Cs; the pad lanes are written as zeros -- round 6: rows of ceil8(C + Cs) floats are what
    // the convolution kernels take in place, 163 channels in rows of 168 at the full-resolution level)
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int c = (int)(i % ct);
    long t = i / ct;
    const int ox = (int)(t % W); t /= W;
    const int oy = (int)(t % H);
    const int b = (int)(t / H);
    float v;
    if (c >= C + Cs) {
        v = 0.f;
    } else if (c >= C) {
        v = skip[(((size_t)b * H + oy) * W + ox) * Cs + (c - C)];
    } else {
        const float sy = rh * oy;
        const int y0 = (int)sy;
        const int y1 = y0 + (y0 < h - 1);
        const float ly = sy - y0, hy = 1.f - ly;
        const float sx = rw * ox;
        const int x0 = (int)sx;
        const int x1 = x0 + (x0 < w - 1);
        const float lx = sx - x0, hx = 1.f - lx;
        const float* p0 = x + (((size_t)b * h + y0) * w) * C + c;
        const float* p1 = x + (((size_t)b * h + y1) * w) * C + c;
        v = hy * (hx * p0[(size_t)x0 * C] + lx * p0[(size_t)x1 * C]) + ly * (hx * p1[(size_t)x0 * C] + lx * p1[(size_t)x1 * C]);
    }
    out[i] = v;
}

// backward of the upsampled part as a GATHER (ATen's nhwc backward scatters with atomics: 0.39 ms per level at config 2):
// gx[b, y, x, c] = sum over the output pixels whose 2 x 2 footprint contains (y, x) of their weight * gout[b, Y, X, c],
// every candidate re-deriving (y0, y1, ly) with the forward's own arithmetic, so the pair is an exact transpose.
__global__ void __launch_bounds__(256) upsample_nhwc_bwd_kernel(const float* __restrict__ gout, float* __restrict__ gx, int C,
                                                                int gcs, int h, int w, int H, int W, float rh, float rw,
                                                                float inv_rh, float inv_rw, long total) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int c = (int)(i % C);
    long t = i / C;
    const int x = (int)(t % w); t /= w;
    const int y = (int)(t % h);
    const int b = (int)(t / h);
    // candidate output rows / columns: src = dst * r in (y - 1, y + 1)
    const int Ya = max(0, (int)floorf((y - 1) * inv_rh) - 1), Yb = min(H - 1, (int)ceilf((y + 1) * inv_rh) + 1);
    const int Xa = max(0, (int)floorf((x - 1) * inv_rw) - 1), Xb = min(W - 1, (int)ceilf((x + 1) * inv_rw) + 1);
    float acc = 0.f;
    for (int Y = Ya; Y <= Yb; ++Y) {
        const float sy = rh * Y;
        const int y0 = (int)sy;
        const int y1 = y0 + (y0 < h - 1);
        const float ly = sy - y0;
        const float wy = (y0 == y ? 1.f - ly : 0.f) + (y1 == y ? ly : 0.f);
        if (wy == 0.f) continue;
        const float* row = gout + (((size_t)b * H + Y) * W) * gcs + c;
        float part = 0.f;
        for (int X = Xa; X <= Xb; ++X) {
            const float sx = rw * X;
            const int x0 = (int)sx;
            const int x1 = x0 + (x0 < w - 1);
            const float lx = sx - x0;
            const float wx = (x0 == x ? 1.f - lx : 0.f) + (x1 == x ? lx : 0.f);
            if (wx != 0.f) part += wx * row[(size_t)X * gcs];
        }
        acc += wy * part;
    }
    gx[i] = acc;
}

extern "C" int occd_upsample_bilinear_cat_nchw(const float* x, const float* skip, float* out, int32_t batch, int32_t C,
                                               int32_t Cskip, int32_t h, int32_t w, int32_t H, int32_t W,
                                               void* stream) {
    if (!x || !out || batch <= 0 || C <= 0 || Cskip < 0 || (Cskip > 0 && !skip) || h <= 0 || w <= 0 || H <= 0 ||
        W <= 0 || (long)batch * (C + Cskip) > 65535)
        return OCCD_EINVAL;
    // at::native::area_pixel_compute_scale<float>(in, out, align_corners=true)
    const float rh = H > 1 ? (float)(h - 1) / (float)(H - 1) : 0.f;
    const float rw = W > 1 ? (float)(w - 1) / (float)(W - 1) : 0.f;
    const dim3 grid((unsigned)((W + 255) / 256), (unsigned)((H + 3) / 4), (unsigned)(batch * (C + Cskip)));
    occd::ProfScope prof("upsample_cat_nchw", (hipStream_t)stream, 0.0,
                         4.0 * batch * ((double)C * h * w + 2.0 * Cskip * H * W + (double)C * H * W));
    hipLaunchKernelGGL(upsample_cat_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, skip, out, C, Cskip, h, w, H, W,
                       rh, rw);
    return occd::check_launch();
}

extern "C" int occd_upsample_bilinear_cat_nhwc_rows(const float* x, const float* skip, float* out, int32_t batch, int32_t C,
                                                    int32_t Cskip, int32_t h, int32_t w, int32_t H, int32_t W, int32_t out_cs,
                                                    void* stream);
extern "C" int occd_upsample_bilinear_cat_nhwc(const float* x, const float* skip, float* out, int32_t batch, int32_t C,
                                               int32_t Cskip, int32_t h, int32_t w, int32_t H, int32_t W, void* stream) {
    return occd_upsample_bilinear_cat_nhwc_rows(x, skip, out, batch, C, Cskip, h, w, H, W, C + Cskip, stream);
}

// the same with output rows of out_cs >= C + Cskip floats (pad lanes zeroed)
extern "C" int occd_upsample_bilinear_cat_nhwc_rows(const float* x, const float* skip, float* out, int32_t batch, int32_t C,
                                                    int32_t Cskip, int32_t h, int32_t w, int32_t H, int32_t W, int32_t out_cs,
                                                    void* stream) {
    if (!x || !out || batch <= 0 || C <= 0 || Cskip < 0 || (Cskip > 0 && !skip) || h <= 0 || w <= 0 || H <= 0 || W <= 0 ||
        out_cs < C + Cskip)
        return OCCD_EINVAL;
    const float rh = H > 1 ? (float)(h - 1) / (float)(H - 1) : 0.f;
    const float rw = W > 1 ? (float)(w - 1) / (float)(W - 1) : 0.f;
    const long total = (long)batch * H * W * out_cs;
    occd::ProfScope prof("upsample_cat_nhwc", (hipStream_t)stream, 0.0,
                         4.0 * batch * ((double)C * h * w + 2.0 * Cskip * H * W + (double)C * H * W));
    hipLaunchKernelGGL(upsample_cat_nhwc_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x,
                       skip, out, C, Cskip, h, w, H, W, rh, rw, total, (int)out_cs);
    return occd::check_launch();
}

extern "C" int occd_upsample_bilinear_nhwc_bwd(const float* gout, float* gx, int32_t batch, int32_t C, int32_t gout_cs,
                                               int32_t h, int32_t w, int32_t H, int32_t W, void* stream) {
    if (!gout || !gx || batch <= 0 || C <= 0 || gout_cs < C || h <= 0 || w <= 0 || H <= 0 || W <= 0) return OCCD_EINVAL;
    const float rh = H > 1 ? (float)(h - 1) / (float)(H - 1) : 0.f;
    const float rw = W > 1 ? (float)(w - 1) / (float)(W - 1) : 0.f;
    // (rh == 0: every output pixel reads source 0 -- all of them are candidates)
    const float inv_rh = rh > 0.f ? 1.f / rh : (float)H, inv_rw = rw > 0.f ? 1.f / rw : (float)W;
    const long total = (long)batch * h * w * C;
    occd::ProfScope prof("upsample_nhwc_bwd", (hipStream_t)stream, 0.0, 4.0 * batch * C * ((double)h * w + (double)H * W));
    hipLaunchKernelGGL(upsample_nhwc_bwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, gout,
                       gx, C, gout_cs, h, w, H, W, rh, rw, inv_rh, inv_rw, total);
    return occd::check_launch();
}

extern "C" int occd_upconv_gather_nchw(const float* z, float* out, int32_t batch, int32_t Cout, int32_t h, int32_t w,
                                       int32_t H, int32_t W, int64_t z_channel_stride, int64_t z_batch_stride,
                                       void* stream) {
    if (!z || !out || batch <= 0 || Cout <= 0 || h <= 0 || w <= 0 || H <= 0 || W <= 0 || (long)batch * Cout > 65535)
        return OCCD_EINVAL;
    const long zcs = z_channel_stride > 0 ? z_channel_stride : (long)h * w;
    const long zbs = z_batch_stride > 0 ? z_batch_stride : 9L * Cout * zcs;
    if (zcs < (long)h * w) return OCCD_EINVAL;
    const float rh = H > 1 ? (float)(h - 1) / (float)(H - 1) : 0.f;
    const float rw = W > 1 ? (float)(w - 1) / (float)(W - 1) : 0.f;
    // (the staged kernel addresses one image's z with 32-bit buffer offsets; OCCD_UPCONV_DIRECT: A/B switch)
    const bool staged = !occd::env_flag("OCCD_UPCONV_DIRECT", false) && rw * 258.f + 3.f <= (float)kUpNC &&
                        9L * Cout * zcs * 4 < (1L << 31);
    const unsigned groups = (unsigned)((H + 3) / 4);
    const dim3 grid((unsigned)((W + 255) / 256), groups, (unsigned)(batch * Cout));
    occd::ProfScope prof("upconv_gather_nchw", (hipStream_t)stream, 2.0 * 36 * batch * Cout * (double)H * W,
                         4.0 * batch * Cout * (9.0 * h * w + (double)H * W));
    // the staged kernel holds the low-resolution columns under 258 output columns in rows of kUpNC floats
    // COOP: shared source rows (needs an upsampling ratio >= 2 so that four output rows see four source rows per ky);
    // opt-in (OCCD_UPCONV_COOP=1): measured slower, see the kernel
    static const bool coop_on = occd::env_flag("OCCD_UPCONV_COOP", false);
    if (staged && coop_on && rh <= 0.5f)
        hipLaunchKernelGGL((upconv_gather_kernel<false, true>), grid, dim3(256), 0, (hipStream_t)stream, z, out, Cout, h, w, H, W,
                           rh, rw, zcs, zbs, UpSkipP{});
    else if (staged)
        hipLaunchKernelGGL(upconv_gather_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, z, out, Cout, h, w, H, W, rh,
                           rw, zcs, zbs, UpSkipP{});
    else
        hipLaunchKernelGGL(upconv_gather_direct_kernel, grid, dim3(256), 0, (hipStream_t)stream, z, out, Cout, h, w, H, W,
                           rh, rw, zcs, zbs);
    return occd::check_launch();
}

extern "C" int occd_upconv_gather_skip_nchw(const float* z, const float* skip, const float* wskip, const float* shift,
                                            float* out, int32_t batch, int32_t Cout, int32_t Cs, int32_t h, int32_t w,
                                            int32_t H, int32_t W, int64_t z_channel_stride, int64_t z_batch_stride,
                                            float slope, void* stream) {
    if (!z || !skip || !wskip || !shift || !out || batch <= 0 || Cout <= 0 || h <= 0 || w <= 0 || H <= 0 || W <= 0 ||
        (long)batch * Cout > 65535 || Cs < 1 || Cs > kUpSkipC)
        return OCCD_EINVAL;
    const long zcs = z_channel_stride > 0 ? z_channel_stride : (long)h * w;
    const long zbs = z_batch_stride > 0 ? z_batch_stride : 9L * Cout * zcs;
    if (zcs < (long)h * w) return OCCD_EINVAL;
    const float rh = H > 1 ? (float)(h - 1) / (float)(H - 1) : 0.f;
    const float rw = W > 1 ? (float)(w - 1) / (float)(W - 1) : 0.f;
    if (rw * 258.f + 3.f > (float)kUpNC) return OCCD_EINVAL;          // (upsampling ratios below ~1.4: use the two-kernel form)
    if (9L * Cout * zcs * 4 >= (1L << 31) || (long)Cs * H * W * 4 >= (1L << 31)) return OCCD_EINVAL;   // 32-bit buffer offsets
    const dim3 grid((unsigned)((W + 255) / 256), (unsigned)((H + 3) / 4), (unsigned)(batch * Cout));
    occd::ProfScope prof("upconv_gather_skip_nchw", (hipStream_t)stream, 2.0 * (36 + 9.0 * Cs) * batch * Cout * (double)H * W,
                         4.0 * batch * (Cout * (9.0 * h * w + (double)H * W) + (double)Cs * H * W));
    UpSkipP sk{skip, wskip, shift, Cs, slope};
    static const bool coop_on = occd::env_flag("OCCD_UPCONV_COOP", false);
    if (coop_on && rh <= 0.5f)
        hipLaunchKernelGGL((upconv_gather_kernel<true, true>), grid, dim3(256), 0, (hipStream_t)stream, z, out, Cout, h, w, H, W,
                           rh, rw, zcs, zbs, sk);
    else
        hipLaunchKernelGGL(upconv_gather_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, z, out, Cout, h, w, H, W, rh, rw,
                           zcs, zbs, sk);
    return occd::check_launch();
}

// ------------------------------------------------------------------------------------------------
// Cascade-head tail (occdepth/models/modules.py:166-173): the narrow half of `conv_classes`,
//   ssc[v][o] = part[v][o] + sum_{tap, c<2} softmax(occ[v + tap])[c] * Wn[o][c][tap],
// where `part` already holds conv_classes[:, :planes](feat) + bias (wide half, MFMA kernel) and the occ
// logits.  K = 27 * 2 is far too thin for the matrix pipe (an N=32 x K=8 MFMA tile wastes 94 %), so this is
// a VALU kernel: one thread per voxel, the 54 x nbr weights broadcast from LDS, the 2-way softmax of each
// neighbour recomputed on the fly (no softmax / concat buffer exists).
namespace {

constexpr int kTailMaxOut = 32;

template <int NG>   // NG float4 groups of outputs per voxel (ceil(nbr / 4))
__global__ void __launch_bounds__(256) cascade_tail_kernel(const float* __restrict__ part, const float* __restrict__ wn,
                                                           float* __restrict__ out, int B, int X, int Y, int Z,
                                                           int part_cs, int occ_off, int out_cs, int nbr) {
    __shared__ __attribute__((aligned(16))) float wl[27 * 2 * NG * 4];   // [tap][c][o]
    for (int i = threadIdx.x; i < 27 * 2 * NG * 4; i += 256) {
        const int o = i % (NG * 4), c = (i / (NG * 4)) & 1, tap = i / (2 * NG * 4);
        wl[i] = o < nbr ? wn[((size_t)o * 2 + c) * 27 + tap] : 0.f;
    }
    __syncthreads();
    const long n = (long)blockIdx.x * 256 + threadIdx.x;
    const long total = (long)B * X * Y * Z;
    if (n >= total) return;
    const int z = (int)(n % Z);
    long t = n / Z;
    const int y = (int)(t % Y);
    t /= Y;
    const int x = (int)(t % X);
    f32x4 acc[NG];
#pragma unroll
    for (int g = 0; g < NG; ++g) acc[g] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float* occ = part + (size_t)n * part_cs + occ_off;
#pragma unroll 1
    for (int dx = -1; dx <= 1; ++dx) {
        if ((unsigned)(x + dx) >= (unsigned)X) continue;
#pragma unroll 1
        for (int dy = -1; dy <= 1; ++dy) {
            if ((unsigned)(y + dy) >= (unsigned)Y) continue;
            // the three z-neighbours of one (dx, dy) are adjacent rows: issue their loads together
            float l0[3], l1[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const int zz = z + k - 1;
                const bool ok = (unsigned)zz < (unsigned)Z;
                const float* l = occ + (((long)dx * Y + dy) * Z + (ok ? k - 1 : 0)) * part_cs;
                l0[k] = ok ? l[0] : 0.f;
                l1[k] = ok ? l[1] : 0.f;
            }
            const int tap0 = ((dx + 1) * 3 + (dy + 1)) * 3;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                if ((unsigned)(z + k - 1) >= (unsigned)Z) continue;
                const float m = fmaxf(l0[k], l1[k]);
                const float e0 = expf(l0[k] - m), e1 = expf(l1[k] - m);
                const float s0 = e0 / (e0 + e1), s1 = e1 / (e0 + e1);
                const f32x4* w0 = (const f32x4*)(wl + ((tap0 + k) * 2 + 0) * NG * 4);
                const f32x4* w1 = (const f32x4*)(wl + ((tap0 + k) * 2 + 1) * NG * 4);
#pragma unroll
                for (int g = 0; g < NG; ++g) acc[g] += s0 * w0[g] + s1 * w1[g];
            }
        }
    }
    const float* pr = part + (size_t)n * part_cs;
    float* po = out + (size_t)n * out_cs;
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        const f32x4 v = acc[g] + *(const f32x4*)(pr + g * 4);
        if (g * 4 + 3 < nbr) {
            *(f32x4*)(po + g * 4) = v;
        } else {
            if (g * 4 + 0 < nbr) po[g * 4 + 0] = v.x;
            if (g * 4 + 1 < nbr) po[g * 4 + 1] = v.y;
            if (g * 4 + 2 < nbr) po[g * 4 + 2] = v.z;
        }
    }
}

// Staged form (Z <= 64, i.e. every shipped head): a workgroup owns TY rows x Z voxels of one x plane.  The direct
// kernel above reads the two occupancy logits of all 27 neighbours straight from the 128-byte voxel rows -- 54 wave
// loads of 64 different cache lines each, so it is bound by the texture-address path (0.30 ms at config 2) -- here
// the 3 x (TY + 2) x (Z + 2) neighbourhood is read ONCE per workgroup (4 row touches per voxel instead of 54), its
// softmax is taken once per voxel instead of 27 times, and the taps run out of LDS.  Same arithmetic per tap, same
// tap order as the direct kernel: results are bit-identical.
template <int NG>
__global__ void __launch_bounds__(256) cascade_tail_lds_kernel(const float* __restrict__ part, const float* __restrict__ wn,
                                                               float* __restrict__ out, int X, int Y, int Z, int TY,
                                                               int part_cs, int occ_off, int out_cs, int nbr) {
    __shared__ __attribute__((aligned(16))) float wl[27 * 2 * NG * 4];   // [tap][c][o]
    extern __shared__ __attribute__((aligned(16))) float sm[];           // [3][TY + 2][Z + 2][2] softmax(occ), 0 outside
    for (int i = threadIdx.x; i < 27 * 2 * NG * 4; i += 256) {
        const int o = i % (NG * 4), c = (i / (NG * 4)) & 1, tap = i / (2 * NG * 4);
        wl[i] = o < nbr ? wn[((size_t)o * 2 + c) * 27 + tap] : 0.f;
    }
    const int b = blockIdx.z, x = blockIdx.y, y0 = blockIdx.x * TY;
    const int ZP = Z + 2, YP = TY + 2;
    const int nhalo = 3 * YP * ZP;
    for (int e = threadIdx.x; e < nhalo; e += 256) {
        const int zi = e % ZP, r = e / ZP;
        const int yi = r % YP, xi = r / YP;
        const int xx = x + xi - 1, yy = y0 + yi - 1, zz = zi - 1;
        const bool ok = (unsigned)xx < (unsigned)X && (unsigned)yy < (unsigned)Y && (unsigned)zz < (unsigned)Z;
        const size_t v = (((size_t)b * X + (ok ? xx : x)) * Y + (ok ? yy : y0)) * Z + (ok ? zz : 0);
        const float l0 = part[v * part_cs + occ_off], l1 = part[v * part_cs + occ_off + 1];
        const float m = fmaxf(l0, l1);
        const float e0 = expf(l0 - m), e1 = expf(l1 - m);
        sm[2 * e] = ok ? e0 / (e0 + e1) : 0.f;
        sm[2 * e + 1] = ok ? e1 / (e0 + e1) : 0.f;
    }
    __syncthreads();
    const int ly = threadIdx.x / Z, z = threadIdx.x - ly * Z;
    const int y = y0 + ly;
    if (ly >= TY || y >= Y) return;
    f32x4 acc[NG];
#pragma unroll
    for (int g = 0; g < NG; ++g) acc[g] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
    for (int dx = 0; dx < 3; ++dx)
#pragma unroll 1
        for (int dy = 0; dy < 3; ++dy) {
            const float* row = sm + 2 * (((dx * YP) + ly + dy) * ZP + z);
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const float s0 = row[2 * k], s1 = row[2 * k + 1];
                const f32x4* w0 = (const f32x4*)(wl + (((dx * 3 + dy) * 3 + k) * 2 + 0) * NG * 4);
                const f32x4* w1 = (const f32x4*)(wl + (((dx * 3 + dy) * 3 + k) * 2 + 1) * NG * 4);
#pragma unroll
                for (int g = 0; g < NG; ++g) acc[g] += s0 * w0[g] + s1 * w1[g];
            }
        }
    const size_t n = (((size_t)b * X + x) * Y + y) * Z + z;
    const float* pr = part + n * part_cs;
    float* po = out + n * out_cs;
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        const f32x4 v = acc[g] + *(const f32x4*)(pr + g * 4);
        if (g * 4 + 3 < nbr) {
            *(f32x4*)(po + g * 4) = v;
        } else {
            if (g * 4 + 0 < nbr) po[g * 4 + 0] = v.x;
            if (g * 4 + 1 < nbr) po[g * 4 + 1] = v.y;
            if (g * 4 + 2 < nbr) po[g * 4 + 2] = v.z;
        }
    }
}

}  // namespace

extern "C" int occd_cascade_tail_fwd(const float* part, const float* wn, float* out, int32_t batch, int32_t X,
                                     int32_t Y, int32_t Z, int32_t part_cs, int32_t occ_off, int32_t out_cs,
                                     int32_t nbr, void* stream) {
    if (!part || !wn || !out || batch <= 0 || X <= 0 || Y <= 0 || Z <= 0) return OCCD_EINVAL;
    if (nbr <= 0 || nbr > kTailMaxOut || (part_cs & 3) || (out_cs & 3) || (occ_off & 1) || occ_off + 2 > part_cs ||
        ((nbr + 3) & ~3) > part_cs || nbr > out_cs)
        return OCCD_EINVAL;
    const long total = (long)batch * X * Y * Z;
    occd::ProfScope prof("cascade_tail", (hipStream_t)stream, 2.0 * total * 54 * nbr, 4.0 * total * (2.0 * nbr + 2));
    const dim3 grid((unsigned)((total + 255) / 256));
    const int ng = (nbr + 3) >> 2;
    hipStream_t st = (hipStream_t)stream;
    if (Z <= 64 && X <= 65535 && batch <= 65535) {
        const int TY = 256 / Z;
        const dim3 tgrid((unsigned)((Y + TY - 1) / TY), (unsigned)X, (unsigned)batch);
        const size_t lds = (size_t)3 * (TY + 2) * (Z + 2) * 2 * sizeof(float);
#define OCCD_TAIL_LDS(NG)                                                                                            \
    hipLaunchKernelGGL(cascade_tail_lds_kernel<NG>, tgrid, dim3(256), lds, st, part, wn, out, X, Y, Z, TY, part_cs, \
                       occ_off, out_cs, nbr)
        if (ng <= 3) OCCD_TAIL_LDS(3);
        else if (ng <= 5) OCCD_TAIL_LDS(5);
        else OCCD_TAIL_LDS(8);
#undef OCCD_TAIL_LDS
        return occd::check_launch();
    }
#define OCCD_TAIL(NG)                                                                                        \
    hipLaunchKernelGGL(cascade_tail_kernel<NG>, grid, dim3(256), 0, st, part, wn, out, batch, X, Y, Z, part_cs, \
                       occ_off, out_cs, nbr)
    if (ng <= 3) OCCD_TAIL(3);
    else if (ng <= 5) OCCD_TAIL(5);
    else OCCD_TAIL(8);
#undef OCCD_TAIL
    return occd::check_launch();
}

// ------------------------------------------------------------------------------------------------
// SURVEY.md 8(f) row N4 (first step): class prediction on the GPU.  scripts/generate_output.py:94-95 copies
// the (B, 20, 256, 256, 32) logits to the host, soft-maxes and arg-maxes them in numpy; argmax(softmax(x)) ==
// argmax(x), so one pass over the channels-last rows yields the uint8 label volume (first maximum wins, like
// numpy), optionally mapped through a LUT (learning_map_inv of generate_kitti_submission.py:74-85).
namespace {
__global__ void __launch_bounds__(256) argmax_rows_kernel(const float* __restrict__ x, long rows, int cs, int coff,
                                                          int C, const uint16_t* __restrict__ lut,
                                                          uint16_t* __restrict__ out) {
    const long r = (long)blockIdx.x * 256 + threadIdx.x;
    if (r >= rows) return;
    const float* p = x + (size_t)r * cs + coff;
    float best = p[0];
    int arg = 0;
    for (int c = 1; c < C; ++c) {
        const float v = p[c];
        if (v > best) { best = v; arg = c; }
    }
    out[r] = lut ? lut[arg] : (uint16_t)arg;
}
}  // namespace

extern "C" int occd_argmax_channels(const float* x, int64_t rows, int32_t cs, int32_t coff, int32_t C,
                                    const uint16_t* lut, uint16_t* out, void* stream) {
    if (!x || !out || rows <= 0 || C <= 0 || coff < 0 || coff + C > cs) return OCCD_EINVAL;
    occd::ProfScope prof("argmax_channels", (hipStream_t)stream, 0.0, (double)rows * (4.0 * C + 2));
    hipLaunchKernelGGL(argmax_rows_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x,
                       (long)rows, cs, coff, C, lut, out);
    return occd::check_launch();
}


extern "C" int occd_dwconv2d_bwd_data_nchw(const float* gy, const float* w, float* dx, int32_t batch, int32_t C, int32_t H,
                                           int32_t W, int32_t k, int32_t stride, int32_t pad_top, int32_t pad_left,
                                           int32_t Ho, int32_t Wo, void* stream) {
    if (!gy || !w || !dx || batch <= 0 || C <= 0 || H <= 0 || W <= 0 || Ho <= 0 || Wo <= 0) return OCCD_EINVAL;
    if ((k != 3 && k != 5) || (stride != 1 && stride != 2) || (long)batch * C > 65535) return OCCD_EINVAL;
    const int items = H * ((W + 3) / 4);
    const dim3 grid((unsigned)((items + 255) / 256), (unsigned)(batch * C));
    hipStream_t st = (hipStream_t)stream;
    occd::ProfScope prof("dwconv2d_bwd_data", st, 2.0 * batch * C * (double)H * W * k * k,
                         4.0 * batch * C * ((double)H * W + (double)Ho * Wo));
#define OCCD_DWB(KK, SS) \
    hipLaunchKernelGGL((dwconv2d_bwd_data_kernel<KK, SS>), grid, dim3(256), 0, st, gy, w, dx, C, H, W, Ho, Wo, pad_top, pad_left)
    if (k == 3 && stride == 1) OCCD_DWB(3, 1);
    else if (k == 3) OCCD_DWB(3, 2);
    else if (stride == 1) OCCD_DWB(5, 1);
    else OCCD_DWB(5, 2);
#undef OCCD_DWB
    return occd::check_launch();
}

extern "C" int64_t occd_dwconv2d_bwd_weight_workspace_floats(int32_t batch, int32_t C, int32_t k, int32_t Ho, int32_t Wo) {
    if (batch <= 0 || C <= 0 || Ho <= 0 || Wo <= 0 || (k != 3 && k != 5)) return OCCD_EINVAL;
    long chunks = ((long)batch * Ho * Wo + 256 * 16 - 1) / (256 * 16);
    if (chunks > 64) chunks = 64;
    return (int64_t)C * chunks * k * k;
}

extern "C" int occd_dwconv2d_bwd_weight_nchw(const float* x, const float* gy, float* dw, float* workspace, int32_t batch,
                                             int32_t C, int32_t H, int32_t W, int32_t k, int32_t stride, int32_t pad_top,
                                             int32_t pad_left, int32_t Ho, int32_t Wo, void* stream) {
    if (!x || !gy || !dw || !workspace || batch <= 0 || C <= 0 || C > 65535 || H <= 0 || W <= 0 || Ho <= 0 || Wo <= 0)
        return OCCD_EINVAL;
    if ((k != 3 && k != 5) || (stride != 1 && stride != 2)) return OCCD_EINVAL;
    long chunks = ((long)batch * Ho * Wo + 256 * 16 - 1) / (256 * 16);
    if (chunks > 64) chunks = 64;
    const dim3 grid((unsigned)chunks, (unsigned)C);
    hipStream_t st = (hipStream_t)stream;
    occd::ProfScope prof("dwconv2d_bwd_weight", st, 2.0 * batch * C * (double)Ho * Wo * k * k,
                         4.0 * batch * C * ((double)H * W + (double)Ho * Wo));
#define OCCD_DWW(KK, SS) \
    hipLaunchKernelGGL((dwconv2d_bwd_weight_kernel<KK, SS>), grid, dim3(256), 0, st, x, gy, workspace, batch, C, H, W, Ho, Wo, pad_top, pad_left)
    if (k == 3 && stride == 1) OCCD_DWW(3, 1);
    else if (k == 3) OCCD_DWW(3, 2);
    else if (stride == 1) OCCD_DWW(5, 1);
    else OCCD_DWW(5, 2);
#undef OCCD_DWW
    hipLaunchKernelGGL(dwconv2d_bwd_weight_reduce_kernel, dim3((unsigned)((C * k * k + 255) / 256)), dim3(256), 0, st,
                       workspace, dw, C, k * k, (int)chunks);
    return occd::check_launch();
}
