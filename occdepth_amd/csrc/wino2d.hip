// Winograd F(2x2, 3x3) input / output transforms for the 2-D decoder convolutions.
//
//   y = conv3x3(x, g), stride 1, pad 1      ==      per 2x2 output tile:  Y = A^T [ sum_cin (G g G^T) (.) (B^T d B) ] A
//
// with d the 4x4 input patch of the tile.  The sum over cin for the 16 positions of the 4x4 Winograd domain is 16
// independent GEMMs  M[xi] (T x Cout) = V[xi] (T x Cin) . U[xi] (Cin x Cout)  -- 2.25x fewer multiplies than the direct
// convolution, and GEMM-shaped, so they run on the MFMA pipe (rocBLAS batched sgemm), whereas MIOpen's fp32 Winograd
// runs on the VALU.  This file holds the two memory-bound ends:
//   wino_input_kernel : x (B, Cin, H, W) NCHW  ->  V (16, T, Cin),  T = B * ceil(H/2) * ceil(W/2) tiles
//   wino_output_kernel: M (16, T, Cout)        ->  y (B, Cout, H, W) NCHW, fused  act(scale[c] * Y + shift[c]) (+ res)
// Both transpose through LDS so that the NCHW side is read/written along W and the (T, C) side along C.
// It pays where tiles are few and channels are many (the 1/16, 1/8, 1/4 decoder levels: 3.6 ms/frame); at the two
// high-resolution levels the V / M round trip costs more than it saves and MIOpen stays.
//
// Reference semantics: nn.Conv2d(k=3, s=1, p=1) + BatchNorm2d (eval) + LeakyReLU of occdepth/models/unet2d.py:24-46.
#include "common.h"

namespace {

constexpr int kTilesX = 32;                 // tiles per workgroup along x (64 output columns)
constexpr int kCh = 32;                     // channels per workgroup
constexpr int kCols = 2 * kTilesX + 2;      // staged input columns (one halo column each side)

__device__ __forceinline__ float act_apply2(float v, int act, float slope) {
    if (act == 1) return fmaxf(v, 0.f);
    if (act == 2) return v / (1.f + expf(-v));
    if (act == 3) return v > 0.f ? v : v * slope;
    return v;
}

// grid: (ceil(tw / 32), th, B * ceil(Cin / 32)); 256 threads
// (ty0, ths): the tile rows [ty0, ty0 + ths) of every image form the T = B * ths * tw rows of this call's V (a strip)
__global__ void __launch_bounds__(256) wino_input_kernel(const float* __restrict__ x, float* __restrict__ V, int B,
                                                         int Cin, int H, int W, int ths, int tw, int ty0) {
    __shared__ float tile[kCh][4 * kCols + 1];            // [channel][row * kCols + col], odd stride: conflict-free
    const int cblocks = (Cin + kCh - 1) / kCh;
    const int b = blockIdx.z / cblocks, c0 = (blockIdx.z - b * cblocks) * kCh;
    const int tys = blockIdx.y, ty = ty0 + tys, tx0 = blockIdx.x * kTilesX;
    const int y0 = 2 * ty - 1, x0 = 2 * tx0 - 1;
    // stage 32 channels x 4 rows x 66 columns (zero outside the image = the convolution's padding)
    for (int i = threadIdx.x; i < kCh * 4 * kCols; i += 256) {
        const int col = i % kCols, rc = i / kCols;
        const int r = rc & 3, c = rc >> 2;
        const int yy = y0 + r, xx = x0 + col, cc = c0 + c;
        float v = 0.f;
        if (cc < Cin && yy >= 0 && yy < H && xx >= 0 && xx < W) v = x[(((size_t)b * Cin + cc) * H + yy) * W + xx];
        tile[c][r * kCols + col] = v;
    }
    __syncthreads();
    const int c = threadIdx.x & 31;
    const size_t T = (size_t)B * ths * tw;
    for (int t = threadIdx.x >> 5; t < kTilesX; t += 8) {
        const int tx = tx0 + t;
        if (tx >= tw || c0 + c >= Cin) continue;
        float d[4][4];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int j = 0; j < 4; ++j) d[r][j] = tile[c][r * kCols + 2 * t + j];
        // B^T d B,  B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1]
        float w[4][4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            w[0][j] = d[0][j] - d[2][j];
            w[1][j] = d[1][j] + d[2][j];
            w[2][j] = d[2][j] - d[1][j];
            w[3][j] = d[1][j] - d[3][j];
        }
        const size_t tidx = ((size_t)b * ths + tys) * tw + tx;
        float* o = V + tidx * Cin + c0 + c;
        const size_t xs = T * Cin;                            // stride between Winograd positions
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            o[(size_t)(4 * r + 0) * xs] = w[r][0] - w[r][2];
            o[(size_t)(4 * r + 1) * xs] = w[r][1] + w[r][2];
            o[(size_t)(4 * r + 2) * xs] = w[r][2] - w[r][1];
            o[(size_t)(4 * r + 3) * xs] = w[r][1] - w[r][3];
        }
    }
}

// grid: (ceil(tw / 32), th, B * ceil(Cout / 32)); 256 threads
__global__ void __launch_bounds__(256) wino_output_kernel(const float* __restrict__ M, const float* __restrict__ scale,
                                                          const float* __restrict__ shift, const float* __restrict__ res,
                                                          float* __restrict__ y, int B, int Cout, int H, int W, int ths,
                                                          int tw, int ty0, int act, float slope, int res_first) {
    __shared__ float tile[kCh][2 * 2 * kTilesX + 1];      // [channel][row * 64 + col]
    const int cblocks = (Cout + kCh - 1) / kCh;
    const int b = blockIdx.z / cblocks, c0 = (blockIdx.z - b * cblocks) * kCh;
    const int tys = blockIdx.y, ty = ty0 + tys, tx0 = blockIdx.x * kTilesX;
    const int c = threadIdx.x & 31;
    const size_t T = (size_t)B * ths * tw;
    const size_t xs = T * Cout;
    for (int t = threadIdx.x >> 5; t < kTilesX; t += 8) {
        const int tx = tx0 + t;
        float o[2][2] = {{0.f, 0.f}, {0.f, 0.f}};
        if (tx < tw && c0 + c < Cout) {
            const size_t tidx = ((size_t)b * ths + tys) * tw + tx;
            const float* m = M + tidx * Cout + c0 + c;
            float v[4][4];
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int j = 0; j < 4; ++j) v[r][j] = m[(size_t)(4 * r + j) * xs];
            // A^T v A,  A^T = [1 1 1 0; 0 1 -1 -1]
            float s[2][4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                s[0][j] = v[0][j] + v[1][j] + v[2][j];
                s[1][j] = v[1][j] - v[2][j] - v[3][j];
            }
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                o[r][0] = s[r][0] + s[r][1] + s[r][2];
                o[r][1] = s[r][1] - s[r][2] - s[r][3];
            }
        }
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            tile[c][r * 2 * kTilesX + 2 * t] = o[r][0];
            tile[c][r * 2 * kTilesX + 2 * t + 1] = o[r][1];
        }
    }
    __syncthreads();
    // 32 channels x 2 rows x 64 columns -> NCHW rows, epilogue fused
    for (int i = threadIdx.x; i < kCh * 2 * 2 * kTilesX; i += 256) {
        const int col = i & (2 * kTilesX - 1), rc = i >> 6;
        const int r = rc & 1, cc = rc >> 1;
        const int yy = 2 * ty + r, xx = 2 * tx0 + col, ch = c0 + cc;
        if (ch >= Cout || yy >= H || xx >= W) continue;
        const size_t idx = (((size_t)b * Cout + ch) * H + yy) * W + xx;
        float v = tile[cc][r * 2 * kTilesX + col];
        v = v * (scale != nullptr ? scale[ch] : 1.f) + (shift != nullptr ? shift[ch] : 0.f);
        if (res != nullptr && res_first) v += res[idx];
        v = act_apply2(v, act, slope);
        if (res != nullptr && !res_first) v += res[idx];
        y[idx] = v;
    }
}

}  // namespace

extern "C" {

int occd_wino_input_transform_nchw(const float* x, float* V, int32_t batch, int32_t Cin, int32_t H, int32_t W,
                                   int32_t ty0, int32_t ths, void* stream) {
    if (!x || !V || batch <= 0 || Cin <= 0 || H <= 0 || W <= 0) return OCCD_EINVAL;
    const int tw = (W + 1) / 2;
    if (ty0 < 0 || ths < 1 || ty0 + ths > (H + 1) / 2) return OCCD_EINVAL;
    const int th = ths;
    const long gz = (long)batch * ((Cin + kCh - 1) / kCh);
    if (gz > 65535 || th > 65535) return OCCD_EINVAL;
    const double T = (double)batch * th * tw;
    occd::ProfScope prof("wino_input", (hipStream_t)stream, 32.0 * T * Cin, 4.0 * ((double)batch * Cin * H * W + 16.0 * T * Cin));
    hipLaunchKernelGGL(wino_input_kernel, dim3((unsigned)((tw + kTilesX - 1) / kTilesX), (unsigned)th, (unsigned)gz), dim3(256),
                       0, (hipStream_t)stream, x, V, batch, Cin, H, W, ths, tw, ty0);
    return occd::check_launch();
}

int occd_wino_output_transform_nchw(const float* M, const float* scale, const float* shift, const float* res, float* y,
                                    int32_t batch, int32_t Cout, int32_t H, int32_t W, int32_t ty0, int32_t ths,
                                    int32_t act, float slope, int32_t res_first, void* stream) {
    if (!M || !y || batch <= 0 || Cout <= 0 || H <= 0 || W <= 0 || act < 0 || act > 3) return OCCD_EINVAL;
    const int tw = (W + 1) / 2;
    if (ty0 < 0 || ths < 1 || ty0 + ths > (H + 1) / 2) return OCCD_EINVAL;
    const int th = ths;
    const long gz = (long)batch * ((Cout + kCh - 1) / kCh);
    if (gz > 65535 || th > 65535) return OCCD_EINVAL;
    const double T = (double)batch * th * tw;
    occd::ProfScope prof("wino_output", (hipStream_t)stream, 24.0 * T * Cout,
                         4.0 * (16.0 * T * Cout + (double)batch * Cout * H * W * (1 + (res != nullptr))));
    hipLaunchKernelGGL(wino_output_kernel, dim3((unsigned)((tw + kTilesX - 1) / kTilesX), (unsigned)th, (unsigned)gz), dim3(256),
                       0, (hipStream_t)stream, M, scale, shift, res, y, batch, Cout, H, W, ths, tw, ty0, act, slope, res_first);
    return occd::check_launch();
}

}  // extern "C"
