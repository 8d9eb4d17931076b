// K15 -- pointwise (1x1x1) convolution / row GEMM for channels-last voxel rows on v_mfma_f32_16x16x4_f32 (VERDICT r2 item 3:
// "the 512>512 / 256>512 1x1x1 CRP GEMMs" at 34-40 TF/s on the generic implicit-GEMM kernel).
//
//   out[r][n] = act_out( sum_k act_in(A[r][k]) * W[k][n] + bias[n] (+ res[r][n]) )        A, out, res: Vox rows; W: [K][N]
//
// K2 stages an input slab in LDS and pays two barriers per (tap, 32-channel chunk); a 1x1x1 convolution has ONE tap, so
// on K2 every 32 channels of K cost a stage-sync-16-MFMA-sync round and the kernel is barrier-bound.  Here the data operand
// never touches LDS: the reduction runs transposed (D^T = W^T . A^T, the idiom of K14 -- csrc/bneck3d.hip): a lane's four k
// values of a 16-channel super-step are four CONSECUTIVE channels, so the B operand is one 16-byte global load per lane (64
// contiguous bytes per row), and a D fragment is a float4 of consecutive output channels of one row: the store format.
// Only the weights go through LDS (fragment order, double buffered, 32 k x 128 n per stage).
// A workgroup = 4 waves x 16 rows x 128 columns.  Reference call sites: occdepth/models/CRP3D.py:54-97 (context_prior_logits,
// the bmm, resize), DDR.py:33,42 (conv1 / conv5 of the strided bottlenecks).
#include "common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

struct RowsP {
    const float* a; const float* w; const float* bias; const float* res; float* out;
    long rows;
    int K, N;                 // K multiple of 16, N multiple of 16
    int a_cs, a_coff, out_cs, out_coff, res_cs, res_coff, w_stride;
    int act_in, act_out;
    int ntiles;               // ceil(N / 128)
};

__device__ __forceinline__ f32x4 act4(f32x4 v, int act) {
    if (act == OCCD_ACT_RELU) {
        v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
    } else if (act == OCCD_ACT_SIGMOID) {      // the same expression as K2's apply_act: bit-identical operands
        v.x = 1.f / (1.f + expf(-v.x)); v.y = 1.f / (1.f + expf(-v.y));
        v.z = 1.f / (1.f + expf(-v.z)); v.w = 1.f / (1.f + expf(-v.w));
    }
    return v;
}

constexpr int kStageK = 32;                    // k per weight stage (2 super-steps)
constexpr int kTileN = 128;                    // columns per workgroup (8 MFMA row blocks of the transposed product)

// Weights arrive PRE-PACKED in fragment order (hip.pack_rows_gemm / a torch permute): for every (32-k stage kt, 128-column
// tile nt) one 16 KB record [t (2)][m (8)][lane 64][e 4] with element = W[32 kt + 16 t + 4 (lane >> 4) + e][128 nt + 16 m +
// (lane & 15)] (zero padded), so a stage is 4 coalesced 16-byte loads per thread, fetched one stage ahead into registers.
// (The first version gathered the fragments from the dense [K][N] matrix with 4-byte loads inside the K loop: 114 us for
// 512>512 on 4096 rows against K2's 63 -- the staging, not the MFMAs, was the kernel.)
__global__ void __launch_bounds__(256) rows_gemm_kernel(const RowsP p) {
    __shared__ __attribute__((aligned(16))) f32x4 wf[2][kStageK * kTileN / 4];      // 2 x 16 KB
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 15, g = lane >> 4;
    const int nt = blockIdx.x % p.ntiles;
    const long rt = blockIdx.x / p.ntiles;
    const int n0 = nt * kTileN;
    const long row = rt * 64 + wave * 16 + j;
    const bool live = row < p.rows;
    const float* arow = p.a + (size_t)(live ? row : 0) * p.a_cs + p.a_coff + 4 * g;
    const int mlim = min(8, (p.N - n0 + 15) >> 4);         // 16-column blocks of this tile that exist
    const int stages = (p.K + kStageK - 1) / kStageK;
    const f32x4* wsrc = (const f32x4*)p.w + (size_t)nt * 1024 + tid;      // record (kt, nt) at ((kt * ntiles + nt) * 1024) float4
    const size_t wstep = (size_t)p.ntiles * 1024;
    f32x4 acc[8];
#pragma unroll
    for (int m = 0; m < 8; ++m) acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 wreg[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) wf[0][tid + 256 * u] = wsrc[256 * u];
    // data rows: one stage ahead in registers too (the load latency would otherwise be paid before every 64 MFMAs)
    auto load_data = [&](int k0, f32x4 (&d)[2]) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const bool on = live && k0 + 16 * t < p.K;
            d[t] = *(const f32x4*)(arow + (on ? k0 + 16 * t : 0));
            if (!on) d[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };
    f32x4 dn[2];
    load_data(0, dn);
    __syncthreads();
    for (int s = 0; s < stages; ++s) {
        const bool more = s + 1 < stages;
        f32x4 d[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) d[t] = act4(dn[t], p.act_in);
        if (p.act_in != OCCD_ACT_NONE) {                   // act(0) of the masked lanes must stay 0 (sigmoid(0) = 0.5)
#pragma unroll
            for (int t = 0; t < 2; ++t)
                if (!live || s * kStageK + 16 * t >= p.K) d[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        if (more) {
#pragma unroll
            for (int u = 0; u < 4; ++u) wreg[u] = wsrc[(size_t)(s + 1) * wstep + 256 * u];
            load_data((s + 1) * kStageK, dn);
        }
        const f32x4* w = wf[s & 1];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int m = 0; m < 8; ++m) {
                if (m >= mlim) continue;                   // (workgroup-uniform)
                const f32x4 wv = w[(t * 8 + m) * 64 + lane];
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[e], d[t][e], acc[m], 0, 0, 0);
            }
        if (more) {
#pragma unroll
            for (int u = 0; u < 4; ++u) wf[(s + 1) & 1][tid + 256 * u] = wreg[u];
        }
        __syncthreads();
    }
    if (!live) return;
#pragma unroll
    for (int m = 0; m < 8; ++m) {
        const int n = n0 + 16 * m + 4 * g;
        if (n >= p.N) continue;
        f32x4 v = acc[m];
        if (p.bias) v += *(const f32x4*)(p.bias + n);
        if (p.act_out == OCCD_ACT_RELU_PRE) v = act4(v, OCCD_ACT_RELU);
        if (p.res) v += *(const f32x4*)(p.res + (size_t)row * p.res_cs + p.res_coff + n);
        if (p.act_out == OCCD_ACT_RELU) v = act4(v, OCCD_ACT_RELU);
        *(f32x4*)(p.out + (size_t)row * p.out_cs + p.out_coff + n) = v;
    }
}

// dense [K][N] (row stride ws) -> the fragment records above
__global__ void __launch_bounds__(256) rows_gemm_pack_kernel(const float* __restrict__ w, float* __restrict__ out, int K, int N,
                                                             int ws, int ntiles, long total) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int e = i & 3, lane = (i >> 2) & 63, m = (i >> 8) & 7, t = (i >> 11) & 1;
    const long rec = i >> 12;
    const int nt = (int)(rec % ntiles), kt = (int)(rec / ntiles);
    const int k = 32 * kt + 16 * t + 4 * (lane >> 4) + e, n = 128 * nt + 16 * m + (lane & 15);
    out[i] = (k < K && n < N) ? w[(size_t)k * ws + n] : 0.f;
}

}  // namespace

extern "C" int occd_rows_gemm_fwd(const occd_rows_gemm_args* a, void* stream) {
    if (!a || !a->a || !a->w || !a->out || a->rows <= 0) return OCCD_EINVAL;
    if (a->K <= 0 || (a->K & 15) || a->N <= 0 || (a->N & 3)) return OCCD_EINVAL;
    if ((a->a_cs & 3) || (a->a_coff & 3) || a->a_coff + a->K > a->a_cs) return OCCD_EINVAL;
    if ((a->out_cs & 3) || (a->out_coff & 3) || a->out_coff + a->N > a->out_cs) return OCCD_EINVAL;
    if (a->res && ((a->res_cs & 3) || (a->res_coff & 3) || a->res_coff + a->N > a->res_cs)) return OCCD_EINVAL;
    if (a->act_in != OCCD_ACT_NONE && a->act_in != OCCD_ACT_RELU && a->act_in != OCCD_ACT_SIGMOID) return OCCD_EINVAL;
    if (a->act_out != OCCD_ACT_NONE && a->act_out != OCCD_ACT_RELU && a->act_out != OCCD_ACT_RELU_PRE) return OCCD_EINVAL;
    if ((reinterpret_cast<uintptr_t>(a->a) | reinterpret_cast<uintptr_t>(a->out) | reinterpret_cast<uintptr_t>(a->res) |
         reinterpret_cast<uintptr_t>(a->bias) | reinterpret_cast<uintptr_t>(a->w)) & 15)
        return OCCD_EINVAL;
    RowsP p;
    p.a = a->a; p.w = a->w; p.bias = a->bias; p.res = a->res; p.out = a->out;
    p.rows = a->rows; p.K = a->K; p.N = a->N;
    p.a_cs = a->a_cs; p.a_coff = a->a_coff; p.out_cs = a->out_cs; p.out_coff = a->out_coff;
    p.res_cs = a->res_cs; p.res_coff = a->res_coff; p.w_stride = a->w_stride;
    p.act_in = a->act_in; p.act_out = a->act_out;
    p.ntiles = (a->N + kTileN - 1) / kTileN;
    const long rtiles = (a->rows + 63) / 64;
    if (rtiles * p.ntiles >= (1L << 31)) return OCCD_EINVAL;
    occd::ProfScope prof("rows_gemm", (hipStream_t)stream, 2.0 * a->rows * a->K * a->N,
                         4.0 * (a->rows * ((double)a->K + a->N * (a->res ? 2.0 : 1.0)) + (double)a->K * a->N));
    hipLaunchKernelGGL(rows_gemm_kernel, dim3((unsigned)(rtiles * p.ntiles)), dim3(256), 0, (hipStream_t)stream, p);
    return occd::check_launch();
}

extern "C" int64_t occd_rows_gemm_packed_floats(int32_t K, int32_t N) {
    if (K <= 0 || N <= 0) return OCCD_EINVAL;
    return (int64_t)((K + 31) / 32) * ((N + 127) / 128) * 4096;
}

extern "C" int occd_rows_gemm_pack(const float* w, float* wpk, int32_t K, int32_t N, int32_t w_stride, void* stream) {
    if (!w || !wpk || K <= 0 || N <= 0 || w_stride < N) return OCCD_EINVAL;
    const int ntiles = (N + 127) / 128;
    const long total = occd_rows_gemm_packed_floats(K, N);
    occd::ProfScope prof("rows_gemm_pack", (hipStream_t)stream, 0.0, 8.0 * total);
    hipLaunchKernelGGL(rows_gemm_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w, wpk,
                       K, N, w_stride, ntiles, total);
    return occd::check_launch();
}
