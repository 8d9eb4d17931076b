// K11 -- pointwise (1x1) convolution on NCHW feature maps as a GEMM on the fp32 matrix pipe, with the whole
// EfficientNet / decoder epilogue fused:
//
//   y[b][co][n] = act( sum_ci (w[co][ci] * scale[co]) * (x[b][ci][n] * gate[b][ci]) + shift[co] ) (+ res[b][co][n])
//
// i.e. conv1x1 + BatchNorm(eval) + Swish / none (+ the MBConv skip add), with the squeeze-excite gate of the
// PRECEDING depthwise stage applied to the input channels on the fly (x * sigmoid(...) never exists in memory).
// GEMM view per image: C (Cout x N) = W (Cout x Cin) . X (Cin x N), N = H*W pixels -- NCHW is already the row-major
// B operand, so nothing is transposed.  Instruction: v_mfma_f32_32x32x2_f32 (exact fp32).
//   * D rows = couts, D columns = pixels: a lane owns ONE pixel column (lane & 31) and 16 couts, so every store
//     instruction writes 32 consecutive pixels (128 B) of a cout row and residual reads are the same shape;
//   * A (weights): pre-packed per layer in fragment order [k/8][cout/32][lane][4] (BatchNorm scale folded in): one
//     global_load_dwordx4 per lane per 8-channel step, L2 resident;
//   * B (activations): read straight from global memory -- the K index is permuted so lane half h owns channels
//     8c + 4h .. 8c + 4h + 3, and for each of them the half-wave reads 32 consecutive pixels (128 B).  A wave's B
//     fragment is reused by its MT cout blocks and its A fragment by its NT pixel blocks; nothing is shared between
//     waves, so there is no LDS and no barrier: a wave is an independent (MT*32) x (NT*32) register-blocked GEMM and
//     the 4 waves of a workgroup sit side by side (WM along couts x WN along pixels) only for L1 locality;
//   * fragments of step c+1 are loaded before the MFMAs of step c.
// Most of these layers are HBM-bound (algorithmic bytes 4 * B * N * (Cin + Cout (+ Cout for the residual))); the
// wide ones (Cin, Cout >= 640) are MFMA-bound (2 * B * N * Cin8 * Cout32 FLOP).
//
// Reference semantics replaced: the conv_pw / conv_pwl / conv_head 1x1 convolutions + BatchNorm + Swish + SE gate
// + skip of the geffnet EfficientNet blocks behind occdepth/models/unet2d.py:175-190, and the `resize_output_1_s`
// 1x1 convolutions of DecoderBN (unet2d.py:137-165), DepthNet.depth_pred (flosp_depth.py:225-227).
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr long kSplitKBelowWgs = 400;    // the split-K variants (K11s) take over when K11 would launch fewer workgroups

struct PwP {
    const float* x;
    const float* wpk;
    const float* shift;
    const float* gate;
    const float* res;
    float* y;
    int Cin, Cout, kchunks, mblocks;
    long N;
    int act;
    float slope;
    int ntiles, mtiles;
    int out_cs;                 // NHWC output: floats per pixel row
};

__device__ __forceinline__ float pw_act(float v, int act, float slope) {
    if (act == 1) return fmaxf(v, 0.f);
    if (act == 2) return occd::swish_fast(v);
    if (act == 3) return v > 0.f ? v : v * slope;
    return v;
}

// (second launch bound = workgroups per CU = waves per SIMD: 128 accumulator registers leave room for 2, 64 for 3)
// NHWC: the output is written pixel-major, y[b][n][co] (rows of out_cs floats) -- what the 2D->3D lift gathers from.
// The MFMA operands are swapped for it (D rows = pixels, D columns = couts), so a store instruction still writes 32
// consecutive floats (128 B): 32 couts of one pixel instead of 32 pixels of one cout.
template <int MT, int NT, int WM, int WN, bool GATE, bool NHWC>
__global__ void __launch_bounds__(256, (MT * NT > 4 ? 2 : 3)) pw_gemm_kernel(const PwP p) {
    static_assert(WM * WN == 4, "4 waves per workgroup");
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave / WN, wn = wave - wm * WN;
    const int li = lane & 31, kk = lane >> 5;
    const int b = blockIdx.z;
    // consecutive workgroups walk the cout tiles of one pixel tile first: they re-read the same X columns from L2
    const int mt_wg = blockIdx.x % p.mtiles, nt_wg = blockIdx.x / p.mtiles;
    const int mb0 = (mt_wg * WM + wm) * MT;                       // first 32-cout block of this wave
    const long n0 = ((long)nt_wg * WN + wn) * NT * 32;            // first pixel of this wave
    if (mb0 >= p.mblocks || n0 >= p.N) return;                    // (no barriers in this kernel)

    const float* const xb = p.x + (size_t)b * p.Cin * p.N;
    const float* const gb = GATE ? p.gate + (size_t)b * p.Cin : nullptr;

    // pixel columns of this lane and their validity; A-fragment offsets of the owned cout blocks
    long col[NT];
    bool colok[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        col[nt] = n0 + nt * 32 + li;
        colok[nt] = col[nt] < p.N;
        col[nt] = colok[nt] ? col[nt] : p.N - 1;
    }
    int wofs[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) wofs[mt] = min(mb0 + mt, p.mblocks - 1) * 256 + lane * 4;
    const size_t w_step = (size_t)p.mblocks * 256;

    f32x16 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;

    auto load_a = [&](int c, f32x4* a) {
        const float* wp = p.wpk + (size_t)c * w_step;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) a[mt] = *(const f32x4*)(wp + wofs[mt]);
    };
    // Every load is unconditional (clamped address); the channel tail is zeroed by a bit mask on the loaded value (a load
    // guarded by a runtime condition makes hipcc branch around it and wait for each one separately).  Loads and their
    // post-processing (gate multiply, tail mask) are separate steps so that the raw loads of step c+1 can be issued
    // ABOVE the MFMAs of step c and touched only below them (sched_barrier pins both sides).
    auto load_b = [&](int c, f32x4* braw, f32x4& g) {
        const int k0 = c * 8 + kk * 4;
        if (GATE) {
#pragma unroll
            for (int q = 0; q < 4; ++q) g[q] = gb[min(k0 + q, p.Cin - 1)];
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float* row = xb + (size_t)min(k0 + q, p.Cin - 1) * p.N;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) braw[nt][q] = row[col[nt]];
        }
    };
    auto finish_b = [&](int c, const f32x4* braw, const f32x4& g, f32x4* bf) {
        const int k0 = c * 8 + kk * 4;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uint32_t keep = 0u - (uint32_t)(k0 + q < p.Cin);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const float v = GATE ? braw[nt][q] * g[q] : braw[nt][q];
                bf[nt][q] = __uint_as_float(__float_as_uint(v) & keep);
            }
        }
    };

    f32x4 a_cur[MT], b_cur[NT], b_raw[NT], g_raw = {1.f, 1.f, 1.f, 1.f};
    load_a(0, a_cur);
    load_b(0, b_raw, g_raw);
    finish_b(0, b_raw, g_raw, b_cur);
    for (int c = 0; c < p.kchunks; ++c) {
        f32x4 a_nxt[MT];
        const int cn = c + 1 < p.kchunks ? c + 1 : c;              // (the last step re-reads itself: no branch)
        load_a(cn, a_nxt);
        load_b(cn, b_raw, g_raw);
        __builtin_amdgcn_sched_barrier(0);                         // the next step's loads stay ABOVE this step's MFMAs
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    acc[mt][nt] = NHWC ? __builtin_amdgcn_mfma_f32_32x32x2f32(b_cur[nt][q], a_cur[mt][q], acc[mt][nt], 0, 0, 0)
                                       : __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[mt][q], b_cur[nt][q], acc[mt][nt], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) a_cur[mt] = a_nxt[mt];
        finish_b(cn, b_raw, g_raw, b_cur);
    }

    if (NHWC) {
        // register r of tile (mt, nt) is pixel n0 + 32 nt + 8 (r >> 2) + 4 kk + (r & 3) at cout 32 (mb0 + mt) + li
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int co = (mb0 + mt) * 32 + li;
            if (mb0 + mt >= p.mblocks || co >= p.out_cs) continue;
            const bool real = co < p.Cout;
            const float sh = real && p.shift != nullptr ? p.shift[co] : 0.f;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const long n = n0 + nt * 32 + 8 * (r >> 2) + 4 * kk + (r & 3);
                    if (n >= p.N) continue;
                    const float v = pw_act(acc[mt][nt][r] + sh, p.act, p.slope);
                    p.y[((size_t)b * p.N + n) * p.out_cs + co] = real ? v : 0.f;     // (channel pad written as zeros)
                }
        }
        return;
    }
    // ---------------- epilogue: register r of tile (mt, nt) is cout 32 (mb0 + mt) + 8 (r >> 2) + 4 kk + (r & 3)
    // at pixel n0 + 32 nt + li.
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        if (mb0 + mt >= p.mblocks) continue;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = (mb0 + mt) * 32 + 8 * (r >> 2) + 4 * kk + (r & 3);
            if (co >= p.Cout) continue;
            const float sh = p.shift != nullptr ? p.shift[co] : 0.f;
            const size_t row = ((size_t)b * p.Cout + co) * p.N;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                if (!colok[nt]) continue;
                float v = pw_act(acc[mt][nt][r] + sh, p.act, p.slope);
                if (p.res != nullptr) v += p.res[row + col[nt]];
                p.y[row + col[nt]] = v;
            }
        }
    }
}

// K11s -- the same GEMM for the low-resolution EfficientNet stages (1/16 and 1/32: < 2000 pixels per view, K up to 3840,
// up to 3840 couts).  There are too few pixel tiles to fill 256 CUs and a wave walking K = 3840 alone is a 50 us serial
// chain, so the KS waves of a workgroup share ONE (MT*32) x (NT*32) output tile and split K between them (wave w takes the
// 8-channel chunks w, w + KS, ...); the partial accumulators meet in LDS once, after the loop, and the reduction doubles
// as a redistribution: wave w sums accumulator registers w, w + KS, ... of all KS partials (fixed order: deterministic)
// and runs the epilogue for them, so the stores are spread over all waves and are the same 128-byte rows as above.
template <int MT, int NT, int KS, bool GATE, bool NHWC>
__global__ void __launch_bounds__(KS * 64) pw_gemm_splitk_kernel(const PwP p) {
    __shared__ float red[KS * MT * NT * 16 * 64];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int li = lane & 31, kk = lane >> 5;
    const int b = blockIdx.z;
    const int mt_wg = blockIdx.x % p.mtiles, nt_wg = blockIdx.x / p.mtiles;
    const int mb0 = mt_wg * MT;
    const long n0 = (long)nt_wg * NT * 32;

    const float* const xb = p.x + (size_t)b * p.Cin * p.N;
    const float* const gb = GATE ? p.gate + (size_t)b * p.Cin : nullptr;

    long col[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) col[nt] = min(n0 + nt * 32 + li, p.N - 1);
    int wofs[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) wofs[mt] = min(mb0 + mt, p.mblocks - 1) * 256 + lane * 4;
    const size_t w_step = (size_t)p.mblocks * 256;

    f32x16 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;

    auto load_a = [&](int c, f32x4* a) {
        const float* wp = p.wpk + (size_t)c * w_step;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) a[mt] = *(const f32x4*)(wp + wofs[mt]);
    };
    auto load_b = [&](int c, f32x4* braw, f32x4& g) {
        const int k0 = c * 8 + kk * 4;
        if (GATE) {
#pragma unroll
            for (int q = 0; q < 4; ++q) g[q] = gb[min(k0 + q, p.Cin - 1)];
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float* row = xb + (size_t)min(k0 + q, p.Cin - 1) * p.N;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) braw[nt][q] = row[col[nt]];
        }
    };
    auto finish_b = [&](int c, const f32x4* braw, const f32x4& g, f32x4* bf) {
        const int k0 = c * 8 + kk * 4;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uint32_t keep = 0u - (uint32_t)(k0 + q < p.Cin);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const float v = GATE ? braw[nt][q] * g[q] : braw[nt][q];
                bf[nt][q] = __uint_as_float(__float_as_uint(v) & keep);
            }
        }
    };

    // Software pipeline of depth PF: the raw loads of step j + PF are issued above the MFMAs of step j (a step is only
    // 4 * MT * NT MFMAs, far shorter than an L2 round trip).  Steps past the wave's last chunk load a clamped address
    // and are zeroed by finish_b's channel mask (chunk index >= kchunks), so the loop body is branch-free.
    {
        constexpr int PF = 3;
        f32x4 a_r[PF][MT], b_r[PF][NT], g_r[PF];
        const int last = p.kchunks - 1;
        const int nsteps = p.kchunks > wave ? (p.kchunks - wave + KS - 1) / KS : 0;
#pragma unroll
        for (int u = 0; u < PF; ++u) {
            g_r[u] = f32x4{1.f, 1.f, 1.f, 1.f};
            load_a(min(wave + u * KS, last), a_r[u]);
            load_b(min(wave + u * KS, last), b_r[u], g_r[u]);
        }
        for (int j0 = 0; j0 < nsteps; j0 += PF) {
#pragma unroll
            for (int u = 0; u < PF; ++u) {
                const int c = wave + (j0 + u) * KS;                 // (unclamped: finish_b zeroes chunks >= kchunks)
                f32x4 a_cur[MT], b_cur[NT];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) a_cur[mt] = a_r[u][mt];
                finish_b(c, b_r[u], g_r[u], b_cur);
                load_a(min(c + PF * KS, last), a_r[u]);
                load_b(min(c + PF * KS, last), b_r[u], g_r[u]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt)
                            acc[mt][nt] = NHWC ? __builtin_amdgcn_mfma_f32_32x32x2f32(b_cur[nt][q], a_cur[mt][q], acc[mt][nt], 0, 0, 0)
                                               : __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[mt][q], b_cur[nt][q], acc[mt][nt], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }

    // partials -> LDS: red[wave][slice = (mt * NT + nt) * 16 + r][lane]
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                red[((wave * MT * NT + mt * NT + nt) * 16 + r) * 64 + lane] = acc[mt][nt][r];
    __syncthreads();
    // slice s = register r of tile (mt, nt): cout 32 (mb0 + mt) + 8 (r >> 2) + 4 kk + (r & 3) at pixel n0 + 32 nt + li
    // (NHWC, operands swapped: pixel n0 + 32 nt + 8 (r >> 2) + 4 kk + (r & 3) at cout 32 (mb0 + mt) + li)
    for (int s = wave; s < MT * NT * 16; s += KS) {
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < KS; ++w) v += red[(w * MT * NT * 16 + s) * 64 + lane];
        const int r = s & 15, t = s >> 4, mt = t / NT, nt = t - mt * NT;
        const int sub = 8 * (r >> 2) + 4 * kk + (r & 3);
        if (NHWC) {
            const int co = (mb0 + mt) * 32 + li;
            const long n = n0 + nt * 32 + sub;
            if (mb0 + mt >= p.mblocks || co >= p.out_cs || n >= p.N) continue;
            const bool real = co < p.Cout;
            v = pw_act(v + (real && p.shift != nullptr ? p.shift[co] : 0.f), p.act, p.slope);
            p.y[((size_t)b * p.N + n) * p.out_cs + co] = real ? v : 0.f;             // (channel pad written as zeros)
        } else {
            const int co = (mb0 + mt) * 32 + sub;
            const long n = n0 + nt * 32 + li;
            if (mb0 + mt >= p.mblocks || co >= p.Cout || n >= p.N) continue;
            const size_t o = ((size_t)b * p.Cout + co) * p.N + n;
            v = pw_act(v + (p.shift != nullptr ? p.shift[co] : 0.f), p.act, p.slope);
            if (p.res != nullptr) v += p.res[o];
            p.y[o] = v;
        }
    }
}

// wpk[k/8][cout/32][lane][4]: cout = blk*32 + (lane & 31), cin = chunk*8 + (lane >> 5)*4 + q, value w[cout][cin] * scale[cout]
__global__ void pw_pack_kernel(const float* __restrict__ w, const float* __restrict__ scale, float* __restrict__ wpk,
                               int cout, int cin, int mblocks, long total) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int q = i & 3, lane = (i >> 2) & 63;
    const long t = i >> 8;
    const int blk = t % mblocks, chunk = (int)(t / mblocks);
    const int co = blk * 32 + (lane & 31), ci = chunk * 8 + (lane >> 5) * 4 + q;
    float v = 0.f;
    if (co < cout && ci < cin) v = w[(size_t)co * cin + ci] * (scale != nullptr ? scale[co] : 1.f);
    wpk[i] = v;
}

template <int MT, int NT, int WM, int WN>
int launch_pw(PwP& p, int batch, hipStream_t st) {
    const bool gate = p.gate != nullptr;
    p.mtiles = (p.mblocks + MT * WM - 1) / (MT * WM);
    p.ntiles = (int)((p.N + (long)NT * WN * 32 - 1) / ((long)NT * WN * 32));
    const long gx = (long)p.mtiles * p.ntiles;
    if (gx > 0x7fffffffL || batch > 65535) return OCCD_EINVAL;
    const dim3 grid((unsigned)gx, 1, (unsigned)batch);
    if (p.out_cs > 0) hipLaunchKernelGGL((pw_gemm_kernel<MT, NT, WM, WN, false, true>), grid, dim3(256), 0, st, p);
    else if (gate) hipLaunchKernelGGL((pw_gemm_kernel<MT, NT, WM, WN, true, false>), grid, dim3(256), 0, st, p);
    else hipLaunchKernelGGL((pw_gemm_kernel<MT, NT, WM, WN, false, false>), grid, dim3(256), 0, st, p);
    return occd::check_launch();
}

template <int MT, int NT, int KS>
int launch_pw_splitk(PwP& p, int batch, hipStream_t st) {
    p.mtiles = (p.mblocks + MT - 1) / MT;
    p.ntiles = (int)((p.N + (long)NT * 32 - 1) / ((long)NT * 32));
    const long gx = (long)p.mtiles * p.ntiles;
    if (gx > 0x7fffffffL || batch > 65535) return OCCD_EINVAL;
    const dim3 grid((unsigned)gx, 1, (unsigned)batch);
    if (p.out_cs > 0) hipLaunchKernelGGL((pw_gemm_splitk_kernel<MT, NT, KS, false, true>), grid, dim3(KS * 64), 0, st, p);
    else if (p.gate != nullptr) hipLaunchKernelGGL((pw_gemm_splitk_kernel<MT, NT, KS, true, false>), grid, dim3(KS * 64), 0, st, p);
    else hipLaunchKernelGGL((pw_gemm_splitk_kernel<MT, NT, KS, false, false>), grid, dim3(KS * 64), 0, st, p);
    return occd::check_launch();
}

}  // namespace

extern "C" {

int64_t occd_pw_packed_floats(int32_t cout, int32_t cin) {
    if (cout < 1 || cin < 1) return OCCD_EINVAL;
    return (int64_t)((cin + 7) / 8) * ((cout + 31) / 32) * 256;
}

int occd_pw_pack_weights(const float* w, const float* scale, float* wpk, int32_t cout, int32_t cin, void* stream) {
    if (w == nullptr || wpk == nullptr || cout < 1 || cin < 1) return OCCD_EINVAL;
    const long total = occd_pw_packed_floats(cout, cin);
    hipLaunchKernelGGL(pw_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w, scale,
                       wpk, cout, cin, (cout + 31) / 32, total);
    return occd::check_launch();
}

int occd_pw_conv_fwd(const occd_pw_args* a, void* stream) {
    if (a == nullptr || a->x == nullptr || a->wpk == nullptr || a->y == nullptr) return OCCD_EINVAL;
    if (a->batch < 1 || a->cin < 1 || a->cout < 1 || a->N < 1 || a->act < 0 || a->act > 3) return OCCD_EINVAL;
    if (a->out_nhwc_cs != 0 && (a->out_nhwc_cs < a->cout || a->gate != nullptr || a->res != nullptr)) return OCCD_EINVAL;
    PwP p{};
    p.out_cs = a->out_nhwc_cs;
    p.x = a->x; p.wpk = a->wpk; p.shift = a->shift; p.gate = a->gate; p.res = a->res; p.y = a->y;
    p.Cin = a->cin; p.Cout = a->cout; p.N = a->N; p.act = a->act; p.slope = a->slope;
    p.kchunks = (a->cin + 7) / 8;
    p.mblocks = (a->cout + 31) / 32;
    hipStream_t st = (hipStream_t)stream;
    occd::ProfScope prof("pw_conv", st, 2.0 * a->batch * (double)a->N * p.kchunks * 8 * p.mblocks * 32,
                         4.0 * a->batch * (double)a->N * (a->cin + a->cout * (a->res != nullptr ? 2.0 : 1.0)));
    // Variant table (tile_hint 1..6 = <MT, NT, WM, WN>); the automatic choice follows tools/bench_pw.py on the B7 / decoder
    // shapes at two views (profiles/r02_pw_gemm_layers.txt): narrow layers take one cout block per wave and 128 pixels
    // (X is re-read once per cout block, from L2, but the grid is large); layers with a long K and few couts take the
    // 64 x 64 wave tile with the 4 waves along the couts.
    const int mb = p.mblocks;
    int hint = a->tile_hint;
    if (hint == 0) {
        hint = mb == 1 ? 1 : mb == 2 ? (a->cin >= 192 ? 6 : 2) : mb == 3 ? 6 : (a->cin <= 256 ? 1 : 6);
        // K11s (hints 7..12 = <MT, NT, KS>) when the streaming variant would leave CUs without a workgroup and K is long
        // enough to split: the largest register tile that still gives every CU a workgroup
        static const int wg_m[7] = {0, 1, 2, 4, 8, 16, 8}, wg_n[7] = {0, 512, 512, 256, 128, 64, 64};   // couts/32, pixels per WG
        const long wgs = (long)((mb + wg_m[hint] - 1) / wg_m[hint]) * ((a->N + wg_n[hint] - 1) / wg_n[hint]) * a->batch;
        if (wgs < kSplitKBelowWgs && p.kchunks >= 36) {
            // measured on the B7 project convolutions (profiles/r02_pw_gemm_layers.txt): the 64 x 64 tile wins down to
            // ~100 workgroups (operand re-reads from L2 cost more than idle CUs), then 64 x 32 and 32 x 32 with 8 waves on K
            auto tiles = [&](int mt, int nt) { return (long)((mb + mt - 1) / mt) * ((a->N + 32 * nt - 1) / (32 * nt)) * a->batch; };
            hint = tiles(2, 2) >= 128 ? 11 : tiles(2, 1) >= 128 ? 9 : 8;
        }
    }
    switch (hint) {
    case 7: return launch_pw_splitk<1, 1, 4>(p, a->batch, st);
    case 8: return launch_pw_splitk<1, 1, 8>(p, a->batch, st);
    case 9: return launch_pw_splitk<2, 1, 8>(p, a->batch, st);
    case 10: return launch_pw_splitk<1, 2, 8>(p, a->batch, st);
    case 11: return launch_pw_splitk<2, 2, 4>(p, a->batch, st);
    case 12: return launch_pw_splitk<2, 1, 4>(p, a->batch, st);
    case 1: return launch_pw<1, 4, 1, 4>(p, a->batch, st);
    case 2: return launch_pw<2, 4, 1, 4>(p, a->batch, st);
    case 3: return launch_pw<4, 2, 1, 4>(p, a->batch, st);
    case 4: return launch_pw<4, 2, 2, 2>(p, a->batch, st);
    case 5: return launch_pw<4, 2, 4, 1>(p, a->batch, st);
    case 6: return launch_pw<2, 2, 4, 1>(p, a->batch, st);
    default: return OCCD_EINVAL;
    }
}

}  // extern "C"
