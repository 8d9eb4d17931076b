// Training-step loss statistics and SSC confusion counts (row N4) for gfx950.
//
// The reference evaluates its scene-completion losses with ~100 full-tensor passes over the (B, C, X, Y, Z)
// logits: a softmax per loss, a 20-class python loop (ssc_loss.py:44-87) and a 64-frustum python loop
// (OccDepth.py:487-521).  Every one of those losses is a function of a few hundred SUMS over voxels:
//     P[c] = sum_{t != 255} p_c      N[c] = sum_{t == c} p_c      T[c] = #{t == c}      M = #{t != 255}
//     CEnum = sum w_t * (-log p_t)   CEden = sum w_t              F[f][c] = sum_{mask_f} p_c
// so one HBM-bound pass computes them all (K5), and one more pass turns the gradient with respect to those sums
// into the gradient with respect to the logits (K6).  The sums are accumulated in 64-bit FIXED POINT (Q32 for
// probabilities, Q24 for the cross-entropy terms): integer addition is associative, so the result does not depend
// on the order of the atomics -- bit-identical from run to run, and more accurate than a float32 tree.
#include "common.h"

namespace {

constexpr int kMaxC = 32;
constexpr double kQ32 = 4294967296.0;
constexpr double kQ24 = 16777216.0;
typedef unsigned long long u64;

struct StatsP {
    const float* logits;       // (B, C, S)
    const uint8_t* target;     // (B, S)
    const uint8_t* masks;      // (B, F, S) or null
    const float* weights;      // (C)
    u64* stats;                // 3C + 3 + F*C
    const float* gstats;       // backward: d loss / d (real-valued sums), same layout
    float* grad;               // backward: (B, C, S)
    long total;                // B * S
    int C, S, F, map_occ;
};

__device__ __forceinline__ int map_target(int t, int map_occ) {
    // cascade head: every labelled non-empty class -> 1 (OccDepth.py:413-415)
    return (map_occ && t != 0 && t != 255) ? 1 : t;
}

// probabilities of one voxel; p[] statically indexed (fully unrolled over the class bucket CB >= C, c < C uniform)
template <int CB>
__device__ __forceinline__ void softmax_regs(const float* lp, int C, int S, float (&p)[CB], float& lse) {
    float m = -INFINITY;
#pragma unroll
    for (int c = 0; c < CB; ++c)
        if (c < C) {
            p[c] = lp[(size_t)c * S];
            m = fmaxf(m, p[c]);
        }
    float sum = 0.f;
#pragma unroll
    for (int c = 0; c < CB; ++c)
        if (c < C) {
            p[c] = __expf(p[c] - m);
            sum += p[c];
        }
    const float inv = 1.f / sum;
#pragma unroll
    for (int c = 0; c < CB; ++c)
        if (c < C) p[c] *= inv;
    lse = m + __logf(sum);
}

// Every wave owns a contiguous span of voxels and walks it 64 at a time, so the successive voxels of one LANE are
// 64 apart in memory: two columns over at Z = 32 -- almost always the same frustum and very often the same label.
// The target- and frustum-indexed sums therefore sit in per-lane registers and are flushed to LDS (64-bit integer
// atomics) only when the lane's label / frustum changes; the uniform-address sums never leave registers until the end.
template <int CB>
__global__ void __launch_bounds__(256) ssc_stats_kernel(const StatsP q) {
    extern __shared__ u64 acc[];                       // 3C + 3 + F*C
    const int C = q.C, n_stats = 3 * C + 3 + q.F * C;
    for (int i = threadIdx.x; i < n_stats; i += 256) acc[i] = 0;
    __syncthreads();
    u64 regP[CB], accF[CB];
#pragma unroll
    for (int c = 0; c < CB; ++c) regP[c] = accF[c] = 0;
    u64 regM = 0, regNum = 0, regDen = 0, accN = 0;
    unsigned accT = 0;
    int cur_t = -1, cur_f = -1;
    auto flush_t = [&]() {
        if (cur_t >= 0) {
            atomicAdd(&acc[C + cur_t], accN);
            atomicAdd(&acc[2 * C + cur_t], (u64)accT);
        }
        accN = 0;
        accT = 0;
    };
    auto flush_f = [&]() {
        if (cur_f >= 0) {
            u64* dst = acc + 3 * C + 3 + cur_f * C;
#pragma unroll
            for (int c = 0; c < CB; ++c)
                if (c < C && accF[c]) atomicAdd(&dst[c], accF[c]);
        }
#pragma unroll
        for (int c = 0; c < CB; ++c) accF[c] = 0;
    };
    const long n_waves = (long)gridDim.x * 4;
    const long span = ((q.total + n_waves - 1) / n_waves + 63) & ~63L;
    const long w = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long v_end = min(w * span + span, q.total);
    for (long v = w * span + (threadIdx.x & 63); v < v_end; v += 64) {
        const long b = v / q.S;
        const int s = (int)(v - b * q.S);
        const float* lp = q.logits + (size_t)b * C * q.S + s;
        float p[CB], lse;
        softmax_regs<CB>(lp, C, q.S, p, lse);
        const int t = map_target(q.target[v], q.map_occ);
        if (t != 255) {
            float pt = 0.f, xt = 0.f;
#pragma unroll
            for (int c = 0; c < CB; ++c)
                if (c < C) {
                    regP[c] += (u64)((double)p[c] * kQ32);
                    if (c == t) { pt = p[c]; xt = lp[(size_t)c * q.S]; }
                }
            if (t < C) {
                if (t != cur_t) {
                    flush_t();
                    cur_t = t;
                }
                accN += (u64)((double)pt * kQ32);
                accT += 1;
                const float wt = q.weights != nullptr ? q.weights[t] : 1.f;
                regNum += (u64)((double)(wt * (lse - xt)) * kQ24);
                regDen += (u64)((double)wt * kQ24);
            }
            regM += 1;
        }
        if (q.masks != nullptr) {
            const uint8_t* mp = q.masks + (size_t)b * q.F * q.S + s;
            for (int f = 0; f < q.F; ++f)
                if (mp[(size_t)f * q.S]) {
                    if (f != cur_f) {
                        flush_f();
                        cur_f = f;
                    }
#pragma unroll
                    for (int c = 0; c < CB; ++c)
                        if (c < C) accF[c] += (u64)((double)p[c] * kQ32);
                }
        }
    }
    flush_t();
    flush_f();
#pragma unroll
    for (int c = 0; c < CB; ++c)
        if (c < C && regP[c]) atomicAdd(&acc[c], regP[c]);
    if (regM) atomicAdd(&acc[3 * C], regM);
    if (regNum) atomicAdd(&acc[3 * C + 1], regNum);
    if (regDen) atomicAdd(&acc[3 * C + 2], regDen);
    __syncthreads();
    for (int i = threadIdx.x; i < n_stats; i += 256)
        if (acc[i]) atomicAdd(&q.stats[i], acc[i]);
}

// d loss / d logit_c = p_c (g_c - sum_k p_k g_k) + [t != 255] gCEnum w_t (p_c - [c == t]),
// g_c = [t != 255] (gP[c] + [t == c] gN[c]) + sum_{f : mask_f} gF[f][c]
template <int CB>
__global__ void __launch_bounds__(256) ssc_grad_kernel(const StatsP q) {
    extern __shared__ float gs[];                      // the gradient table, 3C + 3 + F*C floats
    const int C = q.C, n_stats = 3 * C + 3 + q.F * C;
    for (int i = threadIdx.x; i < n_stats; i += 256) gs[i] = q.gstats[i];
    __syncthreads();
    const float g_num = gs[3 * C + 1];
    for (long v = (long)blockIdx.x * 256 + threadIdx.x; v < q.total; v += (long)gridDim.x * 256) {
        const long b = v / q.S;
        const int s = (int)(v - b * q.S);
        const float* lp = q.logits + (size_t)b * C * q.S + s;
        float p[CB], g[CB], lse;
        softmax_regs<CB>(lp, C, q.S, p, lse);
        const int t = map_target(q.target[v], q.map_occ);
        const bool lab = t != 255;
#pragma unroll
        for (int c = 0; c < CB; ++c)
            if (c < C) g[c] = lab ? gs[c] + (c == t ? gs[C + c] : 0.f) : 0.f;
        if (q.masks != nullptr) {
            const uint8_t* mp = q.masks + (size_t)b * q.F * q.S + s;
            for (int f = 0; f < q.F; ++f)
                if (mp[(size_t)f * q.S]) {
                    const float* src = gs + 3 * C + 3 + f * C;
#pragma unroll
                    for (int c = 0; c < CB; ++c)
                        if (c < C) g[c] += src[c];
                }
        }
        float dot = 0.f;
#pragma unroll
        for (int c = 0; c < CB; ++c)
            if (c < C) dot += p[c] * g[c];
        const float ce = (lab && t < C) ? g_num * (q.weights != nullptr ? q.weights[t] : 1.f) : 0.f;
        float* gp = q.grad + (size_t)b * C * q.S + s;
#pragma unroll
        for (int c = 0; c < CB; ++c)
            if (c < C) gp[(size_t)c * q.S] = p[c] * (g[c] - dot) + ce * (p[c] - (c == t ? 1.f : 0.f));
    }
}

// hist[t * C + argmax] += 1 over labelled voxels (first maximum wins, as np.argmax)
__global__ void __launch_bounds__(256) confusion_kernel(const float* logits, const uint8_t* labels,
                                                        const uint8_t* target, long total, int C, int S, u64* hist) {
    extern __shared__ unsigned int h32[];              // C * C
    for (int i = threadIdx.x; i < C * C; i += 256) h32[i] = 0;
    __syncthreads();
    for (long v = (long)blockIdx.x * 256 + threadIdx.x; v < total; v += (long)gridDim.x * 256) {
        const int t = target[v];
        if (t == 255 || t >= C) continue;
        int pred;
        if (labels != nullptr) {
            pred = labels[v];
        } else {
            const long b = v / S;
            const float* lp = logits + (size_t)b * C * S + (v - b * S);
            float best = lp[0];
            pred = 0;
            for (int c = 1; c < C; ++c) {
                const float x = lp[(size_t)c * S];
                if (x > best) { best = x; pred = c; }
            }
        }
        if (pred < C) atomicAdd(&h32[t * C + pred], 1u);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < C * C; i += 256)
        if (h32[i]) atomicAdd(&hist[i], (u64)h32[i]);
}

__global__ void zero_u64_kernel(u64* p, long n) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = 0;
}

int grid_for(long total) {
    long blocks = (total + 255) / 256;
    const long cap = 256 * 4;                          // 4 workgroups per CU: few final flushes, chip still full
    return (int)(blocks < cap ? (blocks > 0 ? blocks : 1) : cap);
}

int check_stats(const float* logits, const uint8_t* target, int64_t batch, int32_t C, int64_t S, int32_t F) {
    if (logits == nullptr || target == nullptr) return OCCD_EINVAL;
    if (batch < 1 || S < 1 || S > 0x7fffffff || C < 1 || C > kMaxC || F < 0 || F > 1024) return OCCD_EINVAL;
    // the per-workgroup copy of the statistics lives in dynamic LDS (64-bit sums): it must fit the 64 KiB a kernel
    // may request without the large-LDS attribute, e.g. C = 20 -> F <= 406 (the reference uses 64 frustums)
    if ((3 * (int64_t)C + 3 + (int64_t)F * C) * (int64_t)sizeof(u64) > 64 * 1024) return OCCD_EINVAL;
    return OCCD_OK;
}

}  // namespace

extern "C" {

int64_t occd_ssc_stats_len(int32_t C, int32_t F) { return 3 * (int64_t)C + 3 + (int64_t)F * C; }

int occd_ssc_loss_stats_fwd(const float* logits, const uint8_t* target, const uint8_t* masks, const float* weights,
                            int64_t* stats, int64_t batch, int32_t C, int64_t S, int32_t F, int32_t map_occ,
                            void* stream) {
    int rc = check_stats(logits, target, batch, C, S, F);
    if (rc != OCCD_OK) return rc;
    if (stats == nullptr || (F > 0 && masks == nullptr)) return OCCD_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    StatsP q{};
    q.logits = logits; q.target = target; q.masks = F > 0 ? masks : nullptr; q.weights = weights;
    q.stats = (u64*)stats; q.total = batch * S; q.C = C; q.S = (int)S; q.F = F; q.map_occ = map_occ;
    const size_t n = (size_t)occd_ssc_stats_len(C, F);
    // zero the accumulators with a KERNEL, not hipMemsetAsync: inside a captured hipGraph (train_graph.py) the memset node of
    // this odd-sized buffer was observed not to take effect on replays (ROCm 7.0: the statistics kept the previous
    // contents of the graph pool's block: garbage counts from the second replay on, NaN losses)
    hipLaunchKernelGGL(zero_u64_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (u64*)stats, (long)n);
    const double bytes = (double)q.total * (4.0 * C + 1 + F);
    occd::ProfScope prof("ssc_loss_stats", st, (double)q.total * C * 8.0, bytes);
    const dim3 grid(grid_for(q.total));
    if (C <= 4) hipLaunchKernelGGL(ssc_stats_kernel<4>, grid, dim3(256), n * sizeof(u64), st, q);
    else if (C <= 12) hipLaunchKernelGGL(ssc_stats_kernel<12>, grid, dim3(256), n * sizeof(u64), st, q);
    else if (C <= 20) hipLaunchKernelGGL(ssc_stats_kernel<20>, grid, dim3(256), n * sizeof(u64), st, q);
    else hipLaunchKernelGGL(ssc_stats_kernel<32>, grid, dim3(256), n * sizeof(u64), st, q);
    return occd::check_launch();
}

int occd_ssc_loss_stats_bwd(const float* logits, const uint8_t* target, const uint8_t* masks, const float* weights,
                            const float* gstats, float* grad, int64_t batch, int32_t C, int64_t S, int32_t F,
                            int32_t map_occ, void* stream) {
    int rc = check_stats(logits, target, batch, C, S, F);
    if (rc != OCCD_OK) return rc;
    if (gstats == nullptr || grad == nullptr || (F > 0 && masks == nullptr)) return OCCD_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    StatsP q{};
    q.logits = logits; q.target = target; q.masks = F > 0 ? masks : nullptr; q.weights = weights;
    q.gstats = gstats; q.grad = grad; q.total = batch * S; q.C = C; q.S = (int)S; q.F = F; q.map_occ = map_occ;
    const size_t n = (size_t)occd_ssc_stats_len(C, F);
    const double bytes = (double)q.total * (8.0 * C + 1 + F);
    occd::ProfScope prof("ssc_loss_grad", st, (double)q.total * C * 12.0, bytes);
    const dim3 grid(grid_for(q.total));
    if (C <= 4) hipLaunchKernelGGL(ssc_grad_kernel<4>, grid, dim3(256), n * sizeof(float), st, q);
    else if (C <= 12) hipLaunchKernelGGL(ssc_grad_kernel<12>, grid, dim3(256), n * sizeof(float), st, q);
    else if (C <= 20) hipLaunchKernelGGL(ssc_grad_kernel<20>, grid, dim3(256), n * sizeof(float), st, q);
    else hipLaunchKernelGGL(ssc_grad_kernel<32>, grid, dim3(256), n * sizeof(float), st, q);
    return occd::check_launch();
}

int occd_ssc_confusion(const float* logits, const uint8_t* labels, const uint8_t* target, int64_t* hist,
                       int64_t batch, int32_t C, int64_t S, void* stream) {
    if ((logits == nullptr) == (labels == nullptr) || target == nullptr || hist == nullptr) return OCCD_EINVAL;
    if (batch < 1 || S < 1 || S > 0x7fffffff || C < 1 || C > kMaxC) return OCCD_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const long total = batch * S;
    occd::ProfScope prof("ssc_confusion", st, 0.0, (double)total * (logits ? 4.0 * C + 1 : 2.0));
    hipLaunchKernelGGL(confusion_kernel, dim3(grid_for(total)), dim3(256), (size_t)C * C * sizeof(unsigned int), st,
                       logits, labels, target, total, C, (int)S, (u64*)hist);
    return occd::check_launch();
}

}  // extern "C"
