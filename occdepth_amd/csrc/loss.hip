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
    const float* logits;       // element (b, c, s) at b * s_b + c * s_c + s * s_v: (B, C, S) planes (s_c = S, s_v = 1) or
                               // channels-last voxel rows (s_c = 1, s_v = cs) -- the 3-D stack's own layout (round 5)
    const uint8_t* target;     // (B, S)
    const uint8_t* masks;      // (B, F, S) or null
    const float* weights;      // (C)
    u64* stats;                // 3C + 3 + F*C
    const float* gstats;       // backward: d loss / d (real-valued sums), same layout
    float* grad;               // backward: (B, C, S)
    long total;                // B * S
    int C, S, F, map_occ;
    long s_b, s_c, s_v;        // logits strides (floats)
    long g_b, g_c, g_v;        // grad strides
    int g_pad;                 // channels-last grad rows: channels [C, g_pad) are written as zeros
};

__device__ __forceinline__ int map_target(int t, int map_occ) {
    // cascade head: every labelled non-empty class -> 1 (OccDepth.py:413-415)
    return (map_occ && t != 0 && t != 255) ? 1 : t;
}

// probabilities of one voxel; p[] statically indexed (fully unrolled over the class bucket CB >= C, c < C uniform)
// x[] keeps the raw logits (the cross-entropy term needs x_t without a second, dynamically indexed load).
template <int CB>
__device__ __forceinline__ void softmax_regs(const float* lp, int C, long s_c, float (&p)[CB], float (&x)[CB], float& lse) {
    float m = -INFINITY;
    if (s_c == 1) {
        // channels-last row: 16-byte loads (rows are 16-byte aligned and padded to a multiple of 4 floats, so the last
        // vector may read pad channels -- inside the row, never used)
#pragma unroll
        for (int c4 = 0; c4 < CB; c4 += 4)
            if (c4 < C) {
                const float4 v = *(const float4*)(lp + c4);
                x[c4] = v.x;
                if (c4 + 1 < CB) x[c4 + 1] = v.y;
                if (c4 + 2 < CB) x[c4 + 2] = v.z;
                if (c4 + 3 < CB) x[c4 + 3] = v.w;
            }
    } else {
#pragma unroll
        for (int c = 0; c < CB; ++c)
            if (c < C) x[c] = lp[(size_t)c * s_c];
    }
#pragma unroll
    for (int c = 0; c < CB; ++c)
        if (c < C) {
            p[c] = x[c];
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
        const float* lp = q.logits + (size_t)b * q.s_b + (size_t)s * q.s_v;
        float p[CB], x[CB], lse;
        softmax_regs<CB>(lp, C, q.s_c, p, x, lse);
        const int t = map_target(q.target[v], q.map_occ);
        if (t != 255) {
            float pt = 0.f, xt = 0.f;
#pragma unroll
            for (int c = 0; c < CB; ++c)
                if (c < C) {
                    regP[c] += (u64)((double)p[c] * kQ32);
                    if (c == t) { pt = p[c]; xt = x[c]; }
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
        const float* lp = q.logits + (size_t)b * q.s_b + (size_t)s * q.s_v;
        float p[CB], g[CB], x[CB], lse;
        softmax_regs<CB>(lp, C, q.s_c, p, x, lse);
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
        float* gp = q.grad + (size_t)b * q.g_b + (size_t)s * q.g_v;
        if (q.g_c == 1) {
            // channels-last gradient rows (what the convolution backward consumes without a transpose): 16-byte stores,
            // pad channels [C, g_pad) written as zeros
            constexpr int CB4 = (CB + 3) & ~3;
            float o[CB4];
#pragma unroll
            for (int c = 0; c < CB4; ++c) o[c] = 0.f;
#pragma unroll
            for (int c = 0; c < CB; ++c)
                if (c < C) o[c] = p[c] * (g[c] - dot) + ce * (p[c] - (c == t ? 1.f : 0.f));
#pragma unroll
            for (int c4 = 0; c4 < CB4; c4 += 4)
                if (c4 < q.g_pad) *(float4*)(gp + c4) = float4{o[c4], o[c4 + 1], o[c4 + 2], o[c4 + 3]};
            for (int c = CB4; c < q.g_pad; ++c) gp[c] = 0.f;          // (rows wider than the class bucket: never at C = 20 / 2)
        } else {
#pragma unroll
            for (int c = 0; c < CB; ++c)
                if (c < C) gp[(size_t)c * q.g_c] = p[c] * (g[c] - dot) + ce * (p[c] - (c == t ? 1.f : 0.f));
        }
    }
}

// hist[t * C + argmax] += 1 over labelled voxels (first maximum wins, as np.argmax)
__global__ void __launch_bounds__(256) confusion_kernel(const float* logits, const uint8_t* labels,
                                                        const uint8_t* target, long total, int C, int S, u64* hist,
                                                        long s_b, long s_c, long s_v) {
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
            const float* lp = logits + (size_t)b * s_b + (size_t)(v - b * S) * s_v;
            float best = lp[0];
            pred = 0;
            for (int c = 1; c < C; ++c) {
                const float x = lp[(size_t)c * s_c];
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

// (B, C, S) planes or channels-last rows; anything else is the caller's transpose
bool strides_ok(const float* base, int32_t C, int64_t s_b, int64_t s_c, int64_t s_v) {
    if (s_c == 1) return s_v >= C && (s_v & 3) == 0 && (s_b & 3) == 0 && ((uintptr_t)base & 15) == 0;
    return s_v >= 1 && s_c >= 1 && s_b >= 0;
}

// ------------------------------------------------------------------------------------------------
// Relation (context prior) loss, occdepth/loss/CRP_loss.py:4-24: BCEWithLogits(pos_weight_r = #neg_r / #pos_r) over the
// (B, R, M, N) relation logits against the (B, R, N, M) 0/1 matrices, mean over all B R M N elements.  The loss is
//     1 / (R T) * sum_r [ pw_r * Spos_r + Sneg_r ],   Spos_r = sum_{y = 1} softplus(-x),  Sneg_r = sum_{y = 0} softplus(x),
// so ONE pass yields everything (pos_weight depends on the labels only and is applied to the sums afterwards); a second
// pass turns (g pw_r / (R T), g / (R T)) into d loss / d logits.  Sums in Q24 fixed point (order-independent).
// Thread mapping: lanes along the logits' unit-stride axis (coalesced 4-byte loads), every lane walks the other axis; the
// label matrix is read through the same (m, n) index with its own strides (unit stride along m: a lane that walks m
// re-reads its own cache line, a lane that walks n reads coalesced bytes).
struct RelP {
    const float* logits;
    const void* labels;        // (B, R, N, M) contiguous, uint8 or float32
    u64* stats;                // (R, 3): #pos, Spos (Q24), Sneg (Q24)
    const float* coef;         // backward: (R, 2) = (pos, neg) multipliers
    float* grad;               // backward: same strides as logits
    int B, R, M, N, lab_f32;
    long l_b, l_r, l_m, l_n;   // logits strides (floats)
    int lane_is_n;             // the lane axis: n (logits n-contiguous) or m
};

__device__ __forceinline__ float softplus_f(float x) {       // log(1 + e^x), stable
    return fmaxf(x, 0.f) + log1pf(__expf(-fabsf(x)));
}

constexpr int kRelWalk = 16;     // elements of the walk axis per thread

template <bool GRAD>
__global__ void __launch_bounds__(256) relation_bce_kernel(const RelP q) {
    __shared__ u64 sh[3];
    if (!GRAD && threadIdx.x < 3) sh[threadIdx.x] = 0;
    if (!GRAD) __syncthreads();
    const int r = blockIdx.y % q.R, b = blockIdx.y / q.R;
    const int n_lane = q.lane_is_n ? q.N : q.M, n_walk = q.lane_is_n ? q.M : q.N;
    const int lane_blocks = (n_lane + 255) / 256;
    const int lb = blockIdx.x % lane_blocks, wb = blockIdx.x / lane_blocks;
    const int il = lb * 256 + threadIdx.x;
    const int w0 = wb * kRelWalk, w1 = min(w0 + kRelWalk, n_walk);
    u64 cnt = 0, sp = 0, sn = 0;
    if (il < n_lane) {
        const long ls_lane = q.lane_is_n ? q.l_n : q.l_m, ls_walk = q.lane_is_n ? q.l_m : q.l_n;
        const long ys_lane = q.lane_is_n ? q.M : 1, ys_walk = q.lane_is_n ? 1 : q.M;     // labels (N, M): element (n, m) at n M + m
        const size_t lbase = (size_t)b * q.l_b + (size_t)r * q.l_r + (size_t)il * ls_lane;
        const size_t ybase = ((size_t)b * q.R + r) * q.N * q.M + (size_t)il * ys_lane;
        float cp = 0.f, cn = 0.f;
        if (GRAD) { cp = q.coef[2 * r]; cn = q.coef[2 * r + 1]; }
#pragma unroll 4
        for (int w = w0; w < w1; ++w) {
            const float x = q.logits[lbase + (size_t)w * ls_walk];
            const size_t yo = ybase + (size_t)w * ys_walk;
            const bool y = q.lab_f32 ? ((const float*)q.labels)[yo] != 0.f : ((const uint8_t*)q.labels)[yo] != 0;
            if (GRAD) {
                // d/dx [pw y softplus(-x) + (1 - y) softplus(x)] = y ? -pw sigmoid(-x) : sigmoid(x)
                const float e = __expf(-fabsf(x));
                const float s_abs = 1.f / (1.f + e);                      // sigmoid(|x|)
                const float sig = x >= 0.f ? s_abs : e * s_abs;           // sigmoid(x)
                q.grad[lbase + (size_t)w * ls_walk] = y ? -cp * (1.f - sig) : cn * sig;
            } else {
                const float v = softplus_f(y ? -x : x);
                const u64 f = (u64)((double)v * kQ24 + 0.5);
                if (y) { sp += f; cnt += 1; } else { sn += f; }
            }
        }
    }
    if (!GRAD) {
        // wave reduction (integer adds: order-free), one LDS atomic per wave, one global atomic per workgroup and sum
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            cnt += __shfl_down(cnt, o);
            sp += __shfl_down(sp, o);
            sn += __shfl_down(sn, o);
        }
        if ((threadIdx.x & 63) == 0) {
            if (cnt) atomicAdd(&sh[0], cnt);
            if (sp) atomicAdd(&sh[1], sp);
            if (sn) atomicAdd(&sh[2], sn);
        }
        __syncthreads();
        if (threadIdx.x < 3 && sh[threadIdx.x]) atomicAdd(&q.stats[r * 3 + threadIdx.x], sh[threadIdx.x]);
    }
}

int relation_setup(RelP* q, const float* logits, const void* labels, int32_t label_dtype, int64_t batch, int32_t R,
                   int64_t M, int64_t N, int64_t l_b, int64_t l_r, int64_t l_m, int64_t l_n, dim3* grid) {
    if (logits == nullptr || labels == nullptr || batch < 1 || R < 1 || R > 64 || M < 1 || N < 1) return OCCD_EINVAL;
    if (M > 0x7fffffff || N > 0x7fffffff || batch * R > 65535 || (label_dtype != 0 && label_dtype != 1)) return OCCD_EINVAL;
    if (l_m < 1 || l_n < 1 || (l_m != 1 && l_n != 1)) return OCCD_EINVAL;     // one of the two axes must be dense
    q->logits = logits; q->labels = labels; q->lab_f32 = label_dtype;
    q->B = (int)batch; q->R = R; q->M = (int)M; q->N = (int)N;
    q->l_b = l_b; q->l_r = l_r; q->l_m = l_m; q->l_n = l_n;
    q->lane_is_n = l_n == 1 ? 1 : 0;
    const long n_lane = q->lane_is_n ? N : M, n_walk = q->lane_is_n ? M : N;
    const long blocks = ((n_lane + 255) / 256) * ((n_walk + kRelWalk - 1) / kRelWalk);
    if (blocks > 0x7fffffffL) return OCCD_EINVAL;
    *grid = dim3((unsigned)blocks, (unsigned)(batch * R));
    return OCCD_OK;
}

// ------------------------------------------------------------------------------------------------
// Depth-distribution loss of FLoSP-Depth, occdepth/loss/depth_loss.py:18-87: the (Bn, srcH, srcW) sparse metric depth map is
// resampled (nearest) to cell x the (h, w) prediction grid, every cell x cell block keeps its smallest non-zero depth,
// that depth is binned with the LID step -> one-hot over the D bins (bin 0 and out-of-range = no target), and the loss is
// BCE(prob, one-hot) summed over the bins, averaged over the cells that have a target.  One thread per cell.
struct DepthP {
    const float* prob;         // (Bn, D, h, w), image stride p_b
    const float* gt;           // (Bn, srcH, srcW)
    u64* stats;                // [0] = sum of the per-cell BCE over measured cells (Q24), [1] = #measured cells
    const float* gscale;       // backward: device scalar g / max(1, #measured)
    float* grad;               // backward: (Bn, D, h, w) dense
    int Bn, D, h, w, srcH, srcW, cell;
    long p_b;
    float d_off, d_step;       // bin index = (depth - d_off) / d_step, d_off = float32(d_bound[0] - d_bound[2]) rounded by the host
    float inv_step;            // float32(1 / d_step): the divisor as ATen's scalar true-divide applies it
    float sc_h, sc_w;          // ATen's nearest-neighbour source index: min(floor(dst * scale), src - 1), scale = src / dst
};

__device__ __forceinline__ int depth_cell_bin(const DepthP& q, int bn, int y, int x) {
    float best = 1e5f;                                           // "no return" (the reference's 1e5)
    const float* g = q.gt + (size_t)bn * q.srcH * q.srcW;
    for (int dy = 0; dy < q.cell; ++dy) {
        const int sy = min((int)floorf((float)(y * q.cell + dy) * q.sc_h), q.srcH - 1);
        for (int dx = 0; dx < q.cell; ++dx) {
            const int sx = min((int)floorf((float)(x * q.cell + dx) * q.sc_w), q.srcW - 1);
            const float d = g[(size_t)sy * q.srcW + sx];
            if (d != 0.f) best = fminf(best, d);
        }
    }
    // ATen's true-divide by a Python scalar multiplies by the float reciprocal (a / b -> a * (1 / b) in the CUDA / HIP binary-op
    // kernel): with a step that is not a power of two, (best - off) / step and (best - off) * (1 / step) can land on either side
    // of a bin edge.  The reference's `(gt - (d_bound[0] - d_bound[2])) / d_bound[2]` runs on the GPU: follow it.
    const float idx = (best - q.d_off) * q.inv_step;
    // (idx < D + 1) & (idx >= 0) else 0; .long() truncates; one-hot column 0 is dropped -> target bin = k - 1, k >= 1
    const int k = (idx < (float)(q.D + 1) && idx >= 0.f) ? (int)idx : 0;
    return k - 1;                                                // -1: no target
}

template <bool GRAD>
__global__ void __launch_bounds__(256) depth_bce_kernel(const DepthP q) {
    __shared__ u64 sh[2];
    if (!GRAD && threadIdx.x < 2) sh[threadIdx.x] = 0;
    if (!GRAD) __syncthreads();
    const long cells = (long)q.Bn * q.h * q.w;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    u64 loss = 0, cnt = 0;
    if (i < cells) {
        const int x = (int)(i % q.w);
        const long t = i / q.w;
        const int y = (int)(t % q.h), bn = (int)(t / q.h);
        const int kb = depth_cell_bin(q, bn, y, x);
        const size_t hw = (size_t)q.h * q.w;
        const float* p = q.prob + (size_t)bn * q.p_b + (size_t)y * q.w + x;
        if (GRAD) {
            float* gp = q.grad + (size_t)bn * q.D * hw + (size_t)y * q.w + x;
            const float gs = kb >= 0 ? q.gscale[0] : 0.f;
            for (int d = 0; d < q.D; ++d) {
                // ATen binary_cross_entropy_backward: (p - t) / max((1 - p) p, 1e-12) * g
                const float pv = p[(size_t)d * hw];
                const float tv = d == kb ? 1.f : 0.f;
                gp[(size_t)d * hw] = gs * (pv - tv) / fmaxf((1.f - pv) * pv, 1e-12f);
            }
        } else if (kb >= 0) {
            float acc = 0.f;
            for (int d = 0; d < q.D; ++d) {
                const float pv = p[(size_t)d * hw];
                // -(t log p + (1 - t) log(1 - p)), both logs clamped at -100 (F.binary_cross_entropy)
                acc -= fmaxf(d == kb ? logf(pv) : log1pf(-pv), -100.f);
            }
            loss = (u64)((double)acc * kQ24 + 0.5);
            cnt = 1;
        }
    }
    if (!GRAD) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            loss += __shfl_down(loss, o);
            cnt += __shfl_down(cnt, o);
        }
        if ((threadIdx.x & 63) == 0) {
            if (loss) atomicAdd(&sh[0], loss);
            if (cnt) atomicAdd(&sh[1], cnt);
        }
        __syncthreads();
        if (threadIdx.x < 2 && sh[threadIdx.x]) atomicAdd(&q.stats[threadIdx.x], sh[threadIdx.x]);
    }
}

int depth_setup(DepthP* q, const float* prob, const float* gt, int64_t Bn, int32_t D, int32_t h, int32_t w, int32_t srcH,
                int32_t srcW, int32_t cell, int64_t p_b, float d_off, float d_step) {
    if (prob == nullptr || gt == nullptr || Bn < 1 || D < 1 || h < 1 || w < 1 || srcH < 1 || srcW < 1 || cell < 1 || cell > 64)
        return OCCD_EINVAL;
    if (!(d_step > 0.f) || p_b < (int64_t)D * h * w || Bn * (int64_t)h * w > 0x7fffffffL) return OCCD_EINVAL;
    q->prob = prob; q->gt = gt; q->Bn = (int)Bn; q->D = D; q->h = h; q->w = w; q->srcH = srcH; q->srcW = srcW;
    q->cell = cell; q->p_b = p_b; q->d_off = d_off; q->d_step = d_step;
    q->inv_step = 1.0f / d_step;
    // at::native::compute_scales_value<float>: (float) input_size / output_size
    q->sc_h = (float)srcH / (float)(h * cell);
    q->sc_w = (float)srcW / (float)(w * cell);
    return OCCD_OK;
}

}  // namespace

extern "C" {

int64_t occd_ssc_stats_len(int32_t C, int32_t F) { return 3 * (int64_t)C + 3 + (int64_t)F * C; }

int occd_ssc_loss_stats_fwd_strided(const float* logits, const uint8_t* target, const uint8_t* masks, const float* weights,
                                    int64_t* stats, int64_t batch, int32_t C, int64_t S, int32_t F, int32_t map_occ,
                                    int64_t s_b, int64_t s_c, int64_t s_v, void* stream) {
    int rc = check_stats(logits, target, batch, C, S, F);
    if (rc != OCCD_OK) return rc;
    if (stats == nullptr || (F > 0 && masks == nullptr) || !strides_ok(logits, C, s_b, s_c, s_v)) return OCCD_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    StatsP q{};
    q.logits = logits; q.target = target; q.masks = F > 0 ? masks : nullptr; q.weights = weights;
    q.stats = (u64*)stats; q.total = batch * S; q.C = C; q.S = (int)S; q.F = F; q.map_occ = map_occ;
    q.s_b = s_b; q.s_c = s_c; q.s_v = s_v;
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

int occd_ssc_loss_stats_fwd(const float* logits, const uint8_t* target, const uint8_t* masks, const float* weights,
                            int64_t* stats, int64_t batch, int32_t C, int64_t S, int32_t F, int32_t map_occ,
                            void* stream) {
    return occd_ssc_loss_stats_fwd_strided(logits, target, masks, weights, stats, batch, C, S, F, map_occ, (int64_t)C * S, S, 1,
                                           stream);
}

int occd_ssc_loss_stats_bwd_strided(const float* logits, const uint8_t* target, const uint8_t* masks, const float* weights,
                                    const float* gstats, float* grad, int64_t batch, int32_t C, int64_t S, int32_t F,
                                    int32_t map_occ, int64_t s_b, int64_t s_c, int64_t s_v, int64_t g_b, int64_t g_c,
                                    int64_t g_v, int32_t g_pad, void* stream) {
    int rc = check_stats(logits, target, batch, C, S, F);
    if (rc != OCCD_OK) return rc;
    if (gstats == nullptr || grad == nullptr || (F > 0 && masks == nullptr)) return OCCD_EINVAL;
    if (!strides_ok(logits, C, s_b, s_c, s_v) || !strides_ok(grad, C, g_b, g_c, g_v)) return OCCD_EINVAL;
    if (g_c == 1 ? (g_pad < C || g_pad > g_v || (g_pad & 3)) : g_pad != 0) return OCCD_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    StatsP q{};
    q.logits = logits; q.target = target; q.masks = F > 0 ? masks : nullptr; q.weights = weights;
    q.gstats = gstats; q.grad = grad; q.total = batch * S; q.C = C; q.S = (int)S; q.F = F; q.map_occ = map_occ;
    q.s_b = s_b; q.s_c = s_c; q.s_v = s_v; q.g_b = g_b; q.g_c = g_c; q.g_v = g_v; q.g_pad = g_pad;
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

int occd_ssc_loss_stats_bwd(const float* logits, const uint8_t* target, const uint8_t* masks, const float* weights,
                            const float* gstats, float* grad, int64_t batch, int32_t C, int64_t S, int32_t F,
                            int32_t map_occ, void* stream) {
    return occd_ssc_loss_stats_bwd_strided(logits, target, masks, weights, gstats, grad, batch, C, S, F, map_occ,
                                           (int64_t)C * S, S, 1, (int64_t)C * S, S, 1, 0, stream);
}

int occd_ssc_confusion_strided(const float* logits, const uint8_t* labels, const uint8_t* target, int64_t* hist,
                               int64_t batch, int32_t C, int64_t S, int64_t s_b, int64_t s_c, int64_t s_v, void* stream) {
    if ((logits == nullptr) == (labels == nullptr) || target == nullptr || hist == nullptr) return OCCD_EINVAL;
    if (batch < 1 || S < 1 || S > 0x7fffffff || C < 1 || C > kMaxC) return OCCD_EINVAL;
    if (logits != nullptr && (s_c < 1 || s_v < 1 || s_b < 0)) return OCCD_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const long total = batch * S;
    occd::ProfScope prof("ssc_confusion", st, 0.0, (double)total * (logits ? 4.0 * C + 1 : 2.0));
    hipLaunchKernelGGL(confusion_kernel, dim3(grid_for(total)), dim3(256), (size_t)C * C * sizeof(unsigned int), st,
                       logits, labels, target, total, C, (int)S, (u64*)hist, (long)s_b, (long)s_c, (long)s_v);
    return occd::check_launch();
}

int occd_ssc_confusion(const float* logits, const uint8_t* labels, const uint8_t* target, int64_t* hist,
                       int64_t batch, int32_t C, int64_t S, void* stream) {
    return occd_ssc_confusion_strided(logits, labels, target, hist, batch, C, S, (int64_t)C * S, S, 1, stream);
}

int occd_relation_bce_stats(const float* logits, const void* labels, int32_t label_dtype, int64_t* stats, int64_t batch,
                            int32_t R, int64_t M, int64_t N, int64_t l_b, int64_t l_r, int64_t l_m, int64_t l_n,
                            void* stream) {
    RelP q{};
    dim3 grid;
    const int rc = relation_setup(&q, logits, labels, label_dtype, batch, R, M, N, l_b, l_r, l_m, l_n, &grid);
    if (rc != OCCD_OK || stats == nullptr) return OCCD_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    q.stats = (u64*)stats;
    hipLaunchKernelGGL(zero_u64_kernel, dim3(1), dim3(256), 0, st, (u64*)stats, (long)R * 3);
    const double el = (double)batch * R * M * N;
    occd::ProfScope prof("relation_bce_stats", st, el * 8.0, el * (4.0 + (label_dtype ? 4.0 : 1.0)));
    hipLaunchKernelGGL(relation_bce_kernel<false>, grid, dim3(256), 0, st, q);
    return occd::check_launch();
}

int occd_relation_bce_grad(const float* logits, const void* labels, int32_t label_dtype, const float* coef, float* grad,
                           int64_t batch, int32_t R, int64_t M, int64_t N, int64_t l_b, int64_t l_r, int64_t l_m, int64_t l_n,
                           void* stream) {
    RelP q{};
    dim3 grid;
    const int rc = relation_setup(&q, logits, labels, label_dtype, batch, R, M, N, l_b, l_r, l_m, l_n, &grid);
    if (rc != OCCD_OK || coef == nullptr || grad == nullptr) return OCCD_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    q.coef = coef; q.grad = grad;
    const double el = (double)batch * R * M * N;
    occd::ProfScope prof("relation_bce_grad", st, el * 8.0, el * (8.0 + (label_dtype ? 4.0 : 1.0)));
    hipLaunchKernelGGL(relation_bce_kernel<true>, grid, dim3(256), 0, st, q);
    return occd::check_launch();
}

int occd_depth_bce_stats(const float* prob, const float* gt, int64_t* stats, int64_t Bn, int32_t D, int32_t h, int32_t w,
                         int32_t srcH, int32_t srcW, int32_t cell, int64_t p_b, float d_off, float d_step, void* stream) {
    DepthP q{};
    const int rc = depth_setup(&q, prob, gt, Bn, D, h, w, srcH, srcW, cell, p_b, d_off, d_step);
    if (rc != OCCD_OK || stats == nullptr) return OCCD_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    q.stats = (u64*)stats;
    hipLaunchKernelGGL(zero_u64_kernel, dim3(1), dim3(256), 0, st, (u64*)stats, 2L);
    const long cells = (long)Bn * h * w;
    occd::ProfScope prof("depth_bce_stats", st, (double)cells * D * 4.0, (double)cells * (4.0 * D + 4.0 * cell * cell));
    hipLaunchKernelGGL(depth_bce_kernel<false>, dim3((unsigned)((cells + 255) / 256)), dim3(256), 0, st, q);
    return occd::check_launch();
}

int occd_depth_bce_grad(const float* prob, const float* gt, const float* gscale, float* grad, int64_t Bn, int32_t D, int32_t h,
                        int32_t w, int32_t srcH, int32_t srcW, int32_t cell, int64_t p_b, float d_off, float d_step,
                        void* stream) {
    DepthP q{};
    const int rc = depth_setup(&q, prob, gt, Bn, D, h, w, srcH, srcW, cell, p_b, d_off, d_step);
    if (rc != OCCD_OK || gscale == nullptr || grad == nullptr) return OCCD_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    q.gscale = gscale; q.grad = grad;
    const long cells = (long)Bn * h * w;
    occd::ProfScope prof("depth_bce_grad", st, (double)cells * D * 4.0, (double)cells * (8.0 * D + 4.0 * cell * cell));
    hipLaunchKernelGGL(depth_bce_kernel<true>, dim3((unsigned)((cells + 255) / 256)), dim3(256), 0, st, q);
    return occd::check_launch();
}

}  // extern "C"
