// Squeeze-excite gate of the EfficientNet blocks, from the pooled partial sums the depthwise
// kernel leaves behind (occd_dwconv2d_pool_nchw) -- replaces x.mean((2, 3)) -> conv_reduce -> Swish -> conv_expand ->
// sigmoid (5 ATen / rocBLAS launches per block and a full extra read of the activation) by two small launches; the
// gate itself is applied to the input channels of the following pointwise GEMM (K11), never to the tensor in memory.
//   se_reduce_kernel : mean[c] = sum_j part[b][c][j] / S (fixed order);  r[b][i] = swish(sum_c Wr[i][c] mean[c] + br[i])
//   se_expand_kernel : gate[b][c] = sigmoid(sum_i We[c][i] r[b][i] + be[c])
// Both are tiny GEMVs (C <= 3840, Cr <= 160) spread over enough workgroups that no single CU streams the weights.
// Reference: geffnet SqueezeExcite behind occdepth/models/unet2d.py:175-190.
#include "common.h"

#include <atomic>
#include <cstdlib>
#include <mutex>

namespace {

constexpr int kWPre = 16;    // weight values a thread preloads (C <= 256 * kWPre = 4096; beyond, a plain loop)

// Latency, not bandwidth, decides this kernel (the weights are cold every frame -- the network's parameters exceed the
// Infinity Cache -- and every global load costs an HBM round trip): the first versions walked a weight row with one
// wave, 36-60 dependent loads deep (20-26 us per launch, 1.1 ms per frame).  Now ONE WORKGROUP PER OUTPUT: its 256
// threads issue their <= 16 weight loads first, then the pooling partials (tpc threads per channel, 4 independent
// accumulators), and only then touch any of them: two round trips per launch.  All sums keep a fixed order.
__global__ void __launch_bounds__(256) se_reduce_kernel(const float* __restrict__ part, const float* __restrict__ wr,
                                                        const float* __restrict__ br, float* __restrict__ r, int C,
                                                        int Cr, int nblk, float inv_s, int tpc) {
    extern __shared__ float mean[];                       // C floats
    __shared__ float wsum[4];
    const int b = blockIdx.y, i = blockIdx.x, t = threadIdx.x;
    const float* w = wr + (size_t)i * C;
    float wv[kWPre];
#pragma unroll
    for (int u = 0; u < kWPre; ++u) wv[u] = w[min(t + u * 256, C - 1)];        // (clamped; unused tail masked below)
    // ---- pooled means: channel c = c0 + t / tpc, partials j = t % tpc, + tpc, ...
    const float* pb = part + (size_t)b * C * nblk;
    const int cl = t / tpc, jp = t - cl * tpc, cpp = 256 / tpc;
    for (int c0 = 0; c0 < C; c0 += cpp) {
        const int c = c0 + cl;
        const float* pc = pb + (size_t)min(c, C - 1) * nblk;
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        int j = jp;
        for (; j + 3 * tpc < nblk; j += 4 * tpc) {
            s0 += pc[j]; s1 += pc[j + tpc]; s2 += pc[j + 2 * tpc]; s3 += pc[j + 3 * tpc];
        }
        for (; j < nblk; j += tpc) s0 += pc[j];
        float sum = (s0 + s1) + (s2 + s3);
        for (int off = tpc >> 1; off > 0; off >>= 1) sum += __shfl_down(sum, off, 64);   // tpc divides 64: groups stay in a wave
        if (jp == 0 && c < C) mean[c] = sum * inv_s;
    }
    __syncthreads();
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
    for (int u = 0; u < kWPre; u += 4) {
        const int c = t + u * 256;
        if (c < C) a0 += wv[u] * mean[c];
        if (c + 256 < C) a1 += wv[u + 1] * mean[c + 256];
        if (c + 512 < C) a2 += wv[u + 2] * mean[c + 512];
        if (c + 768 < C) a3 += wv[u + 3] * mean[c + 768];
    }
    for (int c = t + kWPre * 256; c < C; c += 256) a0 += w[c] * mean[c];
    float sum = (a0 + a1) + (a2 + a3);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) sum += __shfl_down(sum, off, 64);
    if ((t & 63) == 0) wsum[t >> 6] = sum;
    __syncthreads();
    if (t == 0) {
        const float v = (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]) + br[i];
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

// Cr % 4 == 0 (every EfficientNet stage): 4 lanes per output channel, each lane owns the float4 chunks q, q + 4, ... of
// the weight row (the 4 lanes of a channel read 64 consecutive bytes per step), ALL of its <= kEPre chunk loads are
// issued before the first is used (one memory round trip instead of Cr / 4 dependent ones), quad reduction on DPP.
constexpr int kEPre = 12;                                  // float4 chunks a lane preloads: Cr <= 16 * kEPre = 192
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(256) se_expand4_kernel(const float* __restrict__ r, const float* __restrict__ we,
                                                         const float* __restrict__ be, float* __restrict__ gate, int C,
                                                         int Cr) {
    extern __shared__ __attribute__((aligned(16))) float rs[];            // Cr floats
    const int b = blockIdx.y;
    const int q = threadIdx.x & 3, c = blockIdx.x * 64 + (threadIdx.x >> 2);
    const int cc = min(c, C - 1), nch = Cr >> 2;                          // float4 chunks per row
    const f32x4* w4 = (const f32x4*)(we + (size_t)cc * Cr);
    f32x4 wv[kEPre];
#pragma unroll
    for (int u = 0; u < kEPre; ++u) wv[u] = w4[min(q + 4 * u, nch - 1)];  // (clamped; the unused tail is masked below)
    const float bias = be[cc];
    for (int i = threadIdx.x; i < Cr; i += 256) rs[i] = r[(size_t)b * Cr + i];
    __syncthreads();
    const f32x4* r4 = (const f32x4*)rs;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < kEPre; ++u)
        if (q + 4 * u < nch) acc += wv[u] * r4[q + 4 * u];
    for (int k = q + 4 * kEPre; k < nch; k += 4) acc += w4[k] * r4[k];
    float s = (acc.x + acc.y) + (acc.z + acc.w);
    s += __shfl_xor(s, 1, 64);
    s += __shfl_xor(s, 2, 64);
    if (q == 0 && c < C) gate[(size_t)b * C + c] = 1.f / (1.f + expf(-(bias + s)));
}

// ---- Round 6: reduce + expand as ONE launch (110 launches of 5-6 us per frame were the squeeze-excite
// gates).  The two phases above are latency chains of two memory round trips each, separated by a kernel boundary; fused,
// the grid is max(Cr, C / 64) workgroups per image: workgroup w computes reduce output w (phase 1, exactly se_reduce_kernel's
// arithmetic and summation order) and PUBLISHES it as one self-validating 8-byte word {value : 32, sequence : 32} with an
// agent-scope atomic store; the first C / 64 workgroups then poll the Cr words of their image with agent-scope atomic loads
// (past the non-coherent L1 / per-XCD L2: MI355X_MICROARCH.md, "8-B agent atomics both sides" is a valid hand-off without
// any fence), and run se_expand4_kernel's arithmetic -- whose weight loads were issued BEFORE the poll, so the hand-off hides
// behind them.  Bit-identical to the two-launch form.  The sequence number lives in device memory (a slot of a per-device
// ring, zeroed once): every workgroup reads it at entry, the LAST workgroup to finish (a second counter) advances it, so a
// launch captured in a hipGraph replays correctly and stale words of earlier launches never match.  No workgroup waits for a
// workgroup that waits: publishers never poll before publishing, and <= 2 * 160 workgroups are always co-resident.
constexpr int kSeWords = 4096;                             // batch * Cr words per slot
struct SeSlot {
    unsigned int seq, done;
    unsigned long long pad;
    unsigned long long words[kSeWords];
};

__global__ void __launch_bounds__(256) se_fused_kernel(const float* __restrict__ part, const float* __restrict__ wr,
                                                       const float* __restrict__ br, const float* __restrict__ we,
                                                       const float* __restrict__ be, float* __restrict__ gate,
                                                       float* __restrict__ r_out, SeSlot* __restrict__ slot, int C, int Cr,
                                                       int nblk, float inv_s, int tpc, int nexp) {
    extern __shared__ __attribute__((aligned(16))) float sm[];            // mean[C] | rs[Cr]
    float* const mean = sm;
    float* const rs = sm + ((C + 3) & ~3);
    __shared__ float wsum[4];
    const int b = blockIdx.y, w = blockIdx.x, t = threadIdx.x;
    const unsigned seq = __hip_atomic_load(&slot->seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
    unsigned long long* const words = slot->words + (size_t)b * Cr;
    // expand operands of this workgroup's 64 channels (requested first: they are not needed before the hand-off)
    const bool expands = w < nexp;
    const int q = t & 3, c_e = w * 64 + (t >> 2);
    const int cc = min(c_e, C - 1), nch = Cr >> 2;
    const f32x4* w4 = (const f32x4*)(we + (size_t)cc * Cr);
    f32x4 ev[kEPre];
    float ebias = 0.f;
    if (expands) {
#pragma unroll
        for (int u = 0; u < kEPre; ++u) ev[u] = w4[min(q + 4 * u, nch - 1)];
        ebias = be[cc];
    }
    // ---- phase 1: reduce output i = w (se_reduce_kernel, same order of every sum)
    if (w < Cr) {
        const int i = w;
        const float* wrow = wr + (size_t)i * C;
        float wv[kWPre];
#pragma unroll
        for (int u = 0; u < kWPre; ++u) wv[u] = wrow[min(t + u * 256, C - 1)];
        const float* pb = part + (size_t)b * C * nblk;
        const int cl = t / tpc, jp = t - cl * tpc, cpp = 256 / tpc;
        for (int c0 = 0; c0 < C; c0 += cpp) {
            const int c = c0 + cl;
            const float* pc = pb + (size_t)min(c, C - 1) * nblk;
            float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
            int j = jp;
            for (; j + 3 * tpc < nblk; j += 4 * tpc) {
                s0 += pc[j]; s1 += pc[j + tpc]; s2 += pc[j + 2 * tpc]; s3 += pc[j + 3 * tpc];
            }
            for (; j < nblk; j += tpc) s0 += pc[j];
            float sum = (s0 + s1) + (s2 + s3);
            for (int off = tpc >> 1; off > 0; off >>= 1) sum += __shfl_down(sum, off, 64);
            if (jp == 0 && c < C) mean[c] = sum * inv_s;
        }
        __syncthreads();
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
        for (int u = 0; u < kWPre; u += 4) {
            const int c = t + u * 256;
            if (c < C) a0 += wv[u] * mean[c];
            if (c + 256 < C) a1 += wv[u + 1] * mean[c + 256];
            if (c + 512 < C) a2 += wv[u + 2] * mean[c + 512];
            if (c + 768 < C) a3 += wv[u + 3] * mean[c + 768];
        }
        float sum = (a0 + a1) + (a2 + a3);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) sum += __shfl_down(sum, off, 64);
        if ((t & 63) == 0) wsum[t >> 6] = sum;
        __syncthreads();
        if (t == 0) {
            const float v = (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]) + br[i];
            const float r = v / (1.f + expf(-v));
            r_out[(size_t)b * Cr + i] = r;              // the squeezed activations: the training backward reads them (occd_se_bwd)
            __hip_atomic_store(words + i, (unsigned long long)__float_as_uint(r) | ((unsigned long long)seq << 32),
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    // ---- phase 2: the gates of this workgroup's 64 channels (se_expand4_kernel) once all Cr words of the image carry seq
    if (expands) {
        for (int i = t; i < Cr; i += 256) {
            unsigned long long word;
            while (true) {
                word = __hip_atomic_load(words + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if ((unsigned)(word >> 32) == seq) break;
                __builtin_amdgcn_s_sleep(1);
            }
            rs[i] = __uint_as_float((unsigned)word);
        }
        __syncthreads();
        const f32x4* r4 = (const f32x4*)rs;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < kEPre; ++u)
            if (q + 4 * u < nch) acc += ev[u] * r4[q + 4 * u];
        for (int k = q + 4 * kEPre; k < nch; k += 4) acc += w4[k] * r4[k];
        float s = (acc.x + acc.y) + (acc.z + acc.w);
        s += __shfl_xor(s, 1, 64);
        s += __shfl_xor(s, 2, 64);
        if (q == 0 && c_e < C) gate[(size_t)b * C + c_e] = 1.f / (1.f + expf(-(ebias + s)));
    }
    // ---- the last workgroup of the launch to get here advances the sequence number (everybody has read it by then)
    __syncthreads();
    if (t == 0) {
        const unsigned total = gridDim.x * gridDim.y;
        if (__hip_atomic_fetch_add(&slot->done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == total - 1) {
            __hip_atomic_store(&slot->done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&slot->seq, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// ---- DepthNet's camera-aware gate (occdepth/models/flosp_depth/flosp_depth.py:201-257: Mlp of the scaled pixel size ->
// SELayer): gate[i][c] = sigmoid(We . relu(Wr . (W2 . relu(w1 s_i + b1) + b2) + br) + be), i = image (b, view),
// s_i = 1000 * |(1 / fx_i, 1 / fy_i)| (the diagonal of the inverse pinhole intrinsics) or a given scalar (infer_mode).
// Round 5: ONE launch instead of ~20 ATen / rocBLAS ones (reciprocal, stack, norm, 2 Linear, ReLU, 2 1x1 convolutions on 1x1
// maps, ReLU, sigmoid: the last Cijk_* rows of the eval trace).  One workgroup per image; the three C x C mat-vecs run
// row per wave-quarter: 16 lanes walk a row in coalesced 64-byte pieces and reduce through DPP / shuffles (fixed order).
// (C <= 128, DepthNet's width: ALL of a lane's 64 weights are requested before the first is used -- the launch is two
//  workgroups walking three dependent mat-vecs, i.e. pure latency: the rolled form below took 60 us, 8 L2 round trips per row)
__device__ __forceinline__ void gate_matvec128(const float* __restrict__ W, const float* __restrict__ bias, const float* in,
                                               float* out, int C, int act) {
    const int sub = threadIdx.x & 15, grp = threadIdx.x >> 4;
    float wv[8][8];
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int i = 0; i < 8; ++i)
            wv[j][i] = W[(size_t)min(grp + 16 * j, C - 1) * C + min(sub + 16 * i, C - 1)];
    float iv[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) iv[i] = sub + 16 * i < C ? in[sub + 16 * i] : 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int r = grp + 16 * j;
        float a = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) a += wv[j][i] * iv[i];
        a += __shfl_xor(a, 8, 64);
        a += __shfl_xor(a, 4, 64);
        a += __shfl_xor(a, 2, 64);
        a += __shfl_xor(a, 1, 64);
        if (sub == 0 && r < C) {
            a += bias[r];
            out[r] = act == 1 ? fmaxf(a, 0.f) : act == 2 ? 1.f / (1.f + expf(-a)) : a;
        }
    }
}

__device__ __forceinline__ void gate_matvec(const float* __restrict__ W, const float* __restrict__ bias, const float* in,
                                            float* out, int C, int act) {        // out = act(W in + bias); act 1 relu, 2 sigmoid
    if (C <= 128) { gate_matvec128(W, bias, in, out, C, act); return; }
    const int sub = threadIdx.x & 15, grp = threadIdx.x >> 4;                      // 16 row groups of 16 lanes
    for (int r = grp; r < C; r += 16) {
        const float* w = W + (size_t)r * C;
        float a = 0.f;
        for (int k = sub; k < C; k += 16) a += w[k] * in[k];
        a += __shfl_xor(a, 8, 64);
        a += __shfl_xor(a, 4, 64);
        a += __shfl_xor(a, 2, 64);
        a += __shfl_xor(a, 1, 64);
        if (sub == 0) {
            a += bias[r];
            out[r] = act == 1 ? fmaxf(a, 0.f) : act == 2 ? 1.f / (1.f + expf(-a)) : a;
        }
    }
}

__global__ void __launch_bounds__(256) depthnet_gate_kernel(const float* __restrict__ sps, const float* __restrict__ intrins,
                                                            long intr_stride, float factor, const float* __restrict__ w1,
                                                            const float* __restrict__ b1, const float* __restrict__ w2,
                                                            const float* __restrict__ b2, const float* __restrict__ wr,
                                                            const float* __restrict__ br, const float* __restrict__ we,
                                                            const float* __restrict__ be, float* __restrict__ gate, int C) {
    extern __shared__ float buf[];                       // 2 x C
    float* v0 = buf;
    float* v1 = buf + C;
    const int i = blockIdx.x;
    float s;
    if (sps != nullptr) {
        s = sps[i];
    } else {
        // DepthNet.scaled_pixel_size: diag(K^-1) = 1 / diag(K) for upper-triangular pinhole intrinsics; torch.norm of the pair
        const float d0 = 1.f / intrins[(size_t)i * intr_stride], d1 = 1.f / intrins[(size_t)i * intr_stride + 5];
        s = sqrtf(d0 * d0 + d1 * d1) * factor;
    }
    for (int c = threadIdx.x; c < C; c += 256) v0[c] = fmaxf(w1[c] * s + b1[c], 0.f);     // fc1 (C, 1) + ReLU
    __syncthreads();
    gate_matvec(w2, b2, v0, v1, C, 0);                   // fc2
    __syncthreads();
    gate_matvec(wr, br, v1, v0, C, 1);                   // conv_reduce + ReLU
    __syncthreads();
    gate_matvec(we, be, v0, gate + (size_t)i * C, C, 2); // conv_expand + sigmoid
}

// ---- squeeze-excite in TRAINING (round 5): forward and backward of geffnet's SqueezeExcite
//     m = mean_hw(x),  r = swish(Wr m + br),  g = sigmoid(We r + be),  out = x * g
// as 4 + 4 launches instead of the ~22 of the autograd graph (mean, two 1x1 convolutions with bias on 1x1 maps, swish, sigmoid,
// broadcast multiply; and backwards: 2 multiplies, ~12 `sum` launches, the convolutions' data / weight / bias gradients, the
// mean's expand + divide, a gradient add): 55 blocks x that was 726 tiny `aten::sum` launches of the config-2 step alone.
//   forward : plane_reduce (sum)  ->  occd_se_gate (nblk = 1)  ->  affine (x * g)
//   backward: plane_reduce (dot: gg = sum_hw gout * x)  ->  se_bwd_reduce  ->  se_bwd_expand  ->  affine (gout * g + gm / S)
// All reductions run in a fixed order (deterministic).
constexpr int kSeMaxB = 16;      // images per call (batch x views)

// out[p] = sum_s a[p][s] (* b[p][s]) over the S elements of plane p; TPP threads per plane (64: one wave, 256: the workgroup)
template <int TPP, bool DOT>
__global__ void __launch_bounds__(256) plane_reduce_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                           float* __restrict__ out, long planes, long S) {
    __shared__ float part[4];
    const int sub = threadIdx.x % TPP;
    const long p = (long)blockIdx.x * (256 / TPP) + threadIdx.x / TPP;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (p < planes) {
        const float* pa = a + (size_t)p * S;
        const float* pb = DOT ? b + (size_t)p * S : nullptr;
        long i = sub;
        for (; i + 3 * TPP < S; i += 4 * TPP) {
            if (DOT) {
                s0 += pa[i] * pb[i]; s1 += pa[i + TPP] * pb[i + TPP];
                s2 += pa[i + 2 * TPP] * pb[i + 2 * TPP]; s3 += pa[i + 3 * TPP] * pb[i + 3 * TPP];
            } else {
                s0 += pa[i]; s1 += pa[i + TPP]; s2 += pa[i + 2 * TPP]; s3 += pa[i + 3 * TPP];
            }
        }
        for (; i < S; i += TPP) s0 += DOT ? pa[i] * pb[i] : pa[i];
    }
    float s = (s0 + s1) + (s2 + s3);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
    if (TPP == 64) {
        if (sub == 0 && p < planes) out[p] = s;
    } else {
        if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
        __syncthreads();
        if (threadIdx.x == 0 && p < planes) out[p] = (part[0] + part[1]) + (part[2] + part[3]);
    }
}

struct SeBwdP {
    const float* gg;      // (B, C)  sum_hw gout * x
    const float* gate;    // (B, C)
    const float* sums;    // (B, C)  sum_hw x  (m = sums / S)
    const float* r;       // (B, Cr) swish(z) from the forward
    const float* wr;      // (Cr, C)
    const float* br;      // (Cr)
    const float* we;      // (C, Cr)
    float* dz;            // (B, Cr)   out of phase 1
    float* gbr;           // (Cr)
    float* gm;            // (B, C)    out of phase 2: d loss / d mean
    float* gwe;           // (C, Cr)
    float* gwr;           // (Cr, C)
    float* gbe;           // (C)
    int B, C, Cr;
    float inv_s;
};

// phase 1: 16 squeeze channels i per workgroup; thread (cl = t / 16, il = t % 16) walks c = cl, cl + 16, ...
//   gr[b][i] = sum_c ds[b][c] We[c][i],  ds = gg g (1 - g);   z[b][i] = sum_c Wr[i][c] m[b][c] + br[i]
//   dz[b][i] = gr swish'(z),  swish'(z) = s (1 + z (1 - s)),  s = sigmoid(z);   gbr[i] = sum_b dz[b][i]
__global__ void __launch_bounds__(256) se_bwd_reduce_kernel(const SeBwdP q) {
    __shared__ float red[16][16][2];                   // [cl][il][gr | z] of one image at a time
    const int il = threadIdx.x & 15, cl = threadIdx.x >> 4;
    const int i = blockIdx.x * 16 + il;
    const bool i_ok = i < q.Cr;
    const int ic = i_ok ? i : q.Cr - 1;
    float gbr = 0.f;
    for (int b = 0; b < q.B; ++b) {
        const float* gg = q.gg + (size_t)b * q.C;
        const float* g = q.gate + (size_t)b * q.C;
        const float* sm = q.sums + (size_t)b * q.C;
        float a_gr = 0.f, a_z = 0.f;
        for (int c = cl; c < q.C; c += 16) {
            const float gv = g[c];
            a_gr += gg[c] * gv * (1.f - gv) * q.we[(size_t)c * q.Cr + ic];
            a_z += q.wr[(size_t)ic * q.C + c] * (sm[c] * q.inv_s);
        }
        red[cl][il][0] = a_gr;
        red[cl][il][1] = a_z;
        __syncthreads();
        if (cl == 0) {
            float gr = 0.f, z = 0.f;
#pragma unroll
            for (int k = 0; k < 16; ++k) { gr += red[k][il][0]; z += red[k][il][1]; }
            z += q.br[ic];
            const float sg = 1.f / (1.f + expf(-z));
            const float dz = gr * sg * (1.f + z * (1.f - sg));
            if (i_ok) q.dz[(size_t)b * q.Cr + i] = dz;
            gbr += dz;
        }
        __syncthreads();
    }
    if (cl == 0 && i_ok) q.gbr[i] = gbr;
}

// phase 2: 256 channels c per workgroup
//   gm[b][c] = sum_i dz[b][i] Wr[i][c];  gWr[i][c] = sum_b dz[b][i] m[b][c];  gbe[c] = sum_b ds[b][c];
//   gWe[c][i] = sum_b ds[b][c] r[b][i]   (second pass, threads remapped so that a row's i are written by neighbouring lanes)
__global__ void __launch_bounds__(256) se_bwd_expand_kernel(const SeBwdP q) {
    extern __shared__ float sh[];                      // dz (B, Cr) | r (B, Cr) | ds (B, 256)
    float* s_dz = sh;
    float* s_r = sh + q.B * q.Cr;
    float* s_ds = s_r + q.B * q.Cr;
    for (int k = threadIdx.x; k < q.B * q.Cr; k += 256) { s_dz[k] = q.dz[k]; s_r[k] = q.r[k]; }
    const int c0 = blockIdx.x * 256, c = c0 + threadIdx.x;
    const bool c_ok = c < q.C;
    float m[kSeMaxB], gm[kSeMaxB];
    float gbe = 0.f;
#pragma unroll
    for (int b = 0; b < kSeMaxB; ++b) {
        m[b] = gm[b] = 0.f;
        if (b < q.B) {
            float ds = 0.f;
            if (c_ok) {
                const float gv = q.gate[(size_t)b * q.C + c];
                ds = q.gg[(size_t)b * q.C + c] * gv * (1.f - gv);
                m[b] = q.sums[(size_t)b * q.C + c] * q.inv_s;
            }
            s_ds[b * 256 + threadIdx.x] = ds;
            gbe += ds;
        }
    }
    __syncthreads();
    if (c_ok) {
        for (int i = 0; i < q.Cr; ++i) {
            const float w = q.wr[(size_t)i * q.C + c];
            float gw = 0.f;
#pragma unroll
            for (int b = 0; b < kSeMaxB; ++b)
                if (b < q.B) {
                    const float d = s_dz[b * q.Cr + i];
                    gm[b] += d * w;
                    gw += d * m[b];
                }
            q.gwr[(size_t)i * q.C + c] = gw;
        }
#pragma unroll
        for (int b = 0; b < kSeMaxB; ++b)
            if (b < q.B) q.gm[(size_t)b * q.C + c] = gm[b];
        q.gbe[c] = gbe;
    }
    const int nc = min(256, q.C - c0);
    for (int o = threadIdx.x; o < nc * q.Cr; o += 256) {
        const int cl = o / q.Cr, i = o - cl * q.Cr;
        float v = 0.f;
        for (int b = 0; b < q.B; ++b) v += s_ds[b * 256 + cl] * s_r[b * q.Cr + i];
        q.gwe[(size_t)(c0 + cl) * q.Cr + i] = v;
    }
}

}  // namespace

extern "C" int occd_plane_reduce(const float* a, const float* b, float* out, int64_t planes, int64_t S, void* stream) {
    if (a == nullptr || out == nullptr || planes < 1 || S < 1 || planes > 0x7fffffffL * 4) return OCCD_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    occd::ProfScope prof(b != nullptr ? "plane_dot" : "plane_sum", st, (double)planes * S * (b ? 2.0 : 1.0),
                         4.0 * planes * S * (b ? 2.0 : 1.0));
    // one wave per plane for small planes (the 1/16 and 1/32 encoder stages), the workgroup for large ones
    if (S >= 4096) {
        if (b != nullptr) hipLaunchKernelGGL((plane_reduce_kernel<256, true>), dim3((unsigned)planes), dim3(256), 0, st, a, b, out, (long)planes, (long)S);
        else hipLaunchKernelGGL((plane_reduce_kernel<256, false>), dim3((unsigned)planes), dim3(256), 0, st, a, b, out, (long)planes, (long)S);
    } else {
        const unsigned grid = (unsigned)((planes + 3) / 4);
        if (b != nullptr) hipLaunchKernelGGL((plane_reduce_kernel<64, true>), dim3(grid), dim3(256), 0, st, a, b, out, (long)planes, (long)S);
        else hipLaunchKernelGGL((plane_reduce_kernel<64, false>), dim3(grid), dim3(256), 0, st, a, b, out, (long)planes, (long)S);
    }
    return occd::check_launch();
}

extern "C" int occd_se_bwd(const float* gg, const float* gate, const float* sums, const float* r, const float* w_reduce,
                           const float* b_reduce, const float* w_expand, float* dz_scratch, float* gm, float* gw_reduce,
                           float* gb_reduce, float* gw_expand, float* gb_expand, int32_t batch, int32_t C, int32_t Cr,
                           int64_t S, void* stream) {
    if (!gg || !gate || !sums || !r || !w_reduce || !b_reduce || !w_expand || !dz_scratch || !gm || !gw_reduce || !gb_reduce ||
        !gw_expand || !gb_expand)
        return OCCD_EINVAL;
    if (batch < 1 || batch > kSeMaxB || C < 1 || C > 16384 || Cr < 1 || Cr > 4096 || S < 1) return OCCD_EINVAL;
    const size_t lds = ((size_t)2 * batch * Cr + (size_t)batch * 256) * sizeof(float);
    if (lds > 64 * 1024) return OCCD_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    SeBwdP q{};
    q.gg = gg; q.gate = gate; q.sums = sums; q.r = r; q.wr = w_reduce; q.br = b_reduce; q.we = w_expand;
    q.dz = dz_scratch; q.gbr = gb_reduce; q.gm = gm; q.gwe = gw_expand; q.gwr = gw_reduce; q.gbe = gb_expand;
    q.B = batch; q.C = C; q.Cr = Cr; q.inv_s = (float)(1.0 / (double)S);
    occd::ProfScope prof("se_bwd", st, 8.0 * batch * C * Cr, 16.0 * C * Cr);
    hipLaunchKernelGGL(se_bwd_reduce_kernel, dim3((unsigned)((Cr + 15) / 16)), dim3(256), 0, st, q);
    hipLaunchKernelGGL(se_bwd_expand_kernel, dim3((unsigned)((C + 255) / 256)), dim3(256), lds, st, q);
    return occd::check_launch();
}

extern "C" int occd_depthnet_gate(const float* sps, const float* intrins, int64_t intr_stride, float factor, const float* w1,
                                  const float* b1, const float* w2, const float* b2, const float* wr, const float* br,
                                  const float* we, const float* be, float* gate, int32_t images, int32_t C, void* stream) {
    if ((sps == nullptr) == (intrins == nullptr) || !w1 || !b1 || !w2 || !b2 || !wr || !br || !we || !be || !gate) return OCCD_EINVAL;
    if (images < 1 || C < 1 || C > 4096 || (intrins != nullptr && intr_stride < 6)) return OCCD_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    occd::ProfScope prof("depthnet_gate", st, 6.0 * images * C * C, 12.0 * C * C);
    hipLaunchKernelGGL(depthnet_gate_kernel, dim3((unsigned)images), dim3(256), (size_t)2 * C * sizeof(float), st, sps, intrins,
                       (long)intr_stride, factor, w1, b1, w2, b2, wr, br, we, be, gate, C);
    return occd::check_launch();
}

// Per-device ring of hand-off slots for se_fused_kernel (zeroed once; launches of one stream are ordered, launches captured
// on different streams get different slots).  Allocated at the first call on a device -- the eval graph's warm-up forwards run
// before any capture, like K2s' work-list counters (csrc/conv3d_c32p.hip).
namespace {
constexpr int kSeSlots = 128;
constexpr int kSeMaxDevices = 64;
struct SeDev {
    std::mutex mu;
    SeSlot* slots = nullptr;
};
SeDev g_se_dev[kSeMaxDevices];
std::atomic<unsigned> g_se_next{0};
// Measured and NOT adopted (round 6, same box): the config-2 frame 18.017 ms with the one-launch form, 18.021 ms with two launches;
// per launch under HIP events 14.5 us fused against 11.8 us for the pair at 2304 -> 96 -> 2304 (the expand workgroups wait for the
// SLOWEST of the 96 reduce workgroups plus one agent-scope poll hop: what a kernel boundary costs inside a hipGraph is no more).
// Default: two launches; OCCD_SE_FUSED=1 / occd_se_gate_set_fused(1) select the one-launch kernel (kept: tested, bit-identical).
std::atomic<int> g_se_fused{occd::env_flag("OCCD_SE_FUSED", false) ? 1 : 0};
SeSlot* se_slot() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kSeMaxDevices) return nullptr;
    SeDev& d = g_se_dev[dev];
    std::lock_guard<std::mutex> lock(d.mu);
    if (d.slots == nullptr) {
        SeSlot* p = nullptr;
        if (hipMalloc(&p, sizeof(SeSlot) * kSeSlots) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
        if (hipMemset(p, 0, sizeof(SeSlot) * kSeSlots) != hipSuccess || hipDeviceSynchronize() != hipSuccess) {
            (void)hipGetLastError();
            (void)hipFree(p);
            return nullptr;
        }
        d.slots = p;
    }
    return d.slots + (g_se_next.fetch_add(1) % kSeSlots);
}
}  // namespace

extern "C" int32_t occd_se_gate_set_fused(int32_t on) { return g_se_fused.exchange(on != 0 ? 1 : 0); }

extern "C" int occd_se_gate(const float* pool_part, const float* w_reduce, const float* b_reduce, const float* w_expand,
                            const float* b_expand, float* r_scratch, float* gate, int32_t batch, int32_t C, int32_t Cr,
                            int32_t nblk, int64_t S, void* stream) {
    if (!pool_part || !w_reduce || !b_reduce || !w_expand || !b_expand || !r_scratch || !gate) return OCCD_EINVAL;
    if (batch < 1 || batch > 65535 || C < 1 || C > 16384 || Cr < 1 || Cr > 4096 || nblk < 1 || S < 1) return OCCD_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    occd::ProfScope prof("se_gate", st, 4.0 * batch * C * Cr, 8.0 * C * Cr + 4.0 * batch * C * nblk);
    int tpc = 1;                                           // threads per channel in the pooling phase
    while (tpc < 64 && (long)tpc * 2 * C <= 256 && tpc * 2 <= nblk) tpc *= 2;
    // one launch (se_fused_kernel) where its preloads cover the rows: C <= 4096, Cr % 4 == 0 and <= 192, batch * Cr words fit a
    // slot; opt-in (OCCD_SE_FUSED=1 / occd_se_gate_set_fused(1)): see g_se_fused for the measurement that kept two launches the default
    if (g_se_fused.load() != 0 && C <= 256 * kWPre && (Cr & 3) == 0 && Cr <= 16 * kEPre && (long)batch * Cr <= kSeWords &&
        (reinterpret_cast<uintptr_t>(w_expand) & 15) == 0) {
        SeSlot* slot = se_slot();
        if (slot == nullptr) return OCCD_ELAUNCH;
        const int nexp = (C + 63) / 64;
        const size_t lds = ((size_t)((C + 3) & ~3) + Cr) * sizeof(float);
        hipLaunchKernelGGL(se_fused_kernel, dim3((unsigned)(nexp > Cr ? nexp : Cr), (unsigned)batch), dim3(256), lds, st, pool_part,
                           w_reduce, b_reduce, w_expand, b_expand, gate, r_scratch, slot, C, Cr, nblk, (float)(1.0 / (double)S), tpc,
                           nexp);
        return occd::check_launch();
    }
    hipLaunchKernelGGL(se_reduce_kernel, dim3((unsigned)Cr, (unsigned)batch), dim3(256), (size_t)C * sizeof(float), st,
                       pool_part, w_reduce, b_reduce, r_scratch, C, Cr, nblk, (float)(1.0 / (double)S), tpc);
    static const bool old_expand = occd::env_flag("OCCD_SE_EXPAND_OLD", false);                      // A/B switch
    if (!old_expand && (Cr & 3) == 0 && (reinterpret_cast<uintptr_t>(w_expand) & 15) == 0)
        hipLaunchKernelGGL(se_expand4_kernel, dim3((unsigned)((C + 63) / 64), (unsigned)batch), dim3(256),
                           (size_t)Cr * sizeof(float), st, r_scratch, w_expand, b_expand, gate, C, Cr);
    else
        hipLaunchKernelGGL(se_expand_kernel, dim3((unsigned)((C + 255) / 256), (unsigned)batch), dim3(256),
                           (size_t)Cr * sizeof(float), st, r_scratch, w_expand, b_expand, gate, C, Cr);
    return occd::check_launch();
}
