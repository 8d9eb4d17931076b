// K13 -- training-mode BatchNorm (+ activation + residual) as HBM-bound passes, forward and backward, for both tensor
// layouts of the model: channels-last voxel / pixel rows (the 3-D stack, the bf16 decoder) and NCHW planes (the 2-D
// encoder).  The reference's training step sends every BatchNorm through the backend's batch_norm (MIOpen: statistics
// pass + normalise pass, then separate ReLU / add kernels; backward: three more passes) and its SyncBatchNorm through an
// all_gather per layer; here a layer is
//   forward : statistics pass (shifted sums: sum(x - x0), sum((x - x0)^2), x0 = the tensor's first element of the channel,
//             so the single pass cannot cancel whatever mean / std is) -> per-channel combine in float64 ->
//             [ONE packed all-reduce of 2C + 1 doubles when the statistics span ranks] -> finish (mean, invstd, running
//             statistics, the affine a = gamma * invstd, b = beta - mean * a) -> ONE apply pass
//             y = act(x * a + b [+ res]) [+ res]   (ReLU / LeakyReLU / swish and the residual add fused);
//   backward: reduction pass (g = gy * act'(.), sum g, sum g * xhat) -> [ONE all-reduce of 2C floats] -> finish
//             (gamma / beta gradients, the three per-channel coefficients) -> ONE apply pass gx = g k1 + x k2 + k3.
// Every partial sum is combined in a fixed order: results are deterministic.
// Reference semantics replaced: torch.nn.BatchNorm{2,3}d / SyncBatchNorm in training mode as used by
// occdepth/models/DDR.py:111-139, modules.py:40-46,158-175,278-296, unet2d.py:24-46, scripts/train.py:179 (sync_batchnorm).
#include "common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int kMaxBlocks = 1024;

// ------------------------------------------------------------------------------------------------ element access
template <bool BF16>
__device__ __forceinline__ f32x4 ld4(const void* p, size_t e) {
    if (BF16) {
        const u32x2 v = *(const u32x2*)((const uint16_t*)p + e);
        f32x4 r;
        r.x = __builtin_bit_cast(float, v.x << 16);
        r.y = __builtin_bit_cast(float, v.x & 0xffff0000u);
        r.z = __builtin_bit_cast(float, v.y << 16);
        r.w = __builtin_bit_cast(float, v.y & 0xffff0000u);
        return r;
    }
    return *(const f32x4*)((const float*)p + e);
}

template <bool BF16>
__device__ __forceinline__ void st4(void* p, size_t e, f32x4 v) {
    if (BF16) {
        const bf16x4 o = {(__bf16)v.x, (__bf16)v.y, (__bf16)v.z, (__bf16)v.w};
        *(u32x2*)((uint16_t*)p + e) = __builtin_bit_cast(u32x2, o);
    } else {
        *(f32x4*)((float*)p + e) = v;
    }
}

__device__ __forceinline__ float act_fwd(float v, int act, float slope) {
    if (act == 1) return fmaxf(v, 0.f);
    if (act == 2) return occd::swish_fast(v);
    if (act == 3) return v > 0.f ? v : v * slope;
    return v;
}

// d act / d pre-activation.  `pre` is the pre-activation, or (sign_only) any value with its sign (the saved output).
__device__ __forceinline__ float act_bwd(float pre, int act, float slope) {
    if (act == 1) return pre > 0.f ? 1.f : 0.f;
    if (act == 3) return pre > 0.f ? 1.f : slope;
    if (act == 2) {
        const float s = __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(pre * -1.4426950408889634f));
        return s * (1.f + pre * (1.f - s));
    }
    return 1.f;
}

struct BnP {
    const void* x;
    const void* gy;
    const void* y;
    const void* res;
    void* out;
    void* out2;
    const float* a;
    const float* b;
    const float* mean;
    const float* invstd;
    const float* k1;
    const float* k2;
    const float* k3;
    float* partial;
    long rows;          // layout 0
    long S;             // layout 1: plane size
    int batch;          // layout 1
    int C, C4, cw;
    int x_cs, x_coff, gy_cs, gy_coff, y_cs, y_coff, res_cs, res_coff, out_cs, out_coff, out2_cs, out2_coff;
    int act, res_first;
    float slope;
    int nblk;
    long rows_per_blk;
};

// shared block reduction of (s1[4], s2[4]) over the RPW row-lanes that hold the same channel quad
__device__ __forceinline__ void reduce_rows_and_store(f32x4 s1, f32x4 s2, float* red, int tid, int QW, int RPW, int q_lane,
                                                      int r_lane, bool active, float* dst1, float* dst2, int q) {
    __syncthreads();
    if (active) {
        *(f32x4*)(red + (size_t)tid * 8) = s1;
        *(f32x4*)(red + (size_t)tid * 8 + 4) = s2;
    }
    __syncthreads();
    if (active && r_lane == 0) {
        for (int r = 1; r < RPW; ++r) {                    // fixed order
            s1 += *(const f32x4*)(red + (size_t)(r * QW + q_lane) * 8);
            s2 += *(const f32x4*)(red + (size_t)(r * QW + q_lane) * 8 + 4);
        }
        *(f32x4*)(dst1 + 4 * q) = s1;
        *(f32x4*)(dst2 + 4 * q) = s2;
    }
}

// ------------------------------------------------------------------------------------------------ channels-last rows
// MODE 0: forward statistics (sum(x - x0), sum((x - x0)^2));  MODE 1: backward sums (sum g, sum g * xhat).
// A thread owns one channel quad and walks rows RPW apart; the loop is unrolled 4x (2x in the backward) with the loads of
// all copies issued before the arithmetic: HBM streaming needs several independent 16-byte loads in flight per lane.
template <bool BF16, int MODE>
__global__ void __launch_bounds__(256) bn_reduce_rows_kernel(const BnP p) {
    __shared__ __attribute__((aligned(16))) float red[256 * 8];
    const int tid = threadIdx.x;
    const int QW = min(p.C4, 256), RPW = 256 / QW;
    const int q_lane = tid % QW, r_lane = tid / QW;
    const bool active = r_lane < RPW;
    const long r_begin = (long)blockIdx.x * p.rows_per_blk;
    const long r_end = min(r_begin + p.rows_per_blk, p.rows);
    float* dst1 = p.partial + (size_t)blockIdx.x * 2 * p.C4 * 4;
    float* dst2 = dst1 + p.C4 * 4;
    for (int q = q_lane; q < p.C4; q += QW) {
        f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = {0.f, 0.f, 0.f, 0.f};
        if (active) {
            const size_t xq = (size_t)p.x_coff + 4 * q;
            long r = r_begin + r_lane;
            if (MODE == 0) {
                const f32x4 x0 = ld4<BF16>(p.x, xq);
                f32x4 t1 = {0.f, 0.f, 0.f, 0.f}, t2 = {0.f, 0.f, 0.f, 0.f};
                for (; r + 3L * RPW < r_end; r += 4L * RPW) {
                    const f32x4 v0 = ld4<BF16>(p.x, (size_t)r * p.x_cs + xq);
                    const f32x4 v1 = ld4<BF16>(p.x, (size_t)(r + RPW) * p.x_cs + xq);
                    const f32x4 v2 = ld4<BF16>(p.x, (size_t)(r + 2L * RPW) * p.x_cs + xq);
                    const f32x4 v3 = ld4<BF16>(p.x, (size_t)(r + 3L * RPW) * p.x_cs + xq);
                    const f32x4 d0 = v0 - x0, d1 = v1 - x0, d2 = v2 - x0, d3 = v3 - x0;
                    s1 += d0 + d1;
                    t1 += d2 + d3;
                    s2 += d0 * d0 + d1 * d1;
                    t2 += d2 * d2 + d3 * d3;
                }
                for (; r < r_end; r += RPW) {
                    const f32x4 d = ld4<BF16>(p.x, (size_t)r * p.x_cs + xq) - x0;
                    s1 += d;
                    s2 += d * d;
                }
                s1 += t1;
                s2 += t2;
            } else {
                f32x4 a, b, m, is;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int c = min(4 * q + j, p.C - 1);
                    a[j] = p.a[c]; b[j] = p.b[c]; m[j] = p.mean[c]; is[j] = p.invstd[c];
                }
                const size_t gq = (size_t)p.gy_coff + 4 * q, yq = (size_t)p.y_coff + 4 * q;
                auto one = [&](f32x4 xv, f32x4 g, f32x4 pre) {
                    if (p.act != 0) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) g[j] *= act_bwd(pre[j], p.act, p.slope);
                    }
                    s1 += g;
                    s2 += g * ((xv - m) * is);
                };
                for (; r + RPW < r_end; r += 2L * RPW) {
                    const f32x4 x0v = ld4<BF16>(p.x, (size_t)r * p.x_cs + xq);
                    const f32x4 x1v = ld4<BF16>(p.x, (size_t)(r + RPW) * p.x_cs + xq);
                    const f32x4 g0 = ld4<BF16>(p.gy, (size_t)r * p.gy_cs + gq);
                    const f32x4 g1 = ld4<BF16>(p.gy, (size_t)(r + RPW) * p.gy_cs + gq);
                    f32x4 p0 = x0v * a + b, p1 = x1v * a + b;
                    if (p.act != 0 && p.y != nullptr) {
                        p0 = ld4<BF16>(p.y, (size_t)r * p.y_cs + yq);
                        p1 = ld4<BF16>(p.y, (size_t)(r + RPW) * p.y_cs + yq);
                    }
                    one(x0v, g0, p0);
                    one(x1v, g1, p1);
                }
                for (; r < r_end; r += RPW) {
                    const f32x4 xv = ld4<BF16>(p.x, (size_t)r * p.x_cs + xq);
                    const f32x4 g = ld4<BF16>(p.gy, (size_t)r * p.gy_cs + gq);
                    f32x4 pre = xv * a + b;
                    if (p.act != 0 && p.y != nullptr) pre = ld4<BF16>(p.y, (size_t)r * p.y_cs + yq);
                    one(xv, g, pre);
                }
            }
        }
        reduce_rows_and_store(s1, s2, red, tid, QW, RPW, q_lane, r_lane, active, dst1, dst2, q);
    }
}

// MODE 0: y = act(x a + b [+ res]) [+ res];  MODE 1: gx = g k1 + x k2 + k3 with g = gy act'(.), optionally out2 = g.
// A thread owns one written quad (pads beyond C get zeros) with its coefficients in registers and walks rows RPW apart,
// two rows per iteration.
template <bool BF16, int MODE>
__global__ void __launch_bounds__(256) bn_apply_rows_kernel(const BnP p) {
    const int W4 = p.cw >> 2;                                  // quads written per row
    const int tid = threadIdx.x;
    const int QW = min(W4, 256), RPW = 256 / QW;
    const int q_lane = tid % QW, r_lane = tid / QW;
    if (r_lane >= RPW) return;
    const long r_begin = (long)blockIdx.x * p.rows_per_blk;
    const long r_end = min(r_begin + p.rows_per_blk, p.rows);
    for (int q = q_lane; q < W4; q += QW) {
        const bool live = q < p.C4;
        f32x4 a = {0.f, 0.f, 0.f, 0.f}, b = a, k1 = a, k2 = a, k3 = a, keep = a;
        if (live) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int c = min(4 * q + j, p.C - 1);
                a[j] = p.a[c]; b[j] = p.b[c];
                if (MODE == 1) { k1[j] = p.k1[c]; k2[j] = p.k2[c]; k3[j] = p.k3[c]; }
                keep[j] = 4 * q + j < p.C ? 1.f : 0.f;
            }
        }
        const size_t xq = (size_t)p.x_coff + 4 * q, oq = (size_t)p.out_coff + 4 * q;
        auto row = [&](long r, f32x4 xv, f32x4 second, f32x4 third) {
            // MODE 0: second = res (or 0);  MODE 1: second = gy, third = the pre-activation source (y or unused)
            f32x4 o, o2 = {0.f, 0.f, 0.f, 0.f};
            if (MODE == 0) {
                f32x4 v = xv * a + b;
                if (p.res_first) v += second;
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = act_fwd(v[j], p.act, p.slope);
                if (!p.res_first) v += second;
                o = v * keep;
            } else {
                f32x4 g = second;
                if (p.act != 0) {
                    const f32x4 pre = p.y != nullptr ? third : xv * a + b;
#pragma unroll
                    for (int j = 0; j < 4; ++j) g[j] *= act_bwd(pre[j], p.act, p.slope);
                }
                o = (g * k1 + xv * k2 + k3) * keep;
                o2 = g * keep;
            }
            st4<BF16>(p.out, (size_t)r * p.out_cs + oq, o);
            if (MODE == 1 && p.out2 != nullptr) st4<BF16>(p.out2, (size_t)r * p.out2_cs + p.out2_coff + 4 * q, o2);
        };
        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
        long r = r_begin + r_lane;
        if (!live) {
            for (; r < r_end; r += RPW) {
                st4<BF16>(p.out, (size_t)r * p.out_cs + oq, zero);
                if (MODE == 1 && p.out2 != nullptr) st4<BF16>(p.out2, (size_t)r * p.out2_cs + p.out2_coff + 4 * q, zero);
            }
            continue;
        }
        for (; r + RPW < r_end; r += 2L * RPW) {
            const long r1 = r + RPW;
            const f32x4 x0v = ld4<BF16>(p.x, (size_t)r * p.x_cs + xq), x1v = ld4<BF16>(p.x, (size_t)r1 * p.x_cs + xq);
            f32x4 s0 = zero, s1v = zero, t0 = zero, t1 = zero;
            if (MODE == 0) {
                if (p.res != nullptr) {
                    s0 = ld4<BF16>(p.res, (size_t)r * p.res_cs + p.res_coff + 4 * q);
                    s1v = ld4<BF16>(p.res, (size_t)r1 * p.res_cs + p.res_coff + 4 * q);
                }
            } else {
                s0 = ld4<BF16>(p.gy, (size_t)r * p.gy_cs + p.gy_coff + 4 * q);
                s1v = ld4<BF16>(p.gy, (size_t)r1 * p.gy_cs + p.gy_coff + 4 * q);
                if (p.act != 0 && p.y != nullptr) {
                    t0 = ld4<BF16>(p.y, (size_t)r * p.y_cs + p.y_coff + 4 * q);
                    t1 = ld4<BF16>(p.y, (size_t)r1 * p.y_cs + p.y_coff + 4 * q);
                }
            }
            row(r, x0v, s0, t0);
            row(r1, x1v, s1v, t1);
        }
        for (; r < r_end; r += RPW) {
            const f32x4 xv = ld4<BF16>(p.x, (size_t)r * p.x_cs + xq);
            f32x4 s0 = zero, t0 = zero;
            if (MODE == 0) {
                if (p.res != nullptr) s0 = ld4<BF16>(p.res, (size_t)r * p.res_cs + p.res_coff + 4 * q);
            } else {
                s0 = ld4<BF16>(p.gy, (size_t)r * p.gy_cs + p.gy_coff + 4 * q);
                if (p.act != 0 && p.y != nullptr) t0 = ld4<BF16>(p.y, (size_t)r * p.y_cs + p.y_coff + 4 * q);
            }
            row(r, xv, s0, t0);
        }
    }
}

// ------------------------------------------------------------------------------------------------ NCHW planes (fp32)
__device__ __forceinline__ float block_sum(float v, float* red, int tid) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

// grid (nsplit, C): block (s, c) reduces the span s of every image's plane c.  partial[(s)][2][C4 * 4]
// float4 loads (two in flight per lane) when the plane size is a multiple of 4, scalar otherwise.
template <int MODE>
__global__ void __launch_bounds__(256) bn_reduce_planes_kernel(const BnP p) {
    __shared__ float red[4];
    const int tid = threadIdx.x, c = blockIdx.y;
    long span = (p.S + gridDim.x - 1) / gridDim.x;
    span = (span + 3) & ~3L;
    const long s0 = (long)blockIdx.x * span, s1e = min(s0 + span, p.S);
    const float* x = (const float*)p.x;
    const float* gy = (const float*)p.gy;
    const float* y = (const float*)p.y;
    float a1 = 0.f, a2 = 0.f;
    const float x0 = MODE == 0 ? x[(size_t)c * p.S] : 0.f;
    const float a = MODE == 1 ? p.a[c] : 0.f, b = MODE == 1 ? p.b[c] : 0.f;
    const float m = MODE == 1 ? p.mean[c] : 0.f, is = MODE == 1 ? p.invstd[c] : 0.f;
    const bool use_y = MODE == 1 && p.act != 0 && y != nullptr;
    auto one = [&](float xv, float g, float yv) {
        if (MODE == 0) {
            const float d = xv - x0;
            a1 += d;
            a2 += d * d;
        } else {
            if (p.act != 0) g *= act_bwd(use_y ? yv : xv * a + b, p.act, p.slope);
            a1 += g;
            a2 += g * ((xv - m) * is);
        }
    };
    const bool vec = (p.S & 3) == 0;
    for (int bi = 0; bi < p.batch; ++bi) {
        const size_t off = ((size_t)bi * p.C + c) * p.S;
        if (vec) {
            const long n4 = (s1e - s0) >> 2;                       // s0 and S are multiples of 4
            const f32x4* x4 = (const f32x4*)(x + off + s0);
            const f32x4* g4 = MODE == 1 ? (const f32x4*)(gy + off + s0) : nullptr;
            const f32x4* y4 = use_y ? (const f32x4*)(y + off + s0) : nullptr;
            long i = tid;
            for (; i + 256 < n4; i += 512) {
                const f32x4 xa = x4[i], xb = x4[i + 256];
                f32x4 ga = xa, gb = xb, ya = xa, yb = xb;
                if (MODE == 1) { ga = g4[i]; gb = g4[i + 256]; }
                if (use_y) { ya = y4[i]; yb = y4[i + 256]; }
#pragma unroll
                for (int j = 0; j < 4; ++j) { one(xa[j], ga[j], ya[j]); one(xb[j], gb[j], yb[j]); }
            }
            for (; i < n4; i += 256) {
                const f32x4 xa = x4[i];
                f32x4 ga = xa, ya = xa;
                if (MODE == 1) ga = g4[i];
                if (use_y) ya = y4[i];
#pragma unroll
                for (int j = 0; j < 4; ++j) one(xa[j], ga[j], ya[j]);
            }
        } else {
            for (long i = s0 + tid; i < s1e; i += 256)
                one(x[off + i], MODE == 1 ? gy[off + i] : 0.f, use_y ? y[off + i] : 0.f);
        }
    }
    const float t1 = block_sum(a1, red, tid);
    const float t2 = block_sum(a2, red, tid);
    if (tid == 0) {
        float* dst = p.partial + (size_t)blockIdx.x * 2 * p.C4 * 4;
        dst[c] = t1;
        dst[p.C4 * 4 + c] = t2;
    }
}

// one (b, c) plane per blockIdx.y, grid-stride over the plane
template <int MODE>
__global__ void __launch_bounds__(256) bn_apply_planes_kernel(const BnP p) {
    const int plane = blockIdx.y;
    const int c = plane % p.C;
    const size_t off = (size_t)plane * p.S;
    const float* x = (const float*)p.x;
    const float a = p.a[c], b = p.b[c];
    const bool vec = (p.S & 3) == 0;
    if (MODE == 0) {
        const float* res = (const float*)p.res;
        float* out = (float*)p.out;
        if (vec) {
            for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < (p.S >> 2); i += (long)gridDim.x * 256) {
                f32x4 v = *(const f32x4*)(x + off + 4 * i) * a + b;
                f32x4 rr = {0.f, 0.f, 0.f, 0.f};
                if (res != nullptr) rr = *(const f32x4*)(res + off + 4 * i);
                if (p.res_first) v += rr;
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = act_fwd(v[j], p.act, p.slope);
                if (!p.res_first) v += rr;
                *(f32x4*)(out + off + 4 * i) = v;
            }
        } else {
            for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < p.S; i += (long)gridDim.x * 256) {
                float v = x[off + i] * a + b;
                const float rr = res != nullptr ? res[off + i] : 0.f;
                if (p.res_first) v += rr;
                v = act_fwd(v, p.act, p.slope);
                if (!p.res_first) v += rr;
                out[off + i] = v;
            }
        }
    } else {
        const float* gy = (const float*)p.gy;
        const float* y = (const float*)p.y;
        float* out = (float*)p.out;
        float* out2 = (float*)p.out2;
        const float k1 = p.k1[c], k2 = p.k2[c], k3 = p.k3[c];
        const bool use_y = p.act != 0 && y != nullptr;
        if (vec) {
            for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < (p.S >> 2); i += (long)gridDim.x * 256) {
                const f32x4 xv = *(const f32x4*)(x + off + 4 * i);
                f32x4 g = *(const f32x4*)(gy + off + 4 * i);
                if (p.act != 0) {
                    const f32x4 pre = use_y ? *(const f32x4*)(y + off + 4 * i) : xv * a + b;
#pragma unroll
                    for (int j = 0; j < 4; ++j) g[j] *= act_bwd(pre[j], p.act, p.slope);
                }
                *(f32x4*)(out + off + 4 * i) = g * k1 + xv * k2 + k3;
                if (out2 != nullptr) *(f32x4*)(out2 + off + 4 * i) = g;
            }
        } else {
            for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < p.S; i += (long)gridDim.x * 256) {
                const float xv = x[off + i];
                float g = gy[off + i];
                if (p.act != 0) g *= act_bwd(use_y ? y[off + i] : xv * a + b, p.act, p.slope);
                out[off + i] = g * k1 + xv * k2 + k3;
                if (out2 != nullptr) out2[off + i] = g;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ per-channel stages
// Sum of the nblk partial pairs of channel c: 256 threads = 8 channels x 32 block-lanes, every lane adds its blocks in
// ascending order, the 32 lanes are combined in a fixed order through LDS (deterministic).  A serial walk over up to 2048
// partials per channel was 80 us of dependent loads per layer -- more than the passes over the activation themselves.
__device__ __forceinline__ bool partial_sums(const float* partial, int nblk, int Cp, int C, double* red, double& s1, double& s2,
                                             int& c_out) {
    const int tid = threadIdx.x, cl = tid & 7, kl = tid >> 3;
    const int c = blockIdx.x * 8 + cl;
    double a1 = 0.0, a2 = 0.0;
    if (c < C)
        for (int k = kl; k < nblk; k += 32) {
            a1 += (double)partial[(size_t)k * 2 * Cp + c];
            a2 += (double)partial[(size_t)k * 2 * Cp + Cp + c];
        }
    red[tid * 2] = a1;
    red[tid * 2 + 1] = a2;
    __syncthreads();
    c_out = c;
    if (kl != 0 || c >= C) return false;
    s1 = 0.0; s2 = 0.0;
    for (int k = 0; k < 32; ++k) {
        s1 += red[(k * 8 + cl) * 2];
        s2 += red[(k * 8 + cl) * 2 + 1];
    }
    return true;
}

// partial[nblk][2][Cp] -> packed (float64): forward [n mean_l, M2_l + n mean_l^2, n] (Chan's form, see shard.py) from the
// shifted sums; x0[c] is re-read from the tensor exactly as the statistics pass read it.
__global__ void bn_stats_combine_kernel(const float* partial, int nblk, int Cp, int C, const void* x, int dtype, int layout,
                                        long S, int x_coff, double n_local, double* packed) {
    __shared__ double red[512];
    double s1, s2;
    int c;
    if (blockIdx.x == 0 && threadIdx.x == 0) packed[2 * C] = n_local;
    if (!partial_sums(partial, nblk, Cp, C, red, s1, s2, c)) return;
    double x0;
    if (layout == 1) x0 = (double)((const float*)x)[(size_t)c * S];
    else if (dtype == 1) x0 = (double)__builtin_bit_cast(float, (uint32_t)((const uint16_t*)x)[x_coff + c] << 16);
    else x0 = (double)((const float*)x)[x_coff + c];
    const double mean = x0 + s1 / n_local;
    const double m2 = s2 - s1 * s1 / n_local;
    packed[c] = n_local * mean;
    packed[C + c] = m2 + n_local * mean * mean;
}

__global__ void bn_finish_kernel(const double* packed, int C, float eps, float momentum, const float* gamma, const float* beta,
                                 float* running_mean, float* running_var, long* num_batches, float* mean, float* invstd,
                                 float* a, float* b) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c == 0 && num_batches != nullptr) *num_batches += 1;
    if (c >= C) return;
    const double n = packed[2 * C];
    const double mu = packed[c] / n;
    double var = packed[C + c] / n - mu * mu;
    if (var < 0.0) var = 0.0;
    const float is = (float)(1.0 / sqrt(var + (double)eps));
    mean[c] = (float)mu;
    invstd[c] = is;
    const float g = gamma != nullptr ? gamma[c] : 1.f;
    const float sc = g * is;
    a[c] = sc;
    b[c] = (beta != nullptr ? beta[c] : 0.f) - (float)mu * sc;
    if (running_mean != nullptr) {
        const double unbiased = var * (n / (n > 1.0 ? n - 1.0 : 1.0));
        running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mu;
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
    }
}

__global__ void bn_bwd_combine_kernel(const float* partial, int nblk, int Cp, int C, float* packed) {
    __shared__ double red[512];
    double s1, s2;
    int c;
    if (!partial_sums(partial, nblk, Cp, C, red, s1, s2, c)) return;
    packed[c] = (float)s1;
    packed[C + c] = (float)s2;
}

// local = this rank's [sum g, sum g xhat] (parameter gradients stay per-rank sums: the gradient buckets average them like
// any other), total = the same summed over the ranks that share the statistics.
__global__ void bn_bwd_finish_kernel(const float* local, const float* total, int C, const double* packed_fwd, const float* mean,
                                     const float* invstd, const float* a, float* k1, float* k2, float* k3, float* gw,
                                     float* gb) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float n = (float)packed_fwd[2 * C];
    const float mg = total[c] / n, mgx = total[C + c] / n;
    const float sc = a[c], is = invstd[c], mu = mean[c];
    k1[c] = sc;
    k2[c] = -is * mgx * sc;
    k3[c] = (mu * is * mgx - mg) * sc;
    if (gw != nullptr) gw[c] = local[C + c];
    if (gb != nullptr) gb[c] = local[c];
}

// ---- local statistics (no exchange between ranks): combine + finish in ONE per-channel launch
__global__ void bn_combine_finish_kernel(const float* partial, int nblk, int Cp, int C, const void* x, int dtype, int layout,
                                         long S, int x_coff, double n_local, double* packed, float eps, float momentum,
                                         const float* gamma, const float* beta, float* running_mean, float* running_var,
                                         long* num_batches, float* mean, float* invstd, float* a, float* b) {
    __shared__ double red[512];
    double s1, s2;
    int c;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        packed[2 * C] = n_local;
        if (num_batches != nullptr) *num_batches += 1;
    }
    if (!partial_sums(partial, nblk, Cp, C, red, s1, s2, c)) return;
    double x0;
    if (layout == 1) x0 = (double)((const float*)x)[(size_t)c * S];
    else if (dtype == 1) x0 = (double)__builtin_bit_cast(float, (uint32_t)((const uint16_t*)x)[x_coff + c] << 16);
    else x0 = (double)((const float*)x)[x_coff + c];
    const double mu = x0 + s1 / n_local;
    double var = (s2 - s1 * s1 / n_local) / n_local;
    if (var < 0.0) var = 0.0;
    packed[c] = n_local * mu;
    packed[C + c] = n_local * (var + mu * mu);
    const float is = (float)(1.0 / sqrt(var + (double)eps));
    mean[c] = (float)mu;
    invstd[c] = is;
    const float sc = (gamma != nullptr ? gamma[c] : 1.f) * is;
    a[c] = sc;
    b[c] = (beta != nullptr ? beta[c] : 0.f) - (float)mu * sc;
    if (running_mean != nullptr) {
        const double unbiased = var * (n_local / (n_local > 1.0 ? n_local - 1.0 : 1.0));
        running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mu;
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
    }
}

__global__ void bn_bwd_combine_finish_kernel(const float* partial, int nblk, int Cp, int C, const double* packed_fwd,
                                             const float* mean, const float* invstd, const float* a, float* k1, float* k2,
                                             float* k3, float* gw, float* gb) {
    __shared__ double red[512];
    double s1, s2;
    int c;
    if (!partial_sums(partial, nblk, Cp, C, red, s1, s2, c)) return;
    const float n = (float)packed_fwd[2 * C];
    const float mg = (float)s1 / n, mgx = (float)s2 / n;
    const float sc = a[c], is = invstd[c], mu = mean[c];
    k1[c] = sc;
    k2[c] = -is * mgx * sc;
    k3[c] = (mu * is * mgx - mg) * sc;
    if (gw != nullptr) gw[c] = (float)s2;
    if (gb != nullptr) gb[c] = (float)s1;
}

// ---- small NCHW tensors (the EfficientNet stages at 1/8 ... 1/32 resolution: a few thousand pixels per plane, up to 3840
// channels, ~220 BatchNorm calls per step): the whole layer in ONE launch, one workgroup per channel -- reduction pass,
// per-channel finish in the workgroup, apply pass (the plane comes back from L2).  Local statistics only.
// XCHG (round 5, SyncBatchNorm over several ranks of one node): the per-channel statistics are exchanged INSIDE this launch
// through peer-mapped mailboxes (the fence-free "LL" protocol of csrc/ipc_allreduce.hip: every 32-bit half of the channel's
// three doubles travels in its own 8-byte word {data, sequence}, one 64-byte record per (slot, rank, channel)): thread 0 ..
// world-1 of workgroup c push the channel's packet into every rank's mailbox and poll every rank's packet of the same
// sequence number; the sums are taken in rank order -- so a synchronised small layer stays ONE launch per direction (the
// separate path is statistics + combine + all-reduce + finish + apply: five).  Workgroups of one launch are independent of
// each other; across ranks workgroup c only waits for the peers' workgroup c, which never waits for anything this rank has
// not already pushed (pushes precede polls), so the exchange cannot deadlock as long as every rank's workgroups get
// scheduled.  Every CHANNEL keeps its own sequence counter in the owner's mailbox (read and advanced by workgroup c alone;
// all ranks run the same layers, so the per-channel sequences agree across ranks; stream order separates launches), so
// captured launches replay correctly and no two workgroups ever touch the same word.  No fence and no shared atomic
// anywhere -- measured on the forced single-rank config-2 step (105 ms without exchanges, 125 ms on the five-launch path over
// RCCL): two system-scope fences per workgroup (an L2 write-back + invalidate each) 172 ms; a launch-wide sequence number
// advanced by the last workgroup through one atomic counter (thousands of atomics on one address per launch) 135 ms.
struct BnXchg {
    unsigned char* mbox[16];     // peer-mapped channel mailboxes, [rank] = own
    int rank, world;
    long slot_bytes, row_bytes;  // slot = world rows, row = cmax records of 64 bytes
    long slots_off;              // byte offset of slot 0 = kXchgHeader + 8 cmax
    long long timeout_ticks;     // wall_clock64 ticks (100 MHz); <= 0: unbounded
    int* status;                 // device int, set to 1 when a poll gives up
};
constexpr int kXchgHeader = 256;  // bytes (reserved); then cmax u64 per-channel sequence counters; then the two slots
constexpr int kXchgRec = 64;      // bytes per (slot, rank, channel) record: 6 LL words + pad

// sums over the ranks of (v0, v1, v2) for channel c.  Called by ALL threads of the workgroup.  A peer whose packet does not
// arrive within the budget makes the sums NaN (the layer's output and gradients are then NaN: loud, not a wrong normalisation).
__device__ __forceinline__ void xchg_channel(const BnXchg& q, int c, double v0, double v1, double v2, double (&tot)[3],
                                             double* sh) {
    // channel c's own sequence counter (this workgroup is its only reader and writer; every lane reads the same word)
    unsigned long long* ctr = reinterpret_cast<unsigned long long*>(q.mbox[q.rank] + kXchgHeader) + c;
    const unsigned long long seq = __hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1;
    const unsigned seq32 = (unsigned)seq;
    const int slot = (int)(seq & 1);
    const int t = threadIdx.x;
    if (t < q.world) {
        const size_t rec_off = (size_t)q.slots_off + (size_t)slot * q.slot_bytes + (size_t)c * kXchgRec;
        unsigned long long* dst = reinterpret_cast<unsigned long long*>(q.mbox[t] + rec_off + (size_t)q.rank * q.row_bytes);
        const double v[3] = {v0, v1, v2};
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const unsigned long long bits = (unsigned long long)__double_as_longlong(v[j]);
            __hip_atomic_store(dst + 2 * j, (bits & 0xffffffffULL) | ((unsigned long long)seq32 << 32), __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(dst + 2 * j + 1, (bits >> 32) | ((unsigned long long)seq32 << 32), __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_SYSTEM);
        }
        const unsigned long long* src = reinterpret_cast<const unsigned long long*>(q.mbox[q.rank] + rec_off + (size_t)t * q.row_bytes);
        // all six words in flight at once (an access past the caches is a ~2 us round trip: polled one after the other the
        // exchange cost six of them), re-polled together until every one carries the sequence number
        bool fail = false;
        const long long t0 = wall_clock64();
        unsigned long long w[6];
        while (true) {
#pragma unroll
            for (int j = 0; j < 6; ++j) w[j] = __hip_atomic_load(src + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            bool ready = true;
#pragma unroll
            for (int j = 0; j < 6; ++j) ready = ready && (unsigned)(w[j] >> 32) == seq32;
            if (ready) break;
            if (q.timeout_ticks > 0 && wall_clock64() - t0 > q.timeout_ticks) { fail = true; break; }
            __builtin_amdgcn_s_sleep(1);
        }
#pragma unroll
        for (int j = 0; j < 3; ++j)
            sh[t * 3 + j] = fail ? (double)__builtin_nanf("") : __longlong_as_double((long long)((w[2 * j + 1] << 32) | (w[2 * j] & 0xffffffffULL)));
        if (fail && q.status != nullptr) *q.status = 1;
    }
    __syncthreads();                              // (every lane has read the counter before it advances)
    if (t == 0) __hip_atomic_store(ctr, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    tot[0] = tot[1] = tot[2] = 0.0;
    for (int r = 0; r < q.world; ++r) {          // rank order: the same sum, bit for bit, on every rank
        tot[0] += sh[r * 3]; tot[1] += sh[r * 3 + 1]; tot[2] += sh[r * 3 + 2];
    }
}

template <int MODE, bool XCHG = false>
__global__ void __launch_bounds__(256) bn_small_planes_kernel(const BnP p, float eps, float momentum, const float* gamma,
                                                              const float* beta, float* running_mean, float* running_var,
                                                              long* num_batches, double* packed, float* mean_o,
                                                              float* invstd_o, float* a_o, float* b_o, float* gw, float* gb,
                                                              const BnXchg xq) {
    __shared__ float red[4];
    __shared__ float coef[4];
    __shared__ double xsh[16 * 3];
    const int tid = threadIdx.x, c = blockIdx.x;
    const float* x = (const float*)p.x;
    const float* gy = (const float*)p.gy;
    const float* y = (const float*)p.y;
    const bool vec = (p.S & 3) == 0;
    const double n = (double)p.batch * (double)p.S;
    float a1 = 0.f, a2 = 0.f;
    const float x0 = MODE == 0 ? x[(size_t)c * p.S] : 0.f;
    float a = 0.f, b = 0.f, m = 0.f, is = 0.f;
    if (MODE == 1) { a = p.a[c]; b = p.b[c]; m = p.mean[c]; is = p.invstd[c]; }
    const bool use_y = MODE == 1 && p.act != 0 && y != nullptr;
    auto acc1 = [&](float xv, float g, float yv) {
        if (MODE == 0) {
            const float d = xv - x0;
            a1 += d;
            a2 += d * d;
        } else {
            if (p.act != 0) g *= act_bwd(use_y ? yv : xv * a + b, p.act, p.slope);
            a1 += g;
            a2 += g * ((xv - m) * is);
        }
    };
    for (int bi = 0; bi < p.batch; ++bi) {
        const size_t off = ((size_t)bi * p.C + c) * p.S;
        if (vec) {
            const f32x4* x4 = (const f32x4*)(x + off);
            const f32x4* g4 = MODE == 1 ? (const f32x4*)(gy + off) : nullptr;
            const f32x4* y4 = use_y ? (const f32x4*)(y + off) : nullptr;
            for (long i = tid; i < (p.S >> 2); i += 256) {
                const f32x4 xa = x4[i];
                f32x4 ga = xa, ya = xa;
                if (MODE == 1) ga = g4[i];
                if (use_y) ya = y4[i];
#pragma unroll
                for (int j = 0; j < 4; ++j) acc1(xa[j], ga[j], ya[j]);
            }
        } else {
            for (long i = tid; i < p.S; i += 256) acc1(x[off + i], MODE == 1 ? gy[off + i] : 0.f, use_y ? y[off + i] : 0.f);
        }
    }
    const float t1 = block_sum(a1, red, tid);
    const float t2 = block_sum(a2, red, tid);
    double xt[3] = {0.0, 0.0, 0.0};
    if (XCHG) {
        if (MODE == 0) {
            // this rank's packet in Chan's form (shard.py): [n mean, M2 + n mean^2, n], centred on the rank's own mean
            const double mu_l = (double)x0 + (double)t1 / n;
            const double m2_l = (double)t2 - (double)t1 * (double)t1 / n;
            xchg_channel(xq, c, n * mu_l, m2_l + n * mu_l * mu_l, n, xt, xsh);
        } else {
            xchg_channel(xq, c, (double)t1, (double)t2, 0.0, xt, xsh);
        }
    }
    if (tid == 0) {
        if (MODE == 0) {
            double mu = (double)x0 + (double)t1 / n;
            double var = ((double)t2 - (double)t1 * (double)t1 / n) / n;
            double n_tot = n;
            if (XCHG) {
                n_tot = xt[2];
                mu = xt[0] / n_tot;
                var = xt[1] / n_tot - mu * mu;
            }
            if (var < 0.0) var = 0.0;
            const float isd = (float)(1.0 / sqrt(var + (double)eps));
            const float sc = (gamma != nullptr ? gamma[c] : 1.f) * isd;
            const float sh = (beta != nullptr ? beta[c] : 0.f) - (float)mu * sc;
            packed[c] = XCHG ? xt[0] : n * mu;
            packed[p.C + c] = XCHG ? xt[1] : n * (var + mu * mu);
            if (c == 0) {
                packed[2 * p.C] = n_tot;
                if (num_batches != nullptr) *num_batches += 1;
            }
            mean_o[c] = (float)mu; invstd_o[c] = isd; a_o[c] = sc; b_o[c] = sh;
            if (running_mean != nullptr) {
                const double unbiased = var * (n_tot / (n_tot > 1.0 ? n_tot - 1.0 : 1.0));
                running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mu;
                running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
            }
            coef[0] = sc; coef[1] = sh;
        } else {
            // (synchronised: the batch statistics span the ranks -- totals over n_total = packed_fwd[2C], handed over in
            //  `packed`; the parameter gradients stay this rank's sums, the gradient buckets average them)
            const float nf = XCHG ? (float)packed[2 * p.C] : (float)n;
            const float mg = (XCHG ? (float)xt[0] : t1) / nf, mgx = (XCHG ? (float)xt[1] : t2) / nf;
            coef[0] = a;
            coef[1] = -is * mgx * a;
            coef[2] = (m * is * mgx - mg) * a;
            if (gw != nullptr) gw[c] = t2;
            if (gb != nullptr) gb[c] = t1;
        }
    }
    __syncthreads();
    const float c0 = coef[0], c1 = coef[1], c2 = MODE == 1 ? coef[2] : 0.f;
    const float* res = (const float*)p.res;
    float* out = (float*)p.out;
    float* out2 = (float*)p.out2;
    for (int bi = 0; bi < p.batch; ++bi) {
        const size_t off = ((size_t)bi * p.C + c) * p.S;
        const long n4 = vec ? (p.S >> 2) : 0;
        for (long i = tid; i < n4; i += 256) {
            const f32x4 xv = *(const f32x4*)(x + off + 4 * i);
            if (MODE == 0) {
                f32x4 v = xv * c0 + c1;
                f32x4 rr = {0.f, 0.f, 0.f, 0.f};
                if (res != nullptr) rr = *(const f32x4*)(res + off + 4 * i);
                if (p.res_first) v += rr;
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = act_fwd(v[j], p.act, p.slope);
                if (!p.res_first) v += rr;
                *(f32x4*)(out + off + 4 * i) = v;
            } else {
                f32x4 g = *(const f32x4*)(gy + off + 4 * i);
                if (p.act != 0) {
                    const f32x4 pre = use_y ? *(const f32x4*)(y + off + 4 * i) : xv * a + b;
#pragma unroll
                    for (int j = 0; j < 4; ++j) g[j] *= act_bwd(pre[j], p.act, p.slope);
                }
                *(f32x4*)(out + off + 4 * i) = g * c0 + xv * c1 + c2;
                if (out2 != nullptr) *(f32x4*)(out2 + off + 4 * i) = g;
            }
        }
        for (long i = 4 * n4 + tid; i < p.S; i += 256) {
            const float xv = x[off + i];
            if (MODE == 0) {
                float v = xv * c0 + c1;
                const float rr = res != nullptr ? res[off + i] : 0.f;
                if (p.res_first) v += rr;
                v = act_fwd(v, p.act, p.slope);
                if (!p.res_first) v += rr;
                out[off + i] = v;
            } else {
                float g = gy[off + i];
                if (p.act != 0) g *= act_bwd(use_y ? y[off + i] : xv * a + b, p.act, p.slope);
                out[off + i] = g * c0 + xv * c1 + c2;
                if (out2 != nullptr) out2[off + i] = g;
            }
        }
    }
}

int fill(const occd_bn_args* a, BnP& p, bool need_partial) {
    if (a == nullptr || a->x == nullptr || a->C <= 0 || a->dtype < 0 || a->dtype > 1 || a->layout < 0 || a->layout > 1)
        return OCCD_EINVAL;
    if (a->layout == 1 && a->dtype != 0) return OCCD_EINVAL;          // NCHW planes: fp32 only
    p.x = a->x; p.gy = a->gy; p.y = a->y; p.res = a->res; p.out = a->out; p.out2 = a->out2;
    p.a = a->a; p.b = a->b; p.mean = a->mean; p.invstd = a->invstd; p.k1 = a->k1; p.k2 = a->k2; p.k3 = a->k3;
    p.partial = a->partial;
    p.rows = a->rows; p.S = a->S; p.batch = a->batch;
    p.C = a->C; p.C4 = (a->C + 3) / 4;
    p.cw = a->cw;
    p.x_cs = a->x_cs; p.x_coff = a->x_coff; p.gy_cs = a->gy_cs; p.gy_coff = a->gy_coff; p.y_cs = a->y_cs; p.y_coff = a->y_coff;
    p.res_cs = a->res_cs; p.res_coff = a->res_coff; p.out_cs = a->out_cs; p.out_coff = a->out_coff;
    p.out2_cs = a->out2_cs; p.out2_coff = a->out2_coff;
    p.act = a->act; p.res_first = a->res_first; p.slope = a->slope;
    if (a->act < 0 || a->act > 3) return OCCD_EINVAL;
    if (a->layout == 0) {
        if (a->rows <= 0) return OCCD_EINVAL;
        const int al = a->dtype == 1 ? 3 : 3;                           // 4-channel accesses: 8 B (bf16) / 16 B (fp32)
        if ((a->x_cs & al) || (a->x_coff & al) || a->x_coff + p.C4 * 4 > a->x_cs) return OCCD_EINVAL;
    } else {
        if (a->batch <= 0 || a->S <= 0) return OCCD_EINVAL;
    }
    if (need_partial && (a->partial == nullptr || a->nblk <= 0 || a->nblk > kMaxBlocks)) return OCCD_EINVAL;
    p.nblk = a->nblk;
    p.rows_per_blk = a->layout == 0 ? (a->rows + a->nblk - 1) / (a->nblk > 0 ? a->nblk : 1) : 0;
    return OCCD_OK;
}

// grid of the rows apply kernels: >= 8 rows per row-lane, at most 4096 workgroups; sets the block's row range
long apply_blocks(const occd_bn_args* a, BnP& p) {
    const int W4 = a->cw >> 2;
    const int QW = W4 < 256 ? W4 : 256, RPW = 256 / QW;
    long blocks = (a->rows + (long)RPW * 8 - 1) / ((long)RPW * 8);
    if (blocks < 1) blocks = 1;
    if (blocks > 4096) blocks = 4096;
    p.rows_per_blk = (a->rows + blocks - 1) / blocks;
    return blocks;
}

bool rows_tensor_ok(const void* ptr, int cs, int coff, int width) {
    return ptr == nullptr || ((cs & 3) == 0 && (coff & 3) == 0 && coff + width <= cs);
}

}  // namespace

extern "C" {

/* number of partial blocks the reduction passes of this geometry use (size of `partial`: nblk * 2 * ceil4(C) floats) */
int occd_bn_blocks(const occd_bn_args* a) {
    if (a == nullptr || a->C <= 0) return OCCD_EINVAL;
    if (a->layout == 0) {
        const int C4 = (a->C + 3) / 4;
        const int QW = C4 < 256 ? C4 : 256, RPW = 256 / QW;
        long n = (a->rows + (long)RPW * 16 - 1) / ((long)RPW * 16);     // >= 16 rows per row-lane
        if (n < 1) n = 1;
        if (n > kMaxBlocks) n = kMaxBlocks;
        return (int)n;
    }
    long want = 2048 / a->C;                                          // ~2048 workgroups over the chip
    long cap = (a->S + 1023) / 1024;                                  // >= 1024 elements of a plane per block
    if (want > cap) want = cap;
    if (want < 1) want = 1;
    if (want > 64) want = 64;
    return (int)want;
}

int occd_bn_stats(const occd_bn_args* a, void* stream) {
    BnP p{};
    const int rc = fill(a, p, true);
    if (rc != OCCD_OK) return rc;
    hipStream_t st = (hipStream_t)stream;
    const double bytes = (a->dtype == 1 ? 2.0 : 4.0) * (a->layout == 0 ? (double)a->rows * a->C : (double)a->batch * a->C * a->S);
    occd::ProfScope prof("bn_stats", st, 0.0, bytes);
    if (a->layout == 0) {
        if (a->dtype == 1) hipLaunchKernelGGL((bn_reduce_rows_kernel<true, 0>), dim3(a->nblk), dim3(256), 0, st, p);
        else hipLaunchKernelGGL((bn_reduce_rows_kernel<false, 0>), dim3(a->nblk), dim3(256), 0, st, p);
    } else {
        hipLaunchKernelGGL((bn_reduce_planes_kernel<0>), dim3(a->nblk, a->C), dim3(256), 0, st, p);
    }
    return occd::check_launch();
}

int occd_bn_stats_combine(const occd_bn_args* a, double* packed, void* stream) {
    BnP p{};
    const int rc = fill(a, p, true);
    if (rc != OCCD_OK || packed == nullptr) return rc != OCCD_OK ? rc : OCCD_EINVAL;
    const double n = a->layout == 0 ? (double)a->rows : (double)a->batch * (double)a->S;
    hipLaunchKernelGGL(bn_stats_combine_kernel, dim3((a->C + 7) / 8), dim3(256), 0, (hipStream_t)stream,
                       (const float*)a->partial, a->nblk, p.C4 * 4, a->C, a->x, a->dtype, a->layout, (long)a->S, a->x_coff, n,
                       packed);
    return occd::check_launch();
}

int occd_bn_finish(const double* packed, int32_t C, float eps, float momentum, const float* gamma, const float* beta,
                   float* running_mean, float* running_var, int64_t* num_batches_tracked, float* mean, float* invstd, float* a,
                   float* b, void* stream) {
    if (!packed || C <= 0 || !mean || !invstd || !a || !b || (running_mean == nullptr) != (running_var == nullptr))
        return OCCD_EINVAL;
    hipLaunchKernelGGL(bn_finish_kernel, dim3((C + 127) / 128), dim3(128), 0, (hipStream_t)stream, packed, C, eps, momentum,
                       gamma, beta, running_mean, running_var, (long*)num_batches_tracked, mean, invstd, a, b);
    return occd::check_launch();
}

int occd_bn_apply(const occd_bn_args* a, void* stream) {
    BnP p{};
    const int rc = fill(a, p, false);
    if (rc != OCCD_OK) return rc;
    if (!a->out || !a->a || !a->b) return OCCD_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const double elems = a->layout == 0 ? (double)a->rows * a->C : (double)a->batch * a->C * a->S;
    occd::ProfScope prof("bn_apply", st, 0.0, (a->dtype == 1 ? 2.0 : 4.0) * elems * (2 + (a->res != nullptr)));
    if (a->layout == 0) {
        if (a->cw < p.C4 * 4 || (a->cw & 3) || !rows_tensor_ok(a->out, a->out_cs, a->out_coff, a->cw) ||
            !rows_tensor_ok(a->res, a->res_cs, a->res_coff, p.C4 * 4))
            return OCCD_EINVAL;
        const long blocks = apply_blocks(a, p);
        if (a->dtype == 1) hipLaunchKernelGGL((bn_apply_rows_kernel<true, 0>), dim3((unsigned)blocks), dim3(256), 0, st, p);
        else hipLaunchKernelGGL((bn_apply_rows_kernel<false, 0>), dim3((unsigned)blocks), dim3(256), 0, st, p);
    } else {
        const long planes = (long)a->batch * a->C;
        if (planes > 65535) return OCCD_EINVAL;
        long bx = (a->S / 4 + 255) / 256;
        if (bx < 1) bx = 1;
        if (bx > 64) bx = 64;
        hipLaunchKernelGGL((bn_apply_planes_kernel<0>), dim3((unsigned)bx, (unsigned)planes), dim3(256), 0, st, p);
    }
    return occd::check_launch();
}

int occd_bn_bwd_reduce(const occd_bn_args* a, void* stream) {
    BnP p{};
    const int rc = fill(a, p, true);
    if (rc != OCCD_OK) return rc;
    if (!a->gy || !a->a || !a->b || !a->mean || !a->invstd) return OCCD_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const double elems = a->layout == 0 ? (double)a->rows * a->C : (double)a->batch * a->C * a->S;
    occd::ProfScope prof("bn_bwd_reduce", st, 0.0, (a->dtype == 1 ? 2.0 : 4.0) * elems * (2 + (a->y != nullptr)));
    if (a->layout == 0) {
        if (!rows_tensor_ok(a->gy, a->gy_cs, a->gy_coff, p.C4 * 4) || !rows_tensor_ok(a->y, a->y_cs, a->y_coff, p.C4 * 4))
            return OCCD_EINVAL;
        if (a->dtype == 1) hipLaunchKernelGGL((bn_reduce_rows_kernel<true, 1>), dim3(a->nblk), dim3(256), 0, st, p);
        else hipLaunchKernelGGL((bn_reduce_rows_kernel<false, 1>), dim3(a->nblk), dim3(256), 0, st, p);
    } else {
        hipLaunchKernelGGL((bn_reduce_planes_kernel<1>), dim3(a->nblk, a->C), dim3(256), 0, st, p);
    }
    return occd::check_launch();
}

int occd_bn_bwd_combine(const float* partial, int32_t nblk, int32_t C, float* packed, void* stream) {
    if (!partial || !packed || nblk <= 0 || C <= 0) return OCCD_EINVAL;
    hipLaunchKernelGGL(bn_bwd_combine_kernel, dim3((C + 7) / 8), dim3(256), 0, (hipStream_t)stream, partial, nblk,
                       ((C + 3) / 4) * 4, C, packed);
    return occd::check_launch();
}

int occd_bn_bwd_finish(const float* local, const float* total, int32_t C, const double* packed_fwd, const float* mean,
                       const float* invstd, const float* a, float* k1, float* k2, float* k3, float* gw, float* gb,
                       void* stream) {
    if (!local || !total || C <= 0 || !packed_fwd || !mean || !invstd || !a || !k1 || !k2 || !k3) return OCCD_EINVAL;
    hipLaunchKernelGGL(bn_bwd_finish_kernel, dim3((C + 127) / 128), dim3(128), 0, (hipStream_t)stream, local, total, C,
                       packed_fwd, mean, invstd, a, k1, k2, k3, gw, gb);
    return occd::check_launch();
}

int occd_bn_bwd_apply(const occd_bn_args* a, void* stream) {
    BnP p{};
    const int rc = fill(a, p, false);
    if (rc != OCCD_OK) return rc;
    if (!a->out || !a->gy || !a->a || !a->b || !a->k1 || !a->k2 || !a->k3) return OCCD_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const double elems = a->layout == 0 ? (double)a->rows * a->C : (double)a->batch * a->C * a->S;
    occd::ProfScope prof("bn_bwd_apply", st, 0.0,
                         (a->dtype == 1 ? 2.0 : 4.0) * elems * (3 + (a->y != nullptr) + (a->out2 != nullptr)));
    if (a->layout == 0) {
        if (a->cw < p.C4 * 4 || (a->cw & 3) || !rows_tensor_ok(a->out, a->out_cs, a->out_coff, a->cw) ||
            !rows_tensor_ok(a->out2, a->out2_cs, a->out2_coff, a->cw) || !rows_tensor_ok(a->gy, a->gy_cs, a->gy_coff, p.C4 * 4) ||
            !rows_tensor_ok(a->y, a->y_cs, a->y_coff, p.C4 * 4))
            return OCCD_EINVAL;
        const long blocks = apply_blocks(a, p);
        if (a->dtype == 1) hipLaunchKernelGGL((bn_apply_rows_kernel<true, 1>), dim3((unsigned)blocks), dim3(256), 0, st, p);
        else hipLaunchKernelGGL((bn_apply_rows_kernel<false, 1>), dim3((unsigned)blocks), dim3(256), 0, st, p);
    } else {
        const long planes = (long)a->batch * a->C;
        if (planes > 65535) return OCCD_EINVAL;
        long bx = (a->S + 1023) / 1024;
        if (bx < 1) bx = 1;
        if (bx > 64) bx = 64;
        hipLaunchKernelGGL((bn_apply_planes_kernel<1>), dim3((unsigned)bx, (unsigned)planes), dim3(256), 0, st, p);
    }
    return occd::check_launch();
}

/* Local statistics (one rank, or BatchNorm that is not synchronised): occd_bn_stats_combine + occd_bn_finish as ONE launch
 * (`packed` is still written: the backward reads n from it), and occd_bn_bwd_combine + occd_bn_bwd_finish as ONE launch. */
int occd_bn_stats_finish(const occd_bn_args* a, double* packed, float eps, float momentum, const float* gamma,
                         const float* beta, float* running_mean, float* running_var, int64_t* num_batches_tracked,
                         float* mean, float* invstd, float* av, float* bv, void* stream) {
    BnP p{};
    const int rc = fill(a, p, true);
    if (rc != OCCD_OK) return rc;
    if (!packed || !mean || !invstd || !av || !bv || (running_mean == nullptr) != (running_var == nullptr)) return OCCD_EINVAL;
    const double n = a->layout == 0 ? (double)a->rows : (double)a->batch * (double)a->S;
    hipLaunchKernelGGL(bn_combine_finish_kernel, dim3((a->C + 7) / 8), dim3(256), 0, (hipStream_t)stream,
                       (const float*)a->partial, a->nblk, p.C4 * 4, a->C, a->x, a->dtype, a->layout, (long)a->S, a->x_coff, n,
                       packed, eps, momentum, gamma, beta, running_mean, running_var, (long*)num_batches_tracked, mean, invstd,
                       av, bv);
    return occd::check_launch();
}

int occd_bn_bwd_combine_finish(const float* partial, int32_t nblk, int32_t C, const double* packed_fwd, const float* mean,
                               const float* invstd, const float* a, float* k1, float* k2, float* k3, float* gw, float* gb,
                               void* stream) {
    if (!partial || nblk <= 0 || C <= 0 || !packed_fwd || !mean || !invstd || !a || !k1 || !k2 || !k3) return OCCD_EINVAL;
    hipLaunchKernelGGL(bn_bwd_combine_finish_kernel, dim3((C + 7) / 8), dim3(256), 0, (hipStream_t)stream, partial, nblk,
                       ((C + 3) / 4) * 4, C, packed_fwd, mean, invstd, a, k1, k2, k3, gw, gb);
    return occd::check_launch();
}

/* Small NCHW tensors (layout 1): the whole forward (statistics, finish, apply) / the whole backward (reduction, finish,
 * apply) of a layer with LOCAL statistics as one launch, one workgroup per channel.  Worth it while batch * S is a few
 * thousand elements per channel (occd_bn_small_ok).                                                               */
int occd_bn_small_ok(const occd_bn_args* a) {
    return a != nullptr && a->layout == 1 && a->dtype == 0 && a->C >= 64 && (double)a->batch * (double)a->S <= 8192.0 ? 1 : 0;
}

int occd_bn_fwd_small(const occd_bn_args* a, double* packed, float eps, float momentum, const float* gamma, const float* beta,
                      float* running_mean, float* running_var, int64_t* num_batches_tracked, float* mean, float* invstd,
                      float* av, float* bv, void* stream) {
    BnP p{};
    const int rc = fill(a, p, false);
    if (rc != OCCD_OK) return rc;
    if (a->layout != 1 || !a->out || !packed || !mean || !invstd || !av || !bv ||
        (running_mean == nullptr) != (running_var == nullptr))
        return OCCD_EINVAL;
    const double elems = (double)a->batch * a->C * a->S;
    occd::ProfScope prof("bn_fwd_small", (hipStream_t)stream, 0.0, 4.0 * elems * (3 + (a->res != nullptr)));
    hipLaunchKernelGGL((bn_small_planes_kernel<0>), dim3(a->C), dim3(256), 0, (hipStream_t)stream, p, eps, momentum, gamma, beta,
                       running_mean, running_var, (long*)num_batches_tracked, packed, mean, invstd, av, bv, (float*)nullptr,
                       (float*)nullptr, BnXchg{});
    return occd::check_launch();
}

/* The same launches with the statistics of `world` ranks exchanged inside the kernel (see BnXchg): mailboxes = HOST array of
 * `world` peer-mapped channel mailboxes of occd_bn_xchg_mailbox_bytes(world, cmax) bytes (created / opened with the
 * occd_ipc_mailbox_* calls), [rank] = own.  Forward: `packed` receives the TOTALS [sum n mean, sum (M2 + n mean^2), sum n];
 * backward: `packed_fwd` = that vector (n_total), gw / gb stay this rank's sums.                                      */
static int fill_xchg(BnXchg& x, void* const* mailboxes, int32_t rank, int32_t world, int32_t cmax, int32_t C, int32_t timeout_ms,
                     int32_t* status) {
    if (mailboxes == nullptr || world < 1 || world > 16 || rank < 0 || rank >= world || cmax < C) return OCCD_EINVAL;
    for (int r = 0; r < world; ++r) {
        if (mailboxes[r] == nullptr) return OCCD_EINVAL;
        x.mbox[r] = static_cast<unsigned char*>(mailboxes[r]);
    }
    x.rank = rank; x.world = world;
    x.row_bytes = (long)cmax * kXchgRec;
    x.slots_off = kXchgHeader + (long)cmax * 8;
    x.slot_bytes = (long)world * x.row_bytes;
    x.timeout_ticks = timeout_ms > 0 ? (long long)timeout_ms * 100000LL : 0;
    x.status = status;
    return OCCD_OK;
}

int64_t occd_bn_xchg_mailbox_bytes(int32_t world, int32_t cmax) {
    if (world < 1 || world > 16 || cmax < 1) return OCCD_EINVAL;
    return kXchgHeader + (int64_t)cmax * 8 + 2 * (int64_t)world * cmax * kXchgRec;
}

int occd_bn_fwd_small_xchg(const occd_bn_args* a, double* packed, float eps, float momentum, const float* gamma,
                           const float* beta, float* running_mean, float* running_var, int64_t* num_batches_tracked,
                           float* mean, float* invstd, float* av, float* bv, void* const* mailboxes, int32_t rank, int32_t world,
                           int32_t cmax, int32_t timeout_ms, int32_t* status, void* stream) {
    BnP p{};
    const int rc = fill(a, p, false);
    if (rc != OCCD_OK) return rc;
    if (a->layout != 1 || !a->out || !packed || !mean || !invstd || !av || !bv ||
        (running_mean == nullptr) != (running_var == nullptr))
        return OCCD_EINVAL;
    BnXchg x{};
    if (fill_xchg(x, mailboxes, rank, world, cmax, a->C, timeout_ms, status) != OCCD_OK) return OCCD_EINVAL;
    const double elems = (double)a->batch * a->C * a->S;
    occd::ProfScope prof("bn_fwd_small_xchg", (hipStream_t)stream, 0.0, 4.0 * elems * (3 + (a->res != nullptr)));
    hipLaunchKernelGGL((bn_small_planes_kernel<0, true>), dim3(a->C), dim3(256), 0, (hipStream_t)stream, p, eps, momentum, gamma,
                       beta, running_mean, running_var, (long*)num_batches_tracked, packed, mean, invstd, av, bv, (float*)nullptr,
                       (float*)nullptr, x);
    return occd::check_launch();
}

int occd_bn_bwd_small_xchg(const occd_bn_args* a, const double* packed_fwd, float* gw, float* gb, void* const* mailboxes,
                           int32_t rank, int32_t world, int32_t cmax, int32_t timeout_ms, int32_t* status, void* stream) {
    BnP p{};
    const int rc = fill(a, p, false);
    if (rc != OCCD_OK) return rc;
    if (a->layout != 1 || !a->out || !a->gy || !a->a || !a->b || !a->mean || !a->invstd || !packed_fwd) return OCCD_EINVAL;
    BnXchg x{};
    if (fill_xchg(x, mailboxes, rank, world, cmax, a->C, timeout_ms, status) != OCCD_OK) return OCCD_EINVAL;
    const double elems = (double)a->batch * a->C * a->S;
    occd::ProfScope prof("bn_bwd_small_xchg", (hipStream_t)stream, 0.0,
                         4.0 * elems * (5 + 2 * (a->y != nullptr) + (a->out2 != nullptr)));
    hipLaunchKernelGGL((bn_small_planes_kernel<1, true>), dim3(a->C), dim3(256), 0, (hipStream_t)stream, p, 0.f, 0.f,
                       (const float*)nullptr, (const float*)nullptr, (float*)nullptr, (float*)nullptr, (long*)nullptr,
                       const_cast<double*>(packed_fwd), (float*)nullptr, (float*)nullptr, (float*)nullptr, (float*)nullptr, gw,
                       gb, x);
    return occd::check_launch();
}

int occd_bn_bwd_small(const occd_bn_args* a, float* gw, float* gb, void* stream) {
    BnP p{};
    const int rc = fill(a, p, false);
    if (rc != OCCD_OK) return rc;
    if (a->layout != 1 || !a->out || !a->gy || !a->a || !a->b || !a->mean || !a->invstd) return OCCD_EINVAL;
    const double elems = (double)a->batch * a->C * a->S;
    occd::ProfScope prof("bn_bwd_small", (hipStream_t)stream, 0.0,
                         4.0 * elems * (5 + 2 * (a->y != nullptr) + (a->out2 != nullptr)));
    hipLaunchKernelGGL((bn_small_planes_kernel<1>), dim3(a->C), dim3(256), 0, (hipStream_t)stream, p, 0.f, 0.f,
                       (const float*)nullptr, (const float*)nullptr, (float*)nullptr, (float*)nullptr, (long*)nullptr,
                       (double*)nullptr, (float*)nullptr, (float*)nullptr, (float*)nullptr, (float*)nullptr, gw, gb, BnXchg{});
    return occd::check_launch();
}

}  // extern "C"
