// K1 -- 2D->3D lift (Stereo-SFA gather + FLoSP-Depth frustum sample).
//
// Both kernels are HBM-bandwidth work: per voxel a handful of index loads, V x S
// gathers of one contiguous C-float pixel row from channels-last feature maps,
// a wavefront-level (DPP) reduction for the cosine similarity, and one
// contiguous C-float voxel row written channels-last for the 3-D stack.
//
//   lift_kernel<LPV>: LPV lanes cooperate on one voxel, each lane owns one
//   float4 of channels, so a pixel row is read as one coalesced LPV*16-byte
//   segment and the 64-wide wave covers 64/LPV voxels per instruction.
//
// Reference semantics: occdepth/models/SFA.py:12-106, occdepth/models/OccDepth.py:266-298,339;
// occdepth/models/flosp_depth/flosp_depth.py:561-602, f2v/frustum_grid_generator.py:70-152,
// f2v/utils/{transform_utils.py:5-26,depth_utils.py:24-26,grid_utils.py:4-19}, f2v/sampler.py:59-64.
#include "common.h"
#include <stdlib.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

// ---- sum over the LPV lanes of a voxel group; result in every lane ---------
template <int CTRL>
__device__ __forceinline__ float dpp_add(float v) {
    const int moved = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, false);
    return v + __int_as_float(moved);
}
template <int LPV>
__device__ __forceinline__ float group_sum(float v) {
    v = dpp_add<0xB1>(v);                       // quad_perm [1,0,3,2]   (xor 1)
    v = dpp_add<0x4E>(v);                       // quad_perm [2,3,0,1]   (xor 2)
    if (LPV >= 8) v = dpp_add<0x141>(v);        // row_half_mirror       (i -> 7-i)
    if (LPV >= 16) v = dpp_add<0x140>(v);       // row_mirror            (i -> 15-i)
    if (LPV >= 32) v += __shfl_xor(v, 16, 64);
    if (LPV >= 64) v += __shfl_xor(v, 32, 64);
    return v;
}

// SURVEY.md 8(f) row N2, one voxel: occdepth/data/utils/helpers.py:94-169 with fusion.py:203-217 (vox2world: float32
// origin, float64 arithmetic, float32 store), :518-522 (rigid transform in float64) and :336-337 (round(x * fx / z + cx),
// float32 intrinsics, numpy round-half-even).  Every product / sum is an explicitly rounded IEEE double operation (no FMA
// contraction) so the pixels are the ones numpy computes.  Returns the FOV flag.
__device__ __forceinline__ bool project_one(const double* __restrict__ E, double fx, double fy, double cx, double cy,
                                            double vox_size, const float* origin, int ix, int iy, int iz, int img_w,
                                            int img_h, long& px, long& py, double& camz) {
    const int idx[3] = {ix, iy, iz};
    double pt[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const double a = __dadd_rn((double)origin[j], __dmul_rn(vox_size, (double)(float)idx[j]));
        pt[j] = (double)(float)__dadd_rn(a, __dmul_rn(vox_size, 0.5));
    }
    double cam[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        double acc = __dmul_rn(E[r * 4 + 0], pt[0]);
        acc = __dadd_rn(acc, __dmul_rn(E[r * 4 + 1], pt[1]));
        acc = __dadd_rn(acc, __dmul_rn(E[r * 4 + 2], pt[2]));
        cam[r] = __dadd_rn(acc, E[r * 4 + 3]);
    }
    double xr = rint(__dadd_rn(__ddiv_rn(__dmul_rn(cam[0], fx), cam[2]), cx));
    double yr = rint(__dadd_rn(__ddiv_rn(__dmul_rn(cam[1], fy), cam[2]), cy));
    // non-finite projections (z == 0) are clamped like oracle/inputs.py; they are out of the FOV anyway
    xr = isnan(xr) ? -1e9 : fmin(fmax(xr, -1e9), 1e9);
    yr = isnan(yr) ? -1e9 : fmin(fmax(yr, -1e9), 1e9);
    px = (long)xr;
    py = (long)yr;
    camz = cam[2];
    return px >= 0 && px < img_w && py >= 0 && py < img_h && cam[2] > 0.0;
}

struct LiftP {
    occd_lift_args a;
    // power-of-two fast path of lift_p1_kernel (every shipped KITTI geometry): shifts instead of integer divisions --
    // the generic form spends ~25 VALU instructions per 32-bit division (2 per gathered row) and ~100 per 64-bit one (2
    // per stored row), in a kernel that is VALU / latency bound
    int sshift[OCCD_MAX_SCALES];
    int bc_shift, c_shift;
};

// Stereo-SFA fusion of the V per-view feature vectors of one voxel (SFA.py:46-89).
template <int LPV, int V>
__device__ __forceinline__ f32x4 sfa_fuse(const f32x4 (&f)[V], const float (&m)[V]) {
    if (V == 1) return f[0];
    f32x4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < V; ++i) {
#pragma unroll
        for (int j = i + 1; j < V; ++j) {
            // torch.cosine_similarity: normalise each vector by max(||.||, eps) first
            float ni = f[i].x * f[i].x + f[i].y * f[i].y + f[i].z * f[i].z + f[i].w * f[i].w;
            float nj = f[j].x * f[j].x + f[j].y * f[j].y + f[j].z * f[j].z + f[j].w * f[j].w;
            // x / max(||x||, eps) as x * rcp(max(sqrt(||x||^2), eps)) on the hardware sqrt / rcp instructions (1 ulp each):
            // the IEEE-exact sequences are ~10 VALU instructions apiece, four divisions per vector on top, and this
            // kernel is VALU / latency bound, not HBM bound (profiles/r02_lift_xcd_modes.txt)
            ni = fmaxf(__builtin_amdgcn_sqrtf(group_sum<LPV>(ni)), 1e-8f);
            nj = fmaxf(__builtin_amdgcn_sqrtf(group_sum<LPV>(nj)), 1e-8f);
            const float ri = __builtin_amdgcn_rcpf(ni), rj = __builtin_amdgcn_rcpf(nj);
            const f32x4 xi = f[i] * ri, xj = f[j] * rj;
            float d = xi.x * xj.x + xi.y * xj.y + xi.z * xj.z + xi.w * xj.w;
            d = group_sum<LPV>(d) * (m[i] * m[j]);
            const float wi = d + (m[i] > m[j] ? 1.f : 0.f);
            const float wj = d + (m[j] > m[i] ? 1.f : 0.f);
            o += wi * f[i] + wj * f[j];
        }
    }
    const float inv_den = 1.f / (float)(V * (V - 1));       // (exact for the stereo case: 0.5)
    return o * inv_den;
}

__device__ __forceinline__ void store_voxel_row(const occd_lift_args& a, int b, long n, int c, bool ch_ok, f32x4 v) {
    const long bc = a.dimB * (long)a.dimC;
    const long ia = n / bc;
    const long rem = n - ia * bc;
    const long ib = rem / a.dimC, ic = rem - ib * a.dimC;
    const long row = ia * a.row_a + ib * a.row_b + ic * a.row_c;
    float* o = a.out + ((size_t)b * a.out_rows + row) * a.out_cs + c;
    *(f32x4*)o = ch_ok ? v : f32x4{0.f, 0.f, 0.f, 0.f};
}

// Fast path: one pattern point (every shipped config), V views.  The kernel is latency-bound, so the
// projection indices are fetched first and then the gathers of ALL scales and views are put in flight
// together (V*S independent 16-byte loads per lane) before any arithmetic.
template <int LPV, int V, bool P2>
__global__ void __launch_bounds__(256) lift_p1_kernel(const LiftP pp) {
    const occd_lift_args& a = pp.a;
    const int tid = threadIdx.x;
    const int sub = tid % LPV;
    // Workgroup -> XCD placement (the dispatcher deals consecutive workgroups round-robin to the 8 XCDs, each with its
    // own L2).  A pixel row is re-read by the y / z neighbours of a voxel at the coarse scales and by the voxels further
    // along the same camera ray; which voxels share an L2 decides the HBM fetch (PMC numbers in DESIGN.md).
    uint32_t bid = blockIdx.x;
    if (a.xcd_mode == 1) {
        const uint32_t nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    } else if (a.xcd_mode == 2) {
        const uint32_t wps = (uint32_t)((long)a.dimB * a.dimC * LPV / 256);      // workgroups per a-slab (multiple of 8)
        const uint32_t per = wps >> 3, xcd = bid & 7, idx = bid >> 3;
        bid = (idx / per) * wps + xcd * per + idx % per;
    }
    const long n = ((long)bid * 256 + tid) / LPV;
    const int b = blockIdx.y;
    const bool vox_ok = n < a.N;
    const long nn = vox_ok ? n : (long)a.N - 1;
    const int c = sub * 4;
    const bool ch_ok = c < a.C;
    const int cc = ch_ok ? c : 0;
    const int64_t* pix = a.pix + ((size_t)b * V) * a.N * 2;
    const uint8_t* fov = a.fov + ((size_t)b * V) * a.N;

    // Every load below is UNCONDITIONAL (clamped address) and masked afterwards: a load guarded by a runtime condition
    // makes hipcc branch around it and wait for each one separately, which serialises the V*S gathers of a voxel.
    int px[V], py[V];
    uint32_t keep[V];
    float m[V];
#pragma unroll
    for (int v = 0; v < V; ++v) {
        const size_t pi = (size_t)v * a.N + nn;
        const bool in = fov[pi] != 0;
        const int64_t x64 = pix[pi * 2], y64 = pix[pi * 2 + 1];
        px[v] = in ? (int)x64 : 0;               // in-FOV pixels are non-negative image coordinates
        py[v] = in ? (int)y64 : 0;
        m[v] = in ? 1.f : 0.f;
        keep[v] = in && ch_ok ? 0xFFFFFFFFu : 0u;
    }
    f32x4 g[OCCD_MAX_SCALES][V];
#pragma unroll
    for (int s = 0; s < OCCD_MAX_SCALES; ++s)
#pragma unroll
        for (int v = 0; v < V; ++v) {
            const int ss = s < a.n_scales ? s : 0;                       // (absent scales re-read scale 0: discarded)
            const int dv = a.scale_div[ss], w = a.feat_w[ss], cs = a.feat_cs[ss];
            const int idx = P2 ? (py[v] >> pp.sshift[ss]) * w + (px[v] >> pp.sshift[ss]) : (py[v] / dv) * w + (px[v] / dv);
            const f32x4 t = *(const f32x4*)(a.feat[ss][v] + (size_t)b * a.feat_bstride[ss][v] + (size_t)idx * cs + cc);
            const uint32_t k = s < a.n_scales ? keep[v] : 0u;
            g[s][v] = f32x4{__uint_as_float(__float_as_uint(t.x) & k), __uint_as_float(__float_as_uint(t.y) & k),
                            __uint_as_float(__float_as_uint(t.z) & k), __uint_as_float(__float_as_uint(t.w) & k)};
        }
    f32x4 total = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < OCCD_MAX_SCALES; ++s)
        if (s < a.n_scales) {
            const f32x4 o = sfa_fuse<LPV, V>(g[s], m);   // count == 1: mean over pattern points is the point
            if (s == 0) total = o; else total += o;
        }
    if (a.depth_scale != nullptr) total = total * a.depth_scale[(size_t)b * a.N + nn] * a.scale_const;
    if (P2) {
        if (vox_ok && c < a.out_cs) {
            const uint32_t n32 = (uint32_t)n;
            const uint32_t ia = n32 >> pp.bc_shift, rem = n32 & ((1u << pp.bc_shift) - 1u);
            const uint32_t ib = rem >> pp.c_shift, ic = rem & ((1u << pp.c_shift) - 1u);
            const long row = (long)ia * a.row_a + (long)ib * a.row_b + (long)ic * a.row_c;
            float* o = a.out + ((size_t)b * a.out_rows + row) * a.out_cs + c;
            *(f32x4*)o = ch_ok ? total : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    } else if (vox_ok && c < a.out_cs) store_voxel_row(a, b, n, c, ch_ok, total);
}

// ---------------------------------------------------------------- backward of the single-pattern-point lift (N1)
// y = scale_const * depth_scale[n] * sum_s fuse_s(f_s,0 .. f_s,V-1): given gy (channels-last voxel rows) this kernel
// recomputes the gathers and the fusion weights, and scatters d loss / d f_s,v into per-scale, per-view gradient maps
// with hardware float atomics (pixel rows of the coarse scales receive many voxels), plus d loss / d depth_scale.
// For a pair (a, b) = (f_i, f_j) with masks m:   c = m_i m_j (a^.b^),  a^ = a / max(|a|, eps)
//     out = [(c + [m_i > m_j]) a + (c + [m_j > m_i]) b] / (V (V - 1))
//     d out . g / d a = (c + [m_i > m_j]) g + (g.a + g.b) m_i m_j (b^ - (a^.b^) a^) / |a|        (and symmetrically for b)
// Reference: autograd through occdepth/models/SFA.py:12-106 and OccDepth.py:266-298,339 (training_step).
struct LiftBwdP {
    occd_lift_args a;
    const float* gout;
    float* gfeat[OCCD_MAX_SCALES][OCCD_MAX_VIEWS];
    float* gdepth;
};

template <int LPV, int V>
__global__ void __launch_bounds__(256) lift_p1_bwd_kernel(const LiftBwdP pp) {
    const occd_lift_args& a = pp.a;
    const int tid = threadIdx.x;
    const int sub = tid % LPV;
    const long n = ((long)blockIdx.x * 256 + tid) / LPV;
    const int b = blockIdx.y;
    const bool vox_ok = n < a.N;
    const long nn = vox_ok ? n : (long)a.N - 1;
    const int c = sub * 4;
    const bool ch_ok = c < a.C;
    const int cc = ch_ok ? c : 0;
    const int64_t* pix = a.pix + ((size_t)b * V) * a.N * 2;
    const uint8_t* fov = a.fov + ((size_t)b * V) * a.N;

    int px[V], py[V];
    uint32_t keep[V];
    float m[V];
#pragma unroll
    for (int v = 0; v < V; ++v) {
        const size_t pi = (size_t)v * a.N + nn;
        const bool in = fov[pi] != 0 && vox_ok;
        const int64_t x64 = pix[pi * 2], y64 = pix[pi * 2 + 1];
        px[v] = in ? (int)x64 : 0;
        py[v] = in ? (int)y64 : 0;
        m[v] = in ? 1.f : 0.f;
        keep[v] = in && ch_ok ? 0xFFFFFFFFu : 0u;
    }
    // upstream gradient row of this voxel (same row mapping as the forward store)
    f32x4 g;
    {
        const long bc = a.dimB * (long)a.dimC;
        const long ia = nn / bc, rem = nn - ia * bc;
        const long ib = rem / a.dimC, ic = rem - ib * a.dimC;
        const long row = ia * a.row_a + ib * a.row_b + ic * a.row_c;
        const f32x4 t = *(const f32x4*)(pp.gout + ((size_t)b * a.out_rows + row) * a.out_cs + cc);
        const uint32_t k = vox_ok && ch_ok ? 0xFFFFFFFFu : 0u;
        g = f32x4{__uint_as_float(__float_as_uint(t.x) & k), __uint_as_float(__float_as_uint(t.y) & k),
                  __uint_as_float(__float_as_uint(t.z) & k), __uint_as_float(__float_as_uint(t.w) & k)};
    }
    // (the forward applies depth_scale * scale_const only when a depth volume is given)
    const float dsc = a.depth_scale != nullptr ? a.depth_scale[(size_t)b * a.N + nn] * a.scale_const : 1.f;
    const f32x4 gt = g * dsc;                                    // d loss / d (sum over scales)
    const float inv_den = V > 1 ? 1.f / (float)(V * (V - 1)) : 1.f;
    float g_dot_total = 0.f;                                     // g . sum_s out_s  (for d loss / d depth_scale)

#pragma unroll
    for (int s = 0; s < OCCD_MAX_SCALES; ++s) {
        if (s >= a.n_scales) break;                              // (uniform)
        const int dv = a.scale_div[s], w = a.feat_w[s], cs = a.feat_cs[s];
        f32x4 f[V];
        size_t off[V];
#pragma unroll
        for (int v = 0; v < V; ++v) {
            const int idx = (py[v] / dv) * w + (px[v] / dv);
            off[v] = (size_t)b * a.feat_bstride[s][v] + (size_t)idx * cs + cc;
            const f32x4 t = *(const f32x4*)(a.feat[s][v] + off[v]);
            const uint32_t k = keep[v];
            f[v] = f32x4{__uint_as_float(__float_as_uint(t.x) & k), __uint_as_float(__float_as_uint(t.y) & k),
                         __uint_as_float(__float_as_uint(t.z) & k), __uint_as_float(__float_as_uint(t.w) & k)};
        }
        f32x4 gf[V];
#pragma unroll
        for (int v = 0; v < V; ++v) gf[v] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (V == 1) {
            gf[0] = gt;
            g_dot_total += g.x * f[0].x + g.y * f[0].y + g.z * f[0].z + g.w * f[0].w;
        } else {
#pragma unroll
            for (int i = 0; i < V; ++i)
#pragma unroll
                for (int j = i + 1; j < V; ++j) {
                    auto dot = [](const f32x4& p, const f32x4& q) { return p.x * q.x + p.y * q.y + p.z * q.z + p.w * q.w; };
                    const float aa = group_sum<LPV>(dot(f[i], f[i])), bb = group_sum<LPV>(dot(f[j], f[j]));
                    const float na = fmaxf(sqrtf(aa), 1e-8f), nb = fmaxf(sqrtf(bb), 1e-8f);
                    const float mm = m[i] * m[j];
                    const float cosab = group_sum<LPV>(dot(f[i], f[j])) / (na * nb);
                    const float cw = cosab * mm;
                    const float wi = cw + (m[i] > m[j] ? 1.f : 0.f), wj = cw + (m[j] > m[i] ? 1.f : 0.f);
                    const float ga = group_sum<LPV>(dot(gt, f[i])), gb = group_sum<LPV>(dot(gt, f[j]));
                    const float S = (ga + gb) * mm * inv_den;
                    // d c / d a = (b^ - cos a^) / |a|   (|a| > eps; below it a^ = a / eps and the projection term vanishes)
                    const f32x4 ah = f[i] * (1.f / na), bh = f[j] * (1.f / nb);
                    const float pa = sqrtf(aa) > 1e-8f ? cosab : 0.f, pb = sqrtf(bb) > 1e-8f ? cosab : 0.f;
                    gf[i] += gt * (wi * inv_den) + (bh - ah * pa) * (S / na);
                    gf[j] += gt * (wj * inv_den) + (ah - bh * pb) * (S / nb);
                    const f32x4 o = (f[i] * wi + f[j] * wj) * inv_den;
                    g_dot_total += dot(g, o);
                }
        }
#pragma unroll
        for (int v = 0; v < V; ++v)
            if (keep[v] != 0u) {                                  // in the field of view, real channel
                float* dst = pp.gfeat[s][v] + off[v];
                unsafeAtomicAdd(dst + 0, gf[v].x);
                unsafeAtomicAdd(dst + 1, gf[v].y);
                unsafeAtomicAdd(dst + 2, gf[v].z);
                unsafeAtomicAdd(dst + 3, gf[v].w);
            }
    }
    if (pp.gdepth != nullptr) {
        const float t = group_sum<LPV>(g_dot_total) * a.scale_const;
        if (vox_ok && sub == 0) pp.gdepth[(size_t)b * a.N + n] = t;
    }
}

// General path: any pattern size P (DSO patterns up to 25 points), V views.
template <int LPV, int V>
__global__ void __launch_bounds__(256) lift_any_kernel(const LiftP pp) {
    const occd_lift_args& a = pp.a;
    const int tid = threadIdx.x;
    const int sub = tid % LPV;
    const long n = ((long)blockIdx.x * 256 + tid) / LPV;
    const int b = blockIdx.y;
    const bool vox_ok = n < a.N;
    const long nn = vox_ok ? n : (long)a.N - 1;
    const int c = sub * 4;
    const bool ch_ok = c < a.C;
    const int P = a.P;
    const int64_t* pix = a.pix + (((size_t)b * V) * a.N) * P * 2;
    const uint8_t* fov = a.fov + (((size_t)b * V) * a.N) * P;

    f32x4 total = {0.f, 0.f, 0.f, 0.f};
    for (int s = 0; s < a.n_scales; ++s) {
        const int div = a.scale_div[s], w = a.feat_w[s], cs = a.feat_cs[s];
        f32x4 f[V];
        float m[V];
#pragma unroll
        for (int v = 0; v < V; ++v) {
            f[v] = f32x4{0.f, 0.f, 0.f, 0.f};
            const float* fm = a.feat[s][v] + (size_t)b * a.feat_bstride[s][v] + c;
            int cnt = 0;
            for (int q = 0; q < P; ++q) {          // pattern points accumulate in index order (SFA.py:28-30)
                const size_t pi = ((size_t)v * a.N + nn) * P + q;
                if (fov[pi]) {
                    const long idx = (pix[pi * 2 + 1] / div) * w + (pix[pi * 2] / div);
                    ++cnt;
                    if (ch_ok) f[v] += *(const f32x4*)(fm + (size_t)idx * cs);
                }
            }
            m[v] = cnt > 0 ? 1.f : 0.f;
            if (cnt > 0) {
                const float fc = (float)cnt;
                f[v].x /= fc; f[v].y /= fc; f[v].z /= fc; f[v].w /= fc;
            }
        }
        const f32x4 o = sfa_fuse<LPV, V>(f, m);
        if (s == 0) total = o; else total += o;
    }
    if (a.depth_scale != nullptr) total = total * a.depth_scale[(size_t)b * a.N + nn] * a.scale_const;
    if (vox_ok && c < a.out_cs) store_voxel_row(a, b, n, c, ch_ok, total);
}

template <int LPV>
void launch_lift(const LiftP& p, dim3 grid, hipStream_t st) {
    const int V = p.a.n_views;
    const bool p1 = p.a.P == 1;
    const bool p2 = p.bc_shift >= 0;
#define OCCD_LIFT(VV)                                                                             \
    if (V == VV) {                                                                                \
        if (p1 && p2) hipLaunchKernelGGL((lift_p1_kernel<LPV, VV, true>), grid, dim3(256), 0, st, p);   \
        else if (p1) hipLaunchKernelGGL((lift_p1_kernel<LPV, VV, false>), grid, dim3(256), 0, st, p);   \
        else hipLaunchKernelGGL((lift_any_kernel<LPV, VV>), grid, dim3(256), 0, st, p);           \
    }
    OCCD_LIFT(1) OCCD_LIFT(2) OCCD_LIFT(3) OCCD_LIFT(4)
#undef OCCD_LIFT
}

// ------------------------------------------------------------ frustum sample
struct FlospP {
    occd_flosp_args a;
    float bin_size;
};

__device__ __forceinline__ float from_homog_scale(float w) {
    // kornia.convert_points_from_homogeneous (0.5.0): eps = 1e-8
    return fabsf(w) > 1e-8f ? 1.f / (w + 1e-8f) : 1.f;
}

// Continuous sample position (ix, iy, iz) of voxel n in camera `cam`'s (D, h, w) depth frustum: voxel centre -> camera ->
// image -> LID bin -> ida -> normalised grid -> F.grid_sample(align_corners=False) un-normalisation.  Shared by the forward
// sample (standalone kernel and fused lift) and by its transpose (flosp_sample_bwd_kernel), so that the two apply the very
// same weights.
__device__ __forceinline__ void frustum_coords(const FlospP& pp, int b, long n, int cam, float& ix, float& iy, float& iz) {
    const occd_flosp_args& a = pp.a;
    const long nvox = (long)a.A * a.Bdim * a.C;
    float nx, ny, nz;
    if (a.grids != nullptr) {
        const float* g = a.grids + ((((size_t)cam * a.batch + b) * nvox) + n) * 3;
        nx = g[0]; ny = g[1]; nz = g[2];
    } else {
        const long bc = (long)a.Bdim * a.C;
        const int ia = (int)(n / bc);
        const long rem = n - (long)ia * bc;
        const int ib = (int)(rem / a.C), ic = (int)(rem - (long)ib * a.C);
        const float gx = (float)ia + 0.5f, gy = (float)ib + 0.5f, gz = (float)ic + 0.5f;
        const float* T = a.trans + ((size_t)b * a.n_cams + cam) * 16;
        const float* K = a.proj + ((size_t)b * a.n_cams + cam) * 12;
        const float* I = a.ida + ((size_t)b * a.n_cams + cam) * 16;
        // voxel centre -> camera frame
        float cx = gx * T[0] + gy * T[1] + gz * T[2] + T[3];
        float cy = gx * T[4] + gy * T[5] + gz * T[6] + T[7];
        float cz = gx * T[8] + gy * T[9] + gz * T[10] + T[11];
        const float cw = gx * T[12] + gy * T[13] + gz * T[14] + T[15];
        const float sc = from_homog_scale(cw);
        cx *= sc; cy *= sc; cz *= sc;
        // camera -> image plane, depth
        const float u0 = K[0] * cx + K[1] * cy + K[2] * cz + K[3];
        const float v0 = K[4] * cx + K[5] * cy + K[6] * cz + K[7];
        const float w0 = K[8] * cx + K[9] * cy + K[10] * cz + K[11];
        const float sp = from_homog_scale(w0);
        const float u = u0 * sp, v = v0 * sp;
        const float dep = w0 - K[11];
        // LID depth bin
        const float bin = -0.5f + 0.5f * sqrtf(1.f + 8.f * (dep - a.depth_min) / pp.bin_size);
        // image-data-augmentation matrix
        float fx = u * I[0] + v * I[1] + bin * I[2] + I[3];
        float fy = u * I[4] + v * I[5] + bin * I[6] + I[7];
        float fz = u * I[8] + v * I[9] + bin * I[10] + I[11];
        const float fw = u * I[12] + v * I[13] + bin * I[14] + I[15];
        const float si = from_homog_scale(fw);
        fx *= si; fy *= si; fz *= si;
        // normalise with the FULL image size (reference quirk) and D
        nx = fx / (a.img_w - 1.f) * 2.f + -1.f;
        ny = fy / (a.img_h - 1.f) * 2.f + -1.f;
        nz = fz / ((float)a.D - 1.f) * 2.f + -1.f;
        if (!isfinite(nx)) nx = -2.f;
        if (!isfinite(ny)) ny = -2.f;
        if (!isfinite(nz)) nz = -2.f;
    }
    // F.grid_sample 5-D, bilinear, zeros padding, align_corners=False
    ix = ((nx + 1.f) * (float)a.w - 1.f) / 2.f;
    iy = ((ny + 1.f) * (float)a.h - 1.f) / 2.f;
    iz = ((nz + 1.f) * (float)a.D - 1.f) / 2.f;
}

// The 8 trilinear corners of a sample position: visit(corner element offset in the (D, h, w) volume, weight) for the
// corners inside the volume (zeros padding).
template <class F>
__device__ __forceinline__ void frustum_corners(const occd_flosp_args& a, float ix, float iy, float iz, F&& visit) {
    const float x0f = floorf(ix), y0f = floorf(iy), z0f = floorf(iz);
    const float wx1 = ix - x0f, wy1 = iy - y0f, wz1 = iz - z0f;
    const float wx0 = (x0f + 1.f) - ix, wy0 = (y0f + 1.f) - iy, wz0 = (z0f + 1.f) - iz;
    // float -> int is saturating on the device; far-away coordinates stay out of bounds
    const int x0 = (int)x0f, y0 = (int)y0f, z0 = (int)z0f;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int dx = k & 1, dy = (k >> 1) & 1, dz = k >> 2;
        const int x = x0 + dx, y = y0 + dy, z = z0 + dz;
        const float wgt = (dx ? wx1 : wx0) * (dy ? wy1 : wy0) * (dz ? wz1 : wz0);
        if ((unsigned)x < (unsigned)a.w && (unsigned)y < (unsigned)a.h && (unsigned)z < (unsigned)a.D)
            visit(((size_t)z * a.h + y) * a.w + x, wgt);
    }
}

// One voxel of the frustum sample (shared by the standalone kernel and the fused lift).
__device__ __forceinline__ float frustum_sample_one(const FlospP& pp, int b, long n) {
    const occd_flosp_args& a = pp.a;
    float feat_sum = 0.f, mask_sum = 0.f;
    for (int cam = 0; cam < a.n_cams; ++cam) {
        float ix, iy, iz;
        frustum_coords(pp, b, n, cam, ix, iy, iz);
        const float* vol = a.depth + ((size_t)b * a.n_cams + cam) * a.D * a.h * a.w;
        float acc = 0.f, msk = 0.f;
        frustum_corners(a, ix, iy, iz, [&](size_t off, float wgt) {
            acc += vol[off] * wgt;
            msk += wgt;
        });
        feat_sum += acc;
        mask_sum += msk;
    }
    float r = feat_sum;
    if (a.n_cams > 1 && a.mean_mode && mask_sum > 0.f) r = feat_sum / mask_sum;
    return r;
}

// ---- transpose of the frustum sample (training: d loss / d depth volume; occdepth/models/f2v/sampler.py:59-64 under
// autograd = grid_sampler_3d_backward's float atomics).  Here the scatter is DETERMINISTIC: contributions are accumulated
// as 64-bit fixed point (integer addition is order-free) with a power-of-two scale derived from max |gout| -- itself an
// order-free reduction (atomicMax on the float bits) -- so that nothing can overflow: |sum into one cell| <= B nvox max|g|.
//   scale = 2^(61 - ceil(log2(B nvox)) - (exponent(max|g|) + 1))   (resolution ~2^-40 of max|g| at config 2: float32 has 2^-24)
struct FlospBwdP {
    FlospP f;                 // f.a.depth is not read (the sample is linear in the volume); f.a.out unused
    const float* gout;        // (B, nvox) d loss / d sampled volume
    long long* acc;           // (B, n_cams, D, h, w) fixed-point accumulators
    unsigned* gmax_bits;      // one word: bits of max |gout|
    float* gdepth;            // (B, n_cams, D, h, w)
    long total_out, total_vol;
    int log2_count;           // ceil(log2(B nvox))
};

__device__ __forceinline__ int flosp_scale_exp(unsigned max_bits, int log2_count) {
    const int e = (int)((max_bits >> 23) & 0xff) - 127;              // max|g| < 2^(e + 1)  (denormals: e = -127, still an upper bound)
    return 61 - log2_count - (e + 1);
}

__global__ void flosp_bwd_zero_kernel(unsigned* p) {       // (a kernel, not hipMemsetAsync: see loss.hip zero_u64_kernel)
    if (threadIdx.x < 2) p[threadIdx.x] = 0u;
}

__global__ void __launch_bounds__(256) flosp_bwd_prepare_kernel(const FlospBwdP q) {
    // zero the accumulators (grid-stride) and reduce max |gout| (non-negative floats order like their bit patterns)
    const long i0 = (long)blockIdx.x * 256 + threadIdx.x, step = (long)gridDim.x * 256;
    for (long i = i0; i < q.total_vol; i += step) q.acc[i] = 0;
    unsigned m = 0;
    for (long i = i0; i < q.total_out; i += step) {
        const float g = q.gout[i];
        const unsigned bits = __float_as_uint(g) & 0x7fffffffu;
        if (bits <= 0x7f800000u) m = max(m, bits);                    // (NaN gradients are not a maximum; they propagate below)
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_down((int)m, o));
    if ((threadIdx.x & 63) == 0 && m) atomicMax(q.gmax_bits, m);
}

__global__ void __launch_bounds__(256) flosp_sample_bwd_kernel(const FlospBwdP q) {
    const occd_flosp_args& a = q.f.a;
    const long nvox = (long)a.A * a.Bdim * a.C;
    const long n = (long)blockIdx.x * 256 + threadIdx.x;
    const int b = blockIdx.y;
    if (n >= nvox) return;
    const float g = q.gout[(size_t)b * nvox + n];
    if (g == 0.f) return;
    float ix[OCCD_MAX_VIEWS], iy[OCCD_MAX_VIEWS], iz[OCCD_MAX_VIEWS];
    float mask_sum = 0.f;
    for (int cam = 0; cam < a.n_cams; ++cam) {
        frustum_coords(q.f, b, n, cam, ix[cam], iy[cam], iz[cam]);
        frustum_corners(a, ix[cam], iy[cam], iz[cam], [&](size_t, float wgt) { mask_sum += wgt; });
    }
    float gv = g;
    if (a.n_cams > 1 && a.mean_mode && mask_sum > 0.f) gv = g / mask_sum;
    const double scale = ldexp(1.0, flosp_scale_exp(*q.gmax_bits, q.log2_count));
    for (int cam = 0; cam < a.n_cams; ++cam) {
        long long* dst = q.acc + ((size_t)b * a.n_cams + cam) * a.D * a.h * a.w;
        frustum_corners(a, ix[cam], iy[cam], iz[cam], [&](size_t off, float wgt) {
            const long long v = __double2ll_rn((double)(gv * wgt) * scale);
            if (v) atomicAdd((unsigned long long*)(dst + off), (unsigned long long)v);
        });
    }
}

__global__ void __launch_bounds__(256) flosp_bwd_finish_kernel(const FlospBwdP q) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= q.total_vol) return;
    const double inv = ldexp(1.0, -flosp_scale_exp(*q.gmax_bits, q.log2_count));
    q.gdepth[i] = (float)((double)q.acc[i] * inv);
}

__global__ void __launch_bounds__(256) flosp_sample_kernel(const FlospP pp) {
    const occd_flosp_args& a = pp.a;
    const long nvox = (long)a.A * a.Bdim * a.C;
    const long n = (long)blockIdx.x * 256 + threadIdx.x;
    const int b = blockIdx.y;
    if (n >= nvox) return;
    a.out[(size_t)b * nvox + n] = frustum_sample_one(pp, b, n);
}

// ------------------------------------------------------------ fused lift: projection + frustum sample + SFA gather
// VERDICT r2 item 7 / SURVEY 8(f) N2: the eval lift without its tables.  A workgroup owns T = 4 * (256 / LPV)
// consecutive voxels.  Phase A: T * V threads project one (voxel, view) each in explicitly rounded float64 (project_one:
// the integers the dataloader's numba vox2pix produces) and T more threads sample the depth frustum of one voxel each
// (frustum_sample_one); pixels and depth scales are staged in LDS -- no (B, V, N, 1, 2) int64 table, no fov mask, no
// (B, N) depth_scale vector in HBM.  Phase B: the lane groups walk their 4 voxels two at a time, 2 * V * S independent
// 16-byte gathers in flight per lane, fuse (sfa_fuse) and write channels-last voxel rows.
struct LiftProjP {
    LiftP l;                 // l.a.pix / fov / depth_scale are unused (null)
    const double* cam_E;     // (B, V, 16) device doubles, row major (lidar -> camera)
    const double* cam_k;     // (B, V, 9) device doubles, row major; fx, fy, cx, cy are rounded to float32 here
    double vox_size;
    float origin[3];
    int img_w, img_h;
    FlospP fr;
    int has_frustum;
};

template <int LPV, int V, int U>   // U = voxels in flight per lane group (1, 2 or 4)
__global__ void __launch_bounds__(256) lift_proj_kernel(const LiftProjP pp) {
    constexpr int G = 256 / LPV, T = 4 * G;
    const occd_lift_args& a = pp.l.a;
    __shared__ int s_pix[T * V];        // py << 16 | px, or -1 outside the field of view
    __shared__ float s_ds[T];
    const int tid = threadIdx.x;
    uint32_t bid = blockIdx.x;
    if (a.xcd_mode == 1) {
        const uint32_t nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    } else if (a.xcd_mode == 2) {
        const uint32_t wps = (uint32_t)(((long)a.dimB * a.dimC) / T);       // workgroups per a-slab (multiple of 8)
        const uint32_t per = wps >> 3, xcd = bid & 7, idx = bid >> 3;
        bid = (idx / per) * wps + xcd * per + idx % per;
    }
    const long n0 = (long)bid * T;
    const int b = blockIdx.y;

    // ---- phase A
    for (int i = tid; i < T * V + (pp.has_frustum ? T : 0); i += 256) {
        if (i < T * V) {
            const int t = i / V, v = i - t * V;
            const long n = n0 + t;
            int code = -1;
            if (n < a.N) {
                const uint32_t n32 = (uint32_t)n;
                const int ix = (int)(n32 >> pp.l.bc_shift);
                const uint32_t rem = n32 & ((1u << pp.l.bc_shift) - 1u);
                const int iy = (int)(rem >> pp.l.c_shift), iz = (int)(rem & ((1u << pp.l.c_shift) - 1u));
                const double* E = pp.cam_E + ((size_t)b * V + v) * 16;
                const double* K = pp.cam_k + ((size_t)b * V + v) * 9;
                long px, py;
                double camz;
                if (project_one(E, (double)(float)K[0], (double)(float)K[4], (double)(float)K[2], (double)(float)K[5],
                                pp.vox_size, pp.origin, ix, iy, iz, pp.img_w, pp.img_h, px, py, camz))
                    code = (int)((py << 16) | px);
            }
            s_pix[i] = code;
        } else {
            const int t = i - T * V;
            const long n = n0 + t;
            s_ds[t] = n < a.N ? frustum_sample_one(pp.fr, b, n) : 0.f;
        }
    }
    __syncthreads();

    // ---- phase B
    const int sub = tid % LPV, grp = tid / LPV;
    const int c = sub * 4;
    const bool ch_ok = c < a.C;
    const int cc = ch_ok ? c : 0;
#pragma unroll
    for (int q = 0; q < 4 / U; ++q) {
        f32x4 g[U][OCCD_MAX_SCALES][V];
        float m[U][V];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int t = (U * q + u) * G + grp;
#pragma unroll
            for (int v = 0; v < V; ++v) {
                const int code = s_pix[t * V + v];
                const bool in = code >= 0;
                const int px = in ? code & 0xFFFF : 0, py = in ? code >> 16 : 0;
                m[u][v] = in ? 1.f : 0.f;
                const uint32_t keep = in && ch_ok ? 0xFFFFFFFFu : 0u;
#pragma unroll
                for (int s = 0; s < OCCD_MAX_SCALES; ++s) {
                    const int ss = s < a.n_scales ? s : 0;                   // (absent scales re-read scale 0: discarded)
                    const int w = a.feat_w[ss], cs = a.feat_cs[ss];
                    const int idx = (py >> pp.l.sshift[ss]) * w + (px >> pp.l.sshift[ss]);
                    const f32x4 tv = *(const f32x4*)(a.feat[ss][v] + (size_t)b * a.feat_bstride[ss][v] + (size_t)idx * cs + cc);
                    const uint32_t k = s < a.n_scales ? keep : 0u;
                    g[u][s][v] = f32x4{__uint_as_float(__float_as_uint(tv.x) & k), __uint_as_float(__float_as_uint(tv.y) & k),
                                       __uint_as_float(__float_as_uint(tv.z) & k), __uint_as_float(__float_as_uint(tv.w) & k)};
                }
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int t = (U * q + u) * G + grp;
            const long n = n0 + t;
            f32x4 total = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < OCCD_MAX_SCALES; ++s)
                if (s < a.n_scales) {
                    const f32x4 o = sfa_fuse<LPV, V>(g[u][s], m[u]);
                    if (s == 0) total = o; else total += o;
                }
            if (pp.has_frustum) total = total * s_ds[t] * a.scale_const;
            if (n < a.N && c < a.out_cs) {
                const uint32_t n32 = (uint32_t)n;
                const uint32_t ia = n32 >> pp.l.bc_shift, rem = n32 & ((1u << pp.l.bc_shift) - 1u);
                const uint32_t ib = rem >> pp.l.c_shift, ic = rem & ((1u << pp.l.c_shift) - 1u);
                const long row = (long)ia * a.row_a + (long)ib * a.row_b + (long)ic * a.row_c;
                float* o = a.out + ((size_t)b * a.out_rows + row) * a.out_cs + c;
                *(f32x4*)o = ch_ok ? total : f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
    }
}

// ------------------------------------------------------------ layout helpers
// (B, C, S) -> (B, S, cs): 64 positions x 32 channels per workgroup through LDS
__global__ void __launch_bounds__(256) nchw_to_nhwc_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                           int C, long S, int cs) {
    __shared__ float tile[32][65];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;  // 64 x 4
    const long s0 = (long)blockIdx.x * 64;
    const int c0 = blockIdx.y * 32;
    const int b = blockIdx.z;
    const float* ib = in + (size_t)b * C * S;
    float* ob = out + (size_t)b * S * cs;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int c = c0 + ty * 8 + k;
        const long s = s0 + tx;
        tile[ty * 8 + k][tx] = (c < C && s < S) ? ib[(size_t)c * S + s] : 0.f;
    }
    __syncthreads();
    const int cl = threadIdx.x & 31, sl = threadIdx.x >> 5;  // 32 x 8
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const long s = s0 + sl + k * 8;
        const int c = c0 + cl;
        if (s < S && c < cs) ob[(size_t)s * cs + c] = tile[cl][sl + k * 8];
    }
}

// (B, S, cs)[coff : coff + C] -> (B, C, S)
__global__ void __launch_bounds__(256) nhwc_to_nchw_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                           int C, long S, int cs, int coff) {
    __shared__ float tile[64][33];
    const long s0 = (long)blockIdx.x * 64;
    const int c0 = blockIdx.y * 32;
    const int b = blockIdx.z;
    const float* ib = in + (size_t)b * S * cs + coff;
    float* ob = out + (size_t)b * C * S;
    const int cl = threadIdx.x & 31, sl = threadIdx.x >> 5;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const long s = s0 + sl + k * 8;
        const int c = c0 + cl;
        tile[sl + k * 8][cl] = (s < S && c < C) ? ib[(size_t)s * cs + c] : 0.f;
    }
    __syncthreads();
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int c = c0 + ty * 8 + k;
        const long s = s0 + tx;
        if (c < C && s < S) ob[(size_t)c * S + s] = tile[tx][ty * 8 + k];
    }
}

__global__ void __launch_bounds__(256) softmax_channels_kernel(const float* __restrict__ src,
                                                               float* __restrict__ dst, long rows, int src_cs,
                                                               int src_coff, int dst_cs, int dst_coff, int n, int dst_pad) {
    const long r = (long)blockIdx.x * 256 + threadIdx.x;
    if (r >= rows) return;
    const float* s = src + (size_t)r * src_cs + src_coff;
    float* d = dst + (size_t)r * dst_cs + dst_coff;
    float mx = s[0];
    for (int i = 1; i < n; ++i) mx = fmaxf(mx, s[i]);
    float sum = 0.f;
    for (int i = 0; i < n; ++i) sum += expf(s[i] - mx);
    for (int i = 0; i < n; ++i) d[i] = expf(s[i] - mx) / sum;
    for (int i = 0; i < dst_pad; ++i) d[n + i] = 0.f;
}

}  // namespace

extern "C" int occd_lift_fwd(const occd_lift_args* a, void* stream) {
    if (!a || !a->pix || !a->fov || !a->out) return OCCD_EINVAL;
    if (a->n_scales < 1 || a->n_scales > OCCD_MAX_SCALES || a->n_views < 1 || a->n_views > OCCD_MAX_VIEWS)
        return OCCD_EINVAL;
    if (a->C <= 0 || (a->C & 3) || a->C > 256 || (a->out_cs & 3) || a->out_cs < a->C || a->out_cs > 256)
        return OCCD_EINVAL;
    if (a->N <= 0 || a->P <= 0 || a->batch <= 0) return OCCD_EINVAL;
    if ((long)a->dimA * a->dimB * a->dimC != (long)a->N) return OCCD_EINVAL;
    for (int s = 0; s < a->n_scales; ++s) {
        if (a->scale_div[s] <= 0 || (a->feat_cs[s] & 3) || a->feat_cs[s] < a->C) return OCCD_EINVAL;
        for (int v = 0; v < a->n_views; ++v)
            if (!a->feat[s][v] || (reinterpret_cast<uintptr_t>(a->feat[s][v]) & 15)) return OCCD_EINVAL;
    }
    LiftP p;
    p.a = *a;
    const int need = a->out_cs / 4;  // lanes that must exist per voxel
    const int lpv = need <= 8 ? 8 : need <= 16 ? 16 : need <= 32 ? 32 : 64;
    {   // shift form of the index arithmetic when every divisor is a power of two (pixel coordinates are >= 0)
        auto lg2 = [](long v) { int s = 0; while ((1L << s) < v) ++s; return (1L << s) == v ? s : -1; };
        bool all = true;
        for (int s = 0; s < OCCD_MAX_SCALES; ++s) {
            p.sshift[s] = s < a->n_scales ? lg2(a->scale_div[s]) : 0;
            all = all && p.sshift[s] >= 0;
        }
        p.bc_shift = lg2((long)a->dimB * a->dimC);
        p.c_shift = lg2(a->dimC);
        if (!all || p.bc_shift < 0 || p.c_shift < 0) p.bc_shift = p.c_shift = -1;
    }
    for (int s = 0; s < a->n_scales; ++s)
        for (int v = 0; v < a->n_views; ++v)
            if (p.a.feat_bstride[s][v] == 0) p.a.feat_bstride[s][v] = (int64_t)a->feat_h[s] * a->feat_w[s] * a->feat_cs[s];
    if (p.a.xcd_mode < 0 || p.a.xcd_mode > 2) return OCCD_EINVAL;
    if (p.a.xcd_mode == 2 && ((long)a->dimB * a->dimC * lpv) % 2048 != 0) p.a.xcd_mode = 0;
    const long threads = (long)a->N * lpv;
    if (p.a.xcd_mode == 2 && threads % 256 != 0) p.a.xcd_mode = 0;
    const dim3 grid((unsigned)((threads + 255) / 256), (unsigned)a->batch);
    // algorithmic bytes: output rows + one gathered pixel row per view/scale + indices
    const double bytes = (double)a->batch * a->N *
                         (4.0 * a->C * (1 + (double)a->n_views * a->n_scales) + (double)a->n_views * a->P * 17 +
                          (a->depth_scale ? 4 : 0));
    occd::ProfScope prof("sfa_lift", (hipStream_t)stream, 0.0, bytes);
    hipStream_t st = (hipStream_t)stream;
    switch (lpv) {
        case 8: launch_lift<8>(p, grid, st); break;
        case 16: launch_lift<16>(p, grid, st); break;
        case 32: launch_lift<32>(p, grid, st); break;
        default: launch_lift<64>(p, grid, st); break;
    }
    return occd::check_launch();
}

extern "C" int occd_lift_bwd(const occd_lift_bwd_args* q, void* stream) {
    if (!q || !q->gout) return OCCD_EINVAL;
    const occd_lift_args* a = &q->fwd;
    if (!a->pix || !a->fov) return OCCD_EINVAL;
    if (a->n_scales < 1 || a->n_scales > OCCD_MAX_SCALES || a->n_views < 1 || a->n_views > OCCD_MAX_VIEWS) return OCCD_EINVAL;
    if (a->P != 1) return OCCD_EINVAL;                        // pattern_id 0 only (every shipped config)
    if (a->C <= 0 || (a->C & 3) || a->C > 256 || (a->out_cs & 3) || a->out_cs < a->C || a->out_cs > 256) return OCCD_EINVAL;
    if (a->N <= 0 || a->batch <= 0 || (long)a->dimA * a->dimB * a->dimC != (long)a->N) return OCCD_EINVAL;
    LiftBwdP p;
    p.a = *a;
    p.gout = q->gout;
    p.gdepth = q->gdepth;
    for (int s = 0; s < a->n_scales; ++s) {
        if (a->scale_div[s] <= 0 || (a->feat_cs[s] & 3) || a->feat_cs[s] < a->C) return OCCD_EINVAL;
        for (int v = 0; v < a->n_views; ++v) {
            if (!a->feat[s][v] || !q->gfeat[s][v]) return OCCD_EINVAL;
            p.gfeat[s][v] = q->gfeat[s][v];
            if (p.a.feat_bstride[s][v] == 0) p.a.feat_bstride[s][v] = (int64_t)a->feat_h[s] * a->feat_w[s] * a->feat_cs[s];
        }
    }
    const int need = a->out_cs / 4;
    const int lpv = need <= 8 ? 8 : need <= 16 ? 16 : need <= 32 ? 32 : 64;
    const long threads = (long)a->N * lpv;
    const dim3 grid((unsigned)((threads + 255) / 256), (unsigned)a->batch);
    hipStream_t st = (hipStream_t)stream;
    occd::ProfScope prof("sfa_lift_bwd", st, 0.0,
                         (double)a->batch * a->N * 4.0 * a->C * (1 + 3.0 * a->n_views * a->n_scales));
#define OCCD_LB(L, VV) hipLaunchKernelGGL((lift_p1_bwd_kernel<L, VV>), grid, dim3(256), 0, st, p)
#define OCCD_LBV(L) \
    switch (a->n_views) { case 1: OCCD_LB(L, 1); break; case 2: OCCD_LB(L, 2); break; case 3: OCCD_LB(L, 3); break; default: OCCD_LB(L, 4); }
    switch (lpv) {
        case 8: OCCD_LBV(8); break;
        case 16: OCCD_LBV(16); break;
        case 32: OCCD_LBV(32); break;
        default: OCCD_LBV(64); break;
    }
#undef OCCD_LBV
#undef OCCD_LB
    return occd::check_launch();
}

extern "C" int occd_flosp_sample_fwd(const occd_flosp_args* a, void* stream) {
    if (!a || !a->depth || !a->out) return OCCD_EINVAL;
    if (!a->grids && (!a->trans || !a->proj || !a->ida)) return OCCD_EINVAL;
    if (a->batch <= 0 || a->n_cams <= 0 || a->D <= 1 || a->h <= 0 || a->w <= 0) return OCCD_EINVAL;
    if (a->A <= 0 || a->Bdim <= 0 || a->C <= 0) return OCCD_EINVAL;
    FlospP p;
    p.a = *a;
    // python: bin_size = 2 * (depth_max - depth_min) / (num_bins * (1 + num_bins)) in double,
    // then used as an fp32 scalar operand
    p.bin_size = (float)(2.0 * ((double)a->depth_max - (double)a->depth_min) / ((double)a->D * (1.0 + a->D)));
    const long nvox = (long)a->A * a->Bdim * a->C;
    const double bytes = (double)a->batch * (nvox * (4.0 + 12.0 * (a->grids ? a->n_cams : 0)) +
                                             4.0 * a->n_cams * a->D * a->h * a->w);
    occd::ProfScope prof("flosp_sample", (hipStream_t)stream, 0.0, bytes);
    hipLaunchKernelGGL(flosp_sample_kernel, dim3((unsigned)((nvox + 255) / 256), (unsigned)a->batch), dim3(256), 0,
                       (hipStream_t)stream, p);
    return occd::check_launch();
}

extern "C" int occd_flosp_sample_bwd(const occd_flosp_bwd_args* q, void* stream) {
    if (!q || !q->gout || !q->gdepth || !q->workspace) return OCCD_EINVAL;
    const occd_flosp_args* a = &q->fwd;
    if (!a->grids && (!a->trans || !a->proj || !a->ida)) return OCCD_EINVAL;
    if (a->batch <= 0 || a->n_cams <= 0 || a->n_cams > OCCD_MAX_VIEWS || a->D <= 1 || a->h <= 0 || a->w <= 0) return OCCD_EINVAL;
    if (a->A <= 0 || a->Bdim <= 0 || a->C <= 0) return OCCD_EINVAL;
    const long nvox = (long)a->A * a->Bdim * a->C;
    const long vol = (long)a->batch * a->n_cams * a->D * a->h * a->w;
    if (q->workspace_bytes < (int64_t)vol * 8 + 8 || (reinterpret_cast<uintptr_t>(q->workspace) & 7)) return OCCD_EINVAL;
    FlospBwdP p{};
    p.f.a = *a;
    p.f.bin_size = (float)(2.0 * ((double)a->depth_max - (double)a->depth_min) / ((double)a->D * (1.0 + a->D)));
    p.gout = q->gout; p.gdepth = q->gdepth;
    p.acc = reinterpret_cast<long long*>(q->workspace);
    p.gmax_bits = reinterpret_cast<unsigned*>(p.acc + vol);
    p.total_out = (long)a->batch * nvox; p.total_vol = vol;
    int lg = 0;
    while ((1L << lg) < p.total_out) ++lg;
    p.log2_count = lg;
    hipStream_t st = (hipStream_t)stream;
    occd::ProfScope prof("flosp_sample_bwd", st, 0.0, 4.0 * p.total_out + 20.0 * vol);
    hipLaunchKernelGGL(flosp_bwd_zero_kernel, dim3(1), dim3(64), 0, st, p.gmax_bits);
    const long most = vol > p.total_out ? vol : p.total_out;
    long blocks = (most + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(flosp_bwd_prepare_kernel, dim3((unsigned)blocks), dim3(256), 0, st, p);
    hipLaunchKernelGGL(flosp_sample_bwd_kernel, dim3((unsigned)((nvox + 255) / 256), (unsigned)a->batch), dim3(256), 0, st, p);
    hipLaunchKernelGGL(flosp_bwd_finish_kernel, dim3((unsigned)((vol + 255) / 256)), dim3(256), 0, st, p);
    return occd::check_launch();
}

extern "C" int occd_lift_proj_fwd(const occd_lift_proj_args* q, void* stream) {
    if (!q || !q->cam_E || !q->cam_k || !q->lift.out) return OCCD_EINVAL;
    const occd_lift_args* a = &q->lift;
    if (a->n_scales < 1 || a->n_scales > OCCD_MAX_SCALES || a->n_views < 1 || a->n_views > OCCD_MAX_VIEWS)
        return OCCD_EINVAL;
    if (a->C <= 0 || (a->C & 3) || a->C > 256 || (a->out_cs & 3) || a->out_cs < a->C || a->out_cs > 256)
        return OCCD_EINVAL;
    if (a->N <= 0 || a->batch <= 0 || q->voxel_size <= 0 || q->img_w <= 0 || q->img_h <= 0 || q->img_w > 65535 ||
        q->img_h > 32767)
        return OCCD_EINVAL;
    if ((long)a->dimA * a->dimB * a->dimC != (long)a->N) return OCCD_EINVAL;
    for (int s = 0; s < a->n_scales; ++s) {
        if (a->scale_div[s] <= 0 || (a->feat_cs[s] & 3) || a->feat_cs[s] < a->C) return OCCD_EINVAL;
        for (int v = 0; v < a->n_views; ++v)
            if (!a->feat[s][v] || (reinterpret_cast<uintptr_t>(a->feat[s][v]) & 15)) return OCCD_EINVAL;
    }
    LiftProjP p;
    p.l.a = *a;
    p.l.a.pix = nullptr; p.l.a.fov = nullptr; p.l.a.depth_scale = nullptr; p.l.a.P = 1;
    auto lg2 = [](long v) { int s = 0; while ((1L << s) < v) ++s; return (1L << s) == v ? s : -1; };
    for (int s = 0; s < OCCD_MAX_SCALES; ++s) {
        p.l.sshift[s] = s < a->n_scales ? lg2(a->scale_div[s]) : 0;
        if (p.l.sshift[s] < 0) return OCCD_EINVAL;          // power-of-two scales and grids only (every KITTI config)
    }
    p.l.bc_shift = lg2((long)a->dimB * a->dimC);
    p.l.c_shift = lg2(a->dimC);
    if (p.l.bc_shift < 0 || p.l.c_shift < 0) return OCCD_EINVAL;
    for (int s = 0; s < a->n_scales; ++s)
        for (int v = 0; v < a->n_views; ++v)
            if (p.l.a.feat_bstride[s][v] == 0) p.l.a.feat_bstride[s][v] = (int64_t)a->feat_h[s] * a->feat_w[s] * a->feat_cs[s];
    p.cam_E = q->cam_E;
    p.cam_k = q->cam_k;
    p.vox_size = q->voxel_size;
    for (int j = 0; j < 3; ++j) p.origin[j] = q->origin[j];
    p.img_w = q->img_w; p.img_h = q->img_h;
    p.has_frustum = q->frustum.depth != nullptr;
    if (p.has_frustum) {
        const occd_flosp_args* f = &q->frustum;
        if (!f->grids && (!f->trans || !f->proj || !f->ida)) return OCCD_EINVAL;
        if (f->batch != a->batch || f->n_cams <= 0 || f->D <= 1 || f->h <= 0 || f->w <= 0) return OCCD_EINVAL;
        if (f->A != a->dimA || f->Bdim != a->dimB || f->C != a->dimC) return OCCD_EINVAL;
        p.fr.a = *f;
        p.fr.bin_size = (float)(2.0 * ((double)f->depth_max - (double)f->depth_min) / ((double)f->D * (1.0 + f->D)));
    } else {
        p.fr = FlospP{};
    }
    const int need = a->out_cs / 4;
    const int lpv = need <= 8 ? 8 : need <= 16 ? 16 : need <= 32 ? 32 : 64;
    const int T = 4 * (256 / lpv);
    if (p.l.a.xcd_mode < 0 || p.l.a.xcd_mode > 2) return OCCD_EINVAL;
    if (p.l.a.xcd_mode == 2 && (((long)a->dimB * a->dimC) % ((long)T * 8) != 0 || a->N % T != 0)) p.l.a.xcd_mode = 0;
    const dim3 grid((unsigned)((a->N + T - 1) / T), (unsigned)a->batch);
    // algorithmic bytes: output rows + one gathered pixel row per view / scale (+ the depth volumes once)
    const double bytes = (double)a->batch * a->N * (4.0 * a->C * (1 + (double)a->n_views * a->n_scales)) +
                         (p.has_frustum ? 4.0 * a->batch * q->frustum.n_cams * q->frustum.D * q->frustum.h * q->frustum.w : 0.0);
    occd::ProfScope prof("sfa_lift_proj", (hipStream_t)stream, 0.0, bytes);
    hipStream_t st = (hipStream_t)stream;
    // voxels in flight per lane group: 1 measured best at config 2 (60.8 us; 2: 71.6 us, 4: 83.9 us -- more registers,
    // half the waves, profiles/r03_lift_proj.txt); OCCD_LIFT_INFLIGHT is a tuning knob, read once
    static const int inflight = [] {
        const char* e = getenv("OCCD_LIFT_INFLIGHT");
        const int v = e ? atoi(e) : 1;
        return v == 2 || v == 4 ? v : 1;
    }();
#define OCCD_LP(L, VV)                                                                                            \
    if (lpv == L && a->n_views == VV) {                                                                           \
        if (inflight == 1) hipLaunchKernelGGL((lift_proj_kernel<L, VV, 1>), grid, dim3(256), 0, st, p);           \
        else if (inflight == 4) hipLaunchKernelGGL((lift_proj_kernel<L, VV, 4>), grid, dim3(256), 0, st, p);      \
        else hipLaunchKernelGGL((lift_proj_kernel<L, VV, 2>), grid, dim3(256), 0, st, p);                         \
    }
    OCCD_LP(8, 1) OCCD_LP(8, 2) OCCD_LP(16, 1) OCCD_LP(16, 2) OCCD_LP(32, 1) OCCD_LP(32, 2) OCCD_LP(64, 1) OCCD_LP(64, 2)
#undef OCCD_LP
    if (a->n_views > 2) return OCCD_EINVAL;
    return occd::check_launch();
}

extern "C" int occd_nchw_to_nhwc(const float* in, float* out, int32_t batch, int32_t C, int64_t S, int32_t cs,
                                 void* stream) {
    if (!in || !out || batch <= 0 || C <= 0 || S <= 0 || cs < C) return OCCD_EINVAL;
    occd::ProfScope prof("nchw_to_nhwc", (hipStream_t)stream, 0.0, 4.0 * batch * S * (C + cs));
    hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3((unsigned)((S + 63) / 64), (unsigned)((cs + 31) / 32), (unsigned)batch),
                       dim3(256), 0, (hipStream_t)stream, in, out, C, (long)S, cs);
    return occd::check_launch();
}

extern "C" int occd_nhwc_to_nchw(const float* in, float* out, int32_t batch, int32_t C, int64_t S, int32_t cs,
                                 int32_t coff, void* stream) {
    if (!in || !out || batch <= 0 || C <= 0 || S <= 0 || coff < 0 || coff + C > cs) return OCCD_EINVAL;
    occd::ProfScope prof("nhwc_to_nchw", (hipStream_t)stream, 0.0, 8.0 * batch * S * C);
    hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3((unsigned)((S + 63) / 64), (unsigned)((C + 31) / 32), (unsigned)batch),
                       dim3(256), 0, (hipStream_t)stream, in, out, C, (long)S, cs, coff);
    return occd::check_launch();
}

extern "C" int occd_softmax_channels(const float* src, float* dst, int64_t rows, int32_t src_cs, int32_t src_coff,
                                     int32_t dst_cs, int32_t dst_coff, int32_t n, int32_t dst_pad, void* stream) {
    if (!src || !dst || rows <= 0 || n <= 0 || n > 64 || dst_pad < 0 || src_coff + n > src_cs ||
        dst_coff + n + dst_pad > dst_cs)
        return OCCD_EINVAL;
    occd::ProfScope prof("softmax_channels", (hipStream_t)stream, 0.0, 8.0 * rows * n);
    hipLaunchKernelGGL(softmax_channels_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, src, dst, (long)rows, src_cs, src_coff, dst_cs, dst_coff, n, dst_pad);
    return occd::check_launch();
}

// ------------------------------------------------------------------------------------------------
// voxel-centroid -> pixel projection on the GPU (the dataloader's numba `vox2pix`):
// occdepth/data/utils/helpers.py:94-169 with fusion.py:203-217 (vox2world: float32 origin, float64 arithmetic,
// float32 store), :518-522 (rigid transform in float64) and :336-337 (round(x * fx / z + cx), float32
// intrinsics, numpy round-half-even).  Integer outputs: every product / sum below is an explicitly rounded
// IEEE double operation (no FMA contraction) so the pixels are the ones numpy computes.
namespace {

struct ProjP {
    double E[16];                 // cam_E (world/lidar -> camera), row major
    double fx, fy, cx, cy;        // float32-rounded intrinsics, widened
    double vox_size;
    float origin[3];              // float32(vox_origin)
    int X, Y, Z;                  // voxel grid (ceil(scene / voxel_size))
    int img_w, img_h;
    int64_t* pix;                 // (N, 1, 2) int64
    uint8_t* fov;                 // (N, 1) bool
    float* pix_z;                 // (N,) or nullptr
};

__global__ void __launch_bounds__(256) project_voxels_kernel(const ProjP p) {
    const long n = (long)blockIdx.x * 256 + threadIdx.x;
    const long total = (long)p.X * p.Y * p.Z;
    if (n >= total) return;
    const int iz = (int)(n % p.Z);
    const long t = n / p.Z;
    const int iy = (int)(t % p.Y), ix = (int)(t / p.Y);
    long px, py;
    double camz;
    const bool in = project_one(p.E, p.fx, p.fy, p.cx, p.cy, p.vox_size, p.origin, ix, iy, iz, p.img_w, p.img_h, px, py,
                                camz);
    p.pix[n * 2] = px;
    p.pix[n * 2 + 1] = py;
    p.fov[n] = in ? 1 : 0;
    if (p.pix_z) p.pix_z[n] = (float)camz;
}

}  // namespace

extern "C" int occd_project_voxels(const double* cam_E_host, const double* cam_k_host, const double* vox_origin_host,
                                   double voxel_size, int32_t X, int32_t Y, int32_t Z, int32_t img_w, int32_t img_h,
                                   int64_t* pix, uint8_t* fov, float* pix_z, void* stream) {
    if (!cam_E_host || !cam_k_host || !vox_origin_host || !pix || !fov || X <= 0 || Y <= 0 || Z <= 0 ||
        voxel_size <= 0)
        return OCCD_EINVAL;
    ProjP p;
    for (int i = 0; i < 16; ++i) p.E[i] = cam_E_host[i];
    p.fx = (double)(float)cam_k_host[0];
    p.fy = (double)(float)cam_k_host[4];
    p.cx = (double)(float)cam_k_host[2];
    p.cy = (double)(float)cam_k_host[5];
    p.vox_size = voxel_size;
    for (int j = 0; j < 3; ++j) p.origin[j] = (float)vox_origin_host[j];
    p.X = X; p.Y = Y; p.Z = Z; p.img_w = img_w; p.img_h = img_h;
    p.pix = pix; p.fov = fov; p.pix_z = pix_z;
    const long total = (long)X * Y * Z;
    occd::ProfScope prof("project_voxels", (hipStream_t)stream, 0.0, 17.0 * total);
    hipLaunchKernelGGL(project_voxels_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       p);
    return occd::check_launch();
}
