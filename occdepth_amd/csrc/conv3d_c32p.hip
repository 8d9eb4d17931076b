// Persistent, weights-stationary 3x3x3 convolutions for <=32 -> <=32 channels on Z = 32 k columns (32: config 2, 64:
// BASELINE configs[4]; K2s tiles z in 32-column tiles whose halo is real neighbour data): the
// segmentation-head convolutions at full resolution (7-8 launches, 87 % of the 3-D stack's FLOPs) and, in training,
// their data gradients.  Two kernels, same math as conv3d_igemm_kernel (v_mfma_f32_32x32x2_f32, exact fp32):
//
//   K2s conv3d_c32_slide_kernel  (default)  sliding window along x with three accumulators per wave: every staged
//       input plane feeds all three kx taps (staged once instead of three times); see the comment above it.
//   K2p conv3d_c32_persist_kernel (OCCD_C32P_TILED=1, kept for A/B runs) one accumulator per wave, tile by tile:
//       XCD k owns a contiguous range of x-planes, each workgroup walks along x at a fixed y-tile so the kx halo
//       planes of consecutive tiles are L2 hits; each (kx, 16-channel) slab is register-staged (global -> VGPR -> LDS,
//       one LDS slab buffer, two barriers per slab) under the 72 MFMAs of the previous one.
// Common to both: one 512-thread workgroup per CU stays resident; ALL 27 x 32 x 32 weights (110.6 KB, MFMA B-fragment
// order) are loaded into LDS once per workgroup -- no per-wave weight stream from L2 (3x the activation traffic in the
// generic kernel); wave w owns the 32 voxels of row y0+w (one M tile), 2 waves per SIMD; operands are
// (weights, activations) so a lane ends up with four float4 groups of consecutive couts of ONE voxel (16-byte
// residual loads / stores).
// LDS: 110,592 B weights + (8+2d)(32+2d) rows x 80 B  (d=1: 137.8 KB, d=3: 153.2 KB) -> 1 workgroup / CU.
//
// Reference semantics replaced: occdepth/models/modules.py:158-175 (conv0, conv1.*, conv2.*, conv_classes).
#include <atomic>
#include <mutex>
#include <type_traits>
#include <cstdlib>
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

struct PersistP {
    const float* in;
    const float* wpk;
    const float* bias;
    const float* res1;
    const float* res2;
    float* out;
    int batch, X, Y, Z, in_cs, in_coff;      // Z: a multiple of kTZ (K2s); K2p: Z == kTZ
    int out_cs, out_coff, res1_cs, res1_coff, res2_cs, res2_coff;
    int act_in, act_out, cout_store;
    int ytiles, ztiles, tiles_total;
};

constexpr int kTY = 8, kTZ = 32, kWTaps = 27, kWFloat4 = kWTaps * 4 * 64;  // 6912 float4 = 110,592 B

// Walk order of the x planes: residue classes of the dilation (0, D, 2D, ..., 1, 1+D, ...), so that two
// consecutive tiles of a workgroup always share two of their three kx planes (x-D, x, x+D) in L2.  With the
// natural order a dilated convolution re-fetched every plane three times from the fabric (PMC: 800 MB vs 284 MB).
template <int D>
__device__ __forceinline__ int plane_of(int j, int X) {
    if (D == 1) return j;
#pragma unroll
    for (int r = 0; r < D; ++r) {
        const int n_r = (X - r + D - 1) / D;
        if (j < n_r) return r + D * j;
        j -= n_r;
    }
    return X - 1;
}

template <int D>
__global__ void __launch_bounds__(512, 2) conv3d_c32_persist_kernel(const PersistP p) {
    constexpr int YIN = kTY + 2 * D, ZIN = kTZ + 2 * D, ROWS = YIN * ZIN;
    constexpr int RS4 = 5;                       // 16 floats + 4 pad per LDS row: odd number of 16-B slots
    constexpr int NF4 = ROWS * 4;                // float4 slots of one slab
    constexpr int NLOAD = (NF4 + 511) / 512;
    extern __shared__ __attribute__((aligned(16))) f32x4 lds4[];
    f32x4* const w4 = lds4;
    f32x4* const slab4 = lds4 + kWFloat4;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // 0..7 = local y row
    const int li = lane & 31, kk = lane >> 5;

    // weights: global (packed) -> LDS, once
    for (int i = tid; i < kWFloat4; i += 512) w4[i] = ((const f32x4*)p.wpk)[i];

    // per-thread staging descriptors (constant over tiles)
    int sdst[NLOAD], syi[NLOAD], szoff[NLOAD];
#pragma unroll
    for (int i = 0; i < NLOAD; ++i) {
        const int f = tid + i * 512;
        const bool live = f < NF4;
        const int row = f >> 2, c4 = f & 3;
        const int yi = row / ZIN, zi = row - yi * ZIN;
        const int z = zi - D;
        sdst[i] = live ? row * RS4 + c4 : -1;
        syi[i] = yi;
        szoff[i] = (live && z >= 0 && z < kTZ) ? z * p.in_cs + c4 * 4 : -1;
    }
    const int rowbase = (wave * ZIN + li) * RS4 + kk;   // A fragment row of this lane (tap (0,0), kt 0)

    // tile walk: XCD-major ranges of the (b, x, ytile) list, workgroups of one XCD interleaved by y-tile
    const int nwg = gridDim.x;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int wg_per_xcd = (nwg + 7 - xcd) >> 3;          // workgroups that landed on this XCD id
    const int per_xcd = (p.tiles_total + 7) >> 3;
    const int t_end = min((xcd + 1) * per_xcd, p.tiles_total);
    int tile = xcd * per_xcd + slot;
    if (tile >= t_end) return;

    f32x4 v[NLOAD];
    auto issue = [&](int t, int ci) {
        const int yt = t % p.ytiles;
        const int bj = t / p.ytiles;                       // b * X + (position of the plane in the walk order)
        const int x = plane_of<D>(bj % p.X, p.X);
        const int bx = bj - bj % p.X + x;                  // b * X + x
        const int kx = ci >> 1, h = ci & 1;
        const int xi = x - D + kx * D;
        const bool plane_ok = xi >= 0 && xi < p.X;
        const float* base = p.in + ((size_t)(bx - x + (plane_ok ? xi : 0)) * p.Y) * kTZ * p.in_cs + p.in_coff + h * 16;
        const int y0 = yt * kTY - D;
#pragma unroll
        for (int i = 0; i < NLOAD; ++i) {
            const int y = y0 + syi[i];
            const bool ok = plane_ok && szoff[i] >= 0 && y >= 0 && y < p.Y;
            v[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (ok) v[i] = *(const f32x4*)(base + (size_t)y * kTZ * p.in_cs + szoff[i]);
        }
    };

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;

    issue(tile, 0);

    while (true) {
        const int next_tile = tile + wg_per_xcd;
        const bool has_next = next_tile < t_end;
#pragma unroll 1
        for (int ci = 0; ci < 6; ++ci) {
            __syncthreads();                      // previous slab fully consumed (and weights visible)
#pragma unroll
            for (int i = 0; i < NLOAD; ++i)
                if (sdst[i] >= 0) {
                    f32x4 a = v[i];
                    if (p.act_in == OCCD_ACT_RELU) {
                        a.x = fmaxf(a.x, 0.f); a.y = fmaxf(a.y, 0.f); a.z = fmaxf(a.z, 0.f); a.w = fmaxf(a.w, 0.f);
                    }
                    slab4[sdst[i]] = a;
                }
            __syncthreads();
            if (ci < 5) issue(tile, ci + 1);      // lands while the MFMAs below run
            else if (has_next) issue(next_tile, 0);
            const int kx = ci >> 1, h = ci & 1;
            const f32x4* wb = w4 + (kx * 9 * 4 + h * 2) * 64 + lane;
            const f32x4* ab = slab4 + rowbase;
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kz = 0; kz < 3; ++kz)
#pragma unroll
                    for (int ktl = 0; ktl < 2; ++ktl) {
                        const f32x4 a = ab[(ky * D * ZIN + kz * D) * RS4 + ktl * 2];
                        const f32x4 b = wb[((ky * 3 + kz) * 4 + ktl) * 64];
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(b[q], a[q], acc, 0, 0, 0);   // D = W^T . X^T
                    }
        }
        // ---- epilogue of this tile (the next tile's first slab is already in flight).  Operands are
        // (weights, activations): lane -> voxel z = li, registers -> couts (r & 3) + 8 (r >> 2) + 4 kk, i.e. four
        // float4 groups of consecutive channels per lane -> 16-byte residual loads / stores.
        {
            const int yt = tile % p.ytiles;
            const int bj = tile / p.ytiles;
            const int bx = bj - bj % p.X + plane_of<D>(bj % p.X, p.X);
            const int y = yt * kTY + wave;
            if (y < p.Y) {
                const size_t vox = ((size_t)bx * p.Y + y) * kTZ + li;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int c = 8 * g + 4 * kk;
                    if (c < p.cout_store) {
                        f32x4 o = {acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]};
                        if (p.bias != nullptr) o += *(const f32x4*)(p.bias + c);
                        if (p.act_out == OCCD_ACT_RELU_PRE) {
                            o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f);
                        }
                        if (p.res1 != nullptr) o += *(const f32x4*)(p.res1 + vox * p.res1_cs + p.res1_coff + c);
                        if (p.res2 != nullptr) o += *(const f32x4*)(p.res2 + vox * p.res2_cs + p.res2_coff + c);
                        if (p.act_out == OCCD_ACT_RELU) {
                            o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f);
                        }
                        *(f32x4*)(p.out + vox * p.out_cs + p.out_coff + c) = o;
                    }
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        }
        if (!has_next) break;
        tile = next_tile;
    }
}

// ------------------------------------------------------------------------------------------------
// K2s -- sliding-window form of the persistent kernel.  A workgroup walks a run of x planes (step D) at a fixed
// y tile and keeps THREE accumulators per wave: acc0 = out[x+D] (fed by kx = 0), acc1 = out[x] (kx = 1),
// acc2 = out[x-D] (kx = 2).  Every staged input slab (plane x, 16-channel half) therefore feeds all three kx
// taps: each input plane is staged ONCE per workgroup instead of three times, each A fragment read from LDS is
// used by 12 MFMAs instead of 4, consecutive MFMAs are independent, and there are 216 MFMAs per wave between
// barrier pairs instead of 72.  After a plane, acc2 is complete -> epilogue; the accumulators rotate.
// Work list: per (b, y tile) column the X planes in residue-class order (plane_of) are cut into `segs_per_col`
// equal position ranges; a range that crosses a residue boundary is walked as two runs.  Ranges are handed out
// through a global counter.
struct SlideP {
    PersistP base;
    int* counter;            // {next segment, workgroups finished}: both 0 at launch, re-armed by the kernel itself
    int segs_per_col;        // position ranges per (b, ytile) column
    int seg_len;             // planes per range (the last one of a column may be shorter)
    int total_segs;
};

template <int D>
__global__ void __launch_bounds__(512, 2) conv3d_c32_slide_kernel(const SlideP sp) {
    const PersistP& p = sp.base;
    constexpr int YIN = kTY + 2 * D, ZIN = kTZ + 2 * D, ROWS = YIN * ZIN;
    // channels staged per slab.  (CH = 32 for D = 1 fits LDS -- 110.6 KB of weights + 49 KB -- and halves the barrier
    // pairs, but its 6 staging float4 per thread push the kernel past 256 VGPRs: measured as spills, not as a gain.)
    constexpr int CH = 16, KT = CH / 8, NH = 32 / CH;
    constexpr int RS4 = CH / 4 + 1;              // CH floats + 4 pad per LDS row: odd number of 16-B slots
    constexpr int NF4 = ROWS * (CH / 4);
    constexpr int NLOAD = (NF4 + 511) / 512;
    extern __shared__ __attribute__((aligned(16))) f32x4 lds4[];
    f32x4* const w4 = lds4;
    f32x4* const slab4 = lds4 + kWFloat4;
    int* const mailbox = reinterpret_cast<int*>(slab4 + ROWS * RS4);   // 16 bytes after the slab

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, kk = lane >> 5;

    {   // weights: global (packed) -> LDS, once; 7 loads in flight per thread (6912 float4 = 13.5 x 512)
        static_assert(kWFloat4 == 13 * 512 + 256, "weight staging loop");
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            f32x4 wv[7];
#pragma unroll
            for (int u = 0; u < 7; ++u) {
                const int i = tid + (half * 7 + u) * 512;
                if (i < kWFloat4) wv[u] = ((const f32x4*)p.wpk)[i];
            }
#pragma unroll
            for (int u = 0; u < 7; ++u) {
                const int i = tid + (half * 7 + u) * 512;
                if (i < kWFloat4) w4[i] = wv[u];
            }
        }
    }

    // staging descriptors: slab row (yi, zi) and 4-channel group c4 of this thread's i-th float4; the z tile (columns
    // z0 .. z0 + 31 of a Z = 32, 64, ... volume) enters per segment, so a tile's z halo is the neighbouring tile's data
    int sdst[NLOAD], syi[NLOAD], szi[NLOAD], sc4[NLOAD];
#pragma unroll
    for (int i = 0; i < NLOAD; ++i) {
        const int f = tid + i * 512;
        const bool live = f < NF4;
        const int row = f / (CH / 4), c4 = f - row * (CH / 4);
        const int yi = row / ZIN, zi = row - yi * ZIN;
        sdst[i] = live ? row * RS4 + c4 : -1;
        syi[i] = yi;
        szi[i] = live ? zi - D : -(1 << 20);
        sc4[i] = c4 * 4;
    }
    const f32x4* const ab = slab4 + (wave * ZIN + li) * RS4 + kk;
    const size_t plane_stride = (size_t)p.Y * p.Z * p.in_cs;

    // per-column staging addresses (element offsets of plane x = 0), refreshed per segment
    unsigned coloff[NLOAD];                  // < 2^32 elements: checked by the host
    bool colok[NLOAD];
    f32x4 v[NLOAD];
    auto issue = [&](int xi, int h) {        // global -> registers; xi inside the volume
        const float* base = p.in + (size_t)xi * plane_stride + h * CH;
#pragma unroll
        for (int i = 0; i < NLOAD; ++i) {
            v[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (colok[i]) v[i] = *(const f32x4*)(base + coloff[i]);
        }
    };
    auto commit = [&]() {                    // registers -> LDS slab
#pragma unroll
        for (int i = 0; i < NLOAD; ++i)
            if (sdst[i] >= 0) {
                f32x4 a = v[i];
                if (p.act_in == OCCD_ACT_RELU) {
                    a.x = fmaxf(a.x, 0.f); a.y = fmaxf(a.y, 0.f); a.z = fmaxf(a.z, 0.f); a.w = fmaxf(a.w, 0.f);
                }
                slab4[sdst[i]] = a;
            }
    };

    f32x16 acc0, acc1, acc2;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc0[r] = acc1[r] = acc2[r] = 0.f;

    // one staged slab into the accumulators whose output plane exists (U0: kx = 0 -> acc0, U1: kx = 1 -> acc1,
    // U2: kx = 2 -> acc2; compile-time so every variant is straight-line code); the LDS reads of step s + 1 are in
    // flight under the MFMAs of step s (sched_barrier: the scheduler sinks them otherwise)
    auto mma = [&](int h, auto u0, auto u1, auto u2) {
        constexpr bool U0 = decltype(u0)::value, U1 = decltype(u1)::value, U2 = decltype(u2)::value;
        const f32x4* wb = w4 + (h * KT) * 64 + lane;
        constexpr int KXS = 9 * 4 * 64, NS = 9 * KT;
        auto aoff = [](int s) { return (((s / (3 * KT)) * D * ZIN) + ((s / KT) % 3) * D) * RS4 + (s % KT) * 2; };
        auto woff = [](int s) { return ((s / KT) * 4 + (s % KT)) * 64; };
        f32x4 an = ab[aoff(0)], b0n, b1n, b2n;
        if (U0) b0n = wb[woff(0)];
        if (U1) b1n = wb[woff(0) + KXS];
        if (U2) b2n = wb[woff(0) + 2 * KXS];
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const f32x4 a = an, b0 = b0n, b1 = b1n, b2 = b2n;
            if (s < NS - 1) {
                an = ab[aoff(s + 1)];
                if (U0) b0n = wb[woff(s + 1)];
                if (U1) b1n = wb[woff(s + 1) + KXS];
                if (U2) b2n = wb[woff(s + 1) + 2 * KXS];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (U0) acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(b0[q], a[q], acc0, 0, 0, 0);
                if (U1) acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(b1[q], a[q], acc1, 0, 0, 0);
                if (U2) acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(b2[q], a[q], acc2, 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    using T_ = std::true_type;
    using F_ = std::false_type;
    // epilogue of acc2 (output plane x of column (b, yt)): lane -> voxel z = li, registers -> couts
    // (r & 3) + 8 (r >> 2) + 4 kk, i.e. four float4 groups of consecutive channels per lane.  The residual rows
    // are fetched one slab early (res_fetch) so their latency hides under 216 MFMAs.
    f32x4 r1[4], r2[4];
    f32x4* const bias4 = slab4 + ROWS * RS4 + 1;                        // 32 floats after the mailbox
    if (tid < 8) {
        f32x4 bv = {0.f, 0.f, 0.f, 0.f};
        if (p.bias != nullptr && 4 * tid < p.cout_store) bv = *(const f32x4*)(p.bias + 4 * tid);
        bias4[tid] = bv;
    }
    int z0 = 0;                                                       // first column of the segment's z tile
    auto res_fetch = [&](int b, int yt, int x) {
        const int y = min(yt * kTY + wave, p.Y - 1);
        const size_t vox = ((size_t)(b * p.X + x) * p.Y + y) * p.Z + z0 + li;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int c = 8 * g + 4 * kk;
            if (c < p.cout_store) {
                if (p.res1 != nullptr) r1[g] = *(const f32x4*)(p.res1 + vox * p.res1_cs + p.res1_coff + c);
                if (p.res2 != nullptr) r2[g] = *(const f32x4*)(p.res2 + vox * p.res2_cs + p.res2_coff + c);
            }
        }
    };
    auto store2 = [&](int b, int yt, int x) {
        const int y = yt * kTY + wave;
        if (y < p.Y) {
            const size_t vox = ((size_t)(b * p.X + x) * p.Y + y) * p.Z + z0 + li;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int c = 8 * g + 4 * kk;
                if (c < p.cout_store) {
                    f32x4 o = {acc2[4 * g], acc2[4 * g + 1], acc2[4 * g + 2], acc2[4 * g + 3]};
                    o += bias4[2 * g + kk];
                    if (p.act_out == OCCD_ACT_RELU_PRE) {
                        o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f);
                    }
                    if (p.res1 != nullptr) o += r1[g];
                    if (p.res2 != nullptr) o += r2[g];
                    if (p.act_out == OCCD_ACT_RELU) {
                        o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f);
                    }
                    *(f32x4*)(p.out + vox * p.out_cs + p.out_coff + c) = o;
                }
            }
        }
    };

    while (true) {
        __syncthreads();                                   // mailbox / slab free, weights visible
        if (tid == 0) mailbox[0] = atomicAdd(sp.counter, 1);
        __syncthreads();
        const int seg = mailbox[0];
        if (seg >= sp.total_segs) {
            // the last workgroup to run dry re-arms the pair of counters for the next launch that uses this slot
            if (tid == 0 && atomicAdd(sp.counter + 1, 1) == (int)gridDim.x - 1) {
                sp.counter[1] = 0;
                __threadfence();
                atomicExch(sp.counter, 0);
            }
            break;
        }
        // segments are ordered (b, range, ytile, ztile) with the z tile fastest
        const int tz = seg % (p.ytiles * p.ztiles);
        const int zt = tz % p.ztiles, yt = tz / p.ztiles;
        const int rest = seg / (p.ytiles * p.ztiles);
        const int b = rest / sp.segs_per_col;
        const int q0 = (rest - b * sp.segs_per_col) * sp.seg_len;
        const int q1 = min(q0 + sp.seg_len, p.X);
        z0 = zt * kTZ;
#pragma unroll
        for (int i = 0; i < NLOAD; ++i) {
            const int y = yt * kTY - D + syi[i];
            const int z = z0 + szi[i];
            colok[i] = z >= 0 && z < p.Z && y >= 0 && y < p.Y;
            coloff[i] = (unsigned)((((size_t)b * p.X * p.Y + (colok[i] ? y : 0)) * p.Z + (colok[i] ? z : 0)) * p.in_cs +
                                   p.in_coff + sc4[i]);
        }
        int q = q0;
        while (q < q1) {
            // run: positions q .. q + cnt - 1 of one residue class -> outputs x0, x0 + D, ...
            int r = 0, idx = q, n_r = p.X;
            if (D > 1) {
#pragma unroll
                for (int rr = 0; rr < D; ++rr) {
                    const int n = (p.X - rr + D - 1) / D;
                    if (idx < n || rr == D - 1) { r = rr; n_r = n; break; }
                    idx -= n;
                }
            }
            const int cnt = min(q1 - q, n_r - idx);
            const int x0 = r + D * idx;
            q += cnt;
            // input planes j = 0 .. cnt + 1 : xi = x0 + (j - 1) D feeds out[j] (kx 0), out[j-1] (kx 1), out[j-2] (kx 2)
            const int nj = cnt + 2;
            const int jfirst = x0 - D < 0 ? 1 : 0;                         // plane -D.. is padding
            const int jlast = x0 + cnt * D >= p.X ? (p.X - 1 - x0) / D + 1 : nj - 1;   // last plane inside the volume
            __syncthreads();                                               // previous run's slab consumed
            issue(x0 + (jfirst - 1) * D, 0);
            for (int j = 0; j < nj; ++j) {
                const int xi = x0 + (j - 1) * D;
                const bool u0 = j < cnt, u1 = j >= 1 && j <= cnt, u2 = j >= 2;
                if (j >= jfirst && j <= jlast) {
#pragma unroll 1
                    for (int h = 0; h < NH; ++h) {
                        __syncthreads();                                   // previous slab consumed
                        commit();
                        __syncthreads();
                        if (h + 1 < NH) issue(xi, h + 1);
                        else {
                            if (j < jlast) issue(xi + D, 0);
                            if (u2) res_fetch(b, yt, xi - D);              // out[j-2] completes with this slab
                        }
                        if (u0 && u1 && u2) mma(h, T_{}, T_{}, T_{});      // interior plane
                        else if (u0 && u1) mma(h, T_{}, T_{}, F_{});       // second plane of a run
                        else if (u1 && u2) mma(h, F_{}, T_{}, T_{});       // second to last
                        else if (u0) mma(h, T_{}, F_{}, F_{});             // first
                        else if (u2) mma(h, F_{}, F_{}, T_{});             // last
                        else mma(h, F_{}, T_{}, F_{});                     // run of a single output plane
                    }
                }
                if (u2) {
                    if (j < jfirst || j > jlast) res_fetch(b, yt, xi - D);   // padding plane: nothing was staged
                    store2(b, yt, xi - D);
                }
                acc2 = acc1;
                acc1 = acc0;
#pragma unroll
                for (int rr = 0; rr < 16; ++rr) acc0[rr] = 0.f;
            }
        }
    }
}

// Per-DEVICE launch state (work-list counters live in that device's memory, CU count and the large-LDS function
// attribute belong to it): a process that drives several GPUs gets one of these per device, looked up from the
// current device at every launch.
constexpr int kMaxDevices = 64;
struct DevState {
    std::mutex mu;
    int* counter = nullptr;
    int num_cu = 0;
    bool slide_attr[4] = {};
    bool attr_done[4] = {};
};
DevState g_dev[kMaxDevices];
std::atomic<unsigned> g_slot{0};

DevState* dev_state() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) return nullptr;
    DevState* s = &g_dev[dev];
    std::lock_guard<std::mutex> lock(s->mu);
    if (s->num_cu == 0) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return nullptr;
        s->num_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    return s;
}

template <int D>
int launch_slide(const PersistP& base, hipStream_t st, DevState* ds) {
    constexpr int ROWS = (kTY + 2 * D) * (kTZ + 2 * D), RS4 = 16 / 4 + 1;
    const size_t lds = (size_t)kWFloat4 * 16 + (size_t)ROWS * RS4 * 16 + 16 + 128;
    const int num_cu = ds->num_cu;
    // a ring of self re-arming counter pairs: launches in flight on different streams never share one
    constexpr int kSlots = 256;
    {
        std::lock_guard<std::mutex> lock(ds->mu);
        if (!ds->slide_attr[D]) {
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(conv3d_c32_slide_kernel<D>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
                return OCCD_ELAUNCH;
            ds->slide_attr[D] = true;
        }
        if (ds->counter == nullptr) {
            if (hipMalloc(&ds->counter, kSlots * 2 * sizeof(int)) != hipSuccess) return OCCD_ELAUNCH;
            if (hipMemset(ds->counter, 0, kSlots * 2 * sizeof(int)) != hipSuccess) return OCCD_ELAUNCH;
        }
    }
    SlideP sp;
    sp.base = base;
    sp.counter = ds->counter + 2 * (g_slot.fetch_add(1) % kSlots);
    // ranges per column: k rounds over the grid; a range of L planes costs L + 2 stagings per run (D > 1: up to two
    // runs).  Few long ranges amortise the two extra planes, but the list must fill whole rounds.
    const int cols = base.batch * base.ytiles * base.ztiles;
    int best_s = 1;
    double best_cost = 1e30;
    for (int k = 1; k <= 16; ++k) {
        int S = (int)((long)k * num_cu / cols);
        if (S < 1) S = 1;
        if (S > base.X) S = base.X;
        const int L = (base.X + S - 1) / S;
        const long total = (long)cols * ((base.X + L - 1) / L);
        const long rounds = (total + num_cu - 1) / num_cu;
        const double cost = (double)rounds * (L + 2.0 + (D > 1 ? 1.0 : 0.0));
        if (cost < best_cost - 1e-9) { best_cost = cost; best_s = S; }
    }
    sp.seg_len = (base.X + best_s - 1) / best_s;
    sp.segs_per_col = (base.X + sp.seg_len - 1) / sp.seg_len;
    sp.total_segs = cols * sp.segs_per_col;
    int grid = num_cu < sp.total_segs ? num_cu : sp.total_segs;
    hipLaunchKernelGGL(conv3d_c32_slide_kernel<D>, dim3((unsigned)grid), dim3(512), lds, st, sp);
    return occd::check_launch();
}

template <int D>
int launch(const PersistP& p, hipStream_t st, DevState* ds) {
    constexpr int ROWS = (kTY + 2 * D) * (kTZ + 2 * D);
    const size_t lds = (size_t)kWFloat4 * 16 + (size_t)ROWS * 5 * 16;
    {
        std::lock_guard<std::mutex> lock(ds->mu);
        if (!ds->attr_done[D]) {
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(conv3d_c32_persist_kernel<D>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
                return OCCD_ELAUNCH;
            ds->attr_done[D] = true;
        }
    }
    int grid = ds->num_cu;
    if (grid > p.tiles_total) grid = p.tiles_total;
    hipLaunchKernelGGL(conv3d_c32_persist_kernel<D>, dim3((unsigned)grid), dim3(512), lds, st, p);
    return occd::check_launch();
}

}  // namespace

namespace occd {

// Returns 1 when the launch was taken by the persistent kernel, 0 when the geometry does not qualify,
// <0 on error.
int try_conv3d_c32_persist(const occd_conv3d_args* a, hipStream_t stream) {
    const int d = a->dx;
    const bool geom = a->kx == 3 && a->ky == 3 && a->kz == 3 && a->sx == 1 && a->sy == 1 && a->sz == 1 &&
                      a->dy == d && a->dz == d && d >= 1 && d <= 3 && a->px == d && a->py == d && a->pz == d;
    static const bool tiled = getenv("OCCD_C32P_TILED") != nullptr;   // A/B switch: per-tile variant (K2p) vs sliding window (K2s)
    const bool shape = a->Z % kTZ == 0 && (a->Z == kTZ || !tiled) && a->Xo == a->X && a->Yo == a->Y && a->Zo == a->Z && a->OX == a->X &&
                       a->OY == a->Y && a->OZ == a->Z && a->o_stride_x == 1 && a->o_stride_y == 1 &&
                       a->o_stride_z == 1 && a->o_off_x == 0 && a->o_off_y == 0 && a->o_off_z == 0;
    const int cin8 = (a->cin + 7) & ~7;
    const bool chans = cin8 <= 32 && a->cout <= 32 && a->in_coff + 32 <= a->in_cs && a->act_in != OCCD_ACT_SIGMOID;
    // enough tiles to keep every CU busy for several rounds, otherwise the generic kernel tiles finer
    const long tiles = (long)a->batch * a->X * ((a->Y + kTY - 1) / kTY) * (a->Z / kTZ);
    if (!(geom && shape && chans) || cin8 != 32 || tiles < 512 || a->tile_hint != 0) return 0;
    if ((double)a->batch * a->X * a->Y * a->Z * a->in_cs >= 4294967296.0) return 0;   // 32-bit staging offsets
    PersistP p;
    p.in = a->in; p.wpk = a->wpk; p.bias = a->bias; p.res1 = a->res1; p.res2 = a->res2; p.out = a->out;
    p.batch = a->batch; p.X = a->X; p.Y = a->Y; p.Z = a->Z; p.in_cs = a->in_cs; p.in_coff = a->in_coff;
    p.out_cs = a->out_cs; p.out_coff = a->out_coff;
    p.res1_cs = a->res1_cs; p.res1_coff = a->res1_coff; p.res2_cs = a->res2_cs; p.res2_coff = a->res2_coff;
    p.act_in = a->act_in; p.act_out = a->act_out; p.cout_store = a->cout_store;
    p.ytiles = (a->Y + kTY - 1) / kTY;
    p.ztiles = a->Z / kTZ;
    p.tiles_total = (int)tiles;
    const double pos = (double)a->batch * a->X * a->Y * a->Z;
    const double flops = 2.0 * pos * 27 * a->cin * a->cout;
    const double bytes = 4.0 * (pos * a->cin + pos * a->cout * (1 + (a->res1 != nullptr) + (a->res2 != nullptr)) +
                                27.0 * a->cin * a->cout);
    ProfScope prof("conv3d_c32p", stream, flops, bytes);
    DevState* ds = dev_state();
    if (ds == nullptr) return OCCD_ELAUNCH;
    int rc;
    if (tiled) {
        rc = d == 1 ? launch<1>(p, stream, ds) : d == 2 ? launch<2>(p, stream, ds) : launch<3>(p, stream, ds);
    } else {
        rc = d == 1 ? launch_slide<1>(p, stream, ds) : d == 2 ? launch_slide<2>(p, stream, ds)
                                                              : launch_slide<3>(p, stream, ds);
    }
    return rc == OCCD_OK ? 1 : rc;
}

}  // namespace occd
