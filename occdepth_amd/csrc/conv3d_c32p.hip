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
    int res_early;           // K2s3: residual rows requested in front of the FIRST 16-channel half of their plane
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

// ------------------------------------------------------------------------------------------------
// K2s3 -- K2s on the bf16 matrix pipe with the 3-way operand split.  Same work list, same sliding
// window (three accumulators, every staged plane feeds the three kx taps), same epilogue; what changes:
//   * instruction: v_mfma_f32_32x32x16_bf16, six per 16-channel K step -- (mid,mid), (hi,lo), (lo,hi), (hi,mid), (mid,hi),
//     (hi,hi), smallest first -- on x = hi + mid + lo, w = hi + mid + lo (three bf16 terms each, 24 significant bits):
//     float32-level accuracy at 6 x 32 = 192 matrix-pipe cycles per 16 channels instead of 8 x 64 = 512;
//   * the activation split is paid ONCE per staged element (while the slab is committed: fp32 rows in HBM, three bf16
//     planes in LDS: rows of 3 x 32 B + 16 B pad = 7 sixteen-byte slots, odd -> conflict-free ds_read_b128), not once per
//     tap as an on-the-fly split would;
//   * weights: the hi and mid images (2 x 55,296 B, exactly the 110,592 B K2s keeps) stay in LDS for the kernel's
//     lifetime; the lo image (one of the six products) does not fit beside the slab any more and is streamed from L2
//     two steps ahead (3 x 1 KB per wave and step, the same addresses for all 8 waves of a CU and all CUs);
//   * Z = 32: the z halo is not staged -- every y row of the slab carries ONE all-zero entry and the lanes whose
//     tap leaves the column read that (33 entries per row instead of 32 + 2 D: at D = 3 the slab is 51.7 KB, not 59.6);
//   * Z = 64, 96, ... (round 5, `ZH`: BASELINE configs[4] is 512 x 512 x 64): a workgroup owns a 32-column z tile and stages
//     its z halo like K2s does -- rows of 32 + 2 D entries, the neighbouring tile's data or zeros outside the volume --
//     38.1 KB at D = 1, 48.4 KB at D = 2; at D = 3 the 14 x 38 entries (59.6 KB) do not fit beside the two weight images
//     (53.1 KB left) and every cheaper row format conflicts on the 16-lane groups of ds_read_b128 (a 96-byte entry needs an
//     odd number of 16-byte slots), so the dilation-3 form works on y tiles of SIX rows (`TYV` = 6: 12 x 38 entries = 51.1 KB;
//     waves 6 and 7 stage and synchronise but own no output row -- 3/4 of the matrix-pipe rate, still well ahead of the
//     exact-fp32 K2s: 5.x against 7.8 ms per launch at 512 x 512 x 64).
// Work per launch 115.96 GFLOP algorithmic = 695.8 GFLOP issued on the bf16 pipe.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

constexpr int kX3RowB = 112;              // bytes per slab entry: hi | mid | lo of 16 channels + 16 B pad
constexpr int kX3ZW = 33;                 // entries per y row: z = 0 .. 31 and the zero entry
constexpr int kX3WImg = kWTaps * 2 * 64;  // u32x4 per split image of the weights: [tap][k16 2][lane 64]

// x = hi + mid + lo (round-to-nearest at every step; exact for every float32 whose low parts do not underflow)
__device__ __forceinline__ void split3_bf16(f32x4 a, f32x4 b, u32x4& hi, u32x4& mid, u32x4& lo) {
    bf16x8 h = {(__bf16)a.x, (__bf16)a.y, (__bf16)a.z, (__bf16)a.w, (__bf16)b.x, (__bf16)b.y, (__bf16)b.z, (__bf16)b.w};
    float r[8] = {a.x - (float)h[0], a.y - (float)h[1], a.z - (float)h[2], a.w - (float)h[3],
                  b.x - (float)h[4], b.y - (float)h[5], b.z - (float)h[6], b.w - (float)h[7]};
    bf16x8 m, l;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        m[j] = (__bf16)r[j];
        l[j] = (__bf16)(r[j] - (float)m[j]);
    }
    hi = __builtin_bit_cast(u32x4, h);
    mid = __builtin_bit_cast(u32x4, m);
    lo = __builtin_bit_cast(u32x4, l);
}

// NRES: residual operands compiled in (0: neither, 1: res1, 2: res1 and res2) -- their prefetch registers (16 per
// operand) are what the 5 of 7 head launches without residuals do not pay for.
template <int D, int NRES, bool ZH = false, int TYV = kTY>
__global__ void __launch_bounds__(512, 2) conv3d_c32_slide_x3_kernel(const SlideP sp) {
    static_assert(TYV >= 1 && TYV <= kTY, "output rows per y tile: one wave each");
    const PersistP& p = sp.base;
    constexpr int ZW = ZH ? kTZ + 2 * D : kX3ZW;  // entries per y row: the tile + its z halo (ZH), or the tile + the zero entry
    constexpr int ZST = ZH ? ZW : kTZ;            // entries per row that are (re)staged with every slab
    constexpr int YIN = TYV + 2 * D, ROWS = YIN * ZW;
    constexpr int NITEM = YIN * ZST * 2;          // staging items: (y row, z, 8-channel chunk of the 16-channel half)
    constexpr int NLOAD = (NITEM + 511) / 512;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    u32x4* const w4 = reinterpret_cast<u32x4*>(lds_raw);                 // hi image | mid image
    unsigned char* const slab = lds_raw + 2 * kX3WImg * 16;
    int* const mailbox = reinterpret_cast<int*>(slab + ROWS * kX3RowB);
    f32x4* const bias4 = reinterpret_cast<f32x4*>(slab + ROWS * kX3RowB + 16);
    // lo image: read through a buffer descriptor (voffset = lane * 16, soffset = a compile-time constant per fragment, so
    // the 54 fragment addresses cost no address VGPRs)
    const auto wlo_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<void*>(reinterpret_cast<const void*>(reinterpret_cast<const u32x4*>(p.wpk) + 2 * kX3WImg)), 0,
        kX3WImg * 16, 0x00020000);
    auto wlo_load = [&](int frag) {
        return __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(wlo_rsrc, (unsigned)(threadIdx.x & 63) * 16u, frag * 1024, 0));
    };

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, kk = lane >> 5;

    {   // hi | mid weight images: global -> LDS, once (6912 u32x4 = 13.5 x 512, 7 loads in flight per thread)
        static_assert(2 * kX3WImg == 13 * 512 + 256, "weight staging loop");
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            u32x4 wv[7];
#pragma unroll
            for (int u = 0; u < 7; ++u) {
                const int i = tid + (half * 7 + u) * 512;
                if (i < 2 * kX3WImg) wv[u] = reinterpret_cast<const u32x4*>(p.wpk)[i];
            }
#pragma unroll
            for (int u = 0; u < 7; ++u) {
                const int i = tid + (half * 7 + u) * 512;
                if (i < 2 * kX3WImg) w4[i] = wv[u];
            }
        }
        // the zero entry of every y row (never written again)
        if (!ZH && tid < YIN * 7) {
            const int yi = tid / 7, s16 = tid - yi * 7;
            *reinterpret_cast<u32x4*>(slab + (yi * kX3ZW + kTZ) * kX3RowB + s16 * 16) = u32x4{0u, 0u, 0u, 0u};
        }
    }

    // staging descriptors (constant over the kernel): slab byte offset of the item's hi chunk, y row, z column, channel chunk
    int sdst[NLOAD], syi[NLOAD], sz[NLOAD], sc8[NLOAD];
#pragma unroll
    for (int i = 0; i < NLOAD; ++i) {
        const int f = tid + i * 512;
        const bool live = f < NITEM;
        const int row = f >> 1, c8 = f & 1;
        const int yi = ZH ? row / ZST : row >> 5, z = ZH ? row - yi * ZST : row & 31;
        sdst[i] = live ? (yi * ZW + z) * kX3RowB + c8 * 16 : -1;
        syi[i] = yi;
        sz[i] = ZH ? z - D : z;                    // relative to the tile's first column
        sc8[i] = c8 * 8;
    }
    // A-fragment bases of this lane for kz = 0, 1, 2 (ky adds a constant): the voxel column li + (kz - 1) D, or the zero
    // entry when it leaves [0, 32)
    int abase[3];
#pragma unroll
    for (int kz = 0; kz < 3; ++kz) {
        const int z = li + (kz - 1) * D;
        abase[kz] = ((wave < TYV ? wave : 0) * ZW + (ZH ? z + D : (z >= 0 && z < kTZ) ? z : kTZ)) * kX3RowB + kk * 16;
    }
    int z0 = 0;                                    // first column of the segment's z tile (ZH)
    const size_t plane_stride = (size_t)p.Y * p.Z * p.in_cs;

    unsigned coloff[NLOAD];
    bool colok[NLOAD];
    f32x4 v[NLOAD][2];
    auto issue = [&](int xi, int h) {        // global -> registers; xi inside the volume
        const float* base = p.in + (size_t)xi * plane_stride + h * 16;
#pragma unroll
        for (int i = 0; i < NLOAD; ++i) {
            v[i][0] = f32x4{0.f, 0.f, 0.f, 0.f};
            v[i][1] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (colok[i]) {
                v[i][0] = *(const f32x4*)(base + coloff[i]);
                v[i][1] = *(const f32x4*)(base + coloff[i] + 4);
            }
        }
    };
    auto commit = [&]() {                    // registers -> split -> the three bf16 planes of the slab
#pragma unroll
        for (int i = 0; i < NLOAD; ++i)
            if (sdst[i] >= 0) {
                f32x4 a = v[i][0], b = v[i][1];
                if (p.act_in == OCCD_ACT_RELU) {
                    a.x = fmaxf(a.x, 0.f); a.y = fmaxf(a.y, 0.f); a.z = fmaxf(a.z, 0.f); a.w = fmaxf(a.w, 0.f);
                    b.x = fmaxf(b.x, 0.f); b.y = fmaxf(b.y, 0.f); b.z = fmaxf(b.z, 0.f); b.w = fmaxf(b.w, 0.f);
                }
                u32x4 hi, mid, lo;
                split3_bf16(a, b, hi, mid, lo);
                *reinterpret_cast<u32x4*>(slab + sdst[i]) = hi;
                *reinterpret_cast<u32x4*>(slab + sdst[i] + 32) = mid;
                *reinterpret_cast<u32x4*>(slab + sdst[i] + 64) = lo;
            }
    };

    f32x16 acc0, acc1, acc2;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc0[r] = acc1[r] = acc2[r] = 0.f;
    int mid_img = kX3WImg;                       // (opaque to the optimiser: folded into one base, the mid image's offsets
    asm volatile("" : "+v"(mid_img));            //  exceed the 16-bit ds_read immediate and every fragment costs an address VGPR)

    // lo-weight ring: [step parity][kx], two steps ahead of the MFMAs (a younger load must not be waited for while the
    // next slab's staging loads -- issued at the top of the slab -- are still in flight: vmcnt retires in order)
    u32x4 ring[2][3];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) ring[t][kx] = wlo_load((kx * 9 + t) * 2 + 0);

#define OCCD_X3_MFMA(ACC, W, A) \
    ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, W), __builtin_bit_cast(bf16x8, A), ACC, 0, 0, 0)

    // one staged slab (16-channel half H of a plane) into the accumulators whose output plane exists.  Sub-step =
    // (tap t = (ky, kz), kx): 6 MFMAs into acc[kx]; the hi / mid weight fragments of the next sub-step and the three
    // activation fragments of the next step are read from LDS above the MFMAs of the current one.
    auto mma = [&](auto hsel, auto u0, auto u1, auto u2) {
        constexpr int H = decltype(hsel)::value;
        constexpr bool U0 = decltype(u0)::value, U1 = decltype(u1)::value, U2 = decltype(u2)::value;
        constexpr int NA = (U0 ? 1 : 0) + (U1 ? 1 : 0) + (U2 ? 1 : 0);
        constexpr int KXS[3] = {U0 ? 0 : (U1 ? 1 : 2), (U0 && U1) ? 1 : 2, 2};
        constexpr int NSUB = 9 * NA;
        const u32x4* const whi = w4 + H * 64 + lane;
        const u32x4* const wmid = whi + mid_img;   // its own base register: the ds_read offsets of both images stay below 64 KB
        auto aptr = [&](int t, int term) {
            const int ky = t / 3, kz = t - 3 * ky;
            return reinterpret_cast<const u32x4*>(slab + abase[kz] + ky * D * ZW * kX3RowB + term * 32);
        };
        u32x4 an0 = *aptr(0, 0), an1 = *aptr(0, 1), an2 = *aptr(0, 2);
        u32x4 bhn = whi[(KXS[0] * 9 * 2) * 64], bmn = wmid[(KXS[0] * 9 * 2) * 64];
        u32x4 a0 = an0, a1 = an1, a2 = an2;
#pragma unroll
        for (int i = 0; i < NSUB; ++i) {
            const int t = i / NA, kx = KXS[i % NA];
            const int par = (H * 9 + t) & 1;
            if (i % NA == 0) {
                a0 = an0; a1 = an1; a2 = an2;
                if (t < 8) { an0 = *aptr(t + 1, 0); an1 = *aptr(t + 1, 1); an2 = *aptr(t + 1, 2); }
            }
            const u32x4 bh = bhn, bm = bmn;
            if (i + 1 < NSUB) {
                const int tn = (i + 1) / NA, kxn = KXS[(i + 1) % NA];
                bhn = whi[((kxn * 9 + tn) * 2) * 64];
                bmn = wmid[((kxn * 9 + tn) * 2) * 64];
            }
            const u32x4 bl = ring[par][kx];
            __builtin_amdgcn_sched_barrier(0);
            if (kx == 0) {
                OCCD_X3_MFMA(acc0, bm, a1); OCCD_X3_MFMA(acc0, bh, a2); OCCD_X3_MFMA(acc0, bl, a0);
                OCCD_X3_MFMA(acc0, bh, a1); OCCD_X3_MFMA(acc0, bm, a0); OCCD_X3_MFMA(acc0, bh, a0);
            } else if (kx == 1) {
                OCCD_X3_MFMA(acc1, bm, a1); OCCD_X3_MFMA(acc1, bh, a2); OCCD_X3_MFMA(acc1, bl, a0);
                OCCD_X3_MFMA(acc1, bh, a1); OCCD_X3_MFMA(acc1, bm, a0); OCCD_X3_MFMA(acc1, bh, a0);
            } else {
                OCCD_X3_MFMA(acc2, bm, a1); OCCD_X3_MFMA(acc2, bh, a2); OCCD_X3_MFMA(acc2, bl, a0);
                OCCD_X3_MFMA(acc2, bh, a1); OCCD_X3_MFMA(acc2, bm, a0); OCCD_X3_MFMA(acc2, bh, a0);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (i % NA == NA - 1) {
                // step t done: refill its ring parity with step t + 2 (of this slab, or t + 2 - 9 of the next one, whose
                // half is always the other one) for ALL kx -- the next slab may feed accumulators this one skipped
                const int t2 = t + 2 < 9 ? t + 2 : t + 2 - 9;
                const int h2 = t + 2 < 9 ? H : 1 - H;
#pragma unroll
                for (int k2 = 0; k2 < 3; ++k2) ring[par][k2] = wlo_load((k2 * 9 + t2) * 2 + h2);
            }
        }
    };
    using T_ = std::true_type;
    using F_ = std::false_type;
    using H0 = std::integral_constant<int, 0>;
    using H1 = std::integral_constant<int, 1>;

    f32x4 r1[4], r2[4];
    if (tid < 8) {
        f32x4 bv = {0.f, 0.f, 0.f, 0.f};
        if (p.bias != nullptr && 4 * tid < p.cout_store) bv = *(const f32x4*)(p.bias + 4 * tid);
        bias4[tid] = bv;
    }
    auto res_fetch = [&](int b, int yt, int x) {
        if (NRES == 0) return;
        if (TYV < kTY && wave >= TYV) return;
        const int y = min(yt * TYV + wave, p.Y - 1);
        const size_t vox = ((size_t)(b * p.X + x) * p.Y + y) * p.Z + z0 + li;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int c = 8 * g + 4 * kk;
            if (c < p.cout_store) {
                if (NRES >= 1) r1[g] = *(const f32x4*)(p.res1 + vox * p.res1_cs + p.res1_coff + c);
                if (NRES >= 2) r2[g] = *(const f32x4*)(p.res2 + vox * p.res2_cs + p.res2_coff + c);
            }
        }
    };
    auto store2 = [&](int b, int yt, int x) {
        const int y = yt * TYV + wave;
        if (y < p.Y && (TYV == kTY || wave < TYV)) {
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
                    if (NRES >= 1) o += r1[g];
                    if (NRES >= 2) o += r2[g];
                    if (p.act_out == OCCD_ACT_RELU) {
                        o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f);
                    }
                    *(f32x4*)(p.out + vox * p.out_cs + p.out_coff + c) = o;
                }
            }
        }
    };
    auto slab_mma = [&](auto hsel, bool u0, bool u1, bool u2) {
        if (TYV < kTY && wave >= TYV) return;                 // (a wave without an output row: stages and synchronises only)
        if (u0 && u1 && u2) mma(hsel, T_{}, T_{}, T_{});      // interior plane
        else if (u0 && u1) mma(hsel, T_{}, T_{}, F_{});       // second plane of a run
        else if (u1 && u2) mma(hsel, F_{}, T_{}, T_{});       // second to last
        else if (u0) mma(hsel, T_{}, F_{}, F_{});             // first
        else if (u2) mma(hsel, F_{}, F_{}, T_{});             // last
        else mma(hsel, F_{}, T_{}, F_{});                     // run of a single output plane
    };

    while (true) {
        __syncthreads();                                   // mailbox / slab free, weights visible
        if (tid == 0) mailbox[0] = atomicAdd(sp.counter, 1);
        __syncthreads();
        const int seg = mailbox[0];
        if (seg >= sp.total_segs) {
            if (tid == 0 && atomicAdd(sp.counter + 1, 1) == (int)gridDim.x - 1) {
                sp.counter[1] = 0;
                __threadfence();
                atomicExch(sp.counter, 0);
            }
            break;
        }
        // segments are ordered (b, range, ytile, ztile) with the z tile fastest (ztiles == 1 unless ZH)
        const int tz = ZH ? seg % (p.ytiles * p.ztiles) : seg % p.ytiles;
        const int zt = ZH ? tz % p.ztiles : 0, yt = ZH ? tz / p.ztiles : tz;
        const int rest = ZH ? seg / (p.ytiles * p.ztiles) : seg / p.ytiles;
        const int b = rest / sp.segs_per_col;
        const int q0 = (rest - b * sp.segs_per_col) * sp.seg_len;
        const int q1 = min(q0 + sp.seg_len, p.X);
        z0 = zt * kTZ;
#pragma unroll
        for (int i = 0; i < NLOAD; ++i) {
            const int y = yt * TYV - D + syi[i];
            const int z = z0 + sz[i];
            colok[i] = sdst[i] >= 0 && y >= 0 && y < p.Y && (!ZH || (z >= 0 && z < p.Z));
            coloff[i] = (unsigned)((((size_t)b * p.X * p.Y + (colok[i] ? y : 0)) * p.Z + (colok[i] ? z : 0)) * p.in_cs +
                                   p.in_coff + sc8[i]);
        }
        int q = q0;
        while (q < q1) {
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
            const int nj = cnt + 2;
            const int jfirst = x0 - D < 0 ? 1 : 0;
            const int jlast = x0 + cnt * D >= p.X ? (p.X - 1 - x0) / D + 1 : nj - 1;
            __syncthreads();                                               // previous run's slab consumed
            issue(x0 + (jfirst - 1) * D, 0);
            for (int j = 0; j < nj; ++j) {
                const int xi = x0 + (j - 1) * D;
                const bool u0 = j < cnt, u1 = j >= 1 && j <= cnt, u2 = j >= 2;
                if (j >= jfirst && j <= jlast) {
                    __syncthreads();                                       // previous slab consumed
                    commit();
                    __syncthreads();
                    issue(xi, 1);
                    // the residual rows of out[j-2] (it completes with this plane): requested a whole half-slab before their
                    // use (round 5, `res_early`): issued in front of the second half they sit in the in-order vmcnt queue
                    // right before the lo-weight stream the MFMAs wait for two steps later (0.53 / 0.63 ms per launch with
                    // one / two residual operands against 0.45 - 0.51 without)
                    if (NRES > 0 && sp.res_early && u2) res_fetch(b, yt, xi - D);
                    slab_mma(H0{}, u0, u1, u2);
                    __syncthreads();
                    commit();
                    __syncthreads();
                    if (j < jlast) issue(xi + D, 0);
                    if (NRES > 0 && !sp.res_early && u2) res_fetch(b, yt, xi - D);
                    slab_mma(H1{}, u0, u1, u2);
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
#undef OCCD_X3_MFMA
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
    bool slide_x3_attr[4][3][3] = {};
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

// Work list of the sliding-window kernels: ranges per column = k rounds over the grid; a range of L planes costs L + 2
// stagings per run (D > 1: up to two runs).  Few long ranges amortise the two extra planes, but the list must fill
// whole rounds.  Also hands out the launch's pair of self re-arming counters (a ring: launches in flight on different
// streams never share one).
template <int D>
int plan_slide(const PersistP& base, DevState* ds, SlideP* sp) {
    constexpr int kSlots = 256;
    const int num_cu = ds->num_cu;
    {
        std::lock_guard<std::mutex> lock(ds->mu);
        if (ds->counter == nullptr) {
            if (hipMalloc(&ds->counter, kSlots * 2 * sizeof(int)) != hipSuccess) return OCCD_ELAUNCH;
            if (hipMemset(ds->counter, 0, kSlots * 2 * sizeof(int)) != hipSuccess) return OCCD_ELAUNCH;
        }
    }
    sp->base = base;
    sp->counter = ds->counter + 2 * (g_slot.fetch_add(1) % kSlots);
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
    sp->seg_len = (base.X + best_s - 1) / best_s;
    sp->segs_per_col = (base.X + sp->seg_len - 1) / sp->seg_len;
    sp->total_segs = cols * sp->segs_per_col;
    return OCCD_OK;
}

template <int D>
int launch_slide(const PersistP& base, hipStream_t st, DevState* ds) {
    constexpr int ROWS = (kTY + 2 * D) * (kTZ + 2 * D), RS4 = 16 / 4 + 1;
    const size_t lds = (size_t)kWFloat4 * 16 + (size_t)ROWS * RS4 * 16 + 16 + 128;
    {
        std::lock_guard<std::mutex> lock(ds->mu);
        if (!ds->slide_attr[D]) {
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(conv3d_c32_slide_kernel<D>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
                return OCCD_ELAUNCH;
            ds->slide_attr[D] = true;
        }
    }
    SlideP sp;
    const int rc = plan_slide<D>(base, ds, &sp);
    if (rc != OCCD_OK) return rc;
    const int grid = ds->num_cu < sp.total_segs ? ds->num_cu : sp.total_segs;
    hipLaunchKernelGGL(conv3d_c32_slide_kernel<D>, dim3((unsigned)grid), dim3(512), lds, st, sp);
    return occd::check_launch();
}

template <int D, int NRES, bool ZH, int TYV = kTY>
int launch_slide_x3(const PersistP& base0, hipStream_t st, DevState* ds) {
    PersistP base = base0;
    if (TYV != kTY) {                                     // the work list is cut in y tiles of TYV rows
        base.ytiles = (base.Y + TYV - 1) / TYV;
        base.tiles_total = base.batch * base.X * base.ytiles * base.ztiles;
    }
    constexpr int ROWS = (TYV + 2 * D) * (ZH ? kTZ + 2 * D : kX3ZW);
    constexpr size_t lds = (size_t)2 * kX3WImg * 16 + (size_t)ROWS * kX3RowB + 16 + 128;
    static_assert(lds <= 160 * 1024, "K2s3 LDS budget (Z > 32: D <= 2 only)");
    {
        std::lock_guard<std::mutex> lock(ds->mu);
        constexpr int VAR = ZH ? (TYV == kTY ? 1 : 2) : 0;
        if (!ds->slide_x3_attr[D][NRES][VAR]) {
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(conv3d_c32_slide_x3_kernel<D, NRES, ZH, TYV>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
                return OCCD_ELAUNCH;
            ds->slide_x3_attr[D][NRES][VAR] = true;
        }
    }
    SlideP sp;
    const int rc = plan_slide<D>(base, ds, &sp);
    if (rc != OCCD_OK) return rc;
    static const bool res_early = occd::env_flag("OCCD_C32X3_RES_EARLY", true);      // A/B switch (0: in front of the second half)
    sp.res_early = res_early ? 1 : 0;
    const int grid = ds->num_cu < sp.total_segs ? ds->num_cu : sp.total_segs;
    hipLaunchKernelGGL((conv3d_c32_slide_x3_kernel<D, NRES, ZH, TYV>), dim3((unsigned)grid), dim3(512), lds, st, sp);
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

namespace {
// Geometry shared by K2p / K2s / K2s3: 3x3x3, stride 1, dilation d = padding in {1, 2, 3}, <= 32 -> <= 32 channels on rows
// that hold at least 32 input channels, output grid == input grid, no scatter.
bool c32_geometry(const occd_conv3d_args* a, PersistP* p, double* flops, double* bytes) {
    const int d = a->dx;
    const bool geom = a->kx == 3 && a->ky == 3 && a->kz == 3 && a->sx == 1 && a->sy == 1 && a->sz == 1 &&
                      a->dy == d && a->dz == d && d >= 1 && d <= 3 && a->px == d && a->py == d && a->pz == d;
    const bool shape = a->Z % kTZ == 0 && a->Xo == a->X && a->Yo == a->Y && a->Zo == a->Z && a->OX == a->X &&
                       a->OY == a->Y && a->OZ == a->Z && a->o_stride_x == 1 && a->o_stride_y == 1 &&
                       a->o_stride_z == 1 && a->o_off_x == 0 && a->o_off_y == 0 && a->o_off_z == 0;
    const int cin8 = (a->cin + 7) & ~7;
    const bool chans = cin8 <= 32 && a->cout <= 32 && a->in_coff + 32 <= a->in_cs && a->act_in != OCCD_ACT_SIGMOID;
    // enough tiles to keep every CU busy for several rounds, otherwise the generic kernel tiles finer
    const long tiles = (long)a->batch * a->X * ((a->Y + kTY - 1) / kTY) * (a->Z / kTZ);
    if (!(geom && shape && chans) || cin8 != 32 || tiles < 512 || a->tile_hint != 0) return false;
    if ((double)a->batch * a->X * a->Y * a->Z * a->in_cs >= 4294967296.0) return false;   // 32-bit staging offsets
    p->in = a->in; p->wpk = a->wpk; p->bias = a->bias; p->res1 = a->res1; p->res2 = a->res2; p->out = a->out;
    p->batch = a->batch; p->X = a->X; p->Y = a->Y; p->Z = a->Z; p->in_cs = a->in_cs; p->in_coff = a->in_coff;
    p->out_cs = a->out_cs; p->out_coff = a->out_coff;
    p->res1_cs = a->res1_cs; p->res1_coff = a->res1_coff; p->res2_cs = a->res2_cs; p->res2_coff = a->res2_coff;
    p->act_in = a->act_in; p->act_out = a->act_out; p->cout_store = a->cout_store;
    p->ytiles = (a->Y + kTY - 1) / kTY;
    p->ztiles = a->Z / kTZ;
    p->tiles_total = (int)tiles;
    const double pos = (double)a->batch * a->X * a->Y * a->Z;
    *flops = 2.0 * pos * 27 * a->cin * a->cout;
    *bytes = 4.0 * (pos * a->cin + pos * a->cout * (1 + (a->res1 != nullptr) + (a->res2 != nullptr)) + 27.0 * a->cin * a->cout);
    return true;
}
}  // namespace

namespace occd {

// Returns 1 when the launch was taken by the persistent kernel, 0 when the geometry does not qualify,
// <0 on error.
int try_conv3d_c32_persist(const occd_conv3d_args* a, hipStream_t stream) {
    static const bool tiled = env_flag("OCCD_C32P_TILED", false);   // A/B switch: per-tile variant (K2p) vs sliding window (K2s)
    PersistP p;
    double flops, bytes;
    if (!c32_geometry(a, &p, &flops, &bytes) || (tiled && a->Z != kTZ)) return 0;
    const int d = a->dx;
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

// K2s3: the same launches on the bf16 matrix pipe with the 3-way split (a->wpk = the hi | mid | lo image of
// occd_pack_weights_bf16x3, float32 tensors).  Same return convention; Z == 32, or Z = 64, 96, ... through the z-halo form
// (see the kernel header; dilation 3 of such volumes on six-row y tiles);
// OCCD_C32X3_SLIDE=0 leaves every split launch to the generic K2b skeleton (A/B).
int try_conv3d_c32_slide_x3(const occd_conv3d_args* a, hipStream_t stream) {
    static const bool off = !env_flag("OCCD_C32X3_SLIDE", true);
    PersistP p;
    double flops, bytes;
    if (off || !c32_geometry(a, &p, &flops, &bytes)) return 0;
    if ((a->in_cs & 3) || (a->in_coff & 3)) return 0;
    const int d = a->dx;
    ProfScope prof("conv3d_c32x3", stream, flops, bytes);
    DevState* ds = dev_state();
    if (ds == nullptr) return OCCD_ELAUNCH;
    if (p.res1 == nullptr && p.res2 != nullptr) {   // the kernel's single-residual form reads res1
        p.res1 = p.res2; p.res1_cs = p.res2_cs; p.res1_coff = p.res2_coff; p.res2 = nullptr;
    }
    const int nres = (p.res1 != nullptr) + (p.res2 != nullptr);
    int rc;
#define OCCD_X3_LAUNCH(DD, ZZ) \
    (nres == 0 ? launch_slide_x3<DD, 0, ZZ>(p, stream, ds) : nres == 1 ? launch_slide_x3<DD, 1, ZZ>(p, stream, ds) : launch_slide_x3<DD, 2, ZZ>(p, stream, ds))
    if (a->Z == kTZ) rc = d == 1 ? OCCD_X3_LAUNCH(1, false) : d == 2 ? OCCD_X3_LAUNCH(2, false) : OCCD_X3_LAUNCH(3, false);
    else if (d < 3) rc = d == 1 ? OCCD_X3_LAUNCH(1, true) : OCCD_X3_LAUNCH(2, true);
    else rc = nres == 0 ? launch_slide_x3<3, 0, true, 6>(p, stream, ds) : nres == 1 ? launch_slide_x3<3, 1, true, 6>(p, stream, ds)
                                                                                    : launch_slide_x3<3, 2, true, 6>(p, stream, ds);
#undef OCCD_X3_LAUNCH
    return rc == OCCD_OK ? 1 : rc;
}

}  // namespace occd
