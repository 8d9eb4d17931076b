// K14 -- one DDR Bottleneck3D (stride 1) in two launches instead of five.
//
//   o1 = relu(W1 x + b1)                                  1x1x1   C -> P          (BatchNorm folded into W / b)
//   o2 = conv_z(o1) + b2                                  (1,1,3) dilation d0
//   o3 = conv_y(relu(o2)) + b3 + o2                       (1,3,1) dilation d1
//   o4 = conv_x(relu(o3)) + b4 + o2 + o3                  (3,1,1) dilation d2
//   y  = relu(W5 relu(o4) + b5 + x)                       1x1x1   P -> C
// Reference: occdepth/models/DDR.py:111-139 (stride == 1: the pooled side branches are identities).
//
// The five-launch form (fused.ConvPlan x 5 on K2) moves x, three P-channel intermediates and y through HBM with ~110
// small launches per frame that each run at about a third of the HBM rate; the block is HBM work (C = 4 P, 8.7 kFLOP per
// voxel at P = 16).  Here:
//   bneck_a: NT consecutive voxels (whole Z columns) per workgroup, 16 voxels per MFMA column block: conv1 straight from
//            the global x rows (one 16-byte load per lane and 16 channels, 64 contiguous bytes per voxel), o1 to LDS,
//            conv_z out of LDS, o2 written once (P floats per voxel).
//   bneck_b: a TX x TY x Z tile: o3 on the tile plus its +-d2 planes along X straight from the o2 rows (L2 hits for the
//            halo), o3 and o2 + o3 in LDS, conv_x out of LDS, conv5 from the accumulator registers, residual x and y as
//            16-byte accesses per lane.  No staging tile, no transposition pass.
// Round 3 first tried one THREAD per voxel on the vector ALU with scalar weight operands: 157.6 us per block at
// 128x128x16 against 187.1 us for the five launches in isolation and SLOWER in the replayed frame (one wave per workgroup at
// P = 32 / 64); the matrix pipe form below replaced it.
// HBM bytes per voxel: read x (4C) + write o2 (4P) | read o2 (4P, +halo from L2) + read x (4C) + write y (4C)
//   = 4 (3C + 2P) = 896 B at C = 64 (235 MB per block at 128x128x16) against 4 (3C + 8P) + launch tails before.
#include "common.h"
#include <algorithm>

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

struct BneckP {
    const float* x;
    float* y;
    float* o2;
    const float* w1; const float* b1;     // fragments of W1^T [C][P], [P]
    const float* w2; const float* b2;     // 3 x fragments of [P][P] (tap; in, out), [P]
    const float* w3; const float* b3;
    const float* w4; const float* b4;
    const float* w5; const float* b5;     // fragments of W5^T [P][C], [C]
    int batch, X, Y, Z, C;
    int x_cs, x_coff, y_cs, y_coff;
    int d0, d1, d2;
    int TX, TY, xtiles, ytiles;
    long ncols;                           // batch * X * Y
};

__device__ __forceinline__ f32x4 relu4(f32x4 v) {
    return f32x4{fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f)};
}

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// Every reduction runs as  D^T (couts x voxels) = W (couts x cin) . X^T (cin x voxels)  on v_mfma_f32_16x16x4_f32:
//   A operand = weights: lane l supplies W[cout 16 m + (l & 15)][k], k = l >> 4;
//   B operand = data:    lane l supplies X[voxel l & 15][k];
//   D: lane l holds couts 16 m + 4 (l >> 4) + {0..3} of voxel l & 15  -> ONE float4 of consecutive channels per lane.
// The K order of a 16-channel "super-step" t is permuted so that a lane's four k values are the four CONSECUTIVE channels
// 16 t + 4 (l >> 4) + e, e = 0..3: data arrives as one 16-byte load per lane (global rows or LDS rows, 64 contiguous bytes
// per voxel across its 4 lanes), and a D fragment IS the B operand of the next 1x1 reduction (conv5 reads relu(o4) from
// registers).  Weight fragments live in LDS in that order: [t][m][lane][e].
template <int M>
__device__ __forceinline__ void mma_step(f32x4 (&acc)[M], const float* wfrag, int lane, f32x4 data) {
#pragma unroll
    for (int m = 0; m < M; ++m) {
        const f32x4 w = *(const f32x4*)(wfrag + (m * 64 + lane) * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[m] = mfma16(w[e], data[e], acc[m]);
    }
}

// The packed weight buffer holds every matrix ALREADY in fragment order ([t][m][lane][e], element = W[cin 16 t + 4 (lane >> 4)
// + e][cout 16 m + (lane & 15)]; models/DDR.py builds it with one permute per matrix when the weights change), so filling
// LDS is a linear, coalesced 16-byte copy.  (The first version gathered the fragments from [cin][cout] matrices with 4-byte
// loads: 80 dependent loads per thread at 32 planes, 72 us for a 64x64x8 block whose HBM time is 12 us.)
__device__ __forceinline__ void copy_frags(float* dst, const float* __restrict__ src, int n, int tid, int nthreads) {
    for (int i = tid * 4; i < n; i += nthreads * 4) *(f32x4*)(dst + i) = *(const f32x4*)(src + i);
}

// ---------------------------------------------------------------- A: x -> o1 -> o2      (16 % Z == 0: a tile = whole columns)
template <int P, int NT>
__global__ void __launch_bounds__(NT) bneck_a_kernel(const BneckP p) {
    constexpr int M = P / 16, NW = NT / 64, TPW = 4;     // cout tiles, waves, 16-voxel tiles per wave
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int CT = p.C >> 4;                              // super-steps of conv1
    float* w1f = lds;                                     // [CT][M][64][4]
    float* w2f = w1f + p.C * P;                           // [3][M(t)][M][64][4]
    float* o1s = w2f + 3 * P * P;                         // [NT][P]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 15, g = lane >> 4;
    copy_frags(w1f, p.w1, p.C * P, tid, NT);
    copy_frags(w2f, p.w2, 3 * P * P, tid, NT);
    __syncthreads();
    const long v0 = (long)blockIdx.x * NT;
    const long vtot = p.ncols * p.Z;
    f32x4 b1v[M], b2v[M];
#pragma unroll
    for (int m = 0; m < M; ++m) {
        b1v[m] = *(const f32x4*)(p.b1 + 16 * m + 4 * g);
        b2v[m] = *(const f32x4*)(p.b2 + 16 * m + 4 * g);
    }
    for (int it = 0; it < TPW; ++it) {
        const int row = (wave * TPW + it) * 16 + j;       // voxel of this lane inside the workgroup tile
        const long v = v0 + row;
        const bool live = v < vtot;
        const float* xr = p.x + (size_t)(live ? v : 0) * p.x_cs + p.x_coff + 4 * g;
        f32x4 acc[M];
#pragma unroll
        for (int m = 0; m < M; ++m) acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int t0 = 0; t0 < CT; t0 += 4) {              // 4 super-steps of loads in flight
            f32x4 d[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) d[u] = t0 + u < CT ? *(const f32x4*)(xr + 16 * (t0 + u)) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (t0 + u < CT) mma_step<M>(acc, w1f + (size_t)(t0 + u) * M * 256, lane, d[u]);
        }
#pragma unroll
        for (int m = 0; m < M; ++m) *(f32x4*)(o1s + row * P + 16 * m + 4 * g) = relu4(acc[m] + b1v[m]);
        __syncthreads();                                  // (uniform trip count; the tile's columns are complete)
        const int z = row % p.Z;
        f32x4 o2[M];
#pragma unroll
        for (int m = 0; m < M; ++m) o2[m] = b2v[m];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int zz = z + (k - 1) * p.d0;
            const bool ok = zz >= 0 && zz < p.Z;
            const float* src = o1s + (ok ? row + (k - 1) * p.d0 : row) * P + 4 * g;
#pragma unroll
            for (int t = 0; t < M; ++t) {
                f32x4 d = *(const f32x4*)(src + 16 * t);
                if (!ok) d = f32x4{0.f, 0.f, 0.f, 0.f};
                mma_step<M>(o2, w2f + (size_t)(k * M + t) * M * 256, lane, d);
            }
        }
        if (live) {
#pragma unroll
            for (int m = 0; m < M; ++m) *(f32x4*)(p.o2 + (size_t)v * P + 16 * m + 4 * g) = o2[m];
        }
    }
}

// ---------------------------------------------------------------- B: o2 -> o3 -> o4 -> y    (TY * Z a multiple of 16)
template <int P, int NT>
__global__ void __launch_bounds__(NT) bneck_b_kernel(const BneckP p) {
    constexpr int M = P / 16, NW = NT / 64;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 15, g = lane >> 4;
    const int TYZ = p.TY * p.Z;
    const int nvox = p.TX * TYZ;                           // == NT
    const int XR = p.TX + 2 * p.d2;
    const int CT = p.C >> 4;
    float* w3f = lds;                                      // [3][M][M][64][4]
    float* w4f = w3f + 3 * P * P;
    float* w5f = w4f + 3 * P * P;                          // [M(t)][CT][64][4]
    float* o3s = w5f + P * p.C;                            // [XR * TYZ][P]   raw o3 (zero outside the volume)
    float* cs = o3s + (size_t)XR * TYZ * P;                // [nvox][P]       o2 + o3
    copy_frags(w3f, p.w3, 3 * P * P, tid, NT);
    copy_frags(w4f, p.w4, 3 * P * P, tid, NT);
    copy_frags(w5f, p.w5, P * p.C, tid, NT);
    __syncthreads();
    const int b = blockIdx.y;
    const int xt = blockIdx.x / p.ytiles, yt = blockIdx.x - xt * p.ytiles;
    const int x0 = xt * p.TX, y0 = yt * p.TY;
    const float* o2b = p.o2 + (size_t)b * p.X * p.Y * p.Z * P;
    f32x4 b3v[M], b4v[M];
#pragma unroll
    for (int m = 0; m < M; ++m) {
        b3v[m] = *(const f32x4*)(p.b3 + 16 * m + 4 * g);
        b4v[m] = *(const f32x4*)(p.b4 + 16 * m + 4 * g);
    }

    // ---- o3 on the tile and its X halo, 16 region voxels per MFMA column block
    for (int rt = wave; rt < XR * TYZ / 16; rt += NW) {
        const int r = rt * 16 + j;
        const int xr = r / TYZ, rem = r - xr * TYZ;
        const int ty = rem / p.Z, z = rem - ty * p.Z;
        const int xx = x0 - p.d2 + xr, yy = y0 + ty;
        const bool in = xx >= 0 && xx < p.X && yy < p.Y;
        f32x4 d[3][M];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int yk = yy + (k - 1) * p.d1;
            const bool ok = in && yk >= 0 && yk < p.Y;
            const float* src = o2b + (((size_t)(in ? xx : 0) * p.Y + (ok ? yk : 0)) * p.Z + z) * P + 4 * g;
#pragma unroll
            for (int t = 0; t < M; ++t) {
                d[k][t] = *(const f32x4*)(src + 16 * t);
                if (!ok) d[k][t] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
        f32x4 o3[M];
#pragma unroll
        for (int m = 0; m < M; ++m) o3[m] = b3v[m] + d[1][m];        // + o2 (the centre rows ARE the D-layout residual)
#pragma unroll
        for (int k = 0; k < 3; ++k)
#pragma unroll
            for (int t = 0; t < M; ++t) mma_step<M>(o3, w3f + (size_t)(k * M + t) * M * 256, lane, relu4(d[k][t]));
        const bool centre = xr >= p.d2 && xr < p.d2 + p.TX;
#pragma unroll
        for (int m = 0; m < M; ++m) {
            *(f32x4*)(o3s + (size_t)r * P + 16 * m + 4 * g) = in ? o3[m] : f32x4{0.f, 0.f, 0.f, 0.f};
            if (centre) *(f32x4*)(cs + (size_t)(r - p.d2 * TYZ) * P + 16 * m + 4 * g) = d[1][m] + o3[m];
        }
    }
    __syncthreads();

    // ---- o4 = conv_x(relu(o3)) + b4 + o2 + o3; y = relu(W5 relu(o4) + b5 + x) straight from the D fragments
    for (int ct = wave; ct < nvox / 16; ct += NW) {
        const int tv = ct * 16 + j;
        const int tx = tv / TYZ, rem = tv - tx * TYZ;
        const int ty = rem / p.Z, z = rem - ty * p.Z;
        const int xx = x0 + tx, yy = y0 + ty;
        const bool live = xx < p.X && yy < p.Y;
        const size_t vrow = (((size_t)b * p.X + (live ? xx : 0)) * p.Y + (live ? yy : 0)) * p.Z + z;
        const float* xres = p.x + vrow * p.x_cs + p.x_coff + 4 * g;
        f32x4 o4[M];
#pragma unroll
        for (int m = 0; m < M; ++m) o4[m] = b4v[m] + *(const f32x4*)(cs + (size_t)tv * P + 16 * m + 4 * g);
#pragma unroll
        for (int k = 0; k < 3; ++k)
#pragma unroll
            for (int t = 0; t < M; ++t)
                mma_step<M>(o4, w4f + (size_t)(k * M + t) * M * 256, lane,
                            relu4(*(const f32x4*)(o3s + (size_t)(tv + k * p.d2 * TYZ) * P + 16 * t + 4 * g)));
#pragma unroll
        for (int m = 0; m < M; ++m) o4[m] = relu4(o4[m]);
        float* yrow = p.y + vrow * p.y_cs + p.y_coff + 4 * g;
        for (int mo = 0; mo < CT; mo += 4) {               // 4 output tiles (64 channels) per round, residual loads first
            f32x4 res[4], bias[4], acc[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const bool on = mo + u < CT;
                res[u] = on ? *(const f32x4*)(xres + 16 * (mo + u)) : f32x4{0.f, 0.f, 0.f, 0.f};
                bias[u] = on ? *(const f32x4*)(p.b5 + 16 * (mo + u) + 4 * g) : f32x4{0.f, 0.f, 0.f, 0.f};
                acc[u] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (mo + u >= CT) continue;
#pragma unroll
                for (int t = 0; t < M; ++t) {
                    const f32x4 w = *(const f32x4*)(w5f + ((size_t)(t * CT + mo + u) * 64 + lane) * 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[u] = mfma16(w[e], o4[t][e], acc[u]);
                }
            }
            if (live) {
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (mo + u < CT) *(f32x4*)(yrow + 16 * (mo + u)) = relu4(acc[u] + bias[u] + res[u]);
            }
        }
    }
}

template <int P, int NT>
int launch(const BneckP& p, hipStream_t st) {
    const size_t lds_a = sizeof(float) * ((size_t)p.C * P + 3 * P * P + (size_t)NT * P);
    const long vtot = p.ncols * p.Z;
    const long blocks_a = (vtot + NT - 1) / NT;
    const int TYZ = p.TY * p.Z;
    const size_t lds_b = sizeof(float) * ((size_t)6 * P * P + (size_t)P * p.C + (size_t)(p.TX + 2 * p.d2) * TYZ * P + (size_t)NT * P);
    if (lds_a > 160 * 1024 || lds_b > 160 * 1024) return OCCD_ENOMEM;
    if (lds_a > 64 * 1024 || lds_b > 64 * 1024) {
        if (occd::ensure_big_lds((const void*)bneck_a_kernel<P, NT>) != OCCD_OK ||
            occd::ensure_big_lds((const void*)bneck_b_kernel<P, NT>) != OCCD_OK)
            return OCCD_ELAUNCH;
    }
    hipLaunchKernelGGL((bneck_a_kernel<P, NT>), dim3((unsigned)blocks_a), dim3(NT), lds_a, st, p);
    hipLaunchKernelGGL((bneck_b_kernel<P, NT>), dim3((unsigned)(p.xtiles * p.ytiles), (unsigned)p.batch), dim3(NT), lds_b, st, p);
    return occd::check_launch();
}

}  // namespace

extern "C" int64_t occd_bottleneck3d_weight_floats(int32_t C, int32_t P) {
    if (C <= 0 || P <= 0) return OCCD_EINVAL;
    return (int64_t)C * P + P + 3 * ((int64_t)3 * P * P + P) + (int64_t)P * C + C;
}

extern "C" int occd_bottleneck3d_fwd(const occd_bneck_args* a, void* stream) {
    if (!a || !a->x || !a->y || !a->o2 || !a->w) return OCCD_EINVAL;
    if (a->batch <= 0 || a->X <= 0 || a->Y <= 0 || a->Z <= 0 || a->Z > 16) return OCCD_EINVAL;
    if (a->P != 16 && a->P != 32) return OCCD_EINVAL;
    if (a->C <= 0 || (a->C & 15) || a->d0 <= 0 || a->d1 <= 0 || a->d2 <= 0) return OCCD_EINVAL;
    if ((a->x_cs & 3) || (a->x_coff & 3) || (a->y_cs & 3) || (a->y_coff & 3) || a->x_coff + a->C > a->x_cs ||
        a->y_coff + a->C > a->y_cs)
        return OCCD_EINVAL;
    if ((reinterpret_cast<uintptr_t>(a->x) | reinterpret_cast<uintptr_t>(a->y) | reinterpret_cast<uintptr_t>(a->o2) |
         reinterpret_cast<uintptr_t>(a->w)) & 15)
        return OCCD_EINVAL;
    BneckP p;
    p.x = a->x; p.y = a->y; p.o2 = a->o2;
    const int C = a->C, P = a->P;
    const float* w = a->w;
    p.w1 = w; w += (size_t)C * P; p.b1 = w; w += P;
    p.w2 = w; w += (size_t)3 * P * P; p.b2 = w; w += P;
    p.w3 = w; w += (size_t)3 * P * P; p.b3 = w; w += P;
    p.w4 = w; w += (size_t)3 * P * P; p.b4 = w; w += P;
    p.w5 = w; w += (size_t)P * C; p.b5 = w;
    p.batch = a->batch; p.X = a->X; p.Y = a->Y; p.Z = a->Z; p.C = C;
    p.x_cs = a->x_cs; p.x_coff = a->x_coff; p.y_cs = a->y_cs; p.y_coff = a->y_coff;
    p.d0 = a->d0; p.d1 = a->d1; p.d2 = a->d2;
    p.ncols = (long)a->batch * a->X * a->Y;
    // one lane quartet per voxel: a workgroup of NT threads owns NT voxels (4 tiles of 16 per wave)
    const long nvox_total = p.ncols * a->Z;
    const int NT = P == 16 ? 256 : 128;
    if (16 % a->Z != 0) return OCCD_EINVAL;            // a 16-voxel tile must hold whole columns (Z = 4, 8, 16)
    p.TY = a->Z == 16 ? 2 : 16 / a->Z;                  // TY * Z a multiple of 16
    if (a->Z == 8) p.TY = 2;
    p.TX = NT / (p.TY * a->Z);
    p.xtiles = (a->X + p.TX - 1) / p.TX;
    p.ytiles = (a->Y + p.TY - 1) / p.TY;
    const double vox = (double)nvox_total;
    occd::ProfScope prof("bottleneck3d", (hipStream_t)stream, vox * 2.0 * (2.0 * C * P + 9.0 * P * P),
                         vox * 4.0 * (3.0 * C + 2.0 * P));
    hipStream_t st = (hipStream_t)stream;
    if (P == 16) return launch<16, 256>(p, st);
    if (P == 32) return launch<32, 128>(p, st);
    return OCCD_EINVAL;
}
