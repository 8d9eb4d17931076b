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
//   bneck_a: columns of Z voxels, one thread per voxel: o1 = relu(W1 x) from an LDS-staged tile of x rows (coalesced
//            loads, 32 channels at a time), conv_z through LDS, o2 written once (P floats per voxel).
//   bneck_b: a TX x TY x Z tile, one thread per voxel: o3 on the tile plus its +-d2 planes along X straight from o2 rows
//            (three 4P-byte rows per voxel, L2 hits), o3 and o2 + o3 in LDS, conv_x, then y in chunks of 32 output
//            channels through an LDS transposition so that the residual read and the store are coalesced row segments.
// Arithmetic: float32 FMA on the vector ALU with the weights as SCALAR operands (they are uniform: s_load + v_fmac).  The
// fp32 MFMA rate equals the fp32 vector rate on gfx950 (157 TFLOP/s, MI355X_MICROARCH.md), so for these 16..64-channel
// reductions the matrix pipe would buy nothing but operand shuffles.
// HBM bytes per voxel: read x (4C) + write o2 (4P) | read o2 (4P, +halo from L2) + read x (4C) + write y (4C)
//   = 4 (3C + 2P) = 896 B at C = 64 (235 MB per block at 128x128x16) against 4 (3C + 8P) + launch tails before.
#include "common.h"
#include <algorithm>

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int kCH = 32;        // channel chunk of the x / y staging tiles
constexpr int kPadX = kCH + 4; // floats per staged row (16-byte aligned, conflict-free for 16 lanes of b128)

struct BneckP {
    const float* x;
    float* y;
    float* o2;
    const float* w1; const float* b1;     // [C][P], [P]
    const float* w2; const float* b2;     // [3][P][P] (tap, in, out), [P]
    const float* w3; const float* b3;
    const float* w4; const float* b4;
    const float* w5; const float* b5;     // [P][C], [C]
    int batch, X, Y, Z, C;
    int x_cs, x_coff, y_cs, y_coff;
    int d0, d1, d2;
    int TX, TY, xtiles, ytiles;
    long ncols;                           // batch * X * Y
};

__device__ __forceinline__ f32x4 relu4(f32x4 v) {
    return f32x4{fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f)};
}

// acc[n] += sum_k in[k] * w[k][n], in = K floats at `row` (LDS or global, 16-byte aligned), w uniform ([K][wstride])
template <int N, bool RELU_IN>
__device__ __forceinline__ void fma_rows(float (&acc)[N], const float* __restrict__ w, int wstride,
                                         const float* __restrict__ row, int K, bool valid) {
    for (int k4 = 0; k4 < K; k4 += 4) {
        f32x4 v = *(const f32x4*)(row + k4);
        if (RELU_IN) v = relu4(v);
        if (!valid) v = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float* __restrict__ wr = w + (size_t)(k4 + j) * wstride;
#pragma unroll
            for (int n = 0; n < N; ++n) acc[n] = fmaf(v[j], wr[n], acc[n]);
        }
    }
}

// ---------------------------------------------------------------- A: x -> o1 -> o2
template <int P, int NT>
__global__ void __launch_bounds__(NT) bneck_a_kernel(const BneckP p) {
    constexpr int PS = P + 4;                         // floats per o1 row in LDS
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* xs = lds;                                  // [NT][kPadX]
    float* o1s = lds + NT * kPadX;                    // [NT][PS]
    const int tid = threadIdx.x;
    const int ncol = NT / p.Z;                        // columns per workgroup
    const int nvox = ncol * p.Z;                      // voxels per workgroup (contiguous rows: z fastest, then y, x, b)
    const long v0 = (long)blockIdx.x * nvox;
    const long vtot = p.ncols * p.Z;
    const int z = tid % p.Z;
    const bool live = tid < nvox && v0 + tid < vtot;

    float acc[P];
#pragma unroll
    for (int n = 0; n < P; ++n) acc[n] = p.b1[n];
    for (int c0 = 0; c0 < p.C; c0 += kCH) {           // (C is a multiple of 32: checked by the host)
        __syncthreads();
        for (int i = tid; i < nvox * (kCH / 4); i += NT) {
            const int r = i >> 3, q = i & 7;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (v0 + r < vtot)
                v = *(const f32x4*)(p.x + (size_t)(v0 + r) * p.x_cs + p.x_coff + c0 + q * 4);
            *(f32x4*)(xs + r * kPadX + q * 4) = v;
        }
        __syncthreads();
        fma_rows<P, false>(acc, p.w1 + (size_t)c0 * P, P, xs + (tid < nvox ? tid : 0) * kPadX, kCH, true);
    }
#pragma unroll
    for (int n = 0; n < P; n += 4)
        *(f32x4*)(o1s + tid * PS + n) = relu4(f32x4{acc[n], acc[n + 1], acc[n + 2], acc[n + 3]});
    __syncthreads();

    float o2[P];
#pragma unroll
    for (int n = 0; n < P; ++n) o2[n] = p.b2[n];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int zz = z + (k - 1) * p.d0;
        const bool ok = zz >= 0 && zz < p.Z && tid < nvox;
        fma_rows<P, false>(o2, p.w2 + (size_t)k * P * P, P, o1s + (ok ? tid + (k - 1) * p.d0 : tid) * PS, P, ok);
    }
    if (live) {
        float* dst = p.o2 + (size_t)(v0 + tid) * P;
#pragma unroll
        for (int n = 0; n < P; n += 4) *(f32x4*)(dst + n) = f32x4{o2[n], o2[n + 1], o2[n + 2], o2[n + 3]};
    }
}

// ---------------------------------------------------------------- B: o2 -> o3 -> o4 -> y
template <int P, int NT>
__global__ void __launch_bounds__(NT) bneck_b_kernel(const BneckP p) {
    constexpr int PS = P + 4;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x;
    const int TYZ = p.TY * p.Z;
    const int nvox = p.TX * TYZ;                       // voxels (threads at work) of the tile
    const int XR = p.TX + 2 * p.d2;                    // o3 planes held
    float* o3s = lds;                                  // [XR * TYZ][PS]   raw o3 (zero outside the volume)
    float* ys = lds;                                   // [nvox][kPadX]    aliases o3s after the conv_x pass
    float* cs = lds + std::max((size_t)XR * TYZ * PS, (size_t)nvox * kPadX);   // [nvox][PS]  o2 + o3, later relu(o4)
    const int b = blockIdx.y;
    const int xt = blockIdx.x / p.ytiles, yt = blockIdx.x - xt * p.ytiles;
    const int x0 = xt * p.TX, y0 = yt * p.TY;
    const float* o2b = p.o2 + (size_t)b * p.X * p.Y * p.Z * P;

    // ---- o3 on the tile and its X halo
    for (int r = tid; r < XR * TYZ; r += NT) {
        const int xr = r / TYZ, rem = r - xr * TYZ;
        const int ty = rem / p.Z, z = rem - ty * p.Z;
        const int xx = x0 - p.d2 + xr, yy = y0 + ty;
        float o3[P];
        float o2c[P];
        const bool in = xx >= 0 && xx < p.X && yy < p.Y;
        const size_t col = ((size_t)(in ? xx : 0) * p.Y + (in ? yy : 0)) * p.Z + z;
        {
            const float* src = o2b + col * P;
#pragma unroll
            for (int n = 0; n < P; n += 4) {
                const f32x4 v = *(const f32x4*)(src + n);
                o2c[n] = v.x; o2c[n + 1] = v.y; o2c[n + 2] = v.z; o2c[n + 3] = v.w;
            }
        }
#pragma unroll
        for (int n = 0; n < P; ++n) o3[n] = p.b3[n] + o2c[n];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int yk = yy + (k - 1) * p.d1;
            const bool ok = in && yk >= 0 && yk < p.Y;
            const float* src = o2b + (((size_t)(in ? xx : 0) * p.Y + (ok ? yk : 0)) * p.Z + z) * P;
            fma_rows<P, true>(o3, p.w3 + (size_t)k * P * P, P, src, P, ok);
        }
#pragma unroll
        for (int n = 0; n < P; n += 4)
            *(f32x4*)(o3s + (size_t)r * PS + n) = in ? f32x4{o3[n], o3[n + 1], o3[n + 2], o3[n + 3]} : f32x4{0.f, 0.f, 0.f, 0.f};
        if (xr >= p.d2 && xr < p.d2 + p.TX) {
            const int tc = r - p.d2 * TYZ;
#pragma unroll
            for (int n = 0; n < P; n += 4)
                *(f32x4*)(cs + (size_t)tc * PS + n) = f32x4{o2c[n] + o3[n], o2c[n + 1] + o3[n + 1], o2c[n + 2] + o3[n + 2],
                                                           o2c[n + 3] + o3[n + 3]};
        }
    }
    __syncthreads();

    // ---- o4 = conv_x(relu(o3)) + b4 + o2 + o3, kept as relu(o4) in `cs`
    const int tv = tid < nvox ? tid : 0;
    {
        float o4[P];
#pragma unroll
        for (int n = 0; n < P; n += 4) {
            const f32x4 v = *(const f32x4*)(cs + (size_t)tv * PS + n);
            o4[n] = v.x + p.b4[n]; o4[n + 1] = v.y + p.b4[n + 1]; o4[n + 2] = v.z + p.b4[n + 2]; o4[n + 3] = v.w + p.b4[n + 3];
        }
#pragma unroll
        for (int k = 0; k < 3; ++k)
            fma_rows<P, true>(o4, p.w4 + (size_t)k * P * P, P, o3s + (size_t)(tv + k * p.d2 * TYZ) * PS, P, true);
        __syncthreads();                                // every conv_x read of o3s is done: the staging tile may alias it
        if (tid < nvox) {
#pragma unroll
            for (int n = 0; n < P; n += 4)
                *(f32x4*)(cs + (size_t)tv * PS + n) = relu4(f32x4{o4[n], o4[n + 1], o4[n + 2], o4[n + 3]});
        }
    }
    // (each thread reads back only its own `cs` row below: no barrier needed for it)

    // ---- y = relu(W5 relu(o4) + b5 + x), 32 output channels at a time
    for (int c0 = 0; c0 < p.C; c0 += kCH) {               // (C is a multiple of 32: checked by the host)
        __syncthreads();
        for (int i = tid; i < nvox * (kCH / 4); i += NT) {      // residual rows of x, coalesced
            const int r = i >> 3, q = i & 7;
            const int tx = r / TYZ, rem = r - tx * TYZ;
            const int ty = rem / p.Z, z = rem - ty * p.Z;
            const int xx = x0 + tx, yy = y0 + ty;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (xx < p.X && yy < p.Y)
                v = *(const f32x4*)(p.x + ((((size_t)b * p.X + xx) * p.Y + yy) * p.Z + z) * p.x_cs + p.x_coff + c0 + q * 4);
            *(f32x4*)(ys + r * kPadX + q * 4) = v;
        }
        __syncthreads();
        float acc[kCH];
#pragma unroll
        for (int n = 0; n < kCH; n += 4) {
            const f32x4 v = *(const f32x4*)(ys + tv * kPadX + n);
            acc[n] = v.x; acc[n + 1] = v.y; acc[n + 2] = v.z; acc[n + 3] = v.w;
        }
#pragma unroll
        for (int n = 0; n < kCH; ++n) acc[n] += p.b5[c0 + n];
        fma_rows<kCH, false>(acc, p.w5 + c0, p.C, cs + (size_t)tv * PS, P, true);
        if (tid < nvox) {
#pragma unroll
            for (int n = 0; n < kCH; n += 4)
                *(f32x4*)(ys + tv * kPadX + n) = relu4(f32x4{acc[n], acc[n + 1], acc[n + 2], acc[n + 3]});
        }
        __syncthreads();
        for (int i = tid; i < nvox * (kCH / 4); i += NT) {      // coalesced store
            const int r = i >> 3, q = i & 7;
            const int tx = r / TYZ, rem = r - tx * TYZ;
            const int ty = rem / p.Z, z = rem - ty * p.Z;
            const int xx = x0 + tx, yy = y0 + ty;
            if (xx < p.X && yy < p.Y)
                *(f32x4*)(p.y + ((((size_t)b * p.X + xx) * p.Y + yy) * p.Z + z) * p.y_cs + p.y_coff + c0 + q * 4) =
                    *(const f32x4*)(ys + r * kPadX + q * 4);
        }
    }
}

template <int P, int NT>
int launch(const BneckP& p, hipStream_t st) {
    constexpr int PS = P + 4;
    const int ncol = NT / p.Z;
    const size_t lds_a = (size_t)NT * (kPadX + PS) * sizeof(float);
    const long blocks_a = (p.ncols + ncol - 1) / ncol;
    const int TYZ = p.TY * p.Z, nvox = p.TX * TYZ;
    const size_t lds_b = sizeof(float) * (std::max((size_t)(p.TX + 2 * p.d2) * TYZ * PS, (size_t)nvox * kPadX) + (size_t)nvox * PS);
    if (lds_a > 64 * 1024 || lds_b > 64 * 1024) {
        static bool done = false;                      // per instantiation
        if (!done) {
            if (hipFuncSetAttribute((const void*)bneck_a_kernel<P, NT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
                hipFuncSetAttribute((const void*)bneck_b_kernel<P, NT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
                return OCCD_ELAUNCH;
            done = true;
        }
    }
    if (lds_a > 160 * 1024 || lds_b > 160 * 1024) return OCCD_ENOMEM;
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
    if (a->batch <= 0 || a->X <= 0 || a->Y <= 0 || a->Z <= 0 || a->Z > 64) return OCCD_EINVAL;
    if (a->P != 16 && a->P != 32 && a->P != 64) return OCCD_EINVAL;
    if (a->C <= 0 || (a->C & 31) || a->d0 <= 0 || a->d1 <= 0 || a->d2 <= 0) return OCCD_EINVAL;
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
    // workgroup size: one thread per voxel; fewer threads where the grid is small (more workgroups) or P is wide (LDS)
    const long nvox_total = p.ncols * a->Z;
    int NT = P == 16 ? 256 : P == 32 ? 128 : 64;
    while (NT > 64 && nvox_total / NT < 512) NT >>= 1;
    if (a->Z > NT) return OCCD_EINVAL;
    const int cols = NT / a->Z;                        // columns of the B tile: TX x TY with TY <= 2
    p.TY = cols >= 8 && a->Y >= 2 ? 2 : 1;
    p.TX = std::max(1, cols / p.TY);
    if (p.TX > a->X) p.TX = a->X;
    p.xtiles = (a->X + p.TX - 1) / p.TX;
    p.ytiles = (a->Y + p.TY - 1) / p.TY;
    const double vox = (double)nvox_total;
    occd::ProfScope prof("bottleneck3d", (hipStream_t)stream, vox * 2.0 * (2.0 * C * P + 9.0 * P * P),
                         vox * 4.0 * (3.0 * C + 2.0 * P));
    hipStream_t st = (hipStream_t)stream;
#define OCCD_BN(PP, TT) if (P == PP && NT == TT) return launch<PP, TT>(p, st);
    OCCD_BN(16, 256) OCCD_BN(16, 128) OCCD_BN(16, 64) OCCD_BN(32, 128) OCCD_BN(32, 64) OCCD_BN(64, 64)
#undef OCCD_BN
    return OCCD_EINVAL;
}
