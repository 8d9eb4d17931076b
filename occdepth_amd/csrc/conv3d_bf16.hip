// K2b -- implicit-GEMM 3-D convolution on the CDNA4 bf16 matrix pipe (BASELINE configs[3]: the bf16 training step).
//
//   out[vox][co] = act_out( sum_{tap,ci} bf16(act_in(in[vox*stride - pad + tap*dil][ci])) * bf16(W[co][ci][tap])
//                           + bias[co] + res1[vox][co] + res2[vox][co] )          (fp32 accumulate)
//
// Same GEMM view, tiling, output scatter and epilogue as K2 (conv3d_igemm.hip); what changes is the instruction --
// v_mfma_f32_32x32x16_bf16: 16 K values per instruction at 32 cycles, 16x the fp32 MFMA rate -- and with it the
// bottleneck: operand delivery.  A wave therefore owns up to 4 x 2 tiles of 32 voxels x 32 couts (each B fragment
// loaded from L2 feeds MT MFMAs, each A fragment read from LDS feeds NT), the staged slab holds bf16 (half the LDS
// bytes: [YIN][ZIN][32 channels] rows of 64 B + 16 B pad, an odd number of 16-B slots -> conflict-free ds_read_b128),
// and activations may live in HBM as fp32 (converted once, while staging: "bf16 MFMA, fp32 storage") or as bf16.
// Fragment layout (pinned on hardware by tools/probe_bf16.hip): lane l holds 8 consecutive K values 8 (l >> 5) + j of
// row / column l & 31; D as for every 32x32 MFMA: column = l & 31, row = (r & 3) + 8 (r >> 2) + 4 (l >> 5).
// The 2-D decoder's 3x3 convolutions run through the same kernel as X = 1 volumes of channels-last (NHWC) images.
//
// Reference semantics replaced (training step, N1): occdepth/models/DDR.py:111-139, modules.py:40-46,158-175,278-296,
// CRP3D.py:54-97, unet2d.py:24-46 -- forward and, on dL/dy with flipped weights, the data gradient.
#include "common.h"

#include <algorithm>
#include <cstddef>
#include <cstdlib>
#include <cstring>

using occd::FastDiv;

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

namespace {

// what differs between the sub-pixel phases of one transposed convolution (see K2's PhaseP, csrc/conv3d_igemm.hip)
struct PhaseBP {
    const u32x4* wpk;
    int KX, KY, KZ, oox, ooy, ooz, YIN, ZIN;
    FastDiv div_zin;
};
constexpr int kMaxPhasesB = 8;

struct ConvBP {
    const void* in;
    const float* bias;
    const void* res1;
    const void* res2;
    void* out;
    int X, Y, Z, cin8, cin16, in_cs, in_coff;
    int K16tot, NTtot;
    int out_cs, out_coff, res1_cs, res1_coff, res2_cs, res2_coff;
    int SX, SY, SZ, DX, DY, DZ, PX, PY, PZ;
    int Xo, Yo, Zo, OX, OY, OZ, osx, osy, osz;
    int act_in, act_out, cout_store;
    int TY, TZ, ytiles, ztiles, nwg;
    int nph_log2;           // blockIdx.y = (batch index << nph_log2) | phase, or (ph_fast) linear id = (tile << nph_log2) | phase
    int ph_fast;
    FastDiv div_tz, div_ztiles, div_ytiles;
    PhaseBP ph[kMaxPhasesB];
};

__device__ __forceinline__ f32x4 relu4(f32x4 v) {
    v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
    return v;
}
// (the expf + IEEE division form of K2's apply_act, csrc/conv3d_igemm.hip: same values)
__device__ __forceinline__ f32x4 sigmoid4(f32x4 v) {
    v.x = 1.f / (1.f + expf(-v.x)); v.y = 1.f / (1.f + expf(-v.y));
    v.z = 1.f / (1.f + expf(-v.z)); v.w = 1.f / (1.f + expf(-v.w));
    return v;
}

__device__ __forceinline__ u32x4 pack_bf16x8(f32x4 a, f32x4 b) {
    bf16x8 r = {(__bf16)a.x, (__bf16)a.y, (__bf16)a.z, (__bf16)a.w, (__bf16)b.x, (__bf16)b.y, (__bf16)b.z, (__bf16)b.w};
    return __builtin_bit_cast(u32x4, r);
}

__device__ __forceinline__ f32x4 unpack_lo(u32x2 v) {       // 4 bf16 -> 4 floats
    f32x4 r;
    r.x = __builtin_bit_cast(float, v.x << 16);
    r.y = __builtin_bit_cast(float, v.x & 0xffff0000u);
    r.z = __builtin_bit_cast(float, v.y << 16);
    r.w = __builtin_bit_cast(float, v.y & 0xffff0000u);
    return r;
}

constexpr int kRSB = 80;   // LDS row stride in bytes: 32 bf16 channels + 16 B pad
constexpr int kRSB3 = 208; // SPLIT = 3: three bf16 planes (hi, mid, lo) of the 32 channels + 16 B pad (13 slots: odd)

// x = hi + mid + lo exactly (each bf16, 8 significant bits: 24 together) for every float32 x whose low parts do not
// underflow; the six products hi*hi, hi*mid, mid*hi, hi*lo, lo*hi, mid*mid then reproduce x*w to ~2^-24 relative.
__device__ __forceinline__ void split3(f32x4 a, f32x4 b, u32x4& hi, u32x4& mid, u32x4& lo) {
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

// SPLIT = 1: plain bf16 operands.  SPLIT = 3 (opt-in experiment): float32 activations and weights as
// three bf16 terms each, six MFMAs per K step instead of one -- float32-level accuracy at 6/16 of the fp32-MFMA time.
// KS > 1: in-workgroup split-K as in K2 (conv3d_igemm.hip): the KS wave groups take interleaved 32-channel chunks, each with
// its own LDS slab, and their accumulators are summed through LDS before the epilogue -- the 3x3x3 convolutions of the 1/8
// level (4096 voxels, K = 27 x 256): 128 output tiles for 256 CUs.
template <int MT, int NT, int WM, int WN, bool IN_BF16, bool OUT_BF16, int SPLIT = 1, int KS = 1>
__global__ void __launch_bounds__(WM* WN * KS * 64) conv3d_bf16_kernel(const ConvBP p) {
    constexpr int NTH = WM * WN * 64;   // threads of one K group
    constexpr int RSB = SPLIT == 3 ? kRSB3 : kRSB;
    static_assert(SPLIT == 1 || (SPLIT == 3 && !IN_BF16 && !OUT_BF16), "the 3-way split takes and returns float32");
    extern __shared__ __attribute__((aligned(16))) unsigned char slab_all[];

    const int lane = threadIdx.x & 63;
    const int wave_all = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int kg = KS == 1 ? 0 : wave_all / (WM * WN);          // K group of this wave
    const int wave = wave_all - kg * (WM * WN);
    const int tid = threadIdx.x - kg * NTH;                      // thread index inside the K group
    const int wm = wave / WN;
    const int wn = wave - wm * WN;
    const int li = lane & 31;
    const int h = lane >> 5;

    uint32_t bid = blockIdx.x;   // XCD-aware bijective remap (see K2): an XCD walks a contiguous run of tiles
    {
        const uint32_t nwg = p.nwg, q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    // phase of a merged transposed-convolution launch: the low bits of the (remapped) linear id when ph_fast -- the phases of
    // one output tile run next to each other on one XCD, so their interleaved voxel rows meet in its L2 and the input tile is
    // fetched once -- else the low bits of blockIdx.y (phase-major dispatch, heaviest tap subset first)
    const uint32_t ph_mask = (1u << p.nph_log2) - 1u;
    const uint32_t ph_i = p.ph_fast ? bid & ph_mask : blockIdx.y & ph_mask;
    if (p.ph_fast) bid >>= p.nph_log2;
    const uint32_t t1 = occd_fastdiv(bid, p.div_ztiles);
    const int zt = bid - t1 * p.ztiles;
    const uint32_t t2 = occd_fastdiv(t1, p.div_ytiles);
    const int yt = t1 - t2 * p.ytiles;
    const int xo = t2;
    const int b = p.ph_fast ? blockIdx.y : blockIdx.y >> p.nph_log2;
    const PhaseBP& ph = p.ph[ph_i];
    const int nt0 = (blockIdx.z * WN + wn) * NT;

    int rowbase[MT];   // byte offset of this lane's A row (tap (0,0), k16 0) for each M tile
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const uint32_t m = (wm * MT + mt) * 32 + li;
        uint32_t yl = occd_fastdiv(m, p.div_tz);
        uint32_t zl = m - yl * p.TZ;
        const bool in_tile = yl < (uint32_t)p.TY;
        yl = in_tile ? yl : 0u;
        zl = in_tile ? zl : 0u;
        rowbase[mt] = (int)((yl * p.SY) * ph.ZIN + zl * p.SZ) * RSB + h * 16;
    }
    int wofs[NT];      // u32x4 index of each owned N tile inside one (tap, k16) weight record row
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) wofs[nt] = min(nt0 + nt, p.NTtot - 1) * 64;

    f32x16 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;

    const int y_in0 = yt * p.TY * p.SY - p.PY;
    const int z_in0 = zt * p.TZ * p.SZ - p.PZ;
    const size_t w_step = (size_t)p.NTtot * 64;          // u32x4 per (tap, k16)
    const size_t w_img = (size_t)ph.KX * ph.KY * ph.KZ * p.K16tot * w_step;   // u32x4 per split image of the weights
    const u32x4* const wlane = ph.wpk + lane;
    const int rows = ph.YIN * ph.ZIN;
    unsigned char* const slab = slab_all + (size_t)kg * rows * RSB;   // this K group's slab
    const int n_chunks = (p.cin16 + 31) / 32;
    const int chunk_iters = (n_chunks + KS - 1) / KS;

#define OCCD_MFMA1(WS, XS)                                                                                          \
    _Pragma("unroll") for (int nt = 0; nt < NT; ++nt) _Pragma("unroll") for (int mt = 0; mt < MT; ++mt) acc[mt][nt] = \
        __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, b_cur[nt][WS]),                         \
                                                __builtin_bit_cast(bf16x8, a_cur[mt][XS]), acc[mt][nt], 0, 0, 0)
    // smallest terms first: (mid, mid), (hi, lo), (lo, hi), (hi, mid), (mid, hi), (hi, hi)
#define OCCD_MFMA_BLOCK()                                                  \
    if (SPLIT == 3) {                                                      \
        OCCD_MFMA1(SPLIT == 3 ? 1 : 0, SPLIT == 3 ? 1 : 0);                \
        OCCD_MFMA1(0, SPLIT == 3 ? 2 : 0);                                 \
        OCCD_MFMA1(SPLIT == 3 ? 2 : 0, 0);                                 \
        OCCD_MFMA1(0, SPLIT == 3 ? 1 : 0);                                 \
        OCCD_MFMA1(SPLIT == 3 ? 1 : 0, 0);                                 \
    }                                                                      \
    OCCD_MFMA1(0, 0)

    for (int kx = 0; kx < ph.KX; ++kx) {
        const int xi = xo * p.SX - p.PX + kx * p.DX;
        if (xi < 0 || xi >= p.X) continue;   // workgroup-uniform
        const size_t plane = ((size_t)(b * p.X + xi) * p.Y) * p.Z * p.in_cs + p.in_coff;
        for (int ci = 0; ci < chunk_iters; ++ci) {
            const int c0 = (ci * KS + kg) * 32;
            const bool active = KS == 1 || c0 < p.cin16;     // uniform per K group; every group still joins the barriers
            const int ck = active ? min(32, p.cin16 - c0) : 16;   // 16 or 32 channels in this chunk
            const int k16n = ck >> 4;
            const int sh = k16n;                         // chunks of 8 channels per row: 2 (shift 1) or 4 (shift 2)
            const int S = active ? ph.KY * ph.KZ * k16n : 0;
            const u32x4* wp = wlane + ((size_t)(kx * ph.KY * ph.KZ) * p.K16tot + (active ? c0 >> 4 : 0)) * w_step;

            u32x4 b_cur[NT][SPLIT];   // first B fragments fly while the slab is staged
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int sp = 0; sp < SPLIT; ++sp) b_cur[nt][sp] = wp[wofs[nt] + sp * w_img];

            __syncthreads();   // previous slab fully consumed
            const int F = rows << sh;
            for (int f0 = 0; active && f0 < F; f0 += NTH * 4) {
                u32x4 v[4][SPLIT];
                int dst[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int f = f0 + u * NTH + tid;
                    const int fc = min(f, F - 1);
                    const uint32_t row = (uint32_t)fc >> sh;
                    const int c8 = fc & ((1 << sh) - 1);
                    const uint32_t yi = occd_fastdiv(row, ph.div_zin);
                    const int zi = (int)row - (int)yi * ph.ZIN;
                    const int y = y_in0 + (int)yi, z = z_in0 + zi;
                    const int c = c0 + c8 * 8;
                    const bool ok = f < F && c < p.cin8 && (unsigned)y < (unsigned)p.Y && (unsigned)z < (unsigned)p.Z;
                    const int yc = min(max(y, 0), p.Y - 1), zc = min(max(z, 0), p.Z - 1);
                    const int cc = min(c, p.cin8 - 8);
                    const size_t e = plane + ((size_t)yc * p.Z + zc) * p.in_cs + cc;
                    u32x4 w[SPLIT];
                    if (IN_BF16) {
                        w[0] = *(const u32x4*)((const uint16_t*)p.in + e);
                        if (p.act_in == OCCD_ACT_RELU) {   // bf16 relu on the packed pairs: clear negative halves
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                uint32_t d = w[0][q];
                                if (d & 0x80000000u) d &= 0x0000ffffu;
                                if (d & 0x00008000u) d &= 0xffff0000u;
                                w[0][q] = d;
                            }
                        }
                    } else {
                        f32x4 lo = *(const f32x4*)((const float*)p.in + e);
                        f32x4 hi = *(const f32x4*)((const float*)p.in + e + 4);
                        if (p.act_in == OCCD_ACT_RELU) { lo = relu4(lo); hi = relu4(hi); }
                        else if (SPLIT == 3 && p.act_in == OCCD_ACT_SIGMOID) { lo = sigmoid4(lo); hi = sigmoid4(hi); }
                        if (SPLIT == 3) split3(lo, hi, w[0], w[SPLIT == 3 ? 1 : 0], w[SPLIT == 3 ? 2 : 0]);
                        else w[0] = pack_bf16x8(lo, hi);
                    }
#pragma unroll
                    for (int sp = 0; sp < SPLIT; ++sp) v[u][sp] = ok ? w[sp] : u32x4{0u, 0u, 0u, 0u};
                    dst[u] = f < F ? (int)row * RSB + c8 * 16 : -1;
                }
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (dst[u] >= 0) {
#pragma unroll
                        for (int sp = 0; sp < SPLIT; ++sp) *(u32x4*)(slab + dst[u] + sp * 64) = v[u][sp];
                    }
            }
            __syncthreads();

            int ky = 0, kz = 0, kl = 0;
            int lds_off = 0;   // bytes
            u32x4 a_cur[MT][SPLIT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int sp = 0; sp < SPLIT; ++sp) a_cur[mt][sp] = *(const u32x4*)(slab + rowbase[mt] + sp * 64);

            for (int s = 0; s < S - 1; ++s) {
                ++kl;
                lds_off += 32;
                wp += w_step;
                if (kl == k16n) {
                    kl = 0;
                    wp += (size_t)(p.K16tot - k16n) * w_step;
                    if (++kz == ph.KZ) { kz = 0; ++ky; }
                    lds_off = (ky * p.DY * ph.ZIN + kz * p.DZ) * RSB;
                }
                u32x4 a_nxt[MT][SPLIT], b_nxt[NT][SPLIT];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int sp = 0; sp < SPLIT; ++sp) a_nxt[mt][sp] = *(const u32x4*)(slab + rowbase[mt] + lds_off + sp * 64);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int sp = 0; sp < SPLIT; ++sp) b_nxt[nt][sp] = wp[wofs[nt] + sp * w_img];
                OCCD_MFMA_BLOCK();
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int sp = 0; sp < SPLIT; ++sp) a_cur[mt][sp] = a_nxt[mt][sp];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int sp = 0; sp < SPLIT; ++sp) b_cur[nt][sp] = b_nxt[nt][sp];
            }
            if (active) { OCCD_MFMA_BLOCK(); }
        }
    }
#undef OCCD_MFMA1
#undef OCCD_MFMA_BLOCK

    if (KS > 1) {
        // sum the K groups' accumulators through LDS (slabs are dead after the barrier), group 0 stores
        float* red = reinterpret_cast<float*>(slab_all);
        __syncthreads();
        if (kg > 0) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        red[((((kg - 1) * WM * WN + wave) * MT + mt) * NT + nt) * 1024 + r * 64 + lane] = acc[mt][nt][r];
        }
        __syncthreads();
        if (kg > 0) return;
#pragma unroll
        for (int g = 1; g < KS; ++g)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        acc[mt][nt][r] += red[((((g - 1) * WM * WN + wave) * MT + mt) * NT + nt) * 1024 + r * 64 + lane];
    }

    // ---------------- epilogue (the layout of K2's: lane -> voxel li of the M tile, registers -> couts
    // (r & 3) + 8 (r >> 2) + 4 h: four groups of 4 consecutive output channels of ONE voxel per lane)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const uint32_t m = (wm * MT + mt) * 32 + li;
        const uint32_t yl = occd_fastdiv(m, p.div_tz);
        const uint32_t zl = m - yl * p.TZ;
        const int yo = yt * p.TY + (int)yl, zo = zt * p.TZ + (int)zl;
        const bool ok = yl < (uint32_t)p.TY && yo < p.Yo && zo < p.Zo;
        const size_t vox = ((size_t)(b * p.OX + xo * p.osx + ph.oox) * p.OY + (yo * p.osy + ph.ooy)) * p.OZ +
                           (zo * p.osz + ph.ooz);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int c = (nt0 + nt) * 32 + 8 * g + 4 * h;
                if (ok && c < p.cout_store) {
                    f32x4 v = {acc[mt][nt][4 * g], acc[mt][nt][4 * g + 1], acc[mt][nt][4 * g + 2],
                               acc[mt][nt][4 * g + 3]};
                    if (p.bias != nullptr) v += *(const f32x4*)(p.bias + c);
                    if (p.act_out == OCCD_ACT_RELU_PRE) v = relu4(v);
                    if (OUT_BF16) {
                        if (p.res1 != nullptr)
                            v += unpack_lo(*(const u32x2*)((const uint16_t*)p.res1 + vox * p.res1_cs + p.res1_coff + c));
                        if (p.res2 != nullptr)
                            v += unpack_lo(*(const u32x2*)((const uint16_t*)p.res2 + vox * p.res2_cs + p.res2_coff + c));
                    } else {
                        if (p.res1 != nullptr) v += *(const f32x4*)((const float*)p.res1 + vox * p.res1_cs + p.res1_coff + c);
                        if (p.res2 != nullptr) v += *(const f32x4*)((const float*)p.res2 + vox * p.res2_cs + p.res2_coff + c);
                    }
                    if (p.act_out == OCCD_ACT_RELU) v = relu4(v);
                    if (OUT_BF16) {
                        bf16x4 o = {(__bf16)v.x, (__bf16)v.y, (__bf16)v.z, (__bf16)v.w};
                        *(u32x2*)((uint16_t*)p.out + vox * p.out_cs + p.out_coff + c) = __builtin_bit_cast(u32x2, o);
                    } else {
                        *(f32x4*)((float*)p.out + vox * p.out_cs + p.out_coff + c) = v;
                    }
                }
            }
        }
    }
}

// ---------------------------------------------------------------- weight packing (fp32 master weights -> bf16 fragments)
// wpk[tap][k16][nt][lane][j]: cout = nt * 32 + (lane & 31), cin = k16 * 16 + (lane >> 5) * 8 + j
// nsplit = 3: three consecutive images hi | mid | lo of the float32 weight (see split3)
__global__ void pack_weights_bf16_kernel(const float* __restrict__ w, const float* __restrict__ scale,
                                         uint16_t* __restrict__ wpk, int cout, int cin, int taps, int K16, int NT,
                                         int layout, long total, int nsplit, const occd::TapMap tm) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int j = i & 7;
    const int lane = (i >> 3) & 63;
    long t = i >> 9;
    const int nt = t % NT; t /= NT;
    const int k16 = t % K16; t /= K16;
    const int tap = (int)t;
    const int co = nt * 32 + (lane & 31);
    const int ci = k16 * 16 + (lane >> 5) * 8 + j;
    float v = 0.f;
    if (co < cout && ci < cin) {
        if (layout == 0) v = w[((size_t)co * cin + ci) * taps + tap];
        else if (layout == 1) v = w[((size_t)ci * cout + co) * taps + tap];
        else if (layout == 2) v = w[(size_t)ci * cout + co];
        else v = w[(size_t)co * tm.s_co + (size_t)ci * tm.s_ci + tm.ofs[tap]];     // (see pack_weights_kernel, layout 3)
        if (scale != nullptr) v *= scale[co];
    }
    const __bf16 hi = (__bf16)v;
    wpk[i] = __builtin_bit_cast(uint16_t, hi);
    if (nsplit == 3) {
        const float r = v - (float)hi;
        const __bf16 mid = (__bf16)r;
        wpk[i + total] = __builtin_bit_cast(uint16_t, mid);
        wpk[i + 2 * total] = __builtin_bit_cast(uint16_t, (__bf16)(r - (float)mid));
    }
}

struct VariantB {
    int MT, NT, WM, WN, KS;
    void (*kern[3])(const ConvBP);   // [0] fp32 in / fp32 out, [1] bf16 in / bf16 out, [2] fp32 with the 3-way split (or null)
};

#define OCCD_VARIANT_B(MT, NT, WM, WN) \
    VariantB { MT, NT, WM, WN, 1, { conv3d_bf16_kernel<MT, NT, WM, WN, false, false>, conv3d_bf16_kernel<MT, NT, WM, WN, true, true>, nullptr } }
#define OCCD_VARIANT_B3(MT, NT, WM, WN) \
    VariantB { MT, NT, WM, WN, 1, { conv3d_bf16_kernel<MT, NT, WM, WN, false, false>, conv3d_bf16_kernel<MT, NT, WM, WN, true, true>, \
                                    conv3d_bf16_kernel<MT, NT, WM, WN, false, false, 3> } }

const VariantB kVariantsB[] = {
    OCCD_VARIANT_B(4, 1, 4, 1),    // 0: M512 x N32
    OCCD_VARIANT_B3(2, 1, 4, 1),   // 1: M256 x N32
    OCCD_VARIANT_B3(1, 1, 4, 1),   // 2: M128 x N32
    OCCD_VARIANT_B(4, 2, 4, 1),    // 3: M512 x N64
    OCCD_VARIANT_B3(2, 2, 4, 1),   // 4: M256 x N64
    OCCD_VARIANT_B(4, 2, 2, 2),    // 5: M256 x N128
    OCCD_VARIANT_B3(2, 2, 2, 2),   // 6: M128 x N128
    OCCD_VARIANT_B3(1, 2, 2, 2),   // 7: M64  x N128
    // 8: M32 x N128 with the 3-way split and 4-way in-workgroup split-K (16 waves): small volumes with a long K
    VariantB { 1, 1, 1, 4, 4, { nullptr, nullptr, conv3d_bf16_kernel<1, 1, 1, 4, false, false, 3, 4> } },
};
constexpr int kNumVariantsB = sizeof(kVariantsB) / sizeof(kVariantsB[0]);
constexpr size_t kMaxLdsB = 160 * 1024;

struct TilingB {
    int TY, TZ, YIN, ZIN, ytiles, ztiles, ngroups;
    size_t lds;
    long nwg;
    double cost;
};

// Tile of mwg output positions as TY x TZ: the candidate that stages the fewest input rows per launch (halo + ragged
// edges) among those that fit LDS.  2-D images (Zo = W >> mwg) get a TY > 1 tile instead of a one-row strip.
bool plan_b(const occd_conv3d_args* a, const VariantB& v, int NTtot, TilingB* best, int rsb) {
    const int mwg = v.MT * v.WM * 32;
    bool found = false;
    int cands[16];
    int nc = 0;
    if (a->Zo <= mwg) cands[nc++] = a->Zo;                       // whole columns (the 3-D stack: Z = 4 ... 32)
    for (int tz = 8; tz <= mwg; tz *= 2)
        if (tz < a->Zo) cands[nc++] = tz;
    for (int i = 0; i < nc; ++i) {
        TilingB t;
        t.TZ = cands[i];
        t.TY = mwg / t.TZ;
        if (t.TY < 1) continue;
        if (t.TY > a->Yo) t.TY = a->Yo;
        t.YIN = (t.TY - 1) * a->sy + (a->ky - 1) * a->dy + 1;
        t.ZIN = (t.TZ - 1) * a->sz + (a->kz - 1) * a->dz + 1;
        t.ytiles = (a->Yo + t.TY - 1) / t.TY;
        t.ztiles = (a->Zo + t.TZ - 1) / t.TZ;
        t.lds = (size_t)t.YIN * t.ZIN * rsb * v.KS;
        const size_t red = (size_t)(v.KS - 1) * v.WM * v.WN * v.MT * v.NT * 4096;   // split-K reduction scratch
        if (red > t.lds) t.lds = red;
        const int nwg_n = v.NT * v.WN;
        t.ngroups = (NTtot + nwg_n - 1) / nwg_n;
        t.nwg = (long)a->Xo * t.ytiles * t.ztiles;
        if (t.lds > kMaxLdsB || (long)t.YIN * t.ZIN >= 65536 || t.nwg >= (1L << 24)) continue;
        // staged rows per launch + idle M slots of a tile smaller than mwg (they still cost MFMA time)
        t.cost = (double)t.nwg * ((double)t.YIN * t.ZIN + 2.0 * (mwg - (double)t.TY * t.TZ));
        if (t.lds > 80 * 1024) t.cost *= 1.15;                   // one workgroup per CU instead of two
        if (!found || t.cost < best->cost) { *best = t; found = true; }
    }
    return found;
}

}  // namespace

extern "C" int64_t occd_packed_weight_bf16_elems(int32_t cout, int32_t cin, int32_t taps) {
    if (cout <= 0 || cin <= 0 || taps <= 0) return OCCD_EINVAL;
    const int64_t K16 = (cin + 15) / 16, NT = (cout + 31) / 32;
    return (int64_t)taps * K16 * NT * 512;
}

static int pack_bf16(const float* w, const float* scale, void* wpk, int32_t cout, int32_t cin, int32_t kx, int32_t ky,
                     int32_t kz, int32_t layout, int nsplit, void* stream);

extern "C" int occd_pack_weights_bf16(const float* w, const float* scale, void* wpk, int32_t cout, int32_t cin,
                                      int32_t kx, int32_t ky, int32_t kz, int32_t layout, void* stream) {
    return pack_bf16(w, scale, wpk, cout, cin, kx, ky, kz, layout, 1, stream);
}

// three images (hi | mid | lo), 3 * occd_packed_weight_bf16_elems() elements: the weight operand of the 3-way split
extern "C" int occd_pack_weights_bf16x3(const float* w, const float* scale, void* wpk, int32_t cout, int32_t cin,
                                        int32_t kx, int32_t ky, int32_t kz, int32_t layout, void* stream) {
    return pack_bf16(w, scale, wpk, cout, cin, kx, ky, kz, layout, 3, stream);
}

static int pack_bf16(const float* w, const float* scale, void* wpk, int32_t cout, int32_t cin, int32_t kx, int32_t ky,
                     int32_t kz, int32_t layout, int nsplit, void* stream) {
    if (!w || !wpk || layout < 0 || layout > 2) return OCCD_EINVAL;
    const int taps = kx * ky * kz;
    const int64_t total = occd_packed_weight_bf16_elems(cout, cin, taps);
    if (total <= 0 || (layout == 2 && taps != 1)) return OCCD_EINVAL;
    const int K16 = (cin + 15) / 16, NT = (cout + 31) / 32;
    const int th = 256;
    const long blocks = (total + th - 1) / th;
    occd::ProfScope prof("pack_weights_bf16", (hipStream_t)stream, 0.0, (double)total * 6);
    hipLaunchKernelGGL(pack_weights_bf16_kernel, dim3((unsigned)blocks), dim3(th), 0, (hipStream_t)stream, w, scale,
                       (uint16_t*)wpk, cout, cin, taps, K16, NT, layout, (long)total, nsplit, occd::TapMap{});
    return occd::check_launch();
}

static int pack_bf16_gather(const float* w, const float* scale, void* wpk, int32_t cout, int32_t cin, int32_t ntaps,
                            int64_t s_co, int64_t s_ci, const int32_t* tap_ofs, int nsplit, void* stream);

extern "C" int occd_pack_weights_bf16_gather(const float* w, const float* scale, void* wpk, int32_t cout, int32_t cin,
                                             int32_t ntaps, int64_t s_co, int64_t s_ci, const int32_t* tap_ofs,
                                             void* stream) {
    return pack_bf16_gather(w, scale, wpk, cout, cin, ntaps, s_co, s_ci, tap_ofs, 1, stream);
}

// the hi | mid | lo images of the gathered operator (3 * occd_packed_weight_bf16_elems() elements): the 3-way split of the
// data-gradient operators (float32-accurate dgrad of the head convolutions on K2s3 in the fp32 training mode)
extern "C" int occd_pack_weights_bf16x3_gather(const float* w, const float* scale, void* wpk, int32_t cout, int32_t cin,
                                               int32_t ntaps, int64_t s_co, int64_t s_ci, const int32_t* tap_ofs,
                                               void* stream) {
    return pack_bf16_gather(w, scale, wpk, cout, cin, ntaps, s_co, s_ci, tap_ofs, 3, stream);
}

static int pack_bf16_gather(const float* w, const float* scale, void* wpk, int32_t cout, int32_t cin, int32_t ntaps,
                            int64_t s_co, int64_t s_ci, const int32_t* tap_ofs, int nsplit, void* stream) {
    if (!w || !wpk || !tap_ofs || ntaps <= 0 || ntaps > occd::kMaxTaps) return OCCD_EINVAL;
    const int64_t total = occd_packed_weight_bf16_elems(cout, cin, ntaps);
    if (total <= 0) return OCCD_EINVAL;
    occd::TapMap tm{};
    tm.s_co = s_co; tm.s_ci = s_ci;
    for (int i = 0; i < ntaps; ++i) tm.ofs[i] = tap_ofs[i];
    const int K16 = (cin + 15) / 16, NT = (cout + 31) / 32;
    const int th = 256;
    const long blocks = (total + th - 1) / th;
    occd::ProfScope prof("pack_weights_bf16", (hipStream_t)stream, 0.0, (double)total * 6);
    hipLaunchKernelGGL(pack_weights_bf16_kernel, dim3((unsigned)blocks), dim3(th), 0, (hipStream_t)stream, w, scale,
                       (uint16_t*)wpk, cout, cin, ntaps, K16, NT, 3, (long)total, nsplit, tm);
    return occd::check_launch();
}

// `a->in`, `a->out`, `a->res1`, `a->res2` point at fp32 (dtype 0) or bf16 (dtype 1) channels-last rows -- all four
// the same type --, `a->wpk` at the image of occd_pack_weights_bf16, `a->bias` at fp32.  *_cs / *_coff count ELEMENTS.
// dtype 2 = float32 tensors with the 3-way bf16 split of both operands (wpk from occd_pack_weights_bf16x3): float32-level
// accuracy on the bf16 matrix pipe, an opt-in experiment beside the exact-fp32 kernels.
namespace {

int validate_b(const occd_conv3d_args* a, int32_t dtype) {
    if (!a || !a->in || !a->wpk || !a->out || dtype < 0 || dtype > 2) return OCCD_EINVAL;
    const int ksel = dtype;                           // kernel table column
    if (dtype == 2) dtype = 0;                        // storage: float32
    if (a->batch <= 0 || a->X <= 0 || a->Y <= 0 || a->Z <= 0 || a->cin <= 0 || a->cout <= 0) return OCCD_EINVAL;
    if (a->kx <= 0 || a->ky <= 0 || a->kz <= 0 || a->sx <= 0 || a->sy <= 0 || a->sz <= 0) return OCCD_EINVAL;
    if (a->Xo <= 0 || a->Yo <= 0 || a->Zo <= 0) return OCCD_EINVAL;
    const int cin8 = (a->cin + 7) & ~7;
    const int cin16 = (a->cin + 15) & ~15;
    const int NTtot = (a->cout + 31) / 32;
    const int al = dtype == 1 ? 7 : 3;               // 16-byte staging loads: 8 bf16 / 4 (+4) floats
    const int esz = dtype == 1 ? 2 : 4;
    if ((a->in_cs & al) || (a->in_coff & al) || a->in_coff + cin8 > a->in_cs) return OCCD_EINVAL;
    if ((reinterpret_cast<uintptr_t>(a->in) & 15) || (reinterpret_cast<uintptr_t>(a->wpk) & 15)) return OCCD_EINVAL;
    if (a->cout_store < a->cout || a->cout_store > NTtot * 32 || a->out_coff + a->cout_store > a->out_cs)
        return OCCD_EINVAL;
    // 4-channel epilogue groups: 16 B (fp32) / 8 B (bf16) accesses
    if ((a->cout_store & 3) || (a->out_cs & 3) || (a->out_coff & 3) ||
        (reinterpret_cast<uintptr_t>(a->out) & (4 * esz - 1)))
        return OCCD_EINVAL;
    if (a->res1 && ((a->res1_cs & 3) || (a->res1_coff & 3) || (reinterpret_cast<uintptr_t>(a->res1) & (4 * esz - 1)) ||
                    a->res1_coff + a->cout_store > a->res1_cs))
        return OCCD_EINVAL;
    if (a->res2 && ((a->res2_cs & 3) || (a->res2_coff & 3) || (reinterpret_cast<uintptr_t>(a->res2) & (4 * esz - 1)) ||
                    a->res2_coff + a->cout_store > a->res2_cs))
        return OCCD_EINVAL;
    if (a->bias && (reinterpret_cast<uintptr_t>(a->bias) & 15)) return OCCD_EINVAL;
    if ((a->Xo - 1) * a->o_stride_x + a->o_off_x >= a->OX || (a->Yo - 1) * a->o_stride_y + a->o_off_y >= a->OY ||
        (a->Zo - 1) * a->o_stride_z + a->o_off_z >= a->OZ)
        return OCCD_EINVAL;
    // (input sigmoid: the CRP products sigmoid(P_logits) @ mega, CRP3D.py:80 -- float32 operands with the split only)
    if (a->act_in != OCCD_ACT_NONE && a->act_in != OCCD_ACT_RELU && !(a->act_in == OCCD_ACT_SIGMOID && ksel == 2)) return OCCD_EINVAL;
    if (a->act_out != OCCD_ACT_NONE && a->act_out != OCCD_ACT_RELU && a->act_out != OCCD_ACT_RELU_PRE)
        return OCCD_EINVAL;
    return OCCD_OK;
}

// n phases (n = 1: a plain launch) that differ only in wpk, kx / ky / kz and o_off_*: ONE launch
int launch_b(const occd_conv3d_args* a, int n, int32_t dtype, hipStream_t stream) {
    const int ksel = dtype;                           // kernel table column
    if (dtype == 2) dtype = 0;                        // storage: float32
    const int esz = dtype == 1 ? 2 : 4;
    const int cin8 = (a->cin + 7) & ~7;
    const int cin16 = (a->cin + 15) & ~15;
    const int NTtot = (a->cout + 31) / 32;
    // tiling for the phase with the largest slab (the largest ky / kz over the phases)
    occd_conv3d_args big = a[0];
    for (int i = 1; i < n; ++i) {
        big.ky = a[i].ky > big.ky ? a[i].ky : big.ky;
        big.kz = a[i].kz > big.kz ? a[i].kz : big.kz;
    }
    const long copies = (long)a->batch * n;
    int order[kNumVariantsB];
    int no = 0;
    if (a->tile_hint > 0 && a->tile_hint <= kNumVariantsB) {
        order[no++] = a->tile_hint - 1;
    } else if (NTtot == 1) {
        order[no++] = 0; order[no++] = 1; order[no++] = 2;
    } else if (NTtot == 2) {
        order[no++] = 3; order[no++] = 4; order[no++] = 6; order[no++] = 7;
    } else {
        order[no++] = 5; order[no++] = 6; order[no++] = 7;
    }
    const int rsb = ksel == 2 ? kRSB3 : kRSB;
    int pick = -1;
    TilingB til{};
    for (int i = 0; i < no; ++i) {
        TilingB t{};
        if (kVariantsB[order[i]].kern[ksel] == nullptr) continue;
        if (!plan_b(&big, kVariantsB[order[i]], NTtot, &t, rsb)) continue;
        pick = order[i];
        til = t;
        if (t.nwg * t.ngroups * copies >= 512) break;   // else keep refining to the finest fit
    }
    if (pick < 0) return OCCD_ENOMEM;
    if (ksel == 2 && a->tile_hint == 0 && til.nwg * til.ngroups * copies <= 320 && cin16 >= 128 && NTtot % 4 == 0) {
        // too few output tiles to fill the chip: multiply the waves by splitting K inside the workgroup (K2's rule)
        TilingB t{};
        if (plan_b(&big, kVariantsB[8], NTtot, &t, rsb) && t.nwg * t.ngroups * copies <= 1024) {
            pick = 8;
            til = t;
        }
    }
    const VariantB& v = kVariantsB[pick];

    ConvBP p{};
    p.in = a->in; p.bias = a->bias; p.res1 = a->res1; p.res2 = a->res2; p.out = a->out;
    p.X = a->X; p.Y = a->Y; p.Z = a->Z; p.cin8 = cin8; p.cin16 = cin16; p.in_cs = a->in_cs; p.in_coff = a->in_coff;
    p.K16tot = cin16 / 16; p.NTtot = NTtot;
    p.out_cs = a->out_cs; p.out_coff = a->out_coff;
    p.res1_cs = a->res1_cs; p.res1_coff = a->res1_coff; p.res2_cs = a->res2_cs; p.res2_coff = a->res2_coff;
    p.SX = a->sx; p.SY = a->sy; p.SZ = a->sz;
    p.DX = a->dx; p.DY = a->dy; p.DZ = a->dz; p.PX = a->px; p.PY = a->py; p.PZ = a->pz;
    p.Xo = a->Xo; p.Yo = a->Yo; p.Zo = a->Zo; p.OX = a->OX; p.OY = a->OY; p.OZ = a->OZ;
    p.osx = a->o_stride_x; p.osy = a->o_stride_y; p.osz = a->o_stride_z;
    p.act_in = a->act_in; p.act_out = a->act_out; p.cout_store = a->cout_store;
    p.TY = til.TY; p.TZ = til.TZ;
    p.ytiles = til.ytiles; p.ztiles = til.ztiles; p.nwg = (int)til.nwg;
    p.div_tz = occd::make_fastdiv(til.TZ);
    p.div_ztiles = occd::make_fastdiv(til.ztiles); p.div_ytiles = occd::make_fastdiv(til.ytiles);
    p.nph_log2 = n == 1 ? 0 : n == 2 ? 1 : n == 4 ? 2 : 3;
    // phase-major dispatch is the default (see occd_conv3d_fwd_phases: the tile-major order, OCCD_PHASE_FAST=1, measured slower)
    static const bool phase_fast = occd::env_flag("OCCD_PHASE_FAST", false);
    p.ph_fast = n > 1 && phase_fast ? 1 : 0;
    if (p.ph_fast) p.nwg = (int)(til.nwg * n);
    if (til.nwg * n >= (1L << 24)) return OCCD_EINVAL;
    // heaviest tap subset first in dispatch order: the single-tap phases fill the tail
    int ord[kMaxPhasesB];
    for (int i = 0; i < n; ++i) ord[i] = i;
    std::stable_sort(ord, ord + n, [&](int l, int r) { return a[l].kx * a[l].ky * a[l].kz > a[r].kx * a[r].ky * a[r].kz; });
    double flops = 0.0, bytes = 0.0;
    const double pos = (double)a->batch * a->Xo * a->Yo * a->Zo;
    for (int i = 0; i < n; ++i) {
        const occd_conv3d_args* ai = a + ord[i];
        PhaseBP& ph = p.ph[i];
        ph.wpk = (const u32x4*)ai->wpk;
        ph.KX = ai->kx; ph.KY = ai->ky; ph.KZ = ai->kz;
        ph.oox = ai->o_off_x; ph.ooy = ai->o_off_y; ph.ooz = ai->o_off_z;
        ph.YIN = (til.TY - 1) * ai->sy + (ai->ky - 1) * ai->dy + 1;      // (the shared tile, this phase's extent)
        ph.ZIN = (til.TZ - 1) * ai->sz + (ai->kz - 1) * ai->dz + 1;
        ph.div_zin = occd::make_fastdiv(ph.ZIN);
        if ((size_t)ph.YIN * ph.ZIN * rsb * v.KS > til.lds) return OCCD_EINVAL;
        const double taps = (double)ai->kx * ai->ky * ai->kz;
        flops += 2.0 * pos * taps * ai->cin * ai->cout;
        bytes += (double)esz * ((i == 0 ? (double)a->batch * a->X * a->Y * a->Z * a->cin : 0.0) +
                                pos * a->cout * (1 + (a->res1 != nullptr) + (a->res2 != nullptr))) + 2.0 * taps * a->cin * a->cout;
    }
    if (copies > 65535) return OCCD_EINVAL;

    void (*kern)(const ConvBP) = v.kern[ksel];
    if (til.lds > 64 * 1024 && occd::ensure_big_lds(reinterpret_cast<const void*>(kern)) != OCCD_OK) return OCCD_ELAUNCH;
    occd::ProfScope prof(n > 1 ? "conv3d_bf16x3_phases" : ksel == 2 ? "conv3d_bf16x3" : dtype == 1 ? "conv3d_bf16s" : "conv3d_bf16",
                         stream, flops, bytes);
    hipLaunchKernelGGL(kern, p.ph_fast ? dim3((unsigned)(til.nwg * n), (unsigned)a->batch, (unsigned)til.ngroups)
                                       : dim3((unsigned)til.nwg, (unsigned)copies, (unsigned)til.ngroups),
                       dim3(v.WM * v.WN * v.KS * 64), til.lds, stream, p);
    return occd::check_launch();
}

}  // namespace

extern "C" int occd_conv3d_bf16_fwd(const occd_conv3d_args* a, int32_t dtype, void* stream) {
    const int bad = validate_b(a, dtype);
    if (bad != OCCD_OK) return bad;
    if (dtype == 2) {   // the full-resolution head launches: sliding-window form of the split (K2s3, conv3d_c32p.hip)
        const int r = occd::try_conv3d_c32_slide_x3(a, (hipStream_t)stream);
        if (r != 0) return r < 0 ? r : OCCD_OK;
    }
    return launch_b(a, 1, dtype, (hipStream_t)stream);
}

// The n in {1, 2, 4, 8} sub-pixel phases of ONE transposed convolution on K2b as one launch (the bf16-pipe twin of
// occd_conv3d_fwd_phases, same contract: a[0..n) differ ONLY in wpk, kx / ky / kz and o_off_*).
extern "C" int occd_conv3d_bf16_fwd_phases(const occd_conv3d_args* a, int32_t n, int32_t dtype, void* stream) {
    if (!a || (n != 1 && n != 2 && n != 4 && n != 8)) return OCCD_EINVAL;
    for (int i = 0; i < n; ++i) {
        const int bad = validate_b(a + i, dtype);
        if (bad != OCCD_OK) return bad;
        occd_conv3d_args t = a[i];
        t.wpk = a[0].wpk; t.kx = a[0].kx; t.ky = a[0].ky; t.kz = a[0].kz;
        t.o_off_x = a[0].o_off_x; t.o_off_y = a[0].o_off_y; t.o_off_z = a[0].o_off_z;
        if (memcmp(&t, &a[0], offsetof(occd_conv3d_args, tile_hint) + sizeof(int32_t)) != 0) return OCCD_EINVAL;
    }
    return launch_b(a, n, dtype, (hipStream_t)stream);
}
