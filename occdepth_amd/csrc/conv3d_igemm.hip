// K2 -- implicit-GEMM 3-D convolution on the CDNA4 fp32 matrix pipe.
//
//   out[vox][co] = act_out( sum_{tap,ci} act_in(in[vox*stride - pad + tap*dil][ci]) * W[co][ci][tap]
//                           + bias[co] + res1[vox][co] + res2[vox][co] )
//
// GEMM view: M = output voxels, N = Cout, K = taps * Cin.  One wavefront owns
// MT x NT tiles of 32(M) x 32(N) and issues v_mfma_f32_32x32x2_f32 (exact fp32,
// 64 FLOP/clk/SIMD).  A workgroup (WM x WN waves) walks K as
//   for kx: for cin-chunk(CK): stage the input slab [YIN][ZIN][CK] of plane
//   x_in = xo*SX - PX + kx*DX into LDS (zero filled outside the volume, act_in
//   applied once per element), then for (ky,kz,kt): A fragments are one
//   ds_read_b128 per lane from the slab (row = voxel, 4 consecutive channels),
//   B fragments one global_load_dwordx4 from the pre-packed L2-resident weights.
// A and B for step s+1 are fetched before the MFMAs of step s (register double
// buffer).  LDS rows are CK+4 floats so 16 consecutive voxel rows hit 16
// different 16-byte bank slots (conflict-free ds_read_b128).
// Output scatter (o_stride/o_off) lets a ConvTranspose3d(k3,s2) run as 8
// sub-pixel phase convolutions writing interleaved voxels.
//
// Reference semantics replaced: occdepth/models/DDR.py:111-139,
// occdepth/models/modules.py:40-46,158-175,278-296, occdepth/models/CRP3D.py:54-97.
#include "common.h"

#include <algorithm>
#include <cstddef>
#include <cstdlib>
#include <cstring>

using occd::FastDiv;

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

// What differs between the sub-pixel phases of one transposed convolution (occd_conv3d_fwd_phases): the tap subset (its
// packed weights and extent), where the phase's voxels land in the output, and the staged slab that follows from the extent.
// A plain launch is the one-phase case.
struct PhaseP {
    const float* wpk;
    int KX, KY, KZ, oox, ooy, ooz, YIN, ZIN;
    FastDiv div_zin;
};
constexpr int kMaxPhases = 8;

struct ConvP {
    const float* in;
    const float* bias;
    const float* res1;
    const float* res2;
    float* out;
    int X, Y, Z, cin8, in_cs, in_coff;
    int KTtot, NTtot;
    int out_cs, out_coff, res1_cs, res1_coff, res2_cs, res2_coff;
    int SX, SY, SZ, DX, DY, DZ, PX, PY, PZ;
    int Xo, Yo, Zo, OX, OY, OZ, osx, osy, osz;
    int act_in, act_out, cout_store;
    int TY, TZ, ytiles, ztiles, nwg;
    int nph_log2;           // blockIdx.y = (batch index << nph_log2) | phase, or (ph_fast) linear id = (tile << nph_log2) | phase
    int ph_fast;
    FastDiv div_tz, div_ztiles, div_ytiles;
    PhaseP ph[kMaxPhases];
};

__device__ __forceinline__ f32x4 apply_act(f32x4 v, int act) {
    if (act == OCCD_ACT_RELU) {
        v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
    } else if (act == OCCD_ACT_SIGMOID) {
        v.x = 1.f / (1.f + expf(-v.x)); v.y = 1.f / (1.f + expf(-v.y));
        v.z = 1.f / (1.f + expf(-v.z)); v.w = 1.f / (1.f + expf(-v.w));
    }
    return v;
}

// KS > 1: in-workgroup split-K.  The KS wave groups take interleaved cin chunks (each with its own LDS slab),
// and their accumulators are summed through LDS before the epilogue.  It multiplies the waves per SIMD for
// layers with few output tiles and a long K (the 32x32x4 ASPP / CRP level: 4096 voxels, K = 27 x 256).
template <int MT, int NT, int WM, int WN, int CK, int KS = 1>
__global__ void __launch_bounds__(WM* WN* KS * 64) conv3d_igemm_kernel(const ConvP p) {
    constexpr int NTH = WM * WN * 64;  // threads of one K group
    constexpr int RS4 = CK / 4 + 1;    // LDS row stride in float4: odd number of 16-B slots
    constexpr int C4 = CK / 4;
    extern __shared__ __attribute__((aligned(16))) f32x4 lds_all[];

    const int lane = threadIdx.x & 63;
    const int wave_all = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int kg = wave_all / (WM * WN);          // K group of this wave
    const int wave = wave_all - kg * (WM * WN);
    const int tid = threadIdx.x - kg * NTH;       // thread index inside the K group
    const int wm = wave / WN;
    const int wn = wave - wm * WN;
    const int li = lane & 31;
    const int kk = lane >> 5;

    // XCD-aware bijective remap: the dispatcher round-robins workgroups over
    // the 8 XCDs; give each XCD a contiguous run of tiles so the kx halo
    // (the same input planes are used by 3 neighbouring xo) hits its own L2.
    uint32_t bid = blockIdx.x;
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
    const PhaseP& ph = p.ph[ph_i];
    const int nt0 = (blockIdx.z * WN + wn) * NT;

    // LDS float4 index of this lane's A row for each M tile.
    int rowbase[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const uint32_t m = (wm * MT + mt) * 32 + li;
        uint32_t yl = occd_fastdiv(m, p.div_tz);
        uint32_t zl = m - yl * p.TZ;
        const bool in_tile = yl < (uint32_t)p.TY;
        yl = in_tile ? yl : 0u;
        zl = in_tile ? zl : 0u;
        rowbase[mt] = (int)((yl * p.SY) * ph.ZIN + zl * p.SZ) * RS4 + kk;
    }
    // float offset of each owned N tile inside one (tap, kt) weight record;
    // N tiles past the layer's last one alias it (their results are never stored).
    int wofs[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) wofs[nt] = min(nt0 + nt, p.NTtot - 1) * 256;

    f32x16 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;

    const int y_in0 = yt * p.TY * p.SY - p.PY;
    const int z_in0 = zt * p.TZ * p.SZ - p.PZ;
    const size_t w_step = (size_t)p.NTtot * 256;  // floats per (tap, kt)
    const float* const wlane = ph.wpk + lane * 4;
    const int rows = ph.YIN * ph.ZIN;
    const int F = rows * C4;
    f32x4* const slab4 = lds_all + (size_t)kg * rows * RS4;   // this K group's slab
    const int n_chunks = (p.cin8 + CK - 1) / CK;
    const int chunk_iters = (n_chunks + KS - 1) / KS;

#define OCCD_MFMA_BLOCK()                                                                              \
    _Pragma("unroll") for (int q = 0; q < 4; ++q) _Pragma("unroll") for (int nt = 0; nt < NT; ++nt)   \
        _Pragma("unroll") for (int mt = 0; mt < MT; ++mt) acc[mt][nt] =                               \
            __builtin_amdgcn_mfma_f32_32x32x2f32(b_cur[nt][q], a_cur[mt][q], acc[mt][nt], 0, 0, 0)

    for (int kx = 0; kx < ph.KX; ++kx) {
        const int xi = xo * p.SX - p.PX + kx * p.DX;
        if (xi < 0 || xi >= p.X) continue;  // workgroup-uniform
        const float* const in_plane =
            p.in + ((size_t)(b * p.X + xi) * p.Y) * p.Z * p.in_cs + p.in_coff;
        for (int ci = 0; ci < chunk_iters; ++ci) {
            const int c0 = (ci * KS + kg) * CK;
            const bool active = c0 < p.cin8;      // uniform per K group; every group still joins the barriers
            const int ck = active ? min(CK, p.cin8 - c0) : 8;
            const int ktn = ck >> 3;
            const int S = active ? ph.KY * ph.KZ * ktn : 0;
            const float* wp = wlane + ((size_t)(kx * ph.KY * ph.KZ) * p.KTtot + (c0 >> 3)) * w_step;

            // first B fragments fly while the slab is staged
            f32x4 b_cur[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                b_cur[nt] = active ? *(const f32x4*)(wp + wofs[nt]) : f32x4{0.f, 0.f, 0.f, 0.f};

            __syncthreads();  // previous slab fully consumed
            for (int f0 = 0; active && f0 < F; f0 += NTH * 4) {
                f32x4 v[4];
                int dst[4];
                bool okv[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int f = f0 + u * NTH + tid;
                    const int fc = min(f, F - 1);
                    const uint32_t row = (uint32_t)fc / C4;
                    const int c4 = fc - (int)row * C4;
                    const uint32_t yi = occd_fastdiv(row, ph.div_zin);
                    const int zi = (int)row - (int)yi * ph.ZIN;
                    const int y = y_in0 + (int)yi, z = z_in0 + zi;
                    okv[u] = f < F && c4 * 4 < ck && (unsigned)y < (unsigned)p.Y && (unsigned)z < (unsigned)p.Z;
                    const int yc = min(max(y, 0), p.Y - 1), zc = min(max(z, 0), p.Z - 1);
                    const int cc = min(c4 * 4, ck - 4);
                    v[u] = *(const f32x4*)(in_plane + ((size_t)yc * p.Z + zc) * p.in_cs + c0 + cc);
                    dst[u] = f < F ? (int)row * RS4 + c4 : -1;
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const f32x4 a = apply_act(v[u], p.act_in);
                    if (dst[u] >= 0) slab4[dst[u]] = okv[u] ? a : f32x4{0.f, 0.f, 0.f, 0.f};
                }
            }
            __syncthreads();

            int ky = 0, kz = 0, ktl = 0;
            int lds_off = 0;  // float4 units
            f32x4 a_cur[MT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) a_cur[mt] = slab4[rowbase[mt]];

            for (int s = 0; s < S - 1; ++s) {
                // advance (ky, kz, kt): scalar bookkeeping only
                ++ktl;
                lds_off += 2;
                wp += w_step;
                if (ktl == ktn) {
                    ktl = 0;
                    wp += (size_t)(p.KTtot - ktn) * w_step;
                    if (++kz == ph.KZ) { kz = 0; ++ky; }
                    lds_off = (ky * p.DY * ph.ZIN + kz * p.DZ) * RS4;
                }
                f32x4 a_nxt[MT], b_nxt[NT];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) a_nxt[mt] = slab4[rowbase[mt] + lds_off];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) b_nxt[nt] = *(const f32x4*)(wp + wofs[nt]);
                OCCD_MFMA_BLOCK();
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) a_cur[mt] = a_nxt[mt];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) b_cur[nt] = b_nxt[nt];
            }
            if (active) { OCCD_MFMA_BLOCK(); }
        }
    }
#undef OCCD_MFMA_BLOCK

    if (KS > 1) {
        // sum the K groups' accumulators through LDS (slabs are dead after the barrier), group 0 stores
        float* red = reinterpret_cast<float*>(lds_all);
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

    // ---------------- epilogue: bias + residuals + activation, channels-last store.
    // The MFMA operands are (weights, activations), i.e. D = W^T . X^T: lane -> voxel `li` of the M tile and the
    // 16 accumulator registers -> couts (r & 3) + 8 (r >> 2) + 4 kk, so every lane owns four float4 groups of
    // consecutive output channels of ONE voxel: 16-byte residual loads and stores instead of 4-byte ones.
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
                const int c = (nt0 + nt) * 32 + 8 * g + 4 * kk;
                if (ok && c < p.cout_store) {
                    f32x4 v = {acc[mt][nt][4 * g], acc[mt][nt][4 * g + 1], acc[mt][nt][4 * g + 2],
                               acc[mt][nt][4 * g + 3]};
                    if (p.bias != nullptr) v += *(const f32x4*)(p.bias + c);
                    if (p.act_out == OCCD_ACT_RELU_PRE) {
                        v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
                    }
                    if (p.res1 != nullptr) v += *(const f32x4*)(p.res1 + vox * p.res1_cs + p.res1_coff + c);
                    if (p.res2 != nullptr) v += *(const f32x4*)(p.res2 + vox * p.res2_cs + p.res2_coff + c);
                    if (p.act_out == OCCD_ACT_RELU) {
                        v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
                    }
                    *(f32x4*)(p.out + vox * p.out_cs + p.out_coff + c) = v;
                }
            }
        }
    }
}

// ---------------------------------------------------------------- weight packing
// layout 3 (occd_pack_weights_gather): element (co, ci, tap) of the packed operator is w[co * s_co + ci * s_ci + ofs[tap]] --
// any transposed / flipped / tap-subset view of a dense weight tensor without materialising it (the data-gradient phases)
__global__ void pack_weights_kernel(const float* __restrict__ w, const float* __restrict__ scale,
                                    float* __restrict__ wpk, int cout, int cin, int taps, int KT, int NT,
                                    int layout, long total, const occd::TapMap tm) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int q = i & 3;
    const int lane = (i >> 2) & 63;
    long t = i >> 8;
    const int nt = t % NT; t /= NT;
    const int kt = t % KT; t /= KT;
    const int tap = (int)t;
    const int co = nt * 32 + (lane & 31);
    const int ci = kt * 8 + (lane >> 5) * 4 + q;
    float v = 0.f;
    if (co < cout && ci < cin) {
        if (layout == 0) v = w[((size_t)co * cin + ci) * taps + tap];
        else if (layout == 1) v = w[((size_t)ci * cout + co) * taps + tap];
        else if (layout == 2) v = w[(size_t)ci * cout + co];
        else v = w[(size_t)co * tm.s_co + (size_t)ci * tm.s_ci + tm.ofs[tap]];
        if (scale != nullptr) v *= scale[co];
    }
    wpk[i] = v;
}

struct Variant {
    int MT, NT, WM, WN, CK, KS;
    void (*kern)(const ConvP);
};

#define OCCD_VARIANT(MT, NT, WM, WN, CK) \
    Variant { MT, NT, WM, WN, CK, 1, conv3d_igemm_kernel<MT, NT, WM, WN, CK, 1> }
#define OCCD_VARIANT_KS(MT, NT, WM, WN, CK, KS) \
    Variant { MT, NT, WM, WN, CK, KS, conv3d_igemm_kernel<MT, NT, WM, WN, CK, KS> }

const Variant kVariants[] = {
    OCCD_VARIANT(2, 1, 4, 1, 32),  // 0: M256 x N32   (head, bottleneck mids)
    OCCD_VARIANT(2, 2, 4, 1, 32),  // 1: M256 x N64
    OCCD_VARIANT(2, 2, 2, 2, 32),  // 2: M128 x N128
    OCCD_VARIANT(1, 2, 2, 2, 32),  // 3: M64  x N128
    OCCD_VARIANT(1, 1, 4, 1, 32),  // 4: M128 x N32
    OCCD_VARIANT(1, 1, 2, 2, 32),  // 5: M64  x N64
    OCCD_VARIANT(1, 1, 1, 4, 32),  // 6: M32  x N128
    OCCD_VARIANT_KS(1, 1, 1, 4, 32, 4),  // 7: M32 x N128, 4-way in-workgroup split-K (16 waves)
    OCCD_VARIANT_KS(1, 1, 2, 2, 32, 2),  // 8: M64 x N64, 2-way split-K (8 waves)
};
constexpr int kNumVariants = sizeof(kVariants) / sizeof(kVariants[0]);
constexpr size_t kMaxLds = 160 * 1024;

struct Tiling {
    int TY, TZ, YIN, ZIN, ytiles, ztiles, ngroups;
    size_t lds;
    long nwg;
};

bool plan(const occd_conv3d_args* a, const Variant& v, int NTtot, Tiling* t) {
    const int mwg = v.MT * v.WM * 32;
    if (a->Zo <= mwg) {
        t->TZ = a->Zo;
        t->TY = mwg / a->Zo;
        if (t->TY > a->Yo) t->TY = a->Yo;
    } else {
        t->TZ = mwg;
        t->TY = 1;
    }
    t->YIN = (t->TY - 1) * a->sy + (a->ky - 1) * a->dy + 1;
    t->ZIN = (t->TZ - 1) * a->sz + (a->kz - 1) * a->dz + 1;
    t->ytiles = (a->Yo + t->TY - 1) / t->TY;
    t->ztiles = (a->Zo + t->TZ - 1) / t->TZ;
    t->lds = (size_t)t->YIN * t->ZIN * (v.CK + 4) * sizeof(float) * v.KS;
    const size_t red = (size_t)(v.KS - 1) * v.WM * v.WN * v.MT * v.NT * 4096;   // split-K reduction scratch
    if (red > t->lds) t->lds = red;
    const int nwg_n = v.NT * v.WN;
    t->ngroups = (NTtot + nwg_n - 1) / nwg_n;
    t->nwg = (long)a->Xo * t->ytiles * t->ztiles;
    return t->lds <= kMaxLds && t->YIN * t->ZIN < 65536 && t->nwg < (1L << 24);
}

}  // namespace

extern "C" int64_t occd_packed_weight_floats(int32_t cout, int32_t cin, int32_t taps) {
    if (cout <= 0 || cin <= 0 || taps <= 0) return OCCD_EINVAL;
    const int64_t KT = (cin + 7) / 8, NT = (cout + 31) / 32;
    return (int64_t)taps * KT * NT * 256;
}

extern "C" int occd_pack_weights(const float* w, const float* scale, float* wpk, int32_t cout, int32_t cin,
                                 int32_t kx, int32_t ky, int32_t kz, int32_t layout, void* stream) {
    if (!w || !wpk || layout < 0 || layout > 2) return OCCD_EINVAL;
    const int taps = kx * ky * kz;
    const int64_t total = occd_packed_weight_floats(cout, cin, taps);
    if (total <= 0 || (layout == 2 && taps != 1)) return OCCD_EINVAL;
    const int KT = (cin + 7) / 8, NT = (cout + 31) / 32;
    const int th = 256;
    const long blocks = (total + th - 1) / th;
    occd::ProfScope prof("pack_weights", (hipStream_t)stream, 0.0, (double)total * 8);
    hipLaunchKernelGGL(pack_weights_kernel, dim3((unsigned)blocks), dim3(th), 0, (hipStream_t)stream, w, scale,
                       wpk, cout, cin, taps, KT, NT, layout, (long)total, occd::TapMap{});
    return occd::check_launch();
}

extern "C" int occd_pack_weights_gather(const float* w, const float* scale, float* wpk, int32_t cout, int32_t cin,
                                        int32_t ntaps, int64_t s_co, int64_t s_ci, const int32_t* tap_ofs, void* stream) {
    if (!w || !wpk || !tap_ofs || ntaps <= 0 || ntaps > occd::kMaxTaps) return OCCD_EINVAL;
    const int64_t total = occd_packed_weight_floats(cout, cin, ntaps);
    if (total <= 0) return OCCD_EINVAL;
    occd::TapMap tm{};
    tm.s_co = s_co; tm.s_ci = s_ci;
    for (int i = 0; i < ntaps; ++i) tm.ofs[i] = tap_ofs[i];
    const int KT = (cin + 7) / 8, NT = (cout + 31) / 32;
    const int th = 256;
    const long blocks = (total + th - 1) / th;
    occd::ProfScope prof("pack_weights", (hipStream_t)stream, 0.0, (double)total * 8);
    hipLaunchKernelGGL(pack_weights_kernel, dim3((unsigned)blocks), dim3(th), 0, (hipStream_t)stream, w, scale,
                       wpk, cout, cin, ntaps, KT, NT, 3, (long)total, tm);
    return occd::check_launch();
}

namespace {

int validate(const occd_conv3d_args* a) {
    if (!a || !a->in || !a->wpk || !a->out) return OCCD_EINVAL;
    if (a->batch <= 0 || a->X <= 0 || a->Y <= 0 || a->Z <= 0 || a->cin <= 0 || a->cout <= 0) return OCCD_EINVAL;
    if (a->kx <= 0 || a->ky <= 0 || a->kz <= 0 || a->sx <= 0 || a->sy <= 0 || a->sz <= 0) return OCCD_EINVAL;
    if (a->Xo <= 0 || a->Yo <= 0 || a->Zo <= 0) return OCCD_EINVAL;
    const int cin8 = (a->cin + 7) & ~7;
    const int NTtot = (a->cout + 31) / 32;
    // every staged float4 must be 16-byte aligned and inside the row
    if ((a->in_cs & 3) || (a->in_coff & 3) || a->in_coff + cin8 > a->in_cs) return OCCD_EINVAL;
    if ((reinterpret_cast<uintptr_t>(a->in) & 15) || (reinterpret_cast<uintptr_t>(a->wpk) & 15)) return OCCD_EINVAL;
    if (a->cout_store < a->cout || a->cout_store > NTtot * 32 || a->out_coff + a->cout_store > a->out_cs)
        return OCCD_EINVAL;
    // float4 epilogue: rows, slices and the stored width are multiples of 4 floats, buffers 16-byte aligned
    if ((a->cout_store & 3) || (a->out_cs & 3) || (a->out_coff & 3) || (reinterpret_cast<uintptr_t>(a->out) & 15))
        return OCCD_EINVAL;
    if (a->res1 && ((a->res1_cs & 3) || (a->res1_coff & 3) || (reinterpret_cast<uintptr_t>(a->res1) & 15)))
        return OCCD_EINVAL;
    if (a->res2 && ((a->res2_cs & 3) || (a->res2_coff & 3) || (reinterpret_cast<uintptr_t>(a->res2) & 15)))
        return OCCD_EINVAL;
    if (a->bias && (reinterpret_cast<uintptr_t>(a->bias) & 15)) return OCCD_EINVAL;
    if (a->res1 && a->res1_coff + a->cout_store > a->res1_cs) return OCCD_EINVAL;
    if (a->res2 && a->res2_coff + a->cout_store > a->res2_cs) return OCCD_EINVAL;
    if ((a->Xo - 1) * a->o_stride_x + a->o_off_x >= a->OX || (a->Yo - 1) * a->o_stride_y + a->o_off_y >= a->OY ||
        (a->Zo - 1) * a->o_stride_z + a->o_off_z >= a->OZ)
        return OCCD_EINVAL;
    if (a->act_out != OCCD_ACT_NONE && a->act_out != OCCD_ACT_RELU && a->act_out != OCCD_ACT_RELU_PRE)
        return OCCD_EINVAL;
    return OCCD_OK;
}

// ---- variant choice: widest N tile the layer fills, then the largest M tile that fits LDS and still yields >= 2
// workgroups per CU.  `copies` = launches' worth of workgroups that share the grid (batch x phases).
int choose(const occd_conv3d_args* a, long copies, Tiling* til) {
    const int cin8 = (a->cin + 7) & ~7;
    const int NTtot = (a->cout + 31) / 32;
    int order[kNumVariants];
    int n = 0;
    if (a->tile_hint > 0 && a->tile_hint <= kNumVariants) {
        order[n++] = a->tile_hint - 1;
    } else if (NTtot == 1) {
        order[n++] = 0; order[n++] = 4; order[n++] = 5; order[n++] = 6;
    } else if (NTtot == 2) {
        order[n++] = 1; order[n++] = 5; order[n++] = 6;
    } else {
        order[n++] = 2; order[n++] = 3; order[n++] = 6;
    }
    int pick = -1;
    for (int i = 0; i < n; ++i) {
        Tiling t{};
        if (!plan(a, kVariants[order[i]], NTtot, &t)) continue;
        pick = order[i];
        *til = t;
        if (t.nwg * t.ngroups * copies >= 512) break;  // else keep refining to the finest fit
    }
    if (pick < 0) return -1;
    if (a->tile_hint == 0 && til->nwg * til->ngroups * copies <= 320 && cin8 >= 128) {
        // too few output tiles to fill 1024 SIMDs with one wave each: multiply the waves by splitting K
        for (int cand : {7, 8}) {
            Tiling t{};
            const int nwn = kVariants[cand].NT * kVariants[cand].WN;
            if (NTtot % nwn != 0 && NTtot > nwn) continue;
            if (!plan(a, kVariants[cand], NTtot, &t)) continue;
            if (t.nwg * t.ngroups * copies > 1024) continue;
            pick = cand;
            *til = t;
            break;
        }
    }
    return pick;
}

void fill_shared(const occd_conv3d_args* a, const Tiling& til, ConvP* p) {
    const int cin8 = (a->cin + 7) & ~7;
    p->in = a->in; p->bias = a->bias; p->res1 = a->res1; p->res2 = a->res2; p->out = a->out;
    p->X = a->X; p->Y = a->Y; p->Z = a->Z; p->cin8 = cin8; p->in_cs = a->in_cs; p->in_coff = a->in_coff;
    p->KTtot = cin8 / 8; p->NTtot = (a->cout + 31) / 32;
    p->out_cs = a->out_cs; p->out_coff = a->out_coff;
    p->res1_cs = a->res1_cs; p->res1_coff = a->res1_coff; p->res2_cs = a->res2_cs; p->res2_coff = a->res2_coff;
    p->SX = a->sx; p->SY = a->sy; p->SZ = a->sz;
    p->DX = a->dx; p->DY = a->dy; p->DZ = a->dz; p->PX = a->px; p->PY = a->py; p->PZ = a->pz;
    p->Xo = a->Xo; p->Yo = a->Yo; p->Zo = a->Zo; p->OX = a->OX; p->OY = a->OY; p->OZ = a->OZ;
    p->osx = a->o_stride_x; p->osy = a->o_stride_y; p->osz = a->o_stride_z;
    p->act_in = a->act_in; p->act_out = a->act_out; p->cout_store = a->cout_store;
    p->TY = til.TY; p->TZ = til.TZ;
    p->ytiles = til.ytiles; p->ztiles = til.ztiles; p->nwg = (int)til.nwg;
    p->div_tz = occd::make_fastdiv(til.TZ);
    p->div_ztiles = occd::make_fastdiv(til.ztiles); p->div_ytiles = occd::make_fastdiv(til.ytiles);
}

void fill_phase(const occd_conv3d_args* a, const Tiling& til, PhaseP* ph) {
    ph->wpk = a->wpk;
    ph->KX = a->kx; ph->KY = a->ky; ph->KZ = a->kz;
    ph->oox = a->o_off_x; ph->ooy = a->o_off_y; ph->ooz = a->o_off_z;
    ph->YIN = til.YIN; ph->ZIN = til.ZIN;
    ph->div_zin = occd::make_fastdiv(til.ZIN);
}

void cost(const occd_conv3d_args* a, double* flops, double* bytes, bool first) {
    const double taps = (double)a->kx * a->ky * a->kz;
    const double pos = (double)a->batch * a->Xo * a->Yo * a->Zo;
    *flops += 2.0 * pos * taps * a->cin * a->cout;
    *bytes += 4.0 * ((first ? (double)a->batch * a->X * a->Y * a->Z * a->cin : 0.0) +
                     pos * a->cout * (1 + (a->res1 != nullptr) + (a->res2 != nullptr)) + taps * a->cin * a->cout);
}

}  // namespace

extern "C" int occd_conv3d_fwd(const occd_conv3d_args* a, void* stream) {
    const int bad = validate(a);
    if (bad != OCCD_OK) return bad;
    {
        const int taken = occd::try_conv3d_c32_persist(a, (hipStream_t)stream);
        if (taken != 0) return taken > 0 ? OCCD_OK : taken;
    }
    Tiling til{};
    const int pick = choose(a, a->batch, &til);
    if (pick < 0) return OCCD_ENOMEM;
    const Variant& v = kVariants[pick];
    ConvP p{};
    fill_shared(a, til, &p);
    fill_phase(a, til, &p.ph[0]);
    p.nph_log2 = 0;
    p.ph_fast = 0;
    if (til.lds > 64 * 1024 && occd::ensure_big_lds(reinterpret_cast<const void*>(v.kern)) != OCCD_OK) return OCCD_ELAUNCH;
    double flops = 0.0, bytes = 0.0;
    cost(a, &flops, &bytes, true);
    occd::ProfScope prof("conv3d_igemm", (hipStream_t)stream, flops, bytes);
    hipLaunchKernelGGL(v.kern, dim3((unsigned)til.nwg, (unsigned)a->batch, (unsigned)til.ngroups),
                       dim3(v.WM * v.WN * v.KS * 64), til.lds, (hipStream_t)stream, p);
    return occd::check_launch();
}

// The sub-pixel phases of ONE transposed convolution (ConvTranspose3d(k3, s2, p1, op1) = 8 phase convolutions over tap
// subsets that scatter to interleaved voxels, occdepth/models/modules.py:278-296) as ONE launch: the phase is the low bits
// of blockIdx.y, every phase walks the same output tiling of the (shared) input volume, and the heaviest phase comes first in
// dispatch order so that the single-tap ones fill the tail.  a[0..n): n in {1, 2, 4, 8} descriptors that may differ ONLY
// in wpk, kx / ky / kz and o_off_*; anything else must match a[0].  Replaces n launches of occd_conv3d_fwd (24 per frame
// at config 2, the smallest 16 us long).
extern "C" int occd_conv3d_fwd_phases(const occd_conv3d_args* a, int32_t n, void* stream) {
    if (!a || (n != 1 && n != 2 && n != 4 && n != 8)) return OCCD_EINVAL;
    int heavy = 0;
    for (int i = 0; i < n; ++i) {
        const int bad = validate(a + i);
        if (bad != OCCD_OK) return bad;
        occd_conv3d_args t = a[i];
        t.wpk = a[0].wpk; t.kx = a[0].kx; t.ky = a[0].ky; t.kz = a[0].kz;
        t.o_off_x = a[0].o_off_x; t.o_off_y = a[0].o_off_y; t.o_off_z = a[0].o_off_z;
        if (memcmp(&t, &a[0], offsetof(occd_conv3d_args, tile_hint) + sizeof(int32_t)) != 0) return OCCD_EINVAL;
        if (a[i].kx * a[i].ky * a[i].kz > a[heavy].kx * a[heavy].ky * a[heavy].kz) heavy = i;
    }
    if (a[0].cin <= 32 && n > 1) {             // (the <= 32-channel sliding-window kernels are single-phase: issue them one by one)
        for (int i = 0; i < n; ++i) {
            const int rc = occd_conv3d_fwd(a + i, stream);
            if (rc != OCCD_OK) return rc;
        }
        return OCCD_OK;
    }
    // one variant / output tiling for all phases: the one the heaviest phase would take with n x batch copies in the grid
    occd_conv3d_args big = a[heavy];
    for (int i = 0; i < n; ++i) {               // (slab extents: the largest ky / kz over the phases)
        big.ky = a[i].ky > big.ky ? a[i].ky : big.ky;
        big.kz = a[i].kz > big.kz ? a[i].kz : big.kz;
    }
    Tiling til{};
    const int pick = choose(&big, (long)a[0].batch * n, &til);
    if (pick < 0) return OCCD_ENOMEM;
    const Variant& v = kVariants[pick];
    const int NTtot = (a[0].cout + 31) / 32;
    ConvP p{};
    fill_shared(&a[0], til, &p);
    int order[kMaxPhases];
    for (int i = 0; i < n; ++i) order[i] = i;
    std::stable_sort(order, order + n, [&](int l, int r) {
        return a[l].kx * a[l].ky * a[l].kz > a[r].kx * a[r].ky * a[r].kz;
    });
    double flops = 0.0, bytes = 0.0;
    for (int i = 0; i < n; ++i) {
        const occd_conv3d_args* ai = a + order[i];
        Tiling ti{};
        if (!plan(ai, v, NTtot, &ti) || ti.TY != til.TY || ti.TZ != til.TZ || ti.lds > til.lds) return OCCD_EINVAL;
        fill_phase(ai, ti, &p.ph[i]);
        cost(ai, &flops, &bytes, i == 0);
    }
    p.nph_log2 = n == 1 ? 0 : n == 2 ? 1 : n == 4 ? 2 : 3;
    // phase-major dispatch (all tiles of the heaviest phase first) is the default; the phases of a tile next to each other on
    // one XCD (OCCD_PHASE_FAST=1) measured SLOWER: 64 -> 32 at 128x128x16 0.47 against 0.35 ms on K2, 0.36 against 0.34 on K2b
    static const bool phase_fast = occd::env_flag("OCCD_PHASE_FAST", false);
    p.ph_fast = phase_fast ? 1 : 0;
    if (p.ph_fast) p.nwg = (int)(til.nwg * n);
    if ((long)a[0].batch * n > 65535 || til.nwg * n >= (1L << 24)) return OCCD_EINVAL;
    if (til.lds > 64 * 1024 && occd::ensure_big_lds(reinterpret_cast<const void*>(v.kern)) != OCCD_OK) return OCCD_ELAUNCH;
    occd::ProfScope prof("conv3d_igemm_phases", (hipStream_t)stream, flops, bytes);
    hipLaunchKernelGGL(v.kern, p.ph_fast ? dim3((unsigned)(til.nwg * n), (unsigned)a[0].batch, (unsigned)til.ngroups)
                                         : dim3((unsigned)til.nwg, (unsigned)(a[0].batch * n), (unsigned)til.ngroups),
                       dim3(v.WM * v.WN * v.KS * 64), til.lds, (hipStream_t)stream, p);
    return occd::check_launch();
}
