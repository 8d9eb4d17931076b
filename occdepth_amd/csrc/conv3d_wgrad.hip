// K8 -- weight gradient of a 3-D convolution on fp32 MFMA ("dgrad/wgrad igemm for conv3d").
//
//   dW[co][ci][tap] = sum_{b, xo, yo, zo} gy[b, xo, yo, zo][co] * x[b, xo*s - p + kx*d, ..][ci]
//
// is, per tap, a (Cout x voxels) . (voxels x Cin) GEMM whose reduction dimension is the VOXEL index.  Both tensors
// are channels-last, so for v_mfma_f32_32x32x2_f32 the A operand (M = 32 couts, K = 2 voxels) and the B operand
// (N = 32 cins, K = 2 voxels) are plain coalesced row reads: lane&31 = channel, lane>>5 = voxel parity -- no LDS
// staging at all.  A gy fragment is loaded once and reused for all the taps a wave owns; the shifted x fragments
// come back from L1/L2 (each x element is touched by up to 27 (voxel, tap) pairs).
// Work split: workgroup = (chunk of output rows, cout tile, cin tile), 8 waves.
//   >= 8 taps: the waves split the TAPS (wave w owns taps w, w+8, ...; <= 4 accumulators of 16 registers);
//   <  8 taps: the waves split the ROWS and each owns every tap (<= 7 accumulators).
// The loads of step s+1 (8 output voxels) are issued before the MFMAs of step s (two register sets).
// Each wave writes its partial (tap, 32, 32) tiles to a workspace; a second kernel sums the partials in a fixed
// order, so the result is deterministic (no float atomics).
#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int kTPW = 7;            // accumulator sets per wave in the row-split mode; 4 when the 8 waves split the taps

struct WgradP {
    const float* x;
    const float* gy;
    float* ws;
    int batch, X, Y, Z, cin, x_cs, x_coff;
    int Xo, Yo, Zo, cout, gy_cs, gy_coff;
    int kx, ky, kz, sx, sy, sz, dx, dy, dz, px, py, pz;
    int ntaps, rows, rows_per_chunk, split_taps, cot, cit, slots;
};

// TPW accumulator sets per wave; SPLIT: the 8 waves split the taps (wave w owns taps w, w+8, ...), else the rows.
template <int TPW, bool SPLIT>
__global__ void __launch_bounds__(512) wgrad_kernel(const WgradP p) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int l = lane & 31, kk = lane >> 5;
    const int chunk = blockIdx.x, co0 = blockIdx.y * 32, ci0 = blockIdx.z * 32;
    const bool co_ok = co0 + l < p.cout, ci_ok = ci0 + l < p.cin;

    // taps owned by this wave, decoded once
    int n_mine = 0;
    int ox[TPW], oy[TPW], oz[TPW], tap_id[TPW];
#pragma unroll
    for (int i = 0; i < TPW; ++i) {
        const int t = SPLIT ? wave + 8 * i : i;
        const bool live = t < p.ntaps;
        const int tt = live ? t : 0;
        const int a = tt / (p.ky * p.kz), r = tt - a * (p.ky * p.kz);
        const int b = r / p.kz, c = r - b * p.kz;
        ox[i] = a * p.dx - p.px;
        oy[i] = b * p.dy - p.py;
        oz[i] = c * p.dz - p.pz;
        tap_id[i] = tt;
        if (live) n_mine = i + 1;
    }
    f32x16 acc[TPW];
#pragma unroll
    for (int i = 0; i < TPW; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

    const int r_begin = chunk * p.rows_per_chunk;
    const int r_end = min(r_begin + p.rows_per_chunk, p.rows);
    const int r_first = r_begin + (SPLIT ? 0 : wave), r_step = SPLIT ? 1 : 8;
    const int nzb = (p.Zo + 7) >> 3;
    const int n_rows = r_first < r_end ? (r_end - r_first + r_step - 1) / r_step : 0;
    const int n_steps = n_rows * nzb;                       // one step = 8 output voxels of one row

    // fragments of one step: A = gy (32 couts x 8 voxels), B[i] = x shifted by tap i (32 cins x 8 voxels).
    // Every row (Z voxels x cs floats) is addressed through its own BUFFER descriptor: the hardware range check
    // returns 0 for z < 0 (the 32-bit offset wraps), z >= Z, invalid channel lanes (offset pushed past 2^31) and
    // padding planes (0 records) -- no per-load compare / select, one v_add per load.
    const unsigned kOob = 0x80000000u;
    const unsigned la = co_ok ? (unsigned)((kk * p.gy_cs + l) * 4) : kOob;
    const unsigned lb = ci_ok ? (unsigned)((kk * p.sz * p.x_cs + l) * 4) : kOob;
    const unsigned gy_records = (unsigned)((p.Zo * p.gy_cs - p.gy_coff - co0) * 4);
    const unsigned x_records = (unsigned)((p.Z * p.x_cs - p.x_coff - ci0) * 4);
    // scalar bookkeeping kept incremental (no divisions, no 64-bit multiplies in the loop): the walk position ...
    int w_zb = 0, w_row = r_first;
    int w_yo = r_first % p.Yo, w_xo = (r_first / p.Yo) % p.Xo, w_b = r_first / p.Yo / p.Xo;
    const long x_row_bytes = (long)p.Z * p.x_cs * 4;
    long tap_delta[TPW];                                     // ... and the byte offset of tap i's plane row
    int tap_zoff[TPW];
#pragma unroll
    for (int i = 0; i < TPW; ++i) {
        tap_delta[i] = ((long)ox[i] * p.Y + oy[i]) * x_row_bytes;
        tap_zoff[i] = oz[i] * p.x_cs * 4;
    }
    const char* const x_base = (const char*)(p.x + p.x_coff + ci0);
    const char* const gy_base = (const char*)(p.gy + p.gy_coff + co0);
    const int gy_zstep = 2 * p.gy_cs * 4, x_zstep = 2 * p.sz * p.x_cs * 4;
    auto fetch = [&](float (&a)[4], float (&bv)[TPW][4]) {
        const int z0 = w_zb << 3;
        const auto ra = __builtin_amdgcn_make_buffer_rsrc((void*)(gy_base + (long)w_row * p.Zo * p.gy_cs * 4), 0,
                                                          gy_records, 0x00020000);
        const int a_z0 = z0 * p.gy_cs * 4;
#pragma unroll
        for (int q = 0; q < 4; ++q)
            a[q] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ra, la + (unsigned)(a_z0 + q * gy_zstep), 0, 0));
        const int xb = w_xo * p.sx, yb = w_yo * p.sy;
        const char* centre = x_base + (((long)w_b * p.X + xb) * p.Y + yb) * x_row_bytes;
        const int b_z0 = z0 * p.sz * p.x_cs * 4;
#pragma unroll
        for (int i = 0; i < TPW; ++i) {
            const int xi = xb + ox[i], yi = yb + oy[i];
            const bool plane_ok = i < n_mine && xi >= 0 && xi < p.X && yi >= 0 && yi < p.Y;
            const auto rb = __builtin_amdgcn_make_buffer_rsrc((void*)(centre + (plane_ok ? tap_delta[i] : 0)), 0,
                                                              plane_ok ? x_records : 0u, 0x00020000);
#pragma unroll
            for (int q = 0; q < 4; ++q)
                bv[i][q] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                    rb, lb + (unsigned)(b_z0 + tap_zoff[i] + q * x_zstep), 0, 0));
        }
        // advance the walk by one step
        if (++w_zb == nzb) {
            w_zb = 0;
            w_row += r_step;
            w_yo += r_step;
            while (w_yo >= p.Yo) {
                w_yo -= p.Yo;
                if (++w_xo == p.Xo) { w_xo = 0; ++w_b; }
            }
        }
    };
    auto mma = [&](const float (&a)[4], const float (&bv)[TPW][4]) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int i = 0; i < TPW; ++i)
                if (i < n_mine) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q], bv[i][q], acc[i], 0, 0, 0);
    };

    // software pipeline, two register sets: the loads of step s+1 are in flight under the MFMAs of step s
    float a0[4], b0[TPW][4], a1[4], b1[TPW][4];
    if (n_steps > 0) fetch(a0, b0);
    for (int s = 0; s < n_steps; s += 2) {
        if (s + 1 < n_steps) fetch(a1, b1);
        __builtin_amdgcn_sched_barrier(0);
        mma(a0, b0);
        __builtin_amdgcn_sched_barrier(0);
        if (s + 1 < n_steps) {
            if (s + 2 < n_steps) fetch(a0, b0);
            __builtin_amdgcn_sched_barrier(0);
            mma(a1, b1);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    // partial tiles -> workspace[slot][cot][cit][tap][co 32][ci 32]; D layout: lane&31 = ci, rows co = (r&3)+8(r>>2)+4kk
    const int slot = SPLIT ? chunk : chunk * 8 + wave;
    float* base = p.ws + ((((size_t)slot * p.cot + blockIdx.y) * p.cit + blockIdx.z) * p.ntaps) * 1024;
#pragma unroll
    for (int i = 0; i < TPW; ++i)
        if (i < n_mine) {
            float* t = base + (size_t)tap_id[i] * 1024;
#pragma unroll
            for (int r = 0; r < 16; ++r) t[((r & 3) + 8 * (r >> 2) + 4 * kk) * 32 + l] = acc[i][r];
        }
}

// dw[co][ci][tap] = sum over slots.  A workgroup owns 32 consecutive workspace elements of every slot (coalesced
// 128-byte reads) and splits the slots over 8 thread groups; each group sums its slots in ascending order and the 8
// partial sums are combined in a fixed order through LDS -> deterministic, and parallel in both directions.
__global__ void __launch_bounds__(256) wgrad_reduce_kernel(const float* ws, float* dw, int slots, int cot, int cit,
                                                           int ntaps, int cout, int cin) {
    __shared__ float part[8][32];
    const size_t slot_stride = (size_t)cot * cit * ntaps * 1024;
    const int e = threadIdx.x & 31, grp = threadIdx.x >> 5;
    const size_t elem = (size_t)blockIdx.x * 32 + e;               // index inside one slot: [cot][cit][tap][co 32][ci 32]
    float s = 0.f;
    if (elem < slot_stride) {
        const int per = (slots + 7) / 8;
        const int k0 = grp * per, k1 = min(k0 + per, slots);
        // 8 loads in flight per thread (the slots are 100+ KB apart: every load is a miss, and a loop of dependent
        // adds around single loads waited one HBM round trip per slot: 98 us per launch); fixed combination order
        const float* wp = ws + elem;
        float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        int k = k0;
        for (; k + 8 <= k1; k += 8) {
#pragma unroll
            for (int u = 0; u < 8; ++u) a[u] += wp[(size_t)(k + u) * slot_stride];
        }
        for (; k < k1; ++k) a[0] += wp[(size_t)k * slot_stride];
        s = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
    }
    part[grp][e] = s;
    __syncthreads();
    if (grp == 0 && elem < slot_stride) {
        float t = part[0][e];
#pragma unroll
        for (int g = 1; g < 8; ++g) t += part[g][e];
        const int ci_l = (int)(elem & 31), co_l = (int)((elem >> 5) & 31);
        const size_t tile = elem >> 10;                            // (cot * cit + cit_i) * ntaps + tap
        const int tap = (int)(tile % ntaps);
        const size_t cc = tile / ntaps;
        const int ci = (int)(cc % cit) * 32 + ci_l, co = (int)(cc / cit) * 32 + co_l;
        if (co < cout && ci < cin) dw[((size_t)co * cin + ci) * ntaps + tap] = t;
    }
}

int plan(const occd_conv3d_wgrad_args* a, WgradP& p) {
    if (a == nullptr || a->x == nullptr || a->gy == nullptr) return OCCD_EINVAL;
    if (a->batch < 1 || a->X < 1 || a->Y < 1 || a->Z < 1 || a->Xo < 1 || a->Yo < 1 || a->Zo < 1) return OCCD_EINVAL;
    if (a->cin < 1 || a->cout < 1 || a->kx < 1 || a->ky < 1 || a->kz < 1) return OCCD_EINVAL;
    if (a->sx < 1 || a->sy < 1 || a->sz < 1 || a->dx < 1 || a->dy < 1 || a->dz < 1) return OCCD_EINVAL;
    if (a->x_coff < 0 || a->gy_coff < 0 || a->x_coff + a->cin > a->x_cs || a->gy_coff + a->cout > a->gy_cs) return OCCD_EINVAL;
    const long ntaps = (long)a->kx * a->ky * a->kz;
    if (ntaps > 32) return OCCD_EINVAL;         // kernels beyond 32 taps are not planned
    p.x = a->x; p.gy = a->gy; p.ws = a->workspace;
    p.batch = a->batch; p.X = a->X; p.Y = a->Y; p.Z = a->Z; p.cin = a->cin; p.x_cs = a->x_cs; p.x_coff = a->x_coff;
    p.Xo = a->Xo; p.Yo = a->Yo; p.Zo = a->Zo; p.cout = a->cout; p.gy_cs = a->gy_cs; p.gy_coff = a->gy_coff;
    p.kx = a->kx; p.ky = a->ky; p.kz = a->kz; p.sx = a->sx; p.sy = a->sy; p.sz = a->sz;
    p.dx = a->dx; p.dy = a->dy; p.dz = a->dz; p.px = a->px; p.py = a->py; p.pz = a->pz;
    p.ntaps = (int)ntaps;
    const long rows = (long)a->batch * a->Xo * a->Yo;
    if (rows > 0x7fffffff) return OCCD_EINVAL;
    p.rows = (int)rows;
    p.split_taps = ntaps > kTPW ? 1 : 0;
    p.cot = (a->cout + 31) / 32;
    p.cit = (a->cin + 31) / 32;
    // ~1024 workgroups over the chip (every chunk costs a set of partial tiles in the workspace); a chunk is at
    // least 8 rows (one per wave in the row-split mode)
    long chunks = 1024 / ((long)p.cot * p.cit);
    if (chunks < 1) chunks = 1;
    long rpc = (rows + chunks - 1) / chunks;
    if (rpc < 8) rpc = 8;
    p.rows_per_chunk = (int)rpc;
    const long nchunks = (rows + rpc - 1) / rpc;
    p.slots = (int)(p.split_taps ? nchunks : nchunks * 8);
    return OCCD_OK;
}


// ================================================================================================================
// K8b -- the same weight gradient on the bf16 matrix pipe (v_mfma_f32_32x32x16_bf16, fp32 accumulate; BASELINE
// configs[3]).  The reduction dimension is the voxel index, and a 16-deep bf16 MFMA wants 8 CONSECUTIVE K values per
// lane: 8 voxels of one channel, i.e. a column of the channels-last tile.  gfx950 has the instruction for exactly
// that: ds_read_b64_tr_b16 (a 16-lane group reads a 4 x 16 block of 16-bit elements row-wise and every lane receives a
// column; semantics pinned on hardware by tools/probe_bf16.hip).  So the tiles are staged ROW-MAJOR (voxel rows of
// 32 channels = 64 B, converted from fp32 to bf16 once, coalesced, no transposition on the way in), a tap shift is a
// ROW offset (always aligned, whatever the dilation), and both MFMA operands come out of LDS as two transposed reads.
// Work unit = (b, xo, yo, z tile of ZT output voxels); a 256-thread workgroup stages the gy tile (ZT x NCO couts) and
// the KX x KY input rows ((ZT - 1) sz + (KZ - 1) dz + 1 voxels x 32 cins) of the unit, then
//   TAPSPLIT (NCO = 32; 27-tap 3-D convolutions): wave w owns taps w, w + 4, ... (<= 7 accumulators); the gy fragment of
//            a K group is read once and feeds every tap of the wave;
//   COSPLIT  (NCO = 128; <= 9 taps, the 2-D decoder's 3x3 convolutions as X = 1 volumes): wave w owns the 32 couts
//            co0 + 32 w and every tap (<= 9 accumulators); the staged input rows serve 128 couts.
// Partial (tap, 32 co, 32 ci) tiles go to the workspace layout of K8 and through the same deterministic reduction.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;

struct WgradBP {
    const void* x;
    const void* gy;
    float* ws;
    int batch, X, Y, Z, cin8, x_cs, x_coff;
    int Xo, Yo, Zo, cout8, gy_cs, gy_coff;
    int kx, ky, kz, sx, sy, sz, dx, dy, dz, px, py, pz;
    int ntaps, ZT, ZIN, ztiles, units, units_per_chunk, cot, cit, grs, pls;
    occd::FastDiv div_zin, div_ztiles, div_yo, div_xo;
};

__device__ __forceinline__ u32x4 wg_pack8(f32x4 a, f32x4 b) {
    bf16x8 r = {(__bf16)a.x, (__bf16)a.y, (__bf16)a.z, (__bf16)a.w, (__bf16)b.x, (__bf16)b.y, (__bf16)b.z, (__bf16)b.w};
    return __builtin_bit_cast(u32x4, r);
}

// 8 channels (16 B of bf16) of one voxel row from fp32 or bf16 storage; `e` = element index of the first channel
template <bool IN_BF16>
__device__ __forceinline__ u32x4 wg_load8(const void* base, size_t e) {
    if (IN_BF16) return *(const u32x4*)((const uint16_t*)base + e);
    const f32x4 lo = *(const f32x4*)((const float*)base + e);
    const f32x4 hi = *(const f32x4*)((const float*)base + e + 4);
    return wg_pack8(lo, hi);
}

__device__ __forceinline__ bf16x8 wg_tr_frag(const unsigned char* p0, int step_bytes) {
    const bf16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)p0);
    const bf16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(p0 + step_bytes));
    return __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
}

template <int TPW, bool COSPLIT, bool IN_BF16>
__global__ void __launch_bounds__(256, 2) wgrad_bf16_kernel(const WgradBP p) {
    constexpr int NCO = COSPLIT ? 128 : 32;
    constexpr int GC8 = NCO / 8;                       // 16-byte chunks per gy row
    extern __shared__ __attribute__((aligned(16))) unsigned char wlds[];
    unsigned char* const gyt = wlds;
    unsigned char* const xt = wlds + (size_t)p.ZT * p.grs;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i16 = lane & 15, g = lane >> 4, h = lane >> 5, l = lane & 31;
    const int chunk = blockIdx.x, cog = blockIdx.y;
    const int co0 = cog * NCO, ci0 = blockIdx.z * 32;

    // owned taps
    int n_mine = 0;
    int xoff[TPW], tap_id[TPW];
#pragma unroll
    for (int i = 0; i < TPW; ++i) {
        const int t = COSPLIT ? i : wave + 4 * i;
        const bool live = t < p.ntaps;
        const int tt = live ? t : 0;
        const int a = tt / (p.ky * p.kz), r = tt - a * (p.ky * p.kz);
        const int bq = r / p.kz, c = r - bq * p.kz;
        xoff[i] = (a * p.ky + bq) * p.pls + c * p.dz * 64;
        tap_id[i] = tt;
        if (live) n_mine = i + 1;
    }
    f32x16 acc[TPW];
#pragma unroll
    for (int i = 0; i < TPW; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

    // lane-constant byte offsets of the transposed reads: row 8 h + (i16 >> 2) (+ 4 for the second read), 4 channels
    // 16 (g & 1) + 4 (i16 & 3) ... of the 32-channel block
    const int a_lane = (8 * h + (i16 >> 2)) * p.grs + ((COSPLIT ? 32 * wave : 0) + 16 * (g & 1) + 4 * (i16 & 3)) * 2;
    const int b_lane = (8 * h + (i16 >> 2)) * p.sz * 64 + (16 * (g & 1) + 4 * (i16 & 3)) * 2;
    const int a_step = 4 * p.grs, b_step = 4 * p.sz * 64;

    const int u_begin = chunk * p.units_per_chunk;
    const int u_end = min(u_begin + p.units_per_chunk, p.units);
    const int n_g = p.ZT * GC8;
    const int n_x = p.kx * p.ky * p.ZIN * 4;
    const int kgroups = p.ZT >> 4;

    for (int u = u_begin; u < u_end; ++u) {
        const uint32_t r1 = occd_fastdiv((uint32_t)u, p.div_ztiles);
        const int zt = u - (int)r1 * p.ztiles;
        const uint32_t r2 = occd_fastdiv(r1, p.div_yo);
        const int yo = (int)r1 - (int)r2 * p.Yo;
        const uint32_t b = occd_fastdiv(r2, p.div_xo);
        const int xo = (int)r2 - (int)b * p.Xo;
        const int z0 = zt * p.ZT;

        __syncthreads();   // the previous unit's tiles are consumed
        {   // gy tile: rows z0 .. z0 + ZT - 1 of output row (b, xo, yo), couts co0 .. co0 + NCO - 1
            const size_t row0 = (((size_t)b * p.Xo + xo) * p.Yo + yo) * p.Zo;
            for (int f = tid; f < n_g; f += 256) {
                const int zl = f / GC8, c8 = f - zl * GC8;
                const int z = z0 + zl, co = co0 + c8 * 8;
                const bool ok = z < p.Zo && co < p.cout8;
                const size_t e = (row0 + (ok ? z : 0)) * p.gy_cs + p.gy_coff + (ok ? co : 0);
                u32x4 v = wg_load8<IN_BF16>(p.gy, e);
                if (!ok) v = u32x4{0u, 0u, 0u, 0u};
                *(u32x4*)(gyt + zl * p.grs + c8 * 16) = v;
            }
        }
        {   // input rows: plane (kx, ky) -> row (xi, yi), voxels z0 sz - pz .. (+ ZIN), cins ci0 .. ci0 + 31
            for (int f = tid; f < n_x; f += 256) {
                const int c8 = f & 3;
                const uint32_t r = (uint32_t)f >> 2;
                const uint32_t pl = occd_fastdiv(r, p.div_zin);
                const int zl = (int)r - (int)pl * p.ZIN;
                const int kxi = (int)pl / p.ky, kyi = (int)pl - kxi * p.ky;
                const int xi = xo * p.sx - p.px + kxi * p.dx;
                const int yi = yo * p.sy - p.py + kyi * p.dy;
                const int z = z0 * p.sz - p.pz + zl;
                const int ci = ci0 + c8 * 8;
                const bool ok = xi >= 0 && xi < p.X && yi >= 0 && yi < p.Y && z >= 0 && z < p.Z && ci < p.cin8;
                const size_t e = ((((size_t)b * p.X + (ok ? xi : 0)) * p.Y + (ok ? yi : 0)) * p.Z + (ok ? z : 0)) * p.x_cs +
                                 p.x_coff + (ok ? ci : 0);
                u32x4 v = wg_load8<IN_BF16>(p.x, e);
                if (!ok) v = u32x4{0u, 0u, 0u, 0u};
                *(u32x4*)(xt + (int)pl * p.pls + zl * 64 + c8 * 16) = v;
            }
        }
        __syncthreads();

        for (int kg = 0; kg < kgroups; ++kg) {
            const bf16x8 a = wg_tr_frag(gyt + a_lane + kg * 16 * p.grs, a_step);
            const unsigned char* xb = xt + b_lane + kg * 16 * p.sz * 64;
#pragma unroll
            for (int i = 0; i < TPW; ++i)
                if (i < n_mine) {
                    const bf16x8 bv = wg_tr_frag(xb + xoff[i], b_step);
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bv, acc[i], 0, 0, 0);
                }
        }
    }

    // partial tiles -> workspace[slot = chunk][cot][cit][tap][co 32][ci 32]; D: lane & 31 = ci, rows co = (r&3)+8(r>>2)+4h
    const int cot_i = COSPLIT ? cog * 4 + wave : cog;
    if (cot_i < p.cot) {
        float* base = p.ws + ((((size_t)chunk * p.cot + cot_i) * p.cit + blockIdx.z) * p.ntaps) * 1024;
#pragma unroll
        for (int i = 0; i < TPW; ++i)
            if (i < n_mine) {
                float* t = base + (size_t)tap_id[i] * 1024;
#pragma unroll
                for (int r = 0; r < 16; ++r) t[((r & 3) + 8 * (r >> 2) + 4 * h) * 32 + l] = acc[i][r];
            }
    }
}

int plan_b(const occd_conv3d_wgrad_args* a, int dtype, WgradBP& p, bool& cosplit, size_t& lds) {
    if (a == nullptr || a->x == nullptr || a->gy == nullptr || dtype < 0 || dtype > 1) return OCCD_EINVAL;
    if (a->batch < 1 || a->X < 1 || a->Y < 1 || a->Z < 1 || a->Xo < 1 || a->Yo < 1 || a->Zo < 1) return OCCD_EINVAL;
    if (a->cin < 1 || a->cout < 1 || a->kx < 1 || a->ky < 1 || a->kz < 1) return OCCD_EINVAL;
    if (a->sx < 1 || a->sy < 1 || a->sz < 1 || a->dx < 1 || a->dy < 1 || a->dz < 1) return OCCD_EINVAL;
    const int al = dtype == 1 ? 7 : 3;
    const int cin8 = (a->cin + 7) & ~7, cout8 = (a->cout + 7) & ~7;
    if (a->x_coff < 0 || a->gy_coff < 0 || a->x_coff + cin8 > a->x_cs || a->gy_coff + cout8 > a->gy_cs) return OCCD_EINVAL;
    if ((a->x_cs & al) || (a->x_coff & al) || (a->gy_cs & al) || (a->gy_coff & al)) return OCCD_EINVAL;
    if ((reinterpret_cast<uintptr_t>(a->x) & 15) || (reinterpret_cast<uintptr_t>(a->gy) & 15)) return OCCD_EINVAL;
    const long ntaps = (long)a->kx * a->ky * a->kz;
    if (ntaps > 28) return OCCD_EINVAL;
    cosplit = ntaps <= 9 && a->cout > 32;
    p.x = a->x; p.gy = a->gy; p.ws = a->workspace;
    p.batch = a->batch; p.X = a->X; p.Y = a->Y; p.Z = a->Z; p.cin8 = cin8; p.x_cs = a->x_cs; p.x_coff = a->x_coff;
    p.Xo = a->Xo; p.Yo = a->Yo; p.Zo = a->Zo; p.cout8 = cout8; p.gy_cs = a->gy_cs; p.gy_coff = a->gy_coff;
    p.kx = a->kx; p.ky = a->ky; p.kz = a->kz; p.sx = a->sx; p.sy = a->sy; p.sz = a->sz;
    p.dx = a->dx; p.dy = a->dy; p.dz = a->dz; p.px = a->px; p.py = a->py; p.pz = a->pz;
    p.ntaps = (int)ntaps;
    p.ZT = a->Zo <= 16 ? 16 : a->Zo <= 32 ? 32 : 64;
    p.ZIN = (p.ZT - 1) * a->sz + (a->kz - 1) * a->dz + 1;
    p.ztiles = (a->Zo + p.ZT - 1) / p.ZT;
    const long units = (long)a->batch * a->Xo * a->Yo * p.ztiles;
    if (units > 0x7fffffff) return OCCD_EINVAL;
    p.units = (int)units;
    p.cot = (a->cout + 31) / 32;
    p.cit = (a->cin + 31) / 32;
    const int nco = cosplit ? 128 : 32;
    p.grs = nco * 2 + ((nco * 2) % 256 == 64 ? 0 : 64);       // rows 64 B apart modulo 256: conflict-free transposed reads
    p.pls = p.ZIN * 64;
    lds = (size_t)p.ZT * p.grs + (size_t)a->kx * a->ky * p.pls;
    if (lds > 160 * 1024) return OCCD_ENOMEM;
    const long cogroups = cosplit ? (p.cot + 3) / 4 : p.cot;
    // two workgroups per CU; every chunk costs a set of partial tiles (ntaps x 4 KB per (co, ci) tile pair) written and read back
    long chunks = 512 / (cogroups * p.cit);
    if (chunks < 1) chunks = 1;
    if (chunks > units) chunks = units;
    long upc = (units + chunks - 1) / chunks;
    p.units_per_chunk = (int)upc;
    p.div_zin = occd::make_fastdiv(p.ZIN); p.div_ztiles = occd::make_fastdiv(p.ztiles);
    p.div_yo = occd::make_fastdiv(a->Yo); p.div_xo = occd::make_fastdiv(a->Xo);
    return OCCD_OK;
}

}  // namespace

extern "C" {

int64_t occd_conv3d_wgrad_workspace_floats(const occd_conv3d_wgrad_args* a) {
    WgradP p{};
    const int rc = plan(a, p);
    if (rc != OCCD_OK) return rc;
    return (int64_t)p.slots * p.cot * p.cit * p.ntaps * 1024;
}

int occd_conv3d_wgrad(const occd_conv3d_wgrad_args* a, void* stream) {
    WgradP p{};
    const int rc = plan(a, p);
    if (rc != OCCD_OK) return rc;
    if (a->dw == nullptr || a->workspace == nullptr) return OCCD_EINVAL;
    const int64_t need = (int64_t)p.slots * p.cot * p.cit * p.ntaps * 1024;
    if (a->workspace_floats < need) return OCCD_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const int nchunks = p.split_taps ? p.slots : p.slots / 8;
    const double vox = (double)p.rows * p.Zo;
    {
        occd::ProfScope prof("conv3d_wgrad", st, 2.0 * vox * p.ntaps * p.cin * p.cout,
                             4.0 * (vox * p.cout + (double)p.batch * p.X * p.Y * p.Z * p.cin));
        // the row-split mode leaves tail waves without rows: their partial tiles must still be defined
        const dim3 grid((unsigned)nchunks, (unsigned)p.cot, (unsigned)p.cit);
        if (p.split_taps) hipLaunchKernelGGL((wgrad_kernel<4, true>), grid, dim3(512), 0, st, p);
        else hipLaunchKernelGGL((wgrad_kernel<kTPW, false>), grid, dim3(512), 0, st, p);
        int rc2 = occd::check_launch();
        if (rc2 != OCCD_OK) return rc2;
    }
    const long per_slot = (long)p.cot * p.cit * p.ntaps * 1024;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((per_slot + 31) / 32)), dim3(256), 0, st,
                       (const float*)a->workspace, a->dw, p.slots, p.cot, p.cit, p.ntaps, p.cout, p.cin);
    return occd::check_launch();
}

/* K8b: x / gy are fp32 (dtype 0) or bf16 (dtype 1) channels-last rows (both the same type; *_cs / *_coff count
 * elements), dw and the workspace are fp32 as for occd_conv3d_wgrad.                                              */
int64_t occd_conv3d_wgrad_bf16_workspace_floats(const occd_conv3d_wgrad_args* a, int32_t dtype) {
    WgradBP p{};
    bool cosplit;
    size_t lds;
    const int rc = plan_b(a, dtype, p, cosplit, lds);
    if (rc != OCCD_OK) return rc;
    const long nchunks = (p.units + p.units_per_chunk - 1) / p.units_per_chunk;
    return (int64_t)nchunks * p.cot * p.cit * p.ntaps * 1024;
}

int occd_conv3d_wgrad_bf16(const occd_conv3d_wgrad_args* a, int32_t dtype, void* stream) {
    WgradBP p{};
    bool cosplit;
    size_t lds;
    const int rc = plan_b(a, dtype, p, cosplit, lds);
    if (rc != OCCD_OK) return rc;
    if (a->dw == nullptr || a->workspace == nullptr) return OCCD_EINVAL;
    const int nchunks = (p.units + p.units_per_chunk - 1) / p.units_per_chunk;
    const int64_t need = (int64_t)nchunks * p.cot * p.cit * p.ntaps * 1024;
    if (a->workspace_floats < need) return OCCD_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const double vox = (double)a->batch * a->Xo * a->Yo * a->Zo;
    {
        occd::ProfScope prof("conv3d_wgrad_bf16", st, 2.0 * vox * p.ntaps * a->cin * a->cout,
                             (dtype == 1 ? 2.0 : 4.0) * (vox * a->cout + (double)a->batch * a->X * a->Y * a->Z * a->cin));
        void (*kern)(const WgradBP);
        if (cosplit) kern = dtype == 1 ? wgrad_bf16_kernel<9, true, true> : wgrad_bf16_kernel<9, true, false>;
        else kern = dtype == 1 ? wgrad_bf16_kernel<7, false, true> : wgrad_bf16_kernel<7, false, false>;
        if (lds > 64 * 1024 && occd::ensure_big_lds(reinterpret_cast<const void*>(kern)) != OCCD_OK) return OCCD_ELAUNCH;
        const dim3 grid((unsigned)nchunks, (unsigned)(cosplit ? (p.cot + 3) / 4 : p.cot), (unsigned)p.cit);
        hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, p);
        const int rc2 = occd::check_launch();
        if (rc2 != OCCD_OK) return rc2;
    }
    const long per_slot = (long)p.cot * p.cit * p.ntaps * 1024;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((per_slot + 31) / 32)), dim3(256), 0, st,
                       (const float*)a->workspace, a->dw, nchunks, p.cot, p.cit, p.ntaps, a->cout, a->cin);
    return occd::check_launch();
}

}  // extern "C"
