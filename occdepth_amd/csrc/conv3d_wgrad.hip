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

}  // extern "C"
