// K8 -- weight gradient of a 3-D convolution on fp32 MFMA ("dgrad/wgrad igemm for conv3d").
//
//   dW[co][ci][tap] = sum_{b, xo, yo, zo} gy[b, xo, yo, zo][co] * x[b, xo*s - p + kx*d, ..][ci]
//
// is, per tap, a (Cout x voxels) . (voxels x Cin) GEMM whose reduction dimension is the VOXEL index.  Both tensors
// are channels-last, so for v_mfma_f32_32x32x2_f32 the A operand (M = 32 couts, K = 2 voxels) and the B operand
// (N = 32 cins, K = 2 voxels) are plain coalesced row reads: lane&31 = channel, lane>>5 = voxel parity -- no LDS
// staging at all.  A gy fragment is loaded once and reused for all the taps a wave owns; the shifted x fragments
// come back from L1/L2 (each x element is touched by up to 27 (voxel, tap) pairs).
// Work split: workgroup = (chunk of output rows, cout tile, cin tile), 4 waves.
//   >= 8 taps: the waves split the TAPS (wave w owns taps w, w+4, ...; <= 7 accumulators of 16 registers);
//   <  8 taps: the waves split the ROWS and each owns every tap.
// Each wave writes its partial (tap, 32, 32) tiles to a workspace; a second kernel sums the partials in a fixed
// order, so the result is deterministic (no float atomics).
#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int kTPW = 7;            // accumulator sets per wave

struct WgradP {
    const float* x;
    const float* gy;
    float* ws;
    int batch, X, Y, Z, cin, x_cs, x_coff;
    int Xo, Yo, Zo, cout, gy_cs, gy_coff;
    int kx, ky, kz, sx, sy, sz, dx, dy, dz, px, py, pz;
    int ntaps, rows, rows_per_chunk, split_taps, cot, cit, slots;
};

__global__ void __launch_bounds__(256) wgrad_kernel(const WgradP p) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int l = lane & 31, kk = lane >> 5;
    const int chunk = blockIdx.x, co0 = blockIdx.y * 32, ci0 = blockIdx.z * 32;
    const bool co_ok = co0 + l < p.cout, ci_ok = ci0 + l < p.cin;

    // taps owned by this wave, decoded once
    int n_mine = 0;
    int ox[kTPW], oy[kTPW], oz[kTPW], tap_id[kTPW];
#pragma unroll
    for (int i = 0; i < kTPW; ++i) {
        const int t = p.split_taps ? wave + 4 * i : i;
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
    f32x16 acc[kTPW];
#pragma unroll
    for (int i = 0; i < kTPW; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

    const int r_begin = chunk * p.rows_per_chunk;
    const int r_end = min(r_begin + p.rows_per_chunk, p.rows);
    const int r_step = p.split_taps ? 1 : 4;
    for (int row = r_begin + (p.split_taps ? 0 : wave); row < r_end; row += r_step) {
        const int yo = row % p.Yo;
        const int bx = row / p.Yo;
        const int xo = bx % p.Xo;
        const int b = bx / p.Xo;
        const float* gy_row = p.gy + (size_t)row * p.Zo * p.gy_cs + p.gy_coff + co0 + l;
        const int xb = xo * p.sx, yb = yo * p.sy;
        for (int z0 = 0; z0 < p.Zo; z0 += 8) {
            float a[4];
            int zi0[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int z = z0 + 2 * q + kk;
                const bool ok = co_ok && z < p.Zo;
                a[q] = ok ? gy_row[(size_t)(ok ? z : 0) * p.gy_cs] : 0.f;
                zi0[q] = z < p.Zo ? z * p.sz : -(1 << 28);          // out-of-row voxels fail every bounds test
            }
#pragma unroll
            for (int i = 0; i < kTPW; ++i) {
                if (i < n_mine) {
                    const int xi = xb + ox[i], yi = yb + oy[i];
                    if (xi >= 0 && xi < p.X && yi >= 0 && yi < p.Y) {          // uniform over the wave
                        const float* x_row = p.x + ((size_t)(b * p.X + xi) * p.Y + yi) * p.Z * p.x_cs + p.x_coff + ci0 + l;
                        float bv[4];
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const int zi = zi0[q] + oz[i];
                            const bool ok = ci_ok && zi >= 0 && zi < p.Z;
                            bv[q] = ok ? x_row[(size_t)(ok ? zi : 0) * p.x_cs] : 0.f;
                        }
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q], bv[q], acc[i], 0, 0, 0);
                    }
                }
            }
        }
    }
    // partial tiles -> workspace[slot][cot][cit][tap][co 32][ci 32]; D layout: lane&31 = ci, rows co = (r&3)+8(r>>2)+4kk
    const int slot = p.split_taps ? chunk : chunk * 4 + wave;
    float* base = p.ws + ((((size_t)slot * p.cot + blockIdx.y) * p.cit + blockIdx.z) * p.ntaps) * 1024;
#pragma unroll
    for (int i = 0; i < kTPW; ++i)
        if (i < n_mine) {
            float* t = base + (size_t)tap_id[i] * 1024;
#pragma unroll
            for (int r = 0; r < 16; ++r) t[((r & 3) + 8 * (r >> 2) + 4 * kk) * 32 + l] = acc[i][r];
        }
}

// dw[co][ci][tap] = sum over slots, fixed order
__global__ void __launch_bounds__(256) wgrad_reduce_kernel(const float* ws, float* dw, int slots, int cot, int cit,
                                                           int ntaps, int cout, int cin) {
    const long n = (long)cout * cin * ntaps;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const int tap = (int)(i % ntaps);
        const long cc = i / ntaps;
        const int ci = (int)(cc % cin), co = (int)(cc / cin);
        const size_t tile = (((size_t)(co >> 5) * cit + (ci >> 5)) * ntaps + tap) * 1024 + (co & 31) * 32 + (ci & 31);
        const size_t slot_stride = (size_t)cot * cit * ntaps * 1024;
        float s = 0.f;
        for (int k = 0; k < slots; ++k) s += ws[(size_t)k * slot_stride + tile];
        dw[i] = s;
    }
}

int plan(const occd_conv3d_wgrad_args* a, WgradP& p) {
    if (a == nullptr || a->x == nullptr || a->gy == nullptr) return OCCD_EINVAL;
    if (a->batch < 1 || a->X < 1 || a->Y < 1 || a->Z < 1 || a->Xo < 1 || a->Yo < 1 || a->Zo < 1) return OCCD_EINVAL;
    if (a->cin < 1 || a->cout < 1 || a->kx < 1 || a->ky < 1 || a->kz < 1) return OCCD_EINVAL;
    if (a->sx < 1 || a->sy < 1 || a->sz < 1 || a->dx < 1 || a->dy < 1 || a->dz < 1) return OCCD_EINVAL;
    if (a->x_coff < 0 || a->gy_coff < 0 || a->x_coff + a->cin > a->x_cs || a->gy_coff + a->cout > a->gy_cs) return OCCD_EINVAL;
    const long ntaps = (long)a->kx * a->ky * a->kz;
    if (ntaps > 4 * kTPW) return OCCD_EINVAL;   // kernels beyond 28 taps are not planned
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
    // ~2048 workgroups over the chip; a chunk is at least 4 rows (one per wave in the row-split mode)
    long chunks = 2048 / ((long)p.cot * p.cit);
    if (chunks < 1) chunks = 1;
    long rpc = (rows + chunks - 1) / chunks;
    if (rpc < 4) rpc = 4;
    p.rows_per_chunk = (int)rpc;
    const long nchunks = (rows + rpc - 1) / rpc;
    p.slots = (int)(p.split_taps ? nchunks : nchunks * 4);
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
    const int nchunks = p.split_taps ? p.slots : p.slots / 4;
    const double vox = (double)p.rows * p.Zo;
    {
        occd::ProfScope prof("conv3d_wgrad", st, 2.0 * vox * p.ntaps * p.cin * p.cout,
                             4.0 * (vox * p.cout + (double)p.batch * p.X * p.Y * p.Z * p.cin));
        // the row-split mode leaves tail waves without rows: their partial tiles must still be defined
        hipLaunchKernelGGL(wgrad_kernel, dim3((unsigned)nchunks, (unsigned)p.cot, (unsigned)p.cit), dim3(256), 0, st, p);
        int rc2 = occd::check_launch();
        if (rc2 != OCCD_OK) return rc2;
    }
    const long n = (long)p.cout * p.cin * p.ntaps;
    long blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)blocks), dim3(256), 0, st, (const float*)a->workspace, a->dw,
                       p.slots, p.cot, p.cit, p.ntaps, p.cout, p.cin);
    return occd::check_launch();
}

}  // extern "C"
