// K10 -- fused Winograd F(2x2, 3x3) convolution on the CDNA4 fp32 matrix pipe.
//
//   y = act( conv3x3(x, g * scale[co], stride 1, pad 1) + shift[co] ) (+ res)       x, y, res: NCHW float32
//
// Per 2x2 output tile with 4x4 input patch d:   Y = A^T [ sum_cin U_cin (.) V_cin ] A,   U = G g G^T,  V = B^T d B.
// The 16 positions xi of the Winograd domain are 16 independent GEMMs over cin: 2.25x fewer multiplies than the direct
// form.  csrc/wino2d.hip runs them unfused (V and M round-trip through HBM: 4x the input and 4x the output bytes),
// which only pays where tiles are few; this kernel keeps everything on chip:
//   * a workgroup (4 waves) owns 128 tiles x 32 couts; a wave owns 32 tiles x 32 couts x ALL 16 xi, i.e. 16
//     accumulator tiles of v_mfma_f32_32x32x2_f32 (256 VGPRs): the output transform needs the 16 xi of one
//     (tile, cout) together, and this way they sit in one lane's registers -- M never exists in memory;
//   * cin is walked in chunks of 8: the chunk's input patch (8 x PR x PC, zero padded = the convolution's padding) is
//     staged through LDS (double buffered, one barrier per chunk, the next chunk's global loads fly under the MFMAs);
//   * V never exists in memory either: a lane is the MFMA column of ONE tile and of 4 of the chunk's 8 cins (the K
//     index is permuted so lane half h owns cins 4h..4h+3, as in K2), so it reads exactly its own 4x4 patches from
//     LDS (ds_read_b64, conflict-free row stride), transforms them on the VALU (32 adds per patch) and feeds the 64
//     resulting values to the MFMAs as B operands; no V staging, no LDS writes besides the raw patch;
//   * U (BatchNorm scale folded in) is pre-packed in A-fragment order [chunk][xi][cout/32][lane][4]: one
//     global_load_dwordx4 per lane per (chunk, xi), L2 resident (<= 1 MB for the high-resolution decoder levels);
//   * epilogue: A^T m A in registers (24 adds per (tile, cout)), + shift, activation, optional residual, NCHW
//     stores of 2-pixel pairs (16 consecutive lanes = 16 consecutive tiles = 128 contiguous bytes).
// MFMA work per launch: 2 * 16 * tiles * Cin8 * Cout32 FLOP (= direct FLOPs / 2.25 up to channel padding);
// algorithmic HBM bytes: 4 * (B Cin H W + B Cout H W) + the packed U.
//
// Reference semantics replaced: nn.Conv2d(k=3, s=1, p=1) + BatchNorm2d (eval) + LeakyReLU of
// occdepth/models/unet2d.py:24-46 (UpSampleBN), and the 3x3 convolutions of DepthNet / BasicBlock
// (occdepth/models/flosp_depth/flosp_depth.py:201-257).
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

namespace {

struct WinoP {
    const float* x;
    const float* upk;
    const float* shift;
    const float* res;
    float* y;
    int B, Cin, Cout, H, W;
    int chunks, nblk, wg_ty, wg_tx, nwg;
    int act, res_first;
    float slope;
};

__device__ __forceinline__ float wino_act(float v, int act, float slope) {
    if (act == 1) return fmaxf(v, 0.f);
    if (act == 2) return v / (1.f + expf(-v));
    if (act == 3) return v > 0.f ? v : v * slope;
    return v;
}

// TWV: tiles per wave along x.  16 -> a wave is 2 x 16 tiles (4 x 32 pixels), a workgroup 8 x 16 tiles;
//                               32 -> a wave is 1 x 32 tiles (2 x 64 pixels), a workgroup 4 x 32 tiles.
template <int TWV>
__global__ void __launch_bounds__(256) wino3x3_kernel(const WinoP p) {
    constexpr int RW = 32 / TWV;                 // tile rows per wave
    constexpr int WGR = 4 * RW;                  // tile rows per workgroup
    constexpr int PR = 2 * WGR + 2;              // patch rows / cols (one halo pixel each side)
    constexpr int PC = 2 * TWV + 2;
    // ds_read_b64 serves 32 lanes per cycle over 64 banks: a wave's two tile rows (TWV = 16) must sit 32 banks apart
    constexpr int RS = TWV == 16 ? 48 : 68;      // LDS row stride in floats
    constexpr int PLANE = PR * RS;
    constexpr int NEL = 8 * PR * PC;             // elements of one staged chunk
    constexpr int NLD = (NEL + 255) / 256;
    extern __shared__ __attribute__((aligned(16))) float lds[];      // 2 buffers x 8 planes

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, kk = lane >> 5;

    // XCD-aware bijective remap (cout blocks of one tile block stay on one XCD's L2: they re-read the same patch)
    uint32_t bid = blockIdx.x;
    {
        const uint32_t nwg = p.nwg, q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int nb = bid % p.nblk;
    uint32_t t = bid / p.nblk;
    const int bx = t % p.wg_tx;
    t /= p.wg_tx;
    const int by = t % p.wg_ty;
    const int b = t / p.wg_ty;
    const int ty0 = by * WGR, tx0 = bx * TWV;
    const int gy0 = 2 * ty0 - 1, gx0 = 2 * tx0 - 1;

    // chunk-invariant staging slots of this thread: global offset inside a chunk (-1: outside the image), LDS offset
    int goff[NLD], loff[NLD];
    const size_t plane_hw = (size_t)p.H * p.W;
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
        const int e = i * 256 + tid;
        const int cin = e / (PR * PC), rem = e - cin * (PR * PC);
        const int row = rem / PC, col = rem - row * PC;
        const int gy = gy0 + row, gx = gx0 + col;
        const bool ok = e < NEL && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W;
        goff[i] = ok ? (int)(cin * plane_hw + (size_t)gy * p.W + gx) : -1;
        loff[i] = e < NEL ? cin * PLANE + row * RS + col : -1;
    }
    const float* const xb = p.x + (size_t)b * p.Cin * plane_hw;

    // this lane's tile inside the workgroup and its patch corner in LDS
    const int lr = wave * RW + li / TWV, lc = li % TWV;
    const int pbase = kk * 4 * PLANE + (2 * lr) * RS + 2 * lc;

    f32x16 acc[16];
#pragma unroll
    for (int xi = 0; xi < 16; ++xi)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[xi][r] = 0.f;

    const float* const ulane = p.upk + (size_t)nb * 256 + lane * 4;
    const size_t u_xi = (size_t)p.nblk * 256;            // floats per (chunk, xi) record

    float stage[NLD];
    auto load_chunk = [&](int c) {
        const int c0 = c * 8;
        const float* src = xb + (size_t)c0 * plane_hw;
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int e = i * 256 + tid;
            const int cin = e / (PR * PC);
            const bool ok = goff[i] >= 0 && c0 + cin < p.Cin;
            stage[i] = ok ? src[goff[i]] : 0.f;
        }
    };
    auto store_chunk = [&](int buf) {
        float* dst = lds + buf * 8 * PLANE;
#pragma unroll
        for (int i = 0; i < NLD; ++i)
            if (loff[i] >= 0) dst[loff[i]] = stage[i];
    };

    load_chunk(0);
    store_chunk(0);
    __syncthreads();

    f32x4 u_cur = *(const f32x4*)ulane;
    for (int c = 0; c < p.chunks; ++c) {
        const bool more = c + 1 < p.chunks;
        if (more) load_chunk(c + 1);                      // global loads fly under this chunk's transform + MFMAs

        // ---- input transform of this lane's 4 patches (cins kk*4 .. kk*4+3 of the chunk): V[xi][q]
        const float* pl = lds + (c & 1) * 8 * PLANE + pbase;
        float v[16][4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float d[4][4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const f32x2 lo = *(const f32x2*)(pl + q * PLANE + r * RS);
                const f32x2 hi = *(const f32x2*)(pl + q * PLANE + r * RS + 2);
                d[r][0] = lo.x; d[r][1] = lo.y; d[r][2] = hi.x; d[r][3] = hi.y;
            }
            float tt[4][4];                                // B^T d,  B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1]
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                tt[0][j] = d[0][j] - d[2][j];
                tt[1][j] = d[1][j] + d[2][j];
                tt[2][j] = d[2][j] - d[1][j];
                tt[3][j] = d[1][j] - d[3][j];
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {                  // (B^T d) B
                v[4 * i + 0][q] = tt[i][0] - tt[i][2];
                v[4 * i + 1][q] = tt[i][1] + tt[i][2];
                v[4 * i + 2][q] = tt[i][2] - tt[i][1];
                v[4 * i + 3][q] = tt[i][1] - tt[i][3];
            }
        }

        // ---- 16 xi x 4 k-steps of MFMA; U fragments prefetched one xi ahead
        const float* up = ulane + (size_t)c * 16 * u_xi;
#pragma unroll
        for (int xi = 0; xi < 16; ++xi) {
            f32x4 u_nxt;
            if (xi < 15) u_nxt = *(const f32x4*)(up + (size_t)(xi + 1) * u_xi);
            else u_nxt = *(const f32x4*)(up + (size_t)(more ? 16 : 0) * u_xi);   // next chunk's xi = 0 (or a dummy re-read)
#pragma unroll
            for (int q = 0; q < 4; ++q)
                acc[xi] = __builtin_amdgcn_mfma_f32_32x32x2f32(u_cur[q], v[xi][q], acc[xi], 0, 0, 0);
            u_cur = u_nxt;
        }

        if (more) store_chunk((c + 1) & 1);
        __syncthreads();
    }

    // ---------------- epilogue: Y = A^T m A (A^T = [1 1 1 0; 0 1 -1 -1]), shift, activation, residual, NCHW store.
    // D = U^T-rows x tile-columns: this lane is tile `li` of the wave and register r is cout 8 (r >> 2) + 4 kk + (r & 3).
    const int oy0 = 2 * (ty0 + lr), ox = 2 * (tx0 + lc);
    const bool pair = ox + 1 < p.W && (p.W & 1) == 0;      // 8-byte aligned 2-pixel store
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int co = nb * 32 + 8 * (r >> 2) + 4 * kk + (r & 3);
        float s[2][4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            s[0][j] = acc[j][r] + acc[4 + j][r] + acc[8 + j][r];
            s[1][j] = acc[4 + j][r] - acc[8 + j][r] - acc[12 + j][r];
        }
        if (co >= p.Cout || ox >= p.W) continue;
        const float sh = p.shift != nullptr ? p.shift[co] : 0.f;
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            const int oy = oy0 + a;
            if (oy >= p.H) continue;
            float y0 = s[a][0] + s[a][1] + s[a][2] + sh;
            float y1 = s[a][1] - s[a][2] - s[a][3] + sh;
            const size_t o = (((size_t)b * p.Cout + co) * p.H + oy) * p.W + ox;
            float r0 = 0.f, r1 = 0.f;
            if (p.res != nullptr) {
                r0 = p.res[o];
                if (ox + 1 < p.W) r1 = p.res[o + 1];
            }
            if (p.res_first) { y0 += r0; y1 += r1; }
            y0 = wino_act(y0, p.act, p.slope);
            y1 = wino_act(y1, p.act, p.slope);
            if (!p.res_first) { y0 += r0; y1 += r1; }
            if (pair) {
                *(f32x2*)(p.y + o) = f32x2{y0, y1};
            } else {
                p.y[o] = y0;
                if (ox + 1 < p.W) p.y[o + 1] = y1;
            }
        }
    }
}

// U[chunk][xi][cout/32][lane][4]: cout = blk*32 + (lane & 31), cin = chunk*8 + (lane >> 5)*4 + q, value
// (G g G^T)[xi] * scale[cout]; zero outside (cout, cin).  G = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1]; float64 arithmetic.
__global__ void wino_pack_kernel(const float* __restrict__ w, const float* __restrict__ scale, float* __restrict__ upk,
                                 int cout, int cin, int nblk, long total) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int q = i & 3, lane = (i >> 2) & 63;
    long t = i >> 8;
    const int blk = t % nblk; t /= nblk;
    const int xi = t & 15;
    const int chunk = (int)(t >> 4);
    const int co = blk * 32 + (lane & 31), ci = chunk * 8 + (lane >> 5) * 4 + q;
    float out = 0.f;
    if (co < cout && ci < cin) {
        const double G[4][3] = {{1, 0, 0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0, 0, 1}};
        const float* g = w + ((size_t)co * cin + ci) * 9;
        const int a = xi >> 2, bcol = xi & 3;
        double s = 0.0;
#pragma unroll
        for (int m = 0; m < 3; ++m)
#pragma unroll
            for (int n = 0; n < 3; ++n) s += G[a][m] * (double)g[m * 3 + n] * G[bcol][n];
        if (scale != nullptr) s *= (double)scale[co];
        out = (float)s;
    }
    upk[i] = out;
}

template <int TWV>
int launch_wino(const WinoP& p, hipStream_t st) {
    constexpr int RW = 32 / TWV, PR = 2 * 4 * RW + 2, RS = TWV == 16 ? 48 : 68;
    const size_t lds = (size_t)2 * 8 * PR * RS * sizeof(float);
    hipLaunchKernelGGL(wino3x3_kernel<TWV>, dim3((unsigned)p.nwg), dim3(256), lds, st, p);
    return occd::check_launch();
}

}  // namespace

extern "C" {

int64_t occd_wino_packed_floats(int32_t cout, int32_t cin) {
    if (cout < 1 || cin < 1) return OCCD_EINVAL;
    return (int64_t)((cin + 7) / 8) * 16 * ((cout + 31) / 32) * 256;
}

int occd_wino_pack_weights(const float* w, const float* scale, float* upk, int32_t cout, int32_t cin, void* stream) {
    if (w == nullptr || upk == nullptr || cout < 1 || cin < 1) return OCCD_EINVAL;
    const long total = occd_wino_packed_floats(cout, cin);
    hipLaunchKernelGGL(wino_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w,
                       scale, upk, cout, cin, (cout + 31) / 32, total);
    return occd::check_launch();
}

int occd_wino_conv3x3_fwd(const occd_wino_args* a, void* stream) {
    if (a == nullptr || a->x == nullptr || a->upk == nullptr || a->y == nullptr) return OCCD_EINVAL;
    if (a->batch < 1 || a->cin < 1 || a->cout < 1 || a->H < 1 || a->W < 1) return OCCD_EINVAL;
    if (a->act < 0 || a->act > 3) return OCCD_EINVAL;
    if ((double)a->cin * a->H * a->W >= 2147483648.0) return OCCD_EINVAL;      // 32-bit offsets inside one image
    WinoP p{};
    p.x = a->x; p.upk = a->upk; p.shift = a->shift; p.res = a->res; p.y = a->y;
    p.B = a->batch; p.Cin = a->cin; p.Cout = a->cout; p.H = a->H; p.W = a->W;
    p.chunks = (a->cin + 7) / 8;
    p.nblk = (a->cout + 31) / 32;
    p.act = a->act; p.res_first = a->res_first; p.slope = a->slope;
    const int th = (a->H + 1) / 2, tw = (a->W + 1) / 2;
    // wide waves (1 x 32 tiles) waste less halo; narrow ones (2 x 16) waste fewer tiles on narrow images
    int twv = a->tile_hint == 16 || a->tile_hint == 32 ? a->tile_hint : 0;
    if (twv == 0) {
        const long w16 = (long)((tw + 15) / 16) * 16 * ((th + 7) / 8) * 8;
        const long w32 = (long)((tw + 31) / 32) * 32 * ((th + 3) / 4) * 4;
        twv = w32 <= w16 ? 32 : 16;
    }
    const int wgr = twv == 16 ? 8 : 4;
    p.wg_tx = (tw + twv - 1) / twv;
    p.wg_ty = (th + wgr - 1) / wgr;
    const long nwg = (long)p.nblk * p.wg_tx * p.wg_ty * p.B;
    if (nwg > 0x7fffffffL) return OCCD_EINVAL;
    p.nwg = (int)nwg;
    const double tiles = (double)p.B * th * tw;
    occd::ProfScope prof("wino_conv3x3", (hipStream_t)stream, 2.0 * 16 * tiles * p.chunks * 8 * p.nblk * 32,
                         4.0 * p.B * ((double)p.Cin + p.Cout) * p.H * p.W);
    return twv == 16 ? launch_wino<16>(p, (hipStream_t)stream) : launch_wino<32>(p, (hipStream_t)stream);
}

}  // extern "C"
