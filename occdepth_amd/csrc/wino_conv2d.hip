// K10 -- fused Winograd F(2x2, 3x3) convolution on the CDNA4 fp32 matrix pipe.
//
//   y = act( conv3x3(x, g * scale[co], stride 1, pad 1) + shift[co] ) (+ res)       x, y, res: NCHW float32
//
// Per 2x2 output tile with 4x4 input patch d:   Y = A^T [ sum_cin U_cin (.) V_cin ] A,   U = G g G^T,  V = B^T d B.
// The 16 positions xi of the Winograd domain are 16 independent GEMMs over cin: 2.25x fewer multiplies than the direct
// form.  csrc/wino2d.hip runs them unfused (V and M round-trip through HBM: 4x the input and 4x the output bytes),
// which only pays where tiles are few; this kernel keeps everything on chip:
//   * a workgroup (8 waves) owns 128 tiles x 32 couts; a PAIR of waves owns 32 tiles x 32 couts and all 16 xi, 8 each
//     (8 accumulator tiles of v_mfma_f32_32x32x2_f32 = 128 AGPRs per wave, two waves per SIMD); the output transform
//     needs the 16 xi of one (tile, cout) together, so the pair swaps half of its accumulators through LDS once, after
//     the K loop -- M never exists in memory;
//   * cin is walked in chunks of 8: the chunk's input patch (8 x PR x PC, zero padded = the convolution's padding) is
//     staged into LDS by LDS-DMA (double buffered, one barrier per chunk, chunk c+1 lands under the MFMAs of chunk c);
//   * V never exists in memory either: a lane is the MFMA column of ONE tile and of 4 of the chunk's 8 cins (the K
//     index is permuted so lane half h owns cins 4h..4h+3, as in K2), so it reads exactly its own 4x4 patches from
//     LDS (ds_read_b64, conflict-free row stride), transforms them on the VALU (32 adds per patch) and feeds the 64
//     resulting values to the MFMAs as B operands; no V staging, no LDS writes besides the raw patch;
//   * U (BatchNorm scale folded in) is pre-packed in A-fragment order [chunk][4 q + 2 h + part][cout/32][lane][4 xi] (L2 resident,
//     <= 1 MB for the high-resolution decoder levels) and also arrives by LDS-DMA, 16 KB per chunk, read back as
//     ds_read_b128 fragments by all four waves;
//   * epilogue: A^T m A in registers (24 adds per (tile, cout)), + shift, activation, optional residual, NCHW
//     stores of 2-pixel pairs (16 consecutive lanes = 16 consecutive tiles = 128 contiguous bytes).
// MFMA work per launch: 2 * 16 * tiles * Cin8 * Cout32 FLOP (= direct FLOPs / 2.25 up to channel padding);
// algorithmic HBM bytes: 4 * (B Cin H W + B Cout H W) + the packed U.
//
// Reference semantics replaced: nn.Conv2d(k=3, s=1, p=1) + BatchNorm2d (eval) + LeakyReLU of
// occdepth/models/unet2d.py:24-46 (UpSampleBN), and the 3x3 convolutions of DepthNet / BasicBlock
// (occdepth/models/flosp_depth/flosp_depth.py:201-257).
#include <type_traits>
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

// TWV: tiles per wave pair along x.  16 -> a pair is 2 x 16 tiles (4 x 32 pixels), a workgroup 8 x 16 tiles;
//                                    32 -> a pair is 1 x 32 tiles (2 x 64 pixels), a workgroup 4 x 32 tiles.
//
// 512 threads = 8 waves = 2 per SIMD.  Waves w and w + 4 form a PAIR on the same 32 tiles x 32 couts and split the
// Winograd domain: half h = w >> 2 owns the rows i in {2h, 2h + 1} of the 4 x 4 domain (xi = 4 i + j), i.e. 8
// accumulator tiles = 128 AGPRs.  A half needs only 3 of the 4 patch rows and 8 + 8 adds per patch.  After the loop the
// pair exchanges accumulators through LDS so that each wave finishes 8 of the 16 cout registers.
//
// What limits this kernel (measured on the 688 -> 320 level, 147 TF/s attainable at its tile padding): the fp32 MFMA
// runs at the fp32 VECTOR rate and every other instruction of the SIMD costs matrix-pipe time -- MFMAs + barriers
// alone 124 TF/s, + LDS reads -15, + transform adds -11, + DMA issue -10, in ANY arrangement tried: all 16 xi in one
// wave per SIMD (50 %: nothing hides its LDS waits), two waves in strict MFMA / prepare ping-pong phases (75 TF/s: the
// preparing wave does not run in the shadow of the partner's MFMAs), or the interleaved form below (93 TF/s).  So the
// loop is built to (a) issue every LDS read one block (8 MFMAs) before its use, (b) keep the instruction count per
// MFMA minimal (one instantiation per domain half: no selects; operands re-read per block: no register copies) and
// (c) let the two waves of a SIMD do their VALU work at different points of a block:
//     half 0:  reads(q+1) | 8 MFMAs(q)                      | transform(q+1)
//     half 1:  reads(q+1) | 4 MFMAs(q) | transform(q+1) | 4 MFMAs(q)
// sched_barrier(0) pins the segment order (hipcc otherwise sinks the reads to their first use); the last chunk is
// peeled so the loop body is straight-line code (exact s_waitcnt counts instead of lgkmcnt(0) at a control-flow join).
//
// Staging is LDS-DMA only (no staging VGPRs, no ds_write, no global_load results for hipcc to wait on):
//   patch : buffer_load_dword ... lds through a descriptor of image b's remaining channels -- the LDS image is
//           lane-linear ([cin][row][RS] floats, 64 consecutive floats per wave instruction), every lane supplies the
//           byte offset of ITS element, and the hardware bounds check provides both kinds of zero fill: lanes outside
//           the image (the convolution's padding, the row pad) carry an offset past the descriptor, channels past Cin
//           fall off its end;
//   U     : global_load_lds_dwordx4, one 1-KB fragment record per wave instruction, already in A-fragment order.
// Block (c, 3) reads chunk c+1, so the barrier (vmcnt(0): chunk c+1, issued a chunk earlier, has landed; every wave has
// issued its last reads of chunk c) sits in front of it and the DMA of chunk c+2 into chunk c's buffers follows it.
// EXP (development A/B switches, wrong results): 1 = no DMA after chunk 1, 2 = DMA only (no MFMA)
template <int TWV, int EXP = 0>
__global__ void __launch_bounds__(512, 2) wino3x3_kernel(const WinoP p) {
    constexpr int RW = 32 / TWV;                 // tile rows per wave pair
    constexpr int WGR = 4 * RW;                  // tile rows per workgroup
    constexpr int PR = 2 * WGR + 2;              // patch rows / cols (one halo pixel each side)
    constexpr int PC = 2 * TWV + 2;
    // ds_read_b64 serves 32 lanes per cycle over 64 banks: a pair's two tile rows (TWV = 16) must sit 32 banks apart
    constexpr int RS = TWV == 16 ? 48 : 68;      // LDS row stride in floats
    constexpr int PLANE = PR * RS;
    constexpr int NEL = 8 * PLANE;               // floats of one staged chunk image (pads included)
    constexpr int NDMA = (NEL + 511) / 512;      // patch DMA instructions per wave per chunk
    constexpr int PBUF = NDMA * 512;             // floats per patch buffer (whole wave instructions)
    constexpr int UBUF = 16 * 256;               // floats per U buffer: 16 xi records of 64 lanes x 4
    extern __shared__ __attribute__((aligned(16))) float lds[];      // [2][PBUF] patches, then [2][UBUF] weights

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pr = wave & 3, h = wave >> 2;      // tile group of the pair, half of the Winograd domain
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

    // chunk-invariant byte offset of this lane's element in each of its wave's DMA instructions (relative to the
    // chunk's first channel); 0xFFFFFFF0 = "outside": past any descriptor, the DMA writes 0.0f
    const size_t plane_hw = (size_t)p.H * p.W;
    uint32_t voff[NDMA];
#pragma unroll
    for (int i = 0; i < NDMA; ++i) {
        const int e = (wave * NDMA + i) * 64 + lane;
        const int cin = e / PLANE, rem = e - cin * PLANE;
        const int row = rem / RS, col = rem - row * RS;
        const int gy = gy0 + row, gx = gx0 + col;
        const bool ok = e < NEL && col < PC && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W;
        voff[i] = ok ? (uint32_t)((cin * plane_hw + (size_t)gy * p.W + gx) * 4) : 0xFFFFFFF0u;
    }
    const float* const xb = p.x + (size_t)b * p.Cin * plane_hw;
    const float* const ubase = p.upk + (size_t)nb * 256 + lane * 4;
    const size_t u_xi = (size_t)p.nblk * 256;            // floats per (chunk, xi) record

    auto issue_chunk = [&](int c) {
        const int buf = c & 1;
        // descriptor over channels [8c, Cin) of image b: offsets past the last channel read as zero
        const auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(xb + (size_t)c * 8 * plane_hw), 0,
                                                            (uint32_t)((size_t)(p.Cin - c * 8) * plane_hw * 4), 0x00020000);
        float* pdst = lds + buf * PBUF + wave * NDMA * 64;
#pragma unroll
        for (int i = 0; i < NDMA; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, pdst + i * 64, 4, voff[i], 0, 0, 0);
        float* udst = lds + 2 * PBUF + buf * UBUF + wave * 2 * 256;
        const float* usrc = ubase + ((size_t)c * 16 + wave * 2) * u_xi;
#pragma unroll
        for (int j = 0; j < 2; ++j)
            __builtin_amdgcn_global_load_lds(usrc + (size_t)j * u_xi, udst + j * 256, 16, 0, 0);
    };

    // this lane's tile inside the workgroup and its patch corner in LDS (rows h .. h + 2 of the 4 x 4 patch)
    const int lr = pr * RW + li / TWV, lc = li % TWV;
    const int pbase = kk * 4 * PLANE + (2 * lr + h) * RS + 2 * lc;

    // An (empty) asm with an AGPR operand makes hipcc select the AGPR form of the MFMAs (at 2 waves per SIMD it would
    // otherwise put the accumulators in VGPRs, which leaves too few for the register-level prefetch below).
    asm volatile("" ::"a"(0.f));
    f32x16 acc[8];                                        // xi = 8 h + k
#pragma unroll
    for (int k = 0; k < 8; ++k)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[k][r] = 0.f;

    auto read_d = [&](const float* pl, int q, float (&d)[3][4]) {
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const f32x2 lo = *(const f32x2*)(pl + q * PLANE + r * RS);
            const f32x2 hi = *(const f32x2*)(pl + q * PLANE + r * RS + 2);
            d[r][0] = lo.x; d[r][1] = lo.y; d[r][2] = hi.x; d[r][3] = hi.y;
        }
    };
    // U records of a chunk: r = 4 q + 2 h + part, each [lane][4 xi] (see wino_pack_kernel): 2 x ds_read_b128 per cin
    auto read_u = [&](int buf, int q, float (&u)[8]) {
        const f32x4* ul = (const f32x4*)(lds + 2 * PBUF + buf * UBUF) + ((q * 4 + h * 2) * 64 + lane);
        const f32x4 a = ul[0], b4 = ul[64];
        u[0] = a.x; u[1] = a.y; u[2] = a.z; u[3] = a.w; u[4] = b4.x; u[5] = b4.y; u[6] = b4.z; u[7] = b4.w;
    };
    const bool h0 = h == 0;                                // wave-uniform
    auto transform = [&](auto hc, const float (&d)[3][4], float (&v)[8]) {
        constexpr bool h0 = decltype(hc)::value == 0;
        // rows 2h, 2h + 1 of B^T d (B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1]); d holds patch rows h, h+1, h+2:
        //   h = 0: d0 - d2, d1 + d2        h = 1 (d = rows 1, 2, 3): d2 - d1, d1 - d3
        float tt[2][4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            tt[0][j] = h0 ? d[0][j] - d[2][j] : d[1][j] - d[0][j];
            tt[1][j] = h0 ? d[1][j] + d[2][j] : d[0][j] - d[2][j];
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {                      // (B^T d) B
            v[4 * i + 0] = tt[i][0] - tt[i][2];
            v[4 * i + 1] = tt[i][1] + tt[i][2];
            v[4 * i + 2] = tt[i][2] - tt[i][1];
            v[4 * i + 3] = tt[i][1] - tt[i][3];
        }
    };
    auto mfma4 = [&](const float (&u)[8], const float (&v)[8], int k0) {
#pragma unroll
        for (int k = k0; k < k0 + 4; ++k) {
            if (EXP == 2) acc[k][0] += u[k] * v[k];
            else acc[k] = __builtin_amdgcn_mfma_f32_32x32x2f32(u[k], v[k], acc[k], 0, 0, 0);
        }
    };

    issue_chunk(0);
    __syncthreads();                                      // (drains the DMA: vmcnt(0) + barrier)
    if (p.chunks > 1) issue_chunk(1);

    // The whole K loop is instantiated once per domain half (a wave-uniform branch): no per-lane selects inside.
    auto k_loop = [&](auto hc) {
        constexpr bool first_half = decltype(hc)::value == 0;
        float d[3][4], ua[8], ub[8], va[8], vb[8];         // (a, b): register sets alternating per block, no copies
        // one block: operands (u_cur, v_cur) of cin q; prefetch + transform of the next block into (u_nxt, v_nxt)
        auto block = [&](const float* pl_nxt, int buf_nxt, int q_nxt, float (&u_cur)[8], float (&v_cur)[8],
                         float (&u_nxt)[8], float (&v_nxt)[8]) {
            read_d(pl_nxt, q_nxt, d);
            read_u(buf_nxt, q_nxt, u_nxt);
            __builtin_amdgcn_sched_barrier(0);
            mfma4(u_cur, v_cur, 0);
            if (first_half) {
                mfma4(u_cur, v_cur, 4);
                __builtin_amdgcn_sched_barrier(0);
                transform(hc, d, v_nxt);
            } else {
                __builtin_amdgcn_sched_barrier(0);
                transform(hc, d, v_nxt);
                __builtin_amdgcn_sched_barrier(0);
                mfma4(u_cur, v_cur, 4);
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        read_d(lds + pbase, 0, d);
        read_u(0, 0, ua);
        transform(hc, d, va);
        for (int c = 0; c + 1 < p.chunks; ++c) {
            const float* pl = lds + (c & 1) * PBUF + pbase;
            const float* pn = lds + ((c + 1) & 1) * PBUF + pbase;
            block(pl, c & 1, 1, ua, va, ub, vb);
            block(pl, c & 1, 2, ub, vb, ua, va);
            block(pl, c & 1, 3, ua, va, ub, vb);
            __syncthreads();                               // vmcnt(0) + lgkmcnt(0) + barrier
            if (EXP != 1 && c + 2 < p.chunks) issue_chunk(c + 2);
            block(pn, (c + 1) & 1, 0, ub, vb, ua, va);
        }
        const int lb = (p.chunks - 1) & 1;
        const float* pl = lds + lb * PBUF + pbase;
        block(pl, lb, 1, ua, va, ub, vb);
        block(pl, lb, 2, ub, vb, ua, va);
        block(pl, lb, 3, ua, va, ub, vb);
        mfma4(ub, vb, 0);
        mfma4(ub, vb, 4);
    };
    if (h0) k_loop(std::integral_constant<int, 0>{});
    else k_loop(std::integral_constant<int, 1>{});
    __syncthreads();                                       // every wave is done with the staging buffers

    // The exchange and the epilogue are instantiated once per domain half, like the K loop: with a run-time `h` the
    // accumulator registers are picked by a wave-uniform but DYNAMIC index, which hipcc lowers to s_set_gpr_idx moves
    // through a 16-register VGPR copy of each tile -- that copy was the kernel's 16 - 50 spilled VGPRs.
    // Only the register selection is specialised; the barrier between the exchange write and the exchange read is ONE
    // instruction that every wave of the workgroup reaches on the same path (the two halves used to meet at two
    // different s_barrier instructions, which works on gfx950's arrival-counting barrier but is not defined behaviour).
    // ---------------- pair exchange: wave half h finishes the cout registers r in [8h, 8h + 8) and needs the partner's
    // 8 xi for them.  xchg[pair][writer half][k * 8 + (r & 7)][lane]  (the staging buffers are dead: last barrier above)
    float* const xchg = lds;
    auto xwrite = [&](auto hc) {
        constexpr int H = decltype(hc)::value;
        constexpr bool h0 = H == 0;
        float* mine = xchg + ((pr * 2 + H) * 64) * 64 + lane;
#pragma unroll
        for (int k = 0; k < 8; ++k)
#pragma unroll
            for (int r8 = 0; r8 < 8; ++r8) mine[(k * 8 + r8) * 64] = h0 ? acc[k][8 + r8] : acc[k][r8];
    };
    if (h0) xwrite(std::integral_constant<int, 0>{});
    else xwrite(std::integral_constant<int, 1>{});
    __syncthreads();
    auto finish = [&](auto hc) {
    constexpr int H = decltype(hc)::value;
    constexpr bool h0 = H == 0;
    const float* theirs = xchg + ((pr * 2 + (1 - H)) * 64) * 64 + lane;

    // ---------------- epilogue: Y = A^T m A (A^T = [1 1 1 0; 0 1 -1 -1]), shift, activation, residual, NCHW store.
    // D = U^T-rows x tile-columns: this lane is tile `li` of the pair and register r is cout 8 (r >> 2) + 4 kk + (r & 3).
    const int oy0 = 2 * (ty0 + lr), ox = 2 * (tx0 + lc);
    const bool pair = ox + 1 < p.W && (p.W & 1) == 0;      // 8-byte aligned 2-pixel store
    // One cout register per iteration, fenced by scheduling barriers: left alone, hipcc hoists the 64 exchange reads of
    // all eight unrolled iterations above the first store and spills 16 - 50 VGPRs beside the 128 live accumulators.
#pragma unroll
    for (int r8 = 0; r8 < 8; ++r8) {
        __builtin_amdgcn_sched_barrier(0);
        const int r = H * 8 + r8;
        const int co = nb * 32 + 8 * (r >> 2) + 4 * kk + (r & 3);
        float m[4][4];                                     // m[i][j], xi = 4 i + j; rows 2h, 2h + 1 are this wave's
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float own = h0 ? acc[k][r8] : acc[k][8 + r8];
            const float oth = theirs[(k * 8 + r8) * 64];
            m[k >> 2][k & 3] = h0 ? own : oth;
            m[2 + (k >> 2)][k & 3] = h0 ? oth : own;
        }
        float s[2][4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            s[0][j] = m[0][j] + m[1][j] + m[2][j];
            s[1][j] = m[1][j] - m[2][j] - m[3][j];
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
    };
    if (h0) finish(std::integral_constant<int, 0>{});
    else finish(std::integral_constant<int, 1>{});
}

// upk[chunk][r = 4 q + 2 h + part][cout/32][lane][j]: the A operands of cin 8 chunk + 4 (lane >> 5) + q for the four
// Winograd positions xi = 8 h + 4 part + j of cout = blk*32 + (lane & 31); value (G g G^T)[xi] * scale[cout], zero
// outside (cout, cin).  One record = what a wave half h needs for one cin step, as two lane-linear 1-KB pieces.
// G = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1]; float64 arithmetic.
__global__ void wino_pack_kernel(const float* __restrict__ w, const float* __restrict__ scale, float* __restrict__ upk,
                                 int cout, int cin, int nblk, long total) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int j = i & 3, lane = (i >> 2) & 63;
    long t = i >> 8;
    const int blk = t % nblk; t /= nblk;
    const int rec = t & 15;
    const int chunk = (int)(t >> 4);
    const int q = rec >> 2, hh = (rec >> 1) & 1, part = rec & 1;
    const int xi = 8 * hh + 4 * part + j;
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

template <int TWV, int EXP = 0>
int launch_wino(const WinoP& p, hipStream_t st) {
    constexpr int RW = 32 / TWV, PR = 2 * 4 * RW + 2, RS = TWV == 16 ? 48 : 68;
    constexpr int PBUF = (8 * PR * RS + 511) / 512 * 512;
    size_t lds = ((size_t)2 * PBUF + 2 * 16 * 256) * sizeof(float);
    const size_t xchg = (size_t)4 * 2 * 64 * 64 * sizeof(float);          // the pairs' accumulator exchange (128 KB)
    if (lds < xchg) lds = xchg;
    if (occd::ensure_big_lds(reinterpret_cast<const void*>(wino3x3_kernel<TWV, EXP>)) != OCCD_OK) return OCCD_ELAUNCH;
    hipLaunchKernelGGL((wino3x3_kernel<TWV, EXP>), dim3((unsigned)p.nwg), dim3(512), lds, st, p);
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
    const int exp_mode = a->tile_hint / 100;                  // development switches (see the kernel template)
    const int hint = a->tile_hint % 100;
    int twv = hint == 16 || hint == 32 ? hint : 0;
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
#ifdef OCCD_WINO_DEV_VARIANTS   // development A/B switches (wrong results by design): not compiled into the library
    if (exp_mode == 1) return twv == 16 ? launch_wino<16, 1>(p, (hipStream_t)stream) : launch_wino<32, 1>(p, (hipStream_t)stream);
    if (exp_mode == 2) return twv == 16 ? launch_wino<16, 2>(p, (hipStream_t)stream) : launch_wino<32, 2>(p, (hipStream_t)stream);
#else
    (void)exp_mode;
#endif
    return twv == 16 ? launch_wino<16>(p, (hipStream_t)stream) : launch_wino<32>(p, (hipStream_t)stream);
}

}  // extern "C"
