/*
 * occdepth_amd.h -- C-ABI of libocc_hip.so, the MI355X (gfx950) kernel library
 * under the `occdepth.models` nn.Module surface.
 *
 * The reference (megvii-research/OccDepth) has NO FFI layer of its own: its hot
 * path is a graph of stock ATen ops called from Python.
 * Each entry point below therefore names the reference *Python* call site whose
 * ATen-op sequence it replaces; `occdepth_amd/hip.py` is the ctypes binding.
 *
 * Conventions
 *  - plain pointers + ints only; all pointers are DEVICE pointers (fp32 unless
 *    noted) owned by the caller; nothing is retained after the call returns.
 *  - `stream` is a hipStream_t passed as void* (NULL = default stream).  Every
 *    launch is asynchronous on that stream; no call synchronises the device.
 *  - return value: 0 (OCCD_OK) or a negative OCCD_E* code; occd_strerror()
 *    gives text.  The Python layer converts non-zero into RuntimeError.
 *  - voxel tensors are channels-last: row r = ((b*X + x)*Y + y)*Z + z holds the
 *    C channels of one voxel, `cs` floats apart (cs >= C, multiple of 4), at
 *    channel offset `coff` inside the row (so a tensor can be a channel slice
 *    of a wider concat buffer).  Channel pads must hold zeros.
 */
#ifndef OCCDEPTH_AMD_H
#define OCCDEPTH_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OCCD_OK 0
#define OCCD_EINVAL (-1)   /* bad argument / unsupported geometry          */
#define OCCD_ELAUNCH (-2)  /* hip launch / runtime error                   */
#define OCCD_ENOMEM (-3)   /* tile does not fit LDS                        */

#define OCCD_MAX_VIEWS 4
#define OCCD_MAX_SCALES 4

#define OCCD_ACT_NONE 0
#define OCCD_ACT_RELU 1
#define OCCD_ACT_SIGMOID 2
#define OCCD_ACT_RELU_PRE 3 /* act_out only: relu(conv + bias) + res1 + res2 */

/* ABI version; bumped whenever a struct below changes (13: occd_gemm_args.bias_n / stride_bias_n, occd_gemm_f32x3_splitk). */
int occd_abi_version(void);
const char* occd_strerror(int code);

/* ------------------------------------------------------------------------ *
 * K2: implicit-GEMM 3-D convolution on fp32 MFMA (v_mfma_f32_32x32x2_f32),
 * fused epilogue:  out = act_out( conv(act_in(in)) + bias + res1 + res2 ).
 *
 * Replaces, in eval mode with BatchNorm folded into (w, bias):
 *   nn.Conv3d + BatchNorm3d + ReLU + residual adds   occdepth/models/DDR.py:111-139
 *   ASPP / segmentation-head conv chains             occdepth/models/modules.py:40-46,158-175
 *   AvgPool3d + 1x1x1 conv side branches             occdepth/models/DDR.py:95-109 (as k=stride convs)
 *   nn.ConvTranspose3d(k3,s2,p1,op1)+BN+ReLU         occdepth/models/modules.py:278-296
 *     (run as 8 sub-pixel phases through the o_stride/o_off output scatter)
 *   1x1x1 relation logits, sigmoid + torch.bmm       occdepth/models/CRP3D.py:68-82
 *
 * `wpk` is the packed weight image produced by occd_pack_weights():
 *   wpk[tap][kt][nt][lane][q],  lane = kk*32 + j,  value =
 *   W[cout = nt*32 + j][cin = kt*8 + kk*4 + q][tap], tap = (kx*KY + ky)*KZ + kz,
 *   zero padded to Cin8 = ceil8(Cin) and 32*NT = ceil32(Cout).
 * ------------------------------------------------------------------------ */
typedef struct occd_conv3d_args {
    const float* in;    /* (B, X, Y, Z, in_cs)                               */
    const float* wpk;   /* packed weights, see above                         */
    const float* bias;  /* (32*NT) floats or NULL                            */
    const float* res1;  /* optional residual, laid out like `out`            */
    const float* res2;  /* optional second residual                          */
    float* out;         /* (B, OX, OY, OZ, out_cs)                           */
    int32_t batch;
    int32_t X, Y, Z;            /* input spatial dims                        */
    int32_t cin;                /* logical input channels (weights K extent) */
    int32_t in_cs, in_coff;
    int32_t cout;               /* logical output channels                   */
    int32_t out_cs, out_coff;
    int32_t res1_cs, res1_coff;
    int32_t res2_cs, res2_coff;
    int32_t kx, ky, kz;         /* kernel extent                             */
    int32_t sx, sy, sz;         /* stride                                    */
    int32_t dx, dy, dz;         /* dilation                                  */
    int32_t px, py, pz;         /* leading zero padding                      */
    int32_t Xo, Yo, Zo;         /* number of output positions computed       */
    int32_t OX, OY, OZ;         /* spatial dims of the out/res buffers       */
    int32_t o_stride_x, o_stride_y, o_stride_z; /* output voxel = o*stride+off */
    int32_t o_off_x, o_off_y, o_off_z;
    int32_t act_in;             /* OCCD_ACT_* applied to `in` on load        */
    int32_t act_out;            /* OCCD_ACT_NONE, OCCD_ACT_RELU (after the residual
                                   adds) or OCCD_ACT_RELU_PRE (before them)  */
    int32_t cout_store;         /* channels written per voxel (>= cout,
                                   multiple of 4, <= 32*NT); pads get 0+bias */
    int32_t tile_hint;          /* 0 = auto; else forces a kernel variant    */
} occd_conv3d_args;

int occd_conv3d_fwd(const occd_conv3d_args* a, void* stream);

/* The n in {1, 2, 4, 8} sub-pixel phases of ONE transposed convolution as one launch
 * (ConvTranspose3d(k3, s2, p1, op1) of the reference's Upsample blocks, occdepth/models/modules.py:278-296,
 * = 8 phase convolutions over tap subsets scattering to interleaved voxels).  a[0..n) may differ ONLY in
 * wpk, kx / ky / kz and o_off_*; OCCD_EINVAL otherwise.  Same arithmetic as n calls of occd_conv3d_fwd.   */
int occd_conv3d_fwd_phases(const occd_conv3d_args* a, int32_t n, void* stream);

/* Pack PyTorch-layout conv weights (Cout, Cin, KX, KY, KZ) [transposed=0] or
 * conv-transpose weights (Cin, Cout, KX, KY, KZ) [transposed=1] into `wpk`,
 * multiplying output channel c by scale[c] when scale != NULL (BN fold).
 * For a dynamic GEMM B operand (CRP bmm) pass w as (K=Cin, N=Cout) row-major
 * with layout=2.  Returns the number of floats written, or <0.              */
int64_t occd_packed_weight_floats(int32_t cout, int32_t cin, int32_t taps);
int occd_pack_weights(const float* w, const float* scale, float* wpk,
                      int32_t cout, int32_t cin, int32_t kx, int32_t ky, int32_t kz,
                      int32_t layout, void* stream);

/* ------------------------------------------------------------------------ *
 * K1a: FLoSP-Depth frustum sample.  For every voxel of the (A,B,C) grid and
 * every camera: voxel centre -> camera -> image (u, v, LID depth bin) -> ida ->
 * normalise by (W_img-1, H_img-1, D-1) -> trilinear grid_sample(align_corners=
 * False, zeros) of the (D, h, w) depth-probability frustum, plus the same sample
 * of an all-ones volume; mean over cameras where the ones-sample is > 0.
 * Replaces occdepth/models/flosp_depth/flosp_depth.py:561-602,
 * f2v/frustum_grid_generator.py:70-152, f2v/sampler.py:59-64.
 * ------------------------------------------------------------------------ */
typedef struct occd_flosp_args {
    const float* depth;     /* (B, n_cams, D, h, w) softmaxed depth bins      */
    const float* trans;     /* (B, n_cams, 4, 4)  lidar_to_cam @ grid_to_lidar */
    const float* proj;      /* (B, n_cams, 3, 4)  cam_to_img                   */
    const float* ida;       /* (B, n_cams, 4, 4)                               */
    const float* grids;     /* infer_mode: (n_cams, B, A, B, C, 3) precomputed
                               normalised grids, else NULL                    */
    float* out;             /* (B, A*B*C) */
    int32_t batch, n_cams;
    int32_t D, h, w;
    int32_t A, Bdim, C;     /* voxel grid dims (voxel_num)                    */
    float img_w, img_h;     /* final_dim (W, H) used by the normalisation     */
    float depth_min, depth_max;
    int32_t mean_mode;      /* 1 = "mean" aggregation, 0 = "sum"              */
} occd_flosp_args;

int occd_flosp_sample_fwd(const occd_flosp_args* a, void* stream);

/* Transpose of the frustum sample (what autograd computes through f2v/sampler.py:59-64's
 * F.grid_sample in training_step): gdepth (B, n_cams, D, h, w) = d loss / d depth given gout (B, A*B*C) = d loss / d out.
 * fwd.depth and fwd.out are not read (the sample is linear in the volume).  DETERMINISTIC: contributions are summed as
 * 64-bit fixed point with a power-of-two scale derived from max |gout| (both reductions are order-free), unlike
 * grid_sampler_3d_backward's float atomics.  workspace: >= 8 * (B n_cams D h w) + 8 bytes, 8-byte aligned, no
 * initialisation needed.                                                                                          */
typedef struct occd_flosp_bwd_args {
    occd_flosp_args fwd;
    const float* gout;
    float* gdepth;
    void* workspace;
    int64_t workspace_bytes;
} occd_flosp_bwd_args;
int occd_flosp_sample_bwd(const occd_flosp_bwd_args* a, void* stream);

/* ------------------------------------------------------------------------ *
 * K1b: Stereo-SFA multi-scale lift.  Per voxel: for each 2-D scale gather the
 * projected pixel's C-vector from every view (mean over in-FOV pattern points),
 * fuse the views with the cosine-similarity weights, sum the scales, multiply by
 * depth_scale[voxel] * scale_const when depth_scale != NULL, write a
 * channels-last voxel row.
 * Replaces occdepth/models/SFA.py:12-106 called B x n_scales times from
 * occdepth/models/OccDepth.py:266-298, and the `x3ds * x3ds_depth * 100` of :339.
 * ------------------------------------------------------------------------ */
typedef struct occd_lift_args {
    /* feat[s][v]: (B, H_s*W_s, C) channels-last feature rows of view v, scale s */
    const float* feat[OCCD_MAX_SCALES][OCCD_MAX_VIEWS];
    int32_t feat_h[OCCD_MAX_SCALES], feat_w[OCCD_MAX_SCALES];
    int32_t feat_cs[OCCD_MAX_SCALES];     /* row stride (floats) of feat[s][*] */
    int32_t scale_div[OCCD_MAX_SCALES];   /* scale_2d: pixel // scale_div      */
    int32_t n_scales, n_views, batch;
    int32_t C;                            /* feature channels (multiple of 4)  */
    const int64_t* pix;                   /* (B, V, N, P, 2) int64 (x, y)      */
    const uint8_t* fov;                   /* (B, V, N, P) bool                 */
    int32_t N, P;
    const float* depth_scale;             /* (B, N) or NULL                    */
    float scale_const;                    /* 100.0 in the reference            */
    int32_t dimA, dimB, dimC;             /* n -> (a, b, c), n = (a*dimB+b)*dimC+c */
    int64_t row_a, row_b, row_c;          /* out row = a*row_a + b*row_b + c*row_c  */
    float* out;                           /* (B, rows, out_cs)                 */
    int64_t out_rows;                     /* rows per batch item               */
    int32_t out_cs;                       /* pads [C, out_cs) are zero-filled  */
    /* batch stride (floats) of feat[s][v]; 0 = dense (feat_h * feat_w * feat_cs).  Lets a view of a tensor that
     * interleaves the views (B, V, H, W, cs) be gathered in place.                                             */
    int64_t feat_bstride[OCCD_MAX_SCALES][OCCD_MAX_VIEWS];
    /* workgroup -> XCD placement of the single-pattern-point kernel (speed only, any value is correct):
     * 0 = dispatch order (columns of voxels dealt round-robin to the 8 XCDs), 1 = every XCD a contiguous range of the
     * flat voxel index, 2 = every XCD a contiguous (b, c) range inside every a-slab (needs dimB*dimC*LPV % 2048 == 0,
     * else falls back to 0).                                                                                    */
    int32_t xcd_mode;
} occd_lift_args;

int occd_lift_fwd(const occd_lift_args* a, void* stream);

/* ------------------------------------------------------------------------ *
 * K14: one stride-1 DDR Bottleneck3D (occdepth/models/DDR.py:111-139 with BatchNorm folded) in two launches:
 *   o1 = relu(W1 x + b1); o2 = conv_z(o1) + b2; o3 = conv_y(relu(o2)) + b3 + o2; o4 = conv_x(relu(o3)) + b4 + o2 + o3;
 *   y = relu(W5 relu(o4) + b5 + x).       x, y: channels-last (B, X, Y, Z, cs) rows; C = channels of x and y, P = planes.
 * w: occd_bottleneck3d_weight_floats(C, P) floats = F(W1^T [C][P]) | b1 [P] | F(W2[0]) F(W2[1]) F(W2[2]) (per tap, [P in][P
 *    out]) | b2 | F(W3[.]) | b3 | F(W4[.]) | b4 | F(W5^T [P][C]) | b5 [C]   (tap k <-> offset (k - 1) * dilation), where
 *    F(W [KIN][MOUT]) is the matrix in MFMA fragment order: F[((t * MOUT/16 + m) * 64 + lane) * 4 + e] =
 *    W[16 t + 4 (lane >> 4) + e][16 m + (lane & 15)].   o2: workspace of B*X*Y*Z*P floats.
 * P in {16, 32}, C a multiple of 16, Z in {4, 8, 16} (a 16-voxel MFMA column block holds whole Z columns).
 * d0 / d1 / d2: dilation of the Z / Y / X convolution.
 * ------------------------------------------------------------------------ */
typedef struct occd_bneck_args {
    const float* x;
    float* y;
    float* o2;
    const float* w;
    int32_t batch, X, Y, Z, C, P;
    int32_t x_cs, x_coff, y_cs, y_coff;
    int32_t d0, d1, d2;
} occd_bneck_args;
int64_t occd_bottleneck3d_weight_floats(int32_t C, int32_t P);
int occd_bottleneck3d_fwd(const occd_bneck_args* a, void* stream);

/* ------------------------------------------------------------------------ *
 * K15: pointwise (1x1x1) convolution / row GEMM on channels-last voxel rows:
 *   out[r][n] = act_out( sum_k act_in(a[r][k]) * w[k][n] + bias[n] (+ res[r][n]) ),   r < rows, k < K, n < N.
 * a / out / res: float32 rows with strides *_cs and channel offsets *_coff (a Vox or a channel slice of one); w: the
 * image occd_rows_gemm_pack makes of a dense [K][N] matrix (the (cin, cout) transpose of a 1x1x1 convolution weight with
 * the BatchNorm scale folded, or the mega-voxel rows of the CRP product, occdepth/models/CRP3D.py:80): MFMA fragment
 * records of occd_rows_gemm_packed_floats(K, N) floats.  K a multiple of 16, N a multiple of 4; w_stride is unused.
 * act_in: NONE / RELU / SIGMOID; act_out: NONE / RELU (after the residual) / RELU_PRE (before it).
 * ------------------------------------------------------------------------ */
typedef struct occd_rows_gemm_args {
    const float* a;
    const float* w;
    const float* bias;      /* N floats, padded to a multiple of 4, or NULL */
    const float* res;       /* or NULL */
    float* out;
    int64_t rows;
    int32_t K, N;
    int32_t a_cs, a_coff, out_cs, out_coff, res_cs, res_coff, w_stride;
    int32_t act_in, act_out;
} occd_rows_gemm_args;
int occd_rows_gemm_fwd(const occd_rows_gemm_args* a, void* stream);
int64_t occd_rows_gemm_packed_floats(int32_t K, int32_t N);
int occd_rows_gemm_pack(const float* w, float* wpk, int32_t K, int32_t N, int32_t w_stride, void* stream);

/* K16 (csrc/gemm_x3.hip): row-major float32 GEMM with the 3-way bf16 split of both operands (float32-level accuracy on the
 * bf16 matrix pipe):  C[b][m][n] = act(sum_k A[b][m][k] B[b][k][n] + bias[m]).  Replaces the library GEMMs behind the
 * reference's 2-D network call sites as this repo restates them (occdepth/models/unet2d.py:24-46 -- the first convolution of
 * a decoder level as nine low-resolution tap GEMMs --, :137-165 -- the Winograd-domain products of the second convolutions
 * at 1/8 and 1/16 --, and the geffnet MBConv expand 1x1 convolutions): torch.matmul / torch.bmm / F.conv2d there.
 * A: M x K with k contiguous (lda >= K, K % 8 == 0, 16-byte aligned rows); B: K x N with n contiguous (ldb >= N, N >= 4, any
 * dword alignment); C: M x N (ldc >= N).  stride_a == 0 shares A over the batch (weights).  bias: M floats or NULL.
 * act: OCCD_GEMM_ACT_NONE / _SWISH / _LEAKY (slope).  tile_hint: 0 = choose, 1 .. 5 = force a tile variant, 6 = the
 * wave-specialised 256 x 128 kernel, 7 = the 64 x 64 tile with the in-workgroup split-K, 8 = the panel-stationary kernel K16p
 * (pre = 1, K <= 848, no res / scale_k / act_a: a workgroup keeps 64 or 32 B columns over the whole K split in LDS and walks
 * the row tiles of the pre-split A image against them; hint 0 picks it for M >= 256) (tests / A-B).
 * Rows of C that start on 128-byte boundaries (ldc % 32 == 0, aligned base) are written 2.2x faster than rows 8 bytes off a
 * cache line (profiles/r05_store_alignment.txt): give intermediate results a padded ldc where the consumer takes a stride. */
#define OCCD_GEMM_ACT_NONE 0
#define OCCD_GEMM_ACT_SWISH 1
#define OCCD_GEMM_ACT_LEAKY 2
typedef struct occd_gemm_args {
    const float* A;
    const float* B;
    float* C;
    const float* bias;
    int32_t M, N, K, batch;
    int64_t lda, ldb, ldc;                 /* elements */
    int64_t stride_a, stride_b, stride_c;  /* elements between consecutive batch items */
    int32_t act;
    float slope;
    int32_t tile_hint;
    int32_t pre;                           /* 0: A, B float32; 1: A = occd_gemm_x3_pack(role 0) image; 2: B = role-1 image */
    const float* res;                      /* optional (ABI 9): C += res[b][m][n] after bias / activation, laid out like C
                                              (ldc, stride_c): the skip connection of an MBConv project convolution      */
    const float* scale_k;                  /* optional (ABI 9): (batch, K) floats, B[b][k][:] is multiplied by scale_k[b][k]
                                              (rounded to float32, as the reference's x * gate) while it is staged: the
                                              squeeze-excite gate in front of the project convolution; pre must not be 2   */
    int32_t act_a;                         /* (ABI 11) 1: A[b][m][k] -> sigmoid(A[b][m][k]) while it is staged -- the relation
                                              products torch.bmm(sigmoid(P_logits), mega) of occdepth/models/CRP3D.py:80 as ONE
                                              batched launch over the relations; float32 A only (pre 0 / 2)                 */
    const float* bias_n;                   /* (ABI 13) optional: + bias_n[b][n] (one value per COLUMN) ahead of the activation --
                                              products whose ROWS are voxels and whose columns are output channels, e.g. the
                                              relation-logit 1x1x1 convolutions of occdepth/models/CRP3D.py:54-62 as one batched
                                              launch over the relations; occd_gemm_f32x3 only, not with tile_hint 6 / 8          */
    int64_t stride_bias_n;                 /* floats between the batch items of bias_n (0: shared)                              */
} occd_gemm_args;
int occd_gemm_f32x3(const occd_gemm_args* a, void* stream);
/* K21 (round 6), the skinny long-K GEMMs -- the MBConv project convolutions of the 1/16 and 1/32 EfficientNet stages,
 * `bn3(conv_pwl(x * se_gate)) + skip` (geffnet InvertedResidual behind occdepth/models/unet2d.py:188-196): K16p's
 * panel-stationary loop with K cut into `nz` chunks over the grid's z dimension, float32 partial tiles in `workspace`, and a
 * second launch that sums the chunks in index order (deterministic) and applies bias / activation / res.  Same operand
 * meaning as occd_gemm_f32x3 with pre = 1 (A = the role-0 image of occd_gemm_x3_pack; lda ignored), B float32 n-contiguous,
 * scale_k / res / bias / act optional; tile_hint and act_a must be 0.  occd_gemm_f32x3_splitk_plan proposes
 * (k16_per_z, nz, row_ranges) and the workspace size in floats (128-byte aligned buffer).  OCCD_EINVAL when a chunk would
 * not fit LDS (k16_per_z * 16 * 192 B <= 160 KB), the chunks do not tile K exactly, or the workspace is too small.        */
int occd_gemm_f32x3_splitk_plan(int32_t M, int32_t N, int32_t K, int32_t batch, int32_t* k16_per_z, int32_t* nz,
                                int32_t* row_ranges, int64_t* workspace_floats);
int occd_gemm_f32x3_splitk(const occd_gemm_args* a, int32_t k16_per_z, int32_t nz, int32_t row_ranges, float* workspace,
                           int64_t workspace_floats, void* stream);
/* K16t, the "NT" form: C[b][m][n] = sum_k A[b][m][k] B[b][n][k], BOTH operands with k contiguous (lda, ldb >= K), any dword
 * alignment, any K: autograd's weight gradient of a pointwise convolution, dW = gy (Cout x HW) . x^T, on NCHW tensors as they
 * lie (training step, the geffnet MBConv 1x1 convolutions).  The reduction dimension is split over `act` (>= 1) workgroup
 * groups -- a weight gradient is a small matrix reduced over many pixels --: C receives batch x act partial matrices
 * (batch-major, stride_c elements apart) which the caller sums; occd_gemm_f32x3_nt_splits proposes `act` (every split must own
 * at least one 32-k step: pass exactly what it returns, or 1).  bias / pre must be NULL / 0; tile_hint 0 / 1 / 2.            */
int occd_gemm_f32x3_nt(const occd_gemm_args* a, void* stream);
int32_t occd_gemm_f32x3_nt_splits(int32_t M, int32_t N, int32_t K, int32_t batch);
/* A static operand (weights) split once into its three bf16 terms in MFMA fragment order, so the GEMM reads it straight
 * from L2 (no LDS, no split arithmetic): role 0 = an A operand (rows x K, k contiguous, ld >= K), role 1 = a B operand
 * (K x rows, "row" = column index contiguous, ld >= rows).  `out`: occd_gemm_x3_packed_elems(rows, K) bf16 per batch item,
 * consecutive; in_stride: float elements between the batch items of `w`.  With pre = 1 / 2 the packed operand's lda / ldb is
 * ignored and stride_a / stride_b counts bf16 elements (0 = shared over the batch).                                          */
int64_t occd_gemm_x3_packed_elems(int32_t rows, int32_t K);
int occd_gemm_x3_pack(const float* w, void* out, int32_t rows, int32_t K, int64_t ld, int32_t role, int32_t batch,
                      int64_t in_stride, void* stream);

/* The eval lift without its tables (SURVEY 8(f) N2 fused into K1b): the kernel projects every voxel
 * centroid itself (the arithmetic of occd_project_voxels: occdepth/data/utils/helpers.py:94-169, integer-exact), samples the
 * FLoSP depth frustum for the voxel (the arithmetic of occd_flosp_sample_fwd: flosp_depth.py:561-602) and applies
 * `* depth * scale_const` (OccDepth.py:339) -- no (B, V, N, 1, 2) int64 `projected_pix`, no `fov_mask`, no (B, N)
 * depth vector in HBM.  Single-point pattern (pattern_id 0), power-of-two scales and grid dims, <= 2 views.
 * lift.pix / lift.fov / lift.depth_scale are ignored; lift.dimA/B/C = the voxel grid (X, Y, Z) of the projection.
 * frustum.depth == NULL: no depth scaling (trans_2d_to_3d = "flosp"); frustum.out is ignored.                        */
typedef struct occd_lift_proj_args {
    occd_lift_args lift;
    const double* cam_E;    /* DEVICE (B, V, 4, 4) float64 extrinsics (lidar -> camera), the batch's T_velo_2_cam       */
    const double* cam_k;    /* DEVICE (B, V, 3, 3) float64 intrinsics, the batch's cam_k; fx, fy, cx, cy are rounded to
                               float32 inside the kernel, like the dataloader does (fusion.py:336-337)                */
    double voxel_size;      /* metres per voxel of the lifted grid (0.2 * project_scale)                              */
    float origin[3];        /* float32(vox_origin)                                                                    */
    int32_t img_w, img_h;
    occd_flosp_args frustum;
} occd_lift_proj_args;
int occd_lift_proj_fwd(const occd_lift_proj_args* a, void* stream);

/* Backward of occd_lift_fwd for single-point patterns (P = 1): what autograd computes through
 * SFA.forward x 4 scales and the `* depth * 100` of occdepth/models/OccDepth.py:266-298,339 in training_step.
 * gout (B, out_rows, out_cs) = d loss / d out, same rows as the forward output.  gfeat[s][v]: gradient maps with the
 * layout (and batch stride) of fwd.feat[s][v], ZEROED by the caller, accumulated with float atomics (summation order,
 * hence the last bits, vary from run to run).  gdepth (B, N) = d loss / d depth_scale, or NULL.                   */
typedef struct occd_lift_bwd_args {
    occd_lift_args fwd;
    const float* gout;
    float* gfeat[OCCD_MAX_SCALES][OCCD_MAX_VIEWS];
    float* gdepth;
} occd_lift_bwd_args;
int occd_lift_bwd(const occd_lift_bwd_args* a, void* stream);

/* voxel centroid -> pixel projection on the GPU instead of the dataloader's numba
 * `vox2pix` (occdepth/data/utils/helpers.py:94-169, fusion.py:203-217,336-337,518-522), pattern_id 0.
 * cam_E (4x4 row-major), cam_k (3x3), vox_origin (3) are small HOST arrays of doubles; outputs are device
 * tensors shaped like the batch entries `projected_pix_{s}` (N, 1, 2) int64 and `fov_mask_{s}` (N, 1) bool,
 * N = X*Y*Z voxels in (x, y, z) order; pix_z (N) float32 is optional.                              */
int occd_project_voxels(const double* cam_E_host, const double* cam_k_host, const double* vox_origin_host,
                        double voxel_size, int32_t X, int32_t Y, int32_t Z, int32_t img_w, int32_t img_h,
                        int64_t* pix, uint8_t* fov, float* pix_z, void* stream);

/* ------------------------------------------------------------------------ *
 * Layout / small fused helpers around the two big kernels.
 * ------------------------------------------------------------------------ */
/* (B, C, S) channel-first -> (B, S, cs) channels-last rows, zero pad [C, cs).
 * Feeds K1 from the torch 2-D decoder output (occdepth/models/unet2d.py:148-163)
 * and K2 from externally supplied (B,C,X,Y,Z) tensors.                        */
int occd_nchw_to_nhwc(const float* in, float* out, int32_t batch, int32_t C,
                      int64_t S, int32_t cs, void* stream);
/* inverse: (B, S, cs)[.., coff:coff+C] -> (B, C, S) */
int occd_nhwc_to_nchw(const float* in, float* out, int32_t batch, int32_t C,
                      int64_t S, int32_t cs, int32_t coff, void* stream);
/* softmax over n channels at [src_coff, src_coff+n) of each row, written to
 * [dst_coff, dst_coff+n) of the same-shaped dst rows, followed by dst_pad
 * zeros (cascade head: occdepth/models/modules.py:168-171 softmax + torch.cat). */
int occd_softmax_channels(const float* src, float* dst, int64_t rows,
                          int32_t src_cs, int32_t src_coff, int32_t dst_cs,
                          int32_t dst_coff, int32_t n, int32_t dst_pad, void* stream);

/* Cascade-head tail (occdepth/models/modules.py:166-173, after splitting conv_classes by linearity):
 *   out[v][o] = part[v][o] + sum_{tap<27, c<2} softmax(part[v+tap][occ_off : occ_off+2])[c] * wn[o][c][tap]
 * part: (B, X, Y, Z, part_cs) rows with the wide-half class logits at [0, nbr) and the 2 occupancy logits at
 * occ_off; wn: (nbr, 2, 3, 3, 3) = conv_classes.weight[:, planes:]; out: (B, X, Y, Z, out_cs); zero padding
 * applies to the softmax output.  VALU kernel (K = 54 is too thin for MFMA).                        */
int occd_cascade_tail_fwd(const float* part, const float* wn, float* out, int32_t batch, int32_t X,
                          int32_t Y, int32_t Z, int32_t part_cs, int32_t occ_off, int32_t out_cs,
                          int32_t nbr, void* stream);

/* ------------------------------------------------------------------------ *
 * 2-D NCHW fused memory-bound helpers for the 2-D UNet (first step).
 * act: 0 none, 1 relu, 2 swish (x * sigmoid(x)), 3 leaky relu with `slope`.
 * ------------------------------------------------------------------------ */
/* y = act(x * scale[c] + shift[c]) + res   (res_first = 0)
 * y = act(x * scale[c] + shift[c] + res)   (res_first = 1)      x, res, y: (B, C, S); may alias.
 * BatchNorm2d(eval) + Swish / LeakyReLU / ReLU (+ residual) of the EfficientNet blocks
 * (third-party geffnet), unet2d.py:26-36 (conv+BN+LeakyReLU) and the DepthNet BasicBlocks
 * (flosp_depth.py:211-222).                                                  */
int occd_affine_act_nchw(const float* x, const float* res, float* y, const float* scale,
                         const float* shift, int32_t batch, int32_t C, int64_t S, int32_t act,
                         float slope, int32_t res_first, void* stream);

/* Backward of swish(x) = x * sigmoid(x) (the EfficientNet activation, geffnet behind occdepth/models/unet2d.py:175-190) in
 * one pass: gx = gy * s * (1 + x * (1 - s)), s = sigmoid(x); n contiguous floats, 16-byte aligned pointers.  The forward
 * is occd_affine_act_nchw with act = 2.                                                                             */
int occd_swish_bwd(const float* x, const float* gy, float* gx, int64_t n, void* stream);
/* y = softmax(x, dim=1) of a (B, C, S) map (S = H*W): the depth-bin softmax of FlospDepth.forward
 * (occdepth/models/flosp_depth/flosp_depth.py:548).                                         */
int occd_softmax_nchw(const float* x, float* y, int32_t batch, int32_t C, int64_t S, void* stream);
/* depthwise k x k (k = 3 or 5) convolution, x (B, C, H, W), w (C, 1, k, k), explicit top/left
 * zero padding (TensorFlow SAME), y (B, C, Ho, Wo) = act(conv * scale[c] + shift[c]).   */
int occd_dwconv2d_nchw(const float* x, const float* w, const float* scale, const float* shift,
                       float* y, int32_t batch, int32_t C, int32_t H, int32_t W, int32_t k,
                       int32_t stride, int32_t pad_top, int32_t pad_left, int32_t Ho, int32_t Wo,
                       int32_t act, void* stream);
/* Encoder stem (round 5; geffnet conv_stem + bn1 + act1 behind occdepth/models/unet2d.py:175-190): 3x3 convolution of the
 * 3-channel image x (B, 3, H, W) with w (cout, 3, 3, 3), stride 1 or 2, explicit top/left zero padding (TensorFlow SAME),
 * y (B, cout, Ho, Wo) = act(conv * scale[c] + shift[c]); act: 0 none, 1 relu, 2 swish.                              */
int occd_stem_conv3x3_nchw(const float* x, const float* w, const float* scale, const float* shift, float* y,
                           int32_t batch, int32_t H, int32_t W, int32_t cout, int32_t stride, int32_t pad_top,
                           int32_t pad_left, int32_t Ho, int32_t Wo, int32_t act, void* stream);
/* out (B, C + Cskip, H, W): [:, :C] = bilinear resize of x (B, C, h, w) with align_corners=True,
 * [:, C:] = skip (B, Cskip, H, W)  -- F.interpolate + torch.cat of unet2d.py:38-46.          */
int occd_upsample_bilinear_cat_nchw(const float* x, const float* skip, float* out, int32_t batch,
                                    int32_t C, int32_t Cskip, int32_t h, int32_t w, int32_t H,
                                    int32_t W, void* stream);
/* Channels-last twins for the bf16-mode training step (decoder levels as (B, H, W, C) pixel rows): the same resize + concat
 * with x (B, h, w, C), skip (B, H, W, Cskip), out (B, H, W, C + Cskip) dense rows, and the backward of the resized part as a
 * gather: gx (B, h, w, C) from gout (B, H, W, gout_cs) rows (its first C channels), the exact transpose of the forward.    */
int occd_upsample_bilinear_cat_nhwc(const float* x, const float* skip, float* out, int32_t batch, int32_t C, int32_t Cskip,
                                    int32_t h, int32_t w, int32_t H, int32_t W, void* stream);
/* Round 6: the same with output rows of out_cs >= C + Cskip floats, pad lanes written as zeros -- rows of ceil8(channels)
 * floats are consumed in place by the convolution kernels (the 163-channel concat of the full-resolution decoder level in
 * rows of 168: no strided copy into padded rows in front of the level's first convolution).                              */
int occd_upsample_bilinear_cat_nhwc_rows(const float* x, const float* skip, float* out, int32_t batch, int32_t C, int32_t Cskip,
                                         int32_t h, int32_t w, int32_t H, int32_t W, int32_t out_cs, void* stream);
int occd_upsample_bilinear_nhwc_bwd(const float* gout, float* gx, int32_t batch, int32_t C, int32_t gout_cs, int32_t h,
                                    int32_t w, int32_t H, int32_t W, void* stream);

/* conv3x3(pad 1)(bilinear_up(x))  ==  sum_t shift_t(bilinear_up(W_t . x)): given z (B, 9 * Cout, h, w) = the nine per-tap
 * channel mixings of the LOW-resolution map (one pointwise GEMM, tap t = ky * 3 + kx in channel block t), writes
 * out (B, Cout, H, W) = sum_t [inside] bilinear_{align_corners=True}(z_t)(oy + ky - 1, ox + kx - 1).  Replaces the
 * upsampled-channel part of the first convolution of a decoder level (occdepth/models/unet2d.py:24-46) -- the
 * (Cup + Cskip)-channel upsample+concat tensor is never formed.  z[b][ch][y][x] sits at b * z_batch_stride +
 * ch * z_channel_stride + y * w + x floats (0 = dense NCHW); (h*w*B, h*w) is the layout of ONE GEMM over all images.  */
int occd_upconv_gather_nchw(const float* z, float* out, int32_t batch, int32_t Cout, int32_t h, int32_t w,
                            int32_t H, int32_t W, int64_t z_channel_stride, int64_t z_batch_stride, void* stream);

/* The same with the level's remaining pieces fused, for <= 4 skip channels (the 1/1 level: the raw image):
 *   out = leaky_relu(gather(z) + conv3x3(skip, wskip, pad 1) + shift[co], slope)
 * skip (B, Cs, H, W), wskip (Cout, Cs, 3, 3) with the BatchNorm scale folded in, shift (Cout): the finished
 * `LeakyReLU(BN(conv(cat[up(x), skip])))` of `UpSampleBN` (unet2d.py:24-46) in one launch after the tap GEMM.     */
int occd_upconv_gather_skip_nchw(const float* z, const float* skip, const float* wskip, const float* shift, float* out,
                                 int32_t batch, int32_t Cout, int32_t Cs, int32_t h, int32_t w, int32_t H, int32_t W,
                                 int64_t z_channel_stride, int64_t z_batch_stride, float slope, void* stream);

/* Winograd F(2x2, 3x3) transforms for nn.Conv2d(k=3, s=1, p=1) of the 2-D decoder
 * (occdepth/models/unet2d.py:24-46).  T = B * ceil(H/2) * ceil(W/2) tiles, tile (b, ty, tx) -> row (b*th + ty)*tw + tx.
 *   input : x (B, Cin, H, W) -> V (16, T, Cin),  V[4i+j] = (B^T d B)[i][j] of the 4x4 patch at (2ty-1, 2tx-1), zero padded
 *   (caller): M[xi] (T, Cout) = V[xi] (T, Cin) . U[xi] (Cin, Cout),  U[4i+j][ci][co] = (G g[co][ci] G^T)[i][j]
 *   output: M (16, T, Cout) -> y (B, Cout, H, W) = act(scale[c] * (A^T M A) + shift[c]) (+ res before or after act;
 *           act codes of occd_affine_act_nchw).
 * A call covers the STRIP of tile rows [ty0, ty0 + ths) of every image (ths = ceil(H/2), ty0 = 0: the whole image):
 * T = B * ths * tw, row (b*ths + ty - ty0)*tw + tx.  Strips keep V / M of the high-resolution levels cache-sized.  */
int occd_wino_input_transform_nchw(const float* x, float* V, int32_t batch, int32_t Cin, int32_t H, int32_t W,
                                   int32_t ty0, int32_t ths, void* stream);
int occd_wino_output_transform_nchw(const float* M, const float* scale, const float* shift, const float* res,
                                    float* y, int32_t batch, int32_t Cout, int32_t H, int32_t W, int32_t ty0,
                                    int32_t ths, int32_t act, float slope, int32_t res_first, void* stream);

/* K10: FUSED Winograd F(2x2, 3x3) convolution on the fp32 matrix pipe -- replaces
 * nn.Conv2d(k=3, s=1, p=1) + BatchNorm2d (eval) + activation of the 2-D decoder (occdepth/models/unet2d.py:24-46)
 * and the 3x3 convolutions of DepthNet / BasicBlock (occdepth/models/flosp_depth/flosp_depth.py:201-257):
 *   y = act( conv3x3(x, g * scale[co], pad 1) + shift[co] ) (+ res before or after act), x / y / res NCHW float32.
 * V = B^T d B is formed in registers from LDS-staged input patches, M stays in the MFMA accumulators, the output
 * transform runs in the epilogue: neither exists in memory (cf. the unfused transforms above).
 * upk: occd_wino_pack_weights(g (Cout, Cin, 3, 3), scale or NULL) -> occd_wino_packed_floats(Cout, Cin) floats,
 *      U = G g G^T * scale in MFMA A-fragment order [ceil(Cin/8)][16][ceil(Cout/32)][64 lanes][4].
 * shift: Cout floats or NULL.  act: codes of occd_affine_act_nchw (0 none, 1 relu, 2 swish, 3 leaky(slope)).
 * tile_hint: 0 = choose; 16 / 32 = tiles per wave along x (wave = 2 x 16 or 1 x 32 tiles).                     */
typedef struct occd_wino_args {
    const float* x;
    const float* upk;
    const float* shift;
    const float* res;
    float* y;
    int32_t batch, cin, cout, H, W;
    int32_t act, res_first, tile_hint;
    float slope;
} occd_wino_args;
int64_t occd_wino_packed_floats(int32_t cout, int32_t cin);
int occd_wino_pack_weights(const float* w, const float* scale, float* upk, int32_t cout, int32_t cin, void* stream);
int occd_wino_conv3x3_fwd(const occd_wino_args* a, void* stream);

/* Backward of the depthwise convolution (autograd of the geffnet conv_dw layers in training_step):
 *   data  : dx (B, C, H, W) from gy (B, C, Ho, Wo) and w (C, 1, k, k), same geometry arguments as the forward;
 *   weight: dw (C, 1, k, k) from x and gy; workspace: occd_dwconv2d_bwd_weight_workspace_floats(...) floats
 *           (per-workgroup partial sums, reduced in index order: deterministic).                                    */
int occd_dwconv2d_bwd_data_nchw(const float* gy, const float* w, float* dx, int32_t batch, int32_t C, int32_t H,
                                int32_t W, int32_t k, int32_t stride, int32_t pad_top, int32_t pad_left, int32_t Ho,
                                int32_t Wo, void* stream);
int64_t occd_dwconv2d_bwd_weight_workspace_floats(int32_t batch, int32_t C, int32_t k, int32_t Ho, int32_t Wo);
int occd_dwconv2d_bwd_weight_nchw(const float* x, const float* gy, float* dw, float* workspace, int32_t batch, int32_t C,
                                  int32_t H, int32_t W, int32_t k, int32_t stride, int32_t pad_top, int32_t pad_left,
                                  int32_t Ho, int32_t Wo, void* stream);
/* Depthwise convolution as above that ALSO leaves the squeeze-excite pooling behind: pool_part
 * (B*C, occd_dwconv2d_pool_blocks(Ho, Wo)) holds each workgroup's share of sum_{y,x} y[b][c] (fixed summation order).
 * occd_se_gate turns the partials into the gate  sigmoid(W_e swish(W_r mean + b_r) + b_e)  (B, C) of geffnet's
 * SqueezeExcite (conv_reduce (Cr, C), conv_expand (C, Cr)); r_scratch: B * Cr floats.  The gate is consumed by
 * occd_pw_conv_fwd's `gate` operand: x * gate never exists in memory.
 * x_plane_stride (ABI 12): floats between consecutive (b, c) input planes, 0 = H * W (dense); > H * W when x is the result of
 * an expand GEMM whose rows were put on a 128-byte pitch (rows of an odd pixel count are written 2.2x slower).            */
int32_t occd_dwconv2d_pool_blocks(int32_t Ho, int32_t Wo);
int occd_dwconv2d_pool_nchw(const float* x, const float* w, const float* scale, const float* shift, float* y,
                            float* pool_part, int32_t batch, int32_t C, int32_t H, int32_t W, int32_t k,
                            int32_t stride, int32_t pad_top, int32_t pad_left, int32_t Ho, int32_t Wo,
                            int32_t act, int64_t x_plane_stride, void* stream);
/* Squeeze-excite in TRAINING (autograd of geffnet's SqueezeExcite in training_step):
 *   forward : sums = occd_plane_reduce(x, NULL)  ->  occd_se_gate(sums, ..., nblk = 1, S)  ->  occd_affine_act_nchw (x * gate)
 *   backward: gg = occd_plane_reduce(gout, x)    ->  occd_se_bwd                           ->  occd_affine_act_nchw
 *             (gx = gout * gate + gm / S)
 * occd_plane_reduce: out[p] = sum_s a[p][s] (* b[p][s] when b != NULL) over `planes` dense planes of S floats, fixed order.
 * occd_se_bwd: from gg (B, C), the forward's gate (B, C), sums (B, C) and r = swish(Wr m + br) (B, Cr; occd_se_gate's
 * r_scratch), m = sums / S:  ds = gg g (1 - g);  gr = ds We;  z = Wr m + br;  dz = gr swish'(z)  and
 *   gm (B, C) = dz Wr,  gw_reduce (Cr, C) = dz^T m,  gb_reduce (Cr) = sum_b dz,  gw_expand (C, Cr) = ds^T r,  gb_expand (C) =
 *   sum_b ds;  dz_scratch: B * Cr floats.  batch <= 16.                                                                    */
int occd_plane_reduce(const float* a, const float* b, float* out, int64_t planes, int64_t S, void* stream);
int occd_se_bwd(const float* gg, const float* gate, const float* sums, const float* r, const float* w_reduce,
                const float* b_reduce, const float* w_expand, float* dz_scratch, float* gm, float* gw_reduce,
                float* gb_reduce, float* gw_expand, float* gb_expand, int32_t batch, int32_t C, int32_t Cr, int64_t S,
                void* stream);
/* DepthNet's camera-aware squeeze-excite gate in one launch (round 5; occdepth/models/flosp_depth/flosp_depth.py:201-257:
 * `Mlp(1, C, C)` of the scaled pixel size -> `SELayer(C)`):
 *   gate[i][c] = sigmoid(We relu(Wr (W2 relu(w1 s_i + b1) + b2) + br) + be),   i = 0 .. images-1
 * s_i = sps[i] (infer_mode) or factor * |(1 / K_i[0][0], 1 / K_i[1][1])| from `intrins` (row-major matrices with >= 6
 * floats per image, `intr_stride` floats apart: element [1][1] at offset 5 -- the (B, n_cams, 4, 4) intrinsics of the
 * batch); exactly one of sps / intrins is non-NULL.  w1 (C), w2 / wr / we (C, C) row major, biases (C).          */
int occd_depthnet_gate(const float* sps, const float* intrins, int64_t intr_stride, float factor, const float* w1,
                       const float* b1, const float* w2, const float* b2, const float* wr, const float* br,
                       const float* we, const float* be, float* gate, int32_t images, int32_t C, void* stream);
int occd_se_gate(const float* pool_part, const float* w_reduce, const float* b_reduce, const float* w_expand,
                 const float* b_expand, float* r_scratch, float* gate, int32_t batch, int32_t C, int32_t Cr,
                 int32_t nblk, int64_t S, void* stream);
/* Round 6: an opt-in ONE-launch form of occd_se_gate (reduce + expand with an in-kernel hand-off through self-validating
 * agent-scope words: csrc/se2d.hip se_fused_kernel) for C <= 4096, Cr % 4 == 0, Cr <= 192, batch * Cr <= 4096 -- bit-identical
 * to the two launches (r_scratch receives the squeezed activations in both forms: occd_se_bwd reads them).  Measured equal in
 * the frame and slower per launch (14.5 against 11.8 us), so the default stays two launches; this switch (or OCCD_SE_FUSED=1
 * in the environment) selects the one-launch kernel; returns the previous setting.                                       */
int32_t occd_se_gate_set_fused(int32_t on);

/* Round 6 (ABI 14): captured-graph hygiene.  `graph` is a hipGraph_t that stream capture produced and that has NOT been
 * instantiated yet (torch.cuda.CUDAGraph(keep_graph=True).raw_cuda_graph()).  Every MEMSET node is replaced by a KERNEL node
 * with the same destination, value, extent and edges: on ROCm 7.2 / gfx950 a captured hipMemsetAsync fills with its value on
 * the first launch of the instantiated graph only and with a stale pattern afterwards (tools/probe_graph_memset.py), which
 * leaves ATen's multi-block reductions -- they zero their semaphores that way -- without output on replays.  No reference
 * counterpart (scripts/train.py:176-206 launches the step eagerly).  Returns the number of nodes replaced (>= 0) or a
 * negative OCCD_E* code; element sizes 1 / 2 / 4, flat and pitched extents.                                              */
int occd_graph_replace_memsets(void* graph);

/* K11: pointwise (1x1) convolution on NCHW maps as a GEMM on the fp32 matrix pipe with the
 * EfficientNet / decoder epilogue fused -- replaces conv1x1 + BatchNorm2d (eval) + Swish (+ squeeze-excite gate on the
 * input, + MBConv skip add) of the geffnet blocks behind occdepth/models/unet2d.py:175-190, the `resize_output_1_s`
 * convolutions of DecoderBN (unet2d.py:137-165) and DepthNet.depth_pred (flosp_depth/flosp_depth.py:225-227):
 *   y[b][co][n] = act( sum_ci (w[co][ci] * scale[co]) * (x[b][ci][n] * gate[b][ci]) + shift[co] ) (+ res[b][co][n])
 * x (B, Cin, N), y / res (B, Cout, N), N = H*W; gate (B, Cin) or NULL; shift (Cout) or NULL; act codes as above.
 * wpk: occd_pw_pack_weights(w (Cout, Cin), scale or NULL) -> occd_pw_packed_floats(Cout, Cin) floats in MFMA
 *      A-fragment order [ceil(Cin/8)][ceil(Cout/32)][64 lanes][4].  tile_hint: 0 = choose, 1..6 = fixed streaming
 *      variant, 7..12 = fixed split-K variant (K11s: the waves of a workgroup split Cin, for maps of few pixels).
 * out_nhwc_cs != 0: y is written pixel-major, y[b][n][co] in rows of out_nhwc_cs >= Cout floats (pad written as zeros)
 *      -- the layout occd_lift_fwd gathers from, so the decoder's 1x1 heads feed the lift without a transpose pass
 *      (no gate / res in this mode).                                                                              */
typedef struct occd_pw_args {
    const float* x;
    const float* wpk;
    const float* shift;
    const float* gate;
    const float* res;
    float* y;
    int64_t N;
    int32_t batch, cin, cout;
    int32_t act, tile_hint;
    float slope;
    int32_t out_nhwc_cs;
} occd_pw_args;
int64_t occd_pw_packed_floats(int32_t cout, int32_t cin);
int occd_pw_pack_weights(const float* w, const float* scale, float* wpk, int32_t cout, int32_t cin, void* stream);
int occd_pw_conv_fwd(const occd_pw_args* a, void* stream);

/* SURVEY 8(f) row N4 (first step): out[row] = lut[argmax_c x[row][coff + c]] (first maximum wins; lut may
 * be NULL) as uint16 -- replaces the host softmax + numpy argmax of scripts/generate_output.py:94-95 and the
 * learning_map_inv lookup of scripts/generate_kitti_submission.py:74-85.                               */
int occd_argmax_channels(const float* x, int64_t rows, int32_t cs, int32_t coff, int32_t C,
                         const uint16_t* lut, uint16_t* out, void* stream);

/* ------------------------------------------------------------------------ *
 * K8: weight gradient of a 3-D convolution (training step):
 *   dw[co][ci][kx][ky][kz] = sum_{b,xo,yo,zo} gy[b,xo,yo,zo][co] * x[b, xo*sx - px + kx*dx, ...][ci]
 * -- what autograd computes for nn.Conv3d.weight in the reference's training step
 * (occdepth/models/OccDepth.py:535-537 -> every nn.Conv3d of models/DDR.py, modules.py, CRP3D.py).  The data
 * gradient needs no kernel of its own: it is occd_conv3d_fwd on gy with the flipped, channel-transposed weights
 * (stride 1) or the sub-pixel phases of the transposed convolution (stride > 1).
 * x, gy: channels-last fp32 like occd_conv3d_args; dw: PyTorch weight layout, overwritten.
 * `workspace` must hold occd_conv3d_wgrad_workspace_floats() floats (partial tiles; summed in a fixed order,
 * so the result is deterministic).  At most 28 taps.                                                           */
typedef struct occd_conv3d_wgrad_args {
    const float* x;      /* (B, X, Y, Z, x_cs) input of the convolution        */
    const float* gy;     /* (B, Xo, Yo, Zo, gy_cs) gradient w.r.t. its output  */
    float* dw;           /* (cout, cin, kx, ky, kz)                            */
    float* workspace;
    int64_t workspace_floats;
    int32_t batch;
    int32_t X, Y, Z, cin, x_cs, x_coff;
    int32_t Xo, Yo, Zo, cout, gy_cs, gy_coff;
    int32_t kx, ky, kz, sx, sy, sz, dx, dy, dz, px, py, pz;
} occd_conv3d_wgrad_args;
int64_t occd_conv3d_wgrad_workspace_floats(const occd_conv3d_wgrad_args* a);
int occd_conv3d_wgrad(const occd_conv3d_wgrad_args* a, void* stream);

/* ------------------------------------------------------------------------ *
 * K13: training-mode BatchNorm (+ activation + residual) as HBM-bound passes, forward and backward
 * (torch.nn.BatchNorm{2,3}d / SyncBatchNorm in training mode: occdepth/models/DDR.py:111-139, modules.py:40-46,158-175,
 * 278-296, unet2d.py:24-46; scripts/train.py:179 `sync_batchnorm=True`).
 *   layout 0: channels-last rows (`rows` rows of `*_cs` elements, channels at `*_coff`; dtype 0 fp32 / 1 bf16)
 *   layout 1: NCHW planes (`batch` x C planes of S fp32 elements; the *_cs / *_coff fields are ignored)
 * forward : occd_bn_stats (x -> partial) ; occd_bn_stats_combine (partial -> packed[2C+1] float64 =
 *           [n mean_c, M2_c + n mean_c^2, n], the form that sums over ranks) ; [all-reduce packed] ; occd_bn_finish
 *           (packed -> mean, invstd, a = gamma invstd, b = beta - mean a, running statistics with the unbiased variance,
 *           num_batches_tracked += 1) ; occd_bn_apply: out = act(x a + b [+ res if res_first]) [+ res if !res_first],
 *           act = 0 none / 1 relu / 2 swish / 3 leaky-relu(slope); `cw` (>= ceil4(C), rows layout) channels are written
 *           per row, those >= C as zeros.
 * backward: occd_bn_bwd_reduce (g = gy act'(pre), pre from `y`'s sign when y != NULL else x a + b; partial = sum g,
 *           sum g xhat) ; occd_bn_bwd_combine (partial -> packed[2C] fp32) ; [all-reduce a copy] ; occd_bn_bwd_finish
 *           (k1, k2, k3, gw = local sum g xhat, gb = local sum g) ; occd_bn_bwd_apply: out = g k1 + x k2 + k3 and,
 *           when out2 != NULL, out2 = g (the gradient of a residual that was added before the activation).
 * `partial` holds occd_bn_blocks() * 2 * ceil4(C) floats.
 * ------------------------------------------------------------------------ */
typedef struct occd_bn_args {
    const void* x;
    const void* gy;
    const void* y;
    const void* res;
    void* out;
    void* out2;
    const float* a;
    const float* b;
    const float* mean;
    const float* invstd;
    const float* k1;
    const float* k2;
    const float* k3;
    float* partial;
    int64_t rows;
    int64_t S;
    int32_t batch;
    int32_t C, cw;
    int32_t dtype, layout;
    int32_t x_cs, x_coff, gy_cs, gy_coff, y_cs, y_coff, res_cs, res_coff, out_cs, out_coff, out2_cs, out2_coff;
    int32_t act, res_first;
    float slope;
    int32_t nblk;
} occd_bn_args;
int occd_bn_blocks(const occd_bn_args* a);
int occd_bn_stats(const occd_bn_args* a, void* stream);
int occd_bn_stats_combine(const occd_bn_args* a, double* packed, void* stream);
int occd_bn_finish(const double* packed, int32_t C, float eps, float momentum, const float* gamma, const float* beta,
                   float* running_mean, float* running_var, int64_t* num_batches_tracked, float* mean, float* invstd,
                   float* a, float* b, void* stream);
int occd_bn_apply(const occd_bn_args* a, void* stream);
int occd_bn_bwd_reduce(const occd_bn_args* a, void* stream);
int occd_bn_bwd_combine(const float* partial, int32_t nblk, int32_t C, float* packed, void* stream);
int occd_bn_bwd_finish(const float* local, const float* total, int32_t C, const double* packed_fwd, const float* mean,
                       const float* invstd, const float* a, float* k1, float* k2, float* k3, float* gw, float* gb,
                       void* stream);
int occd_bn_bwd_apply(const occd_bn_args* a, void* stream);
/* launch-count reductions for LOCAL statistics: combine + finish in one launch (forward / backward), and the whole forward /
 * backward of a small NCHW layer (occd_bn_small_ok: layout 1, batch * S <= 8192, C >= 64) in one launch, one workgroup
 * per channel.                                                                                                       */
int occd_bn_stats_finish(const occd_bn_args* a, double* packed, float eps, float momentum, const float* gamma,
                         const float* beta, float* running_mean, float* running_var, int64_t* num_batches_tracked,
                         float* mean, float* invstd, float* av, float* bv, void* stream);
int occd_bn_bwd_combine_finish(const float* partial, int32_t nblk, int32_t C, const double* packed_fwd, const float* mean,
                               const float* invstd, const float* a, float* k1, float* k2, float* k3, float* gw, float* gb,
                               void* stream);
int occd_bn_small_ok(const occd_bn_args* a);
int occd_bn_fwd_small(const occd_bn_args* a, double* packed, float eps, float momentum, const float* gamma, const float* beta,
                      float* running_mean, float* running_var, int64_t* num_batches_tracked, float* mean, float* invstd,
                      float* av, float* bv, void* stream);
int occd_bn_bwd_small(const occd_bn_args* a, float* gw, float* gb, void* stream);
/* Round 5: the one-launch small layer with SYNCHRONISED statistics (SyncBatchNorm over the ranks of one node): workgroup c
 * exchanges channel c's packet with the peers' workgroups c inside the launch, through peer-mapped CHANNEL mailboxes
 * (occd_bn_xchg_mailbox_bytes(world, cmax) bytes each, created / opened with the occd_ipc_mailbox_* calls; one 64-byte
 * record of six self-validating {data : 32, sequence : 32} words per (slot, rank, channel); the fence-free protocol, the
 * determinism and the bounded wait of occd_ipc_allreduce).
 * mailboxes: HOST array of `world` device pointers, [rank] = own; C <= cmax.  Forward: `packed` receives the TOTALS over the
 * ranks [sum n mean, sum (M2 + n mean^2), sum n]; backward: packed_fwd = that vector, gw / gb stay this rank's sums.
 * Every rank must be able to schedule its workgroups while the peers' wait (one GPU per rank).  A peer's packet that does
 * not arrive within timeout_ms makes that channel's totals NaN (and sets *status), as in occd_ipc_allreduce.           */
int64_t occd_bn_xchg_mailbox_bytes(int32_t world, int32_t cmax);
int occd_bn_fwd_small_xchg(const occd_bn_args* a, double* packed, float eps, float momentum, const float* gamma,
                           const float* beta, float* running_mean, float* running_var, int64_t* num_batches_tracked,
                           float* mean, float* invstd, float* av, float* bv, void* const* mailboxes, int32_t rank,
                           int32_t world, int32_t cmax, int32_t timeout_ms, int32_t* status, void* stream);
int occd_bn_bwd_small_xchg(const occd_bn_args* a, const double* packed_fwd, float* gw, float* gb, void* const* mailboxes,
                           int32_t rank, int32_t world, int32_t cmax, int32_t timeout_ms, int32_t* status, void* stream);

/* ------------------------------------------------------------------------ *
 * K2b / K8b: the same convolution forward (data gradient: the forward on dL/dy with flipped weights) and weight
 * gradient on the bf16 matrix pipe -- v_mfma_f32_32x32x16_bf16, fp32 accumulate, fp32 master weights -- for the bf16
 * training step (BASELINE configs[3]; reference call sites as for K2 / K8, plus the 3x3 convolutions of the 2-D decoder,
 * occdepth/models/unet2d.py:24-46, as X = 1 volumes of channels-last images).
 * `dtype` selects the storage type of the activation tensors (in, out, res1, res2 / x, gy): 0 = fp32 (converted to
 * bf16 while staging: "bf16 MFMA, fp32 storage"), 1 = bf16 (pointers are cast; *_cs / *_coff count elements).
 * bias, dw and the workspace stay fp32.  `wpk` is the image of occd_pack_weights_bf16():
 *   wpk[tap][k16][nt][lane][j] (bf16),  value = W[cout = nt*32 + (lane & 31)][cin = k16*16 + (lane >> 5)*8 + j][tap].
 * act_in: NONE or RELU.  K8b stages voxel rows row-major in LDS and reads both MFMA operands with the gfx950
 * transposed LDS read (ds_read_b64_tr_b16); at most 28 taps.
 * ------------------------------------------------------------------------ */
int64_t occd_packed_weight_bf16_elems(int32_t cout, int32_t cin, int32_t taps);
int occd_pack_weights_bf16(const float* w, const float* scale, void* wpk,
                           int32_t cout, int32_t cin, int32_t kx, int32_t ky, int32_t kz,
                           int32_t layout, void* stream);
int occd_conv3d_bf16_fwd(const occd_conv3d_args* a, int32_t dtype, void* stream);
/* the bf16-pipe twin of occd_conv3d_fwd_phases (same contract; dtype as above; with dtype 2 also act_in = SIGMOID) */
int occd_conv3d_bf16_fwd_phases(const occd_conv3d_args* a, int32_t n, int32_t dtype, void* stream);
/* Pack a VIEW of a dense float32 weight tensor: element (co, ci, tap) of the packed operator is
 * w[co * s_co + ci * s_ci + tap_ofs[tap]] (tap_ofs: HOST array of ntaps <= 27 element offsets).  One launch instead of the
 * permute / index_select / contiguous chain autograd's data gradient needs for the transposed, flipped, tap-subset kernels
 * of its sub-pixel phases (torch.nn.grad.conv3d_input semantics; occdepth/models/OccDepth.py:535-537 backward).          */
int occd_pack_weights_gather(const float* w, const float* scale, float* wpk, int32_t cout, int32_t cin, int32_t ntaps,
                             int64_t s_co, int64_t s_ci, const int32_t* tap_ofs, void* stream);
int occd_pack_weights_bf16_gather(const float* w, const float* scale, void* wpk, int32_t cout, int32_t cin,
                                  int32_t ntaps, int64_t s_co, int64_t s_ci, const int32_t* tap_ofs, void* stream);
/* the same view as three images hi | mid | lo (3 x occd_packed_weight_bf16_elems elements): data-gradient operators of the 3-way
 * split (occd_conv3d_bf16_fwd dtype 2) */
int occd_pack_weights_bf16x3_gather(const float* w, const float* scale, void* wpk, int32_t cout, int32_t cin,
                                  int32_t ntaps, int64_t s_co, int64_t s_ci, const int32_t* tap_ofs, void* stream);
/* Opt-in experiment: dtype 2 of occd_conv3d_bf16_fwd = float32 tensors, both operands split into three
 * bf16 terms (x = hi + mid + lo), six bf16 MFMAs per K step (hi*hi, hi*mid, mid*hi, hi*lo, lo*hi, mid*mid), fp32
 * accumulate: float32-level accuracy at 6/16 of the fp32-MFMA time.  Its weight image (3 x
 * occd_packed_weight_bf16_elems() elements: hi | mid | lo) comes from occd_pack_weights_bf16x3.                     */
int occd_pack_weights_bf16x3(const float* w, const float* scale, void* wpk,
                             int32_t cout, int32_t cin, int32_t kx, int32_t ky, int32_t kz,
                             int32_t layout, void* stream);
int64_t occd_conv3d_wgrad_bf16_workspace_floats(const occd_conv3d_wgrad_args* a, int32_t dtype);
int occd_conv3d_wgrad_bf16(const occd_conv3d_wgrad_args* a, int32_t dtype, void* stream);

/* SURVEY 8(f) row N1 (first step): the scene-completion losses of one training step as ONE pass over the
 * logits.  Everything occdepth/loss/ssc_loss.py:17-99 (geo_scal_loss, sem_scal_loss, CE_ssc_loss) and the inline
 * frustum-proportion loss of occdepth/models/OccDepth.py:487-521 compute is a function of these sums over voxels
 * (p = softmax over C of logits (B, C, S); t = target (B, S) uint8, 255 = unlabelled; map_occ != 0 applies the
 * cascade-head relabelling of OccDepth.py:413-415 to t first):
 *   stats[0 .. C)        P[c]  = sum_{t != 255} p_c            Q32 fixed point
 *   stats[C .. 2C)       N[c]  = sum_{t == c} p_c              Q32
 *   stats[2C .. 3C)      T[c]  = #{t == c}                     exact
 *   stats[3C]            M     = #{t != 255}                   exact
 *   stats[3C + 1]        sum_{t != 255} w_t * (-log p_t)       Q24   (w = weights, NULL = 1)
 *   stats[3C + 2]        sum_{t != 255} w_t                    Q24
 *   stats[3C + 3 + f*C + c]  sum_{masks[b, f, s] != 0} p_c     Q32   (masks (B, F, S) bytes; all voxels, labelled
 *                                                                     or not, as the reference)
 * Integer accumulation: results are independent of the order of the atomics (deterministic).  C <= 32.        */
int64_t occd_ssc_stats_len(int32_t C, int32_t F);
int occd_ssc_loss_stats_fwd(const float* logits, const uint8_t* target, const uint8_t* masks,
                            const float* weights, int64_t* stats, int64_t batch, int32_t C, int64_t S,
                            int32_t F, int32_t map_occ, void* stream);
/* grad (B, C, S) = d loss / d logits given gstats = d loss / d (the REAL-valued sums above, same layout; the
 * entries of the counts are ignored):  grad_c = p_c (g_c - sum_k p_k g_k) + [t != 255] gstats[3C+1] w_t (p_c - [c == t])
 * with g_c = [t != 255] (gP[c] + [t == c] gN[c]) + sum_{f : mask} gF[f][c].                                     */
int occd_ssc_loss_stats_bwd(const float* logits, const uint8_t* target, const uint8_t* masks,
                            const float* weights, const float* gstats, float* grad, int64_t batch, int32_t C,
                            int64_t S, int32_t F, int32_t map_occ, void* stream);
/* Round 5: the same three passes on logits (and gradients) that are NOT (B, C, S) planes -- element (b, c, s) at
 * b*s_b + c*s_c + s*s_v floats.  (s_c, s_v) = (S, 1) is the plane layout above; (1, cs) is the 3-D stack's own
 * channels-last voxel rows (cs % 4 == 0, 16-byte aligned base), read with 16-byte loads: the two 168 MB layout copies
 * of `ssc_logit` per training step (profiles/r04_train_step_bf16_aten_ops.txt) are gone.  bwd: grad has its own strides
 * (g_b, g_c, g_v); for channels-last grad rows the channels [C, g_pad) are written as zeros (g_pad % 4 == 0,
 * C <= g_pad <= g_v; 0 for the plane layout).                                                                       */
int occd_ssc_loss_stats_fwd_strided(const float* logits, const uint8_t* target, const uint8_t* masks,
                                    const float* weights, int64_t* stats, int64_t batch, int32_t C, int64_t S,
                                    int32_t F, int32_t map_occ, int64_t s_b, int64_t s_c, int64_t s_v, void* stream);
int occd_ssc_loss_stats_bwd_strided(const float* logits, const uint8_t* target, const uint8_t* masks,
                                    const float* weights, const float* gstats, float* grad, int64_t batch, int32_t C,
                                    int64_t S, int32_t F, int32_t map_occ, int64_t s_b, int64_t s_c, int64_t s_v,
                                    int64_t g_b, int64_t g_c, int64_t g_v, int32_t g_pad, void* stream);
int occd_ssc_confusion_strided(const float* logits, const uint8_t* labels, const uint8_t* target, int64_t* hist,
                               int64_t batch, int32_t C, int64_t S, int64_t s_b, int64_t s_c, int64_t s_v, void* stream);

/* the relation (context prior) loss, occdepth/loss/CRP_loss.py:4-24 -- BCEWithLogits with
 * pos_weight[r] = #negatives / #positives of relation r over the batch, mean over all B*R*M*N elements -- as ONE pass:
 *   loss = 1 / (R T) * sum_r ( pw_r * Spos_r + Sneg_r ),  T = B M N,
 *   stats[3 r + 0] = #{y = 1},  stats[3 r + 1] = sum_{y = 1} softplus(-x)  (Q24),  stats[3 r + 2] = sum_{y = 0} softplus(x)  (Q24)
 * (integer accumulation: deterministic).  logits: element (b, r, m, n) at b*l_b + r*l_r + m*l_m + n*l_n floats with
 * l_m == 1 or l_n == 1 (the HIP forward produces (R, B, N, M) rows, the autograd graph (B, R, M, N)); labels: the batch's
 * CP_mega_matrices, (B, R, N, M) contiguous, label_dtype 0 = uint8 (the dataloader's), 1 = float32.
 * grad: d loss / d logits = y ? -coef[2r] * sigmoid(-x) : coef[2r + 1] * sigmoid(x), written with the logits' strides;
 * coef (R, 2) device floats = g * (pw_r, 1) / (R T).                                                               */
int occd_relation_bce_stats(const float* logits, const void* labels, int32_t label_dtype, int64_t* stats,
                            int64_t batch, int32_t R, int64_t M, int64_t N, int64_t l_b, int64_t l_r, int64_t l_m,
                            int64_t l_n, void* stream);
int occd_relation_bce_grad(const float* logits, const void* labels, int32_t label_dtype, const float* coef,
                           float* grad, int64_t batch, int32_t R, int64_t M, int64_t N, int64_t l_b, int64_t l_r,
                           int64_t l_m, int64_t l_n, void* stream);

/* the depth-distribution loss of FLoSP-Depth, occdepth/loss/depth_loss.py:14-87, one thread per
 * prediction cell: nearest-resample the (Bn, srcH, srcW) metric depth map to cell x (h, w), smallest non-zero depth of
 * every cell x cell block, LID bin k = trunc((d - d_off) / d_step) (d_off = float32(d_bound[0] - d_bound[2])), target =
 * one-hot(k)[1:], BCE(prob, target) with both logs clamped at -100, summed over the D bins of every cell that has a
 * target.  prob: (Bn, D, h, w) probabilities, image stride p_b floats (a camera slice of (B, n_cams, D, h, w)).
 *   stats[0] = sum over measured cells (Q24 fixed point), stats[1] = #measured cells;  loss = stats[0] / max(1, stats[1]).
 * grad (Bn, D, h, w) dense = gscale[0] * (p - t) / max((1 - p) p, 1e-12) on measured cells, 0 elsewhere (ATen's
 * binary_cross_entropy_backward); gscale = device scalar g / max(1, #measured).                                   */
int occd_depth_bce_stats(const float* prob, const float* gt, int64_t* stats, int64_t Bn, int32_t D, int32_t h,
                         int32_t w, int32_t srcH, int32_t srcW, int32_t cell, int64_t p_b, float d_off, float d_step,
                         void* stream);
int occd_depth_bce_grad(const float* prob, const float* gt, const float* gscale, float* grad, int64_t Bn, int32_t D,
                        int32_t h, int32_t w, int32_t srcH, int32_t srcW, int32_t cell, int64_t p_b, float d_off,
                        float d_step, void* stream);

/* hist[t * C + pred] += 1 over voxels with t != 255 -- every counter
 * occdepth/loss/sscMetrics.py:70-204 (SSCMetrics.add_batch) keeps derives from this matrix.  pred is either
 * `labels` (B, S) uint8 or the arg-max over C of `logits` (B, C, S) (first maximum wins, as np.argmax in
 * OccDepth.py:523-526); exactly one of the two is non-NULL.  hist is NOT cleared (it accumulates).         */
int occd_ssc_confusion(const float* logits, const uint8_t* labels, const uint8_t* target, int64_t* hist,
                       int64_t batch, int32_t C, int64_t S, void* stream);

/* ------------------------------------------------------------------------ *
 * Small-message all-reduce over peer-mapped device memory (round 5; csrc/ipc_allreduce.hip): the latency-optimal form
 * of SyncBatchNorm's per-layer exchange of (2C + 1) statistics -- `Trainer(sync_batchnorm=True)`,
 * occdepth/scripts/train.py:176-206 -- one kernel per call, no library, deterministic (rows summed in rank order).
 * Set-up (once per process group, host side): every rank creates a mailbox of occd_ipc_mailbox_bytes(world, max_bytes)
 * bytes (fine-grained device memory, zeroed) and its 64-byte IPC handle, the handles are exchanged by any means
 * (torch.distributed object collectives), every rank opens its peers' handles.  Call: all ranks issue the same sequence of
 * occd_ipc_allreduce calls (dtype 0 float32 / 1 float64, count * elem <= max_bytes) on a stream; `mailboxes` is a HOST
 * array of `world` device pointers, [rank] = the rank's own mailbox; the per-call sequence number lives in the mailbox and
 * is advanced by the kernel, so a captured launch replays correctly.  Wire format: every 32-bit half of the payload in its
 * own 8-byte word {data, sequence}, one system-scope atomic store / load each -- no fence anywhere.  The wait is bounded by timeout_ms (<= 0: unbounded):
 * on expiry the kernel sets *status = 1 (device int, optional) and writes NaN into the elements of `out` whose wait gave up
 * (never a partial sum: a lost peer poisons the step's loss instead of silently skewing it).                          */
int64_t occd_ipc_mailbox_bytes(int32_t world, int64_t max_bytes);
int occd_ipc_mailbox_create(int64_t bytes, void** mailbox, void* handle64);
int occd_ipc_mailbox_open(const void* handle64, void** peer);
int occd_ipc_mailbox_close(void* peer);
int occd_ipc_mailbox_free(void* mailbox);
int occd_ipc_allreduce(const void* in, void* out, int64_t count, int32_t dtype, void* const* mailboxes, int32_t rank,
                       int32_t world, int64_t max_bytes, int32_t timeout_ms, int32_t* status, void* stream);

/* ------------------------------------------------------------------------ *
 * In-library kernel timing (HIP events on the launch stream) used by bench.py
 * for `roofline.achieved`.  occd_prof_enable(1) starts recording one event pair
 * per launch; occd_prof_report() synchronises the recorded events and returns,
 * per kernel tag, launches / total ms / total algorithmic flops / bytes.
 * ------------------------------------------------------------------------ */
typedef struct occd_prof_row {
    char tag[64];
    int64_t launches;
    double ms;
    double flops;
    double bytes;
} occd_prof_row;

int occd_prof_enable(int32_t on);
int occd_prof_set_tag(const char* tag);   /* tag attached to subsequent launches */
int occd_prof_report(occd_prof_row* rows, int32_t max_rows); /* returns #rows; resets */

#ifdef __cplusplus
}
#endif
#endif /* OCCDEPTH_AMD_H */
