// 2-D (NCHW) memory-bound helpers around the MIOpen convolutions of the 2-D UNet (SURVEY.md 8f row N3,
// first step): every kernel replaces a chain of separate ATen elementwise / copy launches by ONE pass.
//
//   affine_act_kernel    y = act(x * scale[c] + shift[c]) (+ residual)   BatchNorm(eval)+Swish/LeakyReLU/ReLU
//                        (+ the MBConv skip add): 1 read + 1 write instead of 3-4 passes
//                        (efficientnet blocks bn1+act1 / bn2+act2 / bn3 (+skip); decoder conv+BN+LeakyReLU)
//   dwconv2d_kernel      depthwise k x k convolution with TensorFlow "SAME" padding computed in-kernel
//                        (no F.pad copy) and the following BatchNorm+Swish fused in the epilogue; replaces
//                        MIOpen's naive grouped-conv fallback
//   upsample_cat_kernel  out[:, :C] = bilinear(x, (H, W), align_corners=True), out[:, C:] = skip
//                        (F.interpolate + torch.cat of unet2d.py:38-46 in one pass)
//
// Reference semantics: occdepth/models/unet2d.py:24-46 and the geffnet EfficientNet blocks (third party).
#include "common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

__device__ __forceinline__ float act_apply(float v, int act, float slope) {
    if (act == 1) return fmaxf(v, 0.f);                    // relu
    if (act == 2) return v / (1.f + expf(-v));             // swish / silu
    if (act == 3) return v > 0.f ? v : v * slope;          // leaky relu
    return v;
}

// one (b, c) plane per blockIdx.y, grid-stride over the plane; float4 when the plane size allows
__global__ void __launch_bounds__(256) affine_act_kernel(const float* __restrict__ x, const float* __restrict__ res,
                                                         float* __restrict__ y, const float* __restrict__ scale,
                                                         const float* __restrict__ shift, int C, long S, int act,
                                                         float slope, int res_first) {
    const int plane = blockIdx.y;
    const int c = plane % C;
    const float s = scale ? scale[c] : 1.f, t = shift ? shift[c] : 0.f;
    const size_t off = (size_t)plane * S;
    const long n4 = ((S & 3) == 0 && ((off & 3) == 0)) ? S >> 2 : 0;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        f32x4 v = *(const f32x4*)(x + off + i * 4);
        v = v * s + t;
        f32x4 r = {0.f, 0.f, 0.f, 0.f};
        if (res) r = *(const f32x4*)(res + off + i * 4);
        if (res_first) v += r;
        v.x = act_apply(v.x, act, slope); v.y = act_apply(v.y, act, slope);
        v.z = act_apply(v.z, act, slope); v.w = act_apply(v.w, act, slope);
        if (!res_first) v += r;
        *(f32x4*)(y + off + i * 4) = v;
    }
    for (long i = n4 * 4 + (long)blockIdx.x * 256 + threadIdx.x; i < S; i += (long)gridDim.x * 256) {
        float v = x[off + i] * s + t;
        const float r = res ? res[off + i] : 0.f;
        if (res_first) v += r;
        v = act_apply(v, act, slope);
        if (!res_first) v += r;
        y[off + i] = v;
    }
}

template <int K>
__global__ void __launch_bounds__(256) dwconv2d_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                       const float* __restrict__ scale, const float* __restrict__ shift,
                                                       float* __restrict__ y, int C, int H, int W, int Ho, int Wo,
                                                       int stride, int pad_t, int pad_l, int act) {
    const int plane = blockIdx.z;
    const int c = plane % C;
    const int ox = blockIdx.x * 64 + (threadIdx.x & 63);
    const int oy = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (ox >= Wo || oy >= Ho) return;
    const float* xp = x + (size_t)plane * H * W;
    const float* wp = w + (size_t)c * K * K;
    float acc = 0.f;
#pragma unroll
    for (int ky = 0; ky < K; ++ky) {
        const int iy = oy * stride - pad_t + ky;
        if ((unsigned)iy >= (unsigned)H) continue;
#pragma unroll
        for (int kx = 0; kx < K; ++kx) {
            const int ix = ox * stride - pad_l + kx;
            if ((unsigned)ix < (unsigned)W) acc += xp[(size_t)iy * W + ix] * wp[ky * K + kx];
        }
    }
    const float s = scale ? scale[c] : 1.f, t = shift ? shift[c] : 0.f;
    y[((size_t)plane * Ho + oy) * Wo + ox] = act_apply(acc * s + t, act, 0.f);
}

__global__ void __launch_bounds__(256) upsample_cat_kernel(const float* __restrict__ x, const float* __restrict__ skip,
                                                           float* __restrict__ out, int C, int Cs, int h, int w, int H,
                                                           int W, float rh, float rw) {
    const int plane = blockIdx.z;                 // b * (C + Cs) + channel
    const int ct = C + Cs;
    const int b = plane / ct, c = plane - b * ct;
    const int ox = blockIdx.x * 64 + (threadIdx.x & 63);
    const int oy = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (ox >= W || oy >= H) return;
    float v;
    if (c >= C) {
        v = skip[(((size_t)b * Cs + (c - C)) * H + oy) * W + ox];
    } else {
        // torch upsample_bilinear2d, align_corners=True: src = dst * (in - 1) / (out - 1)
        const float sy = rh * oy, sx = rw * ox;
        const int y0 = (int)sy, x0 = (int)sx;
        const int y1 = y0 + (y0 < h - 1), x1 = x0 + (x0 < w - 1);
        const float ly = sy - y0, lx = sx - x0;
        const float hy = 1.f - ly, hx = 1.f - lx;
        const float* p = x + ((size_t)b * C + c) * h * w;
        v = hy * (hx * p[(size_t)y0 * w + x0] + lx * p[(size_t)y0 * w + x1]) +
            ly * (hx * p[(size_t)y1 * w + x0] + lx * p[(size_t)y1 * w + x1]);
    }
    out[((size_t)plane * H + oy) * W + ox] = v;
}

}  // namespace

extern "C" int occd_affine_act_nchw(const float* x, const float* res, float* y, const float* scale, const float* shift,
                                    int32_t batch, int32_t C, int64_t S, int32_t act, float slope, int32_t res_first,
                                    void* stream) {
    if (!x || !y || batch <= 0 || C <= 0 || S <= 0 || act < 0 || act > 3) return OCCD_EINVAL;
    const long planes = (long)batch * C;
    if (planes > 65535) return OCCD_EINVAL;
    long bx = (S / 4 + 255) / 256;
    if (bx < 1) bx = 1;
    if (bx > 64) bx = 64;
    occd::ProfScope prof("affine_act_nchw", (hipStream_t)stream, 0.0, 4.0 * planes * S * (2 + (res != nullptr)));
    hipLaunchKernelGGL(affine_act_kernel, dim3((unsigned)bx, (unsigned)planes), dim3(256), 0, (hipStream_t)stream, x,
                       res, y, scale, shift, C, (long)S, act, slope, res_first);
    return occd::check_launch();
}

extern "C" int occd_dwconv2d_nchw(const float* x, const float* w, const float* scale, const float* shift, float* y,
                                  int32_t batch, int32_t C, int32_t H, int32_t W, int32_t k, int32_t stride,
                                  int32_t pad_top, int32_t pad_left, int32_t Ho, int32_t Wo, int32_t act,
                                  void* stream) {
    if (!x || !w || !y || batch <= 0 || C <= 0 || H <= 0 || W <= 0 || Ho <= 0 || Wo <= 0 || stride <= 0)
        return OCCD_EINVAL;
    if ((k != 3 && k != 5) || act < 0 || act > 2 || (long)batch * C > 65535) return OCCD_EINVAL;
    const dim3 grid((unsigned)((Wo + 63) / 64), (unsigned)((Ho + 3) / 4), (unsigned)(batch * C));
    occd::ProfScope prof("dwconv2d_nchw", (hipStream_t)stream, 2.0 * batch * C * (double)Ho * Wo * k * k,
                         4.0 * batch * C * ((double)H * W + (double)Ho * Wo));
    if (k == 3)
        hipLaunchKernelGGL(dwconv2d_kernel<3>, grid, dim3(256), 0, (hipStream_t)stream, x, w, scale, shift, y, C, H, W,
                           Ho, Wo, stride, pad_top, pad_left, act);
    else
        hipLaunchKernelGGL(dwconv2d_kernel<5>, grid, dim3(256), 0, (hipStream_t)stream, x, w, scale, shift, y, C, H, W,
                           Ho, Wo, stride, pad_top, pad_left, act);
    return occd::check_launch();
}

extern "C" int occd_upsample_bilinear_cat_nchw(const float* x, const float* skip, float* out, int32_t batch, int32_t C,
                                               int32_t Cskip, int32_t h, int32_t w, int32_t H, int32_t W,
                                               void* stream) {
    if (!x || !out || batch <= 0 || C <= 0 || Cskip < 0 || (Cskip > 0 && !skip) || h <= 0 || w <= 0 || H <= 0 ||
        W <= 0 || (long)batch * (C + Cskip) > 65535)
        return OCCD_EINVAL;
    // at::native::area_pixel_compute_scale<float>(in, out, align_corners=true)
    const float rh = H > 1 ? (float)(h - 1) / (float)(H - 1) : 0.f;
    const float rw = W > 1 ? (float)(w - 1) / (float)(W - 1) : 0.f;
    const dim3 grid((unsigned)((W + 63) / 64), (unsigned)((H + 3) / 4), (unsigned)(batch * (C + Cskip)));
    occd::ProfScope prof("upsample_cat_nchw", (hipStream_t)stream, 0.0,
                         4.0 * batch * ((double)C * h * w + 2.0 * Cskip * H * W + (double)C * H * W));
    hipLaunchKernelGGL(upsample_cat_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, skip, out, C, Cskip, h, w, H, W,
                       rh, rw);
    return occd::check_launch();
}
