// Shared host-side helpers for libocc_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "../../include/occdepth_amd.h"

namespace occd {

// Launch-time profiling (HIP events on the launch stream); see prof.cpp.
struct ProfScope {
    ProfScope(const char* kind, hipStream_t s, double flops, double bytes);
    ~ProfScope();
    int slot;
    hipStream_t stream;
};

// conv3d_c32p.hip: persistent weights-stationary kernel for the full-resolution head convolutions.
// 1 = launched, 0 = geometry does not qualify (use the generic kernel), <0 = error.
int try_conv3d_c32_persist(const occd_conv3d_args* a, hipStream_t stream);
// K2s3: the same launches with the 3-way bf16 split (weights = the image of occd_pack_weights_bf16x3); same return convention.
int try_conv3d_c32_slide_x3(const occd_conv3d_args* a, hipStream_t stream);

// occd_pack_weights*_gather: where element (co, ci, tap) of a packed operator lives in a dense source tensor
constexpr int kMaxTaps = 27;
struct TapMap {
    int64_t s_co, s_ci;
    int32_t ofs[kMaxTaps];
};

// hipFuncAttributeMaxDynamicSharedMemorySize (> 64 KB of LDS) belongs to the (kernel, DEVICE) pair: set once per pair,
// whatever device the calling thread has current (prof.cpp).  Returns OCCD_OK / OCCD_ELAUNCH.
int ensure_big_lds(const void* kernel);

// A/B switches are read BY VALUE, all the same way: unset -> `dflt`; "0" / "" / "off" / "false" -> false; anything
// else -> true.  (`OCCD_PHASE_FAST=0` used to ENABLE the switch because only presence was tested.)
inline bool env_flag(const char* name, bool dflt) {
    const char* e = getenv(name);
    if (e == nullptr) return dflt;
    return !(e[0] == '\0' || e[0] == '0' || e[0] == 'f' || e[0] == 'F' || ((e[0] == 'o' || e[0] == 'O') && (e[1] == 'f' || e[1] == 'F')));
}

inline int check_launch() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? OCCD_OK : OCCD_ELAUNCH;
}

// q = n / d for n*d < 2^32 via one mul-hi; d == 1 is encoded as magic 0.
struct FastDiv {
    uint32_t magic;
    uint32_t d;
};
inline FastDiv make_fastdiv(uint32_t d) {
    FastDiv f;
    f.d = d;
    f.magic = d <= 1 ? 0u : (uint32_t)((((uint64_t)1 << 32) + d - 1) / d);
    return f;
}

// swish(v) = v * sigmoid(v) on the hardware exp2 / rcp instructions (1 ulp each): the libm expf + IEEE division pair is
// ~25 VALU instructions per element, which in the epilogue of a short-K pointwise GEMM (64 accumulator values per lane)
// costs as much as its MFMA loop.  Result within ~3 ulp of the exact form.
__device__ __forceinline__ float swish_fast(float v) {
    return v * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(v * -1.4426950408889634f));
}

}  // namespace occd

__device__ __forceinline__ uint32_t occd_fastdiv(uint32_t n, occd::FastDiv f) {
    // branch-free: magic == 0 encodes d == 1
    return __umulhi(n, f.magic) + (f.magic == 0u ? n : 0u);
}
