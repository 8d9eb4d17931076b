// Dev tool: what fp32 MFMA rate does this MI355X sustain with NO memory traffic at all?  (The 157.3 TF/s peak
// assumes 2.4 GHz; under a sustained matrix load the part may clock lower.)  Build + run:
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o tools/mfma_peak && tools/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ void __launch_bounds__(512) mfma_loop(float* out, int iters, float a0, float b0) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float a = a0 + threadIdx.x * 1e-9f, b = b0;
    if (a0 < 0.f) {   // "random data" arm: per-lane pseudo-random operands in [-1, 1) (DVFS: toggling costs clock)
        unsigned h = (threadIdx.x + blockIdx.x * 512u) * 2654435761u;
        h ^= h >> 13; h *= 0x5bd1e995u; h ^= h >> 15;
        a = (int)(h & 0xffff) * (1.f / 32768.f) - 1.f;
        b = (int)(h >> 16) * (1.f / 32768.f) - 1.f;
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i)
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    if (s == 12345.678f) out[0] = s;
}

template <int NACC>
void run(int wgs, int threads, int iters, float* d, float a0 = 1.f) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(mfma_loop<NACC>, dim3(wgs), dim3(threads), 0, 0, d, iters, a0, 1.f);
    hipDeviceSynchronize();
    float best = 1e30f, last = 0.f;
    for (int rep = 0; rep < 6; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(mfma_loop<NACC>, dim3(wgs), dim3(threads), 0, 0, d, iters, a0, 1.f);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&last, e0, e1);
        if (last < best) best = last;
    }
    const double fl = (double)wgs * (threads / 64) * iters * 8.0 * NACC * 4096.0;
    printf("%s acc=%d wgs=%d thr=%d iters=%d: best %.3f ms %.1f TF/s, last %.3f ms %.1f TF/s\n", a0 < 0.f ? "random" : "const ", NACC, wgs, threads, iters,
           best, fl / best / 1e9, last, fl / last / 1e9);
}

int main() {
    float* d;
    hipMalloc(&d, 64);
    for (int iters : {2000, 20000, 200000}) {     // ~1 ms, ~10 ms, ~100 ms kernels: does the clock sag with duration?
        run<1>(256, 512, iters, d);
        run<2>(256, 512, iters, d);
        run<4>(256, 256, iters, d);
        run<2>(512, 512, iters, d);
        run<1>(256, 512, iters, d, -1.f);
        run<2>(256, 512, iters, d, -1.f);
    }
    return 0;
}
