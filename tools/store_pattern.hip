// Dev tool: what does the SHAPE of a GEMM epilogue's stores cost on MI355X?  A workgroup of 8 waves owns 64 columns x M rows of
// a row-major (batch, M, pitch) float32 tensor (the K16p tile walk: wave w takes the 32-row tiles w, w + 8, ...) and only stores
// -- no loads, no arithmetic -- in one of these forms:
//   0  the MFMA D layout as it falls out of the accumulators: one dword per lane, an instruction = two 128-byte row segments
//   1  one dword per lane, an instruction = ONE 256-byte row segment (what a half-wave swap of two tiles' registers gives)
//   2  two dwords per lane (global_store_dwordx2): an instruction = two 256-byte row segments
//   3  four dwords per lane (dwordx4): an instruction = four 256-byte row segments
//   4  form 0 with the wave's stores ordered column tile inner (both 128-byte halves of a row back to back)
// Build + run:  hipcc --offload-arch=gfx950 -O3 tools/store_pattern.hip -o /tmp/store_pattern && /tmp/store_pattern
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2), aligned(4)));
typedef float f32x4 __attribute__((ext_vector_type(4), aligned(4)));

template <int FORM>
__global__ void __launch_bounds__(512) store_kernel(float* out, int M, int N, long pitch, int ntiles, float v) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 31, h = lane >> 5;
    const int nt_i = blockIdx.x % ntiles, bz = blockIdx.x / ntiles;
    const int n0 = nt_i * 64;
    float* const C = out + (size_t)bz * M * pitch;
    const int tiles = (M + 31) / 32;
    for (int tb = wave; tb < tiles; tb += 8) {
        const int m0 = tb * 32;
        if (FORM == 0 || FORM == 4) {
#pragma unroll
            for (int a = 0; a < (FORM == 0 ? 2 : 16); ++a)
#pragma unroll
                for (int b = 0; b < (FORM == 0 ? 16 : 2); ++b) {
                    const int nt = FORM == 0 ? a : b, r = FORM == 0 ? b : a;
                    const int m = m0 + 4 * h + (r & 3) + 8 * (r >> 2), n = n0 + nt * 32 + li;
                    if (m < M && n < N) C[(size_t)m * pitch + n] = v + r;
                }
        } else if (FORM == 1) {
#pragma unroll
            for (int r = 0; r < 32; ++r) {
                const int m = m0 + r, n = n0 + lane;
                if (m < M && n < N) C[(size_t)m * pitch + n] = v + r;
            }
        } else if (FORM == 2) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + 2 * r + h, n = n0 + li * 2;
                if (m < M && n + 1 < N) *(f32x2*)(C + (size_t)m * pitch + n) = f32x2{v + r, v};
            }
        } else {
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const int m = m0 + 4 * r + (lane >> 4), n = n0 + (lane & 15) * 4;
                if (m < M && n + 3 < N) *(f32x4*)(C + (size_t)m * pitch + n) = f32x4{v + r, v, v, v};
            }
        }
    }
}

template <int FORM>
float run(float* d, int batch, int M, int N, long pitch) {
    const int ntiles = (N + 63) / 64;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(store_kernel<FORM>, dim3(ntiles * batch), dim3(512), 0, 0, d, M, N, pitch, ntiles, 1.f);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (rep > 0 && ms < best) best = ms;
    }
    return best;
}

int main() {
    const int batch = 2, N = 112850;
    float* d;
    hipMalloc(&d, (size_t)batch * 1440 * 112896 * 4 + 4096);
    for (int M : {192, 720, 1440}) {
        for (long pitch : {(long)N, (long)112864, (long)112896}) {       // as is (= 72 mod 128 bytes); 128-byte rows; 512-byte rows
            for (int off : {0, 2}) {                                       // base offset in floats (8-byte misalignment of the tensor)
                float t[5] = {run<0>(d + off, batch, M, N, pitch), run<1>(d + off, batch, M, N, pitch), run<2>(d + off, batch, M, N, pitch),
                              run<3>(d + off, batch, M, N, pitch), run<4>(d + off, batch, M, N, pitch)};
                const double mb = (double)batch * M * N * 4 / 1e6;
                printf("M %4d pitch %6ld off %d  (%.0f MB):", M, pitch, off, mb);
                for (int f = 0; f < 5; ++f) printf("  form %d %7.1f us %5.0f GB/s", f, t[f] * 1e3, mb / t[f] / 1e3);
                printf("\n");
            }
        }
    }
    hipMemset(d, 0, 1 << 20);
    return 0;
}
