// Hardware probe (gfx950): pins the two layout facts the bf16 kernels rely on.
//  (1) v_mfma_f32_32x32x16_bf16 operand layout: lane l holds A[i = l & 31][k = 8 (l >> 5) + j], j = 0..7 (and B alike),
//      D[i][n]: lane & 31 = n, register r -> i = (r & 3) + 8 (r >> 2) + 4 (l >> 5).
//  (2) ds_read_b64_tr_b16: within each 16-lane group, lane i supplies the address of 4 consecutive bf16 of row (i >> 2),
//      column chunk (i & 3) of a 4 x 16 block, and receives column i: element j = block[j][i].
// Build: hipcc --offload-arch=gfx950 -O2 tools/probe_bf16.hip -o tools/scratch/probe_bf16 ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cmath>
#include <cstring>
#include <vector>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ uint16_t f2bf(float f) {
    uint32_t u = __builtin_bit_cast(uint32_t, f);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}

__global__ void probe_mfma(const float* A, const float* B, float* D) {   // A [32][16], B [16][32] row-major
    const int l = threadIdx.x, i = l & 31, h = l >> 5;
    union { bf16x8 v; uint16_t u[8]; } a, b;
    for (int j = 0; j < 8; ++j) {
        a.u[j] = f2bf(A[i * 16 + 8 * h + j]);
        b.u[j] = f2bf(B[(8 * h + j) * 32 + i]);
    }
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.v, b.v, acc, 0, 0, 0);
    for (int r = 0; r < 16; ++r) D[l * 16 + r] = acc[r];
}

__global__ void probe_tr(uint16_t* out) {       // LDS image: 16 rows x 32 columns of uint16 = row * 32 + col, row stride 64 B
    __shared__ __attribute__((aligned(16))) uint16_t lds[16 * 32];
    const int l = threadIdx.x;
    for (int e = l; e < 16 * 32; e += 64) lds[e] = (uint16_t)e;
    __syncthreads();
    const int g = l >> 4, i = l & 15;
    // group g reads the 4 x 16 block at rows 4 g' .. (g' = g here), columns 16 (g & 1) .. : lane supplies row (i >> 2), chunk (i & 3)
    const uint16_t* p = lds + (4 * g + (i >> 2)) * 32 + 16 * (g & 1) + 4 * (i & 3);
    typedef __attribute__((address_space(3))) bf16x4 lds_v4;
    union { bf16x4 v; uint16_t u[4]; } r;
    r.v = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_v4*)p);
    for (int j = 0; j < 4; ++j) out[l * 4 + j] = r.u[j];
}

static float bfround(float f) {
    uint32_t u; std::memcpy(&u, &f, 4); u += 0x7fffu + ((u >> 16) & 1u); u &= 0xffff0000u; float r; std::memcpy(&r, &u, 4); return r;
}

int main() {
    std::vector<float> A(32 * 16), B(16 * 32), D(64 * 16), ref(32 * 32);
    for (int i = 0; i < 32; ++i) for (int k = 0; k < 16; ++k) A[i * 16 + k] = bfround(0.25f * (float)((i * 7 + k * 3) % 23 - 11));
    for (int k = 0; k < 16; ++k) for (int n = 0; n < 32; ++n) B[k * 32 + n] = bfround(0.5f * (float)((k * 5 + n * 11) % 19 - 9));
    for (int i = 0; i < 32; ++i) for (int n = 0; n < 32; ++n) {
        double s = 0; for (int k = 0; k < 16; ++k) s += (double)A[i * 16 + k] * B[k * 32 + n]; ref[i * 32 + n] = (float)s;
    }
    float *dA, *dB, *dD; uint16_t* dT;
    hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dD, D.size() * 4); hipMalloc(&dT, 64 * 4 * 2);
    hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe_mfma, dim3(1), dim3(64), 0, 0, dA, dB, dD);
    hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l) for (int r = 0; r < 16; ++r) {
        const int n = l & 31, i = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
        if (std::fabs(D[l * 16 + r] - ref[i * 32 + n]) > 1e-3f) ++bad;
    }
    printf("PROBE mfma_f32_32x32x16_bf16 layout (A[i=l&31][k=8(l>>5)+j], D row=(r&3)+8(r>>2)+4(l>>5), col=l&31): %s (%d mismatches)\n",
           bad == 0 ? "CONFIRMED" : "WRONG", bad);
    if (bad) {   // which (i, n) does lane 0 reg 1 / lane 33 reg 0 hold?
        for (int probe = 0; probe < 4; ++probe) {
            const int l = probe == 0 ? 0 : probe == 1 ? 1 : probe == 2 ? 32 : 33, r = probe == 0 ? 1 : 0;
            for (int i = 0; i < 32; ++i) for (int n = 0; n < 32; ++n)
                if (std::fabs(D[l * 16 + r] - ref[i * 32 + n]) < 1e-3f) printf("  lane %d reg %d could be D[%d][%d]\n", l, r, i, n);
        }
    }
    std::vector<uint16_t> T(64 * 4);
    hipLaunchKernelGGL(probe_tr, dim3(1), dim3(64), 0, 0, dT);
    hipMemcpy(T.data(), dT, T.size() * 2, hipMemcpyDeviceToHost);
    bad = 0;
    for (int l = 0; l < 64; ++l) for (int j = 0; j < 4; ++j) {
        const int g = l >> 4, i = l & 15;
        const int want = (4 * g + j) * 32 + 16 * (g & 1) + i;      // block[j][i]
        if (T[l * 4 + j] != want) ++bad;
    }
    printf("PROBE ds_read_b64_tr_b16 (lane i of a 16-group supplies row i>>2 chunk i&3, receives column i): %s (%d mismatches)\n",
           bad == 0 ? "CONFIRMED" : "WRONG", bad);
    if (bad) for (int l = 0; l < 64; l += 1) printf("  lane %2d: %4d %4d %4d %4d\n", l, T[l * 4], T[l * 4 + 1], T[l * 4 + 2], T[l * 4 + 3]);
    hipError_t e = hipDeviceSynchronize();
    printf("PROBE done: %s\n", hipGetErrorString(e));
    return 0;
}
