// Dev tool: where does a K16p workgroup spend its time?  Builds csrc/gemm_x3.hip with -DOCCD_PANEL_TIMELINE (shader-clock probes at
// the kernel's phase boundaries, one workgroup, every wave), runs one shape through the C ABI and prints the timeline.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DOCCD_PANEL_TIMELINE -x hip tools/panel_timeline.cpp occdepth_amd/csrc/prof.cpp -o /tmp/panel_timeline
//   /tmp/panel_timeline M N K batch [probe workgroup] [epilogue 0/1]
#include "../occdepth_amd/csrc/gemm_x3.hip"
#include <cstdio>
#include <vector>

int main(int argc, char** argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 2304, N = argc > 2 ? atoi(argv[2]) : 468, K = argc > 3 ? atoi(argv[3]) : 384;
    const int batch = argc > 4 ? atoi(argv[4]) : 2, probe = argc > 5 ? atoi(argv[5]) : 0, epi = argc > 6 ? atoi(argv[6]) : 1;
    const long ldc = (N + 31) / 32 * 32;
    float *A, *B, *C, *bias;
    void* Apk;
    hipMalloc(&A, (size_t)M * K * 4);
    hipMalloc(&B, (size_t)batch * K * N * 4);
    hipMalloc(&C, (size_t)batch * M * ldc * 4);
    hipMalloc(&bias, (size_t)M * 4);
    std::vector<float> h((size_t)std::max((long)M * K, (long)batch * K * N));
    unsigned s = 12345u;
    for (auto& v : h) { s = s * 1664525u + 1013904223u; v = ((int)(s >> 9) & 0xffff) * (1.f / 32768.f) - 1.f; }
    hipMemcpy(A, h.data(), (size_t)M * K * 4, hipMemcpyHostToDevice);
    hipMemcpy(B, h.data(), (size_t)batch * K * N * 4, hipMemcpyHostToDevice);
    hipMemcpy(bias, h.data(), (size_t)M * 4, hipMemcpyHostToDevice);
    const long per = occd_gemm_x3_packed_elems(M, K);
    hipMalloc(&Apk, (size_t)per * 2);
    if (occd_gemm_x3_pack(A, Apk, M, K, K, 0, 1, 0, nullptr) != 0) { printf("pack failed\n"); return 1; }
    occd_gemm_args a = {};
    a.A = (const float*)Apk; a.B = B; a.C = C; a.bias = epi ? bias : nullptr;
    a.M = M; a.N = N; a.K = K; a.batch = batch;
    a.lda = K; a.ldb = N; a.ldc = ldc; a.stride_a = 0; a.stride_b = (long)K * N; a.stride_c = (long)M * ldc;
    a.act = epi ? OCCD_GEMM_ACT_SWISH : OCCD_GEMM_ACT_NONE; a.slope = 0.f; a.tile_hint = 8; a.pre = 1;
    hipMemcpyToSymbol(HIP_SYMBOL(g_panel_probe_wg), &probe, sizeof(int));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 6; ++rep) {
        hipEventRecord(e0, 0);
        const int rc = occd_gemm_f32x3(&a, nullptr);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        if (rc != 0) { printf("occd_gemm_f32x3 failed: %d\n", rc); return 1; }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep > 0 && ms < best) best = ms;
    }
    unsigned long long tl[8 * 64];
    hipMemcpyFromSymbol(tl, HIP_SYMBOL(g_panel_tl), sizeof(tl));
    const int K16 = (K + 15) / 16;
    printf("M %d N %d K %d batch %d: launch %.1f us (events); workgroup %d, shader-clock cycles relative to the first wave's entry\n", M, N, K, batch, best * 1e3, probe);
    unsigned long long t0 = ~0ull;
    for (int w = 0; w < 8; ++w) if (tl[w * 64] && tl[w * 64] < t0) t0 = tl[w * 64];
    printf("wave   entry  A-req  B-req B-split barrier | first 16-k steps (delta) ... | loop-issued stores-issued  exit\n");
    for (int w = 0; w < 8; ++w) {
        const unsigned long long* t = tl + w * 64;
        if (!t[0]) continue;
        printf("%4d %7llu %6llu %6llu %7llu %7llu |", w, t[0] - t0, t[1] - t0, t[2] - t0, t[3] - t0, t[4] - t0);
        for (int k = 0; k < K16 && k < 40; ++k) printf(" %llu", k == 0 ? t[8] - t[4] : t[8 + k] - t[8 + k - 1]);
        printf(" | %7llu %7llu %7llu\n", t[5] - t0, t[6] - t0, t[7] - t0);
    }
    return 0;
}
