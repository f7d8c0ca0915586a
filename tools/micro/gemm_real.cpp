// The library's GEMM at the roofline launch's shape, driven from C++ with the probe's data and timing loop.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include "hyperpocket_hip.h"
static float gauss() { float u = (rand() + 1.f) / (RAND_MAX + 2.f), v = (rand() + 1.f) / (RAND_MAX + 2.f); return sqrtf(-2 * logf(u)) * cosf(6.2831853f * v); }
int main(int argc, char** argv) {
    const int M = 65536, N = 512, K = 512;
    const int normal = argc > 1 ? atoi(argv[1]) : 0;
    std::vector<float> ha((size_t)M * K), hb((size_t)N * K);
    for (auto& v : ha) v = normal ? gauss() : (float)rand() / RAND_MAX - 0.5f;
    for (auto& v : hb) v = normal ? 0.05f * gauss() : (float)rand() / RAND_MAX - 0.5f;
    float *A, *B, *C, *bias;
    hipMalloc(&A, (size_t)M * K * 4); hipMalloc(&B, (size_t)N * K * 4); hipMalloc(&C, (size_t)M * N * 4); hipMalloc(&bias, N * 4);
    hipMemcpy(A, ha.data(), (size_t)M * K * 4, hipMemcpyHostToDevice);
    hipMemcpy(B, hb.data(), (size_t)N * K * 4, hipMemcpyHostToDevice);
    hipMemset(bias, 0, N * 4);
    HpGemmDesc d = {};
    d.A = A; d.sAi = K; d.sAk = 1; d.B = B; d.sBk = 1; d.sBj = K; d.C = C; d.ldc = N; d.bias = bias;
    d.M = M; d.N = N; d.K = K; d.batch = 1; d.flags = HP_GEMM_BIAS;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 6; ++rep) {
        for (int i = 0; i < 3; ++i) hp_gemm_f32(&d, 0);
        hipEventRecord(e0);
        for (int i = 0; i < 20; ++i) hp_gemm_f32(&d, 0);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 20;
        printf("library gemm (%s data) %7.1f us  %6.1f TFLOP/s\n", normal ? "normal" : "uniform", ms * 1e3, 2.0 * M * N * K / ms / 1e9); fflush(stdout);
    }
    return 0;
}
