// Single-wave issue rate of v_mfma_f32_32x32x2_f32: cycles per MFMA for NACC independent accumulators, 1 or 2 waves per SIMD.
// hipcc --offload-arch=gfx950 -O3 -o tools/micro/mfma_rate tools/micro/mfma_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC, bool LOADS>
__global__ __launch_bounds__(256) void k(float* out, const float* in, long long* cyc, int iters) {
    f32x16 acc[NACC];
    for (int j = 0; j < NACC; ++j) for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
    float a = in[threadIdx.x], b = in[threadIdx.x + 256];
    const float* p = in + threadIdx.x * 2;
    long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        float2 v[8];
        if (LOADS) {
#pragma unroll
            for (int s = 0; s < 8; ++s) v[s] = *reinterpret_cast<const float2*>(p + s * 512 + (it & 7) * 4096);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int s = 0; s < 8; ++s)
#pragma unroll
            for (int j = 0; j < NACC; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(LOADS ? v[s].x : a, LOADS ? v[s].y : b, acc[j], 0, 0, 0);
        if (LOADS) __builtin_amdgcn_sched_barrier(0);
    }
    long long t1 = clock64();
    float s = 0.f;
    for (int j = 0; j < NACC; ++j) for (int e = 0; e < 16; ++e) s += acc[j][e];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int NACC, bool LOADS>
void run(int wgs, float* out, float* in, long long* cyc) {
    const int iters = 2000;
    hipLaunchKernelGGL((k<NACC, LOADS>), dim3(wgs), dim3(256), 0, 0, out, in, cyc, iters);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<NACC, LOADS>), dim3(wgs), dim3(256), 0, 0, out, in, cyc, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long h[4]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    const double nm = (double)iters * 8 * NACC;
    printf("NACC %d loads %d wgs %d: %.1f clock64-cycles per MFMA per wave, kernel %.1f us -> %.1f TFLOP/s\n", NACC, (int)LOADS, wgs, h[0] / nm, ms * 1e3,
           (double)wgs * 4 * nm * 4096 / (ms * 1e-3) * 1e-12);
}
int main() {
    float *out, *in; long long* cyc;
    hipMalloc(&out, 4 << 20); hipMalloc(&in, 64 << 20); hipMalloc(&cyc, 8 << 12);
    hipMemset(in, 0, 64 << 20);
    for (int wgs : {256, 512, 768}) {
        run<1, false>(wgs, out, in, cyc);
        run<2, false>(wgs, out, in, cyc);
        run<4, false>(wgs, out, in, cyc);
        run<4, true>(wgs, out, in, cyc);
    }
    return 0;
}
