// Bare fp32 MFMA loops on random operands: 32x32x2 vs 16x16x4 (same FLOP per cycle on paper) — which clock does each hold?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int SHAPE>
__global__ __launch_bounds__(256) void loop_kernel(const float* __restrict__ in, float* __restrict__ out, int iters) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    float a[8], b[8];
    for (int i = 0; i < 8; ++i) { a[i] = in[(t * 16 + i) & 0xffff]; b[i] = in[(t * 16 + 8 + i) & 0xffff]; }
    if (SHAPE == 32) {
        f32x16 c0 = {0}, c1 = {0};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[i], c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[7 - i], c1, 0, 0, 0);
            }
        }
        float s = 0; for (int e = 0; e < 16; ++e) s += c0[e] + c1[e];
        out[t] = s;
    } else {
        f32x4 c[8] = {};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
#pragma unroll
                for (int j = 0; j < 4; ++j) c[(i & 1) * 4 + j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[(i + j) & 7], c[(i & 1) * 4 + j], 0, 0, 0);
            }
        }
        float s = 0; for (int k = 0; k < 8; ++k) for (int e = 0; e < 4; ++e) s += c[k][e];
        out[t] = s;
    }
}

int main() {
    const int n = 1 << 16;
    std::vector<float> h(n);
    for (auto& v : h) v = (float)rand() / RAND_MAX - 0.5f;
    float *din, *dout;
    hipMalloc(&din, n * 4); hipMalloc(&dout, 4 << 20);
    hipMemcpy(din, h.data(), n * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int waves = 1; waves <= 4; waves *= 2) {
        const int blocks = 256 * waves;   // 4 waves per block -> `waves` waves per SIMD
        for (int shape : {32, 16, 32, 16}) {
            const int iters = 20000;
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                if (shape == 32) hipLaunchKernelGGL(loop_kernel<32>, dim3(blocks), dim3(256), 0, 0, din, dout, iters);
                else hipLaunchKernelGGL(loop_kernel<16>, dim3(blocks), dim3(256), 0, 0, din, dout, iters);
                hipEventRecord(e1); hipEventSynchronize(e1);
            }
            float ms; hipEventElapsedTime(&ms, e0, e1);
            // flops: 32x32x2: 16 mfma/iter * 4096 ; 16x16x4: 32 mfma/iter * 2048  (same)
            const double fl = (double)blocks * 4 * iters * 16 * 4096.0;
            printf("waves/SIMD %d  shape %2d: %.2f ms  %.1f TFLOP/s\n", waves, shape, ms, fl / ms / 1e9);
        }
    }
    return 0;
}
