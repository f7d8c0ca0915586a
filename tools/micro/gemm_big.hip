// Prototype: fp32 MFMA GEMM with a LARGE per-wave tile (register blocking) and a double-buffered LDS stage, one barrier
// per k-tile.  C[M,N] = A[M,K] . B[N,K]^T, both K-contiguous (the encoder conv shapes).  Stand-alone timing probe.
//   hipcc --offload-arch=gfx950 -O3 -o tools/micro/gemm_big tools/micro/gemm_big.hip && tools/micro/gemm_big
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int BM, int BN, int WGM, int WGN, int BK, int OCC>
__global__ __launch_bounds__(WGM* WGN * 64, OCC) void gemm_big(const float* __restrict__ A, const float* __restrict__ B,
                                                                 float* __restrict__ C, int M, int N, int K) {
    constexpr int NT = WGM * WGN * 64, LDK = BK + 4, KQ = BK / 4;
    constexpr int WM = BM / WGM, WN = BN / WGN, TM = WM / 32, TN = WN / 32;
    constexpr int NA = BM * KQ / NT, NB = BN * KQ / NT;
    static_assert(BM * KQ % NT == 0 && BN * KQ % NT == 0, "whole staging passes");
    __shared__ __attribute__((aligned(16))) float As[2][BM * LDK];
    __shared__ __attribute__((aligned(16))) float Bs[2][BN * LDK];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, wm = wid / WGN, wn = wid % WGN, r = lane & 31, h = lane >> 5;
    const int tiles_n = N / BN;
    // XCD-aware remap: consecutive tiles (sharing an A panel) on one XCD
    const int nwg = gridDim.x;
    int bid = blockIdx.x;
    { const int q = nwg >> 3, rr = nwg & 7, xcd = bid & 7; bid = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + (bid >> 3); }
    const int tile_n = bid % tiles_n, tile_m = bid / tiles_n;
    const int row0 = tile_m * BM, col0 = tile_n * BN;
    const float* pa[NA];
    const float* pb[NB];
    int sa[NA], sb[NB];
#pragma unroll
    for (int e = 0; e < NA; ++e) {
        const int idx = tid + e * NT, row = idx / KQ, kq = idx % KQ;
        pa[e] = A + (long)(row0 + row) * K + kq * 4;
        sa[e] = row * LDK + kq * 4;
    }
#pragma unroll
    for (int e = 0; e < NB; ++e) {
        const int idx = tid + e * NT, row = idx / KQ, kq = idx % KQ;
        pb[e] = B + (long)(col0 + row) * K + kq * 4;
        sb[e] = row * LDK + kq * 4;
    }
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    float4 ra[NA], rb[NB];
    auto fetch = [&]() {
#pragma unroll
        for (int e = 0; e < NA; ++e) { ra[e] = *(const float4*)pa[e]; pa[e] += BK; }
#pragma unroll
        for (int e = 0; e < NB; ++e) { rb[e] = *(const float4*)pb[e]; pb[e] += BK; }
    };
    auto stage = [&](int buf) {
#pragma unroll
        for (int e = 0; e < NA; ++e) *(float4*)&As[buf][sa[e]] = ra[e];
#pragma unroll
        for (int e = 0; e < NB; ++e) *(float4*)&Bs[buf][sb[e]] = rb[e];
    };
    fetch();
    stage(0);
    __syncthreads();
    const int nk = K / BK;
    for (int it = 0; it < nk; ++it) {
        const int buf = it & 1;
        if (it + 1 < nk) fetch();
        const float* as = As[buf];
        const float* bs = Bs[buf];
#pragma unroll
        for (int t = 0; t < BK / 8; ++t) {
            float4 a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) a[i] = *(const float4*)&as[(wm * WM + i * 32 + r) * LDK + 8 * t + 4 * h];
#pragma unroll
            for (int j = 0; j < TN; ++j) b[j] = *(const float4*)&bs[(wn * WN + j * 32 + r) * LDK + 8 * t + 4 * h];
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    const float av = s == 0 ? a[i].x : (s == 1 ? a[i].y : (s == 2 ? a[i].z : a[i].w));
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        const float bv = s == 0 ? b[j].x : (s == 1 ? b[j].y : (s == 2 ? b[j].z : b[j].w));
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[i][j], 0, 0, 0);
                    }
                }
        }
        if (it + 1 < nk) stage(buf ^ 1);
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = col0 + wn * WN + j * 32 + r;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = row0 + wm * WM + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                C[(long)row * N + col] = fmaxf(acc[i][j][e], 0.f);
            }
        }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int BM, int BN, int WGM, int WGN, int BK, int OCC>
void run(const char* name, const float* dA, const float* dB, float* dC, int M, int N, int K, const std::vector<float>& hA,
         const std::vector<float>& hB) {
    dim3 grid((M / BM) * (N / BN)), block(WGM * WGN * 64);
    hipEvent_t s, e;
    CK(hipEventCreate(&s)); CK(hipEventCreate(&e));
    for (int i = 0; i < 200; ++i) hipLaunchKernelGGL((gemm_big<BM, BN, WGM, WGN, BK, OCC>), grid, block, 0, 0, dA, dB, dC, M, N, K);
    CK(hipEventRecord(s));
    for (int i = 0; i < 100; ++i) hipLaunchKernelGGL((gemm_big<BM, BN, WGM, WGN, BK, OCC>), grid, block, 0, 0, dA, dB, dC, M, N, K);
    CK(hipEventRecord(e));
    CK(hipEventSynchronize(e));
    float ms;
    CK(hipEventElapsedTime(&ms, s, e));
    ms /= 100;
    // spot check
    std::vector<float> hC(1024);
    double maxerr = 0;
    for (int q = 0; q < 8; ++q) {
        const int row = (q * 7919 + 13) % M, col = (q * 104729 + 7) % N;
        float got;
        CK(hipMemcpy(&got, dC + (long)row * N + col, 4, hipMemcpyDeviceToHost));
        double want = 0;
        for (int k = 0; k < K; ++k) want += (double)hA[(long)row * K + k] * hB[(long)col * K + k];
        want = want > 0 ? want : 0;
        maxerr = std::fmax(maxerr, std::fabs(got - want) / (1e-3 + std::fabs(want)));
    }
    printf("%-28s M=%d N=%d K=%d: %8.1f us  %6.1f TFLOP/s  (max rel err %.1e)\n", name, M, N, K, ms * 1e3, 2.0 * M * N * K / ms / 1e9, maxerr);
}

int main() {
    const int shapes[][3] = {{65536, 512, 512}, {8192, 8192, 8192}};
    for (auto& sh : shapes) {
        const int M = sh[0], N = sh[1], K = sh[2];
        std::vector<float> hA((long)M * K), hB((long)N * K);
        for (auto& v : hA) v = (rand() % 2001 - 1000) * 1e-3f;
        for (auto& v : hB) v = (rand() % 2001 - 1000) * 1e-3f;
        float *dA, *dB, *dC;
        CK(hipMalloc(&dA, hA.size() * 4)); CK(hipMalloc(&dB, hB.size() * 4)); CK(hipMalloc(&dC, (long)M * N * 4));
        CK(hipMemcpy(dA, hA.data(), hA.size() * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(dB, hB.data(), hB.size() * 4, hipMemcpyHostToDevice));
        run<256, 128, 2, 2, 16, 2>("256x128 4w(128x64) BK16 occ2", dA, dB, dC, M, N, K, hA, hB);
        run<256, 128, 4, 2, 16, 2>("256x128 8w(64x64) BK16 occ2", dA, dB, dC, M, N, K, hA, hB);
        run<256, 128, 4, 2, 16, 3>("256x128 8w(64x64) BK16 occ3", dA, dB, dC, M, N, K, hA, hB);
        run<256, 128, 4, 2, 32, 2>("256x128 8w(64x64) BK32 occ2", dA, dB, dC, M, N, K, hA, hB);
        run<128, 128, 4, 2, 16, 6>("128x128 8w(32x64) BK16 occ6", dA, dB, dC, M, N, K, hA, hB);
        run<128, 128, 4, 2, 16, 4>("128x128 8w(32x64) BK16 occ4", dA, dB, dC, M, N, K, hA, hB);
        run<256, 256, 4, 2, 16, 2>("256x256 8w(64x128) BK16 occ2", dA, dB, dC, M, N, K, hA, hB);
        run<128, 256, 2, 4, 16, 4>("128x256 8w(64x64) BK16 occ4", dA, dB, dC, M, N, K, hA, hB);
        CK(hipFree(dA)); CK(hipFree(dB)); CK(hipFree(dC));
    }
    return 0;
}
