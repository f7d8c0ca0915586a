// Where does the 128x128x16 8-wave GEMM body lose its MFMA rate?  Variants of the main loop with parts removed.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int BM = 128, BN = 128, BK = 16, LDK = 20, NT = 512;

// MODE bits: 1 = global loads, 2 = LDS stage writes + barriers, 4 = LDS fragment reads
template <int MODE>
__global__ __launch_bounds__(512, 6) void probe(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C, int M, int N, int K) {
    __shared__ __attribute__((aligned(16))) float smem[(BM + BN) * LDK];
    float* As = smem;
    float* Bs = smem + BM * LDK;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, wm = wid >> 1, wn = wid & 1, r = lane & 31, h = lane >> 5;
    const int tiles_n = N / BN, tile_n = blockIdx.x % tiles_n, tile_m = blockIdx.x / tiles_n;
    const int row0 = tile_m * BM, col0 = tile_n * BN;
    const int srow = tid >> 2, skq = tid & 3;
    const float* pa = A + (long)(row0 + srow) * K + skq * 4;
    const float* pb = B + (long)(col0 + srow) * K + skq * 4;
    f32x16 acc[2];
    for (int e = 0; e < 16; ++e) acc[0][e] = acc[1][e] = 0.f;
    float4 ra = make_float4(1.f, 2.f, 3.f, 4.f), rb = make_float4(.5f, .25f, .125f, 1.f);
    if (MODE & 1) { ra = *(const float4*)pa; rb = *(const float4*)pb; pa += BK; pb += BK; }
    if (!(MODE & 2)) { *(float4*)&As[srow * LDK + skq * 4] = ra; *(float4*)&Bs[srow * LDK + skq * 4] = rb; __syncthreads(); }
    float4 fa[2], fb0[2], fb1[2];
    if (!(MODE & 4)) {
        for (int t = 0; t < 2; ++t) {
            fa[t] = *(const float4*)&As[(wm * 32 + r) * LDK + 8 * t + 4 * h];
            fb0[t] = *(const float4*)&Bs[(wn * 64 + r) * LDK + 8 * t + 4 * h];
            fb1[t] = *(const float4*)&Bs[(wn * 64 + 32 + r) * LDK + 8 * t + 4 * h];
        }
    }
    for (int k0 = 0; k0 < K; k0 += BK) {
        if (MODE & 2) {
            *(float4*)&As[srow * LDK + skq * 4] = ra;
            *(float4*)&Bs[srow * LDK + skq * 4] = rb;
            __syncthreads();
        }
        if ((MODE & 1) && k0 + BK < K) { ra = *(const float4*)pa; rb = *(const float4*)pb; pa += BK; pb += BK; }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            float4 a, b0, b1;
            if (MODE & 4) {
                a = *(const float4*)&As[(wm * 32 + r) * LDK + 8 * t + 4 * h];
                b0 = *(const float4*)&Bs[(wn * 64 + r) * LDK + 8 * t + 4 * h];
                b1 = *(const float4*)&Bs[(wn * 64 + 32 + r) * LDK + 8 * t + 4 * h];
            } else { a = fa[t]; b0 = fb0[t]; b1 = fb1[t]; }
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b0.x, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b1.x, acc[1], 0, 0, 0);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b0.y, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b1.y, acc[1], 0, 0, 0);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b0.z, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b1.z, acc[1], 0, 0, 0);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b0.w, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b1.w, acc[1], 0, 0, 0);
        }
        if (MODE & 2) __syncthreads();
    }
    if (MODE & 8) {           // no C store: one value per thread keeps the accumulators alive
        float s = 0.f;
        for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) s += acc[j][e];
        C[(long)blockIdx.x * NT + tid] = s;
    } else if (MODE & 16) {   // transpose through LDS, 16-byte stores: a wave writes its 32x64 tile as 32 rows x 256 B
        __syncthreads();
        float* T = As + wid * (32 * 17);                    // per-wave 32 x (16+1) staging, reused 4 times (16 cols each)
#pragma unroll
        for (int q = 0; q < 4; ++q) {                        // 16-column slab q of the wave's 64 columns: held by lanes r in [16*(q&1), +16) of tile j=q>>1
            const int j = q >> 1, rb = (q & 1) * 16;
            if (r >= rb && r < rb + 16)
#pragma unroll
                for (int e = 0; e < 16; ++e) T[((e & 3) + 8 * (e >> 2) + 4 * h) * 17 + (r - rb)] = acc[j][e];
            // 32 rows x 16 cols -> lane l stores row l>>1... 64 lanes x float4 = 32 rows x 8 floats: two passes
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const int row = lane >> 1, c4 = (lane & 1) * 4 + half * 8;
                float4 v = make_float4(T[row * 17 + c4], T[row * 17 + c4 + 1], T[row * 17 + c4 + 2], T[row * 17 + c4 + 3]);
                *(float4*)&C[(long)(row0 + wm * 32 + row) * N + col0 + wn * 64 + q * 16 + c4] = v;
            }
        }
    } else {
        for (int j = 0; j < 2; ++j)
            for (int e = 0; e < 16; ++e) {
                const int row = row0 + wm * 32 + (e & 3) + 8 * (e >> 2) + 4 * h, col = col0 + wn * 64 + j * 32 + r;
                C[(long)row * N + col] = acc[j][e];
            }
    }
}

template <int MODE>
void run(const char* name, const float* A, const float* B, float* C, int M, int N, int K) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(probe<MODE>, dim3((M / BM) * (N / BN)), dim3(NT), 0, 0, A, B, C, M, N, K);
    hipEventRecord(e0);
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(probe<MODE>, dim3((M / BM) * (N / BN)), dim3(NT), 0, 0, A, B, C, M, N, K);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 20;
    printf("%-44s %7.1f us  %6.1f TFLOP/s\n", name, ms * 1e3, 2.0 * M * N * K / ms / 1e9); fflush(stdout);
}

int main() {
    const int M = 65536, N = 512, K = 512;
    std::vector<float> h((size_t)M * K);
    for (auto& v : h) v = (float)rand() / RAND_MAX - 0.5f;
    float *A, *B, *C;
    hipMalloc(&A, (size_t)M * K * 4); hipMalloc(&B, (size_t)N * K * 4); hipMalloc(&C, (size_t)M * N * 4);
    hipMemcpy(A, h.data(), (size_t)M * K * 4, hipMemcpyHostToDevice);
    hipMemcpy(B, h.data(), (size_t)N * K * 4, hipMemcpyHostToDevice);
    run<7>("full: global + LDS stage + frag reads", A, B, C, M, N, K);
    run<6>("no global loads", A, B, C, M, N, K);
    run<4>("frag reads + MFMA (no stage, no barriers)", A, B, C, M, N, K);
    run<0>("MFMA only (operands in registers)", A, B, C, M, N, K);
    run<3>("global + stage + barriers, frags in regs", A, B, C, M, N, K);
    run<7>("full again", A, B, C, M, N, K);
    run<8>("MFMA only, no C store", A, B, C, M, N, K);
    run<15>("full loop, no C store", A, B, C, M, N, K);
    run<7>("full again", A, B, C, M, N, K);
    run<23>("full loop, C via LDS transpose + 16-byte stores", A, B, C, M, N, K);
    run<7>("full again", A, B, C, M, N, K);
    return 0;
}
