// Prototype 3: 256x256 tile, BK = 32 floats, 8 waves (2 x 4) of 128x64, LDS filled by global_load_lds_dwordx4 (no staging
// registers), two LDS buffers, one barrier per k-tile; XOR-swizzled LDS image (swizzle applied on the glds SOURCE address).
//   C = relu(A[M,K] . B[N,K]^T)      hipcc --offload-arch=gfx950 -O3 -o tools/micro/gemm_glds tools/micro/gemm_glds.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
#include <algorithm>
typedef float f32x16 __attribute__((ext_vector_type(16)));

#ifndef SETPRIO
#define SETPRIO 0
#endif
#ifndef PIPE
#define PIPE 0
#endif
#ifndef PERSIST
#define PERSIST 0
#endif
constexpr int BM = 256, BN = 256, BK = 32, NT = 512;
constexpr int TM = 4, TN = 2;                       // per wave: 128 x 64
constexpr int kOpFloats = 256 * BK;                 // one operand tile
constexpr int kBufFloats = 2 * kOpFloats;           // A | B

__device__ __forceinline__ float sel(const float4& v, int s) { return s == 0 ? v.x : (s == 1 ? v.y : (s == 2 ? v.z : v.w)); }
__device__ __forceinline__ int swz(int row) { return (row >> 1) & 7; }

__global__ __launch_bounds__(NT, 1) void gemm_glds(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C,
                                                     int M, int N, int K) {
    __shared__ __attribute__((aligned(1024))) float lds[2 * kBufFloats];   // 128 KB
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, wm = wid >> 2, wn = wid & 3, r = lane & 31, h = lane >> 5;
    const int tiles_n = N / BN;
    const int ntiles = (M / BM) * tiles_n;
    const int nwg = PERSIST ? ntiles : gridDim.x;
    f32x16 acc[TM][TN];
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    int bid = tile;
    { const int q = nwg >> 3, rr = nwg & 7, xcd = bid & 7; bid = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + (bid >> 3); }
    const int tile_n = bid % tiles_n, tile_m = bid / tiles_n;
    const int row0 = tile_m * BM, col0 = tile_n * BN;
    // glds: instruction e of wave w fills rows [(e*8 + w)*8, +8) of an operand tile; lane -> (row, physical slot)
    const float* ga[4];
    const float* gb[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int row = (e * 8 + wid) * 8 + (lane >> 3), p = lane & 7, s = p ^ swz(row);
        ga[e] = A + (long)(row0 + row) * K + 4 * s;
        gb[e] = B + (long)(col0 + row) * K + 4 * s;
    }
    // LDS-DMA through inline asm: with the builtin, hipcc orders every later ds_read behind the pending LDS write with an
    // s_waitcnt vmcnt(0) — i.e. it drains the prefetch the moment it is issued.  The asm form is invisible to that analysis;
    // the explicit vmcnt(0) + barrier at the end of the iteration is what orders the next tile's reads.
    auto glds16 = [&](const float* gsrc, unsigned lds_dst) {
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
    };
    const unsigned lds0 = (unsigned)(uintptr_t)lds;
    auto issue = [&](int buf, int k0) {
        const unsigned base = lds0 + (unsigned)(buf * kBufFloats) * 4u;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            glds16(ga[e] + k0, __builtin_amdgcn_readfirstlane(base + (unsigned)((e * 8 + wid) * 8 * BK) * 4u));
            glds16(gb[e] + k0, __builtin_amdgcn_readfirstlane(base + (unsigned)(kOpFloats + (e * 8 + wid) * 8 * BK) * 4u));
        }
    };
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    // fragment addresses (floats) inside an operand tile: row*32 + 4*((2t+h) ^ swz(row))
    int fa[TM], fb[TN], xa[TM], xb[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) { const int row = wm * 128 + i * 32 + r; fa[i] = row * BK; xa[i] = swz(row); }
#pragma unroll
    for (int j = 0; j < TN; ++j) { const int row = wn * 64 + j * 32 + r; fb[j] = kOpFloats + row * BK; xb[j] = swz(row); }
    const int nk = K / BK;
#if PIPE
    struct Frag { float4 a[TM], b[TN]; };
    auto ldfrag = [&](Frag& f, const float* base, int t) {
#pragma unroll
        for (int i = 0; i < TM; ++i) f.a[i] = *(const float4*)&base[fa[i] + 4 * ((2 * t + h) ^ xa[i])];
#pragma unroll
        for (int j = 0; j < TN; ++j) f.b[j] = *(const float4*)&base[fb[j] + 4 * ((2 * t + h) ^ xb[j])];
    };
    auto mfma32 = [&](const Frag& f) {
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(sel(f.a[i], s), sel(f.b[j], s), acc[i][j], 0, 0, 0);
    };
    auto issue_half = [&](int buf, int k0, int half) {   // 4 of the 8 LDS-DMA pieces of a tile
        const unsigned base = lds0 + (unsigned)(buf * kBufFloats) * 4u;
#pragma unroll
        for (int e = 2 * half; e < 2 * half + 2; ++e) {
            glds16(ga[e] + k0, __builtin_amdgcn_readfirstlane(base + (unsigned)((e * 8 + wid) * 8 * BK) * 4u));
            glds16(gb[e] + k0, __builtin_amdgcn_readfirstlane(base + (unsigned)(kOpFloats + (e * 8 + wid) * 8 * BK) * 4u));
        }
    };
    issue(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    Frag f0, f1;
    ldfrag(f0, lds, 0);
    for (int it = 0; it < nk; ++it) {
        const int buf = it & 1;
        const float* base = lds + buf * kBufFloats;
        const float* nbase = lds + (buf ^ 1) * kBufFloats;
        const int k1 = min(it + 1, nk - 1) * BK;           // (past the end: a harmless re-load of the last tile)
        ldfrag(f1, base, 1);
        issue_half(buf ^ 1, k1, 0);
        __builtin_amdgcn_sched_barrier(0);
        mfma32(f0);                                         // t = 0
        __builtin_amdgcn_sched_barrier(0);
        ldfrag(f0, base, 2);
        issue_half(buf ^ 1, k1, 1);
        __builtin_amdgcn_sched_barrier(0);
        mfma32(f1);                                         // t = 1
        __builtin_amdgcn_sched_barrier(0);
        ldfrag(f1, base, 3);
        __builtin_amdgcn_sched_barrier(0);
        mfma32(f0);                                         // t = 2
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // tile it+1 landed; t = 3 fragments in registers
        __builtin_amdgcn_s_barrier();                       // nobody reads `buf` after this point
        ldfrag(f0, nbase, 0);
        __builtin_amdgcn_sched_barrier(0);
        mfma32(f1);                                         // t = 3
        __builtin_amdgcn_sched_barrier(0);
    }
#else
    issue(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int it = 0; it < nk; ++it) {
        const int buf = it & 1;
        if (it + 1 < nk) issue(buf ^ 1, (it + 1) * BK);
        const float* base = lds + buf * kBufFloats;
#pragma unroll
        for (int t = 0; t < BK / 8; ++t) {
            float4 a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) a[i] = *(const float4*)&base[fa[i] + 4 * ((2 * t + h) ^ xa[i])];
#pragma unroll
            for (int j = 0; j < TN; ++j) b[j] = *(const float4*)&base[fb[j] + 4 * ((2 * t + h) ^ xb[j])];
            if (SETPRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(sel(a[i], s), sel(b[j], s), acc[i][j], 0, 0, 0);
            if (SETPRIO) __builtin_amdgcn_s_setprio(0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
#endif
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = col0 + wn * 64 + j * 32 + r;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = row0 + wm * 128 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                C[(long)row * N + col] = fmaxf(acc[i][j][e], 0.f);
            }
        }
    __syncthreads();
  }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

int main() {
    const int shapes[][3] = {{65536, 512, 512}, {65536, 512, 256}, {65536, 256, 128}, {8192, 8192, 8192}};
    for (auto& sh : shapes) {
        const int M = sh[0], N = sh[1], K = sh[2];
        std::vector<float> hA((long)M * K), hB((long)N * K);
        for (auto& v : hA) v = (rand() % 2001 - 1000) * 1e-3f;
        for (auto& v : hB) v = (rand() % 2001 - 1000) * 1e-3f;
        float *dA, *dB, *dC;
        CK(hipMalloc(&dA, hA.size() * 4)); CK(hipMalloc(&dB, hB.size() * 4)); CK(hipMalloc(&dC, (long)M * N * 4));
        CK(hipMemcpy(dA, hA.data(), hA.size() * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(dB, hB.data(), hB.size() * 4, hipMemcpyHostToDevice));
        dim3 grid(PERSIST ? std::min(256, (M / BM) * (N / BN)) : (M / BM) * (N / BN)), block(NT);
        hipEvent_t s, e;
        CK(hipEventCreate(&s)); CK(hipEventCreate(&e));
        for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(gemm_glds, grid, block, 0, 0, dA, dB, dC, M, N, K);
        CK(hipEventRecord(s));
        for (int i = 0; i < 100; ++i) hipLaunchKernelGGL(gemm_glds, grid, block, 0, 0, dA, dB, dC, M, N, K);
        CK(hipEventRecord(e));
        CK(hipEventSynchronize(e));
        float ms;
        CK(hipEventElapsedTime(&ms, s, e));
        ms /= 100;
        double maxerr = 0;
        for (int q = 0; q < 64; ++q) {
            const int row = (q * 7919 + 13) % M, col = (q * 104729 + 7) % N;
            float got;
            CK(hipMemcpy(&got, dC + (long)row * N + col, 4, hipMemcpyDeviceToHost));
            double want = 0;
            for (int k = 0; k < K; ++k) want += (double)hA[(long)row * K + k] * hB[(long)col * K + k];
            want = want > 0 ? want : 0;
            maxerr = std::fmax(maxerr, std::fabs(got - want) / (1e-3 + std::fabs(want)));
        }
        printf("gemm_glds 256x256x32 M=%d N=%d K=%d: %8.1f us  %6.1f TFLOP/s  (max rel err %.1e)\n", M, N, K, ms * 1e3,
               2.0 * M * N * K / ms / 1e9, maxerr);
        CK(hipFree(dA)); CK(hipFree(dB)); CK(hipFree(dC));
    }
    return 0;
}
