// What a "pipeline stage" costs on gfx950 at one wave per SIMD: 24 v_mfma_f32_32x32x16_bf16 (768 cycles) plus, by mode bit,
//   1: their 12 B operands re-read from LDS (ds_read_b128) each stage          2: one s_barrier per stage
//   4: 88 vector instructions (the and/sub/perm split of heads_fwd.hip)         8: 7 global_load_lds_dwordx4 per stage
//  16: the 4 A-operand raw reads (ds_read_b128) feeding the split        32: the split placed BETWEEN the MFMAs by hand (sched_barrier)
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/micro/stage_probe.hip -o tools/micro/stage_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void split3(const float (&x)[8], bf16x8& p1, bf16x8& p2, bf16x8& p3) {
    u32x4 a, b, c;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float x0 = x[2 * i], x1 = x[2 * i + 1];
        const float h0 = __uint_as_float(__float_as_uint(x0) & 0xffff0000u), h1 = __uint_as_float(__float_as_uint(x1) & 0xffff0000u);
        const float r0 = x0 - h0, r1 = x1 - h1;
        const float m0 = __uint_as_float(__float_as_uint(r0) & 0xffff0000u), m1 = __uint_as_float(__float_as_uint(r1) & 0xffff0000u);
        const float s0 = r0 - m0, s1 = r1 - m1;
        a[i] = __builtin_amdgcn_perm(__float_as_uint(x1), __float_as_uint(x0), 0x07060302u);
        b[i] = __builtin_amdgcn_perm(__float_as_uint(r1), __float_as_uint(r0), 0x07060302u);
        c[i] = __builtin_amdgcn_perm(__float_as_uint(s1), __float_as_uint(s0), 0x07060302u);
    }
    p1 = __builtin_bit_cast(bf16x8, a);
    p2 = __builtin_bit_cast(bf16x8, b);
    p3 = __builtin_bit_cast(bf16x8, c);
}
__device__ __forceinline__ void glds16(const void* sbase, unsigned voff, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}

template <int MODE>
__global__ __launch_bounds__(256, 1) void probe(const unsigned char* src, float* out, long long* cyc, int iters) {
    __shared__ __attribute__((aligned(1024))) unsigned char lds[96 * 1024];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    for (int i = tid; i < 96 * 1024 / 16; i += 256) reinterpret_cast<u32x4*>(lds)[i] = u32x4{0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
    __syncthreads();
    f32x16 acc[2];
    for (int t = 0; t < 2; ++t)
        for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
    bf16x8 a1[2], a2[2], a3[2], b1[2][2], b2[2][2], b3[2][2];
    for (int u = 0; u < 2; ++u) {
        a1[u] = a2[u] = a3[u] = *reinterpret_cast<const bf16x8*>(lds + lane * 16);
        for (int t = 0; t < 2; ++t) b1[u][t] = b2[u][t] = b3[u][t] = *reinterpret_cast<const bf16x8*>(lds + 1024 + lane * 16);
    }
    const unsigned lds0 = (unsigned)(size_t)lds;
    const unsigned sdst = __builtin_amdgcn_readfirstlane(lds0 + 64 * 1024 + w * 7168);
    const unsigned long long p = reinterpret_cast<unsigned long long>(src);
    const void* sb = reinterpret_cast<const void*>(((unsigned long long)__builtin_amdgcn_readfirstlane((unsigned)(p >> 32)) << 32) |
                                                   __builtin_amdgcn_readfirstlane((unsigned)p));
    const long long c0 = clock64();
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
        if (MODE & 2) __builtin_amdgcn_s_barrier();
        if (MODE & 8) {
#pragma unroll
            for (int i = 0; i < 7; ++i) glds16(sb, (unsigned)(lane * 16 + i * 1024 + (it & 63) * 8192), sdst + i * 1024);
        }
        const unsigned char* tp0 = lds + (it & 3) * 12288 + lane * 16;
        if (MODE & 1) {
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    b1[u][t] = *reinterpret_cast<const bf16x8*>(tp0 + u * 6144 + (3 * t + 0) * 1024);
                    b2[u][t] = *reinterpret_cast<const bf16x8*>(tp0 + u * 6144 + (3 * t + 1) * 1024);
                    b3[u][t] = *reinterpret_cast<const bf16x8*>(tp0 + u * 6144 + (3 * t + 2) * 1024);
                }
        }
        f32x4 x0[2] = {}, x1[2] = {};
        if (MODE & 16) {
            const unsigned char* wp = lds + 49152 + (it & 3) * 4096 + lane * 16;
            for (int u = 0; u < 2; ++u) {
                x0[u] = *reinterpret_cast<const f32x4*>(wp + u * 2048);
                x1[u] = *reinterpret_cast<const f32x4*>(wp + u * 2048 + 1024);
            }
        }
        if (MODE & 32) {
            // the same 24 MFMAs and the same split, one (two-element) piece of the split pinned behind every third MFMA
            u32x4 na[2], nb[2], nc[2];
            int piece = 0;
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int pr = 0; pr < 6; ++pr)
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        const bf16x8 a = pr == 0 ? a3[u] : (pr == 2 || pr == 3 ? a2[u] : a1[u]);
                        const bf16x8 b = pr == 1 ? b3[u][t] : (pr == 2 || pr == 4 ? b2[u][t] : b1[u][t]);
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[t], 0, 0, 0);
                        if ((u * 12 + pr * 2 + t) % 3 == 2) {
                            __builtin_amdgcn_sched_barrier(0);
                            const int su = piece >> 2, i = piece & 3;
                            const float xa = (MODE & 16) ? (i < 2 ? x0[su][2 * i] : x1[su][2 * i - 4]) : (float)(it + piece);
                            const float xb = (MODE & 16) ? (i < 2 ? x0[su][2 * i + 1] : x1[su][2 * i - 3]) : (float)(it - piece);
                            const float h0 = __uint_as_float(__float_as_uint(xa) & 0xffff0000u), h1 = __uint_as_float(__float_as_uint(xb) & 0xffff0000u);
                            const float r0 = xa - h0, r1 = xb - h1;
                            const float m0 = __uint_as_float(__float_as_uint(r0) & 0xffff0000u), m1 = __uint_as_float(__float_as_uint(r1) & 0xffff0000u);
                            const float s0 = r0 - m0, s1 = r1 - m1;
                            na[su][i] = __builtin_amdgcn_perm(__float_as_uint(xb), __float_as_uint(xa), 0x07060302u);
                            nb[su][i] = __builtin_amdgcn_perm(__float_as_uint(r1), __float_as_uint(r0), 0x07060302u);
                            nc[su][i] = __builtin_amdgcn_perm(__float_as_uint(s1), __float_as_uint(s0), 0x07060302u);
                            ++piece;
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
            for (int u = 0; u < 2; ++u) {
                a1[u] = __builtin_bit_cast(bf16x8, na[u]);
                a2[u] = __builtin_bit_cast(bf16x8, nb[u]);
                a3[u] = __builtin_bit_cast(bf16x8, nc[u]);
            }
        } else {
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int pr = 0; pr < 6; ++pr)
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const bf16x8 a = pr == 0 ? a3[u] : (pr == 2 || pr == 3 ? a2[u] : a1[u]);
                    const bf16x8 b = pr == 1 ? b3[u][t] : (pr == 2 || pr == 4 ? b2[u][t] : b1[u][t]);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[t], 0, 0, 0);
                }
        }
        if ((MODE & 4) && !(MODE & 32)) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                float x[8];
                for (int j = 0; j < 4; ++j) {
                    x[j] = (MODE & 16) ? x0[u][j] : acc[0][j] + (float)it;
                    x[4 + j] = (MODE & 16) ? x1[u][j] : acc[1][j] + (float)it;
                }
                split3(x, a1[u], a2[u], a3[u]);
            }
        }
    }
    const long long c1 = clock64();
    if (MODE & 8) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float s = 0.f;
    for (int t = 0; t < 2; ++t)
        for (int e = 0; e < 16; ++e) s += acc[t][e];
    out[blockIdx.x * 256 + tid] = s;
    if (tid == 0) cyc[blockIdx.x] = c1 - c0;
}

template <int MODE>
void run(const unsigned char* src, float* out, long long* cyc, const char* what) {
    const int iters = 2000, grid = 256;
    hipLaunchKernelGGL(probe<MODE>, dim3(grid), dim3(256), 0, 0, src, out, cyc, iters);
    hipLaunchKernelGGL(probe<MODE>, dim3(grid), dim3(256), 0, 0, src, out, cyc, iters);
    hipDeviceSynchronize();
    std::vector<long long> h(grid);
    hipMemcpy(h.data(), cyc, grid * sizeof(long long), hipMemcpyDeviceToHost);
    double sum = 0;
    for (auto v : h) sum += (double)v;
    printf("mode %2d  %7.0f cycles per stage   %s\n", MODE, sum / grid / iters, what);
    fflush(stdout);
}

int main() {
    unsigned char* src;
    float* out;
    long long* cyc;
    hipMalloc(&src, 64 << 20);
    hipMemset(src, 0, 64 << 20);
    hipMalloc(&out, 256 * 256 * 4);
    hipMalloc(&cyc, 256 * 8);
    run<0>(src, out, cyc, "24 MFMAs, operands in registers");
    run<1>(src, out, cyc, "+ 12 B-operand ds_read_b128");
    run<2>(src, out, cyc, "+ barrier only");
    run<3>(src, out, cyc, "+ reads + barrier");
    run<4>(src, out, cyc, "+ 88-instruction split (inputs from registers)");
    run<5>(src, out, cyc, "+ reads + split");
    run<8>(src, out, cyc, "+ 7 LDS-DMA issues");
    run<10>(src, out, cyc, "+ DMA + barrier");
    run<11>(src, out, cyc, "+ DMA + barrier + reads");
    run<15>(src, out, cyc, "+ DMA + barrier + reads + split");
    run<31>(src, out, cyc, "+ DMA + barrier + reads + split fed by 4 more LDS reads");
    run<36>(src, out, cyc, "split INTERLEAVED by hand (one 11-instruction piece behind every third MFMA), inputs from registers");
    run<63>(src, out, cyc, "everything, split interleaved by hand");
    return 0;
}
