// The hypernetwork's heads at M = B <= 64 as an HBM stream (round 4) — /root/reference/model/hyper_network.py:32-43:
//   theta (B x N) = t5 (B x 2048) . W^T (N x 2048) + b,   N = 19 011 rows for the published target network.
// 156 MB of weights against 5 GFLOP: the tiled fp32 GEMM (gemm.hip, split-K 3 + a reduce launch: 62 + 7 us) spent as long on the
// fp32 matrix pipe (1/16 of the bf16 rate) as on the stream.  Here the matrix work is on the bf16 pipe and small against the
// stream: every fp32 operand is THREE bf16 pieces (truncation splits: x = b1 + b2 + b3 exactly — bf16 keeps fp32's exponent, so
// no scales), a product is six v_mfma_f32_16x16x32_bf16 (b1c1 + b1c2 + b2c1 + b2c2 + b1c3 + b3c1; dropped: <= 2^-23 |xy|), fp32
// accumulation.
//   * a WAVE owns 16 rows of W for the whole contraction: the A fragment of a k-step (32 k) is 128 contiguous bytes of each row,
//     fetched with global_load_lds_dwordx4 (whole 128-byte lines: 8 rows per instruction) into the wave's own LDS slots three
//     pipeline stages (six k-steps) ahead, read back, split on the vector unit, used once.  1 189 row tiles = 5 waves on each of
//     238 workgroups, one round; workgroup g starts the contraction at stage rot(g) so that the rows in flight are not all read at
//     the same offset of their 8 KB pitch.
//   * t5's pieces are laid out once as B fragments (heads_t5_split_kernel: [k-step][cloud tile][piece][lane][16 B], 768 KB) and
//     stream through a 3-slot LDS ring shared by the workgroup's waves (two stages ahead: L2 latency).
//   * a pipeline stage = two k-steps; every wave issues the same ten DMA instructions per stage, so the top of a stage is
//     `s_waitcnt vmcnt(14)` + one barrier (enc_bwd_f16.hip's scheme, with run-time ring slots: the LDS address goes to m0 from an SGPR).
// Measured (B = 64, in the step): 57 + 6 us (split launch) against 63 + 7.5 us for the GEMM + reduce it replaces — NOT yet the stream's
// ~31 us: the waves never wait for memory (vmcnt wait 8 cycles per stage; the same 57 us with every W DMA redirected to one L2-hot
// region), they are serialised inside a stage — per stage and wave (cycle counters): DMA issue 460-750 cycles (ten instructions with
// their m0 set-up), ds_read + split + 48 MFMAs 1 700-2 260 with nothing overlapping at one wave per SIMD, and the four waves that
// have a SIMD to themselves wait ~850 cycles at the barrier for the fifth, which shares one.  More waves per SIMD do NOT help (tried:
// eight waves per workgroup, one k-step per stage: 60 us; four-wave workgroups, two per CU: 62 us — ~0.9 us per k-step and wave
// whatever shares the SIMD: a wave in its MFMA burst keeps the SIMD's issue).  What is left is inside ONE wave's instruction stream:
// the fragments of stage s + 1 read and split in the shadows of stage s's MFMAs (~105 other instructions per 24 MFMAs today).
#include "hp_common.h"
#include <algorithm>
#include <cstdlib>
#include <type_traits>

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int kK = 2048, kSteps = kK / 32;               // MFMA k-steps of 32
constexpr int kWaves = 5, kThreads = kWaves * 64;
constexpr int kT5Step = 12 * 1024;                       // 4 cloud tiles x 3 pieces x 1 KB per k-step
constexpr int kG = 2, kStages = kSteps / kG;             // k-steps per pipeline stage (one barrier per stage), stages
constexpr int kT5Ring = 3, kWRing = 4;                   // stages in flight: t5 two ahead (L2), W three ahead (HBM)
constexpr int kT5Stage = kG * kT5Step;                   // 24 KB
constexpr int kWStage = kWaves * kG * 2048;              // a wave's raw fp32 fragments: two 1 KB halves per k-step
constexpr int kLds = kT5Ring * kT5Stage + kWRing * kWStage;      // 72 + 80 = 152 KB
constexpr long kT5Bytes = (long)kSteps * kT5Step;        // 768 KB

// x = b1 + b2 + b3 exactly (truncation): 8 values -> three fragments of 8 bf16
__device__ __forceinline__ void split3(const float (&x)[8], bf16x8& p1, bf16x8& p2, bf16x8& p3) {
    u32x4 a, b, c;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float x0 = x[2 * i], x1 = x[2 * i + 1];
        const float h0 = __uint_as_float(__float_as_uint(x0) & 0xffff0000u), h1 = __uint_as_float(__float_as_uint(x1) & 0xffff0000u);
        const float r0 = x0 - h0, r1 = x1 - h1;
        const float m0 = __uint_as_float(__float_as_uint(r0) & 0xffff0000u), m1 = __uint_as_float(__float_as_uint(r1) & 0xffff0000u);
        const float s0 = r0 - m0, s1 = r1 - m1;
        a[i] = __builtin_amdgcn_perm(__float_as_uint(x1), __float_as_uint(x0), 0x07060302u);      // (hi16(x1) << 16) | hi16(x0)
        b[i] = __builtin_amdgcn_perm(__float_as_uint(r1), __float_as_uint(r0), 0x07060302u);
        c[i] = __builtin_amdgcn_perm(__float_as_uint(s1), __float_as_uint(s0), 0x07060302u);
    }
    p1 = __builtin_bit_cast(bf16x8, a);
    p2 = __builtin_bit_cast(bf16x8, b);
    p3 = __builtin_bit_cast(bf16x8, c);
}

// t5 (B x 2048) -> B fragments of v_mfma_f32_16x16x32_bf16: block (k-step s, cloud tile ct): lane (c, kg) holds
// t5[16 ct + c][32 s + 8 kg + j], j = 0..7, as three pieces; clouds >= B are zeros
__global__ __launch_bounds__(64) void heads_t5_split_kernel(int B, const float* __restrict__ t5, unsigned char* __restrict__ out) {
    const int s = blockIdx.x >> 2, ct = blockIdx.x & 3, lane = threadIdx.x, c = 16 * ct + (lane & 15), kg = lane >> 4;
    float x[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) x[j] = 0.f;
    if (c < B) {
        const f32x4 v0 = *reinterpret_cast<const f32x4*>(t5 + (long)c * kK + 32 * s + 8 * kg);
        const f32x4 v1 = *reinterpret_cast<const f32x4*>(t5 + (long)c * kK + 32 * s + 8 * kg + 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            x[j] = v0[j];
            x[4 + j] = v1[j];
        }
    }
    bf16x8 p[3];
    split3(x, p[0], p[1], p[2]);
    unsigned char* dst = out + (long)s * kT5Step + (ct * 3) * 1024 + lane * 16;
#pragma unroll
    for (int q = 0; q < 3; ++q) *reinterpret_cast<bf16x8*>(dst + q * 1024) = p[q];
}

// LDS-DMA, 16 bytes per lane: global = uniform base + 32-bit lane offset, LDS = lds_dst (wave-uniform, an SGPR) + 16 * lane
__device__ __forceinline__ void glds16(const void* sbase, unsigned voff, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(sbase), "s"(lds_dst)
                 : "memory");
}

// LDS: [t5 ring: 3 x 24 KB][W ring: 4 x (5 waves x 4 KB)].  A pipeline stage = two k-steps: per stage a wave issues six t5 DMAs (its
// share of the 24 fragments; the fifth wave repeats the first's) and four W DMAs, so at the top of stage s — which needs t5(s),
// issued two stages ago in front of that stage's four W DMAs — all but the youngest 4 + 10 operations must have landed.
__global__ __launch_bounds__(kThreads, 1) void heads_fwd_kernel(int B, int N, const float* __restrict__ W,
                                                                 const float* __restrict__ bias, const unsigned char* __restrict__ t5p,
                                                                 float* __restrict__ theta, int theta_ld) {
    __shared__ __attribute__((aligned(1024))) unsigned char lds[kLds];
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ntiles = (N + 15) >> 4;
    const int tile = blockIdx.x + gridDim.x * w;
    const bool live = tile < ntiles;                                       // (wave-uniform)
    // W DMA map: instruction j of a k-step fetches rows 8 j + (lane >> 3), 16-byte chunk lane & 7 of the k-step's 128 bytes — whole
    // 128-byte lines per row (a lane-per-row map asks for every line twice, 64 bytes at a time); rows past N re-read row N - 1
    const int tile0 = (live ? tile : 0) * 16;
    const int row_lo = min(tile0 + (lane >> 3), N - 1), row_hi = min(tile0 + 8 + (lane >> 3), N - 1);
    auto uniform_ptr = [](const void* q) {
        const unsigned long long p = reinterpret_cast<unsigned long long>(q);
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)p), hi = __builtin_amdgcn_readfirstlane((unsigned)(p >> 32));
        return reinterpret_cast<const void*>(((unsigned long long)hi << 32) | lo);
    };
    const void* wbase = uniform_ptr(W);
    const void* tbase = uniform_ptr(t5p);
    const unsigned lds0 = (unsigned)(size_t)lds;
    const int tw = w < 4 ? w : 0;                                          // t5 share: fragments 6 tw .. 6 tw + 5 of a stage's 24
    const unsigned s_t5 = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)(6 * tw * 1024));
    const unsigned s_w = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)(kT5Ring * kT5Stage + w * kG * 2048));
    // Workgroup g walks the contraction from stage rot(g) on, cyclically: with every workgroup at the same k all ~19 000 rows in
    // flight are read at the SAME offset of their 8 KB pitch (the same few HBM channels).
    const unsigned rot = (blockIdx.x * 5u) & (kStages - 1);
    const unsigned wrow0 = (unsigned)row_lo * (unsigned)(kK * 4) + (unsigned)(lane & 7) * 16u;
    const unsigned wrow1 = (unsigned)row_hi * (unsigned)(kK * 4) + (unsigned)(lane & 7) * 16u;
    const unsigned trow = (unsigned)(6 * tw * 1024 + lane * 16);
    int nis = 0;      // issues so far.  Issue k carries W of stage k and t5 of stage k - 1 (the 3-slot ring holds t5 two stages ahead);
                      // past the end the last stage is issued again (into ring slots nobody reads any more)
    auto issue = [&]() {
        const int ws = min(nis, kStages - 1), ts = min(max(nis - 1, 0), kStages - 1);
        const unsigned to = trow + (((unsigned)ts + rot) & (kStages - 1)) * (unsigned)kT5Stage;
        const unsigned wk = (((unsigned)ws + rot) & (kStages - 1)) * (unsigned)(kG * 128);
        const unsigned lt = s_t5 + (unsigned)(max(nis - 1, 0) % kT5Ring) * (unsigned)kT5Stage, lw = s_w + (unsigned)(nis % kWRing) * (unsigned)kWStage;
#pragma unroll
        for (int i = 0; i < 6; ++i) glds16(tbase, to + i * 1024, lt + i * 1024);
#pragma unroll
        for (int g = 0; g < kG; ++g) {
            glds16(wbase, wrow0 + wk + g * 128, lw + g * 2048);
            glds16(wbase, wrow1 + wk + g * 128, lw + g * 2048 + 1024);
        }
        ++nis;
    };
    issue();      // W 0, t5 0
    issue();      // W 1, t5 0 again
    issue();      // W 2, t5 1
    f32x4 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
    for (int sidx = 0; sidx < kStages; ++sidx) {
        asm volatile("s_waitcnt vmcnt(14)" ::: "memory");      // t5(sidx) sits in issue sidx + 1, in front of its 4 W DMAs and issue sidx + 2
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        issue();      // W(sidx + 3), t5(sidx + 2): into the ring slots stage sidx - 1 just left
        if (live) {
            // lane (r = lane & 15, kg = lane >> 4) of the A fragment: bytes [32 kg, +32) of row r = slots 8 (r & 7) + 2 kg, + 1 of half r >> 3
            const unsigned char* wp = lds + kT5Ring * kT5Stage + (sidx % kWRing) * kWStage + w * kG * 2048 +
                                      ((lane & 15) >> 3) * 1024 + (((lane & 7) * 8) + 2 * (lane >> 4)) * 16;
            const unsigned char* tp0 = lds + (sidx % kT5Ring) * kT5Stage + lane * 16;
#pragma unroll
            for (int g = 0; g < kG; ++g) {
                const f32x4 x0 = *reinterpret_cast<const f32x4*>(wp + g * 2048), x1 = *reinterpret_cast<const f32x4*>(wp + g * 2048 + 16);
                const float x[8] = {x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3]};
                bf16x8 a1, a2, a3;
                split3(x, a1, a2, a3);
                const unsigned char* tp = tp0 + g * kT5Step;
                bf16x8 b1[4], b2[4], b3[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    b1[t] = *reinterpret_cast<const bf16x8*>(tp + (3 * t + 0) * 1024);
                    b2[t] = *reinterpret_cast<const bf16x8*>(tp + (3 * t + 1) * 1024);
                    b3[t] = *reinterpret_cast<const bf16x8*>(tp + (3 * t + 2) * 1024);
                }
                // product-major: consecutive MFMAs go to different accumulators; small terms first
#pragma unroll
                for (int pr = 0; pr < 6; ++pr)
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const bf16x8 a = pr == 0 ? a3 : (pr == 2 || pr == 3 ? a2 : a1);
                        const bf16x8 b = pr == 1 ? b3[t] : (pr == 2 || pr == 4 ? b2[t] : b1[t]);
                        acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[t], 0, 0, 0);
                    }
            }
        }
    }
    // C/D of 16x16: column = lane & 15 (cloud), rows 4 (lane >> 4) + e
    if (live) {
        const int c = lane & 15, n0 = tile * 16 + 4 * (lane >> 4);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int cloud = 16 * t + c;
            if (cloud < B) {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (n0 + e < N) theta[(long)cloud * theta_ld + n0 + e] = acc[t][e] + (bias ? bias[n0 + e] : 0.f);
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the re-issued tail DMAs land before the LDS is released
}

int g_heads_fwd = -1;

}  // namespace

bool hp_heads_fwd_enabled() {
    static const bool env_on = [] {
        const char* e = std::getenv("HP_HEADS_FWD");
        return !(e && e[0] == '0');
    }();
    return g_heads_fwd < 0 ? env_on : g_heads_fwd != 0;
}
int hp_heads_fwd_set(int on) {
    const int prev = g_heads_fwd;
    g_heads_fwd = on < 0 ? -1 : (on != 0);
    return prev;
}
long hp_heads_fwd_ws_floats() { return kT5Bytes / 4; }
bool hp_heads_fwd_ok(int B, int N, int K, const float* t5, const float* W, const float* ws) {
    auto a16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    return B >= 1 && B <= 64 && K == kK && N >= 16 && (long)N * K * 4 < (1L << 32) && a16(t5) && a16(W) && a16(ws);
}
// theta (B x N, ld theta_ld) = t5 (B x 2048) . W (N x 2048)^T + bias; ws: hp_heads_fwd_ws_floats() floats
int hp_heads_fwd(int B, int N, const float* t5, const float* W, const float* bias, float* theta, int theta_ld, float* ws,
                 hipStream_t stream) {
    unsigned char* t5p = reinterpret_cast<unsigned char*>(ws);
    hipLaunchKernelGGL(heads_t5_split_kernel, dim3(kSteps * 4), dim3(64), 0, stream, B, t5, t5p);
    const int ntiles = (N + 15) / 16;
    const int grid = (ntiles + kWaves - 1) / kWaves;      // 238 workgroups for the published network: one per CU, one round
    hipLaunchKernelGGL(heads_fwd_kernel, dim3(grid), dim3(kThreads), 0, stream, B, N, W, bias, t5p, theta, theta_ld);
    HP_RETURN_LAST_ERROR();
}
