// The delta chain of the encoders' conv backward on the f16 matrix pipe (round 4) — the autograd of
//   /root/reference/model/encoder.py:14-28 (Conv1d(k=1)+ReLU x4) below the max-pool, layers 4 -> 1:
//   delta_{l-1} = (delta_l W_l) * (h_{l-1} > 0)     for the critical rows enc_bwd.hip's gather launch left in delta4.
// Same arithmetic as the forward (conv_split.hip / conv_pp.hip): an fp32 operand is two f16 pieces under a power-of-two scale
// (hi = f16(x 2^e), lo = f16(x 2^e - hi): 22 bits), a product is three v_mfma_f32_32x32x16_f16 (hi.hi + hi.lo + lo.hi), fp32
// accumulation; dropped: lo.lo <= 2^-22 |ab| and the pieces' rounding <= 2^-23 — the error of an fp32 fma chain of the same
// length.  Against enc_bwd_chain_kernel (fp32 MFMA at 1/16 of the f16 rate, every 32-row workgroup streaming the 672 KB of
// weights through its registers: 145 us for a HyperPocket step) this kernel is built around three facts:
//  * rows are independent.  A WAVE owns a 32-row block of one cloud and ALL output channels of a layer, with the operands
//    swapped — A = W_l^T (rows = the layer's input channels), B = delta (columns = the 32 rows): a lane then holds ONE row
//    (column lane & 31) and channels in its registers, so the row's maximum (the next layer's scale) is an in-lane maximum plus
//    one exchange with the partner lane, the ReLU mask / unscale / fp32 store of delta_{l-1} are per-lane work, and the
//    finished accumulators ARE the next layer's B fragments (the contraction order is permuted to the C/D register map:
//    k-step s, lane half h holds channels 16 s + 4 h + {0..3} and 16 s + 8 + 4 h + {0..3}).  No cross-wave exchange, no LDS
//    round trip for delta3 / delta2.
//  * the four waves of a workgroup (128 rows) share the weight stream: enc_bwd_wprep_kernel lays W_4, W_3, W_2 out ONCE per
//    step as A fragments in consumption order (transposed, column-scaled, split, [k-step][channel tile][hi|lo][lane][16 B]):
//    672 KB that every workgroup reads front to back as 42 chunks of 16 KB with global_load_lds_dwordx4 into a 4-stage LDS
//    ring — and each wave's delta4 operand (fp32 rows from the gather launch) rides in the same chunks, 2 KB per k-step in
//    fragment order, split into pieces on the way out of LDS with the row's scale (the gather launch leaves the row maxima).
//  * every chunk issues the same six DMA instructions per wave (past the end: re-reads nobody consumes), so the top of a chunk
//    is `s_waitcnt vmcnt(12)` + ONE barrier: all but the two youngest chunks have landed, whatever else the compiler has in
//    flight only makes the wait stricter.
#include "hp_common.h"
#include "hp_enc_bwd.h"
#include "hp_enc_bwd_wprep.h"
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include <vector>

namespace {

using namespace hp_wprep;       // stream layout (kChunk, kC4 .., kUs4 ..), scale_exp / pow2f / split8
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kStage = kChunk + 4 * 2048;     // a chunk + the four waves' delta4 slices of one k-step
constexpr int kStages = 4;

// LDS-DMA, 16 bytes per lane: global address = wave-uniform base + 32-bit lane offset, LDS destination = SGPR base + literal
// + 16 * lane (conv_pp.hip: the asm form keeps the compiler from draining the queue at the next ds_read)
template <int LIT>
__device__ __forceinline__ void glds16s(const void* sbase, unsigned voff, unsigned lds_base) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_add_u32 m0, %3, %4\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(sbase), "s"(lds_base), "n"(LIT)
                 : "memory", "scc");
}

// 64-lane inclusive scan
__device__ __forceinline__ int wave_scan(int v, int ln) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(v, o, 64);
        if (ln >= o) v += t;
    }
    return v;
}

// ---------------------------------------------------------------------------------------------------------------------
// chain: delta3, delta2, delta1 of four 32-row blocks
// ---------------------------------------------------------------------------------------------------------------------
struct ChainCtx {
    unsigned char* lds;
    const void* wt;        // weight stream (wave-uniform)
    const void* d4;        // the wave's delta4 block (wave-uniform)
    unsigned wo[4];        // lane offsets of the wave's four KB of the NEXT chunk to issue
    unsigned dof[2];       // lane offsets of the wave's two delta4 loads of the next chunk
    unsigned s_ldsw, s_ldsd;
    int w, lane;
    bool live;
};

// the top of chunk: everything but the two youngest chunks has landed (six DMA instructions per wave and chunk), for all waves
__device__ __forceinline__ void chunk_top() {
    asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}
template <int S>   // into stage S
__device__ __forceinline__ void chunk_issue(ChainCtx& c, bool adv_w, bool adv_d) {
    glds16s<S * kStage + 0>(c.wt, c.wo[0], c.s_ldsw);
    glds16s<S * kStage + 1024>(c.wt, c.wo[1], c.s_ldsw);
    glds16s<S * kStage + 2048>(c.wt, c.wo[2], c.s_ldsw);
    glds16s<S * kStage + 3072>(c.wt, c.wo[3], c.s_ldsw);
    glds16s<S * kStage + kChunk>(c.d4, c.dof[0], c.s_ldsd);
    glds16s<S * kStage + kChunk + 1024>(c.d4, c.dof[1], c.s_ldsd);
    if (adv_w) {
#pragma unroll
        for (int i = 0; i < 4; ++i) c.wo[i] += kChunk;
    }
    if (adv_d) {
        c.dof[0] += 64;
        c.dof[1] += 64;
    }
}

// the A fragments of G k-steps x T tiles of a stage (k-steps SL0 .. SL0 + G - 1 of its chunk)
template <int T, int G>
struct Frags {
    f16x8 ah[G][T], al[G][T];
};
template <int S, int T, int SL0, int G>
__device__ __forceinline__ void read_frags(const ChainCtx& c, Frags<T, G>& f) {
    const unsigned char* st = c.lds + S * kStage + SL0 * T * 2048 + c.lane * 16;
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
        for (int t = 0; t < T; ++t) {
            f.ah[g][t] = *reinterpret_cast<const f16x8*>(st + (g * T + t) * 2048);
            f.al[g][t] = *reinterpret_cast<const f16x8*>(st + (g * T + t) * 2048 + 1024);
        }
}
// acc[t] += A(k-step g of the set, tile t) x B: product-major, consecutive MFMAs go to different accumulators
template <int T, int G>
__device__ __forceinline__ void mma_step(const Frags<T, G>& f, int g, const f16x8& bh, const f16x8& bl, f32x16 (&acc)[T]) {
#pragma unroll
    for (int pr = 0; pr < 3; ++pr)
#pragma unroll
        for (int t = 0; t < T; ++t)
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(pr == 2 ? f.al[g][t] : f.ah[g][t], pr == 1 ? bl : bh, acc[t], 0, 0, 0);
}

// Epilogue of a layer with N = 32 T output channels: v = acc 2^-e_n 2^-e_row where the row's activation was positive (bit
// drow(e, h) of mk[t], the gather launch's mask words), else 0; stored as fp32 rows (16 bytes per lane and register group);
// returns the row's maximum |v| (both lane halves), v left in acc.
template <int T>
__device__ __forceinline__ float chain_epilogue(f32x16 (&acc)[T], const float* ust, float usr, const unsigned (&mk)[T],
                                                float* __restrict__ dout, long row, int h) {
    constexpr int N = 32 * T;
    float vmax = 0.f;
#pragma unroll
    for (int t = 0; t < T; ++t) {
        const unsigned mh = mk[t] >> (4 * h);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 us = *reinterpret_cast<const f32x4*>(ust + 32 * t + 8 * g + 4 * h);
            f32x4 v;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float x = acc[t][4 * g + i] * us[i] * usr;
                const unsigned keep = (unsigned)-(int)((mh >> (8 * g + i)) & 1u);      // v_bfe_i32 + v_and
                v[i] = __uint_as_float(__float_as_uint(x) & keep);
                acc[t][4 * g + i] = v[i];
                vmax = fmaxf(vmax, fabsf(v[i]));
            }
            *reinterpret_cast<f32x4*>(dout + row * N + 32 * t + 8 * g + 4 * h) = v;
        }
    }
    return fmaxf(vmax, __shfl_xor(vmax, 32, 64));
}

// the accumulators of a finished layer (T tiles) as the next layer's 2 T B-fragment pairs under the row scale sc
template <int T>
__device__ __forceinline__ void acc_to_frags(const f32x16 (&acc)[T], float sc, f16x8 (&bh)[2 * T], f16x8 (&bl)[2 * T]) {
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            float y[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) y[j] = acc[t][8 * q + j];
            split8s(y, sc, bh[2 * t + q], bl[2 * t + q]);
        }
}

template <int V>
using IC = std::integral_constant<int, V>;

__global__ __launch_bounds__(256, 1) void enc_bwd_chain_f16_kernel(const HpEncBwdArgs a) {
    __shared__ __attribute__((aligned(1024))) unsigned char lds[kStages * kStage + HP_EB_WT_US_FLOATS * 4];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, r = lane & 31, h = lane >> 5;
    // workgroup -> (encoder z, group g of four live 32-row blocks): the live groups of both encoders are the first ids
    int z = -1, g = 0, nbz = 0;
    {
        int rel = blockIdx.x;
        for (int e = 0; e < a.n && z < 0; ++e) {
            const int* cn = a.e[e].crit.cnt;
            int tot = 0;
            for (int c0 = 0; c0 < a.B; c0 += 64) {
                const int nb = c0 + lane < a.B ? (cn[c0 + lane] + 31) >> 5 : 0;
                tot += __shfl(wave_scan(nb, lane), 63, 64);
            }
            const int groups = (tot + 3) >> 2;
            if (rel < groups) {
                z = e;
                g = rel;
                nbz = tot;
            } else {
                rel -= groups;
            }
        }
        if (z < 0) return;
    }
    const HpEncBwdSide& s = a.e[z];
    long long* prof = a.prof ? a.prof + (long)blockIdx.x * 10 : nullptr;      // HP_EB_PROF16: phase stamps of wave 0 (debug)
#define HP_STAMP(k) do { if (prof && tid == 0) prof[k] = (long long)wall_clock64(); } while (0)
    HP_STAMP(0);
    // wave -> block j of the encoder = (cloud b, block q)
    const int j = 4 * g + w;
    const bool live = __builtin_amdgcn_readfirstlane((int)(j < nbz)) != 0;      // (wave-uniform, and the compiler knows it)
    int b = 0, q = 0, cnt = 0;
    if (live) {
        int rel = j;
        bool found = false;
        for (int c0 = 0; c0 < a.B && !found; c0 += 64) {
            const int cv = c0 + lane < a.B ? s.crit.cnt[c0 + lane] : 0;
            const int nb = (cv + 31) >> 5;
            const int inc = wave_scan(nb, lane);
            const int tot = __shfl(inc, 63, 64);
            if (rel < tot) {
                const unsigned long long mk = __ballot(inc > rel);
                const int l = __ffsll((long long)mk) - 1;
                b = c0 + l;
                q = rel - (__shfl(inc, l, 64) - __shfl(nb, l, 64));
                cnt = __shfl(cv, l, 64);
                found = true;
            } else {
                rel -= tot;
            }
        }
    }
    const long row0 = (long)b * 512 + q * 32;         // first row of the block in the delta / hc arrays
    const long row = row0 + r;

    ChainCtx c;
    c.lds = lds;
    c.w = w;
    c.lane = lane;
    c.live = live;
    auto uniform_ptr = [](const void* q) {      // the DMA's base operand is an SGPR pair
        const unsigned long long p = reinterpret_cast<unsigned long long>(q);
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)p), hi = __builtin_amdgcn_readfirstlane((unsigned)(p >> 32));
        return reinterpret_cast<const void*>(((unsigned long long)hi << 32) | lo);
    };
    c.wt = uniform_ptr(s.wt);
    c.d4 = uniform_ptr(s.d[4] + row0 * 512);
    const unsigned lds0 = (unsigned)(size_t)lds;
    c.s_ldsw = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)(w * 4096));
    c.s_ldsd = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)(w * 2048));
#pragma unroll
    for (int i = 0; i < 4; ++i) c.wo[i] = (unsigned)((4 * w + i) * 1024 + lane * 16);
    c.dof[0] = (unsigned)(r * 2048 + 16 * h);
    c.dof[1] = c.dof[0] + 32;
    const unsigned dof0 = c.dof[0];

    // pipeline prologue: chunks 0, 1, 2 — the DMAs first, the small loads behind them
    chunk_issue<0>(c, true, true);
    chunk_issue<1>(c, true, true);
    chunk_issue<2>(c, true, true);

    float* ust = reinterpret_cast<float*>(lds + kStages * kStage);
    for (int i = tid; i < HP_EB_WT_US_FLOATS; i += 256) ust[i] = s.wt_us[i];
    if (live) {      // the rows' coordinates (dW1's operand)
        const int u0 = q * 32;
        for (int i = lane; i < 96; i += 64) {
            const int xr = i / 3, xc = i - xr * 3;
            s.hc[0][(row0 + xr) * 3 + xc] =
                u0 + xr < cnt ? s.x[((long)b * a.Np + s.crit.pt[(long)b * 512 + u0 + xr]) * 3 + xc] : 0.f;
        }
    }
    // the row's scale and its ReLU masks (gather launch): words 0..7 h3, 8..11 h2, 12..13 h1
    const float m4 = live ? s.d4max[row] : 0.f;
    const int e4 = scale_exp(m4);
    const float sc4 = pow2f(e4);
    unsigned mk3[8], mk2[4], mk1[2];
    {
        const u32x4* mp = reinterpret_cast<const u32x4*>(s.hmask + (live ? row : 0) * 64);
        const u32x4 m0 = mp[0], m1 = mp[1], m2 = mp[2], m3 = mp[3];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            mk3[i] = m0[i];
            mk3[4 + i] = m1[i];
            mk2[i] = m2[i];
        }
        mk1[0] = m3[0];
        mk1[1] = m3[1];
    }

    HP_STAMP(1);
    // ---- layer 4: delta3 = (delta4 W4) * (h3 > 0)     K = 512: 32 chunks of one k-step, N = 256: 8 tiles.  Software pipeline:
    //      the fragments of k-step c leave LDS (ds_read) in front of the MFMAs of k-step c - 1 (two register sets)
    f32x16 acc4[8];
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc4[t][e] = 0.f;
    Frags<8, 1> fa[2];
    f32x4 xr0[2], xr1[2];      // the raw delta4 values of a k-step (fragment order)
    auto mma4 = [&](auto PC) {
        constexpr int P = decltype(PC)::value;
        float y[8];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            y[i] = xr0[P][i];
            y[4 + i] = xr1[P][i];
        }
        f16x8 bh, bl;
        split8s(y, sc4, bh, bl);
        mma_step<8, 1>(fa[P], 0, bh, bl, acc4);
    };
    auto step4 = [&](auto SC, bool first) {
        constexpr int S = decltype(SC)::value;
        chunk_top();
        chunk_issue<(S + 3) & 3>(c, true, true);      // (chunks 32.. carry stale delta4 bytes nobody reads)
        if (live) {
            const unsigned char* dp = lds + S * kStage + kChunk + w * 2048 + lane * 16;
            xr0[S & 1] = *reinterpret_cast<const f32x4*>(dp);
            xr1[S & 1] = *reinterpret_cast<const f32x4*>(dp + 1024);
            read_frags<S, 8, 0, 1>(c, fa[S & 1]);
            __builtin_amdgcn_sched_barrier(0);      // (hipcc sinks the reads behind most of the MFMAs they are meant to run under)
            if (!first) mma4(IC<(S & 1) ^ 1>{});
            __builtin_amdgcn_sched_barrier(0);
        }
    };
#pragma unroll 1
    for (int it = 0; it < kC4 / 4; ++it) {
        step4(IC<0>{}, it == 0);
        step4(IC<1>{}, false);
        step4(IC<2>{}, false);
        step4(IC<3>{}, false);
    }
    HP_STAMP(2);
    f16x8 b3h[16], b3l[16];
    float us3r = 1.f;
    if (live) {
        mma4(IC<1>{});      // k-step 31
        const float m3 = chain_epilogue<8>(acc4, ust + kUs4, pow2f(-e4), mk3, s.d[3], row, h);
        const int e3 = scale_exp(m3);
        us3r = pow2f(-e3);
        acc_to_frags<8>(acc4, pow2f(e3), b3h, b3l);
    }

    HP_STAMP(3);
    // ---- layer 3: delta2 = (delta3 W3) * (h2 > 0)     K = 256: 8 chunks of two k-steps, N = 128: 4 tiles
    c.dof[0] = dof0;
    c.dof[1] = dof0 + 32;
    f32x16 acc3[4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc3[t][e] = 0.f;
    Frags<4, 2> fb[2];
    auto step3 = [&](auto CI) {
        constexpr int ci = decltype(CI)::value, S = ci & 3;
        chunk_top();
        // chunks 35..41 still hold weights; past the end the stream's first chunk is read again
        constexpr bool more = kC4 + ci + 3 < kC4 + kC3 + kC2;
        if (!more) {
#pragma unroll
            for (int i = 0; i < 4; ++i) c.wo[i] = (unsigned)((4 * w + i) * 1024 + lane * 16);
        }
        chunk_issue<(S + 3) & 3>(c, more, false);
        if (live) {
            read_frags<S, 4, 0, 2>(c, fb[ci & 1]);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (ci > 0) {
                mma_step<4, 2>(fb[(ci & 1) ^ 1], 0, b3h[2 * ci - 2], b3l[2 * ci - 2], acc3);
                mma_step<4, 2>(fb[(ci & 1) ^ 1], 1, b3h[2 * ci - 1], b3l[2 * ci - 1], acc3);
            }
        }
    };
    step3(IC<0>{}); step3(IC<1>{}); step3(IC<2>{}); step3(IC<3>{});
    step3(IC<4>{}); step3(IC<5>{}); step3(IC<6>{}); step3(IC<7>{});
    HP_STAMP(4);
    f16x8 b2h[8], b2l[8];
    float us2r = 1.f;
    if (live) {
        mma_step<4, 2>(fb[1], 0, b3h[14], b3l[14], acc3);
        mma_step<4, 2>(fb[1], 1, b3h[15], b3l[15], acc3);
        const float m2 = chain_epilogue<4>(acc3, ust + kUs3, us3r, mk2, s.d[2], row, h);
        const int e2 = scale_exp(m2);
        us2r = pow2f(-e2);
        acc_to_frags<4>(acc3, pow2f(e2), b2h, b2l);
    }

    HP_STAMP(5);
    // ---- layer 2: delta1 = (delta2 W2) * (h1 > 0)     K = 128: 2 chunks of four k-steps, N = 64: 2 tiles
    f32x16 acc2[2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc2[t][e] = 0.f;
    {
        Frags<2, 4> fc[2];
        chunk_top();
        chunk_issue<3>(c, false, false);
        if (live) read_frags<0, 2, 0, 4>(c, fc[0]);
        chunk_top();
        chunk_issue<0>(c, false, false);
        if (live) {
            read_frags<1, 2, 0, 4>(c, fc[1]);
#pragma unroll
            for (int g2 = 0; g2 < 4; ++g2) mma_step<2, 4>(fc[0], g2, b2h[g2], b2l[g2], acc2);
#pragma unroll
            for (int g2 = 0; g2 < 4; ++g2) mma_step<2, 4>(fc[1], g2, b2h[4 + g2], b2l[4 + g2], acc2);
        }
    }
    HP_STAMP(6);
    if (live) (void)chain_epilogue<2>(acc2, ust + kUs2, us2r, mk1, s.d[1], row, h);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the re-read chunks land before the workgroup's LDS is released
    HP_STAMP(7);
    if (prof && tid == 0) prof[8] = 1;
#undef HP_STAMP
}

int g_chain16 = -1;

}  // namespace

bool hp_enc_bwd_chain_f16_enabled() {
    static const bool env_on = [] {
        const char* e = std::getenv("HP_EB_CHAIN16");
        return !(e && e[0] == '0');
    }();
    return g_chain16 < 0 ? env_on : g_chain16 != 0;
}
int hp_enc_bwd_chain_f16_set(int on) {
    const int prev = g_chain16;
    g_chain16 = on < 0 ? -1 : (on != 0);
    return prev;
}

// the chain (the weight stream was written by the prep launch): B*4 workgroups per encoder, the live ones first
int hp_enc_bwd_chain_f16(const HpEncBwdArgs* a0, hipStream_t stream) {
    static const bool prof_on = std::getenv("HP_EB_PROF16") != nullptr;
    if (prof_on) {      // debug: per-phase averages of the live workgroups (synchronises)
        static long long* buf = nullptr;
        const long nwg = (long)a0->B * 4 * a0->n;
        if (!buf) (void)hipMalloc(&buf, sizeof(long long) * 10 * 65536);
        if (nwg <= 65536) {
            HpEncBwdArgs a = *a0;
            (void)hipMemsetAsync(buf, 0, sizeof(long long) * 10 * nwg, stream);
            a.prof = buf;
            hipLaunchKernelGGL(enc_bwd_chain_f16_kernel, dim3((unsigned)nwg), dim3(256), 0, stream, a);
            (void)hipStreamSynchronize(stream);
            std::vector<long long> hb(10 * nwg);
            (void)hipMemcpy(hb.data(), buf, sizeof(long long) * 10 * nwg, hipMemcpyDeviceToHost);
            double sum[7] = {0}, mx[7] = {0};
            long live = 0;
            long long tmin = -1, tmax = 0;
            for (long i = 0; i < nwg; ++i) {
                const long long* t = &hb[i * 10];
                if (!t[8]) continue;
                ++live;
                if (tmin < 0 || t[0] < tmin) tmin = t[0];
                tmax = std::max(tmax, t[7]);
                for (int k = 0; k < 7; ++k) {
                    const double d = (double)(t[k + 1] - t[k]) * 0.01;      // 100 MHz -> us
                    sum[k] += d;
                    mx[k] = std::max(mx[k], d);
                }
            }
            fprintf(stderr, "[chain16 prof] live %ld span %.1f us | avg(max) us: prologue %.1f(%.1f) L4 %.1f(%.1f) epi4 %.1f(%.1f) L3 %.1f(%.1f) "
                    "epi3 %.1f(%.1f) L2 %.1f(%.1f) epi2+drain %.1f(%.1f)\n", live, (double)(tmax - tmin) * 0.01, sum[0] / live, mx[0],
                    sum[1] / live, mx[1], sum[2] / live, mx[2], sum[3] / live, mx[3], sum[4] / live, mx[4], sum[5] / live, mx[5],
                    sum[6] / live, mx[6]);
            HP_RETURN_LAST_ERROR();
        }
    }
    const HpEncBwdArgs* a = a0;
    hipLaunchKernelGGL(enc_bwd_chain_f16_kernel, dim3((unsigned)(a->B * 4 * a->n)), dim3(256), 0, stream, *a);
    HP_RETURN_LAST_ERROR();
}
