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

#define HP_SB() __builtin_amdgcn_sched_barrier(0)
using namespace hp_wprep;       // stream layout (kChunk, kC4 .., kUs4 ..), scale_exp / pow2f / split8
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int drow(int e, int h) { return (e & 3) + 8 * (e >> 2) + 4 * h; }   // C/D row of register e, lane half h
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
    __shared__ float bmx[4][8];
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
    float us3r = 1.f, m3keep = 0.f, m2keep = 0.f;
    if (live) {
        mma4(IC<1>{});      // k-step 31
        const float m3 = chain_epilogue<8>(acc4, ust + kUs4, pow2f(-e4), mk3, s.d[3], row, h);
        m3keep = m3;
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
        m2keep = m2;
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
    // the exponents of the workgroup's four blocks (128 rows in the dW launch's walk order) for the dW launch, whose contraction
    // runs over rows: one scale per operand and 128 rows — the forward's block size — so that it rescales its accumulators
    // once per four blocks
    float bm[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (live) {
        const float m1 = chain_epilogue<2>(acc2, ust + kUs2, us2r, mk1, s.d[1], row, h);
        bm[0] = m4; bm[1] = m3keep; bm[2] = m2keep; bm[3] = m1;
        bm[4] = s.hmax[row * 4 + 0]; bm[5] = s.hmax[row * 4 + 1]; bm[6] = s.hmax[row * 4 + 2];
#pragma unroll
        for (int k = 0; k < 7; ++k) {
#pragma unroll
            for (int o = 16; o > 0; o >>= 1) bm[k] = fmaxf(bm[k], __shfl_xor(bm[k], o, 64));
        }
    }
    if (lane < 7) {
        float v = bm[0];
#pragma unroll
        for (int k = 1; k < 7; ++k) v = lane == k ? bm[k] : v;
        bmx[w][lane] = v;
    }
    __syncthreads();
    if (live && lane < 7) s.bexp[(row0 >> 5) * 8 + lane] = scale_exp(fmaxf(fmaxf(bmx[0][lane], bmx[1][lane]), fmaxf(bmx[2][lane], bmx[3][lane])));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the re-read chunks land before the workgroup's LDS is released
    HP_STAMP(7);
    if (prof && tid == 0) prof[8] = 1;
#undef HP_STAMP
}

// ---------------------------------------------------------------------------------------------------------------------
// dW on the f16 pipe: dW_l = delta_l^T hc_{l-1}, db_l = column sums of delta_l, l = 4..2 (dW1: three columns, vector unit)
// ---------------------------------------------------------------------------------------------------------------------
// The contraction runs over the ROWS, so a scale must be constant along rows: one exponent per 32-row block and operand (the
// chain launch leaves them in bexp[block][8]: delta4..delta1, h3..h1), the accumulators are rescaled by the exact 2^(E' - E)
// at block boundaries (growth capped at 2^40 per block: a block that small adds nothing the fp32 sum could hold anyway).
// A workgroup (4 waves) owns a 128 x NT tile of one dW over a range of row blocks, 16 rows per chunk: fp32 rows global ->
// registers (four chunks in flight) -> split -> hi / lo f16 images in LDS, row-major [row][channel] with the 64-byte windows of a
// 256-byte row XOR-swizzled by (row & 3); the MFMA fragments (lane = channel, 8 consecutive rows) come out of those images with
// ds_read_b64_tr_b16 — the hardware transpose (tools/micro/tr_read_probe.hip pins its lane map) — conflict-free.
struct RowCursor {
    const int* pre;   // LDS: pre[b] = blocks before cloud b, pre[B] = T
    int b, q, half, nbq;
    __device__ __forceinline__ void seek(int blk, int B) {   // binary search: the cloud holding block blk (< T)
        int lo = 0, hi = B;
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (pre[mid] <= blk) lo = mid;
            else hi = mid;
        }
        b = lo;
        q = blk - pre[lo];
        half = 0;
        nbq = pre[lo + 1] - pre[lo];
    }
    __device__ __forceinline__ long row() const { return (long)b * 512 + q * 32 + half * 16; }
    __device__ __forceinline__ void next() {   // (clouds have at least one block)
        half ^= 1;
        if (half == 0 && ++q == nbq) {
            ++b;
            q = 0;
            nbq = pre[b + 1] - pre[b];
        }
    }
};

typedef short s16x4 __attribute__((__vector_size__(4 * sizeof(short))));
typedef short s16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ float pow2c(int e) { return e < -126 ? 0.f : pow2f(min(e, 127)); }
// 8 consecutive rows of one channel per lane: two transposed 4-row reads
__device__ __forceinline__ f16x8 tr_frag(const unsigned char* p) {
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p + 1024));      // rows + 4
    s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(f16x8, v);
}
// hi / lo pieces of 4 values under sc, packed: 8 bytes each
__device__ __forceinline__ void split4s(const f32x4& x, float sc, uint2& hi, uint2& lo) {
    unsigned hh[2], ll[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        asm("v_fma_mixlo_f16 %0, %1, %3, 0 op_sel_hi:[0,0,0]\n\t"
            "v_fma_mixhi_f16 %0, %2, %3, 0 op_sel_hi:[0,0,0]"
            : "=&v"(hh[i])
            : "v"(x[2 * i]), "v"(x[2 * i + 1]), "v"(sc));
        asm("v_fma_mixlo_f16 %0, %1, %3, -%4 op_sel:[0,0,0] op_sel_hi:[0,0,1]\n\t"
            "v_fma_mixhi_f16 %0, %2, %3, -%4 op_sel:[0,0,1] op_sel_hi:[0,0,1]"
            : "=&v"(ll[i])
            : "v"(x[2 * i]), "v"(x[2 * i + 1]), "v"(sc), "v"(hh[i]));
    }
    hi = make_uint2(hh[0], hh[1]);
    lo = make_uint2(ll[0], ll[1]);
}

constexpr int kImg = 16 * 256;            // one 16-row image (hi or lo) of up to 128 channels
constexpr int kDwBuf = 4 * kImg;          // D hi, D lo, H hi, H lo
// byte offset of (row, channel col) in an image
__device__ __forceinline__ int img_off(int row, int col) { return row * 256 + ((((col >> 5) ^ row) & 3) << 6) + (col & 31) * 2; }

// One workgroup: P(128 x NT at m0, n0) = sum over the range's rows of D[row][m]^T H[row][n]  (D ld M, H ld N; NT = 128, or 64:
// wave (wm, wn) then owns 64 x 32).  iD / iH: the operands' slots in bexp.
template <int NT>
__device__ __forceinline__ void dw16_task(const float* __restrict__ D, int M, const float* __restrict__ H, int N, int m0, int n0,
                                          RowCursor cur, int nch, const int* __restrict__ bexp, int iD, int iH,
                                          float* __restrict__ P, float* __restrict__ Pdb, unsigned char* lds, float* red, int tid) {
    constexpr int TN = NT / 64;               // n tiles per wave
    const int lane = tid & 63, w = tid >> 6, r = lane & 31, h = lane >> 5;
    const int wm = w >> 1, wn = w & 1;
    f32x16 acc[2][TN];
#pragma unroll
    for (int ti = 0; ti < 2; ++ti)
#pragma unroll
        for (int tj = 0; tj < TN; ++tj)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[ti][tj][e] = 0.f;
    f32x4 dsum = {0.f, 0.f, 0.f, 0.f};
    struct RegSet {                           // one chunk's share of a thread
        f32x4 a0, a1, b0, b1;
        int eD, eH;                           // the block's exponents (bexp)
    } r0, r1, r2, r3;
    const int arow = tid >> 5, acol = (tid & 31) * 4;                                   // D tile: rows arow, arow + 8
    const int brow = NT == 128 ? (tid >> 5) : (tid >> 4), bcol = NT == 128 ? (tid & 31) * 4 : (tid & 15) * 4;
    int left = nch - 1;                       // chunks the cursor may still advance
    auto load = [&](RegSet& g) __attribute__((always_inline)) {
        const long row = cur.row();
        const float* dp = D + (row + arow) * M + m0 + acol;
        const float* hp = H + (row + brow) * N + n0 + bcol;
        g.a0 = *reinterpret_cast<const f32x4*>(dp);
        g.a1 = *reinterpret_cast<const f32x4*>(dp + 8L * M);
        g.b0 = *reinterpret_cast<const f32x4*>(hp);
        g.b1 = NT == 128 ? *reinterpret_cast<const f32x4*>(hp + 8L * N) : g.b0;
        const int* be = bexp + (row >> 5) * 8;
        g.eD = be[iD];
        g.eH = be[iH];
        if (left > 0) {                       // else: stay on the last chunk
            --left;
            cur.next();
        }
    };
    // staging state (the exponents the images are written under) and compute state (the exponents the accumulators are in)
    int sD = 0, sH = 0, cD = 0, cH = 0;
    float fD = 1.f, fH = 1.f;
    auto new_block = [&](const RegSet& g, bool first) __attribute__((always_inline)) {
        // (the exponents stay in vector registers until here: hipcc otherwise reads them into scalar registers right behind
        //  the loads, i.e. waits out a memory round trip in every chunk — 1 us per chunk, measured)
        int eD = g.eD, eH = g.eH;
        asm volatile("" : "+v"(eD), "+v"(eH));
        eD = __builtin_amdgcn_readfirstlane(eD);
        eH = __builtin_amdgcn_readfirstlane(eH);
        if (!first) {
            const int excess = (eD + eH) - (sD + sH) - 40;      // growth of the scale beyond 2^40 per block: give it up
            if (excess > 0) eD -= excess;
        }
        eD = max(eD, -120);
        sD = eD;
        sH = eH;
        fD = pow2f(sD);
        fH = pow2f(sH);
    };
    // piece `part` (0..3) of the staging of a chunk's register set into LDS buffer `buf`: split + two 8-byte writes
    auto store_part = [&](int buf, const RegSet& g, int part) __attribute__((always_inline)) {
        unsigned char* base = lds + buf * kDwBuf;
        uint2 hi, lo;
        if (part == 0) {
            split4s(g.a0, fD, hi, lo);
            *reinterpret_cast<uint2*>(base + img_off(arow, acol)) = hi;
            *reinterpret_cast<uint2*>(base + kImg + img_off(arow, acol)) = lo;
        } else if (part == 1) {
            split4s(g.a1, fD, hi, lo);
            *reinterpret_cast<uint2*>(base + img_off(arow + 8, acol)) = hi;
            *reinterpret_cast<uint2*>(base + kImg + img_off(arow + 8, acol)) = lo;
            dsum += g.a0 + g.a1;
        } else if (part == 2) {
            split4s(g.b0, fH, hi, lo);
            *reinterpret_cast<uint2*>(base + 2 * kImg + img_off(brow, bcol)) = hi;
            *reinterpret_cast<uint2*>(base + 3 * kImg + img_off(brow, bcol)) = lo;
        } else if (NT == 128) {
            split4s(g.b1, fH, hi, lo);
            *reinterpret_cast<uint2*>(base + 2 * kImg + img_off(brow + 8, bcol)) = hi;
            *reinterpret_cast<uint2*>(base + 3 * kImg + img_off(brow + 8, bcol)) = lo;
        }
    };
    auto store = [&](int buf, const RegSet& g) __attribute__((always_inline)) {
#pragma unroll
        for (int part = 0; part < 4; ++part) store_part(buf, g, part);
    };
    // transposed-read addresses: lane -> (row 8 h + q [+ 4], channels 16 g1 + 4 p .. + 3) of a 32-channel tile
    const int q = (lane & 15) >> 2, pp = lane & 3, g1 = (lane >> 4) & 1;
    const int tr0 = (8 * h + q) * 256 + 32 * g1 + 8 * pp;
    int offA[2], offB[TN];
#pragma unroll
    for (int t = 0; t < 2; ++t) offA[t] = tr0 + ((((2 * wm + t) ^ q) & 3) << 6);
#pragma unroll
    for (int t = 0; t < TN; ++t) offB[t] = tr0 + ((((NT == 128 ? 2 * wn + t : wn) ^ q) & 3) << 6);
    // the MFMAs of LDS buffer `buf`; with `g` != NULL the staging of the next chunk's register set (into the other buffer) is
    // issued in four pieces BETWEEN the MFMA groups, pinned there: the split is ~40 vector instructions per thread and chunk that
    // otherwise run behind the MFMAs, nothing overlapping (1 us per chunk: both waves of a SIMD serialised, measured)
    auto compute = [&](int buf, const RegSet* g) __attribute__((always_inline)) {
        const unsigned char* base = lds + buf * kDwBuf;
        f16x8 ah[2], al[2], bh[TN], bl[TN];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            ah[t] = tr_frag(base + offA[t]);
            al[t] = tr_frag(base + kImg + offA[t]);
        }
#pragma unroll
        for (int t = 0; t < TN; ++t) {
            bh[t] = tr_frag(base + 2 * kImg + offB[t]);
            bl[t] = tr_frag(base + 3 * kImg + offB[t]);
        }
#pragma unroll
        for (int pr = 0; pr < 3; ++pr) {
#pragma unroll
            for (int ti = 0; ti < 2; ++ti)
#pragma unroll
                for (int tj = 0; tj < TN; ++tj)
                    acc[ti][tj] = __builtin_amdgcn_mfma_f32_32x32x16_f16(pr == 2 ? al[ti] : ah[ti], pr == 1 ? bl[tj] : bh[tj], acc[ti][tj], 0, 0, 0);
            if (g) {
                HP_SB();
                store_part(buf ^ 1, *g, pr);
                if (pr == 2) store_part(buf ^ 1, *g, 3);
                HP_SB();
            }
        }
    };
    auto rescale = [&]() __attribute__((always_inline)) {      // the accumulators move to the staged block's exponents
        const int d = (sD + sH) - (cD + cH);
        if (d != 0) {
            const float f = pow2c(d);
#pragma unroll
            for (int ti = 0; ti < 2; ++ti)
#pragma unroll
                for (int tj = 0; tj < TN; ++tj)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[ti][tj][e] *= f;
        }
        cD = sD;
        cH = sH;
    };
    load(r0);
    load(r1);
    load(r2);
    load(r3);
    new_block(r0, true);
    cD = sD;
    cH = sH;
    if (nch > 0) store(0, r0);      // (an empty range — fewer blocks than ranges — leaves zeros)
    __syncthreads();
    // chunk it + K: its image is LDS buffer K & 1 and register set K is free for chunk it + K + 4; then chunk it + K + 1 goes
    // from its register set to the other LDS buffer (an even chunk opens a block: new exponents)
#define HP_DW_STEP(K, RK, RN)                                                  \
    load(RK);                                                                  \
    if (it + K + 1 < nch) {                                                    \
        if ((K & 1) == 0 && it + K > 0) rescale();                             \
        if ((K & 1) == 1) new_block(RN, false);                                \
        compute(K & 1, &RN);                                                   \
    } else if (it + K < nch) {                                                 \
        if ((K & 1) == 0 && it + K > 0) rescale();                             \
        compute(K & 1, nullptr);                                               \
    }                                                                          \
    __syncthreads();
#pragma unroll 1
    for (int it = 0; it < nch; it += 4) {
        HP_DW_STEP(0, r0, r1)
        HP_DW_STEP(1, r1, r2)
        HP_DW_STEP(2, r2, r3)
        HP_DW_STEP(3, r3, r0)
    }
#undef HP_DW_STEP
    // lane (r, h), register e of tile (ti, tj): P[m0 + 64 wm + 32 ti + drow(e, h)][n0 + (NT / 2) wn + 32 tj + r]
    const float u0 = pow2f(-cD), u1 = pow2f(-cH);
#pragma unroll
    for (int ti = 0; ti < 2; ++ti)
#pragma unroll
        for (int tj = 0; tj < TN; ++tj)
#pragma unroll
            for (int e = 0; e < 16; ++e)
                P[(long)(m0 + 64 * wm + 32 * ti + drow(e, h)) * N + n0 + (NT / 2) * wn + 32 * tj + r] = acc[ti][tj][e] * u0 * u1;
    if (Pdb && n0 == 0) {      // bias gradient = column sums of D: the eight row classes of the staging map, added in order
        __syncthreads();
        *reinterpret_cast<f32x4*>(red + (arow * 32 + (tid & 31)) * 4) = dsum;
        __syncthreads();
        if (tid < 32) {
            f32x4 o = *reinterpret_cast<const f32x4*>(red + tid * 4);
#pragma unroll
            for (int k = 1; k < 8; ++k) o += *reinterpret_cast<const f32x4*>(red + (k * 32 + tid) * 4);
            *reinterpret_cast<f32x4*>(Pdb + m0 + 4 * tid) = o;
        }
    }
}

// partial-sum offsets inside one range (HP_EB_PART_FLOATS; enc_bwd.hip's reduce launch reads them)
constexpr int oW4 = 0, oW3 = oW4 + 512 * 256, oW2 = oW3 + 256 * 128, oW1 = oW2 + 128 * 64, oB4 = oW1 + 64 * 3,
              oB3 = oB4 + 512, oB2 = oB3 + 256, oB1 = oB2 + 128;
static_assert(oB1 + 64 == HP_EB_PART_FLOATS, "partial layout");
constexpr int kRangeWgs = 8 + 2 + 1 + 1;  // dW4: 8 tiles of 128 x 128, dW3: 2, dW2: one 128 x 64 tile, and one for dW1 + db1 (vector unit)
constexpr int kMaxClouds = 2048;          // LDS table of block prefixes (hp_enc_bwd_max_clouds)

// Workgroup id -> (group = (encoder, range), task): the 11 tasks of a group get ids that are congruent mod 8, i.e. ONE XCD and one
// L2 — they read the same delta / hc rows at about the same time (each row 2..4 times).
__global__ __launch_bounds__(256, 2) void enc_bwd_dw_f16_kernel(const HpEncBwdArgs a) {
    __shared__ __attribute__((aligned(1024))) unsigned char lds[2 * kDwBuf];
    __shared__ __attribute__((aligned(16))) float red[8 * 32 * 4];
    __shared__ int pre[kMaxClouds + 1];
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int G = xcd + 8 * (slot / kRangeWgs), t = slot % kRangeWgs;
    if (G >= a.S * a.n) return;
    const int z = G % a.n, split = G / a.n;
    const HpEncBwdSide& s = a.e[z];
    const int S = a.S, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    long long* prof = a.prof ? a.prof + (long)blockIdx.x * 10 : nullptr;      // HP_EB_PROF16 (debug)
    if (prof && tid == 0) prof[0] = (long long)wall_clock64();
    // block prefixes of the clouds (wave 0: 64-lane scans, carried over the chunks of 64 clouds)
    if (w == 0) {
        int carry = 0;
        for (int c0 = 0; c0 < a.B; c0 += 64) {
            const int nb = c0 + lane < a.B ? (s.crit.cnt[c0 + lane] + 31) >> 5 : 0;
            const int inc = wave_scan(nb, lane);
            if (c0 + lane < a.B) pre[c0 + lane] = carry + inc - nb;
            carry += __shfl(inc, 63, 64);
        }
        if (lane == 0) pre[a.B] = carry;
    }
    __syncthreads();
    const int T = pre[a.B];
    const int blk0 = (int)((long)split * T / S), blk1 = (int)((long)(split + 1) * T / S);
    const int nch = 2 * (blk1 - blk0);
    RowCursor cur;
    cur.pre = pre;
    cur.seek(min(blk0, T - 1), a.B);
    float* P = s.part + (long)split * HP_EB_PART_FLOATS;
    if (prof && tid == 0) {
        prof[1] = (long long)wall_clock64();
        prof[3] = t < 8 ? 1 : (t < 10 ? 2 : (t == 10 ? 3 : 4));
        prof[4] = nch;
    }
    if (t < 8) {            // dW4: tile (m = t >> 1, n = t & 1)
        dw16_task<128>(s.d[4], 512, s.hc[3], 256, 128 * (t >> 1), 128 * (t & 1), cur, nch, s.bexp, 0, 4, P + oW4, P + oB4, lds, red, tid);
    } else if (t < 10) {    // dW3: tiles m = t - 8
        dw16_task<128>(s.d[3], 256, s.hc[2], 128, 128 * (t - 8), 0, cur, nch, s.bexp, 1, 5, P + oW3, P + oB3, lds, red, tid);
    } else if (t == 10) {   // dW2: one tile of 128 x 64
        dw16_task<64>(s.d[2], 128, s.hc[1], 64, 0, 0, cur, nch, s.bexp, 2, 6, P + oW2, P + oB2, lds, red, tid);
    } else {
        // dW1 (64 x 3) + db1: channel c = lane, wave w the rows = w mod 4; the four row classes are added in order
        float ax = 0.f, ay = 0.f, az = 0.f, ab = 0.f;
        for (int c = 0; c < nch; ++c) {
            const long row = cur.row() + w;
            const float* dp = s.d[1] + row * 64 + lane;
            const float* xp = s.hc[0] + row * 3;
#pragma unroll
            for (int k = 0; k < 4; ++k) {       // (rows past a cloud's count hold zeros in both operands)
                const float dv = dp[(long)(4 * k) * 64];
                ax = __builtin_fmaf(dv, xp[12 * k + 0], ax);
                ay = __builtin_fmaf(dv, xp[12 * k + 1], ay);
                az = __builtin_fmaf(dv, xp[12 * k + 2], az);
                ab += dv;
            }
            if (c + 1 < nch) cur.next();
        }
        __syncthreads();
        float* r1 = red;
        if (w) {
            r1[((w - 1) * 64 + lane) * 4 + 0] = ax;
            r1[((w - 1) * 64 + lane) * 4 + 1] = ay;
            r1[((w - 1) * 64 + lane) * 4 + 2] = az;
            r1[((w - 1) * 64 + lane) * 4 + 3] = ab;
        }
        __syncthreads();
        if (w == 0) {
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                ax += r1[(k * 64 + lane) * 4 + 0];
                ay += r1[(k * 64 + lane) * 4 + 1];
                az += r1[(k * 64 + lane) * 4 + 2];
                ab += r1[(k * 64 + lane) * 4 + 3];
            }
            P[oW1 + lane * 3 + 0] = ax;
            P[oW1 + lane * 3 + 1] = ay;
            P[oW1 + lane * 3 + 2] = az;
            P[oB1 + lane] = ab;
        }
    }
    if (prof && tid == 0) prof[2] = (long long)wall_clock64();
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
int hp_enc_bwd_dw_f16(const HpEncBwdArgs* a0, hipStream_t stream) {
    static const bool prof_on = std::getenv("HP_EB_PROF16") != nullptr;
    const int groups = a0->S * a0->n;
    if (prof_on) {      // debug: start / prologue / end stamps of every workgroup (synchronises)
        static long long* buf = nullptr;
        const long nwg = 8L * kRangeWgs * ((groups + 7) / 8);
        if (!buf) (void)hipMalloc(&buf, sizeof(long long) * 10 * 65536);
        HpEncBwdArgs a = *a0;
        (void)hipMemsetAsync(buf, 0, sizeof(long long) * 10 * nwg, stream);
        a.prof = buf;
        hipLaunchKernelGGL(enc_bwd_dw_f16_kernel, dim3((unsigned)nwg), dim3(256), 0, stream, a);
        (void)hipStreamSynchronize(stream);
        std::vector<long long> hb(10 * nwg);
        (void)hipMemcpy(hb.data(), buf, sizeof(long long) * 10 * nwg, hipMemcpyDeviceToHost);
        long long tmin = -1, tmax = 0;
        for (long i = 0; i < nwg; ++i)
            if (hb[i * 10 + 3]) {
                if (tmin < 0 || hb[i * 10] < tmin) tmin = hb[i * 10];
                tmax = std::max(tmax, hb[i * 10 + 2]);
            }
        for (int ty = 1; ty <= 4; ++ty) {
            double pro = 0, tot = 0, st = 0, stmax = 0, mx = 0, nchs = 0;
            long cnt = 0;
            for (long i = 0; i < nwg; ++i) {
                const long long* t = &hb[i * 10];
                if (t[3] != ty) continue;
                ++cnt;
                pro += (double)(t[1] - t[0]) * 0.01;
                tot += (double)(t[2] - t[0]) * 0.01;
                mx = std::max(mx, (double)(t[2] - t[0]) * 0.01);
                st += (double)(t[0] - tmin) * 0.01;
                stmax = std::max(stmax, (double)(t[0] - tmin) * 0.01);
                nchs += (double)t[4];
            }
            if (cnt)
                fprintf(stderr, "[dw16 prof] type %d: %ld wgs, start %.1f(max %.1f) us, prologue %.1f, total %.1f(max %.1f) us, chunks %.1f\n", ty, cnt,
                        st / cnt, stmax, pro / cnt, tot / cnt, mx, nchs / cnt);
        }
        fprintf(stderr, "[dw16 prof] span %.1f us\n", (double)(tmax - tmin) * 0.01);
        HP_RETURN_LAST_ERROR();
    }
    const HpEncBwdArgs* a = a0;
    hipLaunchKernelGGL(enc_bwd_dw_f16_kernel, dim3((unsigned)(8 * kRangeWgs * ((groups + 7) / 8))), dim3(256), 0, stream, *a);
    HP_RETURN_LAST_ERROR();
}

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
