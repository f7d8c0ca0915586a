// Fused backward of the PointNet encoders' conv stack on gfx950 — the autograd of
//   /root/reference/model/encoder.py:14-28 (Conv1d(k=1)+ReLU x4, Conv1d) and :45 (max over points)
// for ONE encoder or the TWO encoders of a HyperPocket step (model/full_model.py:106-112) in the same launches.
//
// Only the arg-max ("critical") points carry gradient below the max-pool, and channels that peak at
// the same point share one row: ~170 distinct rows per cloud instead of N.  Round 2 ran that as ~17 dependent launches per
// encoder (sort, gather, delta4, dW5, then a dX GEMM, a dW GEMM and a split-K reduce per layer) — 50 us of matrix work in
// ~300 us.  Here, five launches for both encoders, split by what bounds them:
//   prep    per (encoder, cloud): bitonic sort of the 512 (point, channel) keys -> slots (rows); the VAE head's elementwise
//           backward rides along
//   gather  everything that is a chain of dependent memory round trips, at high occupancy (no LDS tiles, few registers):
//           delta4[row] = (h4[row] > 0) * sum over the row's channels of dg * W5 rows — one wave per row — and
//           dW5[c,:] = sum_b dg[b,c] * h4[b, argmax(b,c), :], db5 — one workgroup per channel
//   chain   matrix cores only: a workgroup owns 32 rows of one cloud, stages their delta4 in LDS and walks
//           delta4 -> delta3 -> delta2 -> delta1 with the previous delta as the A operand and the weights streamed from L2
//           as B fragments (W_l rows run along the output columns: 128-byte segments, no staging); each delta_l and the
//           rows' activations below it are stored once, rows past the cloud's count as zeros
//   dW      dW_l = delta_l^T h_{l-1}, db_l for l = 4..1 of both encoders as EQUAL tasks of one grid: both operands run along
//           the lanes an MFMA fragment wants (fragments straight from global memory, four chunks in flight), the
//           contraction over the rows cut into S ranges of whole 32-row blocks
//   reduce  the ranges' partial sums added in range order
// No atomics, fixed summation orders: run-to-run identical; per row the arithmetic does not depend on how many encoders
// share the launches.
//
// What shaped it (measured, tools/micro/enc_bwd_probe.hip + HP_EB_PROF stamps; docs/DESIGN_HISTORY.md §3.5):
//  * a grid of 16 row blocks per cloud leaves the dead blocks interleaved with the live ones and XCDs 6, 7 without a live
//    block: 296 us against 168 with the live blocks as the first, contiguous ids;
//  * hipcc sinks prefetch loads behind the MFMA block that should cover them, and a load inside a branch makes the waitcnt
//    pass drain the queue at the join: every prefetch is branch-free and pinned with sched_barrier;
//  * a row's channel list is heavy-tailed (up to ~50 of a cloud's 512 channels peak at ONE point): a per-row loop inside a
//    32-row MFMA workgroup stalls all four waves behind the longest row — hence the separate gather launch;
//  * equal tasks: with 8 + 2 + 1 workgroups per range of unequal cost the dW launch took 2.1x its MFMA time.
#include "hp_common.h"
#include "hp_enc_bwd.h"
#include "hp_conv_split.h"
#include "hp_enc_bwd_wprep.h"
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

bool hp_enc_bwd_dw_f16_enabled();

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kRows = 32;                        // rows of a chain workgroup
constexpr int LD3 = 260, LD2 = 132;   // LDS row strides of (half a) delta4 / delta3 and of delta2 (= 4 mod 32: conflict-free b128)

// partial-sum offsets inside one range (HP_EB_PART_FLOATS)
constexpr int oW4 = 0, oW3 = oW4 + 512 * 256, oW2 = oW3 + 256 * 128, oW1 = oW2 + 128 * 64, oB4 = oW1 + 64 * 3,
              oB3 = oB4 + 512, oB2 = oB3 + 256, oB1 = oB2 + 128;
static_assert(oB1 + 64 == HP_EB_PART_FLOATS, "partial layout");

#define HP_SB() __builtin_amdgcn_sched_barrier(0)

__device__ __forceinline__ int drow(int e, int h) { return (e & 3) + 8 * (e >> 2) + 4 * h; }   // C/D map of 32x32 f32
__device__ __forceinline__ int ru32(int v) { return (v + 31) & ~31; }
__device__ __forceinline__ float f4at(const float4& v, int s) { return s == 0 ? v.x : (s == 1 ? v.y : (s == 2 ? v.z : v.w)); }
__device__ __forceinline__ float4 f4fma(float g, const float4& w, const float4& a) {
    return make_float4(__builtin_fmaf(g, w.x, a.x), __builtin_fmaf(g, w.y, a.y), __builtin_fmaf(g, w.z, a.z), __builtin_fmaf(g, w.w, a.w));
}
__device__ __forceinline__ float4 f4mask(const float4& h, const float4& v) {
    return make_float4(h.x > 0.f ? v.x : 0.f, h.y > 0.f ? v.y : 0.f, h.z > 0.f ? v.z : 0.f, h.w > 0.f ? v.w : 0.f);
}

// ---- the forward's activations: fp32 rows, or P-format lines (conv_pp.hip) — (hi + lo) * 2^-e is exact in fp32 ----------------
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ float pexp_unscale(int e) { return __uint_as_float((unsigned)(127 - e) << 23); }   // 2^-e, e in [-100, 54]
// channels c8 .. c8+7 (c8 % 8 == 0) of a row of C channels starting at byte address `rowp`: two float4 (fp32 rows), or one
// 16-byte chunk of the hi pieces + the matching chunk of the lo pieces, unscaled by `us` = 2^-e of the row's block
__device__ __forceinline__ void act8(const unsigned char* rowp, bool pfmt, int c8, float us, float4& a, float4& b) {
    if (!pfmt) {
        a = *reinterpret_cast<const float4*>(rowp + c8 * 4);
        b = *reinterpret_cast<const float4*>(rowp + c8 * 4 + 16);
        return;
    }
    const unsigned char* line = rowp + (c8 >> 5) * 128 + (c8 & 31) * 2;
    const f16x8 hi = *reinterpret_cast<const f16x8*>(line), lo = *reinterpret_cast<const f16x8*>(line + 64);
    a = make_float4(((float)hi[0] + (float)lo[0]) * us, ((float)hi[1] + (float)lo[1]) * us, ((float)hi[2] + (float)lo[2]) * us,
                    ((float)hi[3] + (float)lo[3]) * us);
    b = make_float4(((float)hi[4] + (float)lo[4]) * us, ((float)hi[5] + (float)lo[5]) * us, ((float)hi[6] + (float)lo[6]) * us,
                    ((float)hi[7] + (float)lo[7]) * us);
}

#ifndef HP_EB_GWG
#define HP_EB_GWG 4096
#endif
#ifndef HP_EB_GEB
#define HP_EB_GEB 4
#endif
constexpr int kGatherRowWgs = HP_EB_GWG;    // persistent row workgroups per encoder (gather launch)

// HP_EB_PROF: start / end / a tag of a workgroup, written by thread 0 (timing experiments; prof == NULL in production)
struct Stamp {
    long long* p;
    int tid, type;
    __device__ Stamp(long long* p_, int tid_) : p(p_), tid(tid_), type(0) {
        if (p && tid == 0) p[0] = (long long)wall_clock64();
    }
    __device__ ~Stamp() {
        if (p && tid == 0) {
            p[1] = (long long)wall_clock64();
            p[2] = type;
        }
    }
};

// ---------------------------------------------------------------------------------------------------------------------
// prep: critical-point compaction of cloud b of encoder z (+ the VAE head's backward for row b)
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512) void enc_bwd_prep_kernel(const HpEncBwdArgs a) {
    __shared__ int key[512];
    __shared__ int scan[512];
    const HpEncBwdSide& s = a.e[blockIdx.y];
    const int b = blockIdx.x, t = threadIdx.x;
    if (b >= a.B) {      // the f16 chain's weight stream (enc_bwd_f16.hip): 14 workgroups behind the clouds'
        hp_wprep::task(s, b - a.B, t);
        return;
    }
    if (s.is_vae && t < a.out) {   // model/encoder.py:38-41,49-51: z = eps*exp(lv) + mu, returned "logvar" = exp(lv)
        const long i = (long)b * a.out + t;
        const float gz = s.gout ? s.gout[(long)b * s.gout_ld + t] : 0.f;
        s.dmu[i] = gz + (s.gmu ? s.gmu[i] : 0.f);
        s.dlv[i] = (gz * s.eps[i] + (s.gexplv ? s.gexplv[i] : 0.f)) * expf(s.lv[i]);
    }
    key[t] = (s.argidx[(long)b * 512 + t] << 9) | t;
    __syncthreads();
    for (int k = 2; k <= 512; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            const int o = t ^ j;
            if (o > t) {
                const int x = key[t], y = key[o];
                const bool up = (t & k) == 0;
                if ((x > y) == up) {
                    key[t] = y;
                    key[o] = x;
                }
            }
            __syncthreads();
        }
    const int mine = key[t], p = mine >> 9, ch = mine & 511;
    const int flag = (t == 0 || (key[t - 1] >> 9) != p) ? 1 : 0;
    scan[t] = flag;
    __syncthreads();
    for (int d = 1; d < 512; d <<= 1) {   // inclusive scan
        const int v = t >= d ? scan[t - d] : 0;
        __syncthreads();
        scan[t] += v;
        __syncthreads();
    }
    const int u = scan[t] - 1;
    const HpCrit& c = s.crit;
    c.chan[(long)b * 512 + t] = ch;
    c.eslot[(long)b * 512 + t] = u;
    c.slot[(long)b * 512 + ch] = u;
    if (flag) {
        c.pt[(long)b * 512 + u] = p;
        c.start[(long)b * 513 + u] = t;
    }
    if (t == 511) {
        c.cnt[b] = u + 1;
        c.start[(long)b * 513 + u + 1] = 512;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// gather: delta4 rows and dW5 — chains of dependent memory round trips, hidden by occupancy
// ---------------------------------------------------------------------------------------------------------------------
// Workgroups [kGatherRowWgs, kGatherRowWgs + 512) of an encoder: dW5[c,:] = sum_b dg[b,c] * h4[b, argmax[b,c], :] (one-hot upstream of the max-pool),
// db5[c] = sum_b dg[b,c]: thread (q = tid & 127: 4 consecutive k, g = tid >> 7: cloud parity), the (argmax, dg) pairs of
// 256 clouds at a time through LDS so that the h4 row loads are one round trip deep; even clouds + odd clouds.
// Workgroups [0, kGatherRowWgs): four rows at a time, ONE WAVE per row (cloud b, slot u): delta4[row] = (h4[row] > 0) * sum over
// the row's channels, ascending, of dg[b,c] * W5[c,:]; a lane owns 8 of the 512 columns; rows in [cnt, ru32(cnt)) are
// written as zeros (the matrix-core launches run on whole 32-row blocks).
// Round 5: four W5 rows in flight per wave (HP_EB_GEB, was 8) at four workgroups per CU (98 VGPRs, no spill; was 148 / three):
// the launch is a chain of dependent memory round trips, occupancy hides them — 125 -> 108 us in the step's trace (the step
// itself does not move: the region is bandwidth-bound beside the heads' dW + Adam pass, docs/DESIGN_HISTORY.md 8).
#ifndef HP_EB_GOCC
#define HP_EB_GOCC 4
#endif
__global__ __launch_bounds__(256, HP_EB_GOCC) void enc_bwd_gather_kernel(const HpEncBwdArgs a) {
    __shared__ int sarg[256];
    __shared__ float sdg[256];
    __shared__ float sus[256][4];
    __shared__ float4 red[3][64][2];
    const int id = blockIdx.x;
    const int z = a.n == 2 ? (id & 1) : 0, rest = a.n == 2 ? (id >> 1) : id;
    const HpEncBwdSide& s = a.e[z];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    Stamp stamp(a.prof ? a.prof + (long)id * 4 : nullptr, tid);
    const bool pfmt = *s.fmt == HP_PP_FMT_P;      // (uniform: the format the forward left h1..h4 in)
    const unsigned char* h4b = reinterpret_cast<const unsigned char*>(s.h[4]);
    if (rest >= kGatherRowWgs) {      // (the uniform 11-us channel tasks behind the row tasks, whose length varies)
        // dW5[c,:] = sum_b dg[b,c] * h4[b, argmax[b,c], :], db5[c] = sum_b dg[b,c]: lane q owns channels 8q..8q+7 of a row (one
        // 16-byte chunk of hi pieces + one of lo pieces, or two float4), wave g takes the clouds b = g mod 4; the (argmax, dg,
        // unscale) triples of 256 clouds at a time go through LDS so that the row loads are one round trip deep; the four
        // cloud classes are added in the order 0..3.
        const int c = rest - kGatherRowWgs;
        stamp.type = 1;
        const int q = lane, g = w;
        float4 acc0 = make_float4(0.f, 0.f, 0.f, 0.f), acc1 = acc0;
        float bsum = 0.f;
        for (int c0 = 0; c0 < a.B; c0 += 256) {
            const int nb = min(256, a.B - c0);
            __syncthreads();
            if (tid < nb) {
                const long row = (long)(c0 + tid) * 512 + c;
                const int arg = s.argidx[row];
                sarg[tid] = arg;
                sdg[tid] = s.dg[row];
                if (pfmt) {
                    const long hrow = (long)(c0 + tid) * a.Np + arg;
#pragma unroll
                    for (int k = 0; k < 4; ++k) sus[tid][k] = pexp_unscale(s.pexp[4][(hrow >> 7) * s.pncb[4] + min(k, s.pncb[4] - 1)]);
                }
            }
            __syncthreads();
            if (tid == 0)
                for (int b = 0; b < nb; ++b) bsum += sdg[b];
#pragma unroll 4
            for (int b = g; b < nb; b += 4) {
                float4 h0, h1;
                act8(h4b + ((long)(c0 + b) * a.Np + sarg[b]) * 2048, pfmt, 8 * q, pfmt ? sus[b][(8 * q) >> s.pcbs[4]] : 1.f, h0, h1);
                acc0 = f4fma(sdg[b], h0, acc0);
                acc1 = f4fma(sdg[b], h1, acc1);
            }
        }
        if (tid == 0 && s.gb[4]) s.gb[4][c] = bsum;
        if (g) {
            red[g - 1][q][0] = acc0;
            red[g - 1][q][1] = acc1;
        }
        __syncthreads();
        if (g == 0) {
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const float4 o0 = red[k][q][0], o1 = red[k][q][1];
                acc0 = make_float4(acc0.x + o0.x, acc0.y + o0.y, acc0.z + o0.z, acc0.w + o0.w);
                acc1 = make_float4(acc1.x + o1.x, acc1.y + o1.y, acc1.z + o1.z, acc1.w + o1.w);
            }
            *reinterpret_cast<float4*>(s.gW[4] + (long)c * 512 + 8 * q) = acc0;
            *reinterpret_cast<float4*>(s.gW[4] + (long)c * 512 + 8 * q + 4) = acc1;
        }
        return;
    }
    // row workgroups are PERSISTENT: kGatherRowWgs of them per encoder stride over the B*128 groups of four slots (a grid
    // of one workgroup per group — 16K workgroups, two thirds of them dead — was bound by workgroup dispatch: ~50 us).
    // ONE WAVE per row (cloud b, slot u), a lane owns the 8 columns [8 lane, +8) of the 512:
    //   delta4[row] = (h4[row] > 0) * sum over the row's channels, ascending, of dg[b,c] * W5[c,:]
    // and the row's activations below (h3: lanes 0..31, h2: 32..47, h1: 48..55) are copied — out of the forward's arrays,
    // whatever their format — into the compact fp32 rows hc[1..3] the chain and dW launches read.  Rows in [cnt, ru32(cnt))
    // are written as zeros (the matrix-core launches run on whole 32-row blocks).
    stamp.type = 2;
    const float* w5 = s.W[4] + 8 * lane;
    // which of h3 / h2 / h1 this lane copies, and its 8 channels there
    const int hl = lane < 32 ? 3 : (lane < 48 ? 2 : (lane < 56 ? 1 : 0));
    const int hC = hl == 3 ? 256 : (hl == 2 ? 128 : 64), hc8 = hl == 3 ? 8 * lane : (hl == 2 ? 8 * (lane - 32) : 8 * (lane - 48));
    for (int rg = rest; rg < a.B * 128; rg += kGatherRowWgs) {
        // group rg = (slot group q = rg / B, cloud b = rg % B): a workgroup's groups are then spread over the slot range —
        // only the first ~cnt/4 groups of a cloud are live — instead of all low or all high
        const int b = rg % a.B, u = (rg / a.B) * 4 + w;      // (the four waves take neighbouring rows)
        const int cnt = s.crit.cnt[b];
        if (u >= ru32(cnt)) continue;
        const int* chan = s.crit.chan + (long)b * 512;
        const float* dgb = s.dg + (long)b * 512;
        const long crow = (long)b * 512 + u;
        float* dst = s.d[4] + crow * 512 + 8 * lane;
        float* hdst = hl ? s.hc[hl] + crow * hC + hc8 : nullptr;
        if (u >= cnt) {
            const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
            *reinterpret_cast<float4*>(dst) = zero;
            *reinterpret_cast<float4*>(dst + 4) = zero;
            if (lane == 0) s.d4max[crow] = 0.f;
            if (lane < 4) s.hmax[crow * 4 + lane] = 0.f;
            s.hmask[crow * 64 + lane] = 0;
            if (hl) {
                *reinterpret_cast<float4*>(hdst) = zero;
                *reinterpret_cast<float4*>(hdst + 4) = zero;
            }
            continue;
        }
        const int i0 = s.crit.start[(long)b * 513 + u], i1 = s.crit.start[(long)b * 513 + u + 1];
        const long hrow = (long)b * a.Np + s.crit.pt[(long)b * 512 + u];
        float us4 = 1.f, usl = 1.f;
        if (pfmt) {
            us4 = pexp_unscale(s.pexp[4][(hrow >> 7) * s.pncb[4] + ((8 * lane) >> s.pcbs[4])]);
            if (hl) usl = pexp_unscale(s.pexp[hl][(hrow >> 7) * s.pncb[hl] + (hc8 >> s.pcbs[hl])]);
        }
        float4 h0, h1, g0 = make_float4(0.f, 0.f, 0.f, 0.f), g1 = g0;
        act8(h4b + hrow * 2048, pfmt, 8 * lane, us4, h0, h1);
        if (hl) act8(reinterpret_cast<const unsigned char*>(s.h[hl]) + hrow * (hC * 4L), pfmt, hc8, usl, g0, g1);
        float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0;
        constexpr int EB = HP_EB_GEB;
        for (int i = i0; i < i1; i += EB) {      // (uniform per wave)
            int ch[EB];
            float g[EB];
            float4 w0[EB], w1[EB];
#pragma unroll
            for (int k = 0; k < EB; ++k) ch[k] = chan[min(i + k, i1 - 1)];
#pragma unroll
            for (int k = 0; k < EB; ++k) {
                g[k] = dgb[ch[k]];
                w0[k] = *reinterpret_cast<const float4*>(w5 + (long)ch[k] * 512);
                w1[k] = *reinterpret_cast<const float4*>(w5 + (long)ch[k] * 512 + 4);
            }
#pragma unroll
            for (int k = 0; k < EB; ++k)
                if (i + k < i1) {
                    a0 = f4fma(g[k], w0[k], a0);
                    a1 = f4fma(g[k], w1[k], a1);
                }
        }
        a0 = f4mask(h0, a0);
        a1 = f4mask(h1, a1);
        *reinterpret_cast<float4*>(dst) = a0;
        *reinterpret_cast<float4*>(dst + 4) = a1;
        {   // the row's maximum: the f16 chain's scale (enc_bwd_f16.hip)
            float m = fmaxf(fmaxf(fmaxf(fabsf(a0.x), fabsf(a0.y)), fmaxf(fabsf(a0.z), fabsf(a0.w))),
                            fmaxf(fmaxf(fabsf(a1.x), fabsf(a1.y)), fmaxf(fabsf(a1.z), fabsf(a1.w))));
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
            if (lane == 0) s.d4max[crow] = m;
        }
        if (hl) {
            *reinterpret_cast<float4*>(hdst) = g0;
            *reinterpret_cast<float4*>(hdst + 4) = g1;
        }
        {   // maxima of the row's h3 (lanes 0..31), h2 (32..47), h1 (48..55): the f16 dW launch's block scales
            float m = fmaxf(fmaxf(fmaxf(g0.x, g0.y), fmaxf(g0.z, g0.w)), fmaxf(fmaxf(g1.x, g1.y), fmaxf(g1.z, g1.w)));
            m = fmaxf(m, __shfl_xor(m, 1, 64));
            m = fmaxf(m, __shfl_xor(m, 2, 64));
            m = fmaxf(m, __shfl_xor(m, 4, 64));
            const float m8 = fmaxf(m, __shfl_xor(m, 8, 64));
            m = lane < 48 ? m8 : m;
            const float m16 = fmaxf(m, __shfl_xor(m, 16, 64));
            m = lane < 32 ? m16 : m;
            if (lane == 0 || lane == 32 || lane == 48) s.hmax[crow * 4 + (lane == 0 ? 0 : (lane == 32 ? 1 : 2))] = m;
        }
        // the ReLU masks of the row as bits (lanes 56..63: padding): byte `lane` = the lane's 8 channels
        s.hmask[crow * 64 + lane] = (unsigned char)((g0.x > 0.f) | (g0.y > 0.f) << 1 | (g0.z > 0.f) << 2 | (g0.w > 0.f) << 3 |
                                                    (g1.x > 0.f) << 4 | (g1.y > 0.f) << 5 | (g1.z > 0.f) << 6 | (g1.w > 0.f) << 7);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// block bookkeeping shared by the chain and dW launches
// ---------------------------------------------------------------------------------------------------------------------
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
// chain: delta4 .. delta1 of 32 rows
// ---------------------------------------------------------------------------------------------------------------------
// acc[j] += A(32 x KN, LDS columns kA ..) . W(rows kW .. kW+KN, 32 cols at col0 + 32 j): A rows in LDS (stride LDA, k-permuted
// b128 reads as in gemm.hip: lane half h of k-group t holds k = 8t + 4h + s in step s), W rows (N floats each) from global
// memory — lane (r, h) reads W[k][col0 + 32 j + r]: 128 contiguous bytes per lane half.  32 k per chunk, the next chunk's
// 16*TN loads in flight under the current chunk's MFMAs: branch-free (the last iteration re-reads chunk 0) and pinned in
// front of the MFMA block.
template <int KN, int N, int TN, int LDA>
__device__ __forceinline__ void chain_mfma(const float* As, const float* __restrict__ W, int kA, int kW, int col0, int r, int h,
                                           f32x16 (&acc)[TN]) {
    constexpr int G = 4, NCH = KN / (8 * G);
    static_assert(NCH % 2 == 0, "chunks come in pairs");
    float bw[2][G][4][TN];
    const float* wp = W + (long)(kW + 4 * h) * N + col0 + r;
    auto load = [&](int c, float (&dst)[G][4][TN]) {
#pragma unroll
        for (int g = 0; g < G; ++g)
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int j = 0; j < TN; ++j) dst[g][s][j] = wp[(long)(c * 32 + 8 * g + s) * N + 32 * j];
    };
    auto compute = [&](int c, const float (&src)[G][4][TN]) {
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const float4 av = *reinterpret_cast<const float4*>(&As[r * LDA + kA + c * 32 + 8 * g + 4 * h]);
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(f4at(av, s), src[g][s][j], acc[j], 0, 0, 0);
        }
    };
    load(0, bw[0]);
#pragma unroll 1
    for (int c = 0; c < NCH; c += 2) {
        load(c + 1, bw[1]);
        HP_SB();
        compute(c, bw[0]);
        HP_SB();
        load(c + 2 < NCH ? c + 2 : 0, bw[0]);
        HP_SB();
        compute(c + 1, bw[1]);
        HP_SB();
    }
}

// The same with FOUR interleaved column tiles per wave: lane (r, h) loads the float4 W[k][col0 + 4r .. +3] — tile t covers the
// columns {col0 + 4r + t} — so one 16-byte load feeds four MFMAs and the epilogue moves float4s.  16 k per chunk.
template <int KN, int N, int LDA>
__device__ __forceinline__ void chain_mfma4(const float* As, const float* __restrict__ W, int kA, int kW, int col0, int r, int h,
                                            f32x16 (&acc)[4]) {
    constexpr int G = 2, NCH = KN / (8 * G);
    static_assert(NCH % 2 == 0, "chunks come in pairs");
    float4 bw[2][G][4];
    const float* wp = W + (long)(kW + 4 * h) * N + col0 + 4 * r;
    auto load = [&](int c, float4 (&dst)[G][4]) {
#pragma unroll
        for (int g = 0; g < G; ++g)
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                dst[g][s] = *reinterpret_cast<const float4*>(wp + (long)(c * 16 + 8 * g + s) * N);
            }
    };
    auto compute = [&](int c, const float4 (&src)[G][4]) {
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const float4 av = *reinterpret_cast<const float4*>(&As[r * LDA + kA + c * 16 + 8 * g + 4 * h]);
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const float a1 = f4at(av, s);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, src[g][s].x, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, src[g][s].y, acc[1], 0, 0, 0);
                acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, src[g][s].z, acc[2], 0, 0, 0);
                acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, src[g][s].w, acc[3], 0, 0, 0);
            }
        }
    };
    load(0, bw[0]);
#pragma unroll 1
    for (int c = 0; c < NCH; c += 2) {
        load(c + 1, bw[1]);
        HP_SB();
        compute(c, bw[0]);
        HP_SB();
        load(c + 2 < NCH ? c + 2 : 0, bw[0]);
        HP_SB();
        compute(c + 1, bw[1]);
        HP_SB();
    }
}

// A workgroup is TWO waves and 33 KB of LDS (half a delta4 tile at a time), four of them per CU: the ~750 blocks of a
// HyperPocket step's two encoders are then all resident at once.  (Four waves and the whole 66 KB tile — two workgroups
// per CU — ran the blocks as 512 in lockstep + a second round of ~230: 150 us against the 56 us of its MFMAs.)
constexpr int kChainThreads = 128;

__global__ __launch_bounds__(kChainThreads, 2) void enc_bwd_chain_kernel(const HpEncBwdArgs a) {
    __shared__ __attribute__((aligned(16))) float As[kRows * LD3];
    __shared__ unsigned svalid;
    // block id -> (encoder, cloud, 32-row block): the live blocks of all clouds of both encoders are the FIRST ids, so that
    // consecutive ids — which the dispatcher deals round-robin over the 8 XCDs — are all live.  Every wave finds its block
    // by a 64-lane scan over the clouds' block counts.
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, r = lane & 31, h = lane >> 5;
    int z = 0, b = -1, q = 0, cnt = 0;
    {
        int rel = blockIdx.x;
        for (int e = 0; e < a.n && b < 0; ++e) {
            const int* cn = a.e[e].crit.cnt;
            for (int c0 = 0; c0 < a.B && b < 0; c0 += 64) {
                const int cv = c0 + lane < a.B ? cn[c0 + lane] : 0;
                const int nb = (cv + kRows - 1) / kRows;
                const int inc = wave_scan(nb, lane);
                const int tot = __shfl(inc, 63, 64);
                if (rel < tot) {
                    const unsigned long long m = __ballot(inc > rel);
                    const int l = __ffsll((long long)m) - 1;
                    b = c0 + l;
                    q = rel - (__shfl(inc, l, 64) - __shfl(nb, l, 64));
                    cnt = __shfl(cv, l, 64);
                    z = e;
                } else {
                    rel -= tot;
                }
            }
        }
        if (b < 0) return;
    }
    const HpEncBwdSide& s = a.e[z];
    const long row0 = (long)b * 512 + q * kRows;          // first row of the block in the delta / hc arrays
    const int u0 = q * kRows;
    long long* prof = a.prof ? a.prof + (long)blockIdx.x * 10 : nullptr;
#define HP_STAMP(k) do { if (prof && tid == 0) prof[k] = (long long)wall_clock64(); } while (0)
    HP_STAMP(0);

    // columns [256 half, +256) of the block's delta4 rows (gather launch) -> LDS: thread (rr = tid >> 3, p8 = tid & 7) moves
    // rows rr and rr + 16, eight 16-byte pieces each (8 lanes = 128 contiguous bytes of a row)
    auto stage4 = [&](int half) {
        const int rr = tid >> 3, p8 = tid & 7;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const float* src = s.d[4] + (row0 + rr + 16 * j) * 512 + 256 * half + p8 * 4;
            float4* dst = reinterpret_cast<float4*>(&As[(rr + 16 * j) * LD3 + p8 * 4]);
            const float4 v0 = *reinterpret_cast<const float4*>(src), v1 = *reinterpret_cast<const float4*>(src + 32),
                         v2 = *reinterpret_cast<const float4*>(src + 64), v3 = *reinterpret_cast<const float4*>(src + 96),
                         v4 = *reinterpret_cast<const float4*>(src + 128), v5 = *reinterpret_cast<const float4*>(src + 160),
                         v6 = *reinterpret_cast<const float4*>(src + 192), v7 = *reinterpret_cast<const float4*>(src + 224);
            dst[0] = v0; dst[8] = v1; dst[16] = v2; dst[24] = v3; dst[32] = v4; dst[40] = v5; dst[48] = v6; dst[56] = v7;
        }
    };
    stage4(0);
    if (tid == 0) svalid = cnt - u0 >= 32 ? 0xffffffffu : ((1u << (cnt - u0)) - 1u);
    if (tid < kRows * 3) {   // the rows' coordinates (dW1's operand)
        const int xr = tid / 3, xc = tid - xr * 3;
        s.hc[0][(row0 + xr) * 3 + xc] = u0 + xr < cnt ? s.x[((long)b * a.Np + s.crit.pt[(long)b * 512 + u0 + xr]) * 3 + xc] : 0.f;
    }
    __syncthreads();
    HP_STAMP(1);

    // per lane: bit e of vm: the row of accumulator register e exists (the rows' activations h1..h3 were copied into the compact
    // fp32 rows hc[1..3] by the gather launch, zeros past a cloud's count)
    unsigned vm = 0;
#pragma unroll
    for (int e = 0; e < 16; ++e) vm |= ((svalid >> drow(e, h)) & 1u) << e;

    // ---- delta3 = (delta4 W4) * (h3 > 0)      K = 512 in two staged halves, N = 256: wave w takes the 128 columns
    //      [128 w, +128) as four interleaved tiles
    {
        f32x16 acc[4];
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
        chain_mfma4<256, 256, LD3>(As, s.W[3], 0, 0, 128 * w, r, h, acc);
        __syncthreads();
        stage4(1);
        __syncthreads();
        chain_mfma4<256, 256, LD3>(As, s.W[3], 0, 256, 128 * w, r, h, acc);
        float4 hm[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) hm[e] = *reinterpret_cast<const float4*>(s.hc[3] + (row0 + drow(e, h)) * 256 + 128 * w + 4 * r);
        __syncthreads();   // both waves are done reading delta4
        HP_STAMP(2);
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int row = drow(e, h), col = 128 * w + 4 * r;
            const float4 hv = ((vm >> e) & 1u) ? hm[e] : make_float4(0.f, 0.f, 0.f, 0.f);
            const float4 v = f4mask(hv, make_float4(acc[0][e], acc[1][e], acc[2][e], acc[3][e]));
            *reinterpret_cast<float4*>(&As[row * LD3 + col]) = v;
            *reinterpret_cast<float4*>(s.d[3] + (row0 + row) * 256 + col) = v;
        }
    }
    __syncthreads();
    HP_STAMP(3);

    // ---- delta2 = (delta3 W3) * (h2 > 0)      K = 256, N = 128: both waves take all 128 columns (four interleaved tiles),
    //      wave w the k-half w; the halves are added low + high through LDS
    {
        f32x16 acc[4];
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
        chain_mfma4<128, 128, LD3>(As, s.W[2], 128 * w, 128 * w, 0, r, h, acc);
        float4 hm[16];
        if (w == 0) {
#pragma unroll
            for (int e = 0; e < 16; ++e) hm[e] = *reinterpret_cast<const float4*>(s.hc[2] + (row0 + drow(e, h)) * 128 + 4 * r);
        }
        __syncthreads();
        float* scr = As + kRows * LD2;   // behind delta2: 4 tiles x 16 x 64 floats
        if (w == 1) {
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int e = 0; e < 16; ++e) scr[(t * 16 + e) * 64 + lane] = acc[t][e];
        }
        __syncthreads();
        if (w == 0) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = drow(e, h), col = 4 * r;
                const float4 hv = ((vm >> e) & 1u) ? hm[e] : make_float4(0.f, 0.f, 0.f, 0.f);
                const float4 v = f4mask(hv, make_float4(acc[0][e] + scr[(0 * 16 + e) * 64 + lane], acc[1][e] + scr[(1 * 16 + e) * 64 + lane],
                                                        acc[2][e] + scr[(2 * 16 + e) * 64 + lane], acc[3][e] + scr[(3 * 16 + e) * 64 + lane]));
                *reinterpret_cast<float4*>(&As[row * LD2 + col]) = v;
                *reinterpret_cast<float4*>(s.d[2] + (row0 + row) * 128 + col) = v;
            }
        }
    }
    __syncthreads();
    HP_STAMP(4);

    // ---- delta1 = (delta2 W2) * (h1 > 0)      K = 128, N = 64: wave w takes the column tile [32 w, +32)
    {
        float hm[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) hm[e] = s.hc[1][(row0 + drow(e, h)) * 64 + 32 * w + r];
        f32x16 acc[1];
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[0][e] = 0.f;
        chain_mfma<128, 64, 1, LD2>(As, s.W[1], 0, 0, 32 * w, r, h, acc);
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int row = drow(e, h), col = 32 * w + r;
            const bool ok = (vm >> e) & 1u;
            s.d[1][(row0 + row) * 64 + col] = (ok && hm[e] > 0.f) ? acc[0][e] : 0.f;
        }
    }
    HP_STAMP(5);
    if (prof && tid == 0) prof[8] = 1;
}

// ---------------------------------------------------------------------------------------------------------------------
// dW: the weight / bias gradients of layers 4..1 of the conv stacks as equal tasks of one grid
// ---------------------------------------------------------------------------------------------------------------------
// The rows of an encoder are its 32-row blocks (cloud b has ru32(cnt[b]) / 32 of them, rows b*512 + 32q ..), numbered
// cloud by cloud; range s of S covers blocks [s*T/S, (s+1)*T/S).  A cursor walks a range in 16-row chunks.
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

// One workgroup (4 waves): P(128 x NT at m0, n0) = sum over the range's rows of D[row][m]^T H[row][n]  (D ld M, H ld N;
// NT = 128, or 64 with waves 2, 3 only helping to load).  16 rows per chunk.  The four waves move a chunk's two operand
// tiles global -> registers -> LDS with 16-byte loads (a wave-private float2-per-fragment version issued 4x as many
// vector-memory instructions and ran at half the matrix rate: the memory pipe's issue, not its bandwidth, set the pace);
// the loads of FOUR chunks are in flight (register sets), the LDS image is double-buffered, one barrier per chunk.  Wave
// (wm, wn) owns the 64 x 64 sub-tile at (64 wm, 64 wn) as 2 x 2 interleaved tiles: lane (i, h) reads the float2
// A[k = 2s + h][64 wm + 2i .. +1] of the LDS image — tile t covers columns {.. + 2i + t} — and the epilogue stores float2s.
// Branch-free loads (past the end the last chunk is read again), pinned in front of the MFMA blocks.
template <int NT>
__device__ __forceinline__ void dw_lds_task(const float* __restrict__ D, int M, const float* __restrict__ H, int N, int m0,
                                            int n0, RowCursor cur, int nch, float* __restrict__ P, float* __restrict__ Pdb,
                                            float* sA /* [2][16][128] */, float* sB /* [2][16][NT] */, int tid) {
    constexpr int NBL = NT == 128 ? 2 : 1;    // float4 loads of the H tile per thread and chunk
    const int lane = tid & 63, w = tid >> 6, i = lane & 31, h = lane >> 5;
    const int wm = NT == 128 ? (w >> 1) : w, wn = NT == 128 ? (w & 1) : 0;
    const bool mma = NT == 128 || w < 2;
    f32x16 acc[2][2];
#pragma unroll
    for (int ti = 0; ti < 2; ++ti)
#pragma unroll
        for (int tj = 0; tj < 2; ++tj)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[ti][tj][e] = 0.f;
    float2 dsum = make_float2(0.f, 0.f);
    struct RegSet {                           // one chunk's share of a thread: 2 + NBL 16-byte pieces
        f32x4 a0, a1, b0, b1;
    } r0, r1, r2, r3;
    const int arow = tid >> 5, acol = (tid & 31) * 4;                                   // A tile: rows arow, arow + 8
    const int brow = NT == 128 ? (tid >> 5) : (tid >> 4), bcol = NT == 128 ? (tid & 31) * 4 : (tid & 15) * 4;
    int left = nch - 1;                       // chunks the cursor may still advance
    auto load = [&](RegSet& g) __attribute__((always_inline)) {
        const long row = cur.row();
        const float* dp = D + (row + arow) * M + m0 + acol;
        const float* hp = H + (row + brow) * N + n0 + bcol;
        g.a0 = *reinterpret_cast<const f32x4*>(dp);
        g.a1 = *reinterpret_cast<const f32x4*>(dp + 8L * M);
        g.b0 = *reinterpret_cast<const f32x4*>(hp);
        g.b1 = NBL == 2 ? *reinterpret_cast<const f32x4*>(hp + 8L * N) : g.b0;
        if (left > 0) {                       // else: stay on the last chunk
            --left;
            cur.next();
        }
    };
    auto store = [&](int buf, const RegSet& g) __attribute__((always_inline)) {
        float* pa = sA + buf * 16 * 128 + arow * 128 + acol;
        *reinterpret_cast<f32x4*>(pa) = g.a0;
        *reinterpret_cast<f32x4*>(pa + 8 * 128) = g.a1;
        float* pb = sB + buf * 16 * NT + brow * NT + bcol;
        *reinterpret_cast<f32x4*>(pb) = g.b0;
        if (NBL == 2) *reinterpret_cast<f32x4*>(pb + 8 * NT) = g.b1;
    };
    auto compute = [&](int buf) __attribute__((always_inline)) {
        const float* pa = sA + buf * 16 * 128 + h * 128 + 64 * wm + 2 * i;
        const float* pb = sB + buf * 16 * NT + h * NT + 64 * wn + 2 * i;
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const float2 av = *reinterpret_cast<const float2*>(pa + 2 * s * 128);
            const float2 bv = *reinterpret_cast<const float2*>(pb + 2 * s * NT);
            dsum.x += av.x;
            dsum.y += av.y;
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bv.x, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bv.y, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bv.x, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bv.y, acc[1][1], 0, 0, 0);
        }
    };
    load(r0);
    load(r1);
    load(r2);
    load(r3);
    store(0, r0);
    __syncthreads();
    // chunk it + K: its image is LDS buffer K & 1 and register set K is free for chunk it + K + 4; then chunk it + K + 1 goes
    // from its register set to the other LDS buffer
#define HP_DW_STEP(K, RK, RN)                                    \
    load(RK);                                                    \
    HP_SB();                                                     \
    if (mma && it + K < nch) compute(K & 1);                     \
    HP_SB();                                                     \
    store((K + 1) & 1, RN);                                      \
    __syncthreads();
#pragma unroll 1
    for (int it = 0; it < nch; it += 4) {
        HP_DW_STEP(0, r0, r1)
        HP_DW_STEP(1, r1, r2)
        HP_DW_STEP(2, r2, r3)
        HP_DW_STEP(3, r3, r0)
    }
#undef HP_DW_STEP
    if (!mma) return;
    // lane (i, h), register e of tile (ti, tj): P[m0 + 64 wm + 2*drow(e,h) + ti][n0 + 64 wn + 2*i + tj]
#pragma unroll
    for (int ti = 0; ti < 2; ++ti)
#pragma unroll
        for (int e = 0; e < 16; ++e)
            *reinterpret_cast<float2*>(P + (long)(m0 + 64 * wm + 2 * drow(e, h) + ti) * N + n0 + 64 * wn + 2 * i) =
                make_float2(acc[ti][0][e], acc[ti][1][e]);
    if (Pdb && n0 == 0 && wn == 0) {   // bias gradient = column sums of D: even rows in lanes 0-31, odd rows in lanes 32-63
        const float ox = __shfl_xor(dsum.x, 32, 64), oy = __shfl_xor(dsum.y, 32, 64);
        if (h == 0) *reinterpret_cast<float2*>(Pdb + m0 + 64 * wm + 2 * i) = make_float2(dsum.x + ox, dsum.y + oy);
    }
}

// workgroups per range: dW4 (512 x 256) 8 tiles of 128 x 128, dW3 (256 x 128) 2, and one with dW2's (128 x 64) tile
// followed by dW1 (64 x 3) + db1
constexpr int kRangeWgs = 8 + 2 + 1;
constexpr int kMaxClouds = 2048;     // LDS table of block prefixes

__global__ __launch_bounds__(256, 2) void enc_bwd_dw_kernel(const HpEncBwdArgs a) {
    __shared__ __attribute__((aligned(16))) float sA[2 * 16 * 128];
    __shared__ __attribute__((aligned(16))) float sB[2 * 16 * 128];
    __shared__ int pre[kMaxClouds + 1];
    __shared__ float red1[3][64][4];
    const int id = blockIdx.x;
    const int z = a.n == 2 ? (id & 1) : 0, rest = a.n == 2 ? (id >> 1) : id;
    const HpEncBwdSide& s = a.e[z];
    const int S = a.S, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    Stamp stamp(a.prof ? a.prof + (long)id * 4 : nullptr, tid);
    const int split = rest % S, t = rest / S;      // (the tasks of one kind are neighbours in the dispatch order)
    stamp.type = t < 8 ? 1 : (t < 10 ? 2 : 3);
    // block prefixes of the clouds (wave 0: 64-lane scans, carried over the chunks of 64 clouds)
    if (w == 0) {
        int carry = 0;
        for (int c0 = 0; c0 < a.B; c0 += 64) {
            const int nb = c0 + lane < a.B ? ru32(s.crit.cnt[c0 + lane]) >> 5 : 0;
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
    if (t < 8) {            // dW4: tile (m = t >> 1, n = t & 1)
        dw_lds_task<128>(s.d[4], 512, s.hc[3], 256, 128 * (t >> 1), 128 * (t & 1), cur, nch, P + oW4, P + oB4, sA, sB, tid);
    } else if (t < 10) {    // dW3: tiles m = t - 8
        dw_lds_task<128>(s.d[3], 256, s.hc[2], 128, 128 * (t - 8), 0, cur, nch, P + oW3, P + oB3, sA, sB, tid);
    } else {                // dW2: one tile of 128 x 64
        dw_lds_task<64>(s.d[2], 128, s.hc[1], 64, 0, 0, cur, nch, P + oW2, P + oB2, sA, sB, tid);
        // dW1 (64 x 3) + db1 behind it: channel c = lane, wave w the rows = w mod 4; the four row classes are added in order
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
        if (w) {
            red1[w - 1][lane][0] = ax;
            red1[w - 1][lane][1] = ay;
            red1[w - 1][lane][2] = az;
            red1[w - 1][lane][3] = ab;
        }
        __syncthreads();
        if (w == 0) {
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                ax += red1[q][lane][0];
                ay += red1[q][lane][1];
                az += red1[q][lane][2];
                ab += red1[q][lane][3];
            }
            P[oW1 + lane * 3 + 0] = ax;
            P[oW1 + lane * 3 + 1] = ay;
            P[oW1 + lane * 3 + 2] = az;
            P[oB1 + lane] = ab;
        }
    }
}

// out = sum over the ranges (ascending) of their partial sums; 4 consecutive floats per thread
__global__ __launch_bounds__(256) void enc_bwd_reduce_kernel(const HpEncBwdArgs a) {
    const HpEncBwdSide& s = a.e[blockIdx.y];
    const int i = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= HP_EB_PART_FLOATS) return;
    float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int k0 = 0; k0 < a.S; k0 += 16) {     // 16 ranges' loads in flight, added in range order
        float4 v[16];
#pragma unroll
        for (int k = 0; k < 16; ++k)
            if (k0 + k < a.S) v[k] = *reinterpret_cast<const float4*>(s.part + (long)(k0 + k) * HP_EB_PART_FLOATS + i);
#pragma unroll
        for (int k = 0; k < 16; ++k)
            if (k0 + k < a.S) o = make_float4(o.x + v[k].x, o.y + v[k].y, o.z + v[k].z, o.w + v[k].w);
    }
    float* dst;
    if (i < oW3) dst = s.gW[3] + i;
    else if (i < oW2) dst = s.gW[2] + (i - oW3);
    else if (i < oW1) dst = s.gW[1] + (i - oW2);
    else if (i < oB4) dst = s.gW[0] + (i - oW1);
    else if (i < oB3) dst = s.gb[3] ? s.gb[3] + (i - oB4) : nullptr;
    else if (i < oB2) dst = s.gb[2] ? s.gb[2] + (i - oB3) : nullptr;
    else if (i < oB1) dst = s.gb[1] ? s.gb[1] + (i - oB2) : nullptr;
    else dst = s.gb[0] ? s.gb[0] + (i - oB1) : nullptr;
    if (dst) *reinterpret_cast<float4*>(dst) = o;
}

// ---- HP_EB_PROF: per-launch summaries of the in-kernel stamps (debug; synchronises) ----------------------------------
long long* prof_buffer() {
    static long long* buf = nullptr;
    if (!buf) (void)hipMalloc(&buf, sizeof(long long) * 10 * 262144);
    return buf;
}
void prof_tasks(const char* name, long n, int types, const char* const* nm, hipStream_t stream) {
    (void)hipStreamSynchronize(stream);
    std::vector<long long> hb(4 * n);
    (void)hipMemcpy(hb.data(), prof_buffer(), sizeof(long long) * 4 * n, hipMemcpyDeviceToHost);
    long long tmin = -1;
    for (long i = 0; i < n; ++i)
        if (hb[i * 4 + 2] && (tmin < 0 || hb[i * 4] < tmin)) tmin = hb[i * 4];
    fprintf(stderr, "[%s prof] type: n avg(max) last-end us |", name);
    for (int ty = 1; ty <= types; ++ty) {
        double sum = 0, mx = 0, last = 0;
        long cnt = 0;
        for (long i = 0; i < n; ++i) {
            if (hb[i * 4 + 2] != ty) continue;
            const double d = (double)(hb[i * 4 + 1] - hb[i * 4]) * 0.01;
            sum += d;
            ++cnt;
            mx = std::max(mx, d);
            last = std::max(last, (double)(hb[i * 4 + 1] - tmin) * 0.01);
        }
        fprintf(stderr, " %s: %ld %.1f(%.1f) %.1f |", nm[ty], cnt, cnt ? sum / cnt : 0.0, mx, last);
    }
    fprintf(stderr, "\n");
}

}  // namespace

int hp_enc_bwd_prep(const HpEncBwdArgs* a, hipStream_t stream) {
    const int extra = hp_enc_bwd_chain_f16_enabled() ? hp_wprep::kTasks : 0;
    hipLaunchKernelGGL(enc_bwd_prep_kernel, dim3(a->B + extra, a->n), dim3(512), 0, stream, *a);
    HP_RETURN_LAST_ERROR();
}

int hp_enc_bwd_max_clouds() { return kMaxClouds; }
bool hp_enc_bwd_dw_f16_enabled() {      // HP_EB_DW16 (default on): the dW launch on the f16 pipe (needs the f16 chain's block exponents)
    static const bool on = [] {
        const char* e = getenv("HP_EB_DW16");
        return !(e && e[0] == '0');
    }();
    return on;
}

int hp_enc_bwd_conv(const HpEncBwdArgs* a0, hipStream_t stream) {
    HpEncBwdArgs args = *a0;
    HpEncBwdArgs* a = &args;
    static const bool prof_on = getenv("HP_EB_PROF") != nullptr;
    const long ngat = (long)(512 + kGatherRowWgs) * a->n, nblk = (long)a->B * 16 * a->n, ndw = (long)a->S * kRangeWgs * a->n;
    auto arm = [&](long n, int per) {
        if (!prof_on) return;
        (void)hipMemsetAsync(prof_buffer(), 0, sizeof(long long) * per * n, stream);
        a->prof = prof_buffer();
    };
    arm(ngat, 4);
    hipLaunchKernelGGL(enc_bwd_gather_kernel, dim3((unsigned)ngat), dim3(256), 0, stream, *a);
    if (prof_on) {
        static const char* const nm[] = {"", "dW5", "delta4-rows"};
        prof_tasks("gather", ngat, 2, nm, stream);
    }
    const bool chain16 = hp_enc_bwd_chain_f16_enabled() && !prof_on;
    if (chain16) {
        const int rc = hp_enc_bwd_chain_f16(a, stream);
        if (rc) return rc;
    } else {
        arm(nblk, 10);
        hipLaunchKernelGGL(enc_bwd_chain_kernel, dim3((unsigned)nblk), dim3(kChainThreads), 0, stream, *a);
    }
    if (prof_on) {
        (void)hipStreamSynchronize(stream);
        std::vector<long long> hb(10 * nblk);
        (void)hipMemcpy(hb.data(), prof_buffer(), sizeof(long long) * 10 * nblk, hipMemcpyDeviceToHost);
        double sum[5] = {0}, mx[5] = {0};
        long live = 0;
        long long tmin = -1, tmax = 0;
        for (long i = 0; i < nblk; ++i) {
            const long long* t = &hb[i * 10];
            if (!t[8]) continue;
            ++live;
            if (tmin < 0 || t[0] < tmin) tmin = t[0];
            tmax = std::max(tmax, t[5]);
            for (int k = 0; k < 5; ++k) {
                const double d = (double)(t[k + 1] - t[k]) * 0.01;   // 100 MHz -> us
                sum[k] += d;
                mx[k] = std::max(mx[k], d);
            }
        }
        int hs[32] = {0}, he[32] = {0};
        for (long i = 0; i < nblk; ++i) {
            const long long* t = &hb[i * 10];
            if (!t[8]) continue;
            hs[std::min<long long>(31, (t[0] - tmin) / 1000)]++;
            he[std::min<long long>(31, (t[5] - tmin) / 1000)]++;
        }
        fprintf(stderr, "[chain prof] starts per 10 us:");
        for (int k = 0; k < 20; ++k) fprintf(stderr, " %d", hs[k]);
        fprintf(stderr, "\n[chain prof] ends   per 10 us:");
        for (int k = 0; k < 20; ++k) fprintf(stderr, " %d", he[k]);
        fprintf(stderr, "\n");
        fprintf(stderr, "[chain prof] live %ld span %.1f us | avg(max) us: stage %.1f(%.1f) L4-mfma %.1f(%.1f) L4-epi %.1f(%.1f) L3 %.1f(%.1f) "
                "L2 %.1f(%.1f)\n", live, (double)(tmax - tmin) * 0.01, sum[0] / live, mx[0], sum[1] / live, mx[1], sum[2] / live, mx[2],
                sum[3] / live, mx[3], sum[4] / live, mx[4]);
    }
    if (chain16 && hp_enc_bwd_dw_f16_enabled()) {
        const int rc = hp_enc_bwd_dw_f16(a, stream);
        if (rc) return rc;
    } else {
        arm(ndw, 4);
        hipLaunchKernelGGL(enc_bwd_dw_kernel, dim3((unsigned)ndw), dim3(256), 0, stream, *a);
    }
    if (prof_on) {
        static const char* const nm[] = {"", "dW4", "dW3", "dW2+dW1"};
        prof_tasks("dw", ndw, 3, nm, stream);
    }
    a->prof = nullptr;
    hipLaunchKernelGGL(enc_bwd_reduce_kernel, dim3((HP_EB_PART_FLOATS / 4 + 255) / 256, a->n), dim3(256), 0, stream, *a);
    HP_RETURN_LAST_ERROR();
}
