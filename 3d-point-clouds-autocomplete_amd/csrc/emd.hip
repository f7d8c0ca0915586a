// Approximate earth-mover matching for gfx950 (replaces approxmatch.cu:34-213 + the match-based
// cost/gradient kernels :215-322 of /root/reference/utils/pytorch_structural_losses/).
//
// Algorithm: 9 annealing levels, each with three globally dependent phases.
// MI355X design:
//   * every phase is a chip-wide launch over (cloud, row block); one lane owns one row point and
//     sweeps ALL candidates of the other set.  Candidates are wave-uniform, so they travel on the
//     SCALAR path: packed records (xyz + weights) are fetched with s_load_dwordx8/x16 and used as
//     SGPR operands of the VALU ops — no LDS tile, no barrier, no candidate VGPRs.  Only ~2 waves
//     per SIMD exist (one lane per row point), so the SMEM latency is hidden by an explicit
//     two-stage software pipeline: the next stage's s_loads are issued (inline asm, pinned with
//     sched_barrier) before the VALU work on the current stage, and waited for after it.
//   * phase 3 of level j and phase 1 of level j+1 own the same rows and need the same candidate set,
//     so they are ONE launch (distance evaluated once, two exponentials): 19 launches instead of 27.
//   * per-level scaling vectors ratioL/ratioR are kept in packed "final" records; `match` (API) is
//     produced by one pass that re-evaluates the nine exponentials per pair in level order (the same
//     summation order as the reference's nine `match +=` passes) — or never: the fused cost/gradient
//     kernels consume the records directly (match-free EMD), so the (b,m,n) tensor and
//     its 19 GB of read-modify-write traffic at B=64 disappear from the training step.
//   * candidates are stored as PAIR records ([x0 x1 | y0 y1 | z0 z1 | w0 w1 ...]) so that the distance and
//     weighting arithmetic of two candidates runs on packed fp32 VALU ops (v_pk_add/mul/fma_f32 with an SGPR
//     pair as one operand) — the non-packed VALU rate is only half of the 157 TFLOP/s vector peak.  Each
//     element still sees the reference's operation sequence as nvcc's default -fmad=true contracts it (distance
//     fma chain, products entering the running sums through an fma, ascending candidate order in the
//     accumulators): the C oracle's `contract` variant 3.
//   * padding records carry zero weights and contribute exact zeros.
//   * round 5: (a) the match-free cost / gradient sweep derives four of its nine per-level exponentials as fourth powers of their
//     neighbours (the levels are exact powers of 4 apart; only there — nothing is downstream of those values; match_entry2);
//     (b) the clouds are independent but a launch is not: every one of the 19 dependent launches pays ramp, prologue, epilogue
//     and tail with the whole chip in lockstep, so hp_emd_forward* runs the level sweeps as TWO chains of half the clouds on two
//     streams, enqueued alternately, and one chain's waves cover the other's launch boundaries (LevelChain, emd_forward_impl).
#include "hp_common.h"
#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <map>
#include <mutex>
#include <vector>

// (Round 3's derived-exponential experiment — e(j) = e(j+1)^4, -0.08 ms per step, fails the parity bars — lives as a patch in
// tools/micro/emd_derive.patch, not in the shipped library: docs/DESIGN_HISTORY.md 7b.)

namespace {

#ifndef HP_EMD_PARTS
#define HP_EMD_PARTS 4
#endif
constexpr int kThreads = 64 * HP_EMD_PARTS;   // one wave per candidate range, 64 rows per workgroup
constexpr int kLevels = 9;
constexpr int kStage = 8;   // candidates per software-pipeline stage of the phase kernels (= 4 pair records)
constexpr int kSpare = 8;   // readable zero candidates past the padded range (the prefetch after the last stage)
constexpr float kLog2e = 1.4426950408889634f;

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x8 __attribute__((ext_vector_type(8)));
typedef float f2 __attribute__((ext_vector_type(2)));

// Scalar-memory loads the compiler must neither wait for early nor sink (cdna_hip_programming.md §5.7):
// issue -> sched_barrier -> VALU work on the other stage -> one s_waitcnt that names every destination.
#define HP_SLOAD16(dst, ptr, off) asm volatile("s_load_dwordx16 %0, %1, " #off : "=s"(dst) : "s"(ptr) : "memory")
#define HP_SLOAD8(dst, ptr, off) asm volatile("s_load_dwordx8 %0, %1, " #off : "=s"(dst) : "s"(ptr) : "memory")
#define HP_PIN() __builtin_amdgcn_sched_barrier(0)

__host__ __device__ constexpr float level_l2e(int lev) {
    // level = -4^j, j = 7..-1 (approxmatch.cu:55-56), pre-multiplied by log2(e); powers of 4 scale exactly
    return (lev == 0 ? -16384.f : lev == 1 ? -4096.f : lev == 2 ? -1024.f : lev == 3 ? -256.f : lev == 4 ? -64.f
            : lev == 5 ? -16.f : lev == 6 ? -4.f : lev == 7 ? -1.f : -0.25f) * kLog2e;
}

// Only B*N row points exist (2 waves per SIMD at B=64, N=2048): to fill the SIMDs each row's candidate sweep is cut
// into kParts contiguous ranges handled by different waves of the workgroup and added in range order through LDS
// (ordered, deterministic; a reordering of the reference's sequential sum that moves the cost by ~1e-7 relative —
// measured on the oracle — against 1e-5 for the exp formulation).
constexpr int kParts = HP_EMD_PARTS;
constexpr int kRowsPerWg = kThreads / kParts;
inline int pad_up(int x) { return (x + 2 * kStage * kParts - 1) / (2 * kStage * kParts) * (2 * kStage * kParts); }

// per-cloud workspace (floats), every candidate array padded to NP/MP (+kSpare) entries:
//   PLP  pair records of set1 for phase 2:        8 floats per PAIR  [x0 x1 y0 y1 z0 z1 ratioL0 ratioL1]
//   PRP  pair records of set2 for phases 1/3:     8 floats per PAIR  [x0 x1 y0 y1 z0 z1 ratioR0 ratioR1]
//   RR   remainR per candidate of set2 (linear, so pairs are adjacent)
//   FLP / FRP  "final" pair records (32 floats per PAIR): [x0 x1 y0 y1 z0 z1 | r(lev0)0 r(lev0)1 | ... | r(lev8)0 r(lev8)1 | 8 pad]
struct WsLayout {
    int NP, MP;
    long plp, prp, rr, flp, frp, permL, permR, blkL, blkR, tileL, tileR, flag, per_cloud;
};
// Round 6 (ordered / culling sweeps): per set the k-d order (position -> original index, int32 in a float slot), the bounding
// boxes of its 8-candidate blocks (SoA: [minx | miny | minz | maxx | maxy | maxz] x NP/8) and of its 64-row tiles (x NP/64), and
// one flag block per cloud (flag[0] != 0: the records are in k-d order).
constexpr int kBlk = 8;      // candidates per cull block = one software-pipeline stage of pair records
constexpr int kTile = 64;    // rows per cull tile = the rows one wave owns per row slot
inline WsLayout ws_layout(int n, int m) {
    WsLayout w;
    w.NP = pad_up(n);
    w.MP = pad_up(m);
    w.plp = 0;
    w.prp = w.plp + (long)(w.NP + kSpare) * 4;
    w.rr = w.prp + (long)(w.MP + kSpare) * 4;
    w.flp = w.rr + (long)(w.MP + kSpare);
    w.frp = w.flp + (long)(w.NP + kSpare) * 16;
    w.permL = w.frp + (long)(w.MP + kSpare) * 16;
    w.permR = w.permL + w.NP;
    w.blkL = w.permR + w.MP;
    w.blkR = w.blkL + 6L * (w.NP / kBlk);
    w.tileL = w.blkR + 6L * (w.MP / kBlk);
    w.tileR = w.tileL + 6L * (w.NP / kTile);
    w.flag = w.tileR + 6L * (w.MP / kTile);
    w.per_cloud = w.flag + 16;
    return w;
}

// forced rows-per-lane of the three sweep families (0 = the size heuristic); see hp_emd_set_rows_per_lane
inline int env_rows(const char* name) {
    const char* e = getenv(name);
    const int v = e ? atoi(e) : 0;
    return (v == 1 || v == 2 || v == 4) ? v : 0;
}
std::mutex g_s2_mu;                            // the library's second-chain stream per device (emd_forward_impl)
std::map<int, hipStream_t> g_s2_streams;
// hp_emd_forward* as two chains of half the clouds on two streams (2) or one chain (1): emd_forward_impl
constexpr int kMaxChains = 4;
std::atomic<int> g_chains{[] { const char* e = getenv("HP_EMD_CHAINS"); const int v = e ? atoi(e) : 2; return v < 1 ? 1 : v > kMaxChains ? kMaxChains : v; }()};
// the final cost / gradient sweep with derived exponentials (match_entry2<.., DERIVE>): on unless HP_EMD_FINAL_DERIVE=0
std::atomic<int> g_final_derive{[] { const char* e = getenv("HP_EMD_FINAL_DERIVE"); return (e && atoi(e) == 0) ? 0 : 1; }()};
// hp_emd_forward*: records in k-d order and the first g_cull levels' sweeps culling (0: caller's order, no culling): hp_emd_set_cull
constexpr int kCullDefault = 3;
std::atomic<int> g_cull{[] { const char* e = getenv("HP_EMD_CULL"); const int v = e ? atoi(e) : kCullDefault; return v < 0 ? 0 : v > kLevels ? kLevels : v; }()};
constexpr int kOrderMaxLog = 12;   // k-d order for sets of up to 4096 points (the order kernel's LDS: 19 bytes per point)
std::atomic<int> g_rows1{env_rows("HP_EMD_ROWS1_R")}, g_rows2{env_rows("HP_EMD_ROWS2_R")}, g_grad2{env_rows("HP_EMD_GRAD2_R") == 4 ? 0 : env_rows("HP_EMD_GRAD2_R")};

struct Ctx {
    int n, m, NP, MP;
    const float* xyz1;
    const float* xyz2;
    float* temp;  // (b, 2(n+m)) : [remainL n | remainR m | ratioL n | ratioR m]  (reference layout, approxmatch.cu:35)
    float* ws;
    int plp, prp, rr, flp, frp, permL, permR, blkL, blkR, tileL, tileR, flag;   // float offsets inside a cloud's workspace (< 2^31)
    long per_cloud;
    float acc_scale = 0.f;   // != 0: emd_grad2_kernel stores grad2[i] += acc_scale * d cost / d xyz2[i] instead of the plain gradient
};
inline Ctx make_ctx(int n, int m, const float* xyz1, const float* xyz2, float* temp, float* ws) {
    const WsLayout L = ws_layout(n, m);
    Ctx c;
    c.n = n; c.m = m; c.NP = L.NP; c.MP = L.MP;
    c.xyz1 = xyz1; c.xyz2 = xyz2; c.temp = temp; c.ws = ws;
    c.plp = (int)L.plp; c.prp = (int)L.prp; c.rr = (int)L.rr; c.flp = (int)L.flp; c.frp = (int)L.frp;
    c.permL = (int)L.permL; c.permR = (int)L.permR; c.blkL = (int)L.blkL; c.blkR = (int)L.blkR; c.tileL = (int)L.tileL; c.tileR = (int)L.tileR;
    c.flag = (int)L.flag; c.per_cloud = L.per_cloud;
    c.acc_scale = 0.f;
    return c;
}

// element offsets of candidate i inside the pair-record arrays
__device__ __forceinline__ long pair8(int i, int comp) { return (long)(i >> 1) * 8 + comp * 2 + (i & 1); }            // comp 0..3 = x,y,z,w
__device__ __forceinline__ long pair32(int i, int comp) { return (long)(i >> 1) * 32 + comp * 2 + (i & 1); }          // comp 0..2 = x,y,z ; 3+lev = ratio

// One thread per PAIR of points: its 8-float sweep record and 32-float final record leave as 16-byte stores (the first
// version wrote 20 scattered dwords per point: 17 us for 26 MB).
__global__ __launch_bounds__(256) void emd_init_kernel(Ctx c, float multiL, float multiR) {
    const int cloud = blockIdx.y;
    float* ws = c.ws + (long)cloud * c.per_cloud;
    float* remL = c.temp + (long)cloud * (c.n + c.m) * 2;
    float* remR = remL + c.n;
    const float* P = c.xyz1 + (long)cloud * c.n * 3;
    const float* Q = c.xyz2 + (long)cloud * c.m * 3;
    const int NPp = (c.NP + kSpare) / 2, MPp = (c.MP + kSpare) / 2;   // pairs per set (NP, MP, kSpare are even)
    if (blockIdx.x == 0 && threadIdx.x == 0) ws[c.flag] = 0.f;        // records in the caller's point order (see emd_order_kernel)
    for (int i = blockIdx.x * 256 + threadIdx.x; i < NPp + MPp; i += gridDim.x * 256) {
        const bool left = i < NPp;
        const int pr = left ? i : i - NPp, j0 = 2 * pr, j1 = j0 + 1;
        const int cnt = left ? c.n : c.m;
        const float* src = left ? P : Q;
        const bool ok0 = j0 < cnt, ok1 = j1 < cnt;
        const float x0 = ok0 ? src[j0 * 3] : 0.f, y0 = ok0 ? src[j0 * 3 + 1] : 0.f, z0 = ok0 ? src[j0 * 3 + 2] : 0.f;
        const float x1 = ok1 ? src[j1 * 3] : 0.f, y1 = ok1 ? src[j1 * 3 + 1] : 0.f, z1 = ok1 ? src[j1 * 3 + 2] : 0.f;
        float4* p8 = reinterpret_cast<float4*>(ws + (left ? c.plp : c.prp) + (long)pr * 8);
        float4* p32 = reinterpret_cast<float4*>(ws + (left ? c.flp : c.frp) + (long)pr * 32);
        const float4 a = make_float4(x0, x1, y0, y1), b = make_float4(z0, z1, 0.f, 0.f), zero = make_float4(0.f, 0.f, 0.f, 0.f);
        p8[0] = a;
        p8[1] = b;
        p32[0] = a;
        p32[1] = b;
#pragma unroll
        for (int q = 2; q < 8; ++q) p32[q] = zero;
        if (left) {
            if (ok0) remL[j0] = multiL;
            if (ok1) remL[j1] = multiL;
        } else {
            *reinterpret_cast<float2*>(ws + c.rr + j0) = make_float2(ok0 ? multiR : 0.f, ok1 ? multiR : 0.f);
            if (ok0) remR[j0] = multiR;
            if (ok1) remR[j1] = multiR;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Round 6: the records in HILBERT ORDER, so that the level sweeps can skip work that is exactly zero.
// At level -16384 / -4096 / -1024 / -256 the exponential of a pair further apart than 0.080 / 0.160 / 0.321 / 0.641 underflows to
// exactly +0 (exp2 of less than -152): its terms are exact zeros in every phase (approxmatch.cu:86-87,131-132,185-189) and a sweep
// that leaves them out produces the same sums.  Skipping is only possible for whole (wave, pipeline stage) units, so both sets are
// put in an order in which 64 consecutive rows and 8 consecutive candidates are spatially compact: along the Hilbert curve of a
// 16^3 grid over the set's bounding box — consecutive cells of that curve are always neighbours, so ANY run of consecutive points
// is compact, not only the aligned ones (Morton order jumps).  Decidable from the bounding boxes of a (64-row tile, 8-candidate
// block) unit at the first four levels, uniform clouds: 80 / 70 / 47 / 6 % of the units (Morton 69 / 59 / 33 / 3 %; a k-d order —
// median splits, 85 / 76 / 54 / 10 % — needs eight segment sorts instead of one and costs more than it saves:
// tools/study/emd_cull_hilbert.py).  The algorithm is indifferent to the order of either set; a sweep in another
// order is another summation order of the same sums (oracle on re-ordered inputs: cost within 3e-7,
// tools/study/emd_order_sensitivity.py), and the gradient sweeps write through the permutation, so callers see their own order.
//
// One workgroup per (cloud, set): a counting sort by Hilbert cell in LDS (histogram, scan, scatter; the points of a cell in index
// order, so the order is a pure function of the input).  Then the same workgroup writes everything emd_init_kernel writes, in the
// new order, plus the permutation and the block / tile bounding boxes.
// ------------------------------------------------------------------------------------------------
constexpr int kOrderThreads = 1024;   // 16 waves for the load / key / output phases; the sort network runs on the first P2/8 threads
constexpr float kFar = 1e18f;     // an empty box: min = +kFar, max = -kFar (its gap to anything squares to 1e36 > any radius)

// Hilbert index of the cell (x, y, z), `bits` bits per axis (Skilling's transpose form, then interleaved: 3 * bits bits)
__device__ __forceinline__ uint32_t spread3(uint32_t v) {      // bit b -> bit 3b (b < 10)
    v = (v | (v << 16)) & 0x030000FFu;
    v = (v | (v << 8)) & 0x0300F00Fu;
    v = (v | (v << 4)) & 0x030C30C3u;
    v = (v | (v << 2)) & 0x09249249u;
    return v;
}
__device__ __forceinline__ uint32_t hilbert3(uint32_t x0, uint32_t x1, uint32_t x2, int bits) {
    const uint32_t M = 1u << (bits - 1);
    for (uint32_t Q = M; Q > 1; Q >>= 1) {
        const uint32_t P = Q - 1;
        if (x0 & Q) x0 ^= P;
        if (x1 & Q) x0 ^= P;
        else {
            const uint32_t t = (x0 ^ x1) & P;
            x0 ^= t;
            x1 ^= t;
        }
        if (x2 & Q) x0 ^= P;
        else {
            const uint32_t t = (x0 ^ x2) & P;
            x0 ^= t;
            x2 ^= t;
        }
    }
    x1 ^= x0;
    x2 ^= x1;
    uint32_t t = 0;
    for (uint32_t Q = M; Q > 1; Q >>= 1)
        if (x2 & Q) t ^= Q - 1;
    x0 ^= t;
    x1 ^= t;
    x2 ^= t;
    return (spread3(x0) << 2) | (spread3(x1) << 1) | spread3(x2);
}
constexpr int kHilbertBits = 4;                        // 16^3 cells: finer grids do not make 8-point runs more compact (emd_cull_hilbert.py)
constexpr int kCells = 1 << (3 * kHilbertBits);

__global__ __launch_bounds__(kOrderThreads) void emd_order_kernel(Ctx c, float multiL, float multiR, int logpL, int logpR) {
    extern __shared__ float smem[];
    __shared__ float red[6][kOrderThreads / 64];
    __shared__ float gb[3], gscale[3];
    __shared__ uint32_t wsum[kOrderThreads / 64];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const bool left = blockIdx.x == 0;
    const int cloud = blockIdx.y;
    const int cnt = left ? c.n : c.m, NPx = left ? c.NP : c.MP, logp = left ? logpL : logpR, P2 = 1 << logp;
    const float* src = (left ? c.xyz1 + (long)cloud * c.n * 3 : c.xyz2 + (long)cloud * c.m * 3);
    float* ws = c.ws + (long)cloud * c.per_cloud;
    float* X = smem;
    float* Y = X + P2;
    float* Z = Y + P2;
    uint32_t* hist = reinterpret_cast<uint32_t*>(Z + P2);            // per cell: count -> start -> end of its slot range
    uint16_t* cell = reinterpret_cast<uint16_t*>(hist + kCells);     // per point
    uint16_t* tmp = cell + P2;                                       // slot -> point, arrival order inside a cell
    uint16_t* key = tmp + P2;                                        // position -> point (the order)
    float* box0 = reinterpret_cast<float*>(hist);   // NP/8 block boxes [lo xyz | hi xyz] (the output phase; the histogram is dead by then)

    // the points and their bounding box
    float mn[3] = {kFar, kFar, kFar}, mx[3] = {-kFar, -kFar, -kFar};
    for (int i = tid; i < P2; i += kOrderThreads) {
        float x = 0.f, y = 0.f, z = 0.f;
        if (i < cnt) {
            x = src[i * 3];
            y = src[i * 3 + 1];
            z = src[i * 3 + 2];
            mn[0] = fminf(mn[0], x); mx[0] = fmaxf(mx[0], x);
            mn[1] = fminf(mn[1], y); mx[1] = fmaxf(mx[1], y);
            mn[2] = fminf(mn[2], z); mx[2] = fmaxf(mx[2], z);
        }
        X[i] = x;
        Y[i] = y;
        Z[i] = z;
        key[i] = (uint16_t)i;                       // positions past the count keep the identity
    }
    for (int q = tid; q < kCells; q += kOrderThreads) hist[q] = 0u;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            mn[a] = fminf(mn[a], __shfl_xor(mn[a], o, 64));
            mx[a] = fmaxf(mx[a], __shfl_xor(mx[a], o, 64));
        }
        if (lane == 0) {
            red[a][wid] = mn[a];
            red[3 + a][wid] = mx[a];
        }
    }
    __syncthreads();
    if (tid < 3) {
        float lo = red[tid][0], hi = red[3 + tid][0];
        for (int w = 1; w < kOrderThreads / 64; ++w) {
            lo = fminf(lo, red[tid][w]);
            hi = fmaxf(hi, red[3 + tid][w]);
        }
        gb[tid] = lo;
        gscale[tid] = hi > lo ? (float)(1 << kHilbertBits) * 0.9999f / (hi - lo) : 0.f;
    }
    __syncthreads();
    // A counting sort by Hilbert cell, points of a cell in index order (deterministic: the counts do not depend on the order
    // the atomics execute in, and the arrival order inside a cell is replaced by the index order below).
    for (int i = tid; i < cnt; i += kOrderThreads) {
        const uint32_t cm = (1u << kHilbertBits) - 1u;
        const uint32_t qx = min((uint32_t)((X[i] - gb[0]) * gscale[0]), cm), qy = min((uint32_t)((Y[i] - gb[1]) * gscale[1]), cm),
                       qz = min((uint32_t)((Z[i] - gb[2]) * gscale[2]), cm);
        const uint32_t h = hilbert3(qx, qy, qz, kHilbertBits);
        cell[i] = (uint16_t)h;
        atomicAdd(&hist[h], 1u);
    }
    __syncthreads();
    {   // exclusive scan of the cell counts: kCells / kOrderThreads consecutive cells per thread
        constexpr int kPer = kCells / kOrderThreads;
        uint32_t v[kPer], s = 0;
#pragma unroll
        for (int q = 0; q < kPer; ++q) {
            v[q] = hist[tid * kPer + q];
            s += v[q];
        }
        uint32_t inc = s;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t t = __shfl_up(inc, o, 64);
            if (lane >= o) inc += t;
        }
        if (lane == 63) wsum[wid] = inc;
        __syncthreads();
        uint32_t base = 0;
        for (int w = 0; w < wid; ++w) base += wsum[w];
        uint32_t run = base + inc - s;
#pragma unroll
        for (int q = 0; q < kPer; ++q) {
            hist[tid * kPer + q] = run;
            run += v[q];
        }
    }
    __syncthreads();
    for (int i = tid; i < cnt; i += kOrderThreads) tmp[atomicAdd(&hist[cell[i]], 1u)] = (uint16_t)i;    // hist[h]: start -> end
    __syncthreads();
    for (int i = tid; i < cnt; i += kOrderThreads) {
        const uint32_t h = cell[i];
        const uint32_t s0 = h ? hist[h - 1] : 0u, s1 = hist[h];      // the cell's slots (the previous cell's end .. its own)
        uint32_t rank = 0;
        for (uint32_t q = s0; q < s1; ++q) rank += tmp[q] < (uint16_t)i ? 1u : 0u;
        key[s0 + rank] = (uint16_t)i;
    }
    __syncthreads();
    const uint32_t imask = 0xffffu;

    // ---- outputs (what emd_init_kernel writes, in the new order) ----
    int* perm = reinterpret_cast<int*>(ws + (left ? c.permL : c.permR));
    for (int e = tid; e < NPx; e += kOrderThreads) perm[e] = (int)(key[e] & imask);
    float* remL = c.temp + (long)cloud * (c.n + c.m) * 2;
    float* remR = remL + c.n;
    const int pairs = (NPx + kSpare) / 2;
    for (int pr = tid; pr < pairs; pr += kOrderThreads) {
        const int j0 = 2 * pr, j1 = j0 + 1;
        const bool ok0 = j0 < cnt, ok1 = j1 < cnt;
        const uint32_t i0 = ok0 ? (key[j0] & imask) : 0u, i1 = ok1 ? (key[j1] & imask) : 0u;
        const float x0 = ok0 ? X[i0] : 0.f, y0 = ok0 ? Y[i0] : 0.f, z0 = ok0 ? Z[i0] : 0.f;
        const float x1 = ok1 ? X[i1] : 0.f, y1 = ok1 ? Y[i1] : 0.f, z1 = ok1 ? Z[i1] : 0.f;
        float4* p8 = reinterpret_cast<float4*>(ws + (left ? c.plp : c.prp) + (long)pr * 8);
        float4* p32 = reinterpret_cast<float4*>(ws + (left ? c.flp : c.frp) + (long)pr * 32);
        const float4 a = make_float4(x0, x1, y0, y1), b = make_float4(z0, z1, 0.f, 0.f), zero = make_float4(0.f, 0.f, 0.f, 0.f);
        p8[0] = a;
        p8[1] = b;
        p32[0] = a;
        p32[1] = b;
#pragma unroll
        for (int q = 2; q < 8; ++q) p32[q] = zero;
        if (left) {
            if (ok0) remL[j0] = multiL;
            if (ok1) remL[j1] = multiL;
        } else {
            *reinterpret_cast<float2*>(ws + c.rr + j0) = make_float2(ok0 ? multiR : 0.f, ok1 ? multiR : 0.f);
            if (ok0) remR[j0] = multiR;
            if (ok1) remR[j1] = multiR;
        }
    }
    // bounding boxes of the 8-candidate blocks (exact, over the real points) ...
    const int NB = NPx / kBlk, NT = NPx / kTile;
    float* bb = ws + (left ? c.blkL : c.blkR);
    float* lb = box0;                                  // (both box tables are free now: NB * 6 floats = their size)
    for (int g = tid; g < NB; g += kOrderThreads) {
        float lo[3] = {kFar, kFar, kFar}, hi[3] = {-kFar, -kFar, -kFar};
        for (int j = g * kBlk; j < min(g * kBlk + kBlk, cnt); ++j) {
            const uint32_t i = key[j] & imask;
            lo[0] = fminf(lo[0], X[i]); hi[0] = fmaxf(hi[0], X[i]);
            lo[1] = fminf(lo[1], Y[i]); hi[1] = fmaxf(hi[1], Y[i]);
            lo[2] = fminf(lo[2], Z[i]); hi[2] = fmaxf(hi[2], Z[i]);
        }
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            bb[q * NB + g] = lo[q];
            bb[(3 + q) * NB + g] = hi[q];
            lb[g * 6 + q] = lo[q];
            lb[g * 6 + 3 + q] = hi[q];
        }
    }
    __syncthreads();
    // ... and of the 64-row tiles
    float* tb = ws + (left ? c.tileL : c.tileR);
    for (int t = tid; t < NT; t += kOrderThreads) {
        float lo[3] = {kFar, kFar, kFar}, hi[3] = {-kFar, -kFar, -kFar};
        for (int g = t * (kTile / kBlk); g < (t + 1) * (kTile / kBlk); ++g) {
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                lo[q] = fminf(lo[q], lb[g * 6 + q]);
                hi[q] = fmaxf(hi[q], lb[g * 6 + 3 + q]);
            }
        }
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            tb[q * NT + t] = lo[q];
            tb[(3 + q) * NT + t] = hi[q];
        }
    }
    if (left && tid == 0) ws[c.flag] = 1.f;
}

// pair record `u` (0..3) of a stage held in two x16 SGPR groups: component c (0=x,1=y,2=z,3=w) as a float2
// (indices are compile-time constants once the stage loop is unrolled; elements (2i, 2i+1) form an aligned SGPR pair)
#define PAIRC(lo, hi, u, c) \
    ((u) < 2 ? f2{(lo)[(u)*8 + (c)*2], (lo)[(u)*8 + (c)*2 + 1]} : f2{(hi)[((u)-2) * 8 + (c)*2], (hi)[((u)-2) * 8 + (c)*2 + 1]})

__device__ __forceinline__ f2 splat(float v) { return f2{v, v}; }
// squared distances of a candidate pair to the lane's point: per element fma(dz,dz,fma(dy,dy,dx*dx))
__device__ __forceinline__ f2 sqdist2(f2 dx, f2 dy, f2 dz) {
    return __builtin_elementwise_fma(dz, dz, __builtin_elementwise_fma(dy, dy, dx * dx));
}
__device__ __forceinline__ f2 exp2_2(f2 a) { return f2{__builtin_amdgcn_exp2f(a.x), __builtin_amdgcn_exp2f(a.y)}; }

// Rows = set1.  DO3: phase 3 of level lev3 (remainL update, approxmatch.cu:161-194);
//               DO1: phase 1 of level lev1 (ratioL, :60-93).  Candidates: PRP (+RR) records on the scalar path.
// R rows per lane, as in emd_rows2_kernel.
template <bool DO3, bool DO1, int R>
__global__ __launch_bounds__(kThreads) void emd_rows1_kernel(Ctx c, int lev1, float l2e3, float l2e1) {
    __shared__ float part3[kParts][kRowsPerWg * R], part1[kParts][kRowsPerWg * R];
    const int cloud = blockIdx.y;
    const int lrow = threadIdx.x % kRowsPerWg;
    const int part = __builtin_amdgcn_readfirstlane(threadIdx.x / kRowsPerWg);   // wave-uniform
    float* ws = c.ws + (long)cloud * c.per_cloud;
    float* remL = c.temp + (long)cloud * (c.n + c.m) * 2;
    float* ratioL = remL + c.n + c.m;
    int k[R];
    bool ok[R];
    f2 px2[R], py2[R], pz2[R], rl2[R], acc3[R], acc1[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        k[r] = (blockIdx.x * R + r) * kRowsPerWg + lrow;
        ok[r] = k[r] < c.n;
        float px = 0.f, py = 0.f, pz = 0.f, rl = 0.f;
        if (ok[r]) {   // the row's point from its own record (the records are in the order the sweeps run in: emd_order_kernel)
            px = ws[c.plp + pair8(k[r], 0)];
            py = ws[c.plp + pair8(k[r], 1)];
            pz = ws[c.plp + pair8(k[r], 2)];
            if (DO3) rl = ratioL[k[r]];
        }
        px2[r] = splat(px);
        py2[r] = splat(py);
        pz2[r] = splat(pz);
        rl2[r] = splat(rl);
        // even / odd candidates accumulate in the two halves of a packed register (one v_pk_add_f32 per pair record
        // instead of two dependent v_add_f32); the halves are added once at the end, then the 4 candidate ranges in order
        acc3[r] = splat(0.f);
        acc1[r] = f2{part == 0 ? 1e-9f : 0.f, 0.f};
    }
    const f2 l3 = splat(l2e3), l1 = splat(l2e1);
    const int cand = c.MP / kParts;                                  // candidates of this wave's range
    const float* p = ws + c.prp + (long)part * cand * 4;   // wave-uniform
    const float* q = ws + c.rr + (long)part * cand;
    f32x16 a0, a1, b0, b1;
    f32x8 w0 = {}, w1 = {};
    auto work = [&](const f32x16& lo, const f32x16& hi, const f32x8& w) {
#pragma unroll
        for (int u = 0; u < kStage / 2; ++u) {
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const f2 d = sqdist2(PAIRC(lo, hi, u, 0) - px2[r], PAIRC(lo, hi, u, 1) - py2[r], PAIRC(lo, hi, u, 2) - pz2[r]);
                // the running sums take their product through an fma — what nvcc's default -fmad=true makes of the
                // reference's `w=...; suml+=w` (approxmatch.cu:86-87,185-189); the oracle's `contract` variant 3
                if (DO3) {
                    acc3[r] = __builtin_elementwise_fma(exp2_2(l3 * d) * rl2[r], PAIRC(lo, hi, u, 3), acc3[r]);   // (e * ratioL[k]) * ratioR[l]
                }
                if (DO1) {
                    acc1[r] = __builtin_elementwise_fma(exp2_2(l1 * d), f2{w[u * 2], w[u * 2 + 1]}, acc1[r]);   // e * remainR[l]
                }
            }
        }
    };
    HP_SLOAD16(a0, p, 0x0);
    HP_SLOAD16(a1, p, 0x40);
    if (DO1) HP_SLOAD8(w0, q, 0x0);
    asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(a0), "+s"(a1), "+s"(w0));
    for (int l0 = 0; l0 < cand; l0 += 2 * kStage) {
        p += kStage * 4;
        q += kStage;
        HP_SLOAD16(b0, p, 0x0);
        HP_SLOAD16(b1, p, 0x40);
        if (DO1) HP_SLOAD8(w1, q, 0x0);
        HP_PIN();
        work(a0, a1, w0);
        asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(b0), "+s"(b1), "+s"(w1), "+v"(acc3[0]), "+v"(acc1[0]), "+v"(acc3[R - 1]), "+v"(acc1[R - 1]));
        p += kStage * 4;
        q += kStage;
        HP_SLOAD16(a0, p, 0x0);      // past the last stage this reads the spare (zero) records
        HP_SLOAD16(a1, p, 0x40);
        if (DO1) HP_SLOAD8(w0, q, 0x0);
        HP_PIN();
        work(b0, b1, w1);
        asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(a0), "+s"(a1), "+s"(w0), "+v"(acc3[0]), "+v"(acc1[0]), "+v"(acc3[R - 1]), "+v"(acc1[R - 1]));
    }
    float s3[R], s1[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        s3[r] = acc3[r].x + acc3[r].y;
        s1[r] = acc1[r].x + acc1[r].y;
        part3[part][r * kRowsPerWg + lrow] = s3[r];
        part1[part][r * kRowsPerWg + lrow] = s1[r];
    }
    __syncthreads();
    if (part != 0) return;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        if (!ok[r]) continue;
        float t3 = s3[r], t1 = s1[r];
#pragma unroll
        for (int q2 = 1; q2 < kParts; ++q2) {
            t3 += part3[q2][r * kRowsPerWg + lrow];
            t1 += part1[q2][r * kRowsPerWg + lrow];
        }
        float rem = remL[k[r]];
        if (DO3) {
            rem = fmaxf(0.0f, rem - t3);
            remL[k[r]] = rem;
        }
        if (DO1) {
            const float v = rem / t1;
            ratioL[k[r]] = v;
            ws[c.plp + pair8(k[r], 3)] = v;
            ws[c.flp + pair32(k[r], 3 + lev1)] = v;
        }
    }
}

// Rows = set2: phase 2 (ratioR / remainR update, approxmatch.cu:109-142).  Candidates: PLP records.
// R rows per lane (rows l and l + 64, ...): every candidate record fetched on the scalar path serves R rows, so a stage
// carries R times the VALU work behind its s_waitcnt (R independent accumulation chains per lane) at 1/R of the scalar
// traffic.  Per row the arithmetic and its order are those of R = 1.
template <int R>
__global__ __launch_bounds__(kThreads) void emd_rows2_kernel(Ctx c, int lev, float l2e) {
    __shared__ float parts[kParts][kRowsPerWg * R];
    const int cloud = blockIdx.y;
    const int lrow = threadIdx.x % kRowsPerWg;
    const int part = __builtin_amdgcn_readfirstlane(threadIdx.x / kRowsPerWg);
    float* ws = c.ws + (long)cloud * c.per_cloud;
    float* remR = c.temp + (long)cloud * (c.n + c.m) * 2 + c.n;
    float* ratioR = remR + c.m + c.n;
    int l[R];
    bool ok[R];
    f2 qx2[R], qy2[R], qz2[R], acc2[R];   // acc2: even / odd candidates (see emd_rows1_kernel)
#pragma unroll
    for (int r = 0; r < R; ++r) {
        l[r] = (blockIdx.x * R + r) * kRowsPerWg + lrow;
        ok[r] = l[r] < c.m;
        float qx = 0.f, qy = 0.f, qz = 0.f;
        if (ok[r]) {
            qx = ws[c.prp + pair8(l[r], 0)];
            qy = ws[c.prp + pair8(l[r], 1)];
            qz = ws[c.prp + pair8(l[r], 2)];
        }
        qx2[r] = splat(qx);
        qy2[r] = splat(qy);
        qz2[r] = splat(qz);
        acc2[r] = splat(0.f);
    }
    const f2 lv = splat(l2e);
    const int cand = c.NP / kParts;
    const float* p = ws + c.plp + (long)part * cand * 4;
    f32x16 a0, a1, b0, b1;
    auto work = [&](const f32x16& lo, const f32x16& hi) {
#pragma unroll
        for (int u = 0; u < kStage / 2; ++u) {
#pragma unroll
            for (int r = 0; r < R; ++r) {
                // the reference evaluates (x2-x1) with x2 the set2 point in every phase (approxmatch.cu:85,131,185)
                const f2 d = sqdist2(qx2[r] - PAIRC(lo, hi, u, 0), qy2[r] - PAIRC(lo, hi, u, 1), qz2[r] - PAIRC(lo, hi, u, 2));
                acc2[r] = __builtin_elementwise_fma(exp2_2(lv * d), PAIRC(lo, hi, u, 3), acc2[r]);   // approxmatch.cu:131-132 contracted
            }
        }
    };
    HP_SLOAD16(a0, p, 0x0);
    HP_SLOAD16(a1, p, 0x40);
    asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(a0), "+s"(a1));
    for (int k0 = 0; k0 < cand; k0 += 2 * kStage) {
        p += kStage * 4;
        HP_SLOAD16(b0, p, 0x0);
        HP_SLOAD16(b1, p, 0x40);
        HP_PIN();
        work(a0, a1);
        if (R == 1) asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(b0), "+s"(b1), "+v"(acc2[0]));
        else asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(b0), "+s"(b1), "+v"(acc2[0]), "+v"(acc2[R - 1]));
        p += kStage * 4;
        HP_SLOAD16(a0, p, 0x0);
        HP_SLOAD16(a1, p, 0x40);
        HP_PIN();
        work(b0, b1);
        if (R == 1) asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(a0), "+s"(a1), "+v"(acc2[0]));
        else asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(a0), "+s"(a1), "+v"(acc2[0]), "+v"(acc2[R - 1]));
    }
    float acc[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        acc[r] = acc2[r].x + acc2[r].y;
        parts[part][r * kRowsPerWg + lrow] = acc[r];
    }
    __syncthreads();
    if (part != 0) return;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        if (!ok[r]) continue;
        float a = acc[r];
#pragma unroll
        for (int q2 = 1; q2 < kParts; ++q2) a += parts[q2][r * kRowsPerWg + lrow];
        const float rr = remR[l[r]];
        const float sumr = a * rr;
        const float consumption = fminf(rr / (sumr + 1e-9f), 1.0f);
        const float v = consumption * rr;
        const float rem = fmaxf(0.0f, rr - sumr);
        ratioR[l[r]] = v;
        remR[l[r]] = rem;
        ws[c.prp + pair8(l[r], 3)] = v;
        ws[c.rr + l[r]] = rem;
        ws[c.frp + pair32(l[r], 3 + lev)] = v;
    }
}

// ------------------------------------------------------------------------------------------------
// Culling sweeps (records in Hilbert order).  Same rows, same candidates, same arithmetic per (row, candidate) as emd_rows1_kernel /
// emd_rows2_kernel; the candidate set is cut into blocks of 8 (one pipeline stage), block g belongs to the workgroup's wave
// (g mod P) — interleaved, so that the blocks near a row tile spread over the P = 4 waves (8 or 16 waves per row tile were tried for
// the sparse levels and lost: 1.18 -> 1.22 / 1.30 ms per call) —, and a (64-row tile, block) unit is
// evaluated only if the two bounding boxes are closer than the level's underflow radius.  Every skipped term is an exact zero
// (exp2 of less than -152; fma(0, w, acc) == acc), so a row's sum is the sum over its surviving blocks in ascending order — the
// reference's sum with its zero terms left out.  The culling instances run ONE row per lane whatever the plain kernels' rows per
// lane are (a launch is one round of workgroups and lasts as long as its slowest CU: finer workgroups balance; R = 2: +3 % per
// call), so a row's result does not depend on hp_emd_set_rows_per_lane.  Lane i of a wave tests block i of its range (one ballot per row tile), the
// surviving blocks are walked with s_ff1 on the masks, their records prefetched one block ahead on the scalar path as before.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float box_gap2(const float (&tlo)[3], const float (&thi)[3], float cx0, float cy0, float cz0, float cx1, float cy1,
                                          float cz1) {
    const float gx = fmaxf(0.f, fmaxf(cx0 - thi[0], tlo[0] - cx1));
    const float gy = fmaxf(0.f, fmaxf(cy0 - thi[1], tlo[1] - cy1));
    const float gz = fmaxf(0.f, fmaxf(cz0 - thi[2], tlo[2] - cz1));
    return gx * gx + gy * gy + gz * gz;
}
// tile t's box (uniform address: scalar loads); tiles past the last one are empty
__device__ __forceinline__ void load_tile_box(const float* tb, int NT, int t, float (&lo)[3], float (&hi)[3]) {
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        lo[q] = t < NT ? tb[q * NT + t] : kFar;
        hi[q] = t < NT ? tb[(3 + q) * NT + t] : -kFar;
    }
}

template <bool DO3, bool DO1, int R, int P>
__global__ __launch_bounds__(64 * P) void emd_rows1_cull_kernel(Ctx c, int lev1, float l2e3, float l2e1, float thr3, float thr1) {
    __shared__ float part3[P][kRowsPerWg * R], part1[P][kRowsPerWg * R];
    const int cloud = blockIdx.y;
    const int lrow = threadIdx.x % kRowsPerWg;
    const int part = __builtin_amdgcn_readfirstlane(threadIdx.x / kRowsPerWg);   // wave-uniform
    float* ws = c.ws + (long)cloud * c.per_cloud;
    float* remL = c.temp + (long)cloud * (c.n + c.m) * 2;
    float* ratioL = remL + c.n + c.m;
    int k[R];
    bool ok[R];
    f2 px2[R], py2[R], pz2[R], rl2[R], acc3[R], acc1[R];
    float tlo[R][3], thi[R][3];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        k[r] = (blockIdx.x * R + r) * kRowsPerWg + lrow;
        ok[r] = k[r] < c.n;
        float px = 0.f, py = 0.f, pz = 0.f, rl = 0.f;
        if (ok[r]) {
            px = ws[c.plp + pair8(k[r], 0)];
            py = ws[c.plp + pair8(k[r], 1)];
            pz = ws[c.plp + pair8(k[r], 2)];
            if (DO3) rl = ratioL[k[r]];
        }
        px2[r] = splat(px);
        py2[r] = splat(py);
        pz2[r] = splat(pz);
        rl2[r] = splat(rl);
        acc3[r] = splat(0.f);
        acc1[r] = f2{part == 0 ? 1e-9f : 0.f, 0.f};
        load_tile_box(ws + c.tileL, c.NP / kTile, blockIdx.x * R + r, tlo[r], thi[r]);
    }
    const f2 l3 = splat(l2e3), l1 = splat(l2e1);
    const int NB = c.MP / kBlk, nblk = (NB + P - 1) / P;     // blocks of the set, blocks of this wave's range (g = i * P + part)
    const float* bb = ws + c.blkR;
    const float* prec = ws + c.prp;
    const float* wrec = ws + c.rr;
    const int lane = threadIdx.x & 63;
    f32x16 a0, a1, b0, b1;
    f32x8 w0 = {}, w1 = {};
    for (int ch = 0; ch < nblk; ch += 64) {
        // which blocks of this chunk can any row of tile r reach at the level of phase 1 (mask1) / phase 3 (mask3 <= mask1)
        unsigned long long mask1[R], mask3[R], any = 0ull;
        {
            const int i = ch + lane;
            const bool valid = i * P + part < NB;
            const int g = valid ? i * P + part : 0;
            const float cx0 = bb[g], cy0 = bb[NB + g], cz0 = bb[2 * NB + g], cx1 = bb[3 * NB + g], cy1 = bb[4 * NB + g], cz1 = bb[5 * NB + g];
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const float g2 = box_gap2(tlo[r], thi[r], cx0, cy0, cz0, cx1, cy1, cz1);
                mask3[r] = DO3 ? __ballot(valid && g2 <= thr3) : 0ull;
                mask1[r] = DO1 ? __ballot(valid && g2 <= thr1) : mask3[r];
                any |= mask1[r];
            }
        }
        auto work = [&](const f32x16& lo, const f32x16& hi, const f32x8& w, int bi) {
            if (R > 1) {      // every row tile of the wave reaches the block at both levels: the interleaved form of emd_rows1_kernel
                bool all = true;
#pragma unroll
                for (int r = 0; r < R; ++r) all = all && (((DO3 ? mask3[r] : mask1[r]) >> bi) & 1ull);
                if (all) {
#pragma unroll
                    for (int u = 0; u < kStage / 2; ++u) {
#pragma unroll
                        for (int r = 0; r < R; ++r) {
                            const f2 d = sqdist2(PAIRC(lo, hi, u, 0) - px2[r], PAIRC(lo, hi, u, 1) - py2[r], PAIRC(lo, hi, u, 2) - pz2[r]);
                            if (DO3) acc3[r] = __builtin_elementwise_fma(exp2_2(l3 * d) * rl2[r], PAIRC(lo, hi, u, 3), acc3[r]);
                            if (DO1) acc1[r] = __builtin_elementwise_fma(exp2_2(l1 * d), f2{w[u * 2], w[u * 2 + 1]}, acc1[r]);
                        }
                    }
                    return;
                }
            }
#pragma unroll
            for (int r = 0; r < R; ++r) {
                if (!((mask1[r] >> bi) & 1ull)) continue;
                const bool on3 = DO3 && ((mask3[r] >> bi) & 1ull);
                if (on3 || !DO1) {
#pragma unroll
                    for (int u = 0; u < kStage / 2; ++u) {
                        const f2 d = sqdist2(PAIRC(lo, hi, u, 0) - px2[r], PAIRC(lo, hi, u, 1) - py2[r], PAIRC(lo, hi, u, 2) - pz2[r]);
                        acc3[r] = __builtin_elementwise_fma(exp2_2(l3 * d) * rl2[r], PAIRC(lo, hi, u, 3), acc3[r]);   // (e * ratioL[k]) * ratioR[l]
                        if (DO1) acc1[r] = __builtin_elementwise_fma(exp2_2(l1 * d), f2{w[u * 2], w[u * 2 + 1]}, acc1[r]);   // e * remainR[l]
                    }
                } else {
#pragma unroll
                    for (int u = 0; u < kStage / 2; ++u) {
                        const f2 d = sqdist2(PAIRC(lo, hi, u, 0) - px2[r], PAIRC(lo, hi, u, 1) - py2[r], PAIRC(lo, hi, u, 2) - pz2[r]);
                        acc1[r] = __builtin_elementwise_fma(exp2_2(l1 * d), f2{w[u * 2], w[u * 2 + 1]}, acc1[r]);
                    }
                }
            }
        };
        if (any == 0ull) continue;
        int ia = __builtin_ctzll(any), ib = 0;
        any &= any - 1ull;
        {
            const long g = (long)(ch + ia) * P + part;
            const float* p = prec + g * (kBlk * 4);
            const float* q = wrec + g * kBlk;
            HP_SLOAD16(a0, p, 0x0);
            HP_SLOAD16(a1, p, 0x40);
            if (DO1) HP_SLOAD8(w0, q, 0x0);
            asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(a0), "+s"(a1), "+s"(w0));
        }
        while (true) {
            const bool more_b = any != 0ull;
            if (more_b) {
                ib = __builtin_ctzll(any);
                any &= any - 1ull;
                const long g = (long)(ch + ib) * P + part;
                const float* p = prec + g * (kBlk * 4);
                const float* q = wrec + g * kBlk;
                HP_SLOAD16(b0, p, 0x0);
                HP_SLOAD16(b1, p, 0x40);
                if (DO1) HP_SLOAD8(w1, q, 0x0);
            }
            HP_PIN();
            work(a0, a1, w0, ia);
            asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(b0), "+s"(b1), "+s"(w1), "+v"(acc3[0]), "+v"(acc1[0]), "+v"(acc3[R - 1]), "+v"(acc1[R - 1]));
            if (!more_b) break;
            const bool more_a = any != 0ull;
            if (more_a) {
                ia = __builtin_ctzll(any);
                any &= any - 1ull;
                const long g = (long)(ch + ia) * P + part;
                const float* p = prec + g * (kBlk * 4);
                const float* q = wrec + g * kBlk;
                HP_SLOAD16(a0, p, 0x0);
                HP_SLOAD16(a1, p, 0x40);
                if (DO1) HP_SLOAD8(w0, q, 0x0);
            }
            HP_PIN();
            work(b0, b1, w1, ib);
            asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(a0), "+s"(a1), "+s"(w0), "+v"(acc3[0]), "+v"(acc1[0]), "+v"(acc3[R - 1]), "+v"(acc1[R - 1]));
            if (!more_a) break;
        }
    }
    float s3[R], s1[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        s3[r] = acc3[r].x + acc3[r].y;
        s1[r] = acc1[r].x + acc1[r].y;
        part3[part][r * kRowsPerWg + lrow] = s3[r];
        part1[part][r * kRowsPerWg + lrow] = s1[r];
    }
    __syncthreads();
    if (part != 0) return;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        if (!ok[r]) continue;
        float t3 = s3[r], t1 = s1[r];
#pragma unroll
        for (int q2 = 1; q2 < P; ++q2) {
            t3 += part3[q2][r * kRowsPerWg + lrow];
            t1 += part1[q2][r * kRowsPerWg + lrow];
        }
        float rem = remL[k[r]];
        if (DO3) {
            rem = fmaxf(0.0f, rem - t3);
            remL[k[r]] = rem;
        }
        if (DO1) {
            const float v = rem / t1;
            ratioL[k[r]] = v;
            ws[c.plp + pair8(k[r], 3)] = v;
            ws[c.flp + pair32(k[r], 3 + lev1)] = v;
        }
    }
}

template <int R, int P>
__global__ __launch_bounds__(64 * P) void emd_rows2_cull_kernel(Ctx c, int lev, float l2e, float thr) {
    __shared__ float parts[P][kRowsPerWg * R];
    const int cloud = blockIdx.y;
    const int lrow = threadIdx.x % kRowsPerWg;
    const int part = __builtin_amdgcn_readfirstlane(threadIdx.x / kRowsPerWg);
    float* ws = c.ws + (long)cloud * c.per_cloud;
    float* remR = c.temp + (long)cloud * (c.n + c.m) * 2 + c.n;
    float* ratioR = remR + c.m + c.n;
    int l[R];
    bool ok[R];
    f2 qx2[R], qy2[R], qz2[R], acc2[R];
    float tlo[R][3], thi[R][3];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        l[r] = (blockIdx.x * R + r) * kRowsPerWg + lrow;
        ok[r] = l[r] < c.m;
        float qx = 0.f, qy = 0.f, qz = 0.f;
        if (ok[r]) {
            qx = ws[c.prp + pair8(l[r], 0)];
            qy = ws[c.prp + pair8(l[r], 1)];
            qz = ws[c.prp + pair8(l[r], 2)];
        }
        qx2[r] = splat(qx);
        qy2[r] = splat(qy);
        qz2[r] = splat(qz);
        acc2[r] = splat(0.f);
        load_tile_box(ws + c.tileR, c.MP / kTile, blockIdx.x * R + r, tlo[r], thi[r]);
    }
    const f2 lv = splat(l2e);
    const int NB = c.NP / kBlk, nblk = (NB + P - 1) / P;
    const float* bb = ws + c.blkL;
    const float* prec = ws + c.plp;
    const int lane = threadIdx.x & 63;
    f32x16 a0, a1, b0, b1;
    for (int ch = 0; ch < nblk; ch += 64) {
        unsigned long long mask[R], any = 0ull;
        {
            const int i = ch + lane;
            const bool valid = i * P + part < NB;
            const int g = valid ? i * P + part : 0;
            const float cx0 = bb[g], cy0 = bb[NB + g], cz0 = bb[2 * NB + g], cx1 = bb[3 * NB + g], cy1 = bb[4 * NB + g], cz1 = bb[5 * NB + g];
#pragma unroll
            for (int r = 0; r < R; ++r) {
                mask[r] = __ballot(valid && box_gap2(tlo[r], thi[r], cx0, cy0, cz0, cx1, cy1, cz1) <= thr);
                any |= mask[r];
            }
        }
        auto work = [&](const f32x16& lo, const f32x16& hi, int bi) {
            if (R > 1) {      // every row tile of the wave reaches the block: the interleaved form of emd_rows2_kernel
                bool all = true;
#pragma unroll
                for (int r = 0; r < R; ++r) all = all && ((mask[r] >> bi) & 1ull);
                if (all) {
#pragma unroll
                    for (int u = 0; u < kStage / 2; ++u) {
#pragma unroll
                        for (int r = 0; r < R; ++r) {
                            const f2 d = sqdist2(qx2[r] - PAIRC(lo, hi, u, 0), qy2[r] - PAIRC(lo, hi, u, 1), qz2[r] - PAIRC(lo, hi, u, 2));
                            acc2[r] = __builtin_elementwise_fma(exp2_2(lv * d), PAIRC(lo, hi, u, 3), acc2[r]);
                        }
                    }
                    return;
                }
            }
#pragma unroll
            for (int r = 0; r < R; ++r) {
                if (!((mask[r] >> bi) & 1ull)) continue;
#pragma unroll
                for (int u = 0; u < kStage / 2; ++u) {
                    const f2 d = sqdist2(qx2[r] - PAIRC(lo, hi, u, 0), qy2[r] - PAIRC(lo, hi, u, 1), qz2[r] - PAIRC(lo, hi, u, 2));
                    acc2[r] = __builtin_elementwise_fma(exp2_2(lv * d), PAIRC(lo, hi, u, 3), acc2[r]);   // approxmatch.cu:131-132 contracted
                }
            }
        };
        if (any == 0ull) continue;
        int ia = __builtin_ctzll(any), ib = 0;
        any &= any - 1ull;
        {
            const float* p = prec + ((long)(ch + ia) * P + part) * (kBlk * 4);
            HP_SLOAD16(a0, p, 0x0);
            HP_SLOAD16(a1, p, 0x40);
            asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(a0), "+s"(a1));
        }
        while (true) {
            const bool more_b = any != 0ull;
            if (more_b) {
                ib = __builtin_ctzll(any);
                any &= any - 1ull;
                const float* p = prec + ((long)(ch + ib) * P + part) * (kBlk * 4);
                HP_SLOAD16(b0, p, 0x0);
                HP_SLOAD16(b1, p, 0x40);
            }
            HP_PIN();
            work(a0, a1, ia);
            asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(b0), "+s"(b1), "+v"(acc2[0]), "+v"(acc2[R - 1]));
            if (!more_b) break;
            const bool more_a = any != 0ull;
            if (more_a) {
                ia = __builtin_ctzll(any);
                any &= any - 1ull;
                const float* p = prec + ((long)(ch + ia) * P + part) * (kBlk * 4);
                HP_SLOAD16(a0, p, 0x0);
                HP_SLOAD16(a1, p, 0x40);
            }
            HP_PIN();
            work(b0, b1, ib);
            asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(a0), "+s"(a1), "+v"(acc2[0]), "+v"(acc2[R - 1]));
            if (!more_a) break;
        }
    }
    float acc[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        acc[r] = acc2[r].x + acc2[r].y;
        parts[part][r * kRowsPerWg + lrow] = acc[r];
    }
    __syncthreads();
    if (part != 0) return;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        if (!ok[r]) continue;
        float a = acc[r];
#pragma unroll
        for (int q2 = 1; q2 < P; ++q2) a += parts[q2][r * kRowsPerWg + lrow];
        const float rr = remR[l[r]];
        const float sumr = a * rr;
        const float consumption = fminf(rr / (sumr + 1e-9f), 1.0f);
        const float v = consumption * rr;
        const float rem = fmaxf(0.0f, rr - sumr);
        ratioR[l[r]] = v;
        remR[l[r]] = rem;
        ws[c.prp + pair8(l[r], 3)] = v;
        ws[c.rr + l[r]] = rem;
        ws[c.frp + pair32(l[r], 3 + lev)] = v;
    }
}

// final pair record (two x16 SGPR groups): component q (0..2 xyz, 3+lev ratio) as a float2
#define FINC(lo, hi, q) ((q) < 8 ? f2{(lo)[(q)*2], (lo)[(q)*2 + 1]} : f2{(hi)[((q)-8) * 2], (hi)[((q)-8) * 2 + 1]})

// M(l,k) for the two candidates of a pair record = sum over levels, in level order, of
//   ROW_IS_L: (exp(level*d) * ratioL_lev[row]) * ratioR_lev[cand]     (row = set1 point, candidates = set2)
//   else    : (exp(level*d) * ratioL_lev[cand]) * ratioR_lev[row]
// DERIVE (the cost / gradient sweeps of the training path; never the `match` the API returns): the levels are exact powers of
// 4 apart — l_j * d = 4 * (l_{j+1} * d) bit for bit — so exp2(l_j d) = exp2(l_{j+1} d)^4, and levels 0, 2, 4, 6 are formed as the
// fourth power (two packed multiplies) of the hardware exponential of levels 1, 3, 5, 7: five v_exp_f32 per pair instead of nine.
// A derived value carries ~5 ulp (4 x the exponential's + the two squarings') instead of 1.  Round 3 tried this in the LEVEL sweeps
// as well and the auction amplified it past the parity bars (docs/DESIGN_HISTORY.md 7b); here nothing is downstream of the value: M moves by
// <= 3.5e-7 relative, cost and gradients by less (tests: the cost error map's bars are unchanged, and the exact form stays
// selectable: hp_emd_set_final_derive).
// SKIP (round 6): the first SKIP levels' exponentials are known to be exactly zero for this (row tile, candidate block) — the
// boxes are further apart than those levels' underflow radius (the culling test of the level sweeps; a derived value is the
// fourth power of a hardware exponential below 2^-38, i.e. below 2^-152: zero as well) — and their terms are left out.
template <bool ROW_IS_L, bool DERIVE, int SKIP = 0>
__device__ __forceinline__ f2 match_entry2(f2 d, const float (&row)[kLevels], const f32x16& lo, const f32x16& hi) {
    f2 acc = splat(0.f);
    f2 ev[kLevels];
#pragma unroll
    for (int lev = kLevels - 1; lev >= 0; --lev) {
        if (lev < SKIP) continue;      // (an odd level is the source of the even level below it only: nothing kept derives from a skipped one)
        if (!DERIVE || lev == kLevels - 1 || (lev & 1)) {
            ev[lev] = exp2_2(splat(level_l2e(lev)) * d);
        } else {
            const f2 sq = ev[lev + 1] * ev[lev + 1];
            ev[lev] = sq * sq;
        }
    }
#pragma unroll
    for (int lev = SKIP; lev < kLevels; ++lev) {
        const f2 e = ev[lev];
        const f2 cr = FINC(lo, hi, 3 + lev);
        // the level's term rides on an fma into the running sum (one rounding instead of the reference's two, i.e. a
        // slightly more accurate M; unlike the phase sweeps nothing downstream amplifies it: cost moves by ~1e-7 relative)
        acc = ROW_IS_L ? __builtin_elementwise_fma(e * splat(row[lev]), cr, acc)
                       : __builtin_elementwise_fma(e * cr, splat(row[lev]), acc);
    }
    return acc;
}

__device__ __forceinline__ void load_row_final(const float* rec, int i, float& x, float& y, float& z, float (&r)[kLevels]) {
    x = rec[pair32(i, 0)];
    y = rec[pair32(i, 1)];
    z = rec[pair32(i, 2)];
#pragma unroll
    for (int lev = 0; lev < kLevels; ++lev) r[lev] = rec[pair32(i, 3 + lev)];
}

constexpr int kLT = 64;  // match rows (l) per workgroup in the materialising pass
__global__ __launch_bounds__(kThreads) void emd_match_kernel(Ctx c, float* __restrict__ match) {
    const int cloud = blockIdx.z;
    const int k = blockIdx.x * kThreads + threadIdx.x;
    const int l0 = blockIdx.y * kLT;
    const float* ws = c.ws + (long)cloud * c.per_cloud;
    const bool ok = k < c.n;
    float px = 0.f, py = 0.f, pz = 0.f, rL[kLevels] = {};
    if (ok) load_row_final(ws + c.flp, k, px, py, pz, rL);
    const f2 px2 = splat(px), py2 = splat(py), pz2 = splat(pz);
    float* out = match + ((long)cloud * c.m + l0) * c.n + k;
    const int cnt = min(kLT, c.m - l0);          // l0 is even; spare records exist past MP
    const float* p = ws + c.frp + (long)(l0 >> 1) * 32;
    f32x16 a0, a1, b0, b1;
    HP_SLOAD16(a0, p, 0x0);
    HP_SLOAD16(a1, p, 0x40);
    asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(a0), "+s"(a1));
    for (int l = 0; l < cnt; l += 4) {
        p += 32;
        HP_SLOAD16(b0, p, 0x0);
        HP_SLOAD16(b1, p, 0x40);
        HP_PIN();
        f2 v = match_entry2<true, false>(sqdist2(FINC(a0, a1, 0) - px2, FINC(a0, a1, 1) - py2, FINC(a0, a1, 2) - pz2), rL, a0, a1);
        if (ok) {
            out[(long)l * c.n] = v.x;
            if (l + 1 < cnt) out[(long)(l + 1) * c.n] = v.y;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(b0), "+s"(b1), "+v"(v));
        p += 32;
        HP_SLOAD16(a0, p, 0x0);
        HP_SLOAD16(a1, p, 0x40);
        HP_PIN();
        v = match_entry2<true, false>(sqdist2(FINC(b0, b1, 0) - px2, FINC(b0, b1, 1) - py2, FINC(b0, b1, 2) - pz2), rL, b0, b1);
        if (ok && l + 2 < cnt) {
            out[(long)(l + 2) * c.n] = v.x;
            if (l + 3 < cnt) out[(long)(l + 3) * c.n] = v.y;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(a0), "+s"(a1), "+v"(v));
    }
}

// match-free cost + grad1:  cost_b = sum_{k,l} M(l,k) sqrt(d),  grad1[k] = sum_l M(l,k) (p_k-q_l)/max(|p_k-q_l|,1e-10)
// (approxmatch.cu:215-255, 301-322 without the match tensor).  One lane per k, all l on the scalar path.
template <bool DERIVE>
__global__ __launch_bounds__(kThreads) void emd_cost_grad1_kernel(Ctx c, float* __restrict__ partials, float* __restrict__ grad1) {
    __shared__ float red[kThreads / 64];
    __shared__ float parts[kParts][4][kRowsPerWg];
    const int cloud = blockIdx.y;
    const int lrow = threadIdx.x % kRowsPerWg;
    const int part = __builtin_amdgcn_readfirstlane(threadIdx.x / kRowsPerWg);
    const int k = blockIdx.x * kRowsPerWg + lrow;
    const float* ws = c.ws + (long)cloud * c.per_cloud;
    const bool ok = k < c.n;
    float px = 0.f, py = 0.f, pz = 0.f, rL[kLevels] = {};
    if (ok) load_row_final(ws + c.flp, k, px, py, pz, rL);
    const f2 px2 = splat(px), py2 = splat(py), pz2 = splat(pz);
    f2 cost2 = splat(0.f), dx2 = splat(0.f), dy2 = splat(0.f), dz2 = splat(0.f);   // even / odd candidates (see emd_rows1_kernel)
    auto work = [&](const f32x16& lo, const f32x16& hi) {
        const f2 ex = px2 - FINC(lo, hi, 0), ey = py2 - FINC(lo, hi, 1), ez = pz2 - FINC(lo, hi, 2);   // (x1 - x2), approxmatch.cu:312
        const f2 d2 = sqdist2(ex, ey, ez);       // squares: the sign of the difference does not change a bit
        const f2 mv = match_entry2<true, DERIVE>(d2, rL, lo, hi);
        const f2 w = mv * f2{__builtin_amdgcn_rsqf(fmaxf(d2.x, 1e-20f)), __builtin_amdgcn_rsqf(fmaxf(d2.y, 1e-20f))};
        // M * sqrt(d) as (M / sqrt(d)) * d: the reciprocal root is needed for the gradient anyway (d = 0: w * 0 = 0 = M * sqrt(0))
        cost2 = DERIVE ? __builtin_elementwise_fma(w, d2, cost2)
                       : __builtin_elementwise_fma(mv, f2{__builtin_amdgcn_sqrtf(d2.x), __builtin_amdgcn_sqrtf(d2.y)}, cost2);
        dx2 = __builtin_elementwise_fma(ex, w, dx2);
        dy2 = __builtin_elementwise_fma(ey, w, dy2);
        dz2 = __builtin_elementwise_fma(ez, w, dz2);
    };
    const int cand = c.MP / kParts;
    const float* p = ws + c.frp + (long)part * cand * 16;
    f32x16 a0, a1, b0, b1;
    HP_SLOAD16(a0, p, 0x0);
    HP_SLOAD16(a1, p, 0x40);
    asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(a0), "+s"(a1));
    for (int l = 0; l < cand; l += 4) {
        p += 32;
        HP_SLOAD16(b0, p, 0x0);
        HP_SLOAD16(b1, p, 0x40);
        HP_PIN();
        work(a0, a1);
        asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(b0), "+s"(b1), "+v"(cost2), "+v"(dx2), "+v"(dy2), "+v"(dz2));
        p += 32;
        HP_SLOAD16(a0, p, 0x0);
        HP_SLOAD16(a1, p, 0x40);
        HP_PIN();
        work(b0, b1);
        asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(a0), "+s"(a1), "+v"(cost2), "+v"(dx2), "+v"(dy2), "+v"(dz2));
    }
    float cost = cost2.x + cost2.y, dx = dx2.x + dx2.y, dy = dy2.x + dy2.y, dz = dz2.x + dz2.y;
    parts[part][0][lrow] = dx;
    parts[part][1][lrow] = dy;
    parts[part][2][lrow] = dz;
    parts[part][3][lrow] = cost;
    __syncthreads();
    if (part == 0) {
#pragma unroll
        for (int q2 = 1; q2 < kParts; ++q2) {
            dx += parts[q2][0][lrow];
            dy += parts[q2][1][lrow];
            dz += parts[q2][2][lrow];
            cost += parts[q2][3][lrow];
        }
        if (ok && grad1) {
            // records in k-d order (emd_order_kernel): position k holds the caller's point permL[k]
            const int ko = ws[c.flag] != 0.f ? reinterpret_cast<const int*>(ws + c.permL)[k] : k;
            float* g = grad1 + ((long)cloud * c.n + ko) * 3;
            g[0] = dx;
            g[1] = dy;
            g[2] = dz;
        }
    }
    const float t = hp::block_sum((ok && part == 0) ? cost : 0.f, red);
    if (threadIdx.x == 0) partials[(long)cloud * gridDim.x + blockIdx.x] = t;
}

// match-free grad2[l] = sum_k M(l,k) (q_l-p_k)/max(|q_l-p_k|,1e-10)   (approxmatch.cu:260-300); with WITH_COST the
// same sweep also yields the cost (sum over the same pairs, owned by l instead of k), so a training step that only
// needs d cost / d xyz2 evaluates the match entries once.
template <bool WITH_COST, int R, bool DERIVE>
__global__ __launch_bounds__(kThreads) void emd_grad2_kernel(Ctx c, float* __restrict__ grad2, float* __restrict__ partials, float thr1, float thr2) {
    __shared__ float red[kThreads / 64];
    __shared__ float parts[kParts][4][kRowsPerWg * R];
    const int cloud = blockIdx.y;
    const int lrow = threadIdx.x % kRowsPerWg;
    const int part = __builtin_amdgcn_readfirstlane(threadIdx.x / kRowsPerWg);
    const float* ws = c.ws + (long)cloud * c.per_cloud;
    // thr1 / thr2: the underflow radii of levels 1 and 2 (3e38: that tier is off).  Only with the records in Hilbert order: the
    // boxes exist and mean something.
    const bool tiers = ws[c.flag] != 0.f && thr1 < 1e38f;
    int l[R];
    bool ok[R];
    float rR[R][kLevels];
    float tlo[R][3], thi[R][3];
    f2 qx2[R], qy2[R], qz2[R], sx2[R], sy2[R], sz2[R], cost2[R];   // sums: even / odd candidates (see emd_rows1_kernel)
#pragma unroll
    for (int r = 0; r < R; ++r) {
        l[r] = (blockIdx.x * R + r) * kRowsPerWg + lrow;
        ok[r] = l[r] < c.m;
        float qx = 0.f, qy = 0.f, qz = 0.f;
#pragma unroll
        for (int lev = 0; lev < kLevels; ++lev) rR[r][lev] = 0.f;
        if (ok[r]) load_row_final(ws + c.frp, l[r], qx, qy, qz, rR[r]);
        qx2[r] = splat(qx);
        qy2[r] = splat(qy);
        qz2[r] = splat(qz);
        sx2[r] = sy2[r] = sz2[r] = cost2[r] = splat(0.f);
        if (tiers) load_tile_box(ws + c.tileR, c.MP / kTile, blockIdx.x * R + r, tlo[r], thi[r]);
    }
    unsigned long long far1[R], far2[R];      // per candidate block of the current chunk: beyond level 1's radius / level 2's
    auto work = [&](const f32x16& lo, const f32x16& hi, int bi) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const f2 ex = qx2[r] - FINC(lo, hi, 0), ey = qy2[r] - FINC(lo, hi, 1), ez = qz2[r] - FINC(lo, hi, 2);
            const f2 d2 = sqdist2(ex, ey, ez);
            f2 mv;
            if ((far2[r] >> bi) & 1ull) mv = match_entry2<false, DERIVE, 3>(d2, rR[r], lo, hi);
            else if ((far1[r] >> bi) & 1ull) mv = match_entry2<false, DERIVE, 2>(d2, rR[r], lo, hi);
            else mv = match_entry2<false, DERIVE, 0>(d2, rR[r], lo, hi);
            const f2 w = mv * f2{__builtin_amdgcn_rsqf(fmaxf(d2.x, 1e-20f)), __builtin_amdgcn_rsqf(fmaxf(d2.y, 1e-20f))};
            if (WITH_COST)
                cost2[r] = DERIVE ? __builtin_elementwise_fma(w, d2, cost2[r])      // (M / sqrt(d)) * d, see emd_cost_grad1_kernel
                                  : __builtin_elementwise_fma(mv, f2{__builtin_amdgcn_sqrtf(d2.x), __builtin_amdgcn_sqrtf(d2.y)}, cost2[r]);
            sx2[r] = __builtin_elementwise_fma(ex, w, sx2[r]);
            sy2[r] = __builtin_elementwise_fma(ey, w, sy2[r]);
            sz2[r] = __builtin_elementwise_fma(ez, w, sz2[r]);
        }
    };
    const int cand = c.NP / kParts, nblk = cand / kBlk;      // this wave's candidates: blocks [part * nblk, (part + 1) * nblk) of set1
    const int NB = c.NP / kBlk;
    const float* bb = ws + c.blkL;
    const int lane = threadIdx.x & 63;
    const float* p = ws + c.flp + (long)part * cand * 16;
    f32x16 a0, a1, b0, b1;
    HP_SLOAD16(a0, p, 0x0);
    HP_SLOAD16(a1, p, 0x40);
    asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(a0), "+s"(a1));
    for (int ch = 0; ch < nblk; ch += 64) {
#pragma unroll
        for (int r = 0; r < R; ++r) far1[r] = far2[r] = 0ull;
        if (tiers) {
            const bool valid = ch + lane < nblk;
            const int g = valid ? part * nblk + ch + lane : 0;
            const float cx0 = bb[g], cy0 = bb[NB + g], cz0 = bb[2 * NB + g], cx1 = bb[3 * NB + g], cy1 = bb[4 * NB + g], cz1 = bb[5 * NB + g];
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const float g2 = box_gap2(tlo[r], thi[r], cx0, cy0, cz0, cx1, cy1, cz1);
                far1[r] = __ballot(valid && g2 > thr1);
                far2[r] = __ballot(valid && g2 > thr2);
            }
        }
        const int nb = min(64, nblk - ch);
        for (int bi = 0; bi < nb; ++bi) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {      // a block = 8 candidates = four pair records
                p += 32;
                HP_SLOAD16(b0, p, 0x0);
                HP_SLOAD16(b1, p, 0x40);
                HP_PIN();
                work(a0, a1, bi);
                asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(b0), "+s"(b1), "+v"(sx2[0]), "+v"(sy2[0]), "+v"(sz2[0]), "+v"(cost2[0]),
                             "+v"(sx2[R - 1]), "+v"(sy2[R - 1]), "+v"(sz2[R - 1]), "+v"(cost2[R - 1]));
                p += 32;
                HP_SLOAD16(a0, p, 0x0);
                HP_SLOAD16(a1, p, 0x40);
                HP_PIN();
                work(b0, b1, bi);
                asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(a0), "+s"(a1), "+v"(sx2[0]), "+v"(sy2[0]), "+v"(sz2[0]), "+v"(cost2[0]),
                             "+v"(sx2[R - 1]), "+v"(sy2[R - 1]), "+v"(sz2[R - 1]), "+v"(cost2[R - 1]));
            }
        }
    }
    float sx[R], sy[R], sz[R], cost[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        sx[r] = sx2[r].x + sx2[r].y;
        sy[r] = sy2[r].x + sy2[r].y;
        sz[r] = sz2[r].x + sz2[r].y;
        cost[r] = cost2[r].x + cost2[r].y;
        parts[part][0][r * kRowsPerWg + lrow] = sx[r];
        parts[part][1][r * kRowsPerWg + lrow] = sy[r];
        parts[part][2][r * kRowsPerWg + lrow] = sz[r];
        parts[part][3][r * kRowsPerWg + lrow] = cost[r];
    }
    __syncthreads();
    float total = 0.f;
    if (part == 0) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
#pragma unroll
            for (int q2 = 1; q2 < kParts; ++q2) {   // candidate ranges in ascending order
                sx[r] += parts[q2][0][r * kRowsPerWg + lrow];
                sy[r] += parts[q2][1][r * kRowsPerWg + lrow];
                sz[r] += parts[q2][2][r * kRowsPerWg + lrow];
                cost[r] += parts[q2][3][r * kRowsPerWg + lrow];
            }
            if (ok[r]) {
                const int lo = ws[c.flag] != 0.f ? reinterpret_cast<const int*>(ws + c.permR)[l[r]] : l[r];   // see emd_cost_grad1_kernel
                float* g = grad2 + ((long)cloud * c.m + lo) * 3;
                if (c.acc_scale != 0.f) {   // hp_emd_forward_acc: the caller's running gradient (+= coef * this term)
                    g[0] = __builtin_fmaf(c.acc_scale, sx[r], g[0]);
                    g[1] = __builtin_fmaf(c.acc_scale, sy[r], g[1]);
                    g[2] = __builtin_fmaf(c.acc_scale, sz[r], g[2]);
                } else {
                    g[0] = sx[r];
                    g[1] = sy[r];
                    g[2] = sz[r];
                }
                total += cost[r];
            }
        }
    }
    if (WITH_COST) {
        const float t = hp::block_sum(total, red);
        if (threadIdx.x == 0) partials[(long)cloud * gridDim.x + blockIdx.x] = t;
    }
}

__global__ __launch_bounds__(256) void emd_cost_finish_kernel(const float* __restrict__ partials, int per_cloud, float* __restrict__ out) {
    __shared__ double red[4];
    const float* p = partials + (long)blockIdx.x * per_cloud;
    double s = 0;
    for (int i = threadIdx.x; i < per_cloud; i += 256) s += (double)p[i];
    const double t = hp::block_sum(s, red);
    if (threadIdx.x == 0) out[blockIdx.x] = (float)t;
}

// final_remainL: also run phase 3 of the last level.  Its only products are the last `match +=` term — which the match pass
// and the cost/gradient sweeps re-evaluate from the final records — and remainL's final value in `temp`: the API path
// (hp_approxmatch_ws) returns that, the match-free training path (hp_emd_forward) has no reader for it and skips the
// launch (one of 28 exponential sweeps, 55 us at B=64, N=2048).
// The level sweeps of `b` clouds as a sequence of launches on one stream, one step at a time: step 0 = the record set-up,
// step 1 = phase 1 of level 0, then per level phase 2 and the merged phase-3 / next-phase-1 launch.  Stepwise so that two
// chains (emd_forward_impl) can be ENQUEUED alternately: the host then feeds both streams at the same pace and the chains
// run side by side from the first launch to the last (enqueued one after the other, the second chain trailed the first by
// the host's ~0.2 ms of launch calls and ran its last sweeps alone on a half-empty chip).
struct LevelChain {
    Ctx c;
    int b;
    hipStream_t stream;
    bool final_remainL;
    float multiL, multiR;
    int rows1_r, rows2_r;
    int cull;                      // > 0: records in k-d order, sweeps of levels < cull skip the units that are exactly zero
    int logpL = 0, logpR = 0;      // log2 of the order kernel's sort sizes
    dim3 ginit, g1[3], g2[3];      // grids at 1, 2, 4 rows per lane

    static int log2_ceil64(int x) {
        int l = 6;
        while ((1 << l) < x) ++l;
        return l;
    }
    // largest squared distance whose exponential at `lev` is not exactly zero (exp2 of less than -152 is +0 in fp32, denormals
    // included; the margin over -150 covers the rounding of the box test and of the kernels' own distance)
    float radius2(int lev) const { return lev < cull ? 152.f / -level_l2e(lev) : 3.0e38f; }

    LevelChain(int b_, int n, int m, const float* xyz1, const float* xyz2, float* temp, float* ws, hipStream_t st, bool frl, int cull_)
        : b(b_), stream(st), final_remainL(frl), cull(cull_) {
        const WsLayout L = ws_layout(n, m);
        c = make_ctx(n, m, xyz1, xyz2, temp, ws);
        if (n >= m) {
            multiL = 1;
            multiR = (float)(n / m);  // integer division (approxmatch.cu:37-43)
        } else {
            multiL = (float)(m / n);
            multiR = 1;
        }
        logpL = log2_ceil64(n);
        logpR = log2_ceil64(m);
        if (logpL > kOrderMaxLog || logpR > kOrderMaxLog) cull = 0;
        for (int i = 0; i < 3; ++i) {
            const int r = 1 << i;
            g1[i] = dim3((n + r * kRowsPerWg - 1) / (r * kRowsPerWg), b);
            g2[i] = dim3((m + r * kRowsPerWg - 1) / (r * kRowsPerWg), b);
        }
        ginit = dim3(((L.NP + L.MP + 2 * kSpare) / 2 + 255) / 256, b);
        // rows per lane: the most that still leaves >= 2 waves per SIMD on the chip (measured at B=64, N=2048 on the whole
        // step: phase 1/3 kernel best at 2 — 4 costs occupancy it needs —, phase 2 at 4: -0.10 ms together; tools/emd_rows_sweep.sh).
        // hp_emd_set_rows_per_lane (or HP_EMD_ROWS1_R / HP_EMD_ROWS2_R at load time) overrides: every instance is a
        // per-row-identical evaluation (tests/test_structural_losses_gpu.py compares them bit for bit and with the oracle).
        auto pick = [&](int rows, int cap) {
            for (int r = cap; r > 1; r >>= 1)
                if ((long)b * ((rows + r * kRowsPerWg - 1) / (r * kRowsPerWg)) * (kThreads / 64) >= 2048) return r;
            return 1;
        };
        const int f1 = g_rows1.load(std::memory_order_relaxed), f2 = g_rows2.load(std::memory_order_relaxed);
        rows1_r = f1 ? f1 : pick(n, 2);
        rows2_r = f2 ? f2 : pick(m, 4);
    }

    // phase 3 of level lev3 (D3) merged with phase 1 of level lev1 (D1)
    template <bool D3, bool D1>
    void rows1(int lev3, int lev1) const {
        const float l2e3 = D3 ? level_l2e(lev3) : 0.f, l2e1 = D1 ? level_l2e(lev1) : 0.f;
        const int i = rows1_r == 4 ? 2 : rows1_r == 2 ? 1 : 0;
        // (lev3 < lev1.  A launch whose phase-1 level is past the culling levels visits every block anyway, and then the plain
        // kernel's straight pipeline is faster than skipping half of the phase-3 terms: measured)
        const int clev = D1 ? lev1 : lev3;
        if (cull > 0 && clev < cull) {
            const float t3 = D3 ? radius2(lev3) : 0.f, t1 = D1 ? radius2(lev1) : 0.f;
            hipLaunchKernelGGL((emd_rows1_cull_kernel<D3, D1, 1, kParts>), g1[0], dim3(kThreads), 0, stream, c, lev1, l2e3, l2e1, t3, t1);
            return;
        }
        if (rows1_r == 4) hipLaunchKernelGGL((emd_rows1_kernel<D3, D1, 4>), g1[i], dim3(kThreads), 0, stream, c, lev1, l2e3, l2e1);
        else if (rows1_r == 2) hipLaunchKernelGGL((emd_rows1_kernel<D3, D1, 2>), g1[i], dim3(kThreads), 0, stream, c, lev1, l2e3, l2e1);
        else hipLaunchKernelGGL((emd_rows1_kernel<D3, D1, 1>), g1[i], dim3(kThreads), 0, stream, c, lev1, l2e3, l2e1);
    }
    void rows2(int lev) const {
        const int i = rows2_r == 4 ? 2 : rows2_r == 2 ? 1 : 0;
        if (cull > 0 && lev < cull) {
            const float t = radius2(lev);
            hipLaunchKernelGGL((emd_rows2_cull_kernel<1, kParts>), g2[0], dim3(kThreads), 0, stream, c, lev, level_l2e(lev), t);
            return;
        }
        if (rows2_r == 4) hipLaunchKernelGGL(emd_rows2_kernel<4>, g2[i], dim3(kThreads), 0, stream, c, lev, level_l2e(lev));
        else if (rows2_r == 2) hipLaunchKernelGGL(emd_rows2_kernel<2>, g2[i], dim3(kThreads), 0, stream, c, lev, level_l2e(lev));
        else hipLaunchKernelGGL(emd_rows2_kernel<1>, g2[i], dim3(kThreads), 0, stream, c, lev, level_l2e(lev));
    }
    static constexpr int kSteps = 2 + 2 * kLevels;
    void step(int s) const {
        if (s == 0) {
            if (cull > 0) {
                const int lp = std::max(logpL, logpR);
                const size_t lds = ((size_t)18 << lp) + kCells * 4;    // per point x, y, z + cell, slot, order (u16); the cell histogram
                if (lds > 48 * 1024)
                    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(emd_order_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
                hipLaunchKernelGGL(emd_order_kernel, dim3(2, b), dim3(kOrderThreads), lds, stream, c, multiL, multiR, logpL, logpR);
            } else {
                hipLaunchKernelGGL(emd_init_kernel, ginit, dim3(256), 0, stream, c, multiL, multiR);
            }
        } else if (s == 1) {
            rows1<false, true>(0, 0);
        } else {
            const int lev = (s - 2) >> 1;
            if (((s - 2) & 1) == 0) rows2(lev);
            else if (lev + 1 < kLevels) rows1<true, true>(lev, lev + 1);
            else if (final_remainL) rows1<true, false>(lev, lev);
        }
    }
};

int run_levels(int b, int n, int m, const float* xyz1, const float* xyz2, float* temp, float* ws, Ctx* out, hipStream_t stream,
               bool final_remainL, int cull) {
    const LevelChain ch(b, n, m, xyz1, xyz2, temp, ws, stream, final_remainL, cull);
    for (int s = 0; s < LevelChain::kSteps; ++s) ch.step(s);
    *out = ch.c;
    return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// Workspace-free path: the reference launcher's exact prototype (structural_loss.cpp:11), for callers that own
// nothing but `match` and `temp`.  It keeps the reference's data flow (approxmatch.cu:34-213): the four vectors of
// `temp` are the only state, every phase re-reads the other set's points and one of those vectors through a 16 KB LDS
// tile, phase 3 read-modify-writes `match` once per level.  One lane per row; a row's sum runs over the candidates in
// ascending order in one fp32 accumulator — the reference's per-thread order (:77-93,:125-137,:177-189), which also
// makes this path comparable with the CPU oracle term by term.  ~2.5x the time of the record-based path at B=64,
// N=2048 (it moves the 18 GB the reference moves); a binding that can allocate scratch calls hp_approxmatch_ws.
// ------------------------------------------------------------------------------------------------
constexpr int kPlainTile = 1024;

__global__ __launch_bounds__(kThreads) void emd_plain_init_kernel(int n, int m, float* __restrict__ temp, float multiL, float multiR) {
    float* remL = temp + (long)blockIdx.y * (n + m) * 2;
    for (int i = blockIdx.x * kThreads + threadIdx.x; i < n + m; i += gridDim.x * kThreads) remL[i] = i < n ? multiL : multiR;
}

// PHASE 1: rows = set1, ratioL[k] = remainL[k] / (1e-9 + sum_l e * remainR[l])                     (:60-93)
// PHASE 2: rows = set2, sumr = remainR[l] * sum_k e * ratioL[k]; ratioR / remainR update            (:109-142)
// PHASE 3: rows = set1, w = (e * ratioL[k]) * ratioR[l]; match[l,k] += w; remainL[k] -= sum_l w    (:161-194)
template <int PHASE>
__global__ __launch_bounds__(kThreads) void emd_plain_kernel(int n, int m, const float* __restrict__ xyz1, const float* __restrict__ xyz2,
                                                             float* __restrict__ temp, float* __restrict__ match, float l2e, int first) {
    __shared__ float4 tile[kPlainTile];
    const int cloud = blockIdx.y, tid = threadIdx.x;
    float* remL = temp + (long)cloud * (n + m) * 2;
    float* remR = remL + n;
    float* ratioL = remR + m;
    float* ratioR = ratioL + n;
    constexpr bool kRowsL = PHASE != 2;
    const int rows = kRowsL ? n : m, cands = kRowsL ? m : n;
    const float* R = (kRowsL ? xyz1 : xyz2) + (long)cloud * rows * 3;
    const float* C = (kRowsL ? xyz2 : xyz1) + (long)cloud * cands * 3;
    const float* cw = PHASE == 1 ? remR : PHASE == 2 ? ratioL : ratioR;
    const int row = blockIdx.x * kThreads + tid;
    const bool ok = row < rows;
    float rx = 0.f, ry = 0.f, rz = 0.f, rl = 0.f;
    if (ok) {
        rx = R[row * 3];
        ry = R[row * 3 + 1];
        rz = R[row * 3 + 2];
        if (PHASE == 3) rl = ratioL[row];
    }
    float sum = PHASE == 1 ? 1e-9f : 0.f;
    float* mcol = match + (long)cloud * m * n + row;
    for (int c0 = 0; c0 < cands; c0 += kPlainTile) {
        const int cnt = min(kPlainTile, cands - c0);
        for (int t = tid; t < cnt; t += kThreads) {
            const float* s = C + (long)(c0 + t) * 3;
            tile[t] = make_float4(s[0], s[1], s[2], cw[c0 + t]);
        }
        __syncthreads();
        if (ok) {
#pragma unroll 4
            for (int t = 0; t < cnt; ++t) {
                const float4 c = tile[t];
                // (x2 - x1) with x2 the set2 point in every phase (approxmatch.cu:85,131,185)
                const float d = kRowsL ? hp::sqdist(c.x - rx, c.y - ry, c.z - rz) : hp::sqdist(rx - c.x, ry - c.y, rz - c.z);
                const float e = __builtin_amdgcn_exp2f(l2e * d);
                // products enter the sums through an fma (nvcc's default -fmad=true contraction of the reference source;
                // oracle `contract` variant 3), as in the packed-record sweeps
                if (PHASE == 3) {
                    const float er = e * rl;
                    float* mp = mcol + (long)(c0 + t) * n;
                    *mp = first ? er * c.w : __builtin_fmaf(er, c.w, *mp);   // level 0 writes (zero fill + first `+=`)
                    sum = __builtin_fmaf(er, c.w, sum);
                } else {
                    sum = __builtin_fmaf(e, c.w, sum);
                }
            }
        }
        __syncthreads();
    }
    if (!ok) return;
    if (PHASE == 1) {
        ratioL[row] = remL[row] / sum;
    } else if (PHASE == 2) {
        const float rr = remR[row];
        const float sumr = sum * rr;
        const float consumption = fminf(rr / (sumr + 1e-9f), 1.0f);
        ratioR[row] = consumption * rr;
        remR[row] = fmaxf(0.0f, rr - sumr);
    } else {
        remL[row] = fmaxf(0.0f, remL[row] - sum);
    }
}

}  // namespace

// ---- tuning hook (no counterpart in the reference) -------------------------------------------------
// Rows per lane of the three sweep families: rows1 (phase 3 + phase 1 kernel) in {0,1,2,4}, rows2 (phase 2) in
// {0,1,2,4}, grad2 (final cost/gradient sweep) in {0,1,2}; 0 = the size heuristic.  Process-wide; every instance
// evaluates each row with the same operations in the same order, so results do not depend on the setting.
HP_API int hp_emd_set_rows_per_lane(int rows1, int rows2, int grad2) {
    auto okv = [](int v, bool four) { return v == 0 || v == 1 || v == 2 || (four && v == 4); };
    HP_CHECK_ARG(okv(rows1, true) && okv(rows2, true) && okv(grad2, false));
    g_rows1.store(rows1);
    g_rows2.store(rows2);
    g_grad2.store(grad2);
    return 0;
}

// The training path's final cost / gradient sweep with four of its nine exponentials per pair derived (1, default) or all nine
// from v_exp_f32 (0); returns the previous setting.  `match` as hp_approxmatch* return it is never derived.
HP_API int hp_emd_set_final_derive(int on) { return g_final_derive.exchange(on != 0); }

// hp_emd_forward* / hp_emd_forward_acc: the records in k-d order and the sweeps of the first `levels` annealing levels skipping the
// (64-row tile, 8-candidate block) units whose terms are all exactly zero (emd_order_kernel, emd_rows*_cull_kernel); 0 = the
// caller's point order, every unit evaluated (rounds 1-5).  Default 3 (HP_EMD_CULL at load time).  Returns the previous setting.
HP_API int hp_emd_set_cull(int levels) {
    HP_CHECK_ARG(levels >= 0 && levels <= kLevels);
    return g_cull.exchange(levels);
}

// hp_emd_forward* as two chains of half the clouds on two streams (2, default) or as one chain (1); returns the previous setting.
HP_API int hp_emd_set_chains(int chains) {
    HP_CHECK_ARG(chains >= 1 && chains <= kMaxChains);
    return g_chains.exchange(chains);
}

// replaces approxmatch(...)  structural_loss.cpp:11 / approxmatch.cu:330-338 — the reference's exact argument list:
// match (b,m,n) and temp (b,2(n+m)) are the only buffers.  temp ends as cloud i's [remainL | remainR | ratioL | ratioR].
HP_API int hp_approxmatch(int b, int n, int m, const float* xyz1, const float* xyz2, float* match, float* temp, hipStream_t stream) {
    HP_CHECK_ARG(b >= 0 && n > 0 && m > 0);
    if (b == 0) return 0;
    HP_CHECK_ARG(xyz1 && xyz2 && match && temp && b <= 65535);
    float multiL, multiR;
    if (n >= m) {
        multiL = 1;
        multiR = (float)(n / m);  // integer division (approxmatch.cu:37-43)
    } else {
        multiL = (float)(m / n);
        multiR = 1;
    }
    const dim3 blk(kThreads), gL((n + kThreads - 1) / kThreads, b), gR((m + kThreads - 1) / kThreads, b);
    hipLaunchKernelGGL(emd_plain_init_kernel, dim3((n + m + kThreads - 1) / kThreads, b), blk, 0, stream, n, m, temp, multiL, multiR);
    for (int lev = 0; lev < kLevels; ++lev) {
        const float l2e = level_l2e(lev);
        hipLaunchKernelGGL(emd_plain_kernel<1>, gL, blk, 0, stream, n, m, xyz1, xyz2, temp, match, l2e, 0);
        hipLaunchKernelGGL(emd_plain_kernel<2>, gR, blk, 0, stream, n, m, xyz1, xyz2, temp, match, l2e, 0);
        hipLaunchKernelGGL(emd_plain_kernel<3>, gL, blk, 0, stream, n, m, xyz1, xyz2, temp, match, l2e, lev == 0);
    }
    HP_RETURN_LAST_ERROR();
}

// floats of scratch hp_approxmatch_ws / hp_emd_forward need besides `temp` (packed candidate records)
HP_API long hp_approxmatch_workspace_floats(int b, int n, int m) { return (long)b * ws_layout(n, m).per_cloud; }

// The same result through the packed-record sweeps (the fast path): `ws` is scratch of
// hp_approxmatch_workspace_floats floats; `match` is written once instead of read-modify-written nine times.
HP_API int hp_approxmatch_ws(int b, int n, int m, const float* xyz1, const float* xyz2, float* match, float* temp, float* ws,
                             hipStream_t stream) {
    HP_CHECK_ARG(b >= 0 && n > 0 && m > 0);
    if (b == 0) return 0;
    HP_CHECK_ARG(xyz1 && xyz2 && match && temp && ws && b <= 65535);
    Ctx c;
    int rc = run_levels(b, n, m, xyz1, xyz2, temp, ws, &c, stream, true, 0);   // `match` and `temp` leave in the caller's order
    if (rc) return rc;
    hipLaunchKernelGGL(emd_match_kernel, dim3((n + kThreads - 1) / kThreads, (m + kLT - 1) / kLT, b), dim3(kThreads), 0, stream, c,
                       match);
    HP_RETURN_LAST_ERROR();
}

// Match-free EMD forward (what match_cost's forward = ApproxMatch + MatchCost computes, match_cost.py:9-27):
// cost (b,) and, as a by-product of the same sweep, grad1 = d cost / d xyz1 (b,n,3) (may be NULL).
// `ws` keeps the packed records for hp_emd_backward; partials: b*ceil(n/256) floats.
HP_API long hp_emd_partials_floats(int b, int n, int m) { return (long)b * ((std::max(n, m) + kRowsPerWg - 1) / kRowsPerWg); }

// grad1 / grad2 (either may be NULL): gradients to produce in the same call.  With grad2 != NULL the cost rides on the
// grad2 sweep (one evaluation of the match entries serves both); grad1 then costs a second sweep only if requested.
namespace {
int emd_forward_impl(int b, int n, int m, const float* xyz1, const float* xyz2, float* temp, float* ws, float* partials,
                     float* cost, float* grad1, float* grad2, float acc_scale, hipStream_t stream, hipStream_t after);
}
HP_API int hp_emd_forward(int b, int n, int m, const float* xyz1, const float* xyz2, float* temp, float* ws, float* partials,
                          float* cost, float* grad1, float* grad2, hipStream_t stream) {
    return emd_forward_impl(b, n, m, xyz1, xyz2, temp, ws, partials, cost, grad1, grad2, 0.f, stream, nullptr);
}

// The training step's form: grad2_acc (b,m,3) already holds the other loss terms' gradient with respect to xyz2 (the Chamfer
// term, written on stream `after`, or NULL = same stream) and receives  += scale * d cost / d xyz2  from the gradient sweep
// itself — the caller's separate `g_rec.add_(g_emd, alpha=scale)` launch (core/epoch_loops.py:26-31 forms the sum through
// autograd) is gone.  The sweep launch is ordered behind everything enqueued on `after` so far; scale != 0.
HP_API int hp_emd_forward_acc(int b, int n, int m, const float* xyz1, const float* xyz2, float* temp, float* ws, float* partials,
                              float* cost, float* grad2_acc, float scale, hipStream_t stream, hipStream_t after) {
    HP_CHECK_ARG(grad2_acc && scale != 0.f);
    return emd_forward_impl(b, n, m, xyz1, xyz2, temp, ws, partials, cost, nullptr, grad2_acc, scale, stream, after);
}

namespace {
// ONE chain: the 18 level sweeps + the final sweep of `b` clouds, launch behind launch on `stream`.
int emd_final_sweep(Ctx c, int b, float* partials, float* cost, float* grad1, float* grad2, float acc_scale, hipStream_t stream,
                    hipStream_t after);
int emd_forward_chain(int b, int n, int m, const float* xyz1, const float* xyz2, float* temp, float* ws, float* partials,
                      float* cost, float* grad1, float* grad2, float acc_scale, hipStream_t stream, hipStream_t after) {
    Ctx c;
    int rc = run_levels(b, n, m, xyz1, xyz2, temp, ws, &c, stream, false, g_cull.load(std::memory_order_relaxed));   // temp is scratch here
    if (rc) return rc;
    return emd_final_sweep(c, b, partials, cost, grad1, grad2, acc_scale, stream, after);
}
// the cost / gradient sweep(s) of one chain behind its level sweeps
// the final sweep's tiers: underflow radii of levels 1 and 2 where those levels cull (3e38: tier off; HP_EMD_FINAL_TIERS=0: both off)
void final_tiers(float& thr1, float& thr2) {
    static const bool on = [] { const char* e = getenv("HP_EMD_FINAL_TIERS"); return !(e && atoi(e) == 0); }();
    const int cull = on ? g_cull.load(std::memory_order_relaxed) : 0;
    thr1 = cull > 1 ? 152.f / -level_l2e(1) : 3.0e38f;
    thr2 = cull > 2 ? 152.f / -level_l2e(2) : 3.0e38f;
}
int emd_final_sweep(Ctx c, int b, float* partials, float* cost, float* grad1, float* grad2, float acc_scale, hipStream_t stream,
                    hipStream_t after) {
    const int n = c.n, m = c.m;
    float thr1, thr2;
    final_tiers(thr1, thr2);
    int rc = 0;
    c.acc_scale = acc_scale;
    if (after && after != stream) {   // the accumulated-into gradient was written on `after`: order the sweep behind it
        rc = hp_order_streams(after, stream);
        if (rc) return rc;
    }
    const int nb = (n + kRowsPerWg - 1) / kRowsPerWg, mb = (m + kRowsPerWg - 1) / kRowsPerWg;
    const bool derive = g_final_derive.load(std::memory_order_relaxed) != 0;
    if (grad2) {
        const int genv = g_grad2.load(std::memory_order_relaxed);
        const int mbr = (m + 2 * kRowsPerWg - 1) / (2 * kRowsPerWg);
        // two rows per lane when that still leaves >= 2 waves per SIMD (as in run_levels; -0.01 ms at B=64, N=2048)
        const int gr = genv ? genv : ((long)b * mbr * (kThreads / 64) >= 2048 ? 2 : 1);
        if (gr == 2) {
            if (derive) hipLaunchKernelGGL((emd_grad2_kernel<true, 2, true>), dim3(mbr, b), dim3(kThreads), 0, stream, c, grad2, partials, thr1, thr2);
            else hipLaunchKernelGGL((emd_grad2_kernel<true, 2, false>), dim3(mbr, b), dim3(kThreads), 0, stream, c, grad2, partials, thr1, thr2);
            hipLaunchKernelGGL(emd_cost_finish_kernel, dim3(b), dim3(256), 0, stream, partials, mbr, cost);
        } else {
            if (derive) hipLaunchKernelGGL((emd_grad2_kernel<true, 1, true>), dim3(mb, b), dim3(kThreads), 0, stream, c, grad2, partials, thr1, thr2);
            else hipLaunchKernelGGL((emd_grad2_kernel<true, 1, false>), dim3(mb, b), dim3(kThreads), 0, stream, c, grad2, partials, thr1, thr2);
            hipLaunchKernelGGL(emd_cost_finish_kernel, dim3(b), dim3(256), 0, stream, partials, mb, cost);
        }
        // (the second sweep's partials are unused: cost was already reduced, in stream order, by the finish kernel)
        if (grad1) {
            if (derive) hipLaunchKernelGGL(emd_cost_grad1_kernel<true>, dim3(nb, b), dim3(kThreads), 0, stream, c, partials, grad1);
            else hipLaunchKernelGGL(emd_cost_grad1_kernel<false>, dim3(nb, b), dim3(kThreads), 0, stream, c, partials, grad1);
        }
    } else {
        if (derive) hipLaunchKernelGGL(emd_cost_grad1_kernel<true>, dim3(nb, b), dim3(kThreads), 0, stream, c, partials, grad1);
        else hipLaunchKernelGGL(emd_cost_grad1_kernel<false>, dim3(nb, b), dim3(kThreads), 0, stream, c, partials, grad1);
        hipLaunchKernelGGL(emd_cost_finish_kernel, dim3(b), dim3(256), 0, stream, partials, nb, cost);
    }
    HP_RETURN_LAST_ERROR();
}


// Round 5: the clouds are independent, but a launch is not — each of the 19 dependent launches pays its ramp, prologue, epilogue
// and tail with every wave of the chip in lockstep (at B = 64 the grid is ONE round of workgroups), ~8 us per launch that the
// same launches on several times the clouds amortise (tools/emd_launch_bound.py: 1.371 ms at 64 clouds, 1.217 per 64 at 576).
// So the call runs as TWO chains of half the clouds on two streams: while one chain's launch ramps up or drains, the other's
// waves hold the vector pipes — what a persistent per-cloud kernel would give, without a flag hand-off and without anything that
// could wait for a workgroup the dispatcher has not placed (docs/DESIGN_HISTORY.md 7b).  Per cloud the arithmetic is that of one chain
// (the halves may run other rows-per-lane instances, which are bit-identical per row): match-free cost within the partials'
// regrouping (2e-6, as between instances), gradients identical.  Measured at B = 64, N = 2048: 1.370 -> 1.280 ms per call.
// The second stream is the library's own, one per device, created on first use (non-blocking, high priority: its own hardware
// queue).  HP_EMD_CHAINS=1 / hp_emd_set_chains(1): one chain (rounds 1-4).

// The library's own stream number `i` (1..kMaxChains-1) on the device `stream` belongs to (the null stream: the current device).
// Created on first use on THAT device — the calling thread's current device may be another one — non-blocking, high
// priority: its own hardware queue.
hipStream_t chain_stream(hipStream_t stream, int i) {
    int cur = 0, dev = 0;
    if (hipGetDevice(&cur) != hipSuccess) return nullptr;
    dev = cur;
    if (stream && hipStreamGetDevice(stream, &dev) != hipSuccess) {
        (void)hipGetLastError();
        dev = cur;
    }
    std::lock_guard<std::mutex> lock(g_s2_mu);
    auto it = g_s2_streams.find(dev * kMaxChains + i);
    if (it != g_s2_streams.end()) return it->second;
    hipStream_t s = nullptr;
    if (dev == cur || hipSetDevice(dev) == hipSuccess) {
        int lo = 0, hi = 0;
        (void)hipDeviceGetStreamPriorityRange(&lo, &hi);      // (numerically lowest = highest priority)
        if (hipStreamCreateWithPriority(&s, hipStreamNonBlocking, hi) != hipSuccess) {
            (void)hipGetLastError();
            s = nullptr;
        }
        if (dev != cur) (void)hipSetDevice(cur);
    } else {
        (void)hipGetLastError();
    }
    g_s2_streams[dev * kMaxChains + i] = s;
    return s;
}

int emd_forward_impl(int b, int n, int m, const float* xyz1, const float* xyz2, float* temp, float* ws, float* partials,
                     float* cost, float* grad1, float* grad2, float acc_scale, hipStream_t stream, hipStream_t after) {
    HP_CHECK_ARG(b >= 0 && n > 0 && m > 0);
    if (b == 0) return 0;
    HP_CHECK_ARG(xyz1 && xyz2 && temp && ws && partials && cost && b <= 65535);
    // several chains only where each part still fills the chip (>= 2 waves per SIMD at one row per lane) and a capture is not in
    // progress on the caller's stream (a captured call stays on the stream it was captured on)
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    (void)hipStreamIsCapturing(stream, &cap);
    int nch = cap == hipStreamCaptureStatusNone ? g_chains.load(std::memory_order_relaxed) : 1;
    while (nch > 1 && (long)(b / nch) * ((std::min(n, m) + kRowsPerWg - 1) / kRowsPerWg) * (kThreads / 64) < 2048) --nch;
    hipStream_t st[kMaxChains] = {stream};
    for (int i = 1; i < nch; ++i) {
        st[i] = chain_stream(stream, i);
        if (!st[i]) nch = 1;
    }
    if (nch <= 1) return emd_forward_chain(b, n, m, xyz1, xyz2, temp, ws, partials, cost, grad1, grad2, acc_scale, stream, after);
    const WsLayout L = ws_layout(n, m);
    const int cull = g_cull.load(std::memory_order_relaxed);
    int rc = 0;
    for (int i = 1; i < nch && !rc; ++i) rc = hp_order_streams(stream, st[i]);      // the inputs are ready on `stream`
    if (rc) return rc;      // (nothing is enqueued on the library's streams yet)
    std::vector<LevelChain> ch;
    ch.reserve(nch);
    for (int i = 0; i < nch; ++i) {
        const int c0 = (int)((long)b * i / nch), c1 = (int)((long)b * (i + 1) / nch);
        ch.emplace_back(c1 - c0, n, m, xyz1 + (long)c0 * n * 3, xyz2 + (long)c0 * m * 3, temp + (long)c0 * (n + m) * 2, ws + (long)c0 * L.per_cloud,
                        st[i], false, cull);
    }
    for (int s = 0; s < LevelChain::kSteps; ++s)      // alternately: all streams are fed at the same pace
        for (int i = 0; i < nch; ++i) ch[i].step(s);
    rc = (int)hipGetLastError();
    // the final sweep is ONE launch over all clouds again, behind every chain: it is a single long launch (nothing follows it that
    // could cover its tail), and as two half-size launches the second ran its last ~110 us alone on a half-empty chip
    // (173 + 284 us in the trace against 288 for the whole batch)
    // ... and behind `after` (the stream that wrote the gradient the sweep accumulates into).  Every wait costs the waiting stream a
    // bubble of several microseconds even when the event has long fired, so the caller's stream gets as few as possible: `after` is
    // joined into the last chain's stream first (nothing is queued there behind its chain), the library's streams into one
    // another, and the last one into the caller's.
    const bool ext = after && after != stream && acc_scale != 0.f;
    if (!rc && ext) rc = hp_order_streams(after, st[nch - 1]);
    // On a failure the library's streams are still joined into the caller's (best effort): the caller only knows `stream` and may
    // free or reuse temp / ws / the inputs behind it.
    for (int i = 1; i + 1 < nch; ++i) {
        const int r2 = hp_order_streams(st[i], st[i + 1]);
        if (!rc) rc = r2;
    }
    const int r3 = hp_order_streams(st[nch - 1], stream);
    if (!rc) rc = r3;
    if (rc) return rc;
    Ctx call = ch[0].c;      // the whole batch: chain 0 starts at cloud 0
    return emd_final_sweep(call, b, partials, cost, grad1, grad2, acc_scale, stream, ext ? nullptr : after);
}

}  // namespace

// grad2 = d cost / d xyz2 (b,m,3) from the records hp_emd_forward left in `ws` (match_cost.py:35-46 without match)
HP_API int hp_emd_backward(int b, int n, int m, const float* xyz1, const float* xyz2, const float* ws, float* grad2,
                           hipStream_t stream) {
    HP_CHECK_ARG(b >= 0 && n > 0 && m > 0);
    if (b == 0) return 0;
    HP_CHECK_ARG(ws && grad2 && b <= 65535);
    Ctx c = make_ctx(n, m, xyz1, xyz2, nullptr, const_cast<float*>(ws));
    float thr1, thr2;
    final_tiers(thr1, thr2);
    // rows per lane as in the forward's final sweep (round 6: one row per lane at B = 64, N = 2048 took 465 us against ~250 at two)
    const int genv = g_grad2.load(std::memory_order_relaxed);
    const int mbr = (m + 2 * kRowsPerWg - 1) / (2 * kRowsPerWg);
    const bool two = (genv ? genv : ((long)b * mbr * (kThreads / 64) >= 2048 ? 2 : 1)) == 2;
    const dim3 grid(two ? mbr : (m + kRowsPerWg - 1) / kRowsPerWg, b);
    const bool derive = g_final_derive.load(std::memory_order_relaxed) != 0;
    if (two) {
        if (derive) hipLaunchKernelGGL((emd_grad2_kernel<false, 2, true>), grid, dim3(kThreads), 0, stream, c, grad2, nullptr, thr1, thr2);
        else hipLaunchKernelGGL((emd_grad2_kernel<false, 2, false>), grid, dim3(kThreads), 0, stream, c, grad2, nullptr, thr1, thr2);
    } else {
        if (derive) hipLaunchKernelGGL((emd_grad2_kernel<false, 1, true>), grid, dim3(kThreads), 0, stream, c, grad2, nullptr, thr1, thr2);
        else hipLaunchKernelGGL((emd_grad2_kernel<false, 1, false>), grid, dim3(kThreads), 0, stream, c, grad2, nullptr, thr1, thr2);
    }
    HP_RETURN_LAST_ERROR();
}
