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
//   * accumulation orders are the reference's (ascending candidate index per row), padding records
//     carry zero weights and contribute exact zeros.
#include "hp_common.h"
#include <algorithm>

namespace {

constexpr int kThreads = 256;
constexpr int kLevels = 9;
constexpr int kStage = 8;   // candidate records per software-pipeline stage of the phase kernels
constexpr int kSpare = 8;   // readable zero records past the padded range (the prefetch after the last stage)
constexpr float kLog2e = 1.4426950408889634f;

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x8 __attribute__((ext_vector_type(8)));

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

inline int pad_up(int x) { return (x + 2 * kStage - 1) / (2 * kStage) * (2 * kStage); }

// per-cloud workspace (floats): PL4[(NP+S)*4] | PR4[(MP+S)*4] | RR[MP+S] | FL16[(NP+S)*16] | FR16[(MP+S)*16]
//   PL4 = (p.xyz, ratioL)   PR4 = (q.xyz, ratioR)   RR = remainR   F*16 = (xyz, r[level 0..8], 0,0,0,0)
struct WsLayout {
    int NP, MP;
    long pl4, pr4, rr, fl16, fr16, per_cloud;
};
inline WsLayout ws_layout(int n, int m) {
    WsLayout w;
    w.NP = pad_up(n);
    w.MP = pad_up(m);
    w.pl4 = 0;
    w.pr4 = w.pl4 + (long)(w.NP + kSpare) * 4;
    w.rr = w.pr4 + (long)(w.MP + kSpare) * 4;
    w.fl16 = w.rr + (long)(w.MP + kSpare);
    w.fr16 = w.fl16 + (long)(w.NP + kSpare) * 16;
    w.per_cloud = w.fr16 + (long)(w.MP + kSpare) * 16;
    return w;
}

struct Ctx {
    int n, m, NP, MP;
    const float* xyz1;
    const float* xyz2;
    float* temp;  // (b, 2(n+m)) : [remainL n | remainR m | ratioL n | ratioR m]  (reference layout, approxmatch.cu:35)
    float* ws;
    long pl4, pr4, rr, fl16, fr16, per_cloud;
};

__global__ __launch_bounds__(kThreads) void emd_init_kernel(Ctx c, float multiL, float multiR) {
    const int cloud = blockIdx.y;
    float* ws = c.ws + (long)cloud * c.per_cloud;
    float* remL = c.temp + (long)cloud * (c.n + c.m) * 2;
    float* remR = remL + c.n;
    const float* P = c.xyz1 + (long)cloud * c.n * 3;
    const float* Q = c.xyz2 + (long)cloud * c.m * 3;
    const int NPs = c.NP + kSpare, MPs = c.MP + kSpare;
    const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int i = blockIdx.x * kThreads + threadIdx.x; i < NPs + MPs; i += gridDim.x * kThreads) {
        if (i < NPs) {
            const bool ok = i < c.n;
            const float4 r = ok ? make_float4(P[i * 3], P[i * 3 + 1], P[i * 3 + 2], 0.f) : zero;
            reinterpret_cast<float4*>(ws + c.pl4)[i] = r;
            float4* f = reinterpret_cast<float4*>(ws + c.fl16) + (long)i * 4;
            f[0] = r;
            f[1] = f[2] = f[3] = zero;
            if (ok) remL[i] = multiL;
        } else {
            const int l = i - NPs;
            const bool ok = l < c.m;
            const float4 r = ok ? make_float4(Q[l * 3], Q[l * 3 + 1], Q[l * 3 + 2], 0.f) : zero;
            reinterpret_cast<float4*>(ws + c.pr4)[l] = r;
            ws[c.rr + l] = ok ? multiR : 0.f;
            float4* f = reinterpret_cast<float4*>(ws + c.fr16) + (long)l * 4;
            f[0] = r;
            f[1] = f[2] = f[3] = zero;
            if (ok) remR[l] = multiR;
        }
    }
}

// component `c` of record `u` (0..7) of a stage held in two x16 SGPR groups
#define REC(lo, hi, u, c) ((u) < 4 ? (lo)[(u)*4 + (c)] : (hi)[((u)-4) * 4 + (c)])

// Rows = set1.  DO3: phase 3 of level lev3 (remainL update, approxmatch.cu:161-194);
//               DO1: phase 1 of level lev1 (ratioL, :60-93).  Candidates: PR4 (+RR) records on the scalar path.
template <bool DO3, bool DO1>
__global__ __launch_bounds__(kThreads) void emd_rows1_kernel(Ctx c, int lev1, float l2e3, float l2e1) {
    const int cloud = blockIdx.y;
    const int k = blockIdx.x * kThreads + threadIdx.x;
    float* ws = c.ws + (long)cloud * c.per_cloud;
    float* remL = c.temp + (long)cloud * (c.n + c.m) * 2;
    float* ratioL = remL + c.n + c.m;
    const bool ok = k < c.n;
    float px = 0.f, py = 0.f, pz = 0.f, rl = 0.f;
    if (ok) {
        const float* s = c.xyz1 + ((long)cloud * c.n + k) * 3;
        px = s[0];
        py = s[1];
        pz = s[2];
        if (DO3) rl = ratioL[k];
    }
    float acc3 = 0.f, acc1 = 1e-9f;
    const float* p = ws + c.pr4;   // wave-uniform
    const float* q = ws + c.rr;
    f32x16 a0, a1, b0, b1;
    f32x8 w0 = {}, w1 = {};
    auto work = [&](const f32x16& lo, const f32x16& hi, const f32x8& w) {
#pragma unroll
        for (int u = 0; u < kStage; ++u) {
            const float d = hp::sqdist(REC(lo, hi, u, 0) - px, REC(lo, hi, u, 1) - py, REC(lo, hi, u, 2) - pz);
            if (DO3) acc3 += (__builtin_amdgcn_exp2f(l2e3 * d) * rl) * REC(lo, hi, u, 3);   // e * ratioL[k] * ratioR[l]
            if (DO1) acc1 += __builtin_amdgcn_exp2f(l2e1 * d) * w[u];                      // e * remainR[l]
        }
    };
    HP_SLOAD16(a0, p, 0x0);
    HP_SLOAD16(a1, p, 0x40);
    if (DO1) HP_SLOAD8(w0, q, 0x0);
    asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(a0), "+s"(a1), "+s"(w0));
    for (int l0 = 0; l0 < c.MP; l0 += 2 * kStage) {
        p += kStage * 4;
        q += kStage;
        HP_SLOAD16(b0, p, 0x0);
        HP_SLOAD16(b1, p, 0x40);
        if (DO1) HP_SLOAD8(w1, q, 0x0);
        HP_PIN();
        work(a0, a1, w0);
        asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(b0), "+s"(b1), "+s"(w1), "+v"(acc3), "+v"(acc1));
        p += kStage * 4;
        q += kStage;
        HP_SLOAD16(a0, p, 0x0);      // past the last stage this reads the spare (zero) records
        HP_SLOAD16(a1, p, 0x40);
        if (DO1) HP_SLOAD8(w0, q, 0x0);
        HP_PIN();
        work(b0, b1, w1);
        asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(a0), "+s"(a1), "+s"(w0), "+v"(acc3), "+v"(acc1));
    }
    if (!ok) return;
    float rem = remL[k];
    if (DO3) {
        rem = fmaxf(0.0f, rem - acc3);
        remL[k] = rem;
    }
    if (DO1) {
        const float v = rem / acc1;
        ratioL[k] = v;
        ws[c.pl4 + (long)k * 4 + 3] = v;
        ws[c.fl16 + (long)k * 16 + 3 + lev1] = v;
    }
}

// Rows = set2: phase 2 (ratioR / remainR update, approxmatch.cu:109-142).  Candidates: PL4 records.
__global__ __launch_bounds__(kThreads) void emd_rows2_kernel(Ctx c, int lev, float l2e) {
    const int cloud = blockIdx.y;
    const int l = blockIdx.x * kThreads + threadIdx.x;
    float* ws = c.ws + (long)cloud * c.per_cloud;
    float* remR = c.temp + (long)cloud * (c.n + c.m) * 2 + c.n;
    float* ratioR = remR + c.m + c.n;
    const bool ok = l < c.m;
    float qx = 0.f, qy = 0.f, qz = 0.f;
    if (ok) {
        const float* s = c.xyz2 + ((long)cloud * c.m + l) * 3;
        qx = s[0];
        qy = s[1];
        qz = s[2];
    }
    float acc = 0.f;
    const float* p = ws + c.pl4;
    f32x16 a0, a1, b0, b1;
    auto work = [&](const f32x16& lo, const f32x16& hi) {
#pragma unroll
        for (int u = 0; u < kStage; ++u) {
            // the reference evaluates (x2-x1) with x2 the set2 point in every phase (approxmatch.cu:85,131,185)
            const float d = hp::sqdist(qx - REC(lo, hi, u, 0), qy - REC(lo, hi, u, 1), qz - REC(lo, hi, u, 2));
            acc += __builtin_amdgcn_exp2f(l2e * d) * REC(lo, hi, u, 3);
        }
    };
    HP_SLOAD16(a0, p, 0x0);
    HP_SLOAD16(a1, p, 0x40);
    asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(a0), "+s"(a1));
    for (int k0 = 0; k0 < c.NP; k0 += 2 * kStage) {
        p += kStage * 4;
        HP_SLOAD16(b0, p, 0x0);
        HP_SLOAD16(b1, p, 0x40);
        HP_PIN();
        work(a0, a1);
        asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(b0), "+s"(b1), "+v"(acc));
        p += kStage * 4;
        HP_SLOAD16(a0, p, 0x0);
        HP_SLOAD16(a1, p, 0x40);
        HP_PIN();
        work(b0, b1);
        asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(a0), "+s"(a1), "+v"(acc));
    }
    if (!ok) return;
    const float rr = remR[l];
    const float sumr = acc * rr;
    const float consumption = fminf(rr / (sumr + 1e-9f), 1.0f);
    const float v = consumption * rr;
    const float rem = fmaxf(0.0f, rr - sumr);
    ratioR[l] = v;
    remR[l] = rem;
    ws[c.pr4 + (long)l * 4 + 3] = v;
    ws[c.rr + l] = rem;
    ws[c.fr16 + (long)l * 16 + 3 + lev] = v;
}

// M(l,k) = sum over levels, in level order, of (exp(level*d) * ratioL_lev[k]) * ratioR_lev[l].
// `row` holds the lane's own final record in VGPRs, `cand` the candidate's record in SGPRs; ROW_IS_L says which
// of the two carries the ratioL values.
template <bool ROW_IS_L>
__device__ __forceinline__ float match_entry(float d, const float (&row)[kLevels], const f32x16& cand) {
    float acc = 0.f;
#pragma unroll
    for (int lev = 0; lev < kLevels; ++lev) {
        const float e = __builtin_amdgcn_exp2f(level_l2e(lev) * d);
        acc += ROW_IS_L ? (e * row[lev]) * cand[3 + lev] : (e * cand[3 + lev]) * row[lev];
    }
    return acc;
}

constexpr int kLT = 64;  // match rows (l) per workgroup in the materialising pass
__global__ __launch_bounds__(kThreads) void emd_match_kernel(Ctx c, float* __restrict__ match) {
    const int cloud = blockIdx.z;
    const int k = blockIdx.x * kThreads + threadIdx.x;
    const int l0 = blockIdx.y * kLT;
    const float* ws = c.ws + (long)cloud * c.per_cloud;
    const bool ok = k < c.n;
    float px = 0.f, py = 0.f, pz = 0.f, rL[kLevels] = {};
    if (ok) {
        const float* fl = ws + c.fl16 + (long)k * 16;
        px = fl[0];
        py = fl[1];
        pz = fl[2];
#pragma unroll
        for (int lev = 0; lev < kLevels; ++lev) rL[lev] = fl[3 + lev];
    }
    float* out = match + ((long)cloud * c.m + l0) * c.n + k;
    const int cnt = min(kLT, c.m - l0);          // kLT and MP are multiples of 2, spare records exist past MP
    const float* p = ws + c.fr16 + (long)l0 * 16;
    f32x16 a, b;
    HP_SLOAD16(a, p, 0x0);
    asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(a));
    for (int l = 0; l < cnt; l += 2) {
        HP_SLOAD16(b, p, 0x40);
        HP_PIN();
        float v = match_entry<true>(hp::sqdist(a[0] - px, a[1] - py, a[2] - pz), rL, a);
        if (ok) out[(long)l * c.n] = v;
        asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(b), "+v"(v));
        p += 32;
        HP_SLOAD16(a, p, 0x0);
        HP_PIN();
        v = match_entry<true>(hp::sqdist(b[0] - px, b[1] - py, b[2] - pz), rL, b);
        if (ok && l + 1 < cnt) out[(long)(l + 1) * c.n] = v;
        asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(a), "+v"(v));
    }
}

// match-free cost + grad1:  cost_b = sum_{k,l} M(l,k) sqrt(d),  grad1[k] = sum_l M(l,k) (p_k-q_l)/max(|p_k-q_l|,1e-10)
// (approxmatch.cu:215-255, 301-322 without the match tensor).  One lane per k, all l on the scalar path.
__global__ __launch_bounds__(kThreads) void emd_cost_grad1_kernel(Ctx c, float* __restrict__ partials, float* __restrict__ grad1) {
    __shared__ float red[kThreads / 64];
    const int cloud = blockIdx.y;
    const int k = blockIdx.x * kThreads + threadIdx.x;
    const float* ws = c.ws + (long)cloud * c.per_cloud;
    const bool ok = k < c.n;
    float px = 0.f, py = 0.f, pz = 0.f, rL[kLevels] = {};
    if (ok) {
        const float* fl = ws + c.fl16 + (long)k * 16;
        px = fl[0];
        py = fl[1];
        pz = fl[2];
#pragma unroll
        for (int lev = 0; lev < kLevels; ++lev) rL[lev] = fl[3 + lev];
    }
    float cost = 0.f, dx = 0.f, dy = 0.f, dz = 0.f;
    auto work = [&](const f32x16& r) {
        const float ex = px - r[0], ey = py - r[1], ez = pz - r[2];   // (x1 - x2), approxmatch.cu:312
        const float d2 = hp::sqdist(r[0] - px, r[1] - py, r[2] - pz);
        const float mv = match_entry<true>(d2, rL, r);
        cost = __builtin_fmaf(mv, __builtin_sqrtf(d2), cost);
        const float w = mv * __builtin_amdgcn_rsqf(fmaxf(d2, 1e-20f));
        dx = __builtin_fmaf(ex, w, dx);
        dy = __builtin_fmaf(ey, w, dy);
        dz = __builtin_fmaf(ez, w, dz);
    };
    const float* p = ws + c.fr16;
    f32x16 a, b;
    HP_SLOAD16(a, p, 0x0);
    asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(a));
    for (int l = 0; l < c.MP; l += 2) {
        HP_SLOAD16(b, p, 0x40);
        HP_PIN();
        work(a);
        asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(b), "+v"(cost), "+v"(dx), "+v"(dy), "+v"(dz));
        p += 32;
        HP_SLOAD16(a, p, 0x0);
        HP_PIN();
        work(b);
        asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(a), "+v"(cost), "+v"(dx), "+v"(dy), "+v"(dz));
    }
    if (ok && grad1) {
        float* g = grad1 + ((long)cloud * c.n + k) * 3;
        g[0] = dx;
        g[1] = dy;
        g[2] = dz;
    }
    const float t = hp::block_sum(ok ? cost : 0.f, red);
    if (threadIdx.x == 0) partials[(long)cloud * gridDim.x + blockIdx.x] = t;
}

// match-free grad2[l] = sum_k M(l,k) (q_l-p_k)/max(|q_l-p_k|,1e-10)   (approxmatch.cu:260-300)
__global__ __launch_bounds__(kThreads) void emd_grad2_kernel(Ctx c, float* __restrict__ grad2) {
    const int cloud = blockIdx.y;
    const int l = blockIdx.x * kThreads + threadIdx.x;
    const float* ws = c.ws + (long)cloud * c.per_cloud;
    const bool ok = l < c.m;
    float qx = 0.f, qy = 0.f, qz = 0.f, rR[kLevels] = {};
    if (ok) {
        const float* fr = ws + c.fr16 + (long)l * 16;
        qx = fr[0];
        qy = fr[1];
        qz = fr[2];
#pragma unroll
        for (int lev = 0; lev < kLevels; ++lev) rR[lev] = fr[3 + lev];
    }
    float sx = 0.f, sy = 0.f, sz = 0.f;
    auto work = [&](const f32x16& r) {
        const float ex = qx - r[0], ey = qy - r[1], ez = qz - r[2];
        const float d2 = hp::sqdist(ex, ey, ez);
        const float mv = match_entry<false>(d2, rR, r);
        const float w = mv * __builtin_amdgcn_rsqf(fmaxf(d2, 1e-20f));
        sx = __builtin_fmaf(ex, w, sx);
        sy = __builtin_fmaf(ey, w, sy);
        sz = __builtin_fmaf(ez, w, sz);
    };
    const float* p = ws + c.fl16;
    f32x16 a, b;
    HP_SLOAD16(a, p, 0x0);
    asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(a));
    for (int k = 0; k < c.NP; k += 2) {
        HP_SLOAD16(b, p, 0x40);
        HP_PIN();
        work(a);
        asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(b), "+v"(sx), "+v"(sy), "+v"(sz));
        p += 32;
        HP_SLOAD16(a, p, 0x0);
        HP_PIN();
        work(b);
        asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(a), "+v"(sx), "+v"(sy), "+v"(sz));
    }
    if (!ok) return;
    float* g = grad2 + ((long)cloud * c.m + l) * 3;
    g[0] = sx;
    g[1] = sy;
    g[2] = sz;
}

__global__ __launch_bounds__(256) void emd_cost_finish_kernel(const float* __restrict__ partials, int per_cloud, float* __restrict__ out) {
    __shared__ double red[4];
    const float* p = partials + (long)blockIdx.x * per_cloud;
    double s = 0;
    for (int i = threadIdx.x; i < per_cloud; i += 256) s += (double)p[i];
    const double t = hp::block_sum(s, red);
    if (threadIdx.x == 0) out[blockIdx.x] = (float)t;
}

int run_levels(int b, int n, int m, const float* xyz1, const float* xyz2, float* temp, float* ws, Ctx* out, hipStream_t stream) {
    const WsLayout L = ws_layout(n, m);
    Ctx c{n, m, L.NP, L.MP, xyz1, xyz2, temp, ws, L.pl4, L.pr4, L.rr, L.fl16, L.fr16, L.per_cloud};
    float multiL, multiR;
    if (n >= m) {
        multiL = 1;
        multiR = (float)(n / m);  // integer division (approxmatch.cu:37-43)
    } else {
        multiL = (float)(m / n);
        multiR = 1;
    }
    const dim3 g1((n + kThreads - 1) / kThreads, b), g2((m + kThreads - 1) / kThreads, b);
    hipLaunchKernelGGL(emd_init_kernel, dim3((L.NP + L.MP + 2 * kSpare + kThreads - 1) / kThreads, b), dim3(kThreads), 0, stream, c, multiL, multiR);
    hipLaunchKernelGGL((emd_rows1_kernel<false, true>), g1, dim3(kThreads), 0, stream, c, 0, 0.f, level_l2e(0));
    for (int lev = 0; lev < kLevels; ++lev) {
        hipLaunchKernelGGL(emd_rows2_kernel, g2, dim3(kThreads), 0, stream, c, lev, level_l2e(lev));
        if (lev + 1 < kLevels)
            hipLaunchKernelGGL((emd_rows1_kernel<true, true>), g1, dim3(kThreads), 0, stream, c, lev + 1, level_l2e(lev),
                               level_l2e(lev + 1));
        else
            hipLaunchKernelGGL((emd_rows1_kernel<true, false>), g1, dim3(kThreads), 0, stream, c, lev, level_l2e(lev), 0.f);
    }
    *out = c;
    return (int)hipGetLastError();
}

}  // namespace

// floats of scratch hp_approxmatch / hp_emd_forward need besides `temp` (packed candidate records)
HP_API long hp_approxmatch_workspace_floats(int b, int n, int m) { return (long)b * ws_layout(n, m).per_cloud; }

// replaces approxmatch(...)  structural_loss.cpp:11 / approxmatch.cu:330-338.
// match (b,m,n) and temp (b,2(n+m)) as in the reference; `ws` is extra scratch (see header).
HP_API int hp_approxmatch(int b, int n, int m, const float* xyz1, const float* xyz2, float* match, float* temp, float* ws,
                          hipStream_t stream) {
    HP_CHECK_ARG(b >= 0 && n > 0 && m > 0);
    if (b == 0) return 0;
    HP_CHECK_ARG(xyz1 && xyz2 && match && temp && ws && b <= 65535);
    Ctx c;
    int rc = run_levels(b, n, m, xyz1, xyz2, temp, ws, &c, stream);
    if (rc) return rc;
    hipLaunchKernelGGL(emd_match_kernel, dim3((n + kThreads - 1) / kThreads, (m + kLT - 1) / kLT, b), dim3(kThreads), 0, stream, c,
                       match);
    HP_RETURN_LAST_ERROR();
}

// Match-free EMD forward (what match_cost's forward = ApproxMatch + MatchCost computes, match_cost.py:9-27):
// cost (b,) and, as a by-product of the same sweep, grad1 = d cost / d xyz1 (b,n,3) (may be NULL).
// `ws` keeps the packed records for hp_emd_backward; partials: b*ceil(n/256) floats.
HP_API long hp_emd_partials_floats(int b, int n) { return (long)b * ((n + kThreads - 1) / kThreads); }

HP_API int hp_emd_forward(int b, int n, int m, const float* xyz1, const float* xyz2, float* temp, float* ws, float* partials,
                          float* cost, float* grad1, hipStream_t stream) {
    HP_CHECK_ARG(b >= 0 && n > 0 && m > 0);
    if (b == 0) return 0;
    HP_CHECK_ARG(xyz1 && xyz2 && temp && ws && partials && cost && b <= 65535);
    Ctx c;
    int rc = run_levels(b, n, m, xyz1, xyz2, temp, ws, &c, stream);
    if (rc) return rc;
    const int nb = (n + kThreads - 1) / kThreads;
    hipLaunchKernelGGL(emd_cost_grad1_kernel, dim3(nb, b), dim3(kThreads), 0, stream, c, partials, grad1);
    hipLaunchKernelGGL(emd_cost_finish_kernel, dim3(b), dim3(256), 0, stream, partials, nb, cost);
    HP_RETURN_LAST_ERROR();
}

// grad2 = d cost / d xyz2 (b,m,3) from the records hp_emd_forward left in `ws` (match_cost.py:35-46 without match)
HP_API int hp_emd_backward(int b, int n, int m, const float* xyz1, const float* xyz2, const float* ws, float* grad2,
                           hipStream_t stream) {
    HP_CHECK_ARG(b >= 0 && n > 0 && m > 0);
    if (b == 0) return 0;
    HP_CHECK_ARG(ws && grad2 && b <= 65535);
    const WsLayout L = ws_layout(n, m);
    Ctx c{n, m, L.NP, L.MP, xyz1, xyz2, nullptr, const_cast<float*>(ws), L.pl4, L.pr4, L.rr, L.fl16, L.fr16, L.per_cloud};
    hipLaunchKernelGGL(emd_grad2_kernel, dim3((m + kThreads - 1) / kThreads, b), dim3(kThreads), 0, stream, c, grad2);
    HP_RETURN_LAST_ERROR();
}
