// Skinny-M layer programs on gfx950 — the M = B <= 64 chains of the HyperPocket step:
//   hypernetwork trunk, forward and backward   /root/reference/model/hyper_network.py:16-30 (+ the autograd of it)
//   encoder fc / mu / std tail                  /root/reference/model/encoder.py:30-36,46-53 (+ the autograd of it)
// With 64 rows these layers hold almost no arithmetic (0.35 GFLOP for the whole trunk against 11 MB of weights): as
// tiled GEMMs they were ~26 dependent launches of 5-10 us each, every one a latency chain (a k-loop of round trips to
// memory, then a split-K reduce launch).  Here a layer is ONE launch ("phase") and every launch is built around memory
// latency instead of tiles: all workgroups take (output strip x contraction range) tasks, issue every global load of a
// task up front (two 64/128-deep chunks in flight: a task pays about one memory latency), contract on the matrix cores
// (v_mfma_f32_32x32x2_f32, exact fp32) and leave raw partial slabs; the next phase FINISHES its input while loading it
// (slab sum in range order + bias + ReLU, or the ReLU mask of the backward) — no reduce launches, no atomics,
// run-to-run identical.  Three task shapes:
//   F   out(M x N) = A(M x K) W(N x K)^T : both operands K-contiguous -> staged through LDS with coalesced 16-byte
//       loads along k, fragments read back with the k-permuted ds_read_b128 of gemm.hip
//   X   out(M x K) = A(M x N) W(N x K)   : W rows run along the OUTPUT columns -> B fragments straight from global
//       memory (128-byte segments), A staged like F
//   W   out(N x K) = A(M x N)^T B(M x K) : contraction over the clouds, both operands contiguous along the lanes ->
//       no LDS at all; the bias gradient (column sums of A) rides on the A fragments
// Measured and dropped: the whole program as one persistent launch with a grid-wide barrier between phases.  The 8 XCD
// L2s are not coherent with each other, so a barrier needs either agent-scope release/acquire (buffer_wbl2 + buffer_inv:
// ~11 us per barrier with nothing dirty) or sc1 (memory-side) accesses for everything the phases exchange — then an
// atomic-free flag barrier costs ~4 us, but the slab re-reads that the per-XCD L2 absorbs for free in separate launches
// all go to the memory side: 88 us per direction against 68 us as six launches (docs/DESIGN_HISTORY.md 7b).
#include "hp_common.h"
#include "hp_skinny.h"
#include <algorithm>
#include <cstdlib>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

constexpr int kThreads = 256;
constexpr int kCLMax = 128;                       // contraction chunk staged per pass
constexpr int kLdsFloats = 96 * (kCLMax + 4);     // staged A (64 rows) + W (32 rows) blocks; the cross-wave reduce aliases them

constexpr int kPhaseOps = 8;   // ops per launch (a phase with more is split: its ops are independent)
struct Prog {                  // the kernel argument: only the launch's own ops (the CPU copies it per launch)
    int nops;
    HpSkOp op[kPhaseOps];
};

struct Buf {               // raw buffer over [base, base + 2 GB): 32-bit offsets
    __amdgpu_buffer_rsrc_t r;
    __device__ __forceinline__ explicit Buf(const void* base)
        : r(__builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, 0x7fffffff, 0x00020000)) {}
    __device__ __forceinline__ float4 ld4(long off_floats) const {
        const i32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)(off_floats * 4), 0, 0);
        return make_float4(__int_as_float(v.x), __int_as_float(v.y), __int_as_float(v.z), __int_as_float(v.w));
    }
    __device__ __forceinline__ float ld1(long off_floats) const {
        return __int_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, (int)(off_floats * 4), 0, 0));
    }
};

__device__ __forceinline__ float4 f4add(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float4 f4zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }
__device__ __forceinline__ float f4get(const float4& v, int s) { return s == 0 ? v.x : (s == 1 ? v.y : (s == 2 ? v.z : v.w)); }

// Finish already-summed slab values: + bias, ReLU  |  mask (a source carries a bias or a mask, never both: `aux`).
__device__ __forceinline__ float4 src_finish(const HpSkSrc& a, float4 v, const float4& aux) {
    if (a.bias) v = f4add(v, aux);
    if (a.relu) v = make_float4(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f));
    if (a.mask) v = make_float4(aux.x > 0.f ? v.x : 0.f, aux.y > 0.f ? v.y : 0.f, aux.z > 0.f ? v.z : 0.f, aux.w > 0.f ? v.w : 0.f);
    return v;
}

// finished value of 4 consecutive columns of row `row` of a source (columns c .. c+3, 16-byte aligned); S <= 4 slabs
__device__ __forceinline__ float4 src_load4(const HpSkSrc& a, int row, int c) {
    const Buf buf(a.p);
    const long off = (long)row * a.ld + c;
    float4 t[4];
#pragma unroll
    for (int s = 0; s < 4; ++s)
        if (s < a.S) t[s] = buf.ld4(off + (long)s * a.slab);
    float4 aux = f4zero();
    if (a.bias) aux = *reinterpret_cast<const float4*>(a.bias + c);
    if (a.mask) aux = *reinterpret_cast<const float4*>(a.mask + (long)row * a.ldm + c);
    float4 v = t[0];
#pragma unroll
    for (int s = 1; s < 4; ++s)
        if (s < a.S) v = f4add(v, t[s]);
    return src_finish(a, v, aux);
}

// One staged chunk of a source in flight: columns [c0, c0+CL) of all 64 rows, CL = 16*NG, lanes along the columns, NG
// groups of 4 columns per thread x SB slab slots.  load() only ISSUES, and it is BRANCH-FREE (rows >= M re-read row M-1,
// slab slots >= S re-read slab 0, the auxiliary operand always comes from a valid address): inside one basic block the
// compiler counts the loads in flight exactly, so staging chunk c waits for chunk c's loads only while chunk c+1's stay
// in flight; a predicated load costs that (measured: every chunk then paid a full memory latency).
// store() finishes (range-ordered slab sum, bias, ReLU, mask) into As[64][CL+4] (rows >= M are zero) and, for the
// designated reader, into the source's `mat`.
template <int NG, int SB>   // SB: slab slots held per group (1: an already finished source, 4: up to four partial slabs)
struct SrcChunk {
    static constexpr int CL = NG * 16, ldl = CL + 4, QR = CL / 4;
    float4 t[NG][SB], aux[NG];
    int c0;
    __device__ __forceinline__ void load(const HpSkSrc& a, int M, int c0_) {
        c0 = c0_;
        const Buf buf(a.p);
        // the auxiliary operand: bias(col) | mask(row, col) | (neither: the source itself, value unused)
        const float* auxp = a.bias ? a.bias : (a.mask ? a.mask : a.p);
        const int aux_ld = a.bias ? 0 : (a.mask ? a.ldm : a.ld);
#pragma unroll
        for (int e = 0; e < NG; ++e) {
            const int idx = threadIdx.x + e * kThreads, row = min(idx / QR, M - 1), q = idx % QR;
            const long off = (long)row * a.ld + c0 + 4 * q;
#pragma unroll
            for (int u = 0; u < SB; ++u) t[e][u] = buf.ld4(off + (long)(u < a.S ? u : 0) * a.slab);
            aux[e] = *reinterpret_cast<const float4*>(auxp + (long)row * aux_ld + c0 + 4 * q);
        }
    }
    __device__ __forceinline__ void store(const HpSkSrc& a, int M, float* As, bool materialise) const {
#pragma unroll
        for (int e = 0; e < NG; ++e) {
            const int idx = threadIdx.x + e * kThreads, row = idx / QR, q = idx % QR;
            float4 v = t[e][0];
#pragma unroll
            for (int u = 1; u < SB; ++u)
                if (u < a.S) v = f4add(v, t[e][u]);
            float4 f = src_finish(a, v, aux[e]);
            if (row >= M) f = f4zero();
            *reinterpret_cast<float4*>(&As[row * ldl + 4 * q]) = f;
            if (materialise && row < M) *reinterpret_cast<float4*>(a.mat + (long)row * a.ldmat + c0 + 4 * q) = f;
        }
    }
};

// Columns [c0, c0+CL) of weight rows [n0, n0+32) (clamped to N-1): NG/2 groups per thread, branch-free.
template <int NG>
struct WChunk {
    static constexpr int CL = NG * 16, ldl = CL + 4, QR = CL / 4, NV = NG / 2;
    static_assert(NV * kThreads == 32 * QR, "whole passes of the workgroup over the block");
    float4 v[NV];
    __device__ __forceinline__ void load(const float* __restrict__ W, int ld, int N, int n0, int c0) {
#pragma unroll
        for (int e = 0; e < NV; ++e) {
            const int idx = threadIdx.x + e * kThreads;
            v[e] = *reinterpret_cast<const float4*>(W + (long)min(n0 + idx / QR, N - 1) * ld + c0 + 4 * (idx % QR));
        }
    }
    __device__ __forceinline__ void store(float* Ws) const {
#pragma unroll
        for (int e = 0; e < NV; ++e) {
            const int idx = threadIdx.x + e * kThreads;
            *reinterpret_cast<float4*>(&Ws[(idx / QR) * ldl + 4 * (idx % QR)]) = v[e];
        }
    }
};

// Sum the four waves' accumulators through LDS (wave order) and store the 64 x 32 block: out(row, col0 + i).
__device__ __forceinline__ void reduce_store(f32x16 (&acc)[2], float* red, const HpSkOp& op, float* out, int ncols, int col0,
                                             bool direct) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, i = lane & 31, h = lane >> 5;
    __syncthreads();   // every wave is done with the staged operands the reduce buffer aliases
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int e = 0; e < 16; ++e) red[((w * 2 + mt) * 16 + e) * 64 + lane] = acc[mt][e];
    __syncthreads();
    const int col = col0 + i;
    const float bv = (direct && op.out_bias && col < ncols) ? op.out_bias[col] : 0.f;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int e = 4 * w + q;
            float v = red[((0 * 2 + mt) * 16 + e) * 64 + lane];
            v += red[((1 * 2 + mt) * 16 + e) * 64 + lane];
            v += red[((2 * 2 + mt) * 16 + e) * 64 + lane];
            v += red[((3 * 2 + mt) * 16 + e) * 64 + lane];
            const int row = mt * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
            if (direct) {
                v += bv;
                if (op.out_relu) v = fmaxf(v, 0.f);
            }
            if (row < op.M && col < ncols) out[(long)row * op.out_ld + col] = v;
        }
    __syncthreads();
}

// F: strip = 32 output columns (weight rows), range = op.CL consecutive k, staged in chunks of CL = 16*NG; the loads of
// chunk c+1 are issued before chunk c is staged, so two chunks are always in flight
template <int NG, int SB>
__device__ __forceinline__ void task_f(const HpSkOp& op, int t, float* lds) {
    constexpr int CL = NG * 16, ldl = CL + 4, ks = CL / 4;   // ks: a wave's k-slice of a chunk
    const int strips = (op.N + 31) >> 5;
    const int strip = t % strips, range = t / strips;
    const int nr = op.K / op.CL;                 // ranges
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, i = lane & 31, h = lane >> 5;
    const int chunks = op.CL / CL;
    const bool mat = strip == 0 && op.a.mat != nullptr;
    float* As = lds;
    float* Ws = lds + 64 * ldl;
    f32x16 acc[2] = {};
    SrcChunk<NG, SB> sc[2];
    WChunk<NG> wc[2];
    const int c_first = range * op.CL;
    auto mfma_chunk = [&]() {
#pragma unroll
        for (int kq = 0; kq < ks / 8; ++kq) {
            const int kk = w * ks + kq * 8;
            const float4 a0 = *reinterpret_cast<const float4*>(&As[i * ldl + kk + 4 * h]);
            const float4 a1 = *reinterpret_cast<const float4*>(&As[(32 + i) * ldl + kk + 4 * h]);
            const float4 bb = *reinterpret_cast<const float4*>(&Ws[i * ldl + kk + 4 * h]);
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(f4get(a0, s), f4get(bb, s), acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(f4get(a1, s), f4get(bb, s), acc[1], 0, 0, 0);
            }
        }
    };
    // The next chunk's loads are issued UNCONDITIONALLY (past the end they re-read the last chunk, unused): with a
    // conditional prefetch the compiler cannot know how many loads are in flight when it stages the current chunk and
    // falls back to s_waitcnt vmcnt(0) — which also waits for the prefetch, i.e. no overlap at all.
    wc[0].load(op.w, op.w_ld, op.N, strip * 32, c_first);
    sc[0].load(op.a, op.M, c_first);
#pragma unroll 1
    for (int ch = 0; ch < chunks; ch += 2) {
        {
            const int c1 = c_first + min(ch + 1, chunks - 1) * CL;
            wc[1].load(op.w, op.w_ld, op.N, strip * 32, c1);
            sc[1].load(op.a, op.M, c1);
        }
        if (ch) __syncthreads();                 // the previous chunk's fragments are read
        sc[0].store(op.a, op.M, As, mat);
        wc[0].store(Ws);
        __syncthreads();
        mfma_chunk();
        if (ch + 1 >= chunks) break;             // a single-chunk range
        {
            const int c2 = c_first + min(ch + 2, chunks - 1) * CL;
            wc[0].load(op.w, op.w_ld, op.N, strip * 32, c2);
            sc[0].load(op.a, op.M, c2);
        }
        __syncthreads();
        sc[1].store(op.a, op.M, As, mat);
        wc[1].store(Ws);
        __syncthreads();
        mfma_chunk();
    }
    reduce_store(acc, lds, op, op.out + (long)range * op.out_slab, op.N, strip * 32, nr == 1);
}

// X: unit = 32 output columns (k of the weights), range = op.CL consecutive n (the contraction)
template <int NG, int SB>
__device__ __forceinline__ void task_x(const HpSkOp& op, int t, float* lds) {
    constexpr int CL = NG * 16, ldl = CL + 4, ns = CL / 4;   // ns: a wave's n-slice of a chunk (8, 16 or 32 rows of W)
    const int units = (op.K + 31) >> 5;
    const int unit = t % units, range = t / units;
    const int nr = op.N / op.CL;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, i = lane & 31, h = lane >> 5;
    const int chunks = op.CL / CL;
    const bool mat = unit == 0 && op.a.mat != nullptr;
    float* As = lds;
    const int col = min(unit * 32 + i, op.K - 1);
    f32x16 acc[2] = {};
    SrcChunk<NG, SB> sc[2];
    float bw[2][ns / 2];
    // B fragments straight from global memory (a wave's rows of W, 128-byte segments)
    auto load_b = [&](int n0, float (&dst)[ns / 2]) {
        const float* wp = op.w + (long)(n0 + w * ns + 4 * h) * op.w_ld + col;
#pragma unroll
        for (int g = 0; g < ns / 8; ++g)
#pragma unroll
            for (int s = 0; s < 4; ++s) dst[4 * g + s] = wp[(long)(8 * g + s) * op.w_ld];
    };
    auto mfma_chunk = [&](const float (&bwc)[ns / 2]) {
#pragma unroll
        for (int g = 0; g < ns / 8; ++g) {
            const int kk = w * ns + 8 * g;
            const float4 a0 = *reinterpret_cast<const float4*>(&As[i * ldl + kk + 4 * h]);
            const float4 a1 = *reinterpret_cast<const float4*>(&As[(32 + i) * ldl + kk + 4 * h]);
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(f4get(a0, s), bwc[4 * g + s], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(f4get(a1, s), bwc[4 * g + s], acc[1], 0, 0, 0);
            }
        }
    };
    const int n_first = range * op.CL;
    load_b(n_first, bw[0]);
    sc[0].load(op.a, op.M, n_first);
#pragma unroll 1
    for (int ch = 0; ch < chunks; ch += 2) {     // unconditional prefetch: see task_f
        {
            const int n1 = n_first + min(ch + 1, chunks - 1) * CL;
            load_b(n1, bw[1]);
            sc[1].load(op.a, op.M, n1);
        }
        if (ch) __syncthreads();
        sc[0].store(op.a, op.M, As, mat);
        __syncthreads();
        mfma_chunk(bw[0]);
        if (ch + 1 >= chunks) break;
        {
            const int n2 = n_first + min(ch + 2, chunks - 1) * CL;
            load_b(n2, bw[0]);
            sc[0].load(op.a, op.M, n2);
        }
        __syncthreads();
        sc[1].store(op.a, op.M, As, mat);
        __syncthreads();
        mfma_chunk(bw[1]);
    }
    reduce_store(acc, lds, op, op.out + (long)range * op.out_slab, op.K, unit * 32, nr == 1);
}

// W: one 32 x 32 tile of out(N x K) per wave; contraction over the M <= 64 clouds
__device__ __forceinline__ void task_w(const HpSkOp& op, int t) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, i = lane & 31, h = lane >> 5;
    const int ktiles = (op.K + 31) >> 5, ntiles = (op.N + 31) >> 5;
    const int tile = t * 4 + w;
    if (tile >= ktiles * ntiles) return;
    const int kt = tile % ktiles, nt = tile / ktiles;
    const int ncol = min(nt * 32 + i, op.N - 1), kcol = min(kt * 32 + i, op.K - 1);
    const float* ap = op.a.p + ncol;
    const float* bp = op.w + kcol;
    float av[32], bv[32];
#pragma unroll
    for (int s = 0; s < 32; ++s) {
        const int m = min(2 * s + h, op.M - 1);      // branch-free: all 64 loads in flight; rows >= M zeroed below
        av[s] = ap[(long)m * op.a.ld];
        bv[s] = bp[(long)m * op.w_ld];
    }
#pragma unroll
    for (int s = 0; s < 32; ++s)
        if (2 * s + h >= op.M) av[s] = 0.f;
    f32x16 acc = {};
    float asum = 0.f;
#pragma unroll
    for (int s = 0; s < 32; ++s) {
        asum += av[s];
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s], bv[s], acc, 0, 0, 0);
    }
    if (op.rsum && kt == 0) {
        const float other = __shfl_xor(asum, 32, 64);
        if (h == 0 && nt * 32 + i < op.N) op.rsum[nt * 32 + i] = asum + other;
    }
    const int col = kt * 32 + i;
    if (col < op.K) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int row = nt * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
            if (row < op.N) op.out[(long)row * op.out_ld + col] = acc[e];
        }
    }
}

// FIN: out(M x N) = finished A, 1024 elements per task
__device__ __forceinline__ void task_fin(const HpSkOp& op, int t) {
    const int q_per_row = op.N >> 2;
    const int idx = t * kThreads + threadIdx.x;
    if (idx >= op.M * q_per_row) return;
    const int r = idx / q_per_row, q = idx - r * q_per_row;
    *reinterpret_cast<float4*>(op.out + (long)r * op.out_ld + 4 * q) = src_load4(op.a, r, 4 * q);
}

// One phase (<= kPhaseOps ops of the program), the ops' tasks dealt round-robin over the grid.  Chunk depth: 128 for a
// finished source (one 16-byte load per group), 64 for a source in up to four slabs — either way two chunks = 32 16-byte
// loads per thread are in flight (deeper spills past the 512 registers).
// LIGHT: a phase of W / FIN tasks only — the same loop compiled without the F / X bodies needs a quarter of the registers,
// so its workgroups find room beside another stream's GEMM workgroups instead of waiting for a whole CU to drain (a FIN
// phase of the first encoder's tail sat 270 us behind the other encoder's conv5 that way).
template <bool LIGHT>
__global__ __launch_bounds__(kThreads) void skinny_kernel(const Prog g_arg) {
    __shared__ __attribute__((aligned(16))) float lds[LIGHT ? 4 : kLdsFloats];
    const int G = gridDim.x;
    // The program is indexed with a run-time op number: read it where it lies, in the kernel-argument segment (scalar
    // loads), instead of letting the compiler copy the by-value struct to scratch memory to index it.
    const Prog& g = *(const Prog*)__builtin_amdgcn_kernarg_segment_ptr();
    (void)g_arg;
    int base = 0;
    for (int o = 0; o < g.nops; ++o) {
        const HpSkOp& op = g.op[o];
        // ops of one phase start on different workgroups
        int first = (int)blockIdx.x - base % G;
        if (first < 0) first += G;
        const int cl = min(op.CL, kCLMax);
        for (int t = first; t < op.ntasks; t += G) {
            if (!LIGHT && op.type == HP_SK_F) {
                if (op.a.S == 1) {
                    if (cl == 128) task_f<8, 1>(op, t, lds);
                    else if (cl == 64) task_f<4, 1>(op, t, lds);
                    else task_f<2, 1>(op, t, lds);
                } else {
                    if (cl >= 64) task_f<4, 4>(op, t, lds);
                    else task_f<2, 4>(op, t, lds);
                }
            } else if (!LIGHT && op.type == HP_SK_X) {
                if (op.a.S == 1) {
                    if (cl == 128) task_x<8, 1>(op, t, lds);
                    else if (cl == 64) task_x<4, 1>(op, t, lds);
                    else task_x<2, 1>(op, t, lds);
                } else {
                    if (cl >= 64) task_x<4, 4>(op, t, lds);
                    else task_x<2, 4>(op, t, lds);
                }
            } else if (op.type == HP_SK_W) {
                task_w(op, t);
            } else if (op.type == HP_SK_FIN) {
                task_fin(op, t);
            }
        }
        base += op.ntasks;
    }
}

inline bool aligned16(const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; }
inline bool pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }

bool src_ok(const HpSkSrc& a, int cols) {
    if (!a.p || a.S < 1 || a.S > 4 || !aligned16(a.p) || a.ld % 4 || cols % 4) return false;
    if (a.S > 1 && a.slab % 4) return false;
    if (a.bias && !aligned16(a.bias)) return false;
    if (a.mask && (!aligned16(a.mask) || a.ldm % 4)) return false;
    if (a.mat && (!aligned16(a.mat) || a.ldmat % 4)) return false;
    if (a.bias && a.mask) return false;   // one auxiliary operand per source
    return true;
}

}  // namespace

namespace {
int g_skinny = -1;   // -1: HP_SKINNY from the environment (default on), else the value hp_skinny_set_enabled gave
}
bool hp_skinny_enabled() {
    static const bool env_on = [] {
        const char* e = std::getenv("HP_SKINNY");
        return !(e && e[0] == '0');
    }();
    return g_skinny < 0 ? env_on : g_skinny != 0;
}
// Test/diagnostic switch: 0 sends the M <= 64 chains back to the tiled GEMM launches, 1 to the persistent layer programs,
// -1 restores the default.  Returns the previous setting.
HP_API int hp_skinny_set_enabled(int on) {
    const int prev = g_skinny;
    g_skinny = on < 0 ? -1 : (on != 0);
    return prev;
}

int hp_skinny_run(HpSkProgram* prog, hipStream_t stream) {
    if (!prog || prog->nops < 1 || prog->nops > HP_SK_MAX_OPS) return -2;
    int maxtasks = 0, phases = 1;
    for (int o = 0; o < prog->nops; ++o) {
        HpSkOp& op = prog->op[o];
        if (o && op.phase < prog->op[o - 1].phase) return -2;
        if (o && op.phase != prog->op[o - 1].phase) ++phases;
        if (op.M < 1 || op.M > 64 || op.N < 1 || op.K < 1 || !op.out || !aligned16(op.out)) return -2;
        switch (op.type) {
            case HP_SK_F:
                if (op.K % 32 || !pow2(op.CL) || op.CL < 32 || op.K % op.CL || !src_ok(op.a, op.K) || !op.w || !aligned16(op.w) ||
                    op.w_ld % 4)
                    return -2;
                op.ntasks = ((op.N + 31) / 32) * (op.K / op.CL);
                break;
            case HP_SK_X:
                if (op.N % 32 || !pow2(op.CL) || op.CL < 32 || op.N % op.CL || !src_ok(op.a, op.N) || !op.w) return -2;
                op.ntasks = ((op.K + 31) / 32) * (op.N / op.CL);
                break;
            case HP_SK_W:
                if (!op.a.p || op.a.S != 1 || op.a.bias || op.a.mask || op.a.relu || !op.w) return -2;
                op.ntasks = (((op.N + 31) / 32) * ((op.K + 31) / 32) + 3) / 4;
                break;
            case HP_SK_FIN:
                if (op.N % 4 || op.out_ld % 4 || !src_ok(op.a, op.N)) return -2;
                op.ntasks = (op.M * (op.N / 4) + kThreads - 1) / kThreads;
                break;
            default: return -2;
        }
        maxtasks = std::max(maxtasks, op.ntasks);
    }
    (void)phases;
    (void)maxtasks;
    // one launch per phase: the kernel boundary orders the phases
    for (int b = 0; b < prog->nops;) {
        Prog g;
        g.nops = 0;
        int e = b, tasks = 0;
        bool light = true;
        for (; e < prog->nops && prog->op[e].phase == prog->op[b].phase && g.nops < kPhaseOps; ++e) {
            g.op[g.nops++] = prog->op[e];
            tasks += prog->op[e].ntasks;
            light = light && (prog->op[e].type == HP_SK_W || prog->op[e].type == HP_SK_FIN);
        }
        const dim3 grid(std::max(1, std::min(tasks, 512)));
        if (light) hipLaunchKernelGGL(skinny_kernel<true>, grid, dim3(kThreads), 0, stream, g);
        else hipLaunchKernelGGL(skinny_kernel<false>, grid, dim3(kThreads), 0, stream, g);
        b = e;
    }
    HP_RETURN_LAST_ERROR();
}
