// The encoders' conv layers 2..5 (/root/reference/model/encoder.py:14-28, 43-45) with BOTH operands already split: the
// "piece format" (P-format) GEMM of round 4.
//
// conv_split.hip (round 3) put the layers on the f16 matrix pipe — every fp32 operand as two f16 pieces, three MFMA products —
// but its activation operand arrived as fp32: loaded to registers, split on the VALU, written to LDS as two images, in every
// column workgroup of a row panel, two barriers per 32-deep k-tile: 2650 non-MFMA cycles against 768 MFMA cycles per k-tile,
// 0.29 of the pipe's (f16 / 3) roofline.  Here the layer that PRODUCES an activation tensor stores it already split:
//
//   P-format of a tensor T (rows, C), C % 32 == 0:  row r = C/32 lines of 128 bytes, line kt = [hi(32 f16) | lo(32 f16)] of
//   channels 32 kt .. 32 kt + 31, with  T[r][c] = (hi + lo) * 2^-e,  e = exp[(r >> 7) * ncb + (c / cb)]  one exponent per block
//   of 128 rows x cb channels (cb = the producing workgroup's column extent).  Same bytes as fp32.  hi = rne16(T 2^e),
//   lo = rne16(T 2^e - hi), e = min(14 - exponent(max over the block), 54): the block's max lands in [2^13, 2^14) (hi cannot
//   overflow), blocks whose max is below 2^-40 keep e = 54 (they only lose bits that lie below 2^-79).  The weights use the
//   same line layout with one exponent per output channel (conv_split_prep_kernel).
//
// so the consumer needs NO VALU pass and NO staging registers: both operand tiles go global -> LDS by LDS-DMA
// (global_load_lds_dwordx4, 1 KB per wave instruction), which allows the 256 x 256 x 32 tile on 8 waves (128 x 64 per wave,
// one workgroup per CU, 128 KB of LDS in two buffers, ONE barrier per k-tile with the next tile's DMA in flight across the
// MFMA phase) that cdna_hip_programming.md gives as the structure beyond the two-barrier 128^2 ceiling.  Per k-tile and wave:
// 8 DMA instructions, 24 ds_read_b128, 48 v_mfma_f32_32x32x16_f16.
//   LDS image of an operand tile: 256 rows x 128 bytes, the 16-byte chunk c of row r at slot c ^ ((r >> 1) & 7) — applied on
//   the DMA's SOURCE address (the DMA writes lane-linear) and on the fragment reads: conflict-free ds_read_b128 on unpadded rows.
//   An exponent that changes along k (the A operand's column blocks: h4's two 256-channel blocks in layer 5) rescales the
//   accumulators by the exact 2^(e_next - e_prev) at the block boundary (128 v_mul per wave, once per layer).
// Epilogues:
//   MODE 1 (layer 5)     rows = points (A = activations): bias, then the fused max-pool of model/encoder.py:45 — a wave owns
//                        a whole 128-row tile, so it writes the (tile, channel) partial maxima itself (first row wins).
//   MODE 0 (layers 2-4)  the MFMA operands are swapped, so the accumulator rows are CHANNELS and the lanes are points: a lane
//                        then holds 4 consecutive channels per register group, a v_permlane32_swap pair makes that 8 = 16
//                        bytes of a P-format line, and the layer's output goes out as 16-byte stores — bias, ReLU, the
//                        block maximum (4 waves meet in LDS), the split, 32 global_store_dwordx4 per wave.
// The backward (enc_bwd.hip) reads the same P-format rows: (hi + lo) * 2^-e is exact in fp32.
#include "hp_common.h"
#include "hp_conv_split.h"
#include <algorithm>
#include <cstdlib>
#include <type_traits>

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

constexpr int kTarget = 14;
constexpr int kExpMax = HP_PP_EXP_MAX;

__device__ __forceinline__ int frexp_exp(unsigned bits) {
    const int E = (int)((bits >> 23) & 0xff);
    return E ? E - 126 : 0;
}
__device__ __forceinline__ float pow2f(int e) {
    e = max(-126, min(127, e));
    return __uint_as_float((unsigned)(e + 127) << 23);
}
__device__ __forceinline__ int block_exp(float m) { return min(kTarget - frexp_exp(__float_as_uint(m)), kExpMax); }
__device__ __forceinline__ unsigned pack2(_Float16 a, _Float16 b) { return __builtin_bit_cast(unsigned, f16x2{a, b}); }
// the lo pieces of two values whose hi pieces are packed in `hpk`: lo = f16(x - f32(hi)) in ONE instruction each (v_fma_mix
// takes the f16 operand as it is; through C++ the compiler emits cvt + sub + cvt: 3 of the ~9 vector instructions per output
// element of a store epilogue that the matrix cores sit idle behind)
__device__ __forceinline__ unsigned lo_pair(float x0, float x1, unsigned hpk) {
    unsigned d;
    asm("v_fma_mixlo_f16 %0, %1, 1.0, -%3 op_sel:[0,0,0] op_sel_hi:[0,0,1]\n\t"
        "v_fma_mixhi_f16 %0, %2, 1.0, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]"
        : "=&v"(d)
        : "v"(x0), "v"(x1), "v"(hpk));
    return d;
}
__device__ __forceinline__ int kmap(int e, int h) { return (e & 3) + 8 * (e >> 2) + 4 * h; }   // C/D row of register e, lane half h

// LDS-DMA through inline asm: with the builtin, hipcc orders every later ds_read behind the pending LDS write with an
// s_waitcnt vmcnt(0), i.e. it drains the prefetch the moment it is issued (measured in round 2, tools/micro/gemm_glds.hip).
// The asm form is invisible to that analysis; the explicit vmcnt(0) + barrier at the top of a k-tile orders the reads.
// ... with the address as a wave-uniform base + a 32-bit lane offset, and the LDS destination as an SGPR base + a LITERAL: the
// sum is formed into m0 inside the asm (a compiler-visible sum gets hoisted out of the k-loop as a loop invariant, runs out of
// SGPRs, is spilled to VGPR lanes and comes back as a v_readlane — a VALU instruction in a load phase, see conv_pp_kernel)
template <int LIT>
__device__ __forceinline__ void glds16s(const void* sbase, unsigned voff, unsigned lds_base) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_add_u32 m0, %3, %4\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(sbase), "s"(lds_base), "n"(LIT)
                 : "memory", "scc");
}
__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_dst)
                 : "memory");
}

struct PpParams {
    const _Float16* A;        // P-format activations (M rows, K channels) of encoder 0
    const int* aexp;          // [(row >> 7) * a_ncb + kb]
    const _Float16* Whl;      // P-format weights (N rows, K) of encoder 0
    const int* wexp;          // (N)
    const float* bias;
    _Float16* C;              // MODE 0: P-format output (M, N)
    int* cexp;                // MODE 0: [(row >> 7) * tiles_n + tile_n]
    float* cmax;              // MODE 1: (ceil(M / 128), N) partial maxima (+ bias)
    int* cidx;                //         row (mod group_rows) attaining them
    long sWs;                 // distance in FLOATS between the two encoders' workspaces (A, aexp, Whl, wexp, C, cexp, cmax, cidx)
    long sBiasz;
    int M, N, K;
    int a_ncb, a_kb_steps;    // exponent blocks of the A operand along k, k-tiles per block
    int relu, group_rows, tiles_n;
};

// TN = 32-column tiles per wave: 2 (BN = 256) or 1 (BN = 128); WM = 128-row wave rows per workgroup: 2 (256 rows, 8 waves, one
// workgroup per CU) or 1 (128 rows, 4 waves, 64 KB of LDS at TN = 1: TWO workgroups per CU, so that one's epilogue — vector ALU
// and stores, the matrix cores idle — runs beside the other's k-loop: the store layers)
template <int MODE, int TN, int WM>
__global__ __launch_bounds__(256 * WM, 2 / WM) void conv_pp_kernel(const PpParams p) {
    constexpr int BN = 128 * TN, BM = 128 * WM, NW = 4 * WM;
    constexpr int kOpBytes = BM * 128;       // the A part of an LDS buffer: BM rows x one 128-byte line
    constexpr int NB = (BN / 8) / NW;          // weight-tile DMA instructions per wave
    static_assert(NB == 2 || NB == 4, "tile shapes: 256x256/8 waves, 256x128/8 waves, 128x128/4 waves");
    constexpr int kBufBytes = kOpBytes + BN * 128;
    // [two operand buffers | per-channel epilogue table of this workgroup's BN columns: 2^-e_w, bias | 8 floats of exchange]
    __shared__ __attribute__((aligned(1024))) unsigned char lds[2 * kBufBytes + BN * 8 + 64];
    float* tab = reinterpret_cast<float*>(lds + 2 * kBufBytes);
    float* xch = tab + 2 * BN;
    // PERSISTENT workgroups: gridDim.x (a multiple of 8 and of tiles_n) of them walk the tiles with stride gridDim.x, so a
    // workgroup keeps its column tile (its epilogue table is loaded once) and the first two k-tiles of its NEXT tile are
    // DMA-prefetched before the epilogue of the current one (no prologue bubble, no workgroup turnover between tiles; the
    // epilogue's stores drain under the next tile's k-loop).
    const int nwg = gridDim.x;
    int wg = blockIdx.x;
    {   // XCD-aware bijective remap: neighbouring tiles (the column tiles of a row panel) run on one XCD, at the same time
        const int q = nwg >> 3, r8 = nwg & 7, xcd = wg & 7;
        wg = (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + (wg >> 3);
    }
    const int z = blockIdx.y;
    const int ntiles = ((p.M + BM - 1) / BM) * p.tiles_n;
    const int tile_n = wg % p.tiles_n, col0 = tile_n * BN;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, wm = wid >> 2, wn = wid & 3, r = lane & 31, h = lane >> 5;
    const int K = p.K, M = p.M;
    const long rowbytes = (long)K * 4;
    const unsigned char* Ab = reinterpret_cast<const unsigned char*>(p.A) + z * p.sWs * 4;
    const unsigned char* Wrow0 = reinterpret_cast<const unsigned char*>(p.Whl) + z * p.sWs * 4 + (long)col0 * rowbytes;
    const float* bias = p.bias + z * p.sBiasz;
    const int* wexp = p.wexp + z * p.sWs;
    if (wg >= ntiles) return;

    // epilogue constants.  MODE 1: a lane's columns are fixed -> registers.  MODE 0: a lane's channels vary with the register
    // index -> an LDS table (ds_read in the epilogue: no vector-memory loads beside the DMAs in flight).
    int wx1[TN];
    float bv1[TN];
    if (MODE == 1) {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            wx1[j] = wexp[col0 + wn * 32 * TN + j * 32 + r];
            bv1[j] = bias[col0 + wn * 32 * TN + j * 32 + r];
        }
    } else {
        if (tid < BN) {
            tab[tid] = pow2f(-wexp[col0 + tid]);
            tab[BN + tid] = bias[col0 + tid];
        }
    }

    // DMA pieces: instruction e of wave `wid` fills rows [(e*8 + wid)*8, +8) of an operand tile; lane -> (row, physical slot).
    // Addresses are a wave-uniform base (SGPR pair, advanced per k-tile) + a 32-bit lane offset: 4 + 1 registers.
    const int sub = lane >> 3, slot = lane & 7;
    unsigned oa[4], ob;
    {
        const int row = wid * 8 + sub;        // instruction e of the weight tile: + 8 NW e rows (uniform)
        ob = (unsigned)row * (unsigned)rowbytes + ((slot ^ ((row >> 1) & 7)) << 4);
    }
    const unsigned lds0 = (unsigned)(uintptr_t)lds;
    const unsigned char* Arow0 = Ab;
    auto set_tile = [&](int tile) {           // DMA source of the activation rows of `tile`
        const int row0 = (tile / p.tiles_n) * BM;
        Arow0 = Ab + (long)row0 * rowbytes;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int row = (e * NW + wid) * 8 + sub;
            oa[e] = (unsigned)(min(row0 + row, M - 1) - row0) * (unsigned)rowbytes + ((slot ^ ((row >> 1) & 7)) << 4);
        }
    };
    // The load phases run beside the partner wave's MFMA stream, which owns the VALU issue port (measured with in-kernel stamps:
    // a v_readfirstlane or an address v_add in a load phase waits ~one MFMA, 18 of them = the whole burst).  So a load phase
    // issues NO vector-ALU instruction: DMA destinations are SALU sums of one SGPR base, fragment addresses are sixteen VGPRs
    // computed once (buffer x half x piece x operand; the per-tile offsets are ds_read immediates), kept opaque to the compiler
    // (it would rematerialise them with VALU adds inside the loop), and the k-loop is unrolled by two so that the buffer index
    // is a compile-time constant.
    const unsigned s_ldsw = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)(wid * 1024));
    unsigned obe[NB];                     // weight pieces: the 64-row step between a wave's instructions rides in the lane offset
#pragma unroll
    for (int e = 0; e < NB; ++e) {
        obe[e] = ob + (unsigned)e * (8u * NW) * (unsigned)rowbytes;
        asm volatile("" : "+v"(obe[e]));
    }
    // the 4 + NB pieces of a k-tile in three parts (a k-tile's DMA is spread over three MFMA groups): part 0 = A pieces 0..2,
    // part 1 = A piece 3 + weight pieces 0, 1, part 2 = weight pieces 2, 3 (when there are four); part < 0: all of them
    auto issue = [&](auto bufc, int kt, int part) {
        constexpr int buf = decltype(bufc)::value;
        constexpr int PS = NW * 1024;                    // LDS distance between a wave's consecutive pieces
        const unsigned char* asrc = Arow0 + kt * 128;
        const unsigned char* bsrc = Wrow0 + kt * 128;
        if (part < 0 || part == 0) {
            glds16s<buf * kBufBytes + 0 * PS>(asrc, oa[0], s_ldsw);
            glds16s<buf * kBufBytes + 1 * PS>(asrc, oa[1], s_ldsw);
            glds16s<buf * kBufBytes + 2 * PS>(asrc, oa[2], s_ldsw);
        }
        if (part < 0 || part == 1) {
            glds16s<buf * kBufBytes + 3 * PS>(asrc, oa[3], s_ldsw);
            glds16s<buf * kBufBytes + kOpBytes + 0 * PS>(bsrc, obe[0], s_ldsw);
            glds16s<buf * kBufBytes + kOpBytes + 1 * PS>(bsrc, obe[1], s_ldsw);
        }
        if (NB > 2 && (part < 0 || part == 2)) {
#pragma unroll
            for (int e = 2; e < NB; ++e) {
                if (e == 2) glds16s<buf * kBufBytes + kOpBytes + 2 * PS>(bsrc, obe[NB > 2 ? 2 : 0], s_ldsw);
                if (e == 3) glds16s<buf * kBufBytes + kOpBytes + 3 * PS>(bsrc, obe[NB > 3 ? 3 : 0], s_ldsw);
            }
        }
    };
    using B0 = std::integral_constant<int, 0>;
    using B1 = std::integral_constant<int, 1>;

    // fragment addresses: row * 128 + ((chunk ^ swz) << 4); every row of this lane has swz = (r >> 1) & 7 (rows differ by
    // multiples of 32)
    const int swz = (r >> 1) & 7;
    const int fa0 = (wm * 128 + r) * 128, fb0 = kOpBytes + (wn * 32 * TN + r) * 128;
    unsigned adA[2][2][2], adB[2][2][2];      // [buffer][half t][piece: hi, lo]
#pragma unroll
    for (int b2 = 0; b2 < 2; ++b2)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int ch = ((2 * t + h) ^ swz) << 4, cl = ch ^ 64;     // hi chunk 2t+h, lo chunk 4+2t+h
            adA[b2][t][0] = (unsigned)(b2 * kBufBytes + fa0 + ch);
            adA[b2][t][1] = (unsigned)(b2 * kBufBytes + fa0 + cl);
            adB[b2][t][0] = (unsigned)(b2 * kBufBytes + fb0 + ch);
            adB[b2][t][1] = (unsigned)(b2 * kBufBytes + fb0 + cl);
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                asm volatile("" : "+v"(adA[b2][t][q]));
                asm volatile("" : "+v"(adB[b2][t][q]));
            }
        }
    struct FragA {        // two 32-row tiles (i = 2 ip, 2 ip + 1) of one 16-deep half
        f16x8 h[2], l[2];
    };
    struct FragB {        // the wave's column tiles of one 16-deep half
        f16x8 h[TN], l[TN];
    };
    auto ldA = [&](FragA& f, int buf, int t, int ip) {      // (buf, t, ip are compile-time constants at every call)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            f.h[u] = *reinterpret_cast<const f16x8*>(lds + adA[buf][t][0] + (2 * ip + u) * 32 * 128);
            f.l[u] = *reinterpret_cast<const f16x8*>(lds + adA[buf][t][1] + (2 * ip + u) * 32 * 128);
        }
    };
    auto ldB = [&](FragB& f, int buf, int t) {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            f.h[j] = *reinterpret_cast<const f16x8*>(lds + adB[buf][t][0] + j * 32 * 128);
            f.l[j] = *reinterpret_cast<const f16x8*>(lds + adB[buf][t][1] + j * 32 * 128);
        }
    };
    f32x16 acc[4][TN];
    // 6 TN MFMAs, product-major: consecutive MFMAs go to different accumulators; per accumulator the order is hi.hi, hi.lo, lo.hi
    auto mma = [&](const FragA& a, const FragB& b, int ip) {
#pragma unroll
        for (int pr = 0; pr < 3; ++pr)
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const f16x8 xa = pr == 2 ? a.l[u] : a.h[u], xb = pr == 1 ? b.l[j] : b.h[j];
                    f32x16& c = acc[2 * ip + u][j];
                    if (MODE == 1) c = __builtin_amdgcn_mfma_f32_32x32x16_f16(xa, xb, c, 0, 0, 0);   // rows = points
                    else c = __builtin_amdgcn_mfma_f32_32x32x16_f16(xb, xa, c, 0, 0, 0);             // rows = channels
                }
    };

    const int KT = K >> 5;
    const int* aexp_z = p.aexp + z * p.sWs;
    set_tile(wg);
    issue(B0{}, 0, -1);
    for (int tile = wg; tile < ntiles; tile += nwg) {
        const int tile_m = tile / p.tiles_n, row0 = tile_m * BM;
        const int tile128 = tile_m * WM + wm;            // this wave's row tile
        // exponents of this wave's 128 A rows, per block along k: wave-uniform -> scalar loads (no vector-memory traffic of the
        // compiler's beside the DMAs)
        const int* aexp = aexp_z + (long)__builtin_amdgcn_readfirstlane(min(tile128, (M - 1) >> 7) * p.a_ncb);
        int ex = aexp[0];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

        // STREAM schedule (all 8 waves alike, ONE barrier per k-tile).  A k-tile is four groups of 6 TN MFMAs (half t x row pair
        // ip); while a group's MFMAs issue, the fragments of the NEXT group are on their way from LDS (two A sets, two B sets
        // in registers) and two or three DMA instructions of a later k-tile are issued, so neither an LDS round trip nor a DMA
        // issue sits in front of an MFMA:
        //   G0  A1 <- (t0, ip1)                       | DMA(kt+1) part 1 | MFMA(A0, B0, ip0)
        //   G1  A0 <- (t1, ip0), B1 <- (t1)           | DMA(kt+1) part 2 | MFMA(A1, B0, ip1)
        //   G2  A1 <- (t1, ip1)                                          | MFMA(A0, B1, ip0)
        //       wait: A1 here, my pieces of tile kt+1 landed | barrier (nobody reads this buffer any more; tile kt+1 complete)
        //   G3  A0, B0 <- tile kt+1 (t0, ip0) from the OTHER buffer | DMA(kt+2) part 0 -> this buffer | MFMA(A1, B1, ip1)
        // (Round 4 tried a two-group ping-pong — one wave of a SIMD in an MFMA phase while its partner loads — first: in-kernel
        //  stamps showed the loading wave making no progress beside a back-to-back MFMA stream, even with s_setprio and with
        //  load phases free of vector-ALU instructions: 93 us per conv5 either way.  docs/DESIGN_HISTORY.md 7b.)
        // k-tile 0 of this tile was issued before the previous tile's epilogue (whose stores are younger: vmcnt(0) waits for them
        // too — they have had the whole epilogue to drain).
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                       // tile 0 has landed, everybody's pieces
        if (KT > 1) issue(B1{}, 1, -1);
        FragA A0, A1;
        FragB Bt0, Bt1;
        ldA(A0, 0, 0, 0);
        ldB(Bt0, 0, 0);
        auto ktile = [&](int kt, auto bufc) {
            constexpr int buf = decltype(bufc)::value;
            using OB = std::integral_constant<int, buf ^ 1>;
            if (p.a_ncb > 1 && kt > 0 && kt % p.a_kb_steps == 0) {   // (wave-uniform) the A operand's exponent changes here
                const int en = aexp[kt / p.a_kb_steps];
                const float sc = pow2f(en - ex);
                ex = en;
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
#pragma unroll
                        for (int e = 0; e < 16; ++e) acc[i][j][e] *= sc;
            }
            // ---- G0
            ldA(A1, buf, 0, 1);
            if (kt > 0 && kt + 1 < KT) issue(OB{}, kt + 1, 1);
            __builtin_amdgcn_sched_barrier(0);
            mma(A0, Bt0, 0);
            __builtin_amdgcn_sched_barrier(0);
            // ---- G1
            ldA(A0, buf, 1, 0);
            ldB(Bt1, buf, 1);
            if (kt > 0 && kt + 1 < KT) issue(OB{}, kt + 1, 2);
            __builtin_amdgcn_sched_barrier(0);
            mma(A1, Bt0, 1);
            __builtin_amdgcn_sched_barrier(0);
            // ---- G2
            ldA(A1, buf, 1, 1);
            __builtin_amdgcn_sched_barrier(0);
            mma(A0, Bt1, 0);
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            // ---- G3
            if (kt + 2 < KT) issue(bufc, kt + 2, 0);
            if (kt + 1 < KT) {
                ldA(A0, buf ^ 1, 0, 0);
                ldB(Bt0, buf ^ 1, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            mma(A1, Bt1, 1);
            __builtin_amdgcn_sched_barrier(0);
        };
        for (int kt = 0; kt < KT; kt += 2) {      // (KT is even: launch_pp)
            ktile(kt, B0{});
            ktile(kt + 1, B1{});
        }
        // (the last k-tile's fragments were in registers before its barrier: both buffers are free)
        if (tile + nwg < ntiles) {   // the next tile's first k-tile flies under this tile's epilogue (both buffers are free)
            set_tile(tile + nwg);
            issue(B0{}, 0, -1);
        }
        __builtin_amdgcn_sched_barrier(0);

        if (MODE == 1) {
            // fused max-pool over the wave's 128 rows (model/encoder.py:45); the first row attaining the max wins
            if (row0 + wm * 128 < M) {
                float* cmax = p.cmax + z * p.sWs;
                int* cidx = p.cidx + z * p.sWs;
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int col = col0 + wn * 32 * TN + j * 32 + r;
                    const float us = pow2f(-ex - wx1[j]), bv = bv1[j];
                    float best = -__builtin_inff();
                    int bi = 0x7fffffff;
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int e = 0; e < 16; ++e) {   // rows ascend with (i, e) for a fixed lane half
                            const int row = row0 + wm * 128 + i * 32 + kmap(e, h);
                            const float v = __builtin_fmaf(acc[i][j][e], us, bv);   // (us is a power of two: = acc*us + bv, one rounding)
                            if (row < M && v > best) {
                                best = v;
                                bi = row;
                            }
                        }
                    const float ov = __shfl_xor(best, 32, 64);
                    const int oi = __shfl_xor(bi, 32, 64);
                    if (ov > best || (ov == best && oi < bi)) {
                        best = ov;
                        bi = oi;
                    }
                    if (h == 0) {
                        cmax[(long)tile128 * p.N + col] = best;
                        cidx[(long)tile128 * p.N + col] = bi % p.group_rows;
                    }
                }
            }
            continue;
        }

        // ---- MODE 0: bias, ReLU, block maximum, split, 16-byte stores.  acc[i][j][e]: channel col0 + wn*32*TN + 32 j + kmap(e, h),
        //      point row0 + wm*128 + 32 i + r.  (Rows past M repeat row M-1 — the DMA clamps —, so they cannot raise the maximum.)
        float m = 0.f;
        const float usA = pow2f(-ex), floor_v = p.relu ? 0.f : -__builtin_inff();
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; e += 4) {
                const int cl = wn * 32 * TN + j * 32 + kmap(e, h);      // 4 consecutive channels: e .. e+3
                const f32x4 us4 = *reinterpret_cast<const f32x4*>(tab + cl), bv4 = *reinterpret_cast<const f32x4*>(tab + BN + cl);
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const float us = us4[u] * usA;      // product of two powers of two: exact
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        // (= acc*us + bias: acc*us is exact; ReLU as a max with 0 or -inf: no select per element)
                        const float v = fmaxf(__builtin_fmaf(acc[i][j][e + u], us, bv4[u]), floor_v);
                        acc[i][j][e + u] = v;
                        m = fmaxf(m, fabsf(v));
                    }
                }
            }
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
        if (lane == 0) xch[wid] = m;
        __syncthreads();
        m = fmaxf(fmaxf(xch[wm * 4], xch[wm * 4 + 1]), fmaxf(xch[wm * 4 + 2], xch[wm * 4 + 3]));
        const int eo = block_exp(m);
        if (wn == 0 && lane == 0 && row0 + wm * 128 < M) p.cexp[z * p.sWs + (long)tile128 * p.tiles_n + tile_n] = eo;
        const float sx = pow2f(eo);
        unsigned char* Cb = reinterpret_cast<unsigned char*>(p.C) + z * p.sWs * 4;
        const long crow = (long)p.N * 4;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = row0 + wm * 128 + i * 32 + r;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                unsigned H[4][2], L[4][2];
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int pr = 0; pr < 2; ++pr) {
                        const float x0 = acc[i][j][4 * q + 2 * pr] * sx, x1 = acc[i][j][4 * q + 2 * pr + 1] * sx;
                        H[q][pr] = pack2((_Float16)x0, (_Float16)x1);
                        L[q][pr] = lo_pair(x0, x1, H[q][pr]);
                    }
                // lanes < 32 hold channels 8q + 0..3, lanes >= 32 channels 8q + 4..7 of group q: a half exchange of the pair
                // (q, q+1) leaves chunk q (8 channels, 16 bytes) in the lower lanes and chunk q+1 in the upper ones
                unsigned char* line = Cb + (long)row * crow + (long)((col0 + wn * 32 * TN) / 32 + j) * 128 + (h << 4);
#pragma unroll
                for (int q = 0; q < 4; q += 2) {
#pragma unroll
                    for (int pr = 0; pr < 2; ++pr) {
                        auto s1 = __builtin_amdgcn_permlane32_swap(H[q][pr], H[q + 1][pr], false, false);
                        H[q][pr] = s1[0];
                        H[q + 1][pr] = s1[1];
                        auto s2 = __builtin_amdgcn_permlane32_swap(L[q][pr], L[q + 1][pr], false, false);
                        L[q][pr] = s2[0];
                        L[q + 1][pr] = s2[1];
                    }
                    if (row < M) {
                        *reinterpret_cast<u32x4*>(line + q * 16) = u32x4{H[q][0], H[q][1], H[q + 1][0], H[q + 1][1]};
                        *reinterpret_cast<u32x4*>(line + 64 + q * 16) = u32x4{L[q][0], L[q][1], L[q + 1][0], L[q + 1][1]};
                    }
                }
            }
        }
        __syncthreads();      // xch is rewritten by the next tile's epilogue
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// layer 1 (K = 3) writing h1 in P-format: the fma chain of conv_split.hip's conv1_kernel (bit-identical values), one
// workgroup pass per 128-row tile: all 8 x 4 outputs of a thread in registers, block maximum, split, 8-byte stores.
__global__ __launch_bounds__(256) void conv1_pp_kernel(const float* __restrict__ x, long sXz, const float* __restrict__ W, long sWz,
                                                       const float* __restrict__ b, long sBz, _Float16* __restrict__ h, int* __restrict__ hexp,
                                                       long sWs, long R) {
    __shared__ float smax[4];
    const int z = blockIdx.y;
    x += z * sXz; W += z * sWz; b += z * sBz;
    unsigned char* hb = reinterpret_cast<unsigned char*>(h) + z * sWs * 4;
    hexp += z * sWs;
    const int c4 = (threadIdx.x & 15) * 4;
    float w0[4], w1[4], w2[4], bb[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        w0[u] = W[(c4 + u) * 3];
        w1[u] = W[(c4 + u) * 3 + 1];
        w2[u] = W[(c4 + u) * 3 + 2];
        bb[u] = b[c4 + u];
    }
    for (long tile = blockIdx.x; tile * 128 < R; tile += gridDim.x) {
        const long base = tile * 128 + (threadIdx.x >> 4);
        float xv[8][3];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const long row = min(base + 16 * q, R - 1);
            xv[q][0] = x[row * 3];
            xv[q][1] = x[row * 3 + 1];
            xv[q][2] = x[row * 3 + 2];
        }
        float o[8][4];
        float m = 0.f;
#pragma unroll
        for (int q = 0; q < 8; ++q)
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float v = __builtin_fmaf(xv[q][2], w2[u], __builtin_fmaf(xv[q][1], w1[u], __builtin_fmaf(xv[q][0], w0[u], 0.f)));
                o[q][u] = fmaxf(v + bb[u], 0.f);
                if (base + 16 * q < R) m = fmaxf(m, o[q][u]);
            }
        m = hp::wave_max(m);
        __syncthreads();
        if ((threadIdx.x & 63) == 0) smax[threadIdx.x >> 6] = m;
        __syncthreads();
        const int eo = block_exp(fmaxf(fmaxf(smax[0], smax[1]), fmaxf(smax[2], smax[3])));
        if (threadIdx.x == 0) hexp[tile] = eo;
        const float sx = pow2f(eo);
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const long row = base + 16 * q;
            f16x4 hi, lo;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float xs = o[q][u] * sx;
                const _Float16 hh = (_Float16)xs;
                hi[u] = hh;
                lo[u] = (_Float16)(xs - (float)hh);
            }
            if (row < R) {
                unsigned char* line = hb + row * 256 + (c4 >> 5) * 128 + (c4 & 31) * 2;
                *reinterpret_cast<f16x4*>(line) = hi;
                *reinterpret_cast<f16x4*>(line + 64) = lo;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// fp32 (M, C) -> P-format with one exponent per (128 rows x cb channels) block: the stand-alone primitive's prepare step and
// the tests' way into the format.  One workgroup per block.
__global__ __launch_bounds__(256) void pp_pack_kernel(const float* __restrict__ X, long M, int C, int cb, _Float16* __restrict__ P,
                                                      int* __restrict__ pexp) {
    __shared__ float smax[4];
    const int ncb = C / cb, blk = blockIdx.x % ncb;
    const long tile = blockIdx.x / ncb, r0 = tile * 128;
    const int rows = (int)min((long)128, M - r0);
    const int per_row = cb / 4;
    float m = 0.f;
    for (int i = threadIdx.x; i < rows * per_row; i += 256) {
        const int rr = i / per_row, c = blk * cb + (i % per_row) * 4;
        const f32x4 v = *reinterpret_cast<const f32x4*>(X + (r0 + rr) * C + c);
        m = fmaxf(fmaxf(m, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
    }
    m = hp::wave_max(m);
    if ((threadIdx.x & 63) == 0) smax[threadIdx.x >> 6] = m;
    __syncthreads();
    const int eo = block_exp(fmaxf(fmaxf(smax[0], smax[1]), fmaxf(smax[2], smax[3])));
    if (threadIdx.x == 0) pexp[tile * ncb + blk] = eo;
    const float sx = pow2f(eo);
    unsigned char* Pb = reinterpret_cast<unsigned char*>(P);
    for (int i = threadIdx.x; i < rows * per_row; i += 256) {
        const int rr = i / per_row, c = blk * cb + (i % per_row) * 4;
        const f32x4 v = *reinterpret_cast<const f32x4*>(X + (r0 + rr) * C + c);
        f16x4 hi, lo;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float xs = v[u] * sx;
            const _Float16 hh = (_Float16)xs;
            hi[u] = hh;
            lo[u] = (_Float16)(xs - (float)hh);
        }
        unsigned char* line = Pb + (r0 + rr) * (long)C * 4 + (c >> 5) * 128 + (c & 31) * 2;
        *reinterpret_cast<f16x4*>(line) = hi;
        *reinterpret_cast<f16x4*>(line + 64) = lo;
    }
}

// P-format -> fp32, line by line (one lane per 128-byte line: it reads the whole line before it writes, so `out` may be the
// same memory).  fmt (may be NULL): a device word that says whether `P` holds P-format (HP_PP_FMT_P) at all — anything else:
// nothing to do.
__global__ __launch_bounds__(256) void pp_unpack_kernel(const _Float16* P, const int* __restrict__ pexp, long M, int C, int cb,
                                                        float* out, const int* __restrict__ fmt) {
    if (fmt && *fmt != HP_PP_FMT_P) return;
    const long nlines = M * (C / 32);
    const int ncb = C / cb;
    for (long ln = (long)blockIdx.x * 256 + threadIdx.x; ln < nlines; ln += (long)gridDim.x * 256) {
        const long row = ln / (C / 32);
        const int kt = (int)(ln % (C / 32));
        const u32x4* src = reinterpret_cast<const u32x4*>(reinterpret_cast<const unsigned char*>(P) + ln * 128);
        u32x4 w[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) w[q] = src[q];
        const float us = pow2f(-pexp[(row >> 7) * ncb + (kt * 32) / cb]);
        f32x4* dst = reinterpret_cast<f32x4*>(reinterpret_cast<unsigned char*>(out) + ln * 128);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f16x8 hi = __builtin_bit_cast(f16x8, w[q]), lo = __builtin_bit_cast(f16x8, w[q + 4]);
            f32x4 a, b;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                a[u] = ((float)hi[u] + (float)lo[u]) * us;
                b[u] = ((float)hi[u + 4] + (float)lo[u + 4]) * us;
            }
            dst[2 * q] = a;
            dst[2 * q + 1] = b;
        }
    }
}

__global__ void pp_set_fmt_kernel(int* fmt, int v) { *fmt = v; }

bool g_presplit = [] {
    const char* e = getenv("HP_CONV_PRESPLIT");
    return !(e && e[0] == '0');
}();

// column tile of a launch: 256 where the layer has that many columns — except the STORE layers when HP_PP_BN128=1 (experiment: every
// store layer in 128-column tiles = the two-workgroups-per-CU shape)
int pp_bn(int N, int mode) {
    static const int kBn128 = [] {
        const char* e = getenv("HP_PP_BN128");
        return e ? atoi(e) : 0;
    }();
    return (N >= 256 && !(mode == 0 && kBn128)) ? 256 : 128;
}
int launch_pp(int mode, int n, const PpParams& p, hipStream_t stream) {
    const int bn = pp_bn(p.N, mode);
    if (p.N % bn || p.K % 64 || p.M <= 0) return -1;       // (K % 64: an even number of k-tiles — the two LDS buffers alternate)
    PpParams q = p;
    q.tiles_n = p.N / bn;
    // persistent workgroups: one per CU (256) — two of the 4-wave ones —, fewer when there are fewer tiles; a multiple of 8
    // (XCD remap) and of tiles_n
    static const int kSmall = [] {
        const char* e = getenv("HP_PP_SMALL");      // 1 (default): store layers with 128-column tiles run as 128 x 128 / 4-wave workgroups
        return e ? atoi(e) : 1;
    }();
    const bool small = mode == 0 && bn == 128 && kSmall;
    const int bm = small ? 128 : 256;
    const long tiles = (long)((p.M + bm - 1) / bm) * q.tiles_n;
    static const int kCus = [] {
        const char* e = getenv("HP_PP_WGS");
        return e ? atoi(e) : 256;
    }();
    const int per_cu = small ? 2 : 1;
    long wgs = std::min<long>(tiles, kCus * per_cu / n > 0 ? kCus * per_cu / n : 1);
    const int mult = 8 * q.tiles_n / (q.tiles_n % 8 == 0 ? 8 : (8 % q.tiles_n == 0 ? q.tiles_n : 1));   // lcm(8, tiles_n) for tiles_n in {1,2,4,8}
    if (wgs >= mult) wgs = wgs / mult * mult;
    else wgs = (tiles >= q.tiles_n) ? q.tiles_n : wgs;
    const dim3 grid((unsigned)wgs, n);
    if (mode == 1) {
        if (bn == 256) hipLaunchKernelGGL((conv_pp_kernel<1, 2, 2>), grid, dim3(512), 0, stream, q);
        else hipLaunchKernelGGL((conv_pp_kernel<1, 1, 2>), grid, dim3(512), 0, stream, q);
    } else {
        if (bn == 256) hipLaunchKernelGGL((conv_pp_kernel<0, 2, 2>), grid, dim3(512), 0, stream, q);
        else if (small) hipLaunchKernelGGL((conv_pp_kernel<0, 1, 1>), grid, dim3(256), 0, stream, q);
        else hipLaunchKernelGGL((conv_pp_kernel<0, 1, 2>), grid, dim3(512), 0, stream, q);
    }
    HP_RETURN_LAST_ERROR();
}

}  // namespace

bool hp_conv_presplit_enabled() { return g_presplit; }
int hp_conv_pp_ncb(int l) {
    static const int kC[5] = {0, 64, 128, 256, 512};
    return l <= 1 ? 1 : kC[l] / pp_bn(kC[l], 0);
}
HP_API int hp_conv_presplit_set(int on) {
    const int was = g_presplit;
    g_presplit = on != 0;
    return was;
}

// exponent tables of the P-format activations h1..h4 inside the split area: 1, 1, 1, 2 column blocks per 128-row tile
static inline long pexp_off(int l, long tp) { return hp_conv_pp_exp_offset(l, tp); }

long hp_conv_pp_fmt_offset(long R) { return hp_conv_pp_exp_offset(5, hp_conv_split_tiles_pad(R)); }

int hp_conv_pp_layer1(int n, const float* x, long sXz, const float* W, long sWz, const float* b, long sBz, float* h1, float* area0,
                      long sWs, long R, hipStream_t stream) {
    const long tp = hp_conv_split_tiles_pad(R), blocks = (R + 127) / 128;
    hipLaunchKernelGGL(conv1_pp_kernel, dim3((unsigned)(blocks < 2048 ? blocks : 2048), n), dim3(256), 0, stream, x, sXz, W, sWz, b, sBz,
                       reinterpret_cast<_Float16*>(h1), reinterpret_cast<int*>(area0 + pexp_off(1, tp)), sWs, R);
    HP_RETURN_LAST_ERROR();
}

// layer l = 2..5 on P-format operands: X = h_{l-1}, C = h_l (both inside the workspace of encoder 0; encoder 1's sWs floats on).
// l = 5: colmax (no store): cmax / cidx = per-128-row-tile partial maxima.
int hp_conv_pp_layer(int l, int n, const float* X, const float* bias, long sBiasz, float* C, float* area0, long sWs, long M,
                     float* cmax, int* cidx, int group_rows, hipStream_t stream) {
    static const int kRowsL[5] = {0, 128, 384, 896, 1408};
    static const long kWOffL[4] = {0, 8192, 40960, 172032};
    static const int kKL[4] = {64, 128, 256, 512};
    if (l < 2 || l > 5 || M <= 0) return -1;
    const long tp = hp_conv_split_tiles_pad(M);
    PpParams p{};
    p.A = reinterpret_cast<const _Float16*>(X);
    p.aexp = reinterpret_cast<const int*>(area0 + pexp_off(l - 1, tp));
    p.K = kKL[l - 2];
    p.a_ncb = hp_conv_pp_ncb(l - 1);
    p.a_kb_steps = p.K / 32 / p.a_ncb;
    p.Whl = reinterpret_cast<const _Float16*>(area0 + HP_CS_HI_OFF) + 2 * kWOffL[l - 2];
    p.wexp = reinterpret_cast<const int*>(area0 + HP_CS_WEXP_OFF) + kRowsL[l - 2];
    p.bias = bias; p.sBiasz = sBiasz;
    p.C = reinterpret_cast<_Float16*>(C);
    p.cexp = l < 5 ? reinterpret_cast<int*>(area0 + pexp_off(l, tp)) : nullptr;
    p.cmax = cmax; p.cidx = cidx;
    p.sWs = sWs;
    p.M = (int)M; p.N = kRowsL[l - 1] - kRowsL[l - 2];
    p.relu = l < 5; p.group_rows = group_rows;
    return launch_pp(l == 5 ? 1 : 0, n, p, stream);
}

// marks the workspace's activations as P-format / fp32 (a word in the split area the backward's readers look at)
int hp_conv_pp_mark(int n, float* area0, long sWs, long R, int fmt, hipStream_t stream) {
    for (int z = 0; z < n; ++z)
        hipLaunchKernelGGL(pp_set_fmt_kernel, dim3(1), dim3(1), 0, stream, reinterpret_cast<int*>(area0 + z * sWs + hp_conv_pp_fmt_offset(R)), fmt);
    HP_RETURN_LAST_ERROR();
}

// h1..h4 of a forward workspace back to fp32, in place, if (and only if) the workspace says they are P-format; then the
// mark is cleared.  The layered backward and the tests' view of the workspace; the fused backward reads P-format itself.
int hp_conv_pp_unpack_ws(float* ws, long R, hipStream_t stream) {
    static const int kC[4] = {64, 128, 256, 512};
    const long tp = hp_conv_split_tiles_pad(R);
    float* area = ws + R * (64 + 128 + 256 + 512 + 512);
    int* fmt = reinterpret_cast<int*>(area + hp_conv_pp_fmt_offset(R));
    float* hl = ws;
    for (int l = 1; l <= 4; ++l) {
        const int C = kC[l - 1];
        const long lines = R * (C / 32);
        const unsigned blocks = (unsigned)std::min<long>((lines + 255) / 256, 8192);
        hipLaunchKernelGGL(pp_unpack_kernel, dim3(blocks), dim3(256), 0, stream, reinterpret_cast<const _Float16*>(hl),
                           reinterpret_cast<const int*>(area + pexp_off(l, tp)), R, C, C / hp_conv_pp_ncb(l), hl, fmt);
        hl += R * C;
    }
    hipLaunchKernelGGL(pp_set_fmt_kernel, dim3(1), dim3(1), 0, stream, fmt, HP_PP_FMT_F32);
    HP_RETURN_LAST_ERROR();
}

// ---- the P-format GEMM as a stand-alone primitive (bench.py's roofline leg, tests): C = act(X W^T + b), X (M, K), W (N, K)
// fp32; N % 128 == 0, K % 32 == 0, K <= 512.  ws: hp_gemm_pp_workspace_floats floats =
//   [X in P-format: M*K | exponents of X: tiles*ncb | e_w: N | W in P-format: N*K | C in P-format: M*N | exponents of C | cmax, cidx]
// prepare: packs X (one exponent per 128 rows x xcb channels; xcb = K or a divisor of it that is a multiple of 32) and W.
// run: the conv_pp_kernel launch alone — mode 0: P-format C into ws (hp_gemm_pp_unpack brings it to fp32), mode 1: per-128-row
// column maxima (+ bias) and their rows, the fused max-pool's first stage.
HP_API long hp_gemm_pp_workspace_floats(long M, int N, int K) {
    const long tiles = (M + 127) / 128;
    return M * K + tiles * 16 + N + (long)N * K + M * N + tiles * 16 + 2 * tiles * N + 64;
}
namespace {
struct PpWs {
    float *xp, *xexp, *wexp, *wp, *cp, *cexp, *cmax, *cidx;
};
PpWs pp_ws(float* ws, long M, int N, int K) {
    const long tiles = (M + 127) / 128;
    PpWs w;
    w.xp = ws;
    w.xexp = w.xp + M * K;
    w.wexp = w.xexp + tiles * 16;
    w.wp = w.wexp + N;
    w.cp = w.wp + (long)N * K;
    w.cexp = w.cp + M * N;
    w.cmax = w.cexp + tiles * 16;
    w.cidx = w.cmax + tiles * N;
    return w;
}
}  // namespace
int hp_split_rows_launch(const float* W, int N, int K, float* hl, float* wexp, hipStream_t stream);   // conv_split.hip

HP_API int hp_gemm_pp_prepare(long M, int N, int K, int xcb, const float* X, const float* W, float* ws, hipStream_t stream) {
    HP_CHECK_ARG(M > 0 && N > 0 && K > 0 && N % 128 == 0 && K % 32 == 0 && K <= 512 && xcb > 0 && K % xcb == 0 && xcb % 32 == 0 &&
                 K / xcb <= 16 && X && W && ws);
    const PpWs w = pp_ws(ws, M, N, K);
    const long tiles = (M + 127) / 128;
    hipLaunchKernelGGL(pp_pack_kernel, dim3((unsigned)(tiles * (K / xcb))), dim3(256), 0, stream, X, M, K, xcb,
                       reinterpret_cast<_Float16*>(w.xp), reinterpret_cast<int*>(w.xexp));
    return hp_split_rows_launch(W, N, K, w.wp, w.wexp, stream);
}

HP_API int hp_gemm_pp_run(long M, int N, int K, int xcb, const float* bias, int relu, int mode, int group_rows, float* ws,
                          hipStream_t stream) {
    HP_CHECK_ARG(M > 0 && M < (1L << 31) && N > 0 && K > 0 && N % 128 == 0 && K % 32 == 0 && K <= 512 && xcb > 0 && K % xcb == 0 &&
                 xcb % 32 == 0 && bias && ws && (mode == 0 || (mode == 1 && group_rows > 0)));
    const PpWs w = pp_ws(ws, M, N, K);
    PpParams p{};
    p.A = reinterpret_cast<const _Float16*>(w.xp);
    p.aexp = reinterpret_cast<const int*>(w.xexp);
    p.a_ncb = K / xcb;
    p.a_kb_steps = xcb / 32;
    p.Whl = reinterpret_cast<const _Float16*>(w.wp);
    p.wexp = reinterpret_cast<const int*>(w.wexp);
    p.bias = bias;
    p.C = reinterpret_cast<_Float16*>(w.cp);
    p.cexp = reinterpret_cast<int*>(w.cexp);
    p.cmax = w.cmax;
    p.cidx = reinterpret_cast<int*>(w.cidx);
    p.M = (int)M; p.N = N; p.K = K; p.relu = relu; p.group_rows = group_rows;
    return launch_pp(mode, 1, p, stream);
}

// mode 0's result as fp32 (M, N); mode 1's partials: cmax (ceil(M/128), N) floats and cidx ints copied out
HP_API int hp_gemm_pp_unpack(long M, int N, int K, const float* ws, float* C, hipStream_t stream) {
    HP_CHECK_ARG(M > 0 && N > 0 && ws && C);
    const PpWs w = pp_ws(const_cast<float*>(ws), M, N, K);
    const long lines = M * (N / 32);
    hipLaunchKernelGGL(pp_unpack_kernel, dim3((unsigned)std::min<long>((lines + 255) / 256, 8192)), dim3(256), 0, stream,
                       reinterpret_cast<const _Float16*>(w.cp), reinterpret_cast<const int*>(w.cexp), M, N, pp_bn(N, 0), C,
                       (const int*)nullptr);
    HP_RETURN_LAST_ERROR();
}
HP_API int hp_gemm_pp_partials(long M, int N, int K, const float* ws, float* cmax, int* cidx, hipStream_t stream) {
    HP_CHECK_ARG(M > 0 && N > 0 && ws && cmax && cidx);
    const PpWs w = pp_ws(const_cast<float*>(ws), M, N, K);
    const long n = (M + 127) / 128 * N;
    if (hipMemcpyAsync(cmax, w.cmax, n * 4, hipMemcpyDeviceToDevice, stream) != hipSuccess) return (int)hipGetLastError();
    if (hipMemcpyAsync(cidx, w.cidx, n * 4, hipMemcpyDeviceToDevice, stream) != hipSuccess) return (int)hipGetLastError();
    return 0;
}
