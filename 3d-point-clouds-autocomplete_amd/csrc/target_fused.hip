// Fused per-cloud target network for gfx950 — the published HyperPocket decoder 3 -> 32 -> 64 -> 128 -> 64 -> 3
// (/root/reference/model/target_network.py:6-45 with settings/hyperparams.json "target_network").
//
// The reference builds one TargetNetwork object per cloud and runs 5 torch.mm + 4 ReLU launches on (2048, C)
// tensors (model/full_model.py:70-74).  The layered path in model.hip batches those into one GEMM per layer, but
// every layer still round-trips its (B*N, C) activations through HBM — 290 MB forward, ~750 MB backward at B=64 —
// and K <= 128 leaves each GEMM workgroup three k-tiles of work between a cold prologue and a 64 KB epilogue.
//
// Here a cloud's whole weight vector (19 011 floats, 77 KB) is parked in LDS and the activations of 32 points
// never leave a wave's registers:
//   * "points as columns": a layer is Z (Cout x 32 pts) = W (Cout x Cin) . H (Cin x 32 pts) on
//     v_mfma_f32_32x32x2_f32.  A = W rows from LDS (padded leading dimension: conflict-free), B = the previous
//     layer's accumulator registers USED AS THEY ARE: the C/D layout of a 32x32 tile puts row
//     kmap(e,h) = (e&3) + 8(e>>2) + 4h in register e of lane half h, and a contraction may visit k in any
//     order, so MFMA step s simply takes k = kmap(s,h): no shuffles, no LDS between layers.
//   * backward recomputes the forward from (points, theta) — nothing is saved by the forward at all — then runs
//     dX the same way (A = W^T read from the same LDS image) and the ReLU mask from the live registers.
//   * dW = Delta . H^T contracts over points, which live in lanes: the Delta/H tiles of 64 points are
//     transposed through a 70 KB LDS stage ([channel][point], 68-float rows: conflict-free ds_read_b128) and the
//     dW tiles are partitioned over the 4 waves, which keep them in accumulators across the workgroup's whole
//     point range.  Bias gradients ride on the A fragments (row sums); dW1/db1 come from one padded tile
//     ([x y z 1] as the B operand).
//   * each workgroup writes its partial d theta; a small kernel adds the partials of a cloud in workgroup order
//     (ordered, atomic-free).
// HBM traffic: points + theta + y forward; points + theta + grad_y + S partial d theta backward.
#include "hp_common.h"
#include <cstdlib>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int C1 = 32, C2 = 64, C3 = 128, C4 = 64;
// theta layout [W1 b1 | W2 b2 | W3 b3 | W4 b4 | W5 b5], W row-major (out, in)   (model/target_network.py:18-29)
constexpr int OW1 = 0, OB1 = OW1 + C1 * 3, OW2 = OB1 + C1, OB2 = OW2 + C2 * C1, OW3 = OB2 + C2, OB3 = OW3 + C3 * C2,
              OW4 = OB3 + C3, OB4 = OW4 + C4 * C3, OW5 = OB4 + C4, OB5 = OW5 + 3 * C4, kTheta = OB5 + 3;
static_assert(kTheta == 19011, "theta size of the published target network");
// LDS image of theta: the three MFMA weight matrices get leading dimensions = 4 (mod 32): 16-byte aligned rows,
// conflict-free for the forward's ds_read_b128 (32 rows x 4 consecutive k) and the backward's ds_read_b32 (32 consecutive i)
constexpr int LD2 = C1 + 4, LD3 = C2 + 4, LD4 = C3 + 4;
constexpr int SW1 = 0, SB1 = SW1 + C1 * 3, SW5 = SB1 + C1, SB5 = SW5 + 3 * C4, SB2 = SB5 + 4, SB3 = SB2 + C2, SB4 = SB3 + C3,
              SW2 = SB4 + C4, SW3 = SW2 + C2 * LD2, SW4 = SW3 + C3 * LD3, kWFloats = SW4 + C4 * LD4;
static_assert(kWFloats % 4 == 0, "the stage behind the weights must stay 16-byte aligned");
// transposed stage: [channel row][64 points], 68-float rows
constexpr int kStagePts = 64, LDS_ST = kStagePts + 4;
constexpr int kStageRows = 4 + C4 + C4 + C3;   // largest group: grad_y (3 -> 4), H4, Delta4, H3
constexpr int kStageFloats = kStageRows * LDS_ST;
constexpr int kBwdLds = kWFloats + kStageFloats;
static_assert(kBwdLds * 4 <= 160 * 1024, "LDS budget of a CU");

__device__ __forceinline__ int kmap(int e, int h) { return (e & 3) + 8 * (e >> 2) + 4 * h; }

__device__ __forceinline__ f32x16 mfma(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

// n floats of theta -> lds rows of `cin` floats with leading dimension ld; 8 loads are in flight per thread (a
// load-store loop one element deep would pay the L2 latency 40-80 times over)
template <int NT>
__device__ __forceinline__ void copy_rows(const float* __restrict__ src, float* __restrict__ dst, int n, int cin, int ld, int tid) {
    for (int base = 0; base < n; base += 8 * NT) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = base + u * NT + tid;
            v[u] = i < n ? src[i] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = base + u * NT + tid;
            if (i < n) dst[(i / cin) * ld + (i % cin)] = v[u];
        }
    }
}

// theta of one cloud -> LDS image (all threads of the workgroup)
template <int NT>
__device__ __forceinline__ void load_theta(const float* __restrict__ th, float* __restrict__ lds, int tid) {
    copy_rows<NT>(th + OW4, lds + SW4, C4 * C3, C3, LD4, tid);
    copy_rows<NT>(th + OW3, lds + SW3, C3 * C2, C2, LD3, tid);
    copy_rows<NT>(th + OW2, lds + SW2, C2 * C1, C1, LD2, tid);
    for (int i = tid; i < C1 * 3 + C1; i += NT) lds[SW1 + i] = th[OW1 + i];           // W1, b1 (contiguous in both)
    for (int i = tid; i < 3 * C4 + 3; i += NT) lds[SW5 + i] = th[OW5 + i];            // W5, b5
    for (int i = tid; i < C2; i += NT) lds[SB2 + i] = th[OB2 + i];
    for (int i = tid; i < C3; i += NT) lds[SB3 + i] = th[OB3 + i];
    for (int i = tid; i < C4; i += NT) lds[SB4 + i] = th[OB4 + i];
}

// layer 1 (3 -> 32) on the VALU: h1[e] = relu(W1[c] . p + b1[c]),  c = kmap(e,h)      (model/target_network.py:33-36)
__device__ __forceinline__ void layer1(const float* __restrict__ lds, float x, float y, float z, int h, f32x16& h1) {
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int c = kmap(e, h);
        float t = x * lds[SW1 + c * 3];
        t = __builtin_fmaf(y, lds[SW1 + c * 3 + 1], t);
        t = __builtin_fmaf(z, lds[SW1 + c * 3 + 2], t);
        h1[e] = fmaxf(t + lds[SB1 + c], 0.f);
    }
}

// hidden layer: out (TO tiles) = relu(W (32*TO x 32*TI, lds, leading dim LD) . in (TI tiles) + b).
// Output tiles are computed two at a time: the two accumulation chains are independent (back-to-back MFMAs on one
// accumulator wait for each other) and share their B operand.  Four MFMA steps (k = 8j+4h .. +3, contiguous in a
// W row) take one ds_read_b128 per tile, issued one block ahead of the MFMAs that consume it: with one or two
// waves per SIMD nothing else hides the LDS latency.
template <int TI, int TO, int LD>
__device__ __forceinline__ void layer_fwd(const float* __restrict__ W, const float* __restrict__ b, const f32x16 (&in)[TI],
                                          f32x16 (&out)[TO], int r, int h) {
    static_assert(TO % 2 == 0, "output tiles come in pairs");
    constexpr int NB = TI * 4;
#pragma unroll
    for (int to = 0; to < TO; to += 2) {
        const float* w0 = W + (to * 32 + r) * LD + 4 * h;
        const float* w1 = w0 + 32 * LD;
        f32x16 acc0, acc1;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc0[e] = acc1[e] = 0.f;
        float4 a0 = *reinterpret_cast<const float4*>(w0), a1 = *reinterpret_cast<const float4*>(w1);
#pragma unroll
        for (int blk = 0; blk < NB; ++blk) {
            float4 n0 = a0, n1 = a1;
            if (blk + 1 < NB) {
                n0 = *reinterpret_cast<const float4*>(w0 + 8 * (blk + 1));
                n1 = *reinterpret_cast<const float4*>(w1 + 8 * (blk + 1));
            }
            __builtin_amdgcn_sched_barrier(0);
            const int ti = blk >> 2, s = (blk & 3) * 4;
            acc0 = mfma(a0.x, in[ti][s], acc0);
            acc1 = mfma(a1.x, in[ti][s], acc1);
            acc0 = mfma(a0.y, in[ti][s + 1], acc0);
            acc1 = mfma(a1.y, in[ti][s + 1], acc1);
            acc0 = mfma(a0.z, in[ti][s + 2], acc0);
            acc1 = mfma(a1.z, in[ti][s + 2], acc1);
            acc0 = mfma(a0.w, in[ti][s + 3], acc0);
            acc1 = mfma(a1.w, in[ti][s + 3], acc1);
            __builtin_amdgcn_sched_barrier(0);
            a0 = n0;
            a1 = n1;
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            out[to][e] = fmaxf(acc0[e] + b[to * 32 + kmap(e, h)], 0.f);
            out[to + 1][e] = fmaxf(acc1[e] + b[to * 32 + 32 + kmap(e, h)], 0.f);
        }
    }
}

// output layer (64 -> 3) on the VALU: each lane half holds half of the channels of its point
__device__ __forceinline__ void layer_out(const float* __restrict__ lds, const f32x16 (&h4)[2], int h, float (&y)[3]) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float t = 0.f;
#pragma unroll
        for (int ti = 0; ti < 2; ++ti)
#pragma unroll
            for (int e = 0; e < 16; ++e) t = __builtin_fmaf(lds[SW5 + c * C4 + ti * 32 + kmap(e, h)], h4[ti][e], t);
        t += __shfl_xor(t, 32, 64);
        y[c] = t + lds[SB5 + c];
    }
}

// ------------------------------------------------------------------------------------------------
// forward: 8 waves per workgroup, 32 points per wave per iteration; 2 workgroups per CU (77 KB LDS each)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512) void target_fwd_kernel(int N, int iters, const float* __restrict__ theta, int theta_ld,
                                                         const float* __restrict__ pts, float* __restrict__ yout) {
    __shared__ __attribute__((aligned(16))) float lds[kWFloats];
    const int cloud = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
    load_theta<512>(theta + (long)cloud * theta_ld, lds, tid);
    __syncthreads();
    const float* P = pts + (long)cloud * N * 3;
    float* Y = yout + (long)cloud * N * 3;
    for (int it = 0; it < iters; ++it) {
        const int p0 = (blockIdx.x * iters + it) * 256 + wave * 32;
        if (p0 >= N) break;
        const int pt = p0 + r, pc = min(pt, N - 1);
        const float x = P[pc * 3], y = P[pc * 3 + 1], z = P[pc * 3 + 2];
        f32x16 h1[1], h2[2], h3[4], h4[2];
        layer1(lds, x, y, z, h, h1[0]);
        layer_fwd<1, 2, LD2>(lds + SW2, lds + SB2, h1, h2, r, h);
        layer_fwd<2, 4, LD3>(lds + SW3, lds + SB3, h2, h3, r, h);
        layer_fwd<4, 2, LD4>(lds + SW4, lds + SB4, h3, h4, r, h);
        float o[3];
        layer_out(lds, h4, h, o);
        if (h == 0 && pt < N) {
            Y[pt * 3] = o[0];
            Y[pt * 3 + 1] = o[1];
            Y[pt * 3 + 2] = o[2];
        }
    }
}

// ------------------------------------------------------------------------------------------------
// forward on the f16 matrix pipe (round 3): the hidden layers' products from two f16 pieces per fp32 operand, three
// v_mfma_f32_32x32x16_f16 per 32x32x16 block, fp32 accumulation — csrc/conv_split.hip's arithmetic (error vs fp64 = the fp32
// chain's) with everything local to the workgroup:
//   weights      per output channel: e_w = 14 - exponent(max_k |W[c,k]|), formed with the split while theta is staged into LDS;
//   activations  per wave (its 32 points x all channels of the layer: the scale only has to be constant along the contraction).
// "Points as columns" carries over: registers 8s..8s+7 of a 32x32 accumulator tile are, converted, the B operand of k-step s
// of a 32x32x16 MFMA — lane half h then holds the channels 16s + 8(j>>2) + 4h + (j&3), j = 0..7 — and the weights' LDS image
// stores each row with its 4-channel granules in that order (granule (s, g, h) at position (s, h, g)), so a lane's A fragment
// is ONE ds_read_b128.  Rows are unpadded; the 16-byte chunk index is XOR-swizzled with the row so that the 16 rows a b128 read
// touches per pass fall on different banks.  77 KB of fp32 weights become 72 KB of f16 pieces + 3 KB of fp32 vectors: still two
// workgroups per CU.
// ------------------------------------------------------------------------------------------------
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

// fp32 part of the forward's f16 LDS image (floats), then the f16 images (halfs, offsets from the halfs base)
constexpr int FW1 = 0, FB1 = FW1 + C1 * 3, FW5 = FB1 + C1, FB5 = FW5 + 3 * C4, FB2 = FB5 + 4, FB3 = FB2 + C2, FB4 = FB3 + C3,
              FS2 = FB4 + C4, FS3 = FS2 + C2, FS4 = FS3 + C3, kFwdFloats = FS4 + C4;
static_assert(kFwdFloats % 4 == 0, "f16 images start 16-byte aligned");
constexpr int HW2 = 0, HW3 = HW2 + 2 * C2 * C1, HW4 = HW3 + 2 * C3 * C2, kFwdHalfs = HW4 + 2 * C4 * C3;
static_assert((kFwdFloats * 4 + kFwdHalfs * 2) * 2 <= 160 * 1024, "two forward workgroups per CU");

template <int CIN>
__device__ __forceinline__ int swz_chunk(int row, int chunk) {   // CIN halfs per row = CIN / 8 chunks of 16 bytes
    return CIN == 32 ? (chunk ^ ((row >> 2) & 3)) : CIN == 64 ? (chunk ^ ((row >> 1) & 7)) : (chunk ^ (row & 15));
}

__device__ __forceinline__ int f_frexp(float v) {   // e of v = f 2^e, f in [0.5, 1); 0 for zero / subnormal
    const int E = (int)((__float_as_uint(v) >> 23) & 0xff);
    return E ? E - 126 : 0;
}
__device__ __forceinline__ float f_pow2(int e) {
    e = max(-126, min(127, e));
    return __uint_as_float((unsigned)(e + 127) << 23);
}

// one hidden layer's weights (COUT x CIN, row-major in theta) -> hi / lo images + per-row unscale factors 2^-e_w
template <int NT, int COUT, int CIN>
__device__ __forceinline__ void split_rows(const float* __restrict__ src, _Float16* __restrict__ hi, _Float16* __restrict__ lo,
                                           float* __restrict__ wsc, int tid) {
    constexpr int TPR = CIN / 4;                       // threads per row (a power of two <= 32)
    constexpr int UNITS = COUT * TPR;
    for (int u0 = 0; u0 < UNITS; u0 += NT) {           // UNITS % NT == 0 or the tail is whole rows: TPR divides NT
        const int u = u0 + tid;
        const bool ok = u < UNITS;
        const int row = ok ? u / TPR : 0, k0 = ok ? (u % TPR) * 4 : 0;
        float v[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = ok ? src[row * CIN + k0 + q] : 0.f;   // (theta rows are not 16-byte aligned)
        float m = fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3])));
#pragma unroll
        for (int o = 1; o < TPR; o <<= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
        const int e = 14 - f_frexp(m);
        const float sc = f_pow2(e);
        f16x4 h4, l4;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float xs = v[q] * sc;
            const _Float16 hh = (_Float16)xs;
            h4[q] = hh;
            l4[q] = (_Float16)(xs - (float)hh);
        }
        if (ok) {
            // granule k0 = 32 ti + 16 s + 8 g + 4 hh  ->  position 32 ti + 16 s + 8 hh + 4 g
            const int g = (k0 >> 3) & 1, hh = (k0 >> 2) & 1, pos = (k0 & ~15) + 8 * hh + 4 * g;
            const int off = row * CIN + swz_chunk<CIN>(row, pos >> 3) * 8 + (pos & 7);
            *reinterpret_cast<f16x4*>(hi + off) = h4;
            *reinterpret_cast<f16x4*>(lo + off) = l4;
            if (k0 == 0) wsc[row] = f_pow2(-e);
        }
    }
}

template <int NT>
__device__ __forceinline__ void load_theta_f16(const float* __restrict__ th, float* __restrict__ lf, _Float16* __restrict__ lh, int tid) {
    split_rows<NT, C4, C3>(th + OW4, lh + HW4, lh + HW4 + C4 * C3, lf + FS4, tid);
    split_rows<NT, C3, C2>(th + OW3, lh + HW3, lh + HW3 + C3 * C2, lf + FS3, tid);
    split_rows<NT, C2, C1>(th + OW2, lh + HW2, lh + HW2 + C2 * C1, lf + FS2, tid);
    for (int i = tid; i < C1 * 3 + C1; i += NT) lf[FW1 + i] = th[OW1 + i];            // W1, b1 (contiguous in both)
    for (int i = tid; i < 3 * C4 + 3; i += NT) lf[FW5 + i] = th[OW5 + i];             // W5, b5
    for (int i = tid; i < C2; i += NT) lf[FB2 + i] = th[OB2 + i];
    for (int i = tid; i < C3; i += NT) lf[FB3 + i] = th[OB3 + i];
    for (int i = tid; i < C4; i += NT) lf[FB4 + i] = th[OB4 + i];
}

// hidden layer on the f16 pipe: out (TO tiles) = relu(W (32*TO x 32*TI) . in (TI tiles) + b)
template <int TI, int TO>
__device__ __forceinline__ void layer_fwd_f16(const _Float16* __restrict__ Whi, const _Float16* __restrict__ Wlo,
                                              const float* __restrict__ wsc, const float* __restrict__ b, const f32x16 (&in)[TI],
                                              f32x16 (&out)[TO], int r, int h) {
    constexpr int CIN = 32 * TI;
    // this wave's activation scale (post-ReLU values: >= 0)
    float m = 0.f;
#pragma unroll
    for (int ti = 0; ti < TI; ++ti)
#pragma unroll
        for (int e = 0; e < 16; ++e) m = fmaxf(m, in[ti][e]);
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    const int ex = 14 - f_frexp(m);
    const float sx = f_pow2(ex), inv = f_pow2(-ex);
    // the input tiles as B fragments, once (the fp32 copies die here); then ONE output tile at a time — a single accumulation
    // chain of this MFMA needs no partner for throughput, and 16 + 8 instead of 32 + 16 registers keep two workgroups per CU
    f16x8 bh[TI][2], bl[TI][2];
#pragma unroll
    for (int ti = 0; ti < TI; ++ti)
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float xs = in[ti][8 * s + j] * sx;
                const _Float16 hh = (_Float16)xs;
                bh[ti][s][j] = hh;
                bl[ti][s][j] = (_Float16)(xs - (float)hh);
            }
#pragma unroll
    for (int to = 0; to < TO; ++to) {
        const int row0 = to * 32 + r;
        f32x16 acc0;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc0[e] = 0.f;
#pragma unroll
        for (int ti = 0; ti < TI; ++ti)
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const int o0 = row0 * CIN + swz_chunk<CIN>(row0, 4 * ti + 2 * s + h) * 8;
                const f16x8 ah0 = *reinterpret_cast<const f16x8*>(Whi + o0), al0 = *reinterpret_cast<const f16x8*>(Wlo + o0);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah0, bh[ti][s], acc0, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah0, bl[ti][s], acc0, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(al0, bh[ti][s], acc0, 0, 0, 0);
            }
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int c0 = to * 32 + kmap(e, h);
            out[to][e] = fmaxf(acc0[e] * (wsc[c0] * inv) + b[c0], 0.f);
        }
    }
}

__device__ __forceinline__ void layer1_at(const float* __restrict__ W1, const float* __restrict__ b1, float x, float y, float z, int h,
                                          f32x16& h1) {
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int c = kmap(e, h);
        float t = x * W1[c * 3];
        t = __builtin_fmaf(y, W1[c * 3 + 1], t);
        t = __builtin_fmaf(z, W1[c * 3 + 2], t);
        h1[e] = fmaxf(t + b1[c], 0.f);
    }
}

__global__ __launch_bounds__(512) void target_fwd_f16_kernel(int N, int iters, const float* __restrict__ theta, int theta_ld,
                                                             const float* __restrict__ pts, float* __restrict__ yout) {
    __shared__ __attribute__((aligned(16))) float lf[kFwdFloats];
    __shared__ __attribute__((aligned(16))) _Float16 lh[kFwdHalfs];
    const int cloud = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
    load_theta_f16<512>(theta + (long)cloud * theta_ld, lf, lh, tid);
    __syncthreads();
    const float* P = pts + (long)cloud * N * 3;
    float* Y = yout + (long)cloud * N * 3;
    for (int it = 0; it < iters; ++it) {
        const int p0 = (blockIdx.x * iters + it) * 256 + wave * 32;
        if (p0 >= N) break;
        const int pt = p0 + r, pc = min(pt, N - 1);
        const float x = P[pc * 3], y = P[pc * 3 + 1], z = P[pc * 3 + 2];
        f32x16 h1[1], h2[2], h3[4], h4[2];
        layer1_at(lf + FW1, lf + FB1, x, y, z, h, h1[0]);
        layer_fwd_f16<1, 2>(lh + HW2, lh + HW2 + C2 * C1, lf + FS2, lf + FB2, h1, h2, r, h);
        layer_fwd_f16<2, 4>(lh + HW3, lh + HW3 + C3 * C2, lf + FS3, lf + FB3, h2, h3, r, h);
        layer_fwd_f16<4, 2>(lh + HW4, lh + HW4 + C4 * C3, lf + FS4, lf + FB4, h3, h4, r, h);
        float o[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {   // output layer (64 -> 3) on the VALU, as layer_out
            float t = 0.f;
#pragma unroll
            for (int ti = 0; ti < 2; ++ti)
#pragma unroll
                for (int e = 0; e < 16; ++e) t = __builtin_fmaf(lf[FW5 + c * C4 + ti * 32 + kmap(e, h)], h4[ti][e], t);
            t += __shfl_xor(t, 32, 64);
            o[c] = t + lf[FB5 + c];
        }
        if (h == 0 && pt < N) {
            Y[pt * 3] = o[0];
            Y[pt * 3 + 1] = o[1];
            Y[pt * 3 + 2] = o[2];
        }
    }
}

// ------------------------------------------------------------------------------------------------
// backward
// ------------------------------------------------------------------------------------------------
// dX: out (TI tiles, Cin) = (W^T . delta (TO tiles, Cout)) * (hprev > 0);  W (32*TO x 32*TI) in lds.
// Two output tiles at a time where there are two; A fragments (4 rows of W per block, ds_read_b32 each) are fetched
// one block ahead like the forward's.
template <int TO, int TI, int LD>
__device__ __forceinline__ void layer_dx(const float* __restrict__ W, const f32x16 (&delta)[TO], const f32x16 (&hprev)[TI],
                                         f32x16 (&out)[TI], int r, int h) {
    constexpr int STEP = TI % 2 == 0 ? 2 : 1;
    constexpr int NB = TO * 4;
#pragma unroll
    for (int ti = 0; ti < TI; ti += STEP) {
        const float* w = W + (4 * h) * LD + ti * 32 + r;     // block blk: rows 8*blk + 4h + u
        f32x16 acc0, acc1;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc0[e] = acc1[e] = 0.f;
        float a0[4], a1[4], n0[4], n1[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            a0[u] = w[u * LD];
            a1[u] = STEP == 2 ? w[u * LD + 32] : 0.f;
        }
#pragma unroll
        for (int blk = 0; blk < NB; ++blk) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                n0[u] = a0[u];
                n1[u] = a1[u];
                if (blk + 1 < NB) {
                    n0[u] = w[(8 * (blk + 1) + u) * LD];
                    if (STEP == 2) n1[u] = w[(8 * (blk + 1) + u) * LD + 32];
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            const int to = blk >> 2, s = (blk & 3) * 4;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                acc0 = mfma(a0[u], delta[to][s + u], acc0);
                if (STEP == 2) acc1 = mfma(a1[u], delta[to][s + u], acc1);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                a0[u] = n0[u];
                a1[u] = n1[u];
            }
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            out[ti][e] = hprev[ti][e] > 0.f ? acc0[e] : 0.f;
            if (STEP == 2) out[ti + 1][e] = hprev[ti + 1][e] > 0.f ? acc1[e] : 0.f;
        }
    }
}

// registers (channels x this wave's 32 points) -> stage rows [row0 + channel][pl + point]
template <int T>
__device__ __forceinline__ void stage_put(float* __restrict__ st, int row0, const f32x16 (&v)[T], int pl, int h) {
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) st[(row0 + t * 32 + kmap(e, h)) * LDS_ST + pl] = v[t][e];
}

// acc (32 x 32) += A(rows arow..+31 of the stage) . B(rows brow..+31)^T over the 64 staged points; returns the sum of
// this lane's A fragments (its row's partial bias gradient).  a_rows / b_rows: rows >= that count read as 0, and
// B row == b_rows reads as 1 when `b_one` (the [x y z 1] operand of layer 1).
template <bool A_PAD, bool B_PAD>
__device__ __forceinline__ float dw_tile(const float* __restrict__ st, int arow, int brow, int r, int h, f32x16& acc,
                                         int a_rows = 32, int b_rows = 32, bool b_one = false) {
    float asum = 0.f;
    const bool a_zero = A_PAD && r >= a_rows;
    const bool b_zero = B_PAD && r >= b_rows, b_is_one = B_PAD && b_one && r == b_rows;
    const float* ap = st + (arow + (a_zero ? 0 : r)) * LDS_ST + 4 * h;
    const float* bp = st + (brow + ((b_zero) ? 0 : r)) * LDS_ST + 4 * h;
#pragma unroll
    for (int q = 0; q < kStagePts / 8; ++q) {
        float4 a = *reinterpret_cast<const float4*>(ap + 8 * q);
        float4 b = *reinterpret_cast<const float4*>(bp + 8 * q);
        if (A_PAD && a_zero) a = make_float4(0.f, 0.f, 0.f, 0.f);
        if (B_PAD && b_zero) b = b_is_one ? make_float4(1.f, 1.f, 1.f, 1.f) : make_float4(0.f, 0.f, 0.f, 0.f);
        asum += (a.x + a.y) + (a.z + a.w);
        acc = mfma(a.x, b.x, acc);
        acc = mfma(a.y, b.y, acc);
        acc = mfma(a.z, b.z, acc);
        acc = mfma(a.w, b.w, acc);
    }
    return asum;
}

// two dW tiles that share their B rows (A rows arow0 / arow1): independent accumulation chains, fragments fetched one
// block (4 MFMA steps = 8 points) ahead
__device__ __forceinline__ void dw_pair_a(const float* __restrict__ st, int arow0, int arow1, int brow, int r, int h,
                                          f32x16& acc0, f32x16& acc1, float& asum0, float& asum1) {
    const float* a0p = st + (arow0 + r) * LDS_ST + 4 * h;
    const float* a1p = st + (arow1 + r) * LDS_ST + 4 * h;
    const float* bp = st + (brow + r) * LDS_ST + 4 * h;
    constexpr int NQ = kStagePts / 8;
    float4 a0 = *reinterpret_cast<const float4*>(a0p), a1 = *reinterpret_cast<const float4*>(a1p);
    float4 b = *reinterpret_cast<const float4*>(bp);
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        float4 na0 = a0, na1 = a1, nb = b;
        if (q + 1 < NQ) {
            na0 = *reinterpret_cast<const float4*>(a0p + 8 * (q + 1));
            na1 = *reinterpret_cast<const float4*>(a1p + 8 * (q + 1));
            nb = *reinterpret_cast<const float4*>(bp + 8 * (q + 1));
        }
        __builtin_amdgcn_sched_barrier(0);
        acc0 = mfma(a0.x, b.x, acc0);
        acc1 = mfma(a1.x, b.x, acc1);
        acc0 = mfma(a0.y, b.y, acc0);
        acc1 = mfma(a1.y, b.y, acc1);
        acc0 = mfma(a0.z, b.z, acc0);
        acc1 = mfma(a1.z, b.z, acc1);
        acc0 = mfma(a0.w, b.w, acc0);
        acc1 = mfma(a1.w, b.w, acc1);
        asum0 += (a0.x + a0.y) + (a0.z + a0.w);
        asum1 += (a1.x + a1.y) + (a1.z + a1.w);
        __builtin_amdgcn_sched_barrier(0);
        a0 = na0;
        a1 = na1;
        b = nb;
    }
}

// two dW tiles that share their A rows (B rows brow0 / brow1)
__device__ __forceinline__ void dw_pair_b(const float* __restrict__ st, int arow, int brow0, int brow1, int r, int h,
                                          f32x16& acc0, f32x16& acc1, float& asum) {
    const float* ap = st + (arow + r) * LDS_ST + 4 * h;
    const float* b0p = st + (brow0 + r) * LDS_ST + 4 * h;
    const float* b1p = st + (brow1 + r) * LDS_ST + 4 * h;
    constexpr int NQ = kStagePts / 8;
    float4 a = *reinterpret_cast<const float4*>(ap);
    float4 b0 = *reinterpret_cast<const float4*>(b0p), b1 = *reinterpret_cast<const float4*>(b1p);
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        float4 na = a, nb0 = b0, nb1 = b1;
        if (q + 1 < NQ) {
            na = *reinterpret_cast<const float4*>(ap + 8 * (q + 1));
            nb0 = *reinterpret_cast<const float4*>(b0p + 8 * (q + 1));
            nb1 = *reinterpret_cast<const float4*>(b1p + 8 * (q + 1));
        }
        __builtin_amdgcn_sched_barrier(0);
        acc0 = mfma(a.x, b0.x, acc0);
        acc1 = mfma(a.x, b1.x, acc1);
        acc0 = mfma(a.y, b0.y, acc0);
        acc1 = mfma(a.y, b1.y, acc1);
        acc0 = mfma(a.z, b0.z, acc0);
        acc1 = mfma(a.z, b1.z, acc1);
        acc0 = mfma(a.w, b0.w, acc0);
        acc1 = mfma(a.w, b1.w, acc1);
        asum += (a.x + a.y) + (a.z + a.w);
        __builtin_amdgcn_sched_barrier(0);
        a = na;
        b0 = nb0;
        b1 = nb1;
    }
}

// stage row groups
constexpr int RA_GY = 0, RA_H4 = 4, RA_D4 = RA_H4 + C4, RA_H3 = RA_D4 + C4;   // group A: 260 rows
constexpr int RB_D3 = 0, RB_H2 = C3;                                          // group B: 192 rows
constexpr int RC_D2 = 0, RC_H1 = C2, RC_D1 = RC_H1 + C1, RC_H0 = RC_D1 + C1;   // group C: 131 rows

// One workgroup = 4 waves = 128 points per iteration, `iters` iterations; grid (S, B).  partial: (B, S, kTheta).
__global__ __launch_bounds__(256, 1) void target_bwd_kernel(int N, int iters, const float* __restrict__ theta, int theta_ld,
                                                            const float* __restrict__ pts, const float* __restrict__ gy,
                                                            float* __restrict__ partial) {
    __shared__ __attribute__((aligned(16))) float lds[kBwdLds];
    float* st = lds + kWFloats;
    const int cloud = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
    load_theta<256>(theta + (long)cloud * theta_ld, lds, tid);
    const float* P = pts + (long)cloud * N * 3;
    const float* G = gy + (long)cloud * N * 3;

    // this wave's share of d theta (see the table in the file header)
    f32x16 acc4[2], acc3[2], accs, acc1;   // dW4[:, wave-th cin tile], dW3[wave-th cout tile, :], small tile, dW1|db1
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        acc4[0][e] = acc4[1][e] = acc3[0][e] = acc3[1][e] = accs[e] = acc1[e] = 0.f;
    }
    float db4[2] = {0.f, 0.f}, db3 = 0.f, dbs = 0.f;   // row sums of this lane's A fragments
    const int pl = (wave & 1) * 32 + r;                // this lane's point inside a 64-point stage half

    for (int it = 0; it < iters; ++it) {
        const int p0 = (blockIdx.x * iters + it) * 128;
        if (p0 >= N) break;                            // uniform over the workgroup
        const int pt = p0 + wave * 32 + r, pc = min(pt, N - 1);
        const bool live = pt < N;
        const float x = P[pc * 3], y = P[pc * 3 + 1], z = P[pc * 3 + 2];
        float g[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) g[c] = live ? G[pc * 3 + c] : 0.f;
        if (it == 0) __syncthreads();                  // theta image complete

        // ---- recompute the forward
        f32x16 h1[1], h2[2], h3[4], h4[2];
        layer1(lds, x, y, z, h, h1[0]);
        layer_fwd<1, 2, LD2>(lds + SW2, lds + SB2, h1, h2, r, h);
        layer_fwd<2, 4, LD3>(lds + SW3, lds + SB3, h2, h3, r, h);
        layer_fwd<4, 2, LD4>(lds + SW4, lds + SB4, h3, h4, r, h);

        // ---- delta4 = (W5^T grad_y) * (h4 > 0)
        f32x16 d4[2];
#pragma unroll
        for (int ti = 0; ti < 2; ++ti)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int k = ti * 32 + kmap(e, h);
                float t = g[0] * lds[SW5 + k];
                t = __builtin_fmaf(g[1], lds[SW5 + C4 + k], t);
                t = __builtin_fmaf(g[2], lds[SW5 + 2 * C4 + k], t);
                d4[ti][e] = h4[ti][e] > 0.f ? t : 0.f;
            }

        // ---- group A: dW5 (+db5), dW4 (+db4)
#pragma unroll 1
        for (int half = 0; half < 2; ++half) {
            __syncthreads();
            if ((wave >> 1) == half) {
                if (h == 0) {
                    st[(RA_GY + 0) * LDS_ST + pl] = g[0];
                    st[(RA_GY + 1) * LDS_ST + pl] = g[1];
                    st[(RA_GY + 2) * LDS_ST + pl] = g[2];
                }
                stage_put<2>(st, RA_H4, h4, pl, h);
                stage_put<2>(st, RA_D4, d4, pl, h);
                stage_put<4>(st, RA_H3, h3, pl, h);
            }
            __syncthreads();
            dw_pair_a(st, RA_D4, RA_D4 + 32, RA_H3 + wave * 32, r, h, acc4[0], acc4[1], db4[0], db4[1]);   // db4: wave 0's copy is stored
            if (wave >= 2) {
                const float s5 = dw_tile<true, false>(st, RA_GY, RA_H4 + (wave - 2) * 32, r, h, accs, 3);
                if (wave == 2) dbs += s5;
            }
        }

        // ---- delta3, group B: dW3 (+db3)
        f32x16 d3[4];
        layer_dx<2, 4, LD4>(lds + SW4, d4, h3, d3, r, h);
#pragma unroll 1
        for (int half = 0; half < 2; ++half) {
            __syncthreads();
            if ((wave >> 1) == half) {
                stage_put<4>(st, RB_D3, d3, pl, h);
                stage_put<2>(st, RB_H2, h2, pl, h);
            }
            __syncthreads();
            dw_pair_b(st, RB_D3 + wave * 32, RB_H2, RB_H2 + 32, r, h, acc3[0], acc3[1], db3);
        }

        // ---- delta2, delta1, group C: dW2 (+db2), dW1|db1
        f32x16 d2[2], d1[1];
        layer_dx<4, 2, LD3>(lds + SW3, d3, h2, d2, r, h);
        layer_dx<2, 1, LD2>(lds + SW2, d2, h1, d1, r, h);
#pragma unroll 1
        for (int half = 0; half < 2; ++half) {
            __syncthreads();
            if ((wave >> 1) == half) {
                stage_put<2>(st, RC_D2, d2, pl, h);
                stage_put<1>(st, RC_H1, h1, pl, h);
                stage_put<1>(st, RC_D1, d1, pl, h);
                if (h == 0) {
                    st[(RC_H0 + 0) * LDS_ST + pl] = x;
                    st[(RC_H0 + 1) * LDS_ST + pl] = y;
                    st[(RC_H0 + 2) * LDS_ST + pl] = z;
                }
            }
            __syncthreads();
            if (wave < 2) dbs += dw_tile<false, false>(st, RC_D2 + wave * 32, RC_H1, r, h, accs);
            if (wave == 3) (void)dw_tile<false, true>(st, RC_D1, RC_H0, r, h, acc1, 32, 3, true);
        }
    }

    // ---- this workgroup's partial d theta, theta layout.  D tile: row = kmap(e,h), col = r.
    float* out = partial + ((long)cloud * gridDim.x + blockIdx.x) * kTheta;
#pragma unroll
    for (int to = 0; to < 2; ++to)
#pragma unroll
        for (int e = 0; e < 16; ++e) out[OW4 + (to * 32 + kmap(e, h)) * C3 + wave * 32 + r] = acc4[to][e];
#pragma unroll
    for (int ti = 0; ti < 2; ++ti)
#pragma unroll
        for (int e = 0; e < 16; ++e) out[OW3 + (wave * 32 + kmap(e, h)) * C2 + ti * 32 + r] = acc3[ti][e];
    {
        const float t = db3 + __shfl_xor(db3, 32, 64);
        if (h == 0) out[OB3 + wave * 32 + r] = t;
    }
    if (wave < 2) {
#pragma unroll
        for (int e = 0; e < 16; ++e) out[OW2 + (wave * 32 + kmap(e, h)) * C1 + r] = accs[e];
        const float t = dbs + __shfl_xor(dbs, 32, 64);
        if (h == 0) out[OB2 + wave * 32 + r] = t;
    } else {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int c = kmap(e, h);
            if (c < 3) out[OW5 + c * C4 + (wave - 2) * 32 + r] = accs[e];
        }
        if (wave == 2) {
            const float t = dbs + __shfl_xor(dbs, 32, 64);
            if (h == 0 && r < 3) out[OB5 + r] = t;
        }
    }
    if (wave == 0) {
#pragma unroll
        for (int to = 0; to < 2; ++to) {
            const float t = db4[to] + __shfl_xor(db4[to], 32, 64);
            if (h == 0) out[OB4 + to * 32 + r] = t;
        }
    }
    if (wave == 3) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int c = kmap(e, h);
            if (r < 3) out[OW1 + c * 3 + r] = acc1[e];
            else if (r == 3) out[OB1 + c] = acc1[e];
        }
    }
}

// grad_theta[b][i] = sum_s partial[b][s][i], s ascending
__global__ __launch_bounds__(256) void target_reduce_kernel(int S, const float* __restrict__ partial, float* __restrict__ gth,
                                                            int theta_ld) {
    const int cloud = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
    if (i >= kTheta) return;
    const float* p = partial + (long)cloud * S * kTheta + i;
    float v = 0.f;
#pragma unroll 4
    for (int s = 0; s < S; ++s) v += p[(long)s * kTheta];
    gth[(long)cloud * theta_ld + i] = v;
}

// workgroups per cloud for the backward: enough to cover the chip, each at least one 128-point iteration
int bwd_splits(int B, int N) {
    const int blocks = (N + 127) / 128;
    int s = (256 + B - 1) / B;                          // one resident workgroup per CU (147 KB of LDS each)
    if (s > blocks) s = blocks;
    if (s > 16) s = 16;
    return s < 1 ? 1 : s;
}

bool g_fwd_f16 = [] {
    const char* e = getenv("HP_TARGET_F16");
    return !(e && e[0] == '0');
}();

// (Round 3's f16-pipe backward prototype — 147 us against this kernel's 151, 512 VGPRs and 240 B of scratch — lives as a patch in
// tools/micro/target_bwd_f16.patch, not in the shipped library: docs/DESIGN_HISTORY.md 7b.)
}  // namespace

// The fused forward's hidden layers on the f16 matrix pipe with split fp32 operands (default) or on the fp32 one (0; also
// environment HP_TARGET_F16=0).  Returns the previous setting.
HP_API int hp_target_fused_set_f16(int on) {
    const int was = g_fwd_f16;
    g_fwd_f16 = on != 0;
    return was;
}

// 1 when (n_hidden, channels) is the architecture these kernels are written for
HP_API int hp_target_fused_supported(int n_hidden, const int* channels) {
    return n_hidden == 4 && channels && channels[0] == C1 && channels[1] == C2 && channels[2] == C3 && channels[3] == C4;
}

HP_API long hp_target_fused_workspace_floats(int B, int N) { return (long)B * bwd_splits(B, N) * kTheta; }

HP_API int hp_target_fused_forward(int B, int N, const float* theta, int theta_ld, const float* pts, float* y, hipStream_t stream) {
    HP_CHECK_ARG(B > 0 && B <= 65535 && N > 0 && theta && pts && y && theta_ld >= kTheta);
    const int blocks = (N + 255) / 256;
    int per = (blocks * B + 255) / 256;                // one resident workgroup per CU: theta is staged once per CU
    if (per < 1) per = 1;
    const int gx = (blocks + per - 1) / per;
    if (g_fwd_f16)
        hipLaunchKernelGGL(target_fwd_f16_kernel, dim3(gx, B), dim3(512), 0, stream, N, per, theta, theta_ld, pts, y);
    else
        hipLaunchKernelGGL(target_fwd_kernel, dim3(gx, B), dim3(512), 0, stream, N, per, theta, theta_ld, pts, y);
    HP_RETURN_LAST_ERROR();
}

HP_API int hp_target_fused_backward(int B, int N, const float* theta, int theta_ld, const float* pts, const float* grad_y,
                             float* grad_theta, float* ws, hipStream_t stream) {
    HP_CHECK_ARG(B > 0 && B <= 65535 && N > 0 && theta && pts && grad_y && grad_theta && ws && theta_ld >= kTheta);
    const int S = bwd_splits(B, N);
    const int blocks = (N + 127) / 128;
    const int iters = (blocks + S - 1) / S;
    hipLaunchKernelGGL(target_bwd_kernel, dim3(S, B), dim3(256), 0, stream, N, iters, theta, theta_ld, pts, grad_y, ws);
    hipLaunchKernelGGL(target_reduce_kernel, dim3((kTheta + 255) / 256, B), dim3(256), 0, stream, S, ws, grad_theta, theta_ld);
    HP_RETURN_LAST_ERROR();
}
