// The encoders' pointwise-conv stack (/root/reference/model/encoder.py:14-28, 43-45: Conv1d(k=1)+ReLU x4, Conv1d, max over
// points) on the f16 matrix pipe of gfx950 with the accuracy of the fp32 fma chain.
//
// v_mfma_f32_32x32x2_f32 runs at 1/16 of the f16/bf16 matrix rate (MI355X_MICROARCH.md, Matrix cores) and these four GEMMs
// (113.8 GFLOP per step for the two encoders) were 1.0 ms of a 3.6 ms step at 0.70-0.79 of THAT roofline.  Here every fp32
// operand a is multiplied by an exact power of two and split into two f16 pieces
//     a * 2^e = hi + lo + r,   hi = rne16(a 2^e),  lo = rne16(a 2^e - hi),  |hi - a 2^e| <= 2^-11 |a 2^e|,  |r| <= 2^-23 |a 2^e|
// (f16 carries 11 significant bits: after hi the residual has up to 13 bits of a's 24, lo keeps 11 or 12 of them, so r is zero
// or ONE fp32 ulp of a — for roughly a quarter of the operands), and a.b is formed as hi.hi + hi.lo + lo.hi by three
// v_mfma_f32_32x32x16_f16: exact products, fp32 accumulation; the dropped lo.lo is <= 2^-22 |a b|.  Operands are thus carried
// to 22-23 bits, products to ~2^-22: "as close to fp64 as the fp32 chain" below is a MEASURED statement with stated factors
// (rms <= 1.5x, max <= 2.5x the k-ordered fp32 fma chain's error, max <= 2e-6 of the layer's scale: tests/test_model_gpu.py
// test_conv_stack_split_f16_is_as_close_to_fp64_as_the_fp32_chain), not a consequence of these bounds.
// Three f16 MFMAs per 32x32x16 block cost 96 cycles where eight f32 ones cost 512.  Measured against fp64 on the step's
// shapes: rms error ratio 0.9-1.2, max 0.6-1.5 against the fp32 chain of gemm.hip (tools/micro/gemm_f16x2.hip) — the
// arithmetic type of the path stays f32, only the rounding pattern differs.
// The BACKWARD of these layers (enc_bwd.hip) contracts with the unsplit fp32 weights and the stored activations: it is the
// gradient of the fp32 layer, evaluated at activations the forward computed with W's 22-23-bit image — a 2^-22-relative
// inconsistency between the function differentiated and the function evaluated, far inside the gradient bars (5e-4 against the
// reference, 2e-5 against the fp64 oracle) and stated here so that nobody has to find it.
//
// Scales (f16 has 5 exponent bits; the split needs |a 2^e| in the normal range, and hi must not overflow):
//   weights      per output channel (= per row of W): e_w[n] = 14 - exponent(max_k |W[n,k]|), formed with the split itself by
//                conv_split_prep_kernel once per forward (the weights change only in the optimiser step);
//   activations  per 128-row tile (the rows one workgroup of the NEXT layer contracts): the layer that WRITES h_l also forms
//                max|h_l| over each tile (post-ReLU values are >= 0, so an unsigned integer atomicMax on the float bits —
//                one per workgroup, the tile's column workgroups meet in one word — is an exact, order-independent max:
//                run-to-run identical); the layer that READS the tile derives e_x = 14 - exponent(max) from it.  Elements
//                below 2^-28 of their tile's max lose relative precision in `lo` (f16 subnormals), i.e. the absolute error
//                floor is 2^-39 of the largest activation among the tile's 128 points — far below the
//                2^-24-of-the-largest-term rounding of any fp32 dot product; an outlier point costs precision only to the
//                127 points that share its tile.
//   The accumulator is unscaled by the exact factor 2^-(e_x + e_w[n]) before bias / ReLU.
//
// Kernel: 128 x 128 output tile per 4-wave workgroup (64 x 64 per wave), 32-deep k-tiles; the activation tile is loaded
// as fp32 (16-byte loads), split in registers and stored as two f16 LDS images, the weight tile is copied from the pre-split
// f16 array (hi and lo pieces interleaved per k-tile: full 128-byte lines); swizzled 64-byte LDS rows keep the stage writes and the ds_read_b128 fragment reads conflict-free.
// tools/micro/gemm_f16x2.hip holds the variants measured against this one (cross-tile software pipeline with a raw barrier,
// double-buffered LDS, 256 x 128 and 128 x 256 tiles, the 16x16x32 MFMA shape): all within +-6 % — the kernel sits at the
// ~0.9 PFLOP/s (executed f16) that cdna_hip_programming.md quotes as the ceiling of two-barrier 128^2 structures, at a clock
// the chip holds at ~1.9 GHz under this load (GRBM_GUI_ACTIVE / 8 / time; MFMA busy 0.49 of those cycles).
// Epilogues: bias + ReLU + store + max (layers 2-4), or the fused max-pool of gemm.hip's HP_GEMM_COLMAX (layer 5).
// Layer 1 (K = 3) is a plain fma kernel in the k order of the general GEMM (bit-identical to it), which also forms max|h1|.
#include "hp_common.h"
#include "hp_conv_split.h"

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int BM = 128, BN = 128, BK = 32, ROW = BK;
constexpr int kTarget = 14;   // scaled maxima land in [2^13, 2^14): hi cannot overflow (f16 max 65504)

// exponent e of a float's frexp form v = f 2^e, f in [0.5, 1), from its bits (0 for zero / subnormal inputs)
__device__ __forceinline__ int frexp_exp(unsigned bits) {
    const int E = (int)((bits >> 23) & 0xff);
    return E ? E - 126 : 0;
}
__device__ __forceinline__ float pow2f(int e) {   // 2^e, e clamped into the normal range
    e = max(-126, min(127, e));
    return __uint_as_float((unsigned)(e + 127) << 23);
}

// ---------------------------------------------------------------------------------------------------------------------
// weights -> (hi, lo, e_w): one wave per row of W
struct PrepParams {
    const float* W[4];        // conv_w[1..4] of encoder 0
    long sWz[4];              // distance to encoder 1's tensor
    _Float16* hl;             // split area of encoder 0: interleaved pieces
    int* wexp;
    unsigned* amax;           // 4 * tiles_pad words to clear (the per-tile maxima of layers 1..4)
    long n_amax;
    long sArea;               // distance (floats) between the two encoders' split areas
    int* fmt;                 // the workspace's format word (hp_conv_split.h)
    int fmt_value;
};
constexpr int kRows[5] = {0, 128, 384, 896, 1408};                 // first row of layer 2..5 in wexp
constexpr long kWOff[4] = {0, 8192, 40960, 172032};                // first element of layer 2..5 in hi / lo
constexpr int kK[4] = {64, 128, 256, 512};

__global__ __launch_bounds__(256) void conv_split_prep_kernel(const PrepParams p) {
    const int z = blockIdx.y, lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < p.n_amax; i += (long)gridDim.x * 256) p.amax[z * p.sArea + i] = 0u;
    if (blockIdx.x == 0 && threadIdx.x == 0) p.fmt[z * p.sArea] = p.fmt_value;
    int l = 0;
    while (l < 3 && row >= kRows[l + 1]) ++l;
    const int n = row - kRows[l], K = kK[l];
    const float* w = p.W[l] + z * p.sWz[l] + (long)n * K;
    float v[8];
    float m = 0.f;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int k = lane + 64 * u;
        v[u] = k < K ? w[k] : 0.f;
        m = fmaxf(m, fabsf(v[u]));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    const int e = kTarget - frexp_exp(__float_as_uint(m));
    const float s = pow2f(e);
    // pieces interleaved per 32-deep k-tile: row n = [hi(k 0..31) | lo(k 0..31) | hi(k 32..63) | ...]: one full 128-byte line
    // per (row, k-tile) for the GEMM's weight-tile loads (two half-used lines with separate hi / lo arrays: measured -6 % on conv5)
    _Float16* hl = p.hl + 2 * z * p.sArea + 2 * kWOff[l] + (long)n * 2 * K;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int k = lane + 64 * u;
        if (k < K) {
            const float xs = v[u] * s;
            const _Float16 hh = (_Float16)xs;
            hl[(k >> 5) * 64 + (k & 31)] = hh;
            hl[(k >> 5) * 64 + 32 + (k & 31)] = (_Float16)(xs - (float)hh);
        }
    }
    if (lane == 0) p.wexp[z * p.sArea + row] = e;
}

// ---------------------------------------------------------------------------------------------------------------------
// layer 1: h1 = relu(x W1^T + b1), K = 3.  fma order of gemm.hip's MFMA tiles (k = 0, 1, 2 chained from 0, then + bias).
__global__ __launch_bounds__(256) void conv1_kernel(const float* __restrict__ x, long sXz, const float* __restrict__ W, long sWz,
                                                    const float* __restrict__ b, long sBz, float* __restrict__ h, long sHz,
                                                    unsigned* __restrict__ amax, long sAz, long R) {
    __shared__ float smax[4];
    const int z = blockIdx.y;
    x += z * sXz; W += z * sWz; b += z * sBz; h += z * sHz;
    const int c4 = (threadIdx.x & 15) * 4;
    float w0[4], w1[4], w2[4], bb[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        w0[u] = W[(c4 + u) * 3];
        w1[u] = W[(c4 + u) * 3 + 1];
        w2[u] = W[(c4 + u) * 3 + 2];
        bb[u] = b[c4 + u];
    }
    // one 128-row tile per block pass (8 rows per thread, their coordinate loads in flight together): the block owns the
    // tile's maximum — a plain store
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
        float m = 0.f;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const long row = base + 16 * q;
            f32x4 o;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float v = __builtin_fmaf(xv[q][2], w2[u], __builtin_fmaf(xv[q][1], w1[u], __builtin_fmaf(xv[q][0], w0[u], 0.f)));
                o[u] = fmaxf(v + bb[u], 0.f);
            }
            if (row < R) {
                m = fmaxf(fmaxf(m, fmaxf(o[0], o[1])), fmaxf(o[2], o[3]));
                *reinterpret_cast<f32x4*>(h + row * 64 + c4) = o;
            }
        }
        m = hp::wave_max(m);
        __syncthreads();
        if ((threadIdx.x & 63) == 0) smax[threadIdx.x >> 6] = m;
        __syncthreads();
        if (threadIdx.x == 0) amax[z * sAz + tile] = __float_as_uint(fmaxf(fmaxf(smax[0], smax[1]), fmaxf(smax[2], smax[3])));
    }
}

// ---------------------------------------------------------------------------------------------------------------------
struct CsParams {
    const float* X;            // (M, K) row-major activations of encoder 0
    long sXz;
    const _Float16* Whl;       // (N, 2K) split weights of this layer, encoder 0: per row and 32-deep k-tile [hi 32 | lo 32]
    const int* wexp;           // (N) weight exponents
    const unsigned* amax_in;   // max of X per 128-row tile (float bits)
    unsigned* amax_out;        // max of C per 128-row tile (NULL: not formed)
    long sArea;                // distance (floats) between the encoders' split areas
    const float* bias;
    long sBiasz;
    float* C;                  // (M, N)
    long sCz;
    float* cmax;               // COLMAX: (M / 128, N) partials, as gemm.hip's HP_GEMM_COLMAX
    int* cidx;
    int M, N, K, relu, group_rows, tiles_n;
};

template <bool COLMAX>
__global__ __launch_bounds__(256, 3) void conv_split_kernel(const CsParams p) {
    // four f16 images [rows][32 k] of 64-byte rows; 16-byte chunk c of row r lives at chunk c ^ ((r >> 2) & 3): the b64 / b128
    // stage writes and the b128 fragment reads (16 lanes = 16 rows, one chunk) are bank-conflict-free without padding
    __shared__ __attribute__((aligned(16))) _Float16 Ah[BM * ROW];
    __shared__ __attribute__((aligned(16))) _Float16 Al[BM * ROW];
    __shared__ __attribute__((aligned(16))) _Float16 Bh[BN * ROW];
    __shared__ __attribute__((aligned(16))) _Float16 Bl[BN * ROW];
    // XCD-aware bijective remap of the tile id: the column tiles of one row panel run on one XCD (they share its L2 copy)
    const int nwg = gridDim.x;
    int bid = blockIdx.x;
    {
        const int q = nwg >> 3, r8 = nwg & 7, xcd = bid & 7;
        bid = (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + (bid >> 3);
    }
    const int z = blockIdx.y;
    const int tile_n = bid % p.tiles_n, tile_m = bid / p.tiles_n;
    const int row0 = tile_m * BM, col0 = tile_n * BN;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, r = lane & 31, h = lane >> 5;
    const int wm = w >> 1, wn = w & 1;
    const int K = p.K, M = p.M;
    const float* X = p.X + z * p.sXz;
    const _Float16* Whl = p.Whl + 2 * z * p.sArea;

    const int ex = kTarget - frexp_exp(p.amax_in[z * p.sArea + tile_m]);
    const float sx = pow2f(ex);

    const float* pa[4];
    int a_off[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int idx = tid + e * 256, row = idx >> 3, kq = idx & 7;
        pa[e] = X + (long)min(row0 + row, M - 1) * K + 4 * kq;
        a_off[e] = row * ROW + (((kq >> 1) ^ ((row >> 2) & 3)) << 3) + 4 * (kq & 1);
    }
    // weight tile: 128 rows x [hi 32 | lo 32] = 8 chunks of 16 bytes per row, 4 per thread
    const _Float16* pb[4];
    _Float16* bdst[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int idx = tid + e * 256, row = idx >> 3, c = idx & 7;
        pb[e] = Whl + (long)(col0 + row) * 2 * K + 8 * c;
        bdst[e] = ((c >> 2) ? Bl : Bh) + row * ROW + (((c & 3) ^ ((row >> 2) & 3)) << 3);
    }

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    f32x4 ra[4];
    u32x4 rb[4];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int e = 0; e < 4; ++e) ra[e] = *reinterpret_cast<const f32x4*>(pa[e] + k0);
#pragma unroll
        for (int e = 0; e < 4; ++e) rb[e] = *reinterpret_cast<const u32x4*>(pb[e] + 2 * k0);
    };
    auto stage = [&]() {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            f16x4 hi, lo;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float xs = ra[e][u] * sx;
                const _Float16 hh = (_Float16)xs;
                hi[u] = hh;
                lo[u] = (_Float16)(xs - (float)hh);
            }
            *reinterpret_cast<f16x4*>(&Ah[a_off[e]]) = hi;
            *reinterpret_cast<f16x4*>(&Al[a_off[e]]) = lo;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) *reinterpret_cast<u32x4*>(bdst[e]) = rb[e];
    };
    auto compute = [&]() {
#pragma unroll
        for (int t = 0; t < BK / 16; ++t) {
            f16x8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int ar = wm * 64 + i * 32 + r;
                const int o = ar * ROW + (((2 * t + h) ^ ((ar >> 2) & 3)) << 3);
                ah[i] = *reinterpret_cast<const f16x8*>(&Ah[o]);
                al[i] = *reinterpret_cast<const f16x8*>(&Al[o]);
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int br = wn * 64 + j * 32 + r;
                const int o = br * ROW + (((2 * t + h) ^ ((br >> 2) & 3)) << 3);
                bh[j] = *reinterpret_cast<const f16x8*>(&Bh[o]);
                bl[j] = *reinterpret_cast<const f16x8*>(&Bl[o]);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl[j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh[j], acc[i][j], 0, 0, 0);
                }
        }
    };

    fetch(0);
    for (int k0 = 0; k0 < K; k0 += BK) {
        stage();
        __syncthreads();
        if (k0 + BK < K) fetch(k0 + BK);
        compute();
        __syncthreads();
    }

    // epilogue.  C/D map of the 32x32 f32 tile: col = lane&31, row = (e&3) + 8*(e>>2) + 4*(lane>>5)
    const float* bias = p.bias + z * p.sBiasz;
    const int* wexp = p.wexp + z * p.sArea;
    if (COLMAX) {
        // fused max-pool over this tile's rows (model/encoder.py:45): the first row attaining the max wins
        float* smax = reinterpret_cast<float*>(Ah);   // [2][BN]
        int* sidx = reinterpret_cast<int*>(Al);       // [2][BN]
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int cl = wn * 64 + j * 32 + r, col = col0 + cl;
            const float us = pow2f(-ex - wexp[col]), bv = bias[col];
            float best = -__builtin_inff();
            int bi = 0x7fffffff;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int e = 0; e < 16; ++e) {   // rows ascend with (i, e) for a fixed lane half
                    const int row = row0 + wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                    const float v = acc[i][j][e] * us + bv;
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
                smax[wm * BN + cl] = best;
                sidx[wm * BN + cl] = bi;
            }
        }
        __syncthreads();
        if (tid < BN) {
            float best = smax[tid];
            int bi = sidx[tid];
            if (smax[BN + tid] > best) {   // the second wave row holds larger rows: strict > keeps the first row
                best = smax[BN + tid];
                bi = sidx[BN + tid];
            }
            p.cmax[z * p.sCz + (long)tile_m * p.N + col0 + tid] = best;
            p.cidx[z * p.sCz + (long)tile_m * p.N + col0 + tid] = bi % p.group_rows;
        }
        return;
    }
    float* C = p.C + z * p.sCz;
    float m = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = col0 + wn * 64 + j * 32 + r;
            const float us = pow2f(-ex - wexp[col]), bv = bias[col];
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = row0 + wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                float v = acc[i][j][e] * us + bv;
                if (p.relu) v = fmaxf(v, 0.f);
                if (row < M) {
                    C[(long)row * p.N + col] = v;
                    m = fmaxf(m, v);
                }
            }
        }
    if (p.amax_out) {
        float* smax = reinterpret_cast<float*>(Ah);
        m = hp::wave_max(m);
        if (lane == 0) smax[w] = m;
        __syncthreads();
        if (tid == 0) {
            m = fmaxf(fmaxf(smax[0], smax[1]), fmaxf(smax[2], smax[3]));
            atomicMax(p.amax_out + z * p.sArea + tile_m, __float_as_uint(m));
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// stand-alone form (hp_gemm_f16x2_*): any fp32 X (M, K) and W (N, K)
__global__ __launch_bounds__(256) void absmax_kernel(const float* __restrict__ x, long M, int K, unsigned* __restrict__ amax) {
    __shared__ float smax[4];
    const long r0 = (long)blockIdx.x * 128;
    const long count = (min(r0 + 128, M) - r0) * K;
    const float* p = x + r0 * K;
    float m = 0.f;
    for (long i = threadIdx.x; i < count; i += 256) m = fmaxf(m, fabsf(p[i]));
    m = hp::wave_max(m);
    if ((threadIdx.x & 63) == 0) smax[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) amax[blockIdx.x] = __float_as_uint(fmaxf(fmaxf(smax[0], smax[1]), fmaxf(smax[2], smax[3])));
}

__global__ __launch_bounds__(256) void split_rows_kernel(const float* __restrict__ W, int N, int K, _Float16* __restrict__ hl,
                                                         int* __restrict__ wexp) {
    const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= N) return;
    const float* w = W + (long)row * K;
    float v[8];
    float m = 0.f;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int k = lane + 64 * u;
        v[u] = k < K ? w[k] : 0.f;
        m = fmaxf(m, fabsf(v[u]));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    const int e = kTarget - frexp_exp(__float_as_uint(m));
    const float sc = pow2f(e);
    _Float16* out = hl + (long)row * 2 * K;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int k = lane + 64 * u;
        if (k < K) {
            const float xs = v[u] * sc;
            const _Float16 hh = (_Float16)xs;
            out[(k >> 5) * 64 + (k & 31)] = hh;
            out[(k >> 5) * 64 + 32 + (k & 31)] = (_Float16)(xs - (float)hh);
        }
    }
    if (lane == 0) wexp[row] = e;
}

bool g_enabled = [] {
    const char* e = getenv("HP_CONV_SPLIT");
    return !(e && e[0] == '0');
}();

}  // namespace

long hp_conv_split_area_floats(long R) { return HP_CS_AMAX_OFF + 20 * hp_conv_split_tiles_pad(R) + 4; }   // amax x 4, P-format exponents 4 x 4, format word
bool hp_conv_split_enabled() { return g_enabled; }
HP_API int hp_conv_split_set(int on) {
    const int was = g_enabled;
    g_enabled = on != 0;
    return was;
}

int hp_conv_split_prep(int n, const float* const* W0, const float* const* W1, float* area0, long sArea, long R, int fmt, hipStream_t stream) {
    PrepParams p{};
    for (int l = 0; l < 4; ++l) {
        p.W[l] = W0[l];
        p.sWz[l] = n > 1 ? (long)(W1[l] - W0[l]) : 0;
    }
    p.amax = reinterpret_cast<unsigned*>(area0 + HP_CS_AMAX_OFF);
    p.n_amax = 4 * hp_conv_split_tiles_pad(R);
    p.wexp = reinterpret_cast<int*>(area0 + HP_CS_WEXP_OFF);
    p.hl = reinterpret_cast<_Float16*>(area0 + HP_CS_HI_OFF);
    p.sArea = sArea;
    p.fmt = reinterpret_cast<int*>(area0 + hp_conv_pp_fmt_offset(R));
    p.fmt_value = fmt;
    hipLaunchKernelGGL(conv_split_prep_kernel, dim3(kRows[4] / 4, n), dim3(256), 0, stream, p);
    HP_RETURN_LAST_ERROR();
}

int hp_conv_split_layer1(int n, const float* x, long sXz, const float* W, long sWz, const float* b, long sBz, float* h1, long sHz,
                         float* area0, long sArea, long R, hipStream_t stream) {
    const long blocks = (R + 127) / 128;
    hipLaunchKernelGGL(conv1_kernel, dim3((unsigned)(blocks < 2048 ? blocks : 2048), n), dim3(256), 0, stream, x, sXz, W, sWz, b,
                       sBz, h1, sHz, reinterpret_cast<unsigned*>(area0 + HP_CS_AMAX_OFF), sArea, R);
    HP_RETURN_LAST_ERROR();
}

// layer l = 2..5: C = act(X W_l^T + b_l).  colmax: no store, per-128-row-tile column maxima into cmax / cidx.
int hp_conv_split_layer(int l, int n, const float* X, long sXz, const float* bias, long sBiasz, float* C, long sCz, float* area0,
                        long sArea, long M, int relu, int colmax, float* cmax, int* cidx, int group_rows, hipStream_t stream) {
    if (l < 2 || l > 5 || M <= 0) return -1;
    CsParams p{};
    p.X = X; p.sXz = sXz;
    p.Whl = reinterpret_cast<const _Float16*>(area0 + HP_CS_HI_OFF) + 2 * kWOff[l - 2];
    p.wexp = reinterpret_cast<const int*>(area0 + HP_CS_WEXP_OFF) + kRows[l - 2];
    const long tp = hp_conv_split_tiles_pad(M);
    p.amax_in = reinterpret_cast<const unsigned*>(area0 + HP_CS_AMAX_OFF) + (l - 2) * tp;
    p.amax_out = (relu && !colmax) ? reinterpret_cast<unsigned*>(area0 + HP_CS_AMAX_OFF) + (l - 1) * tp : nullptr;
    p.sArea = sArea;
    p.bias = bias; p.sBiasz = sBiasz;
    p.C = C; p.sCz = sCz;
    p.cmax = cmax; p.cidx = cidx;
    p.M = (int)M; p.K = kK[l - 2]; p.N = kRows[l - 1] - kRows[l - 2];
    p.relu = relu; p.group_rows = group_rows;
    p.tiles_n = p.N / BN;
    const unsigned tiles = (unsigned)((M + BM - 1) / BM) * p.tiles_n;
    if (colmax)
        hipLaunchKernelGGL(conv_split_kernel<true>, dim3(tiles, n), dim3(256), 0, stream, p);
    else
        hipLaunchKernelGGL(conv_split_kernel<false>, dim3(tiles, n), dim3(256), 0, stream, p);
    HP_RETURN_LAST_ERROR();
}

// ---- the split-f16 GEMM as a stand-alone primitive (bench.py's roofline leg, tests): C = act(X W^T + b), X (M, K), W (N, K)
// fp32 of either sign; N % 128 == 0, K % 32 == 0, K <= 512.  ws: hp_gemm_f16x2_workspace_floats(M, N, K) floats.
// prepare: max|X| per 128-row tile and the split of W (what the producing layer's epilogue and conv_split_prep_kernel do inside the stack);
// run: the conv_split_kernel launch alone.
HP_API long hp_gemm_f16x2_workspace_floats(long M, int N, int K) { return hp_conv_split_tiles_pad(M) + N + (long)N * K; }

HP_API int hp_gemm_f16x2_prepare(long M, int N, int K, const float* X, const float* W, float* ws, hipStream_t stream) {
    HP_CHECK_ARG(M > 0 && N > 0 && K > 0 && N % BN == 0 && K % BK == 0 && K <= 512 && X && W && ws);
    const long tp = hp_conv_split_tiles_pad(M);
    hipLaunchKernelGGL(absmax_kernel, dim3((unsigned)((M + 127) / 128)), dim3(256), 0, stream, X, M, K, reinterpret_cast<unsigned*>(ws));
    hipLaunchKernelGGL(split_rows_kernel, dim3((N + 3) / 4), dim3(256), 0, stream, W, N, K, reinterpret_cast<_Float16*>(ws + tp + N),
                       reinterpret_cast<int*>(ws + tp));
    HP_RETURN_LAST_ERROR();
}

// W (N, K) fp32 -> P-format rows + per-row exponents (conv_pp.hip's stand-alone primitive shares this launch)
int hp_split_rows_launch(const float* W, int N, int K, float* hl, float* wexp, hipStream_t stream) {
    hipLaunchKernelGGL(split_rows_kernel, dim3((N + 3) / 4), dim3(256), 0, stream, W, N, K, reinterpret_cast<_Float16*>(hl),
                       reinterpret_cast<int*>(wexp));
    HP_RETURN_LAST_ERROR();
}

HP_API int hp_gemm_f16x2_run(long M, int N, int K, const float* X, const float* bias, float* C, int relu, const float* ws,
                             hipStream_t stream) {
    HP_CHECK_ARG(M > 0 && M < (1L << 31) && N > 0 && K > 0 && N % BN == 0 && K % BK == 0 && K <= 512 && X && bias && C && ws);
    const long tp = hp_conv_split_tiles_pad(M);
    CsParams p{};
    p.X = X;
    p.Whl = reinterpret_cast<const _Float16*>(ws + tp + N);
    p.wexp = reinterpret_cast<const int*>(ws + tp);
    p.amax_in = reinterpret_cast<const unsigned*>(ws);
    p.bias = bias;
    p.C = C;
    p.M = (int)M; p.N = N; p.K = K; p.relu = relu;
    p.tiles_n = N / BN;
    hipLaunchKernelGGL(conv_split_kernel<false>, dim3((unsigned)((M + BM - 1) / BM) * p.tiles_n, 1), dim3(256), 0, stream, p);
    HP_RETURN_LAST_ERROR();
}
