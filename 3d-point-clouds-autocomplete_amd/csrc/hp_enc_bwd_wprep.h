/* Device side of the f16 delta chain's weight stream (enc_bwd_f16.hip reads it, enc_bwd.hip's prep launch writes it as extra
 * workgroups): W_4, W_3, W_2 of an encoder as A fragments in consumption order.  Internal. */
#pragma once
#include "hp_enc_bwd.h"

namespace hp_wprep {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int kChunk = 16384;                 /* weight bytes per chunk of the stream */
constexpr int kC4 = 32, kC3 = 8, kC2 = 2;     /* chunks of layer 4 (1 k-step each), 3 (2 k-steps), 2 (4 k-steps) */
constexpr long kOff3 = (long)kC4 * kChunk, kOff2 = kOff3 + (long)kC3 * kChunk;
static_assert((kC4 + kC3 + kC2) * (long)kChunk == HP_EB_WT_BYTES, "weight stream");
constexpr int kUs4 = 0, kUs3 = 256, kUs2 = 384;   /* unscale table: 2^-e of the weight columns */
static_assert(kUs2 + 64 == HP_EB_WT_US_FLOATS, "unscale table");
constexpr int kTasks = 14;                    /* tiles of 32 input channels: 8 of W4, 4 of W3, 2 of W2 */

/* e with m 2^e in [2^13, 2^14) (m = 0: 14), clamped so that 2^e and 2^-e are normal floats */
__device__ __forceinline__ int scale_exp(float m) {
    const int E = (int)((__float_as_uint(m) >> 23) & 0xff);
    return max(-100, min(100, 14 - (E ? E - 126 : 0)));
}
__device__ __forceinline__ float pow2f(int e) { return __uint_as_float((unsigned)(e + 127) << 23); }
__device__ __forceinline__ unsigned pack2(_Float16 a, _Float16 b) { return __builtin_bit_cast(unsigned, f16x2{a, b}); }
/* lo pieces of two values whose hi pieces are packed in hpk: f16(x - f32(hi)), one v_fma_mix each (conv_pp.hip) */
__device__ __forceinline__ unsigned lo_pair(float x0, float x1, unsigned hpk) {
    unsigned d;
    asm("v_fma_mixlo_f16 %0, %1, 1.0, -%3 op_sel:[0,0,0] op_sel_hi:[0,0,1]\n\t"
        "v_fma_mixhi_f16 %0, %2, 1.0, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]"
        : "=&v"(d)
        : "v"(x0), "v"(x1), "v"(hpk));
    return d;
}
/* the hi and lo fragments of 8 values under the scale sc (a power of two): hi = f16(x sc), lo = f16(x sc - hi), one v_fma_mix per
 * piece and value */
__device__ __forceinline__ void split8s(const float (&x)[8], float sc, f16x8& hi, f16x8& lo) {
    u32x4 H, L;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        unsigned hh, ll;
        asm("v_fma_mixlo_f16 %0, %1, %3, 0 op_sel_hi:[0,0,0]\n\t"
            "v_fma_mixhi_f16 %0, %2, %3, 0 op_sel_hi:[0,0,0]"
            : "=&v"(hh)
            : "v"(x[2 * i]), "v"(x[2 * i + 1]), "v"(sc));
        asm("v_fma_mixlo_f16 %0, %1, %3, -%4 op_sel:[0,0,0] op_sel_hi:[0,0,1]\n\t"
            "v_fma_mixhi_f16 %0, %2, %3, -%4 op_sel:[0,0,1] op_sel_hi:[0,0,1]"
            : "=&v"(ll)
            : "v"(x[2 * i]), "v"(x[2 * i + 1]), "v"(sc), "v"(hh));
        H[i] = hh;
        L[i] = ll;
    }
    hi = __builtin_bit_cast(f16x8, H);
    lo = __builtin_bit_cast(f16x8, L);
}
/* 8 scaled values -> the hi and the lo fragment of a lane */
__device__ __forceinline__ void split8(const float (&y)[8], f16x8& hi, f16x8& lo) {
    u32x4 H, L;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        H[i] = pack2((_Float16)y[2 * i], (_Float16)y[2 * i + 1]);
        L[i] = lo_pair(y[2 * i], y[2 * i + 1], H[i]);
    }
    hi = __builtin_bit_cast(f16x8, H);
    lo = __builtin_bit_cast(f16x8, L);
}

/* Task `id` of kTasks, run by a workgroup of 512 threads.  W_l is (K = kEnc[l] outputs) x (N = kEnc[l-1] inputs), row-major; the
 * chain contracts over K.  A task = one tile of 32 input channels n: the (K x 32) tile in ONE round of loads (32 per thread, all in
 * flight), column maxima over K -> exponent e_n (unscale 2^-e_n into the table), the scaled tile through LDS, then per k-step s
 * and lane (m, h) the 8 values W[16 s + 4 h + (j & 3) + 8 (j >> 2)][32 tn + m] 2^e_n, j = 0..7, as a 16-byte hi and a 16-byte
 * lo piece at [(s T + tn) 2 + piece][lane] — the order enc_bwd_chain_f16_kernel consumes them in. */
__device__ __forceinline__ void task(const HpEncBwdSide& s, int id, int tid) {
    __shared__ float tile[512 * 32];
    __shared__ float smax[16][32];
    __shared__ float sscale[32];
    int l, tn, T, K, N, us0;
    long base;
    if (id < 8) { l = 4; tn = id; T = 8; K = 512; N = 256; base = 0; us0 = kUs4; }
    else if (id < 12) { l = 3; tn = id - 8; T = 4; K = 256; N = 128; base = kOff3; us0 = kUs3; }
    else { l = 2; tn = id - 12; T = 2; K = 128; N = 64; base = kOff2; us0 = kUs2; }
    const float* W = s.W[l - 1] + 32 * tn;
    const int m = tid & 31, sl = tid >> 5;
    float v[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) v[i] = W[(long)min(sl + 16 * i, K - 1) * N + m];      /* (branch-free: past K the last row again) */
    float mx = 0.f;
#pragma unroll
    for (int i = 0; i < 32; ++i) mx = fmaxf(mx, fabsf(v[i]));
    smax[sl][m] = mx;
    __syncthreads();
    if (tid < 32) {
        float q = smax[0][tid];
#pragma unroll
        for (int k = 1; k < 16; ++k) q = fmaxf(q, smax[k][tid]);
        const int e = scale_exp(q);
        sscale[tid] = pow2f(e);
        s.wt_us[us0 + 32 * tn + tid] = pow2f(-e);
    }
    __syncthreads();
    const float sc = sscale[m];
#pragma unroll
    for (int i = 0; i < 32; ++i)
        if (sl + 16 * i < K) tile[(sl + 16 * i) * 32 + m] = v[i] * sc;
    __syncthreads();
    const int lane = tid & 63, fm = lane & 31, h = lane >> 5;
    unsigned char* out = s.wt + base;
    for (int st = tid >> 6; st < K / 16; st += 8) {
        float y[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) y[j] = tile[(16 * st + 4 * h + (j & 3) + 8 * (j >> 2)) * 32 + fm];
        f16x8 hi, lo;
        split8(y, hi, lo);
        unsigned char* dst = out + (long)((st * T + tn) * 2) * 1024 + lane * 16;
        *reinterpret_cast<f16x8*>(dst) = hi;
        *reinterpret_cast<f16x8*>(dst + 1024) = lo;
    }
}

}  // namespace hp_wprep
