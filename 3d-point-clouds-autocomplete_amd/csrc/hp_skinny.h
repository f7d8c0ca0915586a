/* Skinny-M layer programs (internal): the M = B <= 64 chains of the step — hypernetwork trunk (model/hyper_network.py:16-30),
 * encoder fc/mu/std tail (model/encoder.py:30-36) — one launch per phase (layer), ordered by the kernel boundary.  See
 * skinny.hip. */
#pragma once
#include <hip/hip_runtime.h>

#define HP_SK_MAX_OPS 16

enum { HP_SK_F = 1, HP_SK_X = 2, HP_SK_W = 3, HP_SK_FIN = 4 };

/* A (rows <= 64) x cols row-major matrix handed over as S partial slabs; the reader FINISHES it while loading:
 *   v(r,c) = sum_s p[s*slab + r*ld + c]  (+ bias[c])  (max(.,0) if relu)  (0 where mask(r,c) <= 0)
 * mat != NULL: the designated reader of a block also stores the finished values to mat (saved activations). */
typedef struct HpSkSrc {
    const float* p;
    long slab;
    const float* bias;
    const float* mask;
    float* mat;
    int S, ld, ldm, ldmat, relu;
} HpSkSrc;

/* One layer-level operation.  M = clouds (<= 64).
 *   F   out(M x N) = A(M x K) . W(N x K)^T     W rows K-contiguous       tasks: (N/32 strips) x (K/CL ranges)
 *   X   out(M x K) = A(M x N) . W(N x K)       W rows K-contiguous       tasks: (K/32 units)  x (N/CL ranges)
 *   W   out(N x K) = A(M x N)^T . B(M x K)     both materialised         tasks: (N/32)(K/32)/4 (one 32x32 tile per wave)
 *   FIN out(M x N) = finish(A)                                           tasks: M*N/1024
 * F / X write range r of the contraction to slab r of `out` (out + r*out_slab) as raw partial sums; with a single
 * range they apply out_bias / out_relu themselves.  W: rsum(N) = column sums of A (the bias gradient). */
typedef struct HpSkOp {
    int type, phase;
    HpSkSrc a;
    const float* w;   /* F, X: weights (N x K), ld = w_ld.   W: the B operand (M x K), ld = w_ld */
    float* out;
    long out_slab;
    const float* out_bias;
    float* rsum;
    int w_ld, out_ld, out_relu;
    int M, N, K, CL;
    int ntasks;       /* filled by hp_skinny_run */
} HpSkOp;

typedef struct HpSkProgram {
    int nops;
    HpSkOp op[HP_SK_MAX_OPS];   /* phases ascending; each phase is one launch (up to 8 ops), the kernel boundary orders them */
} HpSkProgram;

#ifdef __cplusplus
/* Validates shapes (M <= 64, K/N multiples of 32, CL a power of two in [32,256] dividing the contraction, 16-byte
 * aligned K-contiguous operands) — returns -2 when the program cannot be served (callers fall back to the tiled GEMMs),
 * otherwise launches and returns the hipError_t. */
int hp_skinny_run(HpSkProgram* prog, hipStream_t stream);
bool hp_skinny_enabled();
#endif
