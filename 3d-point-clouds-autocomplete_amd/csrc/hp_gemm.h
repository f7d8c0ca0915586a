/* GEMM descriptor shared by the C ABI (include/hyperpocket_hip.h repeats it for C callers). */
#pragma once

#define HP_GEMM_BIAS 1 /* + bias[j]                                   */
#define HP_GEMM_RELU 2 /* max(., 0)                                   */
#define HP_GEMM_MASK 4 /* * (mask(i,j) > 0)  — ReLU backward, fused   */
#define HP_GEMM_ADD 8  /* + add(i,j)  (before ReLU / mask)             */
#define HP_GEMM_ROWSUM 32 /* also rsum(i) = sum_k A(i,k): the bias gradient rides on the dW = dY^T X contraction */
#define HP_GEMM_COLMAX 16 /* do not store C: per row-tile column max (+bias) and its row -> cmax/cidx (fused max-pool) */

typedef struct HpGemmDesc {
    const float* A;    /* A(i,k) at A + z*sAz + i*sAi + k*sAk (one of sAi,sAk is 1) */
    const float* B;    /* B(k,j) at B + z*sBz + k*sBk + j*sBj (one of sBk,sBj is 1) */
    float* C;          /* C(i,j) at C + z*sCz + i*ldc + j                           */
    const float* bias; /* bias(j) at bias + z*sBiasz + j                            */
    const float* mask; /* mask(i,j) at mask + z*sMaskz + i*ldmask + j               */
    const float* add;  /* add(i,j) at add + z*sAddz + i*ldadd + j                   */
    float* ws;         /* split-K slabs: hp_gemm_workspace_floats(desc) floats       */
    long sAz, sBz, sCz, sBiasz, sMaskz, sAddz;
    long sAi, sAk, sBk, sBj;
    int ldc, ldmask, ldadd;
    int M, N, K, batch;
    int ksplit; /* <=1: no split */
    int flags;
    /* HP_GEMM_COLMAX: rows come in groups of group_rows (one cloud); cmax/cidx are (M / tile_rows, N) with
     * tile_rows = hp_gemm_tile_rows(desc) dividing group_rows; cidx holds the row index inside its group; with batch > 1 the
     * arrays of batch z start z*sCz elements further on */
    float* cmax;
    int* cidx;
    int group_rows;
    /* HP_GEMM_ROWSUM: rsum(i) at rsum + z*sRsumz + i ; with split-K the workspace needs batch*ksplit*M more floats */
    float* rsum;
    long sRsumz;
    /* A size known only on the device: dyn_count (device pointer to one int, or NULL) bounds M (dyn_kind 1: rows of
     * A/C beyond it are neither read nor written) or K (dyn_kind 2: the contraction stops there; split-K ranges
     * partition the real K).  M / K of this descriptor stay the static upper bounds the launch is sized for. */
    const int* dyn_count;
    int dyn_kind;
} HpGemmDesc;
