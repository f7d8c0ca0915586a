/* Fused encoder backward (internal): the conv stack of one or two PointNet encoders on their distinct critical points
 * in three launches — see enc_bwd.hip.  Autograd of /root/reference/model/encoder.py:14-28,43-45. */
#pragma once
#include <hip/hip_runtime.h>

/* Critical-point compaction of one encoder (per cloud: the 512 arg-max channels sorted by point; channels that peak at
 * the same point share one row below the max-pool). */
struct HpCrit {
    int* chan;     /* (B, 512) channels sorted by (point, channel) */
    int* start;    /* (B, 513) start[u] = first sorted position of slot u, start[U] = 512 */
    int* pt;       /* (B, 512) point of slot u */
    int* slot;     /* (B, 512) slot of channel c */
    int* eslot;    /* (B, 512) slot of sorted position t (fused path) */
    int* cnt;      /* (B)      U = number of distinct critical points */
    int* off;      /* (B)      first compact row of the cloud (old path) */
    int* total;    /* (1)      sum of cnt (old path) */
};

/* partial-sum layout of one row range: [dW4 | dW3 | dW2 | dW1 | db4 | db3 | db2 | db1] */
#define HP_EB_PART_FLOATS (512 * 256 + 256 * 128 + 128 * 64 + 64 * 3 + 512 + 256 + 128 + 64)
#define HP_EB_MAX_SPLITS 64
/* enc_bwd_f16.hip: W4, W3, W2 as the chain's A-fragment stream (42 chunks of 16 KB) + the 2^-e table of their columns */
#define HP_EB_WT_BYTES (42L * 16384)
#define HP_EB_WT_US_FLOATS (256 + 128 + 64)

struct HpEncBwdSide {
    /* inputs */
    const float* x;          /* (B, Np, 3) */
    const int* argidx;       /* (B, 512) */
    const float* dg;         /* (B, 512)  d/d (max-pooled features), from the tail's backward */
    const float* W[5];       /* conv weights, W[l-1] = layer l: (kEnc[l], kEnc[l-1]) row-major */
    const float* h[5];       /* h[l] = the forward's per-point activations of layer l (l = 1..4), (B*Np, kEnc[l]): fp32, or — when
                                *fmt == HP_PP_FMT_P — P-format lines [hi 32 | lo 32] f16 with block exponents pexp[l] (conv_pp.hip) */
    const int* fmt;          /* the forward workspace's format word */
    const int* pexp[5];      /* pexp[l][(row >> 7) * pncb[l] + block]: pncb[l] column blocks of kEnc[l] / pncb[l] channels per 128-row tile */
    int pncb[5], pcbs[5];    /* blocks per tile, log2 of a block's channels */
    /* VAE head (prep kernel): d mu = gz + gmu ; d lv = (gz*eps + gexplv) * exp(lv) */
    const float *eps, *lv, *gout, *gmu, *gexplv;
    float *dmu, *dlv;
    int is_vae, gout_ld;
    HpCrit crit;
    /* rows (b*512 + u), u < roundup32(cnt[b]): delta_l, and the rows' activations below (hc[l] = h_l of the row, hc[0] = x) */
    float* d[5];             /* d[l], l = 1..4 */
    float* hc[4];            /* hc[0] = xc (3 per row), hc[1..3] */
    float* part;             /* S * HP_EB_PART_FLOATS partial sums */
    float* d4max;            /* max |delta4[row][:]| of every row the gather launch writes (the f16 chain's row scale) */
    float* hmax;             /* 4 per row: max of h3 / h2 / h1 of the row (gather launch), pad */
    int* bexp;               /* 8 per 32-row block: scale exponents of delta4..delta1, h3..h1 (f16 chain launch, for the f16 dW launch) */
    unsigned char* hmask;    /* 64 bytes per row: bit c of bytes [0,32) = (h3[row][c] > 0), [32,48) h2, [48,56) h1 (gather launch) */
    unsigned char* wt;       /* HP_EB_WT_BYTES: the f16 chain's weight stream (prep launch) */
    float* wt_us;            /* HP_EB_WT_US_FLOATS */
    float* gW[5];            /* d conv_w[l-1], l = 1..5 */
    float* gb[5];
};

struct HpEncBwdArgs {
    HpEncBwdSide e[2];
    int n, B, Np, out, S;
    long long* prof; /* HP_EB_PROF: per-workgroup phase timestamps (debug) */
};

#ifdef __cplusplus
int hp_enc_bwd_prep(const HpEncBwdArgs* a, hipStream_t stream);      /* sort + VAE head */
int hp_enc_bwd_conv(const HpEncBwdArgs* a, hipStream_t stream);      /* gather + chain + dW + reduce */
int hp_enc_bwd_dw_f16(const HpEncBwdArgs* a, hipStream_t stream);    /* enc_bwd_f16.hip: dW4..dW1, db4..db1 partial sums on the f16 pipe */
int hp_enc_bwd_chain_f16(const HpEncBwdArgs* a, hipStream_t stream); /* enc_bwd_f16.hip: weight stream + delta chain on the f16 pipe */
bool hp_enc_bwd_chain_f16_enabled();                                  /* HP_EB_CHAIN16 (default on) / hp_enc_bwd_chain_f16_set */
int hp_enc_bwd_chain_f16_set(int on);
int hp_enc_bwd_max_clouds();                                          /* largest B the dW launch's LDS table serves */
#endif
