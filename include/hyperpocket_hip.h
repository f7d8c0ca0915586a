/*
 * hyperpocket_hip.h — C ABI of libhyperpocket_hip.so (gfx950 / MI355X only).
 *
 * The drop-in boundary for the HyperPocket training-step hot path of
 * gmum/3d-point-clouds-autocomplete.  Plain pointers and sizes; every pointer is a DEVICE
 * pointer unless stated; `stream` is a hipStream_t (NULL = the null stream).  All entry points
 *   - are asynchronous on `stream`, never allocate, free or synchronise (safe under hipGraph
 *     capture; the caller owns every buffer, as the reference binding does with torch::empty —
 *     structural_loss.cpp:32-33,49,64-65,90-93,111-112),
 *   - fully initialise their outputs (callers pass uninitialised memory),
 *   - return 0 on success, -1 on an invalid argument, otherwise the hipError_t of the launch
 *     (the reference throws std::runtime_error("CUDA kernel failed : <code>"),
 *      approxmatch.cu:334-337; its nndistance launchers check nothing, nndistance.cu:131-160).
 * Layouts are the reference's: point sets (b, n, 3) fp32 contiguous, indices int32.
 *
 * TEST HOOKS.  hp_emd_set_rows_per_lane, hp_emd_set_final_derive, hp_emd_set_chains, hp_emd_set_cull, hp_encoder_backward_set_fused, hp_encoder_backward_set_chain_f16, hp_hypernet_set_heads_stream, hp_conv_split_set, hp_skinny_set_enabled,
 * hp_target_fused_set_f16 (and hp_conv_presplit_set below) flip PROCESS-WIDE switches that select between implementations of
 * the same result; they exist so that the parity tests can hold every implementation against the oracle in one process.  They
 * are plain globals: not thread-safe, not per-stream, not meant to be called while another host thread is inside the library.
 * A production caller never needs them (the defaults are the measured-fastest paths; the environment variables named at each
 * hook set the same switch once at load time).
 */
#ifndef HYPERPOCKET_HIP_H
#define HYPERPOCKET_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ihipStream_t* hpStream_t; /* == hipStream_t */

/* ------------------------------------------------------------------------------------------
 * Structural losses — the five launchers the reference's pybind module binds
 * (utils/pytorch_structural_losses/structural_loss.cpp:11-15).
 * ------------------------------------------------------------------------------------------ */

/* nndistance  (structural_loss.cpp:14, nndistance.cu:131-134)
 * result[i,j]  = min_k |xyz[i,j]-xyz2[i,k]|^2, result_i = arg min (smallest k on ties); and the
 * mirrored result2/result2_i over xyz2's points. */
int hp_nndistance(int b, int n, const float* xyz, int m, const float* xyz2, float* result, int* result_i,
                  float* result2, int* result2_i, hpStream_t stream);

/* nndistancegrad  (structural_loss.cpp:15, nndistance.cu:155-160) */
int hp_nndistancegrad(int b, int n, const float* xyz1, int m, const float* xyz2, const float* grad_dist1,
                      const int* idx1, const float* grad_dist2, const int* idx2, float* grad_xyz1, float* grad_xyz2,
                      hpStream_t stream);

/* approxmatch  (structural_loss.cpp:11, approxmatch.cu:330-338) — the reference's exact argument list.
 * match (b, m, n), temp (b, 2*(n+m)) as in the reference; no other buffer.  Keeps the reference's data flow (the
 * four vectors of temp are the only state, match is read-modify-written once per level) on a chip-wide grid. */
int hp_approxmatch(int b, int n, int m, const float* xyz1, const float* xyz2, float* match, float* temp,
                   hpStream_t stream);
/* The same result ~2.5x faster for a caller that can allocate: `ws` = scratch of
 * hp_approxmatch_workspace_floats(b,n,m) floats (packed candidate records incl. the per-level scaling vectors; lets
 * `match` be written once instead of read-modify-written nine times).  What this repo's Python binding calls. */
long hp_approxmatch_workspace_floats(int b, int n, int m);
int hp_approxmatch_ws(int b, int n, int m, const float* xyz1, const float* xyz2, float* match, float* temp, float* ws,
                      hpStream_t stream);
/* Tuning hook (no counterpart in the reference): rows per lane of the packed-record sweeps — rows1 (phase 3 + phase 1
 * kernel) and rows2 (phase 2) in {0,1,2,4}, grad2 (final cost/gradient sweep) in {0,1,2}; 0 = the size heuristic.
 * Process-wide.  Every setting evaluates each row with the same operations in the same order (results identical bit
 * for bit; tests/test_structural_losses_gpu.py). */
/* [test hook: process-wide, not thread-safe — see the header comment] */
int hp_emd_set_rows_per_lane(int rows1, int rows2, int grad2);
/* The match-free cost / gradient sweep (hp_emd_forward*, hp_emd_backward) evaluates the nine per-level exponentials of a point
 * pair; the levels are exact powers of 4 apart, so four of them can be formed as the fourth power of their neighbour's (two
 * multiplies instead of v_exp_f32; ~5 ulp instead of 1 on a value nothing is downstream of).  1 (default; environment
 * HP_EMD_FINAL_DERIVE=0 turns it off at load time): derived; 0: all nine from the hardware exponential.  Returns the previous
 * setting.  The level sweeps and the `match` tensor hp_approxmatch / hp_approxmatch_ws return are never derived. */
/* [test hook: process-wide, not thread-safe — see the header comment] */
int hp_emd_set_final_derive(int on);
/* hp_emd_forward / hp_emd_forward_acc run the clouds as TWO chains of launches — the first half of the batch on the caller's
 * stream, the second half on a stream the library owns (one per device, created on first use), ordered behind everything the
 * caller's stream held at the call and joined back into it before the call returns control of the stream: to the caller it is one
 * asynchronous call on `stream`, as before.  While one chain's launch ramps up or drains, the other's waves hold the vector pipes
 * (B = 64, N = 2048: 1.37 -> 1.28 ms).  Used when each half still fills the chip and `stream` is not being captured; per cloud the
 * results are those of one chain (gradients identical, cost within the 2e-6 of the partial sums' grouping).  2 (default;
 * environment HP_EMD_CHAINS=1 at load time): two chains; 1: one.  Returns the previous setting. */
/* [test hook: process-wide, not thread-safe — see the header comment] */
int hp_emd_set_chains(int chains);
/* hp_emd_forward / hp_emd_forward_acc put both point sets in a k-d order first (one workgroup per cloud and set; 2^k-aligned runs
 * of positions are boxes of 2^k points) and the sweeps of the first `levels` annealing levels skip every (64-row tile, 8-candidate
 * block) unit whose bounding boxes are further apart than the level's underflow radius (d^2 > 152 ln2 / |level|: each of its
 * exponentials is exactly +0 in fp32, so each skipped term of approxmatch.cu:86-87,131-132,185-189 is an exact zero).  Results:
 * the sums of the caller's order with their zero terms left out, accumulated in the k-d order (cost within 3e-7 of the same
 * kernels on the caller's order); gradients are written through the permutation, so callers keep their own point order.
 * levels in 0..9; 0 = the caller's order, every unit evaluated (rounds 1-5).  Default 3 (environment HP_EMD_CULL at load time).
 * Sets of more than 4096 points always run in the caller's order.  hp_approxmatch / hp_approxmatch_ws (whose `match` and `temp`
 * are returned in the caller's order) are never re-ordered.  Returns the previous setting. */
/* [test hook: process-wide, not thread-safe — see the header comment] */
int hp_emd_set_cull(int levels);

/* Match-free EMD (what match_cost.py:9-46 computes through ApproxMatch + MatchCost + MatchCostGrad, without ever
 * writing the (b,m,n) match tensor): cost (b,) plus whichever of grad1 = d cost/d xyz1, grad2 = d cost/d xyz2 the
 * caller asks for (the cost rides on one of those sweeps); hp_emd_backward computes grad2 later from the packed
 * records hp_emd_forward left in `ws` (hp_approxmatch_workspace_floats).
 * partials: hp_emd_partials_floats floats.  temp: scratch of hp_approxmatch's size (its remainL block is left one level
 * short: the last level's phase 3 has no reader on this path and is not launched); ws as in hp_approxmatch_ws. */
long hp_emd_partials_floats(int b, int n, int m);
int hp_emd_forward(int b, int n, int m, const float* xyz1, const float* xyz2, float* temp, float* ws, float* partials,
                   float* cost, float* grad1 /* or NULL */, float* grad2 /* or NULL */, hpStream_t stream);
/* The training step's form of hp_emd_forward: grad2_acc (b,m,3) already holds the gradient of the other loss terms with respect
 * to xyz2 (written on stream `after`; NULL = the same stream) and receives += scale * d cost / d xyz2 from the gradient sweep
 * itself (no separate axpy launch); the sweep is ordered behind everything enqueued on `after` so far.  scale != 0. */
int hp_emd_forward_acc(int b, int n, int m, const float* xyz1, const float* xyz2, float* temp, float* ws, float* partials,
                       float* cost, float* grad2_acc, float scale, hpStream_t stream, hpStream_t after);
int hp_emd_backward(int b, int n, int m, const float* xyz1, const float* xyz2, const float* ws, float* grad2,
                    hpStream_t stream);

/* matchcost  (structural_loss.cpp:12, approxmatch.cu:340-347) — the reference's exact argument list: one
 * 1024-thread workgroup per cloud, ordered sums, no scratch. */
int hp_matchcost(int b, int n, int m, const float* xyz1, const float* xyz2, const float* match, float* out,
                 hpStream_t stream);
/* The same sum chip-wide in two ordered stages for a caller that can allocate `partials`
 * (hp_matchcost_workspace_floats floats): 5.9 TB/s at B=64, N=2048 instead of one workgroup per cloud. */
long hp_matchcost_workspace_floats(int b, int n, int m);
int hp_matchcost_ws(int b, int n, int m, const float* xyz1, const float* xyz2, const float* match, float* out,
                    float* partials, hpStream_t stream);

/* matchcostgrad  (structural_loss.cpp:13, approxmatch.cu:349-357) */
int hp_matchcostgrad(int b, int n, int m, const float* xyz1, const float* xyz2, const float* match, float* grad1,
                     float* grad2, hpStream_t stream);

/* ------------------------------------------------------------------------------------------
 * Fused Chamfer loss — replaces losses/champfer_loss.py:11-35 (ChamferLoss.forward and its
 * autograd backward) without materialising the (b, n, m) distance tensor.
 * ------------------------------------------------------------------------------------------ */
long hp_chamfer_workspace_floats(int b, int n, int m);
int hp_chamfer_forward(int b, int n, const float* preds, int m, const float* gts, float* dist1, int* idx1,
                       float* dist2, int* idx2, float* partials, float* loss /* 1 float */, hpStream_t stream);
int hp_chamfer_backward(int b, int n, const float* preds, int m, const float* gts, const int* idx1, const int* idx2,
                        const float* grad_loss /* device scalar */, float* grad_preds, float* grad_gts,
                        hpStream_t stream);

/* ------------------------------------------------------------------------------------------
 * fp32 matrix-core GEMM family (v_mfma_f32_32x32x2_f32) — the dense contractions PyTorch/cuBLAS
 * perform for the reference's nn.Conv1d(k=1)/nn.Linear/torch.mm calls (model/encoder.py:14-36,
 * model/hyper_network.py:16-43, model/target_network.py:31-38) and their autograd backward.
 *   C[z](i,j) = epi( sum_k A[z](i,k) B[z](k,j) );  epi = (+bias[j]) (+add(i,j)) (ReLU) (*(mask(i,j)>0))
 * ------------------------------------------------------------------------------------------ */
#define HP_GEMM_BIAS 1
#define HP_GEMM_RELU 2
#define HP_GEMM_MASK 4
#define HP_GEMM_ADD 8
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
    int ksplit; /* <=1: no split; >1: ordered (atomic-free) split-K through `ws` */
    int flags;
    /* HP_GEMM_COLMAX: rows come in groups of group_rows (one cloud); cmax/cidx are (M / tile_rows, N) with
     * tile_rows = hp_gemm_tile_rows(desc) dividing group_rows; cidx holds the row index inside its group */
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

long hp_gemm_workspace_floats(const HpGemmDesc* d);
int hp_gemm_tile_rows(const HpGemmDesc* d);
int hp_gemm_f32(const HpGemmDesc* d /* host struct */, hpStream_t stream);
/* out[z][j] = sum_i (mask ? (mask(i,j)>0 ? X(i,j) : 0) : X(i,j))  — bias gradients */
long hp_colsum_workspace_floats(int batch, int M, int N);
int hp_colsum_f32(int batch, int M, int N, const float* X, long sXz, int ldx, const float* mask, long sMaskz,
                  int ldmask, float* out, long sOz, float* ws /* or NULL */, hpStream_t stream);

/* ------------------------------------------------------------------------------------------
 * Model entry points.  Parameter tables hold device pointers to tensors laid out exactly as the
 * reference's nn.Module parameters (row-major (out,in); Conv1d weights (out,in,1)).
 * ------------------------------------------------------------------------------------------ */
#define HP_MAX_HEADS 8
#define HP_MAX_TN_LAYERS 8

typedef struct HpEncoderWeights { /* model/encoder.py:14-36: conv 3-64-128-256-512-512, fc 512, mu/std (out,512) */
    const float* conv_w[5];
    const float* conv_b[5];
    const float* fc_w;
    const float* fc_b;
    const float* mu_w;
    const float* mu_b;
    const float* std_w; /* NULL for a non-VAE encoder */
    const float* std_b;
} HpEncoderWeights;

typedef struct HpEncoderGrads {
    float* conv_w[5];
    float* conv_b[5];
    float* fc_w;
    float* fc_b;
    float* mu_w;
    float* mu_b;
    float* std_w;
    float* std_b;
} HpEncoderGrads;

typedef struct HpHyperWeights { /* model/hyper_network.py:16-36 */
    const float* trunk_w[5];
    const float* trunk_b[5];
    int n_heads;
    int head_out[HP_MAX_HEADS];
    const float* head_w[HP_MAX_HEADS];
    const float* head_b[HP_MAX_HEADS];
} HpHyperWeights;

typedef struct HpHyperGrads {
    float* trunk_w[5];
    float* trunk_b[5];
    float* head_w[HP_MAX_HEADS];
    float* head_b[HP_MAX_HEADS];
} HpHyperGrads;

/* Encoder.forward (model/encoder.py:43-53).  x (B,Np,3) contiguous.  Outputs: argidx/g (B,512) the max-pool
 * arg-max / value, f (B,512) the fc activation, mu (B,out); VAE also lv (raw std_layer output), z = eps*exp(lv)+mu,
 * explv = exp(lv) (what the reference returns as "logvar").  ws: hp_encoder_forward_workspace_floats. */
long hp_encoder_forward_workspace_floats(int B, int Np);
int hp_encoder_forward(int B, int Np, const float* x, const HpEncoderWeights* w, int out_size, int is_vae,
                       const float* eps, int* argidx, float* g, float* f, float* mu, float* lv, float* z, float* explv,
                       float* ws, hpStream_t stream);
/* The two encoders of a HyperPocket step (model/full_model.py:106-112: random_encoder on `missing`, real_encoder on
 * `existing`; same conv stack, own weights) in one call: every conv layer is one batched launch over both.  io[0], io[1]
 * carry hp_encoder_forward's arguments; same B, Np, out_size.  Results are those of two hp_encoder_forward calls. */
typedef struct HpEncoderIO {
    const float* x;
    const HpEncoderWeights* w;
    const float* eps; /* VAE only */
    int* argidx;
    float *g, *f, *mu, *lv, *z, *explv, *ws;
    int is_vae;
    int out_ld; /* row stride of the primary output (z of a VAE encoder, mu of a plain one); 0 = out_size (dense).  The two
                   encoders of a pair can so write the halves of one (B, 2*out) latent [z | real mu] directly */
} HpEncoderIO;
int hp_encoder_forward_pair(int B, int Np, int out_size, const HpEncoderIO* io /* [2] */, hpStream_t stream);
/* Layout of the forward workspace: per-point activations h1..h4 (B*Np rows of 64 | 128 | 256 | 512 channels, 4 bytes per value), the
 * slot of h5 (per-tile maxima, tail slabs), then the split area (weight pieces, exponents).  Since round 4 the fast path (whole
 * 128-point tiles per cloud) stores h1..h4 as f16 piece pairs with block exponents ("P-format", csrc/conv_pp.hip) — the operand
 * format of the next layer's matrix-core launch — and says so in a word of the split area; hp_encoder_backward* read either
 * format.  hp_encoder_workspace_to_f32 converts such a workspace to plain fp32 rows in place (idempotent). */
int hp_encoder_workspace_to_f32(int B, int Np, float* ws, hpStream_t stream);
/* [test hook: process-wide, not thread-safe — see the header comment] 0: the conv stack takes fp32 activations again (round 3's
 * kernels, csrc/conv_split.hip; also: environment HP_CONV_PRESPLIT=0).  Returns the previous setting. */
int hp_conv_presplit_set(int on);
/* The P-format GEMM as a stand-alone primitive (bench.py's roofline leg, tests): C = act(X W^T + b), X (M,K), W (N,K) fp32, N % 128
 * == 0, K % 32 == 0, K <= 512.  prepare packs X (one exponent per 128 rows x xcb channels) and W into ws
 * (hp_gemm_pp_workspace_floats floats); run is the matrix-core launch alone: mode 0 stores C in P-format inside ws
 * (hp_gemm_pp_unpack -> fp32), mode 1 forms per-128-row-tile column maxima of X W^T + b and their rows (hp_gemm_pp_partials). */
long hp_gemm_pp_workspace_floats(long M, int N, int K);
int hp_gemm_pp_prepare(long M, int N, int K, int xcb, const float* X, const float* W, float* ws, hpStream_t stream);
int hp_gemm_pp_run(long M, int N, int K, int xcb, const float* bias, int relu, int mode, int group_rows, float* ws, hpStream_t stream);
int hp_gemm_pp_unpack(long M, int N, int K, const float* ws, float* C, hpStream_t stream);
int hp_gemm_pp_partials(long M, int N, int K, const float* ws, float* cmax, int* cidx, hpStream_t stream);
/* Gradients of every encoder parameter (autograd of the above).  grad_out = d/dz (VAE) or d/dmu (plain);
 * grad_mu / grad_explv = direct gradients on the VAE outputs (may be NULL).  Only the 512 arg-max points of a
 * cloud carry gradient below the max-pool: their activations are copied out of fwd_ws (the workspace
 * hp_encoder_forward ran in, untouched since) or, with fwd_ws = NULL, recomputed from x.  dedup != 0: channels that
 * peak at the same point share one row (their gradients add), so the layers below run on the DISTINCT critical
 * points, about a third of B*512; same gradients up to fp32 summation order.
 * fwd_ws is declared const because the call never changes the VALUES it holds, but the layered path (dedup == 0, or the fused
 * path switched off) rewrites their REPRESENTATION in place: hidden activations the forward left in the piece format are
 * converted to fp32 rows and the workspace's format word is updated (as hp_encoder_workspace_to_f32 does).  Consequences for a
 * foreign caller: do not run two backward calls over the same workspace concurrently on different streams, and a workspace
 * not produced by hp_encoder_forward* must carry a valid format word (call hp_encoder_workspace_to_f32 semantics: fp32 rows +
 * the word hp_encoder_forward writes).  The autograd bridge (ops.py) uses each workspace for exactly one backward. */
long hp_encoder_backward_workspace_floats(int B, int out_size);
int hp_encoder_backward(int B, int Np, const float* x, const HpEncoderWeights* w, int out_size, int is_vae,
                        const float* eps, const int* argidx, const float* g, const float* f, const float* lv,
                        const float* grad_out, const float* grad_mu, const float* grad_explv, const HpEncoderGrads* grads,
                        float* ws, const float* fwd_ws, int dedup, hpStream_t stream);
/* ... with grad_out a column block of a wider matrix (row stride grad_out_ld >= out_size). */
int hp_encoder_backward_ld(int B, int Np, const float* x, const HpEncoderWeights* w, int out_size, int is_vae,
                           const float* eps, const int* argidx, const float* g, const float* f, const float* lv,
                           const float* grad_out, int grad_out_ld, const float* grad_mu, const float* grad_explv,
                           const HpEncoderGrads* grads, float* ws, const float* fwd_ws, int dedup, hpStream_t stream);
/* Both encoders of a HyperPocket step in one call, on one stream (round 3): with the forward's workspaces at hand and
 * dedup != 0 the two conv stacks share four launches (sort, a row-block chain delta4 -> delta1 on the matrix cores, one
 * grouped launch for every weight/bias gradient, an ordered reduce — csrc/enc_bwd.hip) and the two fc/mu/std tails three.
 * Results are those of two hp_encoder_backward_ld calls, bit for bit.  Autograd of model/encoder.py:14-53 for
 * model/full_model.py:106-112's two encoders. */
typedef struct HpEncoderBwdIO {
    const float* x;              /* (B, Np, 3) */
    const HpEncoderWeights* w;
    const float* eps;            /* VAE only */
    const int* argidx;
    const float *g, *f, *lv;
    const float *grad_out, *grad_mu, *grad_explv;
    const HpEncoderGrads* gr;
    float* ws;                   /* hp_encoder_backward_workspace_floats */
    const float* fwd_ws;         /* the forward's workspace, or NULL */
    int is_vae;
    int grad_out_ld;
} HpEncoderBwdIO;
int hp_encoder_backward_pair(int B, int Np, int out_size, const HpEncoderBwdIO* io /* [2] */, int dedup, hpStream_t stream);
/* ... with a second stream `after` (may be NULL) ordered behind the two tails' launches (an event recorded on `stream`, a
 * wait on `after`): what the caller enqueues on `after` next starts when the tails are done and runs beside the conv-stack
 * launches — the engine's HBM-bound heads dW + Adam pass, which would otherwise hold the tails' first launch back. */
int hp_encoder_backward_pair_ordered(int B, int Np, int out_size, const HpEncoderBwdIO* io /* [2] */, int dedup, hpStream_t stream,
                                     hpStream_t after);
/* Parity-test switch: 0 sends every encoder backward through round 2's layered launch sequence (sort, gather, a dX GEMM,
 * a dW GEMM and a split-K reduce per layer), 1 (default) through the fused kernels when fwd_ws != NULL and dedup != 0.
 * Returns the previous setting. */
/* [test hook: process-wide, not thread-safe — see the header comment] */
int hp_encoder_backward_set_fused(int on);
/* The fused backward's delta chain (delta4 -> delta1 of the critical rows) runs on the f16 matrix pipe with split fp32 operands
 * (csrc/enc_bwd_f16.hip; environment HP_EB_CHAIN16, default 1); 0 selects round 3's fp32 MFMA chain, -1 the environment's choice.
 * Returns the previous setting (-1: never set). */
/* [test hook: process-wide, not thread-safe — see the header comment] */
int hp_encoder_backward_set_chain_f16(int on);
/* The encoders' conv stack (model/encoder.py:14-28) runs on the f16 matrix pipe with every fp32 operand split into two
 * f16 pieces (three MFMA products per block; as close to fp64 as the fp32 fma chain — csrc/conv_split.hip).  0 sends it
 * through the fp32 MFMA GEMMs instead (also: environment HP_CONV_SPLIT=0).  Returns the previous setting. */
/* [test hook: process-wide, not thread-safe — see the header comment] */
int hp_conv_split_set(int on);
/* The same split-f16 GEMM as a stand-alone primitive: C = act(X W^T + b), X (M,K) and W (N,K) fp32 of either sign, row-major;
 * N % 128 == 0, K % 32 == 0, K <= 512.  prepare forms max|X| per 128-row tile and the f16 pieces / per-row exponents of W in ws
 * (hp_gemm_f16x2_workspace_floats(M, N, K) floats — inside the encoder stack the producing layer's epilogue and one prep launch
 * per forward do this); run is the GEMM launch alone (bench.py times it).  relu != 0: max(., 0). */
long hp_gemm_f16x2_workspace_floats(long M, int N, int K);
int hp_gemm_f16x2_prepare(long M, int N, int K, const float* X, const float* W, float* ws, hpStream_t stream);
int hp_gemm_f16x2_run(long M, int N, int K, const float* X, const float* bias, float* C, int relu, const float* ws, hpStream_t stream);

/* The heads' forward (theta = t5 . W^T + b at B <= 64, 156 MB of weights) runs as a streaming kernel on the bf16 matrix pipe with
 * every fp32 operand split into three bf16 pieces (exact split, six products: csrc/heads_fwd.hip; environment HP_HEADS_FWD,
 * default 1); 0 selects the tiled fp32 GEMM + split-K reduce, -1 the environment's choice.  Returns the previous setting. */
/* [test hook: process-wide, not thread-safe — see the header comment] */
int hp_hypernet_set_heads_stream(int on);
/* HyperNetwork.forward (model/hyper_network.py:41-43): latent (B,in) -> theta (B,theta_ld); t = saved trunk
 * activations (hp_hypernet_saved_floats floats) for the backward. */
long hp_hypernet_saved_floats(int B);
int hp_hypernet_forward(int B, int in_size, const float* latent, const HpHyperWeights* w, float* t, float* theta,
                        int theta_ld, hpStream_t stream);
long hp_hypernet_backward_workspace_floats(int B);
int hp_hypernet_backward(int B, int in_size, const float* latent, const HpHyperWeights* w, const float* t,
                         const float* grad_theta, int theta_ld, const HpHyperGrads* grads, float* grad_latent /* or NULL */,
                         float* ws, hpStream_t stream);
/* ... with a second stream `after` (may be NULL) ordered behind the last reader of the heads' weights inside the call (d t5 =
 * d theta . W): an in-place update of those weights enqueued on `after` next (hp_hypernet_heads_dw_adam) starts beside the trunk's
 * backward launches instead of behind the whole call. */
int hp_hypernet_backward_ordered(int B, int in_size, const float* latent, const HpHyperWeights* w, const float* t,
                                 const float* grad_theta, int theta_ld, const HpHyperGrads* grads, float* grad_latent /* or NULL */,
                                 float* ws, hpStream_t stream, hpStream_t after);
/* Data-parallel form of the heads' weight gradient (no counterpart in the reference, which has no distributed code:
 * SURVEY 8e).  grads->head_w[0] == NULL makes hp_hypernet_backward skip dW of the heads (bias gradients and d latent are
 * still produced); ranks then all-gather d theta (B x theta_ld) and t5 (B x 2048, hp_hypernet_t5_offset(B) floats into
 * the forward's `t`) and each forms rows [r0, r0+rows) of the GLOBAL gradient of the (19011 x 2048) heads matrix:
 *   dW_rows = dtheta_all[:, r0:r0+rows]^T . t5_all   (contraction over the Kc = world*B gathered clouds). */
long hp_hypernet_t5_offset(int B);
long hp_hypernet_heads_dw_workspace_floats(void);
int hp_hypernet_heads_dw_rows(int Kc, int rows, int r0, const float* dtheta_all, int theta_ld, const float* t5_all,
                              float* dW_rows, float* ws, hpStream_t stream);

/* The heads' weight gradient AND its torch.optim.Adam update in one pass (no counterpart in the reference: PyTorch
 * materialises the 156 MB gradient and the optimiser re-reads it): rows [r0, r0+rows) of the (theta_ld x 2048) heads matrix,
 *   g = dtheta_all[:, r0:r0+rows]^T . t5_all (Kc clouds);  W_rows, m_rows (exp_avg), v_rows (exp_avg_sq) <- Adam(g), in place;
 * g is never written (6 x 156 MB of traffic instead of 8 x).  Pointers address row r0; 16-byte aligned; wd = 0. */
int hp_hypernet_heads_dw_adam(int Kc, int rows, int r0, const float* dtheta_all, int theta_ld, const float* t5_all,
                              float* W_rows, float* m_rows, float* v_rows, float lr, float beta1, float beta2, float eps,
                              int step, hpStream_t stream);
/* ... as a BACKGROUND stream: persistent 16-wave workgroups on `cus` of the 256 CUs (0: 176; environment HP_HEADS_WGS) — for a caller
 * that runs the pass on its own stream beside latency-built launches which need the other CUs.  Same results. */
int hp_hypernet_heads_dw_adam_bg(int Kc, int rows, int r0, const float* dtheta_all, int theta_ld, const float* t5_all,
                                 float* W_rows, float* m_rows, float* v_rows, float lr, float beta1, float beta2, float eps,
                                 int step, int cus, hpStream_t stream);

/* The M = B <= 64 chains (hypernetwork trunk, encoder fc/mu/std tail) run as skinny layer programs — ONE latency-built
 * launch per phase (layer), ordered by the kernel boundary, no reduce launches: csrc/skinny.hip; the one-persistent-launch
 * variant with a grid-wide barrier was measured and dropped — when their shapes allow, otherwise as tiled GEMM launches.  Diagnostic switch for parity tests: 0 forces the GEMM launches, 1 the layer programs, -1 the default
 * (on; HP_SKINNY=0 in the environment turns it off).  Returns the previous setting.  No reference counterpart. */
/* [test hook: process-wide, not thread-safe — see the header comment] */
int hp_skinny_set_enabled(int on);

/* The B per-cloud TargetNetworks of one step at once (model/full_model.py:70-74, model/target_network.py:6-45).
 * theta (B,theta_ld): [W1 b1 | W2 b2 | ... | Wout bout] per cloud; pts (B,N,3) -> y (B,N,3) (rec[b] = y[b]^T).
 * acts: hidden activations kept for the backward (hp_target_saved_floats floats). */
long hp_target_theta_size(int n_hidden, const int* channels);
long hp_target_saved_floats(int B, int N, int n_hidden, const int* channels);
int hp_target_forward(int B, int N, int n_hidden, const int* channels, const float* theta, int theta_ld, const float* pts,
                      float* acts, float* y, hpStream_t stream);
long hp_target_backward_workspace_floats(int B, int N, int n_hidden, const int* channels);
int hp_target_backward(int B, int N, int n_hidden, const int* channels, const float* theta, int theta_ld, const float* pts,
                       const float* acts, const float* grad_y, float* grad_theta, float* ws, hpStream_t stream);

/* The same decoder for the published architecture (n_hidden = 4, channels 32/64/128/64; hp_target_fused_supported
 * says so) as ONE kernel per direction: the cloud's 19 011 weights sit in LDS, activations stay in registers, the
 * backward recomputes the forward (nothing is saved) and writes per-workgroup partial d theta into ws
 * (hp_target_fused_workspace_floats floats), added in order by a second kernel.  Same results as
 * hp_target_forward / hp_target_backward up to fp32 summation order. */
int hp_target_fused_supported(int n_hidden, const int* channels);
long hp_target_fused_workspace_floats(int B, int N);
/* The fused target-network forward computes its hidden layers on the f16 matrix pipe from two f16 pieces per fp32 operand
 * (csrc/target_fused.hip; the arithmetic of csrc/conv_split.hip with per-wave / per-channel scales).  0 selects the fp32 MFMA
 * forward (also: environment HP_TARGET_F16=0).  Returns the previous setting. */
/* [test hook: process-wide, not thread-safe — see the header comment] */
int hp_target_fused_set_f16(int on);
int hp_target_fused_forward(int B, int N, const float* theta, int theta_ld, const float* pts, float* y, hpStream_t stream);
int hp_target_fused_backward(int B, int N, const float* theta, int theta_ld, const float* pts, const float* grad_y,
                             float* grad_theta, float* ws, hpStream_t stream);

/* ------------------------------------------------------------------------------------------
 * Auxiliary kernels of the step
 * ------------------------------------------------------------------------------------------ */
/* Decoder input points (utils/points.py:8-36 distribution) for total = B*N points, Philox(seed, offset). */
int hp_sample_points(long total, float coef, unsigned long long seed, unsigned long long offset, float* out,
                     hpStream_t stream);
/* Random-plane slicer (datasets/utils/dataset_generator.py:26-39) for B clouds: part_a (B,target,3) is the side of an
 * accepted random plane that holds exactly `target` points, part_b (B,N-target,3) the rest, both in original order;
 * plane (B,4) = normal + bias of the accepted plane; status (B) int: 0 ok, 1 none accepted in max_rounds*4 draws. */
int hp_slice_clouds(int B, int N, int target, const float* pts, unsigned long long seed, int max_rounds, float* part_a,
                    float* part_b, float* plane, int* status, hpStream_t stream);
/* The same split with the CALLER's candidate planes: planes (B, R, 4) float64 device memory = (params, bias) of
 * dataset_generator.py:6-8's HyperPlane; cloud i tries planes[i,0..R) in order and classifies every point in float64 exactly
 * as HyperPlane.check_point (:10-11) — with the planes numpy's generator hands the reference, part_a / part_b ARE
 * SlicedDatasetGenerator.generate_item's two return values (tests/golden/slicer.npz).  plane_idx (B): index of the accepted
 * candidate, -1 (and status 1) when none of the R was accepted. */
int hp_slice_clouds_planes(int B, int N, int target, const float* pts, const double* planes, int R, float* part_a,
                           float* part_b, int* plane_idx, int* status, hpStream_t stream);
/* KLD term of core/epoch_loops.py:29-30 and its gradients */
int hp_kld_forward(long n, int batch, const float* explv, const float* mu, float* out, hpStream_t stream);
int hp_kld_backward(long n, int batch, const float* explv, const float* mu, const float* grad_out, float* grad_explv,
                    float* grad_mu, hpStream_t stream);
/* TrainEngine glue — the step's scalar loss terms in one launch (core/epoch_loops.py:26-31 + the optional EMD term):
 * out[0] = loss_r = c_cd*cd[0], out[1] = loss_kld = kld[0], out[2] = loss_emd = c_emd * sum_b cost[b], out[3] = their sum.
 * kld / cost may be NULL (term absent). */
int hp_step_losses(int b, const float* cd, const float* kld, const float* cost, float c_cd, float c_emd, float* out /* 4 */,
                   hpStream_t stream);

/* torch.optim.Adam(lr, betas, eps, weight_decay=0, amsgrad=False) over n contiguous fp32 parameters (core/main.py:62-66) */
int hp_adam_step(long n, float* p, const float* g, float* m, float* v, float lr, float beta1, float beta2, float eps,
                 int step, float grad_scale, hpStream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* HYPERPOCKET_HIP_H */
