/* Parameter-pointer tables of the model entry points (mirrored in include/hyperpocket_hip.h). */
#pragma once

#define HP_MAX_HEADS 8
#define HP_MAX_TN_LAYERS 8

/* model/encoder.py:14-36 — shapes are fixed by the reference's code: conv 3-64-128-256-512-512,
 * fc 512x512, mu/std (out,512).  Weights row-major (out,in) exactly as nn.Conv1d(k=1)/nn.Linear
 * store them. */
typedef struct HpEncoderWeights {
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

/* One encoder's buffers for hp_encoder_forward_pair (the argument list of hp_encoder_forward as a struct). */
typedef struct HpEncoderIO {
    const float* x;              /* (B, Np, 3) */
    const HpEncoderWeights* w;
    const float* eps;            /* VAE only */
    int* argidx;
    float *g, *f, *mu, *lv, *z, *explv, *ws;
    int is_vae;
    int out_ld; /* row stride of the primary output (z of a VAE encoder, mu of a plain one); 0 = out_size (dense).  The two
                   encoders of a pair can so write the halves of one (B, 2*out) latent [z | real mu] directly */
} HpEncoderIO;

/* One encoder's buffers for hp_encoder_backward_pair (the argument list of hp_encoder_backward_ld as a struct). */
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

/* model/hyper_network.py:16-36 — trunk in->64->128->512->1024->2048, heads 2048->head_out[h] */
typedef struct HpHyperWeights {
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

#ifdef __cplusplus
/* hypernet.hip (internal): fragment-direct heads kernels */
bool hp_heads_dw_fast_ok(int cols, const float* t5, const float* out);
/* heads_fwd.hip: the heads' forward at B <= 64 as a streaming bf16-pipe kernel */
bool hp_heads_fwd_enabled();
int hp_heads_fwd_set(int on);
long hp_heads_fwd_ws_floats();
bool hp_heads_fwd_ok(int B, int N, int K, const float* t5, const float* W, const float* ws);
int hp_heads_fwd(int B, int N, const float* t5, const float* W, const float* bias, float* theta, int theta_ld, float* ws,
                 hipStream_t stream);
int hp_heads_dw_launch(int Kc, int rows, int r0, const float* dtheta, int theta_ld, const float* t5, int cols, float* dW,
                       float* db, hipStream_t stream);
#endif
