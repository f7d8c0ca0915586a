// Internal interface of conv_split.hip: the encoders' conv stack on the f16 matrix pipe (split-fp32 operands).
#pragma once
#include <hip/hip_runtime.h>

// The "split area" of one encoder lives behind the activations in its forward workspace (model.hip: enc_fwd_ws):
//   [0, 1408)                   weight exponents e_w of layers 2..5 (128 + 256 + 512 + 512 rows)
//   [HP_CS_HI_OFF, ...)         f16 pieces of W2..W5 (2 x 434176 halfs): per row and 32-deep k-tile [hi 32 | lo 32]
//   [HP_CS_AMAX_OFF, ...)       max|h_l| per 128-row tile, layers 1..4, as float bits: 4 arrays of hp_conv_split_tiles_pad(R)
//   [.. + 4 tp, ...)            round 4 (conv_pp.hip): block exponents of the P-format activations h1..h4: 4 tp ints per layer
//                               ([tile][column block], hp_conv_pp_ncb(l) blocks per 128-row tile in use), then the format word
//                               (HP_PP_FMT_P while h1..h4 of this workspace hold P-format, else fp32) + 3 words of padding
#define HP_CS_WEXP_OFF 0L
#define HP_CS_HI_OFF 1408L
#define HP_CS_LO_OFF (HP_CS_HI_OFF + 434176L / 2)
#define HP_CS_AMAX_OFF (HP_CS_LO_OFF + 434176L / 2)

#define HP_PP_EXP_MAX 54          /* a block whose max is below 2^-40 keeps this exponent */
#define HP_PP_FMT_F32 0
#define HP_PP_FMT_P 0x50464d54   /* 'PFMT' */

inline long hp_conv_split_tiles_pad(long R) { return ((R + 127) / 128 + 3) / 4 * 4; }
long hp_conv_split_area_floats(long R);   // R = B * Np rows
bool hp_conv_split_enabled();
// W0 / W1: conv_w[1..4] of the first / second encoder (W1 ignored when n = 1); sArea: distance in floats between the areas
int hp_conv_split_prep(int n, const float* const* W0, const float* const* W1, float* area0, long sArea, long R, int fmt, hipStream_t stream);   // fmt: the workspace's format word (HP_PP_FMT_*)
int hp_conv_split_layer1(int n, const float* x, long sXz, const float* W, long sWz, const float* b, long sBz, float* h1, long sHz,
                         float* area0, long sArea, long R, hipStream_t stream);
int hp_conv_split_layer(int l, int n, const float* X, long sXz, const float* bias, long sBiasz, float* C, long sCz, float* area0,
                        long sArea, long M, int relu, int colmax, float* cmax, int* cidx, int group_rows, hipStream_t stream);

// ---- conv_pp.hip (round 4): layers 1..5 on P-format activations (both MFMA operands DMA-staged, 256 x 256 tiles)
bool hp_conv_presplit_enabled();
long hp_conv_pp_fmt_offset(long R);     // offset (floats) of the format word inside the split area
int hp_conv_pp_layer1(int n, const float* x, long sXz, const float* W, long sWz, const float* b, long sBz, float* h1, float* area0,
                      long sWs, long R, hipStream_t stream);
int hp_conv_pp_layer(int l, int n, const float* X, const float* bias, long sBiasz, float* C, float* area0, long sWs, long M,
                     float* cmax, int* cidx, int group_rows, hipStream_t stream);
int hp_conv_pp_mark(int n, float* area0, long sWs, long R, int fmt, hipStream_t stream);
int hp_conv_pp_unpack_ws(float* ws, long R, hipStream_t stream);
// exponent table of P-format h_l (l = 1..4) inside the split area (ints), tp = hp_conv_split_tiles_pad(R)
inline long hp_conv_pp_exp_offset(int l, long tp) { return HP_CS_AMAX_OFF + 4 * tp + (long)(l - 1) * 4 * tp; }   // room for 4 column blocks per tile
// column blocks per 128-row tile of P-format h_l (l = 1..4): the producing launch's column tiles (conv_pp.hip: launch_pp's choice)
int hp_conv_pp_ncb(int l);
