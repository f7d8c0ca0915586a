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

/* approxmatch  (structural_loss.cpp:11, approxmatch.cu:330-338)
 * match (b, m, n), temp (b, 2*(n+m)) as in the reference.  One extra argument: `ws`, scratch of
 * hp_approxmatch_workspace_floats(b,n,m) floats (the per-level scaling vectors; lets `match` be
 * written once instead of read-modify-written nine times). */
long hp_approxmatch_workspace_floats(int b, int n, int m);
int hp_approxmatch(int b, int n, int m, const float* xyz1, const float* xyz2, float* match, float* temp, float* ws,
                   hpStream_t stream);

/* matchcost  (structural_loss.cpp:12, approxmatch.cu:340-347); `partials`: scratch of
 * hp_matchcost_workspace_floats floats (ordered two-stage sum instead of one block per cloud). */
long hp_matchcost_workspace_floats(int b, int n, int m);
int hp_matchcost(int b, int n, int m, const float* xyz1, const float* xyz2, const float* match, float* out,
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

#ifdef __cplusplus
}
#endif
#endif /* HYPERPOCKET_HIP_H */
