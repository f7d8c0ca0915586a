// Hypernetwork heads on gfx950 — the three contractions with the (19011 x 2048) heads matrix W at M = B clouds
// (model/hyper_network.py:32-43: theta = t5 W^T + b, and the autograd of it).  W is 90 % of the model (156 MB): each of
// these is one pass over 156 MB with 32 FLOP per byte — neither a wide GEMM nor a GEMV.  The generic tiled GEMM of
// gemm.hip spends its time in prologues/epilogues here (70-78 us per pass); these kernels have no LDS stage at all:
// MFMA operand fragments are loaded from global memory straight into the registers the matrix core reads, and every
// wave is independent.
//
//  * heads_dw_kernel<ADAM>: rows [r0, r0+rows) of dW = dtheta^T t5 (contraction over the Kc gathered clouds, 64 on one
//    GPU).  A wave owns a 32-row x 128-column unit: per k-step one float of dtheta (rows along lanes: coalesced) and
//    one float4 of t5 per lane feed four v_mfma_f32_32x32x2_f32 — the four column tiles are the INTERLEAVED column
//    sets {4j+t}, so lane j ends up holding four consecutive columns of every row and the epilogue moves float4s.
//    ADAM = false stores the gradient (one 156 MB write).  ADAM = true never materialises it: the epilogue reads
//    W / exp_avg / exp_avg_sq at the same coordinates, applies torch.optim.Adam's update (core/main.py:62-66) and
//    writes them back — 6 x 156 MB instead of the 8 x 156 MB of "write dW, then run Adam over it", and one launch.
//    The bias gradient (column sums of d theta) rides on the A operand.
//
// Tried in the same style and NOT kept (measured, B=64): fragment-direct kernels for theta = t5 W^T (K-contiguous W: a
// lane per row of W reads 32-byte pieces at an 8 KB stride — 277 us against the tiled GEMM's 70 us, DRAM pages thrash)
// and for d t5 = d theta W (248 vs 78 us), and for the trunk's M = 64 layers (few waves with long dependent load chains:
// 119 us forward against ~100 us).  Operands whose contiguous direction is the contraction need the LDS transpose of
// gemm.hip; only dW — both operands contiguous along the lanes — streams well without it.
#include "hp_common.h"
#include "hp_model.h"
#include <algorithm>
#include <cmath>
#include <cstdlib>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kHdThreads = 256;          // 4 waves side by side along the columns
constexpr int kUnitRows = 32, kUnitCols = 128;
constexpr int kPref = 4;                 // k-steps of operand prefetch (L2-hit latency ~ 2 steps of 4 MFMAs)

struct AdamArgs {
    float* p;
    float* m;
    float* v;
    float b1, b2, eps, step_size, inv_sqrt_bc2;
};

struct HeadsDw {
    const float* dtheta;   // (Kc, theta_ld): column r0+i is row i of the unit's A operand
    const float* t5;       // (Kc, cols)
    float* dW;             // ADAM = false: (rows, cols) output
    AdamArgs ad;           // ADAM = true: rows of W / exp_avg / exp_avg_sq, same (rows, cols) indexing
    float* db;             // or NULL: db[r] = sum_k dtheta[k][r0 + r]  (bias gradient)
    int Kc, rows, r0, theta_ld, cols;
};

// one wave, one 32-row x 128-column unit
template <bool ADAM>
__device__ __forceinline__ void heads_dw_unit(const HeadsDw& a, int row0, int c0, int lane) {
    const int i = lane & 31, h = lane >> 5;
    const bool row_ok = row0 + i < a.rows;
    const float* ap = a.dtheta + a.r0 + row0 + i;             // + k * theta_ld
    const float* bp = a.t5 + c0 + 4 * i;                      // + k * cols
    const int steps = (a.Kc + 1) >> 1;
    float asum = 0.f;
    f32x16 acc[4] = {};
    float av[kPref];
    float4 bv[kPref];
    auto load = [&](int s, float& x, float4& y) {
        const int k = 2 * s + h;
        const bool ok = k < a.Kc;
        x = (ok && row_ok) ? ap[(long)k * a.theta_ld] : 0.f;
        y = ok ? *reinterpret_cast<const float4*>(bp + (long)k * a.cols) : make_float4(0.f, 0.f, 0.f, 0.f);
    };
#pragma unroll
    for (int s = 0; s < kPref; ++s) load(s, av[s], bv[s]);
    for (int s0 = 0; s0 < steps; s0 += kPref) {
#pragma unroll
        for (int u = 0; u < kPref; ++u) {
            const float x = av[u];
            const float4 y = bv[u];
            load(s0 + u + kPref, av[u], bv[u]);               // past the end: k >= Kc loads nothing
            asum += x;
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y.x, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y.y, acc[1], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y.z, acc[2], 0, 0, 0);
            acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y.w, acc[3], 0, 0, 0);
        }
    }
    if (a.db && c0 == 0) {                                    // even clouds in lanes 0-31, odd clouds in lanes 32-63
        const float other = __shfl_xor(asum, 32, 64);
        if (h == 0 && row_ok) a.db[row0 + i] = asum + other;
    }
    // C layout of 32x32: lane (i, h), register e -> row (e&3) + 8*(e>>2) + 4*h, column i of the tile; tile t's column i
    // is the real column c0 + 4*i + t: the lane holds columns c0+4i .. c0+4i+3 of 16 rows.
    // Epilogue in batches of kEB rows: 3 x kEB float4 loads in flight per lane (measured: prefetching the first batch above
    // the contraction and double-buffering the batches was slower, 210 vs 186 us — the 16 waves per CU already keep the
    // HBM queues full, the extra registers only cost prefetch depth in the contraction).
    const long col = c0 + 4 * i;
    constexpr int kEB = 2;
#pragma unroll
    for (int eb = 0; eb < 16; eb += kEB) {
        float4 pp[kEB], mm[kEB], vv[kEB];
        long off[kEB];
        bool ok[kEB];
#pragma unroll
        for (int q = 0; q < kEB; ++q) {
            const int e = eb + q;
            const int r = row0 + (e & 3) + 8 * (e >> 2) + 4 * h;
            ok[q] = r < a.rows;
            off[q] = (long)r * a.cols + col;
            if (ADAM && ok[q]) {
                pp[q] = *reinterpret_cast<const float4*>(a.ad.p + off[q]);
                mm[q] = *reinterpret_cast<const float4*>(a.ad.m + off[q]);
                vv[q] = *reinterpret_cast<const float4*>(a.ad.v + off[q]);
            }
        }
#pragma unroll
        for (int q = 0; q < kEB; ++q) {
            const int e = eb + q;
            if (!ok[q]) continue;
            const float g[4] = {acc[0][e], acc[1][e], acc[2][e], acc[3][e]};
            if (!ADAM) {
                *reinterpret_cast<float4*>(a.dW + off[q]) = make_float4(g[0], g[1], g[2], g[3]);
                continue;
            }
            float* pa = &pp[q].x;
            float* ma = &mm[q].x;
            float* va = &vv[q].x;
#pragma unroll
            for (int t = 0; t < 4; ++t) {      // torch.optim.Adam (aux_kernels.hip adam_kernel, same operation order)
                ma[t] = a.ad.b1 * ma[t] + (1.0f - a.ad.b1) * g[t];
                va[t] = a.ad.b2 * va[t] + (1.0f - a.ad.b2) * g[t] * g[t];
                const float denom = __builtin_sqrtf(va[t]) * a.ad.inv_sqrt_bc2 + a.ad.eps;
                pa[t] = pa[t] - a.ad.step_size * (ma[t] / denom);
            }
            *reinterpret_cast<float4*>(a.ad.p + off[q]) = pp[q];
            *reinterpret_cast<float4*>(a.ad.m + off[q]) = mm[q];
            *reinterpret_cast<float4*>(a.ad.v + off[q]) = vv[q];
        }
    }
}

// 4 workgroups per CU (<= 128 VGPRs): the Adam epilogue is an HBM stream and needs the waves in flight
template <bool ADAM>
__global__ __launch_bounds__(kHdThreads) __attribute__((amdgpu_waves_per_eu(4, 4))) void heads_dw_kernel(const HeadsDw a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row0 = blockIdx.y * kUnitRows;
    const int c0 = (blockIdx.x * (kHdThreads / 64) + wave) * kUnitCols;
    if (c0 >= a.cols) return;
    heads_dw_unit<ADAM>(a, row0, c0, lane);
}

// The same units walked by PERSISTENT 16-wave workgroups, one per CU (4 waves per SIMD is all a CU takes of this kernel), on
// only `gridDim.x` of the 256 CUs: the pass is an HBM stream — ~190 CUs saturate it — and the CUs it leaves alone are where the
// latency-built launches of the trunk's backward and the encoders' tails run meanwhile (round 4: the pass starts right behind
// the heads' dX instead of behind the tails; no CU mask, no special stream: the footprint is the grid).
template <bool ADAM>
__global__ __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(4, 4))) void heads_dw_persist_kernel(const HeadsDw a, int units_x,
                                                                                                         int units) {
    const int lane = threadIdx.x & 63;
    const int nw = gridDim.x * 16;
    for (int u = blockIdx.x * 16 + (threadIdx.x >> 6); u < units; u += nw)
        heads_dw_unit<ADAM>(a, (u / units_x) * kUnitRows, (u % units_x) * kUnitCols, lane);
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

// Can the fragment-direct kernels serve this shape?  (cols in whole 128-column units, 16-byte aligned rows.)
bool hp_heads_dw_fast_ok(int cols, const float* t5, const float* out) {
    return cols > 0 && cols % kUnitCols == 0 && aligned16(t5) && aligned16(out);
}

// dW_rows (rows x cols) = dtheta[:, r0:r0+rows]^T . t5     (internal: model.hip's hp_hypernet_backward / _heads_dw_rows)
int hp_heads_dw_launch(int Kc, int rows, int r0, const float* dtheta, int theta_ld, const float* t5, int cols, float* dW,
                       float* db, hipStream_t stream) {
    HeadsDw a{dtheta, t5, dW, {}, db, Kc, rows, r0, theta_ld, cols};
    const dim3 grid((cols / kUnitCols + 3) / 4, (rows + kUnitRows - 1) / kUnitRows);
    hipLaunchKernelGGL(heads_dw_kernel<false>, grid, dim3(kHdThreads), 0, stream, a);
    return (int)hipGetLastError();
}

// Heads' weight gradient and its Adam update in one pass (no counterpart in the reference: PyTorch materialises the
// gradient and torch.optim.Adam re-reads it).  Rows [r0, r0+rows) of the (theta_ld x 2048) heads matrix:
//   g = dtheta_all[:, r0:r0+rows]^T . t5_all  (Kc clouds);  W, exp_avg, exp_avg_sq <- Adam(W, g)   (torch semantics, wd = 0)
// W_rows / m_rows / v_rows point at row r0 of the respective (.., 2048) matrices.  The gradient itself is never written.
namespace {
int heads_dw_adam_impl(int Kc, int rows, int r0, const float* dtheta_all, int theta_ld, const float* t5_all, float* W_rows,
                       float* m_rows, float* v_rows, float lr, float beta1, float beta2, float eps, int step, int cus,
                       hipStream_t stream) {
    HP_CHECK_ARG(Kc > 0 && rows >= 0 && r0 >= 0 && r0 + rows <= theta_ld && step >= 1);
    if (rows == 0) return 0;
    HP_CHECK_ARG(dtheta_all && t5_all && W_rows && m_rows && v_rows && aligned16(t5_all) && aligned16(W_rows) &&
                 aligned16(m_rows) && aligned16(v_rows));
    const double bc1 = 1.0 - std::pow((double)beta1, step), bc2 = 1.0 - std::pow((double)beta2, step);
    HeadsDw a{dtheta_all, t5_all, nullptr, {W_rows, m_rows, v_rows, beta1, beta2, eps, (float)((double)lr / bc1),
              (float)(1.0 / std::sqrt(bc2))}, nullptr, Kc, rows, r0, theta_ld, 2048};
    const int ux = 2048 / kUnitCols, uy = (rows + kUnitRows - 1) / kUnitRows;
    if (cus > 0) {
        hipLaunchKernelGGL(heads_dw_persist_kernel<true>, dim3(std::min(cus, (ux * uy + 15) / 16)), dim3(1024), 0, stream, a, ux, ux * uy);
        HP_RETURN_LAST_ERROR();
    }
    const dim3 grid((ux + 3) / 4, uy);
    hipLaunchKernelGGL(heads_dw_kernel<true>, grid, dim3(kHdThreads), 0, stream, a);
    HP_RETURN_LAST_ERROR();
}
}  // namespace

HP_API int hp_hypernet_heads_dw_adam(int Kc, int rows, int r0, const float* dtheta_all, int theta_ld, const float* t5_all,
                                     float* W_rows, float* m_rows, float* v_rows, float lr, float beta1, float beta2, float eps,
                                     int step, hipStream_t stream) {
    return heads_dw_adam_impl(Kc, rows, r0, dtheta_all, theta_ld, t5_all, W_rows, m_rows, v_rows, lr, beta1, beta2, eps, step, 0, stream);
}
// The same pass as a BACKGROUND stream: persistent 16-wave workgroups on `cus` of the 256 CUs (0: the default, 176; environment
// HP_HEADS_WGS overrides).  For a caller that runs it on a stream of its own beside latency-built launches which need the other
// CUs (core/engine.py FusedHeadsAdam behind hp_hypernet_backward_ordered); alone on the chip it is slower than the plain form.
HP_API int hp_hypernet_heads_dw_adam_bg(int Kc, int rows, int r0, const float* dtheta_all, int theta_ld, const float* t5_all,
                                        float* W_rows, float* m_rows, float* v_rows, float lr, float beta1, float beta2, float eps,
                                        int step, int cus, hipStream_t stream) {
    static const int kWgs = [] {
        const char* e = getenv("HP_HEADS_WGS");
        return e ? atoi(e) : 176;
    }();
    const int use = cus > 0 ? cus : kWgs;
    return heads_dw_adam_impl(Kc, rows, r0, dtheta_all, theta_ld, t5_all, W_rows, m_rows, v_rows, lr, beta1, beta2, eps, step, use, stream);
}
