// Pointwise-conv forward GEMM of the PointNet encoders on gfx950:  C = act(X . W^T + b)  with the weight matrix read as
// W^T (K x N, row-major) — /root/reference/model/encoder.py:14-28 (Conv1d(k=1) + ReLU on (B*N) points x C_in channels).
//
// The general GEMM family (gemm.hip) stages BOTH operands through LDS because both are K-contiguous in memory.  The
// weights are small (<= 1 MB) and constant within a step: with W^T at hand (one transpose launch per forward,
// hp_conv_transpose_weights) the B operand needs no staging at all — its rows run along the output columns, so a lane loads
// 16 bytes = four INTERLEAVED column tiles straight from L2 and every such load feeds 4 x TM MFMAs (the idiom of
// enc_bwd.hip's chain kernel).  Only the activation tile goes through LDS (double-buffered, one barrier per 16-deep k-tile,
// k-permuted ds_read_b128 fragments as in gemm.hip).  Per 16 k a wave issues 8 global loads + TM LDS reads for 32 x TM MFMAs.
//
// Arithmetic: v_mfma_f32_32x32x2_f32 in the k order of gemm.hip's tiles (k-groups of 8, lane half h takes k = 8t + 4h + s in
// step s), so every output is bit-identical to the general kernel's — the parity tests and the backward's recompute path do
// not notice which of the two ran.
// PROTOTYPE (round 3, not part of the library): built into libhyperpocket_hip.so for one measurement by copying it to csrc/ —
// tools/bench_xwt.py: conv5 (65536 x 512 x 512) 323 us = 106 TFLOP/s against gemm.hip's 277 us = 124; conv4 171 vs 162 us;
// conv3 54.3 vs 54.5 us; conv2 22.3 vs 20.9 us — bit-identical outputs, but two 256-VGPR waves per SIMD behind one barrier per
// k-tile do not beat six 80-VGPR waves per SIMD.  docs/DESIGN_HISTORY.md 7b.
#include "../../3d-point-clouds-autocomplete_amd/csrc/hp_common.h"
#include "../../3d-point-clouds-autocomplete_amd/csrc/hp_gemm.h"
#include <cstdlib>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define HP_SB() __builtin_amdgcn_sched_barrier(0)

struct XwtParams {
    const float* A;      // (M, K) row-major, lda = K
    const float* Wt;     // (K, N) row-major
    const float* bias;   // (N) or NULL
    float* C;            // (M, ldc)
    float* cmax;         // COLMAX: (M / BM, N) partial maxima
    int* cidx;
    long sAz, sWz, sBiasz, sCz;   // batch strides (floats)
    int M, N, K, ldc, relu, group_rows, tiles_m, tiles_n;
};

__device__ __forceinline__ int drow(int e, int h) { return (e & 3) + 8 * (e >> 2) + 4 * h; }

constexpr int kBK = 16, kLDK = kBK + 4;

// WGM x WGN waves; a wave owns 32*TM rows x 128 columns (four interleaved tiles: tile t = columns {c0 + 4r + t}).
template <int TM, int WGM, int WGN, bool COLMAX>
__global__ __launch_bounds__(WGM* WGN * 64, 2) void xwt_kernel(const XwtParams p) {
    constexpr int NT = WGM * WGN * 64, BM = 32 * TM * WGM, BN = 128 * WGN;
    constexpr int NA = (BM * (kBK / 4)) / NT;      // float4 pieces of the A tile per thread
    static_assert((BM * (kBK / 4)) % NT == 0, "whole passes over the A tile");
    __shared__ __attribute__((aligned(16))) float As[2][BM * kLDK];
    // XCD-aware bijective remap of the tile id (neighbouring tiles share an A panel in one XCD's L2)
    const int nwg = gridDim.x;
    int bid = blockIdx.x;
    {
        const int q = nwg >> 3, r8 = nwg & 7, xcd = bid & 7;
        bid = (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + (bid >> 3);
    }
    const int tile_n = bid % p.tiles_n, tile_m = bid / p.tiles_n, z = blockIdx.y;
    const int row0 = tile_m * BM, col0 = tile_n * BN;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, r = lane & 31, h = lane >> 5;
    const int wm = w / WGN, wn = w % WGN;
    const float* A = p.A + (long)z * p.sAz;
    const float* Wt = p.Wt + (long)z * p.sWz;

    f32x16 acc[TM][4];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][t][e] = 0.f;

    // A staging: piece e of a thread = (row = idx / 4, k-quad = idx % 4), idx = tid + e*NT; rows clamped into range
    const float* pa[NA];
    int sa[NA];
#pragma unroll
    for (int e = 0; e < NA; ++e) {
        const int idx = tid + e * NT, row = idx >> 2, kq = idx & 3;
        pa[e] = A + (long)min(row0 + row, p.M - 1) * p.K + kq * 4;
        sa[e] = row * kLDK + kq * 4;
    }
    f32x4 ra[NA];
    auto fetchA = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int e = 0; e < NA; ++e) {
            ra[e] = *reinterpret_cast<const f32x4*>(pa[e]);
            pa[e] += kBK;
        }
    };
    auto stageA = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int e = 0; e < NA; ++e) *reinterpret_cast<f32x4*>(&As[buf][sa[e]]) = ra[e];
    };
    // B fragments of one k-tile: group g (8 k), step s: row k = 8g + 4h + s of W^T, 16 bytes at column c0 + 4r
    const float* wp = Wt + (long)(4 * h) * p.N + col0 + 128 * wn + 4 * r;
    f32x4 b0[2][4], b1[2][4];
    auto fetchB = [&](f32x4 (&dst)[2][4]) __attribute__((always_inline)) {
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
            for (int s = 0; s < 4; ++s) dst[g][s] = *reinterpret_cast<const f32x4*>(wp + (long)(8 * g + s) * p.N);
        wp += (long)kBK * p.N;
    };
    auto compute = [&](int buf, const f32x4 (&bw)[2][4]) __attribute__((always_inline)) {
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            f32x4 av[TM];
#pragma unroll
            for (int i = 0; i < TM; ++i)
                av[i] = *reinterpret_cast<const f32x4*>(&As[buf][(32 * TM * wm + 32 * i + r) * kLDK + 8 * g + 4 * h]);
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i][s], bw[g][s][0], acc[i][0], 0, 0, 0);
                    acc[i][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i][s], bw[g][s][1], acc[i][1], 0, 0, 0);
                    acc[i][2] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i][s], bw[g][s][2], acc[i][2], 0, 0, 0);
                    acc[i][3] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i][s], bw[g][s][3], acc[i][3], 0, 0, 0);
                }
        }
    };
    const int nk = p.K / kBK;                 // (host: K % 32 == 0, so nk is even)
    fetchA();
    fetchB(b0);
    stageA(0);
    __syncthreads();
    // k-tile kt: A image in LDS buffer kt & 1, B fragments in b0 (even) / b1 (odd); the next tile's loads fly under the MFMAs.
    // Past the last tile the loads re-read the first tile (valid memory, unused).
#pragma unroll 1
    for (int kt = 0; kt < nk; kt += 2) {
        fetchA();
        fetchB(b1);
        HP_SB();
        compute(0, b0);
        HP_SB();
        stageA(1);
        __syncthreads();
        if (kt + 2 >= nk) {                   // wrap the pointers: branch-free loads need valid addresses
#pragma unroll
            for (int e = 0; e < NA; ++e) pa[e] -= (long)nk * kBK;
            wp -= (long)nk * kBK * p.N;
        }
        fetchA();
        fetchB(b0);
        HP_SB();
        compute(1, b1);
        HP_SB();
        stageA(0);
        __syncthreads();
    }

    // epilogue.  Lane (r, h), register e of row tile i: row = row0 + 32*(TM*wm + i) + drow(e,h), columns c .. c+3
    const int c = col0 + 128 * wn + 4 * r;
    f32x4 bv = {0.f, 0.f, 0.f, 0.f};
    if (p.bias) bv = *reinterpret_cast<const f32x4*>(p.bias + (long)z * p.sBiasz + c);
    if (COLMAX) {
        // fused max-pool over this tile's rows (model/encoder.py:45): first row attaining the max wins
        __shared__ float smax[WGM][BN];
        __shared__ int sidx[WGM][BN];
        float best[4];
        int bi[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            best[t] = -__builtin_inff();
            bi[t] = 0x7fffffff;
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) {   // rows ascend with (i, e) for a fixed lane half
                const int row = row0 + 32 * (TM * wm + i) + drow(e, h);
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const float v = acc[i][t][e] + bv[t];
                    if (row < p.M && v > best[t]) {
                        best[t] = v;
                        bi[t] = row;
                    }
                }
            }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const float ov = __shfl_xor(best[t], 32, 64);
            const int oi = __shfl_xor(bi[t], 32, 64);
            if (ov > best[t] || (ov == best[t] && oi < bi[t])) {
                best[t] = ov;
                bi[t] = oi;
            }
            if (h == 0) {
                smax[wm][128 * wn + 4 * r + t] = best[t];
                sidx[wm][128 * wn + 4 * r + t] = bi[t];
            }
        }
        __syncthreads();
        for (int j = tid; j < BN; j += NT) {
            float b2 = smax[0][j];
            int i2 = sidx[0][j];
#pragma unroll
            for (int q = 1; q < WGM; ++q)
                if (smax[q][j] > b2) {   // later wave rows are larger: strict > keeps the first row
                    b2 = smax[q][j];
                    i2 = sidx[q][j];
                }
            p.cmax[(long)z * p.sCz + (long)tile_m * p.N + col0 + j] = b2;
            p.cidx[(long)z * p.sCz + (long)tile_m * p.N + col0 + j] = i2 % p.group_rows;
        }
        return;
    }
    float* C = p.C + (long)z * p.sCz;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int row = row0 + 32 * (TM * wm + i) + drow(e, h);
            if (row >= p.M) continue;
            f32x4 v = {acc[i][0][e] + bv[0], acc[i][1][e] + bv[1], acc[i][2][e] + bv[2], acc[i][3][e] + bv[3]};
            if (p.relu) v = {fmaxf(v[0], 0.f), fmaxf(v[1], 0.f), fmaxf(v[2], 0.f), fmaxf(v[3], 0.f)};
            *reinterpret_cast<f32x4*>(C + (long)row * p.ldc + c) = v;
        }
}

// Wt[z](k, j) = W[z](j, k): one workgroup per 32 x 32 tile through LDS
__global__ __launch_bounds__(256) void transpose_kernel(const float* __restrict__ W, long sWz, float* __restrict__ Wt, long sWtz, int N,
                                                        int K) {
    __shared__ float t[32][33];
    const int z = blockIdx.z, j0 = blockIdx.y * 32, k0 = blockIdx.x * 32;
    const int x = threadIdx.x & 31, y = threadIdx.x >> 5;
    const float* src = W + (long)z * sWz;
    float* dst = Wt + (long)z * sWtz;
    for (int q = y; q < 32; q += 8)
        if (j0 + q < N && k0 + x < K) t[q][x] = src[(long)(j0 + q) * K + k0 + x];
    __syncthreads();
    for (int q = y; q < 32; q += 8)
        if (k0 + q < K && j0 + x < N) dst[(long)(k0 + q) * N + j0 + x] = t[x][q];
}

}  // namespace

// Rows per output tile of hp_conv_xwt (the COLMAX partial layout): 128.
HP_API int hp_conv_xwt_tile_rows() { return 128; }

// Can the problem be served?  (A K-contiguous and 16-byte aligned rows, N a multiple of 128, K a multiple of 32.)
HP_API int hp_conv_xwt_ok(const HpGemmDesc* d) {
    static const bool on = [] { const char* e = getenv("HP_CONV_XWT"); return !(e && *e == '0'); }();
    if (!on || !d) return 0;
    if (d->sAk != 1 || d->sAi != d->K || d->sBk != 1 || d->sBj != d->K) return 0;      // X (M,K) and W (N,K), both dense
    if (d->N % 128 || d->K % 32 || d->K < 32 || d->M < 1) return 0;
    if (d->flags & ~(HP_GEMM_BIAS | HP_GEMM_RELU | HP_GEMM_COLMAX)) return 0;
    if (d->ksplit > 1 || d->dyn_count) return 0;
    if ((reinterpret_cast<uintptr_t>(d->A) | reinterpret_cast<uintptr_t>(d->bias) | reinterpret_cast<uintptr_t>(d->C)) & 15) return 0;
    if ((d->sAz | d->sCz | d->sBiasz) & 3) return 0;
    if (!(d->flags & HP_GEMM_COLMAX) && (d->ldc & 3)) return 0;
    return 1;
}

HP_API int hp_conv_transpose_weights(int batch, int N, int K, const float* W, long sWz, float* Wt, long sWtz, hipStream_t stream) {
    hipLaunchKernelGGL(transpose_kernel, dim3((K + 31) / 32, (N + 31) / 32, batch), dim3(256), 0, stream, W, sWz, Wt, sWtz, N, K);
    HP_RETURN_LAST_ERROR();
}

// C = act(A . Wt + b) for the descriptor `d` of the equivalent hp_gemm_f32 call (A = X, B = W); Wt: the transposed
// weights (K x N per batch entry, sWtz apart).
HP_API int hp_conv_xwt(const HpGemmDesc* d, const float* Wt, long sWtz, hipStream_t stream) {
    if (!hp_conv_xwt_ok(d)) return -1;
    XwtParams p{};
    p.A = d->A; p.Wt = Wt; p.bias = (d->flags & HP_GEMM_BIAS) ? d->bias : nullptr; p.C = d->C;
    p.cmax = d->cmax; p.cidx = d->cidx;
    p.sAz = d->sAz; p.sWz = sWtz; p.sBiasz = d->sBiasz; p.sCz = d->sCz;
    p.M = d->M; p.N = d->N; p.K = d->K; p.ldc = d->ldc; p.relu = (d->flags & HP_GEMM_RELU) ? 1 : 0;
    p.group_rows = d->group_rows;
    const bool colmax = d->flags & HP_GEMM_COLMAX;
    if (d->N % 256 == 0) {                      // 2 x 2 waves: 128 rows x 256 columns per workgroup
        p.tiles_m = (d->M + 127) / 128;
        p.tiles_n = d->N / 256;
        const dim3 grid(p.tiles_m * p.tiles_n, d->batch);
        if (colmax) hipLaunchKernelGGL((xwt_kernel<2, 2, 2, true>), grid, dim3(256), 0, stream, p);
        else hipLaunchKernelGGL((xwt_kernel<2, 2, 2, false>), grid, dim3(256), 0, stream, p);
    } else {                                    // 4 x 1 waves: 128 rows x 128 columns
        p.tiles_m = (d->M + 127) / 128;
        p.tiles_n = d->N / 128;
        const dim3 grid(p.tiles_m * p.tiles_n, d->batch);
        if (colmax) hipLaunchKernelGGL((xwt_kernel<1, 4, 1, true>), grid, dim3(256), 0, stream, p);
        else hipLaunchKernelGGL((xwt_kernel<1, 4, 1, false>), grid, dim3(256), 0, stream, p);
    }
    HP_RETURN_LAST_ERROR();
}
