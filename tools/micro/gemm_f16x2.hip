// PROTOTYPE: the encoders' pointwise-conv GEMM  C = relu(X . W^T + b)  (/root/reference/model/encoder.py:14-28) on the f16
// matrix pipe with fp32-equivalent accuracy.  Every fp32 operand a is split, after an exact power-of-two scale, into two
// f16 pieces  a = hi + lo + r,  hi = rne16(a), lo = rne16(a - hi), |r| <= 2^-24 |a|  (the residual of an fp32 rounding), and
// the product is formed as hi.hi + (hi.lo + lo.hi) by three v_mfma_f32_32x32x16_f16 (exact products, fp32 accumulate); the
// dropped lo.lo term is <= 2^-24 |a b|.  The f16 matrix rate is 16x the f32 one, so three products cost 3/16 of the f32 MFMA
// time.  This file measures (a) the error against fp64 next to the error of the fp32 fma chain gemm.hip computes, and (b) the
// time on the four conv shapes of the step.
//   hipcc -O3 --offload-arch=gfx950 -o gemm_f16x2 gemm_f16x2.hip && ./gemm_f16x2
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#define CK(x)                                                                     \
    do {                                                                          \
        hipError_t e_ = (x);                                                      \
        if (e_ != hipSuccess) {                                                   \
            printf("%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));      \
            exit(1);                                                              \
        }                                                                         \
    } while (0)

constexpr int BM = 128, BN = 128, BK = 32, LDH = BK + 8;   // LDS rows of 40 halfs = 80 bytes: conflict-free b128 reads

template <bool SEP, int MINB>
__global__ __launch_bounds__(256, MINB) void split_gemm(const float* __restrict__ X, const _Float16* __restrict__ Whi,
                                                       const _Float16* __restrict__ Wlo, const float* __restrict__ bias,
                                                       float* __restrict__ C, int M, int N, int K, float sx, float unscale,
                                                       int store) {
    __shared__ __attribute__((aligned(16))) _Float16 Ah[BM * LDH];
    __shared__ __attribute__((aligned(16))) _Float16 Al[BM * LDH];
    __shared__ __attribute__((aligned(16))) _Float16 Bh[BN * LDH];
    __shared__ __attribute__((aligned(16))) _Float16 Bl[BN * LDH];
    const int tiles_n = N / BN;
    const int nwg = gridDim.x;
    int bid = blockIdx.x;
    {
        const int q = nwg >> 3, r8 = nwg & 7, xcd = bid & 7;
        bid = (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + (bid >> 3);
    }
    const int tile_n = bid % tiles_n, tile_m = bid / tiles_n;
    const int row0 = tile_m * BM, col0 = tile_n * BN;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, r = lane & 31, h = lane >> 5;
    const int wm = w >> 1, wn = w & 1;

    const float* pa[4];
    int a_off[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int idx = tid + e * 256, row = idx >> 3, kq = idx & 7;
        pa[e] = X + (long)(row0 + row) * K + 4 * kq;
        a_off[e] = row * LDH + 4 * kq;
    }
    const _Float16* pbh[2];
    const _Float16* pbl[2];
    int b_off[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const int idx = tid + e * 256, row = idx >> 2, c = idx & 3;
        pbh[e] = Whi + (long)(col0 + row) * K + 8 * c;
        pbl[e] = Wlo + (long)(col0 + row) * K + 8 * c;
        b_off[e] = row * LDH + 8 * c;
    }

    f32x16 acc[2][2], cor[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                acc[i][j][e] = 0.f;
                cor[i][j][e] = 0.f;
            }

    f32x4 ra[4];
    u32x4 rbh[2], rbl[2];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int e = 0; e < 4; ++e) ra[e] = *reinterpret_cast<const f32x4*>(pa[e] + k0);
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            rbh[e] = *reinterpret_cast<const u32x4*>(pbh[e] + k0);
            rbl[e] = *reinterpret_cast<const u32x4*>(pbl[e] + k0);
        }
    };
    auto stage = [&]() {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            f16x4 hi, lo;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float xs = ra[e][u] * sx;
                const _Float16 hh = (_Float16)xs;
                hi[u] = hh;
                lo[u] = (_Float16)(xs - (float)hh);
            }
            *reinterpret_cast<f16x4*>(&Ah[a_off[e]]) = hi;
            *reinterpret_cast<f16x4*>(&Al[a_off[e]]) = lo;
        }
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            *reinterpret_cast<u32x4*>(&Bh[b_off[e]]) = rbh[e];
            *reinterpret_cast<u32x4*>(&Bl[b_off[e]]) = rbl[e];
        }
    };
    auto compute = [&]() {
#pragma unroll
        for (int t = 0; t < BK / 16; ++t) {
            f16x8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int o = (wm * 64 + i * 32 + r) * LDH + 16 * t + 8 * h;
                ah[i] = *reinterpret_cast<const f16x8*>(&Ah[o]);
                al[i] = *reinterpret_cast<const f16x8*>(&Al[o]);
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int o = (wn * 64 + j * 32 + r) * LDH + 16 * t + 8 * h;
                bh[j] = *reinterpret_cast<const f16x8*>(&Bh[o]);
                bl[j] = *reinterpret_cast<const f16x8*>(&Bl[o]);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[j], acc[i][j], 0, 0, 0);
                    if (SEP) {
                        cor[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl[j], cor[i][j], 0, 0, 0);
                        cor[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh[j], cor[i][j], 0, 0, 0);
                    } else {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl[j], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh[j], acc[i][j], 0, 0, 0);
                    }
                }
        }
    };

    fetch(0);
    for (int k0 = 0; k0 < K; k0 += BK) {
        stage();
        __syncthreads();
        if (k0 + BK < K) fetch(k0 + BK);
        compute();
        __syncthreads();
    }
    if (!store) return;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = col0 + wn * 64 + j * 32 + r;
            const float bv = bias[col];
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = row0 + wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                float v = acc[i][j][e];
                if (SEP) v += cor[i][j][e];
                v = fmaxf(v * unscale + bv, 0.f);
                C[(long)row * N + col] = v;
            }
        }
}

static double urand() { return (rand() + 0.5) / (RAND_MAX + 1.0); }
static double nrand() { return sqrt(-2.0 * log(urand())) * cos(6.283185307179586 * urand()); }

static float pow2_scale(float amax, int target_exp) {   // 2^e with amax * 2^e in (2^(target-1), 2^target]
    int ex;
    frexpf(amax, &ex);   // amax = f * 2^ex, f in [0.5, 1)
    return ldexpf(1.f, target_exp - ex);
}

template <bool SEP, int MINB>
static void run(const char* name, int M, int N, int K, float xmag, int reps) {
    std::vector<float> X((size_t)M * K), W((size_t)N * K), b(N);
    // post-ReLU-like activations with a wide dynamic range; Xavier-uniform weights (core/setup.py:63-69 of the reference)
    for (size_t i = 0; i < X.size(); ++i) {
        const double g = nrand();
        X[i] = g > 0 ? (float)(g * exp(1.5 * nrand()) * xmag) : 0.f;
    }
    const double wb = sqrt(2.0) * sqrt(6.0 / (N + K));
    for (size_t i = 0; i < W.size(); ++i) W[i] = (float)((2 * urand() - 1) * wb);
    for (int i = 0; i < N; ++i) b[i] = (float)(0.01 * nrand());
    float ax = 0, aw = 0;
    for (float v : X) ax = fmaxf(ax, fabsf(v));
    for (float v : W) aw = fmaxf(aw, fabsf(v));
    const float sx = pow2_scale(ax, 14), sw = pow2_scale(aw, 14);
    std::vector<_Float16> Wh(W.size()), Wl(W.size());
    for (size_t i = 0; i < W.size(); ++i) {
        const float s = W[i] * sw;
        const _Float16 hh = (_Float16)s;
        Wh[i] = hh;
        Wl[i] = (_Float16)(s - (float)hh);
    }
    float *dX, *db, *dC;
    _Float16 *dWh, *dWl;
    CK(hipMalloc(&dX, X.size() * 4));
    CK(hipMalloc(&db, N * 4));
    CK(hipMalloc(&dC, (size_t)M * N * 4));
    CK(hipMalloc(&dWh, W.size() * 2));
    CK(hipMalloc(&dWl, W.size() * 2));
    CK(hipMemcpy(dX, X.data(), X.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(db, b.data(), N * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dWh, Wh.data(), W.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dWl, Wl.data(), W.size() * 2, hipMemcpyHostToDevice));
    const int grid = (M / BM) * (N / BN);
    const float unscale = 1.f / (sx * sw);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int store = 1; store >= 0; --store) {
        for (int i = 0; i < 3; ++i)
            hipLaunchKernelGGL((split_gemm<SEP, MINB>), dim3(grid), dim3(256), 0, 0, dX, dWh, dWl, db, dC, M, N, K, sx, unscale, store);
        CK(hipEventRecord(e0, 0));
        for (int i = 0; i < reps; ++i)
            hipLaunchKernelGGL((split_gemm<SEP, MINB>), dim3(grid), dim3(256), 0, 0, dX, dWh, dWl, db, dC, M, N, K, sx, unscale, store);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        const double us = ms * 1e3 / reps;
        printf("%-6s M=%d N=%d K=%d sep=%d minb=%d store=%d: %8.1f us  %7.1f TFLOP/s (fp32-equivalent)\n", name, M, N, K, (int)SEP, MINB,
               store, us, 2.0 * M * N * K / us * 1e-6);
    }
    // accuracy on sampled rows
    std::vector<float> Cs((size_t)M * N);
    CK(hipMemcpy(Cs.data(), dC, Cs.size() * 4, hipMemcpyDeviceToHost));
    double e_split = 0, e_chain = 0, m_split = 0, m_chain = 0, ref_rms = 0;
    long cnt = 0;
    for (int s = 0; s < 48; ++s) {
        const int row = (int)(urand() * M);
        for (int n = 0; n < N; ++n) {
            double d = b[n];
            float f = 0.f;
            for (int k = 0; k < K; ++k) {
                d += (double)X[(size_t)row * K + k] * (double)W[(size_t)n * K + k];
                f = fmaf(X[(size_t)row * K + k], W[(size_t)n * K + k], f);
            }
            f += b[n];
            const double ref = d > 0 ? d : 0;
            const double fc = f > 0 ? f : 0;
            const double es = fabs(Cs[(size_t)row * N + n] - ref), ec = fabs(fc - ref);
            e_split += es * es;
            e_chain += ec * ec;
            m_split = fmax(m_split, es);
            m_chain = fmax(m_chain, ec);
            ref_rms += ref * ref;
            ++cnt;
        }
    }
    printf("       vs fp64 (rms of outputs %.3e): split rms %.3e max %.3e | fp32 fma chain rms %.3e max %.3e | ratio rms %.2f max %.2f\n",
           sqrt(ref_rms / cnt), sqrt(e_split / cnt), m_split, sqrt(e_chain / cnt), m_chain, sqrt(e_split / e_chain), m_split / m_chain);
    hipFree(dX); hipFree(db); hipFree(dC); hipFree(dWh); hipFree(dWl);
}

int main() {
    srand(2020);
    const int M = 131072;   // both encoders of a B=64 step: 2 x 64 x 1024 points
    run<true, 2>("conv5", M, 512, 512, 1.f, 20);
    run<false, 2>("conv5", M, 512, 512, 1.f, 20);
    run<false, 3>("conv5", M, 512, 512, 1.f, 20);
    run<true, 2>("conv5s", M, 512, 512, 1e-4f, 5);   // small activations: the scale must carry them
    run<false, 3>("conv4", M, 512, 256, 1.f, 20);
    run<false, 3>("conv3", M, 256, 128, 1.f, 20);
    run<false, 3>("conv2", M, 128, 64, 1.f, 20);
    return 0;
}
