// HyperPocket model on gfx950: PointNet encoder, hypernetwork and the batched per-cloud target
// network, forward and backward, as host-side launch sequences over the fp32-MFMA GEMM family
// (gemm.hip) plus the small kernels that are not contractions.
//
// Replaces (behaviour, not code):
//   /root/reference/model/encoder.py:14-53         Conv1d(k=1) x5 -> max over points -> fc -> mu/std
//   /root/reference/model/hyper_network.py:16-43   5-layer trunk + one linear head per target layer
//   /root/reference/model/target_network.py:6-45   per-cloud MLP whose weights are a slice of theta
//   /root/reference/model/full_model.py:70-74      the B-iteration Python loop that instantiates one
//                                                  TargetNetwork per cloud (here: ONE batched launch
//                                                  per layer, weights addressed by a per-cloud stride)
// and the autograd graph PyTorch would build for them.
//
// Encoder backward uses the exact critical-point sparsity of a PointNet:
// d/d h5[n,c] is non-zero only at n = argmax_n h5[n,c]; all layers are pointwise, so only the
// B*512 (cloud, channel) critical rows carry gradient.  Those rows' activations are copied out of the
// forward's workspace (or, if the caller dropped it, recomputed — bit-identical: the GEMM's k-order does
// not depend on the row), and the backward GEMMs run over B*512 rows instead of B*Np.
#include "hp_common.h"
#include "hp_gemm.h"
#include "hp_conv_split.h"
#include "hp_model.h"
#include "hp_skinny.h"
#include "hp_enc_bwd.h"
#include <algorithm>
#include <cstdlib>

extern "C" int hp_gemm_f32(const HpGemmDesc* d, hipStream_t stream);
extern "C" int hp_gemm_tile_rows(const HpGemmDesc* d);
extern "C" int hp_colsum_f32(int batch, int M, int N, const float* X, long sXz, int ldx, const float* mask, long sMaskz,
                             int ldmask, float* out, long sOz, float* ws, hipStream_t stream);

namespace {

constexpr int kEnc[6] = {3, 64, 128, 256, 512, 512};   // model/encoder.py:14-28
constexpr int kTrunk[5] = {64, 128, 512, 1024, 2048};  // model/hyper_network.py:16-30
constexpr long kSplitWs = 4L << 20;                    // floats reserved for split-K slabs

#define TRY(x)            \
    do {                  \
        int _rc = (x);    \
        if (_rc) return _rc; \
    } while (0)

inline long cdiv(long a, long b) { return (a + b - 1) / b; }

// split the contraction when the output alone cannot fill 256 CUs
int pick_ksplit(int outM, int outN, int kc, int batch) {
    const long tiles = cdiv(outM, 64) * cdiv(outN, 64) * batch;
    if (tiles >= 512 || kc < 256) return 1;
    long ks = tiles >= 256 ? std::min<long>(cdiv(768, tiles), kc / 512)      // long skinny K loops (M = B heads GEMM)
                           : std::min<long>(cdiv(512, tiles), kc / 128);
    // tiny outputs (dW1 = 64x3 over 32768 rows) take more, shorter slabs: the reduce kernel sums 16 slabs per round trip
    ks = std::max<long>(1, std::min<long>(ks, (long)outM * outN * batch <= 8192 ? 128 : 64));
    while (ks > 1 && (long)outM * outN * batch * ks > kSplitWs) --ks;
    return (int)ks;
}

struct Op {
    hipStream_t s;
    float* splitws;
    const int* dyn = nullptr;   // device-side row count of the activation matrices (compacted critical rows), or NULL
    // Y(MxN, ldy) = act(X(MxK, ldx) W(NxK)^T + b)        [batched: strides in floats, 0 = shared]
    int lin_fwd(const float* X, long sXz, int ldx, const float* W, long sWz, const float* b, long sbz, float* Y, long sYz,
                int ldy, int M, int N, int K, int batch, bool relu) const {
        HpGemmDesc d{};
        d.A = X; d.sAz = sXz; d.sAi = ldx; d.sAk = 1;
        d.B = W; d.sBz = sWz; d.sBk = 1; d.sBj = K;
        d.C = Y; d.sCz = sYz; d.ldc = ldy;
        d.bias = b; d.sBiasz = sbz;
        d.M = M; d.N = N; d.K = K; d.batch = batch;
        d.flags = (b ? HP_GEMM_BIAS : 0) | (relu ? HP_GEMM_RELU : 0);
        if (splitws) {
            d.ksplit = pick_ksplit(M, N, K, batch);
            d.ws = splitws;
        }
        if (dyn) {
            d.dyn_count = dyn;
            d.dyn_kind = 1;
            d.ksplit = 1;
        }
        return hp_gemm_f32(&d, s);
    }
    // dX(MxK, ldx) = [add +] dY(MxN, ldy) W(NxK), optionally * (mask > 0)
    int lin_dx(const float* dY, long sdYz, int ldy, const float* W, long sWz, float* dX, long sdXz, int ldx, int M, int N,
               int K, int batch, const float* mask, long sMz, int ldm, const float* add, int ldadd) const {
        HpGemmDesc d{};
        d.A = dY; d.sAz = sdYz; d.sAi = ldy; d.sAk = 1;
        d.B = W; d.sBz = sWz; d.sBk = K; d.sBj = 1;
        d.C = dX; d.sCz = sdXz; d.ldc = ldx;
        d.mask = mask; d.sMaskz = sMz; d.ldmask = ldm;
        d.add = add; d.sAddz = sdXz; d.ldadd = ldadd;
        d.M = M; d.N = K; d.K = N; d.batch = batch;
        d.flags = (mask ? HP_GEMM_MASK : 0) | (add ? HP_GEMM_ADD : 0);
        d.ksplit = pick_ksplit(M, K, N, batch);
        d.ws = splitws;
        if (dyn) {
            d.dyn_count = dyn;
            d.dyn_kind = 1;
            d.ksplit = 1;
        }
        return hp_gemm_f32(&d, s);
    }
    // dW(NxK) = dY(MxN, ldy)^T X(MxK, ldx)     (contraction over the M rows);  db(N) = column sums of dY ride along
    int lin_dw(const float* dY, long sdYz, int ldy, const float* X, long sXz, int ldx, float* dW, long sdWz, int M, int N,
               int K, int batch, float* db = nullptr, long sdbz = 0) const {
        HpGemmDesc d{};
        d.A = dY; d.sAz = sdYz; d.sAi = 1; d.sAk = ldy;
        d.B = X; d.sBz = sXz; d.sBk = ldx; d.sBj = 1;
        d.C = dW; d.sCz = sdWz; d.ldc = K;
        d.M = N; d.N = K; d.K = M; d.batch = batch;
        d.ksplit = pick_ksplit(N, K, M, batch);
        d.ws = splitws;
        if (db) {
            d.flags |= HP_GEMM_ROWSUM;
            d.rsum = db;
            d.sRsumz = sdbz;
            while (d.ksplit > 1 && (long)batch * d.ksplit * ((long)N * K + N) > kSplitWs) --d.ksplit;
        }
        if (dyn) {
            d.dyn_count = dyn;
            d.dyn_kind = 2;
        }
        return hp_gemm_f32(&d, s);
    }
    int colsum(const float* X, long sXz, int ldx, int M, int N, int batch, float* out, long sOz) const {
        // the split-K slab area doubles as the row-slab workspace (stream order keeps the uses apart)
        const bool fits = (long)batch * 32 * N <= kSplitWs;
        return hp_colsum_f32(batch, M, N, X, sXz, ldx, nullptr, 0, 0, out, sOz, fits ? splitws : nullptr, s);
    }
};

// ---------------------------------------------------------------------------------------------
// small kernels
// ---------------------------------------------------------------------------------------------

// g[b,c] = max_n h[b,n,c], arg[b,c] = first n attaining it   (model/encoder.py:45 output.max(dim=2))
__global__ __launch_bounds__(256) void colmax_kernel(const float* __restrict__ h, int Np, int C, float* __restrict__ g,
                                                     int* __restrict__ arg) {
    __shared__ float sv[4][64];
    __shared__ int si[4][64];
    const int b = blockIdx.y, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane;
    float best = -__builtin_inff();
    int bi = 0;
    if (c < C) {
        const float* p = h + (long)b * Np * C + c;
        const int per = (Np + 3) / 4, n0 = w * per, n1 = min(Np, n0 + per);
        for (int n = n0; n < n1; ++n) {
            const float v = p[(long)n * C];
            if (v > best || n == n0) {
                best = v;
                bi = n;
            }
        }
        if (n0 >= n1) best = -__builtin_inff();
    }
    sv[w][lane] = best;
    si[w][lane] = bi;
    __syncthreads();
    if (w == 0 && c < C) {
#pragma unroll
        for (int q = 1; q < 4; ++q)
            if (sv[q][lane] > best) {  // strict: earlier point wins ties
                best = sv[q][lane];
                bi = si[q][lane];
            }
        g[(long)b * C + c] = best;
        arg[(long)b * C + c] = bi;
    }
}

// second stage of the fused max-pool: g[b,c] = max over the cloud's row tiles (ascending, strict >: first row wins)
// blockIdx.z = 1: the second encoder of a paired forward (its partials lie zstride floats further on, its outputs are g1/arg1)
__global__ __launch_bounds__(256) void colmax_tiles_kernel(const float* __restrict__ pm, const int* __restrict__ pi, int tiles,
                                                           int tile_rows, int C, float* __restrict__ g, int* __restrict__ arg,
                                                           long zstride, float* __restrict__ g1, int* __restrict__ arg1) {
    const int b = blockIdx.y, c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    if (blockIdx.z) {
        pm += zstride;
        pi += zstride;
        g = g1;
        arg = arg1;
    }
    const long base = (long)b * tiles * C + c;
    float best = pm[base];
    int bi = pi[base];
    for (int t = 1; t < tiles; ++t) {
        const float v = pm[base + (long)t * C];
        if (v > best) {
            best = v;
            bi = pi[base + (long)t * C];
        }
    }
    (void)tile_rows;
    g[(long)b * C + c] = best;
    arg[(long)b * C + c] = bi;
}

// xc[(b,c), :] = x[b, arg[b,c], :]
__global__ __launch_bounds__(256) void gather_rows3_kernel(const float* __restrict__ x, int Np, const int* __restrict__ arg,
                                                           long rows, int C, float* __restrict__ xc) {
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    if (t >= rows) return;
    const long b = t / C;
    const float* s = x + (b * Np + arg[t]) * 3;
    xc[t * 3 + 0] = s[0];
    xc[t * 3 + 1] = s[1];
    xc[t * 3 + 2] = s[2];
}

// One workgroup per critical row t = (b, c): copies the row's x (3), h1 (64), h2 (128), h3 (256) and — unless c4 is
// NULL: its only reader, the layer-5 kernel, can index the forward's array itself — h4 (512) out of the forward's
// per-point activation arrays (source row b*Np + arg[t]) with 16-byte accesses.
__global__ __launch_bounds__(256) void gather_critical_kernel(int Np, const int* __restrict__ arg, const float* __restrict__ x,
                                                              const float* __restrict__ h1, const float* __restrict__ h2,
                                                              const float* __restrict__ h3, const float* __restrict__ h4,
                                                              float* __restrict__ xc, float* __restrict__ c1,
                                                              float* __restrict__ c2, float* __restrict__ c3,
                                                              float* __restrict__ c4) {
    const long t = blockIdx.x;
    const long src = (t >> 9) * Np + arg[t];
    const int i = threadIdx.x;
    if (i < 16) {
        reinterpret_cast<float4*>(c1 + t * 64)[i] = reinterpret_cast<const float4*>(h1 + src * 64)[i];
    } else if (i < 48) {
        reinterpret_cast<float4*>(c2 + t * 128)[i - 16] = reinterpret_cast<const float4*>(h2 + src * 128)[i - 16];
    } else if (i < 112) {
        reinterpret_cast<float4*>(c3 + t * 256)[i - 48] = reinterpret_cast<const float4*>(h3 + src * 256)[i - 48];
    } else if (i < 240) {
        if (c4) reinterpret_cast<float4*>(c4 + t * 512)[i - 112] = reinterpret_cast<const float4*>(h4 + src * 512)[i - 112];
    } else if (i < 243) {
        xc[t * 3 + (i - 240)] = x[src * 3 + (i - 240)];
    }
}

// ---- critical-row compaction -------------------------------------------------------------------------------------
// Of a cloud's 512 arg-max rows only ~170 are distinct points (tools/crit_unique.py): channels that peak at the same
// point share every activation below the max-pool, and their gradients simply add.  The backward therefore runs on the
// DISTINCT critical points: per cloud the channels are sorted by point (bitonic sort in LDS, 512 keys), each distinct
// point gets a slot, the clouds' slots are packed back to back (`off`), and layers 4..1 see `total` rows — a count that
// exists only on the device (HpGemmDesc::dyn_count).
typedef HpCrit Crit;   // hp_enc_bwd.h

__global__ __launch_bounds__(512) void crit_unique_kernel(const int* __restrict__ arg, Crit c) {
    __shared__ int key[512];
    __shared__ int scan[512];
    const int b = blockIdx.x, t = threadIdx.x;
    key[t] = (arg[(long)b * 512 + t] << 9) | t;
    __syncthreads();
    for (int k = 2; k <= 512; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            const int o = t ^ j;
            if (o > t) {
                const int x = key[t], y = key[o];
                const bool up = (t & k) == 0;
                if ((x > y) == up) {
                    key[t] = y;
                    key[o] = x;
                }
            }
            __syncthreads();
        }
    const int mine = key[t], p = mine >> 9, ch = mine & 511;
    const int flag = (t == 0 || (key[t - 1] >> 9) != p) ? 1 : 0;
    scan[t] = flag;
    __syncthreads();
    for (int d = 1; d < 512; d <<= 1) {   // inclusive scan
        const int v = t >= d ? scan[t - d] : 0;
        __syncthreads();
        scan[t] += v;
        __syncthreads();
    }
    const int u = scan[t] - 1;
    c.chan[(long)b * 512 + t] = ch;
    c.slot[(long)b * 512 + ch] = u;
    if (flag) {
        c.pt[(long)b * 512 + u] = p;
        c.start[(long)b * 513 + u] = t;
    }
    if (t == 511) {
        c.cnt[b] = u + 1;
        c.start[(long)b * 513 + u + 1] = 512;
    }
}

// One WAVE per (cloud, slot), four slots per workgroup: the distinct critical point's x, h1, h2, h3 out of the forward's
// per-point arrays into compact row off[b] + slot (16-byte accesses: 112 per slot in two passes of the wave).
// h1 == NULL: coordinates only (the recompute path).  (One 256-thread workgroup per slot — 32768 launches of which a third
// are live and 112 threads work — was bound by workgroup dispatch: 21 us.)
__global__ __launch_bounds__(256) void crit_gather_kernel(int B, int Np, Crit c, const float* __restrict__ x, const float* __restrict__ h1,
                                                          const float* __restrict__ h2, const float* __restrict__ h3,
                                                          float* __restrict__ xc, float* __restrict__ c1,
                                                          float* __restrict__ c2, float* __restrict__ c3) {
    const int slot = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int b = slot >> 9, u = slot & 511;
    if (u >= c.cnt[b]) return;
    const int i = threadIdx.x & 63;
    // the cloud's first compact row = the number of distinct points of the clouds before it: every wave adds those
    // counts up itself (a 64-lane sum) instead of waiting for a one-workgroup scan launch; the wave of slot 0 publishes
    // the offset for the kernels that follow, the last cloud's also the total (the GEMMs' device-side row count)
    int pre = 0;
    for (int q = i; q < b; q += 64) pre += c.cnt[q];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) pre += __shfl_xor(pre, o, 64);
    if (u == 0 && i == 0) {
        c.off[b] = pre;
        if (b == B - 1) c.total[0] = pre + c.cnt[b];
    }
    const long t = pre + u;
    const long src = (long)b * Np + c.pt[(long)b * 512 + u];
    if (h1) {
        reinterpret_cast<float4*>(c3 + t * 256)[i] = reinterpret_cast<const float4*>(h3 + src * 256)[i];
        if (i < 16) reinterpret_cast<float4*>(c1 + t * 64)[i] = reinterpret_cast<const float4*>(h1 + src * 64)[i];
        else if (i < 48) reinterpret_cast<float4*>(c2 + t * 128)[i - 16] = reinterpret_cast<const float4*>(h2 + src * 128)[i - 16];
    }
    if (i >= 48 && i < 51) xc[t * 3 + (i - 48)] = x[src * 3 + (i - 48)];
}

// delta4 of a distinct critical point = (its h4 > 0) * sum over the channels that peak there of dg[b,c] * W5[c,:], channels
// in ascending order.  One workgroup (128 lanes x 4 consecutive k) per (cloud, slot).  h4: the forward's full array
// (rows b*Np + point) when `full`, else the compact recomputed rows.
__global__ __launch_bounds__(128) void crit_l5_dx_kernel(int Np, Crit c, const float* __restrict__ dg, const float* __restrict__ W5,
                                                         const float* __restrict__ h4, int full, float* __restrict__ d4) {
    const int b = blockIdx.x >> 9, u = blockIdx.x & 511;
    if (u >= c.cnt[b]) return;
    const long t = c.off[b] + u;
    const long hrow = full ? (long)b * Np + c.pt[(long)b * 512 + u] : t;
    const int k = threadIdx.x * 4;
    const int i0 = c.start[(long)b * 513 + u], i1 = c.start[(long)b * 513 + u + 1];
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int i = i0; i < i1; ++i) {
        const int ch = c.chan[(long)b * 512 + i];
        const float g = dg[(long)b * 512 + ch];
        const float4 w = *reinterpret_cast<const float4*>(W5 + (long)ch * 512 + k);
        acc.x = __builtin_fmaf(g, w.x, acc.x);
        acc.y = __builtin_fmaf(g, w.y, acc.y);
        acc.z = __builtin_fmaf(g, w.z, acc.z);
        acc.w = __builtin_fmaf(g, w.w, acc.w);
    }
    const float4 hv = *reinterpret_cast<const float4*>(h4 + hrow * 512 + k);
    float4 o;
    o.x = hv.x > 0.f ? acc.x : 0.f;
    o.y = hv.y > 0.f ? acc.y : 0.f;
    o.z = hv.z > 0.f ? acc.z : 0.f;
    o.w = hv.w > 0.f ? acc.w : 0.f;
    *reinterpret_cast<float4*>(d4 + t * 512 + k) = o;
}

// Layer-5 backward on the critical rows (one-hot upstream):
//   dW5[c,k]       = sum_b dg[b,c] * h4c[(b,c),k]
//   d4[(b,c),k]    = dg[b,c] * W5[c,k] * (h4c[(b,c),k] > 0)
//   db5[c]         = sum_b dg[b,c]
// One workgroup per channel c; K % 4 == 0.  Thread (q = tid & 127, g = tid >> 7) owns 4 consecutive k and the
// clouds b = g, g + 4, ...; the four cloud groups are combined through LDS in a fixed order.
// h4c: the critical rows' h4, row (b,c) at h4c + (b*C + c)*K — or, with arg != NULL, the forward's full h4 with row
// (b,c) at h4c + (b*Np + arg[b*C + c])*K (no gathered copy of the widest activation is ever made).
// With slot/off (critical-row compaction) row (b,c) is compact row off[b] + slot[b*C + c]; d4 == NULL: dW5 / db5 only
// (delta4 of the compacted rows comes from crit_l5_dx_kernel).
__global__ __launch_bounds__(512) void enc_l5_bwd_kernel(int B, int C, int K, const float* __restrict__ dg,
                                                         const float* __restrict__ W5, const float* __restrict__ h4c,
                                                         const int* __restrict__ arg, int Np, const int* __restrict__ slot,
                                                         const int* __restrict__ off, float* __restrict__ dW5,
                                                         float* __restrict__ d4, float* __restrict__ db5) {
    __shared__ float4 red[3][128];
    const int c = blockIdx.x;
    const int q = threadIdx.x & 127, g = threadIdx.x >> 7;   // g = 0..3: clouds b = g, g + 4, ...
    if (threadIdx.x == 0) {
        float s = 0.f;
        for (int b = 0; b < B; ++b) s += dg[(long)b * C + c];
        db5[c] = s;
    }
    for (int k = q * 4; k < K; k += 512) {
        const float4 w = *reinterpret_cast<const float4*>(W5 + (long)c * K + k);
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 4
        for (int b = g; b < B; b += 4) {
            const long row = (long)b * C + c;
            const float gbc = dg[row];
            const long src = arg ? (long)b * Np + arg[row] : (slot ? (long)off[b] + slot[row] : row);
            const float4 hv = *reinterpret_cast<const float4*>(h4c + src * K + k);
            s.x = __builtin_fmaf(gbc, hv.x, s.x);
            s.y = __builtin_fmaf(gbc, hv.y, s.y);
            s.z = __builtin_fmaf(gbc, hv.z, s.z);
            s.w = __builtin_fmaf(gbc, hv.w, s.w);
            if (d4) {
                float4 o;
                o.x = hv.x > 0.f ? gbc * w.x : 0.f;
                o.y = hv.y > 0.f ? gbc * w.y : 0.f;
                o.z = hv.z > 0.f ? gbc * w.z : 0.f;
                o.w = hv.w > 0.f ? gbc * w.w : 0.f;
                *reinterpret_cast<float4*>(d4 + row * K + k) = o;
            }
        }
        if (g) red[g - 1][q] = s;
        __syncthreads();
        if (g == 0) {   // the four cloud groups in a fixed order
            const float4 t1 = red[0][q], t2 = red[1][q], t3 = red[2][q];
            *reinterpret_cast<float4*>(dW5 + (long)c * K + k) =
                make_float4(((s.x + t1.x) + t2.x) + t3.x, ((s.y + t1.y) + t2.y) + t3.y, ((s.z + t1.z) + t2.z) + t3.z,
                            ((s.w + t1.w) + t2.w) + t3.w);
        }
        __syncthreads();
    }
}

// VAE head (model/encoder.py:38-41,49-51): z = eps*exp(lv) + mu ; returned "logvar" = exp(lv)
// z may be a column block of a wider matrix (the latent [z | real mu] of a paired forward): row stride z_ld, `out` columns
__global__ __launch_bounds__(256) void vae_head_fwd_kernel(long n, const float* __restrict__ eps, const float* __restrict__ mu,
                                                           const float* __restrict__ lv, float* __restrict__ z, int out, int z_ld,
                                                           float* __restrict__ explv) {
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    if (t >= n) return;
    const float e = expf(lv[t]);   // accurate exp: z and the returned exp(logvar) are parity outputs
    explv[t] = e;
    z[(t / out) * z_ld + t % out] = __builtin_fmaf(eps[t], e, mu[t]);
}

// d mu = gz + gmu ; d lv = (gz*eps + gexplv) * exp(lv)
__global__ __launch_bounds__(256) void vae_head_bwd_kernel(long n, const float* __restrict__ eps, const float* __restrict__ lv,
                                                           const float* __restrict__ gz, int out, int gz_ld,
                                                           const float* __restrict__ gmu, const float* __restrict__ gexplv,
                                                           float* __restrict__ dmu, float* __restrict__ dlv) {
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    if (t >= n) return;
    const float a = gz ? gz[(t / out) * gz_ld + t % out] : 0.f;
    dmu[t] = a + (gmu ? gmu[t] : 0.f);
    dlv[t] = (a * eps[t] + (gexplv ? gexplv[t] : 0.f)) * expf(lv[t]);
}

// activations h1..h5 of every point, then the split area of conv_split.hip (f16 weight pieces, exponents, activation maxima)
long enc_fwd_ws(long B, long Np) { return B * Np * (64 + 128 + 256 + 512 + 512) + hp_conv_split_area_floats(B * Np); }
long enc_bwd_ws(long B, long out) {
    const long Rc = B * 512;
    return Rc * 4 + Rc * (64 + 128 + 256 + 512) * 2 + B * (2 * out + 4 * 512) + kSplitWs + 64 + (B * (5 * 512 + 4) + 16) +
           Rc * 21 + B * 128 + HP_EB_WT_BYTES / 4 + HP_EB_WT_US_FLOATS + 32;      // (row maxima, row masks, weight stream: enc_bwd_f16.hip)
}


// ---- skinny-M layer programs (skinny.hip): shared helpers ---------------------------------------------------------
int env_int(const char* name, int dflt) {
    const char* e = std::getenv(name);
    return e && *e ? std::atoi(e) : dflt;
}
// contraction ranges of a layer: a power of two, <= smax, each range at least 32 long, and no more tasks than CUs
int sk_ranges(int contraction, int out_blocks, int smax) {
    int s = 1;
    while (2 * s <= smax && contraction / (2 * s) >= 32 && contraction % (2 * s) == 0 && out_blocks * 2 * s <= 256) s *= 2;
    return s;
}
inline long up4(long v) { return (v + 3) / 4 * 4; }


// model/encoder.py:46-53 after the max-pool: f = ReLU(fc g) ; mu = mu_layer f ; lv = std_layer f.  Three launches (the fc
// layer in k-ranges whose slabs the mu/std tasks finish on load, writing f on the way; the heads in k-ranges again; a
// finishing launch) instead of 3 un-split GEMMs of ~19 us each.  n encoders (1, or the 2 of a paired forward) share the
// launches: their ops sit in the same phases.  slabs[e]: 4*64*512 + 8*64*out_size floats.  -2: the shapes do not fit.
struct EncTail {
    const float* g;
    const HpEncoderWeights* w;
    int is_vae;
    float *f, *mu, *lv, *slabs;
    int mu_ld;   // row stride of mu (a plain encoder's mu may be a column block of the latent)
};
int enc_tail_forward_skinny(int B, int out_size, int n, const EncTail* t, hipStream_t stream) {
    if (B > 64 || out_size % 32) return -2;
    HpSkProgram pr{};
    const int S = sk_ranges(512, 512 / 32, 4);
    const int Sh = sk_ranges(512, out_size / 32, 4);
    if (S == 1) return -2;
    for (int e = 0; e < n; ++e) {
        HpSkOp& fc = pr.op[pr.nops++];
        fc.type = HP_SK_F; fc.phase = 0;
        fc.a.p = t[e].g; fc.a.S = 1; fc.a.ld = 512;
        fc.w = t[e].w->fc_w; fc.w_ld = 512;
        fc.M = B; fc.N = 512; fc.K = 512; fc.CL = 512 / S;
        fc.out = t[e].slabs; fc.out_slab = 64L * 512; fc.out_ld = 512;
    }
    // mu / std heads: 4 strips each — k-ranges again (16 tasks of one memory latency instead of 4 tasks of eight), their
    // slabs finished by a last launch
    for (int e = 0; e < n; ++e) {
        const HpEncoderWeights* w = t[e].w;
        HpSkSrc fsrc{};
        fsrc.p = t[e].slabs; fsrc.slab = 64L * 512; fsrc.S = S; fsrc.ld = 512; fsrc.bias = w->fc_b; fsrc.relu = 1;
        fsrc.mat = t[e].f; fsrc.ldmat = 512;
        float* hs = t[e].slabs + (long)S * 64 * 512;
        for (int hd = 0; hd < (t[e].is_vae ? 2 : 1); ++hd) {
            HpSkOp& op = pr.op[pr.nops++];
            op.type = HP_SK_F; op.phase = 1;
            op.a = fsrc;
            if (hd) op.a.mat = nullptr;          // f is written once, by the mu tasks
            op.w = hd ? w->std_w : w->mu_w; op.w_ld = 512;
            op.M = B; op.N = out_size; op.K = 512; op.CL = 512 / Sh;
            op.out_ld = out_size;
            if (Sh == 1) {
                op.out = hd ? t[e].lv : t[e].mu; op.out_bias = hd ? w->std_b : w->mu_b;
                if (!hd) op.out_ld = t[e].mu_ld;
            } else {
                op.out = hs + (long)hd * Sh * 64 * out_size; op.out_slab = 64L * out_size;
            }
        }
    }
    for (int e = 0; e < n && Sh > 1; ++e) {
        const HpEncoderWeights* w = t[e].w;
        float* hs = t[e].slabs + (long)S * 64 * 512;
        for (int hd = 0; hd < (t[e].is_vae ? 2 : 1); ++hd) {
            HpSkOp& op = pr.op[pr.nops++];
            op.type = HP_SK_FIN; op.phase = 2;
            op.a.p = hs + (long)hd * Sh * 64 * out_size; op.a.slab = 64L * out_size; op.a.S = Sh; op.a.ld = out_size;
            op.a.bias = hd ? w->std_b : w->mu_b;
            op.out = hd ? t[e].lv : t[e].mu; op.out_ld = hd ? out_size : t[e].mu_ld;
            op.M = B; op.N = out_size; op.K = 1;
        }
    }
    return hp_skinny_run(&pr, stream);
}

// the autograd of that tail: dmu (and dlv) -> d mu_w/b, d std_w/b, dfc = (dmu mu_w + dlv std_w) * (f > 0), d fc_w/b,
// dg = dfc fc_w.  Three launches instead of 7-8, shared by the n encoders (1, or the 2 of a pair: their ops sit in the same
// phases; per encoder the tasks are those of a single-encoder program).  slabs[e]: 8*64*512 floats.
struct EncTailBwd {
    const float *g, *f;
    const HpEncoderWeights* w;
    const float* dmu;
    int dmu_ld;
    const float* dlv;   // NULL: plain encoder
    const HpEncoderGrads* gr;
    float *dfc, *dg, *slabs;
};
int enc_tail_backward_skinny(int B, int out_size, int n, const EncTailBwd* t, hipStream_t stream) {
    if (B > 64 || out_size % 32) return -2;
    HpSkProgram pr{};
    const int S2 = sk_ranges(512, 512 / 32, 4);
    if (S2 == 1) return -2;
    int S[2], nh[2];
    for (int e = 0; e < n; ++e) {
        nh[e] = t[e].dlv ? 2 : 1;
        S[e] = sk_ranges(out_size, 512 / 32, 4 / nh[e]);      // the two heads' ranges land in ONE slab set (<= 4 slabs)
        if (nh[e] * S[e] == 1) return -2;                      // (a single range would apply no mask: not built)
    }
    for (int e = 0; e < n; ++e) {
        for (int hd = 0; hd < nh[e]; ++hd) {
            HpSkOp& op = pr.op[pr.nops++];
            op.type = HP_SK_X; op.phase = 0;
            op.a.p = hd ? t[e].dlv : t[e].dmu; op.a.S = 1; op.a.ld = hd ? out_size : t[e].dmu_ld;
            op.w = hd ? t[e].w->std_w : t[e].w->mu_w; op.w_ld = 512;
            op.M = B; op.N = out_size; op.K = 512; op.CL = out_size / S[e];
            op.out = t[e].slabs + (long)hd * S[e] * 64 * 512; op.out_slab = 64L * 512; op.out_ld = 512;
        }
        for (int hd = 0; hd < nh[e]; ++hd) {
            HpSkOp& op = pr.op[pr.nops++];
            op.type = HP_SK_W; op.phase = 0;
            op.a.p = hd ? t[e].dlv : t[e].dmu; op.a.S = 1; op.a.ld = hd ? out_size : t[e].dmu_ld;
            op.w = t[e].f; op.w_ld = 512;
            op.out = hd ? t[e].gr->std_w : t[e].gr->mu_w; op.out_ld = 512; op.rsum = hd ? t[e].gr->std_b : t[e].gr->mu_b;
            op.M = B; op.N = out_size; op.K = 512;
        }
    }
    for (int e = 0; e < n; ++e) {
        float* s2 = t[e].slabs + (long)nh[e] * S[e] * 64 * 512;
        HpSkOp& xf = pr.op[pr.nops++];
        xf.type = HP_SK_X; xf.phase = 1;
        xf.a.p = t[e].slabs; xf.a.slab = 64L * 512; xf.a.S = nh[e] * S[e]; xf.a.ld = 512; xf.a.mask = t[e].f; xf.a.ldm = 512;
        xf.a.mat = t[e].dfc; xf.a.ldmat = 512;
        xf.w = t[e].w->fc_w; xf.w_ld = 512;
        xf.M = B; xf.N = 512; xf.K = 512; xf.CL = 512 / S2;
        xf.out = s2; xf.out_slab = 64L * 512; xf.out_ld = 512;
    }
    for (int e = 0; e < n; ++e) {
        float* s2 = t[e].slabs + (long)nh[e] * S[e] * 64 * 512;
        HpSkOp& fin = pr.op[pr.nops++];
        fin.type = HP_SK_FIN; fin.phase = 2;
        fin.a.p = s2; fin.a.slab = 64L * 512; fin.a.S = S2; fin.a.ld = 512;
        fin.out = t[e].dg; fin.out_ld = 512;
        fin.M = B; fin.N = 512; fin.K = 1;
        HpSkOp& wf = pr.op[pr.nops++];
        wf.type = HP_SK_W; wf.phase = 2;
        wf.a.p = t[e].dfc; wf.a.S = 1; wf.a.ld = 512;
        wf.w = t[e].g; wf.w_ld = 512;
        wf.out = t[e].gr->fc_w; wf.out_ld = 512; wf.rsum = t[e].gr->fc_b;
        wf.M = B; wf.N = 512; wf.K = 512;
    }
    return hp_skinny_run(&pr, stream);
}
}  // namespace

// =================================================================================================
// Encoder
// =================================================================================================
HP_API long hp_encoder_forward_workspace_floats(int B, int Np) { return enc_fwd_ws(B, Np); }
HP_API long hp_encoder_backward_workspace_floats(int B, int out_size) { return enc_bwd_ws(B, out_size); }

// model/encoder.py:43-53.  x (B,Np,3) contiguous (the layout as loaded; the reference's in-place
// transpose to (B,3,Np) is a view change only).  Outputs: argidx/g (B,512), f (B,512), mu (B,out);
// VAE: lv (raw std_layer output), z, explv (= exp(lv), what the reference returns as "logvar").
namespace {
// n = 1: one encoder.  n = 2: the two encoders of a HyperPocket step (model/full_model.py:106-112: same conv stack, own
// weights, own inputs) — every conv layer is ONE batched launch over both (z = 0, 1; strides = the distances between the
// two encoders' buffers): twice the tiles per launch instead of two launches that each pay the ~23 us of ramp-up and
// tail a wide GEMM launch costs (tools/pair_probe.py: 0.958 ms batched against 0.998 ms back to back and 1.004 ms on two
// streams for the two conv stacks).  Per row the arithmetic is that of the single-encoder launch.
int encoder_forward_impl(int B, int Np, int out_size, int n, const HpEncoderIO* io, hipStream_t stream) {
    const long R = (long)B * Np;
    Op op{stream, nullptr};
    const HpEncoderIO& e0 = io[0];
    const HpEncoderIO& e1 = io[n - 1];
    auto dz = [&](const float* a0, const float* a1) { return n > 1 ? (long)(a1 - a0) : 0L; };
    float* h[6];
    h[0] = nullptr;
    h[1] = e0.ws;
    for (int l = 2; l <= 5; ++l) h[l] = h[l - 1] + R * kEnc[l - 1];
    const long sWs = dz(e0.ws, e1.ws);
    // The conv stack: split-f16 matrix-pipe layers (conv_split.hip) unless switched off (HP_CONV_SPLIT=0 / hp_conv_split_set),
    // else the fp32 MFMA GEMMs.
    const bool split = hp_conv_split_enabled();
    float* area = h[5] + R * 512;
    // Round 4: with whole 128-row tiles per cloud the activations are stored already split ("P-format", conv_pp.hip) and both
    // operands of layers 2..5 are DMA-staged; a word in the split area tells the backward's readers which format h1..h4 hold.
    const bool presplit = split && hp_conv_presplit_enabled() && Np % 128 == 0;
    if (split) {
        const float* W0[4] = {e0.w->conv_w[1], e0.w->conv_w[2], e0.w->conv_w[3], e0.w->conv_w[4]};
        const float* W1[4] = {e1.w->conv_w[1], e1.w->conv_w[2], e1.w->conv_w[3], e1.w->conv_w[4]};
        TRY(hp_conv_split_prep(n, W0, W1, area, sWs, R, presplit ? HP_PP_FMT_P : HP_PP_FMT_F32, stream));
    } else {
        TRY(hp_conv_pp_mark(n, area, sWs, R, HP_PP_FMT_F32, stream));
    }
    if (presplit) {
        TRY(hp_conv_pp_layer1(n, e0.x, dz(e0.x, e1.x), e0.w->conv_w[0], dz(e0.w->conv_w[0], e1.w->conv_w[0]), e0.w->conv_b[0],
                              dz(e0.w->conv_b[0], e1.w->conv_b[0]), h[1], area, sWs, R, stream));
        for (int l = 2; l <= 4; ++l)
            TRY(hp_conv_pp_layer(l, n, h[l - 1], e0.w->conv_b[l - 1], dz(e0.w->conv_b[l - 1], e1.w->conv_b[l - 1]), h[l], area, sWs, R,
                                 nullptr, nullptr, 0, stream));
    } else if (split) {
        TRY(hp_conv_split_layer1(n, e0.x, dz(e0.x, e1.x), e0.w->conv_w[0], dz(e0.w->conv_w[0], e1.w->conv_w[0]), e0.w->conv_b[0],
                                 dz(e0.w->conv_b[0], e1.w->conv_b[0]), h[1], sWs, area, sWs, R, stream));
        for (int l = 2; l <= 4; ++l)
            TRY(hp_conv_split_layer(l, n, h[l - 1], sWs, e0.w->conv_b[l - 1], dz(e0.w->conv_b[l - 1], e1.w->conv_b[l - 1]), h[l], sWs,
                                    area, sWs, R, 1, 0, nullptr, nullptr, 0, stream));
    } else {
        const float* in = e0.x;
        long sIn = dz(e0.x, e1.x);
        for (int l = 1; l <= 4; ++l) {
            TRY(op.lin_fwd(in, sIn, kEnc[l - 1], e0.w->conv_w[l - 1], dz(e0.w->conv_w[l - 1], e1.w->conv_w[l - 1]),
                           e0.w->conv_b[l - 1], dz(e0.w->conv_b[l - 1], e1.w->conv_b[l - 1]), h[l], sWs, kEnc[l], (int)R, kEnc[l],
                           kEnc[l - 1], n, true));
            in = h[l];
            sIn = sWs;
        }
    }
    // layer 5 (no ReLU) + max over points.  When a cloud's points are whole row tiles the max-pool is fused into the
    // GEMM epilogue: h5 (B*Np x 512) is never written; its slot in the workspace holds the per-tile partials.
    HpGemmDesc d5{};
    d5.A = h[4]; d5.sAz = sWs; d5.sAi = 512; d5.sAk = 1;
    d5.B = e0.w->conv_w[4]; d5.sBz = dz(e0.w->conv_w[4], e1.w->conv_w[4]); d5.sBk = 1; d5.sBj = 512;
    d5.bias = e0.w->conv_b[4]; d5.sBiasz = dz(e0.w->conv_b[4], e1.w->conv_b[4]);
    d5.sCz = sWs;
    d5.M = (int)R; d5.N = 512; d5.K = 512; d5.batch = n;
    d5.flags = HP_GEMM_BIAS | HP_GEMM_COLMAX;
    d5.group_rows = Np;
    const int tr = split ? 128 : hp_gemm_tile_rows(&d5);
    long tail_off = -1;
    if (tr > 0 && Np % tr == 0) {
        const long tiles = R / tr;
        d5.cmax = h[5];
        d5.cidx = reinterpret_cast<int*>(h[5] + tiles * 512);
        if (R * 512 - up4(2 * tiles * 512) >= 4L * 64 * 512 + 8L * 64 * out_size) tail_off = up4(2 * tiles * 512);
        if (presplit)
            TRY(hp_conv_pp_layer(5, n, h[4], d5.bias, d5.sBiasz, nullptr, area, sWs, R, d5.cmax, d5.cidx, Np, stream));
        else if (split)
            TRY(hp_conv_split_layer(5, n, h[4], sWs, d5.bias, d5.sBiasz, nullptr, sWs, area, sWs, R, 0, 1, d5.cmax, d5.cidx, Np, stream));
        else
            TRY(hp_gemm_f32(&d5, stream));
        hipLaunchKernelGGL(colmax_tiles_kernel, dim3(2, B, n), dim3(256), 0, stream, d5.cmax, d5.cidx, Np / tr, tr, 512, io[0].g,
                           io[0].argidx, sWs, io[n - 1].g, io[n - 1].argidx);
    } else {
        if (split)
            TRY(hp_conv_split_layer(5, n, h[4], sWs, d5.bias, d5.sBiasz, h[5], sWs, area, sWs, R, 0, 0, nullptr, nullptr, 0, stream));
        else
            TRY(op.lin_fwd(h[4], sWs, 512, e0.w->conv_w[4], d5.sBz, e0.w->conv_b[4], d5.sBiasz, h[5], sWs, 512, (int)R, 512, 512, n,
                           false));
        for (int z = 0; z < n; ++z)
            hipLaunchKernelGGL(colmax_kernel, dim3(512 / 64, B), dim3(256), 0, stream, h[5] + z * sWs, Np, 512, io[z].g,
                               io[z].argidx);
    }
    // the fc / mu / std tails: skinny layer launches shared by the encoders when the shapes allow (the h5 slot of the
    // workspace is free behind the fused max-pool's per-tile partials: it holds the slabs), else un-split GEMMs
    int sk = -2;
    if (hp_skinny_enabled() && tail_off >= 0 && B <= 64) {
        EncTail t[2];
        for (int z = 0; z < n; ++z) t[z] = EncTail{io[z].g, io[z].w, io[z].is_vae, io[z].f, io[z].mu, io[z].lv, h[5] + z * sWs + tail_off,
                           (!io[z].is_vae && io[z].out_ld > 0) ? io[z].out_ld : out_size};
        sk = enc_tail_forward_skinny(B, out_size, n, t, stream);
    }
    if (sk != -2) TRY(sk);
    for (int z = 0; z < n; ++z) {
        const HpEncoderIO& e = io[z];
        if (sk == -2) {
            TRY(op.lin_fwd(e.g, 0, 512, e.w->fc_w, 0, e.w->fc_b, 0, e.f, 0, 512, B, 512, 512, 1, true));
            TRY(op.lin_fwd(e.f, 0, 512, e.w->mu_w, 0, e.w->mu_b, 0, e.mu, 0, (!e.is_vae && e.out_ld > 0) ? e.out_ld : out_size, B,
                           out_size, 512, 1, false));
            if (e.is_vae)
                TRY(op.lin_fwd(e.f, 0, 512, e.w->std_w, 0, e.w->std_b, 0, e.lv, 0, out_size, B, out_size, 512, 1, false));
        }
        if (e.is_vae) {
            const long nel = (long)B * out_size;
            hipLaunchKernelGGL(vae_head_fwd_kernel, dim3((int)cdiv(nel, 256)), dim3(256), 0, stream, nel, e.eps, e.mu, e.lv, e.z,
                               out_size, e.out_ld > 0 ? e.out_ld : out_size, e.explv);
        }
    }
    HP_RETURN_LAST_ERROR();
}

bool encoder_io_ok(const HpEncoderIO& e) {
    return e.x && e.w && e.argidx && e.g && e.f && e.mu && e.ws &&
           (!e.is_vae || (e.eps && e.lv && e.z && e.explv && e.w->std_w && e.w->std_b));
}
}  // namespace

HP_API int hp_encoder_forward(int B, int Np, const float* x, const HpEncoderWeights* w, int out_size, int is_vae,
                              const float* eps, int* argidx, float* g, float* f, float* mu, float* lv, float* z,
                              float* explv, float* ws, hipStream_t stream) {
    HP_CHECK_ARG(B > 0 && Np > 0 && out_size > 0 && x && w && argidx && g && f && mu && ws);
    HP_CHECK_ARG(!is_vae || (eps && lv && z && explv && w->std_w && w->std_b));
    HP_CHECK_ARG(B <= 65535);
    HP_CHECK_ARG((long)B * Np < (1L << 31));
    const HpEncoderIO io{x, w, eps, argidx, g, f, mu, lv, z, explv, ws, is_vae, 0};
    return encoder_forward_impl(B, Np, out_size, 1, &io, stream);
}

// Both encoders of a HyperPocket step in one call: io[0], io[1] as hp_encoder_forward's arguments; same B, Np, out_size.
// The two workspaces (hp_encoder_forward_workspace_floats each) may lie anywhere; results are those of two
// hp_encoder_forward calls.
// h1..h4 inside a forward workspace as fp32 rows (B*Np, 64 | 128 | 256 | 512), whatever format the forward left them in
// (round 4's P-format is converted in place; otherwise nothing happens).  What the layered backward does first; the tests' view.
HP_API int hp_encoder_workspace_to_f32(int B, int Np, float* ws, hipStream_t stream) {
    HP_CHECK_ARG(B > 0 && Np > 0 && ws);
    return hp_conv_pp_unpack_ws(ws, (long)B * Np, stream);
}

HP_API int hp_encoder_forward_pair(int B, int Np, int out_size, const HpEncoderIO* io, hipStream_t stream) {
    HP_CHECK_ARG(B > 0 && Np > 0 && out_size > 0 && io && encoder_io_ok(io[0]) && encoder_io_ok(io[1]));
    HP_CHECK_ARG(B <= 32767 && (long)B * Np < (1L << 31));
    return encoder_forward_impl(B, Np, out_size, 2, io, stream);
}

namespace {
struct EncBwdWs {
    float* xc;
    float* hc[5];
    float* dl[5];
    float *dmu, *dlv, *tmp, *dfc, *dg, *split;
    Crit crit;
    float *d4max, *hmask, *hmax, *bexp, *wt, *wt_us;      // enc_bwd_f16.hip
};
EncBwdWs enc_bwd_layout(float* ws, long B, long out_size) {
    const long Rc = B * 512;
    float* p = ws;
    auto take = [&](long n) { float* r = p; p += (n + 3) / 4 * 4; return r; };
    EncBwdWs L;
    L.xc = take(Rc * 3);
    for (int l = 1; l <= 4; ++l) L.hc[l] = take(Rc * kEnc[l]);
    for (int l = 1; l <= 4; ++l) L.dl[l] = take(Rc * kEnc[l]);
    L.dmu = take(B * out_size);
    L.dlv = take(B * out_size);
    L.tmp = take(B * 512);
    L.dfc = take(B * 512);
    L.dg = take(B * 512);
    L.split = take(kSplitWs);
    int* ip = reinterpret_cast<int*>(take(B * (5 * 512 + 4) + 16));
    L.crit.chan = ip;
    L.crit.start = L.crit.chan + B * 512;
    L.crit.pt = L.crit.start + B * 513;
    L.crit.slot = L.crit.pt + B * 512;
    L.crit.eslot = L.crit.slot + B * 512;
    L.crit.cnt = L.crit.eslot + B * 512;
    L.crit.off = L.crit.cnt + B;
    L.crit.total = L.crit.off + B;
    L.d4max = take(Rc);
    L.hmask = take(Rc * 16);
    L.hmax = take(Rc * 4);
    L.bexp = take(B * 16 * 8);
    L.wt = take(HP_EB_WT_BYTES / 4);
    L.wt_us = take(HP_EB_WT_US_FLOATS);
    return L;
}
}  // namespace

// Activations h1..h4 of the B*512 critical rows (arg-max points) into `ws` (the hp_encoder_backward workspace).
// With the forward's workspace at hand (`fwd_ws`, h1..h4 of every point still in place) they are copied out of it —
// one pass over 3.8 KB per critical row; without it (fwd_ws == NULL: the caller dropped the 15 KB/point forward
// workspace) they are recomputed from the gathered coordinates, the same GEMM chain on B*512 rows.
static int enc_critical_rows(int B, int Np, const float* x, const HpEncoderWeights* w, int out_size, const int* argidx,
                             const float* fwd_ws, float* ws, int dedup, hipStream_t stream) {
    const long Rc = (long)B * 512;
    EncBwdWs L = enc_bwd_layout(ws, B, out_size);
    if (dedup) {
        // distinct critical points only, packed back to back; their number stays on the device (L.crit.total)
        hipLaunchKernelGGL(crit_unique_kernel, dim3(B), dim3(512), 0, stream, argidx, L.crit);
        const long R = (long)B * Np;
        const float* h1 = fwd_ws;
        const float* h2 = fwd_ws ? h1 + R * 64 : nullptr;
        const float* h3 = fwd_ws ? h2 + R * 128 : nullptr;
        hipLaunchKernelGGL(crit_gather_kernel, dim3((unsigned)(Rc / 4)), dim3(256), 0, stream, B, Np, L.crit, x, h1, h2, h3, L.xc, L.hc[1],
                           L.hc[2], L.hc[3]);
        if (!fwd_ws) {
            Op op{stream, nullptr, L.crit.total};
            const float* in = L.xc;
            for (int l = 1; l <= 4; ++l) {
                TRY(op.lin_fwd(in, 0, kEnc[l - 1], w->conv_w[l - 1], 0, w->conv_b[l - 1], 0, L.hc[l], 0, kEnc[l], (int)Rc, kEnc[l],
                               kEnc[l - 1], 1, true));
                in = L.hc[l];
            }
        }
        HP_RETURN_LAST_ERROR();
    }
    if (fwd_ws) {
        const long R = (long)B * Np;
        const float* h1 = fwd_ws;
        const float* h2 = h1 + R * 64;
        const float* h3 = h2 + R * 128;
        const float* h4 = h3 + R * 256;
        hipLaunchKernelGGL(gather_critical_kernel, dim3((unsigned)Rc), dim3(256), 0, stream, Np, argidx, x, h1, h2, h3, h4, L.xc,
                           L.hc[1], L.hc[2], L.hc[3], (float*)nullptr);
        HP_RETURN_LAST_ERROR();
    }
    Op op{stream, nullptr};
    hipLaunchKernelGGL(gather_rows3_kernel, dim3((int)cdiv(Rc, 256)), dim3(256), 0, stream, x, Np, argidx, Rc, 512, L.xc);
    const float* in = L.xc;
    for (int l = 1; l <= 4; ++l) {
        TRY(op.lin_fwd(in, 0, kEnc[l - 1], w->conv_w[l - 1], 0, w->conv_b[l - 1], 0, L.hc[l], 0, kEnc[l], (int)Rc, kEnc[l],
                       kEnc[l - 1], 1, true));
        in = L.hc[l];
    }
    HP_RETURN_LAST_ERROR();
}

// Gradients of every encoder parameter.  grad_out: d/d z (VAE) or d/d mu (plain), a column block of a matrix with row stride
// grad_out_ld >= out_size (the paired forward's latent [z | real mu] hands each encoder its half of d latent without a
// copy); grad_mu / grad_explv: direct gradients on the VAE's mu / exp(logvar) outputs (KLD term), may be NULL.  fwd_ws: the
// workspace hp_encoder_forward ran in, untouched since (NULL: recompute the critical rows' activations instead).
// dedup != 0: channels that peak at the same point share one row below the max-pool (their gradients add): layers 4..1
// run on the DISTINCT critical points (~1/3 of B*512), a count that stays on the device.
namespace {
int& enc_bwd_fused_flag() {
    static int on = env_int("HP_ENC_BWD_FUSED", 1) != 0;
    return on;
}
bool enc_bwd_fused_enabled() { return enc_bwd_fused_flag() != 0; }

// the fc/mu/std tail's backward as tiled GEMM launches (B > 64, or the skinny layer programs switched off)
int enc_tail_backward_gemm(int B, int out_size, const HpEncoderBwdIO& e, const float* dmu_p, int dmu_ld, const EncBwdWs& L,
                           hipStream_t stream) {
    Op op{stream, L.split};
    const HpEncoderWeights* w = e.w;
    const HpEncoderGrads* gr = e.gr;
    if (e.is_vae) {
        TRY(op.lin_dw(L.dlv, 0, out_size, e.f, 0, 512, gr->std_w, 0, B, out_size, 512, 1, gr->std_b));
        TRY(op.lin_dx(L.dlv, 0, out_size, w->std_w, 0, L.tmp, 0, 512, B, out_size, 512, 1, nullptr, 0, 0, nullptr, 0));
    }
    TRY(op.lin_dw(dmu_p, 0, dmu_ld, e.f, 0, 512, gr->mu_w, 0, B, out_size, 512, 1, gr->mu_b));
    TRY(op.lin_dx(dmu_p, 0, dmu_ld, w->mu_w, 0, L.dfc, 0, 512, B, out_size, 512, 1, e.f, 0, 512, e.is_vae ? L.tmp : nullptr, 512));
    TRY(op.lin_dw(L.dfc, 0, 512, e.g, 0, 512, gr->fc_w, 0, B, 512, 512, 1, gr->fc_b));
    TRY(op.lin_dx(L.dfc, 0, 512, w->fc_w, 0, L.dg, 0, 512, B, 512, 512, 1, nullptr, 0, 0, nullptr, 0));
    return 0;
}

// Round-2 launch sequence (one encoder): sort, gather / recompute, delta4, dW5, then a dX GEMM, a dW GEMM and a split-K
// reduce per layer.  Still serves dedup == 0 and the recompute path (fwd_ws == NULL).
int encoder_backward_layered(int B, int Np, int out_size, const HpEncoderBwdIO& e, int dedup, hipStream_t stream) {
    const HpEncoderWeights* w = e.w;
    const HpEncoderGrads* gr = e.gr;
    const int is_vae = e.is_vae, gld = e.grad_out_ld;
    const long Rc = (long)B * 512;
    EncBwdWs L = enc_bwd_layout(e.ws, B, out_size);
    float* xc = L.xc;
    float** hc = L.hc;
    float** dl = L.dl;
    float *dmu = L.dmu, *dlv = L.dlv, *dfc = L.dfc, *dg = L.dg;
    // (the layered launches read h1..h4 as fp32 rows: a forward that left them in P-format is converted in place first — a
    //  no-op when the workspace's format word says fp32)
    if (e.fwd_ws) TRY(hp_conv_pp_unpack_ws(const_cast<float*>(e.fwd_ws), (long)B * Np, stream));
    TRY(enc_critical_rows(B, Np, e.x, w, out_size, e.argidx, e.fwd_ws, e.ws, dedup, stream));

    // ---- heads (model/encoder.py:46-53)
    const float* dmu_p;
    if (is_vae) {
        const long n = (long)B * out_size;
        hipLaunchKernelGGL(vae_head_bwd_kernel, dim3((int)cdiv(n, 256)), dim3(256), 0, stream, n, e.eps, e.lv, e.grad_out, out_size,
                           gld, e.grad_mu, e.grad_explv, dmu, dlv);
        dmu_p = dmu;
    } else {
        dmu_p = e.grad_out;
    }
    const int dmu_ld = is_vae ? out_size : gld;
    int sk = -2;
    if (hp_skinny_enabled() && B <= 64 && dmu_p && dmu_ld % 4 == 0) {
        const EncTailBwd t{e.g, e.f, w, dmu_p, dmu_ld, is_vae ? dlv : nullptr, gr, dfc, dg, L.split};
        sk = enc_tail_backward_skinny(B, out_size, 1, &t, stream);
    }
    if (sk != -2) TRY(sk);
    else TRY(enc_tail_backward_gemm(B, out_size, e, dmu_p, dmu_ld, L, stream));

    // ---- conv stack on the critical rows (B*512 of them, or the distinct ones)
    const float* h4_full = e.fwd_ws ? e.fwd_ws + (long)B * Np * (64 + 128 + 256) : nullptr;
    Op opc{stream, L.split, dedup ? L.crit.total : nullptr};
    if (dedup) {
        hipLaunchKernelGGL(enc_l5_bwd_kernel, dim3(512), dim3(512), 0, stream, B, 512, 512, dg, w->conv_w[4],
                           h4_full ? h4_full : hc[4], h4_full ? e.argidx : (const int*)nullptr, Np,
                           h4_full ? (const int*)nullptr : L.crit.slot, h4_full ? (const int*)nullptr : L.crit.off,
                           gr->conv_w[4], (float*)nullptr, gr->conv_b[4]);
        hipLaunchKernelGGL(crit_l5_dx_kernel, dim3((unsigned)Rc), dim3(128), 0, stream, Np, L.crit, dg, w->conv_w[4],
                           h4_full ? h4_full : hc[4], h4_full ? 1 : 0, dl[4]);
    } else {
        hipLaunchKernelGGL(enc_l5_bwd_kernel, dim3(512), dim3(512), 0, stream, B, 512, 512, dg, w->conv_w[4],
                           h4_full ? h4_full : hc[4], h4_full ? e.argidx : (const int*)nullptr, Np, (const int*)nullptr,
                           (const int*)nullptr, gr->conv_w[4], dl[4], gr->conv_b[4]);
    }
    for (int l = 4; l >= 1; --l) {
        const float* below = l > 1 ? hc[l - 1] : xc;
        TRY(opc.lin_dw(dl[l], 0, kEnc[l], below, 0, kEnc[l - 1], gr->conv_w[l - 1], 0, (int)Rc, kEnc[l], kEnc[l - 1], 1,
                       gr->conv_b[l - 1]));
        if (l > 1)
            TRY(opc.lin_dx(dl[l], 0, kEnc[l], w->conv_w[l - 1], 0, dl[l - 1], 0, kEnc[l - 1], (int)Rc, kEnc[l], kEnc[l - 1], 1,
                           hc[l - 1], 0, kEnc[l - 1], nullptr, 0));
    }
    HP_RETURN_LAST_ERROR();
}

inline bool a16(const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; }

// Round 3: the conv stacks of the n encoders in prep + chain + dW + reduce launches (enc_bwd.hip), their tails in three
// shared skinny launches: 7 launches for a HyperPocket step's two encoders (round 2: ~34 on two streams).
// `after` (may be NULL): a stream to be ordered behind the tails' launches (event record on `stream`, wait on `after`).
int order_behind(hipStream_t stream, hipStream_t after) {
    if (!after || after == stream) return 0;
    return hp_order_streams(stream, after);
}

int encoder_backward_fused(int B, int Np, int out_size, int n, const HpEncoderBwdIO* io, hipStream_t stream, hipStream_t after) {
    HpEncBwdArgs a{};
    a.n = n; a.B = B; a.Np = Np; a.out = out_size;
    // row ranges of the dW launch.  f16 launch (enc_bwd_f16.hip): 12 workgroups per (encoder, range) group, a group on ONE XCD,
    // two workgroups per CU -> 5 groups = 60 of an XCD's 64 slots, 40 groups on the chip: one round (46 groups ran as two).
    const int dflt_splits = hp_enc_bwd_chain_f16_enabled() ? 40 / n : 23;
    a.S = std::max(1, std::min(B, std::min(env_int("HP_EB_SPLITS", dflt_splits), HP_EB_MAX_SPLITS)));   // S <= B: the h4 slot holds the partials
    EncBwdWs L[2];
    EncTailBwd t[2];
    bool skinny_ok = hp_skinny_enabled() && B <= 64;
    for (int z = 0; z < n; ++z) {
        const HpEncoderBwdIO& e = io[z];
        L[z] = enc_bwd_layout(e.ws, B, out_size);
        HpEncBwdSide& s = a.e[z];
        s.x = e.x; s.argidx = e.argidx; s.dg = L[z].dg;
        const long R = (long)B * Np;
        s.h[0] = nullptr;
        s.h[1] = e.fwd_ws;
        for (int l = 2; l <= 4; ++l) s.h[l] = s.h[l - 1] + R * kEnc[l - 1];
        {   // the split area behind h5's slot: the workspace's format word and the P-format block exponents (conv_pp.hip)
            const float* area = e.fwd_ws + R * (64 + 128 + 256 + 512 + 512);
            const long tp = hp_conv_split_tiles_pad(R);
            s.fmt = reinterpret_cast<const int*>(area + hp_conv_pp_fmt_offset(R));
            s.pexp[0] = nullptr;
            for (int l = 1; l <= 4; ++l) {
                s.pexp[l] = reinterpret_cast<const int*>(area + hp_conv_pp_exp_offset(l, tp));
                s.pncb[l] = hp_conv_pp_ncb(l);
                int sh = 0;
                while ((kEnc[l] / s.pncb[l]) >> (sh + 1)) ++sh;
                s.pcbs[l] = sh;
            }
        }
        for (int l = 0; l < 5; ++l) {
            s.W[l] = e.w->conv_w[l];
            s.gW[l] = e.gr->conv_w[l];
            s.gb[l] = e.gr->conv_b[l];
        }
        s.eps = e.eps; s.lv = e.lv; s.gout = e.grad_out; s.gmu = e.grad_mu; s.gexplv = e.grad_explv;
        s.dmu = L[z].dmu; s.dlv = L[z].dlv; s.is_vae = e.is_vae; s.gout_ld = e.grad_out_ld;
        s.crit = L[z].crit;
        for (int l = 1; l <= 4; ++l) s.d[l] = L[z].dl[l];
        s.hc[0] = L[z].xc;
        for (int l = 1; l <= 3; ++l) s.hc[l] = L[z].hc[l];
        s.part = L[z].hc[4];                      // (B*512 x 512 floats; the fused path never gathers h4)
        s.d4max = L[z].d4max;
        s.hmask = reinterpret_cast<unsigned char*>(L[z].hmask);
        s.hmax = L[z].hmax;
        s.bexp = reinterpret_cast<int*>(L[z].bexp);
        s.wt = reinterpret_cast<unsigned char*>(L[z].wt);
        s.wt_us = L[z].wt_us;
        const float* dmu_p = e.is_vae ? L[z].dmu : e.grad_out;
        const int dmu_ld = e.is_vae ? out_size : e.grad_out_ld;
        t[z] = EncTailBwd{e.g, e.f, e.w, dmu_p, dmu_ld, e.is_vae ? L[z].dlv : nullptr, e.gr, L[z].dfc, L[z].dg, L[z].split};
        skinny_ok = skinny_ok && dmu_p && dmu_ld % 4 == 0;
    }
    TRY(hp_enc_bwd_prep(&a, stream));
    int sk = skinny_ok ? enc_tail_backward_skinny(B, out_size, n, t, stream) : -2;
    if (sk != -2) TRY(sk);
    else
        for (int z = 0; z < n; ++z) TRY(enc_tail_backward_gemm(B, out_size, io[z], t[z].dmu, t[z].dmu_ld, L[z], stream));
    TRY(order_behind(stream, after));
    return hp_enc_bwd_conv(&a, stream);
}

bool enc_bwd_io_ok(const HpEncoderBwdIO& e, int out_size) {
    return e.x && e.w && e.argidx && e.g && e.f && e.gr && e.ws && e.grad_out_ld >= out_size &&
           (e.grad_out || e.grad_mu || e.grad_explv) && (!e.is_vae || (e.eps && e.lv));
}
bool enc_bwd_can_fuse(const HpEncoderBwdIO& e) {
    if (!e.fwd_ws || !a16(e.fwd_ws) || !a16(e.ws)) return false;
    for (int l = 0; l < 5; ++l)
        if (!a16(e.w->conv_w[l]) || !a16(e.gr->conv_w[l]) || !a16(e.gr->conv_b[l])) return false;
    return true;
}

int encoder_backward_impl(int B, int Np, int out_size, int n, const HpEncoderBwdIO* io, int dedup, hipStream_t stream,
                          hipStream_t after = nullptr) {
    HP_CHECK_ARG(B > 0 && Np > 0 && out_size > 0 && io && n >= 1 && n <= 2);
    for (int z = 0; z < n; ++z) HP_CHECK_ARG(enc_bwd_io_ok(io[z], out_size));
    HP_CHECK_ARG(!dedup || (long)Np * 512 < (1L << 31));
    bool fuse = dedup && enc_bwd_fused_enabled() && out_size <= 512 && B <= hp_enc_bwd_max_clouds();
    for (int z = 0; z < n; ++z) fuse = fuse && enc_bwd_can_fuse(io[z]);
    if (fuse) return encoder_backward_fused(B, Np, out_size, n, io, stream, after);
    TRY(order_behind(stream, after));      // (the layered launches are small: nothing to keep clear of)
    for (int z = 0; z < n; ++z) TRY(encoder_backward_layered(B, Np, out_size, io[z], dedup, stream));
    return 0;
}
}  // namespace

HP_API int hp_encoder_backward_ld(int B, int Np, const float* x, const HpEncoderWeights* w, int out_size, int is_vae,
                                  const float* eps, const int* argidx, const float* g, const float* f, const float* lv,
                                  const float* grad_out, int grad_out_ld, const float* grad_mu, const float* grad_explv,
                                  const HpEncoderGrads* gr, float* ws, const float* fwd_ws, int dedup, hipStream_t stream) {
    const HpEncoderBwdIO io{x, w, eps, argidx, g, f, lv, grad_out, grad_mu, grad_explv, gr, ws, fwd_ws, is_vae, grad_out_ld};
    return encoder_backward_impl(B, Np, out_size, 1, &io, dedup, stream);
}
HP_API int hp_encoder_backward(int B, int Np, const float* x, const HpEncoderWeights* w, int out_size, int is_vae,
                               const float* eps, const int* argidx, const float* g, const float* f, const float* lv,
                               const float* grad_out, const float* grad_mu, const float* grad_explv,
                               const HpEncoderGrads* gr, float* ws, const float* fwd_ws, int dedup, hipStream_t stream) {
    return hp_encoder_backward_ld(B, Np, x, w, out_size, is_vae, eps, argidx, g, f, lv, grad_out, out_size, grad_mu, grad_explv, gr, ws,
                                  fwd_ws, dedup, stream);
}
// ... and a stream `after` (may be NULL) that is ordered behind the two tails' launches: work the caller enqueues on it
// afterwards starts when the tails are done, beside the conv-stack launches (core/engine.py: the heads' dW + Adam pass).
HP_API int hp_encoder_backward_pair_ordered(int B, int Np, int out_size, const HpEncoderBwdIO* io, int dedup, hipStream_t stream,
                                            hipStream_t after) {
    return encoder_backward_impl(B, Np, out_size, 2, io, dedup, stream, after);
}
// Switches the fused conv-stack backward (enc_bwd.hip) on/off for the parity tests; returns the previous setting.
HP_API int hp_encoder_backward_set_fused(int on) {
    const int prev = enc_bwd_fused_flag();
    enc_bwd_fused_flag() = on != 0;
    return prev;
}
// Test switch for the fused backward's delta chain: 1 (default; HP_EB_CHAIN16) = the f16 matrix pipe with split operands
// (enc_bwd_f16.hip), 0 = round 3's fp32 MFMA chain (enc_bwd.hip), -1 = back to the environment's choice.  Returns the previous setting.
HP_API int hp_encoder_backward_set_chain_f16(int on) { return hp_enc_bwd_chain_f16_set(on); }
// Both encoders of a HyperPocket step in one call (io[0], io[1]: hp_encoder_backward_ld's arguments as structs; same B, Np,
// out_size), on ONE stream: the two conv stacks share the prep / chain / dW / reduce launches, the two tails three skinny
// launches.  Results are those of two hp_encoder_backward_ld calls, bit for bit.
HP_API int hp_encoder_backward_pair(int B, int Np, int out_size, const HpEncoderBwdIO* io, int dedup, hipStream_t stream) {
    return encoder_backward_impl(B, Np, out_size, 2, io, dedup, stream);
}

// =================================================================================================
// Hypernetwork
// =================================================================================================
namespace {
template <typename P>
bool heads_contiguous(P const* hw, P const* hb, const int* out, int n) {
    for (int h = 0; h + 1 < n; ++h)
        if (hw[h + 1] != hw[h] + (long)out[h] * 2048 || hb[h + 1] != hb[h] + out[h]) return false;
    return n > 0;
}

// ---- the trunk as skinny-M layer programs (skinny.hip): one latency-built launch per layer --------------------------
// model/hyper_network.py:16-30, 41 (self.model(x)).  slabs: kSplitWs floats.  Returns -2 when the shapes do not fit.
int trunk_forward_skinny(int B, int in_size, const float* latent, const HpHyperWeights* w, float* t, float* slabs,
                         hipStream_t stream) {
    static const int smax = env_int("HP_SK_SF", 4);
    if (B > 64 || in_size % 32) return -2;
    HpSkProgram pr{};
    HpSkSrc src{};
    src.p = latent; src.S = 1; src.ld = in_size;
    float* tl = t;
    float* sl = slabs;
    int kin = in_size;
    for (int l = 0; l < 5; ++l) {
        const int N = kTrunk[l];
        const int S = sk_ranges(kin, N / 32, smax);
        HpSkOp& op = pr.op[pr.nops++];
        op.type = HP_SK_F; op.phase = l;
        op.a = src;
        op.w = w->trunk_w[l]; op.w_ld = kin;
        op.M = B; op.N = N; op.K = kin; op.CL = kin / S;
        op.out_ld = N;
        HpSkSrc next{};
        next.ld = N;
        if (S == 1) {
            op.out = tl; op.out_bias = w->trunk_b[l]; op.out_relu = l < 4;
            next.p = tl; next.S = 1;
        } else {
            op.out = sl; op.out_slab = 64L * N;
            next.p = sl; next.slab = 64L * N; next.S = S; next.bias = w->trunk_b[l]; next.relu = l < 4;
            next.mat = tl; next.ldmat = N;
            sl += (long)S * 64 * N;
            if (sl - slabs > kSplitWs) return -2;
        }
        src = next;
        kin = N;
        tl += (long)B * N;
    }
    if (src.S > 1) {   // t5 = sum of the last layer's slabs + bias
        HpSkOp& op = pr.op[pr.nops++];
        op.type = HP_SK_FIN; op.phase = 5;
        op.out = src.mat; op.out_ld = src.ldmat;
        src.mat = nullptr;
        op.a = src;
        op.M = B; op.N = 2048; op.K = 1;
    }
    return hp_skinny_run(&pr, stream);
}

// autograd of the trunk: dt[4] (B x 2048) is given; writes every trunk dW/db, dt[0..3] and grad_latent (or skips it)
int trunk_backward_skinny(int B, int in_size, const float* latent, const HpHyperWeights* w, const float* const* act,
                          float* const* dt, const HpHyperGrads* gr, float* grad_latent, float* slabs, hipStream_t stream) {
    static const int smax = env_int("HP_SK_SX", 4);
    if (B > 64 || in_size % 32) return -2;
    HpSkProgram pr{};
    HpSkSrc src{};
    src.p = dt[4]; src.S = 1; src.ld = kTrunk[4];
    float* sl = slabs;
    int phase = 0;
    auto dw_op = [&](int l) {   // dW_l = dt_l^T . (activation below), db_l = column sums of dt_l
        HpSkOp& op = pr.op[pr.nops++];
        op.type = HP_SK_W; op.phase = phase;
        op.a.p = dt[l]; op.a.S = 1; op.a.ld = kTrunk[l];
        op.w = l ? act[l - 1] : latent; op.w_ld = l ? kTrunk[l - 1] : in_size;
        op.out = gr->trunk_w[l]; op.out_ld = op.w_ld; op.rsum = gr->trunk_b[l];
        op.M = B; op.N = kTrunk[l]; op.K = op.w_ld;
    };
    for (int l = 4; l >= 0; --l, ++phase) {
        const int N = kTrunk[l], K = l ? kTrunk[l - 1] : in_size;
        if (l == 0 && !grad_latent) {   // only dt0 is still needed (by dW0)
            HpSkOp& op = pr.op[pr.nops++];
            op.type = HP_SK_FIN; op.phase = phase;
            op.out = src.mat; op.out_ld = src.ldmat;
            src.mat = nullptr;
            op.a = src;
            op.M = B; op.N = N; op.K = 1;
        } else {
            const int S = sk_ranges(N, K / 32, smax);
            HpSkOp& op = pr.op[pr.nops++];
            op.type = HP_SK_X; op.phase = phase;
            op.a = src;
            op.w = w->trunk_w[l]; op.w_ld = K;
            op.M = B; op.N = N; op.K = K; op.CL = N / S;
            op.out_ld = K;
            HpSkSrc next{};
            next.ld = K;
            if (l == 0) {
                if (S == 1) {
                    op.out = grad_latent;
                } else {
                    op.out = sl; op.out_slab = 64L * K;
                    next.p = sl; next.slab = 64L * K; next.S = S;
                    sl += (long)S * 64 * K;
                }
            } else {
                op.out = sl; op.out_slab = 64L * K;
                next.p = sl; next.slab = 64L * K; next.S = S;
                next.mask = act[l - 1]; next.ldm = K; next.mat = dt[l - 1]; next.ldmat = K;
                sl += (long)S * 64 * K;
            }
            if (sl - slabs > kSplitWs) return -2;
            src = next;
        }
        // the weight gradient of the layer whose dt was materialised by the PREVIOUS phase's readers (dt4 is given)
        if (l == 4) dw_op(4);
        else if (l <= 2) dw_op(l + 1);
    }
    // phase 5: grad_latent from its slabs; dW0 (dt0 was materialised in phase 4)
    if (grad_latent && src.p && src.S > 1) {
        HpSkOp& op = pr.op[pr.nops++];
        op.type = HP_SK_FIN; op.phase = phase;
        op.a = src;
        op.out = grad_latent; op.out_ld = in_size;
        op.M = B; op.N = in_size; op.K = 1;
    }
    dw_op(0);
    return hp_skinny_run(&pr, stream);
}
}  // namespace

// saved trunk activations + the split-K slab area the skinny (M = B) forward GEMMs use
HP_API long hp_hypernet_saved_floats(int B) { return (long)B * (64 + 128 + 512 + 1024 + 2048) + kSplitWs + 64; }
HP_API long hp_hypernet_backward_workspace_floats(int B) { return (long)B * (64 + 128 + 512 + 1024 + 2048) + kSplitWs + 64; }

// Test switch: the heads' forward as the streaming bf16-pipe kernel (1, default; HP_HEADS_FWD) or as the tiled fp32 GEMM (0);
// -1 = the environment's choice.  Returns the previous setting.
HP_API int hp_hypernet_set_heads_stream(int on) { return hp_heads_fwd_set(on); }

// model/hyper_network.py:41-43.  t: saved trunk activations (hp_hypernet_saved_floats), theta (B, theta_ld)
HP_API int hp_hypernet_forward(int B, int in_size, const float* latent, const HpHyperWeights* w, float* t, float* theta,
                               int theta_ld, hipStream_t stream) {
    HP_CHECK_ARG(B > 0 && in_size > 0 && latent && w && t && theta && w->n_heads > 0 && w->n_heads <= HP_MAX_HEADS);
    Op op{stream, t + ((long)B * (64 + 128 + 512 + 1024 + 2048) + 3) / 4 * 4};
    const float* in = latent;
    int kin = in_size;
    float* tl = t;
    int sk = hp_skinny_enabled() ? trunk_forward_skinny(B, in_size, latent, w, t, op.splitws, stream) : -2;
    if (sk != -2) {
        TRY(sk);
        in = t + (long)B * (64 + 128 + 512 + 1024);
    } else {
        for (int l = 0; l < 5; ++l) {
            TRY(op.lin_fwd(in, 0, kin, w->trunk_w[l], 0, w->trunk_b[l], 0, tl, 0, kTrunk[l], B, kTrunk[l], kin, 1, l < 4));
            in = tl;
            kin = kTrunk[l];
            tl += (long)B * kTrunk[l];
        }
    }
    int total = 0;
    for (int hd = 0; hd < w->n_heads; ++hd) total += w->head_out[hd];
    HP_CHECK_ARG(total <= theta_ld);
    if (heads_contiguous(w->head_w, w->head_b, w->head_out, w->n_heads)) {
        // the heads' weights form one (total x 2048) matrix (FlatParameters lays them out back to back): one launch — the
        // streaming bf16-pipe kernel of heads_fwd.hip for B <= 64 (its t5 pieces go where the GEMM's split-K slabs would), else one GEMM
        if (hp_heads_fwd_enabled() && hp_heads_fwd_ws_floats() <= kSplitWs && hp_heads_fwd_ok(B, total, 2048, in, w->head_w[0], op.splitws))
            TRY(hp_heads_fwd(B, total, in, w->head_w[0], w->head_b[0], theta, theta_ld, op.splitws, stream));
        else
            TRY(op.lin_fwd(in, 0, 2048, w->head_w[0], 0, w->head_b[0], 0, theta, 0, theta_ld, B, total, 2048, 1, false));
    } else {
        int off = 0;
        for (int hd = 0; hd < w->n_heads; ++hd) {
            TRY(op.lin_fwd(in, 0, 2048, w->head_w[hd], 0, w->head_b[hd], 0, theta + off, 0, theta_ld, B, w->head_out[hd], 2048,
                           1, false));
            off += w->head_out[hd];
        }
    }
    HP_RETURN_LAST_ERROR();
}

namespace {
int hypernet_backward_impl(int B, int in_size, const float* latent, const HpHyperWeights* w, const float* t, const float* grad_theta,
                           int theta_ld, const HpHyperGrads* gr, float* grad_latent, float* ws, hipStream_t stream, hipStream_t after);
}
HP_API int hp_hypernet_backward(int B, int in_size, const float* latent, const HpHyperWeights* w, const float* t,
                                const float* grad_theta, int theta_ld, const HpHyperGrads* gr, float* grad_latent, float* ws,
                                hipStream_t stream) {
    return hypernet_backward_impl(B, in_size, latent, w, t, grad_theta, theta_ld, gr, grad_latent, ws, stream, nullptr);
}
// ... with a second stream `after` ordered behind the LAST READER of the heads' weights (d t5 = d theta . W and its split-K
// reduce): what the caller enqueues on `after` next — the in-place dW + Adam pass over those weights — starts there, beside the
// trunk's backward launches, instead of behind the whole call.
HP_API int hp_hypernet_backward_ordered(int B, int in_size, const float* latent, const HpHyperWeights* w, const float* t,
                                        const float* grad_theta, int theta_ld, const HpHyperGrads* gr, float* grad_latent, float* ws,
                                        hipStream_t stream, hipStream_t after) {
    return hypernet_backward_impl(B, in_size, latent, w, t, grad_theta, theta_ld, gr, grad_latent, ws, stream, after);
}
namespace {
int hypernet_backward_impl(int B, int in_size, const float* latent, const HpHyperWeights* w, const float* t, const float* grad_theta,
                           int theta_ld, const HpHyperGrads* gr, float* grad_latent, float* ws, hipStream_t stream, hipStream_t after) {
    HP_CHECK_ARG(B > 0 && in_size > 0 && latent && w && t && grad_theta && gr && ws);
    const float* act[5];
    {
        const float* tl = t;
        for (int l = 0; l < 5; ++l) {
            act[l] = tl;
            tl += (long)B * kTrunk[l];
        }
    }
    float* p = ws;
    float* dt[5];
    for (int l = 0; l < 5; ++l) {
        dt[l] = p;
        p += (long)B * kTrunk[l];
    }
    Op op{stream, p};
    // heads: dW_h = dtheta_h^T t5 ; db_h = colsum ; dt5 = sum_h dtheta_h W_h
    int off = 0, total = 0;
    for (int hd = 0; hd < w->n_heads; ++hd) total += w->head_out[hd];
    // gr->head_w[0] == NULL: the caller forms the heads' weight gradient itself (data-parallel ranks exchange d theta and
    // t5 and each computes a row slice of the GLOBAL dW: hp_hypernet_heads_dw_rows); only the bias gradients are made here
    const bool skip_dw = gr->head_w[0] == nullptr;
    bool gb_contig = true;
    for (int h = 0; h + 1 < w->n_heads; ++h) gb_contig = gb_contig && gr->head_b[h + 1] == gr->head_b[h] + w->head_out[h];
    HP_CHECK_ARG(!skip_dw || (gb_contig && heads_contiguous(w->head_w, w->head_b, w->head_out, w->n_heads)));
    const bool fused = heads_contiguous(w->head_w, w->head_b, w->head_out, w->n_heads) &&
                       (skip_dw || heads_contiguous(gr->head_w, gr->head_b, w->head_out, w->n_heads));
    if (fused) {
        if (!skip_dw && hp_heads_dw_fast_ok(2048, act[4], gr->head_w[0])) {
            TRY(hp_heads_dw_launch(B, total, 0, grad_theta, theta_ld, act[4], 2048, gr->head_w[0], gr->head_b[0], stream));   // hypernet.hip
        } else if (!skip_dw)
            TRY(op.lin_dw(grad_theta, 0, theta_ld, act[4], 0, 2048, gr->head_w[0], 0, B, total, 2048, 1, gr->head_b[0]));
        else if (gr->head_b[0])
            TRY(op.colsum(grad_theta, 0, theta_ld, B, total, 1, gr->head_b[0], 0));
        TRY(op.lin_dx(grad_theta, 0, theta_ld, w->head_w[0], 0, dt[4], 0, 2048, B, total, 2048, 1, nullptr, 0, 0, nullptr, 2048));
    }
    for (int hd = 0; hd < w->n_heads && !fused; ++hd) {
        const int nh = w->head_out[hd];
        TRY(op.lin_dw(grad_theta + off, 0, theta_ld, act[4], 0, 2048, gr->head_w[hd], 0, B, nh, 2048, 1, gr->head_b[hd]));
        TRY(op.lin_dx(grad_theta + off, 0, theta_ld, w->head_w[hd], 0, dt[4], 0, 2048, B, nh, 2048, 1, nullptr, 0, 0,
                      hd ? dt[4] : nullptr, 2048));
        off += nh;
    }
    if (after && after != stream) TRY(hp_order_streams(stream, after));      // the heads' weights have been read for the last time
    // trunk
    {
        const int sk = hp_skinny_enabled() ? trunk_backward_skinny(B, in_size, latent, w, act, dt, gr, grad_latent, p, stream) : -2;
        if (sk != -2) return sk ? sk : (int)hipGetLastError();
    }
    for (int l = 4; l >= 0; --l) {
        const float* below = l ? act[l - 1] : latent;
        const int kin = l ? kTrunk[l - 1] : in_size;
        TRY(op.lin_dw(dt[l], 0, kTrunk[l], below, 0, kin, gr->trunk_w[l], 0, B, kTrunk[l], kin, 1, gr->trunk_b[l]));
        if (l > 0)
            TRY(op.lin_dx(dt[l], 0, kTrunk[l], w->trunk_w[l], 0, dt[l - 1], 0, kin, B, kTrunk[l], kin, 1, act[l - 1], 0, kin,
                          nullptr, 0));
        else if (grad_latent)
            TRY(op.lin_dx(dt[0], 0, kTrunk[0], w->trunk_w[0], 0, grad_latent, 0, kin, B, kTrunk[0], kin, 1, nullptr, 0, 0,
                          nullptr, 0));
    }
    HP_RETURN_LAST_ERROR();
}
}  // namespace

// Rows [r0, r0+rows) of the heads' weight gradient from factors gathered over the data-parallel ranks:
//   dW_rows (rows x 2048) = dtheta_all[:, r0 : r0+rows]^T (rows x Kc) . t5_all (Kc x 2048),   Kc = sum of the ranks' batches.
// The heads' dW is a rank-B product, so ranks exchange its factors (B x (19011 + 2048) floats each) instead of the
// 156 MB matrix; rank r forms only its row slice of the global gradient (same flops as its local dW: rows shrink by the
// world size, the contraction grows by it).  t5_all is a plain (Kc x 2048) matrix: the caller gathers the t5 block of
// every rank's hp_hypernet_forward `t` (it starts hp_hypernet_t5_offset(B) floats in).  ws: hp_hypernet_heads_dw_workspace_floats() floats.
HP_API long hp_hypernet_t5_offset(int B) { return (long)B * (64 + 128 + 512 + 1024); }
HP_API long hp_hypernet_heads_dw_workspace_floats(void) { return kSplitWs + 64; }
HP_API int hp_hypernet_heads_dw_rows(int Kc, int rows, int r0, const float* dtheta_all, int theta_ld, const float* t5_all,
                                     float* dW_rows, float* ws, hipStream_t stream) {
    HP_CHECK_ARG(Kc > 0 && rows >= 0 && r0 >= 0 && r0 + rows <= theta_ld && dtheta_all && t5_all && ws);
    if (rows == 0) return 0;
    HP_CHECK_ARG(dW_rows);
    if (hp_heads_dw_fast_ok(2048, t5_all, dW_rows))
        return hp_heads_dw_launch(Kc, rows, r0, dtheta_all, theta_ld, t5_all, 2048, dW_rows, nullptr, stream);
    Op op{stream, ws};
    TRY(op.lin_dw(dtheta_all + r0, 0, theta_ld, t5_all, 0, 2048, dW_rows, 0, Kc, rows, 2048, 1));
    HP_RETURN_LAST_ERROR();
}

// =================================================================================================
// Target network (batched over clouds)
// =================================================================================================
namespace {
struct TnLayout {
    int nl;            // number of linear layers (hidden + output)
    int cin[HP_MAX_TN_LAYERS], cout[HP_MAX_TN_LAYERS];
    long woff[HP_MAX_TN_LAYERS], boff[HP_MAX_TN_LAYERS];
    long total, act_per_point;
};
bool tn_layout(int n_hidden, const int* channels, TnLayout* L) {
    if (n_hidden < 1 || n_hidden + 1 > HP_MAX_TN_LAYERS) return false;
    L->nl = n_hidden + 1;
    long off = 0;
    L->act_per_point = 0;
    for (int l = 0; l < L->nl; ++l) {
        L->cin[l] = l ? channels[l - 1] : 3;
        L->cout[l] = l < n_hidden ? channels[l] : 3;
        L->woff[l] = off;
        off += (long)L->cin[l] * L->cout[l];
        L->boff[l] = off;
        off += L->cout[l];
        if (l < n_hidden) L->act_per_point += channels[l];
    }
    L->total = off;
    return true;
}
}  // namespace

HP_API long hp_target_theta_size(int n_hidden, const int* channels) {
    TnLayout L;
    return tn_layout(n_hidden, channels, &L) ? L.total : -1;
}
HP_API long hp_target_saved_floats(int B, int N, int n_hidden, const int* channels) {
    TnLayout L;
    return tn_layout(n_hidden, channels, &L) ? (long)B * N * L.act_per_point : -1;
}
HP_API long hp_target_backward_workspace_floats(int B, int N, int n_hidden, const int* channels) {
    TnLayout L;
    return tn_layout(n_hidden, channels, &L) ? (long)B * N * L.act_per_point + kSplitWs + 64 : -1;
}

// model/target_network.py:31-38 for all B clouds at once.  theta (B, theta_ld): per-cloud weight
// vector [W1 b1 | W2 b2 | ... | Wout bout], W row-major (out,in).  pts (B,N,3) -> y (B,N,3)
// (the reference stores the transpose, rec[b] = y^T: the host returns y.permute(0,2,1)).
// acts: post-ReLU activations of the hidden layers, kept for the backward.
HP_API int hp_target_forward(int B, int N, int n_hidden, const int* channels, const float* theta, int theta_ld,
                             const float* pts, float* acts, float* y, hipStream_t stream) {
    TnLayout L;
    HP_CHECK_ARG(B > 0 && N > 0 && theta && pts && acts && y && tn_layout(n_hidden, channels, &L) && L.total <= theta_ld);
    Op op{stream, nullptr};
    const float* in = pts;
    float* a = acts;
    for (int l = 0; l < L.nl; ++l) {
        const bool last = (l == L.nl - 1);
        float* out = last ? y : a;
        TRY(op.lin_fwd(in, (long)N * L.cin[l], L.cin[l], theta + L.woff[l], theta_ld, theta + L.boff[l], theta_ld, out,
                       (long)N * L.cout[l], L.cout[l], N, L.cout[l], L.cin[l], B, !last));
        in = out;
        if (!last) a += (long)B * N * L.cout[l];
    }
    HP_RETURN_LAST_ERROR();
}

// grad_y (B,N,3) -> grad_theta (B, theta_ld) (every one of the L.total entries is written)
HP_API int hp_target_backward(int B, int N, int n_hidden, const int* channels, const float* theta, int theta_ld,
                              const float* pts, const float* acts, const float* grad_y, float* grad_theta, float* ws,
                              hipStream_t stream) {
    TnLayout L;
    HP_CHECK_ARG(B > 0 && N > 0 && theta && pts && acts && grad_y && grad_theta && ws && tn_layout(n_hidden, channels, &L) &&
                 L.total <= theta_ld);
    const float* act[HP_MAX_TN_LAYERS];
    float* dl[HP_MAX_TN_LAYERS];
    {
        const float* a = acts;
        float* p = ws;
        for (int l = 0; l < n_hidden; ++l) {
            act[l] = a;
            dl[l] = p;
            a += (long)B * N * L.cout[l];
            p += (long)B * N * L.cout[l];
        }
        ws = p;
    }
    Op op{stream, ws};
    const float* d = grad_y;   // gradient w.r.t. the pre-activation output of layer l
    for (int l = L.nl - 1; l >= 0; --l) {
        const float* below = l ? act[l - 1] : pts;
        const int cin = L.cin[l], cout = L.cout[l];
        TRY(op.lin_dw(d, (long)N * cout, cout, below, (long)N * cin, cin, grad_theta + L.woff[l], theta_ld, N, cout, cin, B,
                      grad_theta + L.boff[l], theta_ld));
        if (l > 0) {
            TRY(op.lin_dx(d, (long)N * cout, cout, theta + L.woff[l], theta_ld, dl[l - 1], (long)N * cin, cin, N, cout, cin, B,
                          act[l - 1], (long)N * cin, cin, nullptr, 0));
            d = dl[l - 1];
        }
    }
    HP_RETURN_LAST_ERROR();
}
