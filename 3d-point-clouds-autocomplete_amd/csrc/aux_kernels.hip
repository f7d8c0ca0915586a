// Auxiliary kernels of the training step that are neither contractions nor structural losses:
//   - decoder input sampler        /root/reference/utils/points.py:8-36  (uniform-in-ball points with
//                                  "progressive normalisation"), drawn ON DEVICE for all B clouds in
//                                  one launch instead of B CPU draws + B host-to-device copies
//                                  (model/full_model.py:72-74)
//   - KLD term                     core/epoch_loops.py:29-30 (forward value and its two gradients)
//   - Adam                         core/main.py:62-66 -> torch.optim.Adam(lr, betas, eps, wd=0, amsgrad=False)
#include "hp_common.h"
#include <algorithm>
#include <cmath>

namespace {

// ---- Philox4x32-10 (Salmon et al. 2011), counter-based: reproducible for a (seed, offset) pair
__device__ __forceinline__ uint4 philox4x32_10(uint4 ctr, uint2 key) {
    constexpr uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t hi0 = __umulhi(M0, ctr.x), lo0 = M0 * ctr.x;
        const uint32_t hi1 = __umulhi(M1, ctr.z), lo1 = M1 * ctr.z;
        ctr = make_uint4(hi1 ^ ctr.y ^ key.x, lo1, hi0 ^ ctr.w ^ key.y, lo0);
        key.x += W0;
        key.y += W1;
    }
    return ctr;
}

__device__ __forceinline__ float u01_to_pm1(uint32_t x) {  // [-1, 1): low + (high-low)*u, u in [0,1)
    return __builtin_fmaf((float)(x >> 8), 2.0f / 16777216.0f, -1.0f);
}

// One lane per output point: rejection-sample the unit ball (acceptance pi/6), then push points
// with |p| < coef out to radius coef (utils/points.py:24-33).  Same distribution as the reference's
// "first N accepted rows of a 3N x 3 uniform(-1,1) draw"; the draws themselves differ (Philox vs
// the torch CPU generator) — exact reference draws come from the host path in utils/points.py.
__global__ __launch_bounds__(256) void sample_points_kernel(long total, float coef, unsigned long long seed,
                                                            unsigned long long offset, float* __restrict__ out) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const uint2 key = make_uint2((uint32_t)seed, (uint32_t)(seed >> 32));
    float x = 0.f, y = 0.f, z = 0.f, n2 = 4.f;
    for (uint32_t attempt = 0; attempt < 64 && !(n2 < 1.0f); ++attempt) {
        const uint4 r = philox4x32_10(make_uint4((uint32_t)i, (uint32_t)((unsigned long long)i >> 32) ^ (attempt << 8),
                                                 (uint32_t)offset, (uint32_t)(offset >> 32)), key);
        x = u01_to_pm1(r.x);
        y = u01_to_pm1(r.y);
        z = u01_to_pm1(r.z);
        n2 = __builtin_fmaf(z, z, __builtin_fmaf(y, y, x * x));
    }
    if (!(n2 < 1.0f)) { x = y = z = 0.f; n2 = 0.f; }  // 2^-64 event; keeps the contract |p| < 1
    const float nrm = __builtin_sqrtf(n2);
    if (nrm < coef) {
        if (nrm > 0.f) {
            const float s = coef / nrm;
            x *= s; y *= s; z *= s;
        } else {
            x = coef; y = 0.f; z = 0.f;
        }
    }
    out[i * 3 + 0] = x;
    out[i * 3 + 1] = y;
    out[i * 3 + 2] = z;
}

// ---------------------------------------------------------------------------------------------------------------
// Random-plane slicer (datasets/utils/dataset_generator.py:6-39): split a cloud into two parts of exactly
// `target` and N - target points by a random plane, rejecting planes until one side has exactly `target` points.
// One workgroup per cloud: points staged in LDS, four candidate planes per round (one per wave, Philox-drawn:
// three uniform [0,1) points -> normal = cross product, bias = +dot(normal, p0) as the reference computes it),
// the lowest-numbered accepting wave of the first accepting round wins (= the first accepted plane of an i.i.d.
// sequence, as in the reference's while-loop), then an order-preserving compaction writes both parts.
// ---------------------------------------------------------------------------------------------------------------
constexpr int kSliceMaxPts = 8192;

// PLANES = false: candidate planes are Philox-drawn on the device and points are classified in fp32 (data augmentation at
// device speed: the law of the reference, other draws).  PLANES = true: the caller supplies the candidate sequence
// planes (B, R, 4) float64 = (params, bias) of HyperPlane — e.g. the planes numpy's generator gives the reference — and
// every point is classified as HyperPlane.check_point does it (dataset_generator.py:10-11): float64
// dot(point, params) + bias, products rounded before their adds; the accepted candidate's index is returned.  The result is
// the reference's own split whenever no point lies within float64 rounding of a candidate plane: numpy evaluates that dot
// through BLAS (dgemv), whose use of fma and summation order is the library's choice, so a point within ~1 ulp (1e-16
// relative) of the plane could classify differently there — the six fixture clouds (tests/golden/slicer.npz, up to 5518
// candidates each) reproduce index and both parts exactly.
template <bool PLANES>
__global__ __launch_bounds__(256) void slice_kernel(int N, int target, const float* __restrict__ pts, unsigned long long seed,
                                                    int max_rounds, const double* __restrict__ planes, int R,
                                                    float* __restrict__ part_a, float* __restrict__ part_b,
                                                    float* __restrict__ plane_out, int* __restrict__ plane_idx,
                                                    int* __restrict__ status) {
    extern __shared__ __attribute__((aligned(16))) float sp[];      // N*3 points
    __shared__ int wave_cnt[4];
    __shared__ double sel[4];
    __shared__ int sel_flag, wsum[4];
    const int cloud = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const float* P = pts + (long)cloud * N * 3;
    for (int i = tid; i < N * 3; i += 256) sp[i] = P[i];
    if (tid == 0) sel_flag = -1;
    __syncthreads();
    const uint2 key = make_uint2((uint32_t)seed, (uint32_t)(seed >> 32));
    int chosen_side = 0;   // +1: the "under" (check > 0) side has `target` points, -1: the other side
    int chosen_idx = -1;
    double a = 0, b = 0, c = 0, d = 0;
    // `> 0` of np.sign(dot + bias): NaN compares false, as in the reference
    auto under_of = [&](int i, double nx, double ny, double nz, double bias) -> bool {
        if constexpr (PLANES)
            return ((((double)sp[i * 3] * nx + (double)sp[i * 3 + 1] * ny) + (double)sp[i * 3 + 2] * nz) + bias) > 0.0;
        else
            return (sp[i * 3] * (float)nx + sp[i * 3 + 1] * (float)ny + sp[i * 3 + 2] * (float)nz + (float)bias) > 0.f;
    };
    const int rounds = PLANES ? (R + 3) / 4 : max_rounds;
    for (int round = 0; round < rounds; ++round) {
        // wave-uniform candidate plane
        double nx, ny, nz, bias;
        bool have = true;
        if constexpr (PLANES) {
            const int idx = round * 4 + wid;
            have = idx < R;
            const double* pl = planes + ((long)cloud * R + (have ? idx : 0)) * 4;
            nx = pl[0]; ny = pl[1]; nz = pl[2]; bias = pl[3];
        } else {
            const uint4 r0 = philox4x32_10(make_uint4((uint32_t)cloud, (uint32_t)round, (uint32_t)wid, 0u), key);
            const uint4 r1 = philox4x32_10(make_uint4((uint32_t)cloud, (uint32_t)round, (uint32_t)wid, 1u), key);
            const uint4 r2 = philox4x32_10(make_uint4((uint32_t)cloud, (uint32_t)round, (uint32_t)wid, 2u), key);
            auto u01 = [](uint32_t x) { return (float)(x >> 8) * (1.0f / 16777216.0f); };
            const float p0x = u01(r0.x), p0y = u01(r0.y), p0z = u01(r0.z);
            const float ux = u01(r1.x) - p0x, uy = u01(r1.y) - p0y, uz = u01(r1.z) - p0z;
            const float vx = u01(r2.x) - p0x, vy = u01(r2.y) - p0y, vz = u01(r2.z) - p0z;
            const float fx = uy * vz - uz * vy, fy = uz * vx - ux * vz, fz = ux * vy - uy * vx;
            nx = fx; ny = fy; nz = fz;
            bias = fx * p0x + fy * p0y + fz * p0z;                 // HyperPlane(cp, np.dot(cp, points[0]))
        }
        int cnt = 0;
        if (have)      // (wave-uniform; a wave past the last candidate of the list has nothing to count)
            for (int i = lane; i < N; i += 64) cnt += under_of(i, nx, ny, nz, bias);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) cnt += __shfl_down(cnt, o, 64);
        if (lane == 0) wave_cnt[wid] = have ? cnt : -1;
        __syncthreads();
        if (tid == 0) {
            for (int w = 0; w < 4 && sel_flag < 0; ++w)
                if (wave_cnt[w] >= 0 && (wave_cnt[w] == target || N - wave_cnt[w] == target)) sel_flag = w;
        }
        __syncthreads();
        const int win = sel_flag;
        if (win >= 0) {
            if (wid == win && lane == 0) {
                sel[0] = nx; sel[1] = ny; sel[2] = nz; sel[3] = bias;
            }
            __syncthreads();
            a = sel[0]; b = sel[1]; c = sel[2]; d = sel[3];
            chosen_side = (wave_cnt[win] == target) ? 1 : -1;      // the reference tests the "under" side first
            chosen_idx = round * 4 + win;
            break;
        }
        __syncthreads();
    }
    if (chosen_side == 0) {
        if (tid == 0) {
            status[cloud] = 1;       // no plane accepted within max_rounds*4 draws / among the R candidates
            if (plane_idx) plane_idx[cloud] = -1;
        }
        return;
    }
    if (tid == 0) {
        status[cloud] = 0;
        if (plane_idx) plane_idx[cloud] = chosen_idx;
        if (plane_out) {
            plane_out[cloud * 4 + 0] = (float)a; plane_out[cloud * 4 + 1] = (float)b;
            plane_out[cloud * 4 + 2] = (float)c; plane_out[cloud * 4 + 3] = (float)d;
        }
    }
    // order-preserving compaction: chunk of 256 points per iteration, exclusive scan of the membership flags
    float* A = part_a + (long)cloud * target * 3;
    float* Bp = part_b + (long)cloud * (N - target) * 3;
    int base_a = 0;
    for (int i0 = 0; i0 < N; i0 += 256) {
        const int i = i0 + tid;
        bool under = false, valid = i < N;
        if (valid) under = under_of(i, a, b, c, d);
        const bool in_a = valid && (chosen_side > 0 ? under : !under);
        const unsigned long long m = __ballot(in_a);
        const int before = __popcll(m & ((1ull << lane) - 1ull));
        if (lane == 0) wsum[wid] = __popcll(m);
        __syncthreads();
        int woff = 0, tot = 0;
        for (int w = 0; w < 4; ++w) {
            if (w < wid) woff += wsum[w];
            tot += wsum[w];
        }
        if (valid) {
            const int pa = base_a + woff + before;
            const int pb = i - pa;                                  // points before i that are not in A
            float* o = in_a ? A + (long)pa * 3 : Bp + (long)pb * 3;
            o[0] = sp[i * 3]; o[1] = sp[i * 3 + 1]; o[2] = sp[i * 3 + 2];
        }
        base_a += tot;
        __syncthreads();
    }
}

// KLD = 0.5 * sum(exp(v) + mu^2 - 1 - v) / B with v = the encoder's exp(logvar) output
// single block of 1024 threads (a fixed thread -> element assignment and an ordered block sum in double: deterministic);
// four elements per thread in flight — with 256 threads and one element at a time the 8192 elements of a B=64 step were a
// 41 us chain of 32 dependent round trips
__global__ __launch_bounds__(1024) void kld_kernel(long n, float inv_b, const float* __restrict__ v, const float* __restrict__ mu,
                                                   float* __restrict__ out) {
    __shared__ double red[16];
    double s = 0;
    long t = threadIdx.x;
    for (; t + 3 * 1024 < n; t += 4 * 1024) {
        float a[4], m[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            a[u] = v[t + u * 1024];
            m[u] = mu[t + u * 1024];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) s += (double)(expf(a[u]) + m[u] * m[u] - 1.0f - a[u]);
    }
    for (; t < n; t += 1024) {
        const float a = v[t], m = mu[t];
        s += (double)(expf(a) + m * m - 1.0f - a);
    }
    const double tot = hp::block_sum(s, red);
    if (threadIdx.x == 0) out[0] = (float)(0.5 * tot * (double)inv_b);
}

__global__ __launch_bounds__(256) void kld_grad_kernel(long n, float inv_b, const float* __restrict__ v, const float* __restrict__ mu,
                                                       const float* __restrict__ gout, float* __restrict__ gv,
                                                       float* __restrict__ gmu) {
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    if (t >= n) return;
    const float g = gout[0] * inv_b;
    gv[t] = 0.5f * g * (expf(v[t]) - 1.0f);
    gmu[t] = g * mu[t];
}

// torch.optim.Adam single-tensor update, fused: 4 streams in, 3 out, 16 B per lane
__global__ __launch_bounds__(256) void adam_kernel(long n, float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, float b1, float b2, float eps, float step_size,
                                                   float inv_sqrt_bc2, float grad_scale) {
    const long stride = (long)gridDim.x * 256 * 4;
    for (long t = ((long)blockIdx.x * 256 + threadIdx.x) * 4; t < n; t += stride) {
        if (t + 3 < n) {
            float4 pp = *reinterpret_cast<float4*>(p + t);
            float4 gg = *reinterpret_cast<const float4*>(g + t);
            float4 mm = *reinterpret_cast<float4*>(m + t);
            float4 vv = *reinterpret_cast<float4*>(v + t);
            float* pa = &pp.x; float* ga = &gg.x; float* ma = &mm.x; float* va = &vv.x;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float gr = ga[e] * grad_scale;
                ma[e] = b1 * ma[e] + (1.0f - b1) * gr;
                va[e] = b2 * va[e] + (1.0f - b2) * gr * gr;
                const float denom = __builtin_sqrtf(va[e]) * inv_sqrt_bc2 + eps;
                pa[e] = pa[e] - step_size * (ma[e] / denom);
            }
            *reinterpret_cast<float4*>(p + t) = pp;
            *reinterpret_cast<float4*>(m + t) = mm;
            *reinterpret_cast<float4*>(v + t) = vv;
        } else {
            for (long u = t; u < n; ++u) {
                const float gr = g[u] * grad_scale;
                const float mn = b1 * m[u] + (1.0f - b1) * gr;
                const float vn = b2 * v[u] + (1.0f - b2) * gr * gr;
                m[u] = mn;
                v[u] = vn;
                p[u] = p[u] - step_size * (mn / (__builtin_sqrtf(vn) * inv_sqrt_bc2 + eps));
            }
        }
    }
}

}  // namespace

// Decoder input points for `total` = B*N points: out (B,N,3).  coef = progressive-normalisation
// radius (utils/points.py:21-23): linspace(0,1,E)[epoch-1] for epoch<=E else 1; 0 disables it.
HP_API int hp_sample_points(long total, float coef, unsigned long long seed, unsigned long long offset, float* out,
                            hipStream_t stream) {
    HP_CHECK_ARG(total >= 0 && out);
    if (total == 0) return 0;
    hipLaunchKernelGGL(sample_points_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, total, coef, seed, offset,
                       out);
    HP_RETURN_LAST_ERROR();
}

// Random-plane split of B clouds (datasets/utils/dataset_generator.py:26-39): part_a (B,target,3) = the side with
// exactly `target` points, part_b (B,N-target,3) the rest, both in the cloud's original point order; plane (B,4) =
// (normal, bias) of the accepted plane; status (B) = 0, or 1 if no plane was accepted within max_rounds*4 draws.
HP_API int hp_slice_clouds(int B, int N, int target, const float* pts, unsigned long long seed, int max_rounds, float* part_a,
                           float* part_b, float* plane, int* status, hipStream_t stream) {
    HP_CHECK_ARG(B >= 0 && N > 0 && target > 0 && target < N && N <= kSliceMaxPts && max_rounds > 0);
    if (B == 0) return 0;
    HP_CHECK_ARG(pts && part_a && part_b && plane && status);
    hipLaunchKernelGGL(slice_kernel<false>, dim3(B), dim3(256), (size_t)N * 3 * sizeof(float), stream, N, target, pts, seed,
                       max_rounds, (const double*)nullptr, 0, part_a, part_b, plane, (int*)nullptr, status);
    HP_RETURN_LAST_ERROR();
}

// The same split with the CALLER's candidate planes: planes (B, R, 4) float64 on the device, cloud i tries
// planes[i,0], planes[i,1], ... in order, classifying in float64 exactly as HyperPlane.check_point
// (dataset_generator.py:10-11, 32-39); plane_idx (B) = index of the accepted candidate (-1 and status 1 if none of the
// R was accepted).  With the planes numpy draws for the reference this IS the reference's split.
HP_API int hp_slice_clouds_planes(int B, int N, int target, const float* pts, const double* planes, int R, float* part_a,
                                  float* part_b, int* plane_idx, int* status, hipStream_t stream) {
    HP_CHECK_ARG(B >= 0 && N > 0 && target > 0 && target < N && N <= kSliceMaxPts && R > 0);
    if (B == 0) return 0;
    HP_CHECK_ARG(pts && planes && part_a && part_b && plane_idx && status);
    hipLaunchKernelGGL(slice_kernel<true>, dim3(B), dim3(256), (size_t)N * 3 * sizeof(float), stream, N, target, pts, 0ull, 0,
                       planes, R, part_a, part_b, (float*)nullptr, plane_idx, status);
    HP_RETURN_LAST_ERROR();
}

// core/epoch_loops.py:29-30
HP_API int hp_kld_forward(long n, int batch, const float* explv, const float* mu, float* out, hipStream_t stream) {
    HP_CHECK_ARG(n > 0 && batch > 0 && explv && mu && out);
    hipLaunchKernelGGL(kld_kernel, dim3(1), dim3(1024), 0, stream, n, 1.0f / (float)batch, explv, mu, out);
    HP_RETURN_LAST_ERROR();
}

HP_API int hp_kld_backward(long n, int batch, const float* explv, const float* mu, const float* grad_out, float* grad_explv,
                           float* grad_mu, hipStream_t stream) {
    HP_CHECK_ARG(n > 0 && batch > 0 && explv && mu && grad_out && grad_explv && grad_mu);
    hipLaunchKernelGGL(kld_grad_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, n, 1.0f / (float)batch, explv, mu,
                       grad_out, grad_explv, grad_mu);
    HP_RETURN_LAST_ERROR();
}

// The step's scalar loss terms in one launch (core/epoch_loops.py:26-31 plus the optional EMD term):
//   out[0] = loss_r = c_cd * cd ; out[1] = loss_kld = kld ; out[2] = loss_emd = c_emd * sum_b cost[b] (cloud order) ;
//   out[3] = loss_all = out[0] + out[1] + out[2].   kld / cost may be NULL (term absent: 0).
namespace {
__global__ __launch_bounds__(64) void step_losses_kernel(int b, const float* __restrict__ cd, const float* __restrict__ kld,
                                                         const float* __restrict__ cost, float c_cd, float c_emd,
                                                         float* __restrict__ out) {
    if (threadIdx.x != 0) return;
    float e = 0.f;
    if (cost) {
        int i = 0;
        for (; i + 8 <= b; i += 8) {
            float t[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) t[u] = cost[i + u];
#pragma unroll
            for (int u = 0; u < 8; ++u) e += t[u];
        }
        for (; i < b; ++i) e += cost[i];
    }
    const float lr = c_cd * cd[0], lk = kld ? kld[0] : 0.f, le = c_emd * e;
    out[0] = lr;
    out[1] = lk;
    out[2] = le;
    out[3] = (lr + lk) + le;
}
}  // namespace

HP_API int hp_step_losses(int b, const float* cd, const float* kld, const float* cost, float c_cd, float c_emd, float* out,
                          hipStream_t stream) {
    HP_CHECK_ARG(b >= 0 && cd && out);
    hipLaunchKernelGGL(step_losses_kernel, dim3(1), dim3(64), 0, stream, b, cd, kld, cost, c_cd, c_emd, out);
    HP_RETURN_LAST_ERROR();
}

// One Adam step over a contiguous run of n parameters (step = 1-based step count).
// grad_scale multiplies the gradient first (1 for the reference's semantics).
HP_API int hp_adam_step(long n, float* p, const float* g, float* m, float* v, float lr, float beta1, float beta2, float eps,
                        int step, float grad_scale, hipStream_t stream) {
    HP_CHECK_ARG(n >= 0 && step >= 1);
    if (n == 0) return 0;
    HP_CHECK_ARG(p && g && m && v);
    const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
    const float step_size = (float)((double)lr / bc1);
    const float inv_sqrt_bc2 = (float)(1.0 / sqrt(bc2));
    const bool al = (((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0;
    if (!al) {  // unaligned views: peel to scalar path by launching with n small chunks is overkill; use 1-wide lanes
        // process the unaligned head (< 4 elements) on a tiny launch, then the aligned body
        const long head = std::min<long>(n, (4 - (((uintptr_t)p >> 2) & 3)) & 3);
        const bool same = ((((uintptr_t)p ^ (uintptr_t)g) | ((uintptr_t)p ^ (uintptr_t)m) | ((uintptr_t)p ^ (uintptr_t)v)) & 15) == 0;
        if (!same) return -1;  // the four streams must share their 16-byte phase
        if (head) {
            hipLaunchKernelGGL(adam_kernel, dim3(1), dim3(256), 0, stream, head, p, g, m, v, beta1, beta2, eps, step_size,
                               inv_sqrt_bc2, grad_scale);
            p += head; g += head; m += head; v += head; n -= head;
            if (n == 0) HP_RETURN_LAST_ERROR();
        }
    }
    const long groups = (n + 3) / 4;
    const unsigned blocks = (unsigned)std::min<long>((groups + 255) / 256, 256 * 8);
    hipLaunchKernelGGL(adam_kernel, dim3(blocks), dim3(256), 0, stream, n, p, g, m, v, beta1, beta2, eps, step_size, inv_sqrt_bc2,
                       grad_scale);
    HP_RETURN_LAST_ERROR();
}
