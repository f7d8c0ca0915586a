/*
 * ORACLE — TEST INFRASTRUCTURE ONLY.  Not part of the product path.
 *
 * Plain-C CPU restatement of the reference's structural-loss CUDA kernels
 *   /root/reference/utils/pytorch_structural_losses/nndistance.cu
 *   /root/reference/utils/pytorch_structural_losses/approxmatch.cu
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.
 *
 * Pinning status
 *   nndistance / nndistancegrad : PINNED — checked against the reference's own
 *       losses/champfer_loss.py (imported in the build container) through the
 *       committed fixtures tests/golden/chamfer_*.npz.
 *   approxmatch / matchcost / matchcostgrad : PARITY UNPINNED — the reference holds
 *       no tests or golden vectors for them and its CUDA sources cannot be built in
 *       this image (no nvcc, ATen CUDA headers).  The restatement follows the scalar
 *       spec the reference itself carries in comments (approxmatch.cu:94-107,
 *       143-159, 195-209) and is checked through analytic properties (mass conservation,
 *       permutation equivariance, finite differences) and BOUNDED against exact arithmetic:
 *       ref_approxmatch_f64 evaluates the same algorithm in fp64, ref_approxmatch_ex under
 *       every fma contraction nvcc may have applied; all agree on the cost to < 4e-7
 *       (tests/test_oracle_golden.py).
 *
 * Arithmetic notes
 *   - squared distances are evaluated as the fmaf chain nvcc's default -fmad=true
 *     contraction produces for  x*x+y*y+z*z  (nndistance.cu:31): fma(z,z,fma(y,y,x*x)).
 *     The HIP kernels use the same chain, so distances and arg-min indices are
 *     comparable bit for bit.
 *   - __expf(x) (approxmatch.cu:86,131,185) is CUDA's fast exponential, defined as
 *     exp2(x * log2(e)) evaluated in fp32 (ex2.approx of the rounded product).  It is
 *     restated as exp2f(x * 1.4426950408889634f): the rounding of the product is part
 *     of the reference's arithmetic and the auction amplifies it in single match entries
 *     (replacing it by a correctly rounded expf moves single entries by up to 3e-4; the
 *     cost moves by <= 5e-7 relative — measured on five shapes up to 2x2048^2).  The HIP
 *     kernels evaluate the same product and use the hardware v_exp_f32 (1 ulp), so EMD
 *     parity is a tolerance, not bit equality.
 *   - per-thread sequential float accumulations keep the reference's order
 *     (ascending l / k); block tree reductions (approxmatch.cu:244-252,279-296) are
 *     restated as double accumulations.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

static inline float sqdist(float ax, float ay, float az, float bx, float by, float bz) {
    /* (b - a) as in nndistance.cu:28-31  buf[k]-x1 */
    float dx = bx - ax, dy = by - ay, dz = bz - az;
    return fmaf(dz, dz, fmaf(dy, dy, dx * dx));
}

/* nndistance.cu:8-130 (NmDistanceKernel), one direction.
 * Strict '<' inside a tile keeps the first index (:32,42); 'result>best' across tiles
 * keeps the earliest tile (:122)  =>  smallest index among equal minima. */
static void nm_distance(int b, int n, const float *xyz, int m, const float *xyz2,
                        float *result, int *result_i) {
#pragma omp parallel for schedule(dynamic)
    for (int i = 0; i < b; i++) {
        for (int j = 0; j < n; j++) {
            const float *p = xyz + ((size_t)i * n + j) * 3;
            float best = 0.f;
            int best_i = 0;
            for (int k = 0; k < m; k++) {
                const float *q = xyz2 + ((size_t)i * m + k) * 3;
                float d = sqdist(p[0], p[1], p[2], q[0], q[1], q[2]);
                if (k == 0 || d < best) { best = d; best_i = k; }
            }
            result[(size_t)i * n + j] = best;
            result_i[(size_t)i * n + j] = best_i;
        }
    }
}

/* nndistance.cu:131-134 */
int ref_nndistance(int b, int n, const float *xyz, int m, const float *xyz2,
                   float *result, int *result_i, float *result2, int *result2_i) {
    if (m > 0) nm_distance(b, n, xyz, m, xyz2, result, result_i);
    if (n > 0) nm_distance(b, m, xyz2, n, xyz, result2, result2_i);
    return 0;
}

/* nndistance.cu:135-154 (NmDistanceGradKernel), one direction; += into both grads. */
static void nm_distance_grad(int b, int n, const float *xyz1, int m, const float *xyz2,
                             const float *grad_dist1, const int *idx1,
                             float *grad_xyz1, float *grad_xyz2) {
    for (int i = 0; i < b; i++) {
        for (int j = 0; j < n; j++) {
            const float *p = xyz1 + ((size_t)i * n + j) * 3;
            int j2 = idx1[(size_t)i * n + j];
            const float *q = xyz2 + ((size_t)i * m + j2) * 3;
            float g = grad_dist1[(size_t)i * n + j] * 2;
            float *g1 = grad_xyz1 + ((size_t)i * n + j) * 3;
            float *g2 = grad_xyz2 + ((size_t)i * m + j2) * 3;
            for (int c = 0; c < 3; c++) {
                float t = g * (p[c] - q[c]);
                g1[c] += t;
                g2[c] += -t;
            }
        }
    }
}

/* nndistance.cu:155-160 */
int ref_nndistancegrad(int b, int n, const float *xyz1, int m, const float *xyz2,
                       const float *grad_dist1, const int *idx1,
                       const float *grad_dist2, const int *idx2,
                       float *grad_xyz1, float *grad_xyz2) {
    memset(grad_xyz1, 0, (size_t)b * n * 3 * sizeof(float));
    memset(grad_xyz2, 0, (size_t)b * m * 3 * sizeof(float));
    nm_distance_grad(b, n, xyz1, m, xyz2, grad_dist1, idx1, grad_xyz1, grad_xyz2);
    nm_distance_grad(b, m, xyz2, n, xyz1, grad_dist2, idx2, grad_xyz2, grad_xyz1);
    return 0;
}

static inline float fast_expf(float x) { return exp2f(x * 1.4426950408889634f); }

static inline float pair_d2(const float *p, const float *q) {
    /* (x2-x1)*(x2-x1)+(y2-y1)*(y2-y1)+(z2-z1)*(z2-z1)  (approxmatch.cu:85) with the
     * same left-to-right fma contraction as above */
    float dx = q[0] - p[0], dy = q[1] - p[1], dz = q[2] - p[2];
    return fmaf(dz, dz, fmaf(dy, dy, dx * dx));
}

/* approxmatch.cu:34-213.  temp is (b, 2*(n+m)): [remainL n | remainR m | ratioL n | ratioR m]
 * per cloud (the reference indexes it by blockIdx.x, :35 — scratch, contents after the
 * call are the last state of whatever cloud that block processed; here: cloud i).
 *
 * `contract` selects which single-rounding contractions nvcc's default -fmad=true applies to the
 * reference's source (the reference's setup.py passes no flags; unverifiable here: no nvcc, no
 * reference vectors).  contract = 0 is the literal source (every product rounded before its add);
 * contract = 3 — products entering the running sums through an fma, bits 0 and 1 — is what the
 * default contraction most plausibly yields and what the HIP kernels implement since round 2 (the
 * per-entry parity tests use it; the cost gate is checked against both, they agree to 4e-7):
 *   bit 0: the phase-1/2 sums  `w=__expf(d)*buf; suml+=w`   -> fmaf(e, buf, suml)   (:86-87,:131-132)
 *   bit 1: phase 3             `match+=w; suml+=w`          -> fmaf(e*rl, rr, .)    (:185-187)
 *   bit 2: the phase-2 tail    `sumr+1e-9f`, `remainR-sumr` -> fmaf(sum, rr, 1e-9f), fmaf(-sum, rr, rr) (:137-140)
 * tests/test_oracle_golden.py measures how far each variant is from the fp64 evaluation below. */
int ref_approxmatch_ex(int b, int n, int m, const float *xyz1, const float *xyz2,
                       float *match, float *temp, int contract) {
    float multiL, multiR;
    if (n >= m) { multiL = 1; multiR = (float)(n / m); }      /* integer division, :37-43 */
    else        { multiL = (float)(m / n); multiR = 1; }
    const int c_sum = contract & 1, c_p3 = contract & 2, c_tail = contract & 4;
#pragma omp parallel for schedule(dynamic)
    for (int i = 0; i < b; i++) {
        float *remainL = temp + (size_t)i * (n + m) * 2;
        float *remainR = remainL + n, *ratioL = remainL + n + m, *ratioR = remainL + n + m + n;
        float *M = match + (size_t)i * n * m;
        const float *P = xyz1 + (size_t)i * n * 3, *Q = xyz2 + (size_t)i * m * 3;
        for (size_t j = 0; j < (size_t)n * m; j++) M[j] = 0;
        for (int j = 0; j < n; j++) remainL[j] = multiL;
        for (int j = 0; j < m; j++) remainR[j] = multiR;
        for (int j = 7; j > -2; j--) {                         /* 9 levels, :55 */
            float level = -powf(4.0f, (float)j);
            /* pass 1 (:60-93 / spec :94-107) */
            for (int k = 0; k < n; k++) {
                float suml = 1e-9f;
                for (int l = 0; l < m; l++) {
                    float e = fast_expf(level * pair_d2(P + k * 3, Q + l * 3));
                    if (c_sum) suml = fmaf(e, remainR[l], suml);
                    else { float w = e * remainR[l]; suml += w; }
                }
                ratioL[k] = remainL[k] / suml;
            }
            /* pass 2 (:109-142 / spec :143-159) */
            for (int l = 0; l < m; l++) {
                float sumr = 0;
                for (int k = 0; k < n; k++) {
                    float e = fast_expf(level * pair_d2(P + k * 3, Q + l * 3));
                    if (c_sum) sumr = fmaf(e, ratioL[k], sumr);
                    else { float w = e * ratioL[k]; sumr += w; }
                }
                float rr = remainR[l];
                float denom = c_tail ? fmaf(sumr, rr, 1e-9f) : sumr * rr + 1e-9f;
                float left = c_tail ? fmaf(-sumr, rr, rr) : rr - sumr * rr;
                float consumption = fminf(rr / denom, 1.0f);
                ratioR[l] = consumption * rr;
                remainR[l] = fmaxf(0.0f, left);
            }
            /* pass 3 (:161-194 / spec :195-209) */
            for (int k = 0; k < n; k++) {
                float suml = 0;
                float rl = ratioL[k];
                for (int l = 0; l < m; l++) {
                    float er = fast_expf(level * pair_d2(P + k * 3, Q + l * 3)) * rl;
                    if (c_p3) {
                        M[(size_t)l * n + k] = fmaf(er, ratioR[l], M[(size_t)l * n + k]);
                        suml = fmaf(er, ratioR[l], suml);
                    } else {
                        float w = er * ratioR[l];
                        M[(size_t)l * n + k] += w;
                        suml += w;
                    }
                }
                remainL[k] = fmaxf(0.0f, remainL[k] - suml);
            }
        }
    }
    return 0;
}

int ref_approxmatch(int b, int n, int m, const float *xyz1, const float *xyz2,
                    float *match, float *temp) {
    return ref_approxmatch_ex(b, n, m, xyz1, xyz2, match, temp, 0);
}

/* The same nine-level algorithm evaluated in fp64 with libm's exp (the fp32 inputs are exact in
 * fp64): the exact-arithmetic yardstick the fp32 variants and the HIP kernels are measured against
 * (none of them is expected to match it better than the auction's amplification of fp32 rounding
 * allows — that distance is what justifies the parity tolerances).  cost[i] as in ref_matchcost. */
int ref_approxmatch_f64(int b, int n, int m, const float *xyz1, const float *xyz2,
                        double *match, double *cost) {
    double multiL, multiR;
    if (n >= m) { multiL = 1; multiR = (double)(n / m); }
    else        { multiL = (double)(m / n); multiR = 1; }
#pragma omp parallel for schedule(dynamic)
    for (int i = 0; i < b; i++) {
        double *st = (double *)malloc(sizeof(double) * 2 * ((size_t)n + m));
        double *remainL = st, *remainR = st + n, *ratioL = st + n + m, *ratioR = st + n + m + n;
        double *M = match + (size_t)i * n * m;
        const float *P = xyz1 + (size_t)i * n * 3, *Q = xyz2 + (size_t)i * m * 3;
        for (size_t j = 0; j < (size_t)n * m; j++) M[j] = 0;
        for (int j = 0; j < n; j++) remainL[j] = multiL;
        for (int j = 0; j < m; j++) remainR[j] = multiR;
#define D2(k, l) (((double)Q[(l)*3] - P[(k)*3]) * ((double)Q[(l)*3] - P[(k)*3]) + \
                  ((double)Q[(l)*3+1] - P[(k)*3+1]) * ((double)Q[(l)*3+1] - P[(k)*3+1]) + \
                  ((double)Q[(l)*3+2] - P[(k)*3+2]) * ((double)Q[(l)*3+2] - P[(k)*3+2]))
        for (int j = 7; j > -2; j--) {
            double level = -pow(4.0, (double)j);
            for (int k = 0; k < n; k++) {
                double suml = 1e-9;
                for (int l = 0; l < m; l++) suml += exp(level * D2(k, l)) * remainR[l];
                ratioL[k] = remainL[k] / suml;
            }
            for (int l = 0; l < m; l++) {
                double sumr = 0;
                for (int k = 0; k < n; k++) sumr += exp(level * D2(k, l)) * ratioL[k];
                sumr *= remainR[l];
                double consumption = fmin(remainR[l] / (sumr + 1e-9), 1.0);
                ratioR[l] = consumption * remainR[l];
                remainR[l] = fmax(0.0, remainR[l] - sumr);
            }
            for (int k = 0; k < n; k++) {
                double suml = 0, rl = ratioL[k];
                for (int l = 0; l < m; l++) {
                    double w = exp(level * D2(k, l)) * rl * ratioR[l];
                    M[(size_t)l * n + k] += w;
                    suml += w;
                }
                remainL[k] = fmax(0.0, remainL[k] - suml);
            }
        }
        double s = 0;
        for (int l = 0; l < m; l++)
            for (int k = 0; k < n; k++) s += M[(size_t)l * n + k] * sqrt(D2(k, l));
        cost[i] = s;
#undef D2
        free(st);
    }
    return 0;
}

/* approxmatch.cu:215-255: out[i] = sum_k sum_j match[k*n+j] * sqrt(d2(p_j, q_k)) */
int ref_matchcost(int b, int n, int m, const float *xyz1, const float *xyz2,
                  const float *match, float *out) {
#pragma omp parallel for schedule(dynamic)
    for (int i = 0; i < b; i++) {
        const float *P = xyz1 + (size_t)i * n * 3, *Q = xyz2 + (size_t)i * m * 3;
        const float *M = match + (size_t)i * n * m;
        double s = 0;
        for (int k = 0; k < m; k++)
            for (int j = 0; j < n; j++) {
                float x2 = Q[k * 3 + 0] - P[j * 3 + 0];
                float y2 = Q[k * 3 + 1] - P[j * 3 + 1];
                float z2 = Q[k * 3 + 2] - P[j * 3 + 2];
                float d = sqrtf(fmaf(z2, z2, fmaf(y2, y2, x2 * x2)));
                s += (double)(M[(size_t)k * n + j] * d);
            }
        out[i] = (float)s;
    }
    return 0;
}

/* approxmatch.cu:260-322 */
int ref_matchcostgrad(int b, int n, int m, const float *xyz1, const float *xyz2,
                      const float *match, float *grad1, float *grad2) {
#pragma omp parallel for schedule(dynamic)
    for (int i = 0; i < b; i++) {
        const float *P = xyz1 + (size_t)i * n * 3, *Q = xyz2 + (size_t)i * m * 3;
        const float *M = match + (size_t)i * n * m;
        /* grad1 (:301-322): per-thread sequential float over k */
        for (int l = 0; l < n; l++) {
            float x1 = P[l * 3], y1 = P[l * 3 + 1], z1 = P[l * 3 + 2];
            float dx = 0, dy = 0, dz = 0;
            for (int k = 0; k < m; k++) {
                float ex = x1 - Q[k * 3], ey = y1 - Q[k * 3 + 1], ez = z1 - Q[k * 3 + 2];
                float d2 = fmaf(ez, ez, fmaf(ey, ey, ex * ex));
                float d = M[(size_t)k * n + l] * (1.0f / sqrtf(fmaxf(d2, 1e-20f)));
                dx += ex * d; dy += ey * d; dz += ez * d;
            }
            grad1[((size_t)i * n + l) * 3 + 0] = dx;
            grad1[((size_t)i * n + l) * 3 + 1] = dy;
            grad1[((size_t)i * n + l) * 3 + 2] = dz;
        }
        /* grad2 (:260-300): block tree reduction -> double */
        for (int k = 0; k < m; k++) {
            float x2 = Q[k * 3], y2 = Q[k * 3 + 1], z2 = Q[k * 3 + 2];
            double sx = 0, sy = 0, sz = 0;
            for (int j = 0; j < n; j++) {
                float ex = x2 - P[j * 3], ey = y2 - P[j * 3 + 1], ez = z2 - P[j * 3 + 2];
                float d2 = fmaf(ez, ez, fmaf(ey, ey, ex * ex));
                float d = M[(size_t)k * n + j] * (1.0f / sqrtf(fmaxf(d2, 1e-20f)));
                sx += (double)(ex * d); sy += (double)(ey * d); sz += (double)(ez * d);
            }
            grad2[((size_t)i * m + k) * 3 + 0] = (float)sx;
            grad2[((size_t)i * m + k) * 3 + 1] = (float)sy;
            grad2[((size_t)i * m + k) * 3 + 2] = (float)sz;
        }
    }
    return 0;
}
