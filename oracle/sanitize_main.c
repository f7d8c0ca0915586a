/*
 * ORACLE — TEST INFRASTRUCTURE ONLY (tests/test_oracle_golden.py::test_oracle_under_address_and_ub_sanitizers).
 *
 * Runs every entry point of structural_losses_ref.c on ragged / degenerate shapes with EXACTLY sized heap buffers, built
 * with -fsanitize=address,undefined.  The reference's kernels carry two hazards of this class —
 * /root/reference/utils/pytorch_structural_losses/approxmatch.cu:179 reads xyz2 one tile past m when m is not a multiple
 * of the block size, nndistance.cu:146-151 scatters with float atomics — and a restatement of them must not inherit the
 * first.  CPU only (GPU sanitizers are not available on this pool).
 */
#include <stdio.h>
#include <stdlib.h>
#include <math.h>

int ref_nndistance(int b, int n, const float *xyz, int m, const float *xyz2, float *d1, int *i1, float *d2, int *i2);
int ref_nndistancegrad(int b, int n, const float *xyz1, int m, const float *xyz2, const float *gd1, const int *i1,
                       const float *gd2, const int *i2, float *g1, float *g2);
int ref_approxmatch_ex(int b, int n, int m, const float *xyz1, const float *xyz2, float *match, float *temp, int contract);
int ref_approxmatch_f64(int b, int n, int m, const float *xyz1, const float *xyz2, double *match, double *cost);
int ref_matchcost(int b, int n, int m, const float *xyz1, const float *xyz2, const float *match, float *out);
int ref_matchcostgrad(int b, int n, int m, const float *xyz1, const float *xyz2, const float *match, float *g1, float *g2);

static unsigned long long s = 88172645463325252ULL;
static float rnd(void) {
    s ^= s << 13; s ^= s >> 7; s ^= s << 17;
    return (float)((s >> 11) & 0xFFFFFF) / 16777216.0f - 0.5f;
}
static float *fbuf(size_t n) {
    float *p = (float *)malloc(n * sizeof(float) + (n == 0));
    for (size_t i = 0; i < n; ++i) p[i] = rnd();
    return p;
}

static int run(int b, int n, int m) {
    float *a = fbuf((size_t)b * n * 3), *c = fbuf((size_t)b * m * 3);
    float *d1 = fbuf((size_t)b * n), *d2 = fbuf((size_t)b * m), *g1 = fbuf((size_t)b * n * 3), *g2 = fbuf((size_t)b * m * 3);
    int *i1 = (int *)malloc((size_t)b * n * sizeof(int) + 1), *i2 = (int *)malloc((size_t)b * m * sizeof(int) + 1);
    float *match = fbuf((size_t)b * m * n), *temp = fbuf((size_t)b * 2 * (n + m)), *out = fbuf(b);
    double *m64 = (double *)malloc((size_t)b * m * n * sizeof(double) + 1), *c64 = (double *)malloc(b * sizeof(double) + 1);
    int bad = 0;
    ref_nndistance(b, n, a, m, c, d1, i1, d2, i2);
    for (long i = 0; i < (long)b * n; ++i) bad += !(i1[i] >= 0 && i1[i] < m) || !(d1[i] >= 0.f);
    for (long i = 0; i < (long)b * m; ++i) bad += !(i2[i] >= 0 && i2[i] < n) || !(d2[i] >= 0.f);
    ref_nndistancegrad(b, n, a, m, c, d1, i1, d2, i2, g1, g2);
    for (int contract = 0; contract < 8; contract += contract < 3 ? 3 : 4) {
        ref_approxmatch_ex(b, n, m, a, c, match, temp, contract);
        for (long i = 0; i < (long)b * m * n; ++i) bad += !(match[i] >= 0.f && match[i] <= 1.0001f);
    }
    ref_matchcost(b, n, m, a, c, match, out);
    for (int i = 0; i < b; ++i) bad += !isfinite(out[i]);
    ref_matchcostgrad(b, n, m, a, c, match, g1, g2);
    ref_approxmatch_f64(b, n, m, a, c, m64, c64);
    for (int i = 0; i < b; ++i) bad += !isfinite(c64[i]);
    free(a); free(c); free(d1); free(d2); free(g1); free(g2); free(i1); free(i2); free(match); free(temp); free(out);
    free(m64); free(c64);
    printf("(%d,%d,%d): %s\n", b, n, m, bad ? "BAD VALUES" : "ok");
    return bad;
}

int main(void) {
    /* degenerate, ragged, n != m either way, n or m not a multiple of any tile the reference uses (512, 1024) */
    static const int shapes[][3] = {{1, 1, 1}, {2, 1, 5}, {2, 5, 1}, {2, 1500, 7}, {3, 7, 1500}, {5, 200, 330}, {2, 333, 130},
                                    {1, 513, 1025}, {4, 64, 64}};
    int bad = 0;
    for (unsigned i = 0; i < sizeof(shapes) / sizeof(shapes[0]); ++i) bad += run(shapes[i][0], shapes[i][1], shapes[i][2]);
    return bad ? 1 : 0;
}
