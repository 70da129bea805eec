/* expv_ref.c -- TEST INFRASTRUCTURE / CPU BASELINE ONLY (nothing in the product path links or loads this file).
 *
 * The REFERENCE's algorithm for the hot path, restated in plain C so that it can be timed next to the analytic port (oracle/pade_ref.c) on a box
 * without Julia:
 *     delta_k = x_{k+1} - expv(dt_k, Ghat(u_k), x_k),   Ghat(u) = I_d (x) (G_0 + sum_l u_l G_l)        [REF docs/src/concepts/index.md:21;
 *                                                                                                         src/control/integrators.jl:48]
 *     Jacobian: forward-mode dual numbers pushed THROUGH expv, a chunk of directions per pass, as ForwardDiff does
 *                                                                                                        [REF src/control/integrators.jl:282-285]
 * expv is ExponentialAction.jl 0.2 (an un-vendored dependency, Project.toml:12,53): the action of the matrix exponential by the truncated Taylor
 * series of Al-Mohy & Higham, "Computing the action of the matrix exponential", SIAM J. Sci. Comput. 33 (2011), Algorithm 3.2 -- s scaling steps of
 * a degree-m series, (m, s) from the theta_m table for double precision, early termination when two successive terms vanish against the sum.
 * Restated from the paper (the reference holds no source for it); the degrees are taken from the table's rows m = 5, 10, .., 55 and the shift is
 * zero (tr Ghat = 0 for the iso image of -iH, H Hermitian).  Parameter selection and the termination test look at the primal values only.
 *
 * Directions: ForwardDiff seeds every input of the knot pair; what MUST pass through expv are x_k (x_dim), u_k (m) and dt_k (1) -- the derivative
 * with respect to x_{k+1} is the identity and never enters expv.  This file pushes exactly those x_dim + m + 1 directions, in chunks of `chunk`
 * (ForwardDiff's default for long inputs is 12), so the time it reports is a LOWER bound of the reference's Jacobian cost.
 *
 * Output in the triplet order of oracle/pade_ref.c / pade_oracle.pade_jacobian_values: per interval [ -E copies (d x n x n, column-major) | I copies |
 * tail: per state column c the vectors d/du_1 .. d/du_m, d/ddt (n each) ].  Entries of the dense ForwardDiff Jacobian outside that structure are exactly
 * zero (Ghat is block diagonal); their largest magnitude is returned through *off_structure_max so that the test can check it.
 * Validated against pade_oracle.exp_jacobian_values (scipy expm / expm_frechet): tests/test_oracle_pins.py.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

static const int kDeg[11] = {5, 10, 15, 20, 25, 30, 35, 40, 45, 50, 55};
static const double kTheta[11] = {2.4e-3, 1.4e-1, 6.4e-1, 1.4e0, 2.4e0, 3.5e0, 4.7e0, 6.0e0, 7.2e0, 8.5e0, 9.9e0};  /* Table 3.1, tol = 2^-53 */

typedef struct {
    int nnz;
    int *row, *col; /* entry e: G[row][col] */
} pat_t;

/* Y[i][c] += v * X[j][c] for the pattern's entries; X, Y row-major n x d (the d state columns contiguous: the inner loop vectorises) */
static inline void spmm_acc(const pat_t *P, const double *val, int d, const double *X, double *Y) {
    for (int e = 0; e < P->nnz; ++e) {
        const double v = val[e];
        if (v == 0.0) continue;
        const double *x = X + (size_t)P->col[e] * d;
        double *y = Y + (size_t)P->row[e] * d;
        for (int c = 0; c < d; ++c) y[c] += v * x[c];
    }
}
static double norm_inf(const double *a, size_t len) {
    double m = 0.0;
    for (size_t i = 0; i < len; ++i) {
        const double v = fabs(a[i]);
        m = v > m ? v : m;
    }
    return m;
}

/* one chunk pass: duals with P partials through f = expv(t, A, b); direction p of the chunk is global direction dir0 + p:
 *   [0, xd): x_k[dir] (row-major index of the n x d tile handled by the caller)  |  [xd, xd + m): u_l  |  xd + m: dt.
 * B0/F0: n*d values; Bp/Fp: P x n*d partials; W: scratch (P + 1) x n*d. */
static void expv_dual(const pat_t *Pt, const double *Gval, const double *const *Glval, int n, int d, int m, double t, const double *x0, int dir0, int P, int xd,
                      double *F0, double *Fp, double *B0, double *Bp, double *W0, double *Wp, int *terms_out) {
    const size_t len = (size_t)n * d;
    /* ||t A||_1 of Ghat = ||t G||_1 */
    double *colsum = (double *)calloc((size_t)n, sizeof(double));
    for (int e = 0; e < Pt->nnz; ++e) colsum[Pt->col[e]] += fabs(Gval[e]);
    double a1 = 0.0;
    for (int j = 0; j < n; ++j) a1 = colsum[j] > a1 ? colsum[j] : a1;
    free(colsum);
    a1 *= fabs(t);
    int mstar = kDeg[10], s = 1;
    {
        double best = 1e300;
        for (int i = 0; i < 11; ++i) {
            const double si = ceil(a1 / kTheta[i]) < 1.0 ? 1.0 : ceil(a1 / kTheta[i]);
            if (kDeg[i] * si < best) best = kDeg[i] * si, mstar = kDeg[i], s = (int)si;
        }
    }
    memcpy(B0, x0, len * sizeof(double));
    memset(Bp, 0, (size_t)P * len * sizeof(double));
    for (int p = 0; p < P; ++p)
        if (dir0 + p < xd) Bp[(size_t)p * len + (dir0 + p)] = 1.0; /* seed x_k[dir] */
    memcpy(F0, B0, len * sizeof(double));
    memcpy(Fp, Bp, (size_t)P * len * sizeof(double));
    const double tol = ldexp(1.0, -53);
    int terms = 0;
    for (int i = 0; i < s; ++i) {
        double c1 = norm_inf(B0, len);
        for (int j = 1; j <= mstar; ++j) {
            const double sc = t / ((double)s * j), dsc = 1.0 / ((double)s * j); /* the scalar's value and its partial with respect to dt */
            /* W = A b  (value and partials) */
            memset(W0, 0, len * sizeof(double));
            spmm_acc(Pt, Gval, d, B0, W0);
            for (int p = 0; p < P; ++p) {
                double *wp = Wp + (size_t)p * len;
                memset(wp, 0, len * sizeof(double));
                spmm_acc(Pt, Gval, d, Bp + (size_t)p * len, wp);
                const int dir = dir0 + p;
                if (dir >= xd && dir < xd + m) spmm_acc(Pt, Glval[dir - xd], d, B0, wp); /* dA/du_l b */
            }
            /* b = sc * W  (+ dsc * W0 in the dt direction) */
            for (int p = 0; p < P; ++p) {
                double *bp = Bp + (size_t)p * len;
                const double *wp = Wp + (size_t)p * len;
                const int dir = dir0 + p;
                if (dir == xd + m)
                    for (size_t e = 0; e < len; ++e) bp[e] = sc * wp[e] + dsc * W0[e];
                else
                    for (size_t e = 0; e < len; ++e) bp[e] = sc * wp[e];
            }
            for (size_t e = 0; e < len; ++e) B0[e] = sc * W0[e];
            const double c2 = norm_inf(B0, len);
            for (size_t e = 0; e < len; ++e) F0[e] += B0[e];
            for (size_t e = 0; e < (size_t)P * len; ++e) Fp[e] += Bp[e];
            ++terms;
            if (c1 + c2 <= tol * norm_inf(F0, len)) break;
            c1 = c2;
        }
        memcpy(B0, F0, len * sizeof(double));
        memcpy(Bp, Fp, (size_t)P * len * sizeof(double));
    }
    if (terms_out) *terms_out = terms;
}

long expv_ref_jac_nnz_per_interval(int d, int m) {
    const long n = 2L * d;
    return 2L * d * n * n + n * d * (m + 1);
}

/* Z: N x z_dim (knot-major), G0 / Gj column-major n x n.  Intervals k_first .. k_first + k_count - 1 are evaluated; delta / jac rows of the others are
 * left untouched.  Returns 0. */
int expv_ref_eval_jac(int d, int m, int N, int z_dim, int x_off, int u_off, int dt_off, const double *G0, const double *Gj, const double *Z, double *delta, double *jac,
                      int nthreads, int k_first, int k_count, int chunk, double *off_structure_max, long *terms_total) {
    const int n = 2 * d, xd = n * d, K = N - 1;
    const size_t nn = (size_t)n * n, len = (size_t)n * d;
    if (chunk < 1) chunk = 12;
    if (k_first < 0 || k_first + k_count > K) return 1;
    /* union pattern of the generators */
    pat_t P;
    P.nnz = 0;
    P.row = (int *)malloc(nn * sizeof(int));
    P.col = (int *)malloc(nn * sizeof(int));
    for (int j = 0; j < n; ++j)
        for (int i = 0; i < n; ++i) {
            int any = G0[i + (size_t)n * j] != 0.0;
            for (int l = 0; l < m && !any; ++l) any = Gj[(size_t)l * nn + i + (size_t)n * j] != 0.0;
            if (any) P.row[P.nnz] = i, P.col[P.nnz] = j, ++P.nnz;
        }
    double **Glval = (double **)malloc((size_t)(m > 0 ? m : 1) * sizeof(double *));
    for (int l = 0; l < m; ++l) {
        Glval[l] = (double *)malloc((size_t)P.nnz * sizeof(double));
        for (int e = 0; e < P.nnz; ++e) Glval[l][e] = Gj[(size_t)l * nn + P.row[e] + (size_t)n * P.col[e]];
    }
    const int ndir = xd + m + 1, npass = (ndir + chunk - 1) / chunk;
    const long per = expv_ref_jac_nnz_per_interval(d, m);
    double offmax = 0.0;
    long terms_sum = 0;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#else
    (void)nthreads;
#endif
#pragma omp parallel reduction(max : offmax) reduction(+ : terms_sum)
    {
        double *Gval = (double *)malloc((size_t)P.nnz * sizeof(double));
        double *x0 = (double *)malloc(len * sizeof(double));
        double *F0 = (double *)malloc(len * sizeof(double)), *B0 = (double *)malloc(len * sizeof(double)), *W0 = (double *)malloc(len * sizeof(double));
        double *Fp = (double *)malloc((size_t)chunk * len * sizeof(double)), *Bp = (double *)malloc((size_t)chunk * len * sizeof(double)),
               *Wp = (double *)malloc((size_t)chunk * len * sizeof(double));
#pragma omp for collapse(2) schedule(dynamic, 1)
        for (int kk = 0; kk < k_count; ++kk)
            for (int ps = 0; ps < npass; ++ps) {
                const int k = k_first + kk;
                const double *zk = Z + (size_t)k * z_dim, *zn = zk + z_dim;
                const double h = zk[dt_off];
                for (int e = 0; e < P.nnz; ++e) {
                    double v = G0[P.row[e] + (size_t)n * P.col[e]];
                    for (int l = 0; l < m; ++l) v += zk[u_off + l] * Glval[l][e];
                    Gval[e] = v;
                }
                /* the state tile row-major: x0[i * d + c] = X_k[i, c], X_k column-major in Z (column c = iso-vec entries c*n .. c*n + n - 1) */
                for (int c = 0; c < d; ++c)
                    for (int i = 0; i < n; ++i) x0[(size_t)i * d + c] = zk[x_off + (size_t)c * n + i];
                const int dir0 = ps * chunk, Pn = dir0 + chunk <= ndir ? chunk : ndir - dir0;
                /* direction numbering of this file: x-directions in the tile's row-major order: dir = i * d + c */
                int terms = 0;
                expv_dual(&P, Gval, (const double *const *)Glval, n, d, m, h, x0, dir0, Pn, xd, F0, Fp, B0, Bp, W0, Wp, &terms);
                terms_sum += terms;
                double *J = jac ? jac + (size_t)k * per : NULL;
                if (ps == 0 && delta)
                    for (int c = 0; c < d; ++c)
                        for (int i = 0; i < n; ++i) delta[(size_t)k * xd + (size_t)c * n + i] = zn[x_off + (size_t)c * n + i] - F0[(size_t)i * d + c];
                if (!J) continue;
                if (ps == 0) { /* d/dX_{k+1} = I: never enters expv */
                    double *Jn = J + (size_t)d * nn;
                    memset(Jn, 0, (size_t)d * nn * sizeof(double));
                    for (int c = 0; c < d; ++c)
                        for (int i = 0; i < n; ++i) Jn[(size_t)c * nn + i + (size_t)n * i] = 1.0;
                }
                for (int p = 0; p < Pn; ++p) {
                    const int dir = dir0 + p;
                    const double *fp = Fp + (size_t)p * len; /* d expv / d direction, tile row-major */
                    if (dir < xd) {
                        const int j = dir / d, c = dir - j * d; /* x_k[j, c] */
                        for (int cc = 0; cc < d; ++cc)
                            for (int i = 0; i < n; ++i) {
                                const double v = -fp[(size_t)i * d + cc];
                                if (cc == c)
                                    J[(size_t)c * nn + i + (size_t)n * j] = v; /* block c, column j */
                                else if (fabs(v) > offmax)
                                    offmax = fabs(v);
                            }
                    } else {
                        const int l = dir - xd; /* u_l (l < m) or dt (l == m) */
                        double *T = J + 2 * (size_t)d * nn;
                        for (int c = 0; c < d; ++c)
                            for (int i = 0; i < n; ++i) T[((size_t)c * (m + 1) + l) * n + i] = -fp[(size_t)i * d + c];
                    }
                }
            }
        free(Gval), free(x0), free(F0), free(B0), free(W0), free(Fp), free(Bp), free(Wp);
    }
    if (off_structure_max) *off_structure_max = offmax;
    if (terms_total) *terms_total = terms_sum;
    for (int l = 0; l < m; ++l) free(Glval[l]);
    free(Glval), free(P.row), free(P.col);
    return 0;
}
