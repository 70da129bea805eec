/*
 * pade_ref.c -- plain-C CPU restatement of the Pade-4 collocation constraint
 * evaluator (residual, Jacobian values, Hessian-of-Lagrangian values).
 *
 * TEST INFRASTRUCTURE ONLY: linked/loaded only by tests/, __graft_entry__.smoke()
 * and the cpu_baseline leg of bench.py, as the checker / CPU comparator.  The
 * product library (piccolo.jl_amd/csrc) never links or calls it.
 *
 * Follows (paths relative to /root/reference, Piccolo.jl v2.0.2):
 *   - generator closure Ghat(u) = I_d (x) (G_drift + sum_j u_j G_j)
 *       src/control/integrators.jl:48, src/quantum/systems/quantum_systems.jl:225-226,
 *       src/quantum/systems/composite_quantum_systems.jl:131-132
 *   - iso-vec layout of Utilde (column c = [Re U[:,c]; Im U[:,c]])
 *       src/quantum/primitives/isomorphisms.jl:110-118
 *   - knot-major datavec, (u_k, dt_k) belongs to interval k
 *       src/quantum/trajectories/named_trajectory_conversion.jl:321,339-351;
 *       docs/src/concepts/index.md:21
 * The reference's own arithmetic for this path (DirectTrajOpt.jl, un-vendored)
 * is an exp-action constraint; the Pade-4 formulas are SURVEY.md section 8(a).
 * PARITY UNPINNED for the Pade values (no reference goldens exist); this file is
 * checked against oracle/pade_oracle.py, which is pinned on reference data.
 *
 * All matrices column-major (Julia convention).  Z is z_dim x N column-major.
 * Output orders are documented in oracle/pade_oracle.py (jac_structure /
 * hess_structure) and include/piccolo_hip.h.
 */
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* C(n x c) = A(n x n) * B(n x c), column-major, ld = n */
static void gemm_nn(int n, int c, const double *A, const double *B, double *C) {
    for (int j = 0; j < c; ++j) {
        double *Cj = C + (size_t)j * n;
        for (int i = 0; i < n; ++i) Cj[i] = 0.0;
        for (int k = 0; k < n; ++k) {
            const double b = B[(size_t)j * n + k];
            if (b == 0.0) continue;
            const double *Ak = A + (size_t)k * n;
            for (int i = 0; i < n; ++i) Cj[i] += Ak[i] * b;
        }
    }
}

/* C(n x c) = A^T(n x n) * B(n x c) */
static void gemm_tn(int n, int c, const double *A, const double *B, double *C) {
    for (int j = 0; j < c; ++j)
        for (int i = 0; i < n; ++i) {
            const double *Ai = A + (size_t)i * n;
            const double *Bj = B + (size_t)j * n;
            double s = 0.0;
            for (int k = 0; k < n; ++k) s += Ai[k] * Bj[k];
            C[(size_t)j * n + i] = s;
        }
}

typedef struct {
    int *colptr; /* n+1 */
    int *row;
    double *val;
} csc_t;

static void csc_build(int n, const double *A, csc_t *S) {
    int nnz = 0;
    for (int i = 0; i < n * n; ++i) nnz += (A[i] != 0.0);
    S->colptr = (int *)malloc(sizeof(int) * (n + 1));
    S->row = (int *)malloc(sizeof(int) * (nnz ? nnz : 1));
    S->val = (double *)malloc(sizeof(double) * (nnz ? nnz : 1));
    int p = 0;
    for (int k = 0; k < n; ++k) {
        S->colptr[k] = p;
        for (int i = 0; i < n; ++i)
            if (A[(size_t)k * n + i] != 0.0) {
                S->row[p] = i;
                S->val[p] = A[(size_t)k * n + i];
                ++p;
            }
    }
    S->colptr[n] = p;
}
static void csc_free(csc_t *S) {
    free(S->colptr);
    free(S->row);
    free(S->val);
}
/* C(n x c) = S * B */
static void csc_mm(int n, int c, const csc_t *S, const double *B, double *C) {
    memset(C, 0, sizeof(double) * (size_t)n * c);
    for (int j = 0; j < c; ++j)
        for (int k = 0; k < n; ++k) {
            const double b = B[(size_t)j * n + k];
            for (int p = S->colptr[k]; p < S->colptr[k + 1]; ++p) C[(size_t)j * n + S->row[p]] += S->val[p] * b;
        }
}
/* C(n x c) = S^T * B */
static void csc_tmm(int n, int c, const csc_t *S, const double *B, double *C) {
    for (int j = 0; j < c; ++j)
        for (int k = 0; k < n; ++k) {
            double s = 0.0;
            for (int p = S->colptr[k]; p < S->colptr[k + 1]; ++p) s += S->val[p] * B[(size_t)j * n + S->row[p]];
            C[(size_t)j * n + k] = s;
        }
}

static double dot(size_t len, const double *a, const double *b) {
    double s = 0.0;
    for (size_t i = 0; i < len; ++i) s += a[i] * b[i];
    return s;
}

long pade_ref_jac_nnz_per_interval(int d, int m) {
    const long n = 2L * d;
    return 2L * d * n * n + 2L * d * d * (m + 1);
}
long pade_ref_hess_nnz_per_interval(int d, int m) { return (long)(m + 1) * (m + 2) / 2 + 2L * (2L * d * d) * (m + 1); }

/*
 * Fused residual + Jacobian values for all K = N-1 intervals.
 * delta: x_dim*K ; jac: nnz_per_interval*K (either may be NULL).
 * nthreads <= 0: use the OpenMP default.
 */
int pade_ref_eval_jac(int d, int m, int N, int z_dim, int x_off, int u_off, int dt_off, const double *G0,
                      const double *Gj, const double *Z, double *delta, double *jac, int nthreads) {
    const int n = 2 * d, K = N - 1;
    const size_t nn = (size_t)n * n, xd = (size_t)n * d;
    const size_t per = (size_t)pade_ref_jac_nnz_per_interval(d, m);
    csc_t *S = (csc_t *)malloc(sizeof(csc_t) * (m ? m : 1));
    for (int l = 0; l < m; ++l) csc_build(n, Gj + l * nn, &S[l]);
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#else
    (void)nthreads;
#endif
#pragma omp parallel
    {
        double *G = (double *)malloc(sizeof(double) * nn * 2);
        double *G2 = G + nn;
        double *W = (double *)malloc(sizeof(double) * xd * 9);
        double *Sm = W, *D = W + xd, *GS = W + 2 * xd, *GD = W + 3 * xd, *G2D = W + 4 * xd;
        double *T1 = W + 5 * xd, *T2 = W + 6 * xd, *T3 = W + 7 * xd, *T4 = W + 8 * xd;
#pragma omp for schedule(static)
        for (int k = 0; k < K; ++k) {
            const double *zk = Z + (size_t)k * z_dim, *zn = zk + z_dim;
            const double h = zk[dt_off];
            memcpy(G, G0, sizeof(double) * nn);
            for (int l = 0; l < m; ++l) {
                const double u = zk[u_off + l];
                for (int c = 0; c < n; ++c)
                    for (int p = S[l].colptr[c]; p < S[l].colptr[c + 1]; ++p) G[(size_t)c * n + S[l].row[p]] += u * S[l].val[p];
            }
            for (size_t i = 0; i < xd; ++i) {
                Sm[i] = zn[x_off + i] + zk[x_off + i];
                D[i] = zn[x_off + i] - zk[x_off + i];
            }
            gemm_nn(n, d, G, Sm, GS);
            gemm_nn(n, d, G, D, GD);
            gemm_nn(n, d, G, GD, G2D);
            const double c1 = 0.5 * h, c2 = h * h / 12.0;
            if (delta)
                for (size_t i = 0; i < xd; ++i) delta[(size_t)k * xd + i] = D[i] - c1 * GS[i] + c2 * G2D[i];
            if (jac) {
                double *J = jac + (size_t)k * per;
                gemm_nn(n, n, G, G, G2);
                /* seg 0: -B^+ (column-major flat) repeated d times; seg 1: B^- */
                double *J0 = J, *J1 = J + (size_t)d * nn;
                for (int j = 0; j < n; ++j)
                    for (int i = 0; i < n; ++i) {
                        const double id = (i == j) ? 1.0 : 0.0;
                        const double g = G[(size_t)j * n + i], g2 = G2[(size_t)j * n + i];
                        J0[(size_t)j * n + i] = -(id + c1 * g + c2 * g2);
                        J1[(size_t)j * n + i] = id - c1 * g + c2 * g2;
                    }
                for (int c = 1; c < d; ++c) {
                    memcpy(J0 + (size_t)c * nn, J0, sizeof(double) * nn);
                    memcpy(J1 + (size_t)c * nn, J1, sizeof(double) * nn);
                }
                /* tail, column-major: for state column c: [d/du_0 .. d/du_{m-1} | d/ddt], n doubles each */
                double *Jt = J + 2 * (size_t)d * nn;
                for (int l = 0; l < m; ++l) {
                    csc_mm(n, d, &S[l], Sm, T1); /* G_l S      */
                    csc_mm(n, d, &S[l], GD, T2); /* G_l (G D)  */
                    csc_mm(n, d, &S[l], D, T3);  /* G_l D      */
                    gemm_nn(n, d, G, T3, T4);    /* G (G_l D)  */
                    for (int c = 0; c < d; ++c)
                        for (int i = 0; i < n; ++i) {
                            const size_t e = (size_t)c * n + i;
                            Jt[((size_t)c * (m + 1) + l) * n + i] = -c1 * T1[e] + c2 * (T2[e] + T4[e]);
                        }
                }
                for (int c = 0; c < d; ++c)
                    for (int i = 0; i < n; ++i) {
                        const size_t e = (size_t)c * n + i;
                        Jt[((size_t)c * (m + 1) + m) * n + i] = -0.5 * GS[e] + (h / 6.0) * G2D[e];
                    }
            }
        }
        free(G);
        free(W);
    }
    for (int l = 0; l < m; ++l) csc_free(&S[l]);
    free(S);
    return 0;
}

/*
 * Hessian-of-Lagrangian values  grad^2 sum_k mu_k^T delta_k  (order: see
 * oracle/pade_oracle.py hess_structure).  mu: x_dim*K ; hess: per*K.
 */
int pade_ref_hess(int d, int m, int N, int z_dim, int x_off, int u_off, int dt_off, const double *G0, const double *Gj,
                  const double *Z, const double *mu, double *hess, int nthreads) {
    const int n = 2 * d, K = N - 1;
    const size_t nn = (size_t)n * n, xd = (size_t)n * d;
    const size_t per = (size_t)pade_ref_hess_nnz_per_interval(d, m);
    csc_t *S = (csc_t *)malloc(sizeof(csc_t) * (m ? m : 1));
    for (int l = 0; l < m; ++l) csc_build(n, Gj + l * nn, &S[l]);
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#else
    (void)nthreads;
#endif
#pragma omp parallel
    {
        double *G = (double *)malloc(sizeof(double) * nn);
        double *W = (double *)malloc(sizeof(double) * xd * (8 + (size_t)(m ? m : 1)));
        double *Sm = W, *D = W + xd, *GD = W + 2 * xd, *GtM = W + 3 * xd, *GtGtM = W + 4 * xd;
        double *T1 = W + 5 * xd, *T2 = W + 6 * xd, *T3 = W + 7 * xd;
        double *GlD = W + 8 * xd; /* m blocks: G_l D */
#pragma omp for schedule(static)
        for (int k = 0; k < K; ++k) {
            const double *zk = Z + (size_t)k * z_dim, *zn = zk + z_dim;
            const double *M = mu + (size_t)k * xd;
            double *Hk = hess + (size_t)k * per;
            const double h = zk[dt_off];
            const double c1 = 0.5 * h, c2 = h * h / 12.0;
            memcpy(G, G0, sizeof(double) * nn);
            for (int l = 0; l < m; ++l) {
                const double u = zk[u_off + l];
                for (int c = 0; c < n; ++c)
                    for (int p = S[l].colptr[c]; p < S[l].colptr[c + 1]; ++p) G[(size_t)c * n + S[l].row[p]] += u * S[l].val[p];
            }
            for (size_t i = 0; i < xd; ++i) {
                Sm[i] = zn[x_off + i] + zk[x_off + i];
                D[i] = zn[x_off + i] - zk[x_off + i];
            }
            gemm_nn(n, d, G, D, GD);
            gemm_tn(n, d, G, M, GtM);
            gemm_tn(n, d, G, GtM, GtGtM);
            for (int l = 0; l < m; ++l) csc_mm(n, d, &S[l], D, GlD + (size_t)l * xd);
            size_t p = 0;
            /* seg 0: (u_i,u_j), j<=i :  c2 <M,(G_i G_j + G_j G_i) D> = c2 (<G_i^T M, G_j D> + <G_j^T M, G_i D>) */
            for (int i = 0; i < m; ++i) {
                csc_tmm(n, d, &S[i], M, T1); /* G_i^T M */
                for (int j = 0; j <= i; ++j) {
                    csc_tmm(n, d, &S[j], M, T2);
                    Hk[p++] = c2 * (dot(xd, T1, GlD + (size_t)j * xd) + dot(xd, T2, GlD + (size_t)i * xd));
                }
            }
            /* seg 1: (dt,u_j): -1/2 <M,G_j S> + h/6 <M,(G_j G + G G_j) D> */
            for (int j = 0; j < m; ++j) {
                csc_tmm(n, d, &S[j], M, T1); /* G_j^T M */
                Hk[p++] = -0.5 * dot(xd, T1, Sm) + (h / 6.0) * (dot(xd, T1, GD) + dot(xd, GtM, GlD + (size_t)j * xd));
            }
            /* seg 2: (dt,dt): 1/6 <M, G^2 D> = 1/6 <G^T M, G D> */
            Hk[p++] = dot(xd, GtM, GD) / 6.0;
            /* seg 3/5: (u_l, X_k) and (X_{k+1}, u_l):  (-c1 G_l -+ c2 K_l)^T M,  K_l^T M = G^T G_l^T M + G_l^T G^T M */
            double *H3 = Hk + p, *H4 = H3 + (size_t)m * xd, *H5 = H4 + xd, *H6 = H5 + (size_t)m * xd;
            for (int l = 0; l < m; ++l) {
                csc_tmm(n, d, &S[l], M, T1);   /* G_l^T M        */
                gemm_tn(n, d, G, T1, T2);      /* G^T G_l^T M    */
                csc_tmm(n, d, &S[l], GtM, T3); /* G_l^T G^T M    */
                for (size_t i = 0; i < xd; ++i) {
                    const double kt = c2 * (T2[i] + T3[i]);
                    H3[(size_t)l * xd + i] = -c1 * T1[i] - kt;
                    H5[(size_t)l * xd + i] = -c1 * T1[i] + kt;
                }
            }
            for (size_t i = 0; i < xd; ++i) {
                H4[i] = -0.5 * GtM[i] - (h / 6.0) * GtGtM[i];
                H6[i] = -0.5 * GtM[i] + (h / 6.0) * GtGtM[i];
            }
        }
        free(G);
        free(W);
    }
    for (int l = 0; l < m; ++l) csc_free(&S[l]);
    free(S);
    return 0;
}
