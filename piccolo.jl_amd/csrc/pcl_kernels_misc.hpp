// pcl_kernels_misc.hpp -- expansion of the compact Jacobian, rollout, DerivativeIntegrator / time-consistency rows,
// terminal infidelity.
#pragma once

// ------------------------------------------------------------------------------------------
// Expansion kernel: compact -> full triplet order (replicate the unique blocks d times).
// grid.x = batch*K*d ; each block copies one (b,k,c) pair of n*n blocks; block c==0 also copies the tail.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pcl_expand_kernel(const double *__restrict__ compact, double *__restrict__ full,
                                                         int d, int n, int m, long long n_bk, int nt) {
    const long long nn = (long long)n * n, xd = (long long)n * d;
    const long long cper = 2 * nn + xd * (m + 1), fper = 2 * d * nn + xd * (m + 1);
    const long long bid = blockIdx.x;
    const int c = (int)(bid % d);
    const long long bk = bid / d;
    if (bk >= n_bk) return;
    const double *src = compact + bk * cper;
    double *dst = full + bk * fper;
    for (long long q = threadIdx.x; q < (nn >> 1); q += blockDim.x) {
        const double2_t v0 = *reinterpret_cast<const double2_t *>(src + 2 * q);
        const double2_t v1 = *reinterpret_cast<const double2_t *>(src + nn + 2 * q);
        store2(dst + c * nn + 2 * q, v0[0], v0[1], nt);
        store2(dst + (d + c) * nn + 2 * q, v1[0], v1[1], nt);
    }
    if (c == 0) {
        const long long tail = xd * (m + 1);
        for (long long q = threadIdx.x; q < (tail >> 1); q += blockDim.x) {
            const double2_t v = *reinterpret_cast<const double2_t *>(src + 2 * nn + 2 * q);
            store2(dst + 2 * d * nn + 2 * q, v[0], v[1], nt);
        }
    }
}

// ------------------------------------------------------------------------------------------
// Rollout: exact piecewise-constant propagation  X_{k+1} = exp(dt_k G(u_k)) X_k  from the knot-0 state
// -- what the reference's unitary_rollout(...; interpolation = :constant) integrates with an ODE solver
// [REF src/quantum/dynamics.jl:631-667] and the slot its RolloutStates reserves for "a GPU rollout"
// [REF src/quantum/trajectories/ensemble_trajectory.jl:56-71].
//   pcl_expm_kernel   one workgroup per (b, k): E = exp(h G) by scaling and squaring, Taylor degree 14 (Horner) at
//                     |h| ||G||_1 / 2^s <= 1/4 (truncation < 1e-21), products on the matrix cores
//   pcl_chain_kernel  one workgroup per member / trajectory: the K dependent n x n x cols products
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pcl_expm_kernel(const KParams p) {
    extern __shared__ double lds[];
    const int n = p.n, LD = p.LD;
    const int tid = threadIdx.x, nth = blockDim.x;
    const int k = blockIdx.x % p.K, b = blockIdx.x / p.K;
    double *A = lds, *T = A + LD * n, *T2 = T + LD * n, *us = T2 + LD * n, *red = us + 8 + p.m;
    const double *zk = p.Z + (long long)b * p.z_batch_stride + (long long)k * p.z_dim;
    const double h = zk[p.dt_off];
    build_G(p, p.G0 + (long long)b * p.g0_batch_stride, zk, A, us);
    __syncthreads();
    if (tid < 64) {
        double cs = 0.0;
        if (tid < n)
            for (int i = 0; i < n; ++i) cs += fabs(A[i + LD * tid]);
        red[tid] = cs;
    }
    __syncthreads();
    double nrm = 0.0;
    for (int j = 0; j < n; ++j) nrm = fmax(nrm, red[j]);
    double theta = fabs(h) * nrm;
    int sq = 0;
    while (theta > 0.25 && sq < 60) {
        theta *= 0.5;
        ++sq;
    }
    const double hs = ldexp(h, -sq);
    for (int e = tid; e < n * n; e += nth) T[(e % n) + LD * (e / n)] = (e % n == e / n) ? 1.0 : 0.0;
    __syncthreads();
    for (int j = 14; j >= 1; --j) {  // T <- I + (hs/j) A T
        gemm_lds<true, false>(A, LD, T, LD, T2, LD, n, n, n);
        __syncthreads();
        const double f = hs / j;
        for (int e = tid; e < n * n; e += nth) {
            const int idx = (e % n) + LD * (e / n);
            T[idx] = ((e % n == e / n) ? 1.0 : 0.0) + f * T2[idx];
        }
        __syncthreads();
    }
    double *cur = T, *oth = T2;
    for (int i = 0; i < sq; ++i) {
        gemm_lds<true, false>(cur, LD, cur, LD, oth, LD, n, n, n);
        __syncthreads();
        double *t = cur;
        cur = oth;
        oth = t;
    }
    double *E = p.expm + ((long long)b * p.K + k) * n * n;
    for (int e = tid; e < n * n; e += nth) E[e] = cur[(e % n) + LD * (e / n)];
}

__global__ __launch_bounds__(256) void pcl_chain_kernel(const KParams p) {
    extern __shared__ double lds[];
    const int n = p.n, LD = p.LD, cols = p.cols;
    const int tid = threadIdx.x, nth = blockDim.x;
    const int b = blockIdx.x;
    const long long xd = (long long)n * cols;
    double *E = lds, *Xa = E + LD * n, *Xb = Xa + LD * cols;
    const double *z0 = p.Z + (long long)b * p.z_batch_stride;
    const int x_off = p.x_offs[p.z_batch_stride ? 0 : b];
    double *out = p.xout + (long long)b * (p.K + 1) * xd;
    for (int e = tid; e < xd; e += nth) {
        const double v = z0[x_off + e];
        Xa[(e % n) + LD * (e / n)] = v;
        out[e] = v;
    }
    double *cur = Xa, *oth = Xb;
    for (int k = 0; k < p.K; ++k) {
        const double *Ek = p.expm + ((long long)b * p.K + k) * n * n;
        __syncthreads();  // previous product complete (E and `oth` free)
        for (int e = tid; e < n * n; e += nth) E[(e % n) + LD * (e / n)] = Ek[e];
        __syncthreads();
        gemm_lds<true, false>(E, LD, cur, LD, oth, LD, n, cols, n);
        __syncthreads();
        for (int e = tid; e < xd; e += nth) out[(long long)(k + 1) * xd + e] = oth[(e % n) + LD * (e / n)];
        double *t = cur;
        cur = oth;
        oth = t;
    }
}

// ------------------------------------------------------------------------------------------
// DerivativeIntegrator rows  x_{k+1} - x_k - dt_k * dx_k  and the time-consistency row  t_{k+1} - t_k - dt_k
// (dx_off < 0: dx == 1).  Trivially sparse; one thread per (b, k, r).  Values per (b,k): [-1 (dim) | +1 (dim) |
// -dt_k (dim, absent for time consistency) | -dx_k[r] (dim)].
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pcl_deriv_kernel(const double *__restrict__ Z, double *__restrict__ delta,
                                                        double *__restrict__ vals, int K, int z_dim, int x_off, int dx_off,
                                                        int dim, int dt_off, long long z_batch_stride, long long total) {
    const int nseg = dx_off >= 0 ? 4 : 3;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
        const int r = (int)(e % dim);
        const long long bk = e / dim;
        const int k = (int)(bk % K);
        const long long b = bk / K;
        const double *zk = Z + b * z_batch_stride + (long long)k * z_dim;
        const double h = zk[dt_off];
        const double dx = dx_off >= 0 ? zk[dx_off + r] : 1.0;
        if (delta) delta[e] = zk[z_dim + x_off + r] - zk[x_off + r] - h * dx;
        if (vals) {
            double *v = vals + bk * (long long)nseg * dim;
            v[r] = -1.0;
            v[dim + r] = 1.0;
            if (dx_off >= 0) {
                v[2 * dim + r] = -h;
                v[3 * dim + r] = -dx;
            } else {
                v[2 * dim + r] = -1.0;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// Terminal unitary infidelity  Q * |1 - |tr(U_goal' U_N)|^2 / d^2|  and its gradient w.r.t. the terminal iso-vec
// (reference: src/control/objectives.jl:330-356).  One workgroup per member / seed.
// With X = [Re U; Im U] (n x d, column c at x[c*n ..]) and the goal stored the same way:
//   t = tr(Ug' U) = sum (gr*ur + gi*ui) + i sum (gr*ui - gi*ur);  F = |t|^2 / d^2
//   dF/dur = 2 (tr*gr - ti*gi) / d^2 ,  dF/dui = 2 (tr*gi + ti*gr) / d^2
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pcl_infidelity_kernel(const double *__restrict__ Z, const double *__restrict__ goal,
                                                             const int *__restrict__ x_offs, double *__restrict__ value,
                                                             double *__restrict__ grad, double Q, int d, int N, int z_dim,
                                                             long long z_batch_stride) {
    __shared__ double red[2][8];
    const int n = 2 * d, b = blockIdx.x, tid = threadIdx.x;
    const double *x = Z + (long long)b * z_batch_stride + (long long)(N - 1) * z_dim + x_offs[z_batch_stride ? 0 : b];
    double tr = 0.0, ti = 0.0;
    for (int e = tid; e < d * d; e += 256) {
        const int c = e / d, i = e - c * d;
        const double ur = x[c * n + i], ui = x[c * n + d + i], gr = goal[c * n + i], gi = goal[c * n + d + i];
        tr += gr * ur + gi * ui;
        ti += gr * ui - gi * ur;
    }
    for (int off = 32; off > 0; off >>= 1) {
        tr += __shfl_down(tr, off, 64);
        ti += __shfl_down(ti, off, 64);
    }
    if ((tid & 63) == 0) {
        red[0][tid >> 6] = tr;
        red[1][tid >> 6] = ti;
    }
    __syncthreads();
    tr = red[0][0] + red[0][1] + red[0][2] + red[0][3];
    ti = red[1][0] + red[1][1] + red[1][2] + red[1][3];
    const double inv = 1.0 / ((double)d * d);
    const double F = (tr * tr + ti * ti) * inv;
    const double sgn = (1.0 - F >= 0.0) ? 1.0 : -1.0;
    if (tid == 0 && value) value[b] = Q * fabs(1.0 - F);
    if (grad) {
        double *g = grad + (long long)b * n * d;
        for (int e = tid; e < d * d; e += 256) {
            const int c = e / d, i = e - c * d;
            const double gr = goal[c * n + i], gi = goal[c * n + d + i];
            g[c * n + i] = -sgn * Q * 2.0 * (tr * gr - ti * gi) * inv;
            g[c * n + d + i] = -sgn * Q * 2.0 * (tr * gi + ti * gr) * inv;
        }
    }
}
