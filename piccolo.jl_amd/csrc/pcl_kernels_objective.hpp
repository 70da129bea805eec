// pcl_kernels_objective.hpp -- terminal objectives in one general form, and the Hessian of the whole objective.
//
// Every terminal loss of the reference is  Q |1 - F(x)|  with F at most quadratic in the terminal state(s):
//     F(x) = c' x + sum_r (A_r' x)^2
//   KetInfidelityObjective                      F = |<g|psi>|^2                          2 rows          objectives.jl:24-60
//   CoherentKetInfidelityObjective              F = |sum_i w_i <g_i|psi_i> / sum w|^2    2 rows, JOINT   objectives.jl:96-200
//   DensityMatrix[PureState]InfidelityObjective F = Re tr(rho rho_goal)                  linear          objectives.jl:387-435
//   UnitaryInfidelityObjective                  F = |tr(G'U)|^2 / d^2                    2 rows          objectives.jl:330-337
//     ... with an EmbeddedOperator goal         F = (|M|_F^2 + |tr M|^2) / (ns (ns+1))   2 ns^2 + 2 rows objectives.jl:339-345
// (the rows are built on the host: pcl_set_goal_form, and by pcl_set_goal / pcl_set_goal_subspace for the Hessian of the unitary losses).
// x is one member's terminal state (scope 0: one term per member / seed, weights w_b) or the terminal states of all members
// concatenated in member order (scope 1: one term).  The absolute value is differentiated as the reference's ForwardDiff does: away
// from the kink, sign(1 - F).
//     value     w Q |1 - F|
//     gradient  -s w Q (c + 2 sum_r (A_r' x) A_r)
//     Hessian   -s w Q (2 sum_r A_r A_r') =: -s w Q T      -- T (lower triangle) is formed ONCE per goal (pcl_gram_kernel); a Hessian
//               evaluation is one scaled copy of it per term, plus the regularisers' diagonal, (dt, v) and (dt, dt) entries.
#pragma once

struct PclForm {
    const double *A;  // R x L, row-major
    const double *c;  // L or NULL
    int R, L, scope;  // scope 0: per member (L = x_dim), 1: joint (L = batch x_dim)
};

// element e of term t's argument: the address inside Z
__device__ __forceinline__ long long pcl_form_index(const PclForm &f, int t, int e, const int *__restrict__ x_offs, int x_dim, int N, int z_dim,
                                                    long long z_batch_stride) {
    const int mem = f.scope ? e / x_dim : t, r = f.scope ? e - mem * x_dim : e;
    // MEMBERS: one buffer, member offsets; TRAJ: buffer per seed, one offset
    return (z_batch_stride ? (long long)mem * z_batch_stride + x_offs[0] : (long long)x_offs[mem]) + (long long)(N - 1) * z_dim + r;
}

// value (member[t]), gradient (added to grad at the terminal knot) and the Hessian's coefficient -s w Q sigma (coef[t]) of every term
__global__ __launch_bounds__(256) void pcl_form_kernel(const double *__restrict__ Z, const PclForm f, const int *__restrict__ x_offs, const double *__restrict__ weights,
                                                       double Q, double sigma, int x_dim, int N, int z_dim, long long z_batch_stride, long long grad_batch_stride,
                                                       double *__restrict__ member, double *__restrict__ grad, double *__restrict__ coef) {
    extern __shared__ double lds[];  // p_r (R), then the reduction scratch
    __shared__ double red[8];
    const int t = blockIdx.x, tid = threadIdx.x;
    const double w = f.scope ? 1.0 : (weights ? weights[t] : 1.0);
    double lin = 0.0;
    if (f.c) {
        for (int e = tid; e < f.L; e += 256) lin += f.c[e] * Z[pcl_form_index(f, t, e, x_offs, x_dim, N, z_dim, z_batch_stride)];
    }
    lin = block_sum_256(lin, red);
    double F = lin;
    for (int r = 0; r < f.R; ++r) {
        const double *a = f.A + (long long)r * f.L;
        double s = 0.0;
        for (int e = tid; e < f.L; e += 256) s += a[e] * Z[pcl_form_index(f, t, e, x_offs, x_dim, N, z_dim, z_batch_stride)];
        s = block_sum_256(s, red);
        if (tid == 0) lds[r] = s;
        F += s * s;
    }
    __syncthreads();
    const double sgn = (1.0 - F) >= 0.0 ? 1.0 : -1.0;
    if (tid == 0) {
        if (member) member[t] = w * Q * fabs(1.0 - F);
        if (coef) coef[t] = -sgn * w * Q * sigma;
    }
    if (grad) {
        for (int e = tid; e < f.L; e += 256) {
            double g = f.c ? f.c[e] : 0.0;
            for (int r = 0; r < f.R; ++r) g += 2.0 * lds[r] * f.A[(long long)r * f.L + e];
            const int mem = f.scope ? e / x_dim : t;
            const long long zi = pcl_form_index(f, t, e, x_offs, x_dim, N, z_dim, z_batch_stride) - (z_batch_stride ? (long long)mem * z_batch_stride : 0);
            grad[(z_batch_stride ? (long long)mem * grad_batch_stride : 0) + zi] += -sgn * w * Q * g;
        }
    }
}

// T[i (i + 1) / 2 + j] = 2 sum_r A[r][i] A[r][j],  j <= i   (once per goal)
__global__ __launch_bounds__(256) void pcl_gram_kernel(const PclForm f, double *__restrict__ T) {
    const long long nT = (long long)f.L * (f.L + 1) / 2;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < nT; e += (long long)gridDim.x * 256) {
        long long i = (long long)((sqrt(8.0 * (double)e + 1.0) - 1.0) * 0.5);
        while (i * (i + 1) / 2 > e) --i;
        while ((i + 1) * (i + 2) / 2 <= e) ++i;
        const long long j = e - i * (i + 1) / 2;
        double s = 0.0;
        for (int r = 0; r < f.R; ++r) s += f.A[(long long)r * f.L + i] * f.A[(long long)r * f.L + j];
        T[e] = 2.0 * s;
    }
}
// out[t nT + e] = coef[t] T[e]
__global__ __launch_bounds__(256) void pcl_scale_kernel(const double *__restrict__ T, const double *__restrict__ coef, long long nT, int n_terms, double *__restrict__ out) {
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < nT * n_terms; e += (long long)gridDim.x * 256) out[e] = coef[e / nT] * T[e % nT];
}

// the regularisers' second derivatives, knot k of buffer tb: per regulariser [d2/dv_i^2 (dim) | d2/ddt dv_i (dim; dt_power >= 1) | d2/ddt^2 (1; dt_power 2)]
// J_r = 1/2 sum_k dt_k^p sum_i R_i v_{k,i}^2
__global__ __launch_bounds__(256) void pcl_reg_hess_kernel(const double *__restrict__ Z, const PclReg *__restrict__ regs, int n_regs, const double *__restrict__ Rv,
                                                           double sigma, int N, int z_dim, int dt_off, long long z_batch_stride, long long per_knot,
                                                           double *__restrict__ out) {
    __shared__ double red[8];
    const int k = blockIdx.x, tb = blockIdx.y, tid = threadIdx.x;
    const double *z = Z + (long long)tb * z_batch_stride + (long long)k * z_dim;
    double *o = out + ((long long)tb * N + k) * per_knot;
    const double h = z[dt_off];
    for (int r = 0; r < n_regs; ++r) {
        const PclReg R = regs[r];
        const double wv = R.pw == 0 ? 1.0 : (R.pw == 1 ? h : h * h);
        double s = 0.0;
        for (int i = tid; i < R.dim; i += 256) {
            const double v = z[R.off + i], ri = Rv[R.r0 + i];
            o[i] = sigma * wv * ri;
            if (R.pw >= 1) o[R.dim + i] = sigma * (R.pw == 1 ? 1.0 : 2.0 * h) * ri * v;
            s += ri * v * v;
        }
        if (R.pw == 2) {
            s = block_sum_256(s, red);
            if (tid == 0) o[2 * R.dim] = sigma * s;
        }
        o += R.dim * (R.pw >= 1 ? 2 : 1) + (R.pw == 2 ? 1 : 0);
    }
}
