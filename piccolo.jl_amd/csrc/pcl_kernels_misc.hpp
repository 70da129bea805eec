// pcl_kernels_misc.hpp -- expansion of the compact Jacobian, rollout, DerivativeIntegrator / time-consistency rows,
// terminal infidelity.
#pragma once

// ------------------------------------------------------------------------------------------
// Expansion kernel: compact -> full triplet order (replicate the unique blocks d times).
// Short-lived workgroups in address order (the hardware's dispatcher keeps the front of addresses being written tight):
//   per interval (b,k): 2 S block workgroups -- (sign, slice of cpi state columns): every thread requests its <= 8 element pairs of the
//   n x n tile at once (one round trip; the first version loaded a pair, stored it twice, loaded the next: 2.9 TB/s, bound by that chain),
//   then only issues the 16-byte stores of the slice's copies -- and T tail workgroups (2048 element pairs each of the d/du_l, d/dh run).
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pcl_expand_kernel(const double *__restrict__ compact, double *__restrict__ full,
                                                         int d, int n, int m, long long n_bk, int cpi, int nt) {
    const long long nn = (long long)n * n, xd = (long long)n * d;
    const long long cper = 2 * nn + xd * (m + 1), fper = 2 * d * nn + xd * (m + 1);
    const int S = (d + cpi - 1) / cpi;
    const long long tail2 = (xd * (m + 1)) >> 1;  // element pairs of the tail run
    const int T = (int)((tail2 + 2047) / 2048);
    const int per_bk = 2 * S + T;
    const long long bk = blockIdx.x / per_bk;
    const int r = (int)(blockIdx.x - bk * per_bk);
    if (bk >= n_bk) return;
    const double *src = compact + bk * cper;
    double *dst = full + bk * fper;
    const int tid = threadIdx.x;
    double2_t v[8];
    if (r < 2 * S) {
        const int sign = r / S, sl = r - sign * S;
        const int c0 = sl * cpi, c1 = min(d, c0 + cpi);
        const int nn2 = (int)(nn >> 1);  // n <= 64: at most 2048 pairs = 8 per thread
        src += sign * nn;
#pragma unroll
        for (int q = 0; q < 8; ++q)
            if (tid + 256 * q < nn2) v[q] = *reinterpret_cast<const double2_t *>(src + 2 * (tid + 256 * q));
        double *o = dst + ((long long)sign * d + c0) * nn;
        for (int c = c0; c < c1; ++c, o += nn) {
#pragma unroll
            for (int q = 0; q < 8; ++q)
                if (tid + 256 * q < nn2) store2(o + 2 * (tid + 256 * q), v[q][0], v[q][1], nt);
        }
    } else {
        const long long e0 = (long long)(r - 2 * S) * 2048;
        src += 2 * nn;
        dst += 2 * d * nn;
#pragma unroll
        for (int q = 0; q < 8; ++q)
            if (e0 + tid + 256 * q < tail2) v[q] = *reinterpret_cast<const double2_t *>(src + 2 * (e0 + tid + 256 * q));
#pragma unroll
        for (int q = 0; q < 8; ++q)
            if (e0 + tid + 256 * q < tail2) store2(dst + 2 * (e0 + tid + 256 * q), v[q][0], v[q][1], nt);
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
// Terminal unitary infidelity  w_b Q * |1 - F|  and its gradient w.r.t. the terminal iso-vec.
// One workgroup per member / seed.  Two fidelities, as in the reference:
//   matrix goal      F = |tr(U_goal' U_N)|^2 / d^2                                    src/control/objectives.jl:330-337
//   EmbeddedOperator F = (tr(M'M) + |tr M|^2) / (ns (ns+1)),  M = Ug_sub' U_N[sub,sub]  src/control/objectives.jl:339-345
// With X = [Re U; Im U] (n x d, column c at x[c*n ..]) and the goal stored the same way:
//   t = tr(Ug' U) = sum conj(g) u;  d|t|^2/dRe(u) = 2 Re(t g),  d|t|^2/dIm(u) = 2 Im(t g);
//   d tr(M'M)/d(Re, Im)(U_sub) = 2 (Re, Im)(Ug_sub M).
// `accumulate`: add the gradient into grad (which then is the full-trajectory gradient buffer, terminal knot slot of
// this member) instead of overwriting a per-member x_dim slot.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ double block_sum_256(double v, double *red) {  // fixed order; red: 8 doubles of LDS
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return ((red[0] + red[1]) + red[2]) + red[3];
}

// The objective's final sums inside the infidelity launch (pcl_objective_dev): the workgroup that arrives last at an agent-scope
// ticket (acquire-release) adds the members' terms and the knots' regulariser values in the fixed order of
// pcl_objective_sum_kernel -- same bits, one launch less.  `out` NULL: no fused sum (pcl_infidelity_dev).
struct PclObjSum {
    double *out;            // value[1] (MEMBERS) or value[batch] (TRAJ)
    const double *regval;   // [nbuf][N] per-knot regulariser values (written by the previous launch)
    unsigned int *ticket;   // zero between launches (the last arriver resets it)
    int batch, N, traj_mode;
    int arrivals;           // workgroups that arrive at the ticket: batch (the infidelity launch), batch + nbuf N when the regulariser
                            // workgroups run in the same launch (pcl_ens_tail_kernel: their values are then part of what the last arriver needs)
};
__device__ __forceinline__ double wave_sum_strided_fwd(const double *__restrict__ v, int count) {
    double s = 0.0;
    for (int i = threadIdx.x & 63; i < count; i += 64) s += __hip_atomic_load(v + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // (maybe written in this launch)
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    return s;  // valid in lane 0
}
// HARDWARE ASSUMPTION of the "light" arrivals (here, in pcl_merit_finish_body and in the sc0 sc1 exchange of pcl_hess_sparse4): data published
// with a relaxed agent-scope atomic store (lowered to a write-through `sc1` store on gfx942 / gfx950), then `s_waitcnt vmcnt(0)`, then a
// RELAXED ticket RMW -- no release.  Under the HIP / HSA memory model that is a data race; it is correct on gfx942 / gfx950 because an sc1
// store is performed at the memory side (past the XCD's L2) once vmcnt has counted it, the ticket RMW is performed there too and the reader
// loads with agent scope (sc1: L1 bypassed).  Gated below: any other target takes the acquire-release ticket.  Looped against the
// two-launch result in tests/test_parity_gpu.py (test_ensemble_step_in_two_launches_equals_the_separate_calls).
#if defined(__gfx950__) || defined(__gfx942__)
#define PCL_LIGHT_ARRIVALS 1
#else
#define PCL_LIGHT_ARRIVALS 0
#endif
// light: the arriver has published its one value with an agent-scope atomic store (thread 0) and waits for that store alone -- a release
// at agent scope writes the XCD's whole L2 back, and the regulariser workgroups (a hundred of them, each with a fresh gradient row in
// the L2) would do so one after the other: 40 us for the launch instead of 10
__device__ __forceinline__ void objective_finish(const PclObjSum &fin, const double *member, double *red, bool light = false) {
    if (!fin.out) return;
    light = light && PCL_LIGHT_ARRIVALS;
    __syncthreads();  // this workgroup's member[b] (thread 0) is written
    __shared__ int last;
    if (threadIdx.x == 0) {
        if (light) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned int t = light ? __hip_atomic_fetch_add(fin.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                     : __hip_atomic_fetch_add(fin.ticket, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        last = (t == (unsigned)fin.arrivals - 1u);
    }
    __syncthreads();
    if (!last) return;
    if (threadIdx.x == 0) __hip_atomic_store(fin.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    if (threadIdx.x < 64) {
        if (fin.traj_mode) {
            for (int b = 0; b < fin.batch; ++b) {
                const double s = wave_sum_strided_fwd(fin.regval + (long long)b * fin.N, fin.N);
                if (threadIdx.x == 0) fin.out[b] = __hip_atomic_load(member + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + s;
            }
        } else {
            double sm = 0.0;
            for (int i = threadIdx.x; i < fin.batch; i += 64) sm += __hip_atomic_load(member + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            for (int off = 32; off > 0; off >>= 1) sm += __shfl_down(sm, off, 64);
            const double sr = wave_sum_strided_fwd(fin.regval, fin.N);
            if (threadIdx.x == 0) fin.out[0] = sm + sr;
        }
    }
}

__device__ __forceinline__ void pcl_infidelity_body(const int b, const double *__restrict__ Z, const double *__restrict__ goal,
                                                    const int *__restrict__ sub, int ns,
                                                    const int *__restrict__ x_offs, const double *__restrict__ weights,
                                                    double *__restrict__ value, double *__restrict__ grad,
                                                    long long grad_stride, int accumulate, double Q, int d, int N,
                                                    int z_dim, long long z_batch_stride, PclObjSum fin) {
    extern __shared__ double lds[];
    __shared__ double red[8];
    const int n = 2 * d, tid = threadIdx.x;
    const int xo = x_offs[z_batch_stride ? 0 : b];
    const double *x = Z + (long long)b * z_batch_stride + (long long)(N - 1) * z_dim + xo;
    const double Qw = Q * (weights ? weights[b] : 1.0);
    // accumulate: 0 a per-member x_dim slot, overwritten | 1 the member's terminal block of the full gradient buffer, added to | 2 that
    // block, overwritten (pcl_ens_tail_kernel: the regulariser workgroup of the last knot runs beside this one and leaves the block alone)
    double *g = grad ? grad + (long long)b * grad_stride + (accumulate ? (long long)(N - 1) * z_dim + xo : 0) : nullptr;
    const bool add = accumulate == 1;
    if (ns <= 0) {
        double tr = 0.0, ti = 0.0;
        for (int e = tid; e < d * d; e += 256) {
            const int c = e / d, i = e - c * d;
            const double ur = x[c * n + i], ui = x[c * n + d + i], gr = goal[c * n + i], gi = goal[c * n + d + i];
            tr += gr * ur + gi * ui;
            ti += gr * ui - gi * ur;
        }
        tr = block_sum_256(tr, red);
        ti = block_sum_256(ti, red);
        const double inv = 1.0 / ((double)d * d);
        const double F = (tr * tr + ti * ti) * inv;
        const double sgn = (1.0 - F >= 0.0) ? 1.0 : -1.0;
        if (tid == 0 && value) {
        if (accumulate == 2)
            __hip_atomic_store(value + b, Qw * fabs(1.0 - F), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else
            value[b] = Qw * fabs(1.0 - F);
    }
        if (g) {
            for (int e = tid; e < d * d; e += 256) {
                const int c = e / d, i = e - c * d;
                const double gr = goal[c * n + i], gi = goal[c * n + d + i];
                const double v0 = -sgn * Qw * 2.0 * (tr * gr - ti * gi) * inv, v1 = -sgn * Qw * 2.0 * (tr * gi + ti * gr) * inv;
                if (add) {
                    g[c * n + i] += v0;
                    g[c * n + d + i] += v1;
                } else {
                    g[c * n + i] = v0;
                    g[c * n + d + i] = v1;
                }
            }
        }
        objective_finish(fin, value, red, accumulate == 2);
        return;
    }
    // ---- subspace (EmbeddedOperator) fidelity: ns x ns complex blocks in LDS: Us | Ug | M | W (re, im planes) ----
    const int nn = ns * ns;
    double *Ur = lds, *Ui = Ur + nn, *Gr = Ui + nn, *Gi = Gr + nn, *Mr = Gi + nn, *Mi = Mr + nn;
    const int n2 = 2 * ns;
    for (int e = tid; e < nn; e += 256) {
        const int c = e / ns, i = e - c * ns;  // column-major [i + ns*c]
        Ur[e] = x[sub[c] * n + sub[i]];
        Ui[e] = x[sub[c] * n + d + sub[i]];
        Gr[e] = goal[c * n2 + i];
        Gi[e] = goal[c * n2 + ns + i];
    }
    __syncthreads();
    double tr = 0.0, ti = 0.0, fro = 0.0;
    for (int e = tid; e < nn; e += 256) {  // M = Ug' Us
        const int c = e / ns, i = e - c * ns;
        double mr = 0.0, mi = 0.0;
        for (int k = 0; k < ns; ++k) {
            const double ar = Gr[k + ns * i], ai = -Gi[k + ns * i];  // conj(Ug[k,i])
            const double br = Ur[k + ns * c], bi = Ui[k + ns * c];
            mr += ar * br - ai * bi;
            mi += ar * bi + ai * br;
        }
        Mr[e] = mr;
        Mi[e] = mi;
        fro += mr * mr + mi * mi;
        if (i == c) {
            tr += mr;
            ti += mi;
        }
    }
    tr = block_sum_256(tr, red);
    ti = block_sum_256(ti, red);
    fro = block_sum_256(fro, red);  // (its leading barrier also completes M)
    const double inv = 1.0 / ((double)ns * (ns + 1));
    const double F = (fabs(fro) + tr * tr + ti * ti) * inv;
    const double sgn = (1.0 - F >= 0.0) ? 1.0 : -1.0;
    if (tid == 0 && value) {
        if (accumulate == 2)
            __hip_atomic_store(value + b, Qw * fabs(1.0 - F), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else
            value[b] = Qw * fabs(1.0 - F);
    }
    if (g) {
        if (!add)
            for (int e = tid; e < n * d; e += 256) g[e] = 0.0;
        __syncthreads();
        for (int e = tid; e < nn; e += 256) {  // W = Ug M ; dF = (2 W + 2 t Ug) / (ns (ns+1))
            const int c = e / ns, i = e - c * ns;
            double wr = 0.0, wi = 0.0;
            for (int k = 0; k < ns; ++k) {
                const double ar = Gr[i + ns * k], ai = Gi[i + ns * k];
                const double br = Mr[k + ns * c], bi = Mi[k + ns * c];
                wr += ar * br - ai * bi;
                wi += ar * bi + ai * br;
            }
            const double gr = Gr[e], gi = Gi[e];
            const double dr = (2.0 * wr + 2.0 * (tr * gr - ti * gi)) * inv, di = (2.0 * wi + 2.0 * (tr * gi + ti * gr)) * inv;
            g[sub[c] * n + sub[i]] += -sgn * Qw * dr;
            g[sub[c] * n + d + sub[i]] += -sgn * Qw * di;
        }
    }
    objective_finish(fin, value, red, accumulate == 2);
}

__global__ __launch_bounds__(256) void pcl_infidelity_kernel(const double *__restrict__ Z, const double *__restrict__ goal,
                                                             const int *__restrict__ sub, int ns,
                                                             const int *__restrict__ x_offs, const double *__restrict__ weights,
                                                             double *__restrict__ value, double *__restrict__ grad,
                                                             long long grad_stride, int accumulate, double Q, int d, int N,
                                                             int z_dim, long long z_batch_stride, PclObjSum fin) {
    pcl_infidelity_body((int)blockIdx.x, Z, goal, sub, ns, x_offs, weights, value, grad, grad_stride, accumulate, Q, d, N, z_dim, z_batch_stride, fin);
}

// ------------------------------------------------------------------------------------------
// Quadratic regularisers  J_r = 1/2 sum_k dt_k^p sum_i R_i v_{k,i}^2  (DirectTrajOpt's QuadraticRegularizer [EXT], used by
// every problem template: src/control/templates/smooth_pulse_problem.jl:249-251).  One workgroup per (knot, trajectory
// buffer): writes the knot's whole gradient row (zeros where no term applies), the knot's value into regval.
// reg table (device): per regulariser {off, dim, dt_power, R offset}; R values concatenated.
// ------------------------------------------------------------------------------------------
struct PclReg {
    int off, dim, pw, r0;
};
#define PCL_MAX_REGS 8

__device__ __forceinline__ void pcl_regularizer_body(const int k, const int tb, const double *__restrict__ Z, const PclReg *__restrict__ regs, int n_regs,
                                                     const double *__restrict__ Rv, double *__restrict__ grad,
                                                     double *__restrict__ regval, int N, int z_dim, int dt_off,
                                                     long long z_batch_stride, const PclObjSum fin, const double *__restrict__ member,
                                                     int skip_lo = 0, int skip_hi = 0) {
    // [skip_lo, skip_hi) (pcl_ens_tail_kernel): at the LAST knot these entries -- the terminal states, one contiguous run, written by the
    // infidelity workgroups of the same launch -- are left alone (the host has checked that the states tile the run and that no
    // regulariser covers them)
    __shared__ double red[8];
    __shared__ double sr[PCL_MAX_REGS];
    const int tid = threadIdx.x;
    const double *z = Z + (long long)tb * z_batch_stride + (long long)k * z_dim;
    double *g = grad ? grad + (long long)tb * (long long)z_dim * N + (long long)k * z_dim : nullptr;
    const double h = z[dt_off];
    double val = 0.0, gdt = 0.0;
    for (int r = 0; r < n_regs; ++r) {
        const PclReg R = regs[r];
        double s = 0.0;
        for (int i = tid; i < R.dim; i += 256) {
            const double v = z[R.off + i];
            s += Rv[R.r0 + i] * v * v;
        }
        s = block_sum_256(s, red);
        const double w = R.pw == 0 ? 1.0 : (R.pw == 1 ? h : h * h);
        val += 0.5 * w * s;
        gdt += R.pw == 0 ? 0.0 : (R.pw == 1 ? 0.5 * s : h * s);
        if (tid == 0) sr[r] = w;
    }
    __syncthreads();
    if (g) {  // the knot's whole row: zeros first (no memset launch before this kernel), then the entries that carry a term; terms
              // that overlap are added in regulariser order by the same thread (a thread owns entry i of every regulariser)
        if (skip_hi > skip_lo && k == N - 1) {
            for (int i = tid; i < skip_lo; i += 256) g[i] = 0.0;
            for (int i = skip_hi + tid; i < z_dim; i += 256) g[i] = 0.0;
        } else {
            const bool al = ((reinterpret_cast<unsigned long long>(g) & 15) == 0);
            int i0 = 0;
            if (al) {
                double2_t *g2 = reinterpret_cast<double2_t *>(g);
                for (int i = tid; i < (z_dim >> 1); i += 256) g2[i] = double2_t{0.0, 0.0};
                i0 = z_dim & ~1;
            }
            for (int i = i0 + tid; i < z_dim; i += 256) g[i] = 0.0;
        }
        __syncthreads();
        for (int r = 0; r < n_regs; ++r) {
            const PclReg R = regs[r];
            for (int i = tid; i < R.dim; i += 256)
                if (R.off + i != dt_off) {
                    double gi = 0.0;
                    for (int r2 = 0; r2 < n_regs; ++r2) {  // every regulariser covering this entry (normally just r)
                        const PclReg R2 = regs[r2];
                        const int j = R.off + i - R2.off;
                        if (j >= 0 && j < R2.dim) {
                            if (r2 < r) {  // an earlier regulariser already wrote the complete sum for this entry
                                gi = 0.0;
                                goto next_entry;
                            }
                            gi += sr[r2] * Rv[R2.r0 + j] * z[R.off + i];
                        }
                    }
                    g[R.off + i] = gi;
                next_entry:;
                }
        }
        if (tid == 0) {
            double gi = gdt;
            for (int r = 0; r < n_regs; ++r) {
                const PclReg R = regs[r];
                const int j = dt_off - R.off;
                if (j >= 0 && j < R.dim) gi += sr[r] * Rv[R.r0 + j] * z[dt_off];
            }
            g[dt_off] = gi;
        }
    }
    if (tid == 0) {
        if (fin.out)
            __hip_atomic_store(regval + (long long)tb * N + k, val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else
            regval[(long long)tb * N + k] = val;
    }
    objective_finish(fin, member, red, true);  // (fin.out NULL: a launch of its own, the infidelity launch behind it forms the sums)
}
__global__ __launch_bounds__(256) void pcl_regularizer_kernel(const double *__restrict__ Z, const PclReg *__restrict__ regs, int n_regs,
                                                              const double *__restrict__ Rv, double *__restrict__ grad,
                                                              double *__restrict__ regval, int N, int z_dim, int dt_off,
                                                              long long z_batch_stride) {
    pcl_regularizer_body((int)blockIdx.x, (int)blockIdx.y, Z, regs, n_regs, Rv, grad, regval, N, z_dim, dt_off, z_batch_stride, PclObjSum{}, nullptr);
}

// value[0] = sum_b member[b] + sum_k regval[k]   (MEMBERS), value[b] = member[b] + sum_k regval[b][k]   (TRAJ).
// One wave per output; fixed summation order (lane-strided partial sums, then a shuffle tree): bitwise repeatable.
__device__ __forceinline__ double wave_sum_strided(const double *__restrict__ v, int count) {
    double s = 0.0;
    for (int i = threadIdx.x & 63; i < count; i += 64) s += v[i];
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    return s;  // valid in lane 0
}
__global__ __launch_bounds__(64) void pcl_objective_sum_kernel(const double *__restrict__ member, const double *__restrict__ regval,
                                                               double *__restrict__ value, int batch, int N, int traj_mode) {
    if (traj_mode) {
        const int b = blockIdx.x;
        const double s = wave_sum_strided(regval + (long long)b * N, N);
        if (threadIdx.x == 0) value[b] = member[b] + s;
    } else {
        const double sm = wave_sum_strided(member, batch), sr = wave_sum_strided(regval, N);
        if (threadIdx.x == 0) value[0] = sm + sr;
    }
}

// ------------------------------------------------------------------------------------------
// The reduce payload of a sharded ensemble: for multipliers lam (NULL: lam = delta, i.e. the merit
// phi = 1/2 sum w |delta|^2 and its gradient) over this context's members b:
//     phi_k  = sum_b w_b c <lam_bk, delta_bk>           (c = 1/2 when lam = delta)
//     g[k,l] = sum_b w_b <d delta_bk / d u_l, lam_bk>,  g[k,m] = sum_b w_b <d delta_bk / d dt, lam_bk>
// i.e. J^T lam restricted to the SHARED variables (the only part of the Lagrangian gradient that needs other ranks).
// Two launches, fixed summation order (bitwise repeatable):
//   pcl_merit_part_kernel  one workgroup per (interval, member); wave w owns the jobs l = w, w+8, .. (l < m: d/du_l, l = m:
//                          d/ddt, l = m+1: <lam, delta>): one dot product over the member's tail block, 16-byte loads
//   pcl_merit_sum_kernel   sums the members in order with their weights, then phi over the intervals in order
// out: [phi | g_u (K*m, k-major) | g_dt (K)] per output set (one set; TRAJ mode: one per trajectory, nothing is shared).
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512) void pcl_merit_part_kernel(const double *__restrict__ delta, const double *__restrict__ lam,
                                                             const double *__restrict__ vals, double *__restrict__ part, int K,
                                                             int cols, int n, int m, long long jac_per, long long tail_off) {
    const int k = blockIdx.x, b = blockIdx.y, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long long xd = (long long)n * cols, bk = (long long)b * K + k;
    const double *dl = delta + bk * xd;
    const double *lm = lam ? lam + bk * xd : dl;
    const double *tail = vals + bk * jac_per + tail_off;
    const bool pairs = (n & 1) == 0;  // 16-byte loads when every run of n doubles starts 16-byte aligned (n even)
    for (int l = wave; l <= m + 1; l += 8) {
        double s = 0.0;
        if (l <= m) {
            if (pairs) {
                // four 16-byte loads of the tail and of the multipliers in flight per lane (one at a time, the loop is one
                // dependent round trip to HBM per iteration: 2.5 TB/s); the sums are added in a fixed order
                const int hn = n >> 1, tot = cols * hn;
                for (int e0 = lane; e0 < tot; e0 += 256) {
                    double2_t t[4], v[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int e = e0 + 64 * j;
                        t[j] = v[j] = double2_t{0.0, 0.0};
                        if (e < tot) {
                            const int c = e / hn, i = 2 * (e - c * hn);
                            t[j] = *reinterpret_cast<const double2_t *>(tail + ((long long)c * (m + 1) + l) * n + i);
                            v[j] = *reinterpret_cast<const double2_t *>(lm + c * n + i);
                        }
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j) s += t[j][0] * v[j][0] + t[j][1] * v[j][1];
                }
            } else {
                for (int e = lane; e < cols * n; e += 64) {
                    const int c = e / n, i = e - c * n;
                    s += tail[((long long)c * (m + 1) + l) * n + i] * lm[c * n + i];
                }
            }
        } else {
            for (long long e = lane; e < xd; e += 64) s += lm[e] * dl[e];
            s *= lam ? 1.0 : 0.5;
        }
        for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
        if (lane == 0) part[bk * (m + 2) + l] = s;
    }
}
// Fused variant (pcl_eval_jac_merit_dev): the fused kernel's matrix waves leave the m + 2 dot products per state column
// (pcol[((b*K + k)*cols + c)*(m+2) + l], formed while the column's vectors were in LDS).  ONE launch finishes the payload, every
// sum in a fixed order (bitwise repeatable): workgroup k adds the columns of interval k per (member, l) in column order, then
// the members with their weights in member order (same arithmetic as pcl_merit_sum_kernel) -> g_u[k,:], g_dt[k], phi_k; the
// workgroup that arrives last at the agent-scope ticket (acquire-release) adds phi_k over the intervals.
__device__ __forceinline__ void pcl_merit_finish_body(const int k, const double *__restrict__ pcol, const double *__restrict__ weights,
                                                      double *__restrict__ out, double *__restrict__ phik, unsigned int *ticket,
                                                      int batch, int K, int cols, int m, int traj_mode, bool light = false) {
    // light (pcl_ens_tail_kernel): what the last arriver reads -- phi_k -- is published with agent-scope atomic stores; every thread waits
    // for its own stores, the ticket is relaxed (no write-back of the whole L2 per workgroup beside the gradient rows of the same launch)
    extern __shared__ double part[];  // [batch][m + 2]
    __shared__ int last;
    const int tid = threadIdx.x, m2 = m + 2;
    const long long set_len = 1 + (long long)K * m + K;
    for (int e = tid; e < batch * m2; e += 256) {
        const int b = e / m2, l = e - b * m2;
        const double *src = pcol + ((long long)b * K + k) * cols * m2 + l;
        double s = 0.0;
        for (int c0 = 0; c0 < cols; c0 += 8) {  // eight columns in flight, added in column order
            double v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = c0 + j < cols ? src[(long long)(c0 + j) * m2] : 0.0;
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (c0 + j < cols) s += v[j];
        }
        part[e] = s;
    }
    __syncthreads();
    if (traj_mode) {
        for (int e = tid; e < batch * m2; e += 256) {
            const int b = e / m2, l = e - b * m2;
            double *o = out + (long long)b * set_len;
            const double t = 0.0 + (weights ? weights[b] : 1.0) * part[e];
            if (l < m)
                o[1 + (long long)k * m + l] = t;
            else if (l == m)
                o[1 + (long long)K * m + k] = t;
            else
                __hip_atomic_store(phik + (long long)b * K + k, t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    } else if (tid < m2) {
        const int l = tid;
        double t = 0.0;
        for (int b = 0; b < batch; ++b) t += (weights ? weights[b] : 1.0) * part[b * m2 + l];
        if (l < m)
            out[1 + (long long)k * m + l] = t;
        else if (l == m)
            out[1 + (long long)K * m + k] = t;
        else
            __hip_atomic_store(phik + k, t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (light) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        const unsigned int t = light ? __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                     : __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        last = (t == (unsigned)K - 1u);
    }
    __syncthreads();
    if (!last) return;
    if (tid == 0) __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    const int sets = traj_mode ? batch : 1;
    for (int set = tid >> 6; set < sets; set += 4) {  // one wave per output set: lane-strided partial sums, then a shuffle tree
        const double *ph = phik + (long long)set * K;
        double s = 0.0;
        for (int i = tid & 63; i < K; i += 64) s += __hip_atomic_load(ph + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
        if ((tid & 63) == 0) out[(long long)set * set_len] = s;
    }
}
__global__ __launch_bounds__(256) void pcl_merit_finish_kernel(const double *__restrict__ pcol, const double *__restrict__ weights,
                                                              double *__restrict__ out, double *__restrict__ phik, unsigned int *ticket,
                                                              int batch, int K, int cols, int m, int traj_mode) {
    pcl_merit_finish_body((int)blockIdx.x, pcol, weights, out, phik, ticket, batch, K, cols, m, traj_mode);
}
// The tail of an ensemble step in ONE launch (pcl_eval_jac_merit_objective_dev): what pcl_objective_dev (regulariser + infidelity
// launches) and the payload's finish launch do, as three ranges of workgroups of one grid -- [0, nbuf N) regulariser rows,
// [.., + batch) terminal infidelities, [.., + K) payload intervals.  Nothing here waits for anything: the objective's final sums are
// formed by whichever regulariser / infidelity workgroup arrives last at their ticket, the payload's by its own last arriver; every sum
// keeps its fixed order, so the outputs are bit for bit those of the separate launches.  Three small launches on a GPU whose every CU
// the fused kernel has just occupied cost 8-9 us each, most of it dispatch.
struct PclTailArgs {
    // regulariser rows
    const double *Z;
    const PclReg *regs;
    int n_regs;
    const double *Rv;
    double *grad, *regval;
    int N, z_dim, dt_off, nbuf;
    long long z_batch_stride;
    // infidelities
    const double *goal;
    const int *sub;
    int ns;
    const int *x_offs;
    const double *weights;
    double *member;
    long long grad_stride;
    double Q;
    int d, batch;
    PclObjSum fin;
    // payload
    const double *pcol;
    double *out, *phik;
    unsigned int *mticket;
    int K, cols, m, traj_mode;
    int skip_lo, skip_hi;  // the terminal states' entries of a gradient row (regulariser rows leave them to the infidelity workgroups)
};
__global__ __launch_bounds__(256) void pcl_ens_tail_kernel(const PclTailArgs a) {
    int j = (int)blockIdx.x;
    const int n_reg = a.nbuf * a.N;
    if (j < n_reg) {
        pcl_regularizer_body(j % a.N, j / a.N, a.Z, a.regs, a.n_regs, a.Rv, a.grad, a.regval, a.N, a.z_dim, a.dt_off, a.z_batch_stride, a.fin, a.member, a.skip_lo, a.skip_hi);
        return;
    }
    j -= n_reg;
    if (j < a.batch) {
        pcl_infidelity_body(j, a.Z, a.goal, a.sub, a.ns, a.x_offs, a.weights, a.member, a.grad, a.grad_stride, 2, a.Q, a.d, a.N, a.z_dim, a.z_batch_stride, a.fin);
        return;
    }
    pcl_merit_finish_body(j - a.batch, a.pcol, a.weights, a.out, a.phik, a.mticket, a.batch, a.K, a.cols, a.m, a.traj_mode, true);
}
__global__ __launch_bounds__(1024) void pcl_merit_sum_kernel(const double *__restrict__ part, const double *__restrict__ weights,
                                                            double *__restrict__ out, double *__restrict__ phik, int batch, int K, int m,
                                                            int traj_mode) {
    const int set = blockIdx.x;  // TRAJ mode: one output set per trajectory; MEMBERS mode: one set, summed over the members
    const int b_lo = traj_mode ? set : 0, b_hi = traj_mode ? set + 1 : batch;
    double *o = out + (long long)set * (1 + (long long)K * m + K);
    double *ph = phik + (long long)set * K;
    for (int e = threadIdx.x; e < K * (m + 2); e += 1024) {
        const int k = e / (m + 2), l = e - k * (m + 2);
        double t = 0.0;
        for (int b0 = b_lo; b0 < b_hi; b0 += 8) {  // eight members' partials in flight, added in member order
            double v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = b0 + j < b_hi ? (weights ? weights[b0 + j] : 1.0) * part[((long long)(b0 + j) * K + k) * (m + 2) + l] : 0.0;
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (b0 + j < b_hi) t += v[j];
        }
        if (l < m)
            o[1 + (long long)k * m + l] = t;
        else if (l == m)
            o[1 + (long long)K * m + k] = t;
        else
            ph[k] = t;
    }
    __syncthreads();
    if (threadIdx.x < 64) {
        const double t = wave_sum_strided(ph, K);
        if (threadIdx.x == 0) o[0] = t;
    }
}
