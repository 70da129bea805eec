// pcl_kernel_fused_small.hpp -- residual (+ Jacobian) for SMALL systems: n = 2d <= 16 rows, at most 8 state columns (BASELINE config 1: d = 2,
// config 2: the CNOT problem, d = 4; every two-qubit and multilevel-transmon problem of the reference's docs, d <= 8), ANY diagonal Pade
// order.
//
// At this size a launch is bound by its own start-up, not by memory or arithmetic: the generic kernel 1 staged tables through LDS in
// dependent rounds (7.8 us for 0.6 MB of output).  Here ONE WAVE owns an interval: every global load of the interval is issued in its
// first instructions (one round trip), the generators live in LD x LD LDS words each, lane (i, k0) holds the entries (i, k0 + KS r) of G(u),
// of its powers and of -B^+ / B^-, and rows i of the state columns k0 + KS r of every chain of the recursion
//     W <- c_j Y_j + h G W (delta = W_0),  V <- j c_j Y_j + h G V (d delta / dh),  dW_l <- h (G_l W_old + G dW_l) (d delta / du_l),
//     B^+- = sum_j c_j (+-h)^j G^j,  Y_j = D (j even) | -S (j odd),  D = X_{k+1} - X_k,  S = X_{k+1} + X_k
// (the recursion of pcl_kernel_fused_sparse.hpp, restated from oracle/pade_oracle.py: pade_residual / pade_jacobian_values); the lane's
// row of G stays in registers, operand columns are LDS broadcasts.  Two instances: LD = 8 (n <= 8: one entry and one state element per
// lane) and LD = 16 (n <= 16: four entries, two state elements).  A workgroup is one wave: no atomics, bitwise repeatable.
#pragma once

#define PCL_SM_M 8  // drives (the loops over them are unrolled: registers, not scratch)

template <bool JAC, int LD>
__global__ __launch_bounds__(64) void pcl_fused_small_kernel(const KParams p, const double *__restrict__ Gd /* dense drives: [m][n*n] column-major */) {
    constexpr int KS = 64 / LD;             // columns covered by the 64 lanes at once
    constexpr int R = LD / KS;              // matrix entries per lane (columns k0 + KS r)
    constexpr int RS = (8 + KS - 1) / KS;   // state elements per lane (at most 8 state columns)
    __shared__ double Gs[LD * LD], Ps[LD * LD], GL[JAC ? PCL_SM_M : 1][LD * LD];
    __shared__ double Ws[LD * 8], Vs[JAC ? LD * 8 : 1], dWs[JAC ? PCL_SM_M : 1][LD * 8];
    const int lane = threadIdx.x;
    const int n = p.n, m = p.m, q = p.q, C = p.cols;
    const int i = lane & (LD - 1), k0 = lane / LD;
    const long long nn = (long long)n * n, xd = (long long)n * C;
    const long long blk = p.compact ? nn : (long long)C * nn;  // size of the -B^+ / of the B^- segment
    const int n_items = p.batch * p.K;
    for (int item = blockIdx.x; item < n_items; item += gridDim.x) {
        const int k = item % p.K, b = item / p.K;
        // ---- every global load of the interval, issued together --------------------------------------------------------------
        const double *zc = p.Z + (long long)b * p.z_batch_stride + (long long)k * p.z_dim;
        const double *xk = zc + (p.x_off0 >= 0 ? p.x_off0 : p.x_offs[p.z_batch_stride ? 0 : b]);  // (a load the state loads would wait for)
        const double h = zc[p.dt_off];
        double u[PCL_SM_M], gl[R][PCL_SM_M], g[R], dv[RS], sv[RS];
#pragma unroll
        for (int l = 0; l < PCL_SM_M; ++l) u[l] = l < m ? zc[p.u_off + l] : 0.0;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int kc = k0 + KS * r;
            const bool ent = i < n && kc < n;
#pragma unroll
            for (int l = 0; l < PCL_SM_M; ++l) gl[r][l] = (l < m && ent) ? Gd[(long long)l * nn + i + n * kc] : 0.0;
            g[r] = ent ? p.G0[(p.g0_batch_stride ? (long long)b * p.g0_batch_stride : 0) + i + n * kc] : 0.0;
        }
#pragma unroll
        for (int r = 0; r < RS; ++r) {
            const int c = k0 + KS * r;
            const bool st = i < n && c < C;
            const double x0 = st ? xk[i + n * c] : 0.0, x1 = st ? xk[p.z_dim + i + n * c] : 0.0;
            dv[r] = x1 - x0;
            sv[r] = x1 + x0;
        }
        // ---- G(u_k) and the drives into LDS; this lane's row of G into registers ----------------------------------------------------
#pragma unroll
        for (int r = 0; r < R; ++r) {
#pragma unroll
            for (int l = 0; l < PCL_SM_M; ++l)
                if (l < m) {
                    g[r] = __builtin_fma(u[l], gl[r][l], g[r]);
                    if (JAC) GL[l][lane + 64 * r] = gl[r][l];
                }
            Gs[lane + 64 * r] = g[r];  // (= i + LD (k0 + KS r))
        }
        __syncthreads();
        double gr[LD];
#pragma unroll
        for (int t = 0; t < LD; ++t) gr[t] = Gs[i + LD * t];
        auto Gx = [&](const double *X, int col) {  // (G X)[i, col] for a matrix X in LDS
            double a0 = 0.0, a1 = 0.0;
#pragma unroll
            for (int t = 0; t < LD; t += 2) {
                a0 = __builtin_fma(gr[t], X[t + LD * col], a0);
                a1 = __builtin_fma(gr[t + 1], X[t + 1 + LD * col], a1);
            }
            return a0 + a1;
        };
        auto GLx = [&](int l, const double *X, int col) {  // (G_l X)[i, col]
            double a0 = 0.0, a1 = 0.0;
#pragma unroll
            for (int t = 0; t < LD; t += 2) {
                a0 = __builtin_fma(GL[l][i + LD * t], X[t + LD * col], a0);
                a1 = __builtin_fma(GL[l][i + LD * (t + 1)], X[t + 1 + LD * col], a1);
            }
            return a0 + a1;
        };
        // ---- the chains, Horner from level q down ----------------------------------------------------------------------------------
        double w[RS], v[RS], dw[RS][PCL_SM_M];
#pragma unroll
        for (int r = 0; r < RS; ++r) {
            const double yq = (q & 1) ? -sv[r] : dv[r];
            w[r] = p.pc[q] * yq;
            v[r] = (double)q * p.pc[q] * yq;
            Ws[lane + 64 * r] = w[r];
            if (JAC) Vs[lane + 64 * r] = v[r];
        }
        __syncthreads();
        if (JAC) {
#pragma unroll
            for (int r = 0; r < RS; ++r)
#pragma unroll
                for (int l = 0; l < PCL_SM_M; ++l)
                    if (l < m) dw[r][l] = h * GLx(l, Ws, k0 + KS * r);
#pragma unroll
            for (int r = 0; r < RS; ++r)
#pragma unroll
                for (int l = 0; l < PCL_SM_M; ++l)
                    if (l < m) dWs[l][lane + 64 * r] = dw[r][l];
        }
        // the powers of G and the blocks: entries (i, k0 + KS r) of -B^+ and of B^-
        double bp[R], bm[R], hp = 1.0, hm = 1.0;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            bp[r] = (i == k0 + KS * r) ? -1.0 : 0.0;
            bm[r] = (i == k0 + KS * r) ? 1.0 : 0.0;
        }
        for (int s = 0; s < q; ++s) {
            const int j = q - 1 - s;
            __syncthreads();  // W, V, dW_l of the level above are in their tiles
            if (JAC && s >= 1) {
#pragma unroll
                for (int r = 0; r < RS; ++r) {
                    double t_[PCL_SM_M];
#pragma unroll
                    for (int l = 0; l < PCL_SM_M; ++l)
                        if (l < m) t_[l] = h * (GLx(l, Ws, k0 + KS * r) + Gx(dWs[l], k0 + KS * r));
#pragma unroll
                    for (int l = 0; l < PCL_SM_M; ++l)
                        if (l < m) dw[r][l] = t_[l];
                }
            }
#pragma unroll
            for (int r = 0; r < RS; ++r) {
                const double yj = (j & 1) ? -sv[r] : dv[r];
                const double gw = Gx(Ws, k0 + KS * r), gv = JAC ? Gx(Vs, k0 + KS * r) : 0.0;
                w[r] = __builtin_fma(h, gw, p.pc[j] * yj);
                v[r] = __builtin_fma(j == 0 ? 1.0 : h, gv, (double)j * p.pc[j] * yj);
            }
            double pw[R];
            if (JAC) {  // power s + 1 of G: P_1 = G, P_{s+1} = G P_s
                hp *= h;
                hm *= -h;
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    pw[r] = s == 0 ? g[r] : Gx(Ps, k0 + KS * r);
                    bp[r] = __builtin_fma(-p.pc[s + 1] * hp, pw[r], bp[r]);
                    bm[r] = __builtin_fma(p.pc[s + 1] * hm, pw[r], bm[r]);
                }
            }
            __syncthreads();  // every read of the tiles above is done
            if (JAC) {
#pragma unroll
                for (int r = 0; r < R; ++r) Ps[lane + 64 * r] = pw[r];
#pragma unroll
                for (int r = 0; r < RS; ++r) {
#pragma unroll
                    for (int l = 0; l < PCL_SM_M; ++l)
                        if (l < m && s >= 1) dWs[l][lane + 64 * r] = dw[r][l];
                    Vs[lane + 64 * r] = v[r];
                }
            }
#pragma unroll
            for (int r = 0; r < RS; ++r) Ws[lane + 64 * r] = w[r];
        }
        // ---- outputs --------------------------------------------------------------------------------------------------------------
        const long long bk = (long long)b * p.K + k;
        double *o = JAC ? p.jac + bk * p.jac_per : nullptr;
#pragma unroll
        for (int r = 0; r < RS; ++r) {
            const int c = k0 + KS * r;
            if (i < n && c < C) {
                if (p.delta) p.delta[bk * xd + i + (long long)n * c] = w[r];
                if (JAC) {
                    double *tc = o + 2 * blk + (long long)c * (m + 1) * n + i;  // this column's (m + 1) n tail block
#pragma unroll
                    for (int l = 0; l < PCL_SM_M; ++l)
                        if (l < m) tc[(long long)l * n] = dw[r][l];
                    tc[(long long)m * n] = v[r];
                }
            }
        }
        if (JAC) {
            const int copies = p.compact ? 1 : C;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int kc = k0 + KS * r;
                if (i < n && kc < n)
                    for (int c = 0; c < copies; ++c) {
                        o[c * nn + i + n * kc] = bp[r];
                        o[blk + c * nn + i + n * kc] = bm[r];
                    }
            }
        }
        __syncthreads();  // (the tiles are rewritten by this wave's next interval)
    }
}
