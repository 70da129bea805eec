// pcl_kernel_fused_small.hpp -- residual (+ Jacobian) for SMALL systems: n = 2d <= 8 rows, n * cols <= 64 (BASELINE config 1: d = 2,
// config 2: the CNOT problem, d = 4; every two-qubit problem of the reference's docs), ANY diagonal Pade order.
//
// At this size a launch is bound by its own start-up, not by memory or arithmetic: the generic kernel 1 staged tables through LDS in
// dependent rounds (7.8 us for 0.6 MB of output).  Here ONE WAVE owns an interval: every global load of the interval is issued in its
// first instructions (one round trip), the generators live in 64 LDS words each, lane (i, k) holds entry (i, k) of G(u), of its powers
// and of -B^+ / B^-, lane (i, c) holds row i of state column c of every chain of the recursion
//     W <- c_j Y_j + h G W (delta = W_0),  V <- j c_j Y_j + h G V (d delta / dh),  dW_l <- h (G_l W_old + G dW_l) (d delta / du_l),
//     B^+- = sum_j c_j (+-h)^j G^j,  Y_j = D (j even) | -S (j odd),  D = X_{k+1} - X_k,  S = X_{k+1} + X_k
// (the recursion of pcl_kernel_fused_sparse.hpp, restated from oracle/pade_oracle.py: pade_residual / pade_jacobian_values).
// No workgroup barrier (a workgroup is one wave; LDS operations of a wave complete in order), no atomics, bitwise repeatable.
#pragma once

#define PCL_SM_LD 8  // leading dimension of every LDS matrix (rows i < n <= 8)
#define PCL_SM_M 8   // drives (the loops over them are unrolled: registers, not scratch)

template <bool JAC>
__global__ __launch_bounds__(64) void pcl_fused_small_kernel(const KParams p, const double *__restrict__ Gd /* dense drives: [m][n*n] column-major */) {
    __shared__ double Gs[64], Ps[64], GL[PCL_SM_M][64];
    __shared__ double Ds[64], Ss[64], Ws[64], Vs[64], dWs[PCL_SM_M][64];
    const int lane = threadIdx.x;
    const int n = p.n, m = p.m, q = p.q, C = p.cols;
    const int i = lane & 7, kc = lane >> 3;          // matrix entry (i, kc) / state element (row i, column kc)
    const bool ent = i < n && kc < n, st = i < n && kc < C;
    const long long nn = (long long)n * n, xd = (long long)n * C;
    const long long blk = p.compact ? nn : (long long)C * nn;  // size of the -B^+ / of the B^- segment
    const int n_items = p.batch * p.K;
    for (int item = blockIdx.x; item < n_items; item += gridDim.x) {
        const int k = item % p.K, b = item / p.K;
        // ---- every global load of the interval, issued together --------------------------------------------------------------
        const double *zc = p.Z + (long long)b * p.z_batch_stride + (long long)k * p.z_dim;
        const double *xk = zc + (p.x_off0 >= 0 ? p.x_off0 : p.x_offs[p.z_batch_stride ? 0 : b]);  // (a load the state loads would wait for)
        const double h = zc[p.dt_off];
        double u[PCL_SM_M], gl[PCL_SM_M];
#pragma unroll
        for (int l = 0; l < PCL_SM_M; ++l) {
            u[l] = l < m ? zc[p.u_off + l] : 0.0;
            gl[l] = (l < m && ent) ? Gd[(long long)l * nn + i + n * kc] : 0.0;
        }
        double g = ent ? p.G0[(p.g0_batch_stride ? (long long)b * p.g0_batch_stride : 0) + i + n * kc] : 0.0;
        const double x0 = st ? xk[i + n * kc] : 0.0, x1 = st ? xk[p.z_dim + i + n * kc] : 0.0;
        // ---- G(u_k), the drives, D, S into LDS; this lane's row of G into registers ----------------------------------------------
#pragma unroll
        for (int l = 0; l < PCL_SM_M; ++l)
            if (l < m) {
                g = __builtin_fma(u[l], gl[l], g);
                if (JAC) GL[l][lane] = gl[l];
            }
        Gs[lane] = g;
        const double dv = x1 - x0, sv = x1 + x0;
        Ds[lane] = dv;
        Ss[lane] = sv;
        __syncthreads();
        double gr[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) gr[t] = Gs[i + PCL_SM_LD * t];
        auto Gx = [&](const double *X) {  // (G X)[i, kc] for a matrix X in LDS
            double a0 = 0.0, a1 = 0.0;
#pragma unroll
            for (int t = 0; t < 8; t += 2) {
                a0 = __builtin_fma(gr[t], X[t + PCL_SM_LD * kc], a0);
                a1 = __builtin_fma(gr[t + 1], X[t + 1 + PCL_SM_LD * kc], a1);
            }
            return a0 + a1;
        };
        auto GLx = [&](int l, const double *X) {  // (G_l X)[i, kc]
            double a0 = 0.0, a1 = 0.0;
#pragma unroll
            for (int t = 0; t < 8; t += 2) {
                a0 = __builtin_fma(GL[l][i + PCL_SM_LD * t], X[t + PCL_SM_LD * kc], a0);
                a1 = __builtin_fma(GL[l][i + PCL_SM_LD * (t + 1)], X[t + 1 + PCL_SM_LD * kc], a1);
            }
            return a0 + a1;
        };
        // ---- the chains, Horner from level q down ----------------------------------------------------------------------------------
        const double yq = (q & 1) ? -sv : dv;
        double w = p.pc[q] * yq, v = (double)q * p.pc[q] * yq, dw[PCL_SM_M];
        Ws[lane] = w;
        if (JAC) Vs[lane] = v;
        __syncthreads();
        if (JAC) {
#pragma unroll
            for (int l = 0; l < PCL_SM_M; ++l)
                if (l < m) dw[l] = h * GLx(l, Ws);
#pragma unroll
            for (int l = 0; l < PCL_SM_M; ++l)
                if (l < m) dWs[l][lane] = dw[l];
        }
        // the powers of G and the blocks: entry (i, kc) of -B^+ and of B^-
        double bp = (i == kc) ? -1.0 : 0.0, bm = (i == kc) ? 1.0 : 0.0, hp = 1.0, hm = 1.0;
        for (int s = 0; s < q; ++s) {
            const int j = q - 1 - s;
            __syncthreads();  // W, V, dW_l of the level above are in their tiles
            const double yj = (j & 1) ? -sv : dv;
            if (JAC && s >= 1) {
                double t_[PCL_SM_M];
#pragma unroll
                for (int l = 0; l < PCL_SM_M; ++l)
                    if (l < m) t_[l] = h * (GLx(l, Ws) + Gx(dWs[l]));
#pragma unroll
                for (int l = 0; l < PCL_SM_M; ++l)
                    if (l < m) dw[l] = t_[l];
            }
            const double gw = Gx(Ws), gv = JAC ? Gx(Vs) : 0.0;
            w = __builtin_fma(h, gw, p.pc[j] * yj);
            v = __builtin_fma(j == 0 ? 1.0 : h, gv, (double)j * p.pc[j] * yj);
            if (JAC) {  // power s + 1 of G: P_1 = G, P_{s+1} = G P_s
                const double pw = s == 0 ? g : Gx(Ps);
                hp *= h;
                hm *= -h;
                bp = __builtin_fma(-p.pc[s + 1] * hp, pw, bp);
                bm = __builtin_fma(p.pc[s + 1] * hm, pw, bm);
                __syncthreads();  // every read of the tiles above is done
                Ps[lane] = pw;
#pragma unroll
                for (int l = 0; l < PCL_SM_M; ++l)
                    if (l < m && s >= 1) dWs[l][lane] = dw[l];
                Vs[lane] = v;
            } else {
                __syncthreads();
            }
            Ws[lane] = w;
        }
        // ---- outputs --------------------------------------------------------------------------------------------------------------
        const long long bk = (long long)b * p.K + k;
        if (p.delta && st) p.delta[bk * xd + i + (long long)n * kc] = w;
        if (JAC) {
            double *o = p.jac + bk * p.jac_per;
            if (ent) {
                const int copies = p.compact ? 1 : C;
                for (int c = 0; c < copies; ++c) {
                    o[c * nn + i + n * kc] = bp;
                    o[blk + c * nn + i + n * kc] = bm;
                }
            }
            if (st) {
                double *tc = o + 2 * blk + (long long)kc * (m + 1) * n + i;  // this column's (m + 1) n tail block
#pragma unroll
                for (int l = 0; l < PCL_SM_M; ++l)
                    if (l < m) tc[(long long)l * n] = dw[l];
                tc[(long long)m * n] = v;
            }
        }
        __syncthreads();  // (the tiles are rewritten by this wave's next interval)
    }
}
