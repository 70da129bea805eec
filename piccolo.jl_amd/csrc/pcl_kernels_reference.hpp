// pcl_kernels_reference.hpp -- the single-role kernels: version 1 of the fused residual + Jacobian kernel (one workgroup per
// item; MFMA or plain VALU) and the general-order kernel (diagonal Pade orders 2..10).
#pragma once

// ------------------------------------------------------------------------------------------
// Fused residual + Jacobian kernel.
// LDS map (doubles):  G [LD*n] | G2 [LD*n] | M1 [LD*(2+m)*nc] | W1 [LD*(2+m)*nc] | G2D [LD*nc] | T [LD*nc] | us[8+m]
//   M1 = [S | D | G_1 D .. G_m D] (slice columns), W1 = G*M1 = [GS | GD | G(G_l D)].
// ------------------------------------------------------------------------------------------
template <bool JAC, bool MFMA>
__global__ __launch_bounds__(256) void pcl_fused_kernel(const KParams p) {
    extern __shared__ double lds[];
    const int n = p.n, d = p.cols, m = p.m, LD = p.LD, nc = p.nc;  // d: state columns here (no iso shortcut in this kernel)
    const int tid = threadIdx.x, nth = blockDim.x;

    const int bid = blockIdx.x;
    const int s = bid % p.S;
    const int k = (bid / p.S) % p.K;
    const int b = bid / (p.S * p.K);
    const int c0 = s * nc;
    const int nce = min(nc, d - c0);  // columns actually owned by this slice

    const int ncols1 = JAC ? (2 + m) * nc : 2 * nc;
    double *G = lds;
    double *G2 = G + LD * n;
    double *M1 = G2 + (JAC ? LD * n : 0);
    double *W1 = M1 + LD * ncols1;
    double *G2D = W1 + LD * ncols1;
    double *T = G2D + LD * nc;
    double *us = T + LD * nc;

    const double *Zb = p.Z + (long long)b * p.z_batch_stride;
    const double *zk = Zb + (long long)k * p.z_dim;
    const double *zn = zk + p.z_dim;
    const int x_off = p.x_offs[p.z_batch_stride ? 0 : b];
    const double h = zk[p.dt_off];
    const double c1 = 0.5 * h, c2 = h * h * (1.0 / 12.0);
    const long long xd = (long long)n * d;

    build_G(p, p.G0 + (long long)b * p.g0_batch_stride, zk, G, us);

    // S and D for the slice's columns (unused trailing columns are zero)
    for (int e = tid; e < nc * n; e += nth) {
        const int c = e / n, i = e % n;
        double xs = 0.0, xdv = 0.0;
        if (c < nce) {
            const double xn = zn[x_off + (c0 + c) * n + i], xc = zk[x_off + (c0 + c) * n + i];
            xs = xn + xc;
            xdv = xn - xc;
        }
        M1[i + LD * c] = xs;
        M1[i + LD * (nc + c)] = xdv;
    }
    __syncthreads();

    if (JAC) {
        // G_l D via the CSR rows of G_l
        const double *Dm = M1 + LD * nc;
        for (int e = tid; e < m * nc * n; e += nth) {
            const int i = e % n, c = (e / n) % nc, l = e / (n * nc);
            const int *rp = p.csr_ptr + l * (n + 1);
            double acc = 0.0;
            for (int q = rp[i]; q < rp[i + 1]; ++q) acc += p.csr_val[q] * Dm[p.csr_col[q] + LD * c];
            M1[i + LD * ((2 + l) * nc + c)] = acc;
        }
        __syncthreads();
        gemm_lds<MFMA, false>(G, LD, G, LD, G2, LD, n, n, n);
    }
    gemm_lds<MFMA, false>(G, LD, M1, LD, W1, LD, n, ncols1, n);
    __syncthreads();

    // pass 2: G2D = G * (G D);  T = -c1 S + c2 G D
    gemm_lds<MFMA, false>(G, LD, W1 + LD * nc, LD, G2D, LD, n, nc, n);
    if (JAC) {
        for (int e = tid; e < nc * n; e += nth) {
            const int c = e / n, i = e % n;
            T[i + LD * c] = -c1 * M1[i + LD * c] + c2 * W1[i + LD * (nc + c)];
        }
    }
    __syncthreads();

    // ---- column outputs -------------------------------------------------------------------
    const long long bk = (long long)b * p.K + k;
    double *jb = JAC ? p.jac + bk * p.jac_per : nullptr;
    const long long blk = p.compact ? (long long)n * n : (long long)d * n * n;  // size of seg 0 / seg 1
    for (int e = tid; e < nce * n; e += nth) {
        const int c = e / n, i = e % n;
        const double gs = W1[i + LD * c], g2d = G2D[i + LD * c];
        const long long r = (long long)(c0 + c) * n + i;
        if (p.delta) p.delta[bk * xd + r] = M1[i + LD * (nc + c)] - c1 * gs + c2 * g2d;
        if (JAC) jb[2 * blk + ((long long)(c0 + c) * (m + 1) + m) * n + i] = -0.5 * gs + (h * (1.0 / 6.0)) * g2d;
    }
    if (JAC) {
        for (int e = tid; e < m * nce * n; e += nth) {
            const int i = e % n, c = (e / n) % nce, l = e / (n * nce);
            const int *rp = p.csr_ptr + l * (n + 1);
            double acc = 0.0;
            for (int q = rp[i]; q < rp[i + 1]; ++q) acc += p.csr_val[q] * T[p.csr_col[q] + LD * c];
            jb[2 * blk + ((long long)(c0 + c) * (m + 1) + l) * n + i] = acc + c2 * W1[i + LD * ((2 + l) * nc + c)];
        }

        // ---- replicated diagonal blocks: stream -B^+ and B^- ---------------------------------
        // pair index q covers flat column-major positions 2q, 2q+1 (same column since n is even)
        const int half = (n * n) >> 1;
        int cbeg = c0, cend = c0 + nce;
        if (p.compact) {  // unique blocks only: slice 0 writes the single copy
            cbeg = 0;
            cend = (s == 0) ? 1 : 0;
        }
        for (int q = tid; q < half; q += nth) {
            const int pos = 2 * q;
            const int i = pos % n, j = pos / n;
            const double g0 = G[i + LD * j], g1 = G[i + 1 + LD * j];
            const double h0 = G2[i + LD * j], h1 = G2[i + 1 + LD * j];
            const double id0 = (i == j) ? 1.0 : 0.0, id1 = (i + 1 == j) ? 1.0 : 0.0;
            double bp0, bp1, bm0, bm1;
            bpm_entry(id0, c1, c2, g0, h0, bp0, bm0);
            bpm_entry(id1, c1, c2, g1, h1, bp1, bm1);
            for (int c = cbeg; c < cend; ++c) {
                double *o0 = jb + (long long)c * n * n + pos;
                store2(o0, bp0, bp1, p.nt);
                store2(o0 + blk, bm0, bm1, p.nt);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// General-order kernel: diagonal Pade orders p = 2q, q <= 5 (p = 2, 6, 8, 10; p = 4 only as a cross-check of the
// specialised kernels).  One workgroup per (b, k, slice of nc columns); correctness first, no wave specialisation.
// With Y_j = (-1)^j X_{k+1} - X_k (D for even j, -S for odd j) and c_j the Pade coefficients:
//   residual (Horner)   W_q = c_q Y_q,  W_j = c_j Y_j + h G W_{j+1},  delta = W_0
//   d/dh                V_q = q c_q Y_q, V_j = j c_j Y_j + h G V_{j+1} (j >= 1),  d delta/dh = G V_1
//   d/du_l              dW_q = 0,  dW_j = h (G_l W_{j+1} + G dW_{j+1}),  d delta/du_l = dW_0
//   blocks              B^{+-} = sum_j c_j (+-h)^j G^j, powers by repeated products, sums in registers
// LDS map (doubles): G | Pa | Pb (JAC) | -S | D | W_0..W_q | V (2, JAC) | dWa, dWb (m each, JAC) | us      (column blocks LD*nc)
// ------------------------------------------------------------------------------------------
template <bool JAC>
__global__ __launch_bounds__(512) void pcl_pade_kernel(const KParams p) {  // 512 threads (256 with option general_threads; 1024 measured no better)
    extern __shared__ double lds[];
    const int n = p.n, d = p.cols, m = p.m, LD = p.LD, nc = p.nc, q = p.q;
    const int tid = threadIdx.x, nth = blockDim.x;
    const int bid = blockIdx.x;
    const int s = bid % p.S;
    const int k = (bid / p.S) % p.K;
    const int b = bid / (p.S * p.K);
    const int c0 = s * nc;
    const int nce = min(nc, d - c0);
    const int LDc = LD * nc;

    double *G = lds;
    double *Pa = G + LD * n;
    double *Pb = Pa + (JAC ? LD * n : 0);
    double *Sm = Pb + (JAC ? LD * n : 0);  // -S
    double *Dm = Sm + LDc;
    double *W = Dm + LDc;  // W_j at W + j*LDc
    double *V = W + (q + 1) * LDc;
    double *dWa = V + (JAC ? 2 * LDc : 0);
    double *dWb = dWa + (JAC ? m * LDc : 0);
    double *us = dWb + (JAC ? m * LDc : 0);

    const double *Zb = p.Z + (long long)b * p.z_batch_stride;
    const double *zk = Zb + (long long)k * p.z_dim;
    const double *zn = zk + p.z_dim;
    const int x_off = p.x_offs[p.z_batch_stride ? 0 : b];
    const double h = zk[p.dt_off];
    const long long xd = (long long)n * d;

    build_G(p, p.G0 + (long long)b * p.g0_batch_stride, zk, G, us);
    for (int e = tid; e < nc * n; e += nth) {
        const int c = e / n, i = e % n;
        double xs = 0.0, xdv = 0.0;
        if (c < nce) {
            const double xn = zn[x_off + (c0 + c) * n + i], xc = zk[x_off + (c0 + c) * n + i];
            xs = xn + xc;
            xdv = xn - xc;
        }
        Sm[i + LD * c] = -xs;
        Dm[i + LD * c] = xdv;
    }
    __syncthreads();
    auto Y = [&](int j) { return (j & 1) ? Sm : Dm; };

    // ---- residual: Horner in G ---------------------------------------------------------------------------------
    for (int e = tid; e < nc * n; e += nth) {
        const int idx = (e % n) + LD * (e / n);
        W[q * LDc + idx] = p.pc[q] * Y(q)[idx];
    }
    __syncthreads();
    for (int j = q - 1; j >= 0; --j) {
        gemm_lds<true, false>(G, LD, W + (j + 1) * LDc, LD, W + j * LDc, LD, n, nc, n);
        __syncthreads();
        for (int e = tid; e < nc * n; e += nth) {
            const int idx = (e % n) + LD * (e / n);
            W[j * LDc + idx] = p.pc[j] * Y(j)[idx] + h * W[j * LDc + idx];
        }
        __syncthreads();
    }
    const long long bk = (long long)b * p.K + k;
    if (p.delta)
        for (int e = tid; e < nce * n; e += nth) p.delta[bk * xd + (long long)c0 * n + e] = W[(e % n) + LD * (e / n)];
    if (!JAC) return;

    double *jb = p.jac + bk * p.jac_per;
    const long long blk = p.compact ? (long long)n * n : (long long)d * n * n;
    double *jt = jb + 2 * blk;
    // ---- d/dh ------------------------------------------------------------------------------------------------
    {
        double *cur = V, *oth = V + LDc;
        for (int e = tid; e < nc * n; e += nth) {
            const int idx = (e % n) + LD * (e / n);
            cur[idx] = q * p.pc[q] * Y(q)[idx];
        }
        __syncthreads();
        for (int j = q - 1; j >= 1; --j) {
            gemm_lds<true, false>(G, LD, cur, LD, oth, LD, n, nc, n);
            __syncthreads();
            for (int e = tid; e < nc * n; e += nth) {
                const int idx = (e % n) + LD * (e / n);
                oth[idx] = j * p.pc[j] * Y(j)[idx] + h * oth[idx];
            }
            __syncthreads();
            double *t = cur;
            cur = oth;
            oth = t;
        }
        gemm_lds<true, false>(G, LD, cur, LD, oth, LD, n, nc, n);
        __syncthreads();
        for (int e = tid; e < nce * n; e += nth) {
            const int c = e / n, i = e % n;
            jt[((long long)(c0 + c) * (m + 1) + m) * n + i] = oth[i + LD * c];
        }
    }
    // ---- d/du_l ----------------------------------------------------------------------------------------------
    if (m > 0) {
        double *cur = dWa, *oth = dWb;
        for (int e = tid; e < m * nc * n; e += nth) {
            const int i = e % n, c = (e / n) % nc, l = e / (n * nc);
            const int *rp = p.csr_ptr + l * (n + 1);
            double a = 0.0;
            for (int t = rp[i]; t < rp[i + 1]; ++t) a += p.csr_val[t] * W[q * LDc + p.csr_col[t] + LD * c];
            cur[l * LDc + i + LD * c] = h * a;
        }
        __syncthreads();
        for (int j = q - 2; j >= 0; --j) {
            gemm_lds<true, false>(G, LD, cur, LD, oth, LD, n, m * nc, n);
            __syncthreads();
            for (int e = tid; e < m * nc * n; e += nth) {
                const int i = e % n, c = (e / n) % nc, l = e / (n * nc);
                const int *rp = p.csr_ptr + l * (n + 1);
                double a = 0.0;
                for (int t = rp[i]; t < rp[i + 1]; ++t) a += p.csr_val[t] * W[(j + 1) * LDc + p.csr_col[t] + LD * c];
                oth[l * LDc + i + LD * c] = h * (oth[l * LDc + i + LD * c] + a);
            }
            __syncthreads();
            double *t = cur;
            cur = oth;
            oth = t;
        }
        for (int e = tid; e < m * nce * n; e += nth) {
            const int i = e % n, c = (e / n) % nce, l = e / (n * nce);
            jt[((long long)(c0 + c) * (m + 1) + l) * n + i] = cur[l * LDc + i + LD * c];
        }
    }
    // ---- blocks: B^{+-} = sum_j c_j (+-h)^j G^j ------------------------------------------------------------------
    // each thread owns the flat column-major positions (2q', 2q'+1), q' = tid + blockDim.x r; for an odd n (compact density
    // vectors) the two positions may lie in different columns and the stores are scalar (blocks are not 16-byte aligned)
    if (!(p.compact && s != 0)) {  // compact layout: only slice 0 writes (and therefore forms) the blocks
        const int nn_ = n * n, half = (nn_ + 1) >> 1;
        const bool even = !(n & 1);
        double bp[PCL_NSP][2], bm[PCL_NSP][2];
        int o0_[PCL_NSP], o1_[PCL_NSP];  // LDS offsets of the two positions (-1: none)
#pragma unroll
        for (int r = 0; r < PCL_NSP; ++r) {
            const int pos = 2 * (tid + nth * r);
            o0_[r] = o1_[r] = -1;
            bp[r][0] = bm[r][0] = bp[r][1] = bm[r][1] = 0.0;
            if (pos < nn_) {
                const int i = pos % n, jj = pos / n;
                o0_[r] = i + LD * jj;
                bp[r][0] = bm[r][0] = (i == jj) ? 1.0 : 0.0;
            }
            if (pos + 1 < nn_) {
                const int i = (pos + 1) % n, jj = (pos + 1) / n;
                o1_[r] = i + LD * jj;
                bp[r][1] = bm[r][1] = (i == jj) ? 1.0 : 0.0;
            }
        }
        const double *Pc = G;
        double hp = 1.0, hm = 1.0;
        for (int j = 1; j <= q; ++j) {
            hp *= h;
            hm *= -h;
#pragma unroll
            for (int r = 0; r < PCL_NSP; ++r) {
                const double v0 = o0_[r] >= 0 ? Pc[o0_[r]] : 0.0, v1 = o1_[r] >= 0 ? Pc[o1_[r]] : 0.0;
                bp[r][0] += p.pc[j] * hp * v0;
                bp[r][1] += p.pc[j] * hp * v1;
                bm[r][0] += p.pc[j] * hm * v0;
                bm[r][1] += p.pc[j] * hm * v1;
            }
            if (j < q) {
                double *Pn = (Pc == Pa) ? Pb : Pa;
                gemm_lds<true, false>(G, LD, Pc, LD, Pn, LD, n, n, n);
                __syncthreads();
                Pc = Pn;
            }
        }
        int cbeg = c0, cend = c0 + nce;
        if (p.compact) {
            cbeg = 0;
            cend = (s == 0) ? 1 : 0;
        }
#pragma unroll
        for (int r = 0; r < PCL_NSP; ++r) {
            const int qq = tid + nth * r;
            if (qq < half)
                for (int c = cbeg; c < cend; ++c) {
                    double *o0 = jb + (long long)c * nn_ + 2 * qq;
                    if (even) {
                        store2(o0, -bp[r][0], -bp[r][1], p.nt);
                        store2(o0 + blk, bm[r][0], bm[r][1], p.nt);
                    } else {
                        o0[0] = -bp[r][0];
                        o0[blk] = bm[r][0];
                        if (o1_[r] >= 0) {
                            o0[1] = -bp[r][1];
                            o0[blk + 1] = bm[r][1];
                        }
                    }
                }
        }
    }
}
