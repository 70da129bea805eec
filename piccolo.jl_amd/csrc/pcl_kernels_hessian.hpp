// pcl_kernels_hessian.hpp -- Hessian-of-the-Lagrangian kernels.
#pragma once

// ------------------------------------------------------------------------------------------
// Hessian-of-Lagrangian kernel: one workgroup per (b, k); the d state columns are processed in
// chunks of nc columns (columns are independent except for the (m+1)(m+2)/2 scalar entries,
// whose per-chunk partial sums are accumulated in LDS in a fixed order -> deterministic).
// With M = mu_k (n x d):  A1 = G^T M, A2 = G^T A1, P_l = G_l^T M, Q_l = G^T P_l, R_l = G_l^T A1,
//                         GD = G D, E_l = G_l D.
// LDS map: G [LD*n] | Mm | S | D | GD | A1 | A2 (each LD*nc) | P | Q | E (each m*LD*nc) | us | red | acc
// ------------------------------------------------------------------------------------------

template <bool MFMA>
__global__ __launch_bounds__(256) void pcl_hess_kernel(const KParams p) {
    extern __shared__ double lds[];
    const int n = p.n, d = p.cols, m = p.m, LD = p.LD, nc = p.nc;  // d: state columns here
    const int tid = threadIdx.x, nth = blockDim.x;
    const int k = blockIdx.x % p.K, b = blockIdx.x / p.K;
    const long long xd = (long long)n * d;
    const int LDc = LD * nc;
    const int nscal = (m + 1) * (m + 2) / 2;
    const int nw = nth >> 6, wv = tid >> 6, lane = tid & 63;

    double *G = lds;
    double *Mm = G + LD * n;
    double *Sm = Mm + LDc;
    double *Dm = Sm + LDc;
    double *GD = Dm + LDc;
    double *A1 = GD + LDc;
    double *A2 = A1 + LDc;
    double *P = A2 + LDc;
    double *Q = P + m * LDc;
    double *E = Q + m * LDc;
    double *us = E + m * LDc;
    double *red = us + 8 + m;       // nw * nscal
    double *acc = red + nw * nscal;  // nscal

    const double *Zb = p.Z + (long long)b * p.z_batch_stride;
    const double *zk = Zb + (long long)k * p.z_dim;
    const double *zn = zk + p.z_dim;
    const int x_off = p.x_offs[p.z_batch_stride ? 0 : b];
    const double h = zk[p.dt_off];
    const double c1 = 0.5 * h, c2 = h * h * (1.0 / 12.0), h6 = h * (1.0 / 6.0);
    const long long bk = (long long)b * p.K + k;
    const double *mu = p.mu + bk * xd;
    double *H = p.hess + bk * p.hess_per;
    double *H3 = H + nscal, *H4 = H3 + (long long)m * xd, *H5 = H4 + xd, *H6 = H5 + (long long)m * xd;

    build_G(p, p.G0 + (long long)b * p.g0_batch_stride, zk, G, us);
    for (int e = tid; e < nscal; e += nth) acc[e] = 0.0;

    for (int c0 = 0; c0 < d; c0 += nc) {
        const int nce = min(nc, d - c0);
        __syncthreads();  // previous chunk fully consumed (and G / acc initialised)
        for (int e = tid; e < nce * n; e += nth) {
            const int c = e / n, i = e % n;
            const long long g = (long long)(c0 + c) * n + i;
            const double xn = zn[x_off + g], xc = zk[x_off + g];
            Sm[i + LD * c] = xn + xc;
            Dm[i + LD * c] = xn - xc;
            Mm[i + LD * c] = mu[g];
        }
        __syncthreads();
        // sparse applications: P_l = G_l^T M (CSC columns of G_l), E_l = G_l D (CSR rows)
        for (int e = tid; e < m * nce * n; e += nth) {
            const int i = e % n, c = (e / n) % nce, l = e / (n * nce);
            const int *cp = p.csc_ptr + l * (n + 1);
            double a = 0.0;
            for (int q = cp[i]; q < cp[i + 1]; ++q) a += p.csc_val[q] * Mm[p.csc_row[q] + LD * c];
            P[l * LDc + i + LD * c] = a;
            const int *rp = p.csr_ptr + l * (n + 1);
            double a2 = 0.0;
            for (int q = rp[i]; q < rp[i + 1]; ++q) a2 += p.csr_val[q] * Dm[p.csr_col[q] + LD * c];
            E[l * LDc + i + LD * c] = a2;
        }
        __syncthreads();
        gemm_lds<MFMA, false>(G, LD, Dm, LD, GD, LD, n, nce, n);
        gemm_lds<MFMA, true>(G, LD, Mm, LD, A1, LD, n, nce, n);
        for (int l = 0; l < m; ++l) gemm_lds<MFMA, true>(G, LD, P + l * LDc, LD, Q + l * LDc, LD, n, nce, n);
        __syncthreads();
        gemm_lds<MFMA, true>(G, LD, A1, LD, A2, LD, n, nce, n);

        // ---- scalar segments 0..2: per-wave partial sums -> red[w][pidx] ----------------------
        int pidx = 0;
        for (int i = 0; i < m; ++i)
            for (int j = 0; j <= i; ++j, ++pidx) {
                double v = 0.0;
                for (int e = tid; e < nce * n; e += nth) {
                    const int idx = (e % n) + LD * (e / n);
                    v += P[i * LDc + idx] * E[j * LDc + idx] + P[j * LDc + idx] * E[i * LDc + idx];
                }
                v = wave_sum(v);
                if (lane == 0) red[wv * nscal + pidx] = c2 * v;
            }
        for (int j = 0; j < m; ++j, ++pidx) {
            double v1 = 0.0, v2 = 0.0;
            for (int e = tid; e < nce * n; e += nth) {
                const int idx = (e % n) + LD * (e / n);
                v1 += P[j * LDc + idx] * Sm[idx];
                v2 += P[j * LDc + idx] * GD[idx] + A1[idx] * E[j * LDc + idx];
            }
            v1 = wave_sum(v1);
            v2 = wave_sum(v2);
            if (lane == 0) red[wv * nscal + pidx] = -0.5 * v1 + h6 * v2;
        }
        {
            double v = 0.0;
            for (int e = tid; e < nce * n; e += nth) {
                const int idx = (e % n) + LD * (e / n);
                v += A1[idx] * GD[idx];
            }
            v = wave_sum(v);
            if (lane == 0) red[wv * nscal + pidx] = v * (1.0 / 6.0);
        }
        __syncthreads();  // red complete, A2 complete
        for (int e = tid; e < nscal; e += nth) {
            double t = acc[e];
            for (int w = 0; w < nw; ++w) t += red[w * nscal + e];
            acc[e] = t;
        }
        // ---- vector segments 3..6 for this chunk's columns --------------------------------------
        for (int e = tid; e < m * nce * n; e += nth) {
            const int i = e % n, c = (e / n) % nce, l = e / (n * nce);
            const int *cp = p.csc_ptr + l * (n + 1);
            double r = 0.0;  // R_l = G_l^T (G^T M)
            for (int q = cp[i]; q < cp[i + 1]; ++q) r += p.csc_val[q] * A1[p.csc_row[q] + LD * c];
            const int idx = i + LD * c;
            const double kt = c2 * (Q[l * LDc + idx] + r);
            const double pl = -c1 * P[l * LDc + idx];
            const long long o = (long long)l * xd + (long long)(c0 + c) * n + i;
            H3[o] = pl - kt;
            H5[o] = pl + kt;
        }
        for (int e = tid; e < nce * n; e += nth) {
            const int idx = (e % n) + LD * (e / n);
            const long long o = (long long)c0 * n + e;
            H4[o] = -0.5 * A1[idx] - h6 * A2[idx];
            H6[o] = -0.5 * A1[idx] + h6 * A2[idx];
        }
    }
    __syncthreads();
    for (int e = tid; e < nscal; e += nth) H[e] = acc[e];
}

// ------------------------------------------------------------------------------------------
// Hessian of the Lagrangian at ANY diagonal Pade order p = 2q <= 10 (correctness first: one workgroup per (b, k), column
// chunks, every product through gemm_lds).  With T_j = c_j h^j, T'_j = j c_j h^(j-1), T''_j = j (j-1) c_j h^(j-2),
// Y_j = (-1)^j X_{k+1} - X_k, M = mu_k:
//     W_j = G^T W_{j-1} (W_0 = M),   V_{l,j} = G^T V_{l,j-1} + G_l^T W_{j-1},   U_{il,j} = G^T U_{il,j-1} + G_l^T V_{i,j-1} + G_i^T V_{l,j-1}
//     (u_i,u_l): sum_j T_j <U_{il,j},Y_j>   (h,u_l): sum_j T'_j <V_{l,j},Y_j>   (h,h): sum_j T''_j <W_j,Y_j>
//     d2/du_l dX_{k+1} = sum_j T_j (-1)^j V_{l,j},  d2/du_l dX_k = -sum_j T_j V_{l,j},  d2/dh dX_{k+1} = sum_j T'_j (-1)^j W_j,  d2/dh dX_k = -sum_j T'_j W_j
// Levels j = 1..q are walked with rolling buffers (previous / current level); the scalar entries are accumulated per wave and
// level (wave sums, one row of `red` per wave) and added over the waves in order at the end: deterministic.
// LDS map (doubles, LDc = LD*nc): G | Xn | Xc | Wa | Wb | H4a | H6a | Va[m] | Vb[m] | H3a[m] | H5a[m] | Ua[np] | Ub[np] | us | red[nw][nscal]
// ------------------------------------------------------------------------------------------
template <bool MFMA>
__global__ __launch_bounds__(256) void pcl_hess_general_kernel(const KParams p) {
    extern __shared__ double lds[];
    const int n = p.n, d = p.cols, m = p.m, LD = p.LD, nc = p.nc, q = p.q;
    const int tid = threadIdx.x, nth = blockDim.x;
    const int k = blockIdx.x % p.K, b = blockIdx.x / p.K;
    const long long xd = (long long)n * d;
    const int LDc = LD * nc;
    const int npair = m * (m + 1) / 2, nscal = (m + 1) * (m + 2) / 2;
    const int nw = nth >> 6, wv = tid >> 6, lane = tid & 63;

    double *G = lds;
    double *Xn = G + LD * n, *Xc = Xn + LDc, *Wa = Xc + LDc, *Wb = Wa + LDc, *H4a = Wb + LDc, *H6a = H4a + LDc;
    double *Va = H6a + LDc, *Vb = Va + m * LDc, *H3a = Vb + m * LDc, *H5a = H3a + m * LDc;
    double *Ua = H5a + m * LDc, *Ub = Ua + npair * LDc;
    double *us = Ub + npair * LDc;
    double *red = us + 8 + m;  // [nw][nscal]

    const double *Zb = p.Z + (long long)b * p.z_batch_stride;
    const double *zk = Zb + (long long)k * p.z_dim;
    const double *zn = zk + p.z_dim;
    const int x_off = p.x_offs[p.z_batch_stride ? 0 : b];
    const double h = zk[p.dt_off];
    const long long bk = (long long)b * p.K + k;
    const double *mu = p.mu + bk * xd;
    double *H = p.hess + bk * p.hess_per;
    double *H3 = H + nscal, *H4 = H3 + (long long)m * xd, *H5 = H4 + xd, *H6 = H5 + (long long)m * xd;

    build_G(p, p.G0 + (long long)b * p.g0_batch_stride, zk, G, us);
    for (int e = tid; e < nw * nscal; e += nth) red[e] = 0.0;

    for (int c0 = 0; c0 < d; c0 += nc) {
        const int nce = min(nc, d - c0);
        const int ne = nce * n;  // elements of a chunk array: e -> (row e % n, column e / n)
        __syncthreads();  // previous chunk fully consumed (and G / red initialised)
        for (int e = tid; e < ne; e += nth) {
            const int c = e / n, i = e % n, idx = i + LD * c;
            const long long g = (long long)(c0 + c) * n + i;
            Xn[idx] = zn[x_off + g];
            Xc[idx] = zk[x_off + g];
            Wa[idx] = mu[g];  // W_0 = M
            H4a[idx] = H6a[idx] = 0.0;
        }
        for (int e = tid; e < m * ne; e += nth) {
            const int a = e / ne, r = e - a * ne, idx = a * LDc + (r % n) + LD * (r / n);
            Va[idx] = 0.0;  // V_{l,0} = 0
            H3a[idx] = H5a[idx] = 0.0;
        }
        for (int e = tid; e < npair * ne; e += nth) {
            const int a = e / ne, r = e - a * ne;
            Ua[a * LDc + (r % n) + LD * (r / n)] = 0.0;  // U_{il,0} = 0
        }
        double *Wp = Wa, *Wc = Wb, *Vp = Va, *Vc = Vb, *Up = Ua, *Uc = Ub;
        double hj1 = 1.0, hj2 = 0.0;  // h^(j-1), h^(j-2)
        for (int j = 1; j <= q; ++j) {
            __syncthreads();
            // dense parts of level j: G^T times level j-1 (arrays of one family are consecutive column blocks)
            gemm_lds<MFMA, true>(G, LD, Wp, LD, Wc, LD, n, nc, n);
            if (m > 0) gemm_lds<MFMA, true>(G, LD, Vp, LD, Vc, LD, n, m * nc, n);
            if (npair > 0 && j >= 2) gemm_lds<MFMA, true>(G, LD, Up, LD, Uc, LD, n, npair * nc, n);
            __syncthreads();
            // sparse parts (CSC columns of G_l = rows of G_l^T): V_{l,j} += G_l^T W_{j-1};  U_{il,j} (+)= G_l^T V_{i,j-1} + G_i^T V_{l,j-1}
            for (int e = tid; e < m * ne; e += nth) {
                const int l = e / ne, r = e - l * ne, i = r % n, c = r / n;
                const int *cp = p.csc_ptr + l * (n + 1);
                double a = 0.0;
                for (int t = cp[i]; t < cp[i + 1]; ++t) a += p.csc_val[t] * Wp[p.csc_row[t] + LD * c];
                Vc[l * LDc + i + LD * c] += a;
            }
            for (int e = tid; e < npair * ne; e += nth) {
                const int pr = e / ne, r = e - pr * ne, i = r % n, c = r / n;
                int pi = 0;
                while ((pi + 1) * (pi + 2) / 2 <= pr) ++pi;  // pair index pr = pi (pi + 1) / 2 + pl, pl <= pi
                const int pl = pr - pi * (pi + 1) / 2;
                const int *ci = p.csc_ptr + pi * (n + 1), *cl = p.csc_ptr + pl * (n + 1);
                double a = 0.0;
                for (int t = cl[i]; t < cl[i + 1]; ++t) a += p.csc_val[t] * Vp[pi * LDc + p.csc_row[t] + LD * c];
                for (int t = ci[i]; t < ci[i + 1]; ++t) a += p.csc_val[t] * Vp[pl * LDc + p.csc_row[t] + LD * c];
                double *u = Uc + pr * LDc + i + LD * c;
                *u = (j >= 2 ? *u : 0.0) + a;
            }
            __syncthreads();
            // contributions of level j
            const double cj = p.pc[j], Tj = cj * hj1 * h, T1 = j * cj * hj1, T2 = j >= 2 ? j * (j - 1) * cj * hj2 : 0.0;
            const double sg = (j & 1) ? -1.0 : 1.0;
            for (int e = tid; e < ne; e += nth) {
                const int idx = (e % n) + LD * (e / n);
                H4a[idx] += T1 * Wc[idx];
                H6a[idx] += T1 * sg * Wc[idx];
            }
            for (int e = tid; e < m * ne; e += nth) {
                const int l = e / ne, r = e - l * ne, idx = l * LDc + (r % n) + LD * (r / n);
                H3a[idx] += Tj * Vc[idx];
                H5a[idx] += Tj * sg * Vc[idx];
            }
            // scalars: entry -> sum over the chunk's elements of (family array) * Y_j, Y_j = sg Xn - Xc
            for (int en = 0; en < nscal; ++en) {
                const double *A;
                double coef;
                if (en < npair) {
                    A = Uc + en * LDc;
                    coef = j >= 2 ? Tj : 0.0;
                } else if (en < npair + m) {
                    A = Vc + (en - npair) * LDc;
                    coef = T1;
                } else {
                    A = Wc;
                    coef = T2;
                }
                double v = 0.0;
                if (coef != 0.0)
                    for (int e = tid; e < ne; e += nth) {
                        const int idx = (e % n) + LD * (e / n);
                        v += A[idx] * (sg * Xn[idx] - Xc[idx]);
                    }
                v = wave_sum(v);
                if (lane == 0) red[wv * nscal + en] += coef * v;
            }
            hj2 = hj1;
            hj1 *= h;
            double *t;
            t = Wp, Wp = Wc, Wc = t;
            t = Vp, Vp = Vc, Vc = t;
            t = Up, Up = Uc, Uc = t;
        }
        __syncthreads();
        for (int e = tid; e < m * ne; e += nth) {
            const int l = e / ne, r = e - l * ne, idx = l * LDc + (r % n) + LD * (r / n);
            const long long o = (long long)l * xd + (long long)c0 * n + r;
            H3[o] = -H3a[idx];
            H5[o] = H5a[idx];
        }
        for (int e = tid; e < ne; e += nth) {
            const int idx = (e % n) + LD * (e / n);
            const long long o = (long long)c0 * n + e;
            H4[o] = -H4a[idx];
            H6[o] = H6a[idx];
        }
    }
    __syncthreads();
    for (int e = tid; e < nscal; e += nth) {
        double t = 0.0;
        for (int w = 0; w < nw; ++w) t += red[w * nscal + e];
        H[e] = t;
    }
}

// ------------------------------------------------------------------------------------------
// Hessian-of-Lagrangian kernel, version 2 (default when every drive row / column has <= EW entries and m <= 6):
// persistent workgroups (2 per CU, 4 wavefronts each) over work items (b, k, slice of <= 16 state columns).
// Per item the four waves work wave-synchronously on chunks of NCW = 16/(m+1) columns:
//     operand tile  [M | P_1 .. P_m],  P_l = G_l^T M  (ELL rows of G_l^T in registers, lane = row)
//     one pass of the f64 matrix cores:  G^T [M | P_l] = [A1 | Q_l]
//     R_l = G_l^T A1, E_l = G_l D (registers)  ->  the d2/du dX vectors straight to HBM
//     the (m+1)(m+2)/2 - 1 scalar entries that involve u as per-lane partial sums in registers:
//         <M,(G_i G_j + G_j G_i) D> = <P_i,E_j> + <P_j,E_i>,   <M,G_j S> = <P_j,S>,
//         <M,(G_j G + G G_j) D> = <Q_j,D> + <A1,E_j>
// then, once per item: A2 = G^T A1 for all the slice's columns in ONE matrix-core pass (wave w = row tile w),
// the d2/dh dX vectors, <A2,D>, and a fixed-order reduction lane -> wave -> workgroup -> (slices of the interval,
// summed by the last slice to arrive: partial sums in `hpart`, acquire-release arrival counter in `hcnt`) -> deterministic.
// No G^2, no G D product: every contraction with D is moved onto M's side.
// LDS map (doubles): G [LD*n] | A1s [LD*16] | A2s [LD*16] | Ds [LD*16] | per wave Mw [LD*16] | wsum [4][NSC] | wsum2 [4] | flag
// ------------------------------------------------------------------------------------------

// TD: compile-time Hilbert dimension (0 = run-time).  ANTI: every G_l is exactly antisymmetric (G = iso(-iH) with H
// Hermitian), so the rows of G_l^T are minus the rows of G_l and one ELL table serves both.
template <int EW, int TM, int TD, bool ANTI>
__global__ __launch_bounds__(256, 2) void pcl_hess_kernel_v2(const KParams p) {
    extern __shared__ double lds[];
    constexpr int m = TM;
    constexpr int NCW = 16 / (TM + 1);
    constexpr int NSC = (TM + 1) * (TM + 2) / 2;
    constexpr int NPAIR = TM * (TM + 1) / 2;
    constexpr int NACC = NPAIR + TM;
    const int n = TD ? 2 * TD : p.n, d = p.cols, LD = TD ? ((2 * TD + 3) & ~3) + 2 : p.LD;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int li = lane & 15, lk = lane >> 4;
    const int nn = n * n;
    const long long xd = (long long)n * d;
    const int kfull = n >> 2, krem = n & 3;

    double *G = lds;
    double *A1s = G + LD * n;
    double *A2s = A1s + LD * 16;
    double *Ds = A2s + LD * 16;
    double *Mw = Ds + LD * 16 + wave * (LD * 16);
    double *wsum = Ds + LD * 16 + 4 * (LD * 16);
    double *wsum2 = wsum + 4 * NSC;
    int *lastflag = reinterpret_cast<int *>(wsum2 + 4);

    // ELL rows of G_l (er) and of G_l^T (et) for row = lane
    constexpr int TE = ANTI ? 1 : TM;  // the transposed table is only held when it differs from minus the plain one
    unsigned short er_c[TM][EW], et_c_[TE][EW];
    double er_v[TM][EW], et_v_[TE][EW];
#pragma unroll
    for (int l = 0; l < TM; ++l)
#pragma unroll
        for (int q = 0; q < EW; ++q) {
            er_c[l][q] = 0;
            er_v[l][q] = 0.0;
            if (!ANTI) {
                et_c_[ANTI ? 0 : l][q] = 0;
                et_v_[ANTI ? 0 : l][q] = 0.0;
            }
            if (lane < n) {
                if (q < p.ell_w) {
                    er_c[l][q] = (unsigned short)p.ell_col[(l * n + lane) * p.ell_w + q];
                    er_v[l][q] = p.ell_val[(l * n + lane) * p.ell_w + q];
                }
                if (!ANTI && q < p.ellt_w) {
                    et_c_[ANTI ? 0 : l][q] = (unsigned short)p.ellt_col[(l * n + lane) * p.ellt_w + q];
                    et_v_[ANTI ? 0 : l][q] = p.ellt_val[(l * n + lane) * p.ellt_w + q];
                }
            }
        }
#define ET_C(l, q) (ANTI ? er_c[l][q] : et_c_[ANTI ? 0 : (l)][q])
#define ET_V(l, q) (ANTI ? er_v[l][q] : et_v_[ANTI ? 0 : (l)][q])  // ANTI: the caller negates the sum
    if (!p.g0_batch_stride)
        load_tile<256>(p.G0, G, n, LD, tid);

    const int S = p.S, nc = p.nc;
    const int n_items = p.batch * p.K * S;
#ifdef PCL_PROFILE
    int stamp = 0;
#define PCL_HSTAMP()                                                                                  \
    do {                                                                                             \
        if (p.dbg && blockIdx.x == 0 && tid == 0 && stamp < 60) p.dbg[stamp++] = (long long)__builtin_amdgcn_s_memtime(); \
    } while (0)
#else
#define PCL_HSTAMP() do { } while (0)
#endif
    PCL_HSTAMP();  // prologue done
    for (int item = blockIdx.x; item < n_items; item += gridDim.x) {
        const int s = item % S, k = (item / S) % p.K, b = item / (S * p.K);
        const int c0 = s * nc, nce = min(nc, d - c0);
        const double *zk = p.Z + (long long)b * p.z_batch_stride + (long long)k * p.z_dim;
        const double *zn = zk + p.z_dim;
        const int x_off = p.x_offs[p.z_batch_stride ? 0 : b];
        const double h = zk[p.dt_off];
        const double c1 = 0.5 * h, c2 = h * h * (1.0 / 12.0), h6 = h * (1.0 / 6.0);
        const long long bk = (long long)b * p.K + k;
        const double *mu = p.mu + bk * xd;
        double *H = p.hess + bk * p.hess_per;
        double *H3 = H + NSC, *H4 = H3 + (long long)m * xd, *H5 = H4 + xd, *H6 = H5 + (long long)m * xd;

        // inputs of this wave's first chunk: requested before G is built, consumed after
        const int nchunk = (nce + NCW - 1) / NCW;
        double pxn[NCW], pxc[NCW], pmu[NCW];
#pragma unroll
        for (int c = 0; c < NCW; ++c) {
            pxn[c] = pxc[c] = pmu[c] = 0.0;
            if (lane < n && wave < nchunk && wave * NCW + c < nce) {
                const long long g = (long long)(c0 + wave * NCW + c) * n + lane;
                pxn[c] = zn[x_off + g];
                pxc[c] = zk[x_off + g];
                pmu[c] = mu[g];
            }
        }
        __syncthreads();  // the previous item of this workgroup is fully consumed
        PCL_HSTAMP();  // item start
        // ---- G(u_k): drift everywhere (per-member drift only), then the drives' union pattern ------------------
        {
            const double *G0b = p.G0 + (long long)b * p.g0_batch_stride;
            if (p.g0_batch_stride)
                for (int e = tid; e < nn; e += 256)
                    if (p.umap[e] < 0) G[(e % n) + LD * (e / n)] = G0b[e];
            double uu[TM];
#pragma unroll
            for (int l = 0; l < TM; ++l) uu[l] = zk[p.u_off + l];
            for (int q = tid; q < p.n_upos; q += 256) {
                const int pos = p.upos[q];
                double g = p.g0_batch_stride ? G0b[pos] : p.ug0[q];
                const double *cf = p.ucoef + (long long)q * m;
#pragma unroll
                for (int l = 0; l < TM; ++l) g += uu[l] * cf[l];
                G[(pos % n) + LD * (pos / n)] = g;
            }
        }
        __syncthreads();

        PCL_HSTAMP();  // G built
        double acc[NACC];
#pragma unroll
        for (int e = 0; e < NACC; ++e) acc[e] = 0.0;
        for (int ch = wave; ch < nchunk; ch += 4) {
            const int cl0 = ch * NCW;                // first column of the chunk inside the slice
            const int ncc = min(NCW, nce - cl0);     // columns in this chunk
            double Sv[NCW], Pv[TM][NCW];
            if (lane < n) {
#pragma unroll
                for (int c = 0; c < NCW; ++c) {
                    Sv[c] = 0.0;
                    double mv = 0.0;
                    if (c < ncc) {
                        double xn = pxn[c], xc = pxc[c];
                        mv = pmu[c];
                        if (ch != wave) {  // further chunks of a wide slice load at use
                            const long long g = (long long)(c0 + cl0 + c) * n + lane;
                            xn = zn[x_off + g];
                            xc = zk[x_off + g];
                            mv = mu[g];
                        }
                        Sv[c] = xn + xc;
                        Ds[lane + LD * (cl0 + c)] = xn - xc;
                    }
                    Mw[lane + LD * c] = mv;
                }
            }
            wave_lds_sync();
            PCL_HSTAMP();  // inputs loaded
            if (lane < n) {
#pragma unroll
                for (int l = 0; l < TM; ++l)
#pragma unroll
                    for (int c = 0; c < NCW; ++c) {
                        double pv = 0.0;
                        if (c < ncc) {
#pragma unroll
                            for (int q = 0; q < EW; ++q) pv += ET_V(l, q) * Mw[ET_C(l, q) + LD * c];
                            if (ANTI) pv = -pv;
                        }
                        Pv[l][c] = pv;
                        Mw[lane + LD * (NCW + l * NCW + c)] = pv;
                    }
            }
            wave_lds_sync();
            PCL_HSTAMP();  // P, E done
            // ---- [A1 | Q_l] = G^T [M | P_l]: all row tiles at once (they share the b operand) --------------------
            double4_t ac[PCL_MAXRT];
            {
                const double *Bp = Mw + lk + LD * li;
                const double *Ap[PCL_MAXRT];
                bool rok[PCL_MAXRT];
#pragma unroll
                for (int t = 0; t < PCL_MAXRT; ++t) {
                    rok[t] = t * 16 < n;
                    Ap[t] = G + lk + LD * ((rok[t] ? t * 16 : 0) + li);
                    ac[t] = double4_t{0.0, 0.0, 0.0, 0.0};
                }
                double an[PCL_MAXRT], bn = 0.0;
#pragma unroll
                for (int t = 0; t < PCL_MAXRT; ++t) an[t] = kfull > 0 ? Ap[t][0] : 0.0;
                if (kfull > 0) bn = Bp[0];
                for (int ks = 0; ks < kfull; ++ks) {
                    double a[PCL_MAXRT];
                    const double bb = bn;
#pragma unroll
                    for (int t = 0; t < PCL_MAXRT; ++t) a[t] = an[t];
                    if (ks + 1 < kfull) {
#pragma unroll
                        for (int t = 0; t < PCL_MAXRT; ++t) an[t] = Ap[t][4 * (ks + 1)];
                        bn = Bp[4 * (ks + 1)];
                    }
#pragma unroll
                    for (int t = 0; t < PCL_MAXRT; ++t)
                        if (rok[t]) ac[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[t], bb, ac[t], 0, 0, 0);
                }
                if (krem) {
                    const bool ok = lk < krem;
                    const double bb = ok ? Bp[4 * kfull] : 0.0;
#pragma unroll
                    for (int t = 0; t < PCL_MAXRT; ++t)
                        if (rok[t]) {
                            const double a = ok ? Ap[t][4 * kfull] : 0.0;
                            ac[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bb, ac[t], 0, 0, 0);
                        }
                }
            }
            wave_lds_sync();  // every operand read of this wave is complete before the tile is overwritten
            PCL_HSTAMP();  // MFMA done
            if (li < (TM + 1) * NCW) {
                double *a1 = (li < ncc) ? A1s + LD * (cl0 + li) : nullptr;
#pragma unroll
                for (int t = 0; t < PCL_MAXRT; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int row = t * 16 + lk + 4 * r;
                        if (row < n) {
                            Mw[row + LD * li] = ac[t][r];
                            if (a1) a1[row] = ac[t][r];
                        }
                    }
            }
            wave_lds_sync();
            if (lane < n) {
#pragma unroll
                for (int c = 0; c < NCW; ++c)
                    if (c < ncc) {
                        const double a1v = Mw[lane + LD * c];
                        const double *Dc = Ds + LD * (cl0 + c);
                        const double dv = Dc[lane];
                        const long long o = (long long)(c0 + cl0 + c) * n + lane;
                        double Ev[TM];  // E_l = G_l D, this row and column
#pragma unroll
                        for (int l = 0; l < TM; ++l) {
                            double ev = 0.0, r = 0.0;  // R_l = G_l^T A1
#pragma unroll
                            for (int q = 0; q < EW; ++q) {
                                ev += er_v[l][q] * Dc[er_c[l][q]];
                                r += ET_V(l, q) * Mw[ET_C(l, q) + LD * c];
                            }
                            if (ANTI) r = -r;
                            Ev[l] = ev;
                            const double qv = Mw[lane + LD * (NCW + l * NCW + c)];
                            const double kt = c2 * (qv + r), pl = -c1 * Pv[l][c];
                            H3[(long long)l * xd + o] = pl - kt;
                            H5[(long long)l * xd + o] = pl + kt;
                            acc[NPAIR + l] += -0.5 * Pv[l][c] * Sv[c] + h6 * (qv * dv + a1v * ev);
                        }
                        int e = 0;
#pragma unroll
                        for (int i = 0; i < TM; ++i)
#pragma unroll
                            for (int j = 0; j <= i; ++j, ++e) acc[e] += Pv[i][c] * Ev[j] + Pv[j][c] * Ev[i];
                    }
            }
            wave_lds_sync();  // Mw is rewritten by this wave's next chunk / the reduction below
            PCL_HSTAMP();  // chunk outputs + sums done
        }
        // ---- lane -> wave reduction of the per-lane partial sums on the matrix cores (fixed order) -----------------
        // C += 1_e x v_e : with a = [li == e] and b = the lanes' partial sums of entry e, row e of C collects
        // sum_k v_e[lane j + 16 k] in column j; four DPP steps then add the 16 columns of a row.
        {
            double4_t r0 = {0.0, 0.0, 0.0, 0.0}, r1 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int e = 0; e < NACC; ++e) {
                const double ind = (li == (e & 15)) ? 1.0 : 0.0;
                if (e < 16)
                    r0 = __builtin_amdgcn_mfma_f64_16x16x4f64(ind, acc[e], r0, 0, 0, 0);
                else
                    r1 = __builtin_amdgcn_mfma_f64_16x16x4f64(ind, acc[e], r1, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const double t0 = row16_sum(r0[r]);
                const int e0 = lk + 4 * r;
                if (li == 0 && e0 < NACC) wsum[wave * NSC + e0] = t0;
                if (NACC > 16) {
                    const double t1 = row16_sum(r1[r]);
                    if (li == 0 && 16 + e0 < NACC) wsum[wave * NSC + 16 + e0] = t1;
                }
            }
        }
        PCL_HSTAMP();  // wave reduction done
        __syncthreads();  // A1s, Ds and wsum complete
        PCL_HSTAMP();
        // ---- A2 = G^T A1 for the slice's columns: wave w = row tile w --------------------------------------------
        if (wave * 16 < n) {
            const double *Ap = G + lk + LD * (wave * 16 + li);
            const double *Bp = A1s + lk + LD * li;
            double4_t a2 = {0.0, 0.0, 0.0, 0.0};
            for (int ks = 0; ks < kfull; ++ks) a2 = __builtin_amdgcn_mfma_f64_16x16x4f64(Ap[4 * ks], Bp[4 * ks], a2, 0, 0, 0);
            if (krem) {
                const bool ok = lk < krem;
                a2 = __builtin_amdgcn_mfma_f64_16x16x4f64(ok ? Ap[4 * kfull] : 0.0, ok ? Bp[4 * kfull] : 0.0, a2, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = wave * 16 + lk + 4 * r;
                if (row < n) A2s[row + LD * li] = a2[r];
            }
        }
        __syncthreads();
        {
            double v = 0.0;
            for (int e = tid; e < nce * n; e += 256) {
                const int c = e / n, i = e - c * n;
                v += A2s[i + LD * c] * Ds[i + LD * c];
            }
            v = wave_sum(v);
            if (lane == 0) wsum2[wave] = v;
        }
        __syncthreads();
        PCL_HSTAMP();  // A2 done
        if (tid < NSC) {
            double tot;
            if (tid < NACC) {
                tot = ((wsum[tid] + wsum[NSC + tid]) + wsum[2 * NSC + tid]) + wsum[3 * NSC + tid];
                if (tid < NPAIR) tot *= c2;
            } else {
                tot = (((wsum2[0] + wsum2[1]) + wsum2[2]) + wsum2[3]) * (1.0 / 6.0);
            }
            if (S == 1)
                H[tid] = tot;
            else {
                // agent-scope (write-through) store of this slice's partial entry; drained before the barrier below, so it
                // has left the CU before the arrival counter moves
                __hip_atomic_store(p.hpart + (bk * S + s) * NSC + tid, tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
        }
        unsigned int ticket = 0;
        if (S > 1) {
            __syncthreads();
            // release (this slice's partials, ordered before by the barrier) / acquire (the other slices' partials, for the
            // last arriver) on the arrival counter: the pairing the HIP memory model asks for.  This kernel is the fallback
            // of kernel 3 now (shapes without an instance; S > 1 only for d > 16), so the price of the release -- a
            // write-back of this XCD's dirty L2 lines per item -- no longer sits on the benchmarked path.
            if (tid == 0) ticket = __hip_atomic_fetch_add(p.hcnt + bk, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        }
        // the d2/dh dX vectors go out while the counter's round trip is in flight
        for (int e = tid; e < nce * n; e += 256) {
            const int c = e / n, i = e - c * n;
            const double a1 = A1s[i + LD * c], a2 = A2s[i + LD * c];
            const long long o = (long long)(c0 + c) * n + i;
            H4[o] = -0.5 * a1 - h6 * a2;
            H6[o] = -0.5 * a1 + h6 * a2;
        }
        if (S > 1) {
            if (tid == 0) {
                const int last = ticket == (unsigned int)(S - 1);
                if (last) __hip_atomic_store(p.hcnt + bk, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // ready for the next launch
                *lastflag = last;
            }
            __syncthreads();
            if (*lastflag && tid < NSC) {  // the last slice of the interval to arrive sums the partials in slice order
                double t = 0.0;  // agent-scope loads bypass this XCD's L2
                for (int q = 0; q < S; ++q) t += __hip_atomic_load(p.hpart + (bk * S + q) * NSC + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                H[tid] = t;
            }
        }
        PCL_HSTAMP();  // item done
    }
}
