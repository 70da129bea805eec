// pcl_kernels_fused_v2.hpp -- persistent wave-specialised kernel with two workgroups per CU (version 2): the fallback of
// version 3 for shapes whose double-buffered tiles do not fit LDS, for kets and for the compact mode at such shapes.
#pragma once

// ------------------------------------------------------------------------------------------
// Fused residual + Jacobian kernel, version 2 (default): 8 wavefronts, wave-specialised.
//   waves 0-3 ("matrix" waves) own one 16-row tile each and run every MFMA product;
//   waves 4-7 ("stream" waves) apply the sparse drives (VALU) and then stream the slice's share of
//   the replicated B^{+-} blocks to HBM while the matrix waves compute the slice's columns.
// Phases (separated by one __syncthreads each):
//   0  all   : G(u_k) -> LDS (one pass: drift tile + union-pattern map), S, D, ELL drives -> LDS
//   1  matrix: G2 = G*G (first d columns + mirror if iso)      stream: M1[:, (2+l)nc..] = G_l D
//   2  matrix: W1 = G*M1, G2D = G2*D                            stream: -B^+, B^- block copies -> HBM
//   3  all   : delta, d/ddt, d/du_l columns -> HBM
// LDS map (doubles): G [LD*n] | G2 [LD*n] | M1 [LD*ncols1] | W1 [LD*ncols1] | G2D [LD*nc] | ELL val/col | slack
// MFMA operand loads are unconditional: a tile may read rows/columns past the matrix edge (the
// neighbouring buffer); such lanes only feed output rows/columns that are never stored.
// ------------------------------------------------------------------------------------------
template <int MODE>  // 0: plain store; 1: store + iso mirror (C = G2 first d columns)
__device__ __forceinline__ void wave_rowgemm(const double *__restrict__ A, int lda, const double *__restrict__ B,
                                             int ldb, double *__restrict__ C, int ldc, int M, int Nc, int Kd, int wave,
                                             int nwaves, int lane, int dmir) {
    const int rt_n = (M + 15) >> 4, ct_n = (Nc + 15) >> 4;
    const int kfull = Kd >> 2, krem = Kd & 3;
    const int li = lane & 15, lk = lane >> 4;
    for (int rt = wave; rt < rt_n; rt += nwaves) {
        const double *Ap = A + rt * 16 + li + lda * lk;
        for (int ct = 0; ct < ct_n; ct += 2) {
            const bool two = ct + 1 < ct_n;
            const double *Bp0 = B + lk + ldb * (ct * 16 + li);
            const double *Bp1 = Bp0 + (two ? ldb * 16 : 0);
            double4_t acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
            // operands of step ks+1 are requested before the MFMAs of step ks issue
            double an = 0.0, b0n = 0.0, b1n = 0.0;
            if (kfull > 0) {
                an = Ap[0];
                b0n = Bp0[0];
                if (two) b1n = Bp1[0];
            }
            for (int ks = 0; ks < kfull; ++ks) {
                const double a = an, b0 = b0n, b1 = b1n;
                if (ks + 1 < kfull) {
                    an = Ap[lda * 4 * (ks + 1)];
                    b0n = Bp0[4 * (ks + 1)];
                    if (two) b1n = Bp1[4 * (ks + 1)];
                }
                acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b0, acc0, 0, 0, 0);
                if (two) acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b1, acc1, 0, 0, 0);
            }
            if (krem) {
                const bool ok = lk < krem;
                const double a = ok ? Ap[lda * 4 * kfull] : 0.0;
                const double b0 = ok ? Bp0[4 * kfull] : 0.0;
                const double b1 = ok ? Bp1[4 * kfull] : 0.0;
                acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b0, acc0, 0, 0, 0);
                if (two) acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b1, acc1, 0, 0, 0);
            }
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                if (t == 1 && !two) break;
                const int col = (ct + t) * 16 + li;
                if (col < Nc) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int row = rt * 16 + lk + 4 * r;
                        const double v = t ? acc1[r] : acc0[r];
                        if (row < M) {
                            C[row + ldc * col] = v;
                            if (MODE == 1) {
                                if (row < dmir)
                                    C[row + dmir + ldc * (col + dmir)] = v;
                                else
                                    C[row - dmir + ldc * (col + dmir)] = -v;
                            }
                        }
                    }
                }
            }
        }
    }
}

// Phase-2 product of the matrix waves for a narrow slice (ncols1 <= 32, 2*nc <= 16), one row tile per wave:
//   acc0/acc1 = G * M1[:, tile 0/1],  acc2 = G2 * M1[:, tile 0]  (its columns nc..2nc-1 are G2 D).
// Operands of k-step ks+1 are requested before the MFMAs of step ks issue.
__device__ __forceinline__ void wave_phase2_fused(const double *G, const double *G2, const double *M1, double *W1,
                                                  double *G2D, int LD, int n, int ncols1, int nc, bool want_g2, int wave,
                                                  int lane) {
    const int rt_n = (n + 15) >> 4;
    const bool two = ncols1 > 16;
    const int kfull = n >> 2, krem = n & 3;
    const int li = lane & 15, lk = lane >> 4;
    for (int rt = wave; rt < rt_n; rt += 4) {
        const double *Ap = G + rt * 16 + li + LD * lk;
        const double *A2p = (want_g2 ? G2 : G) + rt * 16 + li + LD * lk;
        const double *Bp0 = M1 + lk + LD * li;
        const double *Bp1 = Bp0 + (two ? LD * 16 : 0);
        double4_t acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0}, acc2 = {0.0, 0.0, 0.0, 0.0};
        double a = 0.0, a2 = 0.0, b0 = 0.0, b1 = 0.0;
        if (kfull > 0) {
            a = Ap[0];
            a2 = A2p[0];
            b0 = Bp0[0];
            b1 = Bp1[0];
        }
        for (int ks = 0; ks < kfull; ++ks) {
            const double ca = a, ca2 = a2, cb0 = b0, cb1 = b1;
            if (ks + 1 < kfull) {
                a = Ap[LD * 4 * (ks + 1)];
                a2 = A2p[LD * 4 * (ks + 1)];
                b0 = Bp0[4 * (ks + 1)];
                b1 = Bp1[4 * (ks + 1)];
            }
            acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(ca, cb0, acc0, 0, 0, 0);
            if (two) acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(ca, cb1, acc1, 0, 0, 0);
            if (want_g2) acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(ca2, cb0, acc2, 0, 0, 0);
        }
        if (krem) {
            const bool ok = lk < krem;
            const double ra = ok ? Ap[LD * 4 * kfull] : 0.0;
            const double ra2 = ok ? A2p[LD * 4 * kfull] : 0.0;
            const double rb0 = ok ? Bp0[4 * kfull] : 0.0;
            const double rb1 = ok ? Bp1[4 * kfull] : 0.0;
            acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(ra, rb0, acc0, 0, 0, 0);
            if (two) acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(ra, rb1, acc1, 0, 0, 0);
            if (want_g2) acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(ra2, rb0, acc2, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = rt * 16 + lk + 4 * r;
            if (row < n) {
                if (li < ncols1) W1[row + LD * li] = acc0[r];
                if (two && li + 16 < ncols1) W1[row + LD * (li + 16)] = acc1[r];
                if (want_g2 && li >= nc && li < 2 * nc) G2D[row + LD * (li - nc)] = acc2[r];
            }
        }
    }
}

// Persistent form: the grid is (workgroups that fit on the chip); each workgroup walks the work
// items (b, k, s) with stride gridDim.x.  What does not depend on the item stays on chip for the
// whole launch: the LDS tile G holds the drift everywhere except on the union pattern of the
// drives, which is the only part rewritten per item (each thread keeps its pattern entries in
// registers), and the ELL form of the drives is staged in LDS once.  Per item only u_k, dt_k and
// the slice's state columns are read from memory, one item ahead.  Element-wise passes give every
// thread a fixed row (tid % n) and walk columns: no integer division inside the item loop.
#define PCL_NUE2 2  // union-pattern entries per thread held in registers (REG path: n_upos <= 1024)

// TD/TM/TNC: compile-time Hilbert dimension, drive count and slice width (0 = run-time values).  With the shape fixed
// every LDS offset, trip count and divisor is a constant: the specialised instances need far fewer scalar registers.
template <bool JAC, int WU, int TD, int TM, int TNC>  // WU: (drive,value) pairs per pattern entry in registers; -1: general
__global__ __launch_bounds__(512, 4) void pcl_fused_kernel_v2(const KParams p) {
    extern __shared__ double lds[];
    const int d = TD ? TD : p.d, n = 2 * d, m = TD ? TM : p.m, LD = TD ? ((2 * TD + 3) & ~3) + 2 : p.LD, nc = TNC ? TNC : p.nc;
    const int C = TD ? TD : p.cols;  // state columns (specialised instances are unitary: C = d)
    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const bool matrix_wave = wave < 4;
    const int stid = tid - 256;  // index among the stream waves' threads
    const int nn = n * n;

    const int ncols1 = JAC ? (2 + m) * nc : 2 * nc;
    const int n_ell = m * n * p.ell_w;
    double *G = lds;
    double *G2 = G + LD * n;
    double *M1 = G2 + (JAC ? LD * n : 0);
    double *W1 = M1 + LD * ncols1;
    double *G2D = W1 + LD * ncols1;
    double *us = G2D + LD * nc;  // 2 x [u_k (m) | dt_k]: current / next item
    double *ellv_l = us + 2 * (m + 1);
    unsigned short *ellc_l = reinterpret_cast<unsigned short *>(ellv_l + n_ell);
    const long long xd = (long long)n * C;
    const int ew = p.ell_w;
    const bool fused_p2 = ncols1 <= 32 && 2 * nc <= 16;

    // ---- fixed thread coordinates ----------------------------------------------------------------------
    const int ri = tid % n, rj0 = tid / n, rstep = 512 / n;  // all threads: row ri, columns rj0, rj0+rstep, ..
    const bool ract = rj0 < rstep;
    const int sstep = 256 / n;                               // stream waves as a 256-thread group: row si
    const int si = matrix_wave ? 0 : stid % n, sj0 = matrix_wave ? 0 : stid / n;
    const bool sact = !matrix_wave && sj0 < sstep;
    const int hn = n >> 1;                                   // stream waves: row pair (pi, pi+1)
    const int pi = matrix_wave ? 0 : 2 * (stid % hn), pj0 = matrix_wave ? 0 : stid / hn, pstep = max(256 / hn, 1);
    const bool pact = !matrix_wave && pj0 < pstep;

    // ---- launch-invariant state -------------------------------------------------------------------------
    if (!p.g0_batch_stride)
        load_tile<512>(p.G0, G, n, LD, tid);
    constexpr int WUR = WU > 0 ? WU : 1;
    int un_idx[PCL_NUE2];
    double un_g0[PCL_NUE2];
    unsigned char un_l[PCL_NUE2][WUR];
    double un_v[PCL_NUE2][WUR];
    if (WU > 0) {
#pragma unroll
        for (int r = 0; r < PCL_NUE2; ++r) {
            const int q = tid + 512 * r;
            un_idx[r] = -1;
            un_g0[r] = 0.0;
#pragma unroll
            for (int w = 0; w < WUR; ++w) {
                un_l[r][w] = 0;
                un_v[r][w] = 0.0;
            }
            if (q < p.n_upos) {
                const int pos = p.upos[q];
                un_idx[r] = (pos % n) + LD * (pos / n);
                un_g0[r] = p.G0[pos];
#pragma unroll
                for (int w = 0; w < WUR; ++w) {
                    un_l[r][w] = p.uell_l[q * WUR + w];
                    un_v[r][w] = p.uell_v[q * WUR + w];
                }
            }
        }
    }
    const bool stage = JAC && p.ell_lds;
    if (stage) {
        for (int e = tid; e < n_ell; e += 512) {
            ellv_l[e] = p.ell_val[e];
            ellc_l[e] = (unsigned short)p.ell_col[e];
        }
    }

    // ---- per-item inputs, requested one item ahead -------------------------------------------------
    const int n_items = p.batch * p.K * p.S;
    const bool pf_x = nc <= rstep;  // one state element per thread: prefetchable
    double pf_v = 0.0, pf_xn = 0.0, pf_xc = 0.0;
    auto request = [&](int item) {
        const int s = item % p.S;
        const int k = (item / p.S) % p.K;
        const int b = item / (p.S * p.K);
        const double *zk = p.Z + (long long)b * p.z_batch_stride + (long long)k * p.z_dim;
        if (tid <= m) pf_v = zk[tid < m ? p.u_off + tid : p.dt_off];
        pf_xn = pf_xc = 0.0;
        if (pf_x && ract && rj0 < min(nc, C - s * nc)) {
            const int x_off = p.x_offs[p.z_batch_stride ? 0 : b];
            const long long o = x_off + (long long)(s * nc + rj0) * n + ri;
            pf_xc = zk[o];
            pf_xn = zk[p.z_dim + o];
        }
    };
    int cur = 0;
    if ((int)blockIdx.x < n_items) {
        request(blockIdx.x);
        if (tid <= m) us[tid] = pf_v;
    }
    __syncthreads();  // G = drift, tables staged, us[0] valid

    for (int item = blockIdx.x; item < n_items; item += gridDim.x) {
        const int s = item % p.S;
        const int k = (item / p.S) % p.K;
        const int b = item / (p.S * p.K);
        const int c0 = s * nc;
        const int nce = min(nc, C - c0);
        const double *zk = p.Z + (long long)b * p.z_batch_stride + (long long)k * p.z_dim;
        const double *zn = zk + p.z_dim;
        const int x_off = p.x_offs[p.z_batch_stride ? 0 : b];
        const double *usc = us + cur * (m + 1);
        const double h = usc[m];
        const double c1 = 0.5 * h, c2 = h * h * (1.0 / 12.0);

        // ---- phase 0: G(u_k) on the union pattern, S, D -> LDS -----------------------------------------
        if (WU < 0 && p.g0_batch_stride) {  // per-member drift: the whole tile changes with b
            const double *G0b = p.G0 + (long long)b * p.g0_batch_stride;
            for (int e = tid; e < nn; e += 512) G[(e % n) + LD * (e / n)] = G0b[e];
            __syncthreads();  // the dense rewrite lands before the pattern update
        }
        {
            if (WU > 0) {
#pragma unroll
                for (int r = 0; r < PCL_NUE2; ++r)
                    if (un_idx[r] >= 0) {
                        double g = un_g0[r];
#pragma unroll
                        for (int w = 0; w < WUR; ++w) g += usc[un_l[r][w]] * un_v[r][w];
                        G[un_idx[r]] = g;
                    }
            } else {
                const double *G0b = p.G0 + (long long)b * p.g0_batch_stride;
                for (int q = tid; q < p.n_upos; q += 512) {
                    const int pos = p.upos[q];
                    double g = G0b[pos];
                    const double *cf = p.ucoef + (long long)q * m;
                    for (int l = 0; l < m; ++l) g += usc[l] * cf[l];
                    G[(pos % n) + LD * (pos / n)] = g;
                }
            }
        }
        if (pf_x) {
            if (ract && rj0 < nc) {
                M1[ri + LD * rj0] = pf_xn + pf_xc;
                M1[ri + LD * (nc + rj0)] = pf_xn - pf_xc;
            }
        } else if (ract) {
            for (int c = rj0; c < nc; c += rstep) {
                double xs = 0.0, xdv = 0.0;
                if (c < nce) {
                    const double xn = zn[x_off + (c0 + c) * n + ri], xc = zk[x_off + (c0 + c) * n + ri];
                    xs = xn + xc;
                    xdv = xn - xc;
                }
                M1[ri + LD * c] = xs;
                M1[ri + LD * (nc + c)] = xdv;
            }
        }
        __syncthreads();

        // ---- phase 1: matrix waves G^2 ; stream waves G_l D ---------------------------------------------
        if (JAC) {
            if (matrix_wave) {
                if (p.iso)
                    wave_rowgemm<1>(G, LD, G, LD, G2, LD, n, d, n, wave, 4, lane, d);
                else
                    wave_rowgemm<0>(G, LD, G, LD, G2, LD, n, n, n, wave, 4, lane, 0);
            } else if (sact) {
                const double *Dm = M1 + LD * nc;
                for (int cl = sj0; cl < m * nc; cl += sstep) {
                    const int l = cl / nc, c = cl - l * nc;
                    const int base = (l * n + si) * ew;
                    double acc = 0.0;
                    if (stage) {
                        for (int q = 0; q < ew; ++q) acc += ellv_l[base + q] * Dm[ellc_l[base + q] + LD * c];
                    } else {
                        for (int q = 0; q < ew; ++q) acc += p.ell_val[base + q] * Dm[p.ell_col[base + q] + LD * c];
                    }
                    M1[si + LD * (2 * nc + cl)] = acc;
                }
            }
            __syncthreads();
        }

        // ---- phase 2: matrix waves W1 = G M1, G2D = G^2 D ; stream waves the block copies -------------------
        const long long bk = (long long)b * p.K + k;
        double *jb = JAC ? p.jac + bk * p.jac_per : nullptr;
        const long long blk = p.compact ? (long long)nn : (long long)C * nn;  // size of seg 0 / seg 1 in `jac`
        if (matrix_wave) {
            if (fused_p2) {
                wave_phase2_fused(G, G2, M1, W1, G2D, LD, n, ncols1, nc, JAC, wave, lane);
            } else {
                wave_rowgemm<0>(G, LD, M1, LD, W1, LD, n, ncols1, n, wave, 4, lane, 0);
                if (JAC) wave_rowgemm<0>(G2, LD, M1 + LD * nc, LD, G2D, LD, n, nc, n, wave, 4, lane, 0);
            }
        } else if (JAC && pact) {
            int cbeg = c0, cend = c0 + nce;
            if (p.compact) {  // unique blocks only: slice 0 writes the single copy
                cbeg = 0;
                cend = (s == 0) ? 1 : 0;
            }
            for (int j = pj0; j < n; j += pstep) {
                const double g0 = G[pi + LD * j], g1 = G[pi + 1 + LD * j];
                const double h0 = G2[pi + LD * j], h1 = G2[pi + 1 + LD * j];
                double bp0, bp1, bm0, bm1;
                bpm_entry((pi == j) ? 1.0 : 0.0, c1, c2, g0, h0, bp0, bm0);
                bpm_entry((pi + 1 == j) ? 1.0 : 0.0, c1, c2, g1, h1, bp1, bm1);
                double *o0 = jb + (long long)cbeg * nn + (pi + n * j);
                for (int c = cbeg; c < cend; ++c, o0 += nn) {
                    store2(o0, bp0, bp1, p.nt);
                    store2(o0 + blk, bm0, bm1, p.nt);
                }
            }
        }
        // inputs of this workgroup's next item: in flight during the rest of this one
        if (item + (int)gridDim.x < n_items) request(item + gridDim.x);
        __syncthreads();
        if (!JAC) {  // eval only: delta needs G (G D), a second dependent product
            if (matrix_wave) wave_rowgemm<0>(G, LD, W1 + LD * nc, LD, G2D, LD, n, nc, n, wave, 4, lane, 0);
            __syncthreads();
        }

        // ---- phase 3: column outputs: block 0 -> delta and d/ddt, block 1+l -> d/du_l ----------------------
        if (ract) {
            const double *GDm = W1 + LD * nc;
            const int ncl = JAC ? (1 + m) * nc : nc;
            for (int cl = rj0; cl < ncl; cl += rstep) {
                const int lb = cl / nc, c = cl - lb * nc;
                if (c >= nce) continue;
                const long long r = (long long)(c0 + c) * n + ri;
                if (lb == 0) {
                    const double gs = W1[ri + LD * c], g2d = G2D[ri + LD * c];
                    if (p.delta) p.delta[bk * xd + r] = M1[ri + LD * (nc + c)] - c1 * gs + c2 * g2d;
                    if (JAC) jb[2 * blk + ((long long)(c0 + c) * (m + 1) + m) * n + ri] = -0.5 * gs + (h * (1.0 / 6.0)) * g2d;
                } else {
                    const int l = lb - 1;
                    const int base = (l * n + ri) * ew;
                    double acc = 0.0;
                    if (stage) {
                        for (int q = 0; q < ew; ++q) {
                            const int col = ellc_l[base + q];
                            acc += ellv_l[base + q] * (-c1 * M1[col + LD * c] + c2 * GDm[col + LD * c]);
                        }
                    } else {
                        for (int q = 0; q < ew; ++q) {
                            const int col = p.ell_col[base + q];
                            acc += p.ell_val[base + q] * (-c1 * M1[col + LD * c] + c2 * GDm[col + LD * c]);
                        }
                    }
                    jb[2 * blk + ((long long)(c0 + c) * (m + 1) + l) * n + ri] = acc + c2 * W1[ri + LD * (nc + cl)];
                }
            }
        }
        if (tid <= m) us[(cur ^ 1) * (m + 1) + tid] = pf_v;  // requested during phase 2
        cur ^= 1;
        __syncthreads();  // LDS is rewritten by the next item's phase 0
    }
}
