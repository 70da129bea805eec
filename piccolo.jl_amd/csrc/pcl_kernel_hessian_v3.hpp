// pcl_kernel_hessian_v3.hpp -- Hessian of the Lagrangian, version 3: one persistent workgroup of
// eight wavefronts per CU, ONE workgroup per interval (no cross-workgroup reduction), jobs split BY DRIVE instead of by
// state-column chunk so that every matrix-core pass uses all 16 operand columns and every output vector leaves the CU as
// one contiguous slab of 16-byte stores.
#pragma once

// ------------------------------------------------------------------------------------------
// With M = mu_k (n x d), D = X_{k+1} - X_k, S = X_{k+1} + X_k, per state column (see pcl_kernels_hessian.hpp for the algebra):
//     A1 = G^T M,  A2 = G^T A1,  P_l = G_l^T M,  Q_l = G^T P_l,  R_l = G_l^T A1,  E_l = G_l D
//     d2/du_l dX_k   = -(h/2) P_l - (h^2/12)(Q_l + R_l)      d2/du_l dX_{k+1} = -(h/2) P_l + (h^2/12)(Q_l + R_l)
//     d2/dh  dX_k    = -A1/2 - (h/6) A2                        d2/dh  dX_{k+1}  = -A1/2 + (h/6) A2
//     scalars: <P_i,E_j> + <P_j,E_i>,   -<P_j,S>/2 + (h/6)(<Q_j,D> + <A1,E_j>),   <A2,D>/6
// The d state columns are cut into groups of 16 (two groups at d = 27).  Roles:
//     state wave g (waves 0, 1)     round 1: A1 of group g = G^T M_g            round 2: A2 of group g = G^T A1_g
//     drive wave l (waves 2 .. 2+m) round r: P_l, E_l of group r (sparse, lane = row), Q_l = G^T P_l of group r
// i.e. two rounds of ONE 16-column matrix-core pass per wave (56 MFMAs at n = 54): 896 MFMAs per interval, all tiles full.
// Per round:  [B] drive waves form P_l (-> own LDS tile = the MFMA b operand, and registers) and E_l (-> LDS, registers)
//             [MFMA]  barrier (A1, P_j, E_j visible)
//             [C] scalar partial sums (read the other drive waves' P_j / E_j tiles)     barrier (tiles may be overwritten)
//             [D] accumulators -> own tile (transpose to lane = row), R_l from A1, the two output vectors staged in the
//                 wave's two tiles and written as contiguous slabs (16 columns x n rows = 6.9 KB each) with 16-byte stores
// Scalars: per-lane partial sums -> wave sums (fixed shuffle tree) -> one owner per entry: deterministic, no atomics.
// LDS map (doubles): G [LD*n] | Ms [LD*32] | Ds [LD*32] | Ss [LD*32] | A1s [LD*32] | T [TM][LD*16] | E [TM][LD*16] | us [2][m+1] | scal
// (162 KB at d = 27 with LD = n: one workgroup per CU).  Shapes whose tiles do not fit run version 2.
// ------------------------------------------------------------------------------------------
#define PCL_NUE_H3 2  // union-pattern entries per thread in registers (512 threads: n_upos <= 1024)

template <int EW, int TM, int TD, bool ANTI>
__global__ __launch_bounds__(512, 2) void pcl_hess_kernel_v3(const KParams p) {
    extern __shared__ double lds[];
    constexpr int m = TM;
    constexpr int NSC = (TM + 1) * (TM + 2) / 2;
    constexpr int NPAIR = TM * (TM + 1) / 2;
    const int d = TD ? TD : p.d, n = 2 * d;
    const int LD = (d & 1) ? n : (TD ? ((2 * TD + 3) & ~3) + 2 : p.LD);  // 2*odd: conflict-free b-operand reads
    const int cols = d;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int li = lane & 15, lk = lane >> 4;
    const int nn = n * n;
    const long long xd = (long long)n * cols;
    const int kfull = n >> 2, krem = n & 3;
    const int ng = (cols + 15) >> 4;  // 1 or 2 groups of <= 16 state columns
    const int tile = LD * 16;

    double *G = lds;
    double *Ms = G + LD * n;
    double *Ds = Ms + 2 * tile;
    double *Ss = Ds + 2 * tile;
    double *A1s = Ss + 2 * tile;
    double *Tt = A1s + 2 * tile;
    double *Et = Tt + TM * tile;
    double *us = Et + TM * tile;
    double *scal = us + 2 * (TM + 1);  // [NSC] entries | [2] <A2,D> of the two groups

    const bool state_wave = wave < 2, drive_wave = wave >= 2 && wave - 2 < TM;
    const int l = wave - 2;                      // drive of a drive wave
    double *Tl = Tt + (drive_wave ? l : 0) * tile, *El = Et + (drive_wave ? l : 0) * tile;

    // ---- launch-invariant state ------------------------------------------------------------------------------------------
    // this drive's ELL rows (row = lane) of G_l and of G_l^T
    unsigned short er_c[EW], et_c[EW];
    double er_v[EW], et_v[EW];
#pragma unroll
    for (int q = 0; q < EW; ++q) {
        er_c[q] = et_c[q] = 0;
        er_v[q] = et_v[q] = 0.0;
        if (drive_wave && lane < n) {
            if (q < p.ell_w) {
                er_c[q] = (unsigned short)p.ell_col[(l * n + lane) * p.ell_w + q];
                er_v[q] = p.ell_val[(l * n + lane) * p.ell_w + q];
            }
            if (ANTI) {  // G_l^T = -G_l: the same table, the sums are negated
                et_c[q] = er_c[q];
                et_v[q] = -er_v[q];
            } else if (q < p.ellt_w) {
                et_c[q] = (unsigned short)p.ellt_col[(l * n + lane) * p.ellt_w + q];
                et_v[q] = p.ellt_val[(l * n + lane) * p.ellt_w + q];
            }
        }
    }
    // union-pattern entries of this thread
    int un_idx[PCL_NUE_H3];
    double un_g0[PCL_NUE_H3], un_v[PCL_NUE_H3][2];
    unsigned char un_l[PCL_NUE_H3][2];
#pragma unroll
    for (int r = 0; r < PCL_NUE_H3; ++r) {
        const int q = tid + 512 * r;
        un_idx[r] = -1;
        un_g0[r] = 0.0;
        un_l[r][0] = un_l[r][1] = 0;
        un_v[r][0] = un_v[r][1] = 0.0;
        if (q < p.n_upos) {
            const int pos = p.upos[q];
            un_idx[r] = (pos % n) + LD * (pos / n);
            un_g0[r] = p.ug0[q];  // drift at the pattern entry (table: no load that depends on upos); per-member drifts are read per member
            for (int w = 0; w < p.uell_w; ++w) {  // host guarantees uell_w <= 2
                un_l[r][w] = p.uell_l[q * p.uell_w + w];
                un_v[r][w] = p.uell_v[q * p.uell_w + w];
            }
        }
    }
    if (!p.g0_batch_stride)
        load_tile<512>(p.G0, G, n, LD, tid);
    for (int e = tid; e < 8 * tile; e += 512) Ms[e] = 0.0;  // Ms, Ds, Ss, A1s: columns beyond d stay zero (they are MFMA operands)

    // ---- work: a contiguous range of intervals per workgroup (consecutive items share the ensemble member) ------------------
    const int n_items = p.batch * p.K;
    const int item_lo = (int)((long long)n_items * blockIdx.x / gridDim.x), item_hi = (int)((long long)n_items * (blockIdx.x + 1) / gridDim.x);
    constexpr int NPF = TD ? (2 * TD * TD + 511) / 512 : 4;  // ceil(x_dim / 512); run-time shapes: n*d <= 2048
    double pf_u = 0.0, pmu[NPF], pxn[NPF], pxc[NPF];
    auto request_u = [&](int item) {  // the next item's controls / time step (one item ahead)
        const int k = item % p.K, b = item / p.K;
        const double *zk = p.Z + (long long)b * p.z_batch_stride + (long long)k * p.z_dim;
        if (tid <= m) pf_u = zk[tid < m ? p.u_off + tid : p.dt_off];
    };
    auto request_x = [&](int item) {  // an item's multipliers and states: requested during the previous item's last output phase
        const int k = item % p.K, b = item / p.K;
        const double *zk = p.Z + (long long)b * p.z_batch_stride + (long long)k * p.z_dim + p.x_offs[p.z_batch_stride ? 0 : b];
        const double *mu = p.mu + ((long long)b * p.K + k) * xd;
#pragma unroll
        for (int i = 0; i < NPF; ++i) {
            const int e = tid + 512 * i;
            pmu[i] = pxn[i] = pxc[i] = 0.0;
            if (e < xd) {
                pmu[i] = mu[e];
                pxc[i] = zk[e];
                pxn[i] = zk[p.z_dim + e];
            }
        }
    };
    int cur = 0, drift_b = -1;
    if (item_lo < item_hi) {
        request_u(item_lo);
        request_x(item_lo);
        if (tid <= m) us[tid] = pf_u;
    }
    __syncthreads();

#ifdef PCL_PROFILE
    int stamp = 0;
#define PCL_H3STAMP()                                                                                                 \
    do {                                                                                                              \
        if (p.dbg && blockIdx.x == 0 && tid == 128 && item == item_lo + 1 && stamp < 60) p.dbg[stamp++] = (long long)__builtin_amdgcn_s_memtime(); \
    } while (0)
#define PCL_H3_STORES (!(p.prof & 1))  // profile builds only: option profile_flags bit 0 drops the output stores
#else
#define PCL_H3STAMP() do { } while (0)
#define PCL_H3_STORES true
#endif
    for (int item = item_lo; item < item_hi; ++item) {
        const int k = item % p.K, b = item / p.K;
        PCL_H3STAMP();  // 0 item start
        const double *usc = us + cur * (TM + 1);
        const double h = usc[m];
        const double c1 = 0.5 * h, c2 = h * h * (1.0 / 12.0), h6 = h * (1.0 / 6.0);
        const long long bk = (long long)b * p.K + k;
        double *H = p.hess + bk * p.hess_per;
        double *H3 = H + NSC, *H4 = H3 + (long long)m * xd, *H5 = H4 + xd, *H6 = H5 + (long long)m * xd;

        // ---- phase A: G(u_k) on the union pattern, the item's inputs -> LDS ---------------------------------------------------
        const double *G0b = p.G0 + (long long)b * p.g0_batch_stride;
        if (p.g0_batch_stride && drift_b != b) {  // per-member drift: every position receives its final value only
            for (int e = tid; e < nn; e += 512)
                if (p.umap[e] < 0) G[(e % n) + LD * (e / n)] = G0b[e];
        }
#pragma unroll
        for (int r = 0; r < PCL_NUE_H3; ++r)
            if (un_idx[r] >= 0) {
                if (p.g0_batch_stride && drift_b != b) un_g0[r] = G0b[p.upos[tid + 512 * r]];
                G[un_idx[r]] = (un_g0[r] + usc[un_l[r][0]] * un_v[r][0]) + usc[un_l[r][1]] * un_v[r][1];
            }
        drift_b = b;
        {  // the item's inputs (requested during the previous item) -> LDS
#pragma unroll
            for (int i = 0; i < NPF; ++i) {
                const int e = tid + 512 * i;
                if (e < xd) {
                    const int c = e / n, o = (e - c * n) + LD * c;  // groups are consecutive 16-column tiles: state column c = tile column c
                    Ms[o] = pmu[i];
                    Ds[o] = pxn[i] - pxc[i];
                    Ss[o] = pxn[i] + pxc[i];
                }
            }
        }
        if (item + 1 < item_hi) request_u(item + 1);  // in flight during the matrix phases
        PCL_H3STAMP();  // 1 phase A issued
        __syncthreads();
        PCL_H3STAMP();  // 2 alpha passed

        // per-lane partial sums of the scalar entries.  Every wave owns two columns of each group in phase [C]:
        //   sc[e], e < NPAIR : <P_i,E_j> + <P_j,E_i>            (e = i(i+1)/2 + j, j <= i)
        //   sc[NPAIR + j]    : -<P_j,S>/2 + (h/6) <A1,E_j>
        // drive wave j additionally <Q_j,D> (accumulator layout), state wave g <A2,D> of its group.
        constexpr int NACC = NPAIR + TM;
        double sc[NACC], sQD = 0.0, sA2D = 0.0;
#pragma unroll
        for (int e = 0; e < NACC; ++e) sc[e] = 0.0;

        for (int rd = 0; rd < 2; ++rd) {
            // group this wave works on in this round: drive waves walk the groups, state wave g owns group g in both rounds
            const int g = state_wave ? wave : rd;
            const bool active = (state_wave || drive_wave) && g < ng;
            const int cg = active ? min(16, cols - 16 * g) : 0;
            // ---- [B] drive waves: P_l, E_l of the group (lane = row); loads of eight columns are issued together ------------------
            if (drive_wave && active && lane < n) {
                const double *Mg = Ms + g * tile, *Dg = Ds + g * tile;
#pragma unroll
                for (int c0 = 0; c0 < 16; c0 += 4) {
                    double mv[4][EW], dv[4][EW];
#pragma unroll
                    for (int c = 0; c < 4; ++c)
#pragma unroll
                        for (int q = 0; q < EW; ++q) {
                            mv[c][q] = Mg[et_c[q] + LD * (c0 + c)];  // (columns beyond the group hold zeros)
                            dv[c][q] = Dg[er_c[q] + LD * (c0 + c)];
                        }
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        double pv = 0.0, ev = 0.0;
#pragma unroll
                        for (int q = 0; q < EW; ++q) {
                            pv += et_v[q] * mv[c][q];
                            ev += er_v[q] * dv[c][q];
                        }
                        Tl[lane + LD * (c0 + c)] = pv;  // the MFMA b operand
                        El[lane + LD * (c0 + c)] = ev;
                    }
                }
            }
            wave_lds_sync();
            PCL_H3STAMP();  // [B] done
            // ---- [MFMA] one 16-column pass: acc = G^T * B,  B = M_g / A1_g (state waves) or P_l (drive waves) -------------------
            double4_t acc[PCL_MAXRT];
#pragma unroll
            for (int t = 0; t < PCL_MAXRT; ++t) acc[t] = double4_t{0.0, 0.0, 0.0, 0.0};
            if (active) {
                const double *B = state_wave ? (rd == 0 ? Ms : A1s) + g * tile : Tl;
                const double *Bp = B + lk + LD * li;
                const double *Ap[PCL_MAXRT];
                bool rok[PCL_MAXRT];
#pragma unroll
                for (int t = 0; t < PCL_MAXRT; ++t) {
                    rok[t] = t * 16 < n;
                    Ap[t] = G + lk + LD * ((rok[t] ? t * 16 : 0) + li);
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
                        if (rok[t]) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[t], bb, acc[t], 0, 0, 0);
                }
                if (krem) {
                    const bool ok = lk < krem;
                    const double bb = ok ? Bp[4 * kfull] : 0.0;
#pragma unroll
                    for (int t = 0; t < PCL_MAXRT; ++t)
                        if (rok[t]) {
                            const double a = ok ? Ap[t][4 * kfull] : 0.0;
                            acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bb, acc[t], 0, 0, 0);
                        }
                }
                if (state_wave && rd == 0) {  // A1 of the group: accumulator layout (row = 16t + lk + 4r, column li) -> A1s
#pragma unroll
                    for (int t = 0; t < PCL_MAXRT; ++t)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int row = t * 16 + lk + 4 * r;
                            if (row < n) A1s[g * tile + row + LD * li] = acc[t][r];
                        }
                }
            }
            PCL_H3STAMP();  // MFMA done
            __syncthreads();  // beta: A1 (round 1) and every drive wave's P_j / E_j tiles are visible
            PCL_H3STAMP();  // beta passed
            // ---- [C] scalar partial sums, balanced: EVERY wave takes two columns of the drive waves' group (rd), all drives ------
            if (rd < ng && lane < n && wave < TM + 2) {  // (the waves that own a parking region below)
                const int cgr = min(16, cols - 16 * rd);
                const double *A1g = A1s + rd * tile + lane, *Sg = Ss + rd * tile + lane;
                for (int c = wave; c < cgr; c += TM + 2) {
                    {
                        double pj[TM], ej[TM];
#pragma unroll
                        for (int j = 0; j < TM; ++j) {
                            pj[j] = Tt[j * tile + lane + LD * c];
                            ej[j] = Et[j * tile + lane + LD * c];
                        }
                        const double a1 = A1g[LD * c], sv = Sg[LD * c];
#pragma unroll
                        for (int i = 0; i < TM; ++i) {
#pragma unroll
                            for (int j = 0; j <= i; ++j) sc[i * (i + 1) / 2 + j] += pj[i] * ej[j] + pj[j] * ej[i];
                            sc[NPAIR + i] += -0.5 * (pj[i] * sv) + h6 * (a1 * ej[i]);
                        }
                    }
                }
            }
            if (drive_wave && active) {  // <Q_l, D> straight from the accumulator layout
                const double *Dg = Ds + g * tile;
#pragma unroll
                for (int t = 0; t < PCL_MAXRT; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int row = t * 16 + lk + 4 * r;
                        if (row < n && li < cg) sQD += acc[t][r] * Dg[row + LD * li];
                    }
            }
            PCL_H3STAMP();  // [C] done
            __syncthreads();  // gamma: nobody reads another wave's tiles any more
            PCL_H3STAMP();  // gamma passed
            if (rd == 1 && item + 1 < item_hi) request_x(item + 1);  // the next item's inputs travel during the output phase
            // ---- [D] outputs: one contiguous run of n doubles per (vector, column), written straight from lane = row -------------
            if (drive_wave && active) {
                double Pv[16];  // P_l (lane = row) back from the wave's tile before Q_l replaces it
#pragma unroll
                for (int c = 0; c < 16; ++c) Pv[c] = lane < n ? Tl[lane + LD * c] : 0.0;
                wave_lds_sync();
#pragma unroll
                for (int t = 0; t < PCL_MAXRT; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int row = t * 16 + lk + 4 * r;
                        if (row < n) Tl[row + LD * li] = acc[t][r];  // Q_l, transposed to lane = row through the wave's tile
                    }
                wave_lds_sync();
                if (lane < n) {
                    const double *A1g = A1s + g * tile;
                    double *o3 = H3 + (long long)l * xd + (long long)(16 * g) * n + lane, *o5 = H5 + (long long)l * xd + (long long)(16 * g) * n + lane;
#pragma unroll
                    for (int c0 = 0; c0 < 16; c0 += 4) {
                        double qv[4], av[4][EW];
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            qv[c] = Tl[lane + LD * (c0 + c)];
#pragma unroll
                            for (int q = 0; q < EW; ++q) av[c][q] = A1g[et_c[q] + LD * (c0 + c)];
                        }
#pragma unroll
                        for (int c = 0; c < 4; ++c)
                            if (c0 + c < cg) {
                                double rr = 0.0;  // R_l = G_l^T A1
#pragma unroll
                                for (int q = 0; q < EW; ++q) rr += et_v[q] * av[c][q];
                                const double kt = c2 * (qv[c] + rr), pl = -c1 * Pv[c0 + c];
                                if (PCL_H3_STORES) {
                                    if (p.nt) {
                                        __builtin_nontemporal_store(pl - kt, o3 + (long long)(c0 + c) * n);
                                        __builtin_nontemporal_store(pl + kt, o5 + (long long)(c0 + c) * n);
                                    } else {
                                        o3[(long long)(c0 + c) * n] = pl - kt;
                                        o5[(long long)(c0 + c) * n] = pl + kt;
                                    }
                                }
                            }
                    }
                }
            } else if (state_wave && active && rd == 1) {
                // A2 of the group -> the group's Ms tile (M is no longer needed): transposed to lane = row
                double *Mg = Ms + g * tile;
#pragma unroll
                for (int t = 0; t < PCL_MAXRT; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int row = t * 16 + lk + 4 * r;
                        if (row < n) Mg[row + LD * li] = acc[t][r];
                    }
                wave_lds_sync();
                if (lane < n) {
                    const double *A1g = A1s + g * tile + lane, *Dg = Ds + g * tile + lane;
                    double *o4 = H4 + (long long)(16 * g) * n + lane, *o6 = H6 + (long long)(16 * g) * n + lane;
#pragma unroll
                    for (int c0 = 0; c0 < 16; c0 += 4) {
                        double a1[4], a2[4], dd[4];
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            a1[c] = A1g[LD * (c0 + c)];
                            a2[c] = Mg[lane + LD * (c0 + c)];
                            dd[c] = Dg[LD * (c0 + c)];
                        }
#pragma unroll
                        for (int c = 0; c < 4; ++c)
                            if (c0 + c < cg) {
                                sA2D += a2[c] * dd[c];
                                if (PCL_H3_STORES) {
                                    o4[(long long)(c0 + c) * n] = -0.5 * a1[c] - h6 * a2[c];
                                    o6[(long long)(c0 + c) * n] = -0.5 * a1[c] + h6 * a2[c];
                                }
                            }
                    }
                }
            }
            PCL_H3STAMP();  // [D] issued
            // (round 2's [B] writes only the drive waves' own tiles, which their own [D] finished reading: no barrier here)
        }
        // ---- scalar entries: the lanes park their partial sums in LDS ([entry][row], in the two regions only this wave still
        //      uses: a drive wave's two tiles, a state wave's M and S tiles), then 16 threads per entry add them in a fixed order ----
        double *park = drive_wave ? Tl : (state_wave ? Ms + wave * tile : nullptr);
        double *park2 = drive_wave ? El : (state_wave ? Ss + wave * tile : nullptr);
        if (park) {
            wave_lds_sync();  // this wave's own reads of the two regions ([D]) are complete
            if (lane < n) {
#pragma unroll
                for (int e = 0; e < NACC; ++e) {
                    const int f = e * n + lane;  // NACC * n <= 2 * tile (= 32 LD): the two regions hold every entry
                    (f < tile ? park + f : park2 + (f - tile))[0] = sc[e];
                }
            }
            const double tq = wave_sum(drive_wave ? sQD : sA2D);
            if (lane == 0) scal[NACC + wave] = tq;  // waves 0, 1: <A2,D> of their group; wave 2 + j: <Q_j,D>
        }
        if (tid <= m) us[(cur ^ 1) * (TM + 1) + tid] = pf_u;
        cur ^= 1;
        PCL_H3STAMP();  // scalars reduced
        __syncthreads();  // scal complete; every LDS region has been read: the next item may rewrite Ms / Ds / Ss / G
        {
            // entry e of wave w: n row partials at flat offset e*n of the wave's two regions; thread (e, s) adds one half of the
            // rows of wave s >> 1 in order, a DPP row sum adds the 16 threads of the entry (the same order in every launch)
            const int e = tid >> 4, sgm = tid & 15;
            double t = 0.0;
            if (e < NACC) {
                const int w = sgm >> 1, hn = (n + 1) >> 1;
                const bool dw = w >= 2 && w - 2 < TM;
                if (dw || w < 2) {
                    const double *r1 = dw ? Tt + (w - 2) * tile : Ms + w * tile;
                    const double *r2 = dw ? Et + (w - 2) * tile : Ss + w * tile;
                    const int lo = (sgm & 1) * hn, hi = min(n, lo + hn);
                    for (int q = lo; q < hi; ++q) {
                        const int f = e * n + q;
                        t += f < tile ? r1[f] : r2[f - tile];
                    }
                }
            }
            t = row16_sum(t);  // the same bits in all 16 threads of the entry
            if (sgm == 0 && e < NACC) scal[e] = t;
        }
        __syncthreads();
        if (tid < NSC) {
            double t;
            if (tid < NACC)
                t = tid < NPAIR ? c2 * scal[tid] : scal[tid] + h6 * scal[NACC + 2 + (tid - NPAIR)];
            else
                t = (scal[NACC] + scal[NACC + 1]) * (1.0 / 6.0);
            H[tid] = t;
        }
    }
}
