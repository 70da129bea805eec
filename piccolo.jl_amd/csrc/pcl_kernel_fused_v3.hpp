// pcl_kernel_fused_v3.hpp -- the matrix-core fused residual + Jacobian kernel (the default of rounds 1-2, now kernel_version = 3).
#pragma once

// ------------------------------------------------------------------------------------------
// Fused residual + Jacobian kernel, version 3 (default): ONE persistent workgroup per CU, four
// "matrix" wavefronts + four "stream" wavefronts, ONE workgroup barrier per work item (b, k, s).
//
//   stream waves  copy the item's -B^+ / B^- values out of the LDS tiles G, G^2 into registers and
//                 then do nothing but issue the replicated 16-byte block stores (the HBM-roofline
//                 stream; they are the waves that sit in the store queue's back-pressure);
//   matrix waves  meanwhile work wave-synchronously (no workgroup barrier among them):
//                 (1) the item's state columns in chunks of ncw columns, one chunk per wave at a time:
//                     M = [S | D | G_l D] -> G*M on the f64 matrix cores -> delta, d/ddt, d/du_l
//                     straight from the accumulator layout to HBM;
//                 (2) G(u) and G^2 of the workgroup's NEXT item into the other half of the
//                     double-buffered G / G^2 tiles (every matrix wave rewrites the whole union
//                     pattern of G itself - identical values - so no wave waits for another before
//                     its G^2 tiles).
// Item time = max(store stream, matrix work); with the matrix work a fraction of the stream the
// kernel runs at the store stream's rate.
// LDS map (doubles): G [2][LD*n] | G2 [2][LD*n] | per matrix wave: M [LD*CW] GD [LD*ncw] G2D [LD*ncw] |
//                    us [3][m+1] | union values | ELL values | (u16) union LDS offsets, ELL columns | (u8) union drives
// ------------------------------------------------------------------------------------------


struct V3Tables {  // launch-invariant tables, in LDS when they fit (else in memory)
    const double *unv;          // [n_upos*uw] drive coefficients of the union pattern
    const unsigned char *unl;   // [n_upos*uw] drive index
    const unsigned short *uni;  // [n_upos] LDS offset (row + LD*col) of the pattern entry
    const double *ung0;         // [n_upos] drift value at the pattern entry (shared-drift case)
    const double *ellv;         // [m*n*ew]
    const unsigned short *ellc;
};

// EW: ELL width held in registers for m <= PCL_MREG drives (0: general, tables in LDS / memory).
// TD/TM/TNCW: compile-time Hilbert dimension, drive count, chunk width (0 = run-time values).
// MERIT: the instance that also forms the reduce payload's dot products per state column (pcl_eval_jac_merit_dev); a separate
// instance because the extra live values cost the plain one registers (246 -> 256 VGPRs and a spill at <2,27,6,2>).
template <int EW, int TD, int TM, int TNCW, bool MERIT = false>
__global__ __launch_bounds__(512, 2) void pcl_fused_kernel_v3(const KParams p) {
    extern __shared__ double lds[];
    const int d = TD ? TD : p.d, n = 2 * d, m = TD ? TM : p.m, LD = TD ? ((2 * TD + 3) & ~3) + 2 : p.LD;
    const int tid = threadIdx.x;
#ifdef PCL_PROFILE
    if (p.dbg && tid == 0 && blockIdx.x < PCL_DBG_WG) p.dbg[64 + blockIdx.x] = (long long)__builtin_amdgcn_s_memrealtime();
#define PCL_S(i)                                                                                                            \
    do {                                                                                                                    \
        if (p.dbg && blockIdx.x == 0 && (threadIdx.x & 63) == 0 && (p.prof & 2)) p.dbg[i] = (long long)__builtin_amdgcn_s_memtime(); \
    } while (0)
#else
#define PCL_S(i) do { } while (0)
#endif
    if (tid == 0) PCL_S(40);  // kernel entry
    const int wave = tid >> 6, lane = tid & 63;
    const int nn = n * n;
    const int ew = p.ell_w, uw = p.uell_w, n_ell = m * n * ew, n_un = p.n_upos;
    const int ncw = TNCW ? TNCW : p.ncw;  // state columns per chunk
    const int colsw = (2 + m) * ncw;   // operand columns per chunk
    const int CW = 16;                 // one 16-column operand tile per chunk (host guarantees colsw <= 16)
    const long long xd = (long long)n * d;
    const int tile = LD * n;

    double *Gb = lds;
    double *G2b = Gb + 2 * tile;
    double *wbuf = G2b + 2 * tile;  // per matrix wave
    const int wsz = LD * (CW + 3 * ncw);
    double *us = wbuf + 4 * wsz;
    double *t_unv = us + 3 * (m + 1) + 2;
    double *t_ung0 = t_unv + (p.tab_lds ? n_un * uw : 0);
    double *t_ellv = t_ung0 + (p.tab_lds ? n_un : 0);
    unsigned short *t_uni = reinterpret_cast<unsigned short *>(t_ellv + (p.tab_lds ? n_ell : 0));
    unsigned short *t_ellc = t_uni + (p.tab_lds ? n_un : 0);
    unsigned char *t_unl = reinterpret_cast<unsigned char *>(t_ellc + (p.tab_lds ? n_ell : 0));

    // Work split.  contig = 0: items (b, k, slice of nc columns) dealt round-robin to the workgroups.
    // contig = 1: the batch*K*d state columns of the launch are cut into gridDim.x equal contiguous ranges (to within one
    // column); a workgroup's items are the pieces of its range that lie in one interval (first and last piece partial),
    // so G, G^2 are built once per interval touched and every CU streams the same number of bytes.
    const long long blk = p.compact ? (long long)nn : (long long)d * nn;  // size of seg 0 / seg 1
    const int nc = p.nc;
    int n_my;
    long long g_lo = 0, g_hi = 0;
    // Role split (contig only, n_stream > 0): workgroups [0, n_stream) stream the B^{+-} blocks of ALL columns (their
    // matrix waves only build G, G^2), workgroups [n_stream, grid) do the column work of ALL columns (their stream waves
    // idle).  The store stream is memory-side bound and half the CUs sustain it; on a CU of its own it is not slowed
    // by the matrix waves' instructions and memory operations.
    // all_matrix (compact Jacobian: one copy of the blocks per interval, nothing to stream): every workgroup takes the
    // matrix role; the workgroup whose range holds an interval's column 0 also writes the interval's two unique blocks.
    const int bx = (int)blockIdx.x;  // dispatched round-robin over the 8 XCDs: both roles land on every XCD
    const bool stream_role = !p.all_matrix && p.n_stream > 0 && bx < p.n_stream;
    const bool matrix_role = p.all_matrix || (p.n_stream > 0 && !stream_role);
    auto alive = [&](int it) { return it < n_my; };
    if (p.contig) {
        const long long tot = (long long)p.batch * p.K * d;
        const long long widx = (matrix_role && !p.all_matrix) ? (long long)bx - p.n_stream : (long long)bx;
        const long long wcnt = (p.n_stream > 0 && !p.all_matrix) ? (stream_role ? (long long)p.n_stream : (long long)gridDim.x - p.n_stream)
                                                                   : (long long)gridDim.x;
        g_lo = tot * widx / wcnt;
        g_hi = tot * (widx + 1) / wcnt;
        n_my = g_hi > g_lo ? (int)((g_hi - 1) / d - g_lo / d) + 1 : 0;
    } else {
        const int n_items = p.batch * p.K * p.S;
        n_my = n_items > bx ? (n_items - bx + (int)gridDim.x - 1) / (int)gridDim.x : 0;
    }
    // item `it` of this workgroup: interval (b, k), state columns [c0, c0 + nce)
    auto decode = [&](int it, int &c0, int &nce, int &k, int &b) {
        if (p.contig) {
            const long long bk = g_lo / d + it;
            c0 = it == 0 ? (int)(g_lo - bk * d) : 0;
            nce = (int)min((long long)d, g_hi - bk * d) - c0;
            k = (int)(bk % p.K);
            b = (int)(bk / p.K);
        } else {
            const int item = bx + it * (int)gridDim.x;
            const int s = item % p.S;
            c0 = s * nc;
            nce = min(nc, d - c0);
            k = (item / p.S) % p.K;
            b = item / (p.S * p.K);
        }
    };

    // item 0's controls / time step: requested before the prologue's table loads so that the latencies overlap
    double u0 = 0.0;
    if (alive(0) && lane <= m) {  // (every wave: all eight take part in the first build)
        int c00, nce0, k0, b0;
        decode(0, c00, nce0, k0, b0);
        const double *z0 = p.Z + (long long)b0 * p.z_batch_stride + (long long)k0 * p.z_dim;
        u0 = z0[lane < m ? p.u_off + lane : p.dt_off];
    }
    unsigned short er_c[PCL_MREG][EW > 0 ? EW : 1];  // ELL rows (drive l, row = lane) of the matrix waves, in registers
    double er_v[PCL_MREG][EW > 0 ? EW : 1];
    // ---- prologue: both G buffers = drift tile, tables -> LDS ----------------------------------------------
    // Every loop issues ALL its global loads before the first LDS write: written as one load -> one store per iteration the
    // compiler waits for each load in turn (s_waitcnt vmcnt(0) per iteration), ~15 dependent round trips to L2 / HBM --
    // most of what an empty launch of this kernel used to cost.
    // ONE gather phase: every global load of the prologue is issued before the first LDS write.  The tables live in
    // separate small allocations and each dependent group of loads pays its own address-translation / L2 round trip at
    // kernel start (~3k cycles measured per group): issued together they cost one.  (Sizes: nn <= 4096, n_un * uw <= 2048
    // when staged, n_un <= 1024 -- checked by the host when it sets tab_lds.)
    {
        double vg[8], vu[4], g0v[2], ve[2];
        unsigned char lb[4];
        int ps[2], ce[2];
        const bool tabs = p.tab_lds && n_un * uw <= 2048 && n_un <= 1024 && n_ell <= 1024;
#pragma unroll
        for (int j = 0; j < 8; ++j) vg[j] = (!p.g0_batch_stride && tid + 512 * j < nn) ? p.G0[tid + 512 * j] : 0.0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bool ok = tabs && tid + 512 * j < n_un * uw;
            vu[j] = ok ? p.uell_v[tid + 512 * j] : 0.0;
            lb[j] = ok ? p.uell_l[tid + 512 * j] : 0;
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const bool ok = tabs && tid + 512 * j < n_un;
            ps[j] = ok ? p.upos[tid + 512 * j] : 0;
            g0v[j] = (ok && !p.g0_batch_stride) ? p.ug0[tid + 512 * j] : 0.0;  // (not G0[pos]: no dependent load in the prologue)
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const bool ok = tabs && tid + 512 * j < n_ell;
            ve[j] = ok ? p.ell_val[tid + 512 * j] : 0.0;
            ce[j] = ok ? p.ell_col[tid + 512 * j] : 0;
        }
        if (!p.g0_batch_stride) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int e = tid + 512 * j;
                if (e < nn) {
                    const int o = (e % n) + LD * (e / n);
                    Gb[o] = vg[j];
                    Gb[tile + o] = vg[j];
                }
            }
        }
        if (tabs) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (tid + 512 * j < n_un * uw) {
                    t_unv[tid + 512 * j] = vu[j];
                    t_unl[tid + 512 * j] = lb[j];
                }
#pragma unroll
            for (int j = 0; j < 2; ++j)
                if (tid + 512 * j < n_un) {
                    t_uni[tid + 512 * j] = (unsigned short)((ps[j] % n) + LD * (ps[j] / n));
                    t_ung0[tid + 512 * j] = g0v[j];
                }
#pragma unroll
            for (int j = 0; j < 2; ++j)
                if (tid + 512 * j < n_ell) {
                    t_ellv[tid + 512 * j] = ve[j];
                    t_ellc[tid + 512 * j] = (unsigned short)ce[j];
                }
        } else if (p.tab_lds) {  // larger tables: plain loops
            for (int e = tid; e < n_ell; e += 512) {
                t_ellv[e] = p.ell_val[e];
                t_ellc[e] = (unsigned short)p.ell_col[e];
            }
            for (int e = tid; e < n_un * uw; e += 512) {
                t_unv[e] = p.uell_v[e];
                t_unl[e] = p.uell_l[e];
            }
            for (int e = tid; e < n_un; e += 512) {
                const int pos = p.upos[e];
                t_uni[e] = (unsigned short)((pos % n) + LD * (pos / n));
                t_ung0[e] = p.g0_batch_stride ? 0.0 : p.ug0[e];
            }
        }
    }
    if (tid == 0) PCL_S(41);  // prologue loads written
    __syncthreads();
    if (tid == 0) PCL_S(42);  // prologue barrier passed

    // ---- first item's G(u), G^2 by ALL eight waves (the stream waves have nothing to write yet): the union pattern is
    //      split over the 512 threads, then one barrier, then one G^2 tile job per wave (4 row tiles x 2 column tiles at
    //      d = 27).  Every element is accumulated in the same order as in `build` below: identical bits. -----------------------
    int first_b = -1;
    if (alive(0)) {
        int c0_, nce_, k0_, b0_;
        decode(0, c0_, nce_, k0_, b0_);
        first_b = b0_;
        double *usn = us;  // slot of item 0
        if (lane <= m) usn[lane] = u0;  // every wave: identical values
        wave_lds_sync();
        if (tid == 0) PCL_S(49);  // controls arrived
        const double *G0b = p.G0 + (long long)b0_ * p.g0_batch_stride;
        if (p.g0_batch_stride) {
            for (int e = tid; e < nn; e += 512)
                if (p.umap[e] < 0) Gb[(e % n) + LD * (e / n)] = G0b[e];
            if (p.tab_lds)
                for (int q = tid; q < n_un; q += 512) t_ung0[q] = G0b[p.upos[q]];
            __syncthreads();
        }
        if (p.tab_lds) {
            for (int q = tid; q < n_un; q += 512) {
                double g = t_ung0[q];
                for (int w = 0; w < uw; ++w) g += usn[t_unl[q * uw + w]] * t_unv[q * uw + w];
                Gb[t_uni[q]] = g;
            }
        } else {
            for (int q = tid; q < n_un; q += 512) {
                const int pos = p.upos[q];
                double g = G0b[pos];
                const double *cf = p.ucoef + (long long)q * m;
                for (int l = 0; l < m; ++l) g += usn[l] * cf[l];
                Gb[(pos % n) + LD * (pos / n)] = g;
            }
        }
    }
    __syncthreads();
    if (tid == 0) PCL_S(50);  // G(u) complete
    if (alive(0)) {
        const int li = lane & 15, lk = lane >> 4;
        const int rt_n = (n + 15) >> 4, kfull = n >> 2, krem = n & 3;
        const int ct_n = p.iso ? (d + 15) >> 4 : rt_n, Nc = p.iso ? d : n;
        const double *G = Gb;
        double *G2 = G2b;
        for (int job = wave; job < rt_n * ct_n; job += 8) {
            const int rt = job % rt_n, ct = job / rt_n;
            const double *Ap = G + rt * 16 + li + LD * lk;
            const double *Bp = G + lk + LD * (ct * 16 + li);
            // k-step ks into accumulator ks mod 4, G^2 = ((s0 + remainder step) + s2) + (s1 + s3): the order of `build` below
            double4_t qa[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) qa[j] = double4_t{0.0, 0.0, 0.0, 0.0};
            double a[16], bb[16];  // every operand of the tile job is requested before the first MFMA (kfull <= 16)
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const bool ok = j < kfull;
                a[j] = ok ? Ap[LD * 4 * j] : 0.0;
                bb[j] = ok ? Bp[4 * j] : 0.0;
            }
#pragma unroll
            for (int j = 0; j < 16; ++j)
                if (j < kfull) qa[j & 3] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[j], bb[j], qa[j & 3], 0, 0, 0);
            if (krem) {
                const bool ok = lk < krem;
                qa[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(ok ? Ap[LD * 4 * kfull] : 0.0, ok ? Bp[4 * kfull] : 0.0, qa[0], 0, 0, 0);
            }
            const double4_t acc = (qa[0] + qa[2]) + (qa[1] + qa[3]);

            const int col = ct * 16 + li;
            if (col < Nc) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = rt * 16 + lk + 4 * r;
                    if (row < n) {
                        G2[row + LD * col] = acc[r];
                        if (p.iso) {
                            if (row < d)
                                G2[row + d + LD * (col + d)] = acc[r];
                            else
                                G2[row - d + LD * (col + d)] = -acc[r];
                        }
                    }
                }
            }
        }
    }
    if (tid == 0) PCL_S(44);  // first G, G^2 built
    __syncthreads();  // item 0's G, G^2 complete
    if (tid == 0) PCL_S(45);

    if (wave < 4 || matrix_role) {
        // ======================================= matrix waves =========================================
        // matrix role: all eight waves work on chunks; G, G^2 are single-buffered there and the second halves of the
        // double buffers hold the chunk buffers of waves 4..7 (the host checks 2*wsz <= tile)
        const int nmw = matrix_role ? 8 : 4;
        double *Mw = wave < 4 ? wbuf + wave * wsz : (wave < 6 ? Gb + tile + (wave - 4) * wsz : G2b + tile + (wave - 6) * wsz);  // [LD*CW]: S | D | G_l D
        double *GDw = Mw + LD * CW;      // [LD*ncw]
        double *G2Dw = GDw + LD * ncw;   // [LD*ncw]
        double *GSw = G2Dw + LD * ncw;   // [LD*ncw]
        const int li = lane & 15, lk = lane >> 4;
        const int rt_n = (n + 15) >> 4;
        const int kfull = n >> 2, krem = n & 3;
        // per-member drift: which member's drift the two G buffers (off the union pattern) and the union table t_ung0 hold.
        // A workgroup's consecutive items almost always belong to the same member (contiguous column ranges), so the
        // drift tile is rewritten -- from memory, by every building wave -- only when the member changes.
        int drift_in_buf0 = first_b, drift_in_buf1 = -1, drift_in_tab = first_b;  // (the first build above loaded buffer 0 and the table)
        // G(u) on the union pattern + this wave's share of the G^2 tiles, for item `it`, into buffer `buf`
        auto build = [&](int it, int buf, double u_lane) {
            int c0_, nce_, k, b;
            decode(it, c0_, nce_, k, b);
            double *G = Gb + buf * tile, *G2 = G2b + buf * tile;
            double *usn = us + (it % 3) * (m + 1);
            if (lane <= m) usn[lane] = u_lane;  // every wave: identical values
            wave_lds_sync();
            const double *G0b = p.G0 + (long long)b * p.g0_batch_stride;
            if (p.g0_batch_stride) {
                // per-member drift: the whole tile changes with b.  Union positions are skipped: every position only ever
                // receives its final value, so the four building waves (which write identical data and then read all of G
                // for their G^2 tiles without a workgroup barrier) cannot observe each other's intermediate state
                int &have = buf ? drift_in_buf1 : drift_in_buf0;
                if (have != b) {  // eight independent loads in flight per lane (the loop is latency-, not bandwidth-bound)
                    for (int e0 = lane; e0 < nn; e0 += 64 * 8) {
                        double v[8];
                        int um[8];
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            const int e = e0 + 64 * j;
                            um[j] = e < nn ? p.umap[e] : 0;
                            v[j] = e < nn ? G0b[e] : 0.0;
                        }
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            const int e = e0 + 64 * j;
                            if (um[j] < 0) G[(e % n) + LD * (e / n)] = v[j];
                        }
                    }
                    have = b;
                }
                if (p.tab_lds && drift_in_tab != b) {  // (every building wave writes the same values)
                    for (int q0 = lane; q0 < n_un; q0 += 64 * 4) {
                        int ps[4];
#pragma unroll
                        for (int j = 0; j < 4; ++j) ps[j] = q0 + 64 * j < n_un ? p.upos[q0 + 64 * j] : 0;
                        double v[4];
#pragma unroll
                        for (int j = 0; j < 4; ++j) v[j] = G0b[ps[j]];
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            if (q0 + 64 * j < n_un) t_ung0[q0 + 64 * j] = v[j];
                    }
                    drift_in_tab = b;
                    wave_lds_sync();
                }
            }
            if (p.tab_lds && uw <= 2) {
                // four pattern entries per lane and pass: every table read of the pass is issued before the first use (the
                // one-entry-at-a-time form costs three dependent LDS round trips per entry)
                for (int q0 = lane; q0 < n_un; q0 += 64 * 4) {
                    double g0v[4], cv[4][2];
                    int dl[4][2], oi[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int q = min(q0 + 64 * j, n_un - 1);
                        g0v[j] = t_ung0[q];
                        oi[j] = t_uni[q];
#pragma unroll
                        for (int w = 0; w < 2; ++w) {
                            const bool ok = w < uw;
                            dl[j][w] = ok ? t_unl[q * uw + w] : 0;
                            cv[j][w] = ok ? t_unv[q * uw + w] : 0.0;
                        }
                    }
                    double uv[4][2];
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int w = 0; w < 2; ++w) uv[j][w] = usn[dl[j][w]];
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (q0 + 64 * j < n_un) {
                            double g = g0v[j];
#pragma unroll
                            for (int w = 0; w < 2; ++w)
                                if (w < uw) g += uv[j][w] * cv[j][w];
                            G[oi[j]] = g;
                        }
                }
            } else if (p.tab_lds) {
                for (int q = lane; q < n_un; q += 64) {
                    double g = t_ung0[q];
                    for (int w = 0; w < uw; ++w) g += usn[t_unl[q * uw + w]] * t_unv[q * uw + w];
                    G[t_uni[q]] = g;
                }
            } else {
                for (int q = lane; q < n_un; q += 64) {
                    const int pos = p.upos[q];
                    double g = G0b[pos];
                    const double *cf = p.ucoef + (long long)q * m;
                    for (int l = 0; l < m; ++l) g += usn[l] * cf[l];
                    G[(pos % n) + LD * (pos / n)] = g;
                }
            }
            wave_lds_sync();
            // G^2: row tile rt = wave (+4..), all column tiles; with the iso structure only the first d columns
            const int ct_n = p.iso ? (d + 15) >> 4 : rt_n;
            const int Nc = p.iso ? d : n;
            for (int rt = wave; rt < rt_n; rt += 4) {
                const double *Ap = G + rt * 16 + li + LD * lk;
                for (int ct = 0; ct < ct_n; ct += 2) {
                    const bool two = ct + 1 < ct_n;
                    const double *Bp0 = G + lk + LD * (ct * 16 + li);
                    const double *Bp1 = Bp0 + (two ? LD * 16 : 0);
                    // Four accumulators per tile, k-step ks into accumulator ks mod 4 (a dependent f64 MFMA waits ~235 cycles
                    // for its accumulator; four independent chains keep the pipe issuing):
                    //     G^2 = ((s0 + remainder step) + s2) + (s1 + s3)   -- the first build above adds in exactly this order.
                    double4_t q0[4], q1[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) q0[j] = q1[j] = double4_t{0.0, 0.0, 0.0, 0.0};
                    for (int ks0 = 0; ks0 < kfull; ks0 += 4) {
                        double a[4], b0[4], b1[4];
#pragma unroll
                        for (int j = 0; j < 4; ++j) {  // the four steps' operands are requested together
                            const bool ok = ks0 + j < kfull;
                            a[j] = ok ? Ap[LD * 4 * (ks0 + j)] : 0.0;
                            b0[j] = ok ? Bp0[4 * (ks0 + j)] : 0.0;
                            b1[j] = ok ? Bp1[4 * (ks0 + j)] : 0.0;
                        }
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            if (ks0 + j < kfull) {
                                q0[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[j], b0[j], q0[j], 0, 0, 0);
                                if (two) q1[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[j], b1[j], q1[j], 0, 0, 0);
                            }
                    }
                    if (krem) {
                        const bool ok = lk < krem;
                        const double a = ok ? Ap[LD * 4 * kfull] : 0.0;
                        const double b0 = ok ? Bp0[4 * kfull] : 0.0;
                        const double b1 = ok ? Bp1[4 * kfull] : 0.0;
                        q0[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b0, q0[0], 0, 0, 0);
                        if (two) q1[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b1, q1[0], 0, 0, 0);
                    }
                    const double4_t acc0 = (q0[0] + q0[2]) + (q0[1] + q0[3]), acc1 = (q1[0] + q1[2]) + (q1[1] + q1[3]);

#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        const int col = (ct + t) * 16 + li;
                        if ((t == 0 || two) && col < Nc) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                const int row = rt * 16 + lk + 4 * r;
                                const double v = t ? acc1[r] : acc0[r];
                                if (row < n) {
                                    G2[row + LD * col] = v;
                                    if (p.iso) {
                                        if (row < d)
                                            G2[row + d + LD * (col + d)] = v;
                                        else
                                            G2[row - d + LD * (col + d)] = -v;
                                    }
                                }
                            }
                        }
                    }
                }
            }
        };

        // the ELL rows are needed by the column work only: requested now, while the stream waves already write
        if (EW > 0) {
#pragma unroll
            for (int l = 0; l < PCL_MREG; ++l)
#pragma unroll
                for (int q = 0; q < (EW > 0 ? EW : 1); ++q) {
                    er_c[l][q] = 0;
                    er_v[l][q] = 0.0;
                    if (l < m && lane < n) {
                        const int ie = (l * n + lane) * ew + q;
                        er_c[l][q] = p.tab_lds ? t_ellc[ie] : (unsigned short)p.ell_col[ie];
                        er_v[l][q] = p.tab_lds ? t_ellv[ie] : p.ell_val[ie];
                    }
                }
        }

        for (int it = 0; alive(it); ++it) {
            const int cur = matrix_role ? 0 : (it & 1);
            int c0, nce, k, b;
            decode(it, c0, nce, k, b);
            const double *zk = p.Z + (long long)b * p.z_batch_stride + (long long)k * p.z_dim;
            const double *zn = zk + p.z_dim;
            const int x_off = p.x_offs[p.z_batch_stride ? 0 : b];
            const double *G = Gb + cur * tile, *G2 = G2b + cur * tile;
            const double h = us[(it % 3) * (m + 1) + m];
            const double c1 = 0.5 * h, c2 = h * h * (1.0 / 12.0), h6 = h * (1.0 / 6.0);
            const long long bk = (long long)b * p.K + k;
            double *jb = p.jac + bk * p.jac_per;
            double *jt = jb + 2 * blk;  // tail: for column c: [d/du_0 .. d/du_{m-1} | d/ddt], n doubles each

            const int nchunk = (nce + ncw - 1) / ncw;
#ifdef PCL_PROFILE
            int stamp = 0;
#define PCL_STAMP()                                                                                     \
    do {                                                                                                \
        if (p.dbg && blockIdx.x == 0 && wave == 0 && lane == 0 && it == 1 && stamp < 60) p.dbg[stamp++] = (long long)__builtin_amdgcn_s_memtime(); \
    } while (0)
#else
#define PCL_STAMP() do { } while (0)
#endif
            PCL_STAMP();
            // All global reads of this item are issued here, before the wave has any of the item's stores in flight:
            // a later load would sit behind them in the CU's saturated memory pipeline (and vmcnt is in-order).
            const bool pf = ncw <= PCL_PFW;
            double pxn[PCL_PFC][PCL_PFW], pxc[PCL_PFC][PCL_PFW];
            if (pf && lane < n && !stream_role) {
#pragma unroll
                for (int t = 0; t < PCL_PFC; ++t)
#pragma unroll
                    for (int c = 0; c < PCL_PFW; ++c) {
                        const int col = c0 + (wave + nmw * t) * ncw + c;
                        pxn[t][c] = pxc[t][c] = 0.0;
                        if (c < ncw && col < c0 + nce) {
                            const long long o = x_off + (long long)col * n + lane;
                            pxn[t][c] = zn[o];
                            pxc[t][c] = zk[o];
                        }
                    }
            }
            double pf_u = 0.0;  // next item's u_k / dt_k (consumed by build)
            if (alive(it + 1) && lane <= m && wave < 4) {
                int c02, nce2, k2, b2;
                decode(it + 1, c02, nce2, k2, b2);
                const double *zk2 = p.Z + (long long)b2 * p.z_batch_stride + (long long)k2 * p.z_dim;
                pf_u = zk2[lane < m ? p.u_off + lane : p.dt_off];
            }
            if (p.all_matrix && p.compact && c0 == 0) {  // the interval's unique -B^+ / B^- blocks
                const int half = nn >> 1;
                for (int q = tid; q < half; q += 512) {
                    const int pos = 2 * q, i = pos % n, j = pos / n;
                    const double g0 = G[i + LD * j], g1 = G[i + 1 + LD * j];
                    const double h0 = G2[i + LD * j], h1 = G2[i + 1 + LD * j];
                    double bp0, bp1, bm0, bm1;
                    bpm_entry((i == j) ? 1.0 : 0.0, c1, c2, g0, h0, bp0, bm0);
                    bpm_entry((i + 1 == j) ? 1.0 : 0.0, c1, c2, g1, h1, bp1, bm1);
                    store2(jb + pos, bp0, bp1, p.nt);
                    store2(jb + blk + pos, bm0, bm1, p.nt);
                }
            }
            int tch = 0;
            for (int ch = wave; ch < nchunk && !stream_role; ch += nmw, ++tch) {
                const int cc0 = c0 + ch * ncw;             // first state column of the chunk
                const int ncc = min(ncw, c0 + nce - cc0);  // columns in this chunk
                const bool pf_lam = MERIT && p.mpart && p.mlam && ncw <= PCL_PFW;  // multipliers of the chunk: requested now, used after the outputs are formed
                double lamr[PCL_PFW];
#pragma unroll
                for (int c = 0; c < PCL_PFW; ++c) lamr[c] = (pf_lam && c < ncc && lane < n) ? p.mlam[bk * xd + (long long)(cc0 + c) * n + lane] : 0.0;
                // ---- M = [S | D | G_l D]   (lane = row) -----------------------------------------------------
                if (lane < n) {
                    if (pf && tch < PCL_PFC) {
#pragma unroll
                        for (int t = 0; t < PCL_PFC; ++t)
                            if (t == tch) {
#pragma unroll
                                for (int c = 0; c < PCL_PFW; ++c)
                                    if (c < ncw) {
                                        Mw[lane + LD * c] = pxn[t][c] + pxc[t][c];
                                        Mw[lane + LD * (ncw + c)] = pxn[t][c] - pxc[t][c];
                                    }
                            }
                    } else {
                        for (int c = 0; c < ncw; ++c) {
                            double xs = 0.0, xdv = 0.0;
                            if (c < ncc) {
                                const long long o = x_off + (long long)(cc0 + c) * n + lane;
                                const double xn = zn[o], xc = zk[o];
                                xs = xn + xc;
                                xdv = xn - xc;
                            }
                            Mw[lane + LD * c] = xs;
                            Mw[lane + LD * (ncw + c)] = xdv;
                        }
                    }
                    for (int c = colsw; c < CW; ++c) Mw[lane + LD * c] = 0.0;
                }
                wave_lds_sync();
                PCL_STAMP();  // S, D loaded
                const double *Dm = Mw + LD * ncw;
                if (lane < n) {
                    if (EW > 0 && TNCW > 0) {
                        // Gathers and results live in the same LDS array: written as load -> multiply -> store per (l, c) the
                        // compiler must wait for every store before the next gather (it cannot prove they do not alias) -- one
                        // LDS round trip per entry.  Three drives' gathers are requested together, then their results stored.
                        constexpr int NCW = TNCW > 0 ? TNCW : 1, EWn = EW > 0 ? EW : 1;
#pragma unroll
                        for (int l0 = 0; l0 < PCL_MREG; l0 += 3) {
                            double dv[3][NCW][EWn];
#pragma unroll
                            for (int j = 0; j < 3; ++j)
#pragma unroll
                                for (int c = 0; c < NCW; ++c)
#pragma unroll
                                    for (int q = 0; q < EWn; ++q) dv[j][c][q] = (l0 + j < m) ? Dm[er_c[(l0 + j) % PCL_MREG][q] + LD * c] : 0.0;
#pragma unroll
                            for (int j = 0; j < 3; ++j)
                                if (l0 + j < m) {
                                    const int l = (l0 + j) % PCL_MREG;
#pragma unroll
                                    for (int c = 0; c < NCW; ++c) {
                                        double acc = 0.0;
#pragma unroll
                                        for (int q = 0; q < EWn; ++q) acc = __builtin_fma(er_v[l][q], dv[j][c][q], acc);
                                        Mw[lane + LD * (2 * ncw + l * ncw + c)] = acc;
                                    }
                                }
                        }
                    } else if (EW > 0) {
#pragma unroll
                        for (int l = 0; l < PCL_MREG; ++l)
                            if (l < m)
                                for (int c = 0; c < ncw; ++c) {
                                    double acc = 0.0;
#pragma unroll
                                    for (int q = 0; q < (EW > 0 ? EW : 1); ++q) acc += er_v[l][q] * Dm[er_c[l][q] + LD * c];
                                    Mw[lane + LD * (2 * ncw + l * ncw + c)] = acc;
                                }
                    } else {
                        for (int l = 0; l < m; ++l)
                            for (int c = 0; c < ncw; ++c) {
                                const int base = (l * n + lane) * ew;
                                double acc = 0.0;
                                for (int q = 0; q < ew; ++q) {
                                    const int col = p.tab_lds ? (int)t_ellc[base + q] : p.ell_col[base + q];
                                    const double ev = p.tab_lds ? t_ellv[base + q] : p.ell_val[base + q];
                                    acc += ev * Dm[col + LD * c];
                                }
                                Mw[lane + LD * (2 * ncw + l * ncw + c)] = acc;
                            }
                    }
                    PCL_STAMP();  // G_l D done
                    // G2D = G^2 D on the VALU, 6 independent partial sums per column
                    if (TNCW > 0) {
                        // all columns of the chunk in one sweep over G^2's row: every G^2 entry is read once (not once per column) and
                        // the operand column entries as neighbouring pairs; per column the same partial sums in the same order
                        constexpr int NCW = TNCW > 0 ? TNCW : 1;
                        double sa[6][NCW];
#pragma unroll
                        for (int j = 0; j < 6; ++j)
#pragma unroll
                            for (int c = 0; c < NCW; ++c) sa[j][c] = 0.0;
                        int kk = 0;
#pragma unroll 3
                        for (; kk + 6 <= n; kk += 6) {
#pragma unroll
                            for (int j = 0; j < 6; ++j) {
                                const double g = G2[lane + LD * (kk + j)];
#pragma unroll
                                for (int c = 0; c < NCW; ++c) sa[j][c] = fma(g, Dm[kk + j + LD * c], sa[j][c]);
                            }
                        }
                        for (; kk < n; ++kk) {
                            const double g = G2[lane + LD * kk];
#pragma unroll
                            for (int c = 0; c < NCW; ++c) sa[0][c] = fma(g, Dm[kk + LD * c], sa[0][c]);
                        }
#pragma unroll
                        for (int c = 0; c < NCW; ++c) G2Dw[lane + LD * c] = ((sa[0][c] + sa[1][c]) + (sa[2][c] + sa[3][c])) + (sa[4][c] + sa[5][c]);
                    } else
                    for (int c = 0; c < ncw; ++c) {
                        double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0, s4 = 0.0, s5 = 0.0;
                        int kk = 0;
                        for (; kk + 6 <= n; kk += 6) {
                            s0 = fma(G2[lane + LD * kk], Dm[kk + LD * c], s0);
                            s1 = fma(G2[lane + LD * (kk + 1)], Dm[kk + 1 + LD * c], s1);
                            s2 = fma(G2[lane + LD * (kk + 2)], Dm[kk + 2 + LD * c], s2);
                            s3 = fma(G2[lane + LD * (kk + 3)], Dm[kk + 3 + LD * c], s3);
                            s4 = fma(G2[lane + LD * (kk + 4)], Dm[kk + 4 + LD * c], s4);
                            s5 = fma(G2[lane + LD * (kk + 5)], Dm[kk + 5 + LD * c], s5);
                        }
                        for (; kk < n; ++kk) s0 = fma(G2[lane + LD * kk], Dm[kk + LD * c], s0);
                        G2Dw[lane + LD * c] = ((s0 + s1) + (s2 + s3)) + (s4 + s5);
                    }
                }
                wave_lds_sync();
                PCL_STAMP();  // G2D done
                // ---- W = G * M on the matrix cores: all row tiles at once (they share the b operand) ----------
                {
                    const double *Bp = Mw + lk + LD * li;
                    const double *Ap[PCL_MAXRT];
                    bool rok[PCL_MAXRT];
#pragma unroll
                    for (int t = 0; t < PCL_MAXRT; ++t) {
                        rok[t] = t * 16 < n;
                        Ap[t] = G + (rok[t] ? t * 16 : 0) + li + LD * lk;
                    }
                    double4_t acc[PCL_MAXRT];
#pragma unroll
                    for (int t = 0; t < PCL_MAXRT; ++t) acc[t] = double4_t{0.0, 0.0, 0.0, 0.0};
                    {
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
                                for (int t = 0; t < PCL_MAXRT; ++t) an[t] = Ap[t][LD * 4 * (ks + 1)];
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
                                    const double a = ok ? Ap[t][LD * 4 * kfull] : 0.0;
                                    acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bb, acc[t], 0, 0, 0);
                                }
                        }
                    }
                    // accumulator layout -> LDS: column li of [G S | G D | G (G_l D)]; the last group goes back into M's
                    // own columns (their operand role is over)
                    double *dst = li < ncw ? GSw + LD * li : (li < 2 * ncw ? GDw + LD * (li - ncw) : Mw + LD * li);
#pragma unroll
                    for (int t = 0; t < PCL_MAXRT; ++t)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int row = t * 16 + lk + 4 * r;
                            if (row < n && li < colsw) dst[row] = acc[t][r];
                        }
                }
                wave_lds_sync();
                PCL_STAMP();  // MFMA + accumulators -> LDS done
                // ---- outputs: finish in LDS (lane = row, in place), then 16-byte stores --------------------------------
                // in place: delta -> D column, d/ddt -> GS column, d/du_l -> the G (G_l D) column
                if (EW > 0 && TNCW > 0) {
                    // Same values as the general form below, with every LDS read of a step requested before its first store (see
                    // the G_l D step above): Y = -c1 S + c2 G D takes the place of S, then d/du_l = G_l Y + c2 G (G_l D).
                    constexpr int NCW = TNCW > 0 ? TNCW : 1, EWn = EW > 0 ? EW : 1;
                    if (lane < n) {
                        double gs[NCW], g2d[NCW], dd[NCW], ss[NCW], gd[NCW];
#pragma unroll
                        for (int c = 0; c < NCW; ++c) {
                            gs[c] = GSw[lane + LD * c];
                            g2d[c] = G2Dw[lane + LD * c];
                            dd[c] = Mw[lane + LD * (ncw + c)];
                            ss[c] = Mw[lane + LD * c];
                            gd[c] = GDw[lane + LD * c];
                        }
#pragma unroll
                        for (int c = 0; c < NCW; ++c) {
                            G2Dw[lane + LD * c] = dd[c] - c1 * gs[c] + c2 * g2d[c];  // delta
                            GSw[lane + LD * c] = -0.5 * gs[c] + h6 * g2d[c];        // d/ddt
                            Mw[lane + LD * c] = __builtin_fma(c2, gd[c], -(c1 * ss[c]));  // Y (S is dead from here on)
                        }
                    }
                    wave_lds_sync();
                    if (lane < n) {
#pragma unroll
                        for (int l0 = 0; l0 < PCL_MREG; l0 += 3) {
                            double yv[3][NCW][EWn], gv[3][NCW];
#pragma unroll
                            for (int j = 0; j < 3; ++j)
#pragma unroll
                                for (int c = 0; c < NCW; ++c) {
                                    const int l = (l0 + j) % PCL_MREG;
#pragma unroll
                                    for (int q = 0; q < EWn; ++q) yv[j][c][q] = (l0 + j < m) ? Mw[er_c[l][q] + LD * c] : 0.0;
                                    gv[j][c] = (l0 + j < m) ? Mw[lane + LD * (2 * ncw + l * ncw + c)] : 0.0;
                                }
#pragma unroll
                            for (int j = 0; j < 3; ++j)
                                if (l0 + j < m) {
                                    const int l = (l0 + j) % PCL_MREG;
#pragma unroll
                                    for (int c = 0; c < NCW; ++c) {
                                        double acc = 0.0;
#pragma unroll
                                        for (int q = 0; q < EWn; ++q) acc = __builtin_fma(er_v[l][q], yv[j][c][q], acc);
                                        Mw[lane + LD * (2 * ncw + l * ncw + c)] = __builtin_fma(c2, gv[j][c], acc);
                                    }
                                }
                        }
                    }
                } else if (lane < n) {
                    for (int c = 0; c < ncw; ++c) {
                        const double gs = GSw[lane + LD * c], g2d = G2Dw[lane + LD * c];
                        G2Dw[lane + LD * c] = Mw[lane + LD * (ncw + c)] - c1 * gs + c2 * g2d;  // delta
                        GSw[lane + LD * c] = -0.5 * gs + h6 * g2d;                                // d/ddt
                    }
                    // d/du_l = G_l (-c1 S + c2 G D) + c2 G (G_l D)
                    if (EW > 0) {
#pragma unroll
                        for (int l = 0; l < PCL_MREG; ++l)
                            if (l < m)
                                for (int c = 0; c < ncw; ++c) {
                                    double acc = 0.0;
#pragma unroll
                                    for (int q = 0; q < (EW > 0 ? EW : 1); ++q)
                                        acc += er_v[l][q] * (-c1 * Mw[er_c[l][q] + LD * c] + c2 * GDw[er_c[l][q] + LD * c]);
                                    Mw[lane + LD * (2 * ncw + l * ncw + c)] = acc + c2 * Mw[lane + LD * (2 * ncw + l * ncw + c)];
                                }
                    } else {
                        for (int l = 0; l < m; ++l)
                            for (int c = 0; c < ncw; ++c) {
                                const int base = (l * n + lane) * ew;
                                double acc = 0.0;
                                for (int q = 0; q < ew; ++q) {
                                    const int col = p.tab_lds ? (int)t_ellc[base + q] : p.ell_col[base + q];
                                    const double ev = p.tab_lds ? t_ellv[base + q] : p.ell_val[base + q];
                                    acc += ev * (-c1 * Mw[col + LD * c] + c2 * GDw[col + LD * c]);
                                }
                                Mw[lane + LD * (2 * ncw + l * ncw + c)] = acc + c2 * Mw[lane + LD * (2 * ncw + l * ncw + c)];
                            }
                    }
                }
                wave_lds_sync();
                // ---- optional: the chunk's share of the reduce payload (pcl_eval_jac_merit_dev), while the vectors are still in
                //      LDS: per column <d delta/d u_l, lam>, <d delta/d dt, lam>, <delta, lam>.  Lane (s = lane & 15, q = lane >> 4)
                //      adds the rows q, q + 4, .. of product s = c*(m+2) + l in row order, then the four parts in a fixed tree:
                //      bitwise repeatable, and the 74 MB of tails are never read back from HBM.
                if (MERIT && p.mpart) {
                    if (p.mlam) {  // the multipliers take the place of S (dead since the epilogue above)
                        if (lane < n) {
                            if (pf_lam) {
#pragma unroll
                                for (int c = 0; c < PCL_PFW; ++c)
                                    if (c < ncw) Mw[lane + LD * c] = lamr[c];
                            } else {
                                for (int c = 0; c < ncw; ++c) Mw[lane + LD * c] = c < ncc ? p.mlam[bk * xd + (long long)(cc0 + c) * n + lane] : 0.0;
                            }
                        }
                        wave_lds_sync();
                    }
                    const int s = lane & 15, q = lane >> 4, c = s / (m + 2), l = s - c * (m + 2);
                    double acc = 0.0;
                    {
                        // every operand of a batch is requested before its first use (a rolled loop is one LDS round trip per row)
                        const bool on = s < (m + 2) * ncw && c < ncc;
                        const double *av = !on ? Mw : (l < m ? Mw + LD * (2 * ncw + l * ncw + c) : (l == m ? GSw + LD * c : G2Dw + LD * c));
                        const double *lv = !on ? Mw : (p.mlam ? Mw + LD * c : G2Dw + LD * c);
                        constexpr int JN = TD ? (2 * TD + 3) / 4 : 16, JB = (JN + 1) / 2;
#pragma unroll
                        for (int j0 = 0; j0 < JN; j0 += JB) {
                            double a[JB], bq[JB];
#pragma unroll
                            for (int j = 0; j < JB; ++j) {
                                const int r = q + 4 * (j0 + j);
                                const bool ok = on && j0 + j < JN && r < n;
                                a[j] = ok ? av[r] : 0.0;
                                bq[j] = ok ? lv[r] : 0.0;
                            }
#pragma unroll
                            for (int j = 0; j < JB; ++j) acc = fma(a[j], bq[j], acc);
                        }
                    }
                    acc += __shfl_xor(acc, 16, 64);
                    acc += __shfl_xor(acc, 32, 64);
                    if (q == 0 && s < (m + 2) * ncw && c < ncc)
                        p.mpart[(bk * d + cc0 + c) * (m + 2) + l] = (l == m + 1 && !p.mlam) ? 0.5 * acc : acc;
                }
                // the chunk's columns are contiguous in every output vector: element e = c*n + row, two per lane
                {
                    const int hn2 = n >> 1;
                    const long long o0 = (long long)cc0 * n;
                    for (int e2 = lane; e2 < ncc * hn2; e2 += 64) {
                        const int c = e2 / hn2, r0 = 2 * (e2 - c * hn2);
                        if (p.delta) store2(p.delta + bk * xd + o0 + (long long)c * n + r0, G2Dw[r0 + LD * c], G2Dw[r0 + 1 + LD * c], false);
                        double *tc = jt + (long long)(cc0 + c) * (m + 1) * n + r0;  // this column's (m+1)*n tail block
                        for (int l = 0; l < m; ++l) {
                            const double *src = Mw + LD * (2 * ncw + l * ncw + c) + r0;
                            store2(tc + (long long)l * n, src[0], src[1], false);
                        }
                        store2(tc + (long long)m * n, GSw[r0 + LD * c], GSw[r0 + 1 + LD * c], false);
                    }
                }
                wave_lds_sync();  // the chunk buffers are rewritten by this wave's next chunk
                PCL_STAMP();  // outputs issued
            }
#ifdef PCL_PROFILE
            if (p.dbg && blockIdx.x == 0 && lane == 0 && it == 1) p.dbg[52 + wave] = (long long)__builtin_amdgcn_s_memtime();  // this wave's chunks done
#endif
            // ---- next item's G(u), G^2 into the other buffer ----------------------------------------------------
            if (matrix_role) __syncthreads();  // single-buffered G, G^2: every wave is done with this item's tiles
            if (alive(it + 1) && wave < 4) build(it + 1, matrix_role ? 0 : cur ^ 1, pf_u);
            PCL_STAMP();  // next G, G^2 built
            __syncthreads();  // item boundary
            PCL_STAMP();  // barrier passed
        }
    } else {
        // ======================================= stream waves =========================================
        const int stid = tid - 256;
        const int hn = n >> 1;
        const int pi = 2 * (stid % hn), pj0 = stid / hn, pstep = max(256 / hn, 1);
        const bool pact = pj0 < pstep;
        for (int it = 0; alive(it); ++it) {
            const int cur = it & 1;
            int c0, nce, k, b;
            decode(it, c0, nce, k, b);
            const double *G = Gb + cur * tile, *G2 = G2b + cur * tile;
            if (pact && !matrix_role) {
                const double h = us[(it % 3) * (m + 1) + m];
                const double c1 = 0.5 * h, c2 = h * h * (1.0 / 12.0);
                double bpr[PCL_NSP][2], bmr[PCL_NSP][2];
#pragma unroll
                for (int r = 0; r < PCL_NSP; ++r) {
                    const int j = pj0 + pstep * r;
                    if (j < n) {
                        const double g0 = G[pi + LD * j], g1 = G[pi + 1 + LD * j];
                        const double h0 = G2[pi + LD * j], h1 = G2[pi + 1 + LD * j];
                        bpm_entry((pi == j) ? 1.0 : 0.0, c1, c2, g0, h0, bpr[r][0], bmr[r][0]);
                        bpm_entry((pi + 1 == j) ? 1.0 : 0.0, c1, c2, g1, h1, bpr[r][1], bmr[r][1]);
                    }
                }
                int cbeg = c0, cend = c0 + nce;
                if (p.compact) {  // unique blocks only: slice 0 writes the single copy
                    cbeg = 0;
                    cend = (c0 == 0) ? 1 : 0;
                }
                double *o = p.jac + ((long long)b * p.K + k) * p.jac_per + (long long)cbeg * nn + pi;
                for (int c = cbeg; c < cend; ++c, o += nn) {
#pragma unroll
                    for (int r = 0; r < PCL_NSP; ++r) {
                        const int j = pj0 + pstep * r;
                        if (j < n) {
                            store2(o + n * j, bpr[r][0], bpr[r][1], p.nt);
                            store2(o + blk + n * j, bmr[r][0], bmr[r][1], p.nt);
                        }
                    }
                }
            }
            if (tid == 256 && it == 0) {
                PCL_S(46);  // first item's stores issued
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                PCL_S(47);  // ... and gone
            }
            __syncthreads();  // item boundary
        }
    }
    if (tid == 0) PCL_S(48);  // matrix wave 0 done
#ifdef PCL_PROFILE
    if (p.dbg && tid == 256 && blockIdx.x < PCL_DBG_WG) {  // wave 4: a stream wave, or a matrix wave of a matrix-role workgroup
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // its stores have left the CU
        p.dbg[64 + PCL_DBG_WG + blockIdx.x] = (long long)__builtin_amdgcn_s_memrealtime();
    }
#endif
}
