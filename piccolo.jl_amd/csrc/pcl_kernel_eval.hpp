// pcl_kernel_eval.hpp -- residual-only kernel (pcl_eval[_dev]): what Ipopt calls in every line-search trial.
#pragma once

// ------------------------------------------------------------------------------------------
// delta_k = D - (h/2) G S + (h^2/12) G (G D),  S = X_{k+1} + X_k, D = X_{k+1} - X_k      (no G^2, no Jacobian)
// Output is 1.2 MB per evaluation: the kernel is bound by its three n x n x d products (0.47 Mflop per interval on the f64
// matrix cores) and by latency, not by HBM.  Persistent workgroups of 4 wavefronts, one interval per item, three
// barriers per item:
//   phase 0  all    : G(u_k) on the drives' union pattern (the drift stays in the LDS tile; per-thread pattern entries in
//                     registers), the wave's chunk of state columns -> its operand tile [S | D] (inputs were requested
//                     one item ahead)
//   phase 1  wave w : chunk w (cw <= 8 columns): [G S | G D] in ONE pass (four row tiles share the b operand); the
//                     operand tile is overwritten in place with P = D - (h/2) G S  and  G D
//   phase 2  wave w : row tile w of G (G D) for ALL columns (b operand gathered from the four chunk tiles);
//                     delta = P + (h^2/12) G (G D), in place
//   phase 3  wave w : its chunk's columns of delta are contiguous in memory: 16-byte stores
// Matrix-core work per interval: 4 x 14 x 4 + 4 x 14 x 2 = 336 MFMAs at d = 27 (the minimum for 16-wide tiles).
// LDS map (doubles): G^T [LD*n] (G stored row-contiguous: G[i][k] at k + LD*i, the conflict-free layout of the MFMA a
// operand -- the column-major layout costs a 2-way bank conflict on 6-10 of every 16 lanes) | per wave M [LD*16] | us [2][m+1]
// ------------------------------------------------------------------------------------------
#define PCL_NUE_EV 4  // union-pattern entries per thread in registers (256 threads: n_upos <= 1024)

template <int WU, int TD>  // WU: (drive, value) pairs per union entry held in registers (-1: tables read from memory); TD: compile-time d
__global__ __launch_bounds__(256, 3) void pcl_eval_kernel(const KParams p) {
    extern __shared__ double lds[];
    const int d = TD ? TD : p.d, n = 2 * d, m = p.m;
    // leading dimension 2*odd (conflict-free b-operand reads): n itself when d is odd, else the padded value of the other kernels
    const int LD = (d & 1) ? n : (TD ? ((2 * TD + 3) & ~3) + 2 : p.LD);
    const int cols = d;  // unitary states only (X is n x d)
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int li = lane & 15, lk = lane >> 4;
    const int nn = n * n;
    const long long xd = (long long)n * cols;
    const int kfull = n >> 2, krem = n & 3;
    const int cw = min(8, (cols + 3) >> 2);  // chunk width: wave w owns state columns [w*cw, min(cols, (w+1)*cw))
    const int ct_n = (cols + 15) >> 4;       // column tiles of phase 2 (<= 2)

    double *G = lds;
    double *Mall = G + LD * n;
    double *Mw = Mall + wave * (LD * 16);
    double *us = Mall + 4 * (LD * 16);

    // ---- launch-invariant state: drift tile, this thread's union-pattern entries ---------------------------------------
    if (!p.g0_batch_stride)
        load_tile<256, true>(p.G0, G, n, LD, tid);
    constexpr int WUR = WU > 0 ? WU : 1;
    int un_idx[PCL_NUE_EV];
    double un_g0[PCL_NUE_EV];
    unsigned char un_l[PCL_NUE_EV][WUR];
    double un_v[PCL_NUE_EV][WUR];
    if (WU > 0) {
#pragma unroll
        for (int r = 0; r < PCL_NUE_EV; ++r) {
            const int q = tid + 256 * r;
            un_idx[r] = -1;
            un_g0[r] = 0.0;
#pragma unroll
            for (int w = 0; w < WUR; ++w) {
                un_l[r][w] = 0;
                un_v[r][w] = 0.0;
            }
            if (q < p.n_upos) {
                const int pos = p.upos[q];
                un_idx[r] = (pos / n) + LD * (pos % n);
                un_g0[r] = p.ug0[q];  // drift at the pattern entry (table: no load that depends on upos); per-member drifts are read per member
#pragma unroll
                for (int w = 0; w < WUR; ++w) {
                    un_l[r][w] = p.uell_l[q * WUR + w];
                    un_v[r][w] = p.uell_v[q * WUR + w];
                }
            }
        }
    }
    // phase-2 b operand of this lane: column ct*16 + li lives in chunk tile (col / cw), at tile column 8 + col % cw
    const double *Bp2[2];
    bool b2ok[2];
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
        const int col = ct * 16 + li;
        b2ok[ct] = ct < ct_n && col < cols;
        const int cc = b2ok[ct] ? col : 0;
        Bp2[ct] = Mall + (cc / cw) * (LD * 16) + LD * (8 + cc % cw) + lk;
    }
    // zero the operand columns that never receive data (they feed accumulator columns nobody stores; keep them finite)
    for (int e = lane; e < LD * 16; e += 64) Mw[e] = 0.0;

    const int n_items = p.batch * p.K;
    const int c_lo = wave * cw, ncc = max(0, min(cw, cols - c_lo));  // this wave's columns
    double pf_u = 0.0, pxn[8], pxc[8];
    auto request = [&](int item) {
        const int k = item % p.K, b = item / p.K;
        const double *zk = p.Z + (long long)b * p.z_batch_stride + (long long)k * p.z_dim;
        if (tid <= m) pf_u = zk[tid < m ? p.u_off + tid : p.dt_off];
        const int x_off = p.x_offs[p.z_batch_stride ? 0 : b];
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            pxn[c] = pxc[c] = 0.0;
            if (c < ncc && lane < n) {
                const long long o = x_off + (long long)(c_lo + c) * n + lane;
                pxc[c] = zk[o];
                pxn[c] = zk[p.z_dim + o];
            }
        }
    };
    // a workgroup walks a CONTIGUOUS range of intervals: consecutive items share the member, so a per-member drift tile
    // is rewritten only when the member changes
    const int item_lo = (int)((long long)n_items * blockIdx.x / gridDim.x), item_hi = (int)((long long)n_items * (blockIdx.x + 1) / gridDim.x);
    int cur = 0, drift_b = -1;
    if (item_lo < item_hi) {
        request(item_lo);
        if (tid <= m) us[tid] = pf_u;
    }
    __syncthreads();

    for (int item = item_lo; item < item_hi; ++item) {
        const int k = item % p.K, b = item / p.K;
        const double *usc = us + cur * (m + 1);
        const double h = usc[m];
        const double c1 = 0.5 * h, c2 = h * h * (1.0 / 12.0);
        // ---- phase 0: G(u_k) on the union pattern; this wave's [S | D] ---------------------------------------------------
        if (p.g0_batch_stride && drift_b != b) {  // per-member drift: the tile (off the union pattern) changes with the member
            const double *G0b = p.G0 + (long long)b * p.g0_batch_stride;
            for (int e0 = tid; e0 < nn; e0 += 256 * 8) {  // eight independent loads in flight per thread
                double v[8];
                int um[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int e = e0 + 256 * j;
                    um[j] = e < nn ? p.umap[e] : 0;
                    v[j] = e < nn ? G0b[e] : 0.0;
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int e = e0 + 256 * j;
                    if (um[j] < 0) G[(e / n) + LD * (e % n)] = v[j];
                }
            }
            if (WU > 0) {
#pragma unroll
                for (int r = 0; r < PCL_NUE_EV; ++r)
                    if (un_idx[r] >= 0) un_g0[r] = G0b[p.upos[tid + 256 * r]];
            }
            drift_b = b;
        }
        if (WU > 0) {
#pragma unroll
            for (int r = 0; r < PCL_NUE_EV; ++r)
                if (un_idx[r] >= 0) {
                    double g = un_g0[r];
#pragma unroll
                    for (int w = 0; w < WUR; ++w) g += usc[un_l[r][w]] * un_v[r][w];
                    G[un_idx[r]] = g;
                }
        } else {
            const double *G0b = p.G0 + (long long)b * p.g0_batch_stride;
            for (int q = tid; q < p.n_upos; q += 256) {
                const int pos = p.upos[q];
                double g = G0b[pos];
                const double *cf = p.ucoef + (long long)q * m;
                for (int l = 0; l < m; ++l) g += usc[l] * cf[l];
                G[(pos / n) + LD * (pos % n)] = g;
            }
        }
        if (lane < n) {
#pragma unroll
            for (int c = 0; c < 8; ++c)
                if (c < ncc) {
                    Mw[lane + LD * c] = pxn[c] + pxc[c];
                    Mw[lane + LD * (8 + c)] = pxn[c] - pxc[c];
                }
        }
        // inputs of this workgroup's next item: in flight during the matrix phases
        if (item + 1 < item_hi) request(item + 1);
        __syncthreads();
        // ---- phase 1: [G S | G D] for this wave's chunk: all row tiles at once ----------------------------------------------
        if (ncc > 0) {
            const double *Bp = Mw + lk + LD * li;
            const double *Ap[PCL_MAXRT];
            bool rok[PCL_MAXRT];
            double4_t acc[PCL_MAXRT];
#pragma unroll
            for (int t = 0; t < PCL_MAXRT; ++t) {
                rok[t] = t * 16 < n;
                Ap[t] = G + lk + LD * ((rok[t] ? t * 16 : 0) + li);
                acc[t] = double4_t{0.0, 0.0, 0.0, 0.0};
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
            // in place: columns 0..7 <- P = D - c1 G S (accumulator column li < 8 holds G S of chunk column li),
            //           columns 8..15 <- G D.  Every lane first reads the D entries it needs, then the tile is rewritten.
            double dv[PCL_MAXRT][4];
#pragma unroll
            for (int t = 0; t < PCL_MAXRT; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = t * 16 + lk + 4 * r;
                    dv[t][r] = (li < 8 && row < n) ? Mw[row + LD * (8 + li)] : 0.0;
                }
            wave_lds_sync();
#pragma unroll
            for (int t = 0; t < PCL_MAXRT; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = t * 16 + lk + 4 * r;
                    if (row < n) Mw[row + LD * li] = li < 8 ? dv[t][r] - c1 * acc[t][r] : acc[t][r];
                }
        }
        __syncthreads();
        // ---- phase 2: row tile `wave` of G (G D), all column tiles; delta = P + c2 G (G D) in place ---------------------------
        if (wave * 16 < n) {
            const double *Ap = G + lk + LD * (wave * 16 + li);
            // k-step ks into accumulator ks mod 2 of each column tile: four independent chains (a dependent f64 MFMA waits
            // ~235 cycles for its accumulator, four in flight keep the pipe issuing every 64)
            double4_t q0[2] = {{0.0, 0.0, 0.0, 0.0}, {0.0, 0.0, 0.0, 0.0}}, q1[2] = {{0.0, 0.0, 0.0, 0.0}, {0.0, 0.0, 0.0, 0.0}};
            for (int ks0 = 0; ks0 < kfull; ks0 += 2) {
                double a[2], b0[2], b1[2];
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const bool ok = ks0 + j < kfull;
                    a[j] = ok ? Ap[4 * (ks0 + j)] : 0.0;
                    b0[j] = ok ? Bp2[0][4 * (ks0 + j)] : 0.0;
                    b1[j] = ok ? Bp2[1][4 * (ks0 + j)] : 0.0;
                }
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    if (ks0 + j < kfull) {
                        q0[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[j], b0[j], q0[j], 0, 0, 0);
                        if (ct_n > 1) q1[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[j], b1[j], q1[j], 0, 0, 0);
                    }
            }
            if (krem) {
                const bool ok = lk < krem;
                const double a = ok ? Ap[4 * kfull] : 0.0;
                q0[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, ok ? Bp2[0][4 * kfull] : 0.0, q0[0], 0, 0, 0);
                if (ct_n > 1) q1[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, ok ? Bp2[1][4 * kfull] : 0.0, q1[0], 0, 0, 0);
            }
            const double4_t a0 = q0[0] + q0[1], a1 = q1[0] + q1[1];
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
                if (b2ok[ct]) {
                    const int col = ct * 16 + li;
                    double *P = Mall + (col / cw) * (LD * 16) + LD * (col % cw);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int row = wave * 16 + lk + 4 * r;
                        if (row < n) P[row] += c2 * (ct ? a1[r] : a0[r]);
                    }
                }
        }
        if (tid <= m) us[(cur ^ 1) * (m + 1) + tid] = pf_u;
        cur ^= 1;
        __syncthreads();
        // ---- phase 3: this wave's columns of delta are contiguous: element e = c*n + row, two per lane -------------------
        if (ncc > 0) {
            double *o = p.delta + ((long long)b * p.K + k) * xd + (long long)c_lo * n;
            const int hn = n >> 1;
            for (int e2 = lane; e2 < ncc * hn; e2 += 64) {
                const int c = e2 / hn, r0 = 2 * (e2 - c * hn);
                store2(o + (long long)c * n + r0, Mw[r0 + LD * c], Mw[r0 + 1 + LD * c], false);
            }
        }
        // (no barrier: the next item's phase 0 rewrites only this wave's own tile, and G, which nobody reads after phase 2)
    }
}
