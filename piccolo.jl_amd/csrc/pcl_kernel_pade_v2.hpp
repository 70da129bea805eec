// Tuned general-order kernel: residual + Jacobian for the diagonal Pade orders p = 2q (q <= 5) that the fused kernels (order 4) do
// not cover.  Same outputs as pcl_pade_kernel<true> (pcl_kernels_reference.hpp, which stays as the second implementation and
// as the fallback for shapes this one does not take: odd n, slices too narrow for the LDS budget).
//
// The three Horner recursions of the reference formulation AND the powers of G run in LOCK STEP, one matrix product per level
// instead of four dependent chains (with Y_j = D for even j, -S for odd j; c_j the Pade coefficients; h the step):
//     level q:            W = c_q Y_q            V = q c_q Y_q               dW_l = 0                          P = G
//     level j = q-1..1:   W <- c_j Y_j + h G W   V <- j c_j Y_j + h G V      dW_l <- h (G_l W_old + G dW_l)    P <- G P
//     level 0:            delta = D + h G W      d delta/dh = G V            d delta/du_l = h (G_l W_old + G dW_l)
// i.e. per level ONE product  G [W | V | dW_0 .. dW_{m-1} | P]  on the matrix cores, from one LDS buffer into the other (one
// workgroup barrier per level; the drives' sparse term G_l W_old is added by the lane that owns the element).  The A operand
// (G, fixed for the interval) stays in registers over all levels; its 16 x 4 blocks without a nonzero are skipped.
// B^{+-} = sum_j c_j (+-h)^j G^j accumulates in registers as the powers appear and is written to its d replicated positions
// by the workgroup that formed it.
//
// One workgroup (1024 threads) per (interval, slice): slice s takes nc of the state columns (W, V, dW) and npc of the n columns
// of the powers, so every workgroup of the grid does the same work.
// LDS (doubles): G | -S | D | X (2+m) | X' (2+m) | P' | zeros | us | drives' ELL rows      (P starts as the slice's columns of G in place;
// LD odd: conflict-free b operand)
#pragma once

#define PV2_KS 16   // k-steps of 4 (n <= 64)
#define PV2_NT 1024  // threads per workgroup
#define PV2_NP 2    // B^{+-} value pairs per thread: n n / 2 / PV2_NT <= 2 for n <= 64

// one 16 x 16 tile of G * B: a[] = this wave's rows of G, Bp = this lane's column of B in LDS + (lane >> 4), kmask = the k-steps
// to take (wave-uniform).  The 16 operand loads are unconditional, base + immediate: rows beyond n meet a = 0, and every
// double of the workgroup's LDS is finite (zero-filled at the start; a column of zeros stands in for columns beyond the last)
__device__ __forceinline__ double4_t pv2_tile(const double (&a)[PV2_KS], const double *__restrict__ Bp, unsigned kmask) {
    double4_t acc = {0.0, 0.0, 0.0, 0.0};
    double b[PV2_KS];
#pragma unroll
    for (int ks = 0; ks < PV2_KS; ++ks) b[ks] = Bp[4 * ks];
#pragma unroll
    for (int ks = 0; ks < PV2_KS; ++ks)
        if (kmask & (1u << ks)) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[ks], b[ks], acc, 0, 0, 0);
    return acc;
}

__global__ __launch_bounds__(PV2_NT) void pcl_pade_v2_kernel(const KParams p) {
    extern __shared__ double lds[];
    const int n = p.n, d = p.cols, m = p.m, LD = p.LD, nc = p.nc, q = p.q, S = p.S;
    const int tid = threadIdx.x, nth = blockDim.x, lane = tid & 63, wave = tid >> 6, nw = nth >> 6;
    const int li = lane & 15, lk = lane >> 4;
    const long long bid = blockIdx.x;
    const long long item = bid / S;
    const int s = (int)(bid % S);
    const int k = (int)(item % p.K), b = (int)(item / p.K);
    const double *zk = p.Z + (long long)b * p.z_batch_stride + (long long)k * p.z_dim;
    const double h = zk[p.dt_off];
    const int rt_n = (n + 15) >> 4;
    const int wpr = nw / rt_n;                // waves per row tile
    const int rt = wave % rt_n, cw = wave / rt_n;  // this wave's row tile and its first column tile
    const bool idle = cw >= wpr;
    const long long nn = (long long)n * n;
    double *jb = p.jac + item * p.jac_per;
    const long long blk = p.compact ? nn : (long long)d * nn;
    const int npc = (n + S - 1) / S, pc0 = s * npc, npce = max(0, min(npc, n - pc0));  // this slice's columns of the powers
    const int c0 = s * nc, nce = max(0, min(nc, d - c0)), LDc = LD * nc, T = 2 + m;
    double *G = lds;
    double *Sm = G + LD * n, *Dm = Sm + LDc, *Xc = Dm + LDc, *Xn = Xc + T * LDc;  // X: W | V | dW_0 .. dW_{m-1}
    double *Pn = Xn + T * LDc, *Pc = G + LD * pc0;
    double *zcol = Pn + LD * npc;  // 64 zeros (+ 16 of slack behind the last tile: operand loads run up to row 63 of a column)
    double *us = zcol + 80;
    for (int e = tid; e < p.lds_doubles; e += nth) lds[e] = 0.0;
    __syncthreads();
    const double *zn = zk + p.z_dim;
    const int x_off = p.x_offs[p.z_batch_stride ? 0 : b];
    // the drives' rows in ELL form (fixed width, zero padded): staged in LDS where the host found room, else read from memory
    const int ew = p.ell_w;
    double *evl = us + ((m + 2) & ~1);
    int *ecl = reinterpret_cast<int *>(evl + m * n * ew);
    if (p.ell_lds)
        for (int e = tid; e < m * n * ew; e += nth) {
            evl[e] = p.ell_val[e];
            ecl[e] = p.ell_col[e];
        }
    build_G(p, p.G0 + (long long)b * p.g0_batch_stride, zk, G, us);
    const double cq = p.pc[q];
    for (int e = tid; e < nc * n; e += nth) {
        const int c = e / n, i = e % n;
        double xs = 0.0, xdv = 0.0;
        if (c < nce) {
            const double xn = zn[x_off + (c0 + c) * n + i], xc = zk[x_off + (c0 + c) * n + i];
            xs = xn + xc;
            xdv = xn - xc;
        }
        const int idx = i + LD * c;
        Sm[idx] = -xs;
        Dm[idx] = xdv;
        const double yq = (q & 1) ? -xs : xdv;
        Xc[idx] = cq * yq;
        Xc[LDc + idx] = q * cq * yq;
        for (int l = 0; l < m; ++l) Xc[(2 + l) * LDc + idx] = 0.0;
    }
    __syncthreads();
    double a[PV2_KS];
#pragma unroll
    for (int ks = 0; ks < PV2_KS; ++ks) {
        const int row = rt * 16 + li, kk = 4 * ks + lk;
        a[ks] = (row < n && kk < n) ? G[row + LD * kk] : 0.0;
    }
    // 16 x 4 blocks of G without a nonzero (the generators are sparse: at BASELINE config 3 G has 21 % nonzeros) are skipped:
    // adding their exact-zero products changes nothing (but the sign of a zero sum)
    unsigned kmask = 0;
#pragma unroll
    for (int ks = 0; ks < PV2_KS; ++ks)
        if (__ballot(a[ks] != 0.0)) kmask |= 1u << ks;
    kmask = __builtin_amdgcn_readfirstlane(kmask);
    // B^{+-}: each thread owns the positions (2 pi, 2 pi + 1), pi = tid + nth r, of the slice's n x npce block of columns (n is
    // even: both in one column); first term c_1 (+-h) G
    double bp[PV2_NP][2], bm[PV2_NP][2];
    int o_[PV2_NP];
    double hp = h, hm = -h;
#pragma unroll
    for (int r = 0; r < PV2_NP; ++r) {
        const int pos = 2 * (tid + nth * r);
        o_[r] = -1;
        bp[r][0] = bm[r][0] = bp[r][1] = bm[r][1] = 0.0;
        if (pos < n * npce) {
            const int i = pos % n, jj = pc0 + pos / n;
            o_[r] = i + LD * (pos / n);
            bp[r][0] = bm[r][0] = (i == jj) ? 1.0 : 0.0;
            bp[r][1] = bm[r][1] = (i + 1 == jj) ? 1.0 : 0.0;
            const double v0 = Pc[o_[r]], v1 = Pc[o_[r] + 1];
            bp[r][0] += p.pc[1] * hp * v0;
            bp[r][1] += p.pc[1] * hp * v1;
            bm[r][0] += p.pc[1] * hm * v0;
            bm[r][1] += p.pc[1] * hm * v1;
        }
    }
    const int cx = T * nc, ctot = cx + npc, ct_n = (ctot + 15) >> 4;
    const unsigned inv_nc = (65536u + nc - 1) / nc;  // vc / nc = (vc inv_nc) >> 16 for vc < 2048
    for (int j = q - 1; j >= 0; --j) {
        const double *Yj = (j & 1) ? Sm : Dm;
        const double cj = p.pc[j];
        if (!idle)
            for (int ct = cw; ct < ct_n; ct += wpr) {
                const int vc = ct * 16 + li;
                const bool isp = vc >= cx;                       // a column of the powers
                const int bl = isp ? T : (int)((vc * inv_nc) >> 16), c = isp ? vc - cx : vc - bl * nc;
                const bool on = isp ? (j > 0 && c < npce) : true;
                // (dW is zero at the first level; the last level forms no power)
                const bool skip = (j == q - 1 && ct * 16 >= 2 * nc && ct * 16 + 15 < cx) || (j == 0 && ct * 16 >= cx);
                double4_t acc = {0.0, 0.0, 0.0, 0.0};
                if (!skip) acc = pv2_tile(a, (on ? (isp ? Pc : Xc + bl * LDc) + LD * c : zcol) + lk, kmask);
                if (on) {
                    // the additive term of each element: c_j Y_j, j c_j Y_j, or the drives' sparse product with the old W.  Loads
                    // are unconditional (rows beyond n - 1 read finite padding; the stores below are predicated)
                    const int r0 = rt * 16 + lk;
                    double y[4] = {0.0, 0.0, 0.0, 0.0};
                    if (bl <= 1) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) y[r] = Yj[r0 + 4 * r + LD * c];
                    } else if (!isp) {
                        const double *wc = Xc + LD * c;
                        // (two code paths, one per address space of the ELL rows: a pointer that may be either is a FLAT load)
                        auto sparse = [&](const auto *evp, const auto *ecp) {
                            if (ew == 2) {  // the common width: all loads of the four rows in flight together
                                double v_[4][2];
                                int c_[4][2];
#pragma unroll
                                for (int r = 0; r < 4; ++r) {
                                    const int eb = ((bl - 2) * n + min(r0 + 4 * r, n - 1)) * 2;
                                    v_[r][0] = evp[eb];
                                    v_[r][1] = evp[eb + 1];
                                    c_[r][0] = ecp[eb];
                                    c_[r][1] = ecp[eb + 1];
                                }
#pragma unroll
                                for (int r = 0; r < 4; ++r) y[r] = v_[r][0] * wc[c_[r][0]] + v_[r][1] * wc[c_[r][1]];
                            } else {
#pragma unroll
                                for (int r = 0; r < 4; ++r) {
                                    const int eb = ((bl - 2) * n + min(r0 + 4 * r, n - 1)) * ew;
                                    double v = 0.0;
                                    for (int e = 0; e < ew; ++e) v += evp[eb + e] * wc[ecp[eb + e]];
                                    y[r] = v;
                                }
                            }
                        };
                        if (p.ell_lds)
                            sparse(evl, ecl);
                        else
                            sparse(p.ell_val, p.ell_col);
                    }
                    double *dst = (isp ? Pn : Xn + bl * LDc) + LD * c;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int rr = r0 + 4 * r;
                        double o;
                        if (bl == 0)
                            o = cj * y[r] + h * acc[r];
                        else if (bl == 1)
                            o = j ? j * cj * y[r] + h * acc[r] : acc[r];
                        else if (!isp)
                            o = h * (acc[r] + y[r]);
                        else
                            o = acc[r];
                        if (rr < n) dst[rr] = o;
                    }
                }
            }
        __syncthreads();  // level j is complete in X' / P'; nothing reads X / P any more
        double *t_ = Xc;
        Xc = Xn;
        Xn = t_;
        if (j > 0) {  // the next power: its term of B^{+-}
            t_ = Pc;
            Pc = Pn;
            Pn = t_;
            const int pw = q + 1 - j;
            hp *= h;
            hm *= -h;
#pragma unroll
            for (int r = 0; r < PV2_NP; ++r)
                if (o_[r] >= 0) {
                    const double v0 = Pc[o_[r]], v1 = Pc[o_[r] + 1];
                    bp[r][0] += p.pc[pw] * hp * v0;
                    bp[r][1] += p.pc[pw] * hp * v1;
                    bm[r][0] += p.pc[pw] * hm * v0;
                    bm[r][1] += p.pc[pw] * hm * v1;
                }
        }
    }
    const long long xd = (long long)n * d;
    if (p.delta)
        for (int e = tid; e < nce * n; e += nth) p.delta[item * xd + (long long)c0 * n + e] = Xc[(e % n) + LD * (e / n)];
    double *jt = jb + 2 * blk + (long long)c0 * (m + 1) * n;
    for (int e = tid; e < (m + 1) * nce * n; e += nth) {
        const int i = e % n, l = (e / n) % (m + 1), c = e / (n * (m + 1));
        jt[e] = Xc[(l < m ? (2 + l) * LDc : LDc) + i + LD * c];
    }
    // -B^+ and B^-: this slice's columns, into every replicated position (compact layout: the one copy)
    const int copies = p.compact ? 1 : d;
#pragma unroll
    for (int r = 0; r < PV2_NP; ++r)
        if (o_[r] >= 0) {
            double *o0 = jb + (long long)pc0 * n + 2 * (tid + nth * r);
            for (int c = 0; c < copies; ++c) {
                store2(o0 + c * nn, -bp[r][0], -bp[r][1], p.nt);
                store2(o0 + blk + c * nn, bm[r][0], bm[r][1], p.nt);
            }
        }
}
