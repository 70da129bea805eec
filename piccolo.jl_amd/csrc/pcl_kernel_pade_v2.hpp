// Tuned general-order kernel: residual + Jacobian for the diagonal Pade orders p = 2q (q <= 5) that the fused kernels (order 4) do
// not cover.  Same outputs as pcl_pade_kernel<true> (pcl_kernels_reference.hpp, which stays as the second implementation and
// as the fallback for shapes this one does not take: odd n, slices too narrow for the LDS budget).
//
// The three Horner recursions of the reference formulation run in LOCK STEP, one matrix product per level instead of three
// dependent chains (with Y_j = D for even j, -S for odd j; c_j the Pade coefficients; h the step):
//     level q:            W = c_q Y_q            V = q c_q Y_q               dW_l = 0
//     level j = q-1..1:   W <- c_j Y_j + h G W   V <- j c_j Y_j + h G V      dW_l <- h (G_l W_old + G dW_l)
//     level 0:            delta = D + h G W      d delta/dh = G V            d delta/du_l = h (G_l W_old + G dW_l)
// i.e. per level ONE product  G [W | V | dW_0 .. dW_{m-1}]  (n x (2+m) nc columns) on the matrix cores, from one LDS buffer into the
// other (one workgroup barrier per level; the drives' sparse term G_l W_old is added by the lane that owns the element).
// The A operand (G, fixed for the interval) stays in registers over all levels.
//
// Workgroup roles (1024 threads, one role per workgroup):
//   columns role (items * S workgroups, first in the grid)   slice s of an interval's state columns: the recursion above,
//                                                            writes delta and the u / dt columns of the Jacobian
//   blocks role  (items workgroups, last in the grid)        powers of G by repeated products, B^{+-} = sum_j c_j (+-h)^j G^j,
//                                                            writes the ONE copy of -B^+ and B^- (compact layout: in place;
//                                                            full layout: into a scratch that pcl_replicate_kernel streams
//                                                            into the d replicated positions at HBM rate)
// LDS (doubles), columns role: G | -S | D | X (2+m) | X' (2+m) | us | drives' ELL rows      blocks role: G | Pa | Pb
// (LD odd: conflict-free b operand)
#pragma once

#define PV2_KS 16   // k-steps of 4 (n <= 64)
#define PV2_NT 1024  // threads per workgroup
#define PV2_NP 2    // B^{+-} value pairs per thread: n n / 2 / PV2_NT <= 2 for n <= 64

// one 16 x 16 tile of G * B: a[] = this wave's rows of G, Bp = this lane's column of B in LDS (nullptr: zero column), kmask = the
// k-steps to take (wave-uniform)
__device__ __forceinline__ double4_t pv2_tile(const double (&a)[PV2_KS], const double *__restrict__ Bp, int n, int lk, unsigned kmask) {
    double4_t acc = {0.0, 0.0, 0.0, 0.0};
    double b[PV2_KS];
#pragma unroll
    for (int ks = 0; ks < PV2_KS; ++ks) {
        const int kk = 4 * ks + lk;
        b[ks] = (Bp && kk < n) ? Bp[kk] : 0.0;
    }
#pragma unroll
    for (int ks = 0; ks < PV2_KS; ++ks)
        if (kmask & (1u << ks)) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[ks], b[ks], acc, 0, 0, 0);
    return acc;
}

__global__ __launch_bounds__(PV2_NT) void pcl_pade_v2_kernel(const KParams p, double *__restrict__ blocks) {
    extern __shared__ double lds[];
    const int n = p.n, d = p.cols, m = p.m, LD = p.LD, nc = p.nc, q = p.q, S = p.S;
    const int tid = threadIdx.x, nth = blockDim.x, lane = tid & 63, wave = tid >> 6, nw = nth >> 6;
    const int li = lane & 15, lk = lane >> 4;
    const long long items = (long long)p.batch * p.K;
    const long long bid = blockIdx.x;
    const bool blocks_role = bid >= items * S;
    const long long item = blocks_role ? bid - items * S : bid / S;
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

    double *G = lds;
    if (blocks_role) {
        double *Pa = G + LD * n, *Pb = Pa + LD * n;
        build_G(p, p.G0 + (long long)b * p.g0_batch_stride, zk, G, Pb);  // (Pb: scratch for the controls)
        __syncthreads();
        double a[PV2_KS];
#pragma unroll
        for (int ks = 0; ks < PV2_KS; ++ks) {
            const int row = rt * 16 + li, kk = 4 * ks + lk;
            a[ks] = (row < n && kk < n) ? G[row + LD * kk] : 0.0;
        }
        unsigned kmask = 0;
#pragma unroll
        for (int ks = 0; ks < PV2_KS; ++ks)
            if (__ballot(a[ks] != 0.0)) kmask |= 1u << ks;
        kmask = __builtin_amdgcn_readfirstlane(kmask);
        // each thread owns the flat column-major positions 2 (tid + nth r), +1 (n is even: same column)
        double bp[PV2_NP][2], bm[PV2_NP][2];
        int o_[PV2_NP];
#pragma unroll
        for (int r = 0; r < PV2_NP; ++r) {
            const int pos = 2 * (tid + nth * r);
            o_[r] = -1;
            bp[r][0] = bm[r][0] = bp[r][1] = bm[r][1] = 0.0;
            if (pos < nn) {
                const int i = pos % n, jj = pos / n;
                o_[r] = i + LD * jj;
                bp[r][0] = bm[r][0] = (i == jj) ? 1.0 : 0.0;
                bp[r][1] = bm[r][1] = (i + 1 == jj) ? 1.0 : 0.0;
            }
        }
        const double *Pc = G;
        double hp = 1.0, hm = 1.0;
        for (int j = 1; j <= q; ++j) {
            hp *= h;
            hm *= -h;
#pragma unroll
            for (int r = 0; r < PV2_NP; ++r) {
                const double v0 = o_[r] >= 0 ? Pc[o_[r]] : 0.0, v1 = o_[r] >= 0 ? Pc[o_[r] + 1] : 0.0;
                bp[r][0] += p.pc[j] * hp * v0;
                bp[r][1] += p.pc[j] * hp * v1;
                bm[r][0] += p.pc[j] * hm * v0;
                bm[r][1] += p.pc[j] * hm * v1;
            }
            if (j < q) {
                double *Pn = (Pc == Pa) ? Pb : Pa;
                if (!idle)
                    for (int ct = cw; ct < rt_n; ct += wpr) {
                        const int col = ct * 16 + li;
                        const double4_t acc = pv2_tile(a, col < n ? Pc + LD * col : nullptr, n, lk, kmask);
                        if (col < n) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                const int rr = rt * 16 + lk + 4 * r;
                                if (rr < n) Pn[rr + LD * col] = acc[r];
                            }
                        }
                    }
                __syncthreads();
                Pc = Pn;
            }
        }
        double *dst = (p.compact || !blocks) ? jb : blocks + item * 2 * nn;  // (one state column: the full layout IS the compact one)
#pragma unroll
        for (int r = 0; r < PV2_NP; ++r) {
            const int pos = 2 * (tid + nth * r);
            if (pos < nn) {
                store2(dst + pos, -bp[r][0], -bp[r][1], 0);
                store2(dst + nn + pos, bm[r][0], bm[r][1], 0);
            }
        }
        return;
    }

    // ---- columns role ---------------------------------------------------------------------------------------------------
    const int s = (int)(bid % S);
    const int c0 = s * nc, nce = min(nc, d - c0), LDc = LD * nc, T = 2 + m;
    double *Sm = G + LD * n, *Dm = Sm + LDc, *Xc = Dm + LDc, *Xn = Xc + T * LDc;  // X: W | V | dW_0 .. dW_{m-1}
    double *us = Xn + T * LDc;
    const double *zn = zk + p.z_dim;
    const int x_off = p.x_offs[p.z_batch_stride ? 0 : b];
    // the drives' rows in ELL form (fixed width, zero padded): staged in LDS where the host found room, else read from memory
    const int ew = p.ell_w;
    const double *ev = p.ell_val;
    const int *ec = p.ell_col;
    if (p.ell_lds) {
        double *evl = us + ((m + 2) & ~1);
        int *ecl = reinterpret_cast<int *>(evl + m * n * ew);
        for (int e = tid; e < m * n * ew; e += nth) {
            evl[e] = p.ell_val[e];
            ecl[e] = p.ell_col[e];
        }
        ev = evl;
        ec = ecl;
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
    const int ctot = T * nc, ct_n = (ctot + 15) >> 4;
    for (int j = q - 1; j >= 0; --j) {
        const double *Yj = (j & 1) ? Sm : Dm;
        const double cj = p.pc[j];
        if (!idle)
            for (int ct = cw; ct < ct_n; ct += wpr) {
                const int vc = ct * 16 + li, bl = vc / nc, c = vc - bl * nc;
                const bool on = vc < ctot;
                double4_t acc = {0.0, 0.0, 0.0, 0.0};
                if (!(j == q - 1 && ct * 16 >= 2 * nc))  // (dW is zero at the first level)
                    acc = pv2_tile(a, on ? Xc + bl * LDc + LD * c : nullptr, n, lk, kmask);
                if (on) {
                    // the additive term of each element: c_j Y_j, j c_j Y_j, or the drives' sparse product with the old W
                    double y[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int rr = rt * 16 + lk + 4 * r;
                        double v = 0.0;
                        if (rr < n) {
                            if (bl <= 1) {
                                v = Yj[rr + LD * c];
                            } else {
                                const int eb = ((bl - 2) * n + rr) * ew;
                                for (int e = 0; e < ew; ++e) v += ev[eb + e] * Xc[ec[eb + e] + LD * c];
                            }
                        }
                        y[r] = v;
                    }
                    double *dst = Xn + bl * LDc + LD * c;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int rr = rt * 16 + lk + 4 * r;
                        if (rr < n) {
                            double o;
                            if (bl == 0)
                                o = cj * y[r] + h * acc[r];
                            else if (bl == 1)
                                o = j ? j * cj * y[r] + h * acc[r] : acc[r];
                            else
                                o = h * (acc[r] + y[r]);
                            dst[rr] = o;
                        }
                    }
                }
            }
        __syncthreads();  // level j is complete in Xn; nothing reads Xc any more
        double *t_ = Xc;
        Xc = Xn;
        Xn = t_;
    }
    const long long xd = (long long)n * d;
    if (p.delta)
        for (int e = tid; e < nce * n; e += nth) p.delta[item * xd + (long long)c0 * n + e] = Xc[(e % n) + LD * (e / n)];
    double *jt = jb + 2 * blk + (long long)c0 * (m + 1) * n;
    for (int e = tid; e < (m + 1) * nce * n; e += nth) {
        const int i = e % n, l = (e / n) % (m + 1), c = e / (n * (m + 1));
        jt[e] = Xc[(l < m ? (2 + l) * LDc : LDc) + i + LD * c];
    }
}

// the single copy of -B^+ / B^- per interval (blocks[item][2][n n]) -> the d replicated positions of the full Jacobian layout
__global__ __launch_bounds__(256) void pcl_replicate_kernel(const double *__restrict__ blocks, double *__restrict__ full, int d, int n,
                                                            long long fper, long long n_bk, int nt) {
    const long long nn = (long long)n * n;
    const long long bid = blockIdx.x;
    const int c = (int)(bid % d);
    const long long bk = bid / d;
    if (bk >= n_bk) return;
    const double *src = blocks + bk * 2 * nn;
    double *dst = full + bk * fper;
    for (long long q = threadIdx.x; q < (nn >> 1); q += blockDim.x) {
        const double2_t v0 = *reinterpret_cast<const double2_t *>(src + 2 * q);
        const double2_t v1 = *reinterpret_cast<const double2_t *>(src + nn + 2 * q);
        store2(dst + c * nn + 2 * q, v0[0], v0[1], nt);
        store2(dst + (d + c) * nn + 2 * q, v1[0], v1[1], nt);
    }
}
