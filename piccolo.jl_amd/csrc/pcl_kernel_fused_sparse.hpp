// pcl_kernel_fused_sparse.hpp -- fused residual + Jacobian, PATTERN-COMPILED, any diagonal Pade order 2q.
// Included by generated source only (pcl_codegen_v4.hpp): SPD (Hilbert dimension), SPM (drives), SPN = 2 SPD, SP4Q (q), the
// resident-coefficient struct sp4_cf, the product sp4_product and the drives' gathers sp4_gather_<l> are defined before this file.
//
// With Y_j = D (j even) or -S (j odd), D = X_{k+1} - X_k, S = X_{k+1} + X_k, c_j the Pade coefficients, h the step:
//     level q:           W = c_q Y_q          V = q c_q Y_q            dW_l = 0                            P = I
//     level j = q-1..0:  W <- c_j Y_j + h G W V <- j c_j Y_j + h G V   dW_l <- h (G_l W_old + G dW_l)      P <- G P
//     (level 0:          delta = W            d delta/dh = G V         d delta/du_l = dW_l)                B^{+-} = sum_j c_j (+-h)^j G^j
// Every chain acts on the state columns from the left: lane (half, c) owns its half of column c, G(u) x is the straight-line
// product sp4_product (coefficients in scalar registers, no LDS operand traffic, no matrix-core padding: 307 multiply-adds
// instead of 112 MFMAs per 27 columns at BASELINE config 3).  ONE persistent workgroup per CU, one WAVE per chain:
//     wave 0            P: the powers of G (first d columns: the generators are exact iso(.) images), B^+ / B^- accumulated in two tiles
//     wave 1, 2         W, V
//     wave 3 + l        dW_l
//     wave 3 + m        loader: D, S of the next item (lane = row, coalesced) -> tiles [column][row]
//     wave 4 + m .. +3  stream: copy the item's -B^+ / B^- values into registers, then only issue the replicated 16-byte stores
// No workgroup barrier after the start: point-to-point monotonic LDS counters (dependencies only point backwards; bounded waits).
// Work items as in kernel 3: contiguous column ranges per workgroup (pieces of one interval), or round-robin slices.
#pragma once

#define SP4CS (SPN + 1)           // odd column stride: the lanes of a half wave, one column each, hit distinct banks
#define SP4TILE (SP4CS * SPD)
#define SP4_WLOAD (SPM + 3)
#define SP4_WSTREAM (SPM + 4)
#define SP4_NSTREAM 4
#define SP4_NWAVES (SPM + 8)
#define SP4_NTILES (SPM + 7)      // D, S, W, V, dW[m], P, B+, B-
#define SP4_SYNC_WORDS 32
enum { SP4_F_IN = 0, SP4_F_DW, SP4_F_DV, SP4_F_W, SP4_F_B, SP4_F_C, SP4_F_G = 8 /* one word per drive wave */ };

static __device__ __forceinline__ bool sp4_wait(int *sync, int word, int target, bool gave_up = false) {
    // Bounded: a logic error must not hang the device (the caller poisons the output instead; once a wave has given up it
    // does not wait again).
    if (gave_up) return true;
    int it = 0;
    for (; __hip_atomic_load(sync + word, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < target && it < (1 << 20); ++it) __builtin_amdgcn_s_sleep(1);
    return it >= (1 << 20);
}
static __device__ __forceinline__ void sp4_post(int *w, int value, int lane) {  // after wave_lds_sync(): this wave's LDS traffic is complete
    if (lane == 0) __hip_atomic_store(w, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
}
static __device__ __forceinline__ void sp4_arrive(int *w, int lane) {
    if (lane == 0) __hip_atomic_fetch_add(w, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
}
static __device__ __forceinline__ unsigned sp4_lds_off(const double *q) {
    return (unsigned)(size_t)(__attribute__((address_space(3))) const double *)q;
}

extern "C" __global__ __launch_bounds__(64 * SP4_NWAVES) void pcl_fused_sparse_kernel(const KParams p, const double *__restrict__ drift_tab, const double *__restrict__ mags_) {
    extern __shared__ double lds[];
    constexpr int d = SPD, n = SPN, m = SPM, q = SP4Q, nn = SPN * SPN;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // scalar: the role branches are uniform
    double *Dt = lds, *St = Dt + SP4TILE, *Wt = St + SP4TILE, *Vt = Wt + SP4TILE, *dWt = Vt + SP4TILE;
    double *Pt = dWt + m * SP4TILE, *Bpt = Pt + SP4TILE, *Bmt = Bpt + SP4TILE;
    int *sync = (int *)(Bmt + SP4TILE);
    for (int e = tid; e < SP4_NTILES * SP4TILE + SP4_SYNC_WORDS / 2; e += 64 * SP4_NWAVES) lds[e] = 0.0;  // (finite everywhere; counters zero)
    __syncthreads();  // the only workgroup barrier

    // ---- work split (as kernel 3) ----------------------------------------------------------------------------------------
    const long long blk = p.compact ? (long long)nn : (long long)d * nn;  // size of the -B^+ / of the B^- segment
    const long long xd = (long long)n * d;
    const int bx = (int)blockIdx.x;
    int n_my;
    long long g_lo = 0, g_hi = 0;
    if (p.contig) {
        const long long tot = (long long)p.batch * p.K * d;
        g_lo = tot * bx / (long long)gridDim.x;
        g_hi = tot * (bx + 1) / (long long)gridDim.x;
        n_my = g_hi > g_lo ? (int)((g_hi - 1) / d - g_lo / d) + 1 : 0;
    } else {
        const int n_items = p.batch * p.K * p.S;
        n_my = n_items > bx ? (n_items - bx + (int)gridDim.x - 1) / (int)gridDim.x : 0;
    }
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
            c0 = s * p.nc;
            nce = min(p.nc, d - c0);
            k = (item / p.S) % p.K;
            b = item / (p.S * p.K);
        }
    };
    bool gave_up = false;

    if (wave < SP4_WLOAD) {
        // ================================== column waves: one chain each ====================================================
        // Lane position, re-derived from an opaque copy of `lane` in every item: computed once, everything that depends on it (the
        // unit vectors, tile addresses, ...) is hoisted out of the item loops and spilled.
#define SP4_LANEPOS()                                                                       \
    int ln_ = lane;                                                                         \
    asm volatile("" : "+v"(ln_));                                                           \
    const int half = ln_ >> 5, c = ln_ & 31;                                                \
    const int cc = c < d ? c : 0;                                                           \
    const int own = cc * SP4CS + half * d, oth = cc * SP4CS + (1 - half) * d
        sp_cptr magc = (sp_cptr)mags_;
        double mg[SP4NMAG];
#pragma unroll
        for (int g = 0; g < SP4NMAG; ++g) mg[g] = magc[g];
        // per-item scalars: step, controls -> resident coefficients, the member's drift table
        auto scalars = [&](int k, int b, double &h, sp4_cf &cf, sp_cptr &tab) {
            sp_cptr zc = (sp_cptr)(p.Z + (long long)b * p.z_batch_stride + (long long)k * p.z_dim);
            double u[SPM > 0 ? SPM : 1];
#pragma unroll
            for (int l = 0; l < SPM; ++l) u[l] = zc[p.u_off + l];
            h = zc[p.dt_off];
            SP4_SET_CF(cf, u, mg);
            tab = (sp_cptr)(drift_tab + (p.g0_batch_stride ? (long long)b * SP4NDRIFT : 0));
        };
        // tile -> global, lane = row pair: column cl of the tile is a run of n consecutive doubles at dst + cl * colstride
        auto store_tile = [&](const double *T, double *dst, long long colstride, int ncols) {
            int l0 = lane;
            asm volatile("" : "+v"(l0));
            for (int e2 = l0; e2 < ncols * d; e2 += 64) {
                const int cl = e2 / d, r0 = 2 * (e2 - cl * d);
                const double *src = T + cl * SP4CS + r0;
                store2(dst + (long long)cl * colstride + r0, src[0], src[1], 0);
            }
        };
        if (wave == 0) {
            // ---- P: powers of G(u_k) and the blocks' values ------------------------------------------------------------------
            for (int it = 0; it < n_my; ++it) {
                int c0, nce, k, b;
                decode(it, c0, nce, k, b);
                SP4_LANEPOS();
                const bool act = c < d;
                const unsigned oP = sp4_lds_off(Pt + own), oPx = sp4_lds_off(Pt + oth);
                double h;
                sp4_cf cf;
                sp_cptr tab;
                scalars(k, b, h, cf, tab);
                gave_up = sp4_wait(sync, SP4_F_C, SP4_NSTREAM * it, gave_up);  // the stream waves hold the previous item's values in registers
                if (act) {  // P = B^+ = B^- = I (first d columns: this lane's rows of column c)
#pragma unroll
                    for (int i = 0; i < SPD; ++i) {
                        const double e = (half == 0 && i == c) ? 1.0 : 0.0;
                        Pt[own + i] = e;
                        Bpt[own + i] = e;
                        Bmt[own + i] = e;
                    }
                }
                double hp = 1.0, hm = 1.0;
#pragma unroll 1
                for (int s = 0; s < q; ++s) {
                    double x[SPD];
                    if (act) {
#pragma unroll
                        for (int i = 0; i < SPD; ++i) x[i] = Pt[own + i];
                    }
                    if (act) sp4_product(x, oP, oP, oPx, 0.0, 1.0, half ? -1.0 : 1.0, tab, cf);
                    hp *= h;
                    hm *= -h;
                    const double cp = p.pc[s + 1] * hp, cm = p.pc[s + 1] * hm;
                    if (act) {  // B^+ += c_j h^j P, B^- += c_j (-h)^j P: this lane's rows, nine at a time (all reads of a batch before its writes)
#pragma unroll
                        for (int i0 = 0; i0 < SPD; i0 += 9) {
                            double pv[9], bp[9], bm[9];
#pragma unroll
                            for (int i = 0; i < 9; ++i)
                                if (i0 + i < SPD) {
                                    pv[i] = Pt[own + i0 + i];
                                    bp[i] = Bpt[own + i0 + i];
                                    bm[i] = Bmt[own + i0 + i];
                                }
#pragma unroll
                            for (int i = 0; i < 9; ++i)
                                if (i0 + i < SPD) {
                                    Bpt[own + i0 + i] = __builtin_fma(cp, pv[i], bp[i]);
                                    Bmt[own + i0 + i] = __builtin_fma(cm, pv[i], bm[i]);
                                }
                        }
                    }
                }
                wave_lds_sync();
                sp4_post(sync + SP4_F_B, it + 1, lane);
            }
        } else if (wave <= 2) {
            // ---- W (delta) and V (d delta / dh) --------------------------------------------------------------------------------
            const bool isW = wave == 1;
            double *Xt = isW ? Wt : Vt;
            for (int it = 0; it < n_my; ++it) {
                int c0, nce, k, b;
                decode(it, c0, nce, k, b);
                SP4_LANEPOS();
                const unsigned oX = sp4_lds_off(Xt + own), oXx = sp4_lds_off(Xt + oth);
                const unsigned oD = sp4_lds_off(Dt + own), oS = sp4_lds_off(St + own);
                const bool act = c < nce;
                double h;
                sp4_cf cf;
                sp_cptr tab;
                scalars(k, b, h, cf, tab);
                gave_up = sp4_wait(sync, SP4_F_IN, it + 1, gave_up);
                {  // level q
                    const double *Yq = (q & 1) ? St : Dt;
                    const double aq = ((q & 1) ? -1.0 : 1.0) * p.pc[q] * (isW ? 1.0 : (double)q);
                    if (act) {
                        double y[SPD];
#pragma unroll
                        for (int i = 0; i < SPD; ++i) y[i] = Yq[own + i];
#pragma unroll
                        for (int i = 0; i < SPD; ++i) Xt[own + i] = aq * y[i];
                    }
                    wave_lds_sync();
                    if (isW) sp4_post(sync + SP4_F_W, it * q + 1, lane);
                }
#pragma unroll 1
                for (int s = 0; s < q; ++s) {
                    const int j = q - 1 - s;
                    double x[SPD];
                    if (act) {
#pragma unroll
                        for (int i = 0; i < SPD; ++i) x[i] = Xt[own + i];
                    }
                    if (isW) {  // every drive wave has gathered the level this product overwrites
                        for (int l = 0; l < SPM; ++l) gave_up = sp4_wait(sync, SP4_F_G + l, it * q + s + 1, gave_up);
                    }
                    const double alpha = ((j & 1) ? -1.0 : 1.0) * p.pc[j] * (isW ? 1.0 : (double)j);
                    const double beta = (!isW && j == 0) ? 1.0 : h;
                    if (act) sp4_product(x, (j & 1) ? oS : oD, oX, oXx, alpha, beta, half ? -beta : beta, tab, cf);
                    if (isW && j >= 1) {
                        wave_lds_sync();
                        sp4_post(sync + SP4_F_W, it * q + s + 2, lane);
                    }
                }
                wave_lds_sync();
                sp4_post(sync + (isW ? SP4_F_DW : SP4_F_DV), it + 1, lane);  // this wave's reads of D, S are complete
                const long long bk = (long long)b * p.K + k;
                if (isW) {
                    if (p.delta) store_tile(Xt, p.delta + bk * xd + (long long)c0 * n, n, nce);
                } else {
                    store_tile(Xt, p.jac + bk * p.jac_per + 2 * blk + ((long long)c0 * (m + 1) + m) * n, (long long)(m + 1) * n, nce);
                }
                wave_lds_sync();  // (the tile is rewritten by the next item's level q)
            }
        } else {
            // ---- dW_l (d delta / du_l) ----------------------------------------------------------------------------------------
            const int l = wave - 3;
            double *Xt = dWt + l * SP4TILE;
            for (int it = 0; it < n_my; ++it) {
                int c0, nce, k, b;
                decode(it, c0, nce, k, b);
                SP4_LANEPOS();
                const unsigned oX = sp4_lds_off(Xt + own), oXx = sp4_lds_off(Xt + oth);
                const double sb = half ? 1.0 : -1.0;
                const bool act = c < nce;
                double h;
                sp4_cf cf;
                sp_cptr tab;
                scalars(k, b, h, cf, tab);
                // level q - 1: dW = h G_l W_q
                gave_up = sp4_wait(sync, SP4_F_W, it * q + 1, gave_up);
                if (act) {
                    SP4_GATHER_SWITCH(l, Wt + own, Wt + oth, Xt + own, h, sb, mg)
                }
                wave_lds_sync();
                sp4_post(sync + SP4_F_G + l, it * q + 1, lane);
#pragma unroll 1
                for (int s = 1; s < q; ++s) {
                    double x[SPD];
                    if (act) {
#pragma unroll
                        for (int i = 0; i < SPD; ++i) x[i] = Xt[own + i];
                    }
                    gave_up = sp4_wait(sync, SP4_F_W, it * q + s + 1, gave_up);
                    wave_lds_sync();  // (x is in registers before the gather rewrites the tile)
                    if (act) {
                        SP4_GATHER_SWITCH(l, Wt + own, Wt + oth, Xt + own, h, sb, mg)
                    }
                    wave_lds_sync();
                    sp4_post(sync + SP4_F_G + l, it * q + s + 1, lane);
                    if (act) sp4_product(x, oX, oX, oXx, 1.0, h, half ? -h : h, tab, cf);  // tile = h G_l W_old + h G dW_old
                }
                const long long bk = (long long)b * p.K + k;
                store_tile(Xt, p.jac + bk * p.jac_per + 2 * blk + ((long long)c0 * (m + 1) + l) * n, (long long)(m + 1) * n, nce);
                wave_lds_sync();
            }
        }
    } else if (wave == SP4_WLOAD) {
        // ================================== loader: D, S of every item (lane = row) ============================================
        constexpr int NB = 9;  // columns per batch of loads
        for (int it = 0; it < n_my; ++it) {
            int c0, nce, k, b;
            decode(it, c0, nce, k, b);
            const double *zk = p.Z + (long long)b * p.z_batch_stride + (long long)k * p.z_dim + p.x_offs[p.z_batch_stride ? 0 : b] + (long long)c0 * n + (lane < n ? lane : 0);
            const double *zn = zk + p.z_dim;
            bool first = true;
            for (int cb = 0; cb < nce; cb += NB) {
                double xc[NB], xn[NB];
#pragma unroll
                for (int j = 0; j < NB; ++j) {
                    xc[j] = xn[j] = 0.0;
                    if (cb + j < nce) {
                        xc[j] = zk[(cb + j) * n];
                        xn[j] = zn[(cb + j) * n];
                    }
                }
                if (first) {  // W and V are done with the previous item's D, S
                    gave_up = sp4_wait(sync, SP4_F_DW, it, gave_up);
                    gave_up = sp4_wait(sync, SP4_F_DV, it, gave_up);
                    first = false;
                }
                if (lane < n) {
#pragma unroll
                    for (int j = 0; j < NB; ++j)
                        if (cb + j < nce) {
                            Dt[(cb + j) * SP4CS + lane] = xn[j] - xc[j];
                            St[(cb + j) * SP4CS + lane] = xn[j] + xc[j];
                        }
                }
            }
            wave_lds_sync();
            sp4_post(sync + SP4_F_IN, it + 1, lane);
        }
    } else {
        // ================================== stream waves ========================================================================
        constexpr int hn = n >> 1;
        constexpr int pstep = (64 * SP4_NSTREAM) / hn > 0 ? (64 * SP4_NSTREAM) / hn : 1;
        for (int it = 0; it < n_my; ++it) {
            int c0, nce, k, b;
            decode(it, c0, nce, k, b);
            // (an opaque copy per item: derived from `tid` directly, the 24 tile addresses below are hoisted out of the item loop
            //  and spilled)
            int stid = tid - 64 * SP4_WSTREAM;
            asm volatile("" : "+v"(stid));
            const int pi = 2 * (stid % hn), pj0 = stid / hn;
            const bool pact = pj0 < pstep;
            gave_up = sp4_wait(sync, SP4_F_B, it + 1, gave_up);
            // entry (i, j) of the n x n iso matrix [[A, -B], [B, A]] whose first d columns are a tile: j >= d mirrors into column
            // j - d, rows i < d from row i + d with the sign flipped, rows i >= d from row i - d
            double bpr[PCL_NSP][2], bmr[PCL_NSP][2];
#pragma unroll
            for (int r = 0; r < PCL_NSP; ++r) {
                const int j = min(pj0 + pstep * r, n - 1);
                const bool mir = j >= d;
                const int jj = mir ? j - d : j;
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const int i = pi + e;
                    const int ii = mir ? (i < d ? i + d : i - d) : i;
                    const double sg = (mir && i < d) ? -1.0 : 1.0;
                    bpr[r][e] = -sg * Bpt[jj * SP4CS + ii];
                    bmr[r][e] = sg * Bmt[jj * SP4CS + ii];
                }
            }
            wave_lds_sync();
            sp4_arrive(sync + SP4_F_C, lane);  // the P wave may form the next item's values
            if (pact) {
                int cbeg = c0, cend = c0 + nce;
                if (p.compact) {  // unique blocks only: the piece that holds column 0 writes the single copy
                    cbeg = 0;
                    cend = (c0 == 0) ? 1 : 0;
                }
                double *o = p.jac + ((long long)b * p.K + k) * p.jac_per + (long long)cbeg * nn + pi;
                for (int cq = cbeg; cq < cend; ++cq, o += nn) {
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
        }
    }
    if (gave_up && lane == 0) p.jac[0] = __builtin_nan("");  // a wait gave up: visible in the values instead of a hung device
}
