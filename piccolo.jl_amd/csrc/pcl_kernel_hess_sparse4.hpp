// pcl_kernel_hess_sparse4.hpp -- Hessian of the Lagrangian at ANY diagonal Pade order 2q, PATTERN-COMPILED.
// Included by generated source only (pcl_codegen_v4.hpp with the Hessian functions): SPD, SPM, SPN, SP4Q, the products
// sp4_product_t / sp4_product0_t (G(u)^T x) and sp4_product0 (G(u) x), the gathers sp4_gather_t_<l> (G_l^T w) and the gather-dots
// sp4_gdot_all (<v, G_l z> for every drive l) are defined before this file.
//
// With M = mu_k (n x d), Y_j = (-1)^j X_{k+1} - X_k, T_j = c_j h^j, T'_j = j c_j h^(j-1), T''_j = j (j-1) c_j h^(j-2):
//     W_0 = M, W_j = G^T W_{j-1}                        V_{l,j} = G^T V_{l,j-1} + G_l^T W_{j-1}  (V_{l,0} = 0)        Z_b(Y) = G^b Y
//     d2/du_l dX_k = -sum_j T_j V_{l,j}    d2/du_l dX_{k+1} = sum_j T_j (-1)^j V_{l,j}    d2/dh dX_k = -sum_j T'_j W_j    d2/dh dX_{k+1} = sum_j T'_j (-1)^j W_j
//     (h,h) = sum_j T''_j <W_j, Y_j>       (h,u_l) = sum_j T'_j <V_{l,j}, Y_j>
//     (u_i,u_l) = sum_j T_j sum_{b=0}^{j-2} ( <V_{i,j-1-b}, G_l Z_b(Y_j)> + <V_{l,j-1-b}, G_i Z_b(Y_j)> )
// -- the second-derivative chains U_{il,j} of the oracle's recursion never exist: unrolled, every term of <U_{il,j}, Y_j> is an inner
// product of a FIRST-derivative chain value with a drive applied to a power chain of Y (9q - 10 big products per interval instead of
// 21 (q - 2) more).  One workgroup of m + 1 waves per interval (slices of the state columns where the tiles do not fit LDS):
//     wave 0      W chain; d2/dh dX vectors, (h,h)                (during the Z phase: Z_b(D))
//     wave 1 + l  V_l chain; d2/du_l dX vectors, (h,u_l), row l of the (u,u) sums   (wave 1 during the Z phase: Z_b(S))
// The output vectors of a chain accumulate in registers (108 per chain; m + 1 waves, two per SIMD: 256 registers per lane).
// Lock step behind workgroup barriers (two per level): every chain value lives in its LDS tile [column][row].  Scalar entries: per-lane partial sums, one wave reduction per
// interval, a fixed order of additions: bitwise repeatable.
#pragma once

#define SP4CS (SPN + 1)
#ifndef SH_VARIANT
#define SH_VARIANT 0
#endif
#ifndef SH_SPLIT
#define SH_SPLIT 1                               // workgroups per interval: each takes SH_MH of the drives (and its own copy of the W and power chains)
#endif
#define SH_MH ((SPM + SH_SPLIT - 1) / SH_SPLIT)  // drives per workgroup
#define SH_NW (SH_MH + 1)                        // the W wave + one wave per drive (two drive chains per wave, four waves with the spills in
                                                 // the accumulator registers of the unified file, measured 2.6x slower)
#define SH_NZ (SP4Q > 2 ? SP4Q - 2 : 0)        // stored power-chain levels beyond Y itself
#define SH_XR (SP4Q == 2 ? 1 : 0)              // order 4: the combined power-chain tile R has a tile of its own (above: the unused top-level tile)
#define SH_NTILES (SH_MH + 3 + 2 * SH_NZ + SH_XR)  // W, V[SH_MH], D, S, ZD[SH_NZ], ZS[SH_NZ] (+ R)
#define SH_NSC ((SPM + 1) * (SPM + 2) / 2)

static __device__ __forceinline__ unsigned sp4_lds_off(const double *q) {
    return (unsigned)(size_t)(__attribute__((address_space(3))) const double *)q;
}
template <int V> struct sp4_ic { static constexpr int value = V; };
template <int J, int N, class F>
static __device__ __forceinline__ void sp4_static_for(F f) {
    if constexpr (J < N) {
        f(sp4_ic<J>{});
        sp4_static_for<J + 1, N>(f);
    }
}

// Exchange between the SH_SPLIT workgroups of an interval (they may run on different XCDs, whose L2s are not coherent with each other):
// write-through stores and cache-bypassing loads at system scope; no fence -- a release would write the workgroup's 160 KB of
// output vectors back from L2 first (measured in round 2: 20-60 % of a launch).
static __device__ __forceinline__ void sh_store_coherent(double *q, double v) { asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1" ::"v"(q), "v"(v) : "memory"); }
static __device__ __forceinline__ double sh_load_coherent(const double *q) {
    double v;
    asm volatile("global_load_dwordx2 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(q) : "memory");
    return v;
}

extern "C" __global__ __launch_bounds__(64 * SH_NW) void pcl_hess_sparse4_kernel(const KParams p, const double *__restrict__ drift_tab, const double *__restrict__ drift_tab_t,
                                                                                const double *__restrict__ mags_, const double *__restrict__ dcf_tab,
                                                                                double *xch /* SH_SPLIT > 1: [interval][workgroup][(SH_MH + 1)(m + 2)] reduced sums */,
                                                                                unsigned int *xcnt /* ... [interval] arrivals (self-resetting) */) {
    extern __shared__ double lds[];
    constexpr int d = SPD, n = SPN, m = SPM, q = SP4Q;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nc = p.nc;                       // state columns per slice (host: the tiles fit LDS)
    const int TS = nc * SP4CS;                 // doubles per tile
    double *Wt = lds, *Vt = Wt + TS, *Dt = Vt + SH_MH * TS, *St = Dt + TS, *ZDt = St + TS, *ZSt = ZDt + SH_NZ * TS;
    // R_j = sum_b w(j + b + 1) Z_b(|Y_{j+b+1}|): every (u,u) term of chain level j is <V_{.,j}, G_i R_j> -- the sum over the power-chain
    // levels b is taken ONCE per level (wave 0), not once per drive pair.  Its tile: Z_{q-2} is needed for Y_q only, i.e. for one
    // parity -- the other top-level tile is never formed.
    double *Rt = q >= 3 ? ((q & 1) ? ZDt : ZSt) + (SH_NZ - 1) * TS : ZSt + SH_NZ * TS;
    double *scal = ZSt + (SH_NZ + SH_XR) * TS;  // [m + 1][m + 2] reduced sums of the interval (slot 0: the W chain, slot 1 + l: drive l)
    const long long xd = (long long)n * d;
    sp_cptr magc = (sp_cptr)mags_;
    double mg[SP4NMAG];
#pragma unroll
    for (int g = 0; g < SP4NMAG; ++g) mg[g] = magc[g];

#ifdef PCL_PROFILE
    // 100 MHz wall stamps per workgroup (dbg[512 + 768 sel + 3 bx + {entry, -, last wave out}], sel = prof & 64): the launch-to-launch period split
    long long *wall_ = p.dbg ? p.dbg + 512 + ((p.prof & 64) ? 768 : 0) + 3 * (blockIdx.x & 255) : nullptr;
    if (wall_ && tid == 0) wall_[0] = (long long)__builtin_amdgcn_s_memrealtime();
    int stamp_ = 0;  // cycle stamps of workgroup 0, first 32 per wave (dbg[32 wave + i])
#define SH_STAMP()                                                                                                           \
    do {                                                                                                                     \
        if (p.dbg && blockIdx.x == 0 && lane == 0 && stamp_ < 32) p.dbg[32 * wave + stamp_++] = (long long)__builtin_amdgcn_s_memtime(); \
    } while (0)
#else
#define SH_STAMP() do { } while (0)
#endif
    const int n_units = p.batch * p.K * SH_SPLIT;  // (interval, drive group)
    const int unit_lo = (int)((long long)n_units * blockIdx.x / gridDim.x), unit_hi = (int)((long long)n_units * (blockIdx.x + 1) / gridDim.x);
    const int S = (d + nc - 1) / nc;
    for (int unit = unit_lo; unit < unit_hi; ++unit) {
        const int item = unit / SH_SPLIT, grp = unit - item * SH_SPLIT;
        const int dl = grp * SH_MH + wave - 1;          // this wave's drive (wave 0: the W chain)
        const bool drv = wave > 0 && dl < m;            // (m odd: the last workgroup has a wave without a drive)
        const bool wout = grp == 0;                     // the W chain's outputs leave through the first workgroup
        const int k = item % p.K, b = item / p.K;
        const long long bk = item;
        double *H = p.hess + bk * p.hess_per;
        // scalars of the interval
        sp_cptr zc = (sp_cptr)(p.Z + (long long)b * p.z_batch_stride + (long long)k * p.z_dim);
        double u[SPM > 0 ? SPM : 1];
#pragma unroll
        for (int l = 0; l < SPM; ++l) u[l] = zc[p.u_off + l];
        const double h = zc[p.dt_off];
        sp4_cf cf;
        SP4_SET_CF(cf, u, mg);
        SP4_SET_DCF(cf, (sp_cptr)(dcf_tab + (p.g0_batch_stride ? (long long)b * SP4NDCFP : 0)));
        sp_cptr tab = (sp_cptr)(drift_tab + (p.g0_batch_stride ? (long long)b * SP4NDRIFT : 0));
        sp_cptr tab_t = (sp_cptr)(drift_tab_t + (p.g0_batch_stride ? (long long)b * SP4NDRIFT : 0));
        // per-lane partial sums of the interval (all slices): s_y = <chain, Y> sums ((h,h) for wave 0, (h,u_l) for a drive wave), s_uu[i] = row l of the (u,u) sums
        double s_y = 0.0, s_uu[SPM > 0 ? SPM : 1];
#pragma unroll
        for (int i = 0; i < SPM; ++i) s_uu[i] = 0.0;

        unsigned xold_ = 0xffffffffu;  // (two workgroups per interval: what the interval's arrival counter held before this workgroup's increment)
        for (int sl = 0; sl < S; ++sl) {
            const int c0 = sl * nc, nce = min(nc, d - c0);
            int ln_ = lane;
            asm volatile("" : "+v"(ln_));
            const int half = ln_ >> 5, c = ln_ & 31;
            const bool act = c < nce;
            const int own = (act ? c : 0) * SP4CS + half * d, oth = (act ? c : 0) * SP4CS + (1 - half) * d;
            // ---- inputs: M -> W tile, D, S (lane = row, one column per load; the waves share the columns) ---------------------
            {
                const double *zk = p.Z + (long long)b * p.z_batch_stride + (long long)k * p.z_dim + (p.x_off0 >= 0 ? p.x_off0 : p.x_offs[p.z_batch_stride ? 0 : b]) + (long long)c0 * n + (ln_ < n ? ln_ : 0);
                const double *zn = zk + p.z_dim;
                const double *mu = p.mu + bk * xd + (long long)c0 * n + (ln_ < n ? ln_ : 0);
                for (int cl = wave; cl < nce; cl += SH_NW) {
                    const double xc = zk[cl * n], xn = zn[cl * n], mv = mu[cl * n];
                    if (ln_ < n) {
                        Wt[cl * SP4CS + ln_] = mv;
                        Dt[cl * SP4CS + ln_] = xn - xc;
                        St[cl * SP4CS + ln_] = xn + xc;
                    }
                }
            }
            SH_STAMP();
            __syncthreads();
            SH_STAMP();
            // ---- Z phase: Z_b(D) = G Z_{b-1}(D) by wave 0, Z_b(S) by wave 1 ---------------------------------------------------
#pragma unroll 1
            for (int bb = 1; bb <= SH_NZ; ++bb) {
                if (wave < 2 && !(bb == SH_NZ && wave == ((q & 1) ? 0 : 1))) {  // (top level: the parity of Y_q only)
                    const double *src = wave == 0 ? (bb == 1 ? Dt : ZDt + (bb - 2) * TS) : (bb == 1 ? St : ZSt + (bb - 2) * TS);
                    double *dst = (wave == 0 ? ZDt : ZSt) + (bb - 1) * TS;
                    double x[SPD];
                    if (act) {
#pragma unroll
                        for (int i = 0; i < SPD; ++i) x[i] = src[own + i];
                        sp4_product0(x, 0u, sp4_lds_off(dst + own), sp4_lds_off(dst + oth), 0.0, 1.0, half ? -1.0 : 1.0, tab, cf);
                    }
                }
                __syncthreads();
            }
            SH_STAMP();
            // ---- V phase ------------------------------------------------------------------------------------------------------
            double *Xt = wave == 0 ? Wt : Vt + (wave - 1) * TS;   // this wave's chain tile
            const unsigned oX = sp4_lds_off(Xt + own), oXx = sp4_lds_off(Xt + oth);
            double accK[SPD], accN[SPD];  // the two output vectors of this wave (X_k / X_{k+1} blocks), this lane's rows
#pragma unroll
            for (int i = 0; i < SPD; ++i) accK[i] = accN[i] = 0.0;
            const double bt = half ? 1.0 : -1.0;  // G^T: the other half receives -V from half 0, +V from half 1
            // R_j = sum_b w(j + b + 1) Z_b(|Y_{j+b+1}|), this lane's half rows of its column (wave 0), level by level.  (With two workgroups per
            // interval the W wave -- its product AND this sum -- is the last to reach every level's barrier; issuing every read of the sum up
            // front, and sharing the first W product with the idle drive waves in row ranges, both measured SLOWER: 34.7 -> 37.0 / 39.0 us.)
            auto combine_R = [&](int j, double hp_) {
                double r[SPD];
#pragma unroll
                for (int i = 0; i < SPD; ++i) r[i] = 0.0;
                double hb = hp_ * h * h;  // h^jj for b = 0
#pragma unroll 1
                for (int bb = 0; j + bb + 1 <= q; ++bb, hb *= h) {
                    const int jj = j + bb + 1;
                    const double *Zt = ((jj & 1) ? (bb == 0 ? St : ZSt + (bb - 1) * TS) : (bb == 0 ? Dt : ZDt + (bb - 1) * TS)) + own;
                    const double wz = ((jj & 1) ? -1.0 : 1.0) * p.pc[jj] * hb;
#pragma unroll
                    for (int i = 0; i < SPD; ++i) r[i] = __builtin_fma(wz, Zt[i], r[i]);
                }
#pragma unroll
                for (int i = 0; i < SPD; ++i) Rt[own + i] = r[i];
            };
            double hp = 1.0;                      // h^(j-1)
#pragma unroll 1
            for (int j = 1; j <= q; ++j) {
                if (wave > 0) {  // V_{l,j} = G^T V_{l,j-1} + G_l^T W_{j-1}
                    const bool on = act && drv;
                    double x[SPD];
                    if (j > 1 && on) {
#pragma unroll
                        for (int i = 0; i < SPD; ++i) x[i] = Xt[own + i];
                    }
                    wave_lds_sync();
                    if (on) {
                        SP4_GATHER_T_SWITCH(dl, Wt + own, Wt + oth, Xt + own, 1.0, (half ? -1.0 : 1.0), mg)
                    }
                    wave_lds_sync();
                    SH_STAMP();
                    __syncthreads();  // every drive wave has read W_{j-1}
                    SH_STAMP();
                    if (j > 1 && on) sp4_product_t(x, oX, oX, oXx, 1.0, 1.0, bt, tab_t, cf);
                } else {
                    __syncthreads();
                    double x[SPD];
                    if (act) {
#pragma unroll
                        for (int i = 0; i < SPD; ++i) x[i] = Xt[own + i];
                        sp4_product0_t(x, 0u, oX, oXx, 0.0, 1.0, bt, tab_t, cf);  // W_j = G^T W_{j-1}, in place
                    }
                    // R_j, behind the barrier (until there the drive waves may still be reading R_{j-1}) and behind the product (formed in
                    // registers ahead of the barrier, its 54 registers beside the output vectors cost 57 more spills: measured 16 % slower)
                    if (act && j < q) combine_R(j, hp);
                }
                SH_STAMP();
                __syncthreads();  // W_j and every V_{l,j} are in their tiles
                SH_STAMP();
                // ---- what level j contributes -------------------------------------------------------------------------------
                const double cj = p.pc[j];
                const double Tj = cj * hp * h, T1 = j * cj * hp, sg = (j & 1) ? -1.0 : 1.0;
                if (act && (wave == 0 ? wout : drv)) {
                    // (memory clobbers between the stages: left alone, the compiler hoists every LDS read of the level to its top --
                    //  the Y column and the Z columns of six drives next to v and the two output vectors: hundreds of spills)
                    double v[SPD];
#pragma unroll
                    for (int i = 0; i < SPD; ++i) v[i] = Xt[own + i];
                    const double wK = wave == 0 ? T1 : Tj;  // the weight of this level in the wave's output vectors
#pragma unroll
                    for (int i = 0; i < SPD; ++i) {
                        accK[i] = __builtin_fma(-wK, v[i], accK[i]);
                        accN[i] = __builtin_fma(wK * sg, v[i], accN[i]);
                    }
                    asm volatile("" ::: "memory");
                    const double *Yj = ((j & 1) ? St : Dt) + own;  // Y_j = D (j even) | -S (j odd)
                    double dot0 = 0.0, dot1 = 0.0;
#pragma unroll
                    for (int i0 = 0; i0 < SPD; i0 += 9) {
                        double y[9];
#pragma unroll
                        for (int i = 0; i < 9; ++i)
                            if (i0 + i < SPD) y[i] = Yj[i0 + i];
#pragma unroll
                        for (int i = 0; i < 9; ++i)
                            if (i0 + i < SPD) {
                                if (i & 1)
                                    dot1 = __builtin_fma(v[i0 + i], y[i], dot1);
                                else
                                    dot0 = __builtin_fma(v[i0 + i], y[i], dot0);
                            }
                        asm volatile("" ::: "memory");
                    }
                    const double dy = sg * (dot0 + dot1);  // <chain_j, Y_j>
                    if (wave == 0) {
                        if (j >= 2) s_y = __builtin_fma(j * (j - 1) * cj * (hp / h), dy, s_y);  // T''_j = j (j-1) c_j h^(j-2)
                    } else {
                        s_y = __builtin_fma(T1, dy, s_y);
                        // (u,u): <V_{l,j}, G_i R_j>, every drive i
                        if (j < q) {
                            double r6[SPM > 0 ? SPM : 1];
                            sp4_gdot_all(Rt + own, Rt + oth, v, (half ? 1.0 : -1.0), mg, r6);
#pragma unroll
                            for (int i = 0; i < SPM; ++i) s_uu[i] += r6[i];
                        }
                    }
                }
                SH_STAMP();
                hp *= h;
            }
#if SH_SPLIT > 1
            // Two workgroups per interval exchange their rows of reduced sums through memory (see below).  The rows leave BEFORE the output
            // vectors (last slice): store, acknowledged, counter bumped -- two of the exchange's three memory round trips then run under the
            // output stage instead of behind it.
            if (sl == S - 1) {
                const double ry = wave_sum(s_y);
                if (lane == 0) scal[wave * (m + 2)] = ry;
#pragma unroll
                for (int i = 0; i < SPM; ++i) {
                    const double r = wave_sum(s_uu[i]);
                    if (lane == 0) scal[wave * (m + 2) + 1 + i] = r;
                }
            }
#endif
            __syncthreads();  // the chain tiles are free: the output vectors leave through them
#if SH_SPLIT > 1
            if (sl == S - 1 && wave == 0) {
                constexpr int XS = (SH_MH + 1) * (m + 2);
                double *xrow = xch + ((long long)item * SH_SPLIT + grp) * XS;
                if (lane < XS) sh_store_coherent(xrow + lane, scal[lane]);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (lane == 0) xold_ = __hip_atomic_fetch_add(xcnt + item, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
#endif
            // ---- output vectors: registers -> this wave's tile -> memory (lane = row, one column per store) -----------------
            {
                double *o1, *o2;  // the wave's X_k / X_{k+1} blocks
                if (wave == 0) {
                    o1 = H + SH_NSC + (long long)m * xd;
                    o2 = H + SH_NSC + (long long)(2 * m + 1) * xd;
                } else {
                    o1 = H + SH_NSC + (long long)dl * xd;
                    o2 = H + SH_NSC + (long long)(m + 1 + dl) * xd;
                }
                const bool outw = wave == 0 ? wout : drv;  // (uniform per wave)
#pragma unroll 1
                for (int pass = 0; pass < 2 && outw; ++pass) {
                    if (act) {
#pragma unroll
                        for (int i = 0; i < SPD; ++i) Xt[own + i] = pass ? accN[i] : accK[i];
                    }
                    wave_lds_sync();
                    if (ln_ < n) {
                        double *ol = (pass ? o2 : o1) + (long long)c0 * n + ln_;
                        const double *Tl = Xt + ln_;
#if defined(SH_VARIANT) && (SH_VARIANT & 16)
                        for (int cl = 0; cl < nce; ++cl) ol[cl * n] = Tl[cl * SP4CS];
#else
                        for (int cl0 = 0; cl0 < nce; cl0 += 9) {  // nine columns per batch: one LDS round trip, not nine
                            double t_[9];
#pragma unroll
                            for (int u_ = 0; u_ < 9; ++u_) t_[u_] = Tl[min(cl0 + u_, nce - 1) * SP4CS];
#pragma unroll
                            for (int u_ = 0; u_ < 9; ++u_)
                                if (cl0 + u_ < nce) ol[(cl0 + u_) * n] = t_[u_];
                        }
#endif
                    }
                    wave_lds_sync();
                }
            }
            SH_STAMP();
            __syncthreads();  // (the tiles are reloaded by the next slice / interval)
        }
        // ---- scalar entries of the interval: one reduction per wave, then a fixed assembly ----------------------------------------
#if SH_SPLIT == 1
        {
            const double ry = wave_sum(s_y);
            if (lane == 0) scal[wave * (m + 2)] = ry;
#pragma unroll
            for (int i = 0; i < SPM; ++i) {
                const double r = wave_sum(s_uu[i]);
                if (lane == 0) scal[wave * (m + 2) + 1 + i] = r;
            }
        }
        __syncthreads();
#else
        // the workgroups of the interval have exchanged their rows of reduced sums through memory; the one that arrived last assembles the
        // entries.  Nobody waits for anybody: no assumption on which workgroups are resident together.
        {
            constexpr int XS = (SH_MH + 1) * (m + 2);
            if (wave == 0) {
                unsigned old_ = __builtin_amdgcn_readfirstlane(xold_);
                if (old_ == SH_SPLIT - 1) {  // every workgroup of the interval has stored its rows
                    if (lane == 0) __hip_atomic_store(xcnt + item, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // (for the next launch)
                    const double *xall = xch + (long long)item * SH_SPLIT * XS;
                    for (int e = lane; e < (m + 1) * (m + 2); e += 64) {  // slot 0: the W chain (first workgroup); slot 1 + l: drive l
                        const int slot = e / (m + 2), col = e - slot * (m + 2);
                        const int g_ = slot == 0 ? 0 : (slot - 1) / SH_MH, r_ = slot == 0 ? 0 : 1 + (slot - 1) % SH_MH;
                        scal[e] = sh_load_coherent(xall + (long long)g_ * XS + r_ * (m + 2) + col);
                    }
                    wave_lds_sync();
                } else {
                    old_ = 0xffffffffu;
                }
                if (old_ != 0xffffffffu && lane < SH_NSC) {
                    // order: (u_i, u_j) for i = 0..m-1, j = 0..i | (h, u_j) j < m | (h, h)
                    double v;
                    if (lane < m * (m + 1) / 2) {
                        int i = 0;
                        while ((i + 1) * (i + 2) / 2 <= lane) ++i;
                        const int j = lane - i * (i + 1) / 2;
                        v = scal[(1 + i) * (m + 2) + 1 + j] + scal[(1 + j) * (m + 2) + 1 + i];  // S[i][j] + S[j][i]
                    } else if (lane < m * (m + 1) / 2 + m) {
                        v = scal[(1 + lane - m * (m + 1) / 2) * (m + 2)];
                    } else {
                        v = scal[0];
                    }
                    H[lane] = v;
                }
            }
        }
#endif
#if SH_SPLIT == 1
        if (wave == 0 && lane < SH_NSC) {
            // order: (u_i, u_j) for i = 0..m-1, j = 0..i | (h, u_j) j < m | (h, h)
            double v;
            if (lane < m * (m + 1) / 2) {
                int i = 0;
                while ((i + 1) * (i + 2) / 2 <= lane) ++i;
                const int j = lane - i * (i + 1) / 2;
                v = scal[(1 + i) * (m + 2) + 1 + j] + scal[(1 + j) * (m + 2) + 1 + i];  // S[i][j] + S[j][i]
            } else if (lane < m * (m + 1) / 2 + m) {
                v = scal[(1 + lane - m * (m + 1) / 2) * (m + 2)];
            } else {
                v = scal[0];
            }
            H[lane] = v;
        }
#endif
        __syncthreads();
    }
#ifdef PCL_PROFILE
    if (wall_ && lane == 0) atomicMax((unsigned long long *)(wall_ + 2), (unsigned long long)__builtin_amdgcn_s_memrealtime());
#endif
}
