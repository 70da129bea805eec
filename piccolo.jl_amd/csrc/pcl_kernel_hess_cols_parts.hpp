// pcl_kernel_hess_cols_parts.hpp -- the phases of a pass of the column-group Hessian kernels as TEXT, included into the body of pcl_hess_cols_kernel (one wave does
// everything) and of pcl_hess_cols_pair_kernel (a chain wave and a contribution wave per column group: one trajectory per launch) -- one source for both, so that
// the two kernels run the same arithmetic in the same order (their results are bitwise equal: tests/test_round6_gpu.py).  Not a header in the usual sense: the
// includer defines which part it wants (HC_PART_*) and the names the part uses (lane mapping, tiles, x, accumulators ...: see pcl_kernel_hess_cols.hpp).
#if defined(HC_PART_RCHAIN)
        HC_MARK("rsetup");
        const bool ract = s < HC_CPW && s < nce;  // (slot = column)
        const int rb = (s < HC_CPW ? s : 0) * SP4CS;
#pragma unroll 1
        for (int a = q - 2; a >= 1; --a) {
            HC_MARK("rchain");
            if (ract) {
                double x[SPD];  // R_{a+1}
                if (a == q - 2) {  // R_{q-1} = +-T_q |Y_q|
                    const double *src = ((q & 1) ? St : Dt) + rb + own;
                    const double wq = wgt(q);
#pragma unroll
                    for (int i = 0; i < SPD; ++i) x[i] = wq * src[i];
                } else {
                    const double *src = Rt + a * HC_RSTRIDE + rb + own;
#pragma unroll
                    for (int i = 0; i < SPD; ++i) x[i] = src[i];
                }
                double *dst = Rt + (a - 1) * HC_RSTRIDE + rb;
                const double *Y = (((a + 1) & 1) ? St : Dt) + rb + own;
                sp4_product(x, hc_lds_off(Y), hc_lds_off(dst + own), hc_lds_off(dst + oth), sp4_uniform(wgt_at(a + 1)), 1.0, half ? -1.0 : 1.0, tab, cf);
            }
            HC_STAMP();
        }
#elif defined(HC_PART_GROUPSUM)
    constexpr bool HC_GROUPSUM = HC_CPW == 4 && HC_ROW <= 2 * HC_CPW;
    constexpr int HC_NACC = HC_GROUPSUM ? (HC_ROW + HC_CPW - 1) / HC_CPW : HC_ROW;
    double s_acc[HC_NACC];
#pragma unroll
    for (int i = 0; i < HC_NACC; ++i) s_acc[i] = 0.0;
    auto group_sum = [&](double v) {  // sum over the HC_CPW column lanes of this lane's (half, chain), the same in each of them
        if constexpr (HC_CPW >= 2) {
            const int lo = __double2loint(v), hi = __double2hiint(v);
            v += __hiloint2double(__builtin_amdgcn_mov_dpp(hi, 0xB1, 0xf, 0xf, true), __builtin_amdgcn_mov_dpp(lo, 0xB1, 0xf, 0xf, true));  // quad_perm [1,0,3,2]
        }
        if constexpr (HC_CPW >= 4) {
            const int lo = __double2loint(v), hi = __double2hiint(v);
            v += __hiloint2double(__builtin_amdgcn_mov_dpp(hi, 0x4E, 0xf, 0xf, true), __builtin_amdgcn_mov_dpp(lo, 0x4E, 0xf, 0xf, true));  // quad_perm [2,3,0,1]
        }
        return v;
    };
#elif defined(HC_PART_GATHER)
        if (on && isV) {  // + G_l^T W_{jp-1}
#if HC_SWITCH_GATHER
            SP4_GATHER_T_SWITCH(ch - 1, Wc + own, Wc + oth, Xs + own, 1.0, (half ? -1.0 : 1.0), mg)
#pragma unroll
            for (int i = 0; i < SPD; ++i) x[i] += Xs[own + i];
#else
            // the lanes of a wave belong to different drives: the entries come from the table (one instruction stream for every drive).
            // HC_GCH rows at a time, staged by hand -- every entry of the batch, then every operand, then the sums: left to itself the
            // compiler keeps two or three rows in flight and the wave waits out an LDS round trip per row (6.3 k cycles per level)
#ifndef HC_GT_WORD  // (where a lane's table words come from: the table in LDS -- or registers the includer has filled)
#define HC_GT_WORD(w) gt[w]
#define HC_GT_WORD_DEFAULTED
#endif
            const unsigned *gt = gtab + ((ch - 1) * 2 + half) * HC_GT_WPC;
            (void)gt;
#pragma unroll
            for (int i0 = 0; i0 < SPD; i0 += HC_GCH) {
                unsigned e_[HC_GCH][SP4_GTK];
#pragma unroll
                for (int i = 0; i < HC_GCH; ++i)
#pragma unroll
                    for (int kk = 0; kk < SP4_GTK; ++kk) {
                        const int row = i0 + i < SPD ? i0 + i : SPD - 1;
                        const int en = sp4_gt_off(row) + (kk < sp4_gt_cnt(row) ? kk : 0);  // entry number: word en / 3, bits 10 (en % 3) ...
#ifdef HC_GT_PREDECODED
                        e_[i][kk] = 0u;
                        (void)en;
#else
                        e_[i][kk] = kk < sp4_gt_cnt(row) ? HC_GT_WORD(en / 3) : 0u;                 // (the dword; the fields come out of it with one v_bfe_u32 each)
#endif
                    }
                double w_[HC_GCH][SP4_GTK], c_[HC_GCH][SP4_GTK];
#pragma unroll
                for (int i = 0; i < HC_GCH; ++i)
#pragma unroll
                    for (int kk = 0; kk < SP4_GTK; ++kk)
                        if (kk < sp4_gt_cnt(i0 + i < SPD ? i0 + i : SPD - 1)) {  // (a row takes as many terms as the drive with the most there)
                            const int row = i0 + i < SPD ? i0 + i : SPD - 1;
#ifdef HC_GT_PREDECODED  // the includer holds every entry's two addresses in a register: pk_[entry] = W row << 3 | (coefficient's address in LDS) << 16
                            const unsigned pk = pk_[sp4_gt_off(row) + kk];
                            w_[i][kk] = *(const __attribute__((address_space(3))) double *)(size_t)((pk & 0xffffu) + wcol_off);
                            c_[i][kk] = *(const __attribute__((address_space(3))) double *)(size_t)(pk >> 16);
#else
                            const unsigned sh = 10u * (unsigned)((sp4_gt_off(row) + kk) % 3);
                            // (addresses as integers: the W column's base has no bit below 512, the tables' bases are constants of the layout)
                            unsigned aw;  // the W row's address: (bits sh + 4 .. sh + 9) << 3 + the column's
                            asm("v_bfe_u32 %0, %1, %2, 6\n\tv_lshl_add_u32 %0, %0, 3, %3" : "=&v"(aw) : "v"(e_[i][kk]), "n"(sh + 4u), "v"(wcol_off));
                            const unsigned fc = sh >= 3u ? (e_[i][kk] >> (sh - 3u)) & 0x78u : (e_[i][kk] << (3u - sh)) & 0x78u;  // coefficient index << 3
                            w_[i][kk] = *(const __attribute__((address_space(3))) double *)(size_t)aw;
                            c_[i][kk] = *(const __attribute__((address_space(3))) double *)(size_t)(fc + cft_off);
#endif
                        }
                asm volatile("" ::: "memory");
#pragma unroll
                for (int i = 0; i < HC_GCH; ++i)
                    if (i0 + i < SPD) {
#pragma unroll
                        for (int kk = 0; kk < SP4_GTK; ++kk)
                            if (kk < sp4_gt_cnt(i0 + i)) x[i0 + i] = __builtin_fma(c_[i][kk], w_[i][kk], x[i0 + i]);
                    }
            }
#endif
        }
#ifdef HC_GT_WORD_DEFAULTED
#undef HC_GT_WORD
#undef HC_GT_WORD_DEFAULTED
#endif
#elif defined(HC_PART_CONTRIB)
        const int jl = isV ? jp : jp - 1;
        const double cj = isV ? p.pc[jp <= q ? jp : q] : p.pc[jp - 1], hp = isV ? hpV : hpW;
        const double Tj = cj * hp * h, T1 = jl * cj * hp, sg = (jl & 1) ? -1.0 : 1.0;
        double cv[HC_ROW];
#pragma unroll
        for (int v = 0; v < HC_ROW; ++v) cv[v] = 0.0;
        // (the pair kernel splits a level's contributions over its two waves: HC_CONTRIB_NO_ACC / _NO_Y / _NO_G leave a piece out, HC_SUM_V0 .. HC_SUM_V1 are the sums taken)
#ifndef HC_SUM_V0
#define HC_SUM_V0 0
#define HC_SUM_V1 HC_ROW
#define HC_SUM_DEFAULTED
#endif
        if (on) {
#ifndef HC_CONTRIB_NO_ACC
            const double wK = isV ? Tj : T1;  // the weight of this level in the lane's output vectors
            if constexpr (odd_pass) {
#pragma unroll
                for (int i = 0; i < SPD; ++i) accB[i] = __builtin_fma(wK, x[i], accB[i]);
            } else {
#pragma unroll
                for (int i = 0; i < SPD; ++i) accA[i] = __builtin_fma(wK, x[i], accA[i]);
            }
            asm volatile("" ::: "memory");
#endif
#ifndef HC_CONTRIB_NO_Y
            const double *Yj = ((jl & 1) ? St : Dt) + cb + own;  // Y_j = D (j even) | -S (j odd)
            double dot0 = 0.0, dot1 = 0.0;
#pragma unroll
            for (int i0 = 0; i0 < SPD; i0 += 9) {  // (nine rows at a time: the whole column next to x and the output vectors spills the scalar sums)
                double y[9];
#pragma unroll
                for (int i = 0; i < 9; ++i)
                    if (i0 + i < SPD) y[i] = Yj[i0 + i];
#pragma unroll
                for (int i = 0; i < 9; ++i)
                    if (i0 + i < SPD) {
                        if (i & 1)
                            dot1 = __builtin_fma(x[i0 + i], y[i], dot1);
                        else
                            dot0 = __builtin_fma(x[i0 + i], y[i], dot0);
                    }
                asm volatile("" ::: "memory");
            }
            const double dy = sg * (dot0 + dot1);  // <chain_j, Y_j>
            cv[0] = !isV ? (jl >= 2 ? jl * (jl - 1) * cj * hpW2 * dy : 0.0)  // T''_j = j (j-1) c_j h^(j-2)
                         : T1 * dy;
#endif
#ifndef HC_CONTRIB_NO_G
            if (isV && jp < q) {  // (u,u): <V_{l,j}, G_i R_j>, every drive i
                // R_jp; the top one is +-T_q |Y_q|: the D or the S tile, the number applied to the sums
                HC_RTILE_WAIT(jp)  // (where R-chain waves deliver the tiles: which tile, and that its copy has landed)
                const double *Rj = (jp == q - 1 ? ((q & 1) ? St : Dt) : HC_RTILE(jp)) + cb;
                const double wr = jp == q - 1 ? wgt(q) : 1.0;
                double r6[SPM];
                // (the magnitudes from the scalar cache again -- ten scalar registers that need not live across the product, whose ~100 scalar operands fill the file:
                //  held, they pushed 18 coefficient pairs into vector-register lanes, read back with v_readlane in front of every product)
                double mgl_[SP4NMAG];
#pragma unroll
                for (int g = 0; g < SP4NMAG; ++g) mgl_[g] = magc[g];
                sp4_gdot_all(Rj + own, Rj + oth, x, (half ? 1.0 : -1.0), mgl_, r6);
#pragma unroll
                for (int i = 0; i < SPM; ++i) cv[1 + i] = wr * r6[i];
            }
#endif
        }
        HC_RTILE_NEXT(jp);
        HC_MARK("pass_sums");
        // this pass's 1 + m values of the chain, summed over its columns; lane `col` adds the values col, col + HC_CPW, ... to its running totals
        // (every lane of the wave takes part in the DPP steps: lanes without a level contribute zeros)
#pragma unroll
        for (int v = HC_SUM_V0; v < HC_SUM_V1; ++v) {
            if constexpr (HC_GROUPSUM) {
                const double tot = group_sum(cv[v]);
                if (col == v % HC_CPW) s_acc[v / HC_CPW] += tot;
            } else
                s_acc[v] += cv[v];
        }
        hpW2 = hpW;
        hpW = hpV;
        hpV *= h;
#ifdef HC_SUM_DEFAULTED
#undef HC_SUM_V0
#undef HC_SUM_V1
#undef HC_SUM_DEFAULTED
#endif
#elif defined(HC_PART_TAIL)
    // ---- the reduced sums of the wave: every lane parks its 1 + m sums in its chain slot (rows 0 .. m of its half; slots of columns
    //      past the end hold zeros); lane e < HC_XS adds the 2 HC_CPW parts of (chain, value) in a fixed order ------------------------------
    asm volatile("" ::: "memory");
    if (inr) {  // lane (half, chain, col) parks the totals of the values col, col + HC_CPW, ... in rows 0, 1, ... of its half of its chain slot
#pragma unroll
        for (int i = 0; i < HC_NACC; ++i) Xs[own + i] = s_acc[i];
    }
    asm volatile("" ::: "memory");
    unsigned xold = 0xffffffffu;
    {
        if (ln_ < HC_XS) {  // lane = (chain, value): the two halves' totals, top first (without the per-pass sums: the 2 HC_CPW parts, column by column)
            const int chn = ln_ / HC_ROW, val = ln_ - chn * HC_ROW;
            double r = 0.0;
            if constexpr (HC_GROUPSUM) {
                const double *sl_ = chain_col(chn, val % HC_CPW) + val / HC_CPW;
                r = sl_[0] + sl_[d];
            } else {
#pragma unroll
                for (int cc = 0; cc < HC_CPW; ++cc) {
                    r += chain_col(chn, cc)[val];
                    r += chain_col(chn, cc)[d + val];
                }
            }
            hc_store_coherent(xch + ((long long)item * HC_NG + grp) * HC_XS + ln_, r);
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    HC_STAMP();
    // ---- output vectors: registers -> the lane's chain slot -> memory (lane = element: a chain's HC_CPW columns are contiguous) ----------
    // (lane = a PAIR of elements: 16-byte stores -- a wave's 8-byte stores are bound by their issue, 100 cycles each)
    typedef double hc_d2 __attribute__((ext_vector_type(2)));
    typedef double hc_d2u __attribute__((ext_vector_type(2), aligned(8)));  // (a vector of the output starts on an 8-byte boundary)
    constexpr int NT2 = (HC_CPW * SPN + 127) / 128;
    static_assert(SPN % 2 == 0, "pairs of rows");
    int eo[NT2], eoW[NT2];  // offsets of the lane's pair of elements in a V chain's block of columns | in the W columns
#pragma unroll
    for (int t = 0; t < NT2; ++t) {
        const int e = 2 * ln_ + 128 * t, cc = e / n;
        eo[t] = e < ne ? cc * SP4CS + (e - cc * n) : -1;
        eoW[t] = e < ne ? cc * HC_WS + (e - cc * n) : 0;
    }
    hc_d2 t_[HC_NCH][NT2];
    auto pass_lds = [&](int pass) {  // the lane's output vector of the pass -> its chain slot; then every read of the pass (two LDS round trips per pass, not one per chain)
        if (act) {
#pragma unroll
            for (int i = 0; i < SPD; ++i) Xs[own + i] = pass ? (isV ? accA[i] - accB[i] : accB[i] - accA[i]) : -(accA[i] + accB[i]);
        }
        asm volatile("" ::: "memory");
#pragma unroll
        for (int c2 = 0; c2 < HC_NCH; ++c2)
#pragma unroll
            for (int t = 0; t < NT2; ++t) {
                const double *src = c2 == 0 ? Wreg + eoW[t] : vslots + (c2 - 1) * CB + (eo[t] >= 0 ? eo[t] : 0);
                t_[c2][t].x = src[0], t_[c2][t].y = src[1];
            }
        asm volatile("" ::: "memory");
    };
    auto pass_store = [&](int pass) {
#pragma unroll
        for (int c2 = 0; c2 < HC_NCH; ++c2) {
            // chain 0 (W): the h blocks m | 2 m + 1;  chain 1 + l: l | m + 1 + l
            const int vec = c2 == 0 ? (pass ? 2 * m + 1 : m) : (pass ? m + c2 : c2 - 1);
            double *o = H + HC_NSC + (long long)vec * xd + (long long)c0 * n + 2 * ln_;
#pragma unroll
            for (int t = 0; t < NT2; ++t)
                if (eo[t] >= 0) *(hc_d2u *)(o + 128 * t) = t_[c2][t];
        }
        asm volatile("" ::: "memory");
    };
    pass_lds(0);
    pass_store(0);
    pass_lds(1);
    // ---- the wave is counted in BETWEEN the passes: its row of sums left before the first pass's stores and has been acknowledged by now (or
    //      nearly: waiting for it behind the last store, and then for the counter, cost a wave 8 k cycles at its end, 4 k here); the output
    //      vectors need no order with the counter -- the wave that arrives last reads rows of sums only ----------------------------------------
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (ln_ == 0) xold = __hip_atomic_fetch_add(xcnt + item, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    pass_store(1);
    HC_STAMP();
    // ---- the scalar entries of the interval: the wave that arrived last adds the rows of all HC_NG waves in a fixed order ----------------
    xold = __builtin_amdgcn_readfirstlane(xold);
    if (xold == HC_NG - 1) {
        if (ln_ == 0) __hip_atomic_store(xcnt + item, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // (for the next launch)
        if (ln_ == 0 && rflag) __hip_atomic_store(rflag + item, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // (every wave of the interval has taken its tiles)
        double *tot = vslots;  // [chain][value]
        if (ln_ < HC_XS) {
            const double *xall = xch + (long long)item * HC_NG * HC_XS + ln_;
            double v_[HC_NG], r = 0.0;  // (every row requested, then added in the order of the waves)
#pragma unroll
            for (int g = 0; g < HC_NG; ++g) v_[g] = hc_load_coherent(xall + g * HC_XS);
#pragma unroll
            for (int g = 0; g < HC_NG; ++g) r += v_[g];
            tot[ln_] = r;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (ln_ < HC_NSC) {
            // order: (u_i, u_j) for i = 0..m-1, j = 0..i | (h, u_j) j < m | (h, h);   S[i][j] = tot[(1 + i) HC_ROW + 1 + j]
            double v;
            if (ln_ < m * (m + 1) / 2) {
                int i = 0;
                while ((i + 1) * (i + 2) / 2 <= ln_) ++i;
                const int j = ln_ - i * (i + 1) / 2;
                v = tot[(1 + i) * HC_ROW + 1 + j] + tot[(1 + j) * HC_ROW + 1 + i];
            } else if (ln_ < m * (m + 1) / 2 + m) {
                v = tot[(1 + ln_ - m * (m + 1) / 2) * HC_ROW];
            } else {
                v = tot[0];
            }
            H[ln_] = v;
        }
    }
#else
#error "pcl_kernel_hess_cols_parts.hpp: define one HC_PART_*"
#endif
