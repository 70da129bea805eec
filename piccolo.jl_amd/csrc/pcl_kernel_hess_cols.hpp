// pcl_kernel_hess_cols.hpp -- Hessian of the Lagrangian at ANY diagonal Pade order 2q, pattern-compiled, ONE WAVE PER GROUP OF STATE COLUMNS
// (DESIGN.md section 4.4, round 4).  Included by generated source only (pcl_codegen_v4.hpp with the Hessian functions), like
// pcl_kernel_hess_sparse4.hpp, whose formulas it shares:
//     W_0 = M, W_j = G^T W_{j-1}          V_{l,j} = G^T V_{l,j-1} + G_l^T W_{j-1}  (V_{l,0} = 0)          Z_b(Y) = G^b Y
//     d2/du_l dX_k = -sum_j T_j V_{l,j}   d2/du_l dX_{k+1} = sum_j T_j (-1)^j V_{l,j}   (the same with W_j, T'_j for d2/dh dX)
//     (h,h) = sum_j T''_j <W_j, Y_j>      (h,u_l) = sum_j T'_j <V_{l,j}, Y_j>
//     (u_i,u_l) = sum_a ( <V_{i,a}, G_l R_a> + <V_{l,a}, G_i R_a> ),   R_a = sum_b (+-T_{a+b+1}) Z_b(|Y_{a+b+1}|)
// Every chain is a product with G or G^T applied to the n x d state tile COLUMN BY COLUMN: the chains of one state column need nothing of
// another column; only the 28 scalar entries sum over the columns.  pcl_hess_sparse4_kernel gives a wave to a chain (all columns) and keeps
// the waves of an interval in lock step behind two workgroup barriers per level -- 93 k cycles per interval at order 8 where the multiply-adds
// need 14 k.  Here a wave takes ALL m + 1 chains of HC_CPW = 32 / (m + 1) state columns:
//     lane = (half, chain, column):  half = lane / 32 (top / bottom rows, as in the products), slot = lane % 32 = chain * HC_CPW + column
// and is a workgroup of its own: no barrier, no other wave's data, its LDS (chain slots, D, S, R_1 .. R_{q-2} of its columns: 20.4 KB at order 8)
// lets eight of them share a CU, each at its own place in its own interval -- what one waits for, another computes.  The products are the
// generated sp4_product0_t / sp4_product unchanged (a lane's LDS offsets are operands).
// What the layout costs: the drives' gathers G_l^T W come from an entry table (the lanes of a wave belong to different drives), the chain of the
// R_a uses HC_CPW of the 32 slots, and level 1 runs the product for the W lanes only (7 q + 7 (q - 2) product passes per interval instead of
// 9 q - 11).
// The reduced sums of a wave ((m + 1) x (m + 1): per chain <chain, Y> and the row of (u,u) sums) leave through memory; the wave of the
// interval that arrives last adds the HC_NG rows in a fixed order and writes the entries -- nobody waits for anybody, the same bits for
// every launch geometry (the exchange of pcl_hess_sparse4_kernel's two-workgroup mode: write-through stores, a relaxed arrival counter,
// cache-bypassing loads; no fence).
#pragma once

#define SP4CS (SPN + 1)
#ifndef HC_SWITCH_GATHER
#define HC_SWITCH_GATHER 0  // 1: the drives' gathers as a switch over the generated functions (six passes under the lanes' masks: 8 k cycles per level)
#endif
#define HC_NCH (SPM + 1)                      // chains per state column: W, V_1 .. V_m
#define HC_CPW (32 / HC_NCH)                  // state columns per wave
#define HC_NG ((SPD + HC_CPW - 1) / HC_CPW)   // waves per interval
#define HC_NSLOT (HC_NCH * HC_CPW)
#define HC_NR (SP4Q > 2 ? SP4Q - 2 : 0)       // stored: R_1 .. R_{q-2}  (R_{q-1} = +-T_q |Y_q| is the D or the S tile times a number)
#define HC_NT ((HC_CPW * SPN + 63) / 64)      // passes of the flat (lane = element) copies of a wave's HC_CPW contiguous columns
#define HC_ROW (SPM + 1)                      // reduced sums per chain: <chain, Y>, row of (u,u)
#define HC_XS (HC_NCH * HC_ROW)               // ... per wave
#define HC_NSC ((SPM + 1) * (SPM + 2) / 2)
#define HC_GCH 7                                                        // rows per batch of the table-driven gathers (7, 9: 105 us per 8 seeds at order 8, 5: 110)
#define HC_NCFT 16                                                      // coefficient table of the gathers: 0, +-mags[g]
// The addresses of the drives' gathers come out of a table: per entry one v_bfe_u32 + one v_lshl_add_u32 for the W row (by hand -- the compiler turns
// the extraction into shift, mask, add) and shift + mask for the coefficient, whose table sits at a constant offset the load takes as its immediate
// (round 5: 6 instructions per entry, 256 per pass for 46 multiply-adds; now 4).  MEASURED AND DROPPED (round 6): W columns on 512-byte boundaries,
// so that the address is (row << 3) | base in one v_and_or_b32 -- the four columns of a chain then share their LDS banks and an 8-seed launch at
// order 8 went from 106 to 134 us (profiles/r06_hess_variants_3.log): the kernel is bound by the LDS, not by the vector instructions it issues.
#define HC_WS SP4CS                                                     // doubles per W column (the slots' stride)
#define HC_GT_WPC ((SP4_GT_TOTAL + 2) / 3)                             // the gathers' entry table (sp4_gt_tab): three 10-bit entries per dword, per (drive, half)
#define HC_GT_DOUBLES ((SPM * 2 * HC_GT_WPC + 1) / 2)
// 20,416 bytes at config 3, order 8: EIGHT of these workgroups share a CU's 160 KB (22.9 KB with R_{q-1} stored, a strip of zeros for the W lanes'
// Y term and 16-bit entries: seven; 64 trajectories per launch 807 -> 784 us)
// The R tiles lie LAST, and how many there are is the launch's choice: HC_NR where the wave forms the chain itself, ONE where the interval's R-chain wave has
// left them in memory -- pass jp reads R_jp only, so the tile of the next pass is copied in behind the current one's last use (global_load_lds: no registers,
// a pass of latency to hide in).  At config 3, order 10 that is 18.7 KB instead of 22.2: EIGHT waves per CU instead of seven (profiles/r06_hess_occupancy_37.log:
// seven against eight waves per CU cost 6 % at order 8).
#define HC_RTS ((HC_CPW * SP4CS + 1) & ~1)                              // doubles per R tile (even: whole 16-byte pieces for the copy)
#define HC_RST_OFF (HC_CPW * HC_WS + (HC_NSLOT - HC_CPW + 2 * HC_CPW) * SP4CS + HC_NCFT + HC_GT_DOUBLES)  // one 64-bit word of state: see rtile_copy below
#define HC_RT_OFF ((HC_RST_OFF + 1 + 1) & ~1)
#define HC_LDS_DOUBLES (HC_RT_OFF + HC_NR * HC_RTS)
static_assert(1 + 2 * SP4NMAG <= 16 && 16 <= HC_NCFT && SPN <= HC_WS, "10-bit entries of the gathers' table: source row (6 bits), coefficient index (4 bits)");
static_assert(HC_CPW >= 1 && SPM >= 1, "chains per column");
static_assert(HC_XS <= 64 && HC_ROW <= SPD, "one lane per reduced sum");

template <bool B> struct hc_bool { static constexpr bool value = B; };
static __device__ __forceinline__ unsigned hc_lds_off(const double *q) {
    return (unsigned)(size_t)(__attribute__((address_space(3))) const double *)q;
}
// The rows of reduced sums cross XCDs (whose L2s are not coherent with each other): write-through stores and cache-bypassing loads at system scope
// (sc0 sc1), as relaxed atomics -- the compiler sees them, so the loads of a row are all in flight together and no register is reused under a store.
static __device__ __forceinline__ void hc_store_coherent(double *q, double v) {
    __hip_atomic_store((unsigned long long *)q, (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
static __device__ __forceinline__ double hc_load_coherent(const double *q) {
    return __longlong_as_double((long long)__hip_atomic_load((const unsigned long long *)q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM));
}
// The R-chain wave of an interval and the interval's column-group waves run on ONE XCD (blockIdx equal mod 8: the host launches chain waves only with
// that placement), so their exchange needs the XCD's L2 and no more: device-scope stores here, the column-group waves' copies (global_load_lds ... sc1) past the CU's vector
// cache, served by the L2 -- at system scope every wave paid a round trip to memory for its tiles and the chain waves bought 3 % instead of 12.
static __device__ __forceinline__ void hc_store_xcd(double *q, double v) {
    __hip_atomic_store((unsigned long long *)q, (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// One R-chain wave: ALL d state columns of interval `item`, lane = (half, column) -- 2 d of 64 lanes.  The lane keeps its half column of D and of S in
// registers (a chain wave has no output vectors to hold), so the only LDS it needs is ONE tile of d columns for the product's exchange of the halves
// (11.9 KB at config 3: inside a column-group wave's 20 KB):
//     R_{q-1} = +-T_q |Y_q| (registers);   a = q-2 .. 1:  tile <- G R_{a+1} (sp4_product0: own half written, the partner's half added),
//                                                          R_a = tile + (+-T_{a+1}) |Y_{a+1}|  (registers), -> tile -> memory, lane = element (coalesced).
// (The column-group waves' own chain adds the Y term before the partner's half, this one behind it: the (u,u) entries of the two modes differ in the
//  last bit; the output vectors do not depend on R.)
static_assert(SPD <= 32, "lane = (half, column)");
static __device__ __forceinline__ void hc_rchain_role(const KParams &p, const double *__restrict__ drift_tab, const double *__restrict__ mags_, const double *__restrict__ dcf_tab,
                                                      double *rout, unsigned int *rflag, int item, double *lds) {
    constexpr int d = SPD, n = SPN, q = SP4Q;
    double *Rt = lds;  // [column][SP4CS]
    if (item >= p.batch * p.K) return;
    const int k = item % p.K, b = item / p.K;
    const long long xd = (long long)n * d;
    int ln_ = threadIdx.x;
    asm volatile("" : "+v"(ln_));
    const int half = ln_ >> 5, col = ln_ & 31;
    const bool act = col < d;
    const int own = half * d, oth = (1 - half) * d, cb = (act ? col : 0) * SP4CS;
    const long long xo = p.x_off0 >= 0 ? p.x_off0 : p.x_offs[p.z_batch_stride ? 0 : b];
    const double *zk = p.Z + (long long)b * p.z_batch_stride + (long long)k * p.z_dim + xo + (long long)(act ? col : 0) * n + own;  // the lane's half column: d contiguous doubles
    const double *zn = zk + p.z_dim;
    double Dv[SPD], Sv[SPD];
#pragma unroll
    for (int i = 0; i < SPD; ++i) Dv[i] = zk[i], Sv[i] = zn[i];
    sp_cptr magc = (sp_cptr)mags_;
    double mg[SP4NMAG];
#pragma unroll
    for (int g = 0; g < SP4NMAG; ++g) mg[g] = magc[g];
    sp_cptr zc = (sp_cptr)(p.Z + (long long)b * p.z_batch_stride + (long long)k * p.z_dim);
    double u[SPM];
#pragma unroll
    for (int l = 0; l < SPM; ++l) u[l] = zc[p.u_off + l];
    const double h = zc[p.dt_off];
    sp4_cf cf;
    SP4_SET_CF(cf, u, mg);
    SP4_SET_DCF(cf, (sp_cptr)(dcf_tab + (p.g0_batch_stride ? (long long)b * SP4NDCFP : 0)));
    sp_cptr tab = (sp_cptr)(drift_tab + (p.g0_batch_stride ? (long long)b * SP4NDRIFT : 0));
#pragma unroll
    for (int i = 0; i < SPD; ++i) {
        const double xc = Dv[i], xn = Sv[i];
        Dv[i] = xn - xc, Sv[i] = xn + xc;
    }
    auto wgt_at = [&](int jj) {  // +-T_jj as the column-group waves form it
        double hj = 1.0;
        for (int t = 0; t < jj; ++t) hj *= h;
        return ((jj & 1) ? -1.0 : 1.0) * p.pc[jj] * hj;
    };
    double pwq = 1.0;
#pragma unroll
    for (int j = 1; j <= q; ++j) pwq *= h;
    const double wq = ((q & 1) ? -1.0 : 1.0) * p.pc[q] * pwq;
    double x[SPD];  // R_{a+1}, this lane's half column
#pragma unroll
    for (int i = 0; i < SPD; ++i) x[i] = wq * ((q & 1) ? Sv[i] : Dv[i]);
    double *rg = rout + (long long)item * HC_NR * (HC_NG * HC_RTS);  // [a][column group][HC_RTS]: a column group's tile as it lies in its wave's LDS (columns past the last one: zeros, never written)
#pragma unroll 1
    for (int a = q - 2; a >= 1; --a) {
        if (act) sp4_product0(x, 0u, hc_lds_off(Rt + cb + own), hc_lds_off(Rt + cb + oth), 0.0, 1.0, half ? -1.0 : 1.0, tab, cf);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const double wa = wgt_at(a + 1);
        if (act) {
            const bool odd = (a + 1) & 1;
#pragma unroll
            for (int i = 0; i < SPD; ++i) x[i] = __builtin_fma(wa, odd ? Sv[i] : Dv[i], Rt[cb + own + i]);
            asm volatile("" ::: "memory");
#pragma unroll
            for (int i = 0; i < SPD; ++i) Rt[cb + own + i] = x[i];
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        double *ro = rg + (long long)(a - 1) * (HC_NG * HC_RTS);  // R_a -> the XCD's L2, lane = element (its readers run on this XCD)
        // (the tile goes out AS IT LIES: [column][SP4CS], the columns' padding row included -- nobody reads it -- so that lane = element on both sides with constant
        //  offsets; per-element index arithmetic here, hoisted out of the chain's loop by the compiler, cost 150 spilled registers)
        constexpr int RPAD = HC_RTS - HC_CPW * SP4CS;  // (0 unless HC_CPW SP4CS is odd)
#pragma unroll
        for (int t = 0; t < (d * SP4CS + 63) / 64; ++t) {
            const int e = ln_ + 64 * t;
            if (e < d * SP4CS) hc_store_xcd(ro + e + (RPAD ? (e / (HC_CPW * SP4CS)) * RPAD : 0), Rt[e]);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (the tile is read before the next product writes it)
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the tiles are acknowledged by the L2)
    if (ln_ == 0 && rflag) __hip_atomic_fetch_add(rflag + item, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---- an R tile from memory into LDS by the memory pipe itself (global_load_lds_dwordx4: lane = 16-byte piece, no data registers): `g` is wave-uniform (scalar
//      registers), the LDS address an immediate.  M0 (the copy's LDS base) is not saved: nothing else in this module uses it (the compiler sets M0 only in front of its
//      own M0 instructions -- none here).  The s_waitcnt in front: the tile's last readers (the same wave's ds_reads) have their data.
#define HC_RTILE_NT ((HC_RTS / 2 + 63) / 64)
template <int LDS_BYTES, int T>
static __device__ __forceinline__ void hc_rtile_dma(const double *g, int ln_) {
    if constexpr (T < HC_RTILE_NT) {
        if (64 * (T + 1) <= HC_RTS / 2 || ln_ + 64 * T < HC_RTS / 2)  // (a full piece: every lane, no mask)
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 sc1" ::"v"((unsigned)(16 * ln_ + 1024 * T)), "s"(g), "n"(LDS_BYTES + 1024 * T) : "memory");
        hc_rtile_dma<LDS_BYTES, T + 1>(g, ln_);
    }
}
template <int A>
static __device__ __forceinline__ void hc_rtile_all(const double *base, int ln_) {  // R_{A+1} .. R_{q-2} into tiles A .. HC_NR - 1
    if constexpr (A < HC_NR) {
        hc_rtile_dma<(HC_RT_OFF + A * HC_RTS) * 8, 0>(base + (long long)A * (HC_NG * HC_RTS), ln_);
        hc_rtile_all<A + 1>(base, ln_);
    }
}

extern "C" __global__ __attribute__((amdgpu_flat_work_group_size(64, 64), amdgpu_waves_per_eu(2, 2))) void pcl_hess_cols_kernel(
    const KParams p, const double *__restrict__ drift_tab, const double *__restrict__ drift_tab_t, const double *__restrict__ mags_, const double *__restrict__ dcf_tab,
    double *xch /* [interval][HC_NG][HC_XS] reduced sums */, unsigned int *xcnt /* [interval] arrivals (self-resetting) */,
    double *rpre /* NULL, or [interval][HC_NR][d][n]: R_1 .. R_{q-2} of every state column, formed by the R-chain waves at the head of this launch (below) */,
    unsigned int *rflag /* with rpre: [interval]: R-chain waves that have delivered (self-resetting) */,
    const int rt_all /* with rpre: the launch's LDS holds all HC_NR tiles (they are all copied in at the start) | 0: ONE tile, R_{jp+1} copied in behind R_jp's last use */) {
    extern __shared__ double lds[];
    constexpr int d = SPD, n = SPN, m = SPM, q = SP4Q;
    // ---- R-CHAIN WAVES (p.n_stream of them, the FIRST workgroups of the grid, one per interval; launches of several trajectories): R_{q-2} .. R_1 of
    //      ALL of an interval's state columns, lane = (half, column) -- the same q - 2 products that every column-group wave of round 5 ran on its own
    //      four columns at 8 of 64 lanes (15 % of an 8-seed launch at order 8), here once per interval.  They are dispatched in front of the
    //      column-group waves, write their tiles through to memory and count themselves in; a column-group wave looks at its interval's count with
    //      its first loads and requests its columns' first tile (or all of them: rt_all) together with its other inputs -- straight into LDS (hc_rtile_dma).
    unsigned bid = blockIdx.x;
    if constexpr (HC_NR > 0) {
        if (bid < (unsigned)p.n_stream) {
            hc_rchain_role(p, drift_tab, mags_, dcf_tab, rpre, rflag, (int)bid, lds);
            return;
        }
        bid -= (unsigned)p.n_stream;
    }
    constexpr int CB = HC_CPW * SP4CS;  // doubles per block of HC_CPW columns
    // LDS: [W columns | the V chains' slots | D | S | R_1 .. R_{q-2} | coefficient table | the gathers' entry table]
    double *Wreg = lds, *vslots = Wreg + HC_CPW * HC_WS, *Dt = vslots + (HC_NSLOT - HC_CPW) * SP4CS, *St = Dt + CB, *Rt = lds + HC_RT_OFF;
    double *cft = St + CB;
    unsigned *gtab = (unsigned *)(cft + HC_NCFT);
    // column `cc` of chain `chn`
    if (hc_lds_off(lds) != 0u) __builtin_trap();  // (the gathers' addresses are built as integers on that base)
    auto chain_col = [&](int chn, int cc) -> double * { return chn == 0 ? Wreg + cc * HC_WS : vslots + ((chn - 1) * HC_CPW + cc) * SP4CS; };
    const long long xd = (long long)n * d;
    // Which (interval, column group) this wave takes.  The HC_NG waves of an interval write neighbouring 1,728-byte runs of every output vector: their
    // first and last 128-byte lines are shared.  Workgroups go to the XCDs round-robin (blockIdx mod 8), and each XCD has its own write-back L2:
    // dealt in blockIdx order the waves of an interval sit on HC_NG different XCDs and every shared line goes to memory twice, partly filled
    // (round 4: 168 MB written for 129.5 MB of values).  With p.S = 8 (the host's choice; 1: blockIdx order) the waves of an interval take
    // blockIdx values that are equal mod 8 -- one XCD, one L2, where the partial lines merge; the grid is padded to a multiple of 8 intervals.
    int item, grp;
    if (p.S > 1) {
        const int x = bid % p.S, r = bid / p.S;
        item = (r / HC_NG) * p.S + x;
        grp = r - (r / HC_NG) * HC_NG;
        if (item >= p.batch * p.K) return;
    } else {
        item = bid / HC_NG;
        grp = bid - item * HC_NG;
    }
    // (the chain wave's count FIRST, in front of every other load: loads return in order, so the wave can look at it while its inputs are still on their
    //  way and request its columns' tiles behind them -- requested behind the inputs' arrival they cost every wave a second round trip)
    unsigned rfl_ = 0u;
    if constexpr (HC_NR > 0) {
        if (rpre && rflag) rfl_ = __hip_atomic_load(rflag + item, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    const int k = item % p.K, b = item / p.K;
    const int c0 = grp * HC_CPW, nce = min(HC_CPW, d - c0), ne = nce * n;
    double *H = p.hess + (long long)item * p.hess_per;
    sp_cptr magc = (sp_cptr)mags_;
    double mg[SP4NMAG];
#pragma unroll
    for (int g = 0; g < SP4NMAG; ++g) mg[g] = magc[g];
#ifdef PCL_PROFILE
    int stamp_ = 0;  // cycle stamps of the first wave of the launch (dbg[i])
#define HC_STAMP()                                                                                                 \
    do {                                                                                                           \
        if (p.dbg && blockIdx.x == (unsigned)p.prof && threadIdx.x == 0 && stamp_ < 32) p.dbg[stamp_++] = (long long)__builtin_amdgcn_s_memtime(); \
    } while (0)
#else
#define HC_STAMP() do { } while (0)
#endif
#ifdef HC_MARKS  // (lab/probes/hc_isa/hc_isa_stats.py compiles with -DHC_MARKS: a comment in the ISA at every phase boundary; never in a launched module --
                 //  the statement orders the compiler's memory operations, and that alone cost 20 % of an 8-seed launch when it was left in)
#define HC_MARK(name) asm volatile("; hc_mark " name ::: "memory")
#else
#define HC_MARK(name) do { } while (0)
#endif
    HC_STAMP();
    int ln_ = threadIdx.x;
    asm volatile("" : "+v"(ln_));
    const int half = ln_ >> 5, s = ln_ & 31;
    const int ch = s / HC_CPW, col = s - ch * HC_CPW;  // chain (0: W, 1 + l: V_l), column of the group
    const bool inr = s < HC_NSLOT, act = inr && col < nce, isV = ch > 0;
    const int own = half * d, oth = (1 - half) * d;
    double *Xs = chain_col(inr ? ch : 0, inr ? col : 0);  // this lane's chain column
    const double *Wc = Wreg + col * HC_WS;                // the W chain's column
    const int cb = col * SP4CS;

    // ---- inputs, first half: every load of the wave is requested before anything waits (lane = element of the wave's contiguous columns) --
    const long long xo = p.x_off0 >= 0 ? p.x_off0 : p.x_offs[p.z_batch_stride ? 0 : b];
    const double *zk = p.Z + (long long)b * p.z_batch_stride + (long long)k * p.z_dim + xo + (long long)c0 * n;
    const double *zn = zk + p.z_dim;
    const double *mu = p.mu + (long long)item * xd + (long long)c0 * n;
    // (every load of the wave first: the state columns, then the tables)
    double xc_[HC_NT], xn_[HC_NT], mv_[HC_NT];
#pragma unroll
    for (int t = 0; t < HC_NT; ++t) {
        const int e = ln_ + 64 * t < ne ? ln_ + 64 * t : 0;
        xc_[t] = zk[e], xn_[t] = zn[e], mv_[t] = mu[e];
    }
    // ---- the R tiles of a launch with R-chain waves: R_a of this wave's columns lies in memory as it lies in LDS (hc_rchain_role), and pass jp reads R_jp only:
    //      ONE tile in LDS, copied in by the memory pipe itself (global_load_lds_dwordx4, lane = 16-byte piece; through the XCD's L2 -- sc1 -- where the chain wave
    //      left it) behind the previous tile's last use, a pass ahead of its first.  No registers, no staging writes, and 3.5 KB of LDS less at order 10.
    // (between two copies the tiles' base lives in an LDS word, not in registers: the kernel sits at the 256-register limit and the product takes ~100 scalar
    //  operands -- a per-lane pointer held across the passes cost 54 vector moves per pass, a scalar one 36 v_readlane of spilled scalars per product.
    //  The word: the base while tile-by-tile copies are on, else 0.)
    unsigned long long *rstate = (unsigned long long *)(lds + HC_RST_OFF);
    bool rtile_on = false;  // wave-uniform
    auto rtile_start = [&]() __attribute__((always_inline)) {  // the chain wave has delivered: R_1 (or all of them) on their way, the state word set
        const double *base = rpre + ((long long)item * HC_NR * HC_NG + grp) * HC_RTS;
        hc_rtile_dma<HC_RT_OFF * 8, 0>(base, ln_);
        if (rt_all) hc_rtile_all<1>(base, ln_);
        if (ln_ == 0) *rstate = rt_all ? 0ull : (unsigned long long)base;
    };
    if (ln_ == 0) *rstate = 0ull;
    if constexpr (HC_NR > 0) {
        if (rpre) rtile_start();  // (requested with the inputs, whether or not the chain wave has delivered: the count -- loaded in front -- tells below)
    }
    constexpr int GTW = SPM * 2 * HC_GT_WPC;  // the gathers' entry table, in dwords
    unsigned gw_[(GTW + 63) / 64];
#pragma unroll
    for (int t = 0; t < (GTW + 63) / 64; ++t) gw_[t] = sp4_gt_tab[ln_ + 64 * t < GTW ? ln_ + 64 * t : 0];
    // scalars of the interval
    sp_cptr zc = (sp_cptr)(p.Z + (long long)b * p.z_batch_stride + (long long)k * p.z_dim);
    double u[SPM];
#pragma unroll
    for (int l = 0; l < SPM; ++l) u[l] = zc[p.u_off + l];
    const double h = zc[p.dt_off];
    sp4_cf cf;
    SP4_SET_CF(cf, u, mg);
    SP4_SET_DCF(cf, (sp_cptr)(dcf_tab + (p.g0_batch_stride ? (long long)b * SP4NDCFP : 0)));
    sp_cptr tab = (sp_cptr)(drift_tab + (p.g0_batch_stride ? (long long)b * SP4NDRIFT : 0));
    sp_cptr tab_t = (sp_cptr)(drift_tab_t + (p.g0_batch_stride ? (long long)b * SP4NDRIFT : 0));
    double pw[SP4Q + 1];  // h^j, multiplied up as pcl_hess_sparse4_kernel does (the same bits)
    pw[0] = 1.0;
#pragma unroll
    for (int j = 1; j <= q; ++j) pw[j] = pw[j - 1] * h;
    // +-T_jj: the weight of Z_b(|Y_jj|) in R_{jj-b-1}  (wgt: jj known at compile time; wgt_at: not -- no indexed register array)
    auto wgt = [&](int jj) { return ((jj & 1) ? -1.0 : 1.0) * p.pc[jj] * pw[jj]; };
    auto wgt_at = [&](int jj) {
        double hj = 1.0;
        for (int t = 0; t < jj; ++t) hj *= h;
        return ((jj & 1) ? -1.0 : 1.0) * p.pc[jj] * hj;
    };

    // ---- inputs, second half: M -> the W slots, D, S, R_{q-1}; the tables of the gathers --------------------------------------------------
    {
        if (ln_ < 1 + 2 * SP4NMAG) cft[ln_] = ln_ == 0 ? 0.0 : ((ln_ & 1) ? magc[(ln_ - 1) >> 1] : -magc[(ln_ - 2) >> 1]);
#pragma unroll
        for (int t = 0; t < (GTW + 63) / 64; ++t)
            if (ln_ + 64 * t < GTW) gtab[ln_ + 64 * t] = gw_[t];
#pragma unroll
        for (int t = 0; t < HC_NT; ++t) {
            const int e = ln_ + 64 * t;
            if (e < ne) {
                const int cc = e / n, o = cc * SP4CS + (e - cc * n);
                const double dv = xn_[t] - xc_[t], sv = xn_[t] + xc_[t];
                Wreg[cc * HC_WS + (e - cc * n)] = mv_[t];
                Dt[o] = dv;
                St[o] = sv;
            }
        }
    }
    if constexpr (HC_NR > 0) {
        if (rpre && __builtin_amdgcn_readfirstlane((int)rfl_) < 1) {  // the interval's R-chain wave had not arrived at the start: wait for it (bounded), the tiles again
            unsigned got = 0u;
            for (int it = 0; got < 1u && it < (1 << 20); ++it) {
                __builtin_amdgcn_s_sleep(8);
                got = (unsigned)__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(rflag + item, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            }
            rtile_on = got >= 1u;
            if (rtile_on)
                rtile_start();
            else {  // (a wait that gave up: the tile is poisoned -- and with it the output -- AND the context's error word is set: the next call that looks at it returns PCL_EINTERNAL)
                if (ln_ == 0 && p.err) __hip_atomic_fetch_or(p.err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                for (int e = ln_; e < (rt_all ? HC_NR : 1) * HC_RTS; e += 64) Rt[e] = __builtin_nan("");
            }
        }
    }
    if constexpr (HC_NR > 0) {
        if (rpre && rt_all) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (all tiles were requested with the inputs)
    }
    asm volatile("" ::: "memory");
    HC_STAMP();
    // ---- R_a = sum_b (+-T_{a+b+1}) G^b |Y_{a+b+1}|, the operands of the (u,u) sums, as ONE chain per state column from the top:
    //      R_{q-1} = +-T_q |Y_q| (stored with the inputs),  R_a = +-T_{a+1} |Y_{a+1}| + G R_{a+1}   (q - 2 products; lanes (half, column)) -----
#define HC_RSTRIDE HC_RTS
    if (q > 2 && !rpre) {
#define HC_PART_RCHAIN
#include "pcl_kernel_hess_cols_parts.hpp"
#undef HC_PART_RCHAIN
    }
    HC_STAMP();
    // ---- the chains, pass by pass.  In pass jp the V lanes are at level jp and the W lanes ONE LEVEL BEHIND, at jp - 1:
    //          jp >= 2:  product in place (no Y term)   V: G^T V_{l,jp-1}      W: W_{jp-1} = G^T W_{jp-2}
    //          then the W slots hold W_{jp-1} (jp = 1: M), what the V lanes' gathers need:   V_{l,jp} = G^T V_{l,jp-1} + G_l^T W_{jp-1}
    //      -- added in registers behind the product, so the Y term never passes through LDS (a write and a read per row and level, and
    //      the product's wait for them).  Pass 1 has no product (V_{l,0} = 0), pass q + 1 the W lanes' last one. --------------------------
    HC_MARK("setup");
    const unsigned oX = hc_lds_off(Xs + own), oXx = hc_lds_off(Xs + oth);
    const double bt = half ? 1.0 : -1.0;  // G^T: the other half receives -V from half 0, +V from half 1
    // chain value; the lane's two output vectors (X_k block: -sum_j w_j chain_j, X_{k+1} block: sum_j (-1)^j w_j chain_j) as the sums over its even and
    // its odd levels.  In pass jp the V lanes are at level jp and the W lanes at jp - 1 -- opposite parities -- so with (accA, accB) = (even, odd) in the V
    // lanes and (odd, even) in the W lanes EVERY lane adds to accB in odd passes and to accA in even ones: one multiply-add per row and pass (round 5
    // updated both output vectors in every pass: two), the vectors are formed at the end: -(A + B) and +-(A - B).
    double x[SPD], accA[SPD], accB[SPD];
#pragma unroll
    for (int i = 0; i < SPD; ++i) accA[i] = accB[i] = 0.0;
    if (act && !isV) {
#pragma unroll
        for (int i = 0; i < SPD; ++i) x[i] = Xs[own + i];  // W_0 = M
    } else {
#pragma unroll
        for (int i = 0; i < SPD; ++i) x[i] = 0.0;  // V_{l,0} = 0
    }
    // The reduced sums of a lane's chain -- <chain, Y> and the m (u,u) sums -- are added over the HC_CPW columns of the chain IN EVERY PASS (the lanes of a
    // (half, chain) are neighbours: quads at HC_CPW = 4; two DPP steps, the same bits in every lane of the group) and lane `col` keeps the running totals of
    // the values col and col + HC_CPW only: two accumulators per lane instead of 1 + m.  (Round 4 kept all seven per lane: the compiler held them in
    // scratch across the product, the gather and the contributions -- 7 stores + 7 loads per pass, and 27 MB of scratch write-back per 8-seed launch.)
    // (HC_CPW = 4 -- six drives, every transmon system of the reference: the column lanes of a chain are a DPP quad.  Other drive counts keep the
    //  1 + m sums per lane and add them over the columns once, at the end.)
#define HC_PART_GROUPSUM
#include "pcl_kernel_hess_cols_parts.hpp"
#undef HC_PART_GROUPSUM
    double hpV = 1.0, hpW = 1.0, hpW2 = 1.0;  // h^(level - 1) of the V lanes' and of the W lanes' level; h^(level - 2) of the W lanes'
    // (the body of a pass, compiled twice: for odd and for even passes -- which of the two accumulators a pass adds to is then a matter of the code, not of
    //  a branch; with a branch inside ONE body the compiler copied the 27 accumulators aside and back around it)
    auto pass_body = [&](const int jp, auto odd_) __attribute__((always_inline)) {
        constexpr bool odd_pass = decltype(odd_)::value;
        const bool on = act && (isV ? jp <= q : jp >= 2);  // lanes with a level in this pass
        HC_MARK("pass_product");
        if (jp >= 2) {
            if (on) {
                sp4_product0_t(x, 0u, oX, oXx, 0.0, 1.0, bt, tab_t, cf);
#pragma unroll
                for (int i = 0; i < SPD; ++i) x[i] = Xs[own + i];
            }
            asm volatile("" ::: "memory");
        }
        HC_STAMP();
        HC_MARK("pass_gather");
        // (the module has no static LDS: the wave's dynamic LDS starts at offset 0 -- checked once at the top of the kernel -- so the coefficient
        //  table's offset is a constant of the layout)
        const unsigned wcol_off = (unsigned)col * (HC_WS * 8u);
        constexpr unsigned cft_off = (HC_CPW * HC_WS + (HC_NSLOT - HC_CPW + 2 * HC_CPW) * SP4CS) * 8u;
#define HC_PART_GATHER
#include "pcl_kernel_hess_cols_parts.hpp"
#undef HC_PART_GATHER
        asm volatile("" ::: "memory");
        HC_STAMP();
        // ---- what the lane's level contributes ----
        HC_MARK("pass_contrib");
// (R_jp: with R-chain waves the ONE tile -- its copy has landed: requested a pass ago -- and R_{jp+1} is requested behind its last use)
#define HC_RSTATE_U(lo_, hi_)                                                \
    const unsigned long long w_ = *rstate;                                    \
    const unsigned lo_ = (unsigned)__builtin_amdgcn_readfirstlane((int)w_), hi_ = (unsigned)__builtin_amdgcn_readfirstlane((int)(w_ >> 32))
#define HC_RTILE_WAIT(jp)                                                                       \
    bool r1_ = false;                                                                           \
    if constexpr (HC_NR > 0) {                                                                  \
        HC_RSTATE_U(lo_, hi_);                                                                  \
        r1_ = (lo_ | hi_) != 0u;                                                                \
        if (r1_ && (jp) <= q - 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              \
    }
#define HC_RTILE(jp) (r1_ ? Rt : Rt + ((jp) - 1) * HC_RSTRIDE)
#define HC_RTILE_NEXT(jp)                                                                                                                   \
    do {                                                                                                                                    \
        if (HC_NR > 0 && (jp) + 1 <= q - 2) {                                                                                               \
            HC_RSTATE_U(lo_, hi_);                                                                                                          \
            if (lo_ | hi_) hc_rtile_dma<HC_RT_OFF * 8, 0>((const double *)(((unsigned long long)hi_ << 32) | lo_) + (long long)(jp) * (HC_NG * HC_RTS), ln_); \
        }                                                                                                                                   \
    } while (0)
#define HC_PART_CONTRIB
#include "pcl_kernel_hess_cols_parts.hpp"
#undef HC_PART_CONTRIB
#undef HC_RTILE
#undef HC_RTILE_WAIT
#undef HC_RTILE_NEXT
#undef HC_RSTATE_U
        HC_STAMP();
    };
#pragma unroll 1
    for (int jp = 1; jp <= q + 1; jp += 2) {
        pass_body(jp, hc_bool<true>{});
        if (jp + 1 <= q + 1) pass_body(jp + 1, hc_bool<false>{});
    }
#undef HC_RSTRIDE
    HC_MARK("tail");
#define HC_PART_TAIL
#include "pcl_kernel_hess_cols_parts.hpp"
#undef HC_PART_TAIL
    HC_STAMP();
}

// ---- ONE TRAJECTORY PER LAUNCH: a chain wave and a contribution wave per column group (round 6) ---------------------------------------------------------
// A launch of at most n_cu / 2 intervals runs every wave at once, one or two per SIMD: its time is the LATENCY of one wave -- q + 1 dependent passes of product,
// gather and contributions (~10 k cycles each) behind the R chain.  Only product and gather are the chain; what a level contributes to the output vectors
// and to the scalar entries (accumulation, the Y dot, the gather-dot: 40 % of a pass) feeds nothing later.  So the column group gets a workgroup of TWO waves:
//     wave 0 (chain):         pass jp:  product (into buffer jp & 1 of the chain slots) -> read-back -> gathers -> the level's column back into its slot -> ready = jp
//     wave 1 (contributions): D, S, the R chain (while wave 0 loads, stages and runs its first pass), then per level: wait ready >= jp, the level's column from the slot
//                             -> consumed = jp -> accumulation, Y dot, gather-dot, the sums over the columns; at the end the sums' exchange and the output vectors
//     (the chain wave is the longer of the two: with the Y dot moved over to it -- the parts file can split a level's contributions -- one trajectory at order 10 took
//      30.5 instead of 29.3 us)
// with two buffers of chain slots (wave 0 writes level jp + 1 while wave 1 reads level jp; it waits for consumed >= jp before it writes level jp + 2) and the
// phases' text shared with pcl_hess_cols_kernel (pcl_kernel_hess_cols_parts.hpp): the same arithmetic in the same order, bitwise the same values.
// LDS: [slots 0 | slots 1 | D | S | R_1 .. R_{q-2} | coefficient table | entry table | 2 sync words]: 34 KB at config 3 -- a launch of this kind has LDS to spare.
#define HP_SLOTS (HC_NSLOT * SP4CS)
#ifndef HP_NBUF
#define HP_NBUF 3  // buffers of chain slots between the two waves (2: the chain wave waits for the other one at every level: 27.6 us at order 10; 3: see profiles/r06_hess_pair_*.log)
#endif
#define HP_LDS_DOUBLES (HP_NBUF * HP_SLOTS + (2 + HC_NR) * HC_CPW * SP4CS + HC_NCFT + HC_GT_DOUBLES + 2)
static __device__ __forceinline__ void hp_wait(int *w, int target) {  // bounded: a logic error must not hang the device (the caller's values are then wrong: the launch's error word is set)
    for (int it = 0; it < (1 << 22); ++it) {
        if (__hip_atomic_load(w, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) >= target) return;
        __builtin_amdgcn_s_sleep(1);
    }
}
extern "C" __global__ __attribute__((amdgpu_flat_work_group_size(128, 128), amdgpu_waves_per_eu(2, 2))) void pcl_hess_cols_pair_kernel(
    const KParams p, const double *__restrict__ drift_tab, const double *__restrict__ drift_tab_t, const double *__restrict__ mags_, const double *__restrict__ dcf_tab,
    double *xch /* [interval][HC_NG][HC_XS] reduced sums */, unsigned int *xcnt /* [interval] arrivals (self-resetting) */) {
    extern __shared__ double lds[];
    constexpr int d = SPD, n = SPN, m = SPM, q = SP4Q;
    constexpr int CB = HC_CPW * SP4CS;
    double *Dt = lds + HP_NBUF * HP_SLOTS, *St = Dt + CB, *Rt = St + CB, *cft = Rt + HC_NR * CB;
    unsigned *gtab = (unsigned *)(cft + HC_NCFT);
    int *sync = (int *)(cft + HC_NCFT + HC_GT_DOUBLES);  // [0] levels the chain wave has published | [1] levels the contribution wave has taken | [2] D, S staged | [3] the chain wave's sums parked
    if (hc_lds_off(lds) != 0u) __builtin_trap();
    const long long xd = (long long)n * d;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (threadIdx.x < 4) sync[threadIdx.x] = 0;
    __syncthreads();  // (the only barrier: the sync words exist)
    const unsigned bid = blockIdx.x;
    int item, grp;
    if (p.S > 1) {
        const int x = bid % p.S, r = bid / p.S;
        item = (r / HC_NG) * p.S + x;
        grp = r - (r / HC_NG) * HC_NG;
        if (item >= p.batch * p.K) return;
    } else {
        item = bid / HC_NG;
        grp = bid - item * HC_NG;
    }
    const int k = item % p.K, b = item / p.K;
    const int c0 = grp * HC_CPW, nce = min(HC_CPW, d - c0), ne = nce * n;
    double *H = p.hess + (long long)item * p.hess_per;
    sp_cptr magc = (sp_cptr)mags_;
    double mg[SP4NMAG];
#pragma unroll
    for (int g = 0; g < SP4NMAG; ++g) mg[g] = magc[g];
#ifdef PCL_PROFILE
    int stamp_ = 32;  // (no stamps in this kernel)
#endif
    int ln_ = threadIdx.x & 63;
    asm volatile("" : "+v"(ln_));
    const int half = ln_ >> 5, s = ln_ & 31;
    const int ch = s / HC_CPW, col = s - ch * HC_CPW;
    const bool inr = s < HC_NSLOT, act = inr && col < nce, isV = ch > 0;
    const int own = half * d, oth = (1 - half) * d;
    const int cb = col * SP4CS;
    const int my_slot = (inr ? s : 0) * SP4CS;  // (the W chain's columns are slots 0 .. HC_CPW - 1, as in pcl_hess_cols_kernel: HC_WS == SP4CS)
    static_assert(HC_WS == SP4CS, "the pair kernel's slots are one array per buffer");
    const long long xo = p.x_off0 >= 0 ? p.x_off0 : p.x_offs[p.z_batch_stride ? 0 : b];
    const double *zk = p.Z + (long long)b * p.z_batch_stride + (long long)k * p.z_dim + xo + (long long)c0 * n;
    const double *zn = zk + p.z_dim;
    const double *mu = p.mu + (long long)item * xd + (long long)c0 * n;
    sp_cptr zc = (sp_cptr)(p.Z + (long long)b * p.z_batch_stride + (long long)k * p.z_dim);
    double u[SPM];
#pragma unroll
    for (int l = 0; l < SPM; ++l) u[l] = zc[p.u_off + l];
    const double h = zc[p.dt_off];
    sp4_cf cf;
    SP4_SET_CF(cf, u, mg);
    SP4_SET_DCF(cf, (sp_cptr)(dcf_tab + (p.g0_batch_stride ? (long long)b * SP4NDCFP : 0)));
    sp_cptr tab = (sp_cptr)(drift_tab + (p.g0_batch_stride ? (long long)b * SP4NDRIFT : 0));
    sp_cptr tab_t = (sp_cptr)(drift_tab_t + (p.g0_batch_stride ? (long long)b * SP4NDRIFT : 0));
    double x[SPD];
    constexpr unsigned cft_off = (HP_NBUF * HP_SLOTS + (2 + HC_NR) * HC_CPW * SP4CS) * 8u;

    if (wave == 0) {
        // ================================================= the chain wave =================================================
        double mv_[HC_NT];
#pragma unroll
        for (int t = 0; t < HC_NT; ++t) mv_[t] = mu[ln_ + 64 * t < ne ? ln_ + 64 * t : 0];
        constexpr int GTW = SPM * 2 * HC_GT_WPC;
        unsigned gw_[(GTW + 63) / 64];
#pragma unroll
        for (int t = 0; t < (GTW + 63) / 64; ++t) gw_[t] = sp4_gt_tab[ln_ + 64 * t < GTW ? ln_ + 64 * t : 0];
        if (ln_ < 1 + 2 * SP4NMAG) cft[ln_] = ln_ == 0 ? 0.0 : ((ln_ & 1) ? magc[(ln_ - 1) >> 1] : -magc[(ln_ - 2) >> 1]);
#pragma unroll
        for (int t = 0; t < (GTW + 63) / 64; ++t)
            if (ln_ + 64 * t < GTW) gtab[ln_ + 64 * t] = gw_[t];
        double *W1 = lds + (1 % HP_NBUF) * HP_SLOTS;  // M = W_0 goes where the first pass's gathers look for it: the W slots of level 1's buffer
#pragma unroll
        for (int t = 0; t < HC_NT; ++t) {
            const int e = ln_ + 64 * t;
            if (e < ne) {
                const int cc = e / n;
                W1[cc * SP4CS + (e - cc * n)] = mv_[t];
            }
        }
        asm volatile("" ::: "memory");
        if (act && !isV) {
#pragma unroll
            for (int i = 0; i < SPD; ++i) x[i] = W1[my_slot + own + i];  // W_0 = M
        } else {
#pragma unroll
            for (int i = 0; i < SPD; ++i) x[i] = 0.0;  // V_{l,0} = 0
        }
        const double bt = half ? 1.0 : -1.0;
        // (this wave holds no output vectors: it has the registers to keep its (drive, half)'s table words for all passes and to gather 14 rows at a time)
        unsigned pk_[SP4_GT_TOTAL];  // per entry: W row << 3 | (the coefficient's LDS address) << 16 -- decoded ONCE (the one-wave kernel decodes in every pass: no registers)
        {
            const unsigned *g0 = sp4_gt_tab + ((isV ? ch - 1 : 0) * 2 + half) * HC_GT_WPC;
            unsigned gtw_[HC_GT_WPC];
#pragma unroll
            for (int t = 0; t < HC_GT_WPC; ++t) gtw_[t] = g0[t];
#pragma unroll
            for (int en = 0; en < SP4_GT_TOTAL; ++en) {
                const unsigned e = (gtw_[en / 3] >> (10 * (en % 3))) & 1023u;
                pk_[en] = ((e >> 4) << 3) | ((((e & 15u) << 3) + cft_off) << 16);
            }
        }
        static_assert(HP_LDS_DOUBLES * 8 < 65536, "16-bit LDS addresses in pk_");
#pragma unroll 1
        for (int jp = 1; jp <= q + 1; ++jp) {
            const bool on = act && (isV ? jp <= q : jp >= 2);
            double *slots = lds + (jp % HP_NBUF) * HP_SLOTS;
            double *Xs = slots + my_slot;
            const double *Wc = slots + cb;
            if (jp > HP_NBUF) hp_wait(sync + 1, jp - HP_NBUF);  // (the contribution wave has taken level jp - HP_NBUF out of this buffer)
            if (jp >= 2) {
                if (on) {
                    sp4_product0_t(x, 0u, hc_lds_off(Xs + own), hc_lds_off(Xs + oth), 0.0, 1.0, bt, tab_t, cf);
#pragma unroll
                    for (int i = 0; i < SPD; ++i) x[i] = Xs[own + i];
                }
                asm volatile("" ::: "memory");
            }
            const unsigned wcol_off = hc_lds_off(Wc);
#pragma push_macro("HC_GCH")
#undef HC_GCH
#define HC_GCH 14
#define HC_GT_PREDECODED
#define HC_PART_GATHER
#include "pcl_kernel_hess_cols_parts.hpp"
#undef HC_PART_GATHER
#undef HC_GT_PREDECODED
#pragma pop_macro("HC_GCH")
            asm volatile("" ::: "memory");
            if (on && isV) {  // the level's column where the contribution wave finds it (the W lanes' is there: the product's read-back)
#pragma unroll
                for (int i = 0; i < SPD; ++i) Xs[own + i] = x[i];
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (ln_ == 0) __hip_atomic_store(sync + 0, jp, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        return;
    }
    // ================================================= the contribution wave =================================================
    double xc_[HC_NT], xn_[HC_NT];
#pragma unroll
    for (int t = 0; t < HC_NT; ++t) {
        const int e = ln_ + 64 * t < ne ? ln_ + 64 * t : 0;
        xc_[t] = zk[e], xn_[t] = zn[e];
    }
    double pw[SP4Q + 1];
    pw[0] = 1.0;
#pragma unroll
    for (int j = 1; j <= q; ++j) pw[j] = pw[j - 1] * h;
    auto wgt = [&](int jj) { return ((jj & 1) ? -1.0 : 1.0) * p.pc[jj] * pw[jj]; };
    auto wgt_at = [&](int jj) {
        double hj = 1.0;
        for (int t = 0; t < jj; ++t) hj *= h;
        return ((jj & 1) ? -1.0 : 1.0) * p.pc[jj] * hj;
    };
#pragma unroll
    for (int t = 0; t < HC_NT; ++t) {
        const int e = ln_ + 64 * t;
        if (e < ne) {
            const int cc = e / n, o = cc * SP4CS + (e - cc * n);
            Dt[o] = xn_[t] - xc_[t];
            St[o] = xn_[t] + xc_[t];
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (ln_ == 0) __hip_atomic_store(sync + 2, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
// (this kernel always forms the chain itself: HC_NR tiles of CB doubles)
#define HC_RSTRIDE CB
#define HC_RTILE(jp) (Rt + ((jp) - 1) * CB)
#define HC_RTILE_WAIT(jp)
#define HC_RTILE_NEXT(jp) do { } while (0)
    if (q > 2) {
#define HC_PART_RCHAIN
#include "pcl_kernel_hess_cols_parts.hpp"
#undef HC_PART_RCHAIN
    }
    double accA[SPD], accB[SPD];
#pragma unroll
    for (int i = 0; i < SPD; ++i) accA[i] = accB[i] = 0.0;
#define HC_PART_GROUPSUM
#include "pcl_kernel_hess_cols_parts.hpp"
#undef HC_PART_GROUPSUM
    double hpV = 1.0, hpW = 1.0, hpW2 = 1.0;
    auto level = [&](const int jp, auto odd_) __attribute__((always_inline)) {
        constexpr bool odd_pass = decltype(odd_)::value;
        const bool on = act && (isV ? jp <= q : jp >= 2);
        const double *Xl = lds + (jp % HP_NBUF) * HP_SLOTS + my_slot;
        hp_wait(sync + 0, jp);
        if (on) {
#pragma unroll
            for (int i = 0; i < SPD; ++i) x[i] = Xl[own + i];
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (ln_ == 0) __hip_atomic_store(sync + 1, jp, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
#define HC_PART_CONTRIB
#include "pcl_kernel_hess_cols_parts.hpp"
#undef HC_PART_CONTRIB
    };
#pragma unroll 1
    for (int jp = 1; jp <= q + 1; jp += 2) {
        level(jp, hc_bool<true>{});
        if (jp + 1 <= q + 1) level(jp + 1, hc_bool<false>{});
    }
#undef HC_RSTRIDE
#undef HC_RTILE
#undef HC_RTILE_WAIT
#undef HC_RTILE_NEXT
    // the tail works in buffer 0 (the chain wave has published its last level: it writes nothing any more)
    double *Wreg = lds, *vslots = lds + HC_CPW * HC_WS;
    auto chain_col = [&](int chn, int cc) -> double * { return chn == 0 ? Wreg + cc * HC_WS : vslots + ((chn - 1) * HC_CPW + cc) * SP4CS; };
    double *Xs = lds + my_slot;
    unsigned int *rflag = nullptr;
#define HC_PART_TAIL
#include "pcl_kernel_hess_cols_parts.hpp"
#undef HC_PART_TAIL
}
