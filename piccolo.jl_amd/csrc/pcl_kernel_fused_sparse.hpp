// pcl_kernel_fused_sparse.hpp -- fused residual + Jacobian, PATTERN-COMPILED, any diagonal Pade order 2q (the residual-only kernels of section 4.3 at the end of the file).
// Included by generated source only (pcl_codegen_v4.hpp): SPD (Hilbert dimension), SPM (drives), SPN = 2 SPD, SP4Q (q), SP4NP
// (tiles of the powers of G), the resident-coefficient struct sp4_cf, the products sp4_product / sp4_product0 and the drives'
// gathers sp4_gather_<l> are defined before this file.
//
// With Y_j = D (j even) or -S (j odd), D = X_{k+1} - X_k, S = X_{k+1} + X_k, c_j the Pade coefficients, h the step:
//     level q:           W = c_q Y_q          V = q c_q Y_q            dW_l = 0                            P = I
//     level j = q-1..0:  W <- c_j Y_j + h G W V <- j c_j Y_j + h G V   dW_l <- h (G_l W_old + G dW_l)      P <- G P
//     (level 0:          delta = W            d delta/dh = G V         d delta/du_l = dW_l)                B^{+-} = sum_j c_j (+-h)^j G^j
// Every chain acts on the state columns from the left: lane (half, c) owns its half of column c, G(u) x is the straight-line
// product sp4_product (coefficients in scalar registers, no LDS operand traffic, no matrix-core padding: 307 multiply-adds
// instead of 112 MFMAs per 27 columns at BASELINE config 3).  ONE persistent workgroup per CU, one WAVE per chain:
//     wave 0            P: the powers of G (first d columns: the generators are exact iso(.) images) into a ring of SP4NP tiles
//     wave 1, 2         W, V
//     wave 3 + l        dW_l
//     wave 3 + m        loader: D, S of the next item (lane = row, coalesced) -> tiles [column][row]
//     wave 4 + m        writer: the reduce payload's dot products per state column (pcl_eval_jac_merit_dev); delta and the tail block of a
//                       finished item only in tail modes 0-2 (default: the stream waves store them behind the item's blocks)
//     wave 5 + m .. +3  stream: fold every power into the item's -B^+ / B^- values in registers as it appears, then only issue the
//                       replicated 16-byte stores
//     (the FIRST item's powers are built by waves 0 .. 3 together, in four row ranges, before they take up the roles above)
// No workgroup barrier after the start: point-to-point monotonic LDS counters (dependencies only point backwards; bounded waits).
// Work items as in kernel 3: contiguous column ranges per workgroup (pieces of one interval), or round-robin slices -- or, for launches
// of several trajectories, GROUPS WITH SLICE TICKETS (p.tick):
//   * the workgroups form groups of tick_G (8: one per XCD); group g walks the intervals g, g + n_groups, ... in a STATIC order, so the
//     interval a workgroup works on next -- hence the controls and the powers of G it needs -- is known in advance;
//   * inside an interval the group's members take slices of tick_cpi state columns (the replicated -B^+ / B^- blocks) from the interval's own
//     counter, one at a time, each when its stream waves have issued the previous slice: the addresses are assigned LATE, the workgroups
//     the memory side serves first take more slices, a member that finds an interval exhausted moves on;
//   * an interval's chains (delta, d/du_l, d/dh, the reduce payload: loader, W, V, dW_l, writer waves) are a second pipeline with its own
//     device-wide ticket counter, taken by whichever workgroup's chain waves are free.
// One more wave, the dispatcher, does everything that waits for memory on behalf of the block pipeline: it takes the slice tickets and keeps
// a ring of the group's next controls and steps in LDS.  Measured with the bare store pattern (lab/probes/wfront.hip, 8 trajectories, 8
// separately allocated buffers, one box): equal contiguous ranges 196.6 us median (163.7 ... 209.3 by where the buffer's pages live), static
// round-robin items 199-229, items of 3 columns by ONE device-wide ticket 176, this scheme 177-182, on every buffer.  (The first version of
// this round took device-wide block tickets in the P wave: parity-exact but 247 us -- every ticket's controls are a cold load behind the CU's
// own stores, 3-8 us, and tickets taken early to hide it cost 5-10 us each: the write front is only as tight as the tickets are late.)
#pragma once

#ifndef SP4_TICKETS
#define SP4_TICKETS 1             // 0: the module of the static work splits (no dispatcher, no ticket branches: 10 KB less code for the launches whose start-up counts)
#endif
#define SP4CS (SPN + 1)           // odd column stride: the lanes of a half wave, one column each, hit distinct banks
#define SP4TILE (SP4CS * SPD)
#define SP4_WLOAD (SPM + 3)
#define SP4_WWRITE (SPM + 4)
#define SP4_WSTREAM (SPM + 5)
#define SP4_NSTREAM 4
#define SP4_NWAVES (SPM + 9)      // static work splits; launches with slice tickets add the dispatcher wave
#define SP4_WDISP (SPM + 9)
#define SP4_UHR 128               // visits whose controls and step the dispatcher keeps in LDS (8 doubles each)
#define SP4_NOUT (SPM + 2)        // output chains: W (delta), V (d/dh), dW_l (d/du_l)
#define SP4_NTILES (SPM + 4 + SP4NP)  // D, S, W, V, dW[m], P[SP4NP]
#define SP4_SYNC_WORDS 48        // 32 counters + the two ticket rings (4 block items with their steps, 4 chain items)
// monotonic counters: IN items whose D, S are in LDS | DW, DV items whose D, S the W / V wave has finished with | W levels of W
// published | B powers published (all items) | C + w powers folded by stream wave w | OC items the writer has taken out of the tiles |
// G + l levels of W the drive wave l has gathered | O + w items whose output chain w is complete
// (one word per arriver wherever arrivers can run ahead of each other: a shared arrival count lies -- a fast wave's extra arrival
//  stands in for a slow wave's missing one)
enum { SP4_F_IN = 0, SP4_F_DW, SP4_F_DV, SP4_F_W, SP4_F_B, SP4_F_OC, SP4_F_CO /* cooperative first item: parts of powers done */, SP4_F_CX /* ... operands read (one-tile ring) */, SP4_F_G = 8, SP4_F_BI = 14 /* tickets: slices published */, SP4_F_CI = 15 /* ... chain items published */, SP4_F_O = 16, SP4_F_C = 24 /* per stream wave */, SP4_F_TS = 28 /* per stream wave: items whose tails it has stored (tickets: slices whose stores it has issued) */,
       SP4_F_V = 44 /* tickets: the visit of the oldest slice in flight */, SP4_F_VI = 45 /* ... visits whose controls are in the ring */, SP4_F_V2 = 46 /* ... the visit of the newest slice in flight */,
       SP4_F_OA = 47 /* output chains that have finished an item, all of them in ONE word: what the stream waves poll between two columns (an item's arrivals begin when every chain has arrived for the item before -- outputs_taken -- so the shared count does not lie here) */ };

static __device__ __forceinline__ bool sp4_wait(int *sync, int word, int target, bool gave_up = false) {
    // Bounded: a logic error must not hang the device (the caller poisons the output instead; once a wave has given up it
    // does not wait again).
    if (gave_up) return true;
    // back-off: a level-scale wait is answered within a few short polls; an item-scale wait (the chains run several times faster than
    // the store stream) must not keep ten waves polling next to it -- every poll takes an issue slot and an LDS cycle from the CU
    int it = 0;
    for (; __hip_atomic_load(sync + word, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < target && it < (1 << 18); ++it) {
        if (it < 16)
            __builtin_amdgcn_s_sleep(1);
        else if (it < 32)
            __builtin_amdgcn_s_sleep(16);
        else
            __builtin_amdgcn_s_sleep(127);
    }
    return it >= (1 << 18);
}
// the two hand-offs of a slice ticket (stream waves -> dispatcher -> stream waves) sit on every slice's critical path and are answered within
// a few thousand cycles: short sleeps for that long, then the usual back-off
static __device__ __forceinline__ bool sp4_wait_soon(int *sync, int word, int target, bool gave_up = false) {
    if (gave_up) return true;
    for (int it = 0; it < 128; ++it) {
        if (__hip_atomic_load(sync + word, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) >= target) return false;
        __builtin_amdgcn_s_sleep(2);
    }
    return sp4_wait(sync, word, target, false);
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

#ifndef SP4_RESIDENT
#define SP4_RESIDENT 0            // 1: the module of the RESIDENT evaluator (pcl_resident_*): the evaluation below as a device function, called once per posted request
#endif
#if SP4_RESIDENT
extern "C" {  // (the dynamic LDS array is declared by extern "C" kernels too: one language linkage)
static __device__ __forceinline__ void sp4_fused_body(const KParams &p, const double *__restrict__ drift_tab, const double *__restrict__ mags_, const double *__restrict__ dcf_tab) {
#else
extern "C" __global__ __launch_bounds__(64 * (SP4_NWAVES + 1)) void pcl_fused_sparse_kernel(const KParams p, const double *__restrict__ drift_tab, const double *__restrict__ mags_, const double *__restrict__ dcf_tab) {
#endif
    extern __shared__ double lds[];
    constexpr int d = SPD, n = SPN, m = SPM, q = SP4Q, nn = SPN * SPN;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // scalar: the role branches are uniform
    double *Dt = lds, *St = Dt + SP4TILE, *Wt = St + SP4TILE, *Vt = Wt + SP4TILE, *dWt = Vt + SP4TILE;
    double *Pt = dWt + m * SP4TILE;  // SP4NP tiles: power j of item `it` lives in tile (it q + j - 1) mod npw
    // npw <= SP4NP tiles in use.  With all q powers of an item in flight the P wave runs a whole item ahead of the stream -- measured
    // 8 % SLOWER on launches of several items per workgroup than a ring of q - 1 (the host picks; one-item launches take all q)
    const int npw = p.v4_np;
    int *sync = (int *)(Pt + SP4NP * SP4TILE);
#ifdef PCL_PROFILE
    // cycle stamps of workgroup 0: the first 32 per wave (dbg[32 wave + i]); experiment flags (results WRONG): prof & 2 no block
    // stores, prof & 4 no column chains (P and the stream only), prof & 8 no tail / delta stores
    int stamp_ = 0;
#define SP4_STAMP()                                                                                                          \
    do {                                                                                                                     \
        if (p.dbg && blockIdx.x == 0 && lane == 0 && stamp_ < 32) p.dbg[32 * wave + stamp_++] = (long long)__builtin_amdgcn_s_memtime(); \
    } while (0)
    const bool no_blocks = p.prof & 2, no_chains = p.prof & 4, no_tails = p.prof & 8;
#else
#define SP4_STAMP() do { } while (0)
    constexpr bool no_blocks = false, no_chains = false, no_tails = false;
#endif
#ifdef PCL_PROFILE
    // 100 MHz wall stamps per workgroup (dbg[512 + 768 sel + 3 bx + {entry, tiles zeroed, last wave out}], sel = prof & 64): where
    // the time between back-to-back launches goes
    long long *wall_ = p.dbg ? p.dbg + 512 + ((p.prof & 64) ? 768 : 0) + 3 * (blockIdx.x & 255) : nullptr;
    if (wall_ && tid == 0) wall_[0] = (long long)__builtin_amdgcn_s_memrealtime();
#endif
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
            // (no division where the split has none to make -- one trajectory in two slices per interval is the launch whose start-up counts:
            //  three integer divisions of cold code were 0.5 k cycles ahead of every role's first loads)
            const int item = bx + it * (int)gridDim.x;
            int s, ik;
            if (p.S == 2) {
                s = item & 1;
                ik = item >> 1;
            } else if (p.S == 1) {
                s = 0;
                ik = item;
            } else {
                ik = item / p.S;
                s = item - ik * p.S;
            }
            c0 = s * p.nc;
            nce = min(p.nc, d - c0);
            b = p.batch == 1 ? 0 : ik / p.K;
            k = ik - b * p.K;
        }
    };
    // ---- slice tickets (see the head of the file): rings in LDS behind the counters -- the slice in flight (bdesc: 32 visit + slice), the visit
    //      each of the P wave's last four builds holds (pvis), the chain items (cdesc: the interval), the controls and step of SP4_UHR visits (uhr)
    const bool tick = SP4_TICKETS && p.tick != nullptr;
    int *bdesc = sync + 32, *cdesc = sync + 36, *pvis = sync + 40;
    double *uhr = (double *)(sync + SP4_SYNC_WORDS);
    const int t_ipi = tick ? (d + p.tick_cpi - 1) / p.tick_cpi : 1;
    const int n_int = p.batch * p.K;
    const int n_groups = tick ? (int)gridDim.x / p.tick_G : 1, grp = tick ? bx / p.tick_G : 0;
    const int n_vis = tick && grp < n_int ? (n_int - grp + n_groups - 1) / n_groups : 0;  // intervals grp, grp + n_groups, ...
    bool gave_up = false;
    // a pipeline's last ticket has come back: the last of the 2 gridDim.x pipelines to leave re-zeroes the counters for the next launch
    auto ticket_leave = [&](unsigned n) {
        unsigned gone = 0;
        if (lane == 0) gone = __hip_atomic_fetch_add(p.tick + 1, n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        gone = __builtin_amdgcn_readfirstlane(gone);
        if (gone + n == 2u * gridDim.x) {
            for (int i = lane; i < n_int; i += 64) __hip_atomic_store(p.tick + 4 + i, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (lane == 0) {
                __hip_atomic_store(p.tick + 2, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(p.tick + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    };
    // item `it` of this workgroup's chain pipeline (W, V, dW_l, writer; the loader takes the tickets): false = no more
    auto chain_item = [&](int it, int &c0, int &nce, int &k, int &b) -> bool {
        if (!tick) {
            if (it >= n_my) return false;
            decode(it, c0, nce, k, b);
            return true;
        }
        gave_up = sp4_wait(sync, SP4_F_CI, it + 1, gave_up);
        const int iv = __builtin_amdgcn_readfirstlane(__hip_atomic_load(cdesc + (it & 3), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
        if (iv < 0 || gave_up) return false;
        b = p.batch == 1 ? 0 : iv / p.K;
        k = iv - b * p.K;
        c0 = 0;
        nce = d;
        return true;
    };
    // Counters zero.  The tiles start as whatever the previous workgroup left: every entry a wave reads has been written by the item
    // that reads it (zeroing 150 KB took 0.84 us of a 24 us workgroup life).  v4_flags & 8 (tests): NaN everywhere first.
    if (p.v4_flags & 8)
        for (int e = tid; e < SP4_NTILES * SP4TILE; e += 64 * SP4_NWAVES) lds[e] = __builtin_nan("");
#ifdef PCL_PROFILE
    if (p.prof & 128) SP4_STAMP();
#endif
    if (tid < SP4_SYNC_WORDS) sync[tid] = 0;
    __syncthreads();  // the only workgroup barrier
#ifdef PCL_PROFILE
    if (p.prof & 128) SP4_STAMP();
    if (wall_ && tid == 0) wall_[1] = (long long)__builtin_amdgcn_s_memrealtime();
#endif

    // delta and the tail block of a finished item, tiles -> memory: the item's nce n residuals and its nce (m + 1) n tail values
    // are ONE contiguous run each, 1 KiB per instruction; `part` of `nparts` waves takes every nparts-th instruction.
    // tail_mode: 0 the writer wave, plain stores | 1 nontemporal | 2 write-through | 3 the stream waves, behind the item's blocks
    const double *Wres = Wt;  // where a finished item's delta stands (the W chain's other tile for one-item launches at odd q: set below)
    auto store_outputs = [&](int c0, int nce, long long bk, int part, int nparts, int nt) {
        int l0 = lane + 64 * part;
        asm volatile("" : "+v"(l0));
        if (p.delta) {
            double *dst = p.delta + bk * xd + (long long)c0 * n;
            for (int e2 = l0; e2 < nce * d; e2 += 64 * nparts) {
                const int cl = e2 / d, r0 = 2 * (e2 - cl * d);
                const double *src = Wres + cl * SP4CS + r0;
                store2(dst + 2 * e2, src[0], src[1], nt);
            }
        }
        double *dst = p.jac + bk * p.jac_per + 2 * blk + (long long)c0 * (m + 1) * n;
        // element pair e2 of the run: column cl, vector v (drives 0 .. m-1, then d/dh), row pair r0
        for (int e2 = l0; e2 < nce * (m + 1) * d; e2 += 64 * nparts) {
            const int cv = e2 / d, r0 = 2 * (e2 - cv * d);
            const int cl = cv / (m + 1), v = cv - cl * (m + 1);
            const double *src = (v < m ? dWt + v * SP4TILE : Vt) + cl * SP4CS + r0;
            store2(dst + 2 * e2, src[0], src[1], nt);
        }
    };
    const bool tails_by_stream = p.tail_mode == 3 && !tick;  // (ticket mode: the writer wave -- the item whose tails are ready is not the stream's)
    // the chains may rewrite their tiles once item it - 1 has left them
    auto outputs_taken = [&](int it) {
        if (!tails_by_stream || p.mpart) gave_up = sp4_wait(sync, SP4_F_OC, it, gave_up);
        if (tails_by_stream) {
            for (int w = 0; w < SP4_NSTREAM; ++w) gave_up = sp4_wait(sync, SP4_F_TS + w, it, gave_up);
        }
    };

    // the drives' magnitudes and the per-item scalars (step, controls -> resident coefficients, the member's drift table), defined inside
    // the roles that use them: at kernel scope their registers stay live across every role (139 spilled scalar registers)
#define SP4_SCALARS_DEF()                                                                                         \
    sp_cptr magc = (sp_cptr)mags_;                                                                                \
    double mg[SP4NMAG];                                                                                           \
    _Pragma("unroll") for (int g = 0; g < SP4NMAG; ++g) mg[g] = magc[g];                                          \
    auto scalars = [&](int k, int b, double &h, sp4_cf &cf, sp_cptr &tab) {                                       \
        sp_cptr zc = (sp_cptr)(p.Z + (long long)b * p.z_batch_stride + (long long)k * p.z_dim);                   \
        /* (the drift classes first: every load of the item is then requested before the first use -- one round trip instead of two) */ \
        SP4_SET_DCF(cf, (sp_cptr)(dcf_tab + (p.g0_batch_stride ? (long long)b * SP4NDCFP : 0)));                  \
        double u[SPM > 0 ? SPM : 1];                                                                              \
        _Pragma("unroll") for (int l = 0; l < SPM; ++l) u[l] = zc[p.u_off + l];                                   \
        h = zc[p.dt_off];                                                                                         \
        SP4_SET_CF(cf, u, mg);                                                                                    \
        tab = (sp_cptr)(drift_tab + (p.g0_batch_stride ? (long long)b * SP4NDRIFT : 0));                          \
    }
    // Cooperative first item: the powers of G of the workgroup's FIRST item are built by four waves together, a quarter of the output rows
    // each (sp4_product0_part: nobody can store before the powers exist; a lone P wave takes 7.7 k cycles for G, G^2 at order 4 -- a
    // fifth of a one-trajectory launch).  Needs fully resident coefficients (the generator then emits the parts).  The P wave's own loop
    // starts with item 1.
    const bool coop = SP4_COOP && !(p.v4_flags & 4) && !tick;
    // ... with ALL q powers resident even where the ring is shorter (orders 8 and 10 at config 3: three power tiles): the chains of the first
    // item start behind the stream's folds anyway (chains_may_start), so until then their dW tiles are free -- powers npw + 1 .. q of the FIRST
    // item live there.  No product waits for a fold, no fold for a tile (order 10 ran without the cooperative start before: five powers
    // through three tiles were slower than a lone P wave).  Needs q - npw <= m and the chains held back (v4_flags & 16 off).
    const bool borrow = coop && q > npw && q - npw <= m && !(p.v4_flags & 16);
    // ONE-item workgroups (one trajectory per launch) at orders 8 and 10 are bound by the CHAINS, not by the stores: W's level s + 1 overwrites
    // the tile the m drive waves gather level s from, so W and the dW_l advance in lock step -- q x (W product + gather + dW product), 44 k cycles
    // at order 8 against 33 k of block stores (stamps, round 5).  With nothing behind the first item the power tiles are free once the stream has
    // folded them: W alternates between its tile and power tile 0 and runs a level ahead of the gathers; the dW chains no longer wait for it.
    // (measured, one trajectory per launch, with / without: order 10 29.5 / 32.0 us, order 8 27.7 / 28.1, order 6 26.2 / 25.7, order 4 equal: from order 8 on;
    //  v4_flags & 64 switches it off)
    const bool pingpong = coop && n_my == 1 && !(p.v4_flags & (16 | 64)) && q >= 4;
    double *const Walt = pingpong ? Pt : Wt;                   // W after an odd number of steps
    const double *const Wfin = (pingpong && (q & 1)) ? Pt : Wt;  // delta = W after q steps
    Wres = Wfin;
    auto first_tile = [&](int jm1) -> double * {  // tile of power jm1 + 1 of the workgroup's FIRST item
        return borrow ? (jm1 < npw ? Pt + jm1 * SP4TILE : dWt + (jm1 - npw) * SP4TILE) : Pt + (jm1 % npw) * SP4TILE;
    };
    // ... and the chains of that item start behind them: twelve waves of products on four SIMDs ran the stream's parts three times
    // slower (4.1 k cycles instead of 1.3 k), and the chains have the whole store phase to finish in
    auto chains_may_start = [&](int it) {
        if (coop && it == 0 && !(p.v4_flags & 16))
            for (int w = 0; w < SP4_NSTREAM; ++w) gave_up = sp4_wait(sync, SP4_F_C + w, q, gave_up);
    };

    // Lane position, re-derived from an opaque copy of `lane` in every item: computed once, everything that depends on it (the
    // unit vectors, tile addresses, ...) is hoisted out of the item loops and spilled.
#define SP4_LANEPOS()                                                                       \
    int ln_ = lane;                                                                         \
    asm volatile("" : "+v"(ln_));                                                           \
    const int half = ln_ >> 5, c = ln_ & 31;                                                \
    const int cc = c < d ? c : 0;                                                           \
    const int own = cc * SP4CS + half * d, oth = cc * SP4CS + (1 - half) * d
#if SP4_COOP
    // ---- cooperative first item: waves 0 .. 3 (P and the first three chains: idle until the item's powers exist) build the powers of
    //      G in four row ranges; the store-stream waves fold every power while the next one is being built.  (Until round 3 the stream
    //      waves built them themselves and the fold of a power sat between two products of the same wave: 27.6 -> 25.7 us per
    //      one-trajectory launch.  Nine parts on nine waves measured slower than four: 26.5 us -- a part's time is its cold start.)
    if (coop && wave < SP4_NPART && n_my > 0) {
#ifdef PCL_PROFILE
        if (p.prof & 128) SP4_STAMP();
#endif
        __builtin_amdgcn_s_setprio(3);  // the other waves of the workgroup are polling counters until these products are done (round 5: -0.3 us)
        int c0, nce, k, b;
        decode(0, c0, nce, k, b);
#ifdef PCL_PROFILE
        if (p.prof & 128) {
            asm volatile("" ::"s"(k), "s"(b), "s"(c0));
            SP4_STAMP();
        }
#endif
        SP4_LANEPOS();
        const bool act = c < d;
        const double bs = half ? -1.0 : 1.0;
        SP4_SCALARS_DEF();
        double hh;
        sp4_cf cf;
        sp_cptr tab;
        scalars(k, b, hh, cf, tab);
#ifdef PCL_PROFILE
        if (p.prof & 128) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            SP4_STAMP();
        }
#endif
        {  // G itself: unit vectors (a block of its own: inside the loop the 27 constants are kept in a second register set)
            double *Po = Pt;
            double x[SPD];
#pragma unroll
            for (int i = 0; i < SPD; ++i) x[i] = (half == 0 && i == c) ? 1.0 : 0.0;
            if (act) sp4_product0_part(wave, x, sp4_lds_off(Po + own), sp4_lds_off(Po + oth), 1.0, bs, tab, cf);
            wave_lds_sync();
            sp4_arrive(sync + SP4_F_CO, lane);
            SP4_STAMP();
        }
#pragma unroll 1
        for (int j = 2; j <= q; ++j) {
            double *Po = first_tile(j - 1);
            const double *Pi = first_tile(j - 2);
            gave_up = sp4_wait(sync, SP4_F_CO, SP4_NPART * (j - 1), gave_up);  // every row of the previous power is in its tile
#ifdef PCL_PROFILE
            if (p.prof & 512) SP4_STAMP();  // (fine stamps of a cooperative level: wait | operand in registers | product | arrived)
#endif
            double x[SPD];
            if (act) {
#pragma unroll
                for (int i = 0; i < SPD; ++i) x[i] = Pi[own + i];
            }
#ifdef PCL_PROFILE
            if (p.prof & 512) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                SP4_STAMP();
            }
#endif
            if (j > npw && !borrow) {  // a ring shorter than q: the stream has folded the power this tile held ...
                for (int w = 0; w < SP4_NSTREAM; ++w) gave_up = sp4_wait(sync, SP4_F_C + w, j - npw, gave_up);
                if (npw == 1) {  // ... and with ONE tile every part has its operand in registers before a row is rewritten
                    wave_lds_sync();
                    sp4_arrive(sync + SP4_F_CX, lane);
                    gave_up = sp4_wait(sync, SP4_F_CX, SP4_NPART * (j - 1), gave_up);
                }
            }
            if (act) sp4_product0_part(wave, x, sp4_lds_off(Po + own), sp4_lds_off(Po + oth), 1.0, bs, tab, cf);
#ifdef PCL_PROFILE
            if (p.prof & 512) SP4_STAMP();
#endif
            wave_lds_sync();
            sp4_arrive(sync + SP4_F_CO, lane);
            SP4_STAMP();
        }
        __builtin_amdgcn_s_setprio(0);
    }
#endif
    if (wave < SP4_WLOAD) {
        // ================================== column waves: one chain each ====================================================
        SP4_SCALARS_DEF();
        if (wave == 0) {
            // ---- P: powers of G(u_k), one tile of the ring per power -----------------------------------------------------------
            if (!(p.v4_flags & 1)) __builtin_amdgcn_s_setprio(2);  // the stream waits for this chain
            // slice tickets: the builds follow the group's static visit order, jumping to the visit the dispatcher is at when that is ahead (the
            // intervals in between were exhausted before this workgroup got there); the controls and the step come from the dispatcher's ring
            int pv = -1;
            int it = coop ? 1 : 0;
            for (;; ++it) {
                int c0, nce, k, b;
                if (!tick) {
                    if (it >= n_my) break;
                    decode(it, c0, nce, k, b);
                } else {
                    // the next build: the visit of the oldest slice in flight if that is still ahead, else of the newest one, else the visit after the
                    // last build (the newest word first: it is written last, so the oldest read behind it is at least as new)
                    const int vb = __hip_atomic_load(sync + SP4_F_V2, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
                    const int va = __hip_atomic_load(sync + SP4_F_V, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
                    pv = pv < va ? va : (pv < vb ? vb : pv + 1);
                    if (pv >= n_vis || gave_up) break;
                    gave_up = sp4_wait(sync, SP4_F_VI, pv + 1, gave_up);
                    const int iv = grp + pv * n_groups;
                    b = p.batch == 1 ? 0 : iv / p.K;
                    k = iv - b * p.K;
                    c0 = 0;
                    nce = d;
                }
                SP4_LANEPOS();
                const bool act = c < d;
                double h;
                sp4_cf cf;
                sp_cptr tab;
                SP4_STAMP();
                if (!tick)
                    scalars(k, b, h, cf, tab);
                else {
                    const double *ur = uhr + (pv & (SP4_UHR - 1)) * 8;
                    SP4_SET_DCF(cf, (sp_cptr)(dcf_tab + (p.g0_batch_stride ? (long long)b * SP4NDCFP : 0)));
                    double u[SPM > 0 ? SPM : 1];
#pragma unroll
                    for (int l = 0; l < SPM; ++l) u[l] = ur[l];
                    h = ur[SPM];
                    SP4_SET_CF(cf, u, mg);
                    tab = (sp_cptr)(drift_tab + (p.g0_batch_stride ? (long long)b * SP4NDRIFT : 0));
                    if (lane == 0) pvis[it & 3] = pv;  // (published with this build's first power)
                }
                SP4_STAMP();
                const double bs = half ? -1.0 : 1.0;
                {  // P_1 = G I: the unit vectors never touch LDS
                    const int L = it * q;
                    double *Po = Pt + (L % npw) * SP4TILE;
                    double x[SPD];
#pragma unroll
                    for (int i = 0; i < SPD; ++i) x[i] = (half == 0 && i == c) ? 1.0 : 0.0;
                    for (int w = 0; w < SP4_NSTREAM; ++w) gave_up = sp4_wait(sync, SP4_F_C + w, L - npw + 1, gave_up);  // the stream has folded the power this tile held
                    if (act) sp4_product0(x, 0u, sp4_lds_off(Po + own), sp4_lds_off(Po + oth), 0.0, 1.0, bs, tab, cf);
                    sp4_post(sync + SP4_F_B, L + 1, lane);
                    SP4_STAMP();
                }
#pragma unroll 1
                for (int j = 2; j <= q; ++j) {
                    const int L = it * q + j - 1;
                    const double *Pi = Pt + ((L - 1) % npw) * SP4TILE;
                    double *Po = Pt + (L % npw) * SP4TILE;
                    for (int w = 0; w < SP4_NSTREAM; ++w) gave_up = sp4_wait(sync, SP4_F_C + w, L - npw + 1, gave_up);
                    double x[SPD];  // (read right before the product: 54 registers that nothing else should have to live beside)
                    if (act) {
#pragma unroll
                        for (int i = 0; i < SPD; ++i) x[i] = Pi[own + i];
                    }
#ifdef PCL_PROFILE
                    SP4_STAMP();
                    const int reps = (p.prof & 32) ? 17 : 1;  // the product's warm rate: sixteen more of it (same result)
                    for (int rep = 0; rep < reps; ++rep) {
                        if (act) sp4_product0(x, 0u, sp4_lds_off(Po + own), sp4_lds_off(Po + oth), 0.0, 1.0, bs, tab, cf);
                        if (rep == 0) SP4_STAMP();
                    }
                    SP4_STAMP();
#else
                    if (act) sp4_product0(x, 0u, sp4_lds_off(Po + own), sp4_lds_off(Po + oth), 0.0, 1.0, bs, tab, cf);
#endif
                    sp4_post(sync + SP4_F_B, L + 1, lane);
                    SP4_STAMP();
                }
            }
        } else if (wave <= 2) {
            // ---- W (delta) and V (d delta / dh) --------------------------------------------------------------------------------
            const bool isW = wave == 1;
            double *Xt = isW ? Wt : Vt;
            for (int it = 0; !no_chains; ++it) {
                int c0, nce, k, b;
                if (!chain_item(it, c0, nce, k, b)) break;
                SP4_LANEPOS();
                const unsigned oX = sp4_lds_off(Xt + own), oXx = sp4_lds_off(Xt + oth);
                const unsigned oD = sp4_lds_off(Dt + own), oS = sp4_lds_off(St + own);
                const bool act = c < nce;
                double h;
                sp4_cf cf;
                sp_cptr tab;
                SP4_STAMP();
                scalars(k, b, h, cf, tab);
                chains_may_start(it);
                gave_up = sp4_wait(sync, SP4_F_IN, it + 1, gave_up);
                outputs_taken(it);  // the previous item has left this tile
                SP4_STAMP();
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
                    const bool pp = isW && pingpong;  // W alternates between two tiles: this step reads `src`, writes `dst`
                    const double *Xsrc = pp && (s & 1) ? Walt : Xt;
                    double *Xdst = pp && !(s & 1) ? Walt : Xt;
                    if (isW) {  // every drive wave has gathered the level this product overwrites (two tiles: the level before it)
                        for (int l = 0; l < SPM; ++l) gave_up = sp4_wait(sync, SP4_F_G + l, it * q + s + (pp ? 0 : 1), gave_up);
                    }
                    const double alpha = sp4_uniform(((j & 1) ? -1.0 : 1.0) * p.pc[j] * (isW ? 1.0 : (double)j));
                    const double beta = sp4_uniform((!isW && j == 0) ? 1.0 : h);
                    SP4_STAMP();
                    double x[SPD];  // (read right before the product: 54 registers that nothing else should have to live beside)
                    if (act) {
#pragma unroll
                        for (int i = 0; i < SPD; ++i) x[i] = Xsrc[own + i];
                    }
                    if (act) sp4_product(x, (j & 1) ? oS : oD, pp ? sp4_lds_off(Xdst + own) : oX, pp ? sp4_lds_off(Xdst + oth) : oXx, alpha, beta, half ? -beta : beta, tab, cf);
                    SP4_STAMP();
                    if (isW && j >= 1) sp4_post(sync + SP4_F_W, it * q + s + 2, lane);
                }
                sp4_post(sync + (isW ? SP4_F_DW : SP4_F_DV), it + 1, lane);  // this wave's reads of D, S are complete
                sp4_post(sync + SP4_F_O + (isW ? 0 : 1), it + 1, lane);       // ... and its output vector
                sp4_arrive(sync + SP4_F_OA, lane);
                SP4_STAMP();
            }
        } else {
            // ---- dW_l (d delta / du_l) ----------------------------------------------------------------------------------------
            const int l = wave - 3;
            double *Xt = dWt + l * SP4TILE;
            for (int it = 0; !no_chains; ++it) {
                int c0, nce, k, b;
                if (!chain_item(it, c0, nce, k, b)) break;
                SP4_LANEPOS();
                SP4_STAMP();
                const unsigned oX = sp4_lds_off(Xt + own), oXx = sp4_lds_off(Xt + oth);
                const double sb = half ? 1.0 : -1.0;
                const bool act = c < nce;
                double h;
                sp4_cf cf;
                sp_cptr tab;
                scalars(k, b, h, cf, tab);
                // level q - 1: dW = h G_l W_q
                gave_up = sp4_wait(sync, SP4_F_W, it * q + 1, gave_up);
                outputs_taken(it);  // the previous item has left this tile
                if (act) {  // (step 0: W's level q stands in its own tile)
                    SP4_GATHER_SWITCH(l, Wt + own, Wt + oth, Xt + own, h, sb, mg)
                }
                wave_lds_sync();
                sp4_post(sync + SP4_F_G + l, it * q + 1, lane);
                const double hu = sp4_uniform(h);
#pragma unroll 1
                for (int s = 1; s < q; ++s) {
                    gave_up = sp4_wait(sync, SP4_F_W, it * q + s + 1, gave_up);
                    double x[SPD];
                    if (act) {
#pragma unroll
                        for (int i = 0; i < SPD; ++i) x[i] = Xt[own + i];
                    }
                    wave_lds_sync();  // (x is in registers before the gather rewrites the tile)
                    if (act) {
                        const double *Ws = (s & 1) ? Walt : Wt;  // W after s steps
                        SP4_GATHER_SWITCH(l, Ws + own, Ws + oth, Xt + own, h, sb, mg)
                    }
                    wave_lds_sync();
                    sp4_post(sync + SP4_F_G + l, it * q + s + 1, lane);
                    SP4_STAMP();
                    if (act) sp4_product(x, oX, oX, oXx, 1.0, hu, half ? -hu : hu, tab, cf);  // tile = h G_l W_old + h G dW_old
                    SP4_STAMP();
                }
                sp4_post(sync + SP4_F_O + 2 + l, it + 1, lane);
                sp4_arrive(sync + SP4_F_OA, lane);
                SP4_STAMP();
            }
        }
    } else if (wave == SP4_WLOAD) {
        // ================================== loader: D, S of every item (lane = row) ============================================
        constexpr int NB = 9;  // columns per batch of loads
        for (int it = 0; !no_chains; ++it) {
            int c0, nce, k, b;
            if (tick) {  // the chain pipeline's own ticket: an interval's chains go to a workgroup whose chain waves are free
                unsigned tk = 0;
                if (lane == 0) tk = __hip_atomic_fetch_add(p.tick + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const int iv = (int)__builtin_amdgcn_readfirstlane(tk);
                const bool end = iv >= p.batch * p.K || iv < 0 || gave_up;
                // (the writer is inside item it - 2 at the earliest: the slot of item it - 4 is free)
                if (lane == 0) cdesc[it & 3] = end ? -1 : iv;
                wave_lds_sync();
                sp4_post(sync + SP4_F_CI, it + 1, lane);
                if (end) {
                    ticket_leave(1u);
                    break;
                }
                b = p.batch == 1 ? 0 : iv / p.K;
                k = iv - b * p.K;
                c0 = 0;
                nce = d;
                // the chains follow the block front (a few visits ahead of it) instead of racing through the launch: their loads, products and
                // tail stores are spread over the launch, and an interval's tails are written when its blocks are
                if (!(p.v4_flags & 64)) gave_up = sp4_wait(sync, SP4_F_V, (iv - grp) / n_groups - 8, gave_up);
            } else if (!chain_item(it, c0, nce, k, b))
                break;
            SP4_STAMP();
            const double *zk = p.Z + (long long)b * p.z_batch_stride + (long long)k * p.z_dim + (p.x_off0 >= 0 ? p.x_off0 : p.x_offs[p.z_batch_stride ? 0 : b]) + (long long)c0 * n + (lane < n ? lane : 0);
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
            SP4_STAMP();
        }
    } else if (wave == SP4_WWRITE) {
        // ================================== writer: a finished item's delta and tail block, tiles -> memory ====================
        // One wave stores the item's nce (m + 1) n tail values (and its nce n residuals) as ONE contiguous run, 1 KiB per
        // instruction: written by the chains themselves, every store covered pieces of three 432-byte runs and neighbouring
        // runs came from different waves at different times (1.7 % of the bytes cost 10 % of the launch).
        // With the stream waves storing the tails (tail_mode 3) this wave only forms the reduce payload, when asked
        // (pcl_eval_jac_merit_dev): per state column <d delta/d u_l, lam>, <d delta/d h, lam>, <delta, lam> (lam = delta: half the
        // squared norm) while the vectors are still in their tiles -- the tails are never read back from memory.  Lane (half, c) adds
        // its half of column c in row order, then the two halves: bitwise repeatable and independent of the work split.
        for (int it = 0; !no_chains && (!tails_by_stream || p.mpart); ++it) {
            int c0, nce, k, b;
            if (!chain_item(it, c0, nce, k, b)) break;
            const long long bk = (long long)b * p.K + k;
            for (int w = 0; w < SP4_NOUT; ++w) gave_up = sp4_wait(sync, SP4_F_O + w, it + 1, gave_up);
            SP4_STAMP();
            if (p.mpart) {
                SP4_LANEPOS();
                const bool act = c < nce;
                double lam[SPD];
                if (p.mlam) {
                    const double *lg = p.mlam + bk * xd + (long long)(c0 + cc) * n + half * d;
#pragma unroll
                    for (int i = 0; i < SPD; ++i) lam[i] = act ? lg[i] : 0.0;
                } else {
#pragma unroll
                    for (int i = 0; i < SPD; ++i) lam[i] = Wres[own + i];
                }
#pragma unroll 1
                for (int l = 0; l < m + 2; ++l) {
                    const double *T = (l < m ? dWt + l * SP4TILE : (l == m ? Vt : Wres)) + own;
                    double a0 = 0.0, a1 = 0.0;
#pragma unroll
                    for (int i = 0; i < SPD; ++i) {
                        if (i & 1)
                            a1 = __builtin_fma(T[i], lam[i], a1);
                        else
                            a0 = __builtin_fma(T[i], lam[i], a0);
                    }
                    double acc = a0 + a1;
                    acc += __shfl_xor(acc, 32, 64);
                    if (act && half == 0) p.mpart[(bk * d + c0 + c) * (m + 2) + l] = (l == m + 1 && !p.mlam) ? 0.5 * acc : acc;
                }
            }
            if (!tails_by_stream && !no_tails) store_outputs(c0, nce, bk, 0, 1, p.tail_mode);
            wave_lds_sync();
            sp4_post(sync + SP4_F_OC, it + 1, lane);  // the chains may rewrite their tiles
            SP4_STAMP();
        }
    } else if (SP4_TICKETS && wave == SP4_WDISP) {
        // ================================== dispatcher (launches with slice tickets only) ======================================
        // Everything of the block pipeline that waits for memory: the slice tickets, each taken when every stream wave has issued the
        // previous slice's stores (taken earlier, a ticket is an address range reserved for later: 5-10 us per ticket held in the bare
        // store pattern), and the controls and steps of the group's next visits (64 visits per refill, two refills ahead).
        auto fetch_chunk = [&](int ch) {  // visits [64 ch, 64 ch + 64): lane = 8 visit + component (u_0 .. u_{m-1}, h)
            double val[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const int v = ch * 64 + r * 8 + (lane >> 3), comp = lane & 7;
                const int iv = grp + v * n_groups;
                val[r] = 0.0;
                if (v < n_vis && comp <= m) {
                    const int b = p.batch == 1 ? 0 : iv / p.K, k = iv - b * p.K;
                    val[r] = p.Z[(long long)b * p.z_batch_stride + (long long)k * p.z_dim + (comp < m ? p.u_off + comp : p.dt_off)];
                }
            }
#pragma unroll
            for (int r = 0; r < 8; ++r) uhr[((ch * 64 + r * 8 + (lane >> 3)) & (SP4_UHR - 1)) * 8 + (lane & 7)] = val[r];
            wave_lds_sync();
            sp4_post(sync + SP4_F_VI, min((ch + 1) * 64, n_vis), lane);
        };
        int next_chunk = 0, v = 0, items = 0, v_last = 0;
        for (; next_chunk < 2 && next_chunk * 64 < n_vis; ++next_chunk) fetch_chunk(next_chunk);
        for (;;) {
            SP4_STAMP();
            for (int w = 0; w < SP4_NSTREAM; ++w) gave_up = sp4_wait_soon(sync, SP4_F_TS + w, items - (p.tick_ahead == 1 ? 1 : 0), gave_up);
            SP4_STAMP();
            int sl = 0;
            for (; v < n_vis && !gave_up; ++v) {  // this visit's interval, or the next one that has a slice left
                unsigned t = 0;
                if (lane == 0) t = __hip_atomic_fetch_add(p.tick + 4 + grp + v * n_groups, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                sl = (int)__builtin_amdgcn_readfirstlane(t);
                if (sl >= 0 && sl < t_ipi) break;
            }
            SP4_STAMP();
            if (v >= n_vis || gave_up) break;
            // (the chunk a refill overwrites lies two behind the visit: nothing the stream or a useful build still reads)
            while (next_chunk * 64 < n_vis && next_chunk * 64 <= v + 64) {
                if (p.tick_ahead && next_chunk - 2 >= v_last / 64)  // (... nor the slice before this one, if it may still be in flight)
                    for (int w = 0; w < SP4_NSTREAM; ++w) gave_up = sp4_wait(sync, SP4_F_TS + w, items, gave_up);
                fetch_chunk(next_chunk++);
            }
            if (lane == 0) bdesc[items & 3] = 32 * v + sl;
            wave_lds_sync();
            sp4_post(sync + SP4_F_V, p.tick_ahead ? v_last : v, lane);
            sp4_post(sync + SP4_F_V2, v, lane);
            sp4_post(sync + SP4_F_BI, ++items, lane);
            v_last = v;
        }
        if (lane == 0) bdesc[items & 3] = -1;
        wave_lds_sync();
        for (int w = 0; w < SP4_NSTREAM; ++w) gave_up = sp4_wait(sync, SP4_F_TS + w, items, gave_up);  // (the last slice's powers are folded)
        sp4_post(sync + SP4_F_VI, n_vis, lane);
        sp4_post(sync + SP4_F_V, n_vis, lane);
        sp4_post(sync + SP4_F_V2, n_vis, lane);
        sp4_post(sync + SP4_F_BI, items + 1, lane);
        ticket_leave(no_chains ? 2u : 1u);
    } else {
        // ================================== stream waves ========================================================================
#ifdef PCL_PROFILE
        if (p.prof & 128) SP4_STAMP();
#endif
        __builtin_amdgcn_s_setprio(3);  // few instructions, each of them keeps the memory system busy
        constexpr int hn = n >> 1;
        constexpr int pstep = (64 * SP4_NSTREAM) / hn > 0 ? (64 * SP4_NSTREAM) / hn : 1;
        constexpr int NSP = (n + pstep - 1) / pstep;  // column positions per thread
#ifdef PCL_PROFILE
        bool dry_ = (p.prof & 256) != 0;  // experiment (results WRONG): the first item twice, the first pass without stores -- the second pass shows the warm timings
#endif
        // -B^+ and B^- of this thread's positions (slice tickets: kept across the slices of one visit)
        double bpr[NSP][2], bmr[NSP][2];
        int cur_v = -1, itS = 0;  // slice tickets: the visit the values belong to, the P wave's next build to look at
        for (int it = 0;; ++it) {
            int c0, nce, k, b, fb = it;
            bool refold = true;
            double h = 0.0;
            if (!tick) {
                if (it >= n_my) break;
                decode(it, c0, nce, k, b);
                h = ((sp_cptr)(p.Z + (long long)b * p.z_batch_stride + (long long)k * p.z_dim))[p.dt_off];
            } else {
                gave_up = sp4_wait_soon(sync, SP4_F_BI, it + 1, gave_up);
                const int t = __builtin_amdgcn_readfirstlane(__hip_atomic_load(bdesc + (it & 3), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
                if (t < 0 || gave_up) break;
                const int v = t >> 5, sl = t & 31, iv = grp + v * n_groups;
                b = p.batch == 1 ? 0 : iv / p.K;
                k = iv - b * p.K;
                c0 = sl * p.tick_cpi;
                nce = min(p.tick_cpi, d - c0);
                refold = v != cur_v;
                if (refold) {  // the build that holds this visit's powers: builds of visits this workgroup found exhausted are let go unread
                    for (;;) {
                        gave_up = sp4_wait(sync, SP4_F_B, itS * q + 1, gave_up);
                        if (gave_up || __builtin_amdgcn_readfirstlane(__hip_atomic_load(pvis + (itS & 3), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) >= v) break;
                        sp4_post(sync + SP4_F_C + (wave - SP4_WSTREAM), (itS + 1) * q, lane);
                        ++itS;
                    }
                    fb = itS++;
                    cur_v = v;
                    h = uhr[(v & (SP4_UHR - 1)) * 8 + m];
                }
            }
            // (an opaque copy per item: derived from `tid` directly, the tile addresses below are hoisted out of the item loop
            //  and spilled)
            SP4_STAMP();
            int stid = tid - 64 * SP4_WSTREAM;
            asm volatile("" : "+v"(stid));  // (after the cooperative products: nothing derived from it lives beside their registers)
            const int pi = 2 * (stid % hn), pj0 = stid / hn;
            const bool pact = pj0 < pstep;
            // the values, folded power by power as the P wave publishes them.  Entry (i, j) of the
            // n x n iso matrix [[A, -B], [B, A]] whose first d columns are a tile: j >= d mirrors into column j - d, rows i < d
            // from row i + d with the sign flipped, rows i >= d from row i - d.
            if (refold) {
            int toff[NSP][2];
            unsigned flip = 0;  // bit 2 r + e: the mirrored entry changes sign
#pragma unroll
            for (int r = 0; r < NSP; ++r) {
                const int j = min(pj0 + pstep * r, n - 1);
                const bool mir = j >= d;
                const int jj = mir ? j - d : j;
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const int i = pi + e;
                    const int ii = mir ? (i < d ? i + d : i - d) : i;
                    toff[r][e] = jj * SP4CS + ii;
                    flip |= (mir && i < d) ? 1u << (2 * r + e) : 0u;
                    const double id = i == j ? 1.0 : 0.0;
                    bpr[r][e] = -id;
                    bmr[r][e] = id;
                }
            }
            double hp = 1.0, hm = 1.0;
#ifdef PCL_PROFILE
            if (p.prof & 128) SP4_STAMP();
#endif
#pragma unroll 1
            for (int j = 1; j <= q; ++j) {
                const int L = fb * q + j - 1;
                const double *T = (coop && it == 0) ? first_tile(j - 1) : Pt + (L % npw) * SP4TILE;
                hp *= h;
                hm *= -h;
                const double cp = p.pc[j] * hp, cm = p.pc[j] * hm;
                if (coop && it == 0)
                    gave_up = sp4_wait(sync, SP4_F_CO, SP4_NPART * j, gave_up);  // every part of this power (cooperative first item)
                else
                    gave_up = sp4_wait(sync, SP4_F_B, L + 1, gave_up);
                double v[NSP][2];
#pragma unroll
                for (int r = 0; r < NSP; ++r)
#pragma unroll
                    for (int e = 0; e < 2; ++e) v[r][e] = T[toff[r][e]];
#pragma unroll
                for (int r = 0; r < NSP; ++r)
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        const double g = (flip >> (2 * r + e)) & 1u ? -v[r][e] : v[r][e];
                        bpr[r][e] = __builtin_fma(-cp, g, bpr[r][e]);
                        bmr[r][e] = __builtin_fma(cm, g, bmr[r][e]);
                    }
                wave_lds_sync();
                sp4_post(sync + SP4_F_C + (wave - SP4_WSTREAM), L + 1, lane);  // the P wave may rewrite this tile
            }
            }
            SP4_STAMP();
            // tail_mode 3: this wave's share of the item's residuals and tails goes into its own store stream as soon as every chain
            // has finished the item -- checked between two columns of blocks, never waited for before the last block is out (the
            // chains run several times faster than the stream; behind the last block the stores would lengthen a one-item launch)
            bool tails_out = !(tails_by_stream && !no_chains);
#ifdef PCL_PROFILE
            if (dry_) tails_out = true;
#endif
            auto try_tails = [&](bool wait) {
                if (tails_out) return;
                if (wait) {
                    for (int w = 0; w < SP4_NOUT; ++w) gave_up = sp4_wait(sync, SP4_F_O + w, it + 1, gave_up);
                } else {
                    // (one word instead of the chains' eight: up to eight dependent LDS round trips between two columns of stores -- 0.1-0.4 us of a one-trajectory launch at orders 8 and 10)
                    const bool ready = __hip_atomic_load(sync + SP4_F_OA, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) >= (it + 1) * SP4_NOUT;
                    if (!ready) return;
                }
                if (!no_tails) store_outputs(c0, nce, (long long)b * p.K + k, wave - SP4_WSTREAM, SP4_NSTREAM, 0);
                wave_lds_sync();
                sp4_post(sync + SP4_F_TS + (wave - SP4_WSTREAM), it + 1, lane);
                tails_out = true;
            };
            {
                int cbeg = c0, cend = c0 + nce;
                if (p.compact) {  // unique blocks only: the piece that holds column 0 writes the single copy
                    cbeg = 0;
                    cend = (c0 == 0) ? 1 : 0;
                }
                // Two round-robin slices of an odd number of columns (one trajectory of config 3: 14 + 13): the -B^+ block of the first slice's
                // last column stays there, its B^- block goes to the second slice -- 27 blocks each instead of 28 and 26: 0.2-0.35 us of a
                // 28-32 us launch (v4_flags & 32 switches it off)
                int half_col = -1;  // the column whose two blocks are shared between the interval's two slices
                if (!(p.v4_flags & 32) && !p.contig && !p.compact && !tick && p.S == 2 && (d & 1)) {
                    half_col = p.nc - 1;
                    if (c0 > 0) cbeg = half_col;
                }
                // nt 3: write-through on every other workgroup = every other XCD (workgroups are dispatched round-robin over the XCDs; the even ones since round 5).  Plain stores
                // leave up to 32 MB dirty in the L2s, written back behind the kernel's end (2 us of the gap between two launches); write-through
                // everywhere makes the kernel itself 0.9 us longer.  Half of the XCDs each way: one trajectory 25.9 -> 25.3 us at order 4
                // (contiguous halves of the XCDs, a quarter or all of them: 25.7).  A performance hint only: any placement gives the same values.
                // (round 5: the EVEN workgroups -- the XCDs the memory side serves first -- take the write-through stores: 24.67 against 25.06 us per one-trajectory launch
                //  at order 4 with the odd ones, three alternating rounds on one box, twice; 3 or 5 of 8 XCDs, 2 of 8, 6 of 8: 25.0-25.2)
                const int nt_b = p.nt == 3 ? ((bx & 1) ? 0 : 2) : p.nt;
                double *o = p.jac + ((long long)b * p.K + k) * p.jac_per + (long long)cbeg * nn + pi;
                for (int cq = cbeg; cq < cend; ++cq, o += nn) {
                    // (tick_ahead 2: the dispatcher asks for the next slice while this one's last column goes out)
                    if (tick && p.tick_ahead == 2 && cq == cend - 1) sp4_post(sync + SP4_F_TS + (wave - SP4_WSTREAM), it + 1, lane);
#ifdef PCL_PROFILE
                    if (dry_ && cq >= cbeg + 2) break;
                    if (pact && !no_blocks && !dry_) {
#else
                    if (pact && !no_blocks) {
#endif
                        const bool sp_ = cq != half_col || c0 == 0, sm_ = cq != half_col || c0 > 0;
#pragma unroll
                        for (int r = 0; r < NSP; ++r) {  // all of -B+'s rows, then all of B-'s: one 23 KB run each (interleaved: +0.3 us per launch)
                            const int j = pj0 + pstep * r;
                            if (j < n && sp_) store2(o + n * j, bpr[r][0], bpr[r][1], nt_b);
                        }
#pragma unroll
                        for (int r = 0; r < NSP; ++r) {
                            const int j = pj0 + pstep * r;
                            if (j < n && sm_) store2(o + blk + n * j, bmr[r][0], bmr[r][1], nt_b);
                        }
                    }
                    if (!(p.v4_flags & 2)) try_tails(false);
#ifdef PCL_PROFILE
                    if ((p.prof & 128) && cq < cbeg + 2) SP4_STAMP();
#endif
                }
            }
            try_tails(true);
            if (tick && p.tick_ahead != 2) sp4_post(sync + SP4_F_TS + (wave - SP4_WSTREAM), it + 1, lane);  // this slice's stores are issued: the dispatcher takes the next ticket
            SP4_STAMP();
#ifdef PCL_PROFILE
            if (dry_) {
                dry_ = false;
                --it;
                continue;
            }
            if (p.prof & 16) {  // ... and gone
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                SP4_STAMP();
            }
#endif
        }
    }
    if (tick && wave >= SP4_WSTREAM && wave < SP4_WSTREAM + SP4_NSTREAM) sp4_post(sync + SP4_F_C + (wave - SP4_WSTREAM), 0x3fffffff, lane);  // (a build the P wave is still at)
    if (gave_up && lane == 0) {  // a wait gave up: the context's error word (the next entry point or pcl_sync returns PCL_EINTERNAL) ...
        if (p.err) __hip_atomic_fetch_or(p.err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        p.jac[0] = __builtin_nan("");  // ... and visible in the values instead of a hung device
    }
#ifdef PCL_PROFILE
    if (wall_ && lane == 0) atomicMax((unsigned long long *)(wall_ + 2), (unsigned long long)__builtin_amdgcn_s_memrealtime());
#endif
}
#if SP4_RESIDENT
}
#endif

#if SP4_RESIDENT
// ------------------------------------------------------------------------------------------------------------------------------
// RESIDENT evaluator (pcl_resident_start / _post / _wait / _stop): the same workgroups, one per CU, stay on the device and run the
// evaluation above once per request -- no launch, no kernel boundary between two evaluations of one solver iteration after the other.
// The trajectory, residual and values arrays are fixed at start (KParams); a request is a number:
//     hbox (host memory, mapped):   [0] requests posted by the host | [1] stop | [16] evaluations complete (written by the device) | [17] workgroup 0 has left
//     dbox (device memory):         [0] requests forwarded | [1] stop | [2] evaluations complete | [16], [17] workgroups that have finished
//                                   an evaluation, by its parity | [32] workgroups gone | [40], [41] address of hbox | [42] the launch's first evaluation |
//                                   [43] idle limit, 100 MHz ticks | [44] the evaluation the launch ends before
// Workgroup 0 reads the host's words (one reader on the bus, not 256) and forwards them; every workgroup starts evaluation e once it is
// posted and evaluation e - 2 is complete everywhere: a workgroup that finishes early starts the next evaluation while others still store
// (its own column range only: nobody writes anybody else's values), and never runs more than one ahead, so two arrival words suffice.
// Between two evaluations: the trajectory may have been rewritten (by a copy engine, by another queue) -- vector L1 / L2 lines of other
// agents' data and the scalar cache are invalidated before, the workgroup's stores are written back behind.  A workgroup leaves when
// told to, after `idle_ticks` (100 MHz) without a request (workgroup 0 decides, the others follow its stop word; their own limit is eight
// times as long), or after max_evals evaluations -- the kernel ENDS by itself whatever the host does.
// ------------------------------------------------------------------------------------------------------------------------------
extern "C" {
static __device__ __forceinline__ double *sp4_dyn_lds() {
    extern __shared__ double lds[];
    return lds;
}
}
// (Nothing of the request loop may live in registers across the evaluation: the evaluation's resident coefficients take every scalar register
//  there is -- the first version kept its pointers and counters in ten of them and the evaluation ran 9 us longer, 85 more spilled scalars.
//  The loop's state is in LDS behind the evaluation's words and in dbox [40 ...]: the host words' address, the idle limit, the first evaluation.)
#define SP4R_FLAG (2 * SP4_NTILES * SP4TILE + SP4_SYNC_WORDS)  // int index into the dynamic LDS: [0] go | [1] the evaluation's number | [2], [3] posted / complete as read beside the last arrival | [4] ... are there
extern "C" __global__ __launch_bounds__(64 * (SP4_NWAVES + 1)) void pcl_fused_sparse_resident(const KParams *pdev, const double *__restrict__ drift_tab, const double *__restrict__ mags_, const double *__restrict__ dcf_tab,
                                                                                           unsigned *dbox) {
    typedef const KParams __attribute__((address_space(4))) *kp_cptr;
    const int v4_flags = ((kp_cptr)pdev)->v4_flags;
    if (threadIdx.x == 0) {
        ((volatile int *)sp4_dyn_lds())[SP4R_FLAG + 1] = (int)__hip_atomic_load(dbox + 42, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ((volatile int *)sp4_dyn_lds())[SP4R_FLAG + 4] = 0;
    }
    for (;;) {
        if (threadIdx.x == 0) {
            volatile int *flag = (volatile int *)sp4_dyn_lds() + SP4R_FLAG;
            const unsigned e = (unsigned)flag[1];
            unsigned *hbox = (unsigned *)(((unsigned long long)__hip_atomic_load(dbox + 41, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) << 32) | __hip_atomic_load(dbox + 40, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            const unsigned idle_ticks = __hip_atomic_load(dbox + 43, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            int go = 0;
            const long long t0 = (long long)__builtin_amdgcn_s_memrealtime();
            const long long limit = blockIdx.x == 0 ? (long long)idle_ticks : 8LL * idle_ticks;
            for (int poll = 0;; ++poll) {
                const bool early = poll == 0 && flag[4] != 0;  // the words as read beside the last arrival (older than a fresh read, never newer: a request they show is there)
                unsigned posted = early ? (unsigned)flag[2] : __hip_atomic_load(dbox + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (blockIdx.x == 0 && (int)(posted - e) <= 0) {  // the forwarder: the host's words, read only when this workgroup would wait (a read takes 1-2 us)
                    const unsigned hp = __hip_atomic_load(hbox + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    if ((int)(hp - posted) > 0) {
                        __hip_atomic_store(dbox + 0, hp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        posted = hp;
                    } else if (__hip_atomic_load(hbox + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM))
                        __hip_atomic_store(dbox + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                const unsigned done = early ? (unsigned)flag[3] : __hip_atomic_load(dbox + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if ((int)(posted - e) > 0 && (int)(done + 1 - e) >= 0) {
                    go = 1;
                    break;
                }
                if (__hip_atomic_load(dbox + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
                if ((long long)__builtin_amdgcn_s_memrealtime() - t0 > limit) {
                    if (blockIdx.x == 0) __hip_atomic_store(dbox + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // (a request that arrives now is the next start's)
                    break;
                }
                if (poll < 16)
                    __builtin_amdgcn_s_sleep(4);
                else
                    __builtin_amdgcn_s_sleep(16);
            }
            if (go && !(v4_flags & 128)) {
                if (v4_flags & 512)
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                else
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");  // other agents' writes (the trajectory) since the last evaluation
                __builtin_amdgcn_s_dcache_inv();                    // ... which this kernel reads through the scalar cache
            }
            if ((v4_flags & 2048) && e - __hip_atomic_load(dbox + 42, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < 16)  // (debugging: 100 MHz stamps per evaluation and workgroup)
                ((long long *)(dbox + 64))[((e - __hip_atomic_load(dbox + 42, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) * 256 + (blockIdx.x & 255)) * 4 + 0] = (long long)__builtin_amdgcn_s_memrealtime();
            flag[0] = go;
        }
        __syncthreads();
        const int go = ((volatile int *)sp4_dyn_lds())[SP4R_FLAG];
        __syncthreads();  // (the evaluation's first act is to write LDS words)
        if (!go) break;
        {  // (the parameter block is read where it is used, in every evaluation anew: hoisted out of the request loop its fields are live everywhere)
            kp_cptr pp = (kp_cptr)pdev;
            asm volatile("" : "+s"(pp));
            sp4_fused_body(*(const KParams *)pp, drift_tab, mags_, dcf_tab);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's stores have reached the L2
        __syncthreads();
        if (threadIdx.x == 0) {
            volatile int *flag = (volatile int *)sp4_dyn_lds() + SP4R_FLAG;
            const unsigned e = (unsigned)flag[1];
            long long *st_ = nullptr;
            if (v4_flags & 2048) {
                const unsigned e0 = __hip_atomic_load(dbox + 42, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (e - e0 < 16) st_ = (long long *)(dbox + 64) + ((e - e0) * 256 + (blockIdx.x & 255)) * 4;
            }
            if (st_) st_[2] = (long long)__builtin_amdgcn_s_memrealtime();
            if (!(v4_flags & 256)) {
                if (v4_flags & 1024)
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                else
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");  // ... and memory
            }
            // (relaxed behind the release fence: the next request's polls are in flight beside it -- one round trip to the memory side instead of
            //  two.  What the last arrival publishes is already in memory: every workgroup's fence stands before its arrival.)
            const unsigned a = __hip_atomic_fetch_add(dbox + 16 + (e & 1u), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned posted_next = __hip_atomic_load(dbox + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned done_next = __hip_atomic_load(dbox + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            flag[2] = (int)posted_next, flag[3] = (int)done_next;
            if (a + 1 == gridDim.x) {  // the last workgroup of evaluation e
                unsigned *hbox = (unsigned *)(((unsigned long long)__hip_atomic_load(dbox + 41, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) << 32) | __hip_atomic_load(dbox + 40, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                __hip_atomic_store(dbox + 16 + (e & 1u), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(dbox + 2, e + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(hbox + 16, e + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            }
            if (st_) st_[3] = (long long)__builtin_amdgcn_s_memrealtime();
            flag[1] = (int)(e + 1);
            flag[4] = 1;
            if (e + 1 == __hip_atomic_load(dbox + 44, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) flag[0] = 0;  // the launch's last evaluation (2^30 after its first: the kernel ends whatever happens)
        }
        __syncthreads();
        if (!((volatile int *)sp4_dyn_lds())[SP4R_FLAG]) break;
    }
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(dbox + 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (blockIdx.x == 0) {  // the host starts the kernel again before it posts
            unsigned *hbox = (unsigned *)(((unsigned long long)__hip_atomic_load(dbox + 41, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) << 32) | __hip_atomic_load(dbox + 40, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            __hip_atomic_store(hbox + 17, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}
#endif

// ------------------------------------------------------------------------------------------------------------------------------
// Residual only (what the solver calls in every line-search trial), any order: delta = W_0 of the recursion above, q products.
// One WAVE per interval and no cooperation between waves (SP4E_NW independent waves per workgroup, three tiles each: D, S, W):
// every coefficient of the product is resident (or streamed from the launch-invariant drift table), so there is no per-interval
// value table to write, flush and read back -- the round-2 kernel of this role spent most of an interval on exactly that.
// ------------------------------------------------------------------------------------------------------------------------------
#define SP4E_NW 4
extern "C" __global__ __launch_bounds__(64 * SP4E_NW) void pcl_eval_sparse4_kernel(const KParams p, const double *__restrict__ drift_tab, const double *__restrict__ mags_, const double *__restrict__ dcf_tab) {
    extern __shared__ double lds[];
    constexpr int d = SPD, n = SPN, q = SP4Q;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    double *Dt = lds + wave * 3 * SP4TILE, *St = Dt + SP4TILE, *Xt = St + SP4TILE;
    sp_cptr magc = (sp_cptr)mags_;
    double mg[SP4NMAG];
#pragma unroll
    for (int g = 0; g < SP4NMAG; ++g) mg[g] = magc[g];
    const int n_items = p.batch * p.K;
    const long long xd = (long long)n * d;
    for (int item = blockIdx.x * SP4E_NW + wave; item < n_items; item += gridDim.x * SP4E_NW) {
        const int k = item % p.K, b = item / p.K;
        int ln_ = lane;
        asm volatile("" : "+v"(ln_));
        const int half = ln_ >> 5, c = ln_ & 31;
        const bool act = c < d;
        const int own = (act ? c : 0) * SP4CS + half * d, oth = (act ? c : 0) * SP4CS + (1 - half) * d;
        sp_cptr zc = (sp_cptr)(p.Z + (long long)b * p.z_batch_stride + (long long)k * p.z_dim);
        double u[SPM > 0 ? SPM : 1];
#pragma unroll
        for (int l = 0; l < SPM; ++l) u[l] = zc[p.u_off + l];
        const double h = zc[p.dt_off];
        sp4_cf cf;
        SP4_SET_CF(cf, u, mg);
        SP4_SET_DCF(cf, (sp_cptr)(dcf_tab + (p.g0_batch_stride ? (long long)b * SP4NDCFP : 0)));
        sp_cptr tab = (sp_cptr)(drift_tab + (p.g0_batch_stride ? (long long)b * SP4NDRIFT : 0));
        // the interval's states, lane = row (coalesced), every load in flight at once (one memory round trip) -> D, S tiles [column][row]
        {
            const double *zk = p.Z + (long long)b * p.z_batch_stride + (long long)k * p.z_dim + (p.x_off0 >= 0 ? p.x_off0 : p.x_offs[p.z_batch_stride ? 0 : b]) + (ln_ < n ? ln_ : 0);
            const double *zn = zk + p.z_dim;
            constexpr int NB = SPD;
#pragma unroll
            for (int cb = 0; cb < SPD; cb += NB) {
                double xc[NB], xn[NB];
#pragma unroll
                for (int j = 0; j < NB; ++j)
                    if (cb + j < SPD) {
                        xc[j] = zk[(cb + j) * n];
                        xn[j] = zn[(cb + j) * n];
                    }
                if (ln_ < n) {
#pragma unroll
                    for (int j = 0; j < NB; ++j)
                        if (cb + j < SPD) {
                            Dt[(cb + j) * SP4CS + ln_] = xn[j] - xc[j];
                            St[(cb + j) * SP4CS + ln_] = xn[j] + xc[j];
                        }
                }
            }
        }
        wave_lds_sync();
        {  // level q
            const double *Yq = (q & 1) ? St : Dt;
            const double aq = ((q & 1) ? -1.0 : 1.0) * p.pc[q];
            if (act) {
                double y[SPD];
#pragma unroll
                for (int i = 0; i < SPD; ++i) y[i] = Yq[own + i];
#pragma unroll
                for (int i = 0; i < SPD; ++i) Xt[own + i] = aq * y[i];
            }
            wave_lds_sync();
        }
        const unsigned oX = sp4_lds_off(Xt + own), oXx = sp4_lds_off(Xt + oth), oD = sp4_lds_off(Dt + own), oS = sp4_lds_off(St + own);
        const double hu = sp4_uniform(h);
#pragma unroll 1
        for (int s = 0; s < q; ++s) {
            const int j = q - 1 - s;
            const double alpha = sp4_uniform(((j & 1) ? -1.0 : 1.0) * p.pc[j]);
            double x[SPD];
            if (act) {
#pragma unroll
                for (int i = 0; i < SPD; ++i) x[i] = Xt[own + i];
            }
            if (act) sp4_product(x, (j & 1) ? oS : oD, oX, oXx, alpha, hu, half ? -hu : hu, tab, cf);
        }
        wave_lds_sync();
        {  // tile -> memory: the interval's n d residuals are one contiguous run, two rows per lane
            double *dst = p.delta + (long long)item * xd;
            for (int e2 = ln_; e2 < d * d; e2 += 64) {
                const int cl = e2 / d, r0 = 2 * (e2 - cl * d);
                const double *src = Xt + cl * SP4CS + r0;
                store2(dst + 2 * e2, src[0], src[1], 0);
            }
        }
        wave_lds_sync();  // (the tiles are rewritten by this wave's next interval)
    }
}

#if SP4_COOP
// ------------------------------------------------------------------------------------------------------------------------------
// Residual only, SMALL launches (fewer intervals than CUs: a line-search trial on one trajectory): the four waves of a workgroup share
// ONE interval -- every product in four row ranges (sp4_product_part: the rows keep their instruction sequences, the same bits as the
// one-wave kernel above), the result of a level double-buffered so that a level costs one workgroup barrier.  One wave per interval is
// the longer chain there (8.8 us per launch for 99 intervals: q cold products of 3-4 k cycles each behind the loads).
extern "C" __global__ __launch_bounds__(64 * SP4_NPART) void pcl_eval_sparse4c_kernel(const KParams p, const double *__restrict__ drift_tab, const double *__restrict__ mags_, const double *__restrict__ dcf_tab) {
    extern __shared__ double lds[];
    constexpr int d = SPD, n = SPN, q = SP4Q;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    double *Dt = lds, *St = Dt + SP4TILE, *X0 = St + SP4TILE, *X1 = X0 + SP4TILE;
    sp_cptr magc = (sp_cptr)mags_;
    double mg[SP4NMAG];
#pragma unroll
    for (int g = 0; g < SP4NMAG; ++g) mg[g] = magc[g];
    int pr0, pr1;
    sp4_part_rows(wave, pr0, pr1);
    const int n_items = p.batch * p.K;
    const long long xd = (long long)n * d;
    for (int item = blockIdx.x; item < n_items; item += gridDim.x) {
        const int k = item % p.K, b = item / p.K;
        int ln_ = lane;
        asm volatile("" : "+v"(ln_));
        const int half = ln_ >> 5, c = ln_ & 31;
        const bool act = c < d;
        const int own = (act ? c : 0) * SP4CS + half * d, oth = (act ? c : 0) * SP4CS + (1 - half) * d;
        sp_cptr zc = (sp_cptr)(p.Z + (long long)b * p.z_batch_stride + (long long)k * p.z_dim);
        double u[SPM > 0 ? SPM : 1];
#pragma unroll
        for (int l = 0; l < SPM; ++l) u[l] = zc[p.u_off + l];
        const double h = zc[p.dt_off];
        sp4_cf cf;
        SP4_SET_CF(cf, u, mg);
        SP4_SET_DCF(cf, (sp_cptr)(dcf_tab + (p.g0_batch_stride ? (long long)b * SP4NDCFP : 0)));
        sp_cptr tab = (sp_cptr)(drift_tab + (p.g0_batch_stride ? (long long)b * SP4NDRIFT : 0));
        {  // the interval's states, lane = row, the waves share the columns: every load of a wave in flight at once
            const double *zk = p.Z + (long long)b * p.z_batch_stride + (long long)k * p.z_dim + (p.x_off0 >= 0 ? p.x_off0 : p.x_offs[p.z_batch_stride ? 0 : b]) + (ln_ < n ? ln_ : 0);
            const double *zn = zk + p.z_dim;
            constexpr int NB = (SPD + SP4_NPART - 1) / SP4_NPART;
            double xc[NB], xn[NB];
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                const int cl = wave + SP4_NPART * j;
                if (cl < SPD) {
                    xc[j] = zk[cl * n];
                    xn[j] = zn[cl * n];
                }
            }
            if (ln_ < n) {
#pragma unroll
                for (int j = 0; j < NB; ++j) {
                    const int cl = wave + SP4_NPART * j;
                    if (cl < SPD) {
                        Dt[cl * SP4CS + ln_] = xn[j] - xc[j];
                        St[cl * SP4CS + ln_] = xn[j] + xc[j];
                    }
                }
            }
        }
        __syncthreads();
        {  // level q: this wave's rows
            const double *Yq = (q & 1) ? St : Dt;
            const double aq = ((q & 1) ? -1.0 : 1.0) * p.pc[q];
            if (act)
                for (int i = pr0; i < pr1; ++i) X0[own + i] = aq * Yq[own + i];
        }
        __syncthreads();
        const unsigned oD = sp4_lds_off(Dt + own), oS = sp4_lds_off(St + own);
        const double hu = sp4_uniform(h);
        double *Xc = X0, *Xn = X1;
#pragma unroll 1
        for (int s = 0; s < q; ++s) {
            const int j = q - 1 - s;
            const double alpha = sp4_uniform(((j & 1) ? -1.0 : 1.0) * p.pc[j]);
            double x[SPD];
            if (act) {
#pragma unroll
                for (int i = 0; i < SPD; ++i) x[i] = Xc[own + i];
                sp4_product_part(wave, x, (j & 1) ? oS : oD, sp4_lds_off(Xn + own), sp4_lds_off(Xn + oth), alpha, hu, half ? -hu : hu, tab, cf);
            }
            __syncthreads();  // every row of the level is in the other tile
            double *t_ = Xc;
            Xc = Xn;
            Xn = t_;
        }
        {  // tile -> memory: the interval's n d residuals are one contiguous run, two rows per lane, a quarter per wave
            double *dst = p.delta + (long long)item * xd;
            for (int e2 = tid; e2 < d * d; e2 += 64 * SP4_NPART) {
                const int cl = e2 / d, r0 = 2 * (e2 - cl * d);
                const double *src = Xc + cl * SP4CS + r0;
                store2(dst + 2 * e2, src[0], src[1], 0);
            }
        }
        __syncthreads();  // (the tiles are rewritten by the next interval)
    }
}
#endif
