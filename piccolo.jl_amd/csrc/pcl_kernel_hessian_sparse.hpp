// pcl_kernel_hessian_sparse.hpp -- Hessian of the Lagrangian, PATTERN-COMPILED version (order 4).
// Included by generated source only (pcl_codegen.hpp): SPD (Hilbert dimension d <= 32), SPM (drives <= 6), SPN = 2 SPD,
// SPNZ / SPNZP and the straight-line functions sp_gt / sp_g / sp_glt_<l> / sp_gltdot<l> are defined before this file.
//
// Every product of the Hessian acts on a state column from the left, so a lane owns ONE column and keeps it in registers:
//     lane = (half, c): half 0 holds the top rows a of column c, half 1 the bottom rows b; both run the same instructions
//     U = A^T x, V = B^T x on their half x and complete  top = U(0) + V(1), bottom = U(1) - V(0)  with one cross-half swap
// (T = [[A, -B], [B, A]]: every generator is an exact iso(.) image).  Coefficients of G(u_k) come from a per-interval value
// table through scalar loads (constant address space); the drives' distinct magnitudes live in scalar registers.
// Roles (one workgroup of SPM + 2 waves per interval, a contiguous range of intervals per workgroup):
//     wave 0      A1 = G^T M -> registers and LDS tile (counter: ready), A2 = G^T A1, outputs d2/dh dX, <A2, D>, the scalar entries
//     wave 1      loader: the next interval's mu, x_k, x_{k+1} travel in its registers during the interval; it also touches
//                 the next interval's value table so that the other waves' scalar loads hit
//     wave 2 + l  P_l = G_l^T M, <P_l, S>, <P_l, G_j D> for every j, R_l = G_l^T A1, Q_l = G^T P_l consumed row by row:
//                 <Q_l + R_l, D> and the two output vectors d2/du_l dX
// Outputs leave through one LDS tile per wave (lane = column -> lane = row): one column of SPN consecutive doubles per store.
// No workgroup barrier in the interval loop (point-to-point LDS counters, below); the 28 scalar entries are 16-lane row sums
// (DPP) added in a fixed order one interval later: repeatable bits.
#pragma once

#define SPXD (SPN * SPD)
// LDS tiles are [column][row] with an ODD column stride (SPN + 1 doubles): the lanes of a half wave, one column each, hit 32
// different 8-byte bank pairs (with stride SPN -- even -- half of all LDS cycles of the kernel were bank conflicts)
#define SPCS (SPN + 1)
#define SPTILE (SPCS * SPD)
#define SPNSC ((SPM + 1) * (SPM + 2) / 2)
#define SPNPAIR (SPM * (SPM + 1) / 2)

// acc += x y, pinned where it is written: the compiler sinks plain dot-product arithmetic to the end of the interval (where the
// sums are used) and keeps every operand alive -- in scratch -- until then.
static __device__ __forceinline__ void sp_fmac(double &acc, double x, double y) { asm volatile("v_fmac_f64 %0, %1, %2" : "+v"(acc) : "v"(x), "v"(y)); }
template <int J, int N, class F>
static __device__ __forceinline__ void sp_static_for(F f) {
    if constexpr (J < N) {
        f(sp_ic<J>{});
        sp_static_for<J + 1, N>(f);
    }
}
// The coefficient table is read through the constant address space (scalar loads).  Two products with the same table would
// share ONE set of loads (every coefficient live between them: hundreds of SGPRs spilled to VGPR lanes); an opaque copy of the
// pointer per product keeps each product's loads next to its multiply-adds.
static __device__ __forceinline__ sp_cptr sp_opaque(sp_cptr q) {
    asm volatile("" : "+s"(q));
    return q;
}
// Point-to-point synchronisation inside the workgroup: monotonic LDS counters (workgroup-scope release / acquire).  No workgroup
// barrier in the interval loop: a wave starts the next interval as soon as ITS inputs are there, so the waves drift apart and
// the LDS-heavy phases of some overlap the arithmetic of others (behind two barriers per interval all seven chains ran in
// lock-step: every resource below 50 %).  Dependencies only point backwards (earlier stage, same or earlier interval).
#define SP_SYNC_WORDS 16
enum { SP_IN_READY = 0, SP_RD_DONE, SP_A1_READY, SP_A1_DONE, SP_COMB_DONE, SP_FIN = 8 /* one word per drive wave: intervals finished */ };
static __device__ __forceinline__ bool sp_wait(int *sync, int word, int target) {
    // Bounded: a logic error must not hang the device.  Returns true when it gave up; the state wave turns that into NaN scalar
    // entries (the drive waves' progress words are the last link of every dependency chain), so the failure shows in the values.
    int it = 0;
    for (; __hip_atomic_load(sync + word, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < target && it < (1 << 22); ++it) __builtin_amdgcn_s_sleep(1);
    return it >= (1 << 22);
}
static __device__ __forceinline__ void sp_post(int *w, int value, int lane) {  // after wave_lds_sync(): this wave's LDS traffic is complete
    if (lane == 0) __hip_atomic_store(w, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
}
static __device__ __forceinline__ void sp_arrive(int *w, int lane) {
    if (lane == 0) __hip_atomic_fetch_add(w, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
}

extern "C" __global__ __launch_bounds__(64 * (SPM + 2)) void pcl_hess_sparse_kernel(const KParams p, const double *__restrict__ gvals_, const double *__restrict__ glv_) {
    extern __shared__ double lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // scalar: the role branches are uniform
    const int half = lane >> 5, c = lane & 31;
    const bool act = c < SPD;
    const int cc = act ? c : 0;
    const double sgn = half ? -1.0 : 1.0;
    const int own = cc * SPCS + half * SPD, oth = cc * SPCS + (1 - half) * SPD;
    double *Mt = lds, *Dt = Mt + SPTILE, *St = Dt + SPTILE, *A1t = St + SPTILE, *Stg = A1t + SPTILE;  // Stg: SPM + 1 staging tiles
    constexpr int NSUM = (SPM * (SPM + 2) + 1) * 4;  // [SPM][SPM + 2][4] drive-wave sums per 16-lane row | [1][4] <A2, D>
    double *scal = Stg + (SPM + 1) * SPTILE;         // two copies (interval parity)
    int *sync = (int *)(scal + 2 * NSUM);
    sp_cptr glv = (sp_cptr)glv_;

    const int n_items = p.batch * p.K;
    const int item_lo = (int)((long long)n_items * blockIdx.x / gridDim.x), item_hi = (int)((long long)n_items * (blockIdx.x + 1) / gridDim.x);
    if (item_lo >= item_hi) return;

    auto step_of = [&](int item) {
        const int k = item % p.K, b = item / p.K;
        return p.Z[(long long)b * p.z_batch_stride + (long long)k * p.z_dim + p.dt_off];
    };
    if (tid < SP_SYNC_WORDS) sync[tid] = 0;
    __syncthreads();  // the only workgroup barrier
#ifdef PCL_PROFILE
    int stamp_ = 0;  // cycle stamps of workgroup 0: 16 slots per wave (dbg[16 wave + i])
#define SP_STAMP()                                                                                                        \
    do {                                                                                                                  \
        if (p.dbg && blockIdx.x == 0 && lane == 0 && wave < 4 && stamp_ < 16) p.dbg[16 * wave + stamp_++] = (long long)__builtin_amdgcn_s_memtime(); \
    } while (0)
    if (p.dbg && tid == 0 && blockIdx.x < PCL_DBG_WG) p.dbg[64 + 2 * blockIdx.x] = (long long)__builtin_amdgcn_s_memrealtime();
#define SP_END()                                                                                                                         \
    do {                                                                                                                                 \
        if (p.dbg && lane == 0 && wave == 0 && blockIdx.x < PCL_DBG_WG) p.dbg[64 + 2 * blockIdx.x + 1] = (long long)__builtin_amdgcn_s_memrealtime(); \
    } while (0)
#else
#define SP_STAMP() do { } while (0)
#define SP_END() do { } while (0)
#endif
    // tile -> global: lane = row, one column (SPN consecutive doubles) per store instruction; every address is a per-lane base plus an
    // immediate (index arithmetic on `lane + 64 i` gets hoisted out of the interval loop: 23 registers per tile, spilled).  Row pairs
    // with 16-byte stores measured 5 % slower (A/B on one box, lab/probes/hess_ab.py).
    auto flush = [&](const double *T, double *out) {
        wave_lds_sync();
        if (lane < SPN) {
            const double *Tl = T + lane;
            double *ol = out + lane;
#pragma unroll
            for (int q = 0; q < SPD; ++q) {
                if (p.nt)
                    __builtin_nontemporal_store(Tl[SPCS * q], ol + SPN * q);
                else
                    ol[SPN * q] = Tl[SPCS * q];
            }
        }
        wave_lds_sync();
    };
    // The three roles run their own loops (the register allocation of one role does not carry the other roles' live values).
    if (wave == 1) {
        // ---- loader ---------------------------------------------------------------------------------------------------------
        double pm[SPD], pxn[SPD], pxc[SPD];  // lane = row, one column per register
        auto request = [&](int item) {
            const int k = item % p.K, b = item / p.K;
            const double *zk = p.Z + (long long)b * p.z_batch_stride + (long long)k * p.z_dim + (p.x_off0 >= 0 ? p.x_off0 : p.x_offs[p.z_batch_stride ? 0 : b]);
            const double *mu = p.mu + ((long long)b * p.K + k) * SPXD;
            const double *mul = mu + lane, *zl = zk + lane, *zn = zk + p.z_dim + lane;
#pragma unroll
            for (int q = 0; q < SPD; ++q) {
                pm[q] = pxn[q] = pxc[q] = 0.0;
                if (lane < SPN) {
                    pm[q] = mul[SPN * q];
                    pxc[q] = zl[SPN * q];
                    pxn[q] = zn[SPN * q];
                }
            }
        };
        request(item_lo);
        for (int item = item_lo; item < item_hi; ++item) {
            const int seq = item - item_lo;
            SP_PREFETCH_G((sp_cptr)(gvals_ + (long long)item * SPNZP));  // the scalar cache is warm when the other waves arrive
            sp_wait(sync, SP_RD_DONE, (SPM + 1) * seq);                   // every reader is done with the previous interval's inputs
            SP_STAMP();
            if (lane < SPN) {
                double *Ml = Mt + lane, *Dl = Dt + lane, *Sl = St + lane;
#pragma unroll
                for (int q = 0; q < SPD; ++q) {
                    Ml[SPCS * q] = pm[q];
                    Dl[SPCS * q] = pxn[q] - pxc[q];
                    Sl[SPCS * q] = pxn[q] + pxc[q];
                }
            }
            wave_lds_sync();
            sp_post(sync + SP_IN_READY, seq + 1, lane);
            SP_STAMP();
            if (item + 1 < item_hi) request(item + 1);
        }
    } else if (wave == 0) {
        // ---- state wave -----------------------------------------------------------------------------------------------------
        double hn = step_of(item_lo);
        double *T = Stg + SPM * SPTILE;
        double hp = 0.0;
        auto combine = [&](int sq, double hh) {  // the 28 scalar entries of interval sq: row sums added in a fixed order
            const double c2 = hh * hh * (1.0 / 12.0), h6 = hh * (1.0 / 6.0);
            const double *sc = scal + (sq & 1) * NSUM;
            double *H = p.hess + (long long)(item_lo + sq) * p.hess_per;
            // the drive waves' sums of that interval are in LDS (one progress word per drive wave: nothing stops a fast wave from
            // finishing the next interval before a slow one finishes this one, so a shared arrival counter would lie)
            bool gave_up = false;
#pragma unroll
            for (int j = 0; j < SPM; ++j) gave_up |= sp_wait(sync, SP_FIN + j, sq + 1);
            wave_lds_sync();
            if (lane < SPNSC) {
                auto rows4 = [&](int e) { return ((sc[4 * e] + sc[4 * e + 1]) + sc[4 * e + 2]) + sc[4 * e + 3]; };
                double v;
                if (lane < SPNPAIR) {
                    int i = 0;
                    while ((i + 1) * (i + 2) / 2 <= lane) ++i;
                    const int j = lane - i * (i + 1) / 2;
                    v = c2 * (rows4(i * (SPM + 2) + j) + rows4(j * (SPM + 2) + i));
                } else if (lane < SPNPAIR + SPM) {
                    const int j = lane - SPNPAIR;
                    v = __builtin_fma(h6, rows4(j * (SPM + 2) + SPM + 1), -0.5 * rows4(j * (SPM + 2) + SPM));
                } else {
                    v = rows4(SPM * (SPM + 2)) * (1.0 / 6.0);
                }
                if (gave_up) v = __builtin_nan("");
                H[lane] = v;
            }
            wave_lds_sync();
            sp_post(sync + SP_COMB_DONE, sq + 1, lane);  // this copy of the sums may be rewritten (interval sq + 2)
        };
        for (int item = item_lo; item < item_hi; ++item) {
            const int seq = item - item_lo;
            const long long bk = item;  // = b K + k
            const double h = hn;
            const double c2 = h * h * (1.0 / 12.0), h6 = h * (1.0 / 6.0);
            double *H = p.hess + bk * p.hess_per;
            double *H4 = H + SPNSC + (long long)SPM * SPXD, *H6 = H4 + SPXD + (long long)SPM * SPXD;
            double *sc = scal + (seq & 1) * NSUM;
            sp_cptr g = (sp_cptr)(gvals_ + bk * SPNZP);
            sp_wait(sync, SP_IN_READY, seq + 1);
            SP_STAMP();
            if (item + 1 < item_hi) hn = step_of(item + 1);
            double A1[SPD];
            {  // A1 = G^T M: registers (input of the second product, outputs) and the A1 tile (the drive waves' R_l)
                double x[SPD];
#pragma unroll
                for (int r = 0; r < SPD; ++r) x[r] = Mt[own + r];
                sp_wait(sync, SP_A1_DONE, SPM * seq);  // the drive waves have read the previous interval's A1
                sp_gt(x, g, sgn, half, [&](int c, double v) {
                    A1[c] = v;
                    A1t[own + c] = v;  // (inactive lanes repeat column 0: the same values to the same addresses)
                });
            }
            wave_lds_sync();
            sp_post(sync + SP_A1_READY, seq + 1, lane);
            SP_STAMP();
            double s0 = 0.0, s1 = 0.0, s2 = 0.0, o6[SPD];
            {  // A2 = G^T A1, consumed row by row: <A2, D>, d2/dh dX_k -> staging tile, d2/dh dX_{k+1} -> registers
                double dq[SPD];
#pragma unroll
                for (int r = 0; r < SPD; ++r) dq[r] = Dt[own + r];
                wave_lds_sync();
                sp_arrive(sync + SP_RD_DONE, lane);  // this wave's reads of the interval's inputs are complete
                sp_gt(A1, sp_opaque(g), sgn, half, [&](int c, double v) {
                    if (c % 3 == 0)
                        sp_fmac(s0, v, dq[c]);
                    else if (c % 3 == 1)
                        sp_fmac(s1, v, dq[c]);
                    else
                        sp_fmac(s2, v, dq[c]);
                    const double x = -0.5 * A1[c], y = h6 * v;
                    T[own + c] = x - y;
                    o6[c] = x + y;
                });
            }
            SP_STAMP();
            flush(T, H4);
#pragma unroll
            for (int r = 0; r < SPD; ++r) T[own + r] = o6[r];
            flush(T, H6);
            SP_STAMP();
            {
                const double s = row16_sum(act ? (s0 + s1) + s2 : 0.0);
                if ((lane & 15) == 0) sc[(SPM * (SPM + 2)) * 4 + (lane >> 4)] = s;
            }
            // the scalar entries of the PREVIOUS interval (its drive waves finished long ago: no wait on the critical path)
            if (seq > 0) combine(seq - 1, hp);
            hp = h;
            SP_STAMP();
        }
        combine(item_hi - 1 - item_lo, hp);
        SP_END();
    } else {
        // ---- drive wave l ---------------------------------------------------------------------------------------------------
        // One copy of the role for all drive waves (the instruction cache holds 64 KB): the two small products with G_l^T sit in
        // a wave-uniform switch whose cases read and write LDS only, so no register webs are merged behind it.
        const int l = wave - 2;
        sp_mags mg;  // the distinct magnitudes of the drives' entries: scalar registers for the whole launch
        SP_LOAD_MAGS(mg, glv);
        double hn = step_of(item_lo);
        double *T = Stg + l * SPTILE;
        for (int item = item_lo; item < item_hi; ++item) {
            const int seq = item - item_lo;
            const long long bk = item;
            const double h = hn;
            const double c1 = 0.5 * h, c2 = h * h * (1.0 / 12.0);
            double *H = p.hess + bk * p.hess_per;
            double *H3 = H + SPNSC + (long long)l * SPXD, *H5 = H3 + (long long)(SPM + 1) * SPXD;
            double *sc = scal + (seq & 1) * NSUM;
            sp_cptr g = (sp_cptr)(gvals_ + bk * SPNZP);
            sp_wait(sync, SP_IN_READY, seq + 1);
            SP_STAMP();
#ifdef PCL_PROFILE
            if (((p.prof & 2) && wave >= 6) || ((p.prof & 4) && wave >= 4)) {  // experiments: fewer drive waves (wrong results)
                sp_arrive(sync + SP_RD_DONE, lane);
                sp_arrive(sync + SP_A1_DONE, lane);
                sp_post(sync + SP_FIN + l, seq + 1, lane);
                continue;
            }
#endif
            if (item + 1 < item_hi) hn = step_of(item + 1);
            double t[SPM + 2];
            double P[SPD], R[SPD];
            {  // P_l = G_l^T M (this half's rows)
                double x[SPD];
#pragma unroll
                for (int r = 0; r < SPD; ++r) x[r] = Mt[own + r];
                // (the results leave the switch through the wave's tile -- register webs merged behind a switch are spilled -- and
                //  the halves are completed there: U is written, the other half adds -sgn V to it with an LDS atomic)
                if (act) {  // (the lanes beyond column d - 1 repeat column 0: harmless for stores, not for atomic adds)
                    SP_GLT_SWITCH(l, x, mg, sgn, T + own, T + oth)
                }
            }
            wave_lds_sync();
#pragma unroll
            for (int r = 0; r < SPD; ++r) P[r] = T[own + r];
            SP_STAMP();
            {  // <P_l, S>  (operands first: a load next to each pinned multiply-add is one LDS round trip per row)
                double sv[SPD];
#pragma unroll
                for (int r = 0; r < SPD; ++r) sv[r] = St[own + r];
                double s0 = 0.0, s1 = 0.0, s2 = 0.0;
#pragma unroll
                for (int r = 0; r < SPD; r += 3) {
                    sp_fmac(s0, P[r], sv[r]);
                    if (r + 1 < SPD) sp_fmac(s1, P[r + 1], sv[r + 1]);
                    if (r + 2 < SPD) sp_fmac(s2, P[r + 2], sv[r + 2]);
                }
                t[SPM] = (s0 + s1) + s2;
            }
            SP_STAMP();
            double down[SPD];
#pragma unroll
            for (int r = 0; r < SPD; ++r) down[r] = Dt[own + r];
            {  // <P_l, G_j D> = <G_j^T P_l, D>: this lane's part is U . D_own - sgn V . D_other
                double doth[SPD];
#pragma unroll
                for (int r = 0; r < SPD; ++r) doth[r] = Dt[oth + r];
                wave_lds_sync();
                sp_arrive(sync + SP_RD_DONE, lane);  // this wave's reads of the interval's inputs are complete
                sp_static_for<0, SPM>([&](auto jc) {
                    constexpr int j = decltype(jc)::value;
                    t[j] = sp_gltdot<j>(P, down, doth, mg, sgn);
                });
            }
            SP_STAMP();
            // R_l = G_l^T A1 (after the dot products: their loads are this wave's last reads of the interval's inputs, and the sooner
            // every reader is done the sooner the loader refills the input tiles -- 4 % at 8 trajectories per launch, 7 % at 16)
            sp_wait(sync, SP_A1_READY, seq + 1);
            {
                double x[SPD];
#pragma unroll
                for (int r = 0; r < SPD; ++r) x[r] = A1t[own + r];
                wave_lds_sync();
                sp_arrive(sync + SP_A1_DONE, lane);
                if (act) {
                    SP_GLT_SWITCH(l, x, mg, sgn, T + own, T + oth)
                }
            }
            wave_lds_sync();
#pragma unroll
            for (int r = 0; r < SPD; ++r) R[r] = T[own + r];
            SP_STAMP();
            {  // Q_l = G^T P_l, consumed row by row: <Q_l + R_l, D>, d2/du_l dX_k -> the wave's tile, d2/du_l dX_{k+1} -> R's registers
                double s0 = 0.0, s1 = 0.0, s2 = 0.0;
                sp_gt(P, g, sgn, half, [&](int c, double v) {
                    const double q = v + R[c];
                    if (c % 3 == 0)
                        sp_fmac(s0, q, down[c]);
                    else if (c % 3 == 1)
                        sp_fmac(s1, q, down[c]);
                    else
                        sp_fmac(s2, q, down[c]);
                    const double pl = -c1 * P[c], kt = c2 * q;
                    T[own + c] = pl - kt;
                    R[c] = pl + kt;
                });
                t[SPM + 1] = (s0 + s1) + s2;
            }
            SP_STAMP();
            flush(T, H3);
#pragma unroll
            for (int r = 0; r < SPD; ++r) T[own + r] = R[r];
            flush(T, H5);
            sp_wait(sync, SP_COMB_DONE, seq - 1);  // the sums of interval seq - 2 (same copy) have been combined
#pragma unroll
            for (int j = 0; j < SPM + 2; ++j) {
                const double s = row16_sum(act ? t[j] : 0.0);
                if ((lane & 15) == 0) sc[(l * (SPM + 2) + j) * 4 + (lane >> 4)] = s;
            }
            wave_lds_sync();
            sp_post(sync + SP_FIN + l, seq + 1, lane);
            SP_STAMP();
        }
    }
}

// Per-interval value table of G(u_k) on the union pattern:  gvals[(b K + k) SPNZP + q] = G0_b[pos_q] + sum_l u_l G_l[pos_q]
extern "C" __global__ __launch_bounds__(256) void pcl_sparse_values_kernel(const KParams p, const int *__restrict__ pos, const double *__restrict__ coef, double *__restrict__ gvals) {
    const int item = blockIdx.x, k = item % p.K, b = item / p.K;
    const double *zk = p.Z + (long long)b * p.z_batch_stride + (long long)k * p.z_dim;
    const double *G0b = p.G0 + (long long)b * p.g0_batch_stride;
    for (int q = threadIdx.x; q < SPNZP; q += 256) {
        double v = 0.0;
        if (q < SPNZ) {
            v = G0b[pos[q]];
#pragma unroll
            for (int l = 0; l < SPM; ++l) v = __builtin_fma(zk[p.u_off + l], coef[q * SPM + l], v);
        }
        gvals[(long long)item * SPNZP + q] = v;
    }
}
