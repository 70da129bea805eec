// pcl_kernel_jac_sparse.hpp -- the COLUMN WORK of the fused residual + Jacobian path, PATTERN-COMPILED (DESIGN.md section 4.8;
// generated source only).  Per interval and state column: delta, d delta / d dt and the m vectors d delta / d u_l -- everything of
// the Jacobian except the replicated -B+ / B- blocks, which fused kernel 3 streams in a launch of its own (every workgroup in the
// stream role).  With Y1 = -(h/2) S + (h^2/12) G D and Y2 = -(1/2) S + (h/6) G D:
//     delta = D + G Y1,    d/ddt = G Y2,    d/du_l = G_l Y1 + G (c2 G_l D)            (c2 = h^2/12)
// i.e. products with G(u_k) (sp_g: straight-line multiply-adds, coefficients in scalar registers from the interval's value table)
// and with the drives' few entries (sp_gl_<l>: resident magnitudes), instead of 14 padded matrix-core passes per interval.
// One WORKGROUP of m + 2 waves per interval, ONE output vector per wave (wave 0: d/ddt, wave 1: delta, wave 2 + l: d/du_l): lane
// (half, c) reads its half of column c of D and S from the staging tiles, forms G D and its own Y, then its own product(s) -- two
// long products per wave, no data passes between the waves -- and leaves through its own LDS tile (lane = column -> lane = row) as
// runs of SPN consecutive doubles.  The NEXT interval is staged by waves 0 and 1 (the two without the drives' small products) after
// their own outputs, with their registers free: half of the next interval's states each (lane = row) -> D = X_{k+1} - X_k and
// S = X_{k+1} + X_k in the other pair of staging tiles; the load latencies hide behind the drive waves' extra work.  The value
// tables of G(u_k) on the union pattern (in sp_g's emission order) come from pcl_sparse_values_kernel, launched before this kernel.
// All waves of a workgroup read the SAME value table (5 KB: it stays in the scalar cache; one interval per wave overflowed it).
// One workgroup barrier per interval.  LDS: (m + 2) tiles + 2 x 2 staging tiles.
#pragma once

extern "C" __global__ __launch_bounds__(512) void pcl_jac_sparse_kernel(const KParams p, const double *gvals_, const double *__restrict__ glv) {
    extern __shared__ double lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nwv = __builtin_amdgcn_readfirstlane((int)blockDim.x >> 6);  // m + 2
    const int half = lane >> 5, c = lane & 31;
    const bool act = c < SPD;
    const int cc = act ? c : 0;
    const double sgn = half ? -1.0 : 1.0;
    const int own = cc * SPCS + half * SPD;
    double *Stage = lds + (SPM + 2) * SPTILE;  // [2][D | S]
    double *T = lds + wave * SPTILE;
    double *Tl = T + own, *To = T + cc * SPCS + (1 - half) * SPD;
    const int n_items = p.batch * p.K;
    const int n_my = n_items > (int)blockIdx.x ? (n_items - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x : 0;

    sp_mags mg;  // the distinct magnitudes of the drives' entries: scalar registers for the whole launch
    SP_LOAD_MAGS(mg, glv);
    const long long blk = p.compact ? (long long)SPN * SPN : (long long)SPD * SPN * SPN;

#ifdef PCL_PROFILE
    int stamp_ = 0;  // cycle stamps of workgroup 0, second interval: 16 slots for waves 0, 1, 2 and the last one
    const int sw_ = wave < 3 ? wave : (wave == nwv - 1 ? 3 : -1);
#define SPJ_STAMP()                                                                                                                      \
    do {                                                                                                                                 \
        if (p.dbg && blockIdx.x == 0 && lane == 0 && sw_ >= 0 && it == 1 && stamp_ < 16) p.dbg[16 * sw_ + stamp_++] = (long long)__builtin_amdgcn_s_memtime(); \
    } while (0)
#else
#define SPJ_STAMP() do { } while (0)
#endif
    // Staging of interval `it` of this workgroup by two waves (this one is number `ws` of them): D, S tiles (lane = row) of its share
    // of the state columns and its share of the value table.  Run by waves 0 and 1 (the two without the small products) after
    // their own outputs, i.e. with their registers free: the load latencies hide behind the drive waves' extra work.
    auto stage = [&](int it, int ws) {
        const int item = blockIdx.x + it * gridDim.x;
        const int k = item % p.K, b = item / p.K;
        const double *zb = p.Z + (long long)b * p.z_batch_stride + (long long)k * p.z_dim;
        const double *zk = zb + p.x_offs[p.z_batch_stride ? 0 : b];
        double *Sd = Stage + (it & 1) * 2 * SPTILE, *Ss = Sd + SPTILE;
        constexpr int NPC = (SPD + 1) / 2;          // columns per staging wave
        const int w = ws;  // (m >= 1: there are always two staging waves)
        {   // the states: half of the columns, all requested together
            double xc[NPC], xn[NPC];
#pragma unroll
            for (int j = 0; j < NPC; ++j) {
                const int q = w + 2 * j;
                xc[j] = xn[j] = 0.0;
                if (q < SPD && lane < SPN) {
                    xc[j] = zk[SPN * q + lane];
                    xn[j] = zk[p.z_dim + SPN * q + lane];
                }
            }
#ifdef PCL_PROFILE
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (p.dbg && blockIdx.x == 0 && lane == 0 && sw_ >= 0 && it == 2 && stamp_ < 16) p.dbg[16 * sw_ + stamp_++] = (long long)__builtin_amdgcn_s_memtime();  // loads arrived
#endif
#pragma unroll
            for (int j = 0; j < NPC; ++j) {
                const int q = w + 2 * j;
                if (q < SPD && lane < SPN) {
                    Sd[SPCS * q + lane] = xn[j] - xc[j];
                    Ss[SPCS * q + lane] = xn[j] + xc[j];
                }
            }
        }
        wave_lds_sync();
    };
    if (n_my > 0 && wave < 2) stage(0, wave);
    __syncthreads();  // interval 0 is staged
    for (int it = 0; it < n_my; ++it) {
        const int item = blockIdx.x + it * gridDim.x;
        const int k = item % p.K, b = item / p.K;
        const double *zb = p.Z + (long long)b * p.z_batch_stride + (long long)k * p.z_dim;
        double *jt = p.jac + (long long)item * p.jac_per + 2 * blk + lane;  // tail: column q: [d/du_0 .. d/du_{m-1} | d/ddt], SPN doubles each
        const double h = zb[p.dt_off];
        const double c1 = 0.5 * h, c2 = h * h * (1.0 / 12.0), h6 = h * (1.0 / 6.0);
        SPJ_STAMP();  // top
        const double *Sd = Stage + (it & 1) * 2 * SPTILE, *Ss = Sd + SPTILE;
        double D[SPD], Y[SPD];
#pragma unroll
        for (int r = 0; r < SPD; ++r) {
            D[r] = Sd[own + r];
            Y[r] = Ss[own + r];  // S for now
        }
        sp_cptr g = (sp_cptr)(gvals_ + (long long)item * SPNZP);
        SPJ_STAMP();  // D, S read
        SP_PREFETCH_G(g);  // all lines of the table at once: left to the products, the cold misses are taken one after the other
        SPJ_STAMP();  // table lines touched
        // G D, then this wave's Y: wave 0: Y2 = -S/2 + (h/6) G D; the others: Y1 = -c1 S + c2 G D
        const double ys = wave == 0 ? 0.5 : c1, yg = wave == 0 ? h6 : c2;
        sp_g(D, g, -sgn, half, [&](int r, double v) { Y[r] = __builtin_fma(yg, v, -(ys * Y[r])); });
        SPJ_STAMP();  // G D
        // ONE copy of the second long product for all roles (the instruction cache holds 64 KB; with a copy per role the kernel
        // was 51 KB and the waves of a workgroup, each in its own copy, kept evicting each other's code: 11 k cycles per product
        // instead of 3 k): every wave leaves the FIRST part of its vector in its tile and the input of the product in X, then
        // T += G X.       wave 0: T = 0, X = Y2        wave 1: T = D, X = Y1        wave 2 + l: T = G_l Y1, X = c2 G_l D
        double *dst;
        long long col_stride = (long long)(SPM + 1) * SPN;
        if (wave == 0) {
#pragma unroll
            for (int r = 0; r < SPD; ++r) T[own + r] = 0.0;
            dst = jt + (long long)SPM * SPN;
        } else if (wave == 1) {
#pragma unroll
            for (int r = 0; r < SPD; ++r) T[own + r] = D[r];
            dst = p.delta + (long long)item * SPXD + lane;
            col_stride = SPN;
        } else {
            // the small products sit in a wave-uniform switch whose cases exchange data with the rest through LDS only (no
            // register webs merged behind it)
            const int l = wave - 2;
            if (act) {  // (the lanes beyond column d - 1 repeat column 0: harmless for stores, not for atomic adds)
                SP_GL_SWITCH(l, D, mg, -sgn, Tl, To);
            }
            wave_lds_sync();
            double P[SPD];
#pragma unroll
            for (int r = 0; r < SPD; ++r) P[r] = c2 * T[own + r];
            wave_lds_sync();
            if (act) {
                SP_GL_SWITCH(l, Y, mg, -sgn, Tl, To);
            }
#pragma unroll
            for (int r = 0; r < SPD; ++r) Y[r] = P[r];  // X = c2 G_l D
            dst = jt + (long long)l * SPN;
        }
        wave_lds_sync();
        {
            // the first part comes back into registers in ONE batch of reads (a read-modify-write of the tile per output was one
            // exposed LDS round trip per row: 11 k cycles for this product instead of 3 k); the lanes beyond column d - 1 repeat
            // column 0 and store the same values
            double B[SPD];
#pragma unroll
            for (int r = 0; r < SPD; ++r) B[r] = T[own + r];
            wave_lds_sync();
            sp_g(Y, sp_opaque(g), -sgn, half, [&](int r, double v) { T[own + r] = B[r] + v; });
        }
        wave_lds_sync();
        SPJ_STAMP();  // own product(s)
        // waves 0 and 1 request the next interval's states BEFORE their output stores (a wave's memory operations complete in
        // order: behind 27 stores into the saturated write path the loads took 20 k cycles)
        __builtin_amdgcn_sched_barrier(0);
        if (it + 1 < n_my && wave < 2) stage(it + 1, wave);
        __builtin_amdgcn_sched_barrier(0);
        // the column vectors leave the tile as runs of SPN doubles: lane = row
        if (lane < SPN && (wave != 1 || p.delta)) {
            const double *Tr = T + lane;
#pragma unroll
            for (int q = 0; q < SPD; ++q) dst[col_stride * q] = Tr[SPCS * q];
        }
        wave_lds_sync();
        SPJ_STAMP();  // staged
        __syncthreads();  // interval it + 1 is staged, every wave is done with the staging tiles of interval it
        SPJ_STAMP();  // barrier passed
    }
}
