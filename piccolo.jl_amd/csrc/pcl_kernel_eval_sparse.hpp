// pcl_kernel_eval_sparse.hpp -- residual only, PATTERN-COMPILED version (round 2; superseded by pcl_eval_sparse4_kernel wherever kernel 4's tiles fit; generated source only).
// delta_k = D - G Y,  Y = (h/2) S - (h^2/12) G D   (= D - (h/2) G S + (h^2/12) G^2 D with two products instead of three).
// One WAVE per interval, no cooperation between waves (up to eight independent streams per workgroup): lane (half, c) holds its half
// of column c of D and S in registers, applies G(u_k) twice as straight-line multiply-adds (sp_g) and leaves through its own
// LDS tile (lane = column -> lane = row).  The value table of G(u_k) on the union pattern is written by the wave itself
// (five entries per lane, vector stores straight to L2) and read back through the scalar cache: every interval has its own
// lines, nobody has loaded them before in this launch.
#pragma once

extern "C" __global__ __launch_bounds__(512) void pcl_eval_sparse_kernel(const KParams p, double *gvals_, const int *__restrict__ pos_, const double *__restrict__ coef_) {
    extern __shared__ double lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, c = lane & 31;
    const bool act = c < SPD;
    const int cc = act ? c : 0;
    const double sgn = half ? -1.0 : 1.0;
    const int own = cc * SPCS + half * SPD;
    double *T = lds + wave * SPTILE;

    constexpr int NQ = (SPNZP + 63) / 64;
    int qpos[NQ];
    double qcoef[NQ][SPM];
#pragma unroll
    for (int j = 0; j < NQ; ++j) {
        const int q = lane + 64 * j;
        qpos[j] = q < SPNZ ? pos_[q] : -1;
#pragma unroll
        for (int i = 0; i < SPM; ++i) qcoef[j][i] = q < SPNZ ? coef_[q * SPM + i] : 0.0;
    }
    const int n_items = p.batch * p.K;
    const int nw = blockDim.x >> 6;  // waves per workgroup: the host spreads the intervals over the CUs first
    for (int item = blockIdx.x * nw + wave; item < n_items; item += gridDim.x * nw) {
        const int k = item % p.K, b = item / p.K;
        const double *zb = p.Z + (long long)b * p.z_batch_stride + (long long)k * p.z_dim;
        const double *zk = zb + p.x_offs[p.z_batch_stride ? 0 : b];
        double *gv = gvals_ + (long long)item * SPNZP;
        // the interval's states (lane = row, one column per load) and its value table
        double xc[SPD], xn[SPD];
        {
            const double *zl = zk + lane, *zn = zk + p.z_dim + lane;
#pragma unroll
            for (int q = 0; q < SPD; ++q) {
                xc[q] = xn[q] = 0.0;
                if (lane < SPN) {
                    xc[q] = zl[SPN * q];
                    xn[q] = zn[SPN * q];
                }
            }
        }
        const double h = zb[p.dt_off];
        {
            const double *G0b = p.G0 + (long long)b * p.g0_batch_stride;
            double u[SPM];
#pragma unroll
            for (int i = 0; i < SPM; ++i) u[i] = zb[p.u_off + i];
#pragma unroll
            for (int j = 0; j < NQ; ++j) {
                const int q = lane + 64 * j;
                double v = 0.0;
                if (qpos[j] >= 0) {
                    v = G0b[qpos[j]];
#pragma unroll
                    for (int i = 0; i < SPM; ++i) v = __builtin_fma(u[i], qcoef[j][i], v);
                }
                if (q < SPNZP) __hip_atomic_store(gv + q, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        const double c1 = 0.5 * h, c2 = h * h * (1.0 / 12.0);
        // D, S: lane = row -> lane = column through the wave's tile
        double D[SPD], S[SPD];
        if (lane < SPN) {
#pragma unroll
            for (int q = 0; q < SPD; ++q) T[SPCS * q + lane] = xn[q] - xc[q];
        }
        wave_lds_sync();
#pragma unroll
        for (int r = 0; r < SPD; ++r) D[r] = T[own + r];
        wave_lds_sync();
        if (lane < SPN) {
#pragma unroll
            for (int q = 0; q < SPD; ++q) T[SPCS * q + lane] = xn[q] + xc[q];
        }
        wave_lds_sync();
#pragma unroll
        for (int r = 0; r < SPD; ++r) S[r] = T[own + r];
        // the table's stores are complete (the readers sit behind the same L2), then the scalar loads may start
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        sp_cptr g = (sp_cptr)gv;
        SP_PREFETCH_G(g);  // all lines of the table at once: left to the products, 20 cold misses are taken one after the other
        double Y[SPD];
        sp_g(D, g, -sgn, half, [&](int r, double v) { Y[r] = __builtin_fma(-c2, v, c1 * S[r]); });
        wave_lds_sync();  // (this wave's reads of S from the tile are complete)
        sp_g(Y, sp_opaque(g), -sgn, half, [&](int r, double v) { T[own + r] = D[r] - v; });
        wave_lds_sync();
        if (lane < SPN) {
            double *ol = p.delta + (long long)item * SPXD + lane;
            const double *Tl = T + lane;
#pragma unroll
            for (int q = 0; q < SPD; ++q) ol[SPN * q] = Tl[SPCS * q];
        }
        wave_lds_sync();
    }
}
