// pcl_device_common.hpp -- kernel parameter block and device helpers shared by every kernel family of libpiccolo_hip
// (included by piccolo_hip.hip only; gfx950).
#pragma once

// debug_timing buffer: 64 phase stamps of workgroup 0, then per-workgroup start / end times (s_memrealtime, 100 MHz) of kernel 3
#define PCL_DBG_WG 1024
#define PCL_DBG_WORDS (64 + 2 * PCL_DBG_WG)

#define PCL_NSP 8   // B^{+-} value pairs per thread of a 256-thread group: (n*n/2) / 256 <= 8 for n <= 64

typedef double double4_t __attribute__((ext_vector_type(4)));
typedef double double2_t __attribute__((ext_vector_type(2)));

// ------------------------------------------------------------------------------------------
// Kernel parameters
// ------------------------------------------------------------------------------------------
struct KParams {
    const double *Z;       // trajectory buffer(s), knot-major
    const double *mu;      // multipliers (Hessian kernel)
    double *delta;         // may be null
    double *jac;           // full or compact Jacobian values; may be null
    double *hess;          // Hessian values
    const double *G0;      // n*n col-major (x batch if per-member)
    const int *upos;       // union pattern of the drives: flat col-major position
    const double *ucoef;   // n_upos x m coefficients (row-major: [q*m + l])
    const int *csr_ptr;    // m*(n+1): CSR row pointers of every G_l (rows of G_l)
    const int *csr_col;
    const double *csr_val;
    const int *csc_ptr;    // m*(n+1): CSC (= CSR of G_l^T) for the Hessian kernel
    const int *csc_row;
    const double *csc_val;
    const int *x_offs;     // per-member state offsets
    int x_off0;            // >= 0: every unit of the launch has this state offset (no dependent load of x_offs); -1: per member
    const int *umap;       // n*n: index into the union-pattern coefficient table, or -1
    const double *ell_val; // ELL form of the drives: [m][n][ell_w] (row-major), zero padded
    const int *ell_col;
    int ell_w, ell_lds;    // ELL width; 1 = stage the ELL arrays in LDS
    const unsigned char *uell_l;  // per union entry: up to uell_w (drive index, value) pairs, zero padded
    const double *uell_v;
    int uell_w;
    const double *ellt_val;  // ELL form of the transposed drives G_l^T: [m][n][ellt_w]
    const int *ellt_col;
    int ellt_w;
    const double *ug0;    // [n_upos] drift value at each union-pattern entry (first / shared drift)
    double *hpart;        // Hessian v2: per (b,k,slice) partial scalar entries
    unsigned int *hcnt;   // Hessian v2: per (b,k) arrival counter (self-resetting)
    int prof;        // builds with -DPCL_PROFILE only: experiment flags (option profile_flags); 0 otherwise
    long long *dbg;  // cycle stamps of workgroup 0 / matrix wave 0 (builds with -DPCL_PROFILE only; NULL otherwise)
    int ncw;      // v3: state columns per matrix-wave chunk ((2+m)*ncw <= 16)
    int tab_lds;  // v3: union / ELL tables staged in LDS
    int contig;   // v3: 1 = contiguous column ranges per workgroup (see the kernel), 0 = items dealt round-robin
    int all_matrix; // v3, contig: 1 = every workgroup takes the matrix role (compact Jacobian)
    int n_stream; // v3, contig: > 0 = role split, this many stream-role workgroups (the rest do the column work)
    int iso;               // 1: G0 and every G_l are exact iso(.) images -> G^2 needs only its first d columns
    long long z_batch_stride;   // doubles between trajectories (0 in MEMBERS mode)
    long long g0_batch_stride;  // n*n if per-member drift else 0
    long long jac_per;          // doubles per (b,k) in `jac`
    long long hess_per;
    int n_upos;
    int d, n, m, K, z_dim, u_off, dt_off, batch;
    int cols;     // state columns: d for a unitary (X is n x d), 1 for a ket; x_dim = n * cols
    int nc;       // state columns per slice
    int S;        // slices per interval
    int LD;       // LDS leading dimension of every n-row tile
    int compact;  // 1: write unique blocks only (jac_per is the compact size)
    int nt;       // streaming stores: 0 plain | 1 nontemporal | 2 write-through (sc0 sc1)
    double *expm;  // rollout: per (b,k) propagator exp(dt_k G(u_k)), n*n col-major
    double *xout;  // rollout: states at every knot, [batch][N][x_dim]
    int q;        // general-order kernel: p/2
    int lds_doubles;  // lock-step general-order kernel: doubles of LDS to zero-fill at the start
    double pc[6]; // general-order kernel: diagonal Pade coefficients c_0..c_q
    double *mpart;       // v3, optional: per state column (b, k, c) the m + 2 dot products of the reduce payload (pcl_eval_jac_merit_dev)
    const double *mlam;  // ... against these multipliers (NULL: lam = delta, the constraint merit)
    int tail_mode;       // pattern-compiled fused kernel: who stores delta and the tails (option v4_tail_mode)
    int v4_np;           // ... tiles of the ring of powers of G this launch rotates through (<= the SP4NP the module allocates)
    int v4_flags;        // ... A/B switches (option v4_flags): 1 no raised priority for the P wave | 2 tails only behind the item's last block | 4 no cooperative first item | 8 tiles NaN at start | 16 chains do not wait | 32 no balanced middle column
    unsigned int *tick;  // ... slice tickets (launches of several trajectories): [1] pipelines that have left, [2] the next chain ticket, [4 + i] the next
                         //     slice of interval i; all zero between launches (the last pipeline out re-zeroes them)
    int tick_cpi;        // ... state columns per slice
    int tick_G;          // ... workgroups per group
    int tick_ahead;      // ... slices the dispatcher may take ahead of the one the stream waves are storing (0 | 1)
    int *err;            // device error word of the context (host-mapped): bit 0 = a bounded wait of a barrier-free kernel gave up
};

// ------------------------------------------------------------------------------------------
// MFMA tile GEMM on LDS operands:  C[0:M,0:Nc] = op(A)[0:M,0:Kd] * B[0:Kd,0:Nc]
//   column-major everywhere; op(A) = A or A^T.  v_mfma_f64_16x16x4_f64 operand maps:
//   a: lane l holds A[i = l&15][k = l>>4], b: B[k = l>>4][j = l&15],
//   c/d: 4 doubles per lane, col = l&15, row = (l>>4) + 4*reg.
//   Out-of-range rows/cols/k are fed as exact zeros, so no tile padding is needed in LDS.
//   Two output tiles are processed together so each wave has two independent accumulators.
// ------------------------------------------------------------------------------------------
template <bool TRANS_A>
__device__ __forceinline__ void mfma_gemm_lds(const double *__restrict__ A, int lda, const double *__restrict__ B,
                                              int ldb, double *__restrict__ C, int ldc, int M, int Nc, int Kd,
                                              int wave, int nwaves, int lane) {
    const int rt_n = (M + 15) >> 4, ct_n = (Nc + 15) >> 4, ks_n = (Kd + 3) >> 2;
    const int nt = rt_n * ct_n;
    const int li = lane & 15, lk = lane >> 4;
    for (int t0 = wave * 2; t0 < nt; t0 += nwaves * 2) {
        const int t1 = t0 + 1;
        const bool has1 = t1 < nt;
        const int rt0 = t0 % rt_n, ct0 = t0 / rt_n;
        const int rt1 = has1 ? t1 % rt_n : rt0, ct1 = has1 ? t1 / rt_n : ct0;
        const int row0 = rt0 * 16 + li, col0 = ct0 * 16 + li;
        const int row1 = rt1 * 16 + li, col1 = ct1 * 16 + li;
        const bool r0 = row0 < M, c0 = col0 < Nc, r1 = has1 && row1 < M, c1 = has1 && col1 < Nc;
        double4_t acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
        for (int ks = 0; ks < ks_n; ++ks) {
            const int k = ks * 4 + lk;
            const bool kok = k < Kd;
            double a0 = 0.0, b0 = 0.0, a1 = 0.0, b1 = 0.0;
            if (r0 && kok) a0 = TRANS_A ? A[k + lda * row0] : A[row0 + lda * k];
            if (c0 && kok) b0 = B[k + ldb * col0];
            if (r1 && kok) a1 = TRANS_A ? A[k + lda * row1] : A[row1 + lda * k];
            if (c1 && kok) b1 = B[k + ldb * col1];
            acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc1, 0, 0, 0);
        }
        if (c0) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int rr = rt0 * 16 + lk + 4 * r;
                if (rr < M) C[rr + ldc * col0] = acc0[r];
            }
        }
        if (c1) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int rr = rt1 * 16 + lk + 4 * r;
                if (rr < M) C[rr + ldc * col1] = acc1[r];
            }
        }
    }
}

// Plain VALU version of the same contract (selected with option use_mfma = 0; used to A/B the
// matrix-core path and as a second implementation in the parity tests).
template <bool TRANS_A>
__device__ __forceinline__ void valu_gemm_lds(const double *__restrict__ A, int lda, const double *__restrict__ B,
                                              int ldb, double *__restrict__ C, int ldc, int M, int Nc, int Kd,
                                              int tid, int nthreads) {
    for (int e = tid; e < M * Nc; e += nthreads) {
        const int i = e % M, j = e / M;
        double s = 0.0;
        for (int k = 0; k < Kd; ++k) s = fma(TRANS_A ? A[k + lda * i] : A[i + lda * k], B[k + ldb * j], s);
        C[i + ldc * j] = s;
    }
}

template <bool MFMA, bool TRANS_A>
__device__ __forceinline__ void gemm_lds(const double *A, int lda, const double *B, int ldb, double *C, int ldc, int M,
                                         int Nc, int Kd) {
    if (MFMA)
        mfma_gemm_lds<TRANS_A>(A, lda, B, ldb, C, ldc, M, Nc, Kd, threadIdx.x >> 6, blockDim.x >> 6,
                               threadIdx.x & 63);
    else
        valu_gemm_lds<TRANS_A>(A, lda, B, ldb, C, ldc, M, Nc, Kd, threadIdx.x, blockDim.x);
}

// nt: 0 plain | 1 nontemporal | 2 write-through at system scope (sc0 sc1).  Write-through halves the rate of launches whose
// output exceeds the 256 MB infinity cache (1.08 GB: 191 -> 379 us) but is 0.7 us faster for one trajectory (135 MB, 27.3 ->
// 26.6 us): the host picks it by launch size.
// (The write-through store is a hand-written instruction the compiler does not see as a store: it must never be followed closely by a
//  write of its data registers -- pcl_hess_cols_kernel lost a few lines per launch that way with plain hand-written 16-byte stores.  The
//  callers pass long-lived accumulators: the nearest overwrite in the generated code is 27 instructions behind, checked in the ISA.)
__device__ __forceinline__ void store2(double *p, double a, double b, int nt) {
    double2_t v = {a, b};
    if (nt == 2)
        asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
    else if (nt)
        __builtin_nontemporal_store(v, reinterpret_cast<double2_t *>(p));
    else
        *reinterpret_cast<double2_t *>(p) = v;
}

// One entry of -B^+ = -(I + (h/2) G + (h^2/12) G^2) and of B^- = I - (h/2) G + (h^2/12) G^2, with the roundings pinned (explicit
// fused multiply-adds): the four kernels that form these blocks are compared bit for bit, and what the compiler contracts on its
// own changes with the surrounding code.  id = the entry of I, c1 = h/2, c2 = h^2/12, g / h2 = the entries of G and G^2.
__device__ __forceinline__ void bpm_entry(double id, double c1, double c2, double g, double h2, double &mbp, double &bm) {
    const double e = __builtin_fma(c2, h2, id);
    mbp = -__builtin_fma(c1, g, e);
    bm = __builtin_fma(-c1, g, e);
}

// Column-major n x n matrix in memory -> LDS tile with leading dimension LD, by NT threads.  All global loads of a block of
// eight iterations are issued before the first LDS write (written as load -> store per iteration, the compiler waits for every
// load in turn: one dependent round trip to L2 / HBM per iteration).
template <int NT, bool TRANSPOSE = false>  // TRANSPOSE: dst[col + LD*row] (the row-contiguous layout of a conflict-free MFMA a operand)
__device__ __forceinline__ void load_tile(const double *__restrict__ src, double *__restrict__ dst, int n, int LD, int tid) {
    const int nn = n * n;
    for (int e0 = tid; e0 < nn; e0 += NT * 8) {
        double v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = e0 + NT * j < nn ? src[e0 + NT * j] : 0.0;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int e = e0 + NT * j;
            if (e < nn) dst[TRANSPOSE ? (e / n) + LD * (e % n) : (e % n) + LD * (e / n)] = v[j];
        }
    }
}

// Assemble G(u_k) into LDS (ld = LD):  G = G0 + sum_l u_l G_l, in drive order (deterministic).
__device__ __forceinline__ void build_G(const KParams &p, const double *__restrict__ G0, const double *__restrict__ zk,
                                        double *__restrict__ G, double *__restrict__ us) {
    const int n = p.n, LD = p.LD;
    for (int e = threadIdx.x; e < n * n; e += blockDim.x) G[(e % n) + LD * (e / n)] = G0[e];
    if ((int)threadIdx.x < p.m) us[threadIdx.x] = zk[p.u_off + threadIdx.x];
    __syncthreads();
    for (int q = threadIdx.x; q < p.n_upos; q += blockDim.x) {
        const int pos = p.upos[q];
        const int idx = (pos % n) + LD * (pos / n);
        double g = G[idx];
        for (int l = 0; l < p.m; ++l) g += us[l] * p.ucoef[(long long)q * p.m + l];
        G[idx] = g;
    }
}

// ------------------------------------------------------------------------------------------
// Wave-level helpers
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void wave_lds_sync() {
    // Lanes of one wave exchange data through LDS: wait for this wave's LDS traffic only (never vmcnt - the
    // wave's global stores may sit in a saturated store queue for microseconds) and pin the compiler's order.
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
}

#define PCL_MAXRT 4 // 16-row tiles of an n <= 64 operand
#define PCL_MREG 8  // drives whose ELL row is held in registers (EW > 0 variants)
#define PCL_PFC 4   // chunks per matrix wave whose state inputs are fetched at the top of the item (registers)
#define PCL_PFW 2   // ... for chunk widths up to this many columns (wider chunks load at use)

__device__ __forceinline__ double wave_sum(double v) {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
// sum over the 16 lanes of a DPP row, the same bits in every lane of the row (xor 1, xor 2, half mirror, mirror)
__device__ __forceinline__ double row16_sum(double v) {
    v += dpp_f64<0xB1>(v);
    v += dpp_f64<0x4E>(v);
    v += dpp_f64<0x141>(v);
    v += dpp_f64<0x140>(v);
    return v;
}
