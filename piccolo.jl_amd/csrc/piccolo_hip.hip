// piccolo_hip.hip -- gfx950 (MI355X / CDNA4) kernels + C ABI for the Pade-4 collocation
// constraint evaluator.  ABI and the reference interfaces it replaces: include/piccolo_hip.h.
//
// Kernel design (DESIGN.md has the full account):
//   one workgroup (4 wavefronts) per (member b, interval k, column slice s).
//   * G(u_k) = G0 + sum_l u_l G_l is assembled in LDS from the dense drift tile and the
//     union sparsity pattern of the drives;
//   * G^2 and G * [S | D | G_l D] run on the f64 matrix cores (v_mfma_f64_16x16x4_f64),
//     operands read straight from LDS tiles, one 16x16 output tile per wavefront at a time;
//   * the slice's columns of delta, d/du_l, d/ddt are combined on the VALU and stored;
//   * the slice's share of the d replicated diagonal blocks of I_d (x) B^{+-} is formed in
//     registers from the LDS-resident G and G^2 and streamed to HBM with 16-byte coalesced
//     stores (this stream is >98 % of the bytes: the kernel is HBM-write-bound).
//
// No CPU fallback exists in this file: every entry point needs a HIP device.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include <dlfcn.h>

#include "piccolo_hip.h"

#define PCL_VERSION_STR "piccolo_hip 0.2.0 (gfx950, pade 2/4/6/8/10)"

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
    long long *dbg;  // optional: cycle stamps of workgroup 0 / matrix wave 0 (option debug_timing)
    int ncw;      // v3: state columns per matrix-wave chunk ((2+m)*ncw <= 16)
    int tab_lds;  // v3: union / ELL tables staged in LDS
    int contig;   // v3: 1 = contiguous column ranges per workgroup (see the kernel), 0 = items dealt round-robin
    int snc;      // v3, role split: > 0 = the stream workgroups take pieces of snc columns round-robin (0: contiguous ranges)
    int flat;     // v3: 1 = line-aligned flat block stream (values recomputed from LDS per store), 0 = per-block stores from registers
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
    int compact;  // 1: write unique blocks only (jac_per is the compact size); 2: split mode - unique blocks go to
                  //    `blocks` (2*n*n per (b,k)) and `flags[b*K+k]` is raised, everything else in the full layout
    double *blocks;
    unsigned int *flags;
    int nt;       // 1: nontemporal streaming stores
    double *expm;  // rollout: per (b,k) propagator exp(dt_k G(u_k)), n*n col-major
    double *xout;  // rollout: states at every knot, [batch][N][x_dim]
    int q;        // general-order kernel: p/2
    double pc[6]; // general-order kernel: diagonal Pade coefficients c_0..c_q
    int ablate;   // DEBUG ONLY (wrong results): bit0 skip matrix products, bit1 skip block streaming, bit2 skip column outputs
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

__device__ __forceinline__ void store2(double *p, double a, double b, bool nt) {
    double2_t v = {a, b};
    if (nt)
        __builtin_nontemporal_store(v, reinterpret_cast<double2_t *>(p));
    else
        *reinterpret_cast<double2_t *>(p) = v;
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
// Fused residual + Jacobian kernel.
// LDS map (doubles):  G [LD*n] | G2 [LD*n] | M1 [LD*(2+m)*nc] | W1 [LD*(2+m)*nc] | G2D [LD*nc] | T [LD*nc] | us[8+m]
//   M1 = [S | D | G_1 D .. G_m D] (slice columns), W1 = G*M1 = [GS | GD | G(G_l D)].
// ------------------------------------------------------------------------------------------
template <bool JAC, bool MFMA>
__global__ __launch_bounds__(256) void pcl_fused_kernel(const KParams p) {
    extern __shared__ double lds[];
    const int n = p.n, d = p.cols, m = p.m, LD = p.LD, nc = p.nc;  // d: state columns here (no iso shortcut in this kernel)
    const int tid = threadIdx.x, nth = blockDim.x;

    const int bid = blockIdx.x;
    const int s = bid % p.S;
    const int k = (bid / p.S) % p.K;
    const int b = bid / (p.S * p.K);
    const int c0 = s * nc;
    const int nce = min(nc, d - c0);  // columns actually owned by this slice

    const int ncols1 = JAC ? (2 + m) * nc : 2 * nc;
    double *G = lds;
    double *G2 = G + LD * n;
    double *M1 = G2 + (JAC ? LD * n : 0);
    double *W1 = M1 + LD * ncols1;
    double *G2D = W1 + LD * ncols1;
    double *T = G2D + LD * nc;
    double *us = T + LD * nc;

    const double *Zb = p.Z + (long long)b * p.z_batch_stride;
    const double *zk = Zb + (long long)k * p.z_dim;
    const double *zn = zk + p.z_dim;
    const int x_off = p.x_offs[p.z_batch_stride ? 0 : b];
    const double h = zk[p.dt_off];
    const double c1 = 0.5 * h, c2 = h * h * (1.0 / 12.0);
    const long long xd = (long long)n * d;

    build_G(p, p.G0 + (long long)b * p.g0_batch_stride, zk, G, us);

    // S and D for the slice's columns (unused trailing columns are zero)
    for (int e = tid; e < nc * n; e += nth) {
        const int c = e / n, i = e % n;
        double xs = 0.0, xdv = 0.0;
        if (c < nce) {
            const double xn = zn[x_off + (c0 + c) * n + i], xc = zk[x_off + (c0 + c) * n + i];
            xs = xn + xc;
            xdv = xn - xc;
        }
        M1[i + LD * c] = xs;
        M1[i + LD * (nc + c)] = xdv;
    }
    __syncthreads();

    if (JAC) {
        // G_l D via the CSR rows of G_l
        const double *Dm = M1 + LD * nc;
        for (int e = tid; e < m * nc * n; e += nth) {
            const int i = e % n, c = (e / n) % nc, l = e / (n * nc);
            const int *rp = p.csr_ptr + l * (n + 1);
            double acc = 0.0;
            for (int q = rp[i]; q < rp[i + 1]; ++q) acc += p.csr_val[q] * Dm[p.csr_col[q] + LD * c];
            M1[i + LD * ((2 + l) * nc + c)] = acc;
        }
        __syncthreads();
        if (!(p.ablate & 1)) gemm_lds<MFMA, false>(G, LD, G, LD, G2, LD, n, n, n);
    }
    if (!(p.ablate & 1)) gemm_lds<MFMA, false>(G, LD, M1, LD, W1, LD, n, ncols1, n);
    __syncthreads();

    // pass 2: G2D = G * (G D);  T = -c1 S + c2 G D
    if (!(p.ablate & 1)) gemm_lds<MFMA, false>(G, LD, W1 + LD * nc, LD, G2D, LD, n, nc, n);
    if (JAC) {
        for (int e = tid; e < nc * n; e += nth) {
            const int c = e / n, i = e % n;
            T[i + LD * c] = -c1 * M1[i + LD * c] + c2 * W1[i + LD * (nc + c)];
        }
    }
    __syncthreads();

    // ---- column outputs -------------------------------------------------------------------
    const long long bk = (long long)b * p.K + k;
    double *jb = JAC ? p.jac + bk * p.jac_per : nullptr;
    const long long blk = p.compact ? (long long)n * n : (long long)d * n * n;  // size of seg 0 / seg 1
    for (int e = tid; e < ((p.ablate & 4) ? 0 : nce * n); e += nth) {
        const int c = e / n, i = e % n;
        const double gs = W1[i + LD * c], g2d = G2D[i + LD * c];
        const long long r = (long long)(c0 + c) * n + i;
        if (p.delta) p.delta[bk * xd + r] = M1[i + LD * (nc + c)] - c1 * gs + c2 * g2d;
        if (JAC) jb[2 * blk + ((long long)(c0 + c) * (m + 1) + m) * n + i] = -0.5 * gs + (h * (1.0 / 6.0)) * g2d;
    }
    if (JAC) {
        for (int e = tid; e < ((p.ablate & 4) ? 0 : m * nce * n); e += nth) {
            const int i = e % n, c = (e / n) % nce, l = e / (n * nce);
            const int *rp = p.csr_ptr + l * (n + 1);
            double acc = 0.0;
            for (int q = rp[i]; q < rp[i + 1]; ++q) acc += p.csr_val[q] * T[p.csr_col[q] + LD * c];
            jb[2 * blk + ((long long)(c0 + c) * (m + 1) + l) * n + i] = acc + c2 * W1[i + LD * ((2 + l) * nc + c)];
        }

        // ---- replicated diagonal blocks: stream -B^+ and B^- ---------------------------------
        // pair index q covers flat column-major positions 2q, 2q+1 (same column since n is even)
        const int half = (n * n) >> 1;
        int cbeg = c0, cend = c0 + nce;
        if (p.compact) {  // unique blocks only: slice 0 writes the single copy
            cbeg = 0;
            cend = (s == 0) ? 1 : 0;
        }
        for (int q = tid; q < ((p.ablate & 2) ? 0 : half); q += nth) {
            const int pos = 2 * q;
            const int i = pos % n, j = pos / n;
            const double g0 = G[i + LD * j], g1 = G[i + 1 + LD * j];
            const double h0 = G2[i + LD * j], h1 = G2[i + 1 + LD * j];
            const double id0 = (i == j) ? 1.0 : 0.0, id1 = (i + 1 == j) ? 1.0 : 0.0;
            const double e0 = id0 + c2 * h0, e1 = id1 + c2 * h1;
            const double bp0 = -(e0 + c1 * g0), bp1 = -(e1 + c1 * g1);
            const double bm0 = e0 - c1 * g0, bm1 = e1 - c1 * g1;
            for (int c = cbeg; c < cend; ++c) {
                double *o0 = jb + (long long)c * n * n + pos;
                store2(o0, bp0, bp1, p.nt);
                store2(o0 + blk, bm0, bm1, p.nt);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// General-order kernel: diagonal Pade orders p = 2q, q <= 5 (p = 2, 6, 8, 10; p = 4 only as a cross-check of the
// specialised kernels).  One workgroup per (b, k, slice of nc columns); correctness first, no wave specialisation.
// With Y_j = (-1)^j X_{k+1} - X_k (D for even j, -S for odd j) and c_j the Pade coefficients:
//   residual (Horner)   W_q = c_q Y_q,  W_j = c_j Y_j + h G W_{j+1},  delta = W_0
//   d/dh                V_q = q c_q Y_q, V_j = j c_j Y_j + h G V_{j+1} (j >= 1),  d delta/dh = G V_1
//   d/du_l              dW_q = 0,  dW_j = h (G_l W_{j+1} + G dW_{j+1}),  d delta/du_l = dW_0
//   blocks              B^{+-} = sum_j c_j (+-h)^j G^j, powers by repeated products, sums in registers
// LDS map (doubles): G | Pa | Pb (JAC) | -S | D | W_0..W_q | V (2, JAC) | dWa, dWb (m each, JAC) | us      (column blocks LD*nc)
// ------------------------------------------------------------------------------------------
template <bool JAC>
__global__ __launch_bounds__(256) void pcl_pade_kernel(const KParams p) {
    extern __shared__ double lds[];
    const int n = p.n, d = p.cols, m = p.m, LD = p.LD, nc = p.nc, q = p.q;
    const int tid = threadIdx.x, nth = blockDim.x;
    const int bid = blockIdx.x;
    const int s = bid % p.S;
    const int k = (bid / p.S) % p.K;
    const int b = bid / (p.S * p.K);
    const int c0 = s * nc;
    const int nce = min(nc, d - c0);
    const int LDc = LD * nc;

    double *G = lds;
    double *Pa = G + LD * n;
    double *Pb = Pa + (JAC ? LD * n : 0);
    double *Sm = Pb + (JAC ? LD * n : 0);  // -S
    double *Dm = Sm + LDc;
    double *W = Dm + LDc;  // W_j at W + j*LDc
    double *V = W + (q + 1) * LDc;
    double *dWa = V + (JAC ? 2 * LDc : 0);
    double *dWb = dWa + (JAC ? m * LDc : 0);
    double *us = dWb + (JAC ? m * LDc : 0);

    const double *Zb = p.Z + (long long)b * p.z_batch_stride;
    const double *zk = Zb + (long long)k * p.z_dim;
    const double *zn = zk + p.z_dim;
    const int x_off = p.x_offs[p.z_batch_stride ? 0 : b];
    const double h = zk[p.dt_off];
    const long long xd = (long long)n * d;

    build_G(p, p.G0 + (long long)b * p.g0_batch_stride, zk, G, us);
    for (int e = tid; e < nc * n; e += nth) {
        const int c = e / n, i = e % n;
        double xs = 0.0, xdv = 0.0;
        if (c < nce) {
            const double xn = zn[x_off + (c0 + c) * n + i], xc = zk[x_off + (c0 + c) * n + i];
            xs = xn + xc;
            xdv = xn - xc;
        }
        Sm[i + LD * c] = -xs;
        Dm[i + LD * c] = xdv;
    }
    __syncthreads();
    auto Y = [&](int j) { return (j & 1) ? Sm : Dm; };

    // ---- residual: Horner in G ---------------------------------------------------------------------------------
    for (int e = tid; e < nc * n; e += nth) {
        const int idx = (e % n) + LD * (e / n);
        W[q * LDc + idx] = p.pc[q] * Y(q)[idx];
    }
    __syncthreads();
    for (int j = q - 1; j >= 0; --j) {
        gemm_lds<true, false>(G, LD, W + (j + 1) * LDc, LD, W + j * LDc, LD, n, nc, n);
        __syncthreads();
        for (int e = tid; e < nc * n; e += nth) {
            const int idx = (e % n) + LD * (e / n);
            W[j * LDc + idx] = p.pc[j] * Y(j)[idx] + h * W[j * LDc + idx];
        }
        __syncthreads();
    }
    const long long bk = (long long)b * p.K + k;
    if (p.delta)
        for (int e = tid; e < nce * n; e += nth) p.delta[bk * xd + (long long)c0 * n + e] = W[(e % n) + LD * (e / n)];
    if (!JAC) return;

    double *jb = p.jac + bk * p.jac_per;
    const long long blk = p.compact ? (long long)n * n : (long long)d * n * n;
    double *jt = jb + 2 * blk;
    // ---- d/dh ------------------------------------------------------------------------------------------------
    {
        double *cur = V, *oth = V + LDc;
        for (int e = tid; e < nc * n; e += nth) {
            const int idx = (e % n) + LD * (e / n);
            cur[idx] = q * p.pc[q] * Y(q)[idx];
        }
        __syncthreads();
        for (int j = q - 1; j >= 1; --j) {
            gemm_lds<true, false>(G, LD, cur, LD, oth, LD, n, nc, n);
            __syncthreads();
            for (int e = tid; e < nc * n; e += nth) {
                const int idx = (e % n) + LD * (e / n);
                oth[idx] = j * p.pc[j] * Y(j)[idx] + h * oth[idx];
            }
            __syncthreads();
            double *t = cur;
            cur = oth;
            oth = t;
        }
        gemm_lds<true, false>(G, LD, cur, LD, oth, LD, n, nc, n);
        __syncthreads();
        for (int e = tid; e < nce * n; e += nth) {
            const int c = e / n, i = e % n;
            jt[((long long)(c0 + c) * (m + 1) + m) * n + i] = oth[i + LD * c];
        }
    }
    // ---- d/du_l ----------------------------------------------------------------------------------------------
    if (m > 0) {
        double *cur = dWa, *oth = dWb;
        for (int e = tid; e < m * nc * n; e += nth) {
            const int i = e % n, c = (e / n) % nc, l = e / (n * nc);
            const int *rp = p.csr_ptr + l * (n + 1);
            double a = 0.0;
            for (int t = rp[i]; t < rp[i + 1]; ++t) a += p.csr_val[t] * W[q * LDc + p.csr_col[t] + LD * c];
            cur[l * LDc + i + LD * c] = h * a;
        }
        __syncthreads();
        for (int j = q - 2; j >= 0; --j) {
            gemm_lds<true, false>(G, LD, cur, LD, oth, LD, n, m * nc, n);
            __syncthreads();
            for (int e = tid; e < m * nc * n; e += nth) {
                const int i = e % n, c = (e / n) % nc, l = e / (n * nc);
                const int *rp = p.csr_ptr + l * (n + 1);
                double a = 0.0;
                for (int t = rp[i]; t < rp[i + 1]; ++t) a += p.csr_val[t] * W[(j + 1) * LDc + p.csr_col[t] + LD * c];
                oth[l * LDc + i + LD * c] = h * (oth[l * LDc + i + LD * c] + a);
            }
            __syncthreads();
            double *t = cur;
            cur = oth;
            oth = t;
        }
        for (int e = tid; e < m * nce * n; e += nth) {
            const int i = e % n, c = (e / n) % nce, l = e / (n * nce);
            jt[((long long)(c0 + c) * (m + 1) + l) * n + i] = cur[l * LDc + i + LD * c];
        }
    }
    // ---- blocks: B^{+-} = sum_j c_j (+-h)^j G^j ------------------------------------------------------------------
    {
        const int half = (n * n) >> 1;
        double bp[PCL_NSP][2], bm[PCL_NSP][2];  // pairs (2q', 2q'+1), q' = tid + 256 r
#pragma unroll
        for (int r = 0; r < PCL_NSP; ++r) {
            const int pos = 2 * (tid + 256 * r);
            const int i = pos % n, jj = pos / n;
            bp[r][0] = bm[r][0] = (i == jj) ? 1.0 : 0.0;
            bp[r][1] = bm[r][1] = (i + 1 == jj) ? 1.0 : 0.0;
        }
        const double *Pc = G;
        double hp = 1.0, hm = 1.0;
        for (int j = 1; j <= q; ++j) {
            hp *= h;
            hm *= -h;
#pragma unroll
            for (int r = 0; r < PCL_NSP; ++r) {
                const int qq = tid + 256 * r;
                if (qq < half) {
                    const int pos = 2 * qq;
                    const int i = pos % n, jj = pos / n;
                    const double v0 = Pc[i + LD * jj], v1 = Pc[i + 1 + LD * jj];
                    bp[r][0] += p.pc[j] * hp * v0;
                    bp[r][1] += p.pc[j] * hp * v1;
                    bm[r][0] += p.pc[j] * hm * v0;
                    bm[r][1] += p.pc[j] * hm * v1;
                }
            }
            if (j < q) {
                double *Pn = (Pc == Pa) ? Pb : Pa;
                gemm_lds<true, false>(G, LD, Pc, LD, Pn, LD, n, n, n);
                __syncthreads();
                Pc = Pn;
            }
        }
        int cbeg = c0, cend = c0 + nce;
        if (p.compact) {
            cbeg = 0;
            cend = (s == 0) ? 1 : 0;
        }
#pragma unroll
        for (int r = 0; r < PCL_NSP; ++r) {
            const int qq = tid + 256 * r;
            if (qq < half)
                for (int c = cbeg; c < cend; ++c) {
                    double *o0 = jb + (long long)c * n * n + 2 * qq;
                    store2(o0, -bp[r][0], -bp[r][1], p.nt);
                    store2(o0 + blk, bm[r][0], bm[r][1], p.nt);
                }
        }
    }
}

// ------------------------------------------------------------------------------------------
// Fused residual + Jacobian kernel, version 2 (default): 8 wavefronts, wave-specialised.
//   waves 0-3 ("matrix" waves) own one 16-row tile each and run every MFMA product;
//   waves 4-7 ("stream" waves) apply the sparse drives (VALU) and then stream the slice's share of
//   the replicated B^{+-} blocks to HBM while the matrix waves compute the slice's columns.
// Phases (separated by one __syncthreads each):
//   0  all   : G(u_k) -> LDS (one pass: drift tile + union-pattern map), S, D, ELL drives -> LDS
//   1  matrix: G2 = G*G (first d columns + mirror if iso)      stream: M1[:, (2+l)nc..] = G_l D
//   2  matrix: W1 = G*M1, G2D = G2*D                            stream: -B^+, B^- block copies -> HBM
//   3  all   : delta, d/ddt, d/du_l columns -> HBM
// LDS map (doubles): G [LD*n] | G2 [LD*n] | M1 [LD*ncols1] | W1 [LD*ncols1] | G2D [LD*nc] | ELL val/col | slack
// MFMA operand loads are unconditional: a tile may read rows/columns past the matrix edge (the
// neighbouring buffer); such lanes only feed output rows/columns that are never stored.
// ------------------------------------------------------------------------------------------
template <int MODE>  // 0: plain store; 1: store + iso mirror (C = G2 first d columns)
__device__ __forceinline__ void wave_rowgemm(const double *__restrict__ A, int lda, const double *__restrict__ B,
                                             int ldb, double *__restrict__ C, int ldc, int M, int Nc, int Kd, int wave,
                                             int nwaves, int lane, int dmir) {
    const int rt_n = (M + 15) >> 4, ct_n = (Nc + 15) >> 4;
    const int kfull = Kd >> 2, krem = Kd & 3;
    const int li = lane & 15, lk = lane >> 4;
    for (int rt = wave; rt < rt_n; rt += nwaves) {
        const double *Ap = A + rt * 16 + li + lda * lk;
        for (int ct = 0; ct < ct_n; ct += 2) {
            const bool two = ct + 1 < ct_n;
            const double *Bp0 = B + lk + ldb * (ct * 16 + li);
            const double *Bp1 = Bp0 + (two ? ldb * 16 : 0);
            double4_t acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
            // operands of step ks+1 are requested before the MFMAs of step ks issue
            double an = 0.0, b0n = 0.0, b1n = 0.0;
            if (kfull > 0) {
                an = Ap[0];
                b0n = Bp0[0];
                if (two) b1n = Bp1[0];
            }
            for (int ks = 0; ks < kfull; ++ks) {
                const double a = an, b0 = b0n, b1 = b1n;
                if (ks + 1 < kfull) {
                    an = Ap[lda * 4 * (ks + 1)];
                    b0n = Bp0[4 * (ks + 1)];
                    if (two) b1n = Bp1[4 * (ks + 1)];
                }
                acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b0, acc0, 0, 0, 0);
                if (two) acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b1, acc1, 0, 0, 0);
            }
            if (krem) {
                const bool ok = lk < krem;
                const double a = ok ? Ap[lda * 4 * kfull] : 0.0;
                const double b0 = ok ? Bp0[4 * kfull] : 0.0;
                const double b1 = ok ? Bp1[4 * kfull] : 0.0;
                acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b0, acc0, 0, 0, 0);
                if (two) acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b1, acc1, 0, 0, 0);
            }
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                if (t == 1 && !two) break;
                const int col = (ct + t) * 16 + li;
                if (col < Nc) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int row = rt * 16 + lk + 4 * r;
                        const double v = t ? acc1[r] : acc0[r];
                        if (row < M) {
                            C[row + ldc * col] = v;
                            if (MODE == 1) {
                                if (row < dmir)
                                    C[row + dmir + ldc * (col + dmir)] = v;
                                else
                                    C[row - dmir + ldc * (col + dmir)] = -v;
                            }
                        }
                    }
                }
            }
        }
    }
}

// Phase-2 product of the matrix waves for a narrow slice (ncols1 <= 32, 2*nc <= 16), one row tile per wave:
//   acc0/acc1 = G * M1[:, tile 0/1],  acc2 = G2 * M1[:, tile 0]  (its columns nc..2nc-1 are G2 D).
// Operands of k-step ks+1 are requested before the MFMAs of step ks issue.
__device__ __forceinline__ void wave_phase2_fused(const double *G, const double *G2, const double *M1, double *W1,
                                                  double *G2D, int LD, int n, int ncols1, int nc, bool want_g2, int wave,
                                                  int lane) {
    const int rt_n = (n + 15) >> 4;
    const bool two = ncols1 > 16;
    const int kfull = n >> 2, krem = n & 3;
    const int li = lane & 15, lk = lane >> 4;
    for (int rt = wave; rt < rt_n; rt += 4) {
        const double *Ap = G + rt * 16 + li + LD * lk;
        const double *A2p = (want_g2 ? G2 : G) + rt * 16 + li + LD * lk;
        const double *Bp0 = M1 + lk + LD * li;
        const double *Bp1 = Bp0 + (two ? LD * 16 : 0);
        double4_t acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0}, acc2 = {0.0, 0.0, 0.0, 0.0};
        double a = 0.0, a2 = 0.0, b0 = 0.0, b1 = 0.0;
        if (kfull > 0) {
            a = Ap[0];
            a2 = A2p[0];
            b0 = Bp0[0];
            b1 = Bp1[0];
        }
        for (int ks = 0; ks < kfull; ++ks) {
            const double ca = a, ca2 = a2, cb0 = b0, cb1 = b1;
            if (ks + 1 < kfull) {
                a = Ap[LD * 4 * (ks + 1)];
                a2 = A2p[LD * 4 * (ks + 1)];
                b0 = Bp0[4 * (ks + 1)];
                b1 = Bp1[4 * (ks + 1)];
            }
            acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(ca, cb0, acc0, 0, 0, 0);
            if (two) acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(ca, cb1, acc1, 0, 0, 0);
            if (want_g2) acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(ca2, cb0, acc2, 0, 0, 0);
        }
        if (krem) {
            const bool ok = lk < krem;
            const double ra = ok ? Ap[LD * 4 * kfull] : 0.0;
            const double ra2 = ok ? A2p[LD * 4 * kfull] : 0.0;
            const double rb0 = ok ? Bp0[4 * kfull] : 0.0;
            const double rb1 = ok ? Bp1[4 * kfull] : 0.0;
            acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(ra, rb0, acc0, 0, 0, 0);
            if (two) acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(ra, rb1, acc1, 0, 0, 0);
            if (want_g2) acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(ra2, rb0, acc2, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = rt * 16 + lk + 4 * r;
            if (row < n) {
                if (li < ncols1) W1[row + LD * li] = acc0[r];
                if (two && li + 16 < ncols1) W1[row + LD * (li + 16)] = acc1[r];
                if (want_g2 && li >= nc && li < 2 * nc) G2D[row + LD * (li - nc)] = acc2[r];
            }
        }
    }
}

// Persistent form: the grid is (workgroups that fit on the chip); each workgroup walks the work
// items (b, k, s) with stride gridDim.x.  What does not depend on the item stays on chip for the
// whole launch: the LDS tile G holds the drift everywhere except on the union pattern of the
// drives, which is the only part rewritten per item (each thread keeps its pattern entries in
// registers), and the ELL form of the drives is staged in LDS once.  Per item only u_k, dt_k and
// the slice's state columns are read from memory, one item ahead.  Element-wise passes give every
// thread a fixed row (tid % n) and walk columns: no integer division inside the item loop.
#define PCL_NUE2 2  // union-pattern entries per thread held in registers (REG path: n_upos <= 1024)

// TD/TM/TNC: compile-time Hilbert dimension, drive count and slice width (0 = run-time values).  With the shape fixed
// every LDS offset, trip count and divisor is a constant: the specialised instances need far fewer scalar registers.
template <bool JAC, int WU, int TD, int TM, int TNC>  // WU: (drive,value) pairs per pattern entry in registers; -1: general
__global__ __launch_bounds__(512, 4) void pcl_fused_kernel_v2(const KParams p) {
    extern __shared__ double lds[];
    const int d = TD ? TD : p.d, n = 2 * d, m = TD ? TM : p.m, LD = TD ? ((2 * TD + 3) & ~3) + 2 : p.LD, nc = TNC ? TNC : p.nc;
    const int C = TD ? TD : p.cols;  // state columns (specialised instances are unitary: C = d)
    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const bool matrix_wave = wave < 4;
    const int stid = tid - 256;  // index among the stream waves' threads
    const int nn = n * n;

    const int ncols1 = JAC ? (2 + m) * nc : 2 * nc;
    const int n_ell = m * n * p.ell_w;
    double *G = lds;
    double *G2 = G + LD * n;
    double *M1 = G2 + (JAC ? LD * n : 0);
    double *W1 = M1 + LD * ncols1;
    double *G2D = W1 + LD * ncols1;
    double *us = G2D + LD * nc;  // 2 x [u_k (m) | dt_k]: current / next item
    double *ellv_l = us + 2 * (m + 1);
    unsigned short *ellc_l = reinterpret_cast<unsigned short *>(ellv_l + n_ell);
    const long long xd = (long long)n * C;
    const int ew = p.ell_w;
    const bool fused_p2 = ncols1 <= 32 && 2 * nc <= 16;

    // ---- fixed thread coordinates ----------------------------------------------------------------------
    const int ri = tid % n, rj0 = tid / n, rstep = 512 / n;  // all threads: row ri, columns rj0, rj0+rstep, ..
    const bool ract = rj0 < rstep;
    const int sstep = 256 / n;                               // stream waves as a 256-thread group: row si
    const int si = matrix_wave ? 0 : stid % n, sj0 = matrix_wave ? 0 : stid / n;
    const bool sact = !matrix_wave && sj0 < sstep;
    const int hn = n >> 1;                                   // stream waves: row pair (pi, pi+1)
    const int pi = matrix_wave ? 0 : 2 * (stid % hn), pj0 = matrix_wave ? 0 : stid / hn, pstep = max(256 / hn, 1);
    const bool pact = !matrix_wave && pj0 < pstep;

    // ---- launch-invariant state -------------------------------------------------------------------------
    if (!p.g0_batch_stride)
        for (int e = tid; e < nn; e += 512) G[(e % n) + LD * (e / n)] = p.G0[e];
    constexpr int WUR = WU > 0 ? WU : 1;
    int un_idx[PCL_NUE2];
    double un_g0[PCL_NUE2];
    unsigned char un_l[PCL_NUE2][WUR];
    double un_v[PCL_NUE2][WUR];
    if (WU > 0) {
#pragma unroll
        for (int r = 0; r < PCL_NUE2; ++r) {
            const int q = tid + 512 * r;
            un_idx[r] = -1;
            un_g0[r] = 0.0;
#pragma unroll
            for (int w = 0; w < WUR; ++w) {
                un_l[r][w] = 0;
                un_v[r][w] = 0.0;
            }
            if (q < p.n_upos) {
                const int pos = p.upos[q];
                un_idx[r] = (pos % n) + LD * (pos / n);
                un_g0[r] = p.G0[pos];
#pragma unroll
                for (int w = 0; w < WUR; ++w) {
                    un_l[r][w] = p.uell_l[q * WUR + w];
                    un_v[r][w] = p.uell_v[q * WUR + w];
                }
            }
        }
    }
    const bool stage = JAC && p.ell_lds;
    if (stage) {
        for (int e = tid; e < n_ell; e += 512) {
            ellv_l[e] = p.ell_val[e];
            ellc_l[e] = (unsigned short)p.ell_col[e];
        }
    }

    // ---- per-item inputs, requested one item ahead -------------------------------------------------
    const int n_items = p.batch * p.K * p.S;
    const bool pf_x = nc <= rstep;  // one state element per thread: prefetchable
    double pf_v = 0.0, pf_xn = 0.0, pf_xc = 0.0;
    auto request = [&](int item) {
        const int s = item % p.S;
        const int k = (item / p.S) % p.K;
        const int b = item / (p.S * p.K);
        const double *zk = p.Z + (long long)b * p.z_batch_stride + (long long)k * p.z_dim;
        if (tid <= m) pf_v = zk[tid < m ? p.u_off + tid : p.dt_off];
        pf_xn = pf_xc = 0.0;
        if (pf_x && ract && rj0 < min(nc, C - s * nc)) {
            const int x_off = p.x_offs[p.z_batch_stride ? 0 : b];
            const long long o = x_off + (long long)(s * nc + rj0) * n + ri;
            pf_xc = zk[o];
            pf_xn = zk[p.z_dim + o];
        }
    };
    int cur = 0;
    if ((int)blockIdx.x < n_items) {
        request(blockIdx.x);
        if (tid <= m) us[tid] = pf_v;
    }
    __syncthreads();  // G = drift, tables staged, us[0] valid

    for (int item = blockIdx.x; item < n_items; item += gridDim.x) {
        const int s = item % p.S;
        const int k = (item / p.S) % p.K;
        const int b = item / (p.S * p.K);
        const int c0 = s * nc;
        const int nce = min(nc, C - c0);
        const double *zk = p.Z + (long long)b * p.z_batch_stride + (long long)k * p.z_dim;
        const double *zn = zk + p.z_dim;
        const int x_off = p.x_offs[p.z_batch_stride ? 0 : b];
        const double *usc = us + cur * (m + 1);
        const double h = usc[m];
        const double c1 = 0.5 * h, c2 = h * h * (1.0 / 12.0);

        // ---- phase 0: G(u_k) on the union pattern, S, D -> LDS -----------------------------------------
        if (WU < 0 && p.g0_batch_stride) {  // per-member drift: the whole tile changes with b
            const double *G0b = p.G0 + (long long)b * p.g0_batch_stride;
            for (int e = tid; e < nn; e += 512) G[(e % n) + LD * (e / n)] = G0b[e];
            __syncthreads();  // the dense rewrite lands before the pattern update
        }
        if (!(p.ablate & 8)) {
            if (WU > 0) {
#pragma unroll
                for (int r = 0; r < PCL_NUE2; ++r)
                    if (un_idx[r] >= 0) {
                        double g = un_g0[r];
#pragma unroll
                        for (int w = 0; w < WUR; ++w) g += usc[un_l[r][w]] * un_v[r][w];
                        G[un_idx[r]] = g;
                    }
            } else {
                const double *G0b = p.G0 + (long long)b * p.g0_batch_stride;
                for (int q = tid; q < p.n_upos; q += 512) {
                    const int pos = p.upos[q];
                    double g = G0b[pos];
                    const double *cf = p.ucoef + (long long)q * m;
                    for (int l = 0; l < m; ++l) g += usc[l] * cf[l];
                    G[(pos % n) + LD * (pos / n)] = g;
                }
            }
        }
        if (pf_x) {
            if (ract && rj0 < nc) {
                M1[ri + LD * rj0] = pf_xn + pf_xc;
                M1[ri + LD * (nc + rj0)] = pf_xn - pf_xc;
            }
        } else if (ract) {
            for (int c = rj0; c < nc; c += rstep) {
                double xs = 0.0, xdv = 0.0;
                if (c < nce) {
                    const double xn = zn[x_off + (c0 + c) * n + ri], xc = zk[x_off + (c0 + c) * n + ri];
                    xs = xn + xc;
                    xdv = xn - xc;
                }
                M1[ri + LD * c] = xs;
                M1[ri + LD * (nc + c)] = xdv;
            }
        }
        __syncthreads();

        // ---- phase 1: matrix waves G^2 ; stream waves G_l D ---------------------------------------------
        if (JAC) {
            if (matrix_wave) {
                if (!(p.ablate & 1)) {
                    if (p.iso)
                        wave_rowgemm<1>(G, LD, G, LD, G2, LD, n, d, n, wave, 4, lane, d);
                    else
                        wave_rowgemm<0>(G, LD, G, LD, G2, LD, n, n, n, wave, 4, lane, 0);
                }
            } else if (sact && !(p.ablate & 16)) {
                const double *Dm = M1 + LD * nc;
                for (int cl = sj0; cl < m * nc; cl += sstep) {
                    const int l = cl / nc, c = cl - l * nc;
                    const int base = (l * n + si) * ew;
                    double acc = 0.0;
                    if (stage) {
                        for (int q = 0; q < ew; ++q) acc += ellv_l[base + q] * Dm[ellc_l[base + q] + LD * c];
                    } else {
                        for (int q = 0; q < ew; ++q) acc += p.ell_val[base + q] * Dm[p.ell_col[base + q] + LD * c];
                    }
                    M1[si + LD * (2 * nc + cl)] = acc;
                }
            }
            __syncthreads();
        }

        // ---- phase 2: matrix waves W1 = G M1, G2D = G^2 D ; stream waves the block copies -------------------
        const long long bk = (long long)b * p.K + k;
        double *jb = JAC ? p.jac + bk * p.jac_per : nullptr;
        const long long blk = p.compact == 1 ? (long long)nn : (long long)C * nn;  // size of seg 0 / seg 1 in `jac`
        if (matrix_wave) {
            if (!(p.ablate & 1)) {
                if (fused_p2) {
                    wave_phase2_fused(G, G2, M1, W1, G2D, LD, n, ncols1, nc, JAC, wave, lane);
                } else {
                    wave_rowgemm<0>(G, LD, M1, LD, W1, LD, n, ncols1, n, wave, 4, lane, 0);
                    if (JAC) wave_rowgemm<0>(G2, LD, M1 + LD * nc, LD, G2D, LD, n, nc, n, wave, 4, lane, 0);
                }
            }
        } else if (JAC && pact && !(p.ablate & 2)) {
            int cbeg = c0, cend = c0 + nce;
            if (p.compact) {  // unique blocks only: slice 0 writes the single copy
                cbeg = 0;
                cend = (s == 0) ? 1 : 0;
            }
            double *ob = p.compact == 2 ? p.blocks + bk * 2 * nn : jb;  // split mode: blocks go to the scratch tiles
            const long long oblk = p.compact == 2 ? (long long)nn : blk;
            for (int j = pj0; j < n; j += pstep) {
                const double g0 = G[pi + LD * j], g1 = G[pi + 1 + LD * j];
                const double h0 = G2[pi + LD * j], h1 = G2[pi + 1 + LD * j];
                const double e0 = ((pi == j) ? 1.0 : 0.0) + c2 * h0, e1 = ((pi + 1 == j) ? 1.0 : 0.0) + c2 * h1;
                const double bp0 = -(e0 + c1 * g0), bp1 = -(e1 + c1 * g1);
                const double bm0 = e0 - c1 * g0, bm1 = e1 - c1 * g1;
                double *o0 = ob + (long long)cbeg * nn + (pi + n * j);
                for (int c = cbeg; c < cend; ++c, o0 += nn) {
                    store2(o0, bp0, bp1, p.nt);
                    store2(o0 + oblk, bm0, bm1, p.nt);
                }
            }
            if (p.compact == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's tile stores have left the CU
        }
        // inputs of this workgroup's next item: in flight during the rest of this one
        if (item + (int)gridDim.x < n_items) request(item + gridDim.x);
        __syncthreads();
        if (JAC && p.compact == 2 && s == 0 && tid == 256) {
            // publish the interval's tiles to the expander kernel (other CUs / XCDs): agent-scope release, then the flag
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_store(p.flags + bk, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (!JAC) {  // eval only: delta needs G (G D), a second dependent product
            if (matrix_wave && !(p.ablate & 1)) wave_rowgemm<0>(G, LD, W1 + LD * nc, LD, G2D, LD, n, nc, n, wave, 4, lane, 0);
            __syncthreads();
        }

        // ---- phase 3: column outputs: block 0 -> delta and d/ddt, block 1+l -> d/du_l ----------------------
        if (ract && !(p.ablate & 4)) {
            const double *GDm = W1 + LD * nc;
            const int ncl = JAC ? (1 + m) * nc : nc;
            for (int cl = rj0; cl < ncl; cl += rstep) {
                const int lb = cl / nc, c = cl - lb * nc;
                if (c >= nce) continue;
                const long long r = (long long)(c0 + c) * n + ri;
                if (lb == 0) {
                    const double gs = W1[ri + LD * c], g2d = G2D[ri + LD * c];
                    if (p.delta) p.delta[bk * xd + r] = M1[ri + LD * (nc + c)] - c1 * gs + c2 * g2d;
                    if (JAC) jb[2 * blk + ((long long)(c0 + c) * (m + 1) + m) * n + ri] = -0.5 * gs + (h * (1.0 / 6.0)) * g2d;
                } else {
                    const int l = lb - 1;
                    const int base = (l * n + ri) * ew;
                    double acc = 0.0;
                    if (stage) {
                        for (int q = 0; q < ew; ++q) {
                            const int col = ellc_l[base + q];
                            acc += ellv_l[base + q] * (-c1 * M1[col + LD * c] + c2 * GDm[col + LD * c]);
                        }
                    } else {
                        for (int q = 0; q < ew; ++q) {
                            const int col = p.ell_col[base + q];
                            acc += p.ell_val[base + q] * (-c1 * M1[col + LD * c] + c2 * GDm[col + LD * c]);
                        }
                    }
                    jb[2 * blk + ((long long)(c0 + c) * (m + 1) + l) * n + ri] = acc + c2 * W1[ri + LD * (nc + cl)];
                }
            }
        }
        if (tid <= m) us[(cur ^ 1) * (m + 1) + tid] = pf_v;  // requested during phase 2
        cur ^= 1;
        __syncthreads();  // LDS is rewritten by the next item's phase 0
    }
}

// ------------------------------------------------------------------------------------------
// Fused residual + Jacobian kernel, version 4: version 2's frame (persistent, 2 workgroups per CU, 4 matrix + 4 stream
// waves, 4 barriers per item) with the block stores of an item SPREAD over four phases instead of one:
//   the stream waves copy the item's -B^+ / B^- values into registers as soon as G^2 exists (start of phase 2) and
//   issue a quarter of the copies in each of  phase 2, phase 3 (this item), phase 0, phase 1 (next item);
//   every other duty (union update of G, S/D, G_l D, MFMA products, column outputs) belongs to the matrix waves.
// Per item the workgroup then spends  sum_j max(matrix phase j, store burst j)  instead of
// (matrix phases 0,1,3) + max(matrix phase 2, all stores): the store queue of the CU is fed in every phase.
// ------------------------------------------------------------------------------------------
#define PCL_NUE4 4  // union-pattern entries per matrix-wave thread held in registers (REG path: n_upos <= 1024)

template <int WU, int TD, int TM, int TNC>
__global__ __launch_bounds__(512, 4) void pcl_fused_kernel_v4(const KParams p) {
    extern __shared__ double lds[];
    const int d = TD ? TD : p.d, n = 2 * d, m = TD ? TM : p.m, LD = TD ? ((2 * TD + 3) & ~3) + 2 : p.LD, nc = TNC ? TNC : p.nc;
    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const bool matrix_wave = wave < 4;
    const int nn = n * n;
    const int ncols1 = (2 + m) * nc;
    const int n_ell = m * n * p.ell_w;
    double *G = lds;
    double *G2 = G + LD * n;
    double *M1 = G2 + LD * n;
    double *W1 = M1 + LD * ncols1;
    double *G2D = W1 + LD * ncols1;
    double *us = G2D + LD * nc;  // 2 x [u_k (m) | dt_k]: current / next item
    double *ellv_l = us + 2 * (m + 1);
    unsigned short *ellc_l = reinterpret_cast<unsigned short *>(ellv_l + n_ell);
    const long long xd = (long long)n * d;
    const int ew = p.ell_w;
    const bool fused_p2 = ncols1 <= 32 && 2 * nc <= 16;
    const bool stage = p.ell_lds;
    const int n_items = p.batch * p.K * p.S;
    const long long blk = p.compact ? (long long)nn : (long long)d * nn;  // size of seg 0 / seg 1

    if (!p.g0_batch_stride)
        for (int e = tid; e < nn; e += 512) G[(e % n) + LD * (e / n)] = p.G0[e];
    if (stage)
        for (int e = tid; e < n_ell; e += 512) {
            ellv_l[e] = p.ell_val[e];
            ellc_l[e] = (unsigned short)p.ell_col[e];
        }

    if (matrix_wave) {
        // ===================================== matrix waves (256 threads) ======================================
        const int ri = tid % n, rj0 = tid / n, rstep = 256 / n;  // row ri, columns rj0, rj0+rstep, ..
        const bool ract = rj0 < rstep;
        constexpr int WUR = WU > 0 ? WU : 1;
        int un_idx[PCL_NUE4];
        double un_g0[PCL_NUE4];
        unsigned char un_l[PCL_NUE4][WUR];
        double un_v[PCL_NUE4][WUR];
        if (WU > 0) {
#pragma unroll
            for (int r = 0; r < PCL_NUE4; ++r) {
                const int q = tid + 256 * r;
                un_idx[r] = -1;
                un_g0[r] = 0.0;
#pragma unroll
                for (int w = 0; w < WUR; ++w) {
                    un_l[r][w] = 0;
                    un_v[r][w] = 0.0;
                }
                if (q < p.n_upos) {
                    const int pos = p.upos[q];
                    un_idx[r] = (pos % n) + LD * (pos / n);
                    un_g0[r] = p.G0[pos];
#pragma unroll
                    for (int w = 0; w < WUR; ++w) {
                        un_l[r][w] = p.uell_l[q * WUR + w];
                        un_v[r][w] = p.uell_v[q * WUR + w];
                    }
                }
            }
        }
        const bool pf_x = nc <= rstep;  // one state element per thread: prefetchable
        double pf_v = 0.0, pf_xn = 0.0, pf_xc = 0.0;
        auto request = [&](int item) {
            const int s = item % p.S;
            const int k = (item / p.S) % p.K;
            const int b = item / (p.S * p.K);
            const double *zk = p.Z + (long long)b * p.z_batch_stride + (long long)k * p.z_dim;
            if (tid <= m) pf_v = zk[tid < m ? p.u_off + tid : p.dt_off];
            pf_xn = pf_xc = 0.0;
            if (pf_x && ract && rj0 < min(nc, d - s * nc)) {
                const int x_off = p.x_offs[p.z_batch_stride ? 0 : b];
                const long long o = x_off + (long long)(s * nc + rj0) * n + ri;
                pf_xc = zk[o];
                pf_xn = zk[p.z_dim + o];
            }
        };
        int cur = 0;
        if ((int)blockIdx.x < n_items) {
            request(blockIdx.x);
            if (tid <= m) us[tid] = pf_v;
        }
        __syncthreads();  // G = drift, tables staged, us[0] valid

        for (int item = blockIdx.x;; item += gridDim.x) {
            const bool have = item < n_items;
            const int s = have ? item % p.S : 0;
            const int k = have ? (item / p.S) % p.K : 0;
            const int b = have ? item / (p.S * p.K) : 0;
            const int c0 = s * nc;
            const int nce = min(nc, d - c0);
            const double *zk = p.Z + (long long)b * p.z_batch_stride + (long long)k * p.z_dim;
            const double *zn = zk + p.z_dim;
            const int x_off = p.x_offs[p.z_batch_stride ? 0 : b];
            const double *usc = us + cur * (m + 1);
            const double h = usc[m];
            const double c1 = 0.5 * h, c2 = h * h * (1.0 / 12.0);

            // ---- phase 0: G(u_k) on the union pattern, S, D -> LDS -------------------------------------------
            if (WU < 0 && p.g0_batch_stride) {  // per-member drift: the whole tile changes with b
                if (have) {
                    const double *G0b = p.G0 + (long long)b * p.g0_batch_stride;
                    for (int e = tid; e < nn; e += 256) G[(e % n) + LD * (e / n)] = G0b[e];
                }
                __syncthreads();  // (stream waves take part) the dense rewrite lands before the pattern update
            }
            if (have) {
                if (WU > 0) {
#pragma unroll
                    for (int r = 0; r < PCL_NUE4; ++r)
                        if (un_idx[r] >= 0) {
                            double g = un_g0[r];
#pragma unroll
                            for (int w = 0; w < WUR; ++w) g += usc[un_l[r][w]] * un_v[r][w];
                            G[un_idx[r]] = g;
                        }
                } else {
                    const double *G0b = p.G0 + (long long)b * p.g0_batch_stride;
                    for (int q = tid; q < p.n_upos; q += 256) {
                        const int pos = p.upos[q];
                        double g = G0b[pos];
                        const double *cf = p.ucoef + (long long)q * m;
                        for (int l = 0; l < m; ++l) g += usc[l] * cf[l];
                        G[(pos % n) + LD * (pos / n)] = g;
                    }
                }
                if (pf_x) {
                    if (ract && rj0 < nc) {
                        M1[ri + LD * rj0] = pf_xn + pf_xc;
                        M1[ri + LD * (nc + rj0)] = pf_xn - pf_xc;
                    }
                } else if (ract) {
                    for (int c = rj0; c < nc; c += rstep) {
                        double xs = 0.0, xdv = 0.0;
                        if (c < nce) {
                            const double xn = zn[x_off + (c0 + c) * n + ri], xc = zk[x_off + (c0 + c) * n + ri];
                            xs = xn + xc;
                            xdv = xn - xc;
                        }
                        M1[ri + LD * c] = xs;
                        M1[ri + LD * (nc + c)] = xdv;
                    }
                }
            }
            __syncthreads();  // B_a
            if (!have) {
                __syncthreads();  // B_b: the stream waves' last burst pair runs through phases 0 and 1 of this empty item
                break;
            }

            // ---- phase 1: G_l D (VALU) then G^2 (MFMA) ----------------------------------------------------------
            if (ract) {
                const double *Dm = M1 + LD * nc;
                for (int cl = rj0; cl < m * nc; cl += rstep) {
                    const int l = cl / nc, c = cl - l * nc;
                    const int base = (l * n + ri) * ew;
                    double acc = 0.0;
                    if (stage) {
                        for (int q = 0; q < ew; ++q) acc += ellv_l[base + q] * Dm[ellc_l[base + q] + LD * c];
                    } else {
                        for (int q = 0; q < ew; ++q) acc += p.ell_val[base + q] * Dm[p.ell_col[base + q] + LD * c];
                    }
                    M1[ri + LD * (2 * nc + cl)] = acc;
                }
            }
            if (!(p.ablate & 1)) {
                if (p.iso)
                    wave_rowgemm<1>(G, LD, G, LD, G2, LD, n, d, n, wave, 4, lane, d);
                else
                    wave_rowgemm<0>(G, LD, G, LD, G2, LD, n, n, n, wave, 4, lane, 0);
            }
            __syncthreads();  // B_b: G, G^2, M1 complete

            // ---- phase 2: W1 = G M1, G2D = G^2 D ; request the next item's inputs -------------------------------
            if (!(p.ablate & 1)) {
                if (fused_p2) {
                    wave_phase2_fused(G, G2, M1, W1, G2D, LD, n, ncols1, nc, true, wave, lane);
                } else {
                    wave_rowgemm<0>(G, LD, M1, LD, W1, LD, n, ncols1, n, wave, 4, lane, 0);
                    wave_rowgemm<0>(G2, LD, M1 + LD * nc, LD, G2D, LD, n, nc, n, wave, 4, lane, 0);
                }
            }
            if (item + (int)gridDim.x < n_items) request(item + gridDim.x);
            __syncthreads();  // B_c

            // ---- phase 3: column outputs: block 0 -> delta and d/ddt, block 1+l -> d/du_l -----------------------
            if (ract && !(p.ablate & 4)) {
                const long long bk = (long long)b * p.K + k;
                double *jb = p.jac + bk * p.jac_per;
                const double *GDm = W1 + LD * nc;
                for (int cl = rj0; cl < (1 + m) * nc; cl += rstep) {
                    const int lb = cl / nc, c = cl - lb * nc;
                    if (c >= nce) continue;
                    const long long r = (long long)(c0 + c) * n + ri;
                    if (lb == 0) {
                        const double gs = W1[ri + LD * c], g2d = G2D[ri + LD * c];
                        if (p.delta) p.delta[bk * xd + r] = M1[ri + LD * (nc + c)] - c1 * gs + c2 * g2d;
                        jb[2 * blk + ((long long)(c0 + c) * (m + 1) + m) * n + ri] = -0.5 * gs + (h * (1.0 / 6.0)) * g2d;
                    } else {
                        const int l = lb - 1;
                        const int base = (l * n + ri) * ew;
                        double acc = 0.0;
                        if (stage) {
                            for (int q = 0; q < ew; ++q) {
                                const int col = ellc_l[base + q];
                                acc += ellv_l[base + q] * (-c1 * M1[col + LD * c] + c2 * GDm[col + LD * c]);
                            }
                        } else {
                            for (int q = 0; q < ew; ++q) {
                                const int col = p.ell_col[base + q];
                                acc += p.ell_val[base + q] * (-c1 * M1[col + LD * c] + c2 * GDm[col + LD * c]);
                            }
                        }
                        jb[2 * blk + ((long long)(c0 + c) * (m + 1) + l) * n + ri] = acc + c2 * W1[ri + LD * (nc + cl)];
                    }
                }
            }
            if (tid <= m) us[(cur ^ 1) * (m + 1) + tid] = pf_v;  // requested during phase 2
            cur ^= 1;
            __syncthreads();  // B_d
        }
    } else {
        // ===================================== stream waves (256 threads) ======================================
        const int stid = tid - 256;
        const int hn = n >> 1;
        const int pi = 2 * (stid % hn), pj0 = stid / hn, pstep = max(256 / hn, 1);
        const bool pact = pj0 < pstep;
        constexpr int NSP = TD ? (2 * TD + (256 / TD) - 1) / (256 / TD) : 8;  // column steps per thread (<= 8 for n <= 64)
        double bpr[NSP][2], bmr[NSP][2];
        double *sjb = nullptr;
        int ncopy = 0;
        // burst j in 0..3: copies q = (j>>1), (j>>1)+2, .. and the (j&1) half of this thread's column steps -> four equal
        // quarters of the item's stores whatever the copy count
        auto burst = [&](int jq) {
            if (!pact || (p.ablate & 2)) return;
            const int half = ncopy >> 1;
            const int rlo = (jq & 1) ? NSP / 2 : 0, rhi = (jq & 1) ? NSP : NSP / 2;
            for (int q = jq >> 1; q < ncopy; q += 2) {
                const bool minus = q >= half;
                double *o = sjb + (minus ? blk + (long long)(q - half) * nn : (long long)q * nn);
                if (minus) {
#pragma unroll
                    for (int r = 0; r < NSP; ++r) {
                        const int j = pj0 + pstep * r;
                        if (r >= rlo && r < rhi && j < n) store2(o + n * j, bmr[r][0], bmr[r][1], p.nt);
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < NSP; ++r) {
                        const int j = pj0 + pstep * r;
                        if (r >= rlo && r < rhi && j < n) store2(o + n * j, bpr[r][0], bpr[r][1], p.nt);
                    }
                }
            }
        };
        int cur = 0;
        __syncthreads();  // prologue barrier
        for (int item = blockIdx.x;; item += gridDim.x) {
            const bool have = item < n_items;
            if (WU < 0 && p.g0_batch_stride) __syncthreads();
            burst(2);  // previous item, third quarter (phase 0)
            __syncthreads();       // B_a
            burst(3);  // previous item, last quarter (phase 1)
            __syncthreads();       // B_b: this item's G, G^2 complete
            if (!have) break;
            {
                const int s = item % p.S;
                const int k = (item / p.S) % p.K;
                const int b = item / (p.S * p.K);
                const double h = us[cur * (m + 1) + m];
                const double c1 = 0.5 * h, c2 = h * h * (1.0 / 12.0);
                if (pact) {
#pragma unroll
                    for (int r = 0; r < NSP; ++r) {
                        const int j = pj0 + pstep * r;
                        if (j < n) {
                            const double g0 = G[pi + LD * j], g1 = G[pi + 1 + LD * j];
                            const double h0 = G2[pi + LD * j], h1 = G2[pi + 1 + LD * j];
                            const double e0 = ((pi == j) ? 1.0 : 0.0) + c2 * h0, e1 = ((pi + 1 == j) ? 1.0 : 0.0) + c2 * h1;
                            bpr[r][0] = -(e0 + c1 * g0);
                            bpr[r][1] = -(e1 + c1 * g1);
                            bmr[r][0] = e0 - c1 * g0;
                            bmr[r][1] = e1 - c1 * g1;
                        }
                    }
                }
                int cbeg = s * nc, cend = s * nc + min(nc, d - s * nc);
                if (p.compact) {  // unique blocks only: slice 0 writes the single copy
                    cbeg = 0;
                    cend = (s == 0) ? 1 : 0;
                }
                ncopy = 2 * (cend - cbeg);  // -B^+ copies, then B^- copies
                sjb = p.jac + ((long long)b * p.K + k) * p.jac_per + (long long)cbeg * nn + pi;
            }
            burst(0);         // phase 2
            __syncthreads();  // B_c
            burst(1);         // phase 3
            cur ^= 1;
            __syncthreads();   // B_d
        }
    }
}

// ------------------------------------------------------------------------------------------
// Fused residual + Jacobian kernel, version 3 (default): ONE persistent workgroup per CU, four
// "matrix" wavefronts + four "stream" wavefronts, ONE workgroup barrier per work item (b, k, s).
//
//   stream waves  copy the item's -B^+ / B^- values out of the LDS tiles G, G^2 into registers and
//                 then do nothing but issue the replicated 16-byte block stores (the HBM-roofline
//                 stream; they are the waves that sit in the store queue's back-pressure);
//   matrix waves  meanwhile work wave-synchronously (no workgroup barrier among them):
//                 (1) the item's state columns in chunks of ncw columns, one chunk per wave at a time:
//                     M = [S | D | G_l D] -> G*M on the f64 matrix cores -> delta, d/ddt, d/du_l
//                     straight from the accumulator layout to HBM;
//                 (2) G(u) and G^2 of the workgroup's NEXT item into the other half of the
//                     double-buffered G / G^2 tiles (every matrix wave rewrites the whole union
//                     pattern of G itself - identical values - so no wave waits for another before
//                     its G^2 tiles).
// Item time = max(store stream, matrix work); with the matrix work a fraction of the stream the
// kernel runs at the store stream's rate.
// LDS map (doubles): G [2][LD*n] | G2 [2][LD*n] | per matrix wave: M [LD*CW] GD [LD*ncw] G2D [LD*ncw] |
//                    us [3][m+1] | union values | ELL values | (u16) union LDS offsets, ELL columns | (u8) union drives
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

struct V3Tables {  // launch-invariant tables, in LDS when they fit (else in memory)
    const double *unv;          // [n_upos*uw] drive coefficients of the union pattern
    const unsigned char *unl;   // [n_upos*uw] drive index
    const unsigned short *uni;  // [n_upos] LDS offset (row + LD*col) of the pattern entry
    const double *ung0;         // [n_upos] drift value at the pattern entry (shared-drift case)
    const double *ellv;         // [m*n*ew]
    const unsigned short *ellc;
};

// EW: ELL width held in registers for m <= PCL_MREG drives (0: general, tables in LDS / memory).
// TD/TM/TNCW: compile-time Hilbert dimension, drive count, chunk width (0 = run-time values).
template <int EW, int TD, int TM, int TNCW>
__global__ __launch_bounds__(512, 2) void pcl_fused_kernel_v3(const KParams p) {
    extern __shared__ double lds[];
    const int d = TD ? TD : p.d, n = 2 * d, m = TD ? TM : p.m, LD = TD ? ((2 * TD + 3) & ~3) + 2 : p.LD;
    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const int nn = n * n;
    const int ew = p.ell_w, uw = p.uell_w, n_ell = m * n * ew, n_un = p.n_upos;
    const int ncw = TNCW ? TNCW : p.ncw;  // state columns per chunk
    const int colsw = (2 + m) * ncw;   // operand columns per chunk
    const int CW = 16;                 // one 16-column operand tile per chunk (host guarantees colsw <= 16)
    const long long xd = (long long)n * d;
    const int tile = LD * n;

    double *Gb = lds;
    double *G2b = Gb + 2 * tile;
    double *wbuf = G2b + 2 * tile;  // per matrix wave
    const int wsz = LD * (CW + 3 * ncw);
    double *us = wbuf + 4 * wsz;
    double *t_unv = us + 3 * (m + 1);
    double *t_ung0 = t_unv + (p.tab_lds ? n_un * uw : 0);
    double *t_ellv = t_ung0 + (p.tab_lds ? n_un : 0);
    unsigned short *t_uni = reinterpret_cast<unsigned short *>(t_ellv + (p.tab_lds ? n_ell : 0));
    unsigned short *t_ellc = t_uni + (p.tab_lds ? n_un : 0);
    unsigned char *t_unl = reinterpret_cast<unsigned char *>(t_ellc + (p.tab_lds ? n_ell : 0));

    // Work split.  contig = 0: items (b, k, slice of nc columns) dealt round-robin to the workgroups.
    // contig = 1: the batch*K*d state columns of the launch are cut into gridDim.x equal contiguous ranges (to within one
    // column); a workgroup's items are the pieces of its range that lie in one interval (first and last piece partial),
    // so G, G^2 are built once per interval touched and every CU streams the same number of bytes.
    const long long blk = p.compact ? (long long)nn : (long long)d * nn;  // size of seg 0 / seg 1
    const int nc = p.nc;
    int n_my;
    long long g_lo = 0, g_hi = 0;
    // Role split (contig only, n_stream > 0): workgroups [0, n_stream) stream the B^{+-} blocks of ALL columns (their
    // matrix waves only build G, G^2), workgroups [n_stream, grid) do the column work of ALL columns (their stream waves
    // idle).  The store stream is memory-side bound and half the CUs sustain it; on a CU of its own it is not slowed
    // by the matrix waves' instructions and memory operations.
    const bool stream_role = p.n_stream > 0 && (int)blockIdx.x < p.n_stream;
    const bool matrix_role = p.n_stream > 0 && !stream_role;
    // stream role, optional: pieces of p.snc columns dealt round-robin to the stream workgroups (at any moment they then
    // write one window of ~n_stream/S consecutive intervals instead of n_stream far-apart ranges)
    const bool srr = stream_role && p.snc > 0;
    const int sS = srr ? (d + p.snc - 1) / p.snc : 1;
    if (p.contig && !srr) {
        const long long tot = (long long)p.batch * p.K * d;
        const long long widx = matrix_role ? (long long)blockIdx.x - p.n_stream : (long long)blockIdx.x;
        const long long wcnt = p.n_stream > 0 ? (stream_role ? (long long)p.n_stream : (long long)gridDim.x - p.n_stream) : (long long)gridDim.x;
        g_lo = tot * widx / wcnt;
        g_hi = tot * (widx + 1) / wcnt;
        n_my = g_hi > g_lo ? (int)((g_hi - 1) / d - g_lo / d) + 1 : 0;
    } else if (srr) {
        const int n_items = p.batch * p.K * sS;
        n_my = n_items > (int)blockIdx.x ? (n_items - (int)blockIdx.x + p.n_stream - 1) / p.n_stream : 0;
    } else {
        const int n_items = p.batch * p.K * p.S;
        n_my = n_items > (int)blockIdx.x ? (n_items - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x : 0;
    }
    // item `it` of this workgroup: interval (b, k), state columns [c0, c0 + nce)
    auto decode = [&](int it, int &c0, int &nce, int &k, int &b) {
        if (p.contig && !srr) {
            const long long bk = g_lo / d + it;
            c0 = it == 0 ? (int)(g_lo - bk * d) : 0;
            nce = (int)min((long long)d, g_hi - bk * d) - c0;
            k = (int)(bk % p.K);
            b = (int)(bk / p.K);
        } else {
            const int S_ = srr ? sS : p.S, nc_ = srr ? p.snc : nc;
            const int item = blockIdx.x + it * (srr ? p.n_stream : (int)gridDim.x);
            const int s = item % S_;
            c0 = s * nc_;
            nce = min(nc_, d - c0);
            k = (item / S_) % p.K;
            b = item / (S_ * p.K);
        }
    };

    // item 0's controls / time step: requested before the prologue's table loads so that the latencies overlap
    double u0 = 0.0;
    if (n_my > 0 && wave < 4 && lane <= m) {
        int c00, nce0, k0, b0;
        decode(0, c00, nce0, k0, b0);
        const double *z0 = p.Z + (long long)b0 * p.z_batch_stride + (long long)k0 * p.z_dim;
        u0 = z0[lane < m ? p.u_off + lane : p.dt_off];
    }
    // ---- prologue: both G buffers = drift tile, tables -> LDS ----------------------------------------------
    if (!p.g0_batch_stride)
        for (int e = tid; e < nn; e += 512) {
            const double g = p.G0[e];
            const int o = (e % n) + LD * (e / n);
            Gb[o] = g;
            Gb[tile + o] = g;
        }
    if (p.tab_lds) {
        for (int e = tid; e < n_un * uw; e += 512) {
            t_unv[e] = p.uell_v[e];
            t_unl[e] = p.uell_l[e];
        }
        for (int e = tid; e < n_un; e += 512) {
            const int pos = p.upos[e];
            t_uni[e] = (unsigned short)((pos % n) + LD * (pos / n));
            t_ung0[e] = p.g0_batch_stride ? 0.0 : p.ug0[e];  // (not G0[pos]: no dependent load in the prologue)
        }
        for (int e = tid; e < n_ell; e += 512) {
            t_ellv[e] = p.ell_val[e];
            t_ellc[e] = (unsigned short)p.ell_col[e];
        }
    }
    __syncthreads();


    if (wave < 4 || matrix_role) {
        // ======================================= matrix waves =========================================
        // matrix role: all eight waves work on chunks; G, G^2 are single-buffered there and the second halves of the
        // double buffers hold the chunk buffers of waves 4..7 (the host checks 2*wsz <= tile)
        const int nmw = matrix_role ? 8 : 4;
        double *Mw = wave < 4 ? wbuf + wave * wsz : (wave < 6 ? Gb + tile + (wave - 4) * wsz : G2b + tile + (wave - 6) * wsz);  // [LD*CW]: S | D | G_l D
        double *GDw = Mw + LD * CW;      // [LD*ncw]
        double *G2Dw = GDw + LD * ncw;   // [LD*ncw]
        double *GSw = G2Dw + LD * ncw;   // [LD*ncw]
        const int li = lane & 15, lk = lane >> 4;
        const int rt_n = (n + 15) >> 4;
        const int kfull = n >> 2, krem = n & 3;
        // ELL rows (drive l, row = lane) in registers
        unsigned short er_c[PCL_MREG][EW > 0 ? EW : 1];
        double er_v[PCL_MREG][EW > 0 ? EW : 1];
        if (EW > 0) {
#pragma unroll
            for (int l = 0; l < PCL_MREG; ++l)
#pragma unroll
                for (int q = 0; q < (EW > 0 ? EW : 1); ++q) {
                    er_c[l][q] = 0;
                    er_v[l][q] = 0.0;
                    if (l < m && lane < n) {
                        er_c[l][q] = (unsigned short)p.ell_col[(l * n + lane) * ew + q];
                        er_v[l][q] = p.ell_val[(l * n + lane) * ew + q];
                    }
                }
        }

        // G(u) on the union pattern + this wave's share of the G^2 tiles, for item `it`, into buffer `buf`
        auto build = [&](int it, int buf, double u_lane) {
            int c0_, nce_, k, b;
            decode(it, c0_, nce_, k, b);
            double *G = Gb + buf * tile, *G2 = G2b + buf * tile;
            double *usn = us + (it % 3) * (m + 1);
            if (lane <= m) usn[lane] = u_lane;  // every wave: identical values
            wave_lds_sync();
            const double *G0b = p.G0 + (long long)b * p.g0_batch_stride;
            if (p.g0_batch_stride)  // per-member drift: the whole tile changes with b
                for (int e = lane; e < nn; e += 64) G[(e % n) + LD * (e / n)] = G0b[e];
            if (p.tab_lds) {
                for (int q = lane; q < n_un; q += 64) {
                    double g = p.g0_batch_stride ? G0b[p.upos[q]] : t_ung0[q];
                    for (int w = 0; w < uw; ++w) g += usn[t_unl[q * uw + w]] * t_unv[q * uw + w];
                    G[t_uni[q]] = g;
                }
            } else {
                for (int q = lane; q < n_un; q += 64) {
                    const int pos = p.upos[q];
                    double g = G0b[pos];
                    const double *cf = p.ucoef + (long long)q * m;
                    for (int l = 0; l < m; ++l) g += usn[l] * cf[l];
                    G[(pos % n) + LD * (pos / n)] = g;
                }
            }
            wave_lds_sync();
            if (p.ablate & 1) return;
            // G^2: row tile rt = wave (+4..), all column tiles; with the iso structure only the first d columns
            const int ct_n = p.iso ? (d + 15) >> 4 : rt_n;
            const int Nc = p.iso ? d : n;
            for (int rt = wave; rt < rt_n; rt += 4) {
                const double *Ap = G + rt * 16 + li + LD * lk;
                for (int ct = 0; ct < ct_n; ct += 2) {
                    const bool two = ct + 1 < ct_n;
                    const double *Bp0 = G + lk + LD * (ct * 16 + li);
                    const double *Bp1 = Bp0 + (two ? LD * 16 : 0);
                    double4_t acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
                    double an = 0.0, b0n = 0.0, b1n = 0.0;
                    if (kfull > 0) {
                        an = Ap[0];
                        b0n = Bp0[0];
                        b1n = Bp1[0];
                    }
                    for (int ks = 0; ks < kfull; ++ks) {
                        const double a = an, b0 = b0n, b1 = b1n;
                        if (ks + 1 < kfull) {
                            an = Ap[LD * 4 * (ks + 1)];
                            b0n = Bp0[4 * (ks + 1)];
                            b1n = Bp1[4 * (ks + 1)];
                        }
                        acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b0, acc0, 0, 0, 0);
                        if (two) acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b1, acc1, 0, 0, 0);
                    }
                    if (krem) {
                        const bool ok = lk < krem;
                        const double a = ok ? Ap[LD * 4 * kfull] : 0.0;
                        const double b0 = ok ? Bp0[4 * kfull] : 0.0;
                        const double b1 = ok ? Bp1[4 * kfull] : 0.0;
                        acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b0, acc0, 0, 0, 0);
                        if (two) acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b1, acc1, 0, 0, 0);
                    }
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        const int col = (ct + t) * 16 + li;
                        if ((t == 0 || two) && col < Nc) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                const int row = rt * 16 + lk + 4 * r;
                                const double v = t ? acc1[r] : acc0[r];
                                if (row < n) {
                                    G2[row + LD * col] = v;
                                    if (p.iso) {
                                        if (row < d)
                                            G2[row + d + LD * (col + d)] = v;
                                        else
                                            G2[row - d + LD * (col + d)] = -v;
                                    }
                                }
                            }
                        }
                    }
                }
            }
        };

        if (n_my > 0 && wave < 4) build(0, 0, u0);
        __syncthreads();  // item 0's G, G^2 complete

        for (int it = 0; it < n_my; ++it) {
            const int cur = matrix_role ? 0 : (it & 1);
            int c0, nce, k, b;
            decode(it, c0, nce, k, b);
            const double *zk = p.Z + (long long)b * p.z_batch_stride + (long long)k * p.z_dim;
            const double *zn = zk + p.z_dim;
            const int x_off = p.x_offs[p.z_batch_stride ? 0 : b];
            const double *G = Gb + cur * tile, *G2 = G2b + cur * tile;
            const double h = us[(it % 3) * (m + 1) + m];
            const double c1 = 0.5 * h, c2 = h * h * (1.0 / 12.0), h6 = h * (1.0 / 6.0);
            const long long bk = (long long)b * p.K + k;
            double *jb = p.jac + bk * p.jac_per;
            double *jt = jb + 2 * blk;  // tail: for column c: [d/du_0 .. d/du_{m-1} | d/ddt], n doubles each

            const int nchunk = (nce + ncw - 1) / ncw;
            int stamp = 0;
#define PCL_STAMP()                                                                                     \
    do {                                                                                                \
        if (p.dbg && blockIdx.x == 0 && wave == 0 && lane == 0 && it == 1 && stamp < 60) p.dbg[stamp++] = (long long)__builtin_amdgcn_s_memtime(); \
    } while (0)
            PCL_STAMP();
            // All global reads of this item are issued here, before the wave has any of the item's stores in flight:
            // a later load would sit behind them in the CU's saturated memory pipeline (and vmcnt is in-order).
            const bool pf = ncw <= PCL_PFW;
            double pxn[PCL_PFC][PCL_PFW], pxc[PCL_PFC][PCL_PFW];
            if (pf && lane < n && !(p.ablate & 4) && !stream_role) {
#pragma unroll
                for (int t = 0; t < PCL_PFC; ++t)
#pragma unroll
                    for (int c = 0; c < PCL_PFW; ++c) {
                        const int col = c0 + (wave + nmw * t) * ncw + c;
                        pxn[t][c] = pxc[t][c] = 0.0;
                        if (c < ncw && col < c0 + nce) {
                            const long long o = x_off + (long long)col * n + lane;
                            if (p.ablate & 8) {  // DEBUG: no state loads
                                pxn[t][c] = 1e-3 * lane;
                                pxc[t][c] = 1e-3 * col;
                            } else {
                                pxn[t][c] = zn[o];
                                pxc[t][c] = zk[o];
                            }
                        }
                    }
            }
            double pf_u = 0.0;  // next item's u_k / dt_k (consumed by build)
            if (it + 1 < n_my && lane <= m && wave < 4) {
                int c02, nce2, k2, b2;
                decode(it + 1, c02, nce2, k2, b2);
                const double *zk2 = p.Z + (long long)b2 * p.z_batch_stride + (long long)k2 * p.z_dim;
                pf_u = zk2[lane < m ? p.u_off + lane : p.dt_off];
            }
            int tch = 0;
            for (int ch = wave; ch < nchunk && !(p.ablate & 4) && !stream_role; ch += nmw, ++tch) {
                const int cc0 = c0 + ch * ncw;             // first state column of the chunk
                const int ncc = min(ncw, c0 + nce - cc0);  // columns in this chunk
                // ---- M = [S | D | G_l D]   (lane = row) -----------------------------------------------------
                if (lane < n) {
                    if (pf && tch < PCL_PFC) {
#pragma unroll
                        for (int t = 0; t < PCL_PFC; ++t)
                            if (t == tch) {
#pragma unroll
                                for (int c = 0; c < PCL_PFW; ++c)
                                    if (c < ncw) {
                                        Mw[lane + LD * c] = pxn[t][c] + pxc[t][c];
                                        Mw[lane + LD * (ncw + c)] = pxn[t][c] - pxc[t][c];
                                    }
                            }
                    } else {
                        for (int c = 0; c < ncw; ++c) {
                            double xs = 0.0, xdv = 0.0;
                            if (c < ncc) {
                                const long long o = x_off + (long long)(cc0 + c) * n + lane;
                                const double xn = zn[o], xc = zk[o];
                                xs = xn + xc;
                                xdv = xn - xc;
                            }
                            Mw[lane + LD * c] = xs;
                            Mw[lane + LD * (ncw + c)] = xdv;
                        }
                    }
                    for (int c = colsw; c < CW; ++c) Mw[lane + LD * c] = 0.0;
                }
                wave_lds_sync();
                PCL_STAMP();  // S, D loaded
                const double *Dm = Mw + LD * ncw;
                if (lane < n) {
                    if (EW > 0) {
#pragma unroll
                        for (int l = 0; l < PCL_MREG; ++l)
                            if (l < m)
                                for (int c = 0; c < ncw; ++c) {
                                    double acc = 0.0;
#pragma unroll
                                    for (int q = 0; q < (EW > 0 ? EW : 1); ++q) acc += er_v[l][q] * Dm[er_c[l][q] + LD * c];
                                    Mw[lane + LD * (2 * ncw + l * ncw + c)] = acc;
                                }
                    } else {
                        for (int l = 0; l < m; ++l)
                            for (int c = 0; c < ncw; ++c) {
                                const int base = (l * n + lane) * ew;
                                double acc = 0.0;
                                for (int q = 0; q < ew; ++q) {
                                    const int col = p.tab_lds ? (int)t_ellc[base + q] : p.ell_col[base + q];
                                    const double ev = p.tab_lds ? t_ellv[base + q] : p.ell_val[base + q];
                                    acc += ev * Dm[col + LD * c];
                                }
                                Mw[lane + LD * (2 * ncw + l * ncw + c)] = acc;
                            }
                    }
                    PCL_STAMP();  // G_l D done
                    // G2D = G^2 D on the VALU, 6 independent partial sums per column
                    for (int c = 0; c < ncw; ++c) {
                        double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0, s4 = 0.0, s5 = 0.0;
                        int kk = 0;
                        for (; kk + 6 <= n; kk += 6) {
                            s0 = fma(G2[lane + LD * kk], Dm[kk + LD * c], s0);
                            s1 = fma(G2[lane + LD * (kk + 1)], Dm[kk + 1 + LD * c], s1);
                            s2 = fma(G2[lane + LD * (kk + 2)], Dm[kk + 2 + LD * c], s2);
                            s3 = fma(G2[lane + LD * (kk + 3)], Dm[kk + 3 + LD * c], s3);
                            s4 = fma(G2[lane + LD * (kk + 4)], Dm[kk + 4 + LD * c], s4);
                            s5 = fma(G2[lane + LD * (kk + 5)], Dm[kk + 5 + LD * c], s5);
                        }
                        for (; kk < n; ++kk) s0 = fma(G2[lane + LD * kk], Dm[kk + LD * c], s0);
                        G2Dw[lane + LD * c] = ((s0 + s1) + (s2 + s3)) + (s4 + s5);
                    }
                }
                wave_lds_sync();
                PCL_STAMP();  // G2D done
                // ---- W = G * M on the matrix cores: all row tiles at once (they share the b operand) ----------
                {
                    const double *Bp = Mw + lk + LD * li;
                    const double *Ap[PCL_MAXRT];
                    bool rok[PCL_MAXRT];
#pragma unroll
                    for (int t = 0; t < PCL_MAXRT; ++t) {
                        rok[t] = t * 16 < n;
                        Ap[t] = G + (rok[t] ? t * 16 : 0) + li + LD * lk;
                    }
                    double4_t acc[PCL_MAXRT];
#pragma unroll
                    for (int t = 0; t < PCL_MAXRT; ++t) acc[t] = double4_t{0.0, 0.0, 0.0, 0.0};
                    if (!(p.ablate & 1)) {
                        double an[PCL_MAXRT], bn = 0.0;
#pragma unroll
                        for (int t = 0; t < PCL_MAXRT; ++t) an[t] = kfull > 0 ? Ap[t][0] : 0.0;
                        if (kfull > 0) bn = Bp[0];
                        for (int ks = 0; ks < kfull; ++ks) {
                            double a[PCL_MAXRT];
                            const double bb = bn;
#pragma unroll
                            for (int t = 0; t < PCL_MAXRT; ++t) a[t] = an[t];
                            if (ks + 1 < kfull) {
#pragma unroll
                                for (int t = 0; t < PCL_MAXRT; ++t) an[t] = Ap[t][LD * 4 * (ks + 1)];
                                bn = Bp[4 * (ks + 1)];
                            }
#pragma unroll
                            for (int t = 0; t < PCL_MAXRT; ++t)
                                if (rok[t]) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[t], bb, acc[t], 0, 0, 0);
                        }
                        if (krem) {
                            const bool ok = lk < krem;
                            const double bb = ok ? Bp[4 * kfull] : 0.0;
#pragma unroll
                            for (int t = 0; t < PCL_MAXRT; ++t)
                                if (rok[t]) {
                                    const double a = ok ? Ap[t][LD * 4 * kfull] : 0.0;
                                    acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bb, acc[t], 0, 0, 0);
                                }
                        }
                    }
                    // accumulator layout -> LDS: column li of [G S | G D | G (G_l D)]; the last group goes back into M's
                    // own columns (their operand role is over)
                    double *dst = li < ncw ? GSw + LD * li : (li < 2 * ncw ? GDw + LD * (li - ncw) : Mw + LD * li);
#pragma unroll
                    for (int t = 0; t < PCL_MAXRT; ++t)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int row = t * 16 + lk + 4 * r;
                            if (row < n && li < colsw) dst[row] = acc[t][r];
                        }
                }
                wave_lds_sync();
                PCL_STAMP();  // MFMA + accumulators -> LDS done
                // ---- outputs: finish in LDS (lane = row, in place), then 16-byte stores --------------------------------
                // in place: delta -> D column, d/ddt -> GS column, d/du_l -> the G (G_l D) column
                if (lane < n) {
                    for (int c = 0; c < ncw; ++c) {
                        const double gs = GSw[lane + LD * c], g2d = G2Dw[lane + LD * c];
                        G2Dw[lane + LD * c] = Mw[lane + LD * (ncw + c)] - c1 * gs + c2 * g2d;  // delta
                        GSw[lane + LD * c] = -0.5 * gs + h6 * g2d;                                // d/ddt
                    }
                    // d/du_l = G_l (-c1 S + c2 G D) + c2 G (G_l D)
                    if (EW > 0) {
#pragma unroll
                        for (int l = 0; l < PCL_MREG; ++l)
                            if (l < m)
                                for (int c = 0; c < ncw; ++c) {
                                    double acc = 0.0;
#pragma unroll
                                    for (int q = 0; q < (EW > 0 ? EW : 1); ++q)
                                        acc += er_v[l][q] * (-c1 * Mw[er_c[l][q] + LD * c] + c2 * GDw[er_c[l][q] + LD * c]);
                                    Mw[lane + LD * (2 * ncw + l * ncw + c)] = acc + c2 * Mw[lane + LD * (2 * ncw + l * ncw + c)];
                                }
                    } else {
                        for (int l = 0; l < m; ++l)
                            for (int c = 0; c < ncw; ++c) {
                                const int base = (l * n + lane) * ew;
                                double acc = 0.0;
                                for (int q = 0; q < ew; ++q) {
                                    const int col = p.tab_lds ? (int)t_ellc[base + q] : p.ell_col[base + q];
                                    const double ev = p.tab_lds ? t_ellv[base + q] : p.ell_val[base + q];
                                    acc += ev * (-c1 * Mw[col + LD * c] + c2 * GDw[col + LD * c]);
                                }
                                Mw[lane + LD * (2 * ncw + l * ncw + c)] = acc + c2 * Mw[lane + LD * (2 * ncw + l * ncw + c)];
                            }
                    }
                }
                wave_lds_sync();
                // the chunk's columns are contiguous in every output vector: element e = c*n + row, two per lane
                {
                    const int hn2 = n >> 1;
                    const long long o0 = (long long)cc0 * n;
                    for (int e2 = lane; e2 < ((p.ablate & 32) ? 0 : ncc * hn2); e2 += 64) {
                        const int c = e2 / hn2, r0 = 2 * (e2 - c * hn2);
                        if (p.delta) store2(p.delta + bk * xd + o0 + (long long)c * n + r0, G2Dw[r0 + LD * c], G2Dw[r0 + 1 + LD * c], false);
                        double *tc = jt + (long long)(cc0 + c) * (m + 1) * n + r0;  // this column's (m+1)*n tail block
                        if (p.ablate & 64) tc = p.jac + (long long)blockIdx.x * 16384 + (wave * 1024 + c * 512) + r0;  // DEBUG: a cache-resident scratch target
                        for (int l = 0; l < m; ++l) {
                            const double *src = Mw + LD * (2 * ncw + l * ncw + c) + r0;
                            store2(tc + (long long)l * n, src[0], src[1], false);
                        }
                        store2(tc + (long long)m * n, GSw[r0 + LD * c], GSw[r0 + 1 + LD * c], false);
                    }
                }
                wave_lds_sync();  // the chunk buffers are rewritten by this wave's next chunk
                PCL_STAMP();  // outputs issued
            }
            // ---- next item's G(u), G^2 into the other buffer ----------------------------------------------------
            if (matrix_role) __syncthreads();  // single-buffered G, G^2: every wave is done with this item's tiles
            if (it + 1 < n_my && wave < 4) build(it + 1, matrix_role ? 0 : cur ^ 1, pf_u);
            PCL_STAMP();  // next G, G^2 built
            __syncthreads();  // item boundary
            PCL_STAMP();  // barrier passed
        }
    } else {
        // ======================================= stream waves =========================================
        const int stid = tid - 256;
        const int hn = n >> 1;
        const int pi = 2 * (stid % hn), pj0 = stid / hn, pstep = max(256 / hn, 1);
        const bool pact = pj0 < pstep;
        __syncthreads();  // item 0's G, G^2 complete
        for (int it = 0; it < n_my; ++it) {
            const int cur = it & 1;
            int c0, nce, k, b;
            decode(it, c0, nce, k, b);
            const double *G = Gb + cur * tile, *G2 = G2b + cur * tile;
            if (p.flat && !(p.ablate & 2) && !matrix_role) {
                // Optional line-aligned flat stream (option aligned_stream): the item's share of a segment (copies
                // cbeg..cend-1 of one n x n block) is ONE contiguous run; after a partial head up to the next 128-byte line
                // every wave-level store covers eight whole lines (1 KiB); values recomputed per store from the LDS tiles.
                // A bare store kernel gains 30-40 % from this alignment (scripts/probes/wstream2.hip); this kernel, whose
                // per-block stores are 1 KiB contiguous per instruction already, does not (28.0 vs 28.2 us/eval).
                const double h = us[(it % 3) * (m + 1) + m];
                const double c1 = 0.5 * h, c2 = h * h * (1.0 / 12.0);
                int cbeg = c0, cend = c0 + nce;
                if (p.compact) {
                    cbeg = 0;
                    cend = (c0 == 0) ? 1 : 0;
                }
                const long long L = (long long)(cend - cbeg) * nn;  // doubles per run
                double *jbk = p.jac + ((long long)b * p.K + k) * p.jac_per + (long long)cbeg * nn;
                const int di = 512 % n, dj = (512 / n) % n;
                auto put = [&](double *A0, long long a, int i, int j, int sg) {
                    const double g0 = G[i + LD * j], g1 = G[i + 1 + LD * j];
                    const double h0 = G2[i + LD * j], h1 = G2[i + 1 + LD * j];
                    const double e0 = ((i == j) ? 1.0 : 0.0) + c2 * h0, e1 = ((i + 1 == j) ? 1.0 : 0.0) + c2 * h1;
                    if (sg == 0)
                        store2(A0 + a, -(e0 + c1 * g0), -(e1 + c1 * g1), p.nt);
                    else
                        store2(A0 + a, e0 - c1 * g0, e1 - c1 * g1, p.nt);
                };
#pragma unroll
                for (int sg = 0; sg < 2; ++sg) {
                    double *A0 = jbk + sg * blk;
                    const int head = (int)(((128 - ((unsigned long long)A0 & 127)) & 127) >> 3);  // doubles (even)
                    if (2 * stid < head && 2 * stid < L) put(A0, 2 * stid, (2 * stid) % n, ((2 * stid) / n) % n, sg);
                    long long a = head + 2LL * stid;
                    int i = (int)(a % n), j = (int)((a / n) % n);
                    for (; a < L; a += 512) {
                        put(A0, a, i, j, sg);
                        i += di;
                        if (i >= n) {
                            i -= n;
                            ++j;
                        }
                        j += dj;
                        if (j >= n) j -= n;
                    }
                }
            } else if (pact && !(p.ablate & 2) && !matrix_role) {
                const double h = us[(it % 3) * (m + 1) + m];
                const double c1 = 0.5 * h, c2 = h * h * (1.0 / 12.0);
                double bpr[PCL_NSP][2], bmr[PCL_NSP][2];
#pragma unroll
                for (int r = 0; r < PCL_NSP; ++r) {
                    const int j = pj0 + pstep * r;
                    if (j < n) {
                        const double g0 = G[pi + LD * j], g1 = G[pi + 1 + LD * j];
                        const double h0 = G2[pi + LD * j], h1 = G2[pi + 1 + LD * j];
                        const double e0 = ((pi == j) ? 1.0 : 0.0) + c2 * h0, e1 = ((pi + 1 == j) ? 1.0 : 0.0) + c2 * h1;
                        bpr[r][0] = -(e0 + c1 * g0);
                        bpr[r][1] = -(e1 + c1 * g1);
                        bmr[r][0] = e0 - c1 * g0;
                        bmr[r][1] = e1 - c1 * g1;
                    }
                }
                int cbeg = c0, cend = c0 + nce;
                if (p.compact) {  // unique blocks only: slice 0 writes the single copy
                    cbeg = 0;
                    cend = (c0 == 0) ? 1 : 0;
                }
                double *o = p.jac + ((long long)b * p.K + k) * p.jac_per + (long long)cbeg * nn + pi;
                for (int c = cbeg; c < cend; ++c, o += nn) {
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
            __syncthreads();  // item boundary
        }
    }
}

// ------------------------------------------------------------------------------------------
// Expansion kernel: compact -> full triplet order (replicate the unique blocks d times).
// grid.x = batch*K*d ; each block copies one (b,k,c) pair of n*n blocks; block c==0 also copies the tail.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pcl_expand_kernel(const double *__restrict__ compact, double *__restrict__ full,
                                                         int d, int n, int m, long long n_bk, int nt) {
    const long long nn = (long long)n * n, xd = (long long)n * d;
    const long long cper = 2 * nn + xd * (m + 1), fper = 2 * d * nn + xd * (m + 1);
    const long long bid = blockIdx.x;
    const int c = (int)(bid % d);
    const long long bk = bid / d;
    if (bk >= n_bk) return;
    const double *src = compact + bk * cper;
    double *dst = full + bk * fper;
    for (long long q = threadIdx.x; q < (nn >> 1); q += blockDim.x) {
        const double2_t v0 = *reinterpret_cast<const double2_t *>(src + 2 * q);
        const double2_t v1 = *reinterpret_cast<const double2_t *>(src + nn + 2 * q);
        store2(dst + c * nn + 2 * q, v0[0], v0[1], nt);
        store2(dst + (d + c) * nn + 2 * q, v1[0], v1[1], nt);
    }
    if (c == 0) {
        const long long tail = xd * (m + 1);
        for (long long q = threadIdx.x; q < (tail >> 1); q += blockDim.x) {
            const double2_t v = *reinterpret_cast<const double2_t *>(src + 2 * nn + 2 * q);
            store2(dst + 2 * d * nn + 2 * q, v[0], v[1], nt);
        }
    }
}

// ------------------------------------------------------------------------------------------
// Split mode, consumer side: persistent expander.  Work item = (b, k, piece): `cpp` of the 2d block copies of one
// interval.  The workgroup waits until the producer kernel (running concurrently on another stream) has raised the
// interval's flag, loads the unique -B^+ / B^- tile values it needs from the scratch tiles (L2) into registers and
// streams the copies.  No LDS, no barrier inside the stream.  Hand-off protocol: producer = stores, s_waitcnt vmcnt(0),
// __syncthreads, one lane: agent-scope release fence + relaxed agent-scope flag store; consumer = one lane polls the
// flag (relaxed, agent scope), agent-scope acquire fence, __syncthreads, plain loads.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pcl_expand_stream_kernel(const double *blocks, const unsigned int *flags,
                                                                double *__restrict__ jac, int d, int n, long long jac_per,
                                                                long long n_bk, int pieces, int cpp, int nt) {
    const int nn = n * n;
    const int tid = threadIdx.x;
    const int hn = n >> 1;
    const int pi = 2 * (tid % hn), pj0 = tid / hn, pstep = max(256 / hn, 1);
    const bool pact = pj0 < pstep;
    const long long n_items = n_bk * pieces;
    const long long blk = (long long)d * nn;
    for (long long item = blockIdx.x; item < n_items; item += gridDim.x) {
        const long long bk = item / pieces;
        const int piece = (int)(item - bk * pieces);
        if (tid == 0) {
            // bounded spin (about a second): a producer that never shows up must not hang the device
            int spins = 0;
            while (__hip_atomic_load(flags + bk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u && spins < (1 << 22)) {
                __builtin_amdgcn_s_sleep(8);
                ++spins;
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        __syncthreads();
        if (pact) {
            const double *src = blocks + bk * 2 * nn + pi;
            double2_t vp[8], vm[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const int j = pj0 + pstep * r;
                if (j < n) {
                    vp[r] = *reinterpret_cast<const double2_t *>(src + n * j);
                    vm[r] = *reinterpret_cast<const double2_t *>(src + nn + n * j);
                }
            }
            // copies q in [piece*cpp, ..): q < d are -B^+ copies, q >= d are B^- copies
            const int q0 = piece * cpp, q1 = min(2 * d, q0 + cpp);
            double *dst = jac + bk * jac_per + pi;
            for (int q = q0; q < q1; ++q) {
                if (q < d) {
                    double *o = dst + (long long)q * nn;
#pragma unroll
                    for (int r = 0; r < 8; ++r) {
                        const int j = pj0 + pstep * r;
                        if (j < n) store2(o + n * j, vp[r][0], vp[r][1], nt);
                    }
                } else {
                    double *o = dst + blk + (long long)(q - d) * nn;
#pragma unroll
                    for (int r = 0; r < 8; ++r) {
                        const int j = pj0 + pstep * r;
                        if (j < n) store2(o + n * j, vm[r][0], vm[r][1], nt);
                    }
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// Hessian-of-Lagrangian kernel: one workgroup per (b, k); the d state columns are processed in
// chunks of nc columns (columns are independent except for the (m+1)(m+2)/2 scalar entries,
// whose per-chunk partial sums are accumulated in LDS in a fixed order -> deterministic).
// With M = mu_k (n x d):  A1 = G^T M, A2 = G^T A1, P_l = G_l^T M, Q_l = G^T P_l, R_l = G_l^T A1,
//                         GD = G D, E_l = G_l D.
// LDS map: G [LD*n] | Mm | S | D | GD | A1 | A2 (each LD*nc) | P | Q | E (each m*LD*nc) | us | red | acc
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ double wave_sum(double v) {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

template <bool MFMA>
__global__ __launch_bounds__(256) void pcl_hess_kernel(const KParams p) {
    extern __shared__ double lds[];
    const int n = p.n, d = p.cols, m = p.m, LD = p.LD, nc = p.nc;  // d: state columns here
    const int tid = threadIdx.x, nth = blockDim.x;
    const int k = blockIdx.x % p.K, b = blockIdx.x / p.K;
    const long long xd = (long long)n * d;
    const int LDc = LD * nc;
    const int nscal = (m + 1) * (m + 2) / 2;
    const int nw = nth >> 6, wv = tid >> 6, lane = tid & 63;

    double *G = lds;
    double *Mm = G + LD * n;
    double *Sm = Mm + LDc;
    double *Dm = Sm + LDc;
    double *GD = Dm + LDc;
    double *A1 = GD + LDc;
    double *A2 = A1 + LDc;
    double *P = A2 + LDc;
    double *Q = P + m * LDc;
    double *E = Q + m * LDc;
    double *us = E + m * LDc;
    double *red = us + 8 + m;       // nw * nscal
    double *acc = red + nw * nscal;  // nscal

    const double *Zb = p.Z + (long long)b * p.z_batch_stride;
    const double *zk = Zb + (long long)k * p.z_dim;
    const double *zn = zk + p.z_dim;
    const int x_off = p.x_offs[p.z_batch_stride ? 0 : b];
    const double h = zk[p.dt_off];
    const double c1 = 0.5 * h, c2 = h * h * (1.0 / 12.0), h6 = h * (1.0 / 6.0);
    const long long bk = (long long)b * p.K + k;
    const double *mu = p.mu + bk * xd;
    double *H = p.hess + bk * p.hess_per;
    double *H3 = H + nscal, *H4 = H3 + (long long)m * xd, *H5 = H4 + xd, *H6 = H5 + (long long)m * xd;

    build_G(p, p.G0 + (long long)b * p.g0_batch_stride, zk, G, us);
    for (int e = tid; e < nscal; e += nth) acc[e] = 0.0;

    for (int c0 = 0; c0 < d; c0 += nc) {
        const int nce = min(nc, d - c0);
        __syncthreads();  // previous chunk fully consumed (and G / acc initialised)
        for (int e = tid; e < nce * n; e += nth) {
            const int c = e / n, i = e % n;
            const long long g = (long long)(c0 + c) * n + i;
            const double xn = zn[x_off + g], xc = zk[x_off + g];
            Sm[i + LD * c] = xn + xc;
            Dm[i + LD * c] = xn - xc;
            Mm[i + LD * c] = mu[g];
        }
        __syncthreads();
        // sparse applications: P_l = G_l^T M (CSC columns of G_l), E_l = G_l D (CSR rows)
        for (int e = tid; e < m * nce * n; e += nth) {
            const int i = e % n, c = (e / n) % nce, l = e / (n * nce);
            const int *cp = p.csc_ptr + l * (n + 1);
            double a = 0.0;
            for (int q = cp[i]; q < cp[i + 1]; ++q) a += p.csc_val[q] * Mm[p.csc_row[q] + LD * c];
            P[l * LDc + i + LD * c] = a;
            const int *rp = p.csr_ptr + l * (n + 1);
            double a2 = 0.0;
            for (int q = rp[i]; q < rp[i + 1]; ++q) a2 += p.csr_val[q] * Dm[p.csr_col[q] + LD * c];
            E[l * LDc + i + LD * c] = a2;
        }
        __syncthreads();
        gemm_lds<MFMA, false>(G, LD, Dm, LD, GD, LD, n, nce, n);
        gemm_lds<MFMA, true>(G, LD, Mm, LD, A1, LD, n, nce, n);
        for (int l = 0; l < m; ++l) gemm_lds<MFMA, true>(G, LD, P + l * LDc, LD, Q + l * LDc, LD, n, nce, n);
        __syncthreads();
        gemm_lds<MFMA, true>(G, LD, A1, LD, A2, LD, n, nce, n);

        // ---- scalar segments 0..2: per-wave partial sums -> red[w][pidx] ----------------------
        int pidx = 0;
        for (int i = 0; i < m; ++i)
            for (int j = 0; j <= i; ++j, ++pidx) {
                double v = 0.0;
                for (int e = tid; e < nce * n; e += nth) {
                    const int idx = (e % n) + LD * (e / n);
                    v += P[i * LDc + idx] * E[j * LDc + idx] + P[j * LDc + idx] * E[i * LDc + idx];
                }
                v = wave_sum(v);
                if (lane == 0) red[wv * nscal + pidx] = c2 * v;
            }
        for (int j = 0; j < m; ++j, ++pidx) {
            double v1 = 0.0, v2 = 0.0;
            for (int e = tid; e < nce * n; e += nth) {
                const int idx = (e % n) + LD * (e / n);
                v1 += P[j * LDc + idx] * Sm[idx];
                v2 += P[j * LDc + idx] * GD[idx] + A1[idx] * E[j * LDc + idx];
            }
            v1 = wave_sum(v1);
            v2 = wave_sum(v2);
            if (lane == 0) red[wv * nscal + pidx] = -0.5 * v1 + h6 * v2;
        }
        {
            double v = 0.0;
            for (int e = tid; e < nce * n; e += nth) {
                const int idx = (e % n) + LD * (e / n);
                v += A1[idx] * GD[idx];
            }
            v = wave_sum(v);
            if (lane == 0) red[wv * nscal + pidx] = v * (1.0 / 6.0);
        }
        __syncthreads();  // red complete, A2 complete
        for (int e = tid; e < nscal; e += nth) {
            double t = acc[e];
            for (int w = 0; w < nw; ++w) t += red[w * nscal + e];
            acc[e] = t;
        }
        // ---- vector segments 3..6 for this chunk's columns --------------------------------------
        for (int e = tid; e < m * nce * n; e += nth) {
            const int i = e % n, c = (e / n) % nce, l = e / (n * nce);
            const int *cp = p.csc_ptr + l * (n + 1);
            double r = 0.0;  // R_l = G_l^T (G^T M)
            for (int q = cp[i]; q < cp[i + 1]; ++q) r += p.csc_val[q] * A1[p.csc_row[q] + LD * c];
            const int idx = i + LD * c;
            const double kt = c2 * (Q[l * LDc + idx] + r);
            const double pl = -c1 * P[l * LDc + idx];
            const long long o = (long long)l * xd + (long long)(c0 + c) * n + i;
            H3[o] = pl - kt;
            H5[o] = pl + kt;
        }
        for (int e = tid; e < nce * n; e += nth) {
            const int idx = (e % n) + LD * (e / n);
            const long long o = (long long)c0 * n + e;
            H4[o] = -0.5 * A1[idx] - h6 * A2[idx];
            H6[o] = -0.5 * A1[idx] + h6 * A2[idx];
        }
    }
    __syncthreads();
    for (int e = tid; e < nscal; e += nth) H[e] = acc[e];
}

// ------------------------------------------------------------------------------------------
// Hessian-of-Lagrangian kernel, version 2 (default when every drive row / column has <= EW entries and m <= 6):
// persistent workgroups (2 per CU, 4 wavefronts each) over work items (b, k, slice of <= 16 state columns).
// Per item the four waves work wave-synchronously on chunks of NCW = 16/(m+1) columns:
//     operand tile  [M | P_1 .. P_m],  P_l = G_l^T M  (ELL rows of G_l^T in registers, lane = row)
//     one pass of the f64 matrix cores:  G^T [M | P_l] = [A1 | Q_l]
//     R_l = G_l^T A1, E_l = G_l D (registers)  ->  the d2/du dX vectors straight to HBM
//     the (m+1)(m+2)/2 - 1 scalar entries that involve u as per-lane partial sums in registers:
//         <M,(G_i G_j + G_j G_i) D> = <P_i,E_j> + <P_j,E_i>,   <M,G_j S> = <P_j,S>,
//         <M,(G_j G + G G_j) D> = <Q_j,D> + <A1,E_j>
// then, once per item: A2 = G^T A1 for all the slice's columns in ONE matrix-core pass (wave w = row tile w),
// the d2/dh dX vectors, <A2,D>, and a fixed-order reduction lane -> wave -> workgroup -> (slices of the interval,
// summed by the last slice to arrive: partial sums in `hpart`, arrival counter in `hcnt`) -> deterministic.
// No G^2, no G D product: every contraction with D is moved onto M's side.
// LDS map (doubles): G [LD*n] | A1s [LD*16] | A2s [LD*16] | Ds [LD*16] | per wave Mw [LD*16] | wsum [4][NSC] | wsum2 [4] | flag
// ------------------------------------------------------------------------------------------
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

// TD: compile-time Hilbert dimension (0 = run-time).  ANTI: every G_l is exactly antisymmetric (G = iso(-iH) with H
// Hermitian), so the rows of G_l^T are minus the rows of G_l and one ELL table serves both.
template <int EW, int TM, int TD, bool ANTI>
__global__ __launch_bounds__(256, 2) void pcl_hess_kernel_v2(const KParams p) {
    extern __shared__ double lds[];
    constexpr int m = TM;
    constexpr int NCW = 16 / (TM + 1);
    constexpr int NSC = (TM + 1) * (TM + 2) / 2;
    constexpr int NPAIR = TM * (TM + 1) / 2;
    constexpr int NACC = NPAIR + TM;
    const int n = TD ? 2 * TD : p.n, d = p.cols, LD = TD ? ((2 * TD + 3) & ~3) + 2 : p.LD;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int li = lane & 15, lk = lane >> 4;
    const int nn = n * n;
    const long long xd = (long long)n * d;
    const int kfull = n >> 2, krem = n & 3;

    double *G = lds;
    double *A1s = G + LD * n;
    double *A2s = A1s + LD * 16;
    double *Ds = A2s + LD * 16;
    double *Mw = Ds + LD * 16 + wave * (LD * 16);
    double *wsum = Ds + LD * 16 + 4 * (LD * 16);
    double *wsum2 = wsum + 4 * NSC;
    int *lastflag = reinterpret_cast<int *>(wsum2 + 4);

    // ELL rows of G_l (er) and of G_l^T (et) for row = lane
    constexpr int TE = ANTI ? 1 : TM;  // the transposed table is only held when it differs from minus the plain one
    unsigned short er_c[TM][EW], et_c_[TE][EW];
    double er_v[TM][EW], et_v_[TE][EW];
#pragma unroll
    for (int l = 0; l < TM; ++l)
#pragma unroll
        for (int q = 0; q < EW; ++q) {
            er_c[l][q] = 0;
            er_v[l][q] = 0.0;
            if (!ANTI) {
                et_c_[ANTI ? 0 : l][q] = 0;
                et_v_[ANTI ? 0 : l][q] = 0.0;
            }
            if (lane < n) {
                if (q < p.ell_w) {
                    er_c[l][q] = (unsigned short)p.ell_col[(l * n + lane) * p.ell_w + q];
                    er_v[l][q] = p.ell_val[(l * n + lane) * p.ell_w + q];
                }
                if (!ANTI && q < p.ellt_w) {
                    et_c_[ANTI ? 0 : l][q] = (unsigned short)p.ellt_col[(l * n + lane) * p.ellt_w + q];
                    et_v_[ANTI ? 0 : l][q] = p.ellt_val[(l * n + lane) * p.ellt_w + q];
                }
            }
        }
#define ET_C(l, q) (ANTI ? er_c[l][q] : et_c_[ANTI ? 0 : (l)][q])
#define ET_V(l, q) (ANTI ? er_v[l][q] : et_v_[ANTI ? 0 : (l)][q])  // ANTI: the caller negates the sum
    if (!p.g0_batch_stride)
        for (int e = tid; e < nn; e += 256) G[(e % n) + LD * (e / n)] = p.G0[e];

    const int S = p.S, nc = p.nc;
    const int n_items = p.batch * p.K * S;
    int stamp = 0;
#define PCL_HSTAMP()                                                                                  \
    do {                                                                                             \
        if (p.dbg && blockIdx.x == 0 && tid == 0 && stamp < 60) p.dbg[stamp++] = (long long)__builtin_amdgcn_s_memtime(); \
    } while (0)
    PCL_HSTAMP();  // prologue done
    for (int item = blockIdx.x; item < n_items; item += gridDim.x) {
        const int s = item % S, k = (item / S) % p.K, b = item / (S * p.K);
        const int c0 = s * nc, nce = min(nc, d - c0);
        const double *zk = p.Z + (long long)b * p.z_batch_stride + (long long)k * p.z_dim;
        const double *zn = zk + p.z_dim;
        const int x_off = p.x_offs[p.z_batch_stride ? 0 : b];
        const double h = zk[p.dt_off];
        const double c1 = 0.5 * h, c2 = h * h * (1.0 / 12.0), h6 = h * (1.0 / 6.0);
        const long long bk = (long long)b * p.K + k;
        const double *mu = p.mu + bk * xd;
        double *H = p.hess + bk * p.hess_per;
        double *H3 = H + NSC, *H4 = H3 + (long long)m * xd, *H5 = H4 + xd, *H6 = H5 + (long long)m * xd;

        // inputs of this wave's first chunk: requested before G is built, consumed after
        const int nchunk = (nce + NCW - 1) / NCW;
        double pxn[NCW], pxc[NCW], pmu[NCW];
#pragma unroll
        for (int c = 0; c < NCW; ++c) {
            pxn[c] = pxc[c] = pmu[c] = 0.0;
            if (lane < n && wave < nchunk && wave * NCW + c < nce) {
                const long long g = (long long)(c0 + wave * NCW + c) * n + lane;
                pxn[c] = zn[x_off + g];
                pxc[c] = zk[x_off + g];
                pmu[c] = mu[g];
            }
        }
        __syncthreads();  // the previous item of this workgroup is fully consumed
        PCL_HSTAMP();  // item start
        // ---- G(u_k): drift everywhere (per-member drift only), then the drives' union pattern ------------------
        {
            const double *G0b = p.G0 + (long long)b * p.g0_batch_stride;
            if (p.g0_batch_stride)
                for (int e = tid; e < nn; e += 256)
                    if (p.umap[e] < 0) G[(e % n) + LD * (e / n)] = G0b[e];
            double uu[TM];
#pragma unroll
            for (int l = 0; l < TM; ++l) uu[l] = zk[p.u_off + l];
            for (int q = tid; q < p.n_upos; q += 256) {
                const int pos = p.upos[q];
                double g = p.g0_batch_stride ? G0b[pos] : p.ug0[q];
                const double *cf = p.ucoef + (long long)q * m;
#pragma unroll
                for (int l = 0; l < TM; ++l) g += uu[l] * cf[l];
                G[(pos % n) + LD * (pos / n)] = g;
            }
        }
        __syncthreads();

        PCL_HSTAMP();  // G built
        double acc[NACC];
#pragma unroll
        for (int e = 0; e < NACC; ++e) acc[e] = 0.0;
        for (int ch = wave; ch < nchunk; ch += 4) {
            const int cl0 = ch * NCW;                // first column of the chunk inside the slice
            const int ncc = min(NCW, nce - cl0);     // columns in this chunk
            double Sv[NCW], Pv[TM][NCW];
            if (lane < n) {
#pragma unroll
                for (int c = 0; c < NCW; ++c) {
                    Sv[c] = 0.0;
                    double mv = 0.0;
                    if (c < ncc) {
                        double xn = pxn[c], xc = pxc[c];
                        mv = pmu[c];
                        if (ch != wave) {  // further chunks of a wide slice load at use
                            const long long g = (long long)(c0 + cl0 + c) * n + lane;
                            xn = zn[x_off + g];
                            xc = zk[x_off + g];
                            mv = mu[g];
                        }
                        Sv[c] = xn + xc;
                        Ds[lane + LD * (cl0 + c)] = xn - xc;
                    }
                    Mw[lane + LD * c] = mv;
                }
            }
            wave_lds_sync();
            PCL_HSTAMP();  // inputs loaded
            if (lane < n) {
#pragma unroll
                for (int l = 0; l < TM; ++l)
#pragma unroll
                    for (int c = 0; c < NCW; ++c) {
                        double pv = 0.0;
                        if (c < ncc) {
#pragma unroll
                            for (int q = 0; q < EW; ++q) pv += ET_V(l, q) * Mw[ET_C(l, q) + LD * c];
                            if (ANTI) pv = -pv;
                        }
                        Pv[l][c] = pv;
                        Mw[lane + LD * (NCW + l * NCW + c)] = pv;
                    }
            }
            wave_lds_sync();
            PCL_HSTAMP();  // P, E done
            // ---- [A1 | Q_l] = G^T [M | P_l]: all row tiles at once (they share the b operand) --------------------
            double4_t ac[PCL_MAXRT];
            {
                const double *Bp = Mw + lk + LD * li;
                const double *Ap[PCL_MAXRT];
                bool rok[PCL_MAXRT];
#pragma unroll
                for (int t = 0; t < PCL_MAXRT; ++t) {
                    rok[t] = t * 16 < n;
                    Ap[t] = G + lk + LD * ((rok[t] ? t * 16 : 0) + li);
                    ac[t] = double4_t{0.0, 0.0, 0.0, 0.0};
                }
                double an[PCL_MAXRT], bn = 0.0;
#pragma unroll
                for (int t = 0; t < PCL_MAXRT; ++t) an[t] = kfull > 0 ? Ap[t][0] : 0.0;
                if (kfull > 0) bn = Bp[0];
                for (int ks = 0; ks < kfull; ++ks) {
                    double a[PCL_MAXRT];
                    const double bb = bn;
#pragma unroll
                    for (int t = 0; t < PCL_MAXRT; ++t) a[t] = an[t];
                    if (ks + 1 < kfull) {
#pragma unroll
                        for (int t = 0; t < PCL_MAXRT; ++t) an[t] = Ap[t][4 * (ks + 1)];
                        bn = Bp[4 * (ks + 1)];
                    }
#pragma unroll
                    for (int t = 0; t < PCL_MAXRT; ++t)
                        if (rok[t]) ac[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[t], bb, ac[t], 0, 0, 0);
                }
                if (krem) {
                    const bool ok = lk < krem;
                    const double bb = ok ? Bp[4 * kfull] : 0.0;
#pragma unroll
                    for (int t = 0; t < PCL_MAXRT; ++t)
                        if (rok[t]) {
                            const double a = ok ? Ap[t][4 * kfull] : 0.0;
                            ac[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bb, ac[t], 0, 0, 0);
                        }
                }
            }
            wave_lds_sync();  // every operand read of this wave is complete before the tile is overwritten
            PCL_HSTAMP();  // MFMA done
            if (li < (TM + 1) * NCW) {
                double *a1 = (li < ncc) ? A1s + LD * (cl0 + li) : nullptr;
#pragma unroll
                for (int t = 0; t < PCL_MAXRT; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int row = t * 16 + lk + 4 * r;
                        if (row < n) {
                            Mw[row + LD * li] = ac[t][r];
                            if (a1) a1[row] = ac[t][r];
                        }
                    }
            }
            wave_lds_sync();
            if (lane < n) {
#pragma unroll
                for (int c = 0; c < NCW; ++c)
                    if (c < ncc) {
                        const double a1v = Mw[lane + LD * c];
                        const double *Dc = Ds + LD * (cl0 + c);
                        const double dv = Dc[lane];
                        const long long o = (long long)(c0 + cl0 + c) * n + lane;
                        double Ev[TM];  // E_l = G_l D, this row and column
#pragma unroll
                        for (int l = 0; l < TM; ++l) {
                            double ev = 0.0, r = 0.0;  // R_l = G_l^T A1
#pragma unroll
                            for (int q = 0; q < EW; ++q) {
                                ev += er_v[l][q] * Dc[er_c[l][q]];
                                r += ET_V(l, q) * Mw[ET_C(l, q) + LD * c];
                            }
                            if (ANTI) r = -r;
                            Ev[l] = ev;
                            const double qv = Mw[lane + LD * (NCW + l * NCW + c)];
                            const double kt = c2 * (qv + r), pl = -c1 * Pv[l][c];
                            if (!(p.ablate & 1)) {
                                H3[(long long)l * xd + o] = pl - kt;
                                H5[(long long)l * xd + o] = pl + kt;
                            }
                            acc[NPAIR + l] += -0.5 * Pv[l][c] * Sv[c] + h6 * (qv * dv + a1v * ev);
                        }
                        int e = 0;
                        if (!(p.ablate & 2)) {
#pragma unroll
                            for (int i = 0; i < TM; ++i)
#pragma unroll
                                for (int j = 0; j <= i; ++j, ++e) acc[e] += Pv[i][c] * Ev[j] + Pv[j][c] * Ev[i];
                        }
                    }
            }
            wave_lds_sync();  // Mw is rewritten by this wave's next chunk / the reduction below
            PCL_HSTAMP();  // chunk outputs + sums done
        }
        // ---- lane -> wave reduction of the per-lane partial sums on the matrix cores (fixed order) -----------------
        // C += 1_e x v_e : with a = [li == e] and b = the lanes' partial sums of entry e, row e of C collects
        // sum_k v_e[lane j + 16 k] in column j; four DPP steps then add the 16 columns of a row.
        {
            double4_t r0 = {0.0, 0.0, 0.0, 0.0}, r1 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int e = 0; e < NACC; ++e) {
                const double ind = (li == (e & 15)) ? 1.0 : 0.0;
                if (e < 16)
                    r0 = __builtin_amdgcn_mfma_f64_16x16x4f64(ind, acc[e], r0, 0, 0, 0);
                else
                    r1 = __builtin_amdgcn_mfma_f64_16x16x4f64(ind, acc[e], r1, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const double t0 = row16_sum(r0[r]);
                const int e0 = lk + 4 * r;
                if (li == 0 && e0 < NACC) wsum[wave * NSC + e0] = t0;
                if (NACC > 16) {
                    const double t1 = row16_sum(r1[r]);
                    if (li == 0 && 16 + e0 < NACC) wsum[wave * NSC + 16 + e0] = t1;
                }
            }
        }
        PCL_HSTAMP();  // wave reduction done
        __syncthreads();  // A1s, Ds and wsum complete
        PCL_HSTAMP();
        // ---- A2 = G^T A1 for the slice's columns: wave w = row tile w --------------------------------------------
        if (wave * 16 < n) {
            const double *Ap = G + lk + LD * (wave * 16 + li);
            const double *Bp = A1s + lk + LD * li;
            double4_t a2 = {0.0, 0.0, 0.0, 0.0};
            for (int ks = 0; ks < kfull; ++ks) a2 = __builtin_amdgcn_mfma_f64_16x16x4f64(Ap[4 * ks], Bp[4 * ks], a2, 0, 0, 0);
            if (krem) {
                const bool ok = lk < krem;
                a2 = __builtin_amdgcn_mfma_f64_16x16x4f64(ok ? Ap[4 * kfull] : 0.0, ok ? Bp[4 * kfull] : 0.0, a2, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = wave * 16 + lk + 4 * r;
                if (row < n) A2s[row + LD * li] = a2[r];
            }
        }
        __syncthreads();
        {
            double v = 0.0;
            for (int e = tid; e < nce * n; e += 256) {
                const int c = e / n, i = e - c * n;
                v += A2s[i + LD * c] * Ds[i + LD * c];
            }
            v = wave_sum(v);
            if (lane == 0) wsum2[wave] = v;
        }
        __syncthreads();
        PCL_HSTAMP();  // A2 done
        if (tid < NSC) {
            double tot;
            if (tid < NACC) {
                tot = ((wsum[tid] + wsum[NSC + tid]) + wsum[2 * NSC + tid]) + wsum[3 * NSC + tid];
                if (tid < NPAIR) tot *= c2;
            } else {
                tot = (((wsum2[0] + wsum2[1]) + wsum2[2]) + wsum2[3]) * (1.0 / 6.0);
            }
            if (S == 1)
                H[tid] = tot;
            else {
                // agent-scope (write-through) store of this slice's partial entry; it has left the CU before the arrival
                // counter moves.  No release fence: that would write back this XCD's whole L2 (full of Hessian output).
                __hip_atomic_store(p.hpart + (bk * S + s) * NSC + tid, tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
        }
        unsigned int ticket = 0;
        if (S > 1) {
            __syncthreads();
            if (tid == 0) ticket = __hip_atomic_fetch_add(p.hcnt + bk, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        // the d2/dh dX vectors go out while the counter's round trip is in flight
        for (int e = tid; e < nce * n; e += 256) {
            const int c = e / n, i = e - c * n;
            const double a1 = A1s[i + LD * c], a2 = A2s[i + LD * c];
            const long long o = (long long)(c0 + c) * n + i;
            H4[o] = -0.5 * a1 - h6 * a2;
            H6[o] = -0.5 * a1 + h6 * a2;
        }
        if (S > 1) {
            if (tid == 0) {
                const int last = ticket == (unsigned int)(S - 1);
                if (last) __hip_atomic_store(p.hcnt + bk, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // ready for the next launch
                *lastflag = last;
            }
            __syncthreads();
            if (*lastflag && tid < NSC) {  // the last slice of the interval to arrive sums the partials in slice order
                double t = 0.0;  // agent-scope loads bypass this XCD's L2
                for (int q = 0; q < S; ++q) t += __hip_atomic_load(p.hpart + (bk * S + q) * NSC + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                H[tid] = t;
            }
        }
        PCL_HSTAMP();  // item done
    }
}

// ------------------------------------------------------------------------------------------
// Rollout: exact piecewise-constant propagation  X_{k+1} = exp(dt_k G(u_k)) X_k  from the knot-0 state
// -- what the reference's unitary_rollout(...; interpolation = :constant) integrates with an ODE solver
// [REF src/quantum/dynamics.jl:631-667] and the slot its RolloutStates reserves for "a GPU rollout"
// [REF src/quantum/trajectories/ensemble_trajectory.jl:56-71].
//   pcl_expm_kernel   one workgroup per (b, k): E = exp(h G) by scaling and squaring, Taylor degree 14 (Horner) at
//                     |h| ||G||_1 / 2^s <= 1/4 (truncation < 1e-21), products on the matrix cores
//   pcl_chain_kernel  one workgroup per member / trajectory: the K dependent n x n x cols products
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pcl_expm_kernel(const KParams p) {
    extern __shared__ double lds[];
    const int n = p.n, LD = p.LD;
    const int tid = threadIdx.x, nth = blockDim.x;
    const int k = blockIdx.x % p.K, b = blockIdx.x / p.K;
    double *A = lds, *T = A + LD * n, *T2 = T + LD * n, *us = T2 + LD * n, *red = us + 8 + p.m;
    const double *zk = p.Z + (long long)b * p.z_batch_stride + (long long)k * p.z_dim;
    const double h = zk[p.dt_off];
    build_G(p, p.G0 + (long long)b * p.g0_batch_stride, zk, A, us);
    __syncthreads();
    if (tid < 64) {
        double cs = 0.0;
        if (tid < n)
            for (int i = 0; i < n; ++i) cs += fabs(A[i + LD * tid]);
        red[tid] = cs;
    }
    __syncthreads();
    double nrm = 0.0;
    for (int j = 0; j < n; ++j) nrm = fmax(nrm, red[j]);
    double theta = fabs(h) * nrm;
    int sq = 0;
    while (theta > 0.25 && sq < 60) {
        theta *= 0.5;
        ++sq;
    }
    const double hs = ldexp(h, -sq);
    for (int e = tid; e < n * n; e += nth) T[(e % n) + LD * (e / n)] = (e % n == e / n) ? 1.0 : 0.0;
    __syncthreads();
    for (int j = 14; j >= 1; --j) {  // T <- I + (hs/j) A T
        gemm_lds<true, false>(A, LD, T, LD, T2, LD, n, n, n);
        __syncthreads();
        const double f = hs / j;
        for (int e = tid; e < n * n; e += nth) {
            const int idx = (e % n) + LD * (e / n);
            T[idx] = ((e % n == e / n) ? 1.0 : 0.0) + f * T2[idx];
        }
        __syncthreads();
    }
    double *cur = T, *oth = T2;
    for (int i = 0; i < sq; ++i) {
        gemm_lds<true, false>(cur, LD, cur, LD, oth, LD, n, n, n);
        __syncthreads();
        double *t = cur;
        cur = oth;
        oth = t;
    }
    double *E = p.expm + ((long long)b * p.K + k) * n * n;
    for (int e = tid; e < n * n; e += nth) E[e] = cur[(e % n) + LD * (e / n)];
}

__global__ __launch_bounds__(256) void pcl_chain_kernel(const KParams p) {
    extern __shared__ double lds[];
    const int n = p.n, LD = p.LD, cols = p.cols;
    const int tid = threadIdx.x, nth = blockDim.x;
    const int b = blockIdx.x;
    const long long xd = (long long)n * cols;
    double *E = lds, *Xa = E + LD * n, *Xb = Xa + LD * cols;
    const double *z0 = p.Z + (long long)b * p.z_batch_stride;
    const int x_off = p.x_offs[p.z_batch_stride ? 0 : b];
    double *out = p.xout + (long long)b * (p.K + 1) * xd;
    for (int e = tid; e < xd; e += nth) {
        const double v = z0[x_off + e];
        Xa[(e % n) + LD * (e / n)] = v;
        out[e] = v;
    }
    double *cur = Xa, *oth = Xb;
    for (int k = 0; k < p.K; ++k) {
        const double *Ek = p.expm + ((long long)b * p.K + k) * n * n;
        __syncthreads();  // previous product complete (E and `oth` free)
        for (int e = tid; e < n * n; e += nth) E[(e % n) + LD * (e / n)] = Ek[e];
        __syncthreads();
        gemm_lds<true, false>(E, LD, cur, LD, oth, LD, n, cols, n);
        __syncthreads();
        for (int e = tid; e < xd; e += nth) out[(long long)(k + 1) * xd + e] = oth[(e % n) + LD * (e / n)];
        double *t = cur;
        cur = oth;
        oth = t;
    }
}

// ------------------------------------------------------------------------------------------
// DerivativeIntegrator rows  x_{k+1} - x_k - dt_k * dx_k  and the time-consistency row  t_{k+1} - t_k - dt_k
// (dx_off < 0: dx == 1).  Trivially sparse; one thread per (b, k, r).  Values per (b,k): [-1 (dim) | +1 (dim) |
// -dt_k (dim, absent for time consistency) | -dx_k[r] (dim)].
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pcl_deriv_kernel(const double *__restrict__ Z, double *__restrict__ delta,
                                                        double *__restrict__ vals, int K, int z_dim, int x_off, int dx_off,
                                                        int dim, int dt_off, long long z_batch_stride, long long total) {
    const int nseg = dx_off >= 0 ? 4 : 3;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
        const int r = (int)(e % dim);
        const long long bk = e / dim;
        const int k = (int)(bk % K);
        const long long b = bk / K;
        const double *zk = Z + b * z_batch_stride + (long long)k * z_dim;
        const double h = zk[dt_off];
        const double dx = dx_off >= 0 ? zk[dx_off + r] : 1.0;
        if (delta) delta[e] = zk[z_dim + x_off + r] - zk[x_off + r] - h * dx;
        if (vals) {
            double *v = vals + bk * (long long)nseg * dim;
            v[r] = -1.0;
            v[dim + r] = 1.0;
            if (dx_off >= 0) {
                v[2 * dim + r] = -h;
                v[3 * dim + r] = -dx;
            } else {
                v[2 * dim + r] = -1.0;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// Terminal unitary infidelity  Q * |1 - |tr(U_goal' U_N)|^2 / d^2|  and its gradient w.r.t. the terminal iso-vec
// (reference: src/control/objectives.jl:330-356).  One workgroup per member / seed.
// With X = [Re U; Im U] (n x d, column c at x[c*n ..]) and the goal stored the same way:
//   t = tr(Ug' U) = sum (gr*ur + gi*ui) + i sum (gr*ui - gi*ur);  F = |t|^2 / d^2
//   dF/dur = 2 (tr*gr - ti*gi) / d^2 ,  dF/dui = 2 (tr*gi + ti*gr) / d^2
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pcl_infidelity_kernel(const double *__restrict__ Z, const double *__restrict__ goal,
                                                             const int *__restrict__ x_offs, double *__restrict__ value,
                                                             double *__restrict__ grad, double Q, int d, int N, int z_dim,
                                                             long long z_batch_stride) {
    __shared__ double red[2][8];
    const int n = 2 * d, b = blockIdx.x, tid = threadIdx.x;
    const double *x = Z + (long long)b * z_batch_stride + (long long)(N - 1) * z_dim + x_offs[z_batch_stride ? 0 : b];
    double tr = 0.0, ti = 0.0;
    for (int e = tid; e < d * d; e += 256) {
        const int c = e / d, i = e - c * d;
        const double ur = x[c * n + i], ui = x[c * n + d + i], gr = goal[c * n + i], gi = goal[c * n + d + i];
        tr += gr * ur + gi * ui;
        ti += gr * ui - gi * ur;
    }
    for (int off = 32; off > 0; off >>= 1) {
        tr += __shfl_down(tr, off, 64);
        ti += __shfl_down(ti, off, 64);
    }
    if ((tid & 63) == 0) {
        red[0][tid >> 6] = tr;
        red[1][tid >> 6] = ti;
    }
    __syncthreads();
    tr = red[0][0] + red[0][1] + red[0][2] + red[0][3];
    ti = red[1][0] + red[1][1] + red[1][2] + red[1][3];
    const double inv = 1.0 / ((double)d * d);
    const double F = (tr * tr + ti * ti) * inv;
    const double sgn = (1.0 - F >= 0.0) ? 1.0 : -1.0;
    if (tid == 0 && value) value[b] = Q * fabs(1.0 - F);
    if (grad) {
        double *g = grad + (long long)b * n * d;
        for (int e = tid; e < d * d; e += 256) {
            const int c = e / d, i = e - c * d;
            const double gr = goal[c * n + i], gi = goal[c * n + d + i];
            g[c * n + i] = -sgn * Q * 2.0 * (tr * gr - ti * gi) * inv;
            g[c * n + d + i] = -sgn * Q * 2.0 * (tr * gi + ti * gr) * inv;
        }
    }
}

// ------------------------------------------------------------------------------------------
// Host side
// ------------------------------------------------------------------------------------------
static thread_local std::string g_create_error;

struct pcl_ctx {
    pcl_desc desc;
    int n, K;
    int cols;  // state columns (d for unitaries, 1 for kets)
    long long x_dim;
    std::vector<int32_t> x_offs;
    int device;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    // device copies
    double *dG0 = nullptr, *ducoef = nullptr, *dcsr_val = nullptr, *dcsc_val = nullptr;
    int *dupos = nullptr, *dcsr_ptr = nullptr, *dcsr_col = nullptr, *dcsc_ptr = nullptr, *dcsc_row = nullptr, *dxoffs = nullptr;
    int n_upos = 0;
    int *dumap = nullptr, *dell_col = nullptr;
    double *dell_val = nullptr;
    int ell_w = 0, iso = 0, uell_w = 0;
    unsigned char *duell_l = nullptr;
    double *duell_v = nullptr;
    int *dellt_col = nullptr;  // ELL form of G_l^T (Hessian kernel v2)
    double *dellt_val = nullptr;
    int ellt_w = 0;
    int drives_antisym = 0;  // every G_l == -G_l^T exactly
    double *dug0 = nullptr;
    double *dexpm = nullptr, *dxout = nullptr;  // rollout scratch: propagators, staged output of the host-pointer call
    double *dhpart = nullptr;  // Hessian v2 scratch: per (b,k,slice) partial scalar entries + per (b,k) arrival counters
    unsigned int *dhcnt = nullptr;
    long long hpart_cap = 0;
    int64_t opt_hess_kernel = 0, last_hess_kernel = 0;  // 0 = auto
    // staging for the host-pointer entry points
    double *dZ = nullptr, *dmu = nullptr, *ddelta = nullptr, *dvals = nullptr, *dhess = nullptr;
    // options
    int64_t opt_cols_per_slice = 0, opt_use_mfma = 1, opt_nt = 0, opt_ablate = 0, opt_kernel = 0;  // 0 = auto: 3 when it applies, else 4 / 2 by work per workgroup
    long long *ddbg = nullptr;
    void *comm = nullptr;  // ncclComm_t
    double *dgoal = nullptr;  // iso-vec of the goal unitary (pcl_set_goal)
    // split mode (producer kernel + concurrent expander kernel)
    double *dblocks = nullptr;
    unsigned int *dflags = nullptr;
    hipStream_t aux_stream = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    int64_t opt_cpp = 6;  // block copies per expander work item
    int64_t opt_specialize = 1;
    int64_t opt_snc = 0;      // v3, role split: stream pieces of this many columns dealt round-robin (0: contiguous ranges)
    int64_t opt_flat = 0;     // v3: line-aligned flat block stream (measured: no gain over the per-block stores, slower for one trajectory)
    int64_t opt_general = 0;  // 1: run the general-order kernel also for pade_order 4 (cross-check)
    int64_t opt_contig = -1;     // v3: contiguous column ranges per workgroup (-1: auto by launch size)
    int64_t opt_stream_wg = -1;  // v3, contiguous: stream-role workgroups (-1: auto = half, 0: every workgroup does both)
    int64_t last_n_stream = 0;  // stream-role workgroups of the last kernel-3 launch (0: fused roles / round-robin)
    int64_t last_kernel = 0;  // 10*version + (1 if shape-specialised) of the last fused launch
    int64_t opt_grid = 0;  // 0: resident workgroups (persistent kernel)
    size_t lds_set[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const void *lds_kern[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};  // last MaxDynamicSharedMemorySize set per kernel variant
    int max_lds = 0;
    int n_cu = 0;
    mutable std::string err;
};

static int fail(const pcl_ctx *ctx, int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (ctx)
        ctx->err = buf;
    else
        g_create_error = buf;
    return code;
}

#define HIP_TRY(ctx, expr)                                                                      \
    do {                                                                                        \
        hipError_t e_ = (expr);                                                                 \
        if (e_ != hipSuccess) return fail(ctx, PCL_EHIP, "%s: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

static long long jac_per_full(const pcl_ctx *c) {
    return 2LL * c->cols * c->n * c->n + c->x_dim * (c->desc.n_drives + 1);
}
static long long jac_per_compact(const pcl_ctx *c) { return 2LL * c->n * c->n + c->x_dim * (c->desc.n_drives + 1); }
static long long hess_per(const pcl_ctx *c) {
    const long long m = c->desc.n_drives;
    return (m + 1) * (m + 2) / 2 + 2 * c->x_dim * (m + 1);
}
static long long z_len(const pcl_ctx *c) {
    return (long long)c->desc.z_dim * c->desc.N * (c->desc.batch_mode == PCL_BATCH_TRAJ ? c->desc.batch : 1);
}
static long long n_rows(const pcl_ctx *c) { return (long long)c->desc.batch * c->x_dim * c->K; }

extern "C" const char *pcl_version(void) { return PCL_VERSION_STR; }

extern "C" const char *pcl_last_error(const pcl_ctx *ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

template <class T>
static int upload(pcl_ctx *ctx, T **dst, const std::vector<T> &src) {
    const size_t bytes = std::max<size_t>(src.size(), 1) * sizeof(T);
    HIP_TRY(ctx, hipMalloc((void **)dst, bytes));
    if (!src.empty()) HIP_TRY(ctx, hipMemcpy(*dst, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice));
    return PCL_OK;
}

extern "C" int pcl_create(const pcl_desc *dsc, pcl_ctx **out) {
    if (!out) return fail(nullptr, PCL_EINVAL, "pcl_create: out is NULL");
    *out = nullptr;
    if (!dsc) return fail(nullptr, PCL_EINVAL, "pcl_create: desc is NULL");
    if (dsc->struct_size != (int32_t)sizeof(pcl_desc))
        return fail(nullptr, PCL_EINVAL, "pcl_create: desc.struct_size=%d, library expects %zu (ABI mismatch)",
                    dsc->struct_size, sizeof(pcl_desc));
    const int d = dsc->d, m = dsc->n_drives, n = 2 * d;
    if (d < 1 || m < 0 || dsc->N < 2 || dsc->batch < 1)
        return fail(nullptr, PCL_EINVAL, "pcl_create: need d>=1, n_drives>=0, N>=2, batch>=1 (got d=%d m=%d N=%d batch=%d)", d,
                    m, dsc->N, dsc->batch);
    if (d > PCL_MAX_D) return fail(nullptr, PCL_ESHAPE, "pcl_create: d=%d exceeds PCL_MAX_D=%d (LDS-resident tiles)", d, PCL_MAX_D);
    if (m > 24) return fail(nullptr, PCL_ESHAPE, "pcl_create: n_drives=%d exceeds 24", m);
    if (dsc->pade_order != 2 && dsc->pade_order != 4 && dsc->pade_order != 6 && dsc->pade_order != 8 && dsc->pade_order != 10)
        return fail(nullptr, PCL_ENOTIMPL, "pcl_create: pade_order=%d; diagonal Pade orders 2, 4, 6, 8, 10 are implemented", dsc->pade_order);
    if (dsc->index_base != 0 && dsc->index_base != 1) return fail(nullptr, PCL_EINVAL, "pcl_create: index_base must be 0 or 1");
    if (dsc->batch_mode != PCL_BATCH_MEMBERS && dsc->batch_mode != PCL_BATCH_TRAJ)
        return fail(nullptr, PCL_EINVAL, "pcl_create: unknown batch_mode %d", dsc->batch_mode);
    if (!dsc->G0 || (m > 0 && !dsc->Gj) || !dsc->x_offs) return fail(nullptr, PCL_EINVAL, "pcl_create: G0/Gj/x_offs must be non-NULL");
    const int cols = dsc->state_cols > 0 ? dsc->state_cols : d;
    if (cols > d) return fail(nullptr, PCL_EINVAL, "pcl_create: state_cols=%d exceeds d=%d", cols, d);
    const long long x_dim = 2LL * d * cols;
    const int n_off = dsc->batch_mode == PCL_BATCH_MEMBERS ? dsc->batch : 1;
    for (int i = 0; i < n_off; ++i)
        if (dsc->x_offs[i] < 0 || dsc->x_offs[i] + x_dim > dsc->z_dim)
            return fail(nullptr, PCL_EINVAL, "pcl_create: x_offs[%d]=%d with x_dim=%lld does not fit z_dim=%d", i, dsc->x_offs[i],
                        x_dim, dsc->z_dim);
    if (dsc->u_off < 0 || dsc->u_off + m > dsc->z_dim || dsc->dt_off < 0 || dsc->dt_off >= dsc->z_dim)
        return fail(nullptr, PCL_EINVAL, "pcl_create: u_off/dt_off outside the knot (z_dim=%d)", dsc->z_dim);
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0)
        return fail(nullptr, PCL_EHIP, "pcl_create: no HIP device available (%s); this library has no CPU path",
                    e == hipSuccess ? "device count 0" : hipGetErrorString(e));
    if (dsc->device_id < 0 || dsc->device_id >= ndev) return fail(nullptr, PCL_EINVAL, "pcl_create: device_id %d of %d", dsc->device_id, ndev);

    pcl_ctx *ctx = new (std::nothrow) pcl_ctx();
    if (!ctx) return fail(nullptr, PCL_ENOMEM, "pcl_create: out of host memory");
    ctx->desc = *dsc;
    ctx->n = n;
    ctx->K = dsc->N - 1;
    ctx->x_dim = x_dim;
    ctx->cols = cols;
    ctx->x_offs.assign(dsc->x_offs, dsc->x_offs + n_off);
    ctx->desc.x_offs = nullptr;
    ctx->desc.G0 = ctx->desc.Gj = nullptr;
    ctx->device = dsc->device_id;

#define CREATE_TRY(expr)                                                  \
    do {                                                                  \
        int rc_ = (expr);                                                 \
        if (rc_ != PCL_OK) {                                              \
            g_create_error = ctx->err;                                    \
            pcl_destroy(ctx);                                             \
            return rc_;                                                   \
        }                                                                 \
    } while (0)
#define CREATE_HIP(expr)                                                                               \
    do {                                                                                               \
        hipError_t e2_ = (expr);                                                                       \
        if (e2_ != hipSuccess) {                                                                       \
            fail(nullptr, PCL_EHIP, "pcl_create: %s: %s", #expr, hipGetErrorString(e2_));              \
            pcl_destroy(ctx);                                                                          \
            return PCL_EHIP;                                                                           \
        }                                                                                              \
    } while (0)

    CREATE_HIP(hipSetDevice(ctx->device));
    hipDeviceProp_t prop;
    CREATE_HIP(hipGetDeviceProperties(&prop, ctx->device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        fail(nullptr, PCL_EHIP, "pcl_create: device %d is %s; this library is built for gfx950 only", ctx->device, prop.gcnArchName);
        pcl_destroy(ctx);
        return PCL_EHIP;
    }
    ctx->max_lds = (int)prop.maxSharedMemoryPerMultiProcessor;
    ctx->n_cu = prop.multiProcessorCount;
    CREATE_HIP(hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking));
    ctx->stream = ctx->own_stream;

    // --- drive structures: union pattern (+ coefficient table), CSR and CSC of every G_l ------
    const size_t nn = (size_t)n * n;
    std::vector<int> upos;
    std::vector<double> ucoef;
    for (size_t pz = 0; pz < nn; ++pz) {
        bool any = false;
        for (int l = 0; l < m; ++l) any |= dsc->Gj[l * nn + pz] != 0.0;
        if (any) {
            upos.push_back((int)pz);
            for (int l = 0; l < m; ++l) ucoef.push_back(dsc->Gj[l * nn + pz]);
        }
    }
    ctx->n_upos = (int)upos.size();
    std::vector<int> csr_ptr((size_t)std::max(m, 1) * (n + 1), 0), csr_col, csc_ptr((size_t)std::max(m, 1) * (n + 1), 0), csc_row;
    std::vector<double> csr_val, csc_val;
    for (int l = 0; l < m; ++l) {
        const double *A = dsc->Gj + l * nn;  // column-major: A[i + n*j]
        for (int i = 0; i < n; ++i) {
            csr_ptr[(size_t)l * (n + 1) + i] = (int)csr_col.size();
            for (int j = 0; j < n; ++j)
                if (A[i + (size_t)n * j] != 0.0) {
                    csr_col.push_back(j);
                    csr_val.push_back(A[i + (size_t)n * j]);
                }
        }
        csr_ptr[(size_t)l * (n + 1) + n] = (int)csr_col.size();
        for (int j = 0; j < n; ++j) {
            csc_ptr[(size_t)l * (n + 1) + j] = (int)csc_row.size();
            for (int i = 0; i < n; ++i)
                if (A[i + (size_t)n * j] != 0.0) {
                    csc_row.push_back(i);
                    csc_val.push_back(A[i + (size_t)n * j]);
                }
        }
        csc_ptr[(size_t)l * (n + 1) + n] = (int)csc_row.size();
    }
    int uell_w = 0;
    for (size_t q = 0; q < upos.size(); ++q) {
        int cnt = 0;
        for (int l = 0; l < m; ++l) cnt += ucoef[q * m + l] != 0.0;
        uell_w = std::max(uell_w, cnt);
    }
    uell_w = std::max(uell_w, 1);
    std::vector<unsigned char> uell_l(std::max<size_t>(upos.size(), 1) * uell_w, 0);
    std::vector<double> uell_v(std::max<size_t>(upos.size(), 1) * uell_w, 0.0);
    for (size_t q = 0; q < upos.size(); ++q) {
        int cnt = 0;
        for (int l = 0; l < m; ++l)
            if (ucoef[q * m + l] != 0.0) {
                uell_l[q * uell_w + cnt] = (unsigned char)l;
                uell_v[q * uell_w + cnt] = ucoef[q * m + l];
                ++cnt;
            }
    }
    ctx->uell_w = uell_w;
    std::vector<int> umap(nn, -1);
    for (size_t q = 0; q < upos.size(); ++q) umap[upos[q]] = (int)q;
    int ell_w = m > 0 ? 1 : 0;
    for (int l = 0; l < m; ++l)
        for (int i = 0; i < n; ++i) ell_w = std::max(ell_w, csr_ptr[(size_t)l * (n + 1) + i + 1] - csr_ptr[(size_t)l * (n + 1) + i]);
    std::vector<int> ell_col((size_t)m * n * ell_w, 0);
    std::vector<double> ell_val((size_t)m * n * ell_w, 0.0);
    for (int l = 0; l < m; ++l)
        for (int i = 0; i < n; ++i) {
            const int beg = csr_ptr[(size_t)l * (n + 1) + i], end = csr_ptr[(size_t)l * (n + 1) + i + 1];
            for (int q = beg; q < end; ++q) {
                ell_col[((size_t)l * n + i) * ell_w + (q - beg)] = csr_col[q];
                ell_val[((size_t)l * n + i) * ell_w + (q - beg)] = csr_val[q];
            }
        }
    ctx->ell_w = ell_w;
    int ellt_w = m > 0 ? 1 : 0;
    for (int l = 0; l < m; ++l)
        for (int i = 0; i < n; ++i) ellt_w = std::max(ellt_w, csc_ptr[(size_t)l * (n + 1) + i + 1] - csc_ptr[(size_t)l * (n + 1) + i]);
    std::vector<int> ellt_col((size_t)m * n * ellt_w, 0);
    std::vector<double> ellt_val((size_t)m * n * ellt_w, 0.0);
    for (int l = 0; l < m; ++l)
        for (int i = 0; i < n; ++i) {
            const int beg = csc_ptr[(size_t)l * (n + 1) + i], end = csc_ptr[(size_t)l * (n + 1) + i + 1];
            for (int q = beg; q < end; ++q) {
                ellt_col[((size_t)l * n + i) * ellt_w + (q - beg)] = csc_row[q];
                ellt_val[((size_t)l * n + i) * ellt_w + (q - beg)] = csc_val[q];
            }
        }
    ctx->ellt_w = ellt_w;
    {
        bool anti = true;
        for (int l = 0; l < m && anti; ++l)
            for (int j = 0; j < n && anti; ++j)
                for (int i = 0; i < n; ++i)
                    if (dsc->Gj[(size_t)l * nn + i + (size_t)n * j] != -dsc->Gj[(size_t)l * nn + j + (size_t)n * i]) {
                        anti = false;
                        break;
                    }
        ctx->drives_antisym = anti ? 1 : 0;
    }
    // exact iso structure  M = [[A, -B], [B, A]]  of the drift(s) and of every drive?
    auto is_iso = [&](const double *A) {
        for (int j = 0; j < d; ++j)
            for (int i = 0; i < d; ++i) {
                if (A[(i + d) + (size_t)n * (j + d)] != A[i + (size_t)n * j]) return false;
                if (A[i + (size_t)n * (j + d)] != -A[(i + d) + (size_t)n * j]) return false;
            }
        return true;
    };
    bool iso = true;
    for (int bb = 0; bb < (dsc->per_member_G0 ? dsc->batch : 1); ++bb) iso = iso && is_iso(dsc->G0 + bb * nn);
    for (int l = 0; l < m; ++l) iso = iso && is_iso(dsc->Gj + l * nn);
    ctx->iso = iso ? 1 : 0;
    std::vector<double> g0(dsc->G0, dsc->G0 + nn * (dsc->per_member_G0 ? dsc->batch : 1));
    CREATE_TRY(upload(ctx, &ctx->dG0, g0));
    CREATE_TRY(upload(ctx, &ctx->dupos, upos));
    CREATE_TRY(upload(ctx, &ctx->ducoef, ucoef));
    CREATE_TRY(upload(ctx, &ctx->dcsr_ptr, csr_ptr));
    CREATE_TRY(upload(ctx, &ctx->dcsr_col, csr_col));
    CREATE_TRY(upload(ctx, &ctx->dcsr_val, csr_val));
    CREATE_TRY(upload(ctx, &ctx->dcsc_ptr, csc_ptr));
    CREATE_TRY(upload(ctx, &ctx->dcsc_row, csc_row));
    CREATE_TRY(upload(ctx, &ctx->dcsc_val, csc_val));
    CREATE_TRY(upload(ctx, &ctx->dumap, umap));
    CREATE_TRY(upload(ctx, &ctx->duell_l, uell_l));
    CREATE_TRY(upload(ctx, &ctx->duell_v, uell_v));
    CREATE_TRY(upload(ctx, &ctx->dell_col, ell_col));
    CREATE_TRY(upload(ctx, &ctx->dell_val, ell_val));
    CREATE_TRY(upload(ctx, &ctx->dellt_col, ellt_col));
    CREATE_TRY(upload(ctx, &ctx->dellt_val, ellt_val));
    {
        std::vector<double> ug0(std::max<size_t>(upos.size(), 1), 0.0);
        for (size_t q = 0; q < upos.size(); ++q) ug0[q] = dsc->G0[upos[q]];
        CREATE_TRY(upload(ctx, &ctx->dug0, ug0));
    }
    std::vector<int> xo(ctx->x_offs.begin(), ctx->x_offs.end());
    CREATE_TRY(upload(ctx, &ctx->dxoffs, xo));
#undef CREATE_TRY
#undef CREATE_HIP
    *out = ctx;
    return PCL_OK;
}

extern "C" int pcl_comm_destroy(pcl_ctx *ctx);
extern "C" void pcl_destroy(pcl_ctx *ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)pcl_comm_destroy(ctx);
    void *ptrs[] = {ctx->dG0, ctx->ducoef, ctx->dcsr_val, ctx->dcsc_val, ctx->dupos, ctx->dcsr_ptr, ctx->dcsr_col,
                    ctx->dcsc_ptr, ctx->dcsc_row, ctx->dxoffs, ctx->dZ, ctx->dmu, ctx->ddelta, ctx->dvals, ctx->dhess,
                    ctx->dumap, ctx->dell_col, ctx->dell_val, ctx->duell_l, ctx->duell_v, ctx->ddbg, ctx->dellt_col, ctx->dellt_val,
                    ctx->dhpart, ctx->dhcnt, ctx->dug0, ctx->dexpm, ctx->dxout};
    for (void *q : ptrs)
        if (q) (void)hipFree(q);
    if (ctx->dgoal) (void)hipFree(ctx->dgoal);
    if (ctx->dblocks) (void)hipFree(ctx->dblocks);
    if (ctx->dflags) (void)hipFree(ctx->dflags);
    if (ctx->ev_fork) (void)hipEventDestroy(ctx->ev_fork);
    if (ctx->ev_join) (void)hipEventDestroy(ctx->ev_join);
    if (ctx->aux_stream) (void)hipStreamDestroy(ctx->aux_stream);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
    delete ctx;
}

extern "C" int pcl_constraint_dim(const pcl_ctx *ctx, int64_t *x_dim, int64_t *rows, int64_t *cols) {
    if (!ctx) return PCL_EINVAL;
    if (x_dim) *x_dim = ctx->x_dim;
    if (rows) *rows = n_rows(ctx);
    if (cols) *cols = z_len(ctx) + ctx->desc.global_dim;
    return PCL_OK;
}
extern "C" int pcl_jac_nnz(const pcl_ctx *ctx, int64_t *nnz, int64_t *per) {
    if (!ctx) return PCL_EINVAL;
    if (per) *per = jac_per_full(ctx);
    if (nnz) *nnz = jac_per_full(ctx) * ctx->desc.batch * ctx->K;
    return PCL_OK;
}
extern "C" int pcl_jac_compact_nnz(const pcl_ctx *ctx, int64_t *nnz, int64_t *per) {
    if (!ctx) return PCL_EINVAL;
    if (per) *per = jac_per_compact(ctx);
    if (nnz) *nnz = jac_per_compact(ctx) * ctx->desc.batch * ctx->K;
    return PCL_OK;
}
extern "C" int pcl_hess_nnz(const pcl_ctx *ctx, int64_t *nnz, int64_t *per) {
    if (!ctx) return PCL_EINVAL;
    if (per) *per = hess_per(ctx);
    if (nnz) *nnz = hess_per(ctx) * ctx->desc.batch * ctx->K;
    return PCL_OK;
}

template <class I>
static int jac_structure_impl(const pcl_ctx *ctx, I *rows, I *cols) {
    if (!ctx) return PCL_EINVAL;
    if (!rows || !cols) return fail(ctx, PCL_EINVAL, "pcl_jac_structure: NULL output");
    const pcl_desc &D = ctx->desc;
    const long long n = ctx->n, d = ctx->cols, m = D.n_drives, xd = ctx->x_dim, zd = D.z_dim, base = D.index_base;
    const long long per = jac_per_full(ctx);
    for (long long b = 0; b < D.batch; ++b) {
        const long long xo = ctx->x_offs[D.batch_mode == PCL_BATCH_MEMBERS ? b : 0];
        const long long voff = D.batch_mode == PCL_BATCH_TRAJ ? b * zd * D.N : 0;
        for (long long k = 0; k < ctx->K; ++k) {
            I *r = rows + (b * ctx->K + k) * per, *c = cols + (b * ctx->K + k) * per;
            const long long r0 = b * xd * ctx->K + k * xd + base;
            long long p = 0;
            for (int seg = 0; seg < 2; ++seg) {
                const long long cb = voff + (k + seg) * zd + xo + base;
                for (long long cc = 0; cc < d; ++cc)
                    for (long long j = 0; j < n; ++j)
                        for (long long i = 0; i < n; ++i, ++p) {
                            r[p] = (I)(r0 + cc * n + i);
                            c[p] = (I)(cb + cc * n + j);
                        }
            }
            for (long long cc = 0; cc < d; ++cc)  // tail: per state column, the m drive blocks then the dt block
                for (long long l = 0; l <= m; ++l) {
                    const long long col = voff + k * zd + (l < m ? D.u_off + l : D.dt_off) + base;
                    for (long long i = 0; i < n; ++i, ++p) {
                        r[p] = (I)(r0 + cc * n + i);
                        c[p] = (I)col;
                    }
                }
        }
    }
    return PCL_OK;
}
extern "C" int pcl_jac_structure(const pcl_ctx *ctx, int32_t *rows, int32_t *cols) {
    if (ctx && (n_rows(ctx) + 1 > INT32_MAX || z_len(ctx) + ctx->desc.global_dim + 1 > INT32_MAX))
        return fail(ctx, PCL_ESHAPE, "pcl_jac_structure: indices exceed int32; use pcl_jac_structure_i64");
    return jac_structure_impl<int32_t>(ctx, rows, cols);
}
extern "C" int pcl_jac_structure_i64(const pcl_ctx *ctx, int64_t *rows, int64_t *cols) {
    return jac_structure_impl<int64_t>(ctx, rows, cols);
}

template <class I>
static int hess_structure_impl(const pcl_ctx *ctx, I *rows, I *cols) {
    if (!ctx) return PCL_EINVAL;
    if (!rows || !cols) return fail(ctx, PCL_EINVAL, "pcl_hess_structure: NULL output");
    const pcl_desc &D = ctx->desc;
    const long long m = D.n_drives, xd = ctx->x_dim, zd = D.z_dim, base = D.index_base;
    const long long per = hess_per(ctx);
    for (long long b = 0; b < D.batch; ++b) {
        const long long xo = ctx->x_offs[D.batch_mode == PCL_BATCH_MEMBERS ? b : 0];
        const long long voff = D.batch_mode == PCL_BATCH_TRAJ ? b * zd * D.N : 0;
        for (long long k = 0; k < ctx->K; ++k) {
            I *r = rows + (b * ctx->K + k) * per, *c = cols + (b * ctx->K + k) * per;
            const long long uk = voff + k * zd + D.u_off, hk = voff + k * zd + D.dt_off;
            const long long xk = voff + k * zd + xo, xn = voff + (k + 1) * zd + xo;
            long long p = 0;
            auto put = [&](long long a, long long bb) {
                r[p] = (I)(std::max(a, bb) + base);
                c[p] = (I)(std::min(a, bb) + base);
                ++p;
            };
            for (long long i = 0; i < m; ++i)
                for (long long j = 0; j <= i; ++j) put(uk + i, uk + j);
            for (long long j = 0; j < m; ++j) put(hk, uk + j);
            put(hk, hk);
            for (long long l = 0; l < m; ++l)
                for (long long q = 0; q < xd; ++q) put(uk + l, xk + q);
            for (long long q = 0; q < xd; ++q) put(hk, xk + q);
            for (long long l = 0; l < m; ++l)
                for (long long q = 0; q < xd; ++q) put(xn + q, uk + l);
            for (long long q = 0; q < xd; ++q) put(xn + q, hk);
        }
    }
    return PCL_OK;
}
extern "C" int pcl_hess_structure(const pcl_ctx *ctx, int32_t *rows, int32_t *cols) {
    if (ctx && z_len(ctx) + ctx->desc.global_dim + 1 > INT32_MAX)
        return fail(ctx, PCL_ESHAPE, "pcl_hess_structure: indices exceed int32; use pcl_hess_structure_i64");
    return hess_structure_impl<int32_t>(ctx, rows, cols);
}
extern "C" int pcl_hess_structure_i64(const pcl_ctx *ctx, int64_t *rows, int64_t *cols) {
    return hess_structure_impl<int64_t>(ctx, rows, cols);
}

// --- launch helpers -------------------------------------------------------------------------
// LD = (n rounded up to 4) + 2  ==  2*odd: conflict-free ds_read_b64 of the MFMA b operand
// (16 columns x 2 k-rows per half-wave land on 32 distinct 8-byte bank pairs).
static int lds_ld(int d) { return ((2 * d + 3) & ~3) + 2; }

static void fill_params(const pcl_ctx *ctx, KParams &p) {
    memset(&p, 0, sizeof p);
    const pcl_desc &D = ctx->desc;
    p.G0 = ctx->dG0;
    p.upos = ctx->dupos;
    p.ucoef = ctx->ducoef;
    p.n_upos = ctx->n_upos;
    p.csr_ptr = ctx->dcsr_ptr;
    p.csr_col = ctx->dcsr_col;
    p.csr_val = ctx->dcsr_val;
    p.csc_ptr = ctx->dcsc_ptr;
    p.csc_row = ctx->dcsc_row;
    p.csc_val = ctx->dcsc_val;
    p.x_offs = ctx->dxoffs;
    p.umap = ctx->dumap;
    p.uell_l = ctx->duell_l;
    p.uell_v = ctx->duell_v;
    p.uell_w = ctx->uell_w;
    p.ell_val = ctx->dell_val;
    p.ell_col = ctx->dell_col;
    p.ell_w = ctx->ell_w;
    p.ellt_val = ctx->dellt_val;
    p.ellt_col = ctx->dellt_col;
    p.ellt_w = ctx->ellt_w;
    p.hpart = ctx->dhpart;
    p.hcnt = ctx->dhcnt;
    p.ug0 = ctx->dug0;
    p.iso = ctx->iso;
    p.z_batch_stride = D.batch_mode == PCL_BATCH_TRAJ ? (long long)D.z_dim * D.N : 0;
    p.g0_batch_stride = D.per_member_G0 ? (long long)ctx->n * ctx->n : 0;
    p.d = D.d;
    p.cols = ctx->cols;
    p.n = ctx->n;
    p.m = D.n_drives;
    p.K = ctx->K;
    p.z_dim = D.z_dim;
    p.u_off = D.u_off;
    p.dt_off = D.dt_off;
    p.batch = D.batch;
    p.LD = lds_ld(D.d);
    p.nt = (int)ctx->opt_nt;
    p.ablate = (int)ctx->opt_ablate;
    p.dbg = ctx->ddbg;
    p.hess_per = hess_per(ctx);
}

static size_t fused_lds_bytes(const KParams &p, bool jac) {  // version-1 kernel
    const size_t ncols1 = jac ? (size_t)(2 + p.m) * p.nc : 2 * (size_t)p.nc;
    size_t dbl = (size_t)p.LD * p.n * (jac ? 2 : 1) + 2 * p.LD * ncols1 + 2 * (size_t)p.LD * p.nc + 8 + p.m;
    return dbl * sizeof(double);
}

static const size_t ELL_LDS_MAX_BYTES = 8192;

static size_t fused2_lds_bytes(const KParams &p, bool jac, bool ell_lds) {  // version-2 kernel
    const size_t ncols1 = jac ? (size_t)(2 + p.m) * p.nc : 2 * (size_t)p.nc;
    const size_t n_ell = (size_t)p.m * p.n * p.ell_w;
    size_t bytes = ((size_t)p.LD * p.n * (jac ? 2 : 1) + 2 * p.LD * ncols1 + (size_t)p.LD * p.nc + 2 * (p.m + 1)) * sizeof(double);
    if (jac && ell_lds) bytes += n_ell * sizeof(double) + (n_ell * sizeof(unsigned short) + 7) / 8 * 8;
    return bytes + 128;  // slack: operand tiles may be read past the last buffer's edge
}

static bool ell_fits_lds(const pcl_ctx *ctx) {
    const size_t n_ell = (size_t)ctx->desc.n_drives * ctx->n * ctx->ell_w;
    return n_ell > 0 && n_ell * (sizeof(double) + sizeof(unsigned short)) <= ELL_LDS_MAX_BYTES;
}

// State columns per workgroup.  Every slice recomputes G(u_k)^2, so fewer, wider slices do less
// arithmetic; but (i) two workgroups must fit in one CU's LDS so that one streams while the other
// computes, and (ii) the grid has to cover the chip a few times over.
static int choose_cols_per_slice(const pcl_ctx *ctx, bool jac) {
    const int d = ctx->cols;
    if (ctx->opt_cols_per_slice > 0) return (int)std::min<int64_t>(ctx->opt_cols_per_slice, d);
    KParams p;
    memset(&p, 0, sizeof p);
    p.n = ctx->n;
    p.m = ctx->desc.n_drives;
    p.LD = ((ctx->n + 3) & ~3) + 2;
    p.ell_w = ctx->ell_w;
    const bool v2 = ctx->opt_kernel == 0 || ctx->opt_kernel >= 2;
    const bool ell = ell_fits_lds(ctx);
    auto bytes = [&](int nc) {
        p.nc = nc;
        return v2 ? fused2_lds_bytes(p, jac, ell) : fused_lds_bytes(p, jac);
    };
    const long long bk = (long long)ctx->desc.batch * ctx->K;
    const long long want = 3LL * std::max(ctx->n_cu, 1);
    int best = 1;
    for (int nc = d; nc >= 1; --nc) {
        if (bytes(nc) > (size_t)ctx->max_lds / 2 && nc > 1) continue;  // keep two workgroups per CU
        const long long S = (d + nc - 1) / nc;
        best = nc;
        if (!jac || bk * S >= want) break;
    }
    return best;
}

static bool v3_supported(const pcl_ctx *ctx) { return 2 + ctx->desc.n_drives <= 16; }
static int v3_ncw(const pcl_ctx *ctx, int nc) { return std::max(1, std::min(nc, 16 / (2 + ctx->desc.n_drives))); }

static size_t fused3_lds_bytes(const pcl_ctx *ctx, const KParams &p, bool tab) {  // version-3 kernel
    const size_t tile = (size_t)p.LD * p.n, wsz = (size_t)p.LD * (16 + 3 * p.ncw);
    const size_t n_ell = (size_t)p.m * p.n * p.ell_w, n_un = (size_t)ctx->n_upos, uw = (size_t)ctx->uell_w;
    size_t bytes = (4 * tile + 4 * wsz + 3 * (size_t)(p.m + 1)) * sizeof(double);
    if (tab) bytes += (n_un * uw + n_un + n_ell) * sizeof(double) + (n_un + n_ell) * 2 + n_un * uw;
    return (bytes + 7) / 8 * 8 + 128;  // slack: operand tiles may be read past the last buffer's edge
}

// v3 cost model: one workgroup per CU walks ceil(items / CUs) items; an item costs max(store stream, matrix work).
// Constants are the measured config-3 phase times (scripts/phase_timing.py, with the stream running): ~7 us per
// 2-column chunk, ~4 us for G(u) + G^2, stream at ~0.85 of the CU's fair HBM share; other shapes scale by MFMA count.
// Role split needs the chunk buffers of matrix waves 4..7 inside the second halves of the G / G^2 double buffers.
static bool v3_role_split_fits(const pcl_ctx *ctx) {
    const int ncw = v3_ncw(ctx, ctx->desc.d), LD = lds_ld(ctx->desc.d);
    return 2 * (size_t)LD * (16 + 3 * ncw) <= (size_t)LD * ctx->n;
}
// Work split of kernel 3: contiguous column ranges (+ role split) pay off once every CU has a few intervals' worth of
// columns; below that the round-robin slices balance a short launch better (measured: batch >= 3 at config 3).
static bool v3_contiguous(const pcl_ctx *ctx) {
    if (ctx->opt_cols_per_slice > 0 || ctx->opt_contig == 0) return false;
    if (ctx->opt_contig > 0) return true;
    const long long cols = (long long)ctx->desc.batch * ctx->K * ctx->desc.d;
    return v3_role_split_fits(ctx) && cols >= 28LL * std::max(ctx->n_cu, 1);
}
static int choose_cols_v3(const pcl_ctx *ctx) {
    const int d = ctx->desc.d, n = ctx->n, m = ctx->desc.n_drives;
    if (ctx->opt_cols_per_slice > 0) return (int)std::min<int64_t>(ctx->opt_cols_per_slice, d);
    const double hbm = 0.85 * 6.3e12;  // store-stream rate the kernel sustains chip-wide
    const long long bk = (long long)ctx->desc.batch * ctx->K;
    const int rt = (n + 15) / 16, ks = (n + 3) / 4;
    const int g2ct = ctx->iso ? (d + 15) / 16 : (n + 15) / 16;
    const double t_chunk = 7.0e-6 * (rt * ks) / (4.0 * 14.0);
    const double t_build = 4.0e-6 * (((rt + 3) / 4) * ((g2ct + 1) / 2) * ks) / 14.0;
    std::vector<double> tt(d + 1, 1e300);
    double best_t = 1e300;
    for (int nc = 1; nc <= d; ++nc) {
        const int ncw = v3_ncw(ctx, nc);
        const long long S = (d + nc - 1) / nc;
        const long long items = bk * S, ncu = std::max(ctx->n_cu, 1);
        const int chunks_per_wave = ((nc + ncw - 1) / ncw + 3) / 4;
        const double t_matrix = t_build + chunks_per_wave * t_chunk;
        const double item_bytes = 2.0 * nc * n * n * 8.0;
        // rounds of up to n_cu items; the HBM rate is shared by the workgroups active in the round (a lone CU tops out
        // at a few times its fair share)
        double t = 0.0;
        for (long long left = items; left > 0; left -= ncu) {
            const double active = (double)std::min(left, ncu);
            const double rate = std::min(hbm / active, 3.0 * hbm / (double)ncu);
            t += std::max(item_bytes / rate, t_matrix);
        }
        // a ragged last slice (d % nc != 0) leaves workgroups with unequal items: charge the mean fill
        const double fill = (double)d / (double)(S * nc);
        tt[nc] = t / std::sqrt(std::max(fill, 0.25)) + 2.0 * t_build;
        best_t = std::min(best_t, tt[nc]);
    }
    // among near-ties take the narrowest slice (more, smaller items balance better across the CUs)
    int best = d;
    for (int nc = d; nc >= 1; --nc)
        if (tt[nc] <= 1.06 * best_t) best = nc;
    return best;
}

static int set_lds_attr(pcl_ctx *ctx, const void *kern, int slot, size_t lds) {
    if (ctx->lds_set[slot] == lds && ctx->lds_kern[slot] == kern) return PCL_OK;
    ctx->lds_kern[slot] = kern;
    HIP_TRY(ctx, hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    ctx->lds_set[slot] = lds;
    return PCL_OK;
}

// General-order kernel (pade_order != 4, or option general_pade_kernel): slice width from the LDS budget.
static size_t pade_lds_bytes(const KParams &p, bool jac) {
    const size_t tiles = (jac ? 3 : 1) * (size_t)p.LD * p.n;
    const size_t percol = (size_t)(p.q + 1) + 2 + (jac ? 2 + 2 * (size_t)p.m : 0);
    return (tiles + percol * p.LD * p.nc + 8 + p.m) * sizeof(double);
}
static int launch_pade_general(pcl_ctx *ctx, KParams &p, bool want_jac) {
    p.q = ctx->desc.pade_order / 2;
    double f[16];
    f[0] = 1.0;
    for (int i = 1; i < 16; ++i) f[i] = f[i - 1] * i;
    for (int j = 0; j <= p.q; ++j) p.pc[j] = f[2 * p.q - j] * f[p.q] / (f[2 * p.q] * f[j] * f[p.q - j]);
    p.nc = ctx->opt_cols_per_slice > 0 ? (int)std::min<int64_t>(ctx->opt_cols_per_slice, p.cols) : p.cols;
    while (p.nc > 1 && pade_lds_bytes(p, want_jac) > (size_t)ctx->max_lds) --p.nc;
    const size_t lds = pade_lds_bytes(p, want_jac);
    if (lds > (size_t)ctx->max_lds)
        return fail(ctx, PCL_ESHAPE, "general-order kernel needs %zu B of LDS (> %d) for d=%d, m=%d, order %d", lds, ctx->max_lds, p.d, p.m, 2 * p.q);
    p.S = (p.cols + p.nc - 1) / p.nc;
    const long long grid = (long long)p.batch * p.K * p.S;
    if (grid > 0x7fffffffLL) return fail(ctx, PCL_ESHAPE, "too many work items");
    typedef void (*kern_t)(const KParams);
    kern_t kern = want_jac ? (kern_t)pcl_pade_kernel<true> : (kern_t)pcl_pade_kernel<false>;
    HIP_TRY(ctx, hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), lds, ctx->stream, p);
    HIP_TRY(ctx, hipGetLastError());
    ctx->last_kernel = 90 + p.q;
    ctx->last_n_stream = 0;
    return PCL_OK;
}

static int launch_fused(pcl_ctx *ctx, const double *Z, double *delta, double *jac, bool compact) {
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    KParams p;
    fill_params(ctx, p);
    p.Z = Z;
    p.delta = delta;
    p.jac = jac;
    p.compact = compact ? 1 : 0;
    p.jac_per = compact ? jac_per_compact(ctx) : jac_per_full(ctx);
    const bool want_jac = jac != nullptr;
    if (ctx->desc.pade_order != 4 || ctx->opt_general) return launch_pade_general(ctx, p, want_jac);
    p.ell_lds = ell_fits_lds(ctx) ? 1 : 0;
    if (want_jac && !compact && (ctx->opt_kernel == 3 || ctx->opt_kernel == 0) && ctx->opt_use_mfma != 0 && v3_supported(ctx) &&
        ctx->cols == ctx->desc.d) {  // the default whenever its LDS budget fits (else kernels 2 / 4 below)
        // default: contiguous column ranges (one item per interval touched); an explicit cols_per_slice or
        // contiguous = 0 selects the round-robin slices
        p.contig = v3_contiguous(ctx) ? 1 : 0;
        p.flat = ctx->opt_flat ? 1 : 0;
        p.snc = (int)std::max<int64_t>(0, std::min<int64_t>(ctx->opt_snc, p.d));
        p.nc = p.contig ? p.d : choose_cols_v3(ctx);
        p.ncw = v3_ncw(ctx, p.nc);
        p.tab_lds = 1;
        size_t lds3 = fused3_lds_bytes(ctx, p, true);
        if (lds3 > (size_t)ctx->max_lds) {  // large union / ELL tables stay in memory
            p.tab_lds = 0;
            lds3 = fused3_lds_bytes(ctx, p, false);
        }
        if (lds3 > (size_t)ctx->max_lds) goto not_v3;  // double-buffered tiles do not fit (n close to 64): kernel v2
        p.S = (p.d + p.nc - 1) / p.nc;
        const long long items = (long long)p.batch * p.K * p.S;
        if (items > 0x7fffffffLL) return fail(ctx, PCL_ESHAPE, "too many work items");
        typedef void (*kern3_t)(const KParams);
        const int ewr = (p.m <= 8 && ctx->ell_w >= 1 && ctx->ell_w <= 2) ? ctx->ell_w : 0;
        kern3_t kern3 = ewr == 1 ? (kern3_t)pcl_fused_kernel_v3<1, 0, 0, 0> : ewr == 2 ? (kern3_t)pcl_fused_kernel_v3<2, 0, 0, 0> : (kern3_t)pcl_fused_kernel_v3<0, 0, 0, 0>;
        // shape-specialised instance (BASELINE config 3/4/5: d = 27, six drives with two entries per row, 2-column chunks)
        if (ewr == 2 && p.d == 27 && p.m == 6 && p.ncw == 2 && ctx->opt_specialize) kern3 = (kern3_t)pcl_fused_kernel_v3<2, 27, 6, 2>;
        ctx->last_kernel = 30 + ((ewr == 2 && p.d == 27 && p.m == 6 && p.ncw == 2 && ctx->opt_specialize) ? 1 : 0);
        int rc = set_lds_attr(ctx, (const void *)kern3, 6, lds3);
        if (rc != PCL_OK) return rc;
        const long long units = p.contig ? (long long)p.batch * p.K * p.d : items;  // what the grid is cut into
        const long long g3 = ctx->opt_grid > 0 ? std::min<long long>(ctx->opt_grid, units) : std::min<long long>(units, std::max(ctx->n_cu, 1));
        p.n_stream = 0;
        if (p.contig && g3 >= 2 && v3_role_split_fits(ctx)) {
            const long long want = ctx->opt_stream_wg < 0 ? g3 / 2 : ctx->opt_stream_wg;  // auto: half the workgroups stream
            if (want > 0) p.n_stream = (int)std::min<long long>(want, g3 - 1);
        }
        ctx->last_n_stream = p.n_stream;
        hipLaunchKernelGGL(kern3, dim3((unsigned)g3), dim3(512), lds3, ctx->stream, p);
        HIP_TRY(ctx, hipGetLastError());
        return PCL_OK;
    }
not_v3:
    const bool v2 = (ctx->opt_kernel == 0 || ctx->opt_kernel >= 2) && ctx->opt_use_mfma != 0;
    const bool unitary = ctx->cols == ctx->desc.d;  // kernels 3, 4, 5 and the specialised instances assume X is n x d
    bool v4 = v2 && ctx->opt_kernel == 4 && want_jac && unitary;
    p.nc = choose_cols_per_slice(ctx, want_jac);
    auto bytes = [&]() { return v2 ? fused2_lds_bytes(p, want_jac, p.ell_lds != 0) : fused_lds_bytes(p, want_jac); };
    size_t lds = bytes();
    while (lds > (size_t)ctx->max_lds && p.nc > 1) {
        p.nc = (p.nc + 1) / 2;
        lds = bytes();
    }
    if (lds > (size_t)ctx->max_lds) return fail(ctx, PCL_ESHAPE, "fused kernel needs %zu B of LDS (> %d)", lds, ctx->max_lds);
    p.S = (p.cols + p.nc - 1) / p.nc;
    const long long grid = (long long)p.batch * p.K * p.S;
    if (grid > 0x7fffffffLL) return fail(ctx, PCL_ESHAPE, "grid too large");
    if (v2) {
        typedef void (*kern_t)(const KParams);
        const int per_cu_guess = std::max(1, std::min(2, (int)((size_t)ctx->max_lds / lds)));
        const int wu = (ctx->uell_w <= 2 && ctx->n_upos <= 1024 && !ctx->desc.per_member_G0) ? ctx->uell_w : -1;
        kern_t kern = want_jac ? (wu == 1 ? (kern_t)pcl_fused_kernel_v2<true, 1, 0, 0, 0> : wu == 2 ? (kern_t)pcl_fused_kernel_v2<true, 2, 0, 0, 0> : (kern_t)pcl_fused_kernel_v2<true, -1, 0, 0, 0>)
                               : (wu == 1 ? (kern_t)pcl_fused_kernel_v2<false, 1, 0, 0, 0> : wu == 2 ? (kern_t)pcl_fused_kernel_v2<false, 2, 0, 0, 0> : (kern_t)pcl_fused_kernel_v2<false, -1, 0, 0, 0>);
        // shape-specialised instance (BASELINE config 3/4/5: three 3-level transmons, d = 27, six drives, 3-column slices)
        if (want_jac && unitary && wu == 1 && p.d == 27 && p.m == 6 && p.nc == 3 && ctx->opt_specialize) kern = (kern_t)pcl_fused_kernel_v2<true, 1, 27, 6, 3>;
        // auto: spreading an item's stores into the next item's phases pays once a workgroup walks several items
        if (ctx->opt_kernel == 0 && want_jac && unitary && grid >= 4 * (long long)per_cu_guess * std::max(ctx->n_cu, 1)) v4 = true;
        if (v4) {
            kern = wu == 1 ? (kern_t)pcl_fused_kernel_v4<1, 0, 0, 0> : wu == 2 ? (kern_t)pcl_fused_kernel_v4<2, 0, 0, 0> : (kern_t)pcl_fused_kernel_v4<-1, 0, 0, 0>;
            if (wu == 1 && p.d == 27 && p.m == 6 && p.nc == 3 && ctx->opt_specialize) kern = (kern_t)pcl_fused_kernel_v4<1, 27, 6, 3>;
        }
        ctx->last_kernel = (v4 ? 40 : 20) + ((unitary && wu == 1 && p.d == 27 && p.m == 6 && p.nc == 3 && ctx->opt_specialize && want_jac) ? 1 : 0);
        int rc = set_lds_attr(ctx, (const void *)kern, want_jac ? 4 : 5, lds);  // wu is fixed per context
        if (rc != PCL_OK) return rc;
        const bool split = ctx->opt_kernel == 5 && want_jac && !compact && unitary;
        if (split) {
            const long long n_bk = (long long)p.batch * p.K;
            if (!ctx->dblocks) {
                HIP_TRY(ctx, hipMalloc((void **)&ctx->dblocks, (size_t)n_bk * 2 * p.n * p.n * sizeof(double)));
                HIP_TRY(ctx, hipMalloc((void **)&ctx->dflags, (size_t)n_bk * sizeof(unsigned int)));
                HIP_TRY(ctx, hipStreamCreateWithFlags(&ctx->aux_stream, hipStreamNonBlocking));
                HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming));
                HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->ev_join, hipEventDisableTiming));
            }
            p.compact = 2;
            p.blocks = ctx->dblocks;
            p.flags = ctx->dflags;
            kern = wu == 1 ? (kern_t)pcl_fused_kernel_v2<true, 1, 0, 0, 0> : wu == 2 ? (kern_t)pcl_fused_kernel_v2<true, 2, 0, 0, 0> : (kern_t)pcl_fused_kernel_v2<true, -1, 0, 0, 0>;
            if (wu == 1 && p.d == 27 && p.m == 6 && p.nc == 3 && ctx->opt_specialize) kern = (kern_t)pcl_fused_kernel_v2<true, 1, 27, 6, 3>;
            ctx->last_kernel = 50;
            HIP_TRY(ctx, hipMemsetAsync(ctx->dflags, (ctx->opt_ablate & 64) ? 0xFF : 0, (size_t)n_bk * sizeof(unsigned int), ctx->stream));
            HIP_TRY(ctx, hipEventRecord(ctx->ev_fork, ctx->stream));
            HIP_TRY(ctx, hipStreamWaitEvent(ctx->aux_stream, ctx->ev_fork, 0));
        }
        // persistent grid: as many workgroups as are resident at once
        const int per_cu = std::max(1, std::min(2, (int)((size_t)ctx->max_lds / lds)));
        const long long resident = (long long)per_cu * std::max(ctx->n_cu, 1);
        const long long g2 = ctx->opt_grid > 0 ? std::min<long long>(ctx->opt_grid, grid) : std::min(grid, resident);
        if (!(split && (ctx->opt_ablate & 64))) hipLaunchKernelGGL(kern, dim3((unsigned)g2), dim3(512), lds, ctx->stream, p);
        if (split) {
            // the expander leaves wave slots and all LDS to the producer: at most 4 x 256 threads per CU
            const int cpp = (int)std::max<int64_t>(1, std::min<int64_t>(ctx->opt_cpp, 2 * p.d));
            const int pieces = (2 * p.d + cpp - 1) / cpp;
            const long long n_bk = (long long)p.batch * p.K;
            const long long eg = std::min<long long>(n_bk * pieces, 4LL * std::max(ctx->n_cu, 1));
            hipLaunchKernelGGL(pcl_expand_stream_kernel, dim3((unsigned)eg), dim3(256), 0, ctx->aux_stream, ctx->dblocks, ctx->dflags, jac,
                               p.d, p.n, jac_per_full(ctx), n_bk, pieces, cpp, p.nt);
            HIP_TRY(ctx, hipGetLastError());
            HIP_TRY(ctx, hipEventRecord(ctx->ev_join, ctx->aux_stream));
            HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_join, 0));
        }
    } else {
        const bool mf = ctx->opt_use_mfma != 0;
        auto kern = want_jac ? (mf ? pcl_fused_kernel<true, true> : pcl_fused_kernel<true, false>)
                             : (mf ? pcl_fused_kernel<false, true> : pcl_fused_kernel<false, false>);
        int rc = set_lds_attr(ctx, (const void *)kern, (want_jac ? 0 : 2) + (mf ? 0 : 1), lds);
        if (rc != PCL_OK) return rc;
        hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), lds, ctx->stream, p);
    }
    HIP_TRY(ctx, hipGetLastError());
    return PCL_OK;
}

static size_t hess_lds_bytes(const KParams &p) {
    const size_t nscal = (size_t)(p.m + 1) * (p.m + 2) / 2;
    return ((size_t)p.LD * p.n + (6 + 3 * (size_t)p.m) * p.LD * p.nc + 8 + p.m + 5 * nscal) * sizeof(double);
}

template <int EW, bool ANTI>
static const void *hess_v2_kernel(int m) {
    switch (m) {
    case 1: return (const void *)pcl_hess_kernel_v2<EW, 1, 0, ANTI>;
    case 2: return (const void *)pcl_hess_kernel_v2<EW, 2, 0, ANTI>;
    case 3: return (const void *)pcl_hess_kernel_v2<EW, 3, 0, ANTI>;
    case 4: return (const void *)pcl_hess_kernel_v2<EW, 4, 0, ANTI>;
    case 5: return (const void *)pcl_hess_kernel_v2<EW, 5, 0, ANTI>;
    case 6: return (const void *)pcl_hess_kernel_v2<EW, 6, 0, ANTI>;
    }
    return nullptr;
}
#define PCL_HESS_EW 2
static bool hess_v2_supported(const pcl_ctx *ctx) {
    const int m = ctx->desc.n_drives;
    return m >= 1 && m <= 6 && ctx->ell_w <= PCL_HESS_EW && ctx->ellt_w <= PCL_HESS_EW;
}
static size_t hess2_lds_bytes(const KParams &p) {
    const size_t nscal = (size_t)(p.m + 1) * (p.m + 2) / 2;
    return ((size_t)p.LD * p.n + 7 * (size_t)p.LD * 16 + 4 * nscal + 4 + 2) * sizeof(double);
}

static int launch_hess(pcl_ctx *ctx, const double *Z, const double *mu, double *hess) {
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    KParams p;
    fill_params(ctx, p);
    p.Z = Z;
    p.mu = mu;
    p.hess = hess;
    const bool mf = ctx->opt_use_mfma != 0;
    if (ctx->desc.pade_order != 4)
        return fail(ctx, PCL_ENOTIMPL, "the Hessian of the Lagrangian is implemented for pade_order 4 only (have %d)", ctx->desc.pade_order);
    if (ctx->opt_hess_kernel == 2 && !hess_v2_supported(ctx))
        return fail(ctx, PCL_ESHAPE, "hess_kernel=2 needs 1..6 drives with at most %d entries per row and column (have m=%d, widths %d/%d)",
                    PCL_HESS_EW, p.m, ctx->ell_w, ctx->ellt_w);
    if (mf && ctx->opt_hess_kernel != 1 && hess_v2_supported(ctx) && hess2_lds_bytes(p) <= (size_t)ctx->max_lds) {
        // one chunk of 16/(m+1) columns per wave and item when the interval's columns allow it, never more than 16 per slice
        const int ncw = 16 / (p.m + 1);
        int S = (p.cols + 4 * ncw - 1) / (4 * ncw);
        if (ctx->opt_cols_per_slice > 0) S = (p.cols + (int)std::min<int64_t>(ctx->opt_cols_per_slice, 16) - 1) / (int)std::min<int64_t>(ctx->opt_cols_per_slice, 16);
        S = std::max(S, (p.cols + 15) / 16);
        p.nc = (p.cols + S - 1) / S;
        p.S = (p.cols + p.nc - 1) / p.nc;
        const long long nbk = (long long)p.batch * p.K;
        if (p.S > 1 && ctx->hpart_cap < nbk * p.S) {
            const size_t nscal = (size_t)(p.m + 1) * (p.m + 2) / 2;
            if (ctx->dhpart) (void)hipFree(ctx->dhpart);
            if (ctx->dhcnt) (void)hipFree(ctx->dhcnt);
            ctx->dhpart = nullptr;
            ctx->dhcnt = nullptr;
            ctx->hpart_cap = 0;
            HIP_TRY(ctx, hipMalloc((void **)&ctx->dhpart, (size_t)nbk * p.S * nscal * sizeof(double)));
            HIP_TRY(ctx, hipMalloc((void **)&ctx->dhcnt, (size_t)nbk * sizeof(unsigned int)));
            HIP_TRY(ctx, hipMemsetAsync(ctx->dhcnt, 0, (size_t)nbk * sizeof(unsigned int), ctx->stream));
            ctx->hpart_cap = nbk * p.S;
        }
        p.hpart = ctx->dhpart;
        p.hcnt = ctx->dhcnt;
        const size_t lds = hess2_lds_bytes(p);
        const void *kern = ctx->drives_antisym ? hess_v2_kernel<PCL_HESS_EW, true>(p.m) : hess_v2_kernel<PCL_HESS_EW, false>(p.m);
        if (ctx->opt_specialize && p.d == 27 && p.m == 6 && ctx->drives_antisym)
            kern = (const void *)pcl_hess_kernel_v2<PCL_HESS_EW, 6, 27, true>;  // BASELINE config 3's shape
        if (int rc = set_lds_attr(ctx, kern, 7, lds)) return rc;
        const long long items = nbk * p.S;
        const int per_cu = std::max(1, std::min(2, (int)((size_t)ctx->max_lds / lds)));
        long long grid = std::min<long long>(items, (long long)per_cu * ctx->n_cu);
        if (ctx->opt_grid > 0) grid = std::min<long long>(items, ctx->opt_grid);
        void *args[] = {(void *)&p};
        HIP_TRY(ctx, hipLaunchKernel(kern, dim3((unsigned)grid), dim3(256), args, lds, ctx->stream));
        HIP_TRY(ctx, hipGetLastError());
        ctx->last_hess_kernel = 2;
        return PCL_OK;
    }
    // column chunk: as many columns as fit in half the LDS (two workgroups per CU)
    p.nc = p.cols;
    while (p.nc > 1 && hess_lds_bytes(p) > (size_t)ctx->max_lds / 2) p.nc = (p.nc + 1) / 2;
    const size_t lds = hess_lds_bytes(p);
    if (lds > (size_t)ctx->max_lds)
        return fail(ctx, PCL_ESHAPE, "Hessian kernel needs %zu B of LDS (> %d) for d=%d, m=%d", lds, ctx->max_lds, p.d, p.m);
    const long long grid = (long long)p.batch * p.K;
    auto kern = mf ? pcl_hess_kernel<true> : pcl_hess_kernel<false>;
    HIP_TRY(ctx, hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), lds, ctx->stream, p);
    HIP_TRY(ctx, hipGetLastError());
    ctx->last_hess_kernel = 1;
    return PCL_OK;
}

// --- device-pointer API -----------------------------------------------------------------------
extern "C" int pcl_set_stream(pcl_ctx *ctx, void *s) {
    if (!ctx) return PCL_EINVAL;
    ctx->stream = (hipStream_t)s;  // NULL is HIP's legacy default stream
    return PCL_OK;
}
extern "C" int pcl_reset_stream(pcl_ctx *ctx) {
    if (!ctx) return PCL_EINVAL;
    ctx->stream = ctx->own_stream;
    return PCL_OK;
}
extern "C" int pcl_sync(pcl_ctx *ctx) {
    if (!ctx) return PCL_EINVAL;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return PCL_OK;
}
extern "C" int pcl_eval_dev(pcl_ctx *ctx, const double *Z, double *delta) {
    if (!ctx) return PCL_EINVAL;
    if (!Z || !delta) return fail(ctx, PCL_EINVAL, "pcl_eval_dev: NULL pointer");
    return launch_fused(ctx, Z, delta, nullptr, false);
}
extern "C" int pcl_eval_jac_dev(pcl_ctx *ctx, const double *Z, double *delta, double *vals) {
    if (!ctx) return PCL_EINVAL;
    if (!Z || !vals) return fail(ctx, PCL_EINVAL, "pcl_eval_jac_dev: NULL pointer");
    return launch_fused(ctx, Z, delta, vals, false);
}
extern "C" int pcl_eval_jac_compact_dev(pcl_ctx *ctx, const double *Z, double *delta, double *compact) {
    if (!ctx) return PCL_EINVAL;
    if (!Z || !compact) return fail(ctx, PCL_EINVAL, "pcl_eval_jac_compact_dev: NULL pointer");
    return launch_fused(ctx, Z, delta, compact, true);
}
extern "C" int pcl_jac_expand_dev(pcl_ctx *ctx, const double *compact, double *vals) {
    if (!ctx) return PCL_EINVAL;
    if (!compact || !vals) return fail(ctx, PCL_EINVAL, "pcl_jac_expand_dev: NULL pointer");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const long long n_bk = (long long)ctx->desc.batch * ctx->K;
    const long long grid = n_bk * ctx->cols;
    hipLaunchKernelGGL(pcl_expand_kernel, dim3((unsigned)grid), dim3(256), 0, ctx->stream, compact, vals, ctx->cols, ctx->n,
                       ctx->desc.n_drives, n_bk, (int)ctx->opt_nt);
    HIP_TRY(ctx, hipGetLastError());
    return PCL_OK;
}
extern "C" int pcl_hess_dev(pcl_ctx *ctx, const double *Z, const double *mu, double *vals) {
    if (!ctx) return PCL_EINVAL;
    if (!Z || !mu || !vals) return fail(ctx, PCL_EINVAL, "pcl_hess_dev: NULL pointer");
    return launch_hess(ctx, Z, mu, vals);
}

extern "C" int pcl_rollout_dev(pcl_ctx *ctx, const double *Z, double *X_out) {
    if (!ctx) return PCL_EINVAL;
    if (!Z || !X_out) return fail(ctx, PCL_EINVAL, "pcl_rollout_dev: NULL pointer");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    KParams p;
    fill_params(ctx, p);
    p.Z = Z;
    p.xout = X_out;
    if (!ctx->dexpm) HIP_TRY(ctx, hipMalloc((void **)&ctx->dexpm, (size_t)p.batch * p.K * p.n * p.n * sizeof(double)));
    p.expm = ctx->dexpm;
    const size_t lds_a = (3 * (size_t)p.LD * p.n + 8 + p.m + 64) * sizeof(double);
    const size_t lds_b = ((size_t)p.LD * p.n + 2 * (size_t)p.LD * p.cols) * sizeof(double);
    if (lds_a > (size_t)ctx->max_lds || lds_b > (size_t)ctx->max_lds) return fail(ctx, PCL_ESHAPE, "pcl_rollout_dev: tiles exceed LDS");
    HIP_TRY(ctx, hipFuncSetAttribute((const void *)pcl_expm_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_a));
    HIP_TRY(ctx, hipFuncSetAttribute((const void *)pcl_chain_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_b));
    hipLaunchKernelGGL(pcl_expm_kernel, dim3((unsigned)((long long)p.batch * p.K)), dim3(256), lds_a, ctx->stream, p);
    HIP_TRY(ctx, hipGetLastError());
    hipLaunchKernelGGL(pcl_chain_kernel, dim3((unsigned)p.batch), dim3(256), lds_b, ctx->stream, p);
    HIP_TRY(ctx, hipGetLastError());
    return PCL_OK;
}

// --- host-pointer API (staging buffers owned by the context) --------------------------------------
static int ensure(pcl_ctx *ctx, double **buf, long long count) {
    if (*buf) return PCL_OK;
    hipError_t e = hipMalloc((void **)buf, (size_t)count * sizeof(double));
    if (e != hipSuccess) return fail(ctx, e == hipErrorOutOfMemory ? PCL_ENOMEM : PCL_EHIP, "hipMalloc(%lld doubles): %s", count, hipGetErrorString(e));
    return PCL_OK;
}
#define TRY(expr)                 \
    do {                          \
        int rc_ = (expr);         \
        if (rc_ != PCL_OK) return rc_; \
    } while (0)

static int host_eval_jac(pcl_ctx *ctx, const double *Z, double *delta, double *vals) {
    if (!ctx) return PCL_EINVAL;
    if (!Z || (!delta && !vals)) return fail(ctx, PCL_EINVAL, "NULL pointer");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const long long nv = jac_per_full(ctx) * ctx->desc.batch * ctx->K;
    TRY(ensure(ctx, &ctx->dZ, z_len(ctx)));
    TRY(ensure(ctx, &ctx->ddelta, n_rows(ctx)));
    if (vals) TRY(ensure(ctx, &ctx->dvals, nv));
    HIP_TRY(ctx, hipMemcpyAsync(ctx->dZ, Z, z_len(ctx) * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    TRY(launch_fused(ctx, ctx->dZ, ctx->ddelta, vals ? ctx->dvals : nullptr, false));
    if (delta) HIP_TRY(ctx, hipMemcpyAsync(delta, ctx->ddelta, n_rows(ctx) * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    if (vals) HIP_TRY(ctx, hipMemcpyAsync(vals, ctx->dvals, nv * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return PCL_OK;
}
extern "C" int pcl_eval(pcl_ctx *ctx, const double *Z, double *delta) {
    if (ctx && !delta) return fail(ctx, PCL_EINVAL, "pcl_eval: delta is NULL");
    return host_eval_jac(ctx, Z, delta, nullptr);
}
extern "C" int pcl_jac(pcl_ctx *ctx, const double *Z, double *vals) {
    if (ctx && !vals) return fail(ctx, PCL_EINVAL, "pcl_jac: vals is NULL");
    return host_eval_jac(ctx, Z, nullptr, vals);
}
extern "C" int pcl_eval_jac(pcl_ctx *ctx, const double *Z, double *delta, double *vals) {
    if (ctx && (!delta || !vals)) return fail(ctx, PCL_EINVAL, "pcl_eval_jac: NULL output");
    return host_eval_jac(ctx, Z, delta, vals);
}
extern "C" int pcl_hess(pcl_ctx *ctx, const double *Z, const double *mu, double *vals) {
    if (!ctx) return PCL_EINVAL;
    if (!Z || !mu || !vals) return fail(ctx, PCL_EINVAL, "pcl_hess: NULL pointer");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const long long nv = hess_per(ctx) * ctx->desc.batch * ctx->K;
    TRY(ensure(ctx, &ctx->dZ, z_len(ctx)));
    TRY(ensure(ctx, &ctx->dmu, n_rows(ctx)));
    TRY(ensure(ctx, &ctx->dhess, nv));
    HIP_TRY(ctx, hipMemcpyAsync(ctx->dZ, Z, z_len(ctx) * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(ctx->dmu, mu, n_rows(ctx) * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    TRY(launch_hess(ctx, ctx->dZ, ctx->dmu, ctx->dhess));
    HIP_TRY(ctx, hipMemcpyAsync(vals, ctx->dhess, nv * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return PCL_OK;
}

extern "C" int pcl_rollout(pcl_ctx *ctx, const double *Z, double *X_out) {
    if (!ctx) return PCL_EINVAL;
    if (!Z || !X_out) return fail(ctx, PCL_EINVAL, "pcl_rollout: NULL pointer");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const long long nv = (long long)ctx->desc.batch * ctx->desc.N * ctx->x_dim;
    TRY(ensure(ctx, &ctx->dZ, z_len(ctx)));
    TRY(ensure(ctx, &ctx->dxout, nv));
    HIP_TRY(ctx, hipMemcpyAsync(ctx->dZ, Z, z_len(ctx) * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    TRY(pcl_rollout_dev(ctx, ctx->dZ, ctx->dxout));
    HIP_TRY(ctx, hipMemcpyAsync(X_out, ctx->dxout, nv * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return PCL_OK;
}

// --- DerivativeIntegrator / time-consistency rows ---------------------------
static int deriv_check(const pcl_ctx *ctx, int x_off, int dx_off, int dim) {
    const int zd = ctx->desc.z_dim;
    if (dim < 1 || x_off < 0 || x_off + dim > zd || (dx_off >= 0 && dx_off + dim > zd))
        return fail(ctx, PCL_EINVAL, "derivative rows: components outside the knot (x_off=%d dx_off=%d dim=%d z_dim=%d)", x_off, dx_off, dim, zd);
    return PCL_OK;
}
extern "C" int pcl_deriv_nnz(const pcl_ctx *ctx, int32_t dx_off, int32_t dim, int64_t *rows, int64_t *nnz) {
    if (!ctx) return PCL_EINVAL;
    const long long nb = ctx->desc.batch_mode == PCL_BATCH_TRAJ ? ctx->desc.batch : 1;
    if (rows) *rows = nb * ctx->K * dim;
    if (nnz) *nnz = nb * ctx->K * dim * (dx_off >= 0 ? 4 : 3);
    return PCL_OK;
}
extern "C" int pcl_deriv_structure(const pcl_ctx *ctx, int32_t x_off, int32_t dx_off, int32_t dim, int64_t *rows, int64_t *cols) {
    if (!ctx) return PCL_EINVAL;
    if (!rows || !cols) return fail(ctx, PCL_EINVAL, "pcl_deriv_structure: NULL output");
    TRY(deriv_check(ctx, x_off, dx_off, dim));
    const pcl_desc &D = ctx->desc;
    const long long nb = D.batch_mode == PCL_BATCH_TRAJ ? D.batch : 1, zd = D.z_dim, base = D.index_base;
    const int nseg = dx_off >= 0 ? 4 : 3;
    long long p = 0;
    for (long long b = 0; b < nb; ++b)
        for (long long k = 0; k < ctx->K; ++k) {
            const long long v0 = b * zd * D.N + k * zd + base, r0 = (b * ctx->K + k) * dim + base;
            for (int seg = 0; seg < nseg; ++seg)
                for (long long r = 0; r < dim; ++r, ++p) {
                    rows[p] = r0 + r;
                    cols[p] = seg == 0 ? v0 + x_off + r : seg == 1 ? v0 + zd + x_off + r : (seg == 2 && dx_off >= 0) ? v0 + dx_off + r : v0 + D.dt_off;
                }
        }
    return PCL_OK;
}
extern "C" int pcl_deriv_eval_jac_dev(pcl_ctx *ctx, int32_t x_off, int32_t dx_off, int32_t dim, const double *Z, double *delta,
                                      double *vals) {
    if (!ctx) return PCL_EINVAL;
    if (!Z || (!delta && !vals)) return fail(ctx, PCL_EINVAL, "pcl_deriv_eval_jac_dev: NULL pointer");
    TRY(deriv_check(ctx, x_off, dx_off, dim));
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const pcl_desc &D = ctx->desc;
    const long long nb = D.batch_mode == PCL_BATCH_TRAJ ? D.batch : 1;
    const long long total = nb * ctx->K * dim;
    const unsigned grid = (unsigned)std::min<long long>((total + 255) / 256, 4096);
    hipLaunchKernelGGL(pcl_deriv_kernel, dim3(grid), dim3(256), 0, ctx->stream, Z, delta, vals, ctx->K, D.z_dim, x_off, dx_off, dim,
                       D.dt_off, D.batch_mode == PCL_BATCH_TRAJ ? (long long)D.z_dim * D.N : 0LL, total);
    HIP_TRY(ctx, hipGetLastError());
    return PCL_OK;
}
extern "C" int pcl_deriv_eval_jac(pcl_ctx *ctx, int32_t x_off, int32_t dx_off, int32_t dim, const double *Z, double *delta, double *vals) {
    if (!ctx) return PCL_EINVAL;
    if (!Z || (!delta && !vals)) return fail(ctx, PCL_EINVAL, "pcl_deriv_eval_jac: NULL pointer");
    TRY(deriv_check(ctx, x_off, dx_off, dim));
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    int64_t nr = 0, nz = 0;
    pcl_deriv_nnz(ctx, dx_off, dim, &nr, &nz);
    double *dd = nullptr, *dv = nullptr;
    TRY(ensure(ctx, &ctx->dZ, z_len(ctx)));
    HIP_TRY(ctx, hipMalloc((void **)&dd, (size_t)nr * sizeof(double)));
    if (hipMalloc((void **)&dv, (size_t)nz * sizeof(double)) != hipSuccess) {
        (void)hipFree(dd);
        return fail(ctx, PCL_ENOMEM, "pcl_deriv_eval_jac: device allocation failed");
    }
    int rc = PCL_OK;
    if (hipMemcpyAsync(ctx->dZ, Z, z_len(ctx) * sizeof(double), hipMemcpyHostToDevice, ctx->stream) != hipSuccess) rc = PCL_EHIP;
    if (rc == PCL_OK) rc = pcl_deriv_eval_jac_dev(ctx, x_off, dx_off, dim, ctx->dZ, dd, dv);
    if (rc == PCL_OK && delta && hipMemcpyAsync(delta, dd, (size_t)nr * sizeof(double), hipMemcpyDeviceToHost, ctx->stream) != hipSuccess) rc = PCL_EHIP;
    if (rc == PCL_OK && vals && hipMemcpyAsync(vals, dv, (size_t)nz * sizeof(double), hipMemcpyDeviceToHost, ctx->stream) != hipSuccess) rc = PCL_EHIP;
    if (hipStreamSynchronize(ctx->stream) != hipSuccess && rc == PCL_OK) rc = PCL_EHIP;
    (void)hipFree(dd);
    (void)hipFree(dv);
    if (rc == PCL_EHIP) return fail(ctx, PCL_EHIP, "pcl_deriv_eval_jac: HIP error %s", hipGetErrorString(hipGetLastError()));
    return rc;
}

// --- terminal infidelity objective ----------------------------------------------
extern "C" int pcl_set_goal(pcl_ctx *ctx, const double *goal_iso_vec) {
    if (!ctx) return PCL_EINVAL;
    if (!goal_iso_vec) return fail(ctx, PCL_EINVAL, "pcl_set_goal: NULL");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (!ctx->dgoal) HIP_TRY(ctx, hipMalloc((void **)&ctx->dgoal, (size_t)ctx->x_dim * sizeof(double)));
    HIP_TRY(ctx, hipMemcpy(ctx->dgoal, goal_iso_vec, (size_t)ctx->x_dim * sizeof(double), hipMemcpyHostToDevice));
    return PCL_OK;
}
extern "C" int pcl_infidelity_dev(pcl_ctx *ctx, const double *Z, double Q, double *value, double *grad) {
    if (!ctx) return PCL_EINVAL;
    if (!Z || (!value && !grad)) return fail(ctx, PCL_EINVAL, "pcl_infidelity_dev: NULL pointer");
    if (!ctx->dgoal) return fail(ctx, PCL_EINVAL, "pcl_infidelity_dev: call pcl_set_goal first");
    if (ctx->cols != ctx->desc.d) return fail(ctx, PCL_ENOTIMPL, "pcl_infidelity_dev: unitary (n x d) states only");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const pcl_desc &D = ctx->desc;
    hipLaunchKernelGGL(pcl_infidelity_kernel, dim3((unsigned)D.batch), dim3(256), 0, ctx->stream, Z, ctx->dgoal, ctx->dxoffs, value, grad, Q,
                       D.d, D.N, D.z_dim, D.batch_mode == PCL_BATCH_TRAJ ? (long long)D.z_dim * D.N : 0LL);
    HIP_TRY(ctx, hipGetLastError());
    return PCL_OK;
}

// --- RCCL sum-reduce of the shared-control payload -----------------------------
// librccl is opened lazily with dlopen so that single-GPU users never load it.  ncclUniqueId is 128 opaque bytes;
// ncclDataType_t ncclFloat64 = 8, ncclRedOp_t ncclSum = 0 (rccl.h of ROCm 7.x).
namespace {
struct RcclApi {
    void *h = nullptr;
    int (*GetUniqueId)(void *) = nullptr;
    int (*CommInitRank)(void **, int, pcl_comm_id, int) = nullptr;
    int (*AllReduce)(const void *, void *, size_t, int, int, void *, hipStream_t) = nullptr;
    int (*CommDestroy)(void *) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
};
RcclApi g_rccl;
int rccl_load(const pcl_ctx *ctx) {
    if (g_rccl.h) return PCL_OK;
    void *h = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
    if (!h) return fail(ctx, PCL_ERCCL, "dlopen(librccl.so): %s", dlerror());
    RcclApi a;
    a.h = h;
    a.GetUniqueId = (int (*)(void *))dlsym(h, "ncclGetUniqueId");
    a.CommInitRank = (int (*)(void **, int, pcl_comm_id, int))dlsym(h, "ncclCommInitRank");
    a.AllReduce = (int (*)(const void *, void *, size_t, int, int, void *, hipStream_t))dlsym(h, "ncclAllReduce");
    a.CommDestroy = (int (*)(void *))dlsym(h, "ncclCommDestroy");
    a.GetErrorString = (const char *(*)(int))dlsym(h, "ncclGetErrorString");
    if (!a.GetUniqueId || !a.CommInitRank || !a.AllReduce || !a.CommDestroy) return fail(ctx, PCL_ERCCL, "librccl lacks the expected symbols");
    g_rccl = a;
    return PCL_OK;
}
int rccl_fail(const pcl_ctx *ctx, const char *what, int rc) {
    return fail(ctx, PCL_ERCCL, "%s: %s", what, g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "RCCL error");
}
}  // namespace

extern "C" int pcl_comm_get_unique_id(pcl_comm_id *out) {
    if (!out) return fail(nullptr, PCL_EINVAL, "pcl_comm_get_unique_id: NULL");
    int rc = rccl_load(nullptr);
    if (rc != PCL_OK) return rc;
    int nrc = g_rccl.GetUniqueId(out);
    return nrc == 0 ? PCL_OK : rccl_fail(nullptr, "ncclGetUniqueId", nrc);
}
extern "C" int pcl_comm_init(pcl_ctx *ctx, const pcl_comm_id *id, int32_t rank, int32_t nranks) {
    if (!ctx || !id) return PCL_EINVAL;
    if (nranks < 1 || rank < 0 || rank >= nranks) return fail(ctx, PCL_EINVAL, "pcl_comm_init: rank %d of %d", rank, nranks);
    if (ctx->comm) return fail(ctx, PCL_EINVAL, "pcl_comm_init: communicator already initialised");
    TRY(rccl_load(ctx));
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    int nrc = g_rccl.CommInitRank(&ctx->comm, nranks, *id, rank);
    if (nrc != 0) {
        ctx->comm = nullptr;
        return rccl_fail(ctx, "ncclCommInitRank", nrc);
    }
    return PCL_OK;
}
extern "C" int pcl_reduce_sum_dev(pcl_ctx *ctx, double *buf_dev, int64_t n) {
    if (!ctx) return PCL_EINVAL;
    if (!buf_dev || n < 0) return fail(ctx, PCL_EINVAL, "pcl_reduce_sum_dev: bad buffer");
    if (!ctx->comm) return fail(ctx, PCL_ERCCL, "pcl_reduce_sum_dev: call pcl_comm_init first");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    int nrc = g_rccl.AllReduce(buf_dev, buf_dev, (size_t)n, /*ncclFloat64*/ 8, /*ncclSum*/ 0, ctx->comm, ctx->stream);
    return nrc == 0 ? PCL_OK : rccl_fail(ctx, "ncclAllReduce", nrc);
}
extern "C" int pcl_comm_destroy(pcl_ctx *ctx) {
    if (!ctx) return PCL_EINVAL;
    if (ctx->comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(ctx->comm);
    ctx->comm = nullptr;
    return PCL_OK;
}

// --- options ---------------------------------------------------------------------------------
extern "C" int pcl_set_option(pcl_ctx *ctx, const char *key, int64_t v) {
    if (!ctx || !key) return PCL_EINVAL;
    if (!strcmp(key, "cols_per_slice")) {
        if (v < 0) return fail(ctx, PCL_EINVAL, "cols_per_slice must be >= 0");
        ctx->opt_cols_per_slice = v;
    } else if (!strcmp(key, "use_mfma"))
        ctx->opt_use_mfma = v != 0;
    else if (!strcmp(key, "nt_stores"))
        ctx->opt_nt = v != 0;
    else if (!strcmp(key, "debug_ablate"))  // profiling aid: results are WRONG when non-zero
        ctx->opt_ablate = v;
    else if (!strcmp(key, "grid"))
        ctx->opt_grid = v;
    else if (!strcmp(key, "copies_per_piece"))
        ctx->opt_cpp = v;
    else if (!strcmp(key, "specialize"))  // 0: always the run-time-shape kernel instances
        ctx->opt_specialize = v != 0;
    else if (!strcmp(key, "stream_piece_cols"))  // kernel 3, role split: > 0 = stream pieces of this many columns, round-robin
        ctx->opt_snc = v;
    else if (!strcmp(key, "aligned_stream"))  // kernel 3: 1 = line-aligned flat block stream, 0 = per-block stores from registers (default)
        ctx->opt_flat = v != 0;
    else if (!strcmp(key, "general_pade_kernel"))  // 1: the general-order kernel also for pade_order 4
        ctx->opt_general = v != 0;
    else if (!strcmp(key, "stream_workgroups"))  // kernel 3, contiguous: > 0 = role split with this many stream-role workgroups
        ctx->opt_stream_wg = v;
    else if (!strcmp(key, "contiguous"))  // kernel 3: 1 = equal contiguous column ranges per workgroup (default), 0 = round-robin slices
        ctx->opt_contig = v < 0 ? -1 : (v != 0);
    else if (!strcmp(key, "hess_kernel")) {  // 0: auto, 1: one workgroup per interval, 2: persistent wave-synchronous kernel
        if (v < 0 || v > 2) return fail(ctx, PCL_EINVAL, "hess_kernel must be 0, 1 or 2");
        ctx->opt_hess_kernel = v;
    }
    else if (!strcmp(key, "debug_timing")) {  // profiling aid: cycle stamps of workgroup 0 (pcl_debug_timing reads them)
        if (v && !ctx->ddbg) {
            HIP_TRY(ctx, hipMalloc((void **)&ctx->ddbg, 64 * sizeof(long long)));
            HIP_TRY(ctx, hipMemset(ctx->ddbg, 0, 64 * sizeof(long long)));
        } else if (!v && ctx->ddbg) {
            (void)hipFree(ctx->ddbg);
            ctx->ddbg = nullptr;
        }
    }
    else if (!strcmp(key, "kernel_version")) {
        if (v < 0 || v > 5) return fail(ctx, PCL_EINVAL, "kernel_version must be 0 (auto), 1, 2, 3, 4 or 5 (split)");
        ctx->opt_kernel = v;
    }
    else
        return fail(ctx, PCL_EINVAL, "unknown option '%s'", key);
    return PCL_OK;
}
extern "C" int pcl_debug_timing(pcl_ctx *ctx, int64_t *out, int64_t cap) {
    if (!ctx || !out || cap < 0) return PCL_EINVAL;
    if (!ctx->ddbg) return fail(ctx, PCL_EINVAL, "pcl_debug_timing: set option debug_timing first");
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    HIP_TRY(ctx, hipMemcpy(out, ctx->ddbg, (size_t)std::min<int64_t>(cap, 64) * sizeof(long long), hipMemcpyDeviceToHost));
    return PCL_OK;
}

extern "C" int pcl_get_option(const pcl_ctx *ctx, const char *key, int64_t *v) {
    if (!ctx || !key || !v) return PCL_EINVAL;
    if (!strcmp(key, "cols_per_slice"))
        *v = ctx->opt_cols_per_slice;
    else if (!strcmp(key, "use_mfma"))
        *v = ctx->opt_use_mfma;
    else if (!strcmp(key, "nt_stores"))
        *v = ctx->opt_nt;
    else if (!strcmp(key, "effective_cols_per_slice"))
        *v = ((ctx->opt_kernel == 3 || ctx->opt_kernel == 0) && ctx->opt_use_mfma && v3_supported(ctx) && ctx->cols == ctx->desc.d)
                 ? (v3_contiguous(ctx) ? ctx->desc.d : choose_cols_v3(ctx))
                 : choose_cols_per_slice(ctx, true);
    else if (!strcmp(key, "n_cu"))
        *v = ctx->n_cu;
    else if (!strcmp(key, "kernel_version"))
        *v = ctx->opt_kernel;
    else if (!strcmp(key, "last_kernel"))
        *v = ctx->last_kernel;
    else if (!strcmp(key, "contiguous"))
        *v = ctx->opt_contig;
    else if (!strcmp(key, "stream_workgroups"))
        *v = ctx->opt_stream_wg;
    else if (!strcmp(key, "last_stream_workgroups"))
        *v = ctx->last_n_stream;
    else if (!strcmp(key, "hess_kernel"))
        *v = ctx->opt_hess_kernel;
    else if (!strcmp(key, "last_hess_kernel"))
        *v = ctx->last_hess_kernel;
    else if (!strcmp(key, "ell_width_t"))
        *v = ctx->ellt_w;
    else if (!strcmp(key, "drives_antisymmetric"))
        *v = ctx->drives_antisym;
    else if (!strcmp(key, "occupancy_v2")) {
        KParams p;
        fill_params(ctx, p);
        p.nc = choose_cols_per_slice(ctx, true);
        const size_t lds = fused2_lds_bytes(p, true, ell_fits_lds(ctx));
        int nb = 0;
        (void)hipFuncSetAttribute((const void *)pcl_fused_kernel_v2<true, 1, 0, 0, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, pcl_fused_kernel_v2<true, 1, 0, 0, 0>, 512, lds) != hipSuccess) nb = -1;
        *v = nb * 1000000LL + (long long)lds;
    }
    else if (!strcmp(key, "iso_structured"))
        *v = ctx->iso;
    else if (!strcmp(key, "ell_width"))
        *v = ctx->ell_w;
    else if (!strcmp(key, "union_width"))
        *v = ctx->uell_w;
    else
        return fail(ctx, PCL_EINVAL, "unknown option '%s'", key);
    return PCL_OK;
}
