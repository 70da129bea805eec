// Round 5: bare store pattern of a STRIPED group (config-3 record geometry, values from registers): groups of `gm` workgroups walk the intervals g, g + n_groups, ...;
// inside an interval member j writes stripe j -- 1458 / gm 16-byte units -- of EVERY one of the 54 blocks (one store instruction of <= 256 lanes per block), and its
// share of the tail run.  Against: members take slices of `cpi` state columns (whole blocks: what the slice tickets do, here in a static order), and equal contiguous
// ranges per workgroup (the static split).  front_groups_probe.hip says: 4 KB per workgroup per step inside a tight window runs at 6.3-6.6 TB/s, 32 KB at 5.5.
//   hipcc --offload-arch=gfx950 -O3 -o lab/probes/stripe_probe lab/probes/stripe_probe.hip ; stripe_probe [trajectories=8]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
typedef double d2 __attribute__((ext_vector_type(2)));
constexpr int D = 27, N = 54, NN = N * N, M = 6, NU = NN / 2;
constexpr long long BLK = (long long)D * NN, TAIL = (long long)N * D * (M + 1), JAC_PER = 2 * BLK + TAIL;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// mode 0: stripes | 1: slices of cpi columns, static round-robin over the members | 2: equal contiguous column ranges per workgroup
// order (mode 0): 0 all -B+ blocks then all B- | 1 column by column (-B+ then B- of the column)
__global__ __launch_bounds__(256) void emu(double *jac, int n_int, int mode, int gm, int cpi, int order) {
    const int tid = threadIdx.x, w = blockIdx.x, G = gridDim.x;
    const d2 vp = d2{(double)tid, 1.0}, vm = d2{-1.0, (double)tid};
    if (mode == 2) {
        const long long tot = (long long)n_int * D, lo = tot * w / G, hi = tot * (w + 1) / G;
        const int pi = 2 * (tid % 27), pj0 = tid / 27;
        for (long long c = lo; c < hi; ++c) {
            const long long bk = c / D;
            const int cq = (int)(c - bk * D);
            double *o = jac + bk * JAC_PER + (long long)cq * NN;
            if (pj0 < 9) {
                for (int r = 0; r < 6; ++r) *(d2 *)(o + N * (pj0 + 9 * r) + pi) = vp;
                for (int r = 0; r < 6; ++r) *(d2 *)(o + BLK + N * (pj0 + 9 * r) + pi) = vm;
            }
            double *t = jac + bk * JAC_PER + 2 * BLK + (long long)cq * (M + 1) * N;
            for (int e2 = tid; e2 < (M + 1) * 27; e2 += 256) *(d2 *)(t + 2 * e2) = vp;
        }
        return;
    }
    const int n_groups = G / gm, g = w / gm, j = w % gm;
    for (int iv = g; iv < n_int; iv += n_groups) {
        double *rec = jac + (long long)iv * JAC_PER;
        if (mode == 0) {
            const int U = (NU + gm - 1) / gm, u = j * U + tid;
            if (tid < U && u < NU) {
                if (order == 0) {
                    for (int b = 0; b < D; ++b) *(d2 *)(rec + (long long)b * NN + 2 * u) = vp;
                    for (int b = 0; b < D; ++b) *(d2 *)(rec + BLK + (long long)b * NN + 2 * u) = vm;
                } else
                    for (int b = 0; b < D; ++b) {
                        *(d2 *)(rec + (long long)b * NN + 2 * u) = vp;
                        *(d2 *)(rec + BLK + (long long)b * NN + 2 * u) = vm;
                    }
            }
            const long long T2 = TAIL / 2, TU = (T2 + gm - 1) / gm;
            for (long long e2 = j * TU + tid; e2 < min(T2, (j + 1) * TU); e2 += 256) *(d2 *)(rec + 2 * BLK + 2 * e2) = vp;
        } else if (mode == 3) {
            // LINE-ALIGNED stripes: member j writes the lines [j L, (j + 1) L) of the block's span of 128-byte lines (L = ceil(lines / gm)): no line is shared
            // between two workgroups except the first and last of a block; the lane -> unit map shifts with the block's alignment (a = 0 .. 7 units)
            const int L = (NU / 8 + 2 + gm - 1) / gm;  // lines per member
            for (int sgn = 0; sgn < 2; ++sgn)
                for (int b = 0; b < D; ++b) {
                    double *blk = rec + sgn * BLK + (long long)b * NN;
                    const int a = (int)(((unsigned long long)blk >> 4) & 7);
                    for (int t = tid; t < 8 * L; t += 256) {
                        const int r = 8 * L * j - a + t;
                        if (r >= 0 && r < NU) *(d2 *)(blk + 2 * r) = sgn ? vm : vp;
                    }
                }
            double *tl = rec + 2 * BLK;
            const int a = (int)(((unsigned long long)tl >> 4) & 7);
            const int T2 = (int)(TAIL / 2), LT = (T2 / 8 + 2 + gm - 1) / gm;
            for (int t = tid; t < 8 * LT; t += 256) {
                const int r = 8 * LT * j - a + t;
                if (r >= 0 && r < T2) *(d2 *)(tl + 2 * r) = vp;
            }
        } else {
            const int pi = 2 * (tid % 27), pj0 = tid / 27;
            const int S = (D + cpi - 1) / cpi;
            for (int sl = j; sl < S; sl += gm) {
                const int c0 = sl * cpi, c1 = min(D, c0 + cpi);
                for (int cq = c0; cq < c1; ++cq) {
                    double *o = rec + (long long)cq * NN;
                    if (pj0 < 9) {
                        for (int r = 0; r < 6; ++r) *(d2 *)(o + N * (pj0 + 9 * r) + pi) = vp;
                        for (int r = 0; r < 6; ++r) *(d2 *)(o + BLK + N * (pj0 + 9 * r) + pi) = vm;
                    }
                    double *t = rec + 2 * BLK + (long long)cq * (M + 1) * N;
                    for (int e2 = tid; e2 < (M + 1) * 27; e2 += 256) *(d2 *)(t + 2 * e2) = vp;
                }
            }
        }
    }
}

int main(int argc, char **argv) {
    const int ntraj = argc > 1 ? atoi(argv[1]) : 8, nbuf = 4;
    const int n_int = ntraj * 99;
    const size_t bytes = (size_t)n_int * JAC_PER * 8;
    double *bufs[4];
    for (auto &b : bufs) CK(hipMalloc(&b, bytes));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    auto timeit = [&](int mode, int gm, int cpi, int order) {
        float tmin = 1e30f, tmax = 0.f;
        for (int bi = 0; bi < nbuf; ++bi) {
            float best = 1e30f;
            for (int rep = 0; rep < 2; ++rep) {
                for (int i = 0; i < 7; ++i) {
                    if (i == 2) CK(hipEventRecord(e0));
                    hipLaunchKernelGGL(emu, dim3(256), dim3(256), 0, 0, bufs[bi], n_int, mode, gm, cpi, order);
                }
                CK(hipEventRecord(e1));
                CK(hipEventSynchronize(e1));
                float ms;
                CK(hipEventElapsedTime(&ms, e0, e1));
                best = std::min(best, ms / 5 * 1e3f);
            }
            tmin = std::min(tmin, best), tmax = std::max(tmax, best);
        }
        printf("  %6.1f - %6.1f us (%.2f - %.2f TB/s)\n", tmin, tmax, bytes / tmax / 1e6, bytes / tmin / 1e6);
    };
    printf("%d trajectories, %.3f GB, 4 buffers: min - max\n", ntraj, bytes / 1e9);
    printf("equal contiguous ranges (static split)      :");
    timeit(2, 1, 0, 0);
    for (int cpi : {3, 4}) {
        printf("groups of 8, static slices of %d columns      :", cpi);
        timeit(1, 8, cpi, 0);
    }
    for (int gm : {6, 8, 16, 32})
        for (int order : {0}) {
            printf("groups of %2d, stripes, %s:", gm, order ? "column by column   " : "all -B+ then all B-");
            timeit(0, gm, 0, order);
        }
    for (int gm : {2, 4, 6, 8, 16, 32, 64}) {
        printf("groups of %2d, LINE-ALIGNED stripes          :", gm);
        timeit(3, gm, 0, 0);
    }
    return 0;
}
