// Round 5: the static split of a multi-trajectory launch takes 181-228 us per 8 seeds depending on where the values array's pages live.  Can the
// MAPPING of workgroups to ranges take the lottery out?  Bare store pattern (config-3 record geometry, values from registers, one persistent
// workgroup of 4 store waves per CU), equal contiguous column ranges, variants of who gets which range and in which order it is walked:
//   0 plain            workgroup w takes range w, first column to last (what the library launches)
//   1 rotated          ... starting (w * 37 mod 64) / 64 of the way into its range, wrapping round
//   2 zigzag           odd workgroups walk their range backwards
//   3 xcd-contiguous   range index = (w mod 8) * (grid / 8) + w / 8: the workgroups of one XCD cover one contiguous eighth of the array
//   4 xcd-contiguous + rotated
//   5 half-step        odd workgroups start in the middle of their range
//   6 bit-reversed     range index = bit reversal of w (neighbouring CUs far apart in the array)
// hipcc --offload-arch=gfx950 -O3 -o lab/probes/static_variants lab/probes/static_variants.hip ; static_variants [buffers=8] [trajectories=8]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef double d2 __attribute__((ext_vector_type(2)));
constexpr int D = 27, N = 54, NN = N * N, HN = 27, M = 6;
constexpr long long BLK = (long long)D * NN, JAC_PER = 2 * BLK + (long long)N * D * (M + 1);
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__global__ __launch_bounds__(256) void emu(double *jac, int n_int, int mode) {
    const int tid = threadIdx.x, w = blockIdx.x, G = gridDim.x;
    double bp[6][2], bm[6][2];
#pragma unroll
    for (int r = 0; r < 6; ++r) bp[r][0] = tid + r, bp[r][1] = -tid, bm[r][0] = 0.5 * tid, bm[r][1] = r;
    int ri = w;
    if (mode == 3 || mode == 4) ri = (w & 7) * (G / 8) + (w >> 3);
    if (mode == 6) {
        ri = 0;
        for (int b = 0; (1 << b) < G; ++b) ri |= ((w >> b) & 1) << (31 - __clz(G - 1) - b);
        if (ri >= G) ri = w;
    }
    const long long tot = (long long)n_int * D;
    const long long lo = tot * ri / G, hi = tot * (ri + 1) / G, len = hi - lo;
    long long start = 0;
    if (mode == 1 || mode == 4) start = len * ((w * 37) & 63) / 64;
    if (mode == 5 && (w & 1)) start = len / 2;
    const bool back = mode == 2 && (w & 1);
    const int pi = 2 * (tid % HN), pj0 = tid / HN;
    for (long long i = 0; i < len; ++i) {
        long long c = lo + (start + (back ? len - 1 - i : i)) % len;
        const long long bk = c / D;
        const int cq = (int)(c - bk * D);
        double *o = jac + bk * JAC_PER + (long long)cq * NN;
        if (pj0 < 9) {
#pragma unroll
            for (int r = 0; r < 6; ++r) {
                const int j = pj0 + 9 * r;
                *(d2 *)(o + N * j + pi) = d2{bp[r][0], bp[r][1]};
                *(d2 *)(o + BLK + N * j + pi) = d2{bm[r][0], bm[r][1]};
            }
        }
        double *t = jac + bk * JAC_PER + 2 * BLK + (long long)cq * (M + 1) * N;  // the column's share of the tail run
        for (int e2 = tid; e2 < (M + 1) * HN; e2 += 256) *(d2 *)(t + 2 * e2) = d2{1.0, 2.0};
    }
}

int main(int argc, char **argv) {
    const int nbuf = argc > 1 ? atoi(argv[1]) : 8, ntraj = argc > 2 ? atoi(argv[2]) : 8;
    const int n_int = ntraj * 99, reps = ntraj > 16 ? 4 : 10;
    const size_t bytes = (size_t)n_int * JAC_PER * 8;
    std::vector<double *> bufs(nbuf);
    for (auto &b : bufs) CK(hipMalloc(&b, bytes));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const char *names[] = {"plain", "rotated", "zigzag", "xcd-contiguous", "xcd-contiguous+rotated", "half-step", "bit-reversed"};
    std::vector<std::vector<float>> res(7);
    for (int bi = 0; bi < nbuf; ++bi)
        for (int round = 0; round < 2; ++round)
            for (int m_ = 0; m_ < 7; ++m_) {
                const int mode = round ? 6 - m_ : m_;
                for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(emu, dim3(256), dim3(256), 0, 0, bufs[bi], n_int, mode);
                CK(hipDeviceSynchronize());
                CK(hipEventRecord(e0));
                for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(emu, dim3(256), dim3(256), 0, 0, bufs[bi], n_int, mode);
                CK(hipEventRecord(e1));
                CK(hipEventSynchronize(e1));
                float ms;
                CK(hipEventElapsedTime(&ms, e0, e1));
                if (round == 0)
                    res[mode].push_back(ms / reps * 1e3f);
                else
                    res[mode][bi] = std::min(res[mode][bi], ms / reps * 1e3f);
            }
    printf("%d trajectories, %d buffers, us per launch (bare stores): min / median / max | per buffer\n", ntraj, nbuf);
    for (int m_ = 0; m_ < 7; ++m_) {
        std::vector<float> s = res[m_];
        std::sort(s.begin(), s.end());
        printf("%-24s %7.1f %7.1f %7.1f |", names[m_], s.front(), s[s.size() / 2], s.back());
        for (float v : res[m_]) printf(" %.0f", v);
        printf("\n");
    }
    return 0;
}
