// Round 5: hipMemsetAsync writes 1.06 GB in 158 us (6.7 TB/s); fills whose workgroups own contiguous ranges (64 KB ... 4 MB) reach 182-218 us (align_probe.hip).
// Is it the ORDER?  Grid-strided fills (every step, the whole grid writes ONE contiguous window; the window walks through the array) from registers, and the
// same order as an EXPANDER: every 16-byte unit of the full values array copied from the compact array (unique tiles; 37 MB, cache-resident) -- config-3 geometry.
//   hipcc --offload-arch=gfx950 -O3 -o lab/probes/front_fill_probe lab/probes/front_fill_probe.hip ; front_fill_probe [trajectories=8]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef double d2 __attribute__((ext_vector_type(2)));
constexpr int D = 27, N = 54, NN = N * N, M = 6;
constexpr long long XD = (long long)N * D, CPER = 2 * NN + XD * (M + 1), FPER = 2LL * D * NN + XD * (M + 1);
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int NT>
__global__ __launch_bounds__(NT) void fill_strided(d2 *dst, long long units) {
    const d2 v = d2{(double)threadIdx.x, 1.0};
    const long long stride = (long long)gridDim.x * NT;
    for (long long u = (long long)blockIdx.x * NT + threadIdx.x; u < units; u += stride) dst[u] = v;
}
// window: the grid writes `win` consecutive units per step, workgroup w the w-th piece of it (win = grid x NT: fill_strided)
template <int NT>
__global__ __launch_bounds__(NT) void expand_strided(const double *__restrict__ compact, d2 *dst, long long units, int src_mode) {
    const long long stride = (long long)gridDim.x * NT;
    constexpr long long UPER = FPER / 2, BLK2 = (long long)D * NN / 2, NN2 = NN / 2;
    for (long long u = (long long)blockIdx.x * NT + threadIdx.x; u < units; u += stride) {
        const long long bk = u / UPER;
        const long long off = u - bk * UPER;  // unit inside the interval's record
        long long s;
        if (off < 2 * BLK2) {
            const long long sign = off >= BLK2 ? 1 : 0, r = off - sign * BLK2;
            s = sign * NN2 + r % NN2;
        } else
            s = NN + (off - 2 * BLK2);
        d2 v;
        if (src_mode == 0)
            v = d2{(double)s, 1.0};
        else
            v = *(const d2 *)(compact + bk * CPER + 2 * s);
        dst[u] = v;
    }
}

int main(int argc, char **argv) {
    const int ntraj = argc > 1 ? atoi(argv[1]) : 8;
    const long long n_bk = 99LL * ntraj, units = n_bk * FPER / 2;
    double *compact;
    d2 *full;
    CK(hipMalloc(&compact, n_bk * CPER * 8));
    CK(hipMalloc(&full, units * 16 + 65536));
    CK(hipMemset(compact, 0, n_bk * CPER * 8));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    auto timeit = [&](auto launch) {
        float best = 1e30f;
        for (int rep = 0; rep < 3; ++rep) {
            launch();
            launch();
            CK(hipEventRecord(e0));
            for (int i = 0; i < 10; ++i) launch();
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            best = std::min(best, ms / 10 * 1e3f);
        }
        return best;
    };
    const double gb = units * 16 / 1e9;
    printf("%d trajectories, %.3f GB of values, %.1f MB compact\n", ntraj, gb, n_bk * CPER * 8 / 1e6);
    float us = timeit([&] { CK(hipMemsetAsync(full, 1, units * 16, 0)); });
    printf("hipMemsetAsync                          : %7.1f us (%.2f TB/s)\n", us, gb / us * 1e3);
    for (int grid : {256, 512, 1024, 2048, 4096, 16384}) {
        us = timeit([&] { hipLaunchKernelGGL(fill_strided<256>, dim3(grid), dim3(256), 0, 0, full, units); });
        printf("fill, grid-strided, %5d x  256 threads : %7.1f us (%.2f TB/s)\n", grid, us, gb / us * 1e3);
    }
    for (int grid : {256, 512, 1024}) {
        us = timeit([&] { hipLaunchKernelGGL(fill_strided<1024>, dim3(grid), dim3(1024), 0, 0, full, units); });
        printf("fill, grid-strided, %5d x 1024 threads : %7.1f us (%.2f TB/s)\n", grid, us, gb / us * 1e3);
    }
    for (int mode : {0, 1})
        for (int grid : {256, 512, 1024, 2048, 4096}) {
            us = timeit([&] { hipLaunchKernelGGL(expand_strided<256>, dim3(grid), dim3(256), 0, 0, compact, full, units, mode); });
            printf("expand (%s), grid-strided, %5d x 256 : %7.1f us (%.2f TB/s)\n", mode ? "compact array" : "registers    ", grid, us, gb / us * 1e3);
        }
    for (int mode : {0, 1})
        for (int grid : {256, 512}) {
            us = timeit([&] { hipLaunchKernelGGL(expand_strided<1024>, dim3(grid), dim3(1024), 0, 0, compact, full, units, mode); });
            printf("expand (%s), grid-strided, %5d x 1024: %7.1f us (%.2f TB/s)\n", mode ? "compact array" : "registers    ", grid, us, gb / us * 1e3);
        }
    return 0;
}
