// Round 5: does the ALIGNMENT of a wave's 1 KB store runs matter?  Kernel 4's block stores are flat 16-byte-per-lane runs (3,888 B per instruction of the four stream
// waves) whose start is wherever the block starts: a multiple of 16 B, of 128 B for one block in four.  Bare fill of 1.08 GB (8 trajectories of config 3) by 16-byte stores,
// every wave writing 1 KB runs that start `shift` bytes behind a 1 KB boundary; short-lived workgroups (64 KB each) or 256 persistent ones (4.2 MB ranges, as the static split).
//   hipcc --offload-arch=gfx950 -O3 -o lab/probes/align_probe lab/probes/align_probe.hip ; align_probe
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef double d2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// run: bytes one instruction of the workgroup's 256 threads covers (4096: flat; 3888: 243 of 256 lanes active, as the kernel)
__global__ __launch_bounds__(256) void fill(char *base, long long bytes, int shift, int run, long long per_wg) {
    const int tid = threadIdx.x;
    const long long lo = (long long)blockIdx.x * per_wg, hi = min(bytes - 4096, lo + per_wg);
    const d2 v = d2{(double)tid, 1.0};
    if (tid * 16 < run)
        for (long long o = lo; o < hi; o += run) *(d2 *)(base + shift + o + tid * 16) = v;
}
// flavours of the store: 0 plain | 1 nontemporal | 2 sc1 | 3 sc0 sc1 (write-through) | 4 plain, 64 B per lane (four consecutive 16-byte stores) | 5 nt, 64 B per lane
template <int NT>
__global__ __launch_bounds__(NT) void fill2(char *base, long long bytes, int flavour, long long per_wg) {
    const int tid = threadIdx.x;
    const long long lo = (long long)blockIdx.x * per_wg, hi = min(bytes - 16384, lo + per_wg);
    const d2 v = d2{(double)tid, 1.0};
    if (flavour < 4) {
        for (long long o = lo; o < hi; o += NT * 16) {
            d2 *q = (d2 *)(base + o + tid * 16);
            if (flavour == 0) *q = v;
            else if (flavour == 1) __builtin_nontemporal_store(v, q);
            else if (flavour == 2) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(q), "v"(v) : "memory");
            else asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(q), "v"(v) : "memory");
        }
    } else {
        for (long long o = lo; o < hi; o += NT * 64) {
            d2 *q = (d2 *)(base + o + tid * 64);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (flavour == 4) q[k] = v;
                else __builtin_nontemporal_store(v, q + k);
            }
        }
    }
}

int main() {
    const long long bytes = 8LL * 99 * (2LL * 27 * 54 * 54 + 54LL * 27 * 7) * 8;
    char *buf;
    CK(hipMalloc(&buf, bytes + 8192));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    auto time = [&](int shift, int run, long long per_wg) {
        const int grid = (int)((bytes + per_wg - 1) / per_wg);
        float best = 1e30f;
        for (int rep = 0; rep < 3; ++rep) {
            for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(fill, dim3(grid), dim3(256), 0, 0, buf, bytes, shift, run, per_wg);
            CK(hipEventRecord(e0));
            for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(fill, dim3(grid), dim3(256), 0, 0, buf, bytes, shift, run, per_wg);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            best = std::min(best, ms / 10 * 1e3f);
        }
        return best;
    };
    printf("%.3f GB per launch; us per launch (TB/s)\n", bytes / 1e9);
    for (long long per_wg : {65536LL, (bytes / 256 / 4096) * 4096}) {
        printf("-- %s workgroups (%lld B each)\n", per_wg == 65536 ? "short-lived" : "256 persistent", per_wg);
        for (int run : {4096, 3888})
            for (int shift : {0, 16, 32, 48, 64, 96, 112}) {
                const float us = time(shift, run, per_wg / run * run);
                printf("run %4d B  shift %3d B : %7.1f us  (%.2f TB/s)\n", run, shift, us, bytes / us / 1e6);
            }
    }
    const char *names[] = {"plain", "nontemporal", "sc1", "sc0 sc1", "plain 64 B/lane", "nt 64 B/lane"};
    auto time2 = [&](int nt, int flavour, long long per_wg) {
        const int grid = (int)((bytes + per_wg - 1) / per_wg);
        float best = 1e30f;
        for (int rep = 0; rep < 3; ++rep) {
            for (int i = 0; i < 12; ++i) {
                if (i == 2) CK(hipEventRecord(e0));
                if (nt == 256) hipLaunchKernelGGL(fill2<256>, dim3(grid), dim3(256), 0, 0, buf, bytes, flavour, per_wg);
                else if (nt == 512) hipLaunchKernelGGL(fill2<512>, dim3(grid), dim3(512), 0, 0, buf, bytes, flavour, per_wg);
                else hipLaunchKernelGGL(fill2<1024>, dim3(grid), dim3(1024), 0, 0, buf, bytes, flavour, per_wg);
            }
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            best = std::min(best, ms / 10 * 1e3f);
        }
        return best;
    };
    for (int nt : {256, 1024})
        for (long long per_wg : {65536LL, 1048576LL, (bytes / 256 / 65536) * 65536})
            for (int fl = 0; fl < 6; ++fl) {
                const float us = time2(nt, fl, per_wg);
                printf("%4d threads, %8lld B per workgroup, %-16s: %7.1f us  (%.2f TB/s)\n", nt, per_wg, names[fl], us, bytes / us / 1e6);
            }
    {
        float best = 1e30f;
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipMemsetAsync(buf, 1, bytes, 0));
            CK(hipEventRecord(e0));
            for (int i = 0; i < 10; ++i) CK(hipMemsetAsync(buf, 1, bytes, 0));
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            best = std::min(best, ms / 10 * 1e3f);
        }
        printf("hipMemsetAsync: %7.1f us  (%.2f TB/s)\n", best, bytes / best / 1e6);
    }
    return 0;
}
