// Write probe on the Jacobian's layout: intervals of 1,341,360 B = [seg0: 27 blocks of 23,328 B][seg1: same][tail, skipped].
//   mode 0  block-wise: every 23,328-byte block is written from its own start (store instructions start at
//           block-relative offsets: 16-byte aligned, not line aligned) -- what the stream waves did so far
//   mode 1  flat: each 629,856-byte segment is one run; a partial head up to the next 128-byte line, then 4 KiB-aligned
//           workgroup iterations, then the tail
// usage: wstream2 <grid> <reps>
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double double2_t __attribute__((ext_vector_type(2)));
#define PER 1341360LL
#define SEG 629856LL
#define BLK 23328LL
__global__ __launch_bounds__(256) void wk(char *out, long long n_int, int mode) {
    const double2_t v = {1.0 + threadIdx.x, 2.0};
    const long long i0 = n_int * blockIdx.x / gridDim.x, i1 = n_int * (blockIdx.x + 1) / gridDim.x;
    for (long long it = i0; it < i1; ++it)
        for (int sg = 0; sg < 2; ++sg) {
            char *base = out + it * PER + sg * SEG;
            if (mode == 0) {
                for (int c = 0; c < 27; ++c)
                    for (long long o = threadIdx.x * 16LL; o < BLK; o += 4096) *(double2_t *)(base + c * BLK + o) = v;
            } else {
                const long long head = (128 - ((unsigned long long)base & 127)) & 127;  // bytes up to the next line
                if ((long long)threadIdx.x * 16 < head) *(double2_t *)(base + threadIdx.x * 16LL) = v;
                for (long long o = head + threadIdx.x * 16LL; o < SEG; o += 4096) *(double2_t *)(base + o) = v;
            }
        }
}
int main(int argc, char **argv) {
    const int grid = argc > 1 ? atoi(argv[1]) : 128;
    const int reps = argc > 2 ? atoi(argv[2]) : 20;
    const long long n_int = 792, bytes = n_int * PER;
    char *buf;
    if (hipMalloc((void **)&buf, bytes) != hipSuccess) return 1;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int mode = 0; mode < 2; ++mode) {
        for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(wk, dim3(grid), dim3(256), 0, 0, buf, n_int, mode);
        hipDeviceSynchronize();
        hipEventRecord(e0, 0);
        for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(wk, dim3(grid), dim3(256), 0, 0, buf, n_int, mode);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        printf("grid %d mode %s: %.1f us  %.2f TB/s\n", grid, mode ? "flat-aligned" : "block-wise", ms / reps * 1e3, n_int * 2 * SEG / (ms / reps * 1e-3) / 1e12);
    }
    return 0;
}
