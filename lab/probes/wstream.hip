// Write-bandwidth probe: the same 1.06 GB written with 16-byte stores by persistent workgroups in two address orders:
//   mode 0  "ranges"  workgroup w streams its own contiguous range            (what the stream-role workgroups do)
//   mode 1  "window"  at step t workgroup w writes chunk t*grid + w           (the whole GPU advances through memory together)
// usage: wstream <grid> <chunk_bytes> <reps>
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double double2_t __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(256) void wk(double2_t *out, long long n16, long long chunk16, int mode) {
    const long long nchunks = n16 / chunk16;
    const double2_t v = {1.0 + threadIdx.x, 2.0};
    if (mode == 0) {
        const long long c0 = nchunks * blockIdx.x / gridDim.x, c1 = nchunks * (blockIdx.x + 1) / gridDim.x;
        for (long long c = c0; c < c1; ++c)
            for (long long i = threadIdx.x; i < chunk16; i += 256) out[c * chunk16 + i] = v;
    } else {
        for (long long c = blockIdx.x; c < nchunks; c += gridDim.x)
            for (long long i = threadIdx.x; i < chunk16; i += 256) out[c * chunk16 + i] = v;
    }
}
int main(int argc, char **argv) {
    const int grid = argc > 1 ? atoi(argv[1]) : 128;
    const long long chunk = argc > 2 ? atoll(argv[2]) : 23328;
    const int reps = argc > 3 ? atoi(argv[3]) : 20;
    const long long bytes = 1062357120LL / chunk * chunk;
    double2_t *buf;
    if (hipMalloc((void **)&buf, bytes) != hipSuccess) return 1;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int mode = 0; mode < 2; ++mode) {
        for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(wk, dim3(grid), dim3(256), 0, 0, buf, bytes / 16, chunk / 16, mode);
        hipDeviceSynchronize();
        hipEventRecord(e0, 0);
        for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(wk, dim3(grid), dim3(256), 0, 0, buf, bytes / 16, chunk / 16, mode);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        printf("grid %d chunk %lld mode %s: %.1f us  %.2f TB/s\n", grid, chunk, mode ? "window" : "ranges", ms / reps * 1e3, bytes / (ms / reps * 1e-3) / 1e12);
    }
    return 0;
}
