// Bare store patterns against the placement of the buffer: (a) grid-stride fill, (b) one contiguous range per workgroup, (c) two streams per
// workgroup 630 KB apart inside 1.34 MB records (the fused kernel's), (d) block-cyclic chunks of C bytes per workgroup.
// hipcc --offload-arch=gfx950 -O3 -o lab/probes/wpattern lab/probes/wpattern.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef double d2 __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(256) void fill_stride(d2 *p, size_t n2) {
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n2; i += (size_t)gridDim.x * 256) p[i] = d2{1.0, 2.0};
}
__global__ __launch_bounds__(256) void fill_range(d2 *p, size_t n2) {
    const size_t lo = n2 * blockIdx.x / gridDim.x, hi = n2 * (blockIdx.x + 1) / gridDim.x;
    for (size_t i = lo + threadIdx.x; i < hi; i += 256) p[i] = d2{1.0, 2.0};
}
__global__ __launch_bounds__(256) void fill_cyclic(d2 *p, size_t n2, size_t chunk2) {  // chunk2 d2 elements per chunk, chunks dealt round-robin
    const size_t nch = (n2 + chunk2 - 1) / chunk2;
    for (size_t c = blockIdx.x; c < nch; c += gridDim.x) {
        const size_t lo = c * chunk2, hi = std::min(n2, lo + chunk2);
        for (size_t i = lo + threadIdx.x; i < hi; i += 256) p[i] = d2{1.0, 2.0};
    }
}
template <int U, int NT>
__global__ __launch_bounds__(NT) void fill_stride_u(d2 *p, size_t n2) {  // U consecutive 16-byte stores per thread, tiles dealt grid-stride
    const size_t tile = (size_t)NT * U;
    for (size_t t0 = blockIdx.x * tile; t0 < n2; t0 += (size_t)gridDim.x * tile) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const size_t i = t0 + (size_t)u * NT + threadIdx.x;
            if (i < n2) p[i] = d2{1.0, 2.0};
        }
    }
}
template <int U, int NT>
__global__ __launch_bounds__(NT) void fill_range_u(d2 *p, size_t n2) {  // one contiguous range per workgroup, U stores in flight per thread
    const size_t lo = n2 * blockIdx.x / gridDim.x, hi = n2 * (blockIdx.x + 1) / gridDim.x;
    for (size_t t0 = lo; t0 < hi; t0 += (size_t)NT * U) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const size_t i = t0 + (size_t)u * NT + threadIdx.x;
            if (i < hi) p[i] = d2{1.0, 2.0};
        }
    }
}
template <int U, int NT>
__global__ __launch_bounds__(NT) void fill_tile(d2 *p, size_t n2) {  // ONE tile of NT U elements per workgroup, no loop (a fill as torch launches it)
    const size_t t0 = (size_t)blockIdx.x * NT * U;
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const size_t i = t0 + (size_t)u * NT + threadIdx.x;
        if (i < n2) p[i] = d2{1.0, 2.0};
    }
}
template <int NT>
__global__ __launch_bounds__(NT) void fill_seg(d2 *p, size_t n2, size_t seg2) {  // one contiguous segment of seg2 elements per workgroup (not persistent)
    const size_t lo = (size_t)blockIdx.x * seg2, hi = std::min(n2, lo + seg2);
    for (size_t i = lo + threadIdx.x; i < hi; i += NT) p[i] = d2{1.0, 2.0};
}
int main(int argc, char **argv) {
    const size_t bytes = 1072916736ull;  // 8 trajectories of config 3
    const size_t n2 = bytes / 16;
    const int nbuf = argc > 1 ? atoi(argv[1]) : 10;
    std::vector<d2 *> bufs(nbuf);
    for (auto &b : bufs) hipMalloc(&b, bytes);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    auto timeit = [&](auto launch) {
        for (int i = 0; i < 2; ++i) launch();
        hipDeviceSynchronize();
        float best = 1e30f;
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            for (int i = 0; i < 10; ++i) launch();
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            best = std::min(best, ms / 10 * 1000);
        }
        return best;
    };
    printf("buffer   memset  strideU4x256(4096wg)  strideU4x1024(2048wg)  strideU8x512(1024wg)  rangeU4x256(256wg)  rangeU4x1024(256wg)  rangeU8x256(1024wg) | stride-1024wg  range-256wg  range-512wg  cyclic-64K  cyclic-1M  cyclic-4M\n");
    for (int b = 0; b < nbuf; ++b) {
        d2 *p = bufs[b];
        {
            const float a1 = timeit([&] { fill_tile<4, 256><<<(unsigned)((n2 + 1023) / 1024), 256>>>(p, n2); });
            const float a2 = timeit([&] { fill_tile<8, 256><<<(unsigned)((n2 + 2047) / 2048), 256>>>(p, n2); });
            const float a3 = timeit([&] { fill_tile<4, 1024><<<(unsigned)((n2 + 4095) / 4096), 1024>>>(p, n2); });
            const size_t seg2 = 629856 / 16;  // one -B^+ / B^- segment of an interval
            const float b1 = timeit([&] { fill_seg<256><<<(unsigned)((n2 + seg2 - 1) / seg2), 256>>>(p, n2, seg2); });
            const float b2 = timeit([&] { fill_seg<1024><<<(unsigned)((n2 + seg2 - 1) / seg2), 1024>>>(p, n2, seg2); });
            const float b3 = timeit([&] { fill_seg<1024><<<(unsigned)((n2 + 2 * seg2 - 1) / (2 * seg2)), 1024>>>(p, n2, 2 * seg2); });
            printf("%4d  tile4x256 %6.1f tile8x256 %6.1f tile4x1024 %6.1f | seg630K x256thr %6.1f x1024thr %6.1f seg1.26M x1024thr %6.1f\n", b, a1, a2, a3, b1, b2, b3);
        }
        const float m0 = timeit([&] { hipMemsetAsync(p, 0, bytes, 0); });
        const float s1 = timeit([&] { fill_stride_u<4, 256><<<4096, 256>>>(p, n2); });
        const float s2 = timeit([&] { fill_stride_u<4, 1024><<<2048, 1024>>>(p, n2); });
        const float s3 = timeit([&] { fill_stride_u<8, 512><<<1024, 512>>>(p, n2); });
        const float r1 = timeit([&] { fill_range_u<4, 256><<<256, 256>>>(p, n2); });
        const float r2 = timeit([&] { fill_range_u<4, 1024><<<256, 1024>>>(p, n2); });
        const float r3 = timeit([&] { fill_range_u<8, 256><<<1024, 256>>>(p, n2); });
        printf("%4d   %7.1f %12.1f %20.1f %20.1f %18.1f %18.1f %18.1f     |", b, m0, s1, s2, s3, r1, r2, r3);
        const float t0 = timeit([&] { fill_stride<<<1024, 256>>>(p, n2); });
        const float t1 = timeit([&] { fill_range<<<256, 256>>>(p, n2); });
        const float t2 = timeit([&] { fill_range<<<512, 256>>>(p, n2); });
        const float t3 = timeit([&] { fill_cyclic<<<256, 256>>>(p, n2, 65536 / 16); });
        const float t4 = timeit([&] { fill_cyclic<<<256, 256>>>(p, n2, (1 << 20) / 16); });
        const float t5 = timeit([&] { fill_cyclic<<<256, 256>>>(p, n2, (4 << 20) / 16); });
        printf(" %8.1f     %8.1f     %8.1f     %8.1f   %8.1f   %8.1f\n", t0, t1, t2, t3, t4, t5);
    }
    return 0;
}
