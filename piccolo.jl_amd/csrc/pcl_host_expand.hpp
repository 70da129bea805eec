// pcl_host_expand.hpp -- host side of the host-pointer entry points (pcl_eval_jac / pcl_jac): a small persistent thread pool
// that expands the COMPACT Jacobian values (unique -B^+ / B^- tiles + tails, 12.7 MB per config-3 evaluation) into the
// caller's full triplet-order array (132.8 MB) while the device-to-host copy of the next chunk of intervals is in flight.
// Over PCIe travel 17.5 MB instead of 134 MB; the 132.8 MB are produced by the host's own cores with streaming stores
// (no read-for-ownership traffic), i.e. at host memory write bandwidth instead of at PCIe bandwidth.  Host-only code.
#pragma once

#include <atomic>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>

namespace pcl_host {

typedef double v2d __attribute__((ext_vector_type(2)));
typedef double v4d __attribute__((ext_vector_type(4)));
typedef double v8d __attribute__((ext_vector_type(8)));

// dst[0..n) = src[0..n) with non-temporal stores (dst is written once and not read back by this library).  One instance per store width: the
// translation unit is built for the x86-64 baseline, where a 32-byte vector store becomes two 16-byte movntps (what rounds 2-4 ran: four stores
// per cache line); the instances below are compiled for AVX2 / AVX-512 (function target attributes) and picked once at run time -- one full
// 64-byte line per store on hosts that have AVX-512, two 32-byte halves with AVX2.
template <class V, int W>
static inline __attribute__((always_inline)) void stream_copy_impl(double *__restrict__ dst, const double *__restrict__ src, size_t n) {
    size_t i = 0;
    while (i < n && (reinterpret_cast<uintptr_t>(dst + i) & (sizeof(V) - 1))) {
        dst[i] = src[i];
        ++i;
    }
    for (; i + W <= n; i += W) {
        V v;
        __builtin_memcpy(&v, src + i, sizeof v);
        __builtin_nontemporal_store(v, reinterpret_cast<V *>(dst + i));
    }
    for (; i < n; ++i) dst[i] = src[i];
}
static void stream_copy_sse2(double *__restrict__ dst, const double *__restrict__ src, size_t n) { stream_copy_impl<v2d, 2>(dst, src, n); }
#if defined(__x86_64__) && !defined(__HIP_DEVICE_COMPILE__)
__attribute__((target("avx2"))) static void stream_copy_avx2(double *__restrict__ dst, const double *__restrict__ src, size_t n) { stream_copy_impl<v4d, 4>(dst, src, n); }
__attribute__((target("avx512f"))) static void stream_copy_avx512(double *__restrict__ dst, const double *__restrict__ src, size_t n) { stream_copy_impl<v8d, 8>(dst, src, n); }
#endif
typedef void (*stream_copy_fn)(double *__restrict__, const double *__restrict__, size_t);
// width: 0 the widest the host has | 16 | 32 | 64 bytes per store (option host_store_bytes; a width the host lacks falls back to the next one down)
static inline stream_copy_fn pick_stream_copy(int width, int *chosen) {
    stream_copy_fn f = stream_copy_sse2;
    int w = 16;
#if defined(__x86_64__) && !defined(__HIP_DEVICE_COMPILE__)
    __builtin_cpu_init();
    if ((width == 0 || width >= 32) && __builtin_cpu_supports("avx2")) f = stream_copy_avx2, w = 32;
    if ((width == 0 || width >= 64) && __builtin_cpu_supports("avx512f")) f = stream_copy_avx512, w = 64;
#endif
    if (chosen) *chosen = w;
    return f;
}

// Persistent workers.  run(): jobs 0..total-1; a job may start once `published` has passed its index (the caller publishes
// jobs as the data they need arrives) -- workers that run ahead of the data spin briefly.
class Pool {
  public:
    explicit Pool(int n_threads) {
        for (int i = 0; i < n_threads; ++i) th_.emplace_back([this] { worker(); });
    }
    ~Pool() {
        {
            std::lock_guard<std::mutex> lk(m_);
            stop_ = true;
            ++gen_;
        }
        cv_.notify_all();
        for (auto &t : th_) t.join();
    }
    int size() const { return (int)th_.size(); }
    void begin(long long total, std::function<void(long long)> fn) {
        std::unique_lock<std::mutex> lk(m_);
        idle_.wait(lk, [&] { return in_drain_ == 0; });  // no worker is still leaving the previous run
        fn_ = std::move(fn);
        total_ = total;
        next_.store(0, std::memory_order_relaxed);
        published_.store(0, std::memory_order_relaxed);
        done_.store(0, std::memory_order_relaxed);
        ++gen_;
        lk.unlock();
        cv_.notify_all();
    }
    void publish(long long upto) { published_.store(upto, std::memory_order_release); }
    void wait() {  // the calling thread works too
        drain();
        while (done_.load(std::memory_order_acquire) < total_) std::this_thread::yield();
    }

  private:
    void drain() {
        for (;;) {
            const long long i = next_.fetch_add(1, std::memory_order_relaxed);
            if (i >= total_) return;
            while (published_.load(std::memory_order_acquire) <= i) std::this_thread::yield();
            fn_(i);
            std::atomic_thread_fence(std::memory_order_seq_cst);  // streaming stores are globally visible before the job counts as done
            done_.fetch_add(1, std::memory_order_release);
        }
    }
    void worker() {
        unsigned long long seen = 0;
        for (;;) {
            {
                std::unique_lock<std::mutex> lk(m_);
                cv_.wait(lk, [&] { return gen_ != seen; });
                seen = gen_;
                if (stop_) return;
                ++in_drain_;  // fn_ / total_ of this run were written under the same lock
            }
            drain();
            {
                std::lock_guard<std::mutex> lk(m_);
                --in_drain_;
            }
            idle_.notify_all();
        }
    }
    std::condition_variable idle_;
    int in_drain_ = 0;
    std::vector<std::thread> th_;
    std::mutex m_;
    std::condition_variable cv_;
    unsigned long long gen_ = 0;
    bool stop_ = false;
    std::function<void(long long)> fn_;
    long long total_ = 0;
    std::atomic<long long> next_{0}, published_{0}, done_{0};
};

// One interval's compact block [-B^+ (nn) | B^- (nn) | tail] -> full block [-B^+ x cols | B^- x cols | tail]; `half`
// selects the -B^+ copies (0) or the B^- copies and the tail (1) so that two threads can share an interval.
static inline void expand_interval(double *__restrict__ full, const double *__restrict__ compact, int cols, long long nn, long long tail, int half, stream_copy_fn copy) {
    if (half == 0) {
        for (int c = 0; c < cols; ++c) copy(full + (long long)c * nn, compact, (size_t)nn);
    } else {
        for (int c = 0; c < cols; ++c) copy(full + (long long)(cols + c) * nn, compact + nn, (size_t)nn);
        copy(full + 2LL * cols * nn, compact + 2 * nn, (size_t)tail);
    }
}

// CPUs the cgroup grants this process (cpu.max = "quota period", cgroup v2; cfs_quota_us / cfs_period_us, v1); 0: no quota.  A container may see
// 256 hardware threads and be allowed 16 CPUs' worth of time: a team above the quota wins single calls and is throttled over a run.
static inline double cgroup_quota_cpus() {
    double q = 0.0, p = 0.0;
    char a[64] = {0};
    if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
        if (fscanf(f, "%63s %lf", a, &p) == 2 && strcmp(a, "max") != 0) q = atof(a);
        fclose(f);
    } else {
        if (FILE *g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) {
            if (fscanf(g, "%lf", &q) != 1) q = 0.0;
            fclose(g);
        }
        if (FILE *g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) {
            if (fscanf(g, "%lf", &p) != 1) p = 0.0;
            fclose(g);
        }
    }
    return (q > 0.0 && p > 0.0) ? q / p : 0.0;
}

}  // namespace pcl_host
