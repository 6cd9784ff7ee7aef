// hip_be.h -- HIP back end of NftPlan (device memory, copies, kernel launches, event timers) and the
// kernel entry template.  The library is built from several translation units so that the ~200 kernel
// instantiations compile in parallel: HipBackend::run<K> is only DECLARED for the host logic
// (hip_backend.hip); each hip_kernels_*.hip defines FA_HIP_RUN_IMPL, sees the definition, and
// instantiates run<K> -- and with it kernel_entry<K> -- explicitly for its group of kernels.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <cstring>
#include <string>
#include <type_traits>
#include <vector>

#include "nft_api.h"
#include "nft_discspec.h"

extern thread_local std::string g_last_error;   // defined in hip_backend.hip

inline bool hip_ok(hipError_t e, const char *what)
{
    if (e == hipSuccess) return true;
    g_last_error = std::string(what) + ": " + hipGetErrorString(e);
    return false;
}

void *fa_pool_alloc(size_t bytes);   // hip_backend.hip
void fa_pool_free(void *p);

template <class K, class = void> struct min_waves_of { static constexpr int value = 1; };
template <class K> struct min_waves_of<K, std::void_t<decltype(K::MIN_WAVES)>> {
    static constexpr int value = K::MIN_WAVES;
};

#ifdef FA_HIP_RUN_IMPL
template <class K>
__global__ void __launch_bounds__(K::THREADS, min_waves_of<K>::value)
kernel_entry(const typename K::Params p)
{
    K::body(p);
}
#endif

struct HipBackend {
    // workgroups one launch of a segmented O(n^2) kernel should have (Aberth sweeps): a few rounds of the 256 CUs
    static constexpr size_t kTargetWorkgroups = 2048;
    hipStream_t stream = nullptr;
    bool failed = false;
    // stage timers
    static constexpr int kMarks = 4;
    hipEvent_t ev[kMarks] = {nullptr, nullptr, nullptr, nullptr};
    bool timing = false;
    // per-launch timers (bench.py's per-kernel roofline): an event pair around every launch while enabled
    struct LaunchRec { const char *name; hipEvent_t a, b; };
    std::vector<LaunchRec> launches;
    size_t launches_used = 0;
    bool launch_timing = false;

    // device memory comes from a per-device cache of released blocks (hip_backend.hip: fa_pool_*): the host-pointer
    // entry points allocate their work arrays per call, and hipMalloc / hipFree cost 0.1-1 ms each (hipFree also
    // waits for the device).  Blocks are reused in stream order on the null stream the host-pointer calls run on.
    void *alloc(size_t b)
    {
        void *p = fa_pool_alloc(b);
        if (!p) { failed = true; return nullptr; }
        return p;
    }
    void free(void *p) { if (p) fa_pool_free(p); }
    void h2d(void *d, const void *s, size_t b)
    {
        // pageable source: make the copy complete before the caller's buffer can go away
        if (!hip_ok(hipMemcpyAsync(d, s, b, hipMemcpyHostToDevice, stream), "hipMemcpyAsync(H2D)")) failed = true;
        if (!hip_ok(hipStreamSynchronize(stream), "hipStreamSynchronize")) failed = true;
    }
    void d2h(void *d, const void *s, size_t b)
    {
        if (!hip_ok(hipMemcpyAsync(d, s, b, hipMemcpyDeviceToHost, stream), "hipMemcpyAsync(D2H)")) failed = true;
    }
    void memset0(void *d, size_t b)
    {
        if (!hip_ok(hipMemsetAsync(d, 0, b, stream), "hipMemsetAsync")) failed = true;
    }
    int sync()
    {
        if (!hip_ok(hipStreamSynchronize(stream), "hipStreamSynchronize")) failed = true;
        return failed ? NFT_EC_OTHER : NFT_SUCCESS;
    }
    void mark(int i)
    {
        if (!timing) return;
        if (!ev[i]) (void)hipEventCreate(&ev[i]);
        (void)hipEventRecord(ev[i], stream);
    }
    double elapsed_ms(int i0, int i1) const
    {
        if (!timing || !ev[i0] || !ev[i1]) return -1.0;
        float ms = -1.f;
        if (hipEventElapsedTime(&ms, ev[i0], ev[i1]) != hipSuccess) return -1.0;
        return (double)ms;
    }
    void destroy_events()
    {
        for (int i = 0; i < kMarks; i++)
            if (ev[i]) { (void)hipEventDestroy(ev[i]); ev[i] = nullptr; }
        for (auto &r : launches) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
        launches.clear();
        launches_used = 0;
    }
    LaunchRec *next_launch_rec(const char *name)
    {
        if (launches_used == launches.size()) {
            LaunchRec r{name, nullptr, nullptr};
            if (hipEventCreate(&r.a) != hipSuccess || hipEventCreate(&r.b) != hipSuccess) return nullptr;
            launches.push_back(r);
        }
        LaunchRec *r = &launches[launches_used++];
        r->name = name;
        return r;
    }

    template <class K> void run(int gx, int gy, const typename K::Params &p);
};

#ifdef FA_HIP_RUN_IMPL
template <class K> void HipBackend::run(int gx, int gy, const typename K::Params &p)
{
        constexpr size_t lds = K::lds_bytes();
        static bool attr_done = false;  // one flag per kernel instantiation
        if (lds > 48 * 1024 && !attr_done) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&kernel_entry<K>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            attr_done = true;
        }
        if (gx <= 0 || gy <= 0) return;
        LaunchRec *rec = launch_timing ? next_launch_rec(__PRETTY_FUNCTION__) : nullptr;
        if (rec) (void)hipEventRecord(rec->a, stream);
        hipLaunchKernelGGL(kernel_entry<K>, dim3((unsigned)gx, (unsigned)gy), dim3(K::THREADS), lds,
                           stream, p);
        if (rec) (void)hipEventRecord(rec->b, stream);
        if (!hip_ok(hipGetLastError(), "kernel launch")) failed = true;
}
// explicit instantiation of the launcher (and thereby the kernel) for one kernel functor
#define FA_INST(...) template void HipBackend::run<__VA_ARGS__>(int, int, const typename __VA_ARGS__::Params &);
#endif
