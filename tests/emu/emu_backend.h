// emu_backend.h -- CPU lane emulator back end for NftPlan (TEST INFRASTRUCTURE ONLY).
// Every "lane" of a workgroup is an OS thread, __syncthreads() is a std::barrier, LDS is a heap
// block per workgroup.  It exists to check the index arithmetic of the kernel bodies in
// fnft_amd/csrc/nft_kernels.h without a GPU; it is never part of libfnft_amd.so.
#pragma once
#include <barrier>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "../../fnft_amd/csrc/dev_compat.h"

struct EmuBackend {
    static constexpr size_t kTargetWorkgroups = 8;   // every emulated workgroup is a set of host threads
    void *alloc(size_t b) { return std::calloc(1, b); }
    void free(void *p) { std::free(p); }
    void h2d(void *d, const void *s, size_t b) { std::memcpy(d, s, b); }
    void d2h(void *d, const void *s, size_t b) { std::memcpy(d, s, b); }
    void memset0(void *d, size_t b) { std::memset(d, 0, b); }
    int sync() { return 0; }
    void mark(int) {}

    template <class K> void run(int gx, int gy, const typename K::Params &p)
    {
        const int T = K::THREADS;
        const size_t lds = K::lds_bytes();
        std::vector<unsigned char> ldsbuf(lds + 64);
        for (int by = 0; by < gy; by++)
            for (int bx = 0; bx < gx; bx++) {
                std::barrier<> bar(T);
                std::vector<std::thread> th;
                th.reserve(T);
                for (int t = 0; t < T; t++)
                    th.emplace_back([&, t] {
                        fa_emu_ctx ctx{t, bx, by, T, gx, &bar, ldsbuf.data()};
                        fa_emu = &ctx;
                        K::body(p);
                        // lanes that return before a barrier would deadlock the others: the
                        // kernels never do that, but drop out of the barrier to be safe
                        bar.arrive_and_drop();
                    });
                for (auto &x : th) x.join();
            }
    }
};
