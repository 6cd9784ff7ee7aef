// dev_compat.h -- the thin layer that lets the kernel bodies in this directory be compiled
//   (a) by hipcc for gfx950 (the product), and
//   (b) by g++ into the thread-per-lane CPU emulator under tests/emu/ (test infrastructure only,
//       used to check index arithmetic without a GPU; never linked into libfnft_amd.so).
#pragma once

#include <stddef.h>
#include <stdint.h>

#if defined(__HIPCC__)
// ------------------------------------------------------------------------------------------
#include <hip/hip_runtime.h>
#define FA_HD __host__ __device__ __forceinline__
#define FA_DEV __device__ __forceinline__
#define FA_KERNEL __global__
#define FA_SYNC() __syncthreads()
// Barrier for data exchanged through LDS only: waits for this wave's LDS operations and joins the
// workgroup barrier, but -- unlike __syncthreads(), whose release fence drains vmcnt when global
// stores are outstanding -- lets global loads and stores in flight stay in flight across it.
#define FA_SYNC_LDS() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#define FA_TID ((int)threadIdx.x)
#define FA_BID ((int)blockIdx.x)
#define FA_BID_Y ((int)blockIdx.y)
#define FA_BDIM ((int)blockDim.x)
#define FA_GDIM ((int)gridDim.x)
// dynamic LDS: one extern array per kernel translation unit
#define FA_LDS_DECL extern __shared__ __attribute__((aligned(16))) unsigned char fa_lds_raw[];
#define FA_LDS_PTR (fa_lds_raw)
FA_DEV void fa_atomic_max_u64(unsigned long long *p, unsigned long long v) { atomicMax(p, v); }
FA_DEV void fa_atomic_add_i32(int *p, int v) { atomicAdd(p, v); }
// slot in a list that the lanes with `want` append to (every lane of the wave calls this): ONE atomic per wave
FA_DEV int fa_wave_append_slot(int *cnt, bool want)
{
    const unsigned long long mask = __ballot(want);
    const int lane = (int)(threadIdx.x & 63);
    int base = 0;
    if (lane == 0) base = atomicAdd(cnt, __popcll(mask));
    base = __builtin_amdgcn_readfirstlane(base);
    return base + __popcll(mask & ((1ull << lane) - 1ull));
}
// max over the 64 lanes of the wave, then ONE atomic per wave (do not rely on the compiler's
// atomic optimizer: without the reduction 2^20 lanes hit a handful of addresses)
FA_DEV void fa_wave_atomic_max_f64bits(unsigned long long *p, double v)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v = fmax(v, __shfl_xor(v, off, 64));
    if ((threadIdx.x & 63) == 0) {
        union { double d; unsigned long long u; } cv;
        cv.d = v;
        atomicMax(p, cv.u);
    }
}
// max over the wave of the high dword of a non-negative double, then ONE atomicMax per wave
FA_DEV void fa_wave_atomic_max_hi32(unsigned *p, double v)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v = fmax(v, __shfl_xor(v, off, 64));
    if ((threadIdx.x & 63) == 0) atomicMax(p, (unsigned)__double2hiint(v));
}
// max of 64 consecutive u32 (one per lane, reduced across the wave); every lane of the wave must be active
FA_DEV unsigned fa_slots_max_u32(const unsigned *p)
{
    unsigned x = p[threadIdx.x & 63];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const unsigned y = (unsigned)__shfl_xor((int)x, off, 64);
        x = y > x ? y : x;
    }
    return (unsigned)__builtin_amdgcn_readfirstlane((int)x);
}
FA_DEV void fa_atomic_or_i32(int *p, int v) { atomicOr(p, v); }
// a counter in LDS that one wave advances and other waves of the workgroup watch (body_peel_leaf): release store /
// acquire load at workgroup scope, so what the writer stored before the counter is visible to who sees the counter
FA_DEV void fa_lds_publish(int *p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP); }
// the same for a writer whose earlier LDS stores come from the SAME wave: a wave's LDS instructions are executed in
// program order, so the counter cannot overtake them and the writer need not wait for them (the release store above
// puts an s_waitcnt lgkmcnt(0) -- one LDS round trip -- in front of the counter); the compiler is kept from reordering
FA_DEV void fa_lds_publish_inorder(int *p, int v)
{
    __asm__ volatile("" ::: "memory");
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __asm__ volatile("" ::: "memory");
}
FA_DEV int fa_lds_observe(const int *p) { return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP); }
// the instruction scheduler moves nothing across this point (hand-placed interleaving of a dependent chain with
// independent work: a wave issues in order, so what follows a stalled instruction waits with it)
FA_DEV void fa_sched_fence() { __builtin_amdgcn_sched_barrier(0); }
FA_DEV void fa_nap() { __builtin_amdgcn_s_sleep(2); }   // ~128 clocks off the issue slots while polling
// wave shuffles of doubles (body_peel_leaf: one wave per workgroup)
FA_DEV double fa_shfl(double v, int src) { return __shfl(v, src, 64); }
// value of lane `src` (the same for the whole wave, a compile-time or scalar index): v_readlane, no LDS crossbar
FA_DEV double fa_readlane(double v, int src)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
    return __hiloint2double(hi, lo);
}
// neighbour lanes of the whole wave by DPP (wave_shr:1 / wave_shl:1 of the gfx9 family): lane l gets lane l-1
// (up) or l+1 (down); the end lanes keep their own value (callers overwrite them)
FA_DEV double fa_shfl_up1(double v)
{
    const int lo = __builtin_amdgcn_update_dpp(__double2loint(v), __double2loint(v), 0x138, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(__double2hiint(v), __double2hiint(v), 0x138, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
FA_DEV double fa_shfl_down1(double v)
{
    const int lo = __builtin_amdgcn_update_dpp(__double2loint(v), __double2loint(v), 0x130, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(__double2hiint(v), __double2hiint(v), 0x130, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
// the same with zero for the last lane (bound_ctrl: lanes without a source read 0)
FA_DEV double fa_shfl_down1_z(double v)
{
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x130, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x130, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
// hardware reciprocals (v_rcp_f64 / v_rcp_f32): starting values, refine where accuracy matters
FA_DEV double fa_rcp_approx(double x) { return __builtin_amdgcn_rcp(x); }
FA_DEV float fa_rcp_approx_f32(float x) { return __builtin_amdgcn_rcpf(x); }
FA_DEV void fa_sincos(double x, double *s, double *c) { sincos(x, s, c); }
// start delay of part of a grid: n sleep periods of ~1024 clocks for the waves that are `late`
FA_DEV void fa_stagger(int n, bool late)
{
    if (late)
        for (int k = 0; k < n; k++) __builtin_amdgcn_s_sleep(16);
}
// value known to be the same in every lane of the wave: move it to scalar registers
FA_DEV double fa_uniform(double x)
{
    const int lo = __builtin_amdgcn_readfirstlane(__double2loint(x));
    const int hi = __builtin_amdgcn_readfirstlane(__double2hiint(x));
    return __hiloint2double(hi, lo);
}
#else
// ------------------------------------------------------------------------------------------
#include <barrier>
#include <cmath>
#define FA_HD inline
#define FA_DEV inline
#define FA_KERNEL
struct fa_emu_ctx {
    int tid, bid, bid_y, bdim, gdim;
    std::barrier<> *bar;
    unsigned char *lds;
};
extern thread_local fa_emu_ctx *fa_emu;
#define FA_SYNC() fa_emu->bar->arrive_and_wait()
#define FA_SYNC_LDS() fa_emu->bar->arrive_and_wait()
#define FA_TID (fa_emu->tid)
#define FA_BID (fa_emu->bid)
#define FA_BID_Y (fa_emu->bid_y)
#define FA_BDIM (fa_emu->bdim)
#define FA_GDIM (fa_emu->gdim)
#define FA_LDS_DECL
#define FA_LDS_PTR (fa_emu->lds)
#define __restrict__
FA_DEV void fa_atomic_max_u64(unsigned long long *p, unsigned long long v)
{
    unsigned long long cur = __atomic_load_n(p, __ATOMIC_RELAXED);
    while (cur < v && !__atomic_compare_exchange_n(p, &cur, v, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
}
FA_DEV void fa_atomic_add_i32(int *p, int v) { __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
FA_DEV int fa_wave_append_slot(int *cnt, bool want) { return want ? __atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED) : 0; }
FA_DEV void fa_wave_atomic_max_f64bits(unsigned long long *p, double v)
{
    union { double d; unsigned long long u; } cv;
    cv.d = v;
    fa_atomic_max_u64(p, cv.u);
}
FA_DEV void fa_wave_atomic_max_hi32(unsigned *p, double v)
{
    union { double d; unsigned long long u; } cv;
    cv.d = v;
    const unsigned hi = (unsigned)(cv.u >> 32);
    unsigned cur = __atomic_load_n(p, __ATOMIC_RELAXED);
    while (cur < hi && !__atomic_compare_exchange_n(p, &cur, hi, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
}
FA_DEV unsigned fa_slots_max_u32(const unsigned *p)
{
    unsigned x = 0u;
    for (int s = 0; s < 64; s++) x = p[s] > x ? p[s] : x;
    return x;
}
FA_DEV void fa_atomic_or_i32(int *p, int v) { __atomic_fetch_or(p, v, __ATOMIC_RELAXED); }
FA_DEV double fa_rcp_approx(double x) { return 1.0 / x; }
// the lane emulator runs no kernel that shuffles (body_peel_leaf is GPU-only): placeholders for the parser
FA_DEV double fa_shfl(double v, int) { return v; }
FA_DEV void fa_lds_publish(int *p, int v) { __atomic_store_n(p, v, __ATOMIC_RELEASE); }
FA_DEV void fa_lds_publish_inorder(int *p, int v) { __atomic_store_n(p, v, __ATOMIC_RELEASE); }
FA_DEV void fa_sched_fence() {}
FA_DEV int fa_lds_observe(const int *p) { return __atomic_load_n(p, __ATOMIC_ACQUIRE); }
FA_DEV void fa_nap() {}
FA_DEV double fa_readlane(double v, int) { return v; }
FA_DEV double fa_shfl_up1(double v) { return v; }
FA_DEV double fa_shfl_down1(double v) { return v; }
FA_DEV double fa_shfl_down1_z(double v) { return v; }
FA_DEV float fa_rcp_approx_f32(float x) { return 1.0f / x; }
FA_DEV void fa_sincos(double x, double *s, double *c) { ::sincos(x, s, c); }
FA_DEV double fa_uniform(double x) { return x; }
FA_DEV void fa_stagger(int, bool) {}
using std::exp;
using std::floor;
using std::fma;
using std::ldexp;
using std::log2;
using std::sqrt;
#endif
