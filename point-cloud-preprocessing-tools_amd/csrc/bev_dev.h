/*
 * bev_dev.h — device helpers shared by the kernels: block -> (frame, tile) mapping, cache-policy loads and stores, LDS-only barrier, LDS-DMA, counted waits, BEV codes with compile-time reciprocal choice
 * Part of the device code of libbev_mi355x.so; included by bev_kernels.hip only (one translation unit).
 */
#ifndef BEV_DEV_H
#define BEV_DEV_H

#include "bev_internal.h"
#include "bev_instr.h"

namespace bevk {
using namespace bevx;

/* Blocks are dealt round-robin over the 8 XCDs (b and b+8 share an L2).  The
 * gather kernel re-reads each point up to 3x (as itself, as the "upper" of the
 * row below, as the "lower" of the row above), so consecutive tiles of ONE
 * frame are given to ONE XCD: block b -> XCD lane x = b % 8, frame = 8*(j/tiles)
 * + x, tile = j % tiles with j = b / 8.  Placement only affects speed. */
__device__ __forceinline__ bool map_block_xcd(int b, int nf, int tiles, int &f, int &t)
{
    const int x = b & 7, j = b >> 3;
    const int fl = j / tiles;
    t = j - fl * tiles;
    f = fl * 8 + x;
    return f < nf;
}
static inline int xcd_grid(int nf, int tiles) { return 8 * ((nf + 7) / 8) * tiles; }

/* Cache policy.  The big streams of the path are touched ONCE by the kernel that moves them: the order scan's read of
 * the input, the walk's stores of the ordered cloud, codes and candidates, the raster's stores of the planes.  Issued
 * with the nontemporal hint (`nt`: stream through L2 / Infinity Cache instead of displacing lines that ARE reused —
 * winner table, candidate lists, codes between two kernels) the pipeline runs 6-9 % faster on the same box
 * (scripts/ab_libs.sh; the scan alone 1.2 -> 0.93 us per frame).  The walk's gather of the points is the exception:
 * `nt` loads there cost 12 % (halo columns and neighbouring strips re-read the same lines), so it keeps the default. */
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
template <class T>
__device__ __forceinline__ T load_once(const T *p) { return __builtin_nontemporal_load(p); }
template <class T>
__device__ __forceinline__ void store_stream(T *p, T v) { __builtin_nontemporal_store(v, p); }
struct alignas(16) Half { uint32_t w[4]; };
__device__ __forceinline__ void store_stream(Half *p, const Half &h)
{
    const u32x4 v = {h.w[0], h.w[1], h.w[2], h.w[3]};
    __builtin_nontemporal_store(v, reinterpret_cast<u32x4 *>(p));
}
__device__ __forceinline__ void store_stream(uint2 *p, uint2 a)
{
    const u32x2 v = {a.x, a.y};
    __builtin_nontemporal_store(v, reinterpret_cast<u32x2 *>(p));
}
__device__ __forceinline__ void store_stream(uint4 *p, uint4 a)
{
    const u32x4 v = {a.x, a.y, a.z, a.w};
    __builtin_nontemporal_store(v, reinterpret_cast<u32x4 *>(p));
}

/* Workgroup barrier for data exchanged through LDS ONLY.  `__syncthreads()` is a release / acquire fence over global
 * memory as well: with global stores (or LDS-DMA) pending, the compiler drains them — `s_waitcnt vmcnt(0)`, which on
 * gfx950 counts loads AND stores — before every barrier, so a loop with one barrier per step can keep nothing in flight
 * across steps.  The kernels below exchange only LDS words between their waves; nothing a wave writes to global memory
 * is read by another wave of the same launch. */
__device__ __forceinline__ void lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

/* A winner entry is (tag << shift) | (input index + 1).  The tag is the sub-batch generation of the workspace set:
 * entries left by earlier sub-batches carry a smaller tag, lose every atomicMax against the current one and read as
 * "empty", so the table needs no memset between sub-batches (bev_capi.hip clears it when the tag would wrap). */
__device__ __forceinline__ uint32_t winner_index(uint32_t w, uint32_t tag, int shift)
{
    return (w != 0u && (w >> shift) == tag) ? (w & ((1u << shift) - 1u)) : 0u;
}

template <class T> using gptr = __attribute__((address_space(1))) T *;
__device__ __forceinline__ uint32_t lds_addr(const void *p)
{
    return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void *)p;
}
/* LDS-DMA: every lane gives its own source address, the 64 x 16 (x 4) bytes land at a wave-uniform LDS address +
 * lane * 16 (* 4); counts on vmcnt like any load (scripts/microbench/glds_test.hip checks both on the box) */
__device__ __forceinline__ void glds16(const void *gsrc, uint32_t lds_dst)
{
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ void glds16x2(const void *ga, uint32_t la, const void *gb, uint32_t lb)
{
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\t"
                 "s_mov_b32 m0, %4\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %3, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(ga), "s"(la), "v"(gb), "s"(lb) : "memory");
}
__device__ __forceinline__ void glds4_nt(const void *gsrc, uint32_t lds_dst)
{
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off nt\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
template <int N>
__device__ __forceinline__ void wait_vm()
{ asm volatile("s_waitcnt vmcnt(%0)" : : "n"(N) : "memory"); }
/* a per-lane value as the row loop's body should see it: NOT loop-invariant, so that the lane predicates made from it
 * (lane < 4, lane & 2 ...) are compared afresh where they are used — hoisted out of the loop each of them is a pair of
 * scalar registers that the loop then spills and restores (k_walk<firing order>: 58 spilled scalars, 60 restores per step) */
__device__ __forceinline__ int fresh(int v)
{
    asm volatile("" : "+v"(v));
    return v;
}
/* a wave-uniform value that only feeds vector instructions: keep it out of the scalar file */
template <class T>
__device__ __forceinline__ T in_vgpr(T v)
{
    asm volatile("" : "+v"(v));
    return v;
}
/* bev_code / bev_code_from_bins (bev_exact.h) with the reciprocal / divide choice made at compile time: interval and
 * height_res are powers of two in every configuration of the reference, and x / 2^k == x * 2^-k bit for bit */
template <bool kPow2>
__device__ __forceinline__ uint32_t code_from_bins_t(int x, int y, float pz, const RasterParams &rp)
{
    const float hq = kPow2 ? pz * rp.inv_height_res : pz / rp.height_res;
    const int layer = cvtt_f32(roundf(hq + rp.lidar_to_ground)); /* BatchMultiBevGen.cpp:281 */
    int h = height_times4(pz + rp.lidar_to_ground);              /* :345 */
    h = h < 0 ? 0 : (h > 255 ? 255 : h);                         /* :346 */
    const uint32_t l = (layer >= 0 && layer < rp.n_layers) ? (uint32_t)layer : kNoLayer;
    return (uint32_t)x | ((uint32_t)y << 9) | ((uint32_t)h << 18) | (l << 26);
}
template <bool kPow2>
__device__ __forceinline__ uint32_t code_t(float px, float py, float pz, int label, const RasterParams &rp)
{
    const float sx = px + rp.max_range_f, sy = py + rp.max_range_f;
    int x, y; /* (bin_of_shifted: round_half_up_bin for a shifted coordinate, for callers that only want bins inside the image) */
    const bool inx = bin_of_shifted(kPow2 ? sx * rp.inv_interval : sx / rp.interval, rp.mat_size, &x); /* :279, :343 */
    const bool iny = bin_of_shifted(kPow2 ? sy * rp.inv_interval : sy / rp.interval, rp.mat_size, &y); /* :280, :344 */
    const bool in = (label != 0) & inx & iny; /* :285, :349 */
    const uint32_t code = code_from_bins_t<kPow2>(in ? x : 0, in ? y : 0, pz, rp);
    return in ? code : kSkip;
}

} /* namespace bevk */

#include "bev_instr_host.h"

#endif /* BEV_DEV_H */
