/*
 * bev_raster.h — both BEV rasters from the code lists (and from a dense code array for the single-cloud entry points)
 * Part of the device code of libbev_mi355x.so; included by bev_kernels.hip only (one translation unit).
 */
#ifndef BEV_RASTER_H
#define BEV_RASTER_H

#include "bev_dev.h"

namespace bevk {
using namespace bevx;

/* ------------------------------------------------------------------------- */
/* Both rasters (BatchMultiBevGen.cpp:271-292 occupancy, 24 layers; :340-356 uint8 max height), one workgroup per
 * (frame, x-band of the images).  The band's 24-bit layer masks and max heights live in LDS (two planes of rows x M
 * words); its input are this band's code lists: one per strip from the walk (slots that are not candidates) and one
 * per part from k_ground_resolve (un-grounded candidates).  Finished planes leave with 16-byte stores, 1 KiB
 * contiguous per wave-instruction. */
constexpr int kRasterTailWords = kMaxStrips + kResolveParts + 2; /* behind the planes: list_end, over_l */
int raster_bands_for(int M) /* uniform bands whose two LDS planes fit; the coarse band height is M / this */
{
    for (int bands = kRasterSplit; bands <= kMaxBands; bands *= 2)
        if (M % bands == 0 && ((size_t)2 * (M / bands) * M + kRasterTailWords) * sizeof(uint32_t) <= (size_t)kRasterLdsCap) return bands;
    return 0;
}
size_t raster_lds_bytes(const Geometry &g) /* the band's two planes + the list prefix behind them */
{
    return ((size_t)2 * g.rp.coarse * g.rp.mat_size + kRasterTailWords) * sizeof(uint32_t);
}

/* both 16-bit halves of v shifted left by the halves of sh (v_pk_lshlrev_b16) */
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pk_shl16(uint32_t v, uint32_t sh)
{
    const u16x2 r = __builtin_bit_cast(u16x2, v) << __builtin_bit_cast(u16x2, sh);
    return __builtin_bit_cast(uint32_t, r);
}
/* one code into the band's LDS planes (the code is known to lie in the band) */
__device__ __forceinline__ void splat_code(uint32_t c, int x0, int M, uint32_t *mask, uint32_t *hmax)
{
    const int idx = (code_x(c) - x0) * M + code_y(c);
    atomicMax(&hmax[idx], (uint32_t)code_h(c));      /* :353-355 */
    const uint32_t l = code_layer(c);
    if (l != kNoLayer) atomicOr(&mask[idx], 1u << l); /* :289-291 */
}

/* the band's planes -> the two images (rows x0 .. x0 + band_rows of every layer) */
__device__ __forceinline__ void store_planes(const uint32_t *mask, const uint32_t *hmax, uint8_t *multi, uint8_t *single,
                                             int f, int x0, int band_rows, int M, int L, int tid, int nthreads)
{
    const int chunks_per_row = M / 16;
    const int n_tasks = band_rows * chunks_per_row;
    const size_t plane = (size_t)M * M;
    for (int task = tid; task < n_tasks; task += nthreads) {
        const int row = task / chunks_per_row, ch = task - row * chunks_per_row;
        const int base = row * M + ch * 16;
        const size_t out_off = (size_t)(x0 + row) * M + (size_t)ch * 16;
        if (single) {
            uint32_t w[4];
#pragma unroll
            for (int q = 0; q < 4; ++q)
                w[q] = hmax[base + 4 * q] | (hmax[base + 4 * q + 1] << 8) | (hmax[base + 4 * q + 2] << 16) |
                       (hmax[base + 4 * q + 3] << 24);
            *reinterpret_cast<uint4 *>(single + (size_t)f * plane + out_off) = make_uint4(w[0], w[1], w[2], w[3]);
        }
        if (multi) {
            uint32_t mk[16];
#pragma unroll
            for (int q = 0; q < 16; ++q) mk[q] = mask[base + q];
            uint8_t *mout = multi + (size_t)f * plane * L + out_off;
            /* byte = 255 where bit l of the cell's mask is set.  v_perm_b32's selectors 8..11 replicate bit 15 / bit 31
             * of its two sources over a byte: with the low halves of two cells' masks side by side in one word (cells a | b
             * << 16, and c | d << 16), one packed 16-bit shift per word brings layer l to bits 15 and 31 and ONE permute
             * writes the four cells' bytes: 3 instructions per word (round 3: 4 bit-field extracts + 3 permutes).  Layers
             * 16 and up come from the masks' high halves the same way. */
            uint32_t plo[8], phi[8]; /* cells 2k | 2k+1 << 16: layers 0..15, layers 16..31 */
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                plo[k] = __builtin_amdgcn_perm(mk[2 * k + 1], mk[2 * k], 0x05040100u);
                phi[k] = __builtin_amdgcn_perm(mk[2 * k + 1], mk[2 * k], 0x07060302u);
            }
            for (int l = 0; l < L; ++l) {
                const uint32_t sh = (uint32_t)(15 - (l & 15)) * 0x00010001u; /* (wave-uniform) */
                uint32_t w[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const uint32_t s01 = pk_shl16(l < 16 ? plo[2 * q] : phi[2 * q], sh);
                    const uint32_t s23 = pk_shl16(l < 16 ? plo[2 * q + 1] : phi[2 * q + 1], sh);
                    w[q] = __builtin_amdgcn_perm(s23, s01, 0x0b0a0908u);
                }
                store_stream(reinterpret_cast<uint4 *>(mout + (size_t)l * plane), make_uint4(w[0], w[1], w[2], w[3]));
            }
        }
    }
}

/* x-band `band` of frame f; lds: raster_lds_bytes(g) bytes — the band's two planes, then kRasterTailWords words */
__device__ __forceinline__ void raster_body(uint32_t *lds, const BatchPtrs &b, const Geometry &g, const int f, const int band, int want_multi, int want_single)
{
    TL_BEGIN;
    uint32_t *list_end = lds + 2 * g.rp.coarse * g.rp.mat_size;   /* [kMaxStrips + kResolveParts + 1] inclusive prefix of this band's code-list lengths */
    uint32_t &over_l = list_end[kMaxStrips + kResolveParts + 1];   /* a writer had more codes for this band than its list holds */
    const int M = g.rp.mat_size, L = g.rp.n_layers, bands = g.raster_bands, E = g.emitters;
    const int x0 = raster_band_x0(band, g.rp), band_rows = raster_band_rows(band, g.rp);
    const int cells = band_rows * M;
    uint32_t *mask = lds;
    uint32_t *hmax = lds + cells;
    const int tid = threadIdx.x;
    PH_DECL;
    PH();

    /* round trip 1: the list lengths; the planes are zeroed meanwhile */
    uint32_t my_cnt = 0u;
    if (tid < E) my_cnt = b.ncode[((size_t)f * E + tid) * bands + band];
    for (int k = tid; k < 2 * cells; k += kRasterThreads) lds[k] = 0u;
    if (tid < E) list_end[tid + 1] = my_cnt;
    if (tid == 0) {
        list_end[0] = 0u;
        over_l = 0u;
    }
    lds_barrier();
    if (my_cnt > g.code_cap) over_l = 1u;
    if (tid == 0) /* few lists (13 for HDL_64E): a serial prefix */
        for (int e = 0; e < E; ++e) list_end[e + 1] += list_end[e];
    lds_barrier();
    const uint32_t over = over_l;
    PH();

    /* this band's lists as ONE index space, so that every load of the workgroup is requested at once.  Which list an index
     * falls in is settled per WAVE: its 64 consecutive indices start in one list (a cursor that only moves forward: the wave's
     * indices ascend from load to load) and cross a list end once in a while (then, and only then, the lanes compare).  Round
     * 4 compared every index with every list end — 15 compare-select pairs per code, four fifths of the kernel's vector
     * instructions. */
    if (!over) {
        constexpr int kU = 8;
        const uint32_t total = list_end[E];
        const uint32_t *fmain = b.code_main + (size_t)f * E * bands * g.code_stride;
        const uint32_t wave0 = (uint32_t)(tid & ~63), lane = (uint32_t)(tid & 63);
        int ue = 0; /* (wave-uniform) the list that holds the wave's first index of the current load, [ulo, uhi) */
        uint32_t ulo = 0u, uhi = __builtin_amdgcn_readfirstlane(list_end[1]);
        for (uint32_t i0 = 0; i0 < total; i0 += kU * kRasterThreads) {
            uint32_t c[kU];
#pragma unroll
            for (int k = 0; k < kU; ++k) {
                const uint32_t iw = i0 + (uint32_t)k * kRasterThreads + wave0; /* (wave-uniform) */
                c[k] = kSkip;
                if (iw >= total) continue;
                while (iw >= uhi && ue + 1 < E) { /* (also past empty lists) */
                    ++ue;
                    ulo = uhi;
                    uhi = __builtin_amdgcn_readfirstlane(list_end[ue + 1]);
                }
                const uint32_t i = iw + lane;
                int e = ue;
                uint32_t e0 = ulo;
                {   /* list ends inside the wave's 64 indices */
                    int ee = ue;
                    uint32_t nx = uhi;
                    while (ee + 1 < E && nx <= iw + 63u) {
                        ++ee;
                        const bool past = i >= nx;
                        e = past ? ee : e;
                        e0 = past ? nx : e0;
                        nx = __builtin_amdgcn_readfirstlane(list_end[ee + 1]);
                    }
                }
                if (i < total) c[k] = fmain[((size_t)e * bands + band) * g.code_stride + (i - e0)];
            }
            /* (the next turn's loads in flight while these codes are entered: measured, no faster) */
#pragma unroll
            for (int k = 0; k < kU; ++k)
                if (c[k] != kSkip) splat_code(c[k], x0, M, mask, hmax);
        }
    } else { /* (workgroup-uniform) the band's cells from the ordered, labelled cloud itself: every slot's code, as
              * bev_multi_bev / bev_single_bev compute it for an arbitrary cloud */
        const bev_point_t *cloud = b.ordered + (size_t)f * g.S;
        for (int i = tid; i < g.S; i += kRasterThreads) {
            const float4 a = *reinterpret_cast<const float4 *>(cloud + i);
            const uint32_t c = bev_code(a.x, a.y, a.z, (int)reinterpret_cast<const int16_t *>(cloud + i)[14], g.rp);
            if (c != kSkip && (uint32_t)(code_x(c) - x0) < (uint32_t)band_rows) splat_code(c, x0, M, mask, hmax);
        }
    }
    lds_barrier();
    PH();
    store_planes(mask, hmax, want_multi ? b.multi : nullptr, want_single ? b.single : nullptr, f, x0, band_rows, M, L, tid,
                 kRasterThreads);
    PH();
    TL_END(K_BEV_RASTER);
    PH_PRINT(band == 7 ? "raster7 setup codes stores" : "raster1 setup codes stores", tid == 0 && f == 100 && (band == 7 || band == 1));
}

__global__ __launch_bounds__(kRasterThreads) void k_bev_raster(BatchPtrs b, Geometry g, int nf, int want_multi, int want_single)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t lds_dyn[];
    /* the bands of a frame on ONE XCD (blocks b and b+8 share an L2), adjacent launch slots */
    const int xl = (int)blockIdx.x & 7, jj = (int)blockIdx.x >> 3;
    const int f = (jj / g.raster_bands) * 8 + xl, band = jj % g.raster_bands;
    if (f >= nf) return;
    raster_body(lds_dyn, b, g, f, band, want_multi, want_single);
}

/* rasters of ONE arbitrary cloud from a dense code array (bev_multi_bev / bev_single_bev): every band scans all codes */
__global__ __launch_bounds__(kRasterThreads) void k_bev_raster_dense(const uint32_t *__restrict__ codes, uint32_t n,
                                                                    uint8_t *__restrict__ multi, uint8_t *__restrict__ single,
                                                                    RasterParams rp)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    const int M = rp.mat_size, L = rp.n_layers;
    const int band = blockIdx.x, x0 = raster_band_x0(band, rp), band_rows = raster_band_rows(band, rp), tid = threadIdx.x;
    const int cells = band_rows * M;
    uint32_t *mask = lds, *hmax = lds + cells;
    for (int k = tid; k < 2 * cells; k += kRasterThreads) lds[k] = 0u;
    __syncthreads();
    constexpr int kU = 8;
    for (uint32_t i0 = 0; i0 < n; i0 += kU * kRasterThreads) {
        uint32_t c[kU];
#pragma unroll
        for (int k = 0; k < kU; ++k) {
            const uint32_t i = i0 + (uint32_t)k * kRasterThreads + tid;
            c[k] = i < n ? codes[i] : kSkip;
        }
#pragma unroll
        for (int k = 0; k < kU; ++k)
            if (c[k] != kSkip && code_x(c[k]) >= x0 && code_x(c[k]) < x0 + band_rows) splat_code(c[k], x0, M, mask, hmax);
    }
    __syncthreads();
    store_planes(mask, hmax, multi, single, 0, x0, band_rows, M, L, tid, kRasterThreads);
}

} /* namespace bevk */

#endif /* BEV_RASTER_H */
