/*
 * bev_resolve.h — markGroundPoints phase C for the candidates
 * Part of the device code of libbev_mi355x.so; included by bev_kernels.hip only (one translation unit).
 */
#ifndef BEV_RESOLVE_H
#define BEV_RESOLVE_H

#include "bev_dev.h"

namespace bevk {
using namespace bevx;

/* ------------------------------------------------------------------------- */
/* markGroundPoints phase C for the candidates, BatchMultiBevGen.cpp:216-250.  A candidate that is higher than a
 * neighbour cell's average + 0.30 stops being ground ("hit"): it keeps / gets back its own label, and its BEV code —
 * rebuilt from key and height, bev_exact.h — is appended to a code list of the raster band it falls into, exactly like
 * the walk's lists (an LDS cursor per band, no global atomics): k_bev_raster reads both kinds the same way.  The walk
 * wrote each candidate's label for its guess (key bit kKeyPredBit); only wrong guesses are patched.
 * kResolveParts code lists per frame, a contiguous quarter of the segments each; kResolveWgs workgroups per frame (one:
 * the frame's tables — 3,750 averages, their neighbour minima, edge bins, band table — cost as much as a part's
 * candidates) walk kResolveParts / kResolveWgs parts each; a wave requests kResolveBatch segments (x 4 slices of 64
 * candidates) at a time. */
constexpr int kResolveBatch = 4;
/* (round 3: the raster constants in vector registers and stores through address-space-1 pointers, as in the walk — a
 * quarter of this kernel's vector instructions were v_readlane restores of spilled 8-dword argument tuples) */
struct ResolveLds {
    /* Per cell, the LOWEST of its in-range 4-neighbours' averages (see resolve_body) */
    float minavg[kGridCells];
    float avg[kGridCells];                               /* the frame's 75 x 50 averages */
    int edge_x[kGridRows], edge_y[kGridCols];            /* BEV bin of every ground-grid row's / column's lower edge */
    uint32_t band_cursor[kMaxBands];
    /* per wave: the un-grounded candidates waiting for their BEV codes (key, height, slot): an eighth of a frame's candidates
     * is un-grounded, five to ten lanes of a slice — coded slice by slice, the code path (sixty-odd instructions) ran for
     * every slice with those few lanes alive; parked here and coded 64 at a time it runs an eighth as often */
    uint32_t ring[kResolveThreads / 64][3][128];
    uint16_t cnt[kMaxSegs / kResolveParts + 8];
    uint8_t band_tab[512];                               /* x bin -> raster band */
};
static_assert(sizeof(ResolveLds) <= 32 * 1280, "a quarter of a CU's LDS");

/* workgroup `wg` (0 .. kResolveWgs - 1) of frame f */
template <bool kPow2>
__device__ __forceinline__ void resolve_body(char *arena, const BatchPtrs &b, const Geometry &g, const int f, const int wg)
{
    TL_BEGIN;
    /* Per cell, the LOWEST of its in-range 4-neighbours' averages: "any neighbour n with fl(z - avg[n]) >= 0.3f" is
     * "fl(z - min_n avg[n]) >= 0.3f" — fl(z - a) does not increase with a, and the minimum passes over NaN averages exactly
     * as the comparisons do (a difference with a NaN is never >= 0.3f).  One look-up and one subtraction per candidate
     * instead of four of each with their range tests (bev_exact.h above_neighbour_ground, BatchMultiBevGen.cpp:227-241). */
    ResolveLds &lds_r = *reinterpret_cast<ResolveLds *>(arena);
    auto &minavg = lds_r.minavg;
    auto &avg = lds_r.avg;
    auto &edge_x = lds_r.edge_x;
    auto &edge_y = lds_r.edge_y;
    auto &band_cursor = lds_r.band_cursor;
    auto &band_tab = lds_r.band_tab;
    auto &cnt = lds_r.cnt;
    auto &ring = lds_r.ring;
    constexpr int kPartsPerWg = kResolveParts / kResolveWgs;
    const int part0 = wg * kPartsPerWg;
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int T = g.segs;
    for (int c = tid; c < kCells; c += kResolveThreads) avg[c] = b.avg[(size_t)f * kCells + c];
    if (tid < kGridRows) edge_x[tid] = cell_edge_bin(tid, 75.0f, g.rp);
    else if (tid < kGridRows + kGridCols) edge_y[tid - kGridRows] = cell_edge_bin(tid - kGridRows, 50.0f, g.rp);
    if (tid < kMaxBands) band_cursor[tid] = 0u;
    for (int x = tid; x < g.rp.mat_size; x += kResolveThreads) band_tab[x] = (uint8_t)raster_band_of_nodiv(x, g.rp);
    lds_barrier();
    for (int c = tid; c < kCells; c += kResolveThreads) {
        const int sr = c / kGridCols, sc = c % kGridCols;
        float m = __uint_as_float(0x7fc00000u); /* NaN: no neighbour yet (fminf returns the other operand) */
        if (sr - 1 >= 0) m = fminf(m, avg[c - kGridCols]);
        if (sc + 1 < kGridCols) m = fminf(m, avg[c + 1]);
        if (sc - 1 >= 0) m = fminf(m, avg[c - 1]);
        if (sr + 1 < kGridRows) m = fminf(m, avg[c + kGridCols]);
        minavg[c] = m;
    }
    lds_barrier();

    constexpr int kSl = kSeg / 64;
    constexpr int kWaves = kResolveThreads / 64;
    const int bands = g.raster_bands, lo_row = g.N - g.G, H = g.H, strips = g.strips;
    const uint2 *fcand = b.cand + (size_t)f * T * kSeg; /* key | height */
    const uint32_t code_cap = g.code_cap;
    const gptr<uint16_t> flabel = (gptr<uint16_t>)(b.ordered + (size_t)f * g.S); /* label @28 of point i: [16 * i + 14] */
    const bev_point_t *fordered = b.ordered + (size_t)f * g.S;
    RasterParams rp = g.rp; /* the fields the BEV code needs, in vector registers */
    rp.max_range_f = in_vgpr(rp.max_range_f);
    rp.lidar_to_ground = in_vgpr(rp.lidar_to_ground);
    rp.mat_size = in_vgpr(rp.mat_size);
    rp.n_layers = in_vgpr(rp.n_layers);
    if (kPow2) {
        rp.inv_interval = in_vgpr(rp.inv_interval);
        rp.inv_height_res = in_vgpr(rp.inv_height_res);
    } else {
        rp.interval = in_vgpr(rp.interval);
        rp.height_res = in_vgpr(rp.height_res);
        rp.inv_interval = 0.0f;
        rp.inv_height_res = 0.0f;
    }
  for (int part = part0; part < part0 + kPartsPerWg; ++part) { /* one code list set per part */
    const int t0 = (int)((long long)T * part / kResolveParts), t1 = (int)((long long)T * (part + 1) / kResolveParts);
    const gptr<uint32_t> flist = (gptr<uint32_t>)(b.code_main + ((size_t)f * g.emitters + g.strips + part) * bands * (size_t)g.code_stride);
    if (part != part0) lds_barrier(); /* the previous part's cursors have been written out, its counts read */
    for (int i = tid; i < t1 - t0; i += kResolveThreads) {
        const uint32_t w = b.ncand[(size_t)f * T + t0 + i]; /* four byte-wide counts: the segment's runs by cell quarter, back to back */
        cnt[i] = (uint16_t)((w & 0xffu) + ((w >> 8) & 0xffu) + ((w >> 16) & 0xffu) + (w >> 24));
    }
    if (tid < kMaxBands) band_cursor[tid] = 0u;
    lds_barrier();
    uint32_t rhead = 0u, rtail = 0u; /* (wave-uniform) entries ever parked in / taken from this wave's ring */
    /* the codes of up to 64 parked candidates (lane l: entry tail + l), appended to this part's lists */
    auto emit = [&](uint32_t n_valid) {
        __builtin_amdgcn_wave_barrier(); /* (the entries were written by other lanes of this wave: LDS keeps a wave's order) */
        const uint32_t at = (rtail + (uint32_t)lane) & 127u;
        const uint32_t kk = ring[wv][0][at], idx = ring[wv][2][at];
        const float zz = __uint_as_float(ring[wv][1][at]);
        if ((uint32_t)lane < n_valid) {
            const int cell = (int)(kk & kKeyCellMask);
            uint32_t code;
            if (!candidate_key_escapes(kk)) {
                code = code_from_bins_t<kPow2>(edge_x[cell / kGridCols] + (int)((kk >> kKeyDxShift) & 3u),
                                               edge_y[cell % kGridCols] + (int)((kk >> kKeyDyShift) & 3u), zz, rp);
            } else { /* cell clamped or bins not next to the cell's edge: x, y from the point itself */
                const float4 a = *reinterpret_cast<const float4 *>(fordered + idx);
                code = code_t<kPow2>(a.x, a.y, a.z, 1 /* not 0: no kKeyNoCodeBit */, rp);
            }
            if (code != kSkip) {
                const int band = band_tab[code_x(code)];
                const uint32_t pos = atomicAdd(&band_cursor[band], 1u);
                flist[(uint32_t)band * g.code_stride + (pos < code_cap ? pos : code_cap - 1u)] = code;
            }
        }
        rtail += n_valid;
        __builtin_amdgcn_wave_barrier();
    };
    /* a wave's segments of the part, kResolveBatch at a time; the next batch's loads are in flight while this one is tested */
    uint32_t key_n[kResolveBatch][kSl];
    float z_n[kResolveBatch][kSl];
    auto request = [&](int s0) {
#pragma unroll
        for (int j = 0; j < kResolveBatch; ++j) {
            const int sg = s0 + j * kWaves;
            const int n = sg < t1 ? (int)cnt[sg - t0] : 0; /* wave-uniform */
#pragma unroll
            for (int k = 0; k < kSl; ++k) { /* whole slices, nothing but the loads inside the uniform test (see k_cell_sums) */
                /* (lanes past the segment's count read its last candidate again — a sector that moves anyway — instead of the
                 * stale entries behind it: a segment holds 140 candidates on average, three slices of 64 fetched 192: a quarter
                 * of this kernel's read bytes) */
                const int last = n > 0 ? n - 1 : 0, mine = lane + 64 * k;
                const size_t at = (size_t)(sg < t1 ? sg : t0) * kSeg + (size_t)(mine < last ? mine : last);
                key_n[j][k] = 0u;
                z_n[j][k] = 0.f;
                if (64 * k < __builtin_amdgcn_readfirstlane(n)) {
                    const uint2 kz = fcand[at];
                    key_n[j][k] = kz.x;
                    z_n[j][k] = __uint_as_float(kz.y);
                }
            }
        }
    };
    request(t0 + wv);
    for (int s0 = t0 + wv; s0 < t1; s0 += kWaves * kResolveBatch) {
        uint32_t key[kResolveBatch][kSl];
        float z[kResolveBatch][kSl];
#pragma unroll
        for (int j = 0; j < kResolveBatch; ++j)
#pragma unroll
            for (int k = 0; k < kSl; ++k) {
                key[j][k] = key_n[j][k];
                z[j][k] = z_n[j][k];
            }
        if (s0 + kWaves * kResolveBatch < t1) request(s0 + kWaves * kResolveBatch); /* (wave-uniform) */
#pragma unroll
        for (int j = 0; j < kResolveBatch; ++j) {
            const int sg = s0 + j * kWaves;
            const int n = sg < t1 ? (int)cnt[sg - t0] : 0;
            const int rr = sg / strips, strip = sg - rr * strips;
            const uint32_t slot0 = (uint32_t)((rr + lo_row - 1) * H + strip * kStripCols);
#pragma unroll
            for (int k = 0; k < kSl; ++k) {
                if (64 * k >= n) break; /* wave-uniform */
                const uint32_t kk = key[j][k];
                const bool have = lane + 64 * k < n;
                const int cell = (int)(kk & kKeyCellMask);
                const bool hit = have && (z[j][k] - minavg[cell]) >= 0.3f;
                const bool pred = (kk & kKeyPredBit) != 0u;
                const bool wrong = have && hit != pred;
                const uint32_t idx = slot0 + ((kk >> kKeyColShift) & 0xffu);
                const bool coded = hit && !(kk & kKeyNoCodeBit);
                const unsigned long long cm = __ballot(coded);
                if (cm != 0ull) { /* (wave-uniform) park them behind the ring's head */
                    const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(cm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)cm, 0u));
                    if (coded) {
                        const uint32_t at = (rhead + rank) & 127u;
                        ring[wv][0][at] = kk;
                        ring[wv][1][at] = __float_as_uint(z[j][k]);
                        ring[wv][2][at] = idx;
                    }
                    rhead += (uint32_t)__popcll(cm);
                    if (rhead - rtail >= 64u) emit(64u);
                }
                if (wrong) { /* the walk's provisional label differs */
                    /* not un-grounded: label = 0, BatchMultiBevGen.cpp:245; un-grounded: the point's own label back —
                     * which is -2: the walk guesses "stays ground" only for points that carry it */
                    flabel[16u * idx + 14u] = hit ? (uint16_t)(int16_t)-2 : (uint16_t)0;
                }
            }
        }
    }
    if (rhead != rtail) emit(rhead - rtail); /* (wave-uniform) what is left of the part's */
    lds_barrier();
    if (tid < bands) b.ncode[((size_t)f * g.emitters + g.strips + part) * bands + tid] = band_cursor[tid];
  }
    TL_END(K_GROUND_RESOLVE);
}

template <bool kPow2>
__global__ __launch_bounds__(kResolveThreads) void k_ground_resolve(BatchPtrs b, Geometry g)
{
    __shared__ __attribute__((aligned(16))) char arena[sizeof(ResolveLds)];
    const int f = (int)blockIdx.x / kResolveWgs;
    resolve_body<kPow2>(arena, b, g, f, (int)blockIdx.x - f * kResolveWgs);
}

} /* namespace bevk */

#endif /* BEV_RESOLVE_H */
