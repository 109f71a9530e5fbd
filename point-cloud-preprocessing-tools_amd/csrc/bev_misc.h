/*
 * bev_misc.h — single-cloud entry points and the 'next' rows: gather only, final ground_mat, dense codes, float BEV, rigid transform, range-image projections
 * Part of the device code of libbev_mi355x.so; included by bev_kernels.hip only (one translation unit).
 */
#ifndef BEV_MISC_H
#define BEV_MISC_H

#include "bev_dev.h"
#include "bev_libm.h"

namespace bevk {
using namespace bevx;

/* getOrderedCloud alone (bev_order_cloud): no ground work. */
__global__ __launch_bounds__(kGatherThreads) void k_gather_only(BatchPtrs b, Geometry g, int nf)
{
    int f, tile;
    if (!map_block_xcd(blockIdx.x, nf, g.tiles, f, tile)) return;
    const size_t fbase = (size_t)f * g.S;
    const bev_point_t *fpts = b.pts + b.frames[f].in_offset;
#pragma unroll
    for (int k = 0; k < kSlotsPerThread; ++k) {
        const int slot = tile * kTile + k * kGatherThreads + threadIdx.x;
        if (slot >= g.S) continue;
        Half lo = {{0, 0, 0, 0}}, hi = {{0, 0, 0, 0}};
        const uint32_t w = winner_index(b.winner[fbase + slot], b.win_tag, b.win_shift);
        if (w) {
            lo = *reinterpret_cast<const Half *>(fpts + (w - 1));
            hi = *(reinterpret_cast<const Half *>(fpts + (w - 1)) + 1);
        }
        Half *dst = reinterpret_cast<Half *>(b.ordered + fbase + slot);
        dst[0] = lo;
        dst[1] = hi;
    }
}

/* ------------------------------------------------------------------------- */
/* Final cv::Mat ground_mat (optional output): phase C writes 0 wherever the
 * neighbour test fires, for EVERY slot (:236-240). */
__global__ __launch_bounds__(kGatherThreads) void k_ground_mat(BatchPtrs b, Geometry g, int8_t *out, int nf)
{
    int f, tile;
    if (!map_block_xcd(blockIdx.x, nf, g.tiles, f, tile)) return;
#pragma unroll
    for (int k = 0; k < kSlotsPerThread; ++k) {
        const int slot = tile * kTile + k * kGatherThreads + threadIdx.x;
        if (slot >= g.S) continue;
        const size_t idx = (size_t)f * g.S + slot;
        const float4 a = *reinterpret_cast<const float4 *>(b.ordered + idx);
        const int cell = ground_cell(a.x, a.y);
        const bool hit = above_neighbour_ground(a.z, cell, b.avg + (size_t)f * kCells);
        out[idx] = hit ? (int8_t)0 : b.gm[idx];
    }
}

/* ------------------------------------------------------------------------- */
/* BEV code of every point of an arbitrary cloud (bev_multi_bev / bev_single_bev). */
__global__ __launch_bounds__(256) void k_cloud_codes(const bev_point_t *__restrict__ cloud, uint32_t n,
                                                     uint32_t *__restrict__ codes, RasterParams rp)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const float4 a = *reinterpret_cast<const float4 *>(cloud + i);
    const int label = (int)reinterpret_cast<const int16_t *>(cloud + i)[14];
    codes[i] = bev_code(a.x, a.y, a.z, label, rp);
}

/* saveAsMat of batch_cloud_manip / cloud_manip (BatchCloudManip.cpp:213-225, CloudManip.cpp:84-95):
 * float32 max of z + 2.0f per cell over a grid initialised to 0.  A stored value is always > 0, and
 * positive IEEE floats order like their bit patterns, so the max is an integer atomicMax. */
__global__ __launch_bounds__(256) void k_float_bev(const bev_point_t *__restrict__ cloud, uint32_t n, float interval,
                                                   int M, int skip_label0, uint32_t *__restrict__ grid)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const float4 a = *reinterpret_cast<const float4 *>(cloud + i);
    const int label = (int)reinterpret_cast<const int16_t *>(cloud + i)[14];
    const int x = bev_bin(a.x, 100.0f, interval); /* MAX_RANGE = 100, :209 / :81 */
    const int y = bev_bin(a.y, 100.0f, interval);
    if (x < 0 || x >= M || y < 0 || y >= M) return;
    if (skip_label0 && label == 0) return;         /* :218 (batch variant only) */
    const float h = a.z + 2.0f;                    /* :222 / :92 */
    if (h > 0.0f) atomicMax(&grid[(size_t)x * M + y], __float_as_uint(h)); /* "h > cell" with cells >= 0 */
}

/* pcl::transformPointCloud with the [R | t] of cloud_manip (CloudManip.cpp:119-128): out.xyz = col0 * x + (col1 * y +
 * (col2 * z + col3)) — the association of pcl::detail::Transformer<float>::se3 — every other field copied.  The matrix
 * is built on the host (sinf / cosf of the host libm), so no transcendental is evaluated here. */
struct Affine34 { float m[12]; };
__global__ __launch_bounds__(256) void k_transform(const bev_point_t *cloud, uint32_t n, Affine34 a, bev_point_t *out)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    Half lo = reinterpret_cast<const Half *>(cloud + i)[0];
    const Half hi = reinterpret_cast<const Half *>(cloud + i)[1];
    const float x = __uint_as_float(lo.w[0]), y = __uint_as_float(lo.w[1]), z = __uint_as_float(lo.w[2]);
    lo.w[0] = __float_as_uint(a.m[0] * x + (a.m[1] * y + (a.m[2] * z + a.m[3])));
    lo.w[1] = __float_as_uint(a.m[4] * x + (a.m[5] * y + (a.m[6] * z + a.m[7])));
    lo.w[2] = __float_as_uint(a.m[8] * x + (a.m[9] * y + (a.m[10] * z + a.m[11])));
    reinterpret_cast<Half *>(out + i)[0] = lo;
    reinterpret_cast<Half *>(out + i)[1] = hi;
}

/* Range-image projection of raw returns (see bev_libm.h): one thread per point. */
__global__ __launch_bounds__(256) void k_project(int kind, const float *__restrict__ xyzi, uint32_t n,
                                                 bev_point_t *__restrict__ out)
{
    const uint32_t k = blockIdx.x * 256u + threadIdx.x;
    if (k >= n) return;
    float x, y, z, it;
    uint16_t row, col;
    if (kind == BEV_PROJECT_MULRAN_OS1_64) {
        const float4 v = reinterpret_cast<const float4 *>(xyzi)[k];
        x = v.x; y = v.y; z = v.z; it = v.w;
        project_mulran(k, x, y, row, col);
    } else {
        x = -xyzi[k]; y = xyzi[(size_t)n + k]; z = -xyzi[2 * (size_t)n + k]; it = xyzi[3 * (size_t)n + k];
        project_oxford(x, y, z, row, col);
    }
    Half lo, hi;
    lo.w[0] = __float_as_uint(x); lo.w[1] = __float_as_uint(y); lo.w[2] = __float_as_uint(z); lo.w[3] = 0u;
    hi.w[0] = __float_as_uint(it); hi.w[1] = (uint32_t)row | ((uint32_t)col << 16); hi.w[2] = 0u;
    hi.w[3] = (uint32_t)(uint16_t)(int16_t)-2; /* label = -2 */
    Half *dst = reinterpret_cast<Half *>(out + k);
    dst[0] = lo;
    dst[1] = hi;
}


/* ---- KITTI projection (see bev_libm.h): crossings -> chain of accepted crossings -> rings -> structured cloud ---- */
/* per point: azimuth, column, crossing flag; per block of 256 points: the ascending list of crossing positions */
__global__ __launch_bounds__(kKittiBlock) void k_kitti_crossings(const float *__restrict__ xyzi, uint32_t n,
                                                                 int32_t *__restrict__ col, uint32_t *__restrict__ cnt,
                                                                 uint32_t *__restrict__ pos, KittiHeader *__restrict__ hdr)
{
    __shared__ float az[kKittiBlock + 1];
    __shared__ uint32_t wave_base[kKittiBlock / 64 + 1];
    const uint32_t tid = threadIdx.x, i = blockIdx.x * (uint32_t)kKittiBlock + tid;
    float a = 0.0f;
    if (i < n) {
        const float4 v = reinterpret_cast<const float4 *>(xyzi)[i];
        a = kitti_azimuth(v.x, v.y);
        col[i] = kitti_col(a);
        if (i == 0) hdr->ring0 = a > 0.0f ? 0 : -1; /* :195-203 */
    }
    az[tid + 1] = a;
    if (tid == 0 && i >= 1 && i < n) {
        const float4 v = reinterpret_cast<const float4 *>(xyzi)[i - 1];
        az[0] = kitti_azimuth(v.x, v.y);
    }
    __syncthreads();
    const bool flag = i >= 1 && i < n && kitti_crossing(az[tid], az[tid + 1]);
    const uint64_t m = __ballot(flag);
    const uint32_t lane = tid & 63u, wave = tid >> 6;
    if (lane == 0) wave_base[wave + 1] = (uint32_t)__popcll(m);
    __syncthreads();
    if (tid == 0) {
        wave_base[0] = 0;
        for (int w = 0; w < kKittiBlock / 64; ++w) wave_base[w + 1] += wave_base[w];
        cnt[blockIdx.x] = wave_base[kKittiBlock / 64];
    }
    __syncthreads();
    if (flag) pos[(size_t)blockIdx.x * kKittiListCap + wave_base[wave] + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = i;
}

/* one wave walks the chain of accepted crossings */
__global__ __launch_bounds__(64) void k_kitti_chain(const uint32_t *__restrict__ cnt, const uint32_t *__restrict__ pos,
                                                    uint32_t n, uint32_t ring_min, KittiHeader *__restrict__ hdr)
{
    const uint32_t lane = threadIdx.x, nblocks = (n + kKittiBlock - 1u) / kKittiBlock;
    int ring = hdr->ring0;
    uint32_t last = 1, links = 0; /* count == i - last; before any crossing count == i - 1 (:210-212) */
    while (ring < kKittiRows && links < (uint32_t)kKittiMaxLinks) {
        const uint64_t target = ring == -1 ? 1ull : (uint64_t)last + ring_min;
        if (target >= n) break;
        uint32_t found = 0; /* crossings are at positions >= 1 */
        const uint32_t b = (uint32_t)(target / kKittiBlock), c = cnt[b];
        for (uint32_t k0 = 0; k0 < c && !found; k0 += 64) {
            const uint32_t k = k0 + lane;
            const uint32_t p = k < c ? pos[(size_t)b * kKittiListCap + k] : 0u;
            const uint64_t hit = __ballot(k < c && p >= target);
            if (hit) found = __shfl(p, __ffsll((long long)hit) - 1);
        }
        for (uint32_t b0 = b + 1; b0 < nblocks && !found; b0 += 64) {
            const uint32_t bb = b0 + lane;
            const uint64_t hit = __ballot(bb < nblocks && cnt[bb] > 0u);
            if (hit) found = pos[(size_t)(b0 + (uint32_t)__ffsll((long long)hit) - 1u) * kKittiListCap];
        }
        if (!found) break;
        ring = ring == -1 ? 0 : ring + 1;
        last = found;
        if (lane == 0) hdr->link[links] = found;
        ++links;
    }
    if (lane == 0) hdr->n_links = links;
}

/* ring of every point, then last-writer-wins on its slot (:240) */
__global__ __launch_bounds__(256) void k_kitti_assign(const int32_t *__restrict__ col, uint32_t n,
                                                      const KittiHeader *__restrict__ hdr, uint32_t *__restrict__ winner)
{
    __shared__ uint32_t link[kKittiMaxLinks];
    const uint32_t n_links = hdr->n_links;
    if (threadIdx.x < n_links) link[threadIdx.x] = hdr->link[threadIdx.x];
    __syncthreads();
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i < 1u || i >= n) return; /* the loop at :212 starts at 1 */
    const int ring = kitti_ring_of(i, hdr->ring0, link, n_links), c = col[i];
    if (ring >= 0 && ring < kKittiRows && c >= 0) atomicMax(&winner[(uint32_t)ring * kKittiCols + (uint32_t)c], i + 1u);
}

/* the structured cloud: winners with intensity = -1, label = -2 (:235-238), empty slots all-zero (:207) */
__global__ __launch_bounds__(256) void k_kitti_gather(const float *__restrict__ xyzi, const uint32_t *__restrict__ winner,
                                                      bev_point_t *__restrict__ out)
{
    const uint32_t s = blockIdx.x * 256u + threadIdx.x;
    if (s >= (uint32_t)(kKittiRows * kKittiCols)) return;
    Half lo{{0, 0, 0, 0}}, hi{{0, 0, 0, 0}};
    const uint32_t w = winner[s];
    if (w != 0u) {
        const float4 v = reinterpret_cast<const float4 *>(xyzi)[w - 1u];
        lo.w[0] = __float_as_uint(v.x); lo.w[1] = __float_as_uint(v.y); lo.w[2] = __float_as_uint(v.z);
        hi.w[0] = __float_as_uint(-1.0f);
        hi.w[1] = (s / (uint32_t)kKittiCols) | ((s % (uint32_t)kKittiCols) << 16);
        hi.w[3] = (uint32_t)(uint16_t)(int16_t)-2;
    }
    Half *dst = reinterpret_cast<Half *>(out + s);
    dst[0] = lo;
    dst[1] = hi;
}

/* test hook: the phase-A angle predicate on raw difference vectors */
__global__ __launch_bounds__(256) void k_angle_debug(const float *dx, const float *dy, const float *dz,
                                                     uint8_t *out, size_t n)
{
    const size_t i = (size_t)blockIdx.x * 256u + threadIdx.x;
    if (i < n) out[i] = angle_is_ground(dx[i], dy[i], dz[i]) ? 1 : 0;
}

} /* namespace bevk */

#endif /* BEV_MISC_H */
