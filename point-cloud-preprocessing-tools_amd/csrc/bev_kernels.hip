/*
 * bev_kernels.hip — hand-written HIP kernels (gfx950, wave64) for the
 * batch_multi_bev_gen hot path.  No MFMA: the path is scatter / stencil /
 * ordered reduction / raster, bounded by HBM.
 *
 * Pipeline for one sub-batch of frames (all launches on one stream):
 *
 *   (winner table: generation-tagged, cleared only when the tag wraps)
 *   order_scan      per input point : winner[slot] = max(index+1)          (getOrderedCloud, last writer wins)
 *   gather_ground   per slot        : ordered cloud, phase-A ground flag,
 *                                     BEV code, candidate list               (getOrderedCloud + markGroundPoints phase A)
 *   cell_sums       per frame       : stable counting sort of candidates by
 *                                     2 m cell, then IN-ORDER float sums     (markGroundPoints phase B + divide)
 *   ground_resolve  per frame row   : 4-neighbour height test, label fix-up (markGroundPoints phase C)
 *   bev_raster      per frame band  : LDS atomics, then coalesced 16 B stores
 *                                     of the 24 occupancy planes + max-height
 *                                     plane                                  (computeAndSave{Multi,Single}Bev rasters)
 *
 * Order-dependent results of the reference are reproduced by construction:
 *   - last-writer-wins scatter  -> atomicMax on (input index + 1);
 *   - row-major float32 accumulation per cell -> candidates are emitted in
 *     slot order, sorted STABLY by cell, and each cell is summed by one lane
 *     sequentially (a tree or atomic float reduction would change low bits).
 */
#include <algorithm>
#include <cstdlib>
#include <type_traits>

#include "bev_internal.h"
#include "bev_libm.h"

#include "bev_dev.h"
#include "bev_front.h"
#include "bev_walk.h"
#include "bev_cell_sums.h"
#include "bev_resolve.h"
#include "bev_raster.h"
#include "bev_misc.h"

using namespace bevx;

namespace bevk {

static const char *const kNames[K_COUNT] = {
    "k_order_scan", "k_walk", "k_cell_sums", "k_ground_resolve", "k_bev_raster",
    "k_gather_only", "k_ground_mat", "k_cloud_codes", "k_angle_debug", "k_float_bev", "k_project", "k_transform",
    "k_probe", "k_walk_general", "k_walk_structured", "k_walk_colmajor", "k_walk_colmajor_gen", "k_verdict", "k_stage",
};
const char *kernel_name(int id) { return (id >= 0 && id < K_COUNT) ? kNames[id] : "?"; }

/* ------------------------------------------------------------------------- */
/* k_stage: the stages of FOUR consecutive sub-batches as workgroups of one grid.
 *
 * Rounds 2-5 ran the front (probe, walk) and the back (cell sums, resolve, rasters) of neighbouring sub-batches on two
 * streams of different priority.  The time lines of round 5 (profiles/r05_timeline_hdl_pipelined.txt) showed what that buys
 * and what it cannot: the back stage's kernels were not the walk's shape (512 threads, 50 KB of LDS) and started only
 * where two walk workgroups of one CU retired together; every hand-over between the streams and every kernel boundary left
 * the chip to a launch's tail.  Here every stage has the walk's shape (bev_internal.h: kStageThreads, kSlotLdsBytes), a
 * launch holds
 *     the walk of sub-batch t  |  phase B of t - 1  |  phase C of t - 2  |  the rasters of t - 3
 * and a workgroup's number says which it is: group slot i (8 frames, one per XCD: blocks b and b + 8 share an L2) holds
 * the walk's strips of frame group i and the later stages' workgroups of frame group i - lead, so that a CU holds a mix of
 * long memory-bound and short latency-bound workgroups at any time, the launch starts with walks alone and ends with the
 * short workgroups of the later stages (a launch's tail is as long as its last workgroups live).  The stages of ONE
 * sub-batch meet only through the order of launches on the stream. */
template <int kSrc, bool kPow2, bool kGm>
__global__ __launch_bounds__(kStageThreads, 4) void k_stage(StageArgs a)
{
    extern __shared__ __attribute__((aligned(16))) char stage_arena[];
    static_assert(kSrc != kSrcColMajor && kSrc != kSrcColMajorGen, "the firing-order walks are launched apart");
    static_assert(kStripThreads == kStageThreads && kSumThreads == kStageThreads && kResolveThreads == kStageThreads && kRasterThreads == kStageThreads, "one workgroup shape");
    const int bid = (int)blockIdx.x, x = bid & 7, j = bid >> 3;
    const int strips = a.g.strips;
    const int per_slot = strips + kSumQ + kResolveWgs + a.g.raster_bands;
    const int i = j / per_slot;
    int k = j - i * per_slot;
    if (k < strips) {
        const int f = i * 8 + x;
        if (f < a.walk.nf) walk_body<kSrc, kPow2, kGm>(stage_arena, a.walk.b, a.g, f, k, a.want_mode, bid);
        return;
    }
    const int f = (i - a.lead) * 8 + x;
    if (i < a.lead) return;
    k -= strips;
    if (k < kSumQ) {
        if (f < a.sums.nf) cell_sums_body(reinterpret_cast<uint32_t *>(stage_arena), a.sums.b, a.g, f, k, bid);
        return;
    }
    k -= kSumQ;
    if (k < kResolveWgs) {
        if (f < a.resolve.nf) resolve_body<kPow2>(stage_arena, a.resolve.b, a.g, f, k);
        return;
    }
    k -= kResolveWgs;
    if (f < a.raster.nf) raster_body(reinterpret_cast<uint32_t *>(stage_arena), a.raster.b, a.g, f, k, a.want_multi, a.want_single);
}

template <int kSrc>
static size_t walk_lds_bytes() { return sizeof(WalkLds<kSrc>); }
size_t stage_lds_bytes(const Geometry &g, int source)
{
    size_t w = 0;
    switch (source) {
    case kSrcIdentity: w = walk_lds_bytes<kSrcIdentity>(); break;
    case kSrcInPlace: w = walk_lds_bytes<kSrcInPlace>(); break;
    case kSrcStructured: w = walk_lds_bytes<kSrcStructured>(); break;
    /* (the firing-order walks are never fused: 50-53 KB of LDS would hold every stage of the launch to three per CU) */
    default: w = walk_lds_bytes<kSrcGather>(); break;
    }
    const size_t others = std::max(std::max(SumDims::lds_bytes(g.segs), sizeof(ResolveLds)), raster_lds_bytes(g));
    return std::max(w, others);
}
template <int kSrc>
static void launch_stage_src(const StageArgs &a, hipStream_t st)
{
    const Geometry &g = a.g;
    const bool pow2 = g.rp.inv_interval != 0.0f && g.rp.inv_height_res != 0.0f; /* every configuration of the reference */
    const bool gm = a.walk.nf > 0 && a.walk.b.gm != nullptr;
    const int back = std::max(a.sums.nf, std::max(a.resolve.nf, a.raster.nf));
    const int lead = back > 0 ? a.lead : 0;
    const int slots = std::max((a.walk.nf + 7) / 8, back > 0 ? (back + 7) / 8 + lead : 0);
    if (slots == 0) return;
    const int per_slot = g.strips + kSumQ + kResolveWgs + g.raster_bands;
    const dim3 gr((unsigned)(8 * slots * per_slot)), bl(kStageThreads);
    const size_t lds = stage_lds_bytes(g, kSrc);
    StageArgs args = a;
    args.lead = lead;
    if (pow2 && !gm) hipLaunchKernelGGL((k_stage<kSrc, true, false>), gr, bl, lds, st, args);
    else if (pow2) hipLaunchKernelGGL((k_stage<kSrc, true, true>), gr, bl, lds, st, args);
    else if (!gm) hipLaunchKernelGGL((k_stage<kSrc, false, false>), gr, bl, lds, st, args);
    else hipLaunchKernelGGL((k_stage<kSrc, false, true>), gr, bl, lds, st, args);
}
void launch_stage(const StageArgs &a, int source, hipStream_t st)
{
    if (a.walk.nf == 0) source = kSrcInPlace; /* (any instantiation will do for a launch without a walk) */
    if (source == kSrcIdentity) launch_stage_src<kSrcIdentity>(a, st);
    else if (source == kSrcInPlace) launch_stage_src<kSrcInPlace>(a, st);
    else if (source == kSrcStructured) launch_stage_src<kSrcStructured>(a, st);
    else launch_stage_src<kSrcGather>(a, st); /* (kSrcColMajor / kSrcColMajorGen: launched apart, see run_pipeline) */
}

/* ------------------------------------------------------------------------- */
/* launchers                                                                  */
hipError_t configure_kernels(const Geometry &g)
{
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_cell_sums),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)SumDims::lds_bytes(kMaxSegs));
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_bev_raster), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)raster_lds_bytes(g));
    if (e != hipSuccess) return e;
    return hipFuncSetAttribute(reinterpret_cast<const void *>(k_bev_raster_dense),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)raster_lds_bytes(g));
}
void launch_order_scan(const Geometry &g, const BatchPtrs &b, int nf, uint32_t max_pts, bool thin, hipStream_t st)
{
    if (max_pts == 0 || nf == 0) return;
    const unsigned per_block = 256u * kScanPerThread;
    const unsigned blocks = (max_pts + per_block - 1u) / per_block;
    /* wide: one workgroup per 1024-point block (every load of a frame in flight at once: 20 % faster when frames do go
     * this way); thin: 8 per frame striding over the blocks, for the launch that is expected to find nothing to do */
    dim3 grid(thin && blocks > 8u ? 8u : blocks, (unsigned)nf);
    hipLaunchKernelGGL(k_order_scan, grid, dim3(256), 0, st, b.pts, b.frames, b.info, b.winner, g.N, g.H, g.S,
                       b.win_tag << b.win_shift);
}
template <int kSrc>
static void launch_walk(const Geometry &g, const BatchPtrs &b, int nf, uint32_t mode, int grid, hipStream_t st)
{
    const bool pow2 = g.rp.inv_interval != 0.0f && g.rp.inv_height_res != 0.0f; /* every configuration of the reference */
    const dim3 gr(grid), bl(kStripThreads);
    if (pow2 && !b.gm) hipLaunchKernelGGL((k_walk<kSrc, true, false>), gr, bl, 0, st, b, g, nf, mode);
    else if (pow2) hipLaunchKernelGGL((k_walk<kSrc, true, true>), gr, bl, 0, st, b, g, nf, mode);
    else if (!b.gm) hipLaunchKernelGGL((k_walk<kSrc, false, false>), gr, bl, 0, st, b, g, nf, mode);
    else hipLaunchKernelGGL((k_walk<kSrc, false, true>), gr, bl, 0, st, b, g, nf, mode);
}
void launch_gather_ground(const Geometry &g, const BatchPtrs &b, int nf, int source, uint32_t mode, hipStream_t st)
{
    if (nf == 0) return;
    const int grid = xcd_grid(nf, g.strips);
    if (source == kSrcIdentity) launch_walk<kSrcIdentity>(g, b, nf, mode, grid, st);
    else if (source == kSrcInPlace) launch_walk<kSrcInPlace>(g, b, nf, mode, grid, st);
    else if (source == kSrcStructured) launch_walk<kSrcStructured>(g, b, nf, mode, grid, st);
    else if (source == kSrcColMajor) launch_walk<kSrcColMajor>(g, b, nf, mode, grid, st);
    else if (source == kSrcColMajorGen) launch_walk<kSrcColMajorGen>(g, b, nf, mode, grid, st);
    else launch_walk<kSrcGather>(g, b, nf, mode, grid, st);
}
void launch_probe(const Geometry &g, const BatchPtrs &b, int nf, bool allow_stream, int layout_hint, hipStream_t st)
{
    if (nf == 0) return;
    hipLaunchKernelGGL(k_probe, dim3(nf), dim3(kProbeThreads), 0, st, b, g, allow_stream ? 1 : 0, layout_hint);
}
void launch_verdict(const Geometry &g, const BatchPtrs &b, int nf, uint32_t *host_hint, hipStream_t st)
{
    if (nf == 0) return;
    hipLaunchKernelGGL(k_verdict, dim3(1), dim3(1024), 0, st, b.info, nf, host_hint, b.cm_sync, g.N);
}
void launch_gather_only(const Geometry &g, const BatchPtrs &b, int nf, hipStream_t st)
{
    if (nf == 0) return;
    hipLaunchKernelGGL(k_gather_only, dim3(xcd_grid(nf, g.tiles)), dim3(kGatherThreads), 0, st, b, g, nf);
}
void launch_cell_sums(const Geometry &g, const BatchPtrs &b, int nf, hipStream_t st)
{
    if (nf == 0) return;
    hipLaunchKernelGGL(k_cell_sums, dim3(xcd_grid(nf, kSumQ)), dim3(kSumThreads), SumDims::lds_bytes(g.segs), st, b, g, nf);
}
void launch_ground_resolve(const Geometry &g, const BatchPtrs &b, int nf, hipStream_t st)
{
    if (nf == 0) return;
    const bool pow2 = g.rp.inv_interval != 0.0f && g.rp.inv_height_res != 0.0f;
    if (pow2) hipLaunchKernelGGL(k_ground_resolve<true>, dim3(nf * kResolveWgs), dim3(kResolveThreads), 0, st, b, g);
    else hipLaunchKernelGGL(k_ground_resolve<false>, dim3(nf * kResolveWgs), dim3(kResolveThreads), 0, st, b, g);
}
void launch_bev_raster(const Geometry &g, const BatchPtrs &b, bool want_multi, bool want_single, int nf, hipStream_t st)
{
    if (nf == 0) return;
    hipLaunchKernelGGL(k_bev_raster, dim3(8 * ((nf + 7) / 8) * g.raster_bands), dim3(kRasterThreads), raster_lds_bytes(g), st, b,
                       g, nf, want_multi ? 1 : 0, want_single ? 1 : 0);
}
void launch_bev_raster_dense(const Geometry &g, const uint32_t *codes, uint32_t n_codes, uint8_t *multi, uint8_t *single,
                             hipStream_t st)
{
    hipLaunchKernelGGL(k_bev_raster_dense, dim3(g.raster_bands), dim3(kRasterThreads), raster_lds_bytes(g), st, codes,
                       n_codes, multi, single, g.rp);
}
void launch_ground_mat(const Geometry &g, const BatchPtrs &b, int8_t *out, int nf, hipStream_t st)
{
    if (nf == 0) return;
    hipLaunchKernelGGL(k_ground_mat, dim3(xcd_grid(nf, g.tiles)), dim3(kGatherThreads), 0, st, b, g, out, nf);
}
void launch_cloud_codes(const Geometry &g, const bev_point_t *cloud, uint32_t n, uint32_t *codes, hipStream_t st)
{
    if (n == 0) return;
    hipLaunchKernelGGL(k_cloud_codes, dim3((n + 255u) / 256u), dim3(256), 0, st, cloud, n, codes, g.rp);
}
void launch_float_bev(const bev_point_t *cloud, uint32_t n, float interval, int M, bool skip_label0, float *grid,
                      hipStream_t st)
{
    if (n == 0) return;
    hipLaunchKernelGGL(k_float_bev, dim3((n + 255u) / 256u), dim3(256), 0, st, cloud, n, interval, M,
                       skip_label0 ? 1 : 0, reinterpret_cast<uint32_t *>(grid));
}
void launch_transform(const bev_point_t *cloud, uint32_t n, const float m[12], bev_point_t *out, hipStream_t st)
{
    if (n == 0) return;
    Affine34 a;
    for (int k = 0; k < 12; ++k) a.m[k] = m[k];
    hipLaunchKernelGGL(k_transform, dim3((n + 255u) / 256u), dim3(256), 0, st, cloud, n, a, out);
}
void launch_project(int kind, const float *xyzi, uint32_t n, bev_point_t *out, hipStream_t st)
{
    if (n == 0) return;
    hipLaunchKernelGGL(k_project, dim3((n + 255u) / 256u), dim3(256), 0, st, kind, xyzi, n, out);
}
void launch_project_kitti(const float *xyzi, uint32_t n, const KittiWork &w, bev_point_t *out, hipStream_t st)
{
    /* n >= 1; w.winner zeroed by the caller on the same stream */
    const unsigned blocks = (n + kKittiBlock - 1u) / kKittiBlock;
    hipLaunchKernelGGL(k_kitti_crossings, dim3(blocks), dim3(kKittiBlock), 0, st, xyzi, n, w.col, w.cnt, w.pos, w.hdr);
    hipLaunchKernelGGL(k_kitti_chain, dim3(1), dim3(64), 0, st, w.cnt, w.pos, n, kitti_ring_min(), w.hdr);
    hipLaunchKernelGGL(k_kitti_assign, dim3((n + 255u) / 256u), dim3(256), 0, st, w.col, n, w.hdr, w.winner);
    hipLaunchKernelGGL(k_kitti_gather, dim3((kKittiRows * kKittiCols + 255) / 256), dim3(256), 0, st, xyzi, w.winner, out);
}
void launch_angle_debug(const float *dx, const float *dy, const float *dz, uint8_t *out, size_t n, hipStream_t st)
{
    if (n == 0) return;
    hipLaunchKernelGGL(k_angle_debug, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, dx, dy, dz, out, n);
}

} /* namespace bevk */
