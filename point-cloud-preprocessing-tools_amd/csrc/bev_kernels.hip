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
#include <cstdlib>
#include <type_traits>

#include "bev_internal.h"
#include "bev_libm.h"

using namespace bevx;

/* developer aid (make clk): phase durations of one workgroup per kernel, printed in 10 ns ticks */
#ifdef BEV_CS_CLOCK
#define PH_DECL long long ph_clk[12]; int ph_n = 0
#define PH() ph_clk[ph_n++] = wall_clock64()
#define PH_PRINT(name, cond)                                                                      \
    do {                                                                                          \
        if (cond) {                                                                               \
            long long d_[6] = {0, 0, 0, 0, 0, 0};                                                 \
            for (int i_ = 1; i_ < ph_n && i_ <= 6; ++i_) d_[i_ - 1] = ph_clk[i_] - ph_clk[i_ - 1]; \
            printf("%s: %lld %lld %lld %lld %lld %lld (x10 ns)\n", name, d_[0], d_[1], d_[2], d_[3], d_[4], d_[5]); \
        }                                                                                         \
    } while (0)
#define PHA_DECL long long pha_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, pha_t = wall_clock64()
#define PHA(i) do { const long long n_ = wall_clock64(); pha_[i] += n_ - pha_t; pha_t = n_; } while (0)
#define PHA_PRINT(name, cond) do { if (cond) printf("%s: %lld %lld %lld %lld %lld %lld %lld %lld (x10 ns)\n", name, pha_[0], pha_[1], pha_[2], pha_[3], pha_[4], pha_[5], pha_[6], pha_[7]); } while (0)
#else
#define PH_DECL
#define PH()
#define PH_PRINT(name, cond)
#define PHA_DECL
#define PHA(i)
#define PHA_PRINT(name, cond)
#endif

namespace bevk {

static const char *const kNames[K_COUNT] = {
    "k_order_scan", "k_strip_ground", "k_cell_sums", "k_ground_resolve", "k_bev_raster",
    "k_gather_only", "k_ground_mat", "k_cloud_codes", "k_angle_debug", "k_float_bev", "k_project", "k_transform",
    "k_probe",
};
const char *kernel_name(int id) { return (id >= 0 && id < K_COUNT) ? kNames[id] : "?"; }

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

/* Loads the compiler does not see, and the counted waits that go with them (stream source of the column walk).
 * gfx950 retires loads in order among loads (and stores among stores) but counts both on vmcnt, so hipcc, which
 * cannot tell how many of the pending operations are stores, waits for ALL of them (vmcnt(0)) before a loaded register
 * is used in a loop that also stores — a software pipeline is drained once per iteration.  The sound rule is weaker: a
 * load has completed once at most as many operations are outstanding as LOADS were issued after it (stores only make
 * that wait longer, never wrong).  glds16 is an LDS-DMA load (global_load_lds_dwordx4): every lane gives its own source
 * address, the 64 x 16 bytes land at a wave-uniform LDS address + lane * 16, no VGPR is involved, and it counts on
 * vmcnt like any load (scripts/microbench/glds_test.hip checks both on the box). */
__device__ __forceinline__ uint32_t lds_addr(const void *p)
{
    return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void *)p;
}
__device__ __forceinline__ void glds16(const void *gsrc, uint32_t lds_dst)
{
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ void ld128(u32x4 &dst, const void *p)
{
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst) : "v"(p));
}
__device__ __forceinline__ void ld128_16(u32x4 &dst, const void *p) /* bytes 16 .. 31 of *p */
{
    asm volatile("global_load_dwordx4 %0, %1, off offset:16" : "=v"(dst) : "v"(p));
}
__device__ __forceinline__ void ld32_nt(uint32_t &dst, const void *p)
{
    asm volatile("global_load_dword %0, %1, off nt" : "=v"(dst) : "v"(p));
}
/* all but the N newest memory operations have completed; the registers named are released by this wait: their uses
 * cannot be scheduled above it */
template <int N>
__device__ __forceinline__ void wait_loads(u32x4 &a, u32x4 &b, uint32_t &c)
{
    asm volatile("s_waitcnt vmcnt(%3)" : "+v"(a), "+v"(b), "+v"(c) : "n"(N) : "memory");
}

/* A winner entry is (tag << shift) | (input index + 1).  The tag is the sub-batch generation of the workspace set:
 * entries left by earlier sub-batches carry a smaller tag, lose every atomicMax against the current one and read as
 * "empty", so the table needs no memset between sub-batches (bev_capi.hip clears it when the tag would wrap). */
__device__ __forceinline__ uint32_t winner_index(uint32_t w, uint32_t tag, int shift)
{
    return (w != 0u && (w >> shift) == tag) ? (w & ((1u << shift) - 1u)) : 0u;
}

/* ------------------------------------------------------------------------- */
/* k_probe: which frames can be read in place.  getOrderedCloud (BatchMultiBevGen.cpp:102-116) scatters the input
 * point by point; when the input already IS in slot order — a sweep written row by row — the scatter is the identity
 * on positions, and reading the input a second time just to learn that (the order scan) is the largest avoidable
 * stream of the path.  One workgroup per frame looks at every 128th point: the leading samples that are in range and
 * strictly ascending bound a prefix [0, T) that is TAKEN for sorted; for every (row, strip) the position of its first
 * slot inside that prefix is estimated by interpolation between the two samples around it.  Nothing here is trusted:
 * the stream walk verifies every point it consumes and a frame that fails is redone the general way. */
constexpr int kProbeThreads = 1024; /* one workgroup per frame: its latency is the kernel's */
__global__ __launch_bounds__(kProbeThreads) void k_probe(BatchPtrs b, Geometry g, int allow_stream)
{
    __shared__ uint32_t samp[kMaxSamples]; /* slot of sample k (position k * kProbeStride) */
    __shared__ uint32_t first_bad, overflow;
    __shared__ uint32_t tcnt[kTailBuckets]; /* tail points listed per (row, strip) */
    const int f = blockIdx.x, tid = threadIdx.x;
    const FrameDesc fd = b.frames[f];
    const uint32_t n = fd.n_pts;
    const bev_point_t *fp = b.pts + fd.in_offset;
    const uint32_t ns = n ? (n - 1u) / kProbeStride + 1u : 0u;
    const bool can = allow_stream && n >= (uint32_t)kStreamMinPrefix && ns <= (uint32_t)kMaxSamples && g.N <= kStreamMaxRows &&
                     g.N * g.strips <= kTailBuckets && n < (1u << 24) && b.tail_list != nullptr;
    if (tid == 0) first_bad = can ? ns : 0u;
    __syncthreads();
    if (can) {
        for (uint32_t k = tid; k < ns; k += (uint32_t)kProbeThreads) {
            const size_t i = (size_t)k * kProbeStride;
            const uint32_t rc = reinterpret_cast<const uint32_t *>(fp + i)[5]; /* row | col << 16 */
            const uint32_t row = rc & 0xffffu, col = rc >> 16;
            uint32_t sl = (row < (uint32_t)g.N && col < (uint32_t)g.H) ? row * (uint32_t)g.H + col : 0xffffffffu;
            if (i + 1 < n) { /* the sample's successor too (mostly the same line): catches column-major orders at once */
                const uint32_t rc1 = reinterpret_cast<const uint32_t *>(fp + i + 1)[5];
                const uint32_t row1 = rc1 & 0xffffu, col1 = rc1 >> 16;
                if (!(row1 < (uint32_t)g.N && col1 < (uint32_t)g.H) || row1 * (uint32_t)g.H + col1 <= sl) sl = 0xffffffffu;
            }
            samp[k] = sl;
        }
        __syncthreads();
        for (uint32_t k = tid; k < ns; k += (uint32_t)kProbeThreads) /* first sample that is out of range or not above its predecessor */
            if (samp[k] == 0xffffffffu || (k > 0u && samp[k] <= samp[k - 1u])) atomicMin(&first_bad, k);
        __syncthreads();
    }
    const uint32_t m = first_bad;                                      /* samples 0 .. m-1 ascend */
    const uint32_t T0 = m ? (m - 1u) * kProbeStride + 1u : 0u;         /* the last of them is position T0 - 1 */
    /* ... and the points after it, one by one, up to the first that does not ascend (at the latest the successor of the
     * sample that failed): a sweep that is sorted to its end has no tail at all, and an appended block of other points
     * starts exactly where the prefix ends — otherwise up to 126 sorted points of ONE (row, strip) would be "tail" */
    __syncthreads();
    if (tid == 0) first_bad = T0 + (uint32_t)kProbeStride + 1u < n ? T0 + (uint32_t)kProbeStride + 1u : n;
    __syncthreads();
    if (can && m && tid <= kProbeStride) {
        const uint32_t i = T0 + (uint32_t)tid;
        if (i < n) {
            const uint32_t rc0 = reinterpret_cast<const uint32_t *>(fp + i - 1u)[5], rc1 = reinterpret_cast<const uint32_t *>(fp + i)[5];
            const uint32_t r0 = rc0 & 0xffffu, c0 = rc0 >> 16, r1 = rc1 & 0xffffu, c1 = rc1 >> 16;
            const bool ok = r0 < (uint32_t)g.N && c0 < (uint32_t)g.H && r1 < (uint32_t)g.N && c1 < (uint32_t)g.H &&
                            r1 * (uint32_t)g.H + c1 > r0 * (uint32_t)g.H + c0;
            if (!ok) atomicMin(&first_bad, i);
        }
    }
    __syncthreads();
    const uint32_t T = m ? first_bad : 0u;
    const bool stream = can && T >= (uint32_t)kStreamMinPrefix && n - T <= (uint32_t)kTailMax;
    if (!stream) { /* (`consumed` of a general frame says why, for bev_debug_get_frame_info: 1 not eligible, 2 prefix too
                    * short, 3 tail too long, 4 a (row, strip) with more than kTailCap tail points) */
        if (tid == 0) b.info[f] = FrameInfo{0u, kFrameGeneral, !can ? 1u : (T < (uint32_t)kStreamMinPrefix ? 2u : 3u), 0u};
        return;
    }
    uint32_t *fest = b.est + (size_t)f * g.N * g.strips;
    uint32_t slot_last; /* of position T - 1 (in range and above the last sample's: checked above) */
    {
        const uint32_t rc = reinterpret_cast<const uint32_t *>(fp + (T - 1u))[5];
        slot_last = (rc & 0xffffu) * (uint32_t)g.H + (rc >> 16);
    }
    for (int i = tid; i < g.N * g.strips; i += kProbeThreads) {
        const int r = i / g.strips, st = i - r * g.strips;
        const long long want = (long long)r * g.H + (long long)st * kStripCols - 2; /* first slot of the strip's window */
        uint32_t est = 0u;
        if (want > (long long)samp[0]) {
            uint32_t lo = 0u, hi = m - 1u; /* largest k with samp[k] <= want */
            while (lo < hi) {
                const uint32_t mid = (lo + hi + 1u) >> 1;
                if ((long long)samp[mid] <= want) lo = mid; else hi = mid - 1u;
            }
            const uint32_t s0 = samp[lo];
            if (lo + 1u < m) {
                const uint32_t s1 = samp[lo + 1u];
                est = lo * kProbeStride + (uint32_t)(((unsigned long long)(want - s0) * kProbeStride) / (s1 - s0));
            } else if (want >= (long long)slot_last) { /* at or beyond the prefix's last point */
                est = want > (long long)slot_last ? T : T - 1u;
            } else { /* between the last sample and the prefix's last point (position T - 1) */
                const uint32_t p0 = lo * kProbeStride;
                est = p0 + (uint32_t)(((unsigned long long)(want - s0) * (T - 1u - p0)) / (slot_last - s0));
            }
        }
        fest[i] = est < T ? est : T;
    }

    /* The tail [T, n): too few points to be worth a pass of the order scan (scattered atomics run at a twentieth of the
     * rate of the scan's coalesced ones), and the stream walk has no winner table to look them up in.  They are listed
     * per (row, strip) — under every strip whose 256 virtual columns hold the slot: its own, a neighbour's halo, strip
     * 0's flat-index halo of the row below, the last strip's wrap-around halo — as column offset | input index << 8, in
     * any order; the walk drops them over the prefix's points of the same row, the last of several points of one slot
     * winning (it settles that per row, among at most kTailCap entries). */
    for (int i = tid; i < g.N * g.strips; i += kProbeThreads) tcnt[i] = 0u;
    if (tid == 0) overflow = 0u;
    __syncthreads();
    uint32_t *flist = b.tail_list + (size_t)f * g.N * g.strips * kTailCap;
    auto append = [&](int row, int strip, int off, uint32_t i) {
        const int bucket = row * g.strips + strip;
        const uint32_t pos = atomicAdd(&tcnt[bucket], 1u);
        if (pos < (uint32_t)kTailCap) flist[(size_t)bucket * kTailCap + pos] = (uint32_t)off | (i << 8);
        else overflow = 1u;
    };
    constexpr int kPer = 8; /* loads in flight per thread: a 5000-point tail is one trip */
    for (uint32_t i0 = T; i0 < n; i0 += (uint32_t)kProbeThreads * kPer) {
        uint32_t rc[kPer];
#pragma unroll
        for (int k = 0; k < kPer; ++k) {
            const uint32_t i = i0 + (uint32_t)kProbeThreads * k + tid;
            rc[k] = load_once(reinterpret_cast<const uint32_t *>(fp + (i < n ? i : n - 1u)) + 5);
        }
#pragma unroll
        for (int k = 0; k < kPer; ++k) {
            const uint32_t i = i0 + (uint32_t)kProbeThreads * k + tid;
            const int row = (int)(rc[k] & 0xffffu), col = (int)(rc[k] >> 16);
            if (i >= n || row >= g.N || col >= g.H) continue; /* :106-111 */
            const int st = col / kStripCols, c = col - st * kStripCols;
            append(row, st, c + 2, i);
            if (c < 2 && st > 0) append(row, st - 1, kStripCols + 2 + c, i);
            if (c >= kStripCols - 2 && st + 1 < g.strips) append(row, st + 1, c - (kStripCols - 2), i);
            if (col >= g.H - 2 && row + 1 < g.N) append(row + 1, 0, col - (g.H - 2), i);
            if (col < 2) {
                const int off = g.H + col - ((g.strips - 1) * kStripCols - 2);
                if (off < kStripThreads) append(row, g.strips - 1, off, i);
            }
        }
    }
    __syncthreads();
    uint32_t *fcnt = b.tail_cnt + (size_t)f * g.N * g.strips;
    for (int i = tid; i < g.N * g.strips; i += kProbeThreads) fcnt[i] = tcnt[i] < (uint32_t)kTailCap ? tcnt[i] : (uint32_t)kTailCap;
    /* a list that does not hold its (row, strip)'s tail points: the frame goes the general way (the scan repeats the
     * scatter of the tail among all the others) */
    if (tid == 0) b.info[f] = overflow ? FrameInfo{0u, kFrameGeneral, 4u, 0u} : FrameInfo{T, kFrameStream, 0u, 0u};
}

/* after the stream walk: a frame whose consumed points do not add up to its prefix, or with a failed check, is redone */
__global__ __launch_bounds__(256) void k_verdict(FrameInfo *info, int nf)
{
    const int f = blockIdx.x * 256 + threadIdx.x;
    if (f >= nf) return;
    FrameInfo fi = info[f];
    if (fi.mode == kFrameStream && (fi.failed != 0u || fi.consumed != fi.T)) info[f].mode = kFrameRedo;
}

/* ------------------------------------------------------------------------- */
/* getOrderedCloud, BatchMultiBevGen.cpp:102-116: bounds test + slot index;
 * "last point in input order wins" == max input index per slot.            */
constexpr int kSeenBits = 11; /* the walk's memo of listed BEV codes: 2048 entries, 8 KB of LDS */
constexpr int kScanPerThread = 4;
constexpr int kScanIdxBits = 10; /* 256 * kScanPerThread = 1024 points per block */
constexpr int kScanRowBins = 128; /* rows the LDS regrouping below can bin (more rows: plain path) */
__global__ __launch_bounds__(256) void k_order_scan(const bev_point_t *__restrict__ pts,
                                                    const FrameDesc *__restrict__ frames,
                                                    const FrameInfo *__restrict__ info, int pass,
                                                    uint32_t *__restrict__ winner, int N, int H, int S,
                                                    uint32_t tag_bits)
{
    const int f = blockIdx.y;
    const FrameDesc fd = frames[f];
    const uint32_t block0 = blockIdx.x * (256u * kScanPerThread);
    const uint32_t base = block0 + threadIdx.x;
    if (block0 >= fd.n_pts) return;
    /* which points of this frame this pass scatters: [first, n) */
    uint32_t first = 0u;
    if (info) {
        const FrameInfo fi = info[f];
        if (pass == 0) {
            if (fi.mode == kFrameStream) return; /* read in place by the stream walk; k_probe has listed its tail */
        } else if (fi.mode != kFrameRedo) {
            return;
        }
    } else if (pass != 0) {
        return;
    }
    if (block0 + 256u * kScanPerThread <= first) return;
    const bev_point_t *fp = pts + fd.in_offset;
    uint32_t slot[kScanPerThread];
    bool spread = false; /* does any wave-instruction's worth of 64 points straddle far-apart slots? */
    uint32_t rcw[kScanPerThread];
#pragma unroll
    for (int k = 0; k < kScanPerThread; ++k) { /* all loads in flight before anything is decoded: clamped address, no branch */
        const uint32_t i = base + 256u * k;
        rcw[k] = load_once(reinterpret_cast<const uint32_t *>(fp + (i < fd.n_pts ? i : fd.n_pts - 1u)) + 5); /* row | col << 16 */
    }
#pragma unroll
    for (int k = 0; k < kScanPerThread; ++k) {
        const uint32_t i = base + 256u * k;
        const uint32_t row = rcw[k] & 0xffffu, col = rcw[k] >> 16;
        slot[k] = (i >= first && i < fd.n_pts && row < (uint32_t)N && col < (uint32_t)H) ? row * (uint32_t)H + col
                                                                                         : 0xffffffffu; /* :106-111 ("< 0" is dead: u16) */
    }
    uint32_t *fw = winner + (size_t)f * S;
#pragma unroll
    for (int k = 0; k < kScanPerThread; ++k) {
        /* slots of a sorted cloud rise by ~1 per lane; a wave whose first and last valid lanes are more
         * than 4 rows apart is scattering (e.g. firing-order input: consecutive points = consecutive rows) */
        const unsigned long long vm = __ballot(slot[k] != 0xffffffffu);
        if (vm) {
            const int lo_lane = __ffsll((long long)vm) - 1, hi_lane = 63 - __clzll((long long)vm);
            const uint32_t a = __shfl(slot[k], lo_lane), z = __shfl(slot[k], hi_lane);
            const uint32_t d = a > z ? a - z : z - a;
            spread = spread || d > 4u * (uint32_t)H;
        }
    }
    __shared__ uint32_t any_spread;
    __shared__ uint32_t row_fill[kScanRowBins];
    /* (slot << kScanIdxBits | index within the block) regrouped by row; 4 B per point, not 8: LDS is what decides how many of these
     * blocks fit on a CU beside a k_cell_sums / k_bev_raster workgroup of another sub-batch */
    __shared__ uint32_t pairs[256 * kScanPerThread];
    static_assert(256 * kScanPerThread == (1 << kScanIdxBits), "bits of block-local index");
    if (threadIdx.x == 0) any_spread = 0u;
    __syncthreads();
    if (spread && (threadIdx.x & 63) == 0) any_spread = 1u;
    __syncthreads();
    if (any_spread == 0u || N > kScanRowBins || S > (1 << (32 - kScanIdxBits))) {
        /* coalesced already (or too many rows to bin): one atomicMax per point, in input order */
#pragma unroll
        for (int k = 0; k < kScanPerThread; ++k)
            if (slot[k] != 0xffffffffu) atomicMax(&fw[slot[k]], tag_bits | (base + 256u * k + 1u));
        return;
    }
    /* Scattering input: regroup the block's (slot, index) pairs by row in LDS (atomicMax is order-free,
     * so an unstable counting sort is enough); a wave then sends its atomics to one row and nearby
     * columns instead of 64 different rows — scattered device atomics run ~15x slower than contiguous ones. */
    for (int r = threadIdx.x; r < kScanRowBins; r += 256) row_fill[r] = 0u;
#pragma unroll
    for (int k = 0; k < kScanPerThread; ++k) pairs[threadIdx.x + 256u * k] = 0xffffffffu; /* empty */
    __syncthreads();
    uint32_t rank[kScanPerThread];
#pragma unroll
    for (int k = 0; k < kScanPerThread; ++k)
        rank[k] = slot[k] != 0xffffffffu ? atomicAdd(&row_fill[slot[k] / (uint32_t)H], 1u) : 0u;
    __syncthreads();
    /* exclusive scan of the row counts (N <= 128 bins: two per thread of the first wave) */
    if (threadIdx.x < 64) {
        const uint32_t c0 = row_fill[2 * threadIdx.x], c1 = row_fill[2 * threadIdx.x + 1];
        uint32_t incl = c0 + c1;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t v = __shfl_up(incl, d);
            if ((int)threadIdx.x >= d) incl += v;
        }
        row_fill[2 * threadIdx.x] = incl - c0 - c1;
        row_fill[2 * threadIdx.x + 1] = incl - c1;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kScanPerThread; ++k)
        if (slot[k] != 0xffffffffu)
            pairs[row_fill[slot[k] / (uint32_t)H] + rank[k]] = (slot[k] << kScanIdxBits) | (threadIdx.x + 256u * k);
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kScanPerThread; ++k) {
        const uint32_t j = threadIdx.x + 256u * k;
        const uint32_t pr = pairs[j];
        if (pr != 0xffffffffu) atomicMax(&fw[pr >> kScanIdxBits], tag_bits | (blockIdx.x * (256u * kScanPerThread) + (pr & ((1u << kScanIdxBits) - 1u)) + 1u));
    }
}

/* ------------------------------------------------------------------------- */
template <bool kIdentity>
struct SlotFetch {
    const uint32_t *win;      /* frame's winner table (unused in identity mode) */
    const bev_point_t *pts;   /* frame's input points, or the ordered cloud itself */
    __device__ __forceinline__ XYZI operator()(long long flat) const
    {
        long long idx = flat;
        if (!kIdentity) {
            const uint32_t w = win[flat];
            if (w == 0u) return XYZI{0.f, 0.f, 0.f, 0.f}; /* untouched slot: value-initialised, :98 */
            idx = (long long)w - 1;
        }
        const float4 a = *reinterpret_cast<const float4 *>(pts + idx);
        const float it = reinterpret_cast<const float *>(pts + idx)[4];
        return XYZI{a.x, a.y, a.z, it};
    }
};

/* ------------------------------------------------------------------------- */
/* getOrderedCloud gather + markGroundPoints phase A, as a COLUMN WALK.
 *
 * A workgroup owns kStripCols (252) adjacent columns of one frame plus two halo
 * columns on each side (256 threads) and walks the rows 0 .. N-1.  Thread tid
 * sits on virtual column v = strip*252 + tid - 2 and, in row r, on flat slot
 * index r*H + v (v >= H wraps to v - H in the SAME row, v < 0 is the flat index
 * r*H + v, i.e. the tail of row r-1 — exactly the two index rules of
 * BatchMultiBevGen.cpp:146-154).  Consequences:
 *   - every input point is loaded exactly once (winner -> point), rows arrive as
 *     8 KiB coalesced pieces, the next row's loads are issued a row ahead;
 *   - the phase-A stencil needs no second pass: "upper" is the thread's own
 *     previous row (registers), its +-2 fallbacks are the neighbours' previous
 *     rows (wave shuffles, LDS only across wave edges), row-2 is the thread's
 *     own row before that;
 *   - status s[r] is evaluated ONCE per slot; ground_mat(r-1) follows from
 *     s[r-1] and s[r] (closed form in bev_exact.h), so row r-1 is finished while
 *     row r is being evaluated, and row r-2 is written out (one barrier per row
 *     covers both the LDS row buffer and the candidate counts).
 * Candidates of one (row, strip) are compacted in column order into their own
 * segment; segments enumerate (row, strip) in row-major order, so the
 * concatenation of all segments is slot order — what phase B's accumulation
 * order needs. */
struct PendingRow {
    Half lo, hi;
    uint32_t code;
    int status;      /* s[row] (kInvalid / kSteep / kGround); kSteep for rows that are not tested */
    int gflag;       /* ground_mat(row) at the end of phase A */
    bool pred;       /* candidate the walk expects phase C to un-ground (see "provisional labels" below) */
    uint32_t key;    /* candidate key (bev_exact.h), valid when gflag == 1 */
};

/* Narrow workspace streams (candidate keys / heights, code lists) are written with the default cache policy: a row's
 * pieces from the four waves are contiguous, so L2 merges them into whole lines before they leave (with `nt` every
 * piece left as partial lines: the walk wrote 6.5 MB per frame where 5.8 MB were needed).  The ordered cloud's stores
 * (whole 1 KiB pieces of a wave's 2 KiB row, see the write-out below) keep `nt`: with the default policy the pipeline is
 * 4-6 % slower. */
template <class T>
__device__ __forceinline__ void store_ws(T *p, T v) { *p = v; }

enum : int { kSrcGather = 0, kSrcIdentity = 1, kSrcStream = 2 };
/* Gives a wave-uniform value a scalar register of its own.  Kernel arguments arrive as 8-register tuples; the walk keeps
 * more uniform values alive than there are scalar registers, and the compiler spills and restores whole tuples (through
 * lanes of a vector register, one VALU instruction per dword): the raster constants came back eight at a time around
 * every use (a third of the STATIC vector instructions of the row loop; executed, about 1 %). */
template <class T>
__device__ __forceinline__ T own_sgpr(T v)
{
    asm volatile("" : "+s"(v));
    return v;
}
template <int kSrc>
__global__ __launch_bounds__(kStripThreads, kSrc == kSrcStream ? 3 : 4) void k_strip_ground(BatchPtrs b, Geometry g, int nf, uint32_t want_mode)
{
    constexpr bool kIdentity = kSrc == kSrcIdentity, kStream = kSrc == kSrcStream;
    /* the strips of a frame share halo columns and the lines at their seams: one XCD (one L2) per frame */
    int f, strip;
    if (!map_block_xcd(blockIdx.x, nf, g.strips, f, strip)) return;
    if (!kIdentity && b.info && b.info[f].mode != want_mode) return; /* another launch of this kernel has the frame */
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int N = own_sgpr(g.N), H = own_sgpr(g.H), lo_row = own_sgpr(g.N - g.G);
    const size_t frame_off = (size_t)f * g.S;
    const int bands = g.raster_bands;

    /* lanes two to the right / left (wrapping inside the wave; the edge lanes are patched from LDS), one to the left */
    const int sh_right = ((lane + 2) & 63) << 2, sh_left = ((lane - 2) & 63) << 2, sh_left1 = ((lane - 1) & 63) << 2;
    auto lane_from = [&](int sel, uint32_t x) -> uint32_t { return (uint32_t)__builtin_amdgcn_ds_bpermute(sel, (int)x); };
    auto lane_from_f = [&](int sel, float x) -> float { return __uint_as_float((uint32_t)__builtin_amdgcn_ds_bpermute(sel, (int)__float_as_uint(x))); };
    const int v = strip * kStripCols + tid - 2;                      /* virtual column */
    const bool provider = (v < H + 2) && (v >= 0 || strip == 0);     /* has a point to load */
    const bool outcol = tid >= 2 && tid < 2 + kStripCols && v < H;   /* owns column v's outputs */
    const int vcol = v >= H ? v - H : v;                             /* wrap; v < 0 keeps the flat rule */
    /* stream source: (row | col << 16) of the thread's slot in row 0 (the tail of row -1 for strip 0's left halo: never
     * there; column 0xffff for threads without a slot: never there either) */
    const uint32_t slot_rc = !provider ? 0xffff0000u : (v < 0 ? ((uint32_t)(H + v) << 16) + 0xffffffffu : (uint32_t)vcol << 16);

    const bev_point_t *fpts = kIdentity ? (b.pts + frame_off) : (b.pts + b.frames[f].in_offset);
    const uint32_t *fwin = b.winner + frame_off;

    /* Neighbour exchange: lanes l+-2 of the same wave are reached with shuffles; only the two edge
     * lanes on each side of a wave go through LDS (768 B instead of a 12 KiB row buffer, so that these
     * workgroups can share a CU with the back end's workgroups of the other lane). */
    constexpr int kWaves = kStripThreads / 64;
    __shared__ float4 edge[3][kWaves][4];                  /* rows r, r-1, (r-2): lanes 0, 1, 62, 63 of every wave */
    __shared__ u32x4 xpose[kWaves][128];                   /* a wave's 64 finished points, to be stored as two whole KiB */
    __shared__ uint32_t wave_cnt[2][kWaves];               /* per-wave candidate counts of the row being written */
    __shared__ uint32_t band_cursor[kMaxBands];            /* entries already in this strip's code list of each band */
    __shared__ uint8_t band_tab[512];                      /* x bin -> raster band */
    constexpr int kSeenB = kStream ? kSeenBits - 1 : kSeenBits; /* (the stream source needs the LDS for its row buffers) */
    __shared__ uint32_t seen[1 << kSeenB];                 /* direct-mapped memo of codes this strip has already listed */
    __shared__ int edge_x[kGridRows], edge_y[kGridCols];   /* BEV bin of every ground-grid row's / column's lower edge */
    /* stream source (see below): three rows of points by column offset, the first wave's extra window positions, the
     * second wave's tail points, the estimates and tail counts of every row */
    __shared__ u32x4 rowbuf[3][2][kStream ? kStripThreads : 1];
    __shared__ u32x4 xwin[kStream ? 2 : 1][2][kStream ? 64 : 1];
    __shared__ u32x4 twin[kStream ? 2 : 1][2][kStream ? 64 : 1];
    __shared__ uint32_t lastrc[kStream ? 2 : 1][kStripThreads / 64];
    __shared__ int est_l[2][kStream ? kStreamMaxRows : 1];
    __shared__ uint32_t tcnt_l[kStream ? kStreamMaxRows : 1];
    if (tid < kMaxBands) band_cursor[tid] = 0u;
    for (int k = tid; k < (1 << kSeenB); k += kStripThreads) seen[k] = kSkip;
    for (int x = tid; x < g.rp.mat_size; x += kStripThreads) band_tab[x] = (uint8_t)raster_band_of(x, g.rp);
    if (tid < kGridRows) edge_x[tid] = cell_edge_bin(tid, 75.0f, g.rp);
    else if (tid < kGridRows + kGridCols) edge_y[tid - kGridRows] = cell_edge_bin(tid - kGridRows, 50.0f, g.rp);
    if (kStream) {
        /* no entry of the row buffers may look like a point of the slot it stands for: row 0xffff does not exist */
        for (int k = 0; k < 3; ++k) rowbuf[k][1][tid] = u32x4{0u, 0xffffffffu, 0u, 0u};
        /* estimates and tail counts into LDS once: a global load the compiler sees inside the row loop would bring back
         * the vmcnt(0) that the counted waits below are there to avoid */
        const uint32_t *fe = b.est + (size_t)f * N * g.strips;
        const uint32_t *fc = b.tail_cnt + (size_t)f * N * g.strips;
        for (int r = tid; r < N; r += kStripThreads) {
            est_l[0][r] = (int)fe[r * g.strips + strip];
            est_l[1][r] = (int)fe[r * g.strips];
            tcnt_l[r] = fc[r * g.strips + strip];
        }
    }

    /* Winner words are loaded UNCONDITIONALLY from a clamped address and decoded only when they are used, two rows
     * later: a predicated load whose result is decoded on the spot makes the compiler branch around the load and wait
     * for it — with vmcnt(0), i.e. for every point load in flight as well — inside the branch, once per row (that was
     * the shape of this loop in round 1: the software pipeline below existed on paper only). */
    auto has_slot = [&](int r) -> bool { return provider && r < N && r * H + vcol >= 0; };
    auto load_winner_raw = [&](int r, uint32_t &raw) {
        raw = 0u;
        if (kIdentity || kStream) return; /* the stream source has no winner table */
        const int fl = has_slot(r) ? r * H + vcol : 0;
        raw = load_once(&fwin[fl]);
    };
    auto winner_of = [&](int r, uint32_t raw) -> uint32_t { /* input index + 1 of slot (r, this column), 0 = empty */
        if (!has_slot(r)) return 0u;
        if (kIdentity) return (uint32_t)(r * H + vcol) + 1u;
        return winner_index(raw, b.win_tag, b.win_shift);
    };

    /* the same for the points: an empty slot loads a dummy (the first point of this frame's OUTPUT, always allocated,
     * one cached line) and is zeroed when the row is consumed, so that every iteration issues exactly three loads and
     * the compiler can wait for "all but the last six" instead of for everything */
    const Half *dummy = reinterpret_cast<const Half *>(b.ordered + frame_off);
    auto load_point = [&](uint32_t w, u32x4 &lo, u32x4 &hi) {
        if (kStream) return; /* window positions instead: issue_P below */
        const Half *src = w != 0u ? reinterpret_cast<const Half *>(fpts + (w - 1u)) : dummy;
        lo = *reinterpret_cast<const u32x4 *>(src);
        hi = *reinterpret_cast<const u32x4 *>(src + 1);
    };

    /* software pipeline: while row r is handled, the points of rows r+1 .. r+kDepth and the raw winner words of the
     * kDepth rows after those are in flight.  gfx950 counts loads and stores on ONE counter (vmcnt) and they complete
     * out of order with respect to each other, so with stores pending the compiler waits for "everything" before a
     * loaded value is used; what the second stage still buys is that row r+2's loads are issued before row r's stores.
     * The stages live in small arrays indexed by r mod 3 / r mod 2 and the row loop is unrolled with compile-time
     * indices: rotating the stages through variables instead ("next = next2") makes the compiler copy registers that a
     * load is still writing, and wait for that load — the newest one — every row. */
    constexpr int kDepth = 2;
    u32x4 plo[3], phi[3];  /* point of row r at [r % 3] */
    bool pfull[3];         /* the slot of that row holds a point (else: the dummy was loaded) */
    uint32_t wraw[2];      /* raw winner word of row r at [r % 2] */
    plo[0] = plo[1] = plo[2] = phi[0] = phi[1] = phi[2] = u32x4{0u, 0u, 0u, 0u};
    {
        uint32_t r0, r1;
        load_winner_raw(0, r0);
        load_winner_raw(1, r1);
        load_winner_raw(2, wraw[0]);
        load_winner_raw(3, wraw[1]);
        const uint32_t w0 = winner_of(0, r0), w1 = winner_of(1, r1);
        pfull[0] = w0 != 0u;
        pfull[1] = w1 != 0u;
        pfull[2] = false;
        load_point(w0, plo[0], phi[0]);
        load_point(w1, plo[1], phi[1]);
    }

    /* ---- stream source (k_probe took the first T input points for sorted, and listed the rest per (row, strip)) ----
     * Row rho's points of this strip's 256 virtual columns are consecutive in the input; they start near
     * est[rho][strip] (interpolated from sampled points).  Every thread owns ONE window position (est - slack + tid; the
     * first wave also the window's last 32 positions and, for the last strip's two wrap-around halo columns, 32
     * positions at the row's start), loads it coalesced and in place four steps before the row is needed, looks at the
     * (row, col) the point carries and, two steps later, drops the point into the row buffer at its column offset.
     * The tail points of the (row, strip) — at most kTailCap, listed by k_probe, none shadowed by a later one — follow
     * one step later through the second wave, so that they overwrite prefix points of the same slot as the reference's
     * scatter would.  One step after that the owner of each column reads its slot from the row buffer; an entry whose
     * (row, col) is not the slot's own is an empty slot.  Steps are separated by the row barrier, three row buffers
     * rotate, nothing but LDS is shared.
     * Verification (results must not depend on the estimate): a position that holds a point of the strip's OWN columns
     * counts it and checks that the position before it lies in the prefix and has a smaller slot (across wave edges
     * one step later, through LDS).  When all T prefix points of a frame have been counted exactly once and no check has
     * failed, the prefix is strictly ascending, every point was where its strip looked, and the result is what
     * getOrderedCloud's scatter gives; otherwise k_verdict sends the frame through the general kernels again.
     * Loads are asm / LDS-DMA, issued one step (the position loads: two steps) before they are used and waited for with
     * ONE s_waitcnt vmcnt(0) at the top of a step — where the row's stores, issued right after the previous barrier, have
     * had a whole step to complete as well.  (Counting — vmcnt(2), "all but the newest two" — measured the same.)  What
     * matters is that the compiler never sees these loads: a load it sees in a loop that also stores is waited for with
     * vmcnt(0) wherever its result is first touched, i.e. in the middle of the step. */
    const uint32_t T = kStream ? b.info[f].T : 0u;
    const bool last_strip = strip == g.strips - 1;
    uint32_t consumed = 0u, failed = 0u;
    const int first_col = strip * kStripCols - 2; /* virtual column of offset 0 */
    /* tail entry of this lane, rows rho at [rho % 3]: te as the asm load delivers it (written by nothing else: a register
     * that a load is still filling must not be redefined on one side of a branch — the compiler merges the two
     * definitions through a copy, and a copy made while the load is in flight carries the stale word), ts after the
     * second wave has settled it */
    uint32_t te[3] = {0u, 0u, 0u}, ts[3] = {0xffffffffu, 0xffffffffu, 0xffffffffu};
    bool dneed = false;                           /* lane 0: the check against the previous wave's last position is due */
    int dflat = 0;
    const uint32_t *ftail = kStream ? b.tail_list + ((size_t)f * N * g.strips + strip) * kTailCap : nullptr;
    const int tail_stride = g.strips * kTailCap;  /* words from one row's list to the next */
    auto clamp_row = [&](int rho) -> int { return rho < N ? rho : N - 1; };
    auto issue_P = [&](auto STG, int rho) { /* the thread's window position of row rho */
        constexpr int sg = decltype(STG)::value;
        const int q = est_l[0][clamp_row(rho)] - kStreamSlack + tid;
        const Half *src = reinterpret_cast<const Half *>(fpts + (q >= 0 && q < (int)T ? q : 0));
        ld128(plo[sg], src);
        ld128_16(phi[sg], src);
    };
    auto issue_X = [&](int rho) { /* first wave: lanes 0 .. 31 positions 256 .. 287, lanes 32 .. 63 the row's first positions */
        const int rc = clamp_row(rho);
        const int q = lane < 32 ? est_l[0][rc] - kStreamSlack + kStripThreads + lane : est_l[1][rc] - (kStreamSlack - 2) + (lane - 32);
        const Half *src = reinterpret_cast<const Half *>(fpts + (q >= 0 && q < (int)T ? q : 0));
        glds16(src, __builtin_amdgcn_readfirstlane(lds_addr(&xwin[rho & 1][0][0])));
        glds16(src + 1, __builtin_amdgcn_readfirstlane(lds_addr(&xwin[rho & 1][1][0])));
    };
    auto issue_TE = [&](auto STG, int rho) { /* second wave: this lane's entry of the row's tail list (stale past the count) */
        constexpr int sg = decltype(STG)::value;
        ld32_nt(te[sg], ftail + (size_t)clamp_row(rho) * tail_stride + lane);
    };
    /* second wave: which of row rho's listed tail points are the last of their slot (getOrderedCloud's scatter keeps the
     * last writer, BatchMultiBevGen.cpp:112).  Every entry counts itself in at its column offset (256 LDS counters in
     * the buffer the row's tail points are about to be loaded into: it is idle right now); only when some offset has
     * been taken twice — a few times per frame — those entries are compared with the others of their offset.  Entries
     * that lose, and lanes past the list's count, become 0xffffffff. */
    auto settle = [&](auto STG, int rho) {
        constexpr int sg = decltype(STG)::value;
        const bool on = rho < N && (uint32_t)lane < tcnt_l[clamp_row(rho)];
        const uint32_t e = te[sg];
        const uint32_t off = e & 0xffu, idx = e >> 8;
        uint32_t *cnt = reinterpret_cast<uint32_t *>(&twin[rho & 1][0][0]); /* [256] */
        twin[rho & 1][0][lane] = u32x4{0u, 0u, 0u, 0u};
        const uint32_t before = on ? atomicAdd(&cnt[off], 1u) : 0u; /* (LDS operations of one wave execute in order) */
        bool dead = !on;
        unsigned long long crowd = __ballot(before != 0u);
        while (crowd) { /* wave-uniform */
            const int j = __ffsll((long long)crowd) - 1;
            crowd &= crowd - 1ull;
            const uint32_t ej = (uint32_t)__builtin_amdgcn_readlane((int)e, j);
            const bool same = on && (ej & 0xffu) == off;
            if (same && (ej >> 8) > idx) dead = true;                          /* lane j's point comes later than this one */
            if (__ballot(same && idx > (ej >> 8)) != 0ull && lane == j) dead = true; /* ... or some other after lane j's */
        }
        ts[sg] = dead ? 0xffffffffu : e;
    };
    auto issue_TP = [&](auto STG, int rho) { /* second wave: the (settled) tail points of row rho, by LDS-DMA */
        constexpr int sg = decltype(STG)::value;
        const bool on = ts[sg] != 0xffffffffu;
        const Half *src = reinterpret_cast<const Half *>(fpts + (on ? (ts[sg] >> 8) : 0u));
        glds16(src, __builtin_amdgcn_readfirstlane(lds_addr(&twin[rho & 1][0][0])));
        glds16(src + 1, __builtin_amdgcn_readfirstlane(lds_addr(&twin[rho & 1][1][0])));
    };
    /* one window position: into the row buffer, counted and checked.  `sflat` is the position's slot, or INT_MAX when the
     * position is outside the prefix or its point outside the range image (such a predecessor fails every check) */
    const int own_cols = (H - first_col - 2) < kStripCols ? (H - first_col - 2) : kStripCols; /* own columns of this strip */
    const int row_span = (H - first_col) < kStripThreads ? (H - first_col) : kStripThreads;  /* offsets that belong to the row */
    auto slot_or_max = [&](int q, uint32_t rcw) -> int {
        const uint32_t row = rcw & 0xffffu, col = rcw >> 16;
        const bool valid = (unsigned)q < T && row < (uint32_t)N && col < (uint32_t)H;
        return valid ? (int)row * H + (int)col : 0x7fffffff;
    };
    auto place = [&](int rho, int q, const u32x4 &lo, const u32x4 &hi, bool first_of_group, int dwave) {
        const int sflat = slot_or_max(q, hi.y);
        const int off = (int)((uint32_t)sflat - (uint32_t)(rho * H + first_col));
        /* (a window of the last strip runs into the next row: those points are not this row's wrap-around halo) */
        const bool inr = (unsigned)off < (unsigned)row_span;
        if (inr) {
            rowbuf[rho % 3][0][off] = lo;
            rowbuf[rho % 3][1][off] = hi;
        }
        const bool own = (unsigned)(off - 2) < (unsigned)own_cols;
        consumed += own ? 1u : 0u;
        /* its predecessor in the input must lie in the prefix and have a smaller slot */
        const bool pok = (int)lane_from(sh_left1, (uint32_t)sflat) < sflat;
        /* (straight-line: `if (a) failed = 1; else dflat = flat;` becomes a store through a selected pointer, and the
         * variables stay in scratch memory — a load the compiler waits for with vmcnt(0) every step) */
        const bool chk = own && q > 0;
        const bool bad = chk && (first_of_group ? dwave < 0 /* window position 0: the estimate was too high */ : !pok);
        const bool defer = chk && first_of_group && dwave >= 0;
        failed |= bad ? 1u : 0u;
        dneed = dneed | defer;
        dflat = defer ? sflat : dflat;
    };
    auto scatter = [&](auto STG, int rho) { /* the prefix positions of row rho -> row buffer */
        constexpr int sg = decltype(STG)::value;
        /* last step's open check: the lane's point against the last position of the wave before */
        {
            const int pflat = (int)lastrc[(rho - 1) & 1][wv == 0 ? kStripThreads / 64 - 1 : wv - 1];
            failed |= (dneed && !(pflat < dflat)) ? 1u : 0u;
            dneed = false;
        }
        if (rho >= N) return;
        const int est = est_l[0][rho];
        {
            const int q = est - kStreamSlack + tid;
            if (lane == 63) lastrc[rho & 1][wv] = (uint32_t)slot_or_max(q, phi[sg].y);
            place(rho, q, plo[sg], phi[sg], lane == 0, wv == 0 ? -1 : wv - 1); /* lane 0 of a later wave: checked one step later */
        }
        if (wv == 0) { /* wave-uniform */
            const u32x4 xl = xwin[rho & 1][0][lane], xh = xwin[rho & 1][1][lane];
            if (lane < 32) {
                /* positions 256 .. 287; lane 0 follows the last wave's last position */
                place(rho, est - kStreamSlack + kStripThreads + lane, xl, xh, lane == 0, kStripThreads / 64 - 1);
            } else if (last_strip) { /* slots rho*H and rho*H + 1 as the wrap-around halo columns H, H + 1 */
                const int q = est_l[1][rho] - (kStreamSlack - 2) + (lane - 32);
                const uint32_t rcw = xh.y;
                const uint32_t row = rcw & 0xffffu, col = rcw >> 16;
                const int off = H + (int)col - first_col;
                if (q >= 0 && q < (int)T && row == (uint32_t)rho && col < 2u && (unsigned)off < (unsigned)kStripThreads) {
                    rowbuf[rho % 3][0][off] = xl;
                    rowbuf[rho % 3][1][off] = xh;
                }
            }
        }
    };
    auto tail_scatter = [&](auto STG, int rho) { /* second wave: the listed tail points of row rho over the prefix's */
        constexpr int sg = decltype(STG)::value;
        if (rho >= N) return;
        if (ts[sg] != 0xffffffffu) {
            const int off = (int)(ts[sg] & 0xffu);
            rowbuf[rho % 3][0][off] = twin[rho & 1][0][lane];
            rowbuf[rho % 3][1][off] = twin[rho & 1][1][lane];
        }
    };
    if (kStream) { /* prologue: rows 0 and 1 placed, row 0's tail over them; the steady state's loads under way */
        using I0 = std::integral_constant<int, 0>;
        using I1 = std::integral_constant<int, 1>;
        using I2 = std::integral_constant<int, 2>;
        lds_barrier(); /* est_l, tcnt_l, row buffers */
        issue_P(I0{}, 0);
        issue_P(I1{}, 1);
        if (wv == 0) {
            issue_X(0);
            issue_X(1);
        }
        issue_TE(I0{}, 0);
        issue_TE(I1{}, 1);
        issue_TE(I2{}, 2);
        wait_loads<0>(plo[0], phi[0], te[0]);
        wait_loads<0>(plo[1], phi[1], te[1]);
        asm volatile("" : "+v"(te[2]));
        if (wv == 1) {
            settle(I0{}, 0);
            settle(I1{}, 1);
            issue_TP(I0{}, 0);
            issue_TP(I1{}, 1);
        }
        scatter(I0{}, 0);
        lds_barrier(); /* lastrc of row 0 */
        scatter(I1{}, 1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        lds_barrier(); /* rows 0 and 1 placed */
        if (wv == 1) tail_scatter(I0{}, 0);
        if (wv == 0) issue_X(2);
        issue_P(I2{}, 2);
        issue_P(I0{}, 3);
        lds_barrier(); /* row 0 complete */
    }

    /* rows r-1 (ground flag still open) and r-2 (ready to write) live in pr[(r-1) % 3], pr[(r-2) % 3]; row r takes the
     * third record.  (Handing them on through variables — p2 = p1; p1 = cur — cost 36 register moves per row.) */
    PendingRow pr[3] = {};

    float zref = __uint_as_float(0x7fc00000u); /* height of the column's last candidate taken for ground (NaN: none yet) */

    const size_t cand_base = (size_t)f * g.segs * kSeg;
    uint2 *const fcand = own_sgpr(b.cand + cand_base);
    uint32_t *const fncand = own_sgpr(b.ncand + (size_t)f * g.segs);
    const uint32_t code_cap = own_sgpr(g.code_cap);
    uint32_t *const flist = own_sgpr(b.code_main + ((size_t)f * g.emitters + strip) * bands * (size_t)code_cap);
    bev_point_t *const fordered = own_sgpr(b.ordered + frame_off);
    int8_t *const fgm = own_sgpr(b.gm ? b.gm + frame_off : nullptr);
    const int strips = own_sgpr(g.strips);
    RasterParams rp = g.rp; /* only the fields the BEV code needs stay alive */
    rp.max_range_f = own_sgpr(rp.max_range_f);
    rp.interval = own_sgpr(rp.interval);
    rp.inv_interval = own_sgpr(rp.inv_interval);
    rp.height_res = own_sgpr(rp.height_res);
    rp.inv_height_res = own_sgpr(rp.inv_height_res);
    rp.lidar_to_ground = own_sgpr(rp.lidar_to_ground);
    rp.mat_size = own_sgpr(rp.mat_size);
    rp.n_layers = own_sgpr(rp.n_layers);

    /* one row; I = r mod 6 (r mod 2 at depth 1) at compile time */
    auto row_step = [&](auto I, const int r) {
        constexpr int ic = decltype(I)::value % (kDepth + 1);              /* stage holding row r */
        constexpr int in = (decltype(I)::value + kDepth) % (kDepth + 1);   /* stage that takes row r + kDepth */
        constexpr int wu = (decltype(I)::value + kDepth) % 2, wl = decltype(I)::value % 2; /* winner word used / reloaded */
        PendingRow &p0 = pr[decltype(I)::value % 3], &p1 = pr[(decltype(I)::value + 2) % 3], &p2 = pr[(decltype(I)::value + 1) % 3];
        const XYZI prev{__uint_as_float(p1.lo.w[0]), __uint_as_float(p1.lo.w[1]), __uint_as_float(p1.lo.w[2]), __uint_as_float(p1.hi.w[0])};
        const XYZI prevprev{__uint_as_float(p2.lo.w[0]), __uint_as_float(p2.lo.w[1]), __uint_as_float(p2.lo.w[2]), __uint_as_float(p2.hi.w[0])};
        const int par = r & 1;
        Half cur_lo, cur_hi;
        if (kStream) {
            /* everything issued in earlier steps has arrived */
            constexpr int s2 = (decltype(I)::value + 2) % 3, s1 = (decltype(I)::value + 1) % 3, s0 = decltype(I)::value % 3;
            /* row r from its buffer (requested before the wait for the global loads: it does not depend on them): the
             * entry is the slot's own point, or the slot is empty (value-initialised, BatchMultiBevGen.cpp:98) */
            const u32x4 a = rowbuf[s0][0][tid], c = rowbuf[s0][1][tid];
            wait_loads<0>(plo[s2], phi[s2], te[s2]);
            const uint32_t want = slot_rc + (uint32_t)r; /* row | col << 16 of this thread's slot in row r */
            const bool hit = r < N && c.y == want;
            cur_lo = Half{{hit ? a.x : 0u, hit ? a.y : 0u, hit ? a.z : 0u, hit ? a.w : 0u}};
            cur_hi = Half{{hit ? c.x : 0u, hit ? c.y : 0u, hit ? c.z : 0u, hit ? c.w : 0u}};
            scatter(std::integral_constant<int, s2>{}, r + 2);
            if (wv == 1) tail_scatter(std::integral_constant<int, s1>{}, r + 1);
            if (wv == 0) issue_X(r + 3);
            if (wv == 1) settle(std::integral_constant<int, s2>{}, r + 2);
            /* (every wave issues this load although only the second one uses it: a register that an asm load is still
             * writing must not be defined on one side of a branch only — the compiler then merges it with its old
             * value through a copy, and a copy made while the load is in flight carries the stale word) */
            issue_TE(std::integral_constant<int, s0>{}, r + 3);
            if (wv == 1) issue_TP(std::integral_constant<int, s2>{}, r + 2);
            issue_P(std::integral_constant<int, s1>{}, r + 4);
        } else {
            cur_lo = Half{{plo[ic].x, plo[ic].y, plo[ic].z, plo[ic].w}};
            cur_hi = Half{{phi[ic].x, phi[ic].y, phi[ic].z, phi[ic].w}};
            if (!pfull[ic]) { /* untouched slot: value-initialised, BatchMultiBevGen.cpp:98 */
                cur_lo = Half{{0, 0, 0, 0}};
                cur_hi = Half{{0, 0, 0, 0}};
            }
        }
        {
            const uint32_t wn = winner_of(r + kDepth, wraw[wu]);
            pfull[in] = wn != 0u;
            load_point(wn, plo[in], phi[in]);
        }
        load_winner_raw(r + 2 * kDepth, wraw[wl]);

        const XYZI cur{__uint_as_float(cur_lo.w[0]), __uint_as_float(cur_lo.w[1]), __uint_as_float(cur_lo.w[2]),
                       __uint_as_float(cur_hi.w[0])};
        if (lane < 2 || lane >= 62) edge[r % 3][wv][lane < 2 ? lane : lane - 60] = make_float4(cur.x, cur.y, cur.z, cur.i);
        /* candidates of row r-2: every wave publishes its count, the write-out after the barrier ranks them */
        const bool cand2 = outcol && p2.gflag == 1;
        const unsigned long long mc = __ballot(cand2);
        if (lane == 0) wave_cnt[par][wv] = (uint32_t)__popcll(mc);
#ifndef BEV_EXP_NOBARRIER
        lds_barrier();
#endif

        /* ---- write out row r-2 (its per-wave counts were published before the barrier) ---- */
        /* (first thing after the barrier: the stores then have the whole status computation to complete in — gfx950 counts
         * them on the same counter as the loads, and the wait at the top of the next step would otherwise sit right
         * behind them) */
        if (r >= 2) {
            const int q = r - 2;
            const bool is_cand = cand2;
            const int rr = q - (lo_row - 1);        /* only rows lo-1 .. N-1 can hold candidates */
            if (rr >= 0) {
                uint32_t before = 0, total = 0;
#pragma unroll
                for (int w = 0; w < kWaves; ++w) {
                    const uint32_t c = wave_cnt[par][w];
                    if (w < wv) before += c;
                    total += c;
                }
                const uint32_t seg = (uint32_t)(rr * strips + strip);
                if (is_cand) {
                    /* the earlier waves' candidates, the earlier lanes' */
                    uint32_t rank = before + (uint32_t)__popcll(mc & ((1ull << lane) - 1ull));
#ifdef BEV_EXP_NOBARRIER /* timing experiment only: results are wrong, accesses stay in range */
                    rank &= (uint32_t)kSeg - 1u;
#endif
                    const uint32_t at = seg * (uint32_t)kSeg + rank; /* < 2^32: a frame's segments hold fewer slots than S */
                    store_ws(&fcand[at], make_uint2(p2.key, p2.lo.w[2])); /* key | height */
                }
                if (tid == 2) fncand[seg] = total;
            }
            /* BEV code of the slot.  A slot that is not a candidate has its final label, so its code is final too: it
             * is appended to this strip's list of the raster band its x bin falls into (the order inside a list does
             * not matter: an LDS cursor per band).  Candidates' codes travel in their keys.  A lane whose left
             * neighbour appends the very same code skips (near the sensor dozens of consecutive returns share a bin). */
            {
                bool has = outcol && !is_cand && p2.code != kSkip;
                const uint32_t left_code = lane_from(sh_left1, p2.code);
                const bool left_has = lane_from(sh_left1, has ? 1u : 0u) != 0u;
                if (lane > 0 && left_has && left_code == p2.code) has = false;
                /* ... and so does one whose code this strip has listed before and still remembers (rings hit the same
                 * cells at the same heights again and again: a HDL_64E frame lists 74 k codes of which 24 k are
                 * distinct).  The rasters are idempotent, so a stale or racing memo entry only costs a duplicate. */
                if (has) {
                    const uint32_t slot = (p2.code * 0x9E3779B1u) >> (32 - kSeenB);
                    if (seen[slot] == p2.code) has = false;
                    else seen[slot] = p2.code;
                }
                if (has) {
                    const int band = band_tab[code_x(p2.code)];
                    const uint32_t pos = atomicAdd(&band_cursor[band], 1u);
                    store_ws(&flist[(uint32_t)band * code_cap + pos], p2.code);
                }
            }
            {
                /* The 64 points of a wave are 2 KiB of consecutive bytes of the output.  Stored as they sit in the
                 * registers — the low halves with one instruction, the high halves with another — every 128-byte line
                 * leaves the CU in two instalments of four 16-byte pieces, and L2 writes some lines back in between.
                 * Through 2 KiB of LDS (wave-private, no barrier) each instruction stores 1 KiB of whole lines instead. */
                Half hi = p2.hi;
                const bool as_ground = is_cand && !p2.pred;
                if (as_ground) hi.w[3] &= 0xffff0000u; /* label = 0, BatchMultiBevGen.cpp:245 (provisional) */
                xpose[wv][2 * lane] = u32x4{p2.lo.w[0], p2.lo.w[1], p2.lo.w[2], p2.lo.w[3]};
                xpose[wv][2 * lane + 1] = u32x4{hi.w[0], hi.w[1], hi.w[2], hi.w[3]};
                const u32x4 pa = xpose[wv][lane], pb = xpose[wv][64 + lane];
                const unsigned long long owners = __ballot(outcol);
                u32x4 *dst = reinterpret_cast<u32x4 *>(fordered + (q * H + (strip * kStripCols - 2 + 64 * wv)));
#ifndef BEV_EXP_NOSTORE /* timing experiment: what the ordered cloud's stores cost (results are wrong without them) */
#ifdef BEV_EXP_WBSTORE /* timing experiment: default (write-back) policy for the ordered cloud's whole-line stores */
                if ((owners >> (lane >> 1)) & 1ull) dst[lane] = pa;
                if ((owners >> (32 + (lane >> 1))) & 1ull) dst[64 + lane] = pb;
#else
                if ((owners >> (lane >> 1)) & 1ull) __builtin_nontemporal_store(pa, dst + lane);
                if ((owners >> (32 + (lane >> 1))) & 1ull) __builtin_nontemporal_store(pb, dst + 64 + lane);
#endif
#else
                if (pa.x == 0x12345678u && pb.x == 0x9abcdef0u && owners) __builtin_nontemporal_store(pa, dst + lane); /* keeps the values alive */
#endif
                if (outcol && fgm) fgm[(uint32_t)(q * H + v)] = (int8_t)p2.gflag;
            }
        }

        /* ---- status of row r (BatchMultiBevGen.cpp:142-182) ---- */
        int s_r = kSteep;
        if (r >= lo_row && r < N) { /* workgroup-uniform */
            /* row r-1 of the threads two to the right / left */
            XYZI right{lane_from_f(sh_right, prev.x), lane_from_f(sh_right, prev.y), lane_from_f(sh_right, prev.z), lane_from_f(sh_right, prev.i)};
            XYZI left{lane_from_f(sh_left, prev.x), lane_from_f(sh_left, prev.y), lane_from_f(sh_left, prev.z), lane_from_f(sh_left, prev.i)};
            const float4(*pe)[4] = edge[(r + 2) % 3];
            if (lane >= 62 && wv + 1 < kWaves) { const float4 q = pe[wv + 1][lane - 62]; right = XYZI{q.x, q.y, q.z, q.w}; }
            if (lane < 2 && wv > 0) { const float4 q = pe[wv - 1][lane + 2]; left = XYZI{q.x, q.y, q.z, q.w}; }
            if (outcol) {
                XYZI up = prev;                                  /* (r-1, c)                  :143     */
                if (up.i == -1.0f) up = right;                   /* (r-1, (c+2) % H)          :146-149 */
                if (up.i == -1.0f) up = left;                    /* flat (r-1)*H + c - 2      :151-154 */
                if (up.i == -1.0f && r >= 2) up = prevprev;      /* (r-2, c)                  :157-160 */
                if (cur.i == -1.0f || up.i == -1.0f) s_r = kInvalid; /* :162-167 */
#ifndef BEV_EXP_NOANGLE
                else s_r = angle_is_ground_flat(up.x - cur.x, up.y - cur.y, up.z - cur.z) ? kGround : kSteep; /* :169-182 */
#else /* timing experiment */
                else s_r = (up.x - cur.x) > 1e30f ? kGround : kSteep;
#endif
            }
        }

        /* ---- ground_mat of row r-1 is now decided (closed form, see bev_exact.h) ---- */
        {
            const int q = r - 1;
            int gf = 0;
            if (q >= lo_row) gf = (p1.status == kInvalid) ? -1 : (p1.status == kGround ? 1 : (s_r == kGround ? 1 : 0));
            else if (q == lo_row - 1) gf = (s_r == kGround) ? 1 : 0;
            p1.gflag = (q >= 0 && q < N) ? gf : 0;
        }
        const bool cand1 = outcol && p1.gflag == 1;
        /* Provisional labels.  Phase C un-grounds a candidate that lies 0.30 m above a neighbour cell's average ground
         * height — known only after the whole frame has been summed.  The walk GUESSES: a candidate 0.30 m above the last
         * candidate of its column that it took for ground is written with its own label, every other candidate with
         * label 0; k_ground_resolve tests every candidate exactly and patches the wrong guesses in either direction.
         * The guess only decides how many sparse 2-byte patches are needed (benchmark frames: 1.3 k instead of 7.9 k per
         * frame). */
        {
            const float zq = __uint_as_float(p1.lo.w[2]);
            /* a candidate whose label is not the -2 every producer writes (MulranPointCloudSelect.cpp:126) keeps its
             * label whatever the guess: phase C can then always patch without looking the input point up again (the
             * key says "-2" or the patch is a 0) */
            const bool plain = (p1.hi.w[3] & 0xffffu) == 0xfffeu;
            p1.pred = cand1 && (!plain || zq - zref >= 0.3f); /* (the comparison is false while zref is NaN) */
            if (cand1 && !p1.pred) zref = zq;
        }
        if (cand1) {
            const int cell = ground_cell(__uint_as_float(p1.lo.w[0]), __uint_as_float(p1.lo.w[1]));
            p1.key = candidate_key_edges(cell, tid - 2, p1.pred, p1.code, (int)(int16_t)(p1.hi.w[3] & 0xffffu),
                                         edge_x[cell / kGridCols], edge_y[cell % kGridCols]);
        }

        /* ---- row r's record (the one row r-3 has left) ---- */
        p0.lo = cur_lo;
        p0.hi = cur_hi;
        p0.status = s_r;
        p0.gflag = 0;
        p0.pred = false;
        p0.key = 0u;
#ifndef BEV_EXP_NOCODE
        p0.code = bev_code(cur.x, cur.y, cur.z, (int)(int16_t)(cur_hi.w[3] & 0xffffu), rp);
#else /* timing experiment */
        p0.code = kSkip;
#endif
    };
    /* two extra iterations drain the pipeline */
    /* (the stream source has no stage that rotates with period 2: three copies of the step instead of six) */
    constexpr int kUnroll = kStream ? 3 : 6;
    for (int r0 = 0; r0 < N + 2; r0 += kUnroll) {
        row_step(std::integral_constant<int, 0>{}, r0);
        if (r0 + 1 < N + 2) row_step(std::integral_constant<int, 1>{}, r0 + 1);
        if (r0 + 2 < N + 2) row_step(std::integral_constant<int, 2>{}, r0 + 2);
        if (kUnroll == 6) {
            if (r0 + 3 < N + 2) row_step(std::integral_constant<int, 3>{}, r0 + 3);
            if (r0 + 4 < N + 2) row_step(std::integral_constant<int, 4>{}, r0 + 4);
            if (r0 + 5 < N + 2) row_step(std::integral_constant<int, 5>{}, r0 + 5);
        }
    }
    if (kStream) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); /* no LDS-DMA may outlive the workgroup's LDS */
    lds_barrier();
    if (tid < bands) b.ncode[((size_t)f * g.emitters + strip) * bands + tid] = band_cursor[tid];
    if (kStream) {
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            consumed += __shfl_xor(consumed, d);
            failed |= __shfl_xor(failed, d);
        }
        if (lane == 0) {
            atomicAdd(&b.info[f].consumed, consumed);
            if (failed) atomicOr(&b.info[f].failed, 1u);
        }
    }
}

/* ------------------------------------------------------------------------- */
/* k_walk (round 3): the column walk for the winner-table and identity sources, rebuilt around three measurements:
 *   1. hipcc drains the memory queue (s_waitcnt vmcnt(0)) at the top of EVERY row step of k_strip_ground: gfx9-family
 *      loads and stores retire out of order with respect to each other, so with stores pending the compiler cannot
 *      count, and the "two rows in flight" were one row in flight plus a full round trip per step.  Here every global
 *      READ of the row loop is an LDS-DMA load (global_load_lds: per-lane source address, data lands in LDS, no VGPR
 *      destination the compiler could copy or spill while the load is in flight), issued two steps ahead and waited
 *      for with a COUNTED s_waitcnt: "a load has completed once at most as many operations are outstanding as loads
 *      were issued after it" holds whatever the stores in between do, and the stores of a step are issued BEFORE its
 *      loads, so that the wait at the top of a step covers stores that are a whole step old and loads that are two.
 *   2. a fifth of the walk's vector instructions were v_readlane restores of spilled scalar registers: the raster
 *      constants came back as an 8-dword tuple for every multiplication, pointers that had been laundered through
 *      asm turned every store into a FLAT store (which also counts on lgkmcnt, the LDS counter).  The raster
 *      constants live in vector registers here (they only feed VALU), the power-of-two / divide choice is a template
 *      parameter, stores go through address-space-1 pointers (global_store, scalar base + 32-bit lane offset).
 *   3. waves without a column (the last strip of a row holds 67 of 256 threads for HDL_64E, 16 for OS1_64) end
 *      before the row loop: an ended wave drops out of s_barrier.
 * The arithmetic, the candidate segments, the provisional labels and the code lists are k_strip_ground's (see there). */
template <class T> using gptr = __attribute__((address_space(1))) T *;
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
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" : : "n"(N) : "memory"); }
/* a wave-uniform value that only feeds vector instructions: keep it out of the scalar file */
template <class T>
__device__ __forceinline__ T in_vgpr(T v)
{
    asm volatile("" : "+v"(v));
    return v;
}
/* row record of the walk: flags = (status + 1) | (ground_mat + 1) << 2 | pred << 4 | slot holds a point << 5 */
struct WalkRow {
    u32x4 lo, hi;
    uint32_t code, key, fl;
};
__device__ __forceinline__ int wr_status(uint32_t fl) { return (int)(fl & 3u) - 1; }
__device__ __forceinline__ int wr_gflag(uint32_t fl) { return (int)((fl >> 2) & 3u) - 1; }

template <int kSrc, bool kPow2, bool kGm>
__global__ __launch_bounds__(kStripThreads, 4) void k_walk(BatchPtrs b, Geometry g, int nf, uint32_t want_mode)
{
    constexpr bool kIdentity = kSrc == kSrcIdentity;
    static_assert(kSrc == kSrcGather || kSrc == kSrcIdentity, "the in-place source has its own kernel");
    int f, strip;
    if (!map_block_xcd(blockIdx.x, nf, g.strips, f, strip)) return;
    if (!kIdentity && b.info && b.info[f].mode != want_mode) return; /* another launch of the walk has the frame */
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int N = g.N, H = g.H, lo_row = g.N - g.G, strips = g.strips;
    const size_t frame_off = (size_t)f * g.S;
    const int bands = g.raster_bands;

    const int sh_right = ((lane + 2) & 63) << 2, sh_left = ((lane - 2) & 63) << 2, sh_left1 = ((lane - 1) & 63) << 2;
    auto lane_from = [&](int sel, uint32_t x) -> uint32_t { return (uint32_t)__builtin_amdgcn_ds_bpermute(sel, (int)x); };
    auto lane_from_f = [&](int sel, float x) -> float { return __uint_as_float((uint32_t)__builtin_amdgcn_ds_bpermute(sel, (int)__float_as_uint(x))); };
    const int v = strip * kStripCols + tid - 2;                      /* virtual column */
    const bool provider = (v < H + 2) && (v >= 0 || strip == 0);     /* has a point to load */
    const bool outcol = tid >= 2 && tid < 2 + kStripCols && v < H;   /* owns column v's outputs */
    const int vcol = v >= H ? v - H : v;                             /* wrap; v < 0 keeps the flat rule */

    constexpr int kWaves = kStripThreads / 64;
    __shared__ u32x4 ring[3][2][kStripThreads];            /* the points of rows r, r+1, r+2 (low / high halves), by thread */
    __shared__ uint32_t wring[3][kStripThreads];           /* raw winner words of rows r+2, r+3, r+4 */
    __shared__ float4 edge[3][kWaves][4];                  /* rows r, r-1, (r-2): lanes 0, 1, 62, 63 of every wave */
    __shared__ uint32_t wave_cnt[2][kWaves];               /* per-wave candidate counts of the row being written */
    __shared__ uint32_t band_cursor[kMaxBands];            /* entries already in this strip's code list of each band */
    __shared__ uint8_t band_tab[512];                      /* x bin -> raster band */
    __shared__ uint32_t seen[1 << kSeenBits];              /* direct-mapped memo of codes this strip has already listed */
    __shared__ int edge_x[kGridRows], edge_y[kGridCols];   /* BEV bin of every ground-grid row's / column's lower edge */
    if (tid < kMaxBands) band_cursor[tid] = 0u;
    if (tid < 2 * kWaves) wave_cnt[tid / kWaves][tid % kWaves] = 0u;
    if (tid < 3 * kWaves * 4) (&edge[0][0][0])[tid] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int k = tid; k < (1 << kSeenBits); k += kStripThreads) seen[k] = kSkip;
    for (int x = tid; x < g.rp.mat_size; x += kStripThreads) band_tab[x] = (uint8_t)raster_band_of(x, g.rp);
    if (tid < kGridRows) edge_x[tid] = cell_edge_bin(tid, 75.0f, g.rp);
    else if (tid < kGridRows + kGridCols) edge_y[tid - kGridRows] = cell_edge_bin(tid - kGridRows, 50.0f, g.rp);
    lds_barrier();
    /* a wave none of whose threads has a column ends here (its counts stay zero, nobody reads its edge lanes: the
     * threads that would are not output columns) */
    if (__ballot(provider) == 0ull) return;

    const bev_point_t *fpts = kIdentity ? (b.pts + frame_off) : (b.pts + b.frames[f].in_offset);
    const uint32_t *fwin = b.winner + frame_off;
    const uint32_t win_tag = b.win_tag;
    const int win_shift = b.win_shift;
    /* an empty slot loads a dummy (the first point of this frame's OUTPUT: always allocated, one cached line) and is
     * zeroed when the row is consumed: every step issues the same loads */
    const Half *dummy = reinterpret_cast<const Half *>(b.ordered + frame_off);
    auto has_slot = [&](int r) -> bool { return provider && r < N && r * H + vcol >= 0; };
    /* LDS addresses of this wave's pieces of the rings */
    const uint32_t ring_l = __builtin_amdgcn_readfirstlane(lds_addr(&ring[0][0][0])) + (uint32_t)wv * 1024u;
    const uint32_t wring_l = __builtin_amdgcn_readfirstlane(lds_addr(&wring[0][0])) + (uint32_t)wv * 256u;
    auto issue_winner = [&](int q, int slot) { /* raw winner word of row q */
        if (kIdentity) return;
        const int fl = has_slot(q) ? q * H + vcol : 0;
        glds4_nt(&fwin[fl], wring_l + (uint32_t)slot * 1024u);
    };
    auto issue_points = [&](uint32_t w, int slot) { /* w: input index + 1, 0 = empty slot */
        const Half *src = w != 0u ? reinterpret_cast<const Half *>(fpts + (w - 1u)) : dummy;
        glds16x2(src, ring_l + (uint32_t)slot * 8192u, src + 1, ring_l + (uint32_t)slot * 8192u + 4096u);
    };
    auto winner_of = [&](int q, uint32_t raw) -> uint32_t { /* input index + 1 of slot (q, this column), 0 = empty */
        if (!has_slot(q)) return 0u;
        if (kIdentity) return (uint32_t)(q * H + vcol) + 1u;
        return winner_index(raw, win_tag, win_shift);
    };
    uint32_t full = 0u; /* bit (row mod 3): the row's slot holds a point */
    {   /* prologue: the queue the row loop expects — points of row 0, winners of row 2, points of row 1, winners of row 3 */
        issue_winner(0, 0);
        issue_winner(1, 1);
        wait_vm<0>();
        const uint32_t w0 = winner_of(0, kIdentity ? 0u : wring[0][tid]), w1 = winner_of(1, kIdentity ? 0u : wring[1][tid]);
        full = (w0 != 0u ? 1u : 0u) | (w1 != 0u ? 2u : 0u);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); /* the words have been read before their ring slots are refilled */
        issue_points(w0, 0);
        issue_winner(2, 2);
        issue_points(w1, 1);
        issue_winner(3, 0);
    }

    WalkRow pr[3] = {};
    float zref = __uint_as_float(0x7fc00000u); /* height of the column's last candidate taken for ground (NaN: none yet) */

    const size_t cand_base = (size_t)f * g.segs * kSeg;
    const gptr<u32x2> fcand = (gptr<u32x2>)(b.cand + cand_base);
    const gptr<uint32_t> fncand = (gptr<uint32_t>)(b.ncand + (size_t)f * g.segs);
    const uint32_t code_cap = g.code_cap;
    const gptr<uint32_t> flist = (gptr<uint32_t>)(b.code_main + ((size_t)f * g.emitters + strip) * bands * (size_t)code_cap);
    const gptr<u32x4> fordered = (gptr<u32x4>)(b.ordered + frame_off);
    const gptr<int8_t> fgm = (gptr<int8_t>)(kGm ? b.gm + frame_off : nullptr);
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
    auto bin_of = [&](float p) -> int { /* bev_bin_rp with the reciprocal / divide choice made at compile time */
        const float s = p + rp.max_range_f;
        return round_half_up_bin(kPow2 ? s * rp.inv_interval : s / rp.interval);
    };
    auto code_of = [&](float px, float py, float pz, int label) -> uint32_t { /* bev_code, BatchMultiBevGen.cpp:279-285, :343-349 */
        const int x = bin_of(px), y = bin_of(py);
        const bool in = (label != 0) & ((unsigned)x < (unsigned)rp.mat_size) & ((unsigned)y < (unsigned)rp.mat_size);
        const float hq = kPow2 ? pz * rp.inv_height_res : pz / rp.height_res;
        int layer = cvtt_f32(roundf(hq + rp.lidar_to_ground)); /* :281 */
        int h = height_times4(pz + rp.lidar_to_ground);        /* :345 */
        h = h < 0 ? 0 : (h > 255 ? 255 : h);                   /* :346 */
        const uint32_t l = (layer >= 0 && layer < rp.n_layers) ? (uint32_t)layer : kNoLayer;
        const uint32_t code = (uint32_t)(in ? x : 0) | ((uint32_t)(in ? y : 0) << 9) | ((uint32_t)h << 18) | (l << 26);
        return in ? code : kSkip;
    };
    /* where a thread's point goes in its wave's 2 KiB transposition area (the ring slot the step has just consumed:
     * 1 KiB in the low plane, 1 KiB in the high plane): points 0..31 of the wave in the first, 32..63 in the second */
    const uint32_t xp_w = (uint32_t)(lane & 31) * 32u + (lane < 32 ? 0u : 4096u);

    auto row_step = [&](auto I, const int r) {
        constexpr int s0 = decltype(I)::value % 3;         /* ring slot of row r (and of row r + 3) */
        constexpr int s2 = (decltype(I)::value + 2) % 3;   /* ... of row r + 2: the slot row r - 1 has left */
        constexpr int s1 = (decltype(I)::value + 1) % 3;   /* winner ring: row r + 4 goes where row r + 1's word was */
        WalkRow &p0 = pr[s0], &p1 = pr[s2], &p2 = pr[s1];
        const int par = r & 1;
        /* everything but the three newest loads (points of row r + 1, winners of row r + 3) has arrived: the points of
         * row r and the winner words of row r + 2 (the identity source issues no winner loads: two newest) */
        wait_vm<kIdentity ? 2 : 3>();
        u32x4 cur_lo = ring[s0][0][tid], cur_hi = ring[s0][1][tid];
        const uint32_t wraw = kIdentity ? 0u : wring[s2][tid];
        if (!((full >> s0) & 1u)) { /* untouched slot: value-initialised, BatchMultiBevGen.cpp:98 */
            cur_lo = u32x4{0u, 0u, 0u, 0u};
            cur_hi = u32x4{0u, 0u, 0u, 0u};
        }
        const XYZI prev{__uint_as_float(p1.lo.x), __uint_as_float(p1.lo.y), __uint_as_float(p1.lo.z), __uint_as_float(p1.hi.x)};
        const XYZI prevprev{__uint_as_float(p2.lo.x), __uint_as_float(p2.lo.y), __uint_as_float(p2.lo.z), __uint_as_float(p2.hi.x)};
        const XYZI cur{__uint_as_float(cur_lo.x), __uint_as_float(cur_lo.y), __uint_as_float(cur_lo.z), __uint_as_float(cur_hi.x)};
        if (lane < 2 || lane >= 62) edge[r % 3][wv][lane < 2 ? lane : lane - 60] = make_float4(cur.x, cur.y, cur.z, cur.i);
        /* candidates of row r-2: every wave publishes its count, the write-out after the barrier ranks them */
        const bool cand2 = outcol && wr_gflag(p2.fl) == 1;
        const unsigned long long mc = __ballot(cand2);
        if (lane == 0) wave_cnt[par][wv] = (uint32_t)__popcll(mc);
        lds_barrier();

        /* ---- write out row r-2 (first thing after the barrier: its stores are the oldest entries of the step) ---- */
        if (r >= 2) {
            const int q = r - 2;
            const int rr = q - (lo_row - 1);        /* only rows lo-1 .. N-1 can hold candidates */
            if (rr >= 0) {
                uint32_t before = 0, total = 0;
#pragma unroll
                for (int w = 0; w < kWaves; ++w) {
                    const uint32_t c = wave_cnt[par][w];
                    if (w < wv) before += c;
                    total += c;
                }
                const uint32_t seg = (uint32_t)(rr * strips + strip);
                if (cand2) {
                    const uint32_t rank = before + (uint32_t)__popcll(mc & ((1ull << lane) - 1ull));
                    fcand[seg * (uint32_t)kSeg + rank] = u32x2{p2.key, p2.lo.z}; /* key | height */
                }
                if (tid == 2) fncand[seg] = total;
            }
            {   /* BEV code of a slot that is not a candidate: final, appended to this strip's list of its raster band */
                bool has = outcol && !cand2 && p2.code != kSkip;
                const uint32_t left_code = lane_from(sh_left1, p2.code);
                const bool left_has = lane_from(sh_left1, has ? 1u : 0u) != 0u;
                if (lane > 0 && left_has && left_code == p2.code) has = false;
                if (has) {
                    const uint32_t slot = (p2.code * 0x9E3779B1u) >> (32 - kSeenBits);
                    if (seen[slot] == p2.code) has = false;
                    else seen[slot] = p2.code;
                }
                if (has) {
                    const int band = band_tab[code_x(p2.code)];
                    const uint32_t pos = atomicAdd(&band_cursor[band], 1u);
                    flist[(uint32_t)band * code_cap + pos] = p2.code;
                }
            }
            {   /* the ordered cloud: a wave's 64 points leave as two whole KiB (see k_strip_ground) */
                u32x4 hi = p2.hi;
                const bool as_ground = cand2 && !((p2.fl >> 4) & 1u);
                if (as_ground) hi.w &= 0xffff0000u; /* label = 0, BatchMultiBevGen.cpp:245 (provisional) */
                char *xb = reinterpret_cast<char *>(&ring[s0][0][64 * wv]);
                *reinterpret_cast<u32x4 *>(xb + xp_w) = p2.lo;
                *reinterpret_cast<u32x4 *>(xb + xp_w + 16) = hi;
                const u32x4 pa = ring[s0][0][64 * wv + lane], pb = ring[s0][1][64 * wv + lane];
                const unsigned long long owners = __ballot(outcol);
                const uint32_t at = (uint32_t)(q * H + (strip * kStripCols - 2 + 64 * wv)) * 2u; /* (never dereferenced below 0) */
                if ((owners >> (lane >> 1)) & 1ull) __builtin_nontemporal_store(pa, fordered + (at + (uint32_t)lane));
                if ((owners >> (32 + (lane >> 1))) & 1ull) __builtin_nontemporal_store(pb, fordered + (at + 64u + (uint32_t)lane));
                if (kGm && outcol) fgm[(uint32_t)(q * H + v)] = (int8_t)wr_gflag(p2.fl);
            }
        }
        /* ---- the loads of this step, behind its stores: points of row r + 2, winner words of row r + 4 ---- */
        {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); /* this wave is done reading the slots that are refilled */
            const uint32_t wn = winner_of(r + 2, wraw);
            full = (full & ~(1u << s2)) | (wn != 0u ? 1u << s2 : 0u);
            issue_points(wn, s2);
            issue_winner(r + 4, s1);
        }

        /* ---- status of row r (BatchMultiBevGen.cpp:142-182) ---- */
        int s_r = kSteep;
        if (r >= lo_row && r < N) { /* workgroup-uniform */
            XYZI right{lane_from_f(sh_right, prev.x), lane_from_f(sh_right, prev.y), lane_from_f(sh_right, prev.z), lane_from_f(sh_right, prev.i)};
            XYZI left{lane_from_f(sh_left, prev.x), lane_from_f(sh_left, prev.y), lane_from_f(sh_left, prev.z), lane_from_f(sh_left, prev.i)};
            const float4(*pe)[4] = edge[(r + 2) % 3];
            if (lane >= 62 && wv + 1 < kWaves) { const float4 q = pe[wv + 1][lane - 62]; right = XYZI{q.x, q.y, q.z, q.w}; }
            if (lane < 2 && wv > 0) { const float4 q = pe[wv - 1][lane + 2]; left = XYZI{q.x, q.y, q.z, q.w}; }
            if (outcol) {
                XYZI up = prev;                                  /* (r-1, c)                  :143     */
                if (up.i == -1.0f) up = right;                   /* (r-1, (c+2) % H)          :146-149 */
                if (up.i == -1.0f) up = left;                    /* flat (r-1)*H + c - 2      :151-154 */
                if (up.i == -1.0f && r >= 2) up = prevprev;      /* (r-2, c)                  :157-160 */
                if (cur.i == -1.0f || up.i == -1.0f) s_r = kInvalid; /* :162-167 */
                else s_r = angle_is_ground_flat(up.x - cur.x, up.y - cur.y, up.z - cur.z) ? kGround : kSteep; /* :169-182 */
            }
        }

        /* ---- ground_mat of row r-1 is now decided (closed form, see bev_exact.h) ---- */
        int gf = 0;
        {
            const int q = r - 1, st1 = wr_status(p1.fl);
            if (q >= lo_row) gf = (st1 == kInvalid) ? -1 : (st1 == kGround ? 1 : (s_r == kGround ? 1 : 0));
            else if (q == lo_row - 1) gf = (s_r == kGround) ? 1 : 0;
            if (!(q >= 0 && q < N)) gf = 0;
        }
        const bool cand1 = outcol && gf == 1;
        bool pred1;
        {   /* provisional label: see k_strip_ground */
            const float zq = __uint_as_float(p1.lo.z);
            const bool plain = (p1.hi.w & 0xffffu) == 0xfffeu;
            pred1 = cand1 && (!plain || zq - zref >= 0.3f); /* (the comparison is false while zref is NaN) */
            if (cand1 && !pred1) zref = zq;
        }
        p1.fl = (p1.fl & 3u) | ((uint32_t)(gf + 1) << 2) | (pred1 ? 16u : 0u);
        if (cand1) {
            const int cell = ground_cell(__uint_as_float(p1.lo.x), __uint_as_float(p1.lo.y));
            p1.key = candidate_key_edges(cell, tid - 2, pred1, p1.code, (int)(int16_t)(p1.hi.w & 0xffffu),
                                         edge_x[cell / kGridCols], edge_y[cell % kGridCols]);
        }

        /* ---- row r's record (the one row r-3 has left) ---- */
        p0.lo = cur_lo;
        p0.hi = cur_hi;
        p0.fl = (uint32_t)(s_r + 1) | (1u << 2);
        p0.key = 0u;
        p0.code = code_of(cur.x, cur.y, cur.z, (int)(int16_t)(cur_hi.w & 0xffffu));
    };
    /* two extra iterations drain the pipeline */
    for (int r0 = 0; r0 < N + 2; r0 += 3) {
        row_step(std::integral_constant<int, 0>{}, r0);
        if (r0 + 1 < N + 2) row_step(std::integral_constant<int, 1>{}, r0 + 1);
        if (r0 + 2 < N + 2) row_step(std::integral_constant<int, 2>{}, r0 + 2);
    }
    wait_vm<0>(); /* no LDS-DMA may outlive the workgroup's LDS */
    lds_barrier();
    if (tid < bands) b.ncode[((size_t)f * g.emitters + strip) * bands + tid] = band_cursor[tid];
}

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
/* markGroundPoints phase B + divide, BatchMultiBevGen.cpp:187-210.
 *
 * What must be reproduced: per 2 m cell, sum += z in ROW-MAJOR SLOT ORDER in float32 (and cnt = cnt + 1 from 0.01f).
 * Cells are independent; only the order inside a cell matters.  Candidates arrive in slot order (segments in
 * (row, strip) order, compacted in column order), so a STABLE sort by cell puts every cell's heights in the order the
 * reference adds them; then one lane per cell adds its run sequentially.
 *
 * One SMALL workgroup per frame (4 waves, the size of a column-walk workgroup, so that it is dispatched into whatever
 * slot a workgroup of the other sub-batch's streaming kernels leaves — an 8-wave / 139 KB workgroup waited for the
 * whole column walk to drain) works through the frame part by part; a part = kPartSegs consecutive segments, so parts
 * in order = slot order.  Per part, everything happens in LDS and registers:
 *   hist    every wave counts its kSegsPerWave segments' candidates per cell (LDS atomics, two 16-bit counters per
 *           word); keys and heights stay in registers
 *   scan    per-cell totals over the waves, exclusive scan over the cells -> the part's runs
 *   place   stable placement into the part's height buffer: lanes of a 64-slice that share a cell rank themselves with
 *           ballots (six key bits, a verification, the other six only when it fails): constant work however many
 *           distinct cells a slice has
 *   sum     thread t continues the running (sum, cnt) of cells t, t + 256, ... through their runs of this part
 * while the next part's keys and heights are already in flight, so the only memory round trip that is ever exposed is
 * the first one.  No intermediate of phase B touches HBM (round 1: the sorted heights bounced through global memory). */
constexpr int kCells = kGridCells;
static_assert(kPartSegs * kSeg <= 4096, "a part's run start (12 bits) and length (13 bits) share a word with room to spare");
/* one workgroup per frame, all 3750 cells: 99 KB of LDS.  (Round 2 also had a form with four workgroups per frame, cells
 * by cell mod 4, 37 KB each: bit-identical, 17 % shorter alone, but its 1000 high-priority workgroups per sub-batch pushed
 * the front stage aside — 230 k instead of 260 k frames/s — and the walk paid four ballots per row for it; removed in
 * round 3, DESIGN.md.) */
struct SumDims {
    static constexpr int cells = kCells;
    static constexpr int hist_stride = ((cells + 1) / 2 + 3) / 4 * 4; /* words per wave's histogram: two 16-bit counters per word */
    static constexpr int touch_words = (cells + 31) / 32;
    static constexpr size_t lds_bytes = sizeof(uint32_t) * ((size_t)kSumWaves * hist_stride + cells + (size_t)kPartSegs * kSeg +
                                                            2 * (size_t)cells + touch_words + (cells + 1) / 2 + 16);
};
size_t cell_sums_lds_bytes() { return SumDims::lds_bytes; }

__global__ __launch_bounds__(kSumThreads) void k_cell_sums(BatchPtrs b, Geometry g, int nf)
{
    using D = SumDims;
    constexpr int kCellsQ = D::cells, kHistStride = D::hist_stride, kTouchWords = D::touch_words;
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    uint32_t *hist = lds;                                 /* [kSumWaves][kHistStride]: 16-bit counts, cells 2i | 2i+1 << 16 */
    uint32_t *start = hist + kSumWaves * kHistStride;     /* [kCellsQ]: the part's runs, start | length << 16 */
    float *zbuf = reinterpret_cast<float *>(start + kCellsQ); /* [kPartSegs * kSeg]: the part's heights by cell */
    float *sumv = zbuf + kPartSegs * kSeg;                /* [kCellsQ] running sums */
    float *cntv = sumv + kCellsQ;                          /* [kCellsQ] running counts */
    uint32_t *tbits = reinterpret_cast<uint32_t *>(cntv + kCellsQ); /* [kTouchWords]: cells this part has touched */
    uint16_t *tlist = reinterpret_cast<uint16_t *>(tbits + kTouchWords); /* [kCellsQ]: ... listed, in any order */
    uint32_t *misc = reinterpret_cast<uint32_t *>(tlist) + (kCellsQ + 1) / 2; /* [0..1] list lengths (by part parity), [4..7] wave sums, [8] carry */
    uint16_t *hist16 = reinterpret_cast<uint16_t *>(hist); /* the same counters, cell c of wave w at [w * 2 * kHistStride + c] */

    const int f = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int T = g.segs, P = g.parts;
    const uint2 *ccand = b.cand + (size_t)f * T * kSeg; /* key | height */
    const uint32_t *fn = b.ncand + (size_t)f * T;
    constexpr int kSl = kSeg / 64;
    PH_DECL;
    PH();

    for (int k = tid; k < kSumWaves * kHistStride; k += kSumThreads) hist[k] = 0u;
    for (int k = tid; k < kTouchWords; k += kSumThreads) tbits[k] = 0u;
    if (tid < 16) misc[tid] = 0u;
    for (int c = tid; c < kCellsQ; c += kSumThreads) {
        sumv[c] = 0.0f;   /* :133-134 */
        cntv[c] = 0.01f;  /* :135-136 */
    }

    /* software pipeline: counts two parts ahead, keys + heights one part ahead */
    auto load_counts = [&](int p) -> uint32_t { /* lane j < kSegsPerWave: count of this wave's segment j of part p */
        const int t = p * kPartSegs + wv * kSegsPerWave + lane;
        return (p < P && lane < kSegsPerWave && t < T) ? fn[t] : 0u;
    };
    uint32_t key_n[kSegsPerWave][kSl]; /* next part (raw keys; lanes past the segment's count hold garbage) */
    float z_n[kSegsPerWave][kSl];
    int n_n[kSegsPerWave];
    auto request = [&](int p, uint32_t counts) {
        const int t0 = p * kPartSegs + wv * kSegsPerWave;
#pragma unroll
        for (int j = 0; j < kSegsPerWave; ++j) {
            n_n[j] = __shfl((int)counts, j);
            /* whole 64-slices, loaded or skipped by a WAVE-UNIFORM test, and nothing but the loads inside the test: a
             * per-lane predicated load makes the compiler branch around it and wait for the data inside the branch —
             * one round trip after the other (this loop took 4 us per part that way).  Lanes past the count read stale
             * entries of the segment (allocated memory) and are masked where the values are used. */
            const int n = __builtin_amdgcn_readfirstlane(n_n[j]);
            const size_t at = (size_t)(t0 + j) * kSeg + lane;
#pragma unroll
            for (int k = 0; k < kSl; ++k) {
                key_n[j][k] = 0u;
                z_n[j][k] = 0.f;
                if (64 * k < n) {
                    const uint2 kz = ccand[at + 64 * k];
                    key_n[j][k] = kz.x;
                    z_n[j][k] = __uint_as_float(kz.y);
                }
            }
        }
    };
    uint32_t cnt_next = load_counts(0);
    request(0, cnt_next);
    cnt_next = load_counts(1);
    lds_barrier(); /* LDS state initialised */

    uint32_t *myhist = hist + wv * kHistStride;
    PHA_DECL;
    for (int p = 0; p < P; ++p) {
        PHA(7);
        const int par = p & 1;
        /* part p's data into the "current" registers, part p + 1 requested */
        uint32_t cell[kSegsPerWave][kSl];
        float zz[kSegsPerWave][kSl];
        int nn[kSegsPerWave];
#pragma unroll
        for (int j = 0; j < kSegsPerWave; ++j) {
            nn[j] = n_n[j];
#pragma unroll
            for (int k = 0; k < kSl; ++k) {
                cell[j][k] = lane + 64 * k < nn[j] ? (key_n[j][k] & kKeyCellMask) : 0xfffu; /* 0xfff: no candidate */
                zz[j][k] = z_n[j][k];
            }
        }
        request(p + 1, cnt_next);
        cnt_next = load_counts(p + 2);
        PHA(5);

        /* hist.  Lanes of a 64-slice that hold the same cell find each other with one ballot per key bit (see below; 12
         * bits cover 3750 cells; 0xfff is not a cell): constant work however many distinct cells the slice has.  Every lane keeps
         * its rank inside its group, the group's size and whether it leads the group in the spare bits of its cell
         * register (cell | rank << 12 | size << 18 | leader << 25), so the placement below needs no second look.
         * Only leaders touch the histogram (64 LDS atomics on one address would serialise).  The first leader to touch
         * a cell in this part lists it: everything after this phase works on the listed cells only. */
#pragma unroll
        for (int j = 0; j < kSegsPerWave; ++j) {
#pragma unroll
            for (int k = 0; k < kSl; ++k) {
                if (64 * k >= nn[j]) break; /* wave-uniform */
                const uint32_t c = cell[j][k];
                const bool valid = c != 0xfffu;
                /* the cells of 64 consecutive slots are neighbours on the grid (numbers that differ by 1, 49, 50, 51):
                 * their low six bits tell them apart; the group's first lane shows its whole cell, and only a slice
                 * where somebody disagrees (far-apart cells with equal low bits) takes all twelve ballots */
                unsigned long long peers = __ballot(valid);
#pragma unroll
                for (int bit = 0; bit < 6; ++bit) {
                    const bool one = (c >> bit) & 1u;
                    const unsigned long long bal = __ballot(one);
                    peers &= one ? bal : ~bal;
                }
                const int first = valid ? __ffsll((long long)peers) - 1 : lane;
                const uint32_t cf = (uint32_t)__builtin_amdgcn_ds_bpermute(first << 2, (int)c);
                if (__ballot(valid && cf != c) != 0ull) { /* wave-uniform, rare */
#pragma unroll
                    for (int bit = 6; bit < 12; ++bit) {
                        const bool one = (c >> bit) & 1u;
                        const unsigned long long bal = __ballot(one);
                        peers &= one ? bal : ~bal;
                    }
                }
                const unsigned long long lower = peers & ((1ull << lane) - 1ull);
                const uint32_t size = (uint32_t)__popcll(peers), rank = (uint32_t)__popcll(lower);
                const bool leader = valid && lower == 0ull;
                if (leader) {
                    atomicAdd(&myhist[c >> 1], size << (16 * (c & 1u)));
                    const uint32_t bit = 1u << (c & 31u);
                    if (!(atomicOr(&tbits[c >> 5], bit) & bit)) tlist[atomicAdd(&misc[par], 1u)] = (uint16_t)c;
                }
                if (valid) cell[j][k] = c | (rank << 12) | (size << 18) | (leader ? 1u << 25 : 0u);
            }
        }
        PHA(6);
        lds_barrier();
        PHA(0);

        /* listed cells: totals over the waves (hist16[w][c] becomes wave w's offset inside cell c's run) and an
         * exclusive scan over the list -> every listed cell's run in zbuf (any order of the cells will do) */
        const int nT = (int)misc[par];
        if (tid == 0) misc[par ^ 1] = 0u; /* the other parity's length, for the next part (nobody reads it now) */
        for (int i0 = 0; i0 < nT; i0 += kSumThreads) {
            const int i = i0 + tid;
            uint32_t c = 0u, tot = 0u;
            if (i < nT) {
                c = tlist[i];
#pragma unroll
                for (int w = 0; w < kSumWaves; ++w) {
                    const uint32_t v = hist16[w * 2 * kHistStride + c];
                    hist16[w * 2 * kHistStride + c] = (uint16_t)tot;
                    tot += v;
                }
            }
            uint32_t incl = tot;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const uint32_t v = __shfl_up(incl, d);
                if (lane >= d) incl += v;
            }
            if (lane == 63) misc[4 + wv] = incl;
            lds_barrier();
            uint32_t run = misc[8] + incl - tot;
            for (int w = 0; w < wv; ++w) run += misc[4 + w];
            if (i < nT) start[c] = run | (tot << 16);
            lds_barrier(); /* wave sums and the carry have been read */
            if (tid == kSumThreads - 1) misc[8] = run + tot; /* carry into the next 256 listed cells */
        }
        lds_barrier();
        if (tid == 0) misc[8] = 0u;
        PHA(2);

        /* stable placement: slices in slot order (segment by segment, 64 candidates at a time); position = the cell's
         * run start + this wave's cursor inside the run + the lane's rank in its group; the group's leader then
         * advances the cursor (the reads are issued before that update: same wave, program order; two cells of one
         * word may both advance: atomic) */
#pragma unroll
        for (int j = 0; j < kSegsPerWave; ++j) {
#pragma unroll
            for (int k = 0; k < kSl; ++k) {
                if (64 * k >= nn[j]) break; /* wave-uniform */
                const uint32_t v = cell[j][k];
                const uint32_t c = v & 0xfffu;
                if (c != 0xfffu) {
                    const uint32_t off = (myhist[c >> 1] >> (16 * (c & 1u))) & 0xffffu;
                    zbuf[(start[c] & 0xffffu) + off + ((v >> 12) & 63u)] = zz[j][k];
                    if (v & (1u << 25)) atomicAdd(&myhist[c >> 1], ((v >> 18) & 127u) << (16 * (c & 1u)));
                }
            }
        }
        lds_barrier();
        PHA(3);

        /* in-order sums of the listed cells, one thread per cell; the part's traces are wiped on the way */
        for (int k = tid; k < kTouchWords; k += kSumThreads) tbits[k] = 0u;
        for (int i = tid; i < nT; i += kSumThreads) {
            const uint32_t c = tlist[i];
            const uint32_t se = start[c];
#pragma unroll
            for (int w = 0; w < kSumWaves; ++w) hist16[w * 2 * kHistStride + c] = 0;
            int q = (int)(se & 0xffffu);
            const int e = q + (int)(se >> 16);
            float sj = sumv[c], cj = cntv[c];
            /* the adds of one cell are a serial chain (that IS the reference's order); what can be hidden is the LDS
             * latency: the next 8 heights are requested before the current 8 are added */
            if (q + 8 <= e) {
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = zbuf[q + u];
#pragma unroll 1
                for (; q + 16 <= e; q += 8) {
                    float nx[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) nx[u] = zbuf[q + 8 + u];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        sj += v[u];       /* :198-199 */
                        cj = cj + 1.0f;   /* :205-206 */
                    }
#pragma unroll
                    for (int u = 0; u < 8; ++u) v[u] = nx[u];
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    sj += v[u];
                    cj = cj + 1.0f;
                }
                q += 8;
            }
#pragma unroll 1
            for (; q < e; ++q) {
                sj += zbuf[q];
                cj = cj + 1.0f;
            }
            sumv[c] = sj;
            cntv[c] = cj;
        }
        lds_barrier(); /* the next part overwrites start and zbuf; hist and tbits are clean */
        PHA(4);
    }
    PHA_PRINT("cell_sums barrier0 - scan place sum request histloop looptop", tid == 0 && blockIdx.x == 100);
    PH();
    float *avg = b.avg + (size_t)f * kCells;
    for (int c = tid; c < kCellsQ; c += kSumThreads) avg[c] = sumv[c] / cntv[c]; /* :210 */
    PH_PRINT("cell_sums all-parts", tid == 0 && blockIdx.x == 100);
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

/* ------------------------------------------------------------------------- */
/* markGroundPoints phase C for the candidates, BatchMultiBevGen.cpp:216-250.  A candidate that is higher than a
 * neighbour cell's average + 0.30 stops being ground ("hit"): it keeps / gets back its own label, and its BEV code —
 * rebuilt from key and height, bev_exact.h — is appended to a code list of the raster band it falls into, exactly like
 * the walk's lists (an LDS cursor per band, no global atomics): k_bev_raster reads both kinds the same way.  The walk
 * wrote each candidate's label for its guess (key bit kKeyPredBit); only wrong guesses are patched.
 * kResolveParts workgroups per frame, each takes a contiguous quarter of the segments; a wave requests kResolveBatch
 * segments (x 4 slices of 64 candidates) at a time. */
constexpr int kResolveBatch = 4;
template <bool kIdentity>
__global__ __launch_bounds__(kResolveThreads) void k_ground_resolve(BatchPtrs b, Geometry g)
{
    __shared__ float avg[kCells];                        /* the frame's 75 x 50 averages: 4 look-ups per candidate */
    __shared__ int edge_x[kGridRows], edge_y[kGridCols]; /* BEV bin of every ground-grid row's / column's lower edge */
    __shared__ uint32_t band_cursor[kMaxBands];
    __shared__ uint8_t band_tab[512];                    /* x bin -> raster band */
    __shared__ uint16_t cnt[kMaxSegs / kResolveParts + 8];
    const int f = blockIdx.x / kResolveParts, part = blockIdx.x - f * kResolveParts;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int T = g.segs;
    const int t0 = (int)((long long)T * part / kResolveParts), t1 = (int)((long long)T * (part + 1) / kResolveParts);
    for (int i = tid; i < t1 - t0; i += kResolveThreads) {
        cnt[i] = (uint16_t)b.ncand[(size_t)f * T + t0 + i];
    }
    for (int c = tid; c < kCells; c += kResolveThreads) avg[c] = b.avg[(size_t)f * kCells + c];
    if (tid < kGridRows) edge_x[tid] = cell_edge_bin(tid, 75.0f, g.rp);
    else if (tid < kGridRows + kGridCols) edge_y[tid - kGridRows] = cell_edge_bin(tid - kGridRows, 50.0f, g.rp);
    if (tid < kMaxBands) band_cursor[tid] = 0u;
    for (int x = tid; x < g.rp.mat_size; x += kResolveThreads) band_tab[x] = (uint8_t)raster_band_of(x, g.rp);
    lds_barrier();

    constexpr int kSl = kSeg / 64;
    constexpr int kWaves = kResolveThreads / 64;
    const int bands = g.raster_bands, lo_row = g.N - g.G;
    const uint2 *fcand = b.cand + (size_t)f * T * kSeg; /* key | height */
    uint32_t *flist = b.code_main + ((size_t)f * g.emitters + g.strips + part) * bands * (size_t)g.code_cap;
    for (int s0 = t0 + wv; s0 < t1; s0 += kWaves * kResolveBatch) {
        uint32_t key[kResolveBatch][kSl];
        float z[kResolveBatch][kSl];
#pragma unroll
        for (int j = 0; j < kResolveBatch; ++j) {
            const int sg = s0 + j * kWaves;
            const int n = sg < t1 ? (int)cnt[sg - t0] : 0; /* wave-uniform */
#pragma unroll
            for (int k = 0; k < kSl; ++k) { /* whole slices, nothing but the loads inside the uniform test (see k_cell_sums) */
                const size_t at = (size_t)(sg < t1 ? sg : t0) * kSeg + lane + 64 * k;
                key[j][k] = 0u;
                z[j][k] = 0.f;
                if (64 * k < __builtin_amdgcn_readfirstlane(n)) {
                    const uint2 kz = fcand[at];
                    key[j][k] = kz.x;
                    z[j][k] = __uint_as_float(kz.y);
                }
            }
        }
#pragma unroll
        for (int j = 0; j < kResolveBatch; ++j) {
            const int sg = s0 + j * kWaves;
            const int n = sg < t1 ? (int)cnt[sg - t0] : 0;
            const int rr = sg / g.strips, strip = sg - rr * g.strips;
            const size_t slot0 = (size_t)f * g.S + (size_t)(rr + lo_row - 1) * g.H + (size_t)strip * kStripCols;
#pragma unroll
            for (int k = 0; k < kSl; ++k) {
                if (64 * k >= n) break; /* wave-uniform */
                const uint32_t kk = key[j][k];
                const bool have = lane + 64 * k < n;
                const int cell = (int)(kk & kKeyCellMask);
                const bool hit = have && above_neighbour_ground(z[j][k], cell, avg);
                const bool pred = (kk & kKeyPredBit) != 0u;
                const bool wrong = have && hit != pred;
                if (!__ballot(hit || wrong)) continue; /* wave-uniform */
                const size_t idx = slot0 + ((kk >> kKeyColShift) & 0xffu);
                if (hit && !(kk & kKeyNoCodeBit)) {
                    uint32_t code;
                    if (!candidate_key_escapes(kk)) {
                        code = bev_code_from_bins(edge_x[cell / kGridCols] + (int)((kk >> kKeyDxShift) & 3u),
                                                  edge_y[cell % kGridCols] + (int)((kk >> kKeyDyShift) & 3u), z[j][k], g.rp);
                    } else { /* cell clamped or bins not next to the cell's edge: x, y from the point itself */
                        const float4 a = *reinterpret_cast<const float4 *>(b.ordered + idx);
                        code = bev_code(a.x, a.y, a.z, 1 /* not 0: no kKeyNoCodeBit */, g.rp);
                    }
                    if (code != kSkip) {
                        const int band = band_tab[code_x(code)];
                        flist[(size_t)band * g.code_cap + atomicAdd(&band_cursor[band], 1u)] = code;
                    }
                }
                if (wrong) { /* the walk's provisional label differs */
                    /* not un-grounded: label = 0, BatchMultiBevGen.cpp:245; un-grounded: the point's own label back —
                     * which is -2: the walk guesses "stays ground" only for points that carry it */
                    const uint16_t label = hit ? (uint16_t)(int16_t)-2 : (uint16_t)0;
                    reinterpret_cast<uint16_t *>(b.ordered + idx)[14] = label; /* label @28 */
                }
            }
        }
    }
    lds_barrier();
    if (tid < bands) b.ncode[((size_t)f * g.emitters + g.strips + part) * bands + tid] = band_cursor[tid];
}

/* ------------------------------------------------------------------------- */
/* Both rasters (BatchMultiBevGen.cpp:271-292 occupancy, 24 layers; :340-356 uint8 max height), one workgroup per
 * (frame, x-band of the images).  The band's 24-bit layer masks and max heights live in LDS (two planes of rows x M
 * words); its input are this band's code lists: one per strip from the walk (slots that are not candidates) and one
 * per part from k_ground_resolve (un-grounded candidates).  Finished planes leave with 16-byte stores, 1 KiB
 * contiguous per wave-instruction. */
int raster_bands_for(int M) /* uniform bands whose two LDS planes fit; the coarse band height is M / this */
{
    for (int bands = kRasterSplit; bands <= 16; bands *= 2)
        if (M % bands == 0 && (size_t)2 * (M / bands) * M * sizeof(uint32_t) <= (size_t)100 * 1024) return bands;
    return 0;
}
size_t raster_lds_bytes(const Geometry &g)
{
    return (size_t)2 * g.rp.coarse * g.rp.mat_size * sizeof(uint32_t);
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
            for (int l = 0; l < L; ++l) {
                uint32_t w[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    /* byte = 255 where bit l of the mask is set */
                    w[q] = (((mk[4 * q] >> l) & 1u) * 0xffu) | (((mk[4 * q + 1] >> l) & 1u) * 0xff00u) |
                           (((mk[4 * q + 2] >> l) & 1u) * 0xff0000u) | (((mk[4 * q + 3] >> l) & 1u) * 0xff000000u);
                }
                store_stream(reinterpret_cast<uint4 *>(mout + (size_t)l * plane), make_uint4(w[0], w[1], w[2], w[3]));
            }
        }
    }
}

__global__ __launch_bounds__(kRasterThreads) void k_bev_raster(BatchPtrs b, Geometry g, int nf, int want_multi, int want_single)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    __shared__ uint32_t list_end[kMaxStrips + kResolveParts + 1]; /* inclusive prefix of this band's code-list lengths */
    const int M = g.rp.mat_size, L = g.rp.n_layers, bands = g.raster_bands, E = g.emitters;
    /* the bands of a frame on ONE XCD (blocks b and b+8 share an L2), adjacent launch slots */
    const int xl = blockIdx.x & 7, jj = blockIdx.x >> 3;
    const int f = (jj / bands) * 8 + xl, band = jj % bands;
    if (f >= nf) return;
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
    if (tid == 0) list_end[0] = 0u;
    lds_barrier();
    if (tid == 0) /* few lists (13 for HDL_64E): a serial prefix */
        for (int e = 0; e < E; ++e) list_end[e + 1] += list_end[e];
    lds_barrier();
    PH();

    /* this band's lists as ONE index space, so that every load of the workgroup is requested at once */
    {
        constexpr int kU = 8;
        const uint32_t total = list_end[E];
        const uint32_t *fmain = b.code_main + (size_t)f * E * bands * g.code_cap;
        uint32_t ends[16]; /* ends[j] = first index of list j (j >= 1) */
#pragma unroll
        for (int j = 0; j < 16; ++j) ends[j] = __builtin_amdgcn_readfirstlane(j <= E ? list_end[j] : 0u);
        for (uint32_t i0 = 0; i0 < total; i0 += kU * kRasterThreads) {
            uint32_t c[kU];
#pragma unroll
            for (int k = 0; k < kU; ++k) {
                const uint32_t i = i0 + (uint32_t)k * kRasterThreads + tid;
                c[k] = kSkip;
                if (i < total) {
                    int e = 0;
                    uint32_t e0 = 0u;
                    if (E <= 16) { /* the list ends are wave-uniform: scalar compares, no dependent LDS reads */
#pragma unroll
                        for (int j = 1; j < 16; ++j) {
                            const bool past = j < E && i >= ends[j];
                            e += past ? 1 : 0;
                            e0 = past ? ends[j] : e0;
                        }
                    } else {
                        int lo = 0, hi = E - 1; /* first e with list_end[e + 1] > i */
                        while (lo < hi) {
                            const int mid = (lo + hi) >> 1;
                            if (list_end[mid + 1] > i) hi = mid; else lo = mid + 1;
                        }
                        e = lo;
                        e0 = list_end[e];
                    }
                    c[k] = fmain[((size_t)e * bands + band) * g.code_cap + (i - e0)];
                }
            }
#pragma unroll
            for (int k = 0; k < kU; ++k)
                if (c[k] != kSkip) splat_code(c[k], x0, M, mask, hmax);
        }
    }
    lds_barrier();
    PH();
    store_planes(mask, hmax, want_multi ? b.multi : nullptr, want_single ? b.single : nullptr, f, x0, band_rows, M, L, tid,
                 kRasterThreads);
    PH();
    PH_PRINT(band == 7 ? "raster7 setup codes stores" : "raster1 setup codes stores", tid == 0 && f == 100 && (band == 7 || band == 1));
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

/* ------------------------------------------------------------------------- */
/* launchers                                                                  */
hipError_t configure_kernels(const Geometry &g)
{
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_cell_sums),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)SumDims::lds_bytes);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_bev_raster), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)raster_lds_bytes(g));
    if (e != hipSuccess) return e;
    return hipFuncSetAttribute(reinterpret_cast<const void *>(k_bev_raster_dense),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)raster_lds_bytes(g));
}
void launch_order_scan(const Geometry &g, const BatchPtrs &b, int nf, uint32_t max_pts, int pass, hipStream_t st)
{
    if (max_pts == 0 || nf == 0) return;
    const unsigned per_block = 256u * kScanPerThread;
    dim3 grid((max_pts + per_block - 1u) / per_block, (unsigned)nf);
    hipLaunchKernelGGL(k_order_scan, grid, dim3(256), 0, st, b.pts, b.frames, b.info, pass, b.winner, g.N, g.H, g.S,
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
    else if (source == kSrcStream)
        hipLaunchKernelGGL(k_strip_ground<kSrcStream>, dim3(grid), dim3(kStripThreads), 0, st, b, g, nf, mode);
    else launch_walk<kSrcGather>(g, b, nf, mode, grid, st);
}
void launch_probe(const Geometry &g, const BatchPtrs &b, int nf, bool allow_stream, hipStream_t st)
{
    if (nf == 0) return;
    hipLaunchKernelGGL(k_probe, dim3(nf), dim3(kProbeThreads), 0, st, b, g, allow_stream ? 1 : 0);
}
void launch_verdict(const BatchPtrs &b, int nf, hipStream_t st)
{
    if (nf == 0) return;
    hipLaunchKernelGGL(k_verdict, dim3((nf + 255) / 256), dim3(256), 0, st, b.info, nf);
}
void launch_gather_only(const Geometry &g, const BatchPtrs &b, int nf, hipStream_t st)
{
    if (nf == 0) return;
    hipLaunchKernelGGL(k_gather_only, dim3(xcd_grid(nf, g.tiles)), dim3(kGatherThreads), 0, st, b, g, nf);
}
void launch_cell_sums(const Geometry &g, const BatchPtrs &b, int nf, hipStream_t st)
{
    if (nf == 0) return;
    hipLaunchKernelGGL(k_cell_sums, dim3(nf), dim3(kSumThreads), SumDims::lds_bytes, st, b, g, nf);
}
void launch_ground_resolve(const Geometry &g, const BatchPtrs &b, int nf, bool identity, hipStream_t st)
{
    if (nf == 0) return;
    if (identity)
        hipLaunchKernelGGL(k_ground_resolve<true>, dim3(nf * kResolveParts), dim3(kResolveThreads), 0, st, b, g);
    else
        hipLaunchKernelGGL(k_ground_resolve<false>, dim3(nf * kResolveParts), dim3(kResolveThreads), 0, st, b, g);
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
