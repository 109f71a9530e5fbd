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
__global__ __launch_bounds__(256) void k_probe(BatchPtrs b, Geometry g, int allow_stream)
{
    __shared__ uint32_t samp[kMaxSamples]; /* slot of sample k (position k * kProbeStride) */
    __shared__ uint32_t first_bad;
    const int f = blockIdx.x, tid = threadIdx.x;
    const FrameDesc fd = b.frames[f];
    const uint32_t n = fd.n_pts;
    const bev_point_t *fp = b.pts + fd.in_offset;
    const uint32_t ns = n ? (n - 1u) / kProbeStride + 1u : 0u;
    const bool can = allow_stream && n >= (uint32_t)kStreamMinPrefix && ns <= (uint32_t)kMaxSamples && g.N <= kStreamMaxRows;
    if (tid == 0) first_bad = can ? ns : 0u;
    __syncthreads();
    if (can) {
        for (uint32_t k = tid; k < ns; k += 256u) {
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
        for (uint32_t k = tid; k < ns; k += 256u) /* first sample that is out of range or not above its predecessor */
            if (samp[k] == 0xffffffffu || (k > 0u && samp[k] <= samp[k - 1u])) atomicMin(&first_bad, k);
        __syncthreads();
    }
    const uint32_t m = first_bad;                                      /* samples 0 .. m-1 ascend */
    const uint32_t T = m ? (m - 1u) * kProbeStride + 1u : 0u;          /* the last of them is position T - 1 */
    const bool stream = can && T >= (uint32_t)kStreamMinPrefix && 2u * T >= n;
    if (tid == 0) b.info[f] = FrameInfo{stream ? T : 0u, stream ? kFrameStream : kFrameGeneral, 0u, 0u};
    if (!stream) return;
    uint32_t *fest = b.est + (size_t)f * g.N * g.strips;
    for (int i = tid; i < g.N * g.strips; i += 256) {
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
            } else {
                est = T; /* beyond the last sample: nothing of the prefix lies there */
            }
        }
        fest[i] = est < T ? est : T;
    }
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
constexpr int kSeenBits = 11, kSeenCodes = 1 << kSeenBits; /* the walk's memo of listed BEV codes: 8 KB of LDS */
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
            if (fi.mode == kFrameStream) first = fi.T; /* the prefix is read in place by the stream walk */
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
 * piece left as partial lines: the walk wrote 6.5 MB per frame where 5.8 MB were needed).  The 32-byte ordered-cloud
 * stores (whole 2 KiB wave rows) keep `nt`. */
template <class T>
__device__ __forceinline__ void store_ws(T *p, T v) { *p = v; }

enum : int { kSrcGather = 0, kSrcIdentity = 1, kSrcStream = 2 };
constexpr int kWinLen = kStripThreads + 2 * kStreamSlack; /* positions of a stream window */
constexpr int kWinSlots = 4;  /* rows of windows in LDS: one being read, one being indexed, two landing */
constexpr int kWrapLen = 32;  /* positions of the wrap-around halo's mini window (last strip) */
template <int kSrc>
__global__ __launch_bounds__(kStripThreads, 3) void k_strip_ground(BatchPtrs b, Geometry g, int nf, uint32_t want_mode)
{
    constexpr bool kIdentity = kSrc == kSrcIdentity, kStream = kSrc == kSrcStream;
    /* the strips of a frame share halo columns and the lines at their seams: one XCD (one L2) per frame */
    int f, strip;
    if (!map_block_xcd(blockIdx.x, nf, g.strips, f, strip)) return;
    if (!kIdentity && b.info && b.info[f].mode != want_mode) return; /* another launch of this kernel has the frame */
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int N = g.N, H = g.H, lo_row = g.N - g.G;
    const size_t frame_off = (size_t)f * g.S;
    const int bands = g.raster_bands;

    const int v = strip * kStripCols + tid - 2;                      /* virtual column */
    const bool provider = (v < H + 2) && (v >= 0 || strip == 0);     /* has a point to load */
    const bool outcol = tid >= 2 && tid < 2 + kStripCols && v < H;   /* owns column v's outputs */
    const int vcol = v >= H ? v - H : v;                             /* wrap; v < 0 keeps the flat rule */

    const bev_point_t *fpts = kIdentity ? (b.pts + frame_off) : (b.pts + b.frames[f].in_offset);
    const uint32_t *fwin = b.winner + frame_off;

    /* Neighbour exchange: lanes l+-2 of the same wave are reached with shuffles; only the two edge
     * lanes on each side of a wave go through LDS (768 B instead of a 12 KiB row buffer, so that these
     * workgroups can share a CU with the back end's workgroups of the other lane). */
    constexpr int kWaves = kStripThreads / 64;
    __shared__ float4 edge[3][kWaves][4];                  /* rows r, r-1, (r-2): lanes 0, 1, 62, 63 of every wave */
    __shared__ uint32_t wave_cnt[2][kWaves];               /* per-wave candidate counts of the row being written */
    __shared__ uint32_t band_cursor[kMaxBands];            /* entries already in this strip's code list of each band */
    __shared__ uint8_t band_tab[512];                      /* x bin -> raster band */
    __shared__ uint32_t seen[kSeenCodes];                  /* direct-mapped memo of codes this strip has already listed */
    __shared__ int edge_x[kGridRows], edge_y[kGridCols];   /* BEV bin of every ground-grid row's / column's lower edge */
    /* stream source: the row's points by column offset (two rows), the row each entry belongs to, and the slot of every
     * window position for the order check */
    __shared__ u32x4 win[kStream ? kWinSlots : 1][2][kStream ? kWinLen : 1];   /* windows as they lie in the input: low / high halves */
    __shared__ u32x4 wrapw[kStream ? kWinSlots : 1][2][kStream ? kWrapLen : 1]; /* the row's first positions (wrap-around halo) */
    __shared__ uint32_t colidx[kStream ? 2 : 1][kStream ? kStripThreads : 1];  /* column offset -> window position | (row + 1) << 16 */
    __shared__ int est_l[kStream ? 2 : 1][kStream ? kStreamMaxRows : 1];       /* k_probe's estimates for this strip / for strip 0, every row */
    if (tid < kMaxBands) band_cursor[tid] = 0u;
    for (int k = tid; k < kSeenCodes; k += kStripThreads) seen[k] = kSkip;
    for (int x = tid; x < g.rp.mat_size; x += kStripThreads) band_tab[x] = (uint8_t)raster_band_of(x, g.rp);
    if (tid < kGridRows) edge_x[tid] = cell_edge_bin(tid, 75.0f, g.rp);
    else if (tid < kGridRows + kGridCols) edge_y[tid - kGridRows] = cell_edge_bin(tid - kGridRows, 50.0f, g.rp);
    if (kStream) {
        colidx[0][tid] = 0u;
        colidx[1][tid] = 0u;
        /* the estimates into LDS once: a global load inside the row loop would be one the compiler sees, and its use would
         * bring back the vmcnt(0) that the counted waits below are there to avoid */
        const uint32_t *fe = b.est + (size_t)f * N * g.strips;
        for (int r = tid; r < N; r += kStripThreads) {
            est_l[0][r] = (int)fe[r * g.strips + strip];
            est_l[1][r] = (int)fe[r * g.strips];
        }
    }

    /* Winner words are loaded UNCONDITIONALLY from a clamped address and decoded only when they are used, two rows
     * later: a predicated load whose result is decoded on the spot makes the compiler branch around the load and wait
     * for it — with vmcnt(0), i.e. for every point load in flight as well — inside the branch, once per row (that was
     * the shape of this loop in round 1: the software pipeline below existed on paper only). */
    auto has_slot = [&](int r) -> bool { return provider && r < N && r * H + vcol >= 0; };
    auto load_winner_raw = [&](int r, uint32_t &raw) { /* stream source: asm load, valid only behind a wait_loads */
        raw = 0u;
        if (kIdentity) return;
        const int fl = has_slot(r) ? r * H + vcol : 0;
        if (kStream) ld32_nt(raw, &fwin[fl]);
        else raw = load_once(&fwin[fl]);
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
        const Half *src = w != 0u ? reinterpret_cast<const Half *>(fpts + (w - 1u)) : dummy;
        if (kStream) {
            ld128(lo, src);
            ld128_16(hi, src);
        } else {
            lo = *reinterpret_cast<const u32x4 *>(src);
            hi = *reinterpret_cast<const u32x4 *>(src + 1);
        }
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
        if (kStream) {
            wait_loads<0>(plo[0], phi[0], r0);
            wait_loads<0>(plo[1], phi[1], r1);
        }
        const uint32_t w0 = winner_of(0, r0), w1 = winner_of(1, r1);
        pfull[0] = w0 != 0u;
        pfull[1] = w1 != 0u;
        pfull[2] = false;
        load_point(w0, plo[0], phi[0]);
        load_point(w1, plo[1], phi[1]);
    }

    /* ---- stream source (k_probe took the first T input points for sorted) ----
     * Row r's points of this strip's 256 virtual columns are consecutive in the input; they start near est[r][strip]
     * (interpolated from sampled points).  Three rows ahead, every thread requests ONE window position (est - slack +
     * tid; the first wave also the window's last 32 positions and, for the last strip's two wrap-around halo columns,
     * 32 positions at the row's start) by LDS-DMA: coalesced, in place, no register.  Two steps later the window has
     * landed (counted wait: everything requested two steps ago, see wait_loads); the thread then looks at the (row, col)
     * of ITS position in LDS and enters the position in the row's column index; one step later the owner of each column
     * follows the index and reads its point from the window — unless the winner table, which holds only the tail [T, n)
     * now, says a later point has overwritten the slot (that path keeps the gather pipeline, through asm loads so that
     * its waits are counted too).
     * Verification (results must not depend on the guess): a thread whose position holds a point of the strip's OWN
     * columns counts it and checks that the position before it lies in the prefix and has a smaller slot.  When all
     * T prefix points of a frame have been counted exactly once and no check has failed, the prefix is strictly
     * ascending, every point was where its owner looked, and the result is what getOrderedCloud's scatter gives;
     * otherwise k_verdict sends the frame through the general kernels again. */
    const uint32_t T = kStream ? b.info[f].T : 0u;
    const bool last_strip = strip == g.strips - 1;
    uint32_t consumed = 0u, failed = 0u;
    const int first_col = strip * kStripCols - 2; /* virtual column of offset 0 */
    auto window_pos = [&](int est, int j) -> int { return est - kStreamSlack + j; }; /* input position of window index j */
    auto stream_issue = [&](int r) { /* LDS-DMA requests of row r's window (2 per wave, 6 for the first wave) */
        if (!kStream) return;
        const int rc = r < N ? r : N - 1;
        const int est = est_l[0][rc], est0 = est_l[1][rc];
        const int slot = r & (kWinSlots - 1);
        {
            const int q = window_pos(est, tid);
            const Half *src = reinterpret_cast<const Half *>(fpts + (q >= 0 && q < (int)T ? q : 0));
            glds16(src, __builtin_amdgcn_readfirstlane(lds_addr(&win[slot][0][wv * 64])));
            glds16(src + 1, __builtin_amdgcn_readfirstlane(lds_addr(&win[slot][1][wv * 64])));
        }
        if (wv == 0) { /* wave-uniform; lanes 32 .. 63 idle */
            if (lane < 2 * kStreamSlack) {
                const int q = window_pos(est, kStripThreads + lane);
                const Half *src = reinterpret_cast<const Half *>(fpts + (q >= 0 && q < (int)T ? q : 0));
                glds16(src, __builtin_amdgcn_readfirstlane(lds_addr(&win[slot][0][kStripThreads])));
                glds16(src + 1, __builtin_amdgcn_readfirstlane(lds_addr(&win[slot][1][kStripThreads])));
            }
            if (lane < kWrapLen) {
                const int q = est0 - (kStreamSlack - 2) + lane;
                const Half *src = reinterpret_cast<const Half *>(fpts + (q >= 0 && q < (int)T ? q : 0));
                glds16(src, __builtin_amdgcn_readfirstlane(lds_addr(&wrapw[slot][0][0])));
                glds16(src + 1, __builtin_amdgcn_readfirstlane(lds_addr(&wrapw[slot][1][0])));
            }
        }
    };
    auto stream_index = [&](int r) { /* row r's window has landed: enter every position in the row's column index */
        if (!kStream || r >= N) return;
        const int est = est_l[0][r], est0 = est_l[1][r];
        const int slot = r & (kWinSlots - 1), pr = r & 1;
        const int row0 = r * H, first = row0 + first_col;
        const uint32_t tag = (uint32_t)(r + 1) << 16;
        auto one = [&](int j) { /* window index j */
            const int q = window_pos(est, j);
            if (q < 0 || q >= (int)T) return;
            const uint32_t rcw = win[slot][1][j].y; /* row | col << 16 */
            const uint32_t row = rcw & 0xffffu, col = rcw >> 16;
            if (row >= (uint32_t)N || col >= (uint32_t)H) return;
            const int flat = (int)row * H + (int)col, off = flat - first;
            if (off < 0 || off >= kStripThreads) return;
            colidx[pr][off] = tag | (uint32_t)j;
            if (off >= 2 && off < 2 + kStripCols && first_col + off < H) { /* a point of this strip's own columns */
                ++consumed;
                if (q > 0) { /* its predecessor in the input must lie in the window and have a smaller slot */
                    if (j == 0) { failed = 1u; return; }
                    const uint32_t pw = win[slot][1][j - 1].y;
                    const uint32_t prow = pw & 0xffffu, pcol = pw >> 16;
                    if (prow >= (uint32_t)N || pcol >= (uint32_t)H || (int)prow * H + (int)pcol >= flat) failed = 1u;
                }
            }
        };
        one(tid);
        if (wv == 0 && lane < 2 * kStreamSlack) one(kStripThreads + lane);
        if (wv == 0 && last_strip && lane < kWrapLen) { /* slots r*H and r*H + 1 as the wrap-around halo columns H, H + 1 */
            const int q = est0 - (kStreamSlack - 2) + lane;
            if (q >= 0 && q < (int)T) {
                const uint32_t rcw = wrapw[slot][1][lane].y;
                const uint32_t row = rcw & 0xffffu, col = rcw >> 16;
                if (row == (uint32_t)r && col < 2u) {
                    const int off = H + (int)col - first_col;
                    if (off >= 0 && off < kStripThreads) colidx[pr][off] = tag | 0x8000u | (uint32_t)lane;
                }
            }
        }
    };
    if (kStream) { /* prologue: rows 0 .. 2 requested, row 0 indexed */
        lds_barrier(); /* est_l */
        stream_issue(0);
        stream_issue(1);
        stream_issue(2);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        lds_barrier(); /* every wave's part of window 0 has landed */
        stream_index(0);
        lds_barrier();
    }

    XYZI prev{0.f, 0.f, 0.f, 0.f}, prevprev{0.f, 0.f, 0.f, 0.f};
    PendingRow p1{}, p2{};              /* rows r-1 (ground flag still open) and r-2 (ready to write) */
    float zref = __uint_as_float(0x7fc00000u); /* height of the column's last candidate taken for ground (NaN: none yet) */
    unsigned long long m_ready = 0;     /* candidate ballot of row r-2 */

    const size_t cand_base = (size_t)f * g.segs * kSeg;
    uint32_t *fncand = b.ncand + (size_t)f * g.segs;
    uint32_t *flist = b.code_main + ((size_t)f * g.emitters + strip) * bands * (size_t)g.code_cap;

    /* one row; I = r mod 6 (r mod 2 at depth 1) at compile time */
    auto row_step = [&](auto I, const int r) {
        constexpr int ic = decltype(I)::value % (kDepth + 1);              /* stage holding row r */
        constexpr int in = (decltype(I)::value + kDepth) % (kDepth + 1);   /* stage that takes row r + kDepth */
        constexpr int wu = (decltype(I)::value + kDepth) % 2, wl = decltype(I)::value % 2; /* winner word used / reloaded */
        const int par = r & 1;
        if (kStream) {
            /* everything requested two steps ago — row r's override point, the winner word of row r + 2, the window of
             * row r + 1 — has arrived once only the last step's requests are outstanding: 3 asm loads + 2 LDS-DMAs
             * (6 in the first wave).  Steps 0 and 1 follow the prologue, whose order differs. */
            if (r < 2) wait_loads<0>(plo[ic], phi[ic], wraw[wu]);
            else if (wv == 0) wait_loads<9>(plo[ic], phi[ic], wraw[wu]);
            else wait_loads<5>(plo[ic], phi[ic], wraw[wu]);
        }
        Half cur_lo{{plo[ic].x, plo[ic].y, plo[ic].z, plo[ic].w}}, cur_hi{{phi[ic].x, phi[ic].y, phi[ic].z, phi[ic].w}};
        if (!pfull[ic]) { /* untouched slot: value-initialised, BatchMultiBevGen.cpp:98 */
            cur_lo = Half{{0, 0, 0, 0}};
            cur_hi = Half{{0, 0, 0, 0}};
            if (kStream && r < N) { /* ... unless the prefix holds the slot's point */
                const uint32_t e = colidx[par][tid];
                if ((e >> 16) == (uint32_t)(r + 1)) {
                    const int slot = r & (kWinSlots - 1), j = (int)(e & 0x7fffu);
                    const u32x4 a = (e & 0x8000u) ? wrapw[slot][0][j] : win[slot][0][j];
                    const u32x4 c = (e & 0x8000u) ? wrapw[slot][1][j] : win[slot][1][j];
                    cur_lo = Half{{a.x, a.y, a.z, a.w}};
                    cur_hi = Half{{c.x, c.y, c.z, c.w}};
                }
            }
        }
        if (kStream) {
            /* a window is filled by all four waves, and the order check looks one position to the left (another wave's
             * part for a wave's first lane): every wave must have passed its wait before the window is indexed */
            lds_barrier();
            stream_index(r + 1); /* the row barrier below separates the index from its readers */
            stream_issue(r + 3);
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
        if (lane == 0) wave_cnt[par][wv] = (uint32_t)__popcll(m_ready);
        lds_barrier();

        /* ---- status of row r (BatchMultiBevGen.cpp:142-182) ---- */
        int s_r = kSteep;
        if (r >= lo_row && r < N) { /* workgroup-uniform */
            /* row r-1 of the threads two to the right / left */
            XYZI right{__shfl(prev.x, lane + 2), __shfl(prev.y, lane + 2), __shfl(prev.z, lane + 2), __shfl(prev.i, lane + 2)};
            XYZI left{__shfl(prev.x, lane - 2), __shfl(prev.y, lane - 2), __shfl(prev.z, lane - 2), __shfl(prev.i, lane - 2)};
            const float4(*pe)[4] = edge[(r + 2) % 3];
            if (lane >= 62 && wv + 1 < kWaves) { const float4 q = pe[wv + 1][lane - 62]; right = XYZI{q.x, q.y, q.z, q.w}; }
            if (lane < 2 && wv > 0) { const float4 q = pe[wv - 1][lane + 2]; left = XYZI{q.x, q.y, q.z, q.w}; }
            if (outcol) {
                XYZI up = prev;                                  /* (r-1, c)                  :143     */
                if (up.i == -1.0f) up = right;                   /* (r-1, (c+2) % H)          :146-149 */
                if (up.i == -1.0f) up = left;                    /* flat (r-1)*H + c - 2      :151-154 */
                if (up.i == -1.0f && r >= 2) up = prevprev;      /* (r-2, c)                  :157-160 */
                if (cur.i == -1.0f || up.i == -1.0f) s_r = kInvalid; /* :162-167 */
                else s_r = angle_is_ground(up.x - cur.x, up.y - cur.y, up.z - cur.z) ? kGround : kSteep; /* :169-182 */
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
        const unsigned long long m_new = __ballot(cand1);
        /* Provisional labels.  Phase C un-grounds a candidate that lies 0.30 m above a neighbour cell's average ground
         * height — known only after the whole frame has been summed.  The walk GUESSES: a candidate 0.30 m above the last
         * candidate of its column that it took for ground is written with its own label, every other candidate with
         * label 0; k_ground_resolve tests every candidate exactly and patches the wrong guesses in either direction.
         * The guess only decides how many sparse 2-byte patches are needed (benchmark frames: 1.3 k instead of 7.9 k per
         * frame). */
        {
            const float zq = __uint_as_float(p1.lo.w[2]);
            p1.pred = cand1 && (zq - zref >= 0.3f); /* false while zref is NaN */
            if (cand1 && !p1.pred) zref = zq;
        }
        if (cand1) {
            const int cell = ground_cell(__uint_as_float(p1.lo.w[0]), __uint_as_float(p1.lo.w[1]));
            p1.key = candidate_key_edges(cell, tid - 2, p1.pred, p1.code, (int)(int16_t)(p1.hi.w[3] & 0xffffu),
                                         edge_x[cell / kGridCols], edge_y[cell % kGridCols]);
        }

        /* ---- write out row r-2 (its per-wave counts were published before the barrier) ---- */
        if (r >= 2) {
            const int q = r - 2;
            const bool is_cand = outcol && p2.gflag == 1;
            const int rr = q - (lo_row - 1);        /* only rows lo-1 .. N-1 can hold candidates */
            if (rr >= 0) {
                uint32_t before = 0, total = 0;
#pragma unroll
                for (int w = 0; w < kWaves; ++w) {
                    const uint32_t c = wave_cnt[par][w];
                    if (w < wv) before += c;
                    total += c;
                }
                const size_t seg = (size_t)rr * g.strips + strip;
                if (is_cand) {
                    const uint32_t rank = before + (uint32_t)__popcll(m_ready & ((1ull << lane) - 1ull));
                    const size_t at = cand_base + seg * kSeg + rank;
                    store_ws(&b.cand_key[at], p2.key);
                    store_ws(&b.cand_z[at], __uint_as_float(p2.lo.w[2]));
                }
                if (tid == 2) fncand[seg] = total;
            }
            /* BEV code of the slot.  A slot that is not a candidate has its final label, so its code is final too: it
             * is appended to this strip's list of the raster band its x bin falls into (the order inside a list does
             * not matter: an LDS cursor per band).  Candidates' codes travel in their keys.  A lane whose left
             * neighbour appends the very same code skips (near the sensor dozens of consecutive returns share a bin). */
            {
                bool has = outcol && !is_cand && p2.code != kSkip;
                const uint32_t left_code = __shfl_up(p2.code, 1);
                const bool left_has = __shfl_up(has ? 1 : 0, 1) != 0;
                if (lane > 0 && left_has && left_code == p2.code) has = false;
                /* ... and so does one whose code this strip has listed before and still remembers (rings hit the same
                 * cells at the same heights again and again: a HDL_64E frame lists 74 k codes of which 24 k are
                 * distinct).  The rasters are idempotent, so a stale or racing memo entry only costs a duplicate. */
                if (has) {
                    const uint32_t slot = (p2.code * 0x9E3779B1u) >> (32 - kSeenBits);
                    if (seen[slot] == p2.code) has = false;
                    else seen[slot] = p2.code;
                }
                if (has) {
                    const int band = band_tab[code_x(p2.code)];
                    const uint32_t pos = atomicAdd(&band_cursor[band], 1u);
                    store_ws(&flist[(size_t)band * g.code_cap + pos], p2.code);
                }
            }
            if (outcol) {
                Half hi = p2.hi;
                const bool as_ground = is_cand && !p2.pred;
                if (as_ground) hi.w[3] &= 0xffff0000u; /* label = 0, BatchMultiBevGen.cpp:245 (provisional) */
                const size_t idx = frame_off + (size_t)(q * H + v);
                Half *dst = reinterpret_cast<Half *>(b.ordered + idx);
                store_stream(dst, p2.lo);
                store_stream(dst + 1, hi);
                if (b.gm) b.gm[idx] = (int8_t)p2.gflag;
            }
        }

        /* ---- shift the pipeline ---- */
        p2 = p1;
        m_ready = m_new;
        p1.lo = cur_lo;
        p1.hi = cur_hi;
        p1.status = s_r;
        p1.gflag = 0;
        p1.key = 0u;
        p1.code = bev_code(cur.x, cur.y, cur.z, (int)(int16_t)(cur_hi.w[3] & 0xffffu), g.rp);
        prevprev = prev;
        prev = cur;
    };
    /* two extra iterations drain the pipeline */
    for (int r0 = 0; r0 < N + 2; r0 += 6) {
        row_step(std::integral_constant<int, 0>{}, r0);
        if (r0 + 1 < N + 2) row_step(std::integral_constant<int, 1>{}, r0 + 1);
        if (r0 + 2 < N + 2) row_step(std::integral_constant<int, 2>{}, r0 + 2);
        if (r0 + 3 < N + 2) row_step(std::integral_constant<int, 3>{}, r0 + 3);
        if (r0 + 4 < N + 2) row_step(std::integral_constant<int, 4>{}, r0 + 4);
        if (r0 + 5 < N + 2) row_step(std::integral_constant<int, 5>{}, r0 + 5);
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
 *           12 ballots (one per key bit): constant work however many distinct cells a slice has
 *   sum     thread t continues the running (sum, cnt) of cells t, t + 256, ... through their runs of this part
 * while the next part's keys and heights are already in flight, so the only memory round trip that is ever exposed is
 * the first one.  No intermediate of phase B touches HBM (round 1: the sorted heights bounced through global memory). */
constexpr int kCells = kGridCells;
constexpr int kCellPairs = (kCells + 1) / 2;        /* two 16-bit counters per 32-bit word */
constexpr int kHistStride = (kCellPairs + 3) / 4 * 4; /* words per wave's histogram */
constexpr int kTouchWords = (kCells + 31) / 32;
static_assert(kPartSegs * kSeg <= 4096, "a part's run start (12 bits) and length (13 bits) share a word with room to spare");

size_t cell_sums_lds_bytes()
{
    return sizeof(uint32_t) * ((size_t)kSumWaves * kHistStride + kCells + (size_t)kPartSegs * kSeg + 2 * (size_t)kCells +
                               kTouchWords + (kCells + 1) / 2 + 16);
}

__global__ __launch_bounds__(kSumThreads) void k_cell_sums(BatchPtrs b, Geometry g)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    uint32_t *hist = lds;                                 /* [kSumWaves][kHistStride]: 16-bit counts, cells 2i | 2i+1 << 16 */
    uint32_t *start = hist + kSumWaves * kHistStride;     /* [kCells]: the part's runs, start | length << 16 */
    float *zbuf = reinterpret_cast<float *>(start + kCells); /* [kPartSegs * kSeg]: the part's heights by cell */
    float *sumv = zbuf + kPartSegs * kSeg;                /* [kCells] running sums */
    float *cntv = sumv + kCells;                          /* [kCells] running counts */
    uint32_t *tbits = reinterpret_cast<uint32_t *>(cntv + kCells); /* [kTouchWords]: cells this part has touched */
    uint16_t *tlist = reinterpret_cast<uint16_t *>(tbits + kTouchWords); /* [kCells]: ... listed, in any order */
    uint32_t *misc = reinterpret_cast<uint32_t *>(tlist) + (kCells + 1) / 2; /* [0..1] list lengths (by part parity), [4..7] wave sums, [8] carry */
    uint16_t *hist16 = reinterpret_cast<uint16_t *>(hist); /* the same counters, cell c of wave w at [w * 2 * kHistStride + c] */

    const int f = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int T = g.segs, P = g.parts;
    const uint32_t *ckey = b.cand_key + (size_t)f * T * kSeg;
    const float *cz = b.cand_z + (size_t)f * T * kSeg;
    const uint32_t *fn = b.ncand + (size_t)f * T;
    constexpr int kSl = kSeg / 64;
    PH_DECL;
    PH();

    for (int k = tid; k < kSumWaves * kHistStride; k += kSumThreads) hist[k] = 0u;
    for (int k = tid; k < kTouchWords; k += kSumThreads) tbits[k] = 0u;
    if (tid < 16) misc[tid] = 0u;
    for (int c = tid; c < kCells; c += kSumThreads) {
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
            n_n[j] = (int)__shfl(counts, j);
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
                    key_n[j][k] = ckey[at + 64 * k];
                    z_n[j][k] = cz[at + 64 * k];
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

        /* hist.  Lanes of a 64-slice that hold the same cell find each other with one ballot per key bit (12 bits cover
         * 3750 cells; 0xfff is not a cell): constant work however many distinct cells the slice has.  Every lane keeps
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
                unsigned long long peers = __ballot(valid);
#pragma unroll
                for (int bit = 0; bit < 12; ++bit) {
                    const bool one = (c >> bit) & 1u;
                    const unsigned long long bal = __ballot(one);
                    peers &= one ? bal : ~bal;
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
    for (int c = tid; c < kCells; c += kSumThreads) avg[c] = sumv[c] / cntv[c]; /* :210 */
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
    for (int i = tid; i < t1 - t0; i += kResolveThreads) cnt[i] = (uint16_t)b.ncand[(size_t)f * T + t0 + i];
    for (int c = tid; c < kCells; c += kResolveThreads) avg[c] = b.avg[(size_t)f * kCells + c];
    if (tid < kGridRows) edge_x[tid] = cell_edge_bin(tid, 75.0f, g.rp);
    else if (tid < kGridRows + kGridCols) edge_y[tid - kGridRows] = cell_edge_bin(tid - kGridRows, 50.0f, g.rp);
    if (tid < kMaxBands) band_cursor[tid] = 0u;
    for (int x = tid; x < g.rp.mat_size; x += kResolveThreads) band_tab[x] = (uint8_t)raster_band_of(x, g.rp);
    lds_barrier();

    constexpr int kSl = kSeg / 64;
    constexpr int kWaves = kResolveThreads / 64;
    const int bands = g.raster_bands, lo_row = g.N - g.G;
    const uint32_t *fkey = b.cand_key + (size_t)f * T * kSeg;
    const float *fz = b.cand_z + (size_t)f * T * kSeg;
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
                    key[j][k] = fkey[at];
                    z[j][k] = fz[at];
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
                    uint16_t label = 0;             /* not un-grounded: label = 0, BatchMultiBevGen.cpp:245 */
                    if (hit) {                      /* un-grounded: the point's own label back */
                        if (kk & kKeyLabelM2Bit) label = (uint16_t)(int16_t)-2;
                        else if (kIdentity) label = reinterpret_cast<const uint16_t *>(b.pts + idx)[14];
                        else {
                            /* an EMPTY slot can be a candidate too (a value-initialised point has intensity 0, not -1,
                             * so phase A tests it like any other): its label is the zero point's 0 */
                            const uint32_t w = winner_index(b.winner[idx], b.win_tag, b.win_shift);
                            if (w != 0u) label = reinterpret_cast<const uint16_t *>(b.pts + b.frames[f].in_offset + (w - 1u))[14];
                        }
                    }
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
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)cell_sums_lds_bytes());
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
void launch_gather_ground(const Geometry &g, const BatchPtrs &b, int nf, int source, uint32_t mode, hipStream_t st)
{
    if (nf == 0) return;
    const int grid = xcd_grid(nf, g.strips);
    if (source == kSrcIdentity)
        hipLaunchKernelGGL(k_strip_ground<kSrcIdentity>, dim3(grid), dim3(kStripThreads), 0, st, b, g, nf, mode);
    else if (source == kSrcStream)
        hipLaunchKernelGGL(k_strip_ground<kSrcStream>, dim3(grid), dim3(kStripThreads), 0, st, b, g, nf, mode);
    else
        hipLaunchKernelGGL(k_strip_ground<kSrcGather>, dim3(grid), dim3(kStripThreads), 0, st, b, g, nf, mode);
}
void launch_probe(const Geometry &g, const BatchPtrs &b, int nf, bool allow_stream, hipStream_t st)
{
    if (nf == 0) return;
    hipLaunchKernelGGL(k_probe, dim3(nf), dim3(256), 0, st, b, g, allow_stream ? 1 : 0);
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
    hipLaunchKernelGGL(k_cell_sums, dim3(nf), dim3(kSumThreads), cell_sums_lds_bytes(), st, b, g);
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
