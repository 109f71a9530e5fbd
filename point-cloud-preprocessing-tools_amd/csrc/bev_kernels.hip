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
            long long d_[8] = {0, 0, 0, 0, 0, 0, 0, 0};                                           \
            for (int i_ = 1; i_ < ph_n && i_ <= 8; ++i_) d_[i_ - 1] = ph_clk[i_] - ph_clk[i_ - 1]; \
            printf("%s: %lld %lld %lld %lld %lld %lld %lld %lld (x10 ns)\n", name, d_[0], d_[1], d_[2], d_[3], d_[4], d_[5], d_[6], d_[7]); \
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

#ifndef BEV_SEENB
#define BEV_SEENB 8
#endif

namespace bevk {

/* developer aid (make tl): start, end and place of EVERY workgroup of the pipeline's kernels since the last reset — what
 * shares the chip with what, and when (scripts/pipeline_timeline.py) */
#ifdef BEV_TL_ALL
constexpr unsigned kTlAllCap = 1u << 17;
__device__ long long g_tl_all[kTlAllCap][4];
__device__ unsigned g_tl_all_n;
#define TL_BEGIN const long long tl_all_t0 = wall_clock64()
#define TL_END(kid)                                                                                               \
    do {                                                                                                          \
        if (threadIdx.x == 0) {                                                                                   \
            const unsigned i_ = atomicAdd(&g_tl_all_n, 1u);                                                       \
            if (i_ < kTlAllCap) {                                                                                 \
                g_tl_all[i_][0] = tl_all_t0;                                                                      \
                g_tl_all[i_][1] = wall_clock64();                                                                 \
                g_tl_all[i_][2] = (long long)(unsigned)__builtin_amdgcn_s_getreg((31 << 11) | 4) |                \
                                  ((long long)((unsigned)__builtin_amdgcn_s_getreg((31 << 11) | 20) & 0xfu) << 32) | \
                                  ((long long)(kid) << 40);                                                       \
                g_tl_all[i_][3] = (long long)blockIdx.x;                                                          \
            }                                                                                                     \
        }                                                                                                         \
    } while (0)
#else
#define TL_BEGIN
#define TL_END(kid)
#endif

static const char *const kNames[K_COUNT] = {
    "k_order_scan", "k_walk", "k_cell_sums", "k_ground_resolve", "k_bev_raster",
    "k_gather_only", "k_ground_mat", "k_cloud_codes", "k_angle_debug", "k_float_bev", "k_project", "k_transform",
    "k_probe", "k_walk_general", "k_walk_structured", "k_walk_colmajor", "k_walk_colmajor_gen", "k_verdict",
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
 * stream of the path.  One workgroup per frame looks at every 63rd point (kProbeStride): the leading samples that are in range and
 * strictly ascending bound a prefix [0, T) that is TAKEN for sorted; for every (row, strip) the position of its first
 * slot inside that prefix is estimated by interpolation between the two samples around it.  Nothing here is trusted:
 * the stream walk verifies every point it consumes and a frame that fails is redone the general way. */
#ifndef BEV_PROBE_THREADS
#define BEV_PROBE_THREADS 256
#endif
constexpr int kProbeThreads = BEV_PROBE_THREADS; /* one workgroup per frame */
__global__ __launch_bounds__(kProbeThreads) void k_probe(BatchPtrs b, Geometry g, int allow_stream)
{
    TL_BEGIN;
    __shared__ uint32_t samp[kMaxSamples]; /* slot of sample k (position k * kProbeStride) */
    __shared__ uint32_t first_bad, overflow;
    __shared__ uint32_t tcnt[kTailBuckets]; /* tail points listed per (row, strip) */
    const int f = blockIdx.x, tid = threadIdx.x;
    const FrameDesc fd = b.frames[f];
    const uint32_t n = fd.n_pts;
    const bev_point_t *fp = b.pts + fd.in_offset;
    const uint32_t ns = n ? (n - 1u) / kProbeStride + 1u : 0u;
    PH_DECL;
    PH();
    const bool can = allow_stream && n >= (uint32_t)kStreamMinPrefix && ns <= (uint32_t)kMaxSamples && g.N <= kStreamMaxRows &&
                     g.N * g.strips <= kTailBuckets && n < (1u << 24) && b.tail_list != nullptr;
    /* a structured cloud (kFrameStructured): exactly S records, every sampled one its own slot's point or empty */
    const bool can_struct = allow_stream && n == (uint32_t)g.S;
    /* ... or S returns in firing order (kFrameColMajor): every sampled record is beam (position mod N) of firing
     * (position / N); its column follows the firing — in either direction, from any start azimuth, with a base of its own
     * per row (staggered beams) — or is out of range, or is column 0 (a no-return record): see below */
    __shared__ uint32_t struct_bad, struct_zero, cm_bad, cm_not_plain;
    const bool can_cm = can_struct && g.N >= 2; /* (the plain sweep) */
    const bool can_cm_gen = can_cm && g.N <= kCmMaxRows && g.strips <= kCmMaxStrips && ns <= (uint32_t)kCmMaxSamples && b.cm_par != nullptr;
    __shared__ uint32_t cmrc[kCmMaxSamples], cmrc1[kCmMaxSamples]; /* row | col << 16 of every sample and of its successor */
    __shared__ uint32_t cm_ref[kCmMaxRows], cm_lo[kCmMaxRows], cm_hi[kCmMaxRows], cm_base[kCmMaxRows], cm_misc[4];
    if (tid == 0) {
        first_bad = can ? ns : 0u;
        struct_bad = 0u;
        struct_zero = 0u;
        cm_bad = 0u;
        cm_not_plain = 0u;
    }
    __syncthreads();
    if (can || can_struct) {
        /* every sample is a sector of its own somewhere in the frame: all of a thread's loads are requested before the
         * first is used (one load per trip of the plain loop was half of the kernel's time: 8 round trips under load) */
        constexpr int kSPer = 9; /* 256 x 9 samples = 145 k points per trip */
        for (uint32_t k0 = 0; k0 < ns; k0 += (uint32_t)kProbeThreads * kSPer) {
            uint32_t rc[kSPer], rc1[kSPer];
#pragma unroll
            for (int u = 0; u < kSPer; ++u) {
                const uint32_t k = k0 + (uint32_t)kProbeThreads * u + tid;
                const size_t i = (size_t)(k < ns ? k : ns - 1u) * kProbeStride;
                rc[u] = load_once(reinterpret_cast<const uint32_t *>(fp + i) + 5);                       /* row | col << 16 */
                rc1[u] = load_once(reinterpret_cast<const uint32_t *>(fp + (i + 1 < n ? i + 1 : i)) + 5); /* the sample's successor (mostly the same line): catches column-major orders at once */
            }
#pragma unroll
            for (int u = 0; u < kSPer; ++u) {
                const uint32_t k = k0 + (uint32_t)kProbeThreads * u + tid;
                if (k >= ns) continue;
                const size_t i = (size_t)k * kProbeStride;
                const uint32_t row = rc[u] & 0xffffu, col = rc[u] >> 16;
                uint32_t sl = (row < (uint32_t)g.N && col < (uint32_t)g.H) ? row * (uint32_t)g.H + col : 0xffffffffu;
                const uint32_t sl0 = sl;
                if (i + 1 < n) {
                    const uint32_t row1 = rc1[u] & 0xffffu, col1 = rc1[u] >> 16;
                    const uint32_t sl1 = (row1 < (uint32_t)g.N && col1 < (uint32_t)g.H) ? row1 * (uint32_t)g.H + col1 : 0xffffffffu;
                    if (sl1 == 0xffffffffu || sl1 <= sl) sl = 0xffffffffu;
                    if (can_struct) { /* the successor: position i + 1 >= 1 */
                        if (sl1 != (uint32_t)(i + 1) && rc1[u] != 0u) struct_bad = 1u;
                        if (rc1[u] == 0u) struct_zero = 1u;
                        if (row1 != (uint32_t)(i + 1) % (uint32_t)g.N) cm_bad = 1u;
                        if (col1 < (uint32_t)g.H && col1 - (uint32_t)(i + 1) / (uint32_t)g.N > 8u) cm_not_plain = 1u; /* (the plain sweep: column = firing + 0 .. 8) */
                    }
                }
                if (can_struct) {
                    if (sl0 != (uint32_t)i && rc[u] != 0u) struct_bad = 1u;
                    if (rc[u] == 0u && i >= 1) struct_zero = 1u;
                    if (row != (uint32_t)i % (uint32_t)g.N) cm_bad = 1u;
                    if (col < (uint32_t)g.H && col - (uint32_t)i / (uint32_t)g.N > 8u) cm_not_plain = 1u;
                }
                if (can_cm_gen) {
                    cmrc[k] = rc[u];
                    cmrc1[k] = i + 1 < n ? rc1[u] : 0xffffffffu; /* (col 0xffff: out of range, not looked at) */
                }
                if (can) samp[k] = sl;
            }
        }
        __syncthreads();
        if (can_struct && !struct_bad) { /* (the walk checks every record; a wrong guess about the empty ones is a failed frame) */
            if (tid == 0) b.info[f] = FrameInfo{n, kFrameStructured, 0u, struct_zero ? kInfoZeroGuess : 0u};
            return;
        }
        /* Firing order: which way does the sweep turn, and where does every row start?  With u = +-firing mod H the
         * displacement d = (col - u) mod H of a row's returns is the row's base plus a few columns of jitter.  Both
         * directions are tried; the one under which every row's SAMPLED displacements lie within kCmProbeDisp columns of
         * each other (and the rows' bases within kCmSpread) is taken and the bases are put kColMaxDisp - spread halves below
         * the smallest sample.  Column 0 is left out (no-return records sit there whatever their firing) and so are
         * columns >= H.  Nothing of this is trusted: the walk checks every record against its row's base. */
        if (can_cm && !cm_bad && !cm_not_plain) {
            /* the plain sweep (BASELINE config 3): starts at azimuth 0, turns forward, every sampled return within 0 .. 8 columns
             * of its firing, no no-return record among the samples: round 4's walk (which takes anything else for a defect) */
            if (tid == 0) b.info[f] = FrameInfo{n, kFrameColMajor, 0u, 0u};
            return;
        }
        if (can_cm_gen && !cm_bad) {
            const uint32_t H = (uint32_t)g.H, N = (uint32_t)g.N;
            constexpr uint32_t kBias = 1u << 20;
            /* pos / N for pos < S <= 2^20 and N <= 128 as a multiplication: with m = ceil(2^32 / N), pos * m / 2^32 exceeds
             * pos / N by less than 2^-12, and pos / N lies 1 / 128 or more below the next integer unless it is one */
            const uint32_t n_magic = (uint32_t)((0x100000000ull + N - 1u) / N);
            auto div_n = [&](uint32_t pos) -> uint32_t { return N == 1u ? pos : __umulhi(pos, n_magic); };
            for (int pass = 0; pass < 2; ++pass) {
                const bool fwd = pass == 0;
                for (uint32_t r = tid; r < N; r += kProbeThreads) {
                    cm_ref[r] = 0xffffffffu;
                    cm_lo[r] = 0xffffffffu;
                    cm_hi[r] = 0u;
                }
                if (tid < 4) cm_misc[tid] = tid == 1 ? 0xffffffffu : 0u; /* [0] failed, [1] smallest / [2] largest base offset (biased), [3] a row that has samples + 1 */
                __syncthreads();
                auto disp = [&](uint32_t pos, uint32_t rcw, uint32_t *row, uint32_t *d) -> bool { /* a sample that says something about its row's base */
                    const uint32_t col = rcw >> 16;
                    if (col == 0u || col >= H) return false;
                    const uint32_t fire = div_n(pos); /* (< H: the frame has S = N * H records) */
                    *row = pos - fire * N;
                    const uint32_t u = fwd ? fire : (fire ? H - fire : 0u);
                    *d = col >= u ? col - u : col + H - u;
                    return true;
                };
                for (uint32_t k = tid; k < ns; k += kProbeThreads) {
                    uint32_t row, d;
                    if (disp(k * kProbeStride, cmrc[k], &row, &d)) cm_ref[row] = d; /* (any sample of the row will do as its reference) */
                    if (disp(k * kProbeStride + 1u, cmrc1[k], &row, &d)) cm_ref[row] = d;
                }
                __syncthreads();
                auto rel = [&](uint32_t d, uint32_t ref) -> uint32_t { /* d - ref as a signed offset around the circle, biased */
                    const uint32_t t = d >= ref ? d - ref : d + H - ref;
                    return t > H / 2u ? kBias + t - H : kBias + t;
                };
                for (uint32_t k = tid; k < ns; k += kProbeThreads) {
                    uint32_t row, d;
                    if (disp(k * kProbeStride, cmrc[k], &row, &d)) {
                        atomicMin(&cm_lo[row], rel(d, cm_ref[row]));
                        atomicMax(&cm_hi[row], rel(d, cm_ref[row]));
                    }
                    if (disp(k * kProbeStride + 1u, cmrc1[k], &row, &d)) {
                        atomicMin(&cm_lo[row], rel(d, cm_ref[row]));
                        atomicMax(&cm_hi[row], rel(d, cm_ref[row]));
                    }
                }
                __syncthreads();
                for (uint32_t r = tid; r < N; r += kProbeThreads) {
                    if (cm_ref[r] == 0xffffffffu) continue; /* a row without a usable sample: takes another row's base below */
                    const uint32_t spread = cm_hi[r] - cm_lo[r];
                    if (spread > (uint32_t)kCmProbeDisp) cm_misc[0] = 1u;
                    /* base = reference + smallest offset - half of the slack, mod H (offsets are small against H, or H is tiny and anything goes) */
                    const uint32_t slack = ((uint32_t)kColMaxDisp - (spread < (uint32_t)kColMaxDisp ? spread : (uint32_t)kColMaxDisp) + 1u) / 2u;
                    const uint32_t off = cm_lo[r] - slack; /* biased */
                    cm_base[r] = (cm_ref[r] + (off % H) + (H - kBias % H)) % H;
                    cm_misc[3] = r + 1u;
                }
                __syncthreads();
                if (cm_misc[0] == 0u && cm_misc[3] != 0u) {
                    const uint32_t r0 = cm_misc[3] - 1u, bc = cm_base[r0];
                    for (uint32_t r = tid; r < N; r += kProbeThreads) {
                        if (cm_ref[r] == 0xffffffffu) cm_base[r] = bc;
                        atomicMin(&cm_misc[1], rel(cm_base[r], bc));
                        atomicMax(&cm_misc[2], rel(cm_base[r], bc));
                    }
                    __syncthreads();
                    const uint32_t max_spread = (uint32_t)kCmSpread;
                    if (cm_misc[2] - cm_misc[1] <= max_spread) {
                        /* does the frame hold no-return records (column 0, away from where the firing's returns lie)?  Then
                         * its strips talk to each other about them (k_walk: listen_band); a frame whose samples show none is
                         * walked without that — and redone if a record the samples missed turns out to matter */
                        if (tid == 0) cm_misc[0] = 0u;
                        __syncthreads();
                        for (uint32_t k = tid; k < ns; k += kProbeThreads) {
#pragma unroll
                            for (int w = 0; w < 2; ++w) {
                                const uint32_t rcw = w ? cmrc1[k] : cmrc[k], pos = k * kProbeStride + (uint32_t)w;
                                if ((rcw >> 16) != 0u) continue;
                                const uint32_t fire = div_n(pos), row = pos - fire * N;
                                const uint32_t u = fwd ? fire : (fire ? H - fire : 0u);
                                const uint32_t d = (2u * H - u - cm_base[row]) % H; /* (0 - u - base) mod H */
                                if ((rcw & 0xffffu) == row && d > (uint32_t)kColMaxDisp) cm_misc[0] = 1u;
                            }
                        }
                        __syncthreads();
                        int32_t *par = b.cm_par + (size_t)f * kCmParWords;
                        uint32_t *sync = b.cm_sync + (size_t)f * kCmSyncWords;
                        for (uint32_t r = tid; r < N; r += kProbeThreads) par[2 + r] = (int32_t)cm_base[r];
                        for (uint32_t i = tid; i < (uint32_t)kCmSyncWords; i += kProbeThreads) sync[i] = 0u;
                        if (tid == 0) {
                            par[0] = fwd ? 1 : -1;
                            par[1] = (int32_t)((bc + (cm_misc[2] % H) + (H - kBias % H)) % H); /* the largest base */
                            par[2 + kCmMaxRows] = (int32_t)(cm_misc[2] - cm_misc[1]);         /* how far apart the bases lie */
                            par[3 + kCmMaxRows] = (int32_t)cm_misc[0];                         /* a sample was a no-return record */
                            b.info[f] = FrameInfo{n, kFrameColMajorGen, 0u, 0u};
                        }
                        return;
                    }
                }
                __syncthreads();
            }
        }
    }
    if (can) {
        for (uint32_t k = tid; k < ns; k += (uint32_t)kProbeThreads) /* first sample that is out of range or not above its predecessor */
            if (samp[k] == 0xffffffffu || (k > 0u && samp[k] <= samp[k - 1u])) atomicMin(&first_bad, k);
        __syncthreads();
    }
    PH(); /* samples */
    const uint32_t m = first_bad;                                      /* samples 0 .. m-1 ascend */
    const uint32_t T0 = m ? (m - 1u) * kProbeStride + 1u : 0u;         /* the last of them is position T0 - 1 */
    /* ... and the points after it, one by one, up to the first that does not ascend (at the latest the successor of the
     * sample that failed): a sweep that is sorted to its end has no tail at all, and an appended block of other points
     * starts exactly where the prefix ends — otherwise up to 62 sorted points of ONE (row, strip) would be "tail" */
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
    PH(); /* prefix end */
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
        fest[st * g.N + r] = est < T ? est : T; /* [strip][row]: a strip's workgroup reads its 64 rows as two lines, not 64 sectors */
    }

    PH(); /* estimates */
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
    constexpr int kPer = 20; /* loads in flight per thread: a 5000-point tail is one trip */
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
            if (col < 2) { /* the wrap-around halo of the last strip — and of the one before it when the last strip owns one column */
                for (int ws = g.strips - 1; ws >= 0 && ws >= g.strips - 2; --ws) {
                    const int off = g.H + col - (ws * kStripCols - 2);
                    if (off < kStripVirt && ws * kStripCols < g.H) append(row, ws, off, i);
                }
            }
        }
    }
    __syncthreads();
    PH(); /* tail lists */
    uint32_t *fcnt = b.tail_cnt + (size_t)f * g.N * g.strips;
    for (int i = tid; i < g.N * g.strips; i += kProbeThreads) { /* [strip][row], as the estimates */
        const int r = i / g.strips, st = i - r * g.strips;
        fcnt[st * g.N + r] = tcnt[i] < (uint32_t)kTailCap ? tcnt[i] : (uint32_t)kTailCap;
    }
    /* a list that does not hold its (row, strip)'s tail points: the frame goes the general way (the scan repeats the
     * scatter of the tail among all the others) */
    if (tid == 0) b.info[f] = overflow ? FrameInfo{0u, kFrameGeneral, 4u, 0u} : FrameInfo{T, kFrameStream, 0u, 0u};
    PH();
    TL_END(K_PROBE);
    PH_PRINT("probe samples prefix-end estimates tail-lists counts", tid == 0 && f == 100);
}

/* after the stream walk: a frame whose consumed points do not add up to its prefix, or with a failed check, is redone */
/* ... and the host is told, without being waited for, how many frames of the sub-batch are NOT read in place (a word in
 * mapped host memory): the next sub-batches' order scan is launched thin or wide by it — a hint about speed, the thin
 * and the wide launch compute the same */
/* host_hint[1]: which modes k_probe gave the sub-batch's frames (bit = mode).  The host launches the walk of a mode only
 * while the workspace set's last sub-batches had frames of it — a frame whose walk was not launched fails the count
 * below and is redone the general way, so a stale hint costs time, never results. */
__global__ __launch_bounds__(1024) void k_verdict(FrameInfo *info, int nf, uint32_t *host_hint, const uint32_t *cm_sync, int N)
{
    __shared__ uint32_t others, modes;
    if (threadIdx.x == 0) others = modes = 0u;
    __syncthreads();
    uint32_t mine = 0u, mask = 0u;
    for (int f = threadIdx.x; f < nf; f += 1024) {
        FrameInfo fi = info[f];
        mask |= 1u << (fi.mode & 31u);
        bool bad_stream = (fi.mode == kFrameStream || fi.mode == kFrameColMajor || fi.mode == kFrameColMajorGen) && ((fi.failed & kInfoFailed) != 0u || fi.consumed != fi.T);
        /* firing order: where a strip's wrap-around halo fell back on column 0 (BatchMultiBevGen.cpp:146-149; rare: the upper
         * point's intensity is -1) it must have taken the record that strip 0 — which hears of every no-return record of the
         * row — put there */
        if (fi.mode == kFrameColMajorGen && !bad_stream && (fi.failed & (kInfoCmUsed | kInfoCmStray)) != 0u && cm_sync) {
            const uint32_t *win0 = cm_sync + (size_t)f * kCmSyncWords + kCmPubWords, *used0 = win0 + kCmMaxRows, *stray = used0 + kCmMaxRows;
            for (int r = 0; r < N && r < kCmMaxRows; ++r) {
                const uint32_t u = used0[r];
                if ((u & kCmUsedBit) != 0u && (u & ~kCmUsedBit) != win0[r]) bad_stream = true;
                /* ... and in a frame whose strips did not talk, no no-return record of another strip's may be later in the
                 * input than what strip 0 put into column 0 */
                if (stray[r] > win0[r]) bad_stream = true;
            }
        }
        /* structured: every record checked, none bad, and the guess about all-zero records (it decided slot 0) was right */
        const bool bad_struct = fi.mode == kFrameStructured &&
                                ((fi.failed & kInfoFailed) != 0u || fi.consumed != fi.T ||
                                 ((fi.failed & kInfoZeroSeen) != 0u) != ((fi.failed & kInfoZeroGuess) != 0u));
        if (bad_stream || bad_struct) {
            info[f].mode = kFrameRedo;
            fi.mode = kFrameRedo;
        }
        mine += frame_read_in_place(fi.mode) ? 0u : 1u;
    }
    if (mine) atomicAdd(&others, mine);
    if (mask) atomicOr(&modes, mask);
    __syncthreads();
    if (threadIdx.x == 0 && host_hint) {
        __hip_atomic_store(host_hint, others, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(host_hint + 1, modes, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
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
                                                    const FrameInfo *__restrict__ info,
                                                    uint32_t *__restrict__ winner, int N, int H, int S,
                                                    uint32_t tag_bits)
{
    /* One launch, after the in-place walk and its verdict: the frames that are NOT read in place — general ones and
     * those whose verification failed.  (A frame read in place has no winner entries; its tail is listed by k_probe.)
     * gridDim.x workgroups per frame stride over its 1024-point blocks (launch_order_scan: one per block, or 8 per frame
     * for the launch that is expected to find nothing to do). */
    const int f = blockIdx.y;
    if (info && frame_read_in_place(info[f].mode)) return;
    const FrameDesc fd = frames[f];
    const bev_point_t *fp = pts + fd.in_offset;
    uint32_t *fw = winner + (size_t)f * S;
    __shared__ uint32_t any_spread2[2]; /* (by block parity: a thread may still read one block's flag while the next block's is cleared) */
    __shared__ uint32_t row_fill[kScanRowBins];
    /* (slot << kScanIdxBits | index within the block) regrouped by row; 4 B per point, not 8: LDS is what decides how many of these
     * blocks fit on a CU beside the back end's workgroups of another sub-batch */
    __shared__ uint32_t pairs[256 * kScanPerThread];
    static_assert(256 * kScanPerThread == (1 << kScanIdxBits), "bits of block-local index");
  uint32_t turn = 0u;
  for (uint32_t blk = blockIdx.x; blk * (256u * kScanPerThread) < fd.n_pts; blk += gridDim.x, ++turn) { /* (uniform trip count) */
    uint32_t &any_spread = any_spread2[turn & 1u];
    const uint32_t block0 = blk * (256u * kScanPerThread);
    const uint32_t base = block0 + threadIdx.x;
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
        slot[k] = (i < fd.n_pts && row < (uint32_t)N && col < (uint32_t)H) ? row * (uint32_t)H + col
                                                                                         : 0xffffffffu; /* :106-111 ("< 0" is dead: u16) */
    }
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
    if (threadIdx.x == 0) any_spread = 0u;
    __syncthreads();
    if (spread && (threadIdx.x & 63) == 0) any_spread = 1u;
    __syncthreads();
    if (any_spread == 0u || N > kScanRowBins || S > (1 << (32 - kScanIdxBits))) {
        /* coalesced already (or too many rows to bin): one atomicMax per point, in input order */
#pragma unroll
        for (int k = 0; k < kScanPerThread; ++k)
            if (slot[k] != 0xffffffffu) atomicMax(&fw[slot[k]], tag_bits | (base + 256u * k + 1u));
        continue; /* (workgroup-uniform) */
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
        if (pr != 0xffffffffu) atomicMax(&fw[pr >> kScanIdxBits], tag_bits | (block0 + (pr & ((1u << kScanIdxBits) - 1u)) + 1u));
    }
  }
}

/* ------------------------------------------------------------------------- */
/* getOrderedCloud gather + markGroundPoints phase A, as a COLUMN WALK.
 *
 * A workgroup owns kStripCols (236) adjacent columns of one frame plus two halo columns on each side (240 virtual columns,
 * 256 threads) and walks the rows 0 .. N-1.  Thread tid sits on virtual column v = strip*236 + tid - 2 and, in row r, on flat slot
 * index r*H + v (v >= H wraps to v - H in the SAME row, v < 0 is the flat index r*H + v, i.e. the tail of row r-1 —
 * exactly the two index rules of BatchMultiBevGen.cpp:146-154).  Consequences:
 *   - every input point is loaded exactly once, rows arrive as 8 KiB coalesced pieces, two rows ahead;
 *   - the phase-A stencil needs no second pass: "upper" is the thread's own previous row (registers), its +-2
 *     fallbacks are the neighbours' previous rows (wave shuffles, LDS only across wave edges), row-2 is the thread's
 *     own row before that;
 *   - status s[r] is evaluated ONCE per slot; ground_mat(r-1) follows from s[r-1] and s[r] (closed form in
 *     bev_exact.h), so row r-1 is finished while row r is being evaluated, and row r-2 is written out.
 * Candidates of one (row, strip) are compacted in column order into their own segment; segments enumerate (row, strip)
 * in row-major order, so the concatenation of all segments is slot order — what phase B's accumulation order needs.
 *
 * Round 3 rebuilt the kernel around three measurements:
 *   1. hipcc drained the memory queue (s_waitcnt vmcnt(0)) at the top of EVERY row step: gfx9-family loads and stores
 *      retire out of order with respect to each other, so with stores pending the compiler cannot count, and the "two
 *      rows in flight" were one row in flight plus a full round trip per step.  Every global READ of the row loop is
 *      now an LDS-DMA load (global_load_lds: per-lane source address, the data lands in LDS, no VGPR destination the
 *      compiler could copy or spill while the load is in flight), issued two steps ahead and waited for with a COUNTED
 *      s_waitcnt: "a load has completed once at most as many operations are outstanding as loads were issued after it"
 *      holds whatever the stores in between do; the stores of a step are issued BEFORE its loads, so that the wait at
 *      the top of a step covers stores that are a whole step old and loads that are two.
 *   2. a fifth of the walk's vector instructions were v_readlane restores of spilled scalar registers: the raster
 *      constants came back as an 8-dword tuple for every multiplication, and pointers laundered through asm turned
 *      every store into a FLAT store (which also counts on lgkmcnt, the LDS counter).  The raster constants live in
 *      vector registers (they only feed VALU), the power-of-two / divide choice is a template parameter, stores go
 *      through address-space-1 pointers (global_store, scalar base + 32-bit lane offset).
 *   3. waves without a column (the last strip of a row holds 67 of 256 threads for HDL_64E, 16 for OS1_64) end before
 *      the row loop: an ended wave drops out of s_barrier.
 *
 * Three sources of the points (template parameter):
 *   kSrcGather    through the winner table of the order scan (any input);
 *   kSrcIdentity  b.pts already is an ordered cloud (bev_mark_ground);
 *   kSrcInPlace   the input's first T points are in strictly ascending slot order (k_probe): they are read IN PLACE,
 *                 coalesced, once — no order scan, no winner table.  Row rho's points of this strip's 256 virtual
 *                 columns are consecutive in the input and start near est[rho][strip]; the workgroup DMAs a window of
 *                 256 positions (est - 12 ..., one per thread) into LDS, every thread looks at the (row, col) its window position
 *                 carries and enters the position into an index row at the point's column offset; the points listed for
 *                 the (row, strip) after the prefix ("tail", at most kTailCap, k_probe) are DMAed beside the window and
 *                 entered with a key that beats every prefix entry and every EARLIER tail point (LDS atomicMax: the
 *                 reference's scatter keeps the last writer, BatchMultiBevGen.cpp:112); after the step's barrier each
 *                 column's owner follows its index entry to its point; an entry whose (row, col) is not the slot's own
 *                 is an empty slot.  Nothing of this is trusted: a position holding a point of the strip's OWN columns
 *                 counts it and checks that its predecessor in the input lies in the prefix and has a smaller slot;
 *                 when all T prefix points of a frame have been counted exactly once and no check has failed, the
 *                 prefix is strictly ascending, every point was where its strip looked, and the result is what
 *                 getOrderedCloud's scatter gives; otherwise k_verdict sends the frame through the general kernels. */
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
/* row record of the walk: flags = (status + 1) | (ground_mat + 1) << 2 | pred << 4 */
struct WalkRow {
    u32x4 lo, hi;
    uint32_t code, key, fl;
};
__device__ __forceinline__ int wr_status(uint32_t fl) { return (int)(fl & 3u) - 1; }
__device__ __forceinline__ int wr_gflag(uint32_t fl) { return (int)((fl >> 2) & 3u) - 1; }

enum : int { kSrcGather = 0, kSrcIdentity = 1, kSrcInPlace = 2, kSrcStructured = 3, kSrcColMajor = 4, kSrcColMajorGen = 5 };
/* Column-major source (kFrameColMajor): input position k holds the return of firing k / N, beam k % N — what the MulRan
 * selector writes (MulranPointCloudSelect.cpp:112-130: row = k % 64, col from the azimuth).  With u = +-firing mod H (the
 * sweep's direction) a return of row r sits in column (u + B[r] + 0 .. kColMaxDisp) mod H (k_probe found the direction and
 * the rows' bases B).  A strip's threads take one u each, from kColMaxDisp + the largest base before the strip's first
 * virtual column on (kCmExt more by wave 0: 272 firings cover 240 columns, the jitter and bases kCmSpread apart); the
 * records of kBandRows consecutive rows of a firing are 64 contiguous bytes of the input, fetched as one band. */
constexpr int kBandRows = 2;
/* the PLAIN sweep (kFrameColMajor: starts at azimuth 0, turns forward, column = firing + 0 .. 8, no no-return records; BASELINE
 * config 3) keeps round 4's walk: a thread per firing from kColLead firings before the strip's first own column, side windows of
 * the first / last kPlainSide firings, 50 KB of LDS.  Everything else in firing order takes the general form below (kFrameColMajorGen). */
constexpr int kPlainDisp = 8, kColLead = 2 + kPlainDisp, kPlainSide = 16;
constexpr int kPlainBuf = kStripThreads * 32 * kBandRows + 2 * kPlainSide * 32 * kBandRows; /* one band buffer: the band, the flat-rule window, the wrap-around window */
static_assert(kPlainSide * 2 * kBandRows == 64 && kStripVirt + kPlainDisp <= kStripThreads && kPlainDisp + 2 <= kPlainSide, "the plain sweep's windows");
constexpr int kSideFirings = 32; /* firings of the side area: the wrap-around halo's window or strip 0's flat-index halo's */
constexpr int kBandBytes = kStripThreads * 32 * kBandRows;
constexpr int kExtBytes = kCmExt * 32 * kBandRows;
constexpr int kSideBytes = kSideFirings * 32 * kBandRows;
constexpr int kSpecialBytes = 32 * kBandRows;            /* strip 0: the last no-return record of either row that another strip owns */
constexpr int kColBuf = kBandBytes + kExtBytes + kSideBytes + kSpecialBytes; /* one band buffer */
/* where a record sits in a band buffer, as the index row remembers it: 0 .. 255 a thread's, then kCmExt extra firings,
 * kSideFirings side firings, the special record; all but the first 256 are 64-byte entries behind the band */
constexpr uint32_t kLocExt = kStripThreads, kLocSide = kLocExt + kCmExt, kLocSpecial = kLocSide + kSideFirings, kLocBits = 9;
static_assert(kLocSpecial < (1u << kLocBits) && kCmExt * 2 * kBandRows == 64, "location bits; the extra firings of a band are one LDS-DMA instruction");
static_assert(kStripVirt + kColMaxDisp + kCmSpread <= kStripThreads + kCmExt && 2 + kColMaxDisp + kCmSpread <= kSideFirings,
              "firings a strip's columns can come from");
constexpr uint32_t kCmSpins = 1u << 20;
constexpr int kWinPos = kStripThreads; /* in-place source: window positions of a (row, strip), one per thread: est - kWinLead ... */
constexpr int kWinLead = 12;
constexpr int kWrapPos = 16;       /* ... the last strip's wrap-around halo: positions around the row's start */
constexpr int kWrapLead = 6;
/* bytes of one ring slot: the window's low halves (4 KiB), its high halves (4 KiB), then, 32 B each, the wrap-around
 * positions and the tail points */
constexpr int kInPlaceSlot = (kWinPos + kWrapPos + kTailCap) * 32;
constexpr uint32_t kIdxTail = 1u << 30;

#ifdef BEV_TL_ALL
} // namespace bevk
extern "C" int bev_tl_all(long long *out, int cap, int reset)
{
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    unsigned n = 0;
    if (hipMemcpyFromSymbol(&n, HIP_SYMBOL(bevk::g_tl_all_n), sizeof n) != hipSuccess) return -1;
    if (n > bevk::kTlAllCap) n = bevk::kTlAllCap;
    if ((int)n > cap) n = (unsigned)cap;
    if (out && n && hipMemcpyFromSymbol(out, HIP_SYMBOL(bevk::g_tl_all), (size_t)n * 4 * sizeof(long long)) != hipSuccess) return -1;
    if (reset) {
        const unsigned z = 0;
        if (hipMemcpyToSymbol(HIP_SYMBOL(bevk::g_tl_all_n), &z, sizeof z) != hipSuccess) return -1;
    }
    return (int)n;
}
namespace bevk {
#endif
#ifdef BEV_CS_CLOCK /* developer build: start, end, HW_ID, XCC_ID of every workgroup of the last in-place walk launch */
constexpr int kWalkTlCap = 8192;
__device__ long long g_walk_tl[kWalkTlCap][4];
} // namespace bevk
extern "C" int bev_clk_walk_timeline(long long *out, int cap)
{
    const int n = cap < bevk::kWalkTlCap ? cap : bevk::kWalkTlCap;
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(bevk::g_walk_tl), (size_t)n * 4 * sizeof(long long)) == hipSuccess ? n : -1;
}
namespace bevk {
#endif
/* A wave's candidates per cell quarter.  Every candidate lane holds a one in the byte of its quarter; an inclusive scan
 * over the wave's lanes (six DPP additions: four inside the rows of 16 lanes, two across rows) leaves in lane 63 the
 * wave's four counts (at most 64 each) and in every lane, in the byte of its quarter, its rank among the wave's candidates
 * of that quarter plus one.  No ballots, no 64-bit lane masks.  Returns the scan; *rank = this lane's rank. */
__device__ __forceinline__ uint32_t quarter_scan(bool c, uint32_t q, uint32_t *rank)
{
    const uint32_t sh = q << 3;
    const uint32_t one = c ? 1u << sh : 0u;
    uint32_t x = one;
    /* (a lane whose source lies outside its row / outside the row mask keeps the 0 given as the old value) */
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111 /* row_shr:1 */, 0xf, 0xf, false);
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112 /* row_shr:2 */, 0xf, 0xf, false);
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114 /* row_shr:4 */, 0xf, 0xf, false);
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118 /* row_shr:8 */, 0xf, 0xf, false);
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142 /* row_bcast:15 */, 0xa, 0xf, false);
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143 /* row_bcast:31 */, 0xc, 0xf, false);
    *rank = __builtin_amdgcn_ubfe(x - one, sh, 8u);
    return x;
}
constexpr int kFlRankShift = 8; /* WalkRow::fl bits 8..13: the lane's rank among its wave's candidates of its quarter */

template <int kSrc, bool kPow2, bool kGm>
__global__ __launch_bounds__(kStripThreads, (kSrc == kSrcColMajor || kSrc == kSrcColMajorGen) ? 3 : 4) void k_walk(BatchPtrs b, Geometry g, int nf, uint32_t want_mode)
{
    TL_BEGIN;
    /* kStructured: the identity source over the caller's INPUT (record i = slot i's point or an all-zero record), every
     * record checked; kIdentity below covers both (no winner table, position = slot) */
    constexpr bool kStructured = kSrc == kSrcStructured, kIdentity = kSrc == kSrcIdentity || kStructured, kInPlace = kSrc == kSrcInPlace;
    /* kIndexed: the sources whose points reach their columns through an index row (LDS atomicMax), after the step's barrier */
    constexpr bool kCmGen = kSrc == kSrcColMajorGen, kColMajor = kSrc == kSrcColMajor || kCmGen, kIndexed = kInPlace || kColMajor;
    constexpr int kCmBuf = kCmGen ? kColBuf : kPlainBuf; /* bytes of one band buffer */
    static_assert(kWinPos == 256 && kStripVirt + 16 <= kWinPos && kTailCap == 64 && kWrapPos == 16, "DMA pieces of the in-place source");
    int f, strip;
#ifdef BEV_CS_CLOCK
    const long long tl_t0 = wall_clock64();
#endif
    if (!map_block_xcd(blockIdx.x, nf, g.strips, f, strip)) return;
    /* firing order: strip 0 listens to the other strips of its frame (no-return records, see listen_band): it is dispatched
     * LAST of them, and finds them under way (dispatched first it waited a quarter of its life for them to start: the walk
     * 5 % slower) */
    if (kSrc == kSrcColMajorGen) strip = g.strips - 1 - strip;
    if (kSrc != kSrcIdentity && b.info) { /* the launch for its mode has the frame; the general launch has every frame that is not read in place */
        const uint32_t fmode = b.info[f].mode;
        if (frame_read_in_place(want_mode) ? fmode != want_mode : frame_read_in_place(fmode)) return;
    }
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int N = g.N, H = g.H, lo_row = g.N - g.G, strips = g.strips;
    const size_t frame_off = (size_t)f * g.S;
    const int bands = g.raster_bands;

    /* the value two lanes to the right / left (wrapping inside the wave; the edge lanes are patched from LDS).  (Two DPP
     * wave shifts instead of each ds_bpermute measured the same.) */
    const int sh_right = ((lane + 2) & 63) << 2, sh_left = ((lane - 2) & 63) << 2;
    auto from_right2 = [&](float x) -> float { return __uint_as_float((uint32_t)__builtin_amdgcn_ds_bpermute(sh_right, (int)__float_as_uint(x))); };
    auto from_left2 = [&](float x) -> float { return __uint_as_float((uint32_t)__builtin_amdgcn_ds_bpermute(sh_left, (int)__float_as_uint(x))); };
    const int v = strip * kStripCols + tid - 2;                      /* virtual column */
    const bool provider = tid < kStripVirt && (v < H + 2) && (v >= 0 || strip == 0); /* has a slot */
    const bool outcol = tid >= 2 && tid < 2 + kStripCols && v < H;   /* owns column v's outputs */
    const int vcol = v >= H ? v - H : v;                             /* wrap; v < 0 keeps the flat rule */

    constexpr int kWaves = kStripThreads / 64;
    constexpr int kSlotBytes = kInPlace ? kInPlaceSlot : 8192;
    constexpr int kSeenB = kIndexed ? BEV_SEENB : kSeenBits;       /* (the in-place source needs the LDS for its windows) */
    /* the points of rows r, r+1, r+2.  Gather / identity: by thread, low halves in the first 4 KiB, high halves in the
     * second.  In place: by window position, 32 B each, then the wrap-around positions, then the tail points */
    /* column-major: two band buffers, then 8 KiB for the write-out's transposition */
    __shared__ __attribute__((aligned(16))) char ring[kColMajor ? 2 * kCmBuf + 8192 : 3 * kSlotBytes];
    __shared__ uint32_t wring[kSrc == kSrcGather ? 3 : 1][kStripThreads]; /* raw winner words of rows r+2, r+3, r+4 */
    __shared__ uint32_t idx[kIndexed ? 2 : 1][kIndexed ? kStripThreads + 1 : 1]; /* column offset -> position + 1 | tail key ([256]: nowhere) */
    __shared__ u32x4 zero16[kInPlace ? 1 : 1];                               /* what an empty slot reads */
    __shared__ uint32_t tlist[kInPlace ? 3 : 1][kInPlace ? 64 : 1];       /* tail lists of rows r+2, r+3, r+4 */
    __shared__ int est_l[2][kInPlace ? kStreamMaxRows : 1];
    __shared__ uint8_t tcnt_l[kInPlace ? kStreamMaxRows : 1];
    __shared__ float4 edge[3][kWaves][4];                  /* rows r, r-1, (r-2): lanes 0, 1, 62, 63 of every wave */
    /* per-wave candidate counts of the row being written, at [.][kWaves + wave] behind kWaves words that stay zero: the
     * three words before a wave's own are the counts of the waves before it, whichever wave it is (no selects) */
    __shared__ __attribute__((aligned(16))) uint32_t wave_cnt[2][2 * kWaves];
    __shared__ uint32_t band_cursor[kMaxBands];            /* entries already in this strip's code list of each band */
    __shared__ uint8_t band_tab[512];                      /* x bin -> raster band */
    __shared__ uint32_t seen[1 << kSeenB];                 /* direct-mapped memo of codes this strip has already listed */
    __shared__ int edge_x[kGridRows], edge_y[kGridCols];   /* BEV bin of every ground-grid row's / column's lower edge */
    if (tid < kMaxBands) band_cursor[tid] = 0u;
    if (tid < 4 * kWaves) (&wave_cnt[0][0])[tid] = 0u;
    if (tid < 3 * kWaves * 4) (&edge[0][0][0])[tid] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int k = tid; k < (1 << kSeenB); k += kStripThreads) seen[k] = kSkip;
    for (int x = tid; x < g.rp.mat_size; x += kStripThreads) band_tab[x] = (uint8_t)raster_band_of_nodiv(x, g.rp);
    if (tid < kGridRows) edge_x[tid] = cell_edge_bin(tid, 75.0f, g.rp);
    else if (tid < kGridRows + kGridCols) edge_y[tid - kGridRows] = cell_edge_bin(tid - kGridRows, 50.0f, g.rp);
    if constexpr (kIndexed) {
        idx[0][tid] = 0u;
        idx[1][tid] = 0u;
        if (tid == 0) zero16[0] = u32x4{0u, 0u, 0u, 0u};
    }
    if constexpr (kInPlace) {
        const uint32_t *fe = b.est + (size_t)f * N * strips;
        const uint32_t *fc = b.tail_cnt + (size_t)f * N * strips;
        for (int r = tid; r < N; r += kStripThreads) {
            est_l[0][r] = (int)fe[strip * N + r];
            est_l[1][r] = (int)fe[r];
            tcnt_l[r] = (uint8_t)fc[strip * N + r];
        }
    }
    lds_barrier();
    /* a wave none of whose threads has a column ends here (its counts stay zero, nobody reads its edge lanes: the
     * threads that would are not output columns; the in-place source needs every wave for its windows) */
    if (!kIndexed && __ballot(provider) == 0ull) return;

    const bev_point_t *fpts = kSrc == kSrcIdentity ? (b.pts + frame_off) : (b.pts + b.frames[f].in_offset);
    const uint32_t *fwin = b.winner + frame_off;
    const uint32_t win_tag = b.win_tag;
    const int win_shift = b.win_shift;
    /* an empty slot loads a dummy (the first point of this frame's OUTPUT: always allocated, one cached line) and is
     * zeroed when the row is consumed: every step issues the same loads */
    const Half *dummy = reinterpret_cast<const Half *>(b.ordered + frame_off);
    auto has_slot = [&](int r) -> bool { return provider && r < N && r * H + vcol >= 0; };
    const uint32_t ring_l = __builtin_amdgcn_readfirstlane(lds_addr(&ring[0]));
    const uint32_t wring_l = __builtin_amdgcn_readfirstlane(lds_addr(&wring[0][0])) + (uint32_t)wv * 256u;
    auto clamp_row = [&](int q) -> int { return q < N ? q : N - 1; };

    /* ---- gather / identity: winner words two steps before the points, points two steps before the row ---- */
    auto issue_winner = [&](int q, int slot) {
        if constexpr (kSrc == kSrcGather) {
            const int fl = has_slot(q) ? q * H + vcol : 0;
            glds4_nt(&fwin[fl], wring_l + (uint32_t)slot * 1024u);
        }
    };
    auto issue_points = [&](uint32_t w, int slot) { /* w: input index + 1, 0 = empty slot */
        const Half *src = w != 0u ? reinterpret_cast<const Half *>(fpts + (w - 1u)) : dummy;
        const uint32_t at = ring_l + (uint32_t)slot * kSlotBytes + (uint32_t)wv * 1024u;
        glds16x2(src, at, src + 1, at + 4096u);
    };
    auto winner_of = [&](int q, uint32_t raw) -> uint32_t { /* input index + 1 of slot (q, this column), 0 = empty */
        if (!has_slot(q)) return 0u;
        if (kIdentity) return (uint32_t)(q * H + vcol) + 1u;
        return winner_index(raw, win_tag, win_shift);
    };
    uint32_t full = 0u; /* bit (row mod 3): the row's slot holds a point */

    /* ---- in place ---- */
    const uint32_t T = kIndexed ? b.info[f].T : 0u;
    /* the strips whose virtual columns reach past the row's end and wrap to its start: the last one — and the one before it
     * when the last strip owns a single column (H mod 236 == 1: column H - 2 then belongs to the strip before, and its
     * (c + 2) % H fallback is column 0).  Found by the round-4 property test on a 473-column sensor: until then only the
     * last strip fetched its wrap-around halo in the indexed sources. */
    const bool last_strip = strip * kStripCols - 2 + kStripVirt > H;
    const int first_col = strip * kStripCols - 2; /* virtual column of offset 0 */
    const int own_cols = (H - first_col - 2) < kStripCols ? (H - first_col - 2) : kStripCols; /* own columns of this strip */
    const int row_span = (H - first_col) < kStripVirt ? (H - first_col) : kStripVirt;        /* offsets that belong to the row */
    const uint32_t *ftail = kInPlace ? b.tail_list + ((size_t)f * N * strips + strip) * kTailCap : nullptr;
    const int tail_stride = strips * kTailCap;  /* words from one row's list to the next */
    const uint32_t tlist_l = __builtin_amdgcn_readfirstlane(lds_addr(&tlist[0][0]));
    uint32_t te[3] = {0u, 0u, 0u}; /* wave 3: this lane's tail entry of rows q at [q % 3] (column offset | input index << 8) */
    uint32_t consumed = 0u, failed = 0u;
    /* structured: the (row | col << 16) word the record of this thread's slot in row r must carry is (r - st_rowadj) | st_col
     * (the flat rule puts virtual columns < 0 into the previous row's tail); whether k_probe expects an all-zero record
     * after the first — slot 0 is all-zero then, whatever record 0 holds (BatchMultiBevGen.cpp:112-115, last writer) */
    const uint32_t st_rowadj = v < 0 ? 1u : 0u, st_col = (uint32_t)(v < 0 ? H + v : vcol) << 16;
    const bool st_zero_guess = kStructured && (b.info[f].failed & kInfoZeroGuess) != 0u;
    const char *fbytes = reinterpret_cast<const char *>(fpts);
    auto pos_addr = [&](int q) -> const char * { /* the point at input position q, or position 0 outside the prefix */
        return fbytes + (size_t)((unsigned)q < T ? q : 0) * 32u;
    };
    auto issue_window = [&](int q, int slot) { /* this wave's 64 positions of row q's window: low halves, high halves */
        const int e = est_l[0][clamp_row(q)] - kWinLead;
        const uint32_t at = ring_l + (uint32_t)slot * kSlotBytes + (uint32_t)wv * 1024u;
        /* (rows past the last one — the two steps that drain the pipeline and the two before them — still issue their
         * loads, so that every step counts the same: all lanes fetch position 0, one line instead of the last row's window again) */
        const char *src = q >= N ? fbytes
                                 : ((e >= 0 && e + kWinPos <= (int)T) ? fbytes + (size_t)(uint32_t)(e + tid) * 32u /* wave-uniform test */
                                                                      : pos_addr(e + tid));
        glds16x2(src, at, src + 16, at + 4096u);
    };
    auto issue_wrap = [&](int q, int slot) { /* last strip, wave 2: the positions around the row's start, 32 B each */
        if (lane < 2 * kWrapPos)
            glds16(pos_addr(est_l[1][clamp_row(q)] - kWrapLead + (lane >> 1)) + 16 * (lane & 1), ring_l + (uint32_t)slot * kSlotBytes + 8192u);
    };
    auto issue_tail_list = [&](int q, int slot) { /* wave 3: the (row, strip)'s list; lanes past its count fetch word 0 again (only the lines that hold entries move) */
        const int qc = clamp_row(q);
        glds4_nt(ftail + (size_t)qc * tail_stride + (lane < (int)tcnt_l[qc] ? lane : 0), tlist_l + (uint32_t)slot * 256u);
    };
    auto issue_tail_points = [&](int q, int slot, int tslot) { /* wave 3: the listed points of row q beside its window, 32 B each */
        const int n = q < N ? (int)tcnt_l[clamp_row(q)] : 0;
        te[tslot] = tlist[tslot][lane];
        const uint32_t ea = tlist[tslot][lane >> 1], eb = tlist[tslot][32 + (lane >> 1)];
        const uint32_t at = ring_l + (uint32_t)slot * kSlotBytes + 8192u + (uint32_t)kWrapPos * 32u;
        glds16x2(fbytes + (size_t)((lane >> 1) < n ? (ea >> 8) : 0u) * 32u + 16 * (lane & 1), at,
                 fbytes + (size_t)(32 + (lane >> 1) < n ? (eb >> 8) : 0u) * 32u + 16 * (lane & 1), at + 1024u);
    };
    /* Row rho's positions -> idx[rho & 1].  Every thread enters ITS window position, counts and checks it: the predecessor
     * in the input must lie in the prefix and have a smaller slot (the lane to the left has it; window position 0 cannot
     * be checked: the estimate was too high).  The first lane of a wave follows a position that ANOTHER wave's DMA brings:
     * that check is made after the step's barrier.  Written without branches: an entry that belongs nowhere goes to the
     * spare word idx[.][256]. */
    bool dneed = false;
    int dflat = 0, dq = 0;
    auto slot_or_max = [&](int q, uint32_t rcw) -> int { /* slot of input position q, INT_MAX outside the prefix / the range image */
        const uint32_t row = rcw & 0xffffu, col = rcw >> 16;
        const bool valid = ((unsigned)q < T) & (row < (uint32_t)N) & (col < (uint32_t)H);
        return valid ? (int)(row * (uint32_t)H + col) : 0x7fffffff;
    };
    auto index_row = [&](int rho, int slot, int tslot) {
        if (rho >= N) return;
        const char *slot_b = &ring[slot * kSlotBytes];
        uint32_t *irow = idx[rho & 1];
        const uint32_t base = (uint32_t)(rho * H + first_col);
        {
            const int q = est_l[0][rho] - kWinLead + tid;
            const u32x4 hi = *reinterpret_cast<const u32x4 *>(slot_b + 4096 + tid * 16); /* (conflict-free; only .y is used) */
            const int sflat = slot_or_max(q, hi.y);
            const uint32_t off = (uint32_t)sflat - base;
            /* (a window of the last strip runs into the next row: those points are not this row's wrap-around halo) */
            atomicMax(&irow[off < (uint32_t)row_span ? off : (uint32_t)kStripThreads], (uint32_t)tid + 1u);
            const bool own = (off - 2u) < (uint32_t)own_cols;
            consumed += own ? 1u : 0u;
            const int pflat = __builtin_amdgcn_update_dpp(0x7fffffff, sflat, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
            const bool chk = own & (q > 0);
            failed |= (chk & ((tid == 0) | ((lane != 0) & !(pflat < sflat)))) ? 1u : 0u;
            /* ... and the window must BRACKET the (row, strip)'s span of slots, halo columns included: the own columns are
             * proven found by the count, the two halo columns on either side are not — a halo point the window misses
             * would read as an empty slot and change phase A's fallbacks (BatchMultiBevGen.cpp:146-154) with nobody
             * noticing.  The prefix is strictly ascending (that is what the checks above prove), so it is enough that the
             * first position's slot is not past the span's first slot (or the window starts at the input's start) and the
             * last position's slot is the span's last or beyond (or the window reaches the prefix's end). */
            const int ibase = rho * H + first_col;
            failed |= (((tid == 0) & (q > 0) & (sflat > ibase)) |
                       ((tid == kWinPos - 1) & (q < (int)T - 1) & (sflat < ibase + row_span - 1))) ? 1u : 0u;
            dneed = chk & (lane == 0) & (tid != 0);
            dflat = sflat;
            dq = q;
        }
        if (last_strip && wv == 2) { /* wave-uniform: slots rho*H and rho*H + 1 as the halo columns H, H + 1 */
            const int k = lane & (kWrapPos - 1);
            const int q = est_l[1][rho] - kWrapLead + k;
            const uint32_t rcw = *reinterpret_cast<const uint32_t *>(slot_b + 8192 + k * 32 + 20);
            const uint32_t row = rcw & 0xffffu, col = rcw >> 16;
            const uint32_t off = (uint32_t)(H - first_col) + col;
            const bool ok = (lane < kWrapPos) & ((unsigned)q < T) & (row == (uint32_t)rho) & (col < 2u) & (off < (uint32_t)kStripVirt);
            atomicMax(&irow[ok ? off : (uint32_t)kStripThreads], (uint32_t)(kWinPos + k) + 1u);
            /* the same bracket for the 16 positions around the row's start: slots rho * H and rho * H + 1 lie inside */
            const int wflat = slot_or_max(q, rcw);
            failed |= (((lane == 0) & (q > 0) & (wflat > rho * H)) |
                       ((lane == kWrapPos - 1) & (q < (int)T - 1) & (wflat < rho * H + 1))) ? 1u : 0u;
        }
        if (wv == 3) { /* later input index beats earlier, any tail point beats the prefix */
            const uint32_t e = te[tslot];
            atomicMax(&irow[lane < (int)tcnt_l[rho] ? (e & 0xffu) : (uint32_t)kStripThreads], kIdxTail | ((e >> 8) << 6) | (uint32_t)lane);
        }
    };
    auto deferred_check = [&](const char *slot_b) { /* after the barrier: every wave's pieces of the row have arrived */
        const uint32_t rcp = *reinterpret_cast<const uint32_t *>(slot_b + 4096 + (tid > 0 ? tid - 1 : 0) * 16 + 4);
        failed |= (dneed && !(slot_or_max(dq - 1, rcp) < dflat)) ? 1u : 0u;
    };

    /* ---- column-major ---- */
    /* the frame's direction and row bases (k_probe), the window of this strip, who counts what */
    __shared__ uint16_t cm_base_l[kCmGen ? kCmMaxRows : 2]; /* (LDS is what holds this source at three workgroups per CU: 53,248 bytes and not one 512-byte granule more) */
    __shared__ uint32_t cm_nr_l[2];     /* no-return firings + 1 this strip owns, rows 2b, 2b + 1 of the band just arrived (LDS atomicMax) */
    __shared__ uint32_t cm_spec_l[2][2]; /* strip 0: [band & 1][row & 1]: the last no-return firing + 1 of the row that another strip owns (0: none) */
    __shared__ uint32_t cm_halo0_l[2];  /* [row & 1]: the index entry that the strip with the wrap-around halo found for virtual column H (= column 0) */
    __shared__ uint16_t cm_win0_l[kCmGen ? kCmMaxRows : 2]; /* strip 0: per row, the firing + 1 whose record it put into column 0 (written out at the end) */
    const bool cm_fwd = kCmGen ? b.cm_par[(size_t)f * kCmParWords] > 0 : true;
    const int cm_bmax = kCmGen ? b.cm_par[(size_t)f * kCmParWords + 1] : 0;
    if constexpr (kCmGen) {
        for (int r = tid; r < N; r += kStripThreads) cm_base_l[r] = (uint16_t)b.cm_par[(size_t)f * kCmParWords + 2 + r];
        if (tid < 2) {
            cm_nr_l[tid] = 0u;
            cm_halo0_l[tid] = 0u;
            cm_spec_l[0][tid] = cm_spec_l[1][tid] = 0u;
        }
    }
    auto mod_h = [&](int x) -> int { /* x mod H for x in (-2 H, 2 H) */
        x = x < 0 ? x + H : x;
        x = x < 0 ? x + H : x;
        return x >= H ? x - H : x;
    };
    auto firing_of = [&](int u) -> int { return cm_fwd ? u : (u ? H - u : 0); }; /* u = +-firing mod H */
    /* What of all this the row loop needs it gets as ONE scalar word of flags and a handful of per-lane values computed here
     * (the first form kept a dozen scalars alive across the loop: 52 spilled scalar registers, the walk 8 % slower). */
    enum : uint32_t { kCfExt = 2u, kCfReports = 4u, kCfListens = 8u, kCfQuiet = 16u, kCfFirst = 32u, kCfBoth = 64u, kCfFlat = 128u, kCfWrap = 256u };
    uint32_t cm_f = 0u;            /* (wave-uniform) */
    int cm_u = 0;                  /* this thread's u = +-firing mod H */
    uint32_t cm_off = 0u, cm_key = 0u, cm_vf = 0u; /* byte offset of its firing's records in the frame; its index key; bit 0 valid, bit 1 counted by this strip */
    uint32_t cm_ext_off = 0u, cm_ext_key = 0u;     /* wave 0: lane = extra firing * 4 + piece: that piece's offset (row 0 of a band); lane < kCmExt: the extra firing's key (0: none) */
    uint32_t cm_side_off[2] = {0u, 0u}, cm_side_key = 0u; /* the side window's wave: the same for its firings (two instructions of 16); lane < 32: a side firing's key */
    /* the plain sweep: this thread's firing */
    const int pl_firing = strip * kStripCols - kColLead + tid;
    const bool pl_valid = (unsigned)pl_firing < (unsigned)H;
    const bool pl_own = (unsigned)(pl_firing - strip * kStripCols) < (unsigned)own_cols; /* counted by this strip */
    if constexpr (kCmGen) {
        const int kind = b.cm_par[(size_t)f * kCmParWords + 3 + kCmMaxRows]; /* 1 a sample was a no-return record, 0 none was */
        const bool first = strip == 0, both = first && last_strip, talk = strips > 1 && kind > 0;
        /* Do this frame's strips talk about no-return records (k_probe saw one)?  If not, a strip other than 0 that owns one
         * after all leaves the row's last in cm_sync and raises kInfoCmStray: k_verdict redoes the frame if it would have won.
         * The kCmExt firings behind the 256 threads' are needed only when the rows' bases lie far apart (staggered beams).
         * (A strip that is the first AND the last of its rows — a sensor of up to 237 columns — holds every firing in its
         * window: its threads enter columns 0, 1 a second time as the wrap-around halo, the side area is the flat-index halo's.) */
        const bool ext = kStripVirt + kColMaxDisp + b.cm_par[(size_t)f * kCmParWords + 2 + kCmMaxRows] > kStripThreads;
        cm_f = ((ext && wv == 0) ? kCfExt : 0u) | ((talk && !first) ? kCfReports : 0u) | ((talk && first && wv == 3) ? kCfListens : 0u) |
               ((strips > 1 && !talk && !first) ? kCfQuiet : 0u) | (first ? kCfFirst : 0u) | (both ? kCfBoth : 0u) |
               ((first && wv == 1) ? kCfFlat : 0u) | ((last_strip && !both && wv == 2) ? kCfWrap : 0u);
        cm_f = __builtin_amdgcn_readfirstlane(cm_f);
        /* this thread's u and firing; a window position past the circle's length repeats an earlier one */
        const int u0 = mod_h((first_col - cm_bmax - kColMaxDisp) % H);
        cm_u = mod_h(u0 + tid % H);
        const int firing = firing_of(cm_u);
        const bool valid = tid < H;
        /* every firing is counted by ONE strip: its window positions own_at .. own_at + own_cols - 1 (the strips' windows start
         * kStripCols apart, so these ranges tile the circle) */
        const int own_at = H >= kStripCols + 16 ? 16 : (H > kStripCols ? H - kStripCols : 0);
        cm_vf = (valid ? 1u : 0u) | (((unsigned)(tid - own_at) < (unsigned)own_cols) ? 2u : 0u);
        cm_off = (uint32_t)(valid ? firing : 0) * (uint32_t)N * 32u;
        cm_key = (((uint32_t)firing + 1u) << kLocBits) | (uint32_t)tid;
        const int i = lane >> 2, piece = lane & 3;
        if (cm_f & kCfExt) {
            const int w = kStripThreads + i, fr = firing_of(mod_h(u0 + w % H));
            cm_ext_off = (uint32_t)(w < H ? fr : 0) * (uint32_t)N * 32u + 16u * (uint32_t)(piece & 1);
            const int wl = kStripThreads + lane, frl = firing_of(mod_h(u0 + wl % H));
            cm_ext_key = (lane < kCmExt && wl < H) ? ((((uint32_t)frl + 1u) << kLocBits) | (kLocExt + (uint32_t)lane)) : 0u;
        }
        if (cm_f & (kCfFlat | kCfWrap)) {
            const int su0 = (cm_f & kCfFlat) ? mod_h((H - 2 - cm_bmax - kColMaxDisp) % H) : mod_h((-cm_bmax - kColMaxDisp) % H);
#pragma unroll
            for (int k0 = 0; k0 < 2; ++k0) {
                const int k = 16 * k0 + i;
                cm_side_off[k0] = (uint32_t)(k < H ? firing_of(mod_h(su0 + k % H)) : 0) * (uint32_t)N * 32u + 16u * (uint32_t)(piece & 1);
            }
            const int kl = lane & (kSideFirings - 1);
            cm_side_key = (lane < kSideFirings && kl < H) ? ((((uint32_t)firing_of(mod_h(su0 + kl % H)) + 1u) << kLocBits) | (kLocSide + (uint32_t)kl)) : 0u;
        }
    }
    const int cm_words_v = in_vgpr((strips - 1) * 2); /* (<= 30: kCmMaxStrips) strip 0 listens to this many words per band (kept in a vector register: see cm_pub_v) */
    /* the frame's words of cm_sync: [band][strip][2], then the per-row words.  (The pointer lives in vector registers: these are
     * rare accesses, and every scalar register kept across the row loop is one more that the loop spills.) */
    const uint64_t cm_pub_v = kCmGen ? in_vgpr((uint64_t)(uintptr_t)(b.cm_sync + (size_t)f * kCmSyncWords)) : 0ull;
    auto cm_pub = [&]() -> gptr<uint32_t> { return (gptr<uint32_t>)(uintptr_t)cm_pub_v; };
    auto cm_buf = [&](int band) -> uint32_t { return (uint32_t)(band & 1) * (uint32_t)kCmBuf; };
    /* rows 2 * band, 2 * band + 1 of this thread's firing: four 16-byte pieces of one 64-byte sector -> piece j at
     * buffer + j * 4 KiB + thread * 16; wave 0: the same of the kCmExt firings behind the window; wave 1 of strip 0: the
     * rows LESS ONE of the firings whose returns can be columns H - 2, H - 1 (slots (r - 1, H - 2), (r - 1, H - 1) are
     * strip 0's virtual columns -2, -1 of row r); wave 2 of a strip with a wrap-around halo: the firings whose returns can
     * be columns 0, 1 (as H, H + 1); lane = firing * 4 + piece */
    auto issue_band = [&](int band) {
        const int r0 = band * kBandRows;
        if (r0 >= N) return; /* (uniform) */
        const uint32_t at = ring_l + cm_buf(band) + (uint32_t)wv * 1024u;
        if constexpr (!kCmGen) { /* the plain sweep: wave 1 of strip 0: the rows LESS ONE of the last kPlainSide firings; wave 2 of the last strip: the first kPlainSide firings */
            const char *src = fbytes + ((size_t)(pl_valid ? pl_firing : 0) * N + r0) * 32u;
            const bool two = r0 + 1 < N;
            glds16x2(src, at, src + 16, at + 4096u);
            glds16x2(src + (two ? 32 : 0), at + 8192u, src + (two ? 48 : 16), at + 12288u);
            if ((strip == 0 && wv == 1) || (last_strip && wv == 2)) {
                const bool flat = wv == 1;
                const int i = lane >> 2, piece = lane & 3;
                const int fr = flat ? H - kPlainSide + i : i;
                int row = r0 + (piece >> 1) - (flat ? 1 : 0);
                const bool ok = (unsigned)fr < (unsigned)H && (unsigned)row < (unsigned)N;
                glds16(fbytes + ((size_t)(ok ? fr : 0) * N + (ok ? row : 0)) * 32u + 16 * (piece & 1),
                       ring_l + cm_buf(band) + (uint32_t)kBandBytes + (flat ? 0u : (uint32_t)(kPlainSide * 32 * kBandRows)));
            }
            return;
        }
        const char *src = fbytes + cm_off + (uint32_t)r0 * 32u;
        /* (N odd or a last band of one row: the second row's pieces come from the next firing or past the frame's end —
         * never used; past the END of the input they would be out of bounds: clamp) */
        const bool two = r0 + 1 < N;
        glds16x2(src, at, src + 16, at + 4096u);
        glds16x2(src + (two ? 32 : 0), at + 8192u, src + (two ? 48 : 16), at + 12288u);
        const int ln = fresh(lane);
        const uint32_t second = ((ln & 2) && two) ? 32u : 0u; /* (piece >> 1: the band's second row) */
        if (cm_f & kCfExt) /* (uniform) the extra firings */
            glds16(fbytes + cm_ext_off + (uint32_t)r0 * 32u + second, ring_l + cm_buf(band) + (uint32_t)kBandBytes);
        if (cm_f & (kCfFlat | kCfWrap)) { /* (uniform) */
            const bool flat = (cm_f & kCfFlat) != 0u;
            /* the flat-index halo wants rows r0 - 1, r0: none before row 0 (that piece fetches row 0 and is not entered) */
            const int row = flat ? r0 - 1 + ((ln & 2) ? 1 : 0) : r0 + (((ln & 2) && two) ? 1 : 0);
            const uint32_t side_at = ring_l + cm_buf(band) + (uint32_t)(kBandBytes + kExtBytes);
            glds16(fbytes + cm_side_off[0] + (uint32_t)(row < 0 ? 0 : row) * 32u, side_at);
            glds16(fbytes + cm_side_off[1] + (uint32_t)(row < 0 ? 0 : row) * 32u, side_at + 16u * 64u);
        }
    };
    /* Strip 0, wave 3: what the other strips have reported for band `band` — the last no-return firing of either row —
     * and the two records themselves into the band buffer's special entry.  The others report when the band ARRIVES in
     * their LDS; strip 0 asks three steps before it uses the band, without waiting (the words come by LDS-DMA and are
     * looked at after the next step's memory wait): once it trails the others by that much it never stalls.  Only when
     * a report is still missing then does it wait for it (bounded), a step before the band is used. */
    __shared__ uint32_t cm_poll_l[2][kCmGen ? 32 : 1];
    auto ask_band = [&](int band) { /* (wave 3) */
        if (band * kBandRows >= N) return; /* (uniform) */
        const int words = __builtin_amdgcn_readfirstlane(cm_words_v);
        if (fresh(lane) < words) glds4_nt((const uint32_t *)(uintptr_t)cm_pub_v + ((size_t)band * kCmMaxStrips + 1) * 2 + lane, __builtin_amdgcn_readfirstlane(lds_addr(&cm_poll_l[band & 1][0])));
    };
    auto take_band = [&](int band, uint32_t w) { /* (wave 3) the reports are in: the larger firing per row, the records */
        const int r0 = band * kBandRows, words = __builtin_amdgcn_readfirstlane(cm_words_v);
        /* even lanes: the band's first row, odd lanes: its second.  (The maxima by v_readlane and scalar compares: as lane
         * shuffles — five LDS round trips on a busy LDS — this cost strip 0 0.7 us at every other step.) */
        uint32_t v0 = 0u, v1 = 0u;
        for (int k = 0; k < words; k += 2) {
            const uint32_t a = (uint32_t)__builtin_amdgcn_readlane((int)w, k) & 0xffffu, c = (uint32_t)__builtin_amdgcn_readlane((int)w, k + 1) & 0xffffu;
            v0 = a > v0 ? a : v0;
            v1 = c > v1 ? c : v1;
        }
        const int ln = fresh(lane);
        if (ln < 2) cm_spec_l[band & 1][ln] = ln ? v1 : v0;
        if ((v0 | v1) != 0u && ln < 4) { /* (uniform test) the records (firing v - 1, row r0 + lane / 2); none: the frame's first record, never entered */
            const uint32_t vv = (ln >> 1) ? v1 : v0;
            const int row = r0 + (ln >> 1);
            const bool ok = vv != 0u && row < N;
            glds16(fbytes + ((size_t)(ok ? vv - 1u : 0u) * N + (ok ? row : 0)) * 32u + 16 * (ln & 1),
                   ring_l + cm_buf(band) + (uint32_t)(kBandBytes + kExtBytes + kSideBytes));
        }
    };
    auto try_band = [&](int band) -> bool { /* (wave 3, after a memory wait) have all the others reported? */
        if (band * kBandRows >= N) return true; /* (uniform) */
        const int ln = fresh(lane), words = __builtin_amdgcn_readfirstlane(cm_words_v);
        const uint32_t w = ln < words ? cm_poll_l[band & 1][ln & 31] : kCmUsedBit;
        if (__ballot((w & kCmUsedBit) == 0u) != 0ull) return false;
        take_band(band, w);
        return true;
    };
    auto listen_band = [&](int band) { /* (wave 3) ... waiting for them — and for those of the band after the next (lanes 32 ..) as
                                        * well: strip 0 then trails the others by the four steps that asking without waiting needs,
                                        * and stays there */
        if (band * kBandRows >= N) return; /* (uniform) */
        const bool more = (band + 2) * kBandRows < N;
        const int words = __builtin_amdgcn_readfirstlane(cm_words_v);
        uint32_t w = 0u, spins = 0u;
        for (;;) {
            const int l = lane & 31;
            w = (l < words && (lane < 32 || more)) ? __hip_atomic_load(cm_pub() + ((size_t)(band + 2 * (lane >> 5)) * kCmMaxStrips + 1) * 2 + l, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                                      : kCmUsedBit;
            if (__ballot((w & kCmUsedBit) == 0u) == 0ull) break;
            if (++spins > kCmSpins) { /* (never seen; the frame is redone the general way) */
                failed |= 1u;
                w = 0u;
                break;
            }
            __builtin_amdgcn_s_sleep(8);
        }
        take_band(band, w);
    };
    int cm_waiting = -1; /* (wave 3 of strip 0) the band whose reports were not all in when asked */
#ifdef BEV_CS_CLOCK
    long long dbg_try_t = 0, dbg_block_t = 0;
    int dbg_fail_n = 0;
#endif
    /* is column `col` of a return of row `row` where firing u's returns of that row lie? */
    auto cm_regular = [&](uint32_t col, int u, int row) -> bool {
        const int d = mod_h((int)col - u - (int)cm_base_l[row]); /* (col < H) */
        return d <= kColMaxDisp;
    };
    /* A band has arrived: the no-return records among the firings this strip owns (column 0, and not where the firing's
     * returns lie), both rows, for strip 0.  (Strip 0 finds its own in its window.) */
    auto report_band = [&](int band) {
        const int r0 = band * kBandRows;
        const char *buf = &ring[cm_buf(band)];
#pragma unroll
        for (int k = 0; k < kBandRows; ++k) {
            const uint32_t rcw = *reinterpret_cast<const uint32_t *>(buf + (k * 2 + 1) * 4096 + tid * 16 + 4);
            const bool zero = cm_vf == 3u && r0 + k < N && rcw == (uint32_t)(r0 + k); /* (valid and counted here) row r0 + k, column 0 */
            if (__ballot(zero) == 0ull) continue; /* (wave-uniform: a sweep without no-return records pays two reads and a compare) */
            if (zero && !cm_regular(0u, cm_u, r0 + k)) atomicMax(&cm_nr_l[k], cm_key >> kLocBits);
        }
    };
    /* Row rho's records -> idx[rho & 1], keyed by (firing + 1) << kLocBits | where the record sits: later firings are
     * later in the input, the larger key wins, as the reference's last writer does (BatchMultiBevGen.cpp:112-115).  Every
     * record this strip OWNS is checked: beam = position mod N, and its column is where its firing's returns lie, or out
     * of range (dropped by the scatter, :109-111), or 0 (a no-return record). */
    auto index_row_cm = [&](int rho) {
        if (rho >= N) return;
        uint32_t *irow = idx[rho & 1];
        const char *buf = &ring[cm_buf(rho / kBandRows)];
        if constexpr (!kCmGen) { /* the plain sweep: column = firing + 0 .. kPlainDisp or out of range; keys are thread numbers (firings ascend with them) */
            {
                const uint32_t rcw = *reinterpret_cast<const uint32_t *>(buf + ((rho & 1) * 2 + 1) * 4096 + tid * 16 + 4);
                const uint32_t row = rcw & 0xffffu, col = rcw >> 16;
                const bool good = (row == (uint32_t)rho) & ((col >= (uint32_t)H) | ((col - (uint32_t)pl_firing) <= (uint32_t)kPlainDisp));
                failed |= (pl_valid & !good) ? 1u : 0u;
                consumed += (pl_valid & pl_own) ? 1u : 0u;
                const uint32_t off = col - (uint32_t)first_col;
                atomicMax(&irow[(pl_valid & (col < (uint32_t)H) & (off < (uint32_t)row_span)) ? off : (uint32_t)kStripThreads], (uint32_t)tid + 1u);
            }
            if ((strip == 0 && wv == 1) || (last_strip && wv == 2)) { /* wave-uniform */
                const bool flat = wv == 1;
                const int i = lane & (kPlainSide - 1);
                const int fr = flat ? H - kPlainSide + i : i;
                const int want_row = flat ? rho - 1 : rho;
                const uint32_t rcw = *reinterpret_cast<const uint32_t *>(buf + kBandBytes + (flat ? 0 : kPlainSide * 32 * kBandRows) + i * 64 + (rho & 1) * 32 + 20);
                const uint32_t row = rcw & 0xffffu, col = rcw >> 16;
                /* flat: columns H - 2, H - 1 of row rho - 1 at offsets 0, 1; wrap: columns 0, 1 of row rho at H - first_col + 0, 1 */
                const uint32_t off = flat ? col - (uint32_t)(H - 2) : (uint32_t)(H - first_col) + col;
                const bool ok = (lane < kPlainSide) & ((unsigned)fr < (unsigned)H) & (want_row >= 0) & (row == (uint32_t)want_row) &
                                (flat ? (col < (uint32_t)H) & (off < 2u) : (col < 2u) & (off < (uint32_t)kStripVirt));
                atomicMax(&irow[ok ? off : (uint32_t)kStripThreads], (uint32_t)(kStripThreads + (flat ? 0 : kPlainSide) + i) + 1u);
            }
            return;
        }
        {
            const int base = (int)cm_base_l[rho]; /* (requested together with the record's word) */
            const uint32_t rcw = *reinterpret_cast<const uint32_t *>(buf + ((rho & 1) * 2 + 1) * 4096 + tid * 16 + 4);
            const uint32_t row = rcw & 0xffffu, col = rcw >> 16;
            const bool good = (row == (uint32_t)rho) && (col >= (uint32_t)H || col == 0u || mod_h((int)col - cm_u - base) <= kColMaxDisp);
            failed |= (cm_vf == 3u && !good) ? 1u : 0u;
            consumed += cm_vf == 3u ? 1u : 0u;
            const uint32_t off = col - (uint32_t)first_col;
            const bool here = (cm_vf & 1u) && row == (uint32_t)rho;
            atomicMax(&irow[(here & (col < (uint32_t)H) & (off < (uint32_t)row_span)) ? off : (uint32_t)kStripThreads], cm_key);
            if (cm_f & kCfBoth) { /* (uniform) columns 0, 1 once more, as the virtual columns H, H + 1 */
                const uint32_t off2 = (uint32_t)(H - first_col) + col;
                atomicMax(&irow[(here & (col < 2u) & (off2 < (uint32_t)kStripVirt)) ? off2 : (uint32_t)kStripThreads], cm_key);
            }
            if ((cm_f & kCfQuiet) && __ballot(cm_vf == 3u && rcw == (uint32_t)rho) != 0ull) { /* (wave-uniform, rare: a record of column 0)
                                                                                             * a no-return record after all, in a frame whose strips do not talk? */
                const bool stray = cm_vf == 3u && rcw == (uint32_t)rho && mod_h(-cm_u - base) > kColMaxDisp;
                if (__ballot(stray) != 0ull) {
                    if (stray) atomicMax((uint32_t *)(uintptr_t)cm_pub_v + kCmPubWords + 2 * kCmMaxRows + rho, cm_key >> kLocBits);
                    failed |= kInfoCmStray;
                }
            }
        }
        const int ln = fresh(lane);
        if ((cm_f & kCfExt) && ln < kCmExt) { /* (uniform per wave) the extra firings: never counted here */
            const uint32_t rcw = *reinterpret_cast<const uint32_t *>(buf + kBandBytes + ln * 64 + (rho & 1) * 32 + 20);
            const uint32_t row = rcw & 0xffffu, col = rcw >> 16;
            const uint32_t off = col - (uint32_t)first_col;
            atomicMax(&irow[((cm_ext_key != 0u) & (row == (uint32_t)rho) & (col < (uint32_t)H) & (off < (uint32_t)row_span)) ? off : (uint32_t)kStripThreads], cm_ext_key);
        }
        if (cm_f & (kCfFlat | kCfWrap)) { /* wave-uniform */
            const bool flat = (cm_f & kCfFlat) != 0u;
            const int e = ln & (kSideFirings - 1); /* entry of the side area */
            const int want_row = flat ? rho - 1 : rho;
            const uint32_t rcw = *reinterpret_cast<const uint32_t *>(buf + kBandBytes + kExtBytes + e * 64 + (rho & 1) * 32 + 20);
            const uint32_t row = rcw & 0xffffu, col = rcw >> 16;
            /* flat: columns H - 2, H - 1 of row rho - 1 at offsets 0, 1; wrap: columns 0, 1 of row rho at H - first_col + 0, 1 */
            const uint32_t off = flat ? col - (uint32_t)(H - 2) : (uint32_t)(H - first_col) + col;
            const bool ok = (ln < kSideFirings) & (cm_side_key != 0u) & (want_row >= 0) & (row == (uint32_t)want_row) &
                            (flat ? (col < (uint32_t)H) & (off < 2u) : (col < 2u) & (off < (uint32_t)kStripVirt));
            atomicMax(&irow[ok ? off : (uint32_t)kStripThreads], cm_side_key);
        }
        if ((cm_f & kCfListens) && ln == 0) { /* the last no-return record of the row that another strip owns: column 0 = offset 2 */
            const uint32_t v = cm_spec_l[(rho / kBandRows) & 1][rho & 1];
            const uint32_t rcw = *reinterpret_cast<const uint32_t *>(buf + kBandBytes + kExtBytes + kSideBytes + (rho & 1) * 32 + 20);
            if (v != 0u) {
                if (rcw != (uint32_t)rho) failed |= 1u; /* (row rho, column 0: what its owner said it was) */
                else atomicMax(&irow[2], (v << kLocBits) | kLocSpecial);
            }
        }
    };

    /* ---- prologue: the queue the row loop expects ---- */
    if constexpr (kColMajor && !kCmGen) {
        issue_band(0);
    } else if constexpr (kCmGen) {
        lds_barrier(); /* the rows' bases */
        issue_band(0);
        if (cm_f & kCfListens) {
            listen_band(0);
            ask_band(1); /* (looked at behind step 0's memory wait) */
        }
    } else if constexpr (kInPlace) {
        if (wv == 3) {
            issue_tail_list(0, 0);
            issue_tail_list(1, 1);
        }
        wait_vm<0>();
        issue_window(0, 0);
        if (last_strip && wv == 2) issue_wrap(0, 0);
        if (wv == 3) {
            issue_tail_points(0, 0, 0);
            issue_tail_list(2, 2);
        }
        issue_window(1, 1);
        if (last_strip && wv == 2) issue_wrap(1, 1);
        if (wv == 3) {
            issue_tail_points(1, 1, 1);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); /* list 0 has been read before its slot is refilled */
            issue_tail_list(3, 0);
        }
    } else {
        issue_winner(0, 0);
        issue_winner(1, 1);
        wait_vm<0>();
        uint32_t r0 = 0u, r1 = 0u;
        if constexpr (kSrc == kSrcGather) {
            r0 = wring[0][tid];
            r1 = wring[1][tid];
        }
        const uint32_t w0 = winner_of(0, r0), w1 = winner_of(1, r1);
        full = (w0 != 0u ? 1u : 0u) | (w1 != 0u ? 2u : 0u);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); /* the words have been read before their ring slots are refilled */
        issue_points(w0, 0);
        issue_winner(2, 2);
        issue_points(w1, 1);
        issue_winner(3, 0);
    }

    WalkRow pr[3] = {};
    PHA_DECL;
#ifdef BEV_CS_CLOCK
    if (lane == 0 && blockIdx.x == 100) printf("walk_prologue %lld (x10 ns)\n", pha_t - tl_t0);
#endif
    float zref = __uint_as_float(0x7fc00000u); /* height of the column's last candidate taken for ground (NaN: none yet) */

    const size_t cand_base = (size_t)f * g.segs * kSeg;
    const gptr<u32x2> fcand = (gptr<u32x2>)(b.cand + cand_base);
    const gptr<uint32_t> fncand = (gptr<uint32_t>)(b.ncand + (size_t)f * g.segs);
    const uint32_t code_last = in_vgpr(g.code_cap - 1u), code_stride = in_vgpr(g.code_stride); /* (they only feed vector instructions) */
    const gptr<uint32_t> flist = (gptr<uint32_t>)(b.code_main + ((size_t)f * g.emitters + strip) * bands * (size_t)g.code_stride);
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
    }
    /* A wave's 64 finished points are 2 KiB of consecutive bytes of the output.  Stored as they sit in the registers — the
     * low halves with one instruction, the high halves with another — every 128-byte line leaves the CU in two
     * instalments and L2 writes some lines back in between (WRITE_SIZE 5.39 MB where 4.9 MB were stored).  Transposed
     * through 2 KiB of LDS each instruction stores 1 KiB of whole lines.  The 2 KiB are a piece of a ring slot that is
     * idle right now and that only this wave's own DMA refills: gather / identity: the slot of the row just consumed
     * (this wave's two 1-KiB pieces, points 0..31 in the first); in place: the same pieces of the slot row r-1 has left. */
    /* (a point's halves swap places in every second group of four points: eight lanes' 16-byte writes at a stride of
     * 32 B then fall into eight different bank quads instead of four — the writes were a two-way conflict) */
    const uint32_t xp_sw = ((uint32_t)lane >> 2) & 1u;
    const uint32_t xp_w = (uint32_t)wv * 1024u + (uint32_t)(lane & 31) * 32u + (lane < 32 ? 0u : 4096u);
    const uint32_t xp_wlo = xp_w + 16u * xp_sw, xp_whi = xp_w + 16u * (xp_sw ^ 1u);
    /* reader lane j wants 16-byte unit j of the KiB = half (j & 1) of point j >> 1 */
    const uint32_t xp_unit = ((uint32_t)lane & ~1u) | (((uint32_t)lane & 1u) ^ (((uint32_t)lane >> 3) & 1u));
    const uint32_t xp_r0 = (uint32_t)wv * 1024u + xp_unit * 16u, xp_r1 = xp_r0 + 4096u; /* first, second KiB */

    /* byte offset of this lane's 16-byte unit in the SECOND KiB of the wave's 64 columns of the ordered cloud's row r - 2
     * (128 units; the first KiB lies 1024 bytes before).  Modulo 2^32 while the row is negative: never used then; from row 0
     * on it is a true offset for every lane (the first strip's first wave starts two columns before the row: its first
     * KiB's first four units do not exist — those lanes do not store — but its second KiB does). */
    uint32_t ord_off = (uint32_t)((-2 * H + strip * kStripCols - 2 + 64 * wv) * 2 + 64 + lane) * 16u;
    const uint32_t row_bytes = (uint32_t)H * 32u;
    auto row_step = [&](auto I, const int r) {
        constexpr int s0 = decltype(I)::value % 3;         /* ring slot of row r (and of row r + 3) */
        constexpr int s2 = (decltype(I)::value + 2) % 3;   /* ... of row r + 2: the slot row r - 1 has left */
        constexpr int s1 = (decltype(I)::value + 1) % 3;   /* winner / list ring: row r + 4 goes where row r + 1's was */
        WalkRow &p0 = pr[s0], &p1 = pr[s2], &p2 = pr[s1];
        const int par = r & 1;
        u32x4 cur_lo, cur_hi;
        uint32_t wraw = 0u;
        PHA(7);
        /* Everything but the newest step's loads has arrived: the points (window) of row r, the winner words (tail list)
         * of row r + 2.  A wave waits for as many operations as it issues loads per step. */
        if constexpr (kColMajor) {
            /* a band's loads are the newest operations but the stores since: they have arrived when nothing is outstanding
             * (the stores of the step before are a step old, as for the other sources) */
            if ((r % kBandRows) == 0) {
                wait_vm<0>();
                if (kCmGen && (cm_f & kCfReports) && r < N) report_band(r / kBandRows);
            }
            PHA(0);
            index_row_cm(r);
            PHA(1);
        } else if constexpr (kInPlace) {
            if (wv == 3) wait_vm<5>();                    /* 2 window pieces, 1 list, 2 tail pieces */
            else if (last_strip && wv == 2) wait_vm<3>(); /* 2 window pieces, the wrap-around positions */
            else wait_vm<2>();
            PHA(0);
            index_row(r, s0, s0);
            PHA(1);
        } else {
            wait_vm<kIdentity ? 2 : 3>();
            PHA(0);
            const char *mine = &ring[s0 * kSlotBytes + tid * 16];
            cur_lo = *reinterpret_cast<const u32x4 *>(mine);
            cur_hi = *reinterpret_cast<const u32x4 *>(mine + 4096);
            if constexpr (kSrc == kSrcGather) wraw = wring[s2][tid];
            if (!((full >> s0) & 1u)) { /* untouched slot: value-initialised, BatchMultiBevGen.cpp:98 */
                cur_lo = u32x4{0u, 0u, 0u, 0u};
                cur_hi = u32x4{0u, 0u, 0u, 0u};
            }
            if constexpr (kStructured) {
                /* the record at flat position r * H + vcol: its slot's point (then the scatter leaves it where it is) or
                 * all-zero (then it lands in slot 0 and its own slot stays value-initialised: all-zero as well); anything
                 * else fails the frame.  Every record is seen by the owner of its column (counted) and by halo threads. */
                const bool rec = (full >> s0) & 1u;
                const uint32_t any = cur_lo.x | cur_lo.y | cur_lo.z | cur_lo.w | cur_hi.x | cur_hi.y | cur_hi.z | cur_hi.w;
                const bool real = cur_hi.y == (((uint32_t)r - st_rowadj) | st_col);
                const bool first = (r == 0) & (vcol == 0); /* flat position 0 */
                failed |= (rec & !real & (any != 0u)) ? kInfoFailed : 0u;
                failed |= (rec & (any == 0u) & !first) ? kInfoZeroSeen : 0u;
                consumed += (rec & outcol) ? 1u : 0u;
                if (first & st_zero_guess) {
                    cur_lo = u32x4{0u, 0u, 0u, 0u};
                    cur_hi = u32x4{0u, 0u, 0u, 0u};
                }
            }
        }
        /* What the waves exchange per step: row r's edge lanes (read by the NEXT step's status) and the per-wave counts of
         * row r-2's candidates (read by this step's write-out).  Gather / identity: published here, before the step's
         * barrier.  In place: the point of row r is known only after the barrier (it makes the index row visible), so
         * both are published at the END of the previous step instead (measured on the gather source, that order costs
         * 7 %: a wave reaches the barrier straight from its memory wait). */
        if constexpr (!kIndexed) {
            if (lane < 2 || lane >= 62)
                edge[r % 3][wv][lane < 2 ? lane : lane - 60] = make_float4(__uint_as_float(cur_lo.x), __uint_as_float(cur_lo.y), __uint_as_float(cur_lo.z), __uint_as_float(cur_hi.x));
            const bool c2 = outcol && wr_gflag(p2.fl) == 1;
            const uint32_t q2 = p2.key & 3u;
            uint32_t rank2;
            const uint32_t scan = quarter_scan(c2, q2, &rank2);
            if (lane == 63) wave_cnt[par][kWaves + wv] = scan;
            p2.fl |= rank2 << kFlRankShift;
        }
        lds_barrier();
        PHA(2);
        if constexpr (kColMajor) {
            /* the column's owner follows its index entry to a record of the band: a thread's, an extra firing's, a side
             * window's, the special one */
            const uint32_t e = idx[par][tid];
            idx[par][tid] = 0u;
            const char *buf = &ring[cm_buf(r / kBandRows)];
            const uint32_t k = kCmGen ? e & ((1u << kLocBits) - 1u) : e - 1u; /* (the plain sweep: thread of the window, or kStripThreads + side firing) */
            const bool main = k < (uint32_t)kStripThreads;
            const uint32_t lo_at = main ? (uint32_t)((r & 1) * 2) * 4096u + k * 16u
                                        : (uint32_t)kBandBytes + (k - (uint32_t)kStripThreads) * 64u + (uint32_t)(r & 1) * 32u;
            const bool have = (e != 0u) & (r < N);
            cur_lo = *(have ? reinterpret_cast<const u32x4 *>(buf + lo_at) : &zero16[0]);
            cur_hi = *(have ? reinterpret_cast<const u32x4 *>(buf + lo_at + (main ? 4096u : 16u)) : &zero16[0]);
            /* Column 0 can hold a no-return record of ANY firing.  Strip 0, which owns the column, hears of the other strips'
             * (listen_band) and says which firing's record it took; a strip whose wrap-around halo shows column 0 as virtual
             * column H sees only the firings of its side window: it remembers what it found there, and if column H - 2 falls
             * back on it (BatchMultiBevGen.cpp:146-149: the upper point's intensity is -1) says so: k_verdict compares. */
            if (kCmGen && r < N) {
                if ((cm_f & kCfFirst) && tid == 2) cm_win0_l[r] = (uint16_t)(e >> kLocBits);
                if (last_strip && v == H) cm_halo0_l[r & 1] = e >> kLocBits;
            }
            /* this strip's no-return records of the band that has just arrived, for strip 0: a word per row */
            if (kCmGen && (cm_f & kCfReports) && (r % kBandRows) == 0 && r < N && tid < kBandRows) {
                const uint32_t nr = cm_nr_l[tid];
                cm_nr_l[tid] = 0u;
                __hip_atomic_store(cm_pub() + ((size_t)(r / kBandRows) * kCmMaxStrips + strip) * 2 + tid, kCmUsedBit | nr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (kCmGen && (cm_f & kCfListens)) { /* (uniform) */
                if ((r % kBandRows) == 0) { /* after this step's memory wait: the reports asked for two steps ago, for the band two steps on;
                                             * and the next band's are asked for (a load asked for at an odd step was waited for half a
                                             * step later, by every wave at the barrier behind: strip 0 18 % slower) */
                    const int band = r / kBandRows + 1;
#ifdef BEV_CS_CLOCK
                    const long long t0_ = wall_clock64();
#endif
                    cm_waiting = try_band(band) ? -1 : band;
#ifdef BEV_CS_CLOCK
                    const long long t1_ = wall_clock64();
                    dbg_block_t += t1_ - t0_;
#endif
                    ask_band(band + 1);
#ifdef BEV_CS_CLOCK
                    dbg_try_t += wall_clock64() - t1_;
                    dbg_fail_n += cm_waiting >= 0 ? 1 : 0;
#endif
                } else if (cm_waiting >= 0) {
#ifdef BEV_CS_CLOCK
                    const long long t0_ = wall_clock64();
#endif
                    listen_band(cm_waiting);
#ifdef BEV_CS_CLOCK
                    dbg_block_t += wall_clock64() - t0_;
#endif
                }
            }
        } else if constexpr (kInPlace) {
            /* the column's owner follows its index entry: a window / wrap-around position, or a tail point; an entry
             * whose (row, col) is not the slot's own is an empty slot (value-initialised, BatchMultiBevGen.cpp:98) */
            if (lane == 0) deferred_check(&ring[s0 * kSlotBytes]);
            const uint32_t e = idx[par][tid];
            idx[par][tid] = 0u; /* (the row after next enters here, two barriers from now) */
            const uint32_t pos = (e & kIdxTail) ? (uint32_t)(kWinPos + kWrapPos) + (e & 63u) : e - 1u;
            const bool inwin = pos < (uint32_t)kWinPos;
            const uint32_t lo_at = inwin ? pos * 16u : 8192u + (pos - (uint32_t)kWinPos) * 32u;
            const char *slot_b = &ring[s0 * kSlotBytes];
            const bool have = (e != 0u) & (r < N);
            /* (an entry leads to a point whose (row, col) ARE this slot's: the offset it was entered at was computed from
             * them; in a frame where that fails — two prefix points of one slot — the order check fails as well) */
            cur_lo = *(have ? reinterpret_cast<const u32x4 *>(slot_b + lo_at) : &zero16[0]);
            cur_hi = *(have ? reinterpret_cast<const u32x4 *>(slot_b + lo_at + (inwin ? 4096u : 16u)) : &zero16[0]);
        }
        const XYZI prev{__uint_as_float(p1.lo.x), __uint_as_float(p1.lo.y), __uint_as_float(p1.lo.z), __uint_as_float(p1.hi.x)};
        const XYZI prevprev{__uint_as_float(p2.lo.x), __uint_as_float(p2.lo.y), __uint_as_float(p2.lo.z), __uint_as_float(p2.hi.x)};
        const XYZI cur{__uint_as_float(cur_lo.x), __uint_as_float(cur_lo.y), __uint_as_float(cur_lo.z), __uint_as_float(cur_hi.x)};

        /* ---- write out row r-2 (first thing after the barrier: its stores are the oldest entries of the step) ---- */
        PHA(3);
        const bool cand2 = outcol && wr_gflag(p2.fl) == 1;
        if (r >= 2) {
            const int q = r - 2;
            const int rr = q - (lo_row - 1);        /* only rows lo-1 .. N-1 can hold candidates */
            if (rr >= 0) {
                /* Candidates of row r-2 by cell quarter (cell mod 4, the low bits of the key): a segment keeps its candidates
                 * as four consecutive runs, one per quarter, each in column order — phase B is four workgroups per frame that
                 * each read one run (cells are independent, only the order inside a cell matters).  A wave's four counts (at
                 * most 64 each; a segment's at most 236 each) travel in one word. */
                const uint32_t q2 = p2.key & 3u;
                static_assert(kWaves == 4, "the four counts are read as one 16-byte word");
                const u32x4 wc = *reinterpret_cast<const u32x4 *>(&wave_cnt[par][kWaves]);
                const uint32_t total = wc.x + wc.y + wc.z + wc.w; /* four byte-wide sums */
                const uint32_t *wb = &wave_cnt[par][wv + 1];
                const uint32_t before = wb[0] + wb[1] + wb[2];
                const uint32_t seg = (uint32_t)(rr * strips + strip);
                if (cand2) {
                    /* where the quarter's run starts (byte q of total * 0x01010100: the quarters below it) + the earlier
                     * waves' candidates of the quarter (no byte overflows: everything stays below the segment's total) +
                     * the earlier lanes' (the rank the scan left in the record) */
                    const uint32_t t8 = total << 8;
                    const uint32_t starts = t8 + (t8 << 8) + (t8 << 16) + before;
                    const uint32_t rank = __builtin_amdgcn_ubfe(starts, q2 << 3, 8u) + ((p2.fl >> kFlRankShift) & 63u);
                    fcand[seg * (uint32_t)kSeg + rank] = u32x2{p2.key, p2.lo.z}; /* key | height */
                }
                if (tid == 2) fncand[seg] = total;
            }
            {   /* BEV code of the slot.  A slot that is not a candidate has its final label, so its code is final too: it
                 * is appended to this strip's list of the raster band its x bin falls into (the order inside a list does
                 * not matter: an LDS cursor per band).  Candidates' codes travel in their keys.  A lane whose left
                 * neighbour appends the very same code skips (near the sensor dozens of consecutive returns share a bin),
                 * and so does one whose code this strip has listed before and still remembers (rings hit the same cells
                 * at the same heights again and again: a HDL_64E frame lists 74 k codes of which 24 k are distinct).  The
                 * rasters are idempotent, so a stale or racing memo entry only costs a duplicate. */
                bool has = outcol && !cand2 && p2.code != kSkip;
                /* (the left neighbour's code and flag by DPP: no LDS round trip) */
                const uint32_t left_code = (uint32_t)__builtin_amdgcn_update_dpp((int)kSkip, (int)p2.code, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
                const bool left_has = __builtin_amdgcn_update_dpp(0, has ? 1 : 0, 0x138, 0xf, 0xf, false) != 0;
                has = has & !((lane > 0) & left_has & (left_code == p2.code));
                /* the memo entry and the band of the code are requested together, then one cursor atomic */
                const uint32_t slot = (p2.code * 0x9E3779B1u) >> (32 - kSeenB);
                const uint32_t remembered = seen[slot];
                const int band = band_tab[code_x(p2.code) & 511];
                has = has & (remembered != p2.code);
                if (has) seen[slot] = p2.code;
                /* (one cursor atomic per wave and band instead of one per code — a ballot loop — measured 2 % slower) */
                if (has) {
                    const uint32_t pos = atomicAdd(&band_cursor[band], 1u);
                    /* (a full list keeps counting and overwrites its last entry: k_bev_raster sees the count) */
                    flist[(uint32_t)band * code_stride + (pos < code_last ? pos : code_last)] = p2.code;
                }
            }
            {   /* the ordered cloud, as whole lines */
                u32x4 hi = p2.hi;
                const bool as_ground = cand2 && !((p2.fl >> 4) & 1u);
                if (as_ground) hi.w &= 0xffff0000u; /* label = 0, BatchMultiBevGen.cpp:245 (provisional) */
                char *xb = &ring[kColMajor ? 2 * kCmBuf : (kInPlace ? s2 : s0) * kSlotBytes];
                *reinterpret_cast<u32x4 *>(xb + xp_wlo) = p2.lo;
                *reinterpret_cast<u32x4 *>(xb + xp_whi) = hi;
                const u32x4 pa = *reinterpret_cast<const u32x4 *>(xb + xp_r0);
                const u32x4 pb = *reinterpret_cast<const u32x4 *>(xb + xp_r1);
                const unsigned long long owners = __ballot(outcol);
                /* (this lane's unit of row q: a byte offset into the frame kept per lane and advanced by one row per step —
                 * base register + 32-bit offset, no 64-bit address arithmetic) */
                const gptr<char> orow = (gptr<char>)fordered + ord_off;
                if ((owners >> (lane >> 1)) & 1ull) __builtin_nontemporal_store(pa, (gptr<u32x4>)(orow - 1024));
                if ((owners >> (32 + (lane >> 1))) & 1ull) __builtin_nontemporal_store(pb, (gptr<u32x4>)orow);
                if (kGm && outcol) fgm[(uint32_t)(q * H + v)] = (int8_t)wr_gflag(p2.fl);
            }
        }
        ord_off += row_bytes;
        /* ---- the loads of this step, behind its stores: row r + 2 (and the winner words / tail list of row r + 4) ---- */
        PHA(4);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); /* this wave is done reading the pieces it refills */
        if constexpr (kColMajor) {
            /* every wave has passed this step's barrier: nobody reads the band before this one any more */
            if ((r % kBandRows) == 0) issue_band(r / kBandRows + 1);
        } else if constexpr (kInPlace) {
            issue_window(r + 2, s2);
            if (last_strip && wv == 2) issue_wrap(r + 2, s2);
            if (wv == 3) {
                issue_tail_points(r + 2, s2, s2);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                issue_tail_list(r + 4, s1);
            }
        } else {
            const uint32_t wn = winner_of(r + 2, wraw);
            full = (full & ~(1u << s2)) | (wn != 0u ? 1u << s2 : 0u);
            issue_points(wn, s2);
            issue_winner(r + 4, s1);
        }

        /* ---- status of row r (BatchMultiBevGen.cpp:142-182) ---- */
        PHA(5);
        int s_r = kSteep;
        if (r >= lo_row && r < N) { /* workgroup-uniform */
            /* row r-1 of the threads two to the right / left */
            XYZI right{from_right2(prev.x), from_right2(prev.y), from_right2(prev.z), from_right2(prev.i)};
            XYZI left{from_left2(prev.x), from_left2(prev.y), from_left2(prev.z), from_left2(prev.i)};
            const float4(*pe)[4] = edge[(r + 2) % 3];
            if (lane >= 62) { /* (the last wave's two have no right neighbour: the value is never used, it is not an output column's) */
                const float4 q = pe[wv + 1 < kWaves ? wv + 1 : wv][lane - 62];
                right = XYZI{q.x, q.y, q.z, q.w};
            }
            if (lane < 2) {
                const float4 q = pe[wv > 0 ? wv - 1 : 0][lane + 2];
                left = XYZI{q.x, q.y, q.z, q.w};
            }
            {   /* (every thread evaluates it: only output columns' statuses are ever used) */
                XYZI up = prev;                                  /* (r-1, c)                  :143     */
                if constexpr (kCmGen) { /* column H - 2 falls back on column 0 of row r - 1: which firing's record this strip took for it */
                    if (last_strip && outcol && v == H - 2 && up.i == -1.0f) {
                        cm_pub()[kCmPubWords + kCmMaxRows + (r - 1)] = kCmUsedBit | cm_halo0_l[(r - 1) & 1];
                        failed |= kInfoCmUsed;
                    }
                }
                if (up.i == -1.0f) up = right;                   /* (r-1, (c+2) % H)          :146-149 */
                if (up.i == -1.0f) up = left;                    /* flat (r-1)*H + c - 2      :151-154 */
                if ((up.i == -1.0f) & (r >= 2)) up = prevprev;   /* (r-2, c)                  :157-160 */
                const bool ground = angle_is_ground_nodiv(up.x - cur.x, up.y - cur.y, up.z - cur.z); /* :169-182 */
                s_r = ((cur.i == -1.0f) | (up.i == -1.0f)) ? kInvalid : (ground ? kGround : kSteep); /* :162-167 */
            }
        }

        /* ---- ground_mat of row r-1 is now decided (closed form, see bev_exact.h) ---- */
        PHA(6);
        int gf = 0;
        {
            const int q = r - 1, st1 = wr_status(p1.fl);
            if (q >= lo_row) gf = (st1 == kInvalid) ? -1 : (st1 == kGround ? 1 : (s_r == kGround ? 1 : 0));
            else if (q == lo_row - 1) gf = (s_r == kGround) ? 1 : 0;
            if (!(q >= 0 && q < N)) gf = 0;
        }
        const bool cand1 = outcol && gf == 1;
        /* Provisional labels.  Phase C un-grounds a candidate that lies 0.30 m above a neighbour cell's average ground
         * height — known only after the whole frame has been summed.  The walk GUESSES: a candidate 0.30 m above the last
         * candidate of its column that it took for ground is written with its own label, every other candidate with
         * label 0; k_ground_resolve tests every candidate exactly and patches the wrong guesses in either direction.
         * The guess only decides how many sparse 2-byte patches are needed (benchmark frames: 1.3 k instead of 7.9 k per
         * frame).  A candidate whose label is not the -2 every producer writes (MulranPointCloudSelect.cpp:126) keeps its
         * label whatever the guess: phase C can then always patch without looking the input point up again (the key says
         * "-2" or the patch is a 0). */
        bool pred1;
        {
            const float zq = __uint_as_float(p1.lo.z);
            const bool plain = (p1.hi.w & 0xffffu) == 0xfffeu;
            pred1 = cand1 && (!plain || zq - zref >= 0.3f); /* (the comparison is false while zref is NaN) */
            if (cand1 && !pred1) zref = zq;
        }
        p1.fl = (p1.fl & 3u) | ((uint32_t)(gf + 1) << 2) | (pred1 ? 16u : 0u);
        if (cand1) {
            int cr, cc;
            const int cell = ground_cell_rc(__uint_as_float(p1.lo.x), __uint_as_float(p1.lo.y), &cr, &cc);
            p1.key = candidate_key_edges(cell, tid - 2, pred1, p1.code, (int)(int16_t)(p1.hi.w & 0xffffu), edge_x[cr], edge_y[cc]);
        }

        /* ---- row r's record (the one row r-3 has left) ---- */
        p0.lo = cur_lo;
        p0.hi = cur_hi;
        p0.fl = (uint32_t)(s_r + 1) | (1u << 2);
        p0.key = 0u;
        p0.code = code_t<kPow2>(cur.x, cur.y, cur.z, (int)(int16_t)(cur_hi.w & 0xffffu), rp);

        /* ---- in place: published for the next step: row r's edge lanes, the candidates of row r-1 per wave ---- */
        if constexpr (kIndexed) {
            if (lane < 2 || lane >= 62) edge[r % 3][wv][lane < 2 ? lane : lane - 60] = make_float4(cur.x, cur.y, cur.z, cur.i);
            const uint32_t q1 = p1.key & 3u;
            uint32_t rank1;
            const uint32_t scan = quarter_scan(cand1, q1, &rank1);
            if (lane == 63) wave_cnt[par ^ 1][kWaves + wv] = scan;
            p1.fl |= rank1 << kFlRankShift;
        }
    };
    /* two extra iterations drain the pipeline */
    for (int r0 = 0; r0 < N + 2; r0 += 3) {
        row_step(std::integral_constant<int, 0>{}, r0);
        if (r0 + 1 < N + 2) row_step(std::integral_constant<int, 1>{}, r0 + 1);
        if (r0 + 2 < N + 2) row_step(std::integral_constant<int, 2>{}, r0 + 2);
    }
    wait_vm<0>(); /* no LDS-DMA may outlive the workgroup's LDS */
    PHA_PRINT(kInPlace ? "walk_inplace vmwait index barrier acquire writeout issue status rest" : "walk_gather vmwait - barrier acquire writeout issue status rest",
              lane == 0 && blockIdx.x == 100);
    PHA_PRINT("walk_cm_strip0 vmwait index barrier acquire writeout issue status rest", kColMajor && lane == 0 && strip == 0 && f == 12);
#ifdef BEV_CS_CLOCK
    if (kColMajor && lane == 0 && wv == 3 && strip == 0 && f == 12) printf("walk_cm_listen try_t %lld block_t %lld fails %d (x10 ns)\n", dbg_try_t, dbg_block_t, dbg_fail_n);
#endif
    PHA_PRINT("walk_cm_strip2 vmwait index barrier acquire writeout issue status rest", kColMajor && lane == 0 && strip == 2 && f == 12);
#ifdef BEV_CS_CLOCK /* where and when the workgroup ran: HW_ID (wave, SIMD, CU, SH, SE), XCC_ID; start and end on the 100 MHz clock */
    if (tid == 0 && (kInPlace || kColMajor) && blockIdx.x < kWalkTlCap) {
        long long *rec = g_walk_tl[blockIdx.x];
        rec[0] = tl_t0;
        rec[1] = wall_clock64();
        rec[2] = (long long)(unsigned)__builtin_amdgcn_s_getreg((31 << 11) | 4);
        rec[3] = (long long)(unsigned)__builtin_amdgcn_s_getreg((31 << 11) | 20);
    }
#endif
    lds_barrier();
    if (tid < bands) b.ncode[((size_t)f * g.emitters + strip) * bands + tid] = band_cursor[tid];
    if constexpr (kCmGen) {
        if (cm_f & kCfFirst)
            for (int r = tid; r < N; r += kStripThreads) cm_pub()[kCmPubWords + r] = (uint32_t)cm_win0_l[r];
    }
    if constexpr (kIndexed || kStructured) {
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            consumed += __shfl_xor(consumed, d);
            failed |= __shfl_xor(failed, d);
        }
        if (lane == 0) {
            atomicAdd(&b.info[f].consumed, consumed);
            if (failed) atomicOr(&b.info[f].failed, failed);
        }
    }
    TL_END(K_GATHER_GROUND);
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
 * Round 3: FOUR workgroups per frame, by cell mod 4 (cells are independent and a cell lies in one quarter, so the order
 * inside a cell is untouched).  The walk keeps every segment's candidates as four consecutive runs, one per quarter, each
 * in column order; workgroup q reads run q of every segment.  A quarter's run of a segment is 35 candidates on average
 * — one 64-slice — so the unit of work is the SLICE: the quarter's slices are numbered in slot order (a prefix sum over
 * the segments' slice counts, once per workgroup), a part is 64 consecutive slices — 16 per wave, in registers — and a
 * quarter walks 8 parts where the one-workgroup form of rounds 1-2 walked 37 (16 segments each): the kernel is a chain
 * of per-part latencies (histogram, scan, placement, sums, five barriers), not of bytes.  39 KB of LDS instead of 99: four
 * workgroups per CU, and room beside the column walk of the other stream.  (Round 2's four-workgroup form kept the
 * 16-segment parts: 37 parts per quarter, 17 % shorter alone and slower in the pipeline; removed, then rebuilt this way.)
 * Per part, everything happens in LDS and registers:
 *   hist    every wave counts its 16 slices' candidates per cell (LDS atomics, two 16-bit counters per word); keys and
 *           heights stay in registers
 *   scan    per-cell totals over the waves, exclusive scan over the touched cells -> the part's runs
 *   place   stable placement into the part's height buffer: lanes of a slice that share a cell rank themselves with
 *           ballots (one per bit of the quarter's cell number, ten: nothing but vector / scalar ALU): constant work
 *           however many distinct cells a slice has
 *   sum     one thread per touched cell continues the cell's running (sum, cnt) through its run of this part
 * while the next part's keys and heights are already in flight, so the only memory round trip that is ever exposed is
 * the first one.  No intermediate of phase B touches HBM. */
constexpr int kCells = kGridCells;
constexpr int kSumQ = 4;                               /* workgroups per frame: cells by cell mod 4 */
constexpr int kSlots = 16;                             /* slices a wave keeps in registers per part */
constexpr int kPartSlices = kSumWaves * kSlots;        /* 64 slices = at most 4096 candidates per part */
struct SumDims {
    static constexpr int cells = (kCells + kSumQ - 1) / kSumQ;
    static constexpr int hist_stride = ((cells + 1) / 2 + 3) / 4 * 4; /* words per wave's histogram: two 16-bit counters per word */
    static constexpr int touch_words = (cells + 31) / 32;
    /* hist, start, zbuf, sumv, cntv, tbits, tlist (u16), misc, then per segment: cpre (u32, T + 1), rs8 (u8, T) */
    static constexpr int start_words = (cells + 3) / 4 * 4; /* (padded: the height buffer behind it is read 16 bytes at a time) */
    static constexpr size_t fixed_words = (size_t)kSumWaves * hist_stride + start_words + (size_t)kPartSlices * 64 + 2 * (size_t)cells + touch_words +
                                          (cells + 1) / 2 + 16;
    static constexpr size_t seg_words(int T) { return (size_t)(T + 4) + (size_t)(T + 3) / 4; } /* cpre: T + 1 entries and three of UINT32_MAX behind them */
    static constexpr size_t lds_bytes(int T) { return sizeof(uint32_t) * (fixed_words + seg_words(T)); } /* HDL_64E (459 segments): 39.5 KB */
};
static_assert(kPartSlices * 64 <= 4096, "a part's run start (12 bits) and length (13 bits) share a word with room to spare");
size_t cell_sums_lds_bytes() { return SumDims::lds_bytes(kMaxSegs); }

__global__ __launch_bounds__(kSumThreads, 4) void k_cell_sums(BatchPtrs b, Geometry g, int nf)
{
    TL_BEGIN;
    using D = SumDims;
    constexpr int kCellsQ = D::cells, kHistStride = D::hist_stride, kTouchWords = D::touch_words;
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    uint32_t *hist = lds;                                 /* [kSumWaves][kHistStride]: 16-bit counts, cells 2i | 2i+1 << 16 */
    uint32_t *start = hist + kSumWaves * kHistStride;     /* [kCellsQ]: the part's runs, start | length << 16 */
    float *zbuf = reinterpret_cast<float *>(start + D::start_words); /* [kPartSlices * 64]: the part's heights by cell (16-byte aligned) */
    float *sumv = zbuf + kPartSlices * 64;                /* [kCellsQ] running sums */
    float *cntv = sumv + kCellsQ;                          /* [kCellsQ] running counts */
    uint32_t *tbits = reinterpret_cast<uint32_t *>(cntv + kCellsQ); /* [kTouchWords]: cells this part has touched */
    uint16_t *tlist = reinterpret_cast<uint16_t *>(tbits + kTouchWords); /* [kCellsQ]: ... listed, in any order */
    uint32_t *misc = reinterpret_cast<uint32_t *>(tlist) + (kCellsQ + 1) / 2; /* [0..1] list lengths (by part parity), [4..7] wave sums, [8] carry */
    const int T = g.segs;
    uint32_t *cpre = misc + 16;                            /* [T + 1]: this quarter's candidates before segment t */
    uint8_t *rs8 = reinterpret_cast<uint8_t *>(cpre + T + 4); /* [T]: where its run starts inside segment t */
    uint16_t *hist16 = reinterpret_cast<uint16_t *>(hist); /* the same counters, cell c of wave w at [w * 2 * kHistStride + c] */

    int f, quarter;
    if (!map_block_xcd(blockIdx.x, nf, kSumQ, f, quarter)) return; /* the quarters of a frame on one XCD: they read the same lines */
    /* ... in a different order from frame to frame.  Measured (scripts/cell_sums_timeline.py): with quarter = position in
     * the frame, a quarter that runs long in every frame (OS1-64 firing order: quarter 3 holds the cells with the longest
     * runs, 103 us per workgroup where the others take 41) ends up four to a CU on every fourth CU — a launch's workgroups go
     * to an XCD's CUs in turn — and the launch's second round of workgroups did not start before THOSE had ended: 55 us
     * with three quarters of the chip idle.  Rotated, every CU holds a mix and the second round starts as the short ones
     * end: 0.43 -> 0.35 us per OS1 frame, config 3 +8 %. */
    static_assert((kSumQ & (kSumQ - 1)) == 0, "the rotation below");
#ifndef BEV_EXP_NO_QROT /* (developer build `make cstl0`: the launch as it was, for scripts/cell_sums_timeline.py) */
    quarter = (quarter + (f >> 3)) & (kSumQ - 1);
#endif
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint2 *ccand = b.cand + (size_t)f * T * kSeg; /* key | height */
    const uint32_t *fn = b.ncand + (size_t)f * T;
    PH_DECL;
    PH();

    for (int k = tid; k < kSumWaves * kHistStride; k += kSumThreads) hist[k] = 0u;
    for (int k = tid; k < kTouchWords; k += kSumThreads) tbits[k] = 0u;
    if (tid < 16) misc[tid] = 0u;
    for (int c = tid; c < kCellsQ; c += kSumThreads) {
        sumv[c] = 0.0f;   /* :133-134 */
        cntv[c] = 0.01f;  /* :135-136 */
    }
    /* this quarter's run of every segment (where it starts inside the segment) and the number of the quarter's candidates
     * before it.  The walk wrote a segment's four counts as four bytes; T <= kMaxSegs = 4 * 256: every thread takes four
     * consecutive segments */
    {
        uint32_t cq[4], mine = 0u;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int t = 4 * tid + k;
            const uint32_t w = t < T ? fn[t] : 0u;
            const uint32_t sh = 8u * (uint32_t)quarter;
            cq[k] = (w >> sh) & 0xffu;
            if (t < T) rs8[t] = (uint8_t)(((w * 0x01010100u) >> sh) & 0xffu); /* the quarters below it (no byte exceeds the segment's 236) */
            mine += cq[k];
        }
        uint32_t incl = mine;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t v = __shfl_up(incl, d);
            if (lane >= d) incl += v;
        }
        if (lane == 63) misc[4 + wv] = incl;
        lds_barrier();
        uint32_t base = incl - mine;
        for (int w = 0; w < wv; ++w) base += misc[4 + w];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int t = 4 * tid + k;
            if (t <= T) cpre[t] = base;
            base += cq[k];
        }
        if (tid == kSumThreads - 1 && 4 * kSumThreads <= T) cpre[T] = base; /* (T == 1024 exactly) */
        if (tid < 3) cpre[T + 1 + tid] = 0xffffffffu; /* (request() looks three segment starts ahead without asking) */
        lds_barrier();
        if (tid == 0) misc[4] = misc[5] = misc[6] = misc[7] = 0u;
    }
    /* The quarter's candidates, segment after segment, are ONE stream in slot order; a slice is 64 consecutive candidates
     * of it — full, whatever the segments' run lengths (a run of a (segment, quarter) is 35 candidates on average: slices
     * cut at segment ends were 55 % full, 1,930 of them per frame where 1,000 do) */
    const int GC = (int)cpre[T];                              /* candidates of this quarter */
    const int G = (GC + 63) >> 6;                             /* slices */
    const int P = (G + kPartSlices - 1) / kPartSlices;

    /* software pipeline: keys + heights one part ahead.  A wave's 16 slices of part p are slices p * 64 + 16 * wave + j */
    uint32_t key_n[kSlots]; /* next part (raw keys; lanes past the slice's count hold garbage) */
    float z_n[kSlots];
    int n_n[kSlots];        /* candidates in the slice (0: no such slice) */
    auto request = [&](int p) {
        /* lane j <= 16 finds the segment of slice g0 + j's first candidate by itself (binary search over the candidate
         * prefix: the searches run side by side); a slice then spans the segments from its own start to the next slice's */
        const int g0 = p * kPartSlices + wv * kSlots;
        const int gl = g0 + (lane < kSlots + 1 ? lane : kSlots);
        int lo = 0;
        if (gl < G) {
            const uint32_t x = 64u * (uint32_t)gl;
            int hi = T - 1; /* largest t with cpre[t] <= x: the segment that holds candidate x (cpre[t + 1] > x) */
            while (lo < hi) {
                const int mid = (lo + hi + 1) >> 1;
                if (cpre[mid] <= x) lo = mid; else hi = mid - 1;
            }
        } else {
            lo = T - 1;
        }
#pragma unroll
        for (int j = 0; j < kSlots; ++j) {
            const bool on = g0 + j < G; /* wave-uniform */
            n_n[j] = on ? (GC - 64 * (g0 + j) < 64 ? GC - 64 * (g0 + j) : 64) : 0;
            key_n[j] = 0u;
            z_n[j] = 0.f;
            if (on) {
                const int t0 = __builtin_amdgcn_readlane(lo, j), t1 = __builtin_amdgcn_readlane(lo, j + 1);
                const uint32_t i = 64u * (uint32_t)(g0 + j) + (uint32_t)lane; /* this lane's candidate (past the end in the last slice) */
                /* the lane's segment: t0 plus the segment starts up to its candidate.  A slice spans two or three
                 * segments: the next three starts are looked at without asking how many there are — a segment past t1
                 * starts after the NEXT slice's first candidate, so past every candidate of this one; more than three
                 * (rare): the loop.  (Rounds 3-4 looped over t0 + 1 ..
                 * t1: a scalar loop per slot with a dependent LDS round trip per turn, 55-60 instructions for a typical
                 * slot; this kernel is short of issue slots.) */
                const uint32_t c1 = cpre[t0 + 1], c2 = cpre[t0 + 2], c3 = cpre[t0 + 3]; /* (t0 < T; UINT32_MAX behind cpre[T]) */
                int t = t0 + (c1 <= i ? 1 : 0) + (c2 <= i ? 1 : 0) + (c3 <= i ? 1 : 0);
                if (t1 - t0 > 3) { /* (wave-uniform) */
#pragma unroll 1
                    for (int u = t0 + 4; u <= t1; ++u) t += cpre[u] <= i ? 1 : 0;
                }
                t = t < T ? t : T - 1; /* (lanes past the stream's end) */
                /* lanes past the stream's end read the last run's stale tail (allocated memory) and are masked where
                 * the values are used */
                const uint32_t at = (uint32_t)t * (uint32_t)kSeg + rs8[t] + (i - cpre[t]);
                const uint2 kz = *reinterpret_cast<const uint2 *>(reinterpret_cast<const char *>(ccand) + 8u * at);
                key_n[j] = kz.x;
                z_n[j] = __uint_as_float(kz.y);
            }
        }
    };
    request(0);
    lds_barrier(); /* LDS state initialised */

    uint32_t *myhist = hist + wv * kHistStride;
    /* THE ORIGIN'S CELL.  A record without a return has x = y = z = 0 — MulRan's no-return records, the all-zero records of a
     * structured cloud, a dropped return of a sweep in firing order — and phase A takes it for ground (angle_is_ground's a == 0 &
     * s == 0 case, :169-182): every one of them is a candidate of the cell that holds the origin, cell 1875, with height 0.  Thousands
     * of them in one cell were one serial chain of additions (OS1-64 firing order with dropped returns: 6,500 of its quarter's 11,500
     * candidates, 54 of its workgroup's 90 us) — additions of ZERO: a running sum is +0 or not a zero at all (it starts at +0, and a
     * sum that cancels is +0 in round-to-nearest), so s + (+-0) = s bit for bit and the reference's chain (:198-199) is the chain of
     * the non-zero heights alone; the count takes its "+ 1" steps in any order (count_advance).  Zero heights of that cell are
     * counted here and never enter the sort: k_cell_sums 0.32 -> 0.23 us per OS1-64 frame. */
    constexpr int kOriginCell = (kGridRows / 2) * kGridCols + kGridCols / 2; /* ground_cell(0, 0): row floor(75 / 2), column floor(50 / 2) */
    static_assert(kOriginCell == 1875, "ground_cell(0.f, 0.f)");
    constexpr uint32_t kOriginIdx = (uint32_t)kOriginCell / kSumQ;
    const bool origin_here = quarter == kOriginCell % kSumQ; /* (workgroup-uniform) */
    uint32_t origin_zeros = 0u;                              /* this lane's zero heights of the cell so far */
    bool origin_look = origin_here;                          /* (wave-uniform) does this wave still look for them? */
    PHA_DECL;
    for (int p = 0; p < P; ++p) {
        PHA(7);
        const int par = p & 1;
        /* part p's data into the "current" registers, part p + 1 requested */
        uint32_t cell[kSlots];
        float zz[kSlots];
        int nn[kSlots];
#pragma unroll
        for (int j = 0; j < kSlots; ++j) {
            nn[j] = n_n[j];
            cell[j] = lane < nn[j] ? ((key_n[j] & kKeyCellMask) >> 2) : 0xfffu; /* 0xfff: no candidate */
            zz[j] = z_n[j];
        }
#ifdef BEV_CS_TL /* (the wait for the part's data apart from the issue of the next part's loads) */
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        PHA(1);
#endif
        request(p + 1);
        if (origin_look) { /* (wave-uniform) */
            uint32_t found = 0u;
#pragma unroll
            for (int j = 0; j < kSlots; ++j) {
                const bool zero = (cell[j] == kOriginIdx) & (zz[j] == 0.0f);
                found += zero ? 1u : 0u;
                cell[j] = zero ? 0xfffu : cell[j]; /* counted; not a candidate of the sort */
            }
            origin_zeros += found;
            /* (leaving a zero IN the sort is as exact as taking it out: a wave whose 1024 candidates of a part held none stops
             * looking — a frame without such records pays for one part's look, not for all) */
            origin_look = __ballot(found != 0u) != 0ull;
        }
        PHA(5);

        /* hist.  Lanes of a 64-slice that hold the same cell find each other with one ballot per key bit (10 bits cover
         * the quarter's 938 cells; 0xfff is not a cell): constant work however many distinct cells the slice has, and
         * nothing but vector / scalar ALU (rounds 1-2 took six ballots, fetched the group leader's cell through the LDS
         * pipe to verify and took the other bits only on a mismatch: a round trip per slice on the critical path; round 5
         * tried a table of lane masks by cell in LDS — OR the lane bit in, read the group back —: 256 entries per wave is
         * what fits, the benchmark's slices hold 64 cells from all over the grid, nearly every slice shared an entry
         * and fell back on the ballots: 0.47 us per frame against 0.385).  Every lane keeps its rank inside its group,
         * the group's size and whether it leads the group in the spare bits of its cell register (cell | rank << 12 |
         * size << 18 | leader << 25), so the placement below needs no second look.  Only leaders touch the histogram (64
         * LDS atomics on one address would serialise) and mark their cell touched — without waiting for an answer: the
         * list of touched cells is made from the marks after the barrier. */
#pragma unroll
        for (int j = 0; j < kSlots; ++j) {
            if (nn[j] == 0) break; /* wave-uniform */
            const uint32_t c = cell[j];
            const bool valid = c != 0xfffu;
            const unsigned long long vb = __ballot(valid);
            /* the lanes that DIFFER from this one in some bit of the cell number: per bit one signed bit-field extract
             * (0 / -1), one compare (the ballot) and, per half of the wave, ONE v_bitop3_b32 (gfx950): d |= ballot ^ mine.
             * 5 vector instructions per bit as compiled (rounds 3-4: selects between the ballot and its complement, 10
             * per bit — two thirds of the kernel's vector instructions, and the kernel is short of VECTOR issue slots,
             * not of latency hiding: 11.4 k vector instructions per wave x 4 waves per SIMD x 4 cycles = its 76 us
             * lifetime; SQ_INSTS_VALU 183 k -> 145 k per frame, 0.44 -> 0.385 us) */
            uint32_t dl = 0u, dh = 0u;
#pragma unroll
            for (int bit = 0; bit < 10; ++bit) {
                int m = __builtin_amdgcn_sbfe((int)c, (uint32_t)bit, 1u);
                asm volatile("" : "+v"(m)); /* (the ballot compares THIS register: left alone the compiler shifts the bit to the sign again for it) */
                const unsigned long long bal = __ballot(m != 0);
                dl = __builtin_amdgcn_bitop3_b32(dl, (uint32_t)bal, (uint32_t)m, 0xF6); /* a | (b ^ c) */
                dh = __builtin_amdgcn_bitop3_b32(dh, (uint32_t)(bal >> 32), (uint32_t)m, 0xF6);
            }
            const uint32_t pl = (uint32_t)vb & ~dl, ph = (uint32_t)(vb >> 32) & ~dh; /* this lane's group */
            const uint32_t size = (uint32_t)__popc(pl) + (uint32_t)__popc(ph);
            const uint32_t rank = __builtin_amdgcn_mbcnt_hi(ph, __builtin_amdgcn_mbcnt_lo(pl, 0u)); /* members in lower lanes */
            const bool leader = valid && rank == 0u;
            if (leader) {
                atomicAdd(&myhist[c >> 1], size << (16 * (c & 1u)));
                atomicOr(&tbits[c >> 5], 1u << (c & 31u));
            }
            if (valid) cell[j] = c | (rank << 12) | (size << 18) | (leader ? 1u << 25 : 0u);
        }
        PHA(6);
        lds_barrier();
        /* the touched cells, listed: the first wave takes one 32-cell word of marks per lane */
        if (wv == 0) {
            uint32_t word = lane < kTouchWords ? tbits[lane] : 0u;
            const uint32_t mine = (uint32_t)__popc(word);
            uint32_t incl = mine;
#pragma unroll
            for (int d = 1; d < 32; d <<= 1) {
                const uint32_t v = __shfl_up(incl, d);
                if (lane >= d) incl += v;
            }
            static_assert(kTouchWords <= 32, "one word of marks per lane of half a wave");
            if (lane == kTouchWords - 1) misc[par] = incl;
            uint32_t at = incl - mine;
            while (word) { /* (at most 32 turns, for the few lanes whose cells are all touched) */
                const uint32_t bit = (uint32_t)__ffs((int)word) - 1u;
                word &= word - 1u;
                tlist[at++] = (uint16_t)(32u * (uint32_t)lane + bit);
            }
        }
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

        /* stable placement: slices in slot order; position = the cell's run start + this wave's cursor inside the run +
         * the lane's rank in its group; the group's leader then advances the cursor (the reads are issued before that
         * update: same wave, program order; two cells of one word may both advance: atomic) */
#pragma unroll
        for (int j = 0; j < kSlots; ++j) {
            if (nn[j] == 0) break; /* wave-uniform */
            const uint32_t v = cell[j];
            const uint32_t c = v & 0xfffu;
            if (c != 0xfffu) {
                const uint32_t off = (myhist[c >> 1] >> (16 * (c & 1u))) & 0xffffu;
                zbuf[(start[c] & 0xffffu) + off + ((v >> 12) & 63u)] = zz[j];
                if (v & (1u << 25)) atomicAdd(&myhist[c >> 1], ((v >> 18) & 127u) << (16 * (c & 1u)));
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
            float sj = sumv[c];
            /* the count of the run in one go (bev_exact.h: the reference's "cnt = cnt + 1" steps are exact inside a
             * binade): the loop below is the sum's chain alone */
            cntv[c] = count_advance(cntv[c], se >> 16); /* :205-206 */
            /* the adds of one cell are a serial chain (that IS the reference's order): what can be saved is everything
             * around the additions.  The heights come as 16-byte reads, THREE reads ahead of their use, into four register
             * quads that take turns (no moves) (rounds 3-4: eight 4-byte reads one group ahead and eight moves per eight
             * additions). */
            for (; q < e && (q & 3); ++q) sj += zbuf[q];
            {
                const float4 *z4 = reinterpret_cast<const float4 *>(zbuf);
                auto add4 = [&](const float4 &v) { sj += v.x; sj += v.y; sj += v.z; sj += v.w; }; /* :198-199 */
                if (q + 12 <= e) {
                    float4 a = z4[q >> 2], bq = z4[(q >> 2) + 1], cq = z4[(q >> 2) + 2];
#pragma unroll 1
                    while (q + 28 <= e) { /* at the top: a, bq, cq = quads q, q + 4, q + 8 */
                        const float4 dq = z4[(q >> 2) + 3];
                        add4(a);
                        a = z4[(q >> 2) + 4];
                        add4(bq);
                        bq = z4[(q >> 2) + 5];
                        add4(cq);
                        cq = z4[(q >> 2) + 6];
                        add4(dq);
                        q += 16;
                    }
                    add4(a);
                    add4(bq);
                    add4(cq);
                    q += 12;
                }
#pragma unroll 1
                for (; q + 4 <= e; q += 4) add4(z4[q >> 2]);
            }
#pragma unroll 1
            for (; q < e; ++q) sj += zbuf[q];
            sumv[c] = sj;
        }
        lds_barrier(); /* the next part overwrites start and zbuf; hist and tbits are clean */
        PHA(4);
    }
    if (origin_here) { /* (workgroup-uniform) the zero heights of the origin's cell: counted, :205-206 */
        if (tid == 0) misc[12] = 0u;
        lds_barrier();
        if (origin_zeros != 0u) atomicAdd(&misc[12], origin_zeros);
        lds_barrier();
        if (tid == 0) cntv[kOriginIdx] = count_advance(cntv[kOriginIdx], misc[12]);
        lds_barrier();
    }
    PHA_PRINT("cell_sums barrier0 - scan place sum request histloop looptop", tid == 0 && blockIdx.x == 100);
    PH();
    float *avg = b.avg + (size_t)f * kCells;
    for (int c = tid; c < kCellsQ; c += kSumThreads)
        if (c * kSumQ + quarter < kCells) avg[c * kSumQ + quarter] = sumv[c] / cntv[c]; /* :210 */
    PH_PRINT("cell_sums all-parts", tid == 0 && blockIdx.x == 100);
    TL_END(K_CELL_SUMS);
#ifdef BEV_CS_TL /* developer build: start / end of every workgroup of this launch in the walk's timeline records */
    if (tid == 0 && f == 12)
        printf("cell_sums frame 12 quarter %d: %d candidates, %d parts; x10 ns: list %lld scan %lld place %lld sum %lld data-wait %lld request %lld hist %lld top %lld\n",
               quarter, GC, P, pha_[0], pha_[2], pha_[3], pha_[4], pha_[1], pha_[5], pha_[6], pha_[7]);
    if (tid == 0 && blockIdx.x < kWalkTlCap) {
        long long *rec = g_walk_tl[blockIdx.x];
        rec[0] = ph_clk[0];
        rec[1] = wall_clock64();
        rec[2] = (long long)(unsigned)__builtin_amdgcn_s_getreg((31 << 11) | 4);
        rec[3] = (long long)(unsigned)__builtin_amdgcn_s_getreg((31 << 11) | 20) | ((long long)GC << 32) | ((long long)quarter << 8);
    }
#endif
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
 * kResolveParts code lists per frame, a contiguous quarter of the segments each; kResolveWgs workgroups per frame (one:
 * the frame's tables — 3,750 averages, their neighbour minima, edge bins, band table — cost as much as a part's
 * candidates) walk kResolveParts / kResolveWgs parts each; a wave requests kResolveBatch segments (x 4 slices of 64
 * candidates) at a time. */
constexpr int kResolveBatch = 4;
/* (round 3: the raster constants in vector registers and stores through address-space-1 pointers, as in the walk — a
 * quarter of this kernel's vector instructions were v_readlane restores of spilled 8-dword argument tuples) */
template <bool kPow2>
__global__ __launch_bounds__(kResolveThreads) void k_ground_resolve(BatchPtrs b, Geometry g)
{
    TL_BEGIN;
    /* Per cell, the LOWEST of its in-range 4-neighbours' averages: "any neighbour n with fl(z - avg[n]) >= 0.3f" is
     * "fl(z - min_n avg[n]) >= 0.3f" — fl(z - a) does not increase with a, and the minimum passes over NaN averages exactly
     * as the comparisons do (a difference with a NaN is never >= 0.3f).  One look-up and one subtraction per candidate
     * instead of four of each with their range tests (bev_exact.h above_neighbour_ground, BatchMultiBevGen.cpp:227-241). */
    __shared__ float minavg[kCells];
    __shared__ float avg[kCells];                        /* the frame's 75 x 50 averages */
    __shared__ int edge_x[kGridRows], edge_y[kGridCols]; /* BEV bin of every ground-grid row's / column's lower edge */
    __shared__ uint32_t band_cursor[kMaxBands];
    __shared__ uint8_t band_tab[512];                    /* x bin -> raster band */
    __shared__ uint16_t cnt[kMaxSegs / kResolveParts + 8];
    constexpr int kPartsPerWg = kResolveParts / kResolveWgs;
    const int f = blockIdx.x / kResolveWgs, part0 = (blockIdx.x - f * kResolveWgs) * kPartsPerWg;
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
                if (!__ballot(hit || wrong)) continue; /* wave-uniform */
                const uint32_t idx = slot0 + ((kk >> kKeyColShift) & 0xffu);
                if (hit && !(kk & kKeyNoCodeBit)) {
                    uint32_t code;
                    if (!candidate_key_escapes(kk)) {
                        code = code_from_bins_t<kPow2>(edge_x[cell / kGridCols] + (int)((kk >> kKeyDxShift) & 3u),
                                                       edge_y[cell % kGridCols] + (int)((kk >> kKeyDyShift) & 3u), z[j][k], rp);
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
                if (wrong) { /* the walk's provisional label differs */
                    /* not un-grounded: label = 0, BatchMultiBevGen.cpp:245; un-grounded: the point's own label back —
                     * which is -2: the walk guesses "stays ground" only for points that carry it */
                    flabel[16u * idx + 14u] = hit ? (uint16_t)(int16_t)-2 : (uint16_t)0;
                }
            }
        }
    }
    lds_barrier();
    if (tid < bands) b.ncode[((size_t)f * g.emitters + g.strips + part) * bands + tid] = band_cursor[tid];
  }
    TL_END(K_GROUND_RESOLVE);
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
        if (M % bands == 0 && (size_t)2 * (M / bands) * M * sizeof(uint32_t) <= (size_t)BEV_RASTER_LDS_CAP) return bands;
    return 0;
}
size_t raster_lds_bytes(const Geometry &g)
{
    return (size_t)2 * g.rp.coarse * g.rp.mat_size * sizeof(uint32_t);
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

__global__ __launch_bounds__(kRasterThreads) void k_bev_raster(BatchPtrs b, Geometry g, int nf, int want_multi, int want_single)
{
    TL_BEGIN;
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    __shared__ uint32_t list_end[kMaxStrips + kResolveParts + 1]; /* inclusive prefix of this band's code-list lengths */
    __shared__ uint32_t over_l;                                    /* a writer had more codes for this band than its list holds */
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
void launch_probe(const Geometry &g, const BatchPtrs &b, int nf, bool allow_stream, hipStream_t st)
{
    if (nf == 0) return;
    hipLaunchKernelGGL(k_probe, dim3(nf), dim3(kProbeThreads), 0, st, b, g, allow_stream ? 1 : 0);
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
