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
#include "bev_internal.h"
#include "bev_libm.h"

using namespace bevx;

namespace bevk {

static const char *const kNames[K_COUNT] = {
    "k_order_scan", "k_strip_ground", "k_cell_sums", "k_ground_resolve", "k_bev_raster",
    "k_gather_only", "k_ground_mat", "k_cloud_codes", "k_angle_debug", "k_float_bev", "k_project", "k_transform",
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

/* A winner entry is (tag << shift) | (input index + 1).  The tag is the sub-batch generation of the workspace set:
 * entries left by earlier sub-batches carry a smaller tag, lose every atomicMax against the current one and read as
 * "empty", so the table needs no memset between sub-batches (bev_capi.hip clears it when the tag would wrap). */
__device__ __forceinline__ uint32_t winner_index(uint32_t w, uint32_t tag, int shift)
{
    return (w != 0u && (w >> shift) == tag) ? (w & ((1u << shift) - 1u)) : 0u;
}

/* ------------------------------------------------------------------------- */
/* getOrderedCloud, BatchMultiBevGen.cpp:102-116: bounds test + slot index;
 * "last point in input order wins" == max input index per slot.            */
constexpr int kScanPerThread = 4;
constexpr int kScanIdxBits = 10; /* 256 * kScanPerThread = 1024 points per block */
constexpr int kScanRowBins = 128; /* rows the LDS regrouping below can bin (more rows: plain path) */
__global__ __launch_bounds__(256) void k_order_scan(const bev_point_t *__restrict__ pts,
                                                    const FrameDesc *__restrict__ frames,
                                                    uint32_t *__restrict__ winner, int N, int H, int S,
                                                    uint32_t tag_bits)
{
    const int f = blockIdx.y;
    const FrameDesc fd = frames[f];
    const uint32_t base = blockIdx.x * (256u * kScanPerThread) + threadIdx.x;
    if (blockIdx.x * (256u * kScanPerThread) >= fd.n_pts) return;
    const bev_point_t *fp = pts + fd.in_offset;
    uint32_t slot[kScanPerThread];
    bool spread = false; /* does any wave-instruction's worth of 64 points straddle far-apart slots? */
#pragma unroll
    for (int k = 0; k < kScanPerThread; ++k) { /* all loads in flight before the first atomic */
        const uint32_t i = base + 256u * k;
        slot[k] = 0xffffffffu;
        if (i < fd.n_pts) {
            const uint32_t rc = load_once(reinterpret_cast<const uint32_t *>(fp + i) + 5); /* row | col << 16 */
            const uint32_t row = rc & 0xffffu, col = rc >> 16;
            if (row < (uint32_t)N && col < (uint32_t)H) slot[k] = row * (uint32_t)H + col; /* :106-111 ("< 0" is dead: u16) */
        }
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
};

template <bool kIdentity>
__global__ __launch_bounds__(kStripThreads) void k_strip_ground(BatchPtrs b, Geometry g)
{
    const int f = blockIdx.x / g.strips, strip = blockIdx.x - f * g.strips;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int N = g.N, H = g.H, lo_row = g.N - g.G;
    const size_t frame_off = (size_t)f * g.S;

    const int v = strip * kStripCols + tid - 2;                      /* virtual column */
    const bool provider = (v < H + 2) && (v >= 0 || strip == 0);     /* has a point to load */
    const bool outcol = tid >= 2 && tid < 2 + kStripCols && v < H;   /* owns column v's outputs */
    const int vcol = v >= H ? v - H : v;                             /* wrap; v < 0 keeps the flat rule */

    const bev_point_t *fpts = kIdentity ? (b.pts + frame_off) : (b.pts + b.frames[f].in_offset);
    const uint32_t *fwin = b.winner + frame_off;

    /* Neighbour exchange: lanes l+-2 of the same wave are reached with shuffles; only the two edge
     * lanes on each side of a wave go through LDS (768 B instead of a 12 KiB row buffer, so that these
     * workgroups can share a CU with the 139 KiB cell-sum and 98 KiB raster workgroups of the other lane). */
    constexpr int kWaves = kStripThreads / 64;
    __shared__ float4 edge[3][kWaves][4];                  /* rows r, r-1, (r-2): lanes 0, 1, 62, 63 of every wave */
    __shared__ uint32_t wave_cnt[2][kWaves];               /* per-wave candidate counts of the row being written */

    auto load_winner = [&](int r) -> uint32_t {
        if (!provider || r >= N) return 0u;
        const int fl = r * H + vcol;
        if (fl < 0) return 0u;
        if (kIdentity) return (uint32_t)fl + 1u;
        return winner_index(load_once(&fwin[fl]), b.win_tag, b.win_shift);
    };

    auto load_point = [&](uint32_t w, Half &lo, Half &hi) {
        lo = Half{{0, 0, 0, 0}};
        hi = Half{{0, 0, 0, 0}};
        if (w != 0u) {
            const Half *src = reinterpret_cast<const Half *>(fpts + (w - 1u));
            lo = src[0];
            hi = src[1];
        }
    };

    /* software pipeline: nxt = point of the row about to be processed, w_next = winner of the row after it */
    Half cur_lo, cur_hi, nxt_lo, nxt_hi;
    Half nx2_lo{{0, 0, 0, 0}}, nx2_hi{{0, 0, 0, 0}}; /* point of row r+2 (two rows in flight) */
    uint32_t w_next = 0u, w_nx2 = 0u;                /* winners of rows r+3 and r+4 */
    load_point(load_winner(0), nxt_lo, nxt_hi);
    load_point(load_winner(1), nx2_lo, nx2_hi);
    w_next = load_winner(2);
    w_nx2 = load_winner(3);

    XYZI prev{0.f, 0.f, 0.f, 0.f}, prevprev{0.f, 0.f, 0.f, 0.f};
    PendingRow p1{}, p2{};              /* rows r-1 (ground flag still open) and r-2 (ready to write) */
    float zref = __uint_as_float(0x7fc00000u); /* height of the column's last candidate taken for ground (NaN: none yet) */
    unsigned long long m_ready = 0;     /* candidate ballot of row r-2 */

    const size_t cand_base = (size_t)f * g.segs * kSeg;
    uint32_t *fncand = b.ncand + (size_t)f * g.segs;

    /* two extra iterations drain the pipeline */
    for (int r = 0; r < N + 2; ++r) {
        cur_lo = nxt_lo;
        cur_hi = nxt_hi;
        /* points of rows r+1 and r+2 and winners of rows r+3 and r+4 are in flight while row r is handled */
        nxt_lo = nx2_lo;
        nxt_hi = nx2_hi;
        load_point(r + 2 < N ? w_next : 0u, nx2_lo, nx2_hi);
        w_next = w_nx2;
        w_nx2 = load_winner(r + 4);

        const XYZI cur{__uint_as_float(cur_lo.w[0]), __uint_as_float(cur_lo.w[1]), __uint_as_float(cur_lo.w[2]),
                       __uint_as_float(cur_hi.w[0])};
        if (lane < 2 || lane >= 62) edge[r % 3][wv][lane < 2 ? lane : lane - 60] = make_float4(cur.x, cur.y, cur.z, cur.i);
        if (lane == 0) wave_cnt[r & 1][wv] = (uint32_t)__popcll(m_ready);
        __syncthreads();

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
        const unsigned long long m_new = __ballot(outcol && p1.gflag == 1);
        /* Provisional labels.  Phase C un-grounds a candidate that lies 0.30 m above a neighbour cell's average ground
         * height — known only after the whole frame has been summed.  The walk GUESSES: a candidate 0.30 m above the last
         * candidate of its column that it took for ground is written with its own label and BEV code, every other
         * candidate with label 0 and no code; k_ground_resolve tests every candidate exactly and patches the wrong
         * guesses in either direction.  The guess only decides how many sparse 2- and 4-byte patches are needed
         * (benchmark frames: 1.3 k instead of 7.9 k per frame; without any patch the pipeline would be 8 % faster). */
        {
            const float zq = __uint_as_float(p1.lo.w[2]);
            const bool cand = outcol && p1.gflag == 1;
            p1.pred = cand && (zq - zref >= 0.3f); /* false while zref is NaN */
            if (cand && !p1.pred) zref = zq;
        }

        /* ---- write out row r-2 (its per-wave counts were published before the barrier) ---- */
        if (r >= 2) {
            const int q = r - 2;
            const bool is_cand = outcol && p2.gflag == 1;
            const int rr = q - (lo_row - 1);        /* only rows lo-1 .. N-1 can hold candidates */
            if (rr >= 0) {
                uint32_t before = 0, total = 0;
#pragma unroll
                for (int w = 0; w < kStripThreads / 64; ++w) {
                    const uint32_t c = wave_cnt[r & 1][w];
                    if (w < wv) before += c;
                    total += c;
                }
                const size_t seg = (size_t)rr * g.strips + strip;
                if (is_cand) {
                    const uint32_t rank = before + (uint32_t)__popcll(m_ready & ((1ull << lane) - 1ull));
                    const size_t at = cand_base + seg * kSeg + rank;
                    store_stream(&b.cand_cell[at], (uint16_t)ground_cell(__uint_as_float(p2.lo.w[0]), __uint_as_float(p2.lo.w[1])));
                    store_stream(&b.cand_cellp[at], (uint16_t)((uint32_t)ground_cell(__uint_as_float(p2.lo.w[0]), __uint_as_float(p2.lo.w[1])) |
                                                               (p2.pred ? kCandPredBit : 0u)));
                    store_stream(&b.cand_z[at], __uint_as_float(p2.lo.w[2]));
                    store_stream(&b.cand_aux[at], make_uint2((uint32_t)(tid - 2) | ((p2.hi.w[3] & 0xffffu) << 8), p2.code));
                }
                if (tid == 2) fncand[seg] = total;
            }
            if (outcol) {
                Half hi = p2.hi;
                const bool as_ground = is_cand && !p2.pred;
                if (as_ground) hi.w[3] &= 0xffff0000u; /* label = 0, BatchMultiBevGen.cpp:245 (provisional) */
                const size_t idx = frame_off + (size_t)(q * H + v);
                Half *dst = reinterpret_cast<Half *>(b.ordered + idx);
                store_stream(dst, p2.lo);
                store_stream(dst + 1, hi);
                store_stream(&b.codes[idx], as_ground ? kSkip : p2.code);
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
        p1.code = bev_code(cur.x, cur.y, cur.z, (int)(int16_t)(cur_hi.w[3] & 0xffffu), g.rp);
        prevprev = prev;
        prev = cur;
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
 * One workgroup (8 waves) per frame.
 *   LDS: hist[8][3750] u32 (later reused as the z staging chunk) | cell_start |
 *        per-wave tile counts | scan scratch
 *   pass 1  wave w owns a contiguous range of tiles (slot order) and counts its
 *           candidates per cell (LDS atomics; order-free, so loads are issued
 *           several slices at a time).
 *   scan    hist[w][c] -> offset of wave w inside cell c's run;
 *           cell_start = exclusive scan of the totals.
 *   pass 2  each wave re-walks its range IN ORDER (next slice prefetched);
 *           inside a 64-slice, lanes of the same cell are ranked with ballots,
 *           so the placement into zsorted is a stable sort by cell, i.e. each
 *           cell's run is in row-major slot order.
 *   pass 3  zsorted is staged through LDS in coalesced chunks; one lane per
 *           cell adds its run sequentially in float32 (sum += z; cnt = cnt + 1
 *           from 0.01f) — the reference's accumulation order — then
 *           avg = sum / cnt.                                                  */
constexpr int kCells = kGridCells;
constexpr int kCellsPerThread = (kCells + kSumThreads - 1) / kSumThreads; /* 8 */
constexpr int kChunk = kSumWaves * kCells;  /* floats staged per pass-3 chunk (the dead hist region) */
constexpr int kMaxSegsPerWave = kMaxSegs / kSumWaves + 1;

size_t cell_sums_lds_bytes()
{
    return sizeof(uint32_t) * ((size_t)kSumWaves * kCells + (kCells + 1) + (size_t)kSumWaves * kMaxSegsPerWave + 16);
}

/* developer aid (make clk): phase durations of one workgroup of k_cell_sums, in 10 ns ticks */
#ifdef BEV_CS_CLOCK
#define CS_CLK(i) cs_clk[i] = wall_clock64()
#else
#define CS_CLK(i)
#endif
__global__ __launch_bounds__(kSumThreads) void k_cell_sums(BatchPtrs b, Geometry g)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    uint32_t *hist = lds;                         /* [kSumWaves][kCells] */
    uint32_t *cell_start = hist + kSumWaves * kCells; /* [kCells + 1], the last entry is the candidate count */
    uint32_t *tile_cnt = cell_start + kCells + 1;    /* [kSumWaves][kMaxSegsPerWave] */
    uint32_t *wave_sum = tile_cnt + kSumWaves * kMaxSegsPerWave; /* [kSumWaves] */

    const int f = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int T = g.segs;
#ifdef BEV_CS_CLOCK
    long long cs_clk[8];
#endif
    CS_CLK(0);
    const uint16_t *ccell = b.cand_cell + (size_t)f * T * kSeg;
    const float *cz = b.cand_z + (size_t)f * T * kSeg;
    const uint32_t *ncand = b.ncand + (size_t)f * T;
    float *zs = b.zsorted + (size_t)f * g.S;

    const int t0 = (int)((long long)T * wv / kSumWaves), t1 = (int)((long long)T * (wv + 1) / kSumWaves);
    uint32_t *mycnt = tile_cnt + wv * kMaxSegsPerWave;
    for (int t = t0 + lane; t < t1; t += 64) mycnt[t - t0] = ncand[t];
    for (int k = tid; k < kSumWaves * kCells; k += kSumThreads) hist[k] = 0u;
    __syncthreads();

    uint32_t *myhist = hist + wv * kCells;
    CS_CLK(1);

    /* pass 1: order-free histogram of this wave's range.  The workgroup is alone on its CU (LDS) with two waves per
     * SIMD, and it usually runs beside another sub-batch's streaming kernels, where a memory round trip takes
     * several microseconds: the candidate loads are therefore requested kBatch1 segments (of at most kSeg = 4 x 64
     * candidates) at a time, not one. */
    constexpr int kSl = kSeg / 64;
    constexpr int kBatch1 = 8;
    for (int tb = t0; tb < t1; tb += kBatch1) {
        uint32_t cc[kBatch1][kSl];
#pragma unroll
        for (int j = 0; j < kBatch1; ++j) {
            const int t = tb + j;
            const int n = t < t1 ? (int)mycnt[t - t0] : 0;
#pragma unroll
            for (int k = 0; k < kSl; ++k)
                cc[j][k] = (lane + 64 * k < n) ? (uint32_t)ccell[(size_t)t * kSeg + lane + 64 * k] : 0xffffu;
        }
#pragma unroll
        for (int j = 0; j < kBatch1; ++j)
#pragma unroll
            for (int k = 0; k < kSl; ++k)
                if (cc[j][k] != 0xffffu) atomicAdd(&myhist[cc[j][k]], 1u);
    }
    __syncthreads();
    CS_CLK(2);

    /* per-cell totals, hist -> wave offsets inside the cell's run */
    for (int c = tid; c < kCells; c += kSumThreads) {
        uint32_t tot = 0;
#pragma unroll
        for (int w = 0; w < kSumWaves; ++w) {
            const uint32_t v = hist[w * kCells + c];
            hist[w * kCells + c] = tot;
            tot += v;
        }
        cell_start[c] = tot; /* the cell's total until the scan below turns it into its start */
    }
    __syncthreads();

    /* exclusive scan of the totals: thread owns kCellsPerThread consecutive cells (reads and rewrites only those) */
    {
        const int c0 = tid * kCellsPerThread;
        uint32_t loc[kCellsPerThread];
        uint32_t s = 0;
#pragma unroll
        for (int k = 0; k < kCellsPerThread; ++k) {
            const int c = c0 + k;
            loc[k] = (c < kCells) ? cell_start[c] : 0u;
            s += loc[k];
        }
        uint32_t incl = s;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t v = __shfl_up(incl, d);
            if (lane >= d) incl += v;
        }
        if (lane == 63) wave_sum[wv] = incl;
        __syncthreads();
        uint32_t base = 0;
        for (int w = 0; w < wv; ++w) base += wave_sum[w];
        uint32_t run = base + incl - s;
#pragma unroll
        for (int k = 0; k < kCellsPerThread; ++k) {
            const int c = c0 + k;
            if (c < kCells) cell_start[c] = run;
            run += loc[k];
            if (c == kCells - 1) cell_start[kCells] = run;
        }
    }
    __syncthreads();

    CS_CLK(3);
    /* pass 2: stable placement.  Segments are walked IN ORDER, kBatch2 at a time: all (cell, z) pairs of a batch
     * are requested first, then its 64-slices are ranked one after the other. */
    constexpr int kBatch2 = 4;
    for (int tb = t0; tb < t1; tb += kBatch2) {
        uint32_t cc[kBatch2][kSl];
        float zz[kBatch2][kSl];
        int nn[kBatch2];
#pragma unroll
        for (int j = 0; j < kBatch2; ++j) {
            const int t = tb + j;
            nn[j] = t < t1 ? (int)mycnt[t - t0] : 0;
#pragma unroll
            for (int k = 0; k < kSl; ++k) {
                const bool ok = lane + 64 * k < nn[j];
                cc[j][k] = ok ? (uint32_t)ccell[(size_t)t * kSeg + lane + 64 * k] : 0xfffu;
                zz[j][k] = ok ? cz[(size_t)t * kSeg + lane + 64 * k] : 0.f;
            }
        }
#pragma unroll
        for (int j = 0; j < kBatch2; ++j) {
#pragma unroll
            for (int k = 0; k < kSl; ++k) {
                if (64 * k >= nn[j]) break; /* wave-uniform */
                const bool valid = lane + 64 * k < nn[j];
                /* lanes holding the same cell find each other with one ballot per key bit (12 bits
                 * cover 3750 cells; 0xfff is not a cell): constant work however many distinct cells
                 * the 64 candidates have */
                const uint32_t cell = cc[j][k];
                unsigned long long peers = __ballot(valid);
#pragma unroll
                for (int bit = 0; bit < 12; ++bit) {
                    const bool one = (cell >> bit) & 1u;
                    const unsigned long long bal = __ballot(one);
                    peers &= one ? bal : ~bal;
                }
                const unsigned long long lower = peers & ((1ull << lane) - 1ull);
                if (valid) {
                    const uint32_t pos = cell_start[cell] + myhist[cell] + (uint32_t)__popcll(lower);
                    zs[pos] = zz[j][k];
                }
                /* the lowest lane of each peer group advances the wave's cursor for that cell; every
                 * read above is issued before this write (same wave, program order) */
                if (valid && lower == 0ull) myhist[cell] += (uint32_t)__popcll(peers);
            }
        }
    }
    __threadfence_block();
    __syncthreads();
    CS_CLK(4);

    /* pass 3: in-order float accumulation; thread owns cells tid + 512*j */
    float sum[kCellsPerThread], cnt[kCellsPerThread];
#pragma unroll
    for (int j = 0; j < kCellsPerThread; ++j) {
        sum[j] = 0.0f;   /* :133-134 */
        cnt[j] = 0.01f;  /* :135-136 */
    }
    const int n_total = (int)cell_start[kCells];
    float *zchunk = reinterpret_cast<float *>(hist);
    for (int chunk0 = 0; chunk0 < n_total; chunk0 += kChunk) {
        const int cn = min(kChunk, n_total - chunk0);
        for (int i = tid; i < cn; i += kSumThreads) zchunk[i] = zs[chunk0 + i];
        __syncthreads();
#pragma unroll
        for (int j = 0; j < kCellsPerThread; ++j) {
            const int c = tid + j * kSumThreads;
            if (c < kCells) {
                const int st = (int)cell_start[c];
                const int a = max(st, chunk0) - chunk0;
                const int e = min((int)cell_start[c + 1], chunk0 + cn) - chunk0;
                float sj = sum[j], cj = cnt[j];
                int i = a;
                for (; i + 4 <= e; i += 4) {
                    const float v0 = zchunk[i], v1 = zchunk[i + 1], v2 = zchunk[i + 2], v3 = zchunk[i + 3];
                    sj += v0; sj += v1; sj += v2; sj += v3;       /* :198-199 */
                    cj += 1.0f; cj += 1.0f; cj += 1.0f; cj += 1.0f; /* :205-206 */
                }
                for (; i < e; ++i) {
                    sj += zchunk[i];
                    cj = cj + 1.0f;
                }
                sum[j] = sj;
                cnt[j] = cj;
            }
        }
        __syncthreads();
    }
    CS_CLK(5);
#ifdef BEV_CS_CLOCK
    if (tid == 0 && f == 100)
        printf("cell_sums f=%d: init %lld pass1 %lld scan %lld pass2 %lld pass3 %lld (x10 ns), %d candidates\n", f,
               cs_clk[1] - cs_clk[0], cs_clk[2] - cs_clk[1], cs_clk[3] - cs_clk[2], cs_clk[4] - cs_clk[3],
               cs_clk[5] - cs_clk[4], n_total);
#endif
    float *avg = b.avg + (size_t)f * kCells;
#pragma unroll
    for (int j = 0; j < kCellsPerThread; ++j) {
        const int c = tid + j * kSumThreads;
        if (c < kCells) avg[c] = sum[j] / cnt[j]; /* :210 */
    }
}

/* ------------------------------------------------------------------------- */
/* markGroundPoints phase C for the candidates, BatchMultiBevGen.cpp:216-250.  A candidate that is
 * higher than a neighbour cell's average + 0.30 stops being ground: its label is restored and it
 * gets its BEV code back.  kResolveGroups workgroups per frame, each stages the frame's averages in
 * LDS once and takes every 8th candidate row; all strips' loads of a row are issued before the
 * first test, so a thread has up to 2 * kMaxResolveStrips loads in flight. */
constexpr int kMaxResolveStrips = 12;
constexpr int kResolveGroups = 8; /* workgroups per frame; each takes every 8th candidate row */
__global__ __launch_bounds__(kSeg) void k_ground_resolve(BatchPtrs b, Geometry g)
{
    __shared__ float avg[kCells];      /* the frame's 75 x 50 averages: 4 look-ups per candidate */
    __shared__ uint32_t cnt[kMaxSegs]; /* the frame's candidate counts per (row, strip) segment */
    const int rows = g.G + 1;
    const int f = blockIdx.x / kResolveGroups, grp = blockIdx.x - f * kResolveGroups;
    const int tid = threadIdx.x;
    for (int c = tid; c < kCells; c += kSeg) avg[c] = b.avg[(size_t)f * kCells + c];
    for (int i = tid; i < g.segs; i += kSeg) cnt[i] = b.ncand[(size_t)f * g.segs + i];
    __syncthreads();

    /* The kernel usually runs beside the streaming kernels of the next sub-batch, where a memory round trip takes
     * several microseconds (kernel timeline: 140 us alone, 410 us beside them): a unit (one candidate row x up to 12
     * strips) has its (cell, z) loads issued BEFORE the previous unit is tested, and the aux loads of the previous
     * unit's hits are in flight at the same time, so a workgroup pays about one round trip per unit instead of three
     * (counts, candidates, aux): 320 us beside the streaming kernels. */
    struct Unit {
        uint32_t cell[kMaxResolveStrips];
        float z[kMaxResolveStrips];
        uint32_t ok; /* bit k: strip s0 + k holds a candidate for this thread */
        int rr, s0;
    };
    const int chunks = (g.strips + kMaxResolveStrips - 1) / kMaxResolveStrips;
    const int my_rows = rows > grp ? (rows - grp + kResolveGroups - 1) / kResolveGroups : 0;
    const int n_units = my_rows * chunks;
    auto issue = [&](int u, Unit &q) {
        q.ok = 0u;
        q.rr = 0;
        q.s0 = 0;
        if (u >= n_units) return;
        q.rr = grp + (u / chunks) * kResolveGroups;
        q.s0 = (u % chunks) * kMaxResolveStrips;
        const size_t seg0 = (size_t)f * g.segs + (size_t)q.rr * g.strips;
#pragma unroll
        for (int k = 0; k < kMaxResolveStrips; ++k) {
            const int st = q.s0 + k;
            const bool ok = st < g.strips && (uint32_t)tid < cnt[q.rr * g.strips + (st < g.strips ? st : 0)];
            const size_t at = (seg0 + st) * kSeg + tid;
            q.cell[k] = ok ? (uint32_t)b.cand_cellp[at] : 0u; /* cell | kCandPredBit */
            q.z[k] = ok ? b.cand_z[at] : 0.f;
            q.ok |= ok ? (1u << k) : 0u;
        }
    };
    auto finish = [&](const Unit &q) {
        if (q.ok == 0u) return;
        const size_t seg0 = (size_t)f * g.segs + (size_t)q.rr * g.strips;
        const size_t row_off = (size_t)f * g.S + (size_t)(q.rr + g.N - g.G - 1) * g.H;
        /* hits: un-grounded by phase C.  wrong: the walk's provisional label / code (written for its guess) differ. */
        uint32_t hits = 0u, wrong = 0u;
#pragma unroll
        for (int k = 0; k < kMaxResolveStrips; ++k) {
            if (!((q.ok >> k) & 1u)) continue;
            const bool hit = above_neighbour_ground(q.z[k], (int)(q.cell[k] & kCandCellMask), avg);
            const bool pred = (q.cell[k] & kCandPredBit) != 0u;
            hits |= hit ? (1u << k) : 0u;
            wrong |= (hit != pred) ? (1u << k) : 0u;
        }
        uint2 aux[kMaxResolveStrips];
#pragma unroll
        for (int k = 0; k < kMaxResolveStrips; ++k) /* all aux loads of the unit's wrong guesses in flight together */
            aux[k] = ((wrong >> k) & 1u) ? b.cand_aux[(seg0 + q.s0 + k) * kSeg + tid] : make_uint2(0u, 0u);
#pragma unroll
        for (int k = 0; k < kMaxResolveStrips; ++k) {
            if ((wrong >> k) & 1u) {
                const size_t idx = row_off + (size_t)(q.s0 + k) * kStripCols + (aux[k].x & 0xffu);
                const bool hit = (hits >> k) & 1u;
                /* label @28: the point's own label back, or 0 for ground (BatchMultiBevGen.cpp:245) */
                reinterpret_cast<uint16_t *>(b.ordered + idx)[14] = hit ? (uint16_t)(aux[k].x >> 8) : (uint16_t)0;
                b.codes[idx] = hit ? aux[k].y : kSkip;
            }
        }
    };
    Unit cur, nxt;
    issue(0, cur);
    for (int u = 0; u < n_units; ++u) {
        issue(u + 1, nxt); /* next unit's loads leave before this unit's tests, aux loads and stores */
        finish(cur);
        cur = nxt;
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

/* ------------------------------------------------------------------------- */
/* Rasters, BatchMultiBevGen.cpp:271-292 (occupancy, 24 layers) and :340-356
 * (uint8 max height).  Workgroup = (frame, x-band of M/4 rows).  The band's
 * 24-bit layer masks and max heights live in LDS (2 * 56 * 224 * 4 B = 98 KiB);
 * codes are streamed with coalesced 4 B loads; the planes leave with 16 B
 * stores, 1 KiB contiguous per wave-instruction.                             */
int raster_bands_for(int M)
{
    for (int bands = kRasterSplit; bands <= 16; bands *= 2)
        if (M % bands == 0 && (size_t)2 * (M / bands) * M * sizeof(uint32_t) <= (size_t)150 * 1024) return bands;
    return 0;
}
size_t raster_lds_bytes(const Geometry &g)
{
    const int M = g.rp.mat_size;
    return (size_t)2 * (M / g.raster_bands) * M * sizeof(uint32_t);
}

__global__ __launch_bounds__(kRasterThreads) void k_bev_raster(const uint32_t *__restrict__ codes, size_t code_stride,
                                                              uint32_t n_codes, uint8_t *__restrict__ multi,
                                                              uint8_t *__restrict__ single, int M, int L, int nf,
                                                              int bands)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    const int band_rows = M / bands;
    const int cells = band_rows * M;
    uint32_t *mask = lds;
    uint32_t *hmax = lds + cells;
    /* the bands of a frame read the same codes: give them to ONE XCD (blocks b and
     * b+8 share an L2) and adjacent launch slots, so three of the four reads are L2 hits */
    const int xl = blockIdx.x & 7, jj = blockIdx.x >> 3;
    const int f = (jj / bands) * 8 + xl, band = jj % bands;
    if (f >= nf) return;
    const int x0 = band * band_rows;
    const int tid = threadIdx.x;

    for (int k = tid; k < 2 * cells; k += kRasterThreads) lds[k] = 0u;
    __syncthreads();

    const uint32_t *fc = codes + (size_t)f * code_stride;
    auto splat = [&](uint32_t c) {
        if (c == kSkip) return;
        const int x = code_x(c) - x0;
        if (x < 0 || x >= band_rows) return;
        const int idx = x * M + code_y(c);
        atomicMax(&hmax[idx], (uint32_t)code_h(c));      /* :353-355 */
        const uint32_t l = code_layer(c);
        if (l != kNoLayer) atomicOr(&mask[idx], 1u << l); /* :289-291 */
    };
    /* 16 coalesced loads in flight per thread before the first LDS atomic.  The code scan is what this kernel's time
     * is made of (phase clocks of one workgroup: zero 1 us, scan 20 us, plane stores 7 us), and it is bound by the four
     * bands of a frame each reading all of the frame's codes through L2 at ~27 GB/s per CU — not by the LDS atomics
     * (scan without them: 18.5 us; merging lanes of the same cell with DPP row shifts before the atomics made the
     * kernel 45 % slower). */
    constexpr int kU = 16;
    uint32_t i = tid;
    for (; i + (kU - 1) * kRasterThreads < n_codes; i += kU * kRasterThreads) {
        uint32_t c[kU];
#pragma unroll
        for (int k = 0; k < kU; ++k) c[k] = fc[i + k * kRasterThreads];
#pragma unroll
        for (int k = 0; k < kU; ++k) splat(c[k]);
    }
    for (; i < n_codes; i += kRasterThreads) splat(fc[i]);
    __syncthreads();

    const int chunks_per_row = M / 16;
    const int n_tasks = band_rows * chunks_per_row;
    const size_t plane = (size_t)M * M;
    for (int task = tid; task < n_tasks; task += kRasterThreads) {
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
    return hipFuncSetAttribute(reinterpret_cast<const void *>(k_bev_raster),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)raster_lds_bytes(g));
}
void launch_order_scan(const Geometry &g, const BatchPtrs &b, int nf, uint32_t max_pts, hipStream_t st)
{
    if (max_pts == 0 || nf == 0) return;
    const unsigned per_block = 256u * kScanPerThread;
    dim3 grid((max_pts + per_block - 1u) / per_block, (unsigned)nf);
    hipLaunchKernelGGL(k_order_scan, grid, dim3(256), 0, st, b.pts, b.frames, b.winner, g.N, g.H, g.S,
                       b.win_tag << b.win_shift);
}
void launch_gather_ground(const Geometry &g, const BatchPtrs &b, int nf, bool identity, hipStream_t st)
{
    if (nf == 0) return;
    const int grid = nf * g.strips;
    if (identity)
        hipLaunchKernelGGL(k_strip_ground<true>, dim3(grid), dim3(kStripThreads), 0, st, b, g);
    else
        hipLaunchKernelGGL(k_strip_ground<false>, dim3(grid), dim3(kStripThreads), 0, st, b, g);
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
void launch_ground_resolve(const Geometry &g, const BatchPtrs &b, int nf, hipStream_t st)
{
    if (nf == 0) return;
    hipLaunchKernelGGL(k_ground_resolve, dim3(nf * kResolveGroups), dim3(kSeg), 0, st, b, g);
}
void launch_bev_raster(const Geometry &g, const uint32_t *codes, size_t code_stride, uint32_t n_codes,
                       uint8_t *multi, uint8_t *single, bool want_multi, bool want_single, int nf,
                       hipStream_t st)
{
    if (nf == 0) return;
    hipLaunchKernelGGL(k_bev_raster, dim3(8 * ((nf + 7) / 8) * g.raster_bands), dim3(kRasterThreads), raster_lds_bytes(g),
                       st, codes, code_stride, n_codes, want_multi ? multi : nullptr, want_single ? single : nullptr,
                       g.rp.mat_size, g.rp.n_layers, nf, g.raster_bands);
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
