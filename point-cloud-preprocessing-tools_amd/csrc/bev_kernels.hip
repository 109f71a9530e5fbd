/*
 * bev_kernels.hip — hand-written HIP kernels (gfx950, wave64) for the
 * batch_multi_bev_gen hot path.  No MFMA: the path is scatter / stencil /
 * ordered reduction / raster, bounded by HBM.
 *
 * Pipeline for one sub-batch of frames (all launches on one stream):
 *
 *   memset winner
 *   order_scan      per input point : winner[slot] = max(index+1)          (getOrderedCloud, last writer wins)
 *   gather_ground   per slot        : ordered cloud, phase-A ground flag,
 *                                     BEV code, candidate list               (getOrderedCloud + markGroundPoints phase A)
 *   cell_sums       per frame       : stable counting sort of candidates by
 *                                     2 m cell, then IN-ORDER float sums     (markGroundPoints phase B + divide)
 *   ground_resolve  per candidate   : 4-neighbour height test, label fix-up (markGroundPoints phase C)
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

using namespace bevx;

namespace bevk {

static const char *const kNames[K_COUNT] = {
    "k_order_scan", "k_gather_ground", "k_cell_sums", "k_ground_resolve", "k_bev_raster",
    "k_gather_only", "k_ground_mat", "k_cloud_codes", "k_angle_debug",
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

/* ------------------------------------------------------------------------- */
/* getOrderedCloud, BatchMultiBevGen.cpp:102-116: bounds test + slot index;
 * "last point in input order wins" == max input index per slot.            */
__global__ __launch_bounds__(256) void k_order_scan(const bev_point_t *__restrict__ pts,
                                                    const FrameDesc *__restrict__ frames,
                                                    uint32_t *__restrict__ winner, int N, int H, int S)
{
    const int f = blockIdx.y;
    const FrameDesc fd = frames[f];
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= fd.n_pts) return;
    const uint32_t rc = reinterpret_cast<const uint32_t *>(pts + fd.in_offset + i)[5]; /* row | col << 16 */
    const uint32_t row = rc & 0xffffu, col = rc >> 16;
    if (row >= (uint32_t)N || col >= (uint32_t)H) return; /* :106-111 (the "< 0" tests are dead: u16) */
    atomicMax(&winner[(size_t)f * S + row * H + col], i + 1u);
}

/* ------------------------------------------------------------------------- */
struct alignas(16) Half { uint32_t w[4]; };

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

/* kGatherThreads threads x kSlotsPerThread slots = one tile of kTile consecutive
 * slots.  Writes the ordered cloud (label already 0 for candidates, restored
 * later if phase C un-grounds them), each slot's BEV code, the tile's candidate
 * list IN SLOT ORDER and optionally the phase-A ground_mat.  The kSlotsPerThread
 * independent load chains per thread (winner -> point -> stencil neighbours) are
 * what keeps enough requests in flight to stream from HBM. */
template <bool kIdentity>
__global__ __launch_bounds__(kGatherThreads) void k_gather_ground(BatchPtrs b, Geometry g, int nf)
{
    int f, tile;
    if (!map_block_xcd(blockIdx.x, nf, g.tiles, f, tile)) return;
    const int tid = threadIdx.x;
    const size_t fbase = (size_t)f * g.S;

    const bev_point_t *fpts = kIdentity ? (b.pts + fbase) : (b.pts + b.frames[f].in_offset);
    SlotFetch<kIdentity> fetch{b.winner + fbase, fpts};

    Half lo[kSlotsPerThread], hi[kSlotsPerThread];
    int gflag[kSlotsPerThread];
    uint32_t code[kSlotsPerThread];
    bool live[kSlotsPerThread];

    /* issue the self loads of all sub-slots first */
#pragma unroll
    for (int k = 0; k < kSlotsPerThread; ++k) {
        const int slot = tile * kTile + k * kGatherThreads + tid;
        live[k] = slot < g.S;
        lo[k] = Half{{0, 0, 0, 0}};
        hi[k] = Half{{0, 0, 0, 0}};
        if (live[k]) {
            long long src = slot;
            bool have = true;
            if (!kIdentity) {
                const uint32_t w = b.winner[fbase + slot];
                have = (w != 0u);
                src = (long long)w - 1;
            }
            if (have) {
                lo[k] = *reinterpret_cast<const Half *>(fpts + src);
                hi[k] = *(reinterpret_cast<const Half *>(fpts + src) + 1);
            }
        }
    }
#pragma unroll
    for (int k = 0; k < kSlotsPerThread; ++k) {
        const int slot = tile * kTile + k * kGatherThreads + tid;
        gflag[k] = 0;
        if (live[k]) {
            const int row = slot / g.H, col = slot - row * g.H;
            XYZI self{__uint_as_float(lo[k].w[0]), __uint_as_float(lo[k].w[1]), __uint_as_float(lo[k].w[2]),
                      __uint_as_float(hi[k].w[0])};
            gflag[k] = phase_a_ground(row, col, g.N, g.H, g.G, self, fetch);
        }
        code[k] = bev_code(__uint_as_float(lo[k].w[0]), __uint_as_float(lo[k].w[1]), __uint_as_float(lo[k].w[2]),
                           (int)(int16_t)(hi[k].w[3] & 0xffffu), g.rp);
    }

    /* compact the tile's candidates in slot order = (k, tid) order */
    constexpr int kWaves = kGatherThreads / 64;
    __shared__ uint32_t wave_cnt[kSlotsPerThread][kWaves];
    const int lane = tid & 63, wv = tid >> 6;
    unsigned long long m[kSlotsPerThread];
#pragma unroll
    for (int k = 0; k < kSlotsPerThread; ++k) {
        m[k] = __ballot(live[k] && gflag[k] == 1);
        if (lane == 0) wave_cnt[k][wv] = (uint32_t)__popcll(m[k]);
    }
    __syncthreads();
    uint32_t running = 0;
    Candidate *tcand = b.cand + ((size_t)f * g.tiles + tile) * kTile;
#pragma unroll
    for (int k = 0; k < kSlotsPerThread; ++k) {
        uint32_t before = running;
#pragma unroll
        for (int w = 0; w < kWaves; ++w) {
            const uint32_t c = wave_cnt[k][w];
            if (w < wv) before += c;
            running += c;
        }
        const bool is_cand = live[k] && gflag[k] == 1;
        const int slot = tile * kTile + k * kGatherThreads + tid;
        if (is_cand) {
            const uint32_t rank = before + (uint32_t)__popcll(m[k] & ((1ull << lane) - 1ull));
            Candidate c;
            c.slot = (uint32_t)slot;
            c.z = __uint_as_float(lo[k].w[2]);
            c.code = code[k];
            c.cell = (uint16_t)ground_cell(__uint_as_float(lo[k].w[0]), __uint_as_float(lo[k].w[1]));
            c.label = (int16_t)(hi[k].w[3] & 0xffffu);
            tcand[rank] = c;
            hi[k].w[3] &= 0xffff0000u; /* label = 0, BatchMultiBevGen.cpp:245 (provisional) */
        }
        if (live[k]) {
            Half *dst = reinterpret_cast<Half *>(b.ordered + fbase + slot);
            dst[0] = lo[k];
            dst[1] = hi[k];
            b.codes[fbase + slot] = is_cand ? kSkip : code[k];
            if (b.gm) b.gm[fbase + slot] = (int8_t)gflag[k];
        }
    }
    if (tid == 0) b.ncand[(size_t)f * g.tiles + tile] = running;
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
        const uint32_t w = b.winner[fbase + slot];
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
 *        cell_total | per-wave tile counts | scan scratch
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
constexpr int kScanPerThread = (kCells + kSumThreads - 1) / kSumThreads; /* 8 */
constexpr int kChunk = kSumWaves * kCells;  /* floats staged per pass-3 chunk (the dead hist region) */
constexpr int kMaxTilesPerWave = kMaxTiles / kSumWaves + 1;

size_t cell_sums_lds_bytes()
{
    return sizeof(uint32_t) * ((size_t)kSumWaves * kCells + 2 * kCells + (size_t)kSumWaves * kMaxTilesPerWave + 16);
}

__global__ __launch_bounds__(kSumThreads) void k_cell_sums(BatchPtrs b, Geometry g)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    uint32_t *hist = lds;                         /* [kSumWaves][kCells] */
    uint32_t *cell_start = hist + kSumWaves * kCells;
    uint32_t *cell_total = cell_start + kCells;
    uint32_t *tile_cnt = cell_total + kCells;     /* [kSumWaves][kMaxTilesPerWave] */
    uint32_t *wave_sum = tile_cnt + kSumWaves * kMaxTilesPerWave; /* [kSumWaves] */

    const int f = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int T = g.tiles;
    const Candidate *cand = b.cand + (size_t)f * T * kTile;
    const uint32_t *ncand = b.ncand + (size_t)f * T;
    float *zs = b.zsorted + (size_t)f * g.S;

    const int t0 = (int)((long long)T * wv / kSumWaves), t1 = (int)((long long)T * (wv + 1) / kSumWaves);
    uint32_t *mycnt = tile_cnt + wv * kMaxTilesPerWave;
    for (int t = t0 + lane; t < t1; t += 64) mycnt[t - t0] = ncand[t];
    for (int k = tid; k < kSumWaves * kCells; k += kSumThreads) hist[k] = 0u;
    __syncthreads();

    uint32_t *myhist = hist + wv * kCells;

    /* pass 1: order-free histogram of this wave's range */
    for (int t = t0; t < t1; ++t) {
        const int n = (int)mycnt[t - t0];
        const Candidate *tc = cand + (size_t)t * kTile;
        int i = lane;
        for (; i + 192 < n; i += 256) {
            const uint32_t c0 = tc[i].cell, c1 = tc[i + 64].cell, c2 = tc[i + 128].cell, c3 = tc[i + 192].cell;
            atomicAdd(&myhist[c0], 1u);
            atomicAdd(&myhist[c1], 1u);
            atomicAdd(&myhist[c2], 1u);
            atomicAdd(&myhist[c3], 1u);
        }
        for (; i < n; i += 64) atomicAdd(&myhist[tc[i].cell], 1u);
    }
    __syncthreads();

    /* per-cell totals, hist -> wave offsets inside the cell's run */
    for (int c = tid; c < kCells; c += kSumThreads) {
        uint32_t tot = 0;
#pragma unroll
        for (int w = 0; w < kSumWaves; ++w) {
            const uint32_t v = hist[w * kCells + c];
            hist[w * kCells + c] = tot;
            tot += v;
        }
        cell_total[c] = tot;
    }
    __syncthreads();

    /* exclusive scan of cell_total: thread owns kScanPerThread consecutive cells */
    {
        const int c0 = tid * kScanPerThread;
        uint32_t loc[kScanPerThread];
        uint32_t s = 0;
#pragma unroll
        for (int k = 0; k < kScanPerThread; ++k) {
            const int c = c0 + k;
            loc[k] = (c < kCells) ? cell_total[c] : 0u;
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
        for (int k = 0; k < kScanPerThread; ++k) {
            const int c = c0 + k;
            if (c < kCells) cell_start[c] = run;
            run += loc[k];
        }
    }
    __syncthreads();

    /* pass 2: stable placement, one 64-slice at a time, next slice prefetched */
    {
        int t = t0, i0 = 0;
        int n = (t < t1) ? (int)mycnt[0] : 0;
        while (t < t1 && n == 0) { ++t; n = (t < t1) ? (int)mycnt[t - t0] : 0; }
        bool valid = false;
        Candidate cur{};
        if (t < t1) {
            valid = (i0 + lane) < n;
            if (valid) cur = cand[(size_t)t * kTile + i0 + lane];
        }
        while (t < t1) {
            /* locate and prefetch the next slice */
            int nt = t, ni0 = i0 + 64, nn = n;
            if (ni0 >= nn) {
                ni0 = 0;
                do { ++nt; nn = (nt < t1) ? (int)mycnt[nt - t0] : 0; } while (nt < t1 && nn == 0);
            }
            bool nvalid = false;
            Candidate nxt{};
            if (nt < t1) {
                nvalid = (ni0 + lane) < nn;
                if (nvalid) nxt = cand[(size_t)nt * kTile + ni0 + lane];
            }
            /* rank the current slice: lanes holding the same cell find each other with
             * one ballot per key bit (12 bits cover 3750 cells) — constant work however
             * many distinct cells the 64 candidates have */
            const uint32_t cell = valid ? (uint32_t)cur.cell : 0xfffu; /* 4095 is not a cell */
            unsigned long long peers = __ballot(valid);
#pragma unroll
            for (int bit = 0; bit < 12; ++bit) {
                const bool one = (cell >> bit) & 1u;
                const unsigned long long bal = __ballot(one);
                peers &= one ? bal : ~bal;
            }
            uint32_t pos = 0;
            if (valid) {
                const uint32_t below = (uint32_t)__popcll(peers & ((1ull << lane) - 1ull));
                pos = cell_start[cell] + myhist[cell] + below;
            }
            /* the lowest lane of each peer group advances the wave's cursor for that cell;
             * every read above is issued before this write (same wave, program order) */
            if (valid && (peers & ((1ull << lane) - 1ull)) == 0ull) myhist[cell] += (uint32_t)__popcll(peers);
            if (valid) zs[pos] = cur.z;
            t = nt; i0 = ni0; n = nn; valid = nvalid; cur = nxt;
        }
    }
    __threadfence_block();
    __syncthreads();

    /* pass 3: in-order float accumulation; thread owns cells tid + 512*j */
    float sum[kScanPerThread], cnt[kScanPerThread];
#pragma unroll
    for (int j = 0; j < kScanPerThread; ++j) {
        sum[j] = 0.0f;   /* :133-134 */
        cnt[j] = 0.01f;  /* :135-136 */
    }
    const int n_total = (int)(cell_start[kCells - 1] + cell_total[kCells - 1]);
    float *zchunk = reinterpret_cast<float *>(hist);
    for (int chunk0 = 0; chunk0 < n_total; chunk0 += kChunk) {
        const int cn = min(kChunk, n_total - chunk0);
        for (int i = tid; i < cn; i += kSumThreads) zchunk[i] = zs[chunk0 + i];
        __syncthreads();
#pragma unroll
        for (int j = 0; j < kScanPerThread; ++j) {
            const int c = tid + j * kSumThreads;
            if (c < kCells) {
                const int st = (int)cell_start[c];
                const int a = max(st, chunk0) - chunk0;
                const int e = min(st + (int)cell_total[c], chunk0 + cn) - chunk0;
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
    float *avg = b.avg + (size_t)f * kCells;
#pragma unroll
    for (int j = 0; j < kScanPerThread; ++j) {
        const int c = tid + j * kSumThreads;
        if (c < kCells) avg[c] = sum[j] / cnt[j]; /* :210 */
    }
}

/* ------------------------------------------------------------------------- */
/* markGroundPoints phase C for the candidates, BatchMultiBevGen.cpp:216-250.
 * A candidate that is higher than a neighbour cell's average + 0.30 stops
 * being ground: its label is restored and it gets its BEV code back.        */
__global__ __launch_bounds__(kGatherThreads) void k_ground_resolve(BatchPtrs b, Geometry g, int nf)
{
    int f, tile;
    if (!map_block_xcd(blockIdx.x, nf, g.tiles, f, tile)) return;
    const uint32_t n = b.ncand[(size_t)f * g.tiles + tile];
    const float *avg = b.avg + (size_t)f * kCells;
    for (uint32_t i = threadIdx.x; i < n; i += kGatherThreads) {
        const Candidate c = b.cand[((size_t)f * g.tiles + tile) * kTile + i];
        if (above_neighbour_ground(c.z, (int)c.cell, avg)) {
            const size_t idx = (size_t)f * g.S + c.slot;
            reinterpret_cast<int16_t *>(b.ordered + idx)[14] = c.label; /* byte offset 28 */
            b.codes[idx] = c.code;
        }
    }
}

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
size_t raster_lds_bytes(const Geometry &g)
{
    const int M = g.rp.mat_size;
    return (size_t)2 * (M / kRasterSplit) * M * sizeof(uint32_t);
}

__global__ __launch_bounds__(kRasterThreads) void k_bev_raster(const uint32_t *__restrict__ codes, size_t code_stride,
                                                              uint32_t n_codes, uint8_t *__restrict__ multi,
                                                              uint8_t *__restrict__ single, int M, int L)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    const int band_rows = M / kRasterSplit;
    const int cells = band_rows * M;
    uint32_t *mask = lds;
    uint32_t *hmax = lds + cells;
    const int f = blockIdx.x / kRasterSplit, band = blockIdx.x % kRasterSplit;
    const int x0 = band * band_rows;
    const int tid = threadIdx.x;

    for (int k = tid; k < 2 * cells; k += kRasterThreads) lds[k] = 0u;
    __syncthreads();

    const uint32_t *fc = codes + (size_t)f * code_stride;
    for (uint32_t i = tid; i < n_codes; i += kRasterThreads) {
        const uint32_t c = fc[i];
        if (c == kSkip) continue;
        const int x = code_x(c) - x0;
        if (x < 0 || x >= band_rows) continue;
        const int idx = x * M + code_y(c);
        atomicMax(&hmax[idx], (uint32_t)code_h(c));      /* :353-355 */
        const uint32_t l = code_layer(c);
        if (l != kNoLayer) atomicOr(&mask[idx], 1u << l); /* :289-291 */
    }
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
                *reinterpret_cast<uint4 *>(mout + (size_t)l * plane) = make_uint4(w[0], w[1], w[2], w[3]);
            }
        }
    }
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
    dim3 grid((max_pts + 255u) / 256u, (unsigned)nf);
    hipLaunchKernelGGL(k_order_scan, grid, dim3(256), 0, st, b.pts, b.frames, b.winner, g.N, g.H, g.S);
}
void launch_gather_ground(const Geometry &g, const BatchPtrs &b, int nf, bool identity, hipStream_t st)
{
    if (nf == 0) return;
    const int grid = xcd_grid(nf, g.tiles);
    if (identity)
        hipLaunchKernelGGL(k_gather_ground<true>, dim3(grid), dim3(kGatherThreads), 0, st, b, g, nf);
    else
        hipLaunchKernelGGL(k_gather_ground<false>, dim3(grid), dim3(kGatherThreads), 0, st, b, g, nf);
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
    hipLaunchKernelGGL(k_ground_resolve, dim3(xcd_grid(nf, g.tiles)), dim3(kGatherThreads), 0, st, b, g, nf);
}
void launch_bev_raster(const Geometry &g, const uint32_t *codes, size_t code_stride, uint32_t n_codes,
                       uint8_t *multi, uint8_t *single, bool want_multi, bool want_single, int nf,
                       hipStream_t st)
{
    if (nf == 0) return;
    hipLaunchKernelGGL(k_bev_raster, dim3(nf * kRasterSplit), dim3(kRasterThreads), raster_lds_bytes(g), st, codes,
                       code_stride, n_codes, want_multi ? multi : nullptr, want_single ? single : nullptr,
                       g.rp.mat_size, g.rp.n_layers);
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
void launch_angle_debug(const float *dx, const float *dy, const float *dz, uint8_t *out, size_t n, hipStream_t st)
{
    if (n == 0) return;
    hipLaunchKernelGGL(k_angle_debug, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, dx, dy, dz, out, n);
}

} /* namespace bevk */
