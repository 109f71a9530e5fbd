/*
 * bev_walk.h — the column walk: getOrderedCloud's gather + markGroundPoints phase A + BEV codes, six sources of the points
 * Part of the device code of libbev_mi355x.so; included by bev_kernels.hip only (one translation unit).
 */
#ifndef BEV_WALK_H
#define BEV_WALK_H

#include "bev_dev.h"

#ifndef BEV_SEENB
#define BEV_SEENB 8
#endif

namespace bevk {
using namespace bevx;

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
constexpr int kColLead = 2 + kPlainDisp, kPlainSide = 16;
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
constexpr uint32_t kCmSpins = 1u << 12; /* polls (a sleep and an agent-scope load each, a microsecond or two) before strip 0 gives up on the others' reports: milliseconds, where a walk workgroup lives a fifth of one */
constexpr int kWinPos = kStripThreads; /* in-place source: window positions of a (row, strip), one per thread: est - kWinLead ... */
constexpr int kWinLead = 12;
constexpr int kWrapPos = 16;       /* ... the last strip's wrap-around halo: positions around the row's start */
constexpr int kWrapLead = 6;
/* bytes of one ring slot: the window's low halves (4 KiB), its high halves (4 KiB), then, 32 B each, the wrap-around
 * positions and the tail points */
constexpr int kInPlaceSlot = (kWinPos + kWrapPos + kTailCap) * 32;
constexpr uint32_t kIdxTail = 1u << 30;

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

/* The walk's LDS, one struct per source so that the same bytes can be another kernel body's in a fused launch (k_stage):
 * every body carves its arrays out of ONE arena; a workgroup runs one body.  Members a source does not use are one element. */
template <int kSrc>
struct WalkLds {
    static constexpr bool kInPlace = kSrc == kSrcInPlace, kCmGen = kSrc == kSrcColMajorGen, kColMajor = kSrc == kSrcColMajor || kCmGen,
                          kIndexed = kInPlace || kColMajor;
    static constexpr int kCmBuf = kCmGen ? kColBuf : kPlainBuf; /* bytes of one band buffer */
    static constexpr int kWaves = kStripThreads / 64;
    static constexpr int kSlotBytes = kInPlace ? kInPlaceSlot : 8192;
    static constexpr int kSeenB = kIndexed ? BEV_SEENB : kSeenBits; /* (the in-place source needs the LDS for its windows) */
    /* the points of rows r, r+1, r+2.  Gather / identity: by thread, low halves in the first 4 KiB, high halves in the
     * second.  In place: by window position, 32 B each, then the wrap-around positions, then the tail points */
    /* column-major: two band buffers, then 8 KiB for the write-out's transposition */
    alignas(16) char ring[kColMajor ? 2 * kCmBuf + 8192 : 3 * kSlotBytes];
    alignas(16) u32x4 zero16[1];                              /* what an empty slot reads */
    alignas(16) float4 edge[3][kWaves][4];                    /* rows r, r-1, (r-2): lanes 0, 1, 62, 63 of every wave */
    /* per-wave candidate counts of the row being written, at [.][kWaves + wave] behind kWaves words that stay zero: the
     * three words before a wave's own are the counts of the waves before it, whichever wave it is (no selects) */
    alignas(16) uint32_t wave_cnt[2][2 * kWaves];
    uint32_t wring[kSrc == kSrcGather ? 3 : 1][kSrc == kSrcGather ? kStripThreads : 1]; /* raw winner words of rows r+2, r+3, r+4 */
    uint32_t idx[kIndexed ? 2 : 1][kIndexed ? kStripThreads + 1 : 1]; /* column offset -> position + 1 | tail key ([256]: nowhere) */
    uint32_t tlist[kInPlace ? 3 : 1][kInPlace ? 64 : 1];      /* tail lists of rows r+2, r+3, r+4 */
    int est_l[2][kInPlace ? kStreamMaxRows : 1];
    uint32_t band_cursor[kMaxBands];                          /* entries already in this strip's code list of each band */
    uint32_t seen[1 << kSeenB];                               /* direct-mapped memo of codes this strip has already listed */
    int edge_x[kGridRows], edge_y[kGridCols];                 /* BEV bin of every ground-grid row's / column's lower edge */
    /* column-major, general form: the frame's direction and row bases (k_probe), the window of this strip, who counts what */
    uint32_t cm_nr_l[2];     /* no-return firings + 1 this strip owns, rows 2b, 2b + 1 of the band just arrived (LDS atomicMax) */
    uint32_t cm_spec_l[2][2]; /* strip 0: [band & 1][row & 1]: the last no-return firing + 1 of the row that another strip owns (0: none) */
    uint32_t cm_halo0_l[2];  /* [row & 1]: the index entry that the strip with the wrap-around halo found for virtual column H (= column 0) */
    uint32_t cm_poll_l[2][kCmGen ? 32 : 1];
    uint16_t cm_base_l[kCmGen ? kCmMaxRows : 2]; /* (LDS is what holds this source at three workgroups per CU: 42 allocation granules of 1,280 bytes and not one more) */
    uint16_t cm_win0_l[kCmGen ? kCmMaxRows : 2]; /* strip 0: per row, the firing + 1 whose record it put into column 0 (written out at the end) */
    uint8_t band_tab[512];                                    /* x bin -> raster band */
    uint8_t tcnt_l[kInPlace ? kStreamMaxRows : 1];
};
static_assert(sizeof(WalkLds<kSrcInPlace>) <= 32 * 1280 && sizeof(WalkLds<kSrcGather>) <= 32 * 1280 && sizeof(WalkLds<kSrcStructured>) <= 32 * 1280,
              "four column-walk workgroups per CU: 32 of the CU's 128 LDS granules (1,280 bytes) each");
static_assert(sizeof(WalkLds<kSrcColMajorGen>) <= 42 * 1280 && sizeof(WalkLds<kSrcColMajor>) <= 42 * 1280, "three firing-order workgroups per CU");

/* the walk of workgroup `bid` of a launch over nf frames (k_walk: a launch of its own; k_stage: beside the other stages) */
template <int kSrc, bool kPow2, bool kGm>
__device__ __forceinline__ void walk_body(char *arena, const BatchPtrs &b, const Geometry &g, const int f, int strip, uint32_t want_mode, int bid /* developer builds: which workgroup prints */)
{
    TL_BEGIN;
    /* kStructured: the identity source over the caller's INPUT (record i = slot i's point or an all-zero record), every
     * record checked; kIdentity below covers both (no winner table, position = slot) */
    constexpr bool kStructured = kSrc == kSrcStructured, kIdentity = kSrc == kSrcIdentity || kStructured, kInPlace = kSrc == kSrcInPlace;
    /* kIndexed: the sources whose points reach their columns through an index row (LDS atomicMax), after the step's barrier */
    constexpr bool kCmGen = kSrc == kSrcColMajorGen, kColMajor = kSrc == kSrcColMajor || kCmGen, kIndexed = kInPlace || kColMajor;
    using Lds = WalkLds<kSrc>;
    constexpr int kCmBuf = Lds::kCmBuf;
    static_assert(kWinPos == 256 && kStripVirt + 16 <= kWinPos && kTailCap == 64 && kWrapPos == 16, "DMA pieces of the in-place source");
#ifdef BEV_CS_CLOCK
    const long long tl_t0 = wall_clock64();
#endif
    (void)bid;
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

    constexpr int kWaves = Lds::kWaves, kSlotBytes = Lds::kSlotBytes, kSeenB = Lds::kSeenB;
    Lds &lds_w = *reinterpret_cast<Lds *>(arena);
    auto &ring = lds_w.ring;
    auto &wring = lds_w.wring;
    auto &idx = lds_w.idx;
    auto &zero16 = lds_w.zero16;
    auto &tlist = lds_w.tlist;
    auto &est_l = lds_w.est_l;
    auto &tcnt_l = lds_w.tcnt_l;
    auto &edge = lds_w.edge;
    auto &wave_cnt = lds_w.wave_cnt;
    auto &band_cursor = lds_w.band_cursor;
    auto &band_tab = lds_w.band_tab;
    auto &seen = lds_w.seen;
    auto &edge_x = lds_w.edge_x;
    auto &edge_y = lds_w.edge_y;
    auto &cm_base_l = lds_w.cm_base_l;
    auto &cm_nr_l = lds_w.cm_nr_l;
    auto &cm_spec_l = lds_w.cm_spec_l;
    auto &cm_halo0_l = lds_w.cm_halo0_l;
    auto &cm_win0_l = lds_w.cm_win0_l;
    auto &cm_poll_l = lds_w.cm_poll_l;
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
            if (++spins > kCmSpins) { /* (never seen; the frame is redone the general way) and strip 0 stops listening: one
                                       * bounded wait per frame, not one per band (advisor, round 5) */
                failed |= 1u;
                cm_f &= ~(uint32_t)kCfListens;
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
    if (lane == 0 && bid == 100) printf("walk_prologue %lld (x10 ns)\n", pha_t - tl_t0);
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
                (void)nr;
#ifndef BEV_EXP_NO_REPORTS /* (developer build, scripts/cm_timeout_check.py: the reports never arrive — strip 0 must give up, once, and the frame be redone) */
                __hip_atomic_store(cm_pub() + ((size_t)(r / kBandRows) * kCmMaxStrips + strip) * 2 + tid, kCmUsedBit | nr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
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
              lane == 0 && bid == 100);
    PHA_PRINT("walk_cm_strip0 vmwait index barrier acquire writeout issue status rest", kColMajor && lane == 0 && strip == 0 && f == 12);
#ifdef BEV_CS_CLOCK
    if (kColMajor && lane == 0 && wv == 3 && strip == 0 && f == 12) printf("walk_cm_listen try_t %lld block_t %lld fails %d (x10 ns)\n", dbg_try_t, dbg_block_t, dbg_fail_n);
#endif
    PHA_PRINT("walk_cm_strip2 vmwait index barrier acquire writeout issue status rest", kColMajor && lane == 0 && strip == 2 && f == 12);
#ifdef BEV_CS_CLOCK /* where and when the workgroup ran: HW_ID (wave, SIMD, CU, SH, SE), XCC_ID; start and end on the 100 MHz clock */
    if (tid == 0 && (kInPlace || kColMajor) && bid < kWalkTlCap) {
        long long *rec = g_walk_tl[bid];
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

template <int kSrc, bool kPow2, bool kGm>
__global__ __launch_bounds__(kStripThreads, (kSrc == kSrcColMajor || kSrc == kSrcColMajorGen) ? 3 : 4) void k_walk(BatchPtrs b, Geometry g, int nf, uint32_t want_mode)
{
    __shared__ __attribute__((aligned(16))) char arena[sizeof(WalkLds<kSrc>)];
    int f, strip;
    if (!map_block_xcd((int)blockIdx.x, nf, g.strips, f, strip)) return;
    walk_body<kSrc, kPow2, kGm>(arena, b, g, f, strip, want_mode, (int)blockIdx.x);
}

} /* namespace bevk */

#endif /* BEV_WALK_H */
