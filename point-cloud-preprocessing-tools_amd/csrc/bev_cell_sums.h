/*
 * bev_cell_sums.h — markGroundPoints phase B: ordered float32 sums per 2 m cell
 * Part of the device code of libbev_mi355x.so; included by bev_kernels.hip only (one translation unit).
 */
#ifndef BEV_CELL_SUMS_H
#define BEV_CELL_SUMS_H

#include "bev_dev.h"

namespace bevk {
using namespace bevx;

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

/* quarter `quarter` (position in the frame's launch slots; turned below) of frame f; lds: SumDims::lds_bytes(g.segs) bytes */
__device__ __forceinline__ void cell_sums_body(uint32_t *lds, const BatchPtrs &b, const Geometry &g, const int f, int quarter, int bid /* developer builds */)
{
    TL_BEGIN;
    using D = SumDims;
    constexpr int kCellsQ = D::cells, kHistStride = D::hist_stride, kTouchWords = D::touch_words;
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

    (void)bid; /* (the quarters of a frame on one XCD: they read the same lines) */
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
    PHA_PRINT("cell_sums barrier0 - scan place sum request histloop looptop", tid == 0 && bid == 100);
    PH();
    float *avg = b.avg + (size_t)f * kCells;
    for (int c = tid; c < kCellsQ; c += kSumThreads)
        if (c * kSumQ + quarter < kCells) avg[c * kSumQ + quarter] = sumv[c] / cntv[c]; /* :210 */
    PH_PRINT("cell_sums all-parts", tid == 0 && bid == 100);
    TL_END(K_CELL_SUMS);
#ifdef BEV_CS_TL /* developer build: start / end of every workgroup of this launch in the walk's timeline records */
    if (tid == 0 && f == 12)
        printf("cell_sums frame 12 quarter %d: %d candidates, %d parts; x10 ns: list %lld scan %lld place %lld sum %lld data-wait %lld request %lld hist %lld top %lld\n",
               quarter, GC, P, pha_[0], pha_[2], pha_[3], pha_[4], pha_[1], pha_[5], pha_[6], pha_[7]);
    if (tid == 0 && bid < kWalkTlCap) {
        long long *rec = g_walk_tl[bid];
        rec[0] = ph_clk[0];
        rec[1] = wall_clock64();
        rec[2] = (long long)(unsigned)__builtin_amdgcn_s_getreg((31 << 11) | 4);
        rec[3] = (long long)(unsigned)__builtin_amdgcn_s_getreg((31 << 11) | 20) | ((long long)GC << 32) | ((long long)quarter << 8);
    }
#endif
}

__global__ __launch_bounds__(kSumThreads, 4) void k_cell_sums(BatchPtrs b, Geometry g, int nf)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t lds_dyn[];
    int f, quarter;
    if (!map_block_xcd((int)blockIdx.x, nf, kSumQ, f, quarter)) return;
    cell_sums_body(lds_dyn, b, g, f, quarter, (int)blockIdx.x);
}

} /* namespace bevk */

#endif /* BEV_CELL_SUMS_H */
