/*
 * bev_front.h — the front end's small kernels: k_probe (which frames can be read in place), k_verdict (which must be redone), k_order_scan (getOrderedCloud's last-writer table)
 * Part of the device code of libbev_mi355x.so; included by bev_kernels.hip only (one translation unit).
 */
#ifndef BEV_FRONT_H
#define BEV_FRONT_H

#include "bev_dev.h"

namespace bevk {
using namespace bevx;

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
/* The probe's LDS — since round 6 a quarter of a CU's, like every kernel of the path (bev_internal.h: kSlotLdsBytes): its
 * workgroups start wherever another one has retired (75 KB: only where two column-walk workgroups of a CU had retired
 * together).  The firing-order analysis and the tail lists never run at the same time: one piece of LDS for both. */
struct ProbeLds {
    uint32_t samp[kMaxSamples]; /* slot of sample k (position k * kProbeStride) */
    union {
        struct {
            uint16_t col[kCmMaxSamples], col1[kCmMaxSamples]; /* column of every sample and of its successor (their rows are position mod N: checked first) */
            uint32_t ref[kCmMaxRows], lo[kCmMaxRows], hi[kCmMaxRows], base[kCmMaxRows], misc[4];
        } cm;
        uint32_t tcnt[kTailBuckets]; /* tail points listed per (row, strip) */
    } u;
    uint32_t first_bad, overflow, struct_bad, struct_zero, cm_bad, cm_not_plain;
};
static_assert(sizeof(ProbeLds) <= kSlotLdsBytes, "a quarter of a CU's LDS");
__device__ __forceinline__ void probe_body(char *arena, const BatchPtrs &b, const Geometry &g, int allow_stream, int layout_hint, const int f)
{
    TL_BEGIN;
    ProbeLds &lds_p = *reinterpret_cast<ProbeLds *>(arena);
    auto &samp = lds_p.samp;
    uint32_t &first_bad = lds_p.first_bad, &overflow = lds_p.overflow;
    auto &tcnt = lds_p.u.tcnt;
    const int tid = threadIdx.x;
    const FrameDesc fd = b.frames_src ? b.frames_src[f] : b.frames[f];
    if (b.frames_copy && threadIdx.x == 0) b.frames_copy[f] = fd; /* (every later kernel of the sub-batch reads this copy: device memory) */
    const uint32_t n = fd.n_pts;
    const bev_point_t *fp = b.pts + fd.in_offset;
    /* (frames of exactly S records may be in firing order: position k = beam k % N — a stride that shares a factor with N
     * would never sample some beams (63 = 3 * 3 * 7 against a 33- or a 7-beam sensor: their rows took another row's base and
     * failed the walk's checks); 61 is prime) */
    const uint32_t stride = (n != (uint32_t)g.S && n >= (uint32_t)g.S - (uint32_t)g.S / 10u) ? (uint32_t)kProbeStrideDense
                            : ((n == (uint32_t)g.S && (g.N % 3 == 0 || g.N % 7 == 0)) ? 61u : (uint32_t)kProbeStride);
    const uint32_t ns = n ? (n - 1u) / stride + 1u : 0u;
    PH_DECL;
    PH();
    /* The caller's layout hint (bev_set_layout_hint): a frame of exactly S records is TAKEN for a structured cloud / a plain
     * sweep in firing order without a look at it — a guess like the samples', verified by the walk record by record and
     * redone when wrong; a frame of another size is probed as ever.  (Structured: the guess that decides slot 0 is "some
     * record after the first is all-zero" — what every KITTI sweep with a dropped return has, KittiPointCloudSelect.cpp:207.) */
    if (allow_stream && n == (uint32_t)g.S && (layout_hint == (int)kFrameStructured || (layout_hint == (int)kFrameColMajor && g.N >= 2))) {
        if (threadIdx.x == 0) b.info[f] = FrameInfo{n, (uint32_t)layout_hint, 0u, layout_hint == (int)kFrameStructured ? kInfoZeroGuess : 0u};
        return;
    }
    const bool can = allow_stream && n >= (uint32_t)kStreamMinPrefix && ns <= (uint32_t)kMaxSamples && g.N <= kStreamMaxRows &&
                     g.N * g.strips <= kTailBuckets && n < (1u << 24) && b.tail_list != nullptr;
    /* a structured cloud (kFrameStructured): exactly S records, every sampled one its own slot's point or empty */
    const bool can_struct = allow_stream && n == (uint32_t)g.S;
    /* ... or S returns in firing order (kFrameColMajor): every sampled record is beam (position mod N) of firing
     * (position / N); its column follows the firing — in either direction, from any start azimuth, with a base of its own
     * per row (staggered beams) — or is out of range, or is column 0 (a no-return record): see below */
    uint32_t &struct_bad = lds_p.struct_bad, &struct_zero = lds_p.struct_zero, &cm_bad = lds_p.cm_bad, &cm_not_plain = lds_p.cm_not_plain;
    const bool can_cm = can_struct && g.N >= 2; /* (the plain sweep) */
    const bool can_cm_gen = can_cm && g.N <= kCmMaxRows && g.strips <= kCmMaxStrips && ns <= (uint32_t)kCmMaxSamples && b.cm_par != nullptr;
    auto &cmcol = lds_p.u.cm.col;
    auto &cmcol1 = lds_p.u.cm.col1;
    auto &cm_ref = lds_p.u.cm.ref;
    auto &cm_lo = lds_p.u.cm.lo;
    auto &cm_hi = lds_p.u.cm.hi;
    auto &cm_base = lds_p.u.cm.base;
    auto &cm_misc = lds_p.u.cm.misc;
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
                const size_t i = (size_t)(k < ns ? k : ns - 1u) * stride;
                rc[u] = load_once(reinterpret_cast<const uint32_t *>(fp + i) + 5);                       /* row | col << 16 */
                rc1[u] = load_once(reinterpret_cast<const uint32_t *>(fp + (i + 1 < n ? i + 1 : i)) + 5); /* the sample's successor (mostly the same line): catches column-major orders at once */
            }
#pragma unroll
            for (int u = 0; u < kSPer; ++u) {
                const uint32_t k = k0 + (uint32_t)kProbeThreads * u + tid;
                if (k >= ns) continue;
                const size_t i = (size_t)k * stride;
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
                        if (col1 < (uint32_t)g.H && col1 - (uint32_t)(i + 1) / (uint32_t)g.N > (uint32_t)kPlainDisp) cm_not_plain = 1u; /* (the plain sweep: column = firing + 0 .. kPlainDisp) */
                    }
                }
                if (can_struct) {
                    if (sl0 != (uint32_t)i && rc[u] != 0u) struct_bad = 1u;
                    if (rc[u] == 0u && i >= 1) struct_zero = 1u;
                    if (row != (uint32_t)i % (uint32_t)g.N) cm_bad = 1u;
                    if (col < (uint32_t)g.H && col - (uint32_t)i / (uint32_t)g.N > (uint32_t)kPlainDisp) cm_not_plain = 1u;
                }
                if (can_cm_gen) {
                    cmcol[k] = (uint16_t)(rc[u] >> 16);
                    cmcol1[k] = i + 1 < n ? (uint16_t)(rc1[u] >> 16) : (uint16_t)0xffffu; /* (col 0xffff: out of range, not looked at) */
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
                auto disp = [&](uint32_t pos, uint32_t col, uint32_t *row, uint32_t *d) -> bool { /* a sample that says something about its row's base */
                    if (col == 0u || col >= H) return false;
                    const uint32_t fire = div_n(pos); /* (< H: the frame has S = N * H records) */
                    *row = pos - fire * N;
                    const uint32_t u = fwd ? fire : (fire ? H - fire : 0u);
                    *d = col >= u ? col - u : col + H - u;
                    return true;
                };
                for (uint32_t k = tid; k < ns; k += kProbeThreads) {
                    uint32_t row, d;
                    if (disp(k * stride, cmcol[k], &row, &d)) cm_ref[row] = d; /* (any sample of the row will do as its reference) */
                    if (disp(k * stride + 1u, cmcol1[k], &row, &d)) cm_ref[row] = d;
                }
                __syncthreads();
                auto rel = [&](uint32_t d, uint32_t ref) -> uint32_t { /* d - ref as a signed offset around the circle, biased */
                    const uint32_t t = d >= ref ? d - ref : d + H - ref;
                    return t > H / 2u ? kBias + t - H : kBias + t;
                };
                for (uint32_t k = tid; k < ns; k += kProbeThreads) {
                    uint32_t row, d;
                    if (disp(k * stride, cmcol[k], &row, &d)) {
                        atomicMin(&cm_lo[row], rel(d, cm_ref[row]));
                        atomicMax(&cm_hi[row], rel(d, cm_ref[row]));
                    }
                    if (disp(k * stride + 1u, cmcol1[k], &row, &d)) {
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
                                const uint32_t col = w ? cmcol1[k] : cmcol[k], pos = k * stride + (uint32_t)w;
                                if (col != 0u) continue; /* (its row is position mod N: cm_bad == 0) */
                                const uint32_t fire = div_n(pos), row = pos - fire * N;
                                const uint32_t u = fwd ? fire : (fire ? H - fire : 0u);
                                const uint32_t d = (2u * H - u - cm_base[row]) % H; /* (0 - u - base) mod H */
                                if (d > (uint32_t)kColMaxDisp) cm_misc[0] = 1u;
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
    const uint32_t T0 = m ? (m - 1u) * stride + 1u : 0u;         /* the last of them is position T0 - 1 */
    /* ... and the points after it, one by one, up to the first that does not ascend (at the latest the successor of the
     * sample that failed): a sweep that is sorted to its end has no tail at all, and an appended block of other points
     * starts exactly where the prefix ends — otherwise up to 62 sorted points of ONE (row, strip) would be "tail" */
    __syncthreads();
    if (tid == 0) first_bad = T0 + stride + 1u < n ? T0 + stride + 1u : n;
    __syncthreads();
    if (can && m && (uint32_t)tid <= stride) {
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
                est = lo * stride + (uint32_t)(((unsigned long long)(want - s0) * stride) / (s1 - s0));
            } else if (want >= (long long)slot_last) { /* at or beyond the prefix's last point */
                est = want > (long long)slot_last ? T : T - 1u;
            } else { /* between the last sample and the prefix's last point (position T - 1) */
                const uint32_t p0 = lo * stride;
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
__global__ __launch_bounds__(kProbeThreads) void k_probe(BatchPtrs b, Geometry g, int allow_stream, int layout_hint)
{
    __shared__ __attribute__((aligned(16))) char arena[sizeof(ProbeLds)];
    probe_body(arena, b, g, allow_stream, layout_hint, (int)blockIdx.x);
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
        /* (sorted prefixes and the plain sweep only ever raise kInfoFailed: any bit fails them; the general firing order also
         * raises kInfoCmUsed / kInfoCmStray, looked at below) */
        bool bad_stream = ((fi.mode == kFrameStream || fi.mode == kFrameColMajor) && (fi.failed != 0u || fi.consumed != fi.T)) ||
                          (fi.mode == kFrameColMajorGen && ((fi.failed & kInfoFailed) != 0u || fi.consumed != fi.T));
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

} /* namespace bevk */

#endif /* BEV_FRONT_H */
