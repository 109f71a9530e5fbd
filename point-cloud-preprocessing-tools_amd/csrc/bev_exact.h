/*
 * bev_exact.h — the per-point / per-slot arithmetic of the hot path, written
 * once for the HIP kernels (and compiled for the host ONLY by tests/, which
 * check these closed forms against the sequential oracle without a GPU).
 *
 * Everything here must give bit-identical results on gfx950 and x86-64:
 *   - only IEEE-754 correctly rounded operations are used (+ - * / sqrt,
 *     conversions, floor/round); no libm transcendental is evaluated;
 *   - floating-point contraction is OFF (pragma below AND -ffp-contract=off),
 *     because the reference is built for baseline x86-64 without FMA
 *     (CMakeLists.txt:5-10);
 *   - float->int conversions reproduce x86-64 cvttss2si / cvttsd2si, which is
 *     what the reference's unchecked casts compile to (NaN / out of range ->
 *     INT_MIN).
 * file:line citations are relative to the reference tree.
 */
#ifndef BEV_EXACT_H
#define BEV_EXACT_H

#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__) || defined(__HIP__)
#define BEVX_HD __host__ __device__ __forceinline__
#else
#define BEVX_HD static inline
#endif

#if defined(__clang__)
#pragma clang fp contract(off)
#endif

namespace bevx {

constexpr int kGridRows = 75;  /* BatchMultiBevGen.cpp:25 */
constexpr int kGridCols = 50;  /* BatchMultiBevGen.cpp:26 */
constexpr int kGridCells = kGridRows * kGridCols;

constexpr int kIntMin = (-2147483647 - 1);

/* x86-64 cvttsd2si / cvttss2si */
BEVX_HD int cvtt_f64(double v)
{
    if (!(v > -2147483649.0 && v < 2147483648.0)) return kIntMin;
    return (int)v;
}
BEVX_HD int cvtt_f32(float v)
{
    if (!(v >= -2147483648.0f && v < 2147483648.0f)) return kIntMin;
    return (int)v;
}

/* ---------------------------------------------------------------------------
 * Phase-A angle test, BatchMultiBevGen.cpp:169-179:
 *     angle = atan2f(dz, sqrtf(dx*dx + dy*dy)) * 180.0 / M_PI   (stored to float)
 *     ground  iff  fabsf(angle - 0.0f) <= 10.0f
 * Restated without the transcendental.  With glibc 2.35's atan2f
 *   (a) fl32(fl64(a) * 180.0 / M_PI) <= 10.0f  <=>  a <= 0x1.657184p-3f
 *       (checked for every non-negative float a),
 *   (b) atanf is monotone non-decreasing over all non-negative floats and
 *       atanf(q) <= 0x1.657184p-3f  <=>  q <= kTanThreshold (exhaustive),
 *   (c) for s >= +0, atan2f(+-dz, s) = +-atanf(fl32(|dz| / s)) in every case that
 *       reaches the comparison (2e9 random + special-value cases, 0 mismatches;
 *       tests/test_angle_predicate.py re-derives (a)-(c) against the libm of
 *       the machine it runs on).
 * So: ground iff (dz == 0 and s == 0) or fl32(|dz| / s) <= kTanThreshold, with
 * NaN comparing false exactly like the reference's "<=".
 * ------------------------------------------------------------------------- */
constexpr uint32_t kTanThresholdBits = 0x3e348f0fu; /* 0.176326975f */

BEVX_HD float bits_to_float(uint32_t u)
{
    union { uint32_t u; float f; } c;
    c.u = u;
    return c.f;
}

/* written without a branch (0 / 0 is NaN and compares false, so the (0, 0) case needs no early return): the device's row
 * loop evaluates this once per slot */
BEVX_HD bool angle_is_ground(float dx, float dy, float dz)
{
    const float xx = dx * dx;
    const float yy = dy * dy;
    const float s = sqrtf(xx + yy);
    const float a = fabsf(dz);
    const float q = a / s;
    return ((a == 0.0f) & (s == 0.0f)) | (q <= bits_to_float(kTanThresholdBits)); /* atan2f(+-0, +0) = +-0 */
}
BEVX_HD bool angle_is_ground_flat(float dx, float dy, float dz) { return angle_is_ground(dx, dy, dz); }
/* The same predicate without the division (a dozen instructions on the device).  With T = kTanThreshold (odd mantissa)
 * and T' its successor, fl32(a / s) <= T  <=>  a / s < M = (T + T') / 2: below the midpoint the quotient rounds to T or
 * less, AT the midpoint the tie goes to the even neighbour T', above it to T' or more.  For s > 0 that is a < M * s, and
 * in double the product of the 25-bit M and a 24-bit float is exact.  s = 0: a < 0 is false like a / 0 = inf <= T
 * (a > 0) — the (0, 0) case is the explicit first clause, as above; a NaN anywhere compares false on both sides; an
 * infinite s gives 0 <= T on one side and a < inf on the other.  tests/hostcheck: hc_angle_nodiv_check (random and
 * threshold-straddling pairs, 0 mismatches). */
BEVX_HD bool angle_is_ground_nodiv(float dx, float dy, float dz)
{
    const float xx = dx * dx;
    const float yy = dy * dy;
    const float s = sqrtf(xx + yy);
    const float a = fabsf(dz);
    const double mid = 0.5 * ((double)bits_to_float(kTanThresholdBits) + (double)bits_to_float(kTanThresholdBits + 1u));
    return ((a == 0.0f) & (s == 0.0f)) | ((double)a < mid * (double)s);
}

/* getBelongingGrid, BatchMultiBevGen.h:73-99 -> cell = row * 50 + col.
 * The reference mixes float and double here; every step has an exact float-only equivalent
 * (checked for ALL 2^32 float inputs by tests/test_exact_forms.py against the literal expressions):
 *   (float)((double)x + 75.0) == x + 75.0f       one correctly rounded sum either way
 *   floor((double)nx / 2.0)    == floorf(nx * 0.5f) scaling by 2 is exact (nx is never subnormal)      */
BEVX_HD int floor_half_to_int(float nx)
{
    const float h = nx * 0.5f; /* exact, except that -FLT_TRUE_MIN * 0.5f rounds to -0 */
    if (nx < 0.0f && h == 0.0f) return -1;
    return (h >= -2147483648.0f && h < 2147483648.0f) ? (int)floorf(h) : kIntMin; /* cvttsd2si on NaN / overflow */
}
BEVX_HD int ground_cell(float x, float y)
{
    const float nx = x + 75.0f; /* :78 */
    const float ny = y + 50.0f; /* :79 */
    int r = floor_half_to_int(nx); /* :81 */
    int c = floor_half_to_int(ny); /* :82 */
    if (r >= kGridRows) r = kGridRows - 1; /* :84-89 */
    if (r < 0) r = 0;
    if (c >= kGridCols) c = kGridCols - 1; /* :91-96 */
    if (c < 0) c = 0;
    return r * kGridCols + c;
}

/* the same, row and column of the cell separately (the walk looks both up in per-row / per-column tables) */
BEVX_HD int ground_cell_rc(float x, float y, int *row, int *col)
{
    int r = floor_half_to_int(x + 75.0f); /* :78, :81 */
    int c = floor_half_to_int(y + 50.0f); /* :79, :82 */
    r = r >= kGridRows ? kGridRows - 1 : (r < 0 ? 0 : r); /* :84-89 */
    c = c >= kGridCols ? kGridCols - 1 : (c < 0 ? 0 : c); /* :91-96 */
    *row = r;
    *col = c;
    return r * kGridCols + c;
}

/* Phase B's count, BatchMultiBevGen.cpp:135-136, :205-206: cnt starts at 0.01f and takes "cnt = cnt + 1" once per ground
 * point of the cell, in float.  n of those steps at once: inside a binade [2^e, 2^(e+1)), e <= 23, cnt sits on the
 * binade's grid and so does cnt + j, so the j-th step is exact as long as the sum stays below 2^(e+1) — all of them
 * together are ONE exact addition; only the step that crosses into the next binade rounds (the grid doubles), and the
 * step from 0.01 to 1.01.  n < 2^24.  (tests/hostcheck: hc_count_advance_check against the step-by-step loop.) */
BEVX_HD float count_advance(float c, uint32_t n)
{
    while (n != 0u) {
        if (c >= 1.0f) {
            union { float f; uint32_t u; } top;
            top.f = c;
            top.u = (top.u & 0x7f800000u) + 0x00800000u;      /* 2^(e+1) */
            const float d = top.f - c;                         /* exact */
            const uint32_t stay = (uint32_t)ceilf(d) - 1u;     /* steps that stay below 2^(e+1): cnt + j < top  <=>  j < d */
            const uint32_t t = n < stay ? n : stay;
            c = c + (float)t;                                  /* exact: on the binade's grid */
            n -= t;
            if (n == 0u) break;
        }
        c = c + 1.0f;                                          /* the step that rounds */
        --n;
    }
    return c;
}

/* Phase C test for one slot, BatchMultiBevGen.cpp:227-241: any in-range
 * 4-neighbour cell (own cell excluded) with (double)(float)(z - avg) > 0.30. */
template <class AvgPtr>
BEVX_HD bool above_neighbour_ground(float z, int cell, AvgPtr avg)
{
    const int sr = cell / kGridCols, sc = cell % kGridCols;
    bool hit = false;
    /* (double)d > 0.30 for a float d  <=>  d >= 0.3f: 0.3f = 0.300000011920929 is the smallest float
     * above the double 0.30 (all 2^32 values of d checked by tests/test_exact_forms.py) */
    if (sr - 1 >= 0)        hit = hit || ((z - avg[cell - kGridCols]) >= 0.3f);
    if (sc + 1 < kGridCols) hit = hit || ((z - avg[cell + 1]) >= 0.3f);
    if (sc - 1 >= 0)        hit = hit || ((z - avg[cell - 1]) >= 0.3f);
    if (sr + 1 < kGridRows) hit = hit || ((z - avg[cell + kGridCols]) >= 0.3f);
    return hit;
}

/* ---------------------------------------------------------------------------
 * BEV bins.  BatchMultiBevGen.cpp:279-281 and :343-346.
 * A point's contribution to both rasters is packed into one 32-bit code:
 *   bits  0.. 8  x bin   (row index of the cv::Mat)       [0, M)
 *   bits  9..17  y bin   (col index)                      [0, M)
 *   bits 18..25  single-BEV height, already clamped       [0, 255]
 *   bits 26..30  layer, 31 = "not in any layer"           [0, n_layers)
 * kSkip marks a point that contributes to neither raster.
 * ------------------------------------------------------------------------- */
constexpr uint32_t kSkip = 0xffffffffu;
constexpr uint32_t kNoLayer = 31u;

struct RasterParams {
    float max_range_f;   /* (float)MAX_RANGE, :266 */
    float interval;      /* :264 */
    float height_res;    /* SensorParams::HEIGHT_RES */
    float lidar_to_ground; /* :269 */
    int mat_size;        /* :267 */
    int n_layers;        /* :268 */
    /* how the rasters are cut into x bands = workgroups (no influence on the result): coarse bands of `coarse` rows
     * outside [z0, z1), fine ones of `fine` rows inside — most returns lie within a few tens of metres of the sensor,
     * i.e. in the middle rows of the image, and uniform bands left two workgroups with 85 % of a frame's codes */
    int bands, coarse, fine, z0, z1;
    /* interval / height_res are powers of two in every configuration of the reference (1.0; 0.25, 0.5, 1.0): dividing
     * by 2^k and multiplying by 2^-k are the same correctly rounded operation, and the multiply is ten instructions
     * shorter on the device.  0 = not a power of two: divide. */
    float inv_interval, inv_height_res;
    /* the band of an x bin without a division (small_div below): ceil(2^20 / coarse), ceil(2^20 / fine) */
    uint32_t coarse_magic, fine_magic;
};
/* x / d for 0 <= x < 512, 1 <= d <= 512 as a multiplication by m = ceil(2^20 / d): with e = m * d - 2^20 < d,
 * x * m / 2^20 = x / d + x * e / (d * 2^20), and the second term is below 1 / d (x * e < 2^18): the floor is x / d.
 * All 512 * 512 pairs are checked by tests/test_exact_forms.py. */
BEVX_HD uint32_t small_div_magic(int d) { return ((1u << 20) + (uint32_t)d - 1u) / (uint32_t)d; }
BEVX_HD int small_div(int x, uint32_t magic) { return (int)(((uint32_t)x * magic) >> 20); }
/* 1 / v if v is a power of two (then x / v == x * (1 / v) bit for bit, for every x), else 0 */
BEVX_HD float exact_reciprocal(float v)
{
    union { float f; uint32_t u; } c;
    c.f = v;
    const uint32_t e = (c.u >> 23) & 0xffu;
    if ((c.u & 0x807fffffu) != 0u || e == 0u || e >= 253u) return 0.0f; /* sign / mantissa set, subnormal, or 1 / v not normal */
    c.u = (254u - e) << 23;
    return c.f;
}
BEVX_HD float div_interval(float a, const RasterParams &rp) { return rp.inv_interval != 0.0f ? a * rp.inv_interval : a / rp.interval; }
BEVX_HD float div_height_res(float a, const RasterParams &rp) { return rp.inv_height_res != 0.0f ? a * rp.inv_height_res : a / rp.height_res; }
BEVX_HD int raster_band_of(int x, const RasterParams &rp)
{
    if (x < rp.z0) return x / rp.coarse;
    if (x < rp.z1) return rp.z0 / rp.coarse + (x - rp.z0) / rp.fine;
    return rp.z0 / rp.coarse + (rp.z1 - rp.z0) / rp.fine + (x - rp.z1) / rp.coarse;
}
/* the same for 0 <= x < 512 with the divisions as multiplications (the kernels fill a table of M entries per workgroup) */
BEVX_HD int raster_band_of_nodiv(int x, const RasterParams &rp)
{
    const int n0 = small_div(rp.z0, rp.coarse_magic), n1 = small_div(rp.z1 - rp.z0, rp.fine_magic);
    if (x < rp.z0) return small_div(x, rp.coarse_magic);
    if (x < rp.z1) return n0 + small_div(x - rp.z0, rp.fine_magic);
    return n0 + n1 + small_div(x - rp.z1, rp.coarse_magic);
}
BEVX_HD int raster_band_x0(int band, const RasterParams &rp)
{
    const int n0 = rp.z0 / rp.coarse, n1 = (rp.z1 - rp.z0) / rp.fine;
    if (band < n0) return band * rp.coarse;
    if (band < n0 + n1) return rp.z0 + (band - n0) * rp.fine;
    return rp.z1 + (band - n0 - n1) * rp.coarse;
}
BEVX_HD int raster_band_rows(int band, const RasterParams &rp)
{
    const int n0 = rp.z0 / rp.coarse, n1 = (rp.z1 - rp.z0) / rp.fine;
    return (band >= n0 && band < n0 + n1) ? rp.fine : rp.coarse;
}

/* (int)round((double)v + 0.5), half away from zero, without doubles.  d = v + 0.5 in double:
 *   v >= 0            : d is exact, round(d) = floor(d + 0.5) = floor(v) + 1
 *   -0.5 <= v < 0     : 0 <= d < 0.5 -> 0, except that for |v| <= 2^-55 the double sum rounds to 0.5 -> 1
 *   v < -0.5          : d is exact and negative, round(d) = -floor(-d + 0.5) = ceil(v)
 * Callers only test the result against [0, M), M <= 1024; values the reference would see outside
 * +-2^31 (cvttsd2si -> INT_MIN) are reported as kIntMin as well.  Checked for all 2^32 floats. */
BEVX_HD int round_half_up_bin(float v)
{
    /* written without early returns: on the device each of them was a divergent branch in the row loop */
    const bool ok = v > -2147483000.0f && v < 2147483000.0f;
    const bool nonneg = v >= 0.0f;
    const float w = ok ? v : 0.0f;
    const float f = nonneg ? floorf(w) : ceilf(w);   /* ceil of (-0.5, 0) is -0 */
    int t = (int)f + (nonneg ? 1 : 0);
    t = (!nonneg && v >= -0.5f) ? (v >= -0x1p-55f ? 1 : 0) : t;
    return ok ? t : kIntMin;
}
/* round_half_up_bin(v) for callers that only want bins in [0, M), 2 <= M <= 4096: true and *bin if it is one.  From the
 * case analysis above: the bin is floor(v) + 1 for v >= 0 — in range iff v < M - 1 —, 0 for -1 < v < -2^-55, 1 for
 * -2^-55 <= v < 0, and negative for v <= -1.  floor(v) + 1 gives 0 on all of (-1, 0): only the sliver below zero needs a
 * patch.  Checked against round_half_up_bin for all 2^32 floats (tests/test_exact_forms.py). */
BEVX_HD bool bin_in_range(float v, int M, int *bin)
{
    const bool in = (v > -1.0f) & (v < (float)(M - 1)); /* (false for NaN) */
    const float w = in ? v : 0.0f;
    const int t = (int)floorf(w) + 1;
    *bin = ((w < 0.0f) & (w >= -0x1p-55f)) ? 1 : t;
    return in;
}
/* bin_in_range for a value that IS (p + MAX_RANGE) scaled by the interval (BatchMultiBevGen.cpp:279): the patch for
 * [-2^-55, 0) is dead code there.  The rounded sum of two floats is zero or at least half an ulp of the larger operand
 * in magnitude (p + R is tiny only for p within a binade of -R, and then it is a multiple of ulp(R) / 2): for the ranges
 * and intervals validate_params admits (an image of 16 bins or more: R / interval >= 8) the scaled value is zero or beyond
 * 2^-22 in magnitude.  Checked against bin_in_range for every coordinate and seven (range, interval) pairs
 * (tests/test_exact_forms.py). */
BEVX_HD bool bin_of_shifted(float v, int M, int *bin)
{
    const bool in = (v > -1.0f) & (v < (float)(M - 1)); /* (false for NaN) */
    *bin = (int)floorf(in ? v : 0.0f) + 1;               /* (-1, 0) -> 0 */
    return in;
}
BEVX_HD int bev_bin(float p, float max_range_f, float interval)
{
    const float shifted = (p + max_range_f) / interval;    /* float, BatchMultiBevGen.cpp:279 */
    return round_half_up_bin(shifted);
}
BEVX_HD int bev_bin_rp(float p, const RasterParams &rp) /* the same with the reciprocal where it is exact */
{
    return round_half_up_bin(div_interval(p + rp.max_range_f, rp));
}
/* (int)((double)(z + 2.0f) * 4.0): scaling by 4 is exact in float too */
BEVX_HD int height_times4(float t)
{
    const float u = t * 4.0f;
    return (u >= -2147483648.0f && u < 2147483648.0f) ? (int)u : kIntMin;
}

/* the part of the code that depends on the height only (layer, clamped height), for bins already known to be in range */
BEVX_HD uint32_t bev_code_from_bins(int x, int y, float pz, const RasterParams &rp)
{
    int layer = cvtt_f32(roundf(div_height_res(pz, rp) + rp.lidar_to_ground)); /* :281 */
    int h = height_times4(pz + rp.lidar_to_ground);                    /* :345 */
    h = h < 0 ? 0 : (h > 255 ? 255 : h);                               /* :346 */
    uint32_t l = (layer >= 0 && layer < rp.n_layers) ? (uint32_t)layer : kNoLayer;
    return (uint32_t)x | ((uint32_t)y << 9) | ((uint32_t)h << 18) | (l << 26);
}
BEVX_HD uint32_t bev_code(float px, float py, float pz, int label, const RasterParams &rp)
{
    /* straight-line (selects, no early return): the walk calls this once per slot */
    const int x = bev_bin_rp(px, rp);                                  /* :279, :343 */
    const int y = bev_bin_rp(py, rp);                                  /* :280, :344 */
    const bool in = (label != 0) &                                     /* :285, :349 */
                    ((unsigned)x < (unsigned)rp.mat_size) & ((unsigned)y < (unsigned)rp.mat_size);
    const uint32_t code = bev_code_from_bins(in ? x : 0, in ? y : 0, pz, rp);
    return in ? code : kSkip;
}
BEVX_HD int code_x(uint32_t c) { return (int)(c & 511u); }
BEVX_HD int code_y(uint32_t c) { return (int)((c >> 9) & 511u); }
BEVX_HD int code_h(uint32_t c) { return (int)((c >> 18) & 255u); }
BEVX_HD uint32_t code_layer(uint32_t c) { return (c >> 26) & 31u; }

/* ---------------------------------------------------------------------------
 * Candidate key.  A "candidate" is a slot that phase A marked ground_mat == 1 (BatchMultiBevGen.cpp:180-181): the only
 * slots whose label — and with it their BEV contribution — depends on the per-cell averages (:236-246).  The column
 * walk hands each one to the later kernels as 8 bytes: its height (float, summed by phase B) and this key:
 *   bits  0..11  getBelongingGrid cell, row * 50 + col                      (phase B, phase C)
 *   bits 12..19  column offset of the slot inside its strip                 (phase C: where to patch the label)
 *   bit  20      the walk's guess that phase C un-grounds it (the point was written with its own label)
 *   bits 21..22  x bin of the point's BEV code minus the x bin of its cell's lower edge, 0..2;  3 = "escape"
 *   bits 23..24  the same for y                                             (3 in either: read the point instead)
 *   bit  25      the point has no BEV code whatever phase C says (label 0 on input, or outside the raster)
 *   bit  26      the point's input label is -2 (what every producer writes, MulranPointCloudSelect.cpp:126): phase C
 *                can put it back without fetching the input point
 * A 2 m cell spans two or three 1 m bins, so together with the height the key reproduces the point's whole code
 * (layer and clamped height are functions of z alone).  Encoding and decoding are exact inverses by construction —
 * no assumption that the float roundings of x + 75 and x + 112 agree: when they do not, or the cell was clamped, the
 * difference is out of range and the escape value is stored.
 * ------------------------------------------------------------------------- */
constexpr uint32_t kKeyCellMask = 0x0fffu;
constexpr int kKeyColShift = 12;
constexpr uint32_t kKeyPredBit = 1u << 20;
constexpr int kKeyDxShift = 21, kKeyDyShift = 23;
constexpr uint32_t kKeyNoCodeBit = 1u << 25;
constexpr uint32_t kKeyLabelM2Bit = 1u << 26;
constexpr uint32_t kKeyEscape = 3u;

/* BEV bin of the lower edge of ground-grid row / column s (the edge is 2 * s - offset, exact in float) */
BEVX_HD int cell_edge_bin(int s, float grid_offset, const RasterParams &rp)
{
    return bev_bin_rp((float)(2 * s) - grid_offset, rp);
}
/* edge_x / edge_y: cell_edge_bin of the cell's row / column (tables in LDS on the device) */
BEVX_HD uint32_t candidate_key_edges(int cell, int col_in_strip, bool pred, uint32_t code, int label, int edge_x, int edge_y)
{
    uint32_t key = (uint32_t)cell | ((uint32_t)col_in_strip << kKeyColShift) | (pred ? kKeyPredBit : 0u) |
                   (label == -2 ? kKeyLabelM2Bit : 0u);
    if (code == kSkip) return key | kKeyNoCodeBit;
    const int dx = (int)(code & 511u) - edge_x;
    const int dy = (int)((code >> 9) & 511u) - edge_y;
    const bool ok = dx >= 0 && dx < (int)kKeyEscape && dy >= 0 && dy < (int)kKeyEscape;
    return key | ((ok ? (uint32_t)dx : kKeyEscape) << kKeyDxShift) | ((ok ? (uint32_t)dy : kKeyEscape) << kKeyDyShift);
}
BEVX_HD uint32_t candidate_key(int cell, int col_in_strip, bool pred, uint32_t code, int label, const RasterParams &rp)
{
    return candidate_key_edges(cell, col_in_strip, pred, code, label, cell_edge_bin(cell / kGridCols, 75.0f, rp),
                               cell_edge_bin(cell % kGridCols, 50.0f, rp));
}
/* the code of a candidate whose key is not an escape and has no kKeyNoCodeBit */
BEVX_HD uint32_t candidate_code(uint32_t key, float z, const RasterParams &rp)
{
    const int cell = (int)(key & kKeyCellMask);
    const int x = cell_edge_bin(cell / kGridCols, 75.0f, rp) + (int)((key >> kKeyDxShift) & 3u);
    const int y = cell_edge_bin(cell % kGridCols, 50.0f, rp) + (int)((key >> kKeyDyShift) & 3u);
    return bev_code_from_bins(x, y, z, rp);
}
BEVX_HD bool candidate_key_escapes(uint32_t key) { return ((key >> kKeyDxShift) & 3u) == kKeyEscape; }

/* ---------------------------------------------------------------------------
 * Phase A per slot.  `fetch(flat_index)` returns (x, y, z, intensity) of the
 * ordered cloud at a flat slot index.
 *
 * slot_status(): BatchMultiBevGen.cpp:142-182 for one (row, col):
 *   upper candidates tried in the reference's order, each step only if the
 *   CURRENT upper has intensity == -1:
 *     (row-1, col) -> (row-1, (col+2) % H) -> flat (row-1)*H + col - 2
 *     -> (row-2, col) if row >= 2.
 *   "(col - 2) % H" keeps the sign in C++, so for col < 2 the third candidate
 *   is the flat index (row-1)*H + col - 2 (tail of row-2), not a wrap.
 * ------------------------------------------------------------------------- */
enum : int { kInvalid = -1, kSteep = 0, kGround = 1 };

struct XYZI { float x, y, z, i; };

template <class Fetch>
BEVX_HD int slot_status(int row, int col, int H, const XYZI &lower, Fetch fetch)
{
    const long long base = (long long)(row - 1) * H;
    XYZI up = fetch(base + col);                               /* :143 */
    if (up.i == -1.0f) up = fetch(base + (col + 2) % H);        /* :146-149 */
    if (up.i == -1.0f) up = fetch(base + (col - 2));            /* :151-154 */
    if (up.i == -1.0f && row >= 2) up = fetch((long long)(row - 2) * H + col); /* :157-160 */
    if (lower.i == -1.0f || up.i == -1.0f) return kInvalid;     /* :162-167 */
    float dx = up.x - lower.x, dy = up.y - lower.y, dz = up.z - lower.z; /* :169-171 */
    return angle_is_ground(dx, dy, dz) ? kGround : kSteep;      /* :173-182 */
}

/* Value of ground_mat(row, col) when phase A ends (before phase C), in closed
 * form.  The reference walks rows N-1 .. lo (lo = N - G) per column; a GROUND
 * row r writes gm[r] = gm[r-1] = 1, an INVALID row writes gm[r] = -1 (after
 * row r+1 may have written 1 there), a STEEP row writes nothing:
 *   lo <= r <= N-1 : -1 if s[r] INVALID;  1 if s[r] GROUND;
 *                    else 1 if (r+1 <= N-1 and s[r+1] GROUND) else 0
 *   r == lo-1      : 1 if s[lo] GROUND else 0
 *   r <  lo-1      : 0                                                      */
template <class Fetch>
BEVX_HD int phase_a_ground(int row, int col, int N, int H, int G, const XYZI &self, Fetch fetch)
{
    const int lo = N - G;
    if (row < lo - 1) return 0;
    if (row >= lo) {
        int s = slot_status(row, col, H, self, fetch);
        if (s == kInvalid) return -1;
        if (s == kGround) return 1;
    }
    if (row + 1 <= N - 1) {
        XYZI below = fetch((long long)(row + 1) * H + col);
        return slot_status(row + 1, col, H, below, fetch) == kGround ? 1 : 0;
    }
    return 0;
}

} /* namespace bevx */
#endif
