/*
 * bev_libm.h — atanf / atan2f for the range-image projection, bit-identical
 * to glibc's single-precision implementations (glibc <= 2.40: the classic fdlibm algorithm,
 * sysdeps/ieee754/flt-32/s_atanf.c and e_atan2f.c), restated from the published algorithm.
 *
 * The reference's keyframe selectors compute row / col with atan2f
 * (MulranPointCloudSelect.cpp:121-125, OxfordPointCloudSelect.cpp:208-218); a device atan2f that
 * differs in the last bit would move points across column boundaries.  These functions use only
 * IEEE + - * / (no FMA: see bev_exact.h) and are checked against the host libm for ALL 2^32 inputs
 * (atanf) and ~10^9 random + special pairs (atan2f) in tests/test_projection_cpu.py.
 */
#ifndef BEV_LIBM_H
#define BEV_LIBM_H

#include "bev_exact.h"

namespace bevx {

BEVX_HD uint32_t float_bits(float f)
{
    union { float f; uint32_t u; } c;
    c.f = f;
    return c.u;
}

BEVX_HD float fd_atanf(float x)
{
    const float atanhi[4] = {4.6364760399e-01f, 7.8539812565e-01f, 9.8279368877e-01f, 1.5707962513e+00f};
    const float atanlo[4] = {5.0121582440e-09f, 3.7748947079e-08f, 3.4473217170e-08f, 7.5497894159e-08f};
    const float aT[11] = {3.3333334327e-01f, -2.0000000298e-01f, 1.4285714924e-01f, -1.1111110449e-01f,
                          9.0908870101e-02f, -7.6918758452e-02f, 6.6610731184e-02f, -5.8335702866e-02f,
                          4.9768779427e-02f, -3.6531571299e-02f, 1.6285819933e-02f};
    const int32_t hx = (int32_t)float_bits(x);
    const int32_t ix = hx & 0x7fffffff;
    int id;
    if (ix >= 0x4c000000) { /* |x| >= 2^25 */
        if (ix > 0x7f800000) return x + x; /* NaN */
        return hx > 0 ? atanhi[3] + atanlo[3] : -atanhi[3] - atanlo[3];
    }
    if (ix < 0x3ee00000) { /* |x| < 0.4375 */
        if (ix < 0x31000000) return x; /* |x| < 2^-29 */
        id = -1;
    } else {
        x = bits_to_float((uint32_t)ix); /* fabsf */
        if (ix < 0x3f980000) {           /* |x| < 1.1875 */
            if (ix < 0x3f300000) { id = 0; x = (2.0f * x - 1.0f) / (2.0f + x); } /* 7/16 <= |x| < 11/16 */
            else { id = 1; x = (x - 1.0f) / (x + 1.0f); }                       /* 11/16 <= |x| < 19/16 */
        } else {
            if (ix < 0x401c0000) { id = 2; x = (x - 1.5f) / (1.0f + 1.5f * x); } /* |x| < 2.4375 */
            else { id = 3; x = -1.0f / x; }
        }
    }
    const float z = x * x;
    const float w = z * z;
    const float s1 = z * (aT[0] + w * (aT[2] + w * (aT[4] + w * (aT[6] + w * (aT[8] + w * aT[10])))));
    const float s2 = w * (aT[1] + w * (aT[3] + w * (aT[5] + w * (aT[7] + w * aT[9]))));
    if (id < 0) return x - x * (s1 + s2);
    const float r = atanhi[id] - ((x * (s1 + s2) - atanlo[id]) - x);
    return hx < 0 ? -r : r;
}

BEVX_HD float fd_atan2f(float y, float x)
{
    const float tiny = 1.0e-30f, pi_o_4 = 7.8539818525e-01f, pi_o_2 = 1.5707963705e+00f, pi = 3.1415927410e+00f,
                pi_lo = -8.7422776573e-08f;
    const int32_t hx = (int32_t)float_bits(x), ix = hx & 0x7fffffff;
    const int32_t hy = (int32_t)float_bits(y), iy = hy & 0x7fffffff;
    if (ix > 0x7f800000 || iy > 0x7f800000) return x + y; /* NaN */
    if (hx == 0x3f800000) return fd_atanf(y);              /* x == 1.0 */
    const int m = ((hy >> 31) & 1) | ((hx >> 30) & 2);     /* 2 * sign(x) + sign(y) */
    if (iy == 0) return m < 2 ? y : (m == 2 ? pi + tiny : -pi - tiny);
    if (ix == 0) return hy < 0 ? -pi_o_2 - tiny : pi_o_2 + tiny;
    if (ix == 0x7f800000) {
        if (iy == 0x7f800000)
            return m == 0 ? pi_o_4 + tiny : m == 1 ? -pi_o_4 - tiny : m == 2 ? 3.0f * pi_o_4 + tiny : -3.0f * pi_o_4 - tiny;
        return m == 0 ? 0.0f : m == 1 ? -0.0f : m == 2 ? pi + tiny : -pi - tiny;
    }
    if (iy == 0x7f800000) return hy < 0 ? -pi_o_2 - tiny : pi_o_2 + tiny;
    const int k = (iy - ix) >> 23;
    float z;
    if (k > 60) z = pi_o_2 + 0.5f * pi_lo;              /* |y / x| > 2^60 */
    else if (hx < 0 && k < -60) z = 0.0f;               /* |y| / x < -2^60 */
    else z = fd_atanf(bits_to_float(float_bits(y / x) & 0x7fffffffu));
    switch (m) {
    case 0: return z;
    case 1: return bits_to_float(float_bits(z) ^ 0x80000000u);
    case 2: return pi - (z - pi_lo);
    default: return (z - pi_lo) - pi;
    }
}

/* ---------------------------------------------------------------------------
 * Range-image projection of raw XYZI returns (the "polar binning" of the pipeline).
 * uint16 conversion of a float: x86-64 truncates cvttss2si's 32-bit result.
 * ------------------------------------------------------------------------- */
BEVX_HD uint16_t to_u16(float v) { return (uint16_t)(uint32_t)cvtt_f32(v); }

/* angle in degrees as the selectors compute it: atan2f(...) / M_PI * 180.0f in double, stored to float */
BEVX_HD float degrees_of(float rad) { return (float)((double)rad / 3.14159265358979323846 * (double)180.0f); }

BEVX_HD float wrap_azimuth(float az) /* MulranPointCloudSelect.cpp:122-124 / OxfordPointCloudSelect.cpp:214-215 */
{
    if (az > 360.0f) return az - 360.0f;
    if (az < 0.0f) return az + 360.0f;
    return az;
}

/* MulranPointCloudSelect.cpp:120-125: row = k % 64, col = round(az / 360.0f * 1024) */
BEVX_HD void project_mulran(uint32_t k, float x, float y, uint16_t &row, uint16_t &col)
{
    row = (uint16_t)(k % 64u);
    const float az = wrap_azimuth(degrees_of(fd_atan2f(y, x)));
    col = to_u16(roundf(az / 360.0f * 1024.0f));
}

/* OxfordPointCloudSelect.cpp:201-218 on the already flipped x, z */
BEVX_HD void project_oxford(float x, float y, float z, uint16_t &row, uint16_t &col)
{
    const float elev = degrees_of(fd_atan2f(z, sqrtf(x * x + y * y)));
    int r = cvtt_f64(round(((double)(-elev) + 10.67) / 1.3335)); /* top-down [0, 31] */
    r = r > 31 ? 31 : r;  /* std::min(31, std::max(0, row_idx)) */
    r = r < 0 ? 0 : r;
    row = (uint16_t)r;
    const float az = wrap_azimuth(degrees_of(fd_atan2f(y, x)));
    uint16_t c = to_u16(roundf(az / 360.0f * 1056.0f));
    if (c >= 1056) c = (uint16_t)(c - 1056);
    col = c;
}

/* ---------------------------------------------------------------------------
 * KITTI (KittiPointCloudSelect.cpp:186-243): the row is a counter of azimuth zero crossings — a sequential
 * state machine in the reference.  Pieces shared by the kernels (bev_kernels.hip, k_kitti_*) and their host
 * mirror (tests/hostcheck):
 *   crossing at i (i >= 1)   az[i-1] <= 0 && az[i] > 0                                        (:214)
 *   state (ring, count)      a crossing takes ring -1 -> 0, or ring -> ring + 1 when count > Horizon_SCAN * 0.60f,
 *                            and zeroes count in both cases; every point then adds one to count  (:215-222, :242)
 * Since count = i - (position of the last accepted crossing), the accepted crossings form a chain: the next one
 * is the first crossing at a position >= last + kitti_ring_min() (or the first crossing at all while ring == -1).
 * The chain has at most 65 links before ring reaches N_SCAN and everything later is dropped, so it is walked by
 * one wave over per-block crossing lists, and each point finds its ring by counting the links at or before it.
 * ------------------------------------------------------------------------- */
constexpr int kKittiRows = 64;    /* N_SCAN, KittiPointCloudSelect.cpp:148 */
constexpr int kKittiCols = 2083;  /* Horizon_SCAN, :149 */
constexpr int kKittiBlock = 256;  /* points per crossing list */
constexpr int kKittiListCap = kKittiBlock / 2; /* two consecutive positions cannot both be crossings */
constexpr int kKittiMaxLinks = kKittiRows + 2;

BEVX_HD float kitti_azimuth(float x, float y) { return degrees_of(fd_atan2f(y, x)); } /* :192 */
BEVX_HD bool kitti_crossing(float az_prev, float az) { return az_prev <= 0.0f && az > 0.0f; } /* :214 */
BEVX_HD bool kitti_ring_full(uint32_t count) { return (float)(int)count > (float)kKittiCols * 0.60f; } /* :218 */
/* smallest count that lets a crossing start a new ring */
inline uint32_t kitti_ring_min()
{
    uint32_t m = 0;
    while (!kitti_ring_full(m)) ++m;
    return m;
}
/* :225-233; -1 where the reference would index outside the row (NaN azimuth only) */
BEVX_HD int kitti_col(float az)
{
    float a = az;                       /* makeAngleSemiPositive, :137-146 */
    if (a >= 360.0f) a = a - 360.0f;
    else if (a < 0.0f) a = a + 360.0f;
    int c = cvtt_f64(round((double)a / (360.0 / kKittiCols)));
    if (c >= kKittiCols) c -= kKittiCols;
    else if (c < 0) c += kKittiCols;
    return (c >= 0 && c < kKittiCols) ? c : -1;
}
/* ring of point i given ring0 (0 or -1) and the ascending positions of the accepted crossings */
BEVX_HD int kitti_ring_of(uint32_t i, int ring0, const uint32_t *links, uint32_t n_links)
{
    uint32_t lo = 0, hi = n_links; /* number of links <= i */
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if (links[mid] <= i) lo = mid + 1;
        else hi = mid;
    }
    return ring0 + (int)lo;
}

} /* namespace bevx */
#endif
