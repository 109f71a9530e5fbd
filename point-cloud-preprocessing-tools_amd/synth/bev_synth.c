/*
 * bev_synth.c — deterministic synthetic LiDAR frames for tests and bench.py.
 *
 * Not part of the product path and not part of the oracle: it only MAKES
 * inputs (BASELINE.json configs).  Bit-reproducible on any
 * IEEE-754 host: integer hashing (splitmix64 finaliser) plus + - * / only;
 * sin/cos come from fixed polynomials evaluated in double (no libm, whose
 * last-ulp results differ between versions).  Build with -ffp-contract=off.
 *
 * The producers these frames imitate are the reference's keyframe selectors:
 * label = -2 on real points (MulranPointCloudSelect.cpp:126,
 * KittiPointCloudSelect.cpp:237), row/col precomputed, intensity == -1 as the
 * "no return" marker honoured by markGroundPoints (BatchMultiBevGen.cpp:146).
 */
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/bev_mi355x.h"

#define SYNTH_PI 3.14159265358979323846

static uint64_t mix64(uint64_t z)
{
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
    return z ^ (z >> 31);
}
static uint64_t hash3(uint64_t seed, uint64_t a, uint64_t b)
{
    return mix64(mix64(seed + 0x9e3779b97f4a7c15ULL * (a + 1)) + 0xd1b54a32d192ed03ULL * (b + 1));
}
/* uniform in [0,1): 24 random bits, exact in float and double */
static double u01(uint64_t h) { return (double)(h >> 40) * (1.0 / 16777216.0); }

/* sin/cos on [-pi, pi] from Taylor polynomials after folding to [-pi/2, pi/2] */
static double poly_sin(double x) /* |x| <= pi/2 */
{
    double x2 = x * x;
    double s = 1.0 / 51090942171709440000.0; /* 1/21! */
    s = s * x2 - 1.0 / 121645100408832000.0; /* 1/19! */
    s = s * x2 + 1.0 / 355687428096000.0;    /* 1/17! */
    s = s * x2 - 1.0 / 1307674368000.0;      /* 1/15! */
    s = s * x2 + 1.0 / 6227020800.0;         /* 1/13! */
    s = s * x2 - 1.0 / 39916800.0;           /* 1/11! */
    s = s * x2 + 1.0 / 362880.0;             /* 1/9! */
    s = s * x2 - 1.0 / 5040.0;               /* 1/7! */
    s = s * x2 + 1.0 / 120.0;
    s = s * x2 - 1.0 / 6.0;
    s = s * x2 + 1.0;
    return s * x;
}
static double det_sin(double x) /* |x| <= 2*pi */
{
    if (x > SYNTH_PI) x -= 2.0 * SYNTH_PI;
    if (x < -SYNTH_PI) x += 2.0 * SYNTH_PI;
    if (x > SYNTH_PI / 2) x = SYNTH_PI - x;
    if (x < -SYNTH_PI / 2) x = -SYNTH_PI - x;
    return poly_sin(x);
}
static double det_cos(double x)
{
    x += SYNTH_PI / 2;
    if (x > SYNTH_PI) x -= 2.0 * SYNTH_PI;
    return det_sin(x);
}

typedef struct synth_geom {
    int n_scan, horizon_scan;
    double elev_top_deg, elev_bottom_deg; /* row 0 .. row n_scan-1 */
    double sensor_height;
} synth_geom_t;

static void geom_for(const bev_params_t *p, synth_geom_t *g)
{
    g->n_scan = p->n_scan;
    g->horizon_scan = p->horizon_scan;
    g->sensor_height = 1.73;
    if (p->n_scan == 32) { /* HDL-32E: +10.67 .. -30.67 (OxfordPointCloudSelect.cpp:209) */
        g->elev_top_deg = 10.67; g->elev_bottom_deg = -30.67;
    } else if (p->horizon_scan == 1024) { /* OS1-64 */
        g->elev_top_deg = 16.6; g->elev_bottom_deg = -16.6;
    } else { /* HDL-64E: +2 .. -24.8 */
        g->elev_top_deg = 2.0; g->elev_bottom_deg = -24.8;
    }
}

typedef struct beam_dir { double se, ce; int down; } beam_dir_t;

static beam_dir_t beam_for_row(const synth_geom_t *g, int row)
{
    double t = g->n_scan > 1 ? (double)row / (double)(g->n_scan - 1) : 0.0;
    double elev = (g->elev_top_deg + (g->elev_bottom_deg - g->elev_top_deg) * t) * (SYNTH_PI / 180.0);
    beam_dir_t b;
    b.se = det_sin(elev);
    b.ce = det_cos(elev);
    b.down = elev < -0.5 * (SYNTH_PI / 180.0);
    return b;
}

/* One return for a beam (elevation `bd`, azimuth direction (ca, sa)), random stream `key`. */
static void make_return_dir(const synth_geom_t *g, uint64_t seed, uint64_t key, beam_dir_t bd,
                            double ca, double sa, bev_point_t *out)
{
    double u1 = u01(hash3(seed, key, 1));
    double u2 = u01(hash3(seed, key, 2));
    uint64_t h4 = hash3(seed, key, 4);
    double range;
    if (bd.down) { /* down-looking: flat ground +-3 % */
        range = g->sensor_height / (-bd.se) * (1.0 + 0.03 * (2.0 * u1 - 1.0));
        if (range > 80.0) range = 80.0;
    } else { /* up-looking: something between 5 and 75 m */
        range = 5.0 + 70.0 * u1;
    }
    if (u2 < 0.25) range *= 0.3 + 0.6 * u01(hash3(seed, key, 3)); /* "objects": shortened returns */
    memset(out, 0, sizeof(*out));
    out->x = (float)(range * bd.ce * ca);
    out->y = (float)(range * bd.ce * sa);
    out->z = (float)(range * bd.se);
    /* 5 % "no return" markers; otherwise a uniform intensity from other bits of the same hash */
    out->intensity = (u01(h4) < 0.05) ? -1.0f : (float)((double)(h4 & 0xffffffu) * (1.0 / 16777216.0));
    out->label = -2;
}

static void make_return(const synth_geom_t *g, uint64_t seed, uint64_t key, int row, double az, bev_point_t *out)
{
    make_return_dir(g, seed, key, beam_for_row(g, row), det_cos(az), det_sin(az), out);
}

/* kind 0 — structured sweep (config 2 / config 1 / config 4):
 * row-major kept slots (keep probability `keep`), then `n_dup` appended
 * duplicates of random slots (exercise last-writer-wins).  Returns the number
 * of points written (<= cap), or the number needed if out == NULL. */
size_t bev_synth_sweep(const bev_params_t *p, uint64_t seed, uint32_t frame_id,
                       double keep, uint32_t n_dup, bev_point_t *out, size_t cap)
{
    synth_geom_t g;
    geom_for(p, &g);
    const uint64_t S = (uint64_t)g.n_scan * (uint64_t)g.horizon_scan;
    const uint64_t fseed = mix64(seed + 0x100000001b3ULL * (uint64_t)frame_id);
    size_t n = 0;
    double *tab = (double *)malloc(sizeof(double) * 2 * (size_t)g.horizon_scan);
    if (!tab) return 0;
    for (int c = 0; c < g.horizon_scan; ++c) {
        double az = 2.0 * SYNTH_PI * (double)c / (double)g.horizon_scan;
        tab[2 * c] = det_cos(az);
        tab[2 * c + 1] = det_sin(az);
    }
    for (int r = 0; r < g.n_scan; ++r) {
        const beam_dir_t bd = beam_for_row(&g, r);
        for (int c = 0; c < g.horizon_scan; ++c) {
            uint64_t slot = (uint64_t)r * g.horizon_scan + c;
            if (u01(hash3(fseed, slot, 0)) >= keep) continue;
            if (out) {
                if (n >= cap) { free(tab); return n; }
                make_return_dir(&g, fseed, slot, bd, tab[2 * c], tab[2 * c + 1], &out[n]);
                out[n].row = (uint16_t)r;
                out[n].col = (uint16_t)c;
                out[n].t = frame_id;
            }
            ++n;
        }
    }
    free(tab);
    for (uint32_t j = 0; j < n_dup; ++j) {
        if (out) {
            if (n >= cap) return n;
            uint64_t slot = hash3(fseed, j, 7) % S;
            int r = (int)(slot / (uint64_t)g.horizon_scan);
            int c = (int)(slot % (uint64_t)g.horizon_scan);
            double az = 2.0 * SYNTH_PI * (double)c / (double)g.horizon_scan;
            make_return(&g, fseed, S + j, r, az, &out[n]); /* different stream -> different xyz */
            out[n].row = (uint16_t)r;
            out[n].col = (uint16_t)c;
            out[n].t = frame_id;
        }
        ++n;
    }
    return n;
}

/* kind 1 — MulRan/Ouster firing order (config 3; MulranPointCloudSelect.cpp:
 * 112-130): point k has row = k % n_scan, col = round(az/360 * H) which can be
 * == H (dropped later by getOrderedCloud's bounds test); native intensity
 * (never -1).  Always n_scan*horizon_scan points. */
size_t bev_synth_firing_order(const bev_params_t *p, uint64_t seed, uint32_t frame_id,
                              bev_point_t *out, size_t cap)
{
    synth_geom_t g;
    geom_for(p, &g);
    const uint64_t S = (uint64_t)g.n_scan * (uint64_t)g.horizon_scan;
    const uint64_t fseed = mix64(seed + 0x100000001b3ULL * (uint64_t)frame_id);
    if (!out) return (size_t)S;
    size_t n = 0;
    for (uint64_t k = 0; k < S && n < cap; ++k, ++n) {
        int r = (int)(k % (uint64_t)g.n_scan);
        uint64_t fire = k / (uint64_t)g.n_scan;
        double jitter = u01(hash3(fseed, k, 6)) - 0.5; /* +-half a column */
        double az_deg = 360.0 * ((double)fire + 0.5 + jitter) / (double)g.horizon_scan;
        if (az_deg < 0.0) az_deg += 360.0;
        double az = az_deg * (SYNTH_PI / 180.0);
        make_return(&g, fseed, k, r, az, &out[n]);
        if (out[n].intensity == -1.0f) out[n].intensity = 0.5f;
        /* round half away from zero of a non-negative value */
        double colf = az_deg / 360.0 * (double)g.horizon_scan;
        uint32_t col = (uint32_t)(colf + 0.5);
        out[n].row = (uint16_t)r;
        out[n].col = (uint16_t)col;
        out[n].t = frame_id;
    }
    return n;
}

/* kind 1b — what mulran_point_cloud_select writes for a REAL Ouster sweep (MulranPointCloudSelect.cpp:112-130), beyond
 * the idealised kind 1: the sweep starts at an arbitrary azimuth (`phase`, in columns) and turns either way (`dir`),
 * the 64 lasers sit in four staggered columns (beam b fires `stagger`-scaled +9, +3, -3, -9 columns off its firing's
 * azimuth, b mod 4), and a share `noret` of the records are no-returns: x = y = z = 0, for which the selector computes
 * atan2(0, 0) = 0 and hence column 0 of the record's row — every such record lands in slot (row, 0), the last one in
 * input order wins.  `t` carries the record's position so that WHICH record won is visible in the output. */
size_t bev_synth_firing_real(const bev_params_t *p, uint64_t seed, uint32_t frame_id, double noret, int phase, int dir,
                             double stagger, bev_point_t *out, size_t cap)
{
    static const double beam_off[4] = { 9.0, 3.0, -3.0, -9.0 };
    synth_geom_t g;
    geom_for(p, &g);
    const uint64_t S = (uint64_t)g.n_scan * (uint64_t)g.horizon_scan;
    const uint64_t fseed = mix64(seed + 0x100000001b3ULL * (uint64_t)frame_id);
    if (!out) return (size_t)S;
    const double H = (double)g.horizon_scan;
    size_t n = 0;
    for (uint64_t k = 0; k < S && n < cap; ++k, ++n) {
        int r = (int)(k % (uint64_t)g.n_scan);
        uint64_t fire = k / (uint64_t)g.n_scan;
        if (u01(hash3(fseed, k, 10)) < noret) { /* no return: the origin, column 0 */
            memset(&out[n], 0, sizeof(out[n]));
            out[n].label = -2;
            out[n].row = (uint16_t)r;
            out[n].col = 0;
            out[n].t = (uint32_t)k;
            continue;
        }
        double jitter = u01(hash3(fseed, k, 6)) - 0.5; /* +-half a column */
        double colpos = (double)phase + (dir < 0 ? -(double)fire : (double)fire) + stagger * beam_off[r & 3] + 0.5 + jitter;
        colpos -= H * (double)(int64_t)(colpos / H); /* into (-H, H) */
        if (colpos < 0.0) colpos += H;
        double az_deg = 360.0 * colpos / H;
        double az = az_deg * (SYNTH_PI / 180.0);
        make_return(&g, fseed, k, r, az, &out[n]);
        if (out[n].intensity == -1.0f) out[n].intensity = 0.5f;
        double colf = az_deg / 360.0 * H;
        uint32_t col = (uint32_t)(colf + 0.5); /* round half away from zero of a non-negative value; may be == H */
        out[n].row = (uint16_t)r;
        out[n].col = (uint16_t)col;
        out[n].t = (uint32_t)k;
    }
    return n;
}

/* kind 2 — Oxford-style concatenation (config 5): `n_sweeps` sweeps of the
 * same sensor with a small per-sweep pose jitter, every sweep row-major, so
 * P ~ n_sweeps * S * keep points land in S slots. */
size_t bev_synth_concat(const bev_params_t *p, uint64_t seed, uint32_t frame_id,
                        uint32_t n_sweeps, double keep, bev_point_t *out, size_t cap)
{
    synth_geom_t g;
    geom_for(p, &g);
    const uint64_t S = (uint64_t)g.n_scan * (uint64_t)g.horizon_scan;
    const uint64_t fseed = mix64(seed + 0x100000001b3ULL * (uint64_t)frame_id);
    size_t n = 0;
    for (uint32_t s = 0; s < n_sweeps; ++s) {
        double jx = 0.2 * (u01(hash3(fseed, s, 8)) - 0.5);
        double jy = 0.2 * (u01(hash3(fseed, s, 9)) - 0.5);
        for (uint64_t slot = 0; slot < S; ++slot) {
            uint64_t key = (uint64_t)s * S + slot;
            if (u01(hash3(fseed, key, 0)) >= keep) continue;
            if (out) {
                if (n >= cap) return n;
                int r = (int)(slot / (uint64_t)g.horizon_scan);
                int c = (int)(slot % (uint64_t)g.horizon_scan);
                double az = 2.0 * SYNTH_PI * (double)c / (double)g.horizon_scan;
                make_return(&g, fseed, key, r, az, &out[n]);
                out[n].x = (float)((double)out[n].x + jx);
                out[n].y = (float)((double)out[n].y + jy);
                out[n].row = (uint16_t)r;
                out[n].col = (uint16_t)c;
                out[n].t = frame_id;
            }
            ++n;
        }
    }
    return n;
}

/* kind 3 — adversarial cloud for edge-case tests: random rows/cols including
 * out-of-range ones, heavy duplication, labels from {-2,-1,0,1,7}, a share of
 * intensity == -1, coordinates that sit exactly on 2 m / 1 m cell boundaries
 * and layer boundaries, far outliers, and (if with_nonfinite) NaN / Inf /
 * huge values. */
size_t bev_synth_adversarial(const bev_params_t *p, uint64_t seed, uint32_t n_points,
                             int with_nonfinite, bev_point_t *out, size_t cap)
{
    static const int16_t labels[5] = { -2, -1, 0, 1, 7 };
    size_t n = 0;
    for (uint32_t i = 0; i < n_points && n < cap; ++i, ++n) {
        uint64_t h0 = hash3(seed, i, 0), h1 = hash3(seed, i, 1), h2 = hash3(seed, i, 2);
        uint64_t h3 = hash3(seed, i, 3), h4 = hash3(seed, i, 4), h5 = hash3(seed, i, 5);
        bev_point_t *q = &out[n];
        memset(q, 0, sizeof(*q));
        int mode = (int)(h0 % 10);
        double x = (u01(h1) - 0.5) * 260.0, y = (u01(h2) - 0.5) * 260.0, z = (u01(h3) - 0.5) * 14.0;
        if (mode == 0) { /* exactly on cell boundaries */
            x = (double)((int64_t)(h1 % 241) - 120);
            y = (double)((int64_t)(h2 % 241) - 120);
            z = (double)((int64_t)(h3 % 41) - 20) * 0.125;
        } else if (mode == 1) { /* half-cell offsets (round-half cases) */
            x = (double)((int64_t)(h1 % 241) - 120) + 0.5;
            y = (double)((int64_t)(h2 % 241) - 120) - 0.5;
            z = (double)((int64_t)(h3 % 41) - 20) * 0.25 + 0.125;
        } else if (mode == 2) { /* near the sensor, dense */
            x = (u01(h1) - 0.5) * 6.0; y = (u01(h2) - 0.5) * 6.0; z = -1.73 + (u01(h3) - 0.5) * 0.2;
        } else if (mode == 3) { /* far outliers */
            x = (u01(h1) - 0.5) * 4000.0; y = (u01(h2) - 0.5) * 4000.0; z = (u01(h3) - 0.5) * 300.0;
        }
        q->x = (float)x; q->y = (float)y; q->z = (float)z;
        if (with_nonfinite && (h4 % 97) == 0) {
            union { uint32_t u; float f; } nanv = { 0x7fc00000u }, infv = { 0x7f800000u };
            switch ((h4 >> 8) % 6) {
            case 0: q->x = nanv.f; break;
            case 1: q->y = infv.f; break;
            case 2: q->z = -infv.f; break;
            case 3: q->z = 3.0e38f; break;
            case 4: q->x = -3.0e38f; break;
            default: q->z = nanv.f; break;
            }
        }
        q->intensity = ((h4 >> 20) % 8 == 0) ? -1.0f : (float)u01(h5);
        uint32_t rr = (uint32_t)((h5 >> 3) % (uint64_t)(p->n_scan + 3));       /* up to 2 rows OOB */
        uint32_t cc = (uint32_t)((h5 >> 23) % (uint64_t)(p->horizon_scan + 5)); /* up to 4 cols OOB */
        if ((h0 >> 8) % 3 == 0) cc = (uint32_t)((h5 >> 23) % 7);                /* pile on cols 0..6 */
        if ((h0 >> 12) % 50 == 0) rr = 0xffffu;
        q->row = (uint16_t)rr; q->col = (uint16_t)cc;
        q->t = i;
        q->label = labels[(h0 >> 16) % 5];
    }
    return n;
}
