/*
 * bev_oracle.c — CPU restatement (plain C, single thread, sequential loops) of
 * the batch_multi_bev_gen hot path.  See bev_oracle.h: TEST INFRASTRUCTURE,
 * PARITY UNPINNED.  Compile with -O3 -ffp-contract=off and NO -march /
 * -ffast-math, mirroring the reference build (CMakeLists.txt:5-10: x86-64
 * baseline, no FMA contraction, IEEE semantics).
 *
 * The loops deliberately keep the reference's iteration ORDER (column-major
 * walk in phase A, row-major float accumulation in phase B, last-writer-wins
 * scatter) because those orders are observable in the results.
 */
#include "bev_oracle.h"

#include <limits.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------
 * x86-64 float->int conversions as the reference binary performs them.
 * C leaves out-of-range conversions undefined; g++ -O3 on x86-64 emits
 * cvttsd2si / cvttss2si, which return 0x80000000 for NaN and out-of-range.
 * ---------------------------------------------------------------------- */
static int cvtt_f64_to_i32(double v)
{
    if (!(v > -2147483649.0 && v < 2147483648.0)) return INT_MIN;
    return (int)v;
}
static int cvtt_f32_to_i32(float v)
{
    if (!(v >= -2147483648.0f && v < 2147483648.0f)) return INT_MIN;
    return (int)v;
}

/* src/Utility.cpp:92-124 */
int oracle_sensor_params(int kind, oracle_sensor_t *out)
{
    switch (kind) {
    case 0: /* HDL_32E, :98-103 */
        out->n_scan = 32; out->horizon_scan = 1056;
        out->ground_upper_scan = 20; out->height_res = 0.5f;
        return 0;
    case 1: /* HDL_64E, :105-111 */
        out->n_scan = 64; out->horizon_scan = 2083;
        out->ground_upper_scan = 50; out->height_res = 0.25f;
        return 0;
    case 2: /* OS1_64, :113-118 */
        out->n_scan = 64; out->horizon_scan = 1024;
        out->ground_upper_scan = 31; out->height_res = 1.0f;
        return 0;
    default:
        return -1;
    }
}

/* BatchMultiBevGen.h:73-99.
 * :78-79  float normalized = p + 75.0 (double add, rounded to float)
 * :81-82  (int) floor(normalized / 2.0) (double)
 * :84-96  clamp to [0,74] x [0,49] */
void oracle_belonging_grid(float x, float y, int *sector_row, int *sector_col)
{
    float nx = (float)((double)x + 75.0);
    float ny = (float)((double)y + 50.0);
    int r = cvtt_f64_to_i32(floor((double)nx / 2.0));
    int c = cvtt_f64_to_i32(floor((double)ny / 2.0));
    if (r >= ORACLE_GRID_ROWS) r = ORACLE_GRID_ROWS - 1;
    if (r < 0) r = 0;
    if (c >= ORACLE_GRID_COLS) c = ORACLE_GRID_COLS - 1;
    if (c < 0) c = 0;
    *sector_row = r;
    *sector_col = c;
}

/* BatchMultiBevGen.cpp:94-117.
 * :98      resize(S) value-initialises -> all-zero points
 * :102-116 input order scatter, bounds test, plain struct assignment */
void oracle_order_cloud(const oracle_sensor_t *sp, const oracle_point_t *in,
                        size_t n_in, oracle_point_t *out)
{
    const size_t S = (size_t)sp->n_scan * (size_t)sp->horizon_scan;
    memset(out, 0, S * sizeof(*out));
    for (size_t i = 0; i < n_in; ++i) {
        int row = in[i].row; /* :103 */
        int col = in[i].col; /* :104 */
        if (row < 0 || row >= sp->n_scan) continue;       /* :106-108 */
        if (col < 0 || col >= sp->horizon_scan) continue; /* :109-111 */
        out[(size_t)row * sp->horizon_scan + col] = in[i]; /* :113-115 */
    }
}

/* BatchMultiBevGen.cpp:169-179 with the float overloads of sqrt / atan2 / abs
 * (SURVEY.md §8(a) row A2): products and sum in float, sqrtf, atan2f, then
 * "* 180.0 / M_PI" in double, stored to a float, compared as float. */
int oracle_angle_is_ground(float diff_x, float diff_y, float diff_z)
{
    float horiz = sqrtf(diff_x * diff_x + diff_y * diff_y);
    float angle = (float)((double)atan2f(diff_z, horiz) * 180.0 / M_PI);
    const float mount = 0.0f; /* :175 */
    return fabsf(angle - mount) <= 10.0f;
}

/* The OTHER overload reading of the same lines.  BatchMultiBevGen.h:38 has "using namespace std" commented out, so
 * the unqualified sqrt / atan2 / abs at :173,:179 bind to whatever the third-party headers made visible: the float
 * overloads (above, adopted) when libstdc++'s <math.h> / <cmath> wrappers are in scope, the C double functions
 * otherwise.  Double reading: the products and their sum are still float (float operands), sqrt and atan2 run in
 * double on the promoted values, the result is stored to the float `angle` (:126), fabs of a float is exact. */
int oracle_angle_is_ground_f64(float diff_x, float diff_y, float diff_z)
{
    double horiz = sqrt((double)(diff_x * diff_x + diff_y * diff_y));
    float angle = (float)(atan2((double)diff_z, horiz) * 180.0 / M_PI);
    const float mount = 0.0f;
    return fabs((double)(angle - mount)) <= 10.0f;
}

/* BatchMultiBevGen.cpp:119-252 */
void oracle_mark_ground(const oracle_sensor_t *sp, oracle_point_t *cloud,
                        int8_t *ground_mat, float *avg_out)
{
    oracle_mark_ground_variant(sp, cloud, ground_mat, avg_out, ORACLE_ANGLE_F32);
}

void oracle_mark_ground_variant(const oracle_sensor_t *sp, oracle_point_t *cloud,
                                int8_t *ground_mat, float *avg_out, int angle_variant)
{
    const int N = sp->n_scan, H = sp->horizon_scan, G = sp->ground_upper_scan;
    const size_t S = (size_t)N * (size_t)H;
    const int cells = ORACLE_GRID_ROWS * ORACLE_GRID_COLS;
    float height_sum[ORACLE_GRID_ROWS * ORACLE_GRID_COLS];
    float height_cnt[ORACLE_GRID_ROWS * ORACLE_GRID_COLS];

    memset(ground_mat, 0, S); /* :123 */
    for (int k = 0; k < cells; ++k) {
        height_sum[k] = 0.0f;                 /* :133-134 */
        height_cnt[k] = (float)(1.0f * 0.01); /* :135-136, MatExpr scale */
    }

    /* phase A, :139-184: columns outer, rows N-1 down to N-G */
    for (int col = 0; col < H; ++col) {
        for (int row = N - 1; row > N - G - 1; --row) {
            size_t lower = (size_t)row * H + col;       /* :142 */
            size_t upper = (size_t)(row - 1) * H + col; /* :143 */
            if (cloud[upper].intensity == -1) {         /* :146-149 */
                int c2 = (col + 2) % H;
                upper = (size_t)(row - 1) * H + c2;
            }
            if (cloud[upper].intensity == -1) {         /* :151-154 */
                int c2 = (col - 2) % H; /* C remainder: negative for col < 2 */
                upper = (size_t)((long)(row - 1) * H + c2);
            }
            if (cloud[upper].intensity == -1 && row >= 2) { /* :157-160 */
                upper = (size_t)(row - 2) * H + col;
            }
            if (cloud[lower].intensity == -1 || cloud[upper].intensity == -1) {
                ground_mat[lower] = -1; /* :162-167 */
                continue;
            }
            float dx = cloud[upper].x - cloud[lower].x; /* :169-171 */
            float dy = cloud[upper].y - cloud[lower].y;
            float dz = cloud[upper].z - cloud[lower].z;
            if (angle_variant == ORACLE_ANGLE_F64 ? oracle_angle_is_ground_f64(dx, dy, dz)
                                                  : oracle_angle_is_ground(dx, dy, dz)) { /* :173-182 */
                ground_mat[(size_t)row * H + col] = 1;
                ground_mat[(size_t)(row - 1) * H + col] = 1;
            }
        }
    }

    /* phase B, :187-208: row-major, float accumulation in that order */
    for (int row = 0; row < N; ++row) {
        for (int col = 0; col < H; ++col) {
            size_t idx = (size_t)row * H + col;
            if (ground_mat[idx] != 1) continue;
            int sr, sc;
            oracle_belonging_grid(cloud[idx].x, cloud[idx].y, &sr, &sc);
            int k = sr * ORACLE_GRID_COLS + sc;
            height_sum[k] += cloud[idx].z;      /* :198-199 */
            height_cnt[k] = height_cnt[k] + 1;  /* :205-206 */
        }
    }
    for (int k = 0; k < cells; ++k) /* :210, element-wise float divide */
        height_sum[k] = height_sum[k] / height_cnt[k];
    if (avg_out) memcpy(avg_out, height_sum, sizeof(height_sum));

    /* phase C, :216-250: every slot against its 4 neighbour cells, in the
     * order of setNeighbors (:73-84): (-1,0) (0,1) (0,-1) (1,0) */
    static const int nb[4][2] = { {-1, 0}, {0, 1}, {0, -1}, {1, 0} };
    for (int row = 0; row < N; ++row) {
        for (int col = 0; col < H; ++col) {
            size_t idx = (size_t)row * H + col;
            int sr, sc;
            oracle_belonging_grid(cloud[idx].x, cloud[idx].y, &sr, &sc);
            for (int k = 0; k < 4; ++k) {
                int nr = sr + nb[k][0], nc = sc + nb[k][1];
                if (nr < 0 || nr >= 75 || nc < 0 || nc >= 50) continue; /* :231-234 */
                float d = cloud[idx].z - height_sum[nr * ORACLE_GRID_COLS + nc];
                if ((double)d > 0.30) { /* :236-240, double literal */
                    ground_mat[idx] = 0;
                    break;
                }
            }
            if (ground_mat[idx] == 1) cloud[idx].label = 0; /* :244-246 */
        }
    }
}

/* x / y bin of both rasters, BatchMultiBevGen.cpp:279-280 and :343-344:
 * (p + MAX_RANGE) and "/ interval" in float, "+ 0.5" and round() in double,
 * then double -> int. */
static int bev_bin(float p, int max_range, float interval)
{
    float shifted = (p + (float)max_range) / interval;
    return cvtt_f64_to_i32(round((double)shifted + 0.5));
}

/* BatchMultiBevGen.cpp:266-292 */
void oracle_multi_bev(const oracle_sensor_t *sp, const oracle_point_t *cloud,
                      size_t n, float interval, uint8_t *out)
{
    const int max_range = 112;                                   /* :266 */
    const int M = cvtt_f32_to_i32((float)(max_range * 2) / interval); /* :267 */
    const int layers = 24;                                       /* :268,:271 */
    const float lidar_to_ground = 2.0f;                          /* :269 */
    memset(out, 0, (size_t)layers * M * M);                      /* :272-275 */
    for (size_t i = 0; i < n; ++i) {
        const oracle_point_t *p = &cloud[i];
        int x = bev_bin(p->x, max_range, interval);              /* :279 */
        int y = bev_bin(p->y, max_range, interval);              /* :280 */
        int layer = cvtt_f32_to_i32(roundf(p->z / sp->height_res + lidar_to_ground)); /* :281 */
        if (x < 0 || x >= M || y < 0 || y >= M || layer < 0 || layer >= layers ||
            p->label == 0)                                       /* :284-287 */
            continue;
        out[((size_t)layer * M + x) * M + y] = 255;              /* :289-291 */
    }
}

/* BatchMultiBevGen.cpp:336-356 */
void oracle_single_bev(const oracle_point_t *cloud, size_t n, float interval,
                       uint8_t *out)
{
    const int max_range = 112;                                   /* :336 */
    const int M = cvtt_f32_to_i32((float)(max_range * 2) / interval); /* :337 */
    const float lidar_to_ground = 2.0f;                          /* :338 */
    memset(out, 0, (size_t)M * M);                               /* :340 */
    for (size_t i = 0; i < n; ++i) {
        const oracle_point_t *p = &cloud[i];
        int x = bev_bin(p->x, max_range, interval);              /* :343 */
        int y = bev_bin(p->y, max_range, interval);              /* :344 */
        int height = cvtt_f64_to_i32((double)(p->z + lidar_to_ground) * 4.0); /* :345 */
        if (height < 0) height = 0;                              /* :346 */
        if (height > 255) height = 255;
        if (x < 0 || x >= M || y < 0 || y >= M || p->label == 0) /* :349-351 */
            continue;
        if (out[(size_t)x * M + y] < height)                     /* :353-355 */
            out[(size_t)x * M + y] = (uint8_t)height;
    }
}

/* BatchMultiBevGen.cpp:735-747 */
void oracle_process_frame(const oracle_sensor_t *sp, const oracle_point_t *in,
                          size_t n_in, oracle_point_t *ordered, int8_t *ground_mat,
                          uint8_t *multi, uint8_t *single)
{
    const size_t S = (size_t)sp->n_scan * (size_t)sp->horizon_scan;
    int8_t *gm = ground_mat ? ground_mat : (int8_t *)malloc(S);
    oracle_order_cloud(sp, in, n_in, ordered);        /* :735 */
    oracle_mark_ground(sp, ordered, gm, NULL);        /* :736 */
    oracle_multi_bev(sp, ordered, S, 1.0f, multi);    /* :746 */
    oracle_single_bev(ordered, S, 1.0f, single);      /* :747 */
    if (!ground_mat) free(gm);
}

/* The file outputs inside the reference's timed region (BatchMultiBevGen.cpp:732-752) other than the PNGs:
 * the .bin of computeAndSaveMultiBev (:294-314: layer by layer, row by row, 224 bytes per write through one ofstream)
 * and the .csv of computeAndSaveSingleBev (:365-372: cv::format(mat, FMT_CSV) streamed into an ofstream).  The CSV
 * framing is OpenCV's, restated FROM MEMORY of modules/core/src/out.cpp (u8 as "%3d", values joined by ", ", rows by
 * "\n", one trailing "\n"): PARITY UNPINNED.  Used by bench.py's cpu_baseline "timed_region"
 * leg only.  Returns 0 on success. */
int oracle_save_bin_csv(const uint8_t *multi, const uint8_t *single, int M, int layers,
                        const char *bin_path, const char *csv_path)
{
    FILE *fb = fopen(bin_path, "wb");
    if (!fb) return -1;
    for (int l = 0; l < layers; ++l)
        for (int r = 0; r < M; ++r)
            if (fwrite(multi + ((size_t)l * M + r) * M, 1, (size_t)M, fb) != (size_t)M) { fclose(fb); return -1; }
    if (fclose(fb) != 0) return -1;
    FILE *fc = fopen(csv_path, "w");
    if (!fc) return -1;
    char *line = (char *)malloc((size_t)M * 5 + 2);
    if (!line) { fclose(fc); return -1; }
    for (int r = 0; r < M; ++r) {
        size_t n = 0;
        for (int c = 0; c < M; ++c) {
            n += (size_t)snprintf(line + n, 6, "%3d", (int)single[(size_t)r * M + c]);
            if (c + 1 < M) { line[n++] = ','; line[n++] = ' '; }
        }
        line[n++] = '\n';
        if (fwrite(line, 1, n, fc) != n) { free(line); fclose(fc); return -1; }
    }
    free(line);
    return fclose(fc) == 0 ? 0 : -1;
}

/* BatchCloudManip.cpp:201-225 / CloudManip.cpp:79-99 */
void oracle_float_bev(const oracle_point_t *cloud, size_t n, float interval,
                      int skip_label0, float *out)
{
    const int max_range = 100;                                   /* :209 / :81 */
    const int M = cvtt_f32_to_i32((float)(max_range * 2) / interval + 1); /* :210 / :82 */
    for (size_t k = 0; k < (size_t)M * M; ++k) out[k] = 0.0f;    /* :211 / :83 */
    for (size_t i = 0; i < n; ++i) {
        const oracle_point_t *p = &cloud[i];
        int x = bev_bin(p->x, max_range, interval);              /* :215 / :85 */
        int y = bev_bin(p->y, max_range, interval);              /* :216 / :86 */
        if (x < 0 || x >= M || y < 0 || y >= M) continue;        /* :218 / :88 */
        if (skip_label0 && p->label == 0) continue;              /* :218 */
        float h = p->z + 2.0f;                                   /* :222 / :92 */
        if (h > out[(size_t)x * M + y]) out[(size_t)x * M + y] = h;
    }
}

/* CloudManip.cpp:119-128 (see bev_oracle.h) */
void oracle_yaw_translate_matrix(float tx, float ty, float tz, float yaw_deg, float m[12])
{
    const float theta = (float)((double)(yaw_deg / 180.0f) * M_PI); /* :124: float / float, then double * M_PI, stored to float */
    const float s = sinf(theta), c = cosf(theta);                   /* Eigen: sin / cos of a float angle */
    const float one_minus_c = 1.0f - c;
    const float ax = 0.0f, ay = 0.0f, az = 1.0f;                    /* Vector3f::UnitZ() */
    const float sx = s * ax, sy = s * ay, sz = s * az;              /* sin_axis */
    const float cx = one_minus_c * ax, cy = one_minus_c * ay, cz = one_minus_c * az; /* cos1_axis */
    float tmp;
    float r[9];
    tmp = cx * ay; r[1] = tmp - sz; r[3] = tmp + sz;
    tmp = cx * az; r[2] = tmp + sy; r[6] = tmp - sy;
    tmp = cy * az; r[5] = tmp - sx; r[7] = tmp + sx;
    r[0] = cx * ax + c; r[4] = cy * ay + c; r[8] = cz * az + c;
    /* T.rotate(R): linear = Identity * R (products with exact 0 / 1: the sums below are what a 3 x 3 product evaluates) */
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            float acc = (i == 0 ? 1.0f : 0.0f) * r[j];
            acc = acc + (i == 1 ? 1.0f : 0.0f) * r[3 + j];
            acc = acc + (i == 2 ? 1.0f : 0.0f) * r[6 + j];
            m[4 * i + j] = acc;
        }
    m[3] = tx; m[7] = ty; m[11] = tz;
}

void oracle_transform_cloud(const oracle_point_t *cloud, size_t n, const float m[12], oracle_point_t *out)
{
    for (size_t i = 0; i < n; ++i) {
        oracle_point_t p = cloud[i];
        const float x = p.x, y = p.y, z = p.z;
        p.x = m[0] * x + (m[1] * y + (m[2] * z + m[3]));
        p.y = m[4] * x + (m[5] * y + (m[6] * z + m[7]));
        p.z = m[8] * x + (m[9] * y + (m[10] * z + m[11]));
        out[i] = p;
    }
}

/* x86-64 static_cast<uint16_t>(float): cvttss2si, low 16 bits */
static uint16_t cvtt_f32_to_u16(float v) { return (uint16_t)(uint32_t)cvtt_f32_to_i32(v); }

/* MulranPointCloudSelect.cpp:112-130 (float overloads of atan2 / round: the file is compiled with
 * "using namespace std") */
void oracle_project_mulran(const float *xyzi, size_t n, oracle_point_t *out)
{
    for (size_t k = 0; k < n; ++k) {
        oracle_point_t p;
        memset(&p, 0, sizeof p);
        p.x = xyzi[4 * k];           /* :116-119 */
        p.y = xyzi[4 * k + 1];
        p.z = xyzi[4 * k + 2];
        p.intensity = xyzi[4 * k + 3];
        p.row = (uint16_t)(k % 64);  /* :120 */
        float az = (float)((double)atan2f(p.y, p.x) / M_PI * 180.0f); /* :121 */
        if (az > 360.0f) az = az - 360.0f;       /* :122 */
        else if (az < 0.0f) az = az + 360.0f;    /* :123 */
        p.col = cvtt_f32_to_u16(roundf(az / 360.0f * 1024)); /* :125 */
        p.label = -2;                /* :126 */
        out[k] = p;
    }
}

/* OxfordPointCloudSelect.cpp:172-218 */
void oracle_project_oxford(const float *xyzi, size_t n, oracle_point_t *out)
{
    for (size_t i = 0; i < n; ++i) {
        oracle_point_t p;
        memset(&p, 0, sizeof p);
        p.x = -xyzi[i];               /* :178, :203 the lidar is mounted upside-down */
        p.y = xyzi[n + i];            /* :184 */
        p.z = -xyzi[2 * n + i];       /* :189, :204 */
        p.intensity = xyzi[3 * n + i];
        p.label = -2;                 /* :206 */
        float elev = (float)((double)atan2f(p.z, sqrtf(p.x * p.x + p.y * p.y)) / M_PI * 180.0f); /* :208 */
        int row = cvtt_f64_to_i32(round(((double)(-elev) + 10.67) / 1.3335)); /* :209 */
        if (row < 0) row = 0;         /* :210 std::min(31, std::max(0, row)) */
        if (row > 31) row = 31;
        p.row = (uint16_t)row;
        float az = (float)((double)atan2f(p.y, p.x) / M_PI * 180.0f); /* :213 */
        if (az > 360.0f) az = az - 360.0f;
        else if (az < 0.0f) az = az + 360.0f;
        p.col = cvtt_f32_to_u16(roundf(az / 360.0f * 1056)); /* :216 */
        if (p.col >= 1056) p.col -= 1056; /* :217 */
        out[i] = p;
    }
}

/* KittiPointCloudSelect.cpp:186-243 (the part of extractPointCloud after the file has been read into `cloud`):
 * the ring index is a counter of azimuth zero crossings, a new ring being accepted only after more than
 * Horizon_SCAN * 0.60f points; N_SCAN = 64 and Horizon_SCAN = 2083 are file constants (:148-149).
 * out: 64 * 2083 points, empty slots all-zero (value-initialised by resize, :207).
 * Defined here where the reference is undefined: n == 0 (the reference reads azimuth_angle[0] of an empty
 * vector) gives an all-zero cloud; a column still outside [0, Horizon_SCAN) after the single wrap at :229-233
 * (only a NaN azimuth gets there; the reference then writes out of bounds) drops the point. */
static float kitti_make_angle_semi_positive(float a) /* :137-146 */
{
    if (a >= 360.0f) return a - 360.0f;
    else if (a < 0) return a + 360.0f;
    else return a;
}

void oracle_project_kitti(const float *xyzi, size_t n, oracle_point_t *out)
{
    const int N_SCAN = 64, Horizon_SCAN = 2083; /* :148-149 */
    memset(out, 0, (size_t)N_SCAN * Horizon_SCAN * sizeof *out); /* :206-207 */
    if (n == 0) return;
    float *azimuth_angle = (float *)malloc(n * sizeof(float));
    for (size_t i = 0; i < n; ++i)  /* :190-193 */
        azimuth_angle[i] = (float)((double)atan2f(xyzi[4 * i + 1], xyzi[4 * i]) / M_PI * 180.0f);
    int32_t ring_idx = -1;          /* :195-203 */
    if (azimuth_angle[0] > 0) ring_idx = 0;
    int num_points_on_this_ring = 0; /* :210 */
    for (size_t i = 1; i < n; ++i) { /* :212 */
        if (azimuth_angle[i - 1] <= 0 && azimuth_angle[i] > 0) { /* :214-222 */
            if (ring_idx == -1) {
                ring_idx = 0;
                num_points_on_this_ring = 0;
            } else if (num_points_on_this_ring > Horizon_SCAN * 0.60f) {
                ring_idx++;
                num_points_on_this_ring = 0;
            }
        }
        const float this_azimuth = kitti_make_angle_semi_positive(azimuth_angle[i]); /* :225 */
        int col_idx = cvtt_f64_to_i32(round((double)this_azimuth / (360.0 / Horizon_SCAN))); /* :226 */
        if (ring_idx >= 0 && ring_idx < N_SCAN) { /* :228 */
            if (col_idx >= Horizon_SCAN) col_idx = col_idx - Horizon_SCAN; /* :229-233 */
            else if (col_idx < 0) col_idx = col_idx + Horizon_SCAN;
            if (col_idx >= 0 && col_idx < Horizon_SCAN) { /* see the header comment */
                oracle_point_t p;
                memset(&p, 0, sizeof p);
                p.x = xyzi[4 * i];
                p.y = xyzi[4 * i + 1];
                p.z = xyzi[4 * i + 2];
                p.row = (uint16_t)ring_idx;   /* :235 */
                p.col = (uint16_t)col_idx;    /* :236 */
                p.label = -2;                 /* :237 */
                p.intensity = -1;             /* :238 */
                out[(size_t)ring_idx * Horizon_SCAN + col_idx] = p; /* :240 */
            }
        }
        num_points_on_this_ring++; /* :242 */
    }
    free(azimuth_angle);
}
