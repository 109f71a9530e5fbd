/*
 * bev_oracle.h — CPU restatement of the batch_multi_bev_gen hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * build, load or call it, and only as the checker / the timed CPU baseline.
 * The product library (libbev_mi355x.so) never links or falls back to it.
 *
 * PARITY UNPINNED.  The reference ships no tests, golden vectors or fixtures
 * for this path and cannot be built in this image
 * (PCL / OpenCV / Eigen / fmt / Boost are absent and may not be stubbed), so
 * this restatement is anchored on the reference SOURCE TEXT only — each
 * function cites the lines it follows — plus hand-derived known-answer cases
 * in tests/.  It has not been checked against outputs of the real binary.
 *
 * Third-party behaviour restated here (not under /root/reference):
 *   - glibc libm atan2f / sqrtf / round / floor — called directly, so the
 *     oracle inherits whatever libm the host has (glibc 2.35 in this image);
 *   - OpenCV: cv::Mat::zeros / 0.01*ones / MatExpr divide on CV_32F are
 *     restated as 0.0f, (float)(1.0f*0.01) and IEEE float division;
 *   - x86-64 SSE2 float->int conversion (cvttss2si / cvttsd2si): out-of-range
 *     and NaN inputs give INT_MIN ("integer indefinite"); the reference casts
 *     without a range check, so that is what its binary does.
 *
 * All file:line citations are relative to the reference tree.
 */
#ifndef BEV_ORACLE_H
#define BEV_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* pcl::PointXYZIRCT in memory, BatchMultiBevGen.h:43-54 (sizeof == 32). */
typedef struct oracle_point {
    float x, y, z, pad0;
    float intensity;
    uint16_t row, col;
    uint32_t t;
    int16_t label;
    uint16_t pad1;
} oracle_point_t;

/* SensorParams, include/Utility.h:30-36. */
typedef struct oracle_sensor {
    int horizon_scan;
    int n_scan;
    int ground_upper_scan;
    float height_res;
} oracle_sensor_t;

#define ORACLE_GRID_ROWS 75 /* BatchMultiBevGen.cpp:25 */
#define ORACLE_GRID_COLS 50 /* BatchMultiBevGen.cpp:26 */

/* getSensorParams, src/Utility.cpp:92-124. kind: 0 HDL_32E, 1 HDL_64E, 2 OS1_64.
 * Returns 0, or -1 for an unknown kind. */
int oracle_sensor_params(int kind, oracle_sensor_t *out);

/* getBelongingGrid, BatchMultiBevGen.h:73-99. */
void oracle_belonging_grid(float x, float y, int *sector_row, int *sector_col);

/* getOrderedCloud, BatchMultiBevGen.cpp:94-117. out: S points, fully written. */
void oracle_order_cloud(const oracle_sensor_t *sp, const oracle_point_t *in,
                        size_t n_in, oracle_point_t *out);

/* The angle test of markGroundPoints phase A, BatchMultiBevGen.cpp:169-179,
 * for one (upper - lower) difference vector. Returns 1 if "ground". */
int oracle_angle_is_ground(float diff_x, float diff_y, float diff_z);
/* The same test under the other overload resolution the reference's source admits (double sqrt / atan2 / fabs;
 * BatchMultiBevGen.h:38, .cpp:173,179).  NOT the adopted reading: it exists so that tests can MEASURE how many slots
 * and labels the choice changes (tests/test_oracle_overloads.py). */
int oracle_angle_is_ground_f64(float diff_x, float diff_y, float diff_z);
#define ORACLE_ANGLE_F32 0
#define ORACLE_ANGLE_F64 1

/* markGroundPoints, BatchMultiBevGen.cpp:119-252.  cloud: S ordered points,
 * labels rewritten in place.  ground_mat: S int8 (required).  avg_out: NULL or
 * 75*50 floats = ground_grid_avg_heights after the divide at :210. */
void oracle_mark_ground(const oracle_sensor_t *sp, oracle_point_t *cloud,
                        int8_t *ground_mat, float *avg_out);
/* markGroundPoints with the angle test of the chosen overload reading (ORACLE_ANGLE_F32 = oracle_mark_ground). */
void oracle_mark_ground_variant(const oracle_sensor_t *sp, oracle_point_t *cloud,
                                int8_t *ground_mat, float *avg_out, int angle_variant);

/* Raster part of computeAndSaveMultiBev, BatchMultiBevGen.cpp:266-292.
 * out: 24 * M * M bytes laid out as the .bin file (:307-314), M = 224/interval. */
void oracle_multi_bev(const oracle_sensor_t *sp, const oracle_point_t *cloud,
                      size_t n, float interval, uint8_t *out);

/* Raster part of computeAndSaveSingleBev, BatchMultiBevGen.cpp:336-356.
 * out: M * M bytes, row index = x. */
void oracle_single_bev(const oracle_point_t *cloud, size_t n, float interval,
                       uint8_t *out);

/* Whole per-frame body of main(), BatchMultiBevGen.cpp:735-747 (no file I/O).
 * ground_mat may be NULL. */
void oracle_process_frame(const oracle_sensor_t *sp, const oracle_point_t *in,
                          size_t n_in, oracle_point_t *ordered, int8_t *ground_mat,
                          uint8_t *multi, uint8_t *single);

/* .bin + .csv writes of the reference's timed region (BatchMultiBevGen.cpp:294-314, :365-372), CSV framing from
 * memory of OpenCV (PARITY UNPINNED); bench.py's cpu_baseline "timed_region" leg. */
int oracle_save_bin_csv(const uint8_t *multi, const uint8_t *single, int M, int layers,
                        const char *bin_path, const char *csv_path);

/* Float max-height BEV of batch_cloud_manip / cloud_manip (saveAsMat,
 * BatchCloudManip.cpp:201-225, CloudManip.cpp:79-99).  skip_label0 = 1 for the
 * batch variant (:218).  out: M*M floats, M = 200/interval + 1. */
void oracle_float_bev(const oracle_point_t *cloud, size_t n, float interval,
                      int skip_label0, float *out);

/* The rigid transform cloud_manip applies before its second saveAsMat (CloudManip.cpp:119-128):
 *   Eigen::Affine3f T = Identity; T.translation() << tx, ty, tz;
 *   theta = stof(argv[5]) / 180.0f * M_PI;  T.rotate(AngleAxisf(theta, UnitZ()));  pcl::transformPointCloud(in, out, T)
 * oracle_yaw_translate_matrix: the 3 x 4 row-major [R | t] that code builds (Eigen 3.3 AngleAxis::toRotationMatrix:
 * the diagonal is (1 - c) * axis^2 + c, so m22 = (1 - c) + c, not a literal 1).
 * oracle_transform_cloud: pcl::detail::Transformer<float>::se3 (PCL >= 1.10, SSE build):
 *   out.xyz = m.col0 * x + (m.col1 * y + (m.col2 * z + m.col3)),  every other field copied.
 * Eigen and PCL are not under /root/reference: both are restated from their published sources (PARITY UNPINNED). */
void oracle_yaw_translate_matrix(float tx, float ty, float tz, float yaw_deg, float m[12]);
void oracle_transform_cloud(const oracle_point_t *cloud, size_t n, const float m[12], oracle_point_t *out);

/* Range-image projection of the keyframe selectors ("next" row N3).  The reference declares the point
 * without initialising it, so t and the padding are indeterminate there; the oracle writes 0.
 * MulRan (MulranPointCloudSelect.cpp:112-130): xyzi = n * (x, y, z, intensity) interleaved.
 * Oxford (OxfordPointCloudSelect.cpp:172-218): xyzi = x[n] y[n] z[n] intensity[n] (four planes). */
void oracle_project_mulran(const float *xyzi, size_t n, oracle_point_t *out);
void oracle_project_oxford(const float *xyzi, size_t n, oracle_point_t *out);
/* KITTI (KittiPointCloudSelect.cpp:186-243): xyzi interleaved like MulRan; the ring index is a counter of
 * azimuth zero crossings.  out: the structured 64 * 2083 cloud (empty slots all-zero, real points get
 * intensity = -1, label = -2). */
void oracle_project_kitti(const float *xyzi, size_t n, oracle_point_t *out);

#ifdef __cplusplus
}
#endif
#endif
