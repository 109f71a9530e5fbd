/*
 * bev_mi355x.h — C ABI of the MI355X-native batch_multi_bev_gen hot path.
 *
 * This is the drop-in boundary ("lower face").  The reference
 * (soytony/Point-Cloud-Preprocessing-Tools) has no FFI layer of its own: its
 * hot path is four C++ free functions that read file-scope globals.  Every
 * entry point below names the reference interface it replaces (file:line are
 * relative to the reference tree).  Only POD types and plain pointers cross
 * this boundary; no C++/torch types.
 *
 * Conventions
 *   - return 0 (BEV_OK) on success, a negative bev_status_t otherwise;
 *     nothing throws, nothing calls exit().
 *   - the caller owns every buffer it passes; the context owns its device
 *     workspace, stream and events.
 *   - one context per GPU, used by one host thread at a time.
 *   - there is NO CPU fallback behind this ABI: if no HIP device is usable,
 *     bev_create() fails with BEV_ERR_NO_DEVICE.
 */
#ifndef BEV_MI355X_H
#define BEV_MI355X_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BEV_ABI_VERSION 1

/* Ground-height grid of getBelongingGrid (BatchMultiBevGen.h:73-99,
 * BatchMultiBevGen.cpp:25-26): 75 x 50 cells of 2 m. */
#define BEV_GROUND_GRID_ROWS 75
#define BEV_GROUND_GRID_COLS 50
#define BEV_GROUND_GRID_CELLS (BEV_GROUND_GRID_ROWS * BEV_GROUND_GRID_COLS)

typedef enum bev_status {
    BEV_OK = 0,
    BEV_ERR_INVALID_ARG = -1,
    BEV_ERR_NO_DEVICE = -2,     /* no usable HIP device: there is no CPU path */
    BEV_ERR_HIP = -3,           /* a HIP runtime call failed (see bev_last_error) */
    BEV_ERR_OOM = -4,
    BEV_ERR_UNSUPPORTED = -5,   /* parameter combination outside the built kernels */
    BEV_ERR_TOO_LARGE = -6      /* n_frames / n_points above what bev_create sized */
} bev_status_t;

/* In-memory layout of pcl::PointXYZIRCT (BatchMultiBevGen.h:43-54): 32 bytes,
 * 16-byte aligned; x@0 y@4 z@8 pad@12 intensity@16 row@20 col@22 t@24 label@28. */
typedef struct bev_point {
    float x, y, z, _pad0;
    float intensity;
    uint16_t row, col;
    uint32_t t;
    int16_t label;
    uint16_t _pad1;
} bev_point_t;

/* SensorParams (include/Utility.h:30-36) plus the constants the reference
 * hard-codes in computeAndSaveMultiBev / computeAndSaveSingleBev
 * (BatchMultiBevGen.cpp:266-269,336-338) and main (:738). */
typedef struct bev_params {
    int32_t n_scan;             /* SensorParams::N_SCAN */
    int32_t horizon_scan;       /* SensorParams::Horizon_SCAN */
    int32_t ground_upper_scan;  /* SensorParams::GROUND_UPPER_SCAN */
    float height_res;           /* SensorParams::HEIGHT_RES */
    float interval;             /* 1.0f  (main :738); others if M = 2 * max_range / interval is a multiple of 16, 16..512 */
    int32_t max_range;          /* 112   (:266,:336) */
    int32_t n_layers;           /* 24    (:268,:271) */
    float lidar_to_ground;      /* 2.0f  (:269,:338) */
} bev_params_t;

typedef struct bev_ctx bev_ctx_t;

/* parseSensorType + getSensorParams (src/Utility.cpp:72-124) and the BEV
 * defaults above.  `sensor` is matched by substring like the reference
 * ("HDL_32E", "HDL_64E", "OS1_64").  Unknown sensor -> BEV_ERR_INVALID_ARG
 * (the reference leaves the struct uninitialised). */
int bev_params_for_sensor(const char *sensor, bev_params_t *out);

/* Slots of the ordered range image: n_scan * horizon_scan. */
size_t bev_num_slots(const bev_params_t *p);
/* Bytes of one multi-layer BEV (.bin payload, BatchMultiBevGen.cpp:307-314)
 * and of one single-layer BEV (cv::Mat single_bev, :340). */
size_t bev_multi_bytes(const bev_params_t *p);
size_t bev_single_bytes(const bev_params_t *p);

/* Replaces the global state the reference sets up in main()
 * (sensor_params_, four_neighbor_iterator_, BatchMultiBevGen.cpp:29,37,712-719).
 * device        : HIP device ordinal (>= 0).
 * max_batch     : frames processed per internal sub-batch (workspace is sized
 *                 for this many; any n_frames is accepted later and looped).
 * max_points    : largest per-frame input point count that will be passed. */
int bev_create(bev_ctx_t **ctx, int device, const bev_params_t *p,
               int max_batch, size_t max_points);
void bev_destroy(bev_ctx_t *ctx);

const char *bev_strerror(int status);
/* Text of the last HIP error seen by this context ("" if none). */
const char *bev_last_error(const bev_ctx_t *ctx);

/* ---- whole hot path, host buffers ------------------------------------
 * One call = the body of the per-file loop of main()
 * (BatchMultiBevGen.cpp:727-757) for n_frames clouds, minus file I/O:
 *   getOrderedCloud (:94-117) -> markGroundPoints (:119-252)
 *   -> computeAndSaveMultiBev raster (:266-292)
 *   -> computeAndSaveSingleBev raster (:336-356).
 * pts[f]         : n_pts[f] unordered input points of frame f.
 * ordered_out[f] : S points; labelled ordered cloud (what savePCDFileBinary
 *                  writes at :756).
 * multi_out[f]   : n_layers*M*M bytes, layer-major then row(x)-major — the
 *                  exact .bin payload.
 * single_out[f]  : M*M bytes, row(x)-major (cv::Mat single_bev).
 * ground_mat_out : optional (NULL or per-frame NULL allowed); S int8 values,
 *                  the final cv::Mat ground_mat of markGroundPoints. */
int bev_process_batch(bev_ctx_t *ctx, int n_frames,
                      const bev_point_t *const *pts, const uint32_t *n_pts,
                      bev_point_t *const *ordered_out,
                      uint8_t *const *multi_out,
                      uint8_t *const *single_out,
                      int8_t *const *ground_mat_out);

/* ---- whole hot path, device-resident ----------------------------------
 * Same computation, every data pointer is DEVICE memory; nothing crosses
 * PCIe except a few hundred bytes of launch metadata.
 * d_pts        : all frames' input points, packed; frame f occupies
 *                [h_offsets[f], h_offsets[f+1]) (element offsets, HOST array
 *                of n_frames+1 entries).
 * d_ordered    : n_frames * S points.
 * d_multi      : n_frames * bev_multi_bytes.
 * d_single     : n_frames * bev_single_bytes.
 * d_ground_mat : NULL or n_frames * S int8.
 * Asynchronous: call bev_synchronize() before reading results.  The stages of a
 * sub-batch ride in consecutive launches beside the stages of its neighbours
 * (see bev_set_lanes); the last stages of a call's last sub-batches are
 * launched by the NEXT call to this function — calls that follow each other
 * without a synchronisation keep the device busy across the call boundary —
 * or by bev_synchronize() (and by every other entry point of the context).
 * A bare hipDeviceSynchronize() is therefore NOT enough: bev_synchronize()
 * launches what is pending, then waits.  The buffers of a call must stay
 * valid until then.  Work the caller has queued on the DEFAULT stream before
 * the call (an upload or a fill of these buffers) is waited for on the device;
 * work on other streams of the caller's is the caller's to synchronise with:
 * the library's streams are non-blocking. */
int bev_process_device_resident(bev_ctx_t *ctx, int n_frames,
                                const bev_point_t *d_pts,
                                const uint64_t *h_offsets,
                                bev_point_t *d_ordered,
                                uint8_t *d_multi,
                                uint8_t *d_single,
                                int8_t *d_ground_mat);
int bev_synchronize(bev_ctx_t *ctx);

/* Page-locked host memory for the buffers handed to bev_process_batch (optional): copies from / to such memory
 * are DMA transfers without intermediate staging by the runtime.  Any host memory works (on the MI355X boxes
 * measured, pageable buffers reached 6.3 k frames/s and page-locked ones 6.5 k: the link, not the staging, is
 * the limit).  hipHostMalloc / hipHostFree. */
int bev_host_alloc(void **out, size_t bytes);
int bev_host_free(void *p);

/* ---- per-function entry points (host buffers, one cloud) ---------------
 * These let the reference-named C++ free functions be re-implemented as thin
 * callers, one ABI call each. */

/* getOrderedCloud (BatchMultiBevGen.cpp:94-117). out: S points. */
int bev_order_cloud(bev_ctx_t *ctx, const bev_point_t *pts, uint32_t n_pts,
                    bev_point_t *ordered_out);
/* markGroundPoints (BatchMultiBevGen.cpp:119-252). `ordered` holds S points;
 * labels are rewritten in place. ground_mat_out: NULL or S int8. */
int bev_mark_ground(bev_ctx_t *ctx, bev_point_t *ordered, int8_t *ground_mat_out);
/* Raster part of computeAndSaveMultiBev (:266-292) for any cloud of n points
 * (points with label == 0 are skipped). out: bev_multi_bytes. */
int bev_multi_bev(bev_ctx_t *ctx, const bev_point_t *cloud, uint32_t n,
                  uint8_t *multi_out);
/* Raster part of computeAndSaveSingleBev (:336-356). out: bev_single_bytes. */
int bev_single_bev(bev_ctx_t *ctx, const bev_point_t *cloud, uint32_t n,
                   uint8_t *single_out);

/* Float max-height BEV of the older tools: saveAsMat of batch_cloud_manip
 * (BatchCloudManip.cpp:201-225, skip_label0 = 1) and cloud_manip
 * (CloudManip.cpp:79-99, skip_label0 = 0): grid M x M, M = 200/interval + 1,
 * cell = max(0, max over points of z + 2.0f), float32.  out: M*M floats,
 * row index = x.  interval must give M <= 1024.  Bit-exact (a max has no
 * rounding), i.e. well inside the 1e-5 the float height channel is allowed. */
int bev_float_bev(bev_ctx_t *ctx, const bev_point_t *cloud, uint32_t n, float interval,
                  int skip_label0, float *out);
size_t bev_float_bev_size(float interval); /* M for a given interval (0 if unsupported) */

/* The rigid transform of cloud_manip (CloudManip.cpp:119-128, pcl::transformPointCloud with an Eigen::Affine3f):
 * out[i] = cloud[i] with xyz replaced by  col0 * x + (col1 * y + (col2 * z + col3))  of the 3 x 4 row-major
 * matrix m (12 floats) — the association of pcl::detail::Transformer<float>::se3 (PCL >= 1.10); every other field
 * is copied.  bev_yaw_translate_matrix builds the matrix the tool builds from its arguments
 * (translation tx ty tz, then rotate(AngleAxisf(yaw_deg / 180.0f * M_PI, UnitZ()))); host only, no device needed.
 * Eigen / PCL are third-party: restated from their published sources (parity unpinned). */
int bev_transform_cloud(bev_ctx_t *ctx, const bev_point_t *cloud, uint32_t n, const float *m /* 12 */,
                        bev_point_t *out);
void bev_yaw_translate_matrix(float tx, float ty, float tz, float yaw_deg, float *m /* 12 */);

/* Range-image projection of raw XYZI returns — the selectors' row / col assignment ("polar binning"):
 *   BEV_PROJECT_MULRAN_OS1_64   extractPointCloud, MulranPointCloudSelect.cpp:112-130:
 *                               xyzi = n * (x, y, z, intensity); row = k % 64, col from the azimuth (0..1024)
 *   BEV_PROJECT_OXFORD_HDL_32E  extractPointCloud, OxfordPointCloudSelect.cpp:172-218:
 *                               xyzi = x[n] y[n] z[n] intensity[n]; x and z are negated, row from the
 *                               elevation (0..31), col from the azimuth (0..1055)
 *   BEV_PROJECT_KITTI_HDL_64E   extractPointCloud, KittiPointCloudSelect.cpp:186-243 (after the file is read):
 *                               xyzi = n * (x, y, z, intensity); row = a counter of azimuth zero crossings (a new
 *                               ring needs more than 2083 * 0.60f points on the current one), col from the azimuth
 *                               (0..2082); out is the reference's STRUCTURED cloud of 64 * 2083 points, last
 *                               writer per slot, real points with intensity = -1, empty slots all-zero.  Point 0
 *                               is never stored (the reference's loop starts at 1).  n = 0 and NaN azimuths are
 *                               undefined behaviour in the reference; here: all-zero cloud / point dropped.
 * out: bev_project_out_points(kind, n) points (n, or 64 * 2083 for KITTI) with label = -2; t and padding are 0
 * (the reference leaves them uninitialised).  atan2f is evaluated on the device by a restatement of glibc's
 * algorithm (bit-identical, csrc/bev_libm.h). */
#define BEV_PROJECT_MULRAN_OS1_64 0
#define BEV_PROJECT_OXFORD_HDL_32E 1
#define BEV_PROJECT_KITTI_HDL_64E 2
int bev_project_xyzi(bev_ctx_t *ctx, int kind, const float *xyzi, uint32_t n, bev_point_t *out);
size_t bev_project_out_points(int kind, uint32_t n); /* 0 for an unknown kind */

/* ---- layout hint -------------------------------------------------------
 * What the caller knows about how its clouds are laid out, so that the library need not look (k_probe reads every 63rd
 * record of a frame to find out: 0.36 MB and 0.07 us of an HDL_64E frame).  Sticky per context; applies to frames of
 * exactly S = n_scan * horizon_scan records, every other frame is probed as ever:
 *   BEV_LAYOUT_STRUCTURED    what kitti_point_cloud_select writes (KittiPointCloudSelect.cpp:206-207,240) and what
 *                            bev_project_xyzi(BEV_PROJECT_KITTI_HDL_64E) returns (that call sets this hint by itself):
 *                            record i is the point of slot i, or all-zero;
 *   BEV_LAYOUT_FIRING_ORDER  the plain sweep in firing order (BASELINE config 3): record k is beam k % n_scan of firing
 *                            k / n_scan, its column the firing's number + 0 .. 8 or >= horizon_scan.  (What
 *                            mulran_point_cloud_select writes for REAL sweeps — MulranPointCloudSelect.cpp:112-130: any
 *                            start azimuth, either direction, staggered beams, no-return records — needs the sweep's
 *                            direction and a base column per row, which k_probe measures: no hint for it.)
 *   BEV_LAYOUT_UNKNOWN       (default) the library looks.
 * The hint is a guess like the library's own: the walk checks every record it reads, a frame that is not what the hint
 * said is done again the general way — a wrong hint costs time, never results (tests/test_gpu_structured.py). */
#define BEV_LAYOUT_UNKNOWN 0
#define BEV_LAYOUT_STRUCTURED 3
#define BEV_LAYOUT_FIRING_ORDER 4
int bev_set_layout_hint(bev_ctx_t *ctx, int layout);

/* ---- measurement ------------------------------------------------------- */
/* Sub-batches (max_batch frames) run as FUSED launches: one launch holds the column walk of sub-batch t and, as further
 * workgroups of the same grid, phase B of an earlier sub-batch, phase C of the one before that and the rasters of the one
 * before that; the launches alternate between two streams of equal priority (a sub-batch's stages stay on its stream),
 * over eight workspace sets; every kernel has one workgroup shape (256 threads, a quarter of a CU's LDS and registers).
 * bev_set_lanes(ctx, 1) makes every kernel a launch of its own, back to back (clean per-kernel durations for
 * profiling; the same device code); bev_set_lanes(ctx, n > 1) switches back.  Returns the number of workspace sets now
 * in rotation (1 or 8), or a negative status.  Environment of bev_create: BEV_LANES=1 starts serial; BEV_STREAM=0
 * sends every frame through the general path (order scan + gather walk); BEV_MODE_TTL=n: sub-batches for which a
 * layout's walk stays launched after the workspace set last saw the layout (default 8); BEV_CODE_CAP=n: entries of a
 * raster-band code list (tests); BEV_STAGE_STREAMS=n (1 .. 4, default 2): streams the fused launches take turns on, with
 * 4 n workspace sets.  These are all the knobs the library reads. */
int bev_set_lanes(bev_ctx_t *ctx, int n);

#define BEV_MAX_KERNELS 16
typedef struct bev_kernel_stat {
    const char *name;      /* kernel symbol as rocprofv3 prints it */
    uint64_t launches;     /* launches timed since the last reset */
    double total_ms;       /* sum of HIP-event durations on the ctx stream */
    uint64_t frames;       /* frames those launches covered */
} bev_kernel_stat_t;
/* When enabled, every kernel launch of bev_process_* is bracketed by HIP
 * events recorded on the context's own stream. */
int bev_profile_enable(bev_ctx_t *ctx, int on);
int bev_profile_reset(bev_ctx_t *ctx);
/* Synchronises, then fills up to `cap` entries; returns the number of kernels
 * (or a negative status). */
int bev_profile_get(bev_ctx_t *ctx, bev_kernel_stat_t *out, int cap);

/* ---- test hooks (used by tests/ only; not part of the reference surface) - */
/* Per-cell average ground heights of the LAST sub-batch processed (bev_process_batch works in chunks of
 * max(1, max_batch / 2) frames, bev_process_device_resident in sub-batches of max_batch): copies
 * n_frames * 3750 floats (ground_grid_avg_heights after BatchMultiBevGen.cpp:210), first_frame counted
 * from the start of that sub-batch. */
int bev_debug_get_cell_avg(bev_ctx_t *ctx, int first_frame, int n_frames, float *out);
/* Test hook: how the frames of the LAST sub-batch reached their slots (getOrderedCloud, BatchMultiBevGen.cpp:94-117).
 * out[4 * i .. 4 * i + 3] = { T, mode, consumed, failed } of frame first_frame + i.  mode 0 = order scan over all points
 * (general path); 1 = the first T points were read in place (sorted prefix, verified: consumed == T, failed == 0);
 * 2 = read in place, verification failed, done again the general way; 3 = a structured cloud of S records (record i =
 * slot i's point or all-zero; KittiPointCloudSelect.cpp:206-207,240) read in place (consumed == T == S; failed bit 1: an
 * all-zero record after the first was seen, bit 2: k_probe expected one); 4 = S returns in firing order
 * (MulranPointCloudSelect.cpp:112-130) read in place.  Results never depend on the mode (reading in place is the default
 * for frames that qualify; BEV_STREAM=0 in the environment of bev_create turns it off). */
int bev_debug_get_frame_info(bev_ctx_t *ctx, int first_frame, int n_frames, uint32_t *out);
/* Test hook: out[i] = how many raster-band code lists of frame first_frame + i of the LAST sub-batch did not hold their
 * codes (those bands of the frame's images were computed from the ordered cloud instead; normally 0; BEV_CODE_CAP in the
 * environment of bev_create shrinks the lists). */
int bev_debug_get_code_overflow(bev_ctx_t *ctx, int first_frame, int n_frames, uint32_t *out);
/* Evaluates the phase-A angle predicate (BatchMultiBevGen.cpp:169-179) on the
 * device for n (dx,dy,dz) triples given as HOST arrays; out[i] = 1 if GROUND. */
int bev_debug_angle_predicate(bev_ctx_t *ctx, const float *dx, const float *dy,
                              const float *dz, uint8_t *out, size_t n);

int bev_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* BEV_MI355X_H */
