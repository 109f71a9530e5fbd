/*
 * bev_internal.h — shared between the HIP kernels (bev_kernels.hip) and the
 * C-ABI / context code (bev_capi.hip).  Not installed; the public boundary is
 * include/bev_mi355x.h.
 */
#ifndef BEV_INTERNAL_H
#define BEV_INTERNAL_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/bev_mi355x.h"
#include "bev_exact.h"

namespace bevk {

constexpr int kGatherThreads = 256;
constexpr int kSlotsPerThread = 4;
constexpr int kTile = kGatherThreads * kSlotsPerThread; /* slots per workgroup of the gather kernel */
constexpr int kMaxTiles = 1024;   /* => at most 2^20 slots per frame (bev_create checks) */
constexpr int kStripThreads = 256;              /* column-walk workgroup: 252 columns + 2 halo columns each side */
constexpr int kStripCols = kStripThreads - 4;
constexpr int kSeg = 256;                       /* capacity of one candidate segment = one (row, strip) */
constexpr int kMaxSegs = 1024;                  /* (G + 1) * strips must not exceed this (bev_create checks) */
constexpr int kSumWaves = 8;      /* waves of the per-frame cell-sum workgroup */
constexpr int kSumThreads = kSumWaves * 64;
constexpr int kRasterThreads = 1024;
constexpr int kRasterSplit = 4;   /* x-bands per frame in the raster kernel at the reference's 224 x 224 (see raster_bands_for) */

/* per-frame launch metadata, copied H2D once per sub-batch */
struct FrameDesc {
    uint64_t in_offset; /* element offset of the frame's points in d_pts */
    uint32_t n_pts;
    uint32_t _pad;
};

/* A "candidate" is a slot that phase A marked ground_mat == 1: the only slots whose
 * label depends on the per-cell averages (BatchMultiBevGen.cpp:244-246).  Stored as
 * three parallel arrays per (row, strip) segment so that the per-cell sum kernel reads
 * 2 (cell) resp. 6 (cell + z) bytes per candidate instead of a 16-byte record:
 *   cand_cell u16 : getBelongingGrid cell, row * 50 + col
 *   cand_cellp u16: the same cell | kCandPredBit if the walk wrote the point with its own label / code (its guess that
 *                   phase C un-grounds it).  A second array because the per-cell sum kernel must not pay for a mask:
 *                   with the bit inside cand_cell the compiler interleaved that kernel's batched loads with its
 *                   scattered stores (in-order vmcnt) and the kernel took 30 % longer.
 *   cand_z    f32 : the height that is summed
 *   cand_aux  2xu32 : x = column offset inside the strip (8 bit) | original label << 8,
 *                     y = BEV code the point gets back if phase C un-grounds it */
static_assert(sizeof(bev_point_t) == 32, "bev_point_t must be 32 bytes");

constexpr uint32_t kCandCellMask = 0x0fffu; /* 75 * 50 = 3750 cells */
constexpr uint32_t kCandPredBit = 0x8000u;

struct Geometry {
    int N, H, G;       /* N_SCAN, Horizon_SCAN, GROUND_UPPER_SCAN */
    int S;             /* N * H */
    int tiles;         /* ceil(S / kTile): slot tiles of the per-slot kernels */
    int strips;        /* ceil(H / kStripCols): column strips of the walk kernel */
    int segs;          /* (G + 1) * strips: candidate segments per frame, row-major */
    int raster_bands;  /* x-bands per frame in the raster kernel: a band's two LDS planes must fit one CU */
    bevx::RasterParams rp;
};

/* device-side views of one sub-batch */
struct BatchPtrs {
    const bev_point_t *pts;      /* packed input points (or ordered cloud in identity mode) */
    const FrameDesc *frames;
    uint32_t *winner;            /* [nf][S]  (win_tag << win_shift) | index+1 of the last input point per slot */
    uint32_t win_tag;            /* generation of this sub-batch in its workspace set (0: table was cleared) */
    int win_shift;               /* bits of index+1 */
    bev_point_t *ordered;        /* [nf][S] */
    uint32_t *codes;             /* [nf][S] */
    uint16_t *cand_cell;         /* [nf][segs][kSeg] */
    float *cand_z;               /* [nf][segs][kSeg] */
    uint2 *cand_aux;             /* [nf][segs][kSeg] */
    uint16_t *cand_cellp;        /* [nf][segs][kSeg] */
    uint32_t *ncand;             /* [nf][segs] */
    float *zsorted;              /* [nf][S] */
    float *avg;                  /* [nf][3750] */
    int8_t *gm;                  /* [nf][S] or nullptr: phase-A ground_mat */
    uint8_t *multi;              /* [nf][L*M*M] */
    uint8_t *single;             /* [nf][M*M] */
};

enum KernelId {
    K_ORDER_SCAN = 0,
    K_GATHER_GROUND,
    K_CELL_SUMS,
    K_GROUND_RESOLVE,
    K_BEV_RASTER,
    K_GATHER_ONLY,
    K_GROUND_MAT,
    K_CLOUD_CODES,
    K_ANGLE_DEBUG,
    K_FLOAT_BEV,
    K_PROJECT,
    K_TRANSFORM,
    K_COUNT
};
const char *kernel_name(int id);
/* smallest of 4, 8, 16 bands whose LDS planes (2 * (M / bands) * M * 4 B) fit; 0 if none does */
int raster_bands_for(int mat_size);

/* launchers (bev_kernels.hip) — all asynchronous on `st` */
void launch_order_scan(const Geometry &g, const BatchPtrs &b, int nf, uint32_t max_pts, hipStream_t st);
/* the column walk; identity: b.pts already is the ordered cloud (bev_mark_ground) */
void launch_gather_ground(const Geometry &g, const BatchPtrs &b, int nf, bool identity, hipStream_t st);
void launch_gather_only(const Geometry &g, const BatchPtrs &b, int nf, hipStream_t st);
void launch_cell_sums(const Geometry &g, const BatchPtrs &b, int nf, hipStream_t st);
void launch_ground_resolve(const Geometry &g, const BatchPtrs &b, int nf, hipStream_t st);
void launch_bev_raster(const Geometry &g, const uint32_t *codes, size_t code_stride, uint32_t n_codes,
                       uint8_t *multi, uint8_t *single, bool want_multi, bool want_single,
                       int nf, hipStream_t st);
void launch_ground_mat(const Geometry &g, const BatchPtrs &b, int8_t *out, int nf, hipStream_t st);
void launch_cloud_codes(const Geometry &g, const bev_point_t *cloud, uint32_t n, uint32_t *codes, hipStream_t st);
void launch_float_bev(const bev_point_t *cloud, uint32_t n, float interval, int M, bool skip_label0, float *grid,
                      hipStream_t st);
void launch_project(int kind, const float *xyzi, uint32_t n, bev_point_t *out, hipStream_t st);
void launch_transform(const bev_point_t *cloud, uint32_t n, const float m[12], bev_point_t *out, hipStream_t st);
/* KITTI projection workspace (device): header with the chain of accepted crossings, per-point column,
 * per-block crossing lists, winner table of the 64 x 2083 structured cloud */
struct KittiHeader {
    int32_t ring0;
    uint32_t n_links;
    uint32_t link[68];
};
struct KittiWork {
    KittiHeader *hdr;
    int32_t *col;     /* [n] */
    uint32_t *cnt;    /* [ceil(n / 256)] */
    uint32_t *pos;    /* [ceil(n / 256)][128] */
    uint32_t *winner; /* [64 * 2083] */
};
void launch_project_kitti(const float *xyzi, uint32_t n, const KittiWork &w, bev_point_t *out, hipStream_t st);
void launch_angle_debug(const float *dx, const float *dy, const float *dz, uint8_t *out, size_t n, hipStream_t st);
/* opt in to > 64 KiB of dynamic LDS for the two kernels that need it */
hipError_t configure_kernels(const Geometry &g);
size_t cell_sums_lds_bytes();
size_t raster_lds_bytes(const Geometry &g);

} /* namespace bevk */
#endif
