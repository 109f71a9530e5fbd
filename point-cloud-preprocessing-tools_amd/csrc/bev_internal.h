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
constexpr int kStripThreads = 256;              /* column-walk workgroup: 236 columns + 2 halo columns each side = 240 virtual columns */
constexpr int kStripCols = 236;                 /* (256 input positions cover them with 16 to spare: the in-place source's window, one per thread) */
constexpr int kStripVirt = kStripCols + 4;
constexpr int kSeg = 256;                       /* capacity of one candidate segment = one (row, strip) */
constexpr int kMaxSegs = 1024;                  /* (G + 1) * strips must not exceed this (bev_create checks) */
constexpr int kSumWaves = 4;      /* waves of the per-frame cell-sum workgroup: the size of a column-walk workgroup */
constexpr int kSumThreads = kSumWaves * 64;
/* ONE workgroup shape for every kernel of the hot path — 256 threads, at most a quarter of a CU's LDS (32 of its 128
 * allocation granules of 1,280 bytes) and of its registers (128 per lane) — so that the stages of different sub-batches can
 * be workgroups of ONE launch (k_stage) and any of them fits wherever another has retired.  (Rounds 3-5 ran the resolve and
 * the rasters as 512-thread workgroups, the rasters with 50 KB of LDS: beside the column walk they started only where two
 * walk workgroups of one CU had retired together.) */
constexpr int kStageThreads = 256;
constexpr int kSlotLdsBytes = 32 * 1280;
constexpr int kResolveThreads = kStageThreads;
constexpr int kResolveParts = 4;  /* code lists per frame written by k_ground_resolve (a contiguous quarter of the segments each) */
/* workgroups per frame in k_ground_resolve, kResolveParts / kResolveWgs consecutive parts each: a workgroup's tables
 * (3,750 averages, their neighbour minima, edge bins, band table) cost as much as a part's candidates */
#ifndef BEV_RESOLVE_WGS
#define BEV_RESOLVE_WGS 2
#endif
constexpr int kResolveWgs = BEV_RESOLVE_WGS;
static_assert(kResolveParts % kResolveWgs == 0, "whole parts per workgroup");
constexpr int kRasterThreads = kStageThreads;
constexpr int kRasterLdsCap = kSlotLdsBytes; /* a raster band's two planes (+ its list prefix) */
constexpr int kRasterFineDiv = 2;  /* the middle bands of the image are cut this many times finer (see fill_geometry) */
constexpr int kRasterSplit = 8;   /* fewest x-bands per frame in the raster kernel (see raster_bands_for) */
constexpr int kMaxBands = 64;     /* coarse + fine raster bands (see RasterParams) */
constexpr int kMaxStrips = 280;   /* ceil(65535 / kStripCols) rounded up */
/* Entries of one (emitter, band) code list.  Measured fill (scripts/list_fill.py: which capacities overflow on the
 * synthetic layouts): HDL_64E sweeps stay under 2,048 codes per list, OS1_64 frames under 3,072 — of the 15,104 slots a
 * strip has (the walk drops repeats of a code before they are listed); sized for the worst case the lists were 21.6 MB
 * per frame and workspace set, now 3.0 MB.  A writer whose list is full keeps counting and overwrites the list's
 * last entry; k_bev_raster sees the count and computes that band of the frame from the ordered cloud instead (it reads
 * all S slots: slow, correct, and only for clouds piled up in one band).  BEV_CODE_CAP (environment of bev_create)
 * overrides the size: the tests run with tiny lists. */
constexpr int kCodeListCap = 4096;
constexpr int kCtxTabWords = bevx::kGridRows + bevx::kGridCols + 512 / 4; /* BatchPtrs::ctx_tab */

/* per-frame launch metadata, copied H2D once per sub-batch */
struct FrameDesc {
    uint64_t in_offset; /* element offset of the frame's points in d_pts */
    uint32_t n_pts;
    uint32_t _pad;
};

/* How a frame's points reach their slots (getOrderedCloud, BatchMultiBevGen.cpp:94-117), decided per frame on the device:
 *   kFrameGeneral  any input: order scan over all points (winner table), the walk gathers through it;
 *   kFrameStream   the input's first T points are in strictly ascending slot order (a sweep written row by row, the
 *                  usual output of a selector): the walk reads them in place, row by row, and only the tail [T, n)
 *                  goes through the order scan; the walk VERIFIES the order of everything it consumes and counts it;
 *   kFrameRedo     a stream / structured frame whose verification failed: done again the general way (results never
 *                  depend on what k_probe guessed);
 *   kFrameStructured  the input IS a structured cloud of S records, what the KITTI selector writes
 *                  (KittiPointCloudSelect.cpp:206-207,240): record i is the point of slot i or an all-zero record.  The
 *                  scatter of getOrderedCloud is then the identity on every slot but slot 0, where every all-zero
 *                  record lands (row = col = 0): slot 0 ends up all-zero iff a record after the first is all-zero.  The
 *                  walk reads the records in place, once, coalesced, checks every one of them, and learns on the way
 *                  whether an all-zero record exists — which k_probe had to guess from its samples (the guess decides
 *                  slot 0 before the walk has seen the frame; a wrong guess is a failed frame);
 *   kFrameColMajor the input is S returns in firing order, the PLAIN sweep (BASELINE config 3): position k holds beam
 *                  k % N of firing k / N, its column is the firing's number plus 0 .. kPlainDisp (or out of range:
 *                  dropped).  The walk fetches a strip's firings band by band (two rows of a firing are one 64-byte
 *                  sector), settles the last writer of every slot in an LDS index row and checks every record it fetches;
 *   kFrameColMajorGen  ... what the MulRan selector writes for real sweeps (MulranPointCloudSelect.cpp:112-130): any start
 *                  azimuth, either direction, a base column per row (staggered beams) + 0 .. kColMaxDisp, no-return
 *                  records in column 0 (see the constants below; BatchPtrs::cm_par / cm_sync are this mode's alone). */
enum : uint32_t { kFrameGeneral = 0, kFrameStream = 1, kFrameRedo = 2, kFrameStructured = 3, kFrameColMajor = 4, kFrameColMajorGen = 5 };
__host__ __device__ inline bool frame_read_in_place(uint32_t mode) { return mode == kFrameStream || mode == kFrameStructured || mode == kFrameColMajor || mode == kFrameColMajorGen; }
struct FrameInfo {
    uint32_t T;        /* length of the prefix taken for sorted (structured: S) */
    uint32_t mode;
    uint32_t consumed; /* prefix points the stream walk has found in their own (row, strip) window (structured: records checked) */
    uint32_t failed;   /* bit 0: a consumed point was not above its predecessor, or could not be checked (structured: a
                        * record is neither its slot's point nor all-zero); structured frames also: bit 1: the walk saw an
                        * all-zero record after the first, bit 2: k_probe guessed that there is one */
};
constexpr uint32_t kInfoFailed = 1u, kInfoZeroSeen = 2u, kInfoZeroGuess = 4u;
constexpr uint32_t kInfoCmStray = 8u; /* firing-order frames whose strips do not talk: a strip other than 0 owns a no-return record: k_verdict checks that it would not have won column 0 */
constexpr uint32_t kInfoCmUsed = 16u; /* (a bit of its own: round 5 shared kInfoZeroSeen's, told apart by the frame's mode) firing-order frames: a wrap-around halo fell back on column 0 somewhere: k_verdict compares what it took with what strip 0 put there */
/* k_probe looks at every 63rd point of a frame (odd: no resonance with firing orders of 2^k beams) — at every 127th of a DENSE
 * sweep (at least nine tenths as many points as slots, and not exactly S records: those may be structured clouds or firing
 * orders, whose analysis wants its samples): the position of a slot between two samples is interpolated, its error grows with
 * the root of (stride x share of dropped returns), and the walk's windows have a dozen positions of slack — a sweep that
 * keeps 60 % of its returns fails its checks at 127 (tests/test_gpu_stream.py) and is fine at 63; the graded sweeps (98 %)
 * are fine at either, and the probe's samples are bytes: +1.8 % frames/s on the same box */
constexpr int kProbeStride = 63, kProbeStrideDense = 127;
constexpr int kMaxSamples = 4096;   /* => sorted frames of up to 258 k points are read in place (k_probe keeps their samples in LDS, a quarter of a CU's); longer ones go the general way */
constexpr int kStreamMinPrefix = 2048;
constexpr int kTailCap = 64;         /* tail points (those after the sorted prefix) a (row, strip) can list; more: general way */
constexpr int kTailMax = 16384;     /* ... a frame can have */
constexpr int kTailBuckets = 2048;  /* (row, strip) pairs of a frame that k_probe can count in LDS */
/* kFrameColMajor (round 5: any start azimuth, either direction of rotation, staggered beams, no-return records): with
 * u = +-firing mod H (the sweep's direction) a return of row r has column (u + B[r] + 0 .. kColMaxDisp) mod H, B[r] the
 * row's base (start azimuth + the beam's azimuth offset), or is out of range (>= H: dropped), or sits in column 0 whatever
 * its firing (a no-return record: x = y = 0 -> atan2(0, 0) = 0, MulranPointCloudSelect.cpp:123-125). */
constexpr int kPlainDisp = 8;        /* the PLAIN sweep (kFrameColMajor): column = firing + 0 .. kPlainDisp, or >= H (k_probe's test and k_walk<4>'s check) */
constexpr int kColMaxDisp = 12;      /* (a row's displacements may spread over 8 columns and still leave k_probe, which sees every 63rd record, two columns of slack on either side) */
constexpr int kCmSpread = 18;        /* the rows' bases lie within this many columns of each other (an OS1-64's four laser columns: +-9) */
constexpr int kCmProbeDisp = kColMaxDisp; /* spread of a row's SAMPLED displacements that k_probe accepts; what is left of kColMaxDisp is put half below, half above them, for the records it did not see */
constexpr int kCmExt = 16;           /* firings behind the 256 threads' that a strip fetches as well: 272 >= 240 columns + kColMaxDisp + kCmSpread */
constexpr int kCmMaxRows = 128;      /* sensors with more rows go the general way */
constexpr int kCmMaxSamples = 4096;  /* k_probe keeps this many samples (and their successors) for the analysis: frames of up to 258 k records */
constexpr int kCmParWords = 4 + kCmMaxRows; /* per frame: direction (+1 / -1 as int), the largest base (mod H), the rows' bases, how far apart the bases lie, whether a sample was a no-return record */
constexpr uint32_t kCmUsedBit = 0x80000000u;
constexpr int kCmMaxStrips = 16;     /* strips other than 0 tell strip 0 about their no-return records, two words per band of two rows each: frames of more strips go the general way */
constexpr int kCmPubWords = (kCmMaxRows / 2) * kCmMaxStrips * 2; /* BatchPtrs::cm_sync per frame: [band][strip][2] reports, then [row] what strip 0 put into column 0, then [row] what a wrap-around halo took for it */
constexpr int kCmSyncWords = kCmPubWords + 3 * kCmMaxRows; /* ... then [row] the last no-return firing + 1 that a strip other than 0 owns, for frames whose strips do not talk (k_verdict compares) */
constexpr int kStreamMaxRows = 64;   /* sensors with more rows go the general way (the stream walk keeps per-row estimates in LDS) */

/* Workspace streams between the kernels of one sub-batch (see bev_exact.h for the candidate key):
 *   cand uint2 (key | height)  [nf][segs][kSeg]   candidates, one segment per (row, strip), compacted in column order;
 *                                                  segments in row-major order => concatenation = slot order
 *   ncand u32                  [nf][segs]
 *   code_main u32              [nf][emitters][bands][code_stride]   BEV codes of the slots that are NOT candidates
 *                                                  (final when the walk writes them), one list per raster band, appended
 *                                                  row by row by the strip's workgroup (no atomics)
 *   ncode u32                  [nf][emitters][bands]
 *   (k_ground_resolve appends the codes of the candidates phase C un-grounds to lists of the same kind, one set per part) */
static_assert(sizeof(bev_point_t) == 32, "bev_point_t must be 32 bytes");

struct Geometry {
    int N, H, G;       /* N_SCAN, Horizon_SCAN, GROUND_UPPER_SCAN */
    int S;             /* N * H */
    int tiles;         /* ceil(S / kTile): slot tiles of the per-slot kernels */
    int strips;        /* ceil(H / kStripCols): column strips of the walk kernel */
    int segs;          /* (G + 1) * strips: candidate segments per frame, row-major */
    int raster_bands;  /* x-bands per frame in the raster kernel (= rp.bands: coarse ones outside, fine ones in the middle) */
    int emitters;      /* strips + kResolveParts: writers of code lists per frame (the walk's strips, the resolve's parts) */
    uint32_t code_cap; /* capacity of one (emitter, band) code list: kCodeListCap, or the worst case (every slot of a strip /
                        * every candidate of a resolve part in one band) where that is smaller; a raster band with a list
                        * that does not hold its codes is computed from the ordered cloud instead */
    uint32_t code_stride; /* words from one (emitter, band) list to the next: code_cap padded to an ODD number of 256-byte pieces —
                           * lists that start a power of two apart put every writer's appends on the same few memory channels
                           * (measured: -3 % frames/s with 8-KiB strides) */
    bevx::RasterParams rp;
};

/* device-side views of one sub-batch */
struct BatchPtrs {
    const bev_point_t *pts;      /* packed input points (or ordered cloud in identity mode) */
    const FrameDesc *frames;     /* [nf] where the walk and the scan read a frame's descriptor: device memory (k_probe's copy) or the host's mapped array */
    const FrameDesc *frames_src; /* [nf] k_probe: the host's mapped array the caller's offsets were written to (16 bytes per frame over the link, once) */
    FrameDesc *frames_copy;      /* [nf] k_probe writes its copy here (== frames), or nullptr */
    FrameInfo *info;             /* [nf] (nullptr: every frame general) */
    uint32_t *est;               /* [nf][strips][N]: stream frames: estimated input position of slot (r, first column of strip - 2) */
    uint32_t *tail_list;         /* [nf][N][strips][kTailCap]: stream frames: column offset | input index << 8 of the tail points (nullptr: no stream mode) */
    uint32_t *tail_cnt;          /* [nf][strips][N] */
    int32_t *cm_par;             /* [nf][kCmParWords]: kFrameColMajorGen: the frame's direction and row bases (k_probe) */
    uint32_t *cm_sync;           /* [nf][kCmSyncWords]: kFrameColMajorGen, zeroed by k_probe: per band of two rows and strip: kCmUsedBit | the last no-return firing + 1 of either row
                                  * that the strip owns; per row: the firing + 1 whose record strip 0 put into column 0; per row: kCmUsedBit | the firing + 1 that the
                                  * strip with the wrap-around halo took for column 0 when column H - 2 fell back on it */
    uint32_t *winner;            /* [nf][S]  (win_tag << win_shift) | index+1 of the last input point per slot */
    uint32_t win_tag;            /* generation of this sub-batch in its workspace set (0: table was cleared) */
    int win_shift;               /* bits of index+1 */
    bev_point_t *ordered;        /* [nf][S] */
    uint2 *cand;                 /* [nf][segs][kSeg]: candidate key (bev_exact.h) | height */
    uint32_t *ncand;             /* [nf][segs] */
    uint32_t *code_main;         /* [nf][emitters][bands][code_stride] */
    uint32_t *ncode;             /* [nf][emitters][bands]: codes the writer had for the list (more than code_cap: the list is incomplete) */
    const uint32_t *ctx_tab;     /* [kCtxTabWords] per context: edge_x[75], edge_y[50] (BEV bins of the ground grid's cell edges), band_tab[512] bytes (x bin -> raster band) */
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
    K_PROBE,
    K_WALK_GENERAL, /* the walk through the winner table (K_GATHER_GROUND: the walk that reads in place, or the identity walk) */
    K_WALK_STRUCTURED, /* the walk over structured clouds (kFrameStructured) */
    K_WALK_COLMAJOR,   /* the walk over clouds in firing order: the plain sweep (kFrameColMajor) */
    K_WALK_COLMAJOR_GEN, /* ... any start azimuth, direction, staggered beams, no-return records (kFrameColMajorGen) */
    K_VERDICT,
    K_STAGE, /* the fused launch: a sub-batch's walk beside the later stages of the sub-batches before it (k_stage) */
    K_COUNT
};
const char *kernel_name(int id);
/* smallest of 4, 8, 16 bands whose LDS planes (2 * (M / bands) * M * 4 B) fit; 0 if none does */
int raster_bands_for(int mat_size);

/* One fused launch (k_stage): the column walk of one sub-batch and, as further workgroups of the same grid, phase B of the
 * sub-batch before it, phase C of the one before that and the rasters of the one before that — every dependency between
 * stages of one sub-batch is the order of launches on one stream; nothing inside a launch waits for anything.  nf == 0: the
 * stage is absent from this launch. */
struct StagePart {
    BatchPtrs b;
    int nf;
};
struct StageArgs {
    Geometry g;
    StagePart walk, sums, resolve, raster;
    uint32_t want_mode;           /* of the walk part (see launch_gather_ground) */
    int want_multi, want_single;  /* of the raster part */
    int lead;                     /* group slots (8 frames each) by which the walk's workgroups precede the other stages' in the grid */
};
/* source: the walk part's (see launch_gather_ground; ignored when a.walk.nf == 0) */
void launch_stage(const StageArgs &a, int source, hipStream_t st);
size_t stage_lds_bytes(const Geometry &g, int source);

/* launchers (bev_kernels.hip) — all asynchronous on `st` */
/* the frames that are not read in place: general ones and (after k_verdict) those whose verification failed */
void launch_order_scan(const Geometry &g, const BatchPtrs &b, int nf, uint32_t max_pts, bool thin, hipStream_t st);
/* the column walk.  source 0: through the winner table (the frames that are not read in place: pass kFrameGeneral);
 * 1: identity, b.pts already is the ordered cloud (bev_mark_ground); 2: in place (frames of mode kFrameStream: pass
 * mode = kFrameStream); 3: structured clouds (kFrameStructured); 4: the plain sweep in firing order (kFrameColMajor);
 * 5: firing order in its general form (kFrameColMajorGen) — sources 2 .. 5 take the frames of the mode passed */
void launch_gather_ground(const Geometry &g, const BatchPtrs &b, int nf, int source, uint32_t mode, hipStream_t st);
void launch_probe(const Geometry &g, const BatchPtrs &b, int nf, bool allow_stream, int layout_hint /* 0, kFrameStructured or kFrameColMajor */, hipStream_t st);
void launch_verdict(const Geometry &g, const BatchPtrs &b, int nf, uint32_t *host_hint, hipStream_t st);
void launch_gather_only(const Geometry &g, const BatchPtrs &b, int nf, hipStream_t st);
void launch_cell_sums(const Geometry &g, const BatchPtrs &b, int nf, hipStream_t st);
void launch_ground_resolve(const Geometry &g, const BatchPtrs &b, int nf, hipStream_t st);
/* the rasters of a sub-batch from its code lists */
void launch_bev_raster(const Geometry &g, const BatchPtrs &b, bool want_multi, bool want_single, int nf, hipStream_t st);
/* rasters of ONE arbitrary cloud from a dense code array (bev_multi_bev / bev_single_bev) */
void launch_bev_raster_dense(const Geometry &g, const uint32_t *codes, uint32_t n_codes, uint8_t *multi, uint8_t *single,
                             hipStream_t st);
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
/* opt in to > 64 KiB of dynamic LDS for the kernels that need it */
hipError_t configure_kernels(const Geometry &g);
size_t cell_sums_lds_bytes();
size_t raster_lds_bytes(const Geometry &g);

} /* namespace bevk */
#endif
