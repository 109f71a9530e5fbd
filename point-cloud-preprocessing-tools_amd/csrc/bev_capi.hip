/*
 * bev_capi.hip — context, workspace and the extern "C" boundary declared in
 * include/bev_mi355x.h.  Host-side only; the kernels are in bev_kernels.hip.
 *
 * There is deliberately no CPU implementation behind these entry points: if
 * HIP cannot give us a device, bev_create() fails.
 */
#include <algorithm>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <thread>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include <roctracer/roctx.h>

#include "bev_internal.h"
#include "bev_libm.h"

using namespace bevk;

namespace {

constexpr int kDescRing = 4; /* calls the host may run ahead of the device (2 and 12 measured the same) */
constexpr int kEventPairs = 2048;

struct ProfSlot {
    hipEvent_t a, b;
    int kid;
    int frames;
};

/* A sub-batch's workspace lives from its column walk to its rasters: four launches of its stream, and two streams take
 * sub-batches in turn (k_stage, see run_pipeline), so eight workspace sets ("lanes", the name rounds 2-5 gave them when
 * each also had a stream) go round.  Lane 0 doubles as
 * the workspace of the single-cloud entry points. */
constexpr int kMaxStageStreams = 4;
constexpr int kMaxLanes = 4 * kMaxStageStreams;
struct Lane {
    FrameInfo *info = nullptr;  /* per frame: how its points reach their slots (k_probe / k_verdict) */
    FrameDesc *desc = nullptr;  /* per frame: k_probe's device copy of the caller's descriptor */
    uint32_t *est = nullptr;    /* stream frames: estimated input position of every (row, strip)'s first slot */
    uint32_t *tail_list = nullptr, *tail_cnt = nullptr; /* ... and their tail points per (row, strip) (stream mode only) */
    int32_t *cm_par = nullptr;   /* firing-order frames: direction and row bases (k_probe) */
    uint32_t *cm_sync = nullptr; /* ... and what their strips tell each other and k_verdict about column 0 */
    uint32_t *winner = nullptr;
    uint32_t win_gen = 0; /* generation tag of the last sub-batch that used this set's winner table */
    uint2 *cand = nullptr; /* candidate key | height */
    uint32_t *ncand = nullptr;
    uint32_t *code_main = nullptr, *ncode = nullptr; /* per-(strip, band) lists of final BEV codes */
    float *avg = nullptr;
    int8_t *gm = nullptr; /* lazily allocated */
};

} // namespace

/* Device -> host side of bev_process_batch.  Copies into pageable host memory block the calling thread, so the
 * downloads of chunk k run on their own thread and stream while the main thread uploads and launches chunk k + 1:
 * PCIe is used in both directions at once.  The thread lives as long as the context (it used to be created and joined
 * by every call). */
struct bev_ctx;
namespace {
struct Downloader {
    struct Task {
        int f0, nb, half;
        bev_point_t *const *ordered_out;
        uint8_t *const *multi_out;
        uint8_t *const *single_out;
        int8_t *const *gm_out;
        int half_frames;
    };
    bev_ctx *c = nullptr;
    std::mutex mu;
    std::condition_variable cv;
    std::deque<Task> queue;
    bool closing = false;
    int finished = 0; /* chunks of the current call whose outputs are in the caller's buffers */
    hipError_t err = hipSuccess;
    std::thread th;

    void run();
    void start(bev_ctx *ctx)
    {
        c = ctx;
        th = std::thread([this] { run(); });
    }
    void begin_call()
    {
        std::lock_guard<std::mutex> lk(mu);
        finished = 0;
        err = hipSuccess;
    }
    void push(Task t)
    {
        {
            std::lock_guard<std::mutex> lk(mu);
            queue.push_back(t);
        }
        cv.notify_all();
    }
    void wait_finished(int n)
    {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return finished >= n; });
    }
    void close()
    {
        {
            std::lock_guard<std::mutex> lk(mu);
            closing = true;
        }
        cv.notify_all();
        if (th.joinable()) th.join();
    }
};
} // namespace

struct bev_ctx {
    int device = -1;
    bev_params_t params{};
    Geometry geo{};
    int max_batch = 0;
    size_t max_points = 0;
    int win_shift = 32;  /* bits an input index + 1 needs; the rest of a winner entry is the generation tag */
    size_t multi_bytes = 0, single_bytes = 0;
    hipStream_t stream = nullptr;

    /* sub-batch workspace sets; the aliases below are lane 0's */
    Lane lanes[kMaxLanes];
    int n_lanes = 8; /* 4 * n_stage_streams */
    /* fused launches alternate between two streams: sub-batch s on stage_st[s % 2], its workspace set s % 8 (always the
     * same stream's), its later stages in that stream's next three launches — a launch's tail is filled by the other
     * stream's launch, and nothing but the order of launches on ONE stream ever orders two stages of one sub-batch */
    hipStream_t stage_st[kMaxStageStreams] = {};
    hipEvent_t stage_ev[kMaxStageStreams] = {};
    hipEvent_t fork_ev = nullptr, null_ev = nullptr;
    int n_stage_streams = 2;   /* BEV_STAGE_STREAMS=1 .. 4 (1: a launch's tail stands empty; 3, 4: measured like 2, with 12 / 16 workspace sets) */
    unsigned sub_seq = 0;      /* sub-batches so far */
    /* fused: a sub-batch's stages ride in consecutive k_stage launches beside the stages of its neighbours (run_pipeline);
     * serial (BEV_LANES=1, bev_set_lanes(ctx, 1)): every kernel a launch of its own, back to back — per-kernel durations */
    bool fused = true;
    int stage_lead = 0;        /* group slots by which a launch's walk workgroups precede its other stages' (0, 4, 12, 24, 32 measured the same) */
    uint32_t *hint = nullptr;  /* mapped host words (k_verdict): [0] frames of the last verdict's sub-batch that were NOT read in place, [1] the modes k_probe gave its frames (bit = mode) */
    int mode_absent[8] = {0, 0, 0, 0, 0, 0, 0, 0}; /* looks at hint[1] since it last showed the mode (see run_pipeline) */
    int mode_ttl = 8;          /* a mode's in-place walk stays launched for this many sub-batches after a verdict last showed the mode (BEV_MODE_TTL) */
    int layout_hint = 0;       /* bev_set_layout_hint: 0, kFrameStructured or kFrameColMajor */
    bool allow_stream = true;  /* sorted-prefix frames are read in place (k_probe); BEV_STREAM=0 turns it off, see bev_create */
    /* sub-batches whose later stages have not been launched yet, oldest first (see run_pipeline / flush_pending) */
    struct Pending {
        BatchPtrs b;
        int nf;
        bool want_multi, want_single;
        int8_t *gm_out; /* final ground_mat wanted (device), or nullptr */
        int next;       /* 1 phase B, 2 phase C, 3 rasters */
        int q;          /* which of the two streams its stages ride on */
    };
    std::deque<Pending> pending;
    uint32_t *winner = nullptr;
    uint32_t *codes = nullptr;
    size_t codes_elems = 0;
    uint32_t *ctx_tab = nullptr; /* per-context tables (BatchPtrs::ctx_tab) */
    float *last_avg = nullptr;
    uint32_t *last_ncode = nullptr;
    FrameInfo *last_info = nullptr;

    /* frame descriptors: ring of pinned host + device arrays */
    FrameDesc *h_desc[kDescRing] = {nullptr, nullptr, nullptr, nullptr};
    FrameDesc *d_desc[kDescRing] = {nullptr, nullptr, nullptr, nullptr}; /* the device's address of h_desc (mapped host memory) */
    size_t desc_cap[kDescRing] = {0, 0, 0, 0};
    hipEvent_t desc_done[kDescRing]{};
    bool desc_used[kDescRing] = {false, false, false, false};
    int desc_next = 0;

    /* staging for the host-buffer entry points (lazily allocated) */
    bev_point_t *st_in = nullptr;
    size_t st_in_elems = 0;
    bev_point_t *st_ordered = nullptr;
    uint8_t *st_multi = nullptr, *st_single = nullptr;
    int8_t *st_gm = nullptr;
    bool staging_ready = false;
    hipStream_t dl_stream = nullptr;             /* device -> host copies of bev_process_batch (own host thread) */
    Downloader *downloader = nullptr;            /* that thread, started with the staging buffers */
    hipEvent_t out_ready[2] = {nullptr, nullptr}; /* per half of the output staging: its chunk has been computed */
    /* KITTI projection workspace, one allocation made on first use and grown on demand */
    void *kitti_buf = nullptr;
    size_t kitti_points = 0;

    /* profiling */
    bool prof_on = false;
    std::vector<ProfSlot> prof_pool;
    size_t prof_used = 0;
    double prof_ms[K_COUNT]{};
    uint64_t prof_launches[K_COUNT]{};
    uint64_t prof_frames[K_COUNT]{};

    int last_sub_frames = 0;
    std::string last_error;
};

namespace {
void Downloader::run()
{
    (void)hipSetDevice(c->device);
    const size_t S = (size_t)c->geo.S;
    for (;;) {
        Task t;
        {
            std::unique_lock<std::mutex> lk(mu);
            cv.wait(lk, [&] { return closing || !queue.empty(); });
            if (queue.empty()) return;
            t = queue.front();
            queue.pop_front();
        }
        hipError_t e = hipStreamWaitEvent(c->dl_stream, c->out_ready[t.half], 0);
        const size_t base = (size_t)t.half * t.half_frames;
        auto copy = [&](void *dst, const void *src, size_t n) {
            if (e == hipSuccess && dst) e = hipMemcpyAsync(dst, src, n, hipMemcpyDeviceToHost, c->dl_stream);
        };
        /* frames whose destination buffers follow each other in host memory (rows of one array) leave in ONE copy:
         * three small copies per frame otherwise cost the link ~15 % in call overhead */
        auto copy_runs = [&](auto *const *dst, const unsigned char *src, size_t bytes) {
            for (int f = 0; f < t.nb;) {
                int e2 = f + 1;
                auto *d0 = reinterpret_cast<unsigned char *>(dst[t.f0 + f]);
                while (e2 < t.nb && d0 && reinterpret_cast<unsigned char *>(dst[t.f0 + e2]) == d0 + (size_t)(e2 - f) * bytes) ++e2;
                if (!d0) e2 = f + 1;
                copy(d0, src + (base + f) * bytes, (size_t)(e2 - f) * bytes);
                f = e2;
            }
        };
        copy_runs(t.ordered_out, reinterpret_cast<const unsigned char *>(c->st_ordered), S * sizeof(bev_point_t));
        if (t.multi_out) copy_runs(t.multi_out, c->st_multi, c->multi_bytes);
        if (t.single_out) copy_runs(t.single_out, c->st_single, c->single_bytes);
        if (t.gm_out) copy_runs(t.gm_out, reinterpret_cast<const unsigned char *>(c->st_gm), S);
        if (e == hipSuccess) e = hipStreamSynchronize(c->dl_stream);
        {
            std::lock_guard<std::mutex> lk(mu);
            if (e != hipSuccess && err == hipSuccess) err = e;
            ++finished;
        }
        cv.notify_all();
    }
}
} // namespace

namespace {

int hip_fail(bev_ctx *c, hipError_t e, const char *what, int line)
{
    char buf[256];
    snprintf(buf, sizeof buf, "%s failed at bev_capi.hip:%d: %s", what, line, hipGetErrorString(e));
    if (c) c->last_error = buf;
    return e == hipErrorOutOfMemory ? BEV_ERR_OOM : BEV_ERR_HIP;
}

#define HIPCK(ctx, expr)                                                   \
    do {                                                                   \
        hipError_t e_ = (expr);                                            \
        if (e_ != hipSuccess) return hip_fail((ctx), e_, #expr, __LINE__); \
    } while (0)

int mat_size_of(const bev_params_t *p)
{
    /* static int MAT_SIZE = MAX_RANGE*2 / interval;  BatchMultiBevGen.cpp:267 */
    return bevx::cvtt_f32((float)(p->max_range * 2) / p->interval);
}

int validate_params(const bev_params_t *p)
{
    if (!p) return BEV_ERR_INVALID_ARG;
    if (p->n_scan < 3 || p->n_scan > 65535 || p->horizon_scan < 5 || p->horizon_scan > 65535) return BEV_ERR_INVALID_ARG;
    /* lo = N - G >= 2 keeps every index phase A touches inside the cloud
     * (the reference would read out of bounds otherwise) */
    if (p->ground_upper_scan < 1 || p->ground_upper_scan > p->n_scan - 2) return BEV_ERR_INVALID_ARG;
    if (!(p->interval > 0.0f) || p->max_range <= 0) return BEV_ERR_INVALID_ARG;
    if ((size_t)p->n_scan * (size_t)p->horizon_scan > (size_t)kMaxTiles * kTile) return BEV_ERR_UNSUPPORTED;
    if ((size_t)(p->ground_upper_scan + 1) * (size_t)((p->horizon_scan + kStripCols - 1) / kStripCols) > (size_t)kMaxSegs)
        return BEV_ERR_UNSUPPORTED;
    const int M = mat_size_of(p);
    if (M < 16 || M > 512 || (M % 16) != 0 || raster_bands_for(M) == 0) return BEV_ERR_UNSUPPORTED;
    if (p->n_layers < 1 || p->n_layers > 30) return BEV_ERR_UNSUPPORTED;
    return BEV_OK;
}

void fill_geometry(const bev_params_t *p, Geometry *g)
{
    g->N = p->n_scan;
    g->H = p->horizon_scan;
    g->G = p->ground_upper_scan;
    g->S = p->n_scan * p->horizon_scan;
    g->tiles = (g->S + kTile - 1) / kTile;
    g->strips = (g->H + kStripCols - 1) / kStripCols;
    g->segs = (g->G + 1) * g->strips;
    {   /* raster bands: M / u rows each, the middle quarter of the image cut four times finer (bev_exact.h) */
        const int M = mat_size_of(p), u = raster_bands_for(M);
        const int coarse = u ? M / u : M;
        int fine = coarse % kRasterFineDiv == 0 ? coarse / kRasterFineDiv : coarse;
        const int z0 = (3 * u / 8) * coarse, z1 = M - z0;
        if (2 * (z0 / coarse) + (z1 - z0) / fine > kMaxBands) fine = coarse; /* (large images: uniform bands) */
        g->rp.coarse = coarse;
        g->rp.fine = fine;
        g->rp.z0 = z0;
        g->rp.z1 = z1;
        g->rp.bands = 2 * (z0 / coarse) + (z1 - z0) / fine;
        g->rp.coarse_magic = bevx::small_div_magic(coarse);
        g->rp.fine_magic = bevx::small_div_magic(fine);
        g->raster_bands = g->rp.bands;
    }
    g->emitters = g->strips + kResolveParts;
    {   /* worst case: every slot of a strip / every candidate of a resolve part in one band; normally far fewer (kCodeListCap) */
        const uint32_t worst = std::max((uint32_t)g->N * (uint32_t)kStripCols, (uint32_t)((g->segs + kResolveParts - 1) / kResolveParts + 1) * (uint32_t)kSeg);
        uint32_t cap = (uint32_t)kCodeListCap;
        if (const char *e = getenv("BEV_CODE_CAP")) cap = (uint32_t)std::max(1, atoi(e));
        g->code_cap = std::min(worst, cap);
        const uint32_t pieces = (g->code_cap + 63u) / 64u; /* 256-byte pieces */
        g->code_stride = (pieces | 1u) * 64u;
    }
    g->rp.max_range_f = (float)p->max_range;
    g->rp.interval = p->interval;
    g->rp.height_res = p->height_res;
    g->rp.lidar_to_ground = p->lidar_to_ground;
    g->rp.mat_size = mat_size_of(p);
    g->rp.n_layers = p->n_layers;
    g->rp.inv_interval = bevx::exact_reciprocal(p->interval);
    g->rp.inv_height_res = bevx::exact_reciprocal(p->height_res);
}

/* ---- profiling -------------------------------------------------------- */
int prof_flush(bev_ctx *c)
{
    for (size_t i = 0; i < c->prof_used; ++i) {
        ProfSlot &s = c->prof_pool[i];
        HIPCK(c, hipEventSynchronize(s.b));
        float ms = 0.f;
        HIPCK(c, hipEventElapsedTime(&ms, s.a, s.b));
        c->prof_ms[s.kid] += ms;
        c->prof_launches[s.kid] += 1;
        c->prof_frames[s.kid] += (uint64_t)s.frames;
    }
    c->prof_used = 0;
    return BEV_OK;
}

/* roctx range on the calling host thread (rocprofv3 --marker-trace): the counterpart of the reference's timer around
 * its per-frame region (BatchMultiBevGen.cpp:724,732,749-752), here around what the host enqueues per sub-batch stage */
struct RoctxRange {
    explicit RoctxRange(const char *name) { roctxRangePush(name); }
    ~RoctxRange() { roctxRangePop(); }
};

struct ProfScope {
    bev_ctx *c;
    ProfSlot *s = nullptr;
    hipStream_t st;
    ProfScope(bev_ctx *ctx, int kid, int frames, hipStream_t stream = nullptr) : c(ctx), st(stream ? stream : ctx->stream)
    {
        if (!c->prof_on) return;
        if (c->prof_used == c->prof_pool.size()) {
            if (prof_flush(c) != BEV_OK) return;
        }
        s = &c->prof_pool[c->prof_used++];
        s->kid = kid;
        s->frames = frames;
        (void)hipEventRecord(s->a, st);
    }
    ~ProfScope()
    {
        if (s) (void)hipEventRecord(s->b, st);
    }
};

/* ---- frame descriptors -------------------------------------------------- */
int acquire_desc(bev_ctx *c, size_t n, int *slot_out)
{
    const int k = c->desc_next;
    c->desc_next = (k + 1) % kDescRing;
    if (c->desc_used[k]) HIPCK(c, hipEventSynchronize(c->desc_done[k]));
    if (c->desc_cap[k] < n) {
        if (c->h_desc[k]) HIPCK(c, hipHostFree(c->h_desc[k]));
        c->h_desc[k] = nullptr;
        c->d_desc[k] = nullptr;
        const size_t cap = std::max<size_t>(n, 64);
        /* mapped: the kernels read a frame's 16 bytes over the link, once per workgroup — no copy command between the
         * launches of a stream (the copies of two streams share a copy queue, where one waits behind the other) */
        HIPCK(c, hipHostMalloc((void **)&c->h_desc[k], cap * sizeof(FrameDesc), hipHostMallocMapped));
        HIPCK(c, hipHostGetDevicePointer((void **)&c->d_desc[k], c->h_desc[k], 0));
        c->desc_cap[k] = cap;
    }
    *slot_out = k;
    return BEV_OK;
}

int ensure_gm(bev_ctx *c)
{
    for (int l = 0; l < c->n_lanes; ++l)
        if (!c->lanes[l].gm) HIPCK(c, hipMalloc((void **)&c->lanes[l].gm, (size_t)c->max_batch * c->geo.S));
    return BEV_OK;
}

/* ---- the stages of earlier sub-batches that have not been launched yet ------------------------------------------- */
/* Fills the later-stage parts of a fused launch from the pending sub-batches: each is advanced by ONE stage.  With every
 * launch advancing every pending sub-batch there is at most one per stage. */
/* (the invariant: every launch of a stream advances every pending sub-batch of that stream by one stage and adds at most one
 * new one, so no two of them wait for the same stage; should it ever not hold, the caller launches what is pending first) */
bool pending_is_consistent(const bev_ctx *c, int q)
{
    unsigned seen = 0u;
    for (const bev_ctx::Pending &p : c->pending) {
        if (p.q != q) continue;
        if (p.next < 1 || p.next > 3 || (seen & (1u << p.next))) return false;
        seen |= 1u << p.next;
    }
    return true;
}
void take_pending(bev_ctx *c, StageArgs *a, int q)
{
    a->sums = StagePart{};
    a->resolve = StagePart{};
    a->raster = StagePart{};
    a->want_multi = a->want_single = 0;
    for (const bev_ctx::Pending &p : c->pending) {
        if (p.q != q) continue;
        StagePart &part = p.next == 1 ? a->sums : (p.next == 2 ? a->resolve : a->raster);
        part.b = p.b;
        part.nf = p.nf;
        if (p.next == 3) {
            a->want_multi = p.want_multi ? 1 : 0;
            a->want_single = p.want_single ? 1 : 0;
            if (!p.want_multi && !p.want_single) part.nf = 0;
        }
    }
}
/* ... after that launch: the optional final ground_mat of the sub-batch whose averages are now final (a plain launch:
 * rarely asked for), sub-batches through their rasters leave the queue */
int advance_pending(bev_ctx *c, int q)
{
    for (bev_ctx::Pending &p : c->pending) {
        if (p.q != q) continue;
        if (p.next == 1 && p.gm_out) {
            ProfScope ps(c, K_GROUND_MAT, p.nf, c->stage_st[q]);
            launch_ground_mat(c->geo, p.b, p.gm_out, p.nf, c->stage_st[q]);
        }
        ++p.next;
    }
    for (auto it = c->pending.begin(); it != c->pending.end();) it = it->next > 3 ? c->pending.erase(it) : it + 1;
    HIPCK(c, hipGetLastError());
    return BEV_OK;
}
/* the context's stream continues behind everything the stage streams hold */
int join_stage_streams(bev_ctx *c)
{
    for (int q = 0; q < kMaxStageStreams; ++q) {
        HIPCK(c, hipEventRecord(c->stage_ev[q], c->stage_st[q]));
        HIPCK(c, hipStreamWaitEvent(c->stream, c->stage_ev[q], 0));
    }
    return BEV_OK;
}
/* launches what is left of every pending sub-batch: up to three launches without a walk per stream; then joins */
int flush_pending(bev_ctx *c)
{
    while (!c->pending.empty()) {
        for (int q = 0; q < kMaxStageStreams; ++q) {
            bool any = false;
            for (const bev_ctx::Pending &p : c->pending) any = any || p.q == q;
            if (!any) continue;
            StageArgs a{};
            a.g = c->geo;
            a.lead = 0;
            take_pending(c, &a, q);
            {
                ProfScope ps(c, K_STAGE, 0, c->stage_st[q]);
                launch_stage(a, -1, c->stage_st[q]);
            }
            int rc = advance_pending(c, q);
            if (rc != BEV_OK) return rc;
        }
    }
    return join_stage_streams(c);
}

/* The whole pipeline on device pointers.  `identity`: d_pts already holds ordered
 * clouds (n_frames * S points) and the order stage is skipped.
 *
 * Fused (the default): per sub-batch t ONE launch of k_stage holds its column walk and, as further workgroups of the
 * same grid, phase B of sub-batch t - 1, phase C of t - 2 and the rasters of t - 3 (bev_kernels.hip).  The stages of a
 * sub-batch that have not been launched when the call returns stay PENDING: the next call's launches carry them, or
 * bev_synchronize() (and every other entry point) launches them — calls that follow each other without a
 * synchronisation keep the pipeline full across the call boundary.  `flush`: launch them before returning.
 * Serial (bev_set_lanes(ctx, 1)): every kernel a launch of its own, back to back. */
int run_pipeline(bev_ctx *c, int n_frames, const bev_point_t *d_pts, const uint64_t *h_offsets, bool identity,
                 bev_point_t *d_ordered, uint8_t *d_multi, uint8_t *d_single, int8_t *d_gm, bool flush = true,
                 int sub_frames = 0 /* frames per sub-batch; 0: max_batch */,
                 bool fork = true /* the stage streams start behind what the caller queued on the context's stream (its uploads) */)
{
    if (n_frames == 0) return BEV_OK;
    const Geometry &g = c->geo;
    const size_t S = (size_t)g.S;
    HIPCK(c, hipSetDevice(c->device));
    if (d_gm) {
        int rc = ensure_gm(c);
        if (rc != BEV_OK) return rc;
    }
    const bool fused = c->fused && !identity;
    if (!fused) { /* (the serial launches, on the context's stream, use the same workspace sets) */
        int rc = flush_pending(c);
        if (rc != BEV_OK) return rc;
    } else if (fork) {
        HIPCK(c, hipEventRecord(c->fork_ev, c->stream));
        for (int q = 0; q < kMaxStageStreams; ++q) HIPCK(c, hipStreamWaitEvent(c->stage_st[q], c->fork_ev, 0));
    }
    if (!fork && !identity) {
        /* device pointers from the caller: whatever it has queued on the default stream up to now — the upload or the fill of
         * these very buffers, typically — comes first (the library's streams are non-blocking: nothing else orders them behind
         * it; work on OTHER streams of the caller's is the caller's to wait for) */
        HIPCK(c, hipEventRecord(c->null_ev, nullptr));
        if (!fused) HIPCK(c, hipStreamWaitEvent(c->stream, c->null_ev, 0));
        for (int q = 0; fused && q < c->n_stage_streams; ++q) HIPCK(c, hipStreamWaitEvent(c->stage_st[q], c->null_ev, 0));
    }

    int ds = 0;
    if (!identity) {
        int rc = acquire_desc(c, (size_t)n_frames, &ds);
        if (rc != BEV_OK) return rc;
        for (int f = 0; f < n_frames; ++f) {
            const uint64_t a = h_offsets[f], b = h_offsets[f + 1];
            if (b < a || b - a > c->max_points) {
                c->desc_used[ds] = false;
                return BEV_ERR_TOO_LARGE;
            }
            c->h_desc[ds][f].in_offset = a;
            c->h_desc[ds][f].n_pts = (uint32_t)(b - a);
            c->h_desc[ds][f]._pad = 0;
        }
    }

    const int sub_size = sub_frames > 0 ? std::min(sub_frames, c->max_batch) : c->max_batch;
    for (int f0 = 0; f0 < n_frames; f0 += sub_size) {
        const int nb = std::min(sub_size, n_frames - f0);
        /* the set's previous tenant is through its rasters: they rode in this stream's launch before this one at the latest */
        const int q = fused ? (int)(c->sub_seq % (unsigned)c->n_stage_streams) : 0;
        hipStream_t st = fused ? c->stage_st[q] : c->stream;
        Lane &ln = c->lanes[c->sub_seq % (unsigned)c->n_lanes];
        ++c->sub_seq;
        BatchPtrs b{};
        b.pts = identity ? d_pts + (size_t)f0 * S : d_pts;
        b.frames = identity ? nullptr : ln.desc;
        b.frames_src = identity ? nullptr : c->d_desc[ds] + f0;
        b.frames_copy = identity ? nullptr : ln.desc;
        b.info = identity ? nullptr : ln.info;
        b.est = ln.est;
        b.tail_list = ln.tail_list;
        b.tail_cnt = ln.tail_cnt;
        b.cm_par = ln.cm_par;
        b.cm_sync = ln.cm_sync;
        b.winner = ln.winner;
        b.win_shift = c->win_shift;
        b.ordered = d_ordered + (size_t)f0 * S;
        b.cand = ln.cand;
        b.ncand = ln.ncand;
        b.code_main = ln.code_main;
        b.ncode = ln.ncode;
        b.ctx_tab = c->ctx_tab;
        b.avg = ln.avg;
        b.gm = d_gm ? ln.gm : nullptr;
        b.multi = d_multi ? d_multi + (size_t)f0 * c->multi_bytes : nullptr;
        b.single = d_single ? d_single + (size_t)f0 * c->single_bytes : nullptr;

        /* the walk of `source` over the frames of `mode`: the sub-batch's FIRST walk launch carries the pending stages of
         * the sub-batches before it (fused), every other one is a plain launch */
        bool carried = !fused;
        auto walk = [&](int kid, int source, uint32_t mode) -> int {
            if (!carried) {
                carried = true;
                StageArgs a{};
                a.g = g;
                a.walk.b = b;
                a.walk.nf = nb;
                a.want_mode = mode;
                a.lead = c->stage_lead;
                if (!pending_is_consistent(c, q)) {
                    const int rc_ = flush_pending(c);
                    if (rc_ != BEV_OK) return rc_;
                }
                take_pending(c, &a, q);
                if (source == 4 || source == 5) {
                    /* the firing-order walks need 50-53 KB of LDS, three workgroups per CU — fused, every stage of the launch
                     * would run three per CU; so the walk is a launch of its own and the later stages a fused launch without a
                     * walk behind it, four per CU (same box, three passes: config 3 + 0.3 ... 1.8 %, real MulRan sweeps + 2.1 ...
                     * 3.3 %; profiles/r06_experiments.txt) */
                    {
                        ProfScope ps(c, kid, nb, st);
                        launch_gather_ground(g, b, nb, source, mode, st);
                    }
                    a.walk.nf = 0;
                    a.lead = 0;
                    ProfScope ps(c, K_STAGE, 0, st);
                    launch_stage(a, -1, st);
                } else {
                    ProfScope ps(c, K_STAGE, nb, st);
                    launch_stage(a, source, st);
                }
                return advance_pending(c, q);
            }
            ProfScope ps(c, kid, nb, st);
            launch_gather_ground(g, b, nb, source, mode, st);
            return BEV_OK;
        };

        if (identity) {
            RoctxRange rr("bev:front (identity walk)");
            int rc = walk(K_GATHER_GROUND, 1, kFrameGeneral);
            if (rc != BEV_OK) return rc;
        } else {
            RoctxRange rr("bev:front (probe, column walk beside the later stages of earlier sub-batches, verdict, order scan)");
            uint32_t max_pts = 0;
            int n_exact_s = 0; /* frames that can be structured clouds */
            for (int f = 0; f < nb; ++f) {
                max_pts = std::max(max_pts, c->h_desc[ds][f0 + f].n_pts);
                n_exact_s += c->h_desc[ds][f0 + f].n_pts == (uint32_t)g.S ? 1 : 0;
            }
            /* winner entries carry the set's generation: no memset between sub-batches (see winner_index) */
            const uint32_t max_gen = c->win_shift <= 28 ? (1u << (32 - c->win_shift)) - 1u : 0u;
            if (ln.win_gen + 1u > max_gen) {
                HIPCK(c, hipMemsetAsync(ln.winner, 0, (size_t)c->max_batch * S * sizeof(uint32_t), st));
                ln.win_gen = 0;
            }
            if (max_gen) ++ln.win_gen;
            b.win_tag = ln.win_gen;
            {   /* which frames are sorted up to a tail and can be read in place */
                ProfScope ps(c, K_PROBE, nb, st);
                launch_probe(g, b, nb, c->allow_stream, c->layout_hint, st);
            }
            /* The walks of the modes that read in place.  A mode's walk is launched unless the last mode_ttl looks at the
             * word k_verdict leaves in mapped host memory (the modes k_probe gave the frames of the last sub-batch whose
             * verdict has run; read without waiting; all ones while nothing is known) did not show the mode: an empty
             * launch costs 5-8 us of a 1 ms sub-batch, a frame whose walk was NOT launched fails k_verdict's count and is
             * redone the general way — the hint decides speed, not results.  Sticky since round 5 (round 4 followed the
             * last word alone: a directory that alternates layouts, or batches that cross a layout change, had every
             * frame of such a sub-batch redone; BEV_MODE_TTL=1 is that rule). */
            uint32_t seen = 0u;
            if (c->allow_stream && c->hint) {
                const uint32_t word = reinterpret_cast<volatile uint32_t *>(c->hint)[1];
                for (int m = 0; m < 8; ++m) {
                    c->mode_absent[m] = (word >> m) & 1u ? 0 : std::min(c->mode_absent[m] + 1, 1 << 20);
                    if (c->mode_absent[m] < c->mode_ttl) seen |= 1u << m;
                }
            }
            if (c->layout_hint) seen |= 1u << c->layout_hint; /* (the probe hands the hinted mode out whatever the last verdicts saw) */
            int rc = BEV_OK;
            if (c->allow_stream && ln.tail_list && (seen & (1u << kFrameStream))) /* frames k_probe found sorted up to a tail: read in place, verified */
                rc = walk(K_GATHER_GROUND, 2, kFrameStream);
            if (rc == BEV_OK && c->allow_stream && n_exact_s > 0 && (seen & (1u << kFrameStructured))) /* structured clouds (only a frame of exactly S records can be one) */
                rc = walk(K_WALK_STRUCTURED, 3, kFrameStructured);
            if (rc == BEV_OK && c->allow_stream && n_exact_s > 0 && (seen & (1u << kFrameColMajor))) /* ... or S returns in firing order */
                rc = walk(K_WALK_COLMAJOR, 4, kFrameColMajor);
            if (rc == BEV_OK && c->allow_stream && n_exact_s > 0 && ln.cm_par && (seen & (1u << kFrameColMajorGen))) /* ... from any start azimuth, in either direction, with staggered beams and no-return records */
                rc = walk(K_WALK_COLMAJOR_GEN, 5, kFrameColMajorGen);
            if (rc != BEV_OK) return rc;
            if (c->allow_stream) {
                ProfScope ps(c, K_VERDICT, nb, st);
                launch_verdict(g, b, nb, c->hint, st);
            }
            {   /* every other frame — general, or read in place and failed (normally none of a sorted sub-batch).  Thin
                 * launch while the sub-batch of the last verdict was read in place entirely (a hint that k_verdict
                 * leaves in mapped host memory — UINT32_MAX until the set's first verdict: wide —, read without waiting:
                 * it only chooses the launch shape) */
                const bool thin = c->allow_stream && c->hint && *reinterpret_cast<volatile uint32_t *>(c->hint) == 0u;
                ProfScope ps(c, K_ORDER_SCAN, nb, st);
                launch_order_scan(g, b, nb, max_pts, thin, st);
            }
            rc = walk(K_WALK_GENERAL, 0, kFrameGeneral);
            if (rc != BEV_OK) return rc;
        }
        if (fused) {
            c->pending.push_back(bev_ctx::Pending{b, nb, d_multi != nullptr, d_single != nullptr, d_gm ? d_gm + (size_t)f0 * S : nullptr, 1, q});
        } else {
            RoctxRange rb("bev:back (cell sums, resolve, rasters)");
            {
                ProfScope ps(c, K_CELL_SUMS, nb, st);
                launch_cell_sums(g, b, nb, st);
            }
            if (d_gm) {
                ProfScope ps(c, K_GROUND_MAT, nb, st);
                launch_ground_mat(g, b, d_gm + (size_t)f0 * S, nb, st);
            }
            {   /* phase C for the candidates: labels, codes of the un-grounded ones */
                ProfScope ps(c, K_GROUND_RESOLVE, nb, st);
                launch_ground_resolve(g, b, nb, st);
            }
            if (d_multi || d_single) {
                ProfScope ps(c, K_BEV_RASTER, nb, st);
                launch_bev_raster(g, b, d_multi != nullptr, d_single != nullptr, nb, st);
            }
        }
        c->last_sub_frames = nb;
        c->last_avg = ln.avg;
        c->last_ncode = ln.ncode;
        c->last_info = identity ? nullptr : ln.info;
        HIPCK(c, hipGetLastError());
    }
    if (fused) { /* (flush_pending joins as well) */
        int rc = flush ? flush_pending(c) : join_stage_streams(c);
        if (rc != BEV_OK) return rc;
    }
    if (!identity) {
        HIPCK(c, hipEventRecord(c->desc_done[ds], c->stream));
        c->desc_used[ds] = true;
    }
    return BEV_OK;
}

int ensure_staging(bev_ctx *c)
{
    if (c->staging_ready) return BEV_OK;
    const size_t S = (size_t)c->geo.S;
    const size_t per_frame = std::max(c->max_points, S);
    c->st_in_elems = per_frame * (size_t)c->max_batch;
    HIPCK(c, hipMalloc((void **)&c->st_in, c->st_in_elems * sizeof(bev_point_t)));
    HIPCK(c, hipMalloc((void **)&c->st_ordered, (size_t)c->max_batch * S * sizeof(bev_point_t)));
    HIPCK(c, hipMalloc((void **)&c->st_multi, (size_t)c->max_batch * c->multi_bytes));
    HIPCK(c, hipMalloc((void **)&c->st_single, (size_t)c->max_batch * c->single_bytes));
    HIPCK(c, hipMalloc((void **)&c->st_gm, (size_t)c->max_batch * S));
    HIPCK(c, hipStreamCreateWithFlags(&c->dl_stream, hipStreamNonBlocking));
    for (auto &e : c->out_ready) HIPCK(c, hipEventCreateWithFlags(&e, hipEventDisableTiming));
    c->downloader = new (std::nothrow) Downloader();
    if (!c->downloader) return BEV_ERR_OOM;
    c->downloader->start(c);
    c->staging_ready = true;
    return BEV_OK;
}

} // namespace

/* ======================================================================== */
extern "C" {

int bev_abi_version(void) { return BEV_ABI_VERSION; }

const char *bev_strerror(int status)
{
    switch (status) {
    case BEV_OK: return "ok";
    case BEV_ERR_INVALID_ARG: return "invalid argument";
    case BEV_ERR_NO_DEVICE: return "no usable HIP device (this library has no CPU path)";
    case BEV_ERR_HIP: return "HIP runtime error (see bev_last_error)";
    case BEV_ERR_OOM: return "out of device memory";
    case BEV_ERR_UNSUPPORTED: return "parameter combination not supported by the built kernels";
    case BEV_ERR_TOO_LARGE: return "batch or point count larger than the context was created for";
    default: return "unknown status";
    }
}

const char *bev_last_error(const bev_ctx_t *ctx) { return ctx ? ctx->last_error.c_str() : ""; }

int bev_params_for_sensor(const char *sensor, bev_params_t *out)
{
    if (!sensor || !out) return BEV_ERR_INVALID_ARG;
    /* parseSensorType: substring match, src/Utility.cpp:74-83; table :96-118 */
    if (strstr(sensor, "HDL_32E")) {
        out->n_scan = 32; out->horizon_scan = 1056; out->ground_upper_scan = 20; out->height_res = 0.5f;
    } else if (strstr(sensor, "HDL_64E")) {
        out->n_scan = 64; out->horizon_scan = 2083; out->ground_upper_scan = 50; out->height_res = 0.25f;
    } else if (strstr(sensor, "OS1_64")) {
        out->n_scan = 64; out->horizon_scan = 1024; out->ground_upper_scan = 31; out->height_res = 1.0f;
    } else {
        return BEV_ERR_INVALID_ARG;
    }
    out->interval = 1.0f;        /* BatchMultiBevGen.cpp:738 */
    out->max_range = 112;        /* :266 */
    out->n_layers = 24;          /* :268 */
    out->lidar_to_ground = 2.0f; /* :269 */
    return BEV_OK;
}

size_t bev_num_slots(const bev_params_t *p) { return p ? (size_t)p->n_scan * (size_t)p->horizon_scan : 0; }
size_t bev_multi_bytes(const bev_params_t *p)
{
    if (!p || validate_params(p) != BEV_OK) return 0;
    const size_t M = (size_t)mat_size_of(p);
    return (size_t)p->n_layers * M * M;
}
size_t bev_single_bytes(const bev_params_t *p)
{
    if (!p || validate_params(p) != BEV_OK) return 0;
    const size_t M = (size_t)mat_size_of(p);
    return M * M;
}

int bev_create(bev_ctx_t **out, int device, const bev_params_t *p, int max_batch, size_t max_points)
{
    if (!out || !p || max_batch < 1 || max_batch > 65535 || max_points >= 0xffffffffull) return BEV_ERR_INVALID_ARG;
    *out = nullptr;
    int rc = validate_params(p);
    if (rc != BEV_OK) return rc;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return BEV_ERR_NO_DEVICE;
    if (device < 0 || device >= ndev) return BEV_ERR_NO_DEVICE;

    bev_ctx *c = new (std::nothrow) bev_ctx();
    if (!c) return BEV_ERR_OOM;
    c->device = device;
    c->params = *p;
    fill_geometry(p, &c->geo);
    c->max_batch = max_batch;
    c->max_points = max_points;
    c->win_shift = 1;
    while (c->win_shift < 32 && (max_points >> c->win_shift) != 0) ++c->win_shift; /* index + 1 <= max_points */
    c->multi_bytes = bev_multi_bytes(p);
    c->single_bytes = bev_single_bytes(p);

    auto fail = [&](int code) {
        bev_destroy(c);
        return code;
    };
#define CK(expr)                                                                     \
    do {                                                                             \
        hipError_t e_ = (expr);                                                      \
        if (e_ != hipSuccess) {                                                      \
            int code_ = hip_fail(c, e_, #expr, __LINE__);                            \
            fprintf(stderr, "bev_create: %s\n", c->last_error.c_str());              \
            return fail(code_);                                                      \
        }                                                                            \
    } while (0)

    CK(hipSetDevice(device));
    CK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    for (int k = 0; k < kDescRing; ++k) {
        CK(hipEventCreateWithFlags(&c->desc_done[k], hipEventDisableTiming));
    }
    const size_t S = (size_t)c->geo.S, nb = (size_t)max_batch;
    c->codes_elems = std::max(std::max(max_points, S), (size_t)1024 * 1024); /* dense codes of one cloud, or the float BEV grid */
    {
        /* BEV_LANES=1: serial launches from the start (what bev_set_lanes(ctx, 1) switches to) */
        const char *e = getenv("BEV_LANES");
        c->fused = !(e && atoi(e) == 1);
        if (const char *ss = getenv("BEV_STAGE_STREAMS")) c->n_stage_streams = std::max(1, std::min(kMaxStageStreams, atoi(ss)));
        c->n_lanes = 4 * c->n_stage_streams;
        /* Frames whose points are in slot order up to a tail (a sweep written row by row with dropped returns ABSENT — none
         * of the reference's three selectors writes exactly that: KITTI's structured clouds and MulRan's firing order have
         * routes of their own, kFrameStructured / kFrameColMajor, Oxford's file order goes the general way) are read in place: no order
         * scan, no winner table, 5.6 MB less HBM traffic per HDL_64E frame.  k_probe decides per frame, the walk
         * verifies every point it consumes, a frame that fails is redone the general way: results never depend on the
         * mode.  BEV_STREAM=0 forces the general path for every frame. */
        const char *sm = getenv("BEV_STREAM");
        c->allow_stream = !(sm && atoi(sm) == 0);
        if (const char *mt = getenv("BEV_MODE_TTL")) c->mode_ttl = std::max(1, atoi(mt));
    }
    {   /* EQUAL priorities (profiles/r06_experiments.txt): the launches of two such streams share the chip workgroup by workgroup,
         * 396-400 k frames/s where different priorities (the higher stream's launch dispatched first, whole) gave 384-386 k
         * and one stream 368-370 k on the same box */
        int prio_least = 0, prio_greatest = 0;
        CK(hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest));
        for (int q = 0; q < kMaxStageStreams; ++q) {
            CK(hipStreamCreateWithPriority(&c->stage_st[q], hipStreamNonBlocking, prio_greatest));
            CK(hipEventCreateWithFlags(&c->stage_ev[q], hipEventDisableTiming));
        }
        CK(hipEventCreateWithFlags(&c->fork_ev, hipEventDisableTiming));
        CK(hipEventCreateWithFlags(&c->null_ev, hipEventDisableTiming));
    }
    CK(hipHostMalloc((void **)&c->hint, 2 * sizeof(uint32_t), hipHostMallocMapped));
    c->hint[0] = 0xffffffffu; /* nothing known yet: the first order scan is launched wide, */
    c->hint[1] = 0xffffffffu; /* ... every walk is launched */
    for (int l = 0; l < c->n_lanes; ++l) {
        Lane &ln = c->lanes[l];
        CK(hipMalloc((void **)&ln.desc, nb * sizeof(FrameDesc)));
        CK(hipMalloc((void **)&ln.info, nb * sizeof(FrameInfo)));
        CK(hipMemset(ln.info, 0, nb * sizeof(FrameInfo)));
        CK(hipMalloc((void **)&ln.est, nb * (size_t)c->geo.N * c->geo.strips * sizeof(uint32_t)));
        if (c->allow_stream && c->geo.N <= kStreamMaxRows && c->geo.N * c->geo.strips <= kTailBuckets) {
            CK(hipMalloc((void **)&ln.tail_list, nb * (size_t)c->geo.N * c->geo.strips * kTailCap * sizeof(uint32_t))); /* (lanes past a list's count fetch its word 0) */
            CK(hipMalloc((void **)&ln.tail_cnt, nb * (size_t)c->geo.N * c->geo.strips * sizeof(uint32_t)));
        }
        if (c->allow_stream && c->geo.N <= kCmMaxRows && c->geo.strips <= kCmMaxStrips) {
            CK(hipMalloc((void **)&ln.cm_par, nb * (size_t)kCmParWords * sizeof(int32_t)));
            CK(hipMalloc((void **)&ln.cm_sync, nb * (size_t)kCmSyncWords * sizeof(uint32_t)));
        }
        CK(hipMalloc((void **)&ln.winner, nb * S * sizeof(uint32_t)));
        CK(hipMemset(ln.winner, 0, nb * S * sizeof(uint32_t)));
        /* (+ one segment of slack: whole 64-slices are read past a short segment's count) */
        CK(hipMalloc((void **)&ln.cand, (nb * (size_t)c->geo.segs + 1) * kSeg * sizeof(uint2)));
        CK(hipMalloc((void **)&ln.ncand, nb * (size_t)c->geo.segs * sizeof(uint32_t)));
        CK(hipMalloc((void **)&ln.code_main, nb * (size_t)c->geo.emitters * c->geo.raster_bands * c->geo.code_stride * sizeof(uint32_t)));
        CK(hipMalloc((void **)&ln.ncode, nb * (size_t)c->geo.emitters * c->geo.raster_bands * sizeof(uint32_t)));
        CK(hipMalloc((void **)&ln.avg, nb * (size_t)bevx::kGridCells * sizeof(float)));
    }
    c->winner = c->lanes[0].winner;
    CK(hipMalloc((void **)&c->codes, c->codes_elems * sizeof(uint32_t))); /* single-cloud entry points */
    {   /* tables that depend on the parameters only, once per context: BEV bins of the ground grid's cell edges, x bin -> raster band */
        uint32_t tab[kCtxTabWords] = {};
        int *ex = reinterpret_cast<int *>(tab), *ey = ex + bevx::kGridRows;
        uint8_t *bt = reinterpret_cast<uint8_t *>(tab + bevx::kGridRows + bevx::kGridCols);
        for (int i = 0; i < bevx::kGridRows; ++i) ex[i] = bevx::cell_edge_bin(i, 75.0f, c->geo.rp);
        for (int i = 0; i < bevx::kGridCols; ++i) ey[i] = bevx::cell_edge_bin(i, 50.0f, c->geo.rp);
        for (int x = 0; x < c->geo.rp.mat_size && x < 512; ++x) bt[x] = (uint8_t)bevx::raster_band_of_nodiv(x, c->geo.rp);
        CK(hipMalloc((void **)&c->ctx_tab, sizeof tab));
        CK(hipMemcpy(c->ctx_tab, tab, sizeof tab, hipMemcpyHostToDevice));
    }
    /* >64 KiB dynamic LDS needs an explicit opt-in per kernel */
    CK(configure_kernels(c->geo));
    c->prof_pool.resize(kEventPairs);
    for (auto &s : c->prof_pool) {
        CK(hipEventCreate(&s.a));
        CK(hipEventCreate(&s.b));
    }
#undef CK
    *out = c;
    return BEV_OK;
}

void bev_destroy(bev_ctx_t *c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    c->pending.clear(); /* (stages never launched: their outputs were never waited for) */
    for (int q = 0; q < kMaxStageStreams; ++q) {
        if (c->stage_st[q]) (void)hipStreamSynchronize(c->stage_st[q]);
        if (c->stage_ev[q]) (void)hipEventDestroy(c->stage_ev[q]);
        if (c->stage_st[q]) (void)hipStreamDestroy(c->stage_st[q]);
    }
    if (c->fork_ev) (void)hipEventDestroy(c->fork_ev);
    if (c->null_ev) (void)hipEventDestroy(c->null_ev);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    for (int l = 0; l < kMaxLanes; ++l) {
        Lane &ln = c->lanes[l];
        void *ws[] = {ln.desc, ln.info, ln.est, ln.tail_list, ln.tail_cnt, ln.cm_par, ln.cm_sync, ln.winner, ln.cand, ln.ncand, ln.code_main, ln.ncode, ln.avg, ln.gm};
        for (void *p : ws)
            if (p) (void)hipFree(p);
    }
    if (c->hint) (void)hipHostFree(c->hint);
    if (c->downloader) {
        c->downloader->close();
        delete c->downloader;
        c->downloader = nullptr;
    }
    if (c->dl_stream) (void)hipStreamDestroy(c->dl_stream);
    for (auto e : c->out_ready)
        if (e) (void)hipEventDestroy(e);
    void *dev[] = {c->st_in, c->st_ordered, c->st_multi, c->st_single, c->st_gm, c->kitti_buf, c->codes, c->ctx_tab};
    for (void *p : dev)
        if (p) (void)hipFree(p);
    for (int k = 0; k < kDescRing; ++k) {
        if (c->h_desc[k]) (void)hipHostFree(c->h_desc[k]);
        if (c->desc_done[k]) (void)hipEventDestroy(c->desc_done[k]);
    }
    for (auto &s : c->prof_pool) {
        if (s.a) (void)hipEventDestroy(s.a);
        if (s.b) (void)hipEventDestroy(s.b);
    }
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

int bev_synchronize(bev_ctx_t *c)
{
    if (!c) return BEV_ERR_INVALID_ARG;
    HIPCK(c, hipSetDevice(c->device));
    int rc = flush_pending(c); /* the later stages of the last sub-batches */
    if (rc != BEV_OK) return rc;
    HIPCK(c, hipStreamSynchronize(c->stream));
    return BEV_OK;
}

int bev_process_device_resident(bev_ctx_t *c, int n_frames, const bev_point_t *d_pts, const uint64_t *h_offsets,
                                bev_point_t *d_ordered, uint8_t *d_multi, uint8_t *d_single, int8_t *d_ground_mat)
{
    if (!c || n_frames < 0 || !h_offsets || !d_ordered) return BEV_ERR_INVALID_ARG;
    if (n_frames > 0 && !d_pts && h_offsets[n_frames] != h_offsets[0]) return BEV_ERR_INVALID_ARG;
    /* (no fork: nothing of this call is on the context's stream, and waiting for it would be waiting for BOTH stage
     * streams' launches of the call before: the overlap across calls) */
    return run_pipeline(c, n_frames, d_pts, h_offsets, false, d_ordered, d_multi, d_single, d_ground_mat,
                        /*flush=*/false, 0, /*fork=*/false);
}

int bev_process_batch(bev_ctx_t *c, int n_frames, const bev_point_t *const *pts, const uint32_t *n_pts,
                      bev_point_t *const *ordered_out, uint8_t *const *multi_out, uint8_t *const *single_out,
                      int8_t *const *ground_mat_out)
{
    if (!c || n_frames < 0 || (n_frames > 0 && (!pts || !n_pts || !ordered_out))) return BEV_ERR_INVALID_ARG;
    for (int f = 0; f < n_frames; ++f) {
        if (n_pts[f] > c->max_points) return BEV_ERR_TOO_LARGE;
        if (n_pts[f] && !pts[f]) return BEV_ERR_INVALID_ARG;
    }
    if (n_frames == 0) return BEV_OK;
    HIPCK(c, hipSetDevice(c->device));
    { /* the later stages of sub-batches still in flight use the workspace this call is about to use */
        const int rc_ = flush_pending(c);
        if (rc_ != BEV_OK) return rc_;
    }
    int rc = ensure_staging(c);
    if (rc != BEV_OK) return rc;
    const size_t S = (size_t)c->geo.S;
    /* the output staging is used as two halves: chunk k is computed into half k % 2 while half (k - 1) % 2 drains */
    const int halves = c->max_batch >= 2 ? 2 : 1;
    const int chunk = c->max_batch / halves;
    bool any_gm = false;
    for (int f = 0; f < n_frames && ground_mat_out; ++f) any_gm = any_gm || ground_mat_out[f] != nullptr;

    Downloader &dl = *c->downloader;
    dl.begin_call();
    std::vector<uint64_t> off;
    int k = 0;
    rc = BEV_OK;
    for (int f0 = 0; f0 < n_frames && rc == BEV_OK; f0 += chunk, ++k) {
        const int nb = std::min(chunk, n_frames - f0), half = k % halves;
        RoctxRange rc_range("bev_process_batch: chunk (upload, pipeline, hand-over to the downloader)");
        off.assign((size_t)nb + 1, 0);
        hipError_t e = hipSuccess;
        for (int f = 0; f < nb; ++f) off[f + 1] = off[f] + n_pts[f0 + f];
        for (int f = 0; f < nb && e == hipSuccess;) { /* the input staging is free again: stream order */
            /* clouds that follow each other in host memory go up in one copy (the staging is packed the same way) */
            int last = f;
            while (last + 1 < nb && n_pts[f0 + last] && pts[f0 + last + 1] == pts[f0 + last] + n_pts[f0 + last]) ++last;
            const uint64_t n = off[last + 1] - off[f];
            if (n)
                e = hipMemcpyAsync(c->st_in + off[f], pts[f0 + f], (size_t)n * sizeof(bev_point_t), hipMemcpyHostToDevice,
                                   c->stream);
            f = last + 1;
        }
        if (e != hipSuccess) {
            rc = hip_fail(c, e, "hipMemcpyAsync (host -> device staging)", __LINE__);
            break;
        }
        dl.wait_finished(k + 1 - halves); /* this half's previous tenant has reached the caller's buffers */
        const size_t base = (size_t)half * chunk;
        rc = run_pipeline(c, nb, c->st_in, off.data(), false, c->st_ordered + base * S,
                          multi_out ? c->st_multi + base * c->multi_bytes : nullptr,
                          single_out ? c->st_single + base * c->single_bytes : nullptr, any_gm ? c->st_gm + base * S : nullptr,
                          /*flush=*/true, /*sub_frames: two sub-batches, so that a chunk's stages overlap*/ nb >= 8 ? (nb + 1) / 2 : 0);
        if (rc != BEV_OK) break;
        e = hipEventRecord(c->out_ready[half], c->stream);
        if (e != hipSuccess) {
            rc = hip_fail(c, e, "hipEventRecord", __LINE__);
            break;
        }
        dl.push({f0, nb, half, ordered_out, multi_out, single_out, any_gm ? ground_mat_out : nullptr, chunk});
    }
    dl.wait_finished(k); /* every chunk that was handed over has reached the caller's buffers */
    if (rc == BEV_OK && dl.err != hipSuccess) rc = hip_fail(c, dl.err, "device -> host copy", __LINE__);
    if (rc != BEV_OK) (void)hipDeviceSynchronize();
    return rc;
}

int bev_host_alloc(void **out, size_t bytes)
{
    if (!out) return BEV_ERR_INVALID_ARG;
    *out = nullptr;
    if (bytes == 0) return BEV_OK;
    const hipError_t e = hipHostMalloc(out, bytes, hipHostMallocDefault);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        *out = nullptr;
        return e == hipErrorOutOfMemory ? BEV_ERR_OOM : BEV_ERR_HIP;
    }
    return BEV_OK;
}

int bev_host_free(void *p)
{
    if (!p) return BEV_OK;
    return hipHostFree(p) == hipSuccess ? BEV_OK : BEV_ERR_HIP;
}

int bev_order_cloud(bev_ctx_t *c, const bev_point_t *pts, uint32_t n_pts, bev_point_t *ordered_out)
{
    if (!c || !ordered_out || (n_pts && !pts)) return BEV_ERR_INVALID_ARG;
    if (n_pts > c->max_points) return BEV_ERR_TOO_LARGE;
    HIPCK(c, hipSetDevice(c->device));
    { /* the later stages of sub-batches still in flight use the workspace this call is about to use */
        const int rc_ = flush_pending(c);
        if (rc_ != BEV_OK) return rc_;
    }
    int rc = ensure_staging(c);
    if (rc != BEV_OK) return rc;
    const Geometry &g = c->geo;
    const size_t S = (size_t)g.S;
    int ds = 0;
    rc = acquire_desc(c, 1, &ds);
    if (rc != BEV_OK) return rc;
    c->h_desc[ds][0] = FrameDesc{0, n_pts, 0};
    if (n_pts)
        HIPCK(c, hipMemcpyAsync(c->st_in, pts, (size_t)n_pts * sizeof(bev_point_t), hipMemcpyHostToDevice, c->stream));
    BatchPtrs b{};
    b.pts = c->st_in;
    b.frames = c->d_desc[ds];
    b.winner = c->winner;
    b.win_shift = c->win_shift; /* tag 0 on a cleared table */
    b.ordered = c->st_ordered;
    HIPCK(c, hipMemsetAsync(c->winner, 0, S * sizeof(uint32_t), c->stream));
    {
        ProfScope ps(c, K_ORDER_SCAN, 1);
        launch_order_scan(g, b, 1, n_pts, false, c->stream); /* b.info == nullptr: the whole cloud */
    }
    {
        ProfScope ps(c, K_GATHER_ONLY, 1);
        launch_gather_only(g, b, 1, c->stream);
    }
    HIPCK(c, hipGetLastError());
    HIPCK(c, hipMemcpyAsync(ordered_out, c->st_ordered, S * sizeof(bev_point_t), hipMemcpyDeviceToHost, c->stream));
    HIPCK(c, hipEventRecord(c->desc_done[ds], c->stream));
    c->desc_used[ds] = true;
    HIPCK(c, hipStreamSynchronize(c->stream));
    return BEV_OK;
}

int bev_mark_ground(bev_ctx_t *c, bev_point_t *ordered, int8_t *ground_mat_out)
{
    if (!c || !ordered) return BEV_ERR_INVALID_ARG;
    HIPCK(c, hipSetDevice(c->device));
    { /* the later stages of sub-batches still in flight use the workspace this call is about to use */
        const int rc_ = flush_pending(c);
        if (rc_ != BEV_OK) return rc_;
    }
    int rc = ensure_staging(c);
    if (rc != BEV_OK) return rc;
    const size_t S = (size_t)c->geo.S;
    HIPCK(c, hipMemcpyAsync(c->st_in, ordered, S * sizeof(bev_point_t), hipMemcpyHostToDevice, c->stream));
    rc = run_pipeline(c, 1, c->st_in, nullptr, true, c->st_ordered, nullptr, nullptr,
                      ground_mat_out ? c->st_gm : nullptr);
    if (rc != BEV_OK) return rc;
    HIPCK(c, hipMemcpyAsync(ordered, c->st_ordered, S * sizeof(bev_point_t), hipMemcpyDeviceToHost, c->stream));
    if (ground_mat_out) HIPCK(c, hipMemcpyAsync(ground_mat_out, c->st_gm, S, hipMemcpyDeviceToHost, c->stream));
    HIPCK(c, hipStreamSynchronize(c->stream));
    return BEV_OK;
}

static int raster_cloud(bev_ctx_t *c, const bev_point_t *cloud, uint32_t n, uint8_t *multi_out, uint8_t *single_out)
{
    if (!c || (n && !cloud)) return BEV_ERR_INVALID_ARG;
    if ((size_t)n > std::max(c->max_points, (size_t)c->geo.S)) return BEV_ERR_TOO_LARGE;
    HIPCK(c, hipSetDevice(c->device));
    { /* the later stages of sub-batches still in flight use the workspace this call is about to use */
        const int rc_ = flush_pending(c);
        if (rc_ != BEV_OK) return rc_;
    }
    int rc = ensure_staging(c);
    if (rc != BEV_OK) return rc;
    if (n) HIPCK(c, hipMemcpyAsync(c->st_in, cloud, (size_t)n * sizeof(bev_point_t), hipMemcpyHostToDevice, c->stream));
    {
        ProfScope ps(c, K_CLOUD_CODES, 1);
        launch_cloud_codes(c->geo, c->st_in, n, c->codes, c->stream);
    }
    {
        ProfScope ps(c, K_BEV_RASTER, 1);
        launch_bev_raster_dense(c->geo, c->codes, n, multi_out ? c->st_multi : nullptr, single_out ? c->st_single : nullptr,
                                c->stream);
    }
    HIPCK(c, hipGetLastError());
    if (multi_out) HIPCK(c, hipMemcpyAsync(multi_out, c->st_multi, c->multi_bytes, hipMemcpyDeviceToHost, c->stream));
    if (single_out)
        HIPCK(c, hipMemcpyAsync(single_out, c->st_single, c->single_bytes, hipMemcpyDeviceToHost, c->stream));
    HIPCK(c, hipStreamSynchronize(c->stream));
    return BEV_OK;
}

int bev_multi_bev(bev_ctx_t *c, const bev_point_t *cloud, uint32_t n, uint8_t *multi_out)
{
    if (!multi_out) return BEV_ERR_INVALID_ARG;
    return raster_cloud(c, cloud, n, multi_out, nullptr);
}
int bev_single_bev(bev_ctx_t *c, const bev_point_t *cloud, uint32_t n, uint8_t *single_out)
{
    if (!single_out) return BEV_ERR_INVALID_ARG;
    return raster_cloud(c, cloud, n, nullptr, single_out);
}

size_t bev_project_out_points(int kind, uint32_t n)
{
    switch (kind) {
    case BEV_PROJECT_MULRAN_OS1_64:
    case BEV_PROJECT_OXFORD_HDL_32E: return n;
    case BEV_PROJECT_KITTI_HDL_64E: return (size_t)bevx::kKittiRows * bevx::kKittiCols;
    default: return 0;
    }
}

namespace {
/* carve the KITTI workspace out of one buffer (256-byte aligned pieces) */
int kitti_workspace(bev_ctx *c, uint32_t n, KittiWork &w)
{
    const size_t blocks = ((size_t)n + bevx::kKittiBlock - 1) / bevx::kKittiBlock;
    auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
    const size_t o_col = up(sizeof(KittiHeader)), o_cnt = o_col + up((size_t)n * 4), o_pos = o_cnt + up(blocks * 4),
                 o_win = o_pos + up(blocks * bevx::kKittiListCap * 4),
                 total = o_win + up((size_t)bevx::kKittiRows * bevx::kKittiCols * 4);
    if (n > c->kitti_points || !c->kitti_buf) {
        if (c->kitti_buf) HIPCK(c, hipFree(c->kitti_buf));
        c->kitti_buf = nullptr;
        c->kitti_points = 0;
        hipError_t e = hipMalloc(&c->kitti_buf, total);
        if (e == hipErrorOutOfMemory) {
            (void)hipGetLastError();
            c->kitti_buf = nullptr;
            return BEV_ERR_OOM;
        }
        HIPCK(c, e);
        c->kitti_points = n;
    }
    char *base = static_cast<char *>(c->kitti_buf);
    w.hdr = reinterpret_cast<KittiHeader *>(base);
    w.col = reinterpret_cast<int32_t *>(base + o_col);
    w.cnt = reinterpret_cast<uint32_t *>(base + o_cnt);
    w.pos = reinterpret_cast<uint32_t *>(base + o_pos);
    w.winner = reinterpret_cast<uint32_t *>(base + o_win);
    return BEV_OK;
}
} // namespace

int bev_project_xyzi(bev_ctx_t *c, int kind, const float *xyzi, uint32_t n, bev_point_t *out)
{
    const size_t n_out = bev_project_out_points(kind, n);
    if (!c || (n && !xyzi) || (n_out && !out)) return BEV_ERR_INVALID_ARG;
    if (kind != BEV_PROJECT_MULRAN_OS1_64 && kind != BEV_PROJECT_OXFORD_HDL_32E && kind != BEV_PROJECT_KITTI_HDL_64E)
        return BEV_ERR_INVALID_ARG;
    if ((size_t)n > std::max(c->max_points, (size_t)c->geo.S)) return BEV_ERR_TOO_LARGE;
    if (n_out == 0) return BEV_OK;
    HIPCK(c, hipSetDevice(c->device));
    { /* the later stages of sub-batches still in flight use the workspace this call is about to use */
        const int rc_ = flush_pending(c);
        if (rc_ != BEV_OK) return rc_;
    }
    int rc = ensure_staging(c);
    if (rc != BEV_OK) return rc;
    /* raw floats are staged in the ordered-cloud staging buffer (16 B per point fit its 32 B per slot) */
    float *d_raw = reinterpret_cast<float *>(c->st_ordered);
    if ((size_t)n * 16 > (size_t)c->max_batch * c->geo.S * sizeof(bev_point_t)) return BEV_ERR_TOO_LARGE;
    if (n_out > c->st_in_elems) return BEV_ERR_TOO_LARGE;
    if (n) HIPCK(c, hipMemcpyAsync(d_raw, xyzi, (size_t)n * 16, hipMemcpyHostToDevice, c->stream));
    if (kind == BEV_PROJECT_KITTI_HDL_64E) c->layout_hint = BEV_LAYOUT_STRUCTURED; /* what this call writes is a structured cloud (bev_set_layout_hint) */
    if (kind == BEV_PROJECT_KITTI_HDL_64E) {
        if (n == 0) { /* defined here, undefined in the reference: an empty file gives the all-zero structured cloud */
            HIPCK(c, hipMemsetAsync(c->st_in, 0, n_out * sizeof(bev_point_t), c->stream));
        } else {
            KittiWork w{};
            rc = kitti_workspace(c, n, w);
            if (rc != BEV_OK) return rc;
            HIPCK(c, hipMemsetAsync(w.winner, 0, n_out * sizeof(uint32_t), c->stream));
            ProfScope ps(c, K_PROJECT, 1);
            launch_project_kitti(d_raw, n, w, c->st_in, c->stream);
        }
    } else {
        ProfScope ps(c, K_PROJECT, 1);
        launch_project(kind, d_raw, n, c->st_in, c->stream);
    }
    HIPCK(c, hipGetLastError());
    HIPCK(c, hipMemcpyAsync(out, c->st_in, n_out * sizeof(bev_point_t), hipMemcpyDeviceToHost, c->stream));
    HIPCK(c, hipStreamSynchronize(c->stream));
    return BEV_OK;
}

size_t bev_float_bev_size(float interval)
{
    if (!(interval > 0.0f)) return 0;
    /* static int MAT_SIZE = MAX_RANGE*2 / interval + 1;  BatchCloudManip.cpp:210 */
    const int M = bevx::cvtt_f32((float)200 / interval + 1);
    return (M >= 1 && M <= 1024) ? (size_t)M : 0;
}

int bev_float_bev(bev_ctx_t *c, const bev_point_t *cloud, uint32_t n, float interval, int skip_label0, float *out)
{
    if (!c || !out || (n && !cloud)) return BEV_ERR_INVALID_ARG;
    const size_t M = bev_float_bev_size(interval);
    if (M == 0) return BEV_ERR_UNSUPPORTED;
    if ((size_t)n > std::max(c->max_points, (size_t)c->geo.S)) return BEV_ERR_TOO_LARGE;
    if (M * M > c->codes_elems) return BEV_ERR_UNSUPPORTED; /* the grid borrows the single-cloud code buffer */
    HIPCK(c, hipSetDevice(c->device));
    { /* the later stages of sub-batches still in flight use the workspace this call is about to use */
        const int rc_ = flush_pending(c);
        if (rc_ != BEV_OK) return rc_;
    }
    int rc = ensure_staging(c);
    if (rc != BEV_OK) return rc;
    float *grid = reinterpret_cast<float *>(c->codes);
    if (n) HIPCK(c, hipMemcpyAsync(c->st_in, cloud, (size_t)n * sizeof(bev_point_t), hipMemcpyHostToDevice, c->stream));
    HIPCK(c, hipMemsetAsync(grid, 0, M * M * sizeof(float), c->stream));
    {
        ProfScope ps(c, K_FLOAT_BEV, 1);
        launch_float_bev(c->st_in, n, interval, (int)M, skip_label0 != 0, grid, c->stream);
    }
    HIPCK(c, hipGetLastError());
    HIPCK(c, hipMemcpyAsync(out, grid, M * M * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    HIPCK(c, hipStreamSynchronize(c->stream));
    return BEV_OK;
}

int bev_transform_cloud(bev_ctx_t *c, const bev_point_t *cloud, uint32_t n, const float *m, bev_point_t *out)
{
    if (!c || !m || (n && (!cloud || !out))) return BEV_ERR_INVALID_ARG;
    if ((size_t)n > std::max(c->max_points, (size_t)c->geo.S)) return BEV_ERR_TOO_LARGE;
    if (n == 0) return BEV_OK;
    HIPCK(c, hipSetDevice(c->device));
    { /* the later stages of sub-batches still in flight use the workspace this call is about to use */
        const int rc_ = flush_pending(c);
        if (rc_ != BEV_OK) return rc_;
    }
    int rc = ensure_staging(c);
    if (rc != BEV_OK) return rc;
    /* in place in the input staging (every thread reads and writes its own point) */
    HIPCK(c, hipMemcpyAsync(c->st_in, cloud, (size_t)n * sizeof(bev_point_t), hipMemcpyHostToDevice, c->stream));
    {
        ProfScope ps(c, K_TRANSFORM, 1);
        launch_transform(c->st_in, n, m, c->st_in, c->stream);
    }
    HIPCK(c, hipGetLastError());
    HIPCK(c, hipMemcpyAsync(out, c->st_in, (size_t)n * sizeof(bev_point_t), hipMemcpyDeviceToHost, c->stream));
    HIPCK(c, hipStreamSynchronize(c->stream));
    return BEV_OK;
}

void bev_yaw_translate_matrix(float tx, float ty, float tz, float yaw_deg, float *m)
{
    if (!m) return;
    /* CloudManip.cpp:119-128: Affine3f = Identity; translation() << t; rotate(AngleAxisf(theta, UnitZ())) with
     * theta = yaw_deg / 180.0f * M_PI stored to float; Eigen's AngleAxis::toRotationMatrix for the axis (0, 0, 1) gives
     * [[c, -s, 0], [s, c, 0], [0, 0, (1 - c) + c]] (every other term of its formulas is an exact 0) */
    const float theta = (float)((double)(yaw_deg / 180.0f) * 3.14159265358979323846);
    const float s = sinf(theta), c = cosf(theta);
    const float one_minus_c = 1.0f - c;
    m[0] = c;    m[1] = 0.0f - s; m[2] = 0.0f;  m[3] = tx;
    m[4] = s;    m[5] = c;        m[6] = 0.0f;  m[7] = ty;
    m[8] = 0.0f; m[9] = 0.0f;     m[10] = one_minus_c + c; m[11] = tz;
}

int bev_set_layout_hint(bev_ctx_t *c, int layout)
{
    if (!c || (layout != BEV_LAYOUT_UNKNOWN && layout != BEV_LAYOUT_STRUCTURED && layout != BEV_LAYOUT_FIRING_ORDER)) return BEV_ERR_INVALID_ARG;
    static_assert(BEV_LAYOUT_STRUCTURED == (int)kFrameStructured && BEV_LAYOUT_FIRING_ORDER == (int)kFrameColMajor, "the hint is the mode k_probe hands out");
    c->layout_hint = layout;
    return BEV_OK;
}

int bev_set_lanes(bev_ctx_t *c, int n)
{
    if (!c || n < 1) return BEV_ERR_INVALID_ARG;
    int rc = bev_synchronize(c);
    if (rc != BEV_OK) return rc;
    c->fused = n > 1;
    return c->fused ? c->n_lanes : 1;
}

int bev_profile_enable(bev_ctx_t *c, int on)
{
    if (!c) return BEV_ERR_INVALID_ARG;
    if (!on && c->prof_on) {
        int rc = prof_flush(c);
        if (rc != BEV_OK) return rc;
    }
    c->prof_on = on != 0;
    return BEV_OK;
}
int bev_profile_reset(bev_ctx_t *c)
{
    if (!c) return BEV_ERR_INVALID_ARG;
    int rc = prof_flush(c);
    if (rc != BEV_OK) return rc;
    for (int k = 0; k < K_COUNT; ++k) {
        c->prof_ms[k] = 0;
        c->prof_launches[k] = 0;
        c->prof_frames[k] = 0;
    }
    return BEV_OK;
}
int bev_profile_get(bev_ctx_t *c, bev_kernel_stat_t *out, int cap)
{
    if (!c || (!out && cap > 0)) return BEV_ERR_INVALID_ARG;
    int rc = prof_flush(c);
    if (rc != BEV_OK) return rc;
    int n = 0;
    for (int k = 0; k < K_COUNT; ++k) {
        if (c->prof_launches[k] == 0) continue;
        if (n < cap) {
            out[n].name = kernel_name(k);
            out[n].launches = c->prof_launches[k];
            out[n].total_ms = c->prof_ms[k];
            out[n].frames = c->prof_frames[k];
        }
        ++n;
    }
    return n;
}

int bev_debug_get_cell_avg(bev_ctx_t *c, int first_frame, int n_frames, float *out)
{
    if (!c || !out || first_frame < 0 || n_frames < 0 || first_frame + n_frames > c->last_sub_frames)
        return BEV_ERR_INVALID_ARG;
    HIPCK(c, hipSetDevice(c->device));
    { /* the later stages of sub-batches still in flight use the workspace this call is about to use */
        const int rc_ = flush_pending(c);
        if (rc_ != BEV_OK) return rc_;
    }
    HIPCK(c, hipStreamSynchronize(c->stream));
    if (!c->last_avg) return BEV_ERR_INVALID_ARG;
    HIPCK(c, hipMemcpy(out, c->last_avg + (size_t)first_frame * bevx::kGridCells,
                       (size_t)n_frames * bevx::kGridCells * sizeof(float), hipMemcpyDeviceToHost));
    return BEV_OK;
}

int bev_debug_get_frame_info(bev_ctx_t *c, int first_frame, int n_frames, uint32_t *out)
{
    if (!c || !out || first_frame < 0 || n_frames < 0 || first_frame + n_frames > c->last_sub_frames || !c->last_info)
        return BEV_ERR_INVALID_ARG;
    HIPCK(c, hipSetDevice(c->device));
    { /* the later stages of sub-batches still in flight use the workspace this call is about to use */
        const int rc_ = flush_pending(c);
        if (rc_ != BEV_OK) return rc_;
    }
    HIPCK(c, hipStreamSynchronize(c->stream));
    HIPCK(c, hipMemcpy(out, c->last_info + first_frame, (size_t)n_frames * sizeof(FrameInfo), hipMemcpyDeviceToHost));
    return BEV_OK;
}

int bev_debug_get_code_overflow(bev_ctx_t *c, int first_frame, int n_frames, uint32_t *out)
{
    if (!c || !out || first_frame < 0 || n_frames < 0 || first_frame + n_frames > c->last_sub_frames || !c->last_ncode)
        return BEV_ERR_INVALID_ARG;
    HIPCK(c, hipSetDevice(c->device));
    { /* the later stages of sub-batches still in flight use the workspace this call is about to use */
        const int rc_ = flush_pending(c);
        if (rc_ != BEV_OK) return rc_;
    }
    HIPCK(c, hipStreamSynchronize(c->stream));
    const size_t per = (size_t)c->geo.emitters * c->geo.raster_bands;
    std::vector<uint32_t> counts((size_t)n_frames * per);
    HIPCK(c, hipMemcpy(counts.data(), c->last_ncode + (size_t)first_frame * per, counts.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
    for (int f = 0; f < n_frames; ++f) {
        out[f] = 0u;
        for (size_t k = 0; k < per; ++k) out[f] += counts[(size_t)f * per + k] > c->geo.code_cap ? 1u : 0u;
    }
    return BEV_OK;
}

int bev_debug_angle_predicate(bev_ctx_t *c, const float *dx, const float *dy, const float *dz, uint8_t *out, size_t n)
{
    if (!c || !dx || !dy || !dz || !out) return BEV_ERR_INVALID_ARG;
    if (n == 0) return BEV_OK;
    HIPCK(c, hipSetDevice(c->device));
    float *d = nullptr;
    uint8_t *o = nullptr;
    HIPCK(c, hipMalloc((void **)&d, 3 * n * sizeof(float)));
    hipError_t e = hipMalloc((void **)&o, n);
    if (e != hipSuccess) {
        (void)hipFree(d);
        return hip_fail(c, e, "hipMalloc", __LINE__);
    }
    int rc = BEV_OK;
    auto ck = [&](hipError_t err, int line) {
        if (err != hipSuccess && rc == BEV_OK) rc = hip_fail(c, err, "bev_debug_angle_predicate", line);
    };
    ck(hipMemcpy(d, dx, n * sizeof(float), hipMemcpyHostToDevice), __LINE__);
    ck(hipMemcpy(d + n, dy, n * sizeof(float), hipMemcpyHostToDevice), __LINE__);
    ck(hipMemcpy(d + 2 * n, dz, n * sizeof(float), hipMemcpyHostToDevice), __LINE__);
    if (rc == BEV_OK) {
        launch_angle_debug(d, d + n, d + 2 * n, o, n, c->stream);
        ck(hipGetLastError(), __LINE__);
        ck(hipStreamSynchronize(c->stream), __LINE__);
        ck(hipMemcpy(out, o, n, hipMemcpyDeviceToHost), __LINE__);
    }
    (void)hipFree(d);
    (void)hipFree(o);
    return rc;
}

} /* extern "C" */
