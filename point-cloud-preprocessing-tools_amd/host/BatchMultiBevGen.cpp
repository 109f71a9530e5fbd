#include "BatchMultiBevGen.h"

#include <dirent.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <thread>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>

#include "../csrc/bev_exact.h"
#include "FileFormats.h"

SensorParams sensor_params_{0, 0, 0, 0.0f};
std::vector<std::pair<int, int>> four_neighbor_iterator_;

namespace {

std::string output_bvm_dir_;
std::string output_multi_bvm_bin_dir_;
std::string output_multi_bvm_img_dir_;
std::string output_single_bvm_img_dir_;
std::string output_single_bvm_csv_dir_;

bev_ctx_t *g_ctx = nullptr;
SensorParams g_ctx_params{0, 0, 0, 0.0f};
int g_device = 0;

bev_params_t to_bev_params(const SensorParams &sp)
{
    bev_params_t bp;
    bp.n_scan = sp.N_SCAN;
    bp.horizon_scan = sp.Horizon_SCAN;
    bp.ground_upper_scan = sp.GROUND_UPPER_SCAN;
    bp.height_res = sp.HEIGHT_RES;
    bp.interval = 1.0f;        /* BatchMultiBevGen.cpp:738 */
    bp.max_range = 112;        /* :266 */
    bp.n_layers = 24;          /* :268 */
    bp.lidar_to_ground = 2.0f; /* :269 */
    return bp;
}

/* the context behind the free functions follows the global sensor_params_ */
bev_ctx_t *context()
{
    const bool same = g_ctx && g_ctx_params.N_SCAN == sensor_params_.N_SCAN &&
                      g_ctx_params.Horizon_SCAN == sensor_params_.Horizon_SCAN &&
                      g_ctx_params.GROUND_UPPER_SCAN == sensor_params_.GROUND_UPPER_SCAN &&
                      g_ctx_params.HEIGHT_RES == sensor_params_.HEIGHT_RES;
    if (same) return g_ctx;
    shutdownBev();
    bev_params_t bp = to_bev_params(sensor_params_);
    const int rc = bev_create(&g_ctx, g_device, &bp, 1, (size_t)4 << 20);
    if (rc != BEV_OK) {
        std::cerr << "bev_create failed: " << bev_strerror(rc) << "\n";
        g_ctx = nullptr;
        return nullptr;
    }
    g_ctx_params = sensor_params_;
    return g_ctx;
}

void report(const char *what, int rc)
{
    if (rc != BEV_OK) std::cerr << what << " failed: " << bev_strerror(rc) << " " << (g_ctx ? bev_last_error(g_ctx) : "") << "\n";
}

void remove_tree(const std::string &path)
{
    DIR *d = opendir(path.c_str());
    if (d) {
        while (dirent *e = readdir(d)) {
            if (!strcmp(e->d_name, ".") || !strcmp(e->d_name, "..")) continue;
            const std::string p = path + "/" + e->d_name;
            struct stat st;
            if (lstat(p.c_str(), &st) == 0 && S_ISDIR(st.st_mode)) remove_tree(p);
            else unlink(p.c_str());
        }
        closedir(d);
        rmdir(path.c_str());
    } else {
        unlink(path.c_str());
    }
}

void make_dirs(const std::string &path)
{
    std::string cur;
    for (size_t i = 0; i < path.size(); ++i) {
        cur += path[i];
        if (path[i] == '/' || i + 1 == path.size()) mkdir(cur.c_str(), 0777);
    }
}

/* "rm -rf" + "mkdir -p" of the reference (BatchMultiBevGen.cpp:49-50 etc.), without system() */
void recreate_dir(std::string path)
{
    while (path.size() > 1 && path.back() == '/') path.pop_back();
    remove_tree(path);
    make_dirs(path);
}

constexpr int kMat = 224, kLayers = 24;

void write_multi_outputs(const std::string &name, const uint8_t *multi, bool write_png)
{
    /* .bin: layer-major, then row, 224 bytes per row (:307-314) */
    const std::string bin = output_multi_bvm_bin_dir_ + name + ".bin";
    if (!bevio::writeFile(bin, multi, (size_t)kLayers * kMat * kMat)) std::cerr << "Can not open file: " << bin << "\n";
    if (!write_png) return;
    const std::string img_dir = output_multi_bvm_img_dir_ + name + "/";
    if (access(img_dir.c_str(), 0) == -1) make_dirs(img_dir); /* :303-306 */
    for (int l = 0; l < kLayers; ++l) {
        char nm[16];
        std::snprintf(nm, sizeof nm, "%02d.png", l); /* :316-317 */
        bevio::writePngGray8(img_dir + nm, multi + (size_t)l * kMat * kMat, kMat, kMat);
    }
}

void write_single_outputs(const std::string &name, const uint8_t *single, bool write_png)
{
    if (write_png) bevio::writePngGray8(output_single_bvm_img_dir_ + name + ".png", single, kMat, kMat); /* :359-361 */
    const std::string csv = output_single_bvm_csv_dir_ + name + ".csv";                               /* :365-372 */
    const std::string text = bevio::formatCsvU8(single, kMat, kMat);
    if (!bevio::writeFile(csv, text.data(), text.size())) std::cerr << "Faied to export csv formatted BEV file: " << csv;
}

} // namespace

void bevhost_recreate_dir(const std::string &dir) { recreate_dir(dir); }
bev_ctx_t *bevhost_context() { return context(); }

void setBevDevice(int device) { g_device = device; }
void shutdownBev()
{
    if (g_ctx) bev_destroy(g_ctx);
    g_ctx = nullptr;
}

void initDirectories(std::string keyframes_root_dir)
{
    if (keyframes_root_dir.empty() || keyframes_root_dir.back() != '/') keyframes_root_dir += "/";
    output_bvm_dir_ = keyframes_root_dir + "output_multi_bev/";                 /* :48-50 */
    recreate_dir(output_bvm_dir_);
    output_multi_bvm_bin_dir_ = keyframes_root_dir + "output_multi_bev/binary/"; /* :53-55 */
    recreate_dir(output_multi_bvm_bin_dir_);
    output_multi_bvm_img_dir_ = keyframes_root_dir + "output_multi_bev/image/";  /* :58-60 */
    recreate_dir(output_multi_bvm_img_dir_);
    output_single_bvm_csv_dir_ = keyframes_root_dir + "output_single_bev/csv/";  /* :63-65 */
    recreate_dir(output_single_bvm_csv_dir_);
    output_single_bvm_img_dir_ = keyframes_root_dir + "output_single_bev/image/"; /* :68-70 */
    recreate_dir(output_single_bvm_img_dir_);
}

void setNeighbors()
{
    /* kept for source compatibility (:73-84); the kernels have the 4-neighbourhood built in */
    four_neighbor_iterator_ = {{-1, 0}, {0, 1}, {0, -1}, {1, 0}};
}

void getOrderedCloud(pcl::PointCloud<pcl::PointXYZIRCT>::Ptr &input_cloud,
                     pcl::PointCloud<pcl::PointXYZIRCT>::Ptr &output_cloud)
{
    output_cloud->resize((size_t)sensor_params_.N_SCAN * sensor_params_.Horizon_SCAN); /* :98 */
    bev_ctx_t *c = context();
    if (!c) return;
    report("getOrderedCloud", bev_order_cloud(c, reinterpret_cast<const bev_point_t *>(input_cloud->points.data()),
                                              (uint32_t)input_cloud->points.size(),
                                              reinterpret_cast<bev_point_t *>(output_cloud->points.data())));
}

void markGroundPoints(pcl::PointCloud<pcl::PointXYZIRCT>::Ptr &output_cloud, cv::Mat &ground_mat)
{
    ground_mat = cv::Mat::zeros(sensor_params_.N_SCAN, sensor_params_.Horizon_SCAN, cv::CV_8S); /* :123 */
    bev_ctx_t *c = context();
    if (!c) return;
    if (output_cloud->points.size() != (size_t)sensor_params_.N_SCAN * sensor_params_.Horizon_SCAN) {
        std::cerr << "markGroundPoints: cloud is not an ordered N_SCAN x Horizon_SCAN cloud\n";
        return;
    }
    report("markGroundPoints", bev_mark_ground(c, reinterpret_cast<bev_point_t *>(output_cloud->points.data()),
                                               ground_mat.ptr<int8_t>()));
}

void computeAndSaveMultiBev(pcl::PointCloud<pcl::PointXYZIRCT>::Ptr cloud, std::string str_cloud_idx, float interval)
{
    if (interval != 1.0f) std::cerr << "computeAndSaveMultiBev: MAT_SIZE is frozen at interval 1.0 (static, :266-267)\n";
    bev_ctx_t *c = context();
    if (!c) return;
    std::vector<uint8_t> multi((size_t)kLayers * kMat * kMat);
    const int rc = bev_multi_bev(c, reinterpret_cast<const bev_point_t *>(cloud->points.data()),
                                 (uint32_t)cloud->points.size(), multi.data());
    report("computeAndSaveMultiBev", rc);
    if (rc == BEV_OK) write_multi_outputs(str_cloud_idx, multi.data(), true);
}

void computeAndSaveSingleBev(pcl::PointCloud<pcl::PointXYZIRCT>::Ptr cloud, std::string str_cloud_idx, float interval)
{
    if (interval != 1.0f) std::cerr << "computeAndSaveSingleBev: MAT_SIZE is frozen at interval 1.0 (static, :336-337)\n";
    bev_ctx_t *c = context();
    if (!c) return;
    std::vector<uint8_t> single((size_t)kMat * kMat);
    const int rc = bev_single_bev(c, reinterpret_cast<const bev_point_t *>(cloud->points.data()),
                                  (uint32_t)cloud->points.size(), single.data());
    report("computeAndSaveSingleBev", rc);
    if (rc == BEV_OK) write_single_outputs(str_cloud_idx, single.data(), true);
}

void getPcdFileNames(std::string path, std::vector<std::string> &filenames)
{
    DIR *dir = opendir(path.c_str());
    if (!dir) {
        std::cerr << "Folder doesn't Exist!" << std::endl; /* :473-476 */
        return;
    }
    while (dirent *e = readdir(dir)) {
        const std::string name = e->d_name;
        const size_t dot = name.find_last_of('.');
        if (name.substr(dot + 1) != "pcd") continue; /* :482-484 (also drops "." and "..") */
        filenames.push_back(path.back() == '/' ? path + name : path + "/" + name);
    }
    closedir(dir);
    std::sort(filenames.begin(), filenames.end()); /* :493 */
}

std::pair<int, int> getBelongingGrid(const pcl::PointCloud<PointType>::Ptr &cloud_ptr, int point_index)
{
    const auto &p = cloud_ptr->points[point_index];
    const int cell = bevx::ground_cell(p.x, p.y);
    return std::make_pair(cell / bevx::kGridCols, cell % bevx::kGridCols);
}

/* ------------------------------------------------------------------------ */
BatchMultiBevGen::BatchMultiBevGen(const std::string &keyframes_root_dir, const std::string &sensor_type, int device,
                                   int batch_frames, std::size_t max_points)
    : root_(keyframes_root_dir), batch_frames_(std::max(1, batch_frames))
{
    if (root_.empty() || root_.back() != '/') root_ += "/";
    params_ = getSensorParams(parseSensorType(sensor_type));
    if (params_.N_SCAN <= 0) return;
    device_ = device;
    initial_max_points_ = std::max<std::size_t>(1, max_points);
    (void)createContext(initial_max_points_);
}

/* (re)creates the GPU context for clouds of up to max_points input points */
bool BatchMultiBevGen::createContext(std::size_t max_points)
{
    if (ctx_) bev_destroy(ctx_);
    ctx_ = nullptr;
    bev_params_t bp = to_bev_params(params_);
    bev_ctx_t *c = nullptr;
    const int rc = bev_create(&c, device_, &bp, batch_frames_, max_points);
    if (rc != BEV_OK) {
        std::cerr << "bev_create failed: " << bev_strerror(rc) << "\n";
        return false;
    }
    ctx_ = c;
    max_points_ = max_points;
    return true;
}

BatchMultiBevGen::~BatchMultiBevGen()
{
    if (ctx_) bev_destroy(ctx_);
}

namespace {
/* The file work of a batch (PCD parse, PNG deflate, CSV text, PCD write) is independent per frame: a few host threads
 * (BEV_IO_THREADS, default min(16, cores)) share it; the reference does all of it on one thread. */
int io_threads()
{
    static const int n = [] {
        const char *e = std::getenv("BEV_IO_THREADS");
        int v = e ? std::atoi(e) : (int)std::min(16u, std::max(1u, std::thread::hardware_concurrency()));
        return std::max(1, std::min(256, v));
    }();
    return n;
}
template <class F>
void parallel_frames(int n, F fn)
{
    const int workers = std::min(io_threads(), n);
    if (workers <= 1) {
        for (int i = 0; i < n; ++i) fn(i);
        return;
    }
    std::atomic<int> next{0};
    std::vector<std::thread> pool;
    for (int w = 0; w < workers; ++w)
        pool.emplace_back([&] {
            for (int i = next.fetch_add(1); i < n; i = next.fetch_add(1)) fn(i);
        });
    for (auto &t : pool) t.join();
}
} // namespace

double BatchMultiBevGen::processFiles(const std::vector<std::string> &files, std::size_t first, std::size_t count,
                                      bool write_png, bool verbose)
{
    if (!ctx_) return 0.0;
    const size_t S = (size_t)params_.N_SCAN * params_.Horizon_SCAN;
    const std::string non_ground_dir = root_ + "non_ground_point_cloud/";
    double timed_ms = 0.0;
    std::vector<pcl::PointCloud<pcl::PointXYZIRCT>> in(batch_frames_), ordered(batch_frames_);
    std::vector<std::vector<uint8_t>> multi(batch_frames_), single(batch_frames_);
    for (size_t b0 = first; b0 < first + count; b0 += batch_frames_) {
        const int nb = (int)std::min<size_t>(batch_frames_, first + count - b0);
        std::vector<const bev_point_t *> pts(nb);
        std::vector<uint32_t> npts(nb);
        std::vector<bev_point_t *> ord(nb);
        std::vector<uint8_t *> mo(nb), so(nb);
        std::vector<std::string> names(nb);
        parallel_frames(nb, [&](int i) {
            in[i].clear();
            if (bevio::loadPCDFile(files[b0 + i], in[i]) != 0) std::cerr << "Failed to load " << files[b0 + i] << "\n"; /* :730 */
        });
        for (int i = 0; i < nb; ++i) {
            const std::string &fn = files[b0 + i];
            ordered[i].resize(S);
            multi[i].resize((size_t)kLayers * kMat * kMat);
            single[i].resize((size_t)kMat * kMat);
            pts[i] = reinterpret_cast<const bev_point_t *>(in[i].points.data());
            npts[i] = (uint32_t)in[i].points.size();
            ord[i] = reinterpret_cast<bev_point_t *>(ordered[i].points.data());
            mo[i] = multi[i].data();
            so[i] = single[i].data();
            const size_t start = fn.find_last_of('/') + 1; /* :739-742 */
            const size_t end = fn.find_last_of('.');
            names[i] = fn.substr(start, end - start);
        }
        /* a cloud larger than the context was created for: grow the context instead of dropping the batch (the
         * reference processes every file whatever its size) */
        uint32_t largest = 0;
        for (int i = 0; i < nb; ++i) largest = std::max(largest, npts[i]);
        if ((std::size_t)largest > max_points_) {
            std::size_t want = max_points_;
            while (want < (std::size_t)largest) want *= 2;
            if (!createContext(std::min<std::size_t>(want, 0xfffffffeull))) {
                failed_frames_ += (std::size_t)nb;
                (void)createContext(initial_max_points_);
                if (!ctx_) return timed_ms;
                continue;
            }
        }
        {   /* What the PCD parser can tell the library about the layout (bev_set_layout_hint): when every cloud of the batch
             * has exactly S records and a few of them, spread over the first cloud, are each their slot's point or all-zero
             * — what kitti_point_cloud_select writes, KittiPointCloudSelect.cpp:206-207,240 — the clouds are announced as
             * structured and the library does not sample them; anything else: it looks by itself.  A wrong guess here costs
             * time only: the walk checks every record. */
            bool structured = nb > 0;
            for (int i = 0; i < nb; ++i) structured = structured && (std::size_t)npts[i] == S;
            for (std::size_t k = 0; structured && k < 64; ++k) {
                const std::size_t at = k * (S / 64) + k; /* (an odd walk over rows and columns) */
                if (at >= S) break;
                const bev_point_t &q = pts[0][at];
                const bool zero = q.x == 0.f && q.y == 0.f && q.z == 0.f && q.intensity == 0.f && q.row == 0 && q.col == 0;
                structured = zero || (std::size_t)q.row * (std::size_t)params_.Horizon_SCAN + q.col == at;
            }
            (void)bev_set_layout_hint(ctx_, structured ? BEV_LAYOUT_STRUCTURED : BEV_LAYOUT_UNKNOWN);
        }
        const auto t0 = std::chrono::steady_clock::now();
        std::vector<char> done(nb, 1);
        int rc = bev_process_batch(ctx_, nb, pts.data(), npts.data(), ord.data(), mo.data(), so.data(), nullptr);
        if (rc != BEV_OK) {
            /* one bad frame must not take the other frames of its batch with it: retry frame by frame, count what
             * still fails; main() exits non-zero when anything failed */
            std::cerr << "bev_process_batch failed: " << bev_strerror(rc) << " " << bev_last_error(ctx_)
                      << "; retrying the batch frame by frame\n";
            for (int i = 0; i < nb; ++i) {
                rc = bev_process_batch(ctx_, 1, &pts[i], &npts[i], &ord[i], &mo[i], &so[i], nullptr);
                if (rc != BEV_OK) {
                    std::cerr << "Failed to process " << files[b0 + i] << ": " << bev_strerror(rc) << " " << bev_last_error(ctx_) << "\n";
                    done[i] = 0;
                    ++failed_frames_;
                }
            }
        }
        if (verbose)
            for (int i = 0; i < nb; ++i)
                if (done[i]) std::cout << "Converting file: " << names[i] << "\n"; /* :744 */
        parallel_frames(nb, [&](int i) {
            if (!done[i]) return;
            write_multi_outputs(names[i], multi[i].data(), write_png);
            write_single_outputs(names[i], single[i].data(), write_png);
        });
        timed_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        parallel_frames(nb, [&](int i) { /* :755-756: the labelled (not filtered) ordered cloud */
            if (done[i]) bevio::savePCDFileBinary(non_ground_dir + names[i] + ".pcd", ordered[i]);
        });
    }
    return timed_ms;
}
