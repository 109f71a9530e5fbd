/*
 * batch_multi_bev_gen <keyframes_root_dir> <sensor_type>
 *
 * Same command line, directory layout and stdout lines as the reference's tool
 * (BatchMultiBevGen.cpp:664-771); the per-file loop (:727-757) runs on MI355X
 * through the C ABI in batches.  Extra, optional environment:
 *   BEV_DEVICES=N   use GPUs 0..N-1 of this node (one host thread + one context
 *                   per GPU, contiguous shards of the sorted file list)
 *   BEV_DEVICE_MAP=a,b,...  the GPU of each of those N ranks instead of 0..N-1; an ordinal may repeat (0,0: two ranks, two
 *                   contexts, two shards on ONE GPU — the sharded path on a one-GPU machine)
 *   BEV_BATCH=B     frames per bev_process_batch call (default 32)
 *   BEV_MAX_POINTS=P input points per cloud the GPU contexts are sized for at first (default 4 Mi; larger clouds make
 *                   a context grow)
 *   BEV_NO_PNG=1    skip the 25 PNG files per frame
 *   BEV_IO_THREADS=T host threads sharing the per-frame file work of a batch (PCD parse, PNG, CSV, PCD write; default 16)
 */
#include <chrono>
#include <cstdint>
#include <cstdlib>
#include <iostream>
#include <thread>

#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include "BatchMultiBevGen.h"
#include "LabelStep.h"

void bevhost_recreate_dir(const std::string &dir); /* BatchMultiBevGen.cpp (host): rm -rf + mkdir -p */

int main(int argc, char **argv)
{
    if (argc < 3 || argv[1] == nullptr || argv[2] == nullptr) {
        std::cout << "Usage: " << (argc > 0 ? argv[0] : "batch_multi_bev_gen") << " [keyframes_root_dir] [sensor_type]\n\n"
                  << "[keyframes_root_dir] should be organized as follows: \n"
                  << "[keyframes_root_dir]\n"
                  << "  keyframe_point_cloud/  <- selected point clouds in pcd format, one per frame\n"
                  << "  keyframe_pose.csv      <- 6-DoF pose for each frame\n\n"
                  << "[sensor_type] could be HDL_32E, HDL_64E or OS1_64. \n\n"
                  << "Writes non_ground_point_cloud/, output_multi_bev/{binary,image}/, output_single_bev/{csv,image}/\n"
                  << "and keyframe_label.csv under [keyframes_root_dir].\n";
        return 1;
    }
    std::string root(argv[1]);
    if (root.empty() || root.back() != '/') root.append("/");
    const std::string pcd_dir = root + "keyframe_point_cloud/";
    const std::string pose_file = root + "keyframe_pose.csv";
    const std::string label_file = root + "keyframe_label.csv";

    std::vector<std::string> files;
    getPcdFileNames(pcd_dir, files);
    setNeighbors();
    initDirectories(root);
    bevhost_recreate_dir(root + "non_ground_point_cloud/"); /* :704-705 */

    const SensorType st = parseSensorType(std::string(argv[2]));
    sensor_params_ = getSensorParams(st);
    if (sensor_params_.N_SCAN <= 0) return 1; /* the reference would go on with garbage parameters */
    std::cout << "Using sensor_type " << argv[2] << ", with params: " << printSensorParams(sensor_params_) << "\n";

    const int n_dev = std::max(1, std::atoi(std::getenv("BEV_DEVICES") ? std::getenv("BEV_DEVICES") : "1"));
    const int batch = std::max(1, std::atoi(std::getenv("BEV_BATCH") ? std::getenv("BEV_BATCH") : "32"));
    const bool png = std::getenv("BEV_NO_PNG") == nullptr;
    const long long max_pts_env = std::getenv("BEV_MAX_POINTS") ? std::atoll(std::getenv("BEV_MAX_POINTS")) : 0;
    const size_t max_pts = max_pts_env > 0 ? (size_t)max_pts_env : ((size_t)4 << 20);

    /* Step 1: frames are independent -> contiguous shards of the sorted list, one GPU each.  GPU 0 owns
     * the frame-range table; the other GPUs get it by an RCCL broadcast (over xGMI on a multi-GPU node) —
     * the only inter-GPU communication of the whole tool. */
    std::vector<double> ms(n_dev, 0.0);
    std::vector<int> bad(n_dev, 0);
    std::vector<size_t> failed(n_dev, 0);
    std::vector<std::thread> workers;
    const size_t F = files.size();
    std::vector<int64_t> ranges(2 * (size_t)n_dev, 0);
    /* rank d runs on GPU dev_of[d]: 0..N-1, or what BEV_DEVICE_MAP lists.  Ranks may share a GPU (BEV_DEVICE_MAP=0,0):
     * every rank still has its own host thread, context, streams and shard; RCCL spans the DISTINCT GPUs (it refuses
     * a device twice), and ranks that share one read their row from that GPU's copy of the table. */
    std::vector<int> dev_of(n_dev);
    for (int d = 0; d < n_dev; ++d) dev_of[d] = d;
    if (const char *map = std::getenv("BEV_DEVICE_MAP")) {
        std::vector<int> listed;
        for (const char *p = map; *p;) {
            char *end = nullptr;
            const long v = std::strtol(p, &end, 10);
            if (end == p || v < 0) { listed.clear(); break; }
            listed.push_back((int)v);
            p = *end == ',' ? end + 1 : end;
            if (*end != ',' && *end != '\0') { listed.clear(); break; }
        }
        if ((int)listed.size() != n_dev) {
            std::cerr << "BEV_DEVICE_MAP must list " << n_dev << " GPU ordinals separated by commas (BEV_DEVICES=" << n_dev << ")\n";
            return 1;
        }
        dev_of = listed;
    }
    std::vector<int> uniq;                 /* distinct GPUs, in order of first use: uniq[0] holds rank 0 */
    std::vector<int> uidx(n_dev, 0);       /* rank -> index into uniq */
    for (int d = 0; d < n_dev; ++d) {
        size_t k = 0;
        while (k < uniq.size() && uniq[k] != dev_of[d]) ++k;
        if (k == uniq.size()) uniq.push_back(dev_of[d]);
        uidx[d] = (int)k;
    }
    {
        const int n_u = (int)uniq.size();
        std::vector<int64_t> table(2 * (size_t)n_dev);
        for (int d = 0; d < n_dev; ++d) {
            table[2 * d] = (int64_t)(F * d / n_dev);
            table[2 * d + 1] = (int64_t)(F * (d + 1) / n_dev) - table[2 * d];
        }
        std::vector<ncclComm_t> comms(n_u);
        std::vector<hipStream_t> streams(n_u);
        std::vector<int64_t *> bufs(n_u, nullptr);
        bool ok = ncclCommInitAll(comms.data(), n_u, uniq.data()) == ncclSuccess;
        for (int u = 0; ok && u < n_u; ++u) {
            ok = hipSetDevice(uniq[u]) == hipSuccess && hipStreamCreate(&streams[u]) == hipSuccess &&
                 hipMalloc((void **)&bufs[u], table.size() * sizeof(int64_t)) == hipSuccess;
            if (ok && u == 0)
                ok = hipMemcpy(bufs[0], table.data(), table.size() * sizeof(int64_t), hipMemcpyHostToDevice) == hipSuccess;
        }
        if (ok) {
            ncclGroupStart();
            for (int u = 0; u < n_u; ++u)
                ok = ok && ncclBroadcast(bufs[u], bufs[u], table.size(), ncclInt64, 0, comms[u], streams[u]) == ncclSuccess;
            ncclGroupEnd();
        }
        for (int u = 0; ok && u < n_u; ++u)
            ok = hipSetDevice(uniq[u]) == hipSuccess && hipStreamSynchronize(streams[u]) == hipSuccess;
        for (int d = 0; ok && d < n_dev; ++d) /* every rank reads ITS row from the copy of the table on ITS GPU */
            ok = hipSetDevice(dev_of[d]) == hipSuccess &&
                 hipMemcpy(&ranges[2 * d], bufs[uidx[d]] + 2 * d, 2 * sizeof(int64_t), hipMemcpyDeviceToHost) == hipSuccess;
        for (int u = 0; u < n_u; ++u) {
            if (bufs[u]) { (void)hipSetDevice(uniq[u]); (void)hipFree(bufs[u]); (void)hipStreamDestroy(streams[u]); }
        }
        if (!ok) {
            std::cerr << "no usable HIP device / RCCL broadcast of the frame ranges failed (ranks: " << n_dev << ", GPUs: " << n_u
                      << "); there is no CPU path\n";
            return 1;
        }
        for (int u = 0; u < n_u; ++u) ncclCommDestroy(comms[u]);
        (void)hipSetDevice(uniq[0]);
    }
    for (int d = 0; d < n_dev; ++d) {
        const size_t first = (size_t)ranges[2 * d], count = (size_t)ranges[2 * d + 1];
        workers.emplace_back([&, d, first, count]() {
            BatchMultiBevGen gen(root, argv[2], dev_of[d], batch, max_pts);
            if (!gen.ok()) { bad[d] = 1; return; }
            ms[d] = gen.processFiles(files, first, count, png, n_dev == 1);
            failed[d] = gen.failedFrames();
        });
    }
    for (auto &w : workers) w.join();
    for (int d = 0; d < n_dev; ++d)
        if (bad[d]) return 1;
    size_t n_failed = 0;
    for (size_t v : failed) n_failed += v;
    double total_ms = 0;
    for (double v : ms) total_ms += v;
    std::cout << "[TIME] Average preprocessing and BEV generation: " << (F ? total_ms / (double)F : 0.0) << "\n";

    /* Step 2 (:762-765) */
    bool ok = false;
    std::vector<Pose6f> poses = readKeyframePose(pose_file, &ok);
    if (!ok) return 1; /* the reference exit(1)s here */
    if (poses.empty()) {
        std::cerr << "no keyframe poses\n";
        return 1;
    }
    std::vector<int32_t> major = selectMajorFrames(poses);
    std::vector<LabelType> labels = getKeyFrameLabel(poses, major);
    if (!saveLabels(labels, label_file)) return 1;
    if (n_failed) { /* outputs of those frames are missing: not a success, whatever else was written */
        std::cerr << n_failed << " of " << F << " frames failed on the GPU path\n";
        return 1;
    }
    std::cout << "Done. " << std::endl;
    return 0;
}
