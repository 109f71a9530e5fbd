/*
 * BatchMultiBevGen.h — host-side mirror of the reference's hot-path surface
 * (BatchMultiBevGen.h / BatchMultiBevGen.cpp): the same free functions, argument
 * order and side effects, implemented as thin callers of the C ABI
 * (include/bev_mi355x.h).  No PCL / OpenCV: see PointCloud.h for the two
 * container types.  Like the reference, the free functions use file-scope state
 * (sensor_params_, output directories) and are not re-entrant.
 *
 * The reference's `class BatchMultiBevGen {}` is an empty stub
 * (BatchMultiBevGen.h:102-104); here it is a real facade that owns the GPU
 * context and processes whole batches of files with one ABI call.
 */
#ifndef BEV_HOST_BATCHMULTIBEVGEN_H
#define BEV_HOST_BATCHMULTIBEVGEN_H

#include <string>
#include <utility>
#include <vector>

#include "PointCloud.h"
#include "Utility.h"

extern SensorParams sensor_params_;                         /* BatchMultiBevGen.cpp:37 */
extern std::vector<std::pair<int, int>> four_neighbor_iterator_; /* :29 */

void initDirectories(std::string keyframes_root_dir);       /* :39-71  */
void setNeighbors();                                        /* :73-84  */
void getOrderedCloud(pcl::PointCloud<pcl::PointXYZIRCT>::Ptr &input_cloud,
                     pcl::PointCloud<pcl::PointXYZIRCT>::Ptr &output_cloud); /* :94-117 */
void markGroundPoints(pcl::PointCloud<pcl::PointXYZIRCT>::Ptr &output_cloud, cv::Mat &ground_mat); /* :119-252 */
void computeAndSaveMultiBev(pcl::PointCloud<pcl::PointXYZIRCT>::Ptr cloud, std::string str_cloud_idx,
                            float interval = 1.0f);         /* :261-321 */
void computeAndSaveSingleBev(pcl::PointCloud<pcl::PointXYZIRCT>::Ptr cloud, std::string str_cloud_idx,
                             float interval = 1.0f);        /* :331-373 */
void getPcdFileNames(std::string path, std::vector<std::string> &filenames); /* :469-494 */
std::pair<int, int> getBelongingGrid(const pcl::PointCloud<PointType>::Ptr &cloud_ptr, int point_index); /* .h:73-99 */

/* GPU used by the free functions above (default 0; call before the first of them). */
void setBevDevice(int device);
/* Releases the context the free functions created lazily. */
void shutdownBev();

class BatchMultiBevGen {
public:
    /* sensor_type as on the command line ("HDL_32E", "HDL_64E", "OS1_64") */
    /* max_points: input points per cloud the GPU context is sized for at first; a larger cloud makes the context grow */
    BatchMultiBevGen(const std::string &keyframes_root_dir, const std::string &sensor_type, int device = 0,
                     int batch_frames = 32, std::size_t max_points = (std::size_t)4 << 20);
    ~BatchMultiBevGen();
    BatchMultiBevGen(const BatchMultiBevGen &) = delete;
    BatchMultiBevGen &operator=(const BatchMultiBevGen &) = delete;

    bool ok() const { return ctx_ != nullptr; }
    const SensorParams &sensorParams() const { return params_; }
    /* Step 1 of main() (BatchMultiBevGen.cpp:727-757) for files[first, first+count):
     * load -> order -> ground -> BEVs -> write .bin/.png/.csv/.pcd.  Returns the
     * accumulated milliseconds of the reference's timed region (GPU path + BEV file
     * writes; PCD load/save excluded, :732-752). */
    double processFiles(const std::vector<std::string> &files, std::size_t first, std::size_t count,
                        bool write_png = true, bool verbose = true);
    /* frames whose outputs are missing because the GPU path failed on them (after a frame-by-frame retry of their
     * batch); the tool exits non-zero when this is not 0 */
    std::size_t failedFrames() const { return failed_frames_; }

private:
    bool createContext(std::size_t max_points);
    struct bev_ctx *ctx_ = nullptr;
    int device_ = 0;
    std::size_t max_points_ = 0, initial_max_points_ = 0;
    std::size_t failed_frames_ = 0;
    SensorParams params_{};
    std::string root_;
    int batch_frames_;
};

#endif
