/*
 * batch_cloud_manip <keyframes_root_dir>
 *
 * The older fork of the hot path (BatchCloudManip.cpp:269-335): HDL-64E constants hard-coded (N_SCAN = 64,
 * Horizon_SCAN = 2083, groundScanInd = 50; :13-14, :85), per file: load -> getOrderedCloud -> markGroundPoints ->
 * saveAsMat (float max-height BEV, 201 x 201 at interval 1.0: <root>/output_bvm/<name>.csv + .png) -> labelled cloud
 * to <root>/non_ground_point_cloud/<name>.pcd.  Same command line, directory tree and stdout lines; order, ground
 * segmentation and the raster run on MI355X through the C ABI.
 * One difference is deliberate: the reference's getOrderedCloud of this tool has no bounds test (:55-62), so a point
 * with row >= 64 or col >= 2083 writes outside the cloud there; here such points are dropped.
 */
#include <chrono>
#include <iostream>

#include "BatchMultiBevGen.h"
#include "CloudManip.h"
#include "FileFormats.h"

void bevhost_recreate_dir(const std::string &dir); /* BatchMultiBevGen.cpp (host): rm -rf + mkdir -p */

int main(int argc, char **argv)
{
    if (argc < 2 || argv[1] == nullptr) {
        std::cout << "Usage: " << (argc > 0 ? argv[0] : "batch_cloud_manip") << " <keyframes_root_dir>" << std::endl; /* :271-274 */
        return 1;
    }
    std::string root(argv[1]);
    if (root.empty() || root.back() != '/') root.append("/");
    const std::string pcd_dir = root + "keyframe_point_cloud/";          /* :276-277 */
    const std::string non_ground_dir = root + "non_ground_point_cloud/"; /* :279-280 */
    bevhost_recreate_dir(non_ground_dir);                                /* :283-284 */

    std::vector<std::string> files;
    getPcdFileNames(pcd_dir, files);                                     /* :286-287 */
    setNeighbors();                                                      /* :289 */
    const std::string bvm_dir = root + "output_bvm/";                    /* :292-295 */
    bevhost_recreate_dir(bvm_dir);

    sensor_params_ = getSensorParams(SensorType::HDL_64E);               /* the tool's constants: 64 x 2083, 50 ground rings */
    double total_ms = 0;
    for (const std::string &input_filename : files) {                    /* :300-328 */
        pcl::PointCloud<pcl::PointXYZIRCT>::Ptr cloud_unordered(new pcl::PointCloud<pcl::PointXYZIRCT>());
        pcl::PointCloud<pcl::PointXYZIRCT>::Ptr cloud_ordered(new pcl::PointCloud<pcl::PointXYZIRCT>());
        if (bevio::loadPCDFile(input_filename, *cloud_unordered) != 0) std::cerr << "Can not read " << input_filename << "\n";

        const auto t0 = std::chrono::system_clock::now();
        cv::Mat ground_mat;
        getOrderedCloud(cloud_unordered, cloud_ordered);
        markGroundPoints(cloud_ordered, ground_mat);

        const float interval_res = 1.0f;                                 /* :311 */
        const size_t start_pos = input_filename.find_last_of('/') + 1;
        const size_t end_pos = input_filename.find_last_of('.');
        const std::string short_name = input_filename.substr(start_pos, end_pos - start_pos);
        std::cout << "Converting file: " << short_name << "\n";
        BatchCloudManip::saveAsMat(cloud_ordered, bvm_dir + short_name, interval_res); /* :319 */

        const auto t1 = std::chrono::system_clock::now();
        const double ms = (double)std::chrono::duration_cast<std::chrono::microseconds>(t1 - t0).count() * 1e-3;
        std::cout << "[TIME] Preprocessing and BEV generation: " << ms << "ms. \n" << std::endl; /* :323 */
        total_ms += ms;

        bevio::savePCDFileBinary(non_ground_dir + short_name + ".pcd", *cloud_ordered); /* :327 */
    }
    std::cout << "[TIME] Average preprocessing and BEV generation: " << (files.empty() ? 0.0 : total_ms / (double)files.size())
              << "\n";
    std::cout << "Done. " << std::endl;
    shutdownBev();
    return 0;
}
