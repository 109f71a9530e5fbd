/*
 * cloud_manip <input.pcd> <trans_x> <trans_y> <trans_z> <yaw_degrees>
 *
 * CloudManip.cpp:111-161 without its interactive PCLVisualizer loop (:143-158, out of scope): loads the cloud,
 * applies translate(t) * rotZ(yaw) (:119-128), writes the float max-height BEV of the input and of the transformed
 * cloud (saveAsMat, no ground filter: <name>_input.csv (+ .png), <name>_output.csv (+ .png), :136-137) and both
 * clouds as binary PCD (<name>_input.pcd, <name>_output.pcd, :139-140) into the current directory, <name> being the
 * last path component of the input INCLUDING its extension (:130-131).  Transform and rasters run on MI355X.
 */
#include <cstdlib>
#include <iostream>
#include <string>

#include "BatchMultiBevGen.h"
#include "CloudManip.h"
#include "FileFormats.h"

int main(int argc, char **argv)
{
    if (argc < 6) { /* the reference dereferences argv[1..5] unchecked */
        std::cout << "Usage: " << (argc > 0 ? argv[0] : "cloud_manip") << " <input.pcd> <trans_x> <trans_y> <trans_z> <yaw_degrees>\n";
        return 1;
    }
    pcl::PointCloud<PointType>::Ptr cloud_input(new pcl::PointCloud<PointType>());
    pcl::PointCloud<PointType>::Ptr cloud_output(new pcl::PointCloud<PointType>());
    const std::string input_filename(argv[1]);
    if (bevio::loadPCDFile(input_filename, *cloud_input) != 0) {
        std::cerr << "Can not read " << input_filename << "\n";
        return 1;
    }
    float t[4];
    for (int k = 0; k < 4; ++k) {
        try {
            t[k] = std::stof(argv[2 + k]);
        } catch (const std::exception &) { /* std::stof throws in the reference too (uncaught there) */
            std::cerr << "not a number: " << argv[2 + k] << "\n";
            return 1;
        }
    }
    const float theta = (float)((double)(t[3] / 180.0f) * 3.14159265358979323846); /* :124 */
    std::cout << "rotating yaw radiance: " << theta << "\n";                        /* :125 */

    /* any sensor table entry will do: the context is only used for its float-BEV and transform entry points */
    sensor_params_ = getSensorParams(SensorType::HDL_64E);
    CloudManip::transformYawTranslate(*cloud_input, *cloud_output, t[0], t[1], t[2], t[3]);

    const size_t slash = input_filename.find_last_of('/');
    const std::string short_name = slash == std::string::npos ? input_filename : input_filename.substr(slash + 1);
    const float interval_res = 1.0f;                                                /* :134 */
    CloudManip::saveAsMat(cloud_input, short_name + "_input.csv", interval_res);
    CloudManip::saveAsMat(cloud_output, short_name + "_output.csv", interval_res);
    bevio::savePCDFileBinary(short_name + "_input.pcd", *cloud_input);
    bevio::savePCDFileBinary(short_name + "_output.pcd", *cloud_output);
    shutdownBev();
    return 0;
}
