/*
 * CloudManip.h — host mirror of the float max-height BEV of the reference's older
 * tools: saveAsMat of batch_cloud_manip
 * (BatchCloudManip.cpp:201-239) and of cloud_manip (CloudManip.cpp:79-109), plus the
 * yaw + translation rigid transform cloud_manip applies (CloudManip.cpp:119-128).
 * The reference's `class CloudManip {}` / `class BatchCloudManip {}` are empty stubs
 * (CloudManip.h:9-11, BatchCloudManip.h:103-105); here they carry these functions.
 * The interactive PCLVisualizer part of cloud_manip is out of scope.
 */
#ifndef BEV_HOST_CLOUDMANIP_H
#define BEV_HOST_CLOUDMANIP_H

#include <string>

#include "PointCloud.h"

class BatchCloudManip {
public:
    /* saveAsMat(cloud, <dir>/<name>, interval): writes <name>.csv ("%.4g" values) and <name>.png;
     * points with label == 0 are skipped (BatchCloudManip.cpp:218).  Returns the grid. */
    static cv::Mat saveAsMat(pcl::PointCloud<pcl::PointXYZIRCT>::Ptr cloud, std::string filename_sin_appendix,
                             float interval = 2.0f);
};

class CloudManip {
public:
    /* saveAsMat(cloud, mat_filename, interval): no label test (CloudManip.cpp:88); writes
     * mat_filename and mat_filename + ".png". */
    static cv::Mat saveAsMat(pcl::PointCloud<PointType>::Ptr cloud, std::string mat_filename, float interval = 2.0f);
    /* pcl::transformPointCloud with Affine3f = translate(tx,ty,tz) * rotZ(yaw_deg / 180.0f * M_PI)
     * (CloudManip.cpp:119-128); x, y, z are transformed, every other field is copied. */
    static void transformYawTranslate(const pcl::PointCloud<PointType> &in, pcl::PointCloud<PointType> &out, float tx,
                                      float ty, float tz, float yaw_deg);
};

#endif
