#include "CloudManip.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <iostream>
#include <vector>

#include "BatchMultiBevGen.h"
#include "FileFormats.h"

bev_ctx_t *bevhost_context(); /* BatchMultiBevGen.cpp (host): the lazily created context of the free functions */

namespace {

/* the float grid comes from the GPU (bev_float_bev); only file framing happens here */
cv::Mat float_bev(const pcl::PointCloud<pcl::PointXYZIRCT> &cloud, float interval, bool skip_label0)
{
    cv::Mat grid;
    const size_t M = bev_float_bev_size(interval);
    bev_ctx_t *c = bevhost_context();
    if (!c || M == 0) return grid;
    grid.create((int)M, (int)M, cv::CV_32F);
    const int rc = bev_float_bev(c, reinterpret_cast<const bev_point_t *>(cloud.points.data()),
                                 (uint32_t)cloud.points.size(), interval, skip_label0 ? 1 : 0, grid.ptr<float>());
    if (rc != BEV_OK) std::cerr << "saveAsMat: " << bev_strerror(rc) << "\n";
    return grid;
}

/* cv::Formatter FMT_CSV with set32fPrecision(4): "%.4g" values, ", " between, "\n" per row
 * (from memory of OpenCV's out.cpp — framing parity unpinned, see FileFormats.h) */
std::string csv_f32(const cv::Mat &m)
{
    std::string s;
    char buf[32];
    for (int r = 0; r < m.rows; ++r) {
        for (int c = 0; c < m.cols; ++c) {
            std::snprintf(buf, sizeof buf, "%.4g", (double)m.at<float>(r, c));
            s += buf;
            if (c + 1 < m.cols) s += ", ";
        }
        s += "\n";
    }
    return s;
}

/* cv::imwrite of a CV_32F Mat to PNG converts with saturate_cast<uchar>: round to nearest, clamp */
bool png_from_f32(const std::string &path, const cv::Mat &m)
{
    std::vector<std::uint8_t> px((size_t)m.rows * m.cols);
    for (int r = 0; r < m.rows; ++r)
        for (int c = 0; c < m.cols; ++c) {
            const float v = m.at<float>(r, c);
            const long q = std::lrintf(v);
            px[(size_t)r * m.cols + c] = (std::uint8_t)std::min(255L, std::max(0L, q));
        }
    return bevio::writePngGray8(path, px.data(), m.rows, m.cols);
}

} // namespace

cv::Mat BatchCloudManip::saveAsMat(pcl::PointCloud<pcl::PointXYZIRCT>::Ptr cloud, std::string filename_sin_appendix,
                                   float interval)
{
    cv::Mat grid = float_bev(*cloud, interval, true);
    if (grid.empty()) return grid;
    const std::string csv = csv_f32(grid);
    if (!bevio::writeFile(filename_sin_appendix + ".csv", csv.data(), csv.size()))
        std::cerr << "Can not open file: " << filename_sin_appendix << ".csv\n";
    png_from_f32(filename_sin_appendix + ".png", grid);
    return grid;
}

cv::Mat CloudManip::saveAsMat(pcl::PointCloud<PointType>::Ptr cloud, std::string mat_filename, float interval)
{
    cv::Mat grid = float_bev(*cloud, interval, false);
    if (grid.empty()) return grid;
    const std::string csv = csv_f32(grid);
    if (!bevio::writeFile(mat_filename, csv.data(), csv.size())) std::cerr << "Can not open file: " << mat_filename << "\n";
    png_from_f32(mat_filename + ".png", grid);
    return grid;
}

void CloudManip::transformYawTranslate(const pcl::PointCloud<PointType> &in, pcl::PointCloud<PointType> &out, float tx,
                                       float ty, float tz, float yaw_deg)
{
    /* Eigen: Affine3f T = Identity; T.translation() << t; T.rotate(AngleAxisf(theta, UnitZ())), then
     * pcl::transformPointCloud(in, out, T) (CloudManip.cpp:119-128): the matrix is built on the host
     * (bev_yaw_translate_matrix), the points are transformed on the GPU (bev_transform_cloud) */
    float m[12];
    bev_yaw_translate_matrix(tx, ty, tz, yaw_deg, m);
    out = in;
    bev_ctx_t *c = bevhost_context();
    if (!c) return;
    const int rc = bev_transform_cloud(c, reinterpret_cast<const bev_point_t *>(in.points.data()), (uint32_t)in.points.size(), m,
                                       reinterpret_cast<bev_point_t *>(out.points.data()));
    if (rc != BEV_OK) std::cerr << "transformPointCloud: " << bev_strerror(rc) << "\n";
}
