/*
 * PointCloud.h — the two container types the hot path's signatures mention,
 * without PCL / OpenCV: pcl::PointXYZIRCT + pcl::PointCloud<T> (a vector of PODs
 * behind a shared_ptr) and a minimal cv::Mat (2-D array of 1-byte or float
 * elements).  Only what the reference's hot-path functions and their callers use.
 */
#ifndef BEV_HOST_POINTCLOUD_H
#define BEV_HOST_POINTCLOUD_H

#include <cstdint>
#include <cstring>
#include <memory>
#include <vector>

#include "../../include/bev_mi355x.h"

namespace pcl {

/* same 32-byte layout as the reference's struct (BatchMultiBevGen.h:43-54) and as bev_point_t */
struct alignas(16) PointXYZIRCT {
    float x, y, z, _pad0;
    float intensity;
    std::uint16_t row;
    std::uint16_t col;
    std::uint32_t t;
    std::int16_t label;
    std::uint16_t _pad1;
};
static_assert(sizeof(PointXYZIRCT) == sizeof(bev_point_t), "PointXYZIRCT must match bev_point_t");

template <class PointT>
class PointCloud {
public:
    using Ptr = std::shared_ptr<PointCloud<PointT>>;
    std::vector<PointT> points;
    std::uint32_t width = 0, height = 0;

    std::size_t size() const { return points.size(); }
    void resize(std::size_t n)
    {
        points.resize(n); /* value-initialises new elements: all-zero points */
        width = static_cast<std::uint32_t>(n);
        height = 1;
    }
    void push_back(const PointT &p)
    {
        points.push_back(p);
        width = static_cast<std::uint32_t>(points.size());
        height = 1;
    }
    void clear() { points.clear(); width = height = 0; }
};

} // namespace pcl

typedef pcl::PointXYZIRCT PointType;

namespace cv {

enum { CV_8U = 0, CV_8S = 1, CV_32F = 5, CV_8UC1 = CV_8U };

class Mat {
public:
    int rows = 0, cols = 0;
    Mat() = default;
    Mat(int r, int c, int type) { create(r, c, type); }
    static Mat zeros(int r, int c, int type) { return Mat(r, c, type); }
    void create(int r, int c, int type)
    {
        rows = r; cols = c; type_ = type;
        buf_.assign(static_cast<std::size_t>(r) * c * elemSize(), 0);
    }
    int type() const { return type_; }
    std::size_t elemSize() const { return type_ == CV_32F ? 4 : 1; }
    bool empty() const { return buf_.empty(); }
    unsigned char *ptr(int r = 0) { return buf_.data() + static_cast<std::size_t>(r) * cols * elemSize(); }
    const unsigned char *ptr(int r = 0) const { return buf_.data() + static_cast<std::size_t>(r) * cols * elemSize(); }
    template <class T> T *ptr(int r = 0) { return reinterpret_cast<T *>(ptr(r)); }
    template <class T> T &at(int r, int c) { return reinterpret_cast<T *>(ptr(r))[c]; }
    template <class T> const T &at(int r, int c) const { return reinterpret_cast<const T *>(ptr(r))[c]; }
    Mat clone() const { return *this; }
    unsigned char *data() { return buf_.data(); }
    const unsigned char *data() const { return buf_.data(); }

private:
    int type_ = CV_8U;
    std::vector<unsigned char> buf_;
};

} // namespace cv

#endif
