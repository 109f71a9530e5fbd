/*
 * FileFormats.h — the on-disk formats around the hot path, without PCL/OpenCV:
 *   PCD   loadPCDFile / savePCDFileBinary for PointXYZIRCT
 *         (call sites BatchMultiBevGen.cpp:730,756)
 *   PNG   8-bit grayscale writer (cv::imwrite call sites :318,:361)
 *   CSV   cv::format(mat, FMT_CSV) for CV_8U (:371)
 *   BIN   the multi-layer .bin payload (:307-314)
 * PARITY UNPINNED for the text/container framing: PCL's header text and OpenCV's
 * CSV formatter are restated from memory; the
 * payload bytes (points, occupancy, heights) are what the parity tests cover.
 * PNG files are valid PNGs of the same pixels, not byte-identical to libpng's.
 */
#ifndef BEV_HOST_FILEFORMATS_H
#define BEV_HOST_FILEFORMATS_H

#include <cstdint>
#include <string>
#include <vector>

#include "PointCloud.h"

namespace bevio {

/* returns 0 on success, -1 if the file cannot be read / parsed */
int loadPCDFile(const std::string &path, pcl::PointCloud<pcl::PointXYZIRCT> &cloud);
int savePCDFileBinary(const std::string &path, const pcl::PointCloud<pcl::PointXYZIRCT> &cloud);

bool writePngGray8(const std::string &path, const std::uint8_t *pixels, int rows, int cols);
std::string formatCsvU8(const std::uint8_t *pixels, int rows, int cols);
bool writeFile(const std::string &path, const void *data, std::size_t n);

} // namespace bevio
#endif
