/*
 * ref_dump.cpp — pins this repo's oracle to the REFERENCE binary.  NOT built here, NOT part of any test run in this
 * image: it needs the reference tree with its real dependencies (PCL, OpenCV 4, Eigen, fmt) and is for a maintainer
 * who has them.  It contains none of the reference's code: it includes the reference's own translation unit where it
 * lies and calls its free functions (BatchMultiBevGen.cpp:94-373) on the committed synthetic inputs.
 *
 * Build (from the reference tree, after its own `cmake` configure has found PCL / OpenCV; ${REPO} = this repo):
 *   g++ -O3 -std=gnu++14 -I. -Iinclude $(pkg-config --cflags opencv4 pcl_common-1.12 pcl_io-1.12 eigen3) \
 *       ${REPO}/tests/golden/ref_dump.cpp src/Utility.cpp \
 *       $(pkg-config --libs opencv4 pcl_common-1.12 pcl_io-1.12) -lfmt -o ref_dump
 *   (no -march / -ffast-math: CMakeLists.txt:5-10)
 * Run:
 *   ./ref_dump ${REPO}/tests/golden/ref_pin HDL_32E tiny_hdl32 config1_16k ; ./ref_dump ... HDL_64E sweep0_hdl64
 * Inputs  ${dir}/<name>.points   raw 32-byte PointXYZIRCT records (tests/golden/make_ref_pin_inputs.py writes them)
 * Outputs ${dir}/<name>.ordered  S x 32 B   the labelled ordered cloud after markGroundPoints
 *         ${dir}/<name>.gm       S x int8   ground_mat
 *         ${dir}/<name>.bin      the multi-BEV .bin exactly as computeAndSaveMultiBev writes it (1,204,224 B)
 *         ${dir}/<name>.csv      the single-BEV .csv exactly as cv::format(FMT_CSV) writes it
 *         ${dir}/<name>.pcd      savePCDFileBinary of the labelled cloud (header text + 26-byte records)
 * tests/test_golden_cpu.py::test_reference_dumps_if_present compares every dump it finds with the oracle, byte for byte
 * (and the .csv / .pcd framing with host/FileFormats.cpp): with the dumps committed, rows A9, (c) and N4 of SURVEY.md
 * section 8 stop being "parity unpinned".
 */
#define main reference_main      /* the reference's own main (BatchMultiBevGen.cpp:664) stays in, under another name */
#include "BatchMultiBevGen.cpp" /* the reference's translation unit where it lies: its globals and free functions */
#undef main

#include <cstdio>
#include <fstream>
#include <string>
#include <vector>

static std::vector<char> slurp(const std::string &path)
{
    std::ifstream f(path, std::ios::binary);
    return std::vector<char>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}
static void spill(const std::string &path, const void *p, size_t n)
{
    std::ofstream f(path, std::ios::binary);
    f.write(static_cast<const char *>(p), (std::streamsize)n);
}

int main(int argc, char **argv)
{
    if (argc < 4) {
        std::fprintf(stderr, "usage: ref_dump <dir> <sensor_type> <name> [<name> ...]\n");
        return 2;
    }
    const std::string dir = std::string(argv[1]) + "/";
    sensor_params_ = getSensorParams(parseSensorType(argv[2]));
    setNeighbors();
    /* the four output directories of the two computeAndSave functions: everything lands in <dir>/out_<kind>/ */
    output_multi_bvm_bin_dir_ = dir + "out_multi_bin/";
    output_multi_bvm_img_dir_ = dir + "out_multi_img/";
    output_single_bvm_img_dir_ = dir + "out_single_img/";
    output_single_bvm_csv_dir_ = dir + "out_single_csv/";
    for (const std::string &d : {output_multi_bvm_bin_dir_, output_multi_bvm_img_dir_, output_single_bvm_img_dir_, output_single_bvm_csv_dir_})
        if (system(("mkdir -p " + d).c_str()) != 0) return 1;
    static_assert(sizeof(pcl::PointXYZIRCT) == 32, "PointXYZIRCT is 32 bytes (BatchMultiBevGen.h:43-54)");
    for (int a = 3; a < argc; ++a) {
        const std::string name = argv[a];
        const std::vector<char> raw = slurp(dir + name + ".points");
        pcl::PointCloud<pcl::PointXYZIRCT>::Ptr in(new pcl::PointCloud<pcl::PointXYZIRCT>());
        in->points.resize(raw.size() / 32);
        std::memcpy(in->points.data(), raw.data(), in->points.size() * 32);
        in->width = (uint32_t)in->points.size();
        in->height = 1;
        pcl::PointCloud<pcl::PointXYZIRCT>::Ptr ordered(new pcl::PointCloud<pcl::PointXYZIRCT>());
        getOrderedCloud(in, ordered);                 /* :94-117 */
        cv::Mat ground_mat;
        markGroundPoints(ordered, ground_mat);        /* :119-252 */
        spill(dir + name + ".ordered", ordered->points.data(), ordered->points.size() * 32);
        if (ground_mat.isContinuous()) spill(dir + name + ".gm", ground_mat.ptr(0), ground_mat.total());
        computeAndSaveMultiBev(ordered, name);        /* :261-321 -> out_multi_bin/<name>.bin */
        computeAndSaveSingleBev(ordered, name);       /* :331-373 -> out_single_csv/<name>.csv */
        const std::vector<char> bin = slurp(output_multi_bvm_bin_dir_ + name + ".bin"), csv = slurp(output_single_bvm_csv_dir_ + name + ".csv");
        spill(dir + name + ".bin", bin.data(), bin.size());
        spill(dir + name + ".csv", csv.data(), csv.size());
        ordered->width = (uint32_t)ordered->points.size();
        ordered->height = 1;
        pcl::io::savePCDFileBinary(dir + name + ".pcd", *ordered); /* :756 */
        std::printf("%s: %zu points in, %zu slots, .bin %zu B, .csv %zu B\n", name.c_str(), in->points.size(), ordered->points.size(),
                    bin.size(), csv.size());
    }
    return 0;
}
