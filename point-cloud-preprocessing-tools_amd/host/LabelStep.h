/*
 * LabelStep.h — Step 2 of batch_multi_bev_gen's main() (reference
 * BatchMultiBevGen.cpp:762-765): keyframe poses -> major frames -> soft one-hot
 * labels -> keyframe_label.csv.  SURVEY.md §8(f) row N1 ("next").  Tiny CPU work
 * that runs once per dataset; no GPU involved.
 *
 * Without Eigen / nanoflann: 3x3 matrices are plain arrays and the k-NN searches
 * restate nanoflann 1.3.2's KD-tree for 3-D float points (split rule, partition,
 * traversal, result-set insertion; distances ((dx*dx) + dy*dy) + dz*dz in float),
 * so that even the choice between exactly equidistant major frames is the
 * reference's.  Checked against the reference's own nanoflann.hpp in
 * tests/test_label_step_cpu.py.
 */
#ifndef BEV_HOST_LABELSTEP_H
#define BEV_HOST_LABELSTEP_H

#include <cstdint>
#include <string>
#include <vector>

struct Pose6f { /* include/Utility.h:38-77 (quaternion omitted: unused by this tool) */
    float x, y, z, roll, pitch, yaw;
    double rotation_matrix[3][3];
    std::vector<float> getPositionVec() const { return std::vector<float>{x, y, z}; }
};

using LabelType = std::vector<float>; /* BatchMultiBevGen.cpp:23 */

bool isRotationMatirx(const double R[3][3]);                          /* src/Utility.cpp:11-19 */
void rotationMatrixToEulerAngles(const double R[3][3], double out[3]); /* src/Utility.cpp:21-41 */
float getDistance(const Pose6f &a, const Pose6f &b);                  /* src/Utility.cpp:43-49 */

/* BatchMultiBevGen.cpp:381-460.  ok=false when the file cannot be opened (the reference exits). */
std::vector<Pose6f> readKeyframePose(std::string pose_filename, bool *ok = nullptr);
std::vector<int32_t> selectMajorFrames(std::vector<Pose6f> &keyframe_pose);                        /* :502-566 */
std::vector<LabelType> getKeyFrameLabel(std::vector<Pose6f> &key_frame_poses,
                                        std::vector<int32_t> &major_frame_indeices);             /* :575-636 */
bool saveLabels(std::vector<LabelType> key_frame_labels, std::string label_filename);              /* :645-661 */

/* the k-NN both functions above use (KD-tree over 3-D positions, see LabelStep.cpp); exposed for the tests */
void nearestPositions(const std::vector<std::vector<float>> &positions, const std::vector<float> &query, size_t k,
                      std::vector<size_t> &indices, std::vector<float> &dists_sqr);

#endif
