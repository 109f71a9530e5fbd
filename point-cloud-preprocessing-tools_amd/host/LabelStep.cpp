#include "LabelStep.h"

#include <cmath>
#include <fstream>
#include <iostream>
#include <iterator>
#include <limits>
#include <sstream>

bool isRotationMatirx(const double R[3][3])
{
    /* || R R^T - I ||_F < 1e-4 */
    double acc = 0.0;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            double v = 0.0;
            for (int k = 0; k < 3; ++k) v += R[i][k] * R[j][k];
            v -= (i == j) ? 1.0 : 0.0;
            acc += v * v;
        }
    return std::sqrt(acc) < 1e-4;
}

void rotationMatrixToEulerAngles(const double R[3][3], double out[3])
{
    if (!isRotationMatirx(R)) std::cerr << "Not A Rotation Matrix. " << std::endl;
    const double sy = std::sqrt(R[0][0] * R[0][0] + R[1][0] * R[1][0]);
    if (!(sy < 1e-6)) {
        out[0] = std::atan2(R[2][1], R[2][2]);
        out[1] = std::atan2(-R[2][0], sy);
        out[2] = std::atan2(R[1][0], R[0][0]);
    } else {
        out[0] = std::atan2(-R[1][2], R[1][1]);
        out[1] = std::atan2(-R[2][0], sy);
        out[2] = 0;
    }
}

float getDistance(const Pose6f &a, const Pose6f &b)
{
    const float dx = a.x - b.x, dy = a.y - b.y, dz = a.z - b.z;
    return std::sqrt(dx * dx + dy * dy + dz * dz);
}

std::vector<Pose6f> readKeyframePose(std::string pose_filename, bool *ok)
{
    std::vector<Pose6f> poses;
    std::ifstream f(pose_filename);
    if (ok) *ok = f.is_open();
    if (!f.is_open()) {
        std::cerr << "failed to load keyframe pose file: " << pose_filename << std::endl;
        return poses;
    }
    std::cout << "loaded keyframe pose file: " << pose_filename << std::endl;
    /* one whitespace-delimited token per keyframe: idx,x,y,z,roll,pitch,yaw,r00..r22 (:395-401) */
    std::string entry;
    while (f >> entry) {
        std::vector<std::string> tok;
        std::stringstream ss(entry);
        std::string t;
        while (std::getline(ss, t, ',')) tok.push_back(t);
        if (tok.size() != 16) {
            std::cerr << "Size of entry_token is: " << tok.size() << ", while expecting 16. " << std::endl;
            break;
        }
        Pose6f p{};
        const double tx = std::stod(tok[1]), ty = std::stod(tok[2]), tz = std::stod(tok[3]);
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) p.rotation_matrix[i][j] = std::stod(tok[7 + 3 * i + j]);
        double e[3];
        rotationMatrixToEulerAngles(p.rotation_matrix, e);
        p.x = float(tx); p.y = float(ty); p.z = float(tz);
        p.roll = float(e[0]); p.pitch = float(e[1]); p.yaw = float(e[2]);
        poses.push_back(p);
    }
    std::cout << "Finish reading all keyframe pose, total " << poses.size() << " entries. " << std::endl;
    return poses;
}

namespace {
/* nanoflann L2 for dim 3: sequential float accumulation of squared differences */
float dist_sqr(const std::vector<float> &a, const std::vector<float> &b)
{
    float r = 0.0f;
    for (int d = 0; d < 3; ++d) {
        const float diff = a[d] - b[d];
        r += diff * diff;
    }
    return r;
}
/* k nearest of `set` to q; unfilled result slots keep index 0 / distance 0 like the
 * value-initialised vectors at :541-542 / :604-605 */
void knn(const std::vector<std::vector<float>> &set, const std::vector<float> &q, size_t k, std::vector<size_t> &idx,
         std::vector<float> &d2)
{
    idx.assign(k, 0);
    d2.assign(k, 0.0f);
    std::vector<float> best(k, std::numeric_limits<float>::max());
    size_t found = 0;
    for (size_t i = 0; i < set.size(); ++i) {
        const float d = dist_sqr(q, set[i]);
        size_t pos = found < k ? found : k;
        while (pos > 0 && best[pos - 1] > d) --pos; /* stable: earlier index wins ties */
        if (pos >= k) continue;
        for (size_t j = std::min(found, k - 1); j > pos; --j) { best[j] = best[j - 1]; idx[j] = idx[j - 1]; }
        best[pos] = d;
        idx[pos] = i;
        if (found < k) ++found;
    }
    for (size_t j = 0; j < found; ++j) d2[j] = best[j];
}
} // namespace

std::vector<int32_t> selectMajorFrames(std::vector<Pose6f> &keyframe_pose)
{
    const float MAJOR_FRAME_INTERVAL = 20.0f;
    std::vector<int32_t> major;
    std::vector<std::vector<float>> major_pos;
    if (keyframe_pose.empty()) return major;
    major.push_back(0);
    major_pos.push_back(keyframe_pose[0].getPositionVec());
    for (int i = 1; i < (int)keyframe_pose.size(); ++i) {
        const Pose6f &last = keyframe_pose[major.back()];
        if (getDistance(keyframe_pose[i], last) < MAJOR_FRAME_INTERVAL) continue; /* :527-531 */
        std::vector<size_t> idx;
        std::vector<float> d2;
        const auto q = keyframe_pose[i].getPositionVec();
        knn(major_pos, q, 1, idx, d2);                                              /* :534-550 */
        if (d2[0] < MAJOR_FRAME_INTERVAL * MAJOR_FRAME_INTERVAL) {
            std::cout << "Key Frame " << i << " overlaps with previous Major Frame " << idx[0] << ", i.e. Key Frame "
                      << major[idx[0]] << ". \n";
            continue;
        }
        major.push_back(i);
        major_pos.push_back(q);
    }
    return major;
}

std::vector<LabelType> getKeyFrameLabel(std::vector<Pose6f> &key_frame_poses, std::vector<int32_t> &major)
{
    std::vector<LabelType> labels(key_frame_poses.size(), LabelType(major.size(), 0));
    std::cout << "One-hot label has length: " << major.size() << std::endl;
    std::vector<std::vector<float>> major_pos;
    for (int32_t m : major) major_pos.push_back(key_frame_poses[m].getPositionVec());
    for (int i = 0; i < (int)key_frame_poses.size(); ++i) {
        std::vector<size_t> idx;
        std::vector<float> d2;
        knn(major_pos, key_frame_poses[i].getPositionVec(), 2, idx, d2);
        if (i == major[idx[0]]) {
            labels[i][idx[0]] = 1.0f;                                  /* :616-618 */
        } else {
            float w0 = (float)(1.0f / ((double)d2[0] + 1e-5));         /* :623-627 */
            float w1 = (float)(1.0f / ((double)d2[1] + 1e-5));
            const float sum = w0 + w1;
            w0 /= sum;
            w1 /= sum;
            labels[i][idx[0]] = w0;
            labels[i][idx[1]] = w1;
        }
    }
    return labels;
}

bool saveLabels(std::vector<LabelType> key_frame_labels, std::string label_filename)
{
    std::ofstream f(label_filename);
    if (!f.is_open()) {
        std::cerr << "failed to open keyframe label file: " << label_filename << std::endl;
        return false;
    }
    for (LabelType &label : key_frame_labels) { /* default float formatting, "," after every value (:653-657) */
        std::ostream_iterator<float> it(f, ",");
        std::copy(label.begin(), label.end(), it);
        f << "\n";
    }
    std::cout << "saved labels from " << key_frame_labels.size() << " key frames. " << std::endl;
    return true;
}
