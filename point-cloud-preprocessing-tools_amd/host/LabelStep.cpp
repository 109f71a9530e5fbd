#include "LabelStep.h"

#include <algorithm>
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
/*
 * The reference searches with nanoflann 1.3.2 (include/nanoflann.hpp, NANOFLANN_VERSION 0x132) through
 * KDTreeVectorOfVectorsAdaptor<std::vector<std::vector<float>>, float>: metric L2, leaf size 10, eps 0.
 * Distances are the same whatever the search order, but WHICH of several exactly equidistant points is returned
 * first depends on the tree: on the order points end up in inside the leaves (the in-place partitions of the
 * build) and on the order leaves are visited.  This is a restatement of that published algorithm for 3-D float
 * points — same split rule, same partition, same traversal and result-set insertion — so that labels agree with
 * the reference even on ties; tests/test_label_step_cpu.py checks it against the reference's own header.
 */
class KdTree3f {
public:
    explicit KdTree3f(const std::vector<std::vector<float>> &pts) : pts_(pts), order_(pts.size())
    {
        for (size_t i = 0; i < order_.size(); ++i) order_[i] = i;
        if (pts_.empty()) return;
        for (int d = 0; d < 3; ++d) root_box_.lo[d] = root_box_.hi[d] = pts_[0][d];
        for (const auto &p : pts_)
            for (int d = 0; d < 3; ++d) {
                if (p[d] < root_box_.lo[d]) root_box_.lo[d] = p[d];
                if (p[d] > root_box_.hi[d]) root_box_.hi[d] = p[d];
            }
        Box box = root_box_;
        root_ = build(0, order_.size(), box);
    }

    /* k nearest to q.  idx / d2 keep their value-initialised contents (0) in slots that are never filled, like the
     * vectors at BatchMultiBevGen.cpp:541-542 / :604-605 — except the last distance, which the result set presets */
    void search(const std::vector<float> &q, size_t k, std::vector<size_t> &idx, std::vector<float> &d2) const
    {
        idx.assign(k, 0);
        d2.assign(k, 0.0f);
        if (k == 0 || pts_.empty()) return;
        Results r{idx, d2, k, 0};
        d2[k - 1] = std::numeric_limits<float>::max();
        float side[3] = {0.0f, 0.0f, 0.0f}; /* squared distance from q to the current cell, per axis */
        float min_d2 = 0.0f;
        for (int d = 0; d < 3; ++d) {
            if (q[d] < root_box_.lo[d]) { side[d] = sq(q[d] - root_box_.lo[d]); min_d2 += side[d]; }
            if (q[d] > root_box_.hi[d]) { side[d] = sq(q[d] - root_box_.hi[d]); min_d2 += side[d]; }
        }
        descend(root_, q, min_d2, side, r);
    }

private:
    static constexpr size_t kLeafSize = 10;
    struct Box { float lo[3], hi[3]; };
    struct Node {
        int child[2] = {-1, -1};  /* inner node */
        int axis = 0;
        float div_lo = 0, div_hi = 0;
        size_t first = 0, last = 0; /* leaf: order_[first, last) */
    };
    struct Results {
        std::vector<size_t> &idx;
        std::vector<float> &d2;
        size_t cap, count;
        float worst() const { return d2[cap - 1]; }
        void add(float d, size_t i) /* insertion behind equal distances: the first one found stays first */
        {
            size_t pos = count;
            for (; pos > 0 && d2[pos - 1] > d; --pos)
                if (pos < cap) { d2[pos] = d2[pos - 1]; idx[pos] = idx[pos - 1]; }
            if (pos < cap) { d2[pos] = d; idx[pos] = i; }
            if (count < cap) ++count;
        }
    };
    static float sq(float v) { return v * v; }

    void min_max(size_t first, size_t count, int axis, float &mn, float &mx) const
    {
        mn = mx = pts_[order_[first]][axis];
        for (size_t i = 1; i < count; ++i) {
            const float v = pts_[order_[first + i]][axis];
            if (v < mn) mn = v;
            if (v > mx) mx = v;
        }
    }

    /* two sweeps of the classic two-pointer partition over order_[first, first + count): afterwards
     * [0, below) < cut, [below, not_above) == cut, the rest > cut */
    void partition(size_t first, size_t count, int axis, float cut, size_t &below, size_t &not_above)
    {
        size_t *ind = &order_[first];
        auto sweep = [&](size_t from, bool strict) {
            size_t l = from, r = count - 1;
            for (;;) {
                while (l <= r && (strict ? pts_[ind[l]][axis] < cut : pts_[ind[l]][axis] <= cut)) ++l;
                while (r && l <= r && (strict ? pts_[ind[r]][axis] >= cut : pts_[ind[r]][axis] > cut)) --r;
                if (l > r || !r) break;
                std::swap(ind[l], ind[r]);
                ++l;
                --r;
            }
            return l;
        };
        below = sweep(0, true);
        not_above = sweep(below, false);
    }

    int build(size_t first, size_t last, Box &box)
    {
        const int id = (int)nodes_.size();
        nodes_.emplace_back();
        const size_t count = last - first;
        if (count <= kLeafSize) {
            nodes_[id].first = first;
            nodes_[id].last = last;
            for (int d = 0; d < 3; ++d) min_max(first, count, d, box.lo[d], box.hi[d]); /* tight box of the leaf */
            return id;
        }
        /* split axis: the largest spread of the points among the axes whose box span is within 1e-5 of the widest */
        const float eps = 0.00001f;
        float widest = box.hi[0] - box.lo[0];
        for (int d = 1; d < 3; ++d) widest = std::max(widest, box.hi[d] - box.lo[d]);
        int axis = 0;
        float best_spread = -1.0f;
        for (int d = 0; d < 3; ++d)
            if (box.hi[d] - box.lo[d] > (1 - eps) * widest) {
                float mn, mx;
                min_max(first, count, d, mn, mx);
                if (mx - mn > best_spread) { axis = d; best_spread = mx - mn; }
            }
        /* cut in the middle of the box, pulled into the range of the points */
        float mn, mx;
        min_max(first, count, axis, mn, mx);
        const float mid = (box.lo[axis] + box.hi[axis]) / 2;
        const float cut = mid < mn ? mn : (mid > mx ? mx : mid);
        size_t below, not_above;
        partition(first, count, axis, cut, below, not_above);
        const size_t half = count / 2;
        const size_t left_n = below > half ? below : (not_above < half ? not_above : half);

        Box lb = box, rb = box;
        lb.hi[axis] = cut;
        rb.lo[axis] = cut;
        const int c0 = build(first, first + left_n, lb);
        const int c1 = build(first + left_n, last, rb);
        Node &n = nodes_[id];
        n.child[0] = c0;
        n.child[1] = c1;
        n.axis = axis;
        n.div_lo = lb.hi[axis];
        n.div_hi = rb.lo[axis];
        for (int d = 0; d < 3; ++d) {
            box.lo[d] = std::min(lb.lo[d], rb.lo[d]);
            box.hi[d] = std::max(lb.hi[d], rb.hi[d]);
        }
        return id;
    }

    void descend(int id, const std::vector<float> &q, float min_d2, float side[3], Results &r) const
    {
        const Node &n = nodes_[id];
        if (n.child[0] < 0) {
            const float worst = r.worst(); /* read once per leaf */
            for (size_t i = n.first; i < n.last; ++i) {
                const float d = dist_sqr(q, pts_[order_[i]]);
                if (d < worst) r.add(d, order_[i]);
            }
            return;
        }
        const float v = q[n.axis];
        const bool low_first = (v - n.div_lo) + (v - n.div_hi) < 0;
        const float cut_d2 = sq(v - (low_first ? n.div_hi : n.div_lo));
        descend(n.child[low_first ? 0 : 1], q, min_d2, side, r);
        const float saved = side[n.axis];
        min_d2 = min_d2 + cut_d2 - saved;
        side[n.axis] = cut_d2;
        if (min_d2 * 1.0f <= r.worst()) descend(n.child[low_first ? 1 : 0], q, min_d2, side, r);
        side[n.axis] = saved;
    }

    const std::vector<std::vector<float>> &pts_;
    std::vector<size_t> order_;
    std::vector<Node> nodes_;
    Box root_box_{};
    int root_ = -1;
};

void knn(const std::vector<std::vector<float>> &set, const std::vector<float> &q, size_t k, std::vector<size_t> &idx,
         std::vector<float> &d2)
{
    KdTree3f(set).search(q, k, idx, d2);
}
} // namespace

void nearestPositions(const std::vector<std::vector<float>> &positions, const std::vector<float> &query, size_t k,
                      std::vector<size_t> &indices, std::vector<float> &dists_sqr)
{
    knn(positions, query, k, indices, dists_sqr);
}

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
