/*
 * ref_knn.cpp — the reference's OWN k-NN, for pinning row N1 (label step).
 *
 * TEST INFRASTRUCTURE ONLY (see bev_oracle.h).  This driver contains no reference code: it includes
 * include/KDTreeVectorOfVectorsAdaptor.h and include/nanoflann.hpp WHERE THEY LIE under /root/reference (both are
 * dependency-free: standard headers only) and calls them exactly as BatchMultiBevGen.cpp does at :22 (tree type),
 * :534-550 and :594-613 (3-D tree over std::vector<std::vector<float>>, leaf size 10, KNNResultSet<float>,
 * SearchParams(10), result vectors value-initialised before the search).  Built by oracle/Makefile into
 * oracle/_ref/libref_knn.so only where /root/reference exists; the prebuilt file travels to the GPU box.
 * The rest of BatchMultiBevGen.cpp needs PCL / OpenCV / Eigen / fmt and is unbuildable here.
 */
#include <KDTreeVectorOfVectorsAdaptor.h>

#include <cstddef>
#include <cstdint>
#include <memory>
#include <vector>

using PosVecMat = std::vector<std::vector<float>>;                 /* BatchMultiBevGen.cpp:21 */
using InvKeyTree = KDTreeVectorOfVectorsAdaptor<PosVecMat, float>; /* :22 */

extern "C" {
/* k nearest of the n 3-D points `pts` to `query`; idx_out / d2_out: k entries, value-initialised like :541-542.
 * Returns the number of results found (min(k, n)). */
int ref_knn(const float *pts, size_t n, const float *query, size_t k, uint64_t *idx_out, float *d2_out)
{
    PosVecMat mat(n, std::vector<float>(3));
    for (size_t i = 0; i < n; ++i)
        for (int d = 0; d < 3; ++d) mat[i][d] = pts[3 * i + d];
    std::unique_ptr<InvKeyTree> tree = std::make_unique<InvKeyTree>(3 /* dim */, mat, 10 /* max leaf */);
    std::vector<size_t> candidate_indexes(k);
    std::vector<float> out_dists_sqr(k);
    nanoflann::KNNResultSet<float> result(k);
    result.init(&candidate_indexes[0], &out_dists_sqr[0]);
    tree->index->findNeighbors(result, query, nanoflann::SearchParams(10));
    for (size_t j = 0; j < k; ++j) {
        idx_out[j] = candidate_indexes[j];
        d2_out[j] = out_dists_sqr[j];
    }
    return (int)result.size();
}
}
