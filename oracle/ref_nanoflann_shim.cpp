/*
 * ref_nanoflann_shim.cpp -- C entry points around REFERENCE CODE compiled where it lies.
 *
 * TEST INFRASTRUCTURE ONLY (like everything under oracle/).  This file holds no reference source: it includes
 *   /root/reference/place_recognition_radar/include/place_recognition_radar/nanoflann.hpp
 *   /root/reference/place_recognition_radar/include/place_recognition_radar/KDTreeVectorOfVectorsAdaptor.h
 * through the include path given by oracle/Makefile's `_ref` target and is built into oracle/_ref/libref_nanoflann.so
 * (git-ignored, travels to the GPU box with the snapshot).  These two headers are the only part of the reference's path
 * that compiles without ROS / PCL / FLANN / Eigen / Ceres / OpenCV.
 *
 * What it pins:
 *  (i)  RSCManager::VanillaKDNNSearch (RadarScancontext.cpp:225-248; the same code in Scancontext.cpp:304-336): the ring-key
 *       retrieval through `InvKeyTree = KDTreeVectorOfVectorsAdaptor<KeyMat, float>` (Scancontext.h:44: metric_L2, size_t
 *       indices, leaf 10), `nanoflann::KNNResultSet<float>` and `findNeighbors(..., SearchParams(10))`, INCLUDING the
 *       result vector's zero initialisation (cand_idx(NUM_CANDIDATES_FROM_TREE)) that shows through when the tree
 *       holds fewer points than requested.
 *  (ii) the FLANN-lineage single-kd-tree semantics SURVEY App. B.2 / B.3 call unverifiable: nanoflann is the header-only
 *       descendant of FLANN's KDTreeSingleIndex, and PCL's KdTreeFLANN<PointXY> (pointnormal.cpp:151-162, 238-254, 291)
 *       uses flann::L2_Simple<float> on that index.  Exercised here with metric_L2_Simple on float 2-D points: the
 *       radius rule (`dist < radius`, squared distances in float, accumulated x then y), the sorted order of a radius
 *       result and the 1-NN answer (ties included).
 */
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <memory>
#include <utility>
#include <vector>

#include "place_recognition_radar/KDTreeVectorOfVectorsAdaptor.h"

namespace {
typedef std::vector<std::vector<float>> KeyMat;                                  // Scancontext.h:41-44
typedef KDTreeVectorOfVectorsAdaptor<KeyMat, float> InvKeyTree;                  // metric_L2, IndexType size_t
typedef KDTreeVectorOfVectorsAdaptor<KeyMat, float, 2, nanoflann::metric_L2_Simple> Tree2f;
}  // namespace

struct ref_keytree {
  KeyMat keys;
  std::unique_ptr<InvKeyTree> tree;
};
struct ref_tree2f {
  KeyMat pts;
  std::unique_ptr<Tree2f> tree;
};

extern "C" {

/* polarcontext_tree_ = std::make_unique<InvKeyTree>(PC_NUM_RING, polarcontext_invkeys_to_search_, 10)  (:229-234) */
ref_keytree* ref_keytree_build(const float* keys, int64_t n, int32_t dim) {
  if (n <= 0 || dim <= 0) return nullptr;
  ref_keytree* t = new ref_keytree();
  t->keys.assign((size_t)n, std::vector<float>((size_t)dim));
  for (int64_t i = 0; i < n; i++) memcpy(t->keys[(size_t)i].data(), keys + i * dim, sizeof(float) * (size_t)dim);
  t->tree = std::make_unique<InvKeyTree>((size_t)dim, t->keys, 10 /* max leaf */);
  return t;
}
void ref_keytree_free(ref_keytree* t) { delete t; }

/* The search of :240-247.  out_idx / out_d2 hold num_candidates entries and start as the reference's vectors do
 * (indices 0, distances 0.f); returns KNNResultSet::size(). */
int64_t ref_keytree_knn(const ref_keytree* t, const float* query, int32_t num_candidates, uint64_t* out_idx, float* out_d2) {
  std::vector<size_t> cand_idx((size_t)num_candidates);
  std::vector<float> out_dists_sqr((size_t)num_candidates);
  nanoflann::KNNResultSet<float> knnsearch_result((size_t)num_candidates);
  knnsearch_result.init(&cand_idx[0], &out_dists_sqr[0]);
  t->tree->index->findNeighbors(knnsearch_result, query, nanoflann::SearchParams(10));
  for (int i = 0; i < num_candidates; i++) { out_idx[i] = (uint64_t)cand_idx[(size_t)i]; out_d2[i] = out_dists_sqr[(size_t)i]; }
  return (int64_t)knnsearch_result.size();
}

/* ---- float 2-D, L2_Simple: the metric PCL's KdTreeFLANN<PointXY> hands to FLANN's single kd-tree ---- */
ref_tree2f* ref_tree2f_build(const float* xy, int64_t n, int32_t leaf_max_size) {
  if (n <= 0) return nullptr;
  ref_tree2f* t = new ref_tree2f();
  t->pts.assign((size_t)n, std::vector<float>(2));
  for (int64_t i = 0; i < n; i++) { t->pts[(size_t)i][0] = xy[2 * i]; t->pts[(size_t)i][1] = xy[2 * i + 1]; }
  t->tree = std::make_unique<Tree2f>(2, t->pts, leaf_max_size);
  return t;
}
void ref_tree2f_free(ref_tree2f* t) { delete t; }

/* radiusSearch(query, radius_sq, ...) with sorted results (SearchParams::sorted defaults to true, as FLANN's does and as
 * pcl::search::KdTree::radiusSearch asks for).  Returns the number found; writes at most cap pairs. */
int64_t ref_tree2f_radius(const ref_tree2f* t, const float* query, float radius_sq, int32_t sorted, int64_t cap,
                          uint64_t* out_idx, float* out_d2) {
  std::vector<std::pair<size_t, float>> res;
  nanoflann::SearchParams sp;
  sp.sorted = sorted != 0;
  const size_t n = t->tree->index->radiusSearch(query, radius_sq, res, sp);
  for (size_t i = 0; i < n && (int64_t)i < cap; i++) { out_idx[i] = (uint64_t)res[i].first; out_d2[i] = res[i].second; }
  return (int64_t)n;
}

/* knnSearch(query, k, ...): returns the number found (<= k). */
int64_t ref_tree2f_knn(const ref_tree2f* t, const float* query, int32_t k, uint64_t* out_idx, float* out_d2) {
  std::vector<size_t> idx((size_t)k);
  std::vector<float> d2((size_t)k);
  const size_t n = t->tree->index->knnSearch(query, (size_t)k, idx.data(), d2.data());
  for (size_t i = 0; i < n; i++) { out_idx[i] = (uint64_t)idx[i]; out_d2[i] = d2[i]; }
  return (int64_t)n;
}

}  // extern "C"
