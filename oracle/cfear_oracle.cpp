/*
 * cfear_oracle.cpp -- CPU restatement of the CFEAR scan-registration hot path of
 * dan11003/tbv_slam_public (polar filter -> motion compensation -> oriented surface points ->
 * many-to-one P2L/P2P/P2D scan matcher with a Ceres-2.1-equivalent Levenberg-Marquardt loop).
 *
 * TEST INFRASTRUCTURE ONLY -- see cfear_oracle.h.  PARITY UNPINNED (no reference test vectors, the
 * reference cannot be built here).  Every function cites the reference file:line it follows; paths
 * are relative to /root/reference/cfear_radarodometry/{src,include}/cfear_radarodometry/ unless
 * stated.  Third-party semantics restated from upstream knowledge:
 *   PCL 1.10 pcl::VoxelGrid / CentroidPoint, FLANN 1.9.1 L2_Simple + RadiusResultSet,
 *   Eigen 3.3 SelfAdjointEigenSolver<Matrix2d>, Ceres 2.1.0 TrustRegionMinimizer +
 *   LevenbergMarquardtStrategy + loss functions + Corrector.
 *
 * Build: g++ -O3 -std=c++17 -ffp-contract=off -fPIC -shared  (no FMA contraction: the reference
 * is an x86-64 -O3 build without -mfma, so its float expressions are never fused).
 *
 * Canonical choices where the reference is order-unspecified (documented in DESIGN.md):
 *   - in-voxel accumulation order of pcl::VoxelGrid (std::sort is unstable): input order;
 *   - radius-search result order for equal distances: input index;
 *   - 1-NN ties (equidistant targets): lowest target index;
 *   - Eigen reductions (sum(), products): sequential in the order above.
 */
#include "cfear_oracle.h"

#include <algorithm>
#include <cfloat>
#include <chrono>
#include <cmath>
#include <cstring>
#include <limits>
#include <unordered_map>
#include <utility>
#include <vector>

namespace {

// ------------------------------------------------------------------------------------------
// small SE(2) helper standing in for Eigen::Affine2d / planar Eigen::Affine3d
// ------------------------------------------------------------------------------------------
struct Aff2 {
  double l[4];  // linear, row-major
  double t[2];
};

Aff2 aff_identity() { return Aff2{{1, 0, 0, 1}, {0, 0}}; }

// registration.cpp:128-135 vectorToAffine3d + n_scan_normal.cpp:350-351 (Translation2d * linear)
Aff2 aff_from_xyt(double x, double y, double th) {
  const double c = std::cos(th), s = std::sin(th);
  return Aff2{{c, -s, s, c}, {x, y}};
}

// Eigen Transform<Affine> * Transform<Affine>
Aff2 aff_mul(const Aff2& a, const Aff2& b) {
  Aff2 r;
  r.l[0] = a.l[0] * b.l[0] + a.l[1] * b.l[2];
  r.l[1] = a.l[0] * b.l[1] + a.l[1] * b.l[3];
  r.l[2] = a.l[2] * b.l[0] + a.l[3] * b.l[2];
  r.l[3] = a.l[2] * b.l[1] + a.l[3] * b.l[3];
  r.t[0] = a.l[0] * b.t[0] + a.l[1] * b.t[1] + a.t[0];
  r.t[1] = a.l[2] * b.t[0] + a.l[3] * b.t[1] + a.t[1];
  return r;
}

// Eigen Transform<Affine>::inverse(): general inverse of the linear part (adjugate / det)
Aff2 aff_inv(const Aff2& a) {
  const double det = a.l[0] * a.l[3] - a.l[2] * a.l[1];
  const double invdet = 1.0 / det;
  Aff2 r;
  r.l[0] = a.l[3] * invdet;
  r.l[1] = -a.l[1] * invdet;
  r.l[2] = -a.l[2] * invdet;
  r.l[3] = a.l[0] * invdet;
  r.t[0] = -(r.l[0] * a.t[0] + r.l[1] * a.t[1]);
  r.t[1] = -(r.l[2] * a.t[0] + r.l[3] * a.t[1]);
  return r;
}

// utils.cpp:115-122 Affine3dToVectorXYeZ: (t_x, t_y, eulerAngles(0,1,2)[2]); for a planar rotation
// Eigen's eulerAngles(0,1,2) yields (0, 0, atan2(R10, R11)).
void aff_to_xyt(const Aff2& a, double p[3]) {
  p[0] = a.t[0];
  p[1] = a.t[1];
  p[2] = std::atan2(a.l[2], a.l[3]);
}

// ------------------------------------------------------------------------------------------
// symmetric 2x2 eigen decomposition standing in for Eigen::SelfAdjointEigenSolver<Matrix2d>
// (pointnormal.cpp:39-45).  One Jacobi rotation; eigenvalues ascending, unit eigenvectors.  Only
// the lower triangle (a, b = m(1,0), d) is read, like Eigen does.
// ------------------------------------------------------------------------------------------
void sym2_eig(double a, double b, double d, double* l0, double* l1, double v0[2], double v1[2]) {
  double c = 1.0, s = 0.0, e0 = a, e1 = d;
  if (b != 0.0) {
    const double theta = (d - a) / (2.0 * b);
    const double t = (theta >= 0.0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
    c = 1.0 / std::sqrt(t * t + 1.0);
    s = t * c;
    e0 = a - t * b;
    e1 = d + t * b;
  }
  // eigenvector of e0: (c, -s); of e1: (s, c)
  if (e0 <= e1) {
    *l0 = e0; *l1 = e1;
    v0[0] = c; v0[1] = -s; v1[0] = s; v1[1] = c;
  } else {
    *l0 = e1; *l1 = e0;
    v0[0] = s; v0[1] = c; v1[0] = c; v1[1] = -s;
  }
}

}  // namespace

// ==========================================================================================
// F: k-strongest  (radar_filters.cpp:198-237 StructuredKStrongest::FilterKstrongest)
// ==========================================================================================
extern "C" int orc_kstrongest(const uint8_t* img, int rows, int cols, int stride, int k, int z_min,
                              int32_t* sel_range, uint8_t* sel_intensity, int32_t* sel_count) {
  if (k < 1) return -1;
  typedef std::pair<uint8_t, int> intensity_range;   // radar_filters.h:84
  const uint8_t u_zmin = (uint8_t)z_min;              // :212  uchar(z_min_)
  std::vector<intensity_range> v;
  for (int bearing = 0; bearing < rows; bearing++) {
    v.clear();
    for (int range = 0; range < cols; range++) {
      const uint8_t intensity = img[(size_t)bearing * stride + range];
      if (intensity < u_zmin) continue;                                    // :217
      if (v.empty()) {
        v.push_back(std::make_pair(intensity, range));                      // :220-222
      } else {
        const intensity_range p = std::make_pair(intensity, range);
        auto it = std::lower_bound(v.cbegin(), v.cend(), p);                // :225
        v.insert(it, p);
        if (v.size() > (size_t)k) v.erase(v.begin());                       // :227-228
      }
    }
    sel_count[bearing] = (int32_t)v.size();
    for (int j = 0; j < k; j++) {
      sel_range[(size_t)bearing * k + j] = j < (int)v.size() ? v[j].second : -1;
      sel_intensity[(size_t)bearing * k + j] = j < (int)v.size() ? v[j].first : 0;
    }
  }
  return 0;
}

// ==========================================================================================
// F: peaks  (radar_filters.cpp:238-298 StructuredKStrongest::AxialNonMaxSupress)
// cv::Mat::at is unchecked in release builds: a column index outside [0,cols) of a continuous
// image addresses the neighbouring row; outside the buffer the oracle reads 0.
// ==========================================================================================
extern "C" int orc_peaks(const uint8_t* img, int rows, int cols, int stride, int k,
                         const int32_t* sel_range, const int32_t* sel_count, uint8_t* is_peak) {
  const int window_size = 3;                                                // :240
  const long total = (long)rows * stride;
  for (int bearing = 0; bearing < rows; bearing++) {
    std::unordered_map<int, uint16_t> score;                                // :245
    const int32_t* ranges = sel_range + (size_t)bearing * k;
    const int cnt = sel_count[bearing];
    for (int j = 0; j < cnt; j++) {
      const int masked_range = ranges[j];
      if (masked_range < window_size || masked_range >= cols - window_size) continue;   // :251
      for (int r_n = masked_range - window_size; r_n <= masked_range + window_size; r_n++) {
        if (score.find(r_n) == score.end()) {
          uint16_t s = 0;
          for (int r_nn = r_n - window_size; r_nn <= r_n + window_size; r_nn++) {       // :258
            const long lin = (long)bearing * stride + r_nn;
            const uint8_t raw = (lin >= 0 && lin < total) ? img[lin] : 0;
            s = (uint16_t)(s + (uint16_t)raw);
          }
          score[r_n] = s;
        }
      }
    }
    for (int j = 0; j < k; j++) is_peak[(size_t)bearing * k + j] = 0;
    for (int j = 0; j < cnt; j++) {                                         // :267-294
      const int masked_range = ranges[j];
      bool largest = true;
      const uint16_t pthis = score[masked_range];
      for (uint16_t i = 1; i <= window_size; i++) {
        const uint16_t pnext = score[masked_range + i];
        const uint16_t pprev = score[masked_range - i];
        if (pprev > pthis || pthis < pnext) { largest = false; break; }     // :282
      }
      is_peak[(size_t)bearing * k + j] = largest ? 1 : 0;
    }
  }
  return 0;
}

// ==========================================================================================
// F: polar -> Cartesian  (radar_filters.cpp:309-337 getPeaksFilteredPointCloud)
// Parameters arrive as float and are widened to double (radar_driver.h:40-45, radar_driver.cpp:58).
// ==========================================================================================
extern "C" int orc_kstrongest_cloud(int rows, int k, const int32_t* sel_range,
                                    const uint8_t* sel_intensity, const int32_t* sel_count,
                                    const uint8_t* mask, float range_res_f, float min_distance_f,
                                    float* xyzi) {
  const double range_res_ = (double)range_res_f, min_distance_ = (double)min_distance_f;
  const int min_range_bin = (int)std::ceil(min_distance_ / range_res_);     // :315
  int n = 0;
  for (int bearing = 0; bearing < rows; bearing++) {
    const double theta = (double(bearing + 1) / rows) * 2. * M_PI;          // :317
    if (sel_count[bearing] == 0) continue;
    const double cos_t = std::cos(theta), sin_t = std::sin(theta);
    const double range_res_half = range_res_ / 2.0;
    for (int j = 0; j < sel_count[bearing]; j++) {
      if (mask && !mask[(size_t)bearing * k + j]) continue;
      const int range = sel_range[(size_t)bearing * k + j];
      if (range > min_range_bin) {                                          // :327
        xyzi[4 * n + 0] = (float)((range_res_half + range_res_ * range) * cos_t);
        xyzi[4 * n + 1] = (float)((range_res_half + range_res_ * range) * sin_t);
        xyzi[4 * n + 2] = 0.f;
        xyzi[4 * n + 3] = (float)sel_intensity[(size_t)bearing * k + j];
        n++;
      }
    }
  }
  return n;
}

// ==========================================================================================
// F: legacy k_strongest_filter / InsertStrongestK (radar_filters.cpp:25-78), used by CorAl's standalone kstrongRadar
// (coral_alignment_quality/src/alignment_checker/ScanType.cpp:104-114), not by TBV.  Literal restatement: the sorted
// vector, the "<= current minimum -> reject, even when not full" rule, sort by intensity only.  std::sort on <= 16
// elements is libstdc++'s insertion sort, i.e. stable -- equal intensities stay in insertion (= ascending range) order;
// for k >= 16 libstdc++ switches to introsort and the reference's tie order becomes implementation-defined (the kept
// intensities are the same); this restatement stays stable for every k.  theta is a FLOAT and cos / sin are the float
// overloads (ros/duration.h pulls <math.h> into the reference's translation unit): x = float(range_res * i * cosf(theta)).
// ==========================================================================================
extern "C" int orc_kstrongest_legacy(const uint8_t* img, int rows, int cols, int stride, int k_strongest, double z_min,
                                     double range_res, double min_distance, float* xyzi, int cap) {
  struct P { float x, y, intensity; };
  const double min_distance_sqrd = min_distance * min_distance;
  int n = 0;
  for (int bearing = 0; bearing < rows; bearing++) {
    const float theta = ((float)(bearing + 1) / rows) * 2 * M_PI;            // :52 (float)
    std::vector<P> pnts_sorted;
    for (int i = 0; i < cols; i++) {
      const uint8_t v = img[(size_t)bearing * stride + i];
      if (v < z_min) continue;                                               // :58 uchar < double
      P p;
      p.x = (float)(range_res * i * std::cos(theta));                        // :62-63, std::cos(float)
      p.y = (float)(range_res * i * std::sin(theta));
      p.intensity = (float)v;
      // InsertStrongestK :25-38
      if (pnts_sorted.empty()) { pnts_sorted.push_back(p); continue; }
      if (p.intensity <= pnts_sorted.back().intensity) continue;
      pnts_sorted.push_back(p);
      std::stable_sort(pnts_sorted.begin(), pnts_sorted.end(), [](const P& a, const P& b) { return a.intensity > b.intensity; });
      if ((int)pnts_sorted.size() > k_strongest) pnts_sorted.pop_back();
    }
    for (const P& p : pnts_sorted) {
      if ((double)(p.x * p.x + p.y * p.y) > min_distance_sqrd) {             // :71 float arithmetic, compared as double
        if (n >= cap) return -1;
        xyzi[4 * n] = p.x; xyzi[4 * n + 1] = p.y; xyzi[4 * n + 2] = 0.f; xyzi[4 * n + 3] = p.intensity;
        n++;
      }
    }
  }
  return n;
}

// ==========================================================================================
// F: CA-CFAR  (cfar.cpp:12-83; constructed at radar_driver.cpp:54 with max_distance 400.0)
// ==========================================================================================
namespace {
double cfar_get_mean(const uint8_t* azimuth, int start_idx, int end_idx) {  // cfar.cpp:73-83
  double sum = 0., N = 0.;
  for (size_t i = (size_t)start_idx; i < (size_t)end_idx; i++) {
    sum += std::pow(double(azimuth[i]), 2.);
    N += 1.;
  }
  return sum / N;
}
}  // namespace

extern "C" int orc_cacfar(const uint8_t* img, int rows, int cols, int stride, int window_size_,
                          int nb_guard_cells_, float false_alarm_rate_f, float range_res_f,
                          float z_min_f, float min_distance_f, double max_distance_, float* xyzi,
                          int32_t* det_rc, int cap) {
  const double false_alarm_rate_ = (double)false_alarm_rate_f;
  const double range_resolution_ = (double)range_res_f;
  const double static_threshold_ = (double)z_min_f;
  const double min_distance_ = (double)min_distance_f;
  const double N = window_size_ * 2;                                        // cfar.cpp:32
  const double scaling_factor_ = N * (std::pow(false_alarm_rate_, -1. / N) - 1.);   // :12-16
  int n = 0;
  for (int azimuth_nb = 0; azimuth_nb < rows; azimuth_nb++) {
    const uint8_t* azimuth = img + (size_t)azimuth_nb * stride;
    const double theta = (double(azimuth_nb + 1) / rows) * 2. * M_PI;       // :40
    for (int range_bin = 0; range_bin < cols; range_bin++) {
      const double range = range_resolution_ * double(range_bin);
      const double intensity = double(azimuth[range_bin]);
      if (range > min_distance_ && range < max_distance_ && intensity > static_threshold_) {   // :45
        const int trailing_window_start = std::max(0, range_bin - nb_guard_cells_ - window_size_);
        const int trailing_window_end = range_bin - nb_guard_cells_;
        const double trailing_mean = cfar_get_mean(azimuth, trailing_window_start, trailing_window_end);
        const int forwarding_window_start = range_bin + nb_guard_cells_;
        const int forwarding_window_end = std::min(cols, range_bin + nb_guard_cells_ + window_size_);
        const double forwarding_mean = cfar_get_mean(azimuth, forwarding_window_start, forwarding_window_end);
        const double mean = (trailing_mean + forwarding_mean) / 2.0;
        const double threshold = scaling_factor_ * mean;
        const double squared_intensity = std::pow(intensity, 2.);
        if (squared_intensity > threshold) {                                // :60
          if (n >= cap) return -1;
          xyzi[4 * n + 0] = (float)(range * std::cos(theta));
          xyzi[4 * n + 1] = (float)(range * std::sin(theta));
          xyzi[4 * n + 2] = 0.f;
          xyzi[4 * n + 3] = (float)intensity;
          if (det_rc) { det_rc[2 * n] = azimuth_nb; det_rc[2 * n + 1] = range_bin; }
          n++;
        }
      }
    }
  }
  return n;
}

// ==========================================================================================
// C: motion compensation  (utils.h:28-32 GetRelTimeStamp; utils.cpp:96-107 Compensate;
//    utils.cpp:130-146 getScaledRotationMatrix / getScaledTranslationVector)
// ==========================================================================================
namespace {
inline double get_rel_time_stamp(const double x, const double y, const bool ccw) {
  double a = std::atan2(y, x);
  double d = ((a > 0.00001 ? a : (2 * M_PI + a)) / (2 * M_PI));
  return ccw ? -(d - 0.5) : (d - 0.5);
}
}  // namespace

extern "C" void orc_compensate(float* xyzi, int n, const double mot[3], int ccw) {
  for (int i = 0; i < n; i++) {
    const float px = xyzi[4 * i], py = xyzi[4 * i + 1];
    const double d = get_rel_time_stamp(px, py, ccw != 0);
    const double s_1 = std::sin(d * mot[2]), c_1 = std::cos(d * mot[2]);
    const double tx = d * mot[0], ty = d * mot[1];
    const double x = (double)px, y = (double)py;
    xyzi[4 * i + 0] = (float)((c_1 * x + (-s_1) * y) + tx);
    xyzi[4 * i + 1] = (float)((s_1 * x + c_1 * y) + ty);
  }
}

// ==========================================================================================
// N: oriented surface points
//   MapPointNormal ctor        pointnormal.cpp:65-90
//   ComputeNormals             pointnormal.cpp:265-297 (pcl::VoxelGrid + radiusSearchT >= 6)
//   cell::cell                 pointnormal.cpp:7-36
//   cell::ComputeNormal        pointnormal.cpp:37-63
// ==========================================================================================
namespace {

struct VoxelOut { float x, y; };

// pcl::VoxelGrid<PointXYZI>::applyFilter (PCL 1.10, downsample_all_data_=true,
// min_points_per_voxel_=0); SURVEY Appendix B.1.  z == 0 for every point of this path.
void voxel_grid(const float* xyzi, int n, float leaf, std::vector<VoxelOut>& out) {
  out.clear();
  if (n == 0) return;
  const float inv_leaf = 1.0f / leaf;
  float minx = FLT_MAX, miny = FLT_MAX, maxx = -FLT_MAX, maxy = -FLT_MAX;
  for (int i = 0; i < n; i++) {
    const float x = xyzi[4 * i], y = xyzi[4 * i + 1];
    minx = std::min(minx, x); maxx = std::max(maxx, x);
    miny = std::min(miny, y); maxy = std::max(maxy, y);
  }
  const int min_bx = (int)std::floor(minx * inv_leaf), max_bx = (int)std::floor(maxx * inv_leaf);
  const int min_by = (int)std::floor(miny * inv_leaf);
  const int div_bx = max_bx - min_bx + 1;
  std::vector<std::pair<unsigned, int>> index_vector;
  index_vector.reserve(n);
  for (int i = 0; i < n; i++) {
    const int ijk0 = (int)(std::floor(xyzi[4 * i] * inv_leaf) - (float)min_bx);
    const int ijk1 = (int)(std::floor(xyzi[4 * i + 1] * inv_leaf) - (float)min_by);
    index_vector.emplace_back((unsigned)(ijk0 + ijk1 * div_bx), i);
  }
  // canonical order: stable on the voxel index (std::sort in PCL is unstable -> unspecified)
  std::stable_sort(index_vector.begin(), index_vector.end(),
                   [](const std::pair<unsigned, int>& a, const std::pair<unsigned, int>& b) {
                     return a.first < b.first;
                   });
  size_t first = 0;
  while (first < index_vector.size()) {
    size_t last = first + 1;
    while (last < index_vector.size() && index_vector[last].first == index_vector[first].first) ++last;
    float sx = 0.f, sy = 0.f;                       // CentroidPoint: AccumulatorXYZ float sums
    for (size_t li = first; li < last; ++li) {
      sx += xyzi[4 * index_vector[li].second];
      sy += xyzi[4 * index_vector[li].second + 1];
    }
    const float cnt = (float)(last - first);
    out.push_back(VoxelOut{sx / cnt, sy / cnt});
    first = last;
  }
}

// cell::cell + cell::ComputeNormal.  idx = neighbour indices in radius-search order.
bool make_cell(const float* xyzi, const std::vector<int>& idx, bool weight_intensity,
               const double origin[2], orc_cell* c) {
  const size_t N = idx.size();
  std::vector<double> w(N), x0(N), x1(N);
  for (size_t i = 0; i < N; i++) {                                            // :13-16
    x0[i] = (double)xyzi[4 * idx[i]];
    x1[i] = (double)xyzi[4 * idx[i] + 1];
    w[i] = weight_intensity ? std::max((double)xyzi[4 * idx[i] + 3] - 60.0, 0.0) : 1.0;
  }
  double sum_intensity = 0.0;
  for (size_t i = 0; i < N; i++) sum_intensity += w[i];                       // :18
  const double avg_intensity = sum_intensity / (double)N;                     // :19
  for (size_t i = 0; i < N; i++) w[i] = w[i] / sum_intensity;                 // :21
  double u0 = 0.0, u1 = 0.0;
  for (size_t i = 0; i < N; i++) { u0 += w[i] * x0[i]; u1 += w[i] * x1[i]; }  // :23-24
  for (size_t i = 0; i < N; i++) { x0[i] -= u0; x1[i] -= u1; }                // :26-27
  double c00 = 0, c01 = 0, c10 = 0, c11 = 0;                                  // :29-33 x^T * (w .* x)
  for (size_t i = 0; i < N; i++) {
    const double xw0 = w[i] * x0[i], xw1 = w[i] * x1[i];
    c00 += x0[i] * xw0; c01 += x0[i] * xw1;
    c10 += x1[i] * xw0; c11 += x1[i] * xw1;
  }
  double lmin, lmax, vmin[2], vmax[2];
  sym2_eig(c00, c10, c11, &lmin, &lmax, vmin, vmax);                          // :39-45
  const double condition_number = std::fabs(lmax / lmin);                     // :53
  const double determinant = lmax * lmin;
  const double det_tolerance = 0.00001;
  const bool cov_reasonable = (condition_number <= 10000) && (determinant > det_tolerance) &&
                              lmin > 0 && lmax > 0;                           // :56
  double n0 = vmin[0], n1 = vmin[1];
  if (n0 * (origin[0] - u0) + n1 * (origin[1] - u1) < 0) { n0 = -n0; n1 = -n1; }   // :59-61
  c->mean[0] = u0; c->mean[1] = u1;
  c->normal[0] = n0; c->normal[1] = n1;
  c->cov[0] = c00; c->cov[1] = c01; c->cov[2] = c10; c->cov[3] = c11;
  c->scale = std::log(1.0 + condition_number / 2);                            // :57
  c->avg_intensity = avg_intensity;
  c->lambda_min = lmin; c->lambda_max = lmax;
  c->nsamples = (int32_t)N; c->pad = 0;
  return cov_reasonable;
}

}  // namespace

extern "C" int orc_surface_points(const float* xyzi, int n, float radius, double downsample_factor,
                                  const double origin[2], int weight_intensity, orc_cell* cells,
                                  int cap, float* centroids, int* n_voxels) {
  if (n_voxels) *n_voxels = 0;
  if (n <= 0) return 0;                         // reference: "error, cloud empty" + exit(0) (:72-75)
  const float leaf = (float)(radius / downsample_factor);                     // :279
  std::vector<VoxelOut> vox;
  voxel_grid(xyzi, n, leaf, vox);
  if (n_voxels) *n_voxels = (int)vox.size();
  if (centroids)
    for (size_t v = 0; v < vox.size(); v++) { centroids[2 * v] = vox[v].x; centroids[2 * v + 1] = vox[v].y; }
  // pcl KdTreeFLANN::radiusSearch passes static_cast<float>(radius*radius) (double product)
  const float r2 = (float)((double)radius * (double)radius);
  int ncell = 0;
  std::vector<std::pair<float, int>> nb;
  std::vector<int> idx;
  for (size_t v = 0; v < vox.size(); v++) {
    nb.clear();
    for (int i = 0; i < n; i++) {
      // FLANN L2_Simple: float accumulation of squared differences, x then y (z == 0)
      const float dx = vox[v].x - xyzi[4 * i], dy = vox[v].y - xyzi[4 * i + 1];
      float d = 0.f;
      d += dx * dx;
      d += dy * dy;
      if (d < r2) nb.emplace_back(d, i);        // RadiusResultSet: strict <
    }
    if (nb.size() >= 6) {                                                     // :291
      std::stable_sort(nb.begin(), nb.end(),
                       [](const std::pair<float, int>& a, const std::pair<float, int>& b) {
                         return a.first < b.first;
                       });                       // sorted_results=true; ties by input index
      idx.resize(nb.size());
      for (size_t i = 0; i < nb.size(); i++) idx[i] = nb[i].second;
      orc_cell c;
      if (make_cell(xyzi, idx, weight_intensity != 0, origin, &c)) {          // :292-294
        if (ncell >= cap) return -1;
        cells[ncell++] = c;
      }
    }
  }
  return ncell;
}

// ==========================================================================================
// M: registration
// ==========================================================================================
namespace {

struct Assoc {          // one residual block (n_scan_normal.cpp:264-318)
  int tar_scan, tar_idx, src_idx;
  double weight;        // weight_after_loss
  double tar_mean_w[2]; // Ttar * tar_mean
  double tar_n_w[2];    // Ttar.linear() * tar_normal     (P2L)
  double L[4];          // sqrt information (lower)        (P2D)
  double src_mean[2];   // source mean in its local frame
};

// registration.cpp:67-75 Weights::GetWeight
double similarity(double x, double y) { return 2 * std::min(x, y) / (x + y); }
double get_weight(int opt, double N1, double N2, double sim_dir, double plan1, double plan2) {
  switch (opt) {
    case 0: return 1.0;
    case 1: return similarity(N1, N2);
    case 2: return sim_dir;
    case 3: return similarity(plan1, plan2);
    case 4: return similarity(N1, N2) + sim_dir + similarity(plan1, plan2);
  }
  return 1.0;
}

// MapPointNormal::GetClosestIdx (pointnormal.cpp:238-254): nearest float mean to float(p), accepted iff its float
// squared distance is < d * d (float promoted to double); lowest index on ties (App. B.3).
int closest_idx(const std::vector<float>& tx, const std::vector<float>& ty, double px, double py, double d) {
  const float qx = (float)px, qy = (float)py;                                 // :240-242
  int best = -1;
  float bestd = 0.f;
  for (int j = 0; j < (int)tx.size(); j++) {
    const float dx = qx - tx[j], dy = qy - ty[j];
    float dd = 0.f;
    dd += dx * dx;
    dd += dy * dy;
    if (best < 0 || dd < bestd) { best = j; bestd = dd; }
  }
  return (best >= 0 && (double)bestd < d * d) ? best : -1;                    // :250
}

// n_scan_normal.cpp:213-261 AddScanPairCost (association part) + :264-318 (block data)
void add_scan_pair(const orc_cell* tar, int n_tar, const orc_cell* src, int n_src, const Aff2& Ttar,
                   const Aff2& Tsrc, int scan_idx_tar, int itr, const orc_reg_params& par,
                   std::vector<Assoc>& out) {
  const double angle_outlier = std::cos(M_PI / 6.0);                          // :217
  const double curr_radius = (itr == 1) ? 2 * par.radius : par.radius;        // :220
  const Aff2 Tsrctotar = aff_mul(aff_inv(Ttar), Tsrc);                        // :222
  if (n_tar <= 0) return;
  std::vector<float> tx(n_tar), ty(n_tar);   // pointnormal.cpp:151-162: float PointXY of the means
  for (int j = 0; j < n_tar; j++) { tx[j] = (float)tar[j].mean[0]; ty[j] = (float)tar[j].mean[1]; }
  for (int s = 0; s < n_src; s++) {
    const double px = Tsrctotar.l[0] * src[s].mean[0] + Tsrctotar.l[1] * src[s].mean[1] + Tsrctotar.t[0];
    const double py = Tsrctotar.l[2] * src[s].mean[0] + Tsrctotar.l[3] * src[s].mean[1] + Tsrctotar.t[1];
    const float qx = (float)px, qy = (float)py;                               // pointnormal.cpp:240-242
    // KdTreeFLANN<PointXY>::nearestKSearch(k=1): exact, float L2_Simple distance
    int best = -1; float bestd = 0.f;
    for (int j = 0; j < n_tar; j++) {
      const float dx = qx - tx[j], dy = qy - ty[j];
      float d = 0.f;
      d += dx * dx;
      d += dy * dy;
      if (best < 0 || d < bestd) { best = j; bestd = d; }
    }
    if (!((double)bestd < curr_radius * curr_radius)) continue;               // pointnormal.cpp:250
    const double nsx = Tsrctotar.l[0] * src[s].normal[0] + Tsrctotar.l[1] * src[s].normal[1];
    const double nsy = Tsrctotar.l[2] * src[s].normal[0] + Tsrctotar.l[3] * src[s].normal[1];
    const double direction_similarity =
        std::max(nsx * tar[best].normal[0] + nsy * tar[best].normal[1], 0.0);  // :244
    if (direction_similarity > angle_outlier) {                               // :245
      Assoc a;
      a.tar_scan = scan_idx_tar; a.tar_idx = best; a.src_idx = s;
      a.weight = get_weight(par.weight_opt, (double)src[s].nsamples, (double)tar[best].nsamples,
                            direction_similarity, src[s].scale, tar[best].scale);   // :247-253, :273
      a.tar_mean_w[0] = Ttar.l[0] * tar[best].mean[0] + Ttar.l[1] * tar[best].mean[1] + Ttar.t[0];
      a.tar_mean_w[1] = Ttar.l[2] * tar[best].mean[0] + Ttar.l[3] * tar[best].mean[1] + Ttar.t[1];
      a.tar_n_w[0] = Ttar.l[0] * tar[best].normal[0] + Ttar.l[1] * tar[best].normal[1];
      a.tar_n_w[1] = Ttar.l[2] * tar[best].normal[0] + Ttar.l[3] * tar[best].normal[1];
      a.src_mean[0] = src[s].mean[0]; a.src_mean[1] = src[s].mean[1];
      a.L[0] = a.L[1] = a.L[2] = a.L[3] = 0.0;
      if (par.cost == 2) {                                                    // :288-297 P2D
        // tar_cov = (reg*I + R*Sigma*R^T) * cov_scale ; L = chol(tar_cov^-1) lower
        const double* S = tar[best].cov;
        const double a00 = Ttar.l[0] * S[0] + Ttar.l[1] * S[2], a01 = Ttar.l[0] * S[1] + Ttar.l[1] * S[3];
        const double a10 = Ttar.l[2] * S[0] + Ttar.l[3] * S[2], a11 = Ttar.l[2] * S[1] + Ttar.l[3] * S[3];
        double c00 = (par.regularization + (a00 * Ttar.l[0] + a01 * Ttar.l[1])) * par.cov_scale;
        double c01 = (0.0 + (a00 * Ttar.l[2] + a01 * Ttar.l[3])) * par.cov_scale;
        double c10 = (0.0 + (a10 * Ttar.l[0] + a11 * Ttar.l[1])) * par.cov_scale;
        double c11 = (par.regularization + (a10 * Ttar.l[2] + a11 * Ttar.l[3])) * par.cov_scale;
        const double det = c00 * c11 - c10 * c01, invdet = 1.0 / det;
        const double i00 = c11 * invdet, i10 = -c10 * invdet, i11 = c00 * invdet;
        const double l00 = std::sqrt(i00);                                    // LLT of the inverse
        const double l10 = i10 / l00;
        const double l11 = std::sqrt(i11 - l10 * l10);
        a.L[0] = l00; a.L[1] = 0.0; a.L[2] = l10; a.L[3] = l11;
      }
      out.push_back(a);
    }
  }
}

// n_scan_normal.cpp:342-366 BuildOptimizationProblem (mode incremental_last_to_previous: every
// fixed scan i < last is paired with the free last scan)
void build_problem(const orc_cell* const* scans, const int32_t* n_cells, int n_scans,
                   const double* poses, int itr, const orc_reg_params& par, std::vector<Assoc>& blocks) {
  blocks.clear();
  std::vector<Aff2> Tvek(n_scans);
  for (int i = 0; i < n_scans; i++) Tvek[i] = aff_from_xyt(poses[3 * i], poses[3 * i + 1], poses[3 * i + 2]);
  const int j = n_scans - 1;
  for (int i = 0; i < n_scans - 1; i++)
    add_scan_pair(scans[i], n_cells[i], scans[j], n_cells[j], Tvek[i], Tvek[j], i, itr, par, blocks);
}

int residuals_per_block(const orc_reg_params& par) { return par.cost == 1 ? 1 : 2; }

// ceres::LossFunction::Evaluate for the losses GetLoss() can return (registration.cpp:77-96),
// wrapped by ScaledLoss(loss, w) (n_scan_normal.cpp:275).  rho[0..2] = rho, rho', rho''.
void loss_eval(int loss, double a, double w, double s, double rho[3]) {
  switch (loss) {
    case 1: {  // HuberLoss(a)
      const double b = a * a;
      if (s > b) {
        const double r = std::sqrt(s);
        rho[0] = 2.0 * a * r - b;
        rho[1] = std::max(std::numeric_limits<double>::min(), a / r);
        rho[2] = -rho[1] / (2.0 * s);
      } else { rho[0] = s; rho[1] = 1.0; rho[2] = 0.0; }
      break;
    }
    case 2: {  // CauchyLoss(a)
      const double b = a * a, c = 1.0 / b;
      const double sum = 1.0 + s * c, inv = 1.0 / sum;
      rho[0] = b * std::log(sum);
      rho[1] = std::max(std::numeric_limits<double>::min(), inv);
      rho[2] = -c * (inv * inv);
      break;
    }
    case 3: {  // SoftLOneLoss(a)
      const double b = a * a, c = 1.0 / b;
      const double sum = 1.0 + s * c, tmp = std::sqrt(sum);
      rho[0] = 2.0 * b * (tmp - 1.0);
      rho[1] = std::max(std::numeric_limits<double>::min(), 1.0 / tmp);
      rho[2] = -(c * rho[1]) / (2.0 * sum);
      break;
    }
    case 4: {  // ComposedLoss(Huber(1), Cauchy(1)): f(g(s))
      double rg[3], rf[3];
      loss_eval(2, 1.0, 1.0, s, rg);
      loss_eval(1, 1.0, 1.0, rg[0], rf);
      rho[0] = rf[0];
      rho[1] = rf[1] * rg[1];
      rho[2] = rf[2] * rg[1] * rg[1] + rf[1] * rg[2];
      break;
    }
    case 5: {  // TukeyLoss(a)
      const double a2 = a * a;
      if (s <= a2) {
        const double value = 1.0 - s / a2, value_sq = value * value;
        rho[0] = a2 / 3.0 * (1.0 - value_sq * value);
        rho[1] = value_sq;
        rho[2] = -2.0 / a2 * value;
      } else { rho[0] = a2 / 3.0; rho[1] = 0.0; rho[2] = 0.0; }
      break;
    }
    default:   // None: ScaledLoss(nullptr, w)
      rho[0] = s; rho[1] = 1.0; rho[2] = 0.0;
  }
  rho[0] *= w; rho[1] *= w; rho[2] *= w;       // ScaledLoss
}

// Evaluate all residual blocks at x (Ceres ResidualBlock::Evaluate + Corrector with alpha = 0,
// valid because rho'' <= 0 for every loss above).  r: robustified residuals, J: robustified
// Jacobian rows (n x 3, row-major), both optional.  Returns 1/2 sum rho(s).
double evaluate(const std::vector<Assoc>& blocks, const orc_reg_params& par, const double x[3],
                std::vector<double>* r, std::vector<double>* J) {
  const int rpb = residuals_per_block(par);
  const double c = std::cos(x[2]), s = std::sin(x[2]);
  if (r) r->assign(blocks.size() * rpb, 0.0);
  if (J) J->assign(blocks.size() * rpb * 3, 0.0);
  double cost = 0.0;
  for (size_t b = 0; b < blocks.size(); b++) {
    const Assoc& a = blocks[b];
    // transformed_mean_src = R(theta) * src_mean + t   (n_scan_normal.h:194-197)
    const double sx = (c * a.src_mean[0] + (-s) * a.src_mean[1]) + x[0];
    const double sy = (s * a.src_mean[0] + c * a.src_mean[1]) + x[1];
    // d(R s)/dtheta
    const double dx = -s * a.src_mean[0] - c * a.src_mean[1];
    const double dy = c * a.src_mean[0] - s * a.src_mean[1];
    double res[2] = {0, 0}, jac[6] = {0, 0, 0, 0, 0, 0};
    if (par.cost == 1) {            // P2LEfficientCost  n_scan_normal.h:180-213
      const double v0 = sx - a.tar_mean_w[0], v1 = sy - a.tar_mean_w[1];
      res[0] = v0 * a.tar_n_w[0] + v1 * a.tar_n_w[1];
      jac[0] = a.tar_n_w[0]; jac[1] = a.tar_n_w[1];
      jac[2] = dx * a.tar_n_w[0] + dy * a.tar_n_w[1];
    } else if (par.cost == 0) {     // P2PEfficientCost  n_scan_normal.h:330-361
      res[0] = a.tar_mean_w[0] - sx;
      res[1] = a.tar_mean_w[1] - sy;
      jac[0] = -1.0; jac[1] = 0.0; jac[2] = -dx;
      jac[3] = 0.0; jac[4] = -1.0; jac[5] = -dy;
    } else {                        // P2DEfficientCost  n_scan_normal.h:216-255
      const double v0 = sx - a.tar_mean_w[0], v1 = sy - a.tar_mean_w[1];
      res[0] = a.L[0] * v0 + a.L[1] * v1;
      res[1] = a.L[2] * v0 + a.L[3] * v1;
      jac[0] = a.L[0]; jac[1] = a.L[1]; jac[2] = a.L[0] * dx + a.L[1] * dy;
      jac[3] = a.L[2]; jac[4] = a.L[3]; jac[5] = a.L[2] * dx + a.L[3] * dy;
    }
    double sq = 0.0;
    for (int i = 0; i < rpb; i++) sq += res[i] * res[i];
    double rho[3];
    loss_eval(par.loss, par.loss_limit, a.weight, sq, rho);
    cost += 0.5 * rho[0];
    const double sqrt_rho1 = std::sqrt(rho[1]);       // Corrector: residual_scaling_, alpha = 0
    for (int i = 0; i < rpb; i++) {
      if (r) (*r)[b * rpb + i] = res[i] * sqrt_rho1;
      if (J) for (int k = 0; k < 3; k++) (*J)[(b * rpb + i) * 3 + k] = jac[i * 3 + k] * sqrt_rho1;
    }
  }
  return cost;
}

struct IterSummary { double cost; double relative_decrease; bool successful; double radius = 0.0; };
struct SolveSummary {
  std::vector<IterSummary> iterations;
  double initial_cost = 0, final_cost = 0;
  int num_residuals = 0;
  bool usable = false;
};

bool chol3_solve(const double A[9], const double b[3], double y[3]) {
  double L[9] = {0};
  for (int i = 0; i < 3; i++)
    for (int j = 0; j <= i; j++) {
      double sum = A[i * 3 + j];
      for (int k = 0; k < j; k++) sum -= L[i * 3 + k] * L[j * 3 + k];
      if (i == j) { if (!(sum > 0.0)) return false; L[i * 3 + i] = std::sqrt(sum); }
      else L[i * 3 + j] = sum / L[j * 3 + j];
    }
  double z[3];
  for (int i = 0; i < 3; i++) { double sum = b[i]; for (int k = 0; k < i; k++) sum -= L[i * 3 + k] * z[k]; z[i] = sum / L[i * 3 + i]; }
  for (int i = 2; i >= 0; i--) { double sum = z[i]; for (int k = i + 1; k < 3; k++) sum -= L[k * 3 + i] * y[k]; y[i] = sum / L[i * 3 + i]; }
  return std::isfinite(y[0]) && std::isfinite(y[1]) && std::isfinite(y[2]);
}

// ceres::Solve with Solver::Options defaults except max_num_iterations (n_scan_normal.cpp:433-450):
// TRUST_REGION / LEVENBERG_MARQUARDT, normal-equation linear solve, jacobi_scaling, monotonic steps
// (Ceres 2.1.0 trust_region_minimizer.cc, levenberg_marquardt_strategy.cc,
// trust_region_step_evaluator.cc, solver.cc SetSummaryFinalCost).
void lm_solve(const std::vector<Assoc>& blocks, const orc_reg_params& par, double x[3], int max_iter,
              SolveSummary& sum) {
  const double function_tolerance = 1e-6, gradient_tolerance = 1e-10, parameter_tolerance = 1e-8;
  const double min_relative_decrease = 1e-3, min_lm_diagonal = 1e-6, max_lm_diagonal = 1e32;
  const double max_radius = 1e16, min_radius = 1e-32;
  double radius = 1e4, decrease_factor = 2.0;
  bool reuse_diagonal = false;
  double diagonal[3] = {0, 0, 0};
  int num_consecutive_invalid_steps = 0;
  const int rpb = residuals_per_block(par);
  const size_t n = blocks.size() * rpb;

  sum.iterations.clear();
  sum.num_residuals = (int)n;
  sum.usable = false;

  std::vector<double> r, J;
  double scale[3];
  double g[3];
  // ---- IterationZero / EvaluateGradientAndJacobian
  double x_cost = evaluate(blocks, par, x, &r, &J);
  auto gradient = [&]() {
    g[0] = g[1] = g[2] = 0.0;
    for (size_t i = 0; i < n; i++) for (int k = 0; k < 3; k++) g[k] += J[i * 3 + k] * r[i];
  };
  gradient();   // evaluator computes g = J^T r before the column scaling
  for (int k = 0; k < 3; k++) {
    double nrm = 0.0;
    for (size_t i = 0; i < n; i++) nrm += J[i * 3 + k] * J[i * 3 + k];
    scale[k] = 1.0 / (1.0 + std::sqrt(nrm));
  }
  auto scale_columns = [&]() { for (size_t i = 0; i < n; i++) for (int k = 0; k < 3; k++) J[i * 3 + k] *= scale[k]; };
  scale_columns();
  double gradient_max_norm = std::max(std::fabs(g[0]), std::max(std::fabs(g[1]), std::fabs(g[2])));
  double x_norm = std::sqrt(x[0] * x[0] + x[1] * x[1] + x[2] * x[2]);
  sum.initial_cost = x_cost;
  IterSummary it{x_cost, 0.0, true};
  int iteration = 0;
  bool converged_or_stopped = true;   // CONVERGENCE / NO_CONVERGENCE are both "usable"

  for (;;) {
    // ---- FinalizeIterationAndCheckIfMinimizerCanContinue
    it.radius = radius;                         // iteration_summary_.trust_region_radius = strategy_->Radius()
    sum.iterations.push_back(it);
    if (iteration >= max_iter) break;                                       // NO_CONVERGENCE
    if (it.successful && gradient_max_norm <= gradient_tolerance) break;    // CONVERGENCE
    if (radius <= min_radius) break;                                        // CONVERGENCE
    iteration++;
    it = IterSummary{0.0, 0.0, false};
    // ---- ComputeTrustRegionStep (LevenbergMarquardtStrategy::ComputeStep)
    if (!reuse_diagonal) {
      for (int k = 0; k < 3; k++) {
        double d = 0.0;
        for (size_t i = 0; i < n; i++) d += J[i * 3 + k] * J[i * 3 + k];
        diagonal[k] = std::min(std::max(d, min_lm_diagonal), max_lm_diagonal);
      }
    }
    double A[9] = {0}, b[3] = {0, 0, 0};
    for (size_t i = 0; i < n; i++)
      for (int k = 0; k < 3; k++) {
        b[k] += J[i * 3 + k] * r[i];
        for (int m = 0; m < 3; m++) A[k * 3 + m] += J[i * 3 + k] * J[i * 3 + m];
      }
    for (int k = 0; k < 3; k++) {
      const double lm = std::sqrt(diagonal[k] / radius);
      A[k * 3 + k] += lm * lm;
    }
    double y[3], step[3];
    const bool solved = chol3_solve(A, b, y);
    reuse_diagonal = true;
    bool step_is_valid = false;
    double model_cost_change = 0.0;
    if (solved) {
      for (int k = 0; k < 3; k++) step[k] = -y[k];
      // model_cost_change = -(J step)^T (r + J step / 2)
      for (size_t i = 0; i < n; i++) {
        const double m = J[i * 3] * step[0] + J[i * 3 + 1] * step[1] + J[i * 3 + 2] * step[2];
        model_cost_change -= m * (r[i] + m / 2.0);
      }
      step_is_valid = model_cost_change > 0.0;
    }
    if (!step_is_valid) {                                                   // HandleInvalidStep
      if (++num_consecutive_invalid_steps >= 5) { converged_or_stopped = false; break; }   // FAILURE
      radius = radius / decrease_factor; decrease_factor *= 2.0; reuse_diagonal = true;
      it.cost = x_cost; it.successful = false; it.relative_decrease = 0.0;
      continue;
    }
    num_consecutive_invalid_steps = 0;
    double cand[3], delta[3];
    for (int k = 0; k < 3; k++) { delta[k] = step[k] * scale[k]; cand[k] = x[k] + delta[k]; }
    const double cand_cost = evaluate(blocks, par, cand, nullptr, nullptr);
    // ---- ParameterToleranceReached
    const double step_norm = std::sqrt((x[0] - cand[0]) * (x[0] - cand[0]) + (x[1] - cand[1]) * (x[1] - cand[1]) +
                                       (x[2] - cand[2]) * (x[2] - cand[2]));
    if (step_norm <= parameter_tolerance * (x_norm + parameter_tolerance)) break;    // CONVERGENCE
    // ---- FunctionToleranceReached
    const double cost_change = x_cost - cand_cost;
    if (std::fabs(cost_change) <= function_tolerance * x_cost) break;                // CONVERGENCE
    // ---- IsStepSuccessful
    it.relative_decrease = cost_change / model_cost_change;
    if (it.relative_decrease > min_relative_decrease) {                     // HandleSuccessfulStep
      for (int k = 0; k < 3; k++) x[k] = cand[k];
      x_norm = std::sqrt(x[0] * x[0] + x[1] * x[1] + x[2] * x[2]);
      x_cost = evaluate(blocks, par, x, &r, &J);
      gradient();
      scale_columns();
      gradient_max_norm = std::max(std::fabs(g[0]), std::max(std::fabs(g[1]), std::fabs(g[2])));
      it.cost = x_cost; it.successful = true;
      radius = radius / std::max(1.0 / 3.0, 1.0 - std::pow(2.0 * it.relative_decrease - 1.0, 3));
      radius = std::min(max_radius, radius);
      decrease_factor = 2.0; reuse_diagonal = false;
    } else {                                                                // StepRejected
      it.cost = cand_cost; it.successful = false;
      radius = radius / decrease_factor; decrease_factor *= 2.0; reuse_diagonal = true;
    }
  }
  // solver.cc SetSummaryFinalCost
  sum.final_cost = sum.initial_cost;
  for (const auto& s : sum.iterations) sum.final_cost = std::min(s.cost, sum.final_cost);
  sum.usable = converged_or_stopped;
}

// n_scan_normal.cpp:82-185 Register
int do_register(const orc_cell* const* scans, const int32_t* n_cells, int n_scans, double* poses,
                const orc_reg_params& par, orc_reg_result* res) {
  std::vector<Assoc> blocks;
  SolveSummary summary;
  double* xl = poses + 3 * (n_scans - 1);        // parameters.back()
  double prev_par[3] = {xl[0], xl[1], xl[2]};
  double prev_score = DBL_MAX;
  bool success = true;
  bool solved_once = false;
  int itr = 1, lm_iters = 0;
  const int rpb = residuals_per_block(par);
  for (itr = 1; itr <= par.max_outer && success; itr++) {
    build_problem(scans, n_cells, n_scans, poses, itr, par, blocks);
    success = (int)blocks.size() * rpb > 1;                                 // :368-369 NumResiduals()<=1
    if (!success) break;
    double x[3] = {xl[0], xl[1], xl[2]};
    lm_solve(blocks, par, x, par.max_inner, summary);
    lm_iters += (int)summary.iterations.size() - 1;
    success = summary.usable;
    if (success) { xl[0] = x[0]; xl[1] = x[1]; xl[2] = x[2]; solved_once = true; }
    const double current_score = summary.final_cost;
    const double rel_improvement = (prev_score - current_score) / prev_score;
    if (itr > par.min_outer) {                                              // :134-149
      if (prev_score < current_score) {
        xl[0] = prev_par[0]; xl[1] = prev_par[1]; xl[2] = prev_par[2];
        break;
      } else if (rel_improvement < par.score_tolerance) {
        break;
      } else if (summary.iterations.back().relative_decrease < par.score_tolerance ||
                 summary.iterations.size() == 1) {
        break;
      }
    }
    prev_score = current_score;
    prev_par[0] = xl[0]; prev_par[1] = xl[1]; prev_par[2] = xl[2];
  }
  (void)solved_once;
  res->outer_iters = itr;
  res->lm_iters = lm_iters;
  res->pose[0] = xl[0]; res->pose[1] = xl[1]; res->pose[2] = xl[2];
  res->final_cost = summary.final_cost;
  res->num_residuals = summary.num_residuals;
  if (success) {
    res->score = summary.final_cost / summary.num_residuals;                // :162
    res->status = 1;
    return 1;
  }
  res->score = 0.0;
  res->status = 0;
  return 0;
}

}  // namespace

extern "C" int orc_register(const orc_cell* const* scans, const int32_t* n_cells, int n_scans,
                            double* poses_xyt, const orc_reg_params* par, orc_reg_result* res) {
  std::memset(res, 0, sizeof(*res));
  if (n_scans < 2) return 0;
  return do_register(scans, n_cells, n_scans, poses_xyt, *par, res);
}

extern "C" void orc_closest_idx(const orc_cell* cells, int n_cells, const double* queries_xy, int n_queries, double d,
                                int32_t* idx) {
  std::vector<float> tx(n_cells), ty(n_cells);             // pointnormal.cpp:151-162: float PointXY of the means
  for (int j = 0; j < n_cells; j++) { tx[j] = (float)cells[j].mean[0]; ty[j] = (float)cells[j].mean[1]; }
  for (int i = 0; i < n_queries; i++) idx[i] = closest_idx(tx, ty, queries_xy[2 * i], queries_xy[2 * i + 1], d);
}

extern "C" int orc_associate(const orc_cell* const* scans, const int32_t* n_cells, int n_scans,
                             const double* poses_xyt, const orc_reg_params* par, int itr,
                             int32_t* pairs, double* weights, int cap) {
  std::vector<Assoc> blocks;
  build_problem(scans, n_cells, n_scans, poses_xyt, itr, *par, blocks);
  if ((int)blocks.size() > cap) return -1;
  for (size_t i = 0; i < blocks.size(); i++) {
    pairs[3 * i] = blocks[i].tar_scan; pairs[3 * i + 1] = blocks[i].tar_idx; pairs[3 * i + 2] = blocks[i].src_idx;
    weights[i] = blocks[i].weight;
  }
  return (int)blocks.size();
}

// Test hook: one ceres::Solve (lm_solve) on the association set built at `poses` with iteration counter `itr`,
// started from the last pose; returns the per-iteration records (cost, relative_decrease, step_is_successful,
// trust_region_radius) so that an independent restatement can be compared iterate by iterate.
extern "C" int orc_lm_trace(const orc_cell* const* scans, const int32_t* n_cells, int n_scans, const double* poses_xyt,
                            const orc_reg_params* par, int itr, int max_iter, double x_out[3], double* trace, int cap,
                            double* final_cost, int32_t* usable) {
  std::vector<Assoc> blocks;
  build_problem(scans, n_cells, n_scans, poses_xyt, itr, *par, blocks);
  double x[3] = {poses_xyt[3 * (n_scans - 1)], poses_xyt[3 * (n_scans - 1) + 1], poses_xyt[3 * (n_scans - 1) + 2]};
  SolveSummary sum;
  lm_solve(blocks, *par, x, max_iter, sum);
  x_out[0] = x[0]; x_out[1] = x[1]; x_out[2] = x[2];
  *final_cost = sum.final_cost;
  *usable = sum.usable ? 1 : 0;
  const int n = (int)sum.iterations.size();
  for (int i = 0; i < n && i < cap; i++) {
    trace[4 * i] = sum.iterations[i].cost;
    trace[4 * i + 1] = sum.iterations[i].relative_decrease;
    trace[4 * i + 2] = sum.iterations[i].successful ? 1.0 : 0.0;
    trace[4 * i + 3] = sum.iterations[i].radius;
  }
  return n;
}

// n_scan_normal.cpp:186-211 GetCost
extern "C" int orc_get_cost(const orc_cell* const* scans, const int32_t* n_cells, int n_scans,
                            const double* poses_xyt, const orc_reg_params* par, double* cost,
                            double* residuals, int32_t* n_res, double* score) {
  std::vector<Assoc> blocks;
  build_problem(scans, n_cells, n_scans, poses_xyt, par->first_itr, *par, blocks);
  const int rpb = residuals_per_block(*par);
  *n_res = 0; *cost = 0.0; *score = 0.0;
  if ((int)blocks.size() * rpb <= 1) return 0;                              // :200-203
  std::vector<double> r;
  const double* x = poses_xyt + 3 * (n_scans - 1);
  *cost = evaluate(blocks, *par, x, &r, nullptr);                           // problem_->Evaluate
  *n_res = (int32_t)r.size();
  if (residuals) std::memcpy(residuals, r.data(), r.size() * sizeof(double));
  *score = *cost / (double)std::max((int)r.size(), 1);                      // :209
  return 1;
}

extern "C" int orc_normal_eq(const orc_cell* const* scans, const int32_t* n_cells, int n_scans,
                             const double* poses_xyt, const orc_reg_params* par, int itr,
                             const double x[3], double H[9], double g[3], double* cost,
                             int32_t* n_res) {
  std::vector<Assoc> blocks;
  build_problem(scans, n_cells, n_scans, poses_xyt, itr, *par, blocks);
  std::vector<double> r, J;
  *cost = evaluate(blocks, *par, x, &r, &J);
  *n_res = (int32_t)r.size();
  for (int k = 0; k < 9; k++) H[k] = 0.0;
  for (int k = 0; k < 3; k++) g[k] = 0.0;
  for (size_t i = 0; i < r.size(); i++)
    for (int k = 0; k < 3; k++) {
      g[k] += J[i * 3 + k] * r[i];
      for (int m = 0; m < 3; m++) H[k * 3 + m] += J[i * 3 + k] * J[i * 3 + m];
    }
  return 1;
}

// ==========================================================================================
// covariance by cost sampling: OdometryKeyframeFuser::approximateCovarianceBySampling
// (odometrykeyframefuser.cpp:261-380) == loopclosure::approximateCovarianceBySampling
// (tbv_slam/src/tbv_slam/loopclosure.cpp:99-208, with its constants as arguments)
// ==========================================================================================
namespace {

// linspace<double>(start, end, num)  (loopclosure.cpp:866-890)
std::vector<double> linspace_ref(double start, double end, int num_in) {
  std::vector<double> v;
  const double num = (double)num_in;
  if (num == 0) return v;
  if (num == 1) { v.push_back(start); return v; }
  const double delta = (end - start) / (num - 1);
  for (int i = 0; i < num - 1; ++i) v.push_back(start + delta * i);
  v.push_back(end);
  return v;
}

// Minimum-norm least squares  argmin |A x - b|  (A: m x n, row-major) -- what Eigen's
// A.bdcSvd(ComputeThinU | ComputeThinV).solve(b) returns.  Restated with a complete orthogonal
// decomposition: Householder QR with column pivoting, numerical rank by the SVD-style threshold
// min(m,n) * eps relative to the largest pivot, then a second Householder pass on R's rows so that
// the basic solution becomes the minimum-norm one.  (Different algorithm than the product's Jacobi
// SVD on purpose; tests pin both against numpy.linalg.lstsq.)
std::vector<double> lstsq_min_norm(std::vector<double> A, std::vector<double> b, int m, int n) {
  std::vector<int> perm(n);
  for (int j = 0; j < n; j++) perm[j] = j;
  const int kmax = std::min(m, n);
  std::vector<double> colnorm(n);
  int rank = 0;
  double first_pivot = 0.0;
  for (int k = 0; k < kmax; k++) {
    int piv = k;
    double best = -1.0;
    for (int j = k; j < n; j++) {
      double sq = 0.0;
      for (int i = k; i < m; i++) sq += A[i * n + j] * A[i * n + j];
      if (sq > best) { best = sq; piv = j; }
    }
    const double nrm = std::sqrt(std::max(best, 0.0));
    if (k == 0) first_pivot = nrm;
    if (!(nrm > first_pivot * (double)kmax * std::numeric_limits<double>::epsilon())) break;
    if (piv != k) {
      for (int i = 0; i < m; i++) std::swap(A[i * n + k], A[i * n + piv]);
      std::swap(perm[k], perm[piv]);
    }
    // Householder vector for column k (rows k..m-1)
    const double alpha = A[k * n + k] > 0 ? -nrm : nrm;
    std::vector<double> v(m - k);
    for (int i = k; i < m; i++) v[i - k] = A[i * n + k];
    v[0] -= alpha;
    double vv = 0.0;
    for (double t : v) vv += t * t;
    if (vv > 0.0) {
      for (int j = k; j < n; j++) {
        double dot = 0.0;
        for (int i = k; i < m; i++) dot += v[i - k] * A[i * n + j];
        const double f = 2.0 * dot / vv;
        for (int i = k; i < m; i++) A[i * n + j] -= f * v[i - k];
      }
      double dot = 0.0;
      for (int i = k; i < m; i++) dot += v[i - k] * b[i];
      const double f = 2.0 * dot / vv;
      for (int i = k; i < m; i++) b[i] -= f * v[i - k];
    }
    rank++;
  }
  // R = A[0:rank, 0:n] (upper trapezoidal), c = b[0:rank].  Reduce [R11 R12] to [T 0] from the right
  // (Householder on rows, last to first) so that y = T^-1 c, x = Z^T [y; 0] has minimum norm.
  const int r = rank;
  std::vector<std::vector<double>> zv(r);          // Householder vectors acting on columns {k, r..n-1}
  std::vector<double> zbeta(r, 0.0);
  if (r < n) {
    for (int k = r - 1; k >= 0; k--) {
      // annihilate A[k, r..n-1] using column k
      double sq = A[k * n + k] * A[k * n + k];
      for (int j = r; j < n; j++) sq += A[k * n + j] * A[k * n + j];
      const double nrm = std::sqrt(sq);
      const double alpha = A[k * n + k] > 0 ? -nrm : nrm;
      std::vector<double> v(1 + n - r);
      v[0] = A[k * n + k] - alpha;
      for (int j = r; j < n; j++) v[1 + j - r] = A[k * n + j];
      double vv = 0.0;
      for (double t : v) vv += t * t;
      zv[k] = v;
      zbeta[k] = vv > 0.0 ? 2.0 / vv : 0.0;
      for (int i = 0; i <= k; i++) {               // apply to rows 0..k
        double dot = v[0] * A[i * n + k];
        for (int j = r; j < n; j++) dot += v[1 + j - r] * A[i * n + j];
        const double f = zbeta[k] * dot;
        A[i * n + k] -= f * v[0];
        for (int j = r; j < n; j++) A[i * n + j] -= f * v[1 + j - r];
      }
    }
  }
  // back substitution with the r x r upper triangular T
  std::vector<double> y(n, 0.0);
  for (int i = r - 1; i >= 0; i--) {
    double t = b[i];
    for (int j = i + 1; j < r; j++) t -= A[i * n + j] * y[j];
    y[i] = t / A[i * n + i];
  }
  if (r < n) {                                     // x_p = Z^T y : reflectors in reverse order of creation
    for (int k = 0; k < r; k++) {
      const std::vector<double>& v = zv[k];
      double dot = v[0] * y[k];
      for (int j = r; j < n; j++) dot += v[1 + j - r] * y[j];
      const double f = zbeta[k] * dot;
      y[k] -= f * v[0];
      for (int j = r; j < n; j++) y[j] -= f * v[1 + j - r];
    }
  }
  std::vector<double> x(n, 0.0);
  for (int j = 0; j < n; j++) x[perm[j]] = y[j];
  return x;
}

// eigenvalues of a symmetric 3x3 matrix, ascending (Eigen::SelfAdjointEigenSolver<Matrix3d>::eigenvalues());
// closed form (trigonometric), only the signs are consumed by the caller.
void sym3_eigenvalues(const double H[9], double ev[3]) {
  const double a = H[0], b = H[4], c = H[8], d = H[1], e = H[5], f = H[2];
  const double p1 = d * d + e * e + f * f;
  if (p1 == 0.0) {
    ev[0] = a; ev[1] = b; ev[2] = c;
    std::sort(ev, ev + 3);
    return;
  }
  const double q = (a + b + c) / 3.0;
  const double p2 = (a - q) * (a - q) + (b - q) * (b - q) + (c - q) * (c - q) + 2.0 * p1;
  const double p = std::sqrt(p2 / 6.0);
  const double B[9] = {(a - q) / p, d / p, f / p, d / p, (b - q) / p, e / p, f / p, e / p, (c - q) / p};
  const double detB = B[0] * (B[4] * B[8] - B[5] * B[7]) - B[1] * (B[3] * B[8] - B[5] * B[6]) +
                      B[2] * (B[3] * B[7] - B[4] * B[6]);
  double r = detB / 2.0;
  r = std::min(1.0, std::max(-1.0, r));
  const double phi = std::acos(r) / 3.0;
  const double e1 = q + 2.0 * p * std::cos(phi);
  const double e3 = q + 2.0 * p * std::cos(phi + 2.0 * M_PI / 3.0);
  const double e2 = 3.0 * q - e1 - e3;
  ev[0] = e3; ev[1] = e2; ev[2] = e1;
}

}  // namespace

// Returns 1 when the sampled covariance is valid (cov36 filled, row-major 6x6), else 0.
// par->first_itr must hold the registration object's leftover itr_ (GetCost picks its radius
// from it, n_scan_normal.cpp:220); final_cost / num_residuals are summary_.final_cost and
// summary_.num_residuals_reduced of the Register call that produced `poses_xyt`
// (GetCovarianceScaler, n_scan_normal.cpp:433-439; num_parameters_reduced = 3).
extern "C" int orc_cov_by_sampling(const orc_cell* const* scans, const int32_t* n_cells, int n_scans,
                                   const double* poses_xyt, const orc_reg_params* par, double final_cost,
                                   int32_t num_residuals, double xy_range, double yaw_range,
                                   int32_t samples_per_axis, double covariance_scaler, double* cov36,
                                   double* samples_out /* [n^3][4] x, y, yaw, cost; may be null */) {
  std::vector<double> T_copy(poses_xyt, poses_xyt + 3 * n_scans);
  const double* best = poses_xyt + 3 * (n_scans - 1);
  const double xy_sample_range = xy_range * 0.5, theta_range = yaw_range * 0.5;          // :276-277
  const int n = samples_per_axis, m = n * n * n;
  const std::vector<double> xy_samples = linspace_ref(-xy_sample_range, xy_sample_range, n);
  const std::vector<double> theta_samples = linspace_ref(-theta_range, theta_range, n);
  std::vector<double> sx(m), sy(m), syaw(m), sc(m);
  double sample_cost = 0;                                                                 // :282
  int vp = 0;
  for (int t = 0; t < n; t++)                                                             // :294-316
    for (int ix = 0; ix < n; ix++)
      for (int iy = 0; iy < n; iy++) {
        double* last = T_copy.data() + 3 * (n_scans - 1);
        last[0] = xy_samples[ix] + best[0];
        last[1] = xy_samples[iy] + best[1];
        last[2] = theta_samples[t] + best[2];          // AngleAxis(theta, z) * R(best): yaw angles add
        double c = 0, score = 0;
        int32_t nres = 0;
        if (orc_get_cost(scans, n_cells, n_scans, T_copy.data(), par, &c, nullptr, &nres, &score))
          sample_cost = c;                             // a failed GetCost leaves sample_cost untouched
        sx[vp] = xy_samples[ix]; sy[vp] = xy_samples[iy]; syaw[vp] = theta_samples[t]; sc[vp] = sample_cost;
        vp++;
      }
  if (samples_out)
    for (int i = 0; i < m; i++) {
      samples_out[4 * i] = sx[i]; samples_out[4 * i + 1] = sy[i]; samples_out[4 * i + 2] = syaw[i];
      samples_out[4 * i + 3] = sc[i];
    }
  // f = a x^2 + b y^2 + c z^2 + d xy + e yz + f zx + g x + h y + i z + j   (:321-337)
  std::vector<double> A((size_t)m * 10);
  for (int i = 0; i < m; i++) {
    double* r = A.data() + (size_t)i * 10;
    r[0] = sx[i] * sx[i]; r[1] = sy[i] * sy[i]; r[2] = syaw[i] * syaw[i];
    r[3] = sx[i] * sy[i]; r[4] = sy[i] * syaw[i]; r[5] = syaw[i] * sx[i];
    r[6] = sx[i]; r[7] = sy[i]; r[8] = syaw[i]; r[9] = 1.0;
  }
  const std::vector<double> q = lstsq_min_norm(A, sc, m, 10);
  const double H[9] = {2 * q[0], q[3], q[5], q[3], 2 * q[1], q[4], q[5], q[4], 2 * q[2]};   // :340-343
  double ev[3];
  sym3_eigenvalues(H, ev);
  if (!(ev[0] == ev[0]) || ev[0] <= 0.0 || ev[1] <= 0.0 || ev[2] <= 0.0) return 0;        // :355-358
  if (num_residuals - 3 == 0) return 0;                                                   // GetCovarianceScaler
  const double score_scale = final_cost / (double)(num_residuals - 3);
  // covariance_3x3 = 2.0 * hessian.inverse() * score_scale * scaler  (:365)
  const double det = H[0] * (H[4] * H[8] - H[5] * H[7]) - H[1] * (H[3] * H[8] - H[5] * H[6]) +
                     H[2] * (H[3] * H[7] - H[4] * H[6]);
  double inv[9];
  inv[0] = (H[4] * H[8] - H[5] * H[7]) / det; inv[1] = (H[2] * H[7] - H[1] * H[8]) / det; inv[2] = (H[1] * H[5] - H[2] * H[4]) / det;
  inv[3] = (H[5] * H[6] - H[3] * H[8]) / det; inv[4] = (H[0] * H[8] - H[2] * H[6]) / det; inv[5] = (H[2] * H[3] - H[0] * H[5]) / det;
  inv[6] = (H[3] * H[7] - H[4] * H[6]) / det; inv[7] = (H[1] * H[6] - H[0] * H[7]) / det; inv[8] = (H[0] * H[4] - H[1] * H[3]) / det;
  double c3[9];
  for (int k = 0; k < 9; k++) c3[k] = 2.0 * inv[k] * score_scale * covariance_scaler;
  for (int k = 0; k < 36; k++) cov36[k] = 0.0;                                             // :368-374
  for (int k = 0; k < 6; k++) cov36[k * 6 + k] = 1.0;
  cov36[0] = c3[0]; cov36[1] = c3[1]; cov36[6] = c3[3]; cov36[7] = c3[4];
  cov36[35] = c3[8];
  cov36[5] = c3[2]; cov36[11] = c3[5]; cov36[30] = c3[6]; cov36[31] = c3[7];
  return 1;
}

// ==========================================================================================
// CorAl alignment quality on radar peak clouds: CorAlRadarQuality
// (coral_alignment_quality/src/alignment_checker/AlignmentQuality.cpp:8-230) as TBV calls it
// (alignmentinterface.cpp:437-456: kstrongStructuredRadar scans built from the stored peak clouds,
// radius 1.0, ent_cfg = any -> ComputeEntropy, weight_res_intensity = false, output_overlap = true).
// Third-party semantics restated: pcl::transformPointCloud<PointXYZI, double> (PCL 1.10
// common/impl/transforms.hpp: per coordinate float(((t_i0 x + t_i1 y) + t_i2 z) + t_i3) in double),
// pcl::KdTreeFLANN<PointXY>::radiusSearch (FLANN L2_Simple float distance, RadiusResultSet keeps
// dist < r^2 strictly, results sorted by distance; equal distances by index -- canonical choice),
// Eigen colwise().mean() / x^T x as sequential sums in that order.
// ==========================================================================================
namespace {

struct Pt2 { float x, y; double intensity; };

// PoseScan::GetCloudCopy(T) (ScanType.cpp:211-215) for a planar pose (x, y, theta) followed by pcl3dto2d
std::vector<Pt2> coral_transform(const float* xyzi, int n, const Aff2& T) {
  std::vector<Pt2> out(n);
  for (int i = 0; i < n; i++) {
    const double x = (double)xyzi[4 * i], y = (double)xyzi[4 * i + 1], z = (double)xyzi[4 * i + 2];
    out[i].x = (float)(((T.l[0] * x + T.l[1] * y) + 0.0 * z) + T.t[0]);
    out[i].y = (float)(((T.l[2] * x + T.l[3] * y) + 0.0 * z) + T.t[1]);
    out[i].intensity = (double)xyzi[4 * i + 3];
  }
  return out;
}

// kd.radiusSearch(query, radius, idx, sqdist): indices sorted by (distance, index)
void coral_radius(const std::vector<Pt2>& cloud, float qx, float qy, double radius, std::vector<int>& idx) {
  const float r2 = (float)(radius * radius);
  std::vector<std::pair<float, int>> hits;
  for (int i = 0; i < (int)cloud.size(); i++) {
    const float dx = qx - cloud[i].x, dy = qy - cloud[i].y;
    const float d = dx * dx + dy * dy;                                       // L2_Simple
    if (d < r2) hits.emplace_back(d, i);                                     // RadiusResultSet::addPoint
  }
  std::sort(hits.begin(), hits.end());
  idx.clear();
  for (auto& h : hits) idx.push_back(h.second);
}

// CorAlRadarQuality::Covariance (AlignmentQuality.cpp:30-53); x: rows of (x, y)
bool coral_covariance(std::vector<double>& x, double cov[4], double mean[2]) {
  const int rows = (int)x.size() / 2;
  if (rows <= 2) return false;
  double sx = 0, sy = 0;
  for (int i = 0; i < rows; i++) { sx += x[2 * i]; sy += x[2 * i + 1]; }
  mean[0] = sx / rows; mean[1] = sy / rows;
  for (int i = 0; i < rows; i++) { x[2 * i] -= mean[0]; x[2 * i + 1] -= mean[1]; }
  double c00 = 0, c01 = 0, c11 = 0;
  for (int i = 0; i < rows; i++) { c00 += x[2 * i] * x[2 * i]; c01 += x[2 * i] * x[2 * i + 1]; c11 += x[2 * i + 1] * x[2 * i + 1]; }
  const float n = (float)rows;                                               // `float n = x.rows()`
  const double den = (double)n - 1.0;
  cov[0] = c00 * 1.0 / den; cov[1] = c01 * 1.0 / den; cov[2] = cov[1]; cov[3] = c11 * 1.0 / den;
  return true;
}

// CorAlRadarQuality::ComputeEntropy (:80-98)
bool coral_entropy(const double cs[4], const double cj[4], double& sep, double& joint) {
  const double det_j = cj[0] * cj[3] - cj[1] * cj[2];
  const double det_s = cs[0] * cs[3] - cs[1] * cs[2];
  if (std::isnan(det_s) || std::isnan(det_j)) return false;
  const double sep_entropy = 1.0 / 2.0 * std::log(2.0 * M_PI * std::exp(1.0) * det_s + 0.00000001);
  const double joint_entropy = 1.0 / 2.0 * std::log(2.0 * M_PI * std::exp(1.0) * det_j + 0.00000001);
  if (std::isnan(sep_entropy) || std::isnan(joint_entropy)) return false;
  sep = sep_entropy; joint = joint_entropy;
  return true;
}

}  // namespace

// quality = {joint_, sep_, overlap_}; returns valid_ (overlap >= 0.1).  per_point (optional)
// [n_src + n_ref][3] = joint_res_, sep_res_, sep_valid in the reference's index order (src first).
extern "C" int orc_coral_quality(const float* ref_xyzi, int n_ref, const float* src_xyzi, int n_src,
                                 const double ref_pose[3], const double src_pose[3], const double offset[3],
                                 double radius, int weight_res_intensity, double quality[3], double* per_point) {
  const Aff2 Tref = aff_from_xyt(ref_pose[0], ref_pose[1], ref_pose[2]);
  const Aff2 Tsrc = aff_mul(aff_from_xyt(src_pose[0], src_pose[1], src_pose[2]),
                            aff_from_xyt(offset[0], offset[1], offset[2]));     // src->GetAffine()*Toffset (:101)
  const std::vector<Pt2> src = coral_transform(src_xyzi, n_src, Tsrc);
  const std::vector<Pt2> ref = coral_transform(ref_xyzi, n_ref, Tref);
  const int merged = n_src + n_ref;
  std::vector<double> sep_res(merged, 100.0), joint_res(merged, 100.0);
  std::vector<char> valid(merged, 0);
  std::vector<int> is, ir;
  const int overlap_req = 1;                                                 // AlignmentQuality.h:244
  for (int pass = 0; pass < 2; pass++) {                                     // :132-176 src queries, then ref queries
    const std::vector<Pt2>& q = pass == 0 ? src : ref;
    for (int k = 0; k < (int)q.size(); k++) {
      const int index = pass == 0 ? k : n_src + k;
      coral_radius(src, q[k].x, q[k].y, radius, is);                         // GetNearby (:8-29)
      coral_radius(ref, q[k].x, q[k].y, radius, ir);
      if ((pass == 0 ? (int)ir.size() : (int)is.size()) < overlap_req) continue;
      std::vector<double> msep, mjoint;
      for (int i : is) { mjoint.push_back(src[i].x); mjoint.push_back(src[i].y); }
      for (int i : ir) { mjoint.push_back(ref[i].x); mjoint.push_back(ref[i].y); }
      const std::vector<Pt2>& own = pass == 0 ? src : ref;
      for (int i : (pass == 0 ? is : ir)) { msep.push_back(own[i].x); msep.push_back(own[i].y); }
      double cs[4], cj[4], ms[2], mj[2];
      if (coral_covariance(msep, cs, ms) && coral_covariance(mjoint, cj, mj)) {
        double se, je;
        if (coral_entropy(cs, cj, se, je)) { sep_res[index] = se; joint_res[index] = je; valid[index] = 1; }
      }
    }
  }
  double sep = 0, joint = 0, w_sum = 0;                                      // :178-194
  int count_valid = 0;
  for (int i = 0; i < merged; i++) {
    if (!valid[i]) continue;
    const double w = weight_res_intensity ? (i < n_src ? src[i].intensity : ref[i - n_src].intensity) : 1.0;
    w_sum += w;
    joint_res[i] = w * joint_res[i];
    sep_res[i] = w * sep_res[i];
    joint += joint_res[i]; sep += sep_res[i];
    count_valid++;
  }
  if (count_valid > 0) { sep /= w_sum; joint /= w_sum; }
  const double overlap = count_valid / ((double)merged);
  quality[0] = joint; quality[1] = sep; quality[2] = overlap;                // output_overlap = true
  if (per_point)
    for (int i = 0; i < merged; i++) { per_point[3 * i] = joint_res[i]; per_point[3 * i + 1] = sep_res[i]; per_point[3 * i + 2] = valid[i]; }
  return overlap < 0.1 ? 0 : 1;                                              // :197-204
}

// ==========================================================================================
// Scan Context on radar clouds: RSCManager::MakeRadarCloudContext and the SCManager distance
// (place_recognition_radar/src/place_recognition_radar/RadarScancontext.cpp:59-131,
//  Scancontext.cpp:60-268) -- the descriptor TBV builds per pose-graph node from the local map of
// peak clouds (tbv_slam loopclosure.cpp:552-590) and the column-shift distance of detectLoopClosureID.
// Descriptors are row-major [ring][sector] here (Eigen's MatrixXd is column-major; indices agree).
// ==========================================================================================
namespace {

// Scancontext.cpp:60-76 (float in, float out; atan is the float overload)
float sc_xy2theta(float x, float y) {
  if ((x >= 0) & (y >= 0)) return (float)((180 / M_PI) * std::atan(y / x));
  if ((x < 0) & (y >= 0)) return (float)(180 - ((180 / M_PI) * std::atan(y / (-x))));
  if ((x < 0) & (y < 0)) return (float)(180 + ((180 / M_PI) * std::atan(y / x)));
  if ((x >= 0) & (y < 0)) return (float)(360 - ((180 / M_PI) * std::atan((-y) / x)));
  return 0;
}

}  // namespace

// RSCManager::MakeRadarCloudContext (RadarScancontext.cpp:59-131) of the cloud translated by (0, shift_y):
// shift_y != 0 restates the augmentation pcl::transformPointCloud(cloud, VectorToAffine3dxyez({0, dy, 0}))
// (:163-170; identity rotation: x stays, y' = float(double(y) + dy)).
// desc_function: 0 = "sum", 1 = "max".  desc [num_ring * num_sector].
extern "C" void orc_sc_descriptor(const float* xyzi, int n, int num_ring, int num_sector, double max_radius,
                                  int desc_function, double desc_divider, double no_point, double shift_y,
                                  double* desc) {
  const int NO_POINT = -1000;
  const int cells = num_ring * num_sector;
  for (int i = 0; i < cells; i++) desc[i] = NO_POINT;
  for (int k = 0; k < n; k++) {
    float px = xyzi[4 * k], py = xyzi[4 * k + 1];
    if (shift_y != 0.0) {
      px = (float)(((1.0 * (double)px + 0.0 * (double)py) + 0.0 * (double)xyzi[4 * k + 2]) + 0.0);
      py = (float)(((0.0 * (double)xyzi[4 * k] + 1.0 * (double)py) + 0.0 * (double)xyzi[4 * k + 2]) + shift_y);
    }
    const float intensity = xyzi[4 * k + 3];
    const float azim_range = std::sqrt(px * px + py * py);
    const float azim_angle = sc_xy2theta(px, py);
    if (azim_range > max_radius) continue;
    double rr = std::ceil((azim_range / max_radius) * num_ring), ss = std::ceil((azim_angle / 360.0) * num_sector);
    // int(NaN) is undefined; the origin itself (0/0) is put in the first sector
    const int ring_idx = std::max(std::min(num_ring, rr == rr ? (int)rr : 1), 1);
    const int sctor_idx = std::max(std::min(num_sector, ss == ss ? (int)ss : 1), 1);
    double& d = desc[(ring_idx - 1) * num_sector + (sctor_idx - 1)];
    if (d == NO_POINT) d = intensity;
    else if (desc_function == 0) d += intensity;
    else d = std::max(d, (double)intensity);
  }
  for (int i = 0; i < cells; i++) desc[i] = desc[i] / desc_divider;       // "Divison before no_point check" (:113)
  for (int i = 0; i < cells; i++)
    if (desc[i] == NO_POINT) desc[i] = no_point;                          // only reachable when desc_divider == 1
}

// makeRingkeyFromScancontext / makeSectorkeyFromScancontext (Scancontext.cpp:239-268): row / column means
extern "C" void orc_sc_keys(const double* desc, int num_ring, int num_sector, double* ringkey, double* sectorkey) {
  for (int r = 0; r < num_ring; r++) {
    double s = 0;
    for (int c = 0; c < num_sector; c++) s += desc[r * num_sector + c];
    ringkey[r] = s / num_sector;
  }
  for (int c = 0; c < num_sector; c++) {
    double s = 0;
    for (int r = 0; r < num_ring; r++) s += desc[r * num_sector + c];
    sectorkey[c] = s / num_ring;
  }
}

// distanceBtnScanContext (Scancontext.cpp:157-189) incl. fastAlignUsingVkey (:134-154), distDirectSC (:110-131)
// and circshift (:80-100).  Returns the minimum distance; *argmin_shift the column shift of sc2.
extern "C" double orc_sc_distance(const double* sc1, const double* sc2, int num_ring, int num_sector,
                                  double search_ratio, int32_t* argmin_shift) {
  const int R = num_ring, S = num_sector;
  std::vector<double> rk(R), v1(S), v2(S);
  orc_sc_keys(sc1, R, S, rk.data(), v1.data());
  orc_sc_keys(sc2, R, S, rk.data(), v2.data());
  int argmin_vkey_shift = 0;
  double min_vkey = 10000000;
  for (int sh = 0; sh < S; sh++) {                       // vkey2 shifted right by sh: shifted[(c + sh) % S] = v2[c]
    double sq = 0;
    for (int c = 0; c < S; c++) {
      const double d = v1[(c + sh) % S] - v2[c];
      sq += d * d;
    }
    const double nrm = std::sqrt(sq);
    if (nrm < min_vkey) { argmin_vkey_shift = sh; min_vkey = nrm; }
  }
  const int SEARCH_RADIUS = (int)std::round(0.5 * search_ratio * S);
  std::vector<int> space{argmin_vkey_shift};
  for (int ii = 1; ii < SEARCH_RADIUS + 1; ii++) {
    space.push_back((argmin_vkey_shift + ii + S) % S);
    space.push_back((argmin_vkey_shift - ii + S) % S);
  }
  std::sort(space.begin(), space.end());
  int best_shift = 0;
  double min_dist = 10000000;
  for (int sh : space) {
    int eff = 0;
    double sum_sim = 0;
    for (int c = 0; c < S; c++) {                        // column c of sc1 against column (c - sh) of sc2
      const int c2 = ((c - sh) % S + S) % S;
      double n1 = 0, n2 = 0, dot = 0;
      for (int r = 0; r < R; r++) {
        const double a = sc1[r * S + c], b = sc2[r * S + c2];
        n1 += a * a; n2 += b * b; dot += a * b;
      }
      n1 = std::sqrt(n1); n2 = std::sqrt(n2);
      if ((n1 == 0) | (n2 == 0)) continue;
      sum_sim += dot / (n1 * n2);
      eff++;
    }
    eff = std::max(eff, 1);
    const double dist = 1.0 - sum_sim / eff;
    if (dist < min_dist) { best_shift = sh; min_dist = dist; }
  }
  *argmin_shift = best_shift;
  return min_dist;
}

// ==========================================================================================
// caller: OdometryKeyframeFuser  (odometrykeyframefuser.cpp:62-94, 143-259, 470-494)
// ==========================================================================================
struct orc_fuser {
  orc_fuser_params par;
  Aff2 T_prev, Tmot, Tcurrent;
  struct Keyframe { Aff2 pose; std::vector<orc_cell> cells; };
  std::vector<Keyframe> keyframes;
  double cov_current[36];
  int cov_sampled = 0;
  // wall time per stage (seconds), named like the reference's `timing` keys: "Filtering" (radar_driver.cpp:87, 111),
  // "compensate" + "build_normals", "register" (odometrykeyframefuser.cpp:253-255) -- for bench.py's cpu_baseline.stage_ms
  double t_stage[4] = {0, 0, 0, 0};
  long long n_stage_frames = 0;
};

extern "C" orc_fuser* orc_fuser_create(const orc_fuser_params* p) {
  orc_fuser* f = new orc_fuser();
  f->par = *p;
  f->T_prev = f->Tmot = f->Tcurrent = aff_identity();                       // :35-39
  for (int k = 0; k < 36; k++) f->cov_current[k] = (k % 7 == 0) ? 1.0 : 0.0;   // cov_current = Identity (:36)
  return f;
}
extern "C" void orc_fuser_last_cov(const orc_fuser* f, double cov36[36], int32_t* sampled) {
  std::memcpy(cov36, f->cov_current, sizeof(f->cov_current));
  if (sampled) *sampled = f->cov_sampled;
}
extern "C" void orc_fuser_destroy(orc_fuser* f) { delete f; }
// accumulated wall seconds {Filtering (run_sequence only), compensate, build_normals, register} and the frames they cover
extern "C" void orc_fuser_stage_times(const orc_fuser* f, double seconds[4], int64_t* n_frames) {
  for (int k = 0; k < 4; k++) seconds[k] = f->t_stage[k];
  if (n_frames) *n_frames = f->n_stage_frames;
}

extern "C" int orc_fuser_process(orc_fuser* f, float* xyzi, int n, double pose_out[3], int32_t info[4]) {
  const orc_fuser_params& par = f->par;
  info[0] = info[1] = info[2] = info[3] = 0;
  const Aff2 TprevMot = f->Tmot;                                            // :146
  const auto tw0 = std::chrono::steady_clock::now();
  if (par.compensate) {                                                     // :147-150
    double mot[3];
    aff_to_xyt(TprevMot, mot);
    orc_compensate(xyzi, n, mot, par.radar_ccw);
  }
  const auto tw1 = std::chrono::steady_clock::now();
  const double origin[2] = {0, 0};
  std::vector<orc_cell> cur(std::max(n, 1));
  int nc = orc_surface_points(xyzi, n, par.res, par.downsample_factor, origin, par.weight_intensity,
                              cur.data(), (int)cur.size(), nullptr, nullptr);   // :161
  const auto tw2 = std::chrono::steady_clock::now();
  f->t_stage[1] += std::chrono::duration<double>(tw1 - tw0).count();
  f->t_stage[2] += std::chrono::duration<double>(tw2 - tw1).count();
  f->n_stage_frames++;
  if (nc < 0) return -1;
  cur.resize(nc);
  info[0] = nc;
  const Aff2 Tguess = par.use_guess ? aff_mul(f->T_prev, TprevMot) : f->T_prev;   // :164-168
  if (f->keyframes.empty()) {                                               // :171-177
    f->keyframes.push_back(orc_fuser::Keyframe{aff_identity(), cur});
    info[1] = 1;
    aff_to_xyt(f->Tcurrent, pose_out);
    return 0;
  }
  // FormatScans :478-494
  const int ns = (int)f->keyframes.size() + 1;
  std::vector<const orc_cell*> scans(ns);
  std::vector<int32_t> ncells(ns);
  std::vector<double> poses(3 * ns);
  for (int i = 0; i < ns - 1; i++) {
    scans[i] = f->keyframes[i].cells.data();
    ncells[i] = (int32_t)f->keyframes[i].cells.size();
    aff_to_xyt(f->keyframes[i].pose, &poses[3 * i]);
  }
  scans[ns - 1] = cur.data();
  ncells[ns - 1] = nc;
  aff_to_xyt(Tguess, &poses[3 * (ns - 1)]);
  orc_reg_result rr;
  const auto tw3 = std::chrono::steady_clock::now();
  orc_register(scans.data(), ncells.data(), ns, poses.data(), &par.reg, &rr);   // :186 (result shadowed)
  f->t_stage[3] += std::chrono::duration<double>(std::chrono::steady_clock::now() - tw3).count();
  info[2] = rr.status; info[3] = rr.outer_iters;
  // cov_current = cov_vek.back() (:196): Register's constant diagonal on success (n_scan_normal.cpp:171-175),
  // FormatScans' Identity66 otherwise; replaced by the sampled covariance when enabled and valid (:203-208)
  for (int k = 0; k < 36; k++) f->cov_current[k] = 0.0;
  if (rr.status) { f->cov_current[0] = 0.1 * 0.1; f->cov_current[7] = 0.1 * 0.1; f->cov_current[35] = 0.01 * 0.01; }
  else for (int k = 0; k < 6; k++) f->cov_current[k * 7] = 1.0;
  f->cov_sampled = 0;
  if (par.estimate_cov_by_sampling) {
    orc_reg_params gp = par.reg;
    gp.first_itr = rr.outer_iters;                                          // GetCost sees the leftover itr_
    double c36[36];
    if (orc_cov_by_sampling(scans.data(), ncells.data(), ns, poses.data(), &gp, rr.final_cost, rr.num_residuals,
                            par.cov_xy_range, par.cov_yaw_range, par.cov_samples_per_axis, par.cov_scaler, c36, nullptr)) {
      std::memcpy(f->cov_current, c36, sizeof(c36));
      f->cov_sampled = 1;
    }
  }
  // Tcurrent = T_vek.back(): Register rewrites Tsrc from the parameters via vectorToAffine3d
  // whenever a solve was usable; if nothing was solved the guess is kept as given.
  Aff2 Tcurrent = aff_from_xyt(poses[3 * (ns - 1)], poses[3 * (ns - 1) + 1], poses[3 * (ns - 1) + 2]);
  // AccelerationVelocitySanityCheck :76-94
  {
    const Aff2 Tmot_current = aff_mul(aff_inv(f->T_prev), Tcurrent);        // :197
    const double dt = 0.25, vel_limit = 200, acc_limit = 200;
    const double vel = std::sqrt((Tmot_current.t[0] / dt) * (Tmot_current.t[0] / dt) +
                                 (Tmot_current.t[1] / dt) * (Tmot_current.t[1] / dt));
    const double ax = (Tmot_current.t[0] - f->Tmot.t[0]) / (dt * dt), ay = (Tmot_current.t[1] - f->Tmot.t[1]) / (dt * dt);
    const double acc = std::sqrt(ax * ax + ay * ay);
    if (acc > acc_limit || vel > vel_limit) Tcurrent = Tguess;              // :198-199
  }
  f->Tmot = aff_mul(aff_inv(f->T_prev), Tcurrent);                          // :200
  // KeyFrameBasedFuse :62-73
  const Aff2 Tkeydiff = aff_mul(aff_inv(f->keyframes.back().pose), Tcurrent);
  bool fuse = true;
  if (par.use_keyframe) {
    const double tn = std::sqrt(Tkeydiff.t[0] * Tkeydiff.t[0] + Tkeydiff.t[1] * Tkeydiff.t[1]);
    const double rot = std::fabs(std::atan2(Tkeydiff.l[2], Tkeydiff.l[3]));
    fuse = tn > par.min_keyframe_dist || rot > (par.min_keyframe_rot_deg * M_PI / 180.0);
  }
  if (fuse) {                                                               // :236-250 + AddToReference :470-476
    f->keyframes.push_back(orc_fuser::Keyframe{Tcurrent, cur});
    if ((int)f->keyframes.size() > par.submap_scan_size) f->keyframes.erase(f->keyframes.begin());
    info[1] = 1;
  }
  f->Tcurrent = Tcurrent;
  f->T_prev = Tcurrent;                                                     // :257
  aff_to_xyt(Tcurrent, pose_out);
  return 0;
}

// radarDriver::CallbackOffline + OdometryKeyframeFuser::pointcloudCallback for a whole sequence in one call (the
// CPU baseline of bench.py runs one of these per host thread, like the reference's NR_WORKERS processes).
extern "C" int orc_fuser_run_sequence(orc_fuser* f, const uint8_t* imgs, int n_frames, int rows, int cols, int k, int z_min,
                                      float range_res, float min_distance, double* poses_out) {
  std::vector<int32_t> sr((size_t)rows * k), sc(rows);
  std::vector<uint8_t> si((size_t)rows * k);
  std::vector<float> xyzi((size_t)rows * k * 4);
  for (int t = 0; t < n_frames; t++) {
    const uint8_t* img = imgs + (size_t)t * rows * cols;
    const auto tf0 = std::chrono::steady_clock::now();
    orc_kstrongest(img, rows, cols, cols, k, z_min, sr.data(), si.data(), sc.data());
    const int n = orc_kstrongest_cloud(rows, k, sr.data(), si.data(), sc.data(), nullptr, range_res, min_distance, xyzi.data());
    f->t_stage[0] += std::chrono::duration<double>(std::chrono::steady_clock::now() - tf0).count();
    int32_t info[4];
    const int rc = orc_fuser_process(f, xyzi.data(), n, poses_out + 3 * (size_t)t, info);
    if (rc != 0) return rc;
  }
  return 0;
}
