// cfear_hip.hpp -- header-only C++ mirror of the reference's cfear_radarodometry classes over the C-ABI
// (include/cfear_hip.h).  Same class and method names as namespace CFEAR_Radarodometry:
//   radarDriver            radar_driver.h:32-118       CallbackOffline -> filtered cloud + peaks cloud
//   MapPointNormal         pointnormal.h:110-243       GetSize / GetCells / device handle
//   n_scan_normal_reg      n_scan_normal.h:27-85       SetParameters / Register / GetCost / getScore
// ROS / PCL / Eigen types are replaced by PODs so the header compiles anywhere a C++14 compiler and
// libcfear_hip.so exist; the Eigen/PCL adapters at the bottom are compiled only where those headers are
// installed (they are not in this image).  Errors are C++ exceptions carrying the C status code.
#pragma once
#include <cstdint>
#include <algorithm>
#include <cmath>
#include <ctime>
#include <iostream>
#include <map>
#include <memory>
#include <stdexcept>
#include <sstream>
#include <string>
#include <tuple>
#include <utility>
#include <vector>

#include "cfear_hip.h"

// Where Eigen, PCL and boost::shared_ptr exist (a ROS Noetic host; they are NOT in this image), the classes below also
// carry the reference's EXACT signatures -- Eigen::Affine3d poses, pcl::PointCloud<pcl::PointXYZI>::Ptr clouds,
// MapNormalPtr, Matrix6d covariances (n_scan_normal.h:37-41, pointnormal.h:110-118, odometrykeyframefuser.h:197-249) --
// forwarding to the POD methods.  tests/test_cpp_shim.py compiles this block against minimal stand-in headers (a syntax
// check of THIS header only; it proves nothing about Eigen or PCL).
#if defined(__has_include)
#if __has_include(<Eigen/Geometry>) && __has_include(<pcl/point_cloud.h>) && __has_include(<pcl/point_types.h>) && __has_include(<boost/shared_ptr.hpp>)
#define CFEAR_HIP_HAVE_EIGEN_PCL 1
#include <Eigen/Geometry>
#include <boost/shared_ptr.hpp>
#include <pcl/point_cloud.h>
#include <pcl/point_types.h>
#if __has_include(<cv_bridge/cv_bridge.h>)
#define CFEAR_HIP_HAVE_CV_BRIDGE 1
#include <cv_bridge/cv_bridge.h>
#endif
#endif
#endif

namespace CFEAR_Radarodometry {

#ifdef CFEAR_HIP_HAVE_EIGEN_PCL
typedef Eigen::Matrix<double, 6, 6> Matrix6d;                // registration.h
typedef Eigen::Matrix<double, 6, 6> Covariance;             // types.h
typedef enum costmetric { P2P, P2L, P2D } cost_metric;                                        // registration.h:55
typedef enum losstype { None, Huber, Cauchy, SoftLOne, Combined, Tukey } loss_type;           // registration.h:60
typedef enum weight_options { Uniform = 0, Sim_N = 1, Sim_direciton = 2, Sim_scale = 3, Combined_weights = 4 } weightoption;   // :50
#endif

struct PointXYZI { float x, y, z, intensity; };            // 16-byte PointXYZI payload
typedef std::vector<PointXYZI> PointCloud;
struct Pose2d { double x, y, theta; };                     // Affine3dToVectorXYeZ (utils.cpp:115-122)

struct CfearError : std::runtime_error {
  int status;
  CfearError(int st, const std::string& msg) : std::runtime_error(msg), status(st) {}
};

class Context {                                            // one per host thread
 public:
  explicit Context(int device = 0, void* hip_stream = nullptr) {
    const int st = cfear_ctx_create(device, hip_stream, &ctx_);
    if (st != CFEAR_OK) throw CfearError(st, cfear_status_string(st));
  }
  ~Context() { cfear_ctx_destroy(ctx_); }
  Context(const Context&) = delete;
  Context& operator=(const Context&) = delete;
  cfear_ctx* get() const { return ctx_; }
  void check(int st) const { if (st != CFEAR_OK) throw CfearError(st, cfear_last_error(ctx_)); }
  void* Stream() const { void* s = nullptr; check(cfear_ctx_get_stream(ctx_, &s)); return s; }      // the hipStream_t it enqueues on
  void SetOption(cfear_option o, int64_t v) { check(cfear_ctx_set_option(ctx_, o, v)); }           // test / measurement hooks
  // The context of the calling thread on device 0, for the reference-signature constructors that take none
  // (the reference's objects are not shared between threads either).
  static Context& Default() { static thread_local Context c(0); return c; }
 private:
  cfear_ctx* ctx_ = nullptr;
};

#ifdef CFEAR_HIP_HAVE_EIGEN_PCL
inline Pose2d Affine3dToPose2d(const Eigen::Affine3d& T) {                                    // utils.cpp:115-122
  const Eigen::Vector3d eul = T.linear().eulerAngles(0, 1, 2);
  return Pose2d{T.translation()(0), T.translation()(1), eul(2)};
}
inline Eigen::Affine3d Pose2dToAffine3d(const Pose2d& p) {                                    // registration.cpp:128-135
  return Eigen::Translation3d(p.x, p.y, 0) * Eigen::AngleAxisd(p.theta, Eigen::Vector3d::UnitZ());
}
inline PointCloud FromPcl(const pcl::PointCloud<pcl::PointXYZI>& c) {
  PointCloud out(c.points.size());
  for (size_t i = 0; i < c.points.size(); i++) out[i] = PointXYZI{c.points[i].x, c.points[i].y, c.points[i].z, c.points[i].intensity};
  return out;
}
#endif

enum filtertype { kstrong, CACFAR };                       // radar_driver.h:25
inline filtertype Str2filter(const std::string& str) { return str == "CA-CFAR" ? CACFAR : kstrong; }       // radar_driver.cpp:7-13
inline std::string Filter2str(const filtertype& filter) { return filter == CACFAR ? "CA-CFAR" : "kstrong"; }   // :15-21

class radarDriver {
 public:
  struct Parameters {                                      // radar_driver.h:35-84
    float z_min = 60; float range_res = 0.0438f; int azimuths = 400, k_strongest = 12;
    int nb_guard_cells = 20, window_size = 10; float false_alarm_rate = 0.01f;
    float min_distance = 2.5f, max_distance = 200; filtertype filter_type_ = kstrong;
    std::string dataset = "oxford";                        // anything else: images arrive as [range bins][azimuths]
    std::string topic_filtered = "/Navtech/Filtered", radar_frameid = "sensor_est";
    std::string ToString() const {                         // radar_driver.h:65-82 (what the evaluation scripts log)
      std::ostringstream ss;
      ss << "range res, " << range_res << std::endl << "z min, " << z_min << std::endl << "min distance, " << min_distance << std::endl
         << "max distance, " << max_distance << std::endl << "k strongest, " << k_strongest << std::endl
         << "topic_filtered, " << topic_filtered << std::endl << "radar_frameid, " << radar_frameid << std::endl
         << "dataset, " << dataset << std::endl << "filter type, " << Filter2str(filter_type_) << std::endl
         << "nb guard cells, " << nb_guard_cells << std::endl << "window size, " << window_size << std::endl
         << "false alarm rate, " << false_alarm_rate << std::endl;
      return ss.str();
    }
  };
  radarDriver(Context& ctx, const Parameters& pars) : ctx_(ctx), par(pars) {}
  // image: row-major uint8; rows = azimuth for dataset "oxford" (CallbackOxford, radar_driver.cpp:99-111), otherwise
  // rows = range bins and the image is rotated 90 degrees counter-clockwise first (Callback, :74-90)
  std::vector<uint8_t> cv_polar_image;                     // the image the filters saw (non-Oxford: rotated copy)
  void CallbackOffline(const uint8_t* image, int rows, int cols, int stride, PointCloud& cloud, PointCloud& cloud_peaks) {
    if (par.dataset != "oxford") {
      cv_polar_image.resize((size_t)rows * cols);
      cfear_polar_desc s{rows, cols, stride, 1, 0};
      ctx_.check(cfear_polar_rotate_ccw(ctx_.get(), image, &s, cv_polar_image.data(), rows, 0));
      image = cv_polar_image.data();
      std::swap(rows, cols);
      stride = cols;
    }
    cfear_polar_desc d{rows, cols, stride, 1, 0};
    int32_t n = 0, n_pk = 0;
    if (par.filter_type_ == CACFAR) {                      // radar_driver.cpp:52-56
      cfear_cacfar_params p{par.window_size, par.nb_guard_cells, par.false_alarm_rate, par.range_res, par.z_min,
                            par.min_distance, 400.0};
      cloud.resize((size_t)rows * cols);
      ctx_.check(cfear_filter_cacfar(ctx_.get(), image, &d, &p, &cloud[0].x, &n, rows * cols, nullptr));
      cloud.resize(n);
      cloud_peaks.clear();
      return;
    }
    cfear_kstrong_params p{par.k_strongest, par.z_min, par.range_res, par.min_distance, 1};
    cloud.resize((size_t)rows * par.k_strongest);
    cloud_peaks.resize((size_t)rows * par.k_strongest);
    cfear_kstrong_out o{};
    o.xyzi = &cloud[0].x; o.n_points = &n; o.xyzi_peaks = &cloud_peaks[0].x; o.n_peaks = &n_pk;
    ctx_.check(cfear_filter_kstrongest(ctx_.get(), image, &d, &p, &o));
    cloud.resize(n);
    cloud_peaks.resize(n_pk);
  }
 private:
  Context& ctx_;
  Parameters par;
};

inline void Compensate(Context& ctx, PointCloud& cloud, const Pose2d& mot, bool ccw) {     // utils.cpp:96-107
  const double m[3] = {mot.x, mot.y, mot.theta};
  if (!cloud.empty()) ctx.check(cfear_compensate(ctx.get(), &cloud[0].x, (int32_t)cloud.size(), m, ccw ? 1 : 0));
}

// ---- statistics.h:19-42 / statistics.cpp: the global `timing` the callers document their stage times into ----------
class statistics {
 public:
  void Document(const std::string& name, const double& value, bool report = false) {
    t_[name].push_back(value);
    if (report) std::cout << "Statistics: \"" << name << "\" = " << value << std::endl;
  }
  void PresentStatistics() {
    std::cout << "-------------- EXECUTION STATISTICS ---------------" << std::endl;
    for (const auto& r : Compute())
      std::cout << "\"" << std::get<0>(r) << "\" - mean: " << std::get<1>(r) << ", sigma: " << std::get<2>(r) << ", N: " << std::get<3>(r) << std::endl;
    std::cout << "---------FINGERS CROSSED FOR GOOD RESUTLS --------" << std::endl;
  }
  std::string GetStatistics() {                                    // the text the evaluation scripts parse
    std::string str;
    for (const auto& r : Compute()) {
      str += std::get<0>(r) + std::string(" avg, ") + std::to_string(std::get<1>(r)) + "\n";
      str += std::get<0>(r) + std::string(" dev [\xcf\x83], ") + std::to_string(std::get<2>(r)) + "\n";   // "σ"; the VARIANCE, as in the reference
      str += std::get<0>(r) + std::string(" count, ") + std::to_string(std::get<3>(r)) + "\n";
    }
    return str;
  }
 private:
  typedef std::tuple<std::string, double, double, int> reports;    // name, mean, variance, N samples
  std::vector<reports> Compute() const {
    std::vector<reports> rep;
    for (const auto& kv : t_) {
      double sum = 0, var = 0;
      for (double v : kv.second) sum += v;
      const double mean = sum / (double)kv.second.size();
      for (double v : kv.second) var += (v - mean) * (v - mean);
      rep.emplace_back(kv.first, mean, var / (double)kv.second.size(), (int)kv.second.size());
    }
    return rep;
  }
  std::map<std::string, std::vector<double>> t_;
};
inline statistics& timing_singleton() { static statistics s; return s; }
static statistics& timing = timing_singleton();                    // `extern statistics timing` (statistics.h:38), header-only
inline double ToMs(double seconds) { return seconds * 1000.0; }
template <class Duration> inline auto ToMs(const Duration& dur) -> decltype(dur.toNSec() / 1000000.0) { return dur.toNSec() / 1000000.0; }   // ros::Duration (:40)
inline double ToMsClock(const double& t) { return 1000 * ((double)t) / ((double)CLOCKS_PER_SEC); }

// ---- StructuredKStrongest (radar_filters.h:84-113, radar_filters.cpp:198-337): the constructor filters (FilterKstrongest +
//      AxialNonMaxSupress), getPeaksFilteredPointCloud APPENDS the kept bins (all of them, or the peaks) to the cloud ------
class StructuredKStrongest {
 public:
  StructuredKStrongest(Context& ctx, const uint8_t* image, int rows, int cols, int stride, const int z_min, const int k_strongest,
                       const double min_distance, const double range_res) {
    cfear_polar_desc d{rows, cols, stride, 1, 0};
    cfear_kstrong_params p{k_strongest, (float)z_min, (float)range_res, (float)min_distance, 1};
    all_.resize((size_t)rows * k_strongest);
    peaks_.resize((size_t)rows * k_strongest);
    int32_t n = 0, n_pk = 0;
    cfear_kstrong_out o{};
    o.xyzi = &all_[0].x; o.n_points = &n; o.xyzi_peaks = &peaks_[0].x; o.n_peaks = &n_pk;
    ctx.check(cfear_filter_kstrongest(ctx.get(), image, &d, &p, &o));
    all_.resize(n);
    peaks_.resize(n_pk);
  }
  void getPeaksFilteredPointCloud(PointCloud& output_pointcloud, bool peaks = false) const {
    const PointCloud& src = peaks ? peaks_ : all_;
    output_pointcloud.insert(output_pointcloud.end(), src.begin(), src.end());
  }
#ifdef CFEAR_HIP_HAVE_CV_BRIDGE
  // the reference's signature (radar_filters.h:88, :90): a MONO8 cv_bridge image, clouds as pcl pointers (created when null)
  StructuredKStrongest(const cv_bridge::CvImagePtr& radar_image, const int z_min, const int k_strongest, const double min_distance,
                       const double range_res)
      : StructuredKStrongest(Context::Default(), radar_image->image.data, radar_image->image.rows, radar_image->image.cols,
                             (int)radar_image->image.step, z_min, k_strongest, min_distance, range_res) {}
  void getPeaksFilteredPointCloud(pcl::PointCloud<pcl::PointXYZI>::Ptr& output_pointcloud, bool peaks = false) const {
    if (!output_pointcloud) output_pointcloud = pcl::PointCloud<pcl::PointXYZI>::Ptr(new pcl::PointCloud<pcl::PointXYZI>());
    for (const PointXYZI& q : (peaks ? peaks_ : all_)) {
      pcl::PointXYZI p; p.x = q.x; p.y = q.y; p.z = q.z; p.intensity = q.intensity;
      output_pointcloud->push_back(p);
    }
  }
#endif
 private:
  PointCloud all_, peaks_;
};

// k_strongest_filter (radar_filters.cpp:40-78, with InsertStrongestK :25-38): the legacy filter CorAl's kstrongRadar calls;
// APPENDS to cloud
inline void k_strongest_filter(Context& ctx, const uint8_t* image, int rows, int cols, int stride, PointCloud& cloud, int k_strongest,
                               double z_min, double range_res, double min_distance) {
  cfear_polar_desc d{rows, cols, stride, 1, 0};
  PointCloud out((size_t)rows * k_strongest);
  int32_t n = 0;
  ctx.check(cfear_filter_kstrongest_legacy(ctx.get(), image, &d, k_strongest, z_min, range_res, min_distance, out.empty() ? nullptr : &out[0].x,
                                           &n, rows * k_strongest));
  cloud.insert(cloud.end(), out.begin(), out.begin() + n);
}
#ifdef CFEAR_HIP_HAVE_CV_BRIDGE
inline void k_strongest_filter(cv_bridge::CvImagePtr& cv_polar_image, pcl::PointCloud<pcl::PointXYZI>::Ptr& cloud, int k_strongest,
                               double z_min, double range_res, double min_distance) {          // radar_filters.cpp:40
  if (!cloud) cloud = pcl::PointCloud<pcl::PointXYZI>::Ptr(new pcl::PointCloud<pcl::PointXYZI>());
  PointCloud c;
  k_strongest_filter(Context::Default(), cv_polar_image->image.data, cv_polar_image->image.rows, cv_polar_image->image.cols,
                     (int)cv_polar_image->image.step, c, k_strongest, z_min, range_res, min_distance);
  for (const PointXYZI& q : c) { pcl::PointXYZI p; p.x = q.x; p.y = q.y; p.z = q.z; p.intensity = q.intensity; cloud->push_back(p); }
}
#endif

// cell::TransformCopy (pointnormal.cpp:515-529) on the POD cell.  The covariance follows the reference's expression literally:
// C = R * T * cov_ * R.transpose() with T an Affine2d multiplies cov_'s COLUMNS as points -- (R R) cov + (R t) 1^T, then * R^T --
// not the similarity transform R cov R^T one would expect; nothing in the reference reads cov_ of a transformed map.
inline cfear_cell TransformCopy(const cfear_cell& c, const Pose2d& T) {
  const double cs = std::cos(T.theta), sn = std::sin(T.theta);
  const double R[4] = {cs, -sn, sn, cs};
  const double RR[4] = {R[0] * R[0] + R[1] * R[2], R[0] * R[1] + R[1] * R[3], R[2] * R[0] + R[3] * R[2], R[2] * R[1] + R[3] * R[3]};
  const double Rt[2] = {R[0] * T.x + R[1] * T.y, R[2] * T.x + R[3] * T.y};
  const double M[4] = {RR[0] * c.cov[0] + RR[1] * c.cov[2] + Rt[0], RR[0] * c.cov[1] + RR[1] * c.cov[3] + Rt[0],
                       RR[2] * c.cov[0] + RR[3] * c.cov[2] + Rt[1], RR[2] * c.cov[1] + RR[3] * c.cov[3] + Rt[1]};
  cfear_cell o = c;
  o.cov[0] = M[0] * R[0] + M[1] * R[1]; o.cov[1] = M[0] * R[2] + M[1] * R[3];     // M * R^T
  o.cov[2] = M[2] * R[0] + M[3] * R[1]; o.cov[3] = M[2] * R[2] + M[3] * R[3];
  o.mean[0] = R[0] * c.mean[0] + R[1] * c.mean[1] + T.x; o.mean[1] = R[2] * c.mean[0] + R[3] * c.mean[1] + T.y;
  o.normal[0] = R[0] * c.normal[0] + R[1] * c.normal[1]; o.normal[1] = R[2] * c.normal[0] + R[3] * c.normal[1];
  return o;
}

class MapPointNormal {
 public:
  static double& downsample_factor() { static double f = 1.0; return f; }                   // pointnormal.cpp:5
  MapPointNormal(Context& ctx, PointCloud& cld, float radius, double ox = 0, double oy = 0, bool weight_intensity = false)
      : ctx_(ctx) {
    cfear_feature_params fp{};
    fp.radius = radius; fp.downsample_factor = downsample_factor(); fp.origin[0] = ox; fp.origin[1] = oy;
    fp.weight_intensity = weight_intensity ? 1 : 0;
    ctx_.check(cfear_scan_create(ctx_.get(), cld.empty() ? nullptr : &cld[0].x, (int32_t)cld.size(), &fp, &scan_));
  }
  MapPointNormal(Context& ctx, const std::vector<cfear_cell>& cells) : ctx_(ctx) {           // Boost load() path
    ctx_.check(cfear_scan_from_cells(ctx_.get(), cells.data(), (int32_t)cells.size(), &scan_));
  }
#ifdef CFEAR_HIP_HAVE_EIGEN_PCL
  // pointnormal.h:116 / pointnormal.cpp:65-90.  raw: one identity cell per point (cell::GetIdentityCell, pointnormal.h:59)
  MapPointNormal(const pcl::PointCloud<pcl::PointXYZI>::Ptr& cld, float radius, const Eigen::Vector2d& origin = Eigen::Vector2d(0, 0),
                 const bool weight_intensity = false, const bool raw = false) : ctx_(Context::Default()) {
    input_ = cld;                                                                  // pointnormal.cpp:78 (GetScan hands it back)
    if (raw) {
      std::vector<cfear_cell> cells(cld->points.size());
      for (size_t i = 0; i < cells.size(); i++) {
        cfear_cell c{};
        c.mean[0] = cld->points[i].x; c.mean[1] = cld->points[i].y;
        c.normal[0] = 1; c.normal[1] = 0; c.cov[0] = 0.1; c.cov[3] = 0.1;          // cov_ default Identity * 0.1 (pointnormal.h:68)
        c.scale = 1.0; c.avg_intensity = 1.0; c.lambda_min = 1; c.lambda_max = 1; c.nsamples = 1;
        cells[i] = c;
      }
      ctx_.check(cfear_scan_from_cells(ctx_.get(), cells.data(), (int32_t)cells.size(), &scan_));
      return;
    }
    PointCloud pts = FromPcl(*cld);
    cfear_feature_params fp{};
    fp.radius = radius; fp.downsample_factor = downsample_factor(); fp.origin[0] = origin(0); fp.origin[1] = origin(1);
    fp.weight_intensity = weight_intensity ? 1 : 0;
    ctx_.check(cfear_scan_create(ctx_.get(), pts.empty() ? nullptr : &pts[0].x, (int32_t)pts.size(), &fp, &scan_));
  }
  Eigen::Vector2d GetMean2d(const size_t i) { const cfear_cell& c = cached()[i]; return Eigen::Vector2d(c.mean[0], c.mean[1]); }       // :127
  Eigen::Vector2d GetNormal2d(const size_t i) { const cfear_cell& c = cached()[i]; return Eigen::Vector2d(c.normal[0], c.normal[1]); } // :131
  Eigen::Matrix2d GetCov2d(const size_t i) {                                                                                          // :135
    const cfear_cell& c = cached()[i];
    Eigen::Matrix2d m;
    m(0, 0) = c.cov[0]; m(0, 1) = c.cov[1]; m(1, 0) = c.cov[2]; m(1, 1) = c.cov[3];
    return m;
  }
  std::vector<int> GetClosestIdx(const Eigen::Vector2d& p, double d) const { return GetClosestIdx(p(0), p(1), d); }                   // :238
  pcl::PointCloud<pcl::PointXYZI>::Ptr GetScan() { return input_; }                                                                 // :170
  // PublishMap (pointnormal.h:235) draws RViz markers through a static ros::Publisher map: visualisation, out of scope --
  // kept as a no-op so that callers (odometrykeyframefuser.cpp:209-216) compile unchanged
  template <class MapPtr, class Affine>
  static void PublishMap(const std::string&, MapPtr, const Affine&, const std::string&, const int = 0, float = 1.0f) {}
  boost::shared_ptr<MapPointNormal> TransformMap(const Eigen::Affine3d& T) {                                                          // :168
    return boost::shared_ptr<MapPointNormal>(new MapPointNormal(ctx_, TransformCells(Affine3dToPose2d(T))));
  }
#endif
  ~MapPointNormal() { cfear_scan_destroy(scan_); }
  MapPointNormal(const MapPointNormal&) = delete;
  MapPointNormal& operator=(const MapPointNormal&) = delete;
  size_t GetSize() const { return (size_t)cfear_scan_size(scan_); }
  std::vector<cfear_cell> GetCells() const {
    std::vector<cfear_cell> c(GetSize());
    if (!c.empty()) { const int n = cfear_scan_get_cells(scan_, c.data(), (int32_t)c.size()); if (n < 0) ctx_.check(n); }
    return c;
  }
  const cfear_cell& GetCell(const size_t i) { return cached()[i]; }                          // pointnormal.h:120
  std::vector<cfear_cell> TransformCells(const Pose2d& T) {                                  // pointnormal.h:140
    std::vector<cfear_cell> out = GetCells();
    for (cfear_cell& c : out) c = TransformCopy(c, T);
    return out;
  }
  // TransformMap (pointnormal.cpp:135-137 -> the constructor at :91-111): the cells moved into the frame T, searchable
  std::unique_ptr<MapPointNormal> TransformMap(const Pose2d& T) { return std::unique_ptr<MapPointNormal>(new MapPointNormal(ctx_, TransformCells(T))); }
  std::vector<int> GetClosestIdx(double px, double py, double d) const {                     // pointnormal.cpp:238-254
    const double q[2] = {px, py};
    int32_t idx = -1;
    ctx_.check(cfear_scan_closest_idx(scan_, q, 1, d, &idx));
    return idx >= 0 ? std::vector<int>{idx} : std::vector<int>{};
  }
  const cfear_scan* device() const { return scan_; }
 private:
  const std::vector<cfear_cell>& cached() { if (cache_.empty()) cache_ = GetCells(); return cache_; }
  std::vector<cfear_cell> cache_;
  Context& ctx_;
  cfear_scan* scan_ = nullptr;
#ifdef CFEAR_HIP_HAVE_EIGEN_PCL
  pcl::PointCloud<pcl::PointXYZI>::Ptr input_;
#endif
};
#ifdef CFEAR_HIP_HAVE_EIGEN_PCL
typedef boost::shared_ptr<MapPointNormal> MapNormalPtr;                                       // pointnormal.h:108
#endif

// The device views of a set of scans, uploaded once (cfear_scan_table): what the loop-closure thread keeps for the graph's
// nodes so that a candidate is two indices and two poses.  Holds no reference on the scans.
class ScanTable {
 public:
  ScanTable(Context& ctx, const std::vector<const MapPointNormal*>& scans) : ctx_(ctx), h_(nullptr) {
    std::vector<const cfear_scan*> h(scans.size());
    for (size_t i = 0; i < scans.size(); i++) h[i] = scans[i]->device();
    ctx_.check(cfear_scan_table_create(ctx_.get(), h.data(), (int32_t)h.size(), &h_));
  }
  ~ScanTable() { if (h_) cfear_scan_table_destroy(h_); }
  ScanTable(const ScanTable&) = delete;
  ScanTable& operator=(const ScanTable&) = delete;
  int size() const { return cfear_scan_table_size(h_); }
  const cfear_scan_table* get() const { return h_; }

 private:
  Context& ctx_;
  cfear_scan_table* h_;
};

class n_scan_normal_reg {
 public:
  explicit n_scan_normal_reg(Context& ctx, int cost = CFEAR_P2L, int loss = CFEAR_LOSS_HUBER, double loss_limit = 0.1,
                             int opt = CFEAR_W_UNIFORM) : ctx_(ctx) {                         // n_scan_normal.h:35
    cfear_reg_params_default(&par_);
    par_.cost = cost; par_.loss = loss; par_.loss_limit = loss_limit; par_.weight_opt = opt;
  }
  void SetParameters(unsigned max_itr_association, unsigned max_itr_solver) {                 // n_scan_normal.cpp:15-19
    par_.max_itr_association = (int32_t)max_itr_association; par_.max_itr_solver = (int32_t)max_itr_solver;
  }
  void SetD2dPar(double cov_scale, double regularization) { par_.cov_scale = cov_scale; par_.regularization = regularization; }
  // Register (n_scan_normal.cpp:82-185): Tsrc in/out; returns false exactly where the reference does.
  bool Register(const std::vector<const MapPointNormal*>& scans, std::vector<Pose2d>& Tsrc) {
    std::vector<const cfear_scan*> h(scans.size());
    for (size_t i = 0; i < scans.size(); i++) h[i] = scans[i]->device();
    const int st = cfear_register(ctx_.get(), h.data(), (int32_t)h.size(), &Tsrc[0].x, &par_, &summary_);
    if (st != CFEAR_OK && st != CFEAR_ERR_TOO_FEW_RESIDUALS && st != CFEAR_ERR_SOLVER) ctx_.check(st);
    par_.itr = summary_.outer_iters;                       // itr_ stays behind for GetCost (n_scan_normal.cpp:220)
    score_ = summary_.score;
    return st == CFEAR_OK;
  }
  bool GetCost(const std::vector<const MapPointNormal*>& scans, const std::vector<Pose2d>& Tsrc, double& score,
               std::vector<double>& residuals) {                                              // n_scan_normal.cpp:186-211
    std::vector<const cfear_scan*> h(scans.size());
    size_t total = 2;
    for (size_t i = 0; i < scans.size(); i++) { h[i] = scans[i]->device(); total += 2 * scans[i]->GetSize(); }
    residuals.resize(total);
    int32_t n = 0;
    double s = 0;
    const int st = cfear_get_cost(ctx_.get(), h.data(), (int32_t)h.size(), &Tsrc[0].x, &par_, &score, residuals.data(),
                                  (int32_t)total, &n, &s);
    if (st != CFEAR_OK && st != CFEAR_ERR_TOO_FEW_RESIDUALS) ctx_.check(st);
    residuals.resize(n);
    score_ = s;
    return st == CFEAR_OK;
  }
#ifdef CFEAR_HIP_HAVE_EIGEN_PCL
  n_scan_normal_reg() : n_scan_normal_reg(Context::Default()) {}                              // n_scan_normal.h:33
  n_scan_normal_reg(const cost_metric& cost, loss_type loss = Huber, double loss_limit = 0.1,
                    const weightoption opt = weightoption::Uniform)                           // n_scan_normal.h:35
      : n_scan_normal_reg(Context::Default(), (int)cost, (int)loss, loss_limit, (int)opt) {}
  // n_scan_normal.h:37 / n_scan_normal.cpp:82-185: on success EVERY Tsrc[i] is rewritten from its (x, y, theta) parameters
  // (:176-177) and every reg_cov[i] is the constant diag(0.01, 0.01, 0, 0, 0, 1e-4) (:171-175).
  // soft_constraints = true is REFUSED (CfearError, CFEAR_ERR_INVALID_ARGUMENT), not ignored: the reference's prior
  // (n_scan_normal.cpp:371-375) is mahalanobisDistanceError created as AutoDiffCostFunction<..., 1, 3> (n_scan_normal.h:279-282)
  // whose functor writes THREE residuals through an Eigen::Map (:271-275) -- two of them past the one-element output Ceres
  // hands it.  Its result is undefined in the reference, so there is nothing to be identical to; the shipped callers pass
  // false (offline_odometry.cpp:274, loopclosure.cpp:60).
  bool Register(std::vector<MapNormalPtr>& scans, std::vector<Eigen::Affine3d>& Tsrc, std::vector<Matrix6d>& reg_cov,
                bool soft_constraints = false) {
    if (soft_constraints)
      throw CfearError(CFEAR_ERR_INVALID_ARGUMENT, "Register(..., soft_constraints = true): the reference's prior is undefined "
                                                   "behaviour (3 residuals written into a 1-residual block) and is not provided");
    std::vector<const MapPointNormal*> h(scans.size());
    std::vector<Pose2d> poses(scans.size());
    for (size_t i = 0; i < scans.size(); i++) { h[i] = scans[i].get(); poses[i] = Affine3dToPose2d(Tsrc[i]); }
    const bool ok = Register(h, poses);
    if (ok) {
      Matrix6d m = Matrix6d::Zero();
      m(0, 0) = 0.1 * 0.1; m(1, 1) = 0.1 * 0.1; m(5, 5) = 0.01 * 0.01;
      for (size_t i = 0; i < scans.size(); i++) {
        if (i < reg_cov.size()) reg_cov[i] = m;
        Tsrc[i] = Pose2dToAffine3d(poses[i]);
      }
    }
    return ok;
  }
  bool GetCost(std::vector<MapNormalPtr>& scans, std::vector<Eigen::Affine3d>& Tsrc, double& score,
               std::vector<double>& residuals) {                                              // n_scan_normal.h:41
    std::vector<const MapPointNormal*> h(scans.size());
    std::vector<Pose2d> poses(scans.size());
    for (size_t i = 0; i < scans.size(); i++) { h[i] = scans[i].get(); poses[i] = Affine3dToPose2d(Tsrc[i]); }
    return GetCost(h, poses, score, residuals);
  }
#endif
  // loopclosure::Register for a whole list of candidate pairs among the scans of a table (loopclosure.cpp:35-97 per candidate):
  // one launch; results[i].status says what Register's bool would have.
  void RegisterCandidates(const ScanTable& table, const std::vector<cfear_candidate>& candidates, std::vector<cfear_reg_result>& results) {
    results.resize(candidates.size());
    ctx_.check(cfear_register_candidates(ctx_.get(), table.get(), candidates.data(), (int32_t)candidates.size(), &par_, results.data()));
  }
  const cfear_reg_params& params() const { return par_; }
  double getScore() const { return score_; }
  bool GetCovarianceScaler(double& cov_scale) const {                                         // n_scan_normal.cpp:433-439
    if (summary_.num_residuals - 3 == 0) return false;
    cov_scale = summary_.final_cost / (double)(summary_.num_residuals - 3);
    return true;
  }
  // OdometryKeyframeFuser::approximateCovarianceBySampling (odometrykeyframefuser.cpp:261-380) /
  // loopclosure::approximateCovarianceBySampling (loopclosure.cpp:99-208): T_vek = poses after Register of
  // THIS object; cov_sampled row-major 6x6.  Returns cov_sampled_success.
  bool approximateCovarianceBySampling(const std::vector<const MapPointNormal*>& scans, const std::vector<Pose2d>& T_vek,
                                       double cov_sampled[36], const cfear_cov_sampling_params* sp = nullptr) {
    cfear_cov_sampling_params def;
    cfear_cov_sampling_params_default(&def);
    std::vector<const cfear_scan*> h(scans.size());
    for (size_t i = 0; i < scans.size(); i++) h[i] = scans[i]->device();
    int32_t ok = 0;
    ctx_.check(cfear_covariance_by_sampling(ctx_.get(), h.data(), (int32_t)h.size(), &T_vek[0].x, &par_, &summary_,
                                            sp ? sp : &def, cov_sampled, nullptr, &ok));
    return ok != 0;
  }
  cfear_reg_result summary_{};
 private:
  Context& ctx_;
  cfear_reg_params par_;
  double score_ = 0;
};

// radarDriver + OdometryKeyframeFuser for n independent sequences, one frame per call (cfear_odometry_*): the whole
// path polar image -> pose on the GPU; with par.keep_nodes the RadarScan of every frame can be collected.
// Sharded candidate steps kept in flight (cfear_candidate_pipe): what the loop-closure thread of a rank
// calls per candidate batch when the node's GPUs share the work (loopclosure.cpp:658-721: independent candidates).  comm: the
// host's ncclComm_t in a cfear_rccl_comm (or one made by cfear_rccl_comm_init), nullptr for a single rank without a collective.
class CandidatePipe {
 public:
  CandidatePipe(Context& ctx, const ScanTable& table, int max_candidates, int rank = 0, int world = 1, const cfear_rccl_comm* comm = nullptr,
                int depth = 2, int flags = 0) : ctx_(ctx), h_(nullptr) {
    ctx_.check(cfear_candidate_pipe_create(ctx_.get(), table.get(), max_candidates, rank, world, comm, depth, flags, &h_));
  }
  ~CandidatePipe() { if (h_) cfear_candidate_pipe_destroy(h_); }
  CandidatePipe(const CandidatePipe&) = delete;
  CandidatePipe& operator=(const CandidatePipe&) = delete;
  // every rank hands in the FULL list; returns without waiting
  int64_t submit(const std::vector<cfear_candidate>& all_candidates, const n_scan_normal_reg& reg) {
    int64_t ticket = -1;
    sizes_[ticket_slot(next_)] = all_candidates.size();
    ctx_.check(cfear_candidate_pipe_submit(h_, all_candidates.data(), (int32_t)all_candidates.size(), &reg.params(), &ticket));
    next_ = ticket + 1;
    return ticket;
  }
  // all records of that batch, every rank's block, candidate order
  void collect(int64_t ticket, std::vector<cfear_reg_result>& results) {
    results.resize(sizes_[ticket_slot(ticket)]);
    cfear_reg_result dummy;
    ctx_.check(cfear_candidate_pipe_collect(h_, ticket, results.empty() ? &dummy : results.data()));
  }

 private:
  static size_t ticket_slot(int64_t t) { return (size_t)(t % 16); }          // (depth <= 16)
  Context& ctx_;
  cfear_candidate_pipe* h_;
  int64_t next_ = 0;
  size_t sizes_[16] = {};
};

class OdometryKeyframeFuser {
 public:
  OdometryKeyframeFuser(Context& ctx, int n_streams, int rows, int cols, const cfear_odometry_params* par = nullptr)
      : ctx_(ctx), n_(n_streams) {
    cfear_odometry_params def;
    if (!par) { cfear_odometry_params_default(&def); par = &def; }
    cfear_polar_desc d{rows, cols, cols, n_streams, (int64_t)rows * cols};
    ctx_.check(cfear_odometry_create(ctx_.get(), n_streams, &d, par, &od_));
  }
  ~OdometryKeyframeFuser() { cfear_odometry_destroy(od_); }
  OdometryKeyframeFuser(const OdometryKeyframeFuser&) = delete;
  OdometryKeyframeFuser& operator=(const OdometryKeyframeFuser&) = delete;
  // polar: [n_streams][rows][cols] uint8 (host or device); polar_next (optional): next frame, filtered ahead
  std::vector<cfear_frame_info> processFrame(const uint8_t* polar, const uint8_t* polar_next = nullptr) {
    std::vector<cfear_frame_info> info((size_t)n_);
    ctx_.check(cfear_odometry_process_prefetch(od_, polar, polar_next, info.data()));
    return info;
  }
  // pointcloudCallback(cloud, cloud_peaks, ...) (odometrykeyframefuser.cpp:413-426) for every stream: filtered clouds in
  std::vector<cfear_frame_info> pointcloudCallback(const std::vector<const PointCloud*>& clouds,
                                                   const std::vector<const PointCloud*>& peaks = {}) {
    std::vector<cfear_sc_cloud> c((size_t)n_), p((size_t)n_);
    for (int i = 0; i < n_; i++) {
      c[i] = cfear_sc_cloud{clouds[i]->empty() ? nullptr : &(*clouds[i])[0].x, (int32_t)clouds[i]->size(), 0};
      if (!peaks.empty()) p[i] = cfear_sc_cloud{peaks[i]->empty() ? nullptr : &(*peaks[i])[0].x, (int32_t)peaks[i]->size(), 0};
    }
    std::vector<cfear_frame_info> info((size_t)n_);
    ctx_.check(cfear_odometry_process_clouds(od_, c.data(), peaks.empty() ? nullptr : p.data(), info.data()));
    return info;
  }
#ifdef CFEAR_HIP_HAVE_EIGEN_PCL
  // The reference's single-sequence object (odometrykeyframefuser.h:72-249): its Parameters by their names, its
  // constructor, and pointcloudCallback with the reference's argument list (Time: ros::Time or anything else; unused).
  struct Parameters {
    std::string cost_type = "P2L", loss_type_ = "Huber";
    weightoption weight_opt = weightoption::Uniform;
    int submap_scan_size = 3;
    bool weight_intensity_ = false, use_guess = true, compensate = true, radar_ccw = false, use_keyframe = true;
    double res = 3.5, min_keyframe_dist_ = 1.5, min_keyframe_rot_deg_ = 5, loss_limit_ = 0.1, covar_scale_ = 1.0, regularization_ = 0.0;
    bool estimate_cov_by_sampling = false;
    double cov_sampling_xy_range = 0.4, cov_sampling_yaw_range = 0.0043625, cov_sampling_covariance_scaler = 4.0;
    unsigned int cov_sampling_samples_per_axis = 3;
    int max_points = 400 * 40;                                  // capacity of a filtered cloud (rows * k of the driver)
  };
  explicit OdometryKeyframeFuser(const Parameters& pars, bool disable_callback = false) : ctx_(Context::Default()), n_(1) {
    (void)disable_callback;
    cfear_odometry_params p;
    cfear_odometry_params_default(&p);
    p.reg.cost = pars.cost_type == "P2P" ? CFEAR_P2P : pars.cost_type == "P2D" ? CFEAR_P2D : CFEAR_P2L;        // Str2Cost
    p.reg.loss = pars.loss_type_ == "Cauchy" ? CFEAR_LOSS_CAUCHY : pars.loss_type_ == "SoftLOne" ? CFEAR_LOSS_SOFTLONE
               : pars.loss_type_ == "Combined" ? CFEAR_LOSS_COMBINED : pars.loss_type_ == "Tukey" ? CFEAR_LOSS_TUKEY
               : pars.loss_type_ == "None" ? CFEAR_LOSS_NONE : CFEAR_LOSS_HUBER;                              // Str2loss
    p.reg.loss_limit = pars.loss_limit_; p.reg.weight_opt = (int)pars.weight_opt;
    p.reg.cov_scale = pars.covar_scale_; p.reg.regularization = pars.regularization_;
    p.res = (float)pars.res; p.submap_scan_size = pars.submap_scan_size; p.weight_intensity = pars.weight_intensity_;
    p.use_guess = pars.use_guess; p.compensate = pars.compensate; p.radar_ccw = pars.radar_ccw; p.use_keyframe = pars.use_keyframe;
    p.min_keyframe_dist = pars.min_keyframe_dist_; p.min_keyframe_rot_deg = pars.min_keyframe_rot_deg_;
    p.estimate_cov_by_sampling = pars.estimate_cov_by_sampling;
    p.cov_sampling.xy_range = pars.cov_sampling_xy_range; p.cov_sampling.yaw_range = pars.cov_sampling_yaw_range;
    p.cov_sampling.samples_per_axis = (int32_t)pars.cov_sampling_samples_per_axis;
    p.cov_sampling.covariance_scaler = pars.cov_sampling_covariance_scaler;
    p.keep_nodes = 1;
    p.kstrong.k_strongest = 1;                                 // the cloud entry sizes its buffers by rows * k
    cfear_polar_desc d{pars.max_points, 1, 1, 1, (int64_t)pars.max_points};
    ctx_.check(cfear_odometry_create(ctx_.get(), 1, &d, &p, &od_));
  }
  bool updated = false;                                        // odometrykeyframefuser.h:198
  template <class Time>
  void pointcloudCallback(pcl::PointCloud<pcl::PointXYZI>::Ptr& cloud_filtered, pcl::PointCloud<pcl::PointXYZI>::Ptr& cloud_filtered_peaks,
                          Eigen::Affine3d& Tcurr, const Time& t) {                                            // :223
    Covariance cov;
    pointcloudCallback(cloud_filtered, cloud_filtered_peaks, Tcurr, t, cov);
  }
  template <class Time>
  void pointcloudCallback(pcl::PointCloud<pcl::PointXYZI>::Ptr& cloud_filtered, pcl::PointCloud<pcl::PointXYZI>::Ptr& cloud_filtered_peaks,
                          Eigen::Affine3d& Tcurr, const Time& /*t*/, Covariance& cov_curr) {                  // :225
    const PointCloud c = FromPcl(*cloud_filtered), pk = FromPcl(*cloud_filtered_peaks);
    const std::vector<cfear_frame_info> info = pointcloudCallback(std::vector<const PointCloud*>{&c}, std::vector<const PointCloud*>{&pk});
    Tcurr = Pose2dToAffine3d(Pose2d{info[0].pose[0], info[0].pose[1], info[0].pose[2]});
    updated = info[0].keyframe_added != 0;
    double cov[36];
    ctx_.check(cfear_odometry_get_covariance(od_, cov, nullptr));
    for (int r = 0; r < 6; r++) for (int q = 0; q < 6; q++) cov_curr(r, q) = cov[r * 6 + q];
  }
#endif
  // graph_ of a stream (odometrykeyframefuser.h:197-249: AddToGraph :428-445, SaveGraph, GetLastNode): call AddToGraph
  // after a frame whose info.keyframe_added is set (the odometry must have been created with par.keep_nodes); the node
  // keeps the pose, the compensated clouds, the surface points and the odometry constraint to the keyframe before it.
  struct GraphNode {
    Pose2d T;
    unsigned idx = 0;
    uint64_t stamp = 0;
    PointCloud cloud_peaks, cloud_nopeaks;
    std::vector<cfear_cell> cells;
    bool has_constraint = false;
    cfear_graph_constraint constraint{};
    bool has_Tgt = false;
    Pose2d Tgt{0, 0, 0};
  };
  void AddToGraph(int stream, const cfear_frame_info& info, uint64_t stamp = 0) {
    if ((int)graph_.size() < n_) graph_.resize((size_t)n_);
    GraphNode nd;
    nd.T = Pose2d{info.pose[0], info.pose[1], info.pose[2]};
    nd.idx = (unsigned)graph_[stream].size();
    nd.stamp = stamp;
    nd.cloud_nopeaks = GetCloud(stream, false);
    nd.cloud_peaks = GetCloud(stream, true);
    cfear_scan* s = GetScan(stream);
    nd.cells.resize((size_t)std::max(cfear_scan_size(s), 0));
    const int got = nd.cells.empty() ? 0 : cfear_scan_get_cells(s, nd.cells.data(), (int32_t)nd.cells.size());   // returns the count
    cfear_scan_destroy(s);
    if (got < 0) ctx_.check(got);
    nd.has_constraint = cfear_odometry_get_constraint(od_, stream, &nd.constraint) == CFEAR_OK;   // none for the first keyframe
    if ((int)status_.size() < n_) status_.resize((size_t)n_);
    if (!graph_[stream].empty()) {                             // a fused keyframe behind the first one
      const GraphNode& last = graph_[stream].back();
      status_[(size_t)stream].distance_traveled += std::hypot(nd.T.x - last.T.x, nd.T.y - last.T.y);
      status_[(size_t)stream].frame_nr++;
    }
    graph_[stream].push_back(std::move(nd));
  }
  // AddGroundTruth (odometrykeyframefuser.cpp:446-463): a node whose stamp appears in gt_vek receives Tgt / has_Tgt_
  void AddGroundTruth(const std::vector<std::pair<uint64_t, Pose2d>>& gt_vek, int stream = 0) {
    std::map<uint64_t, Pose2d> stamp_map;
    for (const auto& gt : gt_vek) stamp_map[gt.first] = gt.second;
    if ((size_t)stream >= graph_.size()) return;
    for (GraphNode& nd : graph_[(size_t)stream]) {
      const auto it = stamp_map.find(nd.stamp);
      if (it != stamp_map.end()) { nd.Tgt = it->second; nd.has_Tgt = true; }
    }
  }
#ifdef CFEAR_HIP_HAVE_EIGEN_PCL
  // the reference's argument (odometrykeyframefuser.h:240): poseStampedVector = elements with .pose (Eigen::Affine3d) and .t (ros::Time)
  template <class PoseStampedVector>
  auto AddGroundTruth(PoseStampedVector& gt_vek) -> decltype(gt_vek.begin()->t.toNSec(), void()) {
    std::vector<std::pair<uint64_t, Pose2d>> v;
    for (auto&& gt : gt_vek) v.emplace_back((uint64_t)gt.t.toNSec(), Affine3dToPose2d(gt.pose));
    AddGroundTruth(v, 0);
  }
#endif
  // GetStatus (odometrykeyframefuser.h:234): distance_traveled grows by |Tkeydiff.translation()| and frame_nr_ by one with
  // every fused keyframe after the first (odometrykeyframefuser.cpp:236, 243)
  std::string GetStatus(int stream = 0) const {
    const bool have = (size_t)stream < status_.size();
    return "Distance traveled: " + std::to_string(have ? status_[(size_t)stream].distance_traveled : 0.0) +
           ", nr sensor readings: " + std::to_string(have ? status_[(size_t)stream].frame_nr : 0u);
  }
  const GraphNode& GetLastNode(int stream = 0) const { return graph_.at((size_t)stream).back(); }
  size_t GraphSize(int stream = 0) const { return (size_t)stream < graph_.size() ? graph_[stream].size() : 0; }
  void SaveGraph(const std::string& path, int stream = 0, float radius = 3.0f, bool weight_intensity = true) const {  // SaveSimpleGraph, types.cpp:103-113
    const std::vector<GraphNode>& g = graph_.at((size_t)stream);
    std::vector<cfear_graph_node> nodes(g.size());
    for (size_t i = 0; i < g.size(); i++) {
      cfear_graph_node& n = nodes[i];
      n = cfear_graph_node{};
      const double xyt[3] = {g[i].T.x, g[i].T.y, g[i].T.theta};
      cfear_pose3d_from_xyt(xyt, &n.T);
      n.idx = g[i].idx; n.stamp = g[i].stamp;
      if (g[i].has_Tgt) { const double gxyt[3] = {g[i].Tgt.x, g[i].Tgt.y, g[i].Tgt.theta}; cfear_pose3d_from_xyt(gxyt, &n.Tgt); n.has_Tgt = 1; }
      for (int k = 0; k < 16; k++) n.motion[k] = (k % 5 == 0) ? 1.0 : 0.0;            // motion_ = Identity
      n.cloud_peaks = cfear_graph_cloud{g[i].cloud_peaks.empty() ? nullptr : &g[i].cloud_peaks[0].x, (int32_t)g[i].cloud_peaks.size(), 0, g[i].stamp, nullptr};
      n.cloud_nopeaks = cfear_graph_cloud{g[i].cloud_nopeaks.empty() ? nullptr : &g[i].cloud_nopeaks[0].x, (int32_t)g[i].cloud_nopeaks.size(), 0, g[i].stamp, nullptr};
      n.has_normal = 1; n.input_is_nopeaks = 1;
      n.cells = g[i].cells.data(); n.n_cells = (int32_t)g[i].cells.size();
      n.radius = radius; n.weight_intensity = weight_intensity ? 1 : 0;
      n.constraints = g[i].has_constraint ? &g[i].constraint : nullptr;
      n.n_constraints = g[i].has_constraint ? 1 : 0;
    }
    const int rc = cfear_graph_save(path.c_str(), nodes.data(), (int32_t)nodes.size());
    if (rc != CFEAR_OK) throw CfearError(rc, cfear_status_string(rc));
  }
  // scan_ of a stream's last frame (odometrykeyframefuser.cpp:172, 244): surface points + the two clouds
  cfear_scan* GetScan(int stream) { cfear_scan* s = nullptr; ctx_.check(cfear_odometry_get_scan(od_, stream, &s)); return s; }
  PointCloud GetCloud(int stream, bool peaks) {
    int32_t n = 0;
    ctx_.check((peaks ? cfear_odometry_get_peaks : cfear_odometry_get_cloud)(od_, stream, nullptr, 0, &n));
    PointCloud c((size_t)n);
    if (n) ctx_.check((peaks ? cfear_odometry_get_peaks : cfear_odometry_get_cloud)(od_, stream, &c[0].x, n, &n));
    return c;
  }
 private:
  Context& ctx_;
  cfear_odometry* od_ = nullptr;
  int n_;
  std::vector<std::vector<GraphNode>> graph_;
  struct Status { double distance_traveled = 0.0; unsigned frame_nr = 0; };
  std::vector<Status> status_;
};

}  // namespace CFEAR_Radarodometry

// CorAlRadarQuality (coral_alignment_quality AlignmentQuality.cpp:99-230) over two peak clouds.
namespace CorAlignment {
class CorAlRadarQuality {
 public:
  CorAlRadarQuality(CFEAR_Radarodometry::Context& ctx, const CFEAR_Radarodometry::PointCloud& ref,
                    const CFEAR_Radarodometry::Pose2d& ref_pose, const CFEAR_Radarodometry::PointCloud& src,
                    const CFEAR_Radarodometry::Pose2d& src_pose,
                    const CFEAR_Radarodometry::Pose2d& Toffset = CFEAR_Radarodometry::Pose2d{0, 0, 0}, double radius = 1.0,
                    bool weight_res_intensity = false) {
    cfear_coral_job j{};
    j.ref_xyzi = &ref[0].x; j.src_xyzi = &src[0].x;
    j.n_ref = (int32_t)ref.size(); j.n_src = (int32_t)src.size();
    j.ref_pose[0] = ref_pose.x; j.ref_pose[1] = ref_pose.y; j.ref_pose[2] = ref_pose.theta;
    j.src_pose[0] = src_pose.x; j.src_pose[1] = src_pose.y; j.src_pose[2] = src_pose.theta;
    j.offset[0] = Toffset.x; j.offset[1] = Toffset.y; j.offset[2] = Toffset.theta;
    cfear_coral_params p;
    cfear_coral_params_default(&p);
    p.radius = radius; p.weight_res_intensity = weight_res_intensity ? 1 : 0;
    ctx.check(cfear_coral_quality(ctx.get(), &j, &p, &res_, nullptr));
    valid_ = res_.valid != 0;
  }
  std::vector<double> GetQualityMeasure() const { return {res_.joint, res_.sep, res_.overlap}; }
  bool valid_ = false;
 private:
  cfear_coral_result res_{};
};
}  // namespace CorAlignment

// RSCManager (place_recognition_radar/RadarScancontext.h): descriptor database + candidate retrieval.
class RSCManager {
 public:
  explicit RSCManager(CFEAR_Radarodometry::Context& ctx, const cfear_sc_manager_params* par = nullptr) : ctx_(ctx) {
    cfear_sc_manager_params def;
    if (!par) { cfear_sc_manager_params_default(&def); par = &def; }
    n_candidates_ = par->n_candidates;
    ctx_.check(cfear_sc_manager_create(ctx_.get(), par, &m_));
  }
  ~RSCManager() { cfear_sc_manager_destroy(m_); }
  RSCManager(const RSCManager&) = delete;
  RSCManager& operator=(const RSCManager&) = delete;
  void makeAndSaveScancontextAndKeysRadarCloud(const CFEAR_Radarodometry::PointCloud& cloud,          // RadarScancontext.cpp:156-180
                                               const CFEAR_Radarodometry::Pose2d& Todom) {
    const double T[3] = {Todom.x, Todom.y, Todom.theta};
    ctx_.check(cfear_sc_manager_add(m_, cloud.empty() ? nullptr : &cloud[0].x, (int32_t)cloud.size(), T));
  }
  std::vector<cfear_sc_candidate> detectLoopClosureID() {                                             // :286-345
    std::vector<cfear_sc_candidate> out((size_t)std::max(n_candidates_, 1));
    int32_t n = 0;
    ctx_.check(cfear_sc_manager_detect(m_, out.data(), (int32_t)out.size(), &n));
    out.resize((size_t)n);
    return out;
  }
 private:
  CFEAR_Radarodometry::Context& ctx_;
  cfear_sc_manager* m_ = nullptr;
  int n_candidates_ = 3;
};

// Loop-candidate verification of tbv_slam::loopclosure (tbv_slam/src/tbv_slam/loopclosure.cpp:320-384, 261-274,
// 776-808) for a batch of candidates.
namespace tbv_slam {
struct LoopCandidate {
  const CFEAR_Radarodometry::MapPointNormal* from;         // (*graph_)[from].cloud_normal_
  const CFEAR_Radarodometry::MapPointNormal* to;
  const CFEAR_Radarodometry::PointCloud* from_peaks;       // (*graph_)[from].cloud_peaks_
  const CFEAR_Radarodometry::PointCloud* to_peaks;
  CFEAR_Radarodometry::Pose2d Tfrom;                       // (*graph_)[from].GetPose()
  CFEAR_Radarodometry::Pose2d t_be_guess;                  // constraint.t_be on entry
  double sc_sim, odom_bounds;                              // quality["sc-sim"], quality["odom-bounds"]
  int query;                                               // candidates of one query node are selected together
};
inline double VerifyByOdometry(const std::vector<CFEAR_Radarodometry::Pose2d>& relative_motions, double odom_sigma_error = 0.03,
                               bool verify_via_odometry = true) {
  double sim = 0.0;
  static_assert(sizeof(CFEAR_Radarodometry::Pose2d) == 3 * sizeof(double), "Pose2d must be three packed doubles");
  const int rc = cfear_verify_by_odometry(relative_motions.empty() ? nullptr : &relative_motions[0].x,
                                          (int32_t)relative_motions.size(), odom_sigma_error, verify_via_odometry ? 1 : 0, &sim);
  if (rc != CFEAR_OK) throw CFEAR_Radarodometry::CfearError(rc, cfear_status_string(rc));
  return sim;
}
inline std::vector<cfear_verify_result> VerifyLoopCandidates(CFEAR_Radarodometry::Context& ctx,
                                                              const std::vector<LoopCandidate>& cands,
                                                              const cfear_verify_params* par = nullptr) {
  cfear_verify_params def;
  if (!par) { cfear_verify_params_default(&def); par = &def; }
  std::vector<cfear_verify_job> jobs(cands.size());
  for (size_t i = 0; i < cands.size(); i++) {
    const LoopCandidate& c = cands[i];
    cfear_verify_job& j = jobs[i];
    j = cfear_verify_job{};
    j.from_scan = c.from->device(); j.to_scan = c.to->device();
    j.from_peaks = c.from_peaks->empty() ? nullptr : &(*c.from_peaks)[0].x; j.n_from = (int32_t)c.from_peaks->size();
    j.to_peaks = c.to_peaks->empty() ? nullptr : &(*c.to_peaks)[0].x; j.n_to = (int32_t)c.to_peaks->size();
    j.from_pose[0] = c.Tfrom.x; j.from_pose[1] = c.Tfrom.y; j.from_pose[2] = c.Tfrom.theta;
    j.t_be_guess[0] = c.t_be_guess.x; j.t_be_guess[1] = c.t_be_guess.y; j.t_be_guess[2] = c.t_be_guess.theta;
    j.sc_sim = c.sc_sim; j.odom_bounds = c.odom_bounds; j.group = c.query;
  }
  std::vector<cfear_verify_result> res(cands.size());
  ctx.check(cfear_verify_loop_candidates(ctx.get(), jobs.data(), (int32_t)jobs.size(), par, res.data()));
  return res;
}
}  // namespace tbv_slam
