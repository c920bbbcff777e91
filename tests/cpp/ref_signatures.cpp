// The reference's call sites, verbatim in shape, against include/cfear_hip.hpp's reference-signature block:
//   loopclosure::Register          tbv_slam/src/tbv_slam/loopclosure.cpp:35-97 (n_scan_normal_reg(Str2Cost("P2L")), SetParameters(4,10),
//                                  Register(scans, T, cov, false), GetCost)
//   OdometryKeyframeFuser          offline_odometry.cpp:103-108 (pointcloudCallback(cloud, peaks, Tcurrent, t))
// argv[1]: a raw float32 file with two clouds [n0][4] [n1][4] preceded by two int32 counts.  Prints
//   n_cells0 n_cells1 ok x y theta cov00 cov55 cost_ok score n_res  fuser_x fuser_y fuser_theta updated
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "cfear_hip.hpp"

#ifndef CFEAR_HIP_HAVE_EIGEN_PCL
#error "the reference-signature block was not enabled"
#endif

using namespace CFEAR_Radarodometry;

int main(int argc, char** argv) {
  if (argc < 2) return 2;
  FILE* f = fopen(argv[1], "rb");
  if (!f) return 3;
  int32_t n[2];
  if (fread(n, 4, 2, f) != 2) return 3;
  pcl::PointCloud<pcl::PointXYZI>::Ptr cloud[2];
  for (int k = 0; k < 2; k++) {
    cloud[k] = pcl::PointCloud<pcl::PointXYZI>::Ptr(new pcl::PointCloud<pcl::PointXYZI>());
    std::vector<float> buf((size_t)n[k] * 4);
    if (fread(buf.data(), 4, buf.size(), f) != buf.size()) return 3;
    for (int i = 0; i < n[k]; i++) { pcl::PointXYZI p; p.x = buf[4 * i]; p.y = buf[4 * i + 1]; p.z = buf[4 * i + 2]; p.intensity = buf[4 * i + 3]; cloud[k]->push_back(p); }
  }
  fclose(f);
  try {
    MapNormalPtr m0(new MapPointNormal(cloud[0], 3.0f, Eigen::Vector2d(0, 0), true, false));
    MapNormalPtr m1(new MapPointNormal(cloud[1], 3.0f, Eigen::Vector2d(0, 0), true));
    // loopclosure::Register
    std::vector<Matrix6d> cov_vek = {Matrix6d::Zero(), Matrix6d::Zero()};
    std::vector<MapNormalPtr> scans_vek = {m0, m1};
    std::vector<Eigen::Affine3d> T_vek = {Eigen::Affine3d::Identity(), Pose2dToAffine3d(Pose2d{2.0, 0.0, 0.0})};
    n_scan_normal_reg radar_reg(P2L);
    radar_reg.SetParameters(4, 10);
    const bool success = radar_reg.Register(scans_vek, T_vek, cov_vek, false);
    const Pose2d out = Affine3dToPose2d(T_vek.back());
    double score = 0;
    std::vector<double> residuals;
    n_scan_normal_reg quality(P2L, Huber, 0.3);
    const bool cost_ok = quality.GetCost(scans_vek, T_vek, score, residuals);
    printf("%zu %zu %d %.17g %.17g %.17g %.17g %.17g %d %.17g %zu ", m0->GetSize(), m1->GetSize(), (int)success, out.x, out.y, out.theta,
           cov_vek[1](0, 0), cov_vek[1](5, 5), (int)cost_ok, score, residuals.size());
    const Eigen::Vector2d u = m0->GetMean2d(0);
    (void)u; (void)m0->GetNormal2d(0); (void)m0->GetCov2d(0);
    // the odometry node
    OdometryKeyframeFuser::Parameters pars;
    pars.cost_type = "P2P"; pars.res = 3.0; pars.submap_scan_size = 4; pars.weight_intensity_ = true; pars.weight_opt = Combined_weights;
    OdometryKeyframeFuser fuser(pars, true);
    Eigen::Affine3d Tcurrent;
    pcl::PointCloud<pcl::PointXYZI>::Ptr peaks(new pcl::PointCloud<pcl::PointXYZI>());
    fuser.pointcloudCallback(cloud[0], peaks, Tcurrent, 0.0);
    Covariance cov;
    fuser.pointcloudCallback(cloud[1], peaks, Tcurrent, 0.25, cov);
    const Pose2d fp = Affine3dToPose2d(Tcurrent);
    printf("%.17g %.17g %.17g %d\n", fp.x, fp.y, fp.theta, (int)fuser.updated);
    // ---- the remaining symbols of SURVEY 8(b), in the reference's own argument types -----------------------------------
    // TransformMap / GetCell (pointnormal.h:120, 168)
    MapNormalPtr moved = m0->TransformMap(Pose2dToAffine3d(Pose2d{1.0, 2.0, 0.25}));
    if (moved->GetSize() != m0->GetSize() || !(moved->GetCell(0).nsamples == m0->GetCell(0).nsamples)) return 4;
    if (m0->GetScan() != cloud[0]) return 4;                                   // GetScan (pointnormal.h:170)
    MapPointNormal::PublishMap("/current_normals", m1, T_vek.back(), "world", 1);   // a no-op here (RViz markers)
    // soft_constraints = true is refused loudly, never ignored (n_scan_normal.cpp:371-375 is undefined behaviour there)
    bool refused = false;
    try { radar_reg.Register(scans_vek, T_vek, cov_vek, true); } catch (const CfearError& e) { refused = e.status == CFEAR_ERR_INVALID_ARGUMENT; }
    if (!refused) return 5;
    // AddGroundTruth(poseStampedVector&) (odometrykeyframefuser.h:240) + GetStatus (:234)
    struct Stamp { uint64_t ns; uint64_t toNSec() const { return ns; } };
    struct poseStamped { Eigen::Affine3d pose; Stamp t; };
    std::vector<poseStamped> gt_vek;
    gt_vek.push_back(poseStamped{Pose2dToAffine3d(Pose2d{1, 2, 0.5}), Stamp{0}});
    fuser.AddGroundTruth(gt_vek);
    if (fuser.GetStatus().find("Distance traveled: ") != 0) return 6;
    // statistics timing / ToMs (statistics.h:38-40), as radar_driver.cpp:87 documents its stage time
    struct Duration { int64_t ns; int64_t toNSec() const { return ns; } };
    timing.Document("Filtering", ToMs(Duration{2500000}));
    if (timing.GetStatistics().find("Filtering avg, 2.500000") != 0) return 7;
#ifdef CFEAR_HIP_HAVE_CV_BRIDGE
    // StructuredKStrongest(cv_bridge image, z_min, k, min_distance, range_res) + getPeaksFilteredPointCloud(cloud, peaks)
    // (radar_driver.cpp:57-60) and k_strongest_filter (coral_alignment_quality ScanType.cpp:104-114)
    cv_bridge::CvImagePtr cv_polar_image(new cv_bridge::CvImage());
    cv_polar_image->image = cv::Mat(8, 256);
    for (int r = 0; r < 8; r++) for (int c = 0; c < 256; c++) cv_polar_image->image.data[r * 256 + c] = (unsigned char)((r * 37 + c * 11) % 251);
    StructuredKStrongest filt(cv_polar_image, 60, 12, 2.5, 0.0438);
    pcl::PointCloud<pcl::PointXYZI>::Ptr cloud_filtered, cloud_filtered_peaks;     // null: the filter allocates, like the reference
    filt.getPeaksFilteredPointCloud(cloud_filtered, false);
    filt.getPeaksFilteredPointCloud(cloud_filtered_peaks, true);
    pcl::PointCloud<pcl::PointXYZI>::Ptr legacy;
    k_strongest_filter(cv_polar_image, legacy, 12, 60, 0.0438, 2.5);
    if (!cloud_filtered || cloud_filtered->size() == 0 || cloud_filtered_peaks->size() > cloud_filtered->size() || !legacy) return 8;
#else
#error "the cv_bridge stand-in was not found"
#endif
  } catch (const CfearError& e) {
    fprintf(stderr, "%s\n", e.what());
    return 1;
  }
  return 0;
}
