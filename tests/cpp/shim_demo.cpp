// A C++ caller of libcfear_hip.so through the header-only mirror of the reference classes
// (include/cfear_hip.hpp).  Reads a raw uint8 polar image pair [2][rows][cols] from argv[1], registers
// frame 1 against frame 0 (P2L, loop-closure settings 4 x 10) and prints the result as one line:
//   n_points0 n_points1 n_cells0 n_cells1 ok x y theta score cov_ok cov_xx cov_yy cov_tt coral_valid joint sep overlap
//   reg_ok t_be.x t_be.y t_be.theta alignment_quality probability accepted odom_bounds
//   fuser.x fuser.y fuser.theta fuser.n_cells node_cells node_peaks
// (cov_*: covariance by cost sampling with the loop-closure constants, loopclosure.cpp:108-112; the last eight: the
// pair verified as a loop candidate, frame 1 = query, frame 0 = candidate, through tbv_slam::VerifyLoopCandidates).
// With a fifth argument "bins-major" the file holds [2][cols][rows] images as a non-Oxford driver publishes them.
#include <cmath>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <memory>
#include <string>
#include <vector>

#include "cfear_hip.hpp"

using namespace CFEAR_Radarodometry;

int main(int argc, char** argv) {
  {  // the small boundary symbols external packages use (radar_driver.h:24-28, 65-82)
    using namespace CFEAR_Radarodometry;
    if (Str2filter("CA-CFAR") != CACFAR || Str2filter("kstrong") != kstrong || Str2filter("anything") != kstrong) return 91;
    if (Filter2str(CACFAR) != "CA-CFAR" || Filter2str(kstrong) != "kstrong") return 92;
    radarDriver::Parameters rp;
    rp.filter_type_ = Str2filter("CA-CFAR");
    const std::string txt = rp.ToString();
    if (txt.find("filter type, CA-CFAR\n") == std::string::npos || txt.find("range res, 0.0438\n") != 0) return 93;
  }

  if (argc < 4) return 2;
  const int rows = atoi(argv[2]), cols = atoi(argv[3]);
  std::vector<uint8_t> img((size_t)2 * rows * cols);
  FILE* f = fopen(argv[1], "rb");
  if (!f || fread(img.data(), 1, img.size(), f) != img.size()) return 3;
  fclose(f);
  try {
    Context ctx(0);
    radarDriver::Parameters p;
    p.k_strongest = 40;
    const bool bins_major = argc > 4 && std::string(argv[4]) == "bins-major";
    if (bins_major) p.dataset = "mulran";
    radarDriver driver(ctx, p);
    PointCloud c0, c1;
    PointCloud pk0, pk1;
    if (bins_major) {                                       // the file's images are cols x rows
      driver.CallbackOffline(img.data(), cols, rows, rows, c0, pk0);
      driver.CallbackOffline(img.data() + (size_t)rows * cols, cols, rows, rows, c1, pk1);
    } else {
      driver.CallbackOffline(img.data(), rows, cols, cols, c0, pk0);
      driver.CallbackOffline(img.data() + (size_t)rows * cols, rows, cols, cols, c1, pk1);
    }
    MapPointNormal m0(ctx, c0, 3.0f, 0, 0, true), m1(ctx, c1, 3.0f, 0, 0, true);
    n_scan_normal_reg reg(ctx, CFEAR_P2L);
    reg.SetParameters(4, 10);
    std::vector<Pose2d> T = {{0, 0, 0}, {2.0, 0.0, 0.0}};
    const bool ok = reg.Register({&m0, &m1}, T);
    {
      // the same candidate through the sharded entry a C++ loop-closure thread would call on each rank (world 1 here):
      // it must reproduce the registration above
      const cfear_scan* hs[2] = {m0.device(), m1.device()};
      const double guess[6] = {0, 0, 0, 2.0, 0.0, 0.0};
      cfear_reg_job job{hs, 2, 0, guess};
      cfear_reg_params par;
      cfear_reg_params_default(&par);
      par.max_itr_association = 4; par.max_itr_solver = 10;
      cfear_reg_result rr{};
      ctx.check(cfear_register_batch_sharded(ctx.get(), &job, 1, &par, 0, 1, nullptr, nullptr, &rr));
      if (rr.pose[0] != T[1].x || rr.pose[1] != T[1].y || rr.pose[2] != T[1].theta) throw CfearError(-1, "sharded entry disagrees");
      // ... and as a candidate pair among the scans of a device-resident table (cfear_register_candidates)
      ScanTable table(ctx, {&m0, &m1});
      n_scan_normal_reg creg(ctx, CFEAR_P2L);
      creg.SetParameters(4, 10);
      std::vector<cfear_reg_result> cres;
      creg.RegisterCandidates(table, {cfear_candidate{0, 1, {0, 0, 0}, {2.0, 0.0, 0.0}}}, cres);
      if (table.size() != 2 || cres[0].pose[0] != T[1].x || cres[0].pose[1] != T[1].y || cres[0].pose[2] != T[1].theta)
        throw CfearError(-1, "candidate table disagrees");
      // ... and through the pipe that keeps sharded steps in flight (cfear_candidate_pipe; one rank, no communicator here): two
      // steps submitted before the first is collected, byte-identical records
      CandidatePipe pipe(ctx, table, 4);
      const std::vector<cfear_candidate> batch = {cfear_candidate{0, 1, {0, 0, 0}, {2.0, 0.0, 0.0}}, cfear_candidate{0, 1, {0, 0, 0}, {2.1, -0.1, 0.01}}};
      const int64_t t0 = pipe.submit(batch, creg), t1 = pipe.submit(batch, creg);
      std::vector<cfear_reg_result> p0, p1, direct;
      pipe.collect(t0, p0);
      pipe.collect(t1, p1);
      creg.RegisterCandidates(table, batch, direct);
      if (p0.size() != 2 || memcmp(p0.data(), direct.data(), 2 * sizeof(cfear_reg_result)) != 0 || memcmp(p1.data(), direct.data(), 2 * sizeof(cfear_reg_result)) != 0)
        throw CfearError(-1, "candidate pipe disagrees");
    }
    double cov[36];
    cfear_cov_sampling_params sp;
    cfear_cov_sampling_params_default(&sp);
    sp.xy_range = 0.4; sp.yaw_range = 0.0044;
    const bool cov_ok = reg.approximateCovarianceBySampling({&m0, &m1}, T, cov, &sp);
    // loop-closure verification feature: CorAl quality of the two peak clouds at the registered pose
    CorAlignment::CorAlRadarQuality coral(ctx, pk0, T[0], pk1, T[1]);
    const std::vector<double> q = coral.GetQualityMeasure();
    printf("%zu %zu %zu %zu %d %.12g %.12g %.12g %.12g %d %.12g %.12g %.12g %d %.12g %.12g %.12g", c0.size(), c1.size(),
           m0.GetSize(), m1.GetSize(), ok ? 1 : 0, T[1].x, T[1].y, T[1].theta, reg.getScore(), cov_ok ? 1 : 0, cov[0], cov[7],
           cov[35], coral.valid_ ? 1 : 0, q[0], q[1], q[2]);
    // the same pair as a loop-closure candidate: query = frame 1 at pose (2.2, 0.1, 0.01), guess for frame 0 relative to it
    const double odom_bounds = tbv_slam::VerifyByOdometry({{1.0, 0.0, 0.0}, {1.2, 0.1, 0.01}});
    tbv_slam::LoopCandidate cand{&m1, &m0, &pk1, &pk0, {2.2, 0.1, 0.01}, {-2.0, 0.2, -0.02}, 0.15, odom_bounds, 0};
    const std::vector<cfear_verify_result> vr = tbv_slam::VerifyLoopCandidates(ctx, {cand});
    printf(" %d %.12g %.12g %.12g %.12g %.12g %d %.12g", vr[0].reg_ok, vr[0].t_be[0], vr[0].t_be[1], vr[0].t_be[2],
           vr[0].alignment_quality, vr[0].probability, vr[0].accepted, odom_bounds);
    // the two sweeps as one sequence through the batched pipeline (1 stream), collecting the second frame's node
    cfear_odometry_params op;
    cfear_odometry_params_default(&op);
    op.keep_nodes = 1;
    op.rotate_ccw = bins_major ? 1 : 0;
    OdometryKeyframeFuser fuser(ctx, 1, bins_major ? cols : rows, bins_major ? rows : cols, &op);
    const std::vector<cfear_frame_info> fi0 = fuser.processFrame(img.data());
    if (fi0[0].keyframe_added) fuser.AddToGraph(0, fi0[0], 1000);
    const std::vector<cfear_frame_info> fi = fuser.processFrame(img.data() + (size_t)rows * cols);
    if (fi[0].keyframe_added) fuser.AddToGraph(0, fi[0], 2000);
    const PointCloud node_peaks = fuser.GetCloud(0, true);
    cfear_scan* node_scan = fuser.GetScan(0);
    printf(" %.12g %.12g %.12g %d %d %zu", fi[0].pose[0], fi[0].pose[1], fi[0].pose[2], fi[0].n_cells,
           cfear_scan_size(node_scan), node_peaks.size());
    cfear_scan_destroy(node_scan);
    // graph_ of the stream written as simple_graph.sgh and read back (SaveGraph / LoadSimpleGraph)
    const std::string sgh = std::string(argv[1]) + ".sgh";
    fuser.SaveGraph(sgh);
    cfear_graph* g = nullptr;
    ctx.check(cfear_graph_load(sgh.c_str(), &g));
    cfear_graph_node last;
    ctx.check(cfear_graph_node_at(g, cfear_graph_size(g) - 1, &last));
    double lxyt[3];
    cfear_pose3d_to_xyt(&last.T, lxyt);
    printf(" %zu %d %d %d %d %.12g %.12g %.12g %llu", fuser.GraphSize(), cfear_graph_size(g), last.n_cells, last.cloud_peaks.n,
           last.n_constraints, lxyt[0], lxyt[1], lxyt[2], (unsigned long long)fuser.GetLastNode().stamp);
    cfear_graph_destroy(g);
    // ---- the smaller boundary symbols: StructuredKStrongest, k_strongest_filter, GetCell / TransformMap,
    //      AddGroundTruth / GetStatus, timing / ToMs -------------------------------------------------------------------
    int small_ok = 1;
    if (!bins_major) {
      StructuredKStrongest filt(ctx, img.data(), rows, cols, cols, 60, 40, 2.5, 0.0438);
      PointCloud all, peaks;
      filt.getPeaksFilteredPointCloud(all, false);
      filt.getPeaksFilteredPointCloud(peaks, true);
      filt.getPeaksFilteredPointCloud(peaks, true);            // appends, like the reference's push_back loop
      small_ok &= all.size() == c0.size() && peaks.size() == 2 * pk0.size();
      for (size_t i = 0; i < all.size() && small_ok; i++) small_ok &= all[i].x == c0[i].x && all[i].y == c0[i].y && all[i].intensity == c0[i].intensity;
      for (size_t i = 0; i < pk0.size() && small_ok; i++) small_ok &= peaks[i].x == pk0[i].x && peaks[pk0.size() + i].y == pk0[i].y;
    }
    PointCloud legacy;
    if (!bins_major) k_strongest_filter(ctx, img.data(), rows, cols, cols, legacy, 12, 60.0, 0.0438, 2.5);
    const cfear_cell& cell0 = m0.GetCell(0);
    const Pose2d Tm{1.5, -0.5, 0.3};
    const std::unique_ptr<MapPointNormal> moved = m0.TransformMap(Tm);
    const cfear_cell& mc0 = moved->GetCell(0);
    const double ex = std::cos(0.3) * cell0.mean[0] - std::sin(0.3) * cell0.mean[1] + 1.5, ey = std::sin(0.3) * cell0.mean[0] + std::cos(0.3) * cell0.mean[1] - 0.5;
    small_ok &= moved->GetSize() == m0.GetSize() && std::fabs(mc0.mean[0] - ex) < 1e-12 && std::fabs(mc0.mean[1] - ey) < 1e-12;
    const std::vector<int> near = moved->GetClosestIdx(ex, ey, 0.5);
    small_ok &= near.size() == 1 && near[0] == 0;
    fuser.AddGroundTruth({{2000, Pose2d{2.5, 0.25, 0.125}}, {777, Pose2d{9, 9, 9}}});
    small_ok &= fuser.GetLastNode().has_Tgt && fuser.GetLastNode().Tgt.x == 2.5 && !fuser.GetStatus().empty();
    timing.Document("Filtering", 1.0); timing.Document("Filtering", 3.0);
    small_ok &= timing.GetStatistics().find("Filtering avg, 2.000000\nFiltering dev [") == 0 && ToMs(0.25) == 250.0;
    printf(" %d %zu %s\n", small_ok, legacy.size(), fuser.GetStatus().c_str());
  } catch (const CfearError& e) {
    fprintf(stderr, "cfear error %d: %s\n", e.status, e.what());
    return 1;
  }
  return 0;
}
