// ref_golden.cpp -- runs the REFERENCE's CFEAR hot path (catkin library cfear_radarodometry, built in the reference's own
// image: tbv_slam/docker/Dockerfile) on the inputs tools/ref_golden/export_inputs.py wrote, and leaves the outputs as raw
// arrays + manifest.txt for pack_outputs.py.  Not buildable in this repository's image (no ROS / PCL / Eigen / Ceres /
// OpenCV / Boost); never run there.  Calls: radar_filters.h:84-113, cfar.h:27-42, utils.h:49, pointnormal.h:110-243,
// n_scan_normal.h:27-85, odometrykeyframefuser.h:197-249, types.h:93-194.   usage: rosrun ref_golden ref_golden IN_DIR OUT_DIR   (roscore must be up)
#include <cv_bridge/cv_bridge.h>
#include <ros/ros.h>

#include <fstream>
#include <map>
#include <sstream>

#include "cfear_radarodometry/cfar.h"
#include "cfear_radarodometry/n_scan_normal.h"
#include "cfear_radarodometry/odometrykeyframefuser.h"
#include "cfear_radarodometry/pointnormal.h"
#include "cfear_radarodometry/radar_filters.h"
#include "cfear_radarodometry/types.h"
#include "cfear_radarodometry/utils.h"

using namespace CFEAR_Radarodometry;
typedef pcl::PointCloud<pcl::PointXYZI> Cloud;

struct Arr { std::string dtype; std::vector<long> dims; std::vector<char> bytes; };
static std::map<std::string, Arr> load_dir(const std::string& dir) {
  std::map<std::string, Arr> out;
  std::ifstream m(dir + "/manifest.txt");
  std::string line;
  while (std::getline(m, line)) {
    std::istringstream ss(line);
    std::string name;
    Arr a;
    ss >> name >> a.dtype;
    for (long d; ss >> d;) a.dims.push_back(d);
    std::ifstream f(dir + "/" + name + ".bin", std::ios::binary);
    a.bytes.assign(std::istreambuf_iterator<char>(f), std::istreambuf_iterator<char>());
    out[name] = a;
  }
  return out;
}
struct Writer {
  std::string dir;
  std::ofstream man;
  explicit Writer(const std::string& d) : dir(d), man(d + "/manifest.txt") {}
  template <typename T>
  void put(const std::string& name, const char* dtype, const std::vector<T>& v, std::vector<long> dims) {
    std::ofstream(dir + "/" + name + ".bin", std::ios::binary).write((const char*)v.data(), v.size() * sizeof(T));
    man << name << " " << dtype;
    for (long d : dims) man << " " << d;
    man << "\n";
  }
};
static Cloud::Ptr to_cloud(const Arr& a) {                       // float32 [n][4]: x, y, z, intensity
  Cloud::Ptr c(new Cloud());
  const float* p = (const float*)a.bytes.data();
  for (long i = 0; i < a.dims[0]; i++) { pcl::PointXYZI q; q.x = p[4 * i]; q.y = p[4 * i + 1]; q.z = p[4 * i + 2]; q.intensity = p[4 * i + 3]; c->push_back(q); }
  return c;
}
static std::vector<float> flat(const Cloud& c) {
  std::vector<float> v;
  for (const auto& p : c.points) { v.push_back(p.x); v.push_back(p.y); v.push_back(p.z); v.push_back(p.intensity); }
  return v;
}
struct KStrongProbe : public StructuredKStrongest {              // the per-row lists are protected members
  using StructuredKStrongest::StructuredKStrongest;
  const std::vector<std::vector<intensity_range>>& lists() const { return dense_filtered_; }
};

int main(int argc, char** argv) {
  if (argc < 3) { fprintf(stderr, "usage: ref_golden IN_DIR OUT_DIR\n"); return 2; }
  ros::init(argc, argv, "ref_golden");                           // Registration's constructor creates a NodeHandle (registration.cpp:4-8)
  auto in = load_dir(argv[1]);
  Writer w(argv[2]);
  // ---- filters (tests/golden/make_golden.py: k = 12, z_min = 60, range_res 0.0438, min_distance 2.5; CFAR 20 / 5 / 0.01 / 40) ----
  {
    const Arr& im = in.at("img");
    const int rows = (int)im.dims[0], cols = (int)im.dims[1], k = 12;
    cv_bridge::CvImagePtr cv(new cv_bridge::CvImage());
    cv->encoding = "mono8";
    cv->image = cv::Mat(rows, cols, CV_8UC1, (void*)im.bytes.data()).clone();
    KStrongProbe f(cv, 60, k, 2.5, 0.0438);
    Cloud::Ptr c(new Cloud()), cp(new Cloud());
    f.getPeaksFilteredPointCloud(c, false);
    f.getPeaksFilteredPointCloud(cp, true);
    std::vector<int32_t> sr((size_t)rows * k, -1), cnt(rows, 0);
    std::vector<uint8_t> si((size_t)rows * k, 0);
    for (int r = 0; r < rows; r++) {
      const auto& l = f.lists()[r];
      cnt[r] = (int32_t)l.size();
      for (size_t j = 0; j < l.size() && j < (size_t)k; j++) { si[(size_t)r * k + j] = l[j].first; sr[(size_t)r * k + j] = l[j].second; }
    }
    w.put("f_sel_range", "int32", sr, {rows, k}); w.put("f_sel_intensity", "uint8", si, {rows, k}); w.put("f_sel_count", "int32", cnt, {rows});
    w.put("f_cloud", "float32", flat(*c), {(long)c->size(), 4}); w.put("f_cloud_peaks", "float32", flat(*cp), {(long)cp->size(), 4});
    AzimuthCACFAR cf(20, 0.01, 5, 0.0438, 40, 2.5, 400.0);        // radar_driver.cpp:54 passes 400.0
    Cloud::Ptr cc(new Cloud());
    cf.getFilteredPointCloud(cv, cc);
    w.put("f_cfar_cloud", "float32", flat(*cc), {(long)cc->size(), 4});
  }
  // ---- compensation, surface points, registration (make_golden.py: r = 3, weight_intensity, poses / motion from the fixture) ----
  std::vector<Cloud::Ptr> clouds = {to_cloud(in.at("cloud0")), to_cloud(in.at("cloud1")), to_cloud(in.at("cloud2"))};
  const double* mot = (const double*)in.at("mot").bytes.data();
  const double* poses = (const double*)in.at("poses").bytes.data();
  Compensate(*clouds[1], vectorToAffine3d(mot[0], mot[1], 0, 0, 0, mot[2]), false);
  w.put("r_comp1", "float32", flat(*clouds[1]), {(long)clouds[1]->size(), 4});
  std::vector<MapNormalPtr> scans;
  for (int i = 0; i < 3; i++) {
    scans.push_back(MapNormalPtr(new MapPointNormal(clouds[i], 3.0f, Eigen::Vector2d(0, 0), true, false)));
    std::vector<double> cells;                                   // per cell: mean[2] normal[2] cov[4] scale avg_intensity lambda_min lambda_max nsamples
    for (const cell& c : scans.back()->GetCells())
      for (double v : {c.u_(0), c.u_(1), c.snormal_(0), c.snormal_(1), c.cov_(0, 0), c.cov_(0, 1), c.cov_(1, 0), c.cov_(1, 1), c.scale_,
                       c.avg_intensity_, c.lambda_min, c.lambda_max, (double)c.Nsamples_}) cells.push_back(v);
    w.put("r_cells" + std::to_string(i), "float64", cells, {(long)cells.size() / 13, 13});
  }
  struct Case { const char* name; cost_metric cost; loss_type loss; weightoption opt; unsigned mo, mi; };
  const Case cases[] = {{"p2l_4x10", P2L, Huber, Uniform, 4, 10}, {"p2p_w4", P2P, Huber, Combined_weights, 8, 20},
                        {"p2d", P2D, Huber, Uniform, 8, 20}, {"p2l_cauchy", P2L, Cauchy, Combined_weights, 8, 20}};
  for (const Case& cs : cases) {
    auto T0 = [&]() { std::vector<Eigen::Affine3d> T; for (int i = 0; i < 3; i++) T.push_back(vectorToAffine3d(poses[3 * i], poses[3 * i + 1], 0, 0, 0, poses[3 * i + 2])); return T; };
    std::vector<Eigen::Affine3d> T = T0(), Tc = T0();
    std::vector<Matrix6d> cov(3, Identity66);
    std::vector<double> residuals, par;
    double cost = 0, score = 0;
    n_scan_normal_reg gc(cs.cost, cs.loss, 0.1, cs.opt);           // GetCost at the initial poses (GetCost associates with radius 2 r: n_scan_normal.cpp:186-211)
    gc.SetParameters(cs.mo, cs.mi);
    const bool okc = gc.GetCost(scans, Tc, cost, residuals);
    n_scan_normal_reg reg(cs.cost, cs.loss, 0.1, cs.opt);
    reg.SetParameters(cs.mo, cs.mi);
    const bool ok = reg.Register(scans, T, cov);
    Affine3dToVectorXYeZ(T.back(), par);
    int nres = 0;
    reg.getScore(score, nres);
    w.put(std::string("r_") + cs.name + "_pose", "float64", par, {3});
    w.put(std::string("r_") + cs.name + "_meta", "float64", std::vector<double>{(double)ok, (double)nres, score, (double)okc, cost, (double)residuals.size()}, {6});
  }
  // ---- the 1-NN tie rule (pointnormal.cpp:238-254 -> pcl::KdTreeFLANN::nearestKSearch(1), FLANN 1.9.1 KDTreeSingleIndex(15)) ----
  // (a) every cell of scan 0 asks for its own mean: cells with bit-identical float means (~17 groups per scan) are exact ties,
  //     the answer is the tree's winner of the group; (b) the queries export_inputs.py wrote (scan 1's means moved by the start
  //     pose: the searches of Register's first association pass), radius 2 r = 6 m.
  //     Compared by tests/test_golden.py::test_reference_tie_rule against the oracle's "lowest index".
  {
    std::vector<int32_t> self_idx;
    for (const cell& c : scans[0]->GetCells()) {
      const std::vector<int> r = scans[0]->GetClosestIdx(Eigen::Vector2d(c.u_(0), c.u_(1)), 6.0);
      self_idx.push_back(r.empty() ? -1 : r[0]);
    }
    w.put("r_tie_self_idx", "int32", self_idx, {(long)self_idx.size()});
    const Arr& tq = in.at("tie_q");
    const double* q = (const double*)tq.bytes.data();
    std::vector<int32_t> q_idx;
    for (long i = 0; i < tq.dims[0]; i++) {
      const std::vector<int> r = scans[0]->GetClosestIdx(Eigen::Vector2d(q[2 * i], q[2 * i + 1]), 6.0);
      q_idx.push_back(r.empty() ? -1 : r[0]);
    }
    w.put("r_tie_query_idx", "int32", q_idx, {(long)q_idx.size()});
  }
  // ---- a 50-frame OdometryKeyframeFuser pose trace (odometrykeyframefuser.cpp:143-259), CFEAR-3 preset, from raw sweeps ----
  // (launch/oxford/eval/params/baseline/oxford_cfear-3: P2P, Huber 0.1, weight option 4, 4 keyframes, res 3, k 40, z_min 60,
  //  intensity weights, compensation on).  One run pins the filter -> compensate -> surface points -> Register (LM path, tie rule)
  //  -> keyframe policy chain end to end.  Compared by tests/test_golden.py::test_reference_odometry_trace.
  {
    const Arr& seq = in.at("seq_img");
    const int nf = (int)seq.dims[0], rows = (int)seq.dims[1], cols = (int)seq.dims[2];
    OdometryKeyframeFuser::Parameters op;
    op.cost_type = "P2P"; op.weight_opt = weightoption::Combined_weights; op.submap_scan_size = 4; op.weight_intensity_ = true;
    op.res = 3.0; op.loss_type_ = "Huber"; op.loss_limit_ = 0.1; op.visualize = false; op.publish_tf_ = false; op.store_graph = false;
    op.compensate = true; op.radar_ccw = false; op.use_guess = true; op.use_keyframe = true;
    MapPointNormal::downsample_factor = 1;
    OdometryKeyframeFuser fuser(op, true);
    std::vector<double> trace;
    std::vector<int32_t> npts;
    for (int f = 0; f < nf; f++) {
      cv_bridge::CvImagePtr cv(new cv_bridge::CvImage());
      cv->encoding = "mono8";
      cv->image = cv::Mat(rows, cols, CV_8UC1, (void*)(seq.bytes.data() + (size_t)f * rows * cols)).clone();
      StructuredKStrongest filt(cv, 60, 40, 2.5, 0.0438);
      Cloud::Ptr c(new Cloud()), cp(new Cloud());
      filt.getPeaksFilteredPointCloud(c, false);
      filt.getPeaksFilteredPointCloud(cp, true);
      npts.push_back((int32_t)c->size());
      Eigen::Affine3d Tcurr = Eigen::Affine3d::Identity();
      fuser.pointcloudCallback(c, cp, Tcurr, ros::Time(1547120000 + 0.25 * f));
      std::vector<double> par;
      Affine3dToVectorXYeZ(Tcurr, par);
      trace.insert(trace.end(), par.begin(), par.end());
    }
    w.put("r_odom_trace", "float64", trace, {nf, 3});
    w.put("r_odom_npts", "int32", npts, {nf});
  }
  // ---- one simple_graph.sgh written by the reference's own Boost archive (types.cpp:103-130) ----
  {
    simple_graph g;
    for (int i = 0; i < 3; i++) {
      const Eigen::Affine3d Ti = vectorToAffine3d(poses[3 * i], poses[3 * i + 1], 0, 0, 0, poses[3 * i + 2]);
      RadarScan s(Ti, Eigen::Affine3d::Identity(), clouds[i], clouds[i], scans[i], ros::Time(1547120000 + i, 0));
      std::vector<Constraint3d> cons;
      if (i > 0) { Constraint3d c; c.id_begin = i; c.id_end = i - 1; c.t_be = PoseEigToCeres(Ti.inverse() * vectorToAffine3d(poses[3 * i - 3], poses[3 * i - 2], 0, 0, 0, poses[3 * i - 1])); c.information = Covariance::Identity(); c.type = ConstraintType::odometry; c.quality["sc-sim"] = 0.25; c.info = "odom"; cons.push_back(c); }
      g.push_back(std::make_pair(s, cons));
    }
    SaveSimpleGraph(std::string(argv[2]) + "/ref_simple_graph.sgh", g);
  }
  printf("ref_golden: done -> %s\n", argv[2]);
  return 0;
}
