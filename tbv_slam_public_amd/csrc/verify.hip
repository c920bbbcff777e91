// verify.hip -- loop-candidate verification: the caller that chains the registration path with its two
// alignment-quality measures and turns them into a loop probability.
//
// Restates, for a BATCH of candidates, what the loop-closure thread does per candidate in
// ScanContextClosure::SearchAndAddConstraint (tbv_slam/src/tbv_slam/loopclosure.cpp:658-725):
//   RegisterLoopCandidate (:320-364) -> loopclosure::Register (:35-97)          [matcher_kernel]
//   VerifyLoopCandidate (:365-384) -> VerifyByAlignment (:759-774) ->
//     ScanLearningInterface::PredAlignment (alignmentinterface.cpp:349-367):
//       getCorAlQualityMeasure (:437-456)                                         [coral_kernel]
//       getCFEARQualityMeasure (:459-478) -> CFEARQuality (AlignmentQuality.cpp:330-354)  [matcher_kernel, cost only]
//       combined logistic model -> quality["alignment_quality"]
//   VerificationModel (:220-238) over {odom-bounds, sc-sim, alignment_quality}
//   ApplyConstratins (:261-274): accept by probability, all candidates or the best of each query.
// Three device launches for the whole batch; the classifiers are a dozen multiply-adds per candidate on the host,
// as in the reference.  Training the classifiers (sklearn through pybind11 in the reference,
// alignmentinterface.cpp:192-222) is not part of the library: coefficients come in through cfear_verify_params.
#include <chrono>
#include <algorithm>
#include <cmath>
#include <numeric>
#include <vector>

#include "common.hpp"

namespace {

// planar Affine3d algebra on (x, y, theta)
inline void xyt_compose(const double a[3], const double b[3], double o[3]) {      // a * b
  const double c = std::cos(a[2]), s = std::sin(a[2]);
  const double x = c * b[0] - s * b[1] + a[0], y = s * b[0] + c * b[1] + a[1];
  o[0] = x; o[1] = y; o[2] = a[2] + b[2];
}
inline void xyt_inverse(const double a[3], double o[3]) {                         // a^-1
  const double c = std::cos(a[2]), s = std::sin(a[2]);
  const double x = -(c * a[0] + s * a[1]), y = s * a[0] - c * a[1];
  o[0] = x; o[1] = y; o[2] = -a[2];
}

}  // namespace

extern "C" void cfear_verify_params_default(cfear_verify_params* p) {
  if (!p) return;
  // tbv_slam/model_parameters/trained_alignment_classifier.txt (intercept, then CorAl x 3, CFEAR x 3), loaded by
  // ScanLearningInterface::LoadCoefficients (alignmentinterface.cpp:396-403) in the shipped launch files
  p->align_intercept = -8.42595;
  const double ac[6] = {-15.2287, 7.47573, -0.0680198, -1.74182, 0.0945444, 0.022217};
  for (int k = 0; k < 6; k++) p->align_coef[k] = ac[k];
  // preset of VerificationModel when no classifier was fitted (loopclosure.cpp:224-232)
  p->loop_intercept = 2.67958289;
  p->loop_coef[0] = -2.89398535; p->loop_coef[1] = -9.40230684; p->loop_coef[2] = 0.23891265;
  p->model_threshold = 0.8;                    // loopclosure.h:133
  p->all_candidates = 1;                       // :137
  p->verification_disabled = 0;                // :120
  p->use_covariance_sampling = 0;              // :130
  p->pad = 0;
  cfear_coral_params_default(&p->coral);       // radius 1.0, no intensity weights (alignmentinterface.cpp:444)
  p->sampling.xy_range = 0.4;                  // loopclosure.cpp:108: linspace(-0.2, 0.2)
  p->sampling.yaw_range = 0.0044;              // :109
  p->sampling.samples_per_axis = 3;            // :110
  p->sampling.pad = 0;
  p->sampling.covariance_scaler = 4.0;         // :112
}

// loopclosure::VerifyByOdometry (loopclosure.cpp:776-808): how far the odometry chain between the two nodes says
// they are apart, relative to the distance travelled.  rel_xyt [n][3]: ConstraintsHandler::RelativeMotion(i, i+1)
// for i = to .. from-1 as planar poses.
extern "C" int cfear_verify_by_odometry(const double* rel_xyt, int32_t n, double odom_sigma_error,
                                        int32_t verify_via_odometry, double* similarity) {
  if (!similarity || n < 0 || (n > 0 && !rel_xyt)) return CFEAR_ERR_INVALID_ARGUMENT;
  if (!verify_via_odometry) { *similarity = 1.0; return CFEAR_OK; }             // :777-781
  double T[3] = {0.0, 0.0, 0.0}, trav = 0.0;
  for (int i = 0; i < n; i++) {
    const double* d = rel_xyt + 3 * (size_t)i;
    trav += std::sqrt(d[0] * d[0] + d[1] * d[1]);
    double t[3];
    xyt_compose(T, d, t);
    T[0] = t[0]; T[1] = t[1]; T[2] = t[2];
  }
  const double est = std::sqrt(T[0] * T[0] + T[1] * T[1]);
  const double error = std::max(est - 5.0, 0.0);                                 // within 5 m: always nearby (:792)
  const double rel = error / trav;                                               // n == 0: 0/0, as the reference
  const double prob = std::exp(-rel * rel / (2.0 * odom_sigma_error * odom_sigma_error));
  *similarity = 1.0 - prob;
  return CFEAR_OK;
}

extern "C" int cfear_verify_loop_candidates(cfear_ctx* ctx, const cfear_verify_job* jobs, int32_t n_jobs,
                                            const cfear_verify_params* par, cfear_verify_result* results) {
  if (!ctx) return CFEAR_ERR_INVALID_ARGUMENT;
  if (!jobs || !par || !results || n_jobs < 0) return cfear_set_error(ctx, CFEAR_ERR_INVALID_ARGUMENT, "null argument");
  if (n_jobs == 0) return CFEAR_OK;
  for (int j = 0; j < n_jobs; j++)
    if (!jobs[j].from_scan || !jobs[j].to_scan)
      return cfear_set_error(ctx, CFEAR_ERR_INVALID_ARGUMENT, "candidate %d: null scan handle", j);
  const size_t n = (size_t)n_jobs;
#ifdef CFEAR_VERIFY_TIMING
  auto t_last = std::chrono::steady_clock::now();
  double t_acc[8] = {0};
  auto mark = [&](int k) { const auto t = std::chrono::steady_clock::now(); t_acc[k] += std::chrono::duration<double, std::micro>(t - t_last).count(); t_last = t; };
#else
  auto mark = [](int) {};
#endif

  // ---- RegisterLoopCandidate: scans {to, from}, poses {Tto = Tfrom * t_be, Tfrom}; P2L, Huber 0.1, uniform
  // weights, SetParameters(4, 10) (loopclosure.cpp:56-57) ------------------------------------------------------
  cfear_reg_params rp;
  cfear_reg_params_default(&rp);
  rp.cost = CFEAR_P2L;
  rp.max_itr_association = 4;
  rp.max_itr_solver = 10;
  std::vector<const cfear_scan*> handles(2 * n);
  std::vector<double> poses(6 * n);
  std::vector<cfear_reg_job> rj(n);
  for (size_t j = 0; j < n; j++) {
    handles[2 * j] = jobs[j].to_scan;
    handles[2 * j + 1] = jobs[j].from_scan;
    xyt_compose(jobs[j].from_pose, jobs[j].t_be_guess, &poses[6 * j]);
    for (int k = 0; k < 3; k++) poses[6 * j + 3 + k] = jobs[j].from_pose[k];
    rj[j].scans = &handles[2 * j]; rj[j].n_scans = 2; rj[j].pad = 0; rj[j].poses_xyt = &poses[6 * j];
  }
  std::vector<cfear_reg_result> reg(n);
  mark(0);
  int rc = cfear_register_batch(ctx, rj.data(), n_jobs, &rp, reg.data());
  if (rc != CFEAR_OK) return rc;
  mark(1);

  // covariance: Register's constant, or the sampled one (:62-71)
  std::vector<double> cov(36 * n, 0.0);
  std::vector<int32_t> sampled(n, 0);
  if (par->use_covariance_sampling) {
    std::vector<double> posed(poses);                       // T_vek after Register
    for (size_t j = 0; j < n; j++)
      if (reg[j].status == CFEAR_OK) for (int k = 0; k < 3; k++) posed[6 * j + 3 + k] = reg[j].pose[k];
    std::vector<cfear_reg_job> cj(rj);
    for (size_t j = 0; j < n; j++) cj[j].poses_xyt = &posed[6 * j];
    rc = cfear_covariance_by_sampling_batch(ctx, cj.data(), n_jobs, &rp, reg.data(), &par->sampling, cov.data(), nullptr,
                                            sampled.data());
    if (rc != CFEAR_OK) return rc;
  } else {
    for (size_t j = 0; j < n; j++) { cov[36 * j] = 0.01; cov[36 * j + 7] = 0.01; cov[36 * j + 35] = 1e-4; }   // n_scan_normal.cpp:171
  }

  // Talign first: the CorAl jobs need nothing else of this block, and their kernel (the longest of a verification step) then
  // runs while the host rotates the covariances and marshals the cost jobs below
  std::vector<double> inv_yaw(n, 0.0);
  for (size_t j = 0; j < n; j++) {
    cfear_verify_result& r = results[j];
    r.reg = reg[j];
    r.reg_ok = reg[j].status == CFEAR_OK ? 1 : 0;
    r.cov_sampled = r.reg_ok ? sampled[j] : 0;
    if (r.reg_ok) {                                         // loopclosure.cpp:90-94
      double inv[3];
      xyt_inverse(reg[j].pose, inv);                        // Trevised^-1
      xyt_compose(inv, &poses[6 * j], r.t_be);              // Talign = Trevised^-1 * Tto
      inv_yaw[j] = inv[2];
    } else {                                                // Tdiff keeps its initial Identity (:351-353)
      r.t_be[0] = r.t_be[1] = r.t_be[2] = 0.0;
    }
  }

  // ---- VerifyByAlignment at Tfrom, Tto = Tfrom * t_be (loopclosure.cpp:367-372): current = from, prev = to ------
  std::vector<double> to_pose(3 * n);
  for (size_t j = 0; j < n; j++) xyt_compose(jobs[j].from_pose, results[j].t_be, &to_pose[3 * j]);
  std::vector<cfear_coral_job> cjobs(n);
  for (size_t j = 0; j < n; j++) {
    cfear_coral_job& c = cjobs[j];
    c.ref_xyzi = jobs[j].from_peaks; c.n_ref = jobs[j].n_from;      // CreateQualityType(scan_curr, scan_prev): ref = current
    c.src_xyzi = jobs[j].to_peaks; c.n_src = jobs[j].n_to;
    for (int k = 0; k < 3; k++) { c.ref_pose[k] = jobs[j].from_pose[k]; c.src_pose[k] = to_pose[3 * j + k]; c.offset[k] = 0.0; }
  }
  std::vector<cfear_coral_result> coral(n);
  mark(2);
  CoralPending pend;
  rc = cfear_coral_enqueue(ctx, cjobs.data(), n_jobs, &par->coral, false, pend);
  if (rc != CFEAR_OK) return rc;
  mark(3);

  for (size_t j = 0; j < n; j++) {                          // (the CorAl kernel is running)
    cfear_verify_result& r = results[j];
    double* C = r.cov;
    if (r.reg_ok) {
      for (int k = 0; k < 36; k++) C[k] = cov[36 * j + k];
      // reg_cov.block<3,3>(0,0) = R^-1 * block * R^-T: only the x-y part of the block is touched by a yaw rotation
      const double c = std::cos(inv_yaw[j]), s = std::sin(inv_yaw[j]);
      const double R[9] = {c, -s, 0, s, c, 0, 0, 0, 1};
      double B[9], T[9];
      for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) B[a * 3 + b] = C[a * 6 + b];
      for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) {
        double t = 0.0;
        for (int k = 0; k < 3; k++) t += R[a * 3 + k] * B[k * 3 + b];
        T[a * 3 + b] = t;
      }
      for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) {
        double t = 0.0;
        for (int k = 0; k < 3; k++) t += T[a * 3 + k] * R[b * 3 + k];
        C[a * 6 + b] = t;
      }
    } else {                                                // Cov keeps its initial Identity (:351-353)
      for (int k = 0; k < 36; k++) C[k] = (k % 7 == 0) ? 1.0 : 0.0;
    }
  }

  cfear_reg_params qp;                                              // n_scan_normal_reg(P2L, Huber, 0.3), fresh: itr_ = 0
  cfear_reg_params_default(&qp);
  qp.cost = CFEAR_P2L;
  qp.loss_limit = 0.3;
  qp.itr = 0;
  std::vector<const cfear_scan*> qh(2 * n);
  std::vector<double> qposes(6 * n);
  std::vector<cfear_reg_job> qj(n);
  for (size_t j = 0; j < n; j++) {
    qh[2 * j] = jobs[j].from_scan;                                  // feature_vek = {ref, src}
    qh[2 * j + 1] = jobs[j].to_scan;
    for (int k = 0; k < 3; k++) { qposes[6 * j + k] = jobs[j].from_pose[k]; qposes[6 * j + 3 + k] = to_pose[3 * j + k]; }
    qj[j].scans = &qh[2 * j]; qj[j].n_scans = 2; qj[j].pad = 0; qj[j].poses_xyt = &qposes[6 * j];
  }
  std::vector<cfear_reg_result> q(n);
  mark(4);
  rc = cfear_get_cost_batch(ctx, qj.data(), n_jobs, &qp, q.data());   // (same stream: returns after the CorAl kernel too)
  const int rc_coral = cfear_coral_collect(ctx, cjobs.data(), pend, coral.data(), nullptr);
  if (rc != CFEAR_OK) return rc;
  if (rc_coral != CFEAR_OK) return rc_coral;
  mark(5);

  for (size_t j = 0; j < n; j++) {
    cfear_verify_result& r = results[j];
    r.coral[0] = coral[j].joint; r.coral[1] = coral[j].sep; r.coral[2] = coral[j].overlap;
    if (q[j].status == CFEAR_OK) {                                  // AlignmentQuality.cpp:344-348
      r.cfear[0] = q[j].final_cost;
      r.cfear[1] = (double)q[j].num_residuals;
      r.cfear[2] = (cfear_scan_size(jobs[j].to_scan) + cfear_scan_size(jobs[j].from_scan)) / 2.0;
    } else {
      r.cfear[0] = r.cfear[1] = r.cfear[2] = 0.0;                   // :349-351
    }
    double z = par->align_intercept;                                // predict_linear (alignmentinterface.cpp:271-279)
    for (int k = 0; k < 3; k++) z += par->align_coef[k] * r.coral[k];
    for (int k = 0; k < 3; k++) z += par->align_coef[3 + k] * r.cfear[k];
    r.alignment_quality = z;
    r.odom_bounds = jobs[j].odom_bounds;
    r.sc_sim = jobs[j].sc_sim;
    if (par->verification_disabled) {
      r.probability = 0.0;                                          // loopclosure.cpp:377
    } else {
      const double zl = par->loop_coef[0] * r.odom_bounds + par->loop_coef[1] * r.sc_sim +
                        par->loop_coef[2] * r.alignment_quality + par->loop_intercept;
      r.probability = 1.0 / (1.0 + std::exp(-zl));
    }
    r.accepted = 0;
    r.rank = 0;
  }

  // ---- ApplyConstratins per query (jobs sharing `group`): sort by probability, larger first ------------------
  std::vector<int> order(n);
  std::iota(order.begin(), order.end(), 0);
  std::stable_sort(order.begin(), order.end(), [&](int a, int b) {
    if (jobs[a].group != jobs[b].group) return jobs[a].group < jobs[b].group;
    return results[a].probability > results[b].probability;
  });
  for (size_t i = 0; i < n;) {
    size_t e = i;
    while (e < n && jobs[order[e]].group == jobs[order[i]].group) e++;
    for (size_t k = i; k < e; k++) {
      cfear_verify_result& r = results[order[k]];
      r.rank = (int32_t)(k - i);
      const bool considered = par->all_candidates || k == i;
      r.accepted = considered && r.probability > par->model_threshold ? 1 : 0;
    }
    i = e;
  }
  mark(6);
#ifdef CFEAR_VERIFY_TIMING
  fprintf(stderr, "verify us: marshal reg %.0f | register_batch %.0f | Talign + coral jobs %.0f | coral enqueue %.0f | cov + cost jobs %.0f | get_cost_batch + coral results %.0f | classify + sort %.0f\n",
          t_acc[0], t_acc[1], t_acc[2], t_acc[3], t_acc[4], t_acc[5], t_acc[6]);
#endif
  return CFEAR_OK;
}
