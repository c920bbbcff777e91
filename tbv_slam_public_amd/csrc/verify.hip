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
#include <map>
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

// ---- the device chain (the usual configuration: Register's constant covariance) ---------------------------------------
// Round 5's step was half host time: between the registration and the two quality measures the host read the poses back,
// composed Talign / Tto per candidate, marshalled 4096 CorAl jobs and 4096 cost jobs and uploaded them, and at the end ran
// the classifiers -- 0.5 ms of a 2.1 ms step with the GPU idle.  Now ONE 296-byte record per candidate goes up and the
// whole chain is enqueued at once:  expand -> matcher (Register) -> prepare (Talign, Tto, the cost and CorAl job records,
// the rotated covariance) -> matcher (cost only: CFEARQuality) -> coral_kernel -> finish (feature vector, both logistic
// models) -> ONE read-back of the 480-byte results.  The host keeps what needs a sort: ApplyConstratins.
namespace {

struct VerifyDev {                      // what a candidate needs on the device
  ScanView to, from;
  const float4* from_peaks;
  const float4* to_peaks;
  int32_t n_from, n_to;
  double from_pose[3], t_be_guess[3];
  double sc_sim, odom_bounds;
};

struct VerifyChain {
  const VerifyDev* cand;
  char* reg_jobs;                       // RegJob records, `stride` bytes apart: scans {to, from}, poses {Tto, Tfrom}
  char* cost_jobs;                      // scans {from, to}, poses {Tfrom, Tto'}: feature_vek = {ref, src} (AlignmentQuality.cpp:330-354)
  size_t stride;
  CoralJob* coral_jobs;
  const cfear_reg_result* reg;
  const cfear_reg_result* q;
  const cfear_coral_result* coral;
  cfear_verify_result* out;
  int32_t* first_bad;                   // smallest candidate index whose CorAl job failed (INT_MAX: none)
  int n;
  cfear_verify_params par;
};

__device__ __forceinline__ void d_xyt_compose(const double a[3], const double b[3], double o[3]) {
  double s, c;
  sincos(a[2], &s, &c);
  const double x = c * b[0] - s * b[1] + a[0], y = s * b[0] + c * b[1] + a[1];
  o[0] = x; o[1] = y; o[2] = a[2] + b[2];
}
__device__ __forceinline__ void d_xyt_inverse(const double a[3], double o[3]) {
  double s, c;
  sincos(a[2], &s, &c);
  const double x = -(c * a[0] + s * a[1]), y = s * a[0] - c * a[1];
  o[0] = x; o[1] = y; o[2] = -a[2];
}

// RegisterLoopCandidate's problem (loopclosure.cpp:35-60): scans {to, from}, poses {Tto = Tfrom * t_be, Tfrom}
__global__ __launch_bounds__(256) void verify_expand_kernel(const VerifyChain c) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j == 0) *c.first_bad = 0x7fffffff;
  if (j >= c.n) return;
  const VerifyDev& v = c.cand[j];
  RegJob* r = (RegJob*)(c.reg_jobs + (size_t)j * c.stride);
  r->n_scans = 2; r->itr = 0;
  double tto[3];
  d_xyt_compose(v.from_pose, v.t_be_guess, tto);
#pragma unroll
  for (int k = 0; k < 3; k++) { r->poses[0][k] = tto[k]; r->poses[1][k] = v.from_pose[k]; }
  r->scans[0] = v.to; r->scans[1] = v.from;
}

// Talign = Trevised^-1 * Tto (loopclosure.cpp:90-94), the covariance in that frame (:62-71 constant; :93), and the job records
// of the two quality measures at Tfrom, Tto' = Tfrom * t_be (:367-372; current = from, prev = to)
__global__ __launch_bounds__(256) void verify_prepare_kernel(const VerifyChain c) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= c.n) return;
  const VerifyDev& v = c.cand[j];
  const cfear_reg_result reg = c.reg[j];
  const RegJob* rj = (const RegJob*)(c.reg_jobs + (size_t)j * c.stride);
  cfear_verify_result& r = c.out[j];
  r.reg = reg;
  r.reg_ok = reg.status == CFEAR_OK ? 1 : 0;
  r.cov_sampled = 0;
  double* C = r.cov;
  if (r.reg_ok) {
    double inv[3];
    d_xyt_inverse(reg.pose, inv);                          // Trevised^-1
    d_xyt_compose(inv, rj->poses[0], r.t_be);              // Talign = Trevised^-1 * Tto
    for (int k = 0; k < 36; k++) C[k] = 0.0;
    C[0] = 0.01; C[7] = 0.01; C[35] = 1e-4;                 // n_scan_normal.cpp:171-175
    // reg_cov.block<3,3>(0,0) = R^-1 * block * R^-T: only the x-y part of the block is touched by a yaw rotation
    double sn, cs;
    sincos(inv[2], &sn, &cs);
    const double R[9] = {cs, -sn, 0, sn, cs, 0, 0, 0, 1};
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
  } else {                                                 // Tdiff and Cov keep their initial Identity (:351-353)
    r.t_be[0] = r.t_be[1] = r.t_be[2] = 0.0;
    for (int k = 0; k < 36; k++) C[k] = (k % 7 == 0) ? 1.0 : 0.0;
  }
  double to_pose[3];
  d_xyt_compose(v.from_pose, r.t_be, to_pose);
  RegJob* qj = (RegJob*)(c.cost_jobs + (size_t)j * c.stride);
  qj->n_scans = 2; qj->itr = 0;
#pragma unroll
  for (int k = 0; k < 3; k++) { qj->poses[0][k] = v.from_pose[k]; qj->poses[1][k] = to_pose[k]; }
  qj->scans[0] = v.from; qj->scans[1] = v.to;
  CoralJob& cj = c.coral_jobs[j];                          // CreateQualityType(scan_curr, scan_prev): ref = current
  cj.ref = v.from_peaks; cj.src = v.to_peaks; cj.n_ref_ptr = nullptr; cj.n_src_ptr = nullptr;
  cj.n_ref = v.n_from; cj.n_src = v.n_to;
#pragma unroll
  for (int k = 0; k < 3; k++) { cj.ref_pose[k] = v.from_pose[k]; cj.src_pose[k] = to_pose[k]; cj.offset[k] = 0.0; }
}

// PredAlignment's feature vector and the two logistic models (alignmentinterface.cpp:349-367, 271-279; loopclosure.cpp:220-238)
__global__ __launch_bounds__(256) void verify_finish_kernel(const VerifyChain c) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= c.n) return;
  const VerifyDev& v = c.cand[j];
  cfear_verify_result& r = c.out[j];
  const cfear_coral_result co = c.coral[j];
  const cfear_reg_result q = c.q[j];
  if (co.status != CFEAR_OK && co.status != CFEAR_ERR_EMPTY_CLOUD) atomicMin(c.first_bad, j);
  r.coral[0] = co.joint; r.coral[1] = co.sep; r.coral[2] = co.overlap;
  if (q.status == CFEAR_OK) {                               // AlignmentQuality.cpp:344-348
    r.cfear[0] = q.final_cost;
    r.cfear[1] = (double)q.num_residuals;
    r.cfear[2] = (*v.to.n_cells + *v.from.n_cells) / 2.0;
  } else {
    r.cfear[0] = r.cfear[1] = r.cfear[2] = 0.0;             // :349-351
  }
  double z = c.par.align_intercept;                         // predict_linear (alignmentinterface.cpp:271-279)
  for (int k = 0; k < 3; k++) z += c.par.align_coef[k] * r.coral[k];
  for (int k = 0; k < 3; k++) z += c.par.align_coef[3 + k] * r.cfear[k];
  r.alignment_quality = z;
  r.odom_bounds = v.odom_bounds;
  r.sc_sim = v.sc_sim;
  if (c.par.verification_disabled) {
    r.probability = 0.0;                                    // loopclosure.cpp:377
  } else {
    const double zl = c.par.loop_coef[0] * r.odom_bounds + c.par.loop_coef[1] * r.sc_sim + c.par.loop_coef[2] * r.alignment_quality +
                      c.par.loop_intercept;
    r.probability = 1.0 / (1.0 + exp(-zl));
  }
  r.accepted = 0;
  r.rank = 0;
}

// ApplyConstratins per query (candidates sharing `group`): sort by probability, larger first; accept above the threshold, every
// candidate or only the best (loopclosure.cpp:261-274).  The keys are pulled out of the 480-byte records first (the sort then
// walks 16-byte entries); candidate lists arrive grouped by query, so the usual case is one pass over short runs.
void apply_constraints_groups(const int32_t* groups, size_t group_stride_bytes, size_t n, const cfear_verify_params* par,
                              cfear_verify_result* results) {
  struct Key { int32_t group, index; double prob; };
  std::vector<Key> keys(n);
  bool grouped = true;
  for (size_t i = 0; i < n; i++) {
    keys[i].group = *(const int32_t*)((const char*)groups + i * group_stride_bytes);
    keys[i].index = (int32_t)i;
    keys[i].prob = results[i].probability;
    grouped = grouped && (i == 0 || keys[i].group >= keys[i - 1].group);
  }
  auto by_prob = [](const Key& a, const Key& b) { return a.prob > b.prob; };
  if (!grouped) std::stable_sort(keys.begin(), keys.end(), [](const Key& a, const Key& b) { return a.group < b.group; });
  for (size_t i = 0; i < n;) {
    size_t e = i;
    while (e < n && keys[e].group == keys[i].group) e++;
    std::stable_sort(keys.begin() + (ptrdiff_t)i, keys.begin() + (ptrdiff_t)e, by_prob);
    for (size_t k = i; k < e; k++) {
      cfear_verify_result& r = results[keys[k].index];
      r.rank = (int32_t)(k - i);
      const bool considered = par->all_candidates || k == i;
      r.accepted = considered && r.probability > par->model_threshold ? 1 : 0;
    }
    i = e;
  }
}
void apply_constraints(const cfear_verify_job* jobs, size_t n, const cfear_verify_params* par, cfear_verify_result* results) {
  if (n) apply_constraints_groups(&jobs[0].group, sizeof(cfear_verify_job), n, par, results);
}

int verify_device_chain(cfear_ctx* ctx, const cfear_verify_job* jobs, int32_t n_jobs, const cfear_verify_params* par,
                        cfear_verify_result* results) {
  const size_t n = (size_t)n_jobs;
#ifdef CFEAR_VERIFY_TIMING
  auto t_last = std::chrono::steady_clock::now();
  double t_acc[8] = {0};
  auto mark = [&](int k) { const auto t = std::chrono::steady_clock::now(); t_acc[k] += std::chrono::duration<double, std::micro>(t - t_last).count(); t_last = t; };
#else
  auto mark = [](int) {};
#endif
  CFEAR_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  // ---- host: one record per candidate; peak clouds that live on the host are staged once each -----------------------
  std::map<const float*, const float4*> where;           // cloud -> its device address (one hipPointerGetAttributes per distinct cloud)
  std::vector<std::pair<const float*, int>> to_stage;
  size_t stage_floats = 0;
  int max_cells = 0, cap = 1;
  for (size_t j = 0; j < n; j++) {
    const cfear_verify_job& jb = jobs[j];
    if (jb.from_scan->ctx != ctx || jb.to_scan->ctx != ctx) return cfear_set_error(ctx, CFEAR_ERR_INVALID_ARGUMENT, "candidate %zu: scan belongs to another context", j);
    if (jb.n_from < 0 || jb.n_to < 0 || (jb.n_from > 0 && !jb.from_peaks) || (jb.n_to > 0 && !jb.to_peaks))
      return cfear_set_error(ctx, CFEAR_ERR_INVALID_ARGUMENT, "job %zu: null cloud", j);
    if ((long long)jb.n_from + jb.n_to > cfear_coral_max_points())
      return cfear_set_error(ctx, CFEAR_ERR_CAPACITY, "job %zu: %d + %d points exceed %d", j, jb.n_from, jb.n_to, cfear_coral_max_points());
    cap = std::max(cap, jb.n_from + jb.n_to);
    const float* ptrs[2] = {jb.from_peaks, jb.to_peaks};
    const int ns[2] = {jb.n_from, jb.n_to};
    for (int c = 0; c < 2; c++) {
      if (ns[c] == 0 || !ptrs[c]) continue;
      auto it = where.find(ptrs[c]);
      if (it == where.end()) {
        if (cfear_is_device_ptr(ptrs[c])) where[ptrs[c]] = (const float4*)ptrs[c];
        else { where[ptrs[c]] = (const float4*)(uintptr_t)(stage_floats * 4); to_stage.emplace_back(ptrs[c], ns[c]); stage_floats += ((size_t)ns[c] * 4 + 3) & ~(size_t)3; }
      } else if (!to_stage.empty()) {
        for (auto& ts : to_stage) if (ts.first == ptrs[c]) { if (ns[c] > ts.second) return cfear_set_error(ctx, CFEAR_ERR_INVALID_ARGUMENT, "job %zu: a host cloud is used with two lengths", j); break; }
      }
    }
    const int ct = cfear_scan_size(jb.to_scan), cf = cfear_scan_size(jb.from_scan);
    if (ct < 0) return ct;
    if (cf < 0) return cf;
    max_cells = std::max(max_cells, std::max(ct, cf));
  }
  float* d_stage = nullptr;
  if (stage_floats) {
    d_stage = (float*)cfear_workspace(ctx, 8, stage_floats * 4);
    if (!d_stage) return cfear_set_error(ctx, CFEAR_ERR_HIP, "workspace allocation failed");
    for (auto& ts : to_stage) {
      const size_t off = (size_t)(uintptr_t)where[ts.first] / 4;
      CFEAR_HIP_CHECK(ctx, hipMemcpyAsync(d_stage + off, ts.first, (size_t)ts.second * 16, hipMemcpyHostToDevice, ctx->stream));
      where[ts.first] = (const float4*)(d_stage + off);
    }
  }
  mark(0);
  VerifyDev* hc = (VerifyDev*)cfear_pinned(ctx, n * sizeof(VerifyDev));
  if (!hc) return cfear_set_error(ctx, CFEAR_ERR_HIP, "pinned staging allocation failed");
  for (size_t j = 0; j < n; j++) {
    const cfear_verify_job& jb = jobs[j];
    VerifyDev& v = hc[j];
    v.to = jb.to_scan->view; v.from = jb.from_scan->view;
    v.from_peaks = jb.n_from > 0 ? where[jb.from_peaks] : nullptr;
    v.to_peaks = jb.n_to > 0 ? where[jb.to_peaks] : nullptr;
    v.n_from = jb.n_from; v.n_to = jb.n_to;
    for (int k = 0; k < 3; k++) { v.from_pose[k] = jb.from_pose[k]; v.t_be_guess[k] = jb.t_be_guess[k]; }
    v.sc_sim = jb.sc_sim; v.odom_bounds = jb.odom_bounds;
  }
  mark(1);
  // ---- device buffers: one workspace, carved ---------------------------------------------------------------------------
  const size_t stride = reg_job_stride(2);
  auto up = [](size_t b) { return (b + 255) / 256 * 256; };
  const size_t o_cand = 0, o_reg_jobs = o_cand + up(n * sizeof(VerifyDev)), o_cost_jobs = o_reg_jobs + up(n * stride),
               o_coral_jobs = o_cost_jobs + up(n * stride), o_reg = o_coral_jobs + up(n * sizeof(CoralJob)),
               o_q = o_reg + up(n * sizeof(cfear_reg_result)), o_coral = o_q + up(n * sizeof(cfear_reg_result)),
               o_out = o_coral + up(n * sizeof(cfear_coral_result)), o_flag = o_out + up(n * sizeof(cfear_verify_result)), total = o_flag + 256;
  char* ws = (char*)cfear_workspace(ctx, 14, total);
  // RegisterLoopCandidate: P2L, Huber 0.1, uniform weights, SetParameters(4, 10) (loopclosure.cpp:56-57); CFEARQuality:
  // n_scan_normal_reg(P2L, Huber, 0.3), fresh -- itr_ = 0 (AlignmentQuality.cpp:330-354)
  cfear_reg_params rp, qp;
  cfear_reg_params_default(&rp);
  rp.cost = CFEAR_P2L; rp.max_itr_association = 4; rp.max_itr_solver = 10;
  cfear_reg_params_default(&qp);
  qp.cost = CFEAR_P2L; qp.loss_limit = 0.3; qp.itr = 0;
  int pairs_cap = 1, pairs_cap_q = 1;
  RegLaunchHint hint, hint_q;
  cfear_reg_pair_geometry(&rp, max_cells, max_cells, &pairs_cap, &hint);
  cfear_reg_pair_geometry(&qp, max_cells, max_cells, &pairs_cap_q, &hint_q);
  hint_q.small_pairs = false;
  char* scr = (char*)cfear_workspace(ctx, 7, cfear_register_scratch_bytes(std::max(pairs_cap, pairs_cap_q)) * n);
  if (!ws || !scr) return cfear_set_error(ctx, CFEAR_ERR_HIP, "workspace allocation failed");
  VerifyChain c;
  c.cand = (const VerifyDev*)(ws + o_cand); c.reg_jobs = ws + o_reg_jobs; c.cost_jobs = ws + o_cost_jobs; c.stride = stride;
  c.coral_jobs = (CoralJob*)(ws + o_coral_jobs); c.reg = (const cfear_reg_result*)(ws + o_reg); c.q = (const cfear_reg_result*)(ws + o_q);
  // results on the DEVICE (a sharded caller gathers them there): the finish kernel writes them in place, nothing but the error
  // flag comes back, and ApplyConstratins is the caller's (cfear_verify_apply_constraints over the gathered list)
  const bool dev_out = cfear_is_device_ptr(results);
  c.coral = (const cfear_coral_result*)(ws + o_coral); c.out = dev_out ? results : (cfear_verify_result*)(ws + o_out); c.first_bad = (int32_t*)(ws + o_flag);
  c.n = n_jobs; c.par = *par;
  const dim3 grid((unsigned)((n + 255) / 256)), block(256);
  // ---- the chain ---------------------------------------------------------------------------------------------------------
  CFEAR_HIP_CHECK(ctx, hipMemcpyAsync(ws + o_cand, hc, n * sizeof(VerifyDev), hipMemcpyHostToDevice, ctx->stream));
  cfear_pinned_mark(ctx);
  int rc = CFEAR_OK;
  { ProfScope ps(ctx, "verify_glue"); hipLaunchKernelGGL(verify_expand_kernel, grid, block, 0, ctx->stream, c); }
  rc = cfear_register_launch(ctx, c.reg_jobs, n_jobs, &rp, pairs_cap, scr, (cfear_reg_result*)(ws + o_reg), nullptr, stride, hint);
  if (rc == CFEAR_OK) {
    { ProfScope ps(ctx, "verify_glue"); hipLaunchKernelGGL(verify_prepare_kernel, grid, block, 0, ctx->stream, c); }
    RegCostMode mode;                                        // GetCost at the jobs' own poses
    mode.blocks_per_job = std::max(1, std::min(1, (1024 + n_jobs - 1) / n_jobs));
    rc = cfear_register_launch(ctx, c.cost_jobs, n_jobs, &qp, pairs_cap_q, scr, (cfear_reg_result*)(ws + o_q), &mode, stride, hint_q);
  }
  if (rc == CFEAR_OK) rc = cfear_coral_launch_device(ctx, c.coral_jobs, n_jobs, cap, &par->coral, (cfear_coral_result*)(ws + o_coral));
  if (rc != CFEAR_OK) { (void)hipStreamSynchronize(ctx->stream); return rc; }
  { ProfScope ps(ctx, "verify_glue"); hipLaunchKernelGGL(verify_finish_kernel, grid, block, 0, ctx->stream, c); }
  CFEAR_HIP_CHECK(ctx, hipGetLastError());
  mark(2);
  int32_t first_bad = 0x7fffffff;
  if (!dev_out) CFEAR_HIP_CHECK(ctx, hipMemcpyAsync(results, ws + o_out, n * sizeof(cfear_verify_result), hipMemcpyDeviceToHost, ctx->stream));
  CFEAR_HIP_CHECK(ctx, hipMemcpyAsync(&first_bad, ws + o_flag, 4, hipMemcpyDeviceToHost, ctx->stream));
  CFEAR_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  if (first_bad != 0x7fffffff) {
    cfear_coral_result bad;
    CFEAR_HIP_CHECK(ctx, hipMemcpy(&bad, ws + o_coral + (size_t)first_bad * sizeof(cfear_coral_result), sizeof(bad), hipMemcpyDeviceToHost));
    return cfear_set_error(ctx, bad.status, "job %d: %s", first_bad, cfear_status_string(bad.status));
  }
  mark(3);
  if (!dev_out) apply_constraints(jobs, n, par, results);
  mark(4);
#ifdef CFEAR_VERIFY_TIMING
  fprintf(stderr, "verify us: scan clouds + sizes %.0f | fill records %.0f | enqueue the chain %.0f | kernels + read-back %.0f | ApplyConstratins %.0f\n",
          t_acc[0], t_acc[1], t_acc[2], t_acc[3], t_acc[4]);
#endif
  return CFEAR_OK;
}

}  // namespace

// The chain with the host between its kernels: kept for par.use_covariance_sampling (the sampled covariance is fitted on the
// host from 27 cost samples per candidate, loopclosure.cpp:62-71, 99-208).
static int verify_host_chain(cfear_ctx* ctx, const cfear_verify_job* jobs, int32_t n_jobs,
                             const cfear_verify_params* par, cfear_verify_result* results) {
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

  apply_constraints(jobs, n, par, results);
  mark(6);
#ifdef CFEAR_VERIFY_TIMING
  fprintf(stderr, "verify us: marshal reg %.0f | register_batch %.0f | Talign + coral jobs %.0f | coral enqueue %.0f | cov + cost jobs %.0f | get_cost_batch + coral results %.0f | classify + sort %.0f\n",
          t_acc[0], t_acc[1], t_acc[2], t_acc[3], t_acc[4], t_acc[5], t_acc[6]);
#endif
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
  if (par->use_covariance_sampling) {
    if (cfear_is_device_ptr(results)) return cfear_set_error(ctx, CFEAR_ERR_INVALID_ARGUMENT, "results on the device are not available with use_covariance_sampling");
    return verify_host_chain(ctx, jobs, n_jobs, par, results);
  }
  return verify_device_chain(ctx, jobs, n_jobs, par, results);
}

extern "C" int cfear_verify_apply_constraints(const int32_t* groups, int32_t n, const cfear_verify_params* par, cfear_verify_result* results) {
  if (!par || n < 0 || (n > 0 && (!groups || !results))) return CFEAR_ERR_INVALID_ARGUMENT;
  if (n > 0) apply_constraints_groups(groups, sizeof(int32_t), (size_t)n, par, results);
  return CFEAR_OK;
}
