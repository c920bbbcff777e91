// register.hip -- stage M of the CFEAR hot path on gfx950: many-to-one scan matcher.
//
// Replaces (cfear_radarodometry/src/cfear_radarodometry/ unless noted):
//   n_scan_normal_reg::Register                        n_scan_normal.cpp:82-185
//   n_scan_normal_reg::BuildOptimizationProblem        n_scan_normal.cpp:342-389
//   n_scan_normal_reg::AddScanPairCost                 n_scan_normal.cpp:213-324
//   MapPointNormal::GetClosestIdx (FLANN 1-NN)         pointnormal.cpp:238-254
//   P2L/P2P/P2DEfficientCost (+ AutoDiff Jacobians)    include/.../n_scan_normal.h:180-361
//   Registration::Weights::GetWeight, GetLoss          registration.cpp:67-96
//   ceres::Solve (TRUST_REGION + LEVENBERG_MARQUARDT)  call site n_scan_normal.cpp:448
//   n_scan_normal_reg::GetCost                         n_scan_normal.cpp:186-211
//
// Kernel design: ONE persistent 256-thread workgroup per registration, one launch for a whole
// batch of registrations; nothing returns to the host between the first association and the
// final pose.  Per outer iteration the float means of each fixed keyframe are staged in LDS and
// every source cell finds its exact nearest neighbour by brute force (same float arithmetic as
// FLANN's L2_Simple, lowest index on ties).  Each LM iteration evaluates all correspondences at
// the candidate pose, reduces {cost, J^T r, J^T J} (10 doubles) with DPP row reductions + a
// 4-entry LDS cross-wave sum in a fixed order, and every thread then runs the Ceres-2.1-equivalent
// trust-region bookkeeping redundantly in registers (wave-uniform control flow, no broadcast).
#include <cfloat>
#include <cmath>
#include <memory>

#include "common.hpp"
#include "fastmath.hpp"

namespace {

#ifdef CFEAR_REG_TIMING   // debug build only: cycle split of workgroup 0, printed at kernel end
__device__ long long g_reg_t[32];
#define REG_T0() long long _t0 = __builtin_readcyclecounter()
#define REG_TACC(k) do { const long long _t1 = __builtin_readcyclecounter(); \
    if (threadIdx.x == 0 && blockIdx.x == 0) g_reg_t[k] += _t1 - _t0; _t0 = _t1; } while (0)
#else
#define REG_T0()
#define REG_TACC(k)
#endif

constexpr int kRegThreads = 256;
constexpr int kMaxScans = 16;
constexpr int kMaxTargetsLds = 8192;         // float2 targets staged in LDS (64 KiB)

// One registration.  The scan views come LAST so that a batch whose jobs use at most m scans can be stored with the
// shorter stride reg_job_stride(m): a two-scan loop-closure candidate is 0.6 KB instead of 2.2 KB to build and upload.
// Kernels only ever touch scans[0 .. n_scans).
struct RegJob {
  int32_t n_scans;
  int32_t itr;                                // cost-only launches: this job's leftover itr_ (0: use par.itr)
  double poses[kMaxScans][3];
  ScanView scans[kMaxScans];
};
inline size_t reg_job_stride(int max_scans) {
  return (offsetof(RegJob, scans) + (size_t)max_scans * sizeof(ScanView) + 15) & ~(size_t)15;
}

struct RegCommon {
  cfear_reg_params par;
  double angle_outlier;                       // std::cos(M_PI/6.0), computed on the host
  char* scratch;                              // per job: 6 doubles per slot
  size_t scratch_stride;
  size_t job_stride;                          // bytes between job records (reg_job_stride)
  int32_t slots_cap;
  int32_t lds_targets;                        // capacity of the staged (x-sorted) target arrays
  int32_t dense_cap_lds;                      // correspondences that fit the LDS dense arrays (slot path)
  int32_t dense_fields;                       // doubles per correspondence (3 P2P, 5 P2L, 6 P2D) + one int32
  uint32_t lds_total;                         // dynamic LDS bytes of the launch
  cfear_reg_result* results;
  // cost-only launches (GetCost / cost sampling): no solve; n_samples > 0 evaluates the samples_per_axis^3
  // pose grid of approximateCovarianceBySampling around each job's source pose
  int32_t cost_only;
  int32_t n_samples;
  int32_t samples_per_axis;
  int32_t big_mode;                           // 0: single launch | 1: first launch, registrations that only fit lds_big are deferred | 2: the deferred ones
  uint32_t lds_big;                           // dynamic LDS of the second launch (one workgroup per CU)
  int32_t only_deferred;                      // launched behind register3_kernel: only the registrations it marked kRegDeferred
  int32_t r3_take_all;                        // register3_kernel (tests, CFEAR_REG3_LDS_KB): keep every registration that can run at all
  int32_t pad2;
  double xy_half, yaw_half;
  const cfear_reg_result* prior;              // cost-only: source pose and itr_ come from these records (device)
};

__host__ __device__ inline size_t slots_bytes(int slots_cap) { return ((size_t)slots_cap * 52 + 255) / 256 * 256; }
#ifndef REG_NW
#define REG_NW 4
#endif
#ifndef REG_LDS_KB
#define REG_LDS_KB 80
#endif
constexpr int kRegNW = REG_NW;         // wavefronts per registration workgroup (register_kernel)
constexpr int kRegMaxNW = 16;
// fixed LDS: [0,2560) reduction partials [2][16][10], [2560,2688) int partials [2][16], [2688,2816) LM control
constexpr size_t kRegIpartOff = 2560, kRegCtrlOff = 2688;
constexpr size_t kRegFixedLds = 2816;
// LDS map: fixed block above, then x-sorted targets (x, y, idx:
// 12 B each), then the dense correspondence arrays (dense_fields doubles each).
__host__ __device__ inline size_t reg_lds_targets_bytes(int lds_targets) { return ((size_t)lds_targets * 12 + 15) / 16 * 16; }
size_t reg_lds_bytes(int lds_targets, int dense_cap, int dense_fields) {
  return kRegFixedLds + reg_lds_targets_bytes(lds_targets) + (size_t)dense_cap * (dense_fields * 8 + 4);
}
constexpr size_t kRegLdsBudget = REG_LDS_KB * 1024 - 256;    // keeps 2 workgroups per CU (160 KiB LDS)
// Compact geometry for small registrations (a two-scan loop-closure candidate needs ~37 KB): 2 wavefronts and 40 KB per
// workgroup, so four registrations share a CU instead of two.  The work of one registration is a chain of short
// dependent phases, so halving its lanes costs little latency while doubling the registrations in flight.
constexpr int kRegDeferred = 1000;            // internal status between the two launches of a batch with large registrations
constexpr int kRegNWBig = 8;                  // wavefronts of the second launch's workgroups (one per CU)
constexpr int kRegNWCompact = 2;
constexpr size_t kRegLdsBudgetCompact = 40 * 1024 - 256;
// Measured on MI355X (round 1): with ~190 VGPRs only 2 wavefronts fit a SIMD, so the wave-per-job
// geometry is latency-bound on its 4x longer per-lane loops (4096 jobs: 6.7 ms vs 6.0 ms); disabled.
int reg_dense_fields(int cost) { return cost == CFEAR_P2P ? 3 : (cost == CFEAR_P2L ? 5 : 6); }

struct Aff2 { double l0, l1, l2, l3, t0, t1; };

// registration.cpp:128-135 vectorToAffine3d + n_scan_normal.cpp:350-351
__device__ __forceinline__ Aff2 aff_from_xyt(const double* p) {
  double s, c;
  sincos(p[2], &s, &c);
  return Aff2{c, -s, s, c, p[0], p[1]};
}
__device__ __forceinline__ Aff2 aff_mul(const Aff2& a, const Aff2& b) {
  Aff2 r;
  r.l0 = a.l0 * b.l0 + a.l1 * b.l2;
  r.l1 = a.l0 * b.l1 + a.l1 * b.l3;
  r.l2 = a.l2 * b.l0 + a.l3 * b.l2;
  r.l3 = a.l2 * b.l1 + a.l3 * b.l3;
  r.t0 = a.l0 * b.t0 + a.l1 * b.t1 + a.t0;
  r.t1 = a.l2 * b.t0 + a.l3 * b.t1 + a.t1;
  return r;
}
__device__ __forceinline__ Aff2 aff_inv(const Aff2& a) {   // Eigen Affine inverse: adjugate / det
  const double det = a.l0 * a.l3 - a.l2 * a.l1;
  const double invdet = 1.0 / det;
  Aff2 r;
  r.l0 = a.l3 * invdet; r.l1 = -a.l1 * invdet; r.l2 = -a.l2 * invdet; r.l3 = a.l0 * invdet;
  r.t0 = -(r.l0 * a.t0 + r.l1 * a.t1);
  r.t1 = -(r.l2 * a.t0 + r.l3 * a.t1);
  return r;
}

__device__ __forceinline__ double similarity(double x, double y) { return 2 * fmin(x, y) / (x + y); }
// registration.cpp:67-75
__device__ __forceinline__ double get_weight(int opt, double N1, double N2, double sim_dir, double plan1, double plan2) {
  switch (opt) {
    case 0: return 1.0;
    case 1: return similarity(N1, N2);
    case 2: return sim_dir;
    case 3: return similarity(plan1, plan2);
    case 4: return similarity(N1, N2) + sim_dir + similarity(plan1, plan2);
  }
  return 1.0;
}

// ceres loss functions wrapped by ScaledLoss(loss, w) (registration.cpp:77-96, n_scan_normal.cpp:275)
__device__ __forceinline__ void loss_eval(int loss, double a, double w, double s, double& rho0, double& rho1) {
  const double dmin = DBL_MIN;
  switch (loss) {
    case 1: {  // Huber: rho = 2 a sqrt(s) - a^2, rho' = a / sqrt(s) for s > a^2
      const double b = a * a;
      if (s > b) {
        // one reciprocal square root instead of sqrt + divide (r = s * rsqrt(s), a / r = a * rsqrt(s));
        // agrees with ceres::HuberLoss to rounding
        const double q = rsqrt(s);
        rho0 = 2.0 * a * (s * q) - b; rho1 = fmax(dmin, a * q);
      } else { rho0 = s; rho1 = 1.0; }
      break;
    }
    case 2: {  // Cauchy
      const double b = a * a, c = 1.0 / b;
      const double sum = 1.0 + s * c, inv = 1.0 / sum;
      rho0 = b * log(sum); rho1 = fmax(dmin, inv);
      break;
    }
    case 3: {  // SoftLOne
      const double b = a * a, c = 1.0 / b;
      const double sum = 1.0 + s * c, tmp = sqrt(sum);
      rho0 = 2.0 * b * (tmp - 1.0); rho1 = fmax(dmin, 1.0 / tmp);
      break;
    }
    case 4: {  // ComposedLoss(Huber(1), Cauchy(1)) = f(g(s))
      const double sum = 1.0 + s, inv = 1.0 / sum;
      const double g0 = log(sum), g1 = fmax(dmin, inv);
      double f0, f1;
      if (g0 > 1.0) { const double r = sqrt(g0); f0 = 2.0 * r - 1.0; f1 = fmax(dmin, 1.0 / r); }
      else { f0 = g0; f1 = 1.0; }
      rho0 = f0; rho1 = f1 * g1;
      break;
    }
    case 5: {  // Tukey
      const double a2 = a * a;
      if (s <= a2) { const double value = 1.0 - s / a2, vs = value * value; rho0 = a2 / 3.0 * (1.0 - vs * value); rho1 = vs; }
      else { rho0 = a2 / 3.0; rho1 = 0.0; }
      break;
    }
    default: rho0 = s; rho1 = 1.0;
  }
  rho0 *= w; rho1 *= w;
}

struct Slots {          // per-job correspondence arrays (global scratch, SoA, index = slot)
  double *tmx, *tmy, *a0, *a1, *a2, *w;
  int32_t* tidx;        // matched target cell
};
__device__ __forceinline__ Slots slots_of(char* scratch, int cap) {
  Slots s;
  double* p = (double*)scratch;
  s.tmx = p; s.tmy = p + cap; s.a0 = p + 2 * (size_t)cap; s.a1 = p + 3 * (size_t)cap;
  s.a2 = p + 4 * (size_t)cap; s.w = p + 5 * (size_t)cap;
  s.tidx = (int32_t*)(p + 6 * (size_t)cap);
  return s;
}

// 10 accumulators: cost, g[3], H upper triangle (00,01,02,11,12,22)
// NW = wavefronts per registration: 4 (one 256-thread workgroup per job, lowest latency) or 1 (one
// wavefront per job: no barriers or LDS exchange at all, 4x more jobs in flight -- large batches).
// ALL = false: only wavefront 0 (the LM bookkeeping wavefront) receives the totals.
template <int NW, bool ALL = true>
__device__ __forceinline__ void block_reduce10(double v[10], double* part /*[2][16][10]*/, int& phase) {
  if (NW == 1) {
#pragma unroll
    for (int k = 0; k < 10; k++) v[k] = wave_sum_f64(v[k]);     // already wave-uniform (readlane)
    return;
  }
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  double* buf = part + phase * (kRegMaxNW * 10);
  if (NW == 4) {
    // Value-splitting butterfly inside each 16-lane DPP row: a step that pairs lanes l and P(l) lets
    // one class of lanes keep value p and the other value q of a pair, so every step halves the
    // number of live values (10 -> 5 -> 3, then two plain steps): 56 instructions instead of 180.
    // bank_mask performs the class select (banks = lane quads): row_mirror splits on lane bit 3
    // (banks 0,1 | 2,3), row_half_mirror on bit 2 (banks 0,2 | 1,3).
    double v1[5];
#pragma unroll
    for (int j = 0; j < 5; j++)
      v1[j] = dpp_sel_f64<0x140, 0x3>(v[j + 5], v[j]) + dpp_sel_f64<0x140, 0xC>(v[j], v[j + 5]);
    double w0 = dpp_sel_f64<0x141, 0x5>(v1[1], v1[0]) + dpp_sel_f64<0x141, 0xA>(v1[0], v1[1]);
    double w1 = dpp_sel_f64<0x141, 0x5>(v1[3], v1[2]) + dpp_sel_f64<0x141, 0xA>(v1[2], v1[3]);
    double w2 = v1[4] + dpp_f64<0x141>(v1[4]);
    w0 += dpp_f64<0x4E>(w0); w1 += dpp_f64<0x4E>(w1); w2 += dpp_f64<0x4E>(w2);
    w0 += dpp_f64<0xB1>(w0); w1 += dpp_f64<0xB1>(w1); w2 += dpp_f64<0xB1>(w2);
    // the quad with lane bits (b3, b2) now holds the row sums of k = b2 + 5 b3 (w0), 2 + b2 + 5 b3 (w1)
    // and 4 + 5 b3 (w2); partial (k, wave, row) goes to buf[k * 16 + wave * 4 + row]
    const int row = lane >> 4, b3 = (lane >> 3) & 1, b2 = (lane >> 2) & 1;
    const int slot = wave * 4 + row;
    if ((lane & 3) == 0) {
      buf[(b2 + 5 * b3) * 16 + slot] = w0;
      buf[(2 + b2 + 5 * b3) * 16 + slot] = w1;
      if (b2 == 0) buf[(4 + 5 * b3) * 16 + slot] = w2;
    }
    __syncthreads();
    if (ALL || wave == 0) {
      // 16 partials per k = one DPP row per k: three registers cover the 160 partials; the totals are
      // read back with readlane, so they are wave-uniform (scalar branches downstream)
      double r[3];
#pragma unroll
      for (int q = 0; q < 3; q++) {
        const int j = q * 64 + lane;
        r[q] = row16_sum_f64(j < 160 ? buf[j] : 0.0);
      }
#pragma unroll
      for (int k = 0; k < 10; k++) v[k] = readlane_f64(r[k / 4], (k % 4) * 16);
    }
    phase ^= 1;
    return;
  }
  // generic NW: k-major buf[k * NW + wave]
#pragma unroll
  for (int k = 0; k < 10; k++) {
    const double t = wave_sum_lane63_f64(v[k]);
    if (lane == 63) buf[k * NW + wave] = t;
  }
  __syncthreads();
  if (ALL || wave == 0) {
    // lane-parallel cross-wave sum: partial j = k * NW + wave sits in lane j % 64 of register j / 64;
    // aligned groups of NW lanes are summed by a DPP butterfly (fixed order)
    constexpr int NR = (10 * NW + 63) / 64;
    double r[NR];
#pragma unroll
    for (int q = 0; q < NR; q++) {
      const int j = q * 64 + lane;
      double t = j < 10 * NW ? buf[j] : 0.0;
      if (NW >= 2) t += dpp_f64<0xB1>(t);      // quad_perm [1,0,3,2]
      if (NW >= 4) t += dpp_f64<0x4E>(t);      // quad_perm [2,3,0,1]
      if (NW >= 8) t += dpp_f64<0x141>(t);     // row_half_mirror
      if (NW >= 16) t += dpp_f64<0x140>(t);    // row_mirror
      r[q] = t;
    }
#pragma unroll
    for (int k = 0; k < 10; k++) v[k] = readlane_f64(r[(k * NW) / 64], (k * NW) % 64);
  }
  phase ^= 1;
}
template <int NW>
__device__ __forceinline__ int block_sum_i32(int v, int* part /*[2][4]*/, int& phase) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int t = wave_sum_i32(v);
  if (NW == 1) { __syncthreads(); return t; }
  int* buf = part + phase * kRegMaxNW;
  if (lane == 0) buf[wave] = t;
  __syncthreads();
  int r = buf[0];
#pragma unroll
  for (int wv = 1; wv < NW; wv++) r += buf[wv];
  r = __builtin_amdgcn_readfirstlane(r);
  phase ^= 1;
  return r;
}

// n_scan_normal.cpp:213-318 for every fixed keyframe i against the free source (last scan):
// fills the slot arrays; returns this thread's number of accepted associations.
struct LdsTargets { float* x; float* y; int* idx; };

template <int NW>
__device__ int associate_all(const RegJob& job, const RegCommon& cm, const double* xsrc, int itr, const Slots& sl,
                             const LdsTargets& lt) {
  const int tid = threadIdx.x;
  const int last = job.n_scans - 1;
  const ScanView& src = job.scans[last];
  const int n_src = *src.n_cells;
  const double curr_radius = (itr == 1) ? 2 * cm.par.radius : cm.par.radius;    // :220
  const double r2 = curr_radius * curr_radius;
  const float rwin = (float)curr_radius + 1e-3f;        // window half-width (slightly generous)
  const Aff2 Tsrc = aff_from_xyt(xsrc);
  int accepted = 0;
  for (int i = 0; i < last; i++) {
    const ScanView& tar = job.scans[i];
    const int n_tar = *tar.n_cells;
    const Aff2 Ttar = aff_from_xyt(job.poses[i]);
    const Aff2 Tst = aff_mul(aff_inv(Ttar), Tsrc);                              // :222
    __syncthreads();                                     // previous keyframe's LDS readers are done
    for (int j = tid; j < n_tar; j += NW * 64) {          // x-sorted float means (+ original index)
      lt.x[j] = tar.sorted_x[j];
      lt.y[j] = tar.sorted_y[j];
      lt.idx[j] = tar.sorted_idx[j];
    }
    __syncthreads();
    for (int s = tid; s < n_src; s += NW * 64) {
      const int slot = i * n_src + s;
      const double2 u = src.mean[s];
      const double px = Tst.l0 * u.x + Tst.l1 * u.y + Tst.t0;
      const double py = Tst.l2 * u.x + Tst.l3 * u.y + Tst.t1;
      const float qx = (float)px, qy = (float)py;                               // pointnormal.cpp:240-242
      // Exact 1-NN with FLANN's L2_Simple float distance, lowest index on ties.  Only targets with
      // |x - qx| <= radius can pass the `dist < radius^2` gate (pointnormal.cpp:250), so the search is
      // restricted to that window of the x-sorted order; the result equals the brute-force scan.
      const float xlo = qx - rwin, xhi = qx + rwin;
      int lo = 0, hi = n_tar;
      while (lo < hi) { const int mid = (lo + hi) >> 1; if (lt.x[mid] < xlo) lo = mid + 1; else hi = mid; }
      int best = -1;
      float bestd = FLT_MAX;
      for (int p = lo; p < n_tar; p++) {
        const float tx = lt.x[p];
        if (tx > xhi) break;
        const float dx = __fsub_rn(qx, tx), dy = __fsub_rn(qy, lt.y[p]);
        const float d = __fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy));
        const int idx = lt.idx[p];
        if (d < bestd || (d == bestd && idx < best)) { best = idx; bestd = d; }
      }
      double w = -1.0;
      if (best >= 0 && (double)bestd < r2) {                                    // pointnormal.cpp:250
        const double2 ns = src.normal[s];
        const double2 nt = tar.normal[best];
        const double nsx = Tst.l0 * ns.x + Tst.l1 * ns.y, nsy = Tst.l2 * ns.x + Tst.l3 * ns.y;
        const double direction_similarity = fmax(nsx * nt.x + nsy * nt.y, 0.0);   // :244
        if (direction_similarity > cm.angle_outlier) {                            // :245
          w = get_weight(cm.par.weight_opt, (double)src.nsamples[s], (double)tar.nsamples[best],
                         direction_similarity, src.scale[s], tar.scale[best]);   // :247-253, :273
          const double2 tm = tar.mean[best];
          sl.tmx[slot] = Ttar.l0 * tm.x + Ttar.l1 * tm.y + Ttar.t0;              // Ttar * tar_mean
          sl.tmy[slot] = Ttar.l2 * tm.x + Ttar.l3 * tm.y + Ttar.t1;
          if (cm.par.cost == CFEAR_P2D) {                                         // :288-297
            const double4 S = tar.cov[best];
            const double a00 = Ttar.l0 * S.x + Ttar.l1 * S.z, a01 = Ttar.l0 * S.y + Ttar.l1 * S.w;
            const double a10 = Ttar.l2 * S.x + Ttar.l3 * S.z, a11 = Ttar.l2 * S.y + Ttar.l3 * S.w;
            const double c00 = (cm.par.regularization + (a00 * Ttar.l0 + a01 * Ttar.l1)) * cm.par.cov_scale;
            const double c01 = (0.0 + (a00 * Ttar.l2 + a01 * Ttar.l3)) * cm.par.cov_scale;
            const double c10 = (0.0 + (a10 * Ttar.l0 + a11 * Ttar.l1)) * cm.par.cov_scale;
            const double c11 = (cm.par.regularization + (a10 * Ttar.l2 + a11 * Ttar.l3)) * cm.par.cov_scale;
            const double det = c00 * c11 - c10 * c01, invdet = 1.0 / det;
            const double i00 = c11 * invdet, i10 = -c10 * invdet, i11 = c00 * invdet;
            const double l00 = sqrt(i00), l10 = i10 / l00;
            sl.a0[slot] = l00; sl.a1[slot] = l10; sl.a2[slot] = sqrt(i11 - l10 * l10);
          } else {
            sl.a0[slot] = Ttar.l0 * nt.x + Ttar.l1 * nt.y;                       // Ttar.linear() * tar_normal
            sl.a1[slot] = Ttar.l2 * nt.x + Ttar.l3 * nt.y;
          }
          sl.tidx[slot] = best;
          accepted++;
        }
      }
      sl.w[slot] = w;
    }
  }
  return accepted;
}

// Residual block of one slot at pose x (c = cos th, s = sin th): adds to acc[10].
// COST / LOSS are compile-time (LOSS = -1: runtime switch) so the hot P2P/P2L + Huber kernels carry no
// per-correspondence branching and constant Jacobian entries fold away.  The accumulations use fma():
// these sums are already reduced in a different order than Ceres', agreement is to rounding either way.
template <int COST, int LOSS, bool WITH_JAC>
__device__ __forceinline__ void eval_slot(const cfear_reg_params& par, double smx, double smy, double tmx, double tmy,
                                          double a0, double a1, double a2, double w, double tx, double ty, double c,
                                          double s, double acc[10]) {
  // R s is shared by the transformed point and its derivative: -s smx - c smy == -(s smx + c smy) and
  // c smx - s smy == c smx + (-s) smy bit for bit, so the reference's four expressions need two.
  const double ru = c * smx + (-s) * smy, rv = s * smx + c * smy;
  const double sx = ru + tx;                            // n_scan_normal.h:194-197
  const double sy = rv + ty;
  const double dx = -rv, dy = ru;                       // d(R s)/dtheta
  double r0, r1 = 0.0, j00, j01, j02, j10 = 0.0, j11 = 0.0, j12 = 0.0;
  if (COST == CFEAR_P2L) {                              // n_scan_normal.h:180-213
    const double v0 = sx - tmx, v1 = sy - tmy;
    r0 = v0 * a0 + v1 * a1;
    j00 = a0; j01 = a1; j02 = dx * a0 + dy * a1;
  } else if (COST == CFEAR_P2P) {                       // n_scan_normal.h:330-361
    r0 = tmx - sx; r1 = tmy - sy;
    j00 = -1.0; j01 = 0.0; j02 = -dx; j10 = 0.0; j11 = -1.0; j12 = -dy;
  } else {                                              // n_scan_normal.h:216-255, L = [a0 0; a1 a2]
    const double v0 = sx - tmx, v1 = sy - tmy;
    r0 = a0 * v0 + 0.0 * v1;
    r1 = a1 * v0 + a2 * v1;
    j00 = a0; j01 = 0.0; j02 = a0 * dx + 0.0 * dy;
    j10 = a1; j11 = a2; j12 = a1 * dx + a2 * dy;
  }
  const double sq = (COST == CFEAR_P2L) ? r0 * r0 : (r0 * r0 + r1 * r1);
  double rho0, rho1;
  loss_eval(LOSS >= 0 ? LOSS : par.loss, par.loss_limit, w, sq, rho0, rho1);
  acc[0] = fma(0.5, rho0, acc[0]);
  if (WITH_JAC && COST == CFEAR_P2P) {
    // J = [-1 0 -dx; 0 -1 -dy]: the products with the constant entries are exact (x * -1, x * 0), so
    // the generic accumulation below reduces to these terms; acc[5] (H01) stays exactly zero.
    const double g0 = rho1 * r0, g1 = rho1 * r1;
    acc[1] -= g0; acc[2] -= g1;
    acc[3] = fma(j02, g0, acc[3]); acc[3] = fma(j12, g1, acc[3]);
    const double h02 = rho1 * j02, h12 = rho1 * j12;
    acc[4] += rho1; acc[6] = fma(-rho1, j02, acc[6]); acc[9] = fma(h02, j02, acc[9]);
    acc[7] += rho1; acc[8] = fma(-rho1, j12, acc[8]); acc[9] = fma(h12, j12, acc[9]);
  } else if (WITH_JAC) {
    // Corrector with alpha = 0 scales residual and Jacobian rows by sqrt(rho'); the normal equations
    // only need the products, (sqrt(rho') J)^T (sqrt(rho') r) = rho' J^T r, so no square root here.
    const double g0 = rho1 * r0;
    acc[1] = fma(j00, g0, acc[1]); acc[2] = fma(j01, g0, acc[2]); acc[3] = fma(j02, g0, acc[3]);
    const double h00 = rho1 * j00, h01 = rho1 * j01, h02 = rho1 * j02;
    acc[4] = fma(h00, j00, acc[4]); acc[5] = fma(h00, j01, acc[5]); acc[6] = fma(h00, j02, acc[6]);
    acc[7] = fma(h01, j01, acc[7]); acc[8] = fma(h01, j02, acc[8]); acc[9] = fma(h02, j02, acc[9]);
    if (COST != CFEAR_P2L) {
      const double g1 = rho1 * r1;
      acc[1] = fma(j10, g1, acc[1]); acc[2] = fma(j11, g1, acc[2]); acc[3] = fma(j12, g1, acc[3]);
      const double h10 = rho1 * j10, h11 = rho1 * j11, h12 = rho1 * j12;
      acc[4] = fma(h10, j10, acc[4]); acc[5] = fma(h10, j11, acc[5]); acc[6] = fma(h10, j12, acc[6]);
      acc[7] = fma(h11, j11, acc[7]); acc[8] = fma(h11, j12, acc[8]); acc[9] = fma(h12, j12, acc[9]);
    }
  }
}

// runtime-cost dispatch for the non-hot callers (eval_kernel)
template <bool WITH_JAC>
__device__ __forceinline__ void eval_slot_rt(const cfear_reg_params& par, double smx, double smy, double tmx, double tmy,
                                             double a0, double a1, double a2, double w, double tx, double ty, double c,
                                             double s, double acc[10]) {
  if (par.cost == CFEAR_P2L) eval_slot<CFEAR_P2L, -1, WITH_JAC>(par, smx, smy, tmx, tmy, a0, a1, a2, w, tx, ty, c, s, acc);
  else if (par.cost == CFEAR_P2P) eval_slot<CFEAR_P2P, -1, WITH_JAC>(par, smx, smy, tmx, tmy, a0, a1, a2, w, tx, ty, c, s, acc);
  else eval_slot<CFEAR_P2D, -1, WITH_JAC>(par, smx, smy, tmx, tmy, a0, a1, a2, w, tx, ty, c, s, acc);
}

// Dense correspondence arrays (SoA, stride cap): doubles 0 tmx, 1 tmy, 2 w, 3 a0, 4 a1, 5 a2, then the
// int32 source-cell index (the source mean is read through it instead of being copied per block).
// They live in LDS when they fit, otherwise in the job's global scratch; the pointers are generic, so
// one code path serves both.
struct Dense { double* p; int* sidx; const double2* smean; int cap; int n; };
__device__ __forceinline__ void dense_bind(Dense& dn, double* base, int cap, int fields, const double2* smean) {
  dn.p = base; dn.cap = cap; dn.sidx = (int*)(base + (size_t)fields * cap); dn.smean = smean;
}

// cost, gradient and Gauss-Newton matrix of all correspondences at x (block-wide collective)
template <int NW, int COST, int LOSS>
__device__ void eval_all(const RegCommon& cm, const Dense& dn, const double x[3], double c, double s, double out[10],
                         double* part, int& phase) {
  REG_T0();
  double acc[10];
#pragma unroll
  for (int k = 0; k < 10; k++) acc[k] = 0.0;
  const size_t cap = (size_t)dn.cap;
  REG_TACC(7);
  for (int i = threadIdx.x; i < dn.n; i += NW * 64) {
    const double2 sm = dn.smean[dn.sidx[i]];
    const double tmx = dn.p[i], tmy = dn.p[cap + i], w = dn.p[2 * cap + i];
    double a0 = 0.0, a1 = 0.0, a2 = 0.0;
    if (COST != CFEAR_P2P) { a0 = dn.p[3 * cap + i]; a1 = dn.p[4 * cap + i]; }
    if (COST == CFEAR_P2D) a2 = dn.p[5 * cap + i];
    eval_slot<COST, LOSS, true>(cm.par, sm.x, sm.y, tmx, tmy, a0, a1, a2, w, x[0], x[1], c, s, acc);
  }
  REG_TACC(4);
  block_reduce10<NW, false>(acc, part, phase);
  REG_TACC(5);
#pragma unroll
  for (int k = 0; k < 10; k++) out[k] = acc[k];
}

// Gathers the accepted slots (thread-major order, deterministic) into the dense arrays.
template <int NW>
__device__ int compact_slots(const RegJob& job, const RegCommon& cm, const Slots& sl, int n_slots, int n_src, int mine,
                             double* lds_dense, double* gl_dense, Dense& dn, int* ipart, int& iphase) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int incl = wave_incl_scan_i32(mine);
  int base = incl - mine, total;
  if (NW == 1) {
    __syncthreads();                                     // every slot array is complete
    total = __builtin_amdgcn_readlane(incl, 63);
  } else {
    int* buf = ipart + iphase * kRegMaxNW;
    if (lane == 63) buf[wave] = incl;
    __syncthreads();                                     // also: every slot array is complete
    for (int wv = 0; wv < wave; wv++) base += buf[wv];
    int tt = buf[0];
#pragma unroll
    for (int wv = 1; wv < NW; wv++) tt += buf[wv];
    total = __builtin_amdgcn_readfirstlane(tt);
    iphase ^= 1;
  }
  const bool in_lds = total <= cm.dense_cap_lds;
  dense_bind(dn, in_lds ? lds_dense : gl_dense, in_lds ? cm.dense_cap_lds : cm.slots_cap, cm.dense_fields,
             job.scans[job.n_scans - 1].mean);
  dn.n = total;
  const size_t cap = (size_t)dn.cap;
  int c = base;
  // same visiting order as associate_all: keyframes outer, this thread's source cells inner
  const int last = job.n_scans - 1;
  for (int i = 0; i < last; i++)
    for (int s = threadIdx.x; s < n_src; s += NW * 64) {
      const int slot = i * n_src + s;
      const double w = sl.w[slot];
      if (w < 0.0) continue;
      dn.sidx[c] = s;
      dn.p[c] = sl.tmx[slot]; dn.p[cap + c] = sl.tmy[slot];
      dn.p[2 * cap + c] = w;
      if (cm.par.cost != CFEAR_P2P) { dn.p[3 * cap + c] = sl.a0[slot]; dn.p[4 * cap + c] = sl.a1[slot]; }
      if (cm.par.cost == CFEAR_P2D) dn.p[5 * cap + c] = sl.a2[slot];
      c++;
    }
  __syncthreads();
  (void)n_slots;
  return total;
}

// ---------------------------------------------------------------------------------------------------
// Fused association for register_kernel: everything the association touches repeatedly is staged in
// LDS once per registration (x-sorted float means of ALL fixed keyframes, the source cells, the
// keyframe transforms); each outer iteration then runs two passes over the flattened (keyframe,
// source cell) pairs -- match + normal gate, block scan, then gather of the matched target's
// attributes straight into the dense correspondence arrays -- with no global scratch in between.
// ---------------------------------------------------------------------------------------------------
constexpr int kGridCellsTotal = 4096;   // grid cells of all keyframes of one registration (LDS cell table)
constexpr int kGridMaxDim = 32;         // cells per axis and keyframe: min(32, floor(sqrt(4096 / keyframes)))

struct FusedLds {
  double* kf;          // [16][12]: Ttar (l0..l3,t0,t1), Tst (l0..l3,t0,t1)
  int* koff;           // [17] prefix of target counts
  float4* txyi;        // [sum targets] (x, y, index-as-int-bits, -) grouped by (keyframe, grid cell): one 16-byte LDS read per candidate
  float2* txy;         // packed form (the 8-wavefront kernel: large registrations): (x, y) here and the index as u16 in tix --
  unsigned short* tix; // 10 bytes per target instead of 16, so 1.6 x the cells fit the CU's LDS
  const void** tptr;   // [16][5] per keyframe: mean, normal, nsamples, scale, cov arrays (global pointers)
  float4* ggeo;        // [16] per keyframe grid: (x0, y0, cells per metre, -)
  unsigned* gext;      // [16][4] order-preserving uint images of the y extent (min, max) while the grid is built
  unsigned short* cstart;   // [keyframes * G * G + 1] first target (absolute index into txyi) of every grid cell
  unsigned* ccnt;      // [keyframes * G * G] build-time counters (aliases the dense arrays)
  int G;               // grid cells per axis
  double2* smean;      // [n_src] source means (read by every evaluation); the source normals / scales / sample
                       // counts are read from global memory next to the target attributes (accepted pairs only)
  unsigned short* match;   // [n_pairs] matched target (index inside its keyframe, < kMaxTargetsLds), 0xFFFF = none
  double* dense;       // rest
  int dense_cap;
};

__host__ __device__ inline int reg_grid_dim(int keyframes) {
  int g = kGridMaxDim;
  while (g > 1 && g * g * keyframes > kGridCellsTotal) g--;
  return g;
}

// Carves the workgroup's dynamic LDS (bytes `lds_total`) for the actual sizes; returns false when the
// fixed parts leave no room (the caller then uses the slot-array path).
__device__ __forceinline__ bool fused_carve(uint8_t* smem, size_t lds_total, int last, int sum_tar, int n_src, int n_pairs,
                                            int fields, FusedLds& f, bool packed) {
  size_t off = kRegFixedLds;
  f.kf = (double*)(smem + off); off += 16 * 12 * 8;
  f.koff = (int*)(smem + off); off += 80;
  f.tptr = (const void**)(smem + off); off += 16 * 5 * 8;
  f.ggeo = (float4*)(smem + off); off += 16 * 16;
  f.gext = (unsigned*)(smem + off); off += 16 * 4 * 4;
  f.G = reg_grid_dim(last);
  const size_t n_cells = (size_t)last * f.G * f.G;
  f.cstart = (unsigned short*)(smem + off); off += ((n_cells + 1) * 2 + 15) & ~(size_t)15;
  off = (off + 15) & ~(size_t)15;
  f.txyi = (float4*)(smem + off); f.txy = (float2*)(smem + off);
  if (packed) {
    off += (size_t)sum_tar * 8;
    f.tix = (unsigned short*)(smem + off); off += (((size_t)sum_tar * 2 + 15) & ~(size_t)15);
  } else {
    f.tix = nullptr; off += (size_t)sum_tar * 16;
  }
  const size_t ns = ((size_t)n_src + 1) & ~(size_t)1;
  f.smean = (double2*)(smem + off); off += ns * 16;
  f.match = (unsigned short*)(smem + off); off += (((size_t)n_pairs + 7) & ~(size_t)7) * 2;
  off = (off + 15) & ~(size_t)15;
  if (sum_tar > 65535 || off + n_cells * 4 + 1024 > lds_total) return false;   // cell table is u16; counters alias dense
  f.dense = (double*)(smem + off);
  f.ccnt = (unsigned*)(smem + off);
  f.dense_cap = (int)((lds_total - off) / ((size_t)fields * 8 + 4)) & ~1;   // even: keeps the int array 8-byte aligned
  return true;
}

template <int NW>
__device__ void fused_stage(const RegJob& job, const FusedLds& f, float cm_radius) {
  const int tid = threadIdx.x, last = job.n_scans - 1;
  REG_T0();
  if (tid == 0) {
    int acc = 0;
    for (int i = 0; i < last; i++) { f.koff[i] = acc; acc += *job.scans[i].n_cells; }
    f.koff[last] = acc;
  }
  if (tid < last) {                                      // Ttar_i = vectorToAffine(pose_i), fixed
    const Aff2 T = aff_from_xyt(job.poses[tid]);
    double* k = f.kf + tid * 12;
    k[0] = T.l0; k[1] = T.l1; k[2] = T.l2; k[3] = T.l3; k[4] = T.t0; k[5] = T.t1;
    const ScanView& tv = job.scans[tid];
    const void** tp = f.tptr + tid * 5;
    tp[0] = tv.mean; tp[1] = tv.normal; tp[2] = tv.nsamples; tp[3] = tv.scale; tp[4] = tv.cov;
  }
  __syncthreads();
  REG_TACC(8);
  // ---- uniform grid per keyframe (replaces the reference's kd-tree): targets grouped by cell -------------
  // The float means cluster along walls, so a window in one coordinate alone can hold a hundred candidates;
  // with cells no smaller than the search radius a query touches the few cells its square overlaps.
  const int G = f.G, GG = G * G, sum_tar = f.koff[last];
  auto ordered = [](float v) { unsigned u = __float_as_uint(v); return (u >> 31) ? ~u : (u | 0x80000000u); };
  auto unordered = [](unsigned u) { return __uint_as_float((u >> 31) ? (u & 0x7fffffffu) : ~u); };
  if (tid < last) { f.gext[tid * 4] = 0xFFFFFFFFu; f.gext[tid * 4 + 1] = 0u; }
  for (int c = tid; c < last * GG; c += NW * 64) f.ccnt[c] = 0;
  __syncthreads();
  REG_TACC(9);
  auto keyframe_of = [&](int t, int& i, int& j) {        // merged target index -> (keyframe, x-sorted position)
    i = 0;
    while (i + 1 < last && t >= f.koff[i + 1]) i++;
    j = t - f.koff[i];
  };
  // every thread keeps its (up to kStageRegs) targets in registers across the three sweeps below
  constexpr int kStageRegs = 8;
  float tx[kStageRegs], ty[kStageRegs];
  int tidx[kStageRegs], tkf[kStageRegs];
#pragma unroll
  for (int k = 0; k < kStageRegs; k++) {
    const int t = tid + k * NW * 64;
    tkf[k] = -1; tx[k] = ty[k] = 0.f; tidx[k] = 0;
    if (t < sum_tar) {
      int i, j;
      keyframe_of(t, i, j);
      const ScanView& tar = job.scans[i];
      tkf[k] = i; tx[k] = tar.sorted_x[j]; ty[k] = tar.sorted_y[j]; tidx[k] = tar.sorted_idx[j];
    }
  }
  auto for_targets = [&](auto&& fn) {                     // fn(keyframe, x, y, idx) over this thread's targets
#pragma unroll
    for (int k = 0; k < kStageRegs; k++)
      if (tkf[k] >= 0) fn(tkf[k], tx[k], ty[k], tidx[k]);
    for (int t = tid + kStageRegs * NW * 64; t < sum_tar; t += NW * 64) {   // oversize jobs: re-read
      int i, j;
      keyframe_of(t, i, j);
      const ScanView& tar = job.scans[i];
      fn(i, tar.sorted_x[j], tar.sorted_y[j], tar.sorted_idx[j]);
    }
  };
  for_targets([&](int i, float, float y, int) {           // y extent (x is already sorted: first / last entry)
    const unsigned u = ordered(y);
    atomicMin(&f.gext[i * 4], u);
    atomicMax(&f.gext[i * 4 + 1], u);
  });
  __syncthreads();
  REG_TACC(10);
  if (tid < last) {
    const ScanView& tar = job.scans[tid];
    const int n = f.koff[tid + 1] - f.koff[tid];
    float x0 = 0.f, y0 = 0.f, span = 0.f;
    if (n > 0) {
      x0 = tar.sorted_x[0];
      y0 = unordered(f.gext[tid * 4]);
      span = fmaxf(tar.sorted_x[n - 1] - x0, unordered(f.gext[tid * 4 + 1]) - y0);
    }
    // cell edge: the extent split into G cells, never below the matcher's radius
    const float edge = fmaxf(span / (float)G * 1.0001f, fmaxf(cm_radius, 1e-3f));
    f.ggeo[tid] = make_float4(x0, y0, 1.0f / edge, 0.f);
  }
  __syncthreads();
  REG_TACC(11);
  auto cell_of = [&](const float4 g, float x, float y) {
    const int cx = min(G - 1, max(0, (int)floorf((x - g.x) * g.z)));
    const int cy = min(G - 1, max(0, (int)floorf((y - g.y) * g.z)));
    return cy * G + cx;
  };
  for_targets([&](int i, float x, float y, int) {         // histogram of the cells
    atomicAdd(&f.ccnt[i * GG + cell_of(f.ggeo[i], x, y)], 1u);
  });
  __syncthreads();
  REG_TACC(12);
  {                                                      // exclusive scan over (keyframe, cell): absolute starts
    const int C = last * GG, per = (C + NW * 64 - 1) / (NW * 64);
    const int c0 = tid * per;
    int tot = 0;
    for (int c = c0; c < min(C, c0 + per); c++) tot += (int)f.ccnt[c];
    int* ip = (int*)(f.gext + 32);                       // [NW] wave totals (second half of the extent array)
    const int incl = wave_incl_scan_i32(tot);
    if ((tid & 63) == 63) ip[tid >> 6] = incl;
    __syncthreads();
  REG_TACC(13);
    int run = incl - tot;
    for (int wv = 0; wv < (tid >> 6); wv++) run += ip[wv];
    for (int c = c0; c < min(C, c0 + per); c++) {
      const int cnt = (int)f.ccnt[c];
      f.cstart[c] = (unsigned short)run;
      f.ccnt[c] = (unsigned)run;                         // becomes the scatter cursor
      run += cnt;
    }
    if (tid == 0) f.cstart[C] = (unsigned short)sum_tar;
  }
  __syncthreads();
  REG_TACC(14);
  for_targets([&](int i, float x, float y, int idx) {     // scatter (order inside a cell is irrelevant: the NN
    const unsigned pos = atomicAdd(&f.ccnt[i * GG + cell_of(f.ggeo[i], x, y)], 1u);   // tie rule is by cell index)
    if (NW == kRegNWBig) { f.txy[pos] = make_float2(x, y); f.tix[pos] = (unsigned short)idx; }
    else f.txyi[pos] = make_float4(x, y, __int_as_float(idx), 0.f);
  });
  const ScanView& src = job.scans[last];
  const int n_src = *src.n_cells;
  for (int s = tid; s < n_src; s += NW * 64) {
    f.smean[s] = src.mean[s];
  }
  __syncthreads();
  REG_TACC(15);
}

// One association pass (n_scan_normal.cpp:213-318) at source pose xsrc; fills `dn`; returns #blocks.
template <int NW>
__device__ int associate_fused(const RegJob& job, const RegCommon& cm, const double* xsrc, int itr, const FusedLds& f,
                               double* gl_dense, Dense& dn, int* ipart, int& iphase) {
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int last = job.n_scans - 1;
  const int n_src = *job.scans[last].n_cells;
  const int n_pairs = last * n_src;
  const double curr_radius = (itr == 1) ? 2 * cm.par.radius : cm.par.radius;    // :220
  const double r2 = curr_radius * curr_radius;
  const float rwin = (float)curr_radius + 1e-3f;
  REG_T0();
  if (tid < last) {                                      // Tsrctotar_i = Ttar_i^-1 * Tsrc  (:222)
    const double* k = f.kf + tid * 12;
    const Aff2 Ttar{k[0], k[1], k[2], k[3], k[4], k[5]};
    const Aff2 Tst = aff_mul(aff_inv(Ttar), aff_from_xyt(xsrc));
    double* o = f.kf + tid * 12 + 6;
    o[0] = Tst.l0; o[1] = Tst.l1; o[2] = Tst.l2; o[3] = Tst.l3; o[4] = Tst.t0; o[5] = Tst.t1;
  }
  __syncthreads();
  REG_TACC(0);
  // ---- pass 1: exact windowed 1-NN + normal gate -> match[] --------------------------------------
  int accepted = 0;
  {
    int i = 0, s = tid;
    while (s >= n_src && i < last) { s -= n_src; i++; }
    for (int p = tid; p < n_pairs; p += NW * 64) {
      const double* T = f.kf + i * 12 + 6;
      const double2 u = f.smean[s];
      const double px = T[0] * u.x + T[1] * u.y + T[4];
      const double py = T[2] * u.x + T[3] * u.y + T[5];
      const float qx = (float)px, qy = (float)py;                               // pointnormal.cpp:240-242
      // Exact 1-NN (FLANN L2_Simple float distance, lowest index on ties) restricted to the window
      // |x - qx| <= radius of the x-sorted order: nothing outside it can pass `dist < radius^2`.
      const float xlo = qx - rwin, xhi = qx + rwin;
      const float4 gg = f.ggeo[i];
      const int G = f.G;
      const int cx0 = min(G - 1, max(0, (int)floorf((xlo - gg.x) * gg.z))), cx1 = min(G - 1, max(0, (int)floorf((xhi - gg.x) * gg.z)));
      const int cy0 = min(G - 1, max(0, (int)floorf((qy - rwin - gg.y) * gg.z))), cy1 = min(G - 1, max(0, (int)floorf((qy + rwin - gg.y) * gg.z)));
      // (distance, index) packed into one 64-bit key: squared distances are non-negative floats, whose bit
      // patterns order like the values, so "nearest, lowest index on ties" is a single unsigned minimum
      unsigned long long bestkey = ~0ull;
      auto visit = [&](const float4 c) {            // candidates outside the radius cannot win: their d exceeds r^2
        const float dx = __fsub_rn(qx, c.x), dy = __fsub_rn(qy, c.y);
        const float d = __fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy));
        const unsigned long long key = ((unsigned long long)__float_as_uint(d) << 32) | (unsigned)__float_as_int(c.z);
        bestkey = key < bestkey ? key : bestkey;
      };
      auto fetch = [&](int q) -> float4 {           // a candidate record in either layout
        if (NW == kRegNWBig) { const float2 v = f.txy[q]; return make_float4(v.x, v.y, __int_as_float((int)f.tix[q]), 0.f); }
        return f.txyi[q];
      };
      const unsigned short* cs = f.cstart + i * G * G;
      // cells of a grid row are contiguous in txyi.  The bounds of the (at most three, unless the radius
      // exceeds the cell edge) rows are fetched together; candidates are visited four per step with the
      // indices clamped to the run -- revisiting the last candidate is harmless (same d, same index).
      auto scan_run = [&](int qb, int qe) {
        for (int q = qb; q < qe; q += 4) {
          const int l = qe - 1;
          const float4 ca = fetch(q), cb = fetch(min(q + 1, l)), cc = fetch(min(q + 2, l)), cd = fetch(min(q + 3, l));
          visit(ca); visit(cb); visit(cc); visit(cd);
        }
      };
      {
        const int r1 = min(cy0 + 1, cy1), r2 = min(cy0 + 2, cy1);
        const int b0 = cs[cy0 * G + cx0], e0 = cs[cy0 * G + cx1 + 1];
        const int b1 = cs[r1 * G + cx0], e1 = cs[r1 * G + cx1 + 1];
        const int b2 = cs[r2 * G + cx0], e2 = cs[r2 * G + cx1 + 1];
        scan_run(b0, e0);
        if (cy1 > cy0) scan_run(b1, e1);
        if (cy1 > cy0 + 1) scan_run(b2, e2);
        for (int cy = cy0 + 3; cy <= cy1; cy++) scan_run((int)cs[cy * G + cx0], (int)cs[cy * G + cx1 + 1]);
      }
      const int best = bestkey == ~0ull ? -1 : (int)(unsigned)(bestkey & 0xFFFFFFFFu);
      const float bestd = __uint_as_float((unsigned)(bestkey >> 32));
      int m = -1;
      if (best >= 0 && (double)bestd < r2) {                                    // pointnormal.cpp:250
        const double2 ns = job.scans[last].normal[s];
        const double2 nt = ((const double2*)f.tptr[i * 5 + 1])[best];
        const double nsx = T[0] * ns.x + T[1] * ns.y, nsy = T[2] * ns.x + T[3] * ns.y;
        if (fmax(nsx * nt.x + nsy * nt.y, 0.0) > cm.angle_outlier) m = best;    // :244-245
      }
      f.match[p] = (unsigned short)m;                                           // -1 -> 0xFFFF
      accepted += (m >= 0);
      s += NW * 64;
      while (s >= n_src && i < last) { s -= n_src; i++; }
    }
  }
  REG_TACC(1);
  // ---- block scan of the accepted counts (thread-major order, deterministic) ----------------------
  const int incl = wave_incl_scan_i32(accepted);
  int base = incl - accepted, total;
  if (NW == 1) {
    total = __builtin_amdgcn_readlane(incl, 63);
  } else {
    int* buf = ipart + iphase * kRegMaxNW;
    if (lane == 63) buf[wave] = incl;
    __syncthreads();
    for (int wv = 0; wv < wave; wv++) base += buf[wv];
    int tt = buf[0];
#pragma unroll
    for (int wv = 1; wv < NW; wv++) tt += buf[wv];
    total = __builtin_amdgcn_readfirstlane(tt);
    iphase ^= 1;
  }
  const bool in_lds = total <= f.dense_cap;
  dense_bind(dn, in_lds ? f.dense : gl_dense, in_lds ? f.dense_cap : cm.slots_cap, cm.dense_fields, f.smean);
  dn.n = total;
  const size_t cap = (size_t)dn.cap;
  REG_TACC(2);
  // ---- pass 2: matched target attributes -> weights + world-frame block data -> dense arrays ------
  // Software-pipelined: the (L2-resident) attribute gathers of this thread's next pair are issued before the
  // arithmetic of the current one, so their latency overlaps it instead of adding to it.
  {
    struct Gathered {
      int best, i, s, tns, sns;
      double2 nt, tm, ns;
      double tsc, ssc;
      double4 S;
    };
    const ScanView& srcv = job.scans[last];
    auto gather = [&](int p, int i, int s) {
      Gathered g;
      g.best = p < n_pairs ? (int)f.match[p] : 0xFFFF;
      if (g.best == 0xFFFF) g.best = -1;
      g.i = i; g.s = s;
      if (g.best >= 0) {
        const void* const* tp = f.tptr + i * 5;           // matched target's attribute arrays
        g.nt = ((const double2*)tp[1])[g.best];
        g.tm = ((const double2*)tp[0])[g.best];
        g.tns = ((const int32_t*)tp[2])[g.best];
        g.tsc = ((const double*)tp[3])[g.best];
        g.ns = srcv.normal[s];
        g.sns = srcv.nsamples[s];
        g.ssc = srcv.scale[s];
        if (cm.par.cost == CFEAR_P2D) g.S = ((const double4*)tp[4])[g.best];
      }
      return g;
    };
    int c = base, i = 0, s = tid;
    while (s >= n_src && i < last) { s -= n_src; i++; }
    Gathered cur = gather(tid, i, s);
    for (int p = tid; p < n_pairs; p += NW * 64) {
      s += NW * 64;
      while (s >= n_src && i < last) { s -= n_src; i++; }
      const Gathered nxt = gather(p + NW * 64, i, s);
      if (cur.best >= 0) {
        const double* K = f.kf + cur.i * 12;              // Ttar
        const double* T = K + 6;                          // Tsrctotar
        const double2 nt = cur.nt, tm = cur.tm, ns = cur.ns;
        const double nsx = T[0] * ns.x + T[1] * ns.y, nsy = T[2] * ns.x + T[3] * ns.y;
        const double direction_similarity = fmax(nsx * nt.x + nsy * nt.y, 0.0);   // :244
        const double w = get_weight(cm.par.weight_opt, (double)cur.sns, (double)cur.tns,
                                    direction_similarity, cur.ssc, cur.tsc);       // :247-253, :273
        dn.sidx[c] = cur.s;
        dn.p[c] = K[0] * tm.x + K[1] * tm.y + K[4];                               // Ttar * tar_mean
        dn.p[cap + c] = K[2] * tm.x + K[3] * tm.y + K[5];
        dn.p[2 * cap + c] = w;
        if (cm.par.cost == CFEAR_P2D) {                                           // :288-297
          const double4 S = cur.S;
          const double a00 = K[0] * S.x + K[1] * S.z, a01 = K[0] * S.y + K[1] * S.w;
          const double a10 = K[2] * S.x + K[3] * S.z, a11 = K[2] * S.y + K[3] * S.w;
          const double c00 = (cm.par.regularization + (a00 * K[0] + a01 * K[1])) * cm.par.cov_scale;
          const double c01 = (0.0 + (a00 * K[2] + a01 * K[3])) * cm.par.cov_scale;
          const double c10 = (0.0 + (a10 * K[0] + a11 * K[1])) * cm.par.cov_scale;
          const double c11 = (cm.par.regularization + (a10 * K[2] + a11 * K[3])) * cm.par.cov_scale;
          const double det = c00 * c11 - c10 * c01, invdet = 1.0 / det;
          const double i00 = c11 * invdet, i10 = -c10 * invdet, i11 = c00 * invdet;
          const double l00 = sqrt(i00), l10 = i10 / l00;
          dn.p[3 * cap + c] = l00; dn.p[4 * cap + c] = l10; dn.p[5 * cap + c] = sqrt(i11 - l10 * l10);
        } else if (cm.par.cost == CFEAR_P2L) {
          dn.p[3 * cap + c] = K[0] * nt.x + K[1] * nt.y;                          // Ttar.linear() * tar_normal
          dn.p[4 * cap + c] = K[2] * nt.x + K[3] * nt.y;
        }
        c++;
      }
      cur = nxt;
    }
  }
  __syncthreads();
  REG_TACC(3);
  return total;
}

// Solves the SPD system A y = b (3x3, A = J^T J + D^2) by LDL^T: three reciprocals (rcp_newton: this chain of dependent
// fp64 instructions is the longest serial piece of an LM iteration), no square roots.
// Ceres factorises the same matrix with a sparse Cholesky; the solutions agree to rounding.
__device__ __forceinline__ bool chol3_solve(const double A[9], const double b[3], double y[3]) {
  const double d0 = A[0];
  if (!(d0 > 0.0)) return false;
  const double i0 = rcp_newton(d0);
  const double l10 = A[3] * i0, l20 = A[6] * i0;
  const double d1 = A[4] - l10 * A[3];
  if (!(d1 > 0.0)) return false;
  const double i1 = rcp_newton(d1);
  const double t21 = A[7] - l20 * A[3];
  const double l21 = t21 * i1;
  const double d2 = A[8] - l20 * A[6] - l21 * t21;
  if (!(d2 > 0.0)) return false;
  const double i2 = rcp_newton(d2);
  const double z0 = b[0];
  const double z1 = b[1] - l10 * z0;
  const double z2 = b[2] - l20 * z0 - l21 * z1;
  y[2] = z2 * i2;
  y[1] = z1 * i1 - l21 * y[2];
  y[0] = z0 * i0 - l10 * y[1] - l20 * y[2];
  return isfinite(y[0]) && isfinite(y[1]) && isfinite(y[2]);
}

// Rotation of an evaluation point: Cody-Waite reduction + Taylor polynomials (fastmath.hpp, an ulp or two from libm) --
// a fifth of libm's instruction chain; headings beyond the reduction's range go to libm.
template <bool PIN = false>
__device__ __forceinline__ void sincos_pose(const double a, double* s, double* c) {
  if (fabs(a) <= 1e5) sincos_reduced<PIN>(a, s, c);
  else sincos(a, s, c);
}

struct LmSummary {
  double initial_cost, final_cost, last_relative_decrease;
  int n_pushed;          // summary_.iterations.size()
  bool usable;
};

// ceres::Solve as configured by the reference (Ceres 2.1 defaults, max_num_iterations = max_iter):
// same bookkeeping as the oracle's lm_solve / SURVEY Appendix B.4.  Block-wide collective.
// The trust-region bookkeeping (a few hundred dependent fp64 instructions per iteration) runs on
// wavefront 0 only; the next evaluation point and the stop flag are broadcast through a small LDS
// control block, so the other wavefronts of the workgroup -- and the second workgroup sharing their
// SIMDs -- do not spend issue slots on redundant scalar math.
template <int NW, int COST, int LOSS>
__device__ void lm_solve(const RegCommon& cm, const Dense& dn, double x[3], int max_iter, LmSummary& sum, double* part,
                         int& phase, double* ctrl /* LDS [16] */) {
  const double function_tolerance = 1e-6, gradient_tolerance = 1e-10, parameter_tolerance = 1e-8;
  const double min_relative_decrease = 1e-3, min_lm_diagonal = 1e-6, max_lm_diagonal = 1e32;
  const double max_radius = 1e16, min_radius = 1e-32;
  const bool w0 = NW == 1 || (threadIdx.x >> 6) == 0;
  const bool writer = threadIdx.x == 0;
  double radius = 1e4, decrease_factor = 2.0;
  bool reuse_diagonal = false;
  double diagonal[3] = {0, 0, 0};
  int num_consecutive_invalid_steps = 0;

  double cur[10];                                        // cost, g, H at the accepted x
  {
    double s0, c0;
    sincos_pose(x[2], &s0, &c0);
    eval_all<NW, COST, LOSS>(cm, dn, x, c0, s0, cur, part, phase);
  }
  double x_cost = cur[0];
  double scale[3] = {1, 1, 1};
  double gradient_max_norm = 0, x_norm = 0, min_iter_cost = x_cost, it_cost = x_cost, it_rel = 0.0;
  double model_cost_change = 0.0;
  bool it_success = true;
  int iteration = 0;
  sum.initial_cost = x_cost;
  sum.n_pushed = 0;
  sum.usable = true;
  sum.last_relative_decrease = 0.0;
  if (w0) {
    scale[0] = 1.0 / (1.0 + sqrt(cur[4]));               // jacobi scaling from iteration 0
    scale[1] = 1.0 / (1.0 + sqrt(cur[7]));
    scale[2] = 1.0 / (1.0 + sqrt(cur[9]));
    gradient_max_norm = fmax(fabs(cur[1]), fmax(fabs(cur[2]), fabs(cur[3])));
    x_norm = sqrt_newton(x[0] * x[0] + x[1] * x[1] + x[2] * x[2]);
  }
  double cand[3] = {x[0], x[1], x[2]};
  double cnd[10];                                        // cost (and, speculatively, g and H) at cand
  bool have_cnd = false;
  for (;;) {
    int done = 0;
    REG_T0();
    if (w0) {
      bool proceed = true;
      if (have_cnd) {                                    // the candidate of the previous round was evaluated
        const double cand_cost = cnd[0];
        const double step_norm2 = (x[0] - cand[0]) * (x[0] - cand[0]) + (x[1] - cand[1]) * (x[1] - cand[1]) +
                                  (x[2] - cand[2]) * (x[2] - cand[2]);
        const double cost_change = x_cost - cand_cost;
        const double step_bound = parameter_tolerance * (x_norm + parameter_tolerance);
        if (step_norm2 <= step_bound * step_bound) { done = 1; proceed = false; }      // ||step|| <= tolerance (||x|| + tolerance)
        else if (fabs(cost_change) <= function_tolerance * x_cost) { done = 1; proceed = false; }
        else {
          it_rel = cost_change / model_cost_change;
          if (it_rel > min_relative_decrease) {
            x[0] = cand[0]; x[1] = cand[1]; x[2] = cand[2];
            x_norm = sqrt_newton(x[0] * x[0] + x[1] * x[1] + x[2] * x[2]);
#pragma unroll
            for (int k = 0; k < 10; k++) cur[k] = cnd[k];
            x_cost = cand_cost;
            gradient_max_norm = fmax(fabs(cur[1]), fmax(fabs(cur[2]), fabs(cur[3])));
            it_cost = x_cost; it_success = true;
            const double q = 2.0 * it_rel - 1.0;
            radius = radius * rcp_newton(fmax(1.0 / 3.0, 1.0 - q * q * q));
            radius = fmin(max_radius, radius);
            decrease_factor = 2.0; reuse_diagonal = false;
          } else {
            it_cost = cand_cost; it_success = false;
            radius = radius / decrease_factor; decrease_factor *= 2.0; reuse_diagonal = true;
          }
        }
      }
      REG_TACC(16);
      while (proceed) {
        // FinalizeIterationAndCheckIfMinimizerCanContinue
        sum.n_pushed++;
        sum.last_relative_decrease = it_rel;
        min_iter_cost = fmin(min_iter_cost, it_cost);
        if (iteration >= max_iter || (it_success && gradient_max_norm <= gradient_tolerance) || radius <= min_radius) {
          done = 1;
          break;
        }
        iteration++;
        it_cost = 0.0; it_rel = 0.0; it_success = false;
        // scaled quantities: J_s = J diag(scale)
        const double gs[3] = {cur[1] * scale[0], cur[2] * scale[1], cur[3] * scale[2]};
        double Hs[9];
        Hs[0] = cur[4] * scale[0] * scale[0]; Hs[1] = cur[5] * scale[0] * scale[1]; Hs[2] = cur[6] * scale[0] * scale[2];
        Hs[3] = Hs[1]; Hs[4] = cur[7] * scale[1] * scale[1]; Hs[5] = cur[8] * scale[1] * scale[2];
        Hs[6] = Hs[2]; Hs[7] = Hs[5]; Hs[8] = cur[9] * scale[2] * scale[2];
        if (!reuse_diagonal) {
          diagonal[0] = fmin(fmax(Hs[0], min_lm_diagonal), max_lm_diagonal);
          diagonal[1] = fmin(fmax(Hs[4], min_lm_diagonal), max_lm_diagonal);
          diagonal[2] = fmin(fmax(Hs[8], min_lm_diagonal), max_lm_diagonal);
        }
        double A[9];
#pragma unroll
        for (int k = 0; k < 9; k++) A[k] = Hs[k];
        const double inv_radius = rcp_newton(radius);                          // radius in [1e-32, 1e16]
#pragma unroll
        for (int k = 0; k < 3; k++) A[k * 3 + k] += diagonal[k] * inv_radius;  // D^2 = diag / radius
        double y[3], step[3] = {0, 0, 0};
        REG_TACC(17);
        const bool solved = chol3_solve(A, gs, y);
        REG_TACC(18);
        reuse_diagonal = true;
        bool step_is_valid = false;
        model_cost_change = 0.0;
        if (solved) {
          step[0] = -y[0]; step[1] = -y[1]; step[2] = -y[2];
          // -(J step)^T (r + J step / 2) = -step^T g_s - 1/2 step^T H_s step
          const double sg = step[0] * gs[0] + step[1] * gs[1] + step[2] * gs[2];
          const double hs0 = Hs[0] * step[0] + Hs[1] * step[1] + Hs[2] * step[2];
          const double hs1 = Hs[3] * step[0] + Hs[4] * step[1] + Hs[5] * step[2];
          const double hs2 = Hs[6] * step[0] + Hs[7] * step[1] + Hs[8] * step[2];
          model_cost_change = -sg - (step[0] * hs0 + step[1] * hs1 + step[2] * hs2) / 2.0;
          step_is_valid = model_cost_change > 0.0;
        }
        if (!step_is_valid) {
          if (++num_consecutive_invalid_steps >= 5) { sum.usable = false; done = 1; break; }
          radius = radius / decrease_factor; decrease_factor *= 2.0; reuse_diagonal = true;
          it_cost = x_cost; it_success = false; it_rel = 0.0;
          continue;                                      // no evaluation for an invalid step
        }
        num_consecutive_invalid_steps = 0;
#pragma unroll
        for (int k = 0; k < 3; k++) cand[k] = x[k] + step[k] * scale[k];
        break;
      }
      REG_TACC(19);
      if (!done) {                                       // the rotation of the next evaluation point, once per
        double sn, cs;                                   // workgroup instead of once per wavefront
        sincos_pose(cand[2], &sn, &cs);
        if (writer) { ctrl[10] = cs; ctrl[11] = sn; }
      }
      if (writer) {
        ctrl[0] = cand[0]; ctrl[1] = cand[1]; ctrl[2] = cand[2];
        ((int*)(ctrl + 3))[0] = done;
      }
      REG_TACC(20);
    }
    __syncthreads();
    done = __builtin_amdgcn_readfirstlane(((const int*)(ctrl + 3))[0]);
    if (done) break;
    cand[0] = ctrl[0]; cand[1] = ctrl[1]; cand[2] = ctrl[2];
    REG_TACC(21);
    eval_all<NW, COST, LOSS>(cm, dn, cand, ctrl[10], ctrl[11], cnd, part, phase);   // its barrier also protects ctrl
    have_cnd = true;
  }
  // results of wavefront 0 -> every wavefront (the outer association loop is wave-uniform)
  if (writer) {
    ctrl[4] = x[0]; ctrl[5] = x[1]; ctrl[6] = x[2];
    ctrl[7] = fmin(sum.initial_cost, min_iter_cost);     // solver.cc SetSummaryFinalCost
    ctrl[8] = sum.last_relative_decrease;
    ((int*)(ctrl + 9))[0] = sum.n_pushed;
    ((int*)(ctrl + 9))[1] = sum.usable ? 1 : 0;
  }
  __syncthreads();
  x[0] = ctrl[4]; x[1] = ctrl[5]; x[2] = ctrl[6];
  sum.final_cost = ctrl[7];
  sum.last_relative_decrease = ctrl[8];
  sum.n_pushed = __builtin_amdgcn_readfirstlane(((const int*)(ctrl + 9))[0]);
  sum.usable = __builtin_amdgcn_readfirstlane(((const int*)(ctrl + 9))[1]) != 0;
}

template <int NW, int COST, int LOSS>
__global__ __launch_bounds__(NW * 64) void register_kernel(const RegJob* __restrict__ jobs, const RegCommon cm) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  double* part = (double*)smem;                          // [2][4][10]
  int* ipart = (int*)(smem + kRegIpartOff);              // [2][16]
  double* ctrl = (double*)(smem + kRegCtrlOff);          // [16] LM control block
  LdsTargets lt;
  lt.x = (float*)(smem + kRegFixedLds);
  lt.y = lt.x + cm.lds_targets;
  lt.idx = (int*)(lt.y + cm.lds_targets);
  double* lds_dense = (double*)(smem + kRegFixedLds + reg_lds_targets_bytes(cm.lds_targets));
  const RegJob& job = *(const RegJob*)((const char*)jobs + (size_t)blockIdx.x * cm.job_stride);
  cfear_reg_result* res = cm.results + blockIdx.x;
  if ((cm.big_mode == 2 || cm.only_deferred) && res->status != kRegDeferred) return;   // only what the launch before deferred
  const int last = job.n_scans - 1;
  const int n_src = *job.scans[last].n_cells;
  int max_tar = 0;
  for (int i = 0; i < last; i++) max_tar = max(max_tar, *job.scans[i].n_cells);
  const int n_slots = last * n_src;
  double x[3] = {job.poses[last][0], job.poses[last][1], job.poses[last][2]};
  if (n_slots > cm.slots_cap || max_tar > cm.lds_targets) {
    if (cm.cost_only) {
      const int m = cm.n_samples > 0 ? cm.n_samples : 1;
      for (int sidx = blockIdx.y * (int)blockDim.x + (int)threadIdx.x; sidx < m; sidx += (int)(gridDim.y * blockDim.x)) {
        cfear_reg_result* r = cm.results + (size_t)blockIdx.x * m + sidx;
        r->pose[0] = x[0]; r->pose[1] = x[1]; r->pose[2] = x[2];
        r->score = 0; r->final_cost = 0; r->num_residuals = 0; r->outer_iters = 0; r->lm_iters = 0;
        r->status = CFEAR_ERR_CAPACITY; r->last_relative_decrease = 0; r->reserved = 0;
      }
      return;
    }
    if (threadIdx.x == 0) {
      res->pose[0] = x[0]; res->pose[1] = x[1]; res->pose[2] = x[2];
      res->score = 0; res->final_cost = 0; res->num_residuals = 0; res->outer_iters = 0; res->lm_iters = 0;
      res->status = CFEAR_ERR_CAPACITY; res->last_relative_decrease = 0; res->reserved = 0;
    }
    return;
  }
  const size_t scr_idx = (size_t)blockIdx.x * gridDim.y + blockIdx.y;
  const Slots sl = slots_of(cm.scratch + scr_idx * cm.scratch_stride, cm.slots_cap);
  double* gl_dense = (double*)(cm.scratch + scr_idx * cm.scratch_stride + slots_bytes(cm.slots_cap));
  Dense dn;
  int phase = 0, iphase = 0;
  int sum_tar = 0;
  for (int i = 0; i < last; i++) sum_tar += *job.scans[i].n_cells;
  FusedLds fl;
  const bool fused = fused_carve(smem, cm.lds_total, last, sum_tar, n_src, n_slots, cm.dense_fields, fl, NW == kRegNWBig);
  if (!fused && cm.big_mode == 1) {
    // Too large for the association in 80 KB of LDS, but not for a workgroup that owns the CU's LDS: leave it to the
    // launches behind this one instead of the x-window path below (13 x slower per registration on 1 400-cell scans).
    // Those are register3_kernel's large forms (keyframe tables staged group by group: the source means and ONE keyframe's
    // grid must fit) and this kernel again with the CU's LDS (every target packed at once) -- whatever none of them can
    // run, the last one takes through the x-window path.
    FusedLds probe;
    const size_t r3_need = 4096 + (size_t)last * 176 + (size_t)n_src * 16 + (size_t)kScanGridStartPad * 2 + (size_t)max_tar * 16;
    if ((r3_need <= cm.lds_big && n_src < 65536 && sum_tar <= 65535) ||
        fused_carve(smem, cm.lds_big, last, sum_tar, n_src, n_slots, cm.dense_fields, probe, true)) {   // (that launch packs its targets)
      if (threadIdx.x == 0) { res->status = kRegDeferred; res->reserved = 1.0; }
      return;
    }
  }
  // Code placement (MI355X_MICROARCH.md, code-placement sensitivity): without these 32 bytes the loops of this build of the
  // 4-wavefront kernel land where they run 1.3 % slower (0.930 against 0.917 ms per 2048 registrations; 16, 32 or 48 bytes
  // all restore it).  Executed once per registration.  Re-check with tools/ab.sh when the kernel changes.
  if (NW == kRegNW) asm volatile("s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0");
  REG_T0();
  if (fused) fused_stage<NW>(job, fl, (float)cm.par.radius);
  REG_TACC(6);
  const int rpb = cm.par.cost == CFEAR_P2L ? 1 : 2;
  if (cm.cost_only) {
    // n_scan_normal_reg::GetCost (n_scan_normal.cpp:186-211): one association pass (radius by the leftover
    // itr_, :220) + the robust cost at the given pose; with n_samples > 0 once per pose of the sampling
    // grid of approximateCovarianceBySampling (odometrykeyframefuser.cpp:287-303: theta outer, x, y inner).
    int git = job.itr ? job.itr : cm.par.itr;
    if (cm.prior) {                                      // sample around the pose a Register launch just produced
      const cfear_reg_result& pr = cm.prior[blockIdx.x];
      x[0] = pr.pose[0]; x[1] = pr.pose[1]; x[2] = pr.pose[2];
      git = pr.outer_iters;
    }
    const int m = cm.n_samples > 0 ? cm.n_samples : 1;
    for (int sidx = blockIdx.y; sidx < m; sidx += gridDim.y) {
      double xs[3] = {x[0], x[1], x[2]};
      if (cm.n_samples > 0) {
        const int n = cm.samples_per_axis;
        auto lin = [n](double half, int i) {             // linspace(-half, half, n)[i] (loopclosure.cpp:866-890)
          if (n == 1) return -half;
          const double delta = (half - (-half)) / ((double)n - 1.0);
          return i < n - 1 ? -half + delta * (double)i : half;
        };
        xs[0] = lin(cm.xy_half, (sidx / n) % n) + x[0];
        xs[1] = lin(cm.xy_half, sidx % n) + x[1];
        xs[2] = lin(cm.yaw_half, sidx / (n * n)) + x[2];
      }
      int n_blocks;
      if (fused) {
        n_blocks = associate_fused<NW>(job, cm, xs, git, fl, gl_dense, dn, ipart, iphase);
      } else {
        const int mine = associate_all<NW>(job, cm, xs, git, sl, lt);
        __threadfence_block();
        n_blocks = compact_slots<NW>(job, cm, sl, n_slots, n_src, mine, lds_dense, gl_dense, dn, ipart, iphase);
      }
      const int nres = n_blocks * rpb;
      double cur[10];
#pragma unroll
      for (int k = 0; k < 10; k++) cur[k] = 0.0;
      if (nres > 1) {
        double s0, c0;
        sincos(xs[2], &s0, &c0);
        eval_all<NW, COST, LOSS>(cm, dn, xs, c0, s0, cur, part, phase);
      }
      __syncthreads();                                   // the dense arrays are rewritten by the next sample
      if (threadIdx.x == 0) {
        cfear_reg_result* r = cm.results + (size_t)blockIdx.x * m + sidx;
        r->pose[0] = xs[0]; r->pose[1] = xs[1]; r->pose[2] = xs[2];
        r->final_cost = cur[0];
        r->score = nres > 1 ? cur[0] / (double)nres : 0.0;                  // score_ (:209)
        r->num_residuals = nres; r->outer_iters = git; r->lm_iters = 0;
        r->status = nres > 1 ? CFEAR_OK : CFEAR_ERR_TOO_FEW_RESIDUALS;      // :200-203
        r->last_relative_decrease = 0.0; r->reserved = 0.0;
      }
    }
    return;
  }
  // n_scan_normal.cpp:82-185
  double prev_par[3] = {x[0], x[1], x[2]};
  double prev_score = DBL_MAX;
  bool success = true;
  int itr = 1, lm_iters = 0, num_residuals = 0, fail_status = CFEAR_OK;
  LmSummary summary;
  summary.final_cost = 0.0; summary.last_relative_decrease = 0.0; summary.n_pushed = 0; summary.usable = false;
#ifdef CFEAR_REG_TIMING
  long long t_assoc = 0, t_total0 = __builtin_readcyclecounter();
#endif
  for (itr = 1; itr <= cm.par.max_itr_association && success; itr++) {
#ifdef CFEAR_REG_TIMING
    const long long ta0 = __builtin_readcyclecounter();
#endif
    int n_blocks;
    if (fused) {
      n_blocks = associate_fused<NW>(job, cm, x, itr, fl, gl_dense, dn, ipart, iphase);
    } else {
      const int mine = associate_all<NW>(job, cm, x, itr, sl, lt);
      __threadfence_block();
      n_blocks = compact_slots<NW>(job, cm, sl, n_slots, n_src, mine, lds_dense, gl_dense, dn, ipart, iphase);
    }
#ifdef CFEAR_REG_TIMING
    t_assoc += __builtin_readcyclecounter() - ta0;
#endif
    num_residuals = n_blocks * rpb;
    success = num_residuals > 1;                                  // :368-369
    if (!success) { fail_status = CFEAR_ERR_TOO_FEW_RESIDUALS; break; }
    double xi[3] = {x[0], x[1], x[2]};
    lm_solve<NW, COST, LOSS>(cm, dn, xi, cm.par.max_itr_solver, summary, part, phase, ctrl);
    lm_iters += summary.n_pushed - 1;
    success = summary.usable;
    if (success) { x[0] = xi[0]; x[1] = xi[1]; x[2] = xi[2]; } else fail_status = CFEAR_ERR_SOLVER;
    const double current_score = summary.final_cost;
    const double rel_improvement = (prev_score - current_score) / prev_score;
    // n_scan_normal.cpp:134-149.  Written with selects instead of nested if/else-break: hipcc
    // (ROCm 7.2, clang 22) mis-merged the phi of prev_par on the "continue" edge of the nested
    // form and kept the stale prev_par (found by tests/test_gpu_register.py P2P-None-1).
    const bool past_min = itr > cm.par.min_itr;
    const bool worse = past_min && (prev_score < current_score);          // recover to prev iteration
    const bool small_outer = past_min && !worse && (rel_improvement < cm.par.score_tolerance);
    const bool small_inner = past_min && !worse && !small_outer &&
                             (summary.last_relative_decrease < cm.par.score_tolerance || summary.n_pushed == 1);
    const bool stop = worse || small_outer || small_inner;
#pragma unroll
    for (int k = 0; k < 3; k++) {
      x[k] = worse ? prev_par[k] : x[k];
      prev_par[k] = stop ? prev_par[k] : x[k];
    }
    prev_score = stop ? prev_score : current_score;
    if (stop) break;
  }
  if (threadIdx.x == 0) {
    res->pose[0] = x[0]; res->pose[1] = x[1]; res->pose[2] = x[2];
    res->final_cost = summary.final_cost;
    res->num_residuals = num_residuals;
    res->outer_iters = itr;
    res->lm_iters = lm_iters;
    res->last_relative_decrease = summary.last_relative_decrease;
    res->reserved = (cm.big_mode == 2 || !fused) ? 1.0 : 0.0;     // a large registration: tells the caller to keep the second launch on
#ifdef CFEAR_REG_TIMING
    res->reserved = (double)t_assoc;
    res->last_relative_decrease = (double)(__builtin_readcyclecounter() - t_total0);
    if (blockIdx.x == 0) {
      printf("reg cycles: total %lld | stage %lld | Tst %lld nn+gate %lld scan %lld gather %lld | sincos %lld eval %lld reduce %lld | outer %d lm %d n %d\n",
             (long long)res->last_relative_decrease + g_reg_t[6], g_reg_t[6], g_reg_t[0], g_reg_t[1], g_reg_t[2], g_reg_t[3],
             g_reg_t[7], g_reg_t[4], g_reg_t[5], itr, lm_iters, num_residuals);
      printf("  stage split:"); for (int k = 8; k < 16; k++) printf(" %lld", g_reg_t[k]); printf("\n");
      printf("  lm bookkeeping: decide %lld | scale+diag %lld | chol %lld | model+cand %lld | sincos+ctrl %lld | barrier+read %lld\n",
             g_reg_t[16], g_reg_t[17], g_reg_t[18], g_reg_t[19], g_reg_t[20], g_reg_t[21]);
      for (int k = 0; k < 32; k++) g_reg_t[k] = 0;
    }
#endif
    if (success) { res->score = summary.final_cost / (double)num_residuals; res->status = CFEAR_OK; }   // :162
    else { res->score = 0.0; res->status = fail_status; }
  }
}


// ---------------------------------------------------------------------------------------------------
// register3_kernel: the regular registration at THREE workgroups per CU (round 4).
//
// register_kernel holds 75 KB of LDS and 231 VGPRs per workgroup -- two registrations per CU -- and idles three of its four
// wavefronts while wavefront 0 walks the trust-region chain.  The serial chain can only be hidden by more co-resident
// registrations, so this form fits 52 KB and 168 VGPRs:
//   * the keyframes' grids come PREBUILT from the scans (ScanView::grid_*, built once per scan by sort_cells_block): staging
//     an association is a copy of ~30 KB from L2 into LDS, not a counting sort;
//   * with restaging that cheap, the association's tables (cell starts + target records) and the LM phase's dense
//     correspondence arrays ALIAS: the tables are copied in again at the start of every outer iteration;
//   * the outer loop's state lives in LDS across the two phases (r3 state block), so neither phase carries the other's
//     registers.
// Same arithmetic as register_kernel's fused path, statement for statement (n_scan_normal.cpp:82-185, 213-318): the
// results are bit-identical.  A registration that does not fit (keyframe tables + source cells beyond the LDS, a scan
// without grid tables) is marked kRegDeferred and left to register_kernel, launched behind this kernel.
// ---------------------------------------------------------------------------------------------------
// LDS-resident solver state of register3_kernel (doubles; the int fields share the last slots).  Nothing of the
// trust-region bookkeeping is held in registers across an evaluation: wavefront 0 loads the block, walks one round of the
// chain and stores it back, so the evaluation's registers and the bookkeeping's never coexist (lm_solve keeps ~70 VGPRs
// of wave-uniform state live across eval_all, which is what holds register_kernel at two wavefronts per SIMD).
enum {
  S_X = 0, S_XCOST = 3, S_CUR = 4 /* g[3], H[6] */, S_SCALE = 13, S_DIAG = 16, S_XNORM = 19, S_GMAX = 20, S_RADIUS = 21,
  S_DEC = 22, S_MINCOST = 23, S_MODEL = 24, S_CAND = 25, S_COS = 28, S_SIN = 29, S_ITCOST = 30, S_ITREL = 31, S_INIT = 32,
  S_LASTREL = 33, S_FINAL = 34, S_INTS = 35 /* 8 ints */, S_OUTER = 39 /* x[3] prev_par[3] prev_score */, S_COUNT = 48
};
enum { SI_ITER = 0, SI_REUSE = 1, SI_INVALID = 2, SI_PUSHED = 3, SI_USABLE = 4, SI_DONE = 5, SI_ITSUCC = 6 };
constexpr size_t kR3StateOff = kRegFixedLds;                // behind register_kernel's fixed block
constexpr size_t kR3FixedLds = kRegFixedLds + S_COUNT * 8;

constexpr int kReg3NW = 4;                      // wavefronts of the regular form (three workgroups per CU)
constexpr int kReg3NWLarge = 16;                // ... of the large form: one workgroup per CU with all of its LDS (register_large16)
constexpr int kReg3NWHalf = 8;                  // ... of the half-CU form: two workgroups per CU, 80 KB each (register_large)
constexpr size_t kReg3Lds = 52 * 1024;          // x 3 = 156 KB of the CU's 160 KB

struct R3Lds {
  double* kf;              // [last][12]: Ttar (l0..l3, t0, t1), Tsrctotar (l0..l3, t0, t1)
  int* koff;               // [last + 1] prefix of the keyframes' cell counts
  const void** tptr;       // [last][6]: mean, normal, nsamples, scale, cov, grid block (cell starts + records): global pointers
  float4* ggeo;            // [last] grid geometry (x0, y0, cells per metre, -)
  double2* smean;          // [n_src]
  unsigned short* match;   // [n_pairs] matched target (index inside its keyframe), 0xFFFF = none
  unsigned short* gmatch;  // ... in the job's global scratch instead, when the LDS copy would leave no room for a keyframe's tables (else null)
  unsigned short* cstart;  // [last][kScanGridStartPad] absolute first record of every grid cell   } association phase; the dense
  float4* txyi;            // [sum_tar] (x, y, index bits, -) grouped by (keyframe, cell)            } arrays alias both
  double* dense;           // = cstart
  int dense_cap;
  int region;              // bytes of the aliased region: the keyframes' tables are staged in groups that fit it
};
constexpr int kR3Ptrs = 6;

// The dense correspondence arrays of register3_kernel: entries [0, cap) in LDS (SoA, stride cap), the rest -- a registration
// with more correspondences than the LDS holds -- in the job's global scratch (SoA, stride gcap).  Two typed pointers, two code
// paths per access: no generic (flat) addressing.
struct Dense3 {
  double* p; unsigned short* sidx; const double2* smean; int cap; int n;     // (source cells < 65 536: 26 bytes per P2P pair)
  double* gp; int* gsidx; size_t gcap;
};

__device__ __forceinline__ bool r3_carve(uint8_t* smem, size_t lds_total, int last, int sum_tar, int max_tar, int n_src, int n_pairs,
                                         int fields, unsigned short* gmatch_buf, R3Lds& f) {
  size_t off = kR3FixedLds;
  f.kf = (double*)(smem + off); off += (size_t)last * 12 * 8;
  f.tptr = (const void**)(smem + off); off += (size_t)last * kR3Ptrs * 8;
  f.ggeo = (float4*)(smem + off); off += (size_t)last * 16;
  f.koff = (int*)(smem + off); off += (((size_t)last + 1) * 4 + 15) & ~(size_t)15;
  f.smean = (double2*)(smem + off); off += (size_t)n_src * 16;
  // the match table: in LDS, unless that leaves no room for the largest keyframe's tables (1 400-cell scans in half a CU's
  // LDS) -- then in the job's global scratch (L2; written and read once per outer iteration by the same thread)
  const size_t match_bytes = (((size_t)n_pairs + 7) & ~(size_t)7) * 2;
  const size_t need_one = (size_t)kScanGridStartPad * 2 + (size_t)max_tar * 16;
  f.match = (unsigned short*)(smem + off); f.gmatch = nullptr;
  if (gmatch_buf && ((off + match_bytes + 15) & ~(size_t)15) + need_one > lds_total) { f.match = nullptr; f.gmatch = gmatch_buf; }
  else off += match_bytes;
  off = (off + 15) & ~(size_t)15;
  // room for one keyframe's tables at the very least (a keyframe that exceeds it: r3_associate reports it)
  if (sum_tar > 65535 || off + (size_t)kScanGridStartPad * 2 + 4096 > lds_total) return false;
  f.cstart = (unsigned short*)(smem + off);
  f.txyi = nullptr;                                        // set per group by r3_restage
  f.dense = (double*)(smem + off);
  f.region = (int)(lds_total - off);
  f.dense_cap = (int)((lds_total - off) / ((size_t)fields * 8 + 2)) & ~3;
  return true;
}

// once per registration: keyframe transforms and attribute pointers, piece prefix, source means
template <int NT>
__device__ bool r3_stage_once(const RegJob& job, const R3Lds& f, int* flag /* LDS */) {
  const int tid = threadIdx.x, last = job.n_scans - 1;
  if (tid == 0) {
    int acc = 0, ok = 1;
    for (int i = 0; i < last; i++) {
      const int n = gload<int>(job.scans[i].n_cells);
      f.koff[i] = acc;
      acc += n;
      const float4 g = gload_f4(job.scans[i].grid_geo);
      f.ggeo[i] = g;
      ok &= (g.w == 1.f);
    }
    f.koff[last] = acc;
    *flag = ok;
  }
  if (tid >= 64 && tid < 64 + last) {
    const int i = tid - 64;
    const Aff2 T = aff_from_xyt(job.poses[i]);
    double* k = f.kf + i * 12;
    k[0] = T.l0; k[1] = T.l1; k[2] = T.l2; k[3] = T.l3; k[4] = T.t0; k[5] = T.t1;
    const ScanView& tv = job.scans[i];
    const void** tp = f.tptr + i * kR3Ptrs;
    tp[0] = tv.mean; tp[1] = tv.normal; tp[2] = tv.nsamples; tp[3] = tv.scale; tp[4] = tv.cov; tp[5] = tv.grid_cstart;
  }
  const ScanView& src = job.scans[last];
  const int n_src = gload<int>(src.n_cells);
  for (int s = tid; s < n_src; s += NT) f.smean[s] = gload_d2(src.mean + s);
  __syncthreads();
  return *flag != 0;
}

// The grid tables of keyframes [i0, i1) from global memory into the aliased LDS region: their cell-start tables first (made
// absolute: + the keyframe's first record inside the group), then their records.  A scan keeps both in ONE block (cell starts,
// then records: ScanView::grid_cstart), so a keyframe is a run of 16-byte pieces from one base address; four keyframes x two
// pieces per thread are in flight before the first one is stored (one memory round trip for the usual registration).
template <int NT>
__device__ __forceinline__ void r3_restage(const R3Lds& f, int i0, int i1) {
  const int tid = threadIdx.x;
  const int k_lo = f.koff[i0];
  constexpr int kCs = kScanGridStartPad / 8;              // pieces of one cell-start table
  constexpr int kKf = 4, kPer = 2;
  const int rec0 = (i1 - i0) * kCs;                       // first record piece
  uint4* out = (uint4*)f.cstart;
  for (int ib = i0; ib < i1; ib += kKf) {
    g_u32x4 v[kKf][kPer];
    int dst[kKf][kPer];
    unsigned add[kKf];
#pragma unroll
    for (int q = 0; q < kKf; q++) {
      const int i = ib + q;
#pragma unroll
      for (int h = 0; h < kPer; h++) dst[q][h] = -1;
      add[q] = 0u;
      if (i < i1) {                                       // (block-uniform)
        const char* blob = (const char*)f.tptr[i * kR3Ptrs + 5];
        const int rel = f.koff[i] - k_lo, np = kCs + f.koff[i + 1] - f.koff[i];
        add[q] = (unsigned)rel * 0x10001u;                // u16 pairs: no carry, the sums stay below 65536
#pragma unroll
        for (int h = 0; h < kPer; h++) {
          const int j = tid + h * NT;
          if (j < np) {
            v[q][h] = gload<g_u32x4>(blob + (size_t)j * 16);
            dst[q][h] = j < kCs ? (i - i0) * kCs + j : rec0 + rel + (j - kCs);
          }
        }
      }
    }
#pragma unroll
    for (int q = 0; q < kKf; q++)
#pragma unroll
      for (int h = 0; h < kPer; h++)
        if (dst[q][h] >= 0) {
          const unsigned a = dst[q][h] < rec0 ? add[q] : 0u;
          out[dst[q][h]] = make_uint4(v[q][h].x + a, v[q][h].y + a, v[q][h].z + a, v[q][h].w + a);
        }
    for (int i = ib; i < min(ib + kKf, i1); i++) {        // keyframes beyond 2 x 256 pieces (more than 383 cells): the rest
      const char* blob = (const char*)f.tptr[i * kR3Ptrs + 5];
      const int rel = f.koff[i] - k_lo, np = kCs + f.koff[i + 1] - f.koff[i];
      for (int j = tid + kPer * NT; j < np; j += NT) {
        const g_u32x4 t = gload<g_u32x4>(blob + (size_t)j * 16);
        out[rec0 + rel + (j - kCs)] = make_uint4(t.x, t.y, t.z, t.w);      // (j >= 512 > kCs: records only)
      }
    }
  }
  __syncthreads();
}

// One association pass (n_scan_normal.cpp:213-318) at the pose xsrc; fills `dn` (which overwrites the staged tables); returns
// the number of blocks, or -1 when a keyframe's tables do not fit the LDS region at all (register_kernel's job).
// associate_fused's two passes with the grid edge fixed at kScanGrid; the keyframes are staged in as many groups as the
// region needs (one, unless the scans are unusually large).
template <int NT>
__device__ int r3_associate(const RegJob& job, const RegCommon& cm, int itr, const R3Lds& f, const double* xsrc, double* gl_dense,
                            Dense3& dn, int* ipart, int& iphase) {
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int last = job.n_scans - 1;
  const int n_src = gload<int>(job.scans[last].n_cells);
  const int n_pairs = last * n_src;
  const double curr_radius = (itr == 1) ? 2 * cm.par.radius : cm.par.radius;    // :220
  const double r2 = curr_radius * curr_radius;
  const float rwin = (float)curr_radius + 1e-3f;
  constexpr int G = kScanGrid;
  constexpr bool GM = NT != kReg3NW * 64;                  // the global match table exists in the large forms only (r3_carve)
  REG_T0();
  // Tsrctotar_i = Ttar_i^-1 * Tsrc  (:222), on the last wavefront, ahead of the first group's table loads.  The
  // rotation of the source pose by the polynomial sincos the LM loop uses for its evaluation points (an ulp or two from libm,
  // the level at which device and host libm differ anyway; the chain is a tenth of libm's).
  auto source_to_keyframes = [&]() {
    const int k0 = tid - (NT - 64);
    if (k0 >= 0 && k0 < last) {
      const double* k = f.kf + k0 * 12;
      const Aff2 Ttar{k[0], k[1], k[2], k[3], k[4], k[5]};
      double sn, cs;
      sincos_pose<true>(xsrc[2], &sn, &cs);
      const Aff2 Tst = aff_mul(aff_inv(Ttar), Aff2{cs, -sn, sn, cs, xsrc[0], xsrc[1]});
      double* o = f.kf + k0 * 12 + 6;
      o[0] = Tst.l0; o[1] = Tst.l1; o[2] = Tst.l2; o[3] = Tst.l3; o[4] = Tst.t0; o[5] = Tst.t1;
    }
  };
  int accepted = 0;
  int pend_p = -1, pend_i = 0, pend_best = -1;             // the pair whose normal gate is still open (see the search loop)
  double2 pend_ns = make_double2(0.0, 0.0), pend_nt = make_double2(0.0, 0.0);
  auto gate_pending = [&]() {
    if (pend_p < 0) return;
    int m = -1;
    if (pend_best >= 0) {
      const double* T = f.kf + pend_i * 12 + 6;
      const double nsx = T[0] * pend_ns.x + T[1] * pend_ns.y, nsy = T[2] * pend_ns.x + T[3] * pend_ns.y;
      if (fmax(nsx * pend_nt.x + nsy * pend_nt.y, 0.0) > cm.angle_outlier) m = pend_best;    // :244-245
    }
    if (GM && f.gmatch) gstore<unsigned short>(f.gmatch + pend_p, (unsigned short)m);   // (block-uniform)
    else f.match[pend_p] = (unsigned short)m;
    accepted += (m >= 0);
  };
  for (int i0 = 0; i0 < last;) {
    int i1 = i0, bytes = 0;                                // the keyframes [i0, i1) whose tables fit the region together
    while (i1 < last) {
      const int b = kScanGridStartPad * 2 + (f.koff[i1 + 1] - f.koff[i1]) * 16;
      if (bytes + b > f.region) break;
      bytes += b; i1++;
    }
    if (i1 == i0) return -1;                               // (block-uniform)
    if (i0 > 0) __syncthreads();                           // the previous group's readers are done
    if (i0 == 0) source_to_keyframes();                    // (in the shadow of the table loads it kept 25 more VGPRs live and bought nothing)
    r3_restage<NT>(f, i0, i1);                             // (its barrier also publishes the transforms)
    REG_TACC(0);
    const float4* txyi = (const float4*)(f.cstart + (size_t)(i1 - i0) * kScanGridStartPad);
    // this thread's pairs p = tid + 256 k (the SAME pairs in every group layout and in pass 2) that fall into the group
    const int lo = i0 * n_src, hi = i1 * n_src;
    int p = lo + ((tid - lo) & (NT - 1));
    int i = i0, s = p - lo;
    while (s >= n_src && i < i1) { s -= n_src; i++; }
    for (; p < hi; p += NT) {
      const double* T = f.kf + i * 12 + 6;
      const double2 u = f.smean[s];
      const double px = T[0] * u.x + T[1] * u.y + T[4];
      const double py = T[2] * u.x + T[3] * u.y + T[5];
      const float qx = (float)px, qy = (float)py;                               // pointnormal.cpp:240-242
      const float4 gg = f.ggeo[i];
      const int cx0 = min(G - 1, max(0, (int)floorf((qx - rwin - gg.x) * gg.z))), cx1 = min(G - 1, max(0, (int)floorf((qx + rwin - gg.x) * gg.z)));
      const int cy0 = min(G - 1, max(0, (int)floorf((qy - rwin - gg.y) * gg.z))), cy1 = min(G - 1, max(0, (int)floorf((qy + rwin - gg.y) * gg.z)));
      // exact 1-NN (FLANN L2_Simple float distance, lowest index on ties): one unsigned minimum over (bits(d^2) << 32 | index)
      unsigned long long bestkey = ~0ull;
      auto visit = [&](const float4 c) {
        const float dx = __fsub_rn(qx, c.x), dy = __fsub_rn(qy, c.y);
        const float d = __fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy));
        const unsigned long long key = ((unsigned long long)__float_as_uint(d) << 32) | (unsigned)__float_as_int(c.z);
        bestkey = key < bestkey ? key : bestkey;
      };
      const unsigned short* cs = f.cstart + (i - i0) * kScanGridStartPad;
      auto scan_run = [&](int qb, int qe) {
        for (int q = qb; q < qe; q += 2) {
          const float4 ca = txyi[q], cb = txyi[min(q + 1, qe - 1)];
          visit(ca); visit(cb);
        }
      };
      {
        const int r1 = min(cy0 + 1, cy1), r2c = min(cy0 + 2, cy1);
        const int b0 = cs[cy0 * G + cx0], e0 = cs[cy0 * G + cx1 + 1];
        const int b1 = cs[r1 * G + cx0], e1 = cs[r1 * G + cx1 + 1];
        const int b2 = cs[r2c * G + cx0], e2 = cs[r2c * G + cx1 + 1];
        scan_run(b0, e0);
        if (cy1 > cy0) scan_run(b1, e1);
        if (cy1 > cy0 + 1) scan_run(b2, e2);
        for (int cy = cy0 + 3; cy <= cy1; cy++) scan_run((int)cs[cy * G + cx0], (int)cs[cy * G + cx1 + 1]);
      }
      const int best = bestkey == ~0ull ? -1 : (int)(unsigned)(bestkey & 0xFFFFFFFFu);
      const float bestd = __uint_as_float((unsigned)(bestkey >> 32));
      // The normal gate of THIS pair is decided one pair later: its two normals come from global memory (L2), the target's
      // address only known now, and waiting for them here exposed an L2 round trip per pair -- the next pair's search
      // covers it instead.
      gate_pending();
      pend_p = p; pend_i = i;
      pend_best = (best >= 0 && (double)bestd < r2) ? best : -1;                // pointnormal.cpp:250
      if (pend_best >= 0) {
        pend_ns = gload_d2(job.scans[last].normal + s);
        pend_nt = gload_d2((const double2*)f.tptr[i * kR3Ptrs + 1] + best);
      }
      s += NT;
      while (s >= n_src && i < i1) { s -= n_src; i++; }
    }
    gate_pending();
    pend_p = -1;
    REG_TACC(1);
    i0 = i1;
  }
  if (GM && f.gmatch) __threadfence_block();              // a thread reads back its OWN entries below: stores before loads
  const int incl = wave_incl_scan_i32(accepted);
  int base = incl - accepted, total;
  {
    int* buf = ipart + iphase * kRegMaxNW;
    if (lane == 63) buf[wave] = incl;
    __syncthreads();                                       // also: every reader of the tables is done -- the dense arrays may overwrite them
    for (int wv = 0; wv < wave; wv++) base += buf[wv];
    int tt = buf[0];
#pragma unroll
    for (int wv = 1; wv < NT / 64; wv++) tt += buf[wv];
    total = __builtin_amdgcn_readfirstlane(tt);
    iphase ^= 1;
  }
  dn.p = f.dense; dn.cap = f.dense_cap; dn.sidx = (unsigned short*)(f.dense + (size_t)cm.dense_fields * f.dense_cap); dn.smean = f.smean;
  dn.gp = gl_dense; dn.gcap = (size_t)cm.slots_cap; dn.gsidx = (int*)(gl_dense + (size_t)cm.dense_fields * cm.slots_cap);
  dn.n = total;
  const size_t cap = (size_t)dn.cap, gcap = dn.gcap;
  REG_TACC(2);
  {
    struct Gathered {
      int best, i, s, tns, sns;
      double2 nt, tm, ns;
      double tsc, ssc;
      double4 S;
    };
    const ScanView& srcv = job.scans[last];
    auto match_at = [&](int p) -> int {
      if (p >= n_pairs) return 0xFFFF;
      return (GM && f.gmatch) ? (int)gload<unsigned short>(f.gmatch + p) : (int)f.match[p];
    };
    auto gather = [&](int best, int i, int s) {
      Gathered g;
      g.best = best;
      if (g.best == 0xFFFF) g.best = -1;
      g.i = i; g.s = s;
      if (g.best >= 0) {
        const void* const* tp = f.tptr + i * kR3Ptrs;
        g.nt = gload_d2((const double2*)tp[1] + g.best);
        g.tm = gload_d2((const double2*)tp[0] + g.best);
        g.tns = gload<int>((const int32_t*)tp[2] + g.best);
        g.tsc = gload<double>((const double*)tp[3] + g.best);
        g.ns = gload_d2(srcv.normal + s);
        g.sns = gload<int>(srcv.nsamples + s);
        g.ssc = gload<double>(srcv.scale + s);
        if (cm.par.cost == CFEAR_P2D) g.S = gload_d4((const double4*)tp[4] + g.best);
      }
      return g;
    };
    int c = base, i = 0, s = tid;
    while (s >= n_src && i < last) { s -= n_src; i++; }
    int m_next = GM ? match_at(tid + NT) : 0;             // the match two rounds ahead is in flight (global table: an L2 round trip)
    Gathered cur = gather(match_at(tid), i, s);
    for (int p = tid; p < n_pairs; p += NT) {
      s += NT;
      while (s >= n_src && i < last) { s -= n_src; i++; }
      int m_use;
      if (GM) { m_use = m_next; m_next = match_at(p + 2 * NT); }
      else m_use = match_at(p + NT);
      const Gathered nxt = gather(m_use, i, s);
      if (cur.best >= 0) {
        const double* K = f.kf + cur.i * 12;              // Ttar
        const double* T = K + 6;                          // Tsrctotar
        const double2 nt = cur.nt, tm = cur.tm, ns = cur.ns;
        const double nsx = T[0] * ns.x + T[1] * ns.y, nsy = T[2] * ns.x + T[3] * ns.y;
        const double direction_similarity = fmax(nsx * nt.x + nsy * nt.y, 0.0);   // :244
        const double w = get_weight(cm.par.weight_opt, (double)cur.sns, (double)cur.tns,
                                    direction_similarity, cur.ssc, cur.tsc);       // :247-253, :273
        double e[6];                                       // tmx, tmy, w, a0, a1, a2
        e[0] = K[0] * tm.x + K[1] * tm.y + K[4];                                  // Ttar * tar_mean
        e[1] = K[2] * tm.x + K[3] * tm.y + K[5];
        e[2] = w; e[3] = 0.0; e[4] = 0.0; e[5] = 0.0;
        if (cm.par.cost == CFEAR_P2D) {                                           // :288-297
          const double4 S = cur.S;
          const double a00 = K[0] * S.x + K[1] * S.z, a01 = K[0] * S.y + K[1] * S.w;
          const double a10 = K[2] * S.x + K[3] * S.z, a11 = K[2] * S.y + K[3] * S.w;
          const double c00 = (cm.par.regularization + (a00 * K[0] + a01 * K[1])) * cm.par.cov_scale;
          const double c01 = (0.0 + (a00 * K[2] + a01 * K[3])) * cm.par.cov_scale;
          const double c10 = (0.0 + (a10 * K[0] + a11 * K[1])) * cm.par.cov_scale;
          const double c11 = (cm.par.regularization + (a10 * K[2] + a11 * K[3])) * cm.par.cov_scale;
          const double det = c00 * c11 - c10 * c01, invdet = 1.0 / det;
          const double i00 = c11 * invdet, i10 = -c10 * invdet, i11 = c00 * invdet;
          const double l00 = sqrt(i00), l10 = i10 / l00;
          e[3] = l00; e[4] = l10; e[5] = sqrt(i11 - l10 * l10);
        } else if (cm.par.cost == CFEAR_P2L) {
          e[3] = K[0] * nt.x + K[1] * nt.y;                                       // Ttar.linear() * tar_normal
          e[4] = K[2] * nt.x + K[3] * nt.y;
        }
        const int nf = cm.dense_fields;
        if (c < (int)cap) {
          dn.sidx[c] = (unsigned short)cur.s;
          for (int k = 0; k < nf; k++) dn.p[k * cap + c] = e[k];
        } else {
          const size_t g = (size_t)c - cap;
          gstore<int>(dn.gsidx + g, cur.s);
          for (int k = 0; k < nf; k++) gstore<double>(dn.gp + k * gcap + g, e[k]);
        }
        c++;
      }
      cur = nxt;
    }
  }
  if (total > (int)cap) __threadfence_block();             // the tail went to global memory
  __syncthreads();
  REG_TACC(3);
  return total;
}

// cost, gradient and Gauss-Newton matrix of all correspondences at x (eval_all over Dense3); wavefront 0 receives the sums
template <int NT, int COST, int LOSS>
__device__ void eval_all3(const RegCommon& cm, const Dense3& dn, const double x[3], double c, double s, double out[10], double* part,
                          int& phase) {
  double acc[10];
#pragma unroll
  for (int k = 0; k < 10; k++) acc[k] = 0.0;
  const size_t cap = (size_t)dn.cap, gcap = dn.gcap;
  REG_T0();
  for (int i = threadIdx.x; i < dn.n; i += NT) {
    double tmx, tmy, w, a0 = 0.0, a1 = 0.0, a2 = 0.0;
    int si;
    if (i < (int)cap) {
      si = (int)dn.sidx[i];
      tmx = dn.p[i]; tmy = dn.p[cap + i]; w = dn.p[2 * cap + i];
      if (COST != CFEAR_P2P) { a0 = dn.p[3 * cap + i]; a1 = dn.p[4 * cap + i]; }
      if (COST == CFEAR_P2D) a2 = dn.p[5 * cap + i];
    } else {
      const size_t g = (size_t)i - cap;
      si = gload<int>(dn.gsidx + g);
      tmx = gload<double>(dn.gp + g); tmy = gload<double>(dn.gp + gcap + g); w = gload<double>(dn.gp + 2 * gcap + g);
      if (COST != CFEAR_P2P) { a0 = gload<double>(dn.gp + 3 * gcap + g); a1 = gload<double>(dn.gp + 4 * gcap + g); }
      if (COST == CFEAR_P2D) a2 = gload<double>(dn.gp + 5 * gcap + g);
    }
    const double2 sm = dn.smean[si];
    eval_slot<COST, LOSS, true>(cm.par, sm.x, sm.y, tmx, tmy, a0, a1, a2, w, x[0], x[1], c, s, acc);
  }
  REG_TACC(4);
  block_reduce10<NT / 64, false>(acc, part, phase);
  REG_TACC(5);
#pragma unroll
  for (int k = 0; k < 10; k++) out[k] = acc[k];
}

// One round of ceres::Solve's trust-region loop on wavefront 0 (same statements as lm_solve): judges the candidate that
// was just evaluated (cnd, when have_cnd), then produces the next candidate or the stop flag.  st = the LDS state block.
__device__ __forceinline__ void r3_lds_order() {          // lane 0's LDS stores above, the wavefront's loads below (one wavefront: the
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");   // LDS executes its operations in order; this orders the compiler)
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
}
__device__ __forceinline__ void r3_lm_round(double* st, const double cnd[10], const bool have_cnd, const int max_iter) {
  const double function_tolerance = 1e-6, gradient_tolerance = 1e-10, parameter_tolerance = 1e-8;
  const double min_relative_decrease = 1e-3, min_lm_diagonal = 1e-6, max_lm_diagonal = 1e32;
  const double max_radius = 1e16, min_radius = 1e-32;
  int* si = (int*)(st + S_INTS);
  const bool l0 = (threadIdx.x & 63) == 0;
  // The scalars of the loop stay in registers for the round; the vectors (x, the candidate, gradient and Gauss-Newton matrix at
  // x, scaling, LM diagonal) are read from the block where a statement needs them and written back where one changes them.
  double x_cost = st[S_XCOST], x_norm = st[S_XNORM], gradient_max_norm = st[S_GMAX], radius = st[S_RADIUS];
  double decrease_factor = st[S_DEC], min_iter_cost = st[S_MINCOST], model_cost_change = st[S_MODEL];
  double it_cost = st[S_ITCOST], it_rel = st[S_ITREL], last_rel = st[S_LASTREL];
  int iteration = si[SI_ITER], num_consecutive_invalid_steps = si[SI_INVALID], n_pushed = si[SI_PUSHED];
  bool reuse_diagonal = si[SI_REUSE] != 0, it_success = si[SI_ITSUCC] != 0, usable = si[SI_USABLE] != 0;
  int done = 0;
  bool proceed = true;
  if (have_cnd) {
    const double cand_cost = cnd[0];
    const double d0 = st[S_X] - st[S_CAND], d1 = st[S_X + 1] - st[S_CAND + 1], d2 = st[S_X + 2] - st[S_CAND + 2];
    const double step_norm2 = d0 * d0 + d1 * d1 + d2 * d2;
    const double cost_change = x_cost - cand_cost;
    const double step_bound = parameter_tolerance * (x_norm + parameter_tolerance);
    if (step_norm2 <= step_bound * step_bound) { done = 1; proceed = false; }      // ||step|| <= tolerance (||x|| + tolerance)
    else if (fabs(cost_change) <= function_tolerance * x_cost) { done = 1; proceed = false; }
    else {
      it_rel = cost_change / model_cost_change;
      if (it_rel > min_relative_decrease) {
        const double c0 = st[S_CAND], c1 = st[S_CAND + 1], c2 = st[S_CAND + 2];
        x_norm = sqrt_newton(c0 * c0 + c1 * c1 + c2 * c2);
        if (l0) {
          st[S_X] = c0; st[S_X + 1] = c1; st[S_X + 2] = c2;
#pragma unroll
          for (int k = 1; k < 10; k++) st[S_CUR + k - 1] = cnd[k];
        }
        x_cost = cand_cost;
        gradient_max_norm = fmax(fabs(cnd[1]), fmax(fabs(cnd[2]), fabs(cnd[3])));
        it_cost = x_cost; it_success = true;
        const double q = 2.0 * it_rel - 1.0;
        radius = radius * rcp_newton(fmax(1.0 / 3.0, 1.0 - q * q * q));
        radius = fmin(max_radius, radius);
        decrease_factor = 2.0; reuse_diagonal = false;
      } else {
        it_cost = cand_cost; it_success = false;
        radius = radius / decrease_factor; decrease_factor *= 2.0; reuse_diagonal = true;
      }
    }
    r3_lds_order();
  }
  while (proceed) {
    // FinalizeIterationAndCheckIfMinimizerCanContinue
    n_pushed++;
    last_rel = it_rel;
    min_iter_cost = fmin(min_iter_cost, it_cost);
    if (iteration >= max_iter || (it_success && gradient_max_norm <= gradient_tolerance) || radius <= min_radius) {
      done = 1;
      break;
    }
    iteration++;
    it_cost = 0.0; it_rel = 0.0; it_success = false;
    const double scale[3] = {st[S_SCALE], st[S_SCALE + 1], st[S_SCALE + 2]};
    double gs[3], Hs[9];
    {
      const double* cur = st + S_CUR - 1;                  // cur[1 .. 9] = g, H upper triangle
      gs[0] = cur[1] * scale[0]; gs[1] = cur[2] * scale[1]; gs[2] = cur[3] * scale[2];
      Hs[0] = cur[4] * scale[0] * scale[0]; Hs[1] = cur[5] * scale[0] * scale[1]; Hs[2] = cur[6] * scale[0] * scale[2];
      Hs[3] = Hs[1]; Hs[4] = cur[7] * scale[1] * scale[1]; Hs[5] = cur[8] * scale[1] * scale[2];
      Hs[6] = Hs[2]; Hs[7] = Hs[5]; Hs[8] = cur[9] * scale[2] * scale[2];
    }
    double diagonal[3];
    if (!reuse_diagonal) {
      diagonal[0] = fmin(fmax(Hs[0], min_lm_diagonal), max_lm_diagonal);
      diagonal[1] = fmin(fmax(Hs[4], min_lm_diagonal), max_lm_diagonal);
      diagonal[2] = fmin(fmax(Hs[8], min_lm_diagonal), max_lm_diagonal);
      if (l0) { st[S_DIAG] = diagonal[0]; st[S_DIAG + 1] = diagonal[1]; st[S_DIAG + 2] = diagonal[2]; }
    } else {
      diagonal[0] = st[S_DIAG]; diagonal[1] = st[S_DIAG + 1]; diagonal[2] = st[S_DIAG + 2];
    }
    double A[9];
#pragma unroll
    for (int k = 0; k < 9; k++) A[k] = Hs[k];
    const double inv_radius = rcp_newton(radius);
#pragma unroll
    for (int k = 0; k < 3; k++) A[k * 3 + k] += diagonal[k] * inv_radius;
    double y[3], step[3] = {0, 0, 0};
    const bool solved = chol3_solve(A, gs, y);
    reuse_diagonal = true;
    bool step_is_valid = false;
    model_cost_change = 0.0;
    if (solved) {
      step[0] = -y[0]; step[1] = -y[1]; step[2] = -y[2];
      const double sg = step[0] * gs[0] + step[1] * gs[1] + step[2] * gs[2];
      const double hs0 = Hs[0] * step[0] + Hs[1] * step[1] + Hs[2] * step[2];
      const double hs1 = Hs[3] * step[0] + Hs[4] * step[1] + Hs[5] * step[2];
      const double hs2 = Hs[6] * step[0] + Hs[7] * step[1] + Hs[8] * step[2];
      model_cost_change = -sg - (step[0] * hs0 + step[1] * hs1 + step[2] * hs2) / 2.0;
      step_is_valid = model_cost_change > 0.0;
    }
    if (!step_is_valid) {
      if (++num_consecutive_invalid_steps >= 5) { usable = false; done = 1; break; }
      radius = radius / decrease_factor; decrease_factor *= 2.0; reuse_diagonal = true;
      it_cost = x_cost; it_success = false; it_rel = 0.0;
      r3_lds_order();                                      // (the diagonal stored above is read again by the next turn)
      continue;
    }
    num_consecutive_invalid_steps = 0;
    double sn, cs;
    const double c2 = st[S_X + 2] + step[2] * scale[2];
    sincos_pose<true>(c2, &sn, &cs);
    if (l0) {
      st[S_CAND] = st[S_X] + step[0] * scale[0]; st[S_CAND + 1] = st[S_X + 1] + step[1] * scale[1]; st[S_CAND + 2] = c2;
      st[S_COS] = cs; st[S_SIN] = sn;
    }
    break;
  }
  if (l0) {
    st[S_XCOST] = x_cost; st[S_XNORM] = x_norm; st[S_GMAX] = gradient_max_norm; st[S_RADIUS] = radius; st[S_DEC] = decrease_factor;
    st[S_MINCOST] = min_iter_cost; st[S_MODEL] = model_cost_change; st[S_ITCOST] = it_cost; st[S_ITREL] = it_rel;
    st[S_LASTREL] = last_rel;
    if (done) st[S_FINAL] = fmin(st[S_INIT], min_iter_cost);          // solver.cc SetSummaryFinalCost
    si[SI_ITER] = iteration; si[SI_REUSE] = reuse_diagonal; si[SI_INVALID] = num_consecutive_invalid_steps;
    si[SI_PUSHED] = n_pushed; si[SI_USABLE] = usable; si[SI_DONE] = done; si[SI_ITSUCC] = it_success;
  }
}

// ceres::Solve for register3_kernel: lm_solve with the state in LDS.  The start pose is st[S_OUTER .. +3); the result is left
// in the state block (S_X, S_FINAL, S_LASTREL, SI_PUSHED, SI_USABLE).  Block-wide collective.
template <int NT, int COST, int LOSS>
__device__ void lm_solve3(const RegCommon& cm, const Dense3& dn, const int max_iter, double* part, int& phase, double* st) {
  const bool w0 = (threadIdx.x >> 6) == 0;
  int* si = (int*)(st + S_INTS);
  double cnd[10];
  {
    const double x[3] = {st[S_OUTER], st[S_OUTER + 1], st[S_OUTER + 2]};
    double s0, c0;
    sincos_pose<true>(x[2], &s0, &c0);
    eval_all3<NT, COST, LOSS>(cm, dn, x, c0, s0, cnd, part, phase);
    if (w0) {
      if ((threadIdx.x & 63) == 0) {
        st[S_X] = x[0]; st[S_X + 1] = x[1]; st[S_X + 2] = x[2]; st[S_XCOST] = cnd[0];
#pragma unroll
        for (int k = 1; k < 10; k++) st[S_CUR + k - 1] = cnd[k];
        st[S_SCALE] = 1.0 / (1.0 + sqrt(cnd[4]));            // jacobi scaling from iteration 0
        st[S_SCALE + 1] = 1.0 / (1.0 + sqrt(cnd[7]));
        st[S_SCALE + 2] = 1.0 / (1.0 + sqrt(cnd[9]));
        st[S_DIAG] = 0.0; st[S_DIAG + 1] = 0.0; st[S_DIAG + 2] = 0.0;
        st[S_XNORM] = sqrt_newton(x[0] * x[0] + x[1] * x[1] + x[2] * x[2]);
        st[S_GMAX] = fmax(fabs(cnd[1]), fmax(fabs(cnd[2]), fabs(cnd[3])));
        st[S_RADIUS] = 1e4; st[S_DEC] = 2.0; st[S_MINCOST] = cnd[0]; st[S_MODEL] = 0.0; st[S_ITCOST] = cnd[0]; st[S_ITREL] = 0.0;
        st[S_CAND] = x[0]; st[S_CAND + 1] = x[1]; st[S_CAND + 2] = x[2];
        st[S_INIT] = cnd[0]; st[S_LASTREL] = 0.0; st[S_FINAL] = cnd[0];
        si[SI_ITER] = 0; si[SI_REUSE] = 0; si[SI_INVALID] = 0; si[SI_PUSHED] = 0; si[SI_USABLE] = 1; si[SI_DONE] = 0; si[SI_ITSUCC] = 1;
      }
      r3_lds_order();                                       // lane 0's block is read by the whole wavefront below
      r3_lm_round(st, cnd, false, max_iter);
    }
  }
  for (;;) {
    REG_T0();
    __syncthreads();
    REG_TACC(21);
    if (__builtin_amdgcn_readfirstlane(si[SI_DONE])) break;
    const double cand[3] = {st[S_CAND], st[S_CAND + 1], st[S_CAND + 2]};
    const double cs = st[S_COS], sn = st[S_SIN];
    eval_all3<NT, COST, LOSS>(cm, dn, cand, cs, sn, cnd, part, phase);        // its barrier also orders the state block
    REG_TACC(22);
    if (w0) r3_lm_round(st, cnd, true, max_iter);
    REG_TACC(16);
  }
}

template <int NW, int COST, int LOSS>
__global__ __launch_bounds__(NW * 64, NW == kReg3NW ? 3 : 4) void register3_kernel(const RegJob* __restrict__ jobs, const RegCommon cm) {
  constexpr int NT = NW * 64;
#ifdef CFEAR_REG_TIMING
  const long long t_total0 = __builtin_readcyclecounter();
#endif
  REG_T0();
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  double* part = (double*)smem;
  int* ipart = (int*)(smem + kRegIpartOff);
  const RegJob& job = *(const RegJob*)((const char*)jobs + (size_t)blockIdx.x * cm.job_stride);
  cfear_reg_result* res = cm.results + blockIdx.x;
  if (cm.only_deferred && res->status != kRegDeferred) return;      // the large form: only what the launches before left
  const int last = job.n_scans - 1;
  const int n_src = gload<int>(job.scans[last].n_cells);
  int sum_tar = 0, max_tar = 0;
  for (int i = 0; i < last; i++) { const int n = gload<int>(job.scans[i].n_cells); sum_tar += n; max_tar = max(max_tar, n); }
  const int n_pairs = last * n_src;
  R3Lds fl;
  // (the head of the job's scratch -- register_kernel's slot arrays, 52 bytes per pair -- is free in this kernel)
  bool ok = n_pairs <= cm.slots_cap && n_src < 65536 &&
            r3_carve(smem, cm.lds_total, last, sum_tar, max_tar, n_src, n_pairs, cm.dense_fields,
                     NW != kReg3NW ? (unsigned short*)(cm.scratch + (size_t)blockIdx.x * cm.scratch_stride) : nullptr, fl);
  ok = ok && kScanGridStartPad * 2 + max_tar * 16 <= fl.region;   // every keyframe's tables fit the region on their own
  // ... and the registration is one this kernel is GOOD at: the keyframes' tables in at most two groups and room for 40 % of
  // the pairs in the LDS arrays (scans of up to ~600 cells).  Beyond that (dense scenes: 1 400 cells per scan) it would
  // restage keyframe by keyframe and evaluate mostly from global memory -- register_kernel's 80 KB / 160 KB forms are faster
  // (1024 dense streams: 0.75 ms in this kernel before every registration was redone by register_large anyway).
  if (ok && !cm.r3_take_all)
    ok = last * kScanGridStartPad * 2 + sum_tar * 16 <= 2 * fl.region && 5 * fl.dense_cap >= 2 * n_pairs;
  REG_TACC(8);
  if (ok) ok = r3_stage_once<NT>(job, fl, (int*)(smem + kR3StateOff + S_INTS * 8) + 7);   // (the state block's spare int)
  REG_TACC(9);
  if (!ok) {                                             // register_kernel takes it (launched behind this kernel)
    if (threadIdx.x == 0) { res->status = kRegDeferred; res->reserved = 0.0; }
    return;
  }
  double* gl_dense = (double*)(cm.scratch + (size_t)blockIdx.x * cm.scratch_stride + slots_bytes(cm.slots_cap));
  Dense3 dn;
  int phase = 0, iphase = 0;
  const int rpb = COST == CFEAR_P2L ? 1 : 2;
  // n_scan_normal.cpp:82-185.  The loop's own state (current pose, previous pose and score) lives in the LDS block too:
  // st[S_OUTER .. +3) = parameters.back(), +3 .. +6 = prev_par, +6 = prev_score.
  double* st = (double*)(smem + kR3StateOff);
  if (threadIdx.x == 0) {
#pragma unroll
    for (int k = 0; k < 3; k++) { st[S_OUTER + k] = job.poses[last][k]; st[S_OUTER + 3 + k] = job.poses[last][k]; }
    st[S_OUTER + 6] = DBL_MAX;
    st[S_FINAL] = 0.0; st[S_LASTREL] = 0.0;
  }
  __syncthreads();
  bool success = true;
  int itr = 1, lm_iters = 0, num_residuals = 0, fail_status = CFEAR_OK;
  for (itr = 1; itr <= cm.par.max_itr_association && success; itr++) {
    const int n_blocks = r3_associate<NT>(job, cm, itr, fl, st + S_OUTER, gl_dense, dn, ipart, iphase);
    if (n_blocks < 0) {                                           // (block-uniform)
      if (threadIdx.x == 0) { res->status = kRegDeferred; res->reserved = 0.0; }
      return;
    }
    num_residuals = n_blocks * rpb;
    success = num_residuals > 1;                                  // :368-369
    if (!success) { fail_status = CFEAR_ERR_TOO_FEW_RESIDUALS; break; }
    lm_solve3<NT, COST, LOSS>(cm, dn, cm.par.max_itr_solver, part, phase, st);
    const int* si = (const int*)(st + S_INTS);
    const int n_pushed = __builtin_amdgcn_readfirstlane(si[SI_PUSHED]);
    lm_iters += n_pushed - 1;
    success = __builtin_amdgcn_readfirstlane(si[SI_USABLE]) != 0;
    if (!success) fail_status = CFEAR_ERR_SOLVER;
    // n_scan_normal.cpp:117-149 (selects: see register_kernel); every thread derives the same decision, thread 0 stores it
    const double current_score = st[S_FINAL], prev_score = st[S_OUTER + 6];
    const double rel_improvement = (prev_score - current_score) / prev_score;
    const bool past_min = itr > cm.par.min_itr;
    const bool worse = past_min && (prev_score < current_score);
    const bool small_outer = past_min && !worse && (rel_improvement < cm.par.score_tolerance);
    const bool small_inner = past_min && !worse && !small_outer && (st[S_LASTREL] < cm.par.score_tolerance || n_pushed == 1);
    const bool stop = worse || small_outer || small_inner;
    double xo[3], pp[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
      const double xk = success ? st[S_X + k] : st[S_OUTER + k];
      pp[k] = st[S_OUTER + 3 + k];
      xo[k] = worse ? pp[k] : xk;
      pp[k] = stop ? pp[k] : xo[k];
    }
    __syncthreads();                                              // every thread has read the block
    if (threadIdx.x == 0) {
#pragma unroll
      for (int k = 0; k < 3; k++) { st[S_OUTER + k] = xo[k]; st[S_OUTER + 3 + k] = pp[k]; }
      st[S_OUTER + 6] = stop ? prev_score : current_score;
    }
    __syncthreads();
    if (stop) break;
  }
  if (threadIdx.x == 0) {
    res->pose[0] = st[S_OUTER]; res->pose[1] = st[S_OUTER + 1]; res->pose[2] = st[S_OUTER + 2];
    const double final_cost = st[S_FINAL];
    res->final_cost = final_cost;
    res->num_residuals = num_residuals;
    res->outer_iters = itr;
    res->lm_iters = lm_iters;
    res->last_relative_decrease = st[S_LASTREL];
    res->reserved = NW != kReg3NW ? 1.0 : 0.0;               // a large registration: tells the caller to keep the large launches on
    if (success) { res->score = final_cost / (double)num_residuals; res->status = CFEAR_OK; }   // :162
    else { res->score = 0.0; res->status = fail_status; }
#ifdef CFEAR_REG_TIMING
    if (blockIdx.x == 0) {
      printf("reg3 cycles: total %lld | sizes+carve %lld stage_once %lld | restage %lld nn+gate %lld scan %lld gather %lld | eval %lld reduce %lld round %lld barrier %lld | outer %d lm %d n %d\n",
             (long long)(__builtin_readcyclecounter() - t_total0), g_reg_t[8], g_reg_t[9], g_reg_t[0], g_reg_t[1], g_reg_t[2], g_reg_t[3], g_reg_t[4], g_reg_t[5],
             g_reg_t[16], g_reg_t[21], itr, lm_iters, num_residuals);
      for (int k = 0; k < 32; k++) g_reg_t[k] = 0;
    }
#endif
  }
}

// association only (GetCost / cfear_cost_prepare): slot arrays stay in the caller's scratch
__global__ __launch_bounds__(kRegThreads) void assoc_kernel(const RegJob* __restrict__ jobs, const RegCommon cm, int itr,
                                                            int32_t* n_blocks_out) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  int* ipart = (int*)(smem + kRegIpartOff);
  LdsTargets lt;
  lt.x = (float*)(smem + kRegFixedLds);
  lt.y = lt.x + cm.lds_targets;
  lt.idx = (int*)(lt.y + cm.lds_targets);
  const RegJob& job = *(const RegJob*)((const char*)jobs + (size_t)blockIdx.x * cm.job_stride);
  const int last = job.n_scans - 1;
  const int n_src = *job.scans[last].n_cells;
  int max_tar = 0;
  for (int i = 0; i < last; i++) max_tar = max(max_tar, *job.scans[i].n_cells);
  if (last * n_src > cm.slots_cap || max_tar > cm.lds_targets) {
    if (threadIdx.x == 0) n_blocks_out[blockIdx.x] = -1;
    return;
  }
  const Slots sl = slots_of(cm.scratch + (size_t)blockIdx.x * cm.scratch_stride, cm.slots_cap);
  int iphase = 0;
  const int mine = associate_all<4>(job, cm, job.poses[last], itr, sl, lt);
  const int n_blocks = block_sum_i32<4>(mine, ipart, iphase);
  if (threadIdx.x == 0) n_blocks_out[blockIdx.x] = n_blocks;
}

// per-slot evaluation at x for one job: raw residuals/Jacobians (Ceres CostFunction::Evaluate
// semantics), robustified residuals, and the reduced normal equations.
struct EvalOut {
  double* raw_r;       // [slots][2]
  double* raw_j;       // [slots][6]
  double* rob_r;       // [slots][2]
  double* neq;         // [10]: cost, g, H upper
};
__global__ __launch_bounds__(kRegThreads) void eval_kernel(const RegJob* __restrict__ jobs, const RegCommon cm, double x0,
                                                           double x1, double x2, EvalOut o) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  double* part = (double*)smem;
  const RegJob& job = jobs[0];
  const int last = job.n_scans - 1;
  const int n_src = *job.scans[last].n_cells;
  const int n_slots = last * n_src;
  const Slots sl = slots_of(cm.scratch, cm.slots_cap);
  double s, c;
  sincos(x2, &s, &c);
  const double2* smean = job.scans[last].mean;
  double acc[10];
#pragma unroll
  for (int k = 0; k < 10; k++) acc[k] = 0.0;
  for (int slot = threadIdx.x; slot < n_slots; slot += kRegThreads) {
    const double w = sl.w[slot];
    if (w < 0.0) continue;
    const double2 sm = smean[slot % n_src];
    const double tmx = sl.tmx[slot], tmy = sl.tmy[slot], a0 = sl.a0[slot], a1 = sl.a1[slot];
    const double a2 = cm.par.cost == CFEAR_P2D ? sl.a2[slot] : 0.0;
    eval_slot_rt<true>(cm.par, sm.x, sm.y, tmx, tmy, a0, a1, a2, w, x0, x1, c, s, acc);
    // raw values, recomputed exactly as eval_slot does
    const double sx = (c * sm.x + (-s) * sm.y) + x0, sy = (s * sm.x + c * sm.y) + x1;
    const double dx = -s * sm.x - c * sm.y, dy = c * sm.x - s * sm.y;
    double r0, r1 = 0.0, j[6] = {0, 0, 0, 0, 0, 0};
    if (cm.par.cost == CFEAR_P2L) {
      const double v0 = sx - tmx, v1 = sy - tmy;
      r0 = v0 * a0 + v1 * a1; j[0] = a0; j[1] = a1; j[2] = dx * a0 + dy * a1;
    } else if (cm.par.cost == CFEAR_P2P) {
      r0 = tmx - sx; r1 = tmy - sy; j[0] = -1.0; j[2] = -dx; j[4] = -1.0; j[5] = -dy;
    } else {
      const double v0 = sx - tmx, v1 = sy - tmy;
      r0 = a0 * v0 + 0.0 * v1; r1 = a1 * v0 + a2 * v1;
      j[0] = a0; j[2] = a0 * dx + 0.0 * dy; j[3] = a1; j[4] = a2; j[5] = a1 * dx + a2 * dy;
    }
    const double sq = (cm.par.cost == CFEAR_P2L) ? r0 * r0 : (r0 * r0 + r1 * r1);
    double rho0, rho1;
    loss_eval(cm.par.loss, cm.par.loss_limit, w, sq, rho0, rho1);
    const double sr = sqrt(rho1);
    if (o.raw_r) { o.raw_r[slot * 2] = r0; o.raw_r[slot * 2 + 1] = r1; }
    if (o.raw_j) for (int k = 0; k < 6; k++) o.raw_j[slot * 6 + k] = j[k];
    if (o.rob_r) { o.rob_r[slot * 2] = r0 * sr; o.rob_r[slot * 2 + 1] = r1 * sr; }
  }
  int phase = 0;
  block_reduce10<4>(acc, part, phase);
  if (threadIdx.x == 0 && o.neq) for (int k = 0; k < 10; k++) o.neq[k] = acc[k];
}

size_t reg_scratch_bytes(int slots_cap) { return slots_bytes(slots_cap) + ((size_t)slots_cap * 64 + 255) / 256 * 256; }

int check_params(cfear_ctx* ctx, const cfear_reg_params* p) {
  if (!p) return cfear_set_error(ctx, CFEAR_ERR_INVALID_ARGUMENT, "null parameters");
  if (p->cost < 0 || p->cost > 2 || p->loss < 0 || p->loss > 5 || p->weight_opt < 0 || p->weight_opt > 4)
    return cfear_set_error(ctx, CFEAR_ERR_INVALID_ARGUMENT, "bad cost/loss/weight option");
  if (p->max_itr_association < 1 || p->max_itr_solver < 0 || !(p->radius > 0))
    return cfear_set_error(ctx, CFEAR_ERR_INVALID_ARGUMENT, "bad iteration limits / radius");
  return CFEAR_OK;
}

}  // namespace

size_t cfear_reg_job_bytes() { return sizeof(RegJob); }
int cfear_reg_max_scans() { return kMaxScans; }

void cfear_reg_job_set_itr(void* job, int itr) { ((RegJob*)job)->itr = itr; }

// Writes the used prefix of a job record: reg_job_stride(n_scans) bytes at dst.
void cfear_reg_fill_job(void* dst, const ScanView* views, int n_scans, const double* poses_xyt) {
  RegJob* j = (RegJob*)dst;
  j->n_scans = n_scans;
  j->itr = 0;
  for (int i = 0; i < n_scans; i++) {
    j->poses[i][0] = poses_xyt[3 * i]; j->poses[i][1] = poses_xyt[3 * i + 1]; j->poses[i][2] = poses_xyt[3 * i + 2];
    j->scans[i] = views[i];
  }
}
size_t cfear_reg_job_stride(int max_scans) { return reg_job_stride(max_scans); }

// Enqueues the batched registration kernel: d_jobs [n_jobs] RegJob records (device), d_results
// [n_jobs] (device).  slots_cap bounds (n_scans-1)*n_src per job, lds_targets the largest target.
// LDS the fused association path needs for one registration (mirrors fused_carve): used by the host to decide whether
// a whole batch fits the compact geometry.
size_t cfear_reg_fused_lds_need(int n_scans, int sum_targets, int n_src, int cost) {
  const int last = n_scans - 1, G = reg_grid_dim(std::max(last, 1));
  const size_t n_pairs = (size_t)last * n_src, fields = (size_t)reg_dense_fields(cost);
  size_t b = kRegFixedLds + 16 * 12 * 8 + 80 + 16 * 5 * 8 + 16 * 16 + 16 * 4 * 4;
  b += (((size_t)last * G * G + 1) * 2 + 15) & ~(size_t)15;
  b += (size_t)sum_targets * 16 + (((size_t)n_src + 1) & ~(size_t)1) * 16 + ((n_pairs + 7) & ~(size_t)7) * 2 + 64;
  b += n_pairs * (fields * 8 + 4) + 64;                 // every pair matched: upper bound of the dense arrays
  return b;
}
// The same without the dense arrays (they may live in global scratch): what fused_carve needs to accept a registration.
static size_t reg_fused_lds_core(int n_scans, int sum_targets, int n_src, int cost) {
  const size_t n_pairs = (size_t)(n_scans - 1) * n_src, fields = (size_t)reg_dense_fields(cost);
  return cfear_reg_fused_lds_need(n_scans, sum_targets, n_src, cost) - (n_pairs * (fields * 8 + 4) + 64) + 1024;
}

int cfear_register_launch(cfear_ctx* ctx, const void* d_jobs, int n_jobs, const cfear_reg_params* par, int slots_cap,
                          int lds_targets, char* d_scratch, cfear_reg_result* d_results, const RegCostMode* mode,
                          size_t job_stride, bool compact, bool big_pass) {
  if (lds_targets > kMaxTargetsLds)
    return cfear_set_error(ctx, CFEAR_ERR_CAPACITY, "target scan with more than %d cells", kMaxTargetsLds);
  if (lds_targets < 1) lds_targets = 1;
  RegCommon cm{};
  cm.par = *par;
  cm.angle_outlier = std::cos(M_PI / 6.0);                                     // n_scan_normal.cpp:217
  cm.scratch = d_scratch;
  cm.scratch_stride = reg_scratch_bytes(slots_cap);
  cm.job_stride = job_stride ? job_stride : sizeof(RegJob);
  cm.slots_cap = slots_cap;
  cm.lds_targets = (lds_targets + 3) & ~3;
  cm.dense_fields = reg_dense_fields(par->cost);
  cm.results = d_results;
  cm.cost_only = mode ? 1 : 0;
  cm.n_samples = mode ? mode->n_samples : 0;
  cm.samples_per_axis = mode ? mode->samples_per_axis : 0;
  cm.only_deferred = 0; cm.r3_take_all = 0; cm.pad2 = 0;
  cm.xy_half = mode ? mode->xy_half : 0.0;
  cm.yaw_half = mode ? mode->yaw_half : 0.0;
  cm.prior = mode ? mode->prior : nullptr;
  // Geometry: one 256-thread workgroup and 80 KB of LDS per registration (two per CU), or -- when the caller has
  // checked that every registration of the batch fits -- the compact 128-thread / 40 KB form (four per CU).
  const size_t fixed = kRegFixedLds + reg_lds_targets_bytes(cm.lds_targets);
  const size_t budget = compact ? kRegLdsBudgetCompact : kRegLdsBudget;
  const size_t avail = fixed < budget ? budget - fixed : 0;
  cm.dense_cap_lds = (int)std::min<size_t>(avail / ((size_t)cm.dense_fields * 8 + 4), (size_t)slots_cap) & ~1;
  size_t lds = reg_lds_bytes(cm.lds_targets, cm.dense_cap_lds, cm.dense_fields);
  if (lds < budget) lds = budget;
  cm.lds_total = (uint32_t)lds;
  typedef void (*KernelFn)(const RegJob*, RegCommon);
  // compile-time specialisations: cost metric x {Huber, any other loss}
  const bool huber = par->loss == CFEAR_LOSS_HUBER;
  KernelFn fn;
  if (compact) {
    switch (par->cost) {
      case CFEAR_P2P: fn = huber ? register_kernel<kRegNWCompact, CFEAR_P2P, CFEAR_LOSS_HUBER> : register_kernel<kRegNWCompact, CFEAR_P2P, -1>; break;
      case CFEAR_P2L: fn = huber ? register_kernel<kRegNWCompact, CFEAR_P2L, CFEAR_LOSS_HUBER> : register_kernel<kRegNWCompact, CFEAR_P2L, -1>; break;
      default: fn = huber ? register_kernel<kRegNWCompact, CFEAR_P2D, CFEAR_LOSS_HUBER> : register_kernel<kRegNWCompact, CFEAR_P2D, -1>; break;
    }
  } else {
    switch (par->cost) {
      case CFEAR_P2P: fn = huber ? register_kernel<kRegNW, CFEAR_P2P, CFEAR_LOSS_HUBER> : register_kernel<kRegNW, CFEAR_P2P, -1>; break;
      case CFEAR_P2L: fn = huber ? register_kernel<kRegNW, CFEAR_P2L, CFEAR_LOSS_HUBER> : register_kernel<kRegNW, CFEAR_P2L, -1>; break;
      default: fn = huber ? register_kernel<kRegNW, CFEAR_P2D, CFEAR_LOSS_HUBER> : register_kernel<kRegNW, CFEAR_P2D, -1>; break;
    }
  }
  CFEAR_HIP_CHECK(ctx, hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  const bool two = big_pass && !mode && !compact;
  cm.big_mode = two ? 1 : 0;
  cm.lds_big = 160 * 1024 - 256;
  // A handful of registrations (a single sequence: one per frame) leave the chip empty: their latency is what counts, and
  // 8 wavefronts per registration cut it by a quarter (with the chip full they cost 23 % throughput instead).
  int threads = (compact ? kRegNWCompact : kRegNW) * 64;
  if (!compact && !mode && n_jobs <= 64) {
    switch (par->cost) {
      case CFEAR_P2P: fn = huber ? register_kernel<kRegNWBig, CFEAR_P2P, CFEAR_LOSS_HUBER> : register_kernel<kRegNWBig, CFEAR_P2P, -1>; break;
      case CFEAR_P2L: fn = huber ? register_kernel<kRegNWBig, CFEAR_P2L, CFEAR_LOSS_HUBER> : register_kernel<kRegNWBig, CFEAR_P2L, -1>; break;
      default: fn = huber ? register_kernel<kRegNWBig, CFEAR_P2D, CFEAR_LOSS_HUBER> : register_kernel<kRegNWBig, CFEAR_P2D, -1>; break;
    }
    CFEAR_HIP_CHECK(ctx, hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    threads = kRegNWBig * 64;
  }
  // Regular batches go through register3_kernel first (three workgroups per CU); register_kernel behind it takes what that
  // launch deferred (CFEAR_NO_REG3=1: register_kernel alone, for A/B runs).
  // ... for batches that fill the chip at two workgroups per CU: a smaller batch is a matter of latency, and register_kernel
  // (tables staged once, not per outer iteration) finishes a registration sooner (512 CA-CFAR streams: 0.34 against 0.38 ms).
  // CFEAR_REG3=1 forces it from 65 registrations on (tests), CFEAR_NO_REG3=1 switches it off (A/B runs); read per launch.
  int n_cu3 = 256;
  (void)hipDeviceGetAttribute(&n_cu3, hipDeviceAttributeMultiprocessorCount, ctx->device);
  const bool use3 = !mode && !compact && n_jobs > 64 && !getenv("CFEAR_NO_REG3") && (n_jobs > 2 * n_cu3 || getenv("CFEAR_REG3"));
  if (use3) {
    KernelFn f3;
    switch (par->cost) {
      case CFEAR_P2P: f3 = huber ? register3_kernel<kReg3NW, CFEAR_P2P, CFEAR_LOSS_HUBER> : register3_kernel<kReg3NW, CFEAR_P2P, -1>; break;
      case CFEAR_P2L: f3 = huber ? register3_kernel<kReg3NW, CFEAR_P2L, CFEAR_LOSS_HUBER> : register3_kernel<kReg3NW, CFEAR_P2L, -1>; break;
      default: f3 = huber ? register3_kernel<kReg3NW, CFEAR_P2D, CFEAR_LOSS_HUBER> : register3_kernel<kReg3NW, CFEAR_P2D, -1>; break;
    }
    // CFEAR_REG3_LDS_KB (tests): a smaller LDS so that ordinary scans exercise the keyframe groups and the global tail of the
    // dense arrays, which only unusually large registrations reach at the real size
    const char* e_kb = getenv("CFEAR_REG3_LDS_KB");
    const long kb = e_kb ? atol(e_kb) : 0;
    const size_t r3_lds = kb >= 8 && kb <= 52 ? (size_t)kb * 1024 : kReg3Lds;
    CFEAR_HIP_CHECK(ctx, hipFuncSetAttribute((const void*)f3, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kReg3Lds));
    RegCommon c3 = cm;
    c3.lds_total = (uint32_t)r3_lds;
    c3.big_mode = 0;
    c3.r3_take_all = e_kb ? 1 : 0;
    {
      ProfScope ps(ctx, "register");
      hipLaunchKernelGGL(f3, dim3(n_jobs), dim3(kReg3NW * 64), r3_lds, ctx->stream, (const RegJob*)d_jobs, c3);
    }
    CFEAR_HIP_CHECK(ctx, hipGetLastError());
    cm.only_deferred = 1;
    static const bool r3_stats = getenv("CFEAR_REG3_STATS") != nullptr;   // debug: how many registrations this launch left to register_kernel
    if (r3_stats) {
      std::vector<cfear_reg_result> h((size_t)n_jobs);
      if (hipMemcpyAsync(h.data(), d_results, h.size() * sizeof(cfear_reg_result), hipMemcpyDeviceToHost, ctx->stream) == hipSuccess &&
          hipStreamSynchronize(ctx->stream) == hipSuccess) {
        int nd = 0;
        for (const cfear_reg_result& r : h) nd += r.status == kRegDeferred;
        fprintf(stderr, "register3: %d of %d registrations deferred\n", nd, n_jobs);
      }
    }
  }
  {
    ProfScope ps(ctx, mode ? "get_cost" : (use3 ? "register_rest" : "register"));
    hipLaunchKernelGGL(fn, dim3(n_jobs, mode ? std::max(mode->blocks_per_job, 1) : 1), dim3(threads), lds, ctx->stream,
                       (const RegJob*)d_jobs, cm);
  }
  CFEAR_HIP_CHECK(ctx, hipGetLastError());
  if (two) {
    // Registrations the launches above deferred (dense scenes: ~1 400 cells per scan), one workgroup per CU with all of its
    // LDS.  First register3_kernel's large form: sixteen wavefronts (its phases need at most 128 VGPRs, so four fit a SIMD
    // where register_kernel's 236 allow two), the keyframes' prebuilt tables staged group by group, dense arrays for ~5 000
    // correspondences in LDS.  CFEAR_NO_REG3 leaves it out.
    if (!getenv("CFEAR_NO_REG3")) {
      // Half a CU each first (two workgroups per CU overlap each other's serial phases: the trust-region chain on one
      // wavefront, the barriers), then one per CU for what even a global match table does not fit into 80 KB.
      for (int form = 0; form < 2; form++) {
        KernelFn fl3;
        if (form == 0) {
          switch (par->cost) {
            case CFEAR_P2P: fl3 = huber ? register3_kernel<kReg3NWHalf, CFEAR_P2P, CFEAR_LOSS_HUBER> : register3_kernel<kReg3NWHalf, CFEAR_P2P, -1>; break;
            case CFEAR_P2L: fl3 = huber ? register3_kernel<kReg3NWHalf, CFEAR_P2L, CFEAR_LOSS_HUBER> : register3_kernel<kReg3NWHalf, CFEAR_P2L, -1>; break;
            default: fl3 = huber ? register3_kernel<kReg3NWHalf, CFEAR_P2D, CFEAR_LOSS_HUBER> : register3_kernel<kReg3NWHalf, CFEAR_P2D, -1>; break;
          }
        } else {
          switch (par->cost) {
            case CFEAR_P2P: fl3 = huber ? register3_kernel<kReg3NWLarge, CFEAR_P2P, CFEAR_LOSS_HUBER> : register3_kernel<kReg3NWLarge, CFEAR_P2P, -1>; break;
            case CFEAR_P2L: fl3 = huber ? register3_kernel<kReg3NWLarge, CFEAR_P2L, CFEAR_LOSS_HUBER> : register3_kernel<kReg3NWLarge, CFEAR_P2L, -1>; break;
            default: fl3 = huber ? register3_kernel<kReg3NWLarge, CFEAR_P2D, CFEAR_LOSS_HUBER> : register3_kernel<kReg3NWLarge, CFEAR_P2D, -1>; break;
          }
        }
        const int nw_l = form == 0 ? kReg3NWHalf : kReg3NWLarge;
        const size_t lds_l = form == 0 ? (size_t)(80 * 1024 - 256) : (size_t)cm.lds_big;
        CFEAR_HIP_CHECK(ctx, hipFuncSetAttribute((const void*)fl3, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        RegCommon c3 = cm;
        c3.lds_total = (uint32_t)lds_l;
        c3.big_mode = 0;
        c3.only_deferred = 1;
        c3.r3_take_all = 1;                               // (whatever it still cannot run stays deferred for the launch behind it)
        ProfScope ps(ctx, form == 0 ? "register_large" : "register_large16");
        hipLaunchKernelGGL(fl3, dim3(n_jobs), dim3(nw_l * 64), lds_l, ctx->stream, (const RegJob*)d_jobs, c3);
        CFEAR_HIP_CHECK(ctx, hipGetLastError());
      }
    }
    // ... then register_kernel with 8 wavefronts and the CU's LDS for what is left (scans without grid tables).  Everything
    // else returns at once (the caller switches these launches off when no large scans show up).
    KernelFn fb;
    switch (par->cost) {
      case CFEAR_P2P: fb = huber ? register_kernel<kRegNWBig, CFEAR_P2P, CFEAR_LOSS_HUBER> : register_kernel<kRegNWBig, CFEAR_P2P, -1>; break;
      case CFEAR_P2L: fb = huber ? register_kernel<kRegNWBig, CFEAR_P2L, CFEAR_LOSS_HUBER> : register_kernel<kRegNWBig, CFEAR_P2L, -1>; break;
      default: fb = huber ? register_kernel<kRegNWBig, CFEAR_P2D, CFEAR_LOSS_HUBER> : register_kernel<kRegNWBig, CFEAR_P2D, -1>; break;
    }
    CFEAR_HIP_CHECK(ctx, hipFuncSetAttribute((const void*)fb, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    cm.big_mode = 2;
    cm.lds_total = cm.lds_big;
    ProfScope ps(ctx, getenv("CFEAR_NO_REG3") ? "register_large" : "register_large_rest");
    hipLaunchKernelGGL(fb, dim3(n_jobs, 1), dim3(kRegNWBig * 64), (size_t)cm.lds_big, ctx->stream, (const RegJob*)d_jobs, cm);
    CFEAR_HIP_CHECK(ctx, hipGetLastError());
  }
  return CFEAR_OK;
}
size_t cfear_register_scratch_bytes(int slots_cap) { return reg_scratch_bytes(slots_cap); }

// ---- host-facing wrappers ----------------------------------------------------------------------
namespace {

struct JobSizes { int slots_cap = 1; int lds_targets = 1; size_t fused_need = 0, fused_core = 0; int cost = CFEAR_P2L; };

int gather_job(cfear_ctx* ctx, const cfear_scan* const* scans, int n_scans, const double* poses, unsigned char* dst,
               JobSizes& sz) {
  if (n_scans < 2 || n_scans > kMaxScans)
    return cfear_set_error(ctx, CFEAR_ERR_INVALID_ARGUMENT, "n_scans must be in [2,%d]", kMaxScans);
  ScanView views[kMaxScans];
  int sum_tar = 0, n_src = 0;
  for (int i = 0; i < n_scans; i++) {
    if (!scans[i]) return cfear_set_error(ctx, CFEAR_ERR_INVALID_ARGUMENT, "null scan handle");
    if (scans[i]->ctx != ctx) return cfear_set_error(ctx, CFEAR_ERR_INVALID_ARGUMENT, "scan belongs to another context");
    views[i] = scans[i]->view;
    const int nc = cfear_scan_size(scans[i]);
    if (nc < 0) return nc;
    if (i < n_scans - 1) { sz.lds_targets = std::max(sz.lds_targets, nc); sum_tar += nc; }
    else { sz.slots_cap = std::max(sz.slots_cap, (n_scans - 1) * std::max(nc, 1)); n_src = nc; }
  }
  sz.fused_need = std::max(sz.fused_need, cfear_reg_fused_lds_need(n_scans, sum_tar, n_src, sz.cost));
  sz.fused_core = std::max(sz.fused_core, reg_fused_lds_core(n_scans, sum_tar, n_src, sz.cost));
  cfear_reg_fill_job(dst, views, n_scans, poses);
  return CFEAR_OK;
}

}  // namespace

// The batch with its results left ON THE DEVICE (enqueued on the context's stream, not synchronised): d_out when given,
// otherwise the context's workspace; *d_used receives the pointer.  cfear_register_batch reads them back; the sharded
// entry hands them straight to the collective (shard.hip).
int cfear_register_batch_device(cfear_ctx* ctx, const cfear_reg_job* jobs, int32_t n_jobs, const cfear_reg_params* par,
                                cfear_reg_result* d_out, cfear_reg_result** d_used) {
  if (!ctx) return CFEAR_ERR_INVALID_ARGUMENT;
  if (!jobs || n_jobs < 0) return cfear_set_error(ctx, CFEAR_ERR_INVALID_ARGUMENT, "null argument");
  int rc = check_params(ctx, par);
  if (rc != CFEAR_OK) return rc;
  if (d_used) *d_used = d_out;
  if (n_jobs == 0) return CFEAR_OK;
  CFEAR_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  // job records are built in pinned memory: 2.3 KB each, so a 4096-candidate batch is a 9 MB upload that a pageable
  // source would stage synchronously at a fraction of the PCIe rate
  int max_scans = 2;
  for (int j = 0; j < n_jobs; j++) max_scans = std::max(max_scans, std::min(jobs[j].n_scans, kMaxScans));
  const size_t stride = reg_job_stride(max_scans);
  const size_t jb = (size_t)n_jobs * stride, rb = (size_t)n_jobs * sizeof(cfear_reg_result);
  unsigned char* hjobs = (unsigned char*)cfear_pinned(ctx, jb);
  if (!hjobs) return cfear_set_error(ctx, CFEAR_ERR_HIP, "pinned staging allocation failed");
  JobSizes sz;
  sz.cost = par->cost;
  for (int j = 0; j < n_jobs; j++) {
    rc = gather_job(ctx, jobs[j].scans, jobs[j].n_scans, jobs[j].poses_xyt, hjobs + (size_t)j * stride, sz);
    if (rc != CFEAR_OK) return rc;
  }
  // every registration of the batch fits 40 KB of LDS: four per CU instead of two (needs a batch that fills them)
  const bool compact = sz.fused_need <= kRegLdsBudgetCompact && n_jobs >= 512;
  const size_t sb = reg_scratch_bytes(sz.slots_cap) * (size_t)n_jobs;
  char* ws = (char*)cfear_workspace(ctx, 6, jb + rb + 512);
  char* scr = (char*)cfear_workspace(ctx, 7, sb);
  if (!ws || !scr) return cfear_set_error(ctx, CFEAR_ERR_HIP, "workspace allocation failed");
  char* d_jobs = ws;
  cfear_reg_result* d_res = d_out ? d_out : (cfear_reg_result*)(ws + (jb + 255) / 256 * 256);
  if (d_used) *d_used = d_res;
  CFEAR_HIP_CHECK(ctx, hipMemcpyAsync(d_jobs, hjobs, jb, hipMemcpyHostToDevice, ctx->stream));
  if (d_out) cfear_pinned_mark(ctx);                        // results stay on the device: nothing below waits for this copy
  // a registration whose keyframes do not fit the 80 KB association: second launch with a CU's whole LDS per workgroup
  const bool big = sz.fused_core > kRegLdsBudget;
  rc = cfear_register_launch(ctx, d_jobs, n_jobs, par, sz.slots_cap, sz.lds_targets, scr, d_res, nullptr, stride, compact, big);
  if (rc != CFEAR_OK) { (void)hipStreamSynchronize(ctx->stream); return rc; }
  return CFEAR_OK;
}

extern "C" int cfear_register_batch(cfear_ctx* ctx, const cfear_reg_job* jobs, int32_t n_jobs,
                                    const cfear_reg_params* par, cfear_reg_result* results) {
  if (!ctx) return CFEAR_ERR_INVALID_ARGUMENT;
  if (!jobs || !results || n_jobs < 0) return cfear_set_error(ctx, CFEAR_ERR_INVALID_ARGUMENT, "null argument");
  // results on the device: the records stay there, the launch is stream-ordered and not synchronised
  if (cfear_is_device_ptr(results)) return cfear_register_batch_device(ctx, jobs, n_jobs, par, results, nullptr);
  cfear_reg_result* d_res = nullptr;
  const int rc = cfear_register_batch_device(ctx, jobs, n_jobs, par, nullptr, &d_res);
  if (rc != CFEAR_OK || n_jobs == 0) return rc;
  CFEAR_HIP_CHECK(ctx, hipMemcpyAsync(results, d_res, (size_t)n_jobs * sizeof(cfear_reg_result), hipMemcpyDeviceToHost, ctx->stream));
  CFEAR_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  return CFEAR_OK;
}

// ---- candidate pairs among a table of scans (loop closure) ---------------------------------------------------------
struct cfear_scan_table {
  cfear_ctx* ctx = nullptr;
  ScanView* d_views = nullptr;          // [n] device
  std::vector<int32_t> n_cells;         // host copy of the cell counts (launch geometry)
};

namespace {
// candidate -> the job record register_kernel / register3_kernel read: scans {target, source}, poses {target, source guess}
__global__ __launch_bounds__(256) void expand_candidates_kernel(const ScanView* __restrict__ views, const cfear_candidate* __restrict__ cands,
                                                                int n, char* __restrict__ jobs, size_t stride) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const cfear_candidate c = cands[i];
  RegJob* j = (RegJob*)(jobs + (size_t)i * stride);
  j->n_scans = 2; j->itr = 0;
#pragma unroll
  for (int k = 0; k < 3; k++) { j->poses[0][k] = c.target_xyt[k]; j->poses[1][k] = c.source_xyt[k]; }
  j->scans[0] = views[c.target];
  j->scans[1] = views[c.source];
}
}  // namespace

extern "C" int cfear_scan_table_create(cfear_ctx* ctx, const cfear_scan* const* scans, int32_t n_scans, cfear_scan_table** out) {
  if (!ctx) return CFEAR_ERR_INVALID_ARGUMENT;
  if (!scans || !out || n_scans < 1) return cfear_set_error(ctx, CFEAR_ERR_INVALID_ARGUMENT, "null argument / empty table");
  *out = nullptr;
  CFEAR_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  std::vector<ScanView> views((size_t)n_scans);
  std::unique_ptr<cfear_scan_table> t(new cfear_scan_table());
  t->ctx = ctx;
  t->n_cells.resize((size_t)n_scans);
  for (int i = 0; i < n_scans; i++) {
    if (!scans[i]) return cfear_set_error(ctx, CFEAR_ERR_INVALID_ARGUMENT, "null scan handle");
    if (scans[i]->ctx != ctx) return cfear_set_error(ctx, CFEAR_ERR_INVALID_ARGUMENT, "scan belongs to another context");
    const int nc = cfear_scan_size(scans[i]);
    if (nc < 0) return nc;
    views[(size_t)i] = scans[i]->view;
    t->n_cells[(size_t)i] = nc;
  }
  CFEAR_HIP_CHECK(ctx, hipMalloc((void**)&t->d_views, views.size() * sizeof(ScanView)));
  if (hipMemcpyAsync(t->d_views, views.data(), views.size() * sizeof(ScanView), hipMemcpyHostToDevice, ctx->stream) != hipSuccess ||
      hipStreamSynchronize(ctx->stream) != hipSuccess) {
    (void)hipFree(t->d_views);
    return cfear_set_error(ctx, CFEAR_ERR_HIP, "scan table upload failed");
  }
  *out = t.release();
  return CFEAR_OK;
}

extern "C" int cfear_scan_table_size(const cfear_scan_table* t) { return t ? (int)t->n_cells.size() : CFEAR_ERR_INVALID_ARGUMENT; }

extern "C" int cfear_scan_table_destroy(cfear_scan_table* t) {
  if (!t) return CFEAR_OK;
  (void)hipSetDevice(t->ctx->device);
  (void)hipStreamSynchronize(t->ctx->stream);
  if (t->d_views) (void)hipFree(t->d_views);
  delete t;
  return CFEAR_OK;
}

extern "C" int cfear_register_candidates(cfear_ctx* ctx, const cfear_scan_table* table, const cfear_candidate* cands, int32_t n,
                                         const cfear_reg_params* par, cfear_reg_result* results) {
  if (!ctx) return CFEAR_ERR_INVALID_ARGUMENT;
  if (!table || !results || n < 0 || (n > 0 && !cands)) return cfear_set_error(ctx, CFEAR_ERR_INVALID_ARGUMENT, "null argument");
  if (table->ctx != ctx) return cfear_set_error(ctx, CFEAR_ERR_INVALID_ARGUMENT, "table belongs to another context");
  int rc = check_params(ctx, par);
  if (rc != CFEAR_OK || n == 0) return rc;
  CFEAR_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  const int nt = (int)table->n_cells.size();
  const size_t stride = reg_job_stride(2);
  const size_t cb = (size_t)n * sizeof(cfear_candidate), jb = (size_t)n * stride, rb = (size_t)n * sizeof(cfear_reg_result);
  cfear_candidate* hc = (cfear_candidate*)cfear_pinned(ctx, cb);
  if (!hc) return cfear_set_error(ctx, CFEAR_ERR_HIP, "pinned staging allocation failed");
  JobSizes sz;
  sz.cost = par->cost;
  for (int i = 0; i < n; i++) {                             // the launch geometry: the same figures gather_job derives per job
    const cfear_candidate& c = cands[i];
    if (c.target < 0 || c.target >= nt || c.source < 0 || c.source >= nt)
      return cfear_set_error(ctx, CFEAR_ERR_INVALID_ARGUMENT, "candidate %d refers to scan %d / %d of a table of %d", i, c.target, c.source, nt);
    const int n_tar = table->n_cells[(size_t)c.target], n_src = table->n_cells[(size_t)c.source];
    sz.lds_targets = std::max(sz.lds_targets, n_tar);
    sz.slots_cap = std::max(sz.slots_cap, std::max(n_src, 1));
    sz.fused_need = std::max(sz.fused_need, cfear_reg_fused_lds_need(2, n_tar, n_src, sz.cost));
    sz.fused_core = std::max(sz.fused_core, reg_fused_lds_core(2, n_tar, n_src, sz.cost));
    hc[i] = c;
  }
  const bool compact = sz.fused_need <= kRegLdsBudgetCompact && n >= 512;
  const size_t c_off = (jb + 255) / 256 * 256, r_off = c_off + (cb + 255) / 256 * 256;
  char* ws = (char*)cfear_workspace(ctx, 6, r_off + rb + 512);
  char* scr = (char*)cfear_workspace(ctx, 7, reg_scratch_bytes(sz.slots_cap) * (size_t)n);
  if (!ws || !scr) return cfear_set_error(ctx, CFEAR_ERR_HIP, "workspace allocation failed");
  const bool dev_out = cfear_is_device_ptr(results);
  cfear_reg_result* d_res = dev_out ? results : (cfear_reg_result*)(ws + r_off);
  CFEAR_HIP_CHECK(ctx, hipMemcpyAsync(ws + c_off, hc, cb, hipMemcpyHostToDevice, ctx->stream));
  if (dev_out) cfear_pinned_mark(ctx);                      // (no synchronisation below: the staging buffer stays in use)
  hipLaunchKernelGGL(expand_candidates_kernel, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, (const ScanView*)table->d_views,
                     (const cfear_candidate*)(ws + c_off), n, ws, stride);
  CFEAR_HIP_CHECK(ctx, hipGetLastError());
  rc = cfear_register_launch(ctx, ws, n, par, sz.slots_cap, sz.lds_targets, scr, d_res, nullptr, stride, compact, sz.fused_core > kRegLdsBudget);
  if (rc != CFEAR_OK) { (void)hipStreamSynchronize(ctx->stream); return rc; }
  if (dev_out) return CFEAR_OK;
  CFEAR_HIP_CHECK(ctx, hipMemcpyAsync(results, d_res, rb, hipMemcpyDeviceToHost, ctx->stream));
  CFEAR_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  return CFEAR_OK;
}

extern "C" int cfear_register(cfear_ctx* ctx, const cfear_scan* const* scans, int32_t n_scans, double* poses_xyt,
                              const cfear_reg_params* par, cfear_reg_result* result) {
  if (!ctx) return CFEAR_ERR_INVALID_ARGUMENT;
  if (!scans || !poses_xyt || !result) return cfear_set_error(ctx, CFEAR_ERR_INVALID_ARGUMENT, "null argument");
  cfear_reg_job job;
  job.scans = scans; job.n_scans = n_scans; job.pad = 0; job.poses_xyt = poses_xyt;
  int rc = cfear_register_batch(ctx, &job, 1, par, result);
  if (rc != CFEAR_OK) return rc;
  // Tsrc.back() = vectorToAffine3d(parameters.back()) whenever a solve was usable (n_scan_normal.cpp:117-119)
  poses_xyt[3 * (n_scans - 1)] = result->pose[0];
  poses_xyt[3 * (n_scans - 1) + 1] = result->pose[1];
  poses_xyt[3 * (n_scans - 1) + 2] = result->pose[2];
  return result->status;
}

// ---- GetCost for batches and covariance by cost sampling ---------------------------------------------
namespace {

// Runs the cost-only kernel over `jobs`; out receives max(mode.n_samples, 1) records per job.
// itrs (nullable) = per-job leftover itr_; otherwise par->itr applies to every job.
int run_cost_batch(cfear_ctx* ctx, const cfear_reg_job* jobs, int n_jobs, const cfear_reg_params* par,
                   const int32_t* itrs, RegCostMode mode, std::vector<cfear_reg_result>& out) {
  int rc = check_params(ctx, par);
  if (rc != CFEAR_OK) return rc;
  const int m = mode.n_samples > 0 ? mode.n_samples : 1;
  out.assign((size_t)n_jobs * m, cfear_reg_result{});
  if (n_jobs == 0) return CFEAR_OK;
  CFEAR_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  int max_scans = 2;
  for (int j = 0; j < n_jobs; j++) max_scans = std::max(max_scans, std::min(jobs[j].n_scans, kMaxScans));
  const size_t stride = reg_job_stride(max_scans);
  const size_t jb = (size_t)n_jobs * stride;
  unsigned char* hjobs = (unsigned char*)cfear_pinned(ctx, jb);          // pinned: see cfear_register_batch
  if (!hjobs) return cfear_set_error(ctx, CFEAR_ERR_HIP, "pinned staging allocation failed");
  JobSizes sz;
  sz.cost = par->cost;
  for (int j = 0; j < n_jobs; j++) {
    unsigned char* dst = hjobs + (size_t)j * stride;
    rc = gather_job(ctx, jobs[j].scans, jobs[j].n_scans, jobs[j].poses_xyt, dst, sz);
    if (rc != CFEAR_OK) return rc;
    if (itrs) cfear_reg_job_set_itr(dst, itrs[j]);
  }
  const bool compact = sz.fused_need <= kRegLdsBudgetCompact && (long long)n_jobs * m >= 512;
  // a few workgroups per job when the batch alone cannot fill the GPU; scratch bounded to 1 GiB per launch
  mode.blocks_per_job = std::max(1, std::min(m, (1024 + n_jobs - 1) / n_jobs));
  const size_t per = reg_scratch_bytes(sz.slots_cap) * (size_t)mode.blocks_per_job;
  const int chunk = (int)std::max<size_t>(1, std::min<size_t>((size_t)n_jobs, ((size_t)1 << 30) / per));
  const size_t rb = out.size() * sizeof(cfear_reg_result);
  char* ws = (char*)cfear_workspace(ctx, 6, (jb + 255) / 256 * 256 + rb + 512);
  char* scr = (char*)cfear_workspace(ctx, 7, per * (size_t)chunk);
  if (!ws || !scr) return cfear_set_error(ctx, CFEAR_ERR_HIP, "workspace allocation failed");
  char* d_jobs = ws;
  cfear_reg_result* d_res = (cfear_reg_result*)(ws + (jb + 255) / 256 * 256);
  CFEAR_HIP_CHECK(ctx, hipMemcpyAsync(d_jobs, hjobs, jb, hipMemcpyHostToDevice, ctx->stream));
  for (int j0 = 0; j0 < n_jobs; j0 += chunk) {
    const int nj = std::min(chunk, n_jobs - j0);
    rc = cfear_register_launch(ctx, d_jobs + (size_t)j0 * stride, nj, par, sz.slots_cap, sz.lds_targets, scr,
                               d_res + (size_t)j0 * m, &mode, stride, compact);
    if (rc != CFEAR_OK) { (void)hipStreamSynchronize(ctx->stream); return rc; }
  }
  CFEAR_HIP_CHECK(ctx, hipMemcpyAsync(out.data(), d_res, rb, hipMemcpyDeviceToHost, ctx->stream));
  CFEAR_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  return CFEAR_OK;
}

}  // namespace

extern "C" int cfear_get_cost_batch(cfear_ctx* ctx, const cfear_reg_job* jobs, int32_t n_jobs,
                                    const cfear_reg_params* par, cfear_reg_result* results) {
  if (!ctx) return CFEAR_ERR_INVALID_ARGUMENT;
  if (!jobs || !results || n_jobs < 0) return cfear_set_error(ctx, CFEAR_ERR_INVALID_ARGUMENT, "null argument");
  std::vector<cfear_reg_result> out;
  const int rc = run_cost_batch(ctx, jobs, n_jobs, par, nullptr, RegCostMode{}, out);
  if (rc != CFEAR_OK) return rc;
  std::copy(out.begin(), out.end(), results);
  return CFEAR_OK;
}

extern "C" void cfear_cov_sampling_params_default(cfear_cov_sampling_params* p) {
  if (!p) return;
  p->xy_range = 0.4;                  // odometrykeyframefuser.h:107 (loopclosure.cpp:108: +-0.2)
  p->yaw_range = 0.0043625;           // :108 (loopclosure.cpp:109 uses +-0.0022)
  p->samples_per_axis = 3;            // :109
  p->pad = 0;
  p->covariance_scaler = 4.0;         // :110
}

extern "C" int cfear_covariance_by_sampling_batch(cfear_ctx* ctx, const cfear_reg_job* jobs, int32_t n_jobs,
                                                  const cfear_reg_params* par, const cfear_reg_result* regs,
                                                  const cfear_cov_sampling_params* sp, double* cov36, double* samples,
                                                  int32_t* success) {
  if (!ctx) return CFEAR_ERR_INVALID_ARGUMENT;
  if (!jobs || !regs || !sp || !cov36 || !success || n_jobs < 0)
    return cfear_set_error(ctx, CFEAR_ERR_INVALID_ARGUMENT, "null argument");
  const int n = sp->samples_per_axis;
  if (n < 1 || n > 15) return cfear_set_error(ctx, CFEAR_ERR_INVALID_ARGUMENT, "samples_per_axis must be in [1,15]");
  RegCostMode mode;
  mode.samples_per_axis = n;
  mode.n_samples = n * n * n;
  mode.xy_half = sp->xy_range * 0.5;                       // odometrykeyframefuser.cpp:276-277
  mode.yaw_half = sp->yaw_range * 0.5;
  std::vector<int32_t> itrs(n_jobs);
  for (int j = 0; j < n_jobs; j++) itrs[j] = regs[j].outer_iters;      // GetCost's radius follows the leftover itr_
  std::vector<cfear_reg_result> out;
  const int rc = run_cost_batch(ctx, jobs, n_jobs, par, itrs.data(), mode, out);
  if (rc != CFEAR_OK) return rc;
  const int m = mode.n_samples;
  CovFit fit;
  fit.prepare(n, mode.xy_half, mode.yaw_half);
  std::vector<double> costs(m);
  for (int j = 0; j < n_jobs; j++) {
    double sample_cost = 0.0;                              // :282; a failed GetCost leaves the previous value (:307)
    for (int s = 0; s < m; s++) {
      const cfear_reg_result& r = out[(size_t)j * m + s];
      if (r.status == CFEAR_OK) sample_cost = r.final_cost;
      costs[s] = sample_cost;
      if (samples) {
        double* o = samples + ((size_t)j * m + s) * 4;
        o[0] = fit.offsets[3 * (size_t)s]; o[1] = fit.offsets[3 * (size_t)s + 1]; o[2] = fit.offsets[3 * (size_t)s + 2];
        o[3] = sample_cost;
      }
    }
    // GetCovarianceScaler (n_scan_normal.cpp:433-439): final_cost / (num_residuals_reduced - num_parameters_reduced)
    bool ok = regs[j].num_residuals - 3 != 0;
    if (ok) {
      const double score_scale = regs[j].final_cost / (double)(regs[j].num_residuals - 3);
      ok = fit.solve(costs.data(), score_scale, sp->covariance_scaler, cov36 + (size_t)j * 36);
    }
    success[j] = ok ? 1 : 0;
    if (!ok) {                                             // caller keeps Register's reg_cov (n_scan_normal.cpp:171-175)
      double* c = cov36 + (size_t)j * 36;
      for (int k = 0; k < 36; k++) c[k] = 0.0;
      c[0] = 0.01; c[7] = 0.01; c[35] = 1e-4;
    }
  }
  return CFEAR_OK;
}

extern "C" int cfear_covariance_by_sampling(cfear_ctx* ctx, const cfear_scan* const* scans, int32_t n_scans,
                                            const double* poses_xyt, const cfear_reg_params* par,
                                            const cfear_reg_result* reg, const cfear_cov_sampling_params* sp,
                                            double* cov36, double* samples, int32_t* success) {
  if (!ctx) return CFEAR_ERR_INVALID_ARGUMENT;
  if (!scans || !poses_xyt) return cfear_set_error(ctx, CFEAR_ERR_INVALID_ARGUMENT, "null argument");
  cfear_reg_job job;
  job.scans = scans; job.n_scans = n_scans; job.pad = 0; job.poses_xyt = poses_xyt;
  return cfear_covariance_by_sampling_batch(ctx, &job, 1, par, reg, sp, cov36, samples, success);
}

// ---- cfear_cost: one association set kept on the device -------------------------------------------
struct cfear_cost {
  cfear_ctx* ctx;
  cfear_reg_params par;
  void* d_job = nullptr;       // RegJob
  char* d_scratch = nullptr;   // slots
  double* d_out = nullptr;     // raw_r | raw_j | rob_r | neq
  int slots_cap = 0, lds_targets = 0, n_src = 0, n_slots = 0, n_blocks = 0, n_scans = 0;
  std::vector<double> h_w;     // slot weights (host copy), < 0 = no association
  std::vector<int32_t> h_tidx; // matched target cell per slot
};

extern "C" int cfear_cost_prepare(cfear_ctx* ctx, const cfear_scan* const* scans, int32_t n_scans,
                                  const double* poses_xyt, const cfear_reg_params* par, int32_t itr, cfear_cost** out) {
  if (!ctx) return CFEAR_ERR_INVALID_ARGUMENT;
  if (!scans || !poses_xyt || !out) return cfear_set_error(ctx, CFEAR_ERR_INVALID_ARGUMENT, "null argument");
  *out = nullptr;
  int rc = check_params(ctx, par);
  if (rc != CFEAR_OK) return rc;
  CFEAR_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  unsigned char hjob[sizeof(RegJob)];
  JobSizes sz;
  rc = gather_job(ctx, scans, n_scans, poses_xyt, hjob, sz);
  if (rc != CFEAR_OK) return rc;
  if (sz.lds_targets > kMaxTargetsLds) return cfear_set_error(ctx, CFEAR_ERR_CAPACITY, "target scan too large");
  cfear_cost* c = new cfear_cost();
  c->ctx = ctx; c->par = *par; c->slots_cap = sz.slots_cap; c->lds_targets = sz.lds_targets; c->n_scans = n_scans;
  c->n_src = cfear_scan_size(scans[n_scans - 1]);
  c->n_slots = (n_scans - 1) * c->n_src;
  auto fail = [&](int status, const char* msg) { cfear_cost_destroy(c); return cfear_set_error(ctx, status, "%s", msg); };
  if (hipMalloc(&c->d_job, sizeof(RegJob) + 256) != hipSuccess) return fail(CFEAR_ERR_HIP, "hipMalloc failed");
  if (hipMalloc((void**)&c->d_scratch, reg_scratch_bytes(c->slots_cap)) != hipSuccess) return fail(CFEAR_ERR_HIP, "hipMalloc failed");
  if (hipMalloc((void**)&c->d_out, ((size_t)c->slots_cap * 10 + 16) * sizeof(double)) != hipSuccess) return fail(CFEAR_ERR_HIP, "hipMalloc failed");
  if (hipMemcpyAsync(c->d_job, hjob, sizeof(RegJob), hipMemcpyHostToDevice, ctx->stream) != hipSuccess) return fail(CFEAR_ERR_HIP, "memcpy failed");
  RegCommon cm{};
  cm.par = *par; cm.angle_outlier = std::cos(M_PI / 6.0);
  cm.scratch = c->d_scratch; cm.scratch_stride = reg_scratch_bytes(c->slots_cap);
  cm.job_stride = sizeof(RegJob);
  cm.slots_cap = c->slots_cap; cm.lds_targets = (c->lds_targets + 3) & ~3; cm.results = nullptr;
  cm.dense_cap_lds = 0; cm.dense_fields = reg_dense_fields(par->cost);
  int32_t* d_nb = (int32_t*)((char*)c->d_job + sizeof(RegJob));
  // per launch: the attribute is per device, and contexts on other threads / devices share this code
  (void)hipFuncSetAttribute((const void*)assoc_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipLaunchKernelGGL(assoc_kernel, dim3(1), dim3(kRegThreads), reg_lds_bytes(cm.lds_targets, 0, 0), ctx->stream,
                     (const RegJob*)c->d_job, cm, (int)itr, d_nb);
  if (hipGetLastError() != hipSuccess) return fail(CFEAR_ERR_HIP, "assoc_kernel launch failed");
  c->h_w.assign(std::max(c->n_slots, 1), -1.0);
  c->h_tidx.assign(std::max(c->n_slots, 1), -1);
  int32_t nb = 0;
  bool ok = hipMemcpyAsync(&nb, d_nb, 4, hipMemcpyDeviceToHost, ctx->stream) == hipSuccess;
  if (c->n_slots > 0) {
    ok = ok && hipMemcpyAsync(c->h_w.data(), (double*)c->d_scratch + 5 * (size_t)c->slots_cap, (size_t)c->n_slots * 8,
                              hipMemcpyDeviceToHost, ctx->stream) == hipSuccess;
    ok = ok && hipMemcpyAsync(c->h_tidx.data(), (double*)c->d_scratch + 6 * (size_t)c->slots_cap, (size_t)c->n_slots * 4,
                              hipMemcpyDeviceToHost, ctx->stream) == hipSuccess;
  }
  ok = ok && hipStreamSynchronize(ctx->stream) == hipSuccess;
  if (!ok) return fail(CFEAR_ERR_HIP, "read-back failed");
  if (nb < 0) return fail(CFEAR_ERR_CAPACITY, "association capacity exceeded");
  c->n_blocks = nb;
  *out = c;
  return CFEAR_OK;
}

extern "C" int cfear_cost_num_blocks(const cfear_cost* c) { return c ? c->n_blocks : CFEAR_ERR_INVALID_ARGUMENT; }
extern "C" int cfear_cost_num_residuals(const cfear_cost* c) {
  return c ? c->n_blocks * (c->par.cost == CFEAR_P2L ? 1 : 2) : CFEAR_ERR_INVALID_ARGUMENT;
}

namespace {
// blocks are ordered (target scan i, source cell s) exactly like AddScanPairCost builds them
int run_eval(cfear_cost* c, const double x[3], bool want_raw) {
  cfear_ctx* ctx = c->ctx;
  RegCommon cm{};
  cm.par = c->par; cm.angle_outlier = 0; cm.scratch = c->d_scratch; cm.scratch_stride = 0; cm.job_stride = sizeof(RegJob);
  cm.slots_cap = c->slots_cap; cm.lds_targets = c->lds_targets; cm.results = nullptr;
  cm.dense_cap_lds = 0; cm.dense_fields = 0;
  EvalOut o;
  const size_t sc = (size_t)c->slots_cap;
  o.raw_r = want_raw ? c->d_out : nullptr;
  o.raw_j = want_raw ? c->d_out + 2 * sc : nullptr;
  o.rob_r = c->d_out + 8 * sc;
  o.neq = c->d_out + 10 * sc;
  hipLaunchKernelGGL(eval_kernel, dim3(1), dim3(kRegThreads), kRegFixedLds, ctx->stream, (const RegJob*)c->d_job, cm, x[0], x[1],
                     x[2], o);
  CFEAR_HIP_CHECK(ctx, hipGetLastError());
  return CFEAR_OK;
}
}  // namespace

extern "C" int cfear_cost_get_blocks(const cfear_cost* c, int32_t* pairs, double* weights) {
  if (!c) return CFEAR_ERR_INVALID_ARGUMENT;
  int b = 0;
  for (int slot = 0; slot < c->n_slots; slot++) {
    if (c->h_w[slot] < 0.0) continue;
    if (pairs) { pairs[3 * b] = slot / c->n_src; pairs[3 * b + 1] = c->h_tidx[slot]; pairs[3 * b + 2] = slot % c->n_src; }
    if (weights) weights[b] = c->h_w[slot];
    b++;
  }
  return b;
}

extern "C" int cfear_cost_evaluate(cfear_cost* c, const double x[3], double* residuals, double* jacobian) {
  if (!c || !x) return CFEAR_ERR_INVALID_ARGUMENT;
  cfear_ctx* ctx = c->ctx;
  CFEAR_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  int rc = run_eval(c, x, true);
  if (rc != CFEAR_OK) return rc;
  const size_t sc = (size_t)c->slots_cap;
  std::vector<double> r(2 * (size_t)std::max(c->n_slots, 1)), j(6 * (size_t)std::max(c->n_slots, 1));
  if (c->n_slots > 0) {
    CFEAR_HIP_CHECK(ctx, hipMemcpyAsync(r.data(), c->d_out, (size_t)c->n_slots * 16, hipMemcpyDeviceToHost, ctx->stream));
    CFEAR_HIP_CHECK(ctx, hipMemcpyAsync(j.data(), c->d_out + 2 * sc, (size_t)c->n_slots * 48, hipMemcpyDeviceToHost, ctx->stream));
  }
  CFEAR_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  const int rpb = c->par.cost == CFEAR_P2L ? 1 : 2;
  int b = 0;
  for (int slot = 0; slot < c->n_slots; slot++) {
    if (c->h_w[slot] < 0.0) continue;
    for (int i = 0; i < rpb; i++) {
      if (residuals) residuals[b * rpb + i] = r[slot * 2 + i];
      if (jacobian) for (int k = 0; k < 3; k++) jacobian[(b * rpb + i) * 3 + k] = j[slot * 6 + i * 3 + k];
    }
    b++;
  }
  return CFEAR_OK;
}

extern "C" int cfear_cost_normal_eq(cfear_cost* c, const double x[3], double H[9], double g[3], double* cost) {
  if (!c || !x) return CFEAR_ERR_INVALID_ARGUMENT;
  cfear_ctx* ctx = c->ctx;
  CFEAR_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  int rc = run_eval(c, x, false);
  if (rc != CFEAR_OK) return rc;
  double neq[10];
  CFEAR_HIP_CHECK(ctx, hipMemcpyAsync(neq, c->d_out + 10 * (size_t)c->slots_cap, sizeof(neq), hipMemcpyDeviceToHost, ctx->stream));
  CFEAR_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  if (cost) *cost = neq[0];
  if (g) { g[0] = neq[1]; g[1] = neq[2]; g[2] = neq[3]; }
  if (H) {
    H[0] = neq[4]; H[1] = neq[5]; H[2] = neq[6];
    H[3] = neq[5]; H[4] = neq[7]; H[5] = neq[8];
    H[6] = neq[6]; H[7] = neq[8]; H[8] = neq[9];
  }
  return CFEAR_OK;
}

extern "C" int cfear_cost_destroy(cfear_cost* c) {
  if (!c) return CFEAR_OK;
  (void)hipSetDevice(c->ctx->device);
  (void)hipStreamSynchronize(c->ctx->stream);
  if (c->d_job) (void)hipFree(c->d_job);
  if (c->d_scratch) (void)hipFree(c->d_scratch);
  if (c->d_out) (void)hipFree(c->d_out);
  delete c;
  return CFEAR_OK;
}

extern "C" int cfear_get_cost(cfear_ctx* ctx, const cfear_scan* const* scans, int32_t n_scans, const double* poses_xyt,
                              const cfear_reg_params* par, double* cost, double* residuals, int32_t cap,
                              int32_t* n_residuals, double* score) {
  if (!ctx) return CFEAR_ERR_INVALID_ARGUMENT;
  if (!cost || !n_residuals || !score) return cfear_set_error(ctx, CFEAR_ERR_INVALID_ARGUMENT, "null argument");
  *cost = 0.0; *n_residuals = 0; *score = 0.0;
  cfear_cost* c = nullptr;
  int rc = cfear_cost_prepare(ctx, scans, n_scans, poses_xyt, par, par ? par->itr : 0, &c);   // radius by itr_ (:220)
  if (rc != CFEAR_OK) return rc;
  const int nres = cfear_cost_num_residuals(c);
  if (nres <= 1) { cfear_cost_destroy(c); return CFEAR_ERR_TOO_FEW_RESIDUALS; }           // :200-203
  const double* x = poses_xyt + 3 * (n_scans - 1);
  rc = run_eval(c, x, false);
  if (rc != CFEAR_OK) { cfear_cost_destroy(c); return rc; }
  const size_t sc = (size_t)c->slots_cap;
  std::vector<double> r(2 * (size_t)c->n_slots);
  double neq[10];
  hipError_t e1 = hipMemcpyAsync(r.data(), c->d_out + 8 * sc, (size_t)c->n_slots * 16, hipMemcpyDeviceToHost, ctx->stream);
  hipError_t e2 = hipMemcpyAsync(neq, c->d_out + 10 * sc, sizeof(neq), hipMemcpyDeviceToHost, ctx->stream);
  hipError_t e3 = hipStreamSynchronize(ctx->stream);
  if (e1 != hipSuccess || e2 != hipSuccess || e3 != hipSuccess) { cfear_cost_destroy(c); return cfear_set_error(ctx, CFEAR_ERR_HIP, "read-back failed"); }
  const int rpb = par->cost == CFEAR_P2L ? 1 : 2;
  int b = 0;
  for (int slot = 0; slot < c->n_slots; slot++) {
    if (c->h_w[slot] < 0.0) continue;
    for (int i = 0; i < rpb; i++)
      if (residuals && b * rpb + i < cap) residuals[b * rpb + i] = r[slot * 2 + i];
    b++;
  }
  *cost = neq[0];
  *n_residuals = nres;
  *score = neq[0] / (double)std::max(nres, 1);                                            // :209
  cfear_cost_destroy(c);
  return CFEAR_OK;
}
