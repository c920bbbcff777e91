// matcher.hip -- stage M of the CFEAR hot path on gfx950: the many-to-one scan matcher, ONE kernel for every caller.
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
// matcher_kernel: one persistent workgroup per registration, one launch per batch; nothing returns to the host between the
// first association and the final pose.
//   * The keyframes' search grids come PREBUILT from the scans (ScanView::grid, built once per scan: what the reference keeps
//     as a kd-tree per MapPointNormal); staging an association is a copy of 10 bytes per target cell from L2 into LDS.
//   * Every (keyframe, source cell) pair finds its exact nearest neighbour among the grid cells its radius square overlaps
//     (FLANN's L2_Simple float arithmetic; "nearest, lowest index on ties" is ONE unsigned 64-bit minimum).
//   * When the LDS of a launch holds the tables AND the correspondence arrays, the tables are staged once; otherwise the
//     two ALIAS (the tables are copied in again for every outer iteration, keyframes in as many groups as the region needs,
//     correspondences beyond the LDS arrays in the job's global scratch).  Which one applies is decided per registration
//     from its sizes and the launch's LDS: the regular form (4 wavefronts, 40 KB: FOUR registrations per CU) aliases, the
//     forms for small batches (one or two workgroups per CU) usually do not.
//   * The trust-region state of ceres::Solve lives in LDS, not in registers: the evaluation's registers and the
//     bookkeeping's never coexist (<= 128 VGPRs: four wavefronts per SIMD).
//   * Cost-only mode (GetCost, covariance by cost sampling) is the same kernel without the solve.
// Forms (template NW = wavefronts per registration; LDS per workgroup is a launch parameter): 2 (two-scan loop-closure
// candidates, eight per CU), 4 (regular), 8 and 16 (what the regular form defers: dense scans; and batches too small to
// fill the chip, whose latency counts).
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <memory>

#include "common.hpp"
#include "fastmath.hpp"

namespace {

#ifdef CFEAR_REG_TIMING   // debug build only: cycle split of workgroup 0, printed at kernel end
// Two levels of marks.  The KERNEL's own timer runs without a gap from its first instruction to the result record: buckets 8,
// 9 (sizes + carve, staging), 10 (every association pass, whole), 11 (every solve, whole), 12 (the outer loop's break tests), 13
// (the tail) add up to the kernel's total by construction.  Inside mt_associate / eval_all / lm_solve, timers of their own split
// those two big buckets (0-3; 4, 5, 16, 21, 22); what a function's inner buckets do not cover shows as the difference to its
// whole -- round 5 printed the inner buckets only and 45 % of the cycles had no name.
__device__ long long g_reg_t[32];
__device__ unsigned long long g_reg_sum[4];   // every workgroup of a launch: sum of cycles, count, longest (printed by the NEXT launch's workgroup 0)
#define REG_T0() long long _t0 = __builtin_readcyclecounter()
#define REG_TACC(k) do { const long long _t1 = __builtin_readcyclecounter(); \
    if (threadIdx.x == 0 && blockIdx.x == 0) g_reg_t[k] += _t1 - _t0; _t0 = _t1; } while (0)
#else
#define REG_T0()
#define REG_TACC(k)
#endif

constexpr int kMaxScans = kRegMaxScans;
constexpr int kRegDeferred = 1000;            // internal status between the launches of a batch with large registrations
// (struct RegJob and reg_job_stride: common.hpp -- verify.hip's kernels write job records too)

struct MatchCommon {
  cfear_reg_params par;
  double angle_outlier;                       // std::cos(M_PI/6.0), computed on the host
  char* scratch;                              // per workgroup: the global match table, then the tail of the correspondence arrays
  size_t scratch_stride;
  size_t job_stride;                          // bytes between job records (reg_job_stride)
  int32_t pairs_cap;                          // (n_scans - 1) * n_src a workgroup's scratch holds
  int32_t dense_fields;                       // doubles per correspondence (3 P2P, 5 P2L, 6 P2D) + one u16
  uint32_t lds_total;                         // dynamic LDS bytes of the launch
  // cost-only launches (GetCost / cost sampling): no solve; n_samples > 0 evaluates the samples_per_axis^3
  // pose grid of approximateCovarianceBySampling around each job's source pose
  int32_t cost_only;
  int32_t n_samples;
  int32_t samples_per_axis;
  int32_t only_deferred;                      // launched behind another form: only the registrations it marked kRegDeferred
  int32_t take_all;                           // run every registration that can run at all (otherwise the awkward ones are deferred)
  int32_t final_launch;                       // nothing is launched behind this one: what cannot run gets CFEAR_ERR_CAPACITY
  uint32_t lds_regular;                       // LDS of the regular form: a registration it is not good at is reported (reserved = 1)
  cfear_reg_result* results;
  double xy_half, yaw_half;
  const cfear_reg_result* prior;              // cost-only: source pose and itr_ come from these records (device)
};

int reg_dense_fields(int cost) { return cost == CFEAR_P2P ? 3 : (cost == CFEAR_P2L ? 5 : 6); }

// Global scratch of one workgroup: the u16 match table of the large forms, then the tail of the dense arrays.
__host__ __device__ inline size_t match_bytes_global(int pairs_cap) { return ((size_t)pairs_cap * 2 + 255) / 256 * 256; }
size_t reg_scratch_bytes(int pairs_cap) { return match_bytes_global(pairs_cap) + ((size_t)pairs_cap * 52 + 255) / 256 * 256; }

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
        const double q = rsqrt_newton(s);                  // (s > b > 0, finite)
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

// ---------------------------------------------------------------------------------------------------
// LDS map of a registration workgroup.  Fixed block: the reduction's partials, the scan's int partials, the solver state.
// Behind it (mt_carve): keyframe transforms and pointers, the source means, the match table, then the REGION that holds the
// keyframes' tables and the dense correspondence arrays -- side by side when both fit, aliased when they do not.
// ---------------------------------------------------------------------------------------------------
// LDS-resident solver state (doubles; the int fields share the last slots).  Nothing of the trust-region bookkeeping is held
// in registers across an evaluation: wavefront 0 loads the block, walks one round of the chain and stores it back.
enum {
  S_X = 0, S_XCOST = 3, S_CUR = 4 /* g[3], H[6] */, S_SCALE = 13, S_DIAG = 16, S_XNORM = 19, S_GMAX = 20, S_RADIUS = 21,
  S_DEC = 22, S_MINCOST = 23, S_MODEL = 24, S_CAND = 25, S_COS = 28, S_SIN = 29, S_ITCOST = 30, S_ITREL = 31, S_INIT = 32,
  S_LASTREL = 33, S_FINAL = 34, S_INTS = 35 /* 8 ints */, S_OUTER = 39 /* x[3] prev_par[3] prev_score */, S_COUNT = 48
};
enum { SI_ITER = 0, SI_REUSE = 1, SI_INVALID = 2, SI_PUSHED = 3, SI_USABLE = 4, SI_DONE = 5, SI_ITSUCC = 6, SI_FLAG = 7 };
constexpr int kMaxNW = 16;
constexpr size_t kPartOff = 0;                               // [10][16] doubles: the block reduction's partials (one buffer: two
constexpr size_t kIpartOff = kMaxNW * 10 * 8;                //   barriers separate consecutive uses); [2][16] ints
constexpr size_t kStateOff = kIpartOff + 2 * kMaxNW * 4;
constexpr size_t kFixedLds = kStateOff + S_COUNT * 8;        // 1792 bytes
constexpr int kPtrs = 6;                                     // per keyframe: mean, normal, nsamples, scale, cov, grid block

// What a registration of these sizes gets from `lds_total` bytes: shared by the kernel (mt_carve) and the host (which form
// and how much LDS a batch wants).  sum_pad / max_pad: the keyframes' padded record counts (scan_grid_pad).
struct MtFit {
  size_t off_smean, off_match, off_region;
  int region;              // bytes of the region
  int tables_all;          // bytes of every keyframe's tables together
  int dense_cap;           // correspondences the LDS arrays hold
  bool gmatch;             // the match table lives in global scratch
  bool resident;           // tables and dense arrays side by side: staged once per registration
  bool can;                // the registration can run at all in this LDS
  bool good;               // ... and is one this LDS is GOOD for: tables in at most two groups, LDS arrays for 40 % of the pairs
};
__host__ __device__ inline MtFit mt_fit(size_t lds_total, int last, int sum_pad, int max_pad, int n_src, int fields, bool allow_gmatch) {
  MtFit m;
  size_t off = kFixedLds + (size_t)last * (12 * 8 + kPtrs * 8 + 16) + ((((size_t)last + 1) * 4 + 15) & ~(size_t)15);
  m.off_smean = off; off += (size_t)n_src * 16;
  const size_t n_pairs = (size_t)last * n_src;
  const size_t match_bytes = ((n_pairs + 7) & ~(size_t)7) * 2;
  const size_t need_one = (size_t)kScanGridStartPad * 2 + (size_t)max_pad * 10;
  m.off_match = off;
  // the match table: in LDS, unless that leaves no room for the largest keyframe's tables (1 400-cell scans in half a CU's
  // LDS) -- then in the job's global scratch (L2; written and read once per outer iteration by the same thread)
  m.gmatch = allow_gmatch && ((off + match_bytes + 15) & ~(size_t)15) + need_one > lds_total;
  if (!m.gmatch) off += match_bytes;
  off = (off + 15) & ~(size_t)15;
  m.off_region = off;
  m.tables_all = (int)((size_t)last * kScanGridStartPad * 2 + (size_t)sum_pad * 10);
  const size_t per = (size_t)fields * 8 + 2;
  m.can = sum_pad <= 65535 && n_src < 65536 && off + need_one <= lds_total && off + 4096 <= lds_total;
  m.region = m.can ? (int)(lds_total - off) : 0;
  const size_t dense_all = ((n_pairs + 3) & ~(size_t)3) * per;
  m.resident = m.can && (size_t)m.tables_all + dense_all + 16 <= (size_t)m.region;
  if (m.resident) m.dense_cap = (int)((n_pairs + 3) & ~(size_t)3);
  else m.dense_cap = (int)((size_t)m.region / per) & ~3;
  m.good = m.can && (m.resident || (m.tables_all <= 2 * m.region && 5 * (size_t)m.dense_cap >= 2 * n_pairs));
  return m;
}

struct MtLds {
  double* kf;              // [last][12]: Ttar (l0..l3, t0, t1), Tsrctotar (l0..l3, t0, t1)
  int* koff;               // [last + 1] prefix of the keyframes' PADDED record counts
  const void** tptr;       // [last][kPtrs] global pointers
  float4* ggeo;            // [last] grid geometry (x0, y0, cells per metre, -)
  double2* smean;          // [n_src]
  unsigned short* match;   // [n_pairs] matched target (cell index inside its keyframe), 0xFFFF = none
  unsigned short* gmatch;  // ... in the job's global scratch instead (else null)
  unsigned short* tables;  // the region: per staged group [g][kScanGridStartPad] cell starts | [NP] (x, y) | [NP] cell index
  double* dense;           // the dense arrays: behind the tables (resident) or over them (aliased)
  int dense_cap;
  int region;
  bool resident;
};

__device__ __forceinline__ void mt_carve(uint8_t* smem, const MtFit& m, int last, unsigned short* gmatch_buf, MtLds& f) {
  size_t off = kFixedLds;
  f.kf = (double*)(smem + off); off += (size_t)last * 12 * 8;
  f.tptr = (const void**)(smem + off); off += (size_t)last * kPtrs * 8;
  f.ggeo = (float4*)(smem + off); off += (size_t)last * 16;
  f.koff = (int*)(smem + off);
  f.smean = (double2*)(smem + m.off_smean);
  f.match = m.gmatch ? nullptr : (unsigned short*)(smem + m.off_match);
  f.gmatch = m.gmatch ? gmatch_buf : nullptr;
  f.tables = (unsigned short*)(smem + m.off_region);
  f.resident = m.resident;
  f.dense = (double*)(smem + m.off_region + (m.resident ? (size_t)m.tables_all : 0));
  f.dense_cap = m.dense_cap;
  f.region = m.region;
}

// 10 accumulators: cost, g[3], H upper triangle (00,01,02,11,12,22).  Wavefront 0 receives the totals (wave-uniform:
// readlane), the others' v[] is unspecified.  One LDS buffer of 10 x 4 NW doubles: the callers separate two reductions by
// two barriers.
// Value-splitting butterfly inside each 16-lane DPP row: a step that pairs lanes l and P(l) lets one class of lanes keep value
// p and the other value q of a pair, so every step halves the number of live values (10 -> 5 -> 3, then two plain steps): 56
// instructions instead of 180 (and instead of the 300 of ten full 64-lane reductions, which the forms other than 4 wavefronts
// used until round 5: a third of an LM iteration of the 8-wavefront form).  bank_mask performs the class select (banks = lane
// quads): row_mirror splits on lane bit 3 (banks 0,1 | 2,3), row_half_mirror on bit 2 (banks 0,2 | 1,3).
constexpr size_t kPartBigBytes = 10 * 4 * kMaxNW * 8;        // the partials of the 8- / 16-wavefront forms: at the END of their LDS
template <int NW>
__device__ __forceinline__ void block_reduce10(double v[10], double* buf /*[10][4 NW]*/) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  constexpr int S = 4 * NW;                                  // partials per value: one per (wavefront, row)
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
  // and 4 + 5 b3 (w2); partial (k, wave, row) goes to buf[k * S + wave * 4 + row]
  const int row = lane >> 4, b3 = (lane >> 3) & 1, b2 = (lane >> 2) & 1;
  const int slot = wave * 4 + row;
  if ((lane & 3) == 0) {
    buf[(b2 + 5 * b3) * S + slot] = w0;
    buf[(2 + b2 + 5 * b3) * S + slot] = w1;
    if (b2 == 0) buf[(4 + 5 * b3) * S + slot] = w2;
  }
  __syncthreads();
  if (wave == 0) {
    // partial j = k * S + slot sits in lane j % 64 of register j / 64; aligned groups of min(S, 16) lanes are summed by a DPP
    // butterfly (fixed order), the rows of one value by wave-uniform adds; the totals are read back with readlane, so they
    // are wave-uniform (scalar branches downstream)
    constexpr int NR = (10 * S + 63) / 64;
    double r[NR];
#pragma unroll
    for (int q = 0; q < NR; q++) {
      const int j = q * 64 + lane;
      double t = j < 10 * S ? buf[j] : 0.0;
      t += dpp_f64<0xB1>(t);                   // quad_perm [1,0,3,2]
      t += dpp_f64<0x4E>(t);                   // quad_perm [2,3,0,1]
      t += dpp_f64<0x141>(t);                  // row_half_mirror (S >= 8)
      if (S >= 16) t += dpp_f64<0x140>(t);     // row_mirror
      r[q] = t;
    }
#pragma unroll
    for (int k = 0; k < 10; k++) {
      if (S <= 16) v[k] = readlane_f64(r[(k * S) / 64], (k * S) % 64);
      else if (S == 32) v[k] = readlane_f64(r[k / 2], (k % 2) * 32) + readlane_f64(r[k / 2], (k % 2) * 32 + 16);
      else v[k] = ((readlane_f64(r[k], 0) + readlane_f64(r[k], 16)) + readlane_f64(r[k], 32)) + readlane_f64(r[k], 48);
    }
  }
}

// The same for the P2P blocks, whose Gauss-Newton matrix has H01 = 0 and H11 = H00 exactly (eval_slot): eight sums -- v[5] and
// v[7] are not read; wavefront 0 receives all ten (v[5] = 0, v[7] = v[4]).  8 -> 4 -> 2 live values, then two plain steps.
template <int NW>
__device__ __forceinline__ void block_reduce8_p2p(double v[10], double* buf /*[8][4 NW]*/) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  constexpr int S = 4 * NW;
  const double a[8] = {v[0], v[1], v[2], v[3], v[4], v[6], v[8], v[9]};
  double v1[4];
#pragma unroll
  for (int j = 0; j < 4; j++)
    v1[j] = dpp_sel_f64<0x140, 0x3>(a[j + 4], a[j]) + dpp_sel_f64<0x140, 0xC>(a[j], a[j + 4]);
  double w0 = dpp_sel_f64<0x141, 0x5>(v1[1], v1[0]) + dpp_sel_f64<0x141, 0xA>(v1[0], v1[1]);
  double w1 = dpp_sel_f64<0x141, 0x5>(v1[3], v1[2]) + dpp_sel_f64<0x141, 0xA>(v1[2], v1[3]);
  w0 += dpp_f64<0x4E>(w0); w1 += dpp_f64<0x4E>(w1);
  w0 += dpp_f64<0xB1>(w0); w1 += dpp_f64<0xB1>(w1);
  // the quad with lane bits (b3, b2) holds the row sums of packed value b2 + 4 b3 (w0) and 2 + b2 + 4 b3 (w1)
  const int row = lane >> 4, b3 = (lane >> 3) & 1, b2 = (lane >> 2) & 1;
  const int slot = wave * 4 + row;
  if ((lane & 3) == 0) {
    buf[(b2 + 4 * b3) * S + slot] = w0;
    buf[(2 + b2 + 4 * b3) * S + slot] = w1;
  }
  __syncthreads();
  if (wave == 0) {
    constexpr int NR = (8 * S + 63) / 64;
    double r[NR];
#pragma unroll
    for (int q = 0; q < NR; q++) {
      const int j = q * 64 + lane;
      double t = j < 8 * S ? buf[j] : 0.0;
      t += dpp_f64<0xB1>(t);
      t += dpp_f64<0x4E>(t);
      t += dpp_f64<0x141>(t);
      if (S >= 16) t += dpp_f64<0x140>(t);
      r[q] = t;
    }
    double o[8];
#pragma unroll
    for (int k = 0; k < 8; k++) {
      if (S <= 16) o[k] = readlane_f64(r[(k * S) / 64], (k * S) % 64);
      else if (S == 32) o[k] = readlane_f64(r[k / 2], (k % 2) * 32) + readlane_f64(r[k / 2], (k % 2) * 32 + 16);
      else o[k] = ((readlane_f64(r[k], 0) + readlane_f64(r[k], 16)) + readlane_f64(r[k], 32)) + readlane_f64(r[k], 48);
    }
    v[0] = o[0]; v[1] = o[1]; v[2] = o[2]; v[3] = o[3]; v[4] = o[4]; v[5] = 0.0; v[6] = o[5]; v[7] = o[4]; v[8] = o[6]; v[9] = o[7];
  }
}

// Residual block of one correspondence at pose x (c = cos th, s = sin th): adds to acc[10].
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
    // (H11 = H00 = the sum of rho': acc[7] is not accumulated -- block_reduce8_p2p and the cost object's evaluation set it from acc[4])
    acc[4] += rho1; acc[6] = fma(-rho1, j02, acc[6]); acc[9] = fma(h02, j02, acc[9]);
    acc[8] = fma(-rho1, j12, acc[8]); acc[9] = fma(h12, j12, acc[9]);
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

// The dense correspondence arrays: entries [0, cap) in LDS (SoA, stride cap: doubles 0 tmx, 1 tmy, 2 w, 3 a0, 4 a1, 5 a2,
// then the u16 source-cell index -- the source mean is read through it instead of being copied per block), the rest -- a
// registration with more correspondences than the LDS holds -- in the job's global scratch (SoA, stride gcap).  Two typed
// pointers, two code paths per access: no generic (flat) addressing.
struct Dense {
  double* p; unsigned short* sidx; const double2* smean; int cap; int n;     // (source cells < 65 536: 26 bytes per P2P pair)
  double* gp; int* gsidx; size_t gcap;
};

// Per-pair correspondence terms in global memory, indexed by pair p = keyframe * n_src + source cell (the Ceres-compatible
// cost object, cfear_cost_*): w < 0 = no correspondence.
struct Slots {
  double *tmx, *tmy, *a0, *a1, *a2, *w;
  int32_t* tidx;        // matched target cell
};
__host__ __device__ inline size_t slots_bytes(int cap) { return ((size_t)cap * 52 + 255) / 256 * 256; }
__device__ __forceinline__ Slots slots_of(char* scratch, int cap) {
  Slots s;
  double* p = (double*)scratch;
  s.tmx = p; s.tmy = p + cap; s.a0 = p + 2 * (size_t)cap; s.a1 = p + 3 * (size_t)cap;
  s.a2 = p + 4 * (size_t)cap; s.w = p + 5 * (size_t)cap;
  s.tidx = (int32_t*)(p + 6 * (size_t)cap);
  return s;
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

// The sizes of a registration, read by EVERY wavefront for itself (lane i: scan i; at most 16 scans): two dependent memory
// round trips in all instead of two per scan in turn, and the results are wave-uniform without an exchange.
struct MtSizes {
  int n_src, sum_pad, max_pad;   // source cells; the keyframes' padded record counts: total, largest
  int koff_lane;                 // lane i < last: exclusive prefix of the padded counts (keyframe i's first record)
  float4 geo_lane;               // lane i < last: keyframe i's grid geometry
  bool grids_ok;                 // every keyframe carries valid grid tables
};
__device__ __forceinline__ MtSizes mt_sizes(const RegJob& job, const int last) {
  const int lane = threadIdx.x & 63;
  int n = 0;
  float4 g = make_float4(0.f, 0.f, 0.f, 1.f);
  if (lane <= last) n = gload<int>(job.scans[lane].n_cells);
  if (lane < last) g = gload_f4(job.scans[lane].grid_geo);
  const int np = lane < last ? scan_grid_pad(n) : 0;
  const int incl = wave_incl_scan_i32(np);
  MtSizes z;
  z.koff_lane = incl - np;
  z.geo_lane = g;
  z.sum_pad = __builtin_amdgcn_readlane(incl, 63);
  z.max_pad = __builtin_amdgcn_readlane(wave_incl_scan_max_i32(np), 63);
  z.n_src = __builtin_amdgcn_readlane(n, last);
  z.grids_ok = __ballot(lane < last && g.w != 1.f) == 0ull;
  return z;
}

// once per registration: keyframe transforms and attribute pointers, padded record prefix, grid geometry, source means
template <int NT>
__device__ void mt_stage_once(const RegJob& job, const MtLds& f, const MtSizes& z) {
  const int tid = threadIdx.x, last = job.n_scans - 1;
  if (tid < last) { f.koff[tid] = z.koff_lane; f.ggeo[tid] = z.geo_lane; }
  if (tid == last) f.koff[last] = z.sum_pad;
  if (tid >= 64 && tid < 64 + last) {
    const int i = tid - 64;
    const Aff2 T = aff_from_xyt(job.poses[i]);
    double* k = f.kf + i * 12;
    k[0] = T.l0; k[1] = T.l1; k[2] = T.l2; k[3] = T.l3; k[4] = T.t0; k[5] = T.t1;
    const ScanView& tv = job.scans[i];
    const void** tp = f.tptr + i * kPtrs;
    tp[0] = tv.mean; tp[1] = tv.normal; tp[2] = tv.nsamples; tp[3] = tv.scale; tp[4] = tv.cov; tp[5] = tv.grid;
  }
  const ScanView& src = job.scans[last];
  for (int s = tid; s < z.n_src; s += NT) f.smean[s] = gload_d2(src.mean + s);
  __syncthreads();
}

// The grid blocks of keyframes [i0, i1) from global memory into the region: their cell-start tables first (made absolute:
// + the keyframe's first record inside the group), then their (x, y) records, then their cell indices.  A scan keeps the
// three in ONE block (ScanView::grid), so a keyframe is a run of 16-byte pieces from one base address; four keyframes x
// two pieces per thread are in flight before the first one is stored (one memory round trip for the usual registration).
template <int NT>
__device__ __forceinline__ void mt_restage(const MtLds& f, int i0, int i1) {
  const int tid = threadIdx.x;
  const int k_lo = f.koff[i0], NP = f.koff[i1] - k_lo;
  constexpr int kCs = kScanGridStartPad / 8;               // pieces of one cell-start table
  constexpr int kKf = NT >= 256 ? 4 : 1, kPer = 2;   // (kPer * NT >= 256 > kCs: the loop behind it only meets records)
  const int xy0 = (i1 - i0) * kCs, ix0 = xy0 + NP / 2;      // first (x, y) piece, first index piece
  uint4* out = (uint4*)f.tables;
  auto dst_of = [&](int i, int j, int rel, int np) {
    return j < kCs ? (i - i0) * kCs + j : (j < kCs + np / 2 ? xy0 + rel / 2 + (j - kCs) : ix0 + rel / 8 + (j - kCs - np / 2));
  };
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
        const char* blob = (const char*)f.tptr[i * kPtrs + 5];
        const int rel = f.koff[i] - k_lo, np = f.koff[i + 1] - f.koff[i], pieces = kCs + np / 2 + np / 8;
        add[q] = (unsigned)rel * 0x10001u;                // u16 pairs: no carry, the sums stay below 65536
#pragma unroll
        for (int h = 0; h < kPer; h++) {
          const int j = tid + h * NT;
          if (j < pieces) {
            v[q][h] = gload<g_u32x4>(blob + (size_t)j * 16);
            dst[q][h] = dst_of(i, j, rel, np);
          }
        }
      }
    }
#pragma unroll
    for (int q = 0; q < kKf; q++)
#pragma unroll
      for (int h = 0; h < kPer; h++)
        if (dst[q][h] >= 0) {
          const unsigned a = dst[q][h] < xy0 ? add[q] : 0u;
          out[dst[q][h]] = make_uint4(v[q][h].x + a, v[q][h].y + a, v[q][h].z + a, v[q][h].w + a);
        }
    for (int i = ib; i < min(ib + kKf, i1); i++) {        // keyframes beyond kPer x NT pieces (more than ~600 cells): the rest
      const char* blob = (const char*)f.tptr[i * kPtrs + 5];
      const int rel = f.koff[i] - k_lo, np = f.koff[i + 1] - f.koff[i], pieces = kCs + np / 2 + np / 8;
      for (int j = tid + kPer * NT; j < pieces; j += NT) {
        const g_u32x4 t = gload<g_u32x4>(blob + (size_t)j * 16);
        out[dst_of(i, j, rel, np)] = make_uint4(t.x, t.y, t.z, t.w);
      }
    }
  }
  __syncthreads();
}

// One association pass (n_scan_normal.cpp:213-318) at the pose xsrc.  SLOTS = false: fills the dense arrays `dn` (which, when
// the region is aliased, overwrite the staged tables) and returns the number of blocks.  SLOTS = true (the cost object):
// writes every pair's terms to the slot arrays `sl` instead (no compaction) and returns this thread's accepted pairs.
// `staged`: the tables of every keyframe are already in the region (resident layout, an earlier pass of this registration).
// COST: the cost function when the caller is compiled for one (the gather then carries no per-pair switches, no covariance
// register and an unrolled store of its fields), -1: read from cm.par (the cost object).
template <int NT, bool SLOTS, int COST = -1>
__device__ int mt_associate(const RegJob& job, const MatchCommon& cm, int itr, const MtLds& f, const double* xsrc, double* gl_dense,
                            Dense& dn, int* ipart, int& iphase, bool& staged, const Slots* sl = nullptr) {
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int last = job.n_scans - 1;
  const int n_src = gload<int>(job.scans[last].n_cells);
  const int n_pairs = last * n_src;
  const double curr_radius = (itr == 1) ? 2 * cm.par.radius : cm.par.radius;    // :220
  const double r2 = curr_radius * curr_radius;
  const float rwin = (float)curr_radius + 1e-3f;
  constexpr int G = kScanGrid;
  constexpr bool GM = NT >= 512;                           // the global match table exists in the large forms only (mt_fit)
  auto cost_is = [&](const int v) { return COST >= 0 ? COST == v : cm.par.cost == v; };
  // pair p = keyframe i, source cell s: p / n_src by a multiplication (floor(2^32 / n_src) or one less: the quotient is exact
  // or one short for p < 2^20, one correction) -- the running (i, s) pair it replaces cost a divergent loop per pair
  const unsigned magic = n_src ? 0xFFFFFFFFu / (unsigned)n_src : 0u;
  auto split = [&](const int p, int& i, int& s) {
    unsigned q = __umulhi((unsigned)p, magic);
    int r = p - (int)q * n_src;
    if (r >= n_src) { q++; r -= n_src; }
    i = (int)q; s = r;
  };
  REG_T0();
  // Tsrctotar_i = Ttar_i^-1 * Tsrc  (:222), on the last wavefront, ahead of the first group's table loads.  The
  // rotation of the source pose by the polynomial sincos the LM loop uses for its evaluation points (an ulp or two from libm,
  // the level at which device and host libm differ anyway; the chain is a tenth of libm's).
  {
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
  }
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
    if (f.resident) i1 = last;
    else
      while (i1 < last) {
        const int b = kScanGridStartPad * 2 + (f.koff[i1 + 1] - f.koff[i1]) * 10;
        if (bytes + b > f.region) break;
        bytes += b; i1++;
      }
    if (i1 == i0) return -1;                               // (block-uniform; mt_fit's `can` rules it out)
    if (i0 > 0) __syncthreads();                           // the previous group's readers are done
    if (!(f.resident && staged)) mt_restage<NT>(f, i0, i1);  // (its barrier also publishes the transforms)
    else __syncthreads();
    REG_TACC(0);
    const int NP = f.koff[i1] - f.koff[i0];
    const float2* txy = (const float2*)(f.tables + (size_t)(i1 - i0) * kScanGridStartPad);
    const unsigned short* tix = (const unsigned short*)(txy + NP);
    // this thread's pairs p = tid + NT k (the SAME pairs in every group layout and in pass 2) that fall into the group
    const int lo = i0 * n_src, hi = i1 * n_src;
    int p = lo + ((tid - lo) & (NT - 1));
    for (; p < hi; p += NT) {
      int i, s;
      split(p, i, s);
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
      auto visit = [&](const float2 c, const unsigned id) {
        const float dx = __fsub_rn(qx, c.x), dy = __fsub_rn(qy, c.y);
        const float d = __fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy));
        const unsigned long long key = ((unsigned long long)__float_as_uint(d) << 32) | id;
        bestkey = key < bestkey ? key : bestkey;
      };
      const unsigned short* cs = f.tables + (i - i0) * kScanGridStartPad;
      auto scan_run = [&](int qb, int qe) {
        for (int q = qb; q < qe; q += 2) {
          const int qn = min(q + 1, qe - 1);
          const float2 ca = txy[q], cb = txy[qn];
          const unsigned ia = tix[q], ib = tix[qn];
          visit(ca, ia); visit(cb, ib);
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
        pend_nt = gload_d2((const double2*)f.tptr[i * kPtrs + 1] + best);
      }
    }
    gate_pending();
    pend_p = -1;
    REG_TACC(1);
    i0 = i1;
  }
  staged = true;
  if (GM && f.gmatch) __threadfence_block();              // a thread reads back its OWN entries below: stores before loads
  int base = 0, total = 0;
  if (!SLOTS) {
    const int incl = wave_incl_scan_i32(accepted);
    base = incl - accepted;
    int* buf = ipart + iphase * kMaxNW;
    if (lane == 63) buf[wave] = incl;
    __syncthreads();                                       // also: every reader of the tables is done -- the dense arrays may overwrite them
    for (int wv = 0; wv < wave; wv++) base += buf[wv];
    int tt = buf[0];
#pragma unroll
    for (int wv = 1; wv < NT / 64; wv++) tt += buf[wv];
    total = __builtin_amdgcn_readfirstlane(tt);
    iphase ^= 1;
    dn.p = f.dense; dn.cap = f.dense_cap; dn.sidx = (unsigned short*)(f.dense + (size_t)cm.dense_fields * f.dense_cap); dn.smean = f.smean;
    dn.gp = gl_dense; dn.gcap = (size_t)cm.pairs_cap; dn.gsidx = (int*)(gl_dense + (size_t)cm.dense_fields * cm.pairs_cap);
    dn.n = total;
  }
  const size_t cap = SLOTS ? 0 : (size_t)dn.cap, gcap = SLOTS ? 0 : dn.gcap;
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
        const void* const* tp = f.tptr + i * kPtrs;
        g.nt = gload_d2((const double2*)tp[1] + g.best);
        g.tm = gload_d2((const double2*)tp[0] + g.best);
        g.tns = gload<int>((const int32_t*)tp[2] + g.best);
        g.tsc = gload<double>((const double*)tp[3] + g.best);
        g.ns = gload_d2(srcv.normal + s);
        g.sns = gload<int>(srcv.nsamples + s);
        g.ssc = gload<double>(srcv.scale + s);
        if (cost_is(CFEAR_P2D)) g.S = gload_d4((const double4*)tp[4] + g.best);
      }
      return g;
    };
    int c = base, i, s;
    split(tid, i, s);
    int m_next = GM ? match_at(tid + NT) : 0;             // the match two rounds ahead is in flight (global table: an L2 round trip)
    Gathered cur = gather(match_at(tid), i, s);
    for (int p = tid; p < n_pairs; p += NT) {
      split(p + NT, i, s);                                // (beyond the last pair: no match, nothing is read through it)
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
        if (cost_is(CFEAR_P2D)) {                                                 // :288-297
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
        } else if (cost_is(CFEAR_P2L)) {
          e[3] = K[0] * nt.x + K[1] * nt.y;                                       // Ttar.linear() * tar_normal
          e[4] = K[2] * nt.x + K[3] * nt.y;
        }
        if (SLOTS) {
          const int pp = p;
          sl->tmx[pp] = e[0]; sl->tmy[pp] = e[1]; sl->w[pp] = e[2]; sl->a0[pp] = e[3]; sl->a1[pp] = e[4]; sl->a2[pp] = e[5];
          sl->tidx[pp] = cur.best;
        } else {
          const int nf = COST == CFEAR_P2P ? 3 : (COST == CFEAR_P2L ? 5 : (COST == CFEAR_P2D ? 6 : cm.dense_fields));
          if (c < (int)cap) {
            dn.sidx[c] = (unsigned short)cur.s;
#pragma unroll
            for (int k = 0; k < 6; k++) if (k < nf) dn.p[k * cap + c] = e[k];
          } else {
            const size_t g = (size_t)c - cap;
            gstore<int>(dn.gsidx + g, cur.s);
#pragma unroll
            for (int k = 0; k < 6; k++) if (k < nf) gstore<double>(dn.gp + k * gcap + g, e[k]);
          }
        }
        c++;
      } else if (SLOTS) {
        sl->w[p] = -1.0; sl->tidx[p] = -1;
      }
      cur = nxt;
    }
  }
  if (SLOTS) return accepted;
  if (total > (int)cap) __threadfence_block();             // the tail went to global memory
  __syncthreads();
  REG_TACC(3);
  return total;
}

// cost, gradient and Gauss-Newton matrix of all correspondences at x; wavefront 0 receives the sums
template <int NT, int COST, int LOSS>
__device__ void eval_all(const MatchCommon& cm, const Dense& dn, const double x[3], double c, double s, double out[10], double* part) {
  double acc[10];
#pragma unroll
  for (int k = 0; k < 10; k++) acc[k] = 0.0;
  const size_t cap = (size_t)dn.cap, gcap = dn.gcap;
  REG_T0();
  // entries [0, cap) from LDS, the rest from the job's global scratch: a thread meets its LDS entries first (i grows), so two
  // loops add in the same order as one -- and the LDS loop, the hot one, carries no address-space branch.  The tail's loads
  // are independent of the arithmetic: kTailU entries per thread are requested before the first one is used.
  const int n_lds = min(dn.n, (int)cap);
  for (int i = threadIdx.x; i < n_lds; i += NT) {
    double a0 = 0.0, a1 = 0.0, a2 = 0.0;
    const int si = (int)dn.sidx[i];
    const double tmx = dn.p[i], tmy = dn.p[cap + i], w = dn.p[2 * cap + i];
    if (COST != CFEAR_P2P) { a0 = dn.p[3 * cap + i]; a1 = dn.p[4 * cap + i]; }
    if (COST == CFEAR_P2D) a2 = dn.p[5 * cap + i];
    const double2 sm = dn.smean[si];
    eval_slot<COST, LOSS, true>(cm.par, sm.x, sm.y, tmx, tmy, a0, a1, a2, w, x[0], x[1], c, s, acc);
  }
  constexpr int kTailU = COST == CFEAR_P2P ? 4 : 2;
  const int n_tail = dn.n - n_lds;
  for (int g0 = threadIdx.x; g0 < n_tail; g0 += kTailU * NT) {
    int si[kTailU];
    double tmx[kTailU], tmy[kTailU], w[kTailU], a0[kTailU], a1[kTailU], a2[kTailU];
#pragma unroll
    for (int u = 0; u < kTailU; u++) {
      const size_t g = (size_t)min(g0 + u * NT, n_tail - 1);
      si[u] = gload<int>(dn.gsidx + g);
      tmx[u] = gload<double>(dn.gp + g); tmy[u] = gload<double>(dn.gp + gcap + g); w[u] = gload<double>(dn.gp + 2 * gcap + g);
      a0[u] = a1[u] = a2[u] = 0.0;
      if (COST != CFEAR_P2P) { a0[u] = gload<double>(dn.gp + 3 * gcap + g); a1[u] = gload<double>(dn.gp + 4 * gcap + g); }
      if (COST == CFEAR_P2D) a2[u] = gload<double>(dn.gp + 5 * gcap + g);
    }
#pragma unroll
    for (int u = 0; u < kTailU; u++)
      if (g0 + u * NT < n_tail) {
        const double2 sm = dn.smean[si[u]];
        eval_slot<COST, LOSS, true>(cm.par, sm.x, sm.y, tmx[u], tmy[u], a0[u], a1[u], a2[u], w[u], x[0], x[1], c, s, acc);
      }
  }
  REG_TACC(4);
  if (COST == CFEAR_P2P) block_reduce8_p2p<NT / 64>(acc, part);
  else block_reduce10<NT / 64>(acc, part);
  REG_TACC(5);
#pragma unroll
  for (int k = 0; k < 10; k++) out[k] = acc[k];
}


// One round of ceres::Solve's trust-region loop on wavefront 0: judges the candidate that was just evaluated (cnd, when
// have_cnd), then produces the next candidate or the stop flag.  st = the LDS state block.
__device__ __forceinline__ void mt_lds_order() {          // lane 0's LDS stores above, the wavefront's loads below (one wavefront: the
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");   // LDS executes its operations in order; this orders the compiler)
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
}
__device__ __forceinline__ void lm_round(double* st, const double cnd[10], const bool have_cnd, const int max_iter) {
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
      it_rel = cost_change / model_cost_change;               // (rcp_newton instead of the IEEE quotient: +-0 on the kernel)
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
    mt_lds_order();
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
      mt_lds_order();                                      // (the diagonal stored above is read again by the next turn)
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

// ceres::Solve: the start pose is st[S_OUTER .. +3); the result is left in the state block (S_X, S_FINAL, S_LASTREL,
// SI_PUSHED, SI_USABLE).  Block-wide collective; per LM iteration: evaluate (all wavefronts) -> reduce -> round (wavefront 0).
// eval_only (GetCost): the evaluation at the start pose and nothing else; its cost is left in st[S_XCOST].
template <int NT, int COST, int LOSS>
__device__ void lm_solve(const MatchCommon& cm, const Dense& dn, const int max_iter, double* part, double* st, const bool eval_only) {
  const bool w0 = (threadIdx.x >> 6) == 0;
  int* si = (int*)(st + S_INTS);
  double cnd[10];
  {
    const double x[3] = {st[S_OUTER], st[S_OUTER + 1], st[S_OUTER + 2]};
    double s0, c0;
    sincos_pose<true>(x[2], &s0, &c0);
    eval_all<NT, COST, LOSS>(cm, dn, x, c0, s0, cnd, part);
    if (eval_only) {                                        // (block-uniform)
      if (threadIdx.x == 0) st[S_XCOST] = cnd[0];
      return;
    }
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
      mt_lds_order();                                       // lane 0's block is read by the whole wavefront below
      lm_round(st, cnd, false, max_iter);
    }
  }
  for (;;) {
    REG_T0();
    __syncthreads();
    REG_TACC(21);
    if (__builtin_amdgcn_readfirstlane(si[SI_DONE])) break;
    const double cand[3] = {st[S_CAND], st[S_CAND + 1], st[S_CAND + 2]};
    const double cs = st[S_COS], sn = st[S_SIN];
    eval_all<NT, COST, LOSS>(cm, dn, cand, cs, sn, cnd, part);        // its barrier also orders the state block
    REG_TACC(22);
    // The round is the serial piece of an LM iteration (three wavefronts wait at the barrier for it): its wavefront takes issue
    // priority over the other registrations' wavefronts on its SIMD for these ~400 instructions (register 1.154 -> 1.145 ms).
    if (w0) { __builtin_amdgcn_s_setprio(3); lm_round(st, cnd, true, max_iter); __builtin_amdgcn_s_setprio(0); }
    REG_TACC(16);
  }
}

__device__ __forceinline__ void write_capacity(cfear_reg_result* r, const double x[3]) {
  r->pose[0] = x[0]; r->pose[1] = x[1]; r->pose[2] = x[2];
  r->score = 0; r->final_cost = 0; r->num_residuals = 0; r->outer_iters = 0; r->lm_iters = 0;
  r->status = CFEAR_ERR_CAPACITY; r->last_relative_decrease = 0; r->reserved = 0;
}

// Wavefronts per SIMD the forms are compiled for: four (<= 128 VGPRs) -- except the regular form with a loss other than Huber
// (~155 VGPRs: three; the host gives those launches 52 KB per workgroup instead of 40).
// CO: the cost-only launches (GetCost / cost sampling) are instantiations of their own -- the same loop without the solve.
// WIDE (8 wavefronts, a batch of at most one workgroup per CU -- the single sequence): the whole register file, no spills.
template <int NW, int COST, int LOSS, bool CO, bool WIDE = false>
__global__ __launch_bounds__(NW * 64, WIDE ? 2 : ((NW == 4 && LOSS < 0) ? 3 : 4)) void matcher_kernel(const RegJob* __restrict__ jobs, const MatchCommon cm) {
  constexpr int NT = NW * 64;
#ifdef CFEAR_REG_TIMING
  if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0 && g_reg_sum[1]) {   // the previous launch's workgroups
    printf("matcher launch: %llu workgroups, mean %llu cycles, longest %llu\n", g_reg_sum[1], g_reg_sum[0] / g_reg_sum[1], g_reg_sum[2]);
    g_reg_sum[0] = g_reg_sum[1] = g_reg_sum[2] = 0;
  }
  // (taken BEHIND that printf: a device printf is a host call of ~0.1 ms -- round 5's "total" started before it, which is where
  //  most of its 45 % of cycles without a name went)
  const long long t_total0 = __builtin_readcyclecounter();
#endif
  REG_T0();
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  // (the large forms keep their reduction partials -- 10 x 4 NW doubles -- behind the carved LDS)
  const uint32_t lds_eff = cm.lds_total - (NW >= 8 ? (uint32_t)kPartBigBytes : 0u);
  double* part = NW >= 8 ? (double*)(smem + lds_eff) : (double*)(smem + kPartOff);
  int* ipart = (int*)(smem + kIpartOff);
  double* st = (double*)(smem + kStateOff);
  const RegJob& job = *(const RegJob*)((const char*)jobs + (size_t)blockIdx.x * cm.job_stride);
  const int m_out = CO ? (cm.n_samples > 0 ? cm.n_samples : 1) : 1;
  cfear_reg_result* res = cm.results + (size_t)blockIdx.x * m_out;
  if (cm.only_deferred && res->status != kRegDeferred) return;      // a later form: only what the launches before left
  const int last = job.n_scans - 1;
  const MtSizes sz = mt_sizes(job, last);
  const int n_src = sz.n_src, sum_pad = sz.sum_pad, max_pad = sz.max_pad;
  const int n_pairs = last * n_src;
  const size_t scr_idx = (size_t)blockIdx.x * gridDim.y + blockIdx.y;
  char* scr = cm.scratch + scr_idx * cm.scratch_stride;
  const MtFit fit = mt_fit(lds_eff, last, sum_pad, max_pad, n_src, cm.dense_fields, NW >= 8);
  MtLds fl;
  // (grids_ok false: a scan without grid tables -- more than 65 535 cells)
  const bool ok = fit.can && sz.grids_ok && n_pairs <= cm.pairs_cap && (cm.take_all || fit.good);
  REG_TACC(8);
  if (ok) {
    mt_carve(smem, fit, last, (unsigned short*)scr, fl);
    mt_stage_once<NT>(job, fl, sz);
  }
  REG_TACC(9);
  if (!ok) {
    if (cm.final_launch) {                                  // nothing behind this launch: report it -- and whether the largest
      // form (a whole CU's LDS, the match table in global scratch) would hold it: a caller that launched without the large
      // forms (RegLaunchHint::big_pass off) launches again with them (odometry.hip)
      const double larger = (mt_fit((uint32_t)(160 * 1024 - 256 - kPartBigBytes), last, sum_pad, max_pad, n_src, cm.dense_fields, true).can &&
                             sz.grids_ok && n_pairs <= cm.pairs_cap && lds_eff < 160u * 1024u - 256u - (uint32_t)kPartBigBytes) ? 1.0 : 0.0;
      for (int sidx = blockIdx.y * NT + (int)threadIdx.x; sidx < m_out; sidx += (int)gridDim.y * NT) {
        write_capacity(res + sidx, job.poses[last]);
        res[sidx].reserved = larger;
      }
    } else if (threadIdx.x == 0) { res->status = kRegDeferred; res->reserved = 0.0; }
    return;
  }
  double* gl_dense = (double*)(scr + match_bytes_global(cm.pairs_cap));
  Dense dn;
  int iphase = 0;
  bool staged = false;
  const int rpb = COST == CFEAR_P2L ? 1 : 2;
  // n_scan_normal.cpp:82-185.  The loop's own state (current pose, previous pose and score) lives in the LDS block too:
  // st[S_OUTER .. +3) = parameters.back(), +3 .. +6 = prev_par, +6 = prev_score.
  // Cost-only launches walk the SAME loop (one call site of the association, one of the evaluation): a "round" is then one
  // sample -- n_scan_normal_reg::GetCost (n_scan_normal.cpp:186-211): one association pass (radius by the leftover itr_, :220) +
  // the robust cost at the given pose; with n_samples > 0 once per pose of the sampling grid of approximateCovarianceBySampling
  // (odometrykeyframefuser.cpp:287-303: theta outer, x, y inner).
  int git = job.itr ? job.itr : cm.par.itr;
  const bool from_prior = CO && cm.prior;        // sample around the pose a Register launch just produced
  if (from_prior) git = cm.prior[blockIdx.x].outer_iters;
  git = __builtin_amdgcn_readfirstlane(git);
  if (threadIdx.x == 0) {
    const double* x0 = from_prior ? cm.prior[blockIdx.x].pose : job.poses[last];
#pragma unroll
    for (int k = 0; k < 3; k++) { st[S_OUTER + k] = x0[k]; st[S_OUTER + 3 + k] = x0[k]; }   // (cost-only: +3 .. +6 = the centre of the samples)
    st[S_OUTER + 6] = DBL_MAX;
    st[S_FINAL] = 0.0; st[S_LASTREL] = 0.0;
  }
  __syncthreads();
  bool success = true;
  int itr = 1, lm_iters = 0, num_residuals = 0, solved_residuals = 0, fail_status = CFEAR_OK;
  for (itr = 1;; itr++) {
    int a_itr = itr;
    const int sidx = (int)blockIdx.y + (itr - 1) * (int)gridDim.y;
    if (CO) {
      if (sidx >= m_out) break;
      a_itr = git;
      if (threadIdx.x == 0 && cm.n_samples > 0) {          // (the previous sample's readers passed the barrier below)
        const int n = cm.samples_per_axis;
        auto lin = [n](double half, int i) {               // linspace(-half, half, n)[i] (loopclosure.cpp:866-890)
          if (n == 1) return -half;
          const double delta = (half - (-half)) / ((double)n - 1.0);
          return i < n - 1 ? -half + delta * (double)i : half;
        };
        st[S_OUTER] = lin(cm.xy_half, (sidx / n) % n) + st[S_OUTER + 3];
        st[S_OUTER + 1] = lin(cm.xy_half, sidx % n) + st[S_OUTER + 4];
        st[S_OUTER + 2] = lin(cm.yaw_half, sidx / (n * n)) + st[S_OUTER + 5];
      }
      __syncthreads();
    } else if (itr > cm.par.max_itr_association || !success) break;
    REG_TACC(12);                                                 // (the outer loop between a solve and the next association)
    const int n_blocks = mt_associate<NT, false, COST>(job, cm, a_itr, fl, st + S_OUTER, gl_dense, dn, ipart, iphase, staged);
    REG_TACC(10);
    num_residuals = n_blocks * rpb;
    success = num_residuals > 1;                                  // :368-369
    if (!success && !CO) { fail_status = CFEAR_ERR_TOO_FEW_RESIDUALS; break; }
    if (success) { solved_residuals = num_residuals; lm_solve<NT, COST, LOSS>(cm, dn, cm.par.max_itr_solver, part, st, CO); }
    REG_TACC(11);
    if (CO) {
      if (threadIdx.x == 0) {
        cfear_reg_result* r = res + sidx;
        const double c = success ? st[S_XCOST] : 0.0;
        r->pose[0] = st[S_OUTER]; r->pose[1] = st[S_OUTER + 1]; r->pose[2] = st[S_OUTER + 2];
        r->final_cost = c;
        r->score = success ? c / (double)num_residuals : 0.0;                    // score_ (:209)
        r->num_residuals = num_residuals; r->outer_iters = git; r->lm_iters = 0;
        r->status = success ? CFEAR_OK : CFEAR_ERR_TOO_FEW_RESIDUALS;            // :200-203
        r->last_relative_decrease = 0.0; r->reserved = 0.0;
      }
      __syncthreads();                                            // the dense arrays and the pose are rewritten by the next sample
      continue;
    }
    const int* si = (const int*)(st + S_INTS);
    const int n_pushed = __builtin_amdgcn_readfirstlane(si[SI_PUSHED]);
    lm_iters += n_pushed - 1;
    success = __builtin_amdgcn_readfirstlane(si[SI_USABLE]) != 0;
    if (!success) fail_status = CFEAR_ERR_SOLVER;
    // n_scan_normal.cpp:117-149, written with selects instead of nested if / else-break (hipcc mis-merged the phi of prev_par
    // on the "continue" edge of the nested form); every thread derives the same decision, thread 0 stores it
    const double current_score = st[S_FINAL], prev_score = st[S_OUTER + 6];
    const double rel_improvement = (prev_score - current_score) / prev_score;
    const bool past_min = itr > cm.par.min_itr;
    const bool worse = past_min && (prev_score < current_score);          // recover to prev iteration
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
  if (CO) return;
  if (threadIdx.x == 0) {
    res->pose[0] = st[S_OUTER]; res->pose[1] = st[S_OUTER + 1]; res->pose[2] = st[S_OUTER + 2];
    const double final_cost = st[S_FINAL];
    res->final_cost = final_cost;
    // summary_.num_residuals: of the last problem that was SOLVED -- an association pass that ends the loop with too few
    // residuals (n_scan_normal.cpp:112-113) leaves the summary of the solve before it (0: there was none)
    res->num_residuals = solved_residuals;
    res->outer_iters = itr;
    res->lm_iters = lm_iters;
    res->last_relative_decrease = st[S_LASTREL];
    // a registration the regular form is not good at (dense scans): tells the caller to keep the large forms on
    res->reserved = mt_fit(cm.lds_regular, last, sum_pad, max_pad, n_src, cm.dense_fields, false).good ? 0.0 : 1.0;
    if (success) { res->score = final_cost / (double)solved_residuals; res->status = CFEAR_OK; }   // :162
    else { res->score = 0.0; res->status = fail_status; }
#ifdef CFEAR_REG_TIMING
    {
      const unsigned long long cyc = (unsigned long long)(__builtin_readcyclecounter() - t_total0);
      atomicAdd(&g_reg_sum[0], cyc); atomicAdd(&g_reg_sum[1], 1ull); atomicMax(&g_reg_sum[2], cyc);
    }
    if (blockIdx.x == 0) {
      REG_TACC(13);
      const long long top = g_reg_t[8] + g_reg_t[9] + g_reg_t[10] + g_reg_t[11] + g_reg_t[12] + g_reg_t[13];
      printf("matcher cycles: total %lld = top-level buckets %lld | sizes+carve %lld stage_once %lld association %lld solve %lld outer-loop tests %lld tail %lld\n",
             (long long)(__builtin_readcyclecounter() - t_total0), top, g_reg_t[8], g_reg_t[9], g_reg_t[10], g_reg_t[11], g_reg_t[12], g_reg_t[13]);
      printf("matcher cycles (inside): association: restage %lld nn+gate %lld scan %lld gather %lld | solve: eval %lld reduce %lld round %lld barrier %lld (an LM iteration's evaluation as a whole, eval + reduce + call: %lld) | outer %d lm %d n %d\n",
             g_reg_t[0], g_reg_t[1], g_reg_t[2], g_reg_t[3], g_reg_t[4], g_reg_t[5], g_reg_t[16], g_reg_t[21], g_reg_t[22], itr, lm_iters, num_residuals);
      for (int k = 0; k < 32; k++) g_reg_t[k] = 0;
    }
#endif
  }
}

// ---- the Ceres-compatible cost object (cfear_cost_*): association once, then evaluations at any pose -----------------
// association only: every pair's terms go to the slot arrays in the caller's scratch (no compaction: a block is a pair)
__global__ __launch_bounds__(256) void assoc_kernel(const RegJob* __restrict__ jobs, const MatchCommon cm, int itr, int32_t* n_blocks_out) {
  constexpr int NT = 256;
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  int* ipart = (int*)(smem + kIpartOff);
  const RegJob& job = jobs[0];
  const int last = job.n_scans - 1;
  const MtSizes sz = mt_sizes(job, last);
  const MtFit fit = mt_fit(cm.lds_total, last, sz.sum_pad, sz.max_pad, sz.n_src, 0, false);
  MtLds fl;
  const bool ok = fit.can && sz.grids_ok && last * sz.n_src <= cm.pairs_cap;
  if (ok) {
    mt_carve(smem, fit, last, nullptr, fl);
    fl.resident = false;                                   // (no dense arrays here: the region is the tables', group by group)
    mt_stage_once<NT>(job, fl, sz);
  }
  if (!ok) {
    if (threadIdx.x == 0) n_blocks_out[0] = -1;
    return;
  }
  const Slots sl = slots_of(cm.scratch, cm.pairs_cap);
  Dense dn;
  int iphase = 0;
  bool staged = false;
  const int mine = mt_associate<NT, true>(job, cm, itr, fl, job.poses[last], nullptr, dn, ipart, iphase, staged, &sl);
  const int incl = wave_incl_scan_i32(mine);
  __syncthreads();
  if ((threadIdx.x & 63) == 63) ipart[threadIdx.x >> 6] = incl;
  __syncthreads();
  if (threadIdx.x == 0) n_blocks_out[0] = ipart[0] + ipart[1] + ipart[2] + ipart[3];
}

// per-slot evaluation at x for one job: raw residuals/Jacobians (Ceres CostFunction::Evaluate
// semantics), robustified residuals, and the reduced normal equations.
struct EvalOut {
  double* raw_r;       // [slots][2]
  double* raw_j;       // [slots][6]
  double* rob_r;       // [slots][2]
  double* neq;         // [10]: cost, g, H upper
};
__global__ __launch_bounds__(256) void eval_kernel(const RegJob* __restrict__ jobs, const MatchCommon cm, double x0, double x1, double x2,
                                                   EvalOut o) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  double* part = (double*)(smem + kPartOff);
  const RegJob& job = jobs[0];
  const int last = job.n_scans - 1;
  const int n_src = *job.scans[last].n_cells;
  const int n_slots = last * n_src;
  const Slots sl = slots_of(cm.scratch, cm.pairs_cap);
  double s, c;
  sincos(x2, &s, &c);
  const double2* smean = job.scans[last].mean;
  double acc[10];
#pragma unroll
  for (int k = 0; k < 10; k++) acc[k] = 0.0;
  for (int slot = threadIdx.x; slot < n_slots; slot += 256) {
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
  if (cm.par.cost == CFEAR_P2P) acc[7] = acc[4];           // H11 = H00 (eval_slot)
  block_reduce10<4>(acc, part);
  if (threadIdx.x == 0 && o.neq) for (int k = 0; k < 10; k++) o.neq[k] = acc[k];
}

int check_params(cfear_ctx* ctx, const cfear_reg_params* p) {
  if (!p) return cfear_set_error(ctx, CFEAR_ERR_INVALID_ARGUMENT, "null parameters");
  if (p->cost < 0 || p->cost > 2 || p->loss < 0 || p->loss > 5 || p->weight_opt < 0 || p->weight_opt > 4)
    return cfear_set_error(ctx, CFEAR_ERR_INVALID_ARGUMENT, "bad cost/loss/weight option");
  if (p->max_itr_association < 1 || p->max_itr_solver < 0 || !(p->radius > 0))
    return cfear_set_error(ctx, CFEAR_ERR_INVALID_ARGUMENT, "bad iteration limits / radius");
  return CFEAR_OK;
}

typedef void (*MatcherFn)(const RegJob*, MatchCommon);
template <int NW, bool CO>
MatcherFn matcher_fn_nw(int cost, bool huber) {
  switch (cost) {
    case CFEAR_P2P: return huber ? matcher_kernel<NW, CFEAR_P2P, CFEAR_LOSS_HUBER, CO> : matcher_kernel<NW, CFEAR_P2P, -1, CO>;
    case CFEAR_P2L: return huber ? matcher_kernel<NW, CFEAR_P2L, CFEAR_LOSS_HUBER, CO> : matcher_kernel<NW, CFEAR_P2L, -1, CO>;
    default: return huber ? matcher_kernel<NW, CFEAR_P2D, CFEAR_LOSS_HUBER, CO> : matcher_kernel<NW, CFEAR_P2D, -1, CO>;
  }
}
MatcherFn matcher_fn_wide(int cost, bool huber) {
  switch (cost) {
    case CFEAR_P2P: return huber ? matcher_kernel<8, CFEAR_P2P, CFEAR_LOSS_HUBER, false, true> : matcher_kernel<8, CFEAR_P2P, -1, false, true>;
    case CFEAR_P2L: return huber ? matcher_kernel<8, CFEAR_P2L, CFEAR_LOSS_HUBER, false, true> : matcher_kernel<8, CFEAR_P2L, -1, false, true>;
    default: return huber ? matcher_kernel<8, CFEAR_P2D, CFEAR_LOSS_HUBER, false, true> : matcher_kernel<8, CFEAR_P2D, -1, false, true>;
  }
}
// full registrations: 2 / 4 / 8 / 16 wavefronts (8 also `wide`); cost-only launches: 4 / 8
MatcherFn matcher_fn(int nw, int cost, bool huber, bool cost_only, bool wide = false) {
  if (cost_only) return nw <= 4 ? matcher_fn_nw<4, true>(cost, huber) : matcher_fn_nw<8, true>(cost, huber);
  if (nw == 8 && wide) return matcher_fn_wide(cost, huber);
  switch (nw) {
    case 2: return matcher_fn_nw<2, false>(cost, huber);
    case 4: return matcher_fn_nw<4, false>(cost, huber);
    case 8: return matcher_fn_nw<8, false>(cost, huber);
    default: return matcher_fn_nw<16, false>(cost, huber);
  }
}

}  // namespace

size_t cfear_reg_job_bytes() { return sizeof(RegJob); }
int cfear_reg_max_scans() { return kMaxScans; }
void cfear_reg_job_set_itr(void* job, int itr) { ((RegJob*)job)->itr = itr; }
size_t cfear_reg_job_stride(int max_scans) { return reg_job_stride(max_scans); }
size_t cfear_register_scratch_bytes(int pairs_cap) { return reg_scratch_bytes(pairs_cap); }

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

namespace {
constexpr size_t kLdsCu = 160 * 1024;                        // LDS of a gfx950 compute unit
constexpr size_t kLdsPairs = 20 * 1024;                      // the 2-wavefront form for two-scan candidates: eight per CU
size_t regular_lds(bool huber) { return huber ? kLdsCu / 4 : (kLdsCu / 3) & ~(size_t)255; }   // four (three: matcher_kernel) per CU

// The first launch's form for a batch of n_wgs workgroups (measured: tools/form_sweep.py).  A batch that fills
// the chip several times over runs the regular form -- 4 wavefronts, four registrations per CU (two-scan candidates from eight
// per CU on: 2 wavefronts, 20 KB) -- for throughput; below that every registration gets as much LDS as the CU can give its
// share of the batch (tables and correspondence arrays side by side: staged once, not per outer iteration), and a batch of at
// most two workgroups per CU is a matter of latency: 8 wavefronts per registration (two-scan candidates stay at 4: ~320 pairs
// do not feed 512 lanes).
struct Form { int nw; size_t lds; };
Form first_form(const cfear_ctx* ctx, int n_wgs, bool huber, bool small_pairs, bool big_pass) {
  const int per_cu = std::max(1, (n_wgs + ctx->n_cu - 1) / ctx->n_cu);
  Form f;
  if (small_pairs && per_cu >= 8) { f.nw = 2; f.lds = kLdsPairs; }
  else if (per_cu <= 2 && !small_pairs) { f.nw = 8; f.lds = (big_pass && per_cu == 1) ? kLdsCu - 256 : kLdsCu / 2 - 256; }   // (large scans around: the CU's whole LDS at once)
  else { f.nw = 4; f.lds = per_cu <= 2 ? kLdsCu / 2 - 256 : std::max(regular_lds(huber), (kLdsCu / std::min(per_cu, huber ? 4 : 3)) & ~(size_t)255); }
  if (ctx->opt[CFEAR_OPT_MATCHER_WAVES]) f.nw = (int)ctx->opt[CFEAR_OPT_MATCHER_WAVES];        // test / measurement hooks
  if (ctx->opt[CFEAR_OPT_MATCHER_LDS_KB]) f.lds = (size_t)ctx->opt[CFEAR_OPT_MATCHER_LDS_KB] * 1024 - (ctx->opt[CFEAR_OPT_MATCHER_LDS_KB] == 160 ? 256 : 0);
  return f;
}

}  // namespace

// Enqueues the matcher over d_jobs [n_jobs] (device records, job_stride bytes apart), results to d_results [n_jobs] (device;
// cost-only: max(n_samples, 1) records per job).  pairs_cap bounds (n_scans - 1) * n_src per job (the scratch's size).
int cfear_register_launch(cfear_ctx* ctx, const void* d_jobs, int n_jobs, const cfear_reg_params* par, int pairs_cap,
                          char* d_scratch, cfear_reg_result* d_results, const RegCostMode* mode, size_t job_stride, RegLaunchHint hint) {
  MatchCommon cm{};
  cm.par = *par;
  cm.angle_outlier = std::cos(M_PI / 6.0);                                     // n_scan_normal.cpp:217
  cm.scratch = d_scratch;
  cm.scratch_stride = reg_scratch_bytes(pairs_cap);
  cm.job_stride = job_stride ? job_stride : sizeof(RegJob);
  cm.pairs_cap = pairs_cap;
  cm.dense_fields = reg_dense_fields(par->cost);
  cm.results = d_results;
  cm.cost_only = mode ? 1 : 0;
  cm.n_samples = mode ? mode->n_samples : 0;
  cm.samples_per_axis = mode ? mode->samples_per_axis : 0;
  cm.xy_half = mode ? mode->xy_half : 0.0;
  cm.yaw_half = mode ? mode->yaw_half : 0.0;
  cm.prior = mode ? mode->prior : nullptr;
  const bool huber = par->loss == CFEAR_LOSS_HUBER;             // compile-time specialisations: cost metric x {Huber, any other loss}
  cm.lds_regular = (uint32_t)regular_lds(huber);
  const int by = mode ? std::max(mode->blocks_per_job, 1) : 1;
  Form f0 = first_form(ctx, n_jobs * by, huber, hint.small_pairs, hint.big_pass);
  if (mode) {                                                // cost-only launches come with 4 or 8 wavefronts
    // A job beyond half a CU's LDS among them (the host's marshalling saw its sizes): ONE launch of the 8-wavefront form with the CU's whole LDS, i.e. the capacity of
    // the full registrations' last form (a deferred chain like theirs would race here: the workgroups that share a job's
    // samples all read the job's first record to see whether it was deferred while the first of them already overwrites it) --
    // GetCost / cost sampling / CFEAR quality of a pair that Register holds no longer come back as CFEAR_ERR_CAPACITY
    if (hint.whole_cu) { f0.nw = 8; f0.lds = kLdsCu - 256; }
    else if (hint.big_pass && f0.nw < 8) { f0.nw = 8; f0.lds = kLdsCu / 2 - 256; }   // cost sampling while dense scans show up
    if (f0.nw == 2) { f0.nw = 4; f0.lds = std::max(f0.lds, regular_lds(huber)); }
    f0.nw = f0.nw <= 4 ? 4 : 8;
  }
  // The launches: the first form; then, when the caller expects registrations it is not good at (dense scans: ~1 400 cells),
  // half a CU each (8 wavefronts, 80 KB: two workgroups per CU overlap each other's serial phases) and a whole CU (16
  // wavefronts) for what even a global match table does not fit into 80 KB.  A launch behind another one only runs what that
  // one marked deferred; everything else returns at once (6-10 us: the caller switches them off when no large scans show up).
  const bool more = !mode && hint.big_pass && !(f0.nw >= 8 && f0.lds >= kLdsCu - 256);
  Form forms[3] = {f0, Form{8, kLdsCu / 2 - 256}, Form{16, kLdsCu - 256}};
  const char* names[3] = {"register", "register_large", "register_large16"};
  int n_forms = more ? 3 : 1;
  if (more && f0.nw >= 8 && f0.lds >= forms[1].lds) { forms[1] = forms[2]; names[1] = names[2]; n_forms = 2; }   // (the first form IS the half-CU form)
  for (int k = 0; k < n_forms; k++) {
    const Form& f = forms[k];
    const MatcherFn fn = matcher_fn(f.nw, par->cost, huber, mode != nullptr, n_jobs * by <= ctx->n_cu);
    { const int rc = cfear_allow_lds(ctx, (const void*)fn, kLdsCu); if (rc != CFEAR_OK) return rc; }
    MatchCommon c = cm;
    c.lds_total = (uint32_t)f.lds;
    c.only_deferred = k > 0;
    c.final_launch = k == n_forms - 1;
    c.take_all = k > 0 || !more;
    ProfScope ps(ctx, mode ? "get_cost" : names[k]);
    hipLaunchKernelGGL(fn, dim3(n_jobs, by), dim3((mode ? (f.nw <= 4 ? 4 : 8) : f.nw) * 64), f.lds, ctx->stream, (const RegJob*)d_jobs, c);
    CFEAR_HIP_CHECK(ctx, hipGetLastError());
  }
  return CFEAR_OK;
}

// ---- host-facing wrappers ----------------------------------------------------------------------
namespace {

// what the host learns about a batch while it marshals it: the scratch size and which forms its registrations want
struct JobSizes {
  int pairs_cap = 1, cost = CFEAR_P2L;
  bool huber = true, small_pairs = true, any_large = false, any_huge = false;
  void add(int n_scans, int sum_pad, int max_pad, int n_src) {
    const int last = n_scans - 1, fields = reg_dense_fields(cost);
    pairs_cap = std::max(pairs_cap, last * std::max(n_src, 1));
    any_huge = any_huge || !mt_fit((uint32_t)(kLdsCu / 2 - 256 - kPartBigBytes), last, sum_pad, max_pad, n_src, fields, true).can;
    small_pairs = small_pairs && n_scans == 2 && mt_fit(kLdsPairs, last, sum_pad, max_pad, n_src, fields, false).good;
    any_large = any_large || !mt_fit(regular_lds(huber), last, sum_pad, max_pad, n_src, fields, false).good;
  }
  RegLaunchHint hint(int n_jobs) const { RegLaunchHint h; h.small_pairs = small_pairs && n_jobs > 0; h.big_pass = any_large; h.whole_cu = any_huge; return h; }
};

int gather_job(cfear_ctx* ctx, const cfear_scan* const* scans, int n_scans, const double* poses, unsigned char* dst, JobSizes& sz) {
  if (n_scans < 2 || n_scans > kMaxScans)
    return cfear_set_error(ctx, CFEAR_ERR_INVALID_ARGUMENT, "n_scans must be in [2,%d]", kMaxScans);
  ScanView views[kMaxScans];
  int sum_pad = 0, max_pad = 0, n_src = 0;
  for (int i = 0; i < n_scans; i++) {
    if (!scans[i]) return cfear_set_error(ctx, CFEAR_ERR_INVALID_ARGUMENT, "null scan handle");
    if (scans[i]->ctx != ctx) return cfear_set_error(ctx, CFEAR_ERR_INVALID_ARGUMENT, "scan belongs to another context");
    views[i] = scans[i]->view;
    const int nc = cfear_scan_size(scans[i]);
    if (nc < 0) return nc;
    if (i < n_scans - 1) { sum_pad += scan_grid_pad(nc); max_pad = std::max(max_pad, scan_grid_pad(nc)); }
    else n_src = nc;
  }
  sz.add(n_scans, sum_pad, max_pad, n_src);
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
  // job records are built in pinned memory: up to 1.9 KB each, so a 4096-candidate batch is an upload of megabytes that a
  // pageable source would stage synchronously at a fraction of the PCIe rate
  int max_scans = 2;
  for (int j = 0; j < n_jobs; j++) max_scans = std::max(max_scans, std::min(jobs[j].n_scans, kMaxScans));
  const size_t stride = reg_job_stride(max_scans);
  const size_t jb = (size_t)n_jobs * stride, rb = (size_t)n_jobs * sizeof(cfear_reg_result);
  unsigned char* hjobs = (unsigned char*)cfear_pinned(ctx, jb);
  if (!hjobs) return cfear_set_error(ctx, CFEAR_ERR_HIP, "pinned staging allocation failed");
  JobSizes sz;
  sz.cost = par->cost; sz.huber = par->loss == CFEAR_LOSS_HUBER;
  for (int j = 0; j < n_jobs; j++) {
    rc = gather_job(ctx, jobs[j].scans, jobs[j].n_scans, jobs[j].poses_xyt, hjobs + (size_t)j * stride, sz);
    if (rc != CFEAR_OK) return rc;
  }
  const size_t sb = reg_scratch_bytes(sz.pairs_cap) * (size_t)n_jobs;
  char* ws = (char*)cfear_workspace(ctx, 6, jb + rb + 512);
  char* scr = (char*)cfear_workspace(ctx, 7, sb);
  if (!ws || !scr) return cfear_set_error(ctx, CFEAR_ERR_HIP, "workspace allocation failed");
  char* d_jobs = ws;
  cfear_reg_result* d_res = d_out ? d_out : (cfear_reg_result*)(ws + (jb + 255) / 256 * 256);
  if (d_used) *d_used = d_res;
  CFEAR_HIP_CHECK(ctx, hipMemcpyAsync(d_jobs, hjobs, jb, hipMemcpyHostToDevice, ctx->stream));
  if (d_out) cfear_pinned_mark(ctx);                        // results stay on the device: nothing below waits for this copy
  rc = cfear_register_launch(ctx, d_jobs, n_jobs, par, sz.pairs_cap, scr, d_res, nullptr, stride, sz.hint(n_jobs));
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
// The table holds a reference on every scan (cfear_scan::refs): a scan destroyed while a table still names it keeps its slab
// until the table goes too, so a candidate can never read another scan's cells through a recycled slab.
struct cfear_scan_table {
  cfear_ctx* ctx = nullptr;
  ScanView* d_views = nullptr;          // [n] device
  std::vector<int32_t> n_cells;         // host copy of the cell counts (launch geometry)
  std::vector<cfear_scan*> scans;       // referenced handles
};

namespace {
// candidate -> the job record matcher_kernel reads: scans {target, source}, poses {target, source guess}
__global__ __launch_bounds__(256) void expand_candidates_kernel(const ScanView* __restrict__ views, const cfear_candidate* __restrict__ cands,
                                                                int n, char* __restrict__ jobs, size_t stride, int32_t* __restrict__ trailer,
                                                                int trailer_status) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i == 0 && trailer) { trailer[0] = trailer_status; trailer[1] = n; }   // the sharded step's {rank status, records} (shard.hip)
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
  t->scans.resize((size_t)n_scans);
  for (int i = 0; i < n_scans; i++) { t->scans[(size_t)i] = const_cast<cfear_scan*>(scans[i]); cfear_scan_retain(t->scans[(size_t)i]); }
  *out = t.release();
  return CFEAR_OK;
}

extern "C" int cfear_scan_table_size(const cfear_scan_table* t) { return t ? (int)t->n_cells.size() : CFEAR_ERR_INVALID_ARGUMENT; }

extern "C" int cfear_scan_table_destroy(cfear_scan_table* t) {
  if (!t) return CFEAR_OK;
  (void)hipSetDevice(t->ctx->device);
  (void)hipStreamSynchronize(t->ctx->stream);
  if (t->d_views) (void)hipFree(t->d_views);
  for (cfear_scan* s : t->scans) (void)cfear_scan_destroy(s);          // drops the table's reference
  delete t;
  return CFEAR_OK;
}

// A candidate batch in two halves (cfear_candidate_pipe runs them on different streams; cfear_candidates_enqueue is both in a row).
// expand: validates the candidates while it copies them into `h_stage` (PINNED host memory, n records, owned by the caller until
// the kernel has run) and turns them into job records at d_jobs (n * reg_job_stride(2) bytes) on `stream` -- no upload: pinned
// memory is mapped into the device's address space, the kernel reads the 56-byte records over PCIe itself.  d_trailer
// (optional, device int32[2]) receives {trailer_status, n}: the status trailer of a sharded step travels behind the block
// without an enqueue of its own.  geom receives what the matcher's launch needs to know about the batch.
int cfear_candidates_expand(cfear_ctx* ctx, hipStream_t stream, const cfear_scan_table* table, const cfear_candidate* cands, int32_t n,
                            const cfear_reg_params* par, cfear_candidate* h_stage, char* d_jobs, int32_t* d_trailer, int trailer_status,
                            CandGeometry* geom) {
  if (table->ctx != ctx) return cfear_set_error(ctx, CFEAR_ERR_INVALID_ARGUMENT, "table belongs to another context");
  const int nt = (int)table->n_cells.size();
  JobSizes sz;
  sz.cost = par->cost; sz.huber = par->loss == CFEAR_LOSS_HUBER;
  int max_tar = 0, max_src = 0;
  for (int i = 0; i < n; i++) {
    const cfear_candidate& c = cands[i];
    if (c.target < 0 || c.target >= nt || c.source < 0 || c.source >= nt)
      return cfear_set_error(ctx, CFEAR_ERR_INVALID_ARGUMENT, "candidate %d refers to scan %d / %d of a table of %d", i, c.target, c.source, nt);
    max_tar = std::max(max_tar, table->n_cells[(size_t)c.target]);
    max_src = std::max(max_src, table->n_cells[(size_t)c.source]);
    h_stage[i] = c;
  }
  // the launch geometry from the LARGEST target and source of the batch (a pair's needs grow with both: if that pair fits a
  // form, every candidate does) -- one evaluation per batch, not per candidate (4096 candidates: 0.1 ms of host time)
  sz.add(2, scan_grid_pad(max_tar), scan_grid_pad(max_tar), max_src);
  geom->pairs_cap = sz.pairs_cap; geom->hint = sz.hint(n);
  hipLaunchKernelGGL(expand_candidates_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, (const ScanView*)table->d_views,
                     (const cfear_candidate*)h_stage, n, d_jobs, reg_job_stride(2), d_trailer, trailer_status);
  CFEAR_HIP_CHECK(ctx, hipGetLastError());
  return CFEAR_OK;
}

// match: the matcher over the job records expand left at d_jobs, on the context's stream; records to d_res (device)
int cfear_candidates_match(cfear_ctx* ctx, const char* d_jobs, int32_t n, const cfear_reg_params* par, const CandGeometry* geom,
                           cfear_reg_result* d_res) {
  char* scr = (char*)cfear_workspace(ctx, 7, reg_scratch_bytes(geom->pairs_cap) * (size_t)n);
  if (!scr) return cfear_set_error(ctx, CFEAR_ERR_HIP, "workspace allocation failed");
  return cfear_register_launch(ctx, d_jobs, n, par, geom->pairs_cap, scr, d_res, nullptr, reg_job_stride(2), geom->hint);
}

int cfear_candidates_enqueue(cfear_ctx* ctx, const cfear_scan_table* table, const cfear_candidate* cands, int32_t n,
                             const cfear_reg_params* par, cfear_candidate* h_stage, cfear_reg_result* d_res, int32_t* d_trailer,
                             int trailer_status) {
  char* ws = (char*)cfear_workspace(ctx, 6, (size_t)n * reg_job_stride(2) + 512);
  if (!ws) return cfear_set_error(ctx, CFEAR_ERR_HIP, "workspace allocation failed");
  CandGeometry geom;
  const int rc = cfear_candidates_expand(ctx, ctx->stream, table, cands, n, par, h_stage, ws, d_trailer, trailer_status, &geom);
  if (rc != CFEAR_OK) return rc;
  return cfear_candidates_match(ctx, ws, n, par, &geom, d_res);
}

extern "C" int cfear_register_candidates(cfear_ctx* ctx, const cfear_scan_table* table, const cfear_candidate* cands, int32_t n,
                                         const cfear_reg_params* par, cfear_reg_result* results) {
  if (!ctx) return CFEAR_ERR_INVALID_ARGUMENT;
  if (!table || !results || n < 0 || (n > 0 && !cands)) return cfear_set_error(ctx, CFEAR_ERR_INVALID_ARGUMENT, "null argument");
  int rc = check_params(ctx, par);
  if (rc != CFEAR_OK || n == 0) return rc;
  CFEAR_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  const size_t cb = (size_t)n * sizeof(cfear_candidate), rb = (size_t)n * sizeof(cfear_reg_result);
  cfear_candidate* hc = (cfear_candidate*)cfear_pinned(ctx, cb);
  if (!hc) return cfear_set_error(ctx, CFEAR_ERR_HIP, "pinned staging allocation failed");
  const bool dev_out = cfear_is_device_ptr(results);
  cfear_reg_result* d_res = results;
  if (!dev_out) {
    d_res = (cfear_reg_result*)cfear_workspace(ctx, 13, rb);
    if (!d_res) return cfear_set_error(ctx, CFEAR_ERR_HIP, "workspace allocation failed");
  }
  rc = cfear_candidates_enqueue(ctx, table, cands, n, par, hc, d_res, nullptr, 0);
  if (dev_out) cfear_pinned_mark(ctx);                      // (no synchronisation below: the staging buffer stays in use)
  if (rc != CFEAR_OK) { (void)hipStreamSynchronize(ctx->stream); return rc; }
  if (dev_out) return CFEAR_OK;
  CFEAR_HIP_CHECK(ctx, hipMemcpyAsync(results, d_res, rb, hipMemcpyDeviceToHost, ctx->stream));
  CFEAR_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  return CFEAR_OK;
}

int cfear_check_reg_params(cfear_ctx* ctx, const cfear_reg_params* par) { return check_params(ctx, par); }

void cfear_reg_pair_geometry(const cfear_reg_params* par, int max_tar_cells, int max_src_cells, int* pairs_cap, RegLaunchHint* hint) {
  JobSizes sz;
  sz.cost = par->cost; sz.huber = par->loss == CFEAR_LOSS_HUBER;
  sz.add(2, scan_grid_pad(max_tar_cells), scan_grid_pad(max_tar_cells), max_src_cells);
  *pairs_cap = sz.pairs_cap;
  *hint = sz.hint(1);
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

// Runs the cost-only mode over `jobs`; out receives max(mode.n_samples, 1) records per job.
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
  sz.cost = par->cost; sz.huber = par->loss == CFEAR_LOSS_HUBER;
  for (int j = 0; j < n_jobs; j++) {
    unsigned char* dst = hjobs + (size_t)j * stride;
    rc = gather_job(ctx, jobs[j].scans, jobs[j].n_scans, jobs[j].poses_xyt, dst, sz);
    if (rc != CFEAR_OK) return rc;
    if (itrs) cfear_reg_job_set_itr(dst, itrs[j]);
  }
  // a few workgroups per job when the batch alone cannot fill the GPU; scratch bounded to 1 GiB per launch
  mode.blocks_per_job = std::max(1, std::min(m, (1024 + n_jobs - 1) / n_jobs));
  const size_t per = reg_scratch_bytes(sz.pairs_cap) * (size_t)mode.blocks_per_job;
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
    rc = cfear_register_launch(ctx, d_jobs + (size_t)j0 * stride, nj, par, sz.pairs_cap, scr, d_res + (size_t)j0 * m, &mode, stride,
                               sz.hint(nj));
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
  int pairs_cap = 0, n_src = 0, n_slots = 0, n_blocks = 0, n_scans = 0;
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
  cfear_cost* c = new cfear_cost();
  c->ctx = ctx; c->par = *par; c->pairs_cap = sz.pairs_cap; c->n_scans = n_scans;
  c->n_src = cfear_scan_size(scans[n_scans - 1]);
  c->n_slots = (n_scans - 1) * c->n_src;
  auto fail = [&](int status, const char* msg) { cfear_cost_destroy(c); return cfear_set_error(ctx, status, "%s", msg); };
  if (hipMalloc(&c->d_job, sizeof(RegJob) + 256) != hipSuccess) return fail(CFEAR_ERR_HIP, "hipMalloc failed");
  if (hipMalloc((void**)&c->d_scratch, slots_bytes(c->pairs_cap)) != hipSuccess) return fail(CFEAR_ERR_HIP, "hipMalloc failed");
  if (hipMalloc((void**)&c->d_out, ((size_t)c->pairs_cap * 10 + 16) * sizeof(double)) != hipSuccess) return fail(CFEAR_ERR_HIP, "hipMalloc failed");
  if (hipMemcpyAsync(c->d_job, hjob, sizeof(RegJob), hipMemcpyHostToDevice, ctx->stream) != hipSuccess) return fail(CFEAR_ERR_HIP, "memcpy failed");
  MatchCommon cm{};
  cm.par = *par; cm.angle_outlier = std::cos(M_PI / 6.0);
  cm.scratch = c->d_scratch; cm.scratch_stride = 0;
  cm.job_stride = sizeof(RegJob);
  cm.pairs_cap = c->pairs_cap; cm.results = nullptr;
  cm.dense_fields = reg_dense_fields(par->cost);
  cm.lds_total = (uint32_t)(kLdsCu - 256);
  int32_t* d_nb = (int32_t*)((char*)c->d_job + sizeof(RegJob));
  if (cfear_allow_lds(ctx, (const void*)assoc_kernel, kLdsCu) != CFEAR_OK) return fail(CFEAR_ERR_HIP, "hipFuncSetAttribute failed");
  hipLaunchKernelGGL(assoc_kernel, dim3(1), dim3(256), (size_t)cm.lds_total, ctx->stream, (const RegJob*)c->d_job, cm, (int)itr, d_nb);
  if (hipGetLastError() != hipSuccess) return fail(CFEAR_ERR_HIP, "assoc_kernel launch failed");
  c->h_w.assign(std::max(c->n_slots, 1), -1.0);
  c->h_tidx.assign(std::max(c->n_slots, 1), -1);
  int32_t nb = 0;
  bool ok = hipMemcpyAsync(&nb, d_nb, 4, hipMemcpyDeviceToHost, ctx->stream) == hipSuccess;
  if (c->n_slots > 0) {
    ok = ok && hipMemcpyAsync(c->h_w.data(), (double*)c->d_scratch + 5 * (size_t)c->pairs_cap, (size_t)c->n_slots * 8,
                              hipMemcpyDeviceToHost, ctx->stream) == hipSuccess;
    ok = ok && hipMemcpyAsync(c->h_tidx.data(), (double*)c->d_scratch + 6 * (size_t)c->pairs_cap, (size_t)c->n_slots * 4,
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
  MatchCommon cm{};
  cm.par = c->par; cm.scratch = c->d_scratch; cm.job_stride = sizeof(RegJob);
  cm.pairs_cap = c->pairs_cap;
  EvalOut o;
  const size_t sc = (size_t)c->pairs_cap;
  o.raw_r = want_raw ? c->d_out : nullptr;
  o.raw_j = want_raw ? c->d_out + 2 * sc : nullptr;
  o.rob_r = c->d_out + 8 * sc;
  o.neq = c->d_out + 10 * sc;
  hipLaunchKernelGGL(eval_kernel, dim3(1), dim3(256), kFixedLds, ctx->stream, (const RegJob*)c->d_job, cm, x[0], x[1], x[2], o);
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
  const size_t sc = (size_t)c->pairs_cap;
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
  CFEAR_HIP_CHECK(ctx, hipMemcpyAsync(neq, c->d_out + 10 * (size_t)c->pairs_cap, sizeof(neq), hipMemcpyDeviceToHost, ctx->stream));
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
  const size_t sc = (size_t)c->pairs_cap;
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
