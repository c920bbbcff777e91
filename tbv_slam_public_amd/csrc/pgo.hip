// pgo.hip -- pose-graph optimisation (host code; SURVEY.md 8f-4, second half: the step AFTER the path).
//
// Replaces (tbv_slam/):
//   CeresLeastSquares::{Solve, BuildOptimizationProblem, AddConstraintType, SolveOptimizationProblem}
//                                                      src/tbv_slam/ceresoptimizer.cpp:13-113
//   PoseGraph3dErrorTerm (AutoDiff<6, 3, 4, 3, 4>)     include/tbv_slam/ceresoptimizer.h:55-112
//   ceres::EigenQuaternionParameterization, ceres::CauchyLoss, ceres::Solve (TRUST_REGION / LEVENBERG_MARQUARDT,
//   SPARSE_NORMAL_CHOLESKY, max_num_iterations 200)    third-party, Ceres 2.1.0
//
// Why this runs on the host: the problem is one sparse nonlinear least squares over a CHAIN of poses (6 (n - 1) unknowns,
// n ~ 5 000 keyframes, block-tridiagonal normal equations plus a few hundred weak loop blocks), solved once per loop
// closure.  Its kernels are block-tridiagonal recurrences -- 2 n dependent 6 x 6 steps per solve, a serial chain that a
// CPU core finishes in well under a millisecond and a GPU wavefront in tens -- and the whole optimisation is ~10^8
// flops, a thousandth of one registration batch.  The reference keeps it on the host as well ("sparse PGO, runs
// once").  The trust-region bookkeeping is the one restated for the matcher (matcher.hip lm_round).
//
// Linear algebra: Ceres factorises J^T J + D^2 with CHOLMOD.  Here the same system is solved by conjugate gradients
// preconditioned with the exact block-tridiagonal Cholesky factor of its odometry chain part; the loop constraints enter
// the normal equations scaled by 1 / loop_scaling = 2e-6 (ceresoptimizer.cpp:85), so the preconditioned system is a tiny
// perturbation of the identity and CG reaches 1e-13 relative residual in a handful of iterations -- the steps agree with
// a direct solve to rounding.
#include <algorithm>
#include <cmath>
#include <limits>
#include <cstring>
#include <vector>

#include "../../include/cfear_hip.h"

namespace {

// ---- forward-mode automatic differentiation over the 14 parameters (p_a 3, q_a 4, p_b 3, q_b 4), like ceres::Jet ------
constexpr int NJ = 14;
struct Jet {
  double a;
  double v[NJ];
  Jet() : a(0) { for (int i = 0; i < NJ; i++) v[i] = 0; }
  Jet(double x) : a(x) { for (int i = 0; i < NJ; i++) v[i] = 0; }
  Jet(double x, int k) : a(x) { for (int i = 0; i < NJ; i++) v[i] = 0; v[k] = 1; }
};
inline Jet operator+(const Jet& x, const Jet& y) { Jet r; r.a = x.a + y.a; for (int i = 0; i < NJ; i++) r.v[i] = x.v[i] + y.v[i]; return r; }
inline Jet operator-(const Jet& x, const Jet& y) { Jet r; r.a = x.a - y.a; for (int i = 0; i < NJ; i++) r.v[i] = x.v[i] - y.v[i]; return r; }
inline Jet operator-(const Jet& x) { Jet r; r.a = -x.a; for (int i = 0; i < NJ; i++) r.v[i] = -x.v[i]; return r; }
inline Jet operator*(const Jet& x, const Jet& y) { Jet r; r.a = x.a * y.a; for (int i = 0; i < NJ; i++) r.v[i] = x.a * y.v[i] + x.v[i] * y.a; return r; }

template <typename T> struct Quat { T x, y, z, w; };
template <typename T> Quat<T> qmul(const Quat<T>& a, const Quat<T>& b) {       // Eigen::Quaternion operator*
  return Quat<T>{a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y, a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z,
                 a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x, a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z};
}
template <typename T> Quat<T> qconj(const Quat<T>& a) { return Quat<T>{-a.x, -a.y, -a.z, a.w}; }
// Eigen::Quaternion * Vector3: v + 2 w (u x v) + 2 u x (u x v), evaluated as Eigen's _transformVector does
template <typename T> void qrot(const Quat<T>& q, const T v[3], T out[3]) {
  const T ux = q.y * v[2] - q.z * v[1], uy = q.z * v[0] - q.x * v[2], uz = q.x * v[1] - q.y * v[0];
  const T two(2.0);
  const T tx = two * ux, ty = two * uy, tz = two * uz;
  out[0] = v[0] + q.w * tx + (q.y * tz - q.z * ty);
  out[1] = v[1] + q.w * ty + (q.z * tx - q.x * tz);
  out[2] = v[2] + q.w * tz + (q.x * ty - q.y * tx);
}

// PoseGraph3dErrorTerm::operator() (ceresoptimizer.h:62-97): residual = L * [p_ab_est - p_ab_meas; 2 vec(q_meas * q_ab_est^-1)]
template <typename T>
void error_term(const T pa[3], const Quat<T>& qa, const T pb[3], const Quat<T>& qb, const cfear_pose3d& meas, const double L[36], T r[6]) {
  const Quat<T> qa_inv = qconj(qa);
  const Quat<T> q_ab = qmul(qa_inv, qb);
  const T d[3] = {pb[0] - pa[0], pb[1] - pa[1], pb[2] - pa[2]};
  T p_ab[3];
  qrot(qa_inv, d, p_ab);
  const Quat<T> qm{T(meas.q[0]), T(meas.q[1]), T(meas.q[2]), T(meas.q[3])};
  const Quat<T> dq = qmul(qm, qconj(q_ab));
  T e[6] = {p_ab[0] - T(meas.p[0]), p_ab[1] - T(meas.p[1]), p_ab[2] - T(meas.p[2]), T(2.0) * dq.x, T(2.0) * dq.y, T(2.0) * dq.z};
  for (int i = 0; i < 6; i++) {                                  // residuals.applyOnTheLeft(sqrt_information)
    T s(0.0);
    for (int k = 0; k < 6; k++) s = s + T(L[i * 6 + k]) * e[k];
    r[i] = s;
  }
}

struct Con {
  int a, b;                 // node indices (0 = the fixed first node)
  cfear_pose3d meas;
  double L[36];             // sqrt_information = I_scaled.llt().matrixL(), row-major
  bool cauchy;
  double r[6];              // robustified residuals at the current point
  double J[6][12];          // robustified, column-scaled Jacobian wrt the tangent (3 + 3 per node): a then b
};

bool llt6(const double A[36], double L[36]) {
  memset(L, 0, 36 * sizeof(double));
  for (int i = 0; i < 6; i++)
    for (int j = 0; j <= i; j++) {
      double s = A[i * 6 + j];
      for (int k = 0; k < j; k++) s -= L[i * 6 + k] * L[j * 6 + k];
      if (i == j) { if (!(s > 0.0)) return false; L[i * 6 + i] = std::sqrt(s); }
      else L[i * 6 + j] = s / L[j * 6 + j];
    }
  return true;
}

struct Problem {
  int n = 0;                                   // nodes; unknown blocks are nodes 1 .. n-1
  std::vector<cfear_pose3d> x;                 // current point
  std::vector<Con> cons;
  double cauchy_a = 0.1;
  std::vector<double> scale;                   // [6 n] jacobi column scaling (node 0 unused)
  bool have_scale = false;

  // cost at `pts`; with_jac: also residuals and (unscaled here, scaled by the caller) Jacobians into cons
  double evaluate(const std::vector<cfear_pose3d>& pts, bool with_jac) {
    double cost = 0.0;
    for (Con& c : cons) {
      const cfear_pose3d &A = pts[c.a], &B = pts[c.b];
      double rho0, rho1;
      if (!with_jac) {
        const Quat<double> qa{A.q[0], A.q[1], A.q[2], A.q[3]}, qb{B.q[0], B.q[1], B.q[2], B.q[3]};
        double r[6];
        error_term<double>(A.p, qa, B.p, qb, c.meas, c.L, r);
        double s = 0;
        for (int i = 0; i < 6; i++) s += r[i] * r[i];
        loss(c.cauchy, s, rho0, rho1);
        cost += 0.5 * rho0;
        continue;
      }
      Jet pa[3] = {Jet(A.p[0], 0), Jet(A.p[1], 1), Jet(A.p[2], 2)}, pb[3] = {Jet(B.p[0], 7), Jet(B.p[1], 8), Jet(B.p[2], 9)};
      const Quat<Jet> qa{Jet(A.q[0], 3), Jet(A.q[1], 4), Jet(A.q[2], 5), Jet(A.q[3], 6)};
      const Quat<Jet> qb{Jet(B.q[0], 10), Jet(B.q[1], 11), Jet(B.q[2], 12), Jet(B.q[3], 13)};
      Jet r[6];
      error_term<Jet>(pa, qa, pb, qb, c.meas, c.L, r);
      double s = 0;
      for (int i = 0; i < 6; i++) s += r[i].a * r[i].a;
      loss(c.cauchy, s, rho0, rho1);
      cost += 0.5 * rho0;
      const double sr = std::sqrt(rho1);                         // Corrector, alpha = 0 (rho'' <= 0 for Cauchy)
      // EigenQuaternionParameterization::ComputeJacobian (4 x 3, Eigen coefficient order x, y, z, w)
      auto local = [](const double q[4], double G[12]) {
        G[0] = q[3];  G[1] = q[2];  G[2] = -q[1];
        G[3] = -q[2]; G[4] = q[3];  G[5] = q[0];
        G[6] = q[1];  G[7] = -q[0]; G[8] = q[3];
        G[9] = -q[0]; G[10] = -q[1]; G[11] = -q[2];
      };
      double Ga[12], Gb[12];
      local(A.q, Ga); local(B.q, Gb);
      for (int i = 0; i < 6; i++) {
        c.r[i] = r[i].a * sr;
        for (int k = 0; k < 3; k++) { c.J[i][k] = r[i].v[k] * sr; c.J[i][6 + k] = r[i].v[7 + k] * sr; }
        for (int k = 0; k < 3; k++) {
          double sa = 0, sb = 0;
          for (int t = 0; t < 4; t++) { sa += r[i].v[3 + t] * Ga[t * 3 + k]; sb += r[i].v[10 + t] * Gb[t * 3 + k]; }
          c.J[i][3 + k] = sa * sr; c.J[i][9 + k] = sb * sr;
        }
      }
    }
    return cost;
  }
  void loss(bool cauchy, double s, double& rho0, double& rho1) const {
    if (!cauchy) { rho0 = s; rho1 = 1.0; return; }
    const double b = cauchy_a * cauchy_a, cc = 1.0 / b, sum = 1.0 + s * cc, inv = 1.0 / sum;   // ceres::CauchyLoss
    rho0 = b * std::log(sum);
    rho1 = std::max(std::numeric_limits<double>::min(), inv);
  }
  // column norms / scaling and column scaling of J (node 0's columns are constant: zeroed)
  void column_sq_norms(std::vector<double>& out) const {
    out.assign((size_t)6 * n, 0.0);
    for (const Con& c : cons)
      for (int i = 0; i < 6; i++)
        for (int k = 0; k < 6; k++) { out[(size_t)6 * c.a + k] += c.J[i][k] * c.J[i][k]; out[(size_t)6 * c.b + k] += c.J[i][6 + k] * c.J[i][6 + k]; }
  }
  void scale_columns() {
    for (Con& c : cons)
      for (int i = 0; i < 6; i++)
        for (int k = 0; k < 6; k++) {
          c.J[i][k] *= c.a == 0 ? 0.0 : scale[(size_t)6 * c.a + k];
          c.J[i][6 + k] *= c.b == 0 ? 0.0 : scale[(size_t)6 * c.b + k];
        }
  }
  void gradient(std::vector<double>& g) const {                  // J^T r
    g.assign((size_t)6 * n, 0.0);
    for (const Con& c : cons)
      for (int i = 0; i < 6; i++)
        for (int k = 0; k < 6; k++) { g[(size_t)6 * c.a + k] += c.J[i][k] * c.r[i]; g[(size_t)6 * c.b + k] += c.J[i][6 + k] * c.r[i]; }
  }
  void Jv(const std::vector<double>& v, std::vector<double>& t) const {   // t[6 m] = J v
    t.assign(cons.size() * 6, 0.0);
    for (size_t ci = 0; ci < cons.size(); ci++) {
      const Con& c = cons[ci];
      for (int i = 0; i < 6; i++) {
        double s = 0;
        for (int k = 0; k < 6; k++) s += c.J[i][k] * v[(size_t)6 * c.a + k] + c.J[i][6 + k] * v[(size_t)6 * c.b + k];
        t[ci * 6 + i] = s;
      }
    }
  }
  void JtJv(const std::vector<double>& v, const std::vector<double>& d2, std::vector<double>& out, std::vector<double>& t) const {
    Jv(v, t);
    out.assign((size_t)6 * n, 0.0);
    for (size_t ci = 0; ci < cons.size(); ci++) {
      const Con& c = cons[ci];
      for (int i = 0; i < 6; i++)
        for (int k = 0; k < 6; k++) { out[(size_t)6 * c.a + k] += c.J[i][k] * t[ci * 6 + i]; out[(size_t)6 * c.b + k] += c.J[i][6 + k] * t[ci * 6 + i]; }
    }
    for (size_t k = 6; k < out.size(); k++) out[k] += d2[k] * v[k];
    for (int k = 0; k < 6; k++) out[k] = 0.0;
  }
};

// Block-tridiagonal Cholesky of the chain part of J^T J + D^2 (nodes 1 .. n-1): diagonal blocks from ALL constraints,
// sub-diagonal blocks from the constraints between neighbouring nodes.  Falls back to the block diagonal if a pivot fails.
struct ChainPrecond {
  int n = 0;
  std::vector<double> Ld, Lo;   // [n][36] diagonal factors (lower), [n][36] sub-diagonal factors L(i, i-1)
  bool build(const Problem& P, const std::vector<double>& d2, bool with_chain) {
    n = P.n;
    std::vector<double> D((size_t)n * 36, 0.0), E((size_t)n * 36, 0.0);   // E[i] = H(i, i-1)
    for (const Con& c : P.cons) {
      for (int r = 0; r < 6; r++)
        for (int q = 0; q < 6; q++) {
          double saa = 0, sbb = 0, sba = 0;
          for (int i = 0; i < 6; i++) { saa += c.J[i][r] * c.J[i][q]; sbb += c.J[i][6 + r] * c.J[i][6 + q]; sba += c.J[i][6 + r] * c.J[i][q]; }
          D[(size_t)c.a * 36 + r * 6 + q] += saa;
          D[(size_t)c.b * 36 + r * 6 + q] += sbb;
          if (with_chain && c.b == c.a + 1) E[(size_t)c.b * 36 + r * 6 + q] += sba;                 // H(b, a)
          if (with_chain && c.a == c.b + 1) E[(size_t)c.a * 36 + q * 6 + r] += sba;                 // H(a, b) = H(b, a)^T
        }
    }
    for (int i = 1; i < n; i++) for (int k = 0; k < 6; k++) D[(size_t)i * 36 + k * 7] += d2[(size_t)6 * i + k];
    Ld.assign((size_t)n * 36, 0.0); Lo.assign((size_t)n * 36, 0.0);
    for (int i = 1; i < n; i++) {
      double S[36];
      memcpy(S, &D[(size_t)i * 36], sizeof(S));
      if (i > 1) {
        // Lo_i = E_i * Ld_{i-1}^-T ; S -= Lo_i Lo_i^T
        const double* Lp = &Ld[(size_t)(i - 1) * 36];
        double* X = &Lo[(size_t)i * 36];
        for (int r = 0; r < 6; r++)                                   // solve X Lp^T = E row by row (forward substitution)
          for (int q = 0; q < 6; q++) {
            double s = E[(size_t)i * 36 + r * 6 + q];
            for (int k = 0; k < q; k++) s -= X[r * 6 + k] * Lp[q * 6 + k];
            X[r * 6 + q] = s / Lp[q * 6 + q];
          }
        for (int r = 0; r < 6; r++) for (int q = 0; q < 6; q++) { double s = 0; for (int k = 0; k < 6; k++) s += X[r * 6 + k] * X[q * 6 + k]; S[r * 6 + q] -= s; }
      }
      if (!llt6(S, &Ld[(size_t)i * 36])) return false;
    }
    return true;
  }
  void apply(const std::vector<double>& r, std::vector<double>& z) const {   // z = M^-1 r
    z.assign(r.size(), 0.0);
    for (int i = 1; i < n; i++) {                                     // forward: L y = r
      double rhs[6];
      for (int k = 0; k < 6; k++) rhs[k] = r[(size_t)6 * i + k];
      if (i > 1) for (int k = 0; k < 6; k++) for (int t = 0; t < 6; t++) rhs[k] -= Lo[(size_t)i * 36 + k * 6 + t] * z[(size_t)6 * (i - 1) + t];
      const double* L = &Ld[(size_t)i * 36];
      for (int k = 0; k < 6; k++) { double s = rhs[k]; for (int t = 0; t < k; t++) s -= L[k * 6 + t] * z[(size_t)6 * i + t]; z[(size_t)6 * i + k] = s / L[k * 6 + k]; }
    }
    for (int i = n - 1; i >= 1; i--) {                                // backward: L^T x = y
      double rhs[6];
      for (int k = 0; k < 6; k++) rhs[k] = z[(size_t)6 * i + k];
      if (i + 1 < n) for (int k = 0; k < 6; k++) for (int t = 0; t < 6; t++) rhs[k] -= Lo[(size_t)(i + 1) * 36 + t * 6 + k] * z[(size_t)6 * (i + 1) + t];
      const double* L = &Ld[(size_t)i * 36];
      for (int k = 5; k >= 0; k--) { double s = rhs[k]; for (int t = k + 1; t < 6; t++) s -= L[t * 6 + k] * z[(size_t)6 * i + t]; z[(size_t)6 * i + k] = s / L[k * 6 + k]; }
    }
  }
};

double dot(const std::vector<double>& a, const std::vector<double>& b) { double s = 0; for (size_t i = 0; i < a.size(); i++) s += a[i] * b[i]; return s; }

// x_plus_delta: p += dp; q = exp(dq) * q (ceres::EigenQuaternionParameterization::Plus)
void plus(const cfear_pose3d& x, const double d[6], cfear_pose3d& out) {
  for (int k = 0; k < 3; k++) out.p[k] = x.p[k] + d[k];
  const double nd = std::sqrt(d[3] * d[3] + d[4] * d[4] + d[5] * d[5]);
  if (nd > 0.0) {
    const double s = std::sin(nd) / nd;
    const Quat<double> dq{s * d[3], s * d[4], s * d[5], std::cos(nd)}, q{x.q[0], x.q[1], x.q[2], x.q[3]};
    const Quat<double> r = qmul(dq, q);
    out.q[0] = r.x; out.q[1] = r.y; out.q[2] = r.z; out.q[3] = r.w;
  } else {
    for (int k = 0; k < 4; k++) out.q[k] = x.q[k];
  }
}

}  // namespace

extern "C" void cfear_pgo_params_default(cfear_pgo_params* p) {          // CeresLeastSquares::Parameters (ceresoptimizer.cpp:18-27)
  p->loop_vxx = 0.01; p->loop_vyy = 0.01; p->loop_vtt = 0.001;
  p->odom_vxx = 0.01; p->odom_vyy = 0.01; p->odom_vtt = 0.001;
  p->loop_scaling = 500000;
  p->replace_cov_by_identity = 1;
  p->max_num_iterations = 200;                                           // :52
  p->loop_loss_limit = 0.1;                                              // :36 CauchyLoss(0.1)
}

extern "C" int cfear_pgo_solve(cfear_pose3d* poses, const uint64_t* ids, int32_t n, const cfear_graph_constraint* constraints,
                               int32_t m, const cfear_pgo_params* par, cfear_pgo_summary* summary) {
  if (!poses || !ids || n < 1 || (m > 0 && !constraints) || m < 1 || !par || !summary) return CFEAR_ERR_INVALID_ARGUMENT;   // CHECK(size != 0)
  memset(summary, 0, sizeof(*summary));
  for (int i = 1; i < n; i++) if (!(ids[i - 1] < ids[i])) return CFEAR_ERR_INVALID_ARGUMENT;   // the node map is ordered by id
  auto find = [&](uint64_t id) { const uint64_t* p = std::lower_bound(ids, ids + n, id); return (p != ids + n && *p == id) ? (int)(p - ids) : -1; };
  Problem P;
  P.n = n;
  P.x.assign(poses, poses + n);
  P.cauchy_a = par->loop_loss_limit;
  for (int pass = 0; pass < 2; pass++)                                   // AddConstraintType(odometry), then (loop_appearance)
    for (int j = 0; j < m; j++) {
      const cfear_graph_constraint& c = constraints[j];
      if (c.type != pass) continue;                                      // mini_loop / candidate constraints are not optimised
      Con k;
      k.a = find(c.id_begin); k.b = find(c.id_end);
      if (k.a < 0 || k.b < 0) return CFEAR_ERR_INVALID_ARGUMENT;         // "Nodes doesn't exist" (:74-75)
      k.meas = c.t_be;
      k.cauchy = pass == 1;
      const double loop_scale_factor = pass == 1 ? 1.0 / par->loop_scaling : 1.0;     // :85
      double I[36] = {0};
      if (par->replace_cov_by_identity) {                                // :86-88: the odom_* variances scale BOTH types
        const double d[6] = {1.0 / par->odom_vxx, 1.0 / par->odom_vyy, 1, 1, 1, 1.0 / par->odom_vtt};
        for (int t = 0; t < 6; t++) I[t * 7] = d[t] * loop_scale_factor;
      } else {
        for (int t = 0; t < 36; t++) I[t] = c.information[t] * loop_scale_factor;
      }
      if (!llt6(I, k.L)) return CFEAR_ERR_INVALID_ARGUMENT;              // Eigen's llt() of a non-SPD matrix is garbage; refuse
      P.cons.push_back(k);
    }
  if (P.cons.empty()) return CFEAR_ERR_INVALID_ARGUMENT;
  // ---- ceres::Solve: trust-region Levenberg-Marquardt (Ceres 2.1 defaults) ------------------------
  const double function_tolerance = 1e-6, gradient_tolerance = 1e-10, parameter_tolerance = 1e-8;
  const double min_relative_decrease = 1e-3, min_lm_diagonal = 1e-6, max_lm_diagonal = 1e32, max_radius = 1e16, min_radius = 1e-32;
  double radius = 1e4, decrease_factor = 2.0;
  bool reuse_diagonal = false;
  std::vector<double> diagonal, g, gs, d2, y, step, t, Ap, r, z, p;
  double x_cost = P.evaluate(P.x, true);
  P.gradient(g);                                                         // unscaled Jacobian
  for (int k = 0; k < 6; k++) g[k] = 0.0;                                // the first node is constant
  {
    std::vector<double> nrm;
    P.column_sq_norms(nrm);
    P.scale.assign((size_t)6 * n, 1.0);
    for (size_t k = 6; k < nrm.size(); k++) P.scale[k] = 1.0 / (1.0 + std::sqrt(nrm[k]));
    P.scale_columns();
  }
  auto max_abs = [](const std::vector<double>& v) { double m2 = 0; for (double e : v) m2 = std::max(m2, std::fabs(e)); return m2; };
  auto x_norm_of = [&]() { double s = 0; for (int i = 1; i < n; i++) { for (int k = 0; k < 3; k++) s += P.x[i].p[k] * P.x[i].p[k]; for (int k = 0; k < 4; k++) s += P.x[i].q[k] * P.x[i].q[k]; } return std::sqrt(s); };
  double gradient_max_norm = max_abs(g), x_norm = x_norm_of();
  summary->initial_cost = x_cost;
  double min_cost = x_cost, it_cost = x_cost, it_rel = 0.0;
  bool it_success = true, usable = true;
  int iteration = 0, invalid = 0, n_pushed = 0;
  std::vector<cfear_pose3d> cand((size_t)n);
  for (;;) {
    n_pushed++;
    min_cost = std::min(min_cost, it_cost);
    if (iteration >= par->max_num_iterations) break;
    if (it_success && gradient_max_norm <= gradient_tolerance) break;
    if (radius <= min_radius) break;
    iteration++;
    it_cost = 0.0; it_rel = 0.0; it_success = false;
    // LevenbergMarquardtStrategy::ComputeStep
    if (!reuse_diagonal) {
      P.column_sq_norms(diagonal);                                       // of the scaled Jacobian
      for (double& dd : diagonal) dd = std::min(std::max(dd, min_lm_diagonal), max_lm_diagonal);
    }
    d2.assign((size_t)6 * n, 0.0);
    for (size_t k = 6; k < d2.size(); k++) d2[k] = diagonal[k] / radius;
    P.gradient(gs);                                                      // J_s^T r
    for (int k = 0; k < 6; k++) gs[k] = 0.0;
    // (J_s^T J_s + D^2) y = J_s^T r by preconditioned conjugate gradients
    ChainPrecond M;
    if (!M.build(P, d2, true) && !M.build(P, d2, false)) { usable = false; break; }
    y.assign((size_t)6 * n, 0.0);
    r = gs;
    M.apply(r, z);
    p = z;
    double rz = dot(r, z);
    const double r0 = std::sqrt(dot(r, r));
    int cg = 0;
    for (; cg < 500 && r0 > 0.0; cg++) {
      P.JtJv(p, d2, Ap, t);
      const double pAp = dot(p, Ap);
      if (!(pAp > 0.0)) break;
      const double alpha = rz / pAp;
      for (size_t k = 0; k < y.size(); k++) { y[k] += alpha * p[k]; r[k] -= alpha * Ap[k]; }
      if (std::sqrt(dot(r, r)) <= 1e-13 * r0) { cg++; break; }
      M.apply(r, z);
      const double rz2 = dot(r, z);
      const double beta = rz2 / rz;
      rz = rz2;
      for (size_t k = 0; k < p.size(); k++) p[k] = z[k] + beta * p[k];
    }
    summary->linear_iterations += cg;
    reuse_diagonal = true;
    step.assign(y.size(), 0.0);
    bool finite = true;
    for (size_t k = 0; k < y.size(); k++) { step[k] = -y[k]; finite = finite && std::isfinite(y[k]); }
    // model_cost_change = -(J step)^T (r + J step / 2)
    double model_cost_change = 0.0;
    if (finite) {
      P.Jv(step, t);
      for (size_t ci = 0; ci < P.cons.size(); ci++) for (int i = 0; i < 6; i++) model_cost_change -= t[ci * 6 + i] * (P.cons[ci].r[i] + t[ci * 6 + i] / 2.0);
    }
    if (!finite || !(model_cost_change > 0.0)) {                         // HandleInvalidStep
      if (++invalid >= 5) { usable = false; break; }
      radius /= decrease_factor; decrease_factor *= 2.0; reuse_diagonal = true;
      it_cost = x_cost; it_success = false; it_rel = 0.0;
      continue;
    }
    invalid = 0;
    double step_norm2 = 0.0;
    cand[0] = P.x[0];
    for (int i = 1; i < n; i++) {
      double d[6];
      for (int k = 0; k < 6; k++) d[k] = step[(size_t)6 * i + k] * P.scale[(size_t)6 * i + k];
      plus(P.x[i], d, cand[i]);
      for (int k = 0; k < 3; k++) step_norm2 += (P.x[i].p[k] - cand[i].p[k]) * (P.x[i].p[k] - cand[i].p[k]);
      for (int k = 0; k < 4; k++) step_norm2 += (P.x[i].q[k] - cand[i].q[k]) * (P.x[i].q[k] - cand[i].q[k]);
    }
    const double cand_cost = P.evaluate(cand, false);
    if (std::sqrt(step_norm2) <= parameter_tolerance * (x_norm + parameter_tolerance)) break;    // ParameterToleranceReached
    const double cost_change = x_cost - cand_cost;
    if (std::fabs(cost_change) <= function_tolerance * x_cost) break;                            // FunctionToleranceReached
    it_rel = cost_change / model_cost_change;
    if (it_rel > min_relative_decrease) {                                // HandleSuccessfulStep
      P.x = cand;
      x_norm = x_norm_of();
      x_cost = P.evaluate(P.x, true);
      P.gradient(g);
      for (int k = 0; k < 6; k++) g[k] = 0.0;
      gradient_max_norm = max_abs(g);
      P.scale_columns();
      it_cost = x_cost; it_success = true;
      const double q = 2.0 * it_rel - 1.0;
      radius = std::min(max_radius, radius / std::max(1.0 / 3.0, 1.0 - q * q * q));
      decrease_factor = 2.0; reuse_diagonal = false;
    } else {
      it_cost = cand_cost; it_success = false;
      radius /= decrease_factor; decrease_factor *= 2.0; reuse_diagonal = true;
    }
  }
  memcpy(poses, P.x.data(), (size_t)n * sizeof(cfear_pose3d));
  summary->final_cost = std::min(summary->initial_cost, min_cost);       // solver.cc SetSummaryFinalCost
  summary->iterations = n_pushed - 1;
  summary->usable = usable ? 1 : 0;
  summary->num_residual_blocks = (int32_t)P.cons.size();
  return CFEAR_OK;
}
