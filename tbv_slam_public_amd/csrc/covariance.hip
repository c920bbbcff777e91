// covariance.hip -- covariance of a registration by cost sampling: host half.
//
// Replaces the dense-algebra tail of OdometryKeyframeFuser::approximateCovarianceBySampling
// (cfear_radarodometry/src/cfear_radarodometry/odometrykeyframefuser.cpp:318-377) and of its copy
// loopclosure::approximateCovarianceBySampling (tbv_slam/src/tbv_slam/loopclosure.cpp:146-205):
// quadratic least-squares fit of the n^3 cost samples, convexity test, 2 H^-1 scaled by the
// registration score.  The samples themselves -- n^3 GetCost evaluations per registration, the
// expensive part -- come from the cost-only mode of matcher_kernel (matcher.hip), one launch for
// all registrations of a batch.  The fit is a 27 x 10 problem per registration: host work, as in the
// reference.
#include <algorithm>
#include <cmath>
#include <limits>
#include <vector>

#include "common.hpp"

namespace {

// Minimum-norm least squares through a one-sided Jacobi (Hestenes) SVD -- the quantity Eigen's
// A.bdcSvd(ComputeThinU | ComputeThinV).solve(b) returns (odometrykeyframefuser.cpp:337), including
// its rank decision: singular values <= sigma_max * min(m, n) * eps count as zero.
// W (rows x cols, column-major here) is rotated until its columns are orthogonal: W R = U S.
void lstsq_jacobi(const std::vector<double>& A /*m x n row-major*/, const std::vector<double>& b, int m, int n,
                  double* x /*[n]*/) {
  const bool tall = m >= n;
  const int rows = tall ? m : n, cols = tall ? n : m;          // work on A (tall) or A^T (wide)
  std::vector<double> W((size_t)rows * cols), R((size_t)cols * cols, 0.0);
  for (int i = 0; i < m; i++)
    for (int j = 0; j < n; j++) {
      if (tall) W[(size_t)j * rows + i] = A[(size_t)i * n + j];
      else W[(size_t)i * rows + j] = A[(size_t)i * n + j];
    }
  for (int j = 0; j < cols; j++) R[(size_t)j * cols + j] = 1.0;
  const double eps = std::numeric_limits<double>::epsilon();
  for (int sweep = 0; sweep < 60; sweep++) {
    bool rotated = false;
    for (int p = 0; p < cols - 1; p++)
      for (int q = p + 1; q < cols; q++) {
        double* wp = &W[(size_t)p * rows];
        double* wq = &W[(size_t)q * rows];
        double alpha = 0.0, beta = 0.0, gamma = 0.0;
        for (int i = 0; i < rows; i++) { alpha += wp[i] * wp[i]; beta += wq[i] * wq[i]; gamma += wp[i] * wq[i]; }
        if (gamma == 0.0 || std::fabs(gamma) <= eps * std::sqrt(alpha * beta)) continue;
        rotated = true;
        const double zeta = (beta - alpha) / (2.0 * gamma);
        const double t = (zeta >= 0.0 ? 1.0 : -1.0) / (std::fabs(zeta) + std::sqrt(1.0 + zeta * zeta));
        const double c = 1.0 / std::sqrt(1.0 + t * t), s = c * t;
        for (int i = 0; i < rows; i++) {
          const double a0 = wp[i], a1 = wq[i];
          wp[i] = c * a0 - s * a1;
          wq[i] = s * a0 + c * a1;
        }
        double* rp = &R[(size_t)p * cols];
        double* rq = &R[(size_t)q * cols];
        for (int i = 0; i < cols; i++) {
          const double a0 = rp[i], a1 = rq[i];
          rp[i] = c * a0 - s * a1;
          rq[i] = s * a0 + c * a1;
        }
      }
    if (!rotated) break;
  }
  std::vector<double> sig2(cols);
  double smax2 = 0.0;
  for (int j = 0; j < cols; j++) {
    double t = 0.0;
    for (int i = 0; i < rows; i++) t += W[(size_t)j * rows + i] * W[(size_t)j * rows + i];
    sig2[j] = t;
    smax2 = std::max(smax2, t);
  }
  const double thr = std::sqrt(smax2) * (double)std::min(m, n) * eps;
  for (int j = 0; j < n; j++) x[j] = 0.0;
  for (int j = 0; j < cols; j++) {
    if (!(std::sqrt(sig2[j]) > thr)) continue;
    if (tall) {          // A = U S R^T:  x += R_j (U_j^T b) / s_j,  U_j s_j = W_j
      double dot = 0.0;
      for (int i = 0; i < m; i++) dot += W[(size_t)j * rows + i] * b[i];
      const double f = dot / sig2[j];
      for (int i = 0; i < n; i++) x[i] += R[(size_t)j * cols + i] * f;
    } else {             // A^T = U S R^T:  x += U_j (R_j^T b) / s_j
      double dot = 0.0;
      for (int i = 0; i < m; i++) dot += R[(size_t)j * cols + i] * b[i];
      const double f = dot / sig2[j];
      for (int i = 0; i < n; i++) x[i] += W[(size_t)j * rows + i] * f;
    }
  }
}

// eigenvalues of a symmetric 3x3 by cyclic Jacobi rotations (SelfAdjointEigenSolver<Matrix3d>, :349)
bool sym3_eigenvalues(const double Hin[9], double ev[3]) {
  double a[9];
  for (int k = 0; k < 9; k++) a[k] = Hin[k];
  for (int sweep = 0; sweep < 50; sweep++) {
    const double off = a[1] * a[1] + a[2] * a[2] + a[5] * a[5];
    const double diag = a[0] * a[0] + a[4] * a[4] + a[8] * a[8];
    if (off <= 1e-32 * diag || off == 0.0) break;
    const int P[3] = {0, 0, 1}, Q[3] = {1, 2, 2};
    for (int r = 0; r < 3; r++) {
      const int p = P[r], q = Q[r];
      const double apq = a[p * 3 + q];
      if (apq == 0.0) continue;
      const double theta = (a[q * 3 + q] - a[p * 3 + p]) / (2.0 * apq);
      const double t = (theta >= 0.0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
      const double c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
      for (int k = 0; k < 3; k++) {                      // A <- A J
        const double akp = a[k * 3 + p], akq = a[k * 3 + q];
        a[k * 3 + p] = c * akp - s * akq;
        a[k * 3 + q] = s * akp + c * akq;
      }
      for (int k = 0; k < 3; k++) {                      // A <- J^T A
        const double apk = a[p * 3 + k], aqk = a[q * 3 + k];
        a[p * 3 + k] = c * apk - s * aqk;
        a[q * 3 + k] = s * apk + c * aqk;
      }
    }
  }
  ev[0] = a[0]; ev[1] = a[4]; ev[2] = a[8];
  std::sort(ev, ev + 3);
  return std::isfinite(ev[0]) && std::isfinite(ev[1]) && std::isfinite(ev[2]);
}

}  // namespace

// The sampling grid -- hence the design matrix A and its pseudo-inverse -- is the same for every
// registration that shares (samples_per_axis, ranges): A^+ (10 x m) is built once by solving the least
// squares problem for the m unit right-hand sides; each registration then costs one 10 x m product.
// (Eigen refactorises per call; the solutions agree to rounding.)
void CovFit::prepare(int samples_per_axis, double xy_half, double yaw_half) {
  n = samples_per_axis;
  m = n * n * n;
  auto lin = [this](double half, int i) {                // linspace(-half, half, n)[i]  (loopclosure.cpp:866-890)
    if (n == 1) return -half;
    const double delta = (half - (-half)) / ((double)n - 1.0);
    return i < n - 1 ? -half + delta * (double)i : half;
  };
  offsets.assign((size_t)m * 3, 0.0);
  std::vector<double> A((size_t)m * 10);
  for (int s = 0; s < m; s++) {                          // theta outer, x, y inner (:294-296)
    const double x = lin(xy_half, (s / n) % n), y = lin(xy_half, s % n), z = lin(yaw_half, s / (n * n));
    offsets[3 * (size_t)s] = x; offsets[3 * (size_t)s + 1] = y; offsets[3 * (size_t)s + 2] = z;
    double* r = A.data() + (size_t)s * 10;               // f = a x^2 + b y^2 + c z^2 + d xy + e yz + f zx + g x + h y + i z + j
    r[0] = x * x; r[1] = y * y; r[2] = z * z; r[3] = x * y; r[4] = y * z; r[5] = z * x;
    r[6] = x; r[7] = y; r[8] = z; r[9] = 1.0;
  }
  pinv.assign((size_t)10 * m, 0.0);
  std::vector<double> e(m, 0.0);
  double q[10];
  for (int s = 0; s < m; s++) {
    e[s] = 1.0;
    lstsq_jacobi(A, e, m, 10, q);
    e[s] = 0.0;
    for (int k = 0; k < 10; k++) pinv[(size_t)k * m + s] = q[k];
  }
}

bool CovFit::solve(const double* costs, double score_scale, double covariance_scaler, double cov36[36]) const {
  double q[10];
  for (int k = 0; k < 10; k++) {
    double t = 0.0;
    const double* row = pinv.data() + (size_t)k * m;
    for (int s = 0; s < m; s++) t += row[s] * costs[s];
    q[k] = t;
  }
  const double H[9] = {2 * q[0], q[3], q[5], q[3], 2 * q[1], q[4], q[5], q[4], 2 * q[2]};      // :340-343
  double ev[3];
  if (!sym3_eigenvalues(H, ev)) return false;                                                  // :351-354
  if (ev[0] <= 0.0 || ev[1] <= 0.0 || ev[2] <= 0.0) return false;                              // :357-360 not convex
  const double det = H[0] * (H[4] * H[8] - H[5] * H[7]) - H[1] * (H[3] * H[8] - H[5] * H[6]) +
                     H[2] * (H[3] * H[7] - H[4] * H[6]);
  double inv[9];
  inv[0] = (H[4] * H[8] - H[5] * H[7]) / det; inv[1] = (H[2] * H[7] - H[1] * H[8]) / det; inv[2] = (H[1] * H[5] - H[2] * H[4]) / det;
  inv[3] = (H[5] * H[6] - H[3] * H[8]) / det; inv[4] = (H[0] * H[8] - H[2] * H[6]) / det; inv[5] = (H[2] * H[3] - H[0] * H[5]) / det;
  inv[6] = (H[3] * H[7] - H[4] * H[6]) / det; inv[7] = (H[1] * H[6] - H[0] * H[7]) / det; inv[8] = (H[0] * H[4] - H[1] * H[3]) / det;
  double c3[9];
  for (int k = 0; k < 9; k++) c3[k] = 2.0 * inv[k] * score_scale * covariance_scaler;          // :365
  for (int k = 0; k < 36; k++) cov36[k] = 0.0;                                                 // :368-374
  for (int k = 0; k < 6; k++) cov36[k * 6 + k] = 1.0;
  cov36[0] = c3[0]; cov36[1] = c3[1]; cov36[6] = c3[3]; cov36[7] = c3[4];
  cov36[35] = c3[8];
  cov36[5] = c3[2]; cov36[11] = c3[5]; cov36[30] = c3[6]; cov36[31] = c3[7];
  return true;
}
