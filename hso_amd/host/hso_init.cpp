// hso_init.cpp — the two-view start (see hso_init.h for what follows the reference and what replaces its OpenCV calls).
#include "hso_init.h"
#include <algorithm>
#include <cmath>
#include <limits>
#include "hso_api.h"

namespace hso {
namespace initialization {

namespace {

// ---------------------------------------------------------------------------------------------- small dense algebra
Matrix3d mat_mul(const Matrix3d& a, const Matrix3d& b)
{
  Matrix3d c{};
  for (int q = 0; q < 9; q++) {
    const int r = q / 3, col = q % 3;
    c.m[r][col] = a.m[r][0] * b.m[0][col] + a.m[r][1] * b.m[1][col] + a.m[r][2] * b.m[2][col];
  }
  return c;
}
Matrix3d mat_t(const Matrix3d& a) { Matrix3d c; for (int q = 0; q < 9; q++) c.m[q / 3][q % 3] = a.m[q % 3][q / 3]; return c; }
void mat_scale(Matrix3d& a, double s) { for (int q = 0; q < 9; q++) a.m[q / 3][q % 3] *= s; }
Vector3d mat_vec(const Matrix3d& a, const Vector3d& v)
{
  return {a.m[0][0] * v[0] + a.m[0][1] * v[1] + a.m[0][2] * v[2], a.m[1][0] * v[0] + a.m[1][1] * v[1] + a.m[1][2] * v[2],
          a.m[2][0] * v[0] + a.m[2][1] * v[1] + a.m[2][2] * v[2]};
}
double mat_det(const Matrix3d& a)
{
  return a.m[0][0] * (a.m[1][1] * a.m[2][2] - a.m[1][2] * a.m[2][1]) - a.m[0][1] * (a.m[1][0] * a.m[2][2] - a.m[1][2] * a.m[2][0]) +
         a.m[0][2] * (a.m[1][0] * a.m[2][1] - a.m[1][1] * a.m[2][0]);
}
Matrix3d mat_identity() { Matrix3d c{}; c.m[0][0] = c.m[1][1] = c.m[2][2] = 1; return c; }
double dot(const Vector3d& a, const Vector3d& b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
Vector3d cross(const Vector3d& a, const Vector3d& b) { return {a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]}; }
Vector3d add(const Vector3d& a, const Vector3d& b) { return {a[0] + b[0], a[1] + b[1], a[2] + b[2]}; }
Vector3d sub(const Vector3d& a, const Vector3d& b) { return {a[0] - b[0], a[1] - b[1], a[2] - b[2]}; }
Vector3d scale(const Vector3d& a, double s) { return {a[0] * s, a[1] * s, a[2] * s}; }
double norm(const Vector3d& a) { return std::sqrt(dot(a, a)); }
Vector2d project2d(const Vector3d& v) { return {v[0] / v[2], v[1] / v[2]}; }      // vikit/math_utils.h:99-102

// rotation matrix <-> the unit quaternion of hso_se3 (x, y, z, w)
Matrix3d quat_to_R(const double q[4])
{
  const double x = q[0], y = q[1], z = q[2], w = q[3];
  Matrix3d R;
  R.m[0][0] = 1 - 2 * (y * y + z * z); R.m[0][1] = 2 * (x * y - z * w);     R.m[0][2] = 2 * (x * z + y * w);
  R.m[1][0] = 2 * (x * y + z * w);     R.m[1][1] = 1 - 2 * (x * x + z * z); R.m[1][2] = 2 * (y * z - x * w);
  R.m[2][0] = 2 * (x * z - y * w);     R.m[2][1] = 2 * (y * z + x * w);     R.m[2][2] = 1 - 2 * (x * x + y * y);
  return R;
}
void R_to_quat(const Matrix3d& R, double q[4])
{
  const double tr = R.m[0][0] + R.m[1][1] + R.m[2][2];
  if (tr > 0) {
    const double s = std::sqrt(tr + 1.0) * 2;
    q[3] = 0.25 * s; q[0] = (R.m[2][1] - R.m[1][2]) / s; q[1] = (R.m[0][2] - R.m[2][0]) / s; q[2] = (R.m[1][0] - R.m[0][1]) / s;
  } else if (R.m[0][0] > R.m[1][1] && R.m[0][0] > R.m[2][2]) {
    const double s = std::sqrt(1.0 + R.m[0][0] - R.m[1][1] - R.m[2][2]) * 2;
    q[3] = (R.m[2][1] - R.m[1][2]) / s; q[0] = 0.25 * s; q[1] = (R.m[0][1] + R.m[1][0]) / s; q[2] = (R.m[0][2] + R.m[2][0]) / s;
  } else if (R.m[1][1] > R.m[2][2]) {
    const double s = std::sqrt(1.0 + R.m[1][1] - R.m[0][0] - R.m[2][2]) * 2;
    q[3] = (R.m[0][2] - R.m[2][0]) / s; q[0] = (R.m[0][1] + R.m[1][0]) / s; q[1] = 0.25 * s; q[2] = (R.m[1][2] + R.m[2][1]) / s;
  } else {
    const double s = std::sqrt(1.0 + R.m[2][2] - R.m[0][0] - R.m[1][1]) * 2;
    q[3] = (R.m[1][0] - R.m[0][1]) / s; q[0] = (R.m[0][2] + R.m[2][0]) / s; q[1] = (R.m[1][2] + R.m[2][1]) / s; q[2] = 0.25 * s;
  }
  const double n = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  for (int i = 0; i < 4; i++) q[i] /= n;
}
SE3 make_se3(const Matrix3d& R, const Vector3d& t)
{
  SE3 T;
  R_to_quat(R, T.v.q);
  T.v.t[0] = t[0]; T.v.t[1] = t[1]; T.v.t[2] = t[2];
  return T;
}
Matrix3d rotation_matrix(const SE3& T) { return quat_to_R(T.v.q); }

// one-sided (Hestenes) Jacobi SVD of a 3x3: A = U diag(s) V^T, s descending, U and V orthogonal (a zero singular value's
// column of U is completed with a cross product)
void svd3(const Matrix3d& A, Matrix3d& U, double s[3], Matrix3d& V)
{
  double u[3][3], v[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) u[i][j] = A.m[i][j];
  for (int sweep = 0; sweep < 60; sweep++) {
    double off = 0;
    for (int p = 0; p < 2; p++)
      for (int q = p + 1; q < 3; q++) {
        double alpha = 0, beta = 0, gamma = 0;
        for (int i = 0; i < 3; i++) { alpha += u[i][p] * u[i][p]; beta += u[i][q] * u[i][q]; gamma += u[i][p] * u[i][q]; }
        if (std::fabs(gamma) <= 1e-300 || std::fabs(gamma) <= 1e-16 * std::sqrt(alpha * beta)) continue;
        off = std::max(off, std::fabs(gamma) / std::sqrt(alpha * beta));
        const double zeta = (beta - alpha) / (2 * gamma);
        const double t = (zeta >= 0 ? 1.0 : -1.0) / (std::fabs(zeta) + std::sqrt(1 + zeta * zeta));
        const double c = 1 / std::sqrt(1 + t * t), sn = c * t;
        for (int i = 0; i < 3; i++) {
          const double a = u[i][p], b = u[i][q];
          u[i][p] = c * a - sn * b; u[i][q] = sn * a + c * b;
          const double va = v[i][p], vb = v[i][q];
          v[i][p] = c * va - sn * vb; v[i][q] = sn * va + c * vb;
        }
      }
    if (off < 1e-15) break;
  }
  double sv[3];
  int order[3] = {0, 1, 2};
  for (int j = 0; j < 3; j++) sv[j] = std::sqrt(u[0][j] * u[0][j] + u[1][j] * u[1][j] + u[2][j] * u[2][j]);
  std::sort(order, order + 3, [&](int a, int b) { return sv[a] > sv[b]; });
  for (int k = 0; k < 3; k++) {
    const int j = order[k];
    s[k] = sv[j];
    for (int i = 0; i < 3; i++) { V.m[i][k] = v[i][j]; U.m[i][k] = sv[j] > 0 ? u[i][j] / sv[j] : 0; }
  }
  if (s[2] <= 1e-12 * s[0]) {               // rank 2: third left vector from the other two
    const Vector3d c = cross({U.m[0][0], U.m[1][0], U.m[2][0]}, {U.m[0][1], U.m[1][1], U.m[2][1]});
    const double n = norm(c);
    for (int i = 0; i < 3; i++) U.m[i][2] = n > 0 ? c[i] / n : 0;
  }
}

// smallest eigenvector of the symmetric 9x9 M (cyclic Jacobi): the null vector of a DLT system A^T A
void smallest_eigvec9(double M[9][9], double out[9])
{
  double V[9][9] = {};
  for (int i = 0; i < 9; i++) V[i][i] = 1;
  for (int sweep = 0; sweep < 60; sweep++) {
    double off = 0, diag = 0;
    for (int i = 0; i < 9; i++) { diag += M[i][i] * M[i][i]; for (int j = i + 1; j < 9; j++) off += M[i][j] * M[i][j]; }
    if (off <= 1e-30 * diag) break;
    for (int p = 0; p < 8; p++)
      for (int q = p + 1; q < 9; q++) {
        if (std::fabs(M[p][q]) < 1e-300) continue;
        const double theta = (M[q][q] - M[p][p]) / (2 * M[p][q]);
        const double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1));
        const double c = 1 / std::sqrt(t * t + 1), s = t * c;
        for (int k = 0; k < 9; k++) { const double a = M[k][p], b = M[k][q]; M[k][p] = c * a - s * b; M[k][q] = s * a + c * b; }
        for (int k = 0; k < 9; k++) { const double a = M[p][k], b = M[q][k]; M[p][k] = c * a - s * b; M[q][k] = s * a + c * b; }
        for (int k = 0; k < 9; k++) { const double a = V[k][p], b = V[k][q]; V[k][p] = c * a - s * b; V[k][q] = s * a + c * b; }
      }
  }
  int best = 0;
  for (int i = 1; i < 9; i++) if (M[i][i] < M[best][best]) best = i;
  for (int i = 0; i < 9; i++) out[i] = V[i][best];
}

// Hartley normalisation: x' = s (x - c), mean distance sqrt(2)
struct Norm2 { double cx, cy, s; };
Norm2 normaliser(const std::vector<Vector2d>& x, const std::vector<int>& idx)
{
  Norm2 n{0, 0, 1};
  for (int i : idx) { n.cx += x[i][0]; n.cy += x[i][1]; }
  n.cx /= idx.size(); n.cy /= idx.size();
  double d = 0;
  for (int i : idx) d += std::hypot(x[i][0] - n.cx, x[i][1] - n.cy);
  d /= idx.size();
  n.s = d > 0 ? std::sqrt(2.0) / d : 1;
  return n;
}
Matrix3d norm_matrix(const Norm2& n) { Matrix3d T{}; T.m[0][0] = n.s; T.m[1][1] = n.s; T.m[0][2] = -n.s * n.cx; T.m[1][2] = -n.s * n.cy; T.m[2][2] = 1; return T; }

// x2^T E x1 = 0 over the points idx (>= 8), the (1, 1, 0) singular values enforced
bool fit_essential(const std::vector<Vector2d>& x1, const std::vector<Vector2d>& x2, const std::vector<int>& idx, Matrix3d& E)
{
  const Norm2 n1 = normaliser(x1, idx), n2 = normaliser(x2, idx);
  double M[9][9] = {};
  for (int i : idx) {
    const double a = (x1[i][0] - n1.cx) * n1.s, b = (x1[i][1] - n1.cy) * n1.s, u = (x2[i][0] - n2.cx) * n2.s, v = (x2[i][1] - n2.cy) * n2.s;
    const double r[9] = {u * a, u * b, u, v * a, v * b, v, a, b, 1};
    for (int p = 0; p < 9; p++) for (int q = 0; q < 9; q++) M[p][q] += r[p] * r[q];
  }
  double e[9];
  smallest_eigvec9(M, e);
  Matrix3d F;
  for (int i = 0; i < 9; i++) F.m[i / 3][i % 3] = e[i];
  F = mat_mul(mat_t(norm_matrix(n2)), mat_mul(F, norm_matrix(n1)));
  Matrix3d U, V; double s[3];
  svd3(F, U, s, V);
  if (!(s[1] > 1e-12 * s[0]) || !std::isfinite(s[0])) return false;
  Matrix3d D{}; D.m[0][0] = D.m[1][1] = 1;
  E = mat_mul(U, mat_mul(D, mat_t(V)));
  return true;
}

double sampson(const Matrix3d& E, const Vector2d& x1, const Vector2d& x2)
{
  const Vector3d a = {x1[0], x1[1], 1}, b = {x2[0], x2[1], 1};
  const Vector3d Ea = mat_vec(E, a), Etb = mat_vec(mat_t(E), b);
  const double e = dot(b, Ea);
  return e * e / (Ea[0] * Ea[0] + Ea[1] * Ea[1] + Etb[0] * Etb[0] + Etb[1] * Etb[1]);
}

struct Lcg {                                  // sampling: a fixed-seed 64-bit LCG (reproducible runs)
  uint64_t s;
  uint32_t next() { s = s * 6364136223846793005ull + 1442695040888963407ull; return (uint32_t)(s >> 33); }
  void sample(int n, int k, int* out)
  {
    for (int i = 0; i < k; i++) {
      for (;;) {
        const int c = (int)(next() % (uint32_t)n);
        bool dup = false;
        for (int j = 0; j < i; j++) dup |= out[j] == c;
        if (!dup) { out[i] = c; break; }
      }
    }
  }
};

int ransac_iterations(double inlier_frac, int sample, int cap)
{
  const double w = std::pow(std::min(std::max(inlier_frac, 1e-6), 1.0 - 1e-9), sample);
  const double n = std::log(1 - 0.99) / std::log(1 - w);
  return (int)std::min((double)cap, std::ceil(std::max(n, 1.0)));
}

// vikit/math_utils.cpp:14-31
Vector3d triangulateFeatureNonLin(const Matrix3d& R, const Vector3d& t, const Vector3d& ray_a, const Vector3d& ray_b)
{
  const Vector3d f2 = mat_vec(R, ray_b);
  const double b0 = dot(t, ray_a), b1 = dot(t, f2);
  const double A00 = dot(ray_a, ray_a), A10 = dot(ray_a, f2), A01 = -A10, A11 = -dot(f2, f2);
  const double det = A00 * A11 - A01 * A10;
  const double l0 = (A11 * b0 - A01 * b1) / det, l1 = (-A10 * b0 + A00 * b1) / det;
  const Vector3d xm = scale(ray_a, l0), xn = add(t, scale(f2, l1));
  return scale(add(xm, xn), 0.5);
}

double reprojError(const Vector3d& f1, const Vector3d& f2, double error_multiplier2)    // vikit/math_utils.cpp:56-63
{
  const Vector2d a = project2d(f1), b = project2d(f2);
  return error_multiplier2 * std::hypot(a[0] - b[0], a[1] - b[1]);
}

// include/hso/point.h:174-184
void jacobian_id2uv(const Vector3d& p_in_f, const Matrix3d& R_th, const Vector3d& t_th, double idH, const Vector3d& fH, double jac[2])
{
  // d(projection) / d(inverse distance along the host bearing): the common denominator once
  const Vector2d uv = project2d(p_in_f);
  const double den = mat_vec(R_th, fH)[2] + t_th[2] * idH;
  for (int r = 0; r < 2; r++) jac[r] = -(t_th[r] - uv[r] * t_th[2]) / den;
}

// src/initialization.cpp:428-474: at most three Gauss-Newton steps on the inverse distance along the reference bearing; a step
// that raises the energy (or is NaN) is taken back
Vector3d distancePointOnce(const Vector3d& pointW, const Vector3d& bearingRef, const Vector3d& bearingCur, const Matrix3d& R_c_r, const Vector3d& t_c_r)
{
  const Vector2d seen = project2d(bearingCur);
  double id_prev = 1. / norm(pointW), id = id_prev, energy_prev = 0;
  for (int iter = 0; iter < 3; ++iter) {
    const Vector3d in_cur = add(mat_vec(R_c_r, scale(bearingRef, 1.0 / id)), t_c_r);
    const Vector2d predicted = project2d(in_cur);
    const double e[2] = {seen[0] - predicted[0], seen[1] - predicted[1]};
    const double energy = e[0] * e[0] + e[1] * e[1];
    double J[2];
    jacobian_id2uv(in_cur, R_c_r, t_c_r, id, bearingRef, J);
    const double gain = 1.0 / (J[0] * J[0] + J[1] * J[1]), pull = -(J[0] * e[0] + J[1] * e[1]);
    const double step = gain * pull;
    if ((iter > 0 && energy > energy_prev) || std::isnan(step)) { id = id_prev; break; }
    id_prev = id; energy_prev = energy;
    id += step;
    if (step <= 0.000001 * id) break;
  }
  return scale(bearingRef, 1.0 / id);
}

template <typename T> T getMedian(std::vector<T>& v)   // vikit/math_utils.h:119-126 (permutes v)
{
  typename std::vector<T>::iterator it = v.begin() + (long)std::floor(v.size() / 2);
  std::nth_element(v.begin(), it, v.end());
  return *it;
}

}  // namespace

// ---------------------------------------------------------------------------------------------- the two models
bool estimateEssential(const std::vector<Vector2d>& x1, const std::vector<Vector2d>& x2, double thresh, Matrix3d& R, Vector3d& t)
{
  const int n = (int)x1.size();
  if (n < 8) return false;
  const double thr2 = thresh * thresh;
  Lcg rng{0x9e3779b97f4a7c15ull};
  std::vector<int> best_in;
  Matrix3d bestE{};
  int budget = 2000;
  for (int it = 0; it < budget; it++) {
    int pick[8];
    rng.sample(n, 8, pick);
    Matrix3d E;
    if (!fit_essential(x1, x2, std::vector<int>(pick, pick + 8), E)) continue;
    std::vector<int> in;
    for (int i = 0; i < n; i++) if (sampson(E, x1[i], x2[i]) < thr2) in.push_back(i);
    if (in.size() > best_in.size()) {
      best_in.swap(in); bestE = E;
      // local optimisation: the eight-point fit of a minimal sample is noise-sensitive, the fit of its consensus set is not
      for (int lo = 0; lo < 3 && best_in.size() >= 12; lo++) {
        Matrix3d E2;
        if (!fit_essential(x1, x2, best_in, E2)) break;
        std::vector<int> in2;
        for (int i = 0; i < n; i++) if (sampson(E2, x1[i], x2[i]) < thr2) in2.push_back(i);
        if (in2.size() <= best_in.size()) break;
        best_in.swap(in2); bestE = E2;
      }
      budget = std::min(budget, std::max(200, ransac_iterations((double)best_in.size() / n, 8, 2000)));
    }
  }
  if (best_in.size() < 8) return false;
  for (int round = 0; round < 2; round++) {                     // refit on the consensus set, re-collect it
    Matrix3d E;
    if (!fit_essential(x1, x2, best_in, E)) break;
    std::vector<int> in;
    for (int i = 0; i < n; i++) if (sampson(E, x1[i], x2[i]) < thr2) in.push_back(i);
    if (in.size() < best_in.size()) break;
    best_in.swap(in); bestE = E;
  }
  // recoverPose: E = U diag(1, 1, 0) V^T -> R = U W V^T or U W^T V^T, t = +-u3; the candidate with most points in front of both
  Matrix3d U, V; double s[3];
  svd3(bestE, U, s, V);
  if (mat_det(U) < 0) mat_scale(U, -1.0);
  if (mat_det(V) < 0) mat_scale(V, -1.0);
  Matrix3d W{}; W.m[0][1] = -1; W.m[1][0] = 1; W.m[2][2] = 1;
  const Matrix3d Rc[2] = {mat_mul(U, mat_mul(W, mat_t(V))), mat_mul(U, mat_mul(mat_t(W), mat_t(V)))};
  const Vector3d u3 = {U.m[0][2], U.m[1][2], U.m[2][2]};
  int best_count = -1;
  for (int c = 0; c < 4; c++) {
    const Matrix3d& Rk = Rc[c >> 1];
    const Vector3d tk = scale(u3, (c & 1) ? -1.0 : 1.0);
    int count = 0;
    for (int i : best_in) {
      const Vector3d f1 = {x1[i][0], x1[i][1], 1}, f2 = {x2[i][0], x2[i][1], 1};
      const Vector3d Xc = triangulateFeatureNonLin(Rk, tk, f2, f1);            // in the second (current) camera
      const Vector3d Xr = mat_vec(mat_t(Rk), sub(Xc, tk));
      if (Xc[2] > 0 && Xr[2] > 0 && Xc[2] < 50 && Xr[2] < 50) count++;        // recoverPose's distance bound of 50 baselines
    }
    if (count > best_count) { best_count = count; R = Rk; t = tk; }
  }
  return best_count > 0;
}

namespace {
bool fit_homography(const std::vector<Vector2d>& x1, const std::vector<Vector2d>& x2, const std::vector<int>& idx, Matrix3d& H)
{
  const Norm2 n1 = normaliser(x1, idx), n2 = normaliser(x2, idx);
  double M[9][9] = {};
  for (int i : idx) {
    const double x = (x1[i][0] - n1.cx) * n1.s, y = (x1[i][1] - n1.cy) * n1.s, u = (x2[i][0] - n2.cx) * n2.s, v = (x2[i][1] - n2.cy) * n2.s;
    const double r0[9] = {-x, -y, -1, 0, 0, 0, u * x, u * y, u}, r1[9] = {0, 0, 0, -x, -y, -1, v * x, v * y, v};
    for (int p = 0; p < 9; p++) for (int q = 0; q < 9; q++) M[p][q] += r0[p] * r0[q] + r1[p] * r1[q];
  }
  double h[9];
  smallest_eigvec9(M, h);
  Matrix3d Hn;
  for (int i = 0; i < 9; i++) Hn.m[i / 3][i % 3] = h[i];
  // H = T2^-1 Hn T1
  Matrix3d T2i{}; T2i.m[0][0] = T2i.m[1][1] = 1 / n2.s; T2i.m[0][2] = n2.cx; T2i.m[1][2] = n2.cy; T2i.m[2][2] = 1;
  H = mat_mul(T2i, mat_mul(Hn, norm_matrix(n1)));
  const double h22 = H.m[2][2];
  if (std::fabs(h22) > 1e-300) for (int q = 0; q < 9; q++) H.m[q / 3][q % 3] /= h22;
  return std::isfinite(H.m[0][0]);
}
double transfer_error2(const Matrix3d& H, const Vector2d& a, const Vector2d& b)
{
  const Vector3d p = mat_vec(H, {a[0], a[1], 1});
  const double du = p[0] / p[2] - b[0], dv = p[1] / p[2] - b[1];
  return du * du + dv * dv;
}
}  // namespace

bool estimateHomography(const std::vector<Vector2d>& x1, const std::vector<Vector2d>& x2, double thresh, Matrix3d& H)
{
  const int n = (int)x1.size();
  if (n < 4) return false;
  const double thr2 = thresh * thresh;
  Lcg rng{0xd1b54a32d192ed03ull};
  std::vector<int> best_in;
  int budget = 2000;
  for (int it = 0; it < budget; it++) {
    int pick[4];
    rng.sample(n, 4, pick);
    Matrix3d Hk;
    if (!fit_homography(x1, x2, std::vector<int>(pick, pick + 4), Hk)) continue;
    std::vector<int> in;
    for (int i = 0; i < n; i++) if (transfer_error2(Hk, x1[i], x2[i]) < thr2) in.push_back(i);
    if (in.size() > best_in.size()) {
      best_in.swap(in); H = Hk;
      for (int lo = 0; lo < 3 && best_in.size() >= 8; lo++) {
        Matrix3d H2;
        if (!fit_homography(x1, x2, best_in, H2)) break;
        std::vector<int> in2;
        for (int i = 0; i < n; i++) if (transfer_error2(H2, x1[i], x2[i]) < thr2) in2.push_back(i);
        if (in2.size() <= best_in.size()) break;
        best_in.swap(in2); H = H2;
      }
      budget = std::min(budget, std::max(200, ransac_iterations((double)best_in.size() / n, 4, 2000)));
    }
  }
  if (best_in.size() < 4) return false;
  for (int round = 0; round < 2; round++) {
    Matrix3d Hk;
    if (!fit_homography(x1, x2, best_in, Hk)) break;
    std::vector<int> in;
    for (int i = 0; i < n; i++) if (transfer_error2(Hk, x1[i], x2[i]) < thr2) in.push_back(i);
    if (in.size() < best_in.size()) break;
    best_in.swap(in); H = Hk;
  }
  return true;
}

// vikit Homography::decompose + computeMatchesInliers + findBestDecomposition (src/vikit/homography.cpp:57-270)
bool decomposeHomography(const Matrix3d& H, const std::vector<Vector2d>& plane_a, const std::vector<Vector2d>& plane_b, double error_multiplier2,
                         double thresh, SE3& T_c2_from_c1)
{
  struct Decomp { Matrix3d R; Vector3d t; double d; Vector3d n; SE3 T; int score; };
  std::vector<Decomp> decompositions;
  Matrix3d U, V; double sv[3];
  svd3(H, U, sv, V);
  const double d1 = std::fabs(sv[0]), d2 = std::fabs(sv[1]), d3 = std::fabs(sv[2]);
  const double s = mat_det(U) * mat_det(V);
  if (!(d1 != d2 && d2 != d3)) return false;                    // nCase != 1: "not implemented or degenerate" (:111-115)
  // Faugeras & Lustman, case of three distinct singular values: the plane normal is (+-a, 0, +-c) in V's frame with
  //   a = sqrt((d1^2 - d2^2) / (d1^2 - d3^2)),  c = sqrt((d2^2 - d3^2) / (d1^2 - d3^2));
  // for d' = +d2 the rotation is about the middle axis by theta (Eq. 13, 14), for d' = -d2 a reflection by phi (Eq. 15, 16)
  const double span = d1 * d1 - d3 * d3;
  const double a = std::sqrt((d1 * d1 - d2 * d2) / span), c = std::sqrt((d2 * d2 - d3 * d3) / span);
  for (int positive = 1; positive >= 0; positive--) {
    for (int k = 0; k < 4; k++) {
      const double sa = (k & 1) ? -1.0 : 1.0, sc = (k & 2) ? -1.0 : 1.0;   // signs of the normal's first and third component
      Decomp dc{};
      dc.d = positive ? s * d2 : s * -d2;
      dc.R = mat_identity();
      if (positive) {
        const double sin_t = (d1 - d3) * a * c * sa * sc / d2, cos_t = (d1 * c * c + d3 * a * a) / d2;
        dc.R.m[0][0] = cos_t; dc.R.m[0][2] = -sin_t; dc.R.m[2][0] = sin_t; dc.R.m[2][2] = cos_t;
        dc.t = {(d1 - d3) * a * sa, 0.0, (d1 - d3) * -c * sc};
      } else {
        const double sin_p = (d1 + d3) * a * c * sa * sc / d2, cos_p = (d3 * a * a - d1 * c * c) / d2;
        dc.R.m[1][1] = -1;
        dc.R.m[0][0] = cos_p; dc.R.m[0][2] = sin_p; dc.R.m[2][0] = sin_p; dc.R.m[2][2] = -cos_p;
        dc.t = {(d1 + d3) * a * sa, 0.0, (d1 + d3) * c * sc};
      }
      dc.n = mat_vec(V, {a * sa, 0.0, c * sc});
      decompositions.push_back(dc);
    }
  }
  for (Decomp& dc : decompositions) {
    Matrix3d R = mat_mul(U, mat_mul(dc.R, mat_t(V)));
    mat_scale(R, s);
    dc.T = make_se3(R, mat_vec(U, dc.t));
  }
  // computeMatchesInliers (:57-71)
  std::vector<char> inliers(plane_a.size());
  for (size_t i = 0; i < plane_a.size(); i++)
    inliers[i] = error_multiplier2 * std::sqrt(transfer_error2(H, plane_a[i], plane_b[i])) < thresh;
  // findBestDecomposition (:196-270)
  for (Decomp& dc : decompositions) {
    int nPositive = 0;
    for (size_t m = 0; m < plane_a.size(); m++) {
      if (!inliers[m]) continue;
      const Vector2d& v2 = plane_a[m];
      if ((H.m[2][0] * v2[0] + H.m[2][1] * v2[1] + H.m[2][2]) / dc.d > 0.0) nPositive++;
    }
    dc.score = -nPositive;
  }
  std::stable_sort(decompositions.begin(), decompositions.end(), [](const Decomp& a, const Decomp& b) { return a.score < b.score; });
  decompositions.resize(4);
  for (Decomp& dc : decompositions) {
    int nPositive = 0;
    for (size_t m = 0; m < plane_a.size(); m++) {
      if (!inliers[m]) continue;
      const Vector3d v3 = {plane_a[m][0], plane_a[m][1], 1};
      if (dot(v3, dc.n) / dc.d > 0.0) nPositive++;
    }
    dc.score = -nPositive;
  }
  std::stable_sort(decompositions.begin(), decompositions.end(), [](const Decomp& a, const Decomp& b) { return a.score < b.score; });
  decompositions.resize(2);
  const double runner_up = (double)decompositions[1].score / (double)decompositions[0].score;
  int keep = 0;
  if (!(runner_up < 0.9)) {                                        // two-way ambiguity: Sampson score over all points
    const double limit = thresh * thresh * 4;
    double score[2];
    for (int i = 0; i < 2; i++) {
      const Matrix3d R = rotation_matrix(decompositions[i].T);
      const Vector3d t = decompositions[i].T.translation();
      Matrix3d sq{}; sq.m[0][1] = -t[2]; sq.m[0][2] = t[1]; sq.m[1][0] = t[2]; sq.m[1][2] = -t[0]; sq.m[2][0] = -t[1]; sq.m[2][1] = t[0];
      const Matrix3d Essential = mat_mul(R, sq);                // as written in the reference (:250)
      double sum = 0;
      for (size_t m = 0; m < plane_a.size(); m++) {
        // sampsonusError(v2Dash = plane_a, Essential, v2 = plane_b), vikit/math_utils.cpp:188-204
        const Vector3d v3Dash = {plane_a[m][0], plane_a[m][1], 1}, v3 = {plane_b[m][0], plane_b[m][1], 1};
        const Vector3d fv3 = mat_vec(Essential, v3), fTv3Dash = mat_vec(mat_t(Essential), v3Dash);
        const double dError = dot(v3Dash, fv3);
        double d = dError * dError / (fv3[0] * fv3[0] + fv3[1] * fv3[1] + fTv3Dash[0] * fTv3Dash[0] + fTv3Dash[1] * fTv3Dash[1]);
        if (d > limit) d = limit;
        sum += d;
      }
      score[i] = sum;
    }
    keep = score[0] <= score[1] ? 0 : 1;
  }
  T_c2_from_c1 = decompositions[keep].T;
  return true;
}

double computeP3D(const std::vector<Vector3d>& rays_first, const std::vector<Vector3d>& rays_second, const Matrix3d& R, const Vector3d& t,
                  double reproj_thresh, double error_multiplier2, std::vector<Vector3d>& vP3D, std::vector<int>& inliers)
{
  inliers.clear(); inliers.reserve(rays_first.size());
  vP3D.clear(); vP3D.reserve(rays_first.size());
  const Matrix3d Rt = mat_t(R);                                 // T_r_c = T_c_r^-1
  double totalEnergy = 0;
  for (size_t i = 0; i < rays_first.size(); ++i) {
    const Vector3d p3d_cur_old = triangulateFeatureNonLin(R, t, rays_first[i], rays_second[i]);
    const Vector3d p3d_ref_old = mat_vec(Rt, sub(p3d_cur_old, t));
    const Vector3d in_first = distancePointOnce(p3d_ref_old, rays_second[i], rays_first[i], R, t);
    const Vector3d in_second = add(mat_vec(R, in_first), t);
    const double e1 = reprojError(rays_first[i], in_second, error_multiplier2);
    totalEnergy += e1;
    vP3D.push_back(in_second);
    if (in_first[2] < 0.01 || in_second[2] < 0.01) continue;
    const float ratio = (float)(norm(p3d_ref_old) / norm(in_first));
    if (ratio < 0.9 || ratio > 1.1) continue;
    if (e1 < reproj_thresh) inliers.push_back((int)i);
  }
  return totalEnergy;
}

void computeInitializeMatrix(const std::vector<Vector3d>& f_ref, const std::vector<Vector3d>& f_cur, double focal_length,
                             double reprojection_threshold, std::vector<int>& inliers, std::vector<Vector3d>& xyz_in_cur, SE3& T_cur_from_ref,
                             int* used_homography)
{
  std::vector<Vector2d> x1(f_ref.size()), x2(f_cur.size());
  for (size_t i = 0; i < f_ref.size(); ++i) {
    // cv::Point2f in the reference (:309-313): the model estimators see float-precision coordinates
    auto as_point2f = [](const Vector3d& b) { const Vector2d n = project2d(b); return Vector2d{(double)(float)n[0], (double)(float)n[1]}; };
    x1[i] = as_point2f(f_ref[i]);
    x2[i] = as_point2f(f_cur[i]);
  }
  const double inf = std::numeric_limits<double>::infinity();
  Matrix3d R{}; Vector3d t{};
  std::vector<int> inliers_E, inliers_H;
  std::vector<Vector3d> xyz_E, xyz_H;
  double E_error = inf, H_error = inf;
  SE3 T_E, T_H;
  if (estimateEssential(x1, x2, 2.0 / focal_length, R, t)) {
    E_error = computeP3D(f_cur, f_ref, R, t, reprojection_threshold, focal_length, xyz_E, inliers_E);
    T_E = make_se3(R, t);
    if (std::isnan(E_error)) E_error = inf;
  }
  std::vector<Vector2d> uv_ref(f_ref.size()), uv_cur(f_cur.size());
  std::transform(f_ref.begin(), f_ref.end(), uv_ref.begin(), project2d);
  std::transform(f_cur.begin(), f_cur.end(), uv_cur.begin(), project2d);
  Matrix3d H;
  if (estimateHomography(x1, x2, 2.0 / focal_length, H) && decomposeHomography(H, uv_ref, uv_cur, focal_length, reprojection_threshold, T_H)) {
    H_error = computeP3D(f_cur, f_ref, rotation_matrix(T_H), T_H.translation(), reprojection_threshold, focal_length, xyz_H, inliers_H);
    if (std::isnan(H_error)) H_error = inf;
  }
  if (H_error < E_error) {                                      // :366-383
    inliers = inliers_H; xyz_in_cur = xyz_H; T_cur_from_ref = T_H;
    if (used_homography) *used_homography = 1;
  } else {
    inliers = inliers_E; xyz_in_cur = xyz_E; T_cur_from_ref = T_E;
    if (used_homography) *used_homography = 0;
  }
}

}  // namespace initialization
}  // namespace hso
