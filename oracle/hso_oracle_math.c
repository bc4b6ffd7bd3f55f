/*
 * hso_oracle_math.c — SE3/SO3, small dense solves, median.  TEST INFRASTRUCTURE
 * (see hso_oracle.h).  Follows thirdparty/Sophus/sophus/{se3,so3}.cpp of the
 * reference; Eigen (absent from the reference tree, version unpinned:
 * README.md:24-27) is restated from its published algorithms:
 * Quaternion product / normalize / _transformVector / toRotationMatrix
 * (Eigen/src/Geometry/Quaternion.h) and LDLT (Eigen/src/Cholesky/LDLT.h).
 */
#include "hso_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define SMALL_EPS 1e-10 /* thirdparty/Sophus/sophus/so3.h:35 */

static void quat_mul(const double a[4], const double b[4], double o[4])
{
  /* Eigen quat_product (x,y,z,w storage) */
  const double ax = a[0], ay = a[1], az = a[2], aw = a[3];
  const double bx = b[0], by = b[1], bz = b[2], bw = b[3];
  o[3] = aw * bw - ax * bx - ay * by - az * bz;
  o[0] = aw * bx + ax * bw + ay * bz - az * by;
  o[1] = aw * by + ay * bw + az * bx - ax * bz;
  o[2] = aw * bz + az * bw + ax * by - ay * bx;
}

static void quat_normalize(double q[4])
{
  const double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  q[0] /= n; q[1] /= n; q[2] /= n; q[3] /= n;
}

static void quat_rotate(const double q[4], const double v[3], double o[3])
{
  /* Eigen QuaternionBase::_transformVector */
  double uv[3];
  uv[0] = q[1] * v[2] - q[2] * v[1];
  uv[1] = q[2] * v[0] - q[0] * v[2];
  uv[2] = q[0] * v[1] - q[1] * v[0];
  uv[0] += uv[0]; uv[1] += uv[1]; uv[2] += uv[2];
  const double c0 = q[1] * uv[2] - q[2] * uv[1];
  const double c1 = q[2] * uv[0] - q[0] * uv[2];
  const double c2 = q[0] * uv[1] - q[1] * uv[0];
  o[0] = (v[0] + q[3] * uv[0]) + c0;
  o[1] = (v[1] + q[3] * uv[1]) + c1;
  o[2] = (v[2] + q[3] * uv[2]) + c2;
}

void hso_or_so3_matrix(const double q[4], double R[9])
{
  /* Eigen QuaternionBase::toRotationMatrix */
  const double tx = 2 * q[0], ty = 2 * q[1], tz = 2 * q[2];
  const double twx = tx * q[3], twy = ty * q[3], twz = tz * q[3];
  const double txx = tx * q[0], txy = ty * q[0], txz = tz * q[0];
  const double tyy = ty * q[1], tyz = tz * q[1], tzz = tz * q[2];
  R[0] = 1 - (tyy + tzz); R[1] = txy - twz;       R[2] = txz + twy;
  R[3] = txy + twz;       R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
  R[6] = txz - twy;       R[7] = tyz + twx;       R[8] = 1 - (txx + tyy);
}

void hso_or_se3_identity(hso_se3* T)
{
  T->q[0] = T->q[1] = T->q[2] = 0; T->q[3] = 1;
  T->t[0] = T->t[1] = T->t[2] = 0;
}

void hso_or_se3_apply(const hso_se3* T, const double p[3], double out[3])
{
  /* se3.cpp:91-95: so3_*xyz + translation_ */
  double r[3];
  quat_rotate(T->q, p, r);
  out[0] = r[0] + T->t[0]; out[1] = r[1] + T->t[1]; out[2] = r[2] + T->t[2];
}

void hso_or_se3_mul(const hso_se3* a, const hso_se3* b, hso_se3* out)
{
  /* se3.cpp:59-66; so3.cpp:64-71 (product then normalize) */
  hso_se3 r;
  double rt[3];
  quat_rotate(a->q, b->t, rt);
  r.t[0] = a->t[0] + rt[0]; r.t[1] = a->t[1] + rt[1]; r.t[2] = a->t[2] + rt[2];
  quat_mul(a->q, b->q, r.q);
  quat_normalize(r.q);
  *out = r;
}

void hso_or_se3_inverse(const hso_se3* a, hso_se3* out)
{
  /* se3.cpp:76-83; SO3(Quaterniond) normalizes, so3.cpp:43-47 */
  hso_se3 r;
  r.q[0] = -a->q[0]; r.q[1] = -a->q[1]; r.q[2] = -a->q[2]; r.q[3] = a->q[3];
  quat_normalize(r.q);
  double nt[3] = { a->t[0] * -1., a->t[1] * -1., a->t[2] * -1. };
  quat_rotate(r.q, nt, r.t);
  *out = r;
}

void hso_or_se3_exp(const double u[6], hso_se3* out)
{
  /* se3.cpp:170-196, so3.cpp:178-202 */
  const double* upsilon = u;
  const double* omega = u + 3;
  const double theta = sqrt(omega[0] * omega[0] + omega[1] * omega[1] + omega[2] * omega[2]);
  const double half_theta = 0.5 * theta;
  double imag_factor;
  const double real_factor = cos(half_theta);
  if (theta < SMALL_EPS) {
    const double theta_sq = theta * theta;
    const double theta_po4 = theta_sq * theta_sq;
    imag_factor = 0.5 - 0.0208333 * theta_sq + 0.000260417 * theta_po4;
  } else {
    const double sin_half_theta = sin(half_theta);
    imag_factor = sin_half_theta / theta;
  }
  hso_se3 r;
  r.q[3] = real_factor;
  r.q[0] = imag_factor * omega[0];
  r.q[1] = imag_factor * omega[1];
  r.q[2] = imag_factor * omega[2];
  quat_normalize(r.q);

  /* Omega = hat(omega), Omega_sq = Omega*Omega */
  const double O[9] = { 0, -omega[2], omega[1], omega[2], 0, -omega[0], -omega[1], omega[0], 0 };
  double O2[9];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      double s = O[i * 3 + 0] * O[0 * 3 + j];
      s += O[i * 3 + 1] * O[1 * 3 + j];
      s += O[i * 3 + 2] * O[2 * 3 + j];
      O2[i * 3 + j] = s;
    }
  double V[9];
  if (theta < SMALL_EPS) {
    hso_or_so3_matrix(r.q, V);
  } else {
    const double theta_sq = theta * theta;
    const double c1 = (1 - cos(theta)) / (theta_sq);
    const double c2 = (theta - sin(theta)) / (theta_sq * theta);
    for (int i = 0; i < 9; i++) {
      const double id = (i == 0 || i == 4 || i == 8) ? 1.0 : 0.0;
      V[i] = (id + c1 * O[i]) + c2 * O2[i];
    }
  }
  for (int i = 0; i < 3; i++) {
    double s = V[i * 3 + 0] * upsilon[0];
    s += V[i * 3 + 1] * upsilon[1];
    s += V[i * 3 + 2] * upsilon[2];
    r.t[i] = s;
  }
  *out = r;
}

void hso_or_se3_log(const hso_se3* T, double out[6])
{
  /* se3.cpp:198-220, so3.cpp:127-176 */
  const double* q = T->q;
  const double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2]);
  const double w = q[3];
  const double squared_w = w * w;
  double two_atan_nbyw_by_n;
  if (n < SMALL_EPS) {
    two_atan_nbyw_by_n = 2. / w - 2. * (n * n) / (w * squared_w);
  } else {
    /* the |w| < eps branch of so3.cpp:160-170 is overwritten by :171 */
    two_atan_nbyw_by_n = 2 * atan(n / w) / n;
  }
  const double theta = two_atan_nbyw_by_n * n;
  double om[3] = { two_atan_nbyw_by_n * q[0], two_atan_nbyw_by_n * q[1], two_atan_nbyw_by_n * q[2] };
  const double O[9] = { 0, -om[2], om[1], om[2], 0, -om[0], -om[1], om[0], 0 };
  double O2[9];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      double s = O[i * 3 + 0] * O[0 * 3 + j];
      s += O[i * 3 + 1] * O[1 * 3 + j];
      s += O[i * 3 + 2] * O[2 * 3 + j];
      O2[i * 3 + j] = s;
    }
  double Vinv[9];
  const double c = (theta < SMALL_EPS) ? (1. / 12.) : (1 - theta / (2 * tan(theta / 2))) / (theta * theta);
  for (int i = 0; i < 9; i++) {
    const double id = (i == 0 || i == 4 || i == 8) ? 1.0 : 0.0;
    Vinv[i] = (id - 0.5 * O[i]) + c * O2[i];
  }
  for (int i = 0; i < 3; i++) {
    double s = Vinv[i * 3 + 0] * T->t[0];
    s += Vinv[i * 3 + 1] * T->t[1];
    s += Vinv[i * 3 + 2] * T->t[2];
    out[i] = s;
  }
  out[3] = om[0]; out[4] = om[1]; out[5] = om[2];
}

/* Eigen::LDLT (lower, in-place, unblocked, diagonal pivoting) + solve.
 * Eigen/src/Cholesky/LDLT.h: ldlt_inplace<Lower>::unblocked and
 * LDLT::_solve_impl.  Used by CoarseTracker.cpp:114 (7x7) and
 * pose_optimizer.cpp:595 (6x6). */
void hso_or_ldlt_solve(const double* Ain, const double* b, int n, double* x)
{
  double m[8 * 8];
  int tr[8];
  double temp[8];
  for (int i = 0; i < n; i++)
    for (int j = 0; j < n; j++) m[i * 8 + j] = Ain[i * n + j];
#define M(i, j) m[(i) * 8 + (j)]
  int all_zero = 0;
  for (int k = 0; k < n; k++) {
    int idx = k;
    double biggest = fabs(M(k, k));
    for (int i = k + 1; i < n; i++)
      if (fabs(M(i, i)) > biggest) { biggest = fabs(M(i, i)); idx = i; }
    tr[k] = idx;
    if (k != idx) {
      const int s = n - idx - 1;
      for (int j = 0; j < k; j++) { double t = M(k, j); M(k, j) = M(idx, j); M(idx, j) = t; }
      for (int i = 0; i < s; i++) {
        double t = M(idx + 1 + i, k); M(idx + 1 + i, k) = M(idx + 1 + i, idx); M(idx + 1 + i, idx) = t;
      }
      { double t = M(k, k); M(k, k) = M(idx, idx); M(idx, idx) = t; }
      for (int i = k + 1; i < idx; i++) { double t = M(i, k); M(i, k) = M(idx, i); M(idx, i) = t; }
    }
    const int rs = n - k - 1;
    if (k > 0) {
      for (int j = 0; j < k; j++) temp[j] = M(j, j) * M(k, j);
      double s = 0;
      for (int j = 0; j < k; j++) s += M(k, j) * temp[j];
      M(k, k) -= s;
      for (int i = 0; i < rs; i++) {
        double a = 0;
        for (int j = 0; j < k; j++) a += M(k + 1 + i, j) * temp[j];
        M(k + 1 + i, k) -= a;
      }
    }
    const double realAkk = M(k, k);
    const int pivot_is_valid = fabs(realAkk) > 0;
    if (k == 0 && !pivot_is_valid) {
      for (int j = 0; j < n; j++) tr[j] = j;
      all_zero = 1;
      break;
    }
    if (rs > 0 && pivot_is_valid)
      for (int i = 0; i < rs; i++) M(k + 1 + i, k) /= realAkk;
  }
  (void)all_zero;
  double d[8];
  for (int i = 0; i < n; i++) d[i] = b[i];
  /* dst = P b */
  for (int k = 0; k < n; k++) if (tr[k] != k) { double t = d[k]; d[k] = d[tr[k]]; d[tr[k]] = t; }
  /* dst = L^-1 dst (unit lower) */
  for (int i = 0; i < n; i++) {
    double s = d[i];
    for (int j = 0; j < i; j++) s -= M(i, j) * d[j];
    d[i] = s;
  }
  /* dst = D^-1 dst, pseudo-inverse with tolerance 1/highest (Eigen 3.3) */
  const double tolerance = 1.0 / 1.7976931348623157e308;
  for (int i = 0; i < n; i++) {
    if (fabs(M(i, i)) > tolerance) d[i] /= M(i, i);
    else d[i] = 0;
  }
  /* dst = L^-T dst */
  for (int i = n - 1; i >= 0; i--) {
    double s = d[i];
    for (int j = i + 1; j < n; j++) s -= M(j, i) * d[j];
    d[i] = s;
  }
  /* dst = P^T dst */
  for (int k = n - 1; k >= 0; k--) if (tr[k] != k) { double t = d[k]; d[k] = d[tr[k]]; d[tr[k]] = t; }
  for (int i = 0; i < n; i++) x[i] = d[i];
#undef M
}

/* getMedian: nth_element at floor(n/2) (include/hso/vikit/math_utils.h:119-126).
 * The selected VALUE is what matters (the permutation nth_element leaves is not
 * observable by the callers), so a quickselect is an exact restatement. */
#define DEFINE_SELECT(NAME, TYPE)                                           \
  static TYPE NAME(TYPE* a, int n, int k)                                   \
  {                                                                         \
    int lo = 0, hi = n - 1;                                                 \
    while (lo < hi) {                                                       \
      const TYPE pivot = a[lo + (hi - lo) / 2];                             \
      int i = lo, j = hi;                                                   \
      while (i <= j) {                                                      \
        while (a[i] < pivot) i++;                                           \
        while (a[j] > pivot) j--;                                           \
        if (i <= j) { TYPE t = a[i]; a[i] = a[j]; a[j] = t; i++; j--; }     \
      }                                                                     \
      if (k <= j) hi = j;                                                   \
      else if (k >= i) lo = i;                                              \
      else break;                                                           \
    }                                                                       \
    return a[k];                                                            \
  }
DEFINE_SELECT(select_f, float)
DEFINE_SELECT(select_d, double)

float hso_or_median_f(float* data, int n) { return select_f(data, n, n / 2); }
double hso_or_median_d(double* data, int n) { return select_d(data, n, n / 2); }

/* ---- g2o::SE3Quat (thirdparty/g2o/g2o/types/se3quat.h), the pose type of the local BA ---- */

static void g2o_normalize_rotation(double q[4])
{
  /* se3quat.h:280-285 */
  if (q[3] < 0) { q[0] *= -1; q[1] *= -1; q[2] *= -1; q[3] *= -1; }
  quat_normalize(q);
}

void hso_or_se3quat_mul(const hso_se3* a, const hso_se3* b, hso_se3* out)
{
  /* se3quat.h:104-110: result._t += _r*tr2._t; result._r *= tr2._r; normalizeRotation() */
  hso_se3 r;
  double rt[3];
  quat_rotate(a->q, b->t, rt);
  r.t[0] = a->t[0] + rt[0]; r.t[1] = a->t[1] + rt[1]; r.t[2] = a->t[2] + rt[2];
  quat_mul(a->q, b->q, r.q);
  g2o_normalize_rotation(r.q);
  *out = r;
}

static void mat3_to_quat(const double m[9], double q[4])
{
  /* Eigen::Quaternion(Matrix3) — quaternionbase_assign_impl<Other,3,3>, "Quaternion Calculus and Fast Animation" */
  double t = m[0] + m[4] + m[8];
  if (t > 0) {
    t = sqrt(t + 1.0);
    q[3] = 0.5 * t;
    t = 0.5 / t;
    q[0] = (m[7] - m[5]) * t;
    q[1] = (m[2] - m[6]) * t;
    q[2] = (m[3] - m[1]) * t;
  } else {
    int i = 0;
    if (m[4] > m[0]) i = 1;
    if (m[8] > m[i * 3 + i]) i = 2;
    const int j = (i + 1) % 3, k = (j + 1) % 3;
    t = sqrt(m[i * 3 + i] - m[j * 3 + j] - m[k * 3 + k] + 1.0);
    q[i] = 0.5 * t;
    t = 0.5 / t;
    q[3] = (m[k * 3 + j] - m[j * 3 + k]) * t;
    q[j] = (m[j * 3 + i] + m[i * 3 + j]) * t;
    q[k] = (m[k * 3 + i] + m[i * 3 + k]) * t;
  }
}

void hso_or_se3quat_exp(const double update[6], hso_se3* out)
{
  /* se3quat.h:223-257: update = [omega, upsilon] */
  const double* omega = update;
  const double* upsilon = update + 3;
  const double theta = sqrt(omega[0] * omega[0] + omega[1] * omega[1] + omega[2] * omega[2]);
  const double O[9] = { 0, -omega[2], omega[1], omega[2], 0, -omega[0], -omega[1], omega[0], 0 };
  double O2[9], R[9], V[9];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      double s = O[i * 3 + 0] * O[0 * 3 + j];
      s += O[i * 3 + 1] * O[1 * 3 + j];
      s += O[i * 3 + 2] * O[2 * 3 + j];
      O2[i * 3 + j] = s;
    }
  if (theta < 0.00001) {
    /* R = I + Omega + Omega*Omega ("TODO: CHECK WHETHER THIS IS CORRECT" in the reference); V = R */
    for (int i = 0; i < 9; i++) {
      const double id = (i == 0 || i == 4 || i == 8) ? 1.0 : 0.0;
      R[i] = (id + O[i]) + O2[i];
      V[i] = R[i];
    }
  } else {
    const double a = sin(theta) / theta;
    const double b = (1 - cos(theta)) / (theta * theta);
    const double c = (theta - sin(theta)) / (pow(theta, 3));
    for (int i = 0; i < 9; i++) {
      const double id = (i == 0 || i == 4 || i == 8) ? 1.0 : 0.0;
      R[i] = (id + a * O[i]) + b * O2[i];
      V[i] = (id + b * O[i]) + c * O2[i];
    }
  }
  hso_se3 r;
  mat3_to_quat(R, r.q);
  for (int i = 0; i < 3; i++) r.t[i] = (V[i * 3 + 0] * upsilon[0] + V[i * 3 + 1] * upsilon[1]) + V[i * 3 + 2] * upsilon[2];
  g2o_normalize_rotation(r.q);   /* SE3Quat(const Quaterniond&, const Vector3d&), se3quat.h:62-64 */
  *out = r;
}

/* ---- decision margins (see hso_oracle.h) ---- */
static __thread double g_margins[HSO_M_COUNT];   /* per thread: the batch forms (hso_oracle_batch.c) run seeds on worker threads */
void hso_or_margins_reset(void) { for (int i = 0; i < HSO_M_COUNT; i++) g_margins[i] = 1e300; }
void hso_or_margins_get(hso_or_margins* out) { memcpy(out, g_margins, sizeof(g_margins)); }
void hso_or_margin_note(int field, double v) { v = fabs(v); if (!(v >= g_margins[field])) g_margins[field] = v; }
