// hso_dev_math.h — SE3 / camera / small dense algebra for gfx950 device code.
//
// Written for the CDNA4 kernels of this library; the formulas are the ones the
// reference uses through Sophus (thirdparty/Sophus/sophus/{se3,so3}.cpp) and
// Eigen (Quaternion product / _transformVector / toRotationMatrix, LDLT), in
// the same operation order, so that with -ffp-contract=off per-feature
// projections are bit-identical to the CPU restatement the parity tests use.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/hso_gpu.h"

#define HSO_DEV __device__ __forceinline__
#define HSO_HD __host__ __device__ __forceinline__

namespace hso_dev {

// ---- lane exchanges at a fixed xor distance, cheapest gfx950 form per distance (no LDS crossbar trip except none at all):
//   32, 16  v_permlane32_swap / v_permlane16_swap (swap the upper half / odd rows of one register with the lower half /
//           even rows of another: called with the same value twice, the second result is the partner's value in the
//           lower lanes and the first result is it in the upper lanes);
//   8       DPP row_ror:8;   2, 1  DPP quad_perm;   4  two DPP moves (quad_perm [3,2,1,0] then row_half_mirror: 3 ^ 7 = 4).
// The value every lane receives is exactly what __shfl_xor(v, M) returns, so butterfly sums keep their bits.
typedef unsigned lane_u32x2 __attribute__((ext_vector_type(2)));
template <int M> HSO_DEV unsigned lane_xor_u32(unsigned v)
{
  if constexpr (M == 32 || M == 16) {
    lane_u32x2 r;
    if constexpr (M == 32) r = __builtin_amdgcn_permlane32_swap(v, v, false, false);
    else r = __builtin_amdgcn_permlane16_swap(v, v, false, false);
    const bool up = (__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)) & M) != 0;
    return up ? r[0] : r[1];
  } else if constexpr (M == 8) {
    return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x128, 0xf, 0xf, false);       // row_ror:8
  } else if constexpr (M == 4) {
    const int t = __builtin_amdgcn_update_dpp(0, (int)v, 0x1b, 0xf, 0xf, false);           // quad_perm:[3,2,1,0]  (lane ^ 3)
    return (unsigned)__builtin_amdgcn_update_dpp(0, t, 0x141, 0xf, 0xf, false);            // row_half_mirror      (lane ^ 7)
  } else if constexpr (M == 2) {
    return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x4e, 0xf, 0xf, false);        // quad_perm:[2,3,0,1]
  } else {
    static_assert(M == 1, "xor distance");
    return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0xb1, 0xf, 0xf, false);        // quad_perm:[1,0,3,2]
  }
}
template <int M> HSO_DEV float lane_xor(float v) { return __uint_as_float(lane_xor_u32<M>(__float_as_uint(v))); }
template <int M> HSO_DEV int lane_xor(int v) { return (int)lane_xor_u32<M>((unsigned)v); }
template <int M> HSO_DEV double lane_xor(double v)
{
  const unsigned long long b = (unsigned long long)__double_as_longlong(v);
  const unsigned lo = lane_xor_u32<M>((unsigned)b), hi = lane_xor_u32<M>((unsigned)(b >> 32));
  return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
// v + partner at distance 32 or 16 straight from the swap: its two results are, in every lane, the lane's own value and
// its partner's (in one order or the other; the sum does not care), so no select is needed
template <int M> HSO_DEV float lane_swap_sum(float v)
{
  lane_u32x2 r;
  if constexpr (M == 32) r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  else r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
template <int M> HSO_DEV int lane_swap_sum(int v)
{
  lane_u32x2 r;
  if constexpr (M == 32) r = __builtin_amdgcn_permlane32_swap((unsigned)v, (unsigned)v, false, false);
  else r = __builtin_amdgcn_permlane16_swap((unsigned)v, (unsigned)v, false, false);
  return (int)(r[0] + r[1]);
}
template <int M> HSO_DEV double lane_swap_sum(double v)
{
  const unsigned long long b = (unsigned long long)__double_as_longlong(v);
  lane_u32x2 r0, r1;
  if constexpr (M == 32) {
    r0 = __builtin_amdgcn_permlane32_swap((unsigned)b, (unsigned)b, false, false);
    r1 = __builtin_amdgcn_permlane32_swap((unsigned)(b >> 32), (unsigned)(b >> 32), false, false);
  } else {
    r0 = __builtin_amdgcn_permlane16_swap((unsigned)b, (unsigned)b, false, false);
    r1 = __builtin_amdgcn_permlane16_swap((unsigned)(b >> 32), (unsigned)(b >> 32), false, false);
  }
  return __longlong_as_double((long long)(((unsigned long long)r1[0] << 32) | r0[0])) +
         __longlong_as_double((long long)(((unsigned long long)r1[1] << 32) | r0[1]));
}
// v + partner at distances 32, 16, 8, 4, 2, 1: every lane ends with the wave total, summed in the order of the
// __shfl_xor butterfly it replaces
template <typename T> HSO_DEV T wave_butterfly_sum(T v)
{
  v = lane_swap_sum<32>(v); v = lane_swap_sum<16>(v); v += lane_xor<8>(v);
  v += lane_xor<4>(v); v += lane_xor<2>(v); v += lane_xor<1>(v);
  return v;
}

struct Se3 {
  double qx, qy, qz, qw;
  double tx, ty, tz;
};

HSO_HD Se3 se3_from(const hso_se3& s)
{
  Se3 r; r.qx = s.q[0]; r.qy = s.q[1]; r.qz = s.q[2]; r.qw = s.q[3];
  r.tx = s.t[0]; r.ty = s.t[1]; r.tz = s.t[2];
  return r;
}
HSO_HD void se3_to(const Se3& r, hso_se3& s)
{
  s.q[0] = r.qx; s.q[1] = r.qy; s.q[2] = r.qz; s.q[3] = r.qw;
  s.t[0] = r.tx; s.t[1] = r.ty; s.t[2] = r.tz;
}

// Eigen QuaternionBase::_transformVector
HSO_HD void quat_rotate(double qx, double qy, double qz, double qw,
                        double vx, double vy, double vz, double& ox, double& oy, double& oz)
{
  double ux = qy * vz - qz * vy;
  double uy = qz * vx - qx * vz;
  double uz = qx * vy - qy * vx;
  ux += ux; uy += uy; uz += uz;
  const double c0 = qy * uz - qz * uy;
  const double c1 = qz * ux - qx * uz;
  const double c2 = qx * uy - qy * ux;
  ox = (vx + qw * ux) + c0;
  oy = (vy + qw * uy) + c1;
  oz = (vz + qw * uz) + c2;
}

// SE3::operator*(Vector3d), se3.cpp:91-95
HSO_HD void se3_apply(const Se3& T, double vx, double vy, double vz, double& ox, double& oy, double& oz)
{
  double rx, ry, rz;
  quat_rotate(T.qx, T.qy, T.qz, T.qw, vx, vy, vz, rx, ry, rz);
  ox = rx + T.tx; oy = ry + T.ty; oz = rz + T.tz;
}

// Eigen normalize() divides the four coefficients by the norm; one reciprocal and four
// products differ from that by at most an ulp per coefficient and keep the serial
// single-lane LM step short
HSO_HD void quat_normalize(Se3& r)
{
#ifdef __HIP_DEVICE_COMPILE__
  const double inv = rsqrt(r.qx * r.qx + r.qy * r.qy + r.qz * r.qz + r.qw * r.qw);
#else
  const double inv = 1.0 / sqrt(r.qx * r.qx + r.qy * r.qy + r.qz * r.qz + r.qw * r.qw);
#endif
  r.qx *= inv; r.qy *= inv; r.qz *= inv; r.qw *= inv;
}

// SE3::operator*(SE3), se3.cpp:59-66 (+ SO3 product normalises, so3.cpp:64-71)
HSO_HD Se3 se3_mul(const Se3& a, const Se3& b)
{
  Se3 r;
  double rx, ry, rz;
  quat_rotate(a.qx, a.qy, a.qz, a.qw, b.tx, b.ty, b.tz, rx, ry, rz);
  r.tx = a.tx + rx; r.ty = a.ty + ry; r.tz = a.tz + rz;
  r.qw = a.qw * b.qw - a.qx * b.qx - a.qy * b.qy - a.qz * b.qz;
  r.qx = a.qw * b.qx + a.qx * b.qw + a.qy * b.qz - a.qz * b.qy;
  r.qy = a.qw * b.qy + a.qy * b.qw + a.qz * b.qx - a.qx * b.qz;
  r.qz = a.qw * b.qz + a.qz * b.qw + a.qx * b.qy - a.qy * b.qx;
  quat_normalize(r);
  return r;
}

// SE3::inverse, se3.cpp:76-83
HSO_HD Se3 se3_inverse(const Se3& a)
{
  Se3 r;
  r.qx = -a.qx; r.qy = -a.qy; r.qz = -a.qz; r.qw = a.qw;
  quat_normalize(r);
  quat_rotate(r.qx, r.qy, r.qz, r.qw, a.tx * -1., a.ty * -1., a.tz * -1., r.tx, r.ty, r.tz);
  return r;
}

HSO_HD void so3_matrix(const Se3& q, double R[9])
{
  const double tx = 2 * q.qx, ty = 2 * q.qy, tz = 2 * q.qz;
  const double twx = tx * q.qw, twy = ty * q.qw, twz = tz * q.qw;
  const double txx = tx * q.qx, txy = ty * q.qx, txz = tz * q.qx;
  const double tyy = ty * q.qy, tyz = tz * q.qy, tzz = tz * q.qz;
  R[0] = 1 - (tyy + tzz); R[1] = txy - twz;       R[2] = txz + twy;
  R[3] = txy + twz;       R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
  R[6] = txz - twy;       R[7] = tyz + twx;       R[8] = 1 - (txx + tyy);
}

// SE3::exp, se3.cpp:170-196; SO3::expAndTheta, so3.cpp:178-202
HSO_HD Se3 se3_exp(const double u[6])
{
  const double SMALL_EPS = 1e-10;
  const double o0 = u[3], o1 = u[4], o2 = u[5];
  const double theta = sqrt(o0 * o0 + o1 * o1 + o2 * o2);
  const double half_theta = 0.5 * theta;
  double imag_factor;
  // one sincos of theta/2 serves both the quaternion and (through the double-angle
  // identities) the V matrix below, instead of the reference's four separate calls
  double sh, ch;
  if (half_theta < 0.25) {
    // LM steps are small rotations: Taylor series to x^13 / x^14 (truncation < 1e-21 for x <= 0.25)
    // instead of the library's range reduction — this runs on one lane while a workgroup waits
    const double x2 = half_theta * half_theta;
    sh = half_theta * (1.0 + x2 * (-1.0 / 6 + x2 * (1.0 / 120 + x2 * (-1.0 / 5040 + x2 * (1.0 / 362880 + x2 * (-1.0 / 39916800 + x2 * (1.0 / 6227020800.0)))))));
    ch = 1.0 + x2 * (-0.5 + x2 * (1.0 / 24 + x2 * (-1.0 / 720 + x2 * (1.0 / 40320 + x2 * (-1.0 / 3628800 + x2 * (1.0 / 479001600.0 + x2 * (-1.0 / 87178291200.0)))))));
  } else {
    sincos(half_theta, &sh, &ch);
  }
  const double real_factor = ch;
  double inv_theta = 0;
  if (theta < SMALL_EPS) {
    const double theta_sq = theta * theta;
    const double theta_po4 = theta_sq * theta_sq;
    imag_factor = 0.5 - 0.0208333 * theta_sq + 0.000260417 * theta_po4;
  } else {
    inv_theta = 1.0 / theta;  // one reciprocal serves imag_factor and the V-matrix coefficients below
    imag_factor = sh * inv_theta;
  }
  Se3 r;
  r.qw = real_factor; r.qx = imag_factor * o0; r.qy = imag_factor * o1; r.qz = imag_factor * o2;
  quat_normalize(r);
  const double O[9] = { 0, -o2, o1, o2, 0, -o0, -o1, o0, 0 };
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
    so3_matrix(r, V);
  } else {
    const double inv_sq = inv_theta * inv_theta;
    const double c1 = (2 * sh * sh) * inv_sq;                          // (1 - cos(theta)) / theta^2
    const double c2 = (theta - 2 * sh * ch) * (inv_sq * inv_theta);    // (theta - sin(theta)) / theta^3
    for (int i = 0; i < 9; i++) {
      const double id = (i == 0 || i == 4 || i == 8) ? 1.0 : 0.0;
      V[i] = (id + c1 * O[i]) + c2 * O2[i];
    }
  }
  double t[3];
  for (int i = 0; i < 3; i++) {
    double s = V[i * 3 + 0] * u[0];
    s += V[i * 3 + 1] * u[1];
    s += V[i * 3 + 2] * u[2];
    t[i] = s;
  }
  r.tx = t[0]; r.ty = t[1]; r.tz = t[2];
  return r;
}

// AbstractCamera::world2cam(Vector3d), src/camera.cpp:89-125 (Pinhole +radtan),
// :196-221 (FOV), :295-303 (Equidistant)
HSO_HD void world2cam(const hso_camera& cam, double x, double y, double z, double& pu, double& pv)
{
  const double u = x / z, v = y / z;
  if (cam.model == HSO_CAM_PINHOLE && cam.distortion) {
    const double r2 = u * u + v * v;
    const double r4 = r2 * r2;
    const double r6 = r4 * r2;
    const double a1 = 2 * u * v;
    const double a2 = r2 + 2 * u * u;
    const double a3 = r2 + 2 * v * v;
    const double cdist = 1 + cam.d[0] * r2 + cam.d[1] * r4 + cam.d[4] * r6;
    const double xd = u * cdist + cam.d[2] * a1 + cam.d[3] * a2;
    const double yd = v * cdist + cam.d[2] * a3 + cam.d[3] * a1;
    pu = xd * cam.fx + cam.cx;
    pv = yd * cam.fy + cam.cy;
  } else if (cam.model == HSO_CAM_FOV && cam.distortion) {
    const double omega = cam.d[0];
    const double dist = sqrt(u * u + v * v);
    const double ratio = (omega == 0 || dist == 0) ? 1 : atan(2 * dist * tan(omega / 2)) / (dist * omega);
    pu = ratio * cam.fx * u + cam.cx;
    pv = ratio * cam.fy * v + cam.cy;
  } else {
    pu = cam.fx * u + cam.cx;
    pv = cam.fy * v + cam.cy;
  }
}

// Frame::jacobian_xyz2uv, include/hso/frame.h:192-212 (rows 0 and 1, 6 each)
HSO_HD void jacobian_xyz2uv(double x, double y, double z, double J0[6], double J1[6])
{
  const double z_inv = 1. / z;
  const double z_inv_2 = z_inv * z_inv;
  J0[0] = -z_inv; J0[1] = 0.0; J0[2] = x * z_inv_2; J0[3] = y * J0[2];
  J0[4] = -(1.0 + x * J0[2]); J0[5] = y * z_inv;
  J1[0] = 0.0; J1[1] = -z_inv; J1[2] = y * z_inv_2; J1[3] = 1.0 + y * J1[2];
  J1[4] = -J0[3]; J1[5] = -x * z_inv;
}

// Eigen::LDLT<Matrix<double,N,N>> solve, pivoted (Eigen/src/Cholesky/LDLT.h),
// used at CoarseTracker.cpp:114 (N=7) and pose_optimizer.cpp:595 (N=6).
template <int N>
HSO_HD void ldlt_solve(const double* A, const double* b, double* x)
{
  double m[N * N];
  int tr[N];
  double temp[N];
  for (int i = 0; i < N * N; i++) m[i] = A[i];
#define HSO_M(i, j) m[(i) * N + (j)]
  for (int k = 0; k < N; k++) {
    int idx = k;
    double biggest = fabs(HSO_M(k, k));
    for (int i = k + 1; i < N; i++)
      if (fabs(HSO_M(i, i)) > biggest) { biggest = fabs(HSO_M(i, i)); idx = i; }
    tr[k] = idx;
    if (k != idx) {
      const int s = N - idx - 1;
      for (int j = 0; j < k; j++) { double t = HSO_M(k, j); HSO_M(k, j) = HSO_M(idx, j); HSO_M(idx, j) = t; }
      for (int i = 0; i < s; i++) {
        double t = HSO_M(idx + 1 + i, k); HSO_M(idx + 1 + i, k) = HSO_M(idx + 1 + i, idx); HSO_M(idx + 1 + i, idx) = t;
      }
      { double t = HSO_M(k, k); HSO_M(k, k) = HSO_M(idx, idx); HSO_M(idx, idx) = t; }
      for (int i = k + 1; i < idx; i++) { double t = HSO_M(i, k); HSO_M(i, k) = HSO_M(idx, i); HSO_M(idx, i) = t; }
    }
    const int rs = N - k - 1;
    if (k > 0) {
      for (int j = 0; j < k; j++) temp[j] = HSO_M(j, j) * HSO_M(k, j);
      double s = 0;
      for (int j = 0; j < k; j++) s += HSO_M(k, j) * temp[j];
      HSO_M(k, k) -= s;
      for (int i = 0; i < rs; i++) {
        double a = 0;
        for (int j = 0; j < k; j++) a += HSO_M(k + 1 + i, j) * temp[j];
        HSO_M(k + 1 + i, k) -= a;
      }
    }
    const double realAkk = HSO_M(k, k);
    const bool pivot_is_valid = fabs(realAkk) > 0;
    if (k == 0 && !pivot_is_valid) {
      for (int j = 0; j < N; j++) tr[j] = j;
      break;
    }
    if (rs > 0 && pivot_is_valid)
      for (int i = 0; i < rs; i++) HSO_M(k + 1 + i, k) /= realAkk;
  }
  double d[N];
  for (int i = 0; i < N; i++) d[i] = b[i];
  for (int k = 0; k < N; k++) if (tr[k] != k) { double t = d[k]; d[k] = d[tr[k]]; d[tr[k]] = t; }
  for (int i = 0; i < N; i++) {
    double s = d[i];
    for (int j = 0; j < i; j++) s -= HSO_M(i, j) * d[j];
    d[i] = s;
  }
  const double tolerance = 1.0 / 1.7976931348623157e308;
  for (int i = 0; i < N; i++) {
    if (fabs(HSO_M(i, i)) > tolerance) d[i] /= HSO_M(i, i);
    else d[i] = 0;
  }
  for (int i = N - 1; i >= 0; i--) {
    double s = d[i];
    for (int j = i + 1; j < N; j++) s -= HSO_M(j, i) * d[j];
    d[i] = s;
  }
  for (int k = N - 1; k >= 0; k--) if (tr[k] != k) { double t = d[k]; d[k] = d[tr[k]]; d[tr[k]] = t; }
  for (int i = 0; i < N; i++) x[i] = d[i];
#undef HSO_M
}

// ---- wave64 / workgroup reductions (gfx950: 64-wide wavefronts, DPP) ----

template <int CTRL, int ROW_MASK>
HSO_DEV double dpp_get(double v)
{
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, 0xf, false);
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, 0xf, false);
  return __hiloint2double(hi, lo);
}
template <int CTRL, int ROW_MASK>
HSO_DEV int dpp_get(int v)
{
  return __builtin_amdgcn_update_dpp(0, v, CTRL, ROW_MASK, 0xf, false);
}

// Sum over the 64 lanes of a wavefront; the total lands in lane 63.
// Fixed order => deterministic for floating point.
template <typename T>
HSO_DEV T wave_sum_to_lane63(T v)
{
  v += dpp_get<0xb1, 0xf>(v);   // quad_perm [1,0,3,2]
  v += dpp_get<0x4e, 0xf>(v);   // quad_perm [2,3,0,1]
  v += dpp_get<0x124, 0xf>(v);  // row_ror:4
  v += dpp_get<0x128, 0xf>(v);  // row_ror:8
  v += dpp_get<0x142, 0xa>(v);  // row_bcast:15 -> rows 1,3
  v += dpp_get<0x143, 0xc>(v);  // row_bcast:31 -> rows 2,3
  return v;
}

}  // namespace hso_dev
