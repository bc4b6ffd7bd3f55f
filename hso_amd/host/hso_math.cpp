// hso_math.cpp — SE3 products and the camera models (see hso_math.h).
#include "hso_math.h"
#include <cmath>

namespace hso {

static void quat_rotate(const double q[4], const double v[3], double o[3])
{
  double uv[3] = {q[1] * v[2] - q[2] * v[1], q[2] * v[0] - q[0] * v[2], q[0] * v[1] - q[1] * v[0]};
  uv[0] += uv[0]; uv[1] += uv[1]; uv[2] += uv[2];
  o[0] = (v[0] + q[3] * uv[0]) + (q[1] * uv[2] - q[2] * uv[1]);
  o[1] = (v[1] + q[3] * uv[1]) + (q[2] * uv[0] - q[0] * uv[2]);
  o[2] = (v[2] + q[3] * uv[2]) + (q[0] * uv[1] - q[1] * uv[0]);
}
static void quat_normalize(double q[4])
{
  const double n = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  for (int i = 0; i < 4; i++) q[i] /= n;
}

SE3 SE3::operator*(const SE3& o) const
{
  SE3 r;
  double rt[3];
  quat_rotate(v.q, o.v.t, rt);
  for (int i = 0; i < 3; i++) r.v.t[i] = v.t[i] + rt[i];
  const double *a = v.q, *b = o.v.q;
  r.v.q[3] = a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2];
  r.v.q[0] = a[3] * b[0] + a[0] * b[3] + a[1] * b[2] - a[2] * b[1];
  r.v.q[1] = a[3] * b[1] + a[1] * b[3] + a[2] * b[0] - a[0] * b[2];
  r.v.q[2] = a[3] * b[2] + a[2] * b[3] + a[0] * b[1] - a[1] * b[0];
  quat_normalize(r.v.q);
  return r;
}

Vector3d SE3::operator*(const Vector3d& p) const
{
  double o[3];
  quat_rotate(v.q, p.data(), o);
  return {o[0] + v.t[0], o[1] + v.t[1], o[2] + v.t[2]};
}

SE3 SE3::inverse() const
{
  SE3 r;
  r.v.q[0] = -v.q[0]; r.v.q[1] = -v.q[1]; r.v.q[2] = -v.q[2]; r.v.q[3] = v.q[3];
  quat_normalize(r.v.q);
  const double nt[3] = {v.t[0] * -1., v.t[1] * -1., v.t[2] * -1.};
  quat_rotate(r.v.q, nt, r.v.t);
  return r;
}

double AbstractCamera::errorMultiplier2() const
{
  return (c_.fx * c_.fy < 0) ? std::fabs(c_.fx) : std::fabs((c_.fx + c_.fy) * 0.5);
}

Vector2d AbstractCamera::world2cam(const Vector3d& xyz) const
{
  const double u = xyz[0] / xyz[2], v = xyz[1] / xyz[2];
  if (c_.model == HSO_CAM_PINHOLE && c_.distortion) {
    const double r2 = u * u + v * v, r4 = r2 * r2, r6 = r4 * r2;
    const double a1 = 2 * u * v, a2 = r2 + 2 * u * u, a3 = r2 + 2 * v * v;
    const double cdist = 1 + c_.d[0] * r2 + c_.d[1] * r4 + c_.d[4] * r6;
    const double xd = u * cdist + c_.d[2] * a1 + c_.d[3] * a2;
    const double yd = v * cdist + c_.d[2] * a3 + c_.d[3] * a1;
    return {xd * c_.fx + c_.cx, yd * c_.fy + c_.cy};
  }
  if (c_.model == HSO_CAM_FOV && c_.distortion) {
    const double omega = c_.d[0];
    const double dist = std::sqrt(u * u + v * v);
    const double ratio = (omega == 0 || dist == 0) ? 1 : std::atan(2 * dist * std::tan(omega / 2)) / (dist * omega);
    return {ratio * c_.fx * u + c_.cx, ratio * c_.fy * v + c_.cy};
  }
  return {c_.fx * u + c_.cx, c_.fy * v + c_.cy};
}

Vector3d AbstractCamera::cam2world(const Vector2d& px) const
{
  double x, y;
  if (c_.model == HSO_CAM_PINHOLE && c_.distortion) {
    // cv::undistortPoints with float K, D and float I/O, five fixed-point iterations (src/camera.cpp:43-45,78-85)
    const double fx = (float)c_.fx, fy = (float)c_.fy, cx = (float)c_.cx, cy = (float)c_.cy;
    const double k0 = (float)c_.d[0], k1 = (float)c_.d[1], p1 = (float)c_.d[2], p2 = (float)c_.d[3], k2 = (float)c_.d[4];
    x = (float)px[0]; y = (float)px[1];
    const double x0 = x = (x - cx) * (1. / fx);
    const double y0 = y = (y - cy) * (1. / fy);
    for (int it = 0; it < 5; it++) {
      const double r2 = x * x + y * y;
      const double icdist = 1 / (1 + ((k2 * r2 + k1) * r2 + k0) * r2);
      const double dX = 2 * p1 * x * y + p2 * (r2 + 2 * x * x);
      const double dY = p1 * (r2 + 2 * y * y) + 2 * p2 * x * y;
      x = (x0 - dX) * icdist; y = (y0 - dY) * icdist;
    }
    x = (float)x; y = (float)y;
  } else if (c_.model == HSO_CAM_FOV && c_.distortion) {
    const double ud = (px[0] - c_.cx) / c_.fx, vd = (px[1] - c_.cy) / c_.fy;
    const double dist = std::sqrt(ud * ud + vd * vd);
    const double rd = std::tan(dist * c_.d[0]) / (2 * dist * std::tan(c_.d[0] / 2));
    x = rd * ud; y = rd * vd;
  } else {
    x = (px[0] - c_.cx) / c_.fx; y = (px[1] - c_.cy) / c_.fy;
  }
  const double n = std::sqrt(x * x + y * y + 1.0);
  return {x / n, y / n, 1.0 / n};
}

}  // namespace hso
