// hso_math.h — the small fixed-size algebra and the camera models the host code shares: Sophus-convention SE3 (unit
// quaternion + translation, tangent [upsilon, omega]; thirdparty/Sophus/sophus/se3.cpp) and AbstractCamera's
// world2cam / cam2world for the three camera records of include/hso_gpu.h (src/camera.cpp).  Plain C++17: neither Eigen
// nor OpenCV exists on the target image.
#pragma once
#include <array>
#include <cstdint>
#include "../../include/hso_gpu.h"

namespace hso {

using Vector2d = std::array<double, 2>;
using Vector3d = std::array<double, 3>;

// Sophus::SE3 subset (thirdparty/Sophus/sophus/se3.cpp)
struct SE3 {
  hso_se3 v{{0, 0, 0, 1}, {0, 0, 0}};
  SE3 operator*(const SE3& o) const;       // se3.cpp:59-66
  Vector3d operator*(const Vector3d& p) const;  // se3.cpp:91-95
  SE3 inverse() const;                     // se3.cpp:76-83
  Vector3d translation() const { return {v.t[0], v.t[1], v.t[2]}; }
};

// include/hso/camera.h — only what the hot path calls
class AbstractCamera {
public:
  explicit AbstractCamera(const hso_camera& c) : c_(c) {}
  int width() const { return c_.width; }
  int height() const { return c_.height; }
  Vector2d focal_length() const { return {c_.fx, c_.fy}; }
  double errorMultiplier2() const;                 // src/camera.cpp:59
  Vector2d world2cam(const Vector3d& xyz) const;   // src/camera.cpp:89-125,196-221
  Vector3d cam2world(const Vector2d& px) const;    // src/camera.cpp:67-87,171-194 (unit bearing)
  const hso_camera& pod() const { return c_; }
private:
  hso_camera c_;
};

}  // namespace hso
