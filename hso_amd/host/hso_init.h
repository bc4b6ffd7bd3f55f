// hso_init.h — the geometry of the two-view start of a sequence (reference src/initialization.cpp:300-474; the frame handling
// around it — KltHomographyInit::addFirstFrame / addSecondFrame, :39-223 — lives in the engine, hso_engine_init.cpp).  The image
// side (trackKlt: pyramidal LK + patchCheck) is one device call (hso_gpu_klt_track); the geometry (computeInitializeMatrix :300-385, computeP3D :387-424, distancePointOnce :428-474,
// vikit's Homography decomposition) is host code in plain C++17 like the rest of the mirror.
//
// What differs from the reference, and why: it estimates the essential matrix and the homography with OpenCV
// (cv::findEssentialMat = five-point RANSAC + cv::recoverPose; cv::findHomography RANSAC) — a dependency that is not in the
// reference tree and not in this image.  Here: Hartley-normalised eight-point RANSAC (Sampson distance, the same threshold
// 2 / errorMultiplier2, confidence 0.99) refitted on its inliers and decomposed with the usual four-fold cheirality test, and a
// normalised four-point DLT RANSAC (same threshold) refitted on its inliers.  Everything after the two models follows the
// reference: Faugeras-Lustman decomposition and pruning of the homography (src/vikit/homography.cpp:89-270), computeP3D for both
// models, the smaller total error wins, then the mapScale rescaling.  Sampling is seeded per call, so a run is reproducible (the
// reference's is not: cv::theRNG() carries state across calls).  Parity of this stage is functional ("initialisation"),
// not bit-level: the selected inliers and the pose agree with ground truth in the synthetic tests.
#pragma once
#include <vector>
#include "hso_math.h"

namespace hso {
namespace initialization {

enum InitResult { FAILURE, NO_KEYFRAME, SUCCESS };

struct Matrix3d { double m[3][3]; };

// src/initialization.cpp:300-385.  f_ref / f_cur: unit bearings; returns the inliers (indices), the points in the current frame
// and T_cur_from_ref of the better model.
void computeInitializeMatrix(const std::vector<Vector3d>& f_ref, const std::vector<Vector3d>& f_cur, double focal_length,
                             double reprojection_threshold, std::vector<int>& inliers, std::vector<Vector3d>& xyz_in_cur, SE3& T_cur_from_ref,
                             int* used_homography = nullptr);
// :387-424
double computeP3D(const std::vector<Vector3d>& rays_first, const std::vector<Vector3d>& rays_second, const Matrix3d& R, const Vector3d& t,
                  double reproj_thresh, double error_multiplier2, std::vector<Vector3d>& vP3D, std::vector<int>& inliers);
// the two model estimators (exposed for the unit tests of the host test driver)
bool estimateEssential(const std::vector<Vector2d>& x1, const std::vector<Vector2d>& x2, double thresh, Matrix3d& R, Vector3d& t);
bool estimateHomography(const std::vector<Vector2d>& x1, const std::vector<Vector2d>& x2, double thresh, Matrix3d& H);
bool decomposeHomography(const Matrix3d& H, const std::vector<Vector2d>& plane_a, const std::vector<Vector2d>& plane_b, double error_multiplier2,
                         double thresh, SE3& T_c2_from_c1);

}  // namespace initialization
}  // namespace hso
