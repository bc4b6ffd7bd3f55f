// hso_host.h — host-side mirror of the reference's C++ call surface for the hot path.
//
// The reference has no FFI: FrameHandlerMono::processFrame() calls C++ classes directly
// (SURVEY.md §8b).  This header keeps those names, argument meanings and side effects so a
// maintainer can swap the bodies: `hso::CoarseTracker(inverse, max_level, min_level, n_iter,
// verbose).run(ref, cur)` is the reference's signature (include/hso/CoarseTracker.h:134,141)
// and mutates cur->T_f_w_ / cur->m_exposure_time exactly as src/CoarseTracker.cpp:198-202 does;
// the numeric body is one call into the C-ABI (include/hso_gpu.h).  Plain C++17, no Eigen /
// OpenCV / Boost (none of them exists on the target image); SE3 is the Sophus convention
// (unit quaternion + translation, tangent [upsilon, omega]).
#pragma once
#include <array>
#include <cstdint>
#include <list>
#include <memory>
#include <stdexcept>
#include <vector>
#include "../../include/hso_gpu.h"
#include "hso_math.h"

namespace hso {

struct Feature;
class Frame;
using FramePtr = std::shared_ptr<Frame>;

// include/hso/point.h:53-116 (fields the tracker reads)
class Point {
public:
  enum PointType { TYPE_DELETED, TYPE_TEMPORARY, TYPE_CANDIDATE, TYPE_UNKNOWN, TYPE_GOOD };  // point.h:53
  PointType type_ = TYPE_UNKNOWN;
  double idist_ = 1.0;            // inverse depth in the host frame (point.h:115)
  Feature* hostFeature_ = nullptr;
  Vector3d pos_{0, 0, 0};         // world position (point.h:63)
  std::list<Feature*> obs_;       // keyframe observations (point.h:66)
  enum FeatureType { FEATURE_GRADIENT, FEATURE_EDGELET, FEATURE_CORNER };                      // point.h:54
  FeatureType ftr_type_ = FEATURE_CORNER;
  int last_projected_kf_id_ = -1;                                                              // point.h:73
  int n_failed_reproj_ = 0, n_succeeded_reproj_ = 0;                                           // point.h:75-76
  bool isBad_ = false;
  static thread_local int point_counter_;
  int id_ = point_counter_++;
  int seedStates_ = 0;            // temporary points: 0 seed still alive, 1 converged, -1 dropped (point.h)
  int nBA_ = 0;
  Point() = default;
  Point(const Vector3d& pos, Feature* ftr) : pos_(pos) { obs_.push_front(ftr); }                // src/point.cpp:55-73
  void addFrameRef(Feature* ftr) { obs_.push_front(ftr); }                                     // :78-82
  bool deleteFrameRef(Frame* frame);                                                           // :92-102
  // src/point.cpp:116-136: the observation whose viewing direction is closest to `framepos`
  bool getCloseViewObs(const Vector3d& framepos, Feature*& ftr) const;
};

// include/hso/feature.h:33-60
struct Feature {
  enum FeatureType { CORNER, EDGELET, GRADIENT };
  FeatureType type = CORNER;
  Frame* frame = nullptr;
  Vector2d px{0, 0};
  Vector3d f{0, 0, 1};
  int level = 0;
  Point* point = nullptr;
  Vector2d grad{1, 0};
};
using Features = std::list<Feature*>;

// include/hso/frame.h — a Frame owns a device-resident pyramid in the shared GPU context
class Frame {
public:
  // `new Frame(cam, img, ts)` -> initFrame (src/frame.cpp:82-96): throws std::runtime_error when
  // the image size differs from the camera's, like the reference
  Frame(hso_gpu_ctx* ctx, AbstractCamera* cam, const uint8_t* img, int width, int height, double timestamp);
  ~Frame();
  Frame(const Frame&) = delete;
  Frame& operator=(const Frame&) = delete;

  static thread_local int frame_counter_;
  static thread_local int id_base_;   // the id the sequence's first frame gets (0; k << 24 for sequence k of the multi-sequence driver: one context holds all frames)
  int id_;
  double timestamp_;
  AbstractCamera* cam_;
  SE3 T_f_w_;
  Vector3d pos() const { return T_f_w_.inverse().translation(); }   // include/hso/frame.h:142
  int keyFrameId_ = 0;
  static thread_local int keyFrameCounter_;   // src/frame.cpp:36
  bool is_keyframe_ = false;
  bool isKeyframe() const { return is_keyframe_; }
  void setKeyframe();                          // src/frame.cpp:98-105
  void setKeyPoints();                         // :121-129
  void checkKeyPoints(Feature* ftr);           // :131-179
  void removeKeyPoint(Feature* ftr);           // :181-192
  bool isVisible(const Vector3d& xyz_w) const; // :194-204
  void addFeature(Feature* ftr) { fts_.push_back(ftr); }   // :107-111
  std::array<Feature*, 5> key_pts_{{nullptr, nullptr, nullptr, nullptr, nullptr}};   // frame.h:88
  std::vector<Frame*> connectedKeyFrames;      // frame.h: covisibility neighbours of this frame (<= 5)
  std::shared_ptr<Frame> m_last_frame;         // the frame it was tracked against
  int lastReprojectFrameId_ = -1;
  size_t m_n_inliers = 0;
  Features fts_;                 // owned, deleted by ~Frame (src/frame.cpp:54-72)
  float integralImage_ = 0;      // src/frame.cpp:238
  float gradMean_ = 0;           // src/frame.cpp:240-245
  double m_exposure_time = -1;
  double Cov_[36] = {0};         // 6x6 pose covariance, row-major (frame.h:93)
  float m_error_in_px = 0;
  hso_gpu_ctx* ctx_;
};

// include/hso/CoarseTracker.h:134-141
class CoarseTracker {
public:
  CoarseTracker(bool inverse_composition, int max_level, int min_level, int n_iter, bool verbose);
  size_t run(FramePtr ref_frame, FramePtr cur_frame);
  bool m_inverse_composition;
  int m_max_level, m_min_level, m_n_iter;
  bool m_verbose;
  SE3 m_T_cur_ref;
  hso_track_result m_last{};     // per-level diagnostics of the last run
};

// include/hso/matcher.h:108-215 — reprojection matching of one map point
class Matcher {
public:
  // src/matcher.cpp:270-375.  Chooses the reference observation (getCloseViewObs), then the
  // warp / Lucas-Kanade / NCC body runs on the device (hso_gpu_align_batch with one job).
  // px_cur: in = projected position, out = refined position (level-0 pixels).
  bool findMatchDirect(const Point& pt, Frame& cur_frame, Vector2d& px_cur);
  // The batched form a Reprojector adapter uses: all candidates of a frame in one launch; the
  // result array is in candidate order so the caller can apply its first-success-per-cell rule.
  static std::vector<hso_align_out> findMatchDirectBatch(const std::vector<const Point*>& pts, Frame& cur_frame,
                                                          const std::vector<Vector2d>& px_cur, std::vector<Feature*>* ref_ftrs);
  Feature* ref_ftr_ = nullptr;
  int search_level_ = 0;
  double A_cur_ref_[4] = {1, 0, 0, 1};
  double h_inv_ = 0;
  hso_align_out last_{};
};

// include/hso/reprojector.h:55-190 — map points of the overlapping keyframes into the new frame
class Reprojector {
public:
  struct Candidate { Point* pt; Vector2d px; int slot; };   // reprojector.h:95-103 (+ the row of the device call)
  // initializeGrid, src/reprojector.cpp:53-75; max_fts stands in for Config::maxFts()
  Reprojector(AbstractCamera* cam, int max_fts);
  // src/reprojector.cpp:88-331 for the map points of `kfs` — the overlap keyframes in the order
  // the reference visits them (covisibility first, then by closeness; that walk belongs to the
  // Map, which the mirror does not have).  One hso_gpu_reproject_match call projects, bins,
  // chooses reference observations and matches; reprojectCell / reprojectCellAll (:352-429,
  // :556-612) then read the results in the reference's visiting order: same features added to
  // frame->fts_, same counters.  Points the reference would hand to Map::safeDeletePoint are
  // marked TYPE_DELETED.
  void reprojectMap(FramePtr frame, const std::vector<FramePtr>& kfs, std::vector<std::pair<FramePtr, size_t>>& overlap_kfs);
  // the visiting order of the cells: the reference reshuffles it with std::random_shuffle on
  // every call (:72, :82); here it is the caller's (identity by default), so runs are reproducible
  std::vector<int> cell_order;
  size_t n_matches_ = 0, n_trials_ = 0, nFeatures_ = 0;
  int cell_size, grid_n_cols, grid_n_rows, max_fts_;
  size_t max_n_kfs = 10;           // Options::max_n_kfs, reprojector.h:67
  virtual ~Reprojector() {}
protected:
  // project every point into `frame`, choose its reference observation and match it: one device call
  // (hso_gpu_reproject_match); fills proj_ / match_ / ref_of_slot_ in the order of `pts`
  void projectAndMatch(FramePtr frame, const std::vector<Point*>& pts);
  // reprojectCellAll or the three reprojectCell passes over the cells (:261-306): the policy runs on the device
  // (hso_gpu_reproject_select), its list of examined candidates is applied here
  void selectMatches(FramePtr frame, const std::vector<Candidate>& all);
  bool applyMatch(const Candidate& c, FramePtr frame);
  // what the reference hands to Map::safeDeletePoint / MapPointCandidates::deleteCandidatePoint (:377-380);
  // without a map behind the reprojector the point is only marked
  virtual void dropUnknownPoint(Point* pt) { pt->type_ = Point::TYPE_DELETED; }
  virtual void dropCandidatePoint(Point* pt) { pt->type_ = Point::TYPE_DELETED; }
  std::vector<hso_align_out> match_;
  std::vector<hso_reproj_point> proj_;
  std::vector<const Feature*> ref_of_slot_;
  void* map_ptr_ = nullptr;
};

// include/hso/pose_optimizer.h — motion-only refinement of frame->T_f_w_ over its features' points
namespace pose_optimizer {
// src/pose_optimizer.cpp:399-771.  Mutates frame->T_f_w_, frame->Cov_ and frame->m_error_in_px,
// sets feature->point = NULL for the observations it rejects (:722-748), like the reference.
void optimizeLevenbergMarquardt3rd(const double reproj_thresh, const size_t n_iter, const bool verbose, FramePtr& frame,
                                   double& estimated_scale, double& error_init, double& error_final, size_t& num_obs);
}

// include/hso/depth_filter.h:45-88
struct Seed {
  static thread_local int batch_counter;      // src/depth_filter.cpp:46
  int batch_id = batch_counter;  // the keyframe batch the seed was created in
  Feature* ftr = nullptr;        // host feature (frame, px, f, level, type, grad)
  float a = 10, b = 10;
  float mu = 0, z_range = 0, sigma2 = 0;
  bool is_update = false, isValid = true;
  bool haveReprojected = false;
  Point* temp = nullptr;         // the temporary point Reprojector::reprojectorSeeds made of it
  std::vector<float> vec_distance;
  std::vector<FramePtr> optFrames_P, optFrames_A;
  float opt_id = 0, converge_thresh = 200;
  Vector2d last_matched_px{0, 0};
  int last_matched_level = 0;
  Seed(Feature* ftr, float depth_mean, float depth_min, float converge_threshold = 200);  // src/depth_filter.cpp:49-68
};

// include/hso/feature_detection.h:283-345 — new candidates of a keyframe
class FeatureExtractor {
public:
  // max_fts stands in for the Config::maxFts() singleton read at construction (:382-385)
  FeatureExtractor(int width, int height, int cellSize, int levels, bool isInit = false, int max_fts = 200);
  // src/feature_detection.cpp:408-497: fastDetectMT + edgeLetDetectMT on the device
  // (hso_gpu_detect_candidates), computeKeyPointsOctTree on the host (hso_gpu_select_octree),
  // then one new Feature per selected key (caller owns them, like the reference).  isInit:
  // fastDetectMT + fillingHole (FAST-12 on level 0, hso_gpu_detect_candidates_init), 2000 features.
  void detect(Frame* frame, float initThresh, float minThresh, Features& fts, Frame* last_frame = nullptr);
  void setExistingFeatures(const Features& fts);   // :1169-1177
  int width_, height_, cellSize_, nLevels_, nFeatures_;
  bool isInit_;
  int minThresh_ = 0;
  size_t extFeatures_ = 0;
  std::vector<hso_keypoint> allFeturesToDistribute_;
};

class DepthFilter {
public:
  explicit DepthFilter(double px_error_angle) : px_error_angle_(px_error_angle) {}
  // src/depth_filter.cpp:146-205 (no thread: addKeyframe -> initializeSeeds): detect new features
  // away from the frame's existing ones and start one seed per feature
  void addKeyframe(FramePtr frame, double depth_mean, double depth_min, float converge_thresh = 200);
  void initializeSeeds(FramePtr frame);
  FeatureExtractor* featureExtractor_ = nullptr;
  double new_keyframe_mean_depth_ = 0, new_keyframe_min_depth_ = 0;
  float convergence_sigma2_thresh_ = 200;
  // src/depth_filter.cpp:557-675: one observation of every seed in `frame`; seeds whose
  // z_inv_min turns NaN are erased like :618-622; returns the number of successful matches
  size_t observeDepth(FramePtr frame);
  std::list<Seed> seeds_;
  double px_error_angle_;
};

}  // namespace hso
