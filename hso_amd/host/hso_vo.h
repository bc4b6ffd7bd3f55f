// hso_vo.h — the per-frame pipeline of the reference on top of the device calls: Map, candidate lists,
// DepthFilter::updateSeeds, ba::LocalBundleAdjustment and FrameHandlerMono with its addImage() entry,
// in the reference's own names, call order and argument values.
//
// What is mirrored: src/frame_handler_mono.cpp:80-123 (addImage), :173-355 (processFrame), :357-386
// (relocalizeFrame), :419-426 (setFirstFrame), :428-507 (needNewKf), :559-647 (createCovisibilityGraph);
// src/frame_handler_base.cpp:95-179; src/map.cpp (keyframes, candidates, temporary points, trash);
// src/reprojector.cpp:88-331 incl. the seed branch; src/depth_filter.cpp:130-205, 330-509;
// src/bundle_adjustment.cpp:556-897.  What is not: the two-view initialisation (src/initialization.cpp —
// OpenCV KLT + essential-matrix RANSAC); a sequence starts from setFirstFrame() with a keyframe whose
// features carry depths (the hook the reference keeps "for synthetic datasets", frame_handler_mono.h:49-50).
//
// Schedule: the reference runs the depth filter in its own thread, racing the tracker.  Here the
// same calls run synchronously in the order the thread would take them when it keeps up: addFrame ->
// updateSeeds(frame); addKeyframe -> updateSeeds(keyframe) + initializeSeeds(keyframe) (depth_filter.cpp:
// 317-327).  The idle-time pass over earlier frames (observeDepthWithPreviousFrameOnce, :677-726) only runs when
// that thread has nothing queued; it is left out, so Seed::optFrames_P stays empty.  Reprojector's
// std::random_shuffle of the cell order is replaced by a fixed order (Reprojector::cell_order).
#pragma once
#include <list>
#include <map>
#include <set>
#include <string>
#include <utility>
#include "hso_host.h"
#include "hso_init.h"

namespace hso {

// src/config.cpp:28-64 — the values the hot path reads
// The reference's Config is a process-wide singleton (include/hso/config.h).  Here every handle owns one (hso_vo::cfg) and the C
// interface makes it the calling thread's current configuration for the duration of each call (vo_guard, hso_vo.cpp), so two
// handles with different max_fts do not change each other's behaviour; code outside a handle sees the defaults.
struct Config {
  static thread_local Config* current_;
  static Config& get() { static Config defaults; return current_ ? *current_ : defaults; }
  int n_pyr_levels = 3, core_n_kfs = 7, grid_size = 36, klt_max_level = 4, klt_min_level = 0;
  double poseoptim_thresh = 2.0;
  int loba_num_iter = 10, max_fts = 200, quality_min_fts = 5, quality_max_drop_fts = 40;
};

// include/hso/map.h:44-98
class MapPointCandidates {
public:
  typedef std::pair<Point*, Feature*> PointCandidate;
  typedef std::list<PointCandidate> PointCandidateList;
  PointCandidateList candidates_;
  std::list<std::pair<Point*, Feature*>> temporaryPoints_;
  std::list<Point*> trash_points_;
  std::set<Point*> graveyard_;   // trashed points: storage kept until reset() (see Map::emptyTrash), freed there
  ~MapPointCandidates() { reset(); }
  void newCandidatePoint(Point* point, double depth_sigma2);     // src/map.cpp:300-306
  void addPauseSeedPoint(Point* point);                         // :308-316
  void addCandidatePointToFrame(FramePtr frame);                // :318-360
  bool deleteCandidatePoint(Point* point);                      // :362-380
  void changeCandidatePosition(Frame* frame);                   // :382-399
  void removeFrameCandidates(FramePtr frame);                   // :401-428
  void reset();                                                 // :430-444
  void deleteCandidate(PointCandidate& c);                      // :446-461
  void emptyTrash();                                            // :463-469
};

// include/hso/map.h:101-180
class Map {
public:
  std::list<FramePtr> keyframes_;
  std::list<Point*> trash_points_;
  std::set<Point*> graveyard_;   // see emptyTrash
  MapPointCandidates point_candidates_;
  ~Map() { reset(); }
  void reset();                                                 // src/map.cpp:42-47
  void removePtFrameRef(Frame* frame, Feature* ftr);            // :102-116
  void safeDeletePoint(Point* pt);                              // :118-129
  void safeDeleteTempPoint(std::pair<Point*, Feature*>& p);     // :131-181
  void deletePoint(Point* pt);                                  // :184-188
  void addKeyframe(FramePtr new_keyframe) { keyframes_.push_back(new_keyframe); }   // :190-193
  void getCloseKeyframes(const FramePtr& frame, std::list<std::pair<FramePtr, double>>& close_kfs) const;  // :195-213
  FramePtr getClosestKeyframe(const FramePtr& frame) const;     // :215-233
  bool getKeyframeById(int id, FramePtr& frame) const;          // :250-261
  void emptyTrash();                                            // :282-290
  FramePtr lastKeyframe() { return keyframes_.back(); }
  size_t size() const { return keyframes_.size(); }
};

namespace frame_utils {
bool getSceneDepth(const Frame& frame, double& depth_mean, double& depth_min);   // src/frame.cpp:323-345
bool getSceneDistance(const Frame& frame, double& distance_mean);               // :347-366
}

class SeedFilter;   // the synchronous DepthFilter of this driver

// the Reprojector of hso_host.h with the map behind it: src/reprojector.cpp:88-331
class MapReprojector : public Reprojector {
public:
  MapReprojector(AbstractCamera* cam, Map& map, int max_fts) : Reprojector(cam, max_fts), map_(map) { map_ptr_ = &map; }
  void reprojectMap(FramePtr frame, std::vector<std::pair<FramePtr, size_t>>& overlap_kfs);
  SeedFilter* depth_filter_ = nullptr;
  size_t n_seeds_ = 0, sum_seed_ = 0;
  bool reproject_unconverged_seeds = true;    // Options, include/hso/reprojector.h:57-63
  float reproject_seed_thresh = 86;
protected:
  void dropUnknownPoint(Point* pt) override { map_.safeDeletePoint(pt); }                           // :377-378
  void dropCandidatePoint(Point* pt) override { map_.point_candidates_.deleteCandidatePoint(pt); }  // :379-380
private:
  Map& map_;
};

// DepthFilter (include/hso/depth_filter.h:97-236) without its thread: see the schedule note above
class SeedFilter : public DepthFilter {
public:
  typedef void (*callback_t)(void* user, Point* point, double sigma2);
  SeedFilter(FeatureExtractor* fe, callback_t cb, void* cb_user) : DepthFilter(-1), seed_converged_cb_(cb), cb_user_(cb_user) { featureExtractor_ = fe; }
  void addFrame(FramePtr frame);                                                        // src/depth_filter.cpp:130-145
  void addKeyframe(FramePtr frame, double depth_mean, double depth_min, float converge_thresh = 200);   // :147-162
  void updateSeeds(FramePtr frame);                                                     // :330-509
  void reset() { seeds_.clear(); }                                                      // :221-236
  size_t nMeanConvergeFrame_ = 6;
  int max_n_kfs = 3;                                                                    // Options, depth_filter.h:121
  std::vector<size_t> m_v_n_converge;
  size_t n_activated_ = 0, n_converged_ = 0;                                            // diagnostics
private:
  void observeDepth();                                                                  // :557-675
  void activateConverged();                                                             // :405-497 with activatePoint :729-852
  callback_t seed_converged_cb_;
  void* cb_user_;
  FramePtr active_frame_;
};

namespace ba {
// src/bundle_adjustment.cpp:556-897
void LocalBundleAdjustment(Frame* center_kf, std::set<Frame*>* core_kfs, Map* map, size_t& n_incorrect_edges_1,
                           size_t& n_incorrect_edges_2, double& init_error, double& final_error);
}

// include/hso/frame_handler_base.h + frame_handler_mono.h
class FrameHandlerMono {
public:
  enum Stage { STAGE_PAUSED, STAGE_FIRST_FRAME, STAGE_SECOND_FRAME, STAGE_DEFAULT_FRAME, STAGE_RELOCALIZING };
  enum TrackingQuality { TRACKING_INSUFFICIENT, TRACKING_BAD, TRACKING_GOOD };
  enum UpdateResult { RESULT_NO_KEYFRAME, RESULT_IS_KEYFRAME, RESULT_FAILURE };

  FrameHandlerMono(hso_gpu_ctx* ctx, AbstractCamera* cam, bool use_pc = false);
  ~FrameHandlerMono();
  // src/frame_handler_mono.cpp:80-123.  img: 8-bit, camera-sized (else std::runtime_error, src/frame.cpp:85-86)
  void addImage(const uint8_t* img, int width, int height, double timestamp);
  // :419-426 — the first keyframe comes from the caller (features with points), no two-view initialisation
  void setFirstFrame(const FramePtr& first_frame);
  // FrameHandlerBase::start() (include/hso/frame_handler_base.h:75): the next addImage is the first frame of the two-view
  // initialisation (processFirstFrame / processSecondFrame, src/frame_handler_mono.cpp:125-172)
  void start() { set_start_ = true; }
  initialization::KltHomographyInit klt_homography_init_;
  FramePtr lastFrame() { return last_frame_; }
  Stage stage() const { return stage_; }
  TrackingQuality trackingQuality() const { return tracking_quality_; }
  UpdateResult lastResult() const { return last_result_; }
  Map map_;
  // per-stage counters of the last processFrame (what the reference logs through HSO_LOG)
  struct FrameLog {
    size_t img_align_n_tracked = 0, repr_n_mps = 0, repr_n_matches = 0, repr_n_seeds = 0, sfba_n_edges_final = 0;
    double sfba_thresh = 0, sfba_error_init = 0, sfba_error_final = 0, loba_err_init = 0, loba_err_fin = 0;
    size_t loba_n_erredges_init = 0, loba_n_erredges_fin = 0, n_seeds = 0, n_candidates = 0;
    int used_inverse = 0;
  } log_;
  SeedFilter* depth_filter_ = nullptr;
  MapReprojector reprojector_;
  std::set<Frame*> LocalMap_;

protected:
  UpdateResult processFirstFrame();
  UpdateResult processSecondFrame();
  UpdateResult processFrame();
  UpdateResult relocalizeFrame(const SE3& T_cur_ref, FramePtr ref_keyframe);
  bool needNewKf(const double& scene_depth_mean, const size_t& num_observations);
  void createCovisibilityGraph(FramePtr currentFrame, size_t n_closest, bool is_keyframe);
  void setTrackingQuality(size_t num_observations);     // src/frame_handler_base.cpp:165-179
  void resetAll();
  bool startFrameProcessingCommon(double timestamp);   // :95-114
  int finishFrameProcessingCommon(size_t update_id, UpdateResult dropout, size_t num_observations);   // :116-152

  hso_gpu_ctx* ctx_;
  AbstractCamera* cam_;
  FeatureExtractor* feature_extractor_ = nullptr;
  Stage stage_ = STAGE_PAUSED;
  bool set_reset_ = false, set_start_ = false;
  TrackingQuality tracking_quality_ = TRACKING_INSUFFICIENT;
  UpdateResult last_result_ = RESULT_NO_KEYFRAME;
  size_t num_obs_last_ = 0;
  FramePtr new_frame_, last_frame_, firstFrame_;
  std::set<FramePtr> core_kfs_;
  std::vector<std::pair<FramePtr, size_t>> overlap_kfs_;
  SE3 motionModel_;
  bool afterInit_ = false;
  int regular_counter_ = 0;
};

}  // namespace hso
