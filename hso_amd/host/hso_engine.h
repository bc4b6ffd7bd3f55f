// hso_engine.h — the sequence engine of libhso_host.so: N independent visual-odometry sequences over one device context.
//
// State is tables with integer links (frames, keyframe features = observation rows, points, seeds), the shapes the device
// library keeps resident (include/hso_gpu.h: hso_gpu_seqmap_*, hso_gpu_seed_table_*).  A step of the engine takes one image per
// sequence and runs every numeric stage ONCE for all sequences:
//
//     frame upload (batch)  ->  CoarseTracker (batch)  ->  reprojection + matching + grid selection + pose optimisation
//     (hso_gpu_reproject_select_pose_frames)  ->  [sequences that take a keyframe: local BA (multi)]  ->  seed observation in the
//     resident table  ->  seed activation (multi)  ->  [keyframes: detection (batch), oct-tree, new seeds]
//
// Between the device calls the sequences' bookkeeping runs in parallel on a small thread pool (it touches nothing but the
// sequence's own tables), and only the main thread talks to the device.  What the bookkeeping decides — which keyframes a frame
// projects and in which order, the quality keys, when a keyframe is taken, which keyframes form the local BA window, what
// happens to seeds, candidates and temporary points — reproduces the reference's decisions (its call order
// and argument values are the drop-in contract): FrameHandlerMono::processFrame (src/frame_handler_mono.cpp:173-355), needNewKf
// (:428-507), createCovisibilityGraph (:559-647), Reprojector::reprojectMap (src/reprojector.cpp:88-331), DepthFilter::updateSeeds
// (src/depth_filter.cpp:330-509), ba::LocalBundleAdjustment (src/bundle_adjustment.cpp:556-897), Map / MapPointCandidates
// (src/map.cpp).  The schedule of the reference's depth-filter thread is the synchronous one (its thread keeping up): a frame's
// seed update runs before the next frame is tracked.
#pragma once
#include <sched.h>
#include <pthread.h>
#include <array>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <functional>
#include <new>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <chrono>
#include <vector>
#include "../../include/hso_vo.h"
#include "../../include/hso_gpu_debug.h"
#include "hso_math.h"

namespace hso {
namespace engine {

using Id = int32_t;
constexpr Id kNone = -1;

// Point::PointType / Point::FeatureType (include/hso/point.h:53-54): the numeric values order the grid selection
enum : int8_t { kPtDeleted = 0, kPtTemporary = 1, kPtCandidate = 2, kPtUnknown = 3, kPtGood = 4 };
enum : int8_t { kOnGradient = 0, kOnEdgelet = 1, kOnCorner = 2 };

struct Settings {                 // src/config.cpp:28-64 and the Options structs the path reads
  int max_fts = 200;
  int n_pyr_levels = 3, core_n_kfs = 7, grid_size = 36, klt_max_level = 4, klt_min_level = 0;
  double poseoptim_thresh = 2.0;
  int loba_num_iter = 10, quality_min_fts = 5, quality_max_drop_fts = 40;
  int reproject_max_kfs = 10;     // Reprojector::Options::max_n_kfs
  float reproject_seed_thresh = 86;
  int seed_max_kfs = 3;           // DepthFilter::Options::max_n_kfs
  bool previous_frame_pass = true;   // the depth thread's idle-time pass (src/depth_filter.cpp:254-263): one sweep per frame
  bool sync_previous = false;        // ... inside the step instead of on the depth filter's own stream (hso_vo_multi_set_option: tests compare the two)
  bool track_no_coop = false;        // hso_gpu_options.track_no_coop of the bank's context
  double map_scale = 1.0, init_min_disparity = 40.0;
  int init_min_tracked = 50, init_min_inliers = 40;
};

struct Feat {                     // a feature; rows of Seq::feats are also the device's observation rows
  double px[2] = {0, 0}, f[3] = {0, 0, 1}, grad[2] = {1, 0};
  Id frame = kNone, point = kNone;
  Id next = kNone;                // the next (older) observation of `point` while `linked`
  int8_t level = 0, type = 0;     // HSO_FTR_*
  bool linked = false;
};

struct Point {
  double pos[3] = {0, 0, 0}, idist = 1;
  Id host = kNone;                // host feature (row of Seq::feats)
  Id host_frame = kNone;          // ... its frame and bearing, copied when the point is made (a point never changes its host): the
  double host_f[3] = {0, 0, 1};   // per-frame loops over a frame's points then touch the point row only
  Id head = kNone;                // newest observation
  int32_t n_obs = 0;
  int8_t kind = kPtUnknown, on = kOnCorner;
  int32_t n_ba = 0;
  int8_t seed_state = 0;          // temporary points: 0 the seed lives, 1 it converged, -1 it was dropped
  bool bad = false;
  // n_failed_reproj_ / n_succeeded_reproj_ live in the device's point row (include/hso_gpu.h: HSO_PT_*): the chain counts them and
  // reports the kind changes they cause; the next patch of the row resets the counters named here (bit 0 failures, bit 1 successes)
  uint8_t dev_reset = 3;
};

struct Frame {
  int64_t dev_id = -1;            // id of the resident frame
  int32_t serial = -1;            // Frame::id_ within the sequence
  int32_t kf_id = 0;              // Frame::keyFrameId_
  int32_t kf_row = -1;            // row in the keyframe table once a keyframe
  double stamp = 0;
  SE3 T;                          // T_f_w_
  float integral = 0, grad_mean = 0;
  double exposure = -1;
  double cov[36] = {0};
  float err_px = 0;
  int32_t n_inliers = 0;
  int32_t refs = 0;               // holders: the handler's last / current frame, the map, seeds' frame lists
  bool in_use = false;
  std::vector<Feat> loose;        // a frame that is not a keyframe owns its features — on the device (the sequence map's frame table);
                                  // here only while the host needs them (promotion to a keyframe, the seed branch, a two-view start)
  int32_t n_fts = 0;              // their number
  std::vector<Id> fts;            // a keyframe lists rows of Seq::feats (Frame::fts_ order)
  int32_t fts_sent = 0;           // ... of which the device's copy of the list holds this many
  std::array<Id, 5> key{{kNone, kNone, kNone, kNone, kNone}};   // Frame::key_pts_
  std::vector<Id> covis;          // connectedKeyFrames (frame slots)
  int32_t visited = -1;           // lastReprojectFrameId_
};

struct Seed {
  Id feat = kNone;                // host feature
  int32_t batch = 0, slot = -1;   // Seed::batch_id; slot in the resident table
  float a = 10, b = 10, mu = 0, z_range = 0, sigma2 = 0, converge = 200;
  bool alive = true, valid = true, updated = false, reprojected = false;
  Id temp = kNone;                // the temporary point made of it
  int32_t n_dist = 1;             // vec_distance.size()
  std::vector<Id> seen;           // optFrames_A: frames it was visible in (<= 15)
  std::vector<Id> seen_before;    // optFrames_P
};

class Pool;                       // the bookkeeping threads
void set_host_share(int banks_in_process);   // how many banks this process runs side by side (each sizes its pool to its share)
int pool_threads_for(int n_sequences);        // worker threads a bank of n sequences would get now
int host_cpu_budget();                        // CPUs the process may keep busy (hardware threads, or the cgroup quota)
int timing_level();                           // HSO_ENGINE_TIMING (developer probe): 0 off, 1 phase split at the end, 2 a line per step

// a grow-only array in page-locked host memory (hso_gpu_host_alloc): the result tables of the batched calls are DMA targets as they
// are — no staging copy on the way back
template <typename T> struct Pinned {
  hso_gpu_ctx* ctx = nullptr; T* p = nullptr; size_t cap = 0;
  ~Pinned() { release(); }
  void release() { if (p) (void)hso_gpu_host_free(ctx, p); p = nullptr; cap = 0; }
  T* need(hso_gpu_ctx* c, size_t n)
  {
    if (n <= cap && p) return p;
    if (p) (void)hso_gpu_host_free(ctx, p);
    ctx = c; p = nullptr; cap = 0;
    void* q = nullptr;
    const size_t want = n + n / 2 + 1024;
    if (hso_gpu_host_alloc(c, want * sizeof(T), &q) < 0) throw std::bad_alloc();
    p = static_cast<T*>(q); cap = want;
    return p;
  }
  T* data() { return p; }
};

struct Trace {                    // device-call recorder (hso_trace.h format), one per sequence
  FILE* f = nullptr;
  bool state = false;             // hso_vo_trace_state: a recorded chain call also keeps the sequence map as it stood before the call
  ~Trace() { close(); }
  bool open(const char* path);
  void close();
  bool on() const { return f != nullptr; }
  void begin(const char* name, uint32_t n_fields);
  void field(const char* key, const void* data, size_t bytes);
  void scalar(const char* key, double v) { field(key, &v, 8); }
};

struct Seq;                       // one sequence's tables (hso_engine.cpp)
struct StepData;                  // one sequence's scratch for the running step

class Bank {
public:
  Bank(hso_gpu_ctx* ctx, bool owns_ctx, const hso_camera& cam, const Settings& cfg, int n_sequences);
  ~Bank();
  Bank(const Bank&) = delete;
  Bank& operator=(const Bank&) = delete;

  int size() const { return (int)seq_.size(); }
  int threads() const { return n_threads_; }
  // imgs[k] == nullptr: sequence k sits the step out
  void set_first_frames(const uint8_t* const* imgs, int w, int h, const double* stamps, const float* const* depth_z, const hso_se3* T_f_w);
  void start(const uint8_t* which);
  // on_device: imgs are device pointers (the images are already in HBM: no PCIe copy in the step)
  void add_images(const uint8_t* const* imgs, int w, int h, const double* stamps, bool on_device = false);
  void status(int k, hso_vo_status* st) const;
  int keyframes(int k, double* stamps, hso_se3* T_f_w, int32_t* frame_ids, int cap) const;
  // every frame the sequence has processed since its start: (timestamp, T_f_w as it stood when the frame was finished)
  int trajectory(int k, double* stamps, hso_se3* T_f_w, int cap) const;
  bool trace(int k, const char* path);
  bool trace_state(int k, bool on);
  void set_options(bool sync_previous, bool track_no_coop, bool no_numa_pin = false);
  void call_counts(int64_t* calls, int64_t* items, int cap) const;
  // algorithmic bytes (SURVEY.md section 8(d) units) of the steps so far: [frame build, tracker, matcher, pose optimiser, seeds]
  void alg_bytes(double* out, int cap) const { for (int i = 0; i < cap && i < 5; i++) out[i] = alg_bytes_[i]; }
  std::string err;
  bool poisoned = false;

private:
  friend struct Seq;
  void step(const uint8_t* const* imgs, int w, int h, const double* stamps, bool on_device);
  // phases of a step (device calls on the caller's thread; per-sequence work through par() / the pool)
  void upload(const std::vector<int>& who, const uint8_t* const* imgs, int w, int h, const double* stamps, bool on_device = false);
  void initialise(const std::vector<int>& who);
  void chain(const std::vector<int>& who);
  void track_group(const std::vector<int>& who, const std::vector<Id>& ref, const std::vector<Id>& cur, const hso_track_params& p);
  void prepare_job(int k, hso_seq_job& job, std::vector<int32_t>& temps);
  void consume_result(int k, const hso_seq_result& r, const std::vector<int32_t>& more_events);
  void trace_chain(const std::vector<int>& who, const std::vector<hso_seq_job>& jobs, const hso_seq_chain_cfg& cfg, const hso_seq_result* res);
  void trace_ba_state(const std::vector<int>& who, const std::vector<hso_seq_ba_job>& jobs, double error_multiplier2, double chi2_corner, double chi2_edgelet);
  void trace_chain_state(const std::vector<int>& who, const std::vector<hso_seq_job>& jobs, const hso_seq_chain_cfg& cfg, const std::vector<int32_t>& temps);
  void fetch_features(const std::vector<int>& who);
  void send_features(const std::vector<int>& who);
  void seed_branch(const std::vector<int>& who);
  void decide(int k);
  void decide_frame(int k);
  void decide_keyframe(int k);
  bool wants_keyframe(int k);
  void link_covisible(int k, bool is_keyframe);
  void promote(int k);
  void make_keyframe(Seq& s, Id fr);
  void window_job(int k);
  void keyframe_ba(const std::vector<int>& who);
  void apply_window(int k);
  void observe_seeds(const std::vector<int>& who);
  void activate_seeds(const std::vector<int>& who);
  void previous_begin(const std::vector<int>& who);
  void previous_collect();
  void start_seeds(const std::vector<int>& who);
  void kill_seed(Seq& s, StepData& d, int i, bool keep_feature);
  void erase_slots(const std::vector<int>& who);
  void drop_sequence_seeds(int k);
  void flush_maps(const std::vector<int>& who);
  void finish(const std::vector<int>& who);
  void par(const std::vector<int>& who, const std::function<void(int)>& fn);
  void check(int rc, const char* what);
  void release_frame(Seq& s, Id fr);
  void release_queued();
  void release_frame_deferred(Seq& s, StepData& d, Id fr);
  void detect(const std::vector<int>& who, const std::vector<Id>& frame, const std::vector<int>& thresh, bool init, int n_levels, int n_features,
              std::vector<std::vector<hso_keypoint>>& keys, std::vector<std::vector<hso_keypoint>>& sel);
  Feat feature_from_key(const hso_keypoint& kp, Id fr) const;

  hso_gpu_ctx* ctx_;
  bool owns_ctx_;
  AbstractCamera cam_;
  Settings cfg_;
  std::vector<Seq*> seq_;
  std::vector<StepData*> step_;
  Pool* pool_ = nullptr;
  int n_threads_ = 0;
  int seed_table_ = -1;
  int cell_size_ = 0, grid_cols_ = 0, grid_rows_ = 0;
  std::vector<int32_t> cell_order_;
  double px_error_angle_ = -1;
  int64_t n_calls_[10] = {0}, n_items_[10] = {0};
  double alg_bytes_[5] = {0, 0, 0, 0, 0};
  // HSO_ENGINE_TIMING: named sections of the step, timed on the engine's own thread (Sub is a scope timer)
  std::vector<std::pair<const char*, double>> sections_;
  bool timing_ = false;
  struct Sub {
    Bank* b; const char* name; std::chrono::steady_clock::time_point t;
    Sub(Bank* bank, const char* n) : b(bank->timing_ ? bank : nullptr), name(n) { if (b) t = std::chrono::steady_clock::now(); }
    ~Sub()
    {
      if (!b) return;
      const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t).count();
      for (auto& e : b->sections_) if (e.first == name) { e.second += ms; return; }
      b->sections_.emplace_back(name, ms);
    }
  };
  double sub_ms_[4] = {0};         // HSO_ENGINE_TIMING: reproject() split into listing / device call / applying
  // the previous-frame pass between previous_begin and previous_collect
  struct PendingPrev { bool on = false, async = false; std::vector<int> who; std::vector<size_t> n_lists; int n_slots = 0;
                       std::vector<hso_seed> before; std::vector<hso_seed_out> full; } pending_prev_;
  bool sync_previous_ = false;     // HSO_ENGINE_SYNC_PREVIOUS=1: the pass runs inside the step (tests compare both modes)
  // hso_vo_options.no_numa_pin == 0: the driving thread and the workers stay on the CPUs of the device's NUMA node (pin_threads)
  bool no_numa_pin_ = false, numa_known_ = false, workers_pinned_ = false;
  cpu_set_t numa_cpus_;
  pthread_t pinned_driver_{};
  bool driver_pinned_ = false;
  void pin_threads();
  std::vector<int64_t> to_release_;
  std::vector<int64_t> after_prev_release_;   // frames the previous-frame pass dropped from its lists: released once the pass is collected
  double phase_ms_[9] = {0};
  int64_t phase_census_[9][7] = {{0}};   // per phase: copies, bytes, staged copies, synchronisations, ns blocked in them, memsets, host-to-device bytes
  int64_t n_steps_ = 0, n_kf_events_ = 0;
  // result tables of the batched calls (kept between steps: no allocation per step)
  Pinned<hso_seq_result> chain_res_;   // the chain's result records
  Pinned<hso_seed_brief> chain_brief_; // ... and the briefs of the seed observation it chained behind the regular frames (one image per tracker mode)
  Pinned<hso_seq_feature> feat_rows_;  // frame feature tables on their way to / from the device
  Pinned<double> track_tables_;
  Pinned<hso_seed> act_seeds_; Pinned<hso_activate_target> act_targets_; Pinned<int32_t> act_ints_; Pinned<int32_t> act_slots_; Pinned<hso_activate_out> act_out_;   // activate_seeds()
  Pinned<hso_seed_brief> seed_brief_;
  Pinned<float> seed_px_;
  Pinned<hso_corner> det_corners_, det_fill_;   // detect(): the candidate lists of a step's new keyframes
  Pinned<hso_edgelet> det_edgelets_;
  int det_corner_cap_ = 8192;                   // corners per (frame, level) the lists hold; grows when a frame has more
};

}  // namespace engine
}  // namespace hso
