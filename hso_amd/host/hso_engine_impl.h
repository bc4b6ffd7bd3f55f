// hso_engine_impl.h — what the engine's translation units share: the per-sequence tables with their list / map operations,
// the scratch a sequence carries through a step, the worker pool.
#pragma once
#include <sched.h>
#include <pthread.h>
#include "hso_engine.h"
#include <algorithm>
#include <atomic>
#include <cassert>
#include <chrono>
#include <cmath>
#include <cstring>
#include <limits>
#include <memory>
#include <numeric>
#include <stdexcept>

namespace hso {
namespace engine {

// ------------------------------------------------------------------------------------------------ small helpers

struct DeviceFault : std::runtime_error { using std::runtime_error::runtime_error; };
// thrown by an entry point BEFORE it has touched any table (wrong image size, missing argument), or after it has put every table
// back in order (a sequence that could not start was reset and paused): the handle stays usable.  Any other exception that leaves
// a mutating entry point means half-updated tables, and the C interface poisons the handle (hso_engine_c.cpp: guarded).
struct Refused : std::invalid_argument { using std::invalid_argument::invalid_argument; };

inline double len3(const double* a) { return std::sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]); }
inline Vector3d along(const double* f, double s) { return {f[0] * s, f[1] * s, f[2] * s}; }

// the element std::nth_element leaves at floor(n / 2) (vikit getMedian: the upper median); v is permuted
template <typename T> inline T upper_median(std::vector<T>& v)
{
  auto mid = v.begin() + (std::ptrdiff_t)(v.size() / 2);
  std::nth_element(v.begin(), mid, v.end());
  return *mid;
}

inline uint8_t quality_key(const Point& p) { return (uint8_t)((p.kind << 4) | p.on); }
inline int8_t point_face(int8_t feature_type) { return feature_type == HSO_FTR_EDGELET ? kOnEdgelet : feature_type == HSO_FTR_CORNER ? kOnCorner : kOnGradient; }


// ------------------------------------------------------------------------------------------------ worker pool
// parallel-for over the sequences of a phase; the caller takes part, so a pool of zero threads is the serial engine.
// A step is a chain of ~20 short parallel phases (0.1-1 ms of work each) separated by device calls, so what a wake-up costs
// decides how well the phases scale.  A worker that finds no work can keep polling the generation counter for HSO_ENGINE_SPIN_US
// microseconds before it blocks on the condition variable — compiled out (spin_us_ = 0): on a host whose CPU time is capped (the GPU boxes of
// this project run under a 16-CPU cgroup quota) polling workers burn the quota the bookkeeping itself needs, and whole steps then
// stall for a scheduler period (measured: 10 k -> 6 k frames/s for one bank of 128, 17 k -> 5 k for three banks).  The caller
// polls for the last stragglers of a phase (microseconds).  Work is handed out through one 64-bit ticket (generation << 32 |
// next index), so a worker that is late leaving one phase can never take an index of the next.
class Pool {
public:
  explicit Pool(int n_threads)
  {
    for (int i = 0; i < n_threads; i++) th_.emplace_back([this] { loop(); });
  }
  // confine the workers to a CPU set (the device's NUMA node: Bank::pin_threads)
  void pin(const cpu_set_t& set) { for (auto& t : th_) (void)pthread_setaffinity_np(t.native_handle(), sizeof(set), &set); }
  ~Pool()
  {
    { std::lock_guard<std::mutex> lk(m_); quit_.store(true); gen_.fetch_add(1); }
    cv_.notify_all();
    for (auto& t : th_) t.join();
  }
  void run(int n, const std::function<void(int)>& fn)
  {
    if (n <= 0) return;
    if (th_.empty() || n == 1) { for (int i = 0; i < n; i++) fn(i); return; }
    // no ticket is valid while the phase's fields change: a worker still looking at the finished phase's ticket (index == its n)
    // must not see the new, larger n and take that index
    ticket_.store(~uint64_t(0));
    fn_.store(&fn, std::memory_order_relaxed); n_.store(n, std::memory_order_relaxed); err_ = nullptr;
    left_.store(n);
    const uint64_t g = gen_.load() + 1;
    ticket_.store(g << 32);
    gen_.store(g);                                   // sequentially consistent against the workers' sleepers_ / gen_ pair below
    if (sleepers_.load() > 0) { std::lock_guard<std::mutex> lk(m_); cv_.notify_all(); }
    drain(g);
    for (int spins = 0; left_.load(std::memory_order_acquire) != 0; spins++) { if (spins < 4096) relax(); else std::this_thread::yield(); }
    fn_.store(nullptr, std::memory_order_relaxed);
    if (err_) std::rethrow_exception(err_);
  }
private:
  static void relax() { __builtin_ia32_pause(); }
  void drain(uint64_t g)
  {
    for (;;) {
      uint64_t t = ticket_.load(std::memory_order_acquire);
      // n_ / fn_ are read by a worker that may be late leaving the finished phase while the caller writes the next phase's values:
      // atomics (ThreadSanitizer, profiles/r5_sanitizers.md); a stale read is harmless — the ticket's compare-exchange below only
      // succeeds on a ticket of the worker's own generation, published after the fields
      if ((t >> 32) != g || (int)(uint32_t)t >= n_.load(std::memory_order_relaxed)) return;
      if (!ticket_.compare_exchange_weak(t, t + 1, std::memory_order_acq_rel)) continue;
      const std::function<void(int)>* fn = fn_.load(std::memory_order_relaxed);
      try { (*fn)((int)(uint32_t)t); }
      catch (...) { std::lock_guard<std::mutex> lk(m_); if (!err_) err_ = std::current_exception(); }
      left_.fetch_sub(1, std::memory_order_acq_rel);
    }
  }
  void loop()
  {
    uint64_t seen = 0;
    for (;;) {
      const auto t0 = std::chrono::steady_clock::now();
      bool got = false;
      for (int k = 0;; k++) {
        if (gen_.load(std::memory_order_acquire) != seen) { got = true; break; }
        relax();
        if ((k & 255) == 255 && std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count() >= spin_us_) break;
      }
      if (!got) {
        std::unique_lock<std::mutex> lk(m_);
        sleepers_.fetch_add(1);
        cv_.wait(lk, [&] { return gen_.load() != seen; });
        sleepers_.fetch_sub(1);
      }
      if (quit_.load()) return;
      seen = gen_.load(std::memory_order_acquire);
      drain(seen);
    }
  }
  std::vector<std::thread> th_;
  std::mutex m_;
  std::condition_variable cv_;
  std::atomic<const std::function<void(int)>*> fn_{nullptr};
  std::atomic<uint64_t> ticket_{0}, gen_{0};
  std::atomic<int> left_{0}, sleepers_{0}, n_{0};
  std::atomic<bool> quit_{false};
  long spin_us_ = 0;
  std::exception_ptr err_;
};

// ------------------------------------------------------------------------------------------------ one sequence
enum Stage { kPaused = 0, kFirst = 1, kSecond = 2, kRunning = 3, kRelocalising = 4 };
enum Outcome { kNoKeyframe = 0, kKeyframe = 1, kFailure = 2 };
enum Quality { kInsufficient = 0, kBad = 1, kGood = 2 };

struct TwoView {                  // KltHomographyInit's state (src/initialization.cpp:39-223) as parallel arrays
  Id ref = kNone, prev = kNone;   // frame slots: the first frame; the frame KLT continues from
  std::vector<Vector2d> px_ref, px_cur;
  std::vector<Vector3d> f_ref, f_cur;
  std::vector<std::array<double, 3>> kind;   // gradient direction + feature type of the reference detection
  void clear() { ref = prev = kNone; px_ref.clear(); px_cur.clear(); f_ref.clear(); f_cur.clear(); kind.clear(); }
};

struct Seq {
  int index = 0, map = -1;
  const AbstractCamera* cam = nullptr;
  const Settings* cfg = nullptr;
  std::vector<Frame> frames;
  std::vector<Id> free_slots;
  std::vector<Feat> feats;
  std::vector<Point> points;
  std::vector<Seed> seeds;        // in batch order (a keyframe's seeds are appended together; compaction keeps the order)
  int n_dead_seeds = 0;
  // the seeds the convergence loop of updateSeeds has to look at (indices): a seed's test sqrt(sigma2) < z_range / thresh can only
  // change its answer when sigma2 changes, and its validity only when an observation clears it — both happen where a brief is
  // applied, which is where the seed is noted.  check_all: the indices were invalidated (the list was compacted): look at every seed once.
  std::vector<int> check_seeds;
  bool check_all = true;
  std::vector<Id> kfs;            // Map::keyframes_
  std::vector<Id> dev_kfs;        // rows of the device keyframe table (promotion order)
  std::vector<Id> candidates, temps;
  std::vector<Id> dirty_pts, dirty_obs;
  std::vector<uint8_t> pt_flag, obs_flag;
  bool kfs_dirty = false;
  bool keys_dirty = false;        // a keyframe's key points changed
  std::vector<Id> dirty_lists;    // keyframes whose feature list grew since the last flush
  std::vector<Id> dev_cands;      // the candidate list as the device holds it (entries the device deleted itself stay until the next full send)
  // handler
  int stage = kPaused, quality = kInsufficient, outcome = kNoKeyframe;
  bool want_start = false, after_init = false;
  Id last = kNone, cur = kNone, first = kNone;
  SE3 motion;
  int regular = 0, n_obs_last = 0;
  int32_t n_frames = 0, n_kfs_made = 0, batch = 0;
  std::vector<Id> local_map;
  std::vector<int> converge_hist;
  size_t n_mean_converge = 6;
  std::vector<std::pair<int, std::vector<Id>>> prior;   // frame_prior_: (batch, frames newest first), the last few batches
  // Seed::pre_frames: every seed of a keyframe starts with the same list (src/depth_filter.cpp:186-192) and every sweep of the
  // idle-time pass takes the first entry from all of them, so the list is kept once per keyframe (batch): frames newest first
  struct PreList { Id host; int32_t batch; std::vector<Id> frames; };
  std::vector<PreList> pre_lists;
  float converge_thresh = 200;
  double kf_depth_mean = 0, kf_depth_min = 0;
  TwoView init;
  std::vector<int> votes;         // scratch of the covisibility count
  std::vector<double> hist_stamp; std::vector<hso_se3> hist_pose;   // every processed frame's final pose (what a harness writes as the trajectory)
  hso_vo_status log{};
  Trace trace;

  // ---- frames
  Id new_frame()
  {
    Id fr;
    if (!free_slots.empty()) { fr = free_slots.back(); free_slots.pop_back(); }
    else { fr = (Id)frames.size(); frames.emplace_back(); }
    Frame& F = frames[fr];
    F = Frame();
    F.in_use = true;
    F.serial = n_frames++;
    F.dev_id = ((int64_t)index << 32) | (int64_t)(uint32_t)F.serial;
    return fr;
  }
  void hold(Id fr) { if (fr != kNone) frames[fr].refs++; }
  // returns true when the frame may leave the device (the caller queues the release)
  bool drop(Id fr)
  {
    if (fr == kNone) return false;
    Frame& F = frames[fr];
    if (--F.refs > 0 || F.kf_row >= 0) return false;
    F.in_use = false;
    F.loose.clear(); F.loose.shrink_to_fit();
    F.fts.clear(); F.covis.clear();
    free_slots.push_back(fr);
    return true;
  }
  size_t n_feats(const Frame& F) const { return F.kf_row >= 0 ? F.fts.size() : (size_t)F.n_fts; }
  // the seeds of one keyframe (Seed::batch_id): a contiguous run of the list
  std::pair<size_t, size_t> seed_range(int32_t batch) const
  {
    auto lo = std::lower_bound(seeds.begin(), seeds.end(), batch, [](const Seed& a, int32_t b) { return a.batch < b; });
    auto hi = std::upper_bound(lo, seeds.end(), batch, [](int32_t b, const Seed& a) { return b < a.batch; });
    return {(size_t)(lo - seeds.begin()), (size_t)(hi - seeds.begin())};
  }
  static bool seed_converged(const Seed& sd) { return std::sqrt(sd.sigma2) < sd.z_range / sd.converge; }   // src/depth_filter.cpp:411
  void list_grew(Id fr) { if (std::find(dirty_lists.begin(), dirty_lists.end(), fr) == dirty_lists.end()) dirty_lists.push_back(fr); }
  Feat& feat_of(Frame& F, size_t i) { return F.kf_row >= 0 ? feats[F.fts[i]] : F.loose[i]; }
  const Feat& feat_of(const Frame& F, size_t i) const { return F.kf_row >= 0 ? feats[F.fts[i]] : F.loose[i]; }
  Vector3d centre(const Frame& F) const { return F.T.inverse().translation(); }

  // ---- device mirror
  void touch_point(Id p) { if ((size_t)p >= pt_flag.size()) pt_flag.resize(points.size() + 64, 0); if (!pt_flag[p]) { pt_flag[p] = 1; dirty_pts.push_back(p); } }
  void touch_obs(Id f) { if ((size_t)f >= obs_flag.size()) obs_flag.resize(feats.size() + 64, 0); if (!obs_flag[f]) { obs_flag[f] = 1; dirty_obs.push_back(f); } }

  // ---- observation lists (Point::obs_: push_front, erase)
  void observe(Id p, Id f)        // Point::addFrameRef
  {
    Feat& o = feats[f];
    Point& P = points[p];
    o.next = P.head; o.linked = true;
    P.head = f; P.n_obs++;
    touch_obs(f); touch_point(p);
  }
  bool unobserve(Id p, Id frame)  // Point::deleteFrameRef: the first observation made in `frame`
  {
    Point& P = points[p];
    Id prev = kNone;
    for (Id o = P.head; o != kNone; prev = o, o = feats[o].next) {
      if (feats[o].frame != frame) continue;
      if (prev == kNone) P.head = feats[o].next; else { feats[prev].next = feats[o].next; touch_obs(prev); }
      feats[o].next = kNone; feats[o].linked = false;
      P.n_obs--;
      touch_point(p);
      return true;
    }
    return false;
  }

  // ---- Frame::key_pts_ (src/frame.cpp:121-192): the feature nearest the image centre and, per quadrant, the one farthest
  // out (largest |dx * dy|); an occupied place changes hands only for a strictly better feature
  void offer_key(Frame& F, Id f)
  {
    const int cu = cam->width() / 2, cv = cam->height() / 2;
    const Feat& n = feats[f];
    const double dx = n.px[0] - cu, dy = n.px[1] - cv;
    auto cheb = [&](Id g) { return std::max(std::fabs(feats[g].px[0] - cu), std::fabs(feats[g].px[1] - cv)); };
    keys_dirty = true;
    if (F.key[0] == kNone || std::max(std::fabs(dx), std::fabs(dy)) < cheb(F.key[0])) F.key[0] = f;
    const int quadrant = n.px[0] >= cu ? (n.px[1] >= cv ? 1 : 2) : (n.px[1] >= cv ? 3 : 4);
    const double sx = (quadrant == 1 || quadrant == 2) ? 1.0 : -1.0, sy = (quadrant == 1 || quadrant == 3) ? 1.0 : -1.0;
    Id& place = F.key[quadrant];
    if (place == kNone) place = f;
    else {
      const Feat& h = feats[place];
      if ((sx * dx) * (sy * dy) > (sx * (h.px[0] - cu)) * (sy * (h.px[1] - cv))) place = f;
    }
  }
  void refresh_keys(Frame& F)     // Frame::setKeyPoints
  {
    keys_dirty = true;
    for (Id& k : F.key) if (k != kNone && feats[k].point == kNone) k = kNone;
    for (Id f : F.fts) if (feats[f].point != kNone) offer_key(F, f);
  }
  void lose_key(Id f)             // Frame::removeKeyPoint
  {
    Frame& F = frames[feats[f].frame];
    bool was = false;
    for (Id& k : F.key) if (k == f) { k = kNone; was = true; }
    if (was) refresh_keys(F);
  }
  bool sees(const Frame& F, const double* xyz_w) const   // Frame::isVisible
  {
    const Vector3d p = F.T * Vector3d{xyz_w[0], xyz_w[1], xyz_w[2]};
    if (p[2] < 0.0) return false;
    const Vector2d px = cam->world2cam(p);
    return px[0] >= 0.0 && px[1] >= 0.0 && px[0] < cam->width() && px[1] < cam->height();
  }

  // ---- Map (src/map.cpp:102-188)
  void erase_point(Id p)          // Map::safeDeletePoint
  {
    Point& P = points[p];
    for (Id o = P.head; o != kNone;) {
      const Id nx = feats[o].next;
      feats[o].point = kNone; feats[o].next = kNone; feats[o].linked = false;
      touch_obs(o);
      lose_key(o);
      o = nx;
    }
    P.head = kNone; P.n_obs = 0;
    P.kind = kPtDeleted;
    touch_point(p);
  }
  void detach(Id frame, Id f)     // Map::removePtFrameRef
  {
    const Id p = feats[f].point;
    if (p == kNone) return;
    feats[f].point = kNone;
    touch_obs(f);
    if (points[p].n_obs <= 2) { erase_point(p); return; }
    unobserve(p, frame);
    lose_key(f);
  }
  void place_in_host(Id p)        // pos_ = T_host^-1 * (f / idist)
  {
    Point& P = points[p];
    const Vector3d w = frames[P.host_frame].T.inverse() * along(P.host_f, 1.0 / P.idist);
    P.pos[0] = w[0]; P.pos[1] = w[1]; P.pos[2] = w[2];
    touch_point(p);
  }
  bool erase_candidate(Id p)      // MapPointCandidates::deleteCandidatePoint: the host feature lives in no frame's list yet
  {
    auto it = std::find(candidates.begin(), candidates.end(), p);
    if (it == candidates.end()) return false;
    candidates.erase(it);
    Point& P = points[p];
    if (P.host != kNone) { feats[P.host].point = kNone; feats[P.host].linked = false; feats[P.host].next = kNone; touch_obs(P.host); }
    P.head = kNone; P.n_obs = 0;
    P.kind = kPtDeleted;
    touch_point(p);
    return true;
  }
  // Map::safeDeleteTempPoint: what becomes of a temporary point once its seed has finished
  void retire_temp(Id p)
  {
    Point& P = points[p];
    if (P.seed_state == -1) {                   // the seed was dropped
      if (P.bad) { erase_point(p); return; }
      place_in_host(p);
      P.dev_reset = 3;                               // n_failed_reproj_ = n_succeeded_reproj_ = 0
      if (P.n_obs == 1) { P.kind = kPtCandidate; candidates.push_back(p); }
      else { P.kind = kPtUnknown; frames[feats[P.host].frame].fts.push_back(P.host); list_grew(feats[P.host].frame); }
      return;
    }
    // the seed converged into a point of its own, which took over the host feature; every other observation of the temporary
    // point is released
    const Id heir = feats[P.host].point;
    for (Id o = P.head; o != kNone;) {
      const Id nx = feats[o].next;
      if (feats[o].point != heir) { feats[o].point = kNone; feats[o].next = kNone; feats[o].linked = false; lose_key(o); }
      else if (o != P.host) { feats[o].next = kNone; feats[o].linked = false; }
      touch_obs(o);
      o = nx;
    }
    P.head = kNone; P.n_obs = 0;
    P.kind = kPtDeleted;
    touch_point(p);
  }
  void closest_keyframes(const Frame& F, std::vector<std::pair<double, Id>>& out) const   // Map::getCloseKeyframes
  {
    for (Id k : kfs) {
      const Frame& K = frames[k];
      for (Id key : K.key) {
        if (key == kNone || feats[key].point == kNone) continue;
        if (!sees(F, points[feats[key].point].pos)) continue;
        const double d[3] = {F.T.v.t[0] - K.T.v.t[0], F.T.v.t[1] - K.T.v.t[1], F.T.v.t[2] - K.T.v.t[2]};
        out.emplace_back(len3(d), k);
        break;
      }
    }
  }
  Id new_point(const Vector3d& pos, Id host_feat, double idist, int8_t kind)
  {
    points.emplace_back();
    const Id p = (Id)points.size() - 1;
    Point& P = points[p];
    P.pos[0] = pos[0]; P.pos[1] = pos[1]; P.pos[2] = pos[2];
    P.idist = idist; P.host = host_feat; P.kind = kind;
    P.host_frame = feats[host_feat].frame;
    P.host_f[0] = feats[host_feat].f[0]; P.host_f[1] = feats[host_feat].f[1]; P.host_f[2] = feats[host_feat].f[2];
    P.on = point_face(feats[host_feat].type);
    return p;
  }
  void reset_tables()
  {
    frames.clear(); free_slots.clear(); feats.clear(); points.clear(); seeds.clear(); n_dead_seeds = 0; check_seeds.clear(); check_all = true;
    kfs.clear(); dev_kfs.clear(); candidates.clear(); temps.clear(); dirty_pts.clear(); dirty_obs.clear(); pt_flag.clear(); obs_flag.clear();
    dirty_lists.clear(); dev_cands.clear(); keys_dirty = true;
    kfs_dirty = true; local_map.clear(); converge_hist.clear(); prior.clear(); pre_lists.clear(); init.clear(); hist_stamp.clear(); hist_pose.clear();
    last = cur = first = kNone;
  }
};

// what a sequence carries through one step
struct StepData {
  bool active = false, tracked = false, ok = false, make_kf = false, seed_path = false, relocalised = false;
  int stage0 = 0;                              // the sequence's stage when the step began
  SE3 reloc_pose;
  hso_pose_result pose{};
  std::vector<uint8_t> pose_mask;
  int n_core = 0;
  std::vector<int64_t> released;               // device frames to release (collected on pool threads)
  Id ref = kNone;                              // the frame the tracker aligns against
  int inverse = 1;
  hso_track_result track{};
  std::vector<Id> visit;                       // overlap keyframes in visiting order
  std::vector<Id> temps_listed;                // the temporary points the frame lists
  hso_seq_result res{};                        // the chain's result record
  bool host_pose = false;                      // the pose was optimised again over host tables (the seed branch)
  bool seeds_observed = false;                 // the chain observed the sequence's seeds behind this frame ...
  const hso_seed_brief* chain_brief = nullptr; // ... and this is where their briefs are (all slots of the table)
  size_t n_inliers = 0;
  double depth_mean = 0, depth_min = 0, dist_mean = 0;
  // keyframe: the local BA window
  std::vector<Id> ba_frames; std::vector<uint8_t> ba_fixed;        // the core keyframes (vertex order)
  std::vector<Id> ba_points; std::vector<double> ba_state;          // out: the window's points, their idist + pos after the optimisation
  std::vector<Id> ba_culled;                                        // out: observations to remove
  hso_seq_ba_result ba{};
  hso_ba_result ba_res{};
  float huber_corner = 0, huber_edge = 0;
  int ba_iters = 0;
  std::vector<Id> moved_kfs;                   // keyframes whose pose local BA changed
  // seeds
  std::vector<int> conv;                       // indices of converged seeds
  std::vector<Id> act_frames; std::vector<int32_t> act_pair_frame;   // the converged seeds' target frames, named once; per (seed, target) pair its index there
  std::vector<hso_keypoint> occupied;          // FeatureExtractor::setGridOccpuancy keys of this keyframe's observation
  std::vector<int32_t> erase_slots;
  std::vector<hso_seed> new_seeds;
};

}  // namespace engine
}  // namespace hso
