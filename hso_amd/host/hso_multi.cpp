// hso_multi.cpp — N independent sequences over ONE device context, advancing in lockstep (include/hso_vo.h: hso_vo_multi_*).
//
// BASELINE.json north_star: "independent sequences shard trivially ... batched"; configs[4] runs 8 sequences, which the
// reference does one after the other (test/euroc_batch.sh:9-18).  Here every sequence is a FrameHandlerMono of its own (the
// reference's class, hso_vo.h: own map, depth filter, counters) on a worker thread of its own, all of them sharing one
// hso_gpu_ctx.  hso_vo_multi_add_images hands each worker its next image; the workers run addImage exactly as the
// single-sequence driver does — and wherever processFrame reaches a device call, the call does not go to the C-ABI but to a
// rendezvous (api::Router, hso_trace.h): when every sequence still inside its frame has arrived at a call, the waiting calls
// leave as ONE batched C-ABI call per kind —
//     Frame construction        hso_gpu_frame_upload_batch
//     CoarseTracker::run        hso_gpu_coarse_track_batch   (N jobs: up to one XCD per sequence, hso_tracker_coop.hip)
//     Reprojector::reprojectMap hso_gpu_reproject_match_multi (tables concatenated, indices rebased) / hso_gpu_align_multi
//     pose optimisation         hso_gpu_pose_optimize_batch
//     DepthFilter::updateSeeds  hso_gpu_seed_observe_multi, activation hso_gpu_seed_activate_multi
//     LocalBundleAdjustment     hso_gpu_ba_optimize_multi
// — and calls without a multi-sequence form (detection, the seed branch of the reprojector, the BA Huber deltas) run one after the
// other in arrival order.  Sequences need not be in the same state: one may insert a keyframe (detection, activation, BA) while the
// others only track; the rendezvous groups whatever kinds are waiting.  The batched entry points return, per item, exactly the
// bytes of the per-item call (tests/test_align.py, test_seed.py, test_activate.py, test_ba.py, test_pose.py, test_track_coop_gpu.py),
// so a sequence run here equals the same sequence run alone through hso_vo_* bit for bit (tests/test_multi_gpu.py).
// No control-plane logic lives here: which calls happen, in which order and on which data is FrameHandlerMono's business.
#include <chrono>
#include <condition_variable>
#include <cstring>
#include <new>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>
#include "hso_vo.h"
#include "hso_api.h"
#include "../../include/hso_vo.h"

namespace {

enum Kind { K_UPLOAD, K_RELEASE, K_TRACK, K_REPROJECT, K_ALIGN, K_POSE, K_SEED, K_ACTIVATE, K_BA, K_SOLO, K_COUNT };

struct Req {
  Kind kind;
  int rc = 0;
  bool done = false;
  std::string* err = nullptr;      // the submitting sequence's error slot
  // arguments (only the fields of `kind` are set)
  const hso_camera* cam = nullptr;
  int64_t id = 0; const uint8_t* img = nullptr; int w = 0, h = 0; hso_frame_stats* st = nullptr;
  const hso_track_params* tp = nullptr; const hso_track_job* tj = nullptr; hso_track_result* tr = nullptr;
  const hso_se3* T = nullptr; double exposure = 0, px_error_angle = 0; int cur_kf_id = 0;
  const hso_kf* kfs = nullptr; int n_kfs = 0; const hso_map_point* pts = nullptr; int n_pts = 0; const hso_obs* obs = nullptr; int n_obs = 0;
  int cell_size = 0, grid_n_cols = 0; hso_reproj_point* proj = nullptr; hso_align_out* match = nullptr;
  const hso_align_job* ajobs = nullptr; int n = 0;
  const hso_pose_job* pj = nullptr; hso_pose_result* pr = nullptr; uint8_t* mask = nullptr;
  const hso_seed* seeds = nullptr; hso_seed_out* sout = nullptr;
  const int32_t* begin = nullptr; const hso_activate_target* targets = nullptr; int n_mean = 0; hso_activate_out* aout = nullptr;
  hso_ba_problem ba{};
  int (*fn)(void*) = nullptr; void* arg = nullptr;
};

// A grow-only array in page-locked memory of the context (hso_gpu_host_alloc): the merged tables of a batched call are DMA sources
// / targets as they are (the library stages pageable memory through a copy of its own).
template <typename T> struct PinnedVec {
  hso_gpu_ctx** ctx; T* p = nullptr; size_t n = 0, cap = 0;
  explicit PinnedVec(hso_gpu_ctx** c) : ctx(c) {}
  PinnedVec(const PinnedVec&) = delete;
  PinnedVec& operator=(const PinnedVec&) = delete;
  ~PinnedVec() { }                         // freed with the context (hso_gpu_destroy releases every host allocation)
  void reserve(size_t want)
  {
    if (want <= cap) return;
    const size_t ncap = want + want / 2 + 1024;
    void* q = nullptr;
    if (hso_gpu_host_alloc(*ctx, ncap * sizeof(T), &q) < 0) throw std::bad_alloc();
    if (n) memcpy(q, p, n * sizeof(T));
    if (p) hso_gpu_host_free(*ctx, p);
    p = static_cast<T*>(q); cap = ncap;
  }
  void clear() { n = 0; }
  size_t size() const { return n; }
  T* data() { return p; }
  T& operator[](size_t i) { return p[i]; }
  void push_back(const T& v) { if (n == cap) reserve(n + 1); p[n++] = v; }
  void append(const T* src, size_t k) { reserve(n + k); if (k) memcpy(p + n, src, k * sizeof(T)); n += k; }
  void resize(size_t k) { reserve(k); n = k; }
};

struct Batcher {
  hso_gpu_ctx* ctx = nullptr;
  std::mutex m;
  std::condition_variable cv;
  std::vector<Req*> pending;
  int active = 0;                      // sequences currently inside a frame (or inside set_first_frame)
  long long n_calls[K_COUNT] = {0};    // batched C-ABI calls issued, per kind
  long long n_items[K_COUNT] = {0};    // requests they carried
  std::chrono::steady_clock::time_point t_last_flush = std::chrono::steady_clock::now();

  // Every batched C-ABI call is issued by ONE thread: the API caller's own (the thread inside hso_vo_multi_add_images etc. serves
  // the rendezvous while the sequences run, serve() below) — the thread that created the context and its stream.
  std::condition_variable cv_dev;
  bool flush_wanted = false;
  int tasks_left = 0;                  // sequence tasks of the current step that have not finished yet
  std::vector<hso_reproj_frame> m_fr;                                                  // run_reproject: the merged tables
  PinnedVec<hso_kf> m_kfs{&ctx}; PinnedVec<hso_map_point> m_pts{&ctx}; PinnedVec<hso_obs> m_obs{&ctx};
  PinnedVec<hso_reproj_point> m_proj{&ctx}; PinnedVec<hso_align_out> m_match{&ctx};
  std::vector<int64_t> release_queue;  // frames whose owners have gone (SeqRouter::frame_release)
  void drain_releases()                // lock held
  {
    for (int64_t id : release_queue) { (void)hso_gpu_frame_release(ctx, id); n_calls[K_RELEASE]++; n_items[K_RELEASE]++; }
    release_queue.clear();
  }
  void serve()
  {
    std::unique_lock<std::mutex> lk(m);
    for (;;) {
      cv_dev.wait(lk, [&] { return flush_wanted || tasks_left == 0; });
      if (flush_wanted) { flush_wanted = false; if (!pending.empty()) flush(); continue; }   // with the lock held: arrivals wait until served
      break;
    }
    drain_releases();
  }
  void enter() { std::lock_guard<std::mutex> lk(m); ++active; }
  void leave()
  {
    std::unique_lock<std::mutex> lk(m);
    --active; --tasks_left;
    if (!pending.empty() && (int)pending.size() >= active) flush_wanted = true;
    if (flush_wanted || tasks_left == 0) cv_dev.notify_one();
  }
  int submit(Req& r)
  {
    std::unique_lock<std::mutex> lk(m);
    pending.push_back(&r);
    if ((int)pending.size() >= active) { flush_wanted = true; cv_dev.notify_one(); }   // the last one in completes the rendezvous
    cv.wait(lk, [&] { return r.done; });
    return r.rc;
  }
  void fail(std::vector<Req*>& v, int rc) { for (Req* r : v) { r->rc = rc; if (rc < 0 && r->err) *r->err = hso_gpu_last_error(ctx); } }

  // called with the lock held
  void flush()
  {
    std::vector<Req*> by[K_COUNT];
    for (Req* r : pending) by[r->kind].push_back(r);
    for (int k = 0; k < K_COUNT; k++) if (!by[k].empty()) { n_items[k] += (long long)by[k].size(); }
    static const bool timing = getenv("HSO_MULTI_TIMING") != nullptr;      // developer probe: wall time of every batched call
    auto t_prev = std::chrono::steady_clock::now();
    auto lap = [&](const char* what, size_t n) {
      if (!timing || n == 0) return;
      const auto t = std::chrono::steady_clock::now();
      fprintf(stderr, "[hso_multi] %-10s %3zu requests %8.3f ms (gap since previous flush %.3f ms)\n", what, n,
              std::chrono::duration<double, std::milli>(t - t_prev).count(), std::chrono::duration<double, std::milli>(t_prev - t_last_flush).count());
      t_prev = t;
    };
    drain_releases();
    run_upload(by[K_UPLOAD]); lap("upload", by[K_UPLOAD].size());
    for (Req* r : by[K_RELEASE]) { r->rc = hso_gpu_frame_release(ctx, r->id); n_calls[K_RELEASE]++; }
    lap("release", by[K_RELEASE].size());
    run_track(by[K_TRACK]); lap("track", by[K_TRACK].size());
    run_reproject(by[K_REPROJECT]); lap("reproject", by[K_REPROJECT].size());
    run_align(by[K_ALIGN]); lap("align", by[K_ALIGN].size());
    run_pose(by[K_POSE]); lap("pose", by[K_POSE].size());
    run_seed(by[K_SEED]); lap("seed", by[K_SEED].size());
    run_activate(by[K_ACTIVATE]); lap("activate", by[K_ACTIVATE].size());
    run_ba(by[K_BA]); lap("ba", by[K_BA].size());
    for (Req* r : by[K_SOLO]) { r->rc = r->fn(r->arg); if (r->rc < 0 && r->err) *r->err = hso_gpu_last_error(ctx); n_calls[K_SOLO]++; }
    lap("solo", by[K_SOLO].size());
    t_last_flush = std::chrono::steady_clock::now();
    for (Req* r : pending) r->done = true;
    pending.clear();
    cv.notify_all();
  }

  void run_upload(std::vector<Req*>& v)
  {
    if (v.empty()) return;
    bool same = true;
    for (Req* r : v) same = same && r->w == v[0]->w && r->h == v[0]->h;
    if (!same) { for (Req* r : v) { std::vector<Req*> one{r}; int rc = hso_gpu_frame_upload(ctx, r->id, r->img, r->w, r->h, 0, r->st); fail(one, rc); n_calls[K_UPLOAD]++; } return; }
    std::vector<int64_t> ids; std::vector<const uint8_t*> imgs; std::vector<hso_frame_stats> st(v.size());
    for (Req* r : v) { ids.push_back(r->id); imgs.push_back(r->img); }
    const int rc = hso_gpu_frame_upload_batch(ctx, ids.data(), imgs.data(), (int)v.size(), v[0]->w, v[0]->h, 0, st.data());
    n_calls[K_UPLOAD]++;
    fail(v, rc);
    if (rc >= 0) for (size_t i = 0; i < v.size(); i++) if (v[i]->st) *v[i]->st = st[i];
  }

  void run_track(std::vector<Req*>& v)
  {
    // one batch per distinct parameter set (a relocalising sequence tracks other levels than the rest)
    std::vector<char> used(v.size(), 0);
    for (size_t a = 0; a < v.size(); a++) {
      if (used[a]) continue;
      std::vector<Req*> g;
      for (size_t b = a; b < v.size(); b++)
        if (!used[b] && !memcmp(v[b]->tp, v[a]->tp, sizeof(hso_track_params)) && !memcmp(v[b]->cam, v[a]->cam, sizeof(hso_camera))) { used[b] = 1; g.push_back(v[b]); }
      std::vector<hso_track_job> jobs; std::vector<hso_track_result> res(g.size());
      for (Req* r : g) jobs.push_back(*r->tj);
      const int rc = hso_gpu_coarse_track_batch(ctx, g[0]->cam, g[0]->tp, jobs.data(), (int)g.size(), res.data());
      n_calls[K_TRACK]++;
      fail(g, rc);
      if (rc >= 0) for (size_t i = 0; i < g.size(); i++) *g[i]->tr = res[i];
    }
  }

  void run_reproject(std::vector<Req*>& v)
  {
    if (v.empty()) return;
    // hso_gpu_reproject_match_multi: one frame record per sequence over concatenated tables; a point's host_kf and an
    // observation's kf stay relative to the frame's kf_begin, obs_begin indexes the concatenated observation table
    bool same = true;
    for (Req* r : v) same = same && r->cell_size == v[0]->cell_size && r->grid_n_cols == v[0]->grid_n_cols && !memcmp(r->cam, v[0]->cam, sizeof(hso_camera));
    if (!same || v.size() == 1) {
      for (Req* r : v) {
        std::vector<Req*> one{r};
        fail(one, hso_gpu_reproject_match(ctx, r->cam, r->id, r->T, r->exposure, r->cur_kf_id, r->kfs, r->n_kfs, r->pts, r->n_pts, r->obs, r->n_obs,
                                          r->cell_size, r->grid_n_cols, r->proj, r->match));
        n_calls[K_REPROJECT]++;
      }
      return;
    }
    // the merged tables keep their storage between steps (members, page-locked): a fresh multi-megabyte vector per step costs its
    // page faults on the way in and an munmap on the way out
    std::vector<hso_reproj_frame>& fr = m_fr; PinnedVec<hso_kf>& kfs = m_kfs; PinnedVec<hso_map_point>& pts = m_pts; PinnedVec<hso_obs>& obs = m_obs;
    PinnedVec<hso_reproj_point>& proj = m_proj; PinnedVec<hso_align_out>& match = m_match;
    fr.assign(v.size(), hso_reproj_frame{}); kfs.clear(); pts.clear(); obs.clear();
    try {
    for (size_t i = 0; i < v.size(); i++) {
      Req* r = v[i];
      hso_reproj_frame& f = fr[i];
      f.cur_frame_id = r->id; f.T_cur_w = *r->T; f.cur_exposure_time = r->exposure; f.cur_keyframe_id = r->cur_kf_id;
      f.kf_begin = (int)kfs.size(); f.kf_count = r->n_kfs; f.point_begin = (int)pts.size(); f.point_count = r->n_pts; f.pad_ = 0;
      const int ob = (int)obs.size();
      kfs.append(r->kfs, (size_t)r->n_kfs);
      pts.reserve(pts.size() + (size_t)r->n_pts);
      for (int k = 0; k < r->n_pts; k++) { hso_map_point p = r->pts[k]; p.obs_begin += ob; pts.push_back(p); }
      if (r->n_obs > 0) obs.append(r->obs, (size_t)r->n_obs);
    }
    proj.resize(pts.size() ? pts.size() : 1); match.resize(pts.size() ? pts.size() : 1);
    } catch (const std::bad_alloc&) { fail(v, HSO_E_NOMEM); return; }
    const int rc = hso_gpu_reproject_match_multi(ctx, v[0]->cam, fr.data(), (int)fr.size(), kfs.data(), (int)kfs.size(), pts.data(), (int)pts.size(),
                                                 obs.data(), (int)obs.size(), v[0]->cell_size, v[0]->grid_n_cols, proj.data(), match.data());
    n_calls[K_REPROJECT]++;
    fail(v, rc);
    if (rc < 0) return;
    for (size_t i = 0; i < v.size(); i++) {
      Req* r = v[i];
      const int ob = r->n_pts > 0 ? pts[fr[i].point_begin].obs_begin - r->pts[0].obs_begin : 0;   // this sequence's offset into obs
      for (int k = 0; k < r->n_pts; k++) {
        hso_reproj_point p = proj[fr[i].point_begin + k];
        if (p.ref_obs >= 0) p.ref_obs -= ob;
        r->proj[k] = p;
        r->match[k] = match[fr[i].point_begin + k];
      }
    }
  }

  void run_align(std::vector<Req*>& v)
  {
    if (v.empty()) return;
    std::vector<int64_t> cur; std::vector<hso_align_job> jobs;
    for (Req* r : v) for (int k = 0; k < r->n; k++) { cur.push_back(r->id); jobs.push_back(r->ajobs[k]); }
    std::vector<hso_align_out> out(jobs.size() ? jobs.size() : 1);
    const int rc = jobs.empty() ? 0 : hso_gpu_align_multi(ctx, v[0]->cam, cur.data(), jobs.data(), (int)jobs.size(), out.data());
    n_calls[K_ALIGN]++;
    fail(v, rc);
    if (rc < 0) return;
    size_t o = 0;
    for (Req* r : v) { for (int k = 0; k < r->n; k++) r->match[k] = out[o + k]; o += r->n; }
  }

  void run_pose(std::vector<Req*>& v)
  {
    if (v.empty()) return;
    std::vector<hso_pose_job> jobs; std::vector<hso_pose_result> res(v.size()); std::vector<uint8_t*> masks;
    for (Req* r : v) { jobs.push_back(*r->pj); masks.push_back(r->mask); }
    const int rc = hso_gpu_pose_optimize_batch(ctx, v[0]->cam, jobs.data(), (int)v.size(), res.data(), masks.data());
    n_calls[K_POSE]++;
    fail(v, rc);
    if (rc >= 0) for (size_t i = 0; i < v.size(); i++) *v[i]->pr = res[i];
  }

  void run_seed(std::vector<Req*>& v)
  {
    if (v.empty()) return;
    // one launch per distinct pixel error angle (a function of the camera: one value when the sequences share it)
    std::vector<char> used(v.size(), 0);
    for (size_t a = 0; a < v.size(); a++) {
      if (used[a]) continue;
      std::vector<Req*> g;
      for (size_t b = a; b < v.size(); b++) if (!used[b] && v[b]->px_error_angle == v[a]->px_error_angle) { used[b] = 1; g.push_back(v[b]); }
      std::vector<hso_seed_frame> fr(g.size()); std::vector<int32_t> sf; std::vector<hso_seed> seeds;
      for (size_t i = 0; i < g.size(); i++) {
        fr[i].frame_id = g[i]->id; fr[i].T_f_w = *g[i]->T; fr[i].exposure_time = g[i]->exposure;
        seeds.insert(seeds.end(), g[i]->seeds, g[i]->seeds + g[i]->n);
        sf.insert(sf.end(), (size_t)g[i]->n, (int32_t)i);
      }
      std::vector<hso_seed_out> out(seeds.size() ? seeds.size() : 1);
      const int rc = seeds.empty() ? 0 : hso_gpu_seed_observe_multi(ctx, g[0]->cam, fr.data(), (int)fr.size(), sf.data(), g[0]->px_error_angle, seeds.data(),
                                                                   (int)seeds.size(), out.data());
      n_calls[K_SEED]++;
      fail(g, rc);
      if (rc < 0) continue;
      size_t o = 0;
      for (Req* r : g) { for (int k = 0; k < r->n; k++) r->sout[k] = out[o + k]; o += r->n; }
    }
  }

  void run_activate(std::vector<Req*>& v)
  {
    if (v.empty()) return;
    std::vector<hso_seed> seeds; std::vector<int32_t> begin{0}; std::vector<hso_activate_target> tg; std::vector<int32_t> nm;
    for (Req* r : v) {
      seeds.insert(seeds.end(), r->seeds, r->seeds + r->n);
      for (int k = 0; k < r->n; k++) { begin.push_back((int32_t)tg.size() + r->begin[k + 1]); nm.push_back(r->n_mean); }
      tg.insert(tg.end(), r->targets, r->targets + r->begin[r->n]);
    }
    std::vector<hso_activate_out> out(seeds.size() ? seeds.size() : 1);
    const int rc = seeds.empty() ? 0 : hso_gpu_seed_activate_multi(ctx, v[0]->cam, seeds.data(), (int)seeds.size(), begin.data(), tg.data(), nm.data(),
                                                                  out.data(), nullptr);
    n_calls[K_ACTIVATE]++;
    fail(v, rc);
    if (rc < 0) return;
    size_t o = 0;
    for (Req* r : v) { for (int k = 0; k < r->n; k++) r->aout[k] = out[o + k]; o += r->n; }
  }

  void run_ba(std::vector<Req*>& v)
  {
    if (v.empty()) return;
    std::vector<hso_ba_problem> pr;
    for (Req* r : v) pr.push_back(r->ba);
    const int rc = hso_gpu_ba_optimize_multi(ctx, pr.data(), (int)pr.size());
    n_calls[K_BA]++;
    fail(v, rc);
  }
};

// a sequence thread's view of the rendezvous
struct SeqRouter : hso::api::Router {
  Batcher* B;
  std::string err;
  explicit SeqRouter(Batcher* b) : B(b) {}
  Req make(Kind k) { Req r; r.kind = k; r.err = &err; return r; }
  int frame_upload(int64_t id, const uint8_t* img, int w, int h, hso_frame_stats* st) override
  { Req r = make(K_UPLOAD); r.id = id; r.img = img; r.w = w; r.h = h; r.st = st; return B->submit(r); }
  int frame_release(int64_t id) override
  {
    // frames are released from destructors on the sequence threads; the device call itself is left to the serving thread (every
    // HIP call of the driver comes from one thread), which drains the list with its next batch or when the step ends
    std::lock_guard<std::mutex> lk(B->m);
    B->release_queue.push_back(id);
    return HSO_OK;
  }
  int coarse_track(const hso_camera* cam, const hso_track_params* p, const hso_track_job* job, hso_track_result* res) override
  { Req r = make(K_TRACK); r.cam = cam; r.tp = p; r.tj = job; r.tr = res; return B->submit(r); }
  int reproject_match(const hso_camera* cam, int64_t cur_id, const hso_se3* T, double exposure, int cur_kf_id, const hso_kf* kfs, int n_kfs,
                      const hso_map_point* pts, int n_pts, const hso_obs* obs, int n_obs, int cell_size, int grid_n_cols, hso_reproj_point* proj,
                      hso_align_out* match) override
  {
    Req r = make(K_REPROJECT);
    r.cam = cam; r.id = cur_id; r.T = T; r.exposure = exposure; r.cur_kf_id = cur_kf_id; r.kfs = kfs; r.n_kfs = n_kfs; r.pts = pts; r.n_pts = n_pts;
    r.obs = obs; r.n_obs = n_obs; r.cell_size = cell_size; r.grid_n_cols = grid_n_cols; r.proj = proj; r.match = match;
    return B->submit(r);
  }
  int align_batch(const hso_camera* cam, int64_t cur_id, const hso_align_job* jobs, int n, hso_align_out* out) override
  { Req r = make(K_ALIGN); r.cam = cam; r.id = cur_id; r.ajobs = jobs; r.n = n; r.match = out; return B->submit(r); }
  int pose_optimize(const hso_camera* cam, const hso_pose_job* job, hso_pose_result* res, uint8_t* mask) override
  { Req r = make(K_POSE); r.cam = cam; r.pj = job; r.pr = res; r.mask = mask; return B->submit(r); }
  int seed_observe(const hso_camera* cam, int64_t cur_id, const hso_se3* T, double exposure, double pea, const hso_seed* seeds, int n,
                   hso_seed_out* out) override
  { Req r = make(K_SEED); r.cam = cam; r.id = cur_id; r.T = T; r.exposure = exposure; r.px_error_angle = pea; r.seeds = seeds; r.n = n; r.sout = out; return B->submit(r); }
  int seed_activate(const hso_camera* cam, const hso_seed* seeds, int n, const int32_t* begin, const hso_activate_target* targets, int n_mean,
                    hso_activate_out* out) override
  { Req r = make(K_ACTIVATE); r.cam = cam; r.seeds = seeds; r.n = n; r.begin = begin; r.targets = targets; r.n_mean = n_mean; r.aout = out; return B->submit(r); }
  int ba_optimize(hso_se3* poses, const uint8_t* fixed, int n_poses, double* idist, int n_points, const hso_ba_edge* edges, int n_edges, double hc,
                  double he, int n_iter, double* chi2, hso_ba_result* res) override
  {
    Req r = make(K_BA);
    r.ba.poses_f_w = poses; r.ba.pose_fixed = fixed; r.ba.idist = idist; r.ba.edges = edges; r.ba.edge_chi2_out = chi2; r.ba.result = res;
    r.ba.n_poses = n_poses; r.ba.n_points = n_points; r.ba.n_edges = n_edges; r.ba.n_iter = n_iter; r.ba.huber_corner = hc; r.ba.huber_edge = he;
    return B->submit(r);
  }
  int solo(int (*fn)(void*), void* arg) override { Req r = make(K_SOLO); r.fn = fn; r.arg = arg; return B->submit(r); }
  const char* last_error() override { return err.c_str(); }
};

// one sequence: its worker thread runs every call that touches the sequence's state, so the thread-local counters of the
// driver (Frame::frame_counter_, keyFrameCounter_, Point::point_counter_, Seed::batch_counter) belong to it alone
struct Seq {
  hso_vo* vo = nullptr;
  SeqRouter* router = nullptr;
  std::thread th;
  std::mutex m;
  std::condition_variable cv;
  std::function<int()> task;
  bool has_task = false, quit = false, task_done = false;
  int rc = 0;
};

}  // namespace

struct hso_vo_multi {
  hso_gpu_ctx* ctx = nullptr;
  Batcher B;
  std::vector<Seq*> seq;
  std::string err;
};

namespace {

void worker(hso_vo_multi* M, int k)
{
  Seq* S = M->seq[k];
  hso::api::router() = S->router;
  // this sequence's counters: ids of different sequences must not collide inside the shared context
  hso::Frame::id_base_ = k << 24;
  hso::Frame::frame_counter_ = hso::Frame::id_base_; hso::Frame::keyFrameCounter_ = 0; hso::Point::point_counter_ = 0; hso::Seed::batch_counter = 0;
  for (;;) {
    std::function<int()> t;
    {
      std::unique_lock<std::mutex> lk(S->m);
      S->cv.wait(lk, [&] { return S->has_task || S->quit; });
      if (S->quit && !S->has_task) break;
      t = S->task; S->has_task = false;
    }
    const int rc = t();
    M->B.leave();
    {
      std::lock_guard<std::mutex> lk(S->m);
      S->rc = rc; S->task_done = true;
    }
    S->cv.notify_all();
  }
  hso::api::router() = nullptr;
}

// run one task per sequence (a null function = the sequence sits this step out) in lockstep; returns the first failure
int step(hso_vo_multi* M, const std::vector<std::function<int()>>& tasks)
{
  int n_run = 0;
  for (auto& t : tasks) if (t) ++n_run;
  { std::lock_guard<std::mutex> lk(M->B.m); M->B.active = n_run; M->B.tasks_left = n_run; }
  for (size_t k = 0; k < tasks.size(); k++) {
    if (!tasks[k]) continue;
    Seq* S = M->seq[k];
    { std::lock_guard<std::mutex> lk(S->m); S->task = tasks[k]; S->has_task = true; S->task_done = false; }
    S->cv.notify_all();
  }
  M->B.serve();                         // this thread issues the batched device calls until every sequence has finished its task
  int rc = HSO_OK;
  for (size_t k = 0; k < tasks.size(); k++) {
    if (!tasks[k]) continue;
    Seq* S = M->seq[k];
    std::unique_lock<std::mutex> lk(S->m);
    S->cv.wait(lk, [&] { return S->task_done; });
    if (S->rc < 0 && rc >= 0) { rc = S->rc; M->err = "sequence " + std::to_string(k) + ": " + hso_vo_last_error(S->vo); }
  }
  return rc;
}

}  // namespace

hso_vo* hso_vo_create_shared(hso_gpu_ctx* ctx, const hso_camera* cam, int max_fts);   // hso_vo.cpp
void hso_vo_destroy_shared(hso_vo* v);

extern "C" {

int hso_vo_multi_create(hso_vo_multi** out, const hso_camera* cam, int max_fts, int n_sequences, int device)
{
  if (!out || !cam || max_fts <= 0 || n_sequences <= 0 || n_sequences > 127) return HSO_E_INVALID;
  *out = nullptr;
  hso_gpu_ctx* ctx = nullptr;
  const int rc = hso_gpu_create(&ctx, device, nullptr);
  if (rc < 0) return rc;
  hso_vo_multi* M = new hso_vo_multi();
  M->ctx = ctx; M->B.ctx = ctx;
  M->seq.resize(n_sequences);
  for (int k = 0; k < n_sequences; k++) { M->seq[k] = new Seq(); M->seq[k]->router = new SeqRouter(&M->B); }
  for (int k = 0; k < n_sequences; k++) M->seq[k]->th = std::thread(worker, M, k);
  // every FrameHandlerMono is built on its own thread (its constructor reads the thread's counters and Config)
  std::vector<std::function<int()>> tasks(n_sequences);
  for (int k = 0; k < n_sequences; k++)
    tasks[k] = [M, k, cam, max_fts]() { M->seq[k]->vo = hso_vo_create_shared(M->ctx, cam, max_fts); return M->seq[k]->vo ? HSO_OK : HSO_E_NOMEM; };
  const int rc2 = step(M, tasks);
  if (rc2 < 0) { hso_vo_multi_destroy(M); return rc2; }
  *out = M;
  return HSO_OK;
}

void hso_vo_multi_destroy(hso_vo_multi* M)
{
  if (!M) return;
  std::vector<std::function<int()>> tasks(M->seq.size());
  for (size_t k = 0; k < M->seq.size(); k++)
    tasks[k] = [M, k]() { if (M->seq[k]->vo) hso_vo_destroy_shared(M->seq[k]->vo); M->seq[k]->vo = nullptr; return HSO_OK; };
  step(M, tasks);
  for (Seq* S : M->seq) {
    { std::lock_guard<std::mutex> lk(S->m); S->quit = true; }
    S->cv.notify_all();
    if (S->th.joinable()) S->th.join();
    delete S->router;
    delete S;
  }
  hso_gpu_destroy(M->ctx);
  delete M;
}

const char* hso_vo_multi_last_error(const hso_vo_multi* M) { return M ? M->err.c_str() : "null handle"; }
int hso_vo_multi_size(const hso_vo_multi* M) { return M ? (int)M->seq.size() : 0; }

int hso_vo_multi_set_first_frames(hso_vo_multi* M, const uint8_t* const* imgs, int width, int height, const double* timestamps,
                                  const float* const* depth_z, const hso_se3* T_f_w)
{
  if (!M || !imgs || !depth_z) return HSO_E_INVALID;
  std::vector<std::function<int()>> tasks(M->seq.size());
  for (size_t k = 0; k < M->seq.size(); k++) {
    if (!imgs[k]) continue;
    const double ts = timestamps ? timestamps[k] : 0.0;
    tasks[k] = [=]() { return hso_vo_set_first_frame(M->seq[k]->vo, imgs[k], width, height, ts, depth_z[k], T_f_w ? &T_f_w[k] : nullptr); };
  }
  return step(M, tasks);
}

int hso_vo_multi_start(hso_vo_multi* M, const uint8_t* which)
{
  if (!M) return HSO_E_INVALID;
  std::vector<std::function<int()>> tasks(M->seq.size());
  for (size_t k = 0; k < M->seq.size(); k++)
    if (!which || which[k]) tasks[k] = [=]() { return hso_vo_start(M->seq[k]->vo); };
  return step(M, tasks);
}

int hso_vo_multi_add_images(hso_vo_multi* M, const uint8_t* const* imgs, int width, int height, const double* timestamps)
{
  if (!M || !imgs) return HSO_E_INVALID;
  std::vector<std::function<int()>> tasks(M->seq.size());
  for (size_t k = 0; k < M->seq.size(); k++) {
    if (!imgs[k]) continue;                 // a sequence without a new image sits the step out
    const double ts = timestamps ? timestamps[k] : 0.0;
    tasks[k] = [=]() { return hso_vo_add_image(M->seq[k]->vo, imgs[k], width, height, ts); };
  }
  return step(M, tasks);
}

int hso_vo_multi_get_status(hso_vo_multi* M, int sequence, hso_vo_status* st)
{
  if (!M || sequence < 0 || sequence >= (int)M->seq.size()) return HSO_E_INVALID;
  return hso_vo_get_status(M->seq[sequence]->vo, st);
}

int hso_vo_multi_get_keyframes(hso_vo_multi* M, int sequence, double* timestamps, hso_se3* T_f_w, int32_t* frame_ids, int cap)
{
  if (!M || sequence < 0 || sequence >= (int)M->seq.size()) return HSO_E_INVALID;
  return hso_vo_get_keyframes(M->seq[sequence]->vo, timestamps, T_f_w, frame_ids, cap);
}

int hso_vo_multi_call_counts(hso_vo_multi* M, int64_t* calls, int64_t* items, int cap)
{
  if (!M) return HSO_E_INVALID;
  std::lock_guard<std::mutex> lk(M->B.m);
  for (int k = 0; k < K_COUNT && k < cap; k++) { if (calls) calls[k] = M->B.n_calls[k]; if (items) items[k] = M->B.n_items[k]; }
  return K_COUNT;
}

}  // extern "C"
