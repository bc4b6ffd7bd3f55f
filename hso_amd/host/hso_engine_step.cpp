// hso_engine_step.cpp — the per-frame phases of a step: frame construction, CoarseTracker, reprojection + selection + pose
// optimisation, and what FrameHandlerMono::processFrame decides from their results.
#include "hso_engine_impl.h"
#include <unistd.h>
#include <chrono>

namespace hso {
namespace engine {

namespace {
// developer probe (HSO_ENGINE_TIMING=1): wall time per phase of a step, printed when the bank goes away
// and what the device library asked of the runtime meanwhile (hso_gpu_debug_census; process-wide, so meaningful for one bank)
struct Clock {
  double* acc; int64_t (*cen)[7]; bool on;
  std::chrono::steady_clock::time_point t;
  int64_t c0[7], begin[7];
  Clock(double* a, int64_t (*c)[7], bool o) : acc(a), cen(c), on(o) { if (on) { t = std::chrono::steady_clock::now(); hso_gpu_debug_census(c0, 7); for (int i = 0; i < 7; i++) begin[i] = c0[i]; } }
  void lap(int k)
  {
    if (!on) return;
    const auto u = std::chrono::steady_clock::now();
    acc[k] += std::chrono::duration<double, std::milli>(u - t).count();
    int64_t c1[7];
    hso_gpu_debug_census(c1, 7);
    for (int i = 0; i < 7; i++) { cen[k][i] += c1[i] - c0[i]; c0[i] = c1[i]; }
    t = std::chrono::steady_clock::now();
  }
};
}  // namespace

// ------------------------------------------------------------------------------------------------ entry points
void Bank::add_images(const uint8_t* const* imgs, int w, int h, const double* stamps, bool on_device)
{
  if (w != cam_.width() || h != cam_.height())      // src/frame.cpp:85-86: thrown before anything is touched
    throw Refused("Frame: provided image has not the same size as the camera model or image is not grayscale");
  if (on_device) for (int k = 0; k < size(); k++) if (imgs[k] && seq_[k]->trace.on()) throw Refused("trace: images must be host images");
  step(imgs, w, h, stamps, on_device);
}

void Bank::start(const uint8_t* which)
{
  for (int k = 0; k < size(); k++) if (!which || which[k]) seq_[k]->want_start = true;
}

// One step.  Every phase takes the list of sequences it applies to; a phase = per-sequence preparation (pool) -> one batched
// device call -> per-sequence consumption (pool).
void Bank::step(const uint8_t* const* imgs, int w, int h, const double* stamps, bool on_device)
{
  std::vector<int> who;
  for (int k = 0; k < size(); k++) {
    Seq& s = *seq_[k];
    StepData& d = *step_[k];
    d = StepData();
    if (!imgs[k]) continue;
    if (s.want_start) {                              // FrameHandlerBase::startFrameProcessingCommon: start() -> reset -> first frame
      for (Frame& F : s.frames) if (F.in_use && F.dev_id >= 0) to_release_.push_back(F.dev_id);
      drop_sequence_seeds(k);
      s.reset_tables();
      s.motion = SE3(); s.after_init = false; s.regular = 0; s.n_obs_last = 0; s.quality = kInsufficient;
      s.stage = kFirst; s.want_start = false;
    }
    if (s.stage == kPaused) continue;
    d.active = true; d.stage0 = s.stage;
    who.push_back(k);
  }
  if (who.empty()) return;
  pin_threads();
  release_queued();

  timing_ = timing_level() > 0;
  Clock ck(phase_ms_, phase_census_, timing_);
  upload(who, imgs, w, h, stamps, on_device);
  ck.lap(0);
  std::vector<int> starting, running;
  for (int k : who) (seq_[k]->stage == kFirst || seq_[k]->stage == kSecond ? starting : running).push_back(k);
  if (!starting.empty()) initialise(starting);
  if (!running.empty()) {
    chain(running);
    ck.lap(1);
    std::vector<int> tracked;
    for (int k : running) if (step_[k]->tracked) tracked.push_back(k);
    if (!tracked.empty()) {
      std::vector<int> thin, ok, kf;
      for (int k : tracked) if (step_[k]->seed_path) thin.push_back(k);
      if (!thin.empty()) { fetch_features(thin); seed_branch(thin); }
      par(tracked, [&](int k) { decide(k); });
      if (!thin.empty()) send_features(thin);                      // the seed branch changed these frames' features
      ck.lap(2);
      for (int k : tracked) if (step_[k]->ok) { ok.push_back(k); if (step_[k]->make_kf) kf.push_back(k); }
      if (!kf.empty()) {
        // a keyframe's features move into the sequence's tables: the host needs them now
        std::vector<int> need;
        for (int k : kf) if (!step_[k]->seed_path) need.push_back(k);
        fetch_features(need);
        par(kf, [&](int k) { decide_keyframe(k); });
      }
      ck.lap(3);
      if (!kf.empty()) {
        { Sub t(this, "ba: flush_maps"); flush_maps(kf); }          // the window is assembled from the resident map: the new keyframe's rows go first
        keyframe_ba(kf);
      }
      ck.lap(4);
      if (!ok.empty()) {
        observe_seeds(ok);
        ck.lap(5);
        activate_seeds(ok);
        ck.lap(6);
      }
      if (!kf.empty()) start_seeds(kf);
      ck.lap(7);
      n_kf_events_ += (int64_t)kf.size();
    }
  }
  { Sub t(this, "end: flush_maps"); flush_maps(who); }
  { Sub t(this, "end: finish"); finish(who); }
  { Sub t(this, "end: release"); release_queued(); }
  // the depth filter's idle-time sweep over the sequences that stepped, beside the next step's tracking
  {
    std::vector<int> swept;
    for (int k : who) if (seq_[k]->stage == kRunning && step_[k]->ok) swept.push_back(k);
    Sub t(this, "end: previous_begin");
    if (!swept.empty()) previous_begin(swept);
  }
  ck.lap(8);
  n_steps_++;
  if (timing_level() >= 2) {   // one line per step: the phases' wall time since the last step's line
    static thread_local double last[9] = {0};
    static thread_local long long last_kf = 0;
    // keyframes taken in this step, then what the step asked of the runtime (all contexts of the process: read it from a lone bank)
    fprintf(stderr, "[hso engine step %lld] kf %lld (+%lld) | copies %lld syncs %lld to-device KB %.1f from-device KB %.1f |", (long long)n_steps_, (long long)n_kf_events_,
            (long long)n_kf_events_ - last_kf, (long long)(ck.c0[0] - ck.begin[0]), (long long)(ck.c0[3] - ck.begin[3]), (double)(ck.c0[6] - ck.begin[6]) / 1024.0,
            (double)((ck.c0[1] - ck.begin[1]) - (ck.c0[6] - ck.begin[6])) / 1024.0);
    last_kf = (long long)n_kf_events_;
    for (int k = 0; k < 9; k++) { fprintf(stderr, " %.2f", phase_ms_[k] - last[k]); last[k] = phase_ms_[k]; }
    fprintf(stderr, "\n");
  }
}

// ------------------------------------------------------------------------------------------------ frame construction
// `new Frame(cam, img, ts)` of every sequence in one batched call (src/frame_handler_mono.cpp:91-97)
void Bank::upload(const std::vector<int>& who, const uint8_t* const* imgs, int w, int h, const double* stamps, bool on_device)
{
  std::vector<int64_t> ids; std::vector<const uint8_t*> ptr;
  for (int k : who) {
    Seq& s = *seq_[k];
    const Id fr = s.new_frame();
    s.hold(fr);
    s.cur = fr;
    Frame& F = s.frames[fr];
    F.stamp = stamps ? stamps[k] : 0.0;
    F.kf_id = s.kfs.empty() ? 0 : s.frames[s.kfs.back()].kf_id;
    ids.push_back(F.dev_id); ptr.push_back(imgs[k]);
  }
  std::vector<hso_frame_stats> st(who.size());
  check(hso_gpu_frame_upload_batch(ctx_, ids.data(), ptr.data(), (int)who.size(), w, h, on_device ? 1 : 0, st.data()), "Frame");
  n_calls_[0]++; n_items_[0] += (int64_t)who.size();
  for (size_t i = 0; i < who.size(); i++) {
    Seq& s = *seq_[who[i]];
    Frame& F = s.frames[s.cur];
    F.integral = st[i].integral_image; F.grad_mean = st[i].grad_mean;
    if (s.trace.on()) {
      s.trace.begin("frame_upload", 5);
      s.trace.scalar("frame_id", (double)F.dev_id); s.trace.scalar("width", w); s.trace.scalar("height", h);
      s.trace.field("img", imgs[who[i]], (size_t)w * h); s.trace.field("stats", &st[i], sizeof(st[i]));
    }
  }
}

// ------------------------------------------------------------------------------------------------ CoarseTracker (value-passing)
namespace {

// the reference frame's features as CoarseTracker::makeDepthRef sees them (src/CoarseTracker.cpp:210-240): the distance of the
// point along the bearing, from its host-frame inverse depth; -1 keeps the slot of a feature without a usable point
// Written straight into the tracker kernel's layout: six arrays px[0] | px[1] | f[0] | f[1] | f[2] | dist of `stride` doubles.
// (The per-frame chain builds this table on the device; the host form serves relocalizeFrame's first alignment, whose reference
// is a keyframe the sequence may not have looked at for a long time.)
void reference_features(const Seq& s, const Frame& R, double* out, size_t stride)
{
  const size_t n = R.fts.size();
  double* px0 = out; double* px1 = out + stride; double* f0 = out + 2 * stride; double* f1 = out + 3 * stride; double* f2 = out + 4 * stride;
  double* dist = out + 5 * stride;
  Id cached = kNone;
  SE3 T_ref_host;
  for (size_t i = 0; i < n; i++) {
    const Feat& ft = s.feats[R.fts[i]];
    px0[i] = ft.px[0]; px1[i] = ft.px[1];
    f0[i] = ft.f[0]; f1[i] = ft.f[1]; f2[i] = ft.f[2];
    dist[i] = -1;
    if (ft.point == kNone) continue;
    const Point& P = s.points[ft.point];
    if (P.host_frame != cached) { T_ref_host = R.T * s.frames[P.host_frame].T.inverse(); cached = P.host_frame; }   // runs of points share a host keyframe
    const Vector3d p = T_ref_host * along(P.host_f, 1.0 / P.idist);
    if (!(p[2] < 0.00001)) dist[i] = std::sqrt(p[0] * p[0] + p[1] * p[1] + p[2] * p[2]);
  }
  for (size_t i = n; i < stride; i++) px0[i] = px1[i] = f0[i] = f1[i] = f2[i] = dist[i] = 0.0;
}

void trace_track(Trace& t, const hso_camera& cam, const hso_track_params& p, int64_t ref_id, int64_t cur_id, const hso_ref_feat* rec, size_t n, const hso_se3& T_cur_ref,
                 float exposure_rat, const hso_track_result& res)
{
  t.begin("coarse_track", 8);
  t.field("cam", &cam, sizeof(cam)); t.field("params", &p, sizeof(p));
  t.scalar("ref_frame_id", (double)ref_id); t.scalar("cur_frame_id", (double)cur_id);
  t.field("feats", rec, sizeof(hso_ref_feat) * n);
  t.field("T_cur_ref", &T_cur_ref, sizeof(hso_se3)); t.scalar("exposure_rat", exposure_rat);
  t.field("result", &res, sizeof(res));
}

}  // namespace

// run one tracker configuration over (reference KEYFRAME, current) pairs of several sequences with host-built tables; results in
// StepData::track
void Bank::track_group(const std::vector<int>& who, const std::vector<Id>& ref, const std::vector<Id>& cur, const hso_track_params& p)
{
  if (who.empty()) return;
  std::vector<hso_track_job> jobs(who.size());
  std::vector<hso_track_result> res(who.size());
  std::vector<size_t> at(who.size() + 1, 0);
  for (size_t i = 0; i < who.size(); i++) at[i + 1] = at[i] + 6 * ((seq_[who[i]]->frames[ref[i]].fts.size() + 31) & ~size_t(31));
  double* const block = track_tables_.need(ctx_, at.back() + 64);
  pool_->run((int)who.size(), [&](int i) {
    Seq& s = *seq_[who[i]];
    const Frame& R = s.frames[ref[i]];
    const Frame& C = s.frames[cur[i]];
    reference_features(s, R, block + at[i], (at[i + 1] - at[i]) / 6);
    hso_track_job& j = jobs[i];
    j = hso_track_job{};
    j.ref_frame_id = R.dev_id; j.cur_frame_id = C.dev_id;
    j.feats = reinterpret_cast<const hso_ref_feat*>(block + at[i]); j.n_feats = (int)R.fts.size(); j.feats_soa = 1;
    j.T_cur_ref = (C.T * R.T.inverse()).v;                        // src/CoarseTracker.cpp:63
    j.exposure_rat = C.integral / R.integral;                     // :60
  });
  check(hso_gpu_coarse_track_batch(ctx_, &cam_.pod(), &p, jobs.data(), (int)jobs.size(), res.data()), "CoarseTracker");
  n_calls_[2]++; n_items_[2] += (int64_t)who.size();
  for (size_t i = 0; i < who.size(); i++) {
    Seq& s = *seq_[who[i]];
    StepData& d = *step_[who[i]];
    d.track = res[i];
    // the write-back of CoarseTracker::run (:198-202)
    Frame& C = s.frames[cur[i]];
    const Frame& R = s.frames[ref[i]];
    SE3 T_cur_ref; T_cur_ref.v = res[i].T_cur_ref;
    C.T = T_cur_ref * R.T;
    C.exposure = (double)res[i].exposure_rat * R.exposure;
    if (res[i].exposure_rat > 0.99 && res[i].exposure_rat < 1.01) C.exposure = R.exposure;
    if (s.trace.on()) {
      const size_t n = (size_t)jobs[i].n_feats, st = (n + 31) & ~size_t(31);
      const double* a = block + at[i];
      std::vector<hso_ref_feat> rec(n);
      for (size_t q = 0; q < n; q++) { rec[q].px[0] = a[q]; rec[q].px[1] = a[st + q]; rec[q].f[0] = a[2 * st + q]; rec[q].f[1] = a[3 * st + q]; rec[q].f[2] = a[4 * st + q]; rec[q].dist = a[5 * st + q]; }
      trace_track(s.trace, cam_.pod(), p, jobs[i].ref_frame_id, jobs[i].cur_frame_id, rec.data(), n, jobs[i].T_cur_ref, jobs[i].exposure_rat, res[i]);
    }
  }
}

// ------------------------------------------------------------------------------------------------ the per-frame chain
// FrameHandlerMono::processFrame from the motion prior to the inputs of its decisions (src/frame_handler_mono.cpp:173-291), on the
// device for all sequences (hso_gpu_seq_chain): CoarseTracker::run against the last frame's features, Reprojector::reprojectMap's
// walk over the overlap keyframes, matching, grid selection, the frame's features, the pose optimiser, the candidates' failure /
// success counters, needNewKf's flow sums, the covisibility votes, the scene depth.  What the host contributes per sequence is one
// job record — the motion prior, which keyframes the last frame was connected to, the temporary points — and what it reads is one
// result record; its own tables follow the device's through the events (kind changes of points).
void Bank::prepare_job(int k, hso_seq_job& job, std::vector<int32_t>& temps)
{
  Seq& s = *seq_[k];
  StepData& d = *step_[k];
  Frame& C = s.frames[s.cur];
  Id last = d.relocalised ? d.ref : s.last;
  C.T = s.motion * s.frames[last].T;                              // processFrame, :176
  if (s.after_init) last = s.first;                               // :180
  d.ref = last;
  d.tracked = true;
  s.log = hso_vo_status{};
  Frame& L = s.frames[last];
  d.inverse = !(C.grad_mean > L.grad_mean + 0.5f) ? 1 : 0;        // :184
  // the temporary points whose seed has finished are retired before the map is projected (src/reprojector.cpp:98-106)
  {
    size_t keep = 0;
    for (size_t i = 0; i < s.temps.size(); i++) {
      const Id p = s.temps[i];
      if (s.points[p].seed_state == 0) { s.temps[keep++] = p; continue; }
      s.retire_temp(p);
    }
    s.temps.resize(keep);
  }
  job = hso_seq_job{};
  job.map = s.map;
  job.ref_frame_id = L.dev_id; job.cur_frame_id = C.dev_id;
  job.T_ref_w = L.T.v; job.T_cur_w = C.T.v; job.ref_exposure = L.exposure;
  job.ref_kf_row = L.kf_row;
  job.n_ref_feats = (int32_t)s.n_feats(L);
  if (job.n_ref_feats == 0) job.flags |= HSO_SEQ_NO_TRACK;        // CoarseTracker::run returns 0 at once (src/CoarseTracker.cpp:53-54)
  if (!s.seeds.empty() && (int)s.seeds.size() > s.n_dead_seeds) job.flags |= HSO_SEQ_SEED_BRANCH;   // see consume_result: seed_path
  if (s.after_init) job.flags |= HSO_SEQ_DEPTH_STATS;             // a keyframe for sure (:279): its scene depth is wanted
  job.cur_keyframe_id = C.kf_id;
  job.exposure_rat = C.integral / L.integral;                     // src/CoarseTracker.cpp:60
  // needNewKf looks at the flow only once three regular frames have passed (:430-437)
  job.last_kf_row = (s.regular >= 3 && s.regular >= std::min(3, int(s.n_mean_converge * 0.8)) && !s.after_init && !s.kfs.empty()) ? s.frames[s.kfs.back()].kf_row : -1;
  // the reference frame's connected keyframes that are still keyframes of the map (src/reprojector.cpp:124-170)
  int nc = 0;
  for (int q = 0; q < 5; q++) job.covis[q] = -1;
  for (Id kf : L.covis) {
    const Frame& K = s.frames[kf];
    if (!K.in_use || K.kf_row < 0 || nc >= 5) continue;
    if (std::find(s.kfs.begin(), s.kfs.end(), kf) == s.kfs.end()) continue;   // Map::getKeyframeById
    job.covis[nc++] = K.kf_row;
  }
  L.covis.clear();
  // the temporary points the frame lists (:228-251): not given up, at the position their seed's current depth puts them
  d.temps_listed.clear();
  job.temps_begin = (int32_t)temps.size();
  for (Id p : s.temps) {
    Point& P = s.points[p];
    if (P.bad) continue;
    s.place_in_host(p);
    d.temps_listed.push_back(p);
    temps.push_back(p);
  }
  job.n_temps = (int32_t)d.temps_listed.size();
  // DepthFilter::addFrame of a regular frame rides behind the frame on the device (hso_seq_chain_cfg: seed_table); a recorded run
  // keeps the call of its own (the trace wants the seeds' records before and after)
  job.seed_group = s.trace.on() ? -1 : s.index;
}

// what the chain reports about a frame, applied to the sequence's own tables
void Bank::consume_result(int k, const hso_seq_result& r, const std::vector<int32_t>& more_events)
{
  Seq& s = *seq_[k];
  StepData& d = *step_[k];
  Frame& C = s.frames[s.cur];
  d.res = r;
  d.track = r.track;
  C.T.v = r.T_tracked;                                            // the write-back of CoarseTracker::run (:198-202)
  C.exposure = r.exposure;
  s.log.n_tracked = d.track.n_tracked; s.log.used_inverse = d.inverse;
  // the kind changes reprojectCell / reprojectCellAll made (src/reprojector.cpp:366-425, 214-222, 247-251), in their order
  const int n_ev = r.n_events;
  for (int i = 0; i < n_ev; i++) {
    const int32_t e = i < HSO_SEQ_EVENTS && more_events.empty() ? r.events[i] : more_events[(size_t)i];
    const int code = (int)((uint32_t)e >> 28);
    const Id p = (Id)(e & 0x0fffffff);
    if (code == HSO_EV_ERASE_POINT) s.erase_point(p);
    else if (code == HSO_EV_ERASE_CANDIDATE) s.erase_candidate(p);
    else if (code == HSO_EV_TEMP_BAD) s.points[p].bad = true;
    else if (code == HSO_EV_GOOD) { s.points[p].kind = kPtGood; s.touch_point(p); }
  }
  C.loose.clear();
  C.n_fts = r.n_feats;
  s.log.n_trials = r.counts[0]; s.log.n_matches = r.counts[1]; s.log.n_seed_matches = 0;
  d.pose = r.pose;
  d.visit.clear();
  for (int q = 0; q < r.n_visit && q < HSO_SEQ_MAX_VISIT; q++) d.visit.push_back(s.dev_kfs[(size_t)r.visit[q]]);
  // too few matches: the nearly converged seeds are tried as well (:309-329) — the frame's features change, so the pose
  // optimisation that ran behind the selection does not count for it
  d.seed_path = s.log.n_matches < 100 && !s.seeds.empty() && (int)s.seeds.size() > s.n_dead_seeds;
}

void Bank::chain(const std::vector<int>& who)
{
  // a sequence that lost track first aligns its LAST frame against the closest keyframe (relocalizeFrame,
  // src/frame_handler_mono.cpp:357-386: inverse compositional, levels 4..0, 15 iterations); more than 30 tracked features
  // and the new frame goes through the normal path with that keyframe as its reference
  {
    std::vector<int> lost; std::vector<Id> ref, cur;
    for (int k : who) {
      Seq& s = *seq_[k];
      if (s.stage != kRelocalising) continue;
      std::vector<std::pair<double, Id>> near;
      s.closest_keyframes(s.frames[s.last], near);               // Map::getClosestKeyframe
      std::stable_sort(near.begin(), near.end(), [](const std::pair<double, Id>& a, const std::pair<double, Id>& b) { return a.first < b.first; });
      Id kf = kNone;
      for (const auto& c : near) if (c.second != s.last) { kf = c.second; break; }
      if (kf == kNone || s.n_feats(s.frames[kf]) == 0) continue;  // no reference keyframe: RESULT_FAILURE
      lost.push_back(k); ref.push_back(kf); cur.push_back(s.last);
    }
    const hso_track_params p{1, cfg_.klt_max_level, cfg_.klt_min_level, 15};
    track_group(lost, ref, cur, p);
    for (size_t i = 0; i < lost.size(); i++) {
      Seq& s = *seq_[lost[i]];
      StepData& d = *step_[lost[i]];
      if (d.track.n_tracked > 30) { d.relocalised = true; d.ref = ref[i]; d.reloc_pose = s.frames[s.last].T; }
    }
  }
  std::vector<int> in;
  for (int k : who) if (!(seq_[k]->stage == kRelocalising && !step_[k]->relocalised)) in.push_back(k);
  if (in.empty()) return;
  { Sub t(this, "chain: previous_collect"); previous_collect(); }   // the idle-time pass of the last step, before seeds are observed again
  // the job records: serial (they append to one list of temporary points; a few dozen assignments per sequence)
  std::vector<hso_seq_job> all(in.size());
  std::vector<int32_t> temps;
  { Sub t(this, "chain: prepare_job"); for (size_t i = 0; i < in.size(); i++) prepare_job(in[i], all[i], temps); }
  { Sub t(this, "chain: flush_maps"); flush_maps(in); }           // the rows, lists and links that changed since the last frame
  hso_seq_chain_cfg cfg{};
  cfg.cell_size = cell_size_; cfg.grid_n_cols = grid_cols_; cfg.n_cells = (int)cell_order_.size(); cfg.max_fts = cfg_.max_fts;
  cfg.cell_order = cell_order_.data(); cfg.max_kfs = cfg_.reproject_max_kfs; cfg.pose_n_iter = 12; cfg.pose_reproj_thresh = cfg_.poseoptim_thresh;
  cfg.quality_min_fts = cfg_.quality_min_fts;
  bool any_trace = false;
  for (int k : in) any_trace |= seq_[k]->trace.on();
  cfg.want_debug = any_trace ? 1 : 0;
  int n_slots = 0, n_live = 0;
  check(hso_gpu_seed_table_size(ctx_, seed_table_, &n_slots, &n_live), "DepthFilter");
  hso_seed_brief* const brief_images = n_live > 0 ? chain_brief_.need(ctx_, 2 * (size_t)n_slots) : nullptr;
  cfg.seed_table = n_live > 0 ? seed_table_ : -1; cfg.n_seed_groups = size(); cfg.px_error_angle = px_error_angle_; cfg.seed_brief_cap = n_slots;
  for (int mode = 0; mode < 2; mode++) {
    // one call per tracker mode (almost always one: the mode follows the gradient statistics of consecutive frames); inside a
    // call the sequences whose reference frame has no features come last (the tracker skips them)
    std::vector<int> grp; std::vector<hso_seq_job> jobs;
    for (int pass = 0; pass < 2; pass++)
      for (size_t i = 0; i < in.size(); i++)
        if (step_[in[i]]->inverse == mode && ((all[i].flags & HSO_SEQ_NO_TRACK) != 0) == (pass == 1)) { grp.push_back(in[i]); jobs.push_back(all[i]); }
    if (grp.empty()) continue;
    cfg.track = hso_track_params{mode, cfg_.klt_max_level, cfg_.klt_min_level + 1, 50};
    cfg.seed_brief_out = brief_images ? brief_images + (size_t)mode * (size_t)n_slots : nullptr;
    hso_seq_result* res = chain_res_.need(ctx_, grp.size());
    if (any_trace) trace_chain_state(grp, jobs, cfg, temps);
    { Sub t(this, "chain: device call"); check(hso_gpu_seq_chain(ctx_, &cam_.pod(), &cfg, jobs.data(), (int)jobs.size(), temps.empty() ? nullptr : temps.data(), (int)temps.size(), res), "processFrame"); }
    n_calls_[2]++; n_items_[2] += (int64_t)grp.size();
    n_calls_[3]++; n_items_[3] += (int64_t)grp.size();
    std::vector<std::vector<int32_t>> more(grp.size());
    for (size_t i = 0; i < grp.size(); i++)
      if (res[i].n_events > HSO_SEQ_EVENTS) {                      // more kind changes than the record holds: fetch them all
        more[i].resize((size_t)res[i].n_events);
        check(hso_gpu_seq_events(ctx_, (int)i, more[i].data(), (int)more[i].size()), "processFrame");
      }
    if (any_trace) trace_chain(grp, jobs, cfg, res);
    Sub t_consume(this, "chain: consume");
    pool_->run((int)grp.size(), [&](int i) {
      consume_result(grp[(size_t)i], res[i], more[(size_t)i]);
      StepData& d = *step_[grp[(size_t)i]];
      d.seeds_observed = res[i].seeds_observed != 0; d.chain_brief = cfg.seed_brief_out;
    });
    if (cfg.seed_table >= 0) { n_calls_[6]++; for (size_t i = 0; i < grp.size(); i++) n_items_[6] += res[i].seeds_observed; }
    // the algorithmic bytes of this call's work (a measurement aid: hso_vo_multi_alg_bytes)
    {
      static const int PA[5] = {25, 21, 13, 13, 9}, PAD[5] = {2, 3, 2, 2, 1};
      const double wh = (double)cam_.width() * cam_.height();
      for (size_t i = 0; i < grp.size(); i++) {
        const hso_seq_result& r = res[i];
        const double n = (double)jobs[i].n_ref_feats;
        alg_bytes_[0] += (1.33 + 0.33 + 1.31 + 5.25) * wh;                      // pyramid read + written, Sobel read + written
        for (int L = cfg.track.min_level; L <= cfg.track.max_level; L++) {
          const double pa = PA[L], u_fwd = (2.0 * PAD[L] + 4) * (2.0 * PAD[L] + 4), u_ic = (2.0 * PAD[L] + 2) * (2.0 * PAD[L] + 2);
          const double b_alg = mode ? n * (32 + 28 * pa + u_ic) : n * (32 + 4 * pa + u_fwd);
          const double b_pre = mode ? n * (16 + u_fwd + 4 * pa + 24 * pa) : n * (16 + u_ic + 4 * pa);
          const double b_sel = n * (32 + 4 * pa + u_ic);
          if (!(jobs[i].flags & HSO_SEQ_NO_TRACK)) alg_bytes_[1] += r.track.n_eval[L] * b_alg + b_pre + b_sel;
        }
        alg_bytes_[2] += (double)r.n_listed * (100 * 4 + 4.2 * 64 * 4);         // per listed point: the warped patch + 4.2 LK iterations (the measured mean)
        alg_bytes_[3] += (double)(r.pose.n_trials_total + 1) * r.n_feats * sizeof(hso_pose_feat);
      }
    }
  }
}

// the features of these sequences' new frames, from the device's tables into Frame::loose
void Bank::fetch_features(const std::vector<int>& who)
{
  if (who.empty()) return;
  const int cap = std::max(cfg_.max_fts, 1);
  std::vector<int32_t> maps, n_out(who.size(), 0); std::vector<int64_t> ids;
  for (int k : who) { maps.push_back(seq_[k]->map); ids.push_back(seq_[k]->frames[seq_[k]->cur].dev_id); }
  hso_seq_feature* rows = feat_rows_.need(ctx_, who.size() * (size_t)cap);
  check(hso_gpu_seq_frame_features(ctx_, maps.data(), ids.data(), (int)who.size(), rows, cap, n_out.data()), "Frame::fts_");
  pool_->run((int)who.size(), [&](int i) {
    Seq& s = *seq_[who[(size_t)i]];
    Frame& C = s.frames[s.cur];
    C.loose.clear();
    C.loose.reserve((size_t)n_out[(size_t)i] + 8);
    for (int q = 0; q < n_out[(size_t)i]; q++) {
      const hso_seq_feature& r = rows[(size_t)i * cap + q];
      Feat nf;
      nf.frame = s.cur; nf.point = r.point;
      nf.px[0] = r.px[0]; nf.px[1] = r.px[1];
      nf.f[0] = r.f[0]; nf.f[1] = r.f[1]; nf.f[2] = r.f[2];
      nf.level = r.level;
      if (r.type == HSO_FTR_EDGELET) { nf.type = HSO_FTR_EDGELET; nf.grad[0] = r.grad[0]; nf.grad[1] = r.grad[1]; }
      else nf.type = r.type == HSO_FTR_GRADIENT ? HSO_FTR_GRADIENT : HSO_FTR_CORNER;
      C.loose.push_back(nf);
    }
    C.n_fts = (int32_t)C.loose.size();
  });
}

// ... and back, after the host changed them (the seed branch added features; the pose optimiser ran over host tables)
void Bank::send_features(const std::vector<int>& who)
{
  for (int k : who) {
    Seq& s = *seq_[k];
    Frame& C = s.frames[s.cur];
    if (C.kf_row >= 0) continue;
    std::vector<hso_seq_feature> rows(C.loose.size());
    for (size_t i = 0; i < C.loose.size(); i++) {
      const Feat& ft = C.loose[i];
      hso_seq_feature& r = rows[i];
      r = hso_seq_feature{};
      r.px[0] = ft.px[0]; r.px[1] = ft.px[1]; r.f[0] = ft.f[0]; r.f[1] = ft.f[1]; r.f[2] = ft.f[2];
      r.grad[0] = (float)ft.grad[0]; r.grad[1] = (float)ft.grad[1]; r.point = ft.point; r.level = ft.level; r.type = ft.type;
    }
    check(hso_gpu_seq_set_frame_features(ctx_, s.map, C.dev_id, rows.data(), (int)rows.size()), "Frame::fts_");
    C.n_fts = (int32_t)rows.size();
  }
}

// the frame's pose result (pose_optimizer::optimizeLevenbergMarquardt3rd's effects, src/pose_optimizer.cpp:692-767) and the
// decisions FrameHandlerMono::processFrame takes from it (:224-291), up to the choice between a regular frame and a keyframe
void Bank::decide(int k)
{
  decide_frame(k);
  const StepData& d = *step_[k];
  // the device observed the seeds exactly where this function lets the frame pass as a regular one
  if (d.seeds_observed && !(d.ok && !d.make_kf)) throw std::logic_error("the chain observed the seeds of a frame the handler did not accept as a regular one");
}

void Bank::decide_frame(int k)
{
  Seq& s = *seq_[k];
  StepData& d = *step_[k];
  Frame& C = s.frames[s.cur];
  const Frame& L = s.frames[d.ref];
  d.ok = false;
  s.outcome = kFailure;
  if (s.log.n_matches < cfg_.quality_min_fts) {                   // :224-230
    C.T = L.T;
    s.quality = kInsufficient;
    return;
  }
  if (d.pose.status == 0) {
    C.T.v = d.pose.T_f_w;
    std::memcpy(C.cov, d.pose.cov, sizeof(C.cov));
    C.err_px = d.pose.error_in_px;
    // the culled features (feature->point = NULL): on the device's table already, unless the pose ran over host tables
    if (d.host_pose) for (size_t i = 0; i < C.loose.size() && i < d.pose_mask.size(); i++) if (d.pose_mask[i]) C.loose[i].point = kNone;
  }
  d.n_inliers = (size_t)d.pose.num_obs;
  C.n_inliers = d.pose.num_obs;
  s.log.pose_error_init = d.pose.error_init; s.log.pose_error_final = d.pose.error_final;
  if ((int)d.n_inliers < cfg_.quality_min_fts) return;            // :253-254
  // setTrackingQuality, src/frame_handler_base.cpp:165-179
  s.quality = kGood;
  if ((int)d.n_inliers < cfg_.quality_min_fts) s.quality = kInsufficient;
  if (std::min(s.n_obs_last, cfg_.max_fts) - (int)d.n_inliers > cfg_.quality_max_drop_fts) s.quality = kBad;
  if (s.quality == kInsufficient) { C.T = L.T; return; }
  d.ok = true;
  // needNewKf: the chain evaluated it on the flow sums it formed (hso_seq_result.make_kf) — and let the frame's seeds be observed
  // behind it when the answer was no; a pose the host optimised again (the seed branch) is judged here
  d.make_kf = s.after_init || (d.host_pose ? wants_keyframe(k) : d.res.make_kf != 0);
  if (!d.make_kf) {
    // createCovisibilityGraph of a regular frame (:559-647): the ranking came with the result
    if (!d.host_pose) {
      for (int q = 0; q < HSO_SEQ_MAX_COVIS && q < 5; q++) if (d.res.covis[q] >= 0) C.covis.push_back(s.dev_kfs[(size_t)d.res.covis[q]]);
    } else link_covisible(k, false);
    s.outcome = kNoKeyframe;
  }
}

// the frame becomes a keyframe: its features are on the host now
void Bank::decide_keyframe(int k)
{
  Seq& s = *seq_[k];
  StepData& d = *step_[k];
  Frame& C = s.frames[s.cur];
  // frame_utils::getSceneDepth / getSceneDistance (src/frame.cpp:323-366).  The reference computes them for every frame
  // (src/frame_handler_mono.cpp:268-271) but only a keyframe uses them (depth_filter_->addKeyframe, :335-338; needNewKf ignores its
  // depth argument)
  if (!d.host_pose && d.res.depth_min >= 0.0) { d.depth_mean = d.res.depth_median; d.dist_mean = d.res.dist_median; d.depth_min = d.res.depth_min; }
  else {
    std::vector<double> z, r;
    d.depth_min = std::numeric_limits<double>::max();
    for (const Feat& ft : C.loose) {
      if (ft.point == kNone) continue;
      const double* w = s.points[ft.point].pos;
      const Vector3d c = C.T * Vector3d{w[0], w[1], w[2]};
      z.push_back(c[2]); r.push_back(std::sqrt(c[0] * c[0] + c[1] * c[1] + c[2] * c[2]));
      d.depth_min = std::fmin(c[2], d.depth_min);
    }
    if (!z.empty()) { d.depth_mean = upper_median(z); d.dist_mean = upper_median(r); }
  }
  s.outcome = kKeyframe;
  promote(k);
}

// FrameHandlerMono::needNewKf (:428-507): the mean optical flow the motion since the last keyframe induces on that keyframe's
// features — once with the full motion, once with its translation alone — weighted the way DSO weights them.  The two sums over
// the keyframe's features came with the chain's result.
bool Bank::wants_keyframe(int k)
{
  Seq& s = *seq_[k];
  StepData& d = *step_[k];
  if (s.regular < 3) return false;
  if (s.regular < std::min(3, int(s.n_mean_converge * 0.8))) return false;
  float flow_full = d.res.flow_full, flow_shift = d.res.flow_shift;
  size_t count = (size_t)d.res.flow_count;
  if (d.host_pose) {
    // the pose changed after the chain (the seed branch): the sums are formed here
    const Frame& C = s.frames[s.cur];
    const Frame& K = s.frames[s.kfs.back()];
    const SE3 T_cur_kf = C.T * K.T.inverse();
    const Vector3d kf_centre = s.centre(K);
    flow_full = 0; flow_shift = 0; count = 0;
    for (Id f : K.fts) {
      const Feat& ft = s.feats[f];
      if (ft.point == kNone) continue;
      const double* w = s.points[ft.point].pos;
      const double off[3] = {w[0] - kf_centre[0], w[1] - kf_centre[1], w[2] - kf_centre[2]};
      const Vector3d in_kf = along(ft.f, len3(off));
      const Vector2d a = cam_.world2cam(T_cur_kf * in_kf);
      const Vector2d b = cam_.world2cam(Vector3d{in_kf[0] + T_cur_kf.v.t[0], in_kf[1] + T_cur_kf.v.t[1], in_kf[2] + T_cur_kf.v.t[2]});
      flow_full += (a[0] - ft.px[0]) * (a[0] - ft.px[0]) + (a[1] - ft.px[1]) * (a[1] - ft.px[1]);
      flow_shift += (b[0] - ft.px[0]) * (b[0] - ft.px[0]) + (b[1] - ft.px[1]) * (b[1] - ft.px[1]);
      ++count;
    }
  }
  flow_full /= count;
  if (flow_full < 133) return false;
  flow_full = sqrtf(flow_full);
  flow_shift = sqrtf(flow_shift / count);
  const int nominal = 752 + 480;
  const float w_shift = 0.04 * nominal, w_full = 0.02 * nominal, w_global = 0.75;
  const int extent = cam_.width() + cam_.height();
  const float score = w_global * w_shift * flow_shift / extent + w_global * w_full * flow_full / extent;
  return score > 1;
}

// createCovisibilityGraph (:559-647) on the host's tables: keyframes ranked by how many of the frame's points they observe (a new
// keyframe, whose features have just joined their points' observation lists; a frame whose features the host changed)
void Bank::link_covisible(int k, bool is_keyframe)
{
  Seq& s = *seq_[k];
  Frame& C = s.frames[s.cur];
  std::vector<int>& votes = s.votes;
  votes.assign(s.frames.size(), 0);
  std::vector<Id> seen;
  int with_point = 0;
  const size_t n = C.kf_row >= 0 ? C.fts.size() : C.loose.size();
  for (size_t i = 0; i < n; i++) {
    const Id p = s.feat_of(C, i).point;
    if (p == kNone) continue;
    ++with_point;
    for (Id o = s.points[p].head; o != kNone; o = s.feats[o].next) {
      const Id fr = s.feats[o].frame;
      if (fr == s.cur) continue;
      if (votes[fr]++ == 0) seen.push_back(fr);
    }
  }
  if (seen.empty()) return;
  std::sort(seen.begin(), seen.end(), [&](Id a, Id b) { return s.frames[a].serial < s.frames[b].serial; });
  const int need = with_point > 30 ? 5 : 3;
  std::vector<Id> ranked;
  Id best = seen[0];
  for (Id fr : seen) {
    if (votes[fr] > votes[best]) best = fr;
    if (votes[fr] >= need) ranked.push_back(fr);
  }
  if (ranked.empty()) ranked.push_back(best);
  std::stable_sort(ranked.begin(), ranked.end(), [&](Id a, Id b) { return votes[a] > votes[b]; });   // ties stay in frame order
  for (size_t i = 0; i < ranked.size() && i < 5; i++) C.covis.push_back(ranked[i]);
  if (!is_keyframe) return;
  s.local_map.clear();
  for (size_t i = 0; i < ranked.size() && i < (size_t)cfg_.core_n_kfs; i++) s.local_map.push_back(ranked[i]);
  const Id last_kf = s.kfs.back();
  if (std::find(s.local_map.begin(), s.local_map.end(), last_kf) == s.local_map.end()) s.local_map.push_back(last_kf);
  s.local_map.push_back(s.cur);
}

// the frame becomes a keyframe (:293-311): its features move into the sequence's feature table (= the device's observation rows),
// every feature with a point joins the point's observations, candidates observed here become map points
void Bank::promote(int k)
{
  Seq& s = *seq_[k];
  StepData& d = *step_[k];
  Frame& C = s.frames[s.cur];
  s.after_init = false;
  s.regular = 0;
  make_keyframe(s, s.cur);
  for (Id f : C.fts) if (s.feats[f].point != kNone) s.observe(s.feats[f].point, f);
  // MapPointCandidates::addCandidatePointToFrame (src/map.cpp:318-360)
  {
    size_t keep = 0;
    for (size_t i = 0; i < s.candidates.size(); i++) {
      const Id p = s.candidates[i];
      Point& P = s.points[p];
      if (P.head == kNone || s.feats[P.head].frame != s.cur) { s.candidates[keep++] = p; continue; }
      P.kind = kPtUnknown; P.dev_reset |= 1;                     // n_failed_reproj_ = 0
      s.touch_point(p);
      s.frames[s.feats[P.host].frame].fts.push_back(P.host);
      s.list_grew(s.feats[P.host].frame);
    }
    s.candidates.resize(keep);
  }
  link_covisible(k, true);
  if (cfg_.loba_num_iter > 0) window_job(k);
  (void)d;
}

// Frame::setKeyframe (src/frame.cpp:98-105) + the move of the features into the shared table
void Bank::make_keyframe(Seq& s, Id fr)
{
  Frame& F = s.frames[fr];
  F.kf_row = (int32_t)s.dev_kfs.size();
  s.dev_kfs.push_back(fr);
  s.kfs_dirty = true;
  F.fts.reserve(F.loose.size());
  for (const Feat& ft : F.loose) {
    s.feats.push_back(ft);
    const Id f = (Id)s.feats.size() - 1;
    s.feats[f].frame = fr; s.feats[f].next = kNone; s.feats[f].linked = false;
    F.fts.push_back(f);
    s.touch_obs(f);
  }
  F.fts_sent = 0;
  s.list_grew(fr);
  F.loose.clear(); F.loose.shrink_to_fit();
  s.refresh_keys(F);
  F.kf_id = ++s.n_kfs_made;
  s.hold(fr);                                                     // the map's reference
}

}  // namespace engine
}  // namespace hso
