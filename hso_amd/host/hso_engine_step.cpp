// hso_engine_step.cpp — the per-frame phases of a step: frame construction, CoarseTracker, reprojection + selection + pose
// optimisation, and what FrameHandlerMono::processFrame decides from their results.
#include "hso_engine_impl.h"
#include <chrono>

namespace hso {
namespace engine {

namespace {
// developer probe (HSO_ENGINE_TIMING=1): wall time per phase of a step, printed when the bank goes away
// and what the device library asked of the runtime meanwhile (hso_gpu_debug_census; process-wide, so meaningful for one bank)
struct Clock {
  double* acc; int64_t (*cen)[6]; bool on;
  std::chrono::steady_clock::time_point t;
  int64_t c0[6];
  Clock(double* a, int64_t (*c)[6], bool o) : acc(a), cen(c), on(o) { if (on) { t = std::chrono::steady_clock::now(); hso_gpu_debug_census(c0, 6); } }
  void lap(int k)
  {
    if (!on) return;
    const auto u = std::chrono::steady_clock::now();
    acc[k] += std::chrono::duration<double, std::milli>(u - t).count();
    int64_t c1[6];
    hso_gpu_debug_census(c1, 6);
    for (int i = 0; i < 6; i++) { cen[k][i] += c1[i] - c0[i]; c0[i] = c1[i]; }
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
  release_queued();

  Clock ck(phase_ms_, phase_census_, getenv("HSO_ENGINE_TIMING") != nullptr);
  upload(who, imgs, w, h, stamps, on_device);
  ck.lap(0);
  std::vector<int> starting, running;
  for (int k : who) (seq_[k]->stage == kFirst || seq_[k]->stage == kSecond ? starting : running).push_back(k);
  if (!starting.empty()) initialise(starting);
  if (!running.empty()) {
    track(running);
    ck.lap(1);
    std::vector<int> tracked;
    for (int k : running) if (step_[k]->tracked) tracked.push_back(k);
    if (!tracked.empty()) {
      reproject(tracked);
      ck.lap(2);
      std::vector<int> thin, ok, kf;
      for (int k : tracked) if (step_[k]->seed_path) thin.push_back(k);
      if (!thin.empty()) seed_branch(thin);
      par(tracked, [&](int k) { decide(k); });
      ck.lap(3);
      for (int k : tracked) if (step_[k]->ok) { ok.push_back(k); if (step_[k]->make_kf) kf.push_back(k); }
      if (!kf.empty()) keyframe_ba(kf);
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
  flush_maps(who);
  finish(who);
  release_queued();
  // the depth filter's idle-time sweep over the sequences that stepped, beside the next step's tracking
  {
    std::vector<int> swept;
    for (int k : who) if (seq_[k]->stage == kRunning && step_[k]->ok) swept.push_back(k);
    if (!swept.empty()) previous_begin(swept);
  }
  ck.lap(8);
  n_steps_++;
  if (const char* e = getenv("HSO_ENGINE_TIMING")) if (atoi(e) >= 2) {   // one line per step: the phases' wall time since the last step's line
    static thread_local double last[9] = {0};
    fprintf(stderr, "[hso engine step %lld] kf %lld |", (long long)n_steps_, (long long)n_kf_events_);
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

// ------------------------------------------------------------------------------------------------ CoarseTracker
namespace {

// the reference frame's features as CoarseTracker::makeDepthRef sees them (src/CoarseTracker.cpp:210-240): the distance of the
// point along the bearing, from its host-frame inverse depth; -1 keeps the slot of a feature without a usable point
// Written straight into the tracker kernel's layout: six arrays px[0] | px[1] | f[0] | f[1] | f[2] | dist of `stride` doubles.
void reference_features(const Seq& s, const Frame& R, double* out, size_t stride)
{
  const size_t n = s.n_feats(R);
  double* px0 = out; double* px1 = out + stride; double* f0 = out + 2 * stride; double* f1 = out + 3 * stride; double* f2 = out + 4 * stride;
  double* dist = out + 5 * stride;
  Id cached = kNone;
  SE3 T_ref_host;
  for (size_t i = 0; i < n; i++) {
    const Feat& ft = s.feat_of(R, i);
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

void trace_track(Trace& t, const hso_camera& cam, const hso_track_params& p, const hso_track_job& job, const hso_track_result& res)
{
  t.begin("coarse_track", 8);
  t.field("cam", &cam, sizeof(cam)); t.field("params", &p, sizeof(p));
  t.scalar("ref_frame_id", (double)job.ref_frame_id); t.scalar("cur_frame_id", (double)job.cur_frame_id);
  {
    // the trace keeps the record form of the table (what the value-passing call takes and the replay reads)
    const size_t n = (size_t)job.n_feats, st = (n + 31) & ~size_t(31);
    const double* a = reinterpret_cast<const double*>(job.feats);
    std::vector<hso_ref_feat> rec(n);
    for (size_t i = 0; i < n; i++) { rec[i].px[0] = a[i]; rec[i].px[1] = a[st + i]; rec[i].f[0] = a[2 * st + i]; rec[i].f[1] = a[3 * st + i]; rec[i].f[2] = a[4 * st + i]; rec[i].dist = a[5 * st + i]; }
    t.field("feats", rec.data(), sizeof(hso_ref_feat) * n);
  }
  t.field("T_cur_ref", &job.T_cur_ref, sizeof(hso_se3)); t.scalar("exposure_rat", job.exposure_rat);
  t.field("result", &res, sizeof(res));
}

}  // namespace

// run one tracker configuration over (reference, current) pairs of several sequences; results in StepData::track
void Bank::track_group(const std::vector<int>& who, const std::vector<Id>& ref, const std::vector<Id>& cur, const hso_track_params& p)
{
  if (who.empty()) return;
  std::vector<hso_track_job> jobs(who.size());
  std::vector<hso_track_result> res(who.size());
  // all jobs' feature tables in one page-locked block, back to back in the kernel's layout: the sequences fill their parts in
  // parallel and the block leaves in one DMA
  std::vector<size_t> at(who.size() + 1, 0);
  for (size_t i = 0; i < who.size(); i++) at[i + 1] = at[i] + 6 * ((seq_[who[i]]->n_feats(seq_[who[i]]->frames[ref[i]]) + 31) & ~size_t(31));
  double* const block = track_tables_.need(ctx_, at.back() + 64);
  pool_->run((int)who.size(), [&](int i) {
    Seq& s = *seq_[who[i]];
    const Frame& R = s.frames[ref[i]];
    const Frame& C = s.frames[cur[i]];
    reference_features(s, R, block + at[i], (at[i + 1] - at[i]) / 6);
    hso_track_job& j = jobs[i];
    j = hso_track_job{};
    j.ref_frame_id = R.dev_id; j.cur_frame_id = C.dev_id;
    j.feats = reinterpret_cast<const hso_ref_feat*>(block + at[i]); j.n_feats = (int)s.n_feats(R); j.feats_soa = 1;
    j.T_cur_ref = (C.T * R.T.inverse()).v;                        // src/CoarseTracker.cpp:63
    j.exposure_rat = C.integral / R.integral;                     // :60
  });
  check(hso_gpu_coarse_track_batch(ctx_, &cam_.pod(), &p, jobs.data(), (int)jobs.size(), res.data()), "CoarseTracker");
  n_calls_[2]++; n_items_[2] += (int64_t)who.size();
  for (size_t i = 0; i < who.size(); i++) {
    Seq& s = *seq_[who[i]];
    StepData& d = *step_[who[i]];
    d.track = res[i];
    d.job = jobs[i];
    // the write-back of CoarseTracker::run (:198-202)
    Frame& C = s.frames[cur[i]];
    const Frame& R = s.frames[ref[i]];
    SE3 T_cur_ref; T_cur_ref.v = res[i].T_cur_ref;
    C.T = T_cur_ref * R.T;
    C.exposure = (double)res[i].exposure_rat * R.exposure;
    if (res[i].exposure_rat > 0.99 && res[i].exposure_rat < 1.01) C.exposure = R.exposure;
    if (s.trace.on()) trace_track(s.trace, cam_.pod(), p, jobs[i], res[i]);
  }
}

void Bank::track(const std::vector<int>& who)
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
  std::vector<int> group[2]; std::vector<Id> ref[2], cur[2];
  for (int k : who) {
    Seq& s = *seq_[k];
    StepData& d = *step_[k];
    if (s.stage == kRelocalising && !d.relocalised) continue;
    Id last = d.relocalised ? d.ref : s.last;
    Frame& C = s.frames[s.cur];
    C.T = s.motion * s.frames[last].T;                            // processFrame, :176
    if (s.after_init) last = s.first;                             // :180
    d.ref = last;
    d.tracked = true;
    s.log = hso_vo_status{};
    if (s.n_feats(s.frames[last]) == 0) { d.track = hso_track_result{}; continue; }   // CoarseTracker::run returns 0 at once (:53-54)
    d.inverse = !(C.grad_mean > s.frames[last].grad_mean + 0.5f) ? 1 : 0;             // :184
    group[d.inverse].push_back(k); ref[d.inverse].push_back(last); cur[d.inverse].push_back(s.cur);
  }
  for (int mode = 0; mode < 2; mode++) {
    const hso_track_params p{mode, cfg_.klt_max_level, cfg_.klt_min_level + 1, 50};
    track_group(group[mode], ref[mode], cur[mode], p);
  }
  for (int k : who) if (step_[k]->tracked) { seq_[k]->log.n_tracked = step_[k]->track.n_tracked; seq_[k]->log.used_inverse = step_[k]->inverse; }
}

// ------------------------------------------------------------------------------------------------ Reprojector::reprojectMap
// Which points the frame projects, in the reference's visiting order (src/reprojector.cpp:98-254): the temporary points whose seed
// has finished are retired first; then the covisible keyframes of the last frame, then the keyframes that see the frame, nearest
// first, up to the keyframe budget — each contributing the points of its features once; then the candidates; then the temporary
// points.
void Bank::list_points(int k)
{
  Seq& s = *seq_[k];
  StepData& d = *step_[k];
  Frame& C = s.frames[s.cur];
  {
    size_t keep = 0;
    for (size_t i = 0; i < s.temps.size(); i++) {
      const Id p = s.temps[i];
      if (s.points[p].seed_state == 0) { s.temps[keep++] = p; continue; }
      s.retire_temp(p);
    }
    s.temps.resize(keep);
  }
  d.visit.clear(); d.list.clear(); d.list_q.clear();
  auto take = [&](Id kf) {
    Frame& K = s.frames[kf];
    K.visited = C.serial;
    d.visit.push_back(kf);
    for (Id f : K.fts) {
      const Id p = s.feats[f].point;
      if (p == kNone) continue;
      Point& P = s.points[p];
      if (P.kind == kPtTemporary || P.stamp == C.serial) continue;
      P.stamp = C.serial;
      d.list.push_back(p); d.list_q.push_back(quality_key(P));
    }
  };
  Frame& L = s.frames[d.ref];                                     // new_frame_->m_last_frame
  for (Id kf : L.covis) {
    const Frame& K = s.frames[kf];
    if (!K.in_use || K.kf_row < 0 || K.visited == C.serial) continue;
    if (std::find(s.kfs.begin(), s.kfs.end(), kf) == s.kfs.end()) continue;   // Map::getKeyframeById
    take(kf);
  }
  L.covis.clear();
  std::vector<std::pair<double, Id>> near;
  s.closest_keyframes(C, near);
  std::stable_sort(near.begin(), near.end(), [](const std::pair<double, Id>& a, const std::pair<double, Id>& b) { return a.first < b.first; });
  size_t n = d.visit.size();
  for (size_t i = 0; i < near.size() && n < (size_t)cfg_.reproject_max_kfs; i++) {
    if (s.frames[near[i].second].visited == C.serial) continue;
    take(near[i].second);
    ++n;
  }
  d.n_kf_points = (int)d.list.size();
  for (Id p : s.candidates) { d.list.push_back(p); d.list_q.push_back(quality_key(s.points[p])); }
  d.n_cand_listed = (int)s.candidates.size();
  for (Id p : s.temps) {
    Point& P = s.points[p];
    if (P.bad) continue;
    P.stamp = C.serial;
    s.place_in_host(p);
    d.list.push_back(p); d.list_q.push_back(quality_key(P));
  }
  hso_map_frame& c = d.call;
  c = hso_map_frame{};
  c.map = s.map; c.cur_keyframe_id = C.kf_id; c.cur_frame_id = C.dev_id; c.T_cur_w = C.T.v; c.cur_exposure_time = C.exposure;
  c.point_ids = d.list.data(); c.quality = d.list_q.data(); c.n_points = (int)d.list.size();
}

// what Reprojector::reprojectCell / reprojectCellAll do with the candidates they examine (:352-429, :556-612), applied to the
// examined records the device returned, in examination order; a record that became a feature adds it to the frame
void Bank::apply_selection(int k, const hso_frame_match* rec, int n_rec, const uint8_t* projected, const double* feat_f)
{
  Seq& s = *seq_[k];
  StepData& d = *step_[k];
  Frame& C = s.frames[s.cur];
  // points the projection rejected: candidates and temporary points pay for it (:214-222, :247-251)
  for (int i = d.n_kf_points; i < (int)d.list.size(); i++) {
    if (projected[i]) continue;
    Point& P = s.points[d.list[i]];
    P.n_fail += 3;
    if (P.n_fail <= 30) continue;
    if (i < d.n_kf_points + d.n_cand_listed) s.erase_candidate(d.list[i]); else P.bad = true;
  }
  C.loose.clear();
  C.loose.reserve((size_t)std::min(n_rec, cfg_.max_fts + 8));
  int taken = 0;
  for (int i = 0; i < n_rec; i++) {
    const hso_frame_match& r = rec[i];
    const Id p = d.list[r.point];
    Point& P = s.points[p];
    if (P.kind == kPtDeleted) continue;
    if (!r.success) {
      P.n_fail++;
      if (P.kind == kPtUnknown && P.n_fail > 15) s.erase_point(p);
      else if (P.kind == kPtCandidate && P.n_fail > 30) s.erase_candidate(p);
      else if (P.kind == kPtTemporary && P.n_fail > 30) P.bad = true;
      continue;
    }
    P.n_ok++;
    if (P.kind == kPtUnknown && P.n_ok > 10) P.kind = kPtGood;
    Feat nf;
    nf.frame = s.cur; nf.point = p;
    nf.px[0] = r.px_cur[0]; nf.px[1] = r.px_cur[1];
    nf.f[0] = feat_f[3 * taken]; nf.f[1] = feat_f[3 * taken + 1]; nf.f[2] = feat_f[3 * taken + 2];
    nf.level = r.search_level;
    if (r.ref_type == HSO_FTR_EDGELET) { nf.type = HSO_FTR_EDGELET; nf.grad[0] = r.grad[0]; nf.grad[1] = r.grad[1]; }
    else nf.type = r.ref_type == HSO_FTR_GRADIENT ? HSO_FTR_GRADIENT : HSO_FTR_CORNER;
    C.loose.push_back(nf);
    ++taken;
  }
}

void Bank::reproject(const std::vector<int>& who)
{
  const bool timing = getenv("HSO_ENGINE_TIMING") != nullptr;
  auto now = [] { return std::chrono::steady_clock::now(); };
  auto t0 = now();
  par(who, [&](int k) { list_points(k); });
  flush_maps(who);                                                // rows the listing touched (temporary points' positions) and earlier changes
  if (timing) { sub_ms_[0] += std::chrono::duration<double, std::milli>(now() - t0).count(); t0 = now(); }
  const int n = (int)who.size(), cap = std::max(cfg_.max_fts, 1);
  std::vector<hso_map_frame> calls(n);
  size_t total = 0;
  std::vector<size_t> list_at(n);
  for (int i = 0; i < n; i++) { calls[i] = step_[who[i]]->call; list_at[i] = total; total += (size_t)calls[i].n_points; }
  bool any_trace = false;
  for (int k : who) any_trace |= seq_[k]->trace.on();
  records_.need(ctx_, std::max(total, (size_t)1));
  if (any_trace) briefs_.need(ctx_, std::max(total, (size_t)1));   // the full 56-byte records only for the trace
  projected_.need(ctx_, std::max(total, (size_t)1));
  std::vector<int32_t> begin(n + 1, 0), counts(4 * (size_t)n, 0), n_feats(n, 0);
  std::vector<hso_pose_result> pose(n);
  mask_.need(ctx_, (size_t)n * cap);
  feat_f_.need(ctx_, (size_t)n * cap * 3);
  hso_pose_chain chain{};
  chain.reproj_thresh = cfg_.poseoptim_thresh; chain.n_iter = 12;
  chain.results = pose.data(); chain.n_feats = n_feats.data(); chain.outlier_mask = mask_.data(); chain.feat_f = feat_f_.data();
  chain.records = records_.data();
  const int rc = hso_gpu_reproject_select_pose_frames(ctx_, &cam_.pod(), calls.data(), n, cell_size_, grid_cols_, cell_order_.data(), (int)cell_order_.size(),
                                                      cfg_.max_fts, any_trace ? briefs_.data() : nullptr, (int)std::max(total, (size_t)1), begin.data(), counts.data(), projected_.data(), &chain);
  check(rc, "Reprojector");
  if (timing) { sub_ms_[1] += std::chrono::duration<double, std::milli>(now() - t0).count(); t0 = now(); }
  n_calls_[3]++; n_items_[3] += n;
  if (any_trace) trace_reproject(who, calls, list_at, begin, counts, pose, n_feats);
  pool_->run(n, [&](int i) {
    const int k = who[i];
    Seq& s = *seq_[k];
    StepData& d = *step_[k];
    Frame& C = s.frames[s.cur];
    apply_selection(k, records_.data() + begin[i], begin[i + 1] - begin[i], projected_.data() + list_at[i], feat_f_.data() + (size_t)i * cap * 3);
    s.log.n_trials = counts[4 * i]; s.log.n_matches = counts[4 * i + 1]; s.log.n_seed_matches = 0;
    d.pose = pose[i];
    d.pose_mask.assign(mask_.data() + (size_t)i * cap, mask_.data() + (size_t)i * cap + C.loose.size());
    // too few matches: the nearly converged seeds are tried as well (:309-329) — the frame's features change, so the pose
    // optimisation that ran behind the selection does not count for it
    d.seed_path = s.log.n_matches < 100 && !s.seeds.empty() && (int)s.seeds.size() > s.n_dead_seeds;
  });
  if (timing) sub_ms_[2] += std::chrono::duration<double, std::milli>(now() - t0).count();
}

// the frame's pose result (pose_optimizer::optimizeLevenbergMarquardt3rd's effects, src/pose_optimizer.cpp:692-767) and the
// decisions FrameHandlerMono::processFrame takes from it (:224-291), up to the choice between a regular frame and a keyframe
void Bank::decide(int k)
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
    for (size_t i = 0; i < C.loose.size() && i < d.pose_mask.size(); i++) if (d.pose_mask[i]) C.loose[i].point = kNone;
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
  d.make_kf = s.after_init || wants_keyframe(k);
  if (!d.make_kf) {
    link_covisible(k, false);
    s.outcome = kNoKeyframe;
    return;
  }
  // frame_utils::getSceneDepth / getSceneDistance (src/frame.cpp:323-366).  The reference computes them for every frame
  // (src/frame_handler_mono.cpp:268-271) but only a keyframe uses them (depth_filter_->addKeyframe, :335-338; needNewKf ignores its
  // depth argument): two medians over the frame's 2000 points were the largest single item of a regular frame's bookkeeping.
  {
    std::vector<double> z, r;
    z.reserve(C.loose.size()); r.reserve(C.loose.size());
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
// features — once with the full motion, once with its translation alone — weighted the way DSO weights them
bool Bank::wants_keyframe(int k)
{
  Seq& s = *seq_[k];
  if (s.regular < 3) return false;
  if (s.regular < std::min(3, int(s.n_mean_converge * 0.8))) return false;
  const Frame& C = s.frames[s.cur];
  const Frame& K = s.frames[s.kfs.back()];
  const SE3 T_cur_kf = C.T * K.T.inverse();
  const Vector3d kf_centre = s.centre(K);
  float flow_full = 0, flow_shift = 0;
  size_t count = 0;
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

// createCovisibilityGraph (:559-647): keyframes ranked by how many of the frame's points they observe
void Bank::link_covisible(int k, bool is_keyframe)
{
  Seq& s = *seq_[k];
  Frame& C = s.frames[s.cur];
  std::vector<int>& votes = s.votes;
  votes.assign(s.frames.size(), 0);
  std::vector<Id> seen;
  int with_point = 0;
  const size_t n = s.n_feats(C);
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
      P.kind = kPtUnknown; P.n_fail = 0;
      s.frames[s.feats[P.host].frame].fts.push_back(P.host);
    }
    s.candidates.resize(keep);
  }
  link_covisible(k, true);
  if (cfg_.loba_num_iter > 0) assemble_window(k);
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
  F.loose.clear(); F.loose.shrink_to_fit();
  s.refresh_keys(F);
  F.kf_id = ++s.n_kfs_made;
  s.hold(fr);                                                     // the map's reference
}

}  // namespace engine
}  // namespace hso
