// hso_engine_init.cpp — how a sequence starts: from a first keyframe with known depths (the setFirstFrame hook the reference
// keeps for synthetic data, src/frame_handler_mono.cpp:419-426) or from images alone (processFirstFrame / processSecondFrame,
// :125-172, with initialization::KltHomographyInit, src/initialization.cpp:39-223, on the engine's tables); and the trace records
// of the chained reprojection call.
#include "hso_engine_impl.h"
#include "hso_init.h"

namespace hso {
namespace engine {

// ------------------------------------------------------------------------------------------------ first keyframe with depths
void Bank::set_first_frames(const uint8_t* const* imgs, int w, int h, const double* stamps, const float* const* depth_z, const hso_se3* T_f_w)
{
  previous_collect();
  if (w != cam_.width() || h != cam_.height())
    throw Refused("Frame: provided image has not the same size as the camera model or image is not grayscale");
  for (int k = 0; k < size(); k++) if (imgs[k] && (!depth_z || !depth_z[k])) throw Refused("set_first_frame: no depth image");   // before anything is touched
  std::vector<int> who;
  for (int k = 0; k < size(); k++) {
    *step_[k] = StepData();
    if (!imgs[k]) continue;
    Seq& s = *seq_[k];
    for (Frame& F : s.frames) if (F.in_use && F.dev_id >= 0) to_release_.push_back(F.dev_id);
    drop_sequence_seeds(k);
    s.reset_tables();                                             // resetAll
    s.motion = SE3(); s.after_init = false; s.regular = 0; s.n_obs_last = 0; s.quality = kInsufficient; s.want_start = false;
    who.push_back(k);
  }
  release_queued();
  if (who.empty()) return;
  upload(who, imgs, w, h, stamps);
  // the detector of the initialisation (2000 features, FAST-12 hole filling), then one point per feature whose depth is known —
  // what the two-view initialisation leaves behind for its inliers
  std::vector<std::vector<hso_keypoint>> keys(who.size()), sel;
  std::vector<Id> frame(who.size()); std::vector<int> thresh(who.size());
  for (size_t i = 0; i < who.size(); i++) {
    Seq& s = *seq_[who[i]];
    Frame& C = s.frames[s.cur];
    if (T_f_w) C.T.v = T_f_w[who[i]];
    C.exposure = 1.0;                                             // processFirstFrame, :138
    frame[i] = s.cur; thresh[i] = (int)C.grad_mean;
  }
  detect(who, frame, thresh, true, cfg_.n_pyr_levels, 2000, keys, sel);
  std::vector<int> started, refused;
  for (size_t i = 0; i < who.size(); i++) {
    const int k = who[i];
    Seq& s = *seq_[k];
    StepData& d = *step_[k];
    Frame& C = s.frames[s.cur];
    const float* depth = depth_z[k];
    std::vector<double> dist_of;
    for (const hso_keypoint& kp : sel[i]) {
      Feat ft = feature_from_key(kp, s.cur);
      const int x = (int)ft.px[0], y = (int)ft.px[1];
      const float z = (x >= 0 && y >= 0 && x < w && y < h) ? depth[(size_t)y * w + x] : 0.f;
      if (!(z > 0)) continue;
      C.loose.push_back(ft);
      dist_of.push_back((double)z / ft.f[2]);                     // the point on the bearing whose depth along the optical axis is z
    }
    if (C.loose.size() < 10) {
      // this sequence cannot start: it goes back to the paused state with empty tables (like a failed two-view start, finish());
      // the others start, and the call reports the refusal once every table is in order
      to_release_.push_back(C.dev_id);
      s.reset_tables();
      s.stage = kPaused; s.quality = kInsufficient; s.outcome = kFailure;
      refused.push_back(k);
      continue;
    }
    started.push_back(k);
    make_keyframe(s, s.cur);
    const SE3 T_w_f = C.T.inverse();
    for (size_t j = 0; j < C.fts.size(); j++) {
      const Id f = C.fts[j];
      const Vector3d X = T_w_f * along(s.feats[f].f, dist_of[j]);
      const Id p = s.new_point(X, f, 1.0 / dist_of[j], kPtUnknown);
      s.feats[f].point = p;
      s.observe(p, f);
    }
    s.refresh_keys(C);
    s.kfs.push_back(s.cur);
    s.stage = kRunning;
    // the first keyframe's seeds: DepthFilter::addKeyframe with the scene's depth statistics
    std::vector<double> z, r;
    d.depth_min = std::numeric_limits<double>::max();
    for (Id f : C.fts) {
      const double* wpos = s.points[s.feats[f].point].pos;
      const Vector3d c = C.T * Vector3d{wpos[0], wpos[1], wpos[2]};
      z.push_back(c[2]); r.push_back(std::sqrt(c[0] * c[0] + c[1] * c[1] + c[2] * c[2]));
      d.depth_min = std::fmin(c[2], d.depth_min);
    }
    d.depth_mean = upper_median(z); d.dist_mean = upper_median(r);
    d.n_inliers = 1000;                                           // convergence threshold 200
    d.ok = true; d.make_kf = true; d.active = true;
    s.outcome = kKeyframe;
    s.log = hso_vo_status{};
  }
  who.swap(started);
  if (!who.empty()) {
    observe_seeds(who);                                           // no seeds yet: only the frame lists
    start_seeds(who);
    flush_maps(who);
  }
  for (int k : who) {
    Seq& s = *seq_[k];
    const Id old = s.last;
    s.last = s.cur; s.cur = kNone;
    if (old != kNone) release_frame(s, old);
    s.n_obs_last = 0;
    s.hist_stamp.push_back(s.frames[s.last].stamp); s.hist_pose.push_back(s.frames[s.last].T.v);
  }
  release_queued();
  if (!refused.empty()) {
    std::string msg = "set_first_frame: fewer than 10 features with a depth in sequence";
    for (int k : refused) msg += " " + std::to_string(k);
    throw Refused(msg + " (left paused; the other sequences started)");
  }
}

// ------------------------------------------------------------------------------------------------ two-view start
namespace {
bool inside(const AbstractCamera& cam, int x, int y, int margin) { return x >= margin && x < cam.width() - margin && y >= margin && y < cam.height() - margin; }
}

void Bank::initialise(const std::vector<int>& who)
{
  previous_collect();
  std::vector<int> first, second;
  for (int k : who) (seq_[k]->stage == kFirst ? first : second).push_back(k);
  if (!first.empty()) {
    // processFirstFrame + KltHomographyInit::addFirstFrame: the initialisation's detector on one level, 2000 features
    std::vector<std::vector<hso_keypoint>> keys(first.size()), sel;
    std::vector<Id> frame(first.size()); std::vector<int> thresh(first.size());
    for (size_t i = 0; i < first.size(); i++) {
      Seq& s = *seq_[first[i]];
      Frame& C = s.frames[s.cur];
      C.T = SE3();
      frame[i] = s.cur; thresh[i] = (int)(C.grad_mean + 0.5f);
    }
    detect(first, frame, thresh, true, 1, 2000, keys, sel);
    for (size_t i = 0; i < first.size(); i++) {
      Seq& s = *seq_[first[i]];
      Frame& C = s.frames[s.cur];
      s.log = hso_vo_status{};
      s.init.clear();
      if (sel[i].size() < 200) { s.outcome = kNoKeyframe; continue; }          // too few corners: wait for a better image (:45-49)
      for (const hso_keypoint& kp : sel[i]) {
        const Feat ft = feature_from_key(kp, s.cur);
        s.init.px_ref.push_back({(double)(float)ft.px[0], (double)(float)ft.px[1]});
        s.init.f_ref.push_back({ft.f[0], ft.f[1], ft.f[2]});
        s.init.kind.push_back({ft.grad[0], ft.grad[1], ft.type == HSO_FTR_EDGELET ? 1.0 : ft.type == HSO_FTR_CORNER ? 0.0 : 2.0});
      }
      s.init.px_cur = s.init.px_ref;
      s.init.ref = s.init.prev = s.cur;
      s.hold(s.cur); s.hold(s.cur);                               // frame_ref_, frame_prev_
      make_keyframe(s, s.cur);                                    // setKeyframe + Map::addKeyframe
      s.kfs.push_back(s.cur);
      s.first = s.cur; s.hold(s.cur);
      C.exposure = 1.0;
      s.stage = kSecond;
      s.outcome = kKeyframe;
    }
  }
  for (int k : second) {
    // KltHomographyInit::addSecondFrame: KLT from the previous frame (one device call per sequence: a handful per run)
    Seq& s = *seq_[k];
    TwoView& I = s.init;
    Frame& C = s.frames[s.cur];
    s.log = hso_vo_status{};
    const size_t n = I.px_cur.size();
    std::vector<float> a(2 * n), b(2 * n);
    // px_prev_ = the positions in the previous frame; while only the reference has been seen both are the reference's
    for (size_t i = 0; i < n; i++) { a[2 * i] = (float)I.px_cur[i][0]; a[2 * i + 1] = (float)I.px_cur[i][1]; b[2 * i] = a[2 * i]; b[2 * i + 1] = a[2 * i + 1]; }
    std::vector<hso_klt_result> res(n);
    hso_klt_params kp{};
    kp.win_size = 30; kp.max_level = 4; kp.max_iter = 30; kp.use_initial_flow = 1; kp.epsilon = 0.0001;
    check(hso_gpu_klt_track(ctx_, s.frames[I.prev].dev_id, C.dev_id, a.data(), b.data(), (int)n, &kp, res.data()), "trackKlt");
    n_calls_[9]++; n_items_[9]++;
    if (s.trace.on()) {
      Trace& t = s.trace;
      t.begin("klt_track", 6);
      t.scalar("prev_frame_id", (double)s.frames[I.prev].dev_id); t.scalar("cur_frame_id", (double)C.dev_id);
      t.field("px_prev", a.data(), sizeof(float) * a.size()); t.field("px_init", b.data(), sizeof(float) * b.size());
      t.field("params", &kp, sizeof(kp)); t.field("result", res.data(), sizeof(hso_klt_result) * n);
    }
    std::vector<double> disparity;
    {
      TwoView K;
      for (size_t i = 0; i < n; i++) {
        if ((res[i].status & (HSO_KLT_TRACKED | HSO_KLT_PATCH_OK)) != (HSO_KLT_TRACKED | HSO_KLT_PATCH_OK)) continue;
        const Vector2d pc = {(double)res[i].px[0], (double)res[i].px[1]};
        K.px_ref.push_back(I.px_ref[i]); K.px_cur.push_back(pc); K.f_ref.push_back(I.f_ref[i]); K.kind.push_back(I.kind[i]);
        K.f_cur.push_back(cam_.cam2world(pc));
        disparity.push_back(std::hypot(I.px_ref[i][0] - pc[0], I.px_ref[i][1] - pc[1]));
      }
      I.px_ref.swap(K.px_ref); I.px_cur.swap(K.px_cur); I.f_ref.swap(K.f_ref); I.f_cur.swap(K.f_cur); I.kind.swap(K.kind);
    }
    { const Id old = I.prev; I.prev = s.cur; s.hold(s.cur); release_frame(s, old); }
    if ((int)disparity.size() < cfg_.init_min_tracked) { s.outcome = kFailure; continue; }
    { std::vector<double> dd = disparity; if (upper_median(dd) < cfg_.init_min_disparity) { s.outcome = kNoKeyframe; continue; } }
    std::vector<int> inliers; std::vector<Vector3d> xyz; SE3 T_cur_ref; int used_h = 0;
    initialization::computeInitializeMatrix(I.f_ref, I.f_cur, cam_.errorMultiplier2(), cfg_.poseoptim_thresh, inliers, xyz, T_cur_ref, &used_h);
    if ((int)inliers.size() < cfg_.init_min_inliers) { s.outcome = kFailure; continue; }
    // the map is rescaled so that the median depth of the triangulated points equals map_scale (:97-104)
    std::vector<double> depths;
    for (const Vector3d& p : xyz) depths.push_back(p[2]);
    const double scale = cfg_.map_scale / upper_median(depths);
    Frame& R = s.frames[I.ref];
    C.T = T_cur_ref * R.T;
    {
      const Vector3d pr = s.centre(R), pc = s.centre(C);
      const Vector3d moved = {pr[0] + (pc[0] - pr[0]) * scale, pr[1] + (pc[1] - pr[1]) * scale, pr[2] + (pc[2] - pr[2]) * scale};
      SE3 rot = C.T; rot.v.t[0] = rot.v.t[1] = rot.v.t[2] = 0;
      const Vector3d t = rot * moved;
      C.T.v.t[0] = -t[0]; C.T.v.t[1] = -t[1]; C.T.v.t[2] = -t[2];
    }
    const SE3 T_w_cur = C.T.inverse();
    C.loose.clear();
    for (int id : inliers) {                                      // one point per inlier, observed in both frames (:110-170)
      const Vector2d pc = I.px_cur[id], pr = I.px_ref[id];
      if (!(inside(cam_, (int)pc[0], (int)pc[1], 10) && inside(cam_, (int)pr[0], (int)pr[1], 10) && xyz[id][2] > 0)) continue;
      const Vector3d pos = T_w_cur * Vector3d{xyz[id][0] * scale, xyz[id][1] * scale, xyz[id][2] * scale};
      const std::array<double, 3>& kind = I.kind[id];
      Feat in_ref, in_cur;
      in_ref.frame = I.ref; in_cur.frame = s.cur;
      in_ref.px[0] = pr[0]; in_ref.px[1] = pr[1]; in_cur.px[0] = pc[0]; in_cur.px[1] = pc[1];
      in_ref.type = in_cur.type = kind[2] == 0 ? HSO_FTR_CORNER : kind[2] == 1 ? HSO_FTR_EDGELET : HSO_FTR_GRADIENT;
      Vector3d fr = I.f_ref[id], fc = I.f_cur[id];
      if (in_ref.type == HSO_FTR_GRADIENT) { fr = cam_.cam2world(pr); fc = cam_.cam2world(pc); }
      if (in_ref.type == HSO_FTR_EDGELET) { in_ref.grad[0] = in_cur.grad[0] = kind[0]; in_ref.grad[1] = in_cur.grad[1] = kind[1]; }
      for (int c = 0; c < 3; c++) { in_ref.f[c] = fr[c]; in_cur.f[c] = fc[c]; }
      s.feats.push_back(in_ref);
      const Id f = (Id)s.feats.size() - 1;
      R.fts.push_back(f);
      s.list_grew(I.ref);
      const double pn[3] = {pos[0], pos[1], pos[2]};
      const Id p = s.new_point(pos, f, 1.0 / len3(pn), kPtUnknown);   // as written (:124): the reference frame sits at the origin
      s.feats[f].point = p;
      s.observe(p, f);
      in_cur.point = p;
      C.loose.push_back(in_cur);
    }
    release_frame(s, I.ref); release_frame(s, I.prev);
    I.clear();
    s.stage = kRunning;
    s.after_init = true;
    s.refresh_keys(R);                                            // firstFrame_->setKeyPoints()
    s.outcome = kKeyframe;
    s.log.n_matches = (int)C.loose.size();
    C.n_fts = (int32_t)C.loose.size();
    C.n_inliers = 0;
  }
}

// ------------------------------------------------------------------------------------------------ trace of the chain call
// One hso_gpu_seq_chain call = four records per traced sequence, with the tables AS THE DEVICE HOLDS THEM (read back through
// hso_gpu_seq_debug_* / hso_gpu_seqmap_read / hso_gpu_debug_fetch — the host mirror is not trusted here): "coarse_track" with the
// feature table the device built, "reproject_match" in the value-passing call's layout (the device's own point list, the
// observation lists flattened), "reproject_select" (the examined candidates) and "pose_optimize" (the feature table the device
// built from them).
// hso_vo_trace_state: the sequence map exactly as the device holds it right before a chain call, with the job and the call's
// configuration (tests/test_seq_chain.py rebuilds the state in a fresh context and in the restatement and compares the two calls)
namespace {
// a sequence map exactly as the device holds it (hso_gpu_seqmap_debug_dump), as eleven fields of a trace record
struct MapDump {
  int64_t sz[HSO_DUMP_N_SIZES];
  std::vector<hso_kf> kfs; std::vector<hso_map_point> pts; std::vector<hso_obs> obs;
  std::vector<int32_t> obs_pt, keys, nfts, lists, cands;
  std::vector<hso_seq_feature> ff0, ff1;
  static constexpr int kFields = 11;
  MapDump(hso_gpu_ctx* ctx, int map, const std::function<void(int, const char*)>& check)
  {
    check(hso_gpu_seqmap_debug_dump(ctx, map, HSO_DUMP_SIZES, sz, sizeof(sz)), "trace");
    const size_t nk = (size_t)sz[0], np = (size_t)sz[1], no = (size_t)sz[2], cap = (size_t)sz[3], nc = (size_t)sz[4];
    kfs.resize(nk); pts.resize(np); obs.resize(no); obs_pt.resize(no); keys.resize(5 * nk); nfts.resize(nk); lists.resize(nk * cap); cands.resize(nc);
    ff0.resize((size_t)sz[5]); ff1.resize((size_t)sz[6]);
    auto get = [&](int what, void* out, size_t bytes) { if (bytes) check(hso_gpu_seqmap_debug_dump(ctx, map, what, out, bytes), "trace"); };
    get(HSO_DUMP_KFS, kfs.data(), sizeof(hso_kf) * nk); get(HSO_DUMP_POINTS, pts.data(), sizeof(hso_map_point) * np); get(HSO_DUMP_OBS, obs.data(), sizeof(hso_obs) * no);
    get(HSO_DUMP_OBS_POINT, obs_pt.data(), 4 * no); get(HSO_DUMP_KEY_POINTS, keys.data(), 4 * keys.size()); get(HSO_DUMP_KF_NFTS, nfts.data(), 4 * nk);
    get(HSO_DUMP_KF_FTS, lists.data(), 4 * lists.size()); get(HSO_DUMP_CANDS, cands.data(), 4 * nc);
    get(HSO_DUMP_FRAME_FEATS0, ff0.data(), sizeof(hso_seq_feature) * ff0.size()); get(HSO_DUMP_FRAME_FEATS1, ff1.data(), sizeof(hso_seq_feature) * ff1.size());
  }
  void fields(Trace& t) const
  {
    t.field("sizes", sz, sizeof(sz)); t.field("kfs", kfs.data(), sizeof(hso_kf) * kfs.size()); t.field("points", pts.data(), sizeof(hso_map_point) * pts.size());
    t.field("obs", obs.data(), sizeof(hso_obs) * obs.size()); t.field("obs_point", obs_pt.data(), 4 * obs_pt.size()); t.field("key_points", keys.data(), 4 * keys.size());
    t.field("kf_nfts", nfts.data(), 4 * nfts.size()); t.field("kf_fts", lists.data(), 4 * lists.size()); t.field("cands", cands.data(), 4 * cands.size());
    t.field("frame_feats0", ff0.data(), sizeof(hso_seq_feature) * ff0.size()); t.field("frame_feats1", ff1.data(), sizeof(hso_seq_feature) * ff1.size());
  }
};
}  // namespace

void Bank::trace_chain_state(const std::vector<int>& who, const std::vector<hso_seq_job>& jobs, const hso_seq_chain_cfg& cfg, const std::vector<int32_t>& temps)
{
  for (size_t i = 0; i < who.size(); i++) {
    Seq& s = *seq_[who[i]];
    if (!s.trace.on() || !s.trace.state) continue;
    const MapDump dump(ctx_, s.map, [this](int rc, const char* what) { check(rc, what); });
    const hso_seq_job& jb = jobs[i];
    const int32_t none = 0;
    Trace& t = s.trace;
    t.begin("seq_chain_state", 6 + MapDump::kFields);
    t.field("cam", &cam_.pod(), sizeof(hso_camera)); t.field("job", &jb, sizeof(jb)); t.field("cfg", &cfg, sizeof(cfg));
    t.field("cell_order", cell_order_.data(), sizeof(int32_t) * cell_order_.size());
    t.field("temps", jb.n_temps > 0 ? temps.data() + jb.temps_begin : &none, sizeof(int32_t) * (size_t)jb.n_temps);
    dump.fields(t);
    t.scalar("max_fts", cfg_.max_fts);
  }
}

// the same before a hso_gpu_seq_local_ba call: the map, the window's core and what the call is asked (tests/test_seq_ba.py)
void Bank::trace_ba_state(const std::vector<int>& who, const std::vector<hso_seq_ba_job>& jobs, double error_multiplier2, double chi2_corner, double chi2_edgelet)
{
  for (size_t i = 0; i < who.size(); i++) {
    Seq& s = *seq_[who[i]];
    if (!s.trace.on() || !s.trace.state) continue;
    const MapDump dump(ctx_, s.map, [this](int rc, const char* what) { check(rc, what); });
    const hso_seq_ba_job& jb = jobs[i];
    Trace& t = s.trace;
    t.begin("seq_ba_state", 7 + MapDump::kFields);
    t.field("core", jb.core, sizeof(int32_t) * (size_t)jb.n_core); t.field("fixed", jb.fixed, (size_t)jb.n_core);
    t.scalar("n_iter", jb.n_iter); t.scalar("error_multiplier2", error_multiplier2); t.scalar("chi2_corner", chi2_corner); t.scalar("chi2_edgelet", chi2_edgelet);
    t.scalar("point_cap", jb.point_cap);
    dump.fields(t);
  }
}

void Bank::trace_chain(const std::vector<int>& who, const std::vector<hso_seq_job>& jobs, const hso_seq_chain_cfg& cfg, const hso_seq_result* res)
{
  const int n = (int)who.size(), cap = std::max(cfg_.max_fts, 1);
  std::vector<int32_t> slices((size_t)n + 1), ex_begin((size_t)n + 1);
  check(hso_gpu_debug_fetch(ctx_, HSO_DBG_SLICES, slices.data(), sizeof(int32_t) * slices.size()), "trace");
  check(hso_gpu_debug_fetch(ctx_, HSO_DBG_EXAMINED_BEGIN, ex_begin.data(), sizeof(int32_t) * ex_begin.size()), "trace");
  const size_t total = (size_t)slices[(size_t)n], n_exam = (size_t)ex_begin[(size_t)n];
  std::vector<hso_reproj_point> proj(total); std::vector<hso_align_out> match(total); std::vector<uint8_t> projected(total);
  std::vector<hso_match_brief> briefs(std::max(n_exam, (size_t)1));
  std::vector<hso_pose_feat> pf((size_t)n * cap); std::vector<hso_se3> pp((size_t)n * 128); std::vector<int32_t> np((size_t)n); std::vector<uint8_t> pmask((size_t)n * cap);
  if (total) {
    check(hso_gpu_debug_fetch(ctx_, HSO_DBG_PROJ, proj.data(), sizeof(hso_reproj_point) * total), "trace");
    check(hso_gpu_debug_fetch(ctx_, HSO_DBG_MATCH, match.data(), sizeof(hso_align_out) * total), "trace");
    check(hso_gpu_debug_fetch(ctx_, HSO_DBG_PROJECTED, projected.data(), total), "trace");
  }
  if (n_exam) check(hso_gpu_debug_fetch(ctx_, HSO_DBG_BRIEF, briefs.data(), sizeof(hso_match_brief) * n_exam), "trace");
  check(hso_gpu_debug_fetch(ctx_, HSO_DBG_POSE_FEATS, pf.data(), sizeof(hso_pose_feat) * pf.size()), "trace");
  check(hso_gpu_debug_fetch(ctx_, HSO_DBG_POSE_POSES, pp.data(), sizeof(hso_se3) * pp.size()), "trace");
  check(hso_gpu_debug_fetch(ctx_, HSO_DBG_POSE_NPOSES, np.data(), sizeof(int32_t) * np.size()), "trace");
  check(hso_gpu_debug_fetch(ctx_, HSO_DBG_POSE_MASK, pmask.data(), pmask.size()), "trace");
  for (int i = 0; i < n; i++) {
    Seq& s = *seq_[who[(size_t)i]];
    if (!s.trace.on()) continue;
    const hso_seq_job& jb = jobs[(size_t)i];
    const hso_seq_result& r = res[i];
    Trace& t = s.trace;
    if (s.trace.state) {
      std::vector<int32_t> ev((size_t)std::max(r.n_events, 1));
      if (r.n_events > 0) check(hso_gpu_seq_events(ctx_, i, ev.data(), (int)ev.size()), "trace");
      std::vector<hso_seq_feature> ff((size_t)std::max(cap, 1));
      int32_t n_ff = 0;
      check(hso_gpu_seq_frame_features(ctx_, &s.map, &jb.cur_frame_id, 1, ff.data(), (int)ff.size(), &n_ff), "trace");
      t.begin("seq_chain_result", 3);
      t.field("result", &r, sizeof(r)); t.field("events", ev.data(), sizeof(int32_t) * (size_t)r.n_events); t.field("features", ff.data(), sizeof(hso_seq_feature) * (size_t)n_ff);
    }
    // ---- CoarseTracker::run over the table the device built
    if (!(jb.flags & HSO_SEQ_NO_TRACK)) {
      std::vector<hso_ref_feat> rec((size_t)std::max(jb.n_ref_feats, 1));
      const int got = hso_gpu_seq_debug_ref_table(ctx_, i, rec.data(), (int)rec.size());
      check(got, "trace");
      const SE3 Tc{jb.T_cur_w}, Tr{jb.T_ref_w};
      const hso_se3 T_cur_ref = (Tc * Tr.inverse()).v;
      t.begin("coarse_track", 8);
      t.field("cam", &cam_.pod(), sizeof(hso_camera)); t.field("params", &cfg.track, sizeof(cfg.track));
      t.scalar("ref_frame_id", (double)jb.ref_frame_id); t.scalar("cur_frame_id", (double)jb.cur_frame_id);
      t.field("feats", rec.data(), sizeof(hso_ref_feat) * (size_t)got);
      t.field("T_cur_ref", &T_cur_ref, sizeof(hso_se3)); t.scalar("exposure_rat", jb.exposure_rat);
      t.field("result", &r.track, sizeof(r.track));
    }
    // ---- the list the device walked, and what became of its points
    const int nl = r.n_listed;
    std::vector<int32_t> ids((size_t)std::max(nl, 1)); std::vector<uint8_t> quality((size_t)std::max(nl, 1));
    check(hso_gpu_seq_debug_list(ctx_, i, ids.data(), quality.data(), (int)ids.size()), "trace");
    int nk = 0, n_pts = 0, n_obs = 0;
    check(hso_gpu_seqmap_size(ctx_, s.map, &nk, &n_pts, &n_obs), "trace");
    std::vector<hso_map_point> rows((size_t)nl);
    std::vector<int32_t> all_obs((size_t)n_obs);
    std::iota(all_obs.begin(), all_obs.end(), 0);
    std::vector<hso_obs> obs_rows((size_t)n_obs);
    check(hso_gpu_seqmap_read(ctx_, s.map, ids.data(), nl, rows.data(), all_obs.data(), n_obs, obs_rows.data()), "trace");
    std::vector<hso_kf> kfs;
    for (Id fr : s.dev_kfs) { const Frame& F = s.frames[fr]; hso_kf k{}; k.frame_id = F.dev_id; k.T_f_w = F.T.v; k.exposure_time = F.exposure; k.keyframe_id = F.kf_id; kfs.push_back(k); }
    std::vector<hso_obs> flat;
    const size_t at0 = (size_t)slices[(size_t)i];
    std::vector<hso_reproj_point> pr(proj.begin() + (std::ptrdiff_t)at0, proj.begin() + (std::ptrdiff_t)(at0 + (size_t)nl));
    for (int j = 0; j < nl; j++) {
      hso_map_point& row = rows[(size_t)j];
      const int first = (int)flat.size();
      int at = -1;
      for (int q = 0, orow = row.obs_begin; q < row.obs_count; q++) {
        hso_obs o = obs_rows[(size_t)orow];
        if (pr[(size_t)j].ref_obs == orow) at = first + q;
        orow = o.pad_; o.pad_ = 0;
        flat.push_back(o);
      }
      row.obs_begin = first;
      row.pad_ = quality[(size_t)j];
      if (pr[(size_t)j].ref_obs >= 0) pr[(size_t)j].ref_obs = at;
      pr[(size_t)j].pad_ = 0;
    }
    hso_obs none_obs{};
    t.begin("reproject_match", 12);
    t.field("cam", &cam_.pod(), sizeof(hso_camera)); t.scalar("cur_frame_id", (double)jb.cur_frame_id); t.field("T_cur_w", &r.T_tracked, sizeof(hso_se3));
    t.scalar("cur_exposure", r.exposure); t.scalar("cur_keyframe_id", jb.cur_keyframe_id);
    t.field("kfs", kfs.data(), sizeof(hso_kf) * kfs.size()); t.field("points", rows.data(), sizeof(hso_map_point) * rows.size());
    t.field("obs", flat.empty() ? &none_obs : flat.data(), sizeof(hso_obs) * flat.size()); t.scalar("cell_size", cell_size_); t.scalar("grid_n_cols", grid_cols_);
    t.field("proj", pr.data(), sizeof(hso_reproj_point) * pr.size()); t.field("match", match.data() + at0, sizeof(hso_align_out) * (size_t)nl);
    t.begin("reproject_select", 6);
    t.field("quality", quality.data(), (size_t)nl); t.field("cell_order", cell_order_.data(), sizeof(int32_t) * cell_order_.size());
    t.scalar("max_fts", cfg_.max_fts); t.field("examined", briefs.data() + ex_begin[(size_t)i], sizeof(hso_match_brief) * (size_t)(ex_begin[(size_t)i + 1] - ex_begin[(size_t)i]));
    t.field("counts", r.counts, sizeof(int32_t) * 4); t.field("projected", projected.data() + at0, (size_t)nl);
    t.begin("pose_optimize", 8);
    t.field("cam", &cam_.pod(), sizeof(hso_camera)); t.field("feats", pf.data() + (size_t)i * cap, sizeof(hso_pose_feat) * (size_t)r.n_feats);
    t.field("poses", pp.data() + (size_t)i * 128, sizeof(hso_se3) * (size_t)std::min(np[(size_t)i], 128)); t.field("T_f_w", &r.T_tracked, sizeof(hso_se3));
    t.scalar("reproj_thresh", cfg_.poseoptim_thresh); t.scalar("n_iter", 12);
    t.field("result", &r.pose, sizeof(hso_pose_result)); t.field("mask", pmask.data() + (size_t)i * cap, (size_t)r.n_feats);
  }
}

}  // namespace engine
}  // namespace hso
