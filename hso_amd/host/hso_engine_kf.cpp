// hso_engine_kf.cpp — the keyframe-rate half of a step: the local BA window, the depth filter's seed bookkeeping around the
// resident seed table, new seeds, the mirror of the sequence tables on the device, the end of a frame.
#include "hso_engine_impl.h"
#include <optional>

namespace hso {
namespace engine {

// ------------------------------------------------------------------------------------------------ local BA
// ba::LocalBundleAdjustment (src/bundle_adjustment.cpp:577-892) runs on the sequence's resident map (hso_gpu_seq_local_ba): the
// engine names the window's core keyframes — the reference walks std::set<Frame*> in address order; frame serials give a
// run-independent one — and which of them are fixed (:595), and mirrors what comes back.
void Bank::window_job(int k)
{
  Seq& s = *seq_[k];
  StepData& d = *step_[k];
  const Frame& C = s.frames[s.cur];
  std::vector<Id> core = s.local_map;
  std::sort(core.begin(), core.end(), [&](Id a, Id b) { return s.frames[a].serial < s.frames[b].serial; });
  if (core.size() > HSO_SEQ_BA_MAX_CORE) throw Refused("LocalBundleAdjustment: more core keyframes than the reduced system holds");
  d.ba_frames = core;
  d.ba_fixed.clear();
  size_t cap = 0;
  for (Id kf : core) {
    const Frame& K = s.frames[kf];
    d.ba_fixed.push_back((K.serial == 0 || K.kf_id + 20 < C.kf_id) ? 1 : 0);   // :595
    cap += K.fts.size();
  }
  d.ba_points.assign(std::min(cap, s.points.size()), kNone);
  d.ba_state.assign(4 * d.ba_points.size(), 0.0);
  d.ba_culled.assign(s.feats.size(), kNone);
  d.ba_iters = 100;                                               // :815-823
  if (s.kfs.size() > 5) d.ba_iters = C.fts.size() < 100 ? cfg_.loba_num_iter + 10 : cfg_.loba_num_iter;
  d.n_core = (int)core.size();
}

void Bank::keyframe_ba(const std::vector<int>& all)
{
  previous_collect();                                              // the keyframes' new poses go to the seed table below
  std::vector<int> who;
  for (int k : all) if (step_[k]->n_core > 0) who.push_back(k);
  const double fmean = cam_.errorMultiplier2();
  const double tight = 1.2 / fmean, loose = 2.0 / fmean;           // :855-892
  std::vector<hso_seq_ba_job> jobs(who.size());
  std::vector<hso_seq_ba_result> res(who.size());
  for (size_t i = 0; i < who.size(); i++) {
    Seq& s = *seq_[who[i]];
    StepData& d = *step_[who[i]];
    hso_seq_ba_job& J = jobs[i];
    memset(&J, 0, sizeof(J));
    J.map = s.map; J.n_core = d.n_core;
    for (int c = 0; c < d.n_core; c++) { J.core[c] = s.frames[d.ba_frames[(size_t)c]].kf_row; J.fixed[c] = d.ba_fixed[(size_t)c]; }
    J.n_iter = d.ba_iters;
    J.point_cap = (int32_t)d.ba_points.size(); J.cull_cap = (int32_t)d.ba_culled.size();
    J.point_ids = d.ba_points.data(); J.point_state = d.ba_state.data(); J.culled = d.ba_culled.data();
  }
  trace_ba_state(who, jobs, fmean, loose * loose, tight * tight);
  if (!who.empty()) {
    Sub t(this, "ba: local call");
    check(hso_gpu_seq_local_ba(ctx_, jobs.data(), (int)jobs.size(), fmean, loose * loose, tight * tight, res.data()), "LocalBundleAdjustment");
    n_calls_[8]++; n_items_[8] += (int64_t)jobs.size();
  }
  std::vector<int> with;
  for (size_t i = 0; i < who.size(); i++) {
    StepData& d = *step_[who[i]];
    d.ba = res[i];
    d.ba_points.resize((size_t)res[i].n_points);
    d.ba_culled.resize((size_t)std::min<int64_t>((int64_t)res[i].n_culled[0] + res[i].n_culled[1], (int64_t)d.ba_culled.size()));
    d.huber_corner = res[i].huber_corner; d.huber_edge = res[i].huber_edge; d.ba_res = res[i].lm;
    if (res[i].status == 0) with.push_back(who[i]);
  }
  // a traced sequence records the window the device assembled and what the optimisation made of it (the records the value-passing
  // calls hso_gpu_ba_huber_deltas / hso_gpu_ba_optimize are replayed from)
  for (size_t i = 0; i < who.size(); i++) {
    Seq& s = *seq_[who[i]];
    StepData& d = *step_[who[i]];
    if (!s.trace.on() || res[i].status != 0) continue;
    const size_t np = (size_t)res[i].n_poses, ne = (size_t)res[i].n_edges, npt = (size_t)res[i].n_points;
    std::vector<hso_se3> poses_in(np), poses_out(np); std::vector<uint8_t> fixed(np); std::vector<hso_ba_edge> edges(ne); std::vector<double> uv(2 * ne), chi2(ne), idist_in(npt), idist_out(npt);
    std::vector<int32_t> rows(np), edge_obs(ne);
    auto get = [&](int what, void* out, size_t bytes) { check(hso_gpu_seq_ba_debug_window(ctx_, (int)i, what, out, bytes), "LocalBundleAdjustment (trace)"); };
    get(HSO_BAW_VERTEX_ROWS, rows.data(), sizeof(int32_t) * np); get(HSO_BAW_FIXED, fixed.data(), np); get(HSO_BAW_EDGES, edges.data(), sizeof(hso_ba_edge) * ne);
    get(HSO_BAW_OBS_UV, uv.data(), sizeof(double) * 2 * ne); get(HSO_BAW_EDGE_OBS, edge_obs.data(), sizeof(int32_t) * ne); get(HSO_BAW_EDGE_CHI2, chi2.data(), sizeof(double) * ne);
    get(HSO_BAW_POSES_OUT, poses_out.data(), sizeof(hso_se3) * np); get(HSO_BAW_POSES_IN, poses_in.data(), sizeof(hso_se3) * np); get(HSO_BAW_IDIST_IN, idist_in.data(), sizeof(double) * npt);
    for (size_t q = 0; q < npt; q++) idist_out[q] = d.ba_state[4 * q];
    Trace& t = s.trace;
    t.begin("ba_huber_deltas", 7);
    t.field("poses", poses_in.data(), sizeof(hso_se3) * np); t.field("idist", idist_in.data(), sizeof(double) * npt);
    t.field("edges", edges.data(), sizeof(hso_ba_edge) * ne); t.field("obs_uv", uv.data(), sizeof(double) * 2 * ne);
    t.scalar("error_multiplier2", fmean); t.scalar("huber_corner", d.huber_corner); t.scalar("huber_edge", d.huber_edge);
    t.begin("ba_optimize", 14);
    t.field("poses_in", poses_in.data(), sizeof(hso_se3) * np); t.field("fixed", fixed.data(), np);
    t.field("idist_in", idist_in.data(), sizeof(double) * npt); t.field("edges", edges.data(), sizeof(hso_ba_edge) * ne);
    t.scalar("huber_corner", d.huber_corner); t.scalar("huber_edge", d.huber_edge); t.scalar("n_iter", d.ba_iters);
    t.field("poses_out", poses_out.data(), sizeof(hso_se3) * np); t.field("idist_out", idist_out.data(), sizeof(double) * npt);
    t.field("edge_chi2", chi2.data(), sizeof(double) * ne); t.field("result", &d.ba_res, sizeof(d.ba_res));
    t.field("vertex_rows", rows.data(), sizeof(int32_t) * np); t.field("edge_obs", edge_obs.data(), sizeof(int32_t) * ne);
    t.field("point_rows", d.ba_points.data(), sizeof(int32_t) * npt);
    if (s.trace.state) {
      t.begin("seq_ba_result", 4);
      t.field("result", &res[i], sizeof(res[i])); t.field("point_ids", d.ba_points.data(), sizeof(int32_t) * npt);
      t.field("point_state", d.ba_state.data(), sizeof(double) * 4 * npt); t.field("culled", d.ba_culled.data(), sizeof(int32_t) * d.ba_culled.size());
    }
  }
  { Sub t(this, "ba: apply_window"); par(with, [&](int k) { apply_window(k); }); }
  Sub t_tail(this, "ba: keys + seed poses");
  // setKeyPoints of the overlap keyframes (src/frame_handler_mono.cpp:331)
  par(all, [&](int k) { Seq& s = *seq_[k]; for (Id kf : step_[k]->visit) s.refresh_keys(s.frames[kf]); });
  // the resident seeds of the moved keyframes follow them
  std::vector<int64_t> ids; std::vector<hso_se3> poses;
  for (int k : with) for (Id kf : step_[k]->moved_kfs) { ids.push_back(seq_[k]->frames[kf].dev_id); poses.push_back(seq_[k]->frames[kf].T.v); }
  if (!ids.empty()) check(hso_gpu_seed_table_set_host_pose(ctx_, seed_table_, ids.data(), poses.data(), (int)ids.size()), "DepthFilter");
}

// what LocalBundleAdjustment does with the optimiser's result (:826-892), on the engine's mirror of the map: the device has written
// its own rows (poses, idist_, pos_), so nothing of it is marked for the next patch
void Bank::apply_window(int k)
{
  Seq& s = *seq_[k];
  StepData& d = *step_[k];
  const double fmean = cam_.errorMultiplier2();
  d.moved_kfs.clear();
  for (int i = 0; i < d.n_core; i++) {
    const Id kf = d.ba_frames[(size_t)i];
    s.frames[kf].T.v = d.ba.core_pose[i];
    if (!d.ba_fixed[(size_t)i]) d.moved_kfs.push_back(kf);
  }
  s.kfs_dirty = true;
  for (size_t i = 0; i < d.ba_points.size(); i++) {
    Point& P = s.points[d.ba_points[i]];
    const double* st = &d.ba_state[4 * i];
    P.idist = st[0]; P.pos[0] = st[1]; P.pos[1] = st[2]; P.pos[2] = st[3];
    P.n_ba++;
  }
  // MapPointCandidates::changeCandidatePosition: the candidates hosted in the window's keyframes follow them (pos_ = T_host^-1 *
  // (f / idist); rows the engine patches)
  {
    std::vector<int8_t> core(s.frames.size(), 0), have(s.frames.size(), 0);
    std::vector<SE3> inv(s.frames.size());
    for (int i = 0; i < d.n_core; i++) core[(size_t)d.ba_frames[(size_t)i]] = 1;
    for (Id c : s.candidates) {
      Point& P = s.points[c];
      const Id h = P.host_frame;
      if (!core[(size_t)s.feats[P.host].frame]) continue;
      if (!have[(size_t)h]) { inv[(size_t)h] = s.frames[h].T.inverse(); have[(size_t)h] = 1; }
      const Vector3d w = inv[(size_t)h] * along(P.host_f, 1.0 / P.idist);
      P.pos[0] = w[0]; P.pos[1] = w[1]; P.pos[2] = w[2];
      s.touch_point(c);
    }
  }
  // the observations whose edge the optimisation left above the threshold, corner edges first (:855-892)
  int dropped[2] = {0, 0};
  for (size_t e = 0; e < d.ba_culled.size(); e++) {
    const int pass = (int64_t)e < (int64_t)d.ba.n_culled[0] ? 0 : 1;
    const Id f = d.ba_culled[e];
    const Id p = s.feats[f].point;
    if (p == kNone) continue;
    if (s.points[p].kind == kPtTemporary) { s.points[p].bad = true; continue; }
    s.detach(s.feats[f].frame, f);
    dropped[pass]++;
  }
  s.log.ba_removed_1 = dropped[0]; s.log.ba_removed_2 = dropped[1];
  s.log.ba_error_init = std::sqrt(d.ba_res.init_chi2) * fmean;
  s.log.ba_error_final = std::sqrt(d.ba_res.final_chi2) * fmean;
}

// ------------------------------------------------------------------------------------------------ depth filter
namespace {

hso_seed seed_record(const Seq& s, const Seed& sd)
{
  const Feat& ft = s.feats[sd.feat];
  const Frame& H = s.frames[ft.frame];
  hso_seed h{};
  h.ref_frame_id = H.dev_id;
  h.level = ft.level; h.type = ft.type;
  h.px[0] = ft.px[0]; h.px[1] = ft.px[1];
  h.f[0] = ft.f[0]; h.f[1] = ft.f[1]; h.f[2] = ft.f[2];
  h.grad[0] = ft.grad[0]; h.grad[1] = ft.grad[1];
  h.T_ref_w = H.T.v; h.ref_exposure = H.exposure;
  h.mu = sd.mu; h.sigma2 = sd.sigma2; h.b = sd.b;
  return h;
}

}  // namespace

void Bank::kill_seed(Seq& s, StepData& d, int i, bool keep_feature)
{
  Seed& sd = s.seeds[i];
  if (!sd.alive) return;
  sd.alive = false; s.n_dead_seeds++;
  if (sd.slot >= 0) d.erase_slots.push_back(sd.slot);
  for (Id fr : sd.seen) release_frame_deferred(s, d, fr);
  for (Id fr : sd.seen_before) release_frame_deferred(s, d, fr);
  sd.seen.clear(); sd.seen_before.clear();
  (void)keep_feature;
}

void Bank::release_frame_deferred(Seq& s, StepData& d, Id fr)
{
  if (fr == kNone) return;
  const int64_t dev = s.frames[fr].dev_id;
  if (s.drop(fr)) d.released.push_back(dev);
}

void Bank::drop_sequence_seeds(int k)
{
  previous_collect();
  Seq& s = *seq_[k];
  std::vector<int32_t> slots;
  for (Seed& sd : s.seeds) if (sd.alive && sd.slot >= 0) slots.push_back(sd.slot);
  if (!slots.empty()) check(hso_gpu_seed_table_erase(ctx_, seed_table_, slots.data(), (int)slots.size()), "DepthFilter");
  s.seeds.clear(); s.n_dead_seeds = 0; s.check_seeds.clear(); s.check_all = true;
}

// DepthFilter::updateSeeds (src/depth_filter.cpp:330-509) for the frames of all sequences: the bookkeeping before the
// observation, ONE observation of every live seed in its sequence's frame, the bookkeeping after it
void Bank::observe_seeds(const std::vector<int>& who)
{
  previous_collect();
  par(who, [&](int k) {
    Seq& s = *seq_[k];
    StepData& d = *step_[k];
    // frame_prior_ (:332-350): the frames between keyframes, newest first; a keyframe opens the next batch's list
    const int slot = d.make_kf ? s.batch + 1 : s.batch;
    auto it = std::find_if(s.prior.begin(), s.prior.end(), [&](const std::pair<int, std::vector<Id>>& p) { return p.first == slot; });
    if (it == s.prior.end()) { s.prior.emplace_back(slot, std::vector<Id>()); it = s.prior.end() - 1; }
    it->second.insert(it->second.begin(), s.cur);
    s.hold(s.cur);
    for (size_t i = 0; i < s.prior.size();) {
      if (s.prior[i].first + 5 > s.batch || !d.make_kf) { ++i; continue; }
      for (Id fr : s.prior[i].second) release_frame_deferred(s, d, fr);
      s.prior.erase(s.prior.begin() + (std::ptrdiff_t)i);
    }
    // seeds older than max_n_kfs keyframes (:368-401)
    for (size_t i = 0; i < s.seeds.size(); i++) {                   // the list is in batch order: the old ones are its head
      Seed& sd = s.seeds[i];
      if (s.batch - sd.batch <= cfg_.seed_max_kfs) break;
      if (!sd.alive) continue;
      if (sd.temp != kNone && sd.reprojected) s.points[sd.temp].seed_state = -1;
      kill_seed(s, d, (int)i, sd.temp != kNone && sd.reprojected);
    }
  });
  erase_slots(who);
  int n_slots = 0, n_live = 0;
  check(hso_gpu_seed_table_size(ctx_, seed_table_, &n_slots, &n_live), "DepthFilter");
  if (n_slots == 0 || n_live == 0) return;
  // the sequences whose frame was a regular one had their seeds observed behind the frame by the chain itself (hso_seq_chain_cfg:
  // seed_table); what is left for a call here are the keyframes (observed after local BA moved their pose) and the frames whose
  // features the host changed (the seed branch)
  std::vector<int> rest;
  for (int k : who) if (!step_[k]->seeds_observed) rest.push_back(k);
  std::vector<hso_seed_frame> frames(seq_.size());
  for (hso_seed_frame& f : frames) { f = hso_seed_frame{}; f.frame_id = -1; f.T_f_w = hso_se3{{0, 0, 0, 1}, {0, 0, 0}}; f.exposure_time = 1; }
  bool want_px = false, tracing = false;
  for (int k : rest) {
    const Seq& s = *seq_[k];
    const Frame& C = s.frames[s.cur];
    frames[k].frame_id = C.dev_id; frames[k].T_f_w = C.T.v; frames[k].exposure_time = C.exposure;
    want_px |= step_[k]->make_kf;
    tracing |= s.trace.on();
  }
  std::vector<hso_seed> before; std::vector<hso_seed_out> full;
  if (!rest.empty()) {
    seed_brief_.need(ctx_, (size_t)n_slots);
    if (want_px) seed_px_.need(ctx_, 2 * (size_t)n_slots);
    if (tracing) {
      before.resize((size_t)n_slots); full.resize((size_t)n_slots);
      check(hso_gpu_seed_table_read(ctx_, seed_table_, 0, n_slots, before.data()), "DepthFilter");
    }
    check(hso_gpu_seed_table_observe_groups(ctx_, &cam_.pod(), seed_table_, frames.data(), (int)frames.size(), px_error_angle_, seed_brief_.data(),
                                            want_px ? seed_px_.data() : nullptr, tracing ? full.data() : nullptr), "DepthFilter");
  }
  if (!rest.empty()) { n_calls_[6]++; n_items_[6] += (int64_t)rest.size(); }
  alg_bytes_[4] += (double)n_live * 13.3 * 64 * 4;                 // per seed: 13.3 epipolar steps of 64 samples (the measured mean)
  par(who, [&](int k) {
    Seq& s = *seq_[k];
    StepData& d = *step_[k];
    const Frame& C = s.frames[s.cur];
    if (s.trace.on()) {
      std::vector<hso_seed> in; std::vector<hso_seed_out> out;
      for (const Seed& sd : s.seeds) if (sd.alive && sd.slot >= 0 && sd.slot < n_slots) { in.push_back(before[sd.slot]); out.push_back(full[sd.slot]); }
      Trace& t = s.trace;
      t.begin("seed_observe", 7);
      t.field("cam", &cam_.pod(), sizeof(hso_camera)); t.scalar("cur_frame_id", (double)C.dev_id); t.field("T_f_w", &C.T.v, sizeof(hso_se3));
      t.scalar("exposure", C.exposure); t.scalar("px_error_angle", px_error_angle_);
      t.field("seeds", in.data(), sizeof(hso_seed) * in.size()); t.field("out", out.data(), sizeof(hso_seed_out) * out.size());
    }
    // DepthFilter::observeDepthRow's effects on the seed (:593-673)
    d.occupied.clear();
    for (Seed& sd : s.seeds) {
      if (!sd.alive || sd.slot < 0 || sd.slot >= n_slots) continue;
      const hso_seed_brief& o = (d.seeds_observed ? d.chain_brief : seed_brief_.data())[sd.slot];
      sd.updated = o.is_update != 0;
      if (!sd.updated) continue;
      if (sd.seen.size() < 15) { sd.seen.push_back(s.cur); s.hold(s.cur); }
      if (!o.is_valid) sd.valid = false;
      sd.mu = o.mu; sd.sigma2 = o.sigma2; sd.b = o.b;
      if (!sd.valid || Seq::seed_converged(sd)) s.check_seeds.push_back((int)(&sd - s.seeds.data()));
      if (o.result != 1) continue;
      sd.n_dist++;
      if (d.make_kf) {                                            // FeatureExtractor::setGridOccpuancy
        hso_keypoint kp{};
        kp.x = seed_px_.data()[2 * (size_t)sd.slot]; kp.y = seed_px_.data()[2 * (size_t)sd.slot + 1]; kp.species = HSO_KP_OCCUR;
        d.occupied.push_back(kp);
      }
    }
  });
}

void Bank::erase_slots(const std::vector<int>& who)
{
  std::vector<int32_t> slots;
  for (int k : who) {
    StepData& d = *step_[k];
    slots.insert(slots.end(), d.erase_slots.begin(), d.erase_slots.end());
    d.erase_slots.clear();
    to_release_.insert(to_release_.end(), d.released.begin(), d.released.end());
    d.released.clear();
  }
  if (!slots.empty()) check(hso_gpu_seed_table_erase(ctx_, seed_table_, slots.data(), (int)slots.size()), "DepthFilter");
}

// the convergence loop of updateSeeds (:405-497): activatePoint (+ seedOptimizer) for every converged seed of every sequence in
// one device call, then the new candidate points
void Bank::activate_seeds(const std::vector<int>& who)
{
  // pass 1 (pool): which seeds converged, which frames they were seen in (each named once per sequence), how many pairs that makes
  std::vector<size_t> n_tg(who.size(), 0);
  pool_->run((int)who.size(), [&](int w) {
    Seq& s = *seq_[who[w]];
    StepData& d = *step_[who[w]];
    d.conv.clear(); d.act_frames.clear(); d.act_pair_frame.clear();
    std::vector<int>& ix = s.votes;                                // scratch: frame slot -> index in this sequence's frame table
    ix.assign(s.frames.size(), -1);
    // which seeds to look at: the ones noted when their brief was applied (ascending, each once), or all of them
    std::vector<int>& chk = s.check_seeds;
    if (s.check_all) { chk.resize(s.seeds.size()); std::iota(chk.begin(), chk.end(), 0); s.check_all = false; }
    else { std::sort(chk.begin(), chk.end()); chk.erase(std::unique(chk.begin(), chk.end()), chk.end()); }
    for (const int ci : chk) {
      const size_t i = (size_t)ci;
      if (i >= s.seeds.size()) continue;
      Seed& sd = s.seeds[i];
      if (!sd.alive) continue;
      if (Seq::seed_converged(sd)) {
        d.conv.push_back((int)i);
        for (const std::vector<Id>* lst : {&sd.seen_before, &sd.seen})
          for (Id fr : *lst) {
            if (ix[(size_t)fr] < 0) { ix[(size_t)fr] = (int)d.act_frames.size(); d.act_frames.push_back(fr); }
            d.act_pair_frame.push_back(ix[(size_t)fr]);
          }
      }
      else if (!sd.valid) kill_seed(s, d, (int)i, false);         // "z_min is NaN" (:494-498)
    }
    chk.clear();
    n_tg[w] = d.act_pair_frame.size();
  });
  // where each sequence's records go in the call's tables (page-locked, kept between steps: the sequences write their parts in
  // parallel and the tables leave as they are)
  std::vector<size_t> at(who.size() + 1, 0), tg_at(who.size() + 1, 0), fr_at(who.size() + 1, 0);
  for (size_t w = 0; w < who.size(); w++) {
    const StepData& d = *step_[who[w]];
    at[w + 1] = at[w] + d.conv.size(); tg_at[w + 1] = tg_at[w] + n_tg[w]; fr_at[w + 1] = fr_at[w] + d.act_frames.size();
  }
  const size_t n_seeds = at.back(), n_pairs = tg_at.back(), n_frames = fr_at.back();
  // the converged seeds are rows of the resident table, which holds what the host mirrors (the observations run from it): the call
  // names them by slot.  Records are assembled only for a recorded run (the trace keeps them) or for a seed that never got a slot.
  bool by_slot = n_frames <= 65536, tracing = false;
  for (size_t w = 0; w < who.size() && by_slot; w++) {
    const Seq& s = *seq_[who[w]];
    if (s.trace.on()) tracing = true;
    for (int i : step_[who[w]]->conv) if (s.seeds[(size_t)i].slot < 0) { by_slot = false; break; }
  }
  const bool want_records = !by_slot || tracing;
  hso_seed* const seeds = act_seeds_.need(ctx_, want_records ? std::max(n_seeds, (size_t)1) : (size_t)1);
  int32_t* const slots = act_slots_.need(ctx_, std::max(n_seeds, (size_t)1));
  hso_activate_target* const targets = act_targets_.need(ctx_, std::max(n_frames, (size_t)1));
  int32_t* const ints = act_ints_.need(ctx_, 2 * n_seeds + 2 + n_pairs);     // [target_begin (n + 1) | n_mean (n) | frame index per pair]
  hso_activate_out* const out = act_out_.need(ctx_, std::max(n_seeds, (size_t)1));
  int32_t* const begin = ints; int32_t* const n_mean = ints + n_seeds + 1; int32_t* const pair_frame = ints + 2 * n_seeds + 1;
  begin[0] = 0;
  pool_->run((int)who.size(), [&](int w) {
    Seq& s = *seq_[who[w]];
    StepData& d = *step_[who[w]];
    for (size_t q = 0; q < d.act_frames.size(); q++) {
      const Frame& F = s.frames[d.act_frames[q]];
      hso_activate_target& a = targets[fr_at[w] + q];
      a = hso_activate_target{};
      a.frame_id = F.dev_id; a.T_f_w = F.T.v; a.exposure = F.exposure;
    }
    size_t t = tg_at[w], local = 0;
    for (size_t c = 0; c < d.conv.size(); c++) {
      const Seed& sd = s.seeds[d.conv[c]];
      if (want_records) seeds[at[w] + c] = seed_record(s, sd);
      slots[at[w] + c] = sd.slot;
      const size_t np = sd.seen_before.size() + sd.seen.size();
      for (size_t q = 0; q < np; q++) pair_frame[t++] = (int32_t)fr_at[w] + d.act_pair_frame[local++];
      begin[at[w] + c + 1] = (int32_t)t;
      n_mean[at[w] + c] = (int32_t)s.n_mean_converge;
    }
  });
  if (n_seeds > 0) {
    if (n_frames == 0) targets[0] = hso_activate_target{};
    if (by_slot) check(hso_gpu_seed_table_activate(ctx_, &cam_.pod(), seed_table_, slots, (int)n_seeds, begin, pair_frame, targets, (int)n_frames, n_mean, out), "DepthFilter::activatePoint");
    else check(hso_gpu_seed_activate_frames(ctx_, &cam_.pod(), seeds, (int)n_seeds, begin, pair_frame, targets, (int)n_frames, n_mean, out), "DepthFilter::activatePoint");
    n_calls_[7]++; n_items_[7] += (int64_t)who.size();
  }
  pool_->run((int)who.size(), [&](int w) {
    const int k = who[w];
    Seq& s = *seq_[k];
    StepData& d = *step_[k];
    const hso_activate_out* const act_out = out + at[w];
    if (s.trace.on() && !d.conv.empty()) {
      Trace& t = s.trace;
      hso_activate_target none{};
      std::vector<int32_t> rel(d.conv.size() + 1);
      for (size_t c = 0; c <= d.conv.size(); c++) rel[c] = begin[at[w] + c] - begin[at[w]];
      t.begin("seed_activate", 6);
      t.field("cam", &cam_.pod(), sizeof(hso_camera)); t.field("seeds", seeds + at[w], sizeof(hso_seed) * d.conv.size());
      t.field("target_begin", rel.data(), sizeof(int32_t) * rel.size());
      std::vector<hso_activate_target> per_pair(n_tg[w]);           // the trace keeps the per-pair form (what the replay feeds the restatement)
      for (size_t q = 0; q < n_tg[w]; q++) per_pair[q] = targets[pair_frame[tg_at[w] + q]];
      t.field("targets", n_tg[w] == 0 ? &none : per_pair.data(), sizeof(hso_activate_target) * n_tg[w]);
      t.scalar("n_mean_converge_frame", (double)s.n_mean_converge); t.field("out", act_out, sizeof(hso_activate_out) * d.conv.size());
    }
    for (size_t c = 0; c < d.conv.size(); c++) {
      const int i = d.conv[c];
      Seed& sd = s.seeds[i];
      const hso_activate_out& o = act_out[c];
      bool valid = o.is_valid != 0;                               // -1: activatePoint left the flag alone
      if (o.activated) sd.mu = (float)o.opt_id;                   // :418-419
      const Feat& ft = s.feats[sd.feat];
      const Vector3d in_host = along(ft.f, 1.0 / sd.mu);
      if (sd.mu < 1e-10 || in_host[2] < 1e-10) valid = false;     // :423-424
      if (!valid) {
        if (sd.temp != kNone && sd.reprojected) s.points[sd.temp].seed_state = -1;
        kill_seed(s, d, i, false);
        continue;
      }
      if (s.converge_hist.size() > (size_t)cfg_.max_fts) s.converge_hist.erase(s.converge_hist.begin());
      s.converge_hist.push_back(sd.n_dist);
      // the seed becomes a candidate point hosted by its feature (:447-463, MapPointCandidates::newCandidatePoint)
      const Vector3d world = s.frames[ft.frame].T.inverse() * in_host;
      const Id p = s.new_point(world, sd.feat, sd.mu, kPtCandidate);
      Feat& host = s.feats[sd.feat];
      host.point = p;
      if (!host.linked) { host.linked = true; host.next = kNone; }   // else: still the oldest observation of the seed's temporary point
      s.points[p].head = sd.feat; s.points[p].n_obs = 1;
      s.touch_point(p); s.touch_obs(sd.feat);
      if (sd.temp != kNone && sd.reprojected) s.points[sd.temp].seed_state = 1;
      s.candidates.push_back(p);
      kill_seed(s, d, i, true);
    }
    // nMeanConvergeFrame_ (:503-507)
    if (s.converge_hist.size() > size_t(0.5 * cfg_.max_fts))
      s.n_mean_converge = (size_t)(std::accumulate(s.converge_hist.begin(), s.converge_hist.end(), 0) / (int)s.converge_hist.size());
    else s.n_mean_converge = 6;
  });
  erase_slots(who);
}

// The idle-time pass of the depth thread (src/depth_filter.cpp:254-263): while no frame is queued, every seed observes the first of
// the frames that preceded its keyframe (observeDepthWithPreviousFrameOnce, :677-726).  The reference runs as much of a sweep as
// fits before the next frame arrives; the device is always idle between frames, so here it is exactly ONE sweep per frame, over
// all sequences in one call: per keyframe with a list left, its seeds observe the list's first frame, which is then dropped.
// Like the reference's depth thread the sweep runs BESIDE the tracker: previous_begin queues it on the depth filter's stream at the
// end of a step, the next step's upload / tracking / reprojection / pose overlap it, and previous_collect — called before the first
// thing that reads or changes seeds — applies its results.  The results do not depend on when they are collected.
void Bank::previous_begin(const std::vector<int>& who)
{
  if (!cfg_.previous_frame_pass || pending_prev_.on) return;
  std::vector<int64_t> hosts; std::vector<hso_seed_frame> pre;
  bool tracing = false;
  par(who, [&](int k) {
    Seq& s = *seq_[k];
    StepData& d = *step_[k];
    // a list whose keyframe has no live seed left goes with them
    for (size_t i = 0; i < s.pre_lists.size();) {
      Seq::PreList& L = s.pre_lists[i];
      bool any = false;
      const auto rg = s.seed_range(L.batch);
      for (size_t q = rg.first; q < rg.second; q++) if (s.seeds[q].alive) { any = true; break; }
      if (any && !L.frames.empty()) { ++i; continue; }
      for (Id fr : L.frames) release_frame_deferred(s, d, fr);
      s.pre_lists.erase(s.pre_lists.begin() + (std::ptrdiff_t)i);
    }
  });
  erase_slots(who);                                                // moves the frames dropped above to the release list
  // ... which waits until the pass has been collected: hso_gpu_frame_release waits for a pass in flight, and releasing these frames
  // at the start of the next step would end the overlap the second stream exists for
  after_prev_release_.insert(after_prev_release_.end(), to_release_.begin(), to_release_.end());
  to_release_.clear();
  pending_prev_.who.clear(); pending_prev_.n_lists.clear();
  for (int k : who) {
    const Seq& s = *seq_[k];
    if (s.pre_lists.empty()) continue;
    for (const Seq::PreList& L : s.pre_lists) {
      const Frame& F = s.frames[L.frames.front()];
      hso_seed_frame f{};
      f.frame_id = F.dev_id; f.T_f_w = F.T.v; f.exposure_time = F.exposure;
      hosts.push_back(s.frames[L.host].dev_id); pre.push_back(f);
    }
    pending_prev_.who.push_back(k); pending_prev_.n_lists.push_back(s.pre_lists.size());
    tracing |= s.trace.on();
  }
  if (hosts.empty()) { to_release_.insert(to_release_.end(), after_prev_release_.begin(), after_prev_release_.end()); after_prev_release_.clear(); return; }
  int n_slots = 0, n_live = 0;
  check(hso_gpu_seed_table_size(ctx_, seed_table_, &n_slots, &n_live), "DepthFilter");
  pending_prev_.n_slots = n_slots;
  pending_prev_.before.clear(); pending_prev_.full.clear();
  if (tracing || sync_previous_) {
    // recorded runs (and HSO_ENGINE_SYNC_PREVIOUS=1) take the synchronous call: the trace wants the records before and the full results
    seed_brief_.need(ctx_, (size_t)std::max(n_slots, 1));
    if (tracing) {
      pending_prev_.before.resize((size_t)n_slots); pending_prev_.full.resize((size_t)n_slots);
      check(hso_gpu_seed_table_read(ctx_, seed_table_, 0, n_slots, pending_prev_.before.data()), "DepthFilter");
    }
    check(hso_gpu_seed_table_observe_previous(ctx_, &cam_.pod(), seed_table_, hosts.data(), pre.data(), (int)hosts.size(), px_error_angle_,
                                              seed_brief_.data(), tracing ? pending_prev_.full.data() : nullptr), "DepthFilter::observeDepthWithPreviousFrameOnce");
    pending_prev_.async = false;
  } else {
    check(hso_gpu_seed_table_observe_previous_begin(ctx_, &cam_.pod(), seed_table_, hosts.data(), pre.data(), (int)hosts.size(), px_error_angle_),
          "DepthFilter::observeDepthWithPreviousFrameOnce");
    pending_prev_.async = true;
  }
  n_calls_[9]++; n_items_[9] += (int64_t)pending_prev_.who.size();
  pending_prev_.on = true;
  if (!pending_prev_.async) previous_collect();                    // nothing to overlap: apply at once
}

void Bank::previous_collect()
{
  if (!pending_prev_.on) return;
  pending_prev_.on = false;
  to_release_.insert(to_release_.end(), after_prev_release_.begin(), after_prev_release_.end());
  after_prev_release_.clear();
  const int n_slots = pending_prev_.n_slots;
  if (pending_prev_.async) {
    seed_brief_.need(ctx_, (size_t)std::max(n_slots, 1));
    check(hso_gpu_seed_table_observe_previous_end(ctx_, seed_table_, seed_brief_.data(), std::max(n_slots, 1)), "DepthFilter::observeDepthWithPreviousFrameOnce");
  }
  const std::vector<int>& who = pending_prev_.who;
  pool_->run((int)who.size(), [&](int w) {
    const int k = who[(size_t)w];
    Seq& s = *seq_[k];
    StepData& d = *step_[k];
    const size_t n_lists = std::min(pending_prev_.n_lists[(size_t)w], s.pre_lists.size());
    for (size_t li = 0; li < n_lists; li++) {
      Seq::PreList& L = s.pre_lists[li];
      const Id fr = L.frames.front();
      if (s.trace.on() && !pending_prev_.full.empty()) {
        const Frame& F = s.frames[fr];
        std::vector<hso_seed> in; std::vector<hso_seed_out> out;
        for (const Seed& sd : s.seeds)
          if (sd.alive && sd.batch == L.batch && sd.slot >= 0 && sd.slot < n_slots) { in.push_back(pending_prev_.before[sd.slot]); out.push_back(pending_prev_.full[sd.slot]); }
        Trace& t = s.trace;
        t.begin("seed_observe_previous", 7);
        t.field("cam", &cam_.pod(), sizeof(hso_camera)); t.scalar("pre_frame_id", (double)F.dev_id); t.field("T_f_w", &F.T.v, sizeof(hso_se3));
        t.scalar("exposure", F.exposure); t.scalar("px_error_angle", px_error_angle_);
        t.field("seeds", in.data(), sizeof(hso_seed) * in.size()); t.field("out", out.data(), sizeof(hso_seed_out) * out.size());
      }
      const auto rg = s.seed_range(L.batch);
      for (size_t q = rg.first; q < rg.second; q++) {
        Seed& sd = s.seeds[q];
        if (!sd.alive || sd.slot < 0 || sd.slot >= n_slots) continue;
        const hso_seed_brief& o = seed_brief_.data()[sd.slot];
        if (!o.is_update) continue;
        if (sd.seen_before.size() < 15) { sd.seen_before.push_back(fr); s.hold(fr); }   // optFrames_P (:702-703)
        if (o.result == 1) {                                                             // updateSeed (:721)
          sd.mu = o.mu; sd.sigma2 = o.sigma2;
          if (Seq::seed_converged(sd)) s.check_seeds.push_back((int)q);
        }
      }
      release_frame_deferred(s, d, fr);                              // pre_frames.erase(begin()) on every path (:693-724)
      L.frames.erase(L.frames.begin());
    }
  });
  erase_slots(who);
}

// FeatureExtractor::detect (src/feature_detection.cpp:408-497) for one new keyframe per sequence: candidates on the device (batched
// per detection threshold), the oct-tree distribution on the host; sel[i] receives the selected keys of who[i]
void Bank::detect(const std::vector<int>& who, const std::vector<Id>& frame, const std::vector<int>& thresh, bool init, int n_levels, int n_features,
                  std::vector<std::vector<hso_keypoint>>& keys, std::vector<std::vector<hso_keypoint>>& sel)
{
  const int W = cam_.width(), H = cam_.height();
  const int second_cap = ((W + 7) / 8) * ((H + 7) / 8);
  sel.assign(who.size(), {});
  std::vector<int> todo(who.size());
  std::iota(todo.begin(), todo.end(), 0);
  while (!todo.empty()) {
    // the keyframes of a step in ONE call, each at its own barrier (hso_gpu_detect_candidates_multi); the initialisation branch —
    // once per sequence — is still grouped by barrier
    const int th = thresh[todo[0]];
    std::vector<int> grp, rest;
    for (int i : todo) ((!init || thresh[i] == th) ? grp : rest).push_back(i);
    todo.swap(rest);
    std::vector<int32_t> ths;
    for (int i : grp) ths.push_back(thresh[i]);
    std::vector<int64_t> ids;
    for (int i : grp) ids.push_back(seq_[who[i]]->frames[frame[i]].dev_id);
    const int n = (int)grp.size();
    // the result tables are the bank's own page-locked arrays, kept between keyframes (fresh pageable vectors of this size cost
    // more in first-touch page faults and staging copies than the detection itself)
    int corner_cap = det_corner_cap_;
    std::vector<int32_t> nc((size_t)n * n_levels), ns((size_t)n * (init ? 1 : n_levels));
    hso_corner* fill = init ? det_fill_.need(ctx_, (size_t)n * second_cap) : nullptr;
    hso_edgelet* ed = init ? nullptr : det_edgelets_.need(ctx_, (size_t)n * n_levels * second_cap);
    hso_corner* co = nullptr;
    for (;;) {
      co = det_corners_.need(ctx_, (size_t)n * n_levels * corner_cap);
      const int rc = init ? hso_gpu_detect_candidates_init(ctx_, ids.data(), n, n_levels, th, co, corner_cap, nc.data(), fill, second_cap, ns.data())
                          : hso_gpu_detect_candidates_multi(ctx_, ids.data(), n, n_levels, ths.data(), co, corner_cap, nc.data(), ed, second_cap, ns.data());
      check(rc, "FeatureExtractor");
      n_calls_[9]++; n_items_[9] += n;
      const int most = *std::max_element(nc.begin(), nc.end());
      if (most <= corner_cap) break;
      corner_cap = det_corner_cap_ = most + most / 4;             // more corners than the lists hold: once more with room for all
    }
    pool_->run(n, [&](int g) {
      const int i = grp[g];
      Seq& s = *seq_[who[i]];
      const hso_corner* cg = co + (size_t)g * n_levels * corner_cap;
      const int32_t* ncg = nc.data() + (size_t)g * n_levels;
      if (s.trace.on()) {
        Trace& t = s.trace;
        t.begin("detect_candidates", 5 + 2 * (uint32_t)n_levels + 1);
        t.scalar("init", init ? 1 : 0); t.scalar("frame_id", (double)ids[g]); t.scalar("n_levels", n_levels); t.scalar("min_thresh", ths[(size_t)g]);
        t.field("corner_counts", ncg, sizeof(int32_t) * (size_t)n_levels);
        for (int L = 0; L < n_levels; L++) t.field(("corners" + std::to_string(L)).c_str(), cg + (size_t)L * corner_cap, sizeof(hso_corner) * (size_t)ncg[L]);
        if (init) {
          t.field("fill", fill + (size_t)g * second_cap, sizeof(hso_corner) * (size_t)ns[g]);
          for (int L = 1; L < n_levels; L++) t.field("unused", nullptr, 0);
          t.field("second_counts", &ns[g], sizeof(int32_t));
        } else {
          for (int L = 0; L < n_levels; L++)
            t.field(("edgelets" + std::to_string(L)).c_str(), ed + ((size_t)g * n_levels + L) * second_cap, sizeof(hso_edgelet) * (size_t)ns[(size_t)g * n_levels + L]);
          t.field("second_counts", &ns[(size_t)g * n_levels], sizeof(int32_t) * (size_t)n_levels);
        }
      }
      // allFeturesToDistribute_: the keys already there (occupancy), then per level the corners followed by the second kind
      std::vector<hso_keypoint>& all = keys[i];
      for (int L = 0; L < n_levels; L++) {
        for (int j = 0; j < ncg[L]; j++) {
          const hso_corner& c = cg[(size_t)L * corner_cap + j];
          hso_keypoint kp{};
          kp.x = (float)(c.x << L); kp.y = (float)(c.y << L); kp.response = c.response; kp.level = L; kp.species = HSO_KP_CORNER_HIGH;
          all.push_back(kp);
        }
        if (init) {
          for (int j = 0; L == 0 && j < ns[g]; j++) {
            const hso_corner& c = fill[(size_t)g * second_cap + j];
            hso_keypoint kp{};
            kp.x = (float)c.x; kp.y = (float)c.y; kp.response = c.response; kp.level = 0; kp.species = HSO_KP_GRAD;
            all.push_back(kp);
          }
        } else {
          for (int j = 0; j < ns[(size_t)g * n_levels + L]; j++) {
            const hso_edgelet& e = ed[((size_t)g * n_levels + L) * second_cap + j];
            hso_keypoint kp{};
            kp.x = (float)(e.x << L); kp.y = (float)(e.y << L); kp.response = e.grad; kp.level = L; kp.species = HSO_KP_EDGELET; kp.gx = e.gx; kp.gy = e.gy;
            all.push_back(kp);
          }
        }
      }
      std::vector<hso_keypoint>& out = sel[i];
      out.resize(all.size() + 1);
      const int m = hso_gpu_select_octree(all.data(), (int)all.size(), 0, W, 0, H, n_features, out.data(), (int)out.size());
      if (m < 0) throw std::runtime_error("FeatureExtractor: oct-tree selection failed");
      out.resize((size_t)m);
      if (s.trace.on()) {
        Trace& t = s.trace;
        t.begin("select_octree", 5);
        t.field("keys", all.data(), sizeof(hso_keypoint) * all.size()); t.scalar("width", W); t.scalar("height", H); t.scalar("n_features", n_features);
        t.field("out", out.data(), sizeof(hso_keypoint) * out.size());
      }
    });
  }
}

// a selected key as a feature of `fr` (src/feature_detection.cpp:457-484)
Feat Bank::feature_from_key(const hso_keypoint& kp, Id fr) const
{
  Feat ft;
  ft.frame = fr;
  ft.px[0] = kp.x; ft.px[1] = kp.y;
  const Vector3d b = cam_.cam2world({ft.px[0], ft.px[1]});
  ft.f[0] = b[0]; ft.f[1] = b[1]; ft.f[2] = b[2];
  ft.level = (int8_t)kp.level;
  if (kp.species == HSO_KP_CORNER_HIGH) ft.type = HSO_FTR_CORNER;
  else {
    ft.type = kp.species == HSO_KP_GRAD ? HSO_FTR_GRADIENT : HSO_FTR_EDGELET;
    const double gx = kp.gx, gy = kp.gy, nrm = std::sqrt(gx * gx + gy * gy);   // fillingHole never sets gx / gy: the default direction stays
    if (nrm > 0) { ft.grad[0] = gx / nrm; ft.grad[1] = gy / nrm; }
  }
  return ft;
}

// DepthFilter::addKeyframe -> initializeSeeds (src/depth_filter.cpp:146-205): new features away from the keyframe's own and from
// the seeds just matched in it, one seed each
void Bank::start_seeds(const std::vector<int>& who)
{
  std::vector<std::vector<hso_keypoint>> keys(who.size()), sel;
  std::vector<Id> frame(who.size()); std::vector<int> thresh(who.size());
  for (size_t i = 0; i < who.size(); i++) {
    Seq& s = *seq_[who[i]];
    StepData& d = *step_[who[i]];
    const Frame& C = s.frames[s.cur];
    s.kf_depth_mean = d.dist_mean; s.kf_depth_min = 0.5 * d.depth_min;        // src/frame_handler_mono.cpp:335-338
    s.converge_thresh = d.n_inliers <= 70 ? 100.f : 200.f;
    keys[i] = d.occupied;
    for (Id f : C.fts) {                                          // FeatureExtractor::setExistingFeatures
      hso_keypoint kp{};
      kp.x = (float)s.feats[f].px[0]; kp.y = (float)s.feats[f].px[1]; kp.species = HSO_KP_OCCUR;
      keys[i].push_back(kp);
    }
    frame[i] = s.cur; thresh[i] = (int)C.grad_mean;
  }
  detect(who, frame, thresh, false, cfg_.n_pyr_levels, cfg_.max_fts + 100, keys, sel);
  pool_->run((int)who.size(), [&](int i) {
    Seq& s = *seq_[who[i]];
    StepData& d = *step_[who[i]];
    ++s.batch;
    const std::vector<Id>* before = nullptr;
    for (const auto& p : s.prior) if (p.first == s.batch - 1) before = &p.second;
    d.new_seeds.clear();
    for (const hso_keypoint& kp : sel[i]) {
      s.feats.push_back(feature_from_key(kp, s.cur));
      Seed sd;
      sd.feat = (Id)s.feats.size() - 1;
      sd.batch = s.batch;
      sd.mu = (float)(1.0 / (float)s.kf_depth_mean); sd.z_range = (float)(1.0 / (float)s.kf_depth_min);   // Seed::Seed, :49-68
      sd.sigma2 = sd.z_range * sd.z_range / 36;
      sd.converge = s.converge_thresh;
      s.seeds.push_back(sd);
      d.new_seeds.push_back(seed_record(s, s.seeds.back()));
    }
    if (before && !before->empty() && !sel[i].empty() && cfg_.previous_frame_pass) {   // Seed::pre_frames (:186-192)
      s.pre_lists.push_back(Seq::PreList{s.cur, s.batch, *before});
      for (Id fr : *before) s.hold(fr);
    }
  });
  std::vector<hso_seed> rows; std::vector<int32_t> group;
  for (int k : who) { const StepData& d = *step_[k]; rows.insert(rows.end(), d.new_seeds.begin(), d.new_seeds.end()); group.insert(group.end(), d.new_seeds.size(), k); }
  if (rows.empty()) return;
  int32_t first = 0;
  check(hso_gpu_seed_table_append(ctx_, seed_table_, rows.data(), group.data(), (int)rows.size(), &first), "DepthFilter");
  for (int k : who) {
    Seq& s = *seq_[k];
    const size_t n_new = step_[k]->new_seeds.size();
    for (size_t j = 0; j < n_new; j++) s.seeds[s.seeds.size() - n_new + j].slot = first++;
  }
}

// The seed branch of Reprojector::reprojectMap (src/reprojector.cpp:309-329, reprojectorSeeds :431-502): with fewer than 100
// matches the nearly converged seeds are matched as well, per cell the one with the smallest variance that matches becomes a
// temporary point observed in this frame.  The frame's feature list changes, so its pose is optimised again, value-passing.
void Bank::seed_branch(const std::vector<int>& who)
{
  previous_collect();
  std::vector<int> again;
  for (int k : who) {
    Seq& s = *seq_[k];
    StepData& d = *step_[k];
    Frame& C = s.frames[s.cur];
    std::vector<int> pick; std::vector<hso_seed> in;
    for (size_t i = 0; i < s.seeds.size(); i++) {
      const Seed& sd = s.seeds[i];
      if (!sd.alive || sd.reprojected || !(std::sqrt(sd.sigma2) < sd.z_range / cfg_.reproject_seed_thresh)) continue;
      pick.push_back((int)i); in.push_back(seed_record(s, sd));
    }
    if (pick.empty()) continue;
    C.n_fts = (int32_t)C.loose.size();
    std::vector<hso_reproj_point> proj(pick.size()); std::vector<hso_align_out> match(pick.size());
    check(hso_gpu_seed_reproject_match(ctx_, &cam_.pod(), C.dev_id, &C.T.v, C.exposure, in.data(), (int)in.size(), cell_size_, grid_cols_, proj.data(),
                                       match.data()), "Reprojector (seeds)");
    if (s.trace.on()) {
      Trace& t = s.trace;
      t.begin("seed_reproject_match", 9);
      t.field("cam", &cam_.pod(), sizeof(hso_camera)); t.scalar("cur_frame_id", (double)C.dev_id); t.field("T_f_w", &C.T.v, sizeof(hso_se3));
      t.scalar("exposure", C.exposure); t.field("seeds", in.data(), sizeof(hso_seed) * in.size());
      t.scalar("cell_size", cell_size_); t.scalar("grid_n_cols", grid_cols_);
      t.field("proj", proj.data(), sizeof(hso_reproj_point) * proj.size()); t.field("match", match.data(), sizeof(hso_align_out) * match.size());
    }
    std::vector<std::vector<int>> cell(cell_order_.size());
    for (size_t j = 0; j < pick.size(); j++) if (proj[j].projected) cell[(size_t)proj[j].cell].push_back((int)j);
    int n_matches = s.log.n_matches;
    bool added = false;
    for (size_t ci = 0; ci < cell.size(); ci++) {
      std::vector<int>& in_cell = cell[(size_t)cell_order_[ci]];
      std::stable_sort(in_cell.begin(), in_cell.end(), [&](int a, int b) { return s.seeds[pick[a]].sigma2 < s.seeds[pick[b]].sigma2; });
      bool got = false;
      for (int j : in_cell) {
        if (!match[j].success) continue;
        Seed& sd = s.seeds[pick[j]];
        const Feat host = s.feats[sd.feat];
        const Vector3d world = s.frames[host.frame].T.inverse() * along(host.f, 1.0 / sd.mu);
        const Id p = s.new_point(world, sd.feat, sd.mu, kPtTemporary);
        s.observe(p, sd.feat);                                    // Point(pos, ftr): the host feature is its first observation
        Feat nf;
        nf.frame = s.cur; nf.point = p;
        nf.px[0] = match[j].px_cur[0]; nf.px[1] = match[j].px_cur[1];
        const Vector3d b = cam_.cam2world({nf.px[0], nf.px[1]});
        nf.f[0] = b[0]; nf.f[1] = b[1]; nf.f[2] = b[2];
        nf.level = (int8_t)match[j].search_level;
        if (host.type == HSO_FTR_EDGELET) {
          nf.type = HSO_FTR_EDGELET;
          const double gx = match[j].A_cur_ref[0] * host.grad[0] + match[j].A_cur_ref[1] * host.grad[1];
          const double gy = match[j].A_cur_ref[2] * host.grad[0] + match[j].A_cur_ref[3] * host.grad[1];
          const double nn = std::sqrt(gx * gx + gy * gy);
          nf.grad[0] = gx / nn; nf.grad[1] = gy / nn;
        } else nf.type = host.type == HSO_FTR_GRADIENT ? HSO_FTR_GRADIENT : HSO_FTR_CORNER;
        C.loose.push_back(nf);
        sd.reprojected = true; sd.temp = p;
        s.points[p].seed_state = 0;
        s.temps.push_back(p);                                     // MapPointCandidates::addPauseSeedPoint
        s.log.n_seed_matches++;
        got = added = true;
        break;
      }
      if (got) ++n_matches;
      if (n_matches >= cfg_.max_fts) break;
    }
    s.log.n_matches = n_matches;
    (void)added; (void)d;
  }
  // every sequence of the branch gets its pose optimised here, over the complete feature list as the host holds it (the chain left
  // the optimiser's culling unapplied for them)
  for (int k : who) {
    Seq& s = *seq_[k];
    StepData& d = *step_[k];
    const Frame& C = s.frames[s.cur];
    bool any = false;
    for (const Feat& ft : C.loose) any |= ft.point != kNone;
    if (any) { again.push_back(k); continue; }
    // nothing to optimise over (no feature with a point: a textureless frame): the optimiser's early return (src/pose_optimizer.cpp:456)
    d.pose = hso_pose_result{};
    d.pose.status = 1; d.pose.T_f_w = C.T.v;
    d.pose_mask.assign(std::max(C.loose.size(), (size_t)1), 0);
    d.host_pose = true;
  }
  if (again.empty()) return;
  // pose_optimizer::optimizeLevenbergMarquardt3rd over the complete feature lists
  std::vector<std::vector<hso_pose_feat>> feats(again.size());
  std::vector<std::vector<hso_se3>> poses(again.size());
  std::vector<hso_pose_job> jobs(again.size());
  std::vector<hso_pose_result> res(again.size());
  std::vector<std::vector<uint8_t>> mask(again.size());
  std::vector<uint8_t*> mask_ptr(again.size());
  for (size_t i = 0; i < again.size(); i++) {
    Seq& s = *seq_[again[i]];
    const Frame& C = s.frames[s.cur];
    std::vector<Id> host_frames;
    for (const Feat& ft : C.loose) {
      hso_pose_feat pf{};
      pf.has_point = ft.point != kNone; pf.type = ft.type; pf.level = ft.level;
      pf.f[0] = ft.f[0]; pf.f[1] = ft.f[1]; pf.f[2] = ft.f[2]; pf.grad[0] = ft.grad[0]; pf.grad[1] = ft.grad[1];
      if (ft.point != kNone) {
        const Point& P = s.points[ft.point];
        const Feat& host = s.feats[P.host];
        pf.temporary = P.kind == kPtTemporary;
        pf.host_f[0] = host.f[0]; pf.host_f[1] = host.f[1]; pf.host_f[2] = host.f[2];
        pf.idist = P.idist;
        size_t h = 0;
        while (h < host_frames.size() && host_frames[h] != host.frame) h++;
        if (h == host_frames.size()) { host_frames.push_back(host.frame); poses[i].push_back(s.frames[host.frame].T.v); }
        pf.host_pose = (int)h;
      }
      feats[i].push_back(pf);
    }
    hso_pose_job& j = jobs[i];
    j = hso_pose_job{};
    j.feats = feats[i].data(); j.n_feats = (int)feats[i].size(); j.poses_f_w = poses[i].data(); j.n_poses = (int)poses[i].size();
    j.T_f_w = C.T.v; j.reproj_thresh = cfg_.poseoptim_thresh; j.n_iter = 12;
    mask[i].assign(std::max(feats[i].size(), (size_t)1), 0);
    mask_ptr[i] = mask[i].data();
  }
  check(hso_gpu_pose_optimize_batch(ctx_, &cam_.pod(), jobs.data(), (int)jobs.size(), res.data(), mask_ptr.data()), "pose_optimizer");
  n_calls_[5]++; n_items_[5] += (int64_t)jobs.size();
  for (size_t i = 0; i < again.size(); i++) {
    Seq& s = *seq_[again[i]];
    StepData& d = *step_[again[i]];
    d.pose = res[i]; d.pose_mask = mask[i]; d.host_pose = true;
    s.frames[s.cur].n_fts = (int32_t)s.frames[s.cur].loose.size();
    if (s.trace.on()) {
      Trace& t = s.trace;
      t.begin("pose_optimize", 8);
      t.field("cam", &cam_.pod(), sizeof(hso_camera)); t.field("feats", feats[i].data(), sizeof(hso_pose_feat) * feats[i].size());
      t.field("poses", poses[i].data(), sizeof(hso_se3) * poses[i].size()); t.field("T_f_w", &jobs[i].T_f_w, sizeof(hso_se3));
      t.scalar("reproj_thresh", jobs[i].reproj_thresh); t.scalar("n_iter", jobs[i].n_iter);
      t.field("result", &res[i], sizeof(res[i])); t.field("mask", mask[i].data(), feats[i].size());
    }
  }
}

// ------------------------------------------------------------------------------------------------ device mirror
// what changed in the sequence tables since the last flush, as the device's record types: keyframe table and key points, point and
// observation rows (with the Feature::point links), the keyframes' feature lists, the candidate list
void Bank::flush_maps(const std::vector<int>& who)
{
  struct Patch {
    std::vector<hso_kf> kfs; std::vector<int32_t> keys; std::vector<int32_t> pid, oid, olink; std::vector<hso_map_point> pts; std::vector<hso_obs> obs;
    std::vector<std::pair<int32_t, std::pair<int32_t, std::vector<int32_t>>>> lists;   // (list, (first, ids))
    bool kf = false, key = false;
  };
  std::vector<Patch> patch(who.size());
  std::optional<Sub> t_sec;
  t_sec.emplace(this, "flush: build (pool)");
  pool_->run((int)who.size(), [&](int i) {
    Seq& s = *seq_[who[i]];
    Patch& P = patch[i];
    if (s.kfs_dirty) {
      P.kf = true;
      for (Id fr : s.dev_kfs) {
        const Frame& F = s.frames[fr];
        hso_kf r{};
        r.frame_id = F.dev_id; r.T_f_w = F.T.v; r.exposure_time = F.exposure; r.keyframe_id = F.kf_id;
        P.kfs.push_back(r);
      }
    }
    if ((s.kfs_dirty || s.keys_dirty) && !s.dev_kfs.empty()) {
      // Frame::key_pts_ as point rows (what Map::getCloseKeyframes looks at)
      P.key = true;
      for (Id fr : s.dev_kfs) for (Id kf_feat : s.frames[fr].key) P.keys.push_back(kf_feat == kNone ? -1 : s.feats[kf_feat].point);
      s.keys_dirty = false;
    }
    for (Id p : s.dirty_pts) {
      Point& pt = s.points[p];
      s.pt_flag[p] = 0;
      if (pt.host == kNone) continue;
      const Feat& host = s.feats[pt.host];
      hso_map_point r{};
      r.pos[0] = pt.pos[0]; r.pos[1] = pt.pos[1]; r.pos[2] = pt.pos[2];
      r.idist = pt.idist;
      r.host_f[0] = host.f[0]; r.host_f[1] = host.f[1]; r.host_f[2] = host.f[2];
      r.host_kf = s.frames[host.frame].kf_row;
      r.obs_begin = pt.head; r.obs_count = pt.n_obs;
      if (r.host_kf < 0) continue;                                // hosted in a frame that never became a keyframe: not projectable
      // the state word: kind and face (the selection's quality key) and the bad flag are the host's; the device keeps counting the
      // failures / successes unless this patch resets them
      r.pad_ = (int32_t)((uint32_t)HSO_PT_WORD(quality_key(pt), 0, pt.bad, 0) | ((pt.dev_reset & 1) ? 0u : HSO_PT_KEEP_NFAIL) | ((pt.dev_reset & 2) ? 0u : HSO_PT_KEEP_NOK));
      pt.dev_reset = 0;
      P.pid.push_back(p); P.pts.push_back(r);
    }
    s.dirty_pts.clear();
    for (Id f : s.dirty_obs) {
      const Feat& ft = s.feats[f];
      s.obs_flag[f] = 0;
      hso_obs r{};
      r.kf = s.frames[ft.frame].kf_row; r.level = ft.level; r.type = ft.type; r.pad_ = ft.linked ? ft.next : -1;
      r.px[0] = ft.px[0]; r.px[1] = ft.px[1]; r.f[0] = ft.f[0]; r.f[1] = ft.f[1]; r.f[2] = ft.f[2]; r.grad[0] = ft.grad[0]; r.grad[1] = ft.grad[1];
      if (r.kf < 0) continue;
      P.oid.push_back(f); P.obs.push_back(r); P.olink.push_back(ft.point);
    }
    s.dirty_obs.clear();
    // Frame::fts_ of the keyframes whose list grew (lists only grow: a new keyframe's own features, later the features of its
    // seeds whose points were promoted)
    for (Id fr : s.dirty_lists) {
      Frame& F = s.frames[fr];
      if (!F.in_use || F.kf_row < 0 || (size_t)F.fts_sent >= F.fts.size()) continue;
      P.lists.emplace_back(F.kf_row, std::make_pair(F.fts_sent, std::vector<int32_t>(F.fts.begin() + F.fts_sent, F.fts.end())));
      F.fts_sent = (int32_t)F.fts.size();
    }
    s.dirty_lists.clear();
    // MapPointCandidates::candidates_: the device's copy is the host's list plus the entries the device deleted itself (it skips
    // those).  While the host's list is the live part of the device's plus a tail, the tail is all that goes; otherwise (a
    // promotion took candidates out) the whole list
    {
      size_t at = 0; bool prefix = true;
      for (Id p : s.dev_cands) {
        if (s.points[p].kind != kPtCandidate) continue;           // deleted on the device (event) — or taken out by the host: then no prefix
        if (at < s.candidates.size() && s.candidates[at] == p) at++; else { prefix = false; break; }
      }
      if (prefix) for (Id p : s.dev_cands) if (s.points[p].kind != kPtCandidate && s.points[p].kind != kPtDeleted) { prefix = false; break; }
      if (prefix && s.dev_cands.size() < 4 * s.candidates.size() + 64) {
        if (at < s.candidates.size()) {
          P.lists.emplace_back((int32_t)HSO_LIST_CANDIDATES, std::make_pair((int32_t)s.dev_cands.size(), std::vector<int32_t>(s.candidates.begin() + (std::ptrdiff_t)at, s.candidates.end())));
          s.dev_cands.insert(s.dev_cands.end(), s.candidates.begin() + (std::ptrdiff_t)at, s.candidates.end());
        }
      } else {
        P.lists.emplace_back((int32_t)HSO_LIST_CANDIDATES, std::make_pair(0, std::vector<int32_t>(s.candidates.begin(), s.candidates.end())));
        s.dev_cands = s.candidates;
      }
    }
  });
  t_sec.emplace(this, "flush: device calls");
  std::vector<hso_seqmap_rows> rows;
  std::vector<hso_seqmap_list_patch> lists;
  for (size_t i = 0; i < who.size(); i++) {
    Seq& s = *seq_[who[i]];
    Patch& P = patch[i];
    if (P.kf) { check(hso_gpu_seqmap_set_keyframes(ctx_, s.map, P.kfs.data(), (int)P.kfs.size()), "Map"); s.kfs_dirty = false; }
    if (!P.pid.empty() || !P.oid.empty()) {
      hso_seqmap_rows r{};
      r.map = s.map; r.n_points = (int)P.pid.size(); r.n_obs = (int)P.oid.size();
      r.point_ids = P.pid.data(); r.points = P.pts.data(); r.obs_ids = P.oid.data(); r.obs = P.obs.data(); r.obs_point = P.olink.data();
      rows.push_back(r);
    }
    for (auto& L : P.lists) {
      hso_seqmap_list_patch lp{};
      lp.map = s.map; lp.list = L.first; lp.first = L.second.first; lp.n = (int32_t)L.second.second.size(); lp.ids = L.second.second.data();
      lists.push_back(lp);
    }
  }
  if (!rows.empty()) check(hso_gpu_seqmap_patch_multi(ctx_, rows.data(), (int)rows.size()), "Map");
  if (!lists.empty()) check(hso_gpu_seqmap_patch_lists(ctx_, lists.data(), (int)lists.size()), "Map");
  for (size_t i = 0; i < who.size(); i++)                          // after the point rows they name exist
    if (patch[i].key) check(hso_gpu_seqmap_set_key_points(ctx_, seq_[who[i]]->map, patch[i].keys.data(), (int)(patch[i].keys.size() / 5)), "Map");
}

// ------------------------------------------------------------------------------------------------ end of the frame
// FrameHandlerMono::addImage's tail (:113-122) and FrameHandlerBase::finishFrameProcessingCommon (src/frame_handler_base.cpp:116-152)
void Bank::finish(const std::vector<int>& who)
{
  // a failed start drops everything and pauses the handler (resetAll): device calls, so on this thread (rare)
  for (int k : who) {
    Seq& s = *seq_[k];
    StepData& d = *step_[k];
    if (d.stage0 == kRunning || d.stage0 == kRelocalising || s.outcome != kFailure) continue;
    for (Frame& F : s.frames) if (F.in_use && F.dev_id >= 0) to_release_.push_back(F.dev_id);
    drop_sequence_seeds(k);
    s.reset_tables();
    s.stage = kPaused; s.quality = kInsufficient; s.n_obs_last = 0;
  }
  // everything else touches one sequence's tables only
  par(who, [&](int k) {
    Seq& s = *seq_[k];
    StepData& d = *step_[k];
    if (d.stage0 == kRunning || d.stage0 == kRelocalising) {
      Frame& C = s.frames[s.cur];
      if (!d.tracked) s.outcome = kFailure;                       // relocalisation found no keyframe / too few tracked features
      if (d.ok) {
        const Frame& L = s.frames[d.ref];
        if (d.make_kf) { s.kfs.push_back(s.cur); s.kfs_dirty = true; }   // Map::addKeyframe
        else s.regular++;
        s.motion = C.T * L.T.inverse();                           // :287, :352
        if (d.relocalised) s.stage = kRunning;
      } else if (d.relocalised) C.T = d.reloc_pose;               // "reset to last well localized pose"
      s.log.n_seeds = (int)s.seeds.size() - s.n_dead_seeds; s.log.n_candidates = (int)s.candidates.size();
      if (s.outcome == kFailure) { s.stage = kRelocalising; s.quality = kInsufficient; }
    } else if (s.stage == kPaused) return;                        // the failed start handled above
    // the new frame becomes the last one
    const Id old = s.last;
    s.last = s.cur; s.cur = kNone;
    if (old != kNone && old != s.last) release_frame_deferred(s, d, old);
    s.n_obs_last = s.frames[s.last].n_inliers;
    s.hist_stamp.push_back(s.frames[s.last].stamp); s.hist_pose.push_back(s.frames[s.last].T.v);
    // dead seeds leave the list once they are the majority (list order of the live ones is kept)
    if (s.n_dead_seeds > 256 && s.n_dead_seeds * 2 > (int)s.seeds.size()) {
      size_t keep = 0;
      for (size_t i = 0; i < s.seeds.size(); i++) if (s.seeds[i].alive) { if (keep != i) s.seeds[keep] = std::move(s.seeds[i]); keep++; }
      s.seeds.resize(keep); s.n_dead_seeds = 0;
      s.check_seeds.clear(); s.check_all = true;                   // the noted indices are stale
    }
  });
  for (int k : who) {
    StepData& d = *step_[k];
    to_release_.insert(to_release_.end(), d.released.begin(), d.released.end());
    d.released.clear();
  }
  // the resident table drops its erased slots once they are the majority
  int n_slots = 0, n_live = 0;
  if (hso_gpu_seed_table_size(ctx_, seed_table_, &n_slots, &n_live) == HSO_OK && n_slots > 4096 && n_live * 2 < n_slots) {
    std::vector<int32_t> remap((size_t)n_slots);
    check(hso_gpu_seed_table_compact(ctx_, seed_table_, remap.data()), "DepthFilter");
    for (Seq* s : seq_) for (Seed& sd : s->seeds) if (sd.alive && sd.slot >= 0) sd.slot = remap[(size_t)sd.slot];
  }
}

}  // namespace engine
}  // namespace hso
