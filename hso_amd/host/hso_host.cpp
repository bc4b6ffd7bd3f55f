// hso_host.cpp — bodies of the host mirror: thin adapters onto the C-ABI.
#include "hso_host.h"
#include "hso_api.h"
#include <algorithm>
#include <cmath>
#include <string>

namespace hso {

api::Trace& api::trace() { static thread_local Trace t; return t; }

// per driver thread: a sequence lives on one thread (the single-sequence driver on its caller's, the multi-sequence driver on one
// worker each), so every sequence counts its own frames, keyframes, points and seed batches like a process of the reference does
thread_local int Frame::frame_counter_ = 0;
thread_local int Frame::id_base_ = 0;
thread_local int Frame::keyFrameCounter_ = 0;
thread_local int Point::point_counter_ = 0;
thread_local int Seed::batch_counter = 0;

bool Point::deleteFrameRef(Frame* frame)
{
  const auto hit = std::find_if(obs_.begin(), obs_.end(), [frame](const Feature* o) { return o->frame == frame; });
  if (hit == obs_.end()) return false;
  obs_.erase(hit);
  return true;
}

Frame::Frame(hso_gpu_ctx* ctx, AbstractCamera* cam, const uint8_t* img, int width, int height, double timestamp)
    : id_(frame_counter_++), timestamp_(timestamp), cam_(cam), ctx_(ctx)
{
  // src/frame.cpp:85-86
  if (!img || width != cam->width() || height != cam->height())
    throw std::runtime_error("Frame: provided image has not the same size as the camera model or image is not grayscale");
  hso_frame_stats st{};
  api::frame_upload(ctx_, id_, img, width, height, &st);
  integralImage_ = st.integral_image;
  gradMean_ = st.grad_mean;
}

Frame::~Frame()
{
  for (Feature* f : fts_) delete f;
  hso_gpu_frame_release(ctx_, id_);
}

CoarseTracker::CoarseTracker(bool inverse_composition, int max_level, int min_level, int n_iter, bool verbose)
    : m_inverse_composition(inverse_composition), m_max_level(max_level), m_min_level(min_level),
      m_n_iter(n_iter), m_verbose(verbose)
{
}

size_t CoarseTracker::run(FramePtr ref, FramePtr cur)
{
  if (ref->fts_.empty()) return 0;  // src/CoarseTracker.cpp:53-54
  // makeDepthRef (:210-240), flattened in fts_ list order; a feature without point keeps its slot
  std::vector<hso_ref_feat> feats;
  feats.reserve(ref->fts_.size());
  for (Feature* ft : ref->fts_) {
    hso_ref_feat r{};
    r.px[0] = ft->px[0]; r.px[1] = ft->px[1];
    r.f[0] = ft->f[0]; r.f[1] = ft->f[1]; r.f[2] = ft->f[2];
    r.dist = -1;
    if (ft->point != nullptr) {
      const Feature* host = ft->point->hostFeature_;
      const double inv = 1.0 / ft->point->idist_;
      const Vector3d p_host{host->f[0] * inv, host->f[1] * inv, host->f[2] * inv};
      const SE3 T_r_h = ref->T_f_w_ * host->frame->T_f_w_.inverse();
      const Vector3d p_ref = T_r_h * p_host;
      if (!(p_ref[2] < 0.00001)) r.dist = std::sqrt(p_ref[0] * p_ref[0] + p_ref[1] * p_ref[1] + p_ref[2] * p_ref[2]);
    }
    feats.push_back(r);
  }
  hso_track_job job{};
  job.ref_frame_id = ref->id_;
  job.cur_frame_id = cur->id_;
  job.feats = feats.data();
  job.n_feats = (int)feats.size();
  job.T_cur_ref = (cur->T_f_w_ * ref->T_f_w_.inverse()).v;           // :63
  job.exposure_rat = cur->integralImage_ / ref->integralImage_;       // :60
  hso_track_params p{m_inverse_composition ? 1 : 0, m_max_level, m_min_level, m_n_iter};
  api::coarse_track(cur->ctx_, &cur->cam_->pod(), &p, &job, &m_last);
  m_T_cur_ref.v = m_last.T_cur_ref;
  // write-back, :198-202
  cur->T_f_w_ = m_T_cur_ref * ref->T_f_w_;
  cur->m_exposure_time = (double)m_last.exposure_rat * ref->m_exposure_time;
  if (m_last.exposure_rat > 0.99 && m_last.exposure_rat < 1.01) cur->m_exposure_time = ref->m_exposure_time;
  return (size_t)m_last.n_tracked;  // :207
}

// ---------------------------------------------------------------- Matcher
static Vector3d sub(const Vector3d& a, const Vector3d& b) { return {a[0] - b[0], a[1] - b[1], a[2] - b[2]}; }
static double norm(const Vector3d& a) { return std::sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]); }

bool Point::getCloseViewObs(const Vector3d& framepos, Feature*& ftr) const
{
  Vector3d obs_dir = sub(framepos, pos_);
  { const double n = norm(obs_dir); for (double& c : obs_dir) c /= n; }
  // the observation whose viewing ray is closest to the new one; the first observation when none has a positive cosine
  Feature* chosen = obs_.front();
  double best = 0;
  for (Feature* o : obs_) {
    Vector3d ray = sub(o->frame->pos(), pos_);
    const double len = norm(ray);
    for (double& x : ray) x /= len;                                 // Eigen's normalize(): component by component
    const double c = obs_dir[0] * ray[0] + obs_dir[1] * ray[1] + obs_dir[2] * ray[2];
    if (c > best) { best = c; chosen = o; }
  }
  ftr = chosen;
  return !(best < 0.5);
}

// flatten (Point, chosen reference feature, current frame) the way findMatchDirect reads them
static hso_align_job make_align_job(const Point& pt, const Feature* ref, const Frame& cur, const Vector2d& px_cur)
{
  hso_align_job j{};
  j.ref_frame_id = ref->frame->id_;
  j.ref_level = ref->level;
  j.type = (int)ref->type;
  j.px_ref[0] = ref->px[0]; j.px_ref[1] = ref->px[1];
  j.f_ref[0] = ref->f[0]; j.f_ref[1] = ref->f[1]; j.f_ref[2] = ref->f[2];
  j.grad[0] = ref->grad[0]; j.grad[1] = ref->grad[1];
  // depth along the reference bearing, src/matcher.cpp:295-306
  j.depth = (ref->frame->id_ == pt.hostFeature_->frame->id_) ? 1.0 / pt.idist_ : norm(sub(ref->frame->pos(), pt.pos_));
  j.T_cur_ref = (cur.T_f_w_ * ref->frame->T_f_w_.inverse()).v;
  j.px_cur[0] = px_cur[0]; j.px_cur[1] = px_cur[1];
  j.exposure_rat = (float)(cur.m_exposure_time / ref->frame->m_exposure_time);
  j.kf_gap_lt4 = (cur.keyFrameId_ - ref->frame->keyFrameId_ < 4) ? 1 : 0;
  return j;
}

std::vector<hso_align_out> Matcher::findMatchDirectBatch(const std::vector<const Point*>& pts, Frame& cur,
                                                         const std::vector<Vector2d>& px_cur, std::vector<Feature*>* ref_ftrs)
{
  std::vector<hso_align_out> out(pts.size());
  std::vector<hso_align_job> jobs;
  std::vector<size_t> slot;
  if (ref_ftrs) ref_ftrs->assign(pts.size(), nullptr);
  for (size_t i = 0; i < pts.size(); i++) {
    out[i] = hso_align_out{};
    out[i].px_cur[0] = px_cur[i][0]; out[i].px_cur[1] = px_cur[i][1];
    Feature* ref = nullptr;
    if (pts[i]->obs_.empty() || !pts[i]->getCloseViewObs(cur.pos(), ref)) continue;  // :276-286
    if (ref_ftrs) (*ref_ftrs)[i] = ref;
    jobs.push_back(make_align_job(*pts[i], ref, cur, px_cur[i]));
    slot.push_back(i);
  }
  if (!jobs.empty()) {
    std::vector<hso_align_out> res(jobs.size());
    const int rc = hso_gpu_align_batch(cur.ctx_, &cur.cam_->pod(), cur.id_, jobs.data(), (int)jobs.size(), res.data());
    api::check(cur.ctx_, rc, "Matcher");
    for (size_t k = 0; k < jobs.size(); k++) out[slot[k]] = res[k];
  }
  return out;
}

bool Matcher::findMatchDirect(const Point& pt, Frame& cur_frame, Vector2d& px_cur)
{
  std::vector<Feature*> refs;
  const std::vector<hso_align_out> r = findMatchDirectBatch({&pt}, cur_frame, {px_cur}, &refs);
  last_ = r[0];
  if (refs[0] == nullptr) return false;
  ref_ftr_ = refs[0];
  search_level_ = r[0].search_level;
  for (int i = 0; i < 4; i++) A_cur_ref_[i] = r[0].A_cur_ref[i];
  h_inv_ = r[0].h_inv;
  if (r[0].stage != HSO_ALIGN_REF_BORDER) { px_cur[0] = r[0].px_cur[0]; px_cur[1] = r[0].px_cur[1]; }  // :373
  return r[0].success != 0;
}

// ---------------------------------------------------------------- Reprojector
Reprojector::Reprojector(AbstractCamera* cam, int max_fts) : max_fts_(max_fts)
{
  cell_size = (int)floorf(std::sqrt((float)(cam->width() * cam->height()) / max_fts) * 0.6);   // caculateGridSize, :53-56
  grid_n_cols = (int)std::ceil((double)cam->width() / cell_size);
  grid_n_rows = (int)std::ceil((double)cam->height() / cell_size);
  cell_order.resize((size_t)grid_n_cols * grid_n_rows);
  for (size_t i = 0; i < cell_order.size(); i++) cell_order[i] = (int)i;
}

// what reprojectCell / reprojectCellAll do with one candidate (:366-412): true = matched
bool Reprojector::applyMatch(const Candidate& c, FramePtr frame)
{
  const hso_align_out& m = match_[c.slot];
  Point* pt = c.pt;
  if (proj_[c.slot].ref_obs < 0 || !m.success) {
    const int fails = ++pt->n_failed_reproj_;
    switch (pt->type_) {                                            // :376-392
      case Point::TYPE_UNKNOWN: if (fails > 15) dropUnknownPoint(pt); break;        // map_.safeDeletePoint
      case Point::TYPE_CANDIDATE: if (fails > 30) dropCandidatePoint(pt); break;    // deleteCandidatePoint
      case Point::TYPE_TEMPORARY: if (fails > 30) pt->isBad_ = true; break;
      default: break;
    }
    return false;
  }
  if (++pt->n_succeeded_reproj_ > 10 && pt->type_ == Point::TYPE_UNKNOWN) pt->type_ = Point::TYPE_GOOD;   // :412-423
  Feature* nf = new Feature();
  nf->frame = frame.get();
  nf->px = {m.px_cur[0], m.px_cur[1]};
  nf->f = frame->cam_->cam2world(nf->px);
  nf->level = m.search_level;
  nf->point = pt;
  const Feature* ref = ref_of_slot_[c.slot];
  if (ref->type == Feature::EDGELET) {
    nf->type = Feature::EDGELET;
    const double gx = m.A_cur_ref[0] * ref->grad[0] + m.A_cur_ref[1] * ref->grad[1];
    const double gy = m.A_cur_ref[2] * ref->grad[0] + m.A_cur_ref[3] * ref->grad[1];
    const double n = std::sqrt(gx * gx + gy * gy);
    nf->grad = {gx / n, gy / n};
  } else {
    nf->type = ref->type == Feature::GRADIENT ? Feature::GRADIENT : Feature::CORNER;
  }
  frame->fts_.push_back(nf);
  return true;
}

void Reprojector::reprojectMap(FramePtr frame, const std::vector<FramePtr>& kfs, std::vector<std::pair<FramePtr, size_t>>& overlap_kfs)
{
  // resetGrid, :77-84
  n_matches_ = 0; n_trials_ = 0; nFeatures_ = 0;
  // the points reprojectPoint would see, in the reference's order (:137-152, :176-199)
  std::vector<Point*> pts;
  std::vector<size_t> kf_of_pt;
  for (const FramePtr& kf : kfs) {
    if (overlap_kfs.size() >= max_n_kfs) break;
    overlap_kfs.push_back({kf, 0});
    const size_t row = overlap_kfs.size() - 1;
    for (Feature* ft : kf->fts_) {
      Point* p = ft->point;
      if (!p || p->type_ == Point::TYPE_TEMPORARY || p->last_projected_kf_id_ == frame->id_) continue;   // seen from an earlier keyframe
      p->last_projected_kf_id_ = frame->id_;
      pts.push_back(p); kf_of_pt.push_back(row);
    }
  }
  if (pts.empty()) return;
  projectAndMatch(frame, pts);
  std::vector<Candidate> all;                       // allPixelToDistribute, in projection order
  for (size_t i = 0; i < pts.size(); i++) {
    if (!proj_[i].projected) continue;              // reprojectPoint returned false
    all.push_back(Candidate{pts[i], {proj_[i].px[0], proj_[i].px[1]}, (int)i});
    overlap_kfs[kf_of_pt[i]].second++;
    nFeatures_++;
  }
  selectMatches(frame, all);
}

void Reprojector::projectAndMatch(FramePtr frame, const std::vector<Point*>& pts)
{
  // flatten: keyframe table over every frame a host feature or an observation lives in
  std::vector<hso_kf> kft;
  std::vector<const Frame*> kf_frames;
  auto kf_index = [&](const Frame* f) {
    for (size_t k = 0; k < kf_frames.size(); k++) if (kf_frames[k] == f) return (int)k;
    hso_kf r{};
    r.frame_id = f->id_; r.T_f_w = f->T_f_w_.v; r.exposure_time = f->m_exposure_time; r.keyframe_id = f->keyFrameId_;
    kft.push_back(r); kf_frames.push_back(f);
    return (int)kft.size() - 1;
  };
  std::vector<hso_map_point> mp(pts.size());
  std::vector<hso_obs> obs;
  std::vector<const Feature*> obs_ftr;
  for (size_t i = 0; i < pts.size(); i++) {
    const Point* p = pts[i];
    hso_map_point& m = mp[i];
    m = hso_map_point{};
    for (int k = 0; k < 3; k++) { m.pos[k] = p->pos_[k]; m.host_f[k] = p->hostFeature_->f[k]; }
    m.idist = p->idist_;
    m.host_kf = kf_index(p->hostFeature_->frame);
    m.obs_begin = (int)obs.size(); m.obs_count = (int)p->obs_.size();
    for (const Feature* o : p->obs_) {
      hso_obs ho{};
      ho.kf = kf_index(o->frame); ho.level = o->level; ho.type = (int)o->type;
      ho.px[0] = o->px[0]; ho.px[1] = o->px[1];
      for (int k = 0; k < 3; k++) ho.f[k] = o->f[k];
      ho.grad[0] = o->grad[0]; ho.grad[1] = o->grad[1];
      obs.push_back(ho); obs_ftr.push_back(o);
    }
  }
  proj_.assign(pts.size(), hso_reproj_point{});
  match_.assign(pts.size(), hso_align_out{});
  api::reproject_match(frame->ctx_, &frame->cam_->pod(), frame->id_, &frame->T_f_w_.v, frame->m_exposure_time, frame->keyFrameId_,
                       kft.data(), (int)kft.size(), mp.data(), (int)mp.size(), obs.data(), (int)obs.size(), cell_size, grid_n_cols,
                       proj_.data(), match_.data());
  ref_of_slot_.assign(pts.size(), nullptr);
  for (size_t i = 0; i < pts.size(); i++)
    if (proj_[i].projected && proj_[i].ref_obs >= 0) ref_of_slot_[i] = obs_ftr[proj_[i].ref_obs];
}

// Which candidates get examined, in which order, and which of them become features (:261-306, :352-429, :556-612) is a function of
// the candidates' cells, quality keys and match results, the cell order and max_fts: hso_gpu_reproject_select evaluates it (the
// engine runs the same policy fused behind the matcher); here the list it returns is walked and the bookkeeping applied.
void Reprojector::selectMatches(FramePtr frame, const std::vector<Candidate>& all)
{
  const int n = (int)all.size();
  if (n == 0) return;
  std::vector<int32_t> cell((size_t)n), examined((size_t)n), begin{0, n};
  std::vector<uint8_t> quality((size_t)n), flags((size_t)n);
  for (int i = 0; i < n; i++) {
    const Candidate& c = all[(size_t)i];
    cell[(size_t)i] = proj_[c.slot].cell;
    const bool gone = c.pt->type_ == Point::TYPE_DELETED;
    quality[(size_t)i] = gone ? 0 : (uint8_t)(((int)c.pt->type_ << 4) | (int)c.pt->ftr_type_);
    flags[(size_t)i] = (uint8_t)((proj_[c.slot].ref_obs >= 0 && match_[c.slot].success ? 1 : 0) | (gone ? 2 : 0));
  }
  int32_t counts[4] = {0, 0, 0, 0};
  api::check(frame->ctx_, hso_gpu_reproject_select(frame->ctx_, begin.data(), 1, cell.data(), quality.data(), flags.data(), cell_order.data(),
                                                   (int)cell_order.size(), max_fts_, examined.data(), counts), "Reprojector");
  for (int k = 0; k < counts[0]; k++) {
    const Candidate& c = all[(size_t)(examined[(size_t)k] & 0x7fffffff)];
    if (c.pt->type_ == Point::TYPE_DELETED) continue;
    const bool made = applyMatch(c, frame);
    (void)made;   // == (examined[k] < 0): the policy saw the same success flag
  }
  n_trials_ = (size_t)counts[0]; n_matches_ = (size_t)counts[1];
}

// ---------------------------------------------------------------- pose_optimizer
void pose_optimizer::optimizeLevenbergMarquardt3rd(const double reproj_thresh, const size_t n_iter, const bool verbose,
                                                   FramePtr& frame, double& estimated_scale, double& error_init,
                                                   double& error_final, size_t& num_obs)
{
  (void)verbose;
  // flatten fts_ in list order; host keyframe poses are de-duplicated into a table
  std::vector<hso_pose_feat> feats;
  std::vector<hso_se3> poses;
  std::vector<const Frame*> pose_of;
  feats.reserve(frame->fts_.size());
  for (Feature* ft : frame->fts_) {
    hso_pose_feat pf{};
    pf.has_point = ft->point != nullptr;
    pf.type = (int)ft->type;
    pf.level = ft->level;
    pf.f[0] = ft->f[0]; pf.f[1] = ft->f[1]; pf.f[2] = ft->f[2];
    pf.grad[0] = ft->grad[0]; pf.grad[1] = ft->grad[1];
    if (ft->point) {
      const Feature* host = ft->point->hostFeature_;
      pf.temporary = ft->point->type_ == Point::TYPE_TEMPORARY;
      pf.host_f[0] = host->f[0]; pf.host_f[1] = host->f[1]; pf.host_f[2] = host->f[2];
      pf.idist = ft->point->idist_;
      size_t k = 0;
      while (k < pose_of.size() && pose_of[k] != host->frame) k++;
      if (k == pose_of.size()) { pose_of.push_back(host->frame); poses.push_back(host->frame->T_f_w_.v); }
      pf.host_pose = (int)k;
    }
    feats.push_back(pf);
  }
  hso_pose_job job{};
  job.feats = feats.data(); job.n_feats = (int)feats.size();
  job.poses_f_w = poses.data(); job.n_poses = (int)poses.size();
  job.T_f_w = frame->T_f_w_.v;
  job.reproj_thresh = reproj_thresh; job.n_iter = (int)n_iter;
  hso_pose_result res{};
  std::vector<uint8_t> mask(feats.size() ? feats.size() : 1, 0);
  api::pose_optimize(frame->ctx_, &frame->cam_->pod(), &job, &res, mask.data());
  estimated_scale = res.estimated_scale; error_init = res.error_init; error_final = res.error_final;
  num_obs = (size_t)res.num_obs;
  if (res.status != 0) return;                      // no residuals: the reference returns early (:456)
  frame->T_f_w_.v = res.T_f_w;
  for (int i = 0; i < 36; i++) frame->Cov_[i] = res.cov[i];
  frame->m_error_in_px = res.error_in_px;
  size_t i = 0;
  for (Feature* ft : frame->fts_) { if (mask[i]) ft->point = nullptr; ++i; }   // :722-748
}

// ---------------------------------------------------------------- DepthFilter
Seed::Seed(Feature* ftr_, float depth_mean, float depth_min, float converge_threshold)
    : ftr(ftr_), mu((float)(1.0 / depth_mean)), z_range((float)(1.0 / depth_min)), sigma2(z_range * z_range / 36)
{
  vec_distance.push_back(depth_mean);    // src/depth_filter.cpp:64
  converge_thresh = converge_threshold;
}

FeatureExtractor::FeatureExtractor(int width, int height, int cellSize, int levels, bool isInit, int max_fts)
    : width_(width), height_(height), cellSize_(cellSize), nLevels_(levels), nFeatures_(isInit ? 2000 : max_fts + 100), isInit_(isInit)
{
}

void FeatureExtractor::setExistingFeatures(const Features& fts)
{
  for (const Feature* ftr : fts) {
    hso_keypoint k{};
    k.x = (float)ftr->px[0]; k.y = (float)ftr->px[1];
    k.species = HSO_KP_OCCUR;
    allFeturesToDistribute_.push_back(k);
  }
  extFeatures_ += fts.size();
}

void FeatureExtractor::detect(Frame* frame, float initThresh, float minThresh, Features& fts, Frame* last_frame)
{
  (void)initThresh; (void)last_frame;   // initThresh_ is never read; the epipolar-hole filter is commented out in the reference (:798-803)
  minThresh_ = (int)minThresh;          // int minThresh_, include/hso/feature_detection.h:339
  const int64_t id = frame->id_;
  int corner_cap = 16384;
  const int edgelet_cap = ((width_ + 7) / 8) * ((height_ + 7) / 8);   // one per grid index; every level has that many
  std::vector<hso_corner> co, fill;
  std::vector<hso_edgelet> ed;
  std::vector<int32_t> nc(nLevels_), ne(nLevels_, 0);
  int32_t n_fill = 0;
  if (isInit_) fill.resize(edgelet_cap); else ed.resize((size_t)nLevels_ * edgelet_cap);
  for (;;) {
    co.resize((size_t)nLevels_ * corner_cap);
    // :439-447: fillingHole on level 0 while initialising, the edgelets of every level otherwise
    api::detect_candidates(frame->ctx_, isInit_, id, nLevels_, minThresh_, co.data(), corner_cap, nc.data(), ed.data(), fill.data(),
                           edgelet_cap, isInit_ ? &n_fill : ne.data());
    int most = 0;
    for (int c : nc) most = c > most ? c : most;
    if (most <= corner_cap) break;
    corner_cap = most;                  // a frame with more corners than the first guess: once more with room for all
  }
  api::trace_candidates(isInit_, id, nLevels_, minThresh_, co.data(), corner_cap, nc.data(), ed.data(), fill.data(), edgelet_cap,
                        isInit_ ? &n_fill : ne.data());
  // featurePerLevel_[L] = corners then edgelets (:518-545, :749-830), appended level by level (:449-451)
  for (int L = 0; L < nLevels_; L++) {
    for (int i = 0; i < nc[L]; i++) {
      const hso_corner& c = co[(size_t)L * corner_cap + i];
      hso_keypoint k{};
      k.x = (float)(c.x << L); k.y = (float)(c.y << L); k.response = c.response; k.level = L; k.species = HSO_KP_CORNER_HIGH;
      allFeturesToDistribute_.push_back(k);
    }
    for (int i = 0; L == 0 && i < n_fill; i++) {   // fillingHole's key points follow the level-0 corners (kGrad, :1150-1151)
      hso_keypoint k{};
      k.x = (float)fill[i].x; k.y = (float)fill[i].y; k.response = fill[i].response; k.level = 0; k.species = HSO_KP_GRAD;
      allFeturesToDistribute_.push_back(k);
    }
    for (int i = 0; i < ne[L]; i++) {
      const hso_edgelet& e = ed[(size_t)L * edgelet_cap + i];
      hso_keypoint k{};
      k.x = (float)(e.x << L); k.y = (float)(e.y << L); k.response = e.grad; k.level = L; k.species = HSO_KP_EDGELET;
      k.gx = e.gx; k.gy = e.gy;
      allFeturesToDistribute_.push_back(k);
    }
  }
  std::vector<hso_keypoint> sel(allFeturesToDistribute_.size() + 1);
  const int n = api::select_octree(allFeturesToDistribute_.data(), (int)allFeturesToDistribute_.size(), width_, height_, nFeatures_,
                                   sel.data(), (int)sel.size());
  for (int i = 0; i < n; i++) {          // :457-484
    const hso_keypoint& k = sel[i];
    Feature* f = new Feature();
    f->frame = frame;
    f->px = {(double)k.x, (double)k.y};
    f->f = frame->cam_->cam2world(f->px);
    f->level = k.level;
    if (k.species == HSO_KP_CORNER_HIGH) {
      f->type = Feature::CORNER;
    } else {
      f->type = k.species == HSO_KP_GRAD ? Feature::GRADIENT : Feature::EDGELET;
      // fillingHole never sets KeyPoint::gx / gy (the reference normalises uninitialised ints there,
      // :466-468); the mirror keeps the default direction for those
      const double gx = k.gx, gy = k.gy, nrm = std::sqrt(gx * gx + gy * gy);
      if (nrm > 0) f->grad = {gx / nrm, gy / nrm};    // Vector2d::normalize()
    }
    fts.push_back(f);
  }
  allFeturesToDistribute_.clear();       // resetGrid + clear, :486-496
  extFeatures_ = 0;
}

void DepthFilter::addKeyframe(FramePtr frame, double depth_mean, double depth_min, float converge_thresh)
{
  new_keyframe_min_depth_ = depth_min;
  new_keyframe_mean_depth_ = depth_mean;
  convergence_sigma2_thresh_ = converge_thresh;
  initializeSeeds(frame);
}

void DepthFilter::initializeSeeds(FramePtr frame)
{
  if (!featureExtractor_) throw std::logic_error("DepthFilter: no FeatureExtractor");
  Features new_features;
  featureExtractor_->setExistingFeatures(frame->fts_);
  featureExtractor_->detect(frame.get(), 20, frame->gradMean_, new_features, nullptr);
  for (Feature* ftr : new_features)
    seeds_.emplace_back(ftr, (float)new_keyframe_mean_depth_, (float)new_keyframe_min_depth_, convergence_sigma2_thresh_);
}

size_t DepthFilter::observeDepth(FramePtr frame)
{
  std::vector<hso_seed> in;
  in.reserve(seeds_.size());
  for (const Seed& s : seeds_) {  // list order = seed index
    hso_seed h{};
    h.ref_frame_id = s.ftr->frame->id_;
    h.level = s.ftr->level; h.type = (int)s.ftr->type;
    h.px[0] = s.ftr->px[0]; h.px[1] = s.ftr->px[1];
    h.f[0] = s.ftr->f[0]; h.f[1] = s.ftr->f[1]; h.f[2] = s.ftr->f[2];
    h.grad[0] = s.ftr->grad[0]; h.grad[1] = s.ftr->grad[1];
    h.T_ref_w = s.ftr->frame->T_f_w_.v;
    h.ref_exposure = s.ftr->frame->m_exposure_time;
    h.mu = s.mu; h.sigma2 = s.sigma2; h.b = s.b;
    in.push_back(h);
  }
  if (in.empty()) return 0;
  std::vector<hso_seed_out> out(in.size());
  api::seed_observe(frame->ctx_, &frame->cam_->pod(), frame->id_, &frame->T_f_w_.v, frame->m_exposure_time, px_error_angle_, in.data(),
                    (int)in.size(), out.data());
  size_t n_ok = 0, k = 0;
  for (auto it = seeds_.begin(); it != seeds_.end(); k++) {
    const hso_seed_out& o = out[k];
    if (!o.is_valid) { it = seeds_.erase(it); continue; }            // :618-622
    it->is_update = o.is_update != 0;
    it->mu = o.mu; it->sigma2 = o.sigma2; it->b = o.b;
    if (o.result == 1) {                                              // :640-650
      it->last_matched_px = {o.px_cur[0], o.px_cur[1]};
      it->last_matched_level = o.search_level;
      ++n_ok;
    }
    ++it;
  }
  return n_ok;
}

}  // namespace hso
