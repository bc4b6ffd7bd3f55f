// hso_vo.cpp — the per-frame pipeline (see hso_vo.h).  Control flow follows the reference line by line; every
// numeric body is a device call through hso_api.h.
#include "hso_vo.h"
#include <algorithm>
#include <cassert>
#include <chrono>
#include <cmath>
#include <limits>
#include <numeric>
#include "hso_api.h"

namespace hso {

thread_local Config* Config::current_ = nullptr;
api::Trace& api::trace() { static thread_local Trace t; return t; }
api::Router*& api::router() { static thread_local Router* r = nullptr; return r; }

static Vector3d vsub(const Vector3d& a, const Vector3d& b) { return {a[0] - b[0], a[1] - b[1], a[2] - b[2]}; }
static double vnorm(const Vector3d& a) { return std::sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]); }
static Vector3d vscale(const Vector3d& a, double s) { return {a[0] * s, a[1] * s, a[2] * s}; }
static Vector2d project2d(const Vector3d& v) { return {v[0] / v[2], v[1] / v[2]}; }   // vikit/math_utils.h:57-60
template <typename T> static T getMedian(std::vector<T>& v)   // vikit/math_utils.h:119-126 (permutes v)
{
  typename std::vector<T>::iterator it = v.begin() + (long)std::floor(v.size() / 2);
  std::nth_element(v.begin(), it, v.end());
  return *it;
}

// ------------------------------------------------------------------------------------------------ Frame
void Frame::setKeyframe()
{
  is_keyframe_ = true;
  setKeyPoints();
  keyFrameCounter_++;
  keyFrameId_ = keyFrameCounter_;
}

void Frame::setKeyPoints()
{
  for (size_t i = 0; i < 5; ++i)
    if (key_pts_[i] != nullptr)
      if (key_pts_[i]->point == nullptr) key_pts_[i] = nullptr;
  for (Feature* ftr : fts_)
    if (ftr->point != nullptr) checkKeyPoints(ftr);
}

void Frame::checkKeyPoints(Feature* ftr)
{
  const int cu = cam_->width() / 2;
  const int cv = cam_->height() / 2;
  const Vector2d uv = ftr->px;
  if (key_pts_[0] == nullptr) key_pts_[0] = ftr;
  else if (std::max(std::fabs(ftr->px[0] - cu), std::fabs(ftr->px[1] - cv)) <
           std::max(std::fabs(key_pts_[0]->px[0] - cu), std::fabs(key_pts_[0]->px[1] - cv)))
    key_pts_[0] = ftr;
  if (uv[0] >= cu && uv[1] >= cv) {
    if (key_pts_[1] == nullptr) key_pts_[1] = ftr;
    else if ((uv[0] - cu) * (uv[1] - cv) > (key_pts_[1]->px[0] - cu) * (key_pts_[1]->px[1] - cv)) key_pts_[1] = ftr;
  }
  if (uv[0] >= cu && uv[1] < cv) {
    if (key_pts_[2] == nullptr) key_pts_[2] = ftr;
    else if ((uv[0] - cu) * -(uv[1] - cv) > (key_pts_[2]->px[0] - cu) * -(key_pts_[2]->px[1] - cv)) key_pts_[2] = ftr;
  }
  if (uv[0] < cu && uv[1] >= cv) {
    if (key_pts_[3] == nullptr) key_pts_[3] = ftr;
    else if (-(uv[0] - cu) * (uv[1] - cv) > -(key_pts_[3]->px[0] - cu) * (key_pts_[3]->px[1] - cv)) key_pts_[3] = ftr;
  }
  if (uv[0] < cu && uv[1] < cv) {
    if (key_pts_[4] == nullptr) key_pts_[4] = ftr;
    else if (-(uv[0] - cu) * -(uv[1] - cv) > -(key_pts_[4]->px[0] - cu) * -(key_pts_[4]->px[1] - cv)) key_pts_[4] = ftr;
  }
}

void Frame::removeKeyPoint(Feature* ftr)
{
  bool found = false;
  for (Feature*& i : key_pts_)
    if (i == ftr) { i = nullptr; found = true; }
  if (found) setKeyPoints();
}

bool Frame::isVisible(const Vector3d& xyz_w) const
{
  const Vector3d xyz_f = T_f_w_ * xyz_w;
  if (xyz_f[2] < 0.0) return false;
  const Vector2d px = cam_->world2cam(xyz_f);
  return px[0] >= 0.0 && px[1] >= 0.0 && px[0] < cam_->width() && px[1] < cam_->height();
}

bool frame_utils::getSceneDepth(const Frame& frame, double& depth_mean, double& depth_min)
{
  std::vector<double> depth_vec;
  depth_vec.reserve(frame.fts_.size());
  depth_min = std::numeric_limits<double>::max();
  for (const Feature* ft : frame.fts_)
    if (ft->point != nullptr) {
      const double z = (frame.T_f_w_ * ft->point->pos_)[2];
      depth_vec.push_back(z);
      depth_min = std::fmin(z, depth_min);
    }
  if (depth_vec.empty()) return false;
  depth_mean = getMedian(depth_vec);
  return true;
}

bool frame_utils::getSceneDistance(const Frame& frame, double& distance_mean)
{
  std::vector<double> distance_vec;
  distance_vec.reserve(frame.fts_.size());
  for (const Feature* ft : frame.fts_) {
    if (ft->point == nullptr) continue;
    distance_vec.push_back(vnorm(frame.T_f_w_ * ft->point->pos_));
  }
  if (distance_vec.empty()) return false;
  distance_mean = getMedian(distance_vec);
  return true;
}

// ------------------------------------------------------------------------------------------------ Map
void Map::reset()
{
  keyframes_.clear();
  point_candidates_.reset();
  emptyTrash();
  // nothing dereferences a trashed point after a reset (the handler drops its frames and seeds with it): free the storage
  for (Point* p : graveyard_) if (!point_candidates_.graveyard_.count(p)) delete p;
  graveyard_.clear();
  for (Point* p : point_candidates_.graveyard_) delete p;
  point_candidates_.graveyard_.clear();
}

void Map::removePtFrameRef(Frame* frame, Feature* ftr)
{
  if (ftr->point == nullptr) return;
  Point* pt = ftr->point;
  ftr->point = nullptr;
  if (pt->obs_.size() <= 2) { safeDeletePoint(pt); return; }
  pt->deleteFrameRef(frame);
  frame->removeKeyPoint(ftr);
}

void Map::safeDeletePoint(Point* pt)
{
  for (Feature* ftr : pt->obs_) { ftr->point = nullptr; ftr->frame->removeKeyPoint(ftr); }
  pt->obs_.clear();
  deletePoint(pt);
}

void Map::safeDeleteTempPoint(std::pair<Point*, Feature*>& p)
{
  if (p.first->seedStates_ == -1) {
    if (p.first->isBad_) safeDeletePoint(p.first);
    else {
      p.first->pos_ = p.first->hostFeature_->frame->T_f_w_.inverse() * vscale(p.first->hostFeature_->f, 1.0 / p.first->idist_);
      if (p.first->obs_.size() == 1) {
        p.first->type_ = Point::TYPE_CANDIDATE;
        p.first->n_failed_reproj_ = 0;
        p.first->n_succeeded_reproj_ = 0;
        point_candidates_.candidates_.push_back(MapPointCandidates::PointCandidate(p.first, p.first->obs_.front()));
      } else {
        p.first->type_ = Point::TYPE_UNKNOWN;
        p.first->n_failed_reproj_ = 0;
        p.first->n_succeeded_reproj_ = 0;
        p.second->frame->addFeature(p.second);
      }
    }
  } else {
    // the seed converged on its own: a regular point replaced the temporary one (depth_filter.cpp:454-457); the
    // observations that point to the temporary point are released, the host feature now belongs to the new point
    for (Feature* o : p.first->obs_)
      if (o->point != p.second->point) { o->point = nullptr; o->frame->removeKeyPoint(o); }
    p.first->obs_.clear();
    deletePoint(p.first);
  }
}

void Map::deletePoint(Point* pt)
{
  pt->type_ = Point::TYPE_DELETED;
  trash_points_.push_back(pt);
}

void Map::getCloseKeyframes(const FramePtr& frame, std::list<std::pair<FramePtr, double>>& close_kfs) const
{
  for (const FramePtr& kf : keyframes_)
    for (Feature* keypoint : kf->key_pts_) {
      if (keypoint == nullptr) continue;
      if (frame->isVisible(keypoint->point->pos_)) {
        close_kfs.push_back(std::make_pair(kf, vnorm(vsub(frame->T_f_w_.translation(), kf->T_f_w_.translation()))));
        break;
      }
    }
}

FramePtr Map::getClosestKeyframe(const FramePtr& frame) const
{
  std::list<std::pair<FramePtr, double>> close_kfs;
  getCloseKeyframes(frame, close_kfs);
  if (close_kfs.empty()) return nullptr;
  close_kfs.sort([](const std::pair<FramePtr, double>& a, const std::pair<FramePtr, double>& b) { return a.second < b.second; });
  if (close_kfs.front().first != frame) return close_kfs.front().first;
  close_kfs.pop_front();
  return close_kfs.empty() ? nullptr : close_kfs.front().first;
}

bool Map::getKeyframeById(int id, FramePtr& frame) const
{
  for (const FramePtr& kf : keyframes_)
    if (kf->id_ == id) { frame = kf; return true; }
  return false;
}

void Map::emptyTrash()
{
  // the reference deletes the points here; observations that the pose optimiser detached (feature->point = NULL
  // without touching obs_) can still name a trashed point there, so the mirror keeps the storage until reset()
  graveyard_.insert(trash_points_.begin(), trash_points_.end());
  trash_points_.clear();
  point_candidates_.emptyTrash();
}

void MapPointCandidates::newCandidatePoint(Point* point, double)
{
  point->type_ = Point::TYPE_CANDIDATE;
  candidates_.push_back(PointCandidate(point, point->obs_.front()));
}

void MapPointCandidates::addPauseSeedPoint(Point* point)
{
  temporaryPoints_.push_back(std::make_pair(point, point->obs_.front()));
}

void MapPointCandidates::addCandidatePointToFrame(FramePtr frame)
{
  PointCandidateList::iterator it = candidates_.begin();
  while (it != candidates_.end()) {
    if (it->first->obs_.front()->frame == frame.get()) {
      it->first->type_ = Point::TYPE_UNKNOWN;
      it->first->n_failed_reproj_ = 0;
      it->second->frame->addFeature(it->second);
      it = candidates_.erase(it);
    } else
      ++it;
  }
}

bool MapPointCandidates::deleteCandidatePoint(Point* point)
{
  for (auto it = candidates_.begin(), ite = candidates_.end(); it != ite; ++it)
    if (it->first == point) { deleteCandidate(*it); candidates_.erase(it); return true; }
  return false;
}

void MapPointCandidates::changeCandidatePosition(Frame* frame)
{
  for (PointCandidate& c : candidates_)
    if (c.second->frame->id_ == frame->id_) c.first->pos_ = frame->T_f_w_.inverse() * vscale(c.second->f, 1.0 / c.first->idist_);
}

void MapPointCandidates::removeFrameCandidates(FramePtr frame)
{
  auto it = candidates_.begin();
  while (it != candidates_.end()) {
    if (it->second->frame == frame.get()) { deleteCandidate(*it); it = candidates_.erase(it); }
    else ++it;
  }
}

void MapPointCandidates::reset()
{
  for (PointCandidate& c : candidates_) { delete c.first; delete c.second; }
  candidates_.clear();
  temporaryPoints_.clear();
}

void MapPointCandidates::deleteCandidate(PointCandidate& c)
{
  delete c.second; c.second = nullptr;        // the host feature lives in no frame's list yet
  c.first->type_ = Point::TYPE_DELETED;
  trash_points_.push_back(c.first);
}

void MapPointCandidates::emptyTrash() { graveyard_.insert(trash_points_.begin(), trash_points_.end()); trash_points_.clear(); }

// ------------------------------------------------------------------------------------------------ seeds
static hso_seed flatten_seed(const Seed& s)
{
  hso_seed h{};
  h.ref_frame_id = s.ftr->frame->id_;
  h.level = s.ftr->level; h.type = (int)s.ftr->type;
  h.px[0] = s.ftr->px[0]; h.px[1] = s.ftr->px[1];
  h.f[0] = s.ftr->f[0]; h.f[1] = s.ftr->f[1]; h.f[2] = s.ftr->f[2];
  h.grad[0] = s.ftr->grad[0]; h.grad[1] = s.ftr->grad[1];
  h.T_ref_w = s.ftr->frame->T_f_w_.v;
  h.ref_exposure = s.ftr->frame->m_exposure_time;
  h.mu = s.mu; h.sigma2 = s.sigma2; h.b = s.b;
  return h;
}

void SeedFilter::addFrame(FramePtr frame) { updateSeeds(frame); }

void SeedFilter::addKeyframe(FramePtr frame, double depth_mean, double depth_min, float converge_thresh)
{
  new_keyframe_min_depth_ = depth_min;
  new_keyframe_mean_depth_ = depth_mean;
  convergence_sigma2_thresh_ = converge_thresh;
  // what updateSeedsLoop does with a keyframe: updateSeeds, then initializeSeeds (src/depth_filter.cpp:317-327)
  updateSeeds(frame);
  ++Seed::batch_counter;       // initializeSeeds, :181
  initializeSeeds(frame);
}

void SeedFilter::updateSeeds(FramePtr frame)
{
  active_frame_ = frame;
  if (px_error_angle_ == -1) {                                  // :360-366
    const double focal_length = frame->cam_->errorMultiplier2();
    const double px_noise = 1.0;
    px_error_angle_ = std::atan(px_noise / (2.0 * focal_length)) * 2.0;
  }
  for (auto it = seeds_.begin(); it != seeds_.end();) {         // :368-401: seeds older than max_n_kfs keyframes
    if ((Seed::batch_counter - it->batch_id) > max_n_kfs) {
      if (it->temp != nullptr && it->haveReprojected) it->temp->seedStates_ = -1;
      else { delete it->ftr; it->ftr = nullptr; }
      it = seeds_.erase(it);
      continue;
    }
    ++it;
  }
  observeDepth();
  activateConverged();
  if (m_v_n_converge.size() > size_t(0.5 * Config::get().max_fts))     // :503-507
    nMeanConvergeFrame_ = std::accumulate(m_v_n_converge.begin(), m_v_n_converge.end(), 0) / m_v_n_converge.size();
  else
    nMeanConvergeFrame_ = 6;
}

// observeDepthRow for every seed (src/depth_filter.cpp:580-675), one device call
void SeedFilter::observeDepth()
{
  if (seeds_.empty()) return;
  FramePtr frame = active_frame_;
  std::vector<hso_seed> in;
  in.reserve(seeds_.size());
  for (const Seed& s : seeds_) in.push_back(flatten_seed(s));
  std::vector<hso_seed_out> out(in.size());
  api::seed_observe(frame->ctx_, &frame->cam_->pod(), frame->id_, &frame->T_f_w_.v, frame->m_exposure_time, px_error_angle_, in.data(),
                    (int)in.size(), out.data());
  size_t k = 0;
  for (auto it = seeds_.begin(); it != seeds_.end(); ++it, ++k) {
    const hso_seed_out& o = out[k];
    it->is_update = o.is_update != 0;
    if (!it->is_update) continue;                                // :593-606: not visible in the active frame
    if (it->optFrames_A.size() < 15) it->optFrames_A.push_back(frame);   // :610-611
    if (!o.is_valid) it->isValid = false;                        // :618
    it->mu = o.mu; it->sigma2 = o.sigma2; it->b = o.b;
    if (o.result != 1) continue;
    it->vec_distance.push_back((float)(1.0 / it->mu));           // :660
    it->last_matched_px = {o.px_cur[0], o.px_cur[1]};
    it->last_matched_level = o.search_level;
    if (frame->isKeyframe()) {                                   // :669-673: FeatureExtractor::setGridOccpuancy
      hso_keypoint kp{};
      kp.x = (float)o.px_cur[0]; kp.y = (float)o.px_cur[1]; kp.species = HSO_KP_OCCUR;
      featureExtractor_->allFeturesToDistribute_.push_back(kp);
      featureExtractor_->extFeatures_++;
    }
  }
}

// the convergence loop of updateSeeds (:405-497): activatePoint (+ seedOptimizer) for every converged seed in one
// device call, then the reference's bookkeeping seed by seed
void SeedFilter::activateConverged()
{
  std::vector<std::list<Seed>::iterator> conv;
  for (auto it = seeds_.begin(); it != seeds_.end();) {
    if (std::sqrt(it->sigma2) < it->z_range / it->converge_thresh) { conv.push_back(it); ++it; }
    else if (!it->isValid) it = seeds_.erase(it);                // "z_min is NaN", :494-498
    else ++it;
  }
  if (conv.empty()) return;
  n_converged_ += conv.size();
  FramePtr frame = active_frame_;
  std::vector<hso_seed> in;
  std::vector<int32_t> begin(1, 0);
  std::vector<hso_activate_target> targets;
  for (auto it : conv) {
    it->opt_id = it->mu;                                         // :731
    in.push_back(flatten_seed(*it));
    for (const std::vector<FramePtr>* lst : {&it->optFrames_P, &it->optFrames_A})
      for (const FramePtr& t : *lst) {
        hso_activate_target a{};
        a.frame_id = t->id_; a.T_f_w = t->T_f_w_.v; a.exposure = t->m_exposure_time;
        targets.push_back(a);
      }
    begin.push_back((int32_t)targets.size());
  }
  std::vector<hso_activate_out> out(in.size());
  hso_activate_target none{};
  api::seed_activate(frame->ctx_, &frame->cam_->pod(), in.data(), (int)in.size(), begin.data(), targets.empty() ? &none : targets.data(),
                     (int)nMeanConvergeFrame_, out.data());
  for (size_t k = 0; k < conv.size(); ++k) {
    auto it = conv[k];
    bool isValid = out[k].is_valid != 0;                         // -1: activatePoint left it untouched (true)
    if (out[k].activated) { it->opt_id = (float)out[k].opt_id; it->mu = it->opt_id; n_activated_++; }   // :418-419
    const Vector3d pHost = vscale(it->ftr->f, 1.0 / it->mu);
    if (it->mu < 1e-10 || pHost[2] < 1e-10) isValid = false;     // :423-424
    if (!isValid) {
      if (it->temp != nullptr && it->haveReprojected) it->temp->seedStates_ = -1;
      seeds_.erase(it);
      continue;
    }
    if (m_v_n_converge.size() > (size_t)Config::get().max_fts) m_v_n_converge.erase(m_v_n_converge.begin());
    m_v_n_converge.push_back(it->vec_distance.size());
    const Vector3d xyz_world = it->ftr->frame->T_f_w_.inverse() * pHost;
    Point* point = new Point(xyz_world, it->ftr);
    point->idist_ = it->mu;
    point->hostFeature_ = it->ftr;
    point->ftr_type_ = it->ftr->type == Feature::EDGELET ? Point::FEATURE_EDGELET
                       : it->ftr->type == Feature::CORNER ? Point::FEATURE_CORNER : Point::FEATURE_GRADIENT;
    it->ftr->point = point;
    if (it->temp != nullptr && it->haveReprojected) it->temp->seedStates_ = 1;
    seed_converged_cb_(cb_user_, point, it->sigma2);              // MapPointCandidates::newCandidatePoint
    seeds_.erase(it);
  }
}

// ------------------------------------------------------------------------------------------------ Reprojector
void MapReprojector::reprojectMap(FramePtr frame, std::vector<std::pair<FramePtr, size_t>>& overlap_kfs)
{
  // resetGrid, :77-86 (the cell order stays as it is: see the header)
  n_matches_ = 0; n_trials_ = 0; n_seeds_ = 0; nFeatures_ = 0;
  for (Cell& c : cells_) c.clear();

  if (!map_.point_candidates_.temporaryPoints_.empty()) {      // :98-118: temporary points whose seed has finished
    size_t n = 0;
    auto ite = map_.point_candidates_.temporaryPoints_.begin();
    while (ite != map_.point_candidates_.temporaryPoints_.end()) {
      if (ite->first->seedStates_ == 0) { ite++; continue; }
      map_.safeDeleteTempPoint(*ite);
      ite = map_.point_candidates_.temporaryPoints_.erase(ite);
      n++;
    }
    sum_seed_ -= n;
  }
  overlap_kfs.reserve(max_n_kfs);

  // the points reprojectPoint would be called for, in the reference's order, with what happens when it fails
  enum Src { SRC_KF, SRC_CANDIDATE, SRC_TEMP };
  std::vector<Point*> pts;
  std::vector<int> src;
  std::vector<size_t> kf_of_pt;
  auto take_kf_points = [&](const FramePtr& kf) {
    overlap_kfs.push_back(std::pair<FramePtr, size_t>(kf, 0));
    for (Feature* ft : kf->fts_) {
      if (ft->point == nullptr) continue;
      if (ft->point->type_ == Point::TYPE_TEMPORARY) continue;
      if (ft->point->last_projected_kf_id_ == frame->id_) continue;
      ft->point->last_projected_kf_id_ = frame->id_;
      pts.push_back(ft->point); src.push_back(SRC_KF); kf_of_pt.push_back(overlap_kfs.size() - 1);
    }
  };
  FramePtr LastFrame = frame->m_last_frame;
  size_t nCovisibilityGraph = 0;
  for (Frame* repframe : LastFrame->connectedKeyFrames) {       // :124-156
    FramePtr repFrame;
    if (!map_.getKeyframeById(repframe->id_, repFrame)) continue;
    if (repFrame->lastReprojectFrameId_ == frame->id_) continue;
    repFrame->lastReprojectFrameId_ = frame->id_;
    take_kf_points(repFrame);
    nCovisibilityGraph++;
  }
  LastFrame->connectedKeyFrames.clear();
  std::list<std::pair<FramePtr, double>> close_kfs;             // :161-199
  map_.getCloseKeyframes(frame, close_kfs);
  close_kfs.sort([](const std::pair<FramePtr, double>& a, const std::pair<FramePtr, double>& b) { return a.second < b.second; });
  size_t n = nCovisibilityGraph;
  for (auto it_frame = close_kfs.begin(); it_frame != close_kfs.end() && n < max_n_kfs; ++it_frame) {
    FramePtr ref_frame = it_frame->first;
    if (ref_frame->lastReprojectFrameId_ == frame->id_) continue;
    ref_frame->lastReprojectFrameId_ = frame->id_;
    take_kf_points(ref_frame);
    ++n;
  }
  for (auto& c : map_.point_candidates_.candidates_) { pts.push_back(c.first); src.push_back(SRC_CANDIDATE); kf_of_pt.push_back(0); }   // :207-226
  for (auto& tp : map_.point_candidates_.temporaryPoints_) {    // :230-254
    if (tp.first->isBad_) continue;
    tp.first->last_projected_kf_id_ = frame->id_;
    tp.first->pos_ = tp.second->frame->T_f_w_.inverse() * vscale(tp.second->f, 1.0 / tp.first->idist_);
    pts.push_back(tp.first); src.push_back(SRC_TEMP); kf_of_pt.push_back(0);
  }

  std::vector<Candidate> all;
  if (!pts.empty()) {
    projectAndMatch(frame, pts);
    for (size_t i = 0; i < pts.size(); ++i) {
      Point* pt = pts[i];
      if (proj_[i].projected) {
        const Candidate c{pt, {proj_[i].px[0], proj_[i].px[1]}, (int)i};
        cells_.at(proj_[i].cell).push_back(c);
        all.push_back(c);
        nFeatures_++;
        if (src[i] == SRC_KF) overlap_kfs[kf_of_pt[i]].second++;
        continue;
      }
      if (src[i] == SRC_CANDIDATE) {                             // :214-222
        pt->n_failed_reproj_ += 3;
        if (pt->n_failed_reproj_ > 30) map_.point_candidates_.deleteCandidatePoint(pt);
      } else if (src[i] == SRC_TEMP) {                           // :247-251
        pt->n_failed_reproj_ += 3;
        if (pt->n_failed_reproj_ > 30) pt->isBad_ = true;
      }
    }
  }
  selectMatches(frame, all);

  // :309-329 — too few matches: try the seeds that have nearly converged
  if (n_matches_ < 100 && reproject_unconverged_seeds && depth_filter_ != nullptr && !depth_filter_->seeds_.empty()) {
    std::vector<std::list<Seed>::iterator> sel;
    std::vector<hso_seed> in;
    for (auto it = depth_filter_->seeds_.begin(); it != depth_filter_->seeds_.end(); ++it)
      if (std::sqrt(it->sigma2) < it->z_range / reproject_seed_thresh && !it->haveReprojected) { sel.push_back(it); in.push_back(flatten_seed(*it)); }
    if (!sel.empty()) {
      std::vector<hso_reproj_point> sp(sel.size());
      std::vector<hso_align_out> sm(sel.size());
      api::seed_reproject_match(frame->ctx_, &frame->cam_->pod(), frame->id_, &frame->T_f_w_.v, frame->m_exposure_time, in.data(),
                                (int)in.size(), cell_size, grid_n_cols, sp.data(), sm.data());
      std::vector<std::vector<size_t>> sells(cells_.size());
      for (size_t i = 0; i < sel.size(); ++i)
        if (sp[i].projected) sells.at(sp[i].cell).push_back(i);
      for (size_t ci = 0; ci < sells.size(); ++ci) {
        std::vector<size_t>& sell = sells.at(cell_order[ci]);
        // reprojectorSeeds, :431-502: smallest variance first, the first seed that matches becomes a temporary point
        std::stable_sort(sell.begin(), sell.end(), [&](size_t a, size_t b) { return sel[a]->sigma2 < sel[b]->sigma2; });
        bool got = false;
        for (size_t i : sell) {
          if (!sm[i].success) continue;
          Seed& seed = *sel[i];
          ++n_seeds_; sum_seed_++;
          const Vector3d pHost = vscale(seed.ftr->f, 1. / seed.mu);
          Point* point = new Point(seed.ftr->frame->T_f_w_.inverse() * pHost, seed.ftr);
          point->idist_ = seed.mu;
          point->hostFeature_ = seed.ftr;
          point->type_ = Point::TYPE_TEMPORARY;
          point->ftr_type_ = seed.ftr->type == Feature::EDGELET ? Point::FEATURE_EDGELET
                             : seed.ftr->type == Feature::CORNER ? Point::FEATURE_CORNER : Point::FEATURE_GRADIENT;
          Feature* nf = new Feature();
          nf->frame = frame.get();
          nf->px = {sm[i].px_cur[0], sm[i].px_cur[1]};
          nf->f = frame->cam_->cam2world(nf->px);
          nf->level = sm[i].search_level;
          if (seed.ftr->type == Feature::EDGELET) {             // matcher_.ref_ftr_ is the seed's feature here
            nf->type = Feature::EDGELET;
            const double gx = sm[i].A_cur_ref[0] * seed.ftr->grad[0] + sm[i].A_cur_ref[1] * seed.ftr->grad[1];
            const double gy = sm[i].A_cur_ref[2] * seed.ftr->grad[0] + sm[i].A_cur_ref[3] * seed.ftr->grad[1];
            const double nn = std::sqrt(gx * gx + gy * gy);
            nf->grad = {gx / nn, gy / nn};
          } else
            nf->type = seed.ftr->type == Feature::GRADIENT ? Feature::GRADIENT : Feature::CORNER;
          nf->point = point;
          frame->addFeature(nf);
          seed.haveReprojected = true;
          seed.temp = point;
          point->seedStates_ = 0;
          map_.point_candidates_.addPauseSeedPoint(point);
          got = true;
          break;
        }
        if (got) ++n_matches_;
        if (n_matches_ >= (size_t)max_fts_) break;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------ local BA
void ba::LocalBundleAdjustment(Frame* center_kf, std::set<Frame*>* core_kfs, Map* map, size_t& n_incorrect_edges_1,
                               size_t& n_incorrect_edges_2, double& init_error, double& final_error)
{
  n_incorrect_edges_1 = 0; n_incorrect_edges_2 = 0;
  init_error = final_error = 0;
  // vertices: core keyframes first (src/bundle_adjustment.cpp:592-614).  The reference walks std::set<Frame*> /
  // std::set<Point*> in address order; frame and point ids give the same sets a run-independent order.
  std::vector<Frame*> core(core_kfs->begin(), core_kfs->end());
  std::sort(core.begin(), core.end(), [](const Frame* a, const Frame* b) { return a->id_ < b->id_; });
  std::vector<Frame*> vframes;
  std::vector<uint8_t> fixed;
  std::map<const Frame*, int> vidx;
  std::vector<Point*> mps;
  for (Frame* kf : core) {
    vidx[kf] = (int)vframes.size();
    vframes.push_back(kf);
    fixed.push_back((kf->id_ == Frame::id_base_ || kf->keyFrameId_ + 20 < center_kf->keyFrameId_) ? 1 : 0);   // id 0 = the sequence's first frame
    for (Feature* ft : kf->fts_)
      if (ft->point != nullptr) mps.push_back(ft->point);
  }
  std::sort(mps.begin(), mps.end(), [](const Point* a, const Point* b) { return a->id_ < b->id_; });
  mps.erase(std::unique(mps.begin(), mps.end()), mps.end());
  auto vertex_of = [&](Frame* f) {
    auto it = vidx.find(f);
    if (it != vidx.end()) return it->second;
    vidx[f] = (int)vframes.size();                               // host / neighbour keyframes outside the core: fixed (:702-737)
    vframes.push_back(f); fixed.push_back(1);
    return (int)vframes.size() - 1;
  };
  std::vector<hso_ba_edge> edges;
  std::vector<double> obs_uv, idist(mps.size());
  std::vector<Feature*> edge_ftr;
  for (size_t p = 0; p < mps.size(); ++p) {
    Point* pt = mps[p];
    idist[p] = pt->idist_;
    pt->nBA_++;
    Frame* host_frame = pt->hostFeature_->frame;
    const int vh = vertex_of(host_frame);
    for (Feature* ob : pt->obs_) {
      if (ob->frame->id_ == host_frame->id_) continue;
      hso_ba_edge e{};
      e.point = (int)p; e.host = vh; e.target = vertex_of(ob->frame);
      e.type = ob->type == Feature::EDGELET ? HSO_FTR_EDGELET : HSO_FTR_CORNER;
      e.level = ob->level;
      for (int k = 0; k < 3; ++k) e.fH[k] = pt->hostFeature_->f[k];
      const Vector2d uv = project2d(ob->f);
      if (ob->type == Feature::EDGELET) {
        e.normal[0] = ob->grad[0]; e.normal[1] = ob->grad[1];
        e.meas[0] = ob->grad[0] * uv[0] + ob->grad[1] * uv[1];
      } else {
        e.normal[0] = 1; e.normal[1] = 0;
        e.meas[0] = uv[0]; e.meas[1] = uv[1];
      }
      edges.push_back(e); edge_ftr.push_back(ob);
      obs_uv.push_back(uv[0]); obs_uv.push_back(uv[1]);
    }
  }
  if (edges.empty() || mps.empty()) return;
  std::vector<hso_se3> poses(vframes.size());
  for (size_t i = 0; i < vframes.size(); ++i) poses[i] = vframes[i]->T_f_w_.v;
  hso_gpu_ctx* ctx = center_kf->ctx_;
  const double fmean = center_kf->cam_->errorMultiplier2();
  float huber_corner = 0, huber_edge = 0;                       // :618-680
  api::ba_huber_deltas(ctx, poses.data(), (int)poses.size(), idist.data(), (int)idist.size(), edges.data(), obs_uv.data(),
                       (int)edges.size(), fmean, &huber_corner, &huber_edge);
  int n_iter = 100;                                             // :815-823
  if (map->size() > 5) n_iter = center_kf->fts_.size() < 100 ? Config::get().loba_num_iter + 10 : Config::get().loba_num_iter;
  std::vector<double> chi2(edges.size());
  hso_ba_result res{};
  api::ba_optimize(ctx, poses.data(), fixed.data(), (int)poses.size(), idist.data(), (int)idist.size(), edges.data(), (int)edges.size(),
                   huber_corner, huber_edge, n_iter, chi2.data(), &res);
  init_error = res.init_chi2; final_error = res.final_chi2;
  for (Frame* kf : core) {                                      // :826-834
    kf->T_f_w_.v = poses[vidx[kf]];
    map->point_candidates_.changeCandidatePosition(kf);
  }
  for (size_t p = 0; p < mps.size(); ++p) {                     // :843-851
    Point* pt = mps[p];
    pt->idist_ = idist[p];
    pt->pos_ = pt->hostFeature_->frame->T_f_w_.inverse() * vscale(pt->hostFeature_->f, 1.0 / pt->idist_);
  }
  const double t2 = 2.0 / fmean, t1 = 1.2 / fmean;             // :855-892
  for (int pass = 0; pass < 2; ++pass)
    for (size_t k = 0; k < edges.size(); ++k) {
      const bool edgelet = edges[k].type == HSO_FTR_EDGELET;
      if (edgelet != (pass == 1)) continue;
      Feature* ft = edge_ftr[k];
      if (ft->point == nullptr) continue;
      if (chi2[k] > (edgelet ? t1 * t1 : t2 * t2)) {
        if (ft->point->type_ == Point::TYPE_TEMPORARY) { ft->point->isBad_ = true; continue; }
        map->removePtFrameRef(ft->frame, ft);
        if (edgelet) ++n_incorrect_edges_2; else ++n_incorrect_edges_1;
      }
    }
  init_error = std::sqrt(init_error) * fmean;
  final_error = std::sqrt(final_error) * fmean;
}

// ------------------------------------------------------------------------------------------------ FrameHandlerMono
static void candidate_cb(void* user, Point* point, double sigma2) { static_cast<MapPointCandidates*>(user)->newCandidatePoint(point, sigma2); }

FrameHandlerMono::FrameHandlerMono(hso_gpu_ctx* ctx, AbstractCamera* cam, bool)
    : reprojector_(cam, map_, Config::get().max_fts), ctx_(ctx), cam_(cam)
{
  // initialize(), src/frame_handler_mono.cpp:61-72
  feature_extractor_ = new FeatureExtractor(cam_->width(), cam_->height(), Config::get().grid_size, Config::get().n_pyr_levels, false,
                                            Config::get().max_fts);
  depth_filter_ = new SeedFilter(feature_extractor_, &candidate_cb, &map_.point_candidates_);
  reprojector_.depth_filter_ = depth_filter_;
}

FrameHandlerMono::~FrameHandlerMono()
{
  delete depth_filter_;
  delete feature_extractor_;
}

bool FrameHandlerMono::startFrameProcessingCommon(double)
{
  if (set_start_) { resetAll(); stage_ = STAGE_FIRST_FRAME; }
  if (stage_ == STAGE_PAUSED) return false;
  map_.emptyTrash();
  return true;
}

int FrameHandlerMono::finishFrameProcessingCommon(size_t, UpdateResult dropout, size_t num_observations)
{
  num_obs_last_ = num_observations;
  if (dropout == RESULT_FAILURE && (stage_ == STAGE_DEFAULT_FRAME || stage_ == STAGE_RELOCALIZING)) {
    stage_ = STAGE_RELOCALIZING;
    tracking_quality_ = TRACKING_INSUFFICIENT;
  } else if (dropout == RESULT_FAILURE)
    resetAll();
  if (set_reset_) resetAll();
  return 0;
}

void FrameHandlerMono::resetAll()
{
  map_.reset();                                                // resetCommon, src/frame_handler_base.cpp:154-163
  stage_ = STAGE_PAUSED;
  set_reset_ = false; set_start_ = false;
  tracking_quality_ = TRACKING_INSUFFICIENT;
  num_obs_last_ = 0;
  last_frame_.reset(); new_frame_.reset(); firstFrame_.reset();
  core_kfs_.clear(); overlap_kfs_.clear();
  klt_homography_init_.reset();
  afterInit_ = false;
  depth_filter_->reset();
}

void FrameHandlerMono::setTrackingQuality(size_t num_observations)
{
  tracking_quality_ = TRACKING_GOOD;
  if (num_observations < (size_t)Config::get().quality_min_fts) tracking_quality_ = TRACKING_INSUFFICIENT;
  const int feature_drop = static_cast<int>(std::min(num_obs_last_, (size_t)Config::get().max_fts)) - (int)num_observations;
  if (feature_drop > Config::get().quality_max_drop_fts) tracking_quality_ = TRACKING_BAD;
}

void FrameHandlerMono::setFirstFrame(const FramePtr& first_frame)
{
  resetAll();
  last_frame_ = first_frame;
  last_frame_->setKeyframe();
  map_.addKeyframe(last_frame_);
  stage_ = STAGE_DEFAULT_FRAME;
}

namespace {
// developer probe (HSO_VO_TIMING=1): wall time per stage of processFrame, printed when the handler goes away
struct StageClock {
  bool on = getenv("HSO_VO_TIMING") != nullptr;
  double acc[6] = {0, 0, 0, 0, 0, 0}; long n = 0;
  std::chrono::steady_clock::time_point t;
  void start() { if (on) t = std::chrono::steady_clock::now(); }
  void lap(int k) { if (!on) return; const auto u = std::chrono::steady_clock::now(); acc[k] += std::chrono::duration<double, std::milli>(u - t).count(); t = u; }
  ~StageClock()
  {
    if (on && n) fprintf(stderr, "[hso_vo] %ld frames: frame build %.3f, track %.3f, reproject %.3f, pose %.3f, seeds (non-keyframes) %.3f, keyframe work %.3f ms per frame\n",
                         n, acc[5] / n, acc[0] / n, acc[1] / n, acc[2] / n, acc[3] / n, acc[4] / n);
  }
};
thread_local StageClock g_clock;
}  // namespace

void FrameHandlerMono::addImage(const uint8_t* img, int width, int height, double timestamp)
{
  if (!startFrameProcessingCommon(timestamp)) return;
  core_kfs_.clear();
  overlap_kfs_.clear();
  g_clock.start();
  new_frame_.reset(new Frame(ctx_, cam_, img, width, height, timestamp));
  g_clock.lap(5);
  new_frame_->keyFrameId_ = map_.size() == 0 ? 0 : map_.lastKeyframe()->keyFrameId_;
  UpdateResult res = RESULT_FAILURE;
  if (stage_ == STAGE_DEFAULT_FRAME) res = processFrame();
  else if (stage_ == STAGE_SECOND_FRAME) res = processSecondFrame();
  else if (stage_ == STAGE_FIRST_FRAME) res = processFirstFrame();
  else if (stage_ == STAGE_RELOCALIZING) res = relocalizeFrame(SE3(), map_.getClosestKeyframe(last_frame_));
  if (last_frame_ && last_frame_ != new_frame_) last_frame_->m_last_frame.reset();   // do not chain every frame ever seen
  last_frame_ = new_frame_;
  new_frame_.reset();
  last_result_ = res;
  finishFrameProcessingCommon((size_t)last_frame_->id_, res, last_frame_->m_n_inliers);
}

FrameHandlerMono::UpdateResult FrameHandlerMono::processFirstFrame()      // src/frame_handler_mono.cpp:125-151
{
  new_frame_->T_f_w_ = SE3();
  klt_homography_init_.poseoptim_thresh = Config::get().poseoptim_thresh;
  if (klt_homography_init_.addFirstFrame(new_frame_) == initialization::FAILURE) return RESULT_NO_KEYFRAME;
  new_frame_->setKeyframe();
  map_.addKeyframe(new_frame_);
  stage_ = STAGE_SECOND_FRAME;
  firstFrame_ = new_frame_;
  firstFrame_->m_exposure_time = 1.0;
  return RESULT_IS_KEYFRAME;
}

FrameHandlerMono::UpdateResult FrameHandlerMono::processSecondFrame()     // :154-172
{
  const initialization::InitResult res = klt_homography_init_.addSecondFrame(new_frame_);
  if (res == initialization::FAILURE) return RESULT_FAILURE;
  if (res == initialization::NO_KEYFRAME) return RESULT_NO_KEYFRAME;
  stage_ = STAGE_DEFAULT_FRAME;
  klt_homography_init_.reset();
  afterInit_ = true;
  firstFrame_->setKeyPoints();
  return RESULT_IS_KEYFRAME;
}

FrameHandlerMono::UpdateResult FrameHandlerMono::processFrame()
{
  const Config& cfg = Config::get();
  log_ = FrameLog();
  g_clock.n++; g_clock.start();
  new_frame_->T_f_w_ = motionModel_ * last_frame_->T_f_w_;      // :176
  if (afterInit_) last_frame_ = firstFrame_;
  new_frame_->m_last_frame = last_frame_;
  {                                                             // :184-209
    const bool inverse = !(new_frame_->gradMean_ > last_frame_->gradMean_ + 0.5);
    CoarseTracker Tracker(inverse, cfg.klt_max_level, cfg.klt_min_level + 1, 50, false);
    log_.img_align_n_tracked = Tracker.run(last_frame_, new_frame_);
    log_.used_inverse = inverse ? 1 : 0;
  }
  g_clock.lap(0);
  reprojector_.reprojectMap(new_frame_, overlap_kfs_);          // :217
  g_clock.lap(1);
  const size_t repr_n_new_references = reprojector_.n_matches_;
  log_.repr_n_matches = repr_n_new_references; log_.repr_n_mps = reprojector_.n_trials_; log_.repr_n_seeds = reprojector_.n_seeds_;
  if (repr_n_new_references < (size_t)cfg.quality_min_fts) {
    new_frame_->T_f_w_ = last_frame_->T_f_w_;
    tracking_quality_ = TRACKING_INSUFFICIENT;
    return RESULT_FAILURE;
  }
  size_t sfba_n_edges_final = 0;                                // :236-253
  double sfba_thresh = 0, sfba_error_init = 0, sfba_error_final = 0;
  pose_optimizer::optimizeLevenbergMarquardt3rd(cfg.poseoptim_thresh, 12, false, new_frame_, sfba_thresh, sfba_error_init,
                                                sfba_error_final, sfba_n_edges_final);
  g_clock.lap(2);
  new_frame_->m_n_inliers = sfba_n_edges_final;
  log_.sfba_n_edges_final = sfba_n_edges_final; log_.sfba_thresh = sfba_thresh; log_.sfba_error_init = sfba_error_init;
  log_.sfba_error_final = sfba_error_final;
  if (sfba_n_edges_final < (size_t)cfg.quality_min_fts) return RESULT_FAILURE;

  core_kfs_.insert(new_frame_);                                 // :258-266
  setTrackingQuality(sfba_n_edges_final);
  if (tracking_quality_ == TRACKING_INSUFFICIENT) {
    new_frame_->T_f_w_ = last_frame_->T_f_w_;
    return RESULT_FAILURE;
  }
  double depth_mean = 0, depth_min = 0, distance_mean = 0;
  frame_utils::getSceneDepth(*new_frame_, depth_mean, depth_min);
  frame_utils::getSceneDistance(*new_frame_, distance_mean);

  if (!needNewKf(distance_mean, sfba_n_edges_final) && !afterInit_) {   // :274-291
    createCovisibilityGraph(new_frame_, cfg.core_n_kfs, false);
    depth_filter_->addFrame(new_frame_);
    g_clock.lap(3);
    regular_counter_++;
    motionModel_ = new_frame_->T_f_w_ * last_frame_->T_f_w_.inverse();
    log_.n_seeds = depth_filter_->seeds_.size(); log_.n_candidates = map_.point_candidates_.candidates_.size();
    return RESULT_NO_KEYFRAME;
  }
  if (afterInit_) afterInit_ = false;
  regular_counter_ = 0;
  new_frame_->setKeyframe();
  for (Feature* ft : new_frame_->fts_)                          // :303-305
    if (ft->point != nullptr) ft->point->addFrameRef(ft);
  map_.point_candidates_.addCandidatePointToFrame(new_frame_);
  createCovisibilityGraph(new_frame_, cfg.core_n_kfs, true);
  if (cfg.loba_num_iter > 0)                                    // :314-327
    ba::LocalBundleAdjustment(new_frame_.get(), &LocalMap_, &map_, log_.loba_n_erredges_init, log_.loba_n_erredges_fin, log_.loba_err_init,
                              log_.loba_err_fin);
  for (auto& kf : overlap_kfs_) kf.first->setKeyPoints();
  if (sfba_n_edges_final <= 70) depth_filter_->addKeyframe(new_frame_, distance_mean, 0.5 * depth_min, 100);   // :335-338
  else depth_filter_->addKeyframe(new_frame_, distance_mean, 0.5 * depth_min, 200);
  map_.addKeyframe(new_frame_);
  g_clock.lap(4);
  motionModel_ = new_frame_->T_f_w_ * last_frame_->T_f_w_.inverse();
  log_.n_seeds = depth_filter_->seeds_.size(); log_.n_candidates = map_.point_candidates_.candidates_.size();
  return RESULT_IS_KEYFRAME;
}

FrameHandlerMono::UpdateResult FrameHandlerMono::relocalizeFrame(const SE3&, FramePtr ref_keyframe)
{
  if (ref_keyframe == nullptr) return RESULT_FAILURE;
  const Config& cfg = Config::get();
  CoarseTracker Tracker(true, cfg.klt_max_level, cfg.klt_min_level, 15, false);   // :366-367
  const size_t img_align_n_tracked = Tracker.run(ref_keyframe, last_frame_);
  if (img_align_n_tracked > 30) {
    const SE3 T_f_w_last = last_frame_->T_f_w_;
    last_frame_ = ref_keyframe;
    const UpdateResult res = processFrame();
    if (res != RESULT_FAILURE) stage_ = STAGE_DEFAULT_FRAME;
    else new_frame_->T_f_w_ = T_f_w_last;
    return res;
  }
  return RESULT_FAILURE;
}

bool FrameHandlerMono::needNewKf(const double&, const size_t&)
{
  if (regular_counter_ < 3) return false;
  const size_t n_mean_converge_frame = depth_filter_->nMeanConvergeFrame_;
  if (regular_counter_ < std::min(3, int(n_mean_converge_frame * 0.8))) return false;
  const FramePtr last_kf = map_.lastKeyframe();
  const SE3 T_c_r_full(new_frame_->T_f_w_ * last_kf->T_f_w_.inverse());
  SE3 T_c_r_nR;
  T_c_r_nR.v.t[0] = T_c_r_full.v.t[0]; T_c_r_nR.v.t[1] = T_c_r_full.v.t[1]; T_c_r_nR.v.t[2] = T_c_r_full.v.t[2];
  float optical_flow_full = 0, optical_flow_nR = 0;
  size_t optical_flow_num = 0;
  for (Feature* ft_kf : last_kf->fts_) {
    if (ft_kf->point == nullptr) continue;
    const Vector3d p_ref = vscale(ft_kf->f, vnorm(vsub(ft_kf->point->pos_, last_kf->pos())));
    const Vector2d uv_cur_full = new_frame_->cam_->world2cam(T_c_r_full * p_ref);
    const Vector2d uv_cur_nR = new_frame_->cam_->world2cam(T_c_r_nR * p_ref);
    const double dfx = uv_cur_full[0] - ft_kf->px[0], dfy = uv_cur_full[1] - ft_kf->px[1];
    const double dnx = uv_cur_nR[0] - ft_kf->px[0], dny = uv_cur_nR[1] - ft_kf->px[1];
    optical_flow_full += dfx * dfx + dfy * dfy;
    optical_flow_nR += dnx * dnx + dny * dny;
    optical_flow_num++;
  }
  optical_flow_full /= optical_flow_num; if (optical_flow_full < 133) return false;
  optical_flow_full = sqrtf(optical_flow_full);
  optical_flow_nR /= optical_flow_num;
  optical_flow_nR = sqrtf(optical_flow_nR);
  const int defult_resolution = 752 + 480;
  const float setting_maxShiftWeightT = 0.04 * defult_resolution;
  const float setting_maxShiftWeightRT = 0.02 * defult_resolution;
  const float setting_kfGlobalWeight = 0.75;
  const int wh = new_frame_->cam_->width() + new_frame_->cam_->height();
  const float DSO_judgement = setting_kfGlobalWeight * setting_maxShiftWeightT * optical_flow_nR / wh +
                              setting_kfGlobalWeight * setting_maxShiftWeightRT * optical_flow_full / wh;
  return DSO_judgement > 1;
}

void FrameHandlerMono::createCovisibilityGraph(FramePtr currentFrame, size_t n_closest, bool is_keyframe)
{
  // the reference counts in a std::map<Frame*, int> (address order); frame ids order the same entries reproducibly
  std::map<int, std::pair<Frame*, int>> KFcounter;
  int n_linliers = 0;
  for (Feature* ft : currentFrame->fts_) {
    if (ft->point == nullptr) continue;
    n_linliers++;
    for (Feature* ob : ft->point->obs_) {
      if (ob->frame->id_ == currentFrame->id_) continue;
      auto& e = KFcounter[ob->frame->id_];
      e.first = ob->frame; e.second++;
    }
  }
  if (KFcounter.empty()) return;
  int nmax = 0;
  Frame* pKFmax = nullptr;
  const int th = n_linliers > 30 ? 5 : 3;
  std::vector<std::pair<int, Frame*>> vPairs;
  vPairs.reserve(KFcounter.size());
  for (auto& kv : KFcounter) {
    Frame* f = kv.second.first; const int cnt = kv.second.second;
    if (cnt > nmax) { nmax = cnt; pKFmax = f; }
    if (cnt >= th) vPairs.push_back(std::make_pair(cnt, f));
    // :607-612 releases the Sobel images of keyframes five generations back; the resident frames keep theirs (the
    // matcher's checkNormal would read freed memory there)
  }
  if (vPairs.empty()) vPairs.push_back(std::make_pair(nmax, pKFmax));
  std::sort(vPairs.begin(), vPairs.end(), [](const std::pair<int, Frame*>& l, const std::pair<int, Frame*>& r) {
    if (l.first != r.first) return l.first > r.first;
    return l.second->id_ < r.second->id_;
  });
  const size_t nCovisibility = 5;
  const size_t k = std::min(nCovisibility, vPairs.size());
  for (size_t i = 0; i < k; ++i) currentFrame->connectedKeyFrames.push_back(vPairs[i].second);
  if (is_keyframe) {
    LocalMap_.clear();
    const size_t n = std::min(n_closest, vPairs.size());
    for (size_t i = 0; i < n; ++i) LocalMap_.insert(vPairs[i].second);
    FramePtr LastKF = map_.lastKeyframe();
    if (LocalMap_.find(LastKF.get()) == LocalMap_.end()) LocalMap_.insert(LastKF.get());
    LocalMap_.insert(currentFrame.get());
  }
}

}  // namespace hso

// ------------------------------------------------------------------------------------------------ C interface
#include "../../include/hso_vo.h"

struct hso_vo {
  hso_gpu_ctx* ctx = nullptr;
  hso::AbstractCamera* cam = nullptr;
  hso::FrameHandlerMono* vo = nullptr;
  std::string err;
  bool poisoned = false;   // a device call failed inside processFrame: the map may be half updated, the handle refuses further frames
  bool owns_ctx = true;    // false: one of the sequences of a multi-sequence driver (hso_multi.cpp), the context is shared
  int id_base = 0;         // Frame::id_base_ of the sequence's thread: reported frame ids are relative to it
  hso::Config cfg;         // this handle's configuration (Config::get() inside its calls)
  // this handle's values of the reference's static counters (Frame::frame_counter_, keyFrameCounter_, Point::point_counter_,
  // Seed::batch_counter): keyframe-id gaps gate decisions (reprojector.cpp: cur - kf < 4), so they must not advance with another
  // handle's keyframes
  int c_frame = 0, c_kf = 0, c_point = 0, c_batch = 0;
};
namespace {
struct ConfigScope {       // the handle's configuration and counters are the thread's current ones while one of its calls runs
  hso_vo* v;
  hso::Config* prev;
  int t_frame, t_kf, t_point, t_batch, t_base;
  explicit ConfigScope(hso_vo* v_) : v(v_), prev(hso::Config::current_)
  {
    hso::Config::current_ = &v->cfg;
    t_frame = hso::Frame::frame_counter_; t_kf = hso::Frame::keyFrameCounter_; t_point = hso::Point::point_counter_; t_batch = hso::Seed::batch_counter;
    t_base = hso::Frame::id_base_;
    hso::Frame::frame_counter_ = v->c_frame; hso::Frame::keyFrameCounter_ = v->c_kf; hso::Point::point_counter_ = v->c_point;
    hso::Seed::batch_counter = v->c_batch; hso::Frame::id_base_ = v->id_base;
  }
  ~ConfigScope()
  {
    v->c_frame = hso::Frame::frame_counter_; v->c_kf = hso::Frame::keyFrameCounter_; v->c_point = hso::Point::point_counter_; v->c_batch = hso::Seed::batch_counter;
    hso::Frame::frame_counter_ = t_frame; hso::Frame::keyFrameCounter_ = t_kf; hso::Point::point_counter_ = t_point; hso::Seed::batch_counter = t_batch;
    hso::Frame::id_base_ = t_base;
    hso::Config::current_ = prev;
  }
};
}  // namespace

template <typename F> static int vo_guard(hso_vo* v, F f)
{
  if (!v) return HSO_E_INVALID;
  if (v->poisoned) {
    if (v->err.find("handle unusable") == std::string::npos) v->err = "handle unusable after a device error (" + v->err + "): destroy it and create a new one";
    return HSO_E_HIP;
  }
  ConfigScope scope(v);
  try { f(); return HSO_OK; }
  catch (const hso::api::DeviceError& e) { v->err = e.what(); v->poisoned = true; return HSO_E_HIP; }
  catch (const std::exception& e) { v->err = e.what(); return HSO_E_INVALID; }
}

extern "C" {

int hso_vo_create(hso_vo** out, const hso_camera* cam, int max_fts, int device)
{
  if (!out || !cam || max_fts <= 0) return HSO_E_INVALID;
  *out = nullptr;
  hso_gpu_ctx* ctx = nullptr;
  const int rc = hso_gpu_create(&ctx, device, nullptr);
  if (rc < 0) return rc;
  hso_vo* v = new hso_vo();               // its counters start at 0 like the reference's statics in a fresh process
  v->cfg.max_fts = max_fts;               // Config::maxFts() is read when the reprojector and the extractor are built
  ConfigScope scope(v);
  v->ctx = ctx;
  v->cam = new hso::AbstractCamera(*cam);
  v->vo = new hso::FrameHandlerMono(ctx, v->cam, false);
  *out = v;
  return HSO_OK;
}

void hso_vo_destroy(hso_vo* v)
{
  if (!v) return;
  { ConfigScope scope(v); delete v->vo; v->vo = nullptr; }
  delete v->cam;
  hso_gpu_destroy(v->ctx);
  delete v;
}

}  // extern "C"

// One sequence of the multi-sequence driver (hso_multi.cpp): called on the sequence's own worker thread, whose thread-local
// counters and router are already set; the context is shared and Config::maxFts() was set by the caller before the threads started.
hso_vo* hso_vo_create_shared(hso_gpu_ctx* ctx, const hso_camera* cam, int max_fts)
{
  hso_vo* v = new hso_vo();
  v->cfg.max_fts = max_fts;
  v->id_base = hso::Frame::id_base_; v->c_frame = hso::Frame::frame_counter_; v->c_kf = hso::Frame::keyFrameCounter_;
  v->c_point = hso::Point::point_counter_; v->c_batch = hso::Seed::batch_counter;       // the worker thread set them for this sequence
  ConfigScope scope(v);
  v->ctx = ctx; v->owns_ctx = false;
  v->cam = new hso::AbstractCamera(*cam);
  v->vo = new hso::FrameHandlerMono(ctx, v->cam, false);
  return v;
}
void hso_vo_destroy_shared(hso_vo* v)
{
  if (!v) return;
  { ConfigScope scope(v); delete v->vo; v->vo = nullptr; }      // releases the sequence's resident frames through the router
  delete v->cam;
  delete v;
}

extern "C" {

const char* hso_vo_last_error(const hso_vo* v) { return v ? v->err.c_str() : "null handle"; }

int hso_vo_trace(hso_vo* v, const char* path)
{
  if (!v) return HSO_E_INVALID;
  if (!path) { hso::api::trace().close(); return HSO_OK; }
  return hso::api::trace().open(path) ? HSO_OK : HSO_E_INVALID;
}

int hso_vo_set_first_frame(hso_vo* v, const uint8_t* img, int width, int height, double timestamp, const float* depth_z,
                           const hso_se3* T_f_w)
{
  if (!v || !img || !depth_z) return HSO_E_INVALID;
  return vo_guard(v, [&]() {
    using namespace hso;
    FramePtr frame(new Frame(v->ctx, v->cam, img, width, height, timestamp));
    if (T_f_w) frame->T_f_w_.v = *T_f_w;
    frame->m_exposure_time = 1.0;                     // processFirstFrame, src/frame_handler_mono.cpp:138
    // the detector of the initialisation (initialization.cpp: FeatureExtractor with isInit), then one point per
    // feature whose depth is known — what the two-view initialisation leaves behind for its inliers
    FeatureExtractor fe(width, height, Config::get().grid_size, Config::get().n_pyr_levels, true, Config::get().max_fts);
    Features fts;
    fe.detect(frame.get(), 20, frame->gradMean_, fts, nullptr);
    for (Feature* ft : fts) {
      const int x = (int)ft->px[0], y = (int)ft->px[1];
      const float z = (x >= 0 && y >= 0 && x < width && y < height) ? depth_z[(size_t)y * width + x] : 0.f;
      if (!(z > 0)) { delete ft; continue; }
      const double dist = (double)z / ft->f[2];       // the point on the bearing with optical-axis depth z
      const Vector3d X = {ft->f[0] * dist, ft->f[1] * dist, ft->f[2] * dist};
      Point* pt = new Point(frame->T_f_w_.inverse() * X, ft);
      pt->idist_ = 1.0 / dist;
      pt->hostFeature_ = ft;
      pt->ftr_type_ = ft->type == Feature::EDGELET ? Point::FEATURE_EDGELET : ft->type == Feature::CORNER ? Point::FEATURE_CORNER : Point::FEATURE_GRADIENT;
      ft->point = pt;
      frame->addFeature(ft);
    }
    if (frame->fts_.size() < 10) throw std::runtime_error("set_first_frame: fewer than 10 features with a depth");
    v->vo->setFirstFrame(frame);
    double depth_mean = 0, depth_min = 0, distance_mean = 0;
    frame_utils::getSceneDepth(*frame, depth_mean, depth_min);
    frame_utils::getSceneDistance(*frame, distance_mean);
    v->vo->depth_filter_->addKeyframe(frame, distance_mean, 0.5 * depth_min, 200);
  });
}

int hso_vo_init_compute_matrix(const double* f_ref, const double* f_cur, int n, double focal_length, double reproj_thresh, hso_se3* T_cur_from_ref,
                            int32_t* inliers, int cap, double* xyz_in_cur, int32_t* used_homography)
{
  if (!f_ref || !f_cur || n < 0 || !T_cur_from_ref) return HSO_E_INVALID;
  try {
    std::vector<hso::Vector3d> a(n), b(n), xyz;
    for (int i = 0; i < n; i++) { a[i] = {f_ref[3 * i], f_ref[3 * i + 1], f_ref[3 * i + 2]}; b[i] = {f_cur[3 * i], f_cur[3 * i + 1], f_cur[3 * i + 2]}; }
    std::vector<int> in;
    hso::SE3 T;
    int used = 0;
    hso::initialization::computeInitializeMatrix(a, b, focal_length, reproj_thresh, in, xyz, T, &used);
    *T_cur_from_ref = T.v;
    if (used_homography) *used_homography = used;
    for (size_t i = 0; i < in.size() && (int)i < cap && inliers; i++) inliers[i] = in[i];
    for (size_t i = 0; i < xyz.size() && xyz_in_cur; i++) { xyz_in_cur[3 * i] = xyz[i][0]; xyz_in_cur[3 * i + 1] = xyz[i][1]; xyz_in_cur[3 * i + 2] = xyz[i][2]; }
    return (int)in.size();
  } catch (const std::exception&) { return HSO_E_INVALID; }
}

int hso_vo_start(hso_vo* v)
{
  if (!v) return HSO_E_INVALID;
  return vo_guard(v, [&]() { v->vo->start(); });
}

int hso_vo_add_image(hso_vo* v, const uint8_t* img, int width, int height, double timestamp)
{
  if (!v || !img) return HSO_E_INVALID;
  return vo_guard(v, [&]() { v->vo->addImage(img, width, height, timestamp); hso::api::trace().flush(); });
}

int hso_vo_get_status(hso_vo* v, hso_vo_status* st)
{
  if (!v || !st) return HSO_E_INVALID;
  return vo_guard(v, [&]() {
    memset(st, 0, sizeof(*st));
    hso::FramePtr f = v->vo->lastFrame();
    if (!f) return;
    const auto& L = v->vo->log_;
    st->T_f_w = f->T_f_w_.v;
    st->timestamp = f->timestamp_;
    st->exposure_time = f->m_exposure_time;
    st->frame_id = f->id_ - v->id_base; st->keyframe_id = f->keyFrameId_;
    st->is_keyframe = f->isKeyframe() ? 1 : 0;
    st->stage = (int)v->vo->stage(); st->tracking_quality = (int)v->vo->trackingQuality(); st->result = (int)v->vo->lastResult();
    st->n_features = (int)f->fts_.size(); st->n_inliers = (int)f->m_n_inliers;
    st->n_tracked = (int)L.img_align_n_tracked; st->n_matches = (int)L.repr_n_matches; st->n_trials = (int)L.repr_n_mps;
    st->n_seed_matches = (int)L.repr_n_seeds; st->n_seeds = (int)L.n_seeds; st->n_candidates = (int)L.n_candidates;
    st->n_keyframes = (int)v->vo->map_.size(); st->used_inverse = L.used_inverse;
    st->pose_error_init = L.sfba_error_init; st->pose_error_final = L.sfba_error_final;
    st->ba_error_init = L.loba_err_init; st->ba_error_final = L.loba_err_fin;
    st->ba_removed_1 = (int)L.loba_n_erredges_init; st->ba_removed_2 = (int)L.loba_n_erredges_fin;
  });
}

int hso_vo_get_keyframes(hso_vo* v, double* timestamps, hso_se3* T_f_w, int32_t* frame_ids, int cap)
{
  if (!v) return HSO_E_INVALID;
  int n = 0;
  for (const hso::FramePtr& kf : v->vo->map_.keyframes_) {
    if (n < cap) {
      if (timestamps) timestamps[n] = kf->timestamp_;
      if (T_f_w) T_f_w[n] = kf->T_f_w_.v;
      if (frame_ids) frame_ids[n] = kf->id_ - v->id_base;
    }
    ++n;
  }
  return n;
}

}  // extern "C"
