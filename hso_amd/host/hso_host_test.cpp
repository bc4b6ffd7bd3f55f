// hso_host_test.cpp — drives the host mirror the way FrameHandlerMono::processFrame does
// (src/frame_handler_mono.cpp:173-209): two Frames, features with points hosted in the
// reference frame, CoarseTracker(...).run(last, new), then Matcher::findMatchDirect on projected
// points and one DepthFilter observation of fresh seeds.  Reads a binary case written by
// tests/test_host_mirror_gpu.py and prints the results as one line of numbers.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "hso_host.h"

int main(int argc, char** argv)
{
  if (argc < 2) { std::fprintf(stderr, "usage: hso_host_test case.bin\n"); return 2; }
  FILE* f = std::fopen(argv[1], "rb");
  if (!f) { std::perror("open"); return 2; }
  int32_t hdr[4];
  double camp[4];
  if (std::fread(hdr, 4, 4, f) != 4 || std::fread(camp, 8, 4, f) != 4) return 2;
  const int w = hdr[0], h = hdr[1], n = hdr[2], inverse = hdr[3];
  std::vector<uint8_t> ref((size_t)w * h), cur((size_t)w * h);
  std::vector<double> tab((size_t)n * 6);  // px, py, fx, fy, fz, idist (idist <= 0: no point)
  if (std::fread(ref.data(), 1, ref.size(), f) != ref.size() || std::fread(cur.data(), 1, cur.size(), f) != cur.size() ||
      std::fread(tab.data(), 8, tab.size(), f) != tab.size())
    return 2;
  std::fclose(f);

  hso_gpu_ctx* ctx = nullptr;
  if (hso_gpu_create(&ctx, 0, nullptr) < 0) { std::fprintf(stderr, "no GPU context\n"); return 3; }
  {
    hso_camera c{};
    c.model = HSO_CAM_PINHOLE; c.width = w; c.height = h;
    c.fx = camp[0]; c.fy = camp[1]; c.cx = camp[2]; c.cy = camp[3];
    hso::AbstractCamera cam(c);
    // wrong size must throw like src/frame.cpp:85-86
    bool threw = false;
    try { hso::Frame bad(ctx, &cam, ref.data(), w - 16, h, 0.0); } catch (const std::runtime_error&) { threw = true; }
    hso::FramePtr last(new hso::Frame(ctx, &cam, ref.data(), w, h, 0.0));
    hso::FramePtr next(new hso::Frame(ctx, &cam, cur.data(), w, h, 0.05));
    last->m_exposure_time = 1.0;
    std::vector<hso::Point*> points;
    for (int i = 0; i < n; i++) {
      hso::Feature* ft = new hso::Feature();
      ft->frame = last.get();
      ft->px = {tab[6 * i + 0], tab[6 * i + 1]};
      ft->f = {tab[6 * i + 2], tab[6 * i + 3], tab[6 * i + 4]};
      if (tab[6 * i + 5] > 0) {
        hso::Point* pt = new hso::Point();
        pt->idist_ = tab[6 * i + 5];
        pt->hostFeature_ = ft;
        ft->point = pt;
        points.push_back(pt);
      }
      last->fts_.push_back(ft);
    }
    next->T_f_w_ = last->T_f_w_;  // motion model = identity (frame_handler_mono.cpp:176)
    // frame_handler_mono.cpp:184-204: forward when the new frame has clearly more gradient
    hso::CoarseTracker tracker(inverse != 0, 4, 1, 50, false);
    const size_t n_tracked = tracker.run(last, next);
    const hso_se3& T = next->T_f_w_.v;
    std::printf("%d %zu %.17g %.17g %.17g %.17g %.17g %.17g %.17g %.9g %.17g", threw ? 1 : 0, n_tracked, T.q[0], T.q[1], T.q[2],
                T.q[3], T.t[0], T.t[1], T.t[2], tracker.m_last.exposure_rat, next->m_exposure_time);
    for (int l = 0; l < 5; l++) std::printf(" %d", tracker.m_last.iters[l]);
    std::printf(" %.9g %.9g\n", last->integralImage_, next->integralImage_);

    // ---- reprojection matching the way Reprojector::reprojectCell drives Matcher
    // (src/reprojector.cpp:352-429): project the point with the tracked pose, refine.
    const int K = (int)std::min<size_t>(points.size(), 96);
    std::printf("%d", K);
    for (int i = 0; i < K; i++) {
      hso::Point* pt = points[i];
      const hso::Feature* hf = pt->hostFeature_;
      const double inv = 1.0 / pt->idist_;
      pt->pos_ = last->T_f_w_.inverse() * hso::Vector3d{hf->f[0] * inv, hf->f[1] * inv, hf->f[2] * inv};
      pt->obs_.push_back(pt->hostFeature_);
      hso::Vector2d px = cam.world2cam(next->T_f_w_ * pt->pos_);
      hso::Matcher matcher;
      const bool ok = matcher.findMatchDirect(*pt, *next, px);
      std::printf(" %d %.17g %.17g %d", ok ? 1 : 0, px[0], px[1], matcher.search_level_);
      if (ok) {  // the new observation, as Reprojector::reprojectCell creates it (src/reprojector.cpp:385-412)
        hso::Feature* nf = new hso::Feature();
        nf->frame = next.get(); nf->px = px; nf->f = cam.cam2world(px); nf->level = matcher.search_level_; nf->point = pt;
        next->fts_.push_back(nf);
      }
    }
    std::printf("\n");

    // ---- depth filter: seeds on the next 32 features, one observation in the new frame
    // (src/depth_filter.cpp:557-675); px_error_angle as DepthFilter's constructor derives it (:360-366)
    hso::DepthFilter df(std::atan(1.0 / (2.0 * cam.errorMultiplier2())) * 2.0);
    for (int i = K; i < (int)std::min<size_t>(points.size(), 2 * (size_t)K); i++) {
      hso::Feature* hf = points[i]->hostFeature_;
      df.seeds_.emplace_back(hf, (float)(1.1 / points[i]->idist_), (float)(0.5 / points[i]->idist_));
    }
    const size_t n_seed_ok = df.observeDepth(next);
    std::printf("%zu %zu", n_seed_ok, df.seeds_.size());
    for (const hso::Seed& sd : df.seeds_) std::printf(" %.9g %.9g %.9g", sd.mu, sd.sigma2, sd.b);
    std::printf("\n");

    // ---- motion-only pose refinement from a perturbed start (frame_handler_mono.cpp:241-243)
    const hso::SE3 T_tracked = next->T_f_w_;
    next->T_f_w_.v.t[0] += 0.004; next->T_f_w_.v.t[1] -= 0.003;
    double scale = 0, e0 = 0, e1 = 0; size_t nobs = 0;
    hso::pose_optimizer::optimizeLevenbergMarquardt3rd(2.0, 12, false, next, scale, e0, e1, nobs);
    size_t culled = 0;
    for (hso::Feature* ft : next->fts_) culled += ft->point == nullptr;
    const hso_se3& P = next->T_f_w_.v;
    std::printf("%zu %zu %zu %.17g %.17g %.17g %.17g %.17g %.17g %.17g %.12g %.12g %.12g %.9g %.17g %.17g\n", next->fts_.size(), nobs,
                culled, P.q[0], P.q[1], P.q[2], P.q[3], P.t[0], P.t[1], P.t[2], scale, e0, e1, next->m_error_in_px, T_tracked.v.t[0],
                next->Cov_[0]);

    // ---- the frame becomes a keyframe: new features away from the existing ones, one seed each
    // (DepthFilter::addKeyframe -> initializeSeeds, src/depth_filter.cpp:146-205)
    hso::FeatureExtractor extractor(w, h, 25, 3, false, 200);
    hso::DepthFilter kf_filter(df.px_error_angle_);
    kf_filter.featureExtractor_ = &extractor;
    kf_filter.addKeyframe(next, 2.0, 0.5);
    std::printf("%zu %.9g", kf_filter.seeds_.size(), next->gradMean_);
    for (const hso::Seed& sd : kf_filter.seeds_) {
      const hso::Feature* ft = sd.ftr;
      std::printf(" %d %d %.9g %.9g %.17g %.17g %.17g %.17g %.17g %.9g %.9g", (int)ft->type, ft->level, ft->px[0], ft->px[1], ft->grad[0],
                  ft->grad[1], ft->f[0], ft->f[1], ft->f[2], sd.mu, sd.sigma2);
    }
    std::printf("\n");
    for (const hso::Seed& sd : kf_filter.seeds_) delete sd.ftr;

    // ---- the initialisation-time extractor on the same frame: fastDetectMT + fillingHole, 2000 features
    // (FeatureExtractor(..., isInit = true), src/feature_detection.cpp:382-384, 439-442)
    hso::FeatureExtractor init_extractor(w, h, 25, 3, true, 200);
    hso::Features init_fts;
    init_extractor.detect(next.get(), 20, next->gradMean_, init_fts);
    std::printf("%zu", init_fts.size());
    for (const hso::Feature* ft : init_fts) std::printf(" %d %d %.9g %.9g", (int)ft->type, ft->level, ft->px[0], ft->px[1]);
    std::printf("\n");
    for (hso::Feature* ft : init_fts) delete ft;

    // ---- Reprojector::reprojectMap of the keyframe's points into a fresh copy of the new frame
    // (src/reprojector.cpp:88-331): once through reprojectCellAll (few candidates), once through
    // the three cell passes (small feature budget)
    for (hso::Point* p : points) {
      const hso::Feature* hf = p->hostFeature_;
      const double inv = 1.0 / p->idist_;
      p->pos_ = last->T_f_w_.inverse() * hso::Vector3d{hf->f[0] * inv, hf->f[1] * inv, hf->f[2] * inv};
      if (p->obs_.empty()) p->obs_.push_back(p->hostFeature_);
    }
    for (int budget : {200, 40}) {
      for (hso::Point* p : points) { p->n_failed_reproj_ = 0; p->n_succeeded_reproj_ = 0; p->type_ = hso::Point::TYPE_UNKNOWN; }
      hso::FramePtr probe(new hso::Frame(ctx, &cam, cur.data(), w, h, 0.1));
      probe->T_f_w_ = T_tracked;
      probe->m_exposure_time = next->m_exposure_time;
      hso::Reprojector reprojector(&cam, budget);
      std::vector<std::pair<hso::FramePtr, size_t>> overlap;
      reprojector.reprojectMap(probe, {last}, overlap);
      std::printf("%zu %zu %zu %zu %d %d", reprojector.n_matches_, reprojector.n_trials_, reprojector.nFeatures_, overlap[0].second,
                  reprojector.cell_size, reprojector.grid_n_cols);
      for (const hso::Feature* ft : probe->fts_) {
        size_t idx = 0;
        while (points[idx] != ft->point) idx++;
        std::printf(" %zu %d %.17g %.17g", idx, ft->level, ft->px[0], ft->px[1]);
      }
      std::printf("\n");
    }
    for (hso::Point* p : points) delete p;
  }
  hso_gpu_destroy(ctx);
  return 0;
}
