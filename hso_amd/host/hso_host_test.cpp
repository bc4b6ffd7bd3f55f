// hso_host_test.cpp — drives the host mirror the way FrameHandlerMono::processFrame does
// (src/frame_handler_mono.cpp:173-209): two Frames, features with points hosted in the
// reference frame, CoarseTracker(...).run(last, new).  Reads a binary case written by
// tests/test_host_mirror_gpu.py and prints the results as one line of numbers.
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
    for (hso::Point* p : points) delete p;
  }
  hso_gpu_destroy(ctx);
  return 0;
}
