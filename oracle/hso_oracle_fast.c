/*
 * hso_oracle_fast.c — FAST-9 corners, their scores, 3x3 non-maximum suppression and the
 * Shi-Tomasi response: FeatureExtractor::fastDetect restated (src/feature_detection.cpp:518-587 (fastDetectST per level, fastDetect)).
 * TEST INFRASTRUCTURE (see hso_oracle.h).
 *
 * PINNED: unlike the rest of the oracle this part is checked against the reference's own code —
 * thirdparty/fast compiles standalone (oracle/_ref/libfast_ref.so, oracle/Makefile) and
 * tests/golden/fast9.json holds its outputs; tests/test_fast.py compares bit for bit.
 *
 * The library's detector (thirdparty/fast/src/faster_corner_9_sse.cpp, fast_9.cpp) is a
 * machine-generated decision tree / SSE2 mask cascade for the published FAST-9 segment test
 * (Rosten & Drummond): a pixel p is a corner at barrier b if 9 contiguous pixels of the 16-pixel
 * Bresenham circle of radius 3 are all > p + b or all < p - b; x in [3, w-3), y in [3, h-3), raster
 * order.  Its score (fast_9_score.cpp) walks the barrier upwards until the test fails and returns
 * the last barrier that passed, i.e. max over arcs of (min over the arc of |I - p|) - 1.  The
 * restatement evaluates both definitions directly.
 */
#include "hso_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* circle offsets in the library's order (fast_9_score.cpp:4661-4678): index 0 = (0,+3), clockwise */
static const int kCircle[16][2] = {
  {0, 3}, {1, 3}, {2, 2}, {3, 1}, {3, 0}, {3, -1}, {2, -2}, {1, -3},
  {0, -3}, {-1, -3}, {-2, -2}, {-3, -1}, {-3, 0}, {-3, 1}, {-2, 2}, {-1, 3}
};

/* largest barrier b >= 0 at which (x, y) is still a FAST-`arc` corner (arc contiguous circle pixels
 * all brighter or all darker), or -1 if it is not one even at b = 0 */
int hso_or_fast_max_barrier(const uint8_t* img, int stride, int x, int y, int arc)
{
  const int p = img[y * stride + x];
  int d[16];
  for (int k = 0; k < 16; k++) d[k] = (int)img[(y + kCircle[k][1]) * stride + x + kCircle[k][0]] - p;
  int best = 0;  /* max over arcs of the minimum |difference| along the arc (0: no arc with a common sign) */
  for (int s = 0; s < 16; s++) {
    int mb = 256, md = 256;
    for (int k = 0; k < arc; k++) {
      const int v = d[(s + k) & 15];
      if (v < mb) mb = v;       /* brighter arc: min (I - p) */
      if (-v < md) md = -v;     /* darker arc:   min (p - I) */
    }
    if (mb > best) best = mb;
    if (md > best) best = md;
  }
  return best - 1;  /* I > p + b for all of the arc  <=>  b <= min(I - p) - 1 */
}

int hso_or_fast9_max_barrier(const uint8_t* img, int stride, int x, int y) { return hso_or_fast_max_barrier(img, stride, x, y, 9); }

/* hso::shiTomasiScore, src/vikit/vision.cpp:111-151 */
float hso_or_shi_tomasi(const uint8_t* img, int cols, int rows, int u, int v)
{
  float dXX = 0.0, dYY = 0.0, dXY = 0.0;
  const int halfbox_size = 4, box_size = 2 * halfbox_size, box_area = box_size * box_size;
  const int x_min = u - halfbox_size, x_max = u + halfbox_size, y_min = v - halfbox_size, y_max = v + halfbox_size;
  if (x_min < 1 || x_max >= cols - 1 || y_min < 1 || y_max >= rows - 1) return 0.0;
  const int stride = cols;
  for (int y = y_min; y < y_max; ++y) {
    const uint8_t* ptr_left = img + stride * y + x_min - 1;
    const uint8_t* ptr_right = img + stride * y + x_min + 1;
    const uint8_t* ptr_top = img + stride * (y - 1) + x_min;
    const uint8_t* ptr_bottom = img + stride * (y + 1) + x_min;
    for (int x = 0; x < box_size; ++x, ++ptr_left, ++ptr_right, ++ptr_top, ++ptr_bottom) {
      const float dx = *ptr_right - *ptr_left;
      const float dy = *ptr_bottom - *ptr_top;
      dXX += dx * dx; dYY += dy * dy; dXY += dx * dy;
    }
  }
  dXX = dXX / (2.0 * box_area);
  dYY = dYY / (2.0 * box_area);
  dXY = dXY / (2.0 * box_area);
  /* sqrt(float) resolves to the float overload in the reference (vision.h pulls in <math.h> through
   * OpenCV's C headers, which brings std::sqrt's overloads into the global namespace) */
  return 0.5 * (dXX + dYY - sqrtf((dXX + dYY) * (dXX + dYY) - 4 * (dXX * dYY - dXY * dXY)));
}

/* fast_corner_detect_9_sse2 + fast_corner_score_9: all corners of one image in raster order.
 * xy / scores hold up to cap entries; returns the number of corners. */
int hso_or_fast_detect_arc(const uint8_t* img, int w, int h, int threshold, int arc, int16_t* xy, int32_t* scores, int cap)
{
  int n = 0;
  if (h < 7 || w < 7) return 0;
  for (int y = 3; y < h - 3; y++)
    for (int x = 3; x < w - 3; x++) {
      const int mb = hso_or_fast_max_barrier(img, w, x, y, arc);
      if (mb >= threshold) {
        if (n < cap) { xy[2 * n] = (int16_t)x; xy[2 * n + 1] = (int16_t)y; scores[n] = mb; }
        n++;
      }
    }
  return n;
}

int hso_or_fast9_detect(const uint8_t* img, int w, int h, int threshold, int16_t* xy, int32_t* scores, int cap)
{
  return hso_or_fast_detect_arc(img, w, h, threshold, 9, xy, scores, cap);
}

/* FeatureExtractor::fastDetect for one level (src/feature_detection.cpp:552-586): detection, score,
 * fast_nonmax_3x3 (thirdparty/fast/src/nonmax_3x3.cpp: a corner survives unless one of its eight
 * neighbours is a corner with score >= its own), border filter, Shi-Tomasi response.
 * Returns the number of surviving corners (only the first cap are written). */
static int detect_level_arc(const uint8_t* img, int w, int h, int threshold, int border, int arc, hso_corner* out, int cap)
{
  int16_t* sc = (int16_t*)malloc(sizeof(int16_t) * (size_t)w * h);  /* score per pixel, -1 = not a corner */
  for (int i = 0; i < w * h; i++) sc[i] = -1;
  if (h >= 7 && w >= 7)
    for (int y = 3; y < h - 3; y++)
      for (int x = 3; x < w - 3; x++) {
        const int mb = hso_or_fast_max_barrier(img, w, x, y, arc);
        if (mb >= threshold) sc[y * w + x] = (int16_t)mb;
      }
  int n = 0;
  for (int y = 3; y < h - 3; y++)
    for (int x = 3; x < w - 3; x++) {
      const int s = sc[y * w + x];
      if (s < 0) continue;
      int keep = 1;
      for (int dy = -1; dy <= 1 && keep; dy++)
        for (int dx = -1; dx <= 1; dx++) {
          if (!dx && !dy) continue;
          if (sc[(y + dy) * w + x + dx] >= s) { keep = 0; break; }  /* neighbours inside [2, w-2) exist: x >= 3 */
        }
      if (!keep) continue;
      if (x < border || x > w - border || y < border || y > h - border) continue;  /* :573 */
      if (n < cap) {
        out[n].x = (int16_t)x; out[n].y = (int16_t)y; out[n].score = s;
        out[n].response = hso_or_shi_tomasi(img, w, h, x, y);
      }
      n++;
    }
  free(sc);
  return n;
}

int hso_or_fast_detect_level(const uint8_t* img, int w, int h, int threshold, int border, hso_corner* out, int cap)
{
  return detect_level_arc(img, w, h, threshold, border, 9, out, cap);
}

/* FeatureExtractor::fillingHole, src/feature_detection.cpp:1125-1154 (the initialisation branch of
 * detect, :439-442, level 0 only): FAST-12 (fast_corner_detect_plain_12 + fast_corner_score_12 +
 * fast_nonmax_3x3) at barrier max(0.6 * minThresh, 6) truncated to short; a survivor inside the
 * border is kept only when its grid index (getCellIndex) holds no feature yet, and then occupies it.
 * have: haveFeatures_[level] as the FAST-9 stage left it (updated).  out: kGrad key points in the
 * order they are pushed; returns their number. */
int hso_or_filling_hole_level(const uint8_t* img, int w, int h, int level, int frame_w, int frame_h, int min_thresh, uint8_t* have,
                              hso_corner* out, int cap)
{
  const short fastThresh = 0.6 * min_thresh > 6 ? 0.6 * min_thresh : 6;
  int grid, gcols, grows, lw, lh;
  hso_or_detect_grid(frame_w, frame_h, level, &grid, &gcols, &grows, &lw, &lh);
  hso_corner* all = (hso_corner*)malloc(sizeof(hso_corner) * (size_t)w * h);
  const int n_all = detect_level_arc(img, w, h, fastThresh, 8, 12, all, w * h);
  int n = 0;
  for (int i = 0; i < n_all; i++) {
    const int index = hso_or_detect_cell_index(all[i].x, all[i].y, grid, gcols, grows);
    if (have[index]) continue;
    have[index] = 1;
    if (n < cap) out[n] = all[i];
    n++;
  }
  free(all);
  return n;
}
