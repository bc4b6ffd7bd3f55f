/*
 * hso_oracle_edgelet.c — edgelet candidates: cv::Canny on the Sobel-5 images and the per-cell
 * strongest edge pixel: FeatureExtractor::edgeLetDetectST restated (src/feature_detection.cpp:
 * 749-830), with getCellIndex (include/hso/feature_detection.h:295-299) for the occupancy the
 * FAST stage leaves behind.  TEST INFRASTRUCTURE (see hso_oracle.h); unpinned: OpenCV is an absent
 * dependency, cv::Canny(dx, dy, edges, t1, t2, L2gradient = true) is restated from its published
 * implementation (imgproc/src/canny.cpp, custom-gradient overload, OpenCV >= 3.2):
 *   - squared thresholds: t = min(32767, t); t *= t; low = floor(t1), high = floor(t2);
 *   - magnitude m = dx^2 + dy^2 in int, zero outside the image;
 *   - non-maximum suppression in three direction sectors chosen with the fixed-point tangent
 *     TG22 = round(tan(22.5 deg) * 2^15): horizontal (m > left && m >= right), vertical
 *     (m > up && m >= down), diagonal (m > both diagonal neighbours, strict);
 *   - hysteresis: survivors with m > high seed a flood fill over survivors with m > low
 *     (8-neighbourhood).  (The implementation's "already pushed" shortcuts only skip seeds that
 *     the fill reaches anyway; the resulting edge map is the standard double-threshold one.)
 * Quirks of the caller kept: the cell origin uses index / vGridRows_ for y (:770) and
 * getCellIndex divides x by vGridRows_ (:298).
 */
#include "hso_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* cv::Canny(dx, dy, edges, low_thresh, high_thresh, true); edges: 0 / 255 */
void hso_or_canny_l2(const int16_t* dx, const int16_t* dy, int w, int h, double low_thresh, double high_thresh, uint8_t* edges)
{
  if (low_thresh > high_thresh) { const double t = low_thresh; low_thresh = high_thresh; high_thresh = t; }
  low_thresh = low_thresh < 32767.0 ? low_thresh : 32767.0;
  high_thresh = high_thresh < 32767.0 ? high_thresh : 32767.0;
  if (low_thresh > 0) low_thresh *= low_thresh;
  if (high_thresh > 0) high_thresh *= high_thresh;
  const int low = (int)floor(low_thresh), high = (int)floor(high_thresh);
  const int CANNY_SHIFT = 15;
  const int TG22 = (int)(0.4142135623730950488016887242097 * (1 << CANNY_SHIFT) + 0.5);
  const int mw = w + 2;
  int* mag = (int*)calloc((size_t)mw * (h + 2), sizeof(int));      /* zero border */
  uint8_t* map = (uint8_t*)malloc((size_t)w * h);                   /* 0 weak survivor, 1 no edge, 2 edge */
#define MAG(x, y) mag[(size_t)((y) + 1) * mw + (x) + 1]
  for (int y = 0; y < h; y++)
    for (int x = 0; x < w; x++) {
      const int gx = dx[(size_t)y * w + x], gy = dy[(size_t)y * w + x];
      MAG(x, y) = gx * gx + gy * gy;
    }
  int* stack = (int*)malloc(sizeof(int) * (size_t)w * h);
  int sp = 0;
  for (int y = 0; y < h; y++)
    for (int x = 0; x < w; x++) {
      const int m = MAG(x, y);
      int keep = 0;
      if (m > low) {
        const int xs = dx[(size_t)y * w + x], ys = dy[(size_t)y * w + x];
        const int ax = abs(xs), ay = abs(ys) << CANNY_SHIFT;
        const int tg22x = ax * TG22;
        if (ay < tg22x) {
          keep = m > MAG(x - 1, y) && m >= MAG(x + 1, y);
        } else {
          const int tg67x = tg22x + (ax << (CANNY_SHIFT + 1));
          if (ay > tg67x) {
            keep = m > MAG(x, y - 1) && m >= MAG(x, y + 1);
          } else {
            const int s = (xs ^ ys) < 0 ? -1 : 1;
            keep = m > MAG(x - s, y - 1) && m > MAG(x + s, y + 1);
          }
        }
      }
      if (!keep) { map[(size_t)y * w + x] = 1; continue; }
      if (m > high) { map[(size_t)y * w + x] = 2; stack[sp++] = y * w + x; }
      else map[(size_t)y * w + x] = 0;
    }
  while (sp > 0) {
    const int p = stack[--sp], px = p % w, py = p / w;
    for (int dyy = -1; dyy <= 1; dyy++)
      for (int dxx = -1; dxx <= 1; dxx++) {
        const int nx = px + dxx, ny = py + dyy;
        if (nx < 0 || ny < 0 || nx >= w || ny >= h) continue;
        if (map[(size_t)ny * w + nx] == 0) { map[(size_t)ny * w + nx] = 2; stack[sp++] = ny * w + nx; }
      }
  }
  for (size_t i = 0; i < (size_t)w * h; i++) edges[i] = map[i] == 2 ? 255 : 0;
#undef MAG
  free(mag); free(map); free(stack);
}

/* grid geometry of one level (FeatureExtractor ctor, src/feature_detection.cpp:355-400) */
void hso_or_detect_grid(int width, int height, int level, int* grid, int* gcols, int* grows, int* lw, int* lh)
{
  int w = width, h = height;
  for (int i = 0; i < level; i++) { w /= 2; h /= 2; }     /* vecWidth_[i] = vecWidth_[i-1] / 2 */
  const int g = 8 / (1 << level);                           /* gridSize_ = 8 */
  *grid = g; *lw = w; *lh = h;
  *gcols = (int)ceil((double)w / g);
  *grows = (int)ceil((double)h / g);
}

/* getCellIndex, include/hso/feature_detection.h:295-299 (x is divided by vGridRows_, as written there) */
int hso_or_detect_cell_index(int x, int y, int grid, int gcols, int grows)
{
  return (int)(y / grid * gcols + x / grows);
}

/* FeatureExtractor::edgeLetDetectST for one level, src/feature_detection.cpp:749-830.
 * gx, gy: the level's Sobel-5 images; have: haveFeatures_[level] (gcols * grows flags, updated);
 * out: up to cap edgelets in cell-index order; returns their number. */
int hso_or_edgelet_level(const int16_t* gx, const int16_t* gy, int w, int h, int level, int frame_w, int frame_h, int min_thresh,
                         uint8_t* have, hso_edgelet* out, int cap)
{
  int grid, gcols, grows, lw, lh;
  hso_or_detect_grid(frame_w, frame_h, level, &grid, &gcols, &grows, &lw, &lh);
  uint8_t* edge = (uint8_t*)malloc((size_t)w * h);
  hso_or_canny_l2(gx, gy, w, h, 31 * min_thresh, 70 * min_thresh, edge);
  const int border = 8;
  const int maxBorderX = w - border, maxBorderY = h - border;
  int n = 0;
  for (int index = 0; index < gcols * grows; ++index) {
    if (have[index]) continue;
    int iniX = index % gcols * grid;
    int iniY = index / grows * grid;                      /* sic: rows, :770 */
    if (iniX > maxBorderX || iniY > maxBorderY) continue;
    int maxX = iniX + grid, maxY = iniY + grid;
    if (maxX > maxBorderX) maxX = maxBorderX;
    if (maxY > maxBorderY) maxY = maxBorderY;
    if (iniX < border) iniX = border;
    if (iniY < border) iniY = border;
    int isSet = 0;
    float maxGrad = 0;
    hso_edgelet kp;
    memset(&kp, 0, sizeof(kp));
    for (int y = iniY; y < maxY; ++y)
      for (int x = iniX; x < maxX; ++x) {
        if (edge[(size_t)y * w + x] == 0) continue;
        const short sx = gx[(size_t)y * w + x], sy = gy[(size_t)y * w + x];
        const float grad = sqrtf(sx * sx + sy * sy);
        if (grad > maxGrad) {
          kp.x = (int16_t)x; kp.y = (int16_t)y; kp.gx = sx; kp.gy = sy; kp.grad = grad;
          isSet = 1; maxGrad = grad;
        }
      }
    if (isSet) {
      if (n < cap) out[n] = kp;
      n++;
      have[index] = 1;
    }
  }
  free(edge);
  return n;
}
