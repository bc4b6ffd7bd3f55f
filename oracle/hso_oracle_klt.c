/* hso_oracle_klt.c — CPU restatement of the image side of the two-view initialisation:
 * initialization::trackKlt (reference src/initialization.cpp:225-300) = cv::calcOpticalFlowPyrLK(img_prev, img_cur, px_prev,
 * px_cur, status, error, Size(30, 30), 4, TermCriteria(COUNT + EPS, 30, 0.0001), OPTFLOW_USE_INITIAL_FLOW), followed by
 * patchCheck / createPatch / checkSSD (:476-563; checkSSD is, despite its name, a zero-mean NCC > 0.8 test) per point.
 *
 * TEST INFRASTRUCTURE (like everything under oracle/): only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may use it.
 *
 * PARITY UNPINNED: OpenCV is an absent third-party dependency (README.md:30 "tested with OpenCV 3.2.0"; CMakeLists.txt:44-50), not
 * installed here, and the reference holds no vectors for this function.  Restated from OpenCV's published algorithm
 * (modules/video/src/lkpyramid.cpp: buildOpticalFlowPyramid, calcSharrDeriv, LKTrackerInvoker; modules/imgproc pyrDown), scalar
 * path:
 *   pyramid     level 0 = the image, level L+1 = pyrDown(level L): separable [1 4 6 4 1] / 16, BORDER_REFLECT_101, 8-bit result
 *               (sum + 128) >> 8, size ((w + 1) / 2, (h + 1) / 2); the pyramid ends at the first level whose successor would not
 *               be larger than the window in both dimensions (752x480, window 30: levels 0..3);
 *   derivatives calcSharrDeriv: 3x3 Scharr (3, 10, 3) into int16, BORDER_REFLECT_101 inside the image; ZERO outside the image
 *               (the derivative pyramid's border is BORDER_CONSTANT) while the intensity outside is the reflected image
 *               (BORDER_REFLECT_101): a 30x30 window may hang over the image edge by up to the window size;
 *   per point   coarse to fine; the template window is sampled with 14-bit fixed-point bilinear weights into int16 (intensity
 *               scaled by 32: CV_DESCALE(.., W_BITS - 5)), the 2x2 gradient matrix accumulated in float and scaled by 2^-20; the
 *               minimum-eigenvalue test (1e-4) and D < FLT_EPSILON drop the level (status false on level 0); up to 30 iterations
 *               delta = A^-1 b with the stop tests |delta|^2 <= eps^2 and the oscillation test (|delta + prevDelta| < 0.01 in both
 *               coordinates: half a step back and stop); a window start outside [-win, size) on level 0 clears the status. */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "hso_oracle.h"

static int reflect101(int i, int n)
{
  if (n == 1) return 0;
  while (i < 0 || i >= n) { if (i < 0) i = -i; else i = 2 * n - 2 - i; }
  return i;
}

/* cv::pyrDown (8-bit, BORDER_REFLECT_101): dst is ((w + 1) / 2) x ((h + 1) / 2) */
void hso_or_pyr_down(const uint8_t* src, int w, int h, uint8_t* dst)
{
  const int dw = (w + 1) / 2, dh = (h + 1) / 2;
  int* rows = (int*)malloc(sizeof(int) * (size_t)dw * 5);
  for (int y = 0; y < dh; y++) {
    for (int k = 0; k < 5; k++) {
      const uint8_t* s = src + (size_t)reflect101(2 * y - 2 + k, h) * w;
      int* r = rows + (size_t)k * dw;
      for (int x = 0; x < dw; x++) {
        const int x0 = reflect101(2 * x - 2, w), x1 = reflect101(2 * x - 1, w), x2 = reflect101(2 * x, w), x3 = reflect101(2 * x + 1, w),
                  x4 = reflect101(2 * x + 2, w);
        r[x] = s[x2] * 6 + (s[x1] + s[x3]) * 4 + s[x0] + s[x4];
      }
    }
    for (int x = 0; x < dw; x++) {
      const int v = rows[2 * dw + x] * 6 + (rows[dw + x] + rows[3 * dw + x]) * 4 + rows[x] + rows[4 * dw + x];
      dst[(size_t)y * dw + x] = (uint8_t)((v + 128) >> 8);
    }
  }
  free(rows);
}

/* calcSharrDeriv: d[2 * (y * w + x)] = Ix, [.. + 1] = Iy */
void hso_or_scharr_deriv(const uint8_t* src, int w, int h, int16_t* d)
{
  for (int y = 0; y < h; y++) {
    const uint8_t* r0 = src + (size_t)reflect101(y - 1, h) * w;
    const uint8_t* r1 = src + (size_t)y * w;
    const uint8_t* r2 = src + (size_t)reflect101(y + 1, h) * w;
    for (int x = 0; x < w; x++) {
      const int xm = reflect101(x - 1, w), xp = reflect101(x + 1, w);
      /* trow0 = 3 (r0 + r2) + 10 r1 (smoothed in y), trow1 = r2 - r0 (differentiated in y) */
      const int t0m = (r0[xm] + r2[xm]) * 3 + r1[xm] * 10, t0p = (r0[xp] + r2[xp]) * 3 + r1[xp] * 10;
      const int t1m = r2[xm] - r0[xm], t1c = r2[x] - r0[x], t1p = r2[xp] - r0[xp];
      d[2 * ((size_t)y * w + x)] = (int16_t)(t0p - t0m);
      d[2 * ((size_t)y * w + x) + 1] = (int16_t)((t1m + t1p) * 3 + t1c * 10);
    }
  }
}

/* effective number of pyramid levels for buildOpticalFlowPyramid(img, win, max_level): returns the last level index */
int hso_or_klt_levels(int w, int h, int win, int max_level)
{
  for (int level = 0; level <= max_level; level++) {
    w = (w + 1) / 2; h = (h + 1) / 2;
    if (w <= win || h <= win) return level;
  }
  return max_level;
}

static inline int cv_round(float v) { return (int)lrintf(v); }   /* cvRound: round half to even */
#define DESCALE(x, n) (((x) + (1 << ((n) - 1))) >> (n))

/* One pyramid level of LKTrackerInvoker for one point.  I / J: level images, dI: the interleaved derivative image of I.
 * Returns 0 when the level was skipped / failed (status semantics are the caller's: cleared only on level 0). */
static int klt_level(const uint8_t* I, const uint8_t* J, const int16_t* dI, int w, int h, int win, int level, int max_level,
                     int use_initial, int max_count, double eps2, float min_eig_thr, const float prev_in[2], float next[2], int* status,
                     float* margin)
{
#define NOTE(v) do { const float m_ = fabsf(v); if (m_ < *margin) *margin = m_; } while (0)
  const float half = (win - 1) * 0.5f;
  float prev[2] = { prev_in[0] * (float)(1. / (1 << level)), prev_in[1] * (float)(1. / (1 << level)) };
  float nxt[2];
  if (level == max_level) {
    if (use_initial) { nxt[0] = next[0] * (float)(1. / (1 << level)); nxt[1] = next[1] * (float)(1. / (1 << level)); }
    else { nxt[0] = prev[0]; nxt[1] = prev[1]; }
  } else { nxt[0] = next[0] * 2.f; nxt[1] = next[1] * 2.f; }
  next[0] = nxt[0]; next[1] = nxt[1];
  prev[0] -= half; prev[1] -= half;
  const int ipx = (int)floorf(prev[0]), ipy = (int)floorf(prev[1]);
  if (ipx < -win || ipx >= w || ipy < -win || ipy >= h) { if (level == 0) *status = 0; return 0; }
  float a = prev[0] - ipx, b = prev[1] - ipy;
  const int W_BITS = 14;
  const float FLT_SCALE = 1.f / (1 << 20);
  int iw00 = cv_round((1.f - a) * (1.f - b) * (1 << W_BITS)), iw01 = cv_round(a * (1.f - b) * (1 << W_BITS)),
      iw10 = cv_round((1.f - a) * b * (1 << W_BITS)), iw11 = (1 << W_BITS) - iw00 - iw01 - iw10;
  short* Iw = (short*)malloc(sizeof(short) * 3 * (size_t)win * win);
  short* dIw = Iw + (size_t)win * win;
  float iA11 = 0, iA12 = 0, iA22 = 0;
  /* intensity outside the image: BORDER_REFLECT_101; derivative outside: 0 (BORDER_CONSTANT) */
#define PIX(img, xx, yy) ((int)(img)[(size_t)reflect101((yy), h) * w + reflect101((xx), w)])
#define DER(c, xx, yy) (((xx) < 0 || (xx) >= w || (yy) < 0 || (yy) >= h) ? 0 : (int)dI[2 * ((size_t)(yy) * w + (xx)) + (c)])
  for (int y = 0; y < win; y++)
    for (int x = 0; x < win; x++) {
      const int X = ipx + x, Y = ipy + y;
      const int ival = DESCALE(PIX(I, X, Y) * iw00 + PIX(I, X + 1, Y) * iw01 + PIX(I, X, Y + 1) * iw10 + PIX(I, X + 1, Y + 1) * iw11, W_BITS - 5);
      const int ixval = DESCALE(DER(0, X, Y) * iw00 + DER(0, X + 1, Y) * iw01 + DER(0, X, Y + 1) * iw10 + DER(0, X + 1, Y + 1) * iw11, W_BITS);
      const int iyval = DESCALE(DER(1, X, Y) * iw00 + DER(1, X + 1, Y) * iw01 + DER(1, X, Y + 1) * iw10 + DER(1, X + 1, Y + 1) * iw11, W_BITS);
      Iw[y * win + x] = (short)ival; dIw[2 * (y * win + x)] = (short)ixval; dIw[2 * (y * win + x) + 1] = (short)iyval;
      iA11 += (float)(ixval * ixval); iA12 += (float)(ixval * iyval); iA22 += (float)(iyval * iyval);
    }
  const float A11 = iA11 * FLT_SCALE, A12 = iA12 * FLT_SCALE, A22 = iA22 * FLT_SCALE;
  float D = A11 * A22 - A12 * A12;
  const float minEig = (A22 + A11 - sqrtf((A11 - A22) * (A11 - A22) + 4.f * A12 * A12)) / (2 * win * win);
  NOTE((minEig - min_eig_thr) / min_eig_thr);
  if (minEig < min_eig_thr || D < 1.1920929e-07f) { if (level == 0) *status = 0; free(Iw); return 0; }
  D = 1.f / D;
  nxt[0] -= half; nxt[1] -= half;
  float pdx = 0, pdy = 0;
  int j;
  for (j = 0; j < max_count; j++) {
    const int inx = (int)floorf(nxt[0]), iny = (int)floorf(nxt[1]);
    if (inx < -win || inx >= w || iny < -win || iny >= h) { if (level == 0) *status = 0; break; }
    a = nxt[0] - inx; b = nxt[1] - iny;
    iw00 = cv_round((1.f - a) * (1.f - b) * (1 << W_BITS)); iw01 = cv_round(a * (1.f - b) * (1 << W_BITS));
    iw10 = cv_round((1.f - a) * b * (1 << W_BITS)); iw11 = (1 << W_BITS) - iw00 - iw01 - iw10;
    float ib1 = 0, ib2 = 0;
    for (int y = 0; y < win; y++)
      for (int x = 0; x < win; x++) {
        const int X = inx + x, Y = iny + y;
        const int diff = DESCALE(PIX(J, X, Y) * iw00 + PIX(J, X + 1, Y) * iw01 + PIX(J, X, Y + 1) * iw10 + PIX(J, X + 1, Y + 1) * iw11, W_BITS - 5)
                         - Iw[y * win + x];
        ib1 += (float)(diff * dIw[2 * (y * win + x)]); ib2 += (float)(diff * dIw[2 * (y * win + x) + 1]);
      }
    const float b1 = ib1 * FLT_SCALE, b2 = ib2 * FLT_SCALE;
    const float dx = (float)((A12 * b2 - A22 * b1) * D), dy = (float)((A12 * b1 - A11 * b2) * D);
    nxt[0] += dx; nxt[1] += dy;
    next[0] = nxt[0] + half; next[1] = nxt[1] + half;
    NOTE((float)(((double)dx * dx + (double)dy * dy - eps2) / eps2));
    if ((double)dx * dx + (double)dy * dy <= eps2) break;
    if (j > 0) {   /* the oscillation test: the distance of the coordinate that decides it to the bound, in units of the bound */
      const float ox = fabsf(dx + pdx) - 0.01f, oy = fabsf(dy + pdy) - 0.01f;
      NOTE((ox > oy ? ox : oy) / 0.01f);
    }
    if (j > 0 && fabsf(dx + pdx) < 0.01f && fabsf(dy + pdy) < 0.01f) { next[0] -= dx * 0.5f; next[1] -= dy * 0.5f; break; }
    pdx = dx; pdy = dy;
  }
  if (j == max_count) NOTE(0.f);   /* the iteration cap ended a level while the point still moved: the result is an unconverged iterate */
#undef PIX
#undef DER
#undef NOTE
  free(Iw);
  return 1;
}

/* cv::calcOpticalFlowPyrLK over host images.  pts_prev / pts_cur: n x 2 floats (pts_cur in: the initial flow; out: the result).
 * margin (nullable, n floats): per point the smallest relative distance of a decision of the iteration to its bound (the
 * minimum-eigenvalue test, the epsilon stop, the oscillation stop): a parity test excuses a point whose decisions were that close. */
void hso_or_klt_track(const uint8_t* prev, const uint8_t* cur, int w, int h, const float* pts_prev, float* pts_cur, uint8_t* status, int n,
                      int win, int max_level, int max_count, double epsilon, int use_initial_flow, float* margin)
{
  const int L = hso_or_klt_levels(w, h, win, max_level);
  uint8_t* pi[8]; uint8_t* pj[8]; int16_t* dd[8]; int lw[8], lh[8];
  lw[0] = w; lh[0] = h; pi[0] = (uint8_t*)prev; pj[0] = (uint8_t*)cur;
  for (int l = 1; l <= L; l++) {
    lw[l] = (lw[l - 1] + 1) / 2; lh[l] = (lh[l - 1] + 1) / 2;
    pi[l] = (uint8_t*)malloc((size_t)lw[l] * lh[l]); pj[l] = (uint8_t*)malloc((size_t)lw[l] * lh[l]);
    hso_or_pyr_down(pi[l - 1], lw[l - 1], lh[l - 1], pi[l]);
    hso_or_pyr_down(pj[l - 1], lw[l - 1], lh[l - 1], pj[l]);
  }
  for (int l = 0; l <= L; l++) { dd[l] = (int16_t*)malloc(sizeof(int16_t) * 2 * (size_t)lw[l] * lh[l]); hso_or_scharr_deriv(pi[l], lw[l], lh[l], dd[l]); }
  if (max_count > 100) max_count = 100;
  if (max_count < 0) max_count = 0;
  if (epsilon < 0) epsilon = 0;
  if (epsilon > 10) epsilon = 10;
  const double eps2 = epsilon * epsilon;
  for (int i = 0; i < n; i++) {
    int st = 1;
    float mg = 3.4e38f;
    float next[2] = { pts_cur[2 * i], pts_cur[2 * i + 1] };
    for (int l = L; l >= 0; l--)
      klt_level(pi[l], pj[l], dd[l], lw[l], lh[l], win, l, L, use_initial_flow, max_count, eps2, 1e-4f, &pts_prev[2 * i], next, &st, &mg);
    if (margin) margin[i] = mg;
    pts_cur[2 * i] = next[0]; pts_cur[2 * i + 1] = next[1];
    status[i] = (uint8_t)st;
  }
  for (int l = 0; l <= L; l++) { free(dd[l]); if (l) { free(pi[l]); free(pj[l]); } }
}

/* initialization::createPatch + checkSSD (src/initialization.cpp:470-520): 8x8 bilinear patches around the two positions; the
 * check passes when both patches exist and their zero-mean SSD stays below the reference's bound. */
static int create_patch(const uint8_t* img, int w, int h, float u, float v, float* patch)
{
  const int ui = (int)floorf(u), vi = (int)floorf(v);
  if (ui < 4 || ui >= w - 4 || vi < 4 || vi >= h - 4) return 0;
  const float su = u - ui, sv = v - vi;
  const float wtl = (1.0 - su) * (1.0 - sv), wtr = su * (1.0 - sv), wbl = (1.0 - su) * sv, wbr = su * sv;
  for (int y = 0; y < 8; y++) {
    const uint8_t* p = img + (size_t)(vi - 4 + y) * w + (ui - 4);
    for (int x = 0; x < 8; x++, p++) patch[y * 8 + x] = wtl * p[0] + wtr * p[1] + wbl * p[w] + wbr * p[w + 1];
  }
  return 1;
}

int hso_or_patch_check(const uint8_t* img_pre, const uint8_t* img_cur, int w, int h, const float px_pre[2], const float px_cur[2], float* ncc_out)
{
  float a[64], b[64];
  if (ncc_out) *ncc_out = -2;
  if (!create_patch(img_pre, w, h, px_pre[0], px_pre[1], a) || !create_patch(img_cur, w, h, px_cur[0], px_cur[1], b)) return 0;
  /* checkSSD (:522-563): zero-mean normalised cross correlation of the 64 pixels against 0.8 */
  float ma = 0, mb = 0;
  for (int i = 0; i < 64; i++) { ma += a[i]; mb += b[i]; }
  ma /= 64; mb /= 64;
  float num = 0, d1 = 0, d2 = 0;
  for (int i = 0; i < 64; i++) {
    num += (a[i] - ma) * (b[i] - mb);
    d1 += (a[i] - ma) * (a[i] - ma);
    d2 += (b[i] - mb) * (b[i] - mb);
  }
  const double ncc = (double)num / ((double)sqrtf(d1 * d2) + 1e-12);   /* C++ sqrt(float) is the float overload; + 1e-12 promotes */
  if (ncc_out) *ncc_out = (float)ncc;
  return ncc > 0.8f;
}
