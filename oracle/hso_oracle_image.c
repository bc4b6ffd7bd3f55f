/*
 * hso_oracle_image.c — pyramid, Sobel, frame statistics, camera models.
 * TEST INFRASTRUCTURE (see hso_oracle.h).
 */
#include "hso_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ---------------------------------------------------------------- pyramid */

/* src/vikit/vision.cpp:19-44 (halfSampleSSE2): _mm_avg_epu8 of the two rows,
 * then _mm_avg_epu16 of the even/odd columns — round-half-up at both stages.
 * sw = w >> 4 sixteen-pixel groups per row, sh = h >> 1 rows. */
static void half_sample_sse2_rounding(const uint8_t* in, uint8_t* out, int w, int h)
{
  const int sw = w >> 4, sh = h >> 1;
  const uint8_t* next = in + w;
  for (int i = 0; i < sh; i++) {
    for (int j = 0; j < sw; j++) {
      for (int k = 0; k < 8; k++) {
        const unsigned a = (in[2 * k] + next[2 * k] + 1u) >> 1;
        const unsigned b = (in[2 * k + 1] + next[2 * k + 1] + 1u) >> 1;
        out[k] = (uint8_t)((a + b + 1u) >> 1);
      }
      in += 16; next += 16; out += 8;
    }
    in += w; next += w;
  }
}

/* src/vikit/vision.cpp:70-108.  cv::Mat buffers are 16-byte aligned (OpenCV
 * fastMalloc), so the SSE2 path is taken iff in.cols % 16 == 0 (:76);
 * otherwise the scalar loop (:92-107) which truncates (a+b+c+d)/4 and walks
 * raw pointers (`top += 2` out_width times, then `top += stride`). */
void hso_or_half_sample(const uint8_t* in, int w, int h, uint8_t* out)
{
  if ((w % 16) == 0) {
    half_sample_sse2_rounding(in, out, w, h);
    return;
  }
  const int stride = w;
  const uint8_t* top = in;
  const uint8_t* bottom = top + stride;
  const uint8_t* end = top + stride * h;
  const int out_width = w / 2;
  uint8_t* p = out;
  while (bottom < end) {
    for (int j = 0; j < out_width; j++) {
      *p = (uint8_t)(((uint16_t)top[0] + top[1] + bottom[0] + bottom[1]) / 4);
      p++; top += 2; bottom += 2;
    }
    top += stride; bottom += stride;
  }
}

/* cvRound: lrint under the default rounding mode = round half to even (OpenCV's SSE2 cvRound) */
static int cv_round(double v) { return (int)lrint(v); }
static int cv_floor(double v) { int i = (int)v; return i - (i > v); }
static short sat_short(float v) { int i = cv_round(v); return (short)(i < -32768 ? -32768 : (i > 32767 ? 32767 : i)); }

/* size of pyramid level `level` of a w x h frame, src/frame.cpp:302-312 */
void hso_or_pyramid_dims(int w, int h, int level, int* lw, int* lh)
{
  if ((w % 16) == 0 && (h % 16) == 0) { *lw = w >> level; *lh = h >> level; return; }
  const float scale = 1.0 / (1 << level);
  *lw = cv_round((float)w * scale);
  *lh = cv_round((float)h * scale);
}

/* cv::resize(src, dst, dsize, 0, 0, cv::INTER_LINEAR) for CV_8UC1 (src/frame.cpp:311), restated from
 * OpenCV's imgproc/resize.cpp (not vendored; README.md:30 "tested 3.2.0"; the C/SIMD path — an IPP
 * build of OpenCV rounds differently): an exact 2x2 decimation is redirected to INTER_AREA's fast
 * path, (a + b + c + d + 2) >> 2; everything else is the fixed-point bilinear kernel with
 * INTER_RESIZE_COEF_BITS = 11: horizontal pass in int with coefficients rounded to short
 * (1 - fx, fx) * 2048, vertical pass ((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2,
 * pixel centres aligned ((d + 0.5) * scale - 0.5), source indices clamped at the borders. */
void hso_or_resize_linear_8u(const uint8_t* src, int sw, int sh, uint8_t* dst, int dw, int dh)
{
  const double inv_scale_x = (double)dw / sw, inv_scale_y = (double)dh / sh;
  const double scale_x = 1. / inv_scale_x, scale_y = 1. / inv_scale_y;
  const int iscale_x = cv_round(scale_x), iscale_y = cv_round(scale_y);
  const int is_area_fast = fabs(scale_x - iscale_x) < 2.220446049250313e-16 && fabs(scale_y - iscale_y) < 2.220446049250313e-16;
  if (is_area_fast && iscale_x == 2 && iscale_y == 2) {
    for (int y = 0; y < dh; y++)
      for (int x = 0; x < dw; x++) {
        const uint8_t* s0 = src + (size_t)(2 * y) * sw + 2 * x;
        dst[(size_t)y * dw + x] = (uint8_t)((s0[0] + s0[1] + s0[sw] + s0[sw + 1] + 2) >> 2);
      }
    return;
  }
  int* xofs = (int*)malloc(sizeof(int) * (size_t)dw);
  short* ialpha = (short*)malloc(sizeof(short) * 2 * (size_t)dw);
  int xmax = dw;
  for (int dx = 0; dx < dw; dx++) {
    float fx = (float)((dx + 0.5) * scale_x - 0.5);
    int sx = cv_floor(fx);
    fx -= sx;
    if (sx < 0) { fx = 0; sx = 0; }
    if (sx + 1 >= sw) { if (dx < xmax) xmax = dx; if (sx >= sw - 1) { fx = 0; sx = sw - 1; } }
    xofs[dx] = sx;
    ialpha[2 * dx] = sat_short((1.f - fx) * 2048);
    ialpha[2 * dx + 1] = sat_short(fx * 2048);
  }
  int* row0 = (int*)malloc(sizeof(int) * (size_t)dw);
  int* row1 = (int*)malloc(sizeof(int) * (size_t)dw);
  for (int dy = 0; dy < dh; dy++) {
    float fy = (float)((dy + 0.5) * scale_y - 0.5);
    const int sy0 = cv_floor(fy);
    fy -= sy0;
    const short b0 = sat_short((1.f - fy) * 2048), b1 = sat_short(fy * 2048);
    int* rows[2] = { row0, row1 };
    for (int k = 0; k < 2; k++) {
      int sy = sy0 + k;
      sy = sy >= 0 ? (sy < sh ? sy : sh - 1) : 0;
      const uint8_t* S = src + (size_t)sy * sw;
      int* D = rows[k];
      int dx = 0;
      for (; dx < xmax; dx++) D[dx] = S[xofs[dx]] * ialpha[2 * dx] + S[xofs[dx] + 1] * ialpha[2 * dx + 1];
      for (; dx < dw; dx++) D[dx] = S[xofs[dx]] * 2048;
    }
    for (int x = 0; x < dw; x++)
      dst[(size_t)dy * dw + x] = (uint8_t)((((b0 * (row0[x] >> 4)) >> 16) + ((b1 * (row1[x] >> 4)) >> 16) + 2) >> 2);
  }
  free(xofs); free(ialpha); free(row0); free(row1);
}

/* frame_utils::createImgPyramid, src/frame.cpp:296-314: halfSample when the level-0 size is a
 * multiple of 16 in both dimensions (:302), cv::resize of the previous level otherwise (:307-312).
 * levels[i] must hold the size hso_or_pyramid_dims reports. */
int hso_or_create_pyramid(const uint8_t* img, int w, int h, uint8_t* const levels[HSO_N_PYR_LEVELS])
{
  memcpy(levels[0], img, (size_t)w * h);
  if ((w % 16) == 0 && (h % 16) == 0) {
    int lw = w, lh = h;
    for (int i = 1; i < HSO_N_PYR_LEVELS; i++) {
      hso_or_half_sample(levels[i - 1], lw, lh, levels[i]);
      lw /= 2; lh /= 2;
    }
    return 0;
  }
  int pw = w, ph = h;
  for (int i = 1; i < HSO_N_PYR_LEVELS; i++) {
    int lw, lh;
    hso_or_pyramid_dims(w, h, i, &lw, &lh);
    hso_or_resize_linear_8u(levels[i - 1], pw, ph, levels[i], lw, lh);
    pw = lw; ph = lh;
  }
  return 0;
}

/* cv::Sobel(src, dst, CV_16S, dx, dy, 5, 1, 0, BORDER_REPLICATE), src/frame.cpp:218-219.
 * OpenCV (not vendored; README.md:30 "tested 3.2.0") builds the separable
 * kernels with getSobelKernels(ksize=5, normalize=false): derivative
 * [-1 -2 0 2 1], smoothing [1 4 6 4 1]; integer arithmetic, exact. */
void hso_or_sobel5(const uint8_t* img, int w, int h, int16_t* gx, int16_t* gy)
{
  static const int kd[5] = { -1, -2, 0, 2, 1 };
  static const int ks[5] = { 1, 4, 6, 4, 1 };
  int* rowd = (int*)malloc(sizeof(int) * (size_t)w * h);
  int* rows = (int*)malloc(sizeof(int) * (size_t)w * h);
  for (int y = 0; y < h; y++)
    for (int x = 0; x < w; x++) {
      int sd = 0, ss = 0;
      for (int i = 0; i < 5; i++) {
        int xx = x + i - 2;
        if (xx < 0) xx = 0;
        if (xx > w - 1) xx = w - 1;
        const int v = img[y * w + xx];
        sd += kd[i] * v; ss += ks[i] * v;
      }
      rowd[y * w + x] = sd; rows[y * w + x] = ss;
    }
  for (int y = 0; y < h; y++)
    for (int x = 0; x < w; x++) {
      int sx = 0, sy = 0;
      for (int j = 0; j < 5; j++) {
        int yy = y + j - 2;
        if (yy < 0) yy = 0;
        if (yy > h - 1) yy = h - 1;
        sx += ks[j] * rowd[yy * w + x];
        sy += kd[j] * rows[yy * w + x];
      }
      gx[y * w + x] = (int16_t)sx;
      gy[y * w + x] = (int16_t)sy;
    }
  free(rowd); free(rows);
}

/* src/frame.cpp:223-245: serial fp32 sums over the level-0 interior. */
void hso_or_frame_stats(const uint8_t* img0, const int16_t* gx0, const int16_t* gy0, int w, int h,
                        hso_frame_stats* out)
{
  float intSum = 0, gradSum = 0;
  int sum = 0;
  for (int y = 16; y < h - 16; y++)
    for (int x = 16; x < w - 16; x++) {
      sum++;
      const float gradx = gx0[y * w + x];
      const float grady = gy0[y * w + x];
      gradSum += sqrtf(gradx * gradx + grady * grady);
      intSum += img0[y * w + x];
    }
  out->integral_image = intSum / sum;
  float gm = gradSum / sum;
  gm /= 30;
  if (gm > 20) gm = 20;
  if (gm < 7) gm = 7;
  out->grad_mean = gm;
  out->width = w; out->height = h;
}

/* ----------------------------------------------------------------- camera */

/* src/camera.cpp:94-125 (Pinhole), :196-221 (FOV), :295-303 (Equidistant) */
void hso_or_world2cam(const hso_camera* cam, const double xyz[3], double px[2])
{
  const double u = xyz[0] / xyz[2], v = xyz[1] / xyz[2];
  if (cam->model == HSO_CAM_PINHOLE && cam->distortion) {
    double x, y, r2, r4, r6, a1, a2, a3, cdist, xd, yd;
    x = u; y = v;
    r2 = x * x + y * y;
    r4 = r2 * r2;
    r6 = r4 * r2;
    a1 = 2 * x * y;
    a2 = r2 + 2 * x * x;
    a3 = r2 + 2 * y * y;
    cdist = 1 + cam->d[0] * r2 + cam->d[1] * r4 + cam->d[4] * r6;
    xd = x * cdist + cam->d[2] * a1 + cam->d[3] * a2;
    yd = y * cdist + cam->d[2] * a3 + cam->d[3] * a1;
    px[0] = xd * cam->fx + cam->cx;
    px[1] = yd * cam->fy + cam->cy;
  } else if (cam->model == HSO_CAM_FOV && cam->distortion) {
    const double omega = cam->d[0];
    const double dist = sqrt(u * u + v * v);
    const double ratio = (omega == 0 || dist == 0) ? 1 : atan(2 * dist * tan(omega / 2)) / (dist * omega);
    px[0] = ratio * cam->fx * u + cam->cx;
    px[1] = ratio * cam->fy * v + cam->cy;
  } else {
    px[0] = cam->fx * u + cam->cx;
    px[1] = cam->fy * v + cam->cy;
  }
}

/* cv::undistortPoints (OpenCV 3.x cvUndistortPoints, 5 fixed-point iterations),
 * float camera matrix / distortion / IO as the reference passes them
 * (src/camera.cpp:43-45,78-85). */
static void undistort_points_cv(const hso_camera* cam, float u, float v, float* xo, float* yo)
{
  const double fx = (float)cam->fx, fy = (float)cam->fy, cx = (float)cam->cx, cy = (float)cam->cy;
  const double k0 = (float)cam->d[0], k1 = (float)cam->d[1], p1 = (float)cam->d[2],
               p2 = (float)cam->d[3], k2 = (float)cam->d[4];
  const double ifx = 1. / fx, ify = 1. / fy;
  double x = u, y = v;
  const double x0 = x = (x - cx) * ifx;
  const double y0 = y = (y - cy) * ify;
  for (int j = 0; j < 5; j++) {
    const double r2 = x * x + y * y;
    const double icdist = (1 + ((0 * r2 + 0) * r2 + 0) * r2) / (1 + ((k2 * r2 + k1) * r2 + k0) * r2);
    const double deltaX = 2 * p1 * x * y + p2 * (r2 + 2 * x * x);
    const double deltaY = p1 * (r2 + 2 * y * y) + 2 * p2 * x * y;
    x = (x0 - deltaX) * icdist;
    y = (y0 - deltaY) * icdist;
  }
  *xo = (float)x; *yo = (float)y;
}

/* src/camera.cpp:67-87 (Pinhole), :171-194 (FOV), :283-286 (Equidistant) */
void hso_or_cam2world(const hso_camera* cam, double u, double v, double f[3])
{
  double xyz[3];
  if (cam->model == HSO_CAM_PINHOLE && cam->distortion) {
    float px, py;
    undistort_points_cv(cam, (float)u, (float)v, &px, &py);
    xyz[0] = px; xyz[1] = py; xyz[2] = 1.0;
  } else if (cam->model == HSO_CAM_FOV && cam->distortion) {
    const double omega = cam->d[0];
    const double ud = (u - cam->cx) / cam->fx;
    const double vd = (v - cam->cy) / cam->fy;
    const double dist = sqrt(ud * ud + vd * vd);
    const double radial_distortion = tan(dist * omega) / (2 * dist * tan(omega / 2));
    xyz[0] = radial_distortion * ud; xyz[1] = radial_distortion * vd; xyz[2] = 1.0;
  } else {
    xyz[0] = (u - cam->cx) / cam->fx; xyz[1] = (v - cam->cy) / cam->fy; xyz[2] = 1.0;
  }
  /* Eigen normalized(): v / sqrt(squaredNorm) */
  const double n = sqrt(xyz[0] * xyz[0] + xyz[1] * xyz[1] + xyz[2] * xyz[2]);
  f[0] = xyz[0] / n; f[1] = xyz[1] / n; f[2] = xyz[2] / n;
}

double hso_or_error_multiplier2(const hso_camera* cam)
{
  /* src/camera.cpp:59 */
  return (cam->fx * cam->fy < 0) ? fabs(cam->fx) : fabs((cam->fx + cam->fy) * 0.5);
}
