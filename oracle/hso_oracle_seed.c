/*
 * hso_oracle_seed.c — depth-filter seed observation: epipolar search with ZMNCC, step-limited
 * KLT refinement, triangulation, uncertainty and the Gaussian inverse-depth update.
 * TEST INFRASTRUCTURE (see hso_oracle.h).  Follows DepthFilter::observeDepthRow
 * (src/depth_filter.cpp:580-675), updateSeed :527-537, computeTau :539-555,
 * Matcher::doLineStereo (src/matcher.cpp:802-1049), KLTLimited2D :1296-1451, KLTLimited1D
 * :1454-1606, warp::createPatch :159-196, depthFromTriangulation :242-255 and ZMNCC_F
 * (include/hso/vikit/patch_score.h:268-305).  Quirks kept: hso::PI = 3.14159265
 * (include/hso/global.h:104); `!eplLength > 0` parses as `(!eplLength) > 0` (:868); the 2-D KLT
 * convergence test multiplies the two step components (:1435); a(b) of the seed are only
 * incremented, never used in the fusion (:634).
 */
#include "hso_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* DepthFilter::updateSeed, src/depth_filter.cpp:527-537 */
#define UNZERO(val) (val < 0 ? (val > -1e-10 ? -1e-10 : val) : (val < 1e-10 ? 1e-10 : val))
void hso_or_update_seed(float x, float tau2, float* mu, float* sigma2)
{
  float id_var = *sigma2 * 1.01f;
  const float w = tau2 / (tau2 + id_var);
  const float new_idepth = (1 - w) * x + w * (*mu);
  *mu = UNZERO(new_idepth);
  id_var *= w;
  if (id_var < *sigma2) *sigma2 = id_var;
}

/* DepthFilter::computeTau, src/depth_filter.cpp:539-555 */
double hso_or_compute_tau(const hso_se3* T_ref_cur, const double f[3], double z, double px_error_angle)
{
  const double PI = 3.14159265; /* include/hso/global.h:104 */
  const double* t = T_ref_cur->t;
  const double a[3] = { f[0] * z - t[0], f[1] * z - t[1], f[2] * z - t[2] };
  const double t_norm = sqrt(t[0] * t[0] + t[1] * t[1] + t[2] * t[2]);
  const double a_norm = sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]);
  const double alpha = acos((f[0] * t[0] + f[1] * t[1] + f[2] * t[2]) / t_norm);
  const double beta = acos((a[0] * -t[0] + a[1] * -t[1] + a[2] * -t[2]) / (t_norm * a_norm));
  const double beta_plus = beta + px_error_angle;
  const double gamma_plus = PI - alpha - beta_plus;
  const double z_plus = t_norm * sin(beta_plus) / sin(gamma_plus);
  return (z_plus - z);
}

/* warp::createPatch, src/matcher.cpp:159-196 */
static void create_patch(float* patch, const double px_scaled[2], const uint8_t* img, int stride)
{
  const int halfpatch_size = 4, patch_size = 8;
  const float u_cur = px_scaled[0], v_cur = px_scaled[1];
  const int ui = floorf(u_cur), vi = floorf(v_cur);
  const float su = u_cur - ui, sv = v_cur - vi;
  const float w_tl = (1.0 - su) * (1.0 - sv);
  const float w_tr = su * (1.0 - sv);
  const float w_bl = (1.0 - su) * sv;
  const float w_br = 1.0 - w_tl - w_tr - w_bl;
  float* pp = patch;
  for (int y = 0; y < patch_size; ++y) {
    const uint8_t* c = img + (vi - halfpatch_size + y) * stride + (ui - halfpatch_size);
    for (int x = 0; x < patch_size; ++x, ++pp, ++c)
      *pp = w_tl * c[0] + w_tr * c[1] + w_bl * c[stride] + w_br * c[stride + 1];
  }
}

/* ZMNCC_F, include/hso/vikit/patch_score.h:268-305 */
static float zmncc_host_mean(const float* host)
{
  float m = 0;
  for (int r = 0; r < 64; r++) m += host[r];
  m /= 64;
  return m;
}
static float zmncc_score(const float* host, float hostMean, const float* target)
{
  float targetMean = 0;
  for (int r = 0; r < 64; r++) targetMean += target[r];
  targetMean /= 64;
  float numerator = 0, d1 = 0, d2 = 0;
  for (int i = 0; i < 64; i++) {
    const float h = host[i] - hostMean, t = target[i] - targetMean;
    numerator += h * t; d1 += h * h; d2 += t * t;
  }
  return (numerator / (sqrtf(d1 * d2) + 1e-12));
}

/* exported for the pinning test against the reference's own template (oracle/_ref/libpatch_score_ref.so) */
float hso_or_zmncc_f8(const float* host, const float* target) { return zmncc_score(host, zmncc_host_mean(host), target); }

/* Matcher::KLTLimited2D, src/matcher.cpp:1296-1451 */
static int klt_limited_2d(const uint8_t* img, int cols, int rows, const float* pwb, const float* host, int n_iter,
                          double px[2], float* targetPatch)
{
  float host_dx[64], host_dy[64], gw[64], H[9];
  for (int i = 0; i < 9; i++) H[i] = 0;
  int k = 0;
  for (int y = 0; y < 8; ++y) {
    const float* it = pwb + (y + 1) * 10 + 1;
    for (int x = 0; x < 8; ++x, ++it, ++k) {
      float J[3];
      J[0] = 0.5 * (it[1] - it[-1]); J[1] = 0.5 * (it[10] - it[-10]); J[2] = 1;
      host_dx[k] = J[0]; host_dy[k] = J[1];
      gw[k] = sqrtf(250.0 / (250.0 + (J[0] * J[0] + J[1] * J[1])));
      for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) H[r * 3 + c] += (J[r] * J[c]) * gw[k];
    }
  }
  for (int i = 0; i < 3; i++) H[i * 3 + i] *= (1 + 0.001);
  float Hinv[9];
  {
    const float c00 = H[4] * H[8] - H[5] * H[7], c01 = H[5] * H[6] - H[3] * H[8], c02 = H[3] * H[7] - H[4] * H[6];
    const float det = H[0] * c00 + H[1] * c01 + H[2] * c02;
    const float invdet = 1.0f / det;
    Hinv[0] = c00 * invdet; Hinv[3] = c01 * invdet; Hinv[6] = c02 * invdet;
    Hinv[1] = (H[2] * H[7] - H[1] * H[8]) * invdet; Hinv[4] = (H[0] * H[8] - H[2] * H[6]) * invdet; Hinv[7] = (H[1] * H[6] - H[0] * H[7]) * invdet;
    Hinv[2] = (H[1] * H[5] - H[2] * H[4]) * invdet; Hinv[5] = (H[2] * H[3] - H[0] * H[5]) * invdet; Hinv[8] = (H[0] * H[4] - H[1] * H[3]) * invdet;
  }
  float mean_diff = 0;
  float bestU = px[0], bestV = px[1];
  float bestEnergy = 1e8;
  float step[3] = { 0, 0, 0 }, stepBack[3] = { 0, 0, 0 }, Jres[3];
  float uBak = bestU, vBak = bestV, meanBak = mean_diff;
  for (int iter = 0; iter < n_iter; ++iter) {
    float* cp = targetPatch;
    const int u_r = (int)floor(bestU), v_r = (int)floor(bestV);
    if (u_r < 4 || v_r < 4 || u_r >= cols - 4 || v_r >= rows - 4) break;
    if (isnan(bestU) || isnan(bestV)) return 0;
    const float sx = bestU - u_r, sy = bestV - v_r;
    const float wTL = (1.0 - sx) * (1.0 - sy), wTR = sx * (1.0 - sy), wBL = (1.0 - sx) * sy, wBR = sx * sy;
    float energy = 0.0;
    Jres[0] = Jres[1] = Jres[2] = 0;
    k = 0;
    for (int y = 0; y < 8; ++y) {
      const uint8_t* it = img + (v_r + y - 4) * cols + u_r - 4;
      for (int x = 0; x < 8; ++x, ++it, ++k, ++cp) {
        const float sp = wTL * it[0] + wTR * it[1] + wBL * it[cols] + wBR * it[cols + 1];
        const float res = sp - host[k] + mean_diff;
        Jres[0] -= res * host_dx[k] * gw[k];
        Jres[1] -= res * host_dy[k] * gw[k];
        Jres[2] -= res * gw[k];
        energy += res * res * gw[k];
        *cp = sp;
      }
    }
    hso_or_margin_note(HSO_M_KLT_ACCEPT, ((double)energy - bestEnergy) / bestEnergy);
    if (energy > bestEnergy) {
      for (int i = 0; i < 3; i++) stepBack[i] *= 0.5;
      bestU = uBak + stepBack[0]; bestV = vBak + stepBack[1]; mean_diff = meanBak + stepBack[2];
    } else {
      for (int r = 0; r < 3; r++) step[r] = (Hinv[r * 3] * Jres[0] + Hinv[r * 3 + 1] * Jres[1]) + Hinv[r * 3 + 2] * Jres[2];
      if (step[0] < -0.5) step[0] = -0.5; else if (step[0] > 0.5) step[0] = 0.5;
      if (step[1] < -0.5) step[1] = -0.5; else if (step[1] > 0.5) step[1] = 0.5;
      if (!isfinite(step[0])) step[0] = step[1] = step[2] = 0;
      uBak = bestU; vBak = bestV; meanBak = mean_diff;
      for (int i = 0; i < 3; i++) stepBack[i] = step[i];
      bestU += step[0]; bestV += step[1]; mean_diff += step[2];
      bestEnergy = energy;
    }
    hso_or_margin_note(HSO_M_KLT_STEP, ((double)(stepBack[0] * stepBack[1]) - 0.01 * 0.01) / (0.01 * 0.01));
    if (stepBack[0] * stepBack[1] < 0.01 * 0.01) break;
  }
  px[0] = bestU; px[1] = bestV;
  hso_or_margin_note(HSO_M_KLT_ENERGY, ((double)bestEnergy - 650 * 64) / (650 * 64));
  if (bestEnergy > 650 * 64) return 0;
  return 1;
}

/* Matcher::KLTLimited1D, src/matcher.cpp:1454-1606 */
static int klt_limited_1d(const uint8_t* img, int cols, int rows, const float* pwb, const float* host, int n_iter,
                          double px[2], const double direct[2], float* targetPatch)
{
  float host_d[64], gw[64], H[4] = { 0, 0, 0, 0 };
  int k = 0;
  for (int y = 0; y < 8; ++y) {
    const float* it = pwb + (y + 1) * 10 + 1;
    for (int x = 0; x < 8; ++x, ++it, ++k) {
      float J[2];
      J[0] = 0.5 * (direct[0] * (it[1] - it[-1]) + direct[1] * (it[10] - it[-10]));
      J[1] = 1;
      host_d[k] = J[0];
      gw[k] = sqrtf(250.0 / (250.0 + (J[0] * J[0])));
      for (int r = 0; r < 2; r++) for (int c = 0; c < 2; c++) H[r * 2 + c] += (J[r] * J[c]) * gw[k];
    }
  }
  for (int i = 0; i < 2; i++) H[i * 2 + i] *= (1 + 0.001);
  float Hinv[4];
  {
    const float det = H[0] * H[3] - H[2] * H[1];
    const float invdet = 1.0f / det;
    Hinv[0] = H[3] * invdet; Hinv[1] = -H[1] * invdet; Hinv[2] = -H[2] * invdet; Hinv[3] = H[0] * invdet;
  }
  float mean_diff = 0;
  float bestU = px[0], bestV = px[1];
  float bestEnergy = 1e8;
  float step[2] = { 0, 0 }, stepBack[2] = { 0, 0 }, Jres[2];
  float uBak = bestU, vBak = bestV, meanBak = mean_diff;
  for (int iter = 0; iter < n_iter; ++iter) {
    float* cp = targetPatch;
    const int u_r = (int)floor(bestU), v_r = (int)floor(bestV);
    if (u_r < 4 || v_r < 4 || u_r >= cols - 4 || v_r >= rows - 4) break;
    if (isnan(bestU) || isnan(bestV)) return 0;
    const float sx = bestU - u_r, sy = bestV - v_r;
    const float wTL = (1.0 - sx) * (1.0 - sy), wTR = sx * (1.0 - sy), wBL = (1.0 - sx) * sy, wBR = sx * sy;
    float energy = 0.0;
    Jres[0] = Jres[1] = 0;
    k = 0;
    for (int y = 0; y < 8; ++y) {
      const uint8_t* it = img + (v_r + y - 4) * cols + u_r - 4;
      for (int x = 0; x < 8; ++x, ++it, ++k) {
        const float sp = wTL * it[0] + wTR * it[1] + wBL * it[cols] + wBR * it[cols + 1];
        const float res = sp - host[k] + mean_diff;
        Jres[0] -= res * host_d[k] * gw[k];
        Jres[1] -= res * gw[k];
        energy += res * res * gw[k];
        if (targetPatch != NULL) { *cp = sp; ++cp; }
      }
    }
    hso_or_margin_note(HSO_M_KLT_ACCEPT, ((double)energy - bestEnergy) / bestEnergy);
    if (energy > bestEnergy) {
      stepBack[0] *= 0.5; stepBack[1] *= 0.5;
      bestU = uBak + stepBack[0] * direct[0];
      bestV = vBak + stepBack[0] * direct[1];
      mean_diff = meanBak + stepBack[1];
    } else {
      step[0] = Hinv[0] * Jres[0] + Hinv[1] * Jres[1];
      step[1] = Hinv[2] * Jres[0] + Hinv[3] * Jres[1];
      if (step[0] < -0.5) step[0] = -0.5; else if (step[0] > 0.5) step[0] = 0.5;
      if (!isfinite(step[0])) step[0] = step[1] = 0;
      uBak = bestU; vBak = bestV; meanBak = mean_diff;
      stepBack[0] = step[0]; stepBack[1] = step[1];
      bestU += step[0] * direct[0];
      bestV += step[0] * direct[1];
      mean_diff += step[1];
      bestEnergy = energy;
    }
    hso_or_margin_note(HSO_M_KLT_STEP, ((double)fabsf(stepBack[0]) - 0.01) / 0.01);
    if (fabsf(stepBack[0]) < 0.01) break;
  }
  px[0] = bestU; px[1] = bestV;
  hso_or_margin_note(HSO_M_KLT_ENERGY, ((double)bestEnergy - 650 * 64) / (650 * 64));
  if (bestEnergy > 650 * 64) return 0;
  return 1;
}

/* depthFromTriangulation, src/matcher.cpp:242-255 */
static int depth_from_triangulation(const hso_se3* T_search_ref, const double f_ref[3], const double f_cur[3], double* depth)
{
  double R[9];
  hso_or_so3_matrix(T_search_ref->q, R);
  double a0[3];
  for (int i = 0; i < 3; i++) a0[i] = R[i * 3 + 0] * f_ref[0] + R[i * 3 + 1] * f_ref[1] + R[i * 3 + 2] * f_ref[2];
  const double* a1 = f_cur;
  const double m00 = a0[0] * a0[0] + a0[1] * a0[1] + a0[2] * a0[2];
  const double m01 = a0[0] * a1[0] + a0[1] * a1[1] + a0[2] * a1[2];
  const double m11 = a1[0] * a1[0] + a1[1] * a1[1] + a1[2] * a1[2];
  const double det = m00 * m11 - m01 * m01;
  if (det < 0.000001) return 0;
  const double invdet = 1.0 / det;
  const double i00 = m11 * invdet, i01 = -m01 * invdet;
  const double* t = T_search_ref->t;
  /* depth2 = -AtA^-1 * A^T * t, evaluated left to right: ((-AtA^-1) * A^T) * t; only row 0 is used */
  const double r0[3] = { (-i00) * a0[0] + (-i01) * a1[0], (-i00) * a0[1] + (-i01) * a1[1], (-i00) * a0[2] + (-i01) * a1[2] };
  const double d0 = r0[0] * t[0] + r0[1] * t[1] + r0[2] * t[2];
  *depth = fabs(d0);
  return 1;
}

static int is_in_frame_level(int w, int h, int ox, int oy, int boundary, int level)
{
  return ox >= boundary && ox < w / (1 << level) - boundary && oy >= boundary && oy < h / (1 << level) - boundary;
}

/* Matcher::doLineStereo, src/matcher.cpp:802-1049 */
static int do_line_stereo(const hso_camera* cam, const hso_seed* s, const hso_se3* T_cur_ref, float exposure_rat,
                          const uint8_t* const ref_pyr[HSO_N_PYR_LEVELS], const uint8_t* const cur_pyr[HSO_N_PYR_LEVELS],
                          const int16_t* const cur_gx[HSO_N_SOBEL_LEVELS], const int16_t* const cur_gy[HSO_N_SOBEL_LEVELS],
                          int w, int h, double min_idepth, double prior_idepth, double max_idepth, hso_seed_out* o)
{
  double A[4];
  hso_or_warp_matrix_affine(cam, cam, s->px, s->f, prior_idepth, T_cur_ref, s->level, A);
  const int search_level = hso_or_best_search_level(A, HSO_N_SOBEL_LEVELS - 1);
  o->search_level = search_level;
  float pwb[100], patch[64];
  int rcols, rrows;  /* img_pyr_[level].cols / rows */
  hso_or_pyramid_dims(w, h, s->level, &rcols, &rrows);
  hso_or_warp_affine(A, ref_pyr[s->level], rcols, rrows, s->px, s->level, search_level, 5, pwb);
  if (fabsf(exposure_rat * 128 - 128) > 30.0f)
    for (int i = 0; i < 100; ++i) pwb[i] = pwb[i] * exposure_rat;
  for (int y = 1; y < 9; ++y) for (int x = 0; x < 8; ++x) patch[(y - 1) * 8 + x] = pwb[y * 10 + 1 + x];

  double pc[3], pf[3];
  { const double v[3] = { s->f[0] * min_idepth, s->f[1] * min_idepth, s->f[2] * min_idepth }; hso_or_se3_apply(T_cur_ref, v, pc); }
  { const double z = pc[2]; pc[0] /= z; pc[1] /= z; pc[2] /= z; }
  { const double v[3] = { s->f[0] * max_idepth, s->f[1] * max_idepth, s->f[2] * max_idepth }; hso_or_se3_apply(T_cur_ref, v, pf); }
  if (pf[2] < 0.001 || max_idepth < min_idepth) return -1;
  { const double z = pf[2]; pf[0] /= z; pf[1] /= z; pf[2] /= z; }
  if (isnan((float)(pf[0] + pc[0]))) return -1;
  double px_close[2], px_far[2];
  { const double v[3] = { pc[0], pc[1], 1.0 }; hso_or_world2cam(cam, v, px_close); }
  o->epl_start[0] = (int)px_close[0]; o->epl_start[1] = (int)px_close[1];
  px_close[0] /= (1 << search_level); px_close[1] /= (1 << search_level);
  { const double v[3] = { pf[0], pf[1], 1.0 }; hso_or_world2cam(cam, v, px_far); }
  o->epl_end[0] = (int)px_far[0]; o->epl_end[1] = (int)px_far[1];
  px_far[0] /= (1 << search_level); px_far[1] /= (1 << search_level);
  double incx = px_close[0] - px_far[0], incy = px_close[1] - px_far[1];
  const double eplLength = sqrt(incx * incx + incy * incy);
  if (((!eplLength) > 0) || isinf(eplLength)) return -1;
  if (eplLength > 100.0) {
    px_close[0] = px_far[0] + incx * 100.0 / eplLength;
    px_close[1] = px_far[1] + incy * 100.0 / eplLength;
  }
  incx *= 1.0 / eplLength; incy *= 1.0 / eplLength;
  px_far[0] -= incx; px_far[1] -= incy; px_close[0] += incx; px_close[1] += incy;
  if (eplLength < (2.0)) {
    const double pad = ((2.0) - (eplLength)) / 2.0f;
    px_far[0] -= incx * pad; px_far[1] -= incy * pad;
    px_close[0] += incx * pad; px_close[1] += incy * pad;
  }
  if (s->type == HSO_FTR_GRADIENT || s->type == HSO_FTR_EDGELET) {
    double g0 = A[0] * s->grad[0] + A[1] * s->grad[1], g1 = A[2] * s->grad[0] + A[3] * s->grad[1];
    const double gn = sqrt(g0 * g0 + g1 * g1); g0 /= gn; g1 /= gn;
    double e0 = px_close[0] - px_far[0], e1 = px_close[1] - px_far[1];
    const double en = sqrt(e0 * e0 + e1 * e1); e0 /= en; e1 /= en;
    const double cosangle = fabs(g0 * e0 + g1 * e1);
    if (cosangle < 0.4) return -1;
  }
  double cpx = px_far[0], cpy = px_far[1];
  const float hostMean = zmncc_host_mean(patch);
  float zmncc_best = 0.1, zmncc_second = zmncc_best;
  double uv_best[2] = { 0, 0 };
  float patch_f[64];
  int loopCounter = 0, loopCBest = -1, loopCSecond = -1;
  int cols, rows;
  hso_or_pyramid_dims(w, h, search_level, &cols, &rows);
  while ((((incx < 0) == (cpx > px_close[0])) && ((incy < 0) == (cpy > px_close[1]))) || loopCounter == 0) {
    const double px[2] = { cpx, cpy };
    if (incx != 0) hso_or_margin_note(HSO_M_MARCH_END, cpx + incx - px_close[0]);   /* what the next test of the loop condition compares */
    if (incy != 0) hso_or_margin_note(HSO_M_MARCH_END, cpy + incy - px_close[1]);
    if (!is_in_frame_level(w, h, (int)px[0], (int)px[1], 8, search_level)) { cpx += incx; cpy += incy; loopCounter++; continue; }
    create_patch(patch_f, px, cur_pyr[search_level], cols);
    const float zmncc = zmncc_score(patch, hostMean, patch_f);
    hso_or_margin_note(HSO_M_ZMNCC_ORDER, (double)zmncc - zmncc_best);
    hso_or_margin_note(HSO_M_ZMNCC_ORDER, (double)zmncc - zmncc_second);
    if (zmncc > zmncc_best) {
      zmncc_second = zmncc_best;
      uv_best[0] = px[0]; uv_best[1] = px[1];
      zmncc_best = zmncc;
      loopCSecond = loopCBest; loopCBest = loopCounter;
    } else if (zmncc > zmncc_second) {
      zmncc_second = zmncc; loopCSecond = loopCounter;
    }
    cpx += incx; cpy += incy; loopCounter++;
  }
  o->n_steps = loopCounter;
  o->zmncc_best = zmncc_best; o->zmncc_second = zmncc_second;
  if (abs(loopCBest - loopCSecond) > 1.0f) hso_or_margin_note(HSO_M_ZMNCC_AMBIG, 1.5 * (double)zmncc_second - zmncc_best);
  if (abs(loopCBest - loopCSecond) > 1.0f && 1.5f * zmncc_second > zmncc_best) return -4;
  hso_or_margin_note(HSO_M_ZMNCC_BEST, (double)zmncc_best - 0.8);
  if (zmncc_best > 0.8) {
    double px_cur[2] = { uv_best[0] * (1 << search_level), uv_best[1] * (1 << search_level) };
    double px_scaled[2] = { px_cur[0] / (1 << search_level), px_cur[1] / (1 << search_level) };
    double ed[2] = { px_close[0] - px_far[0], px_close[1] - px_far[1] };
    { const double en = sqrt(ed[0] * ed[0] + ed[1] * ed[1]); ed[0] /= en; ed[1] /= en; }
    int result = klt_limited_1d(cur_pyr[search_level], cols, rows, pwb, patch, 10, px_scaled, ed, NULL);
    float patch2D[64];
    memset(patch2D, 0, sizeof(patch2D)); /* uninitialised in the reference when the KLT loop exits before sampling */
    double dir_cur[2] = { A[0] * s->grad[0] + A[1] * s->grad[1], A[2] * s->grad[0] + A[3] * s->grad[1] };
    { const double dn = sqrt(dir_cur[0] * dir_cur[0] + dir_cur[1] * dir_cur[1]); dir_cur[0] /= dn; dir_cur[1] /= dn; }
    double* pxr = px_scaled;
    double px_2d[2] = { px_cur[0] / (1 << search_level), px_cur[1] / (1 << search_level) };
    if (!result) pxr = px_2d;
    if (s->type != HSO_FTR_EDGELET) {
      result = klt_limited_2d(cur_pyr[search_level], cols, rows, pwb, patch, 10, pxr, patch2D);
    } else {
      result = klt_limited_1d(cur_pyr[search_level], cols, rows, pwb, patch, 10, pxr, dir_cur, patch2D);
      if (result) {
        const double nd_ = hso_or_normal_dot(cur_gx[search_level], cur_gy[search_level], cols, rows, pxr, dir_cur);
        hso_or_margin_note(HSO_M_NORMAL, nd_ - (float)0.7);
        result = nd_ > (float)0.7;
      }
    }
    px_scaled[0] = pxr[0]; px_scaled[1] = pxr[1];
    if (result) {
      const double ncc_ = hso_or_ncc(patch, patch2D);
      hso_or_margin_note(HSO_M_NCC, ncc_ - (double)(float)0.8);
      result = ncc_ > (double)(float)0.8;
    }
    if (result) {
      px_cur[0] = px_scaled[0] * (1 << search_level); px_cur[1] = px_scaled[1] * (1 << search_level);
      o->px_cur[0] = px_cur[0]; o->px_cur[1] = px_cur[1];
      double fc[3];
      hso_or_cam2world(cam, px_cur[0], px_cur[1], fc);
      if (depth_from_triangulation(T_cur_ref, s->f, fc, &o->z)) return 1;
      return -2;
    }
    return -3;
  }
  return -4;
}

/* DepthFilter::observeDepthRow for one seed, src/depth_filter.cpp:589-673 */
void hso_or_seed_observe(const hso_camera* cam, const hso_seed* s, const hso_se3* cur_T_f_w, double cur_exposure,
                         double px_error_angle, const uint8_t* const ref_pyr[HSO_N_PYR_LEVELS],
                         const uint8_t* const cur_pyr[HSO_N_PYR_LEVELS], const int16_t* const cur_gx[HSO_N_SOBEL_LEVELS],
                         const int16_t* const cur_gy[HSO_N_SOBEL_LEVELS], int w, int h, hso_seed_out* o)
{
  memset(o, 0, sizeof(*o));
  o->mu = s->mu; o->sigma2 = s->sigma2; o->b = s->b; o->is_valid = 1;
  hso_se3 cur_inv, T_ref_cur, T_cur_ref;
  hso_or_se3_inverse(cur_T_f_w, &cur_inv);
  hso_or_se3_mul(&s->T_ref_w, &cur_inv, &T_ref_cur);
  hso_or_se3_inverse(&T_ref_cur, &T_cur_ref);
  const double sc = 1.0 / s->mu;
  const double pr[3] = { sc * s->f[0], sc * s->f[1], sc * s->f[2] };
  double xyz_f[3];
  hso_or_se3_apply(&T_cur_ref, pr, xyz_f);
  if (xyz_f[2] < 0.0) { o->result = 0; o->is_update = 0; return; }
  double c[2];
  hso_or_world2cam(cam, xyz_f, c);
  { const int ox = (int)c[0], oy = (int)c[1];
    if (!(ox >= 0 && ox < w && oy >= 0 && oy < h)) { o->result = 0; o->is_update = 0; return; } }
  o->is_update = 1;
  /* sqrt(float) resolves to the float overload in the reference (std namespace in scope) */
  const float z_inv_min = s->mu + 2 * sqrtf(s->sigma2);
  const float z_inv_max = fmaxf(s->mu - 2 * sqrtf(s->sigma2), 0.00000001f);
  if (isnan(z_inv_min)) o->is_valid = 0;
  /* doLineStereo recomputes T_cur_ref = cur.T_f_w_ * ref.T_f_w_.inverse() (matcher.cpp:807) */
  hso_se3 ref_inv, T_cr;
  hso_or_se3_inverse(&s->T_ref_w, &ref_inv);
  hso_or_se3_mul(cur_T_f_w, &ref_inv, &T_cr);
  const float exposure_rat = cur_exposure / s->ref_exposure;
  const int res = do_line_stereo(cam, s, &T_cr, exposure_rat, ref_pyr, cur_pyr, cur_gx, cur_gy, w, h,
                                 1.0 / z_inv_min, 1.0 / s->mu, 1.0 / z_inv_max, o);
  o->result = res;
  if (res != 1) {
    o->b = s->b + 1;
    o->epl_start[0] = o->epl_start[1] = o->epl_end[0] = o->epl_end[1] = 0;
    return;
  }
  const double z = o->z;
  const double tau = hso_or_compute_tau(&T_ref_cur, s->f, z, px_error_angle);
  const double tau_inverse = 0.5 * (1.0 / fmax(0.0000001, z - tau) - 1.0 / (z + tau));
  hso_or_update_seed(1. / z, tau_inverse * tau_inverse, &o->mu, &o->sigma2);
}

/* Matcher::findEpipolarMatchPrevious, src/matcher.cpp:1051-1293 (cur_frame = the earlier frame).  Returns 1 and the depth, or
 * the reason for `false`: -1 the edgelet / epipolar angle filter (:1078-1084) or too many steps (:1161), -4 the march (ambiguous
 * or best score <= 0.8), -3 the refinement, -2 the triangulation. */
static int find_epipolar_match_previous(const hso_camera* cam, const hso_seed* s, const hso_se3* T_cur_ref, float exposure_rat,
                                        const uint8_t* const ref_pyr[HSO_N_PYR_LEVELS], const uint8_t* const cur_pyr[HSO_N_PYR_LEVELS],
                                        const int16_t* const cur_gx[HSO_N_SOBEL_LEVELS], const int16_t* const cur_gy[HSO_N_SOBEL_LEVELS],
                                        int w, int h, double d_estimate, double d_min, double d_max, hso_seed_out* o)
{
  double A2[2], B2[2];
  { const double v[3] = { s->f[0] * d_min, s->f[1] * d_min, s->f[2] * d_min }; double q[3]; hso_or_se3_apply(T_cur_ref, v, q); A2[0] = q[0] / q[2]; A2[1] = q[1] / q[2]; }
  { const double v[3] = { s->f[0] * d_max, s->f[1] * d_max, s->f[2] * d_max }; double q[3]; hso_or_se3_apply(T_cur_ref, v, q); B2[0] = q[0] / q[2]; B2[1] = q[1] / q[2]; }
  const double epi_dir[2] = { A2[0] - B2[0], A2[1] - B2[1] };
  double A[4];
  hso_or_warp_matrix_affine(cam, cam, s->px, s->f, d_estimate, T_cur_ref, s->level, A);
  const int search_level = hso_or_best_search_level(A, HSO_N_SOBEL_LEVELS - 1);
  o->search_level = search_level;
  double px_A[2], px_B[2];
  { const double v[3] = { A2[0], A2[1], 1.0 }; hso_or_world2cam(cam, v, px_A); }
  { const double v[3] = { B2[0], B2[1], 1.0 }; hso_or_world2cam(cam, v, px_B); }
  const double dAB[2] = { px_A[0] - px_B[0], px_A[1] - px_B[1] };
  const double epi_length = sqrt(dAB[0] * dAB[0] + dAB[1] * dAB[1]) / (1 << search_level);
  float pwb[100], patch[64];
  int rcols, rrows;
  hso_or_pyramid_dims(w, h, s->level, &rcols, &rrows);
  hso_or_warp_affine(A, ref_pyr[s->level], rcols, rrows, s->px, s->level, search_level, 5, pwb);
  if (s->type == HSO_FTR_GRADIENT || s->type == HSO_FTR_EDGELET) {
    double g0 = A[0] * s->grad[0] + A[1] * s->grad[1], g1 = A[2] * s->grad[0] + A[3] * s->grad[1];
    const double gn = sqrt(g0 * g0 + g1 * g1); g0 /= gn; g1 /= gn;
    const double en = sqrt(epi_dir[0] * epi_dir[0] + epi_dir[1] * epi_dir[1]);
    const double cosangle = fabs(g0 * (epi_dir[0] / en) + g1 * (epi_dir[1] / en));
    if (cosangle < 0.4) return -1;
  }
  if (fabsf(exposure_rat * 128 - 128) > 30.0f)
    for (int i = 0; i < 100; ++i) pwb[i] = pwb[i] * exposure_rat;
  for (int y = 1; y < 9; ++y) for (int x = 0; x < 8; ++x) patch[(y - 1) * 8 + x] = pwb[y * 10 + 1 + x];
  int cols, rows;
  hso_or_pyramid_dims(w, h, search_level, &cols, &rows);
  double px_cur[2];
  if (!(epi_length < 2.0)) {
    /* size_t n_steps = epi_length_ / 0.7; if (n_steps > options_.max_epi_search_steps (100, matcher.h:126)) return false.  A NaN
     * length is undefined in that conversion; the reference's x86 build yields 2^63 (cvttsd2si) and rejects: said explicitly */
    if (!(epi_length / 0.7 < 101.0)) return -1;
    size_t n_steps = epi_length / 0.7;
    const double step[2] = { epi_dir[0] / n_steps, epi_dir[1] / n_steps };
    const float hostMean = zmncc_host_mean(patch);
    float zmncc_best = 0.1f, zmncc_second = zmncc_best;
    size_t bestCounter = 0, secondCounter = 0;
    double uv_best[2] = { 0, 0 };
    double uv[2] = { B2[0] - step[0], B2[1] - step[1] };
    ++n_steps;
    float patch_f[64];
    for (size_t i = 0; i < n_steps; ++i, uv[0] += step[0], uv[1] += step[1]) {
      double px[2];
      { const double v[3] = { uv[0], uv[1], 1.0 }; hso_or_world2cam(cam, v, px); }
      const double px_scaled[2] = { px[0] / (1 << search_level), px[1] / (1 << search_level) };
      if (!is_in_frame_level(w, h, (int)px_scaled[0], (int)px_scaled[1], 8, search_level)) continue;
      create_patch(patch_f, px_scaled, cur_pyr[search_level], cols);
      const float zmncc = zmncc_score(patch, hostMean, patch_f);
      hso_or_margin_note(HSO_M_ZMNCC_ORDER, (double)zmncc - zmncc_best);
      hso_or_margin_note(HSO_M_ZMNCC_ORDER, (double)zmncc - zmncc_second);
      if (zmncc > zmncc_best) {
        zmncc_second = zmncc_best; secondCounter = bestCounter;
        zmncc_best = zmncc; bestCounter = i;
        uv_best[0] = uv[0]; uv_best[1] = uv[1];
      } else if (zmncc > zmncc_second) {
        zmncc_second = zmncc; secondCounter = i;
      }
    }
    o->n_steps = (int)n_steps; o->zmncc_best = zmncc_best; o->zmncc_second = zmncc_second;
    /* fabs(bestCounter - secondCounter) on size_t: the difference wraps, so only "second == best" and "second == best - 1"
     * count as adjacent (:1219) */
    const int apart = fabs((double)(size_t)(bestCounter - secondCounter)) > 1.0f;
    if (apart) hso_or_margin_note(HSO_M_ZMNCC_AMBIG, 1.5 * (double)zmncc_second - zmncc_best);
    if (apart && 1.5f * zmncc_second > zmncc_best) return -4;
    hso_or_margin_note(HSO_M_ZMNCC_BEST, (double)zmncc_best - 0.8);
    if (!(zmncc_best > 0.8)) return -4;
    { const double v[3] = { uv_best[0], uv_best[1], 1.0 }; hso_or_world2cam(cam, v, px_cur); }
  } else {
    px_cur[0] = (px_A[0] + px_B[0]) / 2.0; px_cur[1] = (px_A[1] + px_B[1]) / 2.0;
  }
  /* the refinement both branches share (:1102-1150, :1236-1290) */
  double px_scaled[2] = { px_cur[0] / (1 << search_level), px_cur[1] / (1 << search_level) };
  double ed[2] = { dAB[0], dAB[1] };
  { const double en = sqrt(ed[0] * ed[0] + ed[1] * ed[1]); ed[0] /= en; ed[1] /= en; }
  int result = klt_limited_1d(cur_pyr[search_level], cols, rows, pwb, patch, 10, px_scaled, ed, NULL);
  float patch2D[64];
  memset(patch2D, 0, sizeof(patch2D));
  double dir_cur[2] = { A[0] * s->grad[0] + A[1] * s->grad[1], A[2] * s->grad[0] + A[3] * s->grad[1] };
  { const double dn = sqrt(dir_cur[0] * dir_cur[0] + dir_cur[1] * dir_cur[1]); dir_cur[0] /= dn; dir_cur[1] /= dn; }
  double* pxr = px_scaled;
  double px_2d[2] = { px_cur[0] / (1 << search_level), px_cur[1] / (1 << search_level) };
  if (!result) pxr = px_2d;
  if (s->type != HSO_FTR_EDGELET) {
    result = klt_limited_2d(cur_pyr[search_level], cols, rows, pwb, patch, 10, pxr, patch2D);
  } else {
    result = klt_limited_1d(cur_pyr[search_level], cols, rows, pwb, patch, 10, pxr, dir_cur, patch2D);
    if (result) {
      const double nd_ = hso_or_normal_dot(cur_gx[search_level], cur_gy[search_level], cols, rows, pxr, dir_cur);
      hso_or_margin_note(HSO_M_NORMAL, nd_ - (float)0.7);
      result = nd_ > (float)0.7;
    }
  }
  px_scaled[0] = pxr[0]; px_scaled[1] = pxr[1];
  if (result) {
    const double ncc_ = hso_or_ncc(patch, patch2D);
    hso_or_margin_note(HSO_M_NCC, ncc_ - (double)(float)0.8);
    result = ncc_ > (double)(float)0.8;
  }
  if (!result) return -3;
  px_cur[0] = px_scaled[0] * (1 << search_level); px_cur[1] = px_scaled[1] * (1 << search_level);
  o->px_cur[0] = px_cur[0]; o->px_cur[1] = px_cur[1];
  double fc[3];
  hso_or_cam2world(cam, px_cur[0], px_cur[1], fc);
  if (depth_from_triangulation(T_cur_ref, s->f, fc, &o->z)) return 1;
  return -2;
}

/* DepthFilter::observeDepthWithPreviousFrameOnce for one seed and the first of its pre_frames, src/depth_filter.cpp:677-726.
 * o->is_update = 1: the earlier frame saw the seed's point (it joins optFrames_P, :702-703); o->result = 1: matched, mu / sigma2
 * updated (:718-722).  The seed's b is never touched here. */
void hso_or_seed_observe_previous(const hso_camera* cam, const hso_seed* s, const hso_se3* pre_T_f_w, double pre_exposure,
                                  double px_error_angle, const uint8_t* const ref_pyr[HSO_N_PYR_LEVELS],
                                  const uint8_t* const pre_pyr[HSO_N_PYR_LEVELS], const int16_t* const pre_gx[HSO_N_SOBEL_LEVELS],
                                  const int16_t* const pre_gy[HSO_N_SOBEL_LEVELS], int w, int h, hso_seed_out* o)
{
  memset(o, 0, sizeof(*o));
  o->mu = s->mu; o->sigma2 = s->sigma2; o->b = s->b; o->is_valid = 1;
  hso_se3 pre_inv, T_ref_cur, T_cur_ref;
  hso_or_se3_inverse(pre_T_f_w, &pre_inv);
  hso_or_se3_mul(&s->T_ref_w, &pre_inv, &T_ref_cur);
  hso_or_se3_inverse(&T_ref_cur, &T_cur_ref);
  const double sc = 1.0 / s->mu;
  const double pr[3] = { sc * s->f[0], sc * s->f[1], sc * s->f[2] };
  double xyz_f[3];
  hso_or_se3_apply(&T_cur_ref, pr, xyz_f);
  if (xyz_f[2] < 0.0) return;
  double c[2];
  hso_or_world2cam(cam, xyz_f, c);
  { const int ox = (int)c[0], oy = (int)c[1];
    if (!(ox >= 0 && ox < w && oy >= 0 && oy < h)) return; }
  o->is_update = 1;
  const float z_inv_min = s->mu + 2 * sqrtf(s->sigma2);
  const float z_inv_max = fmaxf(s->mu - 2 * sqrtf(s->sigma2), 0.00000001f);
  /* findEpipolarMatchPrevious recomputes T_cur_ref = cur.T_f_w_ * ref.T_f_w_.inverse() (matcher.cpp:1055) */
  hso_se3 ref_inv, T_cr;
  hso_or_se3_inverse(&s->T_ref_w, &ref_inv);
  hso_or_se3_mul(pre_T_f_w, &ref_inv, &T_cr);
  const float exposure_rat = pre_exposure / s->ref_exposure;
  const int res = find_epipolar_match_previous(cam, s, &T_cr, exposure_rat, ref_pyr, pre_pyr, pre_gx, pre_gy, w, h,
                                               1.0 / s->mu, 1.0 / z_inv_min, 1.0 / z_inv_max, o);
  o->result = res;
  if (res != 1) return;
  const double z = o->z;
  const double tau = hso_or_compute_tau(&T_ref_cur, s->f, z, px_error_angle);
  const double tau_inverse = 0.5 * (1.0 / fmax(0.0000001, z - tau) - 1.0 / (z + tau));
  hso_or_update_seed(1. / z, tau_inverse * tau_inverse, &o->mu, &o->sigma2);
}
