/*
 * hso_oracle_align.c — affine warp, 8x8 Lucas-Kanade alignment (2-D and 1-D), NCC and
 * edgelet-normal gates: Matcher::findMatchDirect restated.  TEST INFRASTRUCTURE (see
 * hso_oracle.h).  Follows src/matcher.cpp:46-155,226-238,270-440,
 * src/feature_alignment.cpp:164-308,464-605 and include/hso/vikit/vision.h:49-65 of the
 * reference, with the serial float accumulation order of the originals.  Eigen's fixed-size
 * inverse() (absent dependency) is restated: 2x2 = adjugate * (1/det); 3x3 = cofactors,
 * det from the first column of cofactors, * (1/det) (Eigen/src/LU/InverseImpl.h).
 */
#include "hso_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* include/hso/vikit/vision.h:49-65 */
static float interpolate_mat_8u(const uint8_t* data, int stride, float u, float v)
{
  const int x = (int)floor(u);
  const int y = (int)floor(v);
  const float subpix_x = u - x;
  const float subpix_y = v - y;
  const float w00 = (1.0f - subpix_x) * (1.0f - subpix_y);
  const float w01 = (1.0f - subpix_x) * subpix_y;
  const float w10 = subpix_x * (1.0f - subpix_y);
  const float w11 = 1.0f - w00 - w01 - w10;
  const uint8_t* ptr = data + y * stride + x;
  return w00 * ptr[0] + w01 * ptr[stride] + w10 * ptr[1] + w11 * ptr[stride + 1];
}

/* warp::getWarpMatrixAffine, src/matcher.cpp:46-72 (A row-major 2x2) */
void hso_or_warp_matrix_affine(const hso_camera* cam_ref, const hso_camera* cam_cur, const double px_ref[2],
                               const double f_ref[3], double depth_ref, const hso_se3* T_cur_ref, int level_ref,
                               double A[4])
{
  const int halfpatch_size = 5;
  const double xyz_ref[3] = { f_ref[0] * depth_ref, f_ref[1] * depth_ref, f_ref[2] * depth_ref };
  const int ratio = (1 << level_ref);
  double du[3], dv[3];
  hso_or_cam2world(cam_ref, px_ref[0] + (double)(halfpatch_size * ratio), px_ref[1] + (double)(0 * ratio), du);
  hso_or_cam2world(cam_ref, px_ref[0] + (double)(0 * ratio), px_ref[1] + (double)(halfpatch_size * ratio), dv);
  const double su = xyz_ref[2] / du[2], sv = xyz_ref[2] / dv[2];
  for (int i = 0; i < 3; i++) { du[i] *= su; dv[i] *= sv; }
  double pc[3], pu[3], pv[3], px_cur[2], px_du[2], px_dv[2];
  hso_or_se3_apply(T_cur_ref, xyz_ref, pc);
  hso_or_se3_apply(T_cur_ref, du, pu);
  hso_or_se3_apply(T_cur_ref, dv, pv);
  hso_or_world2cam(cam_cur, pc, px_cur);
  hso_or_world2cam(cam_cur, pu, px_du);
  hso_or_world2cam(cam_cur, pv, px_dv);
  A[0] = (px_du[0] - px_cur[0]) / halfpatch_size; A[2] = (px_du[1] - px_cur[1]) / halfpatch_size;
  A[1] = (px_dv[0] - px_cur[0]) / halfpatch_size; A[3] = (px_dv[1] - px_cur[1]) / halfpatch_size;
}

/* warp::getBestSearchLevel, src/matcher.cpp:74-85 */
int hso_or_best_search_level(const double A[4], int max_level)
{
  int search_level = 0;
  double D = A[0] * A[3] - A[2] * A[1];  /* Eigen 2x2 determinant: m00*m11 - m10*m01 */
  while (D > 3.0 && search_level < max_level) { search_level += 1; D *= 0.25; }
  return search_level;
}

/* warp::warpAffine (float), src/matcher.cpp:120-155.  Returns 0 if the warp is NaN: the
 * reference then leaves the (uninitialised) patch untouched; here it is defined as all zero. */
int hso_or_warp_affine(const double A_cur_ref[4], const uint8_t* img_ref, int cols, int rows, const double px_ref[2],
                       int level_ref, int search_level, int halfpatch_size, float* patch)
{
  const int patch_size = halfpatch_size * 2;
  const double det = A_cur_ref[0] * A_cur_ref[3] - A_cur_ref[2] * A_cur_ref[1];
  const double invdet = 1.0 / det;
  const float a00 = (float)(A_cur_ref[3] * invdet), a01 = (float)(-A_cur_ref[1] * invdet);
  const float a10 = (float)(-A_cur_ref[2] * invdet), a11 = (float)(A_cur_ref[0] * invdet);
  if (isnan(a00)) { memset(patch, 0, sizeof(float) * (size_t)patch_size * patch_size); return 0; }
  float* patch_ptr = patch;
  const float rx = (float)(px_ref[0] / (1 << level_ref)), ry = (float)(px_ref[1] / (1 << level_ref));
  const float scaleTarget = (float)(1 << search_level);
  for (int y = 0; y < patch_size; ++y)
    for (int x = 0; x < patch_size; ++x, ++patch_ptr) {
      float p0 = (float)(x - halfpatch_size), p1 = (float)(y - halfpatch_size);
      p0 *= scaleTarget; p1 *= scaleTarget;
      const float px0 = (a00 * p0 + a01 * p1) + rx;
      const float px1 = (a10 * p0 + a11 * p1) + ry;
      if (px0 < 0 || px1 < 0 || px0 >= cols - 1 || px1 >= rows - 1) *patch_ptr = 0;
      else *patch_ptr = interpolate_mat_8u(img_ref, cols, px0, px1);
    }
  return 1;
}

/* feature_alignment::align2D (float), src/feature_alignment.cpp:464-605 */
int hso_or_align2d(const uint8_t* cur_img, int cols, int rows, const float* ref_patch_with_border, const float* ref_patch,
                   int n_iter, double cur_px_estimate[2], float* cur_patch, int* iters_out, float* chi2_out)
{
  const int halfpatch_size_ = 4, patch_size_ = 8, patch_area_ = 64;
  int converged = 0;
  float ref_patch_dx[64], ref_patch_dy[64], grad_weight[64];
  float H[9];
  for (int i = 0; i < 9; i++) H[i] = 0;
  const int ref_step = patch_size_ + 2;
  int k = 0;
  for (int y = 0; y < patch_size_; ++y) {
    const float* it = ref_patch_with_border + (y + 1) * ref_step + 1;
    for (int x = 0; x < patch_size_; ++x, ++it, ++k) {
      float J[3];
      J[0] = 0.5 * (it[1] - it[-1]);
      J[1] = 0.5 * (it[ref_step] - it[-ref_step]);
      J[2] = 1.;
      ref_patch_dx[k] = J[0];
      ref_patch_dy[k] = J[1];
      grad_weight[k] = sqrtf(250.0 / (250.0 + (J[0] * J[0] + J[1] * J[1])));
      for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) H[r * 3 + c] += (J[r] * J[c]) * grad_weight[k];
    }
  }
  for (int i = 0; i < 3; i++) H[i * 3 + i] *= (1 + 0.001);
  /* Matrix3f::inverse(): cofactor formula */
  float Hinv[9];
  {
    const float c00 = H[4] * H[8] - H[5] * H[7], c01 = H[5] * H[6] - H[3] * H[8], c02 = H[3] * H[7] - H[4] * H[6];
    const float det = H[0] * c00 + H[1] * c01 + H[2] * c02;
    const float invdet = 1.0f / det;
    Hinv[0] = c00 * invdet; Hinv[3] = c01 * invdet; Hinv[6] = c02 * invdet;
    Hinv[1] = (H[2] * H[7] - H[1] * H[8]) * invdet; Hinv[4] = (H[0] * H[8] - H[2] * H[6]) * invdet; Hinv[7] = (H[1] * H[6] - H[0] * H[7]) * invdet;
    Hinv[2] = (H[1] * H[5] - H[2] * H[4]) * invdet; Hinv[5] = (H[2] * H[3] - H[0] * H[5]) * invdet; Hinv[8] = (H[0] * H[4] - H[1] * H[3]) * invdet;
  }
  float u = cur_px_estimate[0];
  float v = cur_px_estimate[1];
  const float min_update_squared = 0.03 * 0.03;
  const int cur_step = cols;
  float mean_diff = 0, chi2 = 0;
  float update[3] = { 0, 0, 0 }, Jres[3];
  int iter = 0;
  for (iter = 0; iter < n_iter; ++iter) {
    float* cur_patch_ptr = cur_patch;
    const int u_r = (int)floor(u);
    const int v_r = (int)floor(v);
    if (u_r < halfpatch_size_ || v_r < halfpatch_size_ || u_r >= cols - halfpatch_size_ || v_r >= rows - halfpatch_size_) break;
    if (isnan(u) || isnan(v)) { if (iters_out) *iters_out = iter; if (chi2_out) *chi2_out = chi2; return 0; }
    const float subpix_x = u - u_r, subpix_y = v - v_r;
    const float wTL = (1.0 - subpix_x) * (1.0 - subpix_y);
    const float wTR = subpix_x * (1.0 - subpix_y);
    const float wBL = (1.0 - subpix_x) * subpix_y;
    const float wBR = subpix_x * subpix_y;
    float new_chi2 = 0.0;
    Jres[0] = Jres[1] = Jres[2] = 0;
    k = 0;
    for (int y = 0; y < patch_size_; ++y) {
      const uint8_t* it = cur_img + (v_r + y - halfpatch_size_) * cur_step + u_r - halfpatch_size_;
      for (int x = 0; x < patch_size_; ++x, ++it, ++k) {
        const float search_pixel = wTL * it[0] + wTR * it[1] + wBL * it[cur_step] + wBR * it[cur_step + 1];
        const float res = search_pixel - ref_patch[k] + mean_diff;
        Jres[0] -= res * ref_patch_dx[k] * grad_weight[k];
        Jres[1] -= res * ref_patch_dy[k] * grad_weight[k];
        Jres[2] -= res * grad_weight[k];
        new_chi2 += res * res * grad_weight[k];
        if (cur_patch != NULL) { *cur_patch_ptr = search_pixel; ++cur_patch_ptr; }
      }
    }
    chi2 = new_chi2;
    for (int r = 0; r < 3; r++) update[r] = (Hinv[r * 3] * Jres[0] + Hinv[r * 3 + 1] * Jres[1]) + Hinv[r * 3 + 2] * Jres[2];
    u += update[0];
    v += update[1];
    mean_diff += update[2];
    hso_or_margin_note(HSO_M_LK_UPDATE, ((double)(update[0] * update[0] + update[1] * update[1]) - min_update_squared) / min_update_squared);
    if (update[0] * update[0] + update[1] * update[1] < min_update_squared) { converged = 1; iter++; break; }
  }
  hso_or_margin_note(HSO_M_LK_CHI2, ((double)chi2 - 1000 * patch_area_) / (1000 * patch_area_));
  if (chi2 > 1000 * patch_area_) converged = 0;
  cur_px_estimate[0] = u; cur_px_estimate[1] = v;
  if (iters_out) *iters_out = iter;
  if (chi2_out) *chi2_out = chi2;
  return converged;
}

/* feature_alignment::align1D (float), src/feature_alignment.cpp:164-308 */
int hso_or_align1d(const uint8_t* cur_img, int cols, int rows, const float dir[2], const float* ref_patch_with_border,
                   const float* ref_patch, int n_iter, double cur_px_estimate[2], double* h_inv, float* cur_patch,
                   int* iters_out, float* chi2_out)
{
  const int halfpatch_size_ = 4, patch_size = 8, patch_area = 64;
  int converged = 0;
  float ref_patch_dv[64], grad_weight[64];
  float H[4] = { 0, 0, 0, 0 };
  const int ref_step = patch_size + 2;
  int k = 0;
  for (int y = 0; y < patch_size; ++y) {
    const float* it = ref_patch_with_border + (y + 1) * ref_step + 1;
    for (int x = 0; x < patch_size; ++x, ++it, ++k) {
      float J[2];
      J[0] = 0.5 * (dir[0] * (it[1] - it[-1]) + dir[1] * (it[ref_step] - it[-ref_step]));
      J[1] = 1.;
      ref_patch_dv[k] = J[0];
      grad_weight[k] = sqrtf(250.0 / (250.0 + J[0] * J[0]));
      for (int r = 0; r < 2; r++)
        for (int c = 0; c < 2; c++) H[r * 2 + c] += (J[r] * J[c]) * grad_weight[k];
    }
  }
  for (int i = 0; i < 2; i++) H[i * 2 + i] *= (1 + 0.001);
  *h_inv = 1.0 / H[0] * patch_size * patch_size;
  float Hinv[4];
  {
    const float det = H[0] * H[3] - H[2] * H[1];
    const float invdet = 1.0f / det;
    Hinv[0] = H[3] * invdet; Hinv[1] = -H[1] * invdet; Hinv[2] = -H[2] * invdet; Hinv[3] = H[0] * invdet;
  }
  float mean_diff = 0;
  float u = cur_px_estimate[0];
  float v = cur_px_estimate[1];
  const float min_update_squared = 0.01 * 0.01;
  const int cur_step = cols;
  float chi2 = 0;
  float update[2] = { 0, 0 }, Jres[2];
  int iter = 0;
  for (iter = 0; iter < n_iter; ++iter) {
    float* cur_patch_ptr = cur_patch;
    const int u_r = (int)floor(u);
    const int v_r = (int)floor(v);
    if (u_r < halfpatch_size_ || v_r < halfpatch_size_ || u_r >= cols - halfpatch_size_ || v_r >= rows - halfpatch_size_) break;
    if (isnan(u) || isnan(v)) { if (iters_out) *iters_out = iter; if (chi2_out) *chi2_out = chi2; return 0; }
    const float subpix_x = u - u_r, subpix_y = v - v_r;
    const float wTL = (1.0 - subpix_x) * (1.0 - subpix_y);
    const float wTR = subpix_x * (1.0 - subpix_y);
    const float wBL = (1.0 - subpix_x) * subpix_y;
    const float wBR = subpix_x * subpix_y;
    float new_chi2 = 0.0;
    Jres[0] = Jres[1] = 0;
    k = 0;
    for (int y = 0; y < patch_size; ++y) {
      const uint8_t* it = cur_img + (v_r + y - halfpatch_size_) * cur_step + u_r - halfpatch_size_;
      for (int x = 0; x < patch_size; ++x, ++it, ++k) {
        const float search_pixel = wTL * it[0] + wTR * it[1] + wBL * it[cur_step] + wBR * it[cur_step + 1];
        const float res = search_pixel - ref_patch[k] + mean_diff;
        Jres[0] -= res * ref_patch_dv[k] * grad_weight[k];
        Jres[1] -= res * grad_weight[k];
        new_chi2 += res * res * grad_weight[k];
        if (cur_patch != NULL) { *cur_patch_ptr = search_pixel; ++cur_patch_ptr; }
      }
    }
    chi2 = new_chi2;
    update[0] = Hinv[0] * Jres[0] + Hinv[1] * Jres[1];
    update[1] = Hinv[2] * Jres[0] + Hinv[3] * Jres[1];
    u += update[0] * dir[0];
    v += update[0] * dir[1];
    mean_diff += update[1];
    hso_or_margin_note(HSO_M_LK_UPDATE, ((double)(update[0] * update[0]) - min_update_squared) / min_update_squared);
    if (update[0] * update[0] < min_update_squared) { converged = 1; iter++; break; }
  }
  hso_or_margin_note(HSO_M_LK_CHI2, ((double)chi2 - 1000 * patch_area) / (1000 * patch_area));
  if (chi2 > 1000 * patch_area) converged = 0;
  cur_px_estimate[0] = u; cur_px_estimate[1] = v;
  if (iters_out) *iters_out = iter;
  if (chi2_out) *chi2_out = chi2;
  return converged;
}

/* Matcher::checkNCC, src/matcher.cpp:379-404; returns the compared value */
double hso_or_ncc(const float* patch1, const float* patch2)
{
  const int NCC_area = 64;
  float mean1 = 0, mean2 = 0;
  for (int i = 0; i < NCC_area; ++i) { mean1 += patch1[i]; mean2 += patch2[i]; }
  mean1 /= NCC_area;
  mean2 /= NCC_area;
  float numerator = 0, demoniator1 = 0, demoniator2 = 0;
  for (int i = 0; i < NCC_area; i++) {
    const float patch1_mean = patch1[i] - mean1;
    const float patch2_mean = patch2[i] - mean2;
    numerator += patch1_mean * patch2_mean;
    demoniator1 += patch1_mean * patch1_mean;
    demoniator2 += patch2_mean * patch2_mean;
  }
  return numerator / (sqrtf(demoniator1 * demoniator2) + 1e-12);
}

/* Matcher::checkNormal, src/matcher.cpp:406-440; returns normal.dot(n) */
double hso_or_normal_dot(const int16_t* gx, const int16_t* gy, int cols, int rows, const double pxLevel[2], const double normal[2])
{
  const float uf = pxLevel[0];
  const float vf = pxLevel[1];
  /* defined behaviour where the reference's is not (see hso_oracle.h): the 2x2 taps must lie inside the gradient image */
  if (!(uf >= 0 && vf >= 0 && uf < (float)(cols - 1) && vf < (float)(rows - 1))) return -2.0;
  const int ui = floorf(pxLevel[0]);
  const int vi = floorf(pxLevel[1]);
  const float subpix_x = uf - ui;
  const float subpix_y = vf - vi;
  const float wTL = (1.0 - subpix_x) * (1.0 - subpix_y);
  const float wTR = subpix_x * (1.0 - subpix_y);
  const float wBL = (1.0 - subpix_x) * subpix_y;
  const float wBR = 1.0 - wTL - wTR - wBL;
  const short gx00 = gx[vi * cols + ui], gx10 = gx[vi * cols + ui + 1], gx01 = gx[(vi + 1) * cols + ui], gx11 = gx[(vi + 1) * cols + ui + 1];
  const short gy00 = gy[vi * cols + ui], gy10 = gy[vi * cols + ui + 1], gy01 = gy[(vi + 1) * cols + ui], gy11 = gy[(vi + 1) * cols + ui + 1];
  double n0 = ((wTL * (double)gx00 + wTR * (double)gx10) + wBL * (double)gx01) + wBR * (double)gx11;
  double n1 = ((wTL * (double)gy00 + wTR * (double)gy10) + wBL * (double)gy01) + wBR * (double)gy11;
  const double nn = sqrt(n0 * n0 + n1 * n1);
  n0 /= nn; n1 /= nn;
  return normal[0] * n0 + normal[1] * n1;
}

/* Matcher::findMatchDirect after the reference feature has been chosen, src/matcher.cpp:286-375.
 * ref_pyr / cur_pyr: level pointers; cur_gx/cur_gy: Sobel images of the current frame's levels 0-2. */
static void find_match(const hso_camera* cam, const hso_align_job* job, const uint8_t* const ref_pyr[HSO_N_PYR_LEVELS],
                       const uint8_t* const cur_pyr[HSO_N_PYR_LEVELS], const int16_t* const cur_gx[HSO_N_SOBEL_LEVELS],
                       const int16_t* const cur_gy[HSO_N_SOBEL_LEVELS], int w, int h, double ncc_thresh, hso_align_out* out)
{
  const int halfpatch_size_ = 4, patch_size_ = 8;
  memset(out, 0, sizeof(*out));
  out->px_cur[0] = job->px_cur[0]; out->px_cur[1] = job->px_cur[1];
  /* isInFrame((px/(1<<level)).cast<int>(), halfpatch_size_+2, level), camera.h:85-89 */
  {
    const int L = job->ref_level, b = halfpatch_size_ + 2;
    const int ox = (int)(job->px_ref[0] / (1 << L)), oy = (int)(job->px_ref[1] / (1 << L));
    if (!(ox >= b && ox < w / (1 << L) - b && oy >= b && oy < h / (1 << L) - b)) { out->stage = HSO_ALIGN_REF_BORDER; return; }
  }
  hso_or_warp_matrix_affine(cam, cam, job->px_ref, job->f_ref, job->depth, &job->T_cur_ref, job->ref_level, out->A_cur_ref);
  const int search_level = hso_or_best_search_level(out->A_cur_ref, HSO_N_SOBEL_LEVELS - 1);  /* Config::nPyrLevels()-1 */
  out->search_level = search_level;
  float temp[100], pwb[100], patch[64], patchNCC[64];
  int rcols, rrows;  /* img_pyr_[level].cols / rows */
  hso_or_pyramid_dims(w, h, job->ref_level, &rcols, &rrows);
  hso_or_warp_affine(out->A_cur_ref, ref_pyr[job->ref_level], rcols, rrows, job->px_ref,
                     job->ref_level, search_level, halfpatch_size_ + 1, temp);
  if (job->kf_gap_lt4 && fabsf(job->exposure_rat * 128 - 128) > 30.0f) {
    for (int i = 0; i < 100; i++) pwb[i] = temp[i] * job->exposure_rat;
  } else {
    memcpy(pwb, temp, sizeof(pwb));
  }
  for (int y = 1; y < patch_size_ + 1; ++y)
    for (int x = 0; x < patch_size_; ++x) patch[(y - 1) * patch_size_ + x] = pwb[y * (patch_size_ + 2) + 1 + x];
  memset(patchNCC, 0, sizeof(patchNCC));
  double px_scaled[2] = { job->px_cur[0] / (1 << search_level), job->px_cur[1] / (1 << search_level) };
  const double px_scaled_orig[2] = { px_scaled[0], px_scaled[1] };
  int cols, rows;
  hso_or_pyramid_dims(w, h, search_level, &cols, &rows);
  int ok;
  if (job->type == HSO_FTR_EDGELET) {
    double d0 = out->A_cur_ref[0] * job->grad[0] + out->A_cur_ref[1] * job->grad[1];
    double d1 = out->A_cur_ref[2] * job->grad[0] + out->A_cur_ref[3] * job->grad[1];
    const double dn = sqrt(d0 * d0 + d1 * d1);
    d0 /= dn; d1 /= dn;
    const float dirf[2] = { (float)d0, (float)d1 };
    ok = hso_or_align1d(cur_pyr[search_level], cols, rows, dirf, pwb, patch, 10, px_scaled, &out->h_inv, patchNCC,
                        &out->iters, &out->chi2);
    if (!ok) out->stage = HSO_ALIGN_NOT_CONVERGED;
    if (ok) {
      const double dir[2] = { d0, d1 };
      const double nd_ = hso_or_normal_dot(cur_gx[search_level], cur_gy[search_level], cols, rows, px_scaled, dir);
      hso_or_margin_note(HSO_M_NORMAL, nd_ - (float)0.86);
      ok = nd_ > (float)0.86;
      if (!ok) out->stage = HSO_ALIGN_NORMAL;
    }
  } else {
    ok = hso_or_align2d(cur_pyr[search_level], cols, rows, pwb, patch, 10, px_scaled, patchNCC, &out->iters, &out->chi2);
    if (!ok) out->stage = HSO_ALIGN_NOT_CONVERGED;
  }
  const double ncc = hso_or_ncc(patch, patchNCC);
  out->ncc = (float)ncc;
  if (ok) {
    hso_or_margin_note(HSO_M_NCC, ncc - ncc_thresh);
    ok = ncc > ncc_thresh;  /* float thresh parameter, matcher.cpp:379 */
    if (!ok) out->stage = HSO_ALIGN_NCC;
  }
  if (ok) {
    const double dx = px_scaled_orig[0] - px_scaled[0], dy = px_scaled_orig[1] - px_scaled[1];
    hso_or_margin_note(HSO_M_JUMP, sqrt(dx * dx + dy * dy) - 20);
    ok = sqrt(dx * dx + dy * dy) < 20;
    if (!ok) out->stage = HSO_ALIGN_JUMP;
  }
  out->px_cur[0] = px_scaled[0] * (1 << search_level);
  out->px_cur[1] = px_scaled[1] * (1 << search_level);
  out->success = ok;
}

void hso_or_find_match_direct(const hso_camera* cam, const hso_align_job* job, const uint8_t* const ref_pyr[HSO_N_PYR_LEVELS],
                              const uint8_t* const cur_pyr[HSO_N_PYR_LEVELS], const int16_t* const cur_gx[HSO_N_SOBEL_LEVELS],
                              const int16_t* const cur_gy[HSO_N_SOBEL_LEVELS], int w, int h, hso_align_out* out)
{
  find_match(cam, job, ref_pyr, cur_pyr, cur_gx, cur_gy, w, h, (double)0.7f, out); /* checkNCC(.., 0.7), matcher.cpp:364 */
}

/* Matcher::findMatchSeed after the parallax test, src/matcher.cpp:451-518: the same body with the
 * exposure compensation unconditional on the keyframe gap (caller sets kf_gap_lt4 = 1) and
 * checkNCC(.., 0.8) — the ncc_thresh argument of the reference function is not used. */
void hso_or_find_match_seed(const hso_camera* cam, const hso_align_job* job, const uint8_t* const ref_pyr[HSO_N_PYR_LEVELS],
                            const uint8_t* const cur_pyr[HSO_N_PYR_LEVELS], const int16_t* const cur_gx[HSO_N_SOBEL_LEVELS],
                            const int16_t* const cur_gy[HSO_N_SOBEL_LEVELS], int w, int h, hso_align_out* out)
{
  find_match(cam, job, ref_pyr, cur_pyr, cur_gx, cur_gy, w, h, (double)0.8f, out);
}
