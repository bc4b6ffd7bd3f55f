/*
 * hso_oracle_tracker.c — CoarseTracker restated statement-for-statement.
 * TEST INFRASTRUCTURE (see hso_oracle.h).  Follows src/CoarseTracker.cpp,
 * include/hso/CoarseTracker.h and include/hso/MatrixAccumulator.h of the
 * reference, including the quirks listed in SURVEY.md §8(a) a4-a8:
 *  - duplicate {-1,0} / missing {0,-1} in the 9-pixel pattern (CoarseTracker.h:69);
 *  - w_br = 1-(tl+tr+bl) for the reference patch (:467) but su*sv for the
 *    current image (:323);
 *  - no outlier saturation and E += hw*r*r at the top level (:350-361);
 *  - J cast to float and H accumulated in 3-tier fp32 (MatrixAccumulator.h:65-140),
 *    where the 1k tier flushes on every 1001-sample flush (numIn1k = 1001 > 1000);
 *  - E is a serial fp32 sum, b is fp64.
 */
#include "hso_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* include/hso/CoarseTracker.h:58-120 */
static const int8_t kPattern[8][40][2] = {
  { {0,0} },
  { {0,-1}, {-1,0}, {0,0}, {1,0}, {0,1} },
  { {-1,-1}, {-1,0}, {-1,1}, {-1,0}, {0,0}, {0,1}, {1,-1}, {1,0}, {1,1} },
  { {0,-2}, {-1,-1}, {1,-1}, {-2,0}, {0,0}, {2,0}, {-1,1}, {1,1}, {0,2}, {0,-1}, {-1,0}, {1,0}, {0,1} },
  { {0,-2}, {-1,-1}, {1,-1}, {-2,0}, {0,0}, {2,0}, {-1,1}, {1,1}, {0,2}, {-2,-2}, {-2,2}, {2,-2}, {2,2} },
  { {0,-2}, {-1,-1}, {1,-1}, {-2,0}, {0,0}, {2,0}, {-1,1}, {1,1}, {0,2}, {-2,-2}, {-2,2}, {2,-2}, {2,2},
    {-3,-1}, {-3,1}, {3,-1}, {3,1}, {1,-3}, {-1,-3}, {1,3}, {-1,3} },
  { {-2,-2}, {-2,-1}, {-2,0}, {-2,1}, {-2,2}, {-1,-2}, {-1,-1}, {-1,0}, {-1,1}, {-1,2},
    {0,-2}, {0,-1}, {0,0}, {0,1}, {0,2}, {1,-2}, {1,-1}, {1,0}, {1,1}, {1,2},
    {2,-2}, {2,-1}, {2,0}, {2,1}, {2,2} },
  { {-4,-4}, {-4,-2}, {-4,0}, {-4,2}, {-4,4}, {-2,-4}, {-2,-2}, {-2,0}, {-2,2}, {-2,4},
    {0,-4}, {0,-2}, {0,0}, {0,2}, {0,4}, {2,-4}, {2,-2}, {2,0}, {2,2}, {2,4},
    {4,-4}, {4,-2}, {4,0}, {4,2}, {4,4} },
};
static const int kPatternNum[8] = { 1, 5, 9, 13, 13, 21, 25, 25 };
static const int kPatternPadding[8] = { 1, 1, 1, 2, 2, 3, 2, 4 };
#define PATTERN_OFFSET 2 /* m_pattern_offset, CoarseTracker.h:122 */

int hso_or_tracker_pattern(int max_level, int level, int* patch_area, int* half_patch, int8_t* offsets_xy)
{
  const int off = max_level - level + PATTERN_OFFSET; /* CoarseTracker.cpp:80 */
  if (off < 0 || off > 7) return -1;
  if (patch_area) *patch_area = kPatternNum[off];
  if (half_patch) *half_patch = kPatternPadding[off];
  if (offsets_xy) memcpy(offsets_xy, kPattern[off], 80);
  return off;
}

/* include/hso/MatrixAccumulator.h:29-141 (lane 0 of each 4-float group; the
 * other three lanes are never written with off=0 and stay 0). */
typedef struct {
  float d1[28], d1k[28], d1m[28];
  float numIn1, numIn1k, numIn1m;
  float H[49];
} acc7_t;

static void acc7_shift_up(acc7_t* a, int force)
{
  if (a->numIn1 > 1000 || force) {
    for (int i = 0; i < 28; i++) a->d1k[i] = a->d1[i] + a->d1k[i];
    a->numIn1k += a->numIn1; a->numIn1 = 0;
    memset(a->d1, 0, sizeof(a->d1));
  }
  if (a->numIn1k > 1000 || force) {
    for (int i = 0; i < 28; i++) a->d1m[i] = a->d1k[i] + a->d1m[i];
    a->numIn1m += a->numIn1k; a->numIn1k = 0;
    memset(a->d1k, 0, sizeof(a->d1k));
  }
}

static void acc7_update(acc7_t* a, float J0, float J1, float J2, float J3, float J4, float J5, float J6, float w)
{
  float* pt = a->d1;
  *pt += J0 * J0 * w; pt++; J0 *= w;
  *pt += J1 * J0; pt++; *pt += J2 * J0; pt++; *pt += J3 * J0; pt++;
  *pt += J4 * J0; pt++; *pt += J5 * J0; pt++; *pt += J6 * J0; pt++;
  *pt += J1 * J1 * w; pt++; J1 *= w;
  *pt += J2 * J1; pt++; *pt += J3 * J1; pt++; *pt += J4 * J1; pt++;
  *pt += J5 * J1; pt++; *pt += J6 * J1; pt++;
  *pt += J2 * J2 * w; pt++; J2 *= w;
  *pt += J3 * J2; pt++; *pt += J4 * J2; pt++; *pt += J5 * J2; pt++; *pt += J6 * J2; pt++;
  *pt += J3 * J3 * w; pt++; J3 *= w;
  *pt += J4 * J3; pt++; *pt += J5 * J3; pt++; *pt += J6 * J3; pt++;
  *pt += J4 * J4 * w; pt++; J4 *= w;
  *pt += J5 * J4; pt++; *pt += J6 * J4; pt++;
  *pt += J5 * J5 * w; pt++; J5 *= w;
  *pt += J6 * J5; pt++;
  *pt += J6 * J6 * w; pt++;
  a->numIn1++;
  acc7_shift_up(a, 0);
}

static void acc7_finish(acc7_t* a)
{
  acc7_shift_up(a, 1);
  int idx = 0;
  for (int r = 0; r < 7; r++)
    for (int c = r; c < 7; c++) {
      const float d = a->d1m[idx] + 0.0f + 0.0f + 0.0f;
      a->H[r * 7 + c] = a->H[c * 7 + r] = d;
      idx++;
    }
}

struct hso_or_tracker {
  hso_camera cam;
  hso_track_params p;
  const uint8_t* ref_pyr[HSO_N_PYR_LEVELS];
  const uint8_t* cur_pyr[HSO_N_PYR_LEVELS];
  int w, h, n;
  const hso_ref_feat* feats;
  int level, offset_all, half_patch, patch_area;
  float* ref_patch;     /* m_ref_patch_cache, n x PATCH_AREA */
  uint8_t* visible;     /* m_visible_fts */
  double* jac_raw;      /* m_jacobian_cache_raw, 6 x (n*PATCH_AREA), column-major */
  double* jac_true;     /* m_jacobian_cache_true */
  double* buf_jac;      /* m_buf_jacobian, 7 per term */
  double* buf_weight;
  double* buf_error;
  int n_buf;
  int total_terms, saturated_terms;
  float huber, outlier;
  int iter;
  double E64; /* diagnostics only: the same fp32 terms as E, summed in fp64 */
  int decide_on_e64; /* diagnostics only, default 0: see hso_or_tracker_decide_on_f64_sum */
};

/* include/hso/frame.h:192-212 */
static void jacobian_xyz2uv(const double xyz[3], double J[12])
{
  const double x = xyz[0], y = xyz[1];
  const double z_inv = 1. / xyz[2];
  const double z_inv_2 = z_inv * z_inv;
  J[0] = -z_inv; J[1] = 0.0; J[2] = x * z_inv_2; J[3] = y * J[2];
  J[4] = -(1.0 + x * J[2]); J[5] = y * z_inv;
  J[6] = 0.0; J[7] = -z_inv; J[8] = y * z_inv_2; J[9] = 1.0 + y * J[8];
  J[10] = -J[3]; J[11] = -x * z_inv;
}

hso_or_tracker* hso_or_tracker_create(const hso_camera* cam, const hso_track_params* p,
                                      const uint8_t* const ref_pyr[HSO_N_PYR_LEVELS],
                                      const uint8_t* const cur_pyr[HSO_N_PYR_LEVELS],
                                      int w, int h, const hso_ref_feat* feats, int n)
{
  hso_or_tracker* t = (hso_or_tracker*)calloc(1, sizeof(*t));
  t->cam = *cam; t->p = *p; t->w = w; t->h = h; t->n = n; t->feats = feats;
  for (int i = 0; i < HSO_N_PYR_LEVELS; i++) { t->ref_pyr[i] = ref_pyr[i]; t->cur_pyr[i] = cur_pyr[i]; }
  const size_t terms = (size_t)n * 25 + 1;
  t->ref_patch = (float*)calloc(terms, sizeof(float));
  t->visible = (uint8_t*)calloc((size_t)n + 1, 1);
  t->jac_raw = (double*)calloc(terms * 6, sizeof(double));
  t->jac_true = (double*)calloc(terms * 6, sizeof(double));
  t->buf_jac = (double*)calloc(terms * 7, sizeof(double));
  t->buf_weight = (double*)calloc(terms, sizeof(double));
  t->buf_error = (double*)calloc(terms, sizeof(double));
  t->huber = 5.2f; t->outlier = 100.f;
  return t;
}

void hso_or_tracker_destroy(hso_or_tracker* t)
{
  if (!t) return;
  free(t->ref_patch); free(t->visible); free(t->jac_raw); free(t->jac_true);
  free(t->buf_jac); free(t->buf_weight); free(t->buf_error); free(t);
}

/* CoarseTracker.cpp:416-497 */
static void precompute_reference_patches(hso_or_tracker* t)
{
  const int border = t->half_patch + 1;
  const uint8_t* ref_img = t->ref_pyr[t->level];
  int cols, rows;  /* img_pyr_[level].cols / rows */
  hso_or_pyramid_dims(t->w, t->h, t->level, &cols, &rows);
  const int stride = cols;
  const float scale = 1.0f / (1 << t->level);
  const double fxl = t->cam.fx * scale;
  const double fyl = t->cam.fy * scale;
  const int8_t (*pat)[2] = kPattern[t->offset_all];
  const int PA = t->patch_area;

  for (int fc = 0; fc < t->n; fc++) {
    const hso_ref_feat* ft = &t->feats[fc];
    /* `point == NULL` features are flattened to dist < 0 together with the
     * p_ref[2] < 1e-5 case (:217,:222); both produce no terms.  The reference
     * sets the visibility bit for the latter in forward mode, which nothing
     * observes; here the bit is set whenever the patch is inside the image so
     * that both modes agree with :444. */
    const float u_ref = ft->px[0] * scale;
    const float v_ref = ft->px[1] * scale;
    const int u_ref_i = floorf(u_ref);
    const int v_ref_i = floorf(v_ref);
    if (ft->dist < 0) continue;
    if (u_ref_i - border < 0 || v_ref_i - border < 0 || u_ref_i + border >= cols || v_ref_i + border >= rows)
      continue;
    t->visible[fc] = 1;

    double frame_jac[12] = { 0 };
    if (t->p.inverse_composition) {
      const double dist = ft->dist;
      const double xyz_ref[3] = { ft->f[0] * dist, ft->f[1] * dist, ft->f[2] * dist };
      jacobian_xyz2uv(xyz_ref, frame_jac);
    }
    const float subpix_u_ref = u_ref - u_ref_i;
    const float subpix_v_ref = v_ref - v_ref_i;
    const float w_ref_tl = (1.0 - subpix_u_ref) * (1.0 - subpix_v_ref);
    const float w_ref_tr = subpix_u_ref * (1.0 - subpix_v_ref);
    const float w_ref_bl = (1.0 - subpix_u_ref) * subpix_v_ref;
    const float w_ref_br = 1.0 - (w_ref_tl + w_ref_tr + w_ref_bl);

    float* cache_ptr = t->ref_patch + (size_t)PA * fc;
    for (int n = 0; n < PA; ++n, ++cache_ptr) {
      const uint8_t* p = ref_img + (v_ref_i + pat[n][1]) * stride + u_ref_i + pat[n][0];
      *cache_ptr = w_ref_tl * p[0] + w_ref_tr * p[1] + w_ref_bl * p[stride] + w_ref_br * p[stride + 1];
      if (t->p.inverse_composition) {
        const float dx = 0.5f * ((w_ref_tl * p[1] + w_ref_tr * p[2] + w_ref_bl * p[stride + 1] + w_ref_br * p[stride + 2])
                               - (w_ref_tl * p[-1] + w_ref_tr * p[0] + w_ref_bl * p[stride - 1] + w_ref_br * p[stride]));
        const float dy = 0.5f * ((w_ref_tl * p[stride] + w_ref_tr * p[1 + stride] + w_ref_bl * p[stride * 2] + w_ref_br * p[stride * 2 + 1])
                               - (w_ref_tl * p[-stride] + w_ref_tr * p[1 - stride] + w_ref_bl * p[0] + w_ref_br * p[1]));
        double* col = t->jac_raw + 6 * ((size_t)fc * PA + n);
        for (int k = 0; k < 6; k++)
          col[k] = ((double)dx * frame_jac[k]) * fxl + ((double)dy * frame_jac[6 + k]) * fyl;
      }
    }
  }
}

void hso_or_tracker_set_level(hso_or_tracker* t, int level)
{
  /* CoarseTracker.cpp:77-91 */
  t->level = level;
  memset(t->visible, 0, (size_t)t->n);
  t->offset_all = t->p.max_level - level + PATTERN_OFFSET;
  t->half_patch = kPatternPadding[t->offset_all];
  t->patch_area = kPatternNum[t->offset_all];
  memset(t->ref_patch, 0, sizeof(float) * (size_t)t->n * t->patch_area);
  precompute_reference_patches(t);
}

void hso_or_tracker_get_cache(const hso_or_tracker* t, float* ref_patch, uint8_t* visible, int* patch_area)
{
  if (ref_patch) memcpy(ref_patch, t->ref_patch, sizeof(float) * (size_t)t->n * t->patch_area);
  if (visible) memcpy(visible, t->visible, (size_t)t->n);
  if (patch_area) *patch_area = t->patch_area;
}

void hso_or_tracker_set_thresholds(hso_or_tracker* t, float huber, float outlier)
{
  t->huber = huber; t->outlier = outlier;
}

/* CoarseTracker.cpp:530-644 */
int hso_or_tracker_select(hso_or_tracker* t, const hso_se3* T_cur_ref, float exposure_rat,
                          float* huber, float* outlier, float* abs_err_out)
{
  const float b = 0;
  const uint8_t* cur_img = t->cur_pyr[t->level];
  int cols, rows;  /* img_pyr_[level].cols / rows */
  hso_or_pyramid_dims(t->w, t->h, t->level, &cols, &rows);
  const int stride = cols;
  const int border = t->half_patch + 1;
  const float scale = 1.0f / (1 << t->level);
  const int8_t (*pat)[2] = kPattern[t->offset_all];
  const int PA = t->patch_area;

  float* errors = (float*)malloc(sizeof(float) * ((size_t)t->n * PA + 1));
  int n_err = 0;
  for (int fc = 0; fc < t->n; fc++) {
    if (!t->visible[fc]) continue;
    const hso_ref_feat* ft = &t->feats[fc];
    const double dist = ft->dist; if (dist < 0) continue;
    const double xyz_ref[3] = { ft->f[0] * dist, ft->f[1] * dist, ft->f[2] * dist };
    double xyz_cur[3];
    hso_or_se3_apply(T_cur_ref, xyz_ref, xyz_cur);
    if (xyz_cur[2] < 0) continue;
    double pxd[2];
    hso_or_world2cam(&t->cam, xyz_cur, pxd);
    const float u_cur = (float)pxd[0] * scale;
    const float v_cur = (float)pxd[1] * scale;
    const int u_cur_i = floorf(u_cur);
    const int v_cur_i = floorf(v_cur);
    if (u_cur_i - border < 0 || v_cur_i - border < 0 || u_cur_i + border >= cols || v_cur_i + border >= rows)
      continue;
    const float subpix_u_cur = u_cur - u_cur_i;
    const float subpix_v_cur = v_cur - v_cur_i;
    const float w_cur_tl = (1.0 - subpix_u_cur) * (1.0 - subpix_v_cur);
    const float w_cur_tr = subpix_u_cur * (1.0 - subpix_v_cur);
    const float w_cur_bl = (1.0 - subpix_u_cur) * subpix_v_cur;
    const float w_cur_br = subpix_u_cur * subpix_v_cur;
    const float* ref_patch_cache_ptr = t->ref_patch + (size_t)PA * fc;
    for (int n = 0; n < PA; ++n, ++ref_patch_cache_ptr) {
      const uint8_t* p = cur_img + (v_cur_i + pat[n][1]) * stride + u_cur_i + pat[n][0];
      const float cur_color = w_cur_tl * p[0] + w_cur_tr * p[1] + w_cur_bl * p[stride] + w_cur_br * p[stride + 1];
      const float residual = cur_color - (exposure_rat * (*ref_patch_cache_ptr) + b);
      errors[n_err++] = fabsf(residual);
    }
  }
  if (abs_err_out) memcpy(abs_err_out, errors, sizeof(float) * (size_t)n_err);
  if (n_err < 30) {
    t->huber = 5.2f; t->outlier = 100.f;
  } else {
    const float residual_median = hso_or_median_f(errors, n_err);
    for (int i = 0; i < n_err; i++) errors[i] = fabsf(errors[i] - residual_median);
    const float standard_deviation = 1.4826 * hso_or_median_f(errors, n_err);
    t->huber = residual_median + standard_deviation;
    t->outlier = 3 * t->huber;
    if (t->outlier < 10) t->outlier = 10;
  }
  free(errors);
  if (huber) *huber = t->huber;
  if (outlier) *outlier = t->outlier;
  return n_err;
}

/* CoarseTracker.cpp:242-414; returns E/m_total_terms, fills the term buffers */
static double compute_residuals(hso_or_tracker* t, const hso_se3* T_cur_ref, float exposure_rat,
                                double cutoff_error, float* E_out)
{
  const float b = 0;
  const int PA = t->patch_area;
  if (t->p.inverse_composition) {
    const size_t cnt = (size_t)t->n * PA * 6;
    for (size_t i = 0; i < cnt; i++) t->jac_true[i] = (double)exposure_rat * t->jac_raw[i];
  }
  const uint8_t* cur_img = t->cur_pyr[t->level];
  int cols, rows;  /* img_pyr_[level].cols / rows */
  hso_or_pyramid_dims(t->w, t->h, t->level, &cols, &rows);
  const int stride = cols;
  const int border = t->half_patch + 1;
  const float scale = 1.0f / (1 << t->level);
  const double fxl = t->cam.fx * scale;
  const double fyl = t->cam.fy * scale;
  const float setting_huberTH = t->huber;
  const float max_energy = 2 * setting_huberTH * cutoff_error - setting_huberTH * setting_huberTH;
  const int8_t (*pat)[2] = kPattern[t->offset_all];

  t->n_buf = 0;
  t->total_terms = t->saturated_terms = 0;
  float E = 0;
  double E64 = 0;

  for (int fc = 0; fc < t->n; fc++) {
    if (!t->visible[fc]) continue;
    const hso_ref_feat* ft = &t->feats[fc];
    const double dist = ft->dist; if (dist < 0) continue;
    const double xyz_ref[3] = { ft->f[0] * dist, ft->f[1] * dist, ft->f[2] * dist };
    double xyz_cur[3];
    hso_or_se3_apply(T_cur_ref, xyz_ref, xyz_cur);
    if (xyz_cur[2] < 0) continue;
    double pxd[2];
    hso_or_world2cam(&t->cam, xyz_cur, pxd);
    const float uv0 = (float)pxd[0], uv1 = (float)pxd[1];
    const float u_cur = uv0 * scale;
    const float v_cur = uv1 * scale;
    const int u_cur_i = floorf(u_cur);
    const int v_cur_i = floorf(v_cur);
    if (u_cur_i - border < 0 || v_cur_i - border < 0 || u_cur_i + border >= cols || v_cur_i + border >= rows)
      continue;
    double frame_jac[12] = { 0 };
    if (!t->p.inverse_composition) jacobian_xyz2uv(xyz_cur, frame_jac);

    const float subpix_u_cur = u_cur - u_cur_i;
    const float subpix_v_cur = v_cur - v_cur_i;
    const float w_cur_tl = (1.0 - subpix_u_cur) * (1.0 - subpix_v_cur);
    const float w_cur_tr = subpix_u_cur * (1.0 - subpix_v_cur);
    const float w_cur_bl = (1.0 - subpix_u_cur) * subpix_v_cur;
    const float w_cur_br = subpix_u_cur * subpix_v_cur;

    const float* ref_patch_cache_ptr = t->ref_patch + (size_t)PA * fc;
    for (int n = 0; n < PA; ++n, ++ref_patch_cache_ptr) {
      const uint8_t* p = cur_img + (v_cur_i + pat[n][1]) * stride + u_cur_i + pat[n][0];
      const float cur_color = w_cur_tl * p[0] + w_cur_tr * p[1] + w_cur_bl * p[stride] + w_cur_br * p[stride + 1];
      if (!isfinite(cur_color)) continue;
      const float residual = cur_color - (exposure_rat * (*ref_patch_cache_ptr) + b);
      const float hw = fabsf(residual) < setting_huberTH ? 1 : setting_huberTH / fabsf(residual);

      if (fabsf(residual) > cutoff_error && t->level < t->p.max_level) {
        E += max_energy; E64 += max_energy;
        t->total_terms++;
        t->saturated_terms++;
      } else {
        if (t->level == t->p.max_level) { E += hw * residual * residual; E64 += (float)(hw * residual * residual); }
        else { E += hw * residual * residual * (2 - hw); E64 += (float)(hw * residual * residual * (2 - hw)); }
        t->total_terms++;
        double* J = t->buf_jac + 7 * (size_t)t->n_buf;
        if (!t->p.inverse_composition) {
          const float dx = 0.5f * ((w_cur_tl * p[1] + w_cur_tr * p[2] + w_cur_bl * p[stride + 1] + w_cur_br * p[stride + 2])
                                 - (w_cur_tl * p[-1] + w_cur_tr * p[0] + w_cur_bl * p[stride - 1] + w_cur_br * p[stride]));
          const float dy = 0.5f * ((w_cur_tl * p[stride] + w_cur_tr * p[1 + stride] + w_cur_bl * p[stride * 2] + w_cur_br * p[stride * 2 + 1])
                                 - (w_cur_tl * p[-stride] + w_cur_tr * p[1 - stride] + w_cur_bl * p[0] + w_cur_br * p[1]));
          for (int k = 0; k < 6; k++)
            J[1 + k] = ((double)dx * frame_jac[k]) * fxl + ((double)dy * frame_jac[6 + k]) * fyl;
        } else {
          const double* col = t->jac_true + 6 * ((size_t)fc * PA + n);
          for (int k = 0; k < 6; k++) J[1 + k] = col[k];
        }
        J[0] = -(*ref_patch_cache_ptr);
        t->buf_weight[t->n_buf] = hw;
        t->buf_error[t->n_buf] = residual;
        t->n_buf++;
      }
    }
  }
  if (E_out) *E_out = E;
  t->E64 = E64;
  if (t->decide_on_e64) return (float)E64 / t->total_terms;  /* diagnostics: the same terms, summed without the serial fp32 rounding */
  return E / t->total_terms;
}

/* CoarseTracker.cpp:499-525 */
static void compute_gs(const hso_or_tracker* t, double H_out[49], double b_out[7])
{
  acc7_t acc;
  memset(&acc, 0, sizeof(acc));
  for (int k = 0; k < 7; k++) b_out[k] = 0;
  for (int i = 0; i < t->n_buf; i++) {
    const double* J = t->buf_jac + 7 * (size_t)i;
    acc7_update(&acc, (float)J[0], (float)J[1], (float)J[2], (float)J[3], (float)J[4], (float)J[5], (float)J[6],
                (float)t->buf_weight[i]);
    for (int k = 0; k < 7; k++) b_out[k] -= J[k] * t->buf_error[i] * t->buf_weight[i];
  }
  acc7_finish(&acc);
  for (int i = 0; i < 49; i++) H_out[i] = (double)acc.H[i];
}

void hso_or_tracker_eval(hso_or_tracker* t, const hso_se3* T, float exposure_rat, hso_eval_out* out)
{
  float E;
  const double cutoff_error = t->outlier;
  memset(out, 0, sizeof(*out));
  out->energy = compute_residuals(t, T, exposure_rat, cutoff_error, &E);
  out->energy_sum = E;
  compute_gs(t, out->H, out->b);
  out->n_terms = t->total_terms;
  out->n_saturated = t->saturated_terms;
  out->huber = t->huber; out->outlier = t->outlier;
  int nv = 0;
  for (int i = 0; i < t->n; i++) nv += t->visible[i];
  out->n_visible = nv;
}

/* diagnostics: fp64 sum of the fp32 energy terms of the last evaluation (the reference's
 * own E is the serial fp32 sum returned in hso_eval_out.energy_sum) */
double hso_or_tracker_energy_f64(const hso_or_tracker* t) { return t->E64; }

/* diagnostics: on != 0 makes run() compare energies formed from the fp64 sum of the same fp32 terms instead of the
 * reference's serial fp32 sum (CoarseTracker.cpp:272,352-361,413).  NOT the reference's behaviour: it exists to measure how
 * many of the reference's accept decisions (:143) are decided by the rounding of that serial sum (~26 000 terms, relative
 * noise ~1e-5) rather than by the energies; bench.py reports it as se3_vs_cpu.accept_sequence_equal_frac_f64_sum. */
void hso_or_tracker_decide_on_f64_sum(hso_or_tracker* t, int on) { t->decide_on_e64 = on; }

/* CoarseTracker.cpp:51-208 (without the frame write-back of :198-202) */
void hso_or_tracker_run(hso_or_tracker* t, const hso_se3* T_init, float exposure_init, hso_track_result* out)
{
  memset(out, 0, sizeof(*out));
  float m_exposure_rat = exposure_init;
  hso_se3 m_T_cur_ref = *T_init;
  if (t->n == 0) { out->T_cur_ref = m_T_cur_ref; out->exposure_rat = m_exposure_rat; return; }

  for (int level = t->p.max_level; level >= t->p.min_level; --level) {
    hso_or_tracker_set_level(t, level);
    out->n_select[level] = hso_or_tracker_select(t, &m_T_cur_ref, m_exposure_rat, NULL, NULL, NULL);
    out->huber[level] = t->huber; out->outlier[level] = t->outlier;
    const double cutoff_error = t->outlier;
    double energy_old = compute_residuals(t, &m_T_cur_ref, m_exposure_rat, cutoff_error, NULL);
    out->n_eval[level] = 1;
    double H[49], b[7];
    compute_gs(t, H, b);
    float lambda = 0.1;
    for (t->iter = 0; t->iter < t->p.n_iter; t->iter++) {
      double Hl[49], step[7];
      memcpy(Hl, H, sizeof(Hl));
      for (int i = 0; i < 7; i++) Hl[i * 7 + i] *= (1 + lambda);
      hso_or_ldlt_solve(Hl, b, 7, step);
      float extrap_fac = 1;
      if (lambda < 0.001) extrap_fac = sqrt(sqrt(0.001 / lambda));
      for (int i = 0; i < 7; i++) step[i] *= extrap_fac;
      double ssum = 0;
      for (int i = 0; i < 7; i++) ssum += step[i];
      if (!isfinite(ssum) || isnan(step[0])) for (int i = 0; i < 7; i++) step[i] = 0;

      const float new_exposure_rat = m_exposure_rat + step[0];
      double neg[6];
      for (int i = 0; i < 6; i++) neg[i] = -step[1 + i];
      hso_se3 dT, new_T;
      hso_or_se3_exp(neg, &dT);
      if (!t->p.inverse_composition) hso_or_se3_mul(&dT, &m_T_cur_ref, &new_T);
      else hso_or_se3_mul(&m_T_cur_ref, &dT, &new_T);

      const double energy_new = compute_residuals(t, &new_T, new_exposure_rat, cutoff_error, NULL);
      out->n_eval[level]++;
      out->iters[level] = t->iter + 1;
      hso_or_margin_note(HSO_M_TRACK_ACCEPT, (energy_new - energy_old) / energy_old);
      if (energy_new < energy_old) {
        compute_gs(t, H, b);
        energy_old = energy_new;
        m_exposure_rat = new_exposure_rat;
        m_T_cur_ref = new_T;
        lambda *= 0.5;
        if (t->iter < 64) out->accept_mask[level] |= (1ull << t->iter);
      } else {
        lambda *= 4;
        if (lambda < 0.001) lambda = 0.001;
      }
      double nrm = 0;
      for (int i = 0; i < 7; i++) nrm += step[i] * step[i];
      nrm = sqrt(nrm);
      if (!(nrm > 1e-4)) break;
    }
    out->energy[level] = energy_old;
  }
  out->T_cur_ref = m_T_cur_ref;
  out->exposure_rat = m_exposure_rat;
  out->n_terms_last = t->total_terms;
  out->n_saturated_last = t->saturated_terms;
  out->n_tracked = (int32_t)(size_t)((float)t->total_terms / t->patch_area);
}

/* CoarseTracker.cpp:210-240 */
void hso_or_make_depth_ref(const hso_depth_ref_in* in, int n, const hso_se3* poses_f_w,
                           const hso_se3* T_ref_w, double* dist_out)
{
  for (int i = 0; i < n; i++) {
    dist_out[i] = -1;
    if (!in[i].has_point) continue;
    const double inv = 1.0 / in[i].idist;
    const double p_host[3] = { in[i].host_f[0] * inv, in[i].host_f[1] * inv, in[i].host_f[2] * inv };
    hso_se3 Thinv, T_r_h;
    hso_or_se3_inverse(&poses_f_w[in[i].host_pose], &Thinv);
    hso_or_se3_mul(T_ref_w, &Thinv, &T_r_h);
    double p_ref[3];
    hso_or_se3_apply(&T_r_h, p_host, p_ref);
    if (p_ref[2] < 0.00001) continue;
    dist_out[i] = sqrt(p_ref[0] * p_ref[0] + p_ref[1] * p_ref[1] + p_ref[2] * p_ref[2]);
  }
}
