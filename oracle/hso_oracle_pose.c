/*
 * hso_oracle_pose.c — motion-only Levenberg-Marquardt on unit-plane reprojection error.
 * TEST INFRASTRUCTURE (see hso_oracle.h).  Follows
 * pose_optimizer::optimizeLevenbergMarquardt3rd, src/pose_optimizer.cpp:399-771, statement
 * for statement, including the float/double mixture of the original (errors pushed as float,
 * Huber weights through a `const float&` parameter, float thresholds).  The robust-cost
 * pieces (MAD scale, Huber weight) are the restatements pinned against the compiled
 * reference (hso_oracle_robust.c).  Eigen's 6x6 inverse() (PartialPivLU, absent dependency) is
 * restated as Gauss-Jordan with partial pivoting.
 */
#include "hso_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

static void jac_xyz2uv(const double p[3], double J[12])
{
  /* include/hso/frame.h:192-212 */
  const double x = p[0], y = p[1];
  const double z_inv = 1. / p[2];
  const double z_inv_2 = z_inv * z_inv;
  J[0] = -z_inv; J[1] = 0.0; J[2] = x * z_inv_2; J[3] = y * J[2];
  J[4] = -(1.0 + x * J[2]); J[5] = y * z_inv;
  J[6] = 0.0; J[7] = -z_inv; J[8] = y * z_inv_2; J[9] = 1.0 + y * J[8];
  J[10] = -J[3]; J[11] = -x * z_inv;
}

static void invert6(const double* A, double* out)
{
  double m[6][12];
  for (int i = 0; i < 6; i++)
    for (int j = 0; j < 6; j++) { m[i][j] = A[i * 6 + j]; m[i][6 + j] = (i == j) ? 1.0 : 0.0; }
  for (int c = 0; c < 6; c++) {
    int piv = c;
    for (int r = c + 1; r < 6; r++) if (fabs(m[r][c]) > fabs(m[piv][c])) piv = r;
    if (piv != c) for (int j = 0; j < 12; j++) { double t = m[c][j]; m[c][j] = m[piv][j]; m[piv][j] = t; }
    const double d = m[c][c];
    for (int j = 0; j < 12; j++) m[c][j] /= d;
    for (int r = 0; r < 6; r++) {
      if (r == c) continue;
      const double f = m[r][c];
      for (int j = 0; j < 12; j++) m[r][j] -= f * m[c][j];
    }
  }
  for (int i = 0; i < 6; i++)
    for (int j = 0; j < 6; j++) out[i * 6 + j] = m[i][6 + j];
}

typedef struct { double e[2]; double pTarget[3]; } resid_t;

static void residual(const hso_pose_feat* ft, const double pHost[3], const hso_se3* T_f_w, const hso_se3* poses, resid_t* r)
{
  /* SE3 Tth = T_f_w * host->T_f_w_.inverse(); pTarget = Tth * pHost;
   * e = project2d(f) - project2d(pTarget); e *= 1.0 / (1 << level)   (:434-440) */
  hso_se3 hinv, Tth;
  hso_or_se3_inverse(&poses[ft->host_pose], &hinv);
  hso_or_se3_mul(T_f_w, &hinv, &Tth);
  hso_or_se3_apply(&Tth, pHost, r->pTarget);
  const double s = 1.0 / (1 << ft->level);
  r->e[0] = (ft->f[0] / ft->f[2] - r->pTarget[0] / r->pTarget[2]) * s;
  r->e[1] = (ft->f[1] / ft->f[2] - r->pTarget[1] / r->pTarget[2]) * s;
}

void hso_or_pose_optimize(const hso_camera* cam, const hso_pose_job* job, hso_pose_result* out, uint8_t* outlier_mask)
{
  const int n = job->n_feats;
  const hso_pose_feat* F = job->feats;
  memset(out, 0, sizeof(*out));
  out->T_f_w = job->T_f_w;
  if (outlier_mask) memset(outlier_mask, 0, (size_t)n);
  double chi2 = 0.0, rho = 0, mu = 0.1, nu = 2.0;
  int stop = 0, n_trials = 0;
  const int n_trials_max = 5;
  const double em2 = hso_or_error_multiplier2(cam);
  hso_se3 T = job->T_f_w;

  double* chi2_vec_init = (double*)malloc(sizeof(double) * (size_t)(n + 1));
  double* chi2_vec_final = (double*)malloc(sizeof(double) * (size_t)(n + 1));
  float* errors_pt = (float*)malloc(sizeof(float) * (size_t)(n + 1));
  float* errors_ls = (float*)malloc(sizeof(float) * (size_t)(n + 1));
  double* v_host = (double*)malloc(sizeof(double) * 3 * (size_t)(n + 1));
  int n_init = 0, n_pt = 0, n_ls = 0, n_final = 0;

  for (int i = 0; i < n; i++) {
    if (!F[i].has_point) continue;
    double* pHost = v_host + 3 * n_init;
    const double inv = 1.0 / F[i].idist;
    pHost[0] = F[i].host_f[0] * inv; pHost[1] = F[i].host_f[1] * inv; pHost[2] = F[i].host_f[2] * inv;
    resid_t r;
    residual(&F[i], pHost, &T, job->poses_f_w, &r);
    if (F[i].type == HSO_FTR_EDGELET) {
      const float error_ls = F[i].grad[0] * r.e[0] + F[i].grad[1] * r.e[1];
      errors_ls[n_ls++] = fabsf(error_ls);
      chi2_vec_init[n_init] = error_ls * error_ls;
    } else {
      const float error_pt = sqrt(r.e[0] * r.e[0] + r.e[1] * r.e[1]);
      errors_pt[n_pt++] = error_pt;
      chi2_vec_init[n_init] = error_pt * error_pt;
    }
    n_init++;
  }
  if (n_pt == 0 && n_ls == 0) { out->status = 1; goto done; }
  float estimated_scale_pt = 0, estimated_scale_ls = 0;
  if (n_pt > 0 && n_ls > 0) {
    estimated_scale_pt = hso_or_mad_scale(errors_pt, n_pt);
    estimated_scale_ls = hso_or_mad_scale(errors_ls, n_ls);
  } else if (n_pt > 0) {
    estimated_scale_pt = hso_or_mad_scale(errors_pt, n_pt);
    estimated_scale_ls = 0.5 * estimated_scale_pt;
  } else {
    estimated_scale_ls = hso_or_mad_scale(errors_ls, n_ls);
    estimated_scale_pt = 2 * estimated_scale_ls;
  }
  double estimated_scale = estimated_scale_pt;
  const float k = 1.345f; /* HuberWeightFunction::DEFAULT_K, robust_cost.cpp:129 */

  int idx_host = 0;
  for (int i = 0; i < n; i++) {
    if (!F[i].has_point) continue;
    resid_t r;
    residual(&F[i], v_host + 3 * idx_host, &T, job->poses_f_w, &r);
    if (F[i].type == HSO_FTR_EDGELET) {
      const double error_ls = F[i].grad[0] * r.e[0] + F[i].grad[1] * r.e[1];
      double weight = hso_or_huber_weight(k, (float)(fabs(error_ls) / estimated_scale_ls));
      if (F[i].temporary) weight *= 0.5;
      chi2 += error_ls * error_ls * weight;
    } else {
      const double error_pt = sqrt(r.e[0] * r.e[0] + r.e[1] * r.e[1]);
      double weight = hso_or_huber_weight(k, (float)(error_pt / estimated_scale_pt));
      if (F[i].temporary) weight *= 0.5;
      chi2 += error_pt * error_pt * weight;
    }
    ++idx_host;
  }
  int num_obs = n_pt + n_ls;
  double A[36], b[6];
  memset(A, 0, sizeof(A)); memset(b, 0, sizeof(b));

  for (int iter = 0; iter < job->n_iter; iter++) {
    out->iters = iter + 1;
    rho = 0;
    n_trials = 0;
    do {
      hso_se3 T_new = T;
      double new_chi2 = 0.0;
      memset(A, 0, sizeof(A)); memset(b, 0, sizeof(b));
      idx_host = 0;
      for (int i = 0; i < n; i++) {
        if (!F[i].has_point) continue;
        resid_t r;
        residual(&F[i], v_host + 3 * idx_host, &T, job->poses_f_w, &r);
        double J[12];
        jac_xyz2uv(r.pTarget, J);
        const double sqrt_inv_cov = 1.0 / (1 << F[i].level);
        for (int q = 0; q < 12; q++) J[q] *= sqrt_inv_cov;
        if (F[i].type == HSO_FTR_EDGELET) {
          double Je[6];
          for (int q = 0; q < 6; q++) Je[q] = F[i].grad[0] * J[q] + F[i].grad[1] * J[6 + q];
          const double e_edge = F[i].grad[0] * r.e[0] + F[i].grad[1] * r.e[1];
          double weight = hso_or_huber_weight(k, (float)(fabs(e_edge) / estimated_scale_ls));
          if (F[i].temporary) weight *= 0.5;
          for (int a = 0; a < 6; a++) {
            for (int c = 0; c < 6; c++) A[a * 6 + c] += (Je[a] * Je[c]) * weight;
            b[a] -= (Je[a] * e_edge) * weight;
          }
        } else {
          double weight = hso_or_huber_weight(k, (float)(sqrt(r.e[0] * r.e[0] + r.e[1] * r.e[1]) / estimated_scale_pt));
          if (F[i].temporary) weight *= 0.5;
          for (int a = 0; a < 6; a++) {
            for (int c = 0; c < 6; c++) A[a * 6 + c] += (J[a] * J[c] + J[6 + a] * J[6 + c]) * weight;
            b[a] -= (J[a] * r.e[0] + J[6 + a] * r.e[1]) * weight;
          }
        }
        ++idx_host;
      }
      for (int a = 0; a < 6; a++) A[a * 6 + a] += A[a * 6 + a] * mu;
      double dT[6];
      hso_or_ldlt_solve(A, b, 6, dT);
      out->n_trials_total++;
      if (!isnan(dT[0])) {
        hso_se3 E;
        hso_or_se3_exp(dT, &E);
        hso_or_se3_mul(&E, &T, &T_new);
        idx_host = 0;
        for (int i = 0; i < n; i++) {
          if (!F[i].has_point) continue;
          resid_t r;
          residual(&F[i], v_host + 3 * idx_host, &T_new, job->poses_f_w, &r);
          if (F[i].type == HSO_FTR_EDGELET) {
            const double error_ls = F[i].grad[0] * r.e[0] + F[i].grad[1] * r.e[1];
            double weight = hso_or_huber_weight(k, (float)(fabs(error_ls) / estimated_scale_ls));
            if (F[i].temporary) weight *= 0.5;
            new_chi2 += error_ls * error_ls * weight;
          } else {
            const double error_pt = sqrt(r.e[0] * r.e[0] + r.e[1] * r.e[1]);
            double weight = hso_or_huber_weight(k, (float)(error_pt / estimated_scale_pt));
            if (F[i].temporary) weight *= 0.5;
            new_chi2 += error_pt * error_pt * weight;
          }
          ++idx_host;
        }
        rho = chi2 - new_chi2;
        hso_or_margin_note(HSO_M_POSE_RHO, rho / (chi2 > 1e-300 ? chi2 : 1e-300));
      } else
        rho = -1;
      if (rho > 0) {
        T = T_new;
        chi2 = new_chi2;
        double nm = -1;
        for (int q = 0; q < 6; q++) { const double a = fabs(dT[q]); if (a > nm) nm = a; }
        stop = nm <= 0.0000000001; /* hso::EPS, include/hso/global.h:103 */
        mu *= fmax(1. / 3., fmin(1. - pow(2 * rho - 1, 3), 2. / 3.));
        nu = 2.;
      } else {
        mu *= nu;
        nu *= 2.;
        if (mu < 0.0001) mu = 0.0001;
        ++n_trials;
        if (n_trials >= n_trials_max) stop = 1;
      }
    } while (!(rho > 0 || stop));
    if (stop) break;
  }

  /* Cov_ = pixel_variance * (A * errorMultiplier2^2).inverse(), :692 */
  {
    double As[36];
    const double s2 = pow(em2, 2);
    for (int q = 0; q < 36; q++) As[q] = A[q] * s2;
    invert6(As, out->cov);
    for (int q = 0; q < 36; q++) out->cov[q] = 1.0f * out->cov[q];
  }
  const float reproj_thresh_scaled_pt = (n < 80) ? sqrt(5.991) / em2 : job->reproj_thresh / em2;
  const float reproj_thresh_scaled_ls = 1.3 / em2;
  int n_deleted_refs = 0;
  idx_host = 0;
  for (int i = 0; i < n; i++) {
    if (!F[i].has_point) continue;
    resid_t r;
    residual(&F[i], v_host + 3 * idx_host, &T, job->poses_f_w, &r);
    if (F[i].type == HSO_FTR_EDGELET) {
      const double error_ls = F[i].grad[0] * r.e[0] + F[i].grad[1] * r.e[1];
      if (fabs(error_ls) > reproj_thresh_scaled_ls) { ++n_deleted_refs; if (outlier_mask) outlier_mask[i] = 1; }
      chi2_vec_final[n_final++] = error_ls * error_ls;
    } else {
      const float error_pt = sqrt(r.e[0] * r.e[0] + r.e[1] * r.e[1]);
      if (error_pt > reproj_thresh_scaled_pt) { ++n_deleted_refs; if (outlier_mask) outlier_mask[i] = 1; }
      chi2_vec_final[n_final++] = error_pt * error_pt;
    }
    ++idx_host;
  }
  out->error_init = 0.0; out->error_final = 0.0;
  if (n_init > 0) out->error_init = sqrt(hso_or_median_d(chi2_vec_init, n_init)) * em2;
  if (n_final > 0) out->error_final = sqrt(hso_or_median_d(chi2_vec_final, n_final)) * em2;
  estimated_scale *= em2;
  num_obs -= n_deleted_refs;
  out->T_f_w = T;
  out->estimated_scale = estimated_scale;
  out->num_obs = num_obs;
  out->n_deleted = n_deleted_refs;
  out->error_in_px = out->error_final < 1.5 ? 1.0 : 1.5 / out->error_final;
done:
  free(chi2_vec_init); free(chi2_vec_final); free(errors_pt); free(errors_ls); free(v_host);
}
