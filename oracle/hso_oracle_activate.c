/*
 * hso_oracle_activate.c — seed activation: re-match the seed in the frames that observed it,
 * gate on the mean drift, then refine the inverse depth with a 1-D Levenberg-Marquardt.
 * TEST INFRASTRUCTURE (see hso_oracle.h).  Follows DepthFilter::activatePoint
 * (src/depth_filter.cpp:729-851), DepthFilter::seedOptimizer (:853-1073),
 * Matcher::findMatchSeed's parallax test (src/matcher.cpp:444-449),
 * Point::jacobian_id2uv (include/hso/point.h:174-184) and MADScaleEstimator::compute
 * (src/vikit/robust_cost.cpp:67-74).  Quirks kept: the Huber weight of an edgelet residual is
 * formed from fabsf() of the double residual (:900,:944,:988); `stop` compares fabsf(step)
 * with 1e-5*new_id (:1037); the per-frame inverse-depth Jacobian uses the linearisation point's
 * projection (:949,:963).
 */
#include "hso_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

static void id2uv(const double p[3], const hso_se3* Tth, double idH, const double fH[3], double J[2])
{
  const double proj0 = p[0] / p[2], proj1 = p[1] / p[2];
  double R[9];
  hso_or_so3_matrix(Tth->q, R);
  const double Rf2 = R[6] * fH[0] + R[7] * fH[1] + R[8] * fH[2];
  J[0] = -(Tth->t[0] - proj0 * Tth->t[2]) / (Rf2 + Tth->t[2] * idH);
  J[1] = -(Tth->t[1] - proj1 * Tth->t[2]) / (Rf2 + Tth->t[2] * idH);
}

static double seed_energy(const hso_seed* s, int n, const hso_se3* Tths, const double* obs, const double* normals,
                          double id, double huberTH)
{
  double E = 0;
  const double sc = 1.0 / id;
  const double pHost[3] = { s->f[0] * sc, s->f[1] * sc, s->f[2] * sc };
  for (int i = 0; i < n; i++) {
    double pT[3];
    hso_or_se3_apply(&Tths[i], pHost, pT);
    const double r0 = obs[2 * i] - pT[0] / pT[2], r1 = obs[2 * i + 1] - pT[1] / pT[2];
    if (s->type == HSO_FTR_EDGELET) {
      const double re = normals[2 * i] * r0 + normals[2 * i + 1] * r1;
      const double hw = fabsf((float)re) < huberTH ? 1 : huberTH / fabsf((float)re);
      E += re * re * hw;
    } else {
      const double rd = sqrt(r0 * r0 + r1 * r1);
      const double hw = rd < huberTH ? 1 : huberTH / rd;
      E += rd * rd * hw;
    }
  }
  return E;
}

/* DepthFilter::seedOptimizer, src/depth_filter.cpp:853-1073 */
static void seed_optimizer(const hso_seed* s, int n, const hso_se3* Tths, const double* obs, const double* normals,
                           hso_activate_out* o)
{
  double oldEnergy = 0.0, rho = 0, mu = 0.1, nu = 2.0;
  int stop = 0, n_trials = 0;
  const int n_trials_max = 5;
  double old_id = s->mu;
  float errors[HSO_ACTIVATE_MAX_TARGETS];
  {
    const double sc = 1.0 / old_id;
    const double pHost[3] = { s->f[0] * sc, s->f[1] * sc, s->f[2] * sc };
    for (int i = 0; i < n; i++) {
      double pT[3];
      hso_or_se3_apply(&Tths[i], pHost, pT);
      const double r0 = obs[2 * i] - pT[0] / pT[2], r1 = obs[2 * i + 1] - pT[1] / pT[2];
      if (s->type == HSO_FTR_EDGELET) errors[i] = fabs(normals[2 * i] * r0 + normals[2 * i + 1] * r1);
      else errors[i] = sqrt(r0 * r0 + r1 * r1);
    }
  }
  const double huberTH = hso_or_mad_scale(errors, n);
  o->huber = huberTH;
  oldEnergy = seed_energy(s, n, Tths, obs, normals, old_id, huberTH);
  double H = 0, b = 0;
  int iter;
  for (iter = 0; iter < 5; ++iter) {
    n_trials = 0;
    do {
      double new_id = old_id, newEnergy = 0;
      H = b = 0;
      const double sc = 1.0 / old_id;
      const double pHost[3] = { s->f[0] * sc, s->f[1] * sc, s->f[2] * sc };
      for (int i = 0; i < n; i++) {
        double pT[3], J[2];
        hso_or_se3_apply(&Tths[i], pHost, pT);
        const double r0 = obs[2 * i] - pT[0] / pT[2], r1 = obs[2 * i + 1] - pT[1] / pT[2];
        id2uv(pT, &Tths[i], old_id, s->f, J);
        if (s->type == HSO_FTR_EDGELET) {
          const double re = normals[2 * i] * r0 + normals[2 * i + 1] * r1;
          const double hw = fabsf((float)re) < huberTH ? 1 : huberTH / fabsf((float)re);
          const double JE = normals[2 * i] * J[0] + normals[2 * i + 1] * J[1];
          H += JE * JE * hw;
          b -= JE * re * hw;
        } else {
          const double rd = sqrt(r0 * r0 + r1 * r1);
          const double hw = rd < huberTH ? 1 : huberTH / rd;
          H += (J[0] * J[0] + J[1] * J[1]) * hw;
          b -= (J[0] * r0 + J[1] * r1) * hw;
        }
      }
      H *= 1.0 + mu;
      const double step = b / H;
      if (!isnan(step)) {
        new_id = old_id + step;
        newEnergy = seed_energy(s, n, Tths, obs, normals, new_id, huberTH);
        rho = oldEnergy - newEnergy;
      } else {
        rho = -1;
      }
      if (rho > 0) {
        oldEnergy = newEnergy;
        old_id = new_id;
        o->opt_id = new_id;
        stop = fabsf((float)step) < 0.00001 * new_id;
        { const double c = 1. - pow(2 * rho - 1, 3); const double m = c < 2. / 3. ? c : 2. / 3.; mu *= (1. / 3. > m ? 1. / 3. : m); }
        nu = 2.;
      } else {
        mu *= nu;
        nu *= 2.;
        ++n_trials;
        if (n_trials >= n_trials_max) stop = 1;
      }
    } while (!(rho > 0 || stop));
    if (stop) break;
  }
  o->energy = oldEnergy;
  o->n_iter = iter < 5 ? iter + 1 : 5;
}

/* DepthFilter::activatePoint for one seed.  tg_pyr: n_tg*5 level pointers, tg_gx / tg_gy: n_tg*3. */
void hso_or_seed_activate(const hso_camera* cam, const hso_seed* s, const hso_activate_target* tg, int n_tg,
                          const uint8_t* const ref_pyr[HSO_N_PYR_LEVELS], const uint8_t* const* tg_pyr,
                          const int16_t* const* tg_gx, const int16_t* const* tg_gy, int w, int h,
                          int n_mean_converge_frame, hso_activate_out* o, hso_align_out* match_out)
{
  memset(o, 0, sizeof(*o));
  o->is_valid = -1;
  o->opt_id = s->mu;                                              /* :731 */
  const double sc = 1.0 / s->mu;
  const double pHost[3] = { s->f[0] * sc, s->f[1] * sc, s->f[2] * sc };
  hso_se3 host_inv;
  hso_or_se3_inverse(&s->T_ref_w, &host_inv);
  int is_target[HSO_ACTIVATE_MAX_TARGETS];
  double px_t[2 * HSO_ACTIVATE_MAX_TARGETS];
  hso_se3 Tth_all[HSO_ACTIVATE_MAX_TARGETS];
  int n_targets = 0;
  for (int i = 0; i < n_tg; i++) {
    is_target[i] = 0;
    if (match_out) memset(&match_out[i], 0, sizeof(hso_align_out));
    hso_or_se3_mul(&tg[i].T_f_w, &host_inv, &Tth_all[i]);
    double pT[3];
    hso_or_se3_apply(&Tth_all[i], pHost, pT);
    if (pT[2] < 0.0001) continue;
    double px[2];
    hso_or_world2cam(cam, pT, px);
    const int ox = (int)px[0], oy = (int)px[1];
    if (!(ox >= 8 && ox < w - 8 && oy >= 8 && oy < h - 8)) continue;
    is_target[i] = 1; px_t[2 * i] = px[0]; px_t[2 * i + 1] = px[1];
    n_targets++;
  }
  o->n_targets = n_targets;
  float n_frame_thresh = n_mean_converge_frame * 0.7;
  if (n_frame_thresh > 8) n_frame_thresh = 8;
  if (n_frame_thresh < 3) n_frame_thresh = 3;
  if (n_targets < n_frame_thresh) return;

  double distMean = 0;
  hso_se3 Tths[HSO_ACTIVATE_MAX_TARGETS];
  double obs[2 * HSO_ACTIVATE_MAX_TARGETS], normals[2 * HSO_ACTIVATE_MAX_TARGETS];
  int n_res = 0;
  /* seed_pos, ref_dir of findMatchSeed, :445-446 */
  double seed_pos[3], host_pos[3] = { host_inv.t[0], host_inv.t[1], host_inv.t[2] };
  hso_or_se3_apply(&host_inv, pHost, seed_pos);
  double ref_dir[3] = { host_pos[0] - seed_pos[0], host_pos[1] - seed_pos[1], host_pos[2] - seed_pos[2] };
  { const double n = sqrt(ref_dir[0] * ref_dir[0] + ref_dir[1] * ref_dir[1] + ref_dir[2] * ref_dir[2]);
    ref_dir[0] /= n; ref_dir[1] /= n; ref_dir[2] /= n; }
  for (int i = 0; i < n_tg; i++) {
    if (!is_target[i]) continue;
    hso_se3 tinv;
    hso_or_se3_inverse(&tg[i].T_f_w, &tinv);
    double cur_dir[3] = { tinv.t[0] - seed_pos[0], tinv.t[1] - seed_pos[1], tinv.t[2] - seed_pos[2] };
    { const double n = sqrt(cur_dir[0] * cur_dir[0] + cur_dir[1] * cur_dir[1] + cur_dir[2] * cur_dir[2]);
      cur_dir[0] /= n; cur_dir[1] /= n; cur_dir[2] /= n; }
    const double cos_angle = ref_dir[0] * cur_dir[0] + ref_dir[1] * cur_dir[1] + ref_dir[2] * cur_dir[2];
    if (cos_angle < 0.5) continue;
    hso_align_job job;
    memset(&job, 0, sizeof(job));
    job.ref_frame_id = s->ref_frame_id; job.ref_level = s->level; job.type = s->type;
    job.px_ref[0] = s->px[0]; job.px_ref[1] = s->px[1];
    job.f_ref[0] = s->f[0]; job.f_ref[1] = s->f[1]; job.f_ref[2] = s->f[2];
    job.depth = 1. / s->mu;
    job.grad[0] = s->grad[0]; job.grad[1] = s->grad[1];
    job.T_cur_ref = Tth_all[i];
    job.px_cur[0] = px_t[2 * i]; job.px_cur[1] = px_t[2 * i + 1];
    job.exposure_rat = tg[i].exposure / s->ref_exposure;
    job.kf_gap_lt4 = 1;
    hso_align_out mo;
    hso_or_find_match_seed(cam, &job, ref_pyr, tg_pyr + 5 * i, tg_gx + 3 * i, tg_gy + 3 * i, w, h, &mo);
    if (match_out) match_out[i] = mo;
    if (!mo.success) continue;
    const double d0 = px_t[2 * i] - mo.px_cur[0], d1 = px_t[2 * i + 1] - mo.px_cur[1];
    if (s->type != HSO_FTR_EDGELET) {
      double err = sqrt(d0 * d0 + d1 * d1);
      err /= (1 << mo.search_level);
      distMean += err;
    } else {
      double n0 = mo.A_cur_ref[0] * s->grad[0] + mo.A_cur_ref[1] * s->grad[1];
      double n1 = mo.A_cur_ref[2] * s->grad[0] + mo.A_cur_ref[3] * s->grad[1];
      const double nn = sqrt(n0 * n0 + n1 * n1);
      n0 /= nn; n1 /= nn;
      normals[2 * n_res] = n0; normals[2 * n_res + 1] = n1;
      double err = fabs(n0 * d0 + n1 * d1);
      err /= (1 << mo.search_level);
      distMean += err;
    }
    double f[3];
    hso_or_cam2world(cam, mo.px_cur[0], mo.px_cur[1], f);
    obs[2 * n_res] = f[0] / f[2]; obs[2 * n_res + 1] = f[1] / f[2];
    Tths[n_res] = Tth_all[i];
    n_res++;
  }
  o->n_matched = n_res;
  if (n_res < n_frame_thresh) return;
  distMean /= n_res;
  o->dist_mean = distMean;
  const int edge = s->type == HSO_FTR_EDGELET;
  if ((!edge && distMean > 3.2) || (edge && distMean > 2.5)) { o->is_valid = 0; return; }
  o->is_valid = 1;
  if ((!edge && distMean > 2.5) || (edge && distMean > 2.0)) return;
  seed_optimizer(s, n_res, Tths, obs, normals, o);
  o->activated = 1;
}
