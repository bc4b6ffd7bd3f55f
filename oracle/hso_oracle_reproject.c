/*
 * hso_oracle_reproject.c — candidate generation of the Reprojector restated:
 * Reprojector::reprojectPoint (src/reprojector.cpp:504-529), Point::getCloseViewObs
 * (src/point.cpp:116-136) and the inputs Matcher::findMatchDirect derives from the chosen
 * observation (src/matcher.cpp:270-319).  TEST INFRASTRUCTURE (see hso_oracle.h); parity unpinned.
 */
#include "hso_oracle.h"
#include <math.h>
#include <string.h>

static void frame_pos(const hso_se3* T_f_w, double p[3])       /* Frame::pos(), include/hso/frame.h:142 */
{
  hso_se3 inv;
  hso_or_se3_inverse(T_f_w, &inv);
  p[0] = inv.t[0]; p[1] = inv.t[1]; p[2] = inv.t[2];
}

static void normalize3(double v[3])                            /* Eigen normalize(): v /= norm */
{
  const double n = sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
  v[0] /= n; v[1] /= n; v[2] /= n;
}

/* reprojectPoint, :504-529: returns 1 and fills px, cell when the point lands inside the frame */
int hso_or_reproject_point(const hso_camera* cam, const hso_se3* T_cur_w, const hso_se3* T_host_w, const double host_f[3], double idist,
                           int cell_size, int grid_n_cols, double px[2], int* cell)
{
  const double s = 1.0 / idist;
  const double pHost[3] = {host_f[0] * s, host_f[1] * s, host_f[2] * s};
  hso_se3 inv, T;
  hso_or_se3_inverse(T_host_w, &inv);
  hso_or_se3_mul(T_cur_w, &inv, &T);
  double pTarget[3];
  hso_or_se3_apply(&T, pHost, pTarget);
  if (pTarget[2] < 0.00001) return 0;
  hso_or_world2cam(cam, pTarget, px);
  const int ix = (int)px[0], iy = (int)px[1];                  /* px.cast<int>() */
  if (!(ix >= 8 && ix < cam->width - 8 && iy >= 8 && iy < cam->height - 8)) return 0;   /* isInFrame(..., 8) */
  *cell = (int)(px[1] / cell_size) * grid_n_cols + (int)(px[0] / cell_size);
  return 1;
}

/* getCloseViewObs, src/point.cpp:116-136: index of the chosen observation, or -1 when the best
 * one is more than 60 degrees away (the function returns false) or the list is empty */
int hso_or_close_view_obs(const double cur_pos[3], const double pos[3], const hso_kf* kfs, const hso_obs* obs, int n_obs)
{
  if (n_obs <= 0) return -1;
  double od[3] = {cur_pos[0] - pos[0], cur_pos[1] - pos[1], cur_pos[2] - pos[2]};
  normalize3(od);
  int min_it = 0;
  double min_cos_angle = 0;
  for (int i = 0; i < n_obs; i++) {
    double fp[3];
    frame_pos(&kfs[obs[i].kf].T_f_w, fp);
    double d[3] = {fp[0] - pos[0], fp[1] - pos[1], fp[2] - pos[2]};
    normalize3(d);
    const double c = od[0] * d[0] + od[1] * d[1] + od[2] * d[2];
    if (c > min_cos_angle) { min_cos_angle = c; min_it = i; }
  }
  if (min_cos_angle < 0.5) return -1;
  return min_it;
}

/* what findMatchDirect reads off the chosen observation, src/matcher.cpp:288-319 */
void hso_or_reproject_make_job(const hso_se3* T_cur_w, double cur_exposure_time, int cur_keyframe_id, const hso_kf* kfs,
                               const hso_map_point* pt, const hso_obs* ref, const double px_cur[2], hso_align_job* j)
{
  memset(j, 0, sizeof(*j));
  const hso_kf* kf = &kfs[ref->kf];
  j->ref_frame_id = kf->frame_id;
  j->ref_level = ref->level;
  j->type = ref->type;
  j->px_ref[0] = ref->px[0]; j->px_ref[1] = ref->px[1];
  j->f_ref[0] = ref->f[0]; j->f_ref[1] = ref->f[1]; j->f_ref[2] = ref->f[2];
  j->grad[0] = ref->grad[0]; j->grad[1] = ref->grad[1];
  if (ref->kf == pt->host_kf) {
    j->depth = 1.0 / pt->idist;
  } else {
    double fp[3];
    frame_pos(&kf->T_f_w, fp);
    const double d[3] = {fp[0] - pt->pos[0], fp[1] - pt->pos[1], fp[2] - pt->pos[2]};
    j->depth = sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
  }
  hso_se3 inv;
  hso_or_se3_inverse(&kf->T_f_w, &inv);
  hso_or_se3_mul(T_cur_w, &inv, &j->T_cur_ref);
  j->px_cur[0] = px_cur[0]; j->px_cur[1] = px_cur[1];
  j->exposure_rat = (float)(cur_exposure_time / kf->exposure_time);
  j->kf_gap_lt4 = (cur_keyframe_id - kf->keyframe_id) < 4;
}
