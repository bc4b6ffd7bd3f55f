/* hso_oracle_batch.c — batch forms of the per-item restatements, for bench.py's cpu_baseline leg ONLY (test
 * infrastructure like everything under oracle/): the per-item Python wrappers cost more interpreter time than the
 * functions they call, which would flatter the GPU.  No new arithmetic: loops over hso_or_find_match_direct /
 * hso_or_seed_observe, optionally on several threads the way the reference's depth filter runs its seed updates
 * (include/hso/IndexThreadReduce.h:27: 4 worker threads, static index ranges; DepthFilter::updateSeeds,
 * src/depth_filter.cpp:420-470). */
#include <pthread.h>
#include <stdlib.h>
#include "hso_oracle.h"

/* reprojected map points against the current frame: job i matches against the pyramid of keyframe job_kf[i] */
void hso_or_find_match_direct_batch(const hso_camera* cam, const hso_align_job* jobs, const int* job_kf, int n,
                                    const uint8_t* const* kf_pyrs /* [n_kf][HSO_N_PYR_LEVELS] */,
                                    const uint8_t* const cur_pyr[HSO_N_PYR_LEVELS], const int16_t* const cur_gx[HSO_N_SOBEL_LEVELS],
                                    const int16_t* const cur_gy[HSO_N_SOBEL_LEVELS], int w, int h, hso_align_out* out)
{
  for (int i = 0; i < n; i++)
    hso_or_find_match_direct(cam, &jobs[i], kf_pyrs + (size_t)job_kf[i] * HSO_N_PYR_LEVELS, cur_pyr, cur_gx, cur_gy, w, h, &out[i]);
}

typedef struct {
  const hso_camera* cam; const hso_seed* seeds; const hso_se3* T; double exposure, px_error_angle;
  const uint8_t* const* ref_pyr; const uint8_t* const* cur_pyr; const int16_t* const* gx; const int16_t* const* gy;
  int w, h, first, last; hso_seed_out* out;
} seed_range;

static void* seed_worker(void* p)
{
  const seed_range* r = (const seed_range*)p;
  for (int i = r->first; i < r->last; i++)
    hso_or_seed_observe(r->cam, &r->seeds[i], r->T, r->exposure, r->px_error_angle, r->ref_pyr, r->cur_pyr, r->gx, r->gy, r->w, r->h, &r->out[i]);
  return NULL;
}

/* all seeds hosted in one reference frame, observed in one current frame; n_threads = 1 or the reference's 4 */
void hso_or_seed_observe_batch(const hso_camera* cam, const hso_seed* seeds, int n, const hso_se3* cur_T_f_w, double cur_exposure,
                               double px_error_angle, const uint8_t* const ref_pyr[HSO_N_PYR_LEVELS],
                               const uint8_t* const cur_pyr[HSO_N_PYR_LEVELS], const int16_t* const cur_gx[HSO_N_SOBEL_LEVELS],
                               const int16_t* const cur_gy[HSO_N_SOBEL_LEVELS], int w, int h, hso_seed_out* out, int n_threads)
{
  if (n_threads < 1) n_threads = 1;
  if (n_threads > 16) n_threads = 16;
  seed_range r[16];
  pthread_t th[16];
  const int step = (n + n_threads - 1) / n_threads;       /* IndexThreadReduce: static ranges of equal size */
  for (int t = 0; t < n_threads; t++) {
    r[t] = (seed_range){ cam, seeds, cur_T_f_w, cur_exposure, px_error_angle, ref_pyr, cur_pyr, cur_gx, cur_gy, w, h,
                         t * step < n ? t * step : n, (t + 1) * step < n ? (t + 1) * step : n, out };
  }
  if (n_threads == 1) { seed_worker(&r[0]); return; }
  for (int t = 0; t < n_threads; t++) pthread_create(&th[t], NULL, seed_worker, &r[t]);
  for (int t = 0; t < n_threads; t++) pthread_join(th[t], NULL);
}
