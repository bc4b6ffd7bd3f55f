/*
 * hso_oracle.h — CPU restatement of the reference's per-frame numeric hot path.
 *
 * TEST INFRASTRUCTURE, NOT PRODUCT.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load this library; the product path
 * (hso_amd/, libhso_gpu.so) never links, imports or calls it.
 *
 * PARITY PINNING: the reference (luodongting/HSO) ships no tests, golden
 * vectors or fixtures for this path and cannot be compiled here (Eigen3, OpenCV,
 * Boost absent), so the restatement below is *unpinned by the reference* except
 * for the three parts whose reference sources build standalone (oracle/_ref, see
 * Makefile): the robust-cost functions (src/vikit/robust_cost.cpp), the FAST-9
 * corner detector (thirdparty/fast, hso_oracle_fast.c) and the ZMNCC patch score
 * (include/hso/vikit/patch_score.h, hso_oracle_seed.c).  Every
 * function cites the reference file:line it follows so it can be diffed by eye.
 * Third-party arithmetic that the reference pulls from outside its tree (Eigen
 * LDLT / Quaternion, OpenCV Sobel) is restated from the published algorithms.
 *
 * Floating-point contract: compiled with -ffp-contract=off and no fast-math, so
 * each C expression is evaluated exactly as written (the reference itself is
 * built with -O3 -march=native, i.e. defined only up to FMA contraction).
 */
#ifndef HSO_ORACLE_H
#define HSO_ORACLE_H

#include "../include/hso_gpu.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- decision margins (SURVEY.md App. C "decision robustness") ----
 * Every comparison that gates control flow in the restatement reports how far its operands were apart; each field
 * keeps the SMALLEST distance seen since the last reset.  A parity test excuses a differing decision only when the
 * margin of the comparison that produces it is below 10x the stated tolerance; everything else must match exactly. */
typedef struct hso_or_margins {
  double lk_update;    /* LK convergence: min over iterations of | |update|^2 - min_update^2 | / min_update^2 (feature_alignment.cpp:296,593) */
  double lk_chi2;      /* | chi2 - 1000 * patch_area | / (1000 * patch_area) (:303,:600) */
  double ncc;          /* | ncc - threshold | of checkNCC (matcher.cpp:379-404) */
  double normal;       /* | normal . gradient - threshold | of checkNormal (:406-440) */
  double jump;         /* | |px - px_orig| - 20 | (:369-370) */
  double zmncc_best;   /* | zmncc_best - 0.8 | (matcher.cpp:1003) */
  double zmncc_ambig;  /* | 1.5 * zmncc_second - zmncc_best | where the index gap allows the test (:1000) */
  double zmncc_order;  /* min over epipolar samples of | zmncc - zmncc_best |, | zmncc - zmncc_second | at the comparisons (:985-996) */
  double klt_energy;   /* | bestEnergy - 650 * 64 | / (650 * 64) (:1449,:1604) */
  double klt_step;     /* min over KLT iterations of the distance of the step test to its bound (:1435,:1590) */
  double klt_accept;   /* min over KLT iterations of | newEnergy - bestEnergy | / bestEnergy (step accepted or halved) */
  double pose_rho;     /* pose LM: min over trials of |rho| / max(chi2, 1e-300) (pose_optimizer.cpp:644-670) */
  double march_end;    /* epipolar march: min over steps of the distance of the position to px_close along an axis the loop
                          condition tests (matcher.cpp:893-900): ~0 = the last step lands ON the end point and the step count is
                          decided by rounding (segments padded to exactly 4 units, :880-886) */
  double track_accept; /* CoarseTracker::run: min over LM iterations of | energy_new - energy_old | / energy_old at the accept test
                          (CoarseTracker.cpp:143); the energies are float quotients, so a gap near 1e-7 is decided by rounding */
} hso_or_margins;
void hso_or_margins_reset(void);
void hso_or_margins_get(hso_or_margins* out);
void hso_or_margin_note(int field, double v);   /* internal: field = index of the double above */
enum { HSO_M_LK_UPDATE, HSO_M_LK_CHI2, HSO_M_NCC, HSO_M_NORMAL, HSO_M_JUMP, HSO_M_ZMNCC_BEST, HSO_M_ZMNCC_AMBIG, HSO_M_ZMNCC_ORDER,
       HSO_M_KLT_ENERGY, HSO_M_KLT_STEP, HSO_M_KLT_ACCEPT, HSO_M_POSE_RHO, HSO_M_MARCH_END, HSO_M_TRACK_ACCEPT, HSO_M_COUNT };

/* ---- Sophus SE3 / SO3 (thirdparty/Sophus/sophus/{se3,so3}.cpp) ---- */
void hso_or_se3_identity(hso_se3* T);
void hso_or_se3_mul(const hso_se3* a, const hso_se3* b, hso_se3* out);      /* se3.cpp:59-66 */
void hso_or_se3_inverse(const hso_se3* a, hso_se3* out);                    /* se3.cpp:76-83 */
void hso_or_se3_apply(const hso_se3* T, const double p[3], double out[3]);  /* se3.cpp:91-95 */
void hso_or_se3_exp(const double upsilon_omega[6], hso_se3* out);           /* se3.cpp:170-196 */
void hso_or_se3_log(const hso_se3* T, double out[6]);                       /* se3.cpp:198-220 */
void hso_or_so3_matrix(const double q[4], double R[9]);                     /* so3.cpp:98-102 (Eigen toRotationMatrix) */

/* ---- small dense algebra (Eigen, restated) ---- */
/* Eigen::LDLT<Matrix<double,n,n>>(A).solve(b), pivoted, lower triangle of A read. n <= 8. */
void hso_or_ldlt_solve(const double* A, const double* b, int n, double* x);
float hso_or_median_f(float* data, int n);   /* hso::getMedian, include/hso/vikit/math_utils.h:119-126 (upper median; permutes data) */
double hso_or_median_d(double* data, int n);

/* ---- robust cost (src/vikit/robust_cost.cpp) — pinned against oracle/_ref ---- */
float hso_or_huber_weight(float k, float t);                    /* robust_cost.cpp:141-148 */
float hso_or_tukey_weight(float b, float x);                    /* :93-103 */
float hso_or_tdist_weight(float dof, float x);                  /* :117-121 */
float hso_or_mad_scale(const float* errors, int n);             /* :67-74 */
float hso_or_tdist_scale(float dof, const float* errors, int n);/* :38-63 */

/* ---- camera (src/camera.cpp) ---- */
void hso_or_world2cam(const hso_camera* cam, const double xyz[3], double px[2]); /* camera.cpp:94-125,196-221,295-303 */
void hso_or_cam2world(const hso_camera* cam, double u, double v, double f[3]);    /* camera.cpp:67-87,171-194,283-286 */
double hso_or_error_multiplier2(const hso_camera* cam);                          /* fxy_mean_, camera.cpp:59 */

/* ---- frame: pyramid, Sobel, stats (src/frame.cpp, src/vikit/vision.cpp) ---- */
void hso_or_half_sample(const uint8_t* in, int w, int h, uint8_t* out);       /* vision.cpp:70-108 incl. SSE2 path :19-44 */
void hso_or_pyramid_dims(int w, int h, int level, int* lw, int* lh);           /* frame.cpp:302-312 */
void hso_or_resize_linear_8u(const uint8_t* src, int sw, int sh, uint8_t* dst, int dw, int dh); /* cv::resize INTER_LINEAR, frame.cpp:311 */
/* levels[] must hold 5 buffers of the sizes hso_or_pyramid_dims reports; levels[0] is copied from img */
int hso_or_create_pyramid(const uint8_t* img, int w, int h, uint8_t* const levels[HSO_N_PYR_LEVELS]); /* frame.cpp:296-314 */
void hso_or_sobel5(const uint8_t* img, int w, int h, int16_t* gx, int16_t* gy); /* cv::Sobel(CV_16S, k=5, BORDER_REPLICATE), frame.cpp:218-219 */
void hso_or_frame_stats(const uint8_t* img0, const int16_t* gx0, const int16_t* gy0, int w, int h,
                        hso_frame_stats* out);                                /* frame.cpp:223-245 */

/* ---- CoarseTracker (src/CoarseTracker.cpp) ---- */
void hso_or_make_depth_ref(const hso_depth_ref_in* in, int n, const hso_se3* poses_f_w,
                           const hso_se3* T_ref_w, double* dist_out);         /* :210-240 */

typedef struct hso_or_tracker hso_or_tracker;
/* ref_pyr/cur_pyr: 5 level pointers of a w x h level-0 frame */
hso_or_tracker* hso_or_tracker_create(const hso_camera* cam, const hso_track_params* p,
                                      const uint8_t* const ref_pyr[HSO_N_PYR_LEVELS],
                                      const uint8_t* const cur_pyr[HSO_N_PYR_LEVELS],
                                      int w, int h, const hso_ref_feat* feats, int n);
void hso_or_tracker_destroy(hso_or_tracker* t);
/* set m_level, pattern, run precomputeReferencePatches (:416-497) */
void hso_or_tracker_set_level(hso_or_tracker* t, int level);
/* selectRobustFunctionLevel (:530-644); returns errors.size() */
int hso_or_tracker_select(hso_or_tracker* t, const hso_se3* T, float exposure_rat,
                          float* huber, float* outlier, float* abs_err_out);
void hso_or_tracker_set_thresholds(hso_or_tracker* t, float huber, float outlier);
/* computeResiduals (:242-414) + computeGS (:499-525) */
void hso_or_tracker_eval(hso_or_tracker* t, const hso_se3* T, float exposure_rat,
                         hso_eval_out* out);
double hso_or_tracker_energy_f64(const hso_or_tracker* t); /* diagnostics, see .c */
void hso_or_tracker_decide_on_f64_sum(hso_or_tracker* t, int on); /* diagnostics, see .c */
/* read-back of m_ref_patch_cache (n*PATCH_AREA) and m_visible_fts */
void hso_or_tracker_get_cache(const hso_or_tracker* t, float* ref_patch, uint8_t* visible,
                              int* patch_area);
/* CoarseTracker::run (:51-208) minus the frame write-back */
void hso_or_tracker_run(hso_or_tracker* t, const hso_se3* T_init, float exposure_init,
                        hso_track_result* out);
/* ---- Matcher::findMatchDirect and its pieces (src/matcher.cpp, src/feature_alignment.cpp) ---- */
void hso_or_warp_matrix_affine(const hso_camera* cam_ref, const hso_camera* cam_cur, const double px_ref[2],
                               const double f_ref[3], double depth_ref, const hso_se3* T_cur_ref, int level_ref,
                               double A[4]);                                       /* matcher.cpp:46-72 */
int hso_or_best_search_level(const double A[4], int max_level);                      /* :74-85 */
int hso_or_warp_affine(const double A_cur_ref[4], const uint8_t* img_ref, int cols, int rows, const double px_ref[2],
                       int level_ref, int search_level, int halfpatch_size, float* patch); /* :120-155 */
int hso_or_align2d(const uint8_t* cur_img, int cols, int rows, const float* ref_patch_with_border, const float* ref_patch,
                   int n_iter, double cur_px_estimate[2], float* cur_patch, int* iters_out, float* chi2_out);
int hso_or_align1d(const uint8_t* cur_img, int cols, int rows, const float dir[2], const float* ref_patch_with_border,
                   const float* ref_patch, int n_iter, double cur_px_estimate[2], double* h_inv, float* cur_patch,
                   int* iters_out, float* chi2_out);                                 /* feature_alignment.cpp */
double hso_or_ncc(const float* patch1, const float* patch2);                         /* matcher.cpp:379-404 */
/* Matcher::checkNormal's dot product.  The reference reads the four gradient taps unchecked (matcher.cpp:421-428): a position
 * that is NaN (an edgelet direction of norm 0 makes KLTLimited1D return a NaN pixel with "success") or outside the image is an
 * out-of-bounds heap read there — undefined.  Defined here, and identically on the device: such a position fails the check
 * (-2 is returned, below every threshold). */
double hso_or_normal_dot(const int16_t* gx, const int16_t* gy, int cols, int rows, const double pxLevel[2], const double normal[2]);
void hso_or_find_match_direct(const hso_camera* cam, const hso_align_job* job, const uint8_t* const ref_pyr[HSO_N_PYR_LEVELS],
                              const uint8_t* const cur_pyr[HSO_N_PYR_LEVELS], const int16_t* const cur_gx[HSO_N_SOBEL_LEVELS],
                              const int16_t* const cur_gy[HSO_N_SOBEL_LEVELS], int w, int h, hso_align_out* out);
/* ---- pose_optimizer::optimizeLevenbergMarquardt3rd (src/pose_optimizer.cpp:399-771) ---- */
void hso_or_pose_optimize(const hso_camera* cam, const hso_pose_job* job, hso_pose_result* out, uint8_t* outlier_mask);
/* ---- local BA linearisation (include/hso/bundle_adjustment.h:204-404 + vendored g2o) ---- */
void hso_or_ba_linearize(const hso_se3* poses, const uint8_t* pose_fixed, int n_poses, const double* idist, int n_points,
                         const hso_ba_edge* edges, int n_edges, double huber_corner, double huber_edge,
                         double* Hpp, double* bp, double* Hpc, double* Hcc, double* bc,
                         double* edge_err, double* edge_chi2, double* chi2_sum);
void hso_or_ba_huber_deltas(const hso_se3* poses, int n_poses, const double* idist, int n_points, const hso_ba_edge* edges,
                            const double* obs_uv, int n_edges, double error_multiplier2, float* huber_corner, float* huber_edge); /* bundle_adjustment.cpp:618-680 */
/* ba::LocalBundleAdjustment's optimisation (bundle_adjustment.cpp:815-823) with g2o's LM driver restated
 * (optimization_algorithm_levenberg.cpp:61-164, sparse_optimizer.cpp:354-420), full dense system */
void hso_or_ba_optimize(hso_se3* poses, const uint8_t* pose_fixed, int n_poses, double* idist, int n_points,
                        const hso_ba_edge* edges, int n_edges, double huber_corner, double huber_edge, int n_iter,
                        double* edge_chi2_out, hso_ba_result* res);
/* g2o::SE3Quat (thirdparty/g2o/g2o/types/se3quat.h): exp of [omega, upsilon] (:223-257), product (:104-110) */
void hso_or_se3quat_exp(const double update[6], hso_se3* out);
void hso_or_se3quat_mul(const hso_se3* a, const hso_se3* b, hso_se3* out);
/* ---- depth-filter seed observation (src/depth_filter.cpp:527-675, src/matcher.cpp:802-1049,1296-1606) ---- */
void hso_or_update_seed(float x, float tau2, float* mu, float* sigma2);
double hso_or_compute_tau(const hso_se3* T_ref_cur, const double f[3], double z, double px_error_angle);
void hso_or_seed_observe(const hso_camera* cam, const hso_seed* s, const hso_se3* cur_T_f_w, double cur_exposure,
                         double px_error_angle, const uint8_t* const ref_pyr[HSO_N_PYR_LEVELS],
                         const uint8_t* const cur_pyr[HSO_N_PYR_LEVELS], const int16_t* const cur_gx[HSO_N_SOBEL_LEVELS],
                         const int16_t* const cur_gy[HSO_N_SOBEL_LEVELS], int w, int h, hso_seed_out* o);
void hso_or_seed_observe_previous(const hso_camera* cam, const hso_seed* s, const hso_se3* pre_T_f_w, double pre_exposure,
                                  double px_error_angle, const uint8_t* const ref_pyr[HSO_N_PYR_LEVELS],
                                  const uint8_t* const pre_pyr[HSO_N_PYR_LEVELS], const int16_t* const pre_gx[HSO_N_SOBEL_LEVELS],
                                  const int16_t* const pre_gy[HSO_N_SOBEL_LEVELS], int w, int h, hso_seed_out* o);
/* ---- seed activation (src/depth_filter.cpp:729-1073, src/matcher.cpp:442-518) ---- */
void hso_or_find_match_seed(const hso_camera* cam, const hso_align_job* job, const uint8_t* const ref_pyr[HSO_N_PYR_LEVELS],
                            const uint8_t* const cur_pyr[HSO_N_PYR_LEVELS], const int16_t* const cur_gx[HSO_N_SOBEL_LEVELS],
                            const int16_t* const cur_gy[HSO_N_SOBEL_LEVELS], int w, int h, hso_align_out* out);
void hso_or_seed_activate(const hso_camera* cam, const hso_seed* s, const hso_activate_target* tg, int n_tg,
                          const uint8_t* const ref_pyr[HSO_N_PYR_LEVELS], const uint8_t* const* tg_pyr,
                          const int16_t* const* tg_gx, const int16_t* const* tg_gy, int w, int h,
                          int n_mean_converge_frame, hso_activate_out* o, hso_align_out* match_out);
/* ---- FAST-9 corner detection (src/feature_detection.cpp:518-587 (fastDetectST per level, fastDetect), thirdparty/fast, vision.cpp:111-151)
 *      — pinned against oracle/_ref/libfast_ref.so and tests/golden/fast9.json ---- */
int hso_or_fast9_max_barrier(const uint8_t* img, int stride, int x, int y);
int hso_or_fast_max_barrier(const uint8_t* img, int stride, int x, int y, int arc);
int hso_or_fast_detect_arc(const uint8_t* img, int w, int h, int threshold, int arc, int16_t* xy, int32_t* scores, int cap);
int hso_or_filling_hole_level(const uint8_t* img, int w, int h, int level, int frame_w, int frame_h, int min_thresh, uint8_t* have,
                              hso_corner* out, int cap);   /* fillingHole, src/feature_detection.cpp:1125-1154 (FAST-12, pinned: tests/golden/fast12.json) */
float hso_or_shi_tomasi(const uint8_t* img, int cols, int rows, int u, int v);
int hso_or_fast9_detect(const uint8_t* img, int w, int h, int threshold, int16_t* xy, int32_t* scores, int cap);
int hso_or_fast_detect_level(const uint8_t* img, int w, int h, int threshold, int border, hso_corner* out, int cap);
float hso_or_zmncc_f8(const float* host, const float* target);   /* ZMNCC_F<4>, include/hso/vikit/patch_score.h:268-305 — pinned (tests/golden/zmncc.json) */
/* ---- Reprojector candidate generation (src/reprojector.cpp:504-529, src/point.cpp:116-136, src/matcher.cpp:270-319) ---- */
int hso_or_reproject_point(const hso_camera* cam, const hso_se3* T_cur_w, const hso_se3* T_host_w, const double host_f[3], double idist,
                           int cell_size, int grid_n_cols, double px[2], int* cell);
int hso_or_close_view_obs(const double cur_pos[3], const double pos[3], const hso_kf* kfs, const hso_obs* obs, int n_obs);
void hso_or_reproject_make_job(const hso_se3* T_cur_w, double cur_exposure_time, int cur_keyframe_id, const hso_kf* kfs,
                               const hso_map_point* pt, const hso_obs* ref, const double px_cur[2], hso_align_job* j);
/* ---- edgelet candidates (src/feature_detection.cpp:749-830; cv::Canny restated, unpinned) ---- */
void hso_or_canny_l2(const int16_t* dx, const int16_t* dy, int w, int h, double low_thresh, double high_thresh, uint8_t* edges);
void hso_or_detect_grid(int width, int height, int level, int* grid, int* gcols, int* grows, int* lw, int* lh);
int hso_or_detect_cell_index(int x, int y, int grid, int gcols, int grows);
int hso_or_edgelet_level(const int16_t* gx, const int16_t* gy, int w, int h, int level, int frame_w, int frame_h, int min_thresh,
                         uint8_t* have, hso_edgelet* out, int cap);
/* ---- two-view initialisation, image side (src/initialization.cpp:225-300, :476-563; cv::calcOpticalFlowPyrLK restated, unpinned:
 *      hso_oracle_klt.c) ---- */
void hso_or_pyr_down(const uint8_t* src, int w, int h, uint8_t* dst);
void hso_or_scharr_deriv(const uint8_t* src, int w, int h, int16_t* d);
int hso_or_klt_levels(int w, int h, int win, int max_level);
void hso_or_klt_track(const uint8_t* prev, const uint8_t* cur, int w, int h, const float* pts_prev, float* pts_cur, uint8_t* status, int n,
                      int win, int max_level, int max_count, double epsilon, int use_initial_flow, float* margin);
int hso_or_patch_check(const uint8_t* img_pre, const uint8_t* img_cur, int w, int h, const float px_pre[2], const float px_cur[2], float* ncc_out);
/* per-term dump of the last evaluation for debugging: returns number of rows written */
int hso_or_tracker_pattern(int max_level, int level, int* patch_area, int* half_patch,
                           int8_t* offsets_xy);

#ifdef __cplusplus
}
#endif
#endif
