// hso_api.h — the device calls of the host driver, each one through a wrapper that (a) turns a negative status into
// std::runtime_error with the context's message, like the reference's exceptions, and (b) appends the call's inputs
// and outputs to the trace when one is open (hso_trace.h).
#pragma once
#include "hso_trace.h"

namespace hso {
namespace api {

inline void frame_upload(hso_gpu_ctx* ctx, int64_t id, const uint8_t* img, int w, int h, hso_frame_stats* st)
{
  check(ctx, hso_gpu_frame_upload(ctx, id, img, w, h, 0, st), "Frame");
  Trace& t = trace();
  if (t.on()) {
    t.begin("frame_upload", 5);
    t.scalar("frame_id", (double)id); t.scalar("width", w); t.scalar("height", h);
    t.field("img", img, (size_t)w * h); t.field("stats", st, sizeof(*st));
  }
}

inline void coarse_track(hso_gpu_ctx* ctx, const hso_camera* cam, const hso_track_params* p, const hso_track_job* job, hso_track_result* res)
{
  check(ctx, hso_gpu_coarse_track_batch(ctx, cam, p, job, 1, res), "CoarseTracker");
  Trace& t = trace();
  if (t.on()) {
    t.begin("coarse_track", 8);
    t.field("cam", cam, sizeof(*cam)); t.field("params", p, sizeof(*p));
    t.scalar("ref_frame_id", (double)job->ref_frame_id); t.scalar("cur_frame_id", (double)job->cur_frame_id);
    t.field("feats", job->feats, sizeof(hso_ref_feat) * (size_t)job->n_feats);
    t.field("T_cur_ref", &job->T_cur_ref, sizeof(hso_se3)); t.scalar("exposure_rat", job->exposure_rat);
    t.field("result", res, sizeof(*res));
  }
}

inline void reproject_match(hso_gpu_ctx* ctx, const hso_camera* cam, int64_t cur_id, const hso_se3* T_cur_w, double cur_exposure,
                            int cur_kf_id, const hso_kf* kfs, int n_kfs, const hso_map_point* pts, int n_pts, const hso_obs* obs,
                            int n_obs, int cell_size, int grid_n_cols, hso_reproj_point* proj, hso_align_out* match)
{
  check(ctx, hso_gpu_reproject_match(ctx, cam, cur_id, T_cur_w, cur_exposure, cur_kf_id, kfs, n_kfs, pts, n_pts, obs, n_obs, cell_size,
                                                grid_n_cols, proj, match), "Reprojector");
  Trace& t = trace();
  if (t.on()) {
    t.begin("reproject_match", 12);
    t.field("cam", cam, sizeof(*cam)); t.scalar("cur_frame_id", (double)cur_id); t.field("T_cur_w", T_cur_w, sizeof(hso_se3));
    t.scalar("cur_exposure", cur_exposure); t.scalar("cur_keyframe_id", cur_kf_id);
    t.field("kfs", kfs, sizeof(hso_kf) * (size_t)n_kfs); t.field("points", pts, sizeof(hso_map_point) * (size_t)n_pts);
    t.field("obs", obs, sizeof(hso_obs) * (size_t)n_obs); t.scalar("cell_size", cell_size); t.scalar("grid_n_cols", grid_n_cols);
    t.field("proj", proj, sizeof(hso_reproj_point) * (size_t)n_pts); t.field("match", match, sizeof(hso_align_out) * (size_t)n_pts);
  }
}

inline void pose_optimize(hso_gpu_ctx* ctx, const hso_camera* cam, const hso_pose_job* job, hso_pose_result* res, uint8_t* mask)
{
  uint8_t* mp = mask;
  check(ctx, hso_gpu_pose_optimize_batch(ctx, cam, job, 1, res, &mp), "pose_optimizer");
  Trace& t = trace();
  if (t.on()) {
    t.begin("pose_optimize", 8);
    t.field("cam", cam, sizeof(*cam)); t.field("feats", job->feats, sizeof(hso_pose_feat) * (size_t)job->n_feats);
    t.field("poses", job->poses_f_w, sizeof(hso_se3) * (size_t)job->n_poses); t.field("T_f_w", &job->T_f_w, sizeof(hso_se3));
    t.scalar("reproj_thresh", job->reproj_thresh); t.scalar("n_iter", job->n_iter);
    t.field("result", res, sizeof(*res)); t.field("mask", mask, (size_t)job->n_feats);
  }
}

inline void klt_track(hso_gpu_ctx* ctx, int64_t prev_id, int64_t cur_id, const float* px_prev, const float* px_init, int n,
                      const hso_klt_params* params, hso_klt_result* out)
{
  check(ctx, hso_gpu_klt_track(ctx, prev_id, cur_id, px_prev, px_init, n, params, out), "trackKlt");
  Trace& t = trace();
  if (t.on()) {
    t.begin("klt_track", 6);
    t.scalar("prev_frame_id", (double)prev_id); t.scalar("cur_frame_id", (double)cur_id);
    t.field("px_prev", px_prev, sizeof(float) * 2 * (size_t)n); t.field("px_init", px_init, sizeof(float) * 2 * (size_t)n);
    t.field("params", params, sizeof(*params)); t.field("result", out, sizeof(hso_klt_result) * (size_t)n);
  }
}

inline void detect_candidates(hso_gpu_ctx* ctx, bool init, int64_t id, int n_levels, int min_thresh, hso_corner* co, int corner_cap,
                              int32_t* nc, hso_edgelet* ed, hso_corner* fill, int second_cap, int32_t* n_second)
{
  const int rc = init ? hso_gpu_detect_candidates_init(ctx, &id, 1, n_levels, min_thresh, co, corner_cap, nc, fill, second_cap, n_second)
                      : hso_gpu_detect_candidates(ctx, &id, 1, n_levels, min_thresh, co, corner_cap, nc, ed, second_cap, n_second);
  check(ctx, rc, "FeatureExtractor");
}

// the candidate lists as the extractor consumed them (after the capacity retry), one record per detect()
inline void trace_candidates(bool init, int64_t id, int n_levels, int min_thresh, const hso_corner* co, int corner_cap, const int32_t* nc,
                             const hso_edgelet* ed, const hso_corner* fill, int second_cap, const int32_t* n_second)
{
  Trace& t = trace();
  if (!t.on()) return;
  t.begin("detect_candidates", 5 + 2 * (uint32_t)n_levels + 1);
  t.scalar("init", init ? 1 : 0); t.scalar("frame_id", (double)id); t.scalar("n_levels", n_levels); t.scalar("min_thresh", min_thresh);
  t.field("corner_counts", nc, sizeof(int32_t) * (size_t)n_levels);
  for (int L = 0; L < n_levels; L++) {
    const std::string k = "corners" + std::to_string(L);
    t.field(k.c_str(), co + (size_t)L * corner_cap, sizeof(hso_corner) * (size_t)nc[L]);
  }
  if (init) {
    t.field("fill", fill, sizeof(hso_corner) * (size_t)n_second[0]);
    for (int L = 1; L < n_levels; L++) t.field("unused", nullptr, 0);
    t.field("second_counts", n_second, sizeof(int32_t));
  } else {
    for (int L = 0; L < n_levels; L++) {
      const std::string k = "edgelets" + std::to_string(L);
      t.field(k.c_str(), ed + (size_t)L * second_cap, sizeof(hso_edgelet) * (size_t)n_second[L]);
    }
    t.field("second_counts", n_second, sizeof(int32_t) * (size_t)n_levels);
  }
}

inline int select_octree(const hso_keypoint* keys, int n, int w, int h, int n_features, hso_keypoint* out, int cap)
{
  const int m = hso_gpu_select_octree(keys, n, 0, w, 0, h, n_features, out, cap);
  if (m < 0) throw std::runtime_error("FeatureExtractor: oct-tree selection failed");
  Trace& t = trace();
  if (t.on()) {
    t.begin("select_octree", 5);
    t.field("keys", keys, sizeof(hso_keypoint) * (size_t)n); t.scalar("width", w); t.scalar("height", h); t.scalar("n_features", n_features);
    t.field("out", out, sizeof(hso_keypoint) * (size_t)m);
  }
  return m;
}

inline void seed_observe(hso_gpu_ctx* ctx, const hso_camera* cam, int64_t cur_id, const hso_se3* T, double exposure, double px_error_angle,
                         const hso_seed* seeds, int n, hso_seed_out* out)
{
  check(ctx, hso_gpu_seed_observe(ctx, cam, cur_id, T, exposure, px_error_angle, seeds, n, out), "DepthFilter");
  Trace& t = trace();
  if (t.on()) {
    t.begin("seed_observe", 7);
    t.field("cam", cam, sizeof(*cam)); t.scalar("cur_frame_id", (double)cur_id); t.field("T_f_w", T, sizeof(hso_se3));
    t.scalar("exposure", exposure); t.scalar("px_error_angle", px_error_angle);
    t.field("seeds", seeds, sizeof(hso_seed) * (size_t)n); t.field("out", out, sizeof(hso_seed_out) * (size_t)n);
  }
}

inline void seed_activate(hso_gpu_ctx* ctx, const hso_camera* cam, const hso_seed* seeds, int n, const int32_t* begin,
                          const hso_activate_target* targets, int n_mean, hso_activate_out* out)
{
  check(ctx, hso_gpu_seed_activate(ctx, cam, seeds, n, begin, targets, n_mean, out, nullptr), "DepthFilter::activatePoint");
  Trace& t = trace();
  if (t.on()) {
    t.begin("seed_activate", 6);
    t.field("cam", cam, sizeof(*cam)); t.field("seeds", seeds, sizeof(hso_seed) * (size_t)n);
    t.field("target_begin", begin, sizeof(int32_t) * (size_t)(n + 1));
    t.field("targets", targets, sizeof(hso_activate_target) * (size_t)begin[n]); t.scalar("n_mean_converge_frame", n_mean);
    t.field("out", out, sizeof(hso_activate_out) * (size_t)n);
  }
}

inline void seed_reproject_match(hso_gpu_ctx* ctx, const hso_camera* cam, int64_t cur_id, const hso_se3* T, double exposure,
                                 const hso_seed* seeds, int n, int cell_size, int grid_n_cols, hso_reproj_point* proj, hso_align_out* match)
{
  check(ctx, hso_gpu_seed_reproject_match(ctx, cam, cur_id, T, exposure, seeds, n, cell_size, grid_n_cols, proj, match),
        "Reprojector (seeds)");
  Trace& t = trace();
  if (t.on()) {
    t.begin("seed_reproject_match", 9);
    t.field("cam", cam, sizeof(*cam)); t.scalar("cur_frame_id", (double)cur_id); t.field("T_f_w", T, sizeof(hso_se3));
    t.scalar("exposure", exposure); t.field("seeds", seeds, sizeof(hso_seed) * (size_t)n);
    t.scalar("cell_size", cell_size); t.scalar("grid_n_cols", grid_n_cols);
    t.field("proj", proj, sizeof(hso_reproj_point) * (size_t)n); t.field("match", match, sizeof(hso_align_out) * (size_t)n);
  }
}

inline void ba_huber_deltas(hso_gpu_ctx* ctx, const hso_se3* poses, int n_poses, const double* idist, int n_points, const hso_ba_edge* edges,
                            const double* obs_uv, int n_edges, double err_mult2, float* hc, float* he)
{
  check(ctx, hso_gpu_ba_huber_deltas(ctx, poses, n_poses, idist, n_points, edges, obs_uv, n_edges, err_mult2, hc, he),
        "LocalBundleAdjustment");
  Trace& t = trace();
  if (t.on()) {
    t.begin("ba_huber_deltas", 7);
    t.field("poses", poses, sizeof(hso_se3) * (size_t)n_poses); t.field("idist", idist, sizeof(double) * (size_t)n_points);
    t.field("edges", edges, sizeof(hso_ba_edge) * (size_t)n_edges); t.field("obs_uv", obs_uv, sizeof(double) * 2 * (size_t)n_edges);
    t.scalar("error_multiplier2", err_mult2); t.scalar("huber_corner", *hc); t.scalar("huber_edge", *he);
  }
}

inline void ba_optimize(hso_gpu_ctx* ctx, hso_se3* poses, const uint8_t* fixed, int n_poses, double* idist, int n_points,
                        const hso_ba_edge* edges, int n_edges, double hc, double he, int n_iter, double* chi2, hso_ba_result* res)
{
  Trace& t = trace();
  std::vector<hso_se3> p0; std::vector<double> i0;
  if (t.on()) { p0.assign(poses, poses + n_poses); i0.assign(idist, idist + n_points); }
  check(ctx, hso_gpu_ba_optimize(ctx, poses, fixed, n_poses, idist, n_points, edges, n_edges, hc, he, n_iter, chi2, res), "LocalBundleAdjustment");
  if (t.on()) {
    t.begin("ba_optimize", 11);
    t.field("poses_in", p0.data(), sizeof(hso_se3) * (size_t)n_poses); t.field("fixed", fixed, (size_t)n_poses);
    t.field("idist_in", i0.data(), sizeof(double) * (size_t)n_points); t.field("edges", edges, sizeof(hso_ba_edge) * (size_t)n_edges);
    t.scalar("huber_corner", hc); t.scalar("huber_edge", he); t.scalar("n_iter", n_iter);
    t.field("poses_out", poses, sizeof(hso_se3) * (size_t)n_poses); t.field("idist_out", idist, sizeof(double) * (size_t)n_points);
    t.field("edge_chi2", chi2, sizeof(double) * (size_t)n_edges); t.field("result", res, sizeof(*res));
  }
}

}  // namespace api
}  // namespace hso
