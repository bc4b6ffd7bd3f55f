/* hso_gpu_debug.h — parity / trace read-backs and developer probes of libhso_gpu.so.
 *
 * Not part of the drop-in boundary (include/hso_gpu.h): nothing a maintainer binds behind the reference's call sites is declared
 * here.  These entry points let tests and recorded runs look at tables the product keeps on the device — the list a chain call
 * walked, a sequence map as it stands, intermediate tables of the last call — and let a driver count what the library asked of
 * the HIP runtime.  The sequence engine uses them only when a sequence is being recorded (hso_vo_trace) or timed
 * (HSO_ENGINE_TIMING).  Same conventions as hso_gpu.h: plain C, POD, caller-owned buffers, int status. */
#ifndef HSO_GPU_DEBUG_H
#define HSO_GPU_DEBUG_H
#include "hso_gpu.h"
#ifdef __cplusplus
extern "C" {
#endif

/* recorded runs / parity: the list of job `job` of the last chain call (point rows and quality keys, n_listed of them), and its
 * reference feature table in hso_ref_feat records */
int hso_gpu_seq_debug_list(hso_gpu_ctx* ctx, int job, int32_t* ids_out, uint8_t* quality_out, int cap);
int hso_gpu_seq_debug_ref_table(hso_gpu_ctx* ctx, int job, hso_ref_feat* out, int cap);

/* trace / parity hook: tables the last hso_gpu_seq_chain call (cfg.want_debug = 1) left in the work area (valid until the next
 * entry point that uses it).  The listed points of the jobs lie in slices of the call: job j's at HSO_DBG_SLICES[j] (n_jobs + 1
 * int32; a slice is as long as the job's list can get, its first n_listed entries are used).  HSO_DBG_PROJ = hso_reproj_point per
 * slice entry, HSO_DBG_MATCH = hso_align_out per slice entry, HSO_DBG_BRIEF = hso_match_brief per EXAMINED candidate (job j's at
 * HSO_DBG_EXAMINED_BEGIN[j], n_jobs + 1 int32; pad_ = the candidate's position in its list), HSO_DBG_PROJECTED = one byte per slice
 * entry (reprojectPoint's return value), HSO_DBG_POSE_FEATS = n_jobs rows of max(max_fts, 1) hso_pose_feat (host_pose = index into
 * HSO_DBG_POSE_POSES' row), HSO_DBG_POSE_POSES = n_jobs rows of 128 hso_se3, HSO_DBG_POSE_NPOSES = n_jobs int32, HSO_DBG_POSE_MASK =
 * n_jobs rows of max(max_fts, 1) bytes.  bytes must equal the table's size. */
enum { HSO_DBG_PROJ = 0, HSO_DBG_MATCH = 1, HSO_DBG_POSE_FEATS = 2, HSO_DBG_POSE_POSES = 3, HSO_DBG_POSE_NPOSES = 4, HSO_DBG_SLICES = 5,
       HSO_DBG_BRIEF = 6, HSO_DBG_EXAMINED_BEGIN = 7, HSO_DBG_PROJECTED = 8, HSO_DBG_POSE_MASK = 9, HSO_DBG_N = 10 };
int hso_gpu_debug_fetch(hso_gpu_ctx* ctx, int what, void* out, size_t bytes);
/* developer census: what the library has asked of the HIP runtime since the process started (all contexts): copies enqueued, their
 * bytes, copies that went through page-locked staging because the caller's memory was pageable, stream synchronisations, nanoseconds
 * the calling threads spent blocked in them, memsets.  out[i] for i < n; entries beyond HSO_CENSUS_N read 0.  A driver that
 * differences it around its phases sees where the host round trips of a step are (hso_amd/host: HSO_ENGINE_TIMING=1). */
enum { HSO_CENSUS_COPIES = 0, HSO_CENSUS_COPY_BYTES = 1, HSO_CENSUS_STAGED = 2, HSO_CENSUS_SYNCS = 3, HSO_CENSUS_SYNC_NS = 4, HSO_CENSUS_MEMSETS = 5, HSO_CENSUS_H2D_BYTES = 6 /* the host-to-device part of COPY_BYTES */, HSO_CENSUS_N = 7 };
void hso_gpu_debug_census(int64_t* out, int n);

/* parity hook: a sequence map exactly as the library holds it now (after every patch queued so far), table by table, so that a
 * test can rebuild the same state in another context — or in the CPU restatement of the same entry points (tests/fakegpu) — and
 * hand ONE map state to both (tests/test_seq_chain.py: hso_gpu_seq_chain per call, bit for bit).
 *   HSO_DUMP_SIZES       int64[HSO_DUMP_N_SIZES]: n_kfs, n_points, n_obs, fts_cap, n_cands, frame-feature table lengths [2], their frame
 *                        ids [2], which of the two is the newer, then sizeof(hso_kf), sizeof(hso_map_point), sizeof(hso_obs),
 *                        sizeof(hso_seq_feature), sizeof(hso_seq_job), sizeof(hso_seq_result)
 *   HSO_DUMP_KFS         hso_kf[n_kfs]                 HSO_DUMP_KEY_POINTS   int32[5 * n_kfs]  (Frame::key_pts_ as point rows)
 *   HSO_DUMP_POINTS      hso_map_point[n_points]       HSO_DUMP_OBS          hso_obs[n_obs]
 *   HSO_DUMP_OBS_POINT   int32[n_obs] (Feature::point) HSO_DUMP_KF_NFTS      int32[n_kfs] (length of every Frame::fts_)
 *   HSO_DUMP_KF_FTS      int32[n_kfs * fts_cap] (row r = keyframe r's list, its first KF_NFTS[r] entries are used)
 *   HSO_DUMP_CANDS       int32[n_cands]                HSO_DUMP_FRAME_FEATS0 / 1   hso_seq_feature[table length]
 * bytes must equal the table's size (read HSO_DUMP_SIZES first).  Synchronises the context's stream. */
enum { HSO_DUMP_SIZES = 0, HSO_DUMP_KFS = 1, HSO_DUMP_POINTS = 2, HSO_DUMP_OBS = 3, HSO_DUMP_OBS_POINT = 4, HSO_DUMP_KEY_POINTS = 5,
       HSO_DUMP_KF_NFTS = 6, HSO_DUMP_KF_FTS = 7, HSO_DUMP_CANDS = 8, HSO_DUMP_FRAME_FEATS0 = 9, HSO_DUMP_FRAME_FEATS1 = 10, HSO_DUMP_N = 11 };
#define HSO_DUMP_N_SIZES 16
int hso_gpu_seqmap_debug_dump(hso_gpu_ctx* ctx, int map, int what, void* out, size_t bytes);

/* The windows of the last hso_gpu_seq_local_ba call of a context as the device assembled them (traces record them; the parity tests
 * compare them with the restatement's and replay them through the value-passing calls).  `job` indexes that call's jobs:
 *   HSO_BAW_SIZES        int32[4]: n_poses, n_points, n_edges, status
 *   HSO_BAW_VERTEX_ROWS  int32[n_poses]  (keyframe row of every vertex)         HSO_BAW_FIXED     uint8[n_poses]
 *   HSO_BAW_EDGES        hso_ba_edge[n_edges]                                   HSO_BAW_OBS_UV    double[2 * n_edges]
 *   HSO_BAW_EDGE_OBS     int32[n_edges]  (observation row of every edge)        HSO_BAW_EDGE_CHI2 double[n_edges] (after the optimisation)
 *   HSO_BAW_POSES_OUT    hso_se3[n_poses]                                       HSO_BAW_POSES_IN / HSO_BAW_IDIST_IN: the state the
 *                                                                               optimisation started from, hso_se3[n_poses] / double[n_points]
 * bytes must equal the table's size.  Valid until the context's next batched call.  Synchronises the context's stream. */
enum { HSO_BAW_SIZES = 0, HSO_BAW_VERTEX_ROWS = 1, HSO_BAW_FIXED = 2, HSO_BAW_EDGES = 3, HSO_BAW_OBS_UV = 4, HSO_BAW_EDGE_OBS = 5,
       HSO_BAW_EDGE_CHI2 = 6, HSO_BAW_POSES_OUT = 7, HSO_BAW_POSES_IN = 8, HSO_BAW_IDIST_IN = 9, HSO_BAW_N = 10 };
int hso_gpu_seq_ba_debug_window(hso_gpu_ctx* ctx, int job, int what, void* out, size_t bytes);

/* test hook: Gaussian pyramid level `level` of a resident frame (cv::pyrDown chain) and its Scharr derivative image
 * (interleaved Ix, Iy); either output may be NULL */
int hso_gpu_klt_debug_level(hso_gpu_ctx* ctx, int64_t frame, int level, uint8_t* img_out, int16_t* deriv_out);

#ifdef __cplusplus
}
#endif
#endif /* HSO_GPU_DEBUG_H */
