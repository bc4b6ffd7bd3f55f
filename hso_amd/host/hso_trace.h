// hso_trace.h — C-ABI call recorder of the host driver.
//
// Every device call the driver makes goes through one of the api:: wrappers below; with a trace
// file open they append the call's inputs and outputs as raw tables (the C-ABI's own POD layouts).
// tests/test_chain_gpu.py replays each record through the CPU restatement (oracle/) and compares —
// stage-by-stage parity from the same evolving state, without the product ever touching the oracle.
// The reference's counterpart is the -DTRACE performance log (include/hso/global.h:108-123): it
// records timings only; this records the data.
//
// Record: u32 magic 'HSTR', u32 name_len, name, u32 n_fields; field: u32 key_len, key, u64 n_bytes, bytes.
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>
#include "../../include/hso_gpu.h"

namespace hso {
namespace api {

struct Trace {
  FILE* f = nullptr;
  uint32_t n_fields_pos = 0, n_fields = 0;
  bool open(const char* path) { close(); f = std::fopen(path, "wb"); return f != nullptr; }
  void close() { if (f) std::fclose(f); f = nullptr; }
  bool on() const { return f != nullptr; }
  void begin(const char* name, uint32_t nf)
  {
    const uint32_t magic = 0x52545348u, nl = (uint32_t)std::strlen(name);
    std::fwrite(&magic, 4, 1, f); std::fwrite(&nl, 4, 1, f); std::fwrite(name, 1, nl, f); std::fwrite(&nf, 4, 1, f);
  }
  void field(const char* key, const void* data, size_t bytes)
  {
    const uint32_t kl = (uint32_t)std::strlen(key); const uint64_t nb = bytes;
    std::fwrite(&kl, 4, 1, f); std::fwrite(key, 1, kl, f); std::fwrite(&nb, 8, 1, f);
    if (bytes) std::fwrite(data, 1, bytes, f);
  }
  void scalar(const char* key, double v) { field(key, &v, 8); }
  void flush() { if (f) std::fflush(f); }
};

Trace& trace();   // one per driver thread (hso_vo.cpp): a sequence of the multi-sequence driver can record its own calls

// Where a driver thread's device calls go.  Null (the default): straight to the C-ABI.  The multi-sequence driver (hso_multi.cpp)
// gives each of its sequence threads a router: the calls of all sequences meet there and leave as ONE batched C-ABI call per kind
// (hso_gpu_coarse_track_batch with N jobs, hso_gpu_reproject_match_multi, hso_gpu_pose_optimize_batch, ...).  Every method
// returns the C-ABI status of the call as this sequence would have seen it alone and fills the same outputs.
struct Router {
  virtual ~Router() {}
  virtual int frame_upload(int64_t id, const uint8_t* img, int w, int h, hso_frame_stats* st) = 0;
  virtual int frame_release(int64_t id) = 0;
  virtual int coarse_track(const hso_camera* cam, const hso_track_params* p, const hso_track_job* job, hso_track_result* res) = 0;
  virtual int reproject_match(const hso_camera* cam, int64_t cur_id, const hso_se3* T_cur_w, double cur_exposure, int cur_kf_id, const hso_kf* kfs,
                              int n_kfs, const hso_map_point* pts, int n_pts, const hso_obs* obs, int n_obs, int cell_size, int grid_n_cols,
                              hso_reproj_point* proj, hso_align_out* match) = 0;
  virtual int align_batch(const hso_camera* cam, int64_t cur_id, const hso_align_job* jobs, int n, hso_align_out* out) = 0;
  virtual int pose_optimize(const hso_camera* cam, const hso_pose_job* job, hso_pose_result* res, uint8_t* mask) = 0;
  virtual int seed_observe(const hso_camera* cam, int64_t cur_id, const hso_se3* T, double exposure, double px_error_angle, const hso_seed* seeds,
                           int n, hso_seed_out* out) = 0;
  virtual int seed_activate(const hso_camera* cam, const hso_seed* seeds, int n, const int32_t* begin, const hso_activate_target* targets,
                            int n_mean, hso_activate_out* out) = 0;
  virtual int ba_optimize(hso_se3* poses, const uint8_t* fixed, int n_poses, double* idist, int n_points, const hso_ba_edge* edges, int n_edges,
                          double hc, double he, int n_iter, double* chi2, hso_ba_result* res) = 0;
  // anything without a multi-sequence form: runs alone, serialised with the other sequences' calls (one context, one stream)
  virtual int solo(int (*fn)(void*), void* arg) = 0;
  virtual const char* last_error() = 0;
};
Router*& router();   // thread_local (hso_vo.cpp)

// A failed device call in the middle of processFrame: unlike the reference's own exceptions (wrong image size, thrown before
// anything is touched) it can leave the pointer graph half updated (a keyframe already registered, features already
// referenced by points), so the driver treats it as fatal for the handle (hso_vo.cpp: vo_guard).
struct DeviceError : std::runtime_error { using std::runtime_error::runtime_error; };

inline void check(hso_gpu_ctx* ctx, int rc, const char* what)
{
  if (rc < 0) throw DeviceError(std::string(what) + ": " + (router() ? router()->last_error() : hso_gpu_last_error(ctx)));
}

// a call without a multi-sequence form, through the router when there is one
template <typename F> inline int routed(F f)
{
  if (!router()) return f();
  return router()->solo([](void* a) { return (*static_cast<F*>(a))(); }, &f);
}

}  // namespace api
}  // namespace hso
