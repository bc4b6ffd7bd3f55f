// hso_engine_c.cpp — the C interface of libhso_host.so (include/hso_vo.h) over the sequence engine: a single sequence is a bank
// of one.  Errors: where the reference throws before touching anything (wrong image size) the call returns HSO_E_INVALID and the
// handle stays usable (engine::Refused); a failed device call — or any other exception — in the middle of a step can leave the
// tables half updated, so the handle is poisoned and every later call reports the first error.
#include "hso_engine_impl.h"
#include "hso_init.h"

using hso::engine::Bank;
using hso::engine::Settings;

struct hso_vo { Bank* bank = nullptr; };
struct hso_vo_multi { Bank* bank = nullptr; };

namespace {

Bank* make_bank(const hso_camera* cam, int max_fts, int n, int device, int* rc)
{
  hso_gpu_ctx* ctx = nullptr;
  // the engine was compiled against include/hso_gpu.h's struct layouts: a device library of another ABI generation is refused
  if (hso_gpu_abi_version() != HSO_GPU_ABI_VERSION) { *rc = HSO_E_UNSUPPORTED; return nullptr; }
  *rc = hso_gpu_create(&ctx, device, nullptr);
  if (*rc < 0) return nullptr;
  Settings cfg;
  cfg.max_fts = max_fts;
  // the bank owns the context from here: its constructor cleans up after itself when it throws (and destroys the context)
  try { return new Bank(ctx, true, *cam, cfg, n); }
  catch (const hso::engine::Refused&) { *rc = HSO_E_INVALID; return nullptr; }
  catch (const std::exception&) { *rc = HSO_E_HIP; return nullptr; }
}

template <typename F> int guarded(Bank* b, F f)
{
  if (!b) return HSO_E_INVALID;
  if (b->poisoned) {
    if (b->err.find("handle unusable") == std::string::npos) b->err = "handle unusable after a device error (" + b->err + "): destroy it and create a new one";
    return HSO_E_HIP;
  }
  try { f(); return HSO_OK; }
  catch (const hso::engine::Refused& e) { b->err = e.what(); return HSO_E_INVALID; }             // nothing was touched (or everything put back)
  catch (const hso::engine::DeviceFault& e) { b->err = e.what(); b->poisoned = true; return HSO_E_HIP; }
  catch (const std::bad_alloc&) { b->err = "out of (page-locked) host memory in the middle of a step"; b->poisoned = true; return HSO_E_NOMEM; }
  // anything else was thrown after a step had begun to change the tables: they may be half updated
  catch (const std::exception& e) { b->err = e.what(); b->poisoned = true; return HSO_E_INVALID; }
}

}  // namespace

extern "C" {

int hso_vo_create(hso_vo** out, const hso_camera* cam, int max_fts, int device)
{
  if (!out || !cam || max_fts <= 0 || cam->width <= 0 || cam->height <= 0) return HSO_E_INVALID;
  *out = nullptr;
  int rc = HSO_OK;
  Bank* b = make_bank(cam, max_fts, 1, device, &rc);
  if (!b) return rc;
  *out = new hso_vo{b};
  return HSO_OK;
}
void hso_vo_destroy(hso_vo* v) { if (v) { delete v->bank; delete v; } }
const char* hso_vo_last_error(const hso_vo* v) { return v ? v->bank->err.c_str() : "null handle"; }
int hso_vo_trace(hso_vo* v, const char* path) { return (v && v->bank->trace(0, path)) ? HSO_OK : HSO_E_INVALID; }
static int set_opts(Bank* b, const hso_vo_options* o)
{
  if (!b || !o || o->size < 12 || o->size > (int32_t)sizeof(hso_vo_options)) return HSO_E_INVALID;
  const bool no_pin = o->size >= 16 && o->no_numa_pin != 0;
  return guarded(b, [&]() { b->set_options(o->sync_previous != 0, o->track_no_coop != 0, no_pin); });
}
int hso_vo_set_options(hso_vo* v, const hso_vo_options* o) { return v ? set_opts(v->bank, o) : HSO_E_INVALID; }
int hso_vo_trace_state(hso_vo* v, int on) { return (v && v->bank->trace_state(0, on != 0)) ? HSO_OK : HSO_E_INVALID; }

int hso_vo_set_first_frame(hso_vo* v, const uint8_t* img, int width, int height, double timestamp, const float* depth_z, const hso_se3* T_f_w)
{
  if (!v || !img || !depth_z) return HSO_E_INVALID;
  return guarded(v->bank, [&]() { v->bank->set_first_frames(&img, width, height, &timestamp, &depth_z, T_f_w); });
}
int hso_vo_start(hso_vo* v) { return v ? guarded(v->bank, [&]() { v->bank->start(nullptr); }) : HSO_E_INVALID; }
int hso_vo_add_image(hso_vo* v, const uint8_t* img, int width, int height, double timestamp)
{
  if (!v || !img) return HSO_E_INVALID;
  return guarded(v->bank, [&]() { v->bank->add_images(&img, width, height, &timestamp); });
}
int hso_vo_get_status(hso_vo* v, hso_vo_status* st)
{
  if (!v || !st) return HSO_E_INVALID;
  return guarded(v->bank, [&]() { v->bank->status(0, st); });
}
int hso_vo_get_keyframes(hso_vo* v, double* timestamps, hso_se3* T_f_w, int32_t* frame_ids, int cap)
{
  return v ? v->bank->keyframes(0, timestamps, T_f_w, frame_ids, cap) : HSO_E_INVALID;
}

int hso_vo_multi_create(hso_vo_multi** out, const hso_camera* cam, int max_fts, int n_sequences, int device)
{
  if (!out || !cam || max_fts <= 0 || cam->width <= 0 || cam->height <= 0 || n_sequences < 1 || n_sequences > 4096) return HSO_E_INVALID;
  *out = nullptr;
  int rc = HSO_OK;
  Bank* b = make_bank(cam, max_fts, n_sequences, device, &rc);
  if (!b) return rc;
  *out = new hso_vo_multi{b};
  return HSO_OK;
}
void hso_vo_multi_destroy(hso_vo_multi* m) { if (m) { delete m->bank; delete m; } }
const char* hso_vo_multi_last_error(const hso_vo_multi* m) { return m ? m->bank->err.c_str() : "null handle"; }
int hso_vo_multi_size(const hso_vo_multi* m) { return m ? m->bank->size() : HSO_E_INVALID; }
int hso_vo_multi_trace(hso_vo_multi* m, int sequence, const char* path) { return (m && m->bank->trace(sequence, path)) ? HSO_OK : HSO_E_INVALID; }
int hso_vo_multi_set_options(hso_vo_multi* m, const hso_vo_options* o) { return m ? set_opts(m->bank, o) : HSO_E_INVALID; }
int hso_vo_multi_trace_state(hso_vo_multi* m, int sequence, int on) { return (m && m->bank->trace_state(sequence, on != 0)) ? HSO_OK : HSO_E_INVALID; }
int hso_vo_multi_set_first_frames(hso_vo_multi* m, const uint8_t* const* imgs, int width, int height, const double* timestamps,
                                  const float* const* depth_z, const hso_se3* T_f_w)
{
  if (!m || !imgs || !depth_z) return HSO_E_INVALID;
  return guarded(m->bank, [&]() { m->bank->set_first_frames(imgs, width, height, timestamps, depth_z, T_f_w); });
}
int hso_vo_multi_start(hso_vo_multi* m, const uint8_t* which) { return m ? guarded(m->bank, [&]() { m->bank->start(which); }) : HSO_E_INVALID; }
int hso_vo_multi_add_images(hso_vo_multi* m, const uint8_t* const* imgs, int width, int height, const double* timestamps)
{
  if (!m || !imgs) return HSO_E_INVALID;
  return guarded(m->bank, [&]() { m->bank->add_images(imgs, width, height, timestamps); });
}
int hso_vo_multi_add_images_device(hso_vo_multi* m, const uint8_t* const* imgs, int width, int height, const double* timestamps)
{
  if (!m || !imgs) return HSO_E_INVALID;
  return guarded(m->bank, [&]() { m->bank->add_images(imgs, width, height, timestamps, true); });
}
int hso_vo_multi_get_status(hso_vo_multi* m, int sequence, hso_vo_status* st)
{
  if (!m || !st || sequence < 0 || sequence >= m->bank->size()) return HSO_E_INVALID;
  return guarded(m->bank, [&]() { m->bank->status(sequence, st); });
}
int hso_vo_multi_get_keyframes(hso_vo_multi* m, int sequence, double* timestamps, hso_se3* T_f_w, int32_t* frame_ids, int cap)
{
  if (!m || sequence < 0 || sequence >= m->bank->size()) return HSO_E_INVALID;
  return m->bank->keyframes(sequence, timestamps, T_f_w, frame_ids, cap);
}
int hso_vo_multi_get_trajectory(hso_vo_multi* m, int sequence, double* timestamps, hso_se3* T_f_w, int cap)
{
  if (!m || sequence < 0 || sequence >= m->bank->size()) return HSO_E_INVALID;
  return m->bank->trajectory(sequence, timestamps, T_f_w, cap);
}
int hso_vo_get_trajectory(hso_vo* v, double* timestamps, hso_se3* T_f_w, int cap) { return v ? v->bank->trajectory(0, timestamps, T_f_w, cap) : HSO_E_INVALID; }
int hso_vo_multi_call_counts(hso_vo_multi* m, int64_t* calls, int64_t* items, int cap)
{
  if (!m) return HSO_E_INVALID;
  m->bank->call_counts(calls, items, cap);
  return 10;
}

int hso_vo_multi_alg_bytes(const hso_vo_multi* m, double* out, int cap) { if (!m || !out) return HSO_E_INVALID; m->bank->alg_bytes(out, cap); return 5; }
int hso_vo_host_share(int banks_in_process) { if (banks_in_process < 1) return HSO_E_INVALID; hso::engine::set_host_share(banks_in_process); return HSO_OK; }
int hso_vo_multi_threads(const hso_vo_multi* m) { return m ? m->bank->threads() : HSO_E_INVALID; }
int hso_vo_host_cpu_quota(void) { return hso::engine::host_cpu_budget(); }

int hso_vo_init_compute_matrix(const double* f_ref, const double* f_cur, int n, double focal_length, double reproj_thresh, hso_se3* T_cur_from_ref,
                               int32_t* inliers, int cap, double* xyz_in_cur, int32_t* used_homography)
{
  if (!f_ref || !f_cur || n < 0 || !T_cur_from_ref) return HSO_E_INVALID;
  try {
    std::vector<hso::Vector3d> a(n), b(n), xyz;
    for (int i = 0; i < n; i++) { a[i] = {f_ref[3 * i], f_ref[3 * i + 1], f_ref[3 * i + 2]}; b[i] = {f_cur[3 * i], f_cur[3 * i + 1], f_cur[3 * i + 2]}; }
    std::vector<int> in;
    hso::SE3 T;
    int used = 0;
    hso::initialization::computeInitializeMatrix(a, b, focal_length, reproj_thresh, in, xyz, T, &used);
    *T_cur_from_ref = T.v;
    if (used_homography) *used_homography = used;
    for (size_t i = 0; i < in.size() && (int)i < cap && inliers; i++) inliers[i] = in[i];
    for (size_t i = 0; i < xyz.size() && xyz_in_cur; i++) { xyz_in_cur[3 * i] = xyz[i][0]; xyz_in_cur[3 * i + 1] = xyz[i][1]; xyz_in_cur[3 * i + 2] = xyz[i][2]; }
    return (int)in.size();
  } catch (const std::exception&) { return HSO_E_INVALID; }
}

}  // extern "C"
