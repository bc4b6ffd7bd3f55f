// hso_fake_gpu.cpp — TEST INFRASTRUCTURE: the entry points of include/hso_gpu.h that the sequence engine calls, implemented on
// the CPU restatement (oracle/).  It lets the engine's host logic (hso_amd/host/hso_engine*.cpp) run, under sanitizers, in a
// container without a GPU: tests/test_engine_cpu.py builds the engine against this file instead of libhso_gpu.so.  Nothing in
// the product links it, and it is also the "reference CPU path" of a whole evolving sequence for end-to-end comparisons.
// The grid selection (Reprojector::reprojectMap's passes, src/reprojector.cpp:253-306, 352-429, 556-612) is written here as the
// sequential walk over per-cell lists — a third, independent statement of it beside k_select and tests/test_select.py.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <list>
#include <map>
#include <string>
#include <vector>
#include "../../oracle/hso_oracle.h"
#include "../../include/hso_gpu_debug.h"

struct FakeFrame {
  int w = 0, h = 0;
  std::vector<uint8_t> lev[HSO_N_PYR_LEVELS];
  std::vector<int16_t> gx[HSO_N_SOBEL_LEVELS], gy[HSO_N_SOBEL_LEVELS];
  int lw[HSO_N_PYR_LEVELS], lh[HSO_N_PYR_LEVELS];
  const uint8_t* pyr[HSO_N_PYR_LEVELS];
  const int16_t* sx[HSO_N_SOBEL_LEVELS];
  const int16_t* sy[HSO_N_SOBEL_LEVELS];
};
struct FakeMap {
  std::vector<hso_kf> kfs; std::vector<hso_map_point> pts; std::vector<hso_obs> obs;
  std::vector<int32_t> obs_pt;                       // Feature::point per observation row
  std::vector<int32_t> keys;                         // Frame::key_pts_ as point rows, 5 per keyframe
  std::vector<std::vector<int32_t>> kf_fts;          // Frame::fts_ per keyframe row
  std::vector<int32_t> cands;                        // MapPointCandidates::candidates_
  int fts_cap = 0;
  std::vector<hso_seq_feature> ff[2]; int64_t ff_frame[2] = {-1, -1}; int ff_newest = 0;
};
struct FakeSeedTable { std::vector<hso_seed> s; std::vector<int> group; std::vector<uint8_t> alive; };

struct hso_gpu_ctx {
  std::string err;
  std::map<int64_t, FakeFrame*> frames;
  std::vector<FakeMap*> maps;
  std::vector<FakeSeedTable*> tables;
  // hso_gpu_debug_fetch / hso_gpu_seq_debug_* / hso_gpu_seq_events: what the last chain call left
  std::vector<hso_reproj_point> dbg_proj; std::vector<hso_align_out> dbg_match; std::vector<hso_pose_feat> dbg_feats;
  std::vector<hso_se3> dbg_poses; std::vector<int32_t> dbg_nposes, dbg_slices, dbg_exbegin; std::vector<hso_match_brief> dbg_brief;
  std::vector<uint8_t> dbg_projected, dbg_mask;
  std::vector<std::vector<int32_t>> last_ids, last_events; std::vector<std::vector<uint8_t>> last_quality; std::vector<std::vector<hso_ref_feat>> last_table;
  std::vector<hso_seed_brief> prev_briefs; int prev_table = -1;   // hso_gpu_seed_table_observe_previous_begin / _end
  // hso_gpu_seq_ba_debug_window: the windows of the last hso_gpu_seq_local_ba call
  struct BaWindow {
    std::vector<int32_t> rows, edge_obs; std::vector<uint8_t> fixed; std::vector<hso_ba_edge> edges; std::vector<double> uv, chi2, idist_in;
    std::vector<hso_se3> poses_in, poses_out; int32_t n_points = 0, status = 0;
  };
  std::vector<BaWindow> ba_windows;
};

static int fail(hso_gpu_ctx* c, int code, const char* msg) { if (c) c->err = msg; return code; }
static FakeFrame* frame_of(hso_gpu_ctx* c, int64_t id) { auto it = c->frames.find(id); return it == c->frames.end() ? nullptr : it->second; }

extern "C" {

int hso_gpu_abi_version(void) { return HSO_GPU_ABI_VERSION; }
int hso_gpu_create(hso_gpu_ctx** out, int, void*) { *out = new hso_gpu_ctx(); return HSO_OK; }
void hso_gpu_destroy(hso_gpu_ctx* c)
{
  if (!c) return;
  for (auto& kv : c->frames) delete kv.second;
  for (auto* m : c->maps) delete m;
  for (auto* t : c->tables) delete t;
  delete c;
}
const char* hso_gpu_last_error(const hso_gpu_ctx* c) { return c ? c->err.c_str() : "null context"; }
int hso_gpu_synchronize(hso_gpu_ctx*) { return HSO_OK; }
int hso_gpu_set_shared_device(hso_gpu_ctx*, int) { return HSO_OK; }
int hso_gpu_device_cpulist(hso_gpu_ctx*, char* out, size_t cap) { if (out && cap) out[0] = 0; return HSO_OK; }   // no device, no node
int hso_gpu_configure(hso_gpu_ctx*, const hso_gpu_options*) { return HSO_OK; }   // kernel shapes and wait modes: nothing to choose here
int hso_gpu_set_host_parallel(hso_gpu_ctx*, hso_parallel_for_fn, void*) { return HSO_OK; }   // the restatement is sequential
int hso_gpu_host_alloc(hso_gpu_ctx*, size_t bytes, void** out) { *out = malloc(bytes ? bytes : 1); return *out ? HSO_OK : HSO_E_NOMEM; }
int hso_gpu_host_free(hso_gpu_ctx*, void* p) { free(p); return HSO_OK; }

int hso_gpu_frame_upload_batch(hso_gpu_ctx* c, const int64_t* ids, const uint8_t* const* imgs, int n, int w, int h, int, hso_frame_stats* st)
{
  for (int i = 0; i < n; i++) {
    if (frame_of(c, ids[i])) { delete c->frames[ids[i]]; c->frames.erase(ids[i]); }
    FakeFrame* F = new FakeFrame();
    F->w = w; F->h = h;
    uint8_t* lv[HSO_N_PYR_LEVELS];
    for (int L = 0; L < HSO_N_PYR_LEVELS; L++) { hso_or_pyramid_dims(w, h, L, &F->lw[L], &F->lh[L]); F->lev[L].assign((size_t)F->lw[L] * F->lh[L] + (size_t)F->lw[L] + 64, 0); lv[L] = F->lev[L].data(); F->pyr[L] = lv[L]; }
    hso_or_create_pyramid(imgs[i], w, h, lv);
    for (int L = 0; L < HSO_N_SOBEL_LEVELS; L++) {
      F->gx[L].assign((size_t)F->lw[L] * F->lh[L], 0); F->gy[L].assign((size_t)F->lw[L] * F->lh[L], 0);
      hso_or_sobel5(F->lev[L].data(), F->lw[L], F->lh[L], F->gx[L].data(), F->gy[L].data());
      F->sx[L] = F->gx[L].data(); F->sy[L] = F->gy[L].data();
    }
    hso_frame_stats s{};
    hso_or_frame_stats(F->lev[0].data(), F->gx[0].data(), F->gy[0].data(), w, h, &s);
    s.width = w; s.height = h;
    if (st) st[i] = s;
    c->frames[ids[i]] = F;
  }
  return HSO_OK;
}
int hso_gpu_frame_release(hso_gpu_ctx* c, int64_t id)
{
  FakeFrame* F = frame_of(c, id);
  if (!F) return fail(c, HSO_E_NOFRAME, "frame_release: not resident");
  for (auto* t : c->tables) if (t) for (size_t i = 0; i < t->s.size(); i++) if (t->alive[i] && t->s[i].ref_frame_id == id) return fail(c, HSO_E_INVALID, "frame_release: hosts live seeds");
  delete F; c->frames.erase(id);
  return HSO_OK;
}

int hso_gpu_frame_release_batch(hso_gpu_ctx* c, const int64_t* ids, int n)
{
  for (int i = 0; i < n; i++) {
    if (!frame_of(c, ids[i])) return fail(c, HSO_E_NOFRAME, "frame_release_batch: not resident");
    for (auto* t : c->tables) if (t) for (size_t k = 0; k < t->s.size(); k++) if (t->alive[k] && t->s[k].ref_frame_id == ids[i]) return fail(c, HSO_E_INVALID, "frame_release_batch: hosts live seeds");
  }
  for (int i = 0; i < n; i++) if (FakeFrame* F = frame_of(c, ids[i])) { delete F; c->frames.erase(ids[i]); }
  return HSO_OK;
}

int hso_gpu_coarse_track_batch(hso_gpu_ctx* c, const hso_camera* cam, const hso_track_params* p, const hso_track_job* jobs, int n, hso_track_result* res)
{
  for (int i = 0; i < n; i++) {
    FakeFrame* R = frame_of(c, jobs[i].ref_frame_id); FakeFrame* C = frame_of(c, jobs[i].cur_frame_id);
    if (!R || !C) return fail(c, HSO_E_NOFRAME, "coarse_track: frame not resident");
    std::vector<hso_ref_feat> rec;
    const hso_ref_feat* feats = jobs[i].feats;
    if (jobs[i].feats_soa) {
      const size_t n = (size_t)jobs[i].n_feats, st = (n + 31) & ~size_t(31);
      const double* a = reinterpret_cast<const double*>(jobs[i].feats);
      rec.resize(n);
      for (size_t q = 0; q < n; q++) { rec[q].px[0] = a[q]; rec[q].px[1] = a[st + q]; rec[q].f[0] = a[2 * st + q]; rec[q].f[1] = a[3 * st + q]; rec[q].f[2] = a[4 * st + q]; rec[q].dist = a[5 * st + q]; }
      feats = rec.data();
    }
    hso_or_tracker* t = hso_or_tracker_create(cam, p, R->pyr, C->pyr, R->w, R->h, feats, jobs[i].n_feats);
    memset(&res[i], 0, sizeof(res[i]));
    hso_or_tracker_run(t, &jobs[i].T_cur_ref, jobs[i].exposure_rat, &res[i]);
    hso_or_tracker_destroy(t);
  }
  return HSO_OK;
}

// ---- sequence maps
int hso_gpu_seqmap_create(hso_gpu_ctx* c, int* out) { c->maps.push_back(new FakeMap()); *out = (int)c->maps.size() - 1; return HSO_OK; }
int hso_gpu_seqmap_destroy(hso_gpu_ctx* c, int m) { delete c->maps[m]; c->maps[m] = nullptr; return HSO_OK; }
int hso_gpu_seqmap_configure(hso_gpu_ctx* c, int m, int fts_cap) { if (fts_cap < 1) return fail(c, HSO_E_INVALID, "seqmap_configure"); c->maps[m]->fts_cap = fts_cap; return HSO_OK; }
int hso_gpu_seqmap_set_keyframes(hso_gpu_ctx* c, int m, const hso_kf* kfs, int n)
{
  for (int k = 0; k < n; k++) if (!frame_of(c, kfs[k].frame_id)) return fail(c, HSO_E_NOFRAME, "seqmap_set_keyframes: keyframe not resident");
  FakeMap* M = c->maps[m];
  M->kfs.assign(kfs, kfs + n);
  M->keys.resize(5 * (size_t)n, -1);
  M->kf_fts.resize((size_t)n);
  return HSO_OK;
}
int hso_gpu_seqmap_set_key_points(hso_gpu_ctx* c, int m, const int32_t* keys, int n)
{
  FakeMap* M = c->maps[m];
  if ((size_t)n != M->kfs.size()) return fail(c, HSO_E_INVALID, "seqmap_set_key_points: one row per keyframe");
  for (int i = 0; i < 5 * n; i++) if (keys[i] < -1 || (keys[i] >= 0 && (size_t)keys[i] >= M->pts.size())) return fail(c, HSO_E_INVALID, "seqmap_set_key_points: point row out of range");
  M->keys.assign(keys, keys + 5 * (size_t)n);
  return HSO_OK;
}
// a patched point row: key and bad flag are the caller's, a counter too unless its KEEP bit says the device's stays
static void store_point(hso_map_point& dst, const hso_map_point& src)
{
  uint32_t w = (uint32_t)src.pad_;
  const uint32_t old = (uint32_t)dst.pad_;
  if (w & HSO_PT_KEEP_NFAIL) w = (w & ~(0x3ffu << 8)) | (old & (0x3ffu << 8));
  if (w & HSO_PT_KEEP_NOK) w = (w & ~(0x7ffu << 20)) | (old & (0x7ffu << 20));
  w &= ~(HSO_PT_KEEP_NFAIL | HSO_PT_KEEP_NOK);
  dst = src;
  dst.pad_ = (int32_t)w;
}
static int patch_rows(hso_gpu_ctx* c, int m, const int32_t* pid, const hso_map_point* pts, int np, const int32_t* oid, const hso_obs* obs, int no, const int32_t* olink)
{
  FakeMap* M = c->maps[m];
  const int nk = (int)M->kfs.size();
  for (int i = 0; i < no; i++) {
    if (oid[i] < 0 || obs[i].kf < 0 || obs[i].kf >= nk) return fail(c, HSO_E_INVALID, "seqmap_patch: observation row out of range");
    if ((size_t)oid[i] >= M->obs.size()) { M->obs.resize((size_t)oid[i] + 1, hso_obs{}); M->obs_pt.resize((size_t)oid[i] + 1, -1); }
    M->obs[(size_t)oid[i]] = obs[i];
    if (olink) M->obs_pt[(size_t)oid[i]] = olink[i];
  }
  for (int i = 0; i < np; i++) {
    if (pid[i] < 0 || pts[i].host_kf < 0 || pts[i].host_kf >= nk || (pts[i].obs_count > 0 && (pts[i].obs_begin < 0 || (size_t)pts[i].obs_begin >= M->obs.size())))
      return fail(c, HSO_E_INVALID, "seqmap_patch: point row out of range");
    if ((size_t)pid[i] >= M->pts.size()) M->pts.resize((size_t)pid[i] + 1, hso_map_point{});
    store_point(M->pts[(size_t)pid[i]], pts[i]);
  }
  return HSO_OK;
}
int hso_gpu_seqmap_patch(hso_gpu_ctx* c, int m, const int32_t* pid, const hso_map_point* pts, int np, const int32_t* oid, const hso_obs* obs, int no)
{
  return patch_rows(c, m, pid, pts, np, oid, obs, no, nullptr);
}
int hso_gpu_seqmap_patch_multi(hso_gpu_ctx* c, const hso_seqmap_rows* p, int n)
{
  for (int i = 0; i < n; i++)
    if (int rc = patch_rows(c, p[i].map, p[i].point_ids, p[i].points, p[i].n_points, p[i].obs_ids, p[i].obs, p[i].n_obs, p[i].obs_point)) return rc;
  return HSO_OK;
}
int hso_gpu_seqmap_patch_links(hso_gpu_ctx* c, int m, const int32_t* oid, const int32_t* link, int n)
{
  FakeMap* M = c->maps[m];
  for (int i = 0; i < n; i++) { if (oid[i] < 0 || (size_t)oid[i] >= M->obs.size()) return fail(c, HSO_E_INVALID, "seqmap_patch_links: row out of range"); M->obs_pt[(size_t)oid[i]] = link[i]; }
  return HSO_OK;
}
int hso_gpu_seqmap_patch_lists(hso_gpu_ctx* c, const hso_seqmap_list_patch* p, int n)
{
  for (int i = 0; i < n; i++) {
    FakeMap* M = c->maps[p[i].map];
    std::vector<int32_t>* L = nullptr;
    if (p[i].list == HSO_LIST_CANDIDATES) L = &M->cands;
    else {
      if (p[i].list < 0 || (size_t)p[i].list >= M->kf_fts.size()) return fail(c, HSO_E_INVALID, "seqmap_patch_lists: no such keyframe row");
      if (p[i].first + p[i].n > M->fts_cap) return fail(c, HSO_E_INVALID, "seqmap_patch_lists: a keyframe's feature list outgrows fts_cap");
      L = &M->kf_fts[(size_t)p[i].list];
    }
    if (p[i].first < 0 || (size_t)p[i].first > L->size()) return fail(c, HSO_E_INVALID, "seqmap_patch_lists: the patch leaves a gap");
    L->resize((size_t)p[i].first);
    L->insert(L->end(), p[i].ids, p[i].ids + p[i].n);
  }
  return HSO_OK;
}
int hso_gpu_seqmap_size(hso_gpu_ctx* c, int m, int* nk, int* np, int* no)
{
  FakeMap* M = c->maps[m];
  if (nk) *nk = (int)M->kfs.size();
  if (np) *np = (int)M->pts.size();
  if (no) *no = (int)M->obs.size();
  return HSO_OK;
}
int hso_gpu_seqmap_read(hso_gpu_ctx* c, int m, const int32_t* pid, int np, hso_map_point* pts, const int32_t* oid, int no, hso_obs* obs)
{
  FakeMap* M = c->maps[m];
  for (int i = 0; i < np; i++) pts[i] = M->pts.at((size_t)pid[i]);
  for (int i = 0; i < no; i++) obs[i] = M->obs.at((size_t)oid[i]);
  return HSO_OK;
}
int hso_gpu_seq_frame_features(hso_gpu_ctx* c, const int32_t* maps, const int64_t* ids, int n, hso_seq_feature* out, int cap, int32_t* n_out)
{
  for (int i = 0; i < n; i++) {
    FakeMap* M = c->maps[maps[i]];
    const int b = M->ff_frame[0] == ids[i] ? 0 : (M->ff_frame[1] == ids[i] ? 1 : -1);
    if (b < 0) return fail(c, HSO_E_NOFRAME, "seq_frame_features: the map holds no feature table of that frame");
    if ((int)M->ff[b].size() > cap) return fail(c, HSO_E_INVALID, "seq_frame_features: cap is smaller than the table");
    n_out[i] = (int)M->ff[b].size();
    std::copy(M->ff[b].begin(), M->ff[b].end(), out + (size_t)i * cap);
  }
  return HSO_OK;
}
int hso_gpu_seq_set_frame_features(hso_gpu_ctx* c, int m, int64_t id, const hso_seq_feature* f, int n)
{
  FakeMap* M = c->maps[m];
  const int b = M->ff_frame[0] == id ? 0 : (M->ff_frame[1] == id ? 1 : 1 - M->ff_newest);
  M->ff[b].assign(f, f + n); M->ff_frame[b] = id; M->ff_newest = b;
  return HSO_OK;
}

// Reprojector::reprojectCellAll / the three reprojectCell passes over per-cell lists: which candidates are examined, in which
// order, which become features.  cand: (cell, quality, matched) in projection order.
struct Cand { int cell; uint8_t quality; bool matched; };
static void select_walk(const std::vector<Cand>& cand, const int32_t* cell_order, int n_cells, int max_fts, std::vector<std::pair<int, bool>>& examined,
                        int32_t counts[4])
{
  examined.clear();
  int n_matches = 0;
  counts[2] = 0; counts[3] = 0;
  if ((int)cand.size() < max_fts + 50) {
    for (size_t i = 0; i < cand.size(); i++) {
      examined.push_back({(int)i, cand[i].matched});
      if (cand[i].matched && ++n_matches >= max_fts) break;
    }
    counts[0] = (int)examined.size(); counts[1] = n_matches;
    return;
  }
  std::vector<std::list<int>> cells((size_t)n_cells);
  for (size_t i = 0; i < cand.size(); i++) cells[(size_t)cand[i].cell].push_back((int)i);
  auto visit = [&](std::list<int>& cell, bool second, bool third) {
    if (cell.empty()) return false;
    if (!second) cell.sort([&](int a, int b) { return cand[a].quality > cand[b].quality; });   // stable: ties keep projection order
    int got = 0;
    while (!cell.empty()) {
      const int i = cell.front(); cell.pop_front();
      examined.push_back({i, cand[i].matched});
      if (!cand[i].matched) continue;
      if (!third) return true;
      ++got; ++n_matches;
      if (got >= 3 || n_matches >= max_fts) return true;
    }
    return false;
  };
  counts[3] = 1; counts[2] = 1;
  for (int k = 0; k < n_cells; k++) { if (visit(cells[(size_t)cell_order[k]], false, false)) ++n_matches; if (n_matches >= max_fts) break; }
  if (n_matches < max_fts) {
    counts[2] = 2;
    for (int k = n_cells - 1; k > 0; k--) { if (visit(cells[(size_t)cell_order[k]], true, false)) ++n_matches; if (n_matches >= max_fts) break; }
  }
  if (n_matches < max_fts) {
    counts[2] = 3;
    for (int k = 0; k < n_cells; k++) { visit(cells[(size_t)cell_order[k]], true, true); if (n_matches >= max_fts) break; }
  }
  counts[0] = (int)examined.size(); counts[1] = n_matches;
}

int hso_gpu_seed_table_observe_groups(hso_gpu_ctx* c, const hso_camera* cam, int t, const hso_seed_frame* frames, int n_frames, double px_error_angle,
                                      hso_seed_brief* brief, float* px, hso_seed_out* full);

// hso_gpu_seq_chain on the restatement: FrameHandlerMono::processFrame from the motion prior to the inputs of its decisions
// (src/frame_handler_mono.cpp:173-291), one sequence after the other, every step the sequential way the reference does it.
int hso_gpu_seq_chain(hso_gpu_ctx* c, const hso_camera* cam, const hso_seq_chain_cfg* cfg, const hso_seq_job* jobs, int n, const int32_t* temps, int,
                      hso_seq_result* results)
{
  const int cap = std::max(cfg->max_fts, 1), n_cells = cfg->n_cells;
  c->dbg_proj.clear(); c->dbg_match.clear(); c->dbg_brief.clear(); c->dbg_projected.clear();
  c->dbg_feats.assign((size_t)n * cap, hso_pose_feat{}); c->dbg_poses.assign((size_t)n * 128, hso_se3{}); c->dbg_nposes.assign((size_t)n, 0);
  c->dbg_mask.assign((size_t)n * cap, 0); c->dbg_slices.assign((size_t)n + 1, 0); c->dbg_exbegin.assign((size_t)n + 1, 0);
  c->last_ids.assign((size_t)n, {}); c->last_events.assign((size_t)n, {}); c->last_quality.assign((size_t)n, {}); c->last_table.assign((size_t)n, {});
  for (int j = 0; j < n; j++) {
    const hso_seq_job& J = jobs[j];
    hso_seq_result& R = results[j];
    memset(&R, 0, sizeof(R));
    FakeMap* M = c->maps[J.map];
    FakeFrame* C = frame_of(c, J.cur_frame_id); FakeFrame* Rf = frame_of(c, J.ref_frame_id);
    if (!C || !Rf) return fail(c, HSO_E_NOFRAME, "seq_chain: frame not resident");
    const int nk = (int)M->kfs.size();
    // ---- CoarseTracker::makeDepthRef (src/CoarseTracker.cpp:210-240) + run (:51-208) + the write-back (:198-202)
    const bool no_track = (J.flags & HSO_SEQ_NO_TRACK) != 0 || J.n_ref_feats == 0;
    int ref_buf = -1;
    std::vector<hso_ref_feat>& table = c->last_table[(size_t)j];
    if (!no_track) {
      std::vector<int32_t> pt;
      if (J.ref_kf_row >= 0) {
        const std::vector<int32_t>& L = M->kf_fts.at((size_t)J.ref_kf_row);
        if ((int)L.size() != J.n_ref_feats) return fail(c, HSO_E_INVALID, "seq_chain: n_ref_feats differs from the keyframe's list");
        for (int32_t f : L) { const hso_obs& o = M->obs.at((size_t)f); hso_ref_feat r{}; r.px[0] = o.px[0]; r.px[1] = o.px[1]; r.f[0] = o.f[0]; r.f[1] = o.f[1]; r.f[2] = o.f[2]; table.push_back(r); pt.push_back(M->obs_pt.at((size_t)f)); }
      } else {
        ref_buf = M->ff_frame[0] == J.ref_frame_id ? 0 : (M->ff_frame[1] == J.ref_frame_id ? 1 : -1);
        if (ref_buf < 0) return fail(c, HSO_E_NOFRAME, "seq_chain: the map holds no feature table of the reference frame");
        if ((int)M->ff[ref_buf].size() != J.n_ref_feats) return fail(c, HSO_E_INVALID, "seq_chain: n_ref_feats differs from the reference frame's table");
        for (const hso_seq_feature& q : M->ff[ref_buf]) { hso_ref_feat r{}; r.px[0] = q.px[0]; r.px[1] = q.px[1]; r.f[0] = q.f[0]; r.f[1] = q.f[1]; r.f[2] = q.f[2]; table.push_back(r); pt.push_back(q.point); }
      }
      for (size_t i = 0; i < table.size(); i++) {
        table[i].dist = -1;
        if (pt[i] < 0 || (size_t)pt[i] >= M->pts.size()) continue;
        const hso_map_point& P = M->pts[(size_t)pt[i]];
        if (P.idist == 0.0) continue;
        hso_se3 inv, T_ref_host;
        hso_or_se3_inverse(&M->kfs[(size_t)P.host_kf].T_f_w, &inv);
        hso_or_se3_mul(&J.T_ref_w, &inv, &T_ref_host);
        const double s = 1.0 / P.idist, in_host[3] = {P.host_f[0] * s, P.host_f[1] * s, P.host_f[2] * s};
        double q[3];
        hso_or_se3_apply(&T_ref_host, in_host, q);
        if (!(q[2] < 0.00001)) table[i].dist = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2]);
      }
    }
    hso_se3 T_cur = J.T_cur_w;
    double exposure = -1.0;
    if (!no_track) {
      hso_se3 inv, T_cur_ref;
      hso_or_se3_inverse(&J.T_ref_w, &inv);
      hso_or_se3_mul(&J.T_cur_w, &inv, &T_cur_ref);
      hso_or_tracker* t = hso_or_tracker_create(cam, &cfg->track, Rf->pyr, C->pyr, Rf->w, Rf->h, table.data(), (int)table.size());
      hso_or_tracker_run(t, &T_cur_ref, J.exposure_rat, &R.track);
      hso_or_tracker_destroy(t);
      hso_or_se3_mul(&R.track.T_cur_ref, &J.T_ref_w, &T_cur);
      exposure = (double)R.track.exposure_rat * J.ref_exposure;
      if (R.track.exposure_rat > 0.99f && R.track.exposure_rat < 1.01f) exposure = J.ref_exposure;
    }
    R.T_tracked = T_cur; R.exposure = exposure;
    hso_se3 Tinv; hso_or_se3_inverse(&T_cur, &Tinv);
    // ---- which keyframes the frame visits (src/reprojector.cpp:108-199; Map::getCloseKeyframes, src/map.cpp:193-213)
    std::vector<int> visit;
    for (int q = 0; q < 5; q++) { const int r = J.covis[q]; if (r >= 0 && r < nk && std::find(visit.begin(), visit.end(), r) == visit.end()) visit.push_back(r); }
    {
      std::vector<std::pair<double, int>> near;
      for (int k = 0; k < nk; k++)
        for (int q = 0; q < 5; q++) {
          const int p = M->keys[5 * (size_t)k + q];
          if (p < 0 || (size_t)p >= M->pts.size()) continue;
          double xc[3], px[2];
          hso_or_se3_apply(&T_cur, M->pts[(size_t)p].pos, xc);
          if (xc[2] < 0.0) continue;
          hso_or_world2cam(cam, xc, px);
          if (!(px[0] >= 0.0 && px[1] >= 0.0 && px[0] < cam->width && px[1] < cam->height)) continue;
          const double* a = T_cur.t; const double* b = M->kfs[(size_t)k].T_f_w.t;
          near.push_back({std::sqrt((a[0] - b[0]) * (a[0] - b[0]) + (a[1] - b[1]) * (a[1] - b[1]) + (a[2] - b[2]) * (a[2] - b[2])), k});
          break;
        }
      std::stable_sort(near.begin(), near.end(), [](const std::pair<double, int>& a, const std::pair<double, int>& b) { return a.first < b.first; });
      size_t cnt = visit.size();
      for (size_t i = 0; i < near.size() && (int)cnt < cfg->max_kfs && visit.size() < HSO_SEQ_MAX_VISIT; i++) {
        if (std::find(visit.begin(), visit.end(), near[i].second) != visit.end()) continue;
        visit.push_back(near[i].second); ++cnt;
      }
    }
    R.n_visit = (int)visit.size();
    for (int q = 0; q < HSO_SEQ_MAX_VISIT; q++) R.visit[q] = q < (int)visit.size() ? visit[(size_t)q] : -1;
    // ---- the list: per visited keyframe the points of its features once each, then candidates, then temporary points
    std::vector<int32_t>& ids = c->last_ids[(size_t)j]; std::vector<uint8_t>& quality = c->last_quality[(size_t)j];
    {
      std::vector<uint8_t> stamped(M->pts.size(), 0);
      for (int r : visit)
        for (int32_t f : M->kf_fts.at((size_t)r)) {
          const int p = M->obs_pt.at((size_t)f);
          if (p < 0 || (size_t)p >= M->pts.size()) continue;
          const int key = (int)HSO_PT_KEY(M->pts[(size_t)p].pad_), kind = key >> 4;
          if (kind == 0 || kind == 1 || stamped[(size_t)p]) continue;
          stamped[(size_t)p] = 1;
          ids.push_back(p); quality.push_back((uint8_t)key);
        }
    }
    R.n_kf_points = (int)ids.size();
    for (int32_t p : M->cands) { const int key = (int)HSO_PT_KEY(M->pts.at((size_t)p).pad_); if ((key >> 4) != 2) continue; ids.push_back(p); quality.push_back((uint8_t)key); }
    R.n_candidates = (int)ids.size() - R.n_kf_points;
    for (int q = 0; q < J.n_temps; q++) { const int32_t p = temps[J.temps_begin + q]; ids.push_back(p); quality.push_back((uint8_t)HSO_PT_KEY(M->pts.at((size_t)p).pad_)); }
    const int n_listed = (int)ids.size();
    R.n_listed = n_listed;
    // ---- projection, reference choice, findMatchDirect
    std::vector<hso_reproj_point> proj((size_t)n_listed); std::vector<hso_align_out> match((size_t)n_listed);
    std::vector<hso_match_brief> brief((size_t)n_listed);
    std::vector<Cand> cand; std::vector<int> cand_at;
    for (int i = 0; i < n_listed; i++) {
      hso_reproj_point& r = proj[(size_t)i];
      memset(&r, 0, sizeof(r)); r.ref_obs = -1;
      memset(&match[(size_t)i], 0, sizeof(hso_align_out));
      hso_match_brief& b = brief[(size_t)i];
      memset(&b, 0, sizeof(b)); b.cell = -1; b.ref_obs = -1;
      const hso_map_point& P = M->pts[(size_t)ids[(size_t)i]];
      int cell = 0;
      if (!hso_or_reproject_point(cam, &T_cur, &M->kfs[(size_t)P.host_kf].T_f_w, P.host_f, P.idist, cfg->cell_size, cfg->grid_n_cols, r.px, &cell)) continue;
      r.projected = 1; r.cell = cell;
      b.cell = cell; b.px[0] = r.px[0]; b.px[1] = r.px[1];
      std::vector<hso_obs> chain; std::vector<int> rows;
      for (int q = 0, row = P.obs_begin; q < P.obs_count; q++) { chain.push_back(M->obs.at((size_t)row)); rows.push_back(row); row = M->obs[(size_t)row].pad_; }
      const int k = chain.empty() ? -1 : hso_or_close_view_obs(Tinv.t, P.pos, M->kfs.data(), chain.data(), (int)chain.size());
      bool matched = false;
      if (k >= 0) {
        r.ref_obs = rows[(size_t)k];
        b.ref_obs = r.ref_obs;
        hso_align_job job;
        hso_or_reproject_make_job(&T_cur, exposure, J.cur_keyframe_id, M->kfs.data(), &P, &chain[(size_t)k], r.px, &job);
        FakeFrame* Rk = frame_of(c, M->kfs[(size_t)chain[(size_t)k].kf].frame_id);
        hso_align_out& m = match[(size_t)i];
        hso_or_find_match_direct(cam, &job, Rk->pyr, C->pyr, C->sx, C->sy, C->w, C->h, &m);
        matched = m.success != 0;
        b.px_cur[0] = m.px_cur[0]; b.px_cur[1] = m.px_cur[1];
        b.stage = (int8_t)m.stage; b.search_level = (int8_t)m.search_level; b.ref_type = (int8_t)job.type;
        const double gx = m.A_cur_ref[0] * job.grad[0] + m.A_cur_ref[1] * job.grad[1], gy = m.A_cur_ref[2] * job.grad[0] + m.A_cur_ref[3] * job.grad[1];
        const double nn = std::sqrt(gx * gx + gy * gy);
        b.grad[0] = nn > 0 ? (float)(gx / nn) : 0.f; b.grad[1] = nn > 0 ? (float)(gy / nn) : 0.f;
      }
      cand.push_back({cell, quality[(size_t)i], matched && (quality[(size_t)i] >> 4) != 0});
      cand_at.push_back(i);
    }
    // ---- the grid selection, the frame's features, the pose optimiser
    std::vector<std::pair<int, bool>> ex;
    select_walk(cand, cfg->cell_order, n_cells, cfg->max_fts, ex, R.counts);
    c->dbg_exbegin[(size_t)j] = (int32_t)c->dbg_brief.size();
    std::vector<hso_pose_feat> feats; std::vector<hso_se3> poses;
    std::vector<hso_seq_feature> ff;
    std::vector<hso_frame_match> records;
    for (auto& e : ex) {
      const int i = cand_at[(size_t)e.first];
      hso_match_brief b = brief[(size_t)i];
      b.success = e.second ? 1 : 0; b.pad_ = i;
      c->dbg_brief.push_back(b);
      hso_frame_match m{};
      m.px_cur[0] = b.px_cur[0]; m.px_cur[1] = b.px_cur[1]; m.grad[0] = b.grad[0]; m.grad[1] = b.grad[1];
      m.point = i; m.success = b.success; m.search_level = b.search_level; m.ref_type = b.ref_type;
      records.push_back(m);
      if (!e.second || (int)feats.size() >= cap) continue;
      const hso_map_point& P = M->pts[(size_t)ids[(size_t)i]];
      hso_pose_feat pf{};
      pf.has_point = 1; pf.type = b.ref_type; pf.level = b.search_level; pf.temporary = ((quality[(size_t)i] >> 4) == 1) ? 1 : 0;
      hso_or_cam2world(cam, b.px_cur[0], b.px_cur[1], pf.f);
      pf.grad[0] = b.grad[0]; pf.grad[1] = b.grad[1];
      pf.host_f[0] = P.host_f[0]; pf.host_f[1] = P.host_f[1]; pf.host_f[2] = P.host_f[2]; pf.idist = P.idist;
      pf.host_pose = P.host_kf;
      feats.push_back(pf);
      hso_seq_feature q{};
      q.px[0] = b.px_cur[0]; q.px[1] = b.px_cur[1]; q.f[0] = pf.f[0]; q.f[1] = pf.f[1]; q.f[2] = pf.f[2]; q.grad[0] = b.grad[0]; q.grad[1] = b.grad[1];
      q.point = ids[(size_t)i]; q.level = b.search_level; q.type = b.ref_type;
      ff.push_back(q);
    }
    std::vector<int> used;
    for (auto& pf : feats) used.push_back(pf.host_pose);
    std::sort(used.begin(), used.end()); used.erase(std::unique(used.begin(), used.end()), used.end());
    for (auto& pf : feats) pf.host_pose = (int)(std::lower_bound(used.begin(), used.end(), pf.host_pose) - used.begin());
    for (int k : used) poses.push_back(M->kfs[(size_t)k].T_f_w);
    hso_pose_job pjob{};
    pjob.feats = feats.data(); pjob.n_feats = (int)feats.size(); pjob.poses_f_w = poses.data(); pjob.n_poses = (int)poses.size();
    pjob.T_f_w = T_cur; pjob.reproj_thresh = cfg->pose_reproj_thresh; pjob.n_iter = cfg->pose_n_iter;
    std::vector<uint8_t> mask(std::max(feats.size(), (size_t)1), 0);
    R.pose.status = 1; R.pose.T_f_w = T_cur;
    if (!feats.empty()) { memset(&R.pose, 0, sizeof(R.pose)); hso_or_pose_optimize(cam, &pjob, &R.pose, mask.data()); }
    R.n_feats = (int)feats.size();
    std::copy(feats.begin(), feats.end(), c->dbg_feats.begin() + (std::ptrdiff_t)((size_t)j * cap));
    std::copy(poses.begin(), poses.begin() + (std::ptrdiff_t)std::min(poses.size(), (size_t)128), c->dbg_poses.begin() + (std::ptrdiff_t)((size_t)j * 128));
    std::copy(mask.begin(), mask.begin() + (std::ptrdiff_t)feats.size(), c->dbg_mask.begin() + (std::ptrdiff_t)((size_t)j * cap));
    c->dbg_nposes[(size_t)j] = (int)poses.size();
    c->dbg_slices[(size_t)j] = (int32_t)c->dbg_proj.size();
    c->dbg_proj.insert(c->dbg_proj.end(), proj.begin(), proj.end());
    c->dbg_match.insert(c->dbg_match.end(), match.begin(), match.end());
    for (int i = 0; i < n_listed; i++) c->dbg_projected.push_back((uint8_t)proj[(size_t)i].projected);
    // ---- what reprojectCell / reprojectCellAll do with the candidates they examine (src/reprojector.cpp:366-425, 214-222, 247-251)
    std::vector<int32_t>& ev = c->last_events[(size_t)j];
    auto word = [&](int p) -> uint32_t& { return reinterpret_cast<uint32_t&>(M->pts[(size_t)p].pad_); };
    auto add_fail = [&](uint32_t& w, uint32_t k) { uint32_t nf = HSO_PT_NFAIL(w) + k; if (nf > 1023) nf = 1023; w = (w & ~(0x3ffu << 8)) | (nf << 8); return nf; };
    for (int i = R.n_kf_points; i < n_listed; i++) {
      if (proj[(size_t)i].projected) continue;
      const int p = ids[(size_t)i];
      uint32_t& w = word(p);
      if (add_fail(w, 3) <= 30) continue;
      if (i < R.n_kf_points + R.n_candidates) { w &= ~0xf0u; ev.push_back((HSO_EV_ERASE_CANDIDATE << 28) | p); }
      else { w |= HSO_PT_BAD; ev.push_back((HSO_EV_TEMP_BAD << 28) | p); }
    }
    for (const hso_frame_match& r : records) {
      const int p = ids[(size_t)r.point];
      uint32_t& w = word(p);
      const uint32_t kind = (w & 0xffu) >> 4;
      if (kind == 0) continue;
      if (!r.success) {
        const uint32_t nf = add_fail(w, 1);
        if (kind == 3 && nf > 15) { w &= ~0xf0u; ev.push_back((HSO_EV_ERASE_POINT << 28) | p); }
        else if (kind == 2 && nf > 30) { w &= ~0xf0u; ev.push_back((HSO_EV_ERASE_CANDIDATE << 28) | p); }
        else if (kind == 1 && nf > 30) { w |= HSO_PT_BAD; ev.push_back((HSO_EV_TEMP_BAD << 28) | p); }
        continue;
      }
      uint32_t nk2 = HSO_PT_NOK(w) + 1; if (nk2 > 2047) nk2 = 2047;
      w = (w & ~(0x7ffu << 20)) | (nk2 << 20);
      if (kind == 3 && nk2 > 10) { w = (w & ~0xf0u) | (4u << 4); ev.push_back((HSO_EV_GOOD << 28) | p); }
    }
    R.n_events = (int)ev.size();
    for (int q = 0; q < HSO_SEQ_EVENTS; q++) R.events[q] = q < (int)ev.size() ? ev[(size_t)q] : 0;
    // ---- the pose optimiser's culling, when processFrame gets that far (:224-243)
    const bool used_pose = R.counts[1] >= cfg->quality_min_fts && R.pose.status == 0 && !((J.flags & HSO_SEQ_SEED_BRANCH) && R.counts[1] < 100);
    if (used_pose) for (size_t i = 0; i < ff.size(); i++) if (mask[i]) ff[i].point = -1;
    const hso_se3 T_fin = used_pose ? R.pose.T_f_w : T_cur;
    // ---- getSceneDepth / getSceneDistance (src/frame.cpp:323-366)
    {
      std::vector<double> z, r;
      R.depth_min = std::numeric_limits<double>::max();
      for (const hso_seq_feature& q : ff) {
        if (q.point < 0) continue;
        double xc[3];
        hso_or_se3_apply(&T_fin, M->pts[(size_t)q.point].pos, xc);
        z.push_back(xc[2]); r.push_back(std::sqrt(xc[0] * xc[0] + xc[1] * xc[1] + xc[2] * xc[2]));
        R.depth_min = std::fmin(xc[2], R.depth_min);
      }
      R.n_with_point = (int)z.size();
      if (!z.empty()) {
        std::nth_element(z.begin(), z.begin() + (std::ptrdiff_t)(z.size() / 2), z.end()); R.depth_median = z[z.size() / 2];
        std::nth_element(r.begin(), r.begin() + (std::ptrdiff_t)(r.size() / 2), r.end()); R.dist_median = r[r.size() / 2];
      }
    }
    // ---- createCovisibilityGraph (src/frame_handler_mono.cpp:559-647)
    {
      std::vector<int> votes((size_t)nk, 0);
      for (const hso_seq_feature& q : ff) {
        if (q.point < 0) continue;
        const hso_map_point& P = M->pts[(size_t)q.point];
        for (int t = 0, o = P.obs_begin; t < P.obs_count && o >= 0; t++) { votes[(size_t)M->obs[(size_t)o].kf]++; o = M->obs[(size_t)o].pad_; }
      }
      const int need = R.n_with_point > 30 ? 5 : 3;
      std::vector<int> ranked; int best = -1, seen = 0;
      for (int k = 0; k < nk; k++) {
        if (votes[(size_t)k] == 0) continue;
        ++seen;
        if (best < 0 || votes[(size_t)k] > votes[(size_t)best]) best = k;
        if (votes[(size_t)k] >= need) ranked.push_back(k);
      }
      if (ranked.empty() && best >= 0) ranked.push_back(best);
      std::stable_sort(ranked.begin(), ranked.end(), [&](int a, int b) { return votes[(size_t)a] > votes[(size_t)b]; });
      for (int q = 0; q < HSO_SEQ_MAX_COVIS; q++) { R.covis[q] = q < (int)ranked.size() ? ranked[(size_t)q] : -1; R.covis_votes[q] = q < (int)ranked.size() ? votes[(size_t)ranked[(size_t)q]] : 0; }
      R.n_covis = seen; R.covis_best = best;
    }
    // ---- needNewKf's two sums (:428-507)
    if (J.last_kf_row >= 0) {
      const hso_kf& K = M->kfs[(size_t)J.last_kf_row];
      hso_se3 Kinv, T_cur_kf;
      hso_or_se3_inverse(&K.T_f_w, &Kinv);
      hso_or_se3_mul(&T_fin, &Kinv, &T_cur_kf);
      float full = 0, shift = 0; int count = 0;
      for (int32_t f : M->kf_fts.at((size_t)J.last_kf_row)) {
        const int p = M->obs_pt.at((size_t)f);
        if (p < 0 || (HSO_PT_KEY(M->pts[(size_t)p].pad_) >> 4) == 0) continue;
        const hso_obs& o = M->obs[(size_t)f];
        const double* w = M->pts[(size_t)p].pos;
        const double off[3] = {w[0] - Kinv.t[0], w[1] - Kinv.t[1], w[2] - Kinv.t[2]};
        const double len = std::sqrt(off[0] * off[0] + off[1] * off[1] + off[2] * off[2]);
        const double in_kf[3] = {o.f[0] * len, o.f[1] * len, o.f[2] * len};
        double xc[3], a[2], b[2];
        hso_or_se3_apply(&T_cur_kf, in_kf, xc);
        hso_or_world2cam(cam, xc, a);
        const double moved[3] = {in_kf[0] + T_cur_kf.t[0], in_kf[1] + T_cur_kf.t[1], in_kf[2] + T_cur_kf.t[2]};
        hso_or_world2cam(cam, moved, b);
        full += (a[0] - o.px[0]) * (a[0] - o.px[0]) + (a[1] - o.px[1]) * (a[1] - o.px[1]);
        shift += (b[0] - o.px[0]) * (b[0] - o.px[0]) + (b[1] - o.px[1]) * (b[1] - o.px[1]);
        ++count;
      }
      R.flow_full = full; R.flow_shift = shift; R.flow_count = count;
    }
    // ---- needNewKf's answer (:486-506) and whether the frame's seeds are observed behind it (a regular frame: DepthFilter::addFrame)
    {
      bool make_kf = (J.flags & HSO_SEQ_DEPTH_STATS) != 0;
      if (!make_kf && J.last_kf_row >= 0 && R.flow_count > 0) {
        float ff_full = R.flow_full / (float)R.flow_count;
        if (!(ff_full < 133.f)) {
          ff_full = sqrtf(ff_full);
          const float ff_shift = sqrtf(R.flow_shift / (float)R.flow_count);
          const int nominal = 752 + 480;
          const float w_shift = 0.04 * nominal, w_full = 0.02 * nominal, w_global = 0.75;
          const int extent = cam->width + cam->height;
          const float score = w_global * w_shift * ff_shift / extent + w_global * w_full * ff_full / extent;
          make_kf = score > 1;
        }
      }
      R.make_kf = make_kf ? 1 : 0;
      const bool observe = cfg->seed_table >= 0 && J.seed_group >= 0 && R.counts[1] >= cfg->quality_min_fts && R.pose.status == 0 && R.pose.num_obs >= cfg->quality_min_fts &&
                           !((J.flags & HSO_SEQ_SEED_BRANCH) && R.counts[1] < 100) && !make_kf;
      R.seeds_observed = observe ? 1 : 0;
    }
    // ---- the new frame's table becomes the map's newest
    const int cur_buf = ref_buf >= 0 ? 1 - ref_buf : 1 - M->ff_newest;
    M->ff[cur_buf] = ff; M->ff_frame[cur_buf] = J.cur_frame_id; M->ff_newest = cur_buf;
  }
  c->dbg_slices[(size_t)n] = (int32_t)c->dbg_proj.size();
  c->dbg_exbegin[(size_t)n] = (int32_t)c->dbg_brief.size();
  if (cfg->seed_table >= 0) {
    std::vector<hso_seed_frame> fr((size_t)cfg->n_seed_groups);
    for (hso_seed_frame& f : fr) { f = hso_seed_frame{}; f.frame_id = -1; f.T_f_w = hso_se3{{0, 0, 0, 1}, {0, 0, 0}}; f.exposure_time = 1; }
    for (int j = 0; j < n; j++) {
      if (!results[j].seeds_observed) continue;
      hso_seed_frame& f = fr[(size_t)jobs[j].seed_group];
      f.frame_id = jobs[j].cur_frame_id; f.T_f_w = results[j].pose.T_f_w; f.exposure_time = results[j].exposure;
    }
    if ((int)c->tables[cfg->seed_table]->s.size() > cfg->seed_brief_cap) return fail(c, HSO_E_INVALID, "seq_chain: seed_brief_out is smaller than the seed table");
    if (int rc = hso_gpu_seed_table_observe_groups(c, cam, cfg->seed_table, fr.data(), cfg->n_seed_groups, cfg->px_error_angle, cfg->seed_brief_out, nullptr, nullptr)) return rc;
  }
  return HSO_OK;
}

int hso_gpu_seq_events(hso_gpu_ctx* c, int job, int32_t* out, int cap)
{
  if (job < 0 || (size_t)job >= c->last_events.size() || (int)c->last_events[(size_t)job].size() > cap) return fail(c, HSO_E_INVALID, "seq_events: bad argument");
  std::copy(c->last_events[(size_t)job].begin(), c->last_events[(size_t)job].end(), out);
  return (int)c->last_events[(size_t)job].size();
}
int hso_gpu_seq_debug_list(hso_gpu_ctx* c, int job, int32_t* ids, uint8_t* q, int cap)
{
  if (job < 0 || (size_t)job >= c->last_ids.size() || (int)c->last_ids[(size_t)job].size() > cap) return fail(c, HSO_E_INVALID, "seq_debug_list: bad argument");
  std::copy(c->last_ids[(size_t)job].begin(), c->last_ids[(size_t)job].end(), ids);
  std::copy(c->last_quality[(size_t)job].begin(), c->last_quality[(size_t)job].end(), q);
  return (int)c->last_ids[(size_t)job].size();
}
int hso_gpu_seq_debug_ref_table(hso_gpu_ctx* c, int job, hso_ref_feat* out, int cap)
{
  if (job < 0 || (size_t)job >= c->last_table.size() || (int)c->last_table[(size_t)job].size() > cap) return fail(c, HSO_E_INVALID, "seq_debug_ref_table: bad argument");
  std::copy(c->last_table[(size_t)job].begin(), c->last_table[(size_t)job].end(), out);
  return (int)c->last_table[(size_t)job].size();
}

int hso_gpu_seqmap_debug_dump(hso_gpu_ctx* c, int m, int what, void* out, size_t bytes)
{
  if (m < 0 || (size_t)m >= c->maps.size() || !c->maps[m] || !out) return fail(c, HSO_E_INVALID, "seqmap_debug_dump: bad argument");
  FakeMap* M = c->maps[m];
  const size_t nk = M->kfs.size();
  const int64_t sizes[HSO_DUMP_N_SIZES] = {(int64_t)nk, (int64_t)M->pts.size(), (int64_t)M->obs.size(), M->fts_cap, (int64_t)M->cands.size(), (int64_t)M->ff[0].size(), (int64_t)M->ff[1].size(),
                                           M->ff_frame[0], M->ff_frame[1], M->ff_newest, (int64_t)sizeof(hso_kf), (int64_t)sizeof(hso_map_point), (int64_t)sizeof(hso_obs),
                                           (int64_t)sizeof(hso_seq_feature), (int64_t)sizeof(hso_seq_job), (int64_t)sizeof(hso_seq_result)};
  std::vector<int32_t> ints;
  const void* src = nullptr; size_t have = 0;
  switch (what) {
    case HSO_DUMP_SIZES: src = sizes; have = sizeof(sizes); break;
    case HSO_DUMP_KFS: src = M->kfs.data(); have = sizeof(hso_kf) * nk; break;
    case HSO_DUMP_POINTS: src = M->pts.data(); have = sizeof(hso_map_point) * M->pts.size(); break;
    case HSO_DUMP_OBS: src = M->obs.data(); have = sizeof(hso_obs) * M->obs.size(); break;
    case HSO_DUMP_OBS_POINT: src = M->obs_pt.data(); have = sizeof(int32_t) * M->obs_pt.size(); break;
    case HSO_DUMP_KEY_POINTS: ints = M->keys; ints.resize(5 * nk, -1); src = ints.data(); have = sizeof(int32_t) * ints.size(); break;
    case HSO_DUMP_KF_NFTS: for (size_t k = 0; k < nk; k++) ints.push_back(k < M->kf_fts.size() ? (int32_t)M->kf_fts[k].size() : 0); src = ints.data(); have = sizeof(int32_t) * ints.size(); break;
    case HSO_DUMP_KF_FTS:
      ints.assign(nk * (size_t)M->fts_cap, 0);
      for (size_t k = 0; k < nk && k < M->kf_fts.size(); k++) std::copy(M->kf_fts[k].begin(), M->kf_fts[k].end(), ints.begin() + (std::ptrdiff_t)(k * (size_t)M->fts_cap));
      src = ints.data(); have = sizeof(int32_t) * ints.size(); break;
    case HSO_DUMP_CANDS: src = M->cands.data(); have = sizeof(int32_t) * M->cands.size(); break;
    case HSO_DUMP_FRAME_FEATS0: src = M->ff[0].data(); have = sizeof(hso_seq_feature) * M->ff[0].size(); break;
    case HSO_DUMP_FRAME_FEATS1: src = M->ff[1].data(); have = sizeof(hso_seq_feature) * M->ff[1].size(); break;
    default: return fail(c, HSO_E_INVALID, "seqmap_debug_dump: no such table");
  }
  if (have != bytes) return fail(c, HSO_E_INVALID, "seqmap_debug_dump: bytes differs from the table's size");
  if (bytes) memcpy(out, src, bytes);
  return HSO_OK;
}

void hso_gpu_debug_census(int64_t* out, int n) { for (int i = 0; i < n; i++) out[i] = 0; }   // no runtime underneath

int hso_gpu_debug_fetch(hso_gpu_ctx* c, int what, void* out, size_t bytes)
{
  const void* src = nullptr; size_t have = 0;
  switch (what) {
    case HSO_DBG_PROJ: src = c->dbg_proj.data(); have = c->dbg_proj.size() * sizeof(hso_reproj_point); break;
    case HSO_DBG_MATCH: src = c->dbg_match.data(); have = c->dbg_match.size() * sizeof(hso_align_out); break;
    case HSO_DBG_POSE_FEATS: src = c->dbg_feats.data(); have = c->dbg_feats.size() * sizeof(hso_pose_feat); break;
    case HSO_DBG_POSE_POSES: src = c->dbg_poses.data(); have = c->dbg_poses.size() * sizeof(hso_se3); break;
    case HSO_DBG_POSE_NPOSES: src = c->dbg_nposes.data(); have = c->dbg_nposes.size() * sizeof(int32_t); break;
    case HSO_DBG_SLICES: src = c->dbg_slices.data(); have = c->dbg_slices.size() * sizeof(int32_t); break;
    case HSO_DBG_EXAMINED_BEGIN: src = c->dbg_exbegin.data(); have = c->dbg_exbegin.size() * sizeof(int32_t); break;
    case HSO_DBG_BRIEF: src = c->dbg_brief.data(); have = c->dbg_brief.size() * sizeof(hso_match_brief); break;
    case HSO_DBG_PROJECTED: src = c->dbg_projected.data(); have = c->dbg_projected.size(); break;
    case HSO_DBG_POSE_MASK: src = c->dbg_mask.data(); have = c->dbg_mask.size(); break;
    default: return fail(c, HSO_E_INVALID, "debug_fetch: no such table");
  }
  if (have != bytes) return fail(c, HSO_E_INVALID, "debug_fetch: size mismatch");
  if (bytes) memcpy(out, src, bytes);
  return HSO_OK;
}

int hso_gpu_pose_optimize_batch(hso_gpu_ctx*, const hso_camera* cam, const hso_pose_job* jobs, int n, hso_pose_result* res, uint8_t* const* mask)
{
  for (int i = 0; i < n; i++) { memset(&res[i], 0, sizeof(res[i])); hso_or_pose_optimize(cam, &jobs[i], &res[i], mask ? mask[i] : nullptr); }
  return HSO_OK;
}

// ---- seeds
int hso_gpu_seed_table_create(hso_gpu_ctx* c, int* out) { c->tables.push_back(new FakeSeedTable()); *out = (int)c->tables.size() - 1; return HSO_OK; }
int hso_gpu_seed_table_destroy(hso_gpu_ctx* c, int t) { delete c->tables[t]; c->tables[t] = nullptr; return HSO_OK; }
int hso_gpu_seed_table_append(hso_gpu_ctx* c, int t, const hso_seed* s, const int32_t* group, int n, int32_t* first)
{
  FakeSeedTable* T = c->tables[t];
  if (first) *first = (int32_t)T->s.size();
  for (int i = 0; i < n; i++) {
    if (!frame_of(c, s[i].ref_frame_id)) return fail(c, HSO_E_NOFRAME, "seed_table_append: host frame not resident");
    T->s.push_back(s[i]); T->group.push_back(group ? group[i] : 0); T->alive.push_back(1);
  }
  return HSO_OK;
}
int hso_gpu_seed_table_erase(hso_gpu_ctx* c, int t, const int32_t* slots, int n)
{
  FakeSeedTable* T = c->tables[t];
  for (int i = 0; i < n; i++) { if (slots[i] < 0 || (size_t)slots[i] >= T->s.size()) return fail(c, HSO_E_INVALID, "seed_table_erase: slot out of range"); T->alive[(size_t)slots[i]] = 0; }
  return HSO_OK;
}
int hso_gpu_seed_table_size(hso_gpu_ctx* c, int t, int* n_slots, int* n_live)
{
  FakeSeedTable* T = c->tables[t];
  if (n_slots) *n_slots = (int)T->s.size();
  if (n_live) { int k = 0; for (uint8_t a : T->alive) k += a; *n_live = k; }
  return HSO_OK;
}
int hso_gpu_seed_table_compact(hso_gpu_ctx* c, int t, int32_t* remap)
{
  FakeSeedTable* T = c->tables[t];
  FakeSeedTable N;
  for (size_t i = 0; i < T->s.size(); i++) {
    if (remap) remap[i] = T->alive[i] ? (int32_t)N.s.size() : -1;
    if (T->alive[i]) { N.s.push_back(T->s[i]); N.group.push_back(T->group[i]); N.alive.push_back(1); }
  }
  *T = N;
  return (int)T->s.size();
}
int hso_gpu_seed_table_read(hso_gpu_ctx* c, int t, int first, int n, hso_seed* out)
{
  FakeSeedTable* T = c->tables[t];
  for (int i = 0; i < n; i++) out[i] = T->s.at((size_t)(first + i));
  return HSO_OK;
}
int hso_gpu_seed_table_set_host_pose(hso_gpu_ctx* c, int t, const int64_t* ids, const hso_se3* T_f_w, int n)
{
  FakeSeedTable* T = c->tables[t];
  for (size_t i = 0; i < T->s.size(); i++) if (T->alive[i]) for (int k = 0; k < n; k++) if (T->s[i].ref_frame_id == ids[k]) T->s[i].T_ref_w = T_f_w[k];
  return HSO_OK;
}
int hso_gpu_seed_table_observe_groups(hso_gpu_ctx* c, const hso_camera* cam, int t, const hso_seed_frame* frames, int n_frames, double px_error_angle,
                                      hso_seed_brief* brief, float* px, hso_seed_out* full)
{
  FakeSeedTable* T = c->tables[t];
  for (size_t i = 0; i < T->s.size(); i++) {
    if (brief) memset(&brief[i], 0, sizeof(hso_seed_brief));
    if (px) { px[2 * i] = 0; px[2 * i + 1] = 0; }
    if (full) memset(&full[i], 0, sizeof(hso_seed_out));
    if (!T->alive[i]) continue;
    const int g = T->group[i];
    if (g >= n_frames) return fail(c, HSO_E_INVALID, "seed_table_observe: a seed's group has no frame");
    if (frames[g].frame_id < 0) continue;
    FakeFrame* C = frame_of(c, frames[g].frame_id); FakeFrame* R = frame_of(c, T->s[i].ref_frame_id);
    if (!C || !R) return fail(c, HSO_E_NOFRAME, "seed_table_observe: frame not resident");
    hso_seed_out o;
    memset(&o, 0, sizeof(o));
    hso_or_seed_observe(cam, &T->s[i], &frames[g].T_f_w, frames[g].exposure_time, px_error_angle, R->pyr, C->pyr, C->sx, C->sy, C->w, C->h, &o);
    T->s[i].mu = o.mu; T->s[i].sigma2 = o.sigma2; T->s[i].b = o.b;
    if (brief) { brief[i].mu = o.mu; brief[i].sigma2 = o.sigma2; brief[i].b = o.b; brief[i].result = (int8_t)o.result; brief[i].is_update = (int8_t)o.is_update;
                 brief[i].is_valid = (int8_t)o.is_valid; brief[i].search_level = (int8_t)o.search_level; }
    if (px) { px[2 * i] = (float)o.px_cur[0]; px[2 * i + 1] = (float)o.px_cur[1]; }
    if (full) full[i] = o;
  }
  return HSO_OK;
}

int hso_gpu_seed_table_observe_previous(hso_gpu_ctx* c, const hso_camera* cam, int t, const int64_t* host_ids, const hso_seed_frame* pre, int n,
                                        double px_error_angle, hso_seed_brief* brief, hso_seed_out* full)
{
  FakeSeedTable* T = c->tables[t];
  for (size_t i = 0; i < T->s.size(); i++) {
    if (brief) memset(&brief[i], 0, sizeof(hso_seed_brief));
    if (full) memset(&full[i], 0, sizeof(hso_seed_out));
    if (!T->alive[i]) continue;
    int k = 0;
    while (k < n && host_ids[k] != T->s[i].ref_frame_id) k++;
    if (k == n) continue;
    FakeFrame* C = frame_of(c, pre[k].frame_id); FakeFrame* R = frame_of(c, T->s[i].ref_frame_id);
    if (!C || !R) return fail(c, HSO_E_NOFRAME, "seed_table_observe_previous: frame not resident");
    hso_seed_out o;
    memset(&o, 0, sizeof(o));
    hso_or_seed_observe_previous(cam, &T->s[i], &pre[k].T_f_w, pre[k].exposure_time, px_error_angle, R->pyr, C->pyr, C->sx, C->sy, C->w, C->h, &o);
    T->s[i].mu = o.mu; T->s[i].sigma2 = o.sigma2;
    if (brief) { brief[i].mu = o.mu; brief[i].sigma2 = o.sigma2; brief[i].b = o.b; brief[i].result = (int8_t)o.result; brief[i].is_update = (int8_t)o.is_update;
                 brief[i].is_valid = (int8_t)o.is_valid; brief[i].search_level = (int8_t)o.search_level; }
    if (full) full[i] = o;
  }
  return HSO_OK;
}

// the overlapped form: the fake has nothing to overlap, _begin runs the pass and keeps the briefs for _end
int hso_gpu_seed_table_observe_previous_begin(hso_gpu_ctx* c, const hso_camera* cam, int t, const int64_t* host_ids, const hso_seed_frame* pre, int n,
                                              double px_error_angle)
{
  FakeSeedTable* T = c->tables[t];
  c->prev_briefs.assign(T->s.size() + 1, hso_seed_brief{});
  c->prev_table = t;
  return hso_gpu_seed_table_observe_previous(c, cam, t, host_ids, pre, n, px_error_angle, c->prev_briefs.data(), nullptr);
}
int hso_gpu_seed_table_observe_previous_end(hso_gpu_ctx* c, int t, hso_seed_brief* brief, int n_brief)
{
  if (c->prev_table != t) return fail(c, HSO_E_INVALID, "seed_table_observe_previous_end: no pass in flight for this table");
  const size_t n = c->prev_briefs.size() - 1;
  if (brief) { if ((size_t)n_brief < n) return fail(c, HSO_E_INVALID, "seed_table_observe_previous_end: brief_out too small"); memcpy(brief, c->prev_briefs.data(), n * sizeof(hso_seed_brief)); }
  c->prev_table = -1;
  return HSO_OK;
}

static int activate_one(hso_gpu_ctx* c, const hso_camera* cam, const hso_seed* s, const hso_activate_target* tg, int n_tg, int n_mean, hso_activate_out* o,
                        hso_align_out* mo)
{
  FakeFrame* R = frame_of(c, s->ref_frame_id);
  if (!R) return fail(c, HSO_E_NOFRAME, "seed_activate: host frame not resident");
  std::vector<const uint8_t*> pyr; std::vector<const int16_t*> gx, gy;
  for (int k = 0; k < n_tg; k++) {
    FakeFrame* F = frame_of(c, tg[k].frame_id);
    if (!F) return fail(c, HSO_E_NOFRAME, "seed_activate: target frame not resident");
    for (int L = 0; L < HSO_N_PYR_LEVELS; L++) pyr.push_back(F->pyr[L]);
    for (int L = 0; L < HSO_N_SOBEL_LEVELS; L++) { gx.push_back(F->sx[L]); gy.push_back(F->sy[L]); }
  }
  const uint8_t* none_p = nullptr; const int16_t* none_g = nullptr;
  std::vector<hso_align_out> scratch((size_t)std::max(n_tg, 1));
  memset(o, 0, sizeof(*o));
  hso_or_seed_activate(cam, s, tg, n_tg, R->pyr, pyr.empty() ? &none_p : pyr.data(), gx.empty() ? &none_g : gx.data(), gy.empty() ? &none_g : gy.data(), R->w, R->h,
                       n_mean, o, mo ? mo : scratch.data());
  return HSO_OK;
}

int hso_gpu_seed_activate_multi(hso_gpu_ctx* c, const hso_camera* cam, const hso_seed* seeds, int n, const int32_t* begin, const hso_activate_target* targets,
                                const int32_t* n_mean, hso_activate_out* out, hso_align_out* match_out)
{
  for (int i = 0; i < n; i++)
    if (int rc = activate_one(c, cam, &seeds[i], targets + begin[i], begin[i + 1] - begin[i], n_mean[i], &out[i], match_out ? match_out + begin[i] : nullptr)) return rc;
  return HSO_OK;
}

int hso_gpu_seed_activate_frames(hso_gpu_ctx* c, const hso_camera* cam, const hso_seed* seeds, int n, const int32_t* begin, const int32_t* target_frame,
                                 const hso_activate_target* frames, int n_frames, const int32_t* n_mean, hso_activate_out* out)
{
  std::vector<hso_activate_target> t;
  for (int i = 0; i < n; i++) {
    t.clear();
    for (int k = begin[i]; k < begin[i + 1]; k++) { if (target_frame[k] < 0 || target_frame[k] >= n_frames) return fail(c, HSO_E_INVALID, "seed_activate_frames: index out of range"); t.push_back(frames[target_frame[k]]); }
    hso_activate_target none{};
    if (int rc = activate_one(c, cam, &seeds[i], t.empty() ? &none : t.data(), (int)t.size(), n_mean[i], &out[i], nullptr)) return rc;
  }
  return HSO_OK;
}

int hso_gpu_seed_table_activate(hso_gpu_ctx* c, const hso_camera* cam, int t, const int32_t* slots, int n, const int32_t* begin, const int32_t* target_frame,
                                const hso_activate_target* frames, int n_frames, const int32_t* n_mean, hso_activate_out* out)
{
  FakeSeedTable* T = c->tables.at((size_t)t);
  std::vector<hso_seed> seeds((size_t)n);
  for (int i = 0; i < n; i++) {
    if (slots[i] < 0 || (size_t)slots[i] >= T->s.size() || !T->alive[(size_t)slots[i]]) return fail(c, HSO_E_INVALID, "seed_table_activate: slot out of range or erased");
    seeds[(size_t)i] = T->s[(size_t)slots[i]];
  }
  return hso_gpu_seed_activate_frames(c, cam, seeds.data(), n, begin, target_frame, frames, n_frames, n_mean, out);
}

int hso_gpu_seed_reproject_match(hso_gpu_ctx* c, const hso_camera* cam, int64_t cur_id, const hso_se3* T_cur_w, double cur_exposure, const hso_seed* seeds, int n,
                                 int cell_size, int grid_n_cols, hso_reproj_point* proj, hso_align_out* match)
{
  for (int i = 0; i < n; i++) {
    memset(&proj[i], 0, sizeof(proj[i])); proj[i].ref_obs = -1;
    memset(&match[i], 0, sizeof(match[i]));
    int cell = 0;
    if (!hso_or_reproject_point(cam, T_cur_w, &seeds[i].T_ref_w, seeds[i].f, (double)seeds[i].mu, cell_size, grid_n_cols, proj[i].px, &cell)) continue;
    // reprojectorSeed rejects z < 0.001 where reprojectPoint rejects z < 0.00001: the activation matcher's own projection test decides
    hso_activate_target t{};
    t.frame_id = cur_id; t.T_f_w = *T_cur_w; t.exposure = cur_exposure;
    hso_activate_out o; hso_align_out mo;
    memset(&mo, 0, sizeof(mo));
    if (int rc = activate_one(c, cam, &seeds[i], &t, 1, 6, &o, &mo)) return rc;
    if (!o.n_targets) continue;
    proj[i].projected = 1; proj[i].cell = cell;
    match[i] = mo;
  }
  return HSO_OK;
}

// ---- local BA
int hso_gpu_ba_huber_deltas(hso_gpu_ctx*, const hso_se3* poses, int n_poses, const double* idist, int n_points, const hso_ba_edge* edges, const double* obs_uv,
                            int n_edges, double em2, float* hc, float* he)
{
  hso_or_ba_huber_deltas(poses, n_poses, idist, n_points, edges, obs_uv, n_edges, em2, hc, he);
  return HSO_OK;
}
int hso_gpu_ba_huber_deltas_multi(hso_gpu_ctx*, hso_ba_deltas_job* j, int n, double em2)
{
  for (int i = 0; i < n; i++) {
    j[i].huber_corner = 0; j[i].huber_edge = 0;
    if (j[i].n_edges > 0) hso_or_ba_huber_deltas(j[i].poses_f_w, j[i].n_poses, j[i].idist, j[i].n_points, j[i].edges, j[i].obs_uv, j[i].n_edges, em2, &j[i].huber_corner, &j[i].huber_edge);
  }
  return HSO_OK;
}
int hso_gpu_ba_optimize_multi(hso_gpu_ctx*, const hso_ba_problem* p, int n)
{
  for (int i = 0; i < n; i++)
    hso_or_ba_optimize(p[i].poses_f_w, p[i].pose_fixed, p[i].n_poses, p[i].idist, p[i].n_points, p[i].edges, p[i].n_edges, p[i].huber_corner, p[i].huber_edge, p[i].n_iter,
                       p[i].edge_chi2_out, p[i].result);
  return HSO_OK;
}

// the two stages in one call: the restatement's deltas, then its optimisation with them
int hso_gpu_ba_local_multi(hso_gpu_ctx*, const hso_ba_problem* p, const double* const* obs_uv, int n, double em2, float* huber_out)
{
  for (int i = 0; i < n; i++) {
    float hc = 0, he = 0;
    hso_or_ba_huber_deltas(p[i].poses_f_w, p[i].n_poses, p[i].idist, p[i].n_points, p[i].edges, obs_uv[i], p[i].n_edges, em2, &hc, &he);
    huber_out[2 * i] = hc; huber_out[2 * i + 1] = he;
    hso_or_ba_optimize(p[i].poses_f_w, p[i].pose_fixed, p[i].n_poses, p[i].idist, p[i].n_points, p[i].edges, p[i].n_edges, (double)hc, (double)he, p[i].n_iter,
                       p[i].edge_chi2_out, p[i].result);
  }
  return HSO_OK;
}

// ---- ba::LocalBundleAdjustment on a sequence map (src/bundle_adjustment.cpp:577-892): the graph from the map's tables in the
// reference's order of construction, the restatement's deltas and optimisation, the write-back, the culling's inputs
int hso_gpu_seq_local_ba(hso_gpu_ctx* c, const hso_seq_ba_job* jobs, int n, double em2, double chi2_corner, double chi2_edgelet, hso_seq_ba_result* res)
{
  if (!c) return HSO_E_INVALID;
  if (n < 0 || (n > 0 && (!jobs || !res))) return fail(c, HSO_E_INVALID, "seq_local_ba: bad argument");
  for (int j = 0; j < n; j++) {
    const hso_seq_ba_job& J = jobs[j];
    if (J.map < 0 || (size_t)J.map >= c->maps.size() || !c->maps[J.map]) return fail(c, HSO_E_INVALID, "seq_local_ba: no such map");
    for (int i = 0; i < j; i++) if (jobs[i].map == J.map) return fail(c, HSO_E_INVALID, "seq_local_ba: a map appears twice");
    if (J.n_core < 1 || J.n_core > HSO_SEQ_BA_MAX_CORE || J.n_iter < 0 || J.point_cap < 0 || J.cull_cap < 0 || (J.point_cap > 0 && (!J.point_ids || !J.point_state)) || (J.cull_cap > 0 && !J.culled))
      return fail(c, HSO_E_INVALID, "seq_local_ba: bad argument");
    const FakeMap* M = c->maps[J.map];
    for (int i = 0; i < J.n_core; i++) {
      if (J.core[i] < 0 || (size_t)J.core[i] >= M->kfs.size()) return fail(c, HSO_E_INVALID, "seq_local_ba: no such keyframe row");
      for (int k = 0; k < i; k++) if (J.core[k] == J.core[i]) return fail(c, HSO_E_INVALID, "seq_local_ba: a core keyframe appears twice");
    }
  }
  c->ba_windows.assign((size_t)n, hso_gpu_ctx::BaWindow());
  for (int j = 0; j < n; j++) {
    const hso_seq_ba_job& J = jobs[j];
    FakeMap* M = c->maps[J.map];
    hso_seq_ba_result& R = res[j];
    memset(&R, 0, sizeof(R));
    hso_gpu_ctx::BaWindow& W = c->ba_windows[(size_t)j];
    std::vector<int> vertex(M->kfs.size(), -1);
    std::vector<uint8_t> in_window(M->pts.size(), 0);
    for (int i = 0; i < J.n_core; i++) {                            // :592-616
      vertex[(size_t)J.core[i]] = (int)W.rows.size();
      W.rows.push_back(J.core[i]); W.fixed.push_back(J.fixed[i] ? 1 : 0);
      if ((size_t)J.core[i] < M->kf_fts.size())
        for (int32_t f : M->kf_fts[(size_t)J.core[i]]) { const int32_t p = M->obs_pt.at((size_t)f); if (p >= 0) in_window.at((size_t)p) = 1; }
    }
    std::vector<int32_t> pts;
    for (size_t p = 0; p < in_window.size(); p++) if (in_window[p]) pts.push_back((int32_t)p);
    auto vertex_of = [&](int row) {
      if (vertex[(size_t)row] < 0) { vertex[(size_t)row] = (int)W.rows.size(); W.rows.push_back(row); W.fixed.push_back(1); }   // :700-737
      return vertex[(size_t)row];
    };
    std::vector<double> idist(pts.size());
    for (size_t i = 0; i < pts.size(); i++) {                       // :690-812
      const hso_map_point& P = M->pts[(size_t)pts[i]];
      idist[i] = P.idist;
      const int vh = vertex_of(P.host_kf);
      for (int q = 0, row = P.obs_begin; q < P.obs_count; q++, row = M->obs[(size_t)row].pad_) {
        const hso_obs& ob = M->obs.at((size_t)row);
        if (ob.kf == P.host_kf) continue;
        hso_ba_edge e{};
        e.point = (int)i; e.host = vh; e.target = vertex_of(ob.kf);
        e.type = ob.type == HSO_FTR_EDGELET ? HSO_FTR_EDGELET : HSO_FTR_CORNER;
        e.level = ob.level;
        e.fH[0] = P.host_f[0]; e.fH[1] = P.host_f[1]; e.fH[2] = P.host_f[2];
        const double u = ob.f[0] / ob.f[2], v = ob.f[1] / ob.f[2];
        if (e.type == HSO_FTR_EDGELET) { e.normal[0] = ob.grad[0]; e.normal[1] = ob.grad[1]; e.meas[0] = ob.grad[0] * u + ob.grad[1] * v; }
        else { e.normal[0] = 1; e.normal[1] = 0; e.meas[0] = u; e.meas[1] = v; }
        W.edges.push_back(e); W.edge_obs.push_back(row);
        W.uv.push_back(u); W.uv.push_back(v);
      }
    }
    W.n_points = (int32_t)pts.size();
    R.n_poses = (int32_t)W.rows.size(); R.n_points = (int32_t)pts.size(); R.n_edges = (int32_t)W.edges.size();
    if ((int)pts.size() > J.point_cap) return fail(c, HSO_E_INVALID, "seq_local_ba: point_cap is smaller than the window");
    for (size_t i = 0; i < pts.size(); i++) J.point_ids[i] = pts[i];
    W.poses_in.resize(W.rows.size());
    for (size_t v = 0; v < W.rows.size(); v++) W.poses_in[v] = M->kfs[(size_t)W.rows[v]].T_f_w;
    W.poses_out = W.poses_in; W.idist_in = idist;
    for (int i = 0; i < J.n_core; i++) R.core_pose[i] = W.poses_in[(size_t)i];
    W.chi2.assign(W.edges.size(), 0.0);
    if (W.edges.empty() || pts.empty()) {
      R.status = 1; W.status = 1;
      for (size_t i = 0; i < pts.size(); i++) { const hso_map_point& P = M->pts[(size_t)pts[i]]; double* o = J.point_state + 4 * i; o[0] = P.idist; o[1] = P.pos[0]; o[2] = P.pos[1]; o[3] = P.pos[2]; }
      continue;
    }
    float hc = 0, he = 0;                                           // :618-680, :815-823
    hso_or_ba_huber_deltas(W.poses_out.data(), (int)W.rows.size(), idist.data(), (int)pts.size(), W.edges.data(), W.uv.data(), (int)W.edges.size(), em2, &hc, &he);
    R.huber_corner = hc; R.huber_edge = he;
    hso_or_ba_optimize(W.poses_out.data(), W.fixed.data(), (int)W.rows.size(), idist.data(), (int)pts.size(), W.edges.data(), (int)W.edges.size(), (double)hc, (double)he, J.n_iter,
                       W.chi2.data(), &R.lm);
    // :826-853 (the poses of the core; pos_ of every point of the window from its host keyframe's new pose)
    for (int i = 0; i < J.n_core; i++) { M->kfs[(size_t)J.core[i]].T_f_w = W.poses_out[(size_t)i]; R.core_pose[i] = W.poses_out[(size_t)i]; }
    for (size_t i = 0; i < pts.size(); i++) {
      hso_map_point& P = M->pts[(size_t)pts[i]];
      P.idist = idist[i];
      hso_se3 inv;
      hso_or_se3_inverse(&M->kfs[(size_t)P.host_kf].T_f_w, &inv);
      const double s = 1.0 / P.idist;
      const double in_host[3] = {P.host_f[0] * s, P.host_f[1] * s, P.host_f[2] * s};
      hso_or_se3_apply(&inv, in_host, P.pos);
      double* o = J.point_state + 4 * i; o[0] = P.idist; o[1] = P.pos[0]; o[2] = P.pos[1]; o[3] = P.pos[2];
    }
    // :855-892: corner edges first, then edgelet edges
    int nc = 0;
    for (int pass = 0; pass < 2; pass++)
      for (size_t e = 0; e < W.edges.size(); e++) {
        const bool edgelet = W.edges[e].type == HSO_FTR_EDGELET;
        if (edgelet != (pass == 1)) continue;
        if (!(W.chi2[e] > (edgelet ? chi2_edgelet : chi2_corner))) continue;
        if (nc < J.cull_cap) J.culled[nc] = W.edge_obs[e];
        nc++; R.n_culled[pass]++;
      }
  }
  return HSO_OK;
}

int hso_gpu_seq_ba_debug_window(hso_gpu_ctx* c, int job, int what, void* out, size_t bytes)
{
  if (!c || job < 0 || (size_t)job >= c->ba_windows.size() || !out) return fail(c, HSO_E_INVALID, "seq_ba_debug_window: bad argument");
  const hso_gpu_ctx::BaWindow& W = c->ba_windows[(size_t)job];
  const int32_t sizes[4] = {(int32_t)W.rows.size(), W.n_points, (int32_t)W.edges.size(), W.status};
  const void* src = nullptr; size_t have = 0;
  switch (what) {
    case HSO_BAW_SIZES: src = sizes; have = sizeof(sizes); break;
    case HSO_BAW_VERTEX_ROWS: src = W.rows.data(); have = sizeof(int32_t) * W.rows.size(); break;
    case HSO_BAW_FIXED: src = W.fixed.data(); have = W.fixed.size(); break;
    case HSO_BAW_EDGES: src = W.edges.data(); have = sizeof(hso_ba_edge) * W.edges.size(); break;
    case HSO_BAW_OBS_UV: src = W.uv.data(); have = sizeof(double) * W.uv.size(); break;
    case HSO_BAW_EDGE_OBS: src = W.edge_obs.data(); have = sizeof(int32_t) * W.edge_obs.size(); break;
    case HSO_BAW_EDGE_CHI2: src = W.chi2.data(); have = sizeof(double) * W.chi2.size(); break;
    case HSO_BAW_POSES_OUT: src = W.poses_out.data(); have = sizeof(hso_se3) * W.poses_out.size(); break;
    case HSO_BAW_POSES_IN: src = W.poses_in.data(); have = sizeof(hso_se3) * W.poses_in.size(); break;
    case HSO_BAW_IDIST_IN: src = W.idist_in.data(); have = sizeof(double) * W.idist_in.size(); break;
    default: return fail(c, HSO_E_INVALID, "seq_ba_debug_window: no such table");
  }
  if (have != bytes) return fail(c, HSO_E_INVALID, "seq_ba_debug_window: bytes differs from the table's size");
  if (bytes) memcpy(out, src, bytes);
  return HSO_OK;
}

// ---- detection
static int detect_impl(hso_gpu_ctx* c, const int64_t* ids, int n_frames, int n_levels, int min_thresh, hso_corner* corners, int corner_cap, int32_t* nc,
                       hso_edgelet* ed, hso_corner* fill, int second_cap, int32_t* ns, bool init)
{
  for (int f = 0; f < n_frames; f++) {
    FakeFrame* F = frame_of(c, ids[f]);
    if (!F) return fail(c, HSO_E_NOFRAME, "detect: frame not resident");
    for (int L = 0; L < n_levels; L++) {
      std::vector<hso_corner> co(65536);
      const int n = hso_or_fast_detect_level(F->lev[L].data(), F->lw[L], F->lh[L], min_thresh, 8, co.data(), (int)co.size());
      nc[f * n_levels + L] = n;
      for (int i = 0; i < n && i < corner_cap; i++) corners[((size_t)f * n_levels + L) * corner_cap + i] = co[(size_t)i];
      int grid, gc, gr, lw, lh;
      hso_or_detect_grid(F->w, F->h, L, &grid, &gc, &gr, &lw, &lh);
      std::vector<uint8_t> have((size_t)gc * gr, 0);
      for (int i = 0; i < n; i++) have[(size_t)hso_or_detect_cell_index(co[(size_t)i].x, co[(size_t)i].y, grid, gc, gr)] = 1;
      if (init) {
        if (L == 0) ns[f] = hso_or_filling_hole_level(F->lev[0].data(), F->lw[0], F->lh[0], 0, F->w, F->h, min_thresh, have.data(), fill + (size_t)f * second_cap, second_cap);
      } else {
        ns[f * n_levels + L] = hso_or_edgelet_level(F->gx[L].data(), F->gy[L].data(), F->lw[L], F->lh[L], L, F->w, F->h, min_thresh, have.data(),
                                                    ed + ((size_t)f * n_levels + L) * second_cap, second_cap);
      }
    }
  }
  return HSO_OK;
}
int hso_gpu_detect_candidates(hso_gpu_ctx* c, const int64_t* ids, int n, int n_levels, int min_thresh, hso_corner* co, int cap, int32_t* nc, hso_edgelet* ed, int ecap,
                              int32_t* ne)
{
  return detect_impl(c, ids, n, n_levels, min_thresh, co, cap, nc, ed, nullptr, ecap, ne, false);
}
int hso_gpu_detect_candidates_multi(hso_gpu_ctx* c, const int64_t* ids, int n, int n_levels, const int32_t* min_thresh, hso_corner* co, int cap, int32_t* nc, hso_edgelet* ed,
                                    int ecap, int32_t* ne)
{
  for (int f = 0; f < n; f++) {   // a frame at a time, each at its own barrier
    const int rc = detect_impl(c, ids + f, 1, n_levels, min_thresh[f], co ? co + (size_t)f * n_levels * cap : nullptr, cap, nc + (size_t)f * n_levels,
                               ed ? ed + (size_t)f * n_levels * ecap : nullptr, nullptr, ecap, ne + (size_t)f * n_levels, false);
    if (rc != HSO_OK) return rc;
  }
  return HSO_OK;
}
int hso_gpu_detect_candidates_init(hso_gpu_ctx* c, const int64_t* ids, int n, int n_levels, int min_thresh, hso_corner* co, int cap, int32_t* nc, hso_corner* fill,
                                   int fcap, int32_t* nf)
{
  return detect_impl(c, ids, n, n_levels, min_thresh, co, cap, nc, nullptr, fill, fcap, nf, true);
}

int hso_gpu_klt_track(hso_gpu_ctx* c, int64_t prev, int64_t cur, const float* px_prev, const float* px_init, int n, const hso_klt_params* p, hso_klt_result* out)
{
  FakeFrame* P = frame_of(c, prev); FakeFrame* C = frame_of(c, cur);
  if (!P || !C) return fail(c, HSO_E_NOFRAME, "klt_track: frame not resident");
  std::vector<float> b(px_init, px_init + 2 * (size_t)n), mg((size_t)n);
  std::vector<uint8_t> st((size_t)n);
  hso_or_klt_track(P->lev[0].data(), C->lev[0].data(), P->w, P->h, px_prev, b.data(), st.data(), n, p->win_size, p->max_level, p->max_iter, p->epsilon,
                   p->use_initial_flow, mg.data());
  for (int i = 0; i < n; i++) {
    out[i].px[0] = b[2 * (size_t)i]; out[i].px[1] = b[2 * (size_t)i + 1];
    out[i].status = st[(size_t)i] ? HSO_KLT_TRACKED : 0;
    float ncc = -2;
    if (st[(size_t)i] && hso_or_patch_check(P->lev[0].data(), C->lev[0].data(), P->w, P->h, px_prev + 2 * (size_t)i, out[i].px, &ncc)) out[i].status |= HSO_KLT_PATCH_OK;
    out[i].ncc = ncc;
  }
  return HSO_OK;
}

}  // extern "C"
