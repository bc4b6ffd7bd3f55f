// hso_align.hip — batched reprojection matching on gfx950: affine warp of the reference
// patch, 8x8 inverse-compositional Lucas-Kanade (2-D, or 1-D along an edgelet direction),
// NCC and edgelet-normal gates.
//
// Replaces Matcher::findMatchDirect after the reference observation has been chosen
// (reference src/matcher.cpp:286-375) and what it calls: warp::getWarpMatrixAffine :46-72,
// getBestSearchLevel :74-85, warpAffine(float) :120-155, createPatchFromPatchWithBorder
// :226-238, feature_alignment::align2D(float) src/feature_alignment.cpp:464-605,
// align1D(float) :164-308, Matcher::checkNCC :379-404, checkNormal :406-440,
// hso::interpolateMat_8u include/hso/vikit/vision.h:49-65.
//
// MI355X mapping: a DPP row of 16 lanes per candidate, four candidates per wavefront, four pixels of the 8x8 patch per lane
// (hso_match_dev.h).  The per-candidate geometry (fp64 warp matrix, search level) is computed one LANE per candidate for the
// wave's whole group first; the 10x10 warped patch lives in LDS; every LK iteration is two 8-byte loads per lane + row sums
// (three adds + four DPP steps; all lanes of a row end with identical bits, so the iteration state stays uniform per row and
// rows diverge freely).  The reference accumulates the same sums serially in fp32; the tree order differs from it by rounding
// only (stated tolerance: 1e-3 px on the result).  Instruction-issue-bound (profiles/r3_stage_sq_align.csv:
// VALU 100 % busy with one candidate per wave), hence the packing.
#include "hso_match_dev.h"
#include <stddef.h>
#include <algorithm>
#include <string.h>
#include <vector>

using namespace hso_dev;

#define ALIGN_WAVES_PER_BLOCK 4

struct AlignConsts {
  hso_camera cam;
  PyrGeom g;
};

struct AlignJobDev {
  const uint8_t* ref_base;
  const uint8_t* cur_base;   // the frame this candidate is searched in (jobs of many frames share a launch)
  hso_align_job j;
};

// One wavefront matches `cpw` consecutive candidates (a power of two <= 64, chosen by the launch: 1 when the batch is small
// enough to spread one candidate per wave over the chip, up to 64 for multi-sequence batches).  Phase 1: one LANE per candidate
// computes the candidate's geometry (match_geometry: the wave-uniform fp64 half of findMatchDirect) into LDS — 64 candidates
// for the instruction issue of one; phase 2: the whole wave walks the candidates (lane = patch pixel).
// SPARSE: candidates with a null reference are skipped (the output array was zeroed): the chained projection + matching calls.
template <bool SPARSE>
__global__ __launch_bounds__(64 * ALIGN_WAVES_PER_BLOCK) void k_align_t(AlignConsts C, const AlignJobDev* jobs, int n_jobs,
                                                                         hso_align_out* outs, int cpw)
{
  __shared__ float s_pwb[ALIGN_WAVES_PER_BLOCK][4][100];
  __shared__ MatchGeom s_geom[ALIGN_WAVES_PER_BLOCK][64];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int first = (blockIdx.x * ALIGN_WAVES_PER_BLOCK + wave) * cpw;
  if (first >= n_jobs) return;
  if (lane < cpw && first + lane < n_jobs) {
    const AlignJobDev& JD = jobs[first + lane];
    if (!SPARSE || JD.ref_base != nullptr) s_geom[wave][lane] = match_geometry(C.cam, C.g, JD.j);
  }
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  // phase 2: row r of the wavefront (16 lanes) walks candidates r, r + 4, r + 8, ... of the wave's group on its own
  const int row = lane >> 4;
  for (int q = row; q < cpw; q += 4) {
    const int jid = first + q;
    if (jid >= n_jobs) break;
    const AlignJobDev& JD = jobs[jid];
    if (SPARSE && JD.ref_base == nullptr) continue;    // outs was zeroed
    const hso_align_out o = match_patch(C.g, JD.cur_base, JD.ref_base, JD.j, s_geom[wave][q], (double)0.7f, s_pwb[wave][row]);  // checkNCC(…, 0.7), :364
    if ((lane & 15) == 0) outs[jid] = o;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");   // the row's next candidate overwrites its patch
  }
}

// candidates per wave for a batch of n: four (one per 16-lane row) until the chip holds ~8 waves per SIMD of them, then doubling
static int align_cpw(const hso_gpu_ctx* ctx, int n)
{
  const long long spread = (long long)ctx->n_cu * 4 * 8;
  int cpw = 4;
  while (cpw < 64 && (long long)n > spread * cpw) cpw *= 2;
  return cpw;
}
static void launch_align(hso_gpu_ctx* ctx, bool sparse, const AlignConsts& C, const AlignJobDev* d_jobs, int n, hso_align_out* d_out)
{
  const int cpw = align_cpw(ctx, n);
  const int waves = (n + cpw - 1) / cpw, blocks = (waves + ALIGN_WAVES_PER_BLOCK - 1) / ALIGN_WAVES_PER_BLOCK;
  if (sparse) hipLaunchKernelGGL(k_align_t<true>, dim3(blocks), dim3(64 * ALIGN_WAVES_PER_BLOCK), 0, ctx->stream, C, d_jobs, n, d_out, cpw);
  else hipLaunchKernelGGL(k_align_t<false>, dim3(blocks), dim3(64 * ALIGN_WAVES_PER_BLOCK), 0, ctx->stream, C, d_jobs, n, d_out, cpw);
}

// cur_frame_ids: one id per job (stride 1) or one id for all jobs (stride 0)
static int align_run(hso_gpu_ctx* ctx, const hso_camera* cam, const int64_t* cur_frame_ids, int id_stride, const hso_align_job* jobs,
                     int n_jobs, hso_align_out* out)
{
  if (!ctx) return HSO_E_INVALID;
  if (!cam || !cur_frame_ids || n_jobs < 0 || (n_jobs > 0 && (!jobs || !out))) return hso_fail(ctx, HSO_E_INVALID, "align_batch: bad argument");
  if (n_jobs == 0) return HSO_OK;
  HSO_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  PyrGeom g{};
  AlignJobDev* h = reinterpret_cast<AlignJobDev*>(hso_pinned(ctx, 0, (size_t)n_jobs * sizeof(AlignJobDev)));
  hso_align_out* h_out = reinterpret_cast<hso_align_out*>(hso_pinned(ctx, 1, (size_t)n_jobs * sizeof(hso_align_out)));
  if (!h || !h_out) return HSO_E_NOMEM;
  int64_t last_id = 0, last_ref_id = 0;
  const uint8_t* last_base = nullptr;
  const uint8_t* last_ref_base = nullptr;
  for (int i = 0; i < n_jobs; i++) {
    const int64_t cid = cur_frame_ids[(size_t)i * id_stride];
    if (i == 0 || cid != last_id) {
      auto itc = ctx->frames.find(cid);
      if (itc == ctx->frames.end()) return hso_fail(ctx, HSO_E_NOFRAME, "align_batch: current frame not resident");
      if (i == 0) g = itc->second.g;
      else if (!same_geom(itc->second.g, g)) return hso_fail(ctx, HSO_E_INVALID, "align_batch: frames must share one size");
      last_id = cid; last_base = itc->second.base;
    }
    if (i == 0 || jobs[i].ref_frame_id != last_ref_id) {   // candidates come grouped by reference keyframe: one lookup per group
      auto itr = ctx->frames.find(jobs[i].ref_frame_id);
      if (itr == ctx->frames.end()) return hso_fail(ctx, HSO_E_NOFRAME, "align_batch: reference frame not resident");
      if (!same_geom(itr->second.g, g)) return hso_fail(ctx, HSO_E_INVALID, "align_batch: frames must share one size");
      last_ref_id = jobs[i].ref_frame_id; last_ref_base = itr->second.base;
    }
    if (jobs[i].ref_level < 0 || jobs[i].ref_level >= HSO_N_PYR_LEVELS) return hso_fail(ctx, HSO_E_INVALID, "align_batch: bad ref_level");
    h[i].ref_base = last_ref_base;
    h[i].cur_base = last_base;
    h[i].j = jobs[i];
  }
  if (cam->width != g.w[0] || cam->height != g.h[0]) return hso_fail(ctx, HSO_E_INVALID, "align_batch: camera size differs from the frame size");
  const size_t b_jobs = ((size_t)n_jobs * sizeof(AlignJobDev) + 255) & ~size_t(255);
  const size_t need = b_jobs + (size_t)n_jobs * sizeof(hso_align_out);
  if (ctx->batch_cap < need) {
    HSO_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->d_batch) (void)hipFree(ctx->d_batch);
    ctx->d_batch = nullptr; ctx->batch_cap = 0;
    HSO_HIP_CHECK(ctx, hipMalloc(reinterpret_cast<void**>(&ctx->d_batch), hso_grown(need)));
    ctx->batch_cap = hso_grown(need);
  }
  AlignJobDev* d_jobs = reinterpret_cast<AlignJobDev*>(ctx->d_batch);
  hso_align_out* d_out = reinterpret_cast<hso_align_out*>(ctx->d_batch + b_jobs);
  HSO_HIP_CHECK(ctx, hipMemcpyAsync(d_jobs, h, (size_t)n_jobs * sizeof(AlignJobDev), hipMemcpyHostToDevice, ctx->stream));
  AlignConsts C;
  C.cam = *cam; C.g = g;
  launch_align(ctx, false, C, d_jobs, n_jobs, d_out);
  HSO_HIP_CHECK(ctx, hipGetLastError());
  HSO_HIP_CHECK(ctx, hipMemcpyAsync(h_out, d_out, (size_t)n_jobs * sizeof(hso_align_out), hipMemcpyDeviceToHost, ctx->stream));
  HSO_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  memcpy(out, h_out, (size_t)n_jobs * sizeof(hso_align_out));
  return HSO_OK;
}

extern "C" int hso_gpu_align_batch(hso_gpu_ctx* ctx, const hso_camera* cam, int64_t cur_frame_id, const hso_align_job* jobs,
                                   int n_jobs, hso_align_out* out)
{
  return align_run(ctx, cam, &cur_frame_id, 0, jobs, n_jobs, out);
}

extern "C" int hso_gpu_align_multi(hso_gpu_ctx* ctx, const hso_camera* cam, const int64_t* cur_frame_ids, const hso_align_job* jobs,
                                   int n_jobs, hso_align_out* out)
{
  return align_run(ctx, cam, cur_frame_ids, 1, jobs, n_jobs, out);
}

// ---------------------------------------------------------------------------------------------
// Reprojector candidate generation chained in front of k_align:
// Reprojector::reprojectPoint (reference src/reprojector.cpp:504-529), Point::getCloseViewObs
// (src/point.cpp:116-136) and the job findMatchDirect derives from the chosen observation
// (src/matcher.cpp:288-319).  One lane per map point writes the point's grid record and its
// AlignJobDev in place; k_align then runs over the same array (ref_base == nullptr = no job), so
// projection, reference choice and matching need no host round trip.  The per-keyframe products
// T_cur_w * T_kf_w^-1 and the keyframe positions are formed once on the host (a dozen of them).
struct ReprojKf {
  Se3 T_cur_kf;              // cur.T_f_w_ * kf.T_f_w_^-1
  double pos[3];             // kf.pos()
  const uint8_t* base;
  int64_t frame_id;
  float exposure_rat;        // float(cur.m_exposure_time / kf.m_exposure_time)
  int32_t kf_gap_lt4;
};

// the frame a point is projected into (points of many current frames / sequences share a launch);
// its keyframes are kfs[kf_begin ...], and the points' keyframe indices are relative to that
struct ReprojFrameDev {
  double cur_pos[3];
  const uint8_t* cur_base;
  int kf_begin, pad_;
};

struct ReprojConsts {
  hso_camera cam;
  const ReprojFrameDev* frames;
  const int* pt_frame;       // frame index of every point
  const ReprojKf* kfs;
  const hso_map_point* pts;
  const hso_obs* obs;
  int n_pts, cell_size, grid_n_cols;
};

// one map point: reprojectPoint + getCloseViewObs + the findMatchDirect job (shared by the value-passing and the resident form)
// LINKED: the point's observations are a list threaded through the observation rows (obs_begin = first row, hso_obs.pad_ = next
// row: the sequence maps); otherwise the rows obs_begin .. obs_begin + obs_count (the value-passing and the stored-map forms)
template <bool LINKED = false>
HSO_DEV hso_reproj_point reproject_one(const hso_camera& cam, const hso_map_point& P, const ReprojFrameDev& F, const ReprojKf* kfs,
                                       const hso_obs* obs, int cell_size, int grid_n_cols, AlignJobDev* JD)
{
  hso_reproj_point o;
  o.projected = 0; o.cell = 0; o.px[0] = 0; o.px[1] = 0; o.ref_obs = -1; o.pad_ = 0;
  JD->ref_base = nullptr; JD->cur_base = F.cur_base;
  // reprojectPoint, :504-529
  const ReprojKf& H = kfs[P.host_kf];
  const double s = 1.0 / P.idist;
  double tx, ty, tz;
  se3_apply(H.T_cur_kf, P.host_f[0] * s, P.host_f[1] * s, P.host_f[2] * s, tx, ty, tz);
  if (!(tz < 0.00001)) {
    double u, v;
    world2cam(cam, tx, ty, tz, u, v);
    const int ix = (int)u, iy = (int)v;
    if (ix >= 8 && ix < cam.width - 8 && iy >= 8 && iy < cam.height - 8) {   // isInFrame(px.cast<int>(), 8)
      o.projected = 1;
      o.px[0] = u; o.px[1] = v;
      o.cell = (int)(v / cell_size) * grid_n_cols + (int)(u / cell_size);
    }
  }
  if (o.projected && P.obs_count > 0) {
    // getCloseViewObs, src/point.cpp:116-136
    double ox = F.cur_pos[0] - P.pos[0], oy = F.cur_pos[1] - P.pos[1], oz = F.cur_pos[2] - P.pos[2];
    { const double n = sqrt(ox * ox + oy * oy + oz * oz); ox /= n; oy /= n; oz /= n; }
    int best = P.obs_begin;
    double min_cos = 0;
    for (int k = 0, row = P.obs_begin; k < P.obs_count; k++) {
      const ReprojKf& K = kfs[obs[row].kf];
      double dx = K.pos[0] - P.pos[0], dy = K.pos[1] - P.pos[1], dz = K.pos[2] - P.pos[2];
      { const double n = sqrt(dx * dx + dy * dy + dz * dz); dx /= n; dy /= n; dz /= n; }
      const double c = ox * dx + oy * dy + oz * dz;
      if (c > min_cos) { min_cos = c; best = row; }
      row = LINKED ? obs[row].pad_ : row + 1;
    }
    if (!(min_cos < 0.5)) {
      o.ref_obs = best;
      const hso_obs ref = obs[o.ref_obs];
      const ReprojKf& K = kfs[ref.kf];
      hso_align_job j;
      j.ref_frame_id = K.frame_id;
      j.ref_level = ref.level; j.type = ref.type;
      j.px_ref[0] = ref.px[0]; j.px_ref[1] = ref.px[1];
      j.f_ref[0] = ref.f[0]; j.f_ref[1] = ref.f[1]; j.f_ref[2] = ref.f[2];
      j.grad[0] = ref.grad[0]; j.grad[1] = ref.grad[1];
      if (ref.kf == P.host_kf) {
        j.depth = 1.0 / P.idist;                                    // matcher.cpp:295-299
      } else {
        const double dx = K.pos[0] - P.pos[0], dy = K.pos[1] - P.pos[1], dz = K.pos[2] - P.pos[2];
        j.depth = sqrt(dx * dx + dy * dy + dz * dz);                // :301-305
      }
      se3_to(K.T_cur_kf, j.T_cur_ref);
      j.px_cur[0] = o.px[0]; j.px_cur[1] = o.px[1];
      j.exposure_rat = K.exposure_rat;
      j.kf_gap_lt4 = K.kf_gap_lt4;
      JD->j = j;
      JD->ref_base = K.base;
    }
  }
  return o;
}

__global__ __launch_bounds__(256) void k_reproject(ReprojConsts R, AlignJobDev* jobs, hso_reproj_point* proj)
{
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= R.n_pts) return;
  const ReprojFrameDev& F = R.frames[R.pt_frame[i]];
  proj[i] = reproject_one(R.cam, R.pts[i], F, R.kfs + F.kf_begin, R.obs, R.cell_size, R.grid_n_cols, &jobs[i]);
}

// ---- resident maps: the tables of many sequences' local maps stay in HBM (one equal-sized region each); a call names its map
struct MapCallDev {
  ReprojFrameDev F;          // kf_begin = first row of the call's ReprojKf block
  int map, point_begin, point_count, pad_;
};
struct MapConsts {
  hso_camera cam;
  const MapCallDev* calls;
  int n_calls, n_total;
  const ReprojKf* kfs;       // per call: max_kfs rows
  const hso_map_point* pts;  // arena: [n_maps][max_points]
  const hso_obs* obs;        // arena: [n_maps][max_obs]
  int max_points, max_obs, cell_size, grid_n_cols;
};

__global__ __launch_bounds__(256) void k_reproject_maps(MapConsts M, AlignJobDev* jobs, hso_reproj_point* proj)
{
  const int g = blockIdx.x * 256 + threadIdx.x;
  if (g >= M.n_total) return;
  int lo = 0, hi = M.n_calls - 1;          // the call whose point range holds g
  while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (M.calls[mid].point_begin <= g) lo = mid; else hi = mid - 1; }
  const MapCallDev& C = M.calls[lo];
  const int i = g - C.point_begin;
  const hso_map_point& pt = M.pts[(size_t)C.map * M.max_points + i];
  hso_reproj_point r = reproject_one(M.cam, pt, C.F, M.kfs + C.F.kf_begin, M.obs + (size_t)C.map * M.max_obs, M.cell_size, M.grid_n_cols, &jobs[g]);
  r.pad_ = pt.pad_;   // the point's quality key rides along for the on-device grid selection (hso_gpu_reproject_select_maps)
  proj[g] = r;
}

// projection + match of one point -> the compact record the host's grid selection consumes
__global__ __launch_bounds__(256) void k_match_brief(int n, const hso_reproj_point* proj, const hso_align_out* match, const AlignJobDev* jobs,
                                                     hso_match_brief* out)
{
  const int g = blockIdx.x * 256 + threadIdx.x;
  if (g >= n) return;
  const hso_reproj_point p = proj[g];
  const hso_align_out m = match[g];
  hso_match_brief b;
  memset(&b, 0, sizeof(b));
  b.cell = p.projected ? p.cell : -1;
  b.ref_obs = p.ref_obs;
  b.px[0] = p.px[0]; b.px[1] = p.px[1];
  b.px_cur[0] = m.px_cur[0]; b.px_cur[1] = m.px_cur[1];
  b.success = (int8_t)m.success; b.stage = (int8_t)m.stage; b.search_level = (int8_t)m.search_level;
  if (p.ref_obs >= 0) {
    const hso_align_job& j = jobs[g].j;
    b.ref_type = (int8_t)j.type;
    // the new feature's gradient direction (reprojector.cpp:400-406): normalised A_cur_ref * ref grad
    const double gx = m.A_cur_ref[0] * j.grad[0] + m.A_cur_ref[1] * j.grad[1], gy = m.A_cur_ref[2] * j.grad[0] + m.A_cur_ref[3] * j.grad[1];
    const double nn = sqrt(gx * gx + gy * gy);
    b.grad[0] = nn > 0 ? (float)(gx / nn) : 0.f; b.grad[1] = nn > 0 ? (float)(gy / nn) : 0.f;
  }
  out[g] = b;
}

// k_align over device-built jobs: a null reference = "findMatchDirect not reached / returned at once"

extern "C" int hso_gpu_reproject_match_multi(hso_gpu_ctx* ctx, const hso_camera* cam, const hso_reproj_frame* frames, int n_frames,
                                             const hso_kf* kfs, int n_kfs, const hso_map_point* points, int n_points,
                                             const hso_obs* obs, int n_obs, int cell_size, int grid_n_cols,
                                             hso_reproj_point* proj_out, hso_align_out* match_out)
{
  if (!ctx) return HSO_E_INVALID;
  if (!cam || n_frames < 0 || n_kfs < 0 || n_points < 0 || n_obs < 0 || cell_size < 1 || grid_n_cols < 1 ||
      (n_points > 0 && (!points || !proj_out || !match_out || !kfs || n_kfs == 0 || !frames || n_frames == 0)) || (n_obs > 0 && !obs))
    return hso_fail(ctx, HSO_E_INVALID, "reproject_match: bad argument");
  if (n_points == 0) return HSO_OK;
  HSO_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  PyrGeom g{};
  std::vector<ReprojKf> hk(n_kfs);
  std::vector<ReprojFrameDev> hf(n_frames);
  std::vector<int> pt_frame(n_points, -1);
  for (int f = 0; f < n_frames; f++) {
    const hso_reproj_frame& FR = frames[f];
    auto itc = ctx->frames.find(FR.cur_frame_id);
    if (itc == ctx->frames.end()) return hso_fail(ctx, HSO_E_NOFRAME, "reproject_match: current frame not resident");
    if (f == 0) g = itc->second.g;
    else if (!same_geom(itc->second.g, g)) return hso_fail(ctx, HSO_E_INVALID, "reproject_match: frames must share one size");
    if (FR.kf_begin < 0 || FR.kf_count < 0 || (long long)FR.kf_begin + FR.kf_count > n_kfs || FR.point_begin < 0 || FR.point_count < 0 ||
        (long long)FR.point_begin + FR.point_count > n_points)
      return hso_fail(ctx, HSO_E_INVALID, "reproject_match: frame table out of range");
    const Se3 Tc = se3_from(FR.T_cur_w);
    const Se3 ci = se3_inverse(Tc);
    hf[f].cur_pos[0] = ci.tx; hf[f].cur_pos[1] = ci.ty; hf[f].cur_pos[2] = ci.tz;
    hf[f].cur_base = itc->second.base; hf[f].kf_begin = FR.kf_begin; hf[f].pad_ = 0;
    for (int k = FR.kf_begin; k < FR.kf_begin + FR.kf_count; k++) {
      auto it = ctx->frames.find(kfs[k].frame_id);
      if (it == ctx->frames.end()) return hso_fail(ctx, HSO_E_NOFRAME, "reproject_match: keyframe not resident");
      if (!same_geom(it->second.g, g)) return hso_fail(ctx, HSO_E_INVALID, "reproject_match: frames must share one size");
      const Se3 inv = se3_inverse(se3_from(kfs[k].T_f_w));
      hk[k].T_cur_kf = se3_mul(Tc, inv);
      hk[k].pos[0] = inv.tx; hk[k].pos[1] = inv.ty; hk[k].pos[2] = inv.tz;
      hk[k].base = it->second.base;
      hk[k].frame_id = kfs[k].frame_id;
      hk[k].exposure_rat = (float)(FR.cur_exposure_time / kfs[k].exposure_time);
      hk[k].kf_gap_lt4 = (FR.cur_keyframe_id - kfs[k].keyframe_id) < 4;
    }
    // the index tables are the caller's: check them here, the kernels trust them
    for (int i = FR.point_begin; i < FR.point_begin + FR.point_count; i++) {
      const hso_map_point& p = points[i];
      if (pt_frame[i] != -1) return hso_fail(ctx, HSO_E_INVALID, "reproject_match: point ranges of two frames overlap");
      pt_frame[i] = f;
      if (p.host_kf < 0 || p.host_kf >= FR.kf_count || p.obs_count < 0 || p.obs_begin < 0 || (long long)p.obs_begin + p.obs_count > n_obs)
        return hso_fail(ctx, HSO_E_INVALID, "reproject_match: point table out of range");
      for (int k = p.obs_begin; k < p.obs_begin + p.obs_count; k++)
        if (obs[k].kf < 0 || obs[k].kf >= FR.kf_count || obs[k].level < 0 || obs[k].level >= HSO_N_PYR_LEVELS)
          return hso_fail(ctx, HSO_E_INVALID, "reproject_match: observation table out of range");
    }
  }
  for (int i = 0; i < n_points; i++)
    if (pt_frame[i] < 0) return hso_fail(ctx, HSO_E_INVALID, "reproject_match: a point belongs to no frame");
  if (cam->width != g.w[0] || cam->height != g.h[0]) return hso_fail(ctx, HSO_E_INVALID, "reproject_match: camera size differs from the frame size");
  auto al = [](size_t b) { return (b + 255) & ~size_t(255); };
  const size_t o_jobs = 0, o_out = o_jobs + al(sizeof(AlignJobDev) * (size_t)n_points);
  const size_t o_proj = o_out + al(sizeof(hso_align_out) * (size_t)n_points);
  const size_t o_kf = o_proj + al(sizeof(hso_reproj_point) * (size_t)n_points);
  const size_t o_pts = o_kf + al(sizeof(ReprojKf) * (size_t)n_kfs);
  const size_t o_obs = o_pts + al(sizeof(hso_map_point) * (size_t)n_points);
  const size_t o_fr = o_obs + al(sizeof(hso_obs) * (size_t)(n_obs > 0 ? n_obs : 1));
  const size_t o_pf = o_fr + al(sizeof(ReprojFrameDev) * (size_t)n_frames);
  const size_t need = o_pf + al(sizeof(int) * (size_t)n_points);
  if (ctx->batch_cap < need) {
    HSO_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->d_batch) (void)hipFree(ctx->d_batch);
    ctx->d_batch = nullptr; ctx->batch_cap = 0;
    HSO_HIP_CHECK(ctx, hipMalloc(reinterpret_cast<void**>(&ctx->d_batch), hso_grown(need)));
    ctx->batch_cap = hso_grown(need);
  }
  char* d = ctx->d_batch;
  // Small calls (one keyframe-sized sequence) are latency-bound: one pinned staging image of
  // [kf | points | obs | frames | point->frame] and one DMA each way.  Large multi-sequence tables
  // go straight from the caller's memory (the runtime pipelines pageable copies in chunks, which
  // beats an extra host pass over tens of megabytes); only the small derived tables are staged.
  const size_t in_bytes = need - o_kf;
  const bool small = in_bytes < ((size_t)1 << 20);
  char* hin = hso_pinned(ctx, 0, small ? in_bytes : (o_pts - o_kf) + (need - o_fr));
  char* hout = small ? hso_pinned(ctx, 1, o_kf - o_out) : nullptr;
  if (!hin || (small && !hout)) return HSO_E_NOMEM;
  if (small) {
    memcpy(hin, hk.data(), sizeof(ReprojKf) * (size_t)n_kfs);
    memcpy(hin + (o_pts - o_kf), points, sizeof(hso_map_point) * (size_t)n_points);
    if (n_obs > 0) memcpy(hin + (o_obs - o_kf), obs, sizeof(hso_obs) * (size_t)n_obs);
    memcpy(hin + (o_fr - o_kf), hf.data(), sizeof(ReprojFrameDev) * (size_t)n_frames);
    memcpy(hin + (o_pf - o_kf), pt_frame.data(), sizeof(int) * (size_t)n_points);
    HSO_HIP_CHECK(ctx, hipMemcpyAsync(d + o_kf, hin, in_bytes, hipMemcpyHostToDevice, ctx->stream));
  } else {
    char* hfr = hin + (o_pts - o_kf);
    memcpy(hin, hk.data(), sizeof(ReprojKf) * (size_t)n_kfs);
    memcpy(hfr, hf.data(), sizeof(ReprojFrameDev) * (size_t)n_frames);
    memcpy(hfr + (o_pf - o_fr), pt_frame.data(), sizeof(int) * (size_t)n_points);
    HSO_HIP_CHECK(ctx, hipMemcpyAsync(d + o_kf, hin, sizeof(ReprojKf) * (size_t)n_kfs, hipMemcpyHostToDevice, ctx->stream));
    HSO_HIP_CHECK(ctx, hipMemcpyAsync(d + o_fr, hfr, need - o_fr, hipMemcpyHostToDevice, ctx->stream));
    HSO_HIP_CHECK(ctx, hipMemcpyAsync(d + o_pts, points, sizeof(hso_map_point) * (size_t)n_points, hipMemcpyHostToDevice, ctx->stream));
    if (n_obs > 0) HSO_HIP_CHECK(ctx, hipMemcpyAsync(d + o_obs, obs, sizeof(hso_obs) * (size_t)n_obs, hipMemcpyHostToDevice, ctx->stream));
  }
  HSO_HIP_CHECK(ctx, hipMemsetAsync(d + o_out, 0, sizeof(hso_align_out) * (size_t)n_points, ctx->stream));
  ReprojConsts R;
  R.cam = *cam;
  R.frames = reinterpret_cast<const ReprojFrameDev*>(d + o_fr);
  R.pt_frame = reinterpret_cast<const int*>(d + o_pf);
  R.kfs = reinterpret_cast<const ReprojKf*>(d + o_kf);
  R.pts = reinterpret_cast<const hso_map_point*>(d + o_pts);
  R.obs = reinterpret_cast<const hso_obs*>(d + o_obs);
  R.n_pts = n_points; R.cell_size = cell_size; R.grid_n_cols = grid_n_cols;
  AlignJobDev* d_jobs = reinterpret_cast<AlignJobDev*>(d + o_jobs);
  hso_align_out* d_out = reinterpret_cast<hso_align_out*>(d + o_out);
  hso_reproj_point* d_proj = reinterpret_cast<hso_reproj_point*>(d + o_proj);
  hipLaunchKernelGGL(k_reproject, dim3((n_points + 255) / 256), dim3(256), 0, ctx->stream, R, d_jobs, d_proj);
  AlignConsts C;
  C.cam = *cam; C.g = g;
  launch_align(ctx, true, C, d_jobs, n_points, d_out);
  HSO_HIP_CHECK(ctx, hipGetLastError());
  if (small) {  // [match | proj] are adjacent on the device: one DMA into pinned memory, then to the caller
    HSO_HIP_CHECK(ctx, hipMemcpyAsync(hout, d + o_out, o_kf - o_out, hipMemcpyDeviceToHost, ctx->stream));
    HSO_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    memcpy(match_out, hout, sizeof(hso_align_out) * (size_t)n_points);
    memcpy(proj_out, hout + (o_proj - o_out), sizeof(hso_reproj_point) * (size_t)n_points);
  } else {
    HSO_HIP_CHECK(ctx, hipMemcpyAsync(proj_out, d_proj, sizeof(hso_reproj_point) * (size_t)n_points, hipMemcpyDeviceToHost, ctx->stream));
    HSO_HIP_CHECK(ctx, hipMemcpyAsync(match_out, d_out, sizeof(hso_align_out) * (size_t)n_points, hipMemcpyDeviceToHost, ctx->stream));
    HSO_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  }
  return HSO_OK;
}

extern "C" int hso_gpu_reproject_match(hso_gpu_ctx* ctx, const hso_camera* cam, int64_t cur_frame_id, const hso_se3* T_cur_w,
                                       double cur_exposure_time, int cur_keyframe_id, const hso_kf* kfs, int n_kfs,
                                       const hso_map_point* points, int n_points, const hso_obs* obs, int n_obs, int cell_size,
                                       int grid_n_cols, hso_reproj_point* proj_out, hso_align_out* match_out)
{
  if (!ctx) return HSO_E_INVALID;
  if (!T_cur_w) return hso_fail(ctx, HSO_E_INVALID, "reproject_match: bad argument");
  hso_reproj_frame f;
  f.cur_frame_id = cur_frame_id; f.T_cur_w = *T_cur_w; f.cur_exposure_time = cur_exposure_time; f.cur_keyframe_id = cur_keyframe_id;
  f.kf_begin = 0; f.kf_count = n_kfs; f.point_begin = 0; f.point_count = n_points; f.pad_ = 0;
  return hso_gpu_reproject_match_multi(ctx, cam, &f, 1, kfs, n_kfs, points, n_points, obs, n_obs, cell_size, grid_n_cols, proj_out, match_out);
}

// ---------------------------------------------------------------------------------------------
// Resident maps: a sequence's local map — keyframe poses, map points, observations,
// the tables of hso_gpu_reproject_match — changes at keyframe rate only, so it is stored once per keyframe
// (hso_gpu_map_store) and every frame in between passes its pose alone; the results come back as 56-byte records.
struct MapArena {
  int n_maps = 0, max_kfs = 0, max_points = 0, max_obs = 0;
  hso_map_point* d_pts = nullptr;
  hso_obs* d_obs = nullptr;
  std::vector<std::vector<hso_kf>> kfs;     // per map (host: the per-call products are formed from them)
  std::vector<int> n_points, n_obs;
  PyrGeom g{}; bool have_g = false;
};

void hso_map_arena_free(hso_gpu_ctx* ctx)
{
  if (!ctx->maps) return;
  (void)hipFree(ctx->maps->d_pts); (void)hipFree(ctx->maps->d_obs);
  delete ctx->maps;
  ctx->maps = nullptr;
}

// what the chained pose optimisation (hso_select.hip) needs of the stored maps
const hso_map_point* hso_map_points_dev(hso_gpu_ctx* ctx) { return ctx->maps ? ctx->maps->d_pts : nullptr; }
int hso_map_max_points(hso_gpu_ctx* ctx) { return ctx->maps ? ctx->maps->max_points : 0; }
int hso_map_max_kfs(hso_gpu_ctx* ctx) { return ctx->maps ? ctx->maps->max_kfs : 0; }
int hso_map_kf_poses(hso_gpu_ctx* ctx, int map, hso_se3* out)
{
  const std::vector<hso_kf>& kfs = ctx->maps->kfs[map];
  for (size_t k = 0; k < kfs.size(); k++) out[k] = kfs[k].T_f_w;
  return (int)kfs.size();
}

extern "C" int hso_gpu_map_reserve(hso_gpu_ctx* ctx, int n_maps, int max_kfs, int max_points, int max_obs)
{
  if (!ctx) return HSO_E_INVALID;
  if (n_maps <= 0 || max_kfs <= 0 || max_points <= 0 || max_obs <= 0) return hso_fail(ctx, HSO_E_INVALID, "map_reserve: bad argument");
  HSO_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  HSO_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  hso_map_arena_free(ctx);
  MapArena* A = new MapArena();
  A->n_maps = n_maps; A->max_kfs = max_kfs; A->max_points = max_points; A->max_obs = max_obs;
  A->kfs.resize(n_maps); A->n_points.assign(n_maps, 0); A->n_obs.assign(n_maps, 0);
  if (hipMalloc(reinterpret_cast<void**>(&A->d_pts), sizeof(hso_map_point) * (size_t)n_maps * max_points) != hipSuccess ||
      hipMalloc(reinterpret_cast<void**>(&A->d_obs), sizeof(hso_obs) * (size_t)n_maps * max_obs) != hipSuccess) {
    (void)hipFree(A->d_pts); (void)hipFree(A->d_obs); delete A;
    return hso_fail(ctx, HSO_E_NOMEM, "map_reserve: out of device memory");
  }
  ctx->maps = A;
  return HSO_OK;
}

extern "C" int hso_gpu_map_store(hso_gpu_ctx* ctx, int map, const hso_kf* kfs, int n_kfs, const hso_map_point* points, int n_points,
                                 const hso_obs* obs, int n_obs)
{
  if (!ctx) return HSO_E_INVALID;
  MapArena* A = ctx->maps;
  if (!A || map < 0 || map >= A->n_maps) return hso_fail(ctx, HSO_E_INVALID, "map_store: no such map (hso_gpu_map_reserve first)");
  if (n_kfs < 0 || n_points < 0 || n_obs < 0 || n_kfs > A->max_kfs || n_points > A->max_points || n_obs > A->max_obs ||
      (n_kfs > 0 && !kfs) || (n_points > 0 && !points) || (n_obs > 0 && !obs))
    return hso_fail(ctx, HSO_E_INVALID, "map_store: table larger than the reserved region, or null");
  for (int k = 0; k < n_kfs; k++) {
    auto it = ctx->frames.find(kfs[k].frame_id);
    if (it == ctx->frames.end()) return hso_fail(ctx, HSO_E_NOFRAME, "map_store: keyframe not resident");
    if (!A->have_g) { A->g = it->second.g; A->have_g = true; }
    if (!same_geom(it->second.g, A->g)) return hso_fail(ctx, HSO_E_INVALID, "map_store: frames must share one size");
  }
  for (int i = 0; i < n_points; i++) {
    const hso_map_point& p = points[i];
    if (p.host_kf < 0 || p.host_kf >= n_kfs || p.obs_count < 0 || p.obs_begin < 0 || (long long)p.obs_begin + p.obs_count > n_obs)
      return hso_fail(ctx, HSO_E_INVALID, "map_store: point table out of range");
  }
  for (int k = 0; k < n_obs; k++)
    if (obs[k].kf < 0 || obs[k].kf >= n_kfs || obs[k].level < 0 || obs[k].level >= HSO_N_PYR_LEVELS)
      return hso_fail(ctx, HSO_E_INVALID, "map_store: observation table out of range");
  HSO_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  const size_t bp = sizeof(hso_map_point) * (size_t)n_points, bo = sizeof(hso_obs) * (size_t)n_obs;
  char* h = hso_pinned(ctx, 0, bp + bo + 64);
  if (!h) return HSO_E_NOMEM;
  if (bp) memcpy(h, points, bp);
  if (bo) memcpy(h + bp, obs, bo);
  if (bp) HSO_HIP_CHECK(ctx, hipMemcpyAsync(A->d_pts + (size_t)map * A->max_points, h, bp, hipMemcpyHostToDevice, ctx->stream));
  if (bo) HSO_HIP_CHECK(ctx, hipMemcpyAsync(A->d_obs + (size_t)map * A->max_obs, h + bp, bo, hipMemcpyHostToDevice, ctx->stream));
  HSO_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  A->kfs[map].assign(kfs, kfs + n_kfs);
  A->n_points[map] = n_points; A->n_obs[map] = n_obs;
  return HSO_OK;
}

// The per-frame part of a stored map.  Between two hso_gpu_map_store calls the reference changes, every frame, what the grid
// selection orders and skips by: n_succeeded_reproj_ > 10 turns TYPE_UNKNOWN into TYPE_GOOD (the comparator's key), n_failed_reproj_
// > 15 / 30 deletes points (src/reprojector.cpp:376-392, 412-423).  Those live in hso_map_point.pad_ — the quality key
// (Point::type_ << 4) | ftr_type_, 0 = deleted — and this call refreshes the keys of one stored map from a byte per point
// for any number of stored maps in one call (the bytes cross PCIe, not the tables; one small kernel scatters them onto the pad_
// column).  Structural changes — new points (candidates
// promoted on non-keyframes, temporary points), new observations, another keyframe set — still need hso_gpu_map_store.
__global__ void k_map_quality(hso_map_point* pts, int max_points, const int* maps, const int* begin, int n_maps, const uint8_t* quality)
{
  const int m = blockIdx.y;
  if (m >= n_maps) return;
  const int n = begin[m + 1] - begin[m];
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
    pts[(size_t)maps[m] * max_points + i].pad_ = quality[begin[m] + i];
}

extern "C" int hso_gpu_map_update_quality(hso_gpu_ctx* ctx, const int32_t* maps, int n_maps, const uint8_t* quality)
{
  if (!ctx) return HSO_E_INVALID;
  MapArena* A = ctx->maps;
  if (!A || n_maps < 0 || (n_maps > 0 && (!maps || !quality))) return hso_fail(ctx, HSO_E_INVALID, "map_update_quality: bad argument");
  if (n_maps == 0) return HSO_OK;
  HSO_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  size_t total = 0;
  for (int m = 0; m < n_maps; m++) {
    if (maps[m] < 0 || maps[m] >= A->n_maps) return hso_fail(ctx, HSO_E_INVALID, "map_update_quality: no such map");
    total += (size_t)A->n_points[maps[m]];
  }
  if (total == 0) return HSO_OK;
  auto al = [](size_t b) { return (b + 255) & ~size_t(255); };
  const size_t b_maps = al(sizeof(int) * (size_t)n_maps), b_begin = al(sizeof(int) * (size_t)(n_maps + 1)), need = b_maps + b_begin + al(total);
  char* h = hso_pinned(ctx, 0, need);
  if (!h) return HSO_E_NOMEM;
  int* hm = reinterpret_cast<int*>(h);
  int* hb = reinterpret_cast<int*>(h + b_maps);
  int t = 0;
  for (int m = 0; m < n_maps; m++) { hm[m] = maps[m]; hb[m] = t; t += A->n_points[maps[m]]; }
  hb[n_maps] = t;
  memcpy(h + b_maps + b_begin, quality, total);
  if (ctx->batch_cap < need) {
    HSO_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->d_batch) (void)hipFree(ctx->d_batch);
    ctx->d_batch = nullptr; ctx->batch_cap = 0;
    HSO_HIP_CHECK(ctx, hipMalloc(reinterpret_cast<void**>(&ctx->d_batch), hso_grown(need)));
    ctx->batch_cap = hso_grown(need);
  }
  char* d = ctx->d_batch;
  HSO_HIP_CHECK(ctx, hipMemcpyAsync(d, h, need, hipMemcpyHostToDevice, ctx->stream));
  int max_n = 0;
  for (int m = 0; m < n_maps; m++) max_n = std::max(max_n, A->n_points[maps[m]]);
  hipLaunchKernelGGL(k_map_quality, dim3((max_n + 255) / 256, n_maps), dim3(256), 0, ctx->stream, A->d_pts, A->max_points,
                     reinterpret_cast<const int*>(d), reinterpret_cast<const int*>(d + b_maps), n_maps, reinterpret_cast<const uint8_t*>(d + b_maps + b_begin));
  HSO_HIP_CHECK(ctx, hipGetLastError());
  HSO_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  return HSO_OK;
}

int hso_map_call_sizes(hso_gpu_ctx* ctx, const hso_map_call* calls, int n_calls, MapArenaSizes* Z)
{
  MapArena* A = ctx->maps;
  Z->total = 0;
  if (!A || n_calls < 0 || (n_calls > 0 && !calls)) return hso_fail(ctx, HSO_E_INVALID, "reproject_match_maps: bad argument");
  for (int c = 0; c < n_calls; c++) {
    if (calls[c].map < 0 || calls[c].map >= A->n_maps) return hso_fail(ctx, HSO_E_INVALID, "reproject_match_maps: no such map");
    Z->total += A->n_points[calls[c].map];
  }
  return HSO_OK;
}

// The launch chain of hso_gpu_reproject_match_maps without the read-back: the records stay on the device (R), with
// `extra_bytes` of the work area reserved behind them for a caller that goes on working there (the grid selection).
int hso_reproject_maps_run(hso_gpu_ctx* ctx, const hso_camera* cam, const hso_map_call* calls, int n_calls, int cell_size,
                           int grid_n_cols, size_t extra_bytes, HsoMapsRun* R)
{
  MapArena* A = ctx->maps;
  if (!A || !cam || n_calls < 0 || (n_calls > 0 && !calls) || cell_size < 1 || grid_n_cols < 1)
    return hso_fail(ctx, HSO_E_INVALID, "reproject_match_maps: bad argument");
  R->n = 0; R->begin.assign(n_calls + 1, 0);
  if (n_calls == 0) return 0;
  HSO_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  size_t total = 0;
  for (int c = 0; c < n_calls; c++) {
    if (calls[c].map < 0 || calls[c].map >= A->n_maps) return hso_fail(ctx, HSO_E_INVALID, "reproject_match_maps: no such map");
    total += (size_t)A->n_points[calls[c].map];
    R->begin[c + 1] = (int)total;
  }
  if (total == 0) return 0;
  if (cam->width != A->g.w[0] || cam->height != A->g.h[0]) return hso_fail(ctx, HSO_E_INVALID, "reproject_match_maps: camera size differs from the frame size");
  auto al = [](size_t b) { return (b + 255) & ~size_t(255); };
  const size_t b_calls = al(sizeof(MapCallDev) * (size_t)n_calls), b_kfs = al(sizeof(ReprojKf) * (size_t)n_calls * A->max_kfs);
  char* hin = hso_pinned(ctx, 0, b_calls + b_kfs);
  if (!hin) return HSO_E_NOMEM;
  MapCallDev* hc = reinterpret_cast<MapCallDev*>(hin);
  ReprojKf* hk = reinterpret_cast<ReprojKf*>(hin + b_calls);
  int begin = 0;
  for (int c = 0; c < n_calls; c++) {
    const hso_map_call& K = calls[c];
    auto itc = ctx->frames.find(K.cur_frame_id);
    if (itc == ctx->frames.end()) return hso_fail(ctx, HSO_E_NOFRAME, "reproject_match_maps: current frame not resident");
    if (!same_geom(itc->second.g, A->g)) return hso_fail(ctx, HSO_E_INVALID, "reproject_match_maps: frames must share one size");
    const Se3 Tc = se3_from(K.T_cur_w);
    const Se3 ci = se3_inverse(Tc);
    MapCallDev& D = hc[c];
    D.F.cur_pos[0] = ci.tx; D.F.cur_pos[1] = ci.ty; D.F.cur_pos[2] = ci.tz;
    D.F.cur_base = itc->second.base; D.F.kf_begin = c * A->max_kfs; D.F.pad_ = 0;
    D.map = K.map; D.point_begin = begin; D.point_count = A->n_points[K.map]; D.pad_ = 0;
    begin += D.point_count;
    const std::vector<hso_kf>& kfs = A->kfs[K.map];
    for (size_t k = 0; k < kfs.size(); k++) {
      auto it = ctx->frames.find(kfs[k].frame_id);
      if (it == ctx->frames.end()) return hso_fail(ctx, HSO_E_NOFRAME, "reproject_match_maps: a stored keyframe is no longer resident");
      ReprojKf& R = hk[(size_t)c * A->max_kfs + k];
      const Se3 inv = se3_inverse(se3_from(kfs[k].T_f_w));
      R.T_cur_kf = se3_mul(Tc, inv);
      R.pos[0] = inv.tx; R.pos[1] = inv.ty; R.pos[2] = inv.tz;
      R.base = it->second.base; R.frame_id = kfs[k].frame_id;
      R.exposure_rat = (float)(K.cur_exposure_time / kfs[k].exposure_time);
      R.kf_gap_lt4 = (K.cur_keyframe_id - kfs[k].keyframe_id) < 4;
    }
  }
  const size_t o_jobs = 0, o_match = o_jobs + al(sizeof(AlignJobDev) * total), o_proj = o_match + al(sizeof(hso_align_out) * total);
  const size_t o_brief = o_proj + al(sizeof(hso_reproj_point) * total), o_in = o_brief + al(sizeof(hso_match_brief) * total);
  const size_t o_extra = o_in + al(b_calls + b_kfs);
  const size_t need = o_extra + extra_bytes;
  if (ctx->batch_cap < need) {
    HSO_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->d_batch) (void)hipFree(ctx->d_batch);
    ctx->d_batch = nullptr; ctx->batch_cap = 0;
    HSO_HIP_CHECK(ctx, hipMalloc(reinterpret_cast<void**>(&ctx->d_batch), hso_grown(need)));
    ctx->batch_cap = hso_grown(need);
  }
  char* d = ctx->d_batch;
  HSO_HIP_CHECK(ctx, hipMemcpyAsync(d + o_in, hin, b_calls + b_kfs, hipMemcpyHostToDevice, ctx->stream));
  HSO_HIP_CHECK(ctx, hipMemsetAsync(d + o_match, 0, sizeof(hso_align_out) * total, ctx->stream));
  MapConsts M;
  M.cam = *cam; M.calls = reinterpret_cast<const MapCallDev*>(d + o_in); M.n_calls = n_calls; M.n_total = (int)total;
  M.kfs = reinterpret_cast<const ReprojKf*>(d + o_in + b_calls); M.pts = A->d_pts; M.obs = A->d_obs;
  M.max_points = A->max_points; M.max_obs = A->max_obs; M.cell_size = cell_size; M.grid_n_cols = grid_n_cols;
  AlignJobDev* d_jobs = reinterpret_cast<AlignJobDev*>(d + o_jobs);
  hso_align_out* d_match = reinterpret_cast<hso_align_out*>(d + o_match);
  hso_reproj_point* d_proj = reinterpret_cast<hso_reproj_point*>(d + o_proj);
  hso_match_brief* d_brief = reinterpret_cast<hso_match_brief*>(d + o_brief);
  const int n = (int)total;
  hipLaunchKernelGGL(k_reproject_maps, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, M, d_jobs, d_proj);
  AlignConsts C;
  C.cam = *cam; C.g = A->g;
  launch_align(ctx, true, C, d_jobs, n, d_match);
  hipLaunchKernelGGL(k_match_brief, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, n, d_proj, d_match, d_jobs, d_brief);
  HSO_HIP_CHECK(ctx, hipGetLastError());
  R->n = n; R->d_proj = d_proj; R->d_brief = d_brief; R->d_extra = d + o_extra;
  return n;
}

extern "C" int hso_gpu_reproject_match_maps(hso_gpu_ctx* ctx, const hso_camera* cam, const hso_map_call* calls, int n_calls, int cell_size,
                                            int grid_n_cols, hso_match_brief* out, int out_capacity)
{
  if (!ctx) return HSO_E_INVALID;
  HsoMapsRun R;
  const int total = hso_reproject_maps_run(ctx, cam, calls, n_calls, cell_size, grid_n_cols, 0, &R);
  if (total <= 0) return total;
  if (!out || out_capacity < total) return hso_fail(ctx, HSO_E_INVALID, "reproject_match_maps: output smaller than the calls' points");
  hso_match_brief* hb = reinterpret_cast<hso_match_brief*>(hso_pinned(ctx, 1, sizeof(hso_match_brief) * (size_t)total));
  if (!hb) return HSO_E_NOMEM;
  HSO_HIP_CHECK(ctx, hipMemcpyAsync(hb, R.d_brief, sizeof(hso_match_brief) * (size_t)total, hipMemcpyDeviceToHost, ctx->stream));
  HSO_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  memcpy(out, hb, sizeof(hso_match_brief) * (size_t)total);
  return total;
}

// ---------------------------------------------------------------------------------------------
// Sequence maps (include/hso_gpu.h: hso_gpu_seqmap_*): the point and observation tables of a whole sequence, indexed by the
// caller's own point / feature ids and patched row by row; a frame names the points it projects as an id list.
struct SeqMap {
  hso_map_point* d_pts = nullptr; size_t pts_cap = 0, n_pts = 0;
  hso_obs* d_obs = nullptr; size_t obs_cap = 0, n_obs = 0;
  std::vector<hso_kf> kfs;
};
struct SeqMaps {
  std::vector<SeqMap*> m;
  PyrGeom g{}; bool have_g = false;
  // where the last hso_gpu_reproject_select_pose_frames call left its tables (hso_gpu_debug_fetch)
  const void* dbg[5] = {nullptr, nullptr, nullptr, nullptr, nullptr}; size_t dbg_bytes[5] = {0, 0, 0, 0, 0};
};

void hso_seqmaps_free(hso_gpu_ctx* ctx)
{
  if (!ctx->seqmaps) return;
  for (SeqMap* m : ctx->seqmaps->m)
    if (m) { (void)hipFree(m->d_pts); (void)hipFree(m->d_obs); delete m; }
  delete ctx->seqmaps;
  ctx->seqmaps = nullptr;
}

static SeqMap* seqmap_of(hso_gpu_ctx* ctx, int map)
{
  if (!ctx->seqmaps || map < 0 || map >= (int)ctx->seqmaps->m.size()) return nullptr;
  return ctx->seqmaps->m[map];
}

// rows of 8-byte granules: dst[ids[i]] = src[i]
static __global__ void k_scatter_rows(unsigned long long* __restrict__ dst, const int* __restrict__ ids, const unsigned long long* __restrict__ src, int n,
                                      int granules)
{
  const size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= (size_t)n * granules) return;
  const size_t i = g / granules, q = g - i * granules;
  dst[(size_t)ids[i] * granules + q] = src[g];
}
static __global__ void k_gather_rows(const unsigned long long* __restrict__ src, const int* __restrict__ ids, unsigned long long* __restrict__ dst, int n,
                                     int granules)
{
  const size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= (size_t)n * granules) return;
  const size_t i = g / granules, q = g - i * granules;
  dst[g] = src[(size_t)ids[i] * granules + q];
}
// rows to arbitrary destinations (the patches of many maps in one launch): row i goes to dst[i]
static __global__ void k_scatter_rows_to(unsigned long long* const* __restrict__ dst, const unsigned long long* __restrict__ src, int n, int granules)
{
  const size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= (size_t)n * granules) return;
  const size_t i = g / granules, q = g - i * granules;
  dst[i][q] = src[g];
}
static_assert(sizeof(hso_map_point) % 8 == 0 && sizeof(hso_obs) % 8 == 0, "rows move in 8-byte granules");

template <typename T> static int seqmap_grow(hso_gpu_ctx* ctx, T** p, size_t* cap, size_t need, size_t keep)
{
  if (*cap >= need) return HSO_OK;
  const size_t ncap = std::max(need + need / 2, (size_t)4096);
  T* q = nullptr;
  HSO_HIP_CHECK(ctx, hipMalloc(reinterpret_cast<void**>(&q), ncap * sizeof(T)));
  hipError_t e = hipMemsetAsync(q, 0, ncap * sizeof(T), ctx->stream);
  if (e == hipSuccess && *p && keep) e = hipMemcpyAsync(q, *p, keep * sizeof(T), hipMemcpyDeviceToDevice, ctx->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  if (e != hipSuccess) { (void)hipFree(q); ctx->err = std::string("seqmap: ") + hipGetErrorString(e); return HSO_E_HIP; }
  if (*p) (void)hipFree(*p);
  *p = q; *cap = ncap;
  return HSO_OK;
}

extern "C" {

int hso_gpu_seqmap_create(hso_gpu_ctx* ctx, int* map_out)
{
  if (!ctx || !map_out) return HSO_E_INVALID;
  if (!ctx->seqmaps) ctx->seqmaps = new SeqMaps();
  ctx->seqmaps->m.push_back(new SeqMap());
  *map_out = (int)ctx->seqmaps->m.size() - 1;
  return HSO_OK;
}

int hso_gpu_seqmap_destroy(hso_gpu_ctx* ctx, int map)
{
  if (!ctx) return HSO_E_INVALID;
  SeqMap* m = seqmap_of(ctx, map);
  if (!m) return hso_fail(ctx, HSO_E_INVALID, "seqmap: no such map");
  HSO_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  (void)hipFree(m->d_pts); (void)hipFree(m->d_obs);
  delete m;
  ctx->seqmaps->m[map] = nullptr;
  return HSO_OK;
}

int hso_gpu_seqmap_set_keyframes(hso_gpu_ctx* ctx, int map, const hso_kf* kfs, int n_kfs)
{
  if (!ctx) return HSO_E_INVALID;
  SeqMap* m = seqmap_of(ctx, map);
  if (!m || n_kfs < 0 || (n_kfs > 0 && !kfs)) return hso_fail(ctx, HSO_E_INVALID, "seqmap_set_keyframes: bad argument");
  SeqMaps* S = ctx->seqmaps;
  for (int k = 0; k < n_kfs; k++) {
    auto it = ctx->frames.find(kfs[k].frame_id);
    if (it == ctx->frames.end()) return hso_fail(ctx, HSO_E_NOFRAME, "seqmap_set_keyframes: keyframe not resident");
    if (!S->have_g) { S->g = it->second.g; S->have_g = true; }
    if (!same_geom(it->second.g, S->g)) return hso_fail(ctx, HSO_E_INVALID, "seqmap_set_keyframes: frames must share one size");
  }
  m->kfs.assign(kfs, kfs + n_kfs);
  return HSO_OK;
}

int hso_gpu_seqmap_patch(hso_gpu_ctx* ctx, int map, const int32_t* point_ids, const hso_map_point* points, int n_points,
                         const int32_t* obs_ids, const hso_obs* obs, int n_obs)
{
  if (!ctx) return HSO_E_INVALID;
  SeqMap* m = seqmap_of(ctx, map);
  if (!m || n_points < 0 || n_obs < 0 || (n_points > 0 && (!point_ids || !points)) || (n_obs > 0 && (!obs_ids || !obs)))
    return hso_fail(ctx, HSO_E_INVALID, "seqmap_patch: bad argument");
  if (n_points == 0 && n_obs == 0) return HSO_OK;
  // the kernels trust the tables: check every index a row carries before it reaches the device
  const int nk = (int)m->kfs.size();
  size_t need_pts = m->n_pts, need_obs = m->n_obs;
  for (int i = 0; i < n_obs; i++) {
    if (obs_ids[i] < 0) return hso_fail(ctx, HSO_E_INVALID, "seqmap_patch: negative observation id");
    need_obs = std::max(need_obs, (size_t)obs_ids[i] + 1);
  }
  for (int i = 0; i < n_obs; i++) {
    const hso_obs& o = obs[i];
    if (o.kf < 0 || o.kf >= nk || o.level < 0 || o.level >= HSO_N_PYR_LEVELS || o.pad_ < -1 || (o.pad_ >= 0 && (size_t)o.pad_ >= need_obs))
      return hso_fail(ctx, HSO_E_INVALID, "seqmap_patch: observation row out of range (keyframe table set first?)");
  }
  for (int i = 0; i < n_points; i++) {
    const hso_map_point& p = points[i];
    if (point_ids[i] < 0) return hso_fail(ctx, HSO_E_INVALID, "seqmap_patch: negative point id");
    need_pts = std::max(need_pts, (size_t)point_ids[i] + 1);
    if (p.host_kf < 0 || p.host_kf >= nk || p.obs_count < 0 || (p.obs_count > 0 && (p.obs_begin < 0 || (size_t)p.obs_begin >= need_obs)))
      return hso_fail(ctx, HSO_E_INVALID, "seqmap_patch: point row out of range");
  }
  HSO_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  if (int rc = seqmap_grow(ctx, &m->d_pts, &m->pts_cap, need_pts, m->n_pts)) return rc;
  if (int rc = seqmap_grow(ctx, &m->d_obs, &m->obs_cap, need_obs, m->n_obs)) return rc;
  auto al = [](size_t b) { return (b + 255) & ~size_t(255); };
  const size_t b_pid = al(sizeof(int) * (size_t)n_points), b_pts = al(sizeof(hso_map_point) * (size_t)n_points);
  const size_t b_oid = al(sizeof(int) * (size_t)n_obs), b_obs = al(sizeof(hso_obs) * (size_t)n_obs);
  const size_t need = b_pid + b_pts + b_oid + b_obs;
  if (ctx->batch_cap < need) {
    HSO_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->d_batch) (void)hipFree(ctx->d_batch);
    ctx->d_batch = nullptr; ctx->batch_cap = 0;
    HSO_HIP_CHECK(ctx, hipMalloc(reinterpret_cast<void**>(&ctx->d_batch), hso_grown(need)));
    ctx->batch_cap = hso_grown(need);
  }
  char* d = ctx->d_batch;
  if (n_points) {
    HSO_HIP_CHECK(ctx, hipMemcpyAsync(d, point_ids, sizeof(int) * (size_t)n_points, hipMemcpyHostToDevice, ctx->stream));
    HSO_HIP_CHECK(ctx, hipMemcpyAsync(d + b_pid, points, sizeof(hso_map_point) * (size_t)n_points, hipMemcpyHostToDevice, ctx->stream));
    const int G = sizeof(hso_map_point) / 8;
    hipLaunchKernelGGL(k_scatter_rows, dim3((unsigned)(((size_t)n_points * G + 255) / 256)), dim3(256), 0, ctx->stream,
                       reinterpret_cast<unsigned long long*>(m->d_pts), reinterpret_cast<const int*>(d), reinterpret_cast<const unsigned long long*>(d + b_pid), n_points, G);
  }
  if (n_obs) {
    char* e = d + b_pid + b_pts;
    HSO_HIP_CHECK(ctx, hipMemcpyAsync(e, obs_ids, sizeof(int) * (size_t)n_obs, hipMemcpyHostToDevice, ctx->stream));
    HSO_HIP_CHECK(ctx, hipMemcpyAsync(e + b_oid, obs, sizeof(hso_obs) * (size_t)n_obs, hipMemcpyHostToDevice, ctx->stream));
    const int G = sizeof(hso_obs) / 8;
    hipLaunchKernelGGL(k_scatter_rows, dim3((unsigned)(((size_t)n_obs * G + 255) / 256)), dim3(256), 0, ctx->stream,
                       reinterpret_cast<unsigned long long*>(m->d_obs), reinterpret_cast<const int*>(e), reinterpret_cast<const unsigned long long*>(e + b_oid), n_obs, G);
  }
  HSO_HIP_CHECK(ctx, hipGetLastError());
  // No synchronisation: the rows were copied out of the caller's memory when the copies were enqueued (pinned staging chunks of
  // the stream, hso_ctx.h), and whatever uses the work area or the tables next is ordered behind the scatter on the same stream.
  // A map patched for many sequences per step thus costs enqueues only; the step's next synchronising call releases the chunks.
  m->n_pts = need_pts; m->n_obs = need_obs;
  return HSO_OK;
}

int hso_gpu_seqmap_patch_multi(hso_gpu_ctx* ctx, const hso_seqmap_rows* patches, int n_patches)
{
  if (!ctx) return HSO_E_INVALID;
  if (n_patches < 0 || (n_patches > 0 && !patches)) return hso_fail(ctx, HSO_E_INVALID, "seqmap_patch_multi: bad argument");
  size_t tp = 0, to = 0;
  for (int i = 0; i < n_patches; i++) {
    const hso_seqmap_rows& P = patches[i];
    SeqMap* m = seqmap_of(ctx, P.map);
    if (!m || P.n_points < 0 || P.n_obs < 0 || (P.n_points > 0 && (!P.point_ids || !P.points)) || (P.n_obs > 0 && (!P.obs_ids || !P.obs)))
      return hso_fail(ctx, HSO_E_INVALID, "seqmap_patch_multi: bad patch");
    const int nk = (int)m->kfs.size();
    size_t need_pts = m->n_pts, need_obs = m->n_obs;
    for (int k = 0; k < P.n_obs; k++) { if (P.obs_ids[k] < 0) return hso_fail(ctx, HSO_E_INVALID, "seqmap_patch_multi: negative observation id"); need_obs = std::max(need_obs, (size_t)P.obs_ids[k] + 1); }
    for (int k = 0; k < P.n_obs; k++) {
      const hso_obs& o = P.obs[k];
      if (o.kf < 0 || o.kf >= nk || o.level < 0 || o.level >= HSO_N_PYR_LEVELS || o.pad_ < -1 || (o.pad_ >= 0 && (size_t)o.pad_ >= need_obs))
        return hso_fail(ctx, HSO_E_INVALID, "seqmap_patch_multi: observation row out of range (keyframe table set first?)");
    }
    for (int k = 0; k < P.n_points; k++) {
      const hso_map_point& p = P.points[k];
      if (P.point_ids[k] < 0) return hso_fail(ctx, HSO_E_INVALID, "seqmap_patch_multi: negative point id");
      need_pts = std::max(need_pts, (size_t)P.point_ids[k] + 1);
      if (p.host_kf < 0 || p.host_kf >= nk || p.obs_count < 0 || (p.obs_count > 0 && (p.obs_begin < 0 || (size_t)p.obs_begin >= need_obs)))
        return hso_fail(ctx, HSO_E_INVALID, "seqmap_patch_multi: point row out of range");
    }
    if (need_pts > m->pts_cap || need_obs > m->obs_cap) {
      HSO_HIP_CHECK(ctx, hipSetDevice(ctx->device));
      if (int rc = seqmap_grow(ctx, &m->d_pts, &m->pts_cap, need_pts, m->n_pts)) return rc;
      if (int rc = seqmap_grow(ctx, &m->d_obs, &m->obs_cap, need_obs, m->n_obs)) return rc;
    }
    m->n_pts = need_pts; m->n_obs = need_obs;
    tp += (size_t)P.n_points; to += (size_t)P.n_obs;
  }
  if (tp + to == 0) return HSO_OK;
  HSO_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  auto al = [](size_t b) { return (b + 255) & ~size_t(255); };
  // [destination row pointers of the points | of the observations | point rows | observation rows]
  const size_t b_dp = al(sizeof(void*) * tp), b_do = al(sizeof(void*) * to), b_p = al(sizeof(hso_map_point) * tp), b_o = al(sizeof(hso_obs) * to);
  const size_t need = b_dp + b_do + b_p + b_o;
  char* h = hso_pinned(ctx, 0, need);
  if (!h) return HSO_E_NOMEM;
  if (ctx->batch_cap < need) {
    HSO_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->d_batch) (void)hipFree(ctx->d_batch);
    ctx->d_batch = nullptr; ctx->batch_cap = 0;
    HSO_HIP_CHECK(ctx, hipMalloc(reinterpret_cast<void**>(&ctx->d_batch), hso_grown(need)));
    ctx->batch_cap = hso_grown(need);
  }
  hso_map_point** dp = reinterpret_cast<hso_map_point**>(h);
  hso_obs** dob = reinterpret_cast<hso_obs**>(h + b_dp);
  hso_map_point* rp = reinterpret_cast<hso_map_point*>(h + b_dp + b_do);
  hso_obs* ro = reinterpret_cast<hso_obs*>(h + b_dp + b_do + b_p);
  size_t ip = 0, io = 0;
  for (int i = 0; i < n_patches; i++) {
    const hso_seqmap_rows& P = patches[i];
    SeqMap* m = seqmap_of(ctx, P.map);
    for (int k = 0; k < P.n_points; k++) { dp[ip] = m->d_pts + P.point_ids[k]; rp[ip] = P.points[k]; ip++; }
    for (int k = 0; k < P.n_obs; k++) { dob[io] = m->d_obs + P.obs_ids[k]; ro[io] = P.obs[k]; io++; }
  }
  char* d = ctx->d_batch;
  // the staging buffer is rewritten by the next entry point: this one waits for its copy (a step patches once or twice)
  HSO_HIP_CHECK(ctx, hipMemcpyAsync(d, h, need, hipMemcpyHostToDevice, ctx->stream));
  if (tp) {
    const int G = sizeof(hso_map_point) / 8;
    hipLaunchKernelGGL(k_scatter_rows_to, dim3((unsigned)((tp * G + 255) / 256)), dim3(256), 0, ctx->stream,
                       reinterpret_cast<unsigned long long* const*>(d), reinterpret_cast<const unsigned long long*>(d + b_dp + b_do), (int)tp, G);
  }
  if (to) {
    const int G = sizeof(hso_obs) / 8;
    hipLaunchKernelGGL(k_scatter_rows_to, dim3((unsigned)((to * G + 255) / 256)), dim3(256), 0, ctx->stream,
                       reinterpret_cast<unsigned long long* const*>(d + b_dp), reinterpret_cast<const unsigned long long*>(d + b_dp + b_do + b_p), (int)to, G);
  }
  HSO_HIP_CHECK(ctx, hipGetLastError());
  HSO_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  return HSO_OK;
}

int hso_gpu_seqmap_size(hso_gpu_ctx* ctx, int map, int* n_kfs, int* n_points, int* n_obs)
{
  if (!ctx) return HSO_E_INVALID;
  SeqMap* m = seqmap_of(ctx, map);
  if (!m) return hso_fail(ctx, HSO_E_INVALID, "seqmap: no such map");
  if (n_kfs) *n_kfs = (int)m->kfs.size();
  if (n_points) *n_points = (int)m->n_pts;
  if (n_obs) *n_obs = (int)m->n_obs;
  return HSO_OK;
}

int hso_gpu_seqmap_read(hso_gpu_ctx* ctx, int map, const int32_t* point_ids, int n_points, hso_map_point* points_out,
                        const int32_t* obs_ids, int n_obs, hso_obs* obs_out)
{
  if (!ctx) return HSO_E_INVALID;
  SeqMap* m = seqmap_of(ctx, map);
  if (!m || n_points < 0 || n_obs < 0 || (n_points > 0 && (!point_ids || !points_out)) || (n_obs > 0 && (!obs_ids || !obs_out)))
    return hso_fail(ctx, HSO_E_INVALID, "seqmap_read: bad argument");
  for (int i = 0; i < n_points; i++) if (point_ids[i] < 0 || (size_t)point_ids[i] >= m->n_pts) return hso_fail(ctx, HSO_E_INVALID, "seqmap_read: point id out of range");
  for (int i = 0; i < n_obs; i++) if (obs_ids[i] < 0 || (size_t)obs_ids[i] >= m->n_obs) return hso_fail(ctx, HSO_E_INVALID, "seqmap_read: observation id out of range");
  if (n_points == 0 && n_obs == 0) return HSO_OK;
  HSO_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  auto al = [](size_t b) { return (b + 255) & ~size_t(255); };
  const size_t b_pid = al(sizeof(int) * (size_t)n_points), b_pts = al(sizeof(hso_map_point) * (size_t)n_points);
  const size_t b_oid = al(sizeof(int) * (size_t)n_obs), b_obs = al(sizeof(hso_obs) * (size_t)n_obs);
  const size_t need = b_pid + b_pts + b_oid + b_obs;
  if (ctx->batch_cap < need) {
    HSO_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->d_batch) (void)hipFree(ctx->d_batch);
    ctx->d_batch = nullptr; ctx->batch_cap = 0;
    HSO_HIP_CHECK(ctx, hipMalloc(reinterpret_cast<void**>(&ctx->d_batch), hso_grown(need)));
    ctx->batch_cap = hso_grown(need);
  }
  char* d = ctx->d_batch;
  if (n_points) {
    HSO_HIP_CHECK(ctx, hipMemcpyAsync(d, point_ids, sizeof(int) * (size_t)n_points, hipMemcpyHostToDevice, ctx->stream));
    const int G = sizeof(hso_map_point) / 8;
    hipLaunchKernelGGL(k_gather_rows, dim3((unsigned)(((size_t)n_points * G + 255) / 256)), dim3(256), 0, ctx->stream,
                       reinterpret_cast<const unsigned long long*>(m->d_pts), reinterpret_cast<const int*>(d), reinterpret_cast<unsigned long long*>(d + b_pid), n_points, G);
    HSO_HIP_CHECK(ctx, hipMemcpyAsync(points_out, d + b_pid, sizeof(hso_map_point) * (size_t)n_points, hipMemcpyDeviceToHost, ctx->stream));
  }
  if (n_obs) {
    char* e = d + b_pid + b_pts;
    HSO_HIP_CHECK(ctx, hipMemcpyAsync(e, obs_ids, sizeof(int) * (size_t)n_obs, hipMemcpyHostToDevice, ctx->stream));
    const int G = sizeof(hso_obs) / 8;
    hipLaunchKernelGGL(k_gather_rows, dim3((unsigned)(((size_t)n_obs * G + 255) / 256)), dim3(256), 0, ctx->stream,
                       reinterpret_cast<const unsigned long long*>(m->d_obs), reinterpret_cast<const int*>(e), reinterpret_cast<unsigned long long*>(e + b_oid), n_obs, G);
    HSO_HIP_CHECK(ctx, hipMemcpyAsync(obs_out, e + b_oid, sizeof(hso_obs) * (size_t)n_obs, hipMemcpyDeviceToHost, ctx->stream));
  }
  HSO_HIP_CHECK(ctx, hipGetLastError());
  HSO_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  return HSO_OK;
}

int hso_gpu_debug_fetch(hso_gpu_ctx* ctx, int what, void* out, size_t bytes)
{
  if (!ctx) return HSO_E_INVALID;
  SeqMaps* S = ctx->seqmaps;
  if (!S || what < 0 || what > 4 || !out || !S->dbg[what] || S->dbg_bytes[what] != bytes) return hso_fail(ctx, HSO_E_INVALID, "debug_fetch: no such table, or a size mismatch");
  if (bytes == 0) return HSO_OK;
  HSO_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  HSO_HIP_CHECK(ctx, hipMemcpyAsync(out, S->dbg[what], bytes, hipMemcpyDeviceToHost, ctx->stream));
  HSO_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  return HSO_OK;
}

}  // extern "C"

void hso_seqmaps_debug_set(hso_gpu_ctx* ctx, int what, const void* d, size_t bytes)
{
  if (!ctx->seqmaps) return;
  ctx->seqmaps->dbg[what] = d; ctx->seqmaps->dbg_bytes[what] = bytes;
}

// one frame of a sequence map: the listed points
struct ListCallDev {
  ReprojFrameDev F;                // kf_begin = first row of the call's ReprojKf block
  const hso_map_point* pts; const hso_obs* obs;
  int n_pts_table, n_kfs;
  int list_begin, list_count;      // the call's slice of the id / quality arrays
};
struct ListConsts {
  hso_camera cam;
  const ListCallDev* calls;
  int n_calls, n_total;
  const ReprojKf* kfs;
  const int32_t* ids; const uint8_t* quality;
  int cell_size, grid_n_cols;
};

__global__ __launch_bounds__(256) void k_reproject_list(ListConsts M, AlignJobDev* jobs, hso_reproj_point* proj)
{
  const int g = blockIdx.x * 256 + threadIdx.x;
  if (g >= M.n_total) return;
  int lo = 0, hi = M.n_calls - 1;          // the call whose slice holds g
  while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (M.calls[mid].list_begin <= g) lo = mid; else hi = mid - 1; }
  const ListCallDev& C = M.calls[lo];
  const int pid = M.ids[g];
  hso_reproj_point r;
  if (pid < 0 || pid >= C.n_pts_table) {   // an id the table does not hold: not projected (the host checked the tables, not the list)
    r.projected = 0; r.cell = 0; r.px[0] = 0; r.px[1] = 0; r.ref_obs = -1;
    jobs[g].ref_base = nullptr; jobs[g].cur_base = C.F.cur_base;
  } else {
    r = reproject_one<true>(M.cam, C.pts[pid], C.F, M.kfs + C.F.kf_begin, C.obs, M.cell_size, M.grid_n_cols, &jobs[g]);
  }
  r.pad_ = M.quality[g];
  proj[g] = r;
}

// The launch chain of hso_gpu_reproject_select_pose_frames up to the per-point records (cf. hso_reproject_maps_run)
int hso_reproject_frames_run(hso_gpu_ctx* ctx, const hso_camera* cam, const hso_map_frame* frames, int n_frames, int cell_size,
                             int grid_n_cols, size_t extra_bytes, HsoMapsRun* R, HsoFramesAux* X)
{
  SeqMaps* S = ctx->seqmaps;
  if (!S || !cam || n_frames < 0 || (n_frames > 0 && !frames) || cell_size < 1 || grid_n_cols < 1)
    return hso_fail(ctx, HSO_E_INVALID, "reproject_select_pose_frames: bad argument");
  R->n = 0; R->begin.assign(n_frames + 1, 0);
  if (n_frames == 0) return 0;
  HSO_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  size_t total = 0, total_kfs = 0;
  X->kf_begin.assign(n_frames + 1, 0);
  for (int c = 0; c < n_frames; c++) {
    const SeqMap* m = seqmap_of(ctx, frames[c].map);
    if (!m) return hso_fail(ctx, HSO_E_INVALID, "reproject_select_pose_frames: no such map");
    if (frames[c].n_points < 0 || (frames[c].n_points > 0 && (!frames[c].point_ids || !frames[c].quality)))
      return hso_fail(ctx, HSO_E_INVALID, "reproject_select_pose_frames: null point list");
    total += (size_t)frames[c].n_points; total_kfs += m->kfs.size();
    R->begin[c + 1] = (int)total; X->kf_begin[c + 1] = (int)total_kfs;
  }
  if (total == 0) return 0;
  if (!S->have_g || cam->width != S->g.w[0] || cam->height != S->g.h[0]) return hso_fail(ctx, HSO_E_INVALID, "reproject_select_pose_frames: camera size differs from the frame size");
  auto al = [](size_t b) { return (b + 255) & ~size_t(255); };
  const size_t b_calls = al(sizeof(ListCallDev) * (size_t)n_frames), b_kfs = al(sizeof(ReprojKf) * std::max(total_kfs, (size_t)1));
  const size_t b_ids = al(sizeof(int32_t) * total), b_q = al(total);
  char* hin = hso_pinned(ctx, 0, b_calls + b_kfs + b_ids + b_q);
  if (!hin) return HSO_E_NOMEM;
  ListCallDev* hc = reinterpret_cast<ListCallDev*>(hin);
  ReprojKf* hk = reinterpret_cast<ReprojKf*>(hin + b_calls);
  int32_t* hid = reinterpret_cast<int32_t*>(hin + b_calls + b_kfs);
  uint8_t* hq = reinterpret_cast<uint8_t*>(hin + b_calls + b_kfs + b_ids);
  for (int c = 0; c < n_frames; c++) {
    const hso_map_frame& K = frames[c];
    const SeqMap* m = seqmap_of(ctx, K.map);
    auto itc = ctx->frames.find(K.cur_frame_id);
    if (itc == ctx->frames.end()) return hso_fail(ctx, HSO_E_NOFRAME, "reproject_select_pose_frames: current frame not resident");
    if (!same_geom(itc->second.g, S->g)) return hso_fail(ctx, HSO_E_INVALID, "reproject_select_pose_frames: frames must share one size");
    const Se3 Tc = se3_from(K.T_cur_w);
    const Se3 ci = se3_inverse(Tc);
    ListCallDev& D = hc[c];
    D.F.cur_pos[0] = ci.tx; D.F.cur_pos[1] = ci.ty; D.F.cur_pos[2] = ci.tz;
    D.F.cur_base = itc->second.base; D.F.kf_begin = X->kf_begin[c]; D.F.pad_ = 0;
    D.pts = m->d_pts; D.obs = m->d_obs; D.n_pts_table = (int)m->n_pts; D.n_kfs = (int)m->kfs.size();
    D.list_begin = R->begin[c]; D.list_count = K.n_points;
    for (size_t k = 0; k < m->kfs.size(); k++) {
      auto it = ctx->frames.find(m->kfs[k].frame_id);
      if (it == ctx->frames.end()) return hso_fail(ctx, HSO_E_NOFRAME, "reproject_select_pose_frames: a keyframe of the map is no longer resident");
      ReprojKf& Q = hk[(size_t)X->kf_begin[c] + k];
      const Se3 inv = se3_inverse(se3_from(m->kfs[k].T_f_w));
      Q.T_cur_kf = se3_mul(Tc, inv);
      Q.pos[0] = inv.tx; Q.pos[1] = inv.ty; Q.pos[2] = inv.tz;
      Q.base = it->second.base; Q.frame_id = m->kfs[k].frame_id;
      Q.exposure_rat = (float)(K.cur_exposure_time / m->kfs[k].exposure_time);
      Q.kf_gap_lt4 = (K.cur_keyframe_id - m->kfs[k].keyframe_id) < 4;
    }
    if (K.n_points) {
      memcpy(hid + R->begin[c], K.point_ids, sizeof(int32_t) * (size_t)K.n_points);
      memcpy(hq + R->begin[c], K.quality, (size_t)K.n_points);
    }
  }
  const size_t o_jobs = 0, o_match = o_jobs + al(sizeof(AlignJobDev) * total), o_proj = o_match + al(sizeof(hso_align_out) * total);
  const size_t o_brief = o_proj + al(sizeof(hso_reproj_point) * total), o_in = o_brief + al(sizeof(hso_match_brief) * total);
  const size_t b_in = b_calls + b_kfs + b_ids + b_q;
  const size_t o_extra = o_in + al(b_in);
  const size_t need = o_extra + extra_bytes;
  if (ctx->batch_cap < need) {
    HSO_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->d_batch) (void)hipFree(ctx->d_batch);
    ctx->d_batch = nullptr; ctx->batch_cap = 0;
    HSO_HIP_CHECK(ctx, hipMalloc(reinterpret_cast<void**>(&ctx->d_batch), hso_grown(need)));
    ctx->batch_cap = hso_grown(need);
  }
  char* d = ctx->d_batch;
  HSO_HIP_CHECK(ctx, hipMemcpyAsync(d + o_in, hin, b_in, hipMemcpyHostToDevice, ctx->stream));
  HSO_HIP_CHECK(ctx, hipMemsetAsync(d + o_match, 0, sizeof(hso_align_out) * total, ctx->stream));
  ListConsts M;
  M.cam = *cam; M.calls = reinterpret_cast<const ListCallDev*>(d + o_in); M.n_calls = n_frames; M.n_total = (int)total;
  M.kfs = reinterpret_cast<const ReprojKf*>(d + o_in + b_calls);
  M.ids = reinterpret_cast<const int32_t*>(d + o_in + b_calls + b_kfs); M.quality = reinterpret_cast<const uint8_t*>(d + o_in + b_calls + b_kfs + b_ids);
  M.cell_size = cell_size; M.grid_n_cols = grid_n_cols;
  AlignJobDev* d_jobs = reinterpret_cast<AlignJobDev*>(d + o_jobs);
  hso_align_out* d_match = reinterpret_cast<hso_align_out*>(d + o_match);
  hso_reproj_point* d_proj = reinterpret_cast<hso_reproj_point*>(d + o_proj);
  hso_match_brief* d_brief = reinterpret_cast<hso_match_brief*>(d + o_brief);
  const int n = (int)total;
  hipLaunchKernelGGL(k_reproject_list, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, M, d_jobs, d_proj);
  AlignConsts C;
  C.cam = *cam; C.g = S->g;
  launch_align(ctx, true, C, d_jobs, n, d_match);
  hipLaunchKernelGGL(k_match_brief, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, n, d_proj, d_match, d_jobs, d_brief);
  HSO_HIP_CHECK(ctx, hipGetLastError());
  R->n = n; R->d_proj = d_proj; R->d_brief = d_brief; R->d_extra = d + o_extra;
  X->d_ids = M.ids; X->d_quality = M.quality; X->d_match = d_match;
  X->pts.resize(n_frames); X->kf_poses.clear();
  for (int c = 0; c < n_frames; c++) {
    const SeqMap* m = seqmap_of(ctx, frames[c].map);
    X->pts[c] = m->d_pts;
    for (const hso_kf& k : m->kfs) X->kf_poses.push_back(k.T_f_w);
  }
  return n;
}
