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
#include "hso_align_dev.h"
#include "hso_pose_dev.h"
#include <stddef.h>
#include <algorithm>
#include <string.h>
#include <vector>

using namespace hso_dev;

#define ALIGN_WAVES_PER_BLOCK 4


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
    const hso_align_out o = match_patch(C.g, JD.cur_base, JD.ref_base, JD.j, s_geom[wave][q], (double)C.ncc_thresh, s_pwb[wave][row]);  // checkNCC(…, 0.7), :364 (0.8 for seeds, :509)
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

int hso_align_launch_sparse(hso_gpu_ctx* ctx, const hso_camera* cam, const PyrGeom& g, float ncc_thresh, const AlignJobDev* d_jobs, int n, hso_align_out* d_out)
{
  if (n <= 0) return HSO_OK;
  AlignConsts C;
  C.cam = *cam; C.g = g; C.ncc_thresh = ncc_thresh;
  launch_align(ctx, true, C, d_jobs, n, d_out);
  HSO_HIP_CHECK(ctx, hipGetLastError());
  return HSO_OK;
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
// Sequence maps (include/hso_gpu.h: hso_gpu_seqmap_*): the tables of a whole sequence, indexed by the caller's own point / feature
// ids and patched row by row — points (with their state words), observations (= keyframe features) with their Feature::point
// links, the keyframe table with Frame::key_pts_, the keyframes' feature lists (Frame::fts_), the candidate list, and the feature
// table of the sequence's newest frame.  The resident per-frame chain (hso_gpu_seq_chain, hso_select.hip) walks them on the device.
#define SEQ_FIRST_UNSET 0x7f7f7f7f
struct SeqMap {
  hso_map_point* d_pts = nullptr; size_t pts_cap = 0, n_pts = 0;
  int32_t* d_first = nullptr; size_t first_cap = 0;          // per point row: the listing's stamp scratch (SEQ_FIRST_UNSET between calls)
  hso_obs* d_obs = nullptr; size_t obs_cap = 0, n_obs = 0;
  int32_t* d_obs_pt = nullptr; size_t obs_pt_cap = 0;        // Feature::point per observation row, -1 none
  std::vector<hso_kf> kfs;
  std::vector<int32_t> key_points;                            // 5 per keyframe row
  SeqKfDev* d_kfs = nullptr; size_t kfs_cap = 0;
  bool kfs_stale = false;                                     // the device copy waits for seqmap_flush_kfs
  int fts_cap = 0;
  int32_t* d_kf_fts = nullptr; size_t kf_rows_cap = 0;       // [rows][fts_cap]
  std::vector<int32_t> kf_nfts;                               // length of every keyframe's list
  int32_t* d_cands = nullptr; size_t cands_cap = 0; int n_cands = 0;
  hso_seq_feature* d_ff[2] = {nullptr, nullptr}; int ff_cap = 0;
  int64_t ff_frame[2] = {-1, -1}; int ff_n[2] = {0, 0}; int ff_newest = 0;
};
struct SeqMaps {
  std::vector<SeqMap*> m;
  PyrGeom g{}; bool have_g = false;
  std::vector<int> stale_kfs;                                 // maps whose keyframe table changed since the last flush
  char* d_kfup = nullptr; size_t kfup_cap = 0;                // [rows | destination per row] of a flush
  // where the last hso_gpu_seq_chain call left its tables (hso_gpu_debug_fetch)
  const void* dbg[HSO_DBG_N] = {nullptr}; size_t dbg_bytes[HSO_DBG_N] = {0};
};

static void seqmap_release(SeqMap* m)
{
  (void)hipFree(m->d_pts); (void)hipFree(m->d_first); (void)hipFree(m->d_obs); (void)hipFree(m->d_obs_pt); (void)hipFree(m->d_kfs);
  (void)hipFree(m->d_kf_fts); (void)hipFree(m->d_cands); (void)hipFree(m->d_ff[0]); (void)hipFree(m->d_ff[1]);
  delete m;
}

void hso_seqmaps_free(hso_gpu_ctx* ctx)
{
  if (!ctx->seqmaps) return;
  for (SeqMap* m : ctx->seqmaps->m) if (m) seqmap_release(m);
  (void)hipFree(ctx->seqmaps->d_kfup);
  delete ctx->seqmaps;
  ctx->seqmaps = nullptr;
}

PyrGeom hso_seqmaps_geom(hso_gpu_ctx* ctx, bool* have)
{
  *have = ctx->seqmaps && ctx->seqmaps->have_g;
  return *have ? ctx->seqmaps->g : PyrGeom{};
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
// point rows: the quality key and the bad flag of a patched row are the caller's (it mirrors the kinds from the chain's events); a
// counter is the caller's too unless the word says to keep the device's (HSO_PT_KEEP_NFAIL / HSO_PT_KEEP_NOK): the device counts the
// reprojection failures / successes, the caller resets them (a new point, a promotion, a retired temporary point)
HSO_DEV void point_row_store(hso_map_point* dst, const hso_map_point& src)
{
  uint32_t w = (uint32_t)src.pad_;
  const uint32_t old = (uint32_t)dst->pad_;
  if (w & HSO_PT_KEEP_NFAIL) w = (w & ~(0x3ffu << 8)) | (old & (0x3ffu << 8));
  if (w & HSO_PT_KEEP_NOK) w = (w & ~(0x7ffu << 20)) | (old & (0x7ffu << 20));
  w &= ~(HSO_PT_KEEP_NFAIL | HSO_PT_KEEP_NOK);
  hso_map_point r = src;
  r.pad_ = (int32_t)w;
  *dst = r;
}
static __global__ void k_scatter_points(hso_map_point* __restrict__ dst, const int* __restrict__ ids, const hso_map_point* __restrict__ src, int n)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) point_row_store(dst + ids[i], src[i]);
}
static __global__ void k_scatter_points_to(hso_map_point* const* __restrict__ dst, const hso_map_point* __restrict__ src, int n)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) point_row_store(dst[i], src[i]);
}
static __global__ void k_scatter_ints_to(int32_t* const* __restrict__ dst, const int32_t* __restrict__ src, int n)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) *dst[i] = src[i];
}
static __global__ void k_scatter_links(int32_t* __restrict__ dst, const int32_t* __restrict__ ids, const int32_t* __restrict__ src, int n)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[ids[i]] = src[i];
}
// inclusive scan of v over a workgroup of W wavefronts; *total = the sum (all threads must call)
template <int W> __device__ int sel_scan_waves(int v, int* s_wave, int& total)
{
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int incl = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_up(incl, d); if (lane >= d) incl += o; }
  __syncthreads();
  if (lane == 63) s_wave[wave] = incl;
  __syncthreads();
  int base = 0; total = 0;
#pragma unroll
  for (int w = 0; w < W; w++) { if (w < wave) base += s_wave[w]; total += s_wave[w]; }
  return incl + base;
}
// inclusive scan of v over a 256-thread workgroup; *total = the sum (all threads must call)
__device__ int sel_scan256(int v, int* s_wave, int& total)
{
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int incl = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_up(incl, d); if (lane >= d) incl += o; }
  __syncthreads();
  if (lane == 63) s_wave[wave] = incl;
  __syncthreads();
  int base = 0; total = 0;
  for (int w = 0; w < 4; w++) { if (w < wave) base += s_wave[w]; total += s_wave[w]; }
  return incl + base;
}
static_assert(sizeof(hso_map_point) % 8 == 0 && sizeof(hso_obs) % 8 == 0, "rows move in 8-byte granules");

// grow-only device array: new memory is filled with `fill` bytes, the first `keep` elements are carried over
template <typename T> static int seqmap_grow(hso_gpu_ctx* ctx, T** p, size_t* cap, size_t need, size_t keep, int fill = 0, size_t min_cap = 65536)
{
  if (*cap >= need) return HSO_OK;
  // a growth = allocation + fill + copy + a wait for the stream: with one map per sequence and hundreds of sequences growing in step
  // with their keyframes, small increments meant ten growths per step of 128 sequences.  So: room for a dozen keyframes' rows at
  // once (a few MB per map of 288 GB), doubling after that.
  const size_t ncap = std::max(need * 2, min_cap);
  T* q = nullptr;
  HSO_HIP_CHECK(ctx, hipMalloc(reinterpret_cast<void**>(&q), ncap * sizeof(T)));
  hipError_t e = hipMemsetAsync(q, fill, ncap * sizeof(T), ctx->stream);
  if (e == hipSuccess && *p && keep) e = hipMemcpyAsync(q, *p, keep * sizeof(T), hipMemcpyDeviceToDevice, ctx->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  if (e != hipSuccess) { (void)hipFree(q); ctx->err = std::string("seqmap: ") + hipGetErrorString(e); return HSO_E_HIP; }
  if (*p) (void)hipFree(*p);
  *p = q; *cap = ncap;
  return HSO_OK;
}

// the tables that grow in step with the point / observation tables
static int seqmap_grow_tables(hso_gpu_ctx* ctx, SeqMap* m, size_t need_pts, size_t need_obs)
{
  if (need_pts > m->pts_cap || need_obs > m->obs_cap || need_pts > m->first_cap || need_obs > m->obs_pt_cap) HSO_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  if (int rc = seqmap_grow(ctx, &m->d_pts, &m->pts_cap, need_pts, m->n_pts)) return rc;
  if (int rc = seqmap_grow(ctx, &m->d_first, &m->first_cap, m->pts_cap, 0, 0x7f)) return rc;          // never holds state between calls
  if (int rc = seqmap_grow(ctx, &m->d_obs, &m->obs_cap, need_obs, m->n_obs)) return rc;
  if (int rc = seqmap_grow(ctx, &m->d_obs_pt, &m->obs_pt_cap, m->obs_cap, m->n_obs, 0xff)) return rc;  // -1: no point
  return HSO_OK;
}

static int seqmap_work_area(hso_gpu_ctx* ctx, size_t need)
{
  if (ctx->batch_cap >= need) return HSO_OK;
  HSO_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  if (ctx->d_batch) (void)hipFree(ctx->d_batch);
  ctx->d_batch = nullptr; ctx->batch_cap = 0;
  HSO_HIP_CHECK(ctx, hipMalloc(reinterpret_cast<void**>(&ctx->d_batch), hso_grown(need)));
  ctx->batch_cap = hso_grown(need);
  return HSO_OK;
}

// the device copy of the keyframe table (+ base pointers, key points): rebuilt whenever either half changes (keyframe rate).  The
// change is only noted here; the rows of ALL the maps that changed go up together, one copy and one scatter launch, when the chain
// next runs (hso_seqmap_flush_kfs) — a copy per map was 10-20 copies per step of 128 sequences.
static int seqmap_upload_kfs(hso_gpu_ctx* ctx, SeqMap* m, int map)
{
  const size_t n = m->kfs.size();
  if (n == 0) return HSO_OK;
  if (m->kfs_cap < n) {
    HSO_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    if (int rc = seqmap_grow(ctx, &m->d_kfs, &m->kfs_cap, n, 0, 0, 256)) return rc;
  }
  if (!m->kfs_stale) { m->kfs_stale = true; ctx->seqmaps->stale_kfs.push_back(map); }
  return HSO_OK;
}
static_assert(sizeof(SeqKfDev) % 8 == 0, "keyframe rows move in 8-byte granules");

int hso_seqmap_flush_kfs(hso_gpu_ctx* ctx)
{
  SeqMaps* S = ctx->seqmaps;
  if (!S || S->stale_kfs.empty()) return HSO_OK;
  size_t n_rows = 0;
  for (int id : S->stale_kfs)
    if (SeqMap* m = seqmap_of(ctx, id))
      if (m->kfs_stale) {
        // nothing is marked clean before every keyframe of every stale map is known to be resident: a failing call leaves the list as it is
        for (const hso_kf& k : m->kfs) if (!ctx->frames.count(k.frame_id)) return hso_fail(ctx, HSO_E_NOFRAME, "seqmap: keyframe not resident");
        n_rows += m->kfs.size();
      }
  std::vector<SeqKfDev> rows; rows.reserve(n_rows);
  std::vector<unsigned long long*> dst; dst.reserve(n_rows);
  for (int id : S->stale_kfs) {
    SeqMap* m = seqmap_of(ctx, id);
    if (!m || !m->kfs_stale) continue;                            // destroyed since
    for (size_t k = 0; k < m->kfs.size(); k++) {
      auto it = ctx->frames.find(m->kfs[k].frame_id);
      if (it == ctx->frames.end()) return hso_fail(ctx, HSO_E_NOFRAME, "seqmap: keyframe not resident");   // checked above
      SeqKfDev r{};
      r.T_f_w = m->kfs[k].T_f_w; r.exposure_time = m->kfs[k].exposure_time; r.base = it->second.base; r.frame_id = m->kfs[k].frame_id;
      r.keyframe_id = m->kfs[k].keyframe_id;
      for (int q = 0; q < 5; q++) r.key_point[q] = 5 * k + q < m->key_points.size() ? m->key_points[5 * k + q] : -1;
      rows.push_back(r); dst.push_back(reinterpret_cast<unsigned long long*>(m->d_kfs + k));
    }
  }
  auto mark_clean = [&] { for (int id : S->stale_kfs) if (SeqMap* m = seqmap_of(ctx, id)) m->kfs_stale = false; S->stale_kfs.clear(); };
  if (rows.empty()) { mark_clean(); return HSO_OK; }
  HSO_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  const size_t row_bytes = sizeof(SeqKfDev) * rows.size(), need = row_bytes + sizeof(void*) * rows.size();
  if (S->kfup_cap < need) {
    HSO_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    if (S->d_kfup) (void)hipFree(S->d_kfup);
    S->d_kfup = nullptr; S->kfup_cap = 0;
    HSO_HIP_CHECK(ctx, hipMalloc(reinterpret_cast<void**>(&S->d_kfup), hso_grown(need)));
    S->kfup_cap = hso_grown(need);
  }
  // staged by the copy wrapper: the vectors may go when this returns
  std::vector<char> image(need);
  memcpy(image.data(), rows.data(), row_bytes); memcpy(image.data() + row_bytes, dst.data(), sizeof(void*) * rows.size());
  HSO_HIP_CHECK(ctx, hipMemcpyAsync(S->d_kfup, image.data(), need, hipMemcpyHostToDevice, ctx->stream));
  const int granules = (int)(sizeof(SeqKfDev) / 8);
  const size_t total = rows.size() * (size_t)granules;
  hipLaunchKernelGGL(k_scatter_rows_to, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream,
                     reinterpret_cast<unsigned long long* const*>(S->d_kfup + row_bytes), reinterpret_cast<const unsigned long long*>(S->d_kfup), (int)rows.size(), granules);
  HSO_HIP_CHECK(ctx, hipGetLastError());
  mark_clean();                                                   // only now: a failing exit above leaves every map on the list
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
  seqmap_release(m);
  ctx->seqmaps->m[map] = nullptr;
  return HSO_OK;
}

int hso_gpu_seqmap_configure(hso_gpu_ctx* ctx, int map, int fts_cap)
{
  if (!ctx) return HSO_E_INVALID;
  SeqMap* m = seqmap_of(ctx, map);
  if (!m || fts_cap < 1) return hso_fail(ctx, HSO_E_INVALID, "seqmap_configure: bad argument");
  if (m->d_kf_fts && fts_cap != m->fts_cap) return hso_fail(ctx, HSO_E_INVALID, "seqmap_configure: the keyframe lists are already in use");
  m->fts_cap = fts_cap;
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
  m->key_points.resize(5 * (size_t)n_kfs, -1);
  m->kf_nfts.resize((size_t)n_kfs, 0);
  return seqmap_upload_kfs(ctx, m, map);
}

int hso_gpu_seqmap_set_key_points(hso_gpu_ctx* ctx, int map, const int32_t* key_points, int n_kfs)
{
  if (!ctx) return HSO_E_INVALID;
  SeqMap* m = seqmap_of(ctx, map);
  if (!m || n_kfs < 0 || (size_t)n_kfs != m->kfs.size() || (n_kfs > 0 && !key_points)) return hso_fail(ctx, HSO_E_INVALID, "seqmap_set_key_points: one row of five per keyframe of the table");
  for (int i = 0; i < 5 * n_kfs; i++) if (key_points[i] < -1 || (key_points[i] >= 0 && (size_t)key_points[i] >= m->n_pts)) return hso_fail(ctx, HSO_E_INVALID, "seqmap_set_key_points: point row out of range");
  m->key_points.assign(key_points, key_points + 5 * (size_t)n_kfs);
  return seqmap_upload_kfs(ctx, m, map);
}

int hso_gpu_seqmap_patch(hso_gpu_ctx* ctx, int map, const int32_t* point_ids, const hso_map_point* points, int n_points,
                         const int32_t* obs_ids, const hso_obs* obs, int n_obs)
{
  hso_seqmap_rows r{};
  r.map = map; r.n_points = n_points; r.n_obs = n_obs; r.point_ids = point_ids; r.points = points; r.obs_ids = obs_ids; r.obs = obs;
  return hso_gpu_seqmap_patch_multi(ctx, &r, 1);
}

int hso_gpu_seqmap_patch_multi(hso_gpu_ctx* ctx, const hso_seqmap_rows* patches, int n_patches)
{
  if (!ctx) return HSO_E_INVALID;
  if (n_patches < 0 || (n_patches > 0 && !patches)) return hso_fail(ctx, HSO_E_INVALID, "seqmap_patch_multi: bad argument");
  size_t tp = 0, to = 0, rows_total = 0;
  for (int i = 0; i < n_patches; i++) {
    const hso_seqmap_rows& P = patches[i];
    SeqMap* m = seqmap_of(ctx, P.map);
    if (!m || P.n_points < 0 || P.n_obs < 0 || (P.n_points > 0 && (!P.point_ids || !P.points)) || (P.n_obs > 0 && (!P.obs_ids || !P.obs)))
      return hso_fail(ctx, HSO_E_INVALID, "seqmap_patch: bad patch");
    rows_total += (size_t)P.n_points + (size_t)P.n_obs;
    for (int q = 0; q < i; q++) if (patches[q].map == P.map) return hso_fail(ctx, HSO_E_INVALID, "seqmap_patch_multi: a map appears twice in one call");
  }
  // the kernels trust the tables: check every index a row carries before it reaches the device (per patch, on a few threads when
  // a keyframe step sends tens of thousands of rows)
  std::vector<const char*> bad((size_t)n_patches, nullptr);
  std::vector<size_t> need_p((size_t)n_patches), need_o((size_t)n_patches);
  hso_host_parallel(ctx, n_patches, rows_total * 64, [&](int i) {
    const hso_seqmap_rows& P = patches[i];
    const SeqMap* m = seqmap_of(ctx, P.map);
    const int nk = (int)m->kfs.size();
    size_t need_pts = m->n_pts, need_obs = m->n_obs;
    for (int k = 0; k < P.n_obs; k++) { if (P.obs_ids[k] < 0) { bad[(size_t)i] = "seqmap_patch: negative observation id"; return; } need_obs = std::max(need_obs, (size_t)P.obs_ids[k] + 1); }
    for (int k = 0; k < P.n_points; k++) { if (P.point_ids[k] < 0) { bad[(size_t)i] = "seqmap_patch: negative point id"; return; } need_pts = std::max(need_pts, (size_t)P.point_ids[k] + 1); }
    for (int k = 0; k < P.n_obs; k++) {
      const hso_obs& o = P.obs[k];
      if (o.kf < 0 || o.kf >= nk || o.level < 0 || o.level >= HSO_N_PYR_LEVELS || o.pad_ < -1 || (o.pad_ >= 0 && (size_t)o.pad_ >= need_obs)) {
        bad[(size_t)i] = "seqmap_patch: observation row out of range (keyframe table set first?)"; return;
      }
      if (P.obs_point && (P.obs_point[k] < -1 || (P.obs_point[k] >= 0 && (size_t)P.obs_point[k] >= need_pts))) { bad[(size_t)i] = "seqmap_patch: Feature::point link out of range"; return; }
    }
    for (int k = 0; k < P.n_points; k++) {
      const hso_map_point& p = P.points[k];
      if (p.host_kf < 0 || p.host_kf >= nk || p.obs_count < 0 || (p.obs_count > 0 && (p.obs_begin < 0 || (size_t)p.obs_begin >= need_obs))) { bad[(size_t)i] = "seqmap_patch: point row out of range"; return; }
    }
    need_p[(size_t)i] = need_pts; need_o[(size_t)i] = need_obs;
  });
  for (int i = 0; i < n_patches; i++) if (bad[(size_t)i]) return hso_fail(ctx, HSO_E_INVALID, bad[(size_t)i]);
  std::vector<size_t> at_p((size_t)n_patches + 1, 0), at_o((size_t)n_patches + 1, 0), at_l((size_t)n_patches + 1, 0);
  for (int i = 0; i < n_patches; i++) {
    const hso_seqmap_rows& P = patches[i];
    SeqMap* m = seqmap_of(ctx, P.map);
    if (int rc = seqmap_grow_tables(ctx, m, need_p[(size_t)i], need_o[(size_t)i])) return rc;
    m->n_pts = need_p[(size_t)i]; m->n_obs = need_o[(size_t)i];
    at_p[(size_t)i + 1] = at_p[(size_t)i] + (size_t)P.n_points; at_o[(size_t)i + 1] = at_o[(size_t)i] + (size_t)P.n_obs;
    at_l[(size_t)i + 1] = at_l[(size_t)i] + (P.obs_point ? (size_t)P.n_obs : 0);
  }
  tp = at_p.back(); to = at_o.back();
  if (tp + to == 0) return HSO_OK;
  HSO_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  auto al = [](size_t b) { return (b + 255) & ~size_t(255); };
  // [destination row pointers of the points | of the observations | of the links | point rows | observation rows | links]
  const size_t tl = at_l.back();
  const size_t b_dp = al(sizeof(void*) * tp), b_do = al(sizeof(void*) * to), b_dl = al(sizeof(void*) * tl);
  const size_t b_p = al(sizeof(hso_map_point) * tp), b_o = al(sizeof(hso_obs) * to), b_l = al(sizeof(int32_t) * tl);
  const size_t need = b_dp + b_do + b_dl + b_p + b_o + b_l;
  // the image is assembled in the stream's page-locked staging chunks and leaves with one DMA: no wait for the copy here — a step
  // patches before its chain call, whose own synchronisation releases the chunks
  if (int rc = seqmap_work_area(ctx, need)) return rc;
  char* h = hso_stage_reserve(ctx->stream, need);
  if (!h) return hso_fail(ctx, HSO_E_NOMEM, "seqmap_patch: no staging memory");
  hso_map_point** dp = reinterpret_cast<hso_map_point**>(h);
  hso_obs** dob = reinterpret_cast<hso_obs**>(h + b_dp);
  int32_t** dl = reinterpret_cast<int32_t**>(h + b_dp + b_do);
  hso_map_point* rp = reinterpret_cast<hso_map_point*>(h + b_dp + b_do + b_dl);
  hso_obs* ro = reinterpret_cast<hso_obs*>(h + b_dp + b_do + b_dl + b_p);
  int32_t* rl = reinterpret_cast<int32_t*>(h + b_dp + b_do + b_dl + b_p + b_o);
  hso_host_parallel(ctx, n_patches, need, [&](int i) {
    const hso_seqmap_rows& P = patches[i];
    const SeqMap* m = seqmap_of(ctx, P.map);
    size_t ip = at_p[(size_t)i], io = at_o[(size_t)i], il = at_l[(size_t)i];
    for (int k = 0; k < P.n_points; k++) { dp[ip] = m->d_pts + P.point_ids[k]; rp[ip] = P.points[k]; ip++; }
    for (int k = 0; k < P.n_obs; k++) {
      dob[io] = m->d_obs + P.obs_ids[k]; ro[io] = P.obs[k]; io++;
      if (P.obs_point) { dl[il] = m->d_obs_pt + P.obs_ids[k]; rl[il] = P.obs_point[k]; il++; }
    }
  });
  char* d = ctx->d_batch;
  HSO_HIP_CHECK(ctx, hipMemcpyAsync(d, h, need, hipMemcpyHostToDevice, ctx->stream));
  if (tp) hipLaunchKernelGGL(k_scatter_points_to, dim3((unsigned)((tp + 255) / 256)), dim3(256), 0, ctx->stream,
                             reinterpret_cast<hso_map_point* const*>(d), reinterpret_cast<const hso_map_point*>(d + b_dp + b_do + b_dl), (int)tp);
  if (to) {
    const int G = sizeof(hso_obs) / 8;
    hipLaunchKernelGGL(k_scatter_rows_to, dim3((unsigned)((to * G + 255) / 256)), dim3(256), 0, ctx->stream,
                       reinterpret_cast<unsigned long long* const*>(d + b_dp), reinterpret_cast<const unsigned long long*>(d + b_dp + b_do + b_dl + b_p), (int)to, G);
  }
  if (tl) hipLaunchKernelGGL(k_scatter_ints_to, dim3((unsigned)((tl + 255) / 256)), dim3(256), 0, ctx->stream,
                             reinterpret_cast<int32_t* const*>(d + b_dp + b_do), reinterpret_cast<const int32_t*>(d + b_dp + b_do + b_dl + b_p + b_o), (int)tl);
  HSO_HIP_CHECK(ctx, hipGetLastError());
  return HSO_OK;
}

int hso_gpu_seqmap_patch_links(hso_gpu_ctx* ctx, int map, const int32_t* obs_ids, const int32_t* obs_point, int n_obs)
{
  if (!ctx) return HSO_E_INVALID;
  SeqMap* m = seqmap_of(ctx, map);
  if (!m || n_obs < 0 || (n_obs > 0 && (!obs_ids || !obs_point))) return hso_fail(ctx, HSO_E_INVALID, "seqmap_patch_links: bad argument");
  if (n_obs == 0) return HSO_OK;
  for (int i = 0; i < n_obs; i++)
    if (obs_ids[i] < 0 || (size_t)obs_ids[i] >= m->n_obs || obs_point[i] < -1 || (obs_point[i] >= 0 && (size_t)obs_point[i] >= m->n_pts))
      return hso_fail(ctx, HSO_E_INVALID, "seqmap_patch_links: row out of range");
  HSO_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  const size_t b_id = ((size_t)n_obs * sizeof(int32_t) + 255) & ~size_t(255);
  if (int rc = seqmap_work_area(ctx, 2 * b_id)) return rc;
  char* d = ctx->d_batch;
  HSO_HIP_CHECK(ctx, hipMemcpyAsync(d, obs_ids, sizeof(int32_t) * (size_t)n_obs, hipMemcpyHostToDevice, ctx->stream));
  HSO_HIP_CHECK(ctx, hipMemcpyAsync(d + b_id, obs_point, sizeof(int32_t) * (size_t)n_obs, hipMemcpyHostToDevice, ctx->stream));
  hipLaunchKernelGGL(k_scatter_links, dim3((unsigned)((n_obs + 255) / 256)), dim3(256), 0, ctx->stream, m->d_obs_pt, reinterpret_cast<const int32_t*>(d),
                     reinterpret_cast<const int32_t*>(d + b_id), n_obs);
  HSO_HIP_CHECK(ctx, hipGetLastError());
  return HSO_OK;
}

int hso_gpu_seqmap_patch_lists(hso_gpu_ctx* ctx, const hso_seqmap_list_patch* patches, int n_patches)
{
  if (!ctx) return HSO_E_INVALID;
  if (n_patches < 0 || (n_patches > 0 && !patches)) return hso_fail(ctx, HSO_E_INVALID, "seqmap_patch_lists: bad argument");
  size_t total = 0;
  for (int i = 0; i < n_patches; i++) {
    const hso_seqmap_list_patch& P = patches[i];
    SeqMap* m = seqmap_of(ctx, P.map);
    if (!m || P.first < 0 || P.n < 0 || (P.n > 0 && !P.ids)) return hso_fail(ctx, HSO_E_INVALID, "seqmap_patch_lists: bad patch");
    if (P.list == HSO_LIST_CANDIDATES) {
      if (P.first > m->n_cands) return hso_fail(ctx, HSO_E_INVALID, "seqmap_patch_lists: a patch of the candidate list must start inside it or at its end");
      for (int k = 0; k < P.n; k++) if (P.ids[k] < 0 || (size_t)P.ids[k] >= m->n_pts) return hso_fail(ctx, HSO_E_INVALID, "seqmap_patch_lists: candidate point row out of range");
      HSO_HIP_CHECK(ctx, hipSetDevice(ctx->device));
      if (int rc = seqmap_grow(ctx, &m->d_cands, &m->cands_cap, (size_t)P.first + (size_t)P.n, (size_t)m->n_cands, 0, 16384)) return rc;
    } else {
      if (P.list < 0 || (size_t)P.list >= m->kfs.size()) return hso_fail(ctx, HSO_E_INVALID, "seqmap_patch_lists: no such keyframe row");
      if (m->fts_cap < 1) return hso_fail(ctx, HSO_E_INVALID, "seqmap_patch_lists: hso_gpu_seqmap_configure first");
      if (P.first > m->kf_nfts[(size_t)P.list] || P.first + P.n > m->fts_cap) return hso_fail(ctx, HSO_E_INVALID, "seqmap_patch_lists: a keyframe's feature list outgrows fts_cap, or the patch leaves a gap");
      for (int k = 0; k < P.n; k++) if (P.ids[k] < 0 || (size_t)P.ids[k] >= m->n_obs) return hso_fail(ctx, HSO_E_INVALID, "seqmap_patch_lists: feature row out of range");
      if (m->kf_rows_cap < m->kfs.size()) {
        HSO_HIP_CHECK(ctx, hipSetDevice(ctx->device));
        size_t cap_el = m->kf_rows_cap * (size_t)m->fts_cap;
        const size_t rows = 2 * m->kfs.size() + 16;
        if (int rc = seqmap_grow(ctx, &m->d_kf_fts, &cap_el, rows * (size_t)m->fts_cap, cap_el)) return rc;
        m->kf_rows_cap = cap_el / (size_t)m->fts_cap;
      }
    }
    total += (size_t)P.n;
  }
  if (total > 0) {
    HSO_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    auto al = [](size_t b) { return (b + 255) & ~size_t(255); };
    const size_t b_d = al(sizeof(void*) * total), need = b_d + al(sizeof(int32_t) * total);
    std::vector<char> img(need);
    int32_t** dst = reinterpret_cast<int32_t**>(img.data());
    int32_t* val = reinterpret_cast<int32_t*>(img.data() + b_d);
    size_t at = 0;
    for (int i = 0; i < n_patches; i++) {
      const hso_seqmap_list_patch& P = patches[i];
      SeqMap* m = seqmap_of(ctx, P.map);
      int32_t* base = P.list == HSO_LIST_CANDIDATES ? m->d_cands : m->d_kf_fts + (size_t)P.list * (size_t)m->fts_cap;
      for (int k = 0; k < P.n; k++) { dst[at] = base + P.first + k; val[at] = P.ids[k]; at++; }
    }
    if (int rc = seqmap_work_area(ctx, need)) return rc;
    HSO_HIP_CHECK(ctx, hipMemcpyAsync(ctx->d_batch, img.data(), need, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(k_scatter_ints_to, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream,
                       reinterpret_cast<int32_t* const*>(ctx->d_batch), reinterpret_cast<const int32_t*>(ctx->d_batch + b_d), (int)total);
    HSO_HIP_CHECK(ctx, hipGetLastError());
  }
  for (int i = 0; i < n_patches; i++) {   // the lengths (host side: the chain call sends them along)
    const hso_seqmap_list_patch& P = patches[i];
    SeqMap* m = seqmap_of(ctx, P.map);
    if (P.list == HSO_LIST_CANDIDATES) m->n_cands = P.first + P.n; else m->kf_nfts[(size_t)P.list] = P.first + P.n;
  }
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
  if (int rc = seqmap_work_area(ctx, b_pid + b_pts + b_oid + b_obs)) return rc;
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

int hso_gpu_seq_frame_features(hso_gpu_ctx* ctx, const int32_t* maps, const int64_t* frame_ids, int n_maps, hso_seq_feature* out, int cap, int32_t* n_out)
{
  if (!ctx) return HSO_E_INVALID;
  if (n_maps < 0 || cap < 0 || (n_maps > 0 && (!maps || !frame_ids || !out || !n_out))) return hso_fail(ctx, HSO_E_INVALID, "seq_frame_features: bad argument");
  if (n_maps == 0) return HSO_OK;
  HSO_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  std::vector<HsoListCopy> lists;
  static_assert(sizeof(hso_seq_feature) % 4 == 0, "feature rows move in 4-byte units");
  for (int i = 0; i < n_maps; i++) {
    SeqMap* m = seqmap_of(ctx, maps[i]);
    if (!m) return hso_fail(ctx, HSO_E_INVALID, "seq_frame_features: no such map");
    const int b = m->ff_frame[0] == frame_ids[i] && m->d_ff[0] ? 0 : (m->ff_frame[1] == frame_ids[i] && m->d_ff[1] ? 1 : -1);
    if (b < 0) return hso_fail(ctx, HSO_E_NOFRAME, "seq_frame_features: the map holds no feature table of that frame");
    if (m->ff_n[b] > cap) return hso_fail(ctx, HSO_E_INVALID, "seq_frame_features: cap is smaller than the table");
    n_out[i] = m->ff_n[b];
    if (m->ff_n[b] > 0) lists.push_back({out + (size_t)i * cap, m->d_ff[b], sizeof(hso_seq_feature) * (size_t)m->ff_n[b]});
  }
  // the tables of all the maps in one DMA (a keyframe step of 128 sequences reads sixteen: it was one copy each)
  return hso_lists_to_host(ctx, lists);
}

int hso_gpu_seq_set_frame_features(hso_gpu_ctx* ctx, int map, int64_t frame_id, const hso_seq_feature* feats, int n)
{
  if (!ctx) return HSO_E_INVALID;
  SeqMap* m = seqmap_of(ctx, map);
  if (!m || n < 0 || (n > 0 && !feats)) return hso_fail(ctx, HSO_E_INVALID, "seq_set_frame_features: bad argument");
  for (int i = 0; i < n; i++) if (feats[i].point < -1 || (feats[i].point >= 0 && (size_t)feats[i].point >= m->n_pts)) return hso_fail(ctx, HSO_E_INVALID, "seq_set_frame_features: point row out of range");
  HSO_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  if (n > m->ff_cap) {
    HSO_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    const int ncap = n + n / 2 + 64;
    for (int b = 0; b < 2; b++) {
      hso_seq_feature* q = nullptr;
      HSO_HIP_CHECK(ctx, hipMalloc(reinterpret_cast<void**>(&q), sizeof(hso_seq_feature) * (size_t)ncap));
      if (m->d_ff[b] && m->ff_n[b] > 0) HSO_HIP_CHECK(ctx, hipMemcpy(q, m->d_ff[b], sizeof(hso_seq_feature) * (size_t)m->ff_n[b], hipMemcpyDeviceToDevice));
      (void)hipFree(m->d_ff[b]);
      m->d_ff[b] = q;
    }
    m->ff_cap = ncap;
  }
  // the table of that frame if the map holds one, else the older of the two
  int b = m->ff_frame[0] == frame_id ? 0 : (m->ff_frame[1] == frame_id ? 1 : 1 - m->ff_newest);
  if (n > 0) HSO_HIP_CHECK(ctx, hipMemcpyAsync(m->d_ff[b], feats, sizeof(hso_seq_feature) * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
  m->ff_frame[b] = frame_id; m->ff_n[b] = n; m->ff_newest = b;
  return HSO_OK;
}

int hso_gpu_debug_fetch(hso_gpu_ctx* ctx, int what, void* out, size_t bytes)
{
  if (!ctx) return HSO_E_INVALID;
  SeqMaps* S = ctx->seqmaps;
  if (!S || what < 0 || what >= HSO_DBG_N || !out || !S->dbg[what] || S->dbg_bytes[what] != bytes) return hso_fail(ctx, HSO_E_INVALID, "debug_fetch: no such table, or a size mismatch");
  if (bytes == 0) return HSO_OK;
  HSO_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  HSO_HIP_CHECK(ctx, hipMemcpyAsync(out, S->dbg[what], bytes, hipMemcpyDeviceToHost, ctx->stream));
  HSO_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  return HSO_OK;
}

// a sequence map as the library holds it (include/hso_gpu_debug.h): host-side tables from the host copies, device tables by one
// copy behind whatever patches are still queued on the stream
int hso_gpu_seqmap_debug_dump(hso_gpu_ctx* ctx, int map, int what, void* out, size_t bytes)
{
  if (!ctx) return HSO_E_INVALID;
  SeqMap* m = seqmap_of(ctx, map);
  if (!m || !out || what < 0 || what >= HSO_DUMP_N) return hso_fail(ctx, HSO_E_INVALID, "seqmap_debug_dump: bad argument");
  const size_t nk = m->kfs.size();
  const void* host = nullptr; const void* dev = nullptr; size_t want = 0;
  int64_t sizes[HSO_DUMP_N_SIZES] = {(int64_t)nk, (int64_t)m->n_pts, (int64_t)m->n_obs, m->fts_cap, m->n_cands, m->ff_n[0], m->ff_n[1], m->ff_frame[0], m->ff_frame[1],
                                     m->ff_newest, (int64_t)sizeof(hso_kf), (int64_t)sizeof(hso_map_point), (int64_t)sizeof(hso_obs), (int64_t)sizeof(hso_seq_feature),
                                     (int64_t)sizeof(hso_seq_job), (int64_t)sizeof(hso_seq_result)};
  std::vector<int32_t> lists;
  switch (what) {
    case HSO_DUMP_SIZES: host = sizes; want = sizeof(sizes); break;
    case HSO_DUMP_KFS: host = m->kfs.data(); want = sizeof(hso_kf) * nk; break;
    case HSO_DUMP_KEY_POINTS: host = m->key_points.data(); want = sizeof(int32_t) * 5 * nk; break;
    case HSO_DUMP_KF_NFTS: host = m->kf_nfts.data(); want = sizeof(int32_t) * nk; break;
    case HSO_DUMP_POINTS: dev = m->d_pts; want = sizeof(hso_map_point) * m->n_pts; break;
    case HSO_DUMP_OBS: dev = m->d_obs; want = sizeof(hso_obs) * m->n_obs; break;
    case HSO_DUMP_OBS_POINT: dev = m->d_obs_pt; want = sizeof(int32_t) * m->n_obs; break;
    case HSO_DUMP_KF_FTS: dev = m->d_kf_fts; want = sizeof(int32_t) * nk * (size_t)m->fts_cap; break;
    case HSO_DUMP_CANDS: dev = m->d_cands; want = sizeof(int32_t) * (size_t)m->n_cands; break;
    case HSO_DUMP_FRAME_FEATS0: dev = m->d_ff[0]; want = sizeof(hso_seq_feature) * (size_t)m->ff_n[0]; break;
    default: dev = m->d_ff[1]; want = sizeof(hso_seq_feature) * (size_t)m->ff_n[1]; break;
  }
  if (want != bytes) return hso_fail(ctx, HSO_E_INVALID, "seqmap_debug_dump: bytes differs from the table's size");
  if (bytes == 0) return HSO_OK;
  if (host) { memcpy(out, host, bytes); return HSO_OK; }
  if (!dev || (what == HSO_DUMP_KF_FTS && m->kf_rows_cap < nk)) return hso_fail(ctx, HSO_E_INVALID, "seqmap_debug_dump: the table was never sent");
  HSO_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  HSO_HIP_CHECK(ctx, hipMemcpyAsync(out, dev, bytes, hipMemcpyDeviceToHost, ctx->stream));
  HSO_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  return HSO_OK;
}

}  // extern "C"

void hso_seqmaps_debug_set(hso_gpu_ctx* ctx, int what, const void* d, size_t bytes)
{
  if (!ctx->seqmaps) return;
  ctx->seqmaps->dbg[what] = d; ctx->seqmaps->dbg_bytes[what] = bytes;
}

// ---------------------------------------------------------------------------------------------
// The front half of the resident chain (include/hso_gpu.h: hso_gpu_seq_chain; orchestrated in hso_select.hip).
size_t hso_chain_sizeof_reproj_kf() { return sizeof(ReprojKf); }
size_t hso_chain_sizeof_align_job() { return sizeof(AlignJobDev); }

// a map's device view for one job.  The reference frame's features: the keyframe's list, or the table the map holds for that frame;
// the new frame's table goes into the other of the map's two buffers.
int hso_seqmap_chain_view(hso_gpu_ctx* ctx, const hso_seq_job& job, SeqMapDev* out, int* n_kfs, const int32_t** kf_nfts_host)
{
  SeqMap* m = seqmap_of(ctx, job.map);
  if (!m) return hso_fail(ctx, HSO_E_INVALID, "seq_chain: no such map");
  if (m->kfs.empty() || !m->d_kfs) return hso_fail(ctx, HSO_E_INVALID, "seq_chain: the map has no keyframes");
  const int nk = (int)m->kfs.size();
  // the visiting order and the covisibility votes are formed in tables of HSO_SEQ_MAX_KFS rows (k_chain_visit, k_chain_finish): a
  // longer keyframe table would silently leave its NEWEST rows out of both.  The reference never gets there — Map keeps at most
  // Config::maxNKfs() = 2000 keyframes by dropping the farthest (src/frame_handler_mono.cpp:340-346) — so a caller that does is refused.
  if (nk > HSO_SEQ_MAX_KFS) return hso_fail(ctx, HSO_E_INVALID, "seq_chain: the map's keyframe table exceeds HSO_SEQ_MAX_KFS rows");
  if (m->fts_cap < 1 || m->kf_rows_cap < (size_t)nk) return hso_fail(ctx, HSO_E_INVALID, "seq_chain: the keyframes' feature lists were never sent (hso_gpu_seqmap_patch_lists)");
  if (job.last_kf_row < -1 || job.last_kf_row >= nk || job.ref_kf_row < -1 || job.ref_kf_row >= nk) return hso_fail(ctx, HSO_E_INVALID, "seq_chain: keyframe row out of range");
  for (int q = 0; q < 5; q++) if (job.covis[q] < -1 || job.covis[q] >= nk) return hso_fail(ctx, HSO_E_INVALID, "seq_chain: covisible keyframe row out of range");
  int ref = -1;
  if (job.n_ref_feats < 0) return hso_fail(ctx, HSO_E_INVALID, "seq_chain: negative feature count");
  if (job.ref_kf_row >= 0) {
    if (job.n_ref_feats != m->kf_nfts[(size_t)job.ref_kf_row]) return hso_fail(ctx, HSO_E_INVALID, "seq_chain: n_ref_feats differs from the keyframe's list");
  } else if (!(job.flags & HSO_SEQ_NO_TRACK)) {
    ref = m->ff_frame[0] == job.ref_frame_id && m->d_ff[0] ? 0 : (m->ff_frame[1] == job.ref_frame_id && m->d_ff[1] ? 1 : -1);
    if (ref < 0) return hso_fail(ctx, HSO_E_NOFRAME, "seq_chain: the map holds no feature table of the reference frame");
    if (job.n_ref_feats != m->ff_n[ref]) return hso_fail(ctx, HSO_E_INVALID, "seq_chain: n_ref_feats differs from the reference frame's table");
  }
  const int cur = ref >= 0 ? 1 - ref : 1 - m->ff_newest;
  out->pts = m->d_pts; out->obs = m->d_obs; out->obs_pt = m->d_obs_pt; out->kfs = m->d_kfs; out->kf_fts = m->d_kf_fts; out->cands = m->d_cands;
  out->first = m->d_first; out->ff_ref = ref >= 0 ? m->d_ff[ref] : nullptr; out->ff_cur = m->d_ff[cur];
  out->n_pts = (int)m->n_pts; out->n_obs = (int)m->n_obs; out->n_kfs = nk; out->fts_cap = m->fts_cap; out->n_cands = m->n_cands; out->ff_cap = m->ff_cap;
  *n_kfs = nk; *kf_nfts_host = m->kf_nfts.data();
  return HSO_OK;
}

// hso_gpu_seq_local_ba (hso_ba.hip): a map's tables as the window assembly reads them, the library's own keyframe table and the
// lengths of the keyframes' feature lists; _ba_set_pose: a core keyframe's pose after the optimisation (the device copy of the
// keyframe table follows with the next flush)
int hso_seqmap_ba_view(hso_gpu_ctx* ctx, int map, SeqMapDev* out, const hso_kf** kfs_host, const int32_t** kf_nfts_host)
{
  SeqMap* m = seqmap_of(ctx, map);
  if (!m) return hso_fail(ctx, HSO_E_INVALID, "seq_local_ba: no such map");
  const int nk = (int)m->kfs.size();
  if (nk == 0) return hso_fail(ctx, HSO_E_INVALID, "seq_local_ba: the map has no keyframes");
  if (nk > HSO_SEQ_MAX_KFS) return hso_fail(ctx, HSO_E_INVALID, "seq_local_ba: the map's keyframe table exceeds HSO_SEQ_MAX_KFS rows");
  if (m->fts_cap < 1 || m->kf_rows_cap < (size_t)nk) return hso_fail(ctx, HSO_E_INVALID, "seq_local_ba: the keyframes' feature lists were never sent (hso_gpu_seqmap_patch_lists)");
  memset(out, 0, sizeof(*out));
  out->pts = m->d_pts; out->obs = m->d_obs; out->obs_pt = m->d_obs_pt; out->kfs = m->d_kfs; out->kf_fts = m->d_kf_fts; out->cands = m->d_cands;
  out->first = m->d_first;
  out->n_pts = (int)m->n_pts; out->n_obs = (int)m->n_obs; out->n_kfs = nk; out->fts_cap = m->fts_cap; out->n_cands = m->n_cands; out->ff_cap = m->ff_cap;
  *kfs_host = m->kfs.data(); *kf_nfts_host = m->kf_nfts.data();
  return HSO_OK;
}
void hso_seqmap_ba_set_pose(hso_gpu_ctx* ctx, int map, int row, const hso_se3& T)
{
  SeqMap* m = seqmap_of(ctx, map);
  if (!m || row < 0 || (size_t)row >= m->kfs.size()) return;
  m->kfs[(size_t)row].T_f_w = T;
  if (!m->kfs_stale) { m->kfs_stale = true; ctx->seqmaps->stale_kfs.push_back(map); }
}

// the new frame's table needs room for max_fts rows in both buffers (before any view is taken)
int hso_seqmap_chain_reserve(hso_gpu_ctx* ctx, int map, int rows)
{
  SeqMap* m = seqmap_of(ctx, map);
  if (!m) return hso_fail(ctx, HSO_E_INVALID, "seq_chain: no such map");
  if (rows <= m->ff_cap) return HSO_OK;
  HSO_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  const int ncap = rows + 64;
  for (int b = 0; b < 2; b++) {
    hso_seq_feature* q = nullptr;
    HSO_HIP_CHECK(ctx, hipMalloc(reinterpret_cast<void**>(&q), sizeof(hso_seq_feature) * (size_t)ncap));
    if (m->d_ff[b] && m->ff_n[b] > 0) HSO_HIP_CHECK(ctx, hipMemcpy(q, m->d_ff[b], sizeof(hso_seq_feature) * (size_t)m->ff_n[b], hipMemcpyDeviceToDevice));
    (void)hipFree(m->d_ff[b]);
    m->d_ff[b] = q;
  }
  m->ff_cap = ncap;
  return HSO_OK;
}

void hso_seqmap_chain_commit(hso_gpu_ctx* ctx, const hso_seq_job& job, int n_feats)
{
  SeqMap* m = seqmap_of(ctx, job.map);
  if (!m) return;
  const int ref = job.ref_kf_row >= 0 ? -1 : (m->ff_frame[0] == job.ref_frame_id && m->d_ff[0] ? 0 : (m->ff_frame[1] == job.ref_frame_id && m->d_ff[1] ? 1 : -1));
  const int cur = ref >= 0 ? 1 - ref : 1 - m->ff_newest;
  m->ff_frame[cur] = job.cur_frame_id; m->ff_n[cur] = n_feats; m->ff_newest = cur;
}

// ---- CoarseTracker::makeDepthRef (src/CoarseTracker.cpp:210-240) into the tracker's own table layout: per reference feature
// (px, f, distance of its point along the bearing in the reference frame, -1 without a usable point); pad columns zero
__global__ __launch_bounds__(256) void k_chain_table(const ChainJobDev* jobs)
{
  const ChainJobDev& J = jobs[blockIdx.y];
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= J.n_ref_stride) return;
  double v[6] = {0, 0, 0, 0, 0, 0};
  if (i < J.n_ref) {
    int p;
    if (J.ref_kf_row >= 0) {
      const int f = J.M.kf_fts[(size_t)J.ref_kf_row * J.M.fts_cap + i];
      const hso_obs& o = J.M.obs[f];
      v[0] = o.px[0]; v[1] = o.px[1]; v[2] = o.f[0]; v[3] = o.f[1]; v[4] = o.f[2];
      p = J.M.obs_pt[f];
    } else {
      const hso_seq_feature& r = J.M.ff_ref[i];
      v[0] = r.px[0]; v[1] = r.px[1]; v[2] = r.f[0]; v[3] = r.f[1]; v[4] = r.f[2];
      p = r.point;
    }
    v[5] = -1;
    if (p >= 0 && p < J.M.n_pts) {
      const hso_map_point& P = J.M.pts[p];
      if (P.idist != 0.0) {      // a row the caller never sent (a point hosted in no keyframe) has no usable depth
        const Se3 T_ref_host = se3_mul(se3_from(J.T_ref_w), se3_inverse(se3_from(J.M.kfs[P.host_kf].T_f_w)));
        const double s = 1.0 / P.idist;
        double x, y, z;
        se3_apply(T_ref_host, P.host_f[0] * s, P.host_f[1] * s, P.host_f[2] * s, x, y, z);
        if (!(z < 0.00001)) v[5] = sqrt(x * x + y * y + z * z);
      }
    }
  }
#pragma unroll
  for (int c = 0; c < 6; c++) J.table[(size_t)c * J.n_ref_stride + i] = v[c];
}

int hso_chain_table_launch(hso_gpu_ctx* ctx, const ChainJobDev* d_jobs, int n_jobs, int n_max_stride)
{
  if (n_jobs <= 0 || n_max_stride <= 0) return HSO_OK;
  hipLaunchKernelGGL(k_chain_table, dim3((unsigned)((n_max_stride + 255) / 256), (unsigned)n_jobs), dim3(256), 0, ctx->stream, d_jobs);
  HSO_HIP_CHECK(ctx, hipGetLastError());
  return HSO_OK;
}

// ---- after the tracker: the write-back of CoarseTracker::run (:198-202), the per-keyframe products of the reprojection, and which
// keyframes the frame visits (Reprojector::reprojectMap, src/reprojector.cpp:108-199: the reference frame's connected keyframes,
// then Map::getCloseKeyframes (src/map.cpp:193-213) sorted by distance, until max_kfs).  One workgroup per job.
#define CHAIN_MAX_KFS HSO_SEQ_MAX_KFS
struct ChainFrontDev {
  const ChainJobDev* jobs; ChainCur* cur; const hso_track_result* track; const int32_t* kf_nfts; const int32_t* temps;
  ReprojKf* kfs; int32_t* ids; uint8_t* quality; PoseJobDev* pose_jobs;
  int n_jobs, n_total, max_kfs, cell_size, grid_n_cols;
};

__global__ __launch_bounds__(256) void k_chain_visit(ChainFrontDev F, hso_camera cam)
{
  __shared__ double s_dist[CHAIN_MAX_KFS];
  __shared__ Se3 s_Tc;
  __shared__ double s_expo;
  const int b = blockIdx.x, tid = threadIdx.x;
  const ChainJobDev& J = F.jobs[b];
  ChainCur& C = F.cur[b];
  if (tid == 0) {
    Se3 Tc = se3_from(J.T_cur_w0);
    double expo = -1.0;                                            // Frame::m_exposure_time as constructed
    if (!(J.flags & HSO_SEQ_NO_TRACK)) {
      const hso_track_result& R = F.track[b];
      Tc = se3_mul(se3_from(R.T_cur_ref), se3_from(J.T_ref_w));  // cur.T_f_w_ = m_T_cur_ref * ref.T_f_w_
      expo = (double)R.exposure_rat * J.ref_exposure;
      if (R.exposure_rat > 0.99f && R.exposure_rat < 1.01f) expo = J.ref_exposure;
    }
    s_Tc = Tc; s_expo = expo;
    se3_to(Tc, C.T_cur_w); C.exposure = expo;
    const Se3 inv = se3_inverse(Tc);
    C.cur_pos[0] = inv.tx; C.cur_pos[1] = inv.ty; C.cur_pos[2] = inv.tz;
    se3_to(Tc, F.pose_jobs[b].T);                                  // the pose optimiser starts from the tracked pose
  }
  __syncthreads();
  const Se3 Tc = s_Tc;
  const int nk = J.M.n_kfs < CHAIN_MAX_KFS ? J.M.n_kfs : CHAIN_MAX_KFS;
  for (int k = tid; k < J.M.n_kfs; k += 256) {
    const SeqKfDev& K = J.M.kfs[k];
    const Se3 Tk = se3_from(K.T_f_w);
    const Se3 inv = se3_inverse(Tk);
    ReprojKf& Q = F.kfs[J.kf_begin + k];
    Q.T_cur_kf = se3_mul(Tc, inv);
    Q.pos[0] = inv.tx; Q.pos[1] = inv.ty; Q.pos[2] = inv.tz;
    Q.base = K.base; Q.frame_id = K.frame_id;
    Q.exposure_rat = (float)(s_expo / K.exposure_time);
    Q.kf_gap_lt4 = (J.cur_keyframe_id - K.keyframe_id) < 4;
    if (k >= CHAIN_MAX_KFS) continue;
    // getCloseKeyframes: the first key point of the keyframe that the frame sees (Frame::isVisible) makes it a neighbour, at the
    // distance of the two T_f_w translations (as written there)
    double dist = -1.0;
    for (int q = 0; q < 5; q++) {
      const int p = K.key_point[q];
      if (p < 0 || p >= J.M.n_pts) continue;
      const hso_map_point& P = J.M.pts[p];
      double x, y, z;
      se3_apply(Tc, P.pos[0], P.pos[1], P.pos[2], x, y, z);
      if (z < 0.0) continue;
      double u, v;
      world2cam(cam, x, y, z, u, v);
      if (!(u >= 0.0 && v >= 0.0 && u < cam.width && v < cam.height)) continue;
      const double dx = Tc.tx - Tk.tx, dy = Tc.ty - Tk.ty, dz = Tc.tz - Tk.tz;
      dist = sqrt(dx * dx + dy * dy + dz * dz);
      break;
    }
    s_dist[k] = dist;
  }
  __syncthreads();
  if (tid == 0) {
    int n = 0;
    for (int q = 0; q < 5; q++) {
      const int r = J.covis[q];
      if (r < 0 || r >= J.M.n_kfs) continue;
      bool dup = false;
      for (int e = 0; e < n; e++) dup |= C.visit[e] == r;
      if (!dup && n < HSO_SEQ_MAX_VISIT) C.visit[n++] = r;
    }
    // the neighbours in ascending distance (stable: equal distances keep table order), skipping the ones already visited
    int count = n;
    double last_d = -1.0; int last_k = -1;
    while (count < F.max_kfs && n < HSO_SEQ_MAX_VISIT) {
      int best = -1; double bd = 0;
      for (int k = 0; k < nk; k++) {
        const double d = s_dist[k];
        if (d < 0.0) continue;
        if (d < last_d || (d == last_d && k <= last_k)) continue;   // already considered
        if (best < 0 || d < bd) { best = k; bd = d; }
      }
      if (best < 0) break;
      last_d = bd; last_k = best;
      bool dup = false;
      for (int e = 0; e < n; e++) dup |= C.visit[e] == best;
      if (dup) continue;
      C.visit[n++] = best; ++count;
    }
    C.n_visit = n;
  }
}

// ---- the list of points the frame projects, in the reference's order: per visited keyframe the points of its features (once each
// per frame: the first feature that names a point lists it; deleted and temporary points are skipped), then the candidates, then the
// temporary points.  One workgroup per job; "first feature that names it" = atomicMin over the flat feature positions, kept
// entries compacted in order by block scans: the result is the sequential walk's.
// 1024 threads per sequence: the walk is three dependent loads per feature row (list -> link -> state word) over ~20 000 rows, and
// what bounds it is how many of those are in flight (256 threads: 0.32 ms per 128 sequences)
#define LIST_THREADS 1024
#define LIST_CACHE 16   // rows per thread and round of the keyframes' feature lists (16 384 rows per round)
__global__ __launch_bounds__(LIST_THREADS) void k_chain_list(ChainFrontDev F)
{
  __shared__ int s_wave[LIST_THREADS / 64];
  __shared__ int s_off[HSO_SEQ_MAX_VISIT + 1];
  __shared__ int s_cnt[LIST_CACHE * (LIST_THREADS / 64)];
  const int b = blockIdx.x, tid = threadIdx.x;
  const ChainJobDev& J = F.jobs[b];
  ChainCur& C = F.cur[b];
  const int nv = C.n_visit;
  if (tid == 0) {
    int t = 0;
    for (int v = 0; v < nv; v++) { s_off[v] = t; t += F.kf_nfts[J.kf_begin + C.visit[v]]; }
    s_off[nv] = t;
  }
  __syncthreads();
  const int total = s_off[nv];
  int32_t* ids = F.ids + J.slice_begin; uint8_t* qual = F.quality + J.slice_begin;
  auto point_at = [&](int g, int& key) -> int {
    int v = 0;
    while (v + 1 < nv && s_off[v + 1] <= g) v++;
    const int f = J.M.kf_fts[(size_t)C.visit[v] * J.M.fts_cap + (g - s_off[v])];
    const int p = J.M.obs_pt[f];
    if (p < 0 || p >= J.M.n_pts) return -1;
    key = (int)((uint32_t)J.M.pts[p].pad_ & 0xffu);
    const int kind = key >> 4;
    if (kind == 0 || kind == 1) return -1;                         // TYPE_DELETED (its features' links are NULL in the reference), TYPE_TEMPORARY
    return p;
  };
  // Rounds of LIST_CACHE x LIST_THREADS rows.  A thread's rows g = base + k * LIST_THREADS + tid (coalesced per k) are resolved in
  // three sweeps of independent loads — list entry, link, state word: LIST_CACHE of each in flight per thread instead of one chain
  // per barrier — and kept in registers as (point << 8 | key); the rows that name their point first are then placed by ONE scan
  // over the (k, wavefront) ballot counts instead of a block scan per k.  "First" only looks at smaller g, so a round needs the
  // stamps of its own and of earlier rounds: the rounds run one after the other, the stamps are reset after the last.
  // (16 rows per thread: 86 registers; 32 rows spill at the 128 a workgroup of 1024 threads has.)
  int n = 0;
  int cache[LIST_CACHE];
  const int lane = tid & 63, wave = tid >> 6;
  for (int base = 0; base < total; base += LIST_CACHE * LIST_THREADS) {
    int v = 0;                                                     // g grows with k: the keyframe a row belongs to only moves forward
#pragma unroll
    for (int k = 0; k < LIST_CACHE; k++) {
      const int g = base + k * LIST_THREADS + tid;
      cache[k] = -1;
      if (g < total) {
        while (v + 1 < nv && s_off[v + 1] <= g) v++;
        cache[k] = J.M.kf_fts[(size_t)C.visit[v] * J.M.fts_cap + (g - s_off[v])];
      }
    }
#pragma unroll
    for (int k = 0; k < LIST_CACHE; k++) { int pt = -1; if (cache[k] >= 0 && cache[k] < J.M.n_obs) pt = J.M.obs_pt[cache[k]]; cache[k] = (pt >= 0 && pt < J.M.n_pts) ? pt : -1; }
#pragma unroll
    for (int k = 0; k < LIST_CACHE; k++) {
      if (cache[k] >= 0) {
        const int key = (int)((uint32_t)J.M.pts[cache[k]].pad_ & 0xffu);
        const int kind = key >> 4;
        cache[k] = (kind != 0 && kind != 1) ? (cache[k] << 8) | key : -1;   // not TYPE_DELETED (its features' links are NULL in the reference), not TYPE_TEMPORARY
      }
    }
#pragma unroll
    for (int k = 0; k < LIST_CACHE; k++) if (cache[k] >= 0) atomicMin(&J.M.first[cache[k] >> 8], base + k * LIST_THREADS + tid);
    __threadfence_block();
    __syncthreads();
    unsigned keep = 0;
#pragma unroll
    for (int k = 0; k < LIST_CACHE; k++) if (cache[k] >= 0 && J.M.first[cache[k] >> 8] == base + k * LIST_THREADS + tid) keep |= 1u << k;
#pragma unroll
    for (int k = 0; k < LIST_CACHE; k++) {
      const unsigned long long bl = __ballot((keep >> k) & 1u);
      if (lane == 0) s_cnt[k * (LIST_THREADS / 64) + wave] = __popcll(bl);
    }
    __syncthreads();
    {
      static_assert(LIST_CACHE * (LIST_THREADS / 64) <= LIST_THREADS, "one thread per (k, wavefront) count");
      const int mine = tid < LIST_CACHE * (LIST_THREADS / 64) ? s_cnt[tid] : 0;
      int tot;
      const int incl = sel_scan_waves<LIST_THREADS / 64>(mine, s_wave, tot);
      __syncthreads();
      if (tid < LIST_CACHE * (LIST_THREADS / 64)) s_cnt[tid] = incl - mine;
      __syncthreads();
#pragma unroll
      for (int k = 0; k < LIST_CACHE; k++) {
        const bool kp = (keep >> k) & 1u;
        const unsigned long long bl = __ballot(kp);
        if (kp) {
          const int pos = n + s_cnt[k * (LIST_THREADS / 64) + wave] + __popcll(bl & ((1ull << lane) - 1ull));
          if (pos < J.slice_cap) { ids[pos] = cache[k] >> 8; qual[pos] = (uint8_t)(cache[k] & 0xff); }
        }
      }
      n += tot;
      __syncthreads();
    }
  }
  // the stamps go back to "unset": every stamped point has exactly one first row, and that row is in the list
  if (n <= J.slice_cap) { for (int i = tid; i < n; i += LIST_THREADS) J.M.first[ids[i]] = SEQ_FIRST_UNSET; }
  else for (int g = tid; g < total; g += LIST_THREADS) { int key; const int p = point_at(g, key); if (p >= 0) J.M.first[p] = SEQ_FIRST_UNSET; }
  const int n_kf_points = n < J.slice_cap ? n : J.slice_cap;
  n = n_kf_points;
  // MapPointCandidates::candidates_ in list order (an entry the device deleted since the caller last sent the list is skipped)
  int n_c = 0;
  for (int i0 = 0; i0 < J.M.n_cands; i0 += LIST_THREADS) {
    const int i = i0 + tid;
    int key = 0, p = -1;
    if (i < J.M.n_cands) { p = J.M.cands[i]; key = (int)((uint32_t)J.M.pts[p].pad_ & 0xffu); if ((key >> 4) != 2) p = -1; }
    int tot;
    const int pos = sel_scan_waves<LIST_THREADS / 64>(p >= 0 ? 1 : 0, s_wave, tot) + n - (p >= 0 ? 1 : 0);
    if (p >= 0 && pos < J.slice_cap) { ids[pos] = p; qual[pos] = (uint8_t)key; }
    n += tot; n_c += tot;
    __syncthreads();
  }
  if (n > J.slice_cap) { n_c -= n - J.slice_cap; n = J.slice_cap; }
  const int n_before_temps = n;
  for (int i = tid; i < J.n_temps; i += LIST_THREADS) {
    const int p = F.temps[J.temps_begin + i];
    if (n_before_temps + i < J.slice_cap) { ids[n_before_temps + i] = p; qual[n_before_temps + i] = (uint8_t)((uint32_t)J.M.pts[p].pad_ & 0xffu); }
  }
  n = n_before_temps + J.n_temps < J.slice_cap ? n_before_temps + J.n_temps : J.slice_cap;
  if (tid == 0) { C.n_listed = n; C.n_kf_points = n_kf_points; C.n_cands_listed = n_c; }
}

// one listed point: reprojectPoint + getCloseViewObs + the findMatchDirect job (slice entries past the list's end: null jobs)
__global__ __launch_bounds__(256) void k_chain_reproject(ChainFrontDev F, hso_camera cam, AlignJobDev* jobs, hso_reproj_point* proj)
{
  const int g = blockIdx.x * 256 + threadIdx.x;
  if (g >= F.n_total) return;
  int lo = 0, hi = F.n_jobs - 1;          // the job whose slice holds g
  while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (F.jobs[mid].slice_begin <= g) lo = mid; else hi = mid - 1; }
  const ChainJobDev& J = F.jobs[lo];
  const ChainCur& C = F.cur[lo];
  const int i = g - J.slice_begin;
  hso_reproj_point r;
  r.projected = 0; r.cell = 0; r.px[0] = 0; r.px[1] = 0; r.ref_obs = -1; r.pad_ = 0;
  jobs[g].ref_base = nullptr; jobs[g].cur_base = J.cur_base;
  if (i < C.n_listed) {
    const int pid = F.ids[g];
    if (pid >= 0 && pid < J.M.n_pts) {
      ReprojFrameDev Fr;
      Fr.cur_pos[0] = C.cur_pos[0]; Fr.cur_pos[1] = C.cur_pos[1]; Fr.cur_pos[2] = C.cur_pos[2];
      Fr.cur_base = J.cur_base; Fr.kf_begin = J.kf_begin; Fr.pad_ = 0;
      r = reproject_one<true>(cam, J.M.pts[pid], Fr, F.kfs + J.kf_begin, J.M.obs, F.cell_size, F.grid_n_cols, &jobs[g]);
    }
    r.pad_ = F.quality[g];
  }
  proj[g] = r;
}

int hso_chain_front_launch(hso_gpu_ctx* ctx, const hso_camera* cam, const ChainFront& A)
{
  SeqMaps* S = ctx->seqmaps;
  if (!S || !S->have_g || cam->width != S->g.w[0] || cam->height != S->g.h[0]) return hso_fail(ctx, HSO_E_INVALID, "seq_chain: camera size differs from the frame size");
  ChainFrontDev F;
  F.jobs = A.d_jobs; F.cur = A.d_cur; F.track = A.d_track; F.kf_nfts = A.d_kf_nfts; F.temps = A.d_temps; F.kfs = A.d_kfs; F.ids = A.d_ids; F.quality = A.d_quality;
  F.pose_jobs = A.d_pose_jobs; F.n_jobs = A.n_jobs; F.n_total = A.n_total; F.max_kfs = A.max_kfs; F.cell_size = A.cell_size; F.grid_n_cols = A.grid_n_cols;
  hipLaunchKernelGGL(k_chain_visit, dim3(A.n_jobs), dim3(256), 0, ctx->stream, F, *cam);
  hipLaunchKernelGGL(k_chain_list, dim3(A.n_jobs), dim3(LIST_THREADS), 0, ctx->stream, F);
  if (A.n_total > 0) {
    HSO_HIP_CHECK(ctx, hipMemsetAsync(A.d_match, 0, sizeof(hso_align_out) * (size_t)A.n_total, ctx->stream));
    hipLaunchKernelGGL(k_chain_reproject, dim3((A.n_total + 255) / 256), dim3(256), 0, ctx->stream, F, *cam, A.d_align, A.d_proj);
    AlignConsts C;
    C.cam = *cam; C.g = S->g;
    launch_align(ctx, true, C, A.d_align, A.n_total, A.d_match);
    hipLaunchKernelGGL(k_match_brief, dim3((A.n_total + 255) / 256), dim3(256), 0, ctx->stream, A.n_total, A.d_proj, A.d_match, A.d_align, A.d_brief);
  }
  HSO_HIP_CHECK(ctx, hipGetLastError());
  return HSO_OK;
}
