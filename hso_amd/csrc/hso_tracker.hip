// hso_tracker.hip — CoarseTracker on gfx950: pyramidal direct photometric alignment
// (7-DoF: exposure ratio + SE(3)) with the whole coarse-to-fine Levenberg-Marquardt
// loop resident on the device.
//
// Replaces CoarseTracker::run and its helpers (reference src/CoarseTracker.cpp):
//   makeDepthRef :210-240, precomputeReferencePatches :416-497,
//   selectRobustFunctionLevel :530-644, computeResiduals :242-414,
//   computeGS :499-525 (+ Accumulator7, include/hso/MatrixAccumulator.h), run :51-208.
//
// MI355X design (details and measurements: DESIGN.md section 3.2)
//   * one workgroup owns one (ref, cur) pair for the whole level/LM loop; independent pairs are pulled from a
//     device-side job counter by a persistent grid, so a batch fills the chip with no host round trip and no
//     inter-workgroup communication.  The device code (hso_tracker_core.h) is compiled in two shapes — 512 threads
//     with the whole LDS, 256 threads with half of it — and a large batch runs the coarse levels two-per-CU and the
//     finest level one-per-CU (see below);
//   * per level the reference image, then the current image, is staged once into LDS (LDS-DMA; <= 90 KB for EuRoC
//     level 1); the taps of a feature's pattern are read as per-row windows (two aligned ds_read2_b32 +
//     v_alignbyte_b32 per row) and every pixel is converted to float once;
//   * the arithmetic that feeds decisions (projection, bilinear intensity, residual, Huber weight, saturation test)
//     mirrors the reference's expressions operation by operation (compiled with -ffp-contract=off), so visibility,
//     term / saturation counts and the MAD thresholds are bit-identical to the CPU restatement; the image gradients
//     and the sums H, b, E only feed tolerance-compared quantities and use FMAs and fixed-tree reductions (fp32 H
//     like the reference's Accumulator7, fp64 b, fp64 E);
//   * J = [-I_ref, dx*A + dy*B] with A = fx_l*J_row0, B = fy_l*J_row1 is never materialised: per feature nine
//     weighted moments of (I_ref, dx, dy, r) are accumulated over the pattern and expanded once into the 28+7
//     normal-equation entries; a halving (reduce-scatter) exchange across the wave (v_permlane32/16_swap + DPP) and
//     one LDS stage finish the sum;
//   * the 7x7 pivoted LDL^T solve and the SE(3) update run on one lane, fully unrolled in registers;
//   * median / MAD are exact order statistics (radix select on float bit patterns: 12-bit leading digit histogrammed
//     while the keys are produced, the winning bin compacted into LDS, the remaining digits over the candidates),
//     equal to nth_element at floor(n/2) (include/hso/vikit/math_utils.h:119-126).
// Nothing here is a dense contraction, so MFMA is not used (BASELINE.json north_star).
#include "hso_ctx.h"
#include <stdlib.h>
#include "hso_dev_math.h"
#include <string.h>
#include <algorithm>
#include <condition_variable>
#include <mutex>

using namespace hso_dev;

#include "hso_tracker_defs.h"


// Two shapes of the same device code (hso_tracker_core.h):
//   trk1: 512 threads, the whole 160 KB of a CU — one job per CU; fits the level-1 image of a 752x480 frame (90 KB) in LDS;
//   trk2: 256 threads, 80 KB — two workgroups share a CU, so one job's serial phases (robust thresholds, the 7x7 solve)
//         overlap the other's evaluations; fits level images up to ~60 KB (levels 4..2).
// A batch that fills the chip twice over runs the coarse levels on trk2 and the last level(s) on trk1 (track_launch below).
// (a third shape, several workgroups per job for small batches, lives in hso_tracker_coop.hip)
#define TRK_COOP 0
#ifndef TRK1_THREADS
#define TRK1_THREADS 512
#endif
#define TRK_THREADS TRK1_THREADS
#define TRK_LDS_KB 160
#define TRK_WAVES_PER_EU (TRK1_THREADS / 256)
#ifndef TRK1_OLD_SHARE
#define TRK1_OLD_SHARE 10
#endif
#define TRK_OLD_SHARE TRK1_OLD_SHARE  // the older wavefront of a SIMD wins issue arbitration: it gets 10/16 of the features
namespace trk1 {
#include "hso_tracker_core.h"
}
#undef TRK_THREADS
#undef TRK_LDS_KB
#undef TRK_OLD_SHARE
#undef TRK_WAVES_PER_EU
#ifndef TRK2_PER_CU
#define TRK2_PER_CU 2     // workgroups of the small shape per CU (3: 168 registers, 52 KB — measured slower)
#endif
#define TRK_THREADS 256
#define TRK_LDS_KB (TRK2_PER_CU == 2 ? 80 : 52)
#define TRK_WAVES_PER_EU TRK2_PER_CU
#ifndef TRK2_OLD_SHARE
#define TRK2_OLD_SHARE 8
#endif
#define TRK_OLD_SHARE TRK2_OLD_SHARE   // one wavefront per SIMD and workgroup: even split
namespace trk2 {
#include "hso_tracker_core.h"
}
#undef TRK_THREADS
#undef TRK_LDS_KB
#undef TRK_OLD_SHARE
#undef TRK_WAVES_PER_EU

// makeDepthRef, CoarseTracker.cpp:210-240
__global__ void k_make_depth_ref(const hso_depth_ref_in* in, int n, const hso_se3* poses, hso_se3 T_ref_w, double* out)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double d = -1;
  if (in[i].has_point) {
    const double inv = 1.0 / in[i].idist;
    const Se3 Th = se3_from(poses[in[i].host_pose]);
    const Se3 T_r_h = se3_mul(se3_from(T_ref_w), se3_inverse(Th));
    double x, y, z;
    se3_apply(T_r_h, in[i].host_f[0] * inv, in[i].host_f[1] * inv, in[i].host_f[2] * inv, x, y, z);
    if (!(z < 0.00001)) d = sqrt(x * x + y * y + z * z);
  }
  out[i] = d;
}

// ------------------------------------------------------------------ host side

struct TrackBatchState {
  TrackConsts C;
  TrackLevel lv[HSO_N_PYR_LEVELS];
  int n_jobs = 0, n_max = 0, grid = 0, grid2 = 0, split_level = -1;
  size_t lds_bytes = 0, lds_bytes2 = 0, scratch_stride = 0;
  TrackJobDev* d_jobs = nullptr; size_t jobs_cap = 0;
  double* d_feats = nullptr; size_t feats_cap = 0;
  char* d_scratch = nullptr; size_t scratch_cap = 0;
  hso_track_result* d_results = nullptr; size_t results_cap = 0;
  int* d_counter = nullptr;
  hso_eval_out* d_eval = nullptr;
  bool attr_set = false;
  std::vector<TrackJobDev> h_jobs;
  // cooperative shape (small batches): coop_K >= 2 workgroups per job
  int coop_K = 0, coop_scatter = 0;
  bool coop_broken = false;       // a cooperative launch timed out once: this context stays on the one-workgroup shapes
  bool coop_turn = false;         // this context holds its device's cooperative-launch turn (launch .. collect)
  TrackJobDev* d_subjobs = nullptr; size_t subjobs_cap = 0;
  CoopJobState* d_coop = nullptr; size_t coop_cap = 0;
  std::vector<TrackJobDev> h_subjobs;
  // hso_gpu_coarse_track_collect_begin / _end: two page-locked result images and their events; pend[] = slots in flight, oldest first
  hso_track_result* h_res[2] = {nullptr, nullptr}; size_t h_res_cap[2] = {0, 0};
  hipEvent_t res_ev[2] = {nullptr, nullptr};
  int pend[2] = {-1, -1}, pend_n[2] = {0, 0}, n_pend = 0;
};

// One cooperative launch per device at a time.  Its workgroups wait for each other, so all of them must become resident; two such
// launches from two contexts (two banks of sequences on their own streams) can each take part of the CUs and then wait for the
// rest forever — until the spin bound fails both and both contexts fall back to the one-workgroup shape for good.  A launch of this
// shape fills the chip anyway, so contexts take turns: the turn is held from the launch to the collect that follows it.
namespace {
struct CoopTurn { std::mutex m; std::condition_variable cv; bool busy = false; };
CoopTurn g_coop_turn[16];
void coop_turn_take(hso_gpu_ctx* ctx, TrackBatchState* st)
{
  if (st->coop_turn) return;
  CoopTurn& T = g_coop_turn[ctx->device & 15];
  std::unique_lock<std::mutex> lk(T.m);
  T.cv.wait(lk, [&] { return !T.busy; });
  T.busy = true; st->coop_turn = true;
}
void coop_turn_give(hso_gpu_ctx* ctx, TrackBatchState* st)
{
  if (!st->coop_turn) return;
  CoopTurn& T = g_coop_turn[ctx->device & 15];
  { std::lock_guard<std::mutex> lk(T.m); T.busy = false; }
  st->coop_turn = false;
  T.cv.notify_one();
}
}  // namespace

void hso_track_state_free(hso_gpu_ctx* ctx)
{
  TrackBatchState* st = ctx->track;
  if (!st) return;
  coop_turn_give(ctx, st);
  (void)hipFree(st->d_jobs); (void)hipFree(st->d_feats); (void)hipFree(st->d_scratch); (void)hipFree(st->d_results);
  (void)hipFree(st->d_counter); (void)hipFree(st->d_eval); (void)hipFree(st->d_subjobs); (void)hipFree(st->d_coop);
  for (int k = 0; k < 2; k++) { if (st->h_res[k]) (void)hipHostFree(st->h_res[k]); if (st->res_ev[k]) (void)hipEventDestroy(st->res_ev[k]); }
  delete st;
  ctx->track = nullptr;
}

template <typename T>
static int grow(hso_gpu_ctx* ctx, T** p, size_t* cap, size_t need_bytes)
{
  if (*cap >= need_bytes && *p) return HSO_OK;
  if (*p) { HSO_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream)); (void)hipFree(*p); *p = nullptr; }
  const size_t bytes = std::max<size_t>(need_bytes + need_bytes / 4, 256);   // headroom: feature tables vary a little from call to call
  HSO_HIP_CHECK(ctx, hipMalloc(reinterpret_cast<void**>(p), bytes));
  *cap = bytes;
  return HSO_OK;
}


static int track_prepare(hso_gpu_ctx* ctx, const hso_camera* cam, const hso_track_params* p,
                         const hso_track_job* jobs, int n_jobs, int max_grid, bool sync_after = true)
{
  if (!ctx) return HSO_E_INVALID;
  if (!cam || !p || !jobs || n_jobs <= 0) return hso_fail(ctx, HSO_E_INVALID, "coarse_track: null argument or no jobs");
  if (p->max_level < 0 || p->max_level >= HSO_N_PYR_LEVELS || p->min_level < 0 || p->min_level > p->max_level)
    return hso_fail(ctx, HSO_E_INVALID, "coarse_track: bad level range");
  if (p->n_iter < 0) return hso_fail(ctx, HSO_E_INVALID, "coarse_track: n_iter < 0");
  HSO_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  if (!ctx->track) ctx->track = new TrackBatchState();
  TrackBatchState* st = ctx->track;

  // geometry: all frames of a batch share it
  auto it0 = ctx->frames.find(jobs[0].cur_frame_id);
  if (it0 == ctx->frames.end()) return hso_fail(ctx, HSO_E_NOFRAME, "coarse_track: current frame not resident");
  const PyrGeom g = it0->second.g;
  if (cam->width != g.w[0] || cam->height != g.h[0])
    return hso_fail(ctx, HSO_E_INVALID, "coarse_track: camera size differs from the frame size");

  int n_max = 1;
  size_t total_feats = 0;
  for (int j = 0; j < n_jobs; j++) {
    if (jobs[j].n_feats < 0 || (jobs[j].n_feats > 0 && !jobs[j].feats))
      return hso_fail(ctx, HSO_E_INVALID, "coarse_track: bad feature table");
    n_max = std::max(n_max, jobs[j].n_feats);
    total_feats += (size_t)((jobs[j].n_feats + 31) & ~31);
  }
  // feats_soa == 2: the tables are already on the device in the kernel's layout (the resident chain builds them there)
  bool resident = n_jobs > 0 && jobs[0].feats_soa == 2;
  for (int j = 0; j < n_jobs; j++) if ((jobs[j].feats_soa == 2) != resident) return hso_fail(ctx, HSO_E_INVALID, "coarse_track: device-resident and host feature tables cannot share a batch");
  // Callers that hand over the kernel's layout (feats_soa) in one contiguous block skip the host pass altogether.
  bool direct = !resident;
  for (int j = 0; j < n_jobs && direct; j++) {
    direct = jobs[j].feats_soa == 1;
    if (direct && j > 0 && jobs[j].n_feats > 0) {
      const double* want = reinterpret_cast<const double*>(jobs[0].feats);
      size_t off = 0;
      for (int q = 0; q < j; q++) off += (size_t)((jobs[q].n_feats + 31) & ~31) * 6;
      direct = reinterpret_cast<const double*>(jobs[j].feats) == want + off;
    }
  }
  // Otherwise the SoA feature tables are formed in the context's page-locked staging buffer (one pass; a std::vector staged by
  // the copy wrapper meant a zero fill, the transpose and a staging copy: three passes over 6 MB for 64 jobs of 2000 features)
  double* const h_feats = resident ? nullptr : direct ? const_cast<double*>(reinterpret_cast<const double*>(jobs[0].feats))
                                 : reinterpret_cast<double*>(hso_pinned(ctx, 0, total_feats * 6 * sizeof(double) + 64));
  if (!h_feats && !resident) return HSO_E_NOMEM;
  st->h_jobs.resize(n_jobs);
  if (!resident) if (int rc = grow(ctx, &st->d_feats, &st->feats_cap, total_feats * 6 * sizeof(double))) return rc;
  size_t foff = 0;
  for (int j = 0; j < n_jobs; j++) {
    auto itr = ctx->frames.find(jobs[j].ref_frame_id);
    auto itc = ctx->frames.find(jobs[j].cur_frame_id);
    if (itr == ctx->frames.end() || itc == ctx->frames.end())
      return hso_fail(ctx, HSO_E_NOFRAME, "coarse_track: frame not resident");
    if (itr->second.g.w[0] != g.w[0] || itr->second.g.h[0] != g.h[0] || itc->second.g.w[0] != g.w[0] ||
        itc->second.g.h[0] != g.h[0])
      return hso_fail(ctx, HSO_E_INVALID, "coarse_track: frames of one batch must share one size");
    const int n = jobs[j].n_feats, ns = (n + 31) & ~31;
    double* dst = resident ? nullptr : h_feats + foff * 6;
    if (direct || resident) {
      // nothing to do: the caller's block is the upload image / the table is on the device
    } else if (jobs[j].feats_soa == 1) {
      memcpy(dst, jobs[j].feats, sizeof(double) * 6 * (size_t)ns);
    } else {
      for (int i = 0; i < n; i++) {
        const hso_ref_feat& f = jobs[j].feats[i];
        dst[0 * ns + i] = f.px[0]; dst[1 * ns + i] = f.px[1];
        dst[2 * ns + i] = f.f[0]; dst[3 * ns + i] = f.f[1]; dst[4 * ns + i] = f.f[2];
        dst[5 * ns + i] = f.dist;
      }
      for (int i = n; i < ns; i++) for (int c = 0; c < 6; c++) dst[c * ns + i] = 0.0;   // the pad columns
    }
    TrackJobDev& d = st->h_jobs[j];
    d.ref_base = itr->second.base;
    d.cur_base = itc->second.base;
    d.feats = resident ? const_cast<double*>(reinterpret_cast<const double*>(jobs[j].feats)) : st->d_feats + foff * 6;
    d.n = n; d.n_stride = ns;
    d.T = jobs[j].T_cur_ref;
    d.a = jobs[j].exposure_rat;
    d.n_total = n;
    d.coop_K = 1; d.coop_pad_ = 0;
    foff += ns;
  }

  TrackConsts& C = st->C;
  C.cam = *cam;
  C.inverse = p->inverse_composition; C.max_level = p->max_level; C.min_level = p->min_level; C.n_iter = p->n_iter;
  C.n_max = (n_max + 31) & ~31;
  for (int l = 0; l < HSO_N_PYR_LEVELS; l++) {
    TrackLevel& V = st->lv[l];
    const int pi = p->max_level - l + PATTERN_OFFSET;  // CoarseTracker.cpp:80
    V.w = g.w[l]; V.h = g.h[l]; V.off = g.off[l];
    V.pa = 0; V.pad = 0; V.pi = -1;
    for (int k = 0; k < TRK_MAX_PA; k++) V.poff[k] = 0;
    if (pi < 0 || pi > 7) continue;
    V.pi = pi;
    V.pa = h_pattern_num[pi]; V.pad = h_pattern_pad[pi];
    for (int k = 0; k < h_pattern_num[pi]; k++) V.poff[k] = h_pattern[pi][k][1] * g.w[l] + h_pattern[pi][k][0];
  }
  C.level_first = C.max_level; C.level_last = C.min_level; C.resume = 0;
  C.lds_img_cap = trk1::kImgCap;
  C.keys_in_memory = 0;
  st->lds_bytes = (size_t)trk1::kImgCap + sizeof(trk1::Shared);
  st->lds_bytes2 = (size_t)trk2::kImgCap + sizeof(trk2::Shared);
  st->n_jobs = n_jobs;
  st->n_max = C.n_max;
  st->grid = std::min(n_jobs, max_grid > 0 ? max_grid : ctx->n_cu);
  // Coarse levels on two 256-thread workgroups per CU when the batch fills the chip at least twice and the coarse images fit
  // the smaller LDS share; a single sequence (latency, not throughput) keeps all eight wavefronts of a CU on its one job.
  st->split_level = -1;
  const int split_min_jobs = 2 * ctx->n_cu;
  if (max_grid <= 0 && n_jobs >= split_min_jobs && C.max_level >= 2 && C.min_level <= 1) {
    const int lvl = 2;   // levels max_level .. 2 on trk2, 1 .. min_level on trk1
    const size_t need = (size_t)g.w[lvl] * g.h[lvl] + g.w[lvl] + 64;
    if (need <= (size_t)trk2::kImgCap) st->split_level = lvl;
  }
  st->grid2 = std::min(n_jobs, TRK2_PER_CU * ctx->n_cu);
  st->scratch_stride = scratch_bytes(C.n_max);
  // Small batches (a single sequence: BASELINE configs[2] / [3]; up to one job per XCD): K_j workgroups share job j
  // (hso_tracker_coop.hip).  K_j follows the job's own feature count — about COOP_FEATS_PER_WG per workgroup — so a job's
  // result does not depend on what else is in the batch; all workgroups of the launch are resident at once (K_j <= CUs per XCD).
  st->coop_K = 0;
  st->coop_scatter = ctx->opt.track_coop_scatter ? 1 : 0;
  // Medium batches (more jobs than XCDs, fewer than CUs: a bank of 16..128 sequences): the chip would run one workgroup per job
  // and leave the other CUs idle for the ~1.3 ms a 2000-feature job takes; instead every job is split over floor(CUs / jobs)
  // workgroups, spread over the XCDs (the placement-independent transport).  Here K does depend on the batch size — results of a
  // job agree with its solo run within the tracker's tolerance, not bit for bit (hso_gpu_options.track_no_coop pins the one-workgroup shape).
  const int share = n_jobs > COOP_MAX_JOBS ? ctx->n_cu / n_jobs : COOP_KMAX;
  if (n_jobs > COOP_MAX_JOBS && share >= 2) st->coop_scatter = 1;
  // a context that shares the device with others (hso_gpu_set_shared_device) leaves medium batches on the one-workgroup shape:
  // the other contexts' kernels fill the CUs a batch of < 256 jobs leaves idle, splitting jobs only adds exchange work (7.8 us per
  // job against 3.3), and cooperative launches of several contexts take turns (one at a time per device)
  const bool medium_ok = !ctx->shared_device;
  if (max_grid <= 0 && (n_jobs <= COOP_MAX_JOBS || (share >= 2 && medium_ok)) && !st->coop_broken && !ctx->opt.track_no_coop) {
    const int per_xcd = std::min(std::min(COOP_KMAX, std::max(1, ctx->n_cu / 8)), share);
    const int fpw = ctx->opt.track_coop_feats_per_wg > 0 ? ctx->opt.track_coop_feats_per_wg : COOP_FEATS_PER_WG;
    int kmax = 0;
    for (int j = 0; j < n_jobs; j++) {
      int K = std::max(1, (st->h_jobs[j].n + fpw - 1) / fpw);
      if (ctx->opt.track_coop_workgroups > 0) K = ctx->opt.track_coop_workgroups;
      K = std::min(K, per_xcd);
      st->h_jobs[j].coop_K = K;
      kmax = std::max(kmax, K);
    }
    // the stride of the slice table.  Every job of a small batch takes the cooperative kernel, also the ones that stay on one
    // workgroup (K_j = 1: no exchange happens) — a job's result must not depend on which kernel its neighbours need
    st->coop_K = kmax;
  }
  if (st->coop_K) st->split_level = -1;

  if (int rc = grow(ctx, &st->d_jobs, &st->jobs_cap, sizeof(TrackJobDev) * n_jobs)) return rc;
  if (int rc = grow(ctx, &st->d_scratch, &st->scratch_cap, st->scratch_stride * std::max(std::max(st->grid, st->split_level >= 0 ? st->grid2 : 0), n_jobs * st->coop_K))) return rc;
  if (st->coop_K) {
    const int KS = st->coop_K;
    st->h_subjobs.assign((size_t)n_jobs * KS, TrackJobDev{});
    for (int j = 0; j < n_jobs; j++) {
      const TrackJobDev& d = st->h_jobs[j];
      const int K = d.coop_K;
      const int chunk = (d.n + K - 1) / K;
      for (int r = 0; r < K; r++) {
        TrackJobDev& q = st->h_subjobs[(size_t)j * KS + r];
        q = d;
        const int f0 = std::min(d.n, r * chunk), f1 = std::min(d.n, f0 + chunk);
        q.feats = d.feats + f0;      // SoA [6][n_stride]: a slice is the same table advanced by f0 columns
        q.n = f1 - f0;
        q.n_total = d.n;
      }
    }
    if (int rc = grow(ctx, &st->d_subjobs, &st->subjobs_cap, sizeof(TrackJobDev) * n_jobs * KS)) return rc;
    if (int rc = grow(ctx, &st->d_coop, &st->coop_cap, sizeof(CoopJobState) * n_jobs)) return rc;
    HSO_HIP_CHECK(ctx, hipMemcpyAsync(st->d_subjobs, st->h_subjobs.data(), sizeof(TrackJobDev) * n_jobs * KS, hipMemcpyHostToDevice, ctx->stream));
  }
  if (int rc = grow(ctx, &st->d_results, &st->results_cap, sizeof(hso_track_result) * n_jobs)) return rc;
  // one small allocation: [0, 256) the job counter, [256, ...) the level table
  if (!st->d_counter) HSO_HIP_CHECK(ctx, hipMalloc(reinterpret_cast<void**>(&st->d_counter), 256 + sizeof(st->lv)));
  C.lv = reinterpret_cast<const TrackLevel*>(reinterpret_cast<const char*>(st->d_counter) + 256);
  HSO_HIP_CHECK(ctx, hipMemcpyAsync(const_cast<TrackLevel*>(C.lv), st->lv, sizeof(st->lv), hipMemcpyHostToDevice, ctx->stream));
  if (!st->d_eval) HSO_HIP_CHECK(ctx, hipMalloc(reinterpret_cast<void**>(&st->d_eval), sizeof(hso_eval_out)));
  if (!resident) HSO_HIP_CHECK(ctx, hipMemcpyAsync(st->d_feats, h_feats, total_feats * 6 * sizeof(double),
                                                   hipMemcpyHostToDevice, ctx->stream));
  HSO_HIP_CHECK(ctx, hipMemcpyAsync(st->d_jobs, st->h_jobs.data(), sizeof(TrackJobDev) * n_jobs,
                                    hipMemcpyHostToDevice, ctx->stream));
  if (!st->attr_set) {
    HSO_HIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(trk1::k_track<false>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)st->lds_bytes));
    HSO_HIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(trk1::k_track<true>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)st->lds_bytes));
    HSO_HIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(trk1::k_eval<false>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)st->lds_bytes));
    HSO_HIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(trk1::k_eval<true>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)st->lds_bytes));
    HSO_HIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(trk2::k_track<false>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)st->lds_bytes2));
    HSO_HIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(trk2::k_track<true>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)st->lds_bytes2));
    st->attr_set = true;
  }
  // the feature tables leave the staging buffer asynchronously: a caller of the split interface may use other entry points (which
  // reuse that buffer) before it launches, so it waits here; hso_gpu_coarse_track_batch goes straight on to launch + collect
  if (sync_after) HSO_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  return HSO_OK;
}

extern "C" {

int hso_gpu_coarse_track_prepare(hso_gpu_ctx* ctx, const hso_camera* cam, const hso_track_params* params,
                                 const hso_track_job* jobs, int n_jobs)
{
  return track_prepare(ctx, cam, params, jobs, n_jobs, 0);
}

int hso_gpu_coarse_track_launch(hso_gpu_ctx* ctx)
{
  if (!ctx || !ctx->track || ctx->track->n_jobs <= 0) return hso_fail(ctx, HSO_E_INVALID, "coarse_track_launch: nothing prepared");
  TrackBatchState* st = ctx->track;
  TrackConsts C = st->C;
  if (st->coop_K) {
    C.lds_img_cap = hso_track_coop_img_cap();
    coop_turn_take(ctx, st);   // given back by the collect
    hipError_t e = hipMemsetAsync(st->d_counter, 0, 2 * sizeof(int), ctx->stream);   // [1] = the fail flag
    if (e == hipSuccess) e = hipMemsetAsync(st->d_coop, 0, sizeof(CoopJobState) * st->n_jobs, ctx->stream);
    if (e == hipSuccess) e = hso_track_coop_launch(ctx->stream, C, st->d_subjobs, st->n_jobs, st->coop_K, st->coop_scatter, st->d_coop,
                                                   reinterpret_cast<unsigned*>(st->d_counter) + 1, st->d_scratch, st->scratch_stride, st->d_results);
    if (e != hipSuccess) coop_turn_give(ctx, st);
    HSO_HIP_CHECK(ctx, e);
    return HSO_OK;
  }
  if (st->split_level >= 0) {
    // coarse levels: two 256-thread workgroups per CU (one job's thresholds / LM solve overlap the other's evaluations)
    C.level_first = C.max_level; C.level_last = st->split_level; C.resume = 0; C.lds_img_cap = trk2::kImgCap;
    HSO_HIP_CHECK(ctx, hipMemsetAsync(st->d_counter, 0, sizeof(int), ctx->stream));
    if (C.inverse)
      hipLaunchKernelGGL(trk2::k_track<true>, dim3(st->grid2), dim3(256), st->lds_bytes2, ctx->stream, C, st->d_jobs, st->n_jobs, st->d_counter,
                         st->d_scratch, st->scratch_stride, st->d_results);
    else
      hipLaunchKernelGGL(trk2::k_track<false>, dim3(st->grid2), dim3(256), st->lds_bytes2, ctx->stream, C, st->d_jobs, st->n_jobs, st->d_counter,
                         st->d_scratch, st->scratch_stride, st->d_results);
    if (st->split_level <= C.min_level) { HSO_HIP_CHECK(ctx, hipGetLastError()); return HSO_OK; }   // every level ran above
    // the remaining level(s): one 512-thread workgroup per CU with the whole LDS, continuing from the records above
    C.level_first = st->split_level - 1; C.level_last = C.min_level; C.resume = 1;
  }
  C.lds_img_cap = trk1::kImgCap;
  HSO_HIP_CHECK(ctx, hipMemsetAsync(st->d_counter, 0, sizeof(int), ctx->stream));
  if (C.inverse)
    hipLaunchKernelGGL(trk1::k_track<true>, dim3(st->grid), dim3(TRK1_THREADS), st->lds_bytes, ctx->stream, C, st->d_jobs,
                       st->n_jobs, st->d_counter, st->d_scratch, st->scratch_stride, st->d_results);
  else
    hipLaunchKernelGGL(trk1::k_track<false>, dim3(st->grid), dim3(TRK1_THREADS), st->lds_bytes, ctx->stream, C, st->d_jobs,
                       st->n_jobs, st->d_counter, st->d_scratch, st->scratch_stride, st->d_results);
  HSO_HIP_CHECK(ctx, hipGetLastError());
  return HSO_OK;
}

int hso_gpu_coarse_track_collect(hso_gpu_ctx* ctx, hso_track_result* results)
{
  if (!ctx || !ctx->track || !results) return hso_fail(ctx, HSO_E_INVALID, "coarse_track_collect: bad argument");
  TrackBatchState* st = ctx->track;
  {
    hipError_t e = hipMemcpyAsync(results, st->d_results, sizeof(hso_track_result) * st->n_jobs, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    coop_turn_give(ctx, st);       // the cooperative kernel has left the device (or the stream failed)
    HSO_HIP_CHECK(ctx, e);
  }
  if (st->coop_K) {
    bool failed = false;
    for (int j = 0; j < st->n_jobs; j++) failed = failed || results[j].status != 0;
    if (failed) {
      // A workgroup of the cooperative launch never became resident within the spin bound (other work held its CU for
      // seconds).  Not a result: rerun the batch on the one-workgroup shapes — same arithmetic per feature, no co-residency
      // requirement — and keep this context on them from now on.
      st->coop_K = 0;
      st->coop_broken = true;
      ctx->err = "note: a cooperative tracker launch timed out; this context stays on the one-workgroup shape (results unaffected)";   // readable with hso_gpu_last_error
      if (int rc = hso_gpu_coarse_track_launch(ctx)) return rc;
      return hso_gpu_coarse_track_collect(ctx, results);
    }
  }
  return HSO_OK;
}

}  // extern "C"

// The tracker inside the resident chain (hso_gpu_seq_chain): prepare + launch over device-built tables; the result records stay on
// the device, where the chain's next kernel reads them.  A cooperative launch (a batch smaller than the chip) is collected here —
// its time-out fallback must be known before the chain goes on changing the sequence tables — so *d_results is valid either way.
int hso_track_chain_launch(hso_gpu_ctx* ctx, const hso_camera* cam, const hso_track_params* p, const hso_track_job* jobs, int n_jobs,
                           const hso_track_result** d_results, bool* cooperative)
{
  int rc = track_prepare(ctx, cam, p, jobs, n_jobs, 0, false);
  if (rc < 0) return rc;
  TrackBatchState* st = ctx->track;
  rc = hso_gpu_coarse_track_launch(ctx);
  if (rc < 0) return rc;
  *cooperative = st->coop_K != 0;
  if (st->coop_K) {
    std::vector<hso_track_result> tmp((size_t)n_jobs);
    rc = hso_gpu_coarse_track_collect(ctx, tmp.data());   // waits; reruns on the one-workgroup shape after a time-out
    if (rc < 0) return rc;
  }
  *d_results = st->d_results;
  return HSO_OK;
}

extern "C" {

// The read-back split in two, so that a caller that steps a resident batch again and again can enqueue the NEXT step's work before
// it waits for this step's records: _begin queues the copy of the result records (into a page-locked image of the library) behind
// the launch and returns; _end waits for that copy alone and hands the records over.  Two read-backs may be in flight.  The device
// records of step k are copied before step k + 1's kernel can overwrite them (one stream), so no record is lost; between a launch and
// the next the stream never runs dry (0.21 ms of a 16.9 ms step at 4096 pairs were the synchronous read-back + the next launches).
int hso_gpu_coarse_track_collect_begin(hso_gpu_ctx* ctx)
{
  if (!ctx) return HSO_E_INVALID;
  if (!ctx->track || ctx->track->n_jobs <= 0) return hso_fail(ctx, HSO_E_INVALID, "coarse_track_collect_begin: nothing launched");
  TrackBatchState* st = ctx->track;
  if (st->coop_K) return hso_fail(ctx, HSO_E_UNSUPPORTED, "coarse_track_collect_begin: a cooperative launch (a batch smaller than the chip) is collected with hso_gpu_coarse_track_collect");
  if (st->n_pend >= 2) return hso_fail(ctx, HSO_E_INVALID, "coarse_track_collect_begin: two read-backs are in flight already");
  const int slot = st->n_pend == 0 ? 0 : 1 - st->pend[0];
  const size_t bytes = sizeof(hso_track_result) * (size_t)st->n_jobs;
  if (st->h_res_cap[slot] < bytes) {
    if (st->h_res[slot]) (void)hipHostFree(st->h_res[slot]);
    st->h_res[slot] = nullptr; st->h_res_cap[slot] = 0;
    HSO_HIP_CHECK(ctx, hipHostMalloc(reinterpret_cast<void**>(&st->h_res[slot]), bytes + bytes / 4, hipHostMallocDefault));
    st->h_res_cap[slot] = bytes + bytes / 4;
  }
  if (!st->res_ev[slot]) HSO_HIP_CHECK(ctx, hipEventCreateWithFlags(&st->res_ev[slot], hipEventDisableTiming));
  HSO_HIP_CHECK(ctx, hipMemcpyAsync(st->h_res[slot], st->d_results, bytes, hipMemcpyDeviceToHost, ctx->stream));
  HSO_HIP_CHECK(ctx, hipEventRecord(st->res_ev[slot], ctx->stream));
  st->pend[st->n_pend] = slot; st->pend_n[st->n_pend] = st->n_jobs; st->n_pend++;
  return HSO_OK;
}

int hso_gpu_coarse_track_collect_end(hso_gpu_ctx* ctx, hso_track_result* results)
{
  if (!ctx) return HSO_E_INVALID;
  if (!ctx->track || !results) return hso_fail(ctx, HSO_E_INVALID, "coarse_track_collect_end: bad argument");
  TrackBatchState* st = ctx->track;
  if (st->n_pend <= 0) return hso_fail(ctx, HSO_E_INVALID, "coarse_track_collect_end: no read-back in flight");
  const int slot = st->pend[0], n = st->pend_n[0];
  st->pend[0] = st->pend[1]; st->pend_n[0] = st->pend_n[1]; st->n_pend--;
  HSO_HIP_CHECK(ctx, hipEventSynchronize(st->res_ev[slot]));
  memcpy(results, st->h_res[slot], sizeof(hso_track_result) * (size_t)n);
  return HSO_OK;
}

int hso_gpu_coarse_track_batch(hso_gpu_ctx* ctx, const hso_camera* cam, const hso_track_params* params,
                               const hso_track_job* jobs, int n_jobs, hso_track_result* results)
{
  if (!results) return hso_fail(ctx, HSO_E_INVALID, "coarse_track_batch: null results");
  int rc = track_prepare(ctx, cam, params, jobs, n_jobs, 0, false);
  if (rc < 0) return rc;
  rc = hso_gpu_coarse_track_launch(ctx);
  if (rc < 0) return rc;
  return hso_gpu_coarse_track_collect(ctx, results);
}

int hso_gpu_tracker_eval(hso_gpu_ctx* ctx, const hso_camera* cam, const hso_track_params* params,
                         const hso_track_job* job, int level, const hso_se3* T_cur_ref, float exposure_rat,
                         float huber_thresh, float outlier_thresh, hso_eval_out* out, float* ref_patch_out,
                         uint8_t* visible_out, float* abs_err_out)
{
  if (!ctx) return HSO_E_INVALID;
  if (!job || !T_cur_ref || !out || !params) return hso_fail(ctx, HSO_E_INVALID, "tracker_eval: null argument");
  if (level < params->min_level || level > params->max_level) return hso_fail(ctx, HSO_E_INVALID, "tracker_eval: level out of range");
  int rc = track_prepare(ctx, cam, params, job, 1, 1);
  if (rc < 0) return rc;
  TrackBatchState* st = ctx->track;
  st->C.keys_in_memory = abs_err_out ? 1 : 0;
  trk1::EvalArgs ea;
  ea.level = level; ea.T = *T_cur_ref; ea.a = exposure_rat; ea.huber = huber_thresh; ea.outlier = outlier_thresh;
  if (st->C.inverse)
    hipLaunchKernelGGL(trk1::k_eval<true>, dim3(1), dim3(TRK1_THREADS), st->lds_bytes, ctx->stream, st->C, st->d_jobs, ea,
                       st->d_scratch, st->d_eval);
  else
    hipLaunchKernelGGL(trk1::k_eval<false>, dim3(1), dim3(TRK1_THREADS), st->lds_bytes, ctx->stream, st->C, st->d_jobs, ea,
                       st->d_scratch, st->d_eval);
  HSO_HIP_CHECK(ctx, hipGetLastError());
  HSO_HIP_CHECK(ctx, hipMemcpyAsync(out, st->d_eval, sizeof(hso_eval_out), hipMemcpyDeviceToHost, ctx->stream));
  HSO_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  const int pat_idx = params->max_level - level + PATTERN_OFFSET;
  const int PA = h_pattern_num[pat_idx], n = job->n_feats, nm = st->C.n_max;
  const Scratch sc = scratch_at(st->d_scratch, nm);
  if (ref_patch_out && n > 0) {
    std::vector<float> tmp((size_t)PA * nm);
    { HSO_HIP_CHECK(ctx, hipMemcpyAsync(tmp.data(), sc.ref_patch, tmp.size() * 4, hipMemcpyDeviceToHost, ctx->stream)); HSO_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream)); }
    std::vector<uint8_t> vis(nm);
    { HSO_HIP_CHECK(ctx, hipMemcpyAsync(vis.data(), sc.visible, nm, hipMemcpyDeviceToHost, ctx->stream)); HSO_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream)); }
    for (int f = 0; f < n; f++)
      for (int pi = 0; pi < PA; pi++) ref_patch_out[(size_t)f * PA + pi] = vis[f] ? tmp[(size_t)pi * nm + f] : 0.0f;
  }
  if (visible_out && n > 0) { HSO_HIP_CHECK(ctx, hipMemcpyAsync(visible_out, sc.visible, n, hipMemcpyDeviceToHost, ctx->stream)); HSO_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream)); }
  if (abs_err_out && n > 0 && huber_thresh <= 0) {
    std::vector<uint32_t> keys((size_t)PA * n);
    { HSO_HIP_CHECK(ctx, hipMemcpyAsync(keys.data(), sc.keys, keys.size() * 4, hipMemcpyDeviceToHost, ctx->stream)); HSO_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream)); }
    size_t k = 0;
    for (size_t i = 0; i < keys.size(); i++)
      if (keys[i] != KEY_INVALID) { float v; memcpy(&v, &keys[i], 4); abs_err_out[k++] = v; }
  }
  return HSO_OK;
}

int hso_gpu_tracker_pattern(int max_level, int level, int* patch_area, int* half_patch, int8_t* offsets_xy)
{
  const int off = max_level - level + PATTERN_OFFSET;
  if (off < 0 || off > 7) return HSO_E_INVALID;
  if (patch_area) *patch_area = h_pattern_num[off];
  if (half_patch) *half_patch = h_pattern_pad[off];
  if (offsets_xy) memcpy(offsets_xy, h_pattern[off], 80);
  return off;
}

int hso_gpu_make_depth_ref(hso_gpu_ctx* ctx, const hso_depth_ref_in* in, int n, const hso_se3* poses_f_w,
                           int n_poses, const hso_se3* T_ref_w, double* dist_out)
{
  if (!ctx) return HSO_E_INVALID;
  if (n < 0 || (n > 0 && (!in || !dist_out)) || !poses_f_w || !T_ref_w || n_poses <= 0)
    return hso_fail(ctx, HSO_E_INVALID, "make_depth_ref: bad argument");
  if (n == 0) return HSO_OK;
  for (int i = 0; i < n; i++)
    if (in[i].has_point && (in[i].host_pose < 0 || in[i].host_pose >= n_poses))
      return hso_fail(ctx, HSO_E_INVALID, "make_depth_ref: host_pose out of range");
  HSO_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  char* d = nullptr;
  const size_t b_in = sizeof(hso_depth_ref_in) * n, b_p = sizeof(hso_se3) * n_poses, b_o = sizeof(double) * n;
  const size_t o1 = (b_in + 255) & ~size_t(255), o2 = o1 + ((b_p + 255) & ~size_t(255));
  HSO_HIP_CHECK(ctx, hipMalloc(reinterpret_cast<void**>(&d), o2 + b_o));
  (void)hipMemcpyAsync(d, in, b_in, hipMemcpyHostToDevice, ctx->stream);
  (void)hipMemcpyAsync(d + o1, poses_f_w, b_p, hipMemcpyHostToDevice, ctx->stream);
  hipLaunchKernelGGL(k_make_depth_ref, dim3((n + 255) / 256), dim3(256), 0, ctx->stream,
                     reinterpret_cast<const hso_depth_ref_in*>(d), n, reinterpret_cast<const hso_se3*>(d + o1),
                     *T_ref_w, reinterpret_cast<double*>(d + o2));
  hipError_t e = hipMemcpyAsync(dist_out, d + o2, b_o, hipMemcpyDeviceToHost, ctx->stream);
  hipError_t e2 = hipStreamSynchronize(ctx->stream);
  (void)hipFree(d);
  if (e != hipSuccess || e2 != hipSuccess) { ctx->err = "make_depth_ref: HIP failure"; return HSO_E_HIP; }
  return HSO_OK;
}

}  // extern "C"

