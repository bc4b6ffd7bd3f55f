// hso_activate.hip — seed activation on gfx950: re-match every converged seed in the frames that
// observed it, gate on the mean drift, refine the inverse depth with a 1-D Levenberg-Marquardt.
//
// Replaces DepthFilter::activatePoint (reference src/depth_filter.cpp:729-851) and
// DepthFilter::seedOptimizer (:853-1073) with Matcher::findMatchSeed (src/matcher.cpp:442-518),
// Point::jacobian_id2uv (include/hso/point.h:174-184) and MADScaleEstimator::compute
// (src/vikit/robust_cost.cpp:67-74).
//
// MI355X mapping: the reference visits a seed's <=30 target frames one after another on the
// mapping thread.  Here every (seed, target) pair gets a thread for its geometry (k_activate_prep: projection and
// parallax tests, the matcher's job record) and a quarter of a wavefront for the 8x8 Lucas-Kanade of findMatchSeed
// (the reprojection matcher's own kernel, k_align_t<true>, over those jobs), and kernel 2 gives every seed one wavefront with
// lane = target frame: residuals and Jacobians of all targets are evaluated in parallel, the
// sums over targets are formed in the reference's order (see k_activate_opt), so they carry the
// reference's rounding (the only non-IEEE step is pow(x,3) in the damping update).
#include "hso_match_dev.h"
#include "hso_align_dev.h"
#include <string.h>
#include <algorithm>
#include <vector>

using namespace hso_dev;

#define ACT_WAVES_PER_BLOCK 4

struct ActSeedDev {
  const uint8_t* ref_base;
  hso_seed s;
  int32_t first, count;  // range in the pair arrays
  int32_t n_mean_converge_frame, _pad;   // DepthFilter::nMeanConvergeFrame_ of the sequence the seed belongs to
};

struct ActPairIn { int32_t seed, frame; };   // (seed, index into the call's table of target frames)
struct ActFrameDev { const uint8_t* base; hso_se3 T_f_w; double exposure; };

struct ActPair {          // kernel 1 -> kernel 2
  int32_t is_target;      // passed the projection test (:741-769)
  int32_t matched;        // findMatchSeed returned true
  double px[2];           // projected position before matching
  hso_se3 Tth;            // target.T_f_w * host.T_f_w^-1
  double obs[2];          // project2d(cam2world(matched px)), :817-818
  double normal[2];       // normalised A_cur_ref * grad (edgelets), :803-805
  hso_align_out mo;
};

struct ActConsts {
  hso_camera cam;
  PyrGeom g;
  double z_min;   // depth below which the projection test fails: 0.0001 in activatePoint (:748), 0.001 in Reprojector::reprojectorSeed
};

// ---- kernel 1a: one THREAD per (seed, target) pair — the projection test of activatePoint (:741-769), the parallax test of
// findMatchSeed (src/matcher.cpp:444-449) and the matcher's job record; a pair that fails a test gets a null job.  The matching
// itself (kernel 1b) is the reprojection matcher's kernel over these jobs (hso_align.hip: k_align_t<true>, four pairs per
// wavefront, NCC threshold 0.8 — checkNCC(.., 0.8), :509): the first version gave every pair a wavefront of its own, whose 64 lanes
// all repeated the fp64 geometry and whose four 16-lane rows all matched the same pair (0.74 ms per step of 128 sequences).
__global__ __launch_bounds__(256) void k_activate_prep(ActConsts C, const ActSeedDev* seeds, const ActPairIn* pin, const ActFrameDev* frames, int n_pairs,
                                                      ActPair* pout, AlignJobDev* jobs)
{
  const int pid = blockIdx.x * 256 + threadIdx.x;
  if (pid >= n_pairs) return;
  const ActPairIn& P = pin[pid];
  const ActFrameDev& Fr = frames[P.frame];
  const ActSeedDev& SD = seeds[P.seed];
  const hso_seed& S = SD.s;
  const int W = C.g.w[0], H = C.g.h[0];
  ActPair o;
  memset(&o, 0, sizeof(o));
  jobs[pid].ref_base = nullptr; jobs[pid].cur_base = Fr.base;
  const Se3 Thost_inv = se3_inverse(se3_from(S.T_ref_w));
  const Se3 Ttw = se3_from(Fr.T_f_w);
  const Se3 Tth = se3_mul(Ttw, Thost_inv);
  se3_to(Tth, o.Tth);
  const double sc = 1.0 / (double)S.mu;
  const double ph0 = S.f[0] * sc, ph1 = S.f[1] * sc, ph2 = S.f[2] * sc;
  bool go = true;
  {
    double x, y, z;
    se3_apply(Tth, ph0, ph1, ph2, x, y, z);
    if (z < C.z_min) go = false;
    if (go) {
      double pu, pv;
      world2cam(C.cam, x, y, z, pu, pv);
      const int ox = (int)pu, oy = (int)pv;
      if (!(ox >= 8 && ox < W - 8 && oy >= 8 && oy < H - 8)) go = false;  // isInFrame(px.cast<int>(), 8), camera.h:79-83
      o.px[0] = pu; o.px[1] = pv;
    }
  }
  if (!go) { pout[pid] = o; return; }
  o.is_target = 1;
  // parallax test of findMatchSeed, matcher.cpp:444-449
  {
    double sx, sy, sz;
    se3_apply(Thost_inv, ph0, ph1, ph2, sx, sy, sz);
    double r0 = Thost_inv.tx - sx, r1 = Thost_inv.ty - sy, r2 = Thost_inv.tz - sz;
    const double rn = sqrt(r0 * r0 + r1 * r1 + r2 * r2);
    r0 /= rn; r1 /= rn; r2 /= rn;
    const Se3 Tt_inv = se3_inverse(Ttw);
    double c0 = Tt_inv.tx - sx, c1 = Tt_inv.ty - sy, c2 = Tt_inv.tz - sz;
    const double cn = sqrt(c0 * c0 + c1 * c1 + c2 * c2);
    c0 /= cn; c1 /= cn; c2 /= cn;
    if (r0 * c0 + r1 * c1 + r2 * c2 < 0.5) go = false;
  }
  pout[pid] = o;
  if (!go) return;
  hso_align_job J;
  J.ref_frame_id = S.ref_frame_id; J.ref_level = S.level; J.type = S.type;
  J.px_ref[0] = S.px[0]; J.px_ref[1] = S.px[1];
  J.f_ref[0] = S.f[0]; J.f_ref[1] = S.f[1]; J.f_ref[2] = S.f[2];
  J.depth = 1. / (double)S.mu;
  J.grad[0] = S.grad[0]; J.grad[1] = S.grad[1];
  J.T_cur_ref = o.Tth;
  J.px_cur[0] = o.px[0]; J.px_cur[1] = o.px[1];
  J.exposure_rat = (float)(Fr.exposure / S.ref_exposure);
  J.kf_gap_lt4 = 1;  // findMatchSeed compensates exposure regardless of the keyframe gap (:472-483)
  jobs[pid].j = J;
  jobs[pid].ref_base = SD.ref_base;
}

// ---- kernel 1c: the pair's record for kernel 2 from the matcher's result
__global__ __launch_bounds__(256) void k_activate_finish(ActConsts C, const ActSeedDev* seeds, const ActPairIn* pin, int n_pairs, const hso_align_out* match, ActPair* pout)
{
  const int pid = blockIdx.x * 256 + threadIdx.x;
  if (pid >= n_pairs) return;
  const hso_seed& S = seeds[pin[pid].seed].s;
  ActPair o = pout[pid];
  o.mo = match[pid];
  o.matched = o.mo.success;
  if (o.matched) {
    double f[3];
    cam2world_dev(C.cam, o.mo.px_cur[0], o.mo.px_cur[1], f);
    o.obs[0] = f[0] / f[2]; o.obs[1] = f[1] / f[2];
    double n0 = o.mo.A_cur_ref[0] * S.grad[0] + o.mo.A_cur_ref[1] * S.grad[1];
    double n1 = o.mo.A_cur_ref[2] * S.grad[0] + o.mo.A_cur_ref[3] * S.grad[1];
    const double nn = sqrt(n0 * n0 + n1 * n1);
    o.normal[0] = n0 / nn; o.normal[1] = n1 / nn;
  }
  pout[pid] = o;
}

// ---- kernel 2: gates + DepthFilter::seedOptimizer, one wavefront per seed, lane = target frame.
// Every lane keeps its target's transform / observation in registers and evaluates its own
// residual and Jacobian; the sums the reference forms in a serial loop over the targets are formed
// in the same order by walking the lanes 0..n-1 with a broadcast each (unmatched lanes contribute
// an exact 0.0), so H, b and the energies carry the reference's rounding.
HSO_DEV double act_bcast(double v, int src)
{
  const int lo = __shfl(__double2loint(v), src), hi = __shfl(__double2hiint(v), src);
  return __hiloint2double(hi, lo);
}
HSO_DEV double act_ordered_sum(double x, int n)
{
  double s = 0;
  for (int i = 0; i < n; i++) s += act_bcast(x, i);
  return s;
}

struct ActLane {   // one target frame of the seed, in this lane's registers
  bool matched;
  Se3 T;
  double obs0, obs1, n0, n1;
};

// this lane's residual at inverse depth `id`: obs - project2d(Tth * f/id); pT = the transformed point
HSO_DEV void act_residual(const hso_seed& S, const ActLane& A, double id, double& r0, double& r1, double pT[3])
{
  const double sc = 1.0 / id;
  se3_apply(A.T, S.f[0] * sc, S.f[1] * sc, S.f[2] * sc, pT[0], pT[1], pT[2]);
  r0 = A.obs0 - pT[0] / pT[2]; r1 = A.obs1 - pT[1] / pT[2];
}

// this lane's robust energy term (0 for an unmatched lane)
HSO_DEV double act_energy_term(const hso_seed& S, const ActLane& A, bool edge, double id, double huberTH)
{
  if (!A.matched) return 0.0;
  double r0, r1, pT[3];
  act_residual(S, A, id, r0, r1, pT);
  if (edge) {
    const double re = A.n0 * r0 + A.n1 * r1;
    const double ab = (double)fabsf((float)re);
    const double hw = ab < huberTH ? 1 : huberTH / ab;
    return re * re * hw;
  }
  const double rd = sqrt(r0 * r0 + r1 * r1);
  const double hw = rd < huberTH ? 1 : huberTH / rd;
  return rd * rd * hw;
}

#define ACT_OPT_WAVES 4
__global__ __launch_bounds__(64 * ACT_OPT_WAVES) void k_activate_opt(ActConsts C, const ActSeedDev* seeds, int n_seeds,
                                                                     const ActPair* pairs, hso_activate_out* outs)
{
  const int lane = threadIdx.x & 63;
  const int sid = blockIdx.x * ACT_OPT_WAVES + (threadIdx.x >> 6);
  if (sid >= n_seeds) return;
  const ActSeedDev& SD = seeds[sid];
  const hso_seed& S = SD.s;
  const int n = SD.count;  // <= HSO_ACTIVATE_MAX_TARGETS = 64
  hso_activate_out o;
  memset(&o, 0, sizeof(o));
  o.is_valid = -1;
  o.opt_id = (double)S.mu;
  ActLane A;
  A.matched = false; A.obs0 = A.obs1 = A.n0 = A.n1 = 0;
  A.T.qx = A.T.qy = A.T.qz = 0; A.T.qw = 1; A.T.tx = A.T.ty = A.T.tz = 0;
  bool is_target = false;
  double err = 0;  // this target's drift (:791-806)
  const bool edge = S.type == HSO_FTR_EDGELET;
  if (lane < n) {
    const ActPair& P = pairs[SD.first + lane];
    is_target = P.is_target != 0;
    A.matched = P.matched != 0;
    if (A.matched) {
      A.T = se3_from(P.Tth);
      A.obs0 = P.obs[0]; A.obs1 = P.obs[1]; A.n0 = P.normal[0]; A.n1 = P.normal[1];
      const double d0 = P.px[0] - P.mo.px_cur[0], d1 = P.px[1] - P.mo.px_cur[1];
      err = edge ? fabs(A.n0 * d0 + A.n1 * d1) : sqrt(d0 * d0 + d1 * d1);
      err /= (double)(1 << P.mo.search_level);
    }
  }
  const int n_targets = __popcll(__ballot(is_target)), n_res = __popcll(__ballot(A.matched));
  o.n_targets = n_targets;
  float n_frame_thresh = (float)((double)seeds[sid].n_mean_converge_frame * 0.7);
  if (n_frame_thresh > 8) n_frame_thresh = 8;
  if (n_frame_thresh < 3) n_frame_thresh = 3;
  if ((float)n_targets < n_frame_thresh) { if (lane == 0) outs[sid] = o; return; }
  o.n_matched = n_res;
  double distMean = act_ordered_sum(err, n);
  if ((float)n_res < n_frame_thresh) { if (lane == 0) outs[sid] = o; return; }
  distMean /= (double)n_res;
  o.dist_mean = distMean;
  if ((!edge && distMean > 3.2) || (edge && distMean > 2.5)) { o.is_valid = 0; if (lane == 0) outs[sid] = o; return; }
  o.is_valid = 1;
  if ((!edge && distMean > 2.5) || (edge && distMean > 2.0)) { if (lane == 0) outs[sid] = o; return; }

  // ---- seedOptimizer (:853-1073)
  double old_id = (double)S.mu;
  // MAD scale: 1.4826f * the element of rank floor(n/2) of the float |residual|s (robust_cost.cpp:67-74),
  // by rank counting: every lane ranks its own value against the others
  double huberTH;
  {
    float e = 0;
    if (A.matched) {
      double r0, r1, pT[3];
      act_residual(S, A, old_id, r0, r1, pT);
      e = edge ? (float)fabs(A.n0 * r0 + A.n1 * r1) : (float)sqrt(r0 * r0 + r1 * r1);
    }
    int lt = 0, le = 0;
    for (int j = 0; j < n; j++) {
      const float ej = __shfl(e, j);
      const int mj = __shfl(A.matched ? 1 : 0, j);
      lt += (mj && ej < e) ? 1 : 0; le += (mj && ej <= e) ? 1 : 0;
    }
    const int k = n_res / 2;
    const unsigned long long hit = __ballot(A.matched && lt <= k && k < le);
    const float med = hit ? __shfl(e, __ffsll((long long)hit) - 1) : 0.0f;
    huberTH = (double)(1.4826f * med);
  }
  o.huber = huberTH;
  double oldEnergy = act_ordered_sum(act_energy_term(S, A, edge, old_id, huberTH), n);
  double rho = 0, mu = 0.1, nu = 2.0;
  bool stop = false;
  int iter;
  for (iter = 0; iter < 5; ++iter) {
    int n_trials = 0;
    do {
      double new_id = old_id, newEnergy = 0;
      double h_i = 0, b_i = 0;  // this lane's contributions
      if (A.matched) {
        double r0, r1, pT[3];
        act_residual(S, A, old_id, r0, r1, pT);
        // Point::jacobian_id2uv(pTarget, Tth, old_id, f), point.h:174-184
        double R[9];
        so3_matrix(A.T, R);
        const double Rf2 = R[6] * S.f[0] + R[7] * S.f[1] + R[8] * S.f[2];
        const double J0 = -(A.T.tx - (pT[0] / pT[2]) * A.T.tz) / (Rf2 + A.T.tz * old_id);
        const double J1 = -(A.T.ty - (pT[1] / pT[2]) * A.T.tz) / (Rf2 + A.T.tz * old_id);
        if (edge) {
          const double re = A.n0 * r0 + A.n1 * r1;
          const double ab = (double)fabsf((float)re);
          const double hw = ab < huberTH ? 1 : huberTH / ab;
          const double JE = A.n0 * J0 + A.n1 * J1;
          h_i = JE * JE * hw;
          b_i = JE * re * hw;
        } else {
          const double rd = sqrt(r0 * r0 + r1 * r1);
          const double hw = rd < huberTH ? 1 : huberTH / rd;
          h_i = (J0 * J0 + J1 * J1) * hw;
          b_i = (J0 * r0 + J1 * r1) * hw;
        }
      }
      double Hh = 0, b = 0;
      for (int i = 0; i < n; i++) { Hh += act_bcast(h_i, i); b -= act_bcast(b_i, i); }  // H += ..., b -= ... (:951-967)
      Hh *= 1.0 + mu;
      const double step = b / Hh;
      if (!isnan(step)) {
        new_id = old_id + step;
        newEnergy = act_ordered_sum(act_energy_term(S, A, edge, new_id, huberTH), n);
        rho = oldEnergy - newEnergy;
      } else {
        rho = -1;
      }
      if (rho > 0) {
        oldEnergy = newEnergy;
        old_id = new_id;
        o.opt_id = new_id;
        stop = (double)fabsf((float)step) < 0.00001 * new_id;
        const double c = 1. - pow(2 * rho - 1, 3);
        const double m = c < 2. / 3. ? c : 2. / 3.;
        mu *= (1. / 3. > m ? 1. / 3. : m);
        nu = 2.;
      } else {
        mu *= nu;
        nu *= 2.;
        ++n_trials;
        if (n_trials >= 5) stop = true;
      }
    } while (!(rho > 0 || stop));
    if (stop) break;
  }
  o.energy = oldEnergy;
  o.n_iter = iter < 5 ? iter + 1 : 5;
  o.activated = 1;
  if (lane == 0) outs[sid] = o;
}

// the (seed, target frame) pairs on the device: prep -> the shared matcher -> finish; d_pout receives the pairs' records.
// The work area `d` must hold [seeds | pairs in | frames | pairs out | jobs | match] as laid out by act_layout.
struct ActLayout { size_t o_seeds, o_pin, o_frames, o_pout, o_jobs, o_match, o_out, need; };
static ActLayout act_layout(int n_seeds, int n_pairs, int n_frames)
{
  auto al = [](size_t v) { return (v + 255) & ~size_t(255); };
  ActLayout L;
  L.o_seeds = 0;
  L.o_pin = al((size_t)n_seeds * sizeof(ActSeedDev));
  L.o_frames = L.o_pin + al((size_t)std::max(n_pairs, 1) * sizeof(ActPairIn));
  L.o_pout = L.o_frames + al((size_t)std::max(n_frames, 1) * sizeof(ActFrameDev));
  L.o_jobs = L.o_pout + al((size_t)std::max(n_pairs, 1) * sizeof(ActPair));
  L.o_match = L.o_jobs + al((size_t)std::max(n_pairs, 1) * sizeof(AlignJobDev));
  L.o_out = L.o_match + al((size_t)std::max(n_pairs, 1) * sizeof(hso_align_out));
  L.need = L.o_out + al((size_t)n_seeds * sizeof(hso_activate_out));
  return L;
}
static int act_match_pairs(hso_gpu_ctx* ctx, const hso_camera* cam, const PyrGeom& g, double z_min, char* d, const ActLayout& L, int n_pairs)
{
  if (n_pairs <= 0) return HSO_OK;
  ActConsts C;
  C.cam = *cam; C.g = g; C.z_min = z_min;
  const ActSeedDev* d_seeds = reinterpret_cast<const ActSeedDev*>(d + L.o_seeds);
  const ActPairIn* d_pin = reinterpret_cast<const ActPairIn*>(d + L.o_pin);
  ActPair* d_pout = reinterpret_cast<ActPair*>(d + L.o_pout);
  AlignJobDev* d_jobs = reinterpret_cast<AlignJobDev*>(d + L.o_jobs);
  hso_align_out* d_match = reinterpret_cast<hso_align_out*>(d + L.o_match);
  HSO_HIP_CHECK(ctx, hipMemsetAsync(d_match, 0, (size_t)n_pairs * sizeof(hso_align_out), ctx->stream));
  hipLaunchKernelGGL(k_activate_prep, dim3((n_pairs + 255) / 256), dim3(256), 0, ctx->stream, C, d_seeds, d_pin, reinterpret_cast<const ActFrameDev*>(d + L.o_frames), n_pairs,
                     d_pout, d_jobs);
  HSO_HIP_CHECK(ctx, hipGetLastError());
  if (int rc = hso_align_launch_sparse(ctx, cam, g, 0.8f, d_jobs, n_pairs, d_match)) return rc;
  hipLaunchKernelGGL(k_activate_finish, dim3((n_pairs + 255) / 256), dim3(256), 0, ctx->stream, C, d_seeds, d_pin, n_pairs, d_match, d_pout);
  HSO_HIP_CHECK(ctx, hipGetLastError());
  return HSO_OK;
}

// target_frame: per pair the index into `frames` (the unique target frames of the call); n_mean_per_seed == nullptr: every seed
// uses n_mean_all
static int seed_activate_impl(hso_gpu_ctx* ctx, const hso_camera* cam, const hso_seed* seeds, int n_seeds,
                              const int32_t* target_begin, const int32_t* target_frame, const hso_activate_target* frames, int n_frames,
                              const int32_t* n_mean_per_seed, int n_mean_all, hso_activate_out* out, hso_align_out* match_out)
{
  if (!ctx) return HSO_E_INVALID;
  if (!cam || n_seeds < 0 || n_frames < 0 || (n_seeds > 0 && (!seeds || !target_begin || !out))) return hso_fail(ctx, HSO_E_INVALID, "seed_activate: bad argument");
  if (n_seeds == 0) return HSO_OK;
  HSO_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  const int n_pairs = target_begin[n_seeds];
  if (target_begin[0] != 0 || n_pairs < 0 || (n_pairs > 0 && (!frames || !target_frame))) return hso_fail(ctx, HSO_E_INVALID, "seed_activate: bad target ranges");
  const ActLayout L = act_layout(n_seeds, n_pairs, n_frames);
  // the call's tables are assembled in page-locked staging (one DMA): [seeds | pairs | frames]
  char* h = hso_pinned(ctx, 0, L.o_pout);
  if (!h) return HSO_E_NOMEM;
  ActSeedDev* hs = reinterpret_cast<ActSeedDev*>(h + L.o_seeds);
  ActPairIn* hp = reinterpret_cast<ActPairIn*>(h + L.o_pin);
  ActFrameDev* hf = reinterpret_cast<ActFrameDev*>(h + L.o_frames);
  PyrGeom g;
  bool have_g = false;
  for (int k = 0; k < n_frames; k++) {
    auto itt = ctx->frames.find(frames[k].frame_id);
    if (itt == ctx->frames.end()) return hso_fail(ctx, HSO_E_NOFRAME, "seed_activate: target frame not resident");
    if (!have_g) { g = itt->second.g; have_g = true; }
    if (!same_geom(itt->second.g, g)) return hso_fail(ctx, HSO_E_INVALID, "seed_activate: frames must share one size");
    hf[k].base = itt->second.base; hf[k].T_f_w = frames[k].T_f_w; hf[k].exposure = frames[k].exposure;
  }
  int64_t last_id = -1; const uint8_t* last_base = nullptr;
  for (int i = 0; i < n_seeds; i++) {
    const int b = target_begin[i], e = target_begin[i + 1];
    if (e < b || e - b > HSO_ACTIVATE_MAX_TARGETS) return hso_fail(ctx, HSO_E_INVALID, "seed_activate: a seed has more than HSO_ACTIVATE_MAX_TARGETS targets");
    if (seeds[i].ref_frame_id != last_id || !last_base) {         // runs of seeds share a host keyframe
      auto itr = ctx->frames.find(seeds[i].ref_frame_id);
      if (itr == ctx->frames.end()) return hso_fail(ctx, HSO_E_NOFRAME, "seed_activate: seed host frame not resident");
      if (!have_g) { g = itr->second.g; have_g = true; }
      if (!same_geom(itr->second.g, g)) return hso_fail(ctx, HSO_E_INVALID, "seed_activate: frames must share one size");
      last_id = seeds[i].ref_frame_id; last_base = itr->second.base;
    }
    if (seeds[i].level < 0 || seeds[i].level >= HSO_N_PYR_LEVELS) return hso_fail(ctx, HSO_E_INVALID, "seed_activate: bad level");
    hs[i].ref_base = last_base; hs[i].s = seeds[i]; hs[i].first = b; hs[i].count = e - b;
    hs[i].n_mean_converge_frame = n_mean_per_seed ? n_mean_per_seed[i] : n_mean_all; hs[i]._pad = 0;
    for (int k = b; k < e; k++) {
      if (target_frame[k] < 0 || target_frame[k] >= n_frames) return hso_fail(ctx, HSO_E_INVALID, "seed_activate: target frame index out of range");
      hp[k].seed = i; hp[k].frame = target_frame[k];
    }
  }
  if (cam->width != g.w[0] || cam->height != g.h[0]) return hso_fail(ctx, HSO_E_INVALID, "seed_activate: camera size differs from the frame size");
  if (ctx->batch_cap < L.need) {
    HSO_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->d_batch) (void)hipFree(ctx->d_batch);
    ctx->d_batch = nullptr; ctx->batch_cap = 0;
    HSO_HIP_CHECK(ctx, hipMalloc(reinterpret_cast<void**>(&ctx->d_batch), hso_grown(L.need)));
    ctx->batch_cap = hso_grown(L.need);
  }
  char* d = ctx->d_batch;
  HSO_HIP_CHECK(ctx, hipMemcpyAsync(d, h, L.o_pout, hipMemcpyHostToDevice, ctx->stream));
  if (int rc = act_match_pairs(ctx, cam, g, 0.0001, d, L, n_pairs)) return rc;
  ActConsts C;
  C.cam = *cam; C.g = g; C.z_min = 0.0001;
  const ActSeedDev* d_seeds = reinterpret_cast<const ActSeedDev*>(d + L.o_seeds);
  ActPair* d_pout = reinterpret_cast<ActPair*>(d + L.o_pout);
  hso_activate_out* d_out = reinterpret_cast<hso_activate_out*>(d + L.o_out);
  hipLaunchKernelGGL(k_activate_opt, dim3((n_seeds + ACT_OPT_WAVES - 1) / ACT_OPT_WAVES), dim3(64 * ACT_OPT_WAVES), 0, ctx->stream, C, d_seeds, n_seeds, d_pout, d_out);
  HSO_HIP_CHECK(ctx, hipGetLastError());
  HSO_HIP_CHECK(ctx, hipMemcpyAsync(out, d_out, (size_t)n_seeds * sizeof(hso_activate_out), hipMemcpyDeviceToHost, ctx->stream));
  std::vector<ActPair> hpo;
  if (match_out && n_pairs > 0) {
    hpo.resize((size_t)n_pairs);
    HSO_HIP_CHECK(ctx, hipMemcpyAsync(hpo.data(), d_pout, (size_t)n_pairs * sizeof(ActPair), hipMemcpyDeviceToHost, ctx->stream));
  }
  HSO_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  if (match_out) for (int k = 0; k < n_pairs; k++) match_out[k] = hpo[k].mo;
  return HSO_OK;
}

// ---- activation by slots: the seeds are rows of a resident seed table (hso_seed.hip) ----
struct ActSlotIn { int32_t slot, first, count, n_mean; };
// one thread per seed: its record from the table's row, its pairs' (seed, frame) records from the 16-bit frame indices
static __global__ void k_activate_expand(const char* __restrict__ rows, size_t stride, size_t seed_offset, const ActSlotIn* __restrict__ in,
                                         const uint16_t* __restrict__ pair_frame, int n_seeds, ActSeedDev* __restrict__ seeds, ActPairIn* __restrict__ pairs,
                                         const ActFrameDev* __restrict__ frames_in, ActFrameDev* __restrict__ frames_out, int n_frames)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n_frames) frames_out[i] = frames_in[i];                 // the frame table moves to its place in the layout
  if (i >= n_seeds) return;
  const ActSlotIn a = in[i];
  const char* row = rows + (size_t)a.slot * stride;
  ActSeedDev r;
  r.ref_base = *reinterpret_cast<const uint8_t* const*>(row);
  r.s = *reinterpret_cast<const hso_seed*>(row + seed_offset);
  r.first = a.first; r.count = a.count; r.n_mean_converge_frame = a.n_mean; r._pad = 0;
  seeds[i] = r;
  for (int k = a.first; k < a.first + a.count; k++) { ActPairIn p; p.seed = i; p.frame = (int32_t)pair_frame[k]; pairs[k] = p; }
}

extern "C" int hso_gpu_seed_table_activate(hso_gpu_ctx* ctx, const hso_camera* cam, int table, const int32_t* slots, int n_seeds, const int32_t* target_begin,
                                           const int32_t* target_frame, const hso_activate_target* frames, int n_frames,
                                           const int32_t* n_mean_converge_frame, hso_activate_out* out)
{
  if (!ctx) return HSO_E_INVALID;
  if (!cam || n_seeds < 0 || n_frames < 0 || n_frames > 65536 || (n_seeds > 0 && (!slots || !target_begin || !n_mean_converge_frame || !out)))
    return hso_fail(ctx, HSO_E_INVALID, "seed_table_activate: bad argument");
  if (n_seeds == 0) return HSO_OK;
  HSO_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  const int n_pairs = target_begin[n_seeds];
  if (target_begin[0] != 0 || n_pairs < 0 || (n_pairs > 0 && (!frames || !target_frame))) return hso_fail(ctx, HSO_E_INVALID, "seed_table_activate: bad target ranges");
  const char* rows = nullptr; size_t stride = 0, seed_offset = 0; PyrGeom g;
  if (int rc = hso_seed_table_rows(ctx, table, slots, n_seeds, &rows, &stride, &seed_offset, &g)) return rc;
  const ActLayout L = act_layout(n_seeds, n_pairs, n_frames);
  // what crosses the bus: [frames | (slot, first, count, n_mean) per seed | 16-bit frame index per pair]
  auto al = [](size_t v) { return (v + 255) & ~size_t(255); };
  const size_t o_in = al((size_t)std::max(n_frames, 1) * sizeof(ActFrameDev)), o_pf = o_in + al((size_t)n_seeds * sizeof(ActSlotIn));
  const size_t up_bytes = o_pf + al((size_t)std::max(n_pairs, 1) * sizeof(uint16_t));
  char* h = hso_pinned(ctx, 0, up_bytes);
  if (!h) return HSO_E_NOMEM;
  ActFrameDev* hf = reinterpret_cast<ActFrameDev*>(h);
  ActSlotIn* hin = reinterpret_cast<ActSlotIn*>(h + o_in);
  uint16_t* hpf = reinterpret_cast<uint16_t*>(h + o_pf);
  for (int k = 0; k < n_frames; k++) {
    auto itt = ctx->frames.find(frames[k].frame_id);
    if (itt == ctx->frames.end()) return hso_fail(ctx, HSO_E_NOFRAME, "seed_table_activate: target frame not resident");
    if (!same_geom(itt->second.g, g)) return hso_fail(ctx, HSO_E_INVALID, "seed_table_activate: frames must share one size");
    hf[k].base = itt->second.base; hf[k].T_f_w = frames[k].T_f_w; hf[k].exposure = frames[k].exposure;
  }
  for (int i = 0; i < n_seeds; i++) {
    const int b = target_begin[i], e = target_begin[i + 1];
    if (e < b || e - b > HSO_ACTIVATE_MAX_TARGETS) return hso_fail(ctx, HSO_E_INVALID, "seed_table_activate: a seed has more than HSO_ACTIVATE_MAX_TARGETS targets");
    hin[i].slot = slots[i]; hin[i].first = b; hin[i].count = e - b; hin[i].n_mean = n_mean_converge_frame[i];
    for (int k = b; k < e; k++) {
      if (target_frame[k] < 0 || target_frame[k] >= n_frames) return hso_fail(ctx, HSO_E_INVALID, "seed_table_activate: target frame index out of range");
      hpf[k] = (uint16_t)target_frame[k];
    }
  }
  if (cam->width != g.w[0] || cam->height != g.h[0]) return hso_fail(ctx, HSO_E_INVALID, "seed_table_activate: camera size differs from the frame size");
  const size_t need = L.need + up_bytes;
  if (ctx->batch_cap < need) {
    HSO_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->d_batch) (void)hipFree(ctx->d_batch);
    ctx->d_batch = nullptr; ctx->batch_cap = 0;
    HSO_HIP_CHECK(ctx, hipMalloc(reinterpret_cast<void**>(&ctx->d_batch), hso_grown(need)));
    ctx->batch_cap = hso_grown(need);
  }
  char* d = ctx->d_batch;
  char* d_up = d + L.need;
  HSO_HIP_CHECK(ctx, hipMemcpyAsync(d_up, h, up_bytes, hipMemcpyHostToDevice, ctx->stream));
  ActSeedDev* d_seeds = reinterpret_cast<ActSeedDev*>(d + L.o_seeds);
  hipLaunchKernelGGL(k_activate_expand, dim3((std::max(n_seeds, n_frames) + 127) / 128), dim3(128), 0, ctx->stream, rows, stride, seed_offset,
                     reinterpret_cast<const ActSlotIn*>(d_up + o_in), reinterpret_cast<const uint16_t*>(d_up + o_pf), n_seeds, d_seeds,
                     reinterpret_cast<ActPairIn*>(d + L.o_pin), reinterpret_cast<const ActFrameDev*>(d_up), reinterpret_cast<ActFrameDev*>(d + L.o_frames), n_frames);
  HSO_HIP_CHECK(ctx, hipGetLastError());
  if (int rc = act_match_pairs(ctx, cam, g, 0.0001, d, L, n_pairs)) return rc;
  ActConsts C;
  C.cam = *cam; C.g = g; C.z_min = 0.0001;
  ActPair* d_pout = reinterpret_cast<ActPair*>(d + L.o_pout);
  hso_activate_out* d_out = reinterpret_cast<hso_activate_out*>(d + L.o_out);
  hipLaunchKernelGGL(k_activate_opt, dim3((n_seeds + ACT_OPT_WAVES - 1) / ACT_OPT_WAVES), dim3(64 * ACT_OPT_WAVES), 0, ctx->stream, C, d_seeds, n_seeds, d_pout, d_out);
  HSO_HIP_CHECK(ctx, hipGetLastError());
  HSO_HIP_CHECK(ctx, hipMemcpyAsync(out, d_out, (size_t)n_seeds * sizeof(hso_activate_out), hipMemcpyDeviceToHost, ctx->stream));
  HSO_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  return HSO_OK;
}

// the per-pair form of the targets: every pair names its own frame record (table = the pairs' records, index = identity)
static int seed_activate_pairs(hso_gpu_ctx* ctx, const hso_camera* cam, const hso_seed* seeds, int n_seeds, const int32_t* target_begin,
                               const hso_activate_target* targets, const int32_t* n_mean_per_seed, int n_mean_all, hso_activate_out* out, hso_align_out* match_out)
{
  if (!ctx) return HSO_E_INVALID;
  if (n_seeds < 0 || (n_seeds > 0 && !target_begin)) return hso_fail(ctx, HSO_E_INVALID, "seed_activate: bad argument");
  const int n_pairs = n_seeds > 0 ? target_begin[n_seeds] : 0;
  if (n_pairs < 0 || (n_pairs > 0 && !targets)) return hso_fail(ctx, HSO_E_INVALID, "seed_activate: bad target ranges");
  std::vector<int32_t> ix((size_t)std::max(n_pairs, 0));
  for (int k = 0; k < n_pairs; k++) ix[(size_t)k] = k;
  return seed_activate_impl(ctx, cam, seeds, n_seeds, target_begin, ix.data(), targets, n_pairs, n_mean_per_seed, n_mean_all, out, match_out);
}

extern "C" int hso_gpu_seed_activate(hso_gpu_ctx* ctx, const hso_camera* cam, const hso_seed* seeds, int n_seeds,
                                     const int32_t* target_begin, const hso_activate_target* targets, int n_mean_converge_frame,
                                     hso_activate_out* out, hso_align_out* match_out)
{
  return seed_activate_pairs(ctx, cam, seeds, n_seeds, target_begin, targets, nullptr, n_mean_converge_frame, out, match_out);
}

extern "C" int hso_gpu_seed_activate_multi(hso_gpu_ctx* ctx, const hso_camera* cam, const hso_seed* seeds, int n_seeds,
                                           const int32_t* target_begin, const hso_activate_target* targets,
                                           const int32_t* n_mean_converge_frame, hso_activate_out* out, hso_align_out* match_out)
{
  if (ctx && n_seeds > 0 && !n_mean_converge_frame) return hso_fail(ctx, HSO_E_INVALID, "seed_activate_multi: null n_mean_converge_frame");
  return seed_activate_pairs(ctx, cam, seeds, n_seeds, target_begin, targets, n_mean_converge_frame, 0, out, match_out);
}

extern "C" int hso_gpu_seed_activate_frames(hso_gpu_ctx* ctx, const hso_camera* cam, const hso_seed* seeds, int n_seeds, const int32_t* target_begin,
                                            const int32_t* target_frame, const hso_activate_target* frames, int n_frames,
                                            const int32_t* n_mean_converge_frame, hso_activate_out* out)
{
  if (ctx && n_seeds > 0 && !n_mean_converge_frame) return hso_fail(ctx, HSO_E_INVALID, "seed_activate_frames: null n_mean_converge_frame");
  return seed_activate_impl(ctx, cam, seeds, n_seeds, target_begin, target_frame, frames, n_frames, n_mean_converge_frame, 0, out, nullptr);
}

// Reprojector::reprojectorSeed (reference src/reprojector.cpp:504-529 for seeds: :531-554) + Matcher::findMatchSeed
// (src/matcher.cpp:442-518) of every given seed in one current frame: the seed branch of reprojectMap (:309-329).
extern "C" int hso_gpu_seed_reproject_match(hso_gpu_ctx* ctx, const hso_camera* cam, int64_t cur_frame_id, const hso_se3* T_cur_w,
                                            double cur_exposure, const hso_seed* seeds, int n_seeds, int cell_size, int grid_n_cols,
                                            hso_reproj_point* proj_out, hso_align_out* match_out)
{
  if (!ctx) return HSO_E_INVALID;
  if (!cam || !T_cur_w || n_seeds < 0 || cell_size <= 0 || grid_n_cols <= 0 || (n_seeds > 0 && (!seeds || !proj_out || !match_out)))
    return hso_fail(ctx, HSO_E_INVALID, "seed_reproject_match: bad argument");
  if (n_seeds == 0) return HSO_OK;
  HSO_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  auto itc = ctx->frames.find(cur_frame_id);
  if (itc == ctx->frames.end()) return hso_fail(ctx, HSO_E_NOFRAME, "seed_reproject_match: current frame not resident");
  const PyrGeom g = itc->second.g;
  if (cam->width != g.w[0] || cam->height != g.h[0]) return hso_fail(ctx, HSO_E_INVALID, "seed_reproject_match: camera size differs from the frame size");
  const ActLayout L = act_layout(n_seeds, n_seeds, 1);
  char* h = hso_pinned(ctx, 0, L.o_pout);
  if (!h) return HSO_E_NOMEM;
  ActSeedDev* hs = reinterpret_cast<ActSeedDev*>(h + L.o_seeds);
  ActPairIn* hp = reinterpret_cast<ActPairIn*>(h + L.o_pin);
  ActFrameDev* hf = reinterpret_cast<ActFrameDev*>(h + L.o_frames);
  hf[0].base = itc->second.base; hf[0].T_f_w = *T_cur_w; hf[0].exposure = cur_exposure;
  for (int i = 0; i < n_seeds; i++) {
    auto itr = ctx->frames.find(seeds[i].ref_frame_id);
    if (itr == ctx->frames.end()) return hso_fail(ctx, HSO_E_NOFRAME, "seed_reproject_match: seed host frame not resident");
    if (itr->second.g.w[0] != g.w[0] || itr->second.g.h[0] != g.h[0]) return hso_fail(ctx, HSO_E_INVALID, "seed_reproject_match: frames must share one size");
    if (seeds[i].level < 0 || seeds[i].level >= HSO_N_PYR_LEVELS) return hso_fail(ctx, HSO_E_INVALID, "seed_reproject_match: bad level");
    hs[i].ref_base = itr->second.base; hs[i].s = seeds[i]; hs[i].first = i; hs[i].count = 1; hs[i].n_mean_converge_frame = 0; hs[i]._pad = 0;
    hp[i].seed = i; hp[i].frame = 0;
  }
  if (ctx->batch_cap < L.need) {
    HSO_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->d_batch) (void)hipFree(ctx->d_batch);
    ctx->d_batch = nullptr; ctx->batch_cap = 0;
    HSO_HIP_CHECK(ctx, hipMalloc(reinterpret_cast<void**>(&ctx->d_batch), hso_grown(L.need)));
    ctx->batch_cap = hso_grown(L.need);
  }
  char* d = ctx->d_batch;
  HSO_HIP_CHECK(ctx, hipMemcpyAsync(d, h, L.o_pout, hipMemcpyHostToDevice, ctx->stream));
  if (int rc = act_match_pairs(ctx, cam, g, 0.001, d, L, n_seeds)) return rc;
  std::vector<ActPair> hpo(n_seeds);
  HSO_HIP_CHECK(ctx, hipMemcpyAsync(hpo.data(), d + L.o_pout, (size_t)n_seeds * sizeof(ActPair), hipMemcpyDeviceToHost, ctx->stream));
  HSO_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  for (int i = 0; i < n_seeds; i++) {
    hso_reproj_point r{};
    r.projected = hpo[i].is_target;
    r.px[0] = hpo[i].px[0]; r.px[1] = hpo[i].px[1];
    r.cell = r.projected ? (int)(r.px[1] / cell_size) * grid_n_cols + (int)(r.px[0] / cell_size) : 0;
    r.ref_obs = -1;
    proj_out[i] = r;
    match_out[i] = hpo[i].mo;
  }
  return HSO_OK;
}
