// hso_pose.hip — motion-only Levenberg-Marquardt pose refinement on gfx950, batched over
// independent frames.
//
// Replaces pose_optimizer::optimizeLevenbergMarquardt3rd (reference
// src/pose_optimizer.cpp:399-771): unit-plane reprojection residuals of every feature with a
// point (2-D for corners, 1-D along the gradient for edgelets, scaled by 1/2^level), MAD scales
// (src/vikit/robust_cost.cpp:67-74), Huber weights (:141-148, k = 1.345), LM with multiplicative
// damping A += diag(A)*mu, the mu/nu schedule of :644-674, covariance, outlier culling and the
// median-based error statistics.
//
// MI355X mapping: one 256-thread workgroup per frame, the whole optimisation resident on the
// device (<= 12 iterations x <= 5 trials, each two passes over <= a few thousand features):
// this stage is latency-bound by construction, so the design goal is zero host round trips and
// many frames in flight, not bandwidth.  Per-feature arithmetic follows the reference's fp64
// expressions; the sums (chi2, A, b) are fixed-tree reductions (reference: serial fp64), the
// MAD scales and medians are exact order statistics (bitwise search on the IEEE bit patterns).
#include "hso_pose_dev.h"
#include "hso_dev_math.h"
#include "hso_wave_reduce.h"
#include <string.h>
#include <algorithm>
#include <vector>

using namespace hso_dev;

#define POSE_THREADS 512
#define POSE_WAVES (POSE_THREADS / 64)
#define POSE_MAX_FEATS HSO_POSE_MAX_FEATS
#define POSE_MAX_POSES HSO_POSE_MAX_POSES

struct PoseShared {
  Se3 T, Tn;
  Se3 hinv[POSE_MAX_POSES];
  Se3 Tth[POSE_MAX_POSES];
  double red[32];      // the 27 sums of the normal equations at s.T (undamped), valid from the first evaluation on
  double red_n[32];    // the same at the trial pose s.Tn; [27] = the trial's chi2
  double wave_part[POSE_WAVES][32];
  double A[36], b[6], dT[8];
  double chi2, new_chi2, mu, nu, rho;
  float scale_pt, scale_ls;
  int n_pt, n_ls, n_obs, stop, accept, n_trials, iters, n_trials_total, n_deleted;
  int iter_cur, done, nan_step;               // the LM loop's control words (written by the first wavefront between two barriers)
  int cnt[POSE_WAVES];
  unsigned hist[3][256];                    // radix select: digit histograms (three order statistics at a time)
  unsigned sel_bin[3], sel_rank[3], sel_cnt[3];
  unsigned long long found;
};

HSO_DEV double p_readlane_d(double v, int src)
{
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), src), hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
  return __hiloint2double(hi, lo);
}

// workgroup sum of 27 doubles per thread: one halving exchange per wave (32 slots, 5 of them zero pads: ~40 lane exchanges
// instead of 27 butterflies = 162), LDS across the waves in wave order => deterministic.  Result in dst[0..27);
// plus the workgroup sum of one more double into dst[27], by the butterfly + the sum of the waves' totals in wave order (the
// chi2 keeps the bits it had when it was a pass of its own).
HSO_DEV void pose_block_sum27_1(PoseShared& s, double (&v)[32], double c, double* dst)
{
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int slot;
  const double x = wave_reduce_scatter32(v, lane, slot);
  const double cw = wave_butterfly_sum(c);
  if ((lane & 1) == 0 && slot < 27) s.wave_part[wave][slot] = x;
  if (lane == 0) s.wave_part[wave][27] = cw;
  __syncthreads();
  if (threadIdx.x < 28) {
    double t = 0;
    for (int w = 0; w < POSE_WAVES; w++) t += s.wave_part[w][threadIdx.x];
    dst[threadIdx.x] = t;
  }
  __syncthreads();
}

HSO_DEV int pose_block_count(PoseShared& s, int v)
{
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  v = wave_butterfly_sum(v);
  __syncthreads();
  if (lane == 0) s.cnt[wave] = v;
  __syncthreads();
  int t = 0;
  for (int w = 0; w < POSE_WAVES; w++) t += s.cnt[w];
  return t;
}

// The bin of a 256-bin histogram that holds rank `rank` (0-based), found by the four wavefronts [4 * half, 4 * half + 4): inclusive scan
// inside each wavefront, then across the four.  Writes s.sel_bin / sel_rank / sel_cnt [which].  All 512 threads call it (the
// barrier inside is the workgroup's); `half` selects which four wavefronts work on this histogram.
HSO_DEV void pose_pick_bin(PoseShared& s, int which, int half, unsigned rank)
{
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const bool mine = (wave >> 2) == half;
  const int bin = tid & 255, w4 = wave & 3;
  unsigned c = 0, incl = 0;
  if (mine) {
    c = s.hist[which][bin];
    incl = c;
    for (int d = 1; d < 64; d <<= 1) { const unsigned o = __shfl_up(incl, d); if (lane >= d) incl += o; }
    if (lane == 63) s.cnt[wave] = (int)incl;
  }
  __syncthreads();
  if (mine) {
    for (int w = 0; w < w4; w++) incl += (unsigned)s.cnt[(half << 2) + w];
    if (rank >= incl - c && rank < incl) { s.sel_bin[which] = (unsigned)bin; s.sel_rank[which] = rank - (incl - c); s.sel_cnt[which] = c; }
  }
}

// Three order statistics in one sweep of radix passes over keys the threads hold in REGISTERS: the element of rank r0 among all
// valid k0 keys (the initial median, :441-455), of rank r1 among the k1 keys of corner features and of rank r2 among the k1 keys
// of edgelet features (the two MAD scales, :459-483).  Exact (the values nth_element would leave at those positions); a select
// whose count is 0 returns garbage that the caller ignores.  Before: three selects one after the other, each over keys written
// to LDS first (12 radix passes, ~60 barriers); now 4 passes.
template <int FPT>
HSO_DEV void pose_select3(PoseShared& s, const unsigned (&k0)[FPT], const unsigned (&k1)[FPT], const int (&kind)[FPT], unsigned r0, unsigned r1, unsigned r2,
                          unsigned (&out)[3])
{
  const int tid = threadIdx.x;
  unsigned prefix[3] = {0, 0, 0}, rank[3] = {r0, r1, r2};
  for (int shift = 24; shift >= 0; shift -= 8) {
    __syncthreads();
    for (int i = tid; i < 768; i += POSE_THREADS) (&s.hist[0][0])[i] = 0;
    __syncthreads();
    const unsigned hi_mask = shift == 24 ? 0u : (~0u << (shift + 8));
#pragma unroll
    for (int q = 0; q < FPT; q++) {
      if (!(kind[q] & 3)) continue;
      if ((k0[q] & hi_mask) == prefix[0]) atomicAdd(&s.hist[0][(k0[q] >> shift) & 255u], 1u);
      const int w = (kind[q] & 3) == 2 ? 2 : 1;
      if ((k1[q] & hi_mask) == prefix[w]) atomicAdd(&s.hist[w][(k1[q] >> shift) & 255u], 1u);
    }
    __syncthreads();
    pose_pick_bin(s, 0, 0, rank[0]);       // wavefronts 0..3
    pose_pick_bin(s, 1, 1, rank[1]);       // wavefronts 4..7, beside the first (the barrier inside each is shared by all)
    pose_pick_bin(s, 2, 0, rank[2]);
    __syncthreads();
#pragma unroll
    for (int w = 0; w < 3; w++) { prefix[w] |= s.sel_bin[w] << shift; rank[w] = s.sel_rank[w]; }
  }
  out[0] = prefix[0]; out[1] = prefix[1]; out[2] = prefix[2];
}

// One order statistic of 64-bit keys held in registers, MSB first, 8 bits per pass; stops as soon as the bin that holds the rank
// holds a single key (squared errors of a frame: after three or four of the eight passes) — that key is then fetched from its owner.
template <int FPT>
HSO_DEV unsigned long long pose_select64(PoseShared& s, const unsigned long long (&k)[FPT], const bool (&valid)[FPT], unsigned r)
{
  const int tid = threadIdx.x;
  unsigned long long prefix = 0;
  unsigned rank = r;
  for (int shift = 56; shift >= 0; shift -= 8) {
    __syncthreads();
    if (tid < 256) s.hist[0][tid] = 0;
    __syncthreads();
    const unsigned long long hi_mask = shift == 56 ? 0ull : (~0ull << (shift + 8));
#pragma unroll
    for (int q = 0; q < FPT; q++)
      if (valid[q] && (k[q] & hi_mask) == prefix) atomicAdd(&s.hist[0][(unsigned)(k[q] >> shift) & 255u], 1u);
    __syncthreads();
    pose_pick_bin(s, 0, 0, rank);
    __syncthreads();
    prefix |= (unsigned long long)s.sel_bin[0] << shift;
    rank = s.sel_rank[0];
    if (s.sel_cnt[0] == 1 && shift > 0) {
      // one key left under this prefix: it is the answer
      const unsigned long long m = ~0ull << shift;
#pragma unroll
      for (int q = 0; q < FPT; q++) if (valid[q] && (k[q] & m) == prefix) s.found = k[q];
      __syncthreads();
      return s.found;
    }
  }
  return prefix;
}

struct Resid { double e0, e1, px, py, pz; };

// What a thread keeps of one of its features for the whole optimisation (registers): the point in its host frame, the
// observation on the unit plane, the level scale, the edgelet direction.  The reference recomputes host_f / idist and
// f.x / f.z, f.y / f.z in every pass (:429-440); the values are the same bits each time, so they are formed once.
struct PoseFeatReg {
  double X0, X1, X2;     // host_f * (1 / idist)
  double u, v;           // f[0] / f[2], f[1] / f[2]
  double sc;             // 1 / 2^level
  double g0, g1;         // grad
  int host_pose;
  int kind;              // 0 no point, 1 corner (2-D residual), 2 edgelet (1-D residual); bit 2: temporary point (weight x 0.5)
};

HSO_DEV PoseFeatReg pose_load_feat(const hso_pose_feat& ft)
{
  PoseFeatReg r;
  r.kind = 0; r.host_pose = 0;
  r.X0 = r.X1 = r.X2 = 0; r.u = r.v = 0; r.sc = 1; r.g0 = r.g1 = 0;
  if (ft.has_point) {
    const double inv = 1.0 / ft.idist;
    r.X0 = ft.host_f[0] * inv; r.X1 = ft.host_f[1] * inv; r.X2 = ft.host_f[2] * inv;
    r.u = ft.f[0] / ft.f[2]; r.v = ft.f[1] / ft.f[2];
    r.sc = 1.0 / (double)(1 << ft.level);
    r.g0 = ft.grad[0]; r.g1 = ft.grad[1];
    r.host_pose = ft.host_pose;
    r.kind = (ft.type == HSO_FTR_EDGELET ? 2 : 1) | (ft.temporary ? 4 : 0);
  }
  return r;
}

HSO_DEV Resid pose_residual(const PoseShared& s, const PoseFeatReg& f)
{
  // pTarget = (T * host^-1) * (host_f / idist); e = project2d(f) - project2d(pTarget), / 2^level (:429-440)
  Resid r;
  se3_apply(s.Tth[f.host_pose], f.X0, f.X1, f.X2, r.px, r.py, r.pz);
  r.e0 = (f.u - r.px / r.pz) * f.sc;
  r.e1 = (f.v - r.py / r.pz) * f.sc;
  return r;
}

// HuberWeightFunction::value(const float&), robust_cost.cpp:141-148, k = 1.345f
HSO_DEV double huber_w(double t_over_scale)
{
  const float t = (float)t_over_scale;
  const float t_abs = fabsf(t);
  return (t_abs < 1.345f) ? 1.0 : (double)(1.345f / t_abs);
}

HSO_DEV void pose_set_Tth(PoseShared& s, const Se3& T, int n_poses)
{
  __syncthreads();
  if ((int)threadIdx.x < n_poses) s.Tth[threadIdx.x] = se3_mul(T, s.hinv[threadIdx.x]);
  __syncthreads();
}

// A.ldlt().solve(b) for the 6x6 system in s.A/s.b (pivoted LDL^T on eight lanes, broadcasts by
// v_readlane_b32; same scheme as the tracker's 7x7 solve).  Result in s.dT[0..5].
// a[i]: lane j < 6 holds A(i, j), lanes >= 6 hold b(i) (lane 6 is the one that is read)
HSO_DEV void pose_ldlt6(PoseShared& s, double (&a)[6])
{
  const int lane = threadIdx.x & 63;
  int perm[6];
#pragma unroll
  for (int i = 0; i < 6; i++) perm[i] = i;
#pragma unroll
  for (int k = 0; k < 6; k++) {
    double best = -1;
    int idx = k;
#pragma unroll
    for (int q = k; q < 6; q++) {
      const double d = fabs(p_readlane_d(a[q], q));
      if (d > best) { best = d; idx = q; }
    }
#pragma unroll
    for (int q = k + 1; q < 6; q++) {
      if (idx == q) {
        const double t = a[k]; a[k] = a[q]; a[q] = t;
        const int tp = perm[k]; perm[k] = perm[q]; perm[q] = tp;
#pragma unroll
        for (int i = 0; i < 6; i++) {
          const double from_q = p_readlane_d(a[i], q), from_k = p_readlane_d(a[i], k);
          a[i] = (lane == k) ? from_q : ((lane == q) ? from_k : a[i]);
        }
      }
    }
    const double akk = p_readlane_d(a[k], k);
    const bool valid = fabs(akk) > 0;
#pragma unroll
    for (int i = k + 1; i < 6; i++) {
      const double aik = p_readlane_d(a[i], k);
      const double lik = valid ? aik / akk : aik;
      if (lane > k) a[i] -= lik * a[k];
      else if (lane == k) a[i] = lik;
    }
  }
  const double tolerance = 1.0 / 1.7976931348623157e308;
  double x[6];
#pragma unroll
  for (int i = 0; i < 6; i++) {
    const double dii = p_readlane_d(a[i], i), yi = p_readlane_d(a[i], 6);
    x[i] = (fabs(dii) > tolerance) ? yi / dii : 0.0;
  }
#pragma unroll
  for (int k = 5; k >= 1; k--) {
#pragma unroll
    for (int i = 0; i < k; i++) x[i] -= p_readlane_d(a[k], i) * x[k];
  }
  if (lane == 0) {
#pragma unroll
    for (int i = 0; i < 6; i++) {
#pragma unroll
      for (int q = 0; q < 6; q++)
        if (perm[i] == q) s.dT[q] = x[i];
    }
  }
}

// One pass over the thread's features at the poses in s.Tth: the 27 sums of the normal equations (:545-592) into acc and,
// with CHI2, the weighted chi2 (:602-641) of the same residuals.  A trial evaluates both at its pose in one pass: if it is
// accepted (the common case) the sums ARE the next iteration's normal equations, if it is rejected the sums at the
// unchanged pose are still in s.red, so after the first evaluation a trial costs one pass instead of two and one pose
// table instead of two.  Term by term the operations are those of the two separate passes: results are bit-identical.
template <int FPT, bool CHI2>
HSO_DEV void pose_normal_pass(const PoseShared& s, const PoseFeatReg (&pf)[FPT], double (&acc)[32], double& chi2)
{
#pragma unroll
  for (int q = 0; q < 32; q++) acc[q] = 0;
  chi2 = 0;
#pragma unroll
  for (int q = 0; q < FPT; q++) {
    const PoseFeatReg& f = pf[q];
    if (!(f.kind & 3)) continue;
    const Resid r = pose_residual(s, f);
    double J0[6], J1[6];
    jacobian_xyz2uv(r.px, r.py, r.pz, J0, J1);
#pragma unroll
    for (int k = 0; k < 6; k++) { J0[k] *= f.sc; J1[k] *= f.sc; }
    if ((f.kind & 3) == 2) {
      double Je[6];
#pragma unroll
      for (int k = 0; k < 6; k++) Je[k] = f.g0 * J0[k] + f.g1 * J1[k];
      const double e_edge = f.g0 * r.e0 + f.g1 * r.e1;
      double w = huber_w(fabs(e_edge) / (double)s.scale_ls);
      if (f.kind & 4) w *= 0.5;
      if (CHI2) chi2 += e_edge * e_edge * w;   // the chi2 term (:488-526 / :602-641)
      int idx = 0;
#pragma unroll
      for (int a = 0; a < 6; a++) {
#pragma unroll
        for (int c = a; c < 6; c++) { acc[idx] += (Je[a] * Je[c]) * w; idx++; }
        acc[21 + a] -= (Je[a] * e_edge) * w;
      }
    } else {
      const double error_pt = sqrt(r.e0 * r.e0 + r.e1 * r.e1);
      double w = huber_w(error_pt / (double)s.scale_pt);
      if (f.kind & 4) w *= 0.5;
      if (CHI2) chi2 += error_pt * error_pt * w;
      int idx = 0;
#pragma unroll
      for (int a = 0; a < 6; a++) {
#pragma unroll
        for (int c = a; c < 6; c++) { acc[idx] += (J0[a] * J0[c] + J1[a] * J1[c]) * w; idx++; }
        acc[21 + a] -= (J0[a] * r.e0 + J1[a] * r.e1) * w;
      }
    }
  }
}

// The serial part of an LM trial, run by the first wavefront between two barriers (see k_pose).  Out of line on purpose: inlined,
// its temporaries (the 6x6 elimination, SE3::exp, the host table) raise the kernel's register demand at the point where every
// thread also holds its features, and the whole kernel spills (100 VGPRs instead of ~30).
__attribute__((noinline)) HSO_DEV void pose_lm_serial(PoseShared& s, int n_iter, int n_poses, bool first)
{
  const int tid = threadIdx.x;
  struct { int n_iter, n_poses; } J{n_iter, n_poses};
  const int lane = tid;
  if (lane == 0 && !first) {
    // ---- the trial just evaluated (:644-674)
    const bool nan_step = s.nan_step != 0;
    const double new_chi2 = nan_step ? 0.0 : s.red_n[27];
    s.rho = nan_step ? -1.0 : (s.chi2 - new_chi2);
    if (s.rho > 0) {
      s.T = s.Tn;
      s.chi2 = new_chi2;
      for (int q = 0; q < 27; q++) s.red[q] = s.red_n[q];
      double nm = -1;
      for (int q = 0; q < 6; q++) { const double a = fabs(s.dT[q]); if (a > nm) nm = a; }
      s.stop = nm <= 0.0000000001;  // hso::EPS
      const double t = 2 * s.rho - 1;
      s.mu *= fmax(1. / 3., fmin(1. - t * t * t, 2. / 3.));
      s.nu = 2.;
    } else {
      s.mu *= s.nu;
      s.nu *= 2.;
      if (s.mu < 0.0001) s.mu = 0.0001;
      ++s.n_trials;
      if (s.n_trials >= 5) s.stop = 1;
    }
    if (s.rho > 0 || s.stop) {                                 // the iteration is over
      if (s.stop || s.iter_cur + 1 >= J.n_iter) s.done = 1;
      else { s.iter_cur++; s.rho = 0; s.n_trials = 0; s.iters = s.iter_cur + 1; }
    }
  }
  // lane 0's words, read back by the whole wavefront (same wavefront: LDS operations complete in program order)
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  const bool done = *(volatile int*)&s.done != 0;
  if (!done) {
    // ---- the next trial: A = sums, A += (A.diagonal() * mu).asDiagonal() (:594), A.ldlt().solve(b) (:595)
    const int j = lane < 6 ? lane : 6;
    const double mu = *(volatile double*)&s.mu;
    double a[6];
#pragma unroll
    for (int i = 0; i < 6; i++) {
      if (j < 6) {
        const int lo = i < j ? i : j, hi = i < j ? j : i;
        double v = *(volatile double*)&s.red[lo * 6 - (lo * (lo - 1)) / 2 + (hi - lo)];
        if (i == j) v += v * mu;
        a[i] = v;
        s.A[i * 6 + j] = v;                                    // the last damped A is what the covariance inverts (:692)
      } else a[i] = *(volatile double*)&s.red[21 + i];
    }
    if (lane == 6) {
#pragma unroll
      for (int i = 0; i < 6; i++) s.b[i] = a[i];
    }
    pose_ldlt6(s, a);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    double dT[6];
#pragma unroll
    for (int i = 0; i < 6; i++) dT[i] = *(volatile double*)&s.dT[i];
    const bool nan_step = isnan(dT[0]);
    if (lane == 0) { s.n_trials_total++; s.nan_step = nan_step ? 1 : 0; }
    if (!nan_step) {
      // every lane forms the same trial pose (no broadcast needed), lane 0 keeps it; the lanes then fill the host table
      const Se3 Tn = se3_mul(se3_exp(dT), s.T);
      if (lane == 0) s.Tn = Tn;
      for (int h = lane; h < J.n_poses; h += 64) s.Tth[h] = se3_mul(Tn, s.hinv[h]);
    }
  }
}

// One 512-thread workgroup per frame (two wavefronts per SIMD); a thread owns up to FPT features (slot i = tid + q * 512 keeps
// the feature order) and holds what it needs of them in registers for the whole optimisation, so a pass touches no global
// memory: the previous form re-read the 96-byte feature records from L2 in every one of the ~60 passes, one dependent load
// chain per feature with a single wavefront per SIMD to hide it (2000 features: 5.0 ms per 256 frames, profiles/r3_*).
template <int FPT>
__global__ __launch_bounds__(POSE_THREADS, 2) void k_pose(hso_camera cam, const PoseJobDev* jobs, hso_pose_result* results)
{
  __shared__ PoseShared s;
  const PoseJobDev& J = jobs[blockIdx.x];
  hso_pose_result& out = results[blockIdx.x];
  const int tid = threadIdx.x, n = J.n_feats;
  const double em2 = (cam.fx * cam.fy < 0) ? fabs(cam.fx) : fabs((cam.fx + cam.fy) * 0.5);  // camera.cpp:59

  if (tid == 0) {
    s.T = se3_from(J.T);
    s.mu = 0.1; s.nu = 2.0; s.rho = 0; s.stop = 0; s.iters = 0; s.n_trials_total = 0; s.n_deleted = 0;
    for (int q = 0; q < 36; q++) s.A[q] = 0;
    for (int q = 0; q < 6; q++) s.b[q] = 0;
  }
  if (tid < J.n_poses) s.hinv[tid] = se3_inverse(se3_from(J.poses[tid]));
  PoseFeatReg pf[FPT];
#pragma unroll
  for (int q = 0; q < FPT; q++) {
    const int i = tid + q * POSE_THREADS;
    if (i < n) { pf[q] = pose_load_feat(J.feats[i]); if (J.mask) J.mask[i] = 0; }
    else { pf[q].kind = 0; pf[q].host_pose = 0; pf[q].X0 = pf[q].X1 = pf[q].X2 = pf[q].u = pf[q].v = pf[q].g0 = pf[q].g1 = 0; pf[q].sc = 1; }
  }
  __syncthreads();
  pose_set_Tth(s, s.T, J.n_poses);

  // ---- pass 0: initial errors (:426-454) and the MAD scales' inputs (:459-483) from ONE evaluation of the residuals.  The squared
  // errors are products of floats (`float error_pt` / `float error_ls`, :441-450, pushed into a vector<double>): non-negative fp32
  // bit patterns order like the doubles they convert to, so the median is a 32-bit select and converts exactly; the scales'
  // keys are the float magnitudes themselves (robust_cost.cpp:67-74).
  int c_pt = 0, c_ls = 0;
  unsigned k0[FPT], k1[FPT];
  int kinds[FPT];
#pragma unroll
  for (int q = 0; q < FPT; q++) {
    k0[q] = k1[q] = 0xFFFFFFFFu;
    kinds[q] = pf[q].kind;
    if (pf[q].kind & 3) {
      const Resid r = pose_residual(s, pf[q]);
      if ((pf[q].kind & 3) == 2) {
        const float error_ls = (float)(pf[q].g0 * r.e0 + pf[q].g1 * r.e1);
        k0[q] = __float_as_uint(error_ls * error_ls);
        k1[q] = __float_as_uint(fabsf(error_ls));
        c_ls++;
      } else {
        const float error_pt = (float)sqrt(r.e0 * r.e0 + r.e1 * r.e1);
        k0[q] = __float_as_uint(error_pt * error_pt);
        k1[q] = __float_as_uint(error_pt);
        c_pt++;
      }
    }
  }
  const int n_pt = pose_block_count(s, c_pt), n_ls = pose_block_count(s, c_ls);
  if (n_pt == 0 && n_ls == 0) {  // :456
    if (tid == 0) {
      out.T_f_w = J.T; for (int q = 0; q < 36; q++) out.cov[q] = 0;
      out.estimated_scale = 0; out.error_init = 0; out.error_final = 0; out.error_in_px = 1.f;
      out.num_obs = 0; out.n_deleted = 0; out.iters = 0; out.n_trials_total = 0; out.status = 1;
    }
    return;
  }
  const int n_init = n_pt + n_ls;
  unsigned sel[3];
  pose_select3<FPT>(s, k0, k1, kinds, (unsigned)(n_init / 2), (unsigned)(n_pt / 2), (unsigned)(n_ls / 2), sel);
  const double med_init = (double)__uint_as_float(sel[0]);
  // ---- MAD scales: 1.4826f * nth_element(|error|) per residual kind
  float scale_pt = n_pt > 0 ? 1.4826f * __uint_as_float(sel[1]) : 0.f, scale_ls = n_ls > 0 ? 1.4826f * __uint_as_float(sel[2]) : 0.f;
  if (n_pt > 0 && n_ls == 0) scale_ls = (float)(0.5 * (double)scale_pt);
  if (n_pt == 0 && n_ls > 0) scale_pt = (float)(2 * scale_ls);
  __syncthreads();
  if (tid == 0) { s.scale_pt = scale_pt; s.scale_ls = scale_ls; }
  __syncthreads();
  const double estimated_scale = (double)scale_pt;

  // ---- LM (:531-689)
  {  // chi2 (:488-526) and normal equations at the initial pose in one pass; every later set comes out of a trial's own pass
    double acc[32], c;
    pose_normal_pass<FPT, true>(s, pf, acc, c);
    pose_block_sum27_1(s, acc, c, s.red);
    if (tid == 0) s.chi2 = s.red[27];
    __syncthreads();
  }
  // The loop of :531-689 — per iteration up to five trials: damp, solve, step, evaluate, accept or raise the damping — as ONE loop
  // whose serial part runs on the first wavefront between two barriers: the decision about the trial just evaluated, and, unless
  // the optimisation is over, the next trial's damped system (lane j builds column j of A from the sums in place), its 6x6 LDL^T,
  // SE3::exp, the trial pose and the table T * host^-1 of every host frame.  Every value is formed by the operations of the
  // reference's statements in their order (the same bits as the one-lane form: tid 0 build / wave solve / tid 0 step / table, each
  // behind a barrier of its own: 8 barriers per trial, now 3).
  if (tid < 64) {
    if (tid == 0) { s.iter_cur = 0; s.done = J.n_iter <= 0; s.rho = 0; s.n_trials = 0; if (J.n_iter > 0) s.iters = 1; s.nan_step = 0; }
  }
  bool first = true;
  for (;;) {
    if (tid < 64) pose_lm_serial(s, J.n_iter, J.n_poses, first);
    first = false;
    __syncthreads();
    if (s.done) break;
    if (!s.nan_step) {
      double acc[32], c;
      pose_normal_pass<FPT, true>(s, pf, acc, c);
      pose_block_sum27_1(s, acc, c, s.red_n);                      // ends with a barrier
    } else __syncthreads();
  }

  // ---- covariance, culling, statistics (:691-767)
  pose_set_Tth(s, s.T, J.n_poses);
  const float thr_pt = (n < 80) ? (float)(sqrt(5.991) / em2) : (float)(J.reproj_thresh / em2);
  const float thr_ls = (float)(1.3 / em2);
  int n_del = 0;
  unsigned long long k64[FPT];
  bool k64_valid[FPT];
#pragma unroll
  for (int q = 0; q < FPT; q++) {
    const int i = tid + q * POSE_THREADS;
    k64[q] = ~0ull; k64_valid[q] = false;
    if (i >= n) continue;
    if (pf[q].kind & 3) {
      const Resid r = pose_residual(s, pf[q]);
      k64_valid[q] = true;
      if ((pf[q].kind & 3) == 2) {
        const double error_ls = pf[q].g0 * r.e0 + pf[q].g1 * r.e1;
        if (fabs(error_ls) > (double)thr_ls) { n_del++; if (J.mask) J.mask[i] = 1; }
        k64[q] = (unsigned long long)__double_as_longlong(error_ls * error_ls);
      } else {
        const float error_pt = (float)sqrt(r.e0 * r.e0 + r.e1 * r.e1);
        if (error_pt > thr_pt) { n_del++; if (J.mask) J.mask[i] = 1; }
        k64[q] = (unsigned long long)__double_as_longlong((double)(error_pt * error_pt));
      }
    }
  }
  n_del = pose_block_count(s, n_del);
  // the final median (:750-760): edgelet errors are squared in double there, so the keys are 64-bit; non-negative doubles order like
  // their bit patterns
  const double med_final = __longlong_as_double((long long)pose_select64<FPT>(s, k64, k64_valid, (unsigned)(n_init / 2)));
  if (tid < 64) {
    // Cov_ = (A * em2^2)^-1 (:692): Gauss-Jordan with partial pivoting on the last damped A — lane r holds row r of
    // [A | I] in registers (12 doubles, static indices: the one-lane form indexed a 6 x 12 array dynamically and lived in
    // scratch memory), rows travel by v_readlane
    const int lane = tid;
    const int r = lane < 6 ? lane : 5;
    double m[12];
    const double s2 = em2 * em2;
#pragma unroll
    for (int j = 0; j < 6; j++) { m[j] = s.A[r * 6 + j] * s2; m[6 + j] = (r == j) ? 1.0 : 0.0; }
#pragma unroll
    for (int c = 0; c < 6; c++) {
      int piv = c;
      double best = fabs(p_readlane_d(m[c], c));
#pragma unroll
      for (int q = c + 1; q < 6; q++) { const double v = fabs(p_readlane_d(m[c], q)); if (v > best) { best = v; piv = q; } }
#pragma unroll
      for (int q = c + 1; q < 6; q++) {
        if (piv == q) {   // swap rows c and q
#pragma unroll
          for (int j = 0; j < 12; j++) {
            const double from_q = p_readlane_d(m[j], q), from_c = p_readlane_d(m[j], c);
            m[j] = (lane == c) ? from_q : ((lane == q) ? from_c : m[j]);
          }
        }
      }
      const double d = p_readlane_d(m[c], c);
      double rowc[12];
#pragma unroll
      for (int j = 0; j < 12; j++) rowc[j] = p_readlane_d(m[j], c) / d;
      const double f = m[c];
#pragma unroll
      for (int j = 0; j < 12; j++) m[j] = (lane == c) ? rowc[j] : m[j] - f * rowc[j];
    }
    if (lane < 6) {
#pragma unroll
      for (int j = 0; j < 6; j++) out.cov[lane * 6 + j] = m[6 + j];
    }
  }
  if (tid == 0) {
    se3_to(s.T, out.T_f_w);
    out.error_init = sqrt(med_init) * em2;
    out.error_final = sqrt(med_final) * em2;
    out.estimated_scale = estimated_scale * em2;
    out.num_obs = n_init - n_del;
    out.n_deleted = n_del;
    out.error_in_px = out.error_final < 1.5 ? 1.0f : (float)(1.5 / out.error_final);
    out.iters = s.iters;
    out.n_trials_total = s.n_trials_total;
    out.status = 0;
  }
}

static void pose_launch(hipStream_t stream, const hso_camera* cam, const PoseJobDev* dj, int n_jobs, int n_max, hso_pose_result* dr)
{
  // features per thread: the smallest power of two that covers the largest table of the batch
  if (n_max <= POSE_THREADS) hipLaunchKernelGGL(k_pose<1>, dim3(n_jobs), dim3(POSE_THREADS), 0, stream, *cam, dj, dr);
  else if (n_max <= 2 * POSE_THREADS) hipLaunchKernelGGL(k_pose<2>, dim3(n_jobs), dim3(POSE_THREADS), 0, stream, *cam, dj, dr);
  else if (n_max <= 4 * POSE_THREADS) hipLaunchKernelGGL(k_pose<4>, dim3(n_jobs), dim3(POSE_THREADS), 0, stream, *cam, dj, dr);
  else hipLaunchKernelGGL(k_pose<8>, dim3(n_jobs), dim3(POSE_THREADS), 0, stream, *cam, dj, dr);
}

int hso_pose_launch_device(hso_gpu_ctx* ctx, const hso_camera* cam, const PoseJobDev* d_jobs, int n_jobs, int n_max_feats,
                           hso_pose_result* d_results)
{
  if (n_jobs <= 0) return HSO_OK;
  if (n_max_feats > POSE_MAX_FEATS) return hso_fail(ctx, HSO_E_INVALID, "pose_optimize: n_feats out of range (max 4096)");
  pose_launch(ctx->stream, cam, d_jobs, n_jobs, n_max_feats, d_results);
  HSO_HIP_CHECK(ctx, hipGetLastError());
  return HSO_OK;
}

extern "C" int hso_gpu_pose_optimize_batch(hso_gpu_ctx* ctx, const hso_camera* cam, const hso_pose_job* jobs, int n_jobs,
                                           hso_pose_result* results, uint8_t* const* outlier_mask)
{
  if (!ctx) return HSO_E_INVALID;
  if (!cam || n_jobs < 0 || (n_jobs > 0 && (!jobs || !results))) return hso_fail(ctx, HSO_E_INVALID, "pose_optimize: bad argument");
  if (n_jobs == 0) return HSO_OK;
  HSO_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  size_t tot_feats = 0, tot_poses = 0;
  for (int j = 0; j < n_jobs; j++) {
    if (jobs[j].n_feats < 0 || jobs[j].n_feats > POSE_MAX_FEATS) return hso_fail(ctx, HSO_E_INVALID, "pose_optimize: n_feats out of range (max 4096)");
    if (jobs[j].n_poses <= 0 || jobs[j].n_poses > POSE_MAX_POSES || !jobs[j].poses_f_w) return hso_fail(ctx, HSO_E_INVALID, "pose_optimize: n_poses out of range (1..128)");
    if (jobs[j].n_feats > 0 && !jobs[j].feats) return hso_fail(ctx, HSO_E_INVALID, "pose_optimize: null feature table");
    for (int i = 0; i < jobs[j].n_feats; i++) {
      const hso_pose_feat& f = jobs[j].feats[i];
      if (f.has_point && (f.host_pose < 0 || f.host_pose >= jobs[j].n_poses || f.level < 0 || f.level > 30))
        return hso_fail(ctx, HSO_E_INVALID, "pose_optimize: host_pose or level out of range");
    }
    tot_feats += jobs[j].n_feats; tot_poses += jobs[j].n_poses;
  }
  auto al = [](size_t b) { return (b + 255) & ~size_t(255); };
  const size_t o_feats = al(sizeof(PoseJobDev) * n_jobs);
  const size_t o_poses = o_feats + al(sizeof(hso_pose_feat) * tot_feats);
  const size_t o_mask = o_poses + al(sizeof(hso_se3) * tot_poses);
  const size_t o_res = o_mask + al(tot_feats);
  const size_t need = o_res + sizeof(hso_pose_result) * n_jobs;
  // device work area and pinned host staging are the context's grow-only buffers: no allocation per call
  if (ctx->batch_cap < need) {
    HSO_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->d_batch) (void)hipFree(ctx->d_batch);
    ctx->d_batch = nullptr; ctx->batch_cap = 0;
    HSO_HIP_CHECK(ctx, hipMalloc(reinterpret_cast<void**>(&ctx->d_batch), hso_grown(need)));
    ctx->batch_cap = hso_grown(need);
  }
  char* d = ctx->d_batch;
  char* h = hso_pinned(ctx, 0, o_res);
  char* hm = hso_pinned(ctx, 1, tot_feats + sizeof(hso_pose_result) * n_jobs + 64);
  if (!h || !hm) return HSO_E_NOMEM;
  memset(h, 0, o_res);
  PoseJobDev* hj = reinterpret_cast<PoseJobDev*>(h);
  size_t fo = 0, po = 0;
  for (int j = 0; j < n_jobs; j++) {
    memcpy(h + o_feats + sizeof(hso_pose_feat) * fo, jobs[j].feats, sizeof(hso_pose_feat) * jobs[j].n_feats);
    memcpy(h + o_poses + sizeof(hso_se3) * po, jobs[j].poses_f_w, sizeof(hso_se3) * jobs[j].n_poses);
    hj[j].feats = reinterpret_cast<const hso_pose_feat*>(d + o_feats) + fo;
    hj[j].poses = reinterpret_cast<const hso_se3*>(d + o_poses) + po;
    hj[j].mask = reinterpret_cast<uint8_t*>(d + o_mask) + fo;
    hj[j].n_feats = jobs[j].n_feats; hj[j].n_poses = jobs[j].n_poses;
    hj[j].T = jobs[j].T_f_w; hj[j].reproj_thresh = jobs[j].reproj_thresh; hj[j].n_iter = jobs[j].n_iter; hj[j]._pad = 0;
    fo += jobs[j].n_feats; po += jobs[j].n_poses;
  }
  hipError_t e = hipMemcpyAsync(d, h, o_res, hipMemcpyHostToDevice, ctx->stream);
  if (e == hipSuccess) {
    int n_max = 0;
    for (int j = 0; j < n_jobs; j++) n_max = std::max(n_max, (int)jobs[j].n_feats);
    pose_launch(ctx->stream, cam, reinterpret_cast<const PoseJobDev*>(d), n_jobs, n_max, reinterpret_cast<hso_pose_result*>(d + o_res));
    e = hipGetLastError();
  }
  char* hres = hm + ((tot_feats + 63) & ~size_t(63));
  if (e == hipSuccess) e = hipMemcpyAsync(hres, d + o_res, sizeof(hso_pose_result) * n_jobs, hipMemcpyDeviceToHost, ctx->stream);
  if (e == hipSuccess && outlier_mask && tot_feats > 0) e = hipMemcpyAsync(hm, d + o_mask, tot_feats, hipMemcpyDeviceToHost, ctx->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  if (e != hipSuccess) { ctx->err = std::string("pose_optimize: ") + hipGetErrorString(e); return HSO_E_HIP; }
  memcpy(results, hres, sizeof(hso_pose_result) * n_jobs);
  if (outlier_mask) {
    fo = 0;
    for (int j = 0; j < n_jobs; j++) {
      if (outlier_mask[j] && jobs[j].n_feats > 0) memcpy(outlier_mask[j], hm + fo, jobs[j].n_feats);
      fo += jobs[j].n_feats;
    }
  }
  return HSO_OK;
}
