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
  int cnt[POSE_WAVES];
  unsigned hist[256];                       // radix select: digit histogram
  unsigned sel_bin, sel_rank;
  unsigned long long keys64[POSE_MAX_FEATS];   // also the 32-bit keys of the MAD scales (never live together)
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

// k-th smallest (0-based) of n keys held in LDS — exactly the element nth_element would leave at position k
// (math_utils.h:119-126, robust_cost.cpp:70-71) — by an MSB-first radix select, 8 bits per pass: histogram of the digit over
// the keys that match the prefix found so far (LDS atomics), the bin that holds the rank located by the first 256 threads.
// Empty slots hold all-ones keys (larger than any valid key; k is always below the number of valid keys).
template <typename KeyT, int BITS>
HSO_DEV KeyT pose_select(PoseShared& s, const KeyT* keys, int n, int k)
{
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  KeyT prefix = 0;
  unsigned rank = (unsigned)k;
  for (int shift = BITS - 8; shift >= 0; shift -= 8) {
    __syncthreads();
    if (tid < 256) s.hist[tid] = 0;
    __syncthreads();
    const KeyT hi_mask = (shift + 8 >= BITS) ? (KeyT)0 : (KeyT)(~(KeyT)0 << (shift + 8));
    for (int i = tid; i < n; i += POSE_THREADS) {
      const KeyT key = keys[i];
      if ((key & hi_mask) == prefix) atomicAdd(&s.hist[(unsigned)(key >> shift) & 255u], 1u);
    }
    __syncthreads();
    // inclusive scan of the 256 bins: within each of the first four wavefronts, then across them
    unsigned c = 0, incl = 0;
    if (tid < 256) {
      c = s.hist[tid];
      incl = c;
      for (int d = 1; d < 64; d <<= 1) { const unsigned o = __shfl_up(incl, d); if (lane >= d) incl += o; }
      if (lane == 63) s.cnt[wave] = (int)incl;
    }
    __syncthreads();
    if (tid < 256) {
      for (int w = 0; w < wave; w++) incl += (unsigned)s.cnt[w];
      if (rank >= incl - c && rank < incl) { s.sel_bin = (unsigned)tid; s.sel_rank = rank - (incl - c); }
    }
    __syncthreads();
    prefix |= (KeyT)s.sel_bin << shift;
    rank = s.sel_rank;
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
HSO_DEV void pose_ldlt6(PoseShared& s)
{
  const int lane = threadIdx.x & 63;
  const int j = lane < 7 ? lane : 6;  // lanes 0..5: columns, lane 6: rhs
  double a[6];
#pragma unroll
  for (int i = 0; i < 6; i++) a[i] = (j < 6) ? s.A[i * 6 + j] : s.b[i];
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
  unsigned* const keys32 = reinterpret_cast<unsigned*>(s.keys64);

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

  // ---- pass 0: initial errors (:426-454).  Slot i keeps the feature order; empty slots hold
  // all-ones keys (larger than any valid key) so that order statistics ignore them.  The squared errors are products of
  // floats (`float error_pt` / `float error_ls`, :441-450, pushed into a vector<double>): non-negative fp32 bit patterns order
  // like the doubles they convert to, so the median is a 32-bit select (4 radix passes instead of 8) and converts exactly.
  int c_pt = 0, c_ls = 0;
#pragma unroll
  for (int q = 0; q < FPT; q++) {
    const int i = tid + q * POSE_THREADS;
    if (i >= n) continue;
    unsigned k32 = 0xFFFFFFFFu;
    if (pf[q].kind & 3) {
      const Resid r = pose_residual(s, pf[q]);
      if ((pf[q].kind & 3) == 2) {
        const float error_ls = (float)(pf[q].g0 * r.e0 + pf[q].g1 * r.e1);
        k32 = __float_as_uint(error_ls * error_ls);
        c_ls++;
      } else {
        const float error_pt = (float)sqrt(r.e0 * r.e0 + r.e1 * r.e1);
        k32 = __float_as_uint(error_pt * error_pt);
        c_pt++;
      }
    }
    keys32[i] = k32;
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
  const double med_init = (double)__uint_as_float(pose_select<unsigned, 32>(s, keys32, n, n_init / 2));

  // ---- MAD scales (:459-483): 1.4826f * nth_element(|error|) per residual kind
  float scale_pt = 0, scale_ls = 0;
  for (int kind = 0; kind < 2; kind++) {
    const int cnt = kind == 0 ? n_pt : n_ls;
    __syncthreads();
#pragma unroll
    for (int q = 0; q < FPT; q++) {
      const int i = tid + q * POSE_THREADS;
      if (i >= n) continue;
      unsigned key = 0xFFFFFFFFu;
      if ((pf[q].kind & 3) == (kind == 1 ? 2 : 1)) {
        const Resid r = pose_residual(s, pf[q]);
        const float e = (kind == 1) ? fabsf((float)(pf[q].g0 * r.e0 + pf[q].g1 * r.e1)) : (float)sqrt(r.e0 * r.e0 + r.e1 * r.e1);
        key = __float_as_uint(e);
      }
      keys32[i] = key;
    }
    __syncthreads();
    if (cnt > 0) {
      const float med = __uint_as_float(pose_select<unsigned, 32>(s, keys32, n, cnt / 2));
      if (kind == 0) scale_pt = 1.4826f * med; else scale_ls = 1.4826f * med;
    }
  }
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
  for (int iter = 0; iter < J.n_iter; iter++) {
    if (tid == 0) { s.rho = 0; s.n_trials = 0; s.iters = iter + 1; }
    __syncthreads();
    for (;;) {
      if (tid == 0) {
        int idx = 0;
        for (int a = 0; a < 6; a++)
          for (int c = a; c < 6; c++) { s.A[a * 6 + c] = s.A[c * 6 + a] = s.red[idx]; idx++; }
        for (int a = 0; a < 6; a++) s.b[a] = s.red[21 + a];
        for (int a = 0; a < 6; a++) s.A[a * 6 + a] += s.A[a * 6 + a] * s.mu;  // A += (A.diagonal()*mu).asDiagonal()
        s.n_trials_total++;
      }
      __syncthreads();
      if (tid < 64) pose_ldlt6(s);
      __syncthreads();
      const bool nan_step = isnan(s.dT[0]);
      if (!nan_step) {
        if (tid == 0) s.Tn = se3_mul(se3_exp(s.dT), s.T);
        __syncthreads();
        pose_set_Tth(s, s.Tn, J.n_poses);
        double acc[32], c;
        pose_normal_pass<FPT, true>(s, pf, acc, c);
        pose_block_sum27_1(s, acc, c, s.red_n);
      }
      if (tid == 0) {
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
      }
      __syncthreads();
      if (s.rho > 0 || s.stop) break;
    }
    if (s.stop) break;
  }

  // ---- covariance, culling, statistics (:691-767)
  pose_set_Tth(s, s.T, J.n_poses);
  const float thr_pt = (n < 80) ? (float)(sqrt(5.991) / em2) : (float)(J.reproj_thresh / em2);
  const float thr_ls = (float)(1.3 / em2);
  int n_del = 0;
#pragma unroll
  for (int q = 0; q < FPT; q++) {
    const int i = tid + q * POSE_THREADS;
    if (i >= n) continue;
    unsigned long long k64 = ~0ull;
    if (pf[q].kind & 3) {
      const Resid r = pose_residual(s, pf[q]);
      if ((pf[q].kind & 3) == 2) {
        const double error_ls = pf[q].g0 * r.e0 + pf[q].g1 * r.e1;
        if (fabs(error_ls) > (double)thr_ls) { n_del++; if (J.mask) J.mask[i] = 1; }
        k64 = (unsigned long long)__double_as_longlong(error_ls * error_ls);
      } else {
        const float error_pt = (float)sqrt(r.e0 * r.e0 + r.e1 * r.e1);
        if (error_pt > thr_pt) { n_del++; if (J.mask) J.mask[i] = 1; }
        k64 = (unsigned long long)__double_as_longlong((double)(error_pt * error_pt));
      }
    }
    s.keys64[i] = k64;
  }
  n_del = pose_block_count(s, n_del);
  const double med_final = __longlong_as_double((long long)pose_select<unsigned long long, 64>(s, s.keys64, n, n_init / 2));
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
