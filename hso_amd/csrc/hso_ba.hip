// hso_ba.hip — local bundle adjustment on gfx950: per-edge errors, analytic Jacobians and the
// robustified normal-equation blocks that the reference builds through g2o.
//
// Replaces, for one linearisation point, g2o::BlockSolver::buildSystem over the edges
// ba::LocalBundleAdjustment creates (reference src/bundle_adjustment.cpp:690-812):
// EdgeProjectID2UV::computeError/linearizeOplus (include/hso/bundle_adjustment.h:219-287),
// EdgeProjectID2UVEdgeLet (:317-384), BaseMultiEdge::constructQuadraticForm
// (thirdparty/g2o/g2o/core/base_multi_edge.hpp:36-48,171-222), RobustKernelHuber
// (robust_kernel_impl.cpp:78-91), robustInformation = rho' * Omega (base_edge.h:96-102).
//
// MI355X mapping: three small launches, no atomics, every output owned by exactly one thread
// or one workgroup, so the result is deterministic:
//   k_ba_edges   thread per edge: Tth, error, J_point (2x1), J_host / J_target (2x6), Huber
//                weight; the 29-double linearisation is written once (coalesced SoA).
//   k_ba_points  thread per point: walks its edges (CSR built on the host, edge order kept) ->
//                Hpp, bp and the 1x6 point-pose blocks.
//   k_ba_poses   workgroup per pose pair (i <= j): strided pass over the pair's own edges (CSR
//                built on the host), 36 (+6) partial sums per thread, fixed-tree reduction -> the
//                6x6 block (and b_i on the diagonal).
// The reference accumulates serially in edge order in fp64; the tree order differs by rounding
// (parity tolerance 1e-11 relative).  Sizes are small (<= ~15 poses, a few hundred points,
// 1-3 k edges): latency-bound; throughput comes from issuing many keyframes' problems on one stream.
// Around them (further down): the Levenberg loop as a device state machine (BaLmDev, ba_decide at the tail of k_ba_chi2, ba_run), the Huber deltas
// (k_ba_mad_*), and hso_gpu_seq_local_ba — the window assembled from a sequence map's resident tables (k_rba_*).
#include "hso_ctx.h"
#include "hso_dev_math.h"
#include <string.h>
#include <algorithm>
#include <cmath>
#include <functional>
#include <mutex>
#include <vector>

using namespace hso_dev;

#define BA_THREADS 256
#define BA_WAVES (BA_THREADS / 64)
#define BA_LIN 32  // doubles per edge in the linearisation record
#define BA_BACKSUB_BLOCKS 16   // workgroups per window of k_ba_backsub (<= 32: their partial sums lie in the window's 256-byte sum slot)

// record layout (SoA over edges, stride = n_edges): [0..1] err, [2..3] Jp, [4..15] Jh, [16..27] Jt,
// [28] omega (= rho' * information), [29] rho' * information applied to -err is folded as omega_r = -omega*err,
// [30] dim, [31] unused
struct BaArgs {
  const hso_se3* poses;
  const uint8_t* fixed;
  const double* idist;
  const hso_ba_edge* edges;
  int n_poses, n_points, n_edges;
  double huber_corner, huber_edge;
  double* lin;        // [BA_LIN][n_edges]
  double* edge_err;   // [n_edges][2]
  double* edge_chi2;  // [n_edges]
  double* edge_rho;   // [n_edges]
};

// One local-BA window on the device.  Every kernel below takes the table of windows and a list of the windows it works on
// (blockIdx.y indexes the list): the windows of many sequences share every launch.
struct BaProb {
  BaArgs a;
  const int* off; const int* list;      // CSR of edges by point (edge order kept)
  const int* poff; const int* plist;    // CSR of edges by pose-pair block
  double* Hpp; double* bp; double* Hpc; double* Hcc; double* bc;
  double* sum;                          // [0] chi2, [1] robust chi2, [2] sum over points of xp (lambda xp + bp)
  const int* col;                       // [n_poses] first row of the pose in the reduced system, -1 = fixed
  double* S; double* rhs;               // reduced (Schur) system [M * M], [M]; S[M * M] doubles as the "solvable" flag slot
  const double* trial;                  // [0] lambda, [1 ...] pose steps xc [6 * n_poses], then != 0: the solve failed, no step
  double* xp;                           // [n_points] point steps of the last back-substitution
  double* part;                         // [BA_BACKSUB_BLOCKS] the back-substitution's partial sums of the points' computeScale
  double* idist_rw;                     // a.idist, writable
  double* idist_bak;                    // the state before the last step (g2o's push())
  double* trial_rw;                     // = trial, writable: the solve kernel leaves the pose steps and the "no step" flag there
  const double* lam;                    // this window's damping of the current trial (one contiguous array for all windows)
  hso_se3* poses_rw;                    // = a.poses, writable
  hso_se3* poses_bak;
  int M, n_pairs;
  char* zero_begin; size_t zero_bytes;  // [Hpp ... chi2 sums]: what a linearisation accumulates into (16-byte units)
  // hso_gpu_ba_local_multi (the Huber deltas formed on the device before the optimisation): the observations' project2d(obs->f)
  // [2 * n_edges], per-edge error magnitudes (scratch: the edge_err table before its first use), the two deltas as floats
  const double* uv; float* mad; float* hub;
};

// the damping of a window's current trial; a negative value = "computeLambdaInit of the linearisation just made": tau (1e-5) x the
// largest diagonal entry (k_ba_maxdiag -> sum[5]) — the first trial of a window rides in the round of its first linearisation
__device__ inline double ba_lambda(const BaProb& P) { const double l = P.lam[0]; return l < 0 ? 1e-5 * P.sum[5] : l; }

// The Levenberg loop's state of one window, on the device (OptimizationAlgorithmLevenberg::solve as a state machine: see BaLm's
// comment below, which this is the device form of).  `want` says which device work the window takes part in during the NEXT round;
// every kernel of a round is launched for all windows (blockIdx.y = window) and leaves at once for a window that does not want it.
enum { BA_W_ERRORS = 1, BA_W_RESTORE = 2, BA_W_LINEARIZE = 4, BA_W_TRIAL = 8 };
enum { BA_ST_INIT_WAIT = 0, BA_ST_LIN_WAIT = 1, BA_ST_STEP_WAIT = 2, BA_ST_FINAL_WAIT = 3, BA_ST_DONE = 4 };
struct BaLmDev {
  double lambda, ni, currentChi, tempChi, iniChi, rho;
  hso_ba_result res;
  int32_t nBad, stop, it, qmax, n_iter, st, want, need_restore, first_lin, written, pad_[2];
};
// the window a workgroup works on; null: the window sits this kernel out (lm == null: a call outside the loop, every window takes part)
__device__ inline const BaProb* ba_window(const BaProb* probs, const BaLmDev* lm, int mask)
{
  const int w = blockIdx.y;
  if (lm && !(lm[w].want & mask)) return nullptr;
  return probs + w;
}

// ---- g2o's SE3Quat update (host and device: the optimiser applies it on the device, the tests' helpers on the host)
struct Q4 { double x, y, z, w; };

HSO_HD Q4 qmul(const Q4& a, const Q4& b)
{
  return { a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y, a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z,
           a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x, a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z };
}

HSO_HD void qrot(const Q4& q, const double v[3], double o[3])   // Eigen QuaternionBase::_transformVector
{
  double uv[3] = { q.y * v[2] - q.z * v[1], q.z * v[0] - q.x * v[2], q.x * v[1] - q.y * v[0] };
  uv[0] += uv[0]; uv[1] += uv[1]; uv[2] += uv[2];
  o[0] = (v[0] + q.w * uv[0]) + (q.y * uv[2] - q.z * uv[1]);
  o[1] = (v[1] + q.w * uv[1]) + (q.z * uv[0] - q.x * uv[2]);
  o[2] = (v[2] + q.w * uv[2]) + (q.x * uv[1] - q.y * uv[0]);
}

HSO_HD void normalize_rotation(Q4& q)   // se3quat.h:280-285
{
  if (q.w < 0) { q.x = -q.x; q.y = -q.y; q.z = -q.z; q.w = -q.w; }
  const double n = sqrt(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w);
  q.x /= n; q.y /= n; q.z /= n; q.w /= n;
}

// SE3Quat::exp(update) * pose, update = [omega, upsilon] (se3quat.h:223-257, :104-110)
HSO_HD void se3quat_exp_times(const double* upd, hso_se3& pose)
{
  const double wx = upd[0], wy = upd[1], wz = upd[2];
  const double theta = sqrt(wx * wx + wy * wy + wz * wz);
  const double O[9] = { 0, -wz, wy, wz, 0, -wx, -wy, wx, 0 };
  double O2[9], R[9], V[9];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) O2[i * 3 + j] = (O[i * 3] * O[j] + O[i * 3 + 1] * O[3 + j]) + O[i * 3 + 2] * O[6 + j];
  if (theta < 0.00001) {
    for (int i = 0; i < 9; i++) { R[i] = ((i % 4 == 0 ? 1.0 : 0.0) + O[i]) + O2[i]; V[i] = R[i]; }
  } else {
    const double a = sin(theta) / theta, b = (1 - cos(theta)) / (theta * theta), c = (theta - sin(theta)) / pow(theta, 3.0);
    for (int i = 0; i < 9; i++) {
      const double id = (i % 4 == 0) ? 1.0 : 0.0;
      R[i] = (id + a * O[i]) + b * O2[i];
      V[i] = (id + b * O[i]) + c * O2[i];
    }
  }
  // Eigen::Quaterniond(Matrix3d)
  Q4 q;
  double t = R[0] + R[4] + R[8];
  if (t > 0) {
    t = sqrt(t + 1.0);
    q.w = 0.5 * t; t = 0.5 / t;
    q.x = (R[7] - R[5]) * t; q.y = (R[2] - R[6]) * t; q.z = (R[3] - R[1]) * t;
  } else {
    int i = 0;
    if (R[4] > R[0]) i = 1;
    if (R[8] > R[i * 4]) i = 2;
    const int j = (i + 1) % 3, k = (j + 1) % 3;
    double qv[3];
    t = sqrt(R[i * 4] - R[j * 4] - R[k * 4] + 1.0);
    qv[i] = 0.5 * t; t = 0.5 / t;
    q.w = (R[k * 3 + j] - R[j * 3 + k]) * t;
    qv[j] = (R[j * 3 + i] + R[i * 3 + j]) * t;
    qv[k] = (R[k * 3 + i] + R[i * 3 + k]) * t;
    q.x = qv[0]; q.y = qv[1]; q.z = qv[2];
  }
  normalize_rotation(q);
  const double tv[3] = { (V[0] * upd[3] + V[1] * upd[4]) + V[2] * upd[5], (V[3] * upd[3] + V[4] * upd[4]) + V[5] * upd[5],
                         (V[6] * upd[3] + V[7] * upd[4]) + V[8] * upd[5] };
  // result = exp * pose
  double rt[3];
  qrot(q, pose.t, rt);
  Q4 qp = { pose.q[0], pose.q[1], pose.q[2], pose.q[3] };
  Q4 qn = qmul(q, qp);
  normalize_rotation(qn);
  pose.q[0] = qn.x; pose.q[1] = qn.y; pose.q[2] = qn.z; pose.q[3] = qn.w;
  pose.t[0] = tv[0] + rt[0]; pose.t[1] = tv[1] + rt[1]; pose.t[2] = tv[2] + rt[2];
}


template <bool LIN>   // LIN = false: errors, chi2 and rho only (an LM trial's computeActiveErrors)
__global__ __launch_bounds__(BA_THREADS) void k_ba_edges(const BaProb* probs, const BaLmDev* lm, int mask)
{
  const BaProb* PP = ba_window(probs, lm, mask);
  if (!PP) return;
  const BaArgs a = PP->a;
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= a.n_edges) return;
  const hso_ba_edge e = a.edges[k];
  // SE3(Quaterniond, Vector3d) normalises (so3.cpp:43-47); Tth = Ttw * Thw^-1
  Se3 Ttw = se3_from(a.poses[e.target]), Thw = se3_from(a.poses[e.host]);
  quat_normalize(Ttw); quat_normalize(Thw);
  const Se3 Tth = se3_mul(Ttw, se3_inverse(Thw));
  const double idHost = a.idist[e.point];
  const double inv = 1.0 / idHost;
  double x, y, z;
  se3_apply(Tth, e.fH[0] * inv, e.fH[1] * inv, e.fH[2] * inv, x, y, z);
  const double proj0 = x / z, proj1 = y / z;
  double R[9];
  so3_matrix(Tth, R);
  const double t0 = Tth.tx, t1 = Tth.ty, t2 = Tth.tz;
  const double Rf2 = R[6] * e.fH[0] + R[7] * e.fH[1] + R[8] * e.fH[2];
  const double Juvdd0 = -(t0 - proj0 * t2) / (Rf2 + idHost * t2);
  const double Juvdd1 = -(t1 - proj1 * t2) / (Rf2 + idHost * t2);
  const double z_2 = z * z;
  double Jp6[12];
  Jp6[0] = x * y / z_2; Jp6[1] = -(1 + (x * x / z_2)); Jp6[2] = y / z; Jp6[3] = -1. / z; Jp6[4] = 0; Jp6[5] = x / z_2;
  Jp6[6] = (1 + y * y / z_2); Jp6[7] = -x * y / z_2; Jp6[8] = -x / z; Jp6[9] = 0; Jp6[10] = -1. / z; Jp6[11] = y / z_2;
  // -Tth.Adj() = -[[R, hat(t) R], [0, R]]  (Sophus block layout, se3.cpp:108-118)
  const double hat[9] = { 0, -t2, t1, t2, 0, -t0, -t1, t0, 0 };
  double nAdj[36];
#pragma unroll
  for (int i = 0; i < 36; i++) nAdj[i] = 0;
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) {
      nAdj[i * 6 + j] = -R[i * 3 + j];
      nAdj[(3 + i) * 6 + 3 + j] = -R[i * 3 + j];
      double s = hat[i * 3 + 0] * R[0 * 3 + j];
      s += hat[i * 3 + 1] * R[1 * 3 + j];
      s += hat[i * 3 + 2] * R[2 * 3 + j];
      nAdj[i * 6 + 3 + j] = -s;
    }
  double rec[BA_LIN];
#pragma unroll
  for (int i = 0; i < BA_LIN; i++) rec[i] = 0;
  int dim;
  if (e.type == HSO_FTR_EDGELET) {
    dim = 1;
    const double n0 = e.normal[0], n1 = e.normal[1];
    rec[0] = e.meas[0] - (n0 * proj0 + n1 * proj1);
    rec[2] = n0 * Juvdd0 + n1 * Juvdd1;
    double nJ[6];
#pragma unroll
    for (int q = 0; q < 6; q++) nJ[q] = n0 * Jp6[q] + n1 * Jp6[6 + q];
#pragma unroll
    for (int c = 0; c < 6; c++) {
      double s = 0;
#pragma unroll
      for (int q = 0; q < 6; q++) s += nJ[q] * nAdj[q * 6 + c];
      rec[4 + c] = s;
      rec[16 + c] = nJ[c];
    }
  } else {
    dim = 2;
    rec[0] = e.meas[0] - proj0; rec[1] = e.meas[1] - proj1;
    rec[2] = Juvdd0; rec[3] = Juvdd1;
#pragma unroll
    for (int r = 0; r < 2; r++)
#pragma unroll
      for (int c = 0; c < 6; c++) {
        double s = 0;
#pragma unroll
        for (int q = 0; q < 6; q++) s += Jp6[r * 6 + q] * nAdj[q * 6 + c];
        rec[4 + r * 6 + c] = s;
        rec[16 + r * 6 + c] = Jp6[r * 6 + c];
      }
  }
  const float inv_sigma2 = (float)(1.0 / (double)((1 << e.level) * (1 << e.level)));
  const double om = (double)inv_sigma2;
  double chi2 = 0;
  chi2 += rec[0] * om * rec[0];                 // static indices: a loop to the run-time `dim` put the whole record in scratch memory
  if (dim > 1) chi2 += rec[1] * om * rec[1];
  const double delta = (e.type == HSO_FTR_EDGELET) ? a.huber_edge : a.huber_corner;
  const double dsqr = delta * delta;
  double rho0, rho1;
  if (chi2 <= dsqr) { rho0 = chi2; rho1 = 1.; }
  else { const double sqrte = sqrt(chi2); rho0 = 2 * sqrte * delta - dsqr; rho1 = delta / sqrte; }
  rec[28] = rho1 * om;   // robustInformation
  rec[29] = om * rho1;   // factor of omega_r = -(om * err) * rho1 (kept separate to mirror the expression order)
  rec[30] = (double)dim;
  if (LIN) {
#pragma unroll
    for (int i = 0; i < BA_LIN; i++) a.lin[(size_t)i * a.n_edges + k] = rec[i];
  }
  a.edge_err[2 * k] = rec[0]; a.edge_err[2 * k + 1] = rec[1];
  a.edge_chi2[k] = chi2;
  a.edge_rho[k] = rho0;
}

struct EdgeLin {
  double err[2], Jp[2], Jh[12], Jt[12], omega, om, rho1;
  int dim;
};

HSO_DEV EdgeLin ba_load(const BaArgs& a, int k)
{
  EdgeLin L;
  const size_t n = a.n_edges;
  L.err[0] = a.lin[0 * n + k]; L.err[1] = a.lin[1 * n + k];
  L.Jp[0] = a.lin[2 * n + k]; L.Jp[1] = a.lin[3 * n + k];
#pragma unroll
  for (int i = 0; i < 12; i++) { L.Jh[i] = a.lin[(4 + i) * n + k]; L.Jt[i] = a.lin[(16 + i) * n + k]; }
  L.omega = a.lin[28 * n + k];
  L.dim = (int)a.lin[30 * n + k];
  return L;
}

// omega_r[d] = -(om * err[d]) * rho1 with om*rho1 = omega up to one rounding; the reference
// computes (-(om*err))*rho1 (base_multi_edge.hpp:43-44); reproduce from the stored factors
HSO_DEV double ba_omega_r(const BaArgs& a, int k, const EdgeLin& L, int d)
{
  const hso_ba_edge& e = a.edges[k];
  const double om = (double)(float)(1.0 / (double)((1 << e.level) * (1 << e.level)));
  const double rho1 = L.omega / om;  // exact: om is a power of two
  return -(om * L.err[d]) * rho1;
}

__global__ __launch_bounds__(BA_THREADS) void k_ba_points(const BaProb* probs, const BaLmDev* lm, int mask)
{
  const BaProb* PP = ba_window(probs, lm, mask);
  if (!PP) return;
  const BaProb& P = *PP;
  const BaArgs a = P.a;
  const int* pt_off = P.off; const int* pt_edges = P.list;
  double* Hpp = P.Hpp; double* bp = P.bp; double* Hpc = P.Hpc;
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= a.n_points) return;
  double hpp = 0, b = 0;
  for (int q = pt_off[p]; q < pt_off[p + 1]; q++) {
    const int k = pt_edges[q];
    const EdgeLin L = ba_load(a, k);
    const hso_ba_edge& e = a.edges[k];
    // static indices (the residual's second row behind `two`): the edge record stays in registers (it lived in scratch memory
    // with run-time loop bounds); operations and their order are unchanged
    const bool two = L.dim > 1;
    double s = 0, g = 0;
    const double AtO0 = L.Jp[0] * L.omega;
    s += AtO0 * L.Jp[0]; g += L.Jp[0] * ba_omega_r(a, k, L, 0);
    double AtO1 = 0;
    if (two) { AtO1 = L.Jp[1] * L.omega; s += AtO1 * L.Jp[1]; g += L.Jp[1] * ba_omega_r(a, k, L, 1); }
    hpp += s; b += g;
    if (!a.fixed[e.host]) {
#pragma unroll
      for (int c = 0; c < 6; c++) {
        double v = 0;
        v += AtO0 * L.Jh[c];
        if (two) v += AtO1 * L.Jh[6 + c];
        Hpc[((size_t)p * a.n_poses + e.host) * 6 + c] += v;
      }
    }
    if (!a.fixed[e.target]) {
#pragma unroll
      for (int c = 0; c < 6; c++) {
        double v = 0;
        v += AtO0 * L.Jt[c];
        if (two) v += AtO1 * L.Jt[6 + c];
        Hpc[((size_t)p * a.n_poses + e.target) * 6 + c] += v;
      }
    }
  }
  Hpp[p] = hpp; bp[p] = b;
}

// pr_off / pr_edges: CSR of the edges that touch each block's pose pair (host-built, edge order
// kept): diagonal block i lists every edge with host == i or target == i, block (i, j) every edge
// whose two frames are {i, j} — a block reads only its own edges instead of scanning all of them.
__global__ __launch_bounds__(BA_THREADS) void k_ba_poses(const BaProb* probs, const BaLmDev* lm, int mask)
{
  __shared__ double s_part[BA_WAVES][44];
  const BaProb* PP = ba_window(probs, lm, mask);
  if (!PP) return;
  const BaProb& P = *PP;
  const BaArgs a = P.a;
  const int* pr_off = P.poff; const int* pr_edges = P.plist;
  double* Hcc = P.Hcc; double* bc = P.bc; double* chi2_sum = P.sum;
  // block -> (i, j), i <= j; one extra block sums the chi2 values
  const int np = a.n_poses;
  int b = blockIdx.x, i = 0;
  const int n_pairs = np * (np + 1) / 2;
  if (b > n_pairs) return;   // the grid is sized for the largest window of the launch
  const bool chi_block = (b == n_pairs);
  const int q0 = chi_block ? 0 : pr_off[b], q1 = chi_block ? 0 : pr_off[b + 1];
  int j = 0;
  if (!chi_block) { while (b >= np - i) { b -= np - i; i++; } j = i + b; }
  // a block with a fixed pose stays zero (the linearisation zeroes Hcc / bc first): most blocks of a local-BA window, whose
  // observing keyframes outside the core are all fixed — they leave before the 44-value reduction
  if (!chi_block && (a.fixed[i] || a.fixed[j])) return;
  double acc[44];
#pragma unroll
  for (int q = 0; q < 44; q++) acc[q] = 0;
  if (chi_block) {
    for (int k = threadIdx.x; k < a.n_edges; k += BA_THREADS) { acc[0] += a.edge_chi2[k]; acc[1] += a.edge_rho[k]; }
  } else {
    for (int q = q0 + (int)threadIdx.x; q < q1; q += BA_THREADS) {
      const int k = pr_edges[q];
      const hso_ba_edge& e = a.edges[k];
      const int h = e.host, t = e.target;
      // Everything below is indexed statically (r, c unrolled, the residual's second row behind `two`): the edge's linearisation
      // and the 44 sums stay in registers.  With run-time loop bounds and a run-time choice between L.Jh and L.Jt the
      // record and the sums lived in scratch memory (264 B per lane) and a diagonal block of the newest keyframe — 2-3 k
      // edges, ten per thread — took 150 us.  Operations and their order per sum are unchanged.
      if (i == j) {
        if (h != i && t != i) continue;
        const EdgeLin L = ba_load(a, k);
        const bool two = L.dim > 1, hs = (h == i);
        double Jx[12];
#pragma unroll
        for (int m = 0; m < 12; m++) Jx[m] = hs ? L.Jh[m] : L.Jt[m];
        const double w0 = ba_omega_r(a, k, L, 0), w1 = two ? ba_omega_r(a, k, L, 1) : 0.0;
#pragma unroll
        for (int r = 0; r < 6; r++) {
          acc[36 + r] += Jx[r] * w0;
          if (two) acc[36 + r] += Jx[6 + r] * w1;
#pragma unroll
          for (int c = 0; c < 6; c++) {
            double v = 0;
            v += (Jx[r] * L.omega) * Jx[c];
            if (two) v += (Jx[6 + r] * L.omega) * Jx[6 + c];
            acc[r * 6 + c] += v;
          }
        }
      } else {
        const bool fwd = (h == i && t == j), rev = (h == j && t == i);
        if (!fwd && !rev) continue;
        const EdgeLin L = ba_load(a, k);
        const bool two = L.dim > 1;
        // block (host, target) = Jh^T Omega Jt; stored at (i,j) directly or transposed
        double v[36];
#pragma unroll
        for (int r = 0; r < 6; r++)
#pragma unroll
          for (int c = 0; c < 6; c++) {
            double x = 0;
            x += (L.Jh[r] * L.omega) * L.Jt[c];
            if (two) x += (L.Jh[6 + r] * L.omega) * L.Jt[6 + c];
            v[r * 6 + c] = x;
          }
#pragma unroll
        for (int r = 0; r < 6; r++)
#pragma unroll
          for (int c = 0; c < 6; c++) acc[r * 6 + c] += fwd ? v[r * 6 + c] : v[c * 6 + r];
      }
    }
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int q = 0; q < 44; q++) {
    double x = acc[q];
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
      const int lo = __shfl_xor(__double2loint(x), m), hi = __shfl_xor(__double2hiint(x), m);
      x += __hiloint2double(hi, lo);
    }
    if (lane == 0) s_part[wave][q] = x;
  }
  __syncthreads();
  if (threadIdx.x < 44) {
    double t = 0;
    for (int w = 0; w < BA_WAVES; w++) t += s_part[w][threadIdx.x];
    // ([6], [7]: the linearisation's own copy — a trial launched behind it in the same round overwrites [0], [1])
    if (chi_block) { if (threadIdx.x < 2) { chi2_sum[threadIdx.x] = t; chi2_sum[6 + threadIdx.x] = t; } }
    else if (threadIdx.x < 36) Hcc[((size_t)i * np + j) * 36 + threadIdx.x] = t;
    else if (i == j && threadIdx.x < 42) bc[i * 6 + threadIdx.x - 36] = t;
  }
}


// the blocks a linearisation accumulates into, zeroed for every window of the launch at once (was: one memset per window)
__global__ __launch_bounds__(BA_THREADS) void k_ba_begin(const BaProb* probs, const BaLmDev* lm)
{
  const int want = lm ? lm[blockIdx.y].want : BA_W_LINEARIZE;
  if (!(want & (BA_W_RESTORE | BA_W_LINEARIZE))) return;
  const BaProb& P = probs[blockIdx.y];
  const size_t first = (size_t)blockIdx.x * BA_THREADS + threadIdx.x, step = (size_t)gridDim.x * BA_THREADS;
  if (want & BA_W_RESTORE) {                                        // g2o's pop(): the state before the rejected step
    for (size_t p = first; p < (size_t)P.a.n_points; p += step) P.idist_rw[p] = P.idist_bak[p];
    for (size_t i = first; i < (size_t)P.a.n_poses; i += step) P.poses_rw[i] = P.poses_bak[i];
  }
  if (want & BA_W_LINEARIZE) {
    uint4* z = reinterpret_cast<uint4*>(P.zero_begin);
    const size_t n = P.zero_bytes / 16;
    for (size_t i = first; i < n; i += step) z[i] = make_uint4(0, 0, 0, 0);
  }
}

// computeLambdaInit (thirdparty/g2o/g2o/core/optimization_algorithm_levenberg.cpp:191-201): the largest |diagonal entry| of the
// Hessian blocks of every free vertex — points: Hpp, free poses: the diagonal of their Hcc block — into sum[5].  A maximum does
// not depend on the order it is taken in; the host used to read Hpp and the whole Hcc table of every window for this one number.
__global__ __launch_bounds__(BA_THREADS) void k_ba_maxdiag(const BaProb* probs, const BaLmDev* lm, int mask)
{
  __shared__ double s_part[BA_WAVES];
  const BaProb* PP = ba_window(probs, lm, mask);
  if (!PP) return;
  const BaProb& P = *PP;
  const int np = P.a.n_poses;
  double m = 0;
  for (int p = threadIdx.x; p < P.a.n_points; p += BA_THREADS) m = fmax(m, fabs(P.Hpp[p]));
  for (int k = threadIdx.x; k < np * 6; k += BA_THREADS) {
    const int i = k / 6, q = k - 6 * i;
    if (!P.a.fixed[i]) m = fmax(m, fabs(P.Hcc[((size_t)i * np + i) * 36 + q * 7]));
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) m = fmax(m, __hiloint2double(__shfl_xor(__double2hiint(m), d), __shfl_xor(__double2loint(m), d)));
  if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) { double t = 0; for (int w = 0; w < BA_WAVES; w++) t = fmax(t, s_part[w]); P.sum[5] = t; }
}

// The accept / reject decision of OptimizationAlgorithmLevenberg::solve (optimization_algorithm_levenberg.cpp:61-164) and the
// iteration control of SparseOptimizer::optimize (sparse_optimizer.cpp:354-420) for every window, from the sums the round's
// kernels left: one thread per window.  It writes what the window wants in the next round and the damping its next trial uses
// (negative: computeLambdaInit of the linearisation that precedes it, taken on the device: ba_lambda).  Until round 6 the host
// took this decision — five doubles per window came back and one synchronisation ended every round, ~10 per call; now a call
// enqueues its rounds back to back and waits once.  It runs at the tail of k_ba_chi2, the kernel that ends every round a window
// takes part in (one thread of the window's workgroup, after the sums are out).
__device__ void ba_decide(const double* sum, BaLmDev& L, double* lam_next)
{
  if (L.st == BA_ST_DONE) { L.want = 0; return; }
  hso_ba_result& R = L.res;
  auto finish = [&]() {
    R.stop = L.stop; R.lambda = L.lambda;
    // _optimizer->pop() of a last rejected step still has to happen: one more round that only restores
    if (L.need_restore) { L.need_restore = 0; L.st = BA_ST_FINAL_WAIT; L.want = BA_W_RESTORE; } else { L.st = BA_ST_DONE; L.want = 0; }
  };
  switch (L.st) {
    case BA_ST_INIT_WAIT:
      R.init_chi2 = sum[0]; R.robust_chi2 = sum[1]; R.final_chi2 = sum[0];
      if (L.it >= L.n_iter) { finish(); return; }
      lam_next[0] = -1.0;
      L.st = BA_ST_LIN_WAIT; L.want = BA_W_LINEARIZE | BA_W_TRIAL;
      return;
    case BA_ST_FINAL_WAIT:
      L.st = BA_ST_DONE; L.want = 0;
      return;
    case BA_ST_LIN_WAIT:
      // a linearisation and the first trial behind it ran in ONE round; the linearisation's sums are in [6], [7], [5]
      if (L.first_lin) { R.init_chi2 = sum[6]; R.robust_chi2 = sum[7]; L.first_lin = 0; }
      R.final_chi2 = sum[6];
      L.currentChi = sum[7]; L.tempChi = L.currentChi; L.iniChi = L.currentChi;
      if (L.it == 0) { L.lambda = 1e-5 * sum[5]; L.ni = 2; L.nBad = 0; }   // computeLambdaInit
      L.rho = 0; L.qmax = 0;
      L.st = BA_ST_STEP_WAIT;
      [[fallthrough]];
    case BA_ST_STEP_WAIT: {
      const bool ok2 = sum[4] != 0.0;
      R.n_solves++;
      R.final_chi2 = sum[0];                                        // activeChi2() of the last computeActiveErrors
      L.tempChi = ok2 ? sum[1] : 1.7976931348623157e308;
      double rho = L.currentChi - L.tempChi;
      double scale = sum[2] + sum[3];                               // computeScale (zero when the solve failed: no step)
      scale += 1e-3;
      rho /= scale;
      L.rho = rho;
      bool restore = false;
      if (rho > 0 && isfinite(L.tempChi)) {
        double alpha = 1. - pow(2 * rho - 1, 3.0);
        alpha = fmin(alpha, 2. / 3.);
        L.lambda *= fmax(1. / 3., alpha);
        L.ni = 2;
        L.currentChi = L.tempChi;
        R.n_accepted++;
      } else {
        L.lambda *= L.ni;
        L.ni *= 2;
        restore = true;                                             // _optimizer->pop(): vertices only, edge errors stay
      }
      L.qmax++;
      if (rho < 0 && L.qmax < 5) {                                  // setMaxTrialsAfterFailure(5), src/bundle_adjustment.cpp:571
        lam_next[0] = L.lambda;
        L.st = BA_ST_STEP_WAIT; L.want = (restore ? BA_W_RESTORE : 0) | BA_W_TRIAL;
        return;
      }
      L.need_restore = restore ? 1 : 0;
      R.iterations = L.it + 1;
      R.robust_chi2 = L.currentChi;
      if (L.qmax == 5 || rho == 0) { L.stop = 1; finish(); return; }
      if ((L.iniChi - L.currentChi) * 1e3 < L.iniChi) L.nBad++; else L.nBad = 0;   // optimization_algorithm_levenberg.cpp:154-161
      if (L.nBad >= 3) { L.stop = 2; finish(); return; }
      L.it++;
      if (L.it >= L.n_iter) { finish(); return; }
      lam_next[0] = L.lambda;                                       // the damping of the trial that rides behind the next linearisation
      L.st = BA_ST_LIN_WAIT;
      L.want = BA_W_LINEARIZE | BA_W_TRIAL | (L.need_restore ? BA_W_RESTORE : 0);   // (need_restore: only after a step with a NaN gain ratio)
      L.need_restore = 0;
      return;
    }
  }
}

// sum of chi2 and of the robustified rho(chi2) over all edges (activeChi2 / activeRobustChi2,
// thirdparty/g2o/g2o/core/sparse_optimizer.cpp:100-113): one workgroup, fixed tree => deterministic
__global__ __launch_bounds__(BA_THREADS) void k_ba_chi2(const BaProb* probs, BaLmDev* lm, int mask, int with_scale, double* lam_next)
{
  __shared__ double s_part[BA_WAVES][2];
  // a window whose last step was rejected spends one more round restoring (k_ba_begin): that round ends here
  if (lm && mask == BA_W_TRIAL && lm[blockIdx.y].st == BA_ST_FINAL_WAIT) {
    if (threadIdx.x == 0) { lm[blockIdx.y].st = BA_ST_DONE; lm[blockIdx.y].want = 0; }
    return;
  }
  const BaProb* PP = ba_window(probs, lm, mask);
  if (!PP) return;
  const BaProb& P = *PP;
  const BaArgs a = P.a;
  double* chi2_sum = P.sum;
  double c = 0, r = 0;
  for (int k = threadIdx.x; k < a.n_edges; k += BA_THREADS) { c += a.edge_chi2[k]; r += a.edge_rho[k]; }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) {
    c += __hiloint2double(__shfl_xor(__double2hiint(c), m), __shfl_xor(__double2loint(c), m));
    r += __hiloint2double(__shfl_xor(__double2hiint(r), m), __shfl_xor(__double2loint(r), m));
  }
  if (lane == 0) { s_part[wave][0] = c; s_part[wave][1] = r; }
  __syncthreads();
  if (threadIdx.x < 2) {
    double t = 0;
    for (int w = 0; w < BA_WAVES; w++) t += s_part[w][threadIdx.x];
    chi2_sum[threadIdx.x] = t;
  }
  // the end of an LM trial: the points' part of computeScale from k_ba_backsub's workgroups, in index order
  if (with_scale && threadIdx.x == 2) { double t = 0; for (int g = 0; g < BA_BACKSUB_BLOCKS; g++) t += P.part[g]; chi2_sum[2] = t; }
  if (!lm) return;
  __threadfence_block();
  __syncthreads();
  if (threadIdx.x == 0) ba_decide(chi2_sum, lm[blockIdx.y], lam_next + blockIdx.y);
}

// The reduced system of one LM trial on the device: S = Hcc_free + lambda I - sum_p Hpc_p^T (Hpp_p + lambda)^-1 Hpc_p and
// rhs = bc - sum_p Hpc_p^T bp_p / (Hpp_p + lambda) (the points are 1-D, so the Schur complement is scalar).  One block per
// pose-pair block (i <= j) of the window: threads stride over the points (rows of Hpc of unconnected poses are zero), then
// the fixed tree of k_ba_poses; the block writes its 6x6 piece to both triangles of S.  Only M * M + M doubles go back to the
// host for the dense factorisation — not the n_points x n_poses x 6 block table.
__global__ __launch_bounds__(BA_THREADS) void k_ba_schur(const BaProb* probs, const BaLmDev* lm, int mask)
{
  __shared__ double s_part[BA_WAVES][44];
  const BaProb* PP = ba_window(probs, lm, mask);
  if (!PP) return;
  const BaProb& P = *PP;
  const int np = P.a.n_poses, n_pairs = np * (np + 1) / 2, M = P.M;
  int b = blockIdx.x, i = 0;
  if (b > n_pairs) return;
  const double lambda = ba_lambda(P);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (b == n_pairs) {   // the extra block: is every point diagonal invertible?  (the host solver's `ok`)
    int bad = 0;
    for (int p = threadIdx.x; p < P.a.n_points; p += BA_THREADS) { const double dpp = P.Hpp[p] + lambda; if (!(dpp != 0.0) || !isfinite(dpp)) bad = 1; }
    bad = __syncthreads_or(bad);
    if (threadIdx.x == 0) P.S[(size_t)M * M] = bad ? 0.0 : 1.0;
    return;
  }
  while (b >= np - i) { b -= np - i; i++; }
  const int j = i + b;
  const int ci = P.col[i], cj = P.col[j];
  if (ci < 0 || cj < 0) return;
  double acc[44];
#pragma unroll
  for (int q = 0; q < 44; q++) acc[q] = 0;
  for (int p = threadIdx.x; p < P.a.n_points; p += BA_THREADS) {
    const double dpp = P.Hpp[p] + lambda;
    if (!(dpp != 0.0) || !isfinite(dpp)) continue;
    const double inv = 1.0 / dpp;
    const double* Wi = P.Hpc + ((size_t)p * np + i) * 6;
    const double* Wj = P.Hpc + ((size_t)p * np + j) * 6;
    double wi[6], wj[6];
#pragma unroll
    for (int r = 0; r < 6; r++) { wi[r] = Wi[r]; wj[r] = Wj[r]; }
#pragma unroll
    for (int r = 0; r < 6; r++) {
      const double wr = wi[r] * inv;
#pragma unroll
      for (int c = 0; c < 6; c++) acc[r * 6 + c] += wr * wj[c];
    }
    if (i == j) {
      const double g = P.bp[p] * inv;
#pragma unroll
      for (int r = 0; r < 6; r++) acc[36 + r] += wi[r] * g;
    }
  }
#pragma unroll
  for (int q = 0; q < 44; q++) {
    const double x = wave_butterfly_sum(acc[q]);
    if (lane == 0) s_part[wave][q] = x;
  }
  __syncthreads();
  if (threadIdx.x < 42) {
    double t = 0;
    for (int w = 0; w < BA_WAVES; w++) t += s_part[w][threadIdx.x];
    if (threadIdx.x < 36) {
      const int r = threadIdx.x / 6, c = threadIdx.x % 6;
      double v = P.Hcc[((size_t)i * np + j) * 36 + threadIdx.x];
      if (i == j && r == c) v += lambda;
      v -= t;
      P.S[(size_t)(ci + r) * M + cj + c] = v;
      if (i != j) P.S[(size_t)(cj + c) * M + ci + r] = v;
    } else if (i == j) {
      P.rhs[ci + threadIdx.x - 36] = P.bc[i * 6 + threadIdx.x - 36] - t;
    }
  }
}

// The dense LDL^T of the reduced system (no pivoting, like the reference's SimplicialLDLT), the triangular solves, the pose
// part of computeScale and the pose update SE3Quat::exp(dx) * pose — one workgroup per window, the matrix in LDS.
// Right-looking factorisation: column j is scaled by 1 / d_j, then every entry (i, m) of the trailing lower triangle loses
// (L_ij L_mj) d_j — per entry the same subtractions in the same order (j ascending) as a sequential left-looking loop, spread
// over the threads; the substitutions are column sweeps (x_k final, then x_i -= L_ik x_k for all i at once).  A vanishing or
// non-finite pivot, or a point diagonal that cannot be inverted (flag from k_ba_schur), means "no step": g2o's solver reports
// failure and the trial is rejected.
__global__ __launch_bounds__(BA_THREADS) void k_ba_solve(const BaProb* probs, const BaLmDev* lm, int mask)
{
  extern __shared__ double s_S[];
  __shared__ double s_x[96], s_col[96], s_sc[16];   // s_sc: one entry per FREE pose (M <= 96 unknowns = 16 poses)
  __shared__ int s_ok;
  const BaProb* PP = ba_window(probs, lm, mask);
  if (!PP) return;
  const BaProb& P = *PP;
  const int M = P.M, np = P.a.n_poses, tid = threadIdx.x;
  const double lambda = ba_lambda(P);
  for (int q = tid; q < M * M; q += BA_THREADS) s_S[q] = P.S[q];
  if (tid == 0) s_ok = (P.S[(size_t)M * M] != 0.0) ? 1 : 0;
  __syncthreads();
  for (int j = 0; j < M; j++) {
    if (!s_ok) break;                       // uniform: read after a barrier
    const double dj = s_S[j * M + j];
    if (!(dj != 0.0) || !isfinite(dj)) { __syncthreads(); if (tid == 0) s_ok = 0; __syncthreads(); break; }
    for (int i = j + 1 + tid; i < M; i += BA_THREADS) { const double l = s_S[i * M + j] / dj; s_col[i] = l; }
    __syncthreads();
    // trailing lower triangle (i >= m > j), one entry per thread and step
    const int n = M - j - 1;
    for (int e = tid; e < n * n; e += BA_THREADS) {
      const int ii = e / n, mm = e - ii * n;
      if (mm > ii) continue;
      const int i = j + 1 + ii, m = j + 1 + mm;
      s_S[i * M + m] -= (s_col[i] * s_col[m]) * dj;
    }
    for (int i = j + 1 + tid; i < M; i += BA_THREADS) s_S[i * M + j] = s_col[i];
    __syncthreads();
  }
  __syncthreads();
  const bool ok = s_ok != 0;
  if (ok) {
    for (int i = tid; i < M; i += BA_THREADS) s_x[i] = P.rhs[i];
    __syncthreads();
    for (int k = 0; k < M; k++) {           // L y = rhs
      const double xk = s_x[k];
      __syncthreads();
      for (int i = k + 1 + tid; i < M; i += BA_THREADS) s_x[i] -= s_S[i * M + k] * xk;
      __syncthreads();
    }
    for (int i = tid; i < M; i += BA_THREADS) s_x[i] /= s_S[i * M + i];
    __syncthreads();
    for (int k = M - 1; k >= 0; k--) {      // L^T x = y
      const double xk = s_x[k];
      __syncthreads();
      for (int i = tid; i < k; i += BA_THREADS) s_x[i] -= s_S[k * M + i] * xk;
      __syncthreads();
    }
  }
  // pose steps, push(), SE3Quat::exp(dx) * pose, the pose part of computeScale: one pose per thread
  double* xc = P.trial_rw + 1;
  for (int i = tid; i < np; i += BA_THREADS) {   // any number of poses in the window: the fixed ones (host / neighbour keyframes) only get a zero step
    const int c = P.col[i];
    double x6[6];
    for (int q = 0; q < 6; q++) { x6[q] = (ok && c >= 0) ? s_x[c + q] : 0.0; xc[i * 6 + q] = x6[q]; }
    P.poses_bak[i] = P.poses_rw[i];                                                    // _optimizer->push()
    if (c >= 0) {
      double sc = 0;
      hso_se3 pose = P.poses_rw[i];
      se3quat_exp_times(x6, pose);                                                     // VertexSE3Expmap::oplusImpl
      P.poses_rw[i] = pose;
      for (int q = 0; q < 6; q++) sc += x6[q] * (lambda * x6[q] + P.bc[i * 6 + q]);     // computeScale, pose part
      s_sc[c / 6] = sc;                                                                // columns are handed out in pose order
    }
  }
  __syncthreads();
  if (tid == 0) {
    double sc = 0;
    for (int f = 0; f < M / 6; f++) sc += s_sc[f];                                     // pose order, as the serial loop adds them
    P.trial_rw[1 + 6 * np] = ok ? 0.0 : 1.0;
    P.sum[3] = sc;
    P.sum[4] = ok ? 1.0 : 0.0;
  }
}

// Back-substitution of the points, their update and the point part of computeScale: x_p = (bp_p - Hpc_p . xc) / (Hpp_p +
// lambda) with the poses in index order (products with the zero rows of unconnected poses change nothing, so the value
// equals the sparse loop's); idist_bak keeps the state before the step (g2o's push()).  BA_BACKSUB_BLOCKS workgroups per window
// (one workgroup streamed a window's 2-3 MB of Hpc rows at the latency of 256 threads: 53 us per launch, 8.5 launches per window):
// workgroup g takes the points g * 256 + t, g * 256 + t + 16 * 256, ... and leaves its part of the scale in part[g]; k_ba_chi2,
// which ends every trial, adds the parts in index order — a fixed tree, whatever else runs beside the window.
__global__ __launch_bounds__(BA_THREADS) void k_ba_backsub(const BaProb* probs, const BaLmDev* lm, int mask)
{
  __shared__ double s_part[BA_WAVES];
  const BaProb* PP = ba_window(probs, lm, mask);
  if (!PP) return;
  const BaProb& P = *PP;
  const int np = P.a.n_poses;
  const double lambda = ba_lambda(P);
  const double* xc = P.trial + 1;
  const bool no_step = P.trial[1 + 6 * np] != 0.0;   // the reduced system could not be solved: x = 0 (the trial is rejected)
  double sc = 0;
  for (int p = blockIdx.x * BA_THREADS + threadIdx.x; p < P.a.n_points; p += BA_BACKSUB_BLOCKS * BA_THREADS) {
    const double dpp = P.Hpp[p] + lambda;
    const double inv = 1.0 / dpp;
    double s = P.bp[p];
    for (int i = 0; i < np; i++) {
      if (P.col[i] < 0) continue;
      const double* W = P.Hpc + ((size_t)p * np + i) * 6;
#pragma unroll
      for (int r = 0; r < 6; r++) s -= W[r] * xc[i * 6 + r];
    }
    const double x = no_step ? 0.0 : s * inv;
    P.xp[p] = x;
    const double id = P.idist_rw[p];
    P.idist_bak[p] = id;
    P.idist_rw[p] = id + x;                          // VertexSBAPointID::oplusImpl
    sc += x * (lambda * x + P.bp[p]);
  }
  sc = wave_butterfly_sum(sc);
  if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = sc;
  __syncthreads();
  if (threadIdx.x == 0) { double t = 0; for (int w = 0; w < BA_WAVES; w++) t += s_part[w]; P.part[blockIdx.x] = t; }
}

// g2o's pop() for the points: the state before the rejected step

// ------------------------------------------------------------------ host side

// One BA problem resident in the context's work area: inputs uploaded once, the state (poses, inverse depths)
// refreshed per evaluation, the blocks read back per linearisation.
// ---------------------------------------------------------------------------------------------------------------- host side
// A set of windows laid out one after the other in the context's work area, with one table of BaProb records and the lists
// of windows each launch works on.  Pinned staging mirrors the small per-trial blocks (in) and the small results (out).
struct BaWin {
  int n_poses, n_points, n_edges, n_pairs, M;
  double huber_corner, huber_edge;
  std::vector<int> col;
  // byte offsets inside the window's device slice
  size_t o_trial, o_poses, o_fixed, o_col, small_bytes, o_idist, o_edges, o_off, o_list, o_poff, o_plist, o_uv, o_eobs, in_bytes;
  size_t o_lin, o_rho, o_out, o_Hpp, o_bp, o_Hpc, o_Hcc, o_bc, o_err, o_chi, o_sum, o_S, o_rhs, o_xp, o_bak, o_pbak, o_pcnt, o_cull, total;
  int n_free = 0, n_bins = 0, n_chunks = 0;   // resident windows: free poses, their pair blocks, chunks of RBA_CHUNK edges
  // the device slice comes in pieces.  Value-passing windows: the upload images of all windows of a batch lie side by side (one
  // copy brings them all), the work areas behind them.  Resident windows (hso_gpu_seq_local_ba): only the head of the image — trial
  // block, poses, fixed flags, columns: what the host writes — is uploaded (the heads lie side by side behind the batch header); the
  // rest of the image is filled by kernels.  An offset below small_bytes is in the head, one below in_bytes in the image, any other
  // in the work area.
  char* ds;             // the head MINUS 0 (ds + off, off < small_bytes)
  char* d;              // the image (d + off, small_bytes <= off < in_bytes; value-passing windows: ds == d)
  char* dw;             // the work area MINUS in_bytes (so that dw + o_x is the address of a work table)
  char* h_in;           // pinned: the window's upload image (value-passing) / its head (resident)
  char* at(size_t off) const { return off < small_bytes ? ds + off : (off < in_bytes ? d + off : dw + off); }
};

struct BaBatch {
  hso_gpu_ctx* ctx;
  std::vector<BaWin> win;
  BaProb* d_probs = nullptr;
  BaLmDev* d_lm = nullptr;       // [n] the Levenberg loop's state per window (ba_decide)
  BaLmDev* h_lm = nullptr;       // pinned: the initial states go up with the header
  double* d_lambda = nullptr;    // [n] the damping of each window's current trial
  double* h_lambda = nullptr;    // pinned
  double* d_sums = nullptr;      // [n][8] chi2, robust chi2, scale (points), scale (poses), solvable
  float* d_hub = nullptr;        // [n][2] the Huber deltas formed on the device (hso_gpu_ba_local_multi)
  int n = 0;
};

static bool ba_edges_ok(const hso_ba_edge* edges, int n_edges, int n_points, int n_poses)
{
  for (int k = 0; k < n_edges; k++) {
    const hso_ba_edge& e = edges[k];
    if (e.point < 0 || e.point >= n_points || e.host < 0 || e.host >= n_poses || e.target < 0 || e.target >= n_poses ||
        e.host == e.target || e.level < 0 || e.level > 14) return false;
  }
  return true;
}
static int ba_check_edges(hso_gpu_ctx* ctx, const hso_ba_edge* edges, int n_edges, int n_points, int n_poses, const char* who)
{
  for (int k = 0; k < n_edges; k++) {
    const hso_ba_edge& e = edges[k];
    if (e.point < 0 || e.point >= n_points || e.host < 0 || e.host >= n_poses || e.target < 0 || e.target >= n_poses ||
        e.host == e.target || e.level < 0 || e.level > 14) {
      ctx->err = std::string(who) + ": edge index out of range";
      return HSO_E_INVALID;
    }
  }
  return HSO_OK;
}

// sizes and offsets of one window (no device work)
#define RBA_CHUNK 1024   // edges per workgroup of the pair-block lists' counting sort (resident windows)
static void ba_layout(BaWin& B, int n_poses, int n_points, const uint8_t* pose_fixed, int n_edges,
                      double huber_corner, double huber_edge, bool with_uv = false, bool resident = false)
{
  B.n_poses = n_poses; B.n_points = n_points; B.n_edges = n_edges;
  B.huber_corner = huber_corner; B.huber_edge = huber_edge;
  // (the CSR tables of the edges — by point, by pose-pair block — are built straight into the upload image: ba_stage_window)
  B.n_pairs = n_poses * (n_poses + 1) / 2;
  const int n_pairs = B.n_pairs;
  // the reduced system: free poses in index order
  B.col.assign(n_poses, -1);
  int n_free = 0;
  for (int i = 0; i < n_poses; i++) if (!pose_fixed[i]) B.col[i] = 6 * n_free++;
  B.M = 6 * n_free;
  B.n_free = n_free; B.n_bins = n_free * (n_free + 1) / 2; B.n_chunks = (n_edges + RBA_CHUNK - 1) / RBA_CHUNK;

  auto al = [](size_t b) { return (b + 255) & ~size_t(255); };
  size_t o = 0;
  // [lambda | xc | pad | poses | fixed | col]: the trial block first, the state right behind it; then what only goes up
  B.o_trial = o; o += al(sizeof(double) * (2 + 6 * (size_t)n_poses));   // lambda, xc, "no step" flag
  B.o_poses = o; o += al(sizeof(hso_se3) * n_poses);
  B.o_fixed = o; o += al(n_poses);
  B.o_col = o; o += al(sizeof(int) * n_poses);
  B.small_bytes = resident ? o : 0;
  B.o_idist = o; o += al(sizeof(double) * n_points);
  B.o_edges = o; o += al(sizeof(hso_ba_edge) * n_edges);
  B.o_off = o; o += al(sizeof(int) * (n_points + 1));
  B.o_list = o; o += al(sizeof(int) * n_edges);
  B.o_poff = o; o += al(sizeof(int) * (n_pairs + 1));
  B.o_plist = o; o += al(sizeof(int) * 3 * (size_t)n_edges);
  B.o_uv = o; if (with_uv) o += al(sizeof(double) * 2 * (size_t)n_edges);
  B.o_eobs = o; if (resident) o += al(sizeof(int) * (size_t)n_edges);
  B.in_bytes = o;
  B.o_lin = o; o += al(sizeof(double) * BA_LIN * n_edges);
  B.o_rho = o; o += al(sizeof(double) * n_edges);
  B.o_out = o;
  B.o_Hpp = o; o += al(sizeof(double) * n_points);
  B.o_bp = o; o += al(sizeof(double) * n_points);
  B.o_Hpc = o; o += al(sizeof(double) * (size_t)n_points * n_poses * 6);
  B.o_Hcc = o; o += al(sizeof(double) * (size_t)n_poses * n_poses * 36);
  B.o_bc = o; o += al(sizeof(double) * n_poses * 6);
  B.o_err = o; o += al(sizeof(double) * 2 * n_edges);
  B.o_chi = o; o += al(sizeof(double) * n_edges);
  // the reduced system of a trial: S, the "solvable" flag behind it, rhs (they stay on the device)
  B.o_sum = o; o += 256;
  B.o_S = o; o += sizeof(double) * ((size_t)B.M * B.M + 1);
  B.o_rhs = o; o += sizeof(double) * (size_t)B.M;
  o = al(o);
  B.o_xp = o; o += al(sizeof(double) * n_points);
  B.o_bak = o; o += al(sizeof(double) * n_points);
  B.o_pbak = o; o += al(sizeof(hso_se3) * n_poses);
  B.o_pcnt = o; if (resident) o += al(sizeof(int) * ((size_t)B.n_chunks + 1) * (size_t)std::max(B.n_bins, 1));
  B.o_cull = o; if (resident) o += al(sizeof(int) * (2 + (size_t)n_edges));
  B.total = o;
}

// One window's upload image, written where it is copied from (B.h_in): state, edges, and the two CSR tables of the edges — by
// point (edge order kept inside a point: g2o visits edges in insertion order) and by pose-pair block (same block numbering as
// k_ba_poses; per edge the blocks (h,h), (t,t), (min,max) in that order).  False: an edge index is out of range (nothing to trust).
static bool ba_stage_window(const BaWin& B, const hso_ba_problem& P, const double* obs_uv)
{
  char* w = B.h_in;
  const int n_poses = B.n_poses, n_points = B.n_points, n_edges = B.n_edges, n_pairs = B.n_pairs;
  memset(w + B.o_trial, 0, B.o_poses - B.o_trial);   // lambda, pose steps, the "no step" flag (the gaps between tables are never read)
  memcpy(w + B.o_poses, P.poses_f_w, sizeof(hso_se3) * n_poses);
  memcpy(w + B.o_idist, P.idist, sizeof(double) * n_points);
  memcpy(w + B.o_fixed, P.pose_fixed, n_poses);
  memcpy(w + B.o_col, B.col.data(), sizeof(int) * n_poses);
  if (obs_uv) memcpy(w + B.o_uv, obs_uv, sizeof(double) * 2 * (size_t)n_edges);
  hso_ba_edge* ed = reinterpret_cast<hso_ba_edge*>(w + B.o_edges);
  int* off = reinterpret_cast<int*>(w + B.o_off); int* list = reinterpret_cast<int*>(w + B.o_list);
  int* poff = reinterpret_cast<int*>(w + B.o_poff); int* plist = reinterpret_cast<int*>(w + B.o_plist);
  std::fill(off, off + n_points + 1, 0);
  std::fill(poff, poff + n_pairs + 1, 0);
  auto pair_id = [n_poses](int i, int j) { return i * n_poses - i * (i - 1) / 2 + (j - i); };  // i <= j
  for (int k = 0; k < n_edges; k++) {
    const hso_ba_edge e = P.edges[k];
    if (e.point < 0 || e.point >= n_points || e.host < 0 || e.host >= n_poses || e.target < 0 || e.target >= n_poses ||
        e.host == e.target || e.level < 0 || e.level > 14) return false;
    ed[k] = e;
    off[e.point + 1]++;
    const int h_ = e.host, t_ = e.target;
    poff[pair_id(h_, h_) + 1]++; poff[pair_id(t_, t_) + 1]++;
    poff[pair_id(h_ < t_ ? h_ : t_, h_ < t_ ? t_ : h_) + 1]++;
  }
  for (int p = 0; p < n_points; p++) off[p + 1] += off[p];
  for (int q = 0; q < n_pairs; q++) poff[q + 1] += poff[q];
  thread_local std::vector<int> cur;
  cur.assign(off, off + n_points);
  for (int k = 0; k < n_edges; k++) list[cur[ed[k].point]++] = k;
  cur.assign(poff, poff + n_pairs);
  for (int k = 0; k < n_edges; k++) {
    const int h_ = ed[k].host, t_ = ed[k].target;
    plist[cur[pair_id(h_, h_)]++] = k; plist[cur[pair_id(t_, t_)]++] = k;
    plist[cur[pair_id(h_ < t_ ? h_ : t_, h_ < t_ ? t_ : h_)]++] = k;
  }
  return true;
}

// reserve the work area and staging for all windows, upload everything that does not change between evaluations
// Resident windows (hso_gpu_seq_local_ba; problems == null): `head` writes window q's head image (trial block, poses, fixed flags,
// columns) where it is copied from, the rest of the image is the caller's kernels' to fill; extra_hdr bytes behind the batch
// header go up with it (the caller's own per-window records: *h_extra to write them, *d_extra where they will be).
struct BaResident { std::function<void(int, char*)> head; size_t extra_hdr = 0; char* d_extra = nullptr; char* h_extra = nullptr; };
static int ba_batch_begin(BaBatch& Q, hso_gpu_ctx* ctx, const hso_ba_problem* problems, int n, const double* const* obs_uv = nullptr,
                          BaResident* res = nullptr)
{
  Q.ctx = ctx; Q.n = n;
  HSO_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  auto al = [](size_t b) { return (b + 255) & ~size_t(255); };
  const size_t o_act = al(sizeof(BaProb) * (size_t)n), o_lam = o_act + al(sizeof(BaLmDev) * (size_t)n);
  const size_t o_sums = o_lam + al(sizeof(double) * (size_t)n);
  const size_t o_hub = o_sums + al(sizeof(double) * 8 * (size_t)n);
  const size_t o_extra = o_hub + al(sizeof(float) * 2 * (size_t)n);
  size_t dev = o_extra + al(res ? res->extra_hdr : 0), pin_in = dev;
  const size_t hdr = dev;
  for (int q = 0; q < n; q++) { dev += Q.win[q].total; pin_in += res ? Q.win[q].small_bytes : Q.win[q].in_bytes; }
  if (ctx->batch_cap < dev) {  // grow-only work area of the context (shared with the other batched entry points)
    HSO_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->d_batch) (void)hipFree(ctx->d_batch);
    ctx->d_batch = nullptr; ctx->batch_cap = 0;
    HSO_HIP_CHECK(ctx, hipMalloc(reinterpret_cast<void**>(&ctx->d_batch), hso_grown(dev)));
    ctx->batch_cap = hso_grown(dev);
  }
  char* d = reinterpret_cast<char*>(ctx->d_batch);
  char* h = hso_pinned(ctx, 0, pin_in);
  if (!h) return HSO_E_NOMEM;
  Q.d_probs = reinterpret_cast<BaProb*>(d);
  Q.d_lm = reinterpret_cast<BaLmDev*>(d + o_act);
  Q.d_lambda = reinterpret_cast<double*>(d + o_lam);
  Q.d_sums = reinterpret_cast<double*>(d + o_sums);
  BaProb* hp = reinterpret_cast<BaProb*>(h);
  Q.h_lm = reinterpret_cast<BaLmDev*>(h + o_act);
  Q.h_lambda = reinterpret_cast<double*>(h + o_lam);
  memset(Q.h_lm, 0, sizeof(BaLmDev) * (size_t)n);                  // want == 0: a window takes part in nothing until ba_run sets it up
  for (int q = 0; q < n; q++) Q.h_lambda[q] = -1.0;
  Q.d_hub = reinterpret_cast<float*>(d + o_hub);
  size_t ow = pin_in, oh = hdr;   // [header | upload images (resident: their heads) | the rest]: the pinned block mirrors the first two
  for (int q = 0; q < n; q++) {
    BaWin& B = Q.win[q];
    B.ds = d + oh; B.h_in = h + oh;
    if (res) { B.d = B.dw = d + ow - B.small_bytes; ow += B.total - B.small_bytes; oh += B.small_bytes; }
    else { B.d = B.ds; B.dw = d + ow - B.in_bytes; ow += B.total - B.in_bytes; oh += B.in_bytes; }
  }
  if (res) {
    res->d_extra = d + o_extra; res->h_extra = h + o_extra;
    for (int q = 0; q < n; q++) res->head(q, Q.win[q].h_in);
  } else {
    // the windows' input images, assembled side by side in the page-locked block (tens of megabytes per keyframe step): ONE pass over
    // a window's edges checks their indices, copies them and counts both adjacency tables, a second fills the tables — in place
    std::vector<uint8_t> bad((size_t)n, 0);
    hso_host_parallel(ctx, n, pin_in - hdr, [&](int q) {
      if (!ba_stage_window(Q.win[q], problems[q], obs_uv ? obs_uv[q] : nullptr)) bad[(size_t)q] = 1;
    });
    for (int q = 0; q < n; q++)
      if (bad[(size_t)q]) return ba_check_edges(ctx, problems[q].edges, problems[q].n_edges, problems[q].n_points, problems[q].n_poses, "ba_optimize");
  }
  for (int q = 0; q < n; q++) {
    BaWin& B = Q.win[q];
    BaProb& R = hp[q];
    R.a.poses = reinterpret_cast<const hso_se3*>(B.at(B.o_poses)); R.a.fixed = reinterpret_cast<const uint8_t*>(B.at(B.o_fixed));
    R.a.idist = reinterpret_cast<const double*>(B.at(B.o_idist)); R.a.edges = reinterpret_cast<const hso_ba_edge*>(B.at(B.o_edges));
    R.a.n_poses = B.n_poses; R.a.n_points = B.n_points; R.a.n_edges = B.n_edges;
    R.a.huber_corner = B.huber_corner; R.a.huber_edge = B.huber_edge;
    R.a.lin = reinterpret_cast<double*>(B.at(B.o_lin)); R.a.edge_err = reinterpret_cast<double*>(B.at(B.o_err));
    R.a.edge_chi2 = reinterpret_cast<double*>(B.at(B.o_chi)); R.a.edge_rho = reinterpret_cast<double*>(B.at(B.o_rho));
    R.off = reinterpret_cast<const int*>(B.at(B.o_off)); R.list = reinterpret_cast<const int*>(B.at(B.o_list));
    R.poff = reinterpret_cast<const int*>(B.at(B.o_poff)); R.plist = reinterpret_cast<const int*>(B.at(B.o_plist));
    R.Hpp = reinterpret_cast<double*>(B.at(B.o_Hpp)); R.bp = reinterpret_cast<double*>(B.at(B.o_bp));
    R.Hpc = reinterpret_cast<double*>(B.at(B.o_Hpc)); R.Hcc = reinterpret_cast<double*>(B.at(B.o_Hcc));
    R.bc = reinterpret_cast<double*>(B.at(B.o_bc)); R.sum = Q.d_sums + 8 * (size_t)q; R.lam = Q.d_lambda + q;
    R.col = reinterpret_cast<const int*>(B.at(B.o_col));
    R.S = reinterpret_cast<double*>(B.at(B.o_S)); R.rhs = reinterpret_cast<double*>(B.at(B.o_rhs));
    R.trial = reinterpret_cast<const double*>(B.at(B.o_trial));
    R.xp = reinterpret_cast<double*>(B.at(B.o_xp)); R.part = reinterpret_cast<double*>(B.at(B.o_sum));
    R.idist_rw = reinterpret_cast<double*>(B.at(B.o_idist)); R.idist_bak = reinterpret_cast<double*>(B.at(B.o_bak));
    R.trial_rw = reinterpret_cast<double*>(B.at(B.o_trial));
    R.poses_rw = reinterpret_cast<hso_se3*>(B.at(B.o_poses)); R.poses_bak = reinterpret_cast<hso_se3*>(B.at(B.o_pbak));
    R.M = B.M; R.n_pairs = B.n_pairs;
    R.zero_begin = B.at(B.o_out); R.zero_bytes = B.o_sum + 256 - B.o_out;
    R.uv = (obs_uv || res) ? reinterpret_cast<const double*>(B.at(B.o_uv)) : nullptr;
    R.mad = reinterpret_cast<float*>(B.at(B.o_err)); R.hub = Q.d_hub + 2 * (size_t)q;
  }
  // the window records, the (not yet filled) launch lists and every window's upload image in ONE copy (it was one per window)
  HSO_HIP_CHECK(ctx, hipMemcpyAsync(d, h, pin_in, hipMemcpyHostToDevice, ctx->stream));
  return HSO_OK;
}

// Launch helpers: `which` = the windows (indices) this launch works on, written to list slot `slot` (a slot may be reused
// only after a synchronise, which every round of the driver ends with).
static int ba_max(const BaBatch& Q, int BaWin::*field)
{
  int m = 0;
  for (const BaWin& B : Q.win) m = std::max(m, B.*field);
  return m;
}
// Launch helpers.  Every kernel runs for all windows of the batch (blockIdx.y = window); lm != null: a window takes part when its
// state record wants `mask` (the loop), lm == null: every window does (a call outside the loop).
// computeActiveErrors + buildSystem at the resident state; the blocks stay on the device
static int ba_launch_linearize(BaBatch& Q, const BaLmDev* lm)
{
  hso_gpu_ctx* ctx = Q.ctx;
  const int ny = Q.n, m = BA_W_LINEARIZE;
  hipLaunchKernelGGL(k_ba_begin, dim3(64, ny), dim3(BA_THREADS), 0, ctx->stream, Q.d_probs, lm);   // (the loop: also the pop() of a rejected step)
  hipLaunchKernelGGL(k_ba_edges<true>, dim3((ba_max(Q, &BaWin::n_edges) + BA_THREADS - 1) / BA_THREADS, ny), dim3(BA_THREADS), 0, ctx->stream, Q.d_probs, lm, m);
  hipLaunchKernelGGL(k_ba_points, dim3((ba_max(Q, &BaWin::n_points) + BA_THREADS - 1) / BA_THREADS, ny), dim3(BA_THREADS), 0, ctx->stream, Q.d_probs, lm, m);
  hipLaunchKernelGGL(k_ba_poses, dim3(ba_max(Q, &BaWin::n_pairs) + 1, ny), dim3(BA_THREADS), 0, ctx->stream, Q.d_probs, lm, m);
  hipLaunchKernelGGL(k_ba_maxdiag, dim3(1, ny), dim3(BA_THREADS), 0, ctx->stream, Q.d_probs, lm, m);
  HSO_HIP_CHECK(ctx, hipGetLastError());
  return HSO_OK;
}
// computeActiveErrors only: per-edge error / chi2 / rho and the two sums (mask BA_W_ERRORS: the errors-only round; BA_W_TRIAL: the end
// of an LM trial, where k_ba_chi2 also adds up the back-substitution's parts of computeScale)
static int ba_launch_errors(BaBatch& Q, BaLmDev* lm, int mask)
{
  hso_gpu_ctx* ctx = Q.ctx;
  const int ny = Q.n;
  hipLaunchKernelGGL(k_ba_edges<false>, dim3((ba_max(Q, &BaWin::n_edges) + BA_THREADS - 1) / BA_THREADS, ny), dim3(BA_THREADS), 0, ctx->stream, Q.d_probs, lm, mask);
  hipLaunchKernelGGL(k_ba_chi2, dim3(1, ny), dim3(BA_THREADS), 0, ctx->stream, Q.d_probs, lm, mask, mask == BA_W_TRIAL ? 1 : 0, Q.d_lambda);
  HSO_HIP_CHECK(ctx, hipGetLastError());
  return HSO_OK;
}
static int ba_get(BaBatch& Q, int q, void* dst, size_t off, size_t bytes)
{
  HSO_HIP_CHECK(Q.ctx, hipMemcpyAsync(dst, Q.win[q].at(off), bytes, hipMemcpyDeviceToHost, Q.ctx->stream));
  return HSO_OK;
}

extern "C" int hso_gpu_ba_linearize(hso_gpu_ctx* ctx, const hso_se3* poses_f_w, const uint8_t* pose_fixed, int n_poses,
                                    const double* idist, int n_points, const hso_ba_edge* edges, int n_edges,
                                    double huber_corner, double huber_edge, double* Hpp, double* bp, double* Hpc,
                                    double* Hcc, double* bc, double* edge_err, double* edge_chi2, double* chi2_sum)
{
  if (!ctx) return HSO_E_INVALID;
  if (!poses_f_w || !pose_fixed || !idist || !edges || n_poses <= 0 || n_points <= 0 || n_edges <= 0 || !Hpp || !bp || !Hpc ||
      !Hcc || !bc || !edge_err || !edge_chi2 || !chi2_sum)
    return hso_fail(ctx, HSO_E_INVALID, "ba_linearize: bad argument");
  if (int rc = ba_check_edges(ctx, edges, n_edges, n_points, n_poses, "ba_linearize")) return rc;
  BaBatch Q;
  Q.win.resize(1);
  ba_layout(Q.win[0], n_poses, n_points, pose_fixed, n_edges, huber_corner, huber_edge);
  hso_ba_problem P;
  memset(&P, 0, sizeof(P));
  P.poses_f_w = const_cast<hso_se3*>(poses_f_w); P.pose_fixed = pose_fixed; P.idist = const_cast<double*>(idist); P.edges = edges;
  if (int rc = ba_batch_begin(Q, ctx, &P, 1)) return rc;
  if (int rc = ba_launch_linearize(Q, nullptr)) return rc;
  const BaWin& B = Q.win[0];
  int rc = HSO_OK;
  if (!rc) rc = ba_get(Q, 0, Hpp, B.o_Hpp, sizeof(double) * n_points);
  if (!rc) rc = ba_get(Q, 0, bp, B.o_bp, sizeof(double) * n_points);
  if (!rc) rc = ba_get(Q, 0, Hpc, B.o_Hpc, sizeof(double) * (size_t)n_points * n_poses * 6);
  if (!rc) rc = ba_get(Q, 0, Hcc, B.o_Hcc, sizeof(double) * (size_t)n_poses * n_poses * 36);
  if (!rc) rc = ba_get(Q, 0, bc, B.o_bc, sizeof(double) * n_poses * 6);
  if (!rc) rc = ba_get(Q, 0, edge_err, B.o_err, sizeof(double) * 2 * n_edges);
  if (!rc) rc = ba_get(Q, 0, edge_chi2, B.o_chi, sizeof(double) * n_edges);
  if (!rc) { HSO_HIP_CHECK(ctx, hipMemcpyAsync(chi2_sum, Q.d_sums, sizeof(double) * 2, hipMemcpyDeviceToHost, ctx->stream)); }
  if (rc) return rc;
  HSO_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  return HSO_OK;
}

// hso::getMedian (include/hso/vikit/math_utils.h:119-126): nth_element at floor(n/2)
static float upper_median(std::vector<float>& v)
{
  std::nth_element(v.begin(), v.begin() + v.size() / 2, v.end());
  return v[v.size() / 2];
}

// per-edge error magnitudes for the Huber deltas of LocalBundleAdjustment (src/bundle_adjustment.cpp:618-656):
// e = (project2d(obs->f) - project2d(Tth * fH / idist)) / 2^level; corners |e|, edgelets |grad^T e| (floats).
// One table of (tables, edge count) per window, blockIdx.y = window.
struct MadWin { const hso_se3* poses; const double* idist; const hso_ba_edge* edges; const double* uv; float* err; int n_edges, pad_; };

__global__ __launch_bounds__(BA_THREADS) void k_ba_mad_errors(const MadWin* wins)
{
  const MadWin W = wins[blockIdx.y];
  for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < W.n_edges; k += gridDim.x * blockDim.x) {
    const hso_ba_edge e = W.edges[k];
    const Se3 Tth = se3_mul(se3_from(W.poses[e.target]), se3_inverse(se3_from(W.poses[e.host])));
    const double inv = 1.0 / W.idist[e.point];
    double x, y, z;
    se3_apply(Tth, e.fH[0] * inv, e.fH[1] * inv, e.fH[2] * inv, x, y, z);
    double ex = W.uv[2 * k] - x / z, ey = W.uv[2 * k + 1] - y / z;
    const double sc = 1.0 / (double)(1 << e.level);
    ex *= sc; ey *= sc;
    W.err[k] = (e.type == HSO_FTR_EDGELET) ? (float)fabs(e.normal[0] * ex + e.normal[1] * ey) : (float)sqrt(ex * ex + ey * ey);
  }
}

// The same error magnitudes for the windows of an optimisation batch (hso_gpu_ba_local_multi): the tables are the window's own,
// already on the device for the optimisation that follows; blockIdx.y = window.
__global__ __launch_bounds__(BA_THREADS) void k_ba_mad_errors_win(const BaProb* probs)
{
  const BaProb& P = probs[blockIdx.y];
  for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < P.a.n_edges; k += gridDim.x * blockDim.x) {
    const hso_ba_edge e = P.a.edges[k];
    const Se3 Tth = se3_mul(se3_from(P.a.poses[e.target]), se3_inverse(se3_from(P.a.poses[e.host])));
    const double inv = 1.0 / P.a.idist[e.point];
    double x, y, z;
    se3_apply(Tth, e.fH[0] * inv, e.fH[1] * inv, e.fH[2] * inv, x, y, z);
    double ex = P.uv[2 * k] - x / z, ey = P.uv[2 * k + 1] - y / z;
    const double sc = 1.0 / (double)(1 << e.level);
    ex *= sc; ey *= sc;
    P.mad[k] = (e.type == HSO_FTR_EDGELET) ? (float)fabs(e.normal[0] * ex + e.normal[1] * ey) : (float)sqrt(ex * ex + ey * ey);
  }
}

// huber_corner / huber_edge of a window (src/bundle_adjustment.cpp:664-680): 1.4826 x hso::getMedian of the corner / edgelet error
// magnitudes = the element nth_element leaves at floor(n / 2) (include/hso/vikit/math_utils.h:119-126).  The magnitudes are
// non-negative floats, so their bit patterns order like their values: an exact radix select (four 8-bit digits, histogram in LDS)
// per kind; one workgroup per window.  The deltas go into the window record the optimisation's kernels read, and out as floats.
#define BA_MAD_THREADS 1024   // one workgroup per window sweeps its edges eight times (two kinds, four digits)
__global__ __launch_bounds__(BA_MAD_THREADS) void k_ba_mad_select(BaProb* probs, double error_multiplier2)
{
  BaProb& P = probs[blockIdx.x];
  const int n = P.a.n_edges, tid = threadIdx.x;
  const hso_ba_edge* E = P.a.edges; const float* err = P.mad;
  __shared__ unsigned s_hist[256];
  __shared__ unsigned s_prefix, s_rank;
  __shared__ int s_cnt[2];
  if (tid < 2) s_cnt[tid] = 0;
  __syncthreads();
  { int c[2] = {0, 0}; for (int k = tid; k < n; k += BA_MAD_THREADS) c[E[k].type == HSO_FTR_EDGELET ? 1 : 0]++; if (c[0]) atomicAdd(&s_cnt[0], c[0]); if (c[1]) atomicAdd(&s_cnt[1], c[1]); }
  __syncthreads();
  float med[2] = {0.f, 0.f};
  for (int kind = 0; kind < 2; kind++) {
    const int cnt = s_cnt[kind];
    if (cnt == 0) continue;            // uniform: every thread reads the same shared count
    if (tid == 0) { s_prefix = 0u; s_rank = (unsigned)(cnt / 2); }
    unsigned mask = 0u;
    for (int shift = 24; shift >= 0; shift -= 8) {
      if (tid < 256) s_hist[tid] = 0u;
      __syncthreads();
      const unsigned prefix = s_prefix;
      for (int k = tid; k < n; k += BA_MAD_THREADS) {
        if ((E[k].type == HSO_FTR_EDGELET ? 1 : 0) != kind) continue;
        const unsigned key = __float_as_uint(err[k]);
        if ((key & mask) == prefix) atomicAdd(&s_hist[(key >> shift) & 255u], 1u);
      }
      __syncthreads();
      if (tid == 0) {
        unsigned r = s_rank, b = 0;
        for (; b < 255u; b++) { const unsigned hcount = s_hist[b]; if (r < hcount) break; r -= hcount; }
        s_rank = r; s_prefix = prefix | (b << shift);
      }
      mask |= 255u << shift;
      __syncthreads();
    }
    med[kind] = __uint_as_float(s_prefix);
    __syncthreads();
  }
  if (tid == 0) {
    float hc = 0.f, he = 0.f;
    const bool pt = s_cnt[0] > 0, ls = s_cnt[1] > 0;
    if (pt && ls) { hc = (float)(1.4826 * (double)med[0]); he = (float)(1.4826 * (double)med[1]); }
    else if (!pt && ls) { hc = (float)(1.0 / error_multiplier2); he = (float)(1.4826 * (double)med[1]); }
    else if (pt && !ls) { hc = (float)(1.4826 * (double)med[0]); he = (float)(0.5 / error_multiplier2); }
    P.a.huber_corner = (double)hc; P.a.huber_edge = (double)he;
    P.hub[0] = hc; P.hub[1] = he;
  }
}

// All windows of a step in one upload, one launch, one read-back and one synchronisation (a keyframe step of a bank of sequences
// asks for two dozen windows' deltas; one call each was two dozen host round trips).
extern "C" int hso_gpu_ba_huber_deltas_multi(hso_gpu_ctx* ctx, hso_ba_deltas_job* jobs, int n_jobs, double error_multiplier2)
{
  if (!ctx) return HSO_E_INVALID;
  if (n_jobs < 0 || (n_jobs > 0 && !jobs)) return hso_fail(ctx, HSO_E_INVALID, "ba_huber_deltas: bad argument");
  auto al = [](size_t b) { return (b + 255) & ~size_t(255); };
  struct Lay { size_t o_poses, o_idist, o_edges, o_uv, o_err; };
  std::vector<Lay> lay((size_t)n_jobs);
  std::vector<int> live;
  size_t o = al(sizeof(MadWin) * (size_t)std::max(n_jobs, 1));
  int max_edges = 0;
  for (int j = 0; j < n_jobs; j++) {
    hso_ba_deltas_job& J = jobs[j];
    if (!J.poses_f_w || !J.idist || J.n_poses <= 0 || J.n_points <= 0 || J.n_edges < 0 || (J.n_edges > 0 && (!J.edges || !J.obs_uv)))
      return hso_fail(ctx, HSO_E_INVALID, "ba_huber_deltas: bad argument");
    J.huber_corner = 0; J.huber_edge = 0;
    if (J.n_edges == 0) continue;   // both error lists empty: the reference leaves the deltas uninitialised
    Lay& L = lay[(size_t)j];
    L.o_poses = o; o += al(sizeof(hso_se3) * (size_t)J.n_poses);
    L.o_idist = o; o += al(sizeof(double) * (size_t)J.n_points);
    L.o_edges = o; o += al(sizeof(hso_ba_edge) * (size_t)J.n_edges);
    L.o_uv = o; o += al(sizeof(double) * 2 * (size_t)J.n_edges);
    live.push_back(j); max_edges = std::max(max_edges, (int)J.n_edges);
  }
  if (live.empty()) return HSO_OK;
  {
    std::vector<uint8_t> bad(live.size(), 0);
    hso_host_parallel(ctx, (int)live.size(), o, [&](int wi) {
      const hso_ba_deltas_job& J = jobs[live[(size_t)wi]];
      bad[(size_t)wi] = ba_edges_ok(J.edges, J.n_edges, J.n_points, J.n_poses) ? 0 : 1;
    });
    for (size_t w = 0; w < live.size(); w++)
      if (bad[w]) { const hso_ba_deltas_job& J = jobs[live[w]]; return ba_check_edges(ctx, J.edges, J.n_edges, J.n_points, J.n_poses, "ba_huber_deltas"); }
  }
  const size_t in_bytes = o;
  for (int j : live) { lay[(size_t)j].o_err = o; o += al(sizeof(float) * (size_t)jobs[j].n_edges); }
  const size_t err_bytes = o - in_bytes;
  HSO_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  if (ctx->batch_cap < o) {
    HSO_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->d_batch) (void)hipFree(ctx->d_batch);
    ctx->d_batch = nullptr; ctx->batch_cap = 0;
    HSO_HIP_CHECK(ctx, hipMalloc(reinterpret_cast<void**>(&ctx->d_batch), hso_grown(o)));
    ctx->batch_cap = hso_grown(o);
  }
  char* d = reinterpret_cast<char*>(ctx->d_batch);
  char* h = hso_pinned(ctx, 0, in_bytes);
  char* he = hso_pinned(ctx, 1, err_bytes);
  if (!h || !he) return HSO_E_NOMEM;
  MadWin* hw = reinterpret_cast<MadWin*>(h);
  hso_host_parallel(ctx, (int)live.size(), in_bytes, [&](int wi) {
    const size_t w = (size_t)wi;
    const hso_ba_deltas_job& J = jobs[live[w]];
    const Lay& L = lay[(size_t)live[w]];
    memcpy(h + L.o_poses, J.poses_f_w, sizeof(hso_se3) * (size_t)J.n_poses);
    memcpy(h + L.o_idist, J.idist, sizeof(double) * (size_t)J.n_points);
    memcpy(h + L.o_edges, J.edges, sizeof(hso_ba_edge) * (size_t)J.n_edges);
    memcpy(h + L.o_uv, J.obs_uv, sizeof(double) * 2 * (size_t)J.n_edges);
    hw[w].poses = reinterpret_cast<const hso_se3*>(d + L.o_poses); hw[w].idist = reinterpret_cast<const double*>(d + L.o_idist);
    hw[w].edges = reinterpret_cast<const hso_ba_edge*>(d + L.o_edges); hw[w].uv = reinterpret_cast<const double*>(d + L.o_uv);
    hw[w].err = reinterpret_cast<float*>(d + L.o_err); hw[w].n_edges = J.n_edges; hw[w].pad_ = 0;
  });
  HSO_HIP_CHECK(ctx, hipMemcpyAsync(d, h, in_bytes, hipMemcpyHostToDevice, ctx->stream));
  hipLaunchKernelGGL(k_ba_mad_errors, dim3((max_edges + BA_THREADS - 1) / BA_THREADS, (unsigned)live.size()), dim3(BA_THREADS), 0, ctx->stream,
                     reinterpret_cast<const MadWin*>(d));
  HSO_HIP_CHECK(ctx, hipGetLastError());
  HSO_HIP_CHECK(ctx, hipMemcpyAsync(he, d + in_bytes, err_bytes, hipMemcpyDeviceToHost, ctx->stream));
  HSO_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  // the medians, a window per item (an nth_element over tens of thousands of errors each)
  hso_host_parallel(ctx, (int)live.size(), err_bytes * 16, [&](int wi) {
    const int j = live[(size_t)wi];
    hso_ba_deltas_job& J = jobs[j];
    const float* err = reinterpret_cast<const float*>(he + (lay[(size_t)j].o_err - in_bytes));
    std::vector<float> errors_pt, errors_ls;
    for (int k = 0; k < J.n_edges; k++) (J.edges[k].type == HSO_FTR_EDGELET ? errors_ls : errors_pt).push_back(err[k]);
    // src/bundle_adjustment.cpp:664-680
    if (!errors_pt.empty() && !errors_ls.empty()) {
      J.huber_corner = (float)(1.4826 * upper_median(errors_pt));
      J.huber_edge = (float)(1.4826 * upper_median(errors_ls));
    } else if (errors_pt.empty() && !errors_ls.empty()) {
      J.huber_corner = (float)(1.0 / error_multiplier2);
      J.huber_edge = (float)(1.4826 * upper_median(errors_ls));
    } else if (!errors_pt.empty() && errors_ls.empty()) {
      J.huber_corner = (float)(1.4826 * upper_median(errors_pt));
      J.huber_edge = (float)(0.5 / error_multiplier2);
    }
  });
  return HSO_OK;
}

extern "C" int hso_gpu_ba_huber_deltas(hso_gpu_ctx* ctx, const hso_se3* poses_f_w, int n_poses, const double* idist, int n_points,
                                       const hso_ba_edge* edges, const double* obs_uv, int n_edges, double error_multiplier2,
                                       float* huber_corner, float* huber_edge)
{
  if (!ctx) return HSO_E_INVALID;
  if (!huber_corner || !huber_edge) return hso_fail(ctx, HSO_E_INVALID, "ba_huber_deltas: bad argument");
  hso_ba_deltas_job J;
  J.poses_f_w = poses_f_w; J.n_poses = n_poses; J.idist = idist; J.n_points = n_points; J.edges = edges; J.obs_uv = obs_uv; J.n_edges = n_edges;
  J.huber_corner = 0; J.huber_edge = 0;
  *huber_corner = 0; *huber_edge = 0;
  const int rc = hso_gpu_ba_huber_deltas_multi(ctx, &J, 1, error_multiplier2);
  if (rc == HSO_OK) { *huber_corner = J.huber_corner; *huber_edge = J.huber_edge; }
  return rc;
}

// ---- g2o::SE3Quat on the host (thirdparty/g2o/g2o/types/se3quat.h): the pose update of VertexSE3Expmap ----
namespace {

}  // namespace

// The Levenberg loop of OptimizationAlgorithmLevenberg::solve for the windows of a batch, in lockstep rounds.  A round = for every
// window the device work its state record asks for — restore (g2o's pop()), linearisation (computeActiveErrors + buildSystem), an LM
// trial as ONE device sequence: reduced system (k_ba_schur), dense LDL^T + pose update (k_ba_solve), back-substitution + point
// update (k_ba_backsub), error evaluation (k_ba_edges<false>, k_ba_chi2) — and at k_ba_chi2's tail ba_decide, which takes the accept / reject
// decision and says what the window does next.  Each kind of work is launched ONCE per round for all windows (blockIdx.y = window;
// a window that does not want it leaves at once), so the small kernels of different windows run side by side; a window's arithmetic
// does not depend on what runs beside it.  The host enqueues rounds back to back and looks at the state records only every
// BA_ROUNDS_FIRST / BA_ROUNDS_MORE rounds (a window of the engine needs 8-10): one wait per call where every round had one.
// `finalise` (resident windows): enqueued behind every block of rounds — the write-back kernels of the windows that are done — and
// told which read-backs to make; value-passing windows hand their state back to the caller's arrays.
#define BA_ROUNDS_FIRST 10
#define BA_ROUNDS_MORE 4
struct BaLmHost {   // what the caller gives and gets per window
  hso_se3* poses_f_w; double* idist; double* edge_chi2_out; hso_ba_result* result; int n_iter;
};
typedef std::function<int(std::vector<HsoListCopy>& back)> BaFinalHook;
static int ba_run(hso_gpu_ctx* ctx, BaBatch& Q, const std::vector<BaLmHost>& lm, float* huber_out, const BaFinalHook* finalise)
{
  const int n = Q.n;
  // the loop's initial state per window (runSparseBAOptimizer: computeActiveErrors(); init_error = activeChi2()).  The first
  // linearisation evaluates the errors at the very state init_error is taken at, and sums them the same way (k_ba_poses' extra
  // block = k_ba_chi2's reduction): with at least one iteration to run, the errors-only round is left out and init_chi2 comes
  // from the first linearisation's sums
  bool any_errors_round = false;
  for (int q = 0; q < n; q++) {
    BaLmDev& L = Q.h_lm[q];
    memset(&L, 0, sizeof(L));
    L.lambda = -1.; L.ni = 2.; L.n_iter = lm[(size_t)q].n_iter;
    L.first_lin = L.n_iter > 0 ? 1 : 0;
    if (L.first_lin) { L.st = BA_ST_LIN_WAIT; L.want = BA_W_LINEARIZE | BA_W_TRIAL; } else { L.st = BA_ST_INIT_WAIT; L.want = BA_W_ERRORS; any_errors_round = true; }
    Q.h_lambda[q] = -1.0;
  }
  HSO_HIP_CHECK(ctx, hipMemcpyAsync(Q.d_lm, Q.h_lm, (size_t)(reinterpret_cast<char*>(Q.h_lambda + n) - reinterpret_cast<char*>(Q.h_lm)), hipMemcpyHostToDevice, ctx->stream));
  // 96 x 96 doubles = 72 KiB of dynamic LDS: above the 64 KiB a launch gets without asking (once per process; several engines call this
  // function from their own threads)
  static std::once_flag solve_attr;
  hipError_t attr_rc = hipSuccess;
  std::call_once(solve_attr, [&] { attr_rc = hipFuncSetAttribute(reinterpret_cast<const void*>(k_ba_solve), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 96 * (int)sizeof(double)); });
  HSO_HIP_CHECK(ctx, attr_rc);
  const int max_m = ba_max(Q, &BaWin::M), max_pairs = ba_max(Q, &BaWin::n_pairs);
  std::vector<BaLmDev> state((size_t)n);
  std::vector<float> hub(2 * (size_t)n);
  bool first = true;
  for (int rounds = BA_ROUNDS_FIRST;; rounds = BA_ROUNDS_MORE) {
    for (int r = 0; r < rounds; r++) {
      // (the errors-only round of a window that runs zero iterations: only the call's first round can hold one)
      if (first && any_errors_round) { if (int rc = ba_launch_errors(Q, Q.d_lm, BA_W_ERRORS)) return rc; }
      if (int rc = ba_launch_linearize(Q, Q.d_lm)) return rc;
      hipLaunchKernelGGL(k_ba_schur, dim3(max_pairs + 1, n), dim3(BA_THREADS), 0, ctx->stream, Q.d_probs, Q.d_lm, (int)BA_W_TRIAL);
      hipLaunchKernelGGL(k_ba_solve, dim3(1, n), dim3(BA_THREADS), sizeof(double) * (size_t)std::max(max_m * max_m, 1), ctx->stream, Q.d_probs, Q.d_lm, (int)BA_W_TRIAL);
      hipLaunchKernelGGL(k_ba_backsub, dim3(BA_BACKSUB_BLOCKS, n), dim3(BA_THREADS), 0, ctx->stream, Q.d_probs, Q.d_lm, (int)BA_W_TRIAL);
      if (int rc = ba_launch_errors(Q, Q.d_lm, BA_W_TRIAL)) return rc;   // ... and the round's decision (ba_decide)
      first = false;
    }
    // what comes back: the state records (is every window done?), once the Huber deltas, and the finished windows' results
    std::vector<HsoListCopy> back;
    back.push_back({state.data(), Q.d_lm, sizeof(BaLmDev) * (size_t)n});
    if (huber_out) back.push_back({hub.data(), Q.d_hub, sizeof(float) * 2 * (size_t)n});
    if (finalise) { if (int rc = (*finalise)(back)) return rc; }
    else for (int q = 0; q < n; q++) {
      const BaWin& B = Q.win[(size_t)q];
      back.push_back({lm[(size_t)q].idist, B.at(B.o_idist), sizeof(double) * (size_t)B.n_points});
      back.push_back({lm[(size_t)q].poses_f_w, B.at(B.o_poses), sizeof(hso_se3) * (size_t)B.n_poses});
      if (lm[(size_t)q].edge_chi2_out) back.push_back({lm[(size_t)q].edge_chi2_out, B.at(B.o_chi), sizeof(double) * (size_t)B.n_edges});
    }
    if (int rc = hso_lists_to_host(ctx, back)) return rc;   // synchronises
    bool done = true;
    for (int q = 0; q < n; q++) done = done && state[(size_t)q].st == BA_ST_DONE;
    if (done) break;
  }
  for (int q = 0; q < n; q++) *lm[(size_t)q].result = state[(size_t)q].res;
  if (huber_out) memcpy(huber_out, hub.data(), sizeof(float) * 2 * (size_t)n);
  return HSO_OK;
}

// obs_uv != null: the Huber deltas are formed on the device from the windows' initial state first (hso_gpu_ba_local_multi) and
// returned in huber_out [2 * n_problems]; the problems' own huber_corner / huber_edge are ignored then
static int ba_optimize_multi_impl(hso_gpu_ctx* ctx, const hso_ba_problem* problems, int n_problems, const double* const* obs_uv,
                                  double error_multiplier2, float* huber_out)
{
  if (!ctx) return HSO_E_INVALID;
  if (n_problems < 0 || (n_problems > 0 && !problems)) return hso_fail(ctx, HSO_E_INVALID, "ba_optimize_multi: bad argument");
  if (n_problems == 0) return HSO_OK;
  if (obs_uv) {
    if (!huber_out) return hso_fail(ctx, HSO_E_INVALID, "ba_local_multi: bad argument");
    for (int q = 0; q < n_problems; q++) if (!obs_uv[q]) return hso_fail(ctx, HSO_E_INVALID, "ba_local_multi: bad argument");
  }
  BaBatch Q;
  Q.win.resize(n_problems);
  for (int q = 0; q < n_problems; q++) {
    const hso_ba_problem& P = problems[q];
    if (!P.poses_f_w || !P.pose_fixed || !P.idist || !P.edges || !P.result || P.n_poses <= 0 || P.n_points <= 0 || P.n_edges <= 0 || P.n_iter < 0)
      return hso_fail(ctx, HSO_E_INVALID, "ba_optimize: bad argument");
  }
  for (int q = 0; q < n_problems; q++) {
    const hso_ba_problem& P = problems[q];
    ba_layout(Q.win[q], P.n_poses, P.n_points, P.pose_fixed, P.n_edges, P.huber_corner, P.huber_edge, obs_uv != nullptr);
    if (Q.win[q].M > 96) return hso_fail(ctx, HSO_E_INVALID, "ba_optimize: more than 16 free poses in one window (the reference's core is 7 keyframes)");
  }
  // (the edges' index check rides in the staging pass of ba_batch_begin)
  if (int rc = ba_batch_begin(Q, ctx, problems, n_problems, obs_uv)) return rc;
  if (obs_uv) {
    int max_edges = 0;
    for (int q = 0; q < n_problems; q++) max_edges = std::max(max_edges, problems[q].n_edges);
    hipLaunchKernelGGL(k_ba_mad_errors_win, dim3((max_edges + BA_THREADS - 1) / BA_THREADS, n_problems), dim3(BA_THREADS), 0, ctx->stream, Q.d_probs);
    hipLaunchKernelGGL(k_ba_mad_select, dim3(n_problems), dim3(BA_MAD_THREADS), 0, ctx->stream, Q.d_probs, error_multiplier2);
    HSO_HIP_CHECK(ctx, hipGetLastError());
  }
  std::vector<BaLmHost> lm((size_t)n_problems);
  for (int q = 0; q < n_problems; q++) {
    const hso_ba_problem& P = problems[q];
    lm[(size_t)q] = {P.poses_f_w, P.idist, P.edge_chi2_out, P.result, P.n_iter};
  }
  return ba_run(ctx, Q, lm, obs_uv ? huber_out : nullptr, nullptr);
}

extern "C" int hso_gpu_ba_optimize_multi(hso_gpu_ctx* ctx, const hso_ba_problem* problems, int n_problems)
{
  return ba_optimize_multi_impl(ctx, problems, n_problems, nullptr, 0.0, nullptr);
}

extern "C" int hso_gpu_ba_local_multi(hso_gpu_ctx* ctx, const hso_ba_problem* problems, const double* const* obs_uv, int n_problems,
                                      double error_multiplier2, float* huber_out)
{
  if (!ctx) return HSO_E_INVALID;
  if (n_problems > 0 && !obs_uv) return hso_fail(ctx, HSO_E_INVALID, "ba_local_multi: bad argument");
  return ba_optimize_multi_impl(ctx, problems, n_problems, obs_uv, error_multiplier2, huber_out);
}

extern "C" int hso_gpu_ba_optimize(hso_gpu_ctx* ctx, hso_se3* poses_f_w, const uint8_t* pose_fixed, int n_poses, double* idist,
                                   int n_points, const hso_ba_edge* edges, int n_edges, double huber_corner, double huber_edge,
                                   int n_iter, double* edge_chi2_out, hso_ba_result* result)
{
  hso_ba_problem P;
  P.poses_f_w = poses_f_w; P.pose_fixed = pose_fixed; P.n_poses = n_poses; P.idist = idist; P.n_points = n_points;
  P.edges = edges; P.n_edges = n_edges; P.huber_corner = huber_corner; P.huber_edge = huber_edge; P.n_iter = n_iter;
  P.edge_chi2_out = edge_chi2_out; P.result = result;
  return hso_gpu_ba_optimize_multi(ctx, &P, 1);
}

// ================================================================================================================================
// ba::LocalBundleAdjustment on a sequence map (hso_gpu_seq_local_ba; src/bundle_adjustment.cpp:577-892): the window is assembled
// from the map's resident tables by the kernels below, optimised by the kernels and the driver above, and written back here.
//   stage A (one wait): k_rba_mark -> k_rba_compact -> k_rba_count -> k_rba_sizes: the window's points (ascending rows), the edges per
//            point, the vertex of every keyframe (first appearance in the reference's walk), the three sizes;
//   stage B: k_rba_fill (state, edges, the by-point table), k_rba_pairs<false> -> k_rba_pair_scan -> k_rba_pairs<true> (the edges of
//            every pose-pair block whose two poses are free, ascending — a stable counting sort; blocks with a fixed pose are never
//            read, k_ba_poses leaves before it looks at their list);
//   stage C (with the loop's last round): k_rba_writeback (idist_, pos_ into the point rows and out), k_rba_cull (:855-892's inputs).
#define RBA_MAX_POSES 512        // vertices of a window the sizing read-back names (the dense Hcc table of 512 poses is 75 MB)
#define RBA_SIZES (4 + RBA_MAX_POSES)
#define RBA_NO_KEY 0xffffffffffffffffull
#define RBA_CULL_FIRST 2046      // culled observations that ride in the final read-back; a longer list costs one more copy
#define RBA_MAX_BINS (HSO_SEQ_BA_MAX_CORE * (HSO_SEQ_BA_MAX_CORE + 1) / 2)

struct RbaJobDev {
  hso_map_point* pts; const hso_obs* obs; const int32_t* obs_pt; const int32_t* kf_fts;
  int n_pts, n_obs, n_kfs, fts_cap;
  int n_core, nb;                                // nb: bound of the window's points (rows of ids / eoff / idist0 / state)
  int core[HSO_SEQ_BA_MAX_CORE], core_len[HSO_SEQ_BA_MAX_CORE];
  int32_t* mark;                                 // [n_pts] 1: a feature of a core keyframe observes the point
  int32_t* ids;                                  // [nb] the window's points, ascending
  int32_t* eoff;                                 // [nb + 1] edges per point, then their exclusive prefix
  unsigned long long* fkey;                      // [n_kfs] (window point << 32 | position in its walk) of a keyframe's first appearance
  int32_t* vtx;                                  // [n_kfs] vertex of a keyframe row, -1: not in the window
  int32_t* sizes;                                // [RBA_SIZES] n_points, n_edges, n_poses, overflow flag, then the vertices' rows
  double* idist0;                                // [nb] idist_ before the optimisation (hso_gpu_seq_ba_debug_window)
  double* state;                                 // [4 * nb] idist, pos after it
};
struct RbaWinDev {
  RbaJobDev J;
  double* idist; hso_ba_edge* edges; double* uv; int* off; int* list; int* poff; int* plist; int* eobs; int* pcnt; int* cull;
  const int* col; const hso_se3* poses; const double* chi2;
  int n_points, n_edges, n_poses, n_pairs, n_free, n_bins, n_chunks, pad_;
  double chi2_corner, chi2_edgelet;
};

// exclusive prefix of v over the workgroup (W wavefronts), total = the sum; s_wave: W ints
template <int W> __device__ inline int rba_scan(int v, int* s_wave, int& total)
{
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int x = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) { const int y = __shfl_up(x, d); if (lane >= d) x += y; }
  if (lane == 63) s_wave[wave] = x;
  __syncthreads();
  int base = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < W; w++) { const int c = s_wave[w]; if (w < wave) base += c; tot += c; }
  total = tot;
  __syncthreads();
  return base + x - v;
}

// :592-616: the core keyframes are vertices 0 .. n_core-1; every point one of their features observes is in the window
__global__ __launch_bounds__(256) void k_rba_mark(const RbaJobDev* jobs)
{
  const RbaJobDev& J = jobs[blockIdx.y];
  const int c = blockIdx.z;
  if (c >= J.n_core) return;
  const int r = J.core[c];
  if (blockIdx.x == 0 && threadIdx.x == 0) J.vtx[r] = c;
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= J.core_len[c]) return;
  const int f = J.kf_fts[(size_t)r * J.fts_cap + t];
  if (f < 0 || f >= J.n_obs) return;
  const int p = J.obs_pt[f];
  if (p >= 0 && p < J.n_pts) J.mark[p] = 1;
}

// the window's points in ascending row order (one workgroup per job)
__global__ __launch_bounds__(1024) void k_rba_compact(const RbaJobDev* jobs)
{
  __shared__ int s_wave[16];
  const RbaJobDev& J = jobs[blockIdx.x];
  int n = 0;
  for (int base = 0; base < J.n_pts; base += 1024) {
    const int p = base + (int)threadIdx.x;
    const int m = (p < J.n_pts && J.mark[p]) ? 1 : 0;
    int tot;
    const int pos = n + rba_scan<16>(m, s_wave, tot);
    if (m && pos < J.nb) J.ids[pos] = p;
    n += tot;
  }
  if (threadIdx.x == 0) J.sizes[0] = n < J.nb ? n : J.nb;
}

// :690-812, first pass: per window point its edges (observations in keyframes other than the host's) and, for every keyframe outside
// the core that the walk meets, where it meets it first
__global__ __launch_bounds__(256) void k_rba_count(const RbaJobDev* jobs)
{
  const RbaJobDev& J = jobs[blockIdx.y];
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= J.sizes[0]) return;
  const hso_map_point& P = J.pts[J.ids[i]];
  const int hk = P.host_kf;
  if (hk >= 0 && hk < J.n_kfs && J.vtx[hk] < 0) atomicMin(&J.fkey[hk], (unsigned long long)i << 32);
  int cnt = 0, row = P.obs_begin;
  for (int q = 0; q < P.obs_count; q++) {
    if (row < 0 || row >= J.n_obs) break;
    const int kf = J.obs[row].kf, next = J.obs[row].pad_;
    if (kf != hk && kf >= 0 && kf < J.n_kfs) {
      cnt++;
      if (J.vtx[kf] < 0) atomicMin(&J.fkey[kf], ((unsigned long long)i << 32) | (unsigned)(q + 1));
    }
    row = next;
  }
  J.eoff[i] = cnt;
}

// the edges' offsets per point, the vertices of the keyframes outside the core in order of first appearance, the sizes
__global__ __launch_bounds__(1024) void k_rba_sizes(const RbaJobDev* jobs)
{
  __shared__ int s_wave[16];
  __shared__ unsigned long long s_key[HSO_SEQ_MAX_KFS];
  __shared__ int s_extra;
  const RbaJobDev& J = jobs[blockIdx.x];
  const int np = J.sizes[0], tid = threadIdx.x;
  int n = 0;
  for (int base = 0; base < np; base += 1024) {
    const int i = base + tid;
    const int v = i < np ? J.eoff[i] : 0;
    int tot;
    const int ex = rba_scan<16>(v, s_wave, tot);
    if (i < np) J.eoff[i] = n + ex;
    n += tot;
  }
  if (tid == 0) { J.eoff[np] = n; J.sizes[1] = n; s_extra = 0; }
  for (int r = tid; r < J.n_kfs; r += 1024) s_key[r] = J.fkey[r];
  __syncthreads();
  for (int r = tid; r < J.n_kfs; r += 1024) {
    const unsigned long long k = s_key[r];
    if (k == RBA_NO_KEY) continue;
    int rank = 0;
    for (int o = 0; o < J.n_kfs; o++) rank += s_key[o] < k ? 1 : 0;
    const int v = J.n_core + rank;
    J.vtx[r] = v;
    if (v < RBA_MAX_POSES) J.sizes[4 + v] = r;
    atomicAdd(&s_extra, 1);
  }
  if (tid < J.n_core) J.sizes[4 + tid] = J.core[tid];
  __syncthreads();
  if (tid == 0) { const int nv = J.n_core + s_extra; J.sizes[2] = nv; J.sizes[3] = nv > RBA_MAX_POSES ? 1 : 0; }
}

// :690-812, second pass: the window's state and edges (the reference's setHostBearing / setMeasurement / setTargetNormal /
// information, bundle_adjustment.cpp:740-800), the by-point table (an edge list in point order IS grouped by point)
__global__ __launch_bounds__(256) void k_rba_fill(const RbaWinDev* wins)
{
  const RbaWinDev& W = wins[blockIdx.y];
  const RbaJobDev& J = W.J;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= W.n_points) return;
  const hso_map_point& P = J.pts[J.ids[i]];
  W.idist[i] = P.idist; J.idist0[i] = P.idist;
  int k = J.eoff[i];
  W.off[i] = k;
  if (i == 0) W.off[W.n_points] = W.n_edges;
  const int hk = P.host_kf;
  const int vh = (hk >= 0 && hk < J.n_kfs) ? J.vtx[hk] : 0;
  int row = P.obs_begin;
  for (int q = 0; q < P.obs_count; q++) {
    if (row < 0 || row >= J.n_obs) break;
    const hso_obs ob = J.obs[row];
    if (ob.kf != hk && ob.kf >= 0 && ob.kf < J.n_kfs) {
      hso_ba_edge e;
      e.point = i; e.host = vh; e.target = J.vtx[ob.kf];
      e.type = ob.type == HSO_FTR_EDGELET ? HSO_FTR_EDGELET : HSO_FTR_CORNER;
      e.level = ob.level; e._pad = 0;
      e.fH[0] = P.host_f[0]; e.fH[1] = P.host_f[1]; e.fH[2] = P.host_f[2];
      const double u = ob.f[0] / ob.f[2], v = ob.f[1] / ob.f[2];
      if (e.type == HSO_FTR_EDGELET) { e.normal[0] = ob.grad[0]; e.normal[1] = ob.grad[1]; e.meas[0] = ob.grad[0] * u + ob.grad[1] * v; e.meas[1] = 0.0; }
      else { e.normal[0] = 1.0; e.normal[1] = 0.0; e.meas[0] = u; e.meas[1] = v; }
      W.edges[k] = e; W.uv[2 * k] = u; W.uv[2 * k + 1] = v; W.eobs[k] = row; W.list[k] = k;
      k++;
    }
    row = ob.pad_;
  }
}

// The edges of every pose-pair block, ascending (the order the value-passing call's host-built table has, so that the sums of
// k_ba_poses are formed in the same order): a stable counting sort over the blocks of two FREE poses — at most 16 of them, 136 blocks.
// An edge belongs to (h, h), (t, t) and (min, max).  A workgroup takes RBA_CHUNK edges as 16 slices of 64; inside a slice a lane's
// rank in a block's list = the number of lower lanes in the same block (one ballot per distinct block of the slice).
// SCATTER = false: the chunk's count per block.  SCATTER = true: the same ranks again, now behind the chunk's and the slice's base.
__device__ inline int rba_bin(int a, int b, int nf) { return a * nf - a * (a - 1) / 2 + (b - a); }   // a <= b
template <bool SCATTER>
__global__ __launch_bounds__(256) void k_rba_pairs(const RbaWinDev* wins)
{
  __shared__ unsigned short s_cnt[16][RBA_MAX_BINS];
  __shared__ int s_base[RBA_MAX_BINS];
  const RbaWinDev& W = wins[blockIdx.y];
  const int chunk = blockIdx.x;
  if (chunk >= W.n_chunks || W.n_bins == 0) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nf = W.n_free, nb = W.n_bins;
  for (int q = tid; q < 16 * RBA_MAX_BINS; q += 256) (&s_cnt[0][0])[q] = 0;
  __syncthreads();
  int key[4][3], rnk[4][3];
#pragma unroll
  for (int r = 0; r < 4; r++) {
    const int sl = r * 4 + wave, e = chunk * RBA_CHUNK + sl * 64 + lane;
    int k0 = -1, k1 = -1, k2 = -1;
    if (e < W.n_edges) {
      const int ch = W.col[W.edges[e].host], ct = W.col[W.edges[e].target];
      const int fh = ch >= 0 ? ch / 6 : -1, ft = ct >= 0 ? ct / 6 : -1;
      if (fh >= 0) k0 = rba_bin(fh, fh, nf);
      if (ft >= 0) k1 = rba_bin(ft, ft, nf);
      if (fh >= 0 && ft >= 0) k2 = fh < ft ? rba_bin(fh, ft, nf) : rba_bin(ft, fh, nf);
    }
    key[r][0] = k0; key[r][1] = k1; key[r][2] = k2;
    int p0 = k0, p1 = k1, p2 = k2, r0 = 0, r1 = 0, r2 = 0;
    for (;;) {
      int mine = 0x7fffffff;
      if (p0 >= 0) mine = p0;
      if (p1 >= 0 && p1 < mine) mine = p1;
      if (p2 >= 0 && p2 < mine) mine = p2;
      const unsigned long long any = __ballot(mine != 0x7fffffff);
      if (!any) break;
      const int B = __shfl(mine, __ffsll((long long)any) - 1);
      const bool hit = p0 == B || p1 == B || p2 == B;
      const unsigned long long m = __ballot(hit);
      if (hit) {
        const int rank = __popcll(m & ((1ull << lane) - 1ull));
        if (p0 == B) { r0 = rank; p0 = -1; } else if (p1 == B) { r1 = rank; p1 = -1; } else { r2 = rank; p2 = -1; }
      }
      if (lane == 0) s_cnt[sl][B] = (unsigned short)__popcll(m);
    }
    rnk[r][0] = r0; rnk[r][1] = r1; rnk[r][2] = r2;
  }
  __syncthreads();
  if (!SCATTER) {
    for (int b = tid; b < nb; b += 256) { int t = 0; for (int sl = 0; sl < 16; sl++) t += s_cnt[sl][b]; W.pcnt[(size_t)chunk * nb + b] = t; }
    return;
  }
  for (int b = tid; b < nb; b += 256) {
    int run = 0;
    for (int sl = 0; sl < 16; sl++) { const int c = s_cnt[sl][b]; s_cnt[sl][b] = (unsigned short)run; run += c; }
    s_base[b] = W.pcnt[(size_t)W.n_chunks * nb + b] + W.pcnt[(size_t)chunk * nb + b];
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < 4; r++) {
    const int sl = r * 4 + wave, e = chunk * RBA_CHUNK + sl * 64 + lane;
#pragma unroll
    for (int kk = 0; kk < 3; kk++) { const int b = key[r][kk]; if (b >= 0) W.plist[s_base[b] + s_cnt[sl][b] + rnk[r][kk]] = e; }
  }
}

// per block: the chunks' counts become offsets inside the block's list, the blocks' totals the lists' starts (row n_chunks of the
// table); then the offset of EVERY pose pair of the window (k_ba_poses reads two of them before it asks whether a pose is fixed):
// a pair with a fixed pose owns no entries
__global__ __launch_bounds__(256) void k_rba_pair_scan(const RbaWinDev* wins)
{
  __shared__ int s_tot[RBA_MAX_BINS + 1];
  __shared__ int s_cf[HSO_SEQ_BA_MAX_CORE + 1];
  const RbaWinDev& W = wins[blockIdx.x];
  const int tid = threadIdx.x, nb = W.n_bins, nf = W.n_free, np = W.n_poses;
  for (int b = tid; b < nb; b += 256) {
    int run = 0;
    for (int c = 0; c < W.n_chunks; c++) { const int v = W.pcnt[(size_t)c * nb + b]; W.pcnt[(size_t)c * nb + b] = run; run += v; }
    s_tot[b] = run;
  }
  __syncthreads();
  if (tid == 0) {
    int run = 0;
    for (int b = 0; b < nb; b++) { const int v = s_tot[b]; s_tot[b] = run; run += v; }
    s_tot[nb] = run;
    int cf = 0;   // free poses below vertex v (they are all among the core: v <= n_core)
    for (int v = 0; v <= HSO_SEQ_BA_MAX_CORE; v++) { s_cf[v] = cf; if (v < np && v < HSO_SEQ_BA_MAX_CORE && W.col[v] >= 0) cf++; }
  }
  __syncthreads();
  for (int b = tid; b < nb; b += 256) W.pcnt[(size_t)W.n_chunks * nb + b] = s_tot[b];
  // pair (i, j), i <= j, has id i * np - i (i - 1) / 2 + (j - i); the free-free blocks before it: those of the free rows below i, and
  // in row i (when i is free) the free columns in [i, j)
  for (int i = 0; i < np; i++) {
    const int ci = s_cf[i < HSO_SEQ_BA_MAX_CORE ? i : HSO_SEQ_BA_MAX_CORE];
    const bool fi = i < HSO_SEQ_BA_MAX_CORE && W.col[i] >= 0;
    const int row0 = ci * nf - ci * (ci - 1) / 2;
    const int pid0 = i * np - i * (i - 1) / 2;
    for (int j = i + tid; j < np; j += 256) {
      const int cj = s_cf[j < HSO_SEQ_BA_MAX_CORE ? j : HSO_SEQ_BA_MAX_CORE];
      W.poff[pid0 + (j - i)] = s_tot[row0 + (fi ? cj - ci : 0)];
    }
  }
  if (tid == 0) W.poff[W.n_pairs] = s_tot[nb];
}

// :826-853: idist_ and pos_ = T_host^-1 * (host_f * (1 / idist)) of the window's points, into the map's rows and out
__global__ __launch_bounds__(256) void k_rba_writeback(const RbaWinDev* wins, const BaLmDev* lm)
{
  if (lm[blockIdx.y].st != BA_ST_DONE || lm[blockIdx.y].written) return;   // not finished yet / written by an earlier block of rounds
  const RbaWinDev& W = wins[blockIdx.y];
  const RbaJobDev& J = W.J;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= W.n_points) return;
  hso_map_point& P = J.pts[J.ids[i]];
  const double idv = W.idist[i];
  const int hk = P.host_kf;
  const int vh = (hk >= 0 && hk < J.n_kfs) ? J.vtx[hk] : 0;
  const Se3 inv = se3_inverse(se3_from(W.poses[vh]));
  const double s = 1.0 / idv;
  double x, y, z;
  se3_apply(inv, P.host_f[0] * s, P.host_f[1] * s, P.host_f[2] * s, x, y, z);
  P.idist = idv; P.pos[0] = x; P.pos[1] = y; P.pos[2] = z;
  double* o = J.state + 4 * (size_t)i;
  o[0] = idv; o[1] = x; o[2] = y; o[3] = z;
}

// :855-892: the observations of the edges above the threshold, corner edges first, in edge order (one workgroup per window)
__global__ __launch_bounds__(1024) void k_rba_cull(const RbaWinDev* wins, BaLmDev* lm)
{
  __shared__ int s_wave[16];
  if (lm[blockIdx.x].st != BA_ST_DONE || lm[blockIdx.x].written) return;
  const RbaWinDev& W = wins[blockIdx.x];
  int n = 0;
  for (int pass = 0; pass < 2; pass++) {
    const int n0 = n;
    for (int base = 0; base < W.n_edges; base += 1024) {
      const int e = base + (int)threadIdx.x;
      int m = 0;
      if (e < W.n_edges) {
        const bool edgelet = W.edges[e].type == HSO_FTR_EDGELET;
        m = (edgelet == (pass == 1) && W.chi2[e] > (edgelet ? W.chi2_edgelet : W.chi2_corner)) ? 1 : 0;
      }
      int tot;
      const int pos = n + rba_scan<16>(m, s_wave, tot);
      if (m) W.cull[2 + pos] = W.eobs[e];
      n += tot;
    }
    if (threadIdx.x == 0) W.cull[pass] = n - n0;
  }
  if (threadIdx.x == 0) lm[blockIdx.x].written = 1;               // (k_rba_writeback ran before this kernel)
}

// what the last call of a context left for hso_gpu_seq_ba_debug_window
struct RbaLast {
  struct Win {
    int status = 1, n_poses = 0, n_points = 0, n_edges = 0;
    std::vector<int32_t> rows; std::vector<uint8_t> fixed; std::vector<hso_se3> poses_in;
    const hso_ba_edge* d_edges = nullptr; const double* d_uv = nullptr; const int* d_eobs = nullptr; const double* d_chi2 = nullptr;
    const hso_se3* d_poses = nullptr; const double* d_idist0 = nullptr;
  };
  std::vector<Win> win;
};

void hso_rba_forget(hso_gpu_ctx* ctx)
{
  if (ctx->d_rba) (void)hipFree(ctx->d_rba);
  ctx->d_rba = nullptr; ctx->rba_cap = 0;
  delete ctx->rba_last;
  ctx->rba_last = nullptr;
}

extern "C" int hso_gpu_seq_local_ba(hso_gpu_ctx* ctx, const hso_seq_ba_job* jobs, int n_jobs, double error_multiplier2, double chi2_corner,
                                    double chi2_edgelet, hso_seq_ba_result* results)
{
  if (!ctx) return HSO_E_INVALID;
  if (n_jobs < 0 || (n_jobs > 0 && (!jobs || !results))) return hso_fail(ctx, HSO_E_INVALID, "seq_local_ba: bad argument");
  if (ctx->rba_last) ctx->rba_last->win.clear();
  if (n_jobs == 0) return HSO_OK;
  HSO_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  auto al = [](size_t b) { return (b + 255) & ~size_t(255); };
  // ---- the jobs against their maps; the per-job tables of stage A
  std::vector<SeqMapDev> view((size_t)n_jobs);
  std::vector<const hso_kf*> kfs_host((size_t)n_jobs);
  std::vector<RbaJobDev> hj((size_t)n_jobs);
  size_t zero_bytes = 0, ones_bytes = 0, rest_bytes = 0;
  int max_len = 1, max_core = 1, max_nb = 1;
  for (int j = 0; j < n_jobs; j++) for (int i = 0; i < j; i++) if (jobs[i].map == jobs[j].map) return hso_fail(ctx, HSO_E_INVALID, "seq_local_ba: a map appears twice");
  for (int j = 0; j < n_jobs; j++) {
    const hso_seq_ba_job& A = jobs[j];
    if (A.n_core < 1 || A.n_core > HSO_SEQ_BA_MAX_CORE || A.n_iter < 0 || A.point_cap < 0 || A.cull_cap < 0 || (A.point_cap > 0 && (!A.point_ids || !A.point_state)) ||
        (A.cull_cap > 0 && !A.culled))
      return hso_fail(ctx, HSO_E_INVALID, "seq_local_ba: bad argument");
    const int32_t* nfts = nullptr;
    if (int rc = hso_seqmap_ba_view(ctx, A.map, &view[(size_t)j], &kfs_host[(size_t)j], &nfts)) return rc;
    const SeqMapDev& M = view[(size_t)j];
    RbaJobDev& J = hj[(size_t)j];
    memset(&J, 0, sizeof(J));
    J.pts = M.pts; J.obs = M.obs; J.obs_pt = M.obs_pt; J.kf_fts = M.kf_fts;
    J.n_pts = M.n_pts; J.n_obs = M.n_obs; J.n_kfs = M.n_kfs; J.fts_cap = M.fts_cap; J.n_core = A.n_core;
    size_t len = 0;
    for (int c = 0; c < A.n_core; c++) {
      if (A.core[c] < 0 || A.core[c] >= M.n_kfs) return hso_fail(ctx, HSO_E_INVALID, "seq_local_ba: no such keyframe row");
      for (int k = 0; k < c; k++) if (A.core[k] == A.core[c]) return hso_fail(ctx, HSO_E_INVALID, "seq_local_ba: a core keyframe appears twice");
      J.core[c] = A.core[c]; J.core_len[c] = nfts[A.core[c]];
      len += (size_t)J.core_len[c]; max_len = std::max(max_len, J.core_len[c]);
    }
    J.nb = (int)std::min<size_t>(len, (size_t)M.n_pts);
    max_core = std::max(max_core, A.n_core); max_nb = std::max(max_nb, J.nb);
    zero_bytes += al(sizeof(int32_t) * (size_t)M.n_pts);
    ones_bytes += al(sizeof(unsigned long long) * (size_t)M.n_kfs) + al(sizeof(int32_t) * (size_t)M.n_kfs);
    rest_bytes += al(sizeof(int32_t) * (size_t)J.nb) + al(sizeof(int32_t) * ((size_t)J.nb + 1)) + al(sizeof(double) * (size_t)J.nb) + al(sizeof(double) * 4 * (size_t)J.nb);
  }
  const size_t sz_bytes = sizeof(int32_t) * RBA_SIZES * (size_t)n_jobs;   // the jobs' sizes side by side: one copy brings them
  const size_t o_jobs = 0, o_sizes = al(sizeof(RbaJobDev) * (size_t)n_jobs), o_zero = o_sizes + al(sz_bytes), o_ones = o_zero + zero_bytes, o_rest = o_ones + ones_bytes,
               need = o_rest + rest_bytes;
  if (ctx->rba_cap < need) {
    HSO_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->d_rba) (void)hipFree(ctx->d_rba);
    ctx->d_rba = nullptr; ctx->rba_cap = 0;
    HSO_HIP_CHECK(ctx, hipMalloc(reinterpret_cast<void**>(&ctx->d_rba), hso_grown(need)));
    ctx->rba_cap = hso_grown(need);
  }
  char* dr = ctx->d_rba;
  {
    size_t z = o_zero, o = o_ones, r = o_rest;
    for (int j = 0; j < n_jobs; j++) {
      RbaJobDev& J = hj[(size_t)j];
      J.mark = reinterpret_cast<int32_t*>(dr + z); z += al(sizeof(int32_t) * (size_t)J.n_pts);
      J.fkey = reinterpret_cast<unsigned long long*>(dr + o); o += al(sizeof(unsigned long long) * (size_t)J.n_kfs);
      J.vtx = reinterpret_cast<int32_t*>(dr + o); o += al(sizeof(int32_t) * (size_t)J.n_kfs);
      J.ids = reinterpret_cast<int32_t*>(dr + r); r += al(sizeof(int32_t) * (size_t)J.nb);
      J.eoff = reinterpret_cast<int32_t*>(dr + r); r += al(sizeof(int32_t) * ((size_t)J.nb + 1));
      J.sizes = reinterpret_cast<int32_t*>(dr + o_sizes) + RBA_SIZES * (size_t)j;
      J.idist0 = reinterpret_cast<double*>(dr + r); r += al(sizeof(double) * (size_t)J.nb);
      J.state = reinterpret_cast<double*>(dr + r); r += al(sizeof(double) * 4 * (size_t)J.nb);
    }
  }
  char* hp0 = hso_pinned(ctx, 0, sizeof(RbaJobDev) * (size_t)n_jobs);
  char* hp1 = hso_pinned(ctx, 1, std::max<size_t>(sz_bytes, 256));
  if (!hp0 || !hp1) return HSO_E_NOMEM;
  memcpy(hp0, hj.data(), sizeof(RbaJobDev) * (size_t)n_jobs);
  HSO_HIP_CHECK(ctx, hipMemcpyAsync(dr + o_jobs, hp0, sizeof(RbaJobDev) * (size_t)n_jobs, hipMemcpyHostToDevice, ctx->stream));
  if (zero_bytes) HSO_HIP_CHECK(ctx, hipMemsetAsync(dr + o_zero, 0, zero_bytes, ctx->stream));
  HSO_HIP_CHECK(ctx, hipMemsetAsync(dr + o_ones, 0xff, ones_bytes, ctx->stream));
  const RbaJobDev* d_jobs = reinterpret_cast<const RbaJobDev*>(dr + o_jobs);
  hipLaunchKernelGGL(k_rba_mark, dim3((max_len + 255) / 256, n_jobs, max_core), dim3(256), 0, ctx->stream, d_jobs);
  hipLaunchKernelGGL(k_rba_compact, dim3(n_jobs), dim3(1024), 0, ctx->stream, d_jobs);
  hipLaunchKernelGGL(k_rba_count, dim3((max_nb + 255) / 256, n_jobs), dim3(256), 0, ctx->stream, d_jobs);
  hipLaunchKernelGGL(k_rba_sizes, dim3(n_jobs), dim3(1024), 0, ctx->stream, d_jobs);
  HSO_HIP_CHECK(ctx, hipGetLastError());
  HSO_HIP_CHECK(ctx, hipMemcpyAsync(hp1, dr + o_sizes, sz_bytes, hipMemcpyDeviceToHost, ctx->stream));
  HSO_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  std::vector<int32_t> sizes(reinterpret_cast<int32_t*>(hp1), reinterpret_cast<int32_t*>(hp1) + RBA_SIZES * (size_t)n_jobs);   // (the pinned blocks serve the batch next)

  // ---- the windows that have something to optimise
  if (!ctx->rba_last) ctx->rba_last = new RbaLast();
  RbaLast& last = *ctx->rba_last;
  last.win.assign((size_t)n_jobs, RbaLast::Win());
  std::vector<int> act;                          // window q of the batch = job act[q]
  for (int j = 0; j < n_jobs; j++) {
    const int32_t* sz = &sizes[RBA_SIZES * (size_t)j];
    hso_seq_ba_result& R = results[j];
    memset(&R, 0, sizeof(R));
    R.n_points = sz[0]; R.n_edges = sz[1]; R.n_poses = sz[2];
    if (sz[3]) return hso_fail(ctx, HSO_E_UNSUPPORTED, "seq_local_ba: more than 512 keyframes in one window");
    if (R.n_points > jobs[j].point_cap) return hso_fail(ctx, HSO_E_INVALID, "seq_local_ba: point_cap is smaller than the window");
    RbaLast::Win& L = last.win[(size_t)j];
    L.n_poses = R.n_poses; L.n_points = R.n_points; L.n_edges = R.n_edges;
    L.rows.assign(sz + 4, sz + 4 + R.n_poses);
    L.fixed.assign((size_t)R.n_poses, 1);
    for (int c = 0; c < jobs[j].n_core; c++) L.fixed[(size_t)c] = jobs[j].fixed[c] ? 1 : 0;
    L.poses_in.resize((size_t)R.n_poses);
    for (int v = 0; v < R.n_poses; v++) L.poses_in[(size_t)v] = kfs_host[(size_t)j][L.rows[(size_t)v]].T_f_w;
    for (int c = 0; c < jobs[j].n_core; c++) R.core_pose[c] = L.poses_in[(size_t)c];
    R.status = (R.n_edges > 0 && R.n_points > 0) ? 0 : 1;
    L.status = R.status;
    if (R.status == 0) act.push_back(j);
  }
  const int nw = (int)act.size();
  std::vector<HsoListCopy> back;
  if (nw == 0) {
    for (int j = 0; j < n_jobs; j++) back.push_back({jobs[j].point_ids, hj[(size_t)j].ids, sizeof(int32_t) * (size_t)results[j].n_points});
    return hso_lists_to_host(ctx, back);
  }
  BaBatch Q;
  Q.win.resize((size_t)nw);
  for (int q = 0; q < nw; q++) {
    const RbaLast::Win& L = last.win[(size_t)act[q]];
    ba_layout(Q.win[(size_t)q], L.n_poses, L.n_points, L.fixed.data(), L.n_edges, 0.0, 0.0, true, true);
    if (Q.win[(size_t)q].M > 96) return hso_fail(ctx, HSO_E_INVALID, "seq_local_ba: more than 16 free poses in one window");
  }
  BaResident res;
  res.extra_hdr = sizeof(RbaWinDev) * (size_t)nw;
  res.head = [&](int q, char* w) {
    const BaWin& B = Q.win[(size_t)q];
    const RbaLast::Win& L = last.win[(size_t)act[q]];
    memset(w + B.o_trial, 0, B.o_poses - B.o_trial);
    memcpy(w + B.o_poses, L.poses_in.data(), sizeof(hso_se3) * (size_t)B.n_poses);
    memcpy(w + B.o_fixed, L.fixed.data(), (size_t)B.n_poses);
    memcpy(w + B.o_col, B.col.data(), sizeof(int) * (size_t)B.n_poses);
  };
  if (int rc = ba_batch_begin(Q, ctx, nullptr, nw, nullptr, &res)) return rc;
  // (ba_batch_begin has queued the header's copy; the records behind the header were not written yet: they go up now)
  RbaWinDev* hw = reinterpret_cast<RbaWinDev*>(res.h_extra);
  int max_np = 1, max_chunks = 1;
  for (int q = 0; q < nw; q++) {
    const BaWin& B = Q.win[(size_t)q];
    RbaWinDev& W = hw[q];
    memset(&W, 0, sizeof(W));
    W.J = hj[(size_t)act[q]];
    W.idist = reinterpret_cast<double*>(B.at(B.o_idist)); W.edges = reinterpret_cast<hso_ba_edge*>(B.at(B.o_edges)); W.uv = reinterpret_cast<double*>(B.at(B.o_uv));
    W.off = reinterpret_cast<int*>(B.at(B.o_off)); W.list = reinterpret_cast<int*>(B.at(B.o_list)); W.poff = reinterpret_cast<int*>(B.at(B.o_poff));
    W.plist = reinterpret_cast<int*>(B.at(B.o_plist)); W.eobs = reinterpret_cast<int*>(B.at(B.o_eobs)); W.pcnt = reinterpret_cast<int*>(B.at(B.o_pcnt));
    W.cull = reinterpret_cast<int*>(B.at(B.o_cull)); W.col = reinterpret_cast<const int*>(B.at(B.o_col)); W.poses = reinterpret_cast<const hso_se3*>(B.at(B.o_poses));
    W.chi2 = reinterpret_cast<const double*>(B.at(B.o_chi));
    W.n_points = B.n_points; W.n_edges = B.n_edges; W.n_poses = B.n_poses; W.n_pairs = B.n_pairs; W.n_free = B.n_free; W.n_bins = B.n_bins; W.n_chunks = B.n_chunks;
    W.chi2_corner = chi2_corner; W.chi2_edgelet = chi2_edgelet;
    max_np = std::max(max_np, B.n_points); max_chunks = std::max(max_chunks, B.n_chunks);
    RbaLast::Win& L = last.win[(size_t)act[q]];
    L.d_edges = W.edges; L.d_uv = W.uv; L.d_eobs = W.eobs; L.d_chi2 = W.chi2; L.d_poses = W.poses; L.d_idist0 = W.J.idist0;
  }
  HSO_HIP_CHECK(ctx, hipMemcpyAsync(res.d_extra, res.h_extra, sizeof(RbaWinDev) * (size_t)nw, hipMemcpyHostToDevice, ctx->stream));
  const RbaWinDev* d_wins = reinterpret_cast<const RbaWinDev*>(res.d_extra);
  hipLaunchKernelGGL(k_rba_fill, dim3((max_np + 255) / 256, nw), dim3(256), 0, ctx->stream, d_wins);
  hipLaunchKernelGGL(k_rba_pairs<false>, dim3(max_chunks, nw), dim3(256), 0, ctx->stream, d_wins);
  hipLaunchKernelGGL(k_rba_pair_scan, dim3(nw), dim3(256), 0, ctx->stream, d_wins);
  hipLaunchKernelGGL(k_rba_pairs<true>, dim3(max_chunks, nw), dim3(256), 0, ctx->stream, d_wins);
  // :618-680: the Huber deltas of the windows' initial state
  {
    int max_edges = 0;
    for (int q = 0; q < nw; q++) max_edges = std::max(max_edges, Q.win[(size_t)q].n_edges);
    hipLaunchKernelGGL(k_ba_mad_errors_win, dim3((max_edges + BA_THREADS - 1) / BA_THREADS, nw), dim3(BA_THREADS), 0, ctx->stream, Q.d_probs);
    hipLaunchKernelGGL(k_ba_mad_select, dim3(nw), dim3(BA_MAD_THREADS), 0, ctx->stream, Q.d_probs, error_multiplier2);
    HSO_HIP_CHECK(ctx, hipGetLastError());
  }
  std::vector<BaLmHost> lm((size_t)nw);
  std::vector<float> hub(2 * (size_t)nw);
  std::vector<std::vector<int32_t>> cull((size_t)nw);
  std::vector<std::vector<hso_se3>> poses_out((size_t)nw);
  for (int q = 0; q < nw; q++) {
    const BaWin& B = Q.win[(size_t)q];
    lm[(size_t)q] = {nullptr, nullptr, nullptr, &results[act[q]].lm, jobs[act[q]].n_iter};
    cull[(size_t)q].assign(2 + (size_t)std::min(B.n_edges, RBA_CULL_FIRST), 0);
    poses_out[(size_t)q].resize((size_t)jobs[act[q]].n_core);
  }
  // behind every block of rounds: the write-back of the windows that are done (a window is written once), and everything the caller
  // mirrors — final for the finished windows; the loop ends when all are
  const BaFinalHook finalise = [&](std::vector<HsoListCopy>& out) -> int {
    hipLaunchKernelGGL(k_rba_writeback, dim3((max_np + 255) / 256, nw), dim3(256), 0, ctx->stream, d_wins, Q.d_lm);
    hipLaunchKernelGGL(k_rba_cull, dim3(nw), dim3(1024), 0, ctx->stream, d_wins, Q.d_lm);
    HSO_HIP_CHECK(ctx, hipGetLastError());
    for (int q = 0; q < nw; q++) {
      const BaWin& B = Q.win[(size_t)q];
      const hso_seq_ba_job& A = jobs[act[q]];
      const RbaJobDev& J = hj[(size_t)act[q]];
      out.push_back({A.point_ids, J.ids, sizeof(int32_t) * (size_t)B.n_points});
      out.push_back({A.point_state, J.state, sizeof(double) * 4 * (size_t)B.n_points});
      out.push_back({poses_out[(size_t)q].data(), B.at(B.o_poses), sizeof(hso_se3) * (size_t)A.n_core});
      out.push_back({cull[(size_t)q].data(), B.at(B.o_cull), sizeof(int32_t) * cull[(size_t)q].size()});
    }
    return HSO_OK;
  };
  if (int rc = ba_run(ctx, Q, lm, hub.data(), &finalise)) return rc;
  // ---- what the caller mirrors
  for (int q = 0; q < nw; q++) {
    const int j = act[q];
    const hso_seq_ba_job& A = jobs[j];
    hso_seq_ba_result& R = results[j];
    const BaWin& B = Q.win[(size_t)q];
    R.huber_corner = hub[2 * (size_t)q]; R.huber_edge = hub[2 * (size_t)q + 1];
    for (int c = 0; c < A.n_core; c++) { R.core_pose[c] = poses_out[(size_t)q][(size_t)c]; hso_seqmap_ba_set_pose(ctx, A.map, A.core[c], R.core_pose[c]); }
    R.n_culled[0] = cull[(size_t)q][0]; R.n_culled[1] = cull[(size_t)q][1];
    const int nc = R.n_culled[0] + R.n_culled[1], first = (int)cull[(size_t)q].size() - 2;
    for (int e = 0; e < nc && e < first && e < A.cull_cap; e++) A.culled[e] = cull[(size_t)q][2 + (size_t)e];
    if (nc > first && A.cull_cap > first) {      // a list longer than what rode in the final read-back: the rest in its own copy
      const int more = std::min(nc, A.cull_cap) - first;
      HSO_HIP_CHECK(ctx, hso_copy_sync(A.culled + first, reinterpret_cast<const int32_t*>(B.at(B.o_cull)) + 2 + first, sizeof(int32_t) * (size_t)more, hipMemcpyDeviceToHost));
    }
  }
  // the windows that were left alone still name their points
  back.clear();
  for (int j = 0; j < n_jobs; j++) if (results[j].status != 0) back.push_back({jobs[j].point_ids, hj[(size_t)j].ids, sizeof(int32_t) * (size_t)results[j].n_points});
  if (!back.empty()) return hso_lists_to_host(ctx, back);
  return HSO_OK;
}

extern "C" int hso_gpu_seq_ba_debug_window(hso_gpu_ctx* ctx, int job, int what, void* out, size_t bytes)
{
  if (!ctx) return HSO_E_INVALID;
  if (!ctx->rba_last || job < 0 || (size_t)job >= ctx->rba_last->win.size() || !out) return hso_fail(ctx, HSO_E_INVALID, "seq_ba_debug_window: bad argument");
  const RbaLast::Win& L = ctx->rba_last->win[(size_t)job];
  const int32_t sizes[4] = {L.n_poses, L.n_points, L.n_edges, L.status};
  const void* host = nullptr; const void* dev = nullptr; size_t have = 0;
  switch (what) {
    case HSO_BAW_SIZES: host = sizes; have = sizeof(sizes); break;
    case HSO_BAW_VERTEX_ROWS: host = L.rows.data(); have = sizeof(int32_t) * L.rows.size(); break;
    case HSO_BAW_FIXED: host = L.fixed.data(); have = L.fixed.size(); break;
    case HSO_BAW_POSES_IN: host = L.poses_in.data(); have = sizeof(hso_se3) * L.poses_in.size(); break;
    case HSO_BAW_EDGES: dev = L.d_edges; have = sizeof(hso_ba_edge) * (size_t)L.n_edges; break;
    case HSO_BAW_OBS_UV: dev = L.d_uv; have = sizeof(double) * 2 * (size_t)L.n_edges; break;
    case HSO_BAW_EDGE_OBS: dev = L.d_eobs; have = sizeof(int32_t) * (size_t)L.n_edges; break;
    case HSO_BAW_EDGE_CHI2: dev = L.d_chi2; have = sizeof(double) * (size_t)L.n_edges; break;
    case HSO_BAW_POSES_OUT: dev = L.d_poses; have = sizeof(hso_se3) * (size_t)L.n_poses; break;
    case HSO_BAW_IDIST_IN: dev = L.d_idist0; have = sizeof(double) * (size_t)L.n_points; break;
    default: return hso_fail(ctx, HSO_E_INVALID, "seq_ba_debug_window: no such table");
  }
  if (L.status != 0 && !host) have = 0;          // a window that was left alone has no device tables
  if (have != bytes) return hso_fail(ctx, HSO_E_INVALID, "seq_ba_debug_window: bytes differs from the table's size");
  HSO_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  if (bytes == 0) return HSO_OK;
  if (host) { memcpy(out, host, bytes); return HSO_OK; }
  HSO_HIP_CHECK(ctx, hso_copy_sync(out, dev, bytes, hipMemcpyDeviceToHost));
  return HSO_OK;
}
