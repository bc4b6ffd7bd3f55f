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
#include "hso_ctx.h"
#include "hso_dev_math.h"
#include <string.h>
#include <algorithm>
#include <cmath>
#include <vector>

using namespace hso_dev;

#define BA_THREADS 256
#define BA_WAVES (BA_THREADS / 64)
#define BA_LIN 32  // doubles per edge in the linearisation record

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
__global__ __launch_bounds__(BA_THREADS) void k_ba_edges(const BaProb* probs, const int* active)
{
  const BaArgs a = probs[active[blockIdx.y]].a;
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

__global__ __launch_bounds__(BA_THREADS) void k_ba_points(const BaProb* probs, const int* active)
{
  const BaProb& P = probs[active[blockIdx.y]];
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
__global__ __launch_bounds__(BA_THREADS) void k_ba_poses(const BaProb* probs, const int* active)
{
  __shared__ double s_part[BA_WAVES][44];
  const BaProb& P = probs[active[blockIdx.y]];
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
__global__ __launch_bounds__(BA_THREADS) void k_ba_zero(const BaProb* probs, const int* active)
{
  const BaProb& P = probs[active[blockIdx.y]];
  uint4* z = reinterpret_cast<uint4*>(P.zero_begin);
  const size_t n = P.zero_bytes / 16;
  for (size_t i = (size_t)blockIdx.x * BA_THREADS + threadIdx.x; i < n; i += (size_t)gridDim.x * BA_THREADS) z[i] = make_uint4(0, 0, 0, 0);
}

// computeLambdaInit (thirdparty/g2o/g2o/core/optimization_algorithm_levenberg.cpp:191-201): the largest |diagonal entry| of the
// Hessian blocks of every free vertex — points: Hpp, free poses: the diagonal of their Hcc block — into sum[5].  A maximum does
// not depend on the order it is taken in; the host used to read Hpp and the whole Hcc table of every window for this one number.
__global__ __launch_bounds__(BA_THREADS) void k_ba_maxdiag(const BaProb* probs, const int* active)
{
  __shared__ double s_part[BA_WAVES];
  const BaProb& P = probs[active[blockIdx.y]];
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

// sum of chi2 and of the robustified rho(chi2) over all edges (activeChi2 / activeRobustChi2,
// thirdparty/g2o/g2o/core/sparse_optimizer.cpp:100-113): one workgroup, fixed tree => deterministic
__global__ __launch_bounds__(BA_THREADS) void k_ba_chi2(const BaProb* probs, const int* active)
{
  __shared__ double s_part[BA_WAVES][2];
  const BaProb& P = probs[active[blockIdx.y]];
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
}

// The reduced system of one LM trial on the device: S = Hcc_free + lambda I - sum_p Hpc_p^T (Hpp_p + lambda)^-1 Hpc_p and
// rhs = bc - sum_p Hpc_p^T bp_p / (Hpp_p + lambda) (the points are 1-D, so the Schur complement is scalar).  One block per
// pose-pair block (i <= j) of the window: threads stride over the points (rows of Hpc of unconnected poses are zero), then
// the fixed tree of k_ba_poses; the block writes its 6x6 piece to both triangles of S.  Only M * M + M doubles go back to the
// host for the dense factorisation — not the n_points x n_poses x 6 block table.
__global__ __launch_bounds__(BA_THREADS) void k_ba_schur(const BaProb* probs, const int* active)
{
  __shared__ double s_part[BA_WAVES][44];
  const BaProb& P = probs[active[blockIdx.y]];
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
__global__ __launch_bounds__(BA_THREADS) void k_ba_solve(const BaProb* probs, const int* active)
{
  extern __shared__ double s_S[];
  __shared__ double s_x[96], s_col[96], s_sc[16];   // s_sc: one entry per FREE pose (M <= 96 unknowns = 16 poses)
  __shared__ int s_ok;
  const BaProb& P = probs[active[blockIdx.y]];
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
// equals the sparse loop's); idist_bak keeps the state before the step (g2o's push()).  One block per window.
__global__ __launch_bounds__(BA_THREADS) void k_ba_backsub(const BaProb* probs, const int* active)
{
  __shared__ double s_part[BA_WAVES];
  const BaProb& P = probs[active[blockIdx.y]];
  const int np = P.a.n_poses;
  const double lambda = ba_lambda(P);
  const double* xc = P.trial + 1;
  const bool no_step = P.trial[1 + 6 * np] != 0.0;   // the reduced system could not be solved: x = 0 (the trial is rejected)
  double sc = 0;
  for (int p = threadIdx.x; p < P.a.n_points; p += BA_THREADS) {
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
  if (threadIdx.x == 0) { double t = 0; for (int w = 0; w < BA_WAVES; w++) t += s_part[w]; P.sum[2] = t; }
}

// g2o's pop() for the points: the state before the rejected step
__global__ __launch_bounds__(BA_THREADS) void k_ba_restore(const BaProb* probs, const int* active)
{
  const BaProb& P = probs[active[blockIdx.y]];
  for (int p = threadIdx.x; p < P.a.n_points; p += BA_THREADS) P.idist_rw[p] = P.idist_bak[p];
  for (int i = threadIdx.x; i < P.a.n_poses; i += BA_THREADS) P.poses_rw[i] = P.poses_bak[i];
}

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
  size_t o_trial, o_poses, o_idist, o_fixed, o_edges, o_off, o_list, o_poff, o_plist, o_col, o_uv, in_bytes;
  size_t o_lin, o_rho, o_out, o_Hpp, o_bp, o_Hpc, o_Hcc, o_bc, o_err, o_chi, o_sum, o_S, o_rhs, o_xp, o_bak, o_pbak, total;
  // the device slice comes in two pieces: the upload images of all windows of a batch lie side by side (one copy brings them all),
  // the work areas behind them; an offset below in_bytes is in the first piece, any other in the second
  char* d;              // the window's upload image on the device (offsets < in_bytes)
  char* dw;             // its work area MINUS in_bytes (so that dw + o_x is the address of a work table)
  char* h_in;           // pinned: the window's upload image
  char* at(size_t off) const { return off < in_bytes ? d + off : dw + off; }
};

struct BaBatch {
  hso_gpu_ctx* ctx;
  std::vector<BaWin> win;
  BaProb* d_probs = nullptr;
  int* d_active = nullptr;       // BA_N_LISTS lists of n windows each
  int* h_active = nullptr;       // pinned
  double* d_lambda = nullptr;    // [n] the damping of each window's current trial
  double* h_lambda = nullptr;    // pinned
  double* d_sums = nullptr;      // [n][8] chi2, robust chi2, scale (points), scale (poses), solvable
  double* h_sums = nullptr;      // pinned (results staging)
  float* d_hub = nullptr;        // [n][2] the Huber deltas formed on the device (hso_gpu_ba_local_multi)
  float* h_hub = nullptr;        // pinned
  int n = 0;
};
#define BA_N_LISTS 6

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
static void ba_layout(BaWin& B, int n_poses, int n_points, const uint8_t* pose_fixed, int n_edges,
                      double huber_corner, double huber_edge, bool with_uv = false)
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

  auto al = [](size_t b) { return (b + 255) & ~size_t(255); };
  size_t o = 0;
  // [lambda | xc | pad | poses | idist]: the trial block first, the state right behind it
  B.o_trial = o; o += al(sizeof(double) * (2 + 6 * (size_t)n_poses));   // lambda, xc, "no step" flag
  B.o_poses = o; o += al(sizeof(hso_se3) * n_poses);
  B.o_idist = o; o += al(sizeof(double) * n_points);
  B.o_fixed = o; o += al(n_poses);
  B.o_edges = o; o += al(sizeof(hso_ba_edge) * n_edges);
  B.o_off = o; o += al(sizeof(int) * (n_points + 1));
  B.o_list = o; o += al(sizeof(int) * n_edges);
  B.o_poff = o; o += al(sizeof(int) * (n_pairs + 1));
  B.o_plist = o; o += al(sizeof(int) * 3 * (size_t)n_edges);
  B.o_col = o; o += al(sizeof(int) * n_poses);
  B.o_uv = o; if (with_uv) o += al(sizeof(double) * 2 * (size_t)n_edges);
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
static int ba_batch_begin(BaBatch& Q, hso_gpu_ctx* ctx, const hso_ba_problem* problems, int n, const double* const* obs_uv = nullptr)
{
  Q.ctx = ctx; Q.n = n;
  HSO_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  auto al = [](size_t b) { return (b + 255) & ~size_t(255); };
  const size_t o_act = al(sizeof(BaProb) * (size_t)n), o_lam = o_act + al(sizeof(int) * (size_t)n * BA_N_LISTS);
  const size_t o_sums = o_lam + al(sizeof(double) * (size_t)n);
  const size_t o_hub = o_sums + al(sizeof(double) * 8 * (size_t)n);
  size_t dev = o_hub + al(sizeof(float) * 2 * (size_t)n), pin_in = dev, pin_out = al(sizeof(double) * 8 * (size_t)n) + al(sizeof(float) * 2 * (size_t)n);
  const size_t hdr = dev;
  for (int q = 0; q < n; q++) { dev += Q.win[q].total; pin_in += Q.win[q].in_bytes; }
  if (ctx->batch_cap < dev) {  // grow-only work area of the context (shared with the other batched entry points)
    HSO_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->d_batch) (void)hipFree(ctx->d_batch);
    ctx->d_batch = nullptr; ctx->batch_cap = 0;
    HSO_HIP_CHECK(ctx, hipMalloc(reinterpret_cast<void**>(&ctx->d_batch), hso_grown(dev)));
    ctx->batch_cap = hso_grown(dev);
  }
  char* d = reinterpret_cast<char*>(ctx->d_batch);
  char* h = hso_pinned(ctx, 0, pin_in);
  char* ho = hso_pinned(ctx, 1, std::max<size_t>(pin_out, 256));
  if (!h || !ho) return HSO_E_NOMEM;
  Q.d_probs = reinterpret_cast<BaProb*>(d);
  Q.d_active = reinterpret_cast<int*>(d + o_act);
  Q.d_lambda = reinterpret_cast<double*>(d + o_lam);
  Q.d_sums = reinterpret_cast<double*>(d + o_sums);
  BaProb* hp = reinterpret_cast<BaProb*>(h);
  Q.h_active = reinterpret_cast<int*>(h + o_act);
  Q.h_lambda = reinterpret_cast<double*>(h + o_lam);
  Q.h_sums = reinterpret_cast<double*>(ho);
  Q.d_hub = reinterpret_cast<float*>(d + o_hub);
  Q.h_hub = reinterpret_cast<float*>(ho + al(sizeof(double) * 8 * (size_t)n));
  size_t ow = pin_in, oh = hdr;   // [header | upload images | work areas]: the pinned block mirrors the first two
  for (int q = 0; q < n; q++) {
    BaWin& B = Q.win[q];
    B.d = d + oh; B.h_in = h + oh; B.dw = d + ow - B.in_bytes;
    ow += B.total - B.in_bytes; oh += B.in_bytes;
  }
  // the windows' input images, assembled side by side in the page-locked block (tens of megabytes per keyframe step): ONE pass over
  // a window's edges checks their indices, copies them and counts both adjacency tables, a second fills the tables — in place
  std::vector<uint8_t> bad((size_t)n, 0);
  hso_host_parallel(ctx, n, pin_in - hdr, [&](int q) {
    if (!ba_stage_window(Q.win[q], problems[q], obs_uv ? obs_uv[q] : nullptr)) bad[(size_t)q] = 1;
  });
  for (int q = 0; q < n; q++)
    if (bad[(size_t)q]) return ba_check_edges(ctx, problems[q].edges, problems[q].n_edges, problems[q].n_points, problems[q].n_poses, "ba_optimize");
  for (int q = 0; q < n; q++) {
    BaWin& B = Q.win[q];
    BaProb& R = hp[q];
    char* dd = B.d;
    R.a.poses = reinterpret_cast<const hso_se3*>(dd + B.o_poses); R.a.fixed = reinterpret_cast<const uint8_t*>(dd + B.o_fixed);
    R.a.idist = reinterpret_cast<const double*>(dd + B.o_idist); R.a.edges = reinterpret_cast<const hso_ba_edge*>(dd + B.o_edges);
    R.a.n_poses = B.n_poses; R.a.n_points = B.n_points; R.a.n_edges = B.n_edges;
    R.a.huber_corner = B.huber_corner; R.a.huber_edge = B.huber_edge;
    R.a.lin = reinterpret_cast<double*>(B.at(B.o_lin)); R.a.edge_err = reinterpret_cast<double*>(B.at(B.o_err));
    R.a.edge_chi2 = reinterpret_cast<double*>(B.at(B.o_chi)); R.a.edge_rho = reinterpret_cast<double*>(B.at(B.o_rho));
    R.off = reinterpret_cast<const int*>(dd + B.o_off); R.list = reinterpret_cast<const int*>(dd + B.o_list);
    R.poff = reinterpret_cast<const int*>(dd + B.o_poff); R.plist = reinterpret_cast<const int*>(dd + B.o_plist);
    R.Hpp = reinterpret_cast<double*>(B.at(B.o_Hpp)); R.bp = reinterpret_cast<double*>(B.at(B.o_bp));
    R.Hpc = reinterpret_cast<double*>(B.at(B.o_Hpc)); R.Hcc = reinterpret_cast<double*>(B.at(B.o_Hcc));
    R.bc = reinterpret_cast<double*>(B.at(B.o_bc)); R.sum = Q.d_sums + 8 * (size_t)q; R.lam = Q.d_lambda + q;
    R.col = reinterpret_cast<const int*>(dd + B.o_col);
    R.S = reinterpret_cast<double*>(B.at(B.o_S)); R.rhs = reinterpret_cast<double*>(B.at(B.o_rhs));
    R.trial = reinterpret_cast<const double*>(dd + B.o_trial);
    R.xp = reinterpret_cast<double*>(B.at(B.o_xp));
    R.idist_rw = reinterpret_cast<double*>(dd + B.o_idist); R.idist_bak = reinterpret_cast<double*>(B.at(B.o_bak));
    R.trial_rw = reinterpret_cast<double*>(dd + B.o_trial);
    R.poses_rw = reinterpret_cast<hso_se3*>(dd + B.o_poses); R.poses_bak = reinterpret_cast<hso_se3*>(B.at(B.o_pbak));
    R.M = B.M; R.n_pairs = B.n_pairs;
    R.zero_begin = B.at(B.o_out); R.zero_bytes = B.o_sum + 256 - B.o_out;
    R.uv = obs_uv ? reinterpret_cast<const double*>(dd + B.o_uv) : nullptr;
    R.mad = reinterpret_cast<float*>(B.at(B.o_err)); R.hub = Q.d_hub + 2 * (size_t)q;
  }
  // the window records, the (not yet filled) launch lists and every window's upload image in ONE copy (it was one per window)
  HSO_HIP_CHECK(ctx, hipMemcpyAsync(d, h, pin_in, hipMemcpyHostToDevice, ctx->stream));
  return HSO_OK;
}

// Launch helpers: `which` = the windows (indices) this launch works on, written to list slot `slot` (a slot may be reused
// only after a synchronise, which every round of the driver ends with).
static int ba_list(BaBatch& Q, int slot, const std::vector<int>& which, const int** d_list)
{
  int* hl = Q.h_active + (size_t)slot * Q.n;
  for (size_t k = 0; k < which.size(); k++) hl[k] = which[k];
  int* dl = Q.d_active + (size_t)slot * Q.n;
  HSO_HIP_CHECK(Q.ctx, hipMemcpyAsync(dl, hl, sizeof(int) * which.size(), hipMemcpyHostToDevice, Q.ctx->stream));
  *d_list = dl;
  return HSO_OK;
}
static int ba_max(const BaBatch& Q, const std::vector<int>& which, int BaWin::*field)
{
  int m = 0;
  for (int q : which) m = std::max(m, Q.win[q].*field);
  return m;
}
// computeActiveErrors + buildSystem at the resident state; the blocks stay on the device
static int ba_launch_linearize(BaBatch& Q, int slot, const std::vector<int>& which, const int* dl = nullptr)
{
  hso_gpu_ctx* ctx = Q.ctx;
  if (which.empty()) return HSO_OK;
  if (!dl) if (int rc = ba_list(Q, slot, which, &dl)) return rc;   // (dl given: the round's lists went up in one copy)
  const int ny = (int)which.size();
  hipLaunchKernelGGL(k_ba_zero, dim3(64, ny), dim3(BA_THREADS), 0, ctx->stream, Q.d_probs, dl);
  hipLaunchKernelGGL(k_ba_edges<true>, dim3((ba_max(Q, which, &BaWin::n_edges) + BA_THREADS - 1) / BA_THREADS, ny), dim3(BA_THREADS), 0, ctx->stream, Q.d_probs, dl);
  hipLaunchKernelGGL(k_ba_points, dim3((ba_max(Q, which, &BaWin::n_points) + BA_THREADS - 1) / BA_THREADS, ny), dim3(BA_THREADS), 0, ctx->stream, Q.d_probs, dl);
  hipLaunchKernelGGL(k_ba_poses, dim3(ba_max(Q, which, &BaWin::n_pairs) + 1, ny), dim3(BA_THREADS), 0, ctx->stream, Q.d_probs, dl);
  hipLaunchKernelGGL(k_ba_maxdiag, dim3(1, ny), dim3(BA_THREADS), 0, ctx->stream, Q.d_probs, dl);
  HSO_HIP_CHECK(ctx, hipGetLastError());
  return HSO_OK;
}
// computeActiveErrors only (an LM trial): per-edge error / chi2 / rho and the two sums
static int ba_launch_errors(BaBatch& Q, int slot, const std::vector<int>& which, const int* dl = nullptr)
{
  hso_gpu_ctx* ctx = Q.ctx;
  if (which.empty()) return HSO_OK;
  if (!dl) if (int rc = ba_list(Q, slot, which, &dl)) return rc;
  const int ny = (int)which.size();
  hipLaunchKernelGGL(k_ba_edges<false>, dim3((ba_max(Q, which, &BaWin::n_edges) + BA_THREADS - 1) / BA_THREADS, ny), dim3(BA_THREADS), 0, ctx->stream, Q.d_probs, dl);
  hipLaunchKernelGGL(k_ba_chi2, dim3(1, ny), dim3(BA_THREADS), 0, ctx->stream, Q.d_probs, dl);
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
  const std::vector<int> all(1, 0);
  if (int rc = ba_launch_linearize(Q, 0, all)) return rc;
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
__global__ __launch_bounds__(BA_THREADS) void k_ba_mad_select(BaProb* probs, double error_multiplier2)
{
  BaProb& P = probs[blockIdx.x];
  const int n = P.a.n_edges, tid = threadIdx.x;
  const hso_ba_edge* E = P.a.edges; const float* err = P.mad;
  __shared__ unsigned s_hist[256];
  __shared__ unsigned s_prefix, s_rank;
  __shared__ int s_cnt[2];
  if (tid < 2) s_cnt[tid] = 0;
  __syncthreads();
  { int c[2] = {0, 0}; for (int k = tid; k < n; k += BA_THREADS) c[E[k].type == HSO_FTR_EDGELET ? 1 : 0]++; if (c[0]) atomicAdd(&s_cnt[0], c[0]); if (c[1]) atomicAdd(&s_cnt[1], c[1]); }
  __syncthreads();
  float med[2] = {0.f, 0.f};
  for (int kind = 0; kind < 2; kind++) {
    const int cnt = s_cnt[kind];
    if (cnt == 0) continue;            // uniform: every thread reads the same shared count
    if (tid == 0) { s_prefix = 0u; s_rank = (unsigned)(cnt / 2); }
    unsigned mask = 0u;
    for (int shift = 24; shift >= 0; shift -= 8) {
      s_hist[tid & 255] = 0u;          // BA_THREADS == 256
      __syncthreads();
      const unsigned prefix = s_prefix;
      for (int k = tid; k < n; k += BA_THREADS) {
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

// One local-BA window being optimised: the Levenberg loop of OptimizationAlgorithmLevenberg::solve written as a state
// machine, so that many windows advance in lockstep.  A round of the driver asks every unfinished window what it needs
// next (want), launches each kind of device work ONCE for all windows that want it (blockIdx.y = window), synchronises once,
// and lets every window consume its results (advance).  An LM trial is ONE device sequence — reduced system (k_ba_schur),
// dense LDL^T + pose update (k_ba_solve), back-substitution + point update (k_ba_backsub), error evaluation
// (k_ba_edges<false>, k_ba_chi2) — and five doubles come back; the host only takes the accept / reject decision.
struct BaLm {
  enum Want { W_ERRORS, W_LINEARIZE, W_TRIAL, W_RESTORE_THEN_TRIAL, W_FINAL, W_NONE };
  enum State { INIT_WAIT, LIN_WAIT, STEP_WAIT, FINAL_WAIT, DONE };
  BaWin* B;
  hso_se3* poses_f_w; const uint8_t* pose_fixed; double* idist;
  int n_poses, n_points, n_edges, n_iter;
  double* edge_chi2_out; hso_ba_result* result;
  double lambda, ni, currentChi, tempChi, iniChi, rho;
  int nBad, stop, it, qmax;
  bool need_restore, first_lin;
  State st; Want want;

  double* sums;       // pinned: chi2, robust chi2, scale (points), scale (poses), solvable — of the last device step
  double* lam_stage;  // pinned: the damping the next trial uses
  double* out_sum() const { return sums; }

  void begin()   // runSparseBAOptimizer: computeActiveErrors(); init_error = activeChi2()
  {
    memset(result, 0, sizeof(*result));
    lambda = -1.; ni = 2.; nBad = 0; stop = 0; it = 0; qmax = 0; rho = 0; currentChi = tempChi = iniChi = 0; need_restore = false;
    // The first linearisation evaluates the errors at the very state init_error is taken at, and sums them the same way
    // (k_ba_poses' extra block = k_ba_chi2's reduction): with at least one iteration to run, the errors-only round is left out
    // and init_chi2 comes from the first linearisation's sums (one round = one synchronisation less per call)
    first_lin = n_iter > 0;
    if (first_lin) { st = LIN_WAIT; want = W_LINEARIZE; *lam_stage = -1.0; } else { st = INIT_WAIT; want = W_ERRORS; }
  }
  void finish() { result->stop = stop; result->lambda = lambda; st = FINAL_WAIT; want = W_FINAL; }

  // consume the results of the device work asked for by `want`; decide what is needed next
  void advance()
  {
    switch (st) {
      case INIT_WAIT:
        result->init_chi2 = out_sum()[0];
        result->robust_chi2 = out_sum()[1];
        result->final_chi2 = out_sum()[0];
        if (it >= n_iter) { finish(); return; }
        *lam_stage = -1.0;
        st = LIN_WAIT; want = W_LINEARIZE;
        return;
      case LIN_WAIT:
        // A linearisation and the first trial behind it ran in ONE round (the damping of the trial is known before: carried over,
        // or formed on the device from the linearisation's largest diagonal entry).  The linearisation's sums are in [6], [7], [5].
        if (first_lin) { result->init_chi2 = out_sum()[6]; result->robust_chi2 = out_sum()[7]; first_lin = false; }
        // solve(): computeActiveErrors, currentChi = activeRobustChi2, buildSystem
        result->final_chi2 = out_sum()[6];
        currentChi = out_sum()[7]; tempChi = currentChi;
        iniChi = currentChi;
        if (it == 0) {   // computeLambdaInit: tau (1e-5) * the largest diagonal entry over all free vertices (k_ba_maxdiag)
          lambda = 1e-5 * out_sum()[5];
          ni = 2; nBad = 0;
        }
        rho = 0; qmax = 0;
        st = STEP_WAIT;
        [[fallthrough]];   // the trial's results are in this round's sums too
      case STEP_WAIT: {
        const bool ok2 = out_sum()[4] != 0.0;
        result->n_solves++;
        result->final_chi2 = out_sum()[0];   // activeChi2() of the last computeActiveErrors
        tempChi = ok2 ? out_sum()[1] : 1.7976931348623157e308;
        rho = currentChi - tempChi;
        double scale = out_sum()[2] + out_sum()[3];                      // computeScale (zero when the solve failed: no step)
        scale += 1e-3;
        rho /= scale;
        bool restore = false;
        if (rho > 0 && std::isfinite(tempChi)) {
          double alpha = 1. - std::pow((2 * rho - 1), 3);
          alpha = std::min(alpha, 2. / 3.);
          lambda *= std::max(1. / 3., alpha);
          ni = 2;
          currentChi = tempChi;
          result->n_accepted++;
        } else {
          lambda *= ni;
          ni *= 2;
          restore = true;                                                // _optimizer->pop(): vertices only, edge errors stay
        }
        qmax++;
        if (rho < 0 && qmax < 5) {   // setMaxTrialsAfterFailure(5), src/bundle_adjustment.cpp:571
          *lam_stage = lambda;
          st = STEP_WAIT; want = restore ? W_RESTORE_THEN_TRIAL : W_TRIAL;
          return;
        }
        need_restore = restore;
        result->iterations = it + 1;
        result->robust_chi2 = currentChi;
        if (qmax == 5 || rho == 0) { stop = 1; finish(); return; }
        if ((iniChi - currentChi) * 1e3 < iniChi) nBad++; else nBad = 0;   // optimization_algorithm_levenberg.cpp:154-161
        if (nBad >= 3) { stop = 2; finish(); return; }
        it++;
        if (it >= n_iter) { finish(); return; }
        *lam_stage = lambda;                 // the damping of the trial that rides behind the next linearisation
        st = LIN_WAIT; want = W_LINEARIZE;   // (need_restore: only after a step with a NaN gain ratio; the driver restores first)
        return;
      }
      case FINAL_WAIT:
        st = DONE; want = W_NONE;
        return;
      case DONE:
        return;
    }
  }
};

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
  std::vector<BaLm> lm(n_problems);
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
  bool hub_pending = false;
  if (obs_uv) {
    int max_edges = 0;
    for (int q = 0; q < n_problems; q++) max_edges = std::max(max_edges, problems[q].n_edges);
    hipLaunchKernelGGL(k_ba_mad_errors_win, dim3((max_edges + BA_THREADS - 1) / BA_THREADS, n_problems), dim3(BA_THREADS), 0, ctx->stream, Q.d_probs);
    hipLaunchKernelGGL(k_ba_mad_select, dim3(n_problems), dim3(BA_THREADS), 0, ctx->stream, Q.d_probs, error_multiplier2);
    HSO_HIP_CHECK(ctx, hipGetLastError());
    HSO_HIP_CHECK(ctx, hipMemcpyAsync(Q.h_hub, Q.d_hub, sizeof(float) * 2 * (size_t)n_problems, hipMemcpyDeviceToHost, ctx->stream));
    hub_pending = true;   // read after the first round's synchronisation
  }
  for (int q = 0; q < n_problems; q++) {
    const hso_ba_problem& P = problems[q];
    BaLm& L = lm[q];
    L.B = &Q.win[q]; L.poses_f_w = P.poses_f_w; L.pose_fixed = P.pose_fixed; L.idist = P.idist;
    L.n_poses = P.n_poses; L.n_points = P.n_points; L.n_edges = P.n_edges; L.n_iter = P.n_iter;
    L.edge_chi2_out = P.edge_chi2_out; L.result = P.result;
    L.sums = Q.h_sums + 8 * (size_t)q; L.lam_stage = Q.h_lambda + q;
    L.begin();
  }
  std::vector<int> w_err, w_lin, w_trial, w_restore, w_final;
  for (;;) {
    w_err.clear(); w_lin.clear(); w_trial.clear(); w_restore.clear(); w_final.clear();
    for (int q = 0; q < n_problems; q++)
      switch (lm[q].want) {
        case BaLm::W_ERRORS: w_err.push_back(q); break;
        case BaLm::W_LINEARIZE: w_lin.push_back(q); w_trial.push_back(q); if (lm[q].need_restore) { w_restore.push_back(q); lm[q].need_restore = false; } break;
        case BaLm::W_RESTORE_THEN_TRIAL: w_restore.push_back(q); w_trial.push_back(q); break;
        case BaLm::W_TRIAL: w_trial.push_back(q); break;
        case BaLm::W_FINAL: w_final.push_back(q); if (lm[q].need_restore) w_restore.push_back(q); break;
        case BaLm::W_NONE: break;
      }
    if (w_err.empty() && w_lin.empty() && w_trial.empty() && w_final.empty()) break;
    // this round's launch lists and the damping of its trials: ONE copy (the lists and the damping lie side by side in the header
    // of the batch, in page-locked memory and on the device; poses and points are owned by the device during the optimisation)
    auto fill = [&](int slot, const std::vector<int>& which) -> const int* {
      int* hl = Q.h_active + (size_t)slot * Q.n;
      for (size_t k = 0; k < which.size(); k++) hl[k] = which[k];
      return Q.d_active + (size_t)slot * Q.n;
    };
    const int* dl_err = fill(0, w_err); const int* dl_restore = fill(1, w_restore); const int* dl_lin = fill(2, w_lin); const int* dl_trial = fill(3, w_trial);
    HSO_HIP_CHECK(ctx, hipMemcpyAsync(Q.d_active, Q.h_active, (size_t)(reinterpret_cast<char*>(Q.h_lambda + n_problems) - reinterpret_cast<char*>(Q.h_active)),
                                      hipMemcpyHostToDevice, ctx->stream));
    if (int rc = ba_launch_errors(Q, 0, w_err, dl_err)) return rc;
    if (!w_restore.empty())   // g2o's pop() after a rejected step, before anything reads the state again
      hipLaunchKernelGGL(k_ba_restore, dim3(1, (int)w_restore.size()), dim3(BA_THREADS), 0, ctx->stream, Q.d_probs, dl_restore);
    if (int rc = ba_launch_linearize(Q, 2, w_lin, dl_lin)) return rc;
    if (!w_trial.empty()) {
      const int* dl = dl_trial;
      const int ny = (int)w_trial.size(), max_m = ba_max(Q, w_trial, &BaWin::M);
      hipLaunchKernelGGL(k_ba_schur, dim3(ba_max(Q, w_trial, &BaWin::n_pairs) + 1, ny), dim3(BA_THREADS), 0, ctx->stream, Q.d_probs, dl);
      static bool solve_attr = false;   // 96 x 96 doubles = 72 KiB of dynamic LDS: above the 64 KiB a launch gets without asking
      if (!solve_attr) {
        HSO_HIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(k_ba_solve), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 96 * (int)sizeof(double)));
        solve_attr = true;
      }
      hipLaunchKernelGGL(k_ba_solve, dim3(1, ny), dim3(BA_THREADS), sizeof(double) * (size_t)std::max(max_m * max_m, 1), ctx->stream, Q.d_probs, dl);
      hipLaunchKernelGGL(k_ba_backsub, dim3(1, ny), dim3(BA_THREADS), 0, ctx->stream, Q.d_probs, dl);
      if (int rc = ba_launch_errors(Q, 4, w_trial, dl_trial)) return rc;
    }
    HSO_HIP_CHECK(ctx, hipGetLastError());
    // --- results
    // the sums of every window in one copy (windows that did nothing this round keep their old values, nobody reads them)
    HSO_HIP_CHECK(ctx, hipMemcpyAsync(Q.h_sums, Q.d_sums, sizeof(double) * 8 * (size_t)n_problems, hipMemcpyDeviceToHost, ctx->stream));
    // the windows that finished this round hand back their state: every table of every such window in one DMA
    std::vector<HsoListCopy> back;
    for (int q : w_final) {
      const BaWin& B = Q.win[q];
      back.push_back({lm[q].idist, B.d + B.o_idist, sizeof(double) * (size_t)B.n_points});
      back.push_back({lm[q].poses_f_w, B.d + B.o_poses, sizeof(hso_se3) * (size_t)B.n_poses});
      if (lm[q].edge_chi2_out) back.push_back({lm[q].edge_chi2_out, B.at(B.o_chi), sizeof(double) * (size_t)B.n_edges});
    }
    if (int rc = hso_lists_to_host(ctx, back)) return rc;   // synchronises (also when there is nothing to read back)
    if (hub_pending) { memcpy(huber_out, Q.h_hub, sizeof(float) * 2 * (size_t)n_problems); hub_pending = false; }
    for (int q = 0; q < n_problems; q++) if (lm[q].want != BaLm::W_NONE) lm[q].advance();
  }
  return HSO_OK;
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
