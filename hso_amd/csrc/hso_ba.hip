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

template <bool LIN>   // LIN = false: errors, chi2 and rho only (an LM trial's computeActiveErrors)
__global__ __launch_bounds__(BA_THREADS) void k_ba_edges(BaArgs a)
{
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
  for (int d = 0; d < dim; d++) chi2 += rec[d] * om * rec[d];
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

__global__ __launch_bounds__(BA_THREADS) void k_ba_points(BaArgs a, const int* pt_off, const int* pt_edges,
                                                          double* Hpp, double* bp, double* Hpc)
{
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= a.n_points) return;
  double hpp = 0, b = 0;
  for (int q = pt_off[p]; q < pt_off[p + 1]; q++) {
    const int k = pt_edges[q];
    const EdgeLin L = ba_load(a, k);
    const hso_ba_edge& e = a.edges[k];
    double s = 0, g = 0;
    double AtO[2];
    for (int d = 0; d < L.dim; d++) { AtO[d] = L.Jp[d] * L.omega; s += AtO[d] * L.Jp[d]; g += L.Jp[d] * ba_omega_r(a, k, L, d); }
    hpp += s; b += g;
    if (!a.fixed[e.host])
      for (int c = 0; c < 6; c++) { double v = 0; for (int d = 0; d < L.dim; d++) v += AtO[d] * L.Jh[d * 6 + c]; Hpc[((size_t)p * a.n_poses + e.host) * 6 + c] += v; }
    if (!a.fixed[e.target])
      for (int c = 0; c < 6; c++) { double v = 0; for (int d = 0; d < L.dim; d++) v += AtO[d] * L.Jt[d * 6 + c]; Hpc[((size_t)p * a.n_poses + e.target) * 6 + c] += v; }
  }
  Hpp[p] = hpp; bp[p] = b;
}

// pr_off / pr_edges: CSR of the edges that touch each block's pose pair (host-built, edge order
// kept): diagonal block i lists every edge with host == i or target == i, block (i, j) every edge
// whose two frames are {i, j} — a block reads only its own edges instead of scanning all of them.
__global__ __launch_bounds__(BA_THREADS) void k_ba_poses(BaArgs a, const int* pr_off, const int* pr_edges, double* Hcc, double* bc,
                                                         double* chi2_sum)
{
  __shared__ double s_part[BA_WAVES][44];
  // block -> (i, j), i <= j; one extra block sums the chi2 values
  const int np = a.n_poses;
  int b = blockIdx.x, i = 0;
  const int n_pairs = np * (np + 1) / 2;
  const bool chi_block = (b == n_pairs);
  const int q0 = chi_block ? 0 : pr_off[b], q1 = chi_block ? 0 : pr_off[b + 1];
  int j = 0;
  if (!chi_block) { while (b >= np - i) { b -= np - i; i++; } j = i + b; }
  double acc[44];
#pragma unroll
  for (int q = 0; q < 44; q++) acc[q] = 0;
  if (chi_block) {
    for (int k = threadIdx.x; k < a.n_edges; k += BA_THREADS) { acc[0] += a.edge_chi2[k]; acc[1] += a.edge_rho[k]; }
  } else if (!a.fixed[i] && !a.fixed[j]) {
    for (int q = q0 + (int)threadIdx.x; q < q1; q += BA_THREADS) {
      const int k = pr_edges[q];
      const hso_ba_edge& e = a.edges[k];
      const int h = e.host, t = e.target;
      if (i == j) {
        if (h != i && t != i) continue;
        const EdgeLin L = ba_load(a, k);
        const double* Jx = (h == i) ? L.Jh : L.Jt;
        for (int r = 0; r < 6; r++) {
          for (int d = 0; d < L.dim; d++) acc[36 + r] += Jx[d * 6 + r] * ba_omega_r(a, k, L, d);
          for (int c = 0; c < 6; c++) {
            double v = 0;
            for (int d = 0; d < L.dim; d++) v += (Jx[d * 6 + r] * L.omega) * Jx[d * 6 + c];
            acc[r * 6 + c] += v;
          }
        }
      } else {
        const bool fwd = (h == i && t == j), rev = (h == j && t == i);
        if (!fwd && !rev) continue;
        const EdgeLin L = ba_load(a, k);
        // block (host, target) = Jh^T Omega Jt; stored at (i,j) directly or transposed
        for (int r = 0; r < 6; r++)
          for (int c = 0; c < 6; c++) {
            double v = 0;
            for (int d = 0; d < L.dim; d++) v += (L.Jh[d * 6 + r] * L.omega) * L.Jt[d * 6 + c];
            if (fwd) acc[r * 6 + c] += v; else acc[c * 6 + r] += v;
          }
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
    if (chi_block) { if (threadIdx.x < 2) chi2_sum[threadIdx.x] = t; }
    else if (threadIdx.x < 36) Hcc[((size_t)i * np + j) * 36 + threadIdx.x] = t;
    else if (i == j && threadIdx.x < 42) bc[i * 6 + threadIdx.x - 36] = t;
  }
}


// sum of chi2 and of the robustified rho(chi2) over all edges (activeChi2 / activeRobustChi2,
// thirdparty/g2o/g2o/core/sparse_optimizer.cpp:100-113): one workgroup, fixed tree => deterministic
__global__ __launch_bounds__(BA_THREADS) void k_ba_chi2(BaArgs a, double* chi2_sum)
{
  __shared__ double s_part[BA_WAVES][2];
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

// per-edge error magnitudes for the Huber deltas of LocalBundleAdjustment (src/bundle_adjustment.cpp:618-656):
// e = (project2d(obs->f) - project2d(Tth * fH / idist)) / 2^level; corners |e|, edgelets |grad^T e| (floats)
__global__ __launch_bounds__(BA_THREADS) void k_ba_mad_errors(const hso_se3* poses, const double* idist, const hso_ba_edge* edges,
                                                              const double* obs_uv, int n_edges, float* err_out)
{
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n_edges) return;
  const hso_ba_edge e = edges[k];
  const Se3 Tth = se3_mul(se3_from(poses[e.target]), se3_inverse(se3_from(poses[e.host])));
  const double inv = 1.0 / idist[e.point];
  double x, y, z;
  se3_apply(Tth, e.fH[0] * inv, e.fH[1] * inv, e.fH[2] * inv, x, y, z);
  double ex = obs_uv[2 * k] - x / z, ey = obs_uv[2 * k + 1] - y / z;
  const double sc = 1.0 / (double)(1 << e.level);
  ex *= sc; ey *= sc;
  err_out[k] = (e.type == HSO_FTR_EDGELET) ? (float)fabs(e.normal[0] * ex + e.normal[1] * ey) : (float)sqrt(ex * ex + ey * ey);
}

// ------------------------------------------------------------------ host side

// One BA problem resident in the context's work area: inputs uploaded once, the state (poses, inverse depths)
// refreshed per evaluation, the blocks read back per linearisation.
struct BaDev {
  hso_gpu_ctx* ctx;
  int n_poses, n_points, n_edges, n_pairs;
  size_t o_poses, o_fixed, o_idist, o_edges, o_off, o_list, o_poff, o_plist, in_bytes, o_lin, o_rho, o_out, o_Hpp, o_bp, o_Hpc, o_Hcc,
      o_bc, o_err, o_chi, o_sum, total;
  char* d;
  char* h;
  BaArgs a;
  std::vector<int> off, list, poff, plist;   // CSR of edges by point / by pose-pair block (host copies, uploaded by ba_place)
  double huber_corner, huber_edge;
};

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

// sizes, offsets and the two CSR tables of one problem (no device work)
static void ba_layout(BaDev& B, hso_gpu_ctx* ctx, int n_poses, int n_points, const hso_ba_edge* edges, int n_edges,
                      double huber_corner, double huber_edge)
{
  B.ctx = ctx; B.n_poses = n_poses; B.n_points = n_points; B.n_edges = n_edges;
  B.huber_corner = huber_corner; B.huber_edge = huber_edge;
  // CSR of edges by point, edge order kept inside a point (g2o visits edges in insertion order)
  std::vector<int>& off = B.off; std::vector<int>& list = B.list;
  off.assign(n_points + 1, 0); list.assign(n_edges, 0);
  for (int k = 0; k < n_edges; k++) off[edges[k].point + 1]++;
  for (int p = 0; p < n_points; p++) off[p + 1] += off[p];
  { std::vector<int> cur(off.begin(), off.end() - 1); for (int k = 0; k < n_edges; k++) list[cur[edges[k].point]++] = k; }

  // CSR of edges by pose-pair block (same block numbering as k_ba_poses), edge order kept
  const int n_pairs = n_poses * (n_poses + 1) / 2;
  B.n_pairs = n_pairs;
  auto pair_id = [n_poses](int i, int j) { return i * n_poses - i * (i - 1) / 2 + (j - i); };  // i <= j
  std::vector<int>& poff = B.poff; std::vector<int>& plist = B.plist;
  poff.assign(n_pairs + 1, 0); plist.assign((size_t)3 * n_edges, 0);
  for (int k = 0; k < n_edges; k++) {
    const int h_ = edges[k].host, t_ = edges[k].target;
    poff[pair_id(h_, h_) + 1]++; poff[pair_id(t_, t_) + 1]++;
    poff[pair_id(h_ < t_ ? h_ : t_, h_ < t_ ? t_ : h_) + 1]++;
  }
  for (int q = 0; q < n_pairs; q++) poff[q + 1] += poff[q];
  {
    std::vector<int> cur(poff.begin(), poff.end() - 1);
    for (int k = 0; k < n_edges; k++) {
      const int h_ = edges[k].host, t_ = edges[k].target;
      plist[cur[pair_id(h_, h_)]++] = k; plist[cur[pair_id(t_, t_)]++] = k;
      plist[cur[pair_id(h_ < t_ ? h_ : t_, h_ < t_ ? t_ : h_)]++] = k;
    }
  }

  auto al = [](size_t b) { return (b + 255) & ~size_t(255); };
  size_t o = 0;
  B.o_poses = o; o += al(sizeof(hso_se3) * n_poses);
  B.o_idist = o; o += al(sizeof(double) * n_points);      // poses | idist: the state, contiguous
  B.o_fixed = o; o += al(n_poses);
  B.o_edges = o; o += al(sizeof(hso_ba_edge) * n_edges);
  B.o_off = o; o += al(sizeof(int) * (n_points + 1));
  B.o_list = o; o += al(sizeof(int) * n_edges);
  B.o_poff = o; o += al(sizeof(int) * (n_pairs + 1));
  B.o_plist = o; o += al(sizeof(int) * 3 * (size_t)n_edges);
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
  B.o_sum = o; o += 256;
  B.total = o;
}

// one grow-only device work area + pinned staging for a set of problems laid out one after the other
static int ba_reserve(hso_gpu_ctx* ctx, size_t dev_bytes, size_t pinned_bytes, char** d, char** h)
{
  HSO_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  if (ctx->batch_cap < dev_bytes) {  // grow-only work area of the context (shared with the other batched entry points)
    HSO_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->d_batch) (void)hipFree(ctx->d_batch);
    ctx->d_batch = nullptr; ctx->batch_cap = 0;
    HSO_HIP_CHECK(ctx, hipMalloc(reinterpret_cast<void**>(&ctx->d_batch), dev_bytes));
    ctx->batch_cap = dev_bytes;
  }
  *d = reinterpret_cast<char*>(ctx->d_batch);
  *h = hso_pinned(ctx, 0, pinned_bytes);
  return *h ? HSO_OK : HSO_E_NOMEM;
}

// put a laid-out problem at d / h and upload everything that does not change between evaluations
static int ba_place(BaDev& B, char* d, char* h, const hso_se3* poses_f_w, const uint8_t* pose_fixed, const double* idist,
                    const hso_ba_edge* edges)
{
  hso_gpu_ctx* ctx = B.ctx;
  const int n_poses = B.n_poses, n_points = B.n_points, n_edges = B.n_edges;
  B.d = d; B.h = h;
  memset(h, 0, B.in_bytes);
  memcpy(h + B.o_poses, poses_f_w, sizeof(hso_se3) * n_poses);
  memcpy(h + B.o_fixed, pose_fixed, n_poses);
  memcpy(h + B.o_idist, idist, sizeof(double) * n_points);
  memcpy(h + B.o_edges, edges, sizeof(hso_ba_edge) * n_edges);
  memcpy(h + B.o_off, B.off.data(), sizeof(int) * (n_points + 1));
  memcpy(h + B.o_list, B.list.data(), sizeof(int) * n_edges);
  memcpy(h + B.o_poff, B.poff.data(), sizeof(int) * (B.n_pairs + 1));
  memcpy(h + B.o_plist, B.plist.data(), sizeof(int) * 3 * (size_t)n_edges);
  HSO_HIP_CHECK(ctx, hipMemcpyAsync(B.d, h, B.in_bytes, hipMemcpyHostToDevice, ctx->stream));
  BaArgs& a = B.a;
  a.poses = reinterpret_cast<const hso_se3*>(d + B.o_poses); a.fixed = reinterpret_cast<const uint8_t*>(d + B.o_fixed);
  a.idist = reinterpret_cast<const double*>(d + B.o_idist); a.edges = reinterpret_cast<const hso_ba_edge*>(d + B.o_edges);
  a.n_poses = n_poses; a.n_points = n_points; a.n_edges = n_edges;
  a.huber_corner = B.huber_corner; a.huber_edge = B.huber_edge;
  a.lin = reinterpret_cast<double*>(d + B.o_lin); a.edge_err = reinterpret_cast<double*>(d + B.o_err);
  a.edge_chi2 = reinterpret_cast<double*>(d + B.o_chi); a.edge_rho = reinterpret_cast<double*>(d + B.o_rho);
  return HSO_OK;
}

// single problem: layout + reserve + place
static int ba_setup(BaDev& B, hso_gpu_ctx* ctx, const hso_se3* poses_f_w, const uint8_t* pose_fixed, int n_poses, const double* idist,
                    int n_points, const hso_ba_edge* edges, int n_edges, double huber_corner, double huber_edge)
{
  ba_layout(B, ctx, n_poses, n_points, edges, n_edges, huber_corner, huber_edge);
  char *d, *h;
  if (int rc = ba_reserve(ctx, B.total, B.in_bytes, &d, &h)) return rc;
  return ba_place(B, d, h, poses_f_w, pose_fixed, idist, edges);
}

// new state -> device (poses and inverse depths sit next to each other at the start of the work area)
static int ba_put_state(BaDev& B, const hso_se3* poses, const double* idist)
{
  memcpy(B.h + B.o_poses, poses, sizeof(hso_se3) * B.n_poses);
  memcpy(B.h + B.o_idist, idist, sizeof(double) * B.n_points);
  HSO_HIP_CHECK(B.ctx, hipMemcpyAsync(B.d, B.h, B.o_fixed, hipMemcpyHostToDevice, B.ctx->stream));
  return HSO_OK;
}

// computeActiveErrors + buildSystem at the resident state; the blocks stay on the device
static int ba_launch_linearize(BaDev& B)
{
  hso_gpu_ctx* ctx = B.ctx;
  char* d = B.d;
  HSO_HIP_CHECK(ctx, hipMemsetAsync(d + B.o_out, 0, B.total - B.o_out, ctx->stream));
  hipLaunchKernelGGL(k_ba_edges<true>, dim3((B.n_edges + BA_THREADS - 1) / BA_THREADS), dim3(BA_THREADS), 0, ctx->stream, B.a);
  hipLaunchKernelGGL(k_ba_points, dim3((B.n_points + BA_THREADS - 1) / BA_THREADS), dim3(BA_THREADS), 0, ctx->stream, B.a,
                     reinterpret_cast<const int*>(d + B.o_off), reinterpret_cast<const int*>(d + B.o_list),
                     reinterpret_cast<double*>(d + B.o_Hpp), reinterpret_cast<double*>(d + B.o_bp), reinterpret_cast<double*>(d + B.o_Hpc));
  hipLaunchKernelGGL(k_ba_poses, dim3(B.n_pairs + 1), dim3(BA_THREADS), 0, ctx->stream, B.a,
                     reinterpret_cast<const int*>(d + B.o_poff), reinterpret_cast<const int*>(d + B.o_plist),
                     reinterpret_cast<double*>(d + B.o_Hcc), reinterpret_cast<double*>(d + B.o_bc), reinterpret_cast<double*>(d + B.o_sum));
  HSO_HIP_CHECK(ctx, hipGetLastError());
  return HSO_OK;
}

// computeActiveErrors only (an LM trial): per-edge error / chi2 / rho and the two sums
static int ba_launch_errors(BaDev& B)
{
  hso_gpu_ctx* ctx = B.ctx;
  hipLaunchKernelGGL(k_ba_edges<false>, dim3((B.n_edges + BA_THREADS - 1) / BA_THREADS), dim3(BA_THREADS), 0, ctx->stream, B.a);
  hipLaunchKernelGGL(k_ba_chi2, dim3(1), dim3(BA_THREADS), 0, ctx->stream, B.a, reinterpret_cast<double*>(B.d + B.o_sum));
  HSO_HIP_CHECK(ctx, hipGetLastError());
  return HSO_OK;
}

static int ba_get(BaDev& B, void* dst, size_t off, size_t bytes)
{
  HSO_HIP_CHECK(B.ctx, hipMemcpyAsync(dst, B.d + off, bytes, hipMemcpyDeviceToHost, B.ctx->stream));
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
  BaDev B;
  if (int rc = ba_setup(B, ctx, poses_f_w, pose_fixed, n_poses, idist, n_points, edges, n_edges, huber_corner, huber_edge)) return rc;
  if (int rc = ba_launch_linearize(B)) return rc;
  int rc = HSO_OK;
  if (!rc) rc = ba_get(B, Hpp, B.o_Hpp, sizeof(double) * n_points);
  if (!rc) rc = ba_get(B, bp, B.o_bp, sizeof(double) * n_points);
  if (!rc) rc = ba_get(B, Hpc, B.o_Hpc, sizeof(double) * (size_t)n_points * n_poses * 6);
  if (!rc) rc = ba_get(B, Hcc, B.o_Hcc, sizeof(double) * (size_t)n_poses * n_poses * 36);
  if (!rc) rc = ba_get(B, bc, B.o_bc, sizeof(double) * n_poses * 6);
  if (!rc) rc = ba_get(B, edge_err, B.o_err, sizeof(double) * 2 * n_edges);
  if (!rc) rc = ba_get(B, edge_chi2, B.o_chi, sizeof(double) * n_edges);
  if (!rc) rc = ba_get(B, chi2_sum, B.o_sum, sizeof(double) * 2);
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

extern "C" int hso_gpu_ba_huber_deltas(hso_gpu_ctx* ctx, const hso_se3* poses_f_w, int n_poses, const double* idist, int n_points,
                                       const hso_ba_edge* edges, const double* obs_uv, int n_edges, double error_multiplier2,
                                       float* huber_corner, float* huber_edge)
{
  if (!ctx) return HSO_E_INVALID;
  if (!poses_f_w || !idist || !huber_corner || !huber_edge || n_poses <= 0 || n_points <= 0 || n_edges < 0 ||
      (n_edges > 0 && (!edges || !obs_uv)))
    return hso_fail(ctx, HSO_E_INVALID, "ba_huber_deltas: bad argument");
  *huber_corner = 0; *huber_edge = 0;
  if (n_edges == 0) return HSO_OK;   // both error lists empty: the reference leaves the deltas uninitialised
  if (int rc = ba_check_edges(ctx, edges, n_edges, n_points, n_poses, "ba_huber_deltas")) return rc;
  HSO_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  auto al = [](size_t b) { return (b + 255) & ~size_t(255); };
  size_t o = 0;
  const size_t o_poses = o; o += al(sizeof(hso_se3) * n_poses);
  const size_t o_idist = o; o += al(sizeof(double) * n_points);
  const size_t o_edges = o; o += al(sizeof(hso_ba_edge) * n_edges);
  const size_t o_uv = o; o += al(sizeof(double) * 2 * n_edges);
  const size_t in_bytes = o;
  const size_t o_err = o; o += al(sizeof(float) * n_edges);
  if (ctx->batch_cap < o) {
    HSO_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->d_batch) (void)hipFree(ctx->d_batch);
    ctx->d_batch = nullptr; ctx->batch_cap = 0;
    HSO_HIP_CHECK(ctx, hipMalloc(reinterpret_cast<void**>(&ctx->d_batch), o));
    ctx->batch_cap = o;
  }
  char* d = reinterpret_cast<char*>(ctx->d_batch);
  char* h = hso_pinned(ctx, 0, in_bytes);
  if (!h) return HSO_E_NOMEM;
  memcpy(h + o_poses, poses_f_w, sizeof(hso_se3) * n_poses);
  memcpy(h + o_idist, idist, sizeof(double) * n_points);
  memcpy(h + o_edges, edges, sizeof(hso_ba_edge) * n_edges);
  memcpy(h + o_uv, obs_uv, sizeof(double) * 2 * n_edges);
  HSO_HIP_CHECK(ctx, hipMemcpyAsync(d, h, in_bytes, hipMemcpyHostToDevice, ctx->stream));
  hipLaunchKernelGGL(k_ba_mad_errors, dim3((n_edges + BA_THREADS - 1) / BA_THREADS), dim3(BA_THREADS), 0, ctx->stream,
                     reinterpret_cast<const hso_se3*>(d + o_poses), reinterpret_cast<const double*>(d + o_idist),
                     reinterpret_cast<const hso_ba_edge*>(d + o_edges), reinterpret_cast<const double*>(d + o_uv), n_edges,
                     reinterpret_cast<float*>(d + o_err));
  HSO_HIP_CHECK(ctx, hipGetLastError());
  std::vector<float> err(n_edges);
  HSO_HIP_CHECK(ctx, hipMemcpyAsync(err.data(), d + o_err, sizeof(float) * n_edges, hipMemcpyDeviceToHost, ctx->stream));
  HSO_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  std::vector<float> errors_pt, errors_ls;
  for (int k = 0; k < n_edges; k++) (edges[k].type == HSO_FTR_EDGELET ? errors_ls : errors_pt).push_back(err[k]);
  // src/bundle_adjustment.cpp:664-680
  if (!errors_pt.empty() && !errors_ls.empty()) {
    *huber_corner = (float)(1.4826 * upper_median(errors_pt));
    *huber_edge = (float)(1.4826 * upper_median(errors_ls));
  } else if (errors_pt.empty() && !errors_ls.empty()) {
    *huber_corner = (float)(1.0 / error_multiplier2);
    *huber_edge = (float)(1.4826 * upper_median(errors_ls));
  } else if (!errors_pt.empty() && errors_ls.empty()) {
    *huber_corner = (float)(1.4826 * upper_median(errors_pt));
    *huber_edge = (float)(0.5 / error_multiplier2);
  }
  return HSO_OK;
}

// ---- g2o::SE3Quat on the host (thirdparty/g2o/g2o/types/se3quat.h): the pose update of VertexSE3Expmap ----
namespace {

struct Q4 { double x, y, z, w; };

inline Q4 qmul(const Q4& a, const Q4& b)
{
  return { a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y, a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z,
           a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x, a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z };
}

inline void qrot(const Q4& q, const double v[3], double o[3])   // Eigen QuaternionBase::_transformVector
{
  double uv[3] = { q.y * v[2] - q.z * v[1], q.z * v[0] - q.x * v[2], q.x * v[1] - q.y * v[0] };
  uv[0] += uv[0]; uv[1] += uv[1]; uv[2] += uv[2];
  o[0] = (v[0] + q.w * uv[0]) + (q.y * uv[2] - q.z * uv[1]);
  o[1] = (v[1] + q.w * uv[1]) + (q.z * uv[0] - q.x * uv[2]);
  o[2] = (v[2] + q.w * uv[2]) + (q.x * uv[1] - q.y * uv[0]);
}

inline void normalize_rotation(Q4& q)   // se3quat.h:280-285
{
  if (q.w < 0) { q.x = -q.x; q.y = -q.y; q.z = -q.z; q.w = -q.w; }
  const double n = std::sqrt(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w);
  q.x /= n; q.y /= n; q.z /= n; q.w /= n;
}

// SE3Quat::exp(update) * pose, update = [omega, upsilon] (se3quat.h:223-257, :104-110)
void se3quat_exp_times(const double* upd, hso_se3& pose)
{
  const double wx = upd[0], wy = upd[1], wz = upd[2];
  const double theta = std::sqrt(wx * wx + wy * wy + wz * wz);
  const double O[9] = { 0, -wz, wy, wz, 0, -wx, -wy, wx, 0 };
  double O2[9], R[9], V[9];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) O2[i * 3 + j] = (O[i * 3] * O[j] + O[i * 3 + 1] * O[3 + j]) + O[i * 3 + 2] * O[6 + j];
  if (theta < 0.00001) {
    for (int i = 0; i < 9; i++) { R[i] = ((i % 4 == 0 ? 1.0 : 0.0) + O[i]) + O2[i]; V[i] = R[i]; }
  } else {
    const double a = std::sin(theta) / theta, b = (1 - std::cos(theta)) / (theta * theta), c = (theta - std::sin(theta)) / std::pow(theta, 3);
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
    t = std::sqrt(t + 1.0);
    q.w = 0.5 * t; t = 0.5 / t;
    q.x = (R[7] - R[5]) * t; q.y = (R[2] - R[6]) * t; q.z = (R[3] - R[1]) * t;
  } else {
    int i = 0;
    if (R[4] > R[0]) i = 1;
    if (R[8] > R[i * 4]) i = 2;
    const int j = (i + 1) % 3, k = (j + 1) % 3;
    double qv[3];
    t = std::sqrt(R[i * 4] - R[j * 4] - R[k * 4] + 1.0);
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

// (H + lambda I) x = b through the scalar Schur complement of the inverse-depth unknowns.  Returns false where g2o's
// factorisation would fail (a vanishing or non-finite pivot).
struct SchurSolver {
  int n_points, n_poses, n_free, M;
  std::vector<int> col;                    // pose -> first row of its block in the reduced system, -1 = fixed
  std::vector<int> pp_off, pp_pose;        // CSR: free poses connected to each point
  std::vector<double> S, rhs, xc, w;

  void init(int n_points_, int n_poses_, const uint8_t* fixed, const hso_ba_edge* edges, int n_edges)
  {
    n_points = n_points_; n_poses = n_poses_;
    col.assign(n_poses, -1);
    n_free = 0;
    for (int i = 0; i < n_poses; i++) if (!fixed[i]) col[i] = 6 * n_free++;
    M = 6 * n_free;
    std::vector<std::vector<int>> con(n_points);
    for (int k = 0; k < n_edges; k++) {
      const hso_ba_edge& e = edges[k];
      for (int v : { e.host, e.target })
        if (col[v] >= 0 && std::find(con[e.point].begin(), con[e.point].end(), v) == con[e.point].end()) con[e.point].push_back(v);
    }
    pp_off.assign(n_points + 1, 0);
    for (int p = 0; p < n_points; p++) { std::sort(con[p].begin(), con[p].end()); pp_off[p + 1] = pp_off[p] + (int)con[p].size(); }
    pp_pose.resize(pp_off[n_points]);
    for (int p = 0; p < n_points; p++) std::copy(con[p].begin(), con[p].end(), pp_pose.begin() + pp_off[p]);
    S.resize((size_t)M * M); rhs.resize(M); xc.resize(M); w.resize(n_points);
  }

  // x = [points | poses (n_poses * 6, zeros at fixed ones)]
  bool solve(const double* Hpp, const double* bp, const double* Hpc, const double* Hcc, const double* bc, double lambda,
             double* x_points, double* x_poses)
  {
    std::fill(S.begin(), S.end(), 0.0);
    for (int i = 0; i < n_poses; i++) {
      if (col[i] < 0) continue;
      for (int q = 0; q < 6; q++) rhs[col[i] + q] = bc[i * 6 + q];
      for (int j = i; j < n_poses; j++) {
        if (col[j] < 0) continue;
        const double* blk = Hcc + ((size_t)i * n_poses + j) * 36;
        for (int r = 0; r < 6; r++)
          for (int c = 0; c < 6; c++) {
            S[(size_t)(col[i] + r) * M + col[j] + c] = blk[r * 6 + c];
            S[(size_t)(col[j] + c) * M + col[i] + r] = blk[r * 6 + c];
          }
      }
    }
    for (int k = 0; k < M; k++) S[(size_t)k * M + k] += lambda;
    bool ok = true;
    for (int p = 0; p < n_points; p++) {
      const double dpp = Hpp[p] + lambda;
      if (!(dpp != 0.0) || !std::isfinite(dpp)) { ok = false; w[p] = 0; continue; }
      const double inv = 1.0 / dpp;
      w[p] = inv;
      const double g = bp[p] * inv;
      for (int a = pp_off[p]; a < pp_off[p + 1]; a++) {
        const int ia = pp_pose[a];
        const double* Wa = Hpc + ((size_t)p * n_poses + ia) * 6;
        for (int r = 0; r < 6; r++) rhs[col[ia] + r] -= Wa[r] * g;
        for (int b = a; b < pp_off[p + 1]; b++) {
          const int ib = pp_pose[b];
          const double* Wb = Hpc + ((size_t)p * n_poses + ib) * 6;
          for (int r = 0; r < 6; r++) {
            const double wr = Wa[r] * inv;
            for (int c = 0; c < 6; c++) {
              const double v = wr * Wb[c];
              S[(size_t)(col[ia] + r) * M + col[ib] + c] -= v;
              if (ia != ib) S[(size_t)(col[ib] + c) * M + col[ia] + r] -= v;
            }
          }
        }
      }
    }
    // dense LDL^T of the reduced system (lower triangle), no pivoting like the reference's SimplicialLDLT
    for (int j = 0; j < M && ok; j++) {
      double dj = S[(size_t)j * M + j];
      for (int k = 0; k < j; k++) dj -= S[(size_t)j * M + k] * S[(size_t)j * M + k] * S[(size_t)k * M + k];
      if (!(dj != 0.0) || !std::isfinite(dj)) { ok = false; break; }
      S[(size_t)j * M + j] = dj;
      for (int i = j + 1; i < M; i++) {
        double s = S[(size_t)i * M + j];
        for (int k = 0; k < j; k++) s -= S[(size_t)i * M + k] * S[(size_t)j * M + k] * S[(size_t)k * M + k];
        S[(size_t)i * M + j] = s / dj;
      }
    }
    std::fill(x_poses, x_poses + (size_t)n_poses * 6, 0.0);
    if (!ok) { std::fill(x_points, x_points + n_points, 0.0); return false; }
    for (int i = 0; i < M; i++) { double s = rhs[i]; for (int k = 0; k < i; k++) s -= S[(size_t)i * M + k] * xc[k]; xc[i] = s; }
    for (int i = 0; i < M; i++) xc[i] /= S[(size_t)i * M + i];
    for (int i = M - 1; i >= 0; i--) { double s = xc[i]; for (int k = i + 1; k < M; k++) s -= S[(size_t)k * M + i] * xc[k]; xc[i] = s; }
    for (int i = 0; i < n_poses; i++)
      if (col[i] >= 0) for (int q = 0; q < 6; q++) x_poses[i * 6 + q] = xc[col[i] + q];
    for (int p = 0; p < n_points; p++) {
      double s = bp[p];
      for (int a = pp_off[p]; a < pp_off[p + 1]; a++) {
        const int ia = pp_pose[a];
        const double* Wa = Hpc + ((size_t)p * n_poses + ia) * 6;
        for (int r = 0; r < 6; r++) s -= Wa[r] * xc[col[ia] + r];
      }
      x_points[p] = s * w[p];
    }
    return true;
  }
};

}  // namespace

// One local-BA problem being optimised: the Levenberg loop of OptimizationAlgorithmLevenberg::solve written as a state
// machine, so that many problems advance in lockstep — advance() does host work until the next device result is needed,
// queues the device work on the context's stream and returns; the caller synchronises ONCE for all problems and calls
// advance() again.  The arithmetic per problem is the same statement sequence whether it runs alone or among others.
struct BaLm {
  enum State { INIT_WAIT, ITER_BEGIN, LIN_WAIT, TRIAL_BEGIN, TRIAL_WAIT, FINISH, CHI_WAIT, DONE };
  hso_gpu_ctx* ctx;
  BaDev B;
  SchurSolver sol;
  hso_se3* poses_f_w; const uint8_t* pose_fixed; double* idist;
  int n_poses, n_points, n_edges, n_iter;
  double* edge_chi2_out; hso_ba_result* result;
  std::vector<double> Hpp, bp, Hpc, Hcc, bc, xp, xc, idist_bak;
  std::vector<hso_se3> poses_bak;
  double chi[2];
  double lambda, ni, currentChi, tempChi, iniChi, rho;
  int nBad, stop, it, qmax;
  bool ok2;
  State st;

  int begin()   // runSparseBAOptimizer: computeActiveErrors(); init_error = activeChi2()
  {
    memset(result, 0, sizeof(*result));
    Hpp.resize(n_points); bp.resize(n_points); Hpc.resize((size_t)n_points * n_poses * 6); Hcc.resize((size_t)n_poses * n_poses * 36);
    bc.resize((size_t)n_poses * 6); xp.resize(n_points); xc.resize((size_t)n_poses * 6); idist_bak.resize(n_points);
    poses_bak.resize(n_poses);
    chi[0] = chi[1] = 0;
    lambda = -1.; ni = 2.; nBad = 0; stop = 0; it = 0; qmax = 0; rho = 0; currentChi = tempChi = iniChi = 0; ok2 = true;
    if (int rc = ba_launch_errors(B)) return rc;
    if (int rc = ba_get(B, chi, B.o_sum, sizeof(chi))) return rc;
    st = INIT_WAIT;
    return HSO_OK;
  }

  // returns < 0 on error; afterwards st == DONE or device work is queued and a synchronise is due
  int advance()
  {
    for (;;) {
      switch (st) {
        case INIT_WAIT:
          result->init_chi2 = chi[0];
          result->robust_chi2 = chi[1];
          st = ITER_BEGIN;
          break;
        case ITER_BEGIN: {
          if (it >= n_iter) { st = FINISH; break; }
          // solve(): computeActiveErrors, currentChi = activeRobustChi2, buildSystem
          if (int rc = ba_launch_linearize(B)) return rc;
          int rc = ba_get(B, Hpp.data(), B.o_Hpp, sizeof(double) * n_points);
          if (!rc) rc = ba_get(B, bp.data(), B.o_bp, sizeof(double) * n_points);
          if (!rc) rc = ba_get(B, Hpc.data(), B.o_Hpc, sizeof(double) * Hpc.size());
          if (!rc) rc = ba_get(B, Hcc.data(), B.o_Hcc, sizeof(double) * Hcc.size());
          if (!rc) rc = ba_get(B, bc.data(), B.o_bc, sizeof(double) * bc.size());
          if (!rc) rc = ba_get(B, chi, B.o_sum, sizeof(chi));
          if (rc) return rc;
          st = LIN_WAIT;
          return HSO_OK;
        }
        case LIN_WAIT:
          currentChi = chi[1]; tempChi = currentChi;
          iniChi = currentChi;
          if (it == 0) {   // computeLambdaInit: tau (1e-5) * the largest diagonal entry over all free vertices
            double maxDiagonal = 0.;
            for (int p = 0; p < n_points; p++) maxDiagonal = std::max(std::fabs(Hpp[p]), maxDiagonal);
            for (int i = 0; i < n_poses; i++)
              if (!pose_fixed[i]) for (int q = 0; q < 6; q++) maxDiagonal = std::max(std::fabs(Hcc[((size_t)i * n_poses + i) * 36 + q * 7]), maxDiagonal);
            lambda = 1e-5 * maxDiagonal;
            ni = 2; nBad = 0;
          }
          rho = 0; qmax = 0;
          st = TRIAL_BEGIN;
          break;
        case TRIAL_BEGIN: {
          std::copy(poses_f_w, poses_f_w + n_poses, poses_bak.begin());   // _optimizer->push()
          std::copy(idist, idist + n_points, idist_bak.begin());
          ok2 = sol.solve(Hpp.data(), bp.data(), Hpc.data(), Hcc.data(), bc.data(), lambda, xp.data(), xc.data());
          result->n_solves++;
          for (int p = 0; p < n_points; p++) idist[p] += xp[p];                          // VertexSBAPointID::oplusImpl
          for (int i = 0; i < n_poses; i++) if (!pose_fixed[i]) se3quat_exp_times(&xc[(size_t)i * 6], poses_f_w[i]);  // VertexSE3Expmap::oplusImpl
          int rc = ba_put_state(B, poses_f_w, idist);
          if (!rc) rc = ba_launch_errors(B);
          if (!rc) rc = ba_get(B, chi, B.o_sum, sizeof(chi));
          if (rc) return rc;
          st = TRIAL_WAIT;
          return HSO_OK;
        }
        case TRIAL_WAIT: {
          tempChi = ok2 ? chi[1] : 1.7976931348623157e308;
          rho = currentChi - tempChi;
          double scale = 0.;                                               // computeScale
          for (int p = 0; p < n_points; p++) scale += xp[p] * (lambda * xp[p] + bp[p]);
          for (int i = 0; i < n_poses; i++)
            if (!pose_fixed[i]) for (int q = 0; q < 6; q++) scale += xc[i * 6 + q] * (lambda * xc[i * 6 + q] + bc[i * 6 + q]);
          scale += 1e-3;
          rho /= scale;
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
            std::copy(poses_bak.begin(), poses_bak.end(), poses_f_w);     // _optimizer->pop(): vertices only, edge errors stay
            std::copy(idist_bak.begin(), idist_bak.end(), idist);
            if (int rc2 = ba_put_state(B, poses_f_w, idist)) return rc2;
          }
          qmax++;
          if (rho < 0 && qmax < 5) { st = TRIAL_BEGIN; break; }   // setMaxTrialsAfterFailure(5), src/bundle_adjustment.cpp:571
          result->iterations = it + 1;
          result->robust_chi2 = currentChi;
          if (qmax == 5 || rho == 0) { stop = 1; st = FINISH; break; }
          if ((iniChi - currentChi) * 1e3 < iniChi) nBad++; else nBad = 0;   // optimization_algorithm_levenberg.cpp:154-161
          if (nBad >= 3) { stop = 2; st = FINISH; break; }
          it++;
          st = ITER_BEGIN;
          break;
        }
        case FINISH:
          result->stop = stop;
          result->lambda = lambda;
          result->final_chi2 = chi[0];   // activeChi2() of the last computeActiveErrors
          if (edge_chi2_out) {
            if (int rc = ba_get(B, edge_chi2_out, B.o_chi, sizeof(double) * n_edges)) return rc;
            st = CHI_WAIT;
            return HSO_OK;
          }
          st = DONE;
          return HSO_OK;
        case CHI_WAIT:
          st = DONE;
          return HSO_OK;
        case DONE:
          return HSO_OK;
      }
    }
  }
};

// pinned staging is used by ba_put_state of every problem between two synchronises: each problem keeps its own slice
extern "C" int hso_gpu_ba_optimize_multi(hso_gpu_ctx* ctx, const hso_ba_problem* problems, int n_problems)
{
  if (!ctx) return HSO_E_INVALID;
  if (n_problems < 0 || (n_problems > 0 && !problems)) return hso_fail(ctx, HSO_E_INVALID, "ba_optimize_multi: bad argument");
  if (n_problems == 0) return HSO_OK;
  std::vector<BaLm> lm(n_problems);
  size_t dev = 0, pin = 0;
  std::vector<size_t> d_off(n_problems), h_off(n_problems);
  for (int q = 0; q < n_problems; q++) {
    const hso_ba_problem& P = problems[q];
    if (!P.poses_f_w || !P.pose_fixed || !P.idist || !P.edges || !P.result || P.n_poses <= 0 || P.n_points <= 0 || P.n_edges <= 0 || P.n_iter < 0)
      return hso_fail(ctx, HSO_E_INVALID, "ba_optimize: bad argument");
    if (int rc = ba_check_edges(ctx, P.edges, P.n_edges, P.n_points, P.n_poses, "ba_optimize")) return rc;
    BaLm& L = lm[q];
    L.ctx = ctx; L.poses_f_w = P.poses_f_w; L.pose_fixed = P.pose_fixed; L.idist = P.idist;
    L.n_poses = P.n_poses; L.n_points = P.n_points; L.n_edges = P.n_edges; L.n_iter = P.n_iter;
    L.edge_chi2_out = P.edge_chi2_out; L.result = P.result;
    ba_layout(L.B, ctx, P.n_poses, P.n_points, P.edges, P.n_edges, P.huber_corner, P.huber_edge);
    d_off[q] = dev; dev += L.B.total;
    h_off[q] = pin; pin += L.B.in_bytes;
  }
  char *d, *h;
  if (int rc = ba_reserve(ctx, dev, pin, &d, &h)) return rc;
  for (int q = 0; q < n_problems; q++) {
    const hso_ba_problem& P = problems[q];
    if (int rc = ba_place(lm[q].B, d + d_off[q], h + h_off[q], P.poses_f_w, P.pose_fixed, P.idist, P.edges)) return rc;
    lm[q].sol.init(P.n_points, P.n_poses, P.pose_fixed, P.edges, P.n_edges);
    if (int rc = lm[q].begin()) return rc;
  }
  for (;;) {
    HSO_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    bool pending = false;
    for (int q = 0; q < n_problems; q++) {
      if (lm[q].st == BaLm::DONE) continue;
      if (int rc = lm[q].advance()) return rc;
      if (lm[q].st != BaLm::DONE) pending = true;
    }
    if (!pending) break;
  }
  HSO_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  return HSO_OK;
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
