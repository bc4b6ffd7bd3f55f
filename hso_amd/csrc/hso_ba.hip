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
#pragma unroll
  for (int i = 0; i < BA_LIN; i++) a.lin[(size_t)i * a.n_edges + k] = rec[i];
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

extern "C" int hso_gpu_ba_linearize(hso_gpu_ctx* ctx, const hso_se3* poses_f_w, const uint8_t* pose_fixed, int n_poses,
                                    const double* idist, int n_points, const hso_ba_edge* edges, int n_edges,
                                    double huber_corner, double huber_edge, double* Hpp, double* bp, double* Hpc,
                                    double* Hcc, double* bc, double* edge_err, double* edge_chi2, double* chi2_sum)
{
  if (!ctx) return HSO_E_INVALID;
  if (!poses_f_w || !pose_fixed || !idist || !edges || n_poses <= 0 || n_points <= 0 || n_edges <= 0 || !Hpp || !bp || !Hpc ||
      !Hcc || !bc || !edge_err || !edge_chi2 || !chi2_sum)
    return hso_fail(ctx, HSO_E_INVALID, "ba_linearize: bad argument");
  for (int k = 0; k < n_edges; k++) {
    const hso_ba_edge& e = edges[k];
    if (e.point < 0 || e.point >= n_points || e.host < 0 || e.host >= n_poses || e.target < 0 || e.target >= n_poses ||
        e.host == e.target || e.level < 0 || e.level > 14)
      return hso_fail(ctx, HSO_E_INVALID, "ba_linearize: edge index out of range");
  }
  HSO_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  // CSR of edges by point, edge order kept inside a point (g2o visits edges in insertion order)
  std::vector<int> off(n_points + 1, 0), list(n_edges);
  for (int k = 0; k < n_edges; k++) off[edges[k].point + 1]++;
  for (int p = 0; p < n_points; p++) off[p + 1] += off[p];
  { std::vector<int> cur(off.begin(), off.end() - 1); for (int k = 0; k < n_edges; k++) list[cur[edges[k].point]++] = k; }

  // CSR of edges by pose-pair block (same block numbering as k_ba_poses), edge order kept
  const int n_pairs = n_poses * (n_poses + 1) / 2;
  auto pair_id = [n_poses](int i, int j) { return i * n_poses - i * (i - 1) / 2 + (j - i); };  // i <= j
  std::vector<int> poff(n_pairs + 1, 0), plist((size_t)3 * n_edges);
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
  const size_t o_poses = o; o += al(sizeof(hso_se3) * n_poses);
  const size_t o_fixed = o; o += al(n_poses);
  const size_t o_idist = o; o += al(sizeof(double) * n_points);
  const size_t o_edges = o; o += al(sizeof(hso_ba_edge) * n_edges);
  const size_t o_off = o; o += al(sizeof(int) * (n_points + 1));
  const size_t o_list = o; o += al(sizeof(int) * n_edges);
  const size_t o_poff = o; o += al(sizeof(int) * (n_pairs + 1));
  const size_t o_plist = o; o += al(sizeof(int) * 3 * (size_t)n_edges);
  const size_t in_bytes = o;
  const size_t o_lin = o; o += al(sizeof(double) * BA_LIN * n_edges);
  const size_t o_rho = o; o += al(sizeof(double) * n_edges);
  const size_t o_out = o;
  const size_t o_Hpp = o; o += al(sizeof(double) * n_points);
  const size_t o_bp = o; o += al(sizeof(double) * n_points);
  const size_t o_Hpc = o; o += al(sizeof(double) * (size_t)n_points * n_poses * 6);
  const size_t o_Hcc = o; o += al(sizeof(double) * (size_t)n_poses * n_poses * 36);
  const size_t o_bc = o; o += al(sizeof(double) * n_poses * 6);
  const size_t o_err = o; o += al(sizeof(double) * 2 * n_edges);
  const size_t o_chi = o; o += al(sizeof(double) * n_edges);
  const size_t o_sum = o; o += 256;
  if (ctx->batch_cap < o) {  // grow-only staging buffer of the context (shared with the other batched entry points)
    HSO_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->d_batch) (void)hipFree(ctx->d_batch);
    ctx->d_batch = nullptr; ctx->batch_cap = 0;
    HSO_HIP_CHECK(ctx, hipMalloc(reinterpret_cast<void**>(&ctx->d_batch), o));
    ctx->batch_cap = o;
  }
  char* d = reinterpret_cast<char*>(ctx->d_batch);
  char* h = hso_pinned(ctx, 0, in_bytes);
  if (!h) return HSO_E_NOMEM;
  memset(h, 0, in_bytes);
  memcpy(h + o_poses, poses_f_w, sizeof(hso_se3) * n_poses);
  memcpy(h + o_fixed, pose_fixed, n_poses);
  memcpy(h + o_idist, idist, sizeof(double) * n_points);
  memcpy(h + o_edges, edges, sizeof(hso_ba_edge) * n_edges);
  memcpy(h + o_off, off.data(), sizeof(int) * (n_points + 1));
  memcpy(h + o_list, list.data(), sizeof(int) * n_edges);
  memcpy(h + o_poff, poff.data(), sizeof(int) * (n_pairs + 1));
  memcpy(h + o_plist, plist.data(), sizeof(int) * 3 * (size_t)n_edges);
  hipError_t e = hipMemcpyAsync(d, h, in_bytes, hipMemcpyHostToDevice, ctx->stream);
  if (e == hipSuccess) e = hipMemsetAsync(d + o_out, 0, o - o_out, ctx->stream);
  if (e == hipSuccess) {
    BaArgs a;
    a.poses = reinterpret_cast<const hso_se3*>(d + o_poses); a.fixed = reinterpret_cast<const uint8_t*>(d + o_fixed);
    a.idist = reinterpret_cast<const double*>(d + o_idist); a.edges = reinterpret_cast<const hso_ba_edge*>(d + o_edges);
    a.n_poses = n_poses; a.n_points = n_points; a.n_edges = n_edges;
    a.huber_corner = huber_corner; a.huber_edge = huber_edge;
    a.lin = reinterpret_cast<double*>(d + o_lin); a.edge_err = reinterpret_cast<double*>(d + o_err);
    a.edge_chi2 = reinterpret_cast<double*>(d + o_chi); a.edge_rho = reinterpret_cast<double*>(d + o_rho);
    hipLaunchKernelGGL(k_ba_edges, dim3((n_edges + BA_THREADS - 1) / BA_THREADS), dim3(BA_THREADS), 0, ctx->stream, a);
    hipLaunchKernelGGL(k_ba_points, dim3((n_points + BA_THREADS - 1) / BA_THREADS), dim3(BA_THREADS), 0, ctx->stream, a,
                       reinterpret_cast<const int*>(d + o_off), reinterpret_cast<const int*>(d + o_list),
                       reinterpret_cast<double*>(d + o_Hpp), reinterpret_cast<double*>(d + o_bp), reinterpret_cast<double*>(d + o_Hpc));
    hipLaunchKernelGGL(k_ba_poses, dim3(n_pairs + 1), dim3(BA_THREADS), 0, ctx->stream, a,
                       reinterpret_cast<const int*>(d + o_poff), reinterpret_cast<const int*>(d + o_plist),
                       reinterpret_cast<double*>(d + o_Hcc), reinterpret_cast<double*>(d + o_bc), reinterpret_cast<double*>(d + o_sum));
    e = hipGetLastError();
  }
  auto back = [&](void* dst, size_t off_, size_t bytes) { if (e == hipSuccess) e = hipMemcpyAsync(dst, d + off_, bytes, hipMemcpyDeviceToHost, ctx->stream); };
  back(Hpp, o_Hpp, sizeof(double) * n_points);
  back(bp, o_bp, sizeof(double) * n_points);
  back(Hpc, o_Hpc, sizeof(double) * (size_t)n_points * n_poses * 6);
  back(Hcc, o_Hcc, sizeof(double) * (size_t)n_poses * n_poses * 36);
  back(bc, o_bc, sizeof(double) * n_poses * 6);
  back(edge_err, o_err, sizeof(double) * 2 * n_edges);
  back(edge_chi2, o_chi, sizeof(double) * n_edges);
  back(chi2_sum, o_sum, sizeof(double) * 2);
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  if (e != hipSuccess) { ctx->err = std::string("ba_linearize: ") + hipGetErrorString(e); return HSO_E_HIP; }
  return HSO_OK;
}
