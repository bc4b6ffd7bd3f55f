/*
 * hso_oracle_ba.c — local bundle adjustment: edge errors, analytic Jacobians and the g2o-style
 * accumulation of the (robustified) normal equations.  TEST INFRASTRUCTURE (see hso_oracle.h).
 * Follows include/hso/bundle_adjustment.h:204-404 (EdgeProjectID2UV, EdgeProjectID2UVEdgeLet)
 * and, for the accumulation, the vendored g2o of the reference:
 * thirdparty/g2o/g2o/core/base_multi_edge.hpp:36-48 (constructQuadraticForm), :171-222
 * (computeQuadraticForm), robust_kernel_impl.cpp:78-91 (Huber), base_edge.h:96-102
 * (robustInformation = rho[1] * Omega).  Edges are visited in array order, as g2o visits its
 * active edge list.  The reference's mixed tangent conventions are reproduced as written:
 * Jpdxi has g2o's [omega, upsilon] column order but is right-multiplied by -Tth.Adj() built
 * with Sophus' [upsilon, omega] block layout (bundle_adjustment.h:264-284, se3.cpp:108-118).
 */
#include "hso_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
  double err[2];
  double Jp[2];      /* d err / d idist        (2x1, or 1x1 in [0] for edgelets) */
  double Jh[12];     /* d err / d host pose    (2x6 row-major, or 1x6 in the first row) */
  double Jt[12];     /* d err / d target pose */
  int dim;           /* 2 or 1 */
} ba_lin_t;

static void ba_edge_linearize(const hso_ba_edge* e, const hso_se3* poses, const double* idist, ba_lin_t* L)
{
  /* SE3 Ttw(target), Thw(host): SE3(Quaterniond, Vector3d) normalises the quaternion */
  hso_se3 I, Ttw, Thw, Thw_inv, Tth;
  hso_or_se3_identity(&I);
  hso_or_se3_mul(&poses[e->target], &I, &Ttw);
  hso_or_se3_mul(&poses[e->host], &I, &Thw);
  hso_or_se3_inverse(&Thw, &Thw_inv);
  hso_or_se3_mul(&Ttw, &Thw_inv, &Tth);
  const double idHost = idist[e->point];
  const double inv = 1.0 / idHost;
  const double pH[3] = { e->fH[0] * inv, e->fH[1] * inv, e->fH[2] * inv };
  double pT[3];
  hso_or_se3_apply(&Tth, pH, pT);
  const double proj[2] = { pT[0] / pT[2], pT[1] / pT[2] };
  double R[9];
  hso_or_so3_matrix(Tth.q, R);
  const double* t = Tth.t;
  const double Rf2 = R[6] * e->fH[0] + R[7] * e->fH[1] + R[8] * e->fH[2];
  double Juvdd[2];
  Juvdd[0] = -(t[0] - proj[0] * t[2]) / (Rf2 + idHost * t[2]);
  Juvdd[1] = -(t[1] - proj[1] * t[2]) / (Rf2 + idHost * t[2]);
  double Jpdxi[12];
  const double x = pT[0], y = pT[1], z = pT[2], z_2 = z * z;
  Jpdxi[0] = x * y / z_2; Jpdxi[1] = -(1 + (x * x / z_2)); Jpdxi[2] = y / z; Jpdxi[3] = -1. / z; Jpdxi[4] = 0; Jpdxi[5] = x / z_2;
  Jpdxi[6] = (1 + y * y / z_2); Jpdxi[7] = -x * y / z_2; Jpdxi[8] = -x / z; Jpdxi[9] = 0; Jpdxi[10] = -1. / z; Jpdxi[11] = y / z_2;
  /* adHost = -Tth.Adj(): [[R, hat(t) R], [0, R]] (se3.cpp:108-118) */
  double Adj[36];
  memset(Adj, 0, sizeof(Adj));
  const double hat[9] = { 0, -t[2], t[1], t[2], 0, -t[0], -t[1], t[0], 0 };
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      Adj[i * 6 + j] = R[i * 3 + j];
      Adj[(3 + i) * 6 + 3 + j] = R[i * 3 + j];
      double s = hat[i * 3 + 0] * R[0 * 3 + j];
      s += hat[i * 3 + 1] * R[1 * 3 + j];
      s += hat[i * 3 + 2] * R[2 * 3 + j];
      Adj[i * 6 + 3 + j] = s;
    }
  double JhFull[12];
  for (int r = 0; r < 2; r++)
    for (int c = 0; c < 6; c++) {
      double s = 0;
      for (int k = 0; k < 6; k++) s += Jpdxi[r * 6 + k] * (-Adj[k * 6 + c]);
      JhFull[r * 6 + c] = s;
    }
  memset(L, 0, sizeof(*L));
  if (e->type == HSO_FTR_EDGELET) {
    L->dim = 1;
    const double n0 = e->normal[0], n1 = e->normal[1];
    L->err[0] = e->meas[0] - (n0 * proj[0] + n1 * proj[1]);
    L->Jp[0] = n0 * Juvdd[0] + n1 * Juvdd[1];
    for (int c = 0; c < 6; c++) {
      /* _normal.transpose()*Jpdxi*adHost evaluates left to right: (n^T Jpdxi) * adHost */
      double nJ[6];
      for (int k = 0; k < 6; k++) nJ[k] = n0 * Jpdxi[k] + n1 * Jpdxi[6 + k];
      double s = 0;
      for (int k = 0; k < 6; k++) s += nJ[k] * (-Adj[k * 6 + c]);
      L->Jh[c] = s;
      L->Jt[c] = nJ[c];
    }
  } else {
    L->dim = 2;
    L->err[0] = e->meas[0] - proj[0];
    L->err[1] = e->meas[1] - proj[1];
    L->Jp[0] = Juvdd[0]; L->Jp[1] = Juvdd[1];
    memcpy(L->Jh, JhFull, sizeof(JhFull));
    memcpy(L->Jt, Jpdxi, sizeof(Jpdxi));
  }
}

void hso_or_ba_linearize(const hso_se3* poses, const uint8_t* pose_fixed, int n_poses, const double* idist, int n_points,
                         const hso_ba_edge* edges, int n_edges, double huber_corner, double huber_edge,
                         double* Hpp, double* bp, double* Hpc, double* Hcc, double* bc,
                         double* edge_err, double* edge_chi2, double* chi2_sum)
{
  memset(Hpp, 0, sizeof(double) * (size_t)n_points);
  memset(bp, 0, sizeof(double) * (size_t)n_points);
  memset(Hpc, 0, sizeof(double) * (size_t)n_points * n_poses * 6);
  memset(Hcc, 0, sizeof(double) * (size_t)n_poses * n_poses * 36);
  memset(bc, 0, sizeof(double) * (size_t)n_poses * 6);
  chi2_sum[0] = chi2_sum[1] = 0;
  for (int k = 0; k < n_edges; k++) {
    const hso_ba_edge* e = &edges[k];
    ba_lin_t L;
    ba_edge_linearize(e, poses, idist, &L);
    const float inv_sigma2 = 1.0 / ((1 << e->level) * (1 << e->level));
    const double om = inv_sigma2;
    double chi2 = 0;
    for (int d = 0; d < L.dim; d++) chi2 += L.err[d] * om * L.err[d];
    edge_err[2 * k] = L.err[0]; edge_err[2 * k + 1] = L.dim == 2 ? L.err[1] : 0.0;
    edge_chi2[k] = chi2;
    /* RobustKernelHuber::robustify */
    const double delta = (e->type == HSO_FTR_EDGELET) ? huber_edge : huber_corner;
    const double dsqr = delta * delta;
    double rho0, rho1;
    if (chi2 <= dsqr) { rho0 = chi2; rho1 = 1.; }
    else { const double sqrte = sqrt(chi2); rho0 = 2 * sqrte * delta - dsqr; rho1 = delta / sqrte; }
    chi2_sum[0] += chi2; chi2_sum[1] += rho0;
    const double omega = rho1 * om;         /* robustInformation(rho) */
    double omega_r[2];
    for (int d = 0; d < L.dim; d++) { omega_r[d] = -(om * L.err[d]); omega_r[d] *= rho1; }
    /* vertex 0: point (never fixed, bundle_adjustment.cpp:695) */
    const int p = e->point, h = e->host, t = e->target;
    double AtO_p[2];
    for (int d = 0; d < L.dim; d++) AtO_p[d] = L.Jp[d] * omega;
    { double s = 0, g = 0; for (int d = 0; d < L.dim; d++) { s += AtO_p[d] * L.Jp[d]; g += L.Jp[d] * omega_r[d]; } Hpp[p] += s; bp[p] += g; }
    if (!pose_fixed[h])
      for (int c = 0; c < 6; c++) { double s = 0; for (int d = 0; d < L.dim; d++) s += AtO_p[d] * L.Jh[d * 6 + c]; Hpc[((size_t)p * n_poses + h) * 6 + c] += s; }
    if (!pose_fixed[t])
      for (int c = 0; c < 6; c++) { double s = 0; for (int d = 0; d < L.dim; d++) s += AtO_p[d] * L.Jt[d * 6 + c]; Hpc[((size_t)p * n_poses + t) * 6 + c] += s; }
    /* vertex 1: host pose */
    if (!pose_fixed[h]) {
      for (int r = 0; r < 6; r++) {
        double g = 0;
        for (int d = 0; d < L.dim; d++) g += L.Jh[d * 6 + r] * omega_r[d];
        bc[h * 6 + r] += g;
        for (int c = 0; c < 6; c++) {
          double s = 0;
          for (int d = 0; d < L.dim; d++) s += (L.Jh[d * 6 + r] * omega) * L.Jh[d * 6 + c];
          Hcc[((size_t)h * n_poses + h) * 36 + r * 6 + c] += s;
        }
      }
      if (!pose_fixed[t] && t != h) {
        /* off-diagonal block, stored in the upper half: (min,max); transposed when host > target */
        for (int r = 0; r < 6; r++)
          for (int c = 0; c < 6; c++) {
            double s = 0;
            for (int d = 0; d < L.dim; d++) s += (L.Jh[d * 6 + r] * omega) * L.Jt[d * 6 + c];
            if (h < t) Hcc[((size_t)h * n_poses + t) * 36 + r * 6 + c] += s;
            else Hcc[((size_t)t * n_poses + h) * 36 + c * 6 + r] += s;
          }
      }
    }
    /* vertex 2: target pose */
    if (!pose_fixed[t]) {
      for (int r = 0; r < 6; r++) {
        double g = 0;
        for (int d = 0; d < L.dim; d++) g += L.Jt[d * 6 + r] * omega_r[d];
        bc[t * 6 + r] += g;
        for (int c = 0; c < 6; c++) {
          double s = 0;
          for (int d = 0; d < L.dim; d++) s += (L.Jt[d * 6 + r] * omega) * L.Jt[d * 6 + c];
          Hcc[((size_t)t * n_poses + t) * 36 + r * 6 + c] += s;
        }
      }
    }
  }
}

/* ---- Huber deltas, src/bundle_adjustment.cpp:618-680 ---- */
void hso_or_ba_huber_deltas(const hso_se3* poses, int n_poses, const double* idist, int n_points, const hso_ba_edge* edges,
                            const double* obs_uv, int n_edges, double error_multiplier2, float* huber_corner, float* huber_edge)
{
  (void)n_poses; (void)n_points;
  float* errors_pt = (float*)malloc(sizeof(float) * (size_t)(n_edges > 0 ? n_edges : 1));
  float* errors_ls = (float*)malloc(sizeof(float) * (size_t)(n_edges > 0 ? n_edges : 1));
  int n_pt = 0, n_ls = 0;
  for (int k = 0; k < n_edges; k++) {
    const hso_ba_edge* e = &edges[k];
    /* SE3 Tth = (*it_ft)->frame->T_f_w_ * host_frame->T_f_w_.inverse(); pTarget = Tth * pHost (:626-634) */
    hso_se3 Thw_inv, Tth;
    hso_or_se3_inverse(&poses[e->host], &Thw_inv);
    hso_or_se3_mul(&poses[e->target], &Thw_inv, &Tth);
    const double inv = 1.0 / idist[e->point];
    const double pH[3] = { e->fH[0] * inv, e->fH[1] * inv, e->fH[2] * inv };
    double pT[3];
    hso_or_se3_apply(&Tth, pH, pT);
    double ex = obs_uv[2 * k] - pT[0] / pT[2], ey = obs_uv[2 * k + 1] - pT[1] / pT[2];
    const double sc = 1.0 / (1 << e->level);
    ex *= sc; ey *= sc;
    if (e->type == HSO_FTR_EDGELET) errors_ls[n_ls++] = (float)fabs(e->normal[0] * ex + e->normal[1] * ey);
    else errors_pt[n_pt++] = (float)sqrt(ex * ex + ey * ey);
  }
  float hc = 0, he = 0;
  if (n_pt > 0 && n_ls > 0) {
    hc = (float)(1.4826 * hso_or_median_f(errors_pt, n_pt));
    he = (float)(1.4826 * hso_or_median_f(errors_ls, n_ls));
  } else if (n_pt == 0 && n_ls > 0) {
    hc = (float)(1.0 / error_multiplier2);
    he = (float)(1.4826 * hso_or_median_f(errors_ls, n_ls));
  } else if (n_pt > 0 && n_ls == 0) {
    hc = (float)(1.4826 * hso_or_median_f(errors_pt, n_pt));
    he = (float)(0.5 / error_multiplier2);
  }  /* else: the reference leaves both uninitialised (:678-680) */
  *huber_corner = hc; *huber_edge = he;
  free(errors_pt); free(errors_ls);
}

/* ---- the g2o Levenberg-Marquardt driver over the blocks above ----
 * SparseOptimizer::optimize (sparse_optimizer.cpp:354-420) -> OptimizationAlgorithmLevenberg::solve
 * (optimization_algorithm_levenberg.cpp:61-164).  The reference factors the full sparse system
 * (BlockSolverX without marginalised vertices + LinearSolverEigen = SimplicialLDLT); this restatement
 * assembles the same system densely — unknowns: the points (1 each) then the free poses (6 each) —
 * and factors it with an unpivoted LDL^T, which is what SimplicialLDLT computes up to the fill-reducing
 * permutation. */
static int dense_ldlt_solve(double* A, double* b, int n)   /* A destroyed (lower triangle used), b -> x; 0 on failure */
{
  for (int j = 0; j < n; j++) {
    double d = A[(size_t)j * n + j];
    for (int k = 0; k < j; k++) d -= A[(size_t)j * n + k] * A[(size_t)j * n + k] * A[(size_t)k * n + k];
    if (!(d != 0.0) || !isfinite(d)) return 0;
    A[(size_t)j * n + j] = d;
    for (int i = j + 1; i < n; i++) {
      double s = A[(size_t)i * n + j];
      for (int k = 0; k < j; k++) s -= A[(size_t)i * n + k] * A[(size_t)j * n + k] * A[(size_t)k * n + k];
      A[(size_t)i * n + j] = s / d;
    }
  }
  for (int i = 0; i < n; i++) { double s = b[i]; for (int k = 0; k < i; k++) s -= A[(size_t)i * n + k] * b[k]; b[i] = s; }
  for (int i = 0; i < n; i++) b[i] /= A[(size_t)i * n + i];
  for (int i = n - 1; i >= 0; i--) { double s = b[i]; for (int k = i + 1; k < n; k++) s -= A[(size_t)k * n + i] * b[k]; b[i] = s; }
  return 1;
}

void hso_or_ba_optimize(hso_se3* poses, const uint8_t* pose_fixed, int n_poses, double* idist, int n_points,
                        const hso_ba_edge* edges, int n_edges, double huber_corner, double huber_edge, int n_iter,
                        double* edge_chi2_out, hso_ba_result* res)
{
  int* col = (int*)malloc(sizeof(int) * (size_t)n_poses);
  int n_free = 0;
  for (int i = 0; i < n_poses; i++) col[i] = pose_fixed[i] ? -1 : n_points + 6 * n_free++;
  const int N = n_points + 6 * n_free;
  double* Hpp = (double*)malloc(sizeof(double) * (size_t)n_points);
  double* bp = (double*)malloc(sizeof(double) * (size_t)n_points);
  double* Hpc = (double*)malloc(sizeof(double) * (size_t)n_points * n_poses * 6);
  double* Hcc = (double*)malloc(sizeof(double) * (size_t)n_poses * n_poses * 36);
  double* bc = (double*)malloc(sizeof(double) * (size_t)n_poses * 6);
  double* eerr = (double*)malloc(sizeof(double) * 2 * (size_t)n_edges);
  double* echi = (double*)malloc(sizeof(double) * (size_t)n_edges);
  double* A = (double*)malloc(sizeof(double) * (size_t)N * N);
  double* Al = (double*)malloc(sizeof(double) * (size_t)N * N);
  double* b = (double*)malloc(sizeof(double) * (size_t)N);
  double* x = (double*)malloc(sizeof(double) * (size_t)N);
  hso_se3* poses_bak = (hso_se3*)malloc(sizeof(hso_se3) * (size_t)n_poses);
  double* idist_bak = (double*)malloc(sizeof(double) * (size_t)n_points);
  double chi[2];
  memset(res, 0, sizeof(*res));

  /* runSparseBAOptimizer: computeActiveErrors(); init_error = activeChi2() */
  hso_or_ba_linearize(poses, pose_fixed, n_poses, idist, n_points, edges, n_edges, huber_corner, huber_edge,
                      Hpp, bp, Hpc, Hcc, bc, eerr, echi, chi);
  res->init_chi2 = chi[0];
  double lambda = -1., ni = 2.;
  int nBad = 0;
  int stop = 0;
  for (int it = 0; it < n_iter; it++) {
    /* solve(): computeActiveErrors, currentChi = activeRobustChi2, buildSystem */
    hso_or_ba_linearize(poses, pose_fixed, n_poses, idist, n_points, edges, n_edges, huber_corner, huber_edge,
                        Hpp, bp, Hpc, Hcc, bc, eerr, echi, chi);
    double currentChi = chi[1], tempChi = currentChi;
    const double iniChi = currentChi;
    memset(A, 0, sizeof(double) * (size_t)N * N);
    for (int p = 0; p < n_points; p++) {
      A[(size_t)p * N + p] = Hpp[p]; b[p] = bp[p];
      for (int c = 0; c < n_poses; c++) {
        if (col[c] < 0) continue;
        for (int q = 0; q < 6; q++) {
          const double v = Hpc[((size_t)p * n_poses + c) * 6 + q];
          A[(size_t)p * N + col[c] + q] = v; A[(size_t)(col[c] + q) * N + p] = v;
        }
      }
    }
    for (int i = 0; i < n_poses; i++) {
      if (col[i] < 0) continue;
      for (int q = 0; q < 6; q++) b[col[i] + q] = bc[i * 6 + q];
      for (int j = i; j < n_poses; j++) {
        if (col[j] < 0) continue;
        for (int r = 0; r < 6; r++)
          for (int c = 0; c < 6; c++) {
            const double v = Hcc[((size_t)i * n_poses + j) * 36 + r * 6 + c];
            A[(size_t)(col[i] + r) * N + col[j] + c] = v;
            A[(size_t)(col[j] + c) * N + col[i] + r] = v;
          }
      }
    }
    if (it == 0) {   /* computeLambdaInit: tau * max |diagonal| */
      double maxDiagonal = 0.;
      for (int k = 0; k < N; k++) { const double d = fabs(A[(size_t)k * N + k]); if (d > maxDiagonal) maxDiagonal = d; }
      lambda = 1e-5 * maxDiagonal;
      ni = 2; nBad = 0;
    }
    double rho = 0;
    int qmax = 0;
    do {
      memcpy(poses_bak, poses, sizeof(hso_se3) * (size_t)n_poses);      /* _optimizer->push() */
      memcpy(idist_bak, idist, sizeof(double) * (size_t)n_points);
      memcpy(Al, A, sizeof(double) * (size_t)N * N);
      for (int k = 0; k < N; k++) Al[(size_t)k * N + k] += lambda;          /* setLambda(_currentLambda, true) */
      memcpy(x, b, sizeof(double) * (size_t)N);
      const int ok2 = dense_ldlt_solve(Al, x, N);
      res->n_solves++;
      /* _optimizer->update(x): VertexSBAPointID::oplusImpl, VertexSE3Expmap::oplusImpl */
      for (int p = 0; p < n_points; p++) idist[p] += x[p];
      for (int i = 0; i < n_poses; i++) {
        if (col[i] < 0) continue;
        hso_se3 d, nw;
        hso_or_se3quat_exp(&x[col[i]], &d);
        hso_or_se3quat_mul(&d, &poses[i], &nw);
        poses[i] = nw;
      }
      /* computeActiveErrors; tempChi = activeRobustChi2 */
      hso_or_ba_linearize(poses, pose_fixed, n_poses, idist, n_points, edges, n_edges, huber_corner, huber_edge,
                          Hpp, bp, Hpc, Hcc, bc, eerr, echi, chi);
      tempChi = chi[1];
      if (!ok2) tempChi = 1.7976931348623157e308;
      rho = (currentChi - tempChi);
      double scale = 0.;                                                  /* computeScale */
      for (int j = 0; j < N; j++) scale += x[j] * (lambda * x[j] + b[j]);
      scale += 1e-3;
      rho /= scale;
      if (rho > 0 && isfinite(tempChi)) {
        double alpha = 1. - pow((2 * rho - 1), 3);
        alpha = alpha < 2. / 3. ? alpha : 2. / 3.;
        const double scaleFactor = (1. / 3.) > alpha ? (1. / 3.) : alpha;
        lambda *= scaleFactor;
        ni = 2;
        currentChi = tempChi;
        res->n_accepted++;
      } else {
        lambda *= ni;
        ni *= 2;
        memcpy(poses, poses_bak, sizeof(hso_se3) * (size_t)n_poses);      /* _optimizer->pop() */
        memcpy(idist, idist_bak, sizeof(double) * (size_t)n_points);
      }
      qmax++;
    } while (rho < 0 && qmax < 5);                                       /* setMaxTrialsAfterFailure(5), bundle_adjustment.cpp:571 */
    res->iterations = it + 1;
    res->robust_chi2 = currentChi;
    if (qmax == 5 || rho == 0) { stop = 1; break; }
    if ((iniChi - currentChi) * 1e3 < iniChi) nBad++; else nBad = 0;     /* "Stop criterium (Raul)" */
    if (nBad >= 3) { stop = 2; break; }
  }
  res->stop = stop;
  res->lambda = lambda;
  /* final_error = activeChi2(): the errors of the last computeActiveErrors, wherever that was */
  res->final_chi2 = chi[0];
  if (n_iter <= 0) res->robust_chi2 = chi[1];
  if (edge_chi2_out) memcpy(edge_chi2_out, echi, sizeof(double) * (size_t)n_edges);
  free(col); free(Hpp); free(bp); free(Hpc); free(Hcc); free(bc); free(eerr); free(echi); free(A); free(Al); free(b); free(x);
  free(poses_bak); free(idist_bak);
}
