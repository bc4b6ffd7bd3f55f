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
