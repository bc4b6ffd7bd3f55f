// hso_match_dev.h — device-side Matcher::findMatchDirect / findMatchSeed body shared by the
// reprojection-matching kernel (hso_align.hip) and the seed-activation kernels (hso_activate.hip).
// A DPP row of 16 lanes executes match_patch() for one candidate (four pixels of the 8x8 patch per lane), so a wavefront
// matches four candidates at once; match_one() is the one-candidate-per-wave form (its four rows do the same candidate).
// See hso_align.hip for the reference citations of each step.
#pragma once
#include "hso_ctx.h"
#include "hso_dev_math.h"

namespace hso_dev {


HSO_DEV float wave_sum_all(float v) { return wave_butterfly_sum(v); }

// AbstractCamera::cam2world, src/camera.cpp:67-87 (pinhole; radtan through the 5-iteration
// cv::undistortPoints with float K, D and float I/O, :43-45,78-85), :171-194 (FOV)
HSO_DEV void cam2world_dev(const hso_camera& cam, double u, double v, double f[3])
{
  double x, y;
  if (cam.model == HSO_CAM_PINHOLE && cam.distortion) {
    const double fx = (float)cam.fx, fy = (float)cam.fy, cx = (float)cam.cx, cy = (float)cam.cy;
    const double k0 = (float)cam.d[0], k1 = (float)cam.d[1], p1 = (float)cam.d[2], p2 = (float)cam.d[3], k2 = (float)cam.d[4];
    const double ifx = 1. / fx, ify = 1. / fy;
    x = (float)u; y = (float)v;
    const double x0 = x = (x - cx) * ifx;
    const double y0 = y = (y - cy) * ify;
    for (int it = 0; it < 5; it++) {
      const double r2 = x * x + y * y;
      const double icdist = (1 + ((0 * r2 + 0) * r2 + 0) * r2) / (1 + ((k2 * r2 + k1) * r2 + k0) * r2);
      const double deltaX = 2 * p1 * x * y + p2 * (r2 + 2 * x * x);
      const double deltaY = p1 * (r2 + 2 * y * y) + 2 * p2 * x * y;
      x = (x0 - deltaX) * icdist;
      y = (y0 - deltaY) * icdist;
    }
    x = (float)x; y = (float)y;
  } else if (cam.model == HSO_CAM_FOV && cam.distortion) {
    const double omega = cam.d[0];
    const double ud = (u - cam.cx) / cam.fx, vd = (v - cam.cy) / cam.fy;
    const double dist = sqrt(ud * ud + vd * vd);
    const double rd = tan(dist * omega) / (2 * dist * tan(omega / 2));
    x = rd * ud; y = rd * vd;
  } else {
    x = (u - cam.cx) / cam.fx; y = (v - cam.cy) / cam.fy;
  }
  const double n = sqrt(x * x + y * y + 1.0);
  f[0] = x / n; f[1] = y / n; f[2] = 1.0 / n;
}

// two horizontally adjacent pixels by ONE unaligned 16-bit load (low byte = p[0]): half the memory instructions of the
// bilinear taps in the latency-bound per-candidate loops
typedef uint16_t __attribute__((aligned(1))) u16_unaligned;
// (the images live in device memory: say so, or the access is a FLAT load that has to resolve the aperture first)
HSO_DEV unsigned load_px_pair(const uint8_t* p) { return *(const __attribute__((address_space(1))) u16_unaligned*)p; }

// ---- sixteen lanes per patch: a DPP row owns an 8x8 patch, a lane four horizontally adjacent pixels of it (pixel (px0 + j, py),
// j = 0..3, of lane l16 = lane & 15: py = l16 >> 1, px0 = (l16 & 1) * 4).  Four patches per wavefront, sums inside the row.
HSO_DEV float row_sum_all(float v)
{
  v += __uint_as_float((unsigned)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(v), 0x128, 0xf, 0xf, false));   // row_ror:8
  v += __uint_as_float((unsigned)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(v), 0x124, 0xf, 0xf, false));   // row_ror:4
  v += __uint_as_float((unsigned)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(v), 0x4e, 0xf, 0xf, false));    // quad_perm:[2,3,0,1]
  v += __uint_as_float((unsigned)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(v), 0xb1, 0xf, 0xf, false));    // quad_perm:[1,0,3,2]
  return v;   // every lane of the row holds the same bits (each step adds the same two partial sums in both partners)
}
HSO_DEV float row_sum4(const float (&v)[4]) { return row_sum_all((v[0] + v[1]) + (v[2] + v[3])); }

// the 5 + 5 bytes a lane's four bilinear samples need: two unaligned 8-byte loads (the rows are followed by at least one
// padded row + 64 bytes, so the three bytes read beyond the fifth stay inside the frame allocation)
typedef unsigned long long __attribute__((aligned(1))) u64_unaligned;
HSO_DEV unsigned long long load_px8(const uint8_t* p) { return *(const __attribute__((address_space(1))) u64_unaligned*)p; }
HSO_DEV float byte_f(unsigned long long w, int j) { return (float)(unsigned)((w >> (8 * j)) & 0xffull); }

// hso::interpolateMat_8u, include/hso/vikit/vision.h:49-65
HSO_DEV float interpolate_8u(const uint8_t* data, int stride, float u, float v)
{
  const int x = (int)floor((double)u);
  const int y = (int)floor((double)v);
  const float sx = u - (float)x, sy = v - (float)y;
  const float w00 = (1.0f - sx) * (1.0f - sy);
  const float w01 = (1.0f - sx) * sy;
  const float w10 = sx * (1.0f - sy);
  const float w11 = ((1.0f - w00) - w01) - w10;
  const uint8_t* p = data + y * stride + x;
  const unsigned r0 = load_px_pair(p), r1 = load_px_pair(p + stride);
  return ((w00 * (float)(r0 & 0xffu) + w01 * (float)(r1 & 0xffu)) + w10 * (float)(r0 >> 8)) + w11 * (float)(r1 >> 8);
}

// The part of findMatchDirect that is the same in all 64 lanes of the wavefront that matches a candidate — the border test of
// the reference position, warp::getWarpMatrixAffine (two cam2world, three rigid transforms, three projections: ~700 fp64
// instructions, nearly half of a candidate's instruction count) and warp::getBestSearchLevel.  Callable by ONE lane per
// candidate: the batched kernels compute it for 64 candidates at a time, one per lane, before they walk the candidates with
// the whole wave (hso_align.hip).
struct MatchGeom {
  double A00, A01, A10, A11;     // A_cur_ref
  int32_t search_level;
  int32_t ref_border;            // != 0: the reference position fails isInFrame (HSO_ALIGN_REF_BORDER)
};

HSO_DEV MatchGeom match_geometry(const hso_camera& cam, const PyrGeom& g, const hso_align_job& J)
{
  MatchGeom G;
  G.A00 = G.A01 = G.A10 = G.A11 = 0; G.search_level = 0; G.ref_border = 0;
  const int W = g.w[0], H = g.h[0];
  const int halfpatch_size_ = 4;
  // isInFrame((px/(1<<level)).cast<int>(), halfpatch_size_+2, level), matcher.cpp:288, camera.h:85-89
  {
    const int L = J.ref_level, b = halfpatch_size_ + 2;
    const int ox = (int)(J.px_ref[0] / (double)(1 << L)), oy = (int)(J.px_ref[1] / (double)(1 << L));
    if (!(ox >= b && ox < W / (1 << L) - b && oy >= b && oy < H / (1 << L) - b)) { G.ref_border = 1; return G; }
  }
  // ---- warp::getWarpMatrixAffine, matcher.cpp:46-72
  const Se3 T = se3_from(J.T_cur_ref);
  double A00, A01, A10, A11;
  {
    const int hp = 5;
    const double xr = J.f_ref[0] * J.depth, yr = J.f_ref[1] * J.depth, zr = J.f_ref[2] * J.depth;
    const int ratio = 1 << J.ref_level;
    double du[3], dv[3];
    cam2world_dev(cam, J.px_ref[0] + (double)(hp * ratio), J.px_ref[1] + (double)(0 * ratio), du);
    cam2world_dev(cam, J.px_ref[0] + (double)(0 * ratio), J.px_ref[1] + (double)(hp * ratio), dv);
    const double su = zr / du[2], sv = zr / dv[2];
    for (int i = 0; i < 3; i++) { du[i] *= su; dv[i] *= sv; }
    double cx, cy, cz, ux, uy, uz, vx, vy, vz;
    se3_apply(T, xr, yr, zr, cx, cy, cz);
    se3_apply(T, du[0], du[1], du[2], ux, uy, uz);
    se3_apply(T, dv[0], dv[1], dv[2], vx, vy, vz);
    double pc0, pc1, pu0, pu1, pv0, pv1;
    world2cam(cam, cx, cy, cz, pc0, pc1);
    world2cam(cam, ux, uy, uz, pu0, pu1);
    world2cam(cam, vx, vy, vz, pv0, pv1);
    A00 = (pu0 - pc0) / hp; A10 = (pu1 - pc1) / hp;
    A01 = (pv0 - pc0) / hp; A11 = (pv1 - pc1) / hp;
  }
  G.A00 = A00; G.A01 = A01; G.A10 = A10; G.A11 = A11;
  // ---- warp::getBestSearchLevel, :74-85 (max_level = Config::nPyrLevels()-1 = 2)
  int search_level = 0;
  {
    double D = A00 * A11 - A10 * A01;
    while (D > 3.0 && search_level < HSO_N_SOBEL_LEVELS - 1) { search_level += 1; D *= 0.25; }
  }
  G.search_level = search_level;
  return G;
}

// Matcher::findMatchDirect after the reference feature has been chosen (src/matcher.cpp:286-375), given the candidate's
// geometry; findMatchSeed (:442-518) is the same body with ncc_thresh = 0.8 and J.kf_gap_lt4 = 1.
// One DPP row of 16 lanes, four pixels of the 8x8 patch per lane.  pwb_lds: 100 floats of LDS private to this row.
// The wave-uniform arithmetic of the LK loop (weights, the 3x3 update, the convergence test) used to serve one candidate per
// wave-instruction — k_align_t ran at 100 % VALU busy with ~1100 wave-instructions per candidate, most of them uniform
// (profiles/r3_stage_sq_align.csv); now it serves four, and a patch sum is three adds + four DPP steps.
HSO_DEV hso_align_out match_patch(const PyrGeom& g, const uint8_t* cur_base, const uint8_t* ref_base,
                                  const hso_align_job& J, const MatchGeom& G, double ncc_thresh, float* pwb_lds)
{
  const int l16 = threadIdx.x & 15;
  hso_align_out o;
  o.success = 0; o.stage = HSO_ALIGN_OK; o.search_level = 0; o.iters = 0;
  o.px_cur[0] = J.px_cur[0]; o.px_cur[1] = J.px_cur[1];
  o.A_cur_ref[0] = o.A_cur_ref[1] = o.A_cur_ref[2] = o.A_cur_ref[3] = 0; o.h_inv = 0; o.ncc = 0; o.chi2 = 0;
  const int halfpatch_size_ = 4;
  if (G.ref_border) { o.stage = HSO_ALIGN_REF_BORDER; return o; }
  const double A00 = G.A00, A01 = G.A01, A10 = G.A10, A11 = G.A11;
  o.A_cur_ref[0] = A00; o.A_cur_ref[1] = A01; o.A_cur_ref[2] = A10; o.A_cur_ref[3] = A11;
  const int search_level = G.search_level;
  o.search_level = search_level;

  // ---- warp::warpAffine (float), :120-155: 10x10 samples of the reference level
  {
    const double det = A00 * A11 - A10 * A01;
    const double invdet = 1.0 / det;
    const float a00 = (float)(A11 * invdet), a01 = (float)(-A01 * invdet);
    const float a10 = (float)(-A10 * invdet), a11 = (float)(A00 * invdet);
    const bool warp_nan = isnan(a00);  // reference: patch left untouched (uninitialised); defined as 0 here
    const int L = J.ref_level;
    const int cols = g.w[L], rows = g.h[L];  // img_pyr_[L].cols / rows
    const uint8_t* img = ref_base + g.off[L];
    const float rx = (float)(J.px_ref[0] / (double)(1 << L)), ry = (float)(J.px_ref[1] / (double)(1 << L));
    const float scaleTarget = (float)(1 << search_level);
    const bool scale_exposure = J.kf_gap_lt4 && fabsf(J.exposure_rat * 128 - 128) > 30.0f;  // :317-320, LIGHT_THRESHOLD
    for (int idx = l16; idx < 100; idx += 16) {
      const int y = idx / 10, x = idx - 10 * y;
      float p0 = (float)(x - 5), p1 = (float)(y - 5);
      p0 *= scaleTarget; p1 *= scaleTarget;
      const float px0 = (a00 * p0 + a01 * p1) + rx;
      const float px1 = (a10 * p0 + a11 * p1) + ry;
      float val = 0;
      if (!warp_nan && !(px0 < 0 || px1 < 0 || px0 >= (float)(cols - 1) || px1 >= (float)(rows - 1)))
        val = interpolate_8u(img, cols, px0, px1);
      if (scale_exposure) val = val * J.exposure_rat;
      pwb_lds[idx] = val;
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");

  // ---- template gradients, weights, Hessian (align2D :483-512 / align1D :183-207)
  const int px0_ = (l16 & 1) * 4, py_ = l16 >> 1;
  const float* pwb = pwb_lds;
  float ref_px[4], gxr[4], gyr[4];
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const int c = (py_ + 1) * 10 + px0_ + j + 1;
    ref_px[j] = pwb[c];
    gxr[j] = pwb[c + 1] - pwb[c - 1]; gyr[j] = pwb[c + 10] - pwb[c - 10];
  }
  const bool edgelet = (J.type == HSO_FTR_EDGELET);
  double dir0 = 0, dir1 = 0;
  float dirf0 = 0, dirf1 = 0;
  if (edgelet) {
    dir0 = A00 * J.grad[0] + A01 * J.grad[1];
    dir1 = A10 * J.grad[0] + A11 * J.grad[1];
    const double dn = sqrt(dir0 * dir0 + dir1 * dir1);
    dir0 /= dn; dir1 /= dn;
    dirf0 = (float)dir0; dirf1 = (float)dir1;
  }
  float Jx[4], Jy[4], wgt[4];  // align2D: (dx, dy); align1D: (dv, unused)
#pragma unroll
  for (int j = 0; j < 4; j++) {
    if (edgelet) { Jx[j] = (float)(0.5 * (double)(dirf0 * gxr[j] + dirf1 * gyr[j])); Jy[j] = 0; }
    else { Jx[j] = (float)(0.5 * (double)gxr[j]); Jy[j] = (float)(0.5 * (double)gyr[j]); }
    wgt[j] = edgelet ? sqrtf((float)(250.0 / (250.0 + (double)(Jx[j] * Jx[j]))))
                     : sqrtf((float)(250.0 / (250.0 + (double)(Jx[j] * Jx[j] + Jy[j] * Jy[j]))));
  }
  float Hi[9];  // align2D: 3x3; align1D: [0],[1],[3],[4] used as 2x2 over (dv, 1)
  float t0[4], t1[4], t2[4];
#pragma unroll
  for (int j = 0; j < 4; j++) { t0[j] = (Jx[j] * Jx[j]) * wgt[j]; t1[j] = (Jx[j] * 1.0f) * wgt[j]; t2[j] = (1.0f * 1.0f) * wgt[j]; }
  float h_xx = row_sum4(t0), h_x1 = row_sum4(t1), h_11 = row_sum4(t2);
  float h_xy = 0, h_yy = 0, h_y1 = 0;
  if (!edgelet) {
#pragma unroll
    for (int j = 0; j < 4; j++) { t0[j] = (Jx[j] * Jy[j]) * wgt[j]; t1[j] = (Jy[j] * Jy[j]) * wgt[j]; t2[j] = (Jy[j] * 1.0f) * wgt[j]; }
    h_xy = row_sum4(t0); h_yy = row_sum4(t1); h_y1 = row_sum4(t2);
  }
  if (edgelet) {
    float H00 = h_xx, H01 = h_x1, H11 = h_11;
    H00 = (float)((double)H00 * (1 + 0.001)); H11 = (float)((double)H11 * (1 + 0.001));
    o.h_inv = 1.0 / (double)H00 * 8 * 8;  // :207
    const float det = H00 * H11 - H01 * H01;
    const float invdet = 1.0f / det;
    Hi[0] = H11 * invdet; Hi[1] = -H01 * invdet; Hi[3] = -H01 * invdet; Hi[4] = H00 * invdet;
  } else {
    float H0 = h_xx, H1 = h_xy, H2 = h_x1, H4 = h_yy, H5 = h_y1, H8 = h_11;
    H0 = (float)((double)H0 * (1 + 0.001)); H4 = (float)((double)H4 * (1 + 0.001)); H8 = (float)((double)H8 * (1 + 0.001));
    const float H3 = H1, H6 = H2, H7 = H5;
    const float c00 = H4 * H8 - H5 * H7, c01 = H5 * H6 - H3 * H8, c02 = H3 * H7 - H4 * H6;
    const float det = H0 * c00 + H1 * c01 + H2 * c02;
    const float invdet = 1.0f / det;
    Hi[0] = c00 * invdet; Hi[3] = c01 * invdet; Hi[6] = c02 * invdet;
    Hi[1] = (H2 * H7 - H1 * H8) * invdet; Hi[4] = (H0 * H8 - H2 * H6) * invdet; Hi[7] = (H1 * H6 - H0 * H7) * invdet;
    Hi[2] = (H1 * H5 - H2 * H4) * invdet; Hi[5] = (H2 * H3 - H0 * H5) * invdet; Hi[8] = (H0 * H4 - H1 * H3) * invdet;
  }

  // ---- LK iterations (align2D :526-598 / align1D :214-301)
  const int cols = g.w[search_level], rows = g.h[search_level];
  const uint8_t* cur = cur_base + g.off[search_level];
  double pxs0 = J.px_cur[0] / (double)(1 << search_level), pxs1 = J.px_cur[1] / (double)(1 << search_level);
  const double orig0 = pxs0, orig1 = pxs1;
  float u = (float)pxs0, v = (float)pxs1;
  const float min_update_squared = edgelet ? (float)(0.01 * 0.01) : (float)(0.03 * 0.03);
  float mean_diff = 0, chi2 = 0;
  float search_pixel[4] = { 0, 0, 0, 0 };
  bool converged = false, nan_exit = false;
  int iter = 0;
  for (iter = 0; iter < 10; ++iter) {  // options_.align_max_iter, matcher.h:124
    const int u_r = (int)floor((double)u), v_r = (int)floor((double)v);
    if (u_r < halfpatch_size_ || v_r < halfpatch_size_ || u_r >= cols - halfpatch_size_ || v_r >= rows - halfpatch_size_) break;
    if (isnan(u) || isnan(v)) { nan_exit = true; break; }
    const float sx = u - (float)u_r, sy = v - (float)v_r;
    const float wTL = (float)((1.0 - sx) * (1.0 - sy));
    const float wTR = (float)(sx * (1.0 - sy));
    const float wBL = (float)((1.0 - sx) * sy);
    const float wBR = sx * sy;
    const uint8_t* it = cur + (v_r + py_ - halfpatch_size_) * cols + u_r - halfpatch_size_ + px0_;
    const unsigned long long it0 = load_px8(it), it1 = load_px8(it + cols);
    float a0[4], a1[4], a2[4], a3[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
      search_pixel[j] = ((wTL * byte_f(it0, j) + wTR * byte_f(it0, j + 1)) + wBL * byte_f(it1, j)) + wBR * byte_f(it1, j + 1);
      const float res = (search_pixel[j] - ref_px[j]) + mean_diff;
      a0[j] = (res * Jx[j]) * wgt[j]; a1[j] = res * wgt[j]; a2[j] = (res * res) * wgt[j]; a3[j] = (res * Jy[j]) * wgt[j];
    }
    const float j0 = row_sum4(a0);
    const float j2 = row_sum4(a1);
    chi2 = row_sum4(a2);
    if (edgelet) {
      const float J0 = -j0, J1 = -j2;
      const float up0 = Hi[0] * J0 + Hi[1] * J1, up1 = Hi[3] * J0 + Hi[4] * J1;
      u += up0 * dirf0; v += up0 * dirf1; mean_diff += up1;
      if (up0 * up0 < min_update_squared) { converged = true; iter++; break; }
    } else {
      const float j1 = row_sum4(a3);
      const float J0 = -j0, J1 = -j1, J2 = -j2;
      const float up0 = (Hi[0] * J0 + Hi[1] * J1) + Hi[2] * J2;
      const float up1 = (Hi[3] * J0 + Hi[4] * J1) + Hi[5] * J2;
      const float up2 = (Hi[6] * J0 + Hi[7] * J1) + Hi[8] * J2;
      u += up0; v += up1; mean_diff += up2;
      if (up0 * up0 + up1 * up1 < min_update_squared) { converged = true; iter++; break; }
    }
  }
  if (chi2 > (float)(1000 * 64)) converged = false;
  if (!nan_exit) { pxs0 = (double)u; pxs1 = (double)v; }  // `return false` at :230/:537 skips the write-back
  o.iters = iter; o.chi2 = chi2;
  bool ok = converged && !nan_exit;
  if (!ok) o.stage = HSO_ALIGN_NOT_CONVERGED;

  // ---- Matcher::checkNormal, :406-440 (edgelets only)
  if (ok && edgelet) {
    const int16_t* gx = reinterpret_cast<const int16_t*>(cur_base + g.sob_off[search_level][0]);
    const int16_t* gy = reinterpret_cast<const int16_t*>(cur_base + g.sob_off[search_level][1]);
    const float uf = (float)pxs0, vf = (float)pxs1;
    // (an LK result is inside the level with a 4-pixel margin by its own loop test, :234; the guard only keeps the read defined)
    const bool inside = uf >= 0 && vf >= 0 && uf < (float)(cols - 1) && vf < (float)(rows - 1);
    const int ui = inside ? (int)floorf((float)pxs0) : 0, vi = inside ? (int)floorf((float)pxs1) : 0;
    const float sx = uf - (float)ui, sy = vf - (float)vi;
    const float wTL = (float)((1.0 - sx) * (1.0 - sy));
    const float wTR = (float)(sx * (1.0 - sy));
    const float wBL = (float)((1.0 - sx) * sy);
    const float wBR = (float)(((1.0 - wTL) - wTR) - wBL);
    const int gs = g.sob_stride[search_level];   // gradient rows are padded (hso_ctx.h)
    const int a = vi * gs + ui;
    double n0 = (((double)wTL * (double)gx[a] + (double)wTR * (double)gx[a + 1]) + (double)wBL * (double)gx[a + gs]) + (double)wBR * (double)gx[a + gs + 1];
    double n1 = (((double)wTL * (double)gy[a] + (double)wTR * (double)gy[a + 1]) + (double)wBL * (double)gy[a + gs]) + (double)wBR * (double)gy[a + gs + 1];
    const double nn = sqrt(n0 * n0 + n1 * n1);
    n0 /= nn; n1 /= nn;
    ok = inside && (dir0 * n0 + dir1 * n1) > (double)(float)0.86;  // Config::edgeLetCosAngle() through a float parameter
    if (!ok) o.stage = HSO_ALIGN_NORMAL;
  }

  // ---- Matcher::checkNCC, :379-404 on (ref patch, last iteration's samples)
  {
    const float mean1 = row_sum4(ref_px) / 64, mean2 = row_sum4(search_pixel) / 64;
    float qq[4], q11[4], q22[4];
#pragma unroll
    for (int j = 0; j < 4; j++) { const float d1 = ref_px[j] - mean1, d2 = search_pixel[j] - mean2; qq[j] = d1 * d2; q11[j] = d1 * d1; q22[j] = d2 * d2; }
    const float num = row_sum4(qq), den1 = row_sum4(q11), den2 = row_sum4(q22);
    const double ncc = (double)num / ((double)sqrtf(den1 * den2) + 1e-12);
    o.ncc = (float)ncc;
    if (ok) {
      ok = ncc > ncc_thresh;
      if (!ok) o.stage = HSO_ALIGN_NCC;
    }
  }
  if (ok) {
    const double dx = orig0 - pxs0, dy = orig1 - pxs1;
    ok = sqrt(dx * dx + dy * dy) < 20;  // :369-370
    if (!ok) o.stage = HSO_ALIGN_JUMP;
  }
  o.px_cur[0] = pxs0 * (double)(1 << search_level);
  o.px_cur[1] = pxs1 * (double)(1 << search_level);
  o.success = ok ? 1 : 0;
  return o;
}

// geometry by every lane + patch: for the kernels that handle one candidate per wave (seed activation, seed reprojection); the
// four rows of the wave match the same candidate (identical results; their patch writes coincide)
HSO_DEV hso_align_out match_one(const hso_camera& cam, const PyrGeom& g, const uint8_t* cur_base, const uint8_t* ref_base,
                                const hso_align_job& J, double ncc_thresh, float* pwb_lds)
{
  const MatchGeom G = match_geometry(cam, g, J);
  return match_patch(g, cur_base, ref_base, J, G, ncc_thresh, pwb_lds);
}

}  // namespace hso_dev
