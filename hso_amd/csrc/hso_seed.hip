// hso_seed.hip — depth-filter seed observation on gfx950: epipolar ZMNCC search, step-limited
// KLT refinement, triangulation and the Gaussian inverse-depth update.
//
// Replaces DepthFilter::observeDepthRow (reference src/depth_filter.cpp:580-675) with
// updateSeed :527-537 and computeTau :539-555, and Matcher::doLineStereo
// (src/matcher.cpp:802-1049) with KLTLimited2D :1296-1451, KLTLimited1D :1454-1606,
// warp::createPatch :159-196, ZMNCC_F (include/hso/vikit/patch_score.h:268-305),
// checkNormal :406-440, checkNCC :379-404 and depthFromTriangulation :242-255.
//
// MI355X mapping: the reference spreads seeds over 4 CPU threads (IndexThreadReduce, MAPPING_THREADS 4).  Here an observation
// is three kernels (k_seed_pre / k_seed_image / k_seed_post below): the fp64 geometry before and after the image work runs one
// THREAD per seed; the image work of a seed (the 8x8 patch: createPatch, the epipolar ZMNCC march, the two KLT refinements, the
// checks) runs on eight lanes, eight seeds per wavefront.  The march's steps are independent, so a lane evaluates a whole step
// serially — in the reference's summation order, the scores are bit-identical to the CPU arithmetic — and the 64 lanes of the
// wave share the steps of its eight seeds; the KLT iterations are sequential, so there the eight lanes share an iteration
// (packed fp32 for the per-pixel arithmetic, a patch sum = a fixed in-lane tree + three DPP steps; rounding-level differences
// from the reference's serial sums, excused by margin in the tests).  Throughput comes from thousands of seeds per keyframe x
// many sequences in flight.  Round-4 counters: profiles/r4_stage_sq_seed_k_seed_image.csv (1 866 -> 1 006 VALU instructions
// per seed against profiles/r3_stage_sq_seed.csv).
#include "hso_match_dev.h"
#include <string.h>
#include <algorithm>
#include <unordered_map>
#include <vector>

using namespace hso_dev;

#define SEED_WAVES_PER_BLOCK 2

// the active frame a seed is observed in (seeds of many frames / sequences share a launch)
struct SeedConsts {
  hso_camera cam;
  PyrGeom g;
  const SeedFrameDev* frames;
  double px_error_angle;
  int update_in_place;       // resident tables: write mu / sigma2 / b back into the seed record
  hso_seed_brief* brief;     // resident tables: compact per-slot result (may be null)
  float* px;                 // resident tables: Matcher::px_cur_ of the matched seeds, two floats per slot (may be null)
  // previous-frame pass (observeDepthWithPreviousFrameOnce): frames[k] is the earlier frame the seeds hosted in keyframe
  // frame_keys[k] (ascending) observe; seeds of other keyframes sit the call out.  Null: the ordinary pass (frames by group).
  const int64_t* frame_keys;
  int n_frame_keys;
};

struct SeedDev {
  const uint8_t* ref_base;
  const uint8_t* cur_base;
  int32_t frame;             // index into SeedConsts::frames
  int32_t pad_;
  hso_seed s;
};

// row_sum_all / row_sum4, load_px8 / byte_f, cam2world_dev and interpolate_8u come from hso_match_dev.h (shared with the matcher)

// ---- eight lanes per seed ------------------------------------------------------------------------------------------------------
// The image phase of a seed works on an 8x8 patch.  Round 3 gave a DPP row of 16 lanes to a seed (four pixels per lane); the SQ
// counters (profiles/r3_stage_sq_seed.csv: VALU saturated, 1 870 wave-instructions per seed) and the ISA showed where those
// went: of the march's 218 instructions per step ~110 are the same in all 16 lanes (position, bounds, bilinear weights, the
// fp64 ZMNCC quotient, the best / second bookkeeping) and 21 are DPP row sums, so a step cost 54 instructions per seed of which
// 12 touch pixels.  Now a group of EIGHT lanes owns a seed (eight seeds per wavefront, lane l8 = lane & 7 owns row l8 of the
// patch), and the two loops are laid out for what they are:
//   * the epipolar march has independent steps, so a LANE evaluates a whole step — all 64 bilinear samples and the three ZMNCC
//     sums, serially, in the reference's own summation order (patch_score.h:268-305; the score is now bit-identical to the
//     serial CPU arithmetic, no tree sums) — eight steps of a seed at a time; nothing in a step is redundant any more
//     (~17 wave-instructions per seed and step);
//   * the KLT iterations depend on one another, so there the eight lanes share an iteration (eight pixels each, a patch sum is
//     seven adds and three DPP steps inside the group) and the uniform arithmetic of an iteration serves eight seeds.
HSO_DEV float grp_sum_all(float v)
{
  v += __uint_as_float((unsigned)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(v), 0x141, 0xf, 0xf, false));   // row_half_mirror: l8 <-> 7 - l8
  v += __uint_as_float((unsigned)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(v), 0xb1, 0xf, 0xf, false));    // quad_perm:[1,0,3,2]
  v += __uint_as_float((unsigned)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(v), 0x4e, 0xf, 0xf, false));    // quad_perm:[2,3,0,1]
  return v;   // the same bits in all eight lanes: every step adds the same two partial sums in both partners
}
HSO_DEV float grp_sum8(const float (&v)[8]) { return grp_sum_all(((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]))); }
// the value lane `j` of the caller's group of eight holds
HSO_DEV float grp_get(float v, int j) { return __uint_as_float((unsigned)__builtin_amdgcn_ds_bpermute((int)((((threadIdx.x & 63) & ~7) | j) << 2), (int)__float_as_uint(v))); }
HSO_DEV int grp_get(int v, int j) { return __builtin_amdgcn_ds_bpermute((int)((((threadIdx.x & 63) & ~7) | j) << 2), v); }
HSO_DEV double grp_get(double v, int j)
{
  const unsigned long long b = (unsigned long long)__double_as_longlong(v);
  const unsigned lo = (unsigned)grp_get((int)(unsigned)b, j), hi = (unsigned)grp_get((int)(unsigned)(b >> 32), j);
  return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

// twelve bytes from an unaligned address in ONE load (the three beyond the ninth stay inside the frame allocation, like load_px8's)
struct Px12 { unsigned long long lo; unsigned hi; };
typedef unsigned __attribute__((ext_vector_type(3), aligned(1))) u32x3_unaligned;
HSO_DEV Px12 load_px12(const uint8_t* p)
{
  const u32x3_unaligned v = *(const __attribute__((address_space(1))) u32x3_unaligned*)p;
  Px12 t;
  t.lo = ((unsigned long long)v.y << 32) | v.x; t.hi = v.z;
  return t;
}
// Packed fp32 (v_pk_mul_f32 / v_pk_add_f32: two IEEE fp32 operations per lane and instruction, the rate the MI355X's vector
// fp32 peak is quoted at).  The per-pixel arithmetic of the march and of the KLT iterations is elementwise, so two horizontally
// adjacent pixels share an instruction; each element is rounded exactly as the scalar operation would round it.
typedef float f2 __attribute__((ext_vector_type(2)));
HSO_DEV f2 mk2(float a, float b) { f2 r; r.x = a; r.y = b; return r; }
// nine bytes of an image row as floats, in the two register layouts the eight bilinear samples of a patch row read them in:
// A[k] = (b[2k], b[2k+1]) under the left taps of pixels 2k, 2k+1, S[k] = (b[2k+1], b[2k+2]) under their right taps
struct Row9 { f2 A[4], S[4]; };
HSO_DEV Row9 row9_of(const Px12& t)
{
  Row9 r;
#pragma unroll
  for (int k = 0; k < 4; k++) r.A[k] = mk2(byte_f(t.lo, 2 * k), byte_f(t.lo, 2 * k + 1));
#pragma unroll
  for (int k = 0; k < 3; k++) r.S[k] = mk2(byte_f(t.lo, 2 * k + 1), byte_f(t.lo, 2 * k + 2));
  r.S[3] = mk2(byte_f(t.lo, 7), (float)(t.hi & 0xffu));
  return r;
}
// the eight samples of a patch row: ((w_tl * tl + w_tr * tr) + w_bl * bl) + w_br * br, the reference's expression order
HSO_DEV void bilin_row(float wtl, float wtr, float wbl, float wbr, const Row9& top, const Row9& bot, f2 (&out)[4])
{
#pragma unroll
  for (int k = 0; k < 4; k++) out[k] = ((wtl * top.A[k] + wtr * top.S[k]) + wbl * bot.A[k]) + wbr * bot.S[k];
}
// warp::createPatch's weights (matcher.cpp:159-196).  The reference forms the three products in double from float operands
// and rounds to float; for a position >= 1 px the factors 1 - su, 1 - sv are exact in float (su is a multiple of ulp(u) >=
// 2^-23 below 1) and the double product of two floats is exact, so the float product rounds the same real number: bit-equal.
struct Bilin { float tl, tr, bl, br; };
HSO_DEV Bilin bilin_weights(float su, float sv)
{
  Bilin w;
  const float cu = 1.0f - su, cv = 1.0f - sv;
  w.tl = cu * cv; w.tr = su * cv; w.bl = cu * sv;
  w.br = (float)(((1.0 - (double)w.tl) - (double)w.tr) - (double)w.bl);
  return w;
}

// One step of the epipolar march on ONE lane: createPatch at (u, v) = (float)px of the search level and ZMNCC_F::computeScore against the
// host patch (`hd`: host[i] - hostMean in LDS, `d1` their squared sum), every sum in the reference's order.
HSO_DEV float lane_zmncc(const uint8_t* img, int stride, float u, float v, const float* hd, float d1)
{
  const int ui = (int)floorf(u), vi = (int)floorf(v);
  const Bilin w = bilin_weights(u - (float)ui, v - (float)vi);
  const uint8_t* c = img + (vi - 4) * stride + (ui - 4);
  Px12 raw[9];
#pragma unroll
  for (int r = 0; r < 9; r++) raw[r] = load_px12(c + r * stride);
  f2 sp[32];
  float tmean = 0;
  Row9 top = row9_of(raw[0]);
#pragma unroll
  for (int r = 0; r < 8; r++) {
    const Row9 bot = row9_of(raw[r + 1]);
    f2 row[4];
    bilin_row(w.tl, w.tr, w.bl, w.br, top, bot, row);
#pragma unroll
    for (int k = 0; k < 4; k++) { sp[r * 4 + k] = row[k]; tmean += row[k].x; tmean += row[k].y; }
    top = bot;
  }
  tmean /= 64;
  float num = 0, d2 = 0;
  const f2* hd2 = reinterpret_cast<const f2*>(hd);
#pragma unroll
  for (int i = 0; i < 32; i++) {
    const f2 t = sp[i] - tmean;
    const f2 ht = hd2[i] * t, tt = t * t;
    num += ht.x; num += ht.y;
    d2 += tt.x; d2 += tt.y;
  }
  return (float)((double)num / ((double)sqrtf(d1 * d2) + 1e-12));
}

// Matcher::KLTLimited2D / KLTLimited1D (matcher.cpp:1296-1606), eight lanes per seed, a patch row (eight pixels) per lane.
// ONE_D: motion restricted to `d0,d1` (double, as passed by the reference).  Returns the bool.
// a sum over the 64 pixels of the patch, eight per lane as four pairs: a fixed tree (the reference sums serially)
HSO_DEV float grp_sum8(const f2 (&v)[4])
{
  const f2 s = (v[0] + v[1]) + (v[2] + v[3]);
  return grp_sum_all(s.x + s.y);
}

template <bool ONE_D>
HSO_DEV bool g_klt_limited(const uint8_t* img, int cols, int rows, const f2 (&gxr)[4], const f2 (&gyr)[4], const f2 (&ref_px)[4],
                           double d0, double d1, double& pxs0, double& pxs1, f2 (&last_sample)[4], int l8)
{
  f2 Jx[4], Jy[4], wgt[4];
#pragma unroll
  for (int k = 0; k < 4; k++) {
    if (ONE_D) {
      Jx[k] = mk2((float)(0.5 * (d0 * (double)gxr[k].x + d1 * (double)gyr[k].x)), (float)(0.5 * (d0 * (double)gxr[k].y + d1 * (double)gyr[k].y)));
      Jy[k] = mk2(0.f, 0.f);
    } else { Jx[k] = 0.5f * gxr[k]; Jy[k] = 0.5f * gyr[k]; }   // (float)(0.5 * (double)g): halving is exact
    const f2 jj = ONE_D ? Jx[k] * Jx[k] : Jx[k] * Jx[k] + Jy[k] * Jy[k];
    wgt[k] = mk2(sqrtf((float)(250.0 / (250.0 + (double)jj.x))), sqrtf((float)(250.0 / (250.0 + (double)jj.y))));
  }
  float Hi[9];
  {
    f2 t0[4], t1[4], t2[4];
#pragma unroll
    for (int k = 0; k < 4; k++) { t0[k] = (Jx[k] * Jx[k]) * wgt[k]; t1[k] = (Jx[k] * 1.0f) * wgt[k]; t2[k] = wgt[k]; }
    const float h_xx = grp_sum8(t0), h_x1 = grp_sum8(t1), h_11 = grp_sum8(t2);
    if (ONE_D) {
      const float H00 = (float)((double)h_xx * (1 + 0.001)), H11 = (float)((double)h_11 * (1 + 0.001)), H01 = h_x1;
      const float det = H00 * H11 - H01 * H01;
      const float invdet = 1.0f / det;
      Hi[0] = H11 * invdet; Hi[1] = -H01 * invdet; Hi[3] = -H01 * invdet; Hi[4] = H00 * invdet;
    } else {
#pragma unroll
      for (int k = 0; k < 4; k++) { t0[k] = (Jx[k] * Jy[k]) * wgt[k]; t1[k] = (Jy[k] * Jy[k]) * wgt[k]; t2[k] = (Jy[k] * 1.0f) * wgt[k]; }
      const float h_xy = grp_sum8(t0), h_yy = grp_sum8(t1), h_y1 = grp_sum8(t2);
      const float H0 = (float)((double)h_xx * (1 + 0.001)), H4 = (float)((double)h_yy * (1 + 0.001)), H8 = (float)((double)h_11 * (1 + 0.001));
      const float H1 = h_xy, H2 = h_x1, H5 = h_y1, H3 = H1, H6 = H2, H7 = H5;
      const float c00 = H4 * H8 - H5 * H7, c01 = H5 * H6 - H3 * H8, c02 = H3 * H7 - H4 * H6;
      const float det = H0 * c00 + H1 * c01 + H2 * c02;
      const float invdet = 1.0f / det;
      Hi[0] = c00 * invdet; Hi[3] = c01 * invdet; Hi[6] = c02 * invdet;
      Hi[1] = (H2 * H7 - H1 * H8) * invdet; Hi[4] = (H0 * H8 - H2 * H6) * invdet; Hi[7] = (H1 * H6 - H0 * H7) * invdet;
      Hi[2] = (H1 * H5 - H2 * H4) * invdet; Hi[5] = (H2 * H3 - H0 * H5) * invdet; Hi[8] = (H0 * H4 - H1 * H3) * invdet;
    }
  }
  float mean_diff = 0;
  float bestU = (float)pxs0, bestV = (float)pxs1;
  float bestEnergy = 1e8f;
  float sb0 = 0, sb1 = 0, sb2 = 0;
  float uBak = bestU, vBak = bestV, meanBak = mean_diff;
  for (int iter = 0; iter < 10; ++iter) {
    const int u_r = (int)floorf(bestU), v_r = (int)floorf(bestV);   // (int)floor((double)bestU): the same integer
    if (u_r < 4 || v_r < 4 || u_r >= cols - 4 || v_r >= rows - 4) break;
    if (isnan(bestU) || isnan(bestV)) return false;
    const float sx = bestU - (float)u_r, sy = bestV - (float)v_r;
    // (float)((1.0 - sx) * (1.0 - sy)) etc.: exact factors, see bilin_weights; wBR is the reference's plain float product here
    const float cx = 1.0f - sx, cy = 1.0f - sy;
    const float wTL = cx * cy, wTR = sx * cy, wBL = cx * sy, wBR = sx * sy;
    const uint8_t* it = img + (v_r + l8 - 4) * cols + u_r - 4;
    const Row9 top = row9_of(load_px12(it)), bot = row9_of(load_px12(it + cols));
    bilin_row(wTL, wTR, wBL, wBR, top, bot, last_sample);
    f2 a0[4], a1[4], a2[4], a3[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const f2 res = (last_sample[k] - ref_px[k]) + mean_diff;
      a0[k] = (res * Jx[k]) * wgt[k]; a1[k] = res * wgt[k]; a2[k] = (res * res) * wgt[k];
      if (!ONE_D) a3[k] = (res * Jy[k]) * wgt[k];
    }
    const float j0 = -grp_sum8(a0);
    const float j2 = -grp_sum8(a1);
    const float energy = grp_sum8(a2);
    float j1 = 0;
    if (!ONE_D) j1 = -grp_sum8(a3);
    if (energy > bestEnergy) {
      sb0 *= 0.5f; sb1 *= 0.5f; sb2 *= 0.5f;
      if (ONE_D) { bestU = (float)((double)uBak + (double)sb0 * d0); bestV = (float)((double)vBak + (double)sb0 * d1); mean_diff = meanBak + sb1; }
      else { bestU = uBak + sb0; bestV = vBak + sb1; mean_diff = meanBak + sb2; }
    } else {
      float st0, st1, st2 = 0;
      if (ONE_D) {
        st0 = Hi[0] * j0 + Hi[1] * j2; st1 = Hi[3] * j0 + Hi[4] * j2;
        if (st0 < -0.5f) st0 = -0.5f; else if (st0 > 0.5f) st0 = 0.5f;
        if (!isfinite(st0)) { st0 = 0; st1 = 0; }
      } else {
        st0 = (Hi[0] * j0 + Hi[1] * j1) + Hi[2] * j2;
        st1 = (Hi[3] * j0 + Hi[4] * j1) + Hi[5] * j2;
        st2 = (Hi[6] * j0 + Hi[7] * j1) + Hi[8] * j2;
        if (st0 < -0.5f) st0 = -0.5f; else if (st0 > 0.5f) st0 = 0.5f;
        if (st1 < -0.5f) st1 = -0.5f; else if (st1 > 0.5f) st1 = 0.5f;
        if (!isfinite(st0)) { st0 = 0; st1 = 0; st2 = 0; }
      }
      uBak = bestU; vBak = bestV; meanBak = mean_diff;
      sb0 = st0; sb1 = st1; sb2 = st2;
      if (ONE_D) { bestU = (float)((double)bestU + (double)st0 * d0); bestV = (float)((double)bestV + (double)st0 * d1); mean_diff += st1; }
      else { bestU += st0; bestV += st1; mean_diff += st2; }
      bestEnergy = energy;
    }
    if (ONE_D) { if ((double)fabsf(sb0) < 0.01) break; }
    else { if ((double)(sb0 * sb1) < 0.01 * 0.01) break; }  // the reference multiplies the two components (:1435)
  }
  pxs0 = (double)bestU; pxs1 = (double)bestV;
  return !(bestEnergy > (float)(650 * 64));
}

// the common exit: full record, and for resident tables the state update + the compact record
HSO_DEV void seed_finish(const SeedConsts& C, SeedDev* seeds, int sid, hso_seed_out* outs, const hso_seed_out& o, int lane)
{
  if (lane != 0) return;
  if (outs) outs[sid] = o;
  if (C.update_in_place) { seeds[sid].s.mu = o.mu; seeds[sid].s.sigma2 = o.sigma2; seeds[sid].s.b = o.b; }
  if (C.brief) {
    hso_seed_brief br;
    br.mu = o.mu; br.sigma2 = o.sigma2; br.b = o.b;
    br.result = (int8_t)o.result; br.is_update = (int8_t)o.is_update; br.is_valid = (int8_t)o.is_valid; br.search_level = (int8_t)o.search_level;
    C.brief[sid] = br;
  }
  if (C.px) { C.px[2 * sid] = (float)o.px_cur[0]; C.px[2 * sid + 1] = (float)o.px_cur[1]; }
}

// observeDepthRow in three phases, so that what is identical in all 64 lanes of the wave that observes a seed is computed by
// ONE lane per seed for a group of seeds at a time (SEED_CPW per wave) instead of by every lane of every wave — about half of
// the kernel's instructions (two se3 products, the visibility projection, the affine warp matrix with its two radtan
// cam2world, the epipolar end points; afterwards cam2world of the match, the triangulation, computeTau's acos / sin and
// the update):
//   pre   thread = seed     everything up to the first image access                      -> SeedPre
//   image 8 lanes = seed    createPatch, the epipolar march, the two KLT refinements, checkNormal / checkNCC -> SeedMid
//   post  thread = seed     depthFromTriangulation, computeTau, updateSeed, the output records
// Each phase runs the statements of the one-phase kernel it replaces in their order; only the lane that executes them changed.
struct SeedPre {
  double pxc0, pxc1, pxf0, pxf1, incx, incy, ed0, ed1, dc0, dc1;
  float a00, a01, a10, a11, exposure_rat;
  int32_t sl, epl_start[2], epl_end[2];
  int32_t frame;     // index into SeedConsts::frames of the frame observed
  int32_t n_march;   // previous-frame pass: the number of steps of the march (after "expand one step", matcher.cpp:1177)
  int8_t state;      // 0: run the image phase; 1: not visible in the active frame; 2: doLineStereo returns -1 before any image
                     // access; 3: nothing to observe (erased slot, or the seed's group sits this call out); 4 (previous-frame pass):
                     // the epipolar segment is shorter than two pixels, no march, the refinement starts at (pxc0, pxc1) (:1102)
  int8_t is_valid, warp_nan, scale_exposure;
};
struct SeedMid {
  double px0, px1;                 // px_cur (level 0) when res_code == 1
  float zmncc_best, zmncc_second;
  int32_t n_steps, res_code;       // 1: matched (triangulate next), -4 / -3: rejected by the march / the refinement
};

// What the ordinary pass and the previous-frame pass share before the epipolar geometry: the visibility test (depth_filter.cpp:
// 590-606 / :688-699), the inverse-depth interval, T_cur_ref, the affine warp (warp::getWarpMatrixAffine, matcher.cpp:46-72), the
// search level, the inverse warp matrix and the exposure ratio.  Returns false when the point is not visible (P.state = 1).
struct SeedGeom { Se3 T; double A00, A01, A10, A11, min_idepth, prior_idepth, max_idepth; };
HSO_DEV bool seed_pre_common(const SeedConsts& C, const SeedDev& SD, const SeedFrameDev& F, SeedPre& P, SeedGeom& G)
{
  const hso_seed& S = SD.s;
  const int W = C.g.w[0], H = C.g.h[0];
  memset(&P, 0, sizeof(P));
  P.is_valid = 1;
  const Se3 Tcw = se3_from(F.T_f_w), Trw = se3_from(S.T_ref_w);
  const Se3 T_ref_cur = se3_mul(Trw, se3_inverse(Tcw));
  {
    const Se3 Tinv = se3_inverse(T_ref_cur);
    const double sc = 1.0 / (double)S.mu;
    double x, y, z;
    se3_apply(Tinv, sc * S.f[0], sc * S.f[1], sc * S.f[2], x, y, z);
    bool vis = !(z < 0.0);
    if (vis) {
      double cu, cv;
      world2cam(C.cam, x, y, z, cu, cv);
      const int ox = (int)cu, oy = (int)cv;
      vis = (ox >= 0 && ox < W && oy >= 0 && oy < H);
    }
    if (!vis) { P.state = 1; return false; }
  }
  const float z_inv_min = S.mu + 2 * sqrtf(S.sigma2);
  const float z_inv_max = fmaxf(S.mu - 2 * sqrtf(S.sigma2), 0.00000001f);
  if (isnan(z_inv_min)) P.is_valid = 0;
  G.min_idepth = 1.0 / (double)z_inv_min; G.prior_idepth = 1.0 / (double)S.mu; G.max_idepth = 1.0 / (double)z_inv_max;
  G.T = se3_mul(Tcw, se3_inverse(Trw));  // T_cur_ref, matcher.cpp:807 / :1055
  const Se3& T = G.T;
  P.state = 2;
  {
    const int hp = 5;
    const double xr = S.f[0] * G.prior_idepth, yr = S.f[1] * G.prior_idepth, zr = S.f[2] * G.prior_idepth;
    const int ratio = 1 << S.level;
    double du[3], dv[3];
    cam2world_dev(C.cam, S.px[0] + (double)(hp * ratio), S.px[1] + (double)(0 * ratio), du);
    cam2world_dev(C.cam, S.px[0] + (double)(0 * ratio), S.px[1] + (double)(hp * ratio), dv);
    const double su = zr / du[2], sv = zr / dv[2];
    for (int i = 0; i < 3; i++) { du[i] *= su; dv[i] *= sv; }
    double cx, cy, cz, ux, uy, uz, vx, vy, vz, pc0, pc1, pu0, pu1, pv0, pv1;
    se3_apply(T, xr, yr, zr, cx, cy, cz);
    se3_apply(T, du[0], du[1], du[2], ux, uy, uz);
    se3_apply(T, dv[0], dv[1], dv[2], vx, vy, vz);
    world2cam(C.cam, cx, cy, cz, pc0, pc1);
    world2cam(C.cam, ux, uy, uz, pu0, pu1);
    world2cam(C.cam, vx, vy, vz, pv0, pv1);
    G.A00 = (pu0 - pc0) / hp; G.A10 = (pu1 - pc1) / hp; G.A01 = (pv0 - pc0) / hp; G.A11 = (pv1 - pc1) / hp;
  }
  int sl = 0;
  { double D = G.A00 * G.A11 - G.A10 * G.A01; while (D > 3.0 && sl < HSO_N_SOBEL_LEVELS - 1) { sl += 1; D *= 0.25; } }
  P.sl = sl;
  P.exposure_rat = (float)(F.exposure / S.ref_exposure);
  {
    const double det = G.A00 * G.A11 - G.A10 * G.A01;
    const double invdet = 1.0 / det;
    P.a00 = (float)(G.A11 * invdet); P.a01 = (float)(-G.A01 * invdet); P.a10 = (float)(-G.A10 * invdet); P.a11 = (float)(G.A00 * invdet);
    P.warp_nan = isnan(P.a00) ? 1 : 0;
    P.scale_exposure = fabsf(P.exposure_rat * 128 - 128) > 30.0f ? 1 : 0;  // :818-826 (no keyframe-gap test here)
  }
  return true;
}

HSO_DEV SeedPre seed_pre(const SeedConsts& C, const SeedDev& SD, const SeedFrameDev& F)
{
  const hso_seed& S = SD.s;
  SeedPre P;
  SeedGeom G;
  if (!seed_pre_common(C, SD, F, P, G)) return P;
  // ---- Matcher::doLineStereo (matcher.cpp:802-1049), the part before the first image access
  const Se3& T = G.T;
  const double A00 = G.A00, A01 = G.A01, A10 = G.A10, A11 = G.A11;
  const double min_idepth = G.min_idepth, max_idepth = G.max_idepth;
  const int sl = P.sl;
  // close / far points on the unit plane, :834-852
  double pcx, pcy, pcz, pfx, pfy, pfz;
  se3_apply(T, S.f[0] * min_idepth, S.f[1] * min_idepth, S.f[2] * min_idepth, pcx, pcy, pcz);
  pcx /= pcz; pcy /= pcz;
  se3_apply(T, S.f[0] * max_idepth, S.f[1] * max_idepth, S.f[2] * max_idepth, pfx, pfy, pfz);
  if (pfz < 0.001 || max_idepth < min_idepth) return P;
  pfx /= pfz; pfy /= pfz;
  if (isnan((float)(pfx + pcx))) return P;
  double pxc0, pxc1, pxf0, pxf1;
  world2cam(C.cam, pcx, pcy, 1.0, pxc0, pxc1);
  P.epl_start[0] = (int)pxc0; P.epl_start[1] = (int)pxc1;
  pxc0 /= (double)(1 << sl); pxc1 /= (double)(1 << sl);
  world2cam(C.cam, pfx, pfy, 1.0, pxf0, pxf1);
  P.epl_end[0] = (int)pxf0; P.epl_end[1] = (int)pxf1;
  pxf0 /= (double)(1 << sl); pxf1 /= (double)(1 << sl);
  double incx = pxc0 - pxf0, incy = pxc1 - pxf1;
  const double eplLength = sqrt(incx * incx + incy * incy);
  if (((!eplLength) > 0) || isinf(eplLength)) return P;  // `!eplLength > 0`, :868
  if (eplLength > 100.0) { pxc0 = pxf0 + incx * 100.0 / eplLength; pxc1 = pxf1 + incy * 100.0 / eplLength; }
  incx *= 1.0 / eplLength; incy *= 1.0 / eplLength;
  pxf0 -= incx; pxf1 -= incy; pxc0 += incx; pxc1 += incy;
  if (eplLength < 2.0) {
    const double pad = (2.0 - eplLength) / 2.0;
    pxf0 -= incx * pad; pxf1 -= incy * pad; pxc0 += incx * pad; pxc1 += incy * pad;
  }
  double ed0 = pxc0 - pxf0, ed1 = pxc1 - pxf1;
  { const double en = sqrt(ed0 * ed0 + ed1 * ed1); ed0 /= en; ed1 /= en; }
  double dc0 = A00 * S.grad[0] + A01 * S.grad[1], dc1 = A10 * S.grad[0] + A11 * S.grad[1];
  { const double dn = sqrt(dc0 * dc0 + dc1 * dc1); dc0 /= dn; dc1 /= dn; }
  if (S.type == HSO_FTR_GRADIENT || S.type == HSO_FTR_EDGELET) {
    if (fabs(dc0 * ed0 + dc1 * ed1) < 0.4) return P;  // epi_search_edgelet_max_angle, matcher.h:130
  }
  P.pxc0 = pxc0; P.pxc1 = pxc1; P.pxf0 = pxf0; P.pxf1 = pxf1; P.incx = incx; P.incy = incy;
  P.ed0 = ed0; P.ed1 = ed1; P.dc0 = dc0; P.dc1 = dc1;
  P.state = 0;
  return P;
}

// Matcher::findEpipolarMatchPrevious (matcher.cpp:1051-1293), the part before the first image access.  The march of this
// matcher walks the unit plane in equal steps from B - step towards A and projects every step (:1174-1186): P.pxf = the first
// point, P.inc = the step, P.n_march = the number of steps; P.ed = (px_A - px_B).normalized(), P.dc = the warped gradient.
HSO_DEV SeedPre seed_pre_prev(const SeedConsts& C, const SeedDev& SD, const SeedFrameDev& F)
{
  const hso_seed& S = SD.s;
  SeedPre P;
  SeedGeom G;
  if (!seed_pre_common(C, SD, F, P, G)) return P;
  const Se3& T = G.T;
  double ax, ay, az, bx, by, bz;
  se3_apply(T, S.f[0] * G.min_idepth, S.f[1] * G.min_idepth, S.f[2] * G.min_idepth, ax, ay, az);
  ax /= az; ay /= az;
  se3_apply(T, S.f[0] * G.max_idepth, S.f[1] * G.max_idepth, S.f[2] * G.max_idepth, bx, by, bz);
  bx /= bz; by /= bz;
  const double epi0 = ax - bx, epi1 = ay - by;
  double pxa0, pxa1, pxb0, pxb1;
  world2cam(C.cam, ax, ay, 1.0, pxa0, pxa1);
  world2cam(C.cam, bx, by, 1.0, pxb0, pxb1);
  const double dab0 = pxa0 - pxb0, dab1 = pxa1 - pxb1;
  const double dabn = sqrt(dab0 * dab0 + dab1 * dab1);
  const double epi_length = dabn / (double)(1 << P.sl);
  double dc0 = G.A00 * S.grad[0] + G.A01 * S.grad[1], dc1 = G.A10 * S.grad[0] + G.A11 * S.grad[1];
  { const double dn = sqrt(dc0 * dc0 + dc1 * dc1); dc0 /= dn; dc1 /= dn; }
  if (S.type == HSO_FTR_GRADIENT || S.type == HSO_FTR_EDGELET) {
    const double en = sqrt(epi0 * epi0 + epi1 * epi1);
    if (fabs(dc0 * (epi0 / en) + dc1 * (epi1 / en)) < 0.4) return P;   // :1078-1084 (state 2: "false" before any image access)
  }
  P.ed0 = dab0 / dabn; P.ed1 = dab1 / dabn; P.dc0 = dc0; P.dc1 = dc1;
  if (epi_length < 2.0) {
    P.pxc0 = (pxa0 + pxb0) / 2.0; P.pxc1 = (pxa1 + pxb1) / 2.0;   // px_cur_, level 0 (:1100)
    P.state = 4;
    return P;
  }
  // size_t n_steps = epi_length_ / 0.7; ... if(n_steps > max_epi_search_steps) return false (:1158-1162).  A NaN length converts to
  // 2^63 on the reference's x86 build (cvttsd2si) and is rejected there too; said explicitly here and in the CPU restatement
  const double q_steps = epi_length / 0.7;
  if (!(q_steps < 101.0)) return P;
  const unsigned long long n_steps = (unsigned long long)q_steps;
  const double st0 = epi0 / (double)n_steps, st1 = epi1 / (double)n_steps;
  P.pxf0 = bx - st0; P.pxf1 = by - st1; P.incx = st0; P.incy = st1;
  P.n_march = (int)n_steps + 1;
  P.state = 0;
  return P;
}

// Matcher::doLineStereo's refinement of a march result (matcher.cpp:966-1046; the same statements close
// findEpipolarMatchPrevious, :1102-1150 / :1236-1290): KLTLimited1D along the epipolar direction, then KLTLimited2D (or, for an
// edgelet, KLTLimited1D along the warped gradient + checkNormal), then checkNCC.  `ps0, ps1`: the start position on the search
// level, replaced by the refined one.  Eight lanes per seed; `pwb`: the seed's 10x10 patch in LDS.
HSO_DEV bool g_refine(const SeedConsts& C, const uint8_t* cur_base, int sl, int type, const float* pwb, double ed0, double ed1,
                      double dc0, double dc1, double& ps0, double& ps1, int l8)
{
  const int cols = C.g.w[sl], rows = C.g.h[sl];
  const uint8_t* cur = cur_base + C.g.off[sl];
  f2 ref_px[4], gxr[4], gyr[4];
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const int c = (l8 + 1) * 10 + 2 * k + 1;
    ref_px[k] = mk2(pwb[c], pwb[c + 1]);
    gxr[k] = mk2(pwb[c + 1] - pwb[c - 1], pwb[c + 2] - pwb[c]);
    gyr[k] = mk2(pwb[c + 10] - pwb[c - 10], pwb[c + 11] - pwb[c - 9]);
  }
  const double start0 = ps0, start1 = ps1;
  f2 samp[4];
#pragma unroll
  for (int k = 0; k < 4; k++) samp[k] = mk2(0.f, 0.f);
  bool result = g_klt_limited<true>(cur, cols, rows, gxr, gyr, ref_px, ed0, ed1, ps0, ps1, samp, l8);
  if (!result) { ps0 = start0; ps1 = start1; }
#pragma unroll
  for (int k = 0; k < 4; k++) samp[k] = mk2(0.f, 0.f);  // patch2D: written only by the second KLT (zero where the reference leaves it uninitialised)
  if (type != HSO_FTR_EDGELET) {
    result = g_klt_limited<false>(cur, cols, rows, gxr, gyr, ref_px, 0, 0, ps0, ps1, samp, l8);
  } else {
    result = g_klt_limited<true>(cur, cols, rows, gxr, gyr, ref_px, dc0, dc1, ps0, ps1, samp, l8);
    if (result) {
      // Matcher::checkNormal(cur_frame, search_level_, px, dir_cur, 0.7), :406-440
      const int16_t* gx = reinterpret_cast<const int16_t*>(cur_base + C.g.sob_off[sl][0]);
      const int16_t* gy = reinterpret_cast<const int16_t*>(cur_base + C.g.sob_off[sl][1]);
      const float uf = (float)ps0, vf = (float)ps1;
      // The reference reads the four taps unchecked (:421-428): a NaN position (an edgelet direction of norm 0 lets
      // KLTLimited1D "succeed" with a NaN pixel) or one outside the image is an out-of-bounds read there.  Defined here and in
      // the CPU restatement alike: such a position fails the check.
      if (!(uf >= 0 && vf >= 0 && uf < (float)(cols - 1) && vf < (float)(rows - 1))) {
        result = false;
      } else {
        const int ui = (int)floorf((float)ps0), vi = (int)floorf((float)ps1);
        const float sx = uf - (float)ui, sy = vf - (float)vi;
        const float wTL = (float)((1.0 - sx) * (1.0 - sy)), wTR = (float)(sx * (1.0 - sy)), wBL = (float)((1.0 - sx) * sy);
        const float wBR = (float)(((1.0 - wTL) - wTR) - wBL);
        const int gs = C.g.sob_stride[sl];
        const int a = vi * gs + ui;
        double n0 = (((double)wTL * (double)gx[a] + (double)wTR * (double)gx[a + 1]) + (double)wBL * (double)gx[a + gs]) + (double)wBR * (double)gx[a + gs + 1];
        double n1 = (((double)wTL * (double)gy[a] + (double)wTR * (double)gy[a + 1]) + (double)wBL * (double)gy[a + gs]) + (double)wBR * (double)gy[a + gs + 1];
        const double nn = sqrt(n0 * n0 + n1 * n1);
        n0 /= nn; n1 /= nn;
        result = (dc0 * n0 + dc1 * n1) > (double)(float)0.7;
      }
    }
  }
  if (result) {
    // Matcher::checkNCC(patch_f_, patch2D, 0.8), :379-404
    const float mean1 = grp_sum8(ref_px) / 64, mean2 = grp_sum8(samp) / 64;
    f2 qq[4], q11[4], q22[4];
#pragma unroll
    for (int k = 0; k < 4; k++) { const f2 q1 = ref_px[k] - mean1, q2 = samp[k] - mean2; qq[k] = q1 * q2; q11[k] = q1 * q1; q22[k] = q2 * q2; }
    const float num = grp_sum8(qq), den1 = grp_sum8(q11), den2 = grp_sum8(q22);
    result = ((double)num / ((double)sqrtf(den1 * den2) + 1e-12)) > (double)(float)0.8;
  }
  return result;
}

// ZMNCC_F's constructor part (patch_score.h:268-285) for the patch in `pwb`: host[i] - hostMean into `hd` (LDS, 64 floats),
// returns d1 = sum of their squares.  Serial sums in the reference's order, every lane of the group computes the same bits.
HSO_DEV float g_host_terms(const float* pwb, float* hd, int l8)
{
  float m = 0;
#pragma unroll
  for (int r = 0; r < 8; r++)
#pragma unroll
    for (int j = 0; j < 8; j++) m += pwb[(r + 1) * 10 + j + 1];
  m /= 64;
#pragma unroll
  for (int j = 0; j < 8; j++) hd[l8 * 8 + j] = pwb[(l8 + 1) * 10 + j + 1] - m;
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  float d1 = 0;
#pragma unroll
  for (int i = 0; i < 64; i++) d1 += hd[i] * hd[i];
  return d1;
}

// the best / second-best bookkeeping of a march (matcher.cpp:940-957), for the scores of up to eight steps held one per lane
// (`z`: -inf where the lane has no step or the step is outside the image; `k`: the step's index), applied in step order
struct MarchBest { float best, second; int i_best, i_second; };
HSO_DEV void g_march_fold(MarchBest& B, float z, int k)
{
  // a score can only matter if it beats the second best the group held before these eight steps
  const unsigned long long hot = __ballot(z > B.second);
  const unsigned mine = (unsigned)(hot >> ((threadIdx.x & 63) & ~7)) & 0xffu;
  for (unsigned m = mine; m; m &= m - 1) {
    const int j = __builtin_ctz(m);
    const float zj = grp_get(z, j);
    const int kj = grp_get(k, j);
    if (zj > B.best) { B.second = B.best; B.i_second = B.i_best; B.best = zj; B.i_best = kj; }
    else if (zj > B.second) { B.second = zj; B.i_second = kj; }
  }
}

// warp::createPatch (matcher.cpp:159-196) for the host patch of a seed: 10x10 samples of the reference level into `pwb_lds`
HSO_DEV void g_host_patch(const SeedConsts& C, const SeedDev& SD, const SeedPre& P, float* pwb_lds, int l8)
{
  const hso_seed& S = SD.s;
  const float a00 = P.a00, a01 = P.a01, a10 = P.a10, a11 = P.a11, exposure_rat = P.exposure_rat;
  const bool warp_nan = P.warp_nan != 0, scale_exposure = P.scale_exposure != 0;
  const int L = S.level, cols = C.g.w[L], rows = C.g.h[L];  // img_pyr_[L].cols / rows
  const uint8_t* img = SD.ref_base + C.g.off[L];
  const float rx = (float)(S.px[0] / (double)(1 << L)), ry = (float)(S.px[1] / (double)(1 << L));
  const float scaleTarget = (float)(1 << P.sl);
  // thirteen samples per lane, unrolled and without a branch around the loads, so that all 26 of them are in flight at once
  // (the loop form paid thirteen memory latencies one after the other: the longest wait of the kernel)
#pragma unroll
  for (int it = 0; it < 13; it++) {
    const int idx = l8 + 8 * it;
    const int y = idx / 10, x = idx - 10 * y;
    float p0 = (float)(x - 5), p1 = (float)(y - 5);
    p0 *= scaleTarget; p1 *= scaleTarget;
    const float px0 = (a00 * p0 + a01 * p1) + rx, px1 = (a10 * p0 + a11 * p1) + ry;
    const bool inside = !warp_nan && !(px0 < 0 || px1 < 0 || px0 >= (float)(cols - 1) || px1 >= (float)(rows - 1));
    float val = interpolate_8u(img, cols, inside ? px0 : 0.f, inside ? px1 : 0.f);
    if (!inside) val = 0;
    if (scale_exposure) val = val * exposure_rat;
    if (idx < 100) pwb_lds[idx] = val;
  }
}

// The march of a seed (matcher.cpp:906-960) in three steps, so that the 64 lanes of the wave share the steps of its eight seeds
// evenly however unequal the eight epipolar segments are (7 .. 27 steps in profiles' seed stage: a group that evaluated only
// its own steps, eight at a time, kept the wave for max over groups of ceil(steps / 8) rounds = 3.2 on average, the shared
// list needs ceil(sum / 64) = 2.1):
//   1. every group walks its segment and lists the steps that lie inside the image: float position + step index (LDS);
//   2. the wave evaluates the listed steps of all eight seeds, one step per lane and round (lane_zmncc);
//   3. every group folds its scores in step order into best / second.
// A list holds SEED_MARCH_CAP steps per seed (a segment is at most ~104 steps long); longer marches (the defensive 4096 bound:
// NaN increments) take further passes of the three steps.
#define SEED_MARCH_CAP 112
struct MarchSlot { float a, b; };   // step 1: u, v of the step (u NaN: outside the image, no score); after step 2: a = its score
struct MarchWalk { double x, y; int count; bool more; };

// step 1; returns the number of steps listed (step `first + j` in slot j, `first` = Wk.count on entry)
HSO_DEV int g_march_list(const SeedPre& P, MarchWalk& Wk, int lim_x, int lim_y, MarchSlot* slots, int l8)
{
  const double pxc0 = P.pxc0, pxc1 = P.pxc1, incx = P.incx, incy = P.incy;
  int m = 0;
#pragma unroll 1
  for (; m < SEED_MARCH_CAP; m++) {
    if (!((((incx < 0) == (Wk.x > pxc0)) && ((incy < 0) == (Wk.y > pxc1))) || Wk.count == 0)) { Wk.more = false; break; }
    const int ox = (int)Wk.x, oy = (int)Wk.y;
    const bool inside = ox >= 8 && ox < lim_x && oy >= 8 && oy < lim_y;
    if (l8 == 0) { slots[m].a = inside ? (float)Wk.x : __builtin_nanf(""); slots[m].b = (float)Wk.y; }
    Wk.x += incx; Wk.y += incy; Wk.count++;
    if (Wk.count > 4096) { Wk.more = false; m++; break; }  // defensive bound (NaN increments would never terminate)
  }
  return m;
}

HSO_DEV hso_seed_out seed_post(const SeedConsts& C, const SeedDev& SD, const SeedFrameDev& F, const SeedPre& P, const SeedMid& M)
{
  const hso_seed& S = SD.s;
  hso_seed_out o;
  memset(&o, 0, sizeof(o));
  o.mu = S.mu; o.sigma2 = S.sigma2; o.b = S.b; o.is_valid = 1;
  if (P.state == 1) { o.result = 0; o.is_update = 0; return o; }   // not visible: nothing else is touched
  o.is_update = 1;
  o.is_valid = P.is_valid;
  o.search_level = P.sl;
  const Se3 Tcw = se3_from(F.T_f_w), Trw = se3_from(S.T_ref_w);
  const Se3 T_ref_cur = se3_mul(Trw, se3_inverse(Tcw));
  const Se3 T = se3_mul(Tcw, se3_inverse(Trw));  // T_cur_ref, :807
  const bool prev = C.frame_keys != nullptr;   // observeDepthWithPreviousFrameOnce: a failed match leaves b alone (:710-714)
  int res_code = -1;
  if (P.state == 0 || P.state == 4) {
    o.epl_start[0] = P.epl_start[0]; o.epl_start[1] = P.epl_start[1]; o.epl_end[0] = P.epl_end[0]; o.epl_end[1] = P.epl_end[1];
    o.n_steps = M.n_steps; o.zmncc_best = M.zmncc_best; o.zmncc_second = M.zmncc_second;
    res_code = M.res_code;
    if (res_code == 1) do {
      const double pxcur0 = M.px0, pxcur1 = M.px1;
      o.px_cur[0] = pxcur0; o.px_cur[1] = pxcur1;
      // depthFromTriangulation(T_cur_ref, f_ref, cam2world(px_cur_)), :242-255
      double fc[3];
      cam2world_dev(C.cam, pxcur0, pxcur1, fc);
      double R[9];
      so3_matrix(T, R);
      double a0[3];
      for (int i = 0; i < 3; i++) a0[i] = R[i * 3 + 0] * S.f[0] + R[i * 3 + 1] * S.f[1] + R[i * 3 + 2] * S.f[2];
      const double m00 = a0[0] * a0[0] + a0[1] * a0[1] + a0[2] * a0[2];
      const double m01 = a0[0] * fc[0] + a0[1] * fc[1] + a0[2] * fc[2];
      const double m11 = fc[0] * fc[0] + fc[1] * fc[1] + fc[2] * fc[2];
      const double det = m00 * m11 - m01 * m01;
      if (det < 0.000001) { res_code = -2; break; }
      const double invdet = 1.0 / det;
      const double i00 = m11 * invdet, i01 = -m01 * invdet;
      const double r0x = (-i00) * a0[0] + (-i01) * fc[0], r0y = (-i00) * a0[1] + (-i01) * fc[1], r0z = (-i00) * a0[2] + (-i01) * fc[2];
      o.z = fabs(r0x * T.tx + r0y * T.ty + r0z * T.tz);
    } while (0);
  }
  o.result = res_code;
  if (res_code != 1) {
    if (!prev) o.b = S.b + 1;  // :634
    o.epl_start[0] = o.epl_start[1] = o.epl_end[0] = o.epl_end[1] = 0;
  } else {
    // computeTau (:539-555, with hso::PI = 3.14159265) and updateSeed (:527-537)
    const double PI_ = 3.14159265;
    const double z = o.z;
    const double t0 = T_ref_cur.tx, t1 = T_ref_cur.ty, t2 = T_ref_cur.tz;
    const double a0 = S.f[0] * z - t0, a1 = S.f[1] * z - t1, a2 = S.f[2] * z - t2;
    const double t_norm = sqrt(t0 * t0 + t1 * t1 + t2 * t2), a_norm = sqrt(a0 * a0 + a1 * a1 + a2 * a2);
    const double alpha = acos((S.f[0] * t0 + S.f[1] * t1 + S.f[2] * t2) / t_norm);
    const double beta = acos((a0 * -t0 + a1 * -t1 + a2 * -t2) / (t_norm * a_norm));
    const double beta_plus = beta + C.px_error_angle;
    const double gamma_plus = PI_ - alpha - beta_plus;
    const double z_plus = t_norm * sin(beta_plus) / sin(gamma_plus);
    const double tau = z_plus - z;
    const double tau_inverse = 0.5 * (1.0 / fmax(0.0000001, z - tau) - 1.0 / (z + tau));
    const float x = (float)(1. / z), tau2 = (float)(tau_inverse * tau_inverse);
    float id_var = S.sigma2 * 1.01f;
    const float w = tau2 / (tau2 + id_var);
    const float new_idepth = (1 - w) * x + w * S.mu;
    // UNZERO (:526): clamp away from zero (comparisons against double constants)
    const double nd = (double)new_idepth;
    o.mu = (float)(nd < 0 ? (nd > -1e-10 ? -1e-10 : nd) : (nd < 1e-10 ? 1e-10 : nd));
    id_var *= w;
    o.sigma2 = (id_var < S.sigma2) ? id_var : S.sigma2;
  }
  return o;
}

// step 1 of the march for the previous-frame matcher (matcher.cpp:1174-1186): the steps are equal steps on the unit plane,
// every one projected by world2cam; lane l8 projects steps l8, l8 + 8, ... of the group's seed
HSO_DEV int g_march_list_prev(const SeedConsts& C, const SeedPre& P, MarchWalk& Wk, int sl, int lim_x, int lim_y, MarchSlot* slots, int l8)
{
  const double incx = P.incx, incy = P.incy;
  const int n_march = P.n_march;
  int m = 0;
  while (m < SEED_MARCH_CAP - 7 && Wk.count < n_march) {
    double ux = 0, uy = 0;
    int taken = 0;
#pragma unroll 1
    for (int j = 0; j < 8 && Wk.count < n_march; j++) {
      if (j == l8) { ux = Wk.x; uy = Wk.y; }
      Wk.x += incx; Wk.y += incy; Wk.count++; taken++;
    }
    if (l8 < taken) {
      double px, py;
      world2cam(C.cam, ux, uy, 1.0, px, py);
      px /= (double)(1 << sl); py /= (double)(1 << sl);
      const int ox = (int)px, oy = (int)py;
      const bool inside = ox >= 8 && ox < lim_x && oy >= 8 && oy < lim_y;
      slots[m + l8].a = inside ? (float)px : __builtin_nanf(""); slots[m + l8].b = (float)py;
    }
    m += taken;
  }
  Wk.more = Wk.count < n_march;
  return m;
}

// ---- the three kernels of an observation ---------------------------------------------------------------------------------
// pre and post run one THREAD per seed (all 64 lanes of a wave busy with fp64 geometry; in round 3 they ran inside the image
// kernel on the 4-16 lanes of a wave that owned a seed: ~440 wave-instructions per seed at 16 seeds per wave, ~110 now), the
// image kernel eight lanes per seed, eight seeds per wave and exactly one seed per group, so a batch of n seeds is n / 8 waves
// whatever its size.  SeedPre / SeedMid travel through HBM (160 B per seed written once and read once or twice: < 0.1 ms per
// million seeds at HBM rate).
static __global__ __launch_bounds__(256) void k_seed_pre(SeedConsts C, const SeedDev* __restrict__ seeds, int n_seeds, SeedPre* __restrict__ pre)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_seeds) return;
  const SeedDev& SD = seeds[i];
  if (SD.ref_base == nullptr) { pre[i].state = 3; return; }   // an erased slot of a resident table
  if (C.frame_keys) {
    // previous-frame pass: the frame this seed's keyframe observes, if it is in the call (keys ascending)
    const int64_t key = SD.s.ref_frame_id;
    int lo = 0, hi = C.n_frame_keys;
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (C.frame_keys[mid] < key) lo = mid + 1; else hi = mid; }
    if (lo >= C.n_frame_keys || C.frame_keys[lo] != key || C.frames[lo].cur_base == nullptr) { pre[i].state = 3; return; }
    SeedPre P = seed_pre_prev(C, SD, C.frames[lo]);
    P.frame = lo;
    pre[i] = P;
    return;
  }
  // a null active frame: the seed's group sits this call out
  if (SD.cur_base == nullptr && C.frames[SD.frame].cur_base == nullptr) { pre[i].state = 3; return; }
  SeedPre P = seed_pre(C, SD, C.frames[SD.frame]);
  P.frame = SD.frame;
  pre[i] = P;
}

// three waves per SIMD: the lane-serial march step holds its 64 samples in registers (168 VGPRs); LDS 12.4 KB per wave
template <bool PREV>
static __global__ __launch_bounds__(64 * SEED_WAVES_PER_BLOCK) __attribute__((amdgpu_waves_per_eu(3, 3)))
void k_seed_image(SeedConsts C, const SeedDev* __restrict__ seeds, int n_seeds, const SeedPre* __restrict__ pre, SeedMid* __restrict__ mid)
{
  __shared__ float s_pwb[SEED_WAVES_PER_BLOCK][8][100];
  __shared__ __attribute__((aligned(16))) float s_hd[SEED_WAVES_PER_BLOCK][8][64];
  __shared__ MarchSlot s_slot[SEED_WAVES_PER_BLOCK][8][SEED_MARCH_CAP];
  struct GroupImg { const uint8_t* cur; int cols; float d1; int count; int pad_; };
  __shared__ GroupImg s_grp[SEED_WAVES_PER_BLOCK][8];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, grp = lane >> 3, l8 = lane & 7;
  const int first = (blockIdx.x * SEED_WAVES_PER_BLOCK + wave) * 8;
  if (first >= n_seeds) return;
  const int sid = min(first + grp, n_seeds - 1);
  const SeedPre& P = pre[sid];
  const SeedDev& SD = seeds[sid];
  // a group without a seed to observe stays: its lanes evaluate march steps of the other groups
  const bool active = first + grp < n_seeds && (P.state == 0 || (PREV && P.state == 4));
  const uint8_t* const cur_base = (!PREV && SD.cur_base) ? SD.cur_base : C.frames[active ? P.frame : 0].cur_base;
  float* const pwb = s_pwb[wave][grp];
  float* const hd = s_hd[wave][grp];
  const int sl = active ? P.sl : 0;
  if (active) {
    g_host_patch(C, SD, P, pwb, l8);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    const float d1 = g_host_terms(pwb, hd, l8);
    if (l8 == 0) { s_grp[wave][grp].cur = cur_base + C.g.off[sl]; s_grp[wave][grp].cols = C.g.w[sl]; s_grp[wave][grp].d1 = d1; }
  }

  // ---- march along the epipolar line, ZMNCC per step (:906-960)
  MarchBest B;
  B.best = 0.1f; B.second = 0.1f; B.i_best = PREV ? 0 : -1; B.i_second = PREV ? 0 : -1;   // bestCounter / secondCounter start at 0 (:1168)
  MarchWalk Wk;
  Wk.x = active ? P.pxf0 : 0.0; Wk.y = active ? P.pxf1 : 0.0; Wk.count = 0; Wk.more = active && P.state == 0;
  const int lim_x = C.g.w[0] / (1 << sl) - 8, lim_y = C.g.h[0] / (1 << sl) - 8;
  do {
    int m = 0;
    const int k_first = Wk.count;
    if (Wk.more) m = PREV ? g_march_list_prev(C, P, Wk, sl, lim_x, lim_y, s_slot[wave][grp], l8) : g_march_list(P, Wk, lim_x, lim_y, s_slot[wave][grp], l8);
    if (l8 == 0) s_grp[wave][grp].count = m;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    int end[8];
    {
      int acc = 0;
#pragma unroll
      for (int g = 0; g < 8; g++) { acc += s_grp[wave][g].count; end[g] = acc; }
    }
    for (int item = lane; item < end[7]; item += 64) {
      int g = 0, base = 0;
#pragma unroll
      for (int h = 0; h < 7; h++) if (item >= end[h]) { g = h + 1; base = end[h]; }
      MarchSlot& slot = s_slot[wave][g][item - base];
      const GroupImg& G = s_grp[wave][g];
      const float u = slot.a;
      slot.a = isnan(u) ? -__builtin_inff() : lane_zmncc(G.cur, G.cols, u, slot.b, s_hd[wave][g], G.d1);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    for (int j0 = 0; j0 < m; j0 += 8) {
      const int j = j0 + l8;
      const float z = j < m ? s_slot[wave][grp][j].a : -__builtin_inff();
      const int k = j < m ? k_first + j : -2;
      g_march_fold(B, z, k);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
  } while (__ballot(Wk.more) != 0ull);
  if (!active) return;

  SeedMid M;
  M.px0 = M.px1 = 0; M.res_code = -4;
  M.n_steps = Wk.count; M.zmncc_best = B.best; M.zmncc_second = B.second;
  do {
    double ps0, ps1;
    if (PREV && P.state == 4) {
      M.zmncc_best = M.zmncc_second = 0;   // no march
      ps0 = P.pxc0 / (double)(1 << sl); ps1 = P.pxc1 / (double)(1 << sl);
    } else {
      const int dl = B.i_best - B.i_second;
      // doLineStereo: abs(loopCBest - loopCSecond) > 1 (:962).  findEpipolarMatchPrevious: fabs() of a size_t difference
      // (:1219), which wraps: only "second == best" and "second == best - 1" count as adjacent
      const bool apart = PREV ? !(dl == 0 || dl == 1) : (float)(dl < 0 ? -dl : dl) > 1.0f;
      if (apart && 1.5f * B.second > B.best) break;
      if (!((double)B.best > 0.8)) break;
      // uv_best: the position of step i_best, by the additions that led there (zmncc_best > 0.8 > 0.1: a step was recorded)
      double uvb0 = P.pxf0, uvb1 = P.pxf1;
      {
        const double incx = P.incx, incy = P.incy;
#pragma unroll 1
        for (int k = 0; k < B.i_best; k++) { uvb0 += incx; uvb1 += incy; }
      }
      if (PREV) {
        double px, py;
        world2cam(C.cam, uvb0, uvb1, 1.0, px, py);   // px_cur_ = world2cam(uv_best), :1226
        ps0 = px / (double)(1 << sl); ps1 = py / (double)(1 << sl);
      } else {
        const double pxcur0 = uvb0 * (double)(1 << sl), pxcur1 = uvb1 * (double)(1 << sl);
        ps0 = pxcur0 / (double)(1 << sl); ps1 = pxcur1 / (double)(1 << sl);
      }
    }
    // ---- refinement (:966-1046 / :1102-1150, :1236-1290)
    if (!g_refine(C, cur_base, sl, SD.s.type, pwb, P.ed0, P.ed1, P.dc0, P.dc1, ps0, ps1, l8)) { M.res_code = -3; break; }
    M.px0 = ps0 * (double)(1 << sl); M.px1 = ps1 * (double)(1 << sl);
    M.res_code = 1;
  } while (0);
  if (l8 == 0) mid[sid] = M;
}

static __global__ __launch_bounds__(256) void k_seed_post(SeedConsts C, SeedDev* __restrict__ seeds, int n_seeds, const SeedPre* __restrict__ pre,
                                                           const SeedMid* __restrict__ mid, hso_seed_out* __restrict__ outs)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_seeds) return;
  const SeedPre P = pre[i];
  if (P.state == 3) {
    if (C.brief) { hso_seed_brief br; memset(&br, 0, sizeof(br)); C.brief[i] = br; }
    if (C.px) { C.px[2 * i] = 0.f; C.px[2 * i + 1] = 0.f; }
    return;
  }
  SeedMid M;
  memset(&M, 0, sizeof(M));
  if (P.state == 0 || P.state == 4) M = mid[i];
  const hso_seed_out o = seed_post(C, seeds[i], C.frames[P.frame], P, M);
  seed_finish(C, seeds, i, outs, o, 0);
}

// one observation of `n` seed records on `stream` (the context's, or the depth filter's own), through the scratch buffer given
static int seed_observe_launch_on(hso_gpu_ctx* ctx, hipStream_t stream, char** scratch, size_t* scratch_cap, const SeedConsts& C, SeedDev* seeds, int n,
                                  hso_seed_out* outs)
{
  auto al = [](size_t b) { return (b + 255) & ~size_t(255); };
  const size_t need = al((size_t)n * sizeof(SeedPre)) + al((size_t)n * sizeof(SeedMid));
  if (*scratch_cap < need) {
    HSO_HIP_CHECK(ctx, hipStreamSynchronize(stream));
    if (*scratch) (void)hipFree(*scratch);
    *scratch = nullptr; *scratch_cap = 0;
    HSO_HIP_CHECK(ctx, hipMalloc(reinterpret_cast<void**>(scratch), hso_grown(need)));
    *scratch_cap = hso_grown(need);
  }
  SeedPre* pre = reinterpret_cast<SeedPre*>(*scratch);
  SeedMid* mid = reinterpret_cast<SeedMid*>(*scratch + al((size_t)n * sizeof(SeedPre)));
  const int per_block = 8 * SEED_WAVES_PER_BLOCK;
  hipLaunchKernelGGL(k_seed_pre, dim3((n + 255) / 256), dim3(256), 0, stream, C, seeds, n, pre);
  if (C.frame_keys) hipLaunchKernelGGL(k_seed_image<true>, dim3((n + per_block - 1) / per_block), dim3(64 * SEED_WAVES_PER_BLOCK), 0, stream, C, seeds, n, pre, mid);
  else hipLaunchKernelGGL(k_seed_image<false>, dim3((n + per_block - 1) / per_block), dim3(64 * SEED_WAVES_PER_BLOCK), 0, stream, C, seeds, n, pre, mid);
  hipLaunchKernelGGL(k_seed_post, dim3((n + 255) / 256), dim3(256), 0, stream, C, seeds, n, pre, mid, outs);
  HSO_HIP_CHECK(ctx, hipGetLastError());
  return HSO_OK;
}
static int seed_observe_launch(hso_gpu_ctx* ctx, const SeedConsts& C, SeedDev* seeds, int n, hso_seed_out* outs)
{
  return seed_observe_launch_on(ctx, ctx->stream, &ctx->d_seed_scratch, &ctx->seed_scratch_cap, C, seeds, n, outs);
}

// A previous-frame pass in flight on the depth filter's stream: wait for it (its briefs stay in page-locked memory until
// hso_gpu_seed_table_observe_previous_end collects them).  Called by everything that touches seed tables or releases frames.
int hso_seed_async_quiesce(hso_gpu_ctx* ctx)
{
  if (!ctx->seed_inflight || !ctx->seed_stream) return HSO_OK;
  HSO_HIP_CHECK(ctx, hipStreamSynchronize(ctx->seed_stream));
  return HSO_OK;
}

extern "C" int hso_gpu_seed_observe_multi(hso_gpu_ctx* ctx, const hso_camera* cam, const hso_seed_frame* frames, int n_frames,
                                          const int32_t* seed_frame, double px_error_angle, const hso_seed* seeds, int n_seeds,
                                          hso_seed_out* out)
{
  if (!ctx) return HSO_E_INVALID;
  if (int rc = hso_seed_async_quiesce(ctx)) return rc;
  if (!cam || n_frames < 0 || n_seeds < 0 || (n_seeds > 0 && (!seeds || !out || !frames || !seed_frame || n_frames == 0)))
    return hso_fail(ctx, HSO_E_INVALID, "seed_observe: bad argument");
  if (n_seeds == 0) return HSO_OK;
  HSO_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  PyrGeom g{};
  std::vector<const uint8_t*> cur_base(n_frames);
  std::vector<SeedFrameDev> hf(n_frames);
  for (int k = 0; k < n_frames; k++) {
    auto itc = ctx->frames.find(frames[k].frame_id);
    if (itc == ctx->frames.end()) return hso_fail(ctx, HSO_E_NOFRAME, "seed_observe: active frame not resident");
    if (k == 0) g = itc->second.g;
    else if (!same_geom(itc->second.g, g)) return hso_fail(ctx, HSO_E_INVALID, "seed_observe: frames must share one size");
    cur_base[k] = itc->second.base;
    hf[k].T_f_w = frames[k].T_f_w; hf[k].exposure = frames[k].exposure_time; hf[k].cur_base = itc->second.base;
  }
  if (cam->width != g.w[0] || cam->height != g.h[0]) return hso_fail(ctx, HSO_E_INVALID, "seed_observe: camera size differs from the frame size");
  SeedDev* h = reinterpret_cast<SeedDev*>(hso_pinned(ctx, 0, (size_t)n_seeds * sizeof(SeedDev)));
  hso_seed_out* h_out = reinterpret_cast<hso_seed_out*>(hso_pinned(ctx, 1, (size_t)n_seeds * sizeof(hso_seed_out)));
  if (!h || !h_out) return HSO_E_NOMEM;
  int64_t last_id = -1;
  const uint8_t* last_base = nullptr;
  for (int i = 0; i < n_seeds; i++) {
    if (seed_frame[i] < 0 || seed_frame[i] >= n_frames) return hso_fail(ctx, HSO_E_INVALID, "seed_observe: seed_frame out of range");
    if (i == 0 || seeds[i].ref_frame_id != last_id) {
      auto itr = ctx->frames.find(seeds[i].ref_frame_id);
      if (itr == ctx->frames.end()) return hso_fail(ctx, HSO_E_NOFRAME, "seed_observe: seed host frame not resident");
      if (!same_geom(itr->second.g, g)) return hso_fail(ctx, HSO_E_INVALID, "seed_observe: frames must share one size");
      last_id = seeds[i].ref_frame_id; last_base = itr->second.base;
    }
    if (seeds[i].level < 0 || seeds[i].level >= HSO_N_PYR_LEVELS) return hso_fail(ctx, HSO_E_INVALID, "seed_observe: bad level");
    h[i].ref_base = last_base;
    h[i].cur_base = cur_base[seed_frame[i]];
    h[i].frame = seed_frame[i]; h[i].pad_ = 0;
    h[i].s = seeds[i];
  }
  auto al = [](size_t b) { return (b + 255) & ~size_t(255); };
  const size_t b_in = al((size_t)n_seeds * sizeof(SeedDev)), b_out = al((size_t)n_seeds * sizeof(hso_seed_out));
  const size_t need = b_in + b_out + al((size_t)n_frames * sizeof(SeedFrameDev));
  if (ctx->batch_cap < need) {
    HSO_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->d_batch) (void)hipFree(ctx->d_batch);
    ctx->d_batch = nullptr; ctx->batch_cap = 0;
    HSO_HIP_CHECK(ctx, hipMalloc(reinterpret_cast<void**>(&ctx->d_batch), hso_grown(need)));
    ctx->batch_cap = hso_grown(need);
  }
  SeedDev* d_in = reinterpret_cast<SeedDev*>(ctx->d_batch);
  hso_seed_out* d_out = reinterpret_cast<hso_seed_out*>(ctx->d_batch + b_in);
  SeedFrameDev* d_fr = reinterpret_cast<SeedFrameDev*>(ctx->d_batch + b_in + b_out);
  HSO_HIP_CHECK(ctx, hipMemcpyAsync(d_in, h, (size_t)n_seeds * sizeof(SeedDev), hipMemcpyHostToDevice, ctx->stream));
  HSO_HIP_CHECK(ctx, hipMemcpyAsync(d_fr, hf.data(), (size_t)n_frames * sizeof(SeedFrameDev), hipMemcpyHostToDevice, ctx->stream));
  SeedConsts C;
  C.cam = *cam; C.g = g; C.frames = d_fr; C.px_error_angle = px_error_angle; C.update_in_place = 0; C.brief = nullptr; C.px = nullptr; C.frame_keys = nullptr; C.n_frame_keys = 0;
  if (int rc = seed_observe_launch(ctx, C, d_in, n_seeds, d_out)) return rc;
  HSO_HIP_CHECK(ctx, hipMemcpyAsync(h_out, d_out, (size_t)n_seeds * sizeof(hso_seed_out), hipMemcpyDeviceToHost, ctx->stream));
  HSO_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  memcpy(out, h_out, (size_t)n_seeds * sizeof(hso_seed_out));
  return HSO_OK;
}

extern "C" int hso_gpu_seed_observe(hso_gpu_ctx* ctx, const hso_camera* cam, int64_t cur_frame_id, const hso_se3* cur_T_f_w,
                                    double cur_exposure, double px_error_angle, const hso_seed* seeds, int n_seeds,
                                    hso_seed_out* out)
{
  if (!ctx) return HSO_E_INVALID;
  if (!cam || !cur_T_f_w || n_seeds < 0 || (n_seeds > 0 && (!seeds || !out))) return hso_fail(ctx, HSO_E_INVALID, "seed_observe: bad argument");
  if (n_seeds == 0) return HSO_OK;
  hso_seed_frame f;
  f.frame_id = cur_frame_id; f.T_f_w = *cur_T_f_w; f.exposure_time = cur_exposure;
  const std::vector<int32_t> zero((size_t)n_seeds, 0);
  return hso_gpu_seed_observe_multi(ctx, cam, &f, 1, zero.data(), px_error_angle, seeds, n_seeds, out);
}

// ---------------------------------------------------------------------------------------------
// Resident seed tables: the seeds of a DepthFilter (of one sequence, or of many — a seed names its group) live in HBM
// between calls.  DepthFilter::initializeSeeds appends (src/depth_filter.cpp:164-205), updateSeeds' erase sites erase
// (:368-401, :423-497), and one observation of every live seed (observeDepth, :557-675) runs over the table in place: what
// crosses PCIe per frame is one hso_seed_frame per group in and one 16-byte hso_seed_brief per slot out, instead of the
// 216-byte seed record in and the 88-byte result out of the value-passing call.
static __global__ void k_seed_mark_dead(SeedDev* seeds, const int32_t* slots, int n)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) seeds[slots[i]].ref_base = nullptr;
}

struct SeedTable {
  SeedDev* d = nullptr;
  size_t cap = 0, n = 0;              // slots allocated / used (erased slots keep their index)
  std::vector<uint8_t> alive;
  std::vector<int64_t> host_frame;    // per slot: the frame the seed is hosted in (the device record caches that frame's base pointer)
  std::unordered_map<int64_t, int> pins;   // host frame id -> live seeds hosted there: such a frame must stay resident
  hso_seed_brief* d_brief = nullptr; size_t brief_cap = 0;
  hso_seed_out* d_full = nullptr; size_t full_cap = 0;
  float* d_px = nullptr; size_t px_cap = 0;
  SeedFrameDev* d_frames = nullptr; size_t frames_cap = 0;
  int64_t* d_keys = nullptr; size_t keys_cap = 0;   // previous-frame pass: host keyframe ids, ascending
  PyrGeom g{}; bool have_g = false;
  int max_group = -1;
};
struct SeedTables { std::vector<SeedTable*> t; };

void hso_seed_tables_free(hso_gpu_ctx* ctx)
{
  if (!ctx->seed_tables) return;
  for (SeedTable* t : ctx->seed_tables->t)
    if (t) { (void)hipFree(t->d); (void)hipFree(t->d_brief); (void)hipFree(t->d_full); (void)hipFree(t->d_frames); (void)hipFree(t->d_px); (void)hipFree(t->d_keys); delete t; }
  delete ctx->seed_tables;
  ctx->seed_tables = nullptr;
}

// hso_gpu_frame_release asks: a resident table caches the base pointer of every live seed's host frame, so releasing such a frame
// (its buffer goes back to the free list and to another frame id) would make the table read another frame's pyramid
bool hso_seed_tables_pin(hso_gpu_ctx* ctx, int64_t frame_id)
{
  if (!ctx->seed_tables) return false;
  for (SeedTable* t : ctx->seed_tables->t)
    if (t && t->pins.count(frame_id)) return true;
  return false;
}

static SeedTable* seed_table_of(hso_gpu_ctx* ctx, int table)
{
  if (!ctx->seed_tables || table < 0 || table >= (int)ctx->seed_tables->t.size()) return nullptr;
  return ctx->seed_tables->t[table];
}

// for the activation by slots (hso_activate.hip): the table's device rows — `ref_base` first, the hso_seed at *seed_offset — after
// the depth filter's own stream has drained; every named slot must hold a live seed
int hso_seed_table_rows(hso_gpu_ctx* ctx, int table, const int32_t* slots, int n, const char** rows, size_t* stride, size_t* seed_offset, PyrGeom* g)
{
  if (int rc = hso_seed_async_quiesce(ctx)) return rc;
  SeedTable* t = seed_table_of(ctx, table);
  if (!t || n < 0 || (n > 0 && !slots)) return hso_fail(ctx, HSO_E_INVALID, "seed_table_activate: bad argument");
  for (int i = 0; i < n; i++)
    if (slots[i] < 0 || (size_t)slots[i] >= t->n || !t->alive[(size_t)slots[i]]) return hso_fail(ctx, HSO_E_INVALID, "seed_table_activate: slot out of range or erased");
  if (n > 0 && !t->have_g) return hso_fail(ctx, HSO_E_INVALID, "seed_table_activate: empty table");
  static_assert(offsetof(SeedDev, ref_base) == 0, "hso_activate.hip reads the base pointer at the start of a row");
  *rows = reinterpret_cast<const char*>(t->d); *stride = sizeof(SeedDev); *seed_offset = offsetof(SeedDev, s); *g = t->g;
  return HSO_OK;
}

template <typename T> static int grow_dev(hso_gpu_ctx* ctx, T** p, size_t* cap, size_t need, size_t keep)
{
  if (*cap >= need) return HSO_OK;
  const size_t ncap = std::max(need, *cap * 2 + 1024);
  T* q = nullptr;
  HSO_HIP_CHECK(ctx, hipMalloc(reinterpret_cast<void**>(&q), ncap * sizeof(T)));
  if (*p && keep) {
    hipError_t e = hipMemcpyAsync(q, *p, keep * sizeof(T), hipMemcpyDeviceToDevice, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) { (void)hipFree(q); ctx->err = hipGetErrorString(e); return HSO_E_HIP; }
  } else if (*p) {
    HSO_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  }
  if (*p) (void)hipFree(*p);
  *p = q; *cap = ncap;
  return HSO_OK;
}

extern "C" {

int hso_gpu_seed_table_create(hso_gpu_ctx* ctx, int* table_out)
{
  if (!ctx || !table_out) return HSO_E_INVALID;
  if (!ctx->seed_tables) ctx->seed_tables = new SeedTables();
  ctx->seed_tables->t.push_back(new SeedTable());
  *table_out = (int)ctx->seed_tables->t.size() - 1;
  return HSO_OK;
}

int hso_gpu_seed_table_destroy(hso_gpu_ctx* ctx, int table)
{
  if (!ctx) return HSO_E_INVALID;
  if (int rc = hso_seed_async_quiesce(ctx)) return rc;
  SeedTable* t = seed_table_of(ctx, table);
  if (!t) return hso_fail(ctx, HSO_E_INVALID, "seed_table: no such table");
  HSO_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  (void)hipFree(t->d); (void)hipFree(t->d_brief); (void)hipFree(t->d_full); (void)hipFree(t->d_frames); (void)hipFree(t->d_px); (void)hipFree(t->d_keys);
  delete t;
  ctx->seed_tables->t[table] = nullptr;
  return HSO_OK;
}

int hso_gpu_seed_table_append(hso_gpu_ctx* ctx, int table, const hso_seed* seeds, const int32_t* group, int n, int32_t* first_slot)
{
  if (!ctx) return HSO_E_INVALID;
  if (int rc = hso_seed_async_quiesce(ctx)) return rc;
  SeedTable* t = seed_table_of(ctx, table);
  if (!t || n < 0 || (n > 0 && !seeds)) return hso_fail(ctx, HSO_E_INVALID, "seed_table_append: bad argument");
  if (first_slot) *first_slot = (int32_t)t->n;
  if (n == 0) return HSO_OK;
  HSO_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  SeedDev* h = reinterpret_cast<SeedDev*>(hso_pinned(ctx, 0, (size_t)n * sizeof(SeedDev)));
  if (!h) return HSO_E_NOMEM;
  int64_t last_id = -1;
  const uint8_t* last_base = nullptr;
  for (int i = 0; i < n; i++) {
    if (i == 0 || seeds[i].ref_frame_id != last_id) {
      auto itr = ctx->frames.find(seeds[i].ref_frame_id);
      if (itr == ctx->frames.end()) return hso_fail(ctx, HSO_E_NOFRAME, "seed_table_append: seed host frame not resident");
      if (!t->have_g) { t->g = itr->second.g; t->have_g = true; }
      if (!same_geom(itr->second.g, t->g)) return hso_fail(ctx, HSO_E_INVALID, "seed_table_append: frames must share one size");
      last_id = seeds[i].ref_frame_id; last_base = itr->second.base;
    }
    if (seeds[i].level < 0 || seeds[i].level >= HSO_N_PYR_LEVELS) return hso_fail(ctx, HSO_E_INVALID, "seed_table_append: bad level");
    const int gi = group ? group[i] : 0;
    if (gi < 0) return hso_fail(ctx, HSO_E_INVALID, "seed_table_append: negative group");
    h[i].ref_base = last_base; h[i].cur_base = nullptr; h[i].frame = gi; h[i].pad_ = 0; h[i].s = seeds[i];
    t->max_group = std::max(t->max_group, gi);
  }
  if (int rc = grow_dev(ctx, &t->d, &t->cap, t->n + (size_t)n, t->n)) return rc;
  HSO_HIP_CHECK(ctx, hipMemcpyAsync(t->d + t->n, h, (size_t)n * sizeof(SeedDev), hipMemcpyHostToDevice, ctx->stream));
  HSO_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));   // the pinned staging buffer is reused by the next call
  t->n += (size_t)n;
  t->alive.resize(t->n, 1);
  for (int i = 0; i < n; i++) { t->host_frame.push_back(seeds[i].ref_frame_id); t->pins[seeds[i].ref_frame_id]++; }
  return HSO_OK;
}

int hso_gpu_seed_table_erase(hso_gpu_ctx* ctx, int table, const int32_t* slots, int n)
{
  if (!ctx) return HSO_E_INVALID;
  if (int rc = hso_seed_async_quiesce(ctx)) return rc;
  SeedTable* t = seed_table_of(ctx, table);
  if (!t || n < 0 || (n > 0 && !slots)) return hso_fail(ctx, HSO_E_INVALID, "seed_table_erase: bad argument");
  for (int i = 0; i < n; i++)
    if (slots[i] < 0 || (size_t)slots[i] >= t->n) return hso_fail(ctx, HSO_E_INVALID, "seed_table_erase: slot out of range");
  std::vector<int32_t> dead;
  for (int i = 0; i < n; i++) {
    if (!t->alive[slots[i]]) continue;
    t->alive[slots[i]] = 0;
    if (--t->pins[t->host_frame[slots[i]]] <= 0) t->pins.erase(t->host_frame[slots[i]]);
    dead.push_back(slots[i]);
  }
  if (dead.empty()) return HSO_OK;
  // a null ref_base marks a slot dead for the kernels: ONE slot list + one small kernel (a copy per slot was ~900 eight-byte
  // copies per step of 64 sequences: 3 ms of copy kernels and as much enqueue time)
  HSO_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  const size_t need = sizeof(int32_t) * dead.size();
  if (ctx->batch_cap < need) {
    HSO_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->d_batch) (void)hipFree(ctx->d_batch);
    ctx->d_batch = nullptr; ctx->batch_cap = 0;
    HSO_HIP_CHECK(ctx, hipMalloc(reinterpret_cast<void**>(&ctx->d_batch), hso_grown(need)));
    ctx->batch_cap = hso_grown(need);
  }
  HSO_HIP_CHECK(ctx, hipMemcpyAsync(ctx->d_batch, dead.data(), need, hipMemcpyHostToDevice, ctx->stream));
  hipLaunchKernelGGL(k_seed_mark_dead, dim3((unsigned)((dead.size() + 255) / 256)), dim3(256), 0, ctx->stream, t->d, reinterpret_cast<const int32_t*>(ctx->d_batch),
                     (int)dead.size());
  HSO_HIP_CHECK(ctx, hipGetLastError());
  // no wait: the slot list was staged by the copy wrapper, nothing comes back, and every reader of the table runs on this stream
  return HSO_OK;
}

}  // extern "C"

// local BA moved a keyframe: the live seeds hosted there take its new pose
static __global__ void k_seed_set_host_pose(SeedDev* seeds, int n_slots, const int64_t* ids, const hso_se3* poses, int n)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_slots || seeds[i].ref_base == nullptr) return;
  const int64_t id = seeds[i].s.ref_frame_id;
  for (int k = 0; k < n; k++)
    if (ids[k] == id) { seeds[i].s.T_ref_w = poses[k]; return; }
}

// dst[i] = src[idx[i]]: 16 bytes per thread, a record = sizeof(SeedDev) / 16 threads
static __global__ void k_seed_gather(const uint4* __restrict__ src, const int* __restrict__ idx, uint4* __restrict__ dst, int n_live)
{
  constexpr int Q = sizeof(SeedDev) / 16;
  const size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= (size_t)n_live * Q) return;
  const size_t i = g / Q, q = g - i * Q;
  dst[i * Q + q] = src[(size_t)idx[i] * Q + q];
}
static_assert(sizeof(SeedDev) % 16 == 0, "k_seed_gather copies a slot in 16-byte pieces");

extern "C" {

int hso_gpu_seed_table_compact(hso_gpu_ctx* ctx, int table, int32_t* remap)
{
  if (!ctx) return HSO_E_INVALID;
  if (int rc = hso_seed_async_quiesce(ctx)) return rc;
  SeedTable* t = seed_table_of(ctx, table);
  if (!t) return hso_fail(ctx, HSO_E_INVALID, "seed_table_compact: no such table");
  std::vector<int> idx;
  idx.reserve(t->n);
  for (size_t i = 0; i < t->n; i++) {
    if (remap) remap[i] = t->alive[i] ? (int32_t)idx.size() : -1;
    if (t->alive[i]) idx.push_back((int)i);
  }
  if (idx.size() == t->n) return (int)t->n;                      // nothing erased
  HSO_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  const size_t n_live = idx.size();
  if (n_live) {
    SeedDev* q = nullptr;
    int* d_idx = nullptr;
    const size_t ncap = n_live + 1024;
    HSO_HIP_CHECK(ctx, hipMalloc(reinterpret_cast<void**>(&q), ncap * sizeof(SeedDev)));
    if (hipMalloc(reinterpret_cast<void**>(&d_idx), n_live * sizeof(int)) != hipSuccess) { (void)hipFree(q); return hso_fail(ctx, HSO_E_NOMEM, "seed_table_compact"); }
    hipError_t e = hipMemcpyAsync(d_idx, idx.data(), n_live * sizeof(int), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) {
      const size_t threads = n_live * (sizeof(SeedDev) / 16);
      k_seed_gather<<<(unsigned)((threads + 255) / 256), 256, 0, ctx->stream>>>(reinterpret_cast<const uint4*>(t->d), d_idx, reinterpret_cast<uint4*>(q), (int)n_live);
      e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    (void)hipFree(d_idx);
    if (e != hipSuccess) { (void)hipFree(q); ctx->err = std::string("seed_table_compact: ") + hipGetErrorString(e); return HSO_E_HIP; }
    (void)hipFree(t->d);
    t->d = q; t->cap = ncap;
  } else {
    HSO_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  }
  std::vector<int64_t> hf(n_live);
  for (size_t k = 0; k < n_live; k++) hf[k] = t->host_frame[idx[k]];
  t->host_frame.swap(hf);
  t->alive.assign(n_live, 1);
  t->n = n_live;
  return (int)n_live;
}

int hso_gpu_seed_table_size(hso_gpu_ctx* ctx, int table, int* n_slots, int* n_live)
{
  if (!ctx) return HSO_E_INVALID;
  SeedTable* t = seed_table_of(ctx, table);
  if (!t) return hso_fail(ctx, HSO_E_INVALID, "seed_table: no such table");
  if (n_slots) *n_slots = (int)t->n;
  if (n_live) { int c = 0; for (uint8_t a : t->alive) c += a; *n_live = c; }
  return HSO_OK;
}

static int seed_table_observe_impl(hso_gpu_ctx* ctx, const hso_camera* cam, int table, const hso_seed_frame* frames, int n_frames,
                                   double px_error_angle, hso_seed_brief* brief_out, float* px_out, hso_seed_out* full_out, bool allow_skip)
{
  if (!ctx) return HSO_E_INVALID;
  if (int rc = hso_seed_async_quiesce(ctx)) return rc;
  SeedTable* t = seed_table_of(ctx, table);
  if (!t || !cam || !frames || n_frames <= 0) return hso_fail(ctx, HSO_E_INVALID, "seed_table_observe: bad argument");
  if (t->n == 0) return HSO_OK;
  if (t->max_group >= n_frames) return hso_fail(ctx, HSO_E_INVALID, "seed_table_observe: a seed's group has no frame");
  HSO_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  std::vector<SeedFrameDev> hf(n_frames);
  for (int k = 0; k < n_frames; k++) {
    if (allow_skip && frames[k].frame_id < 0) { hf[k].T_f_w = frames[k].T_f_w; hf[k].exposure = 1.0; hf[k].cur_base = nullptr; continue; }
    auto itc = ctx->frames.find(frames[k].frame_id);
    if (itc == ctx->frames.end()) return hso_fail(ctx, HSO_E_NOFRAME, "seed_table_observe: active frame not resident");
    if (!same_geom(itc->second.g, t->g)) return hso_fail(ctx, HSO_E_INVALID, "seed_table_observe: frames must share one size");
    hf[k].T_f_w = frames[k].T_f_w; hf[k].exposure = frames[k].exposure_time; hf[k].cur_base = itc->second.base;
  }
  if (cam->width != t->g.w[0] || cam->height != t->g.h[0]) return hso_fail(ctx, HSO_E_INVALID, "seed_table_observe: camera size differs from the frame size");
  if (int rc = grow_dev(ctx, &t->d_frames, &t->frames_cap, (size_t)n_frames, 0)) return rc;
  if (int rc = grow_dev(ctx, &t->d_brief, &t->brief_cap, t->n, 0)) return rc;
  if (full_out) if (int rc = grow_dev(ctx, &t->d_full, &t->full_cap, t->n, 0)) return rc;
  if (px_out) if (int rc = grow_dev(ctx, &t->d_px, &t->px_cap, 2 * t->n, 0)) return rc;
  SeedFrameDev* hfp = reinterpret_cast<SeedFrameDev*>(hso_pinned(ctx, 0, (size_t)n_frames * sizeof(SeedFrameDev)));
  if (!hfp) return HSO_E_NOMEM;
  memcpy(hfp, hf.data(), (size_t)n_frames * sizeof(SeedFrameDev));
  HSO_HIP_CHECK(ctx, hipMemcpyAsync(t->d_frames, hfp, (size_t)n_frames * sizeof(SeedFrameDev), hipMemcpyHostToDevice, ctx->stream));
  SeedConsts C;
  C.cam = *cam; C.g = t->g; C.frames = t->d_frames; C.px_error_angle = px_error_angle; C.update_in_place = 1; C.brief = t->d_brief; C.px = px_out ? t->d_px : nullptr; C.frame_keys = nullptr; C.n_frame_keys = 0;
  const int n = (int)t->n;
  if (int rc = seed_observe_launch(ctx, C, t->d, n, full_out ? t->d_full : nullptr)) return rc;
  if (px_out) HSO_HIP_CHECK(ctx, hipMemcpyAsync(px_out, t->d_px, 2 * t->n * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
  if (brief_out) {
    hso_seed_brief* hb = reinterpret_cast<hso_seed_brief*>(hso_pinned(ctx, 1, t->n * sizeof(hso_seed_brief)));
    if (!hb) return HSO_E_NOMEM;
    HSO_HIP_CHECK(ctx, hipMemcpyAsync(hb, t->d_brief, t->n * sizeof(hso_seed_brief), hipMemcpyDeviceToHost, ctx->stream));
    if (full_out) HSO_HIP_CHECK(ctx, hipMemcpyAsync(full_out, t->d_full, t->n * sizeof(hso_seed_out), hipMemcpyDeviceToHost, ctx->stream));
    HSO_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    memcpy(brief_out, hb, t->n * sizeof(hso_seed_brief));
  } else {
    if (full_out) HSO_HIP_CHECK(ctx, hipMemcpyAsync(full_out, t->d_full, t->n * sizeof(hso_seed_out), hipMemcpyDeviceToHost, ctx->stream));
    HSO_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  }
  return HSO_OK;
}

}  // extern "C"

int hso_seed_table_chain_frames(hso_gpu_ctx* ctx, int table, int n_groups, SeedFrameDev** d_frames)
{
  if (int rc = hso_seed_async_quiesce(ctx)) return rc;
  SeedTable* t = seed_table_of(ctx, table);
  if (!t || n_groups <= 0) return hso_fail(ctx, HSO_E_INVALID, "seq_chain: no such seed table");
  if (t->max_group >= n_groups) return hso_fail(ctx, HSO_E_INVALID, "seq_chain: a seed's group lies beyond n_seed_groups");
  if (int rc = grow_dev(ctx, &t->d_frames, &t->frames_cap, (size_t)n_groups, 0)) return rc;
  HSO_HIP_CHECK(ctx, hipMemsetAsync(t->d_frames, 0, (size_t)n_groups * sizeof(SeedFrameDev), ctx->stream));   // cur_base = null: every group sits out
  *d_frames = t->d_frames;
  return HSO_OK;
}

int hso_seed_table_chain_launch(hso_gpu_ctx* ctx, const hso_camera* cam, int table, int n_groups, double px_error_angle, const hso_seed_brief** d_brief, int* n_slots)
{
  SeedTable* t = seed_table_of(ctx, table);
  if (!t) return hso_fail(ctx, HSO_E_INVALID, "seq_chain: no such seed table");
  *n_slots = (int)t->n; *d_brief = nullptr;
  if (t->n == 0) return HSO_OK;
  if (cam->width != t->g.w[0] || cam->height != t->g.h[0]) return hso_fail(ctx, HSO_E_INVALID, "seq_chain: camera size differs from the seeds' frame size");
  if (int rc = grow_dev(ctx, &t->d_brief, &t->brief_cap, t->n, 0)) return rc;
  SeedConsts C;
  C.cam = *cam; C.g = t->g; C.frames = t->d_frames; C.px_error_angle = px_error_angle; C.update_in_place = 1; C.brief = t->d_brief; C.px = nullptr; C.frame_keys = nullptr; C.n_frame_keys = 0;
  if (int rc = seed_observe_launch(ctx, C, t->d, (int)t->n, nullptr)) return rc;
  *d_brief = t->d_brief;
  (void)n_groups;
  return HSO_OK;
}

extern "C" {

int hso_gpu_seed_table_observe(hso_gpu_ctx* ctx, const hso_camera* cam, int table, const hso_seed_frame* frames, int n_frames,
                               double px_error_angle, hso_seed_brief* brief_out, hso_seed_out* full_out)
{
  return seed_table_observe_impl(ctx, cam, table, frames, n_frames, px_error_angle, brief_out, nullptr, full_out, false);
}

int hso_gpu_seed_table_observe_groups(hso_gpu_ctx* ctx, const hso_camera* cam, int table, const hso_seed_frame* frames, int n_frames,
                                      double px_error_angle, hso_seed_brief* brief_out, float* px_out, hso_seed_out* full_out)
{
  return seed_table_observe_impl(ctx, cam, table, frames, n_frames, px_error_angle, brief_out, px_out, full_out, true);
}

// the arguments of a previous-frame pass checked and staged: (keyframe, earlier frame) pairs sorted by keyframe id — the pre kernel
// finds a seed's entry by bisection — into `stage` (page-locked): [SeedFrameDev x n | int64 x n]
static int seed_previous_stage(hso_gpu_ctx* ctx, SeedTable* t, const hso_camera* cam, const int64_t* host_frame_ids, const hso_seed_frame* pre_frames, int n,
                               char* stage, size_t b_fr)
{
  if (cam->width != t->g.w[0] || cam->height != t->g.h[0]) return hso_fail(ctx, HSO_E_INVALID, "seed_table_observe_previous: camera size differs from the frame size");
  std::vector<int> order(n);
  for (int k = 0; k < n; k++) order[k] = k;
  std::sort(order.begin(), order.end(), [&](int a, int b) { return host_frame_ids[a] < host_frame_ids[b]; });
  SeedFrameDev* hf = reinterpret_cast<SeedFrameDev*>(stage);
  int64_t* hk = reinterpret_cast<int64_t*>(stage + b_fr);
  for (int k = 0; k < n; k++) {
    const int src = order[k];
    if (k > 0 && host_frame_ids[src] == hk[k - 1]) return hso_fail(ctx, HSO_E_INVALID, "seed_table_observe_previous: a keyframe is listed twice");
    auto itc = ctx->frames.find(pre_frames[src].frame_id);
    if (itc == ctx->frames.end()) return hso_fail(ctx, HSO_E_NOFRAME, "seed_table_observe_previous: earlier frame not resident");
    if (!same_geom(itc->second.g, t->g)) return hso_fail(ctx, HSO_E_INVALID, "seed_table_observe_previous: frames must share one size");
    hk[k] = host_frame_ids[src];
    hf[k].T_f_w = pre_frames[src].T_f_w; hf[k].exposure = pre_frames[src].exposure_time; hf[k].cur_base = itc->second.base;
  }
  return HSO_OK;
}

int hso_gpu_seed_table_observe_previous(hso_gpu_ctx* ctx, const hso_camera* cam, int table, const int64_t* host_frame_ids,
                                        const hso_seed_frame* pre_frames, int n, double px_error_angle, hso_seed_brief* brief_out,
                                        hso_seed_out* full_out)
{
  if (!ctx) return HSO_E_INVALID;
  if (int rc = hso_seed_async_quiesce(ctx)) return rc;
  if (ctx->seed_inflight) return hso_fail(ctx, HSO_E_INVALID, "seed_table_observe_previous: a pass started with _begin has not been collected");
  SeedTable* t = seed_table_of(ctx, table);
  if (!t || !cam || n < 0 || (n > 0 && (!host_frame_ids || !pre_frames))) return hso_fail(ctx, HSO_E_INVALID, "seed_table_observe_previous: bad argument");
  if (t->n == 0) return HSO_OK;
  HSO_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  const size_t b_fr = ((size_t)std::max(n, 1) * sizeof(SeedFrameDev) + 255) & ~size_t(255);
  char* stage = reinterpret_cast<char*>(hso_pinned(ctx, 0, b_fr + (size_t)std::max(n, 1) * sizeof(int64_t)));
  if (!stage) return HSO_E_NOMEM;
  if (int rc = seed_previous_stage(ctx, t, cam, host_frame_ids, pre_frames, n, stage, b_fr)) return rc;
  if (int rc = grow_dev(ctx, &t->d_frames, &t->frames_cap, (size_t)std::max(n, 1), 0)) return rc;
  if (int rc = grow_dev(ctx, &t->d_keys, &t->keys_cap, (size_t)std::max(n, 1), 0)) return rc;
  if (int rc = grow_dev(ctx, &t->d_brief, &t->brief_cap, t->n, 0)) return rc;
  if (full_out) if (int rc = grow_dev(ctx, &t->d_full, &t->full_cap, t->n, 0)) return rc;
  if (n > 0) {
    HSO_HIP_CHECK(ctx, hipMemcpyAsync(t->d_frames, stage, (size_t)n * sizeof(SeedFrameDev), hipMemcpyHostToDevice, ctx->stream));
    HSO_HIP_CHECK(ctx, hipMemcpyAsync(t->d_keys, stage + b_fr, (size_t)n * sizeof(int64_t), hipMemcpyHostToDevice, ctx->stream));
  }
  SeedConsts C;
  C.cam = *cam; C.g = t->g; C.frames = t->d_frames; C.px_error_angle = px_error_angle; C.update_in_place = 1; C.brief = t->d_brief; C.px = nullptr;
  C.frame_keys = t->d_keys; C.n_frame_keys = n;
  if (int rc = seed_observe_launch(ctx, C, t->d, (int)t->n, full_out ? t->d_full : nullptr)) return rc;
  hso_seed_brief* hb = nullptr;
  if (brief_out) {
    hb = reinterpret_cast<hso_seed_brief*>(hso_pinned(ctx, 1, t->n * sizeof(hso_seed_brief)));
    if (!hb) return HSO_E_NOMEM;
    HSO_HIP_CHECK(ctx, hipMemcpyAsync(hb, t->d_brief, t->n * sizeof(hso_seed_brief), hipMemcpyDeviceToHost, ctx->stream));
  }
  if (full_out) HSO_HIP_CHECK(ctx, hipMemcpyAsync(full_out, t->d_full, t->n * sizeof(hso_seed_out), hipMemcpyDeviceToHost, ctx->stream));
  HSO_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  if (brief_out) memcpy(brief_out, hb, t->n * sizeof(hso_seed_brief));
  return HSO_OK;
}

int hso_gpu_seed_table_observe_previous_begin(hso_gpu_ctx* ctx, const hso_camera* cam, int table, const int64_t* host_frame_ids,
                                              const hso_seed_frame* pre_frames, int n, double px_error_angle)
{
  if (!ctx) return HSO_E_INVALID;
  if (ctx->seed_inflight) return hso_fail(ctx, HSO_E_INVALID, "seed_table_observe_previous_begin: the previous pass has not been collected");
  SeedTable* t = seed_table_of(ctx, table);
  if (!t || !cam || n < 0 || (n > 0 && (!host_frame_ids || !pre_frames))) return hso_fail(ctx, HSO_E_INVALID, "seed_table_observe_previous_begin: bad argument");
  HSO_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  if (!ctx->seed_stream) {
    HSO_HIP_CHECK(ctx, hipStreamCreateWithFlags(&ctx->seed_stream, hipStreamNonBlocking));
    HSO_HIP_CHECK(ctx, hipEventCreateWithFlags(&ctx->seed_go, hipEventDisableTiming));
    HSO_HIP_CHECK(ctx, hipEventCreateWithFlags(&ctx->seed_done, hipEventDisableTiming));
  }
  ctx->seed_inflight_table = table; ctx->seed_inflight_n = t->n;
  if (t->n == 0) { ctx->seed_inflight = true; return HSO_OK; }   // nothing to launch; _end reports nothing
  // page-locked staging of its own (the context's slots are reused by the calls that overlap this pass): [frames | keys | briefs]
  const size_t b_fr = ((size_t)std::max(n, 1) * sizeof(SeedFrameDev) + 255) & ~size_t(255);
  const size_t b_key = ((size_t)std::max(n, 1) * sizeof(int64_t) + 255) & ~size_t(255);
  const size_t need = b_fr + b_key + t->n * sizeof(hso_seed_brief);
  if (ctx->h_seed_pin_cap < need) {
    if (ctx->h_seed_pin) (void)hipHostFree(ctx->h_seed_pin);
    ctx->h_seed_pin = nullptr; ctx->h_seed_pin_cap = 0;
    HSO_HIP_CHECK(ctx, hipHostMalloc(reinterpret_cast<void**>(&ctx->h_seed_pin), hso_grown(need), hipHostMallocDefault));
    ctx->h_seed_pin_cap = hso_grown(need);
  }
  if (int rc = seed_previous_stage(ctx, t, cam, host_frame_ids, pre_frames, n, ctx->h_seed_pin, b_fr)) return rc;
  if (int rc = grow_dev(ctx, &t->d_frames, &t->frames_cap, (size_t)std::max(n, 1), 0)) return rc;
  if (int rc = grow_dev(ctx, &t->d_keys, &t->keys_cap, (size_t)std::max(n, 1), 0)) return rc;
  if (int rc = grow_dev(ctx, &t->d_brief, &t->brief_cap, t->n, 0)) return rc;
  // everything queued on the context's stream so far (appends, erases, pose updates of this table, uploads of the frames) first.
  // From here on work may be queued on the depth filter's stream: a failing exit waits for it before it returns, so no later entry
  // point (which would skip hso_seed_async_quiesce: seed_inflight stays false) can meet an in-place update still running.
  const int rc = [&]() -> int {
    HSO_HIP_CHECK(ctx, hipEventRecord(ctx->seed_go, ctx->stream));
    HSO_HIP_CHECK(ctx, hipStreamWaitEvent(ctx->seed_stream, ctx->seed_go, 0));
    if (n > 0) {
      HSO_HIP_CHECK(ctx, hipMemcpyAsync(t->d_frames, ctx->h_seed_pin, (size_t)n * sizeof(SeedFrameDev), hipMemcpyHostToDevice, ctx->seed_stream));
      HSO_HIP_CHECK(ctx, hipMemcpyAsync(t->d_keys, ctx->h_seed_pin + b_fr, (size_t)n * sizeof(int64_t), hipMemcpyHostToDevice, ctx->seed_stream));
    }
    SeedConsts C;
    C.cam = *cam; C.g = t->g; C.frames = t->d_frames; C.px_error_angle = px_error_angle; C.update_in_place = 1; C.brief = t->d_brief; C.px = nullptr;
    C.frame_keys = t->d_keys; C.n_frame_keys = n;
    if (int rc = seed_observe_launch_on(ctx, ctx->seed_stream, &ctx->d_seed_scratch_async, &ctx->seed_scratch_async_cap, C, t->d, (int)t->n, nullptr)) return rc;
    ctx->h_seed_brief_off = b_fr + b_key;
    HSO_HIP_CHECK(ctx, hipMemcpyAsync(ctx->h_seed_pin + b_fr + b_key, t->d_brief, t->n * sizeof(hso_seed_brief), hipMemcpyDeviceToHost, ctx->seed_stream));
    HSO_HIP_CHECK(ctx, hipEventRecord(ctx->seed_done, ctx->seed_stream));
    return HSO_OK;
  }();
  if (rc != HSO_OK) { (void)hso_stream_sync(ctx->seed_stream); return rc; }
  ctx->seed_inflight = true;
  return HSO_OK;
}

int hso_gpu_seed_table_observe_previous_end(hso_gpu_ctx* ctx, int table, hso_seed_brief* brief_out, int n_brief)
{
  if (!ctx) return HSO_E_INVALID;
  if (!ctx->seed_inflight) return hso_fail(ctx, HSO_E_INVALID, "seed_table_observe_previous_end: no pass in flight");
  if (table != ctx->seed_inflight_table) return hso_fail(ctx, HSO_E_INVALID, "seed_table_observe_previous_end: the pass in flight is another table's");
  if (brief_out && (size_t)n_brief < ctx->seed_inflight_n) return hso_fail(ctx, HSO_E_INVALID, "seed_table_observe_previous_end: brief_out is smaller than the table was at _begin");
  const size_t n = ctx->seed_inflight_n;
  if (n > 0) {
    HSO_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    HSO_HIP_CHECK(ctx, hipEventSynchronize(ctx->seed_done));
    // the context's stream continues behind the pass (the table's records changed)
    HSO_HIP_CHECK(ctx, hipStreamWaitEvent(ctx->stream, ctx->seed_done, 0));
    if (brief_out) {
      // the briefs sit behind the staged frames and keys: their offsets are those _begin used for this pass
      SeedTable* t = seed_table_of(ctx, table);
      (void)t;
      memcpy(brief_out, ctx->h_seed_pin + ctx->h_seed_brief_off, n * sizeof(hso_seed_brief));
    }
  }
  ctx->seed_inflight = false;
  return HSO_OK;
}

int hso_gpu_seed_table_set_host_pose(hso_gpu_ctx* ctx, int table, const int64_t* frame_ids, const hso_se3* T_f_w, int n)
{
  if (!ctx) return HSO_E_INVALID;
  if (int rc = hso_seed_async_quiesce(ctx)) return rc;
  SeedTable* t = seed_table_of(ctx, table);
  if (!t || n < 0 || (n > 0 && (!frame_ids || !T_f_w))) return hso_fail(ctx, HSO_E_INVALID, "seed_table_set_host_pose: bad argument");
  if (n == 0 || t->n == 0) return HSO_OK;
  HSO_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  auto al = [](size_t b) { return (b + 255) & ~size_t(255); };
  const size_t b_id = al(sizeof(int64_t) * (size_t)n), need = b_id + al(sizeof(hso_se3) * (size_t)n);
  if (ctx->batch_cap < need) {
    HSO_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->d_batch) (void)hipFree(ctx->d_batch);
    ctx->d_batch = nullptr; ctx->batch_cap = 0;
    HSO_HIP_CHECK(ctx, hipMalloc(reinterpret_cast<void**>(&ctx->d_batch), hso_grown(need)));
    ctx->batch_cap = hso_grown(need);
  }
  HSO_HIP_CHECK(ctx, hipMemcpyAsync(ctx->d_batch, frame_ids, sizeof(int64_t) * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
  HSO_HIP_CHECK(ctx, hipMemcpyAsync(ctx->d_batch + b_id, T_f_w, sizeof(hso_se3) * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
  hipLaunchKernelGGL(k_seed_set_host_pose, dim3((unsigned)((t->n + 255) / 256)), dim3(256), 0, ctx->stream, t->d, (int)t->n,
                     reinterpret_cast<const int64_t*>(ctx->d_batch), reinterpret_cast<const hso_se3*>(ctx->d_batch + b_id), n);
  HSO_HIP_CHECK(ctx, hipGetLastError());
  HSO_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  return HSO_OK;
}

int hso_gpu_seed_table_read(hso_gpu_ctx* ctx, int table, int first, int n, hso_seed* seeds_out)
{
  if (!ctx) return HSO_E_INVALID;
  if (int rc = hso_seed_async_quiesce(ctx)) return rc;
  SeedTable* t = seed_table_of(ctx, table);
  if (!t || first < 0 || n < 0 || (size_t)first + (size_t)n > t->n || (n > 0 && !seeds_out)) return hso_fail(ctx, HSO_E_INVALID, "seed_table_read: bad argument");
  if (n == 0) return HSO_OK;
  std::vector<SeedDev> h((size_t)n);
  HSO_HIP_CHECK(ctx, hipMemcpyAsync(h.data(), t->d + first, (size_t)n * sizeof(SeedDev), hipMemcpyDeviceToHost, ctx->stream));
  HSO_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  for (int i = 0; i < n; i++) seeds_out[i] = h[i].s;
  return HSO_OK;
}

}  // extern "C"
