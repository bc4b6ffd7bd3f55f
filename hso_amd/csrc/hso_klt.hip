// hso_klt.hip — the image side of the two-view initialisation on the device: initialization::trackKlt
// (reference src/initialization.cpp:225-300) = cv::calcOpticalFlowPyrLK(img_prev, img_cur, px_prev, px_cur, .., Size(30, 30), 4,
// TermCriteria(COUNT + EPS, 30, 1e-4), OPTFLOW_USE_INITIAL_FLOW) followed by patchCheck (:476-563) per point.
//
// OpenCV is not part of the reference tree; the arithmetic follows its published pyramidal LK (14-bit fixed-point bilinear
// windows, Scharr derivatives, Gaussian 5x5 pyramid; oracle/hso_oracle_klt.c restates it on the CPU and states what is and is
// not pinned).  Data flow per call:
//   k_pyr_down   levels 1..L of both frames' Gaussian pyramids (level 0 = the resident frame's level-0 image, not copied)
//   k_scharr     interleaved int16 (Ix, Iy) of every level of the PREVIOUS frame
//   k_klt        one 256-thread workgroup per point, coarse to fine: the 30x30 template window and its gradients live in LDS as
//                int16 (5.4 KB), the 2x2 matrix and the right-hand side are summed as exact 64-bit integers (OpenCV's float
//                accumulation order differs between its scalar and SIMD paths; the exact sum is the value both approximate),
//                then the 8x8 patch check on level 0.
// All of it is HBM/L2-latency bound integer work at a few hundred microseconds per call and runs a handful of times per
// sequence (until the median disparity reaches Config::initMinDisparity); it is here so that the product path starts from two
// images without leaving the device interface, not because it is hot.
#include "hso_ctx.h"

namespace {

constexpr int KLT_THREADS = 256;
constexpr int KLT_MAX_WIN = 32;
constexpr int KLT_MAX_LEVELS = 8;

__device__ __forceinline__ int reflect101(int i, int n)
{
  if (n == 1) return 0;
  while (i < 0 || i >= n) i = i < 0 ? -i : 2 * n - 2 - i;
  return i;
}

// cv::pyrDown, 8-bit, BORDER_REFLECT_101: [1 4 6 4 1]^2 / 256 with rounding
__global__ void k_pyr_down(const uint8_t* __restrict__ src, int w, int h, uint8_t* __restrict__ dst, int dw, int dh)
{
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y;
  if (x >= dw || y >= dh) return;
  int xs[5];
  for (int k = 0; k < 5; k++) xs[k] = reflect101(2 * x - 2 + k, w);
  const int wk[5] = {1, 4, 6, 4, 1};
  int v = 0;
  for (int k = 0; k < 5; k++) {
    const uint8_t* s = src + (size_t)reflect101(2 * y - 2 + k, h) * w;
    v += wk[k] * (s[xs[2]] * 6 + (s[xs[1]] + s[xs[3]]) * 4 + s[xs[0]] + s[xs[4]]);
  }
  dst[(size_t)y * dw + x] = (uint8_t)((v + 128) >> 8);
}

// calcSharrDeriv: d[2 * (y * w + x)] = Ix, + 1 = Iy
__global__ void k_scharr(const uint8_t* __restrict__ src, int w, int h, short2* __restrict__ d)
{
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y;
  if (x >= w || y >= h) return;
  const uint8_t* r0 = src + (size_t)reflect101(y - 1, h) * w;
  const uint8_t* r1 = src + (size_t)y * w;
  const uint8_t* r2 = src + (size_t)reflect101(y + 1, h) * w;
  const int xm = reflect101(x - 1, w), xp = reflect101(x + 1, w);
  const int t0m = (r0[xm] + r2[xm]) * 3 + r1[xm] * 10, t0p = (r0[xp] + r2[xp]) * 3 + r1[xp] * 10;
  const int t1m = r2[xm] - r0[xm], t1c = r2[x] - r0[x], t1p = r2[xp] - r0[xp];
  d[(size_t)y * w + x] = make_short2((short)(t0p - t0m), (short)((t1m + t1p) * 3 + t1c * 10));
}

struct KltLevels {
  const uint8_t* prev[KLT_MAX_LEVELS];
  const uint8_t* cur[KLT_MAX_LEVELS];
  const short2* deriv[KLT_MAX_LEVELS];
  int w[KLT_MAX_LEVELS], h[KLT_MAX_LEVELS];
  int last;       // index of the coarsest level
};

__device__ __forceinline__ long long wave_sum_i64(long long v)
{
  for (int o = 32; o > 0; o >>= 1) {
    const int lo = __shfl_xor((int)(v & 0xffffffffll), o), hi = __shfl_xor((int)(v >> 32), o);
    v += ((long long)hi << 32) | (unsigned)lo;
  }
  return v;
}

#define DESCALE(x, n) (((x) + (1 << ((n) - 1))) >> (n))

struct KltShared {
  short I[KLT_MAX_WIN * KLT_MAX_WIN];
  short2 dI[KLT_MAX_WIN * KLT_MAX_WIN];
  long long part[3][KLT_THREADS / 64];
  float patch[2][64];
  int ok[2];
};

// block-wide exact sums of up to three 64-bit values; every thread receives the totals
__device__ __forceinline__ void block_sum3(KltShared& s, long long& a, long long& b, long long& c)
{
  a = wave_sum_i64(a); b = wave_sum_i64(b); c = wave_sum_i64(c);
  const int wv = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) { s.part[0][wv] = a; s.part[1][wv] = b; s.part[2][wv] = c; }
  __syncthreads();
  a = b = c = 0;
  for (int k = 0; k < KLT_THREADS / 64; k++) { a += s.part[0][k]; b += s.part[1][k]; c += s.part[2][k]; }
}

__device__ __forceinline__ void bilinear_weights(float a, float b, int& iw00, int& iw01, int& iw10, int& iw11)
{
  const int W_BITS = 14;
  iw00 = __float2int_rn((1.f - a) * (1.f - b) * (1 << W_BITS));
  iw01 = __float2int_rn(a * (1.f - b) * (1 << W_BITS));
  iw10 = __float2int_rn((1.f - a) * b * (1 << W_BITS));
  iw11 = (1 << W_BITS) - iw00 - iw01 - iw10;
}

__global__ __launch_bounds__(KLT_THREADS) void k_klt(KltLevels L, const float2* __restrict__ px_prev, const float2* __restrict__ px_init,
                                                     hso_klt_result* __restrict__ out, int n, int win, int max_count, double eps2,
                                                     int use_initial)
{
  __shared__ KltShared s;
  const int i = blockIdx.x, tid = threadIdx.x;
  if (i >= n) return;
  const int area = win * win;
  const float half = (win - 1) * 0.5f;
  const float FLT_SCALE = 1.f / (1 << 20);
  const float2 p0 = px_prev[i];
  float2 next = px_init[i];
  int status = 1;
  for (int level = L.last; level >= 0; level--) {
    const int w = L.w[level], h = L.h[level];
    const uint8_t* __restrict__ I = L.prev[level];
    const uint8_t* __restrict__ J = L.cur[level];
    const short2* __restrict__ dI = L.deriv[level];
    float2 prev = make_float2(p0.x * (float)(1. / (1 << level)), p0.y * (float)(1. / (1 << level)));
    float2 nxt;
    if (level == L.last) nxt = use_initial ? make_float2(next.x * (float)(1. / (1 << level)), next.y * (float)(1. / (1 << level))) : prev;
    else nxt = make_float2(next.x * 2.f, next.y * 2.f);
    next = nxt;
    prev.x -= half; prev.y -= half;
    const int ipx = (int)floorf(prev.x), ipy = (int)floorf(prev.y);
    if (ipx < -win || ipx >= w || ipy < -win || ipy >= h) { if (level == 0) status = 0; continue; }   // uniform over the workgroup
    int iw00, iw01, iw10, iw11;
    bilinear_weights(prev.x - ipx, prev.y - ipy, iw00, iw01, iw10, iw11);
    long long a11 = 0, a12 = 0, a22 = 0;
    __syncthreads();                                    // the previous level's readers of s.I / s.dI are done
    for (int k = tid; k < area; k += KLT_THREADS) {
      const int y = k / win, x = k - y * win;
      const int X0 = ipx + x, Y0 = ipy + y, X1 = X0 + 1, Y1 = Y0 + 1;
      // intensity outside the image: BORDER_REFLECT_101; derivative outside: 0 (BORDER_CONSTANT)
      const int rx0 = reflect101(X0, w), rx1 = reflect101(X1, w);
      const uint8_t* row0 = I + (size_t)reflect101(Y0, h) * w;
      const uint8_t* row1 = I + (size_t)reflect101(Y1, h) * w;
      const int ival = DESCALE(row0[rx0] * iw00 + row0[rx1] * iw01 + row1[rx0] * iw10 + row1[rx1] * iw11, 14 - 5);
      const bool x0in = X0 >= 0 && X0 < w, x1in = X1 >= 0 && X1 < w, y0in = Y0 >= 0 && Y0 < h, y1in = Y1 >= 0 && Y1 < h;
      const short2 z = make_short2(0, 0);
      const short2 d00 = x0in && y0in ? dI[(size_t)Y0 * w + X0] : z, d01 = x1in && y0in ? dI[(size_t)Y0 * w + X1] : z;
      const short2 d10 = x0in && y1in ? dI[(size_t)Y1 * w + X0] : z, d11 = x1in && y1in ? dI[(size_t)Y1 * w + X1] : z;
      const int ixval = DESCALE(d00.x * iw00 + d01.x * iw01 + d10.x * iw10 + d11.x * iw11, 14);
      const int iyval = DESCALE(d00.y * iw00 + d01.y * iw01 + d10.y * iw10 + d11.y * iw11, 14);
      s.I[k] = (short)ival; s.dI[k] = make_short2((short)ixval, (short)iyval);
      a11 += (long long)ixval * ixval; a12 += (long long)ixval * iyval; a22 += (long long)iyval * iyval;
    }
    block_sum3(s, a11, a12, a22);
    const float A11 = (float)a11 * FLT_SCALE, A12 = (float)a12 * FLT_SCALE, A22 = (float)a22 * FLT_SCALE;
    float D = A11 * A22 - A12 * A12;
    const float minEig = (A22 + A11 - sqrtf((A11 - A22) * (A11 - A22) + 4.f * A12 * A12)) / (2 * win * win);
    if (minEig < 1e-4f || D < 1.1920929e-07f) { if (level == 0) status = 0; continue; }
    D = 1.f / D;
    nxt.x -= half; nxt.y -= half;
    float pdx = 0, pdy = 0;
    for (int j = 0; j < max_count; j++) {
      const int inx = (int)floorf(nxt.x), iny = (int)floorf(nxt.y);
      if (inx < -win || inx >= w || iny < -win || iny >= h) { if (level == 0) status = 0; break; }
      bilinear_weights(nxt.x - inx, nxt.y - iny, iw00, iw01, iw10, iw11);
      long long b1 = 0, b2 = 0, unused = 0;
      for (int k = tid; k < area; k += KLT_THREADS) {
        const int y = k / win, x = k - y * win;
        const int rx0 = reflect101(inx + x, w), rx1 = reflect101(inx + x + 1, w);
        const uint8_t* row0 = J + (size_t)reflect101(iny + y, h) * w;
        const uint8_t* row1 = J + (size_t)reflect101(iny + y + 1, h) * w;
        const int diff = DESCALE(row0[rx0] * iw00 + row0[rx1] * iw01 + row1[rx0] * iw10 + row1[rx1] * iw11, 14 - 5) - s.I[k];
        const short2 d = s.dI[k];
        b1 += (long long)diff * d.x; b2 += (long long)diff * d.y;
      }
      block_sum3(s, b1, b2, unused);
      const float fb1 = (float)b1 * FLT_SCALE, fb2 = (float)b2 * FLT_SCALE;
      const float dx = (A12 * fb2 - A22 * fb1) * D, dy = (A12 * fb1 - A11 * fb2) * D;
      nxt.x += dx; nxt.y += dy;
      next = make_float2(nxt.x + half, nxt.y + half);
      if ((double)dx * dx + (double)dy * dy <= eps2) break;
      if (j > 0 && fabsf(dx + pdx) < 0.01f && fabsf(dy + pdy) < 0.01f) { next.x -= dx * 0.5f; next.y -= dy * 0.5f; break; }
      pdx = dx; pdy = dy;
    }
  }
  // patchCheck (src/initialization.cpp:476-563) on the level-0 images: 8x8 bilinear patches, zero-mean NCC > 0.8.  The patch
  // values are computed by 64 lanes per image; the sums run serially on one lane in the reference's order.
  __syncthreads();
  const int w = L.w[0], h = L.h[0];
  if (tid < 128) {
    const int which = tid >> 6, k = tid & 63;
    const float u = which ? next.x : p0.x, v = which ? next.y : p0.y;
    const uint8_t* img = which ? L.cur[0] : L.prev[0];
    const int ui = (int)floorf(u), vi = (int)floorf(v);
    const bool in = !(ui < 4 || ui >= w - 4 || vi < 4 || vi >= h - 4);     // NaN positions fail here: floorf(NaN) converts to INT_MIN
    if (k == 0) s.ok[which] = in ? 1 : 0;
    if (in) {
      const float su = u - ui, sv = v - vi;
      const float wtl = (float)((1.0 - su) * (1.0 - sv)), wtr = (float)(su * (1.0 - sv)), wbl = (float)((1.0 - su) * sv), wbr = su * sv;
      const uint8_t* p = img + (size_t)(vi - 4 + (k >> 3)) * w + (ui - 4 + (k & 7));
      s.patch[which][k] = wtl * p[0] + wtr * p[1] + wbl * p[w] + wbr * p[w + 1];
    }
  }
  __syncthreads();
  if (tid == 0) {
    float ncc = -2.f;
    int patch_ok = 0;
    if (s.ok[0] && s.ok[1]) {
      float ma = 0, mb = 0;
      for (int k = 0; k < 64; k++) { ma += s.patch[0][k]; mb += s.patch[1][k]; }
      ma /= 64; mb /= 64;
      float num = 0, d1 = 0, d2 = 0;
      for (int k = 0; k < 64; k++) {
        const float a = s.patch[0][k] - ma, b = s.patch[1][k] - mb;
        num += a * b; d1 += a * a; d2 += b * b;
      }
      const double r = (double)num / ((double)sqrtf(d1 * d2) + 1e-12);
      ncc = (float)r;
      patch_ok = r > (double)0.8f;
    }
    hso_klt_result o;
    o.px[0] = next.x; o.px[1] = next.y; o.ncc = ncc;
    o.status = (status ? HSO_KLT_TRACKED : 0) | (patch_ok ? HSO_KLT_PATCH_OK : 0);
    out[i] = o;
  }
}

}  // namespace

extern "C" int hso_gpu_klt_levels(int width, int height, int win, int max_level)
{
  for (int level = 0; level <= max_level; level++) {
    width = (width + 1) / 2; height = (height + 1) / 2;
    if (width <= win || height <= win) return level;
  }
  return max_level;
}

extern "C" int hso_gpu_klt_track(hso_gpu_ctx* ctx, int64_t frame_prev, int64_t frame_cur, const float* px_prev, const float* px_init, int n,
                                 const hso_klt_params* params, hso_klt_result* out)
{
  if (!ctx || !params || (n > 0 && (!px_prev || !px_init || !out)) || n < 0) return HSO_E_INVALID;
  if (params->win_size < 3 || params->win_size > KLT_MAX_WIN || params->max_level < 0 || params->max_level >= KLT_MAX_LEVELS)
    return hso_fail(ctx, HSO_E_INVALID, "klt_track: window size must be 3..32 and max_level 0..7");
  auto ip = ctx->frames.find(frame_prev), ic = ctx->frames.find(frame_cur);
  if (ip == ctx->frames.end() || ic == ctx->frames.end()) return hso_fail(ctx, HSO_E_NOFRAME, "klt_track: frame not resident");
  if (!same_geom(ip->second.g, ic->second.g)) return hso_fail(ctx, HSO_E_INVALID, "klt_track: the two frames differ in size");
  if (n == 0) return HSO_OK;
  const PyrGeom& g = ip->second.g;
  const int win = params->win_size;
  KltLevels L;
  L.last = hso_gpu_klt_levels(g.w[0], g.h[0], win, params->max_level);
  // one allocation: [levels 1.. of prev | levels 1.. of cur | derivatives of every prev level | px_prev | px_init | results]
  size_t off = 0, img_off[2][KLT_MAX_LEVELS], der_off[KLT_MAX_LEVELS];
  L.w[0] = g.w[0]; L.h[0] = g.h[0];
  for (int l = 1; l <= L.last; l++) { L.w[l] = (L.w[l - 1] + 1) / 2; L.h[l] = (L.h[l - 1] + 1) / 2; }
  auto take = [&](size_t bytes) { const size_t o = off; off += (bytes + 255) & ~(size_t)255; return o; };
  for (int f = 0; f < 2; f++) for (int l = 1; l <= L.last; l++) img_off[f][l] = take((size_t)L.w[l] * L.h[l]);
  for (int l = 0; l <= L.last; l++) der_off[l] = take(sizeof(short2) * (size_t)L.w[l] * L.h[l]);
  const size_t o_prev = take(sizeof(float2) * (size_t)n), o_init = take(sizeof(float2) * (size_t)n), o_out = take(sizeof(hso_klt_result) * (size_t)n);
  // the context's grow-only work area (no allocation per call: hipFree synchronises the device)
  if (ctx->batch_cap < off) {
    HSO_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->d_batch) (void)hipFree(ctx->d_batch);
    ctx->d_batch = nullptr; ctx->batch_cap = 0;
    HSO_HIP_CHECK(ctx, hipMalloc(reinterpret_cast<void**>(&ctx->d_batch), hso_grown(off)));
    ctx->batch_cap = hso_grown(off);
  }
  char* d = ctx->d_batch;
  int rc = HSO_OK;
  auto body = [&]() -> int {
    L.prev[0] = ip->second.base + g.off[0]; L.cur[0] = ic->second.base + g.off[0];
    for (int l = 1; l <= L.last; l++) { L.prev[l] = (const uint8_t*)(d + img_off[0][l]); L.cur[l] = (const uint8_t*)(d + img_off[1][l]); }
    for (int l = 0; l <= L.last; l++) L.deriv[l] = (const short2*)(d + der_off[l]);
    HSO_HIP_CHECK(ctx, hipMemcpyAsync(d + o_prev, px_prev, sizeof(float2) * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
    HSO_HIP_CHECK(ctx, hipMemcpyAsync(d + o_init, px_init, sizeof(float2) * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
    const dim3 tb(32, 8);
    for (int l = 1; l <= L.last; l++) {
      const dim3 gr((L.w[l] + 31) / 32, (L.h[l] + 7) / 8);
      k_pyr_down<<<gr, tb, 0, ctx->stream>>>(L.prev[l - 1], L.w[l - 1], L.h[l - 1], (uint8_t*)(d + img_off[0][l]), L.w[l], L.h[l]);
      k_pyr_down<<<gr, tb, 0, ctx->stream>>>(L.cur[l - 1], L.w[l - 1], L.h[l - 1], (uint8_t*)(d + img_off[1][l]), L.w[l], L.h[l]);
    }
    for (int l = 0; l <= L.last; l++) {
      const dim3 gr((L.w[l] + 31) / 32, (L.h[l] + 7) / 8);
      k_scharr<<<gr, tb, 0, ctx->stream>>>(L.prev[l], L.w[l], L.h[l], (short2*)(d + der_off[l]));
    }
    int max_count = params->max_iter;                    // TermCriteria clamping of calcOpticalFlowPyrLK
    if (max_count < 0) max_count = 0;
    if (max_count > 100) max_count = 100;
    double eps = params->epsilon;
    if (eps < 0) eps = 0;
    if (eps > 10) eps = 10;
    k_klt<<<n, KLT_THREADS, 0, ctx->stream>>>(L, (const float2*)(d + o_prev), (const float2*)(d + o_init), (hso_klt_result*)(d + o_out), n, win,
                                              max_count, eps * eps, params->use_initial_flow ? 1 : 0);
    HSO_HIP_CHECK(ctx, hipGetLastError());
    HSO_HIP_CHECK(ctx, hipMemcpyAsync(out, d + o_out, sizeof(hso_klt_result) * (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
    HSO_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    return HSO_OK;
  };
  rc = body();
  if (rc != HSO_OK) (void)hipStreamSynchronize(ctx->stream);
  return rc;
}

// the Gaussian pyramid level / Scharr image the tracker would use, for the parity tests of the two kernels
extern "C" int hso_gpu_klt_debug_level(hso_gpu_ctx* ctx, int64_t frame, int level, uint8_t* img_out, int16_t* deriv_out)
{
  if (!ctx || level < 0 || level >= KLT_MAX_LEVELS) return HSO_E_INVALID;
  auto it = ctx->frames.find(frame);
  if (it == ctx->frames.end()) return hso_fail(ctx, HSO_E_NOFRAME, "klt_debug_level: frame not resident");
  const PyrGeom& g = it->second.g;
  int w[KLT_MAX_LEVELS], h[KLT_MAX_LEVELS];
  w[0] = g.w[0]; h[0] = g.h[0];
  for (int l = 1; l <= level; l++) { w[l] = (w[l - 1] + 1) / 2; h[l] = (h[l - 1] + 1) / 2; }
  uint8_t* d = nullptr;
  const size_t lvl_bytes = ((size_t)g.w[0] * g.h[0] + 255) & ~(size_t)255;
  if (hipMalloc(&d, 2 * lvl_bytes + sizeof(short2) * (size_t)w[level] * h[level]) != hipSuccess) return hso_fail(ctx, HSO_E_NOMEM, "klt_debug_level");
  const uint8_t* src = it->second.base + g.off[0];
  const dim3 tb(32, 8);
  for (int l = 1; l <= level; l++) {
    uint8_t* dst = d + (l & 1) * lvl_bytes;
    k_pyr_down<<<dim3((w[l] + 31) / 32, (h[l] + 7) / 8), tb, 0, ctx->stream>>>(src, w[l - 1], h[l - 1], dst, w[l], h[l]);
    src = dst;
  }
  short2* dd = (short2*)(d + 2 * lvl_bytes);
  k_scharr<<<dim3((w[level] + 31) / 32, (h[level] + 7) / 8), tb, 0, ctx->stream>>>(src, w[level], h[level], dd);
  hipError_t e = hipSuccess;
  if (img_out) e = hipMemcpyAsync(img_out, src, (size_t)w[level] * h[level], hipMemcpyDeviceToHost, ctx->stream);
  if (e == hipSuccess && deriv_out) e = hipMemcpyAsync(deriv_out, dd, sizeof(short2) * (size_t)w[level] * h[level], hipMemcpyDeviceToHost, ctx->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  (void)hipFree(d);
  if (e != hipSuccess) { ctx->err = std::string("klt_debug_level: ") + hipGetErrorString(e); hso_stream_abandon(ctx->stream); return HSO_E_HIP; }
  return HSO_OK;
}
