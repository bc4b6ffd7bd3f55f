// hso_fast.hip — FAST-9 corner candidates on gfx950: segment test, corner score, 3x3 non-maximum
// suppression, border filter and Shi-Tomasi response, emitted in raster order.
//
// Replaces the body of FeatureExtractor::fastDetect (reference src/feature_detection.cpp:547-587)
// and what it calls: fast::fast_corner_detect_9_sse2 (thirdparty/fast/src/faster_corner_9_sse.cpp,
// fast_9.cpp), fast::fast_corner_score_9 (fast_9_score.cpp), fast::fast_nonmax_3x3
// (nonmax_3x3.cpp) and hso::shiTomasiScore (src/vikit/vision.cpp:111-151).
//
// The library's detector is a generated decision tree / SSE2 mask cascade for the segment test
// "9 contiguous pixels of the 16-pixel circle all > p + b or all < p - b"; its score function walks
// the barrier up until the test fails.  Both reduce to one quantity per pixel,
//     S(p) = max over the 16 arcs of min over the arc of |I - p|   (common sign along the arc),
// corner at barrier b <=> S - 1 >= b, score = S - 1.  Here every pixel computes S with a
// log-step sliding minimum over the circle (min over 2, 4, 8, then 9 neighbours: 64 integer mins
// per polarity) — branch-free, so a wavefront never diverges on image content.
//
// MI355X mapping (byte work, HBM/L2-bound by design: the image is read once into LDS tiles):
//   k_fast_mask  64x16 tile + 4-pixel halo in LDS -> scores of the tile + 1 ring in LDS -> each
//                wavefront owns one 64-pixel row segment at a time, so "survives non-max and the
//                border test" becomes one ballot = one 64-bit word of a bitmask image; per-row
//                corner counts by integer atomics.
//   k_fast_scan  exclusive prefix of the row counts (one workgroup).
//   k_fast_emit  one wavefront per mask word: lane = bit; rank inside the word by popcount, so the
//                corners land in raster order — the reference's list order, i.e. the candidate
//                index — and each lane recomputes its score and the 8x8 Shi-Tomasi box sums
//                (exact in fp32: every partial sum is an integer < 2^24).
// Everything is integer / exactly-representable arithmetic: results are bit-identical to the
// reference library (tests/golden/fast9.json).
#include "hso_ctx.h"
#include <vector>

#define FAST_TW 64
#define FAST_TH 16

__device__ __forceinline__ int fast9_strength(const int d[16])
{
  // S = max over starts s of min_{k<9} (+-d[s+k]); sliding minimum by doubling
  int best = 0;
#pragma unroll
  for (int pol = 0; pol < 2; pol++) {
    int a[16], m2[16], m4[16], m8[16];
#pragma unroll
    for (int i = 0; i < 16; i++) a[i] = pol ? -d[i] : d[i];
#pragma unroll
    for (int i = 0; i < 16; i++) m2[i] = min(a[i], a[(i + 1) & 15]);
#pragma unroll
    for (int i = 0; i < 16; i++) m4[i] = min(m2[i], m2[(i + 2) & 15]);
#pragma unroll
    for (int i = 0; i < 16; i++) m8[i] = min(m4[i], m4[(i + 4) & 15]);
#pragma unroll
    for (int i = 0; i < 16; i++) best = max(best, min(m8[i], a[(i + 8) & 15]));
  }
  return best;
}

// circle offsets in the library's order (fast_9_score.cpp:4661-4678)
__device__ __constant__ int8_t c_circle[16][2] = {
  {0, 3}, {1, 3}, {2, 2}, {3, 1}, {3, 0}, {3, -1}, {2, -2}, {1, -3},
  {0, -3}, {-1, -3}, {-2, -2}, {-3, -1}, {-3, 0}, {-3, 1}, {-2, 2}, {-1, 3}
};

__global__ __launch_bounds__(256) void k_fast_mask(const uint8_t* img, int W, int H, int threshold, int border,
                                                   unsigned long long* mask, int words_per_row, int* row_count)
{
  __shared__ uint8_t s_src[FAST_TH + 8][FAST_TW + 8];
  __shared__ short s_sc[FAST_TH + 2][FAST_TW + 2];
  const int x0 = blockIdx.x * FAST_TW, y0 = blockIdx.y * FAST_TH;
  const int t = threadIdx.x;
  for (int i = t; i < (FAST_TH + 8) * (FAST_TW + 8); i += 256) {
    const int ly = i / (FAST_TW + 8), lx = i - ly * (FAST_TW + 8);
    const int yy = y0 + ly - 4, xx = x0 + lx - 4;
    s_src[ly][lx] = (yy >= 0 && yy < H && xx >= 0 && xx < W) ? img[(size_t)yy * W + xx] : (uint8_t)0;
  }
  __syncthreads();
  // scores of the tile and one ring around it; -1 = not a corner / outside [3, W-3) x [3, H-3)
  for (int i = t; i < (FAST_TH + 2) * (FAST_TW + 2); i += 256) {
    const int ly = i / (FAST_TW + 2), lx = i - ly * (FAST_TW + 2);
    const int y = y0 + ly - 1, x = x0 + lx - 1;
    short sc = -1;
    if (x >= 3 && x < W - 3 && y >= 3 && y < H - 3) {
      const int cy = ly + 3, cx = lx + 3;  // position in s_src
      const int p = s_src[cy][cx];
      int d[16];
#pragma unroll
      for (int k = 0; k < 16; k++) d[k] = (int)s_src[cy + c_circle[k][1]][cx + c_circle[k][0]] - p;
      const int sb = fast9_strength(d) - 1;
      if (sb >= threshold) sc = (short)sb;
    }
    s_sc[ly][lx] = sc;
  }
  __syncthreads();
  const int lane = t & 63, wv = t >> 6;
#pragma unroll
  for (int k = 0; k < FAST_TH / 4; k++) {
    const int ly = wv + 4 * k;  // one 64-pixel row segment per wavefront and step
    const int y = y0 + ly, x = x0 + lane;
    const int s = s_sc[ly + 1][lane + 1];
    bool keep = s >= 0;
    if (keep) {
      // fast_nonmax_3x3: suppressed by any neighbouring corner with score >= its own
      keep = s_sc[ly][lane] < s && s_sc[ly][lane + 1] < s && s_sc[ly][lane + 2] < s && s_sc[ly + 1][lane] < s &&
             s_sc[ly + 1][lane + 2] < s && s_sc[ly + 2][lane] < s && s_sc[ly + 2][lane + 1] < s && s_sc[ly + 2][lane + 2] < s;
      keep = keep && !(x < border || x > W - border || y < border || y > H - border);  // feature_detection.cpp:573
    }
    const unsigned long long m = __ballot(keep);
    if (lane == 0 && y < H) {
      mask[(size_t)y * words_per_row + blockIdx.x] = m;
      if (m) atomicAdd(&row_count[y], __popcll(m));
    }
  }
}

__global__ __launch_bounds__(256) void k_fast_scan(const int* row_count, int H, int* row_off, int* total_out)
{
  __shared__ int s_part[256];
  const int t = threadIdx.x;
  const int per = (H + 255) / 256;
  int sum = 0;
  for (int i = t * per; i < min(H, (t + 1) * per); i++) sum += row_count[i];
  s_part[t] = sum;
  __syncthreads();
  if (t == 0) {
    int acc = 0;
    for (int i = 0; i < 256; i++) { const int v = s_part[i]; s_part[i] = acc; acc += v; }
    *total_out = acc;
  }
  __syncthreads();
  int acc = s_part[t];
  for (int i = t * per; i < min(H, (t + 1) * per); i++) { row_off[i] = acc; acc += row_count[i]; }
}

__global__ __launch_bounds__(256) void k_fast_emit(const uint8_t* img, int W, int H, const unsigned long long* mask,
                                                   int words_per_row, const int* row_off, hso_corner* out, int cap)
{
  const int lane = threadIdx.x & 63;
  const int widx = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (widx >= H * words_per_row) return;
  const int y = widx / words_per_row, wx = widx - y * words_per_row;
  const unsigned long long m = mask[widx];
  if (m == 0) return;
  int before = row_off[y];
  for (int k = 0; k < wx; k++) before += __popcll(mask[(size_t)y * words_per_row + k]);
  if (!((m >> lane) & 1ull)) return;
  const int idx = before + __popcll(m & ((1ull << lane) - 1ull));
  if (idx >= cap) return;
  const int x = wx * 64 + lane;
  const uint8_t* c = img + (size_t)y * W + x;
  int d[16];
  const int p = c[0];
#pragma unroll
  for (int k = 0; k < 16; k++) d[k] = (int)c[c_circle[k][1] * W + c_circle[k][0]] - p;
  hso_corner o;
  o.x = (int16_t)x; o.y = (int16_t)y;
  o.score = fast9_strength(d) - 1;
  // hso::shiTomasiScore, vision.cpp:111-151 (integer-valued float sums: exact)
  float resp = 0.0f;
  {
    const int x_min = x - 4, x_max = x + 4, y_min = y - 4, y_max = y + 4;
    if (!(x_min < 1 || x_max >= W - 1 || y_min < 1 || y_max >= H - 1)) {
      float dXX = 0, dYY = 0, dXY = 0;
      for (int yy = y_min; yy < y_max; ++yy) {
        const uint8_t* r = img + (size_t)yy * W + x_min;
#pragma unroll
        for (int xx = 0; xx < 8; ++xx) {
          const float dx = (float)((int)r[xx + 1] - (int)r[xx - 1]);
          const float dy = (float)((int)r[xx + W] - (int)r[xx - W]);
          dXX += dx * dx; dYY += dy * dy; dXY += dx * dy;
        }
      }
      dXX = (float)((double)dXX / (2.0 * 64)); dYY = (float)((double)dYY / (2.0 * 64)); dXY = (float)((double)dXY / (2.0 * 64));
      // sqrt(float) is the float overload in the reference; the 0.5 factor is exact
      resp = (float)(0.5 * (double)((dXX + dYY) - sqrtf((dXX + dYY) * (dXX + dYY) - 4 * (dXX * dYY - dXY * dXY))));
    }
  }
  o.response = resp;
  out[idx] = o;
}

extern "C" int hso_gpu_fast_detect(hso_gpu_ctx* ctx, int64_t frame_id, int n_levels, int threshold, int border,
                                   hso_corner* out, int cap, int32_t* counts)
{
  if (!ctx) return HSO_E_INVALID;
  if (n_levels < 1 || n_levels > HSO_N_PYR_LEVELS || threshold < 0 || threshold > 255 || border < 0 || cap < 0 || !counts ||
      (cap > 0 && !out))
    return hso_fail(ctx, HSO_E_INVALID, "fast_detect: bad argument");
  HSO_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  auto it = ctx->frames.find(frame_id);
  if (it == ctx->frames.end()) return hso_fail(ctx, HSO_E_NOFRAME, "fast_detect: frame not resident");
  const PyrGeom& g = it->second.g;
  auto al = [](size_t b) { return (b + 255) & ~size_t(255); };
  // per level: mask, row counts, row offsets (+ total), output list
  size_t o = 0, o_mask[HSO_N_PYR_LEVELS], o_cnt[HSO_N_PYR_LEVELS], o_off[HSO_N_PYR_LEVELS], o_out[HSO_N_PYR_LEVELS];
  int wpr[HSO_N_PYR_LEVELS];
  size_t zero_begin = 0, zero_end = 0;
  for (int l = 0; l < n_levels; l++) {
    wpr[l] = (g.w[l] + FAST_TW - 1) / FAST_TW;
    o_cnt[l] = o; o += al(sizeof(int) * (size_t)g.h[l]);
  }
  zero_end = o;
  for (int l = 0; l < n_levels; l++) {
    o_mask[l] = o; o += al(sizeof(unsigned long long) * (size_t)g.h[l] * wpr[l]);
    o_off[l] = o; o += al(sizeof(int) * ((size_t)g.h[l] + 1));
    o_out[l] = o; o += al(sizeof(hso_corner) * (size_t)(cap > 0 ? cap : 1));
  }
  if (ctx->batch_cap < o) {
    HSO_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->d_batch) (void)hipFree(ctx->d_batch);
    ctx->d_batch = nullptr; ctx->batch_cap = 0;
    HSO_HIP_CHECK(ctx, hipMalloc(reinterpret_cast<void**>(&ctx->d_batch), o));
    ctx->batch_cap = o;
  }
  char* d = reinterpret_cast<char*>(ctx->d_batch);
  HSO_HIP_CHECK(ctx, hipMemsetAsync(d + zero_begin, 0, zero_end - zero_begin, ctx->stream));
  for (int l = 0; l < n_levels; l++) {
    const uint8_t* img = it->second.base + g.off[l];
    const int W = g.w[l], H = g.h[l];
    unsigned long long* mask = reinterpret_cast<unsigned long long*>(d + o_mask[l]);
    int* cnt = reinterpret_cast<int*>(d + o_cnt[l]);
    int* off = reinterpret_cast<int*>(d + o_off[l]);
    hso_corner* dout = reinterpret_cast<hso_corner*>(d + o_out[l]);
    hipLaunchKernelGGL(k_fast_mask, dim3(wpr[l], (H + FAST_TH - 1) / FAST_TH), dim3(256), 0, ctx->stream, img, W, H, threshold,
                       border, mask, wpr[l], cnt);
    hipLaunchKernelGGL(k_fast_scan, dim3(1), dim3(256), 0, ctx->stream, cnt, H, off, off + H);
    hipLaunchKernelGGL(k_fast_emit, dim3((H * wpr[l] + 3) / 4), dim3(256), 0, ctx->stream, img, W, H, mask, wpr[l], off, dout, cap);
    HSO_HIP_CHECK(ctx, hipGetLastError());
  }
  std::vector<int> totals(n_levels, 0);
  for (int l = 0; l < n_levels; l++)
    HSO_HIP_CHECK(ctx, hipMemcpyAsync(&totals[l], d + o_off[l] + sizeof(int) * (size_t)g.h[l], sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
  HSO_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  for (int l = 0; l < n_levels; l++) {
    counts[l] = totals[l];
    const int n = totals[l] < cap ? totals[l] : cap;
    if (n > 0) HSO_HIP_CHECK(ctx, hipMemcpyAsync(out + (size_t)l * cap, d + o_out[l], sizeof(hso_corner) * (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
  }
  HSO_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  return HSO_OK;
}
