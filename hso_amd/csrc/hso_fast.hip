// hso_fast.hip — FAST-9 corner candidates on gfx950: segment test, corner score, 3x3 non-maximum
// suppression, border filter and Shi-Tomasi response, emitted in raster order.
//
// Replaces the body of FeatureExtractor::fastDetect (reference src/feature_detection.cpp:518-587 (fastDetectST per level, fastDetect))
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
#include "hso_fast_plan.h"
#include <vector>

template <int ARC>
__device__ __forceinline__ int fast_strength(const int d[16])
{
  // S = max over starts s of min_{k<ARC} (+-d[s+k]); sliding minimum by doubling.  Both polarities
  // ride in the two 16-bit halves of one register (v_pk_min_i16 / v_pk_max_i16): |d| <= 255.
  static_assert(ARC == 9 || ARC == 12, "FAST-9 (fastDetect) and FAST-12 (fillingHole)");
  typedef short s16x2 __attribute__((ext_vector_type(2)));
  s16x2 a[16], m2[16], m4[16], m8[16];
#pragma unroll
  for (int i = 0; i < 16; i++) { a[i].x = (short)d[i]; a[i].y = (short)-d[i]; }
#pragma unroll
  for (int i = 0; i < 16; i++) m2[i] = __builtin_elementwise_min(a[i], a[(i + 1) & 15]);
#pragma unroll
  for (int i = 0; i < 16; i++) m4[i] = __builtin_elementwise_min(m2[i], m2[(i + 2) & 15]);
#pragma unroll
  for (int i = 0; i < 16; i++) m8[i] = __builtin_elementwise_min(m4[i], m4[(i + 4) & 15]);
  s16x2 best = {0, 0};
#pragma unroll
  for (int i = 0; i < 16; i++)
    best = __builtin_elementwise_max(best, __builtin_elementwise_min(m8[i], ARC == 9 ? a[(i + 8) & 15] : m4[(i + 8) & 15]));
  return max((int)best.x, (int)best.y);
}

// circle offsets in the library's order (fast_9_score.cpp:4661-4678)
__device__ __constant__ int8_t c_circle[16][2] = {
  {0, 3}, {1, 3}, {2, 2}, {3, 1}, {3, 0}, {3, -1}, {2, -2}, {1, -3},
  {0, -3}, {-1, -3}, {-2, -2}, {-3, -1}, {-3, 0}, {-3, 1}, {-2, 2}, {-1, 3}
};

// one level of a batch of frames: blockIdx.z = frame; every frame has its own slice of the work area
struct FastArgs {
  const uint8_t* const* bases;  // frame base pointers (device table)
  uint32_t img_off;             // byte offset of the level inside a frame
  int W, H, threshold, border, words_per_row, cap;
  char* work;
  size_t per_frame, o_cnt, o_mask, o_off, o_out;  // slice size and the level's offsets inside a slice
  int* totals;                  // [n_frames][n_levels]
  int n_levels, level;
  const int* thr;               // the threshold of every frame (null: `threshold` for all)
};

template <int ARC>
__global__ __launch_bounds__(256) void k_fast_mask(FastArgs A)
{
  __shared__ uint8_t s_src[FAST_TH + 8][FAST_TW + 8];
  __shared__ short s_sc[FAST_TH + 2][FAST_TW + 2];
  const uint8_t* img = A.bases[blockIdx.z] + A.img_off;
  char* slice = A.work + (size_t)blockIdx.z * A.per_frame;
  unsigned long long* mask = reinterpret_cast<unsigned long long*>(slice + A.o_mask);
  int* row_count = reinterpret_cast<int*>(slice + A.o_cnt);
  const int W = A.W, H = A.H, threshold = A.thr ? A.thr[blockIdx.z] : A.threshold, border = A.border, words_per_row = A.words_per_row;
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
      const int sb = fast_strength<ARC>(d) - 1;
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

__global__ __launch_bounds__(256) void k_fast_scan(FastArgs A)
{
  __shared__ int s_part[256];
  char* slice = A.work + (size_t)blockIdx.x * A.per_frame;
  const int* row_count = reinterpret_cast<const int*>(slice + A.o_cnt);
  int* row_off = reinterpret_cast<int*>(slice + A.o_off);
  int* total_out = A.totals + (size_t)blockIdx.x * A.n_levels + A.level;
  const int H = A.H;
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

template <int ARC>
__global__ __launch_bounds__(256) void k_fast_emit(FastArgs A)
{
  const uint8_t* img = A.bases[blockIdx.z] + A.img_off;
  char* slice = A.work + (size_t)blockIdx.z * A.per_frame;
  const unsigned long long* mask = reinterpret_cast<const unsigned long long*>(slice + A.o_mask);
  const int* row_off = reinterpret_cast<const int*>(slice + A.o_off);
  hso_corner* out = reinterpret_cast<hso_corner*>(slice + A.o_out);
  const int W = A.W, H = A.H, words_per_row = A.words_per_row, cap = A.cap;
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
  o.score = fast_strength<ARC>(d) - 1;
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

size_t hso_fast_plan(const PyrGeom& g, int n_frames, int n_levels, int cap, FastPlan* plan)
{
  FastPlan& P = *plan;
  P.g = g; P.n_frames = n_frames; P.n_levels = n_levels; P.cap = cap;
  auto al = [](size_t b) { return (b + 255) & ~size_t(255); };
  // slice of one frame: [row counts of all levels | per level: mask, row offsets, corner list]
  size_t o = 0;
  for (int l = 0; l < n_levels; l++) {
    P.wpr[l] = (g.w[l] + FAST_TW - 1) / FAST_TW;
    P.o_cnt[l] = o; o += al(sizeof(int) * (size_t)g.h[l]);
  }
  P.cnt_bytes = o;
  for (int l = 0; l < n_levels; l++) {
    P.o_mask[l] = o; o += al(sizeof(unsigned long long) * (size_t)g.h[l] * P.wpr[l]);
    P.o_off[l] = o; o += al(sizeof(int) * (size_t)g.h[l]);
    P.o_out[l] = o; o += al(sizeof(hso_corner) * (size_t)cap);
  }
  P.per_frame = o;
  P.o_tab = P.per_frame * (size_t)n_frames;
  P.o_tot = P.o_tab + al(sizeof(void*) * (size_t)n_frames);
  P.o_thr = P.o_tot + al(sizeof(int) * (size_t)n_frames * n_levels);
  P.o_extra = P.o_thr + al(sizeof(int) * 3 * (size_t)n_frames);
  return P.o_extra;
}

int hso_fast_launch(hso_gpu_ctx* ctx, const FastPlan& P, const uint8_t* const* d_bases, int threshold, int border, int arc, const int* d_thr)
{
  const PyrGeom& g = P.g;
  const int n_frames = P.n_frames, n_levels = P.n_levels, cap = P.cap;
  HSO_HIP_CHECK(ctx, hipMemset2DAsync(P.d, P.per_frame, 0, P.cnt_bytes, (size_t)n_frames, ctx->stream));  // the row counters of every slice
  FastArgs A;
  A.bases = d_bases;
  A.threshold = threshold; A.border = border; A.cap = cap; A.thr = d_thr;
  A.work = P.d; A.per_frame = P.per_frame;
  A.totals = reinterpret_cast<int*>(P.d + P.o_tot);
  A.n_levels = n_levels;
  for (int l = 0; l < n_levels; l++) {
    A.img_off = g.off[l]; A.W = g.w[l]; A.H = g.h[l]; A.words_per_row = P.wpr[l]; A.level = l;
    A.o_cnt = P.o_cnt[l]; A.o_mask = P.o_mask[l]; A.o_off = P.o_off[l]; A.o_out = P.o_out[l];
    const dim3 gm(P.wpr[l], (A.H + FAST_TH - 1) / FAST_TH, n_frames), ge((A.H * P.wpr[l] + 3) / 4, 1, n_frames);
    if (arc == 12) hipLaunchKernelGGL(k_fast_mask<12>, gm, dim3(256), 0, ctx->stream, A);
    else hipLaunchKernelGGL(k_fast_mask<9>, gm, dim3(256), 0, ctx->stream, A);
    hipLaunchKernelGGL(k_fast_scan, dim3(n_frames), dim3(256), 0, ctx->stream, A);
    if (cap > 0) {
      if (arc == 12) hipLaunchKernelGGL(k_fast_emit<12>, ge, dim3(256), 0, ctx->stream, A);
      else hipLaunchKernelGGL(k_fast_emit<9>, ge, dim3(256), 0, ctx->stream, A);
    }
    HSO_HIP_CHECK(ctx, hipGetLastError());
  }
  return HSO_OK;
}

int hso_fast_enqueue(hso_gpu_ctx* ctx, const int64_t* frame_ids, int n_frames, int n_levels, int threshold, int border, int cap,
                     size_t extra, FastPlan* plan, const int32_t* per_frame3)
{
  FastPlan& P = *plan;
  HSO_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  std::vector<const uint8_t*> h_bases(n_frames);
  PyrGeom g{};
  for (int i = 0; i < n_frames; i++) {
    auto it = ctx->frames.find(frame_ids[i]);
    if (it == ctx->frames.end()) return hso_fail(ctx, HSO_E_NOFRAME, "fast_detect: frame not resident");
    if (i == 0) g = it->second.g;
    else if (!same_geom(it->second.g, g)) return hso_fail(ctx, HSO_E_INVALID, "fast_detect: frames of one batch must share one size");
    h_bases[i] = it->second.base;
  }
  const size_t need = hso_fast_plan(g, n_frames, n_levels, cap, &P) + ((extra + 255) & ~size_t(255));
  if (ctx->batch_cap < need) {
    HSO_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->d_batch) (void)hipFree(ctx->d_batch);
    ctx->d_batch = nullptr; ctx->batch_cap = 0;
    HSO_HIP_CHECK(ctx, hipMalloc(reinterpret_cast<void**>(&ctx->d_batch), hso_grown(need)));
    ctx->batch_cap = hso_grown(need);
  }
  char* d = reinterpret_cast<char*>(ctx->d_batch);
  P.d = d;
  HSO_HIP_CHECK(ctx, hipMemcpyAsync(d + P.o_tab, h_bases.data(), sizeof(void*) * (size_t)n_frames, hipMemcpyHostToDevice, ctx->stream));
  const int* d_thr = nullptr;
  if (per_frame3) {
    HSO_HIP_CHECK(ctx, hipMemcpyAsync(d + P.o_thr, per_frame3, sizeof(int32_t) * 3 * (size_t)n_frames, hipMemcpyHostToDevice, ctx->stream));
    d_thr = reinterpret_cast<const int*>(d + P.o_thr);
  }
  return hso_fast_launch(ctx, P, reinterpret_cast<const uint8_t* const*>(d + P.o_tab), threshold, border, 9, d_thr);
}

int hso_fast_collect(hso_gpu_ctx* ctx, const FastPlan& P, hso_corner* out, int32_t* counts, std::vector<HsoListCopy>* more)
{
  const int n_frames = P.n_frames, n_levels = P.n_levels, cap = P.cap;
  HSO_HIP_CHECK(ctx, hipMemcpyAsync(counts, P.d + P.o_tot, sizeof(int) * (size_t)n_frames * n_levels, hipMemcpyDeviceToHost, ctx->stream));
  HSO_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  // the corner lists of every frame and level (and whatever lists the caller reads back with them) in one DMA
  std::vector<HsoListCopy> own;
  std::vector<HsoListCopy>& L = more ? *more : own;
  for (int i = 0; i < n_frames && cap > 0; i++)
    for (int l = 0; l < n_levels; l++) {
      const int c = counts[(size_t)i * n_levels + l];
      const int n = c < cap ? c : cap;
      if (n > 0) L.push_back({out + ((size_t)i * n_levels + l) * cap, P.d + (size_t)i * P.per_frame + P.o_out[l], sizeof(hso_corner) * (size_t)n});
    }
  if (more) return HSO_OK;   // the caller adds its lists and runs hso_lists_to_host
  return hso_lists_to_host(ctx, L);
}

extern "C" int hso_gpu_fast_detect_batch(hso_gpu_ctx* ctx, const int64_t* frame_ids, int n_frames, int n_levels, int threshold,
                                         int border, hso_corner* out, int cap, int32_t* counts)
{
  if (!ctx) return HSO_E_INVALID;
  if (!frame_ids || n_frames < 0 || n_levels < 1 || n_levels > HSO_N_PYR_LEVELS || threshold < 0 || threshold > 255 || border < 0 ||
      cap < 0 || !counts || (cap > 0 && !out))
    return hso_fail(ctx, HSO_E_INVALID, "fast_detect: bad argument");
  if (n_frames == 0) return HSO_OK;
  FastPlan P;
  const int rc = hso_fast_enqueue(ctx, frame_ids, n_frames, n_levels, threshold, border, cap, 0, &P);
  if (rc != HSO_OK) return rc;
  return hso_fast_collect(ctx, P, out, counts);
}

extern "C" int hso_gpu_fast_detect(hso_gpu_ctx* ctx, int64_t frame_id, int n_levels, int threshold, int border, hso_corner* out,
                                   int cap, int32_t* counts)
{
  return hso_gpu_fast_detect_batch(ctx, &frame_id, 1, n_levels, threshold, border, out, cap, counts);
}
