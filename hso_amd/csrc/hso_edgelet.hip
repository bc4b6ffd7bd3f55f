// hso_edgelet.hip — corner + edgelet candidates of a new keyframe on gfx950, chained on the device
// behind the FAST stage: the non-init branch of FeatureExtractor::detect up to the oct-tree
// (reference src/feature_detection.cpp:408-447: fastDetectMT :498-545, edgeLetDetectMT :731-830).
//
// edgeLetDetectST = cv::Canny(sobelX, sobelY, edges, 31*minThresh, 70*minThresh, L2gradient) on the
// Sobel-5 images the frame already holds in HBM, then one arg-max per free grid index.  cv::Canny
// (OpenCV imgproc/src/canny.cpp, custom-gradient overload; an absent dependency, restated in
// oracle/hso_oracle_edgelet.c) is integer arithmetic:
//   m = gx^2 + gy^2; survivor of non-maximum suppression in one of three sectors picked by the
//   fixed-point tangent test; edge <=> survivor with m > low^2 connected (8-neighbourhood, through
//   survivors) to a survivor with m > high^2.
// The connectivity closure is the only sequential part (a stack flood fill in OpenCV).  It is a
// monotone fixed point — labels only move weak -> edge — so any evaluation order ends in the same
// map; here every 64x32 tile closes itself in LDS and whole-image passes repeat until no tile
// changed a pixel (a flag per pass lets the remaining queued passes return at once).
//
// MI355X mapping (HBM/L2-bound byte and short work; all levels of all frames in one launch:
// blockIdx.y = level, blockIdx.z = frame, so a single keyframe still fills ~200 workgroups):
//   k_cell_mark      FAST mask words -> haveFeatures_ flags (getCellIndex)
//   k_canny_nms      64x16 tile + 1 ring of packed (gx, gy) in LDS -> label bytes 0 weak / 1 none / 2 edge
//   k_canny_close    tile closure in LDS; repeated
//   k_edgelet_cells  one lane per grid index: arg-max of sqrtf(m) over its window
//   k_edgelet_pack   ordered compaction (grid-index order = the reference's push order)
#include "hso_fast_plan.h"
#include <vector>

#define EDGE_LEVELS HSO_N_SOBEL_LEVELS
#define EDGE_MAX_PASSES 64
#define NMS_TW 64
#define NMS_TH 16
#define CLOSE_TW 64
#define CLOSE_TH 32

struct EdgeLevel {
  int W, H;                 // level image
  int grid, gcols, grows;   // occupancy grid of the level (FeatureExtractor ctor :393-400)
  uint32_t gx_off, gy_off;  // Sobel images inside a frame
  size_t o_map, o_res, o_out, o_have;  // offsets inside a frame's edgelet slice
  size_t o_fmask;           // FAST mask inside a frame's FAST slice
  int wpr;
};

struct EdgeArgs {
  EdgeLevel lv[EDGE_LEVELS];
  const uint8_t* const* bases;
  char* fast_work; size_t fast_per_frame;
  char* work; size_t per_frame;      // edgelet slices
  int* flags;                        // [EDGE_MAX_PASSES + 1]; flags[p + 1] != 0: pass p changed a pixel
  int* totals;                       // [n_frames][n_levels]
  int n_levels, low, high, cap, pass;
};

// static selection of the level record: a dynamic index into the by-value argument would force a
// scratch copy of the whole block
__device__ __forceinline__ const EdgeLevel& level_of(const EdgeArgs& A, int l) { return l == 0 ? A.lv[0] : (l == 1 ? A.lv[1] : A.lv[2]); }

__global__ __launch_bounds__(256) void k_cell_mark(EdgeArgs A)
{
  const EdgeLevel& L = level_of(A, blockIdx.y);
  const int widx = blockIdx.x * 256 + threadIdx.x;
  if (widx >= L.H * L.wpr) return;
  const unsigned long long* mask = reinterpret_cast<const unsigned long long*>(A.fast_work + (size_t)blockIdx.z * A.fast_per_frame + L.o_fmask);
  unsigned long long m = mask[widx];
  if (!m) return;
  uint8_t* have = reinterpret_cast<uint8_t*>(A.work + (size_t)blockIdx.z * A.per_frame + L.o_have);
  const int y = widx / L.wpr, x0 = (widx - y * L.wpr) * 64;
  while (m) {
    const int b = __builtin_ctzll(m);
    m &= m - 1;
    have[y / L.grid * L.gcols + (x0 + b) / L.grows] = 1;   // getCellIndex, feature_detection.h:295-299
  }
}

__device__ __forceinline__ int mag_of(uint32_t g)
{
  const int gx = (int)(short)(g & 0xffffu), gy = (int)(short)(g >> 16);
  return gx * gx + gy * gy;
}

__global__ __launch_bounds__(256) void k_canny_nms(EdgeArgs A)
{
  __shared__ uint32_t s_g[NMS_TH + 2][NMS_TW + 2];   // packed (gx, gy); zero outside the image = zero magnitude
  const EdgeLevel& L = level_of(A, blockIdx.y);
  const int W = L.W, H = L.H;
  const int tiles_x = (W + NMS_TW - 1) / NMS_TW, tiles_y = (H + NMS_TH - 1) / NMS_TH;
  if ((int)blockIdx.x >= tiles_x * tiles_y) return;
  const int ty = blockIdx.x / tiles_x, tx = blockIdx.x - ty * tiles_x;
  const int x0 = tx * NMS_TW, y0 = ty * NMS_TH;
  const uint8_t* base = A.bases[blockIdx.z];
  const int16_t* gxp = reinterpret_cast<const int16_t*>(base + L.gx_off);
  const int16_t* gyp = reinterpret_cast<const int16_t*>(base + L.gy_off);
  const int t = threadIdx.x;
  for (int i = t; i < (NMS_TH + 2) * (NMS_TW + 2); i += 256) {
    const int ly = i / (NMS_TW + 2), lx = i - ly * (NMS_TW + 2);
    const int y = y0 + ly - 1, x = x0 + lx - 1;
    uint32_t v = 0;
    if (x >= 0 && x < W && y >= 0 && y < H) {
      const size_t o = (size_t)y * W + x;
      v = (uint32_t)(uint16_t)gxp[o] | ((uint32_t)(uint16_t)gyp[o] << 16);
    }
    s_g[ly][lx] = v;
  }
  __syncthreads();
  const int ly = t >> 4, lx0 = (t & 15) * 4;     // 4 consecutive pixels of one row
  const int y = y0 + ly;
  if (y >= H) return;
  uint8_t* map = reinterpret_cast<uint8_t*>(A.work + (size_t)blockIdx.z * A.per_frame + L.o_map);
  const int low = A.low, high = A.high;
  const int TG22 = 13573;                        // (int)(0.41421356... * 2^15 + 0.5)
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const int lx = lx0 + k, x = x0 + lx;
    if (x >= W) break;
    const uint32_t g = s_g[ly + 1][lx + 1];
    const int m = mag_of(g);
    uint8_t label = 1;
    if (m > low) {
      const int xs = (int)(short)(g & 0xffffu), ys = (int)(short)(g >> 16);
      const int ax = abs(xs), ay = abs(ys) << 15;
      const int tg22x = ax * TG22;
      bool keep;
      if (ay < tg22x) {
        keep = m > mag_of(s_g[ly + 1][lx]) && m >= mag_of(s_g[ly + 1][lx + 2]);
      } else {
        const int tg67x = tg22x + (ax << 16);
        if (ay > tg67x) {
          keep = m > mag_of(s_g[ly][lx + 1]) && m >= mag_of(s_g[ly + 2][lx + 1]);
        } else {
          const int s = (xs ^ ys) < 0 ? -1 : 1;
          keep = m > mag_of(s_g[ly][lx + 1 - s]) && m > mag_of(s_g[ly + 2][lx + 1 + s]);
        }
      }
      if (keep) label = m > high ? 2 : 0;
    }
    map[(size_t)y * W + x] = label;
  }
}

__global__ __launch_bounds__(256) void k_canny_close(EdgeArgs A)
{
  __shared__ uint8_t s_m[CLOSE_TH + 2][CLOSE_TW + 4];
  if (A.pass > 0 && A.flags[A.pass] == 0) return;      // the previous pass changed nothing: closed
  const EdgeLevel& L = level_of(A, blockIdx.y);
  const int W = L.W, H = L.H;
  const int tiles_x = (W + CLOSE_TW - 1) / CLOSE_TW, tiles_y = (H + CLOSE_TH - 1) / CLOSE_TH;
  if ((int)blockIdx.x >= tiles_x * tiles_y) return;
  const int ty = blockIdx.x / tiles_x, tx = blockIdx.x - ty * tiles_x;
  const int x0 = tx * CLOSE_TW, y0 = ty * CLOSE_TH;
  uint8_t* map = reinterpret_cast<uint8_t*>(A.work + (size_t)blockIdx.z * A.per_frame + L.o_map);
  const int t = threadIdx.x;
  int has_weak = 0;
  for (int i = t; i < (CLOSE_TH + 2) * (CLOSE_TW + 2); i += 256) {
    const int ly = i / (CLOSE_TW + 2), lx = i - ly * (CLOSE_TW + 2);
    const int y = y0 + ly - 1, x = x0 + lx - 1;
    const uint8_t v = (x >= 0 && x < W && y >= 0 && y < H) ? map[(size_t)y * W + x] : (uint8_t)1;
    s_m[ly][lx] = v;
    has_weak |= (v == 0);
  }
  if (!__syncthreads_or(has_weak)) return;
  const int ly = (t >> 3) + 1, lx0 = (t & 7) * 8 + 1;    // 8 consecutive interior pixels of one row
  unsigned grown = 0;                                    // bit k: pixel k turned into an edge here
  for (;;) {
    int changed = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) {
      const int lx = lx0 + k;
      if (s_m[ly][lx] != 0) continue;
      const bool any = s_m[ly - 1][lx - 1] == 2 || s_m[ly - 1][lx] == 2 || s_m[ly - 1][lx + 1] == 2 || s_m[ly][lx - 1] == 2 ||
                       s_m[ly][lx + 1] == 2 || s_m[ly + 1][lx - 1] == 2 || s_m[ly + 1][lx] == 2 || s_m[ly + 1][lx + 1] == 2;
      if (any) { s_m[ly][lx] = 2; grown |= 1u << k; changed = 1; }
    }
#pragma unroll
    for (int k = 6; k >= 0; k--) {                       // and back, so a run closes in one sweep pair
      const int lx = lx0 + k;
      if (s_m[ly][lx] == 0 && s_m[ly][lx + 1] == 2) { s_m[ly][lx] = 2; grown |= 1u << k; changed = 1; }
    }
    if (!__syncthreads_or(changed)) break;
  }
  const int y = y0 + ly - 1;
  if (grown) {
#pragma unroll
    for (int k = 0; k < 8; k++)
      if ((grown >> k) & 1u) map[(size_t)y * W + (x0 + lx0 - 1 + k)] = 2;   // interior pixels were weak, hence inside the image
    A.flags[A.pass + 1] = 1;
  }
}

__global__ __launch_bounds__(256) void k_edgelet_cells(EdgeArgs A)
{
  const EdgeLevel& L = level_of(A, blockIdx.y);
  const int index = blockIdx.x * 256 + threadIdx.x;
  if (index >= L.gcols * L.grows) return;
  char* slice = A.work + (size_t)blockIdx.z * A.per_frame;
  hso_edgelet* res = reinterpret_cast<hso_edgelet*>(slice + L.o_res);
  hso_edgelet r;
  r.x = -1; r.y = -1; r.gx = 0; r.gy = 0; r.grad = 0.0f;
  const uint8_t* have = reinterpret_cast<const uint8_t*>(slice + L.o_have);
  const int W = L.W, H = L.H, g = L.grid;
  const int border = 8, maxBorderX = W - border, maxBorderY = H - border;
  int iniX = index % L.gcols * g;
  int iniY = index / L.grows * g;                        // sic, feature_detection.cpp:770
  if (!have[index] && !(iniX > maxBorderX || iniY > maxBorderY)) {
    const int maxX = min(iniX + g, maxBorderX), maxY = min(iniY + g, maxBorderY);
    iniX = max(iniX, border); iniY = max(iniY, border);
    const uint8_t* map = reinterpret_cast<const uint8_t*>(slice + L.o_map);
    const uint8_t* base = A.bases[blockIdx.z];
    const int16_t* gxp = reinterpret_cast<const int16_t*>(base + L.gx_off);
    const int16_t* gyp = reinterpret_cast<const int16_t*>(base + L.gy_off);
    float maxGrad = 0.0f;
    for (int y = iniY; y < maxY; ++y)
      for (int x = iniX; x < maxX; ++x) {
        const size_t o = (size_t)y * W + x;
        if (map[o] != 2) continue;
        const int sx = gxp[o], sy = gyp[o];
        const float grad = sqrtf((float)(sx * sx + sy * sy));
        if (grad > maxGrad) { r.x = (int16_t)x; r.y = (int16_t)y; r.gx = (int16_t)sx; r.gy = (int16_t)sy; r.grad = grad; maxGrad = grad; }
      }
  }
  res[index] = r;
}

__global__ __launch_bounds__(256) void k_edgelet_pack(EdgeArgs A)
{
  __shared__ int s_w[4];
  const EdgeLevel& L = level_of(A, blockIdx.y);
  char* slice = A.work + (size_t)blockIdx.z * A.per_frame;
  const hso_edgelet* res = reinterpret_cast<const hso_edgelet*>(slice + L.o_res);
  hso_edgelet* out = reinterpret_cast<hso_edgelet*>(slice + L.o_out);
  const int cells = L.gcols * L.grows, cap = A.cap;
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  int base = 0;
  for (int c0 = 0; c0 < cells; c0 += 256) {
    const int i = c0 + t;
    hso_edgelet r;
    r.x = -1;
    if (i < cells) r = res[i];
    const bool set = r.x >= 0;
    const unsigned long long m = __ballot(set);
    if (lane == 0) s_w[wv] = __popcll(m);
    __syncthreads();
    int before = base;
    for (int k = 0; k < wv; k++) before += s_w[k];
    const int idx = before + __popcll(m & ((1ull << lane) - 1ull));
    if (set && idx < cap) out[idx] = r;
    base += s_w[0] + s_w[1] + s_w[2] + s_w[3];
    __syncthreads();
  }
  if (t == 0) A.totals[(size_t)blockIdx.z * A.n_levels + blockIdx.y] = base;
}

extern "C" int hso_gpu_detect_candidates(hso_gpu_ctx* ctx, const int64_t* frame_ids, int n_frames, int n_levels, int min_thresh,
                                         hso_corner* corners, int corner_cap, int32_t* corner_counts,
                                         hso_edgelet* edgelets, int edgelet_cap, int32_t* edgelet_counts)
{
  if (!ctx) return HSO_E_INVALID;
  if (!frame_ids || n_frames < 0 || n_levels < 1 || n_levels > EDGE_LEVELS || min_thresh < 0 || min_thresh > 255 || corner_cap < 0 ||
      edgelet_cap < 0 || !corner_counts || !edgelet_counts || (corner_cap > 0 && !corners) || (edgelet_cap > 0 && !edgelets))
    return hso_fail(ctx, HSO_E_INVALID, "detect_candidates: bad argument");
  if (n_frames == 0) return HSO_OK;
  auto it0 = ctx->frames.find(frame_ids[0]);
  if (it0 == ctx->frames.end()) return hso_fail(ctx, HSO_E_NOFRAME, "detect_candidates: frame not resident");
  const PyrGeom g = it0->second.g;
  auto al = [](size_t b) { return (b + 255) & ~size_t(255); };
  // extra area: [have flags of every frame | pass flags] (one memset), edgelet totals, then the frame slices
  EdgeArgs A{};
  size_t have_bytes = 0, o = 0;
  int vw = g.w[0], vh = g.h[0], max_cells = 0, max_nms = 0, max_close = 0, max_words = 0;
  for (int l = 0; l < n_levels; l++) {
    EdgeLevel& L = A.lv[l];
    L.W = g.w[l]; L.H = g.h[l];
    L.grid = 8 / (1 << l);                                   // gridSize_ = 8, :395
    L.gcols = (vw + L.grid - 1) / L.grid; L.grows = (vh + L.grid - 1) / L.grid;   // vecWidth_[l] = vecWidth_[l-1] / 2, :362-366
    vw /= 2; vh /= 2;
    L.gx_off = g.sob_off[l][0]; L.gy_off = g.sob_off[l][1];
    const int cells = L.gcols * L.grows;
    // the windows and getCellIndex stay inside the level image / the flag array for the sizes the
    // frame store accepts; .at() would throw in the reference otherwise
    if ((L.H - 1) / L.grid * L.gcols + (L.W - 1) / L.grows >= cells)
      return hso_fail(ctx, HSO_E_INVALID, "detect_candidates: grid index out of range for this image size");
    L.o_have = have_bytes; have_bytes += al((size_t)cells);
    L.o_map = o; o += al((size_t)L.W * L.H);
    L.o_res = o; o += al(sizeof(hso_edgelet) * (size_t)cells);
    L.o_out = o; o += al(sizeof(hso_edgelet) * (size_t)edgelet_cap);
    max_cells = cells > max_cells ? cells : max_cells;
    const int nms = ((L.W + NMS_TW - 1) / NMS_TW) * ((L.H + NMS_TH - 1) / NMS_TH);
    const int cl = ((L.W + CLOSE_TW - 1) / CLOSE_TW) * ((L.H + CLOSE_TH - 1) / CLOSE_TH);
    max_nms = nms > max_nms ? nms : max_nms; max_close = cl > max_close ? cl : max_close;
  }
  // have flags live per frame in front of the slices: offsets above are relative to a frame's have block
  const size_t have_per_frame = have_bytes;
  const size_t slice = al(o + have_per_frame);
  for (int l = 0; l < n_levels; l++) A.lv[l].o_have += o;    // have block sits at the end of each slice
  const size_t o_flags = 0, o_totals = al(sizeof(int) * (EDGE_MAX_PASSES + 1));
  const size_t o_slices = o_totals + al(sizeof(int) * (size_t)n_frames * n_levels);
  const size_t extra = o_slices + slice * (size_t)n_frames;

  FastPlan P;
  const int rc = hso_fast_enqueue(ctx, frame_ids, n_frames, n_levels, min_thresh, 8, corner_cap, extra, &P);   // fastThresh = floor(minThresh_), border 8 (:520-522)
  if (rc != HSO_OK) return rc;
  char* x = P.d + P.o_extra;
  for (int l = 0; l < n_levels; l++) {
    A.lv[l].o_fmask = P.o_mask[l]; A.lv[l].wpr = P.wpr[l];
    const int words = A.lv[l].H * P.wpr[l];
    max_words = words > max_words ? words : max_words;
  }
  A.bases = reinterpret_cast<const uint8_t* const*>(P.d + P.o_tab);
  A.fast_work = P.d; A.fast_per_frame = P.per_frame;
  A.work = x + o_slices; A.per_frame = slice;
  A.flags = reinterpret_cast<int*>(x + o_flags);
  A.totals = reinterpret_cast<int*>(x + o_totals);
  A.n_levels = n_levels; A.cap = edgelet_cap;
  {
    // cv::Canny threshold preparation (L2gradient): clamp to 32767, square, floor
    double lo = 31.0 * min_thresh, hi = 70.0 * min_thresh;
    lo = lo < 32767.0 ? lo : 32767.0; hi = hi < 32767.0 ? hi : 32767.0;
    A.low = (int)(lo * lo); A.high = (int)(hi * hi);
  }
  HSO_HIP_CHECK(ctx, hipMemsetAsync(x + o_flags, 0, sizeof(int) * (EDGE_MAX_PASSES + 1), ctx->stream));
  HSO_HIP_CHECK(ctx, hipMemset2DAsync(A.work + o, slice, 0, have_per_frame, (size_t)n_frames, ctx->stream));
  const dim3 blk(256);
  hipLaunchKernelGGL(k_cell_mark, dim3((max_words + 255) / 256, n_levels, n_frames), blk, 0, ctx->stream, A);
  hipLaunchKernelGGL(k_canny_nms, dim3(max_nms, n_levels, n_frames), blk, 0, ctx->stream, A);
  int pass = 0, flag = 1;
  int32_t* h_flags = edgelet_counts;   // scratch until the totals arrive (n_frames * n_levels >= 1 ints)
  while (flag) {
    if (pass + 4 > EDGE_MAX_PASSES) return hso_fail(ctx, HSO_E_INVALID, "detect_candidates: edge closure did not converge");
    for (int k = 0; k < 4; k++, pass++) {
      A.pass = pass;
      hipLaunchKernelGGL(k_canny_close, dim3(max_close, n_levels, n_frames), blk, 0, ctx->stream, A);
    }
    HSO_HIP_CHECK(ctx, hipGetLastError());
    HSO_HIP_CHECK(ctx, hipMemcpyAsync(h_flags, A.flags + pass, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    if (pass == 4) {
      // optimistic: queue the rest behind the flag read; redone below in the rare case the closure needed more passes
      hipLaunchKernelGGL(k_edgelet_cells, dim3((max_cells + 255) / 256, n_levels, n_frames), blk, 0, ctx->stream, A);
      hipLaunchKernelGGL(k_edgelet_pack, dim3(1, n_levels, n_frames), blk, 0, ctx->stream, A);
    }
    HSO_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    flag = h_flags[0];
  }
  if (pass > 4) {
    hipLaunchKernelGGL(k_edgelet_cells, dim3((max_cells + 255) / 256, n_levels, n_frames), blk, 0, ctx->stream, A);
    hipLaunchKernelGGL(k_edgelet_pack, dim3(1, n_levels, n_frames), blk, 0, ctx->stream, A);
  }
  HSO_HIP_CHECK(ctx, hipGetLastError());
  HSO_HIP_CHECK(ctx, hipMemcpyAsync(edgelet_counts, A.totals, sizeof(int) * (size_t)n_frames * n_levels, hipMemcpyDeviceToHost, ctx->stream));
  const int rc2 = hso_fast_collect(ctx, P, corners, corner_counts);     // synchronises
  if (rc2 != HSO_OK) return rc2;
  for (int i = 0; i < n_frames && edgelet_cap > 0; i++)
    for (int l = 0; l < n_levels; l++) {
      const int c = edgelet_counts[(size_t)i * n_levels + l];
      const int n = c < edgelet_cap ? c : edgelet_cap;
      if (n > 0)
        HSO_HIP_CHECK(ctx, hipMemcpyAsync(edgelets + ((size_t)i * n_levels + l) * edgelet_cap, A.work + (size_t)i * slice + A.lv[l].o_out,
                                          sizeof(hso_edgelet) * (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
    }
  HSO_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  return HSO_OK;
}
