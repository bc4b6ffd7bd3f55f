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
// changed a pixel (a flag per pass lets the remaining queued passes return at once, and a tile
// reloads its labels only while it still holds weak pixels and a neighbouring tile just grew).
//
// MI355X mapping (HBM/L2-bound byte and short work; all levels of all frames in one launch:
// blockIdx.y = level, blockIdx.z = frame, so a single keyframe still fills ~200 workgroups):
//   k_cell_mark      FAST mask words -> haveFeatures_ flags (getCellIndex)
//   k_canny_nms      64x32 tile + 2 rings of packed (gx, gy) in LDS -> label bytes 0 weak / 1 none /
//                    2 edge of the tile + 1 ring, closed inside the tile before they reach HBM
//   k_canny_close    closure across tile borders; repeated
//   k_edgelet_cells  eight lanes per grid index (one per window row): arg-max of sqrtf(m) over its window
//   k_edgelet_pack   ordered compaction (grid-index order = the reference's push order)
#include "hso_fast_plan.h"
#include <vector>

#define EDGE_LEVELS HSO_N_SOBEL_LEVELS
// Cross-tile closure passes repeat until one changes nothing.  cv::Canny's flood fill has no iteration limit; here the
// worst case is bounded by the number of tiles a weak contour can wind through (one tile border per pass), so the loop
// below stops on its own and only guards against a broken flag with a bound far above any frame's tile count.
#define EDGE_PASS_GUARD (1 << 20)
#define EDGE_ROUND 4          // closure passes queued per host round trip
#define CLOSE_TW 64
#define CLOSE_TH 32
#define M_STRIDE (CLOSE_TW + 8)   // label tile row: ring column -1 at byte 3, interior at bytes 4..67 (4-byte aligned), ring column 64 at byte 68
#define MCOL(lx) ((lx) + 3)      // lx = 0..CLOSE_TW+1 in ring coordinates

struct EdgeLevel {
  int W, H;                 // level image
  int grid, gcols, grows;   // occupancy grid of the level (FeatureExtractor ctor :393-400)
  int tiles_x, tiles_y, tile_base;   // 64x32 tiles; tile_base = first tile of the level in a frame's tile tables
  uint32_t gx_off, gy_off;  // Sobel images inside a frame
  int gs;                   // their row stride in pixels (PyrGeom::sob_stride)
  size_t o_map, o_res, o_out, o_have;  // offsets inside a frame's edgelet slice
  size_t o_fmask;           // FAST mask inside a frame's FAST slice
  int wpr;
};

struct EdgeArgs {
  EdgeLevel lv[EDGE_LEVELS];
  const uint8_t* const* bases;
  char* fast_work; size_t fast_per_frame;
  char* work; size_t per_frame;      // edgelet slices
  size_t o_weak, o_chg;              // per frame: weak[tiles], chg[EDGE_ROUND + 1][tiles]
  int n_tiles;                       // tiles of all levels of one frame
  int* flags;                        // [EDGE_ROUND + 1], per round: flags[k + 1] != 0: pass k of the round changed a pixel; flags[0] = the round before
  int* totals;                       // [n_frames][n_levels]
  int n_levels, low, high, cap, pass, slot;
  const int* low_tab; const int* high_tab;   // the Canny thresholds of every frame (null: low / high for all)
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

// non-maximum suppression of one pixel; (ly, lx) index s_g
__device__ __forceinline__ uint8_t canny_label(const uint32_t (*s_g)[CLOSE_TW + 4], int ly, int lx, int low, int high)
{
  const int TG22 = 13573;                        // (int)(0.41421356... * 2^15 + 0.5)
  const uint32_t g = s_g[ly][lx];
  const int m = mag_of(g);
  if (!(m > low)) return 1;
  const int xs = (int)(short)(g & 0xffffu), ys = (int)(short)(g >> 16);
  const int ax = abs(xs), ay = abs(ys) << 15;
  const int tg22x = ax * TG22;
  bool keep;
  if (ay < tg22x) {
    keep = m > mag_of(s_g[ly][lx - 1]) && m >= mag_of(s_g[ly][lx + 1]);
  } else {
    const int tg67x = tg22x + (ax << 16);
    if (ay > tg67x) {
      keep = m > mag_of(s_g[ly - 1][lx]) && m >= mag_of(s_g[ly + 1][lx]);
    } else {
      const int s = (xs ^ ys) < 0 ? -1 : 1;
      keep = m > mag_of(s_g[ly - 1][lx - s]) && m > mag_of(s_g[ly + 1][lx + s]);
    }
  }
  if (!keep) return 1;
  return m > high ? 2 : 0;
}

// Closure of the tile interior (rows 1..CLOSE_TH, columns 1..CLOSE_TW of s_m) against its ring, as
// bit-board arithmetic in ONE wavefront: lane = label row, a 64-bit word per lane holds the row's
// 64 interior columns (edge bits E, weak bits Wk); the ring columns are two constant source bits.
// One step ORs the rows above and below (two lane shuffles), spreads by one column, and floods every
// weak run that touches a source along the row with a 6-step Kogge-Stone fill in each direction —
// so a step costs ~100 scalar-like instructions and no barrier, and the number of steps is the
// number of ROW changes along the longest weak chain, not its length.  Result in s_E / s_W (final
// edge / still-weak bits per row); all threads call, wave 0 works.
__device__ __forceinline__ void close_tile(const uint8_t (*s_m)[M_STRIDE], unsigned long long* s_E, unsigned long long* s_W)
{
  if (threadIdx.x < 64) {
    const int r = threadIdx.x;
    unsigned long long E = 0, Wk = 0;
    unsigned hl = 0, hr = 0;
    if (r < CLOSE_TH + 2) {
#pragma unroll 8
      for (int c = 0; c < CLOSE_TW; c++) {
        const uint8_t v = s_m[r][MCOL(c + 1)];
        E |= (unsigned long long)(v == 2) << c;
        Wk |= (unsigned long long)(v == 0) << c;
      }
      hl = s_m[r][MCOL(0)] == 2; hr = s_m[r][MCOL(CLOSE_TW + 1)] == 2;
      if (r == 0 || r == CLOSE_TH + 1) Wk = 0;               // ring rows are sources only
    }
    // ring columns seen from row r: rows r-1, r, r+1 (lanes beyond the tile hold zeros)
    const unsigned hl_up = __shfl_up(hl, 1), hl_dn = __shfl_down(hl, 1), hr_up = __shfl_up(hr, 1), hr_dn = __shfl_down(hr, 1);
    const unsigned long long ring = ((hl | (r > 0 ? hl_up : 0u) | hl_dn) ? 1ull : 0ull) | ((hr | (r > 0 ? hr_up : 0u) | hr_dn) ? (1ull << 63) : 0ull);
    for (;;) {
      unsigned long long up = __shfl_up(E, 1), dn = __shfl_down(E, 1);
      if (r == 0) up = 0;
      if (r == 63) dn = 0;
      const unsigned long long X = E | up | dn;
      unsigned long long gen = Wk & (X | (X << 1) | (X >> 1) | ring);
      const int any = __any(gen != 0);
      if (!any) break;
      // flood the weak runs that hold a seed: towards higher columns, then lower
      unsigned long long g = gen, p = Wk;
      g |= p & (g << 1); p &= p << 1;
      g |= p & (g << 2); p &= p << 2;
      g |= p & (g << 4); p &= p << 4;
      g |= p & (g << 8); p &= p << 8;
      g |= p & (g << 16); p &= p << 16;
      g |= p & (g << 32);
      p = Wk;
      g |= p & (g >> 1); p &= p >> 1;
      g |= p & (g >> 2); p &= p >> 2;
      g |= p & (g >> 4); p &= p >> 4;
      g |= p & (g >> 8); p &= p >> 8;
      g |= p & (g >> 16); p &= p >> 16;
      g |= p & (g >> 32);
      E |= g; Wk &= ~g;
    }
    if (r < CLOSE_TH + 2) { s_E[r] = E; s_W[r] = Wk; }
  }
  __syncthreads();
}

// labels of a 64x32 tile and its ring from the packed gradients of the tile + 2 rings, then the
// tile's own closure while the labels are still in LDS: the map reaches HBM tile-closed
__global__ __launch_bounds__(256) void k_canny_nms(EdgeArgs A)
{
  __shared__ uint32_t s_g[CLOSE_TH + 4][CLOSE_TW + 4];   // packed (gx, gy); zero outside the image = zero magnitude
  __shared__ __attribute__((aligned(8))) uint8_t s_m[CLOSE_TH + 2][M_STRIDE];
  __shared__ unsigned long long s_E[CLOSE_TH + 2], s_W[CLOSE_TH + 2];
  const EdgeLevel& L = level_of(A, blockIdx.y);
  const int W = L.W, H = L.H;
  if ((int)blockIdx.x >= L.tiles_x * L.tiles_y) return;
  const int ty = blockIdx.x / L.tiles_x, tx = blockIdx.x - ty * L.tiles_x;
  const int x0 = tx * CLOSE_TW, y0 = ty * CLOSE_TH;
  const uint8_t* base = A.bases[blockIdx.z];
  const int16_t* gxp = reinterpret_cast<const int16_t*>(base + L.gx_off);
  const int16_t* gyp = reinterpret_cast<const int16_t*>(base + L.gy_off);
  const int t = threadIdx.x;
  if ((W & 3) == 0) {
    // interior columns four pixels (8 bytes of gx, 8 of gy) at a time, the 2 + 2 ring columns singly
    for (int i = t; i < (CLOSE_TH + 4) * (CLOSE_TW / 4); i += 256) {
      const int ly = i / (CLOSE_TW / 4), q = i - ly * (CLOSE_TW / 4);
      const int y = y0 + ly - 2, x = x0 + 4 * q;
      uint2 a = make_uint2(0, 0), b = make_uint2(0, 0);
      if (y >= 0 && y < H && x < W) {
        const size_t o = (size_t)y * L.gs + x;
        a = *reinterpret_cast<const uint2*>(gxp + o);
        b = *reinterpret_cast<const uint2*>(gyp + o);
      }
      uint32_t* d = &s_g[ly][2 + 4 * q];
      d[0] = (a.x & 0xffffu) | (b.x << 16); d[1] = (a.x >> 16) | (b.x & 0xffff0000u);
      d[2] = (a.y & 0xffffu) | (b.y << 16); d[3] = (a.y >> 16) | (b.y & 0xffff0000u);
    }
    for (int i = t; i < (CLOSE_TH + 4) * 4; i += 256) {
      const int ly = i >> 2, h = i & 3;
      const int lx = h < 2 ? h : CLOSE_TW + h;
      const int y = y0 + ly - 2, x = x0 + lx - 2;
      uint32_t v = 0;
      if (x >= 0 && x < W && y >= 0 && y < H) {
        const size_t o = (size_t)y * L.gs + x;
        v = (uint32_t)(uint16_t)gxp[o] | ((uint32_t)(uint16_t)gyp[o] << 16);
      }
      s_g[ly][lx] = v;
    }
  } else {
    for (int i = t; i < (CLOSE_TH + 4) * (CLOSE_TW + 4); i += 256) {
      const int ly = i / (CLOSE_TW + 4), lx = i - ly * (CLOSE_TW + 4);
      const int y = y0 + ly - 2, x = x0 + lx - 2;
      uint32_t v = 0;
      if (x >= 0 && x < W && y >= 0 && y < H) {
        const size_t o = (size_t)y * L.gs + x;
        v = (uint32_t)(uint16_t)gxp[o] | ((uint32_t)(uint16_t)gyp[o] << 16);
      }
      s_g[ly][lx] = v;
    }
  }
  __syncthreads();
  const int low = A.low_tab ? A.low_tab[blockIdx.z] : A.low, high = A.high_tab ? A.high_tab[blockIdx.z] : A.high;
  for (int i = t; i < (CLOSE_TH + 2) * (CLOSE_TW + 2); i += 256) {
    const int ly = i / (CLOSE_TW + 2), lx = i - ly * (CLOSE_TW + 2);
    const int y = y0 + ly - 1, x = x0 + lx - 1;
    s_m[ly][MCOL(lx)] = (x >= 0 && x < W && y >= 0 && y < H) ? canny_label(s_g, ly + 1, lx + 1, low, high) : (uint8_t)1;
  }
  __syncthreads();
  close_tile(s_m, s_E, s_W);
  const int ly = (t >> 3) + 1, c0 = (t & 7) * 8;           // 8 consecutive interior pixels of one row
  uint8_t* slice = reinterpret_cast<uint8_t*>(A.work + (size_t)blockIdx.z * A.per_frame);
  uint8_t* map = slice + L.o_map;
  const int y = y0 + ly - 1, xb = x0 + c0;
  const unsigned eb = (unsigned)(s_E[ly] >> c0) & 0xffu, wb = (unsigned)(s_W[ly] >> c0) & 0xffu;
  int weak = 0;
  if (y < H) {
    if ((W & 7) == 0 && xb + 7 < W) {
      unsigned long long pk = 0;
#pragma unroll
      for (int k = 0; k < 8; k++) pk |= (unsigned long long)(((eb >> k) & 1u) ? 2u : (((wb >> k) & 1u) ? 0u : 1u)) << (8 * k);
      *reinterpret_cast<unsigned long long*>(map + (size_t)y * W + xb) = pk;
    } else {
#pragma unroll
      for (int k = 0; k < 8; k++)
        if (xb + k < W) map[(size_t)y * W + xb + k] = ((eb >> k) & 1u) ? 2 : (((wb >> k) & 1u) ? 0 : 1);
    }
    weak = wb != 0;
  }
  weak = __syncthreads_or(weak);
  if (t == 0) slice[A.o_weak + L.tile_base + blockIdx.x] = (uint8_t)(weak != 0);
}

// one whole-image pass of the closure across tile borders: a tile runs only while it still holds
// weak pixels and (after the first pass) one of its 8 neighbours grew an edge in the pass before
__global__ __launch_bounds__(256) void k_canny_close(EdgeArgs A)
{
  __shared__ __attribute__((aligned(8))) uint8_t s_m[CLOSE_TH + 2][M_STRIDE];
  __shared__ unsigned long long s_E[CLOSE_TH + 2], s_W[CLOSE_TH + 2];
  if (A.pass > 0 && A.flags[A.slot] == 0) return;      // the previous pass changed nothing: closed
  const EdgeLevel& L = level_of(A, blockIdx.y);
  const int W = L.W, H = L.H;
  if ((int)blockIdx.x >= L.tiles_x * L.tiles_y) return;
  const int ty = blockIdx.x / L.tiles_x, tx = blockIdx.x - ty * L.tiles_x;
  uint8_t* slice = reinterpret_cast<uint8_t*>(A.work + (size_t)blockIdx.z * A.per_frame);
  uint8_t* weak_flag = slice + A.o_weak + L.tile_base + blockIdx.x;
  if (!*weak_flag) return;
  if (A.pass > 0) {
    const uint8_t* chg = slice + A.o_chg + (size_t)A.slot * A.n_tiles + L.tile_base;
    int any = 0;
    for (int dy = -1; dy <= 1; dy++)
      for (int dx = -1; dx <= 1; dx++) {
        const int ny = ty + dy, nx = tx + dx;
        if ((dy | dx) != 0 && ny >= 0 && ny < L.tiles_y && nx >= 0 && nx < L.tiles_x) any |= chg[ny * L.tiles_x + nx];
      }
    if (!any) return;
  }
  const int x0 = tx * CLOSE_TW, y0 = ty * CLOSE_TH;
  uint8_t* map = slice + L.o_map;
  const int t = threadIdx.x;
  if ((W & 3) == 0) {
    // interior columns as aligned dwords (x0 and W are multiples of 4), the two ring columns as bytes
    for (int i = t; i < (CLOSE_TH + 2) * (CLOSE_TW / 4); i += 256) {
      const int ly = i / (CLOSE_TW / 4), q = i - ly * (CLOSE_TW / 4);
      const int y = y0 + ly - 1, x = x0 + 4 * q;
      uint32_t v = 0x01010101u;
      if (y >= 0 && y < H && x < W) v = *reinterpret_cast<const uint32_t*>(map + (size_t)y * W + x);
      *reinterpret_cast<uint32_t*>(&s_m[ly][MCOL(1) + 4 * q]) = v;
    }
    for (int i = t; i < (CLOSE_TH + 2) * 2; i += 256) {
      const int ly = i >> 1, lx = (i & 1) ? CLOSE_TW + 1 : 0;
      const int y = y0 + ly - 1, x = x0 + lx - 1;
      s_m[ly][MCOL(lx)] = (x >= 0 && x < W && y >= 0 && y < H) ? map[(size_t)y * W + x] : (uint8_t)1;
    }
  } else {
    for (int i = t; i < (CLOSE_TH + 2) * (CLOSE_TW + 2); i += 256) {
      const int ly = i / (CLOSE_TW + 2), lx = i - ly * (CLOSE_TW + 2);
      const int y = y0 + ly - 1, x = x0 + lx - 1;
      s_m[ly][MCOL(lx)] = (x >= 0 && x < W && y >= 0 && y < H) ? map[(size_t)y * W + x] : (uint8_t)1;
    }
  }
  __syncthreads();
  close_tile(s_m, s_E, s_W);
  const int ly = (t >> 3) + 1, c0 = (t & 7) * 8;         // 8 consecutive interior pixels of one row
  const unsigned eb = (unsigned)(s_E[ly] >> c0) & 0xffu;
  const int y = y0 + ly - 1;
  unsigned grown = 0;
#pragma unroll
  for (int k = 0; k < 8; k++)
    if (((eb >> k) & 1u) && s_m[ly][MCOL(c0 + 1 + k)] == 0) {   // was weak (hence inside the image), now an edge
      map[(size_t)y * W + (x0 + c0 + k)] = 2;
      grown = 1;
    }
  int weak = ((unsigned)(s_W[ly] >> c0) & 0xffu) != 0;
  const int grew = __syncthreads_or(grown != 0);
  weak = __syncthreads_or(weak);
  if (t == 0) {
    *weak_flag = (uint8_t)(weak != 0);
    if (grew) {
      slice[A.o_chg + (size_t)(A.slot + 1) * A.n_tiles + L.tile_base + blockIdx.x] = 1;
      A.flags[A.slot + 1] = 1;
    }
  }
}

__global__ __launch_bounds__(256) void k_edgelet_cells(EdgeArgs A)
{
  // 8 lanes per grid index, one per window row (the windows are at most 8 x 8); the row results
  // merge by shuffles: larger gradient wins, equal gradients keep the earlier raster position —
  // the serial scan's strict ">" rule
  const EdgeLevel& L = level_of(A, blockIdx.y);
  const int index = (blockIdx.x * 256 + threadIdx.x) >> 3, sub = threadIdx.x & 7;
  const bool live = index < L.gcols * L.grows;
  char* slice = A.work + (size_t)blockIdx.z * A.per_frame;
  const uint8_t* have = reinterpret_cast<const uint8_t*>(slice + L.o_have);
  const int W = L.W, H = L.H, g = L.grid;
  const int border = 8, maxBorderX = W - border, maxBorderY = H - border;
  float grad_best = 0.0f;
  int pos_best = 0x7fffffff, g_best = 0;                 // position y * W + x; packed (gx, gy)
  if (live) {
    int iniX = index % L.gcols * g;
    int iniY = index / L.grows * g;                      // sic, feature_detection.cpp:770
    if (!have[index] && !(iniX > maxBorderX || iniY > maxBorderY)) {
      const int maxX = min(iniX + g, maxBorderX), maxY = min(iniY + g, maxBorderY);
      iniX = max(iniX, border); iniY = max(iniY, border);
      const int y = iniY + sub;
      if (y < maxY) {
        const uint8_t* map = reinterpret_cast<const uint8_t*>(slice + L.o_map);
        const uint8_t* base = A.bases[blockIdx.z];
        const int16_t* gxp = reinterpret_cast<const int16_t*>(base + L.gx_off);
        const int16_t* gyp = reinterpret_cast<const int16_t*>(base + L.gy_off);
        for (int x = iniX; x < maxX; ++x) {
          const size_t o = (size_t)y * W + x;
          if (map[o] != 2) continue;
          const size_t og = (size_t)y * L.gs + x;
          const int sx = gxp[og], sy = gyp[og];
          const float grad = sqrtf((float)(sx * sx + sy * sy));
          if (grad > grad_best) { grad_best = grad; pos_best = (int)o; g_best = (sx & 0xffff) | (sy << 16); }
        }
      }
    }
  }
#pragma unroll
  for (int d = 1; d < 8; d <<= 1) {
    const float og = __shfl_xor(grad_best, d);
    const int op = __shfl_xor(pos_best, d), ogg = __shfl_xor(g_best, d);
    if (og > grad_best || (og == grad_best && op < pos_best)) { grad_best = og; pos_best = op; g_best = ogg; }
  }
  if (live && sub == 0) {
    hso_edgelet r;
    r.x = -1; r.y = -1; r.gx = 0; r.gy = 0; r.grad = 0.0f;
    if (grad_best > 0.0f) {
      const int y = pos_best / W;
      r.x = (int16_t)(pos_best - y * W); r.y = (int16_t)y;
      r.gx = (int16_t)(g_best & 0xffff); r.gy = (int16_t)(g_best >> 16);
      r.grad = grad_best;
    }
    reinterpret_cast<hso_edgelet*>(slice + L.o_res)[index] = r;
  }
}

#define PACK_THREADS 1024
__global__ __launch_bounds__(PACK_THREADS) void k_edgelet_pack(EdgeArgs A)
{
  __shared__ int s_w[PACK_THREADS / 64];
  const EdgeLevel& L = level_of(A, blockIdx.y);
  char* slice = A.work + (size_t)blockIdx.z * A.per_frame;
  const hso_edgelet* res = reinterpret_cast<const hso_edgelet*>(slice + L.o_res);
  hso_edgelet* out = reinterpret_cast<hso_edgelet*>(slice + L.o_out);
  const int cells = L.gcols * L.grows, cap = A.cap;
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  int base = 0;
  for (int c0 = 0; c0 < cells; c0 += PACK_THREADS) {
    const int i = c0 + t;
    hso_edgelet r;
    r.x = -1;
    if (i < cells) r = res[i];
    const bool set = r.x >= 0;
    const unsigned long long m = __ballot(set);
    if (lane == 0) s_w[wv] = __popcll(m);
    __syncthreads();
    int before = base, total = 0;
#pragma unroll
    for (int k = 0; k < PACK_THREADS / 64; k++) { const int v = s_w[k]; before += k < wv ? v : 0; total += v; }
    const int idx = before + __popcll(m & ((1ull << lane) - 1ull));
    if (set && idx < cap) out[idx] = r;
    base += total;
    __syncthreads();
  }
  if (t == 0) A.totals[(size_t)blockIdx.z * A.n_levels + blockIdx.y] = base;
}

// cv::Canny threshold preparation (L2gradient) for minThresh_: 31 x and 70 x, clamped to 32767, squared, floored
static void canny_thresholds(int min_thresh, int* low, int* high)
{
  double lo = 31.0 * min_thresh, hi = 70.0 * min_thresh;
  lo = lo < 32767.0 ? lo : 32767.0; hi = hi < 32767.0 ? hi : 32767.0;
  *low = (int)(lo * lo); *high = (int)(hi * hi);
}

// thresh_per_frame != null: every frame has its own minThresh_ (min_thresh is ignored)
static int detect_candidates_impl(hso_gpu_ctx* ctx, const int64_t* frame_ids, int n_frames, int n_levels, int min_thresh, const int32_t* thresh_per_frame,
                                  hso_corner* corners, int corner_cap, int32_t* corner_counts,
                                  hso_edgelet* edgelets, int edgelet_cap, int32_t* edgelet_counts)
{
  if (!ctx) return HSO_E_INVALID;
  if (!frame_ids || n_frames < 0 || n_levels < 1 || n_levels > EDGE_LEVELS || min_thresh < 0 || min_thresh > 255 || corner_cap < 0 ||
      edgelet_cap < 0 || !corner_counts || !edgelet_counts || (corner_cap > 0 && !corners) || (edgelet_cap > 0 && !edgelets))
    return hso_fail(ctx, HSO_E_INVALID, "detect_candidates: bad argument");
  if (n_frames == 0) return HSO_OK;
  std::vector<int32_t> tab3;   // [FAST threshold | Canny low | Canny high] per frame
  if (thresh_per_frame) {
    tab3.resize(3 * (size_t)n_frames);
    for (int i = 0; i < n_frames; i++) {
      if (thresh_per_frame[i] < 0 || thresh_per_frame[i] > 255) return hso_fail(ctx, HSO_E_INVALID, "detect_candidates: bad argument");
      tab3[(size_t)i] = thresh_per_frame[i];
      canny_thresholds(thresh_per_frame[i], &tab3[(size_t)n_frames + i], &tab3[2 * (size_t)n_frames + i]);
    }
  }
  auto it0 = ctx->frames.find(frame_ids[0]);
  if (it0 == ctx->frames.end()) return hso_fail(ctx, HSO_E_NOFRAME, "detect_candidates: frame not resident");
  const PyrGeom g = it0->second.g;
  auto al = [](size_t b) { return (b + 255) & ~size_t(255); };
  // extra area: [have flags of every frame | pass flags] (one memset), edgelet totals, then the frame slices
  EdgeArgs A{};
  size_t have_bytes = 0, o = 0;
  int vw = g.w[0], vh = g.h[0], max_cells = 0, max_close = 0, max_words = 0, n_tiles = 0;
  for (int l = 0; l < n_levels; l++) {
    EdgeLevel& L = A.lv[l];
    L.W = g.w[l]; L.H = g.h[l];
    L.grid = 8 / (1 << l);                                   // gridSize_ = 8, :395
    L.gcols = (vw + L.grid - 1) / L.grid; L.grows = (vh + L.grid - 1) / L.grid;   // vecWidth_[l] = vecWidth_[l-1] / 2, :362-366
    vw /= 2; vh /= 2;
    L.gx_off = g.sob_off[l][0]; L.gy_off = g.sob_off[l][1]; L.gs = g.sob_stride[l];
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
    L.tiles_x = (L.W + CLOSE_TW - 1) / CLOSE_TW; L.tiles_y = (L.H + CLOSE_TH - 1) / CLOSE_TH;
    L.tile_base = n_tiles;
    const int cl = L.tiles_x * L.tiles_y;
    n_tiles += cl;
    max_close = cl > max_close ? cl : max_close;
  }
  // tile tables behind the maps: weak[n_tiles], chg[EDGE_ROUND + 1][n_tiles]
  A.n_tiles = n_tiles;
  A.o_weak = o; o += al((size_t)n_tiles);
  A.o_chg = o; o += al((size_t)(EDGE_ROUND + 1) * n_tiles);
  const size_t chg_bytes = (size_t)(EDGE_ROUND + 1) * n_tiles;
  // have flags live per frame in front of the slices: offsets above are relative to a frame's have block
  const size_t have_per_frame = have_bytes;
  const size_t slice = al(o + have_per_frame);
  for (int l = 0; l < n_levels; l++) A.lv[l].o_have += o;    // have block sits at the end of each slice
  const size_t o_flags = 0, o_totals = al(sizeof(int) * (EDGE_ROUND + 1));
  const size_t o_slices = o_totals + al(sizeof(int) * (size_t)n_frames * n_levels);
  const size_t extra = o_slices + slice * (size_t)n_frames;

  FastPlan P;
  const int rc = hso_fast_enqueue(ctx, frame_ids, n_frames, n_levels, min_thresh, 8, corner_cap, extra, &P,   // fastThresh = floor(minThresh_), border 8 (:520-522)
                                  thresh_per_frame ? tab3.data() : nullptr);
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
  canny_thresholds(min_thresh, &A.low, &A.high);
  if (thresh_per_frame) {
    A.low_tab = reinterpret_cast<const int*>(P.d + P.o_thr) + n_frames;
    A.high_tab = reinterpret_cast<const int*>(P.d + P.o_thr) + 2 * (size_t)n_frames;
  }
  HSO_HIP_CHECK(ctx, hipMemsetAsync(x + o_flags, 0, sizeof(int) * (EDGE_ROUND + 1), ctx->stream));
  HSO_HIP_CHECK(ctx, hipMemset2DAsync(A.work + o, slice, 0, have_per_frame, (size_t)n_frames, ctx->stream));
  const dim3 blk(256);
  hipLaunchKernelGGL(k_cell_mark, dim3((max_words + 255) / 256, n_levels, n_frames), blk, 0, ctx->stream, A);
  HSO_HIP_CHECK(ctx, hipMemset2DAsync(A.work + A.o_chg, slice, 0, chg_bytes, (size_t)n_frames, ctx->stream));
  hipLaunchKernelGGL(k_canny_nms, dim3(max_close, n_levels, n_frames), blk, 0, ctx->stream, A);
  int pass = 0, flag = 1;
  int32_t* h_flags = reinterpret_cast<int32_t*>(hso_pinned(ctx, 1, 64));   // private flag word (never the caller's output array)
  if (!h_flags) return HSO_E_NOMEM;
  while (flag) {
    if (pass > EDGE_PASS_GUARD) return hso_fail(ctx, HSO_E_HIP, "detect_candidates: edge closure flag never cleared");
    if (pass > 0) {
      // next round: the last pass's flag and tile changes become slot 0, the other slots start clear
      HSO_HIP_CHECK(ctx, hipMemcpyAsync(A.flags, A.flags + EDGE_ROUND, sizeof(int), hipMemcpyDeviceToDevice, ctx->stream));
      HSO_HIP_CHECK(ctx, hipMemsetAsync(A.flags + 1, 0, sizeof(int) * EDGE_ROUND, ctx->stream));
      // next round: the last pass's tile changes become slot 0, the other slots start clear
      HSO_HIP_CHECK(ctx, hipMemcpy2DAsync(A.work + A.o_chg, slice, A.work + A.o_chg + (size_t)EDGE_ROUND * n_tiles, slice, (size_t)n_tiles,
                                          (size_t)n_frames, hipMemcpyDeviceToDevice, ctx->stream));
      HSO_HIP_CHECK(ctx, hipMemset2DAsync(A.work + A.o_chg + n_tiles, slice, 0, (size_t)EDGE_ROUND * n_tiles, (size_t)n_frames, ctx->stream));
    }
    for (int k = 0; k < EDGE_ROUND; k++, pass++) {
      A.pass = pass; A.slot = k;
      hipLaunchKernelGGL(k_canny_close, dim3(max_close, n_levels, n_frames), blk, 0, ctx->stream, A);
    }
    HSO_HIP_CHECK(ctx, hipGetLastError());
    HSO_HIP_CHECK(ctx, hipMemcpyAsync(h_flags, A.flags + EDGE_ROUND, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    if (pass == EDGE_ROUND) {
      // optimistic: queue the rest behind the flag read; redone below in the rare case the closure needed more passes
      hipLaunchKernelGGL(k_edgelet_cells, dim3((max_cells * 8 + 255) / 256, n_levels, n_frames), blk, 0, ctx->stream, A);
      hipLaunchKernelGGL(k_edgelet_pack, dim3(1, n_levels, n_frames), dim3(PACK_THREADS), 0, ctx->stream, A);
    }
    HSO_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    flag = h_flags[0];
  }
  if (pass > EDGE_ROUND) {
    hipLaunchKernelGGL(k_edgelet_cells, dim3((max_cells * 8 + 255) / 256, n_levels, n_frames), blk, 0, ctx->stream, A);
    hipLaunchKernelGGL(k_edgelet_pack, dim3(1, n_levels, n_frames), dim3(PACK_THREADS), 0, ctx->stream, A);
  }
  HSO_HIP_CHECK(ctx, hipGetLastError());
  HSO_HIP_CHECK(ctx, hipMemcpyAsync(edgelet_counts, A.totals, sizeof(int) * (size_t)n_frames * n_levels, hipMemcpyDeviceToHost, ctx->stream));
  std::vector<HsoListCopy> lists;
  const int rc2 = hso_fast_collect(ctx, P, corners, corner_counts, &lists);     // synchronises: both count tables are on the host
  if (rc2 != HSO_OK) return rc2;
  for (int i = 0; i < n_frames && edgelet_cap > 0; i++)
    for (int l = 0; l < n_levels; l++) {
      const int c = edgelet_counts[(size_t)i * n_levels + l];
      const int n = c < edgelet_cap ? c : edgelet_cap;
      if (n > 0) lists.push_back({edgelets + ((size_t)i * n_levels + l) * edgelet_cap, A.work + (size_t)i * slice + A.lv[l].o_out, sizeof(hso_edgelet) * (size_t)n});
    }
  return hso_lists_to_host(ctx, lists);     // corners and edgelets of every frame and level: one DMA
}

extern "C" int hso_gpu_detect_candidates(hso_gpu_ctx* ctx, const int64_t* frame_ids, int n_frames, int n_levels, int min_thresh,
                                         hso_corner* corners, int corner_cap, int32_t* corner_counts,
                                         hso_edgelet* edgelets, int edgelet_cap, int32_t* edgelet_counts)
{
  return detect_candidates_impl(ctx, frame_ids, n_frames, n_levels, min_thresh, nullptr, corners, corner_cap, corner_counts, edgelets, edgelet_cap, edgelet_counts);
}

extern "C" int hso_gpu_detect_candidates_multi(hso_gpu_ctx* ctx, const int64_t* frame_ids, int n_frames, int n_levels, const int32_t* min_thresh,
                                               hso_corner* corners, int corner_cap, int32_t* corner_counts,
                                               hso_edgelet* edgelets, int edgelet_cap, int32_t* edgelet_counts)
{
  if (!ctx) return HSO_E_INVALID;
  if (n_frames > 0 && !min_thresh) return hso_fail(ctx, HSO_E_INVALID, "detect_candidates: bad argument");
  return detect_candidates_impl(ctx, frame_ids, n_frames, n_levels, 0, min_thresh, corners, corner_cap, corner_counts, edgelets, edgelet_cap, edgelet_counts);
}

// ---------------------------------------------------------------------------------------------
// The initialisation branch of FeatureExtractor::detect (reference src/feature_detection.cpp:439-442):
// fastDetectMT as above, then fillingHole on level 0 (:1125-1154) — FAST-12 at barrier
// max(0.6 * minThresh, 6), non-maximum suppression, border, and a survivor is kept only when its
// grid index is still free, occupying it (first come in raster order).  On the device: the FAST-12
// survivors are emitted in raster order, an atomicMin per free grid index finds the first of them,
// and an ordered compaction keeps exactly those.
struct FillArgs {
  const hso_corner* list; size_t list_stride;     // FAST-12 survivors of level 0 per frame (bytes between frames)
  const int* totals;                              // [n_frames] survivors found
  int list_cap;
  char* work; size_t per_frame;                   // edgelet-style slices: have flags, first[], out[]
  size_t o_have, o_first, o_out;
  int grid, gcols, grows, cells, cap;
  int* out_totals;                                // [n_frames]
};

__global__ __launch_bounds__(256) void k_fill_first(FillArgs A)
{
  const int i = blockIdx.x * 256 + threadIdx.x;
  const int n = min(A.totals[blockIdx.y], A.list_cap);
  if (i >= n) return;
  const hso_corner c = reinterpret_cast<const hso_corner*>(reinterpret_cast<const char*>(A.list) + (size_t)blockIdx.y * A.list_stride)[i];
  char* slice = A.work + (size_t)blockIdx.y * A.per_frame;
  const int idx = c.y / A.grid * A.gcols + c.x / A.grows;                       // getCellIndex
  if (!reinterpret_cast<const uint8_t*>(slice + A.o_have)[idx]) atomicMin(reinterpret_cast<int*>(slice + A.o_first) + idx, i);
}

__global__ __launch_bounds__(PACK_THREADS) void k_fill_pack(FillArgs A)
{
  __shared__ int s_w[PACK_THREADS / 64];
  const int n = min(A.totals[blockIdx.x], A.list_cap);
  const hso_corner* list = reinterpret_cast<const hso_corner*>(reinterpret_cast<const char*>(A.list) + (size_t)blockIdx.x * A.list_stride);
  char* slice = A.work + (size_t)blockIdx.x * A.per_frame;
  const uint8_t* have = reinterpret_cast<const uint8_t*>(slice + A.o_have);
  const int* first = reinterpret_cast<const int*>(slice + A.o_first);
  hso_corner* out = reinterpret_cast<hso_corner*>(slice + A.o_out);
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  int base = 0;
  for (int c0 = 0; c0 < n; c0 += PACK_THREADS) {
    const int i = c0 + t;
    hso_corner c;
    bool keep = false;
    if (i < n) {
      c = list[i];
      const int idx = c.y / A.grid * A.gcols + c.x / A.grows;
      keep = !have[idx] && first[idx] == i;
    }
    const unsigned long long m = __ballot(keep);
    if (lane == 0) s_w[wv] = __popcll(m);
    __syncthreads();
    int before = base, total = 0;
#pragma unroll
    for (int k = 0; k < PACK_THREADS / 64; k++) { const int v = s_w[k]; before += k < wv ? v : 0; total += v; }
    const int idx_out = before + __popcll(m & ((1ull << lane) - 1ull));
    if (keep && idx_out < A.cap) out[idx_out] = c;
    base += total;
    __syncthreads();
  }
  if (t == 0) A.out_totals[blockIdx.x] = base;
}

extern "C" int hso_gpu_detect_candidates_init(hso_gpu_ctx* ctx, const int64_t* frame_ids, int n_frames, int n_levels, int min_thresh,
                                              hso_corner* corners, int corner_cap, int32_t* corner_counts,
                                              hso_corner* fill, int fill_cap, int32_t* fill_counts)
{
  if (!ctx) return HSO_E_INVALID;
  if (!frame_ids || n_frames < 0 || n_levels < 1 || n_levels > EDGE_LEVELS || min_thresh < 0 || min_thresh > 255 || corner_cap < 0 ||
      fill_cap < 0 || !corner_counts || !fill_counts || (corner_cap > 0 && !corners) || (fill_cap > 0 && !fill))
    return hso_fail(ctx, HSO_E_INVALID, "detect_candidates_init: bad argument");
  if (n_frames == 0) return HSO_OK;
  auto it0 = ctx->frames.find(frame_ids[0]);
  if (it0 == ctx->frames.end()) return hso_fail(ctx, HSO_E_NOFRAME, "detect_candidates_init: frame not resident");
  const PyrGeom g = it0->second.g;
  auto al = [](size_t b) { return (b + 255) & ~size_t(255); };
  EdgeArgs A{};
  size_t o = 0;
  int vw = g.w[0], vh = g.h[0], max_words = 0;
  for (int l = 0; l < n_levels; l++) {
    EdgeLevel& L = A.lv[l];
    L.W = g.w[l]; L.H = g.h[l];
    L.grid = 8 / (1 << l);
    L.gcols = (vw + L.grid - 1) / L.grid; L.grows = (vh + L.grid - 1) / L.grid;
    vw /= 2; vh /= 2;
    if ((L.H - 1) / L.grid * L.gcols + (L.W - 1) / L.grows >= L.gcols * L.grows)
      return hso_fail(ctx, HSO_E_INVALID, "detect_candidates_init: grid index out of range for this image size");
    L.o_have = o; o += al((size_t)L.gcols * L.grows);
  }
  const size_t have_bytes = o;
  const int cells0 = A.lv[0].gcols * A.lv[0].grows;
  const size_t o_first = o; o += al(sizeof(int) * (size_t)cells0);
  const size_t o_fout = o; o += al(sizeof(hso_corner) * (size_t)fill_cap);
  const size_t slice = o;
  // FAST-12 survivors of level 0: at most one per 2x2 pixels survives a 3x3 non-maximum suppression
  const int cap12 = (g.w[0] / 2 + 1) * (g.h[0] / 2 + 1);
  FastPlan P12;
  const size_t bytes12 = hso_fast_plan(g, n_frames, 1, cap12, &P12);
  const size_t o_p12 = 0, o_slices = al(bytes12), o_ftot = o_slices + slice * (size_t)n_frames;
  const size_t extra = o_ftot + al(sizeof(int) * (size_t)n_frames);

  FastPlan P;
  const int rc = hso_fast_enqueue(ctx, frame_ids, n_frames, n_levels, min_thresh, 8, corner_cap, extra, &P);
  if (rc != HSO_OK) return rc;
  char* x = P.d + P.o_extra;
  for (int l = 0; l < n_levels; l++) {
    A.lv[l].o_fmask = P.o_mask[l]; A.lv[l].wpr = P.wpr[l];
    const int words = A.lv[l].H * P.wpr[l];
    max_words = words > max_words ? words : max_words;
  }
  const uint8_t* const* d_bases = reinterpret_cast<const uint8_t* const*>(P.d + P.o_tab);
  A.bases = d_bases;
  A.fast_work = P.d; A.fast_per_frame = P.per_frame;
  A.work = x + o_slices; A.per_frame = slice;
  A.n_levels = n_levels;
  HSO_HIP_CHECK(ctx, hipMemset2DAsync(A.work, slice, 0, have_bytes, (size_t)n_frames, ctx->stream));
  HSO_HIP_CHECK(ctx, hipMemset2DAsync(A.work + o_first, slice, 0x7f, sizeof(int) * (size_t)cells0, (size_t)n_frames, ctx->stream));
  hipLaunchKernelGGL(k_cell_mark, dim3((max_words + 255) / 256, n_levels, n_frames), dim3(256), 0, ctx->stream, A);
  // const short fastThresh = 0.6*minThresh_ > 6 ? 0.6*minThresh_ : 6, :1129
  const int thr12 = (int)(short)(0.6 * min_thresh > 6 ? 0.6 * min_thresh : 6);
  P12.d = x + o_p12;
  const int rc12 = hso_fast_launch(ctx, P12, d_bases, thr12, 8, 12);
  if (rc12 != HSO_OK) return rc12;
  FillArgs F;
  F.list = reinterpret_cast<const hso_corner*>(P12.d + P12.o_out[0]); F.list_stride = P12.per_frame;
  F.totals = reinterpret_cast<const int*>(P12.d + P12.o_tot);
  F.list_cap = cap12;
  F.work = A.work; F.per_frame = slice;
  F.o_have = A.lv[0].o_have; F.o_first = o_first; F.o_out = o_fout;
  F.grid = A.lv[0].grid; F.gcols = A.lv[0].gcols; F.grows = A.lv[0].grows; F.cells = cells0; F.cap = fill_cap;
  F.out_totals = reinterpret_cast<int*>(x + o_ftot);
  hipLaunchKernelGGL(k_fill_first, dim3((cap12 + 255) / 256, n_frames), dim3(256), 0, ctx->stream, F);
  hipLaunchKernelGGL(k_fill_pack, dim3(n_frames), dim3(PACK_THREADS), 0, ctx->stream, F);
  HSO_HIP_CHECK(ctx, hipGetLastError());
  HSO_HIP_CHECK(ctx, hipMemcpyAsync(fill_counts, F.out_totals, sizeof(int) * (size_t)n_frames, hipMemcpyDeviceToHost, ctx->stream));
  std::vector<HsoListCopy> lists;
  const int rc2 = hso_fast_collect(ctx, P, corners, corner_counts, &lists);     // synchronises: both count tables are on the host
  if (rc2 != HSO_OK) return rc2;
  for (int i = 0; i < n_frames && fill_cap > 0; i++) {
    const int n = fill_counts[i] < fill_cap ? fill_counts[i] : fill_cap;
    if (n > 0) lists.push_back({fill + (size_t)i * fill_cap, A.work + (size_t)i * slice + o_fout, sizeof(hso_corner) * (size_t)n});
  }
  return hso_lists_to_host(ctx, lists);
}
