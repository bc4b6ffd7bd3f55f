// hso_frame.hip — Frame construction on gfx950: 5-level u8 pyramid, 5x5 Sobel of
// levels 0-2 and the two frame statistics.
//
// Replaces Frame::initFrame (reference src/frame.cpp:82-96):
//   frame_utils::createImgPyramid (src/frame.cpp:296-314) -> hso::halfSample
//   (src/vikit/vision.cpp:19-44 SSE2 rounding, :92-107 scalar truncation) and
//   Frame::prepareForFeatureDetect (src/frame.cpp:205-246).
// All of it is integer work and bit-exact with the CPU restatement, except the
// two means, which the reference accumulates serially in fp32 and this kernel
// accumulates exactly (integer) / in fp64 with a fixed reduction tree.
//
// Roofline: HBM-bound.  Per frame W*H bytes are read once for the pyramid
// (1.33*W*H written), levels 0-2 are read once more for Sobel (1.31*W*H) and
// 5.25*W*H bytes of int16 gradients are written.
#include "hso_ctx.h"
#include "hso_dev_math.h"

using namespace hso_dev;

// ------------------------------------------------------------------ pyramid

HSO_DEV uint32_t byte_of(uint32_t w, int k) { return (w >> (8 * k)) & 0xffu; }

// One output pixel of halfSample.  sse: round-half-up twice
// (_mm_avg_epu8 then _mm_avg_epu16, vision.cpp:32-35); else (a+b+c+d)/4 (vision.cpp:100).
HSO_DEV uint32_t hs_px(uint32_t a, uint32_t b, uint32_t c, uint32_t d, bool sse)
{
  return sse ? ((((a + c + 1u) >> 1) + ((b + d + 1u) >> 1) + 1u) >> 1) : ((a + b + c + d) >> 2);
}

template <int N, int WIN, int WOUT>
HSO_DEV void half_block(const uint32_t (&in)[N][WIN], uint32_t (&out)[N / 2][WOUT], bool sse)
{
#pragma unroll
  for (int i = 0; i < N / 2; i++) {
#pragma unroll
    for (int wj = 0; wj < WOUT; wj++) {
      uint32_t word = 0;
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const int j = wj * 4 + k;
        if (j < N / 2) {
          const uint32_t a = byte_of(in[2 * i][(2 * j) / 4], (2 * j) % 4);
          const uint32_t b = byte_of(in[2 * i][(2 * j + 1) / 4], (2 * j + 1) % 4);
          const uint32_t c = byte_of(in[2 * i + 1][(2 * j) / 4], (2 * j) % 4);
          const uint32_t d = byte_of(in[2 * i + 1][(2 * j + 1) / 4], (2 * j + 1) % 4);
          word |= hs_px(a, b, c, d, sse) << (8 * k);
        }
      }
      out[i][wj] = word;
    }
  }
}

// One thread owns one 16x16 cell of level 0 and produces its 8x8, 4x4, 2x2 and
// 1x1 descendants entirely in registers (a 2x2 box filter has no halo), so the
// level-0 image is read exactly once with 16-byte coalesced loads.
__global__ __launch_bounds__(256) void k_pyramid(PyrGeom g, uint8_t* const* bases, const uint8_t* const* srcs)
{
  const int cells_x = g.w[0] >> 4, cells_y = g.h[0] >> 4;
  const int cell = blockIdx.x * blockDim.x + threadIdx.x;
  if (cell >= cells_x * cells_y) return;
  const int cx = cell % cells_x, cy = cell / cells_x;
  uint8_t* base = bases[blockIdx.y];

  uint32_t r0[16][4];
  const size_t cell_off = (size_t)(cy * 16) * g.w[0] + cx * 16;
  uint8_t* l0 = base + g.off[0] + cell_off;
  // level 0 either already sits in the frame (host upload) or is pulled from a resident
  // source image and stored into the frame on the way (device upload, copy fused here)
  const uint8_t* src = srcs ? srcs[blockIdx.y] : nullptr;
  const uint8_t* in0 = src ? src + cell_off : l0;
#pragma unroll
  for (int i = 0; i < 16; i++) {
    const uint4 v = *reinterpret_cast<const uint4*>(in0 + (size_t)i * g.w[0]);
    r0[i][0] = v.x; r0[i][1] = v.y; r0[i][2] = v.z; r0[i][3] = v.w;
    if (src) *reinterpret_cast<uint4*>(l0 + (size_t)i * g.w[0]) = v;
  }
  uint32_t r1[8][2], r2[4][1], r3[2][1], r4[1][1];
  half_block<16, 4, 2>(r0, r1, (g.w[0] % 16) == 0);
  half_block<8, 2, 1>(r1, r2, (g.w[1] % 16) == 0);
  half_block<4, 1, 1>(r2, r3, (g.w[2] % 16) == 0);
  half_block<2, 1, 1>(r3, r4, (g.w[3] % 16) == 0);

  uint8_t* l1 = base + g.off[1] + (size_t)(cy * 8) * g.w[1] + cx * 8;
#pragma unroll
  for (int i = 0; i < 8; i++) *reinterpret_cast<uint2*>(l1 + (size_t)i * g.w[1]) = make_uint2(r1[i][0], r1[i][1]);
  uint8_t* l2 = base + g.off[2] + (size_t)(cy * 4) * g.w[2] + cx * 4;
#pragma unroll
  for (int i = 0; i < 4; i++) *reinterpret_cast<uint32_t*>(l2 + (size_t)i * g.w[2]) = r2[i][0];
  uint8_t* l3 = base + g.off[3] + (size_t)(cy * 2) * g.w[3] + cx * 2;
#pragma unroll
  for (int i = 0; i < 2; i++) *reinterpret_cast<uint16_t*>(l3 + (size_t)i * g.w[3]) = (uint16_t)r3[i][0];
  base[g.off[4] + (size_t)cy * g.w[4] + cx] = (uint8_t)r4[0][0];
}

// -------------------------------------------------------------------- Sobel

#define SOB_TW 64
#define SOB_TH 16

// cv::Sobel(CV_16S, ksize 5, BORDER_REPLICATE) for levels 0..2 (src/frame.cpp:216-220):
// derivative [-1 -2 0 2 1], smoothing [1 4 6 4 1], separable, exact integers.
// Level-0 blocks also emit one (sum intensity, sum |grad|) partial each over the
// 16-px-margin interior (src/frame.cpp:223-236).
__global__ __launch_bounds__(256) void k_sobel(PyrGeom g, uint8_t* const* bases)
{
  __shared__ uint8_t s_src[SOB_TH + 4][SOB_TW + 4];
  __shared__ short s_hd[SOB_TH + 4][SOB_TW];
  __shared__ short s_hs[SOB_TH + 4][SOB_TW];
  __shared__ double s_part[4][2];

  int b = blockIdx.x, level = 0;
  while (level < HSO_N_SOBEL_LEVELS - 1 && b >= g.sobel_blocks[level]) { b -= g.sobel_blocks[level]; level++; }
  const int W = g.w[level], H = g.h[level];
  const int bx = b % g.sobel_bx[level], by = b / g.sobel_bx[level];
  const int x0 = bx * SOB_TW, y0 = by * SOB_TH;
  uint8_t* base = bases[blockIdx.y];
  const uint8_t* img = base + g.off[level];
  const int t = threadIdx.x;

  for (int i = t; i < (SOB_TH + 4) * (SOB_TW + 4); i += 256) {
    const int ly = i / (SOB_TW + 4), lx = i % (SOB_TW + 4);
    int yy = y0 + ly - 2, xx = x0 + lx - 2;
    yy = yy < 0 ? 0 : (yy > H - 1 ? H - 1 : yy);
    xx = xx < 0 ? 0 : (xx > W - 1 ? W - 1 : xx);
    s_src[ly][lx] = img[(size_t)yy * W + xx];
  }
  __syncthreads();
  const int tx = t & 63, tg = t >> 6;
  for (int r = tg * 5; r < tg * 5 + 5; r++) {
    const int p0 = s_src[r][tx], p1 = s_src[r][tx + 1], p2 = s_src[r][tx + 2], p3 = s_src[r][tx + 3], p4 = s_src[r][tx + 4];
    s_hd[r][tx] = (short)(-p0 - 2 * p1 + 2 * p3 + p4);
    s_hs[r][tx] = (short)(p0 + 4 * p1 + 6 * p2 + 4 * p3 + p4);
  }
  __syncthreads();
  int16_t* gx = reinterpret_cast<int16_t*>(base + g.sob_off[level][0]);
  int16_t* gy = reinterpret_cast<int16_t*>(base + g.sob_off[level][1]);
  unsigned isum = 0;
  double gsum = 0;
  const int x = x0 + tx;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const int ry = tg * 4 + k;
    const int y = y0 + ry;
    const int sx = s_hd[ry][tx] + 4 * s_hd[ry + 1][tx] + 6 * s_hd[ry + 2][tx] + 4 * s_hd[ry + 3][tx] + s_hd[ry + 4][tx];
    const int sy = -s_hs[ry][tx] - 2 * s_hs[ry + 1][tx] + 2 * s_hs[ry + 3][tx] + s_hs[ry + 4][tx];
    if (x < W && y < H) {
      gx[(size_t)y * W + x] = (int16_t)sx;
      gy[(size_t)y * W + x] = (int16_t)sy;
      if (level == 0 && x >= 16 && x < W - 16 && y >= 16 && y < H - 16) {
        const float fx = (float)sx, fy = (float)sy;
        gsum += (double)sqrtf(fx * fx + fy * fy);
        isum += s_src[ry + 2][tx + 2];
      }
    }
  }
  if (level == 0) {
    double is = wave_sum_to_lane63((double)isum);
    double gs = wave_sum_to_lane63(gsum);
    if (tx == 63) { s_part[tg][0] = is; s_part[tg][1] = gs; }
    __syncthreads();
    if (t == 0) {
      double* part = reinterpret_cast<double*>(base + g.part_off);
      part[2 * b + 0] = ((s_part[0][0] + s_part[1][0]) + s_part[2][0]) + s_part[3][0];
      part[2 * b + 1] = ((s_part[0][1] + s_part[1][1]) + s_part[2][1]) + s_part[3][1];
    }
  }
}

// integralImage_ and gradMean_ (src/frame.cpp:238-245) from the per-block partials,
// summed in a fixed order (lane-strided, then the DPP tree).
__global__ __launch_bounds__(64) void k_frame_stats(PyrGeom g, uint8_t* const* bases, hso_frame_stats* stats_out)
{
  uint8_t* base = bases[blockIdx.x];
  const double* part = reinterpret_cast<const double*>(base + g.part_off);
  double is = 0, gs = 0;
  for (int i = threadIdx.x; i < g.sobel_blocks[0]; i += 64) { is += part[2 * i]; gs += part[2 * i + 1]; }
  is = wave_sum_to_lane63(is);
  gs = wave_sum_to_lane63(gs);
  if (threadIdx.x == 63) {
    const double cnt = (double)(g.w[0] - 32) * (double)(g.h[0] - 32);
    hso_frame_stats* st = reinterpret_cast<hso_frame_stats*>(base + g.stats_off);
    st->integral_image = (float)(is / cnt);
    float gm = (float)(gs / cnt);
    gm /= 30;
    if (gm > 20) gm = 20;
    if (gm < 7) gm = 7;
    st->grad_mean = gm;
    st->width = g.w[0];
    st->height = g.h[0];
    if (stats_out) stats_out[blockIdx.x] = *st;
  }
}

int hso_frame_build(hso_gpu_ctx* ctx, const PyrGeom& g, uint8_t* const* d_bases, const uint8_t* const* d_srcs,
                    hso_frame_stats* d_stats, int n)
{
  const int cells = (g.w[0] >> 4) * (g.h[0] >> 4);
  dim3 gp((cells + 255) / 256, n);
  hipLaunchKernelGGL(k_pyramid, gp, dim3(256), 0, ctx->stream, g, d_bases, d_srcs);
  const int sob_total = g.sobel_blocks[0] + g.sobel_blocks[1] + g.sobel_blocks[2];
  hipLaunchKernelGGL(k_sobel, dim3(sob_total, n), dim3(256), 0, ctx->stream, g, d_bases);
  hipLaunchKernelGGL(k_frame_stats, dim3(n), dim3(64), 0, ctx->stream, g, d_bases, d_stats);
  HSO_HIP_CHECK(ctx, hipGetLastError());
  return HSO_OK;
}

PyrGeom make_geom(int w, int h)
{
  PyrGeom g{};
  uint32_t off = 0;
  for (int l = 0; l < HSO_N_PYR_LEVELS; l++) {
    g.w[l] = w >> l; g.h[l] = h >> l;
    g.off[l] = off;
    const uint32_t bytes = (uint32_t)g.w[l] * g.h[l] + (uint32_t)g.w[l] + 64;  // + one zero row + slack
    off += (bytes + 255u) & ~255u;
  }
  g.pyr_bytes = off;
  for (int l = 0; l < HSO_N_SOBEL_LEVELS; l++) {
    g.sobel_bx[l] = (g.w[l] + SOB_TW - 1) / SOB_TW;
    g.sobel_blocks[l] = g.sobel_bx[l] * ((g.h[l] + SOB_TH - 1) / SOB_TH);
    for (int k = 0; k < 2; k++) {
      g.sob_off[l][k] = off;
      off += ((uint32_t)g.w[l] * g.h[l] * 2u + 255u) & ~255u;
    }
  }
  g.part_off = off;
  off += ((uint32_t)g.sobel_blocks[0] * 16u + 255u) & ~255u;
  g.stats_off = off;
  off += 256;
  g.frame_bytes = off;
  return g;
}
