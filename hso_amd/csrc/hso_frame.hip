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

// ---- cv::resize branch of createImgPyramid (src/frame.cpp:307-312) for level-0 sizes that are
// not multiples of 16 (e.g. TUM-mono: 1280x1024 is shrunk to 920x736 before it reaches the Frame).
// OpenCV's INTER_LINEAR for CV_8UC1 restated (imgproc/resize.cpp, C/SIMD path): an exact 2x2
// decimation takes INTER_AREA's fast path (a + b + c + d + 2) >> 2, anything else the fixed-point
// bilinear kernel (coefficients rounded to short at 11 bits, horizontal pass in int, vertical pass
// ((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2).  One thread per output pixel;
// levels are produced one after the other (level l reads level l-1).  Rare path: plain kernel.
__global__ __launch_bounds__(256) void k_copy_level0(PyrGeom g, uint8_t* const* bases, const uint8_t* const* srcs)
{
  const size_t n = (size_t)g.w[0] * g.h[0];
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) bases[blockIdx.y][g.off[0] + i] = srcs[blockIdx.y][i];
}

// one output pixel of cv::resize(src, dst, Size(dw, dh)) with INTER_LINEAR, CV_8UC1
HSO_DEV uint8_t resize_linear_px(const uint8_t* src, int sw, int sh, int dw, int dh, int dx, int dy)
{
  const double scale_x = 1. / ((double)dw / sw), scale_y = 1. / ((double)dh / sh);
  const int iscale_x = __double2int_rn(scale_x), iscale_y = __double2int_rn(scale_y);
  const bool area_fast = fabs(scale_x - iscale_x) < 2.220446049250313e-16 && fabs(scale_y - iscale_y) < 2.220446049250313e-16;
  if (area_fast && iscale_x == 2 && iscale_y == 2) {
    const uint8_t* s0 = src + (size_t)(2 * dy) * sw + 2 * dx;
    return (uint8_t)((s0[0] + s0[1] + s0[sw] + s0[sw + 1] + 2) >> 2);
  }
  float fx = (float)((dx + 0.5) * scale_x - 0.5);
  int sx = (int)floor((double)fx);
  fx -= (float)sx;
  if (sx < 0) { fx = 0; sx = 0; }
  const bool tail = sx + 1 >= sw;  // dx >= xmax: the horizontal pass copies S[sx] * 2048
  if (tail && sx >= sw - 1) { fx = 0; sx = sw - 1; }
  const int a0 = max(-32768, min(32767, __float2int_rn((1.f - fx) * 2048))), a1 = max(-32768, min(32767, __float2int_rn(fx * 2048)));
  float fy = (float)((dy + 0.5) * scale_y - 0.5);
  const int sy0 = (int)floor((double)fy);
  fy -= (float)sy0;
  const int b0 = max(-32768, min(32767, __float2int_rn((1.f - fy) * 2048))), b1 = max(-32768, min(32767, __float2int_rn(fy * 2048)));
  int r[2];
#pragma unroll
  for (int k = 0; k < 2; k++) {
    int sy = sy0 + k;
    sy = sy >= 0 ? (sy < sh ? sy : sh - 1) : 0;
    const uint8_t* S = src + (size_t)sy * sw;
    r[k] = tail ? (int)S[sx] * 2048 : (int)S[sx] * a0 + (int)S[sx + 1] * a1;
  }
  return (uint8_t)((((b0 * (r[0] >> 4)) >> 16) + ((b1 * (r[1] >> 4)) >> 16) + 2) >> 2);
}

__global__ __launch_bounds__(256) void k_resize_level(PyrGeom g, uint8_t* const* bases, int l)
{
  const int dw = g.w[l], dh = g.h[l], sw = g.w[l - 1], sh = g.h[l - 1];
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= dw * dh) return;
  const int dx = i % dw, dy = i / dw;
  uint8_t* base = bases[blockIdx.y];
  base[g.off[l] + (size_t)dy * dw + dx] = resize_linear_px(base + g.off[l - 1], sw, sh, dw, dh, dx, dy);
}

// ImageReader::readImage's cv::resize(image, image, m_img_new_size) (reference src/ImageReader.cpp:79):
// the camera file's > 848x800 rule shrinks the sensor image before it becomes a Frame
__global__ __launch_bounds__(256) void k_resize_image(const uint8_t* src, int sw, int sh, uint8_t* dst, int dw, int dh)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= dw * dh) return;
  const int dx = i % dw, dy = i / dw;
  dst[(size_t)dy * dw + dx] = resize_linear_px(src, sw, sh, dw, dh, dx, dy);
}

int hso_frame_resize_into(hso_gpu_ctx* ctx, const uint8_t* d_src, int sw, int sh, uint8_t* d_dst, int dw, int dh)
{
  hipLaunchKernelGGL(k_resize_image, dim3((dw * dh + 255) / 256), dim3(256), 0, ctx->stream, d_src, sw, sh, d_dst, dw, dh);
  HSO_HIP_CHECK(ctx, hipGetLastError());
  return HSO_OK;
}

// -------------------------------------------------------------------- Sobel

#define SOB_STRIP 15           // rows per lane
#define SOB_TW 64              // wavefront tile: 16 lanes x 4 pixels wide,
#define SOB_TH (4 * SOB_STRIP) //                 4 lane groups x SOB_STRIP rows high (60: divides 480, 240, 120)
#define SOB_WAVES 4            // tiles (wavefronts) per block

typedef const __attribute__((address_space(1))) uint8_t* SobGlbCU8;
typedef const __attribute__((address_space(1))) uint32_t* SobGlbCU32;
typedef __attribute__((address_space(1))) int16_t* SobGlbI16;
typedef __attribute__((address_space(1))) unsigned long long* SobGlbU64;

// One lane's strip on a level whose width is a multiple of 4 (every halfSample pyramid): four adjacent output pixels of
// SOB_STRIP rows from aligned dwords, no branch between a row's loads and the arithmetic of the rows before it, and the
// loads of row i + PF issued before row i is worked on.  Same arithmetic as the general strip inside k_sobel<false>.
// Measured on 4096 EuRoC frames (rocprofv3, profiles/r4_sobel_variants.md): PF 0 / 1 / 2 / 3 / 4 = 2.41 / 2.30 / 2.31 /
// 2.39 / 2.40 ms at 54 / 61 / 65 / 69 / 72 VGPRs; all 57 loads at the top (82 VGPRs) 2.79 ms; the general kernel 2.50 ms.
template <int PF>
HSO_DEV void sobel_strip(const SobGlbCU8 img, const SobGlbI16 gx, const SobGlbI16 gy, const int W, const int H, const int GS,
                         const int level, const int x, const int ys, unsigned& isum, double& gsum)
{
  typedef SobGlbCU8 GlbCU8;
  typedef SobGlbCU32 GlbCU32;
  typedef SobGlbU64 GlbU64;
  const bool has_left = x >= 4, has_right = x + 8 <= W;
  int o_left = has_left ? -1 : 0, o_right = has_right ? 1 : 0;
  // keep the compiler from turning the two neighbour loads back into loads under has_left / has_right (it does, and the
  // branches put every row's loads behind the previous row's arithmetic)
  asm volatile("" : "+v"(o_left), "+v"(o_right));
  // Packed 16-bit arithmetic: every intermediate fits int16 (|hd| <= 6*255, hs <= 16*255,
  // |gx|, |gy| <= 16*6*255 = 24480), so two pixels ride in one register (v_pk_*_i16) and the
  // output dwords come out already packed.  P[j] = {p[j], p[j+1]} for the columns x-2+j.
  typedef short s16x2 __attribute__((ext_vector_type(2)));
  s16x2 hd[5][2], hs[5][2];
  uint32_t cdw[5];     // bytes x .. x+3 of the row (the four centre pixels), for the intensity sum
  const s16x2 k2 = {2, 2}, k4 = {4, 4}, k6 = {6, 6};
  constexpr int NR = SOB_STRIP + 4;
  uint32_t ld[NR][3];   // statically indexed (the strip loop is fully unrolled): registers, live from issue to use
  auto issue = [&](const int i) {
    int yy = ys + i - 2;
    yy = yy < 0 ? 0 : (yy > H - 1 ? H - 1 : yy);
    const GlbCU8 row = img + (size_t)yy * W;
    // three unconditional loads (a lane on the image border re-reads its own dword and replicates its edge pixel
    // below, BORDER_REPLICATE)
    const GlbCU32 r32 = (GlbCU32)(row + x);
    ld[i][1] = r32[0];
    ld[i][0] = r32[o_left];
    ld[i][2] = r32[o_right];
  };
#pragma unroll
  for (int i = 0; i < PF; i++) issue(i);
#pragma unroll
  for (int i = 0; i < NR; i++) {
    // the loads of row i + PF go out before row i's arithmetic; the scheduling barriers keep the compiler from
    // moving them (it would otherwise issue all 57 at the top: 82 VGPRs, 2.79 ms against 2.49 ms)
    if (i + PF < NR) issue(i + PF);
    if (PF > 0) __builtin_amdgcn_sched_barrier(0);
    uint32_t d0 = ld[i][0], d1 = ld[i][1], d2 = ld[i][2];   // bytes x-4 .. x+7 (only x-2 .. x+5 are used)
    d0 = has_left ? d0 : (d1 & 0xffu) * 0x01010101u;
    d2 = has_right ? d2 : (d1 >> 24) * 0x01010101u;
    // v_perm_b32: bytes 0-3 of the selector space = second operand, 4-7 = first, 0x0c = zero
    union { uint32_t u; s16x2 v; } P0, P1, P2, P3, P4, P5, P6;
    P0.u = __builtin_amdgcn_perm(d0, d0, 0x0c030c02u);
    P1.u = __builtin_amdgcn_perm(d1, d0, 0x0c040c03u);
    P2.u = __builtin_amdgcn_perm(d1, d1, 0x0c010c00u);
    P3.u = __builtin_amdgcn_perm(d1, d1, 0x0c020c01u);
    P4.u = __builtin_amdgcn_perm(d1, d1, 0x0c030c02u);
    P5.u = __builtin_amdgcn_perm(d2, d1, 0x0c040c03u);
    P6.u = __builtin_amdgcn_perm(d2, d2, 0x0c010c00u);
    const int sl = i % 5;
    hd[sl][0] = (P4.v - P0.v) + k2 * (P3.v - P1.v);
    hd[sl][1] = (P6.v - P2.v) + k2 * (P5.v - P3.v);
    hs[sl][0] = (P0.v + P4.v) + k4 * (P1.v + P3.v) + k6 * P2.v;
    hs[sl][1] = (P2.v + P6.v) + k4 * (P3.v + P5.v) + k6 * P4.v;
    cdw[sl] = d1;
    if (i >= 4) {
      const int y = ys + i - 4;
      const int ra = (i - 4) % 5, rb = (i - 3) % 5, rc = (i - 2) % 5, rd = (i - 1) % 5, re = i % 5;
      if (y < H) {
        union { uint32_t u; s16x2 v; } sx0, sx1, sy0, sy1;
        sx0.v = (hd[ra][0] + hd[re][0]) + k4 * (hd[rb][0] + hd[rd][0]) + k6 * hd[rc][0];
        sx1.v = (hd[ra][1] + hd[re][1]) + k4 * (hd[rb][1] + hd[rd][1]) + k6 * hd[rc][1];
        sy0.v = (hs[re][0] - hs[ra][0]) + k2 * (hs[rd][0] - hs[rb][0]);
        sy1.v = (hs[re][1] - hs[ra][1]) + k2 * (hs[rd][1] - hs[rb][1]);
        typedef unsigned long long u64;
        __builtin_nontemporal_store(((u64)sx1.u << 32) | sx0.u, (GlbU64)(gx + (size_t)y * GS + x));
        __builtin_nontemporal_store(((u64)sy1.u << 32) | sy0.u, (GlbU64)(gy + (size_t)y * GS + x));
        if (level == 0 && x >= 16 && x < W - 16 && y >= 16 && y < H - 16) {
          // |grad| of the four pixels.  gx^2 + gy^2 <= 2 * 24480^2 < 2^31 is formed exactly by one
          // v_dot2_i32_i16 per pixel on the {gx, gy} pair (v_perm_b32 pairs them up), converted once
          // and rooted by the hardware v_sqrt_f32 (1 ulp).  The reference (src/frame.cpp:231) rounds
          // both squares and their sum to fp32 and adds 3e5 roots into ONE fp32 accumulator; the
          // difference of this form to a correctly rounded root is < 1e-7 of a term, four orders
          // below what the reference's own running fp32 sum loses (tests: means to 2e-5).
          union { uint32_t u; s16x2 v; } q0, q1, q2, q3;
          q0.u = __builtin_amdgcn_perm(sy0.u, sx0.u, 0x05040100u);
          q1.u = __builtin_amdgcn_perm(sy0.u, sx0.u, 0x07060302u);
          q2.u = __builtin_amdgcn_perm(sy1.u, sx1.u, 0x05040100u);
          q3.u = __builtin_amdgcn_perm(sy1.u, sx1.u, 0x07060302u);
          float mag[4];
          mag[0] = __builtin_amdgcn_sqrtf((float)__builtin_amdgcn_sdot2(q0.v, q0.v, 0, false));
          mag[1] = __builtin_amdgcn_sqrtf((float)__builtin_amdgcn_sdot2(q1.v, q1.v, 0, false));
          mag[2] = __builtin_amdgcn_sqrtf((float)__builtin_amdgcn_sdot2(q2.v, q2.v, 0, false));
          mag[3] = __builtin_amdgcn_sqrtf((float)__builtin_amdgcn_sdot2(q3.v, q3.v, 0, false));
          gsum += (double)((mag[0] + mag[1]) + (mag[2] + mag[3]));   // four values < 3.5e4 each: fp32 pair sums, then fp64
          isum = __builtin_amdgcn_sad_u8(cdw[rc], 0u, isum);   // + the four centre bytes
        }
      }
    }
  }
}

// cv::Sobel(CV_16S, ksize 5, BORDER_REPLICATE) for levels 0..2 (src/frame.cpp:216-220):
// derivative [-1 -2 0 2 1], smoothing [1 4 6 4 1], separable, exact integers.
// Level-0 blocks also emit one (sum intensity, sum |grad|) partial each over the
// 16-px-margin interior (src/frame.cpp:223-236).
//
// No LDS: a lane owns four adjacent output pixels of a 15-row strip; a wavefront covers a
// 64 x 60 tile (16 lanes across, 4 strips down), so VGA / EuRoC levels tile without remainder.
// Per source row a lane reads three dwords (bytes x-4 .. x+7, shared with its neighbours through
// L1), forms the horizontal derivative / smoothing sums of its four columns, keeps them in a
// five-row register window and emits one output row per input row: two 8-byte stores per lane =
// one full 128-byte line per 16 lanes (the previous LDS version stored 2 bytes per lane).  The
// strip loop is fully unrolled so the window is indexed statically.
// ALLFAST: every Sobel level's width is a multiple of 4 (chosen on the host): the pipelined strip above, 61 VGPRs.
template <bool ALLFAST>
__global__ __launch_bounds__(256) void k_sobel(PyrGeom g, uint8_t* const* bases)
{
  __shared__ double s_part[4][2];
  int b = blockIdx.x, level = 0;
  while (level < HSO_N_SOBEL_LEVELS - 1 && b >= g.sobel_blocks[level]) { b -= g.sobel_blocks[level]; level++; }
  const int W = g.w[level], H = g.h[level], GS = g.sob_stride[level];
  // pointers read from a table are generic; say "global" so the accesses are global_load / global_store
  // instead of FLAT (which also probes the LDS aperture and ties up both wait counters)
  typedef const __attribute__((address_space(1))) uint8_t* GlbCU8;
  typedef const __attribute__((address_space(1))) uint32_t* GlbCU32;
  typedef __attribute__((address_space(1))) int16_t* GlbI16;
  typedef __attribute__((address_space(1))) unsigned long long* GlbU64;
  uint8_t* base = bases[blockIdx.y];
  const GlbCU8 img = (GlbCU8)(base + g.off[level]);
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  const int tile = b * SOB_WAVES + wv;  // tiles of a level in row-major order, SOB_WAVES per block
  const int tiles_x = g.sobel_bx[level];
  const int tx = tile % tiles_x, ty = tile / tiles_x;
  const int x = tx * SOB_TW + (lane & 15) * 4;
  const int ys = ty * SOB_TH + (lane >> 4) * SOB_STRIP;
  const GlbI16 gx = (GlbI16)(base + g.sob_off[level][0]);
  const GlbI16 gy = (GlbI16)(base + g.sob_off[level][1]);
  unsigned isum = 0;
  double gsum = 0;
  if (ALLFAST) {
    if (x < W && ys < H) sobel_strip<1>((SobGlbCU8)img, (SobGlbI16)gx, (SobGlbI16)gy, W, H, GS, level, x, ys, isum, gsum);
  } else
  if (x < W && ys < H) {
    // widths that are multiples of 4 (every halfSample pyramid): aligned dwords for every lane; a
    // lane on the left / right image border replicates its first / last pixel instead of loading
    const bool fast = (W & 3) == 0;
    const bool has_left = x >= 4, has_right = x + 8 <= W;
    // Packed 16-bit arithmetic: every intermediate fits int16 (|hd| <= 6*255, hs <= 16*255,
    // |gx|, |gy| <= 16*6*255 = 24480), so two pixels ride in one register (v_pk_*_i16) and the
    // output dwords come out already packed.  P[j] = {p[j], p[j+1]} for the columns x-2+j.
    typedef short s16x2 __attribute__((ext_vector_type(2)));
    s16x2 hd[5][2], hs[5][2];
    uint32_t cdw[5];     // bytes x .. x+3 of the row (the four centre pixels), for the intensity sum
    const s16x2 k2 = {2, 2}, k4 = {4, 4}, k6 = {6, 6};
#pragma unroll
    for (int i = 0; i < SOB_STRIP + 4; i++) {
      int yy = ys + i - 2;
      yy = yy < 0 ? 0 : (yy > H - 1 ? H - 1 : yy);
      const GlbCU8 row = img + (size_t)yy * W;
      uint32_t d0, d1, d2;   // bytes x-4 .. x+7 (only x-2 .. x+5 are used)
      if (fast) {
        const GlbCU32 r32 = (GlbCU32)(row + x);
        d1 = r32[0];
        d0 = has_left ? r32[-1] : (d1 & 0xffu) * 0x01010101u;        // BORDER_REPLICATE
        d2 = has_right ? r32[1] : (d1 >> 24) * 0x01010101u;
      } else {
        uint32_t p[8];
#pragma unroll
        for (int k = 0; k < 8; k++) {
          int xx = x - 2 + k;
          xx = xx < 0 ? 0 : (xx > W - 1 ? W - 1 : xx);
          p[k] = row[xx];
        }
        d0 = (p[0] << 16) | (p[1] << 24);
        d1 = p[2] | (p[3] << 8) | (p[4] << 16) | (p[5] << 24);
        d2 = p[6] | (p[7] << 8);
      }
      // v_perm_b32: bytes 0-3 of the selector space = second operand, 4-7 = first, 0x0c = zero
      union { uint32_t u; s16x2 v; } P0, P1, P2, P3, P4, P5, P6;
      P0.u = __builtin_amdgcn_perm(d0, d0, 0x0c030c02u);
      P1.u = __builtin_amdgcn_perm(d1, d0, 0x0c040c03u);
      P2.u = __builtin_amdgcn_perm(d1, d1, 0x0c010c00u);
      P3.u = __builtin_amdgcn_perm(d1, d1, 0x0c020c01u);
      P4.u = __builtin_amdgcn_perm(d1, d1, 0x0c030c02u);
      P5.u = __builtin_amdgcn_perm(d2, d1, 0x0c040c03u);
      P6.u = __builtin_amdgcn_perm(d2, d2, 0x0c010c00u);
      const int sl = i % 5;
      hd[sl][0] = (P4.v - P0.v) + k2 * (P3.v - P1.v);
      hd[sl][1] = (P6.v - P2.v) + k2 * (P5.v - P3.v);
      hs[sl][0] = (P0.v + P4.v) + k4 * (P1.v + P3.v) + k6 * P2.v;
      hs[sl][1] = (P2.v + P6.v) + k4 * (P3.v + P5.v) + k6 * P4.v;
      cdw[sl] = d1;
      if (i >= 4) {
        const int y = ys + i - 4;
        const int ra = (i - 4) % 5, rb = (i - 3) % 5, rc = (i - 2) % 5, rd = (i - 1) % 5, re = i % 5;
        if (y < H) {
          union { uint32_t u; s16x2 v; } sx0, sx1, sy0, sy1;
          sx0.v = (hd[ra][0] + hd[re][0]) + k4 * (hd[rb][0] + hd[rd][0]) + k6 * hd[rc][0];
          sx1.v = (hd[ra][1] + hd[re][1]) + k4 * (hd[rb][1] + hd[rd][1]) + k6 * hd[rc][1];
          sy0.v = (hs[re][0] - hs[ra][0]) + k2 * (hs[rd][0] - hs[rb][0]);
          sy1.v = (hs[re][1] - hs[ra][1]) + k2 * (hs[rd][1] - hs[rb][1]);
          typedef unsigned long long u64;
          if (x + 4 <= W && ((W & 3) == 0)) {
            __builtin_nontemporal_store(((u64)sx1.u << 32) | sx0.u, (GlbU64)(gx + (size_t)y * GS + x));
            __builtin_nontemporal_store(((u64)sy1.u << 32) | sy0.u, (GlbU64)(gy + (size_t)y * GS + x));
          } else {  // widths that are not multiples of 4 (cv::resize pyramids): element-wise, last lane clipped
            const short sxv[4] = {sx0.v.x, sx0.v.y, sx1.v.x, sx1.v.y}, syv[4] = {sy0.v.x, sy0.v.y, sy1.v.x, sy1.v.y};
#pragma unroll
            for (int k = 0; k < 4; k++)
              if (x + k < W) { gx[(size_t)y * GS + x + k] = sxv[k]; gy[(size_t)y * GS + x + k] = syv[k]; }
          }
          if (level == 0 && x >= 16 && x < W - 16 && y >= 16 && y < H - 16) {
            // |grad| of the four pixels.  gx^2 + gy^2 <= 2 * 24480^2 < 2^31 is formed exactly by one
            // v_dot2_i32_i16 per pixel on the {gx, gy} pair (v_perm_b32 pairs them up), converted once
            // and rooted by the hardware v_sqrt_f32 (1 ulp).  The reference (src/frame.cpp:231) rounds
            // both squares and their sum to fp32 and adds 3e5 roots into ONE fp32 accumulator; the
            // difference of this form to a correctly rounded root is < 1e-7 of a term, four orders
            // below what the reference's own running fp32 sum loses (tests: means to 2e-5).
            union { uint32_t u; s16x2 v; } q0, q1, q2, q3;
            q0.u = __builtin_amdgcn_perm(sy0.u, sx0.u, 0x05040100u);
            q1.u = __builtin_amdgcn_perm(sy0.u, sx0.u, 0x07060302u);
            q2.u = __builtin_amdgcn_perm(sy1.u, sx1.u, 0x05040100u);
            q3.u = __builtin_amdgcn_perm(sy1.u, sx1.u, 0x07060302u);
            float mag[4];
            mag[0] = __builtin_amdgcn_sqrtf((float)__builtin_amdgcn_sdot2(q0.v, q0.v, 0, false));
            mag[1] = __builtin_amdgcn_sqrtf((float)__builtin_amdgcn_sdot2(q1.v, q1.v, 0, false));
            mag[2] = __builtin_amdgcn_sqrtf((float)__builtin_amdgcn_sdot2(q2.v, q2.v, 0, false));
            mag[3] = __builtin_amdgcn_sqrtf((float)__builtin_amdgcn_sdot2(q3.v, q3.v, 0, false));
            gsum += (double)((mag[0] + mag[1]) + (mag[2] + mag[3]));   // four values < 3.5e4 each: fp32 pair sums, then fp64
            isum = __builtin_amdgcn_sad_u8(cdw[rc], 0u, isum);   // + the four centre bytes
          }
        }
      }
    }
  }
  if (level == 0) {
    double is = wave_sum_to_lane63((double)isum);
    double gs = wave_sum_to_lane63(gsum);
    if (lane == 63) { s_part[wv][0] = is; s_part[wv][1] = gs; }
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

#ifndef HSO_FRAME_CHUNK_MB
#define HSO_FRAME_CHUNK_MB 96   // pyramid levels a chunk of frames leaves for its Sobel pass, in MB (0: one launch pair for the whole batch)
#endif
int hso_frame_build(hso_gpu_ctx* ctx, const PyrGeom& g, uint8_t* const* d_bases, const uint8_t* const* d_srcs,
                    hso_frame_stats* d_stats, int n)
{
  // Large batches are built in chunks: the Sobel pass re-reads levels 0..2 (1.31 W H bytes per frame) that the pyramid pass has
  // just written; with both passes over a few hundred frames at a time those bytes come out of the 256 MB Infinity Cache instead
  // of HBM (4096 EuRoC frames leave 1.9 GB of levels: nothing of it survives to the Sobel pass of a whole-batch launch).  This
  // is what a fused pyramid + Sobel kernel would save, without its halo bookkeeping (profiles/r6_fused_frame.md).
  const bool half = (g.w[0] % 16) == 0 && (g.h[0] % 16) == 0;
  const size_t per_frame = (size_t)g.pyr_bytes;
  int chunk = n;
  if (HSO_FRAME_CHUNK_MB > 0 && half && (size_t)n * per_frame > ((size_t)2 * HSO_FRAME_CHUNK_MB << 20))
    chunk = (int)std::max<size_t>(64, ((size_t)HSO_FRAME_CHUNK_MB << 20) / per_frame);
  const int sob_total = g.sobel_blocks[0] + g.sobel_blocks[1] + g.sobel_blocks[2];
  bool all_fast = true;
  for (int l = 0; l < HSO_N_SOBEL_LEVELS; l++) all_fast = all_fast && (g.w[l] & 3) == 0;
  for (int c0 = 0; c0 < n; c0 += chunk) {
    const int m = std::min(chunk, n - c0);
    uint8_t* const* bases = d_bases + c0;
    const uint8_t* const* srcs = d_srcs ? d_srcs + c0 : nullptr;
    if (half) {  // halfSample pyramid, src/frame.cpp:302-305
      const int cells = (g.w[0] >> 4) * (g.h[0] >> 4);
      dim3 gp((cells + 255) / 256, m);
      hipLaunchKernelGGL(k_pyramid, gp, dim3(256), 0, ctx->stream, g, bases, srcs);
    } else {                                         // cv::resize pyramid, :307-312
      if (srcs) hipLaunchKernelGGL(k_copy_level0, dim3((g.w[0] * g.h[0] + 255) / 256, m), dim3(256), 0, ctx->stream, g, bases, srcs);
      for (int l = 1; l < HSO_N_PYR_LEVELS; l++)
        hipLaunchKernelGGL(k_resize_level, dim3((g.w[l] * g.h[l] + 255) / 256, m), dim3(256), 0, ctx->stream, g, bases, l);
    }
    if (all_fast) hipLaunchKernelGGL(k_sobel<true>, dim3(sob_total, m), dim3(256), 0, ctx->stream, g, bases);
    else hipLaunchKernelGGL(k_sobel<false>, dim3(sob_total, m), dim3(256), 0, ctx->stream, g, bases);
  }
  hipLaunchKernelGGL(k_frame_stats, dim3(n), dim3(64), 0, ctx->stream, g, d_bases, d_stats);
  HSO_HIP_CHECK(ctx, hipGetLastError());
  return HSO_OK;
}

PyrGeom make_geom(int w, int h)
{
  PyrGeom g{};
  uint32_t off = 0;
  for (int l = 0; l < HSO_N_PYR_LEVELS; l++) {
    if ((w % 16) == 0 && (h % 16) == 0) {
      g.w[l] = w >> l; g.h[l] = h >> l;
    } else {  // cv::Size(cvRound((float)cols * scale), cvRound((float)rows * scale)), src/frame.cpp:309-310
      const float scale = 1.0 / (1 << l);
      g.w[l] = (int)lrint((double)((float)w * scale));
      g.h[l] = (int)lrint((double)((float)h * scale));
    }
    g.off[l] = off;
    const uint32_t bytes = (uint32_t)g.w[l] * g.h[l] + (uint32_t)g.w[l] + 64;  // + one zero row + slack
    off += (bytes + 255u) & ~255u;
  }
  g.pyr_bytes = off;
  for (int l = 0; l < HSO_N_SOBEL_LEVELS; l++) {
    g.sobel_bx[l] = (g.w[l] + SOB_TW - 1) / SOB_TW;  // tiles per row
    g.sobel_blocks[l] = (g.sobel_bx[l] * ((g.h[l] + SOB_TH - 1) / SOB_TH) + SOB_WAVES - 1) / SOB_WAVES;
    g.sob_stride[l] = (g.w[l] + 63) & ~63;
    for (int k = 0; k < 2; k++) {
      g.sob_off[l][k] = off;
      off += ((uint32_t)g.sob_stride[l] * g.h[l] * 2u + 255u) & ~255u;
    }
  }
  g.part_off = off;
  off += ((uint32_t)g.sobel_blocks[0] * 16u + 255u) & ~255u;
  g.stats_off = off;
  off += 256;
  g.frame_bytes = off;
  return g;
}
