// hso_align.hip — batched reprojection matching on gfx950: affine warp of the reference
// patch, 8x8 inverse-compositional Lucas-Kanade (2-D, or 1-D along an edgelet direction),
// NCC and edgelet-normal gates.
//
// Replaces Matcher::findMatchDirect after the reference observation has been chosen
// (reference src/matcher.cpp:286-375) and what it calls: warp::getWarpMatrixAffine :46-72,
// getBestSearchLevel :74-85, warpAffine(float) :120-155, createPatchFromPatchWithBorder
// :226-238, feature_alignment::align2D(float) src/feature_alignment.cpp:464-605,
// align1D(float) :164-308, Matcher::checkNCC :379-404, checkNormal :406-440,
// hso::interpolateMat_8u include/hso/vikit/vision.h:49-65.
//
// MI355X mapping: one wavefront per candidate, lane = pixel of the 8x8 patch (64 lanes = 64
// pixels, the reason this path is a natural fit for wave64).  The per-candidate geometry
// (fp64 warp matrix, search level) is computed redundantly by all lanes — it is uniform, so
// there is no divergence and no broadcast; the 10x10 warped patch lives in LDS; every LK
// iteration is one 4-tap bilinear fetch per lane + xor-butterfly sums (all lanes end with
// identical bits, so the iteration state stays uniform).  The reference accumulates the same
// sums serially in fp32; the butterfly order differs from it by rounding only (stated
// tolerance: 1e-3 px on the result).  HBM-bound in principle (100 + <=640
// taps per candidate); in practice latency-bound per candidate and throughput comes from
// having thousands of candidates in flight.
#include "hso_match_dev.h"
#include <vector>

using namespace hso_dev;

#define ALIGN_WAVES_PER_BLOCK 4

struct AlignConsts {
  hso_camera cam;
  PyrGeom g;
};

struct AlignJobDev {
  const uint8_t* ref_base;
  const uint8_t* cur_base;   // the frame this candidate is searched in (jobs of many frames share a launch)
  hso_align_job j;
};

__global__ __launch_bounds__(64 * ALIGN_WAVES_PER_BLOCK) void k_align(AlignConsts C, const AlignJobDev* jobs, int n_jobs,
                                                                       hso_align_out* outs)
{
  __shared__ float s_pwb[ALIGN_WAVES_PER_BLOCK][100];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int jid = blockIdx.x * ALIGN_WAVES_PER_BLOCK + wave;
  if (jid >= n_jobs) return;
  const AlignJobDev& JD = jobs[jid];
  const hso_align_out o = match_one(C.cam, C.g, JD.cur_base, JD.ref_base, JD.j, (double)0.7f, s_pwb[wave]);  // checkNCC(…, 0.7), :364
  if (lane == 0) outs[jid] = o;
}

// cur_frame_ids: one id per job (stride 1) or one id for all jobs (stride 0)
static int align_run(hso_gpu_ctx* ctx, const hso_camera* cam, const int64_t* cur_frame_ids, int id_stride, const hso_align_job* jobs,
                     int n_jobs, hso_align_out* out)
{
  if (!ctx) return HSO_E_INVALID;
  if (!cam || !cur_frame_ids || n_jobs < 0 || (n_jobs > 0 && (!jobs || !out))) return hso_fail(ctx, HSO_E_INVALID, "align_batch: bad argument");
  if (n_jobs == 0) return HSO_OK;
  HSO_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  PyrGeom g{};
  std::vector<AlignJobDev> h(n_jobs);
  int64_t last_id = 0;
  const uint8_t* last_base = nullptr;
  for (int i = 0; i < n_jobs; i++) {
    const int64_t cid = cur_frame_ids[(size_t)i * id_stride];
    if (i == 0 || cid != last_id) {
      auto itc = ctx->frames.find(cid);
      if (itc == ctx->frames.end()) return hso_fail(ctx, HSO_E_NOFRAME, "align_batch: current frame not resident");
      if (i == 0) g = itc->second.g;
      else if (itc->second.g.frame_bytes != g.frame_bytes) return hso_fail(ctx, HSO_E_INVALID, "align_batch: frames must share one size");
      last_id = cid; last_base = itc->second.base;
    }
    auto itr = ctx->frames.find(jobs[i].ref_frame_id);
    if (itr == ctx->frames.end()) return hso_fail(ctx, HSO_E_NOFRAME, "align_batch: reference frame not resident");
    if (itr->second.g.frame_bytes != g.frame_bytes) return hso_fail(ctx, HSO_E_INVALID, "align_batch: frames must share one size");
    if (jobs[i].ref_level < 0 || jobs[i].ref_level >= HSO_N_PYR_LEVELS) return hso_fail(ctx, HSO_E_INVALID, "align_batch: bad ref_level");
    h[i].ref_base = itr->second.base;
    h[i].cur_base = last_base;
    h[i].j = jobs[i];
  }
  if (cam->width != g.w[0] || cam->height != g.h[0]) return hso_fail(ctx, HSO_E_INVALID, "align_batch: camera size differs from the frame size");
  const size_t b_jobs = ((size_t)n_jobs * sizeof(AlignJobDev) + 255) & ~size_t(255);
  const size_t need = b_jobs + (size_t)n_jobs * sizeof(hso_align_out);
  if (ctx->batch_cap < need) {
    HSO_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->d_batch) (void)hipFree(ctx->d_batch);
    ctx->d_batch = nullptr; ctx->batch_cap = 0;
    HSO_HIP_CHECK(ctx, hipMalloc(reinterpret_cast<void**>(&ctx->d_batch), need));
    ctx->batch_cap = need;
  }
  AlignJobDev* d_jobs = reinterpret_cast<AlignJobDev*>(ctx->d_batch);
  hso_align_out* d_out = reinterpret_cast<hso_align_out*>(ctx->d_batch + b_jobs);
  HSO_HIP_CHECK(ctx, hipMemcpyAsync(d_jobs, h.data(), (size_t)n_jobs * sizeof(AlignJobDev), hipMemcpyHostToDevice, ctx->stream));
  AlignConsts C;
  C.cam = *cam; C.g = g;
  const int blocks = (n_jobs + ALIGN_WAVES_PER_BLOCK - 1) / ALIGN_WAVES_PER_BLOCK;
  hipLaunchKernelGGL(k_align, dim3(blocks), dim3(64 * ALIGN_WAVES_PER_BLOCK), 0, ctx->stream, C, d_jobs, n_jobs, d_out);
  HSO_HIP_CHECK(ctx, hipGetLastError());
  HSO_HIP_CHECK(ctx, hipMemcpyAsync(out, d_out, (size_t)n_jobs * sizeof(hso_align_out), hipMemcpyDeviceToHost, ctx->stream));
  HSO_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  return HSO_OK;
}

extern "C" int hso_gpu_align_batch(hso_gpu_ctx* ctx, const hso_camera* cam, int64_t cur_frame_id, const hso_align_job* jobs,
                                   int n_jobs, hso_align_out* out)
{
  return align_run(ctx, cam, &cur_frame_id, 0, jobs, n_jobs, out);
}

extern "C" int hso_gpu_align_multi(hso_gpu_ctx* ctx, const hso_camera* cam, const int64_t* cur_frame_ids, const hso_align_job* jobs,
                                   int n_jobs, hso_align_out* out)
{
  return align_run(ctx, cam, cur_frame_ids, 1, jobs, n_jobs, out);
}
