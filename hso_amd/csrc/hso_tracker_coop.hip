// hso_tracker_coop.hip — the cooperative shape of the tracker: ONE (ref, cur) pair split across K workgroups.
//
// CoarseTracker::run is called once per frame on one sequence (reference src/frame_handler_mono.cpp:190-204); BASELINE
// configs[2] / [3] are a single sequence on a single MI355X.  The batch shapes of hso_tracker.hip give one job to one
// workgroup = one CU of 256; with 2000 features a job then takes ~1.5 ms, most of it the 30-odd evaluations walking 2000
// features on 512 threads.  Here K workgroups (K = 2..32, the CUs of one XCD) own feature slices of the same pair and run the
// same level / Levenberg-Marquardt loop in lockstep:
//   * every workgroup stages the level images into its own LDS and keeps its slice's patch cache, keys and visibility flags;
//   * one evaluation = the slice's residuals + normal-equation sums, then ONE exchange: each workgroup publishes its 38
//     partial sums as 8-byte {tag, value} granules and reads everybody's; all add them in rank order, so all K hold the same
//     bits and take the same LM step, accept decision and stop decision redundantly — no leader, no second exchange;
//   * transport: placement-independent by construction (agent-scope granules / atomics, Guideline 16).  At start the
//     workgroups of a job exchange their XCC ids through that safe transport; only if all K sit on ONE XCD — whose L2 is then
//     their common point of coherence — they switch to plain stores and L2-scope atomics read back with L1-bypassing loads
//     (the same words, the same arithmetic, ~3x shorter exchange); any other placement keeps the agent-scope forms;
//   * the robust thresholds (exact median / MAD over ALL features' |residual| keys) merge the workgroups' leading-digit
//     histograms by device-scope atomics, gather the keys of the winning bin, and finish on every workgroup alike.  The order
//     statistic is exact, so thresholds, visibility and term counts equal the one-workgroup path's bit for bit; the sums
//     H, b, E differ from it by rounding only (a different summation tree), like trk1 and trk2 differ from each other.
// The device code is hso_tracker_core.h compiled with TRK_COOP (hooks at the three places above); its own translation unit so
// that the batch shapes' register allocation is untouched and the two compile in parallel.
#include "hso_ctx.h"
#include <stdlib.h>
#include "hso_dev_math.h"
#include <string.h>
#include <algorithm>

using namespace hso_dev;

#include "hso_tracker_defs.h"

#define TRK_COOP 1
#define TRK_THREADS 512
#define TRK_LDS_KB 160
#define TRK_WAVES_PER_EU 2
#define TRK_OLD_SHARE 8     // at most one feature per thread: an even split keeps the waves short
namespace trkc {
#include "hso_tracker_core.h"
}

size_t hso_track_coop_lds_bytes() { return (size_t)trkc::kImgCap + sizeof(trkc::Shared); }
int hso_track_coop_img_cap() { return trkc::kImgCap; }

hipError_t hso_track_coop_launch(hipStream_t stream, const TrackConsts& C, const TrackJobDev* subjobs, int n_jobs, int k_stride,
                                 int scatter, CoopJobState* state, unsigned* fail_flag, char* scratch, size_t scratch_stride,
                                 hso_track_result* results)
{
  static bool attr_set = false;
  const size_t lds = hso_track_coop_lds_bytes();
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(trkc::k_track_coop<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(trkc::k_track_coop<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    attr_set = true;
  }
  // block b runs on XCD b % 8 (observed; relied on for speed only): job j = b % 8, rank = b / 8 keeps a job on one XCD
  const dim3 grid(scatter ? n_jobs * k_stride : 8 * k_stride), block(TRK_THREADS);
  if (C.inverse)
    hipLaunchKernelGGL(trkc::k_track_coop<true>, grid, block, lds, stream, C, subjobs, n_jobs, k_stride, scatter, state, fail_flag, scratch, scratch_stride, results);
  else
    hipLaunchKernelGGL(trkc::k_track_coop<false>, grid, block, lds, stream, C, subjobs, n_jobs, k_stride, scatter, state, fail_flag, scratch, scratch_stride, results);
  return hipGetLastError();
}
