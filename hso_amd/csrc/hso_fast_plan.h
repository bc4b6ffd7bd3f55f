// hso_fast_plan.h — the device-side layout the FAST stage leaves in ctx->d_batch, shared with the
// stage chained behind it (hso_edgelet.hip) so the corner masks never leave the GPU.
#pragma once
#include <vector>
#include "hso_ctx.h"

#define FAST_TW 64
#define FAST_TH 16

struct FastPlan {
  PyrGeom g;                 // geometry shared by every frame of the batch
  int n_frames, n_levels, cap;
  char* d;                   // ctx->d_batch
  size_t cnt_bytes;          // the row counters at the head of a slice
  size_t per_frame;          // one slice per frame: [row counts | per level: mask, row offsets, corners]
  size_t o_cnt[HSO_N_PYR_LEVELS], o_mask[HSO_N_PYR_LEVELS], o_off[HSO_N_PYR_LEVELS], o_out[HSO_N_PYR_LEVELS];
  int wpr[HSO_N_PYR_LEVELS]; // 64-bit mask words per image row
  size_t o_tab;              // frame base pointers (device table, const uint8_t* [n_frames])
  size_t o_tot;              // corner totals [n_frames][n_levels]
  size_t o_thr;              // per-frame barriers, int [3][n_frames]: FAST threshold, Canny low, Canny high (when a caller gives them)
  size_t o_extra;            // start of the caller's `extra` bytes
};

// Validates the frames, sizes ctx->d_batch (FAST work area + `extra` bytes for the caller) and
// enqueues mask / scan / emit for every level on ctx->stream; no synchronisation.
// per_frame3 != nullptr: int [3][n_frames] — the FAST threshold of every frame (row 0; `threshold` is ignored) and two more rows the
// caller's own kernels read from plan->d + plan->o_thr (the edgelet detector's Canny thresholds)
int hso_fast_enqueue(hso_gpu_ctx* ctx, const int64_t* frame_ids, int n_frames, int n_levels, int threshold, int border, int cap,
                     size_t extra, FastPlan* plan, const int32_t* per_frame3 = nullptr);
// The two halves of hso_fast_enqueue for callers that place a second FAST pass (FAST-12 of
// fillingHole) in their own region: the layout (offsets relative to plan->d; returns the bytes it
// needs), and the launches for arc = 9 or 12 over a device table of frame base pointers.
size_t hso_fast_plan(const PyrGeom& g, int n_frames, int n_levels, int cap, FastPlan* plan);
int hso_fast_launch(hso_gpu_ctx* ctx, const FastPlan& plan, const uint8_t* const* d_bases, int threshold, int border, int arc,
                    const int* d_thr = nullptr);   // d_thr: the threshold of every frame (device, [n_frames]) instead of `threshold`
// counts (+ corners when cap > 0) to the host; synchronises the stream.
// more != nullptr: the corner lists are appended to *more instead of being read back; the caller adds its own lists and runs
// hso_lists_to_host once (one DMA for everything the call returns)
int hso_fast_collect(hso_gpu_ctx* ctx, const FastPlan& plan, hso_corner* out, int32_t* counts, std::vector<HsoListCopy>* more = nullptr);
