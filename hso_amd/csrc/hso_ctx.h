// hso_ctx.h — context, resident frames and shared launch helpers behind the C-ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <algorithm>
#include <string>
#include <type_traits>
#include <unordered_map>
#include <functional>
#include <thread>
#include <vector>
#include "../../include/hso_gpu.h"
#include "../../include/hso_gpu_debug.h"

// Geometry of one resident frame: 5-level u8 pyramid + Sobel images of levels 0-2.
// Every level starts 256-byte aligned and is followed by at least one zeroed
// row + 64 bytes: the reference's dy taps (src/CoarseTracker.cpp:370,491) read
// row `rows` of a level for features on the bottom border (a heap over-read in
// the reference, value undefined there); here those bytes are defined as 0.
struct PyrGeom {
  int w[HSO_N_PYR_LEVELS], h[HSO_N_PYR_LEVELS];
  uint32_t off[HSO_N_PYR_LEVELS];     // byte offset of each level in the pyramid block
  uint32_t pyr_bytes;                 // total bytes of the pyramid block (multiple of 256)
  uint32_t sob_off[HSO_N_SOBEL_LEVELS][2];  // byte offsets (from the frame base) of gx, gy
  int sob_stride[HSO_N_SOBEL_LEVELS];       // row stride of the gradient images in pixels: the width rounded up to 64, so every
                                            // row starts on a 128-byte line (752-wide frames: 36 % faster stores than stride 752)
  uint32_t part_off;                  // stats partials (double2 per level-0 Sobel block)
  uint32_t stats_off;                 // hso_frame_stats
  uint32_t frame_bytes;
  int sobel_blocks[HSO_N_SOBEL_LEVELS]; // blocks per level
  int sobel_bx[HSO_N_SOBEL_LEVELS];     // blocks in x per level
};

PyrGeom make_geom(int w, int h);
// two frames share kernels' index arithmetic only when their level-0 sizes agree (equal byte totals are not enough:
// transposed sizes have them too)
inline bool same_geom(const PyrGeom& a, const PyrGeom& b) { return a.w[0] == b.w[0] && a.h[0] == b.h[0]; }

struct FrameRec {
  int64_t id;
  PyrGeom g;
  uint8_t* base;  // device
};

struct TrackBatchState;  // hso_tracker.hip
struct SeedTables;       // hso_seed.hip: resident seed tables
struct SeqMaps;          // hso_align.hip: sequence maps (tables mirrored row for row, patched in place)

struct hso_gpu_ctx {
  int device;
  hipStream_t stream;
  bool own_stream;
  int n_cu;
  hso_parallel_for_fn par_fn = nullptr; void* par_user = nullptr;   // hso_gpu_set_host_parallel: the caller's own worker pool
  bool shared_device = false;   // hso_gpu_set_shared_device: other contexts keep the device busy beside this one
  hso_gpu_options opt{};        // hso_gpu_configure
  std::string err;
  std::unordered_map<int64_t, FrameRec> frames;
  // recycled frame allocations, one free list per geometry (key = width << 32 | height; their padding rows are still zero)
  std::unordered_map<uint64_t, std::vector<uint8_t*>> free_frames;
  std::vector<uint8_t*> frame_slabs;  // what hipMalloc returned: slabs of frames (hso_ctx.hip: hso_frame_alloc), freed with the context
  TrackBatchState* track;
  SeedTables* seed_tables;
  SeqMaps* seqmaps;
  // staging for batched frame uploads: [bases | srcs | stats]
  char* d_batch; size_t batch_cap;
  // hso_gpu_seq_local_ba: the window assembly's per-job tables (they outlive the batch's own area: the write-back reads them), and what
  // hso_gpu_seq_ba_debug_window serves
  char* d_rba = nullptr; size_t rba_cap = 0; struct RbaLast* rba_last = nullptr;
  // between the three kernels of a seed observation (hso_seed.hip): [SeedPre | SeedMid] per seed, grow-only
  char* d_seed_scratch; size_t seed_scratch_cap;
  // The depth filter's own stream (the reference runs it on its own thread, src/depth_filter.cpp:130-162): the previous-frame pass
  // of a seed table can be started on it and collected later (hso_gpu_seed_table_observe_previous_begin / _end), overlapping the
  // tracker's work on `stream`.  While a pass is in flight, every entry point that touches seed tables or releases a frame waits
  // for it first (hso_seed_async_quiesce).
  hipStream_t seed_stream; hipEvent_t seed_go, seed_done;
  bool seed_inflight; int seed_inflight_table; size_t seed_inflight_n;
  char* d_seed_scratch_async; size_t seed_scratch_async_cap;
  char* h_seed_pin; size_t h_seed_pin_cap, h_seed_brief_off;
  // pinned host staging (grow-only): record tables go through it so the DMA runs at PCIe rate
  // instead of the pageable-memory rate, and the per-call std::vector + page faults disappear
  char* h_pin[2]; size_t h_pin_cap[2];
  std::vector<void*> host_allocs;   // hso_gpu_host_alloc
  // hso_lists_to_host: the packed image of many short device lists (device side, page-locked host side), grow-only
  char* d_pack; size_t d_pack_cap;
  char* h_pack; size_t h_pack_cap;
};

// Many short device lists (per frame and level: corners, edgelets ...) to caller memory in ONE DMA: a gather kernel packs them back
// to back, one copy brings the image to page-locked memory, the pieces are copied out from there.  One copy per list costs ~10 us
// of launch + DMA set-up each; a keyframe step of 24 sequences reads 240 lists.  Synchronises the context's stream.
struct HsoListCopy { void* dst; const void* src; size_t bytes; };   // bytes: a multiple of 4, src 4-byte aligned
int hso_lists_to_host(hso_gpu_ctx* ctx, const std::vector<HsoListCopy>& lists);

// ---- host memory and the runtime's copy calls -----------------------------------------------------------------------------
// The entry points accept any host pointer.  Handing PAGEABLE caller memory to hipMemcpyAsync is not harmless on this stack: the
// runtime registers such ranges with the kernel driver to DMA from / to them, and when the caller later returns the memory to the
// kernel (a large std::vector or numpy array is an mmap'd region; malloc arenas of threads are trimmed) the driver's MMU notifier EVICTS the process's GPU queues and
// restores them >= 10 ms later — the next launch, of whatever kernel, starts 10-35 ms late.  Measured in the multi-sequence driver
// at 32 sequences (tables of a few MB built and freed every step): 20-30 ms per step instead of 4.5 (hip trace + kernel trace:
// the launch returns in microseconds, the GPU idles, the kernel starts tens of milliseconds later; gone with
// MALLOC_MMAP_THRESHOLD_ raised so that free() never unmaps).  So every copy between host and device in this library goes through
// the wrappers below: page-locked host memory (hso_pinned, hso_gpu_host_alloc, anything hipHostMalloc'ed / registered) is passed
// through as it is; pageable memory is staged through page-locked chunks owned by the stream (host-to-device: copied into the
// chunk now, DMA from there; device-to-host: DMA into the chunk, copied out by the stream synchronisation that every entry point
// performs before it returns).  The runtime never sees a pageable pointer of the caller.
hipError_t hso_copy_async(void* dst, const void* src, size_t bytes, hipMemcpyKind kind, hipStream_t stream);
hipError_t hso_copy2d_async(void* dst, size_t dpitch, const void* src, size_t spitch, size_t width, size_t height, hipMemcpyKind kind,
                            hipStream_t stream);
hipError_t hso_copy_sync(void* dst, const void* src, size_t bytes, hipMemcpyKind kind);
hipError_t hso_memset_async(void* dst, int value, size_t bytes, hipStream_t stream);   // counted (hso_gpu_debug_census)
hipError_t hso_stream_sync(hipStream_t stream);
// room in the stream's page-locked staging chunks, the caller's until the stream's next synchronisation: a table assembled there
// goes to the device with one DMA and without the second copy hso_copy_async makes for pageable memory
char* hso_stage_reserve(hipStream_t stream, size_t bytes);
void hso_stream_forget(hipStream_t stream);   // context teardown: free the stream's staging chunks
void hso_stream_set_wait(hipStream_t stream, int mode);   // how waits on the stream wait: HSO_WAIT_POLL / _NAP / _BLOCK
void hso_stream_abandon(hipStream_t stream);  // error path: wait for the stream, then DROP the pending copies into caller memory
                                              // (the entry point returns an error; the caller may free its buffers at once)
#ifndef HSO_RAW_HIP_COPIES
#define hipMemcpyAsync(dst, src, bytes, kind, stream) hso_copy_async((dst), (src), (bytes), (kind), (stream))
#define hipMemcpy2DAsync(dst, dpitch, src, spitch, width, height, kind, stream) \
  hso_copy2d_async((dst), (dpitch), (src), (spitch), (width), (height), (kind), (stream))
#define hipMemcpy(dst, src, bytes, kind) hso_copy_sync((dst), (src), (bytes), (kind))
#define hipStreamSynchronize(stream) hso_stream_sync(stream)
#define hipMemsetAsync(dst, value, bytes, stream) hso_memset_async((dst), (value), (bytes), (stream))
#endif

#define HSO_HIP_CHECK(ctx, expr)                                              \
  do {                                                                        \
    hipError_t _e = (expr);                                                   \
    if (_e != hipSuccess) {                                                   \
      (ctx)->err = std::string(#expr) + ": " + hipGetErrorString(_e);         \
      hso_stream_abandon((ctx)->stream);                                      \
      return HSO_E_HIP;                                                       \
    }                                                                         \
  } while (0)

int hso_fail(hso_gpu_ctx* ctx, int code, const char* msg);

// The host side of a batched entry point — staging many megabytes of the caller's tables, checking their indices, sorting per
// window — runs on the calling thread while the device waits: 2.7 + 5.2 ms per keyframe step of 128 sequences in the local-BA calls,
// 1 ms per step in the map patches.  hso_host_parallel(ctx, n, work, fn) spreads such a loop over the worker pool the caller lent
// (hso_gpu_set_host_parallel: the sequence engine lends its own, whose workers are idle while the engine's thread is inside the
// library): fn(i) for i in [0, n); `work` = bytes (or comparable) the loop touches — below ~1 MB, or without a lent pool, the
// calling thread does it alone.  fn must not touch the HIP runtime or ctx->err.  (Round 5 also kept three helper threads per
// context behind an environment switch; they cost six engines 20-35 % and are gone.)
template <class F> void hso_host_parallel(hso_gpu_ctx* ctx, int n, size_t work, F&& fn)
{
  if (n >= 2 && work >= (size_t(1) << 20) && ctx->par_fn) {
    using Fn = typename std::remove_reference<F>::type;
    ctx->par_fn(ctx->par_user, n, [](void* a, int i) { (*static_cast<Fn*>(a))(i); }, const_cast<void*>(static_cast<const void*>(&fn)));
    return;
  }
  for (int i = 0; i < n; i++) fn(i);
}
// size of a grow-only buffer that must hold `need` bytes now: half again as much + 1 MB, so that tables that grow a little with
// every keyframe do not re-allocate (hipFree synchronises the device: 0.2-0.4 ms each, 18 per step at 32 sequences before this)
inline size_t hso_grown(size_t need) { return need + need / 2 + (size_t(1) << 20); }
// pinned staging buffer `slot` (0: inputs, 1: results) of at least `bytes`; nullptr if the allocation fails.
// Contents are only valid until the next call that uses the slot; every entry point synchronises before it returns.
char* hso_pinned(hso_gpu_ctx* ctx, int slot, size_t bytes);


// frame kernels (hso_frame.hip): build pyramid + Sobel + stats for `n` frames whose
// level-0 bytes are already in place at base+off[0].
int hso_frame_build(hso_gpu_ctx* ctx, const PyrGeom& g, uint8_t* const* d_bases, const uint8_t* const* d_srcs,
                    hso_frame_stats* d_stats, int n);
// cv::resize INTER_LINEAR of a device image into a device buffer (hso_frame.hip)
int hso_frame_resize_into(hso_gpu_ctx* ctx, const uint8_t* d_src, int sw, int sh, uint8_t* d_dst, int dw, int dh);
void hso_track_state_free(hso_gpu_ctx* ctx);
void hso_seed_tables_free(hso_gpu_ctx* ctx);
int hso_seed_table_rows(hso_gpu_ctx* ctx, int table, const int32_t* slots, int n, const char** rows, size_t* stride, size_t* seed_offset, PyrGeom* g);
int hso_seed_async_quiesce(hso_gpu_ctx* ctx);   // wait for a previous-frame pass in flight (its results stay collectable)
bool hso_seed_tables_pin(hso_gpu_ctx* ctx, int64_t frame_id);   // a resident seed table hosts live seeds in this frame
void hso_seqmaps_free(hso_gpu_ctx* ctx);
// a frame allocation of geometry g: recycled when the free list holds that geometry, else fresh with zeroed padding rows.
// hso_frame_free returns it to the list (or the allocator); neither touches ctx->frames.
// ---- the resident per-frame chain (hso_gpu_seq_chain: orchestrated in hso_select.hip; the sequence maps' storage and the kernels
// that walk it live in hso_align.hip, the tracker in hso_tracker.hip) ----
struct SeqKfDev {              // a keyframe row of a sequence map on the device
  hso_se3 T_f_w;
  double exposure_time;
  const uint8_t* base;         // the resident frame
  int64_t frame_id;
  int32_t keyframe_id;
  int32_t key_point[5];        // Frame::key_pts_ as point rows, -1: none
};
struct SeqMapDev {             // one sequence map as the chain's kernels see it
  hso_map_point* pts; const hso_obs* obs; const int32_t* obs_pt; const SeqKfDev* kfs;
  const int32_t* kf_fts;       // list of keyframe row r: kf_fts + r * fts_cap
  const int32_t* cands;
  int32_t* first;              // n_pts scratch integers, all 0x7fffffff between calls (the listing's "seen in this frame" stamps)
  const hso_seq_feature* ff_ref; hso_seq_feature* ff_cur;   // the reference frame's feature table (null: it is a keyframe) / the new frame's
  int n_pts, n_obs, n_kfs, fts_cap, n_cands, ff_cap;
};
struct ChainJobDev {
  SeqMapDev M;
  const uint8_t* cur_base;
  double* table;               // the tracker's SoA feature table of this job: [6][n_ref_stride]
  hso_se3 T_ref_w, T_cur_w0;
  double ref_exposure;
  int ref_kf_row, n_ref, n_ref_stride, flags;
  int cur_keyframe_id, last_kf_row;
  int covis[5];
  int slice_begin, slice_cap;  // the job's slice of the per-listed-point arrays
  int temps_begin, n_temps;
  int kf_begin;                // first row of the job's blocks in the per-keyframe arrays (ReprojKf rows, list lengths)
  int seed_group;              // the job's group in the call's seed table, -1 none
};
struct ChainCur {              // what the chain's kernels hand on about a job's new frame
  hso_se3 T_cur_w;             // after CoarseTracker::run
  double exposure;
  double cur_pos[3];           // the new frame's position in the world (Frame::pos())
  int n_listed, n_kf_points, n_cands_listed, n_visit;
  int visit[HSO_SEQ_MAX_VISIT];
};
struct ReprojKf;               // hso_align.hip
struct AlignJobDev;
// a map's device view for one chain job; flips the map's frame-feature tables (the previous new frame becomes the reference)
int hso_seqmap_flush_kfs(hso_gpu_ctx* ctx);   // the keyframe tables that changed since the last chain, in one copy + one launch
int hso_seqmap_chain_view(hso_gpu_ctx* ctx, const hso_seq_job& job, SeqMapDev* out, int* n_kfs, const int32_t** kf_nfts_host);
int hso_seqmap_ba_view(hso_gpu_ctx* ctx, int map, SeqMapDev* out, const hso_kf** kfs_host, const int32_t** kf_nfts_host);   // hso_gpu_seq_local_ba
void hso_seqmap_ba_set_pose(hso_gpu_ctx* ctx, int map, int row, const hso_se3& T);
int hso_seqmap_chain_reserve(hso_gpu_ctx* ctx, int map, int rows);        // room for `rows` features in the map's two frame tables (before any view)
void hso_seqmap_chain_commit(hso_gpu_ctx* ctx, const hso_seq_job& job, int n_feats);   // after a successful call: the new frame's table is the map's newest
// stage launchers (asynchronous on the context's stream); d_* are device pointers into the work area
struct ChainFront {
  const ChainJobDev* d_jobs; ChainCur* d_cur; const hso_track_result* d_track; const int32_t* d_kf_nfts; const int32_t* d_temps;
  ReprojKf* d_kfs; int32_t* d_ids; uint8_t* d_quality; AlignJobDev* d_align; hso_align_out* d_match; hso_reproj_point* d_proj;
  hso_match_brief* d_brief; struct PoseJobDev* d_pose_jobs;
  int n_jobs, n_total, max_kfs, cell_size, grid_n_cols;
};
int hso_chain_table_launch(hso_gpu_ctx* ctx, const ChainJobDev* d_jobs, int n_jobs, int n_max_stride);
int hso_chain_front_launch(hso_gpu_ctx* ctx, const hso_camera* cam, const ChainFront& F);
size_t hso_chain_sizeof_reproj_kf();
size_t hso_chain_sizeof_align_job();
PyrGeom hso_seqmaps_geom(hso_gpu_ctx* ctx, bool* have);
void hso_rba_forget(hso_gpu_ctx* ctx);     // hso_ba.hip: the same for hso_gpu_seq_local_ba
void hso_chain_forget(hso_gpu_ctx* ctx);   // hso_select.hip: drop what the last chain call of a context left (context teardown)
// hso_seed.hip: the frame a group of a resident seed table is observed in (cur_base == null: the group sits the observation out)
struct SeedFrameDev {
  hso_se3 T_f_w;
  double exposure;
  const uint8_t* cur_base;   // resident tables: the seed's own cur_base is null and the frame's is used
};
// The chain's tail observes the seeds of the sequences whose frame turned out a regular one (no keyframe, enough inliers):
// _frames gives the table's per-group frame records on the device, all set to "sits out" (asynchronous memset), for the chain's last
// kernel to fill in; _launch then queues one observation of every live seed (hso_gpu_seed_table_observe_groups' kernels) and
// returns where the briefs will be (device) and how many slots the table has.  Nothing here waits for the device.
int hso_seed_table_chain_frames(hso_gpu_ctx* ctx, int table, int n_groups, SeedFrameDev** d_frames);
int hso_seed_table_chain_launch(hso_gpu_ctx* ctx, const hso_camera* cam, int table, int n_groups, double px_error_angle, const hso_seed_brief** d_brief, int* n_slots);
// hso_tracker.hip: the tracker over device-built feature tables (hso_track_job.feats_soa == 2: `feats` is a DEVICE pointer to the
// kernel's layout); the result records stay on the device (*d_results).  Asynchronous for the batch shapes; a cooperative launch
// (a batch smaller than the chip) is waited for, because its time-out fallback has to be known before the chain goes on.
int hso_track_chain_launch(hso_gpu_ctx* ctx, const hso_camera* cam, const hso_track_params* p, const hso_track_job* jobs, int n_jobs,
                           const hso_track_result** d_results, bool* cooperative);
void hso_seqmaps_debug_set(hso_gpu_ctx* ctx, int what, const void* d, size_t bytes);
int hso_frame_alloc(hso_gpu_ctx* ctx, const PyrGeom& g, uint8_t** base);
void hso_frame_free(hso_gpu_ctx* ctx, const PyrGeom& g, uint8_t* base);
