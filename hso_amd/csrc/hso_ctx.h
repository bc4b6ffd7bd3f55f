// hso_ctx.h — context, resident frames and shared launch helpers behind the C-ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <unordered_map>
#include <vector>
#include "../../include/hso_gpu.h"

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
struct MapArena;         // hso_align.hip: resident map tables
struct SeqMaps;          // hso_align.hip: sequence maps (tables mirrored row for row, patched in place)

struct hso_gpu_ctx {
  int device;
  hipStream_t stream;
  bool own_stream;
  int n_cu;
  std::string err;
  std::unordered_map<int64_t, FrameRec> frames;
  // recycled frame allocations, one free list per geometry (key = width << 32 | height; their padding rows are still zero)
  std::unordered_map<uint64_t, std::vector<uint8_t*>> free_frames;
  std::vector<uint8_t*> frame_slabs;  // what hipMalloc returned: slabs of frames (hso_ctx.hip: hso_frame_alloc), freed with the context
  TrackBatchState* track;
  SeedTables* seed_tables;
  MapArena* maps;
  SeqMaps* seqmaps;
  // staging for batched frame uploads: [bases | srcs | stats]
  char* d_batch; size_t batch_cap;
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
void hso_stream_forget(hipStream_t stream);   // context teardown: free the stream's staging chunks
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
int hso_seed_async_quiesce(hso_gpu_ctx* ctx);   // wait for a previous-frame pass in flight (its results stay collectable)
bool hso_seed_tables_pin(hso_gpu_ctx* ctx, int64_t frame_id);   // a resident seed table hosts live seeds in this frame
void hso_map_arena_free(hso_gpu_ctx* ctx);
void hso_seqmaps_free(hso_gpu_ctx* ctx);
// a frame allocation of geometry g: recycled when the free list holds that geometry, else fresh with zeroed padding rows.
// hso_frame_free returns it to the list (or the allocator); neither touches ctx->frames.
// hso_align.hip: project + reference choice + findMatchDirect for every point of every call's stored map, results left on the
// device (begin[c] = first record of call c); extra_bytes of the work area are reserved behind them (hso_select.hip chains the
// grid selection there).  Returns the number of records or a status < 0.
struct HsoMapsRun {
  int n;
  std::vector<int> begin;
  const hso_reproj_point* d_proj;
  const hso_match_brief* d_brief;
  char* d_extra;
};
struct MapArenaSizes { long long total; };   // points of a set of calls
int hso_map_call_sizes(hso_gpu_ctx* ctx, const hso_map_call* calls, int n_calls, MapArenaSizes* Z);
int hso_reproject_maps_run(hso_gpu_ctx* ctx, const hso_camera* cam, const hso_map_call* calls, int n_calls, int cell_size,
                           int grid_n_cols, size_t extra_bytes, HsoMapsRun* R);
// the sequence-map form (hso_gpu_reproject_select_pose_frames): the same records for the points each frame LISTS; what the
// chained pose optimisation needs beside them comes back in X
struct HsoFramesAux {
  std::vector<int> kf_begin;                   // per frame: first row of its keyframes in kf_poses (n_frames + 1)
  std::vector<hso_se3> kf_poses;               // T_f_w of every frame's map keyframes, frames back to back
  std::vector<const hso_map_point*> pts;       // per frame: its map's point table (device)
  const int32_t* d_ids; const uint8_t* d_quality;   // the listed ids / quality keys, frames back to back (device)
  const hso_align_out* d_match;
};
int hso_reproject_frames_run(hso_gpu_ctx* ctx, const hso_camera* cam, const hso_map_frame* frames, int n_frames, int cell_size,
                             int grid_n_cols, size_t extra_bytes, HsoMapsRun* R, HsoFramesAux* X);
void hso_seqmaps_debug_set(hso_gpu_ctx* ctx, int what, const void* d, size_t bytes);
const hso_map_point* hso_map_points_dev(hso_gpu_ctx* ctx);
int hso_map_max_points(hso_gpu_ctx* ctx);
int hso_map_max_kfs(hso_gpu_ctx* ctx);
int hso_map_kf_poses(hso_gpu_ctx* ctx, int map, hso_se3* out);   // T_f_w of map `map`'s keyframes in table order; returns their number
int hso_frame_alloc(hso_gpu_ctx* ctx, const PyrGeom& g, uint8_t** base);
void hso_frame_free(hso_gpu_ctx* ctx, const PyrGeom& g, uint8_t* base);
