// hso_ctx.hip — context lifecycle and the frame entry points of include/hso_gpu.h.
#include <algorithm>
#define HSO_RAW_HIP_COPIES          // this file implements the copy wrappers: it alone calls the runtime's copy functions
#include "hso_ctx.h"
#include <string.h>
#include <atomic>
#include <chrono>
#include <time.h>
#include <sys/prctl.h>
#include <condition_variable>
#include <mutex>
#include <unordered_set>
#include <vector>

// ---- staging of pageable host memory (see hso_ctx.h) ----
namespace {
struct StageChunk { char* p; size_t cap, used; };
struct StageFix { void* dst; const char* src; size_t bytes, dpitch, width, height; };   // height 0: a flat copy of `bytes`
struct Stager {
  std::mutex m;                     // a stream's stager is its own: the banks sharing a device do not queue behind one another's copies
  std::vector<StageChunk> chunks;
  std::vector<StageFix> fixes;      // device-to-host copies to finish after the next synchronisation
  int wait_mode = HSO_WAIT_POLL;    // hso_stream_set_wait
  hipEvent_t wait_ev = nullptr;     // ... through this event (hipEventBlockingSync)
};
std::mutex g_stage_mutex;           // the table itself (element addresses are stable)
std::unordered_map<hipStream_t, Stager> g_stagers;
Stager& stager_of(hipStream_t stream) { std::lock_guard<std::mutex> lk(g_stage_mutex); return g_stagers[stream]; }
Stager* stager_find(hipStream_t stream)
{
  std::lock_guard<std::mutex> lk(g_stage_mutex);
  auto it = g_stagers.find(stream);
  return it == g_stagers.end() ? nullptr : &it->second;
}
std::unordered_set<hipStream_t> g_streams_in_use;   // caller-provided streams that a live context launches on

bool host_is_page_locked(const void* p)
{
  hipPointerAttribute_t at{};
  if (hipPointerGetAttributes(&at, p) != hipSuccess) { (void)hipGetLastError(); return false; }   // unknown to the runtime: ordinary memory
  return at.type == hipMemoryTypeHost || at.type == hipMemoryTypeDevice || at.type == hipMemoryTypeManaged || at.type == hipMemoryTypeArray;
}

char* stage_alloc(Stager& S, size_t bytes)
{
  const size_t need = (bytes + 255) & ~size_t(255);
  for (StageChunk& c : S.chunks)
    if (c.cap - c.used >= need) { char* r = c.p + c.used; c.used += need; return r; }
  StageChunk c{nullptr, std::max(need, size_t(8) << 20), 0};
  if (hipHostMalloc(reinterpret_cast<void**>(&c.p), c.cap, hipHostMallocDefault) != hipSuccess) return nullptr;
  c.used = need;
  S.chunks.push_back(c);
  return c.p;
}
}  // namespace

// developer census (hso_gpu_debug_census): what the library asked of the runtime since the process started
static std::atomic<int64_t> g_census[HSO_CENSUS_N];
static inline void census(int what, int64_t by = 1) { g_census[what].fetch_add(by, std::memory_order_relaxed); }

hipError_t hso_memset_async(void* dst, int value, size_t bytes, hipStream_t stream)
{
  census(HSO_CENSUS_MEMSETS);
  return hipMemsetAsync(dst, value, bytes, stream);
}

hipError_t hso_copy_async(void* dst, const void* src, size_t bytes, hipMemcpyKind kind, hipStream_t stream)
{
  if (bytes == 0) return hipSuccess;
  census(HSO_CENSUS_COPIES); census(HSO_CENSUS_COPY_BYTES, (int64_t)bytes);
  if (kind == hipMemcpyHostToDevice) census(HSO_CENSUS_H2D_BYTES, (int64_t)bytes);
  if (kind == hipMemcpyHostToDevice && !host_is_page_locked(src)) {
    census(HSO_CENSUS_STAGED);
    Stager& S = stager_of(stream);
    std::lock_guard<std::mutex> lk(S.m);
    char* p = stage_alloc(S, bytes);
    if (!p) return hipErrorOutOfMemory;
    memcpy(p, src, bytes);
    return hipMemcpyAsync(dst, p, bytes, hipMemcpyHostToDevice, stream);
  }
  if (kind == hipMemcpyDeviceToHost && !host_is_page_locked(dst)) {
    census(HSO_CENSUS_STAGED);
    Stager& S = stager_of(stream);
    std::lock_guard<std::mutex> lk(S.m);
    char* p = stage_alloc(S, bytes);
    if (!p) return hipErrorOutOfMemory;
    S.fixes.push_back({dst, p, bytes, 0, 0, 0});
    return hipMemcpyAsync(p, src, bytes, hipMemcpyDeviceToHost, stream);
  }
  return hipMemcpyAsync(dst, src, bytes, kind, stream);
}

hipError_t hso_copy2d_async(void* dst, size_t dpitch, const void* src, size_t spitch, size_t width, size_t height, hipMemcpyKind kind,
                            hipStream_t stream)
{
  if (width == 0 || height == 0) return hipSuccess;
  census(HSO_CENSUS_COPIES); census(HSO_CENSUS_COPY_BYTES, (int64_t)(width * height));
  if (kind == hipMemcpyHostToDevice) census(HSO_CENSUS_H2D_BYTES, (int64_t)(width * height));
  if (kind == hipMemcpyDeviceToHost && !host_is_page_locked(dst)) {   // rows packed in the chunk, spread out after the synchronisation
    Stager& S = stager_of(stream);
    std::lock_guard<std::mutex> lk(S.m);
    char* p = stage_alloc(S, width * height);
    if (!p) return hipErrorOutOfMemory;
    S.fixes.push_back({dst, p, width * height, dpitch, width, height});
    return hipMemcpy2DAsync(p, width, src, spitch, width, height, hipMemcpyDeviceToHost, stream);
  }
  if (kind == hipMemcpyHostToDevice && !host_is_page_locked(src)) {
    Stager& S = stager_of(stream);
    std::lock_guard<std::mutex> lk(S.m);
    char* p = stage_alloc(S, width * height);
    if (!p) return hipErrorOutOfMemory;
    for (size_t r = 0; r < height; r++) memcpy(p + r * width, static_cast<const char*>(src) + r * spitch, width);
    return hipMemcpy2DAsync(dst, dpitch, p, width, width, height, hipMemcpyHostToDevice, stream);
  }
  return hipMemcpy2DAsync(dst, dpitch, src, spitch, width, height, kind, stream);
}

// A context that shares the device with others (several banks of sequences per GPU) waits for its stream many times per step, and the
// runtime's hipStreamSynchronize polls: six banks waiting = six of the host's CPUs spent polling while the pools that do the
// bookkeeping compete for the rest of a 16-CPU quota.  A yielding stream records an event and sleeps between queries instead (or, with
// HSO_SYNC_MODE=block, waits on an event created with hipEventBlockingSync): a few tens of microseconds later than a poll would
// notice — nothing against a multi-millisecond step whose device time the other banks fill anyway.
void hso_stream_set_wait(hipStream_t stream, int mode)
{
  Stager& S = stager_of(stream);
  std::lock_guard<std::mutex> lk(S.m);
  if (S.wait_ev && ((S.wait_mode == HSO_WAIT_BLOCK) != (mode == HSO_WAIT_BLOCK))) { (void)hipEventDestroy(S.wait_ev); S.wait_ev = nullptr; }   // the event's flags follow the mode
  S.wait_mode = mode;
}

// how a stream waits (hso_gpu_configure: wait_mode): POLL = hipStreamSynchronize; NAP = record an event, query it, briefly spinning,
// then in short naps; BLOCK = wait on an event created with hipEventBlockingSync.  The blocking wait's wake-up comes with the
// interrupt and was measured bimodal on the GPU boxes (the same 6 x 128 run at 24.7 k or at 13-14 k frames/s from one launch to the
// next); the napping wait is bounded by the nap.
static hipError_t stream_wait(hipStream_t stream)
{
  hipEvent_t ev = nullptr;
  int mode = HSO_WAIT_POLL;
  if (Stager* S = stager_find(stream)) {
    std::lock_guard<std::mutex> lk(S->m);
    mode = S->wait_mode;
    if (mode != HSO_WAIT_POLL) {
      const unsigned flags = (mode == HSO_WAIT_BLOCK ? hipEventBlockingSync : 0u) | hipEventDisableTiming;
      if (!S->wait_ev && hipEventCreateWithFlags(&S->wait_ev, flags) != hipSuccess) S->wait_ev = nullptr;
      ev = S->wait_ev;
    }
  }
  if (!ev) return hipStreamSynchronize(stream);
  hipError_t e = hipEventRecord(ev, stream);
  if (e != hipSuccess) return e;
  if (mode == HSO_WAIT_BLOCK) return hipEventSynchronize(ev);
  const auto t0 = std::chrono::steady_clock::now();
  long slack = -1;                                                  // the caller's timer slack, put back before this returns
  for (;;) {
    e = hipEventQuery(ev);
    if (e != hipErrorNotReady) break;
    if (std::chrono::steady_clock::now() - t0 < std::chrono::microseconds(40)) continue;   // short waits: no sleep at all
    if (slack < 0) { slack = prctl(PR_GET_TIMERSLACK); if (slack < 0) slack = 0; (void)prctl(PR_SET_TIMERSLACK, 2000UL); }   // the default slack (50 us) would triple the nap
    timespec ts{0, 20000};
    nanosleep(&ts, nullptr);
  }
  if (slack > 0) (void)prctl(PR_SET_TIMERSLACK, (unsigned long)slack);
  return e;
}

hipError_t hso_stream_sync(hipStream_t stream)
{
  const auto t0 = std::chrono::steady_clock::now();
  const hipError_t e = stream_wait(stream);
  census(HSO_CENSUS_SYNCS);
  census(HSO_CENSUS_SYNC_NS, std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count());
  Stager* const Sp = stager_find(stream);
  if (!Sp) return e;
  Stager& S = *Sp;
  std::lock_guard<std::mutex> lk(S.m);
  if (e == hipSuccess)
    for (const StageFix& f : S.fixes) {
      if (f.height == 0) memcpy(f.dst, f.src, f.bytes);
      else for (size_t r = 0; r < f.height; r++) memcpy(static_cast<char*>(f.dst) + r * f.dpitch, f.src + r * f.width, f.width);
    }
  S.fixes.clear();
  for (StageChunk& c : S.chunks) c.used = 0;
  return e;
}

// the blocking copy on the null stream (debug / test read-backs): through a chunk of the null stream's stager
hipError_t hso_copy_sync(void* dst, const void* src, size_t bytes, hipMemcpyKind kind)
{
  const hipError_t e = hso_copy_async(dst, src, bytes, kind, nullptr);
  if (e != hipSuccess) return e;
  return hso_stream_sync(nullptr);
}

char* hso_stage_reserve(hipStream_t stream, size_t bytes)
{
  Stager& S = stager_of(stream);
  std::lock_guard<std::mutex> lk(S.m);
  return stage_alloc(S, std::max<size_t>(bytes, 1));
}

void hso_stream_abandon(hipStream_t stream)
{
  (void)hipStreamSynchronize(stream);                      // the raw call: nothing is copied out
  (void)hipGetLastError();
  Stager* const S = stager_find(stream);
  if (!S) return;
  std::lock_guard<std::mutex> lk(S->m);
  S->fixes.clear();
  for (StageChunk& c : S->chunks) c.used = 0;
}

void hso_stream_forget(hipStream_t stream)
{
  std::lock_guard<std::mutex> lk(g_stage_mutex);
  auto it = g_stagers.find(stream);
  if (it == g_stagers.end()) return;
  for (StageChunk& c : it->second.chunks) (void)hipHostFree(c.p);
  if (it->second.wait_ev) (void)hipEventDestroy(it->second.wait_ev);
  g_stagers.erase(it);
}

#define hipMemcpyAsync(dst, src, bytes, kind, stream) hso_copy_async((dst), (src), (bytes), (kind), (stream))
#define hipMemcpy2DAsync(dst, dpitch, src, spitch, width, height, kind, stream) \
  hso_copy2d_async((dst), (dpitch), (src), (spitch), (width), (height), (kind), (stream))
#define hipStreamSynchronize(stream) hso_stream_sync(stream)
#define hipMemsetAsync(dst, value, bytes, stream) hso_memset_async((dst), (value), (bytes), (stream))

// Every failing exit of an entry point ends here or in HSO_HIP_CHECK: device-to-host copies already enqueued into the CALLER's
// memory (finished by the next synchronisation, hso_stream_sync) must not outlive the failed call — the caller may free its
// buffers the moment it sees the error.  The common case (argument checks before anything was enqueued) costs a map lookup.
int hso_fail(hso_gpu_ctx* ctx, int code, const char* msg)
{
  if (!ctx) return code;
  ctx->err = msg;
  bool pending = false;
  {
    if (Stager* S = stager_find(ctx->stream)) { std::lock_guard<std::mutex> lk(S->m); pending = !S->fixes.empty(); }
  }
  if (pending) hso_stream_abandon(ctx->stream);
  return code;
}

// Frame storage comes in slabs: the device allocator takes 0.1-0.2 ms per call on this stack, and a bank of 64 sequences
// constructs 64 frames per step (10 ms of a step went into 64 hipMalloc calls until the first frames were released).  A slab
// holds HSO_FRAMES_PER_SLAB frames of one geometry; pieces are handed out and returned through the free list of that geometry and
// the slabs are freed with the context.
#define HSO_FRAMES_PER_SLAB 32
static inline uint64_t geom_key(const PyrGeom& g) { return ((uint64_t)(uint32_t)g.w[0] << 32) | (uint32_t)g.h[0]; }
int hso_frame_alloc(hso_gpu_ctx* ctx, const PyrGeom& g, uint8_t** base)
{
  std::vector<uint8_t*>& fl = ctx->free_frames[geom_key(g)];
  if (!fl.empty()) {
    *base = fl.back();
    fl.pop_back();
    return HSO_OK;
  }
  *base = nullptr;
  const size_t stride = ((size_t)g.frame_bytes + 255) & ~size_t(255);
  uint8_t* slab = nullptr;
  HSO_HIP_CHECK(ctx, hipMalloc(reinterpret_cast<void**>(&slab), stride * HSO_FRAMES_PER_SLAB));
  ctx->frame_slabs.push_back(slab);
  // zero once: the inter-level padding rows must read as 0 (see hso_ctx.h)
  hipError_t e = hipMemsetAsync(slab, 0, stride * HSO_FRAMES_PER_SLAB, ctx->stream);
  if (e != hipSuccess) { ctx->err = hipGetErrorString(e); return HSO_E_HIP; }
  for (int i = HSO_FRAMES_PER_SLAB - 1; i >= 1; i--) fl.push_back(slab + stride * (size_t)i);
  *base = slab;
  return HSO_OK;
}

void hso_frame_free(hso_gpu_ctx* ctx, const PyrGeom& g, uint8_t* base)
{
  if (!base) return;
  ctx->free_frames[geom_key(g)].push_back(base);   // every piece goes back to the free list of its own geometry
}

// run `expr`; on failure give the frame allocation back and return the status
#define HSO_FRAME_TRY(ctx, g, base, expr)                                      \
  do {                                                                        \
    hipError_t _e = (expr);                                                   \
    if (_e != hipSuccess) {                                                   \
      (ctx)->err = std::string(#expr) + ": " + hipGetErrorString(_e);         \
      (void)hipStreamSynchronize((ctx)->stream);                              \
      hso_frame_free(ctx, g, base);                                           \
      return HSO_E_HIP;                                                       \
    }                                                                         \
  } while (0)

extern "C" {

int hso_gpu_abi_version(void) { return HSO_GPU_ABI_VERSION; }

int hso_gpu_create(hso_gpu_ctx** out, int device, void* stream)
{
  if (!out) return HSO_E_INVALID;
  *out = nullptr;
  int n_dev = 0;
  if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0) return HSO_E_HIP;
  if (device < 0 || device >= n_dev) return HSO_E_INVALID;
  if (hipSetDevice(device) != hipSuccess) return HSO_E_HIP;
  hso_gpu_ctx* ctx = new hso_gpu_ctx();
  ctx->device = device;
  ctx->track = nullptr;
  ctx->seed_tables = nullptr;
  ctx->seqmaps = nullptr;
  ctx->d_batch = nullptr;
  ctx->d_seed_scratch = nullptr; ctx->seed_scratch_cap = 0;
  ctx->seed_stream = nullptr; ctx->seed_go = nullptr; ctx->seed_done = nullptr;
  ctx->seed_inflight = false; ctx->seed_inflight_table = -1; ctx->seed_inflight_n = 0;
  ctx->d_seed_scratch_async = nullptr; ctx->seed_scratch_async_cap = 0;
  ctx->h_seed_pin = nullptr; ctx->h_seed_pin_cap = 0; ctx->h_seed_brief_off = 0;
  ctx->batch_cap = 0;
  ctx->h_pin[0] = ctx->h_pin[1] = nullptr;
  ctx->h_pin_cap[0] = ctx->h_pin_cap[1] = 0;
  ctx->d_pack = nullptr; ctx->d_pack_cap = 0; ctx->h_pack = nullptr; ctx->h_pack_cap = 0;
  if (stream) {
    // "one context per stream" is enforced: the page-locked staging of a stream (chunks, copies waiting for its next
    // synchronisation) belongs to the one context that launches on it; two contexts on one stream, on different threads, would
    // finish and recycle each other's copies
    std::lock_guard<std::mutex> lk(g_stage_mutex);
    if (g_streams_in_use.count(reinterpret_cast<hipStream_t>(stream))) { delete ctx; return HSO_E_INVALID; }
    g_streams_in_use.insert(reinterpret_cast<hipStream_t>(stream));
    ctx->stream = reinterpret_cast<hipStream_t>(stream);
    ctx->own_stream = false;
  } else {
    if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) { delete ctx; return HSO_E_HIP; }
    ctx->own_stream = true;
  }
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) != hipSuccess) {
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->stream);
    else { std::lock_guard<std::mutex> lk(g_stage_mutex); g_streams_in_use.erase(ctx->stream); }
    delete ctx;
    return HSO_E_HIP;
  }
  ctx->n_cu = prop.multiProcessorCount;
  *out = ctx;
  return HSO_OK;
}

void hso_gpu_destroy(hso_gpu_ctx* ctx)
{
  if (!ctx) return;
  (void)hipSetDevice(ctx->device);
  (void)hipStreamSynchronize(ctx->stream);
  hso_track_state_free(ctx);
  hso_seed_tables_free(ctx);
  hso_seqmaps_free(ctx);
  hso_chain_forget(ctx);
  for (auto* p : ctx->frame_slabs) (void)hipFree(p);
  if (ctx->d_batch) (void)hipFree(ctx->d_batch);
  hso_rba_forget(ctx);
  if (ctx->d_seed_scratch) (void)hipFree(ctx->d_seed_scratch);
  if (ctx->seed_stream) { (void)hipStreamSynchronize(ctx->seed_stream); (void)hipStreamDestroy(ctx->seed_stream); }
  if (ctx->seed_go) (void)hipEventDestroy(ctx->seed_go);
  if (ctx->seed_done) (void)hipEventDestroy(ctx->seed_done);
  if (ctx->d_seed_scratch_async) (void)hipFree(ctx->d_seed_scratch_async);
  if (ctx->h_seed_pin) (void)hipHostFree(ctx->h_seed_pin);
  for (int k = 0; k < 2; k++) if (ctx->h_pin[k]) (void)hipHostFree(ctx->h_pin[k]);
  if (ctx->d_pack) (void)hipFree(ctx->d_pack);
  if (ctx->h_pack) (void)hipHostFree(ctx->h_pack);
  for (void* p : ctx->host_allocs) (void)hipHostFree(p);
  hso_stream_forget(ctx->stream);
  if (ctx->own_stream) (void)hipStreamDestroy(ctx->stream);
  else { std::lock_guard<std::mutex> lk(g_stage_mutex); g_streams_in_use.erase(ctx->stream); }
  delete ctx;
}

extern "C++" char* hso_pinned(hso_gpu_ctx* ctx, int slot, size_t bytes)
{
  if (ctx->h_pin_cap[slot] >= bytes && ctx->h_pin[slot]) return ctx->h_pin[slot];
  if (ctx->h_pin[slot]) { (void)hipStreamSynchronize(ctx->stream); (void)hipHostFree(ctx->h_pin[slot]); ctx->h_pin[slot] = nullptr; ctx->h_pin_cap[slot] = 0; }
  const size_t cap = bytes + bytes / 2 + 4096;
  void* p = nullptr;
  if (hipHostMalloc(&p, cap, hipHostMallocDefault) != hipSuccess) { ctx->err = "hipHostMalloc failed"; return nullptr; }
  ctx->h_pin[slot] = static_cast<char*>(p); ctx->h_pin_cap[slot] = cap;
  return ctx->h_pin[slot];
}

}  // extern "C"

// ---- hso_lists_to_host ----
struct ListRec { const uint32_t* src; uint64_t dst_word, n_words; };

__global__ __launch_bounds__(256) void k_gather_lists(const ListRec* recs, uint32_t* out)
{
  const ListRec r = recs[blockIdx.y];
  uint32_t* o = out + r.dst_word;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < r.n_words; i += (uint64_t)gridDim.x * blockDim.x) o[i] = r.src[i];
}

int hso_lists_to_host(hso_gpu_ctx* ctx, const std::vector<HsoListCopy>& lists)
{
  std::vector<const HsoListCopy*> live;
  for (const HsoListCopy& c : lists) if (c.bytes > 0) live.push_back(&c);
  if (live.empty()) { HSO_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream)); return HSO_OK; }
  const size_t n = live.size();
  const size_t tab = (sizeof(ListRec) * n + 255) & ~size_t(255);
  size_t words = 0, longest = 0;
  for (const HsoListCopy* c : live) {
    if ((c->bytes & 3) || (reinterpret_cast<uintptr_t>(c->src) & 3)) return hso_fail(ctx, HSO_E_INVALID, "lists_to_host: list not in 4-byte units");
    words += c->bytes / 4; longest = std::max(longest, c->bytes / 4);
  }
  const size_t need = tab + words * 4;
  if (ctx->d_pack_cap < need || ctx->h_pack_cap < need) {
    HSO_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->d_pack) (void)hipFree(ctx->d_pack);
    if (ctx->h_pack) (void)hipHostFree(ctx->h_pack);
    ctx->d_pack = nullptr; ctx->h_pack = nullptr; ctx->d_pack_cap = ctx->h_pack_cap = 0;
    const size_t cap = hso_grown(need);
    HSO_HIP_CHECK(ctx, hipMalloc(reinterpret_cast<void**>(&ctx->d_pack), cap));
    ctx->d_pack_cap = cap;
    HSO_HIP_CHECK(ctx, hipHostMalloc(reinterpret_cast<void**>(&ctx->h_pack), cap, hipHostMallocDefault));
    ctx->h_pack_cap = cap;
  }
  ListRec* hr = reinterpret_cast<ListRec*>(ctx->h_pack);
  size_t at = 0;
  for (size_t i = 0; i < n; i++) { hr[i].src = static_cast<const uint32_t*>(live[i]->src); hr[i].dst_word = at; hr[i].n_words = live[i]->bytes / 4; at += live[i]->bytes / 4; }
  HSO_HIP_CHECK(ctx, hipMemcpyAsync(ctx->d_pack, ctx->h_pack, sizeof(ListRec) * n, hipMemcpyHostToDevice, ctx->stream));
  const unsigned gx = (unsigned)std::min<size_t>(16, (longest + 1023) / 1024);
  for (size_t y0 = 0; y0 < n; y0 += 32768)     // blockIdx.y is a 16-bit quantity
    hipLaunchKernelGGL(k_gather_lists, dim3(gx ? gx : 1, (unsigned)std::min<size_t>(32768, n - y0)), dim3(256), 0, ctx->stream,
                       reinterpret_cast<const ListRec*>(ctx->d_pack) + y0, reinterpret_cast<uint32_t*>(ctx->d_pack + tab));
  HSO_HIP_CHECK(ctx, hipGetLastError());
  HSO_HIP_CHECK(ctx, hipMemcpyAsync(ctx->h_pack + tab, ctx->d_pack + tab, words * 4, hipMemcpyDeviceToHost, ctx->stream));
  HSO_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  at = 0;
  for (size_t i = 0; i < n; i++) { memcpy(live[i]->dst, ctx->h_pack + tab + at, live[i]->bytes); at += live[i]->bytes; }
  return HSO_OK;
}

extern "C" {

const char* hso_gpu_last_error(const hso_gpu_ctx* ctx) { return ctx ? ctx->err.c_str() : "null context"; }

int hso_gpu_host_alloc(hso_gpu_ctx* ctx, size_t bytes, void** out)
{
  if (!ctx || !out) return HSO_E_INVALID;
  *out = nullptr;
  void* p = nullptr;
  if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) return hso_fail(ctx, HSO_E_NOMEM, "host_alloc: hipHostMalloc failed");
  ctx->host_allocs.push_back(p);
  *out = p;
  return HSO_OK;
}

int hso_gpu_host_free(hso_gpu_ctx* ctx, void* p)
{
  if (!ctx) return HSO_E_INVALID;
  for (size_t i = 0; i < ctx->host_allocs.size(); i++)
    if (ctx->host_allocs[i] == p) {
      HSO_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
      (void)hipHostFree(p);
      ctx->host_allocs.erase(ctx->host_allocs.begin() + (long)i);
      return HSO_OK;
    }
  return hso_fail(ctx, HSO_E_INVALID, "host_free: not an allocation of hso_gpu_host_alloc");
}

void hso_gpu_debug_census(int64_t* out, int n)
{
  for (int i = 0; i < n; i++) out[i] = i < HSO_CENSUS_N ? g_census[i].load(std::memory_order_relaxed) : 0;
}

static void hso_apply_wait_mode(hso_gpu_ctx* ctx)
{
  const int m = ctx->opt.wait_mode != HSO_WAIT_DEFAULT ? ctx->opt.wait_mode : (ctx->shared_device ? HSO_WAIT_NAP : HSO_WAIT_POLL);
  hso_stream_set_wait(ctx->stream, m);
}

int hso_gpu_set_host_parallel(hso_gpu_ctx* ctx, hso_parallel_for_fn parallel_for, void* user)
{
  if (!ctx) return HSO_E_INVALID;
  ctx->par_fn = parallel_for; ctx->par_user = parallel_for ? user : nullptr;
  return HSO_OK;
}

int hso_gpu_device_cpulist(hso_gpu_ctx* ctx, char* out, size_t cap)
{
  if (!ctx) return HSO_E_INVALID;
  if (!out || cap == 0) return hso_fail(ctx, HSO_E_INVALID, "device_cpulist: bad argument");
  out[0] = 0;
  char bus[64] = {0};
  if (hipDeviceGetPCIBusId(bus, (int)sizeof(bus) - 1, ctx->device) != hipSuccess) { (void)hipGetLastError(); return HSO_OK; }
  for (char* c = bus; *c; ++c) if (*c >= 'A' && *c <= 'F') *c = (char)(*c - 'A' + 'a');   // sysfs spells bus addresses in lower case
  char path[160];
  snprintf(path, sizeof(path), "/sys/bus/pci/devices/%s/numa_node", bus);
  int node = -1;
  if (FILE* f = fopen(path, "r")) { if (fscanf(f, "%d", &node) != 1) node = -1; fclose(f); }
  if (node < 0) return HSO_OK;
  snprintf(path, sizeof(path), "/sys/devices/system/node/node%d/cpulist", node);
  if (FILE* f = fopen(path, "r")) {
    if (!fgets(out, (int)cap, f)) out[0] = 0;
    fclose(f);
    for (char* c = out; *c; ++c) if (*c == '\n' || *c == ' ') { *c = 0; break; }
  }
  return HSO_OK;
}

int hso_gpu_set_shared_device(hso_gpu_ctx* ctx, int shared)
{
  if (!ctx) return HSO_E_INVALID;
  ctx->shared_device = shared != 0;
  hso_apply_wait_mode(ctx);
  return HSO_OK;
}

int hso_gpu_configure(hso_gpu_ctx* ctx, const hso_gpu_options* o)
{
  if (!ctx) return HSO_E_INVALID;
  if (!o || o->size < (int32_t)(4 * sizeof(int32_t)) || o->size > (int32_t)sizeof(hso_gpu_options)) return hso_fail(ctx, HSO_E_INVALID, "configure: options missing or of an unknown size");
  hso_gpu_options n{};
  memcpy(&n, o, (size_t)o->size);
  if (n.wait_mode < HSO_WAIT_DEFAULT || n.wait_mode > HSO_WAIT_BLOCK || n.track_coop_feats_per_wg < 0 || n.track_coop_workgroups < 0)
    return hso_fail(ctx, HSO_E_INVALID, "configure: value out of range");
  ctx->opt = n;
  hso_apply_wait_mode(ctx);
  return HSO_OK;
}

int hso_gpu_synchronize(hso_gpu_ctx* ctx)
{
  if (!ctx) return HSO_E_INVALID;
  if (int rc = hso_seed_async_quiesce(ctx)) return rc;
  HSO_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  return HSO_OK;
}

int hso_gpu_frame_upload(hso_gpu_ctx* ctx, int64_t frame_id, const uint8_t* img, int width, int height,
                         int img_is_device, hso_frame_stats* stats_out)
{
  if (!ctx) return HSO_E_INVALID;
  if (!img || width <= 0 || height <= 0) return hso_fail(ctx, HSO_E_INVALID, "frame_upload: null image or bad size");
  // src/frame.cpp:302-312: halfSample pyramid when cols and rows are multiples of 16, cv::resize
  // pyramid otherwise; the kernels group four level-0 pixels per lane
  if ((width % 4) != 0)
    return hso_fail(ctx, HSO_E_INVALID, "frame_upload: width must be a multiple of 4");
  if (width < 64 || height < 64) return hso_fail(ctx, HSO_E_INVALID, "frame_upload: image smaller than 64x64");
  if (ctx->frames.count(frame_id)) return hso_fail(ctx, HSO_E_INVALID, "frame_upload: frame id already resident");
  HSO_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  FrameRec rec;
  rec.id = frame_id;
  rec.g = make_geom(width, height);
  rec.base = nullptr;
  if (int rc = hso_frame_alloc(ctx, rec.g, &rec.base)) return rc;
  HSO_FRAME_TRY(ctx, rec.g, rec.base, hipMemcpyAsync(rec.base + rec.g.off[0], img, (size_t)width * height,
                                                     img_is_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, ctx->stream));
  // the kernels take a device array of frame base pointers (batched form); one entry here
  uint8_t** d_bases = reinterpret_cast<uint8_t**>(rec.base + rec.g.stats_off + 128);
  HSO_FRAME_TRY(ctx, rec.g, rec.base, hipMemcpyAsync(d_bases, &rec.base, sizeof(uint8_t*), hipMemcpyHostToDevice, ctx->stream));
  int rc = hso_frame_build(ctx, rec.g, d_bases, nullptr, nullptr, 1);
  if (rc < 0) { (void)hipStreamSynchronize(ctx->stream); hso_frame_free(ctx, rec.g, rec.base); return rc; }
  if (stats_out) {
    HSO_FRAME_TRY(ctx, rec.g, rec.base, hipMemcpyAsync(stats_out, rec.base + rec.g.stats_off, sizeof(hso_frame_stats),
                                                       hipMemcpyDeviceToHost, ctx->stream));
    HSO_FRAME_TRY(ctx, rec.g, rec.base, hipStreamSynchronize(ctx->stream));
  }
  ctx->frames[frame_id] = rec;
  return HSO_OK;
}

int hso_gpu_frame_upload_resized(hso_gpu_ctx* ctx, int64_t frame_id, const uint8_t* img, int src_width, int src_height,
                                 int width, int height, int img_is_device, hso_frame_stats* stats_out)
{
  if (!ctx) return HSO_E_INVALID;
  if (!img || src_width <= 0 || src_height <= 0) return hso_fail(ctx, HSO_E_INVALID, "frame_upload_resized: null image or bad size");
  if (src_width == width && src_height == height) return hso_gpu_frame_upload(ctx, frame_id, img, width, height, img_is_device, stats_out);
  if ((width % 4) != 0 || width < 64 || height < 64)
    return hso_fail(ctx, HSO_E_INVALID, "frame_upload_resized: width must be a multiple of 4, width and height >= 64");
  if (ctx->frames.count(frame_id)) return hso_fail(ctx, HSO_E_INVALID, "frame_upload_resized: frame id already resident");
  HSO_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  const uint8_t* d_src = img;
  if (!img_is_device) {
    const size_t need = (size_t)src_width * src_height;
    if (ctx->batch_cap < need) {
      HSO_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
      if (ctx->d_batch) (void)hipFree(ctx->d_batch);
      ctx->d_batch = nullptr; ctx->batch_cap = 0;
      HSO_HIP_CHECK(ctx, hipMalloc(reinterpret_cast<void**>(&ctx->d_batch), hso_grown(need)));
      ctx->batch_cap = hso_grown(need);
    }
    HSO_HIP_CHECK(ctx, hipMemcpyAsync(ctx->d_batch, img, need, hipMemcpyHostToDevice, ctx->stream));
    d_src = reinterpret_cast<const uint8_t*>(ctx->d_batch);
  }
  FrameRec rec;
  rec.id = frame_id;
  rec.g = make_geom(width, height);
  rec.base = nullptr;
  if (int rc = hso_frame_alloc(ctx, rec.g, &rec.base)) return rc;
  int rc = hso_frame_resize_into(ctx, d_src, src_width, src_height, rec.base + rec.g.off[0], width, height);
  if (rc < 0) { (void)hipStreamSynchronize(ctx->stream); hso_frame_free(ctx, rec.g, rec.base); return rc; }
  uint8_t** d_bases = reinterpret_cast<uint8_t**>(rec.base + rec.g.stats_off + 128);
  HSO_FRAME_TRY(ctx, rec.g, rec.base, hipMemcpyAsync(d_bases, &rec.base, sizeof(uint8_t*), hipMemcpyHostToDevice, ctx->stream));
  rc = hso_frame_build(ctx, rec.g, d_bases, nullptr, nullptr, 1);
  if (rc < 0) { (void)hipStreamSynchronize(ctx->stream); hso_frame_free(ctx, rec.g, rec.base); return rc; }
  if (stats_out)
    HSO_FRAME_TRY(ctx, rec.g, rec.base, hipMemcpyAsync(stats_out, rec.base + rec.g.stats_off, sizeof(hso_frame_stats), hipMemcpyDeviceToHost, ctx->stream));
  HSO_FRAME_TRY(ctx, rec.g, rec.base, hipStreamSynchronize(ctx->stream));   // the staging buffer is reused by later calls
  ctx->frames[frame_id] = rec;
  return HSO_OK;
}

int hso_gpu_frame_upload_batch(hso_gpu_ctx* ctx, const int64_t* frame_ids, const uint8_t* const* imgs, int n,
                               int width, int height, int img_is_device, hso_frame_stats* stats_out)
{
  if (!ctx) return HSO_E_INVALID;
  if (!frame_ids || !imgs || n <= 0) return hso_fail(ctx, HSO_E_INVALID, "frame_upload_batch: null argument or n <= 0");
  if ((width % 4) != 0 || width < 64 || height < 64)
    return hso_fail(ctx, HSO_E_INVALID, "frame_upload_batch: width must be a multiple of 4, width and height >= 64");
  HSO_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  const PyrGeom g = make_geom(width, height);
  // validate every entry, then allocate the new ones; ctx->frames is only touched once nothing can fail any more,
  // so an error in the middle of a batch leaves no half-built frame resident
  std::vector<uint8_t*> bases(n);
  std::vector<int> fresh;
  for (int i = 0; i < n; i++) {
    if (!imgs[i]) return hso_fail(ctx, HSO_E_INVALID, "frame_upload_batch: null image");
    auto it = ctx->frames.find(frame_ids[i]);
    if (it != ctx->frames.end() && !same_geom(it->second.g, g))
      return hso_fail(ctx, HSO_E_INVALID, "frame_upload_batch: resident frame has another size");
    for (int k = 0; k < i && it == ctx->frames.end(); k++)
      if (frame_ids[k] == frame_ids[i]) return hso_fail(ctx, HSO_E_INVALID, "frame_upload_batch: a new frame id appears twice");
  }
  auto undo = [&]() { hso_stream_abandon(ctx->stream); for (int i : fresh) hso_frame_free(ctx, g, bases[i]); };
  for (int i = 0; i < n; i++) {
    auto it = ctx->frames.find(frame_ids[i]);
    if (it != ctx->frames.end()) { bases[i] = it->second.base; continue; }   // refresh in place
    if (int rc = hso_frame_alloc(ctx, g, &bases[i])) { undo(); return rc; }
    fresh.push_back(i);
  }
#define HSO_BATCH_TRY(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) { ctx->err = std::string(#expr) + ": " + hipGetErrorString(_e); undo(); return HSO_E_HIP; } } while (0)
  const size_t b_ptr = ((size_t)n * sizeof(void*) + 255) & ~size_t(255);
  const size_t need = 2 * b_ptr + (size_t)n * sizeof(hso_frame_stats);
  if (ctx->batch_cap < need) {
    HSO_BATCH_TRY(hipStreamSynchronize(ctx->stream));
    if (ctx->d_batch) (void)hipFree(ctx->d_batch);
    ctx->d_batch = nullptr; ctx->batch_cap = 0;
    HSO_BATCH_TRY(hipMalloc(reinterpret_cast<void**>(&ctx->d_batch), hso_grown(need)));
    ctx->batch_cap = hso_grown(need);
  }
  uint8_t** d_bases = reinterpret_cast<uint8_t**>(ctx->d_batch);
  const uint8_t** d_srcs = reinterpret_cast<const uint8_t**>(ctx->d_batch + b_ptr);
  hso_frame_stats* d_stats = reinterpret_cast<hso_frame_stats*>(ctx->d_batch + 2 * b_ptr);
  HSO_BATCH_TRY(hipMemcpyAsync(d_bases, bases.data(), (size_t)n * sizeof(void*), hipMemcpyHostToDevice, ctx->stream));
  if (img_is_device) {
    HSO_BATCH_TRY(hipMemcpyAsync(d_srcs, imgs, (size_t)n * sizeof(void*), hipMemcpyHostToDevice, ctx->stream));
  } else {
    for (int i = 0; i < n; i++)
      HSO_BATCH_TRY(hipMemcpyAsync(bases[i] + g.off[0], imgs[i], (size_t)width * height, hipMemcpyHostToDevice, ctx->stream));
  }
  int rc = hso_frame_build(ctx, g, d_bases, img_is_device ? d_srcs : nullptr, stats_out ? d_stats : nullptr, n);
  if (rc < 0) { undo(); return rc; }
  if (stats_out) {
    HSO_BATCH_TRY(hipMemcpyAsync(stats_out, d_stats, (size_t)n * sizeof(hso_frame_stats), hipMemcpyDeviceToHost, ctx->stream));
    HSO_BATCH_TRY(hipStreamSynchronize(ctx->stream));
  }
#undef HSO_BATCH_TRY
  for (int i : fresh) {   // commit
    FrameRec rec;
    rec.id = frame_ids[i]; rec.g = g; rec.base = bases[i];
    ctx->frames[frame_ids[i]] = rec;
  }
  return HSO_OK;
}

int hso_gpu_frame_release(hso_gpu_ctx* ctx, int64_t frame_id)
{
  if (!ctx) return HSO_E_INVALID;
  auto it = ctx->frames.find(frame_id);
  if (it == ctx->frames.end()) return hso_fail(ctx, HSO_E_NOFRAME, "frame_release: frame not resident");
  if (hso_seed_tables_pin(ctx, frame_id))
    return hso_fail(ctx, HSO_E_INVALID, "frame_release: live seeds of a resident seed table are hosted in this frame (erase them or destroy the table first)");
  if (int rc = hso_seed_async_quiesce(ctx)) return rc;   // a pass in flight may be reading this frame
  HSO_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  hso_frame_free(ctx, it->second.g, it->second.base);
  ctx->frames.erase(it);
  return HSO_OK;
}

int hso_gpu_frame_release_batch(hso_gpu_ctx* ctx, const int64_t* frame_ids, int n)
{
  if (!ctx) return HSO_E_INVALID;
  if (n < 0 || (n > 0 && !frame_ids)) return hso_fail(ctx, HSO_E_INVALID, "frame_release_batch: bad argument");
  if (n == 0) return HSO_OK;
  // nothing is released unless everything can be
  for (int i = 0; i < n; i++) {
    if (ctx->frames.find(frame_ids[i]) == ctx->frames.end()) return hso_fail(ctx, HSO_E_NOFRAME, "frame_release_batch: frame not resident");
    if (hso_seed_tables_pin(ctx, frame_ids[i]))
      return hso_fail(ctx, HSO_E_INVALID, "frame_release_batch: live seeds of a resident seed table are hosted in one of the frames");
  }
  if (int rc = hso_seed_async_quiesce(ctx)) return rc;   // a pass in flight may be reading these frames
  HSO_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));  // ONE wait for all of them (a bank of 128 sequences releases ~50 frames per step)
  for (int i = 0; i < n; i++) {
    auto it = ctx->frames.find(frame_ids[i]);
    if (it == ctx->frames.end()) continue;                 // an id named twice
    hso_frame_free(ctx, it->second.g, it->second.base);
    ctx->frames.erase(it);
  }
  return HSO_OK;
}

int hso_gpu_frame_download_level(hso_gpu_ctx* ctx, int64_t frame_id, int level, uint8_t* out, int* w_out, int* h_out)
{
  if (!ctx || !out) return HSO_E_INVALID;
  if (level < 0 || level >= HSO_N_PYR_LEVELS) return hso_fail(ctx, HSO_E_INVALID, "download_level: bad level");
  auto it = ctx->frames.find(frame_id);
  if (it == ctx->frames.end()) return hso_fail(ctx, HSO_E_NOFRAME, "download_level: frame not resident");
  const PyrGeom& g = it->second.g;
  HSO_HIP_CHECK(ctx, hipMemcpyAsync(out, it->second.base + g.off[level], (size_t)g.w[level] * g.h[level],
                                    hipMemcpyDeviceToHost, ctx->stream));
  HSO_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  if (w_out) *w_out = g.w[level];
  if (h_out) *h_out = g.h[level];
  return HSO_OK;
}

int hso_gpu_frame_download_sobel(hso_gpu_ctx* ctx, int64_t frame_id, int level, int16_t* gx, int16_t* gy)
{
  if (!ctx || !gx || !gy) return HSO_E_INVALID;
  if (level < 0 || level >= HSO_N_SOBEL_LEVELS) return hso_fail(ctx, HSO_E_INVALID, "download_sobel: bad level");
  auto it = ctx->frames.find(frame_id);
  if (it == ctx->frames.end()) return hso_fail(ctx, HSO_E_NOFRAME, "download_sobel: frame not resident");
  const PyrGeom& g = it->second.g;
  const size_t row = (size_t)g.w[level] * 2, pitch = (size_t)g.sob_stride[level] * 2;   // rows are padded to 128-byte lines on the device
  HSO_HIP_CHECK(ctx, hipMemcpy2DAsync(gx, row, it->second.base + g.sob_off[level][0], pitch, row, (size_t)g.h[level], hipMemcpyDeviceToHost, ctx->stream));
  HSO_HIP_CHECK(ctx, hipMemcpy2DAsync(gy, row, it->second.base + g.sob_off[level][1], pitch, row, (size_t)g.h[level], hipMemcpyDeviceToHost, ctx->stream));
  HSO_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  return HSO_OK;
}

}  // extern "C"
