// hso_engine.cpp — the sequence engine (see hso_engine.h): per-sequence tables, the phases of a step, the keyframe work.
#include "hso_engine_impl.h"

namespace hso {
namespace engine {

// ------------------------------------------------------------------------------------------------ trace
bool Trace::open(const char* path) { close(); f = std::fopen(path, "wb"); return f != nullptr; }
void Trace::close() { if (f) std::fclose(f); f = nullptr; }
void Trace::begin(const char* name, uint32_t nf)
{
  const uint32_t magic = 0x52545348u, nl = (uint32_t)std::strlen(name);
  std::fwrite(&magic, 4, 1, f); std::fwrite(&nl, 4, 1, f); std::fwrite(name, 1, nl, f); std::fwrite(&nf, 4, 1, f);
}
void Trace::field(const char* key, const void* data, size_t bytes)
{
  const uint32_t kl = (uint32_t)std::strlen(key); const uint64_t nb = bytes;
  std::fwrite(&kl, 4, 1, f); std::fwrite(key, 1, kl, f); std::fwrite(&nb, 8, 1, f);
  if (bytes) std::fwrite(data, 1, bytes, f);
}

// ------------------------------------------------------------------------------------------------ bank
// How many CPUs the process may keep busy: the hardware threads, or less when a cgroup caps its CPU time (cpu.max: "quota period" in
// microseconds).  The GPU boxes this was measured on show 256 hardware threads under a quota of 16 CPUs: 31 bookkeeping threads per
// bank, three banks, ran into the throttle (whole steps stalled for a scheduler period); about one worker per CPU of the quota and
// bank is where the throughput peaks (tools/bank_threads.sh: 3 banks x 96 sequences 17.2 k frames/s with 31 workers each, 19.6 k with 12).
static unsigned cpu_budget()
{
  unsigned n = std::thread::hardware_concurrency();
  if (n == 0) n = 1;
  double quota = -1, period = -1;
  if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {               // cgroup v2
    char q[64] = {0};
    if (fscanf(f, "%63s %lf", q, &period) == 2 && strcmp(q, "max") != 0) quota = atof(q);
    fclose(f);
  } else {
    if (FILE* g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) { if (fscanf(g, "%lf", &quota) != 1) quota = -1; fclose(g); }   // v1
    if (FILE* g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (fscanf(g, "%lf", &period) != 1) period = -1; fclose(g); }
  }
  if (quota > 0 && period > 0) n = std::min<unsigned>(n, (unsigned)std::max(1.0, std::floor(quota / period + 0.5)));
  return n;
}

// How many engines share this process's CPU budget (hso_vo_host_share) and how many processes share the host (LOCAL_WORLD_SIZE of a
// torchrun launch, one rank per GPU): a bank sizes its pool to its share.  Round 4 sized every pool to the whole quota: 8 ranks x 3
// banks x 15 workers on a 16-CPU quota would have spent the run in the CFS throttle.
// developer probe: HSO_ENGINE_TIMING=1 prints the phases' wall time when a bank goes away, =2 one line per step as well
int timing_level() { static const int v = [] { const char* e = getenv("HSO_ENGINE_TIMING"); return e ? std::max(1, atoi(e)) : 0; }(); return v; }

static std::atomic<int> g_host_share{1};
void set_host_share(int banks_in_process) { g_host_share.store(std::max(1, banks_in_process)); }
int host_cpu_budget() { return (int)cpu_budget(); }
int pool_threads_for(int n_sequences)
{
  if (n_sequences <= 1) return 0;
  if (const char* e = getenv("HSO_ENGINE_THREADS")) return std::max(0, atoi(e));
  int ranks = 1;
  if (const char* e = getenv("LOCAL_WORLD_SIZE")) ranks = std::max(1, atoi(e));
  const int budget = (int)cpu_budget(), banks = g_host_share.load();
  // Workers sleep between a step's phases and while the bank waits for the device, so the pools together may hold about 3.5 x the
  // budget before the scheduler's quota bites (measured on the 16-CPU GPU boxes, 120-frame runs at 2000 features: 4 banks x 9
  // workers 20.2 k frames/s against 16.8 k with 6 and 15.8 k with 3; 6 banks x 7: 22.0 k against 17.6 k with 4; 6 x 15: 20.0 k with
  // 38 of 75 scheduler periods throttled).  A lone bank keeps one CPU for its own thread.
  // (With the banks' own threads asleep while they wait for the device — hso_gpu_set_shared_device — 6 banks x 9 workers gave
  // 24.7 k over the whole run / 22.9 k in the steady state, 8 x 7 24.8 k / 20.3 k: 3.5 x the budget.)
  const int workers = banks * ranks == 1 ? budget - 1 : (7 * budget) / (2 * ranks * banks);
  return std::max(1, std::min(std::min(workers, n_sequences - 1), 31));
}

// the CPUs of a kernel list ("0-63,128-191")
static bool parse_cpulist(const char* s, cpu_set_t* out)
{
  CPU_ZERO(out);
  bool any = false;
  while (*s) {
    char* e = nullptr;
    const long a = strtol(s, &e, 10);
    if (e == s || a < 0) return false;
    long b = a;
    s = e;
    if (*s == '-') { b = strtol(s + 1, &e, 10); if (e == s + 1 || b < a) return false; s = e; }
    for (long c = a; c <= b && c < CPU_SETSIZE; c++) { CPU_SET((int)c, out); any = true; }
    if (*s == ',') s++; else if (*s) return false;
  }
  return any;
}

Bank::Bank(hso_gpu_ctx* ctx, bool owns_ctx, const hso_camera& cam, const Settings& cfg, int n_sequences)
    : ctx_(ctx), owns_ctx_(owns_ctx), cam_(cam), cfg_(cfg)
{
  // a constructor that throws runs no destructor: everything made so far is taken down here (and the context, when it is the bank's)
  auto undo = [&]() {
    if (ctx_) (void)hso_gpu_set_host_parallel(ctx_, nullptr, nullptr);
    delete pool_; pool_ = nullptr;
    for (Seq* s : seq_) { if (s->map >= 0) (void)hso_gpu_seqmap_destroy(ctx_, s->map); delete s; }
    for (StepData* d : step_) delete d;
    seq_.clear(); step_.clear();
    if (seed_table_ >= 0) (void)hso_gpu_seed_table_destroy(ctx_, seed_table_);
    seed_table_ = -1;
    if (owns_ctx_) hso_gpu_destroy(ctx_);
  };
  try {
    // Reprojector::initializeGrid (src/reprojector.cpp:53-75); the cell order the reference shuffles is the identity here
    if (cam.width <= 0 || cam.height <= 0 || cfg.max_fts <= 0) throw Refused("camera size and max_fts must be positive");
    cell_size_ = (int)floorf(std::sqrt((float)(cam.width * cam.height) / cfg.max_fts) * 0.6);
    if (cell_size_ < 1) throw Refused("max_fts is too large for the image: the reprojection grid's cell size would be zero");
    grid_cols_ = (int)std::ceil((double)cam.width / cell_size_);
    grid_rows_ = (int)std::ceil((double)cam.height / cell_size_);
    cell_order_.resize((size_t)grid_cols_ * grid_rows_);
    std::iota(cell_order_.begin(), cell_order_.end(), 0);
    sync_previous_ = cfg.sync_previous;
    px_error_angle_ = std::atan(1.0 / (2.0 * cam_.errorMultiplier2())) * 2.0;   // one pixel of noise (src/depth_filter.cpp:360-366)
    for (int k = 0; k < n_sequences; k++) {
      Seq* s = new Seq();
      s->index = k; s->cam = &cam_; s->cfg = &cfg_;
      seq_.push_back(s);
      step_.push_back(new StepData());
      check(hso_gpu_seqmap_create(ctx_, &s->map), "seqmap_create");
      // a keyframe's feature list: its own features (the first keyframe: up to the initialisation's 2000) + the features of its
      // seeds that became points (at most max_fts + 100 seeds per keyframe)
      check(hso_gpu_seqmap_configure(ctx_, s->map, std::max(2000, cfg_.max_fts) + cfg_.max_fts + 128), "seqmap_configure");
    }
    check(hso_gpu_seed_table_create(ctx_, &seed_table_), "seed_table_create");
    if (g_host_share.load() > 1) check(hso_gpu_set_shared_device(ctx_, 1), "set_shared_device");   // throughput shapes: the other banks fill the device
    n_threads_ = pool_threads_for(n_sequences);
    pool_ = new Pool(n_threads_);
    // the device library's own host loops (local-BA window staging, map-patch staging) run on this pool too: the engine's thread is
    // inside the library then and the workers are idle
    check(hso_gpu_set_host_parallel(ctx_, [](void* user, int n, void (*body)(void*, int), void* arg) {
      static_cast<Pool*>(user)->run(n, [&](int i) { body(arg, i); });
    }, pool_), "set_host_parallel");
    // where the device hangs (pin_threads, from the first step on)
    {
      char list[1024] = {0};
      cpu_set_t node, mine;
      if (hso_gpu_device_cpulist(ctx_, list, sizeof(list)) == HSO_OK && list[0] && parse_cpulist(list, &node) &&
          pthread_getaffinity_np(pthread_self(), sizeof(mine), &mine) == 0) {
        CPU_AND(&numa_cpus_, &node, &mine);
        numa_known_ = CPU_COUNT(&numa_cpus_) >= 2 && !CPU_EQUAL(&numa_cpus_, &mine);
      }
    }
  } catch (...) { undo(); throw; }
}

Bank::~Bank()
{
  if (timing_level() > 0 && n_steps_ > 0)
    fprintf(stderr, "[hso engine] %lld steps of %d sequences, %lld keyframes; ms per step: upload %.3f, track %.3f, reproject+select+pose %.3f, decide %.3f, "
            "local BA %.3f, seed observe %.3f, seed activate %.3f, new seeds %.3f, flush+finish %.3f\n", (long long)n_steps_, size(), (long long)n_kf_events_,
            phase_ms_[0] / n_steps_, phase_ms_[1] / n_steps_, phase_ms_[2] / n_steps_, phase_ms_[3] / n_steps_, phase_ms_[4] / n_steps_, phase_ms_[5] / n_steps_,
            phase_ms_[6] / n_steps_, phase_ms_[7] / n_steps_, phase_ms_[8] / n_steps_);
  // the same phases under the reference's timer names (g_permon, src/frame_handler_base.cpp:54-66; HSO_START_TIMER in
  // src/frame_handler_mono.cpp:90, 188, 215, 237, 316): the resident chain is ONE device call, so four of its timers are one figure
  if (timing_level() > 0 && n_steps_ > 0)
    fprintf(stderr, "[hso engine] reference timers, ms per step: pyramid_creation %.3f | sparse_img_align + reproject (reproject_kfs, reproject_candidates, "
            "feature_align) + pose_optimizer %.3f | local_ba %.3f | (depth filter: seeds observed %.3f, point_optimizer / activation %.3f, new seeds %.3f) | tot_time %.3f\n",
            phase_ms_[0] / n_steps_, (phase_ms_[1] + phase_ms_[2]) / n_steps_, phase_ms_[4] / n_steps_, phase_ms_[5] / n_steps_, phase_ms_[6] / n_steps_, phase_ms_[7] / n_steps_,
            (phase_ms_[0] + phase_ms_[1] + phase_ms_[2] + phase_ms_[3] + phase_ms_[4] + phase_ms_[5] + phase_ms_[6] + phase_ms_[7] + phase_ms_[8]) / n_steps_);
  if (timing_level() > 0 && n_steps_ > 0) {
    static const char* const names[9] = {"upload", "track", "reproject+select+pose", "decide", "local BA", "seed observe", "seed activate", "new seeds", "flush+finish"};
    for (int k = 0; k < 9; k++)
      fprintf(stderr, "[hso engine]   %-22s per step: %6.1f copies (%5.1f staged, %8.1f KB of which %8.1f KB to the device), %5.1f syncs (%.3f ms blocked), %5.1f memsets\n", names[k],
              (double)phase_census_[k][0] / n_steps_, (double)phase_census_[k][2] / n_steps_, (double)phase_census_[k][1] / n_steps_ / 1024.0, (double)phase_census_[k][6] / n_steps_ / 1024.0,
              (double)phase_census_[k][3] / n_steps_, (double)phase_census_[k][4] / n_steps_ * 1e-6, (double)phase_census_[k][5] / n_steps_);
  }
  if (timing_level() > 0 && n_steps_ > 0)
    fprintf(stderr, "[hso engine] reproject+select+pose = list points + patch maps %.3f, device call %.3f, apply %.3f\n", sub_ms_[0] / n_steps_, sub_ms_[1] / n_steps_,
            sub_ms_[2] / n_steps_);
  if (timing_ && n_steps_ > 0) {
    fprintf(stderr, "[hso engine] sections, ms per step:");
    for (const auto& e : sections_) fprintf(stderr, " %s %.3f,", e.first, e.second / n_steps_);
    fprintf(stderr, "\n");
  }
  (void)hso_gpu_set_host_parallel(ctx_, nullptr, nullptr);   // before the pool goes
  delete pool_;
  if (seed_table_ >= 0) (void)hso_gpu_seed_table_destroy(ctx_, seed_table_);   // before the frames its seeds are hosted in (waits for a pass in flight)
  for (int64_t id : after_prev_release_) (void)hso_gpu_frame_release(ctx_, id);
  for (int64_t id : to_release_) (void)hso_gpu_frame_release(ctx_, id);
  for (Seq* s : seq_) {
    if (s->map >= 0) (void)hso_gpu_seqmap_destroy(ctx_, s->map);
    for (Frame& F : s->frames) if (F.in_use && F.dev_id >= 0) (void)hso_gpu_frame_release(ctx_, F.dev_id);
    delete s;
  }
  for (StepData* d : step_) delete d;
  chain_res_.release(); chain_brief_.release(); feat_rows_.release(); track_tables_.release(); act_seeds_.release(); act_targets_.release(); act_ints_.release(); act_slots_.release(); act_out_.release(); seed_brief_.release(); seed_px_.release(); det_corners_.release(); det_fill_.release(); det_edgelets_.release();   // before the context goes
  if (owns_ctx_) hso_gpu_destroy(ctx_);
}

void Bank::check(int rc, const char* what)
{
  if (rc < 0) throw DeviceFault(std::string(what) + ": " + hso_gpu_last_error(ctx_));
}

void Bank::par(const std::vector<int>& who, const std::function<void(int)>& fn)
{
  pool_->run((int)who.size(), [&](int i) { fn(who[i]); });
}

bool Bank::trace(int k, const char* path)
{
  if (k < 0 || k >= size()) return false;
  if (!path) { seq_[k]->trace.close(); return true; }
  return seq_[k]->trace.open(path);
}

bool Bank::trace_state(int k, bool on)
{
  if (k < 0 || k >= size()) return false;
  seq_[k]->trace.state = on;
  return true;
}

// Keep the bank's threads on the NUMA node the device is attached to: the page-locked staging the library allocates on their
// behalf, the runtime's queues and the doorbells are then node-local.  On the two-socket GPU boxes (256 hardware threads, 16-CPU
// quota) six engines measured 29.5-29.9 k frames/s steady with it and 25.2-29.4 k without, on one box, alternating
// (profiles/r6_engine_host.md section 7).  The node's CPUs are intersected with the affinity the thread already has (a cpuset or a
// taskset of the caller's stays in force); an unknown node, or an intersection of fewer than two CPUs, leaves everything alone.
void Bank::pin_threads()
{
  if (no_numa_pin_ || !numa_known_) return;
  if (!workers_pinned_) { if (pool_) pool_->pin(numa_cpus_); workers_pinned_ = true; }
  const pthread_t me = pthread_self();
  if (driver_pinned_ && pthread_equal(me, pinned_driver_)) return;
  (void)pthread_setaffinity_np(me, sizeof(numa_cpus_), &numa_cpus_);
  pinned_driver_ = me; driver_pinned_ = true;
}

void Bank::set_options(bool sync_previous, bool track_no_coop, bool no_numa_pin)
{
  previous_collect();                                            // a pass in flight is applied under the old setting
  sync_previous_ = sync_previous;
  no_numa_pin_ = no_numa_pin;
  hso_gpu_options o{};
  o.size = (int32_t)sizeof(o); o.track_no_coop = track_no_coop ? 1 : 0;
  check(hso_gpu_configure(ctx_, &o), "configure");
}

void Bank::call_counts(int64_t* calls, int64_t* items, int cap) const
{
  for (int i = 0; i < cap && i < 10; i++) { if (calls) calls[i] = n_calls_[i]; if (items) items[i] = n_items_[i]; }
}

void Bank::status(int k, hso_vo_status* st) const
{
  const Seq& s = *seq_[k];
  *st = s.log;
  st->stage = s.stage; st->tracking_quality = s.quality; st->result = s.outcome;
  st->n_keyframes = (int)s.kfs.size();
  if (s.last != kNone) {
    const Frame& F = s.frames[s.last];
    st->T_f_w = F.T.v; st->timestamp = F.stamp; st->exposure_time = F.exposure;
    st->frame_id = F.serial; st->keyframe_id = F.kf_id; st->is_keyframe = F.kf_row >= 0 ? 1 : 0;
    st->n_features = (int)s.n_feats(F); st->n_inliers = F.n_inliers;
  }
}

int Bank::keyframes(int k, double* stamps, hso_se3* T_f_w, int32_t* frame_ids, int cap) const
{
  const Seq& s = *seq_[k];
  int n = 0;
  for (Id kf : s.kfs) {
    if (n < cap) {
      const Frame& F = s.frames[kf];
      if (stamps) stamps[n] = F.stamp;
      if (T_f_w) T_f_w[n] = F.T.v;
      if (frame_ids) frame_ids[n] = F.serial;
    }
    ++n;
  }
  return n;
}

int Bank::trajectory(int k, double* stamps, hso_se3* T_f_w, int cap) const
{
  const Seq& s = *seq_[k];
  const int n = (int)s.hist_pose.size();
  for (int i = 0; i < n && i < cap; i++) { if (stamps) stamps[i] = s.hist_stamp[i]; if (T_f_w) T_f_w[i] = s.hist_pose[i]; }
  return n;
}

// the frames the sequences dropped since the last call leave the device: one call, one wait
void Bank::release_queued()
{
  if (to_release_.empty()) return;
  std::sort(to_release_.begin(), to_release_.end());
  to_release_.erase(std::unique(to_release_.begin(), to_release_.end()), to_release_.end());
  std::vector<int64_t> again;
  if (hso_gpu_frame_release_batch(ctx_, to_release_.data(), (int)to_release_.size()) < 0)
    for (int64_t id : to_release_)                                  // one of them cannot go (yet): the others still do, and
      if (hso_gpu_frame_release(ctx_, id) == HSO_E_INVALID) again.push_back(id);   // a frame still hosting live seeds stays queued for the next step
  to_release_.swap(again);
}

void Bank::release_frame(Seq& s, Id fr)
{
  if (fr == kNone) return;
  const int64_t dev = s.frames[fr].dev_id;
  if (s.drop(fr)) to_release_.push_back(dev);
}

}  // namespace engine
}  // namespace hso
