/*
 * hso_vo.h — C interface of the host driver (libhso_host.so): the reference's FrameHandlerMono with its
 * addImage() entry (include/hso/frame_handler_mono.h:43-50, src/frame_handler_mono.cpp:80-123), as a maintainer's
 * harness, a language binding or `python -m hso_amd.run_sequence` drives it.  Behind it sits the sequence engine
 * (hso_amd/host/hso_engine.h): per-sequence state as index-linked tables mirrored on the device, every numeric step a
 * batched call into the device library (include/hso_gpu.h), the reference's decisions (keyframe choice, covisibility,
 * reprojection order, seed / candidate life cycle) reproduced from its call order and argument values.  A single handle is
 * an engine of one sequence.  What test/test_dataset.cpp does with the class (:264-286, :312-335) maps to:
 *   new FrameHandlerMono(cam, false)          hso_vo_create
 *   vo->addImage(img, id, &stamp)             hso_vo_add_image
 *   vo->lastFrame(), map_.keyframes_          hso_vo_get_status, hso_vo_get_keyframes
 *   vo->start()                               hso_vo_start
 * A sequence starts either like the reference's harness — hso_vo_start, then images: the two-view initialisation
 * (src/initialization.cpp; pyramidal KLT on the device, essential matrix / homography on the host, see
 * hso_amd/host/hso_init.h for what replaces its OpenCV calls) builds the first map once the median disparity reaches
 * Config::initMinDisparity — or from hso_vo_set_first_frame, the setFirstFrame hook the reference keeps for synthetic data
 * (frame_handler_mono.h:49-50, .cpp:419-426), with a depth image for the first keyframe.
 */
#ifndef HSO_VO_H
#define HSO_VO_H
#include "hso_gpu.h"
#ifdef __cplusplus
extern "C" {
#endif

typedef struct hso_vo hso_vo;

/* max_fts = Config::maxFts() (200 by default in the reference; it sizes the reprojection grid and the detector budget) */
int hso_vo_create(hso_vo** out, const hso_camera* cam, int max_fts, int device);
void hso_vo_destroy(hso_vo* vo);
const char* hso_vo_last_error(const hso_vo* vo);
/* record every device call of the driver (inputs and outputs in the C-ABI's table layouts) to `path`; NULL stops */
int hso_vo_trace(hso_vo* vo, const char* path);
/* on != 0: while recording, every per-frame chain call (hso_gpu_seq_chain) is preceded by a "seq_chain_state" record — the job, the
 * call's configuration and the sequence map exactly as the device holds it at that moment (hso_gpu_seqmap_debug_dump,
 * include/hso_gpu_debug.h) — and followed by a "seq_chain_result" record (the raw result, every event, the frame's feature table).
 * With it a test hands ONE map state to the device call and to its CPU restatement (tests/test_seq_chain.py).  A few MB per frame
 * at 2000 features: meant for a handful of frames, not for whole runs. */
int hso_vo_trace_state(hso_vo* vo, int on);
/* Choices that were process-wide environment variables until round 5, per handle; zero = default.  May be set between frames. */
typedef struct hso_vo_options {
  int32_t size;               /* sizeof(hso_vo_options) */
  int32_t sync_previous;      /* 1: the depth filter's idle-time pass runs inside the step instead of on its own stream (same results: tests compare) */
  int32_t track_no_coop;      /* 1: the tracker keeps one workgroup per job whatever the batch size (hso_gpu_options.track_no_coop) */
  int32_t no_numa_pin;        /* 0 (default): from a handle's next step on, the thread that drives it and its worker threads run on the CPUs of the NUMA node
                                 the device is attached to (hso_gpu_device_cpulist, intersected with the affinity the process has) — page-locked
                                 staging and the runtime's queues stay node-local; 1: affinities are left alone (set before the first frame) */
  int32_t reserved[4];
} hso_vo_options;
int hso_vo_set_options(hso_vo* vo, const hso_vo_options* options);
/* first keyframe: features are detected the way the initialisation detects them and every feature with
 * depth_z[y * width + x] > 0 (depth along the optical axis, metres) becomes a map point hosted in this frame */
int hso_vo_set_first_frame(hso_vo* vo, const uint8_t* img, int width, int height, double timestamp, const float* depth_z,
                           const hso_se3* T_f_w /* NULL = identity */);
/* FrameHandlerBase::start(): the next image is the first frame of the two-view initialisation (stage 1, then 2 until the
 * disparity suffices, then 3).  hso_vo_status.result is 1 for the frames that became keyframes, 0 while the disparity is short,
 * 2 when it failed (too few tracked points / inliers: the handle is paused again, call hso_vo_start to retry). */
int hso_vo_start(hso_vo* vo);
/* FrameHandlerMono::addImage.  HSO_E_INVALID with hso_vo_last_error() where the reference throws (wrong image size). */
int hso_vo_add_image(hso_vo* vo, const uint8_t* img, int width, int height, double timestamp);

typedef struct hso_vo_status {
  hso_se3 T_f_w;              /* lastFrame()->T_f_w_ */
  double timestamp, exposure_time;
  int32_t frame_id, keyframe_id, is_keyframe;
  int32_t stage;              /* FrameHandlerBase::Stage: 0 paused, 1 first, 2 second, 3 default, 4 relocalizing */
  int32_t tracking_quality;   /* 0 insufficient, 1 bad, 2 good */
  int32_t result;             /* UpdateResult: 0 no keyframe, 1 keyframe, 2 failure */
  int32_t n_features, n_inliers, n_tracked, n_matches, n_trials, n_seed_matches, n_seeds, n_candidates, n_keyframes, used_inverse;
  int32_t ba_removed_1, ba_removed_2;
  double pose_error_init, pose_error_final, ba_error_init, ba_error_final;
} hso_vo_status;
int hso_vo_get_status(hso_vo* vo, hso_vo_status* st);
/* map_.keyframes_ in list order (what BenchmarkNode::saveResult writes): returns the number of keyframes, fills at most cap */
int hso_vo_get_keyframes(hso_vo* vo, double* timestamps, hso_se3* T_f_w, int32_t* frame_ids, int cap);

/* every frame processed since the sequence started: (timestamp, T_f_w when the frame was finished) — the per-frame trajectory a
 * harness gathers (BASELINE configs[4]); returns the number of frames, fills at most cap */
int hso_vo_get_trajectory(hso_vo* vo, double* timestamps, hso_se3* T_f_w, int cap);

/* ---- N independent sequences over one device context, in lockstep (hso_amd/host/hso_engine*.cpp): BASELINE north_star
 * "independent sequences ... batched"; configs[4] = what test/euroc_batch.sh:9-18 runs one after the other.  One step takes one
 * image per sequence and runs every stage ONCE for all of them: hso_gpu_frame_upload_batch, hso_gpu_seq_chain (the tracker's reference tables,
 * CoarseTracker::run for N jobs, projection + matching + grid selection + pose optimisation and the keyframe decision's inputs
 * on the sequences' resident maps), hso_gpu_ba_local_multi (Huber deltas + optimisation of all windows in one call) for the sequences that take a keyframe,
 * hso_gpu_seed_table_observe_groups, hso_gpu_seed_activate_multi, hso_gpu_detect_candidates; the depth filter's idle-time pass
 * (hso_gpu_seed_table_observe_previous_begin / _end) runs on its own stream beside the next step's tracking.  A sequence run
 * among up to 8 equals the same sequence run alone through hso_vo_* bit for bit (tests/test_multi_gpu.py); in larger banks the
 * tracker splits a job over a number of workgroups that depends on the batch size, and the results agree within the tracker's
 * stated tolerance instead.  Several engines ("banks") may run in one process on their own threads and contexts;
 * that is how one GPU is kept busy (bench.py: sequences). */
typedef struct hso_vo_multi hso_vo_multi;
int hso_vo_multi_create(hso_vo_multi** out, const hso_camera* cam, int max_fts, int n_sequences, int device);
void hso_vo_multi_destroy(hso_vo_multi* m);
const char* hso_vo_multi_last_error(const hso_vo_multi* m);
int hso_vo_multi_size(const hso_vo_multi* m);
/* arrays of n_sequences entries; a NULL image = the sequence sits this step out (sequences need not have equal lengths) */
int hso_vo_multi_set_first_frames(hso_vo_multi* m, const uint8_t* const* imgs, int width, int height, const double* timestamps,
                                  const float* const* depth_z, const hso_se3* T_f_w /* n_sequences or NULL */);
/* hso_vo_start for every sequence (which = NULL) or for the sequences with which[k] != 0: their next images run the two-view
 * initialisation; its KLT call has no multi-sequence form and is serialised with the other sequences' device calls */
int hso_vo_multi_start(hso_vo_multi* m, const uint8_t* which);
int hso_vo_multi_add_images(hso_vo_multi* m, const uint8_t* const* imgs, int width, int height, const double* timestamps);
/* the same with images that already live in device memory (imgs[k] = device pointer to width * height bytes): a capture or
 * decode pipeline that ends on the GPU, and the form throughput is measured with (inputs resident in HBM) */
int hso_vo_multi_add_images_device(hso_vo_multi* m, const uint8_t* const* imgs, int width, int height, const double* timestamps);
/* hso_vo_trace for one sequence of the bank */
int hso_vo_multi_trace(hso_vo_multi* m, int sequence, const char* path);
int hso_vo_multi_trace_state(hso_vo_multi* m, int sequence, int on);
int hso_vo_multi_set_options(hso_vo_multi* m, const hso_vo_options* options);
int hso_vo_multi_get_status(hso_vo_multi* m, int sequence, hso_vo_status* st);
int hso_vo_multi_get_keyframes(hso_vo_multi* m, int sequence, double* timestamps, hso_se3* T_f_w, int32_t* frame_ids, int cap);
int hso_vo_multi_get_trajectory(hso_vo_multi* m, int sequence, double* timestamps, hso_se3* T_f_w, int cap);
/* batched C-ABI calls issued so far and the per-sequence requests they carried, per kind: [0] frame upload, [1] frame release,
 * [2] tracker, [3] reprojection + matching, [4] matching alone, [5] pose, [6] seed observation, [7] seed activation, [8] local BA,
 * [9] calls without a multi-sequence form.  Returns the number of kinds. */
int hso_vo_multi_call_counts(hso_vo_multi* m, int64_t* calls, int64_t* items, int cap);
/* The host side of a bank is a small thread pool (per-sequence bookkeeping between the device calls).  A process that runs several
 * banks side by side (one thread each, independent sequences shard freely within a GPU too) says so BEFORE it creates them: every
 * bank then sizes its pool to max(1, (7 * cpu quota) / (2 * LOCAL_WORLD_SIZE * banks_in_process)) workers (3.5 x oversubscribed: most
 * workers of a bank sleep while its device call runs; a lone bank: quota - 1) and shares the device — the CPUs the process may keep
 * busy (hardware threads, or its cgroup's cpu.max) shared among the ranks of a node (LOCAL_WORLD_SIZE of the launcher) and its own
 * banks.  HSO_ENGINE_THREADS overrides.  hso_vo_multi_threads: the workers a bank got; hso_vo_host_cpu_quota: the quota. */
int hso_vo_host_share(int banks_in_process);
/* measurement aid: the algorithmic bytes (SURVEY.md section 8(d) units) of the steps so far, by stage: out[0] frame construction,
 * [1] CoarseTracker (n_eval x B_alg + B_pre + B_sel per level, from the evaluation counts the kernel reports), [2] the matcher
 * (per listed point), [3] the pose optimiser, [4] seed observation.  Returns 5. */
int hso_vo_multi_alg_bytes(const hso_vo_multi* m, double* out, int cap);
int hso_vo_multi_threads(const hso_vo_multi* m);
int hso_vo_host_cpu_quota(void);

/* initialization::computeInitializeMatrix (src/initialization.cpp:300-385) alone, for tests and tools: n unit bearings per frame
 * (3 doubles each) -> T_cur_from_ref, the inlier indices (at most cap; returns their number), the triangulated points in the
 * current frame (n x 3, unscaled: |t| = 1) and which model won (0 essential, 1 homography).  Pure host code: no device call. */
int hso_vo_init_compute_matrix(const double* f_ref, const double* f_cur, int n, double focal_length, double reproj_thresh, hso_se3* T_cur_from_ref,
                            int32_t* inliers, int cap, double* xyz_in_cur, int32_t* used_homography);

/* ---- The path's only exchange, native (hso_amd/host/hso_gather.cpp, libhso_gather.so): one ncclAllGather (RCCL over xGMI)
 * of the per-frame records of all ranks — BASELINE.json north_star "RCCL over xGMI used only to gather results", configs[4]:
 * what a C++ harness calls after hso_vo_multi_get_trajectory where bench.py uses torch.distributed.  One process per GPU.
 * Rank 0 makes the communicator id and the launcher's own channel (environment, file, MPI, a socket) carries its 128 bytes to
 * the other ranks; every rank then creates its handle (collective: returns when all `world` ranks have called).
 * hso_gather_records: every rank passes n_rows records of HSO_GATHER_RECORD doubles (quaternion x y z w, translation, time
 * stamp or exposure ratio: the row hso_amd/dist.py packs; same n_rows on every rank, short sequences padded with NaN rows);
 * `all` receives [world][n_rows][HSO_GATHER_RECORD] in rank order on every rank.  Host pointers; the staging through HBM is
 * the library's.  Errors: HSO_E_INVALID (arguments), HSO_E_HIP (HIP or RCCL failure, text in hso_gather_last_error). */
#define HSO_GATHER_ID_BYTES 128
#define HSO_GATHER_RECORD 8
typedef struct hso_gather hso_gather;
int hso_gather_unique_id(uint8_t id[HSO_GATHER_ID_BYTES]);
int hso_gather_create(hso_gather** out, const uint8_t id[HSO_GATHER_ID_BYTES], int rank, int world, int device);
void hso_gather_destroy(hso_gather* g);
int hso_gather_size(const hso_gather* g);
int hso_gather_rank(const hso_gather* g);
int hso_gather_records(hso_gather* g, const double* mine, int n_rows, double* all);
const char* hso_gather_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
