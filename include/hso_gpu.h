/*
 * hso_gpu.h — C-ABI of the MI355X-native HSO per-frame numeric hot path.
 *
 * Plain C, POD structs, caller-owned buffers, no exceptions, no callbacks.
 * Every entry point returns an int status: >= 0 ok, < 0 error (HSO_E_*);
 * hso_gpu_last_error() gives a human-readable reason for the last failure on
 * the context.  One hso_gpu_ctx per host thread / HIP stream.
 *
 * The reference (luodongting/HSO) has no FFI layer: its boundary is the C++
 * call surface FrameHandlerMono::processFrame() uses.  Each entry point below
 * names the reference interface (file:line, relative to the reference root)
 * whose body it replaces; the C++ adapters with the reference's own class and
 * function names live in hso_amd/host/ and call only this header.
 *
 * Conventions
 *   - poses are Sophus-style SE3: unit quaternion (x,y,z,w) + translation,
 *     tangent order [upsilon(0:3), omega(3:6)]
 *     (thirdparty/Sophus/sophus/se3.cpp:170-196);
 *   - images are 8-bit, row-major, stride == width
 *     (the reference assumes stride == cols, src/CoarseTracker.cpp:248);
 *   - feature arrays are in Frame::fts_ list order (include/hso/frame.h:86);
 *     an index into them is "the feature index" parity tests compare.
 */
#ifndef HSO_GPU_H
#define HSO_GPU_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HSO_GPU_ABI_VERSION 2
#define HSO_N_PYR_LEVELS 5   /* max(n_pyr_levels=3, klt_max_level+1=5), src/frame.cpp:92 */
#define HSO_N_SOBEL_LEVELS 3 /* Config::nPyrLevels(), src/frame.cpp:214 */

enum {
  HSO_OK = 0,
  HSO_E_INVALID = -1,   /* bad argument (null pointer, bad size, bad level) */
  HSO_E_NOFRAME = -2,   /* frame id not resident in the context */
  HSO_E_HIP = -3,       /* HIP runtime error, see hso_gpu_last_error */
  HSO_E_NOMEM = -4,
  HSO_E_UNSUPPORTED = -5
};

/* ---- camera: AbstractCamera {Pinhole, FOV, Equidistant}, src/camera.cpp ---- */
enum { HSO_CAM_PINHOLE = 0, HSO_CAM_FOV = 1, HSO_CAM_EQUIDISTANT = 2 };

typedef struct hso_camera {
  int32_t model;       /* HSO_CAM_* */
  int32_t width, height;
  int32_t distortion;  /* pinhole: |d0| > 1e-7 (camera.cpp:37); FOV: !undistort_ (camera.cpp:208) */
  double fx, fy, cx, cy;
  double d[5];         /* pinhole radtan k1,k2,p1,p2,k3 (camera.cpp:105-120); FOV: d[0] = omega */
} hso_camera;

typedef struct hso_se3 {
  double q[4]; /* x, y, z, w */
  double t[3];
} hso_se3;

/* ---- frames: Frame::initFrame + createImgPyramid + prepareForFeatureDetect,
 *      src/frame.cpp:82-96,205-246,296-314 ---- */
typedef struct hso_frame_stats {
  float integral_image; /* Frame::integralImage_ : mean level-0 intensity, 16 px margin */
  float grad_mean;      /* Frame::gradMean_ : mean |sobel5| / 30 clamped to [7,20]      */
  int32_t width, height;
} hso_frame_stats;

typedef struct hso_gpu_ctx hso_gpu_ctx;

/* stream: a hipStream_t (as void*) the context launches on, or NULL for the
 * default stream.  device: HIP device ordinal. */
int hso_gpu_create(hso_gpu_ctx** out, int device, void* stream);
void hso_gpu_destroy(hso_gpu_ctx* ctx);
const char* hso_gpu_last_error(const hso_gpu_ctx* ctx);
int hso_gpu_abi_version(void);
int hso_gpu_synchronize(hso_gpu_ctx* ctx);
/* shared != 0: other contexts of the process keep the device busy beside this one (several banks of sequences per GPU, each on its
 * own stream).  Batches smaller than the chip then stay on the one-workgroup-per-job kernel shapes instead of being split over the
 * idle CUs — the other contexts' work fills those — which is what gives the best THROUGHPUT; the default (0) gives a lone batch
 * the best LATENCY.  Results agree within the tracker's stated tolerance either way. */
int hso_gpu_set_shared_device(hso_gpu_ctx* ctx, int shared);
/* The CPUs of the NUMA node the context's device is attached to, in the kernel's list format ("0-63,128-191"); an empty string when
 * the node is unknown.  A driver whose threads stay there keeps its page-locked staging memory, the runtime's queues and the
 * device's doorbells on one node (the sequence engine does: hso_vo_options.no_numa_pin). */
int hso_gpu_device_cpulist(hso_gpu_ctx* ctx, char* out, size_t cap);
/* The remaining choices of kernel shape and of how a context waits, per context (until round 5: process-wide environment
 * variables).  Zero = the default everywhere; the struct may be extended at its end (size = sizeof of the caller's). */
enum { HSO_WAIT_DEFAULT = 0,   /* a lone context polls (hipStreamSynchronize); one that shares its device naps between event queries */
       HSO_WAIT_POLL = 1, HSO_WAIT_NAP = 2, HSO_WAIT_BLOCK = 3 /* a blocking event: lowest CPU use, wake-ups measured bimodal on the GPU boxes */ };
typedef struct hso_gpu_options {
  int32_t size;                /* sizeof(hso_gpu_options) */
  int32_t wait_mode;           /* HSO_WAIT_* */
  int32_t track_no_coop;       /* 1: batches smaller than the chip stay on the one-workgroup-per-job tracker (a job's result then does
                                  not depend on the batch it is in; the cooperative shape agrees within the tracker's tolerance) */
  int32_t track_coop_scatter;  /* 1: a job's workgroups spread over all XCDs (the placement-independent exchange; same bits) */
  int32_t track_coop_feats_per_wg;   /* features per workgroup the cooperative shape aims for (0: 256) */
  int32_t track_coop_workgroups;     /* workgroups per job of the cooperative shape (0: from the feature count) */
  int32_t reserved[2];
} hso_gpu_options;
int hso_gpu_configure(hso_gpu_ctx* ctx, const hso_gpu_options* options);
/* The host side of the batched entry points (staging tens of megabytes of local-BA windows, checking and staging map patches) is a
 * loop over independent items.  A caller that keeps a worker pool of its own lends it here: the library then calls
 * parallel_for(user, n, body, arg) and expects body(arg, i) to have run for every i in [0, n), on any threads, when it returns;
 * it does so only from inside an entry point called with this context, on the calling thread, for loops above ~1 MB of work.
 * body never touches the HIP runtime.  parallel_for == NULL takes the pool away (the loops run on the calling thread). */
typedef void (*hso_parallel_for_fn)(void* user, int n, void (*body)(void* arg, int i), void* arg);
int hso_gpu_set_host_parallel(hso_gpu_ctx* ctx, hso_parallel_for_fn parallel_for, void* user);
/* Page-locked host memory for the tables a caller hands to / receives from the entry points.  Every entry point accepts any host
 * pointer; result and input tables that live in memory from this allocator are DMA targets / sources as they are (tens of GB/s),
 * pageable memory goes through the runtime's staging copies (and its first-touch page faults) at a fraction of that — with tens
 * of megabytes of match / seed records per multi-sequence step, the difference is several milliseconds.
 * Freed by hso_gpu_host_free or with the context. */
int hso_gpu_host_alloc(hso_gpu_ctx* ctx, size_t bytes, void** out);
int hso_gpu_host_free(hso_gpu_ctx* ctx, void* p);

/* Replaces `new Frame(cam, img, ts)` -> Frame::initFrame (src/frame.cpp:82-96):
 * builds the 5-level u8 pyramid (halfSample, src/vikit/vision.cpp:19-108; the
 * per-level choice between the SSE2 rounding and the scalar truncation follows
 * vision.cpp:76), the 5x5 Sobel images of levels 0-2 and the two frame
 * statistics, all on the device.  `img` is a HOST pointer (img_is_device = 0)
 * or a DEVICE pointer (img_is_device = 1) to width*height bytes.
 * Sizes that are not multiples of 16 in both dimensions take the cv::resize branch of
 * createImgPyramid (src/frame.cpp:307-312; e.g. TUM-mono's 920x736): level sizes are
 * cvRound(size * 2^-L) and the levels come from OpenCV's INTER_LINEAR restated.
 * Errors: HSO_E_INVALID if width % 4 != 0, the image is smaller than 64x64 or the
 * frame id is already resident. */
int hso_gpu_frame_upload(hso_gpu_ctx* ctx, int64_t frame_id, const uint8_t* img,
                         int width, int height, int img_is_device,
                         hso_frame_stats* stats_out);
/* The same for a sensor image larger than the camera model: ImageReader::readImage's
 * cv::resize(image, image, Size(cam.width, cam.height)) (src/ImageReader.cpp:79; the calibration
 * loader shrinks anything above 848x800, test/test_dataset.cpp:162-173, e.g. TUM-mono 1280x1024 ->
 * 920x736) runs on the device — OpenCV's INTER_LINEAR for CV_8UC1 restated, bit-exact with the
 * oracle — and the result becomes level 0.  src == dst size: plain hso_gpu_frame_upload. */
int hso_gpu_frame_upload_resized(hso_gpu_ctx* ctx, int64_t frame_id, const uint8_t* img, int src_width, int src_height,
                                 int width, int height, int img_is_device, hso_frame_stats* stats_out);
/* Batched form: n frames of one size in three launches.  imgs[i] are n HOST
 * pointers or n DEVICE pointers (img_is_device).  A frame id that is already
 * resident is refreshed in place (same size required) — the batched analogue of
 * constructing the next Frame of each of n independent sequences.  With device
 * images and stats_out == NULL the call is asynchronous on the context stream
 * (hipMemcpyAsync of the pointer tables from pageable memory returns after the
 * staging copy, so the host arrays may be reused). */
int hso_gpu_frame_upload_batch(hso_gpu_ctx* ctx, const int64_t* frame_ids,
                               const uint8_t* const* imgs, int n, int width, int height,
                               int img_is_device, hso_frame_stats* stats_out /* n or NULL */);
int hso_gpu_frame_release(hso_gpu_ctx* ctx, int64_t frame_id);
/* n frames with one wait for the stream (~Frame of the frames n sequences dropped in a step); all or nothing: HSO_E_NOFRAME /
 * HSO_E_INVALID (a frame hosts live seeds of a resident table) leave every frame resident. */
int hso_gpu_frame_release_batch(hso_gpu_ctx* ctx, const int64_t* frame_ids, int n);
/* parity/debug read-back: level in [0,5); out has (w>>level)*(h>>level) bytes */
int hso_gpu_frame_download_level(hso_gpu_ctx* ctx, int64_t frame_id, int level,
                                 uint8_t* out, int* w_out, int* h_out);
/* level in [0,3); gx/gy each (w>>level)*(h>>level) int16 */
int hso_gpu_frame_download_sobel(hso_gpu_ctx* ctx, int64_t frame_id, int level,
                                 int16_t* gx, int16_t* gy);

/* ---- CoarseTracker, src/CoarseTracker.cpp, include/hso/CoarseTracker.h ---- */

/* One reference-frame feature, flattened from Feature{px,f} + the distance
 * CoarseTracker::makeDepthRef computes (CoarseTracker.cpp:210-240).
 * dist < 0  <=>  feature has no point or the point is behind the reference
 * camera; such a slot produces no residual terms but keeps its index. */
typedef struct hso_ref_feat {
  double px[2]; /* Feature::px, level-0 pixels (include/hso/feature.h:40) */
  double f[3];  /* Feature::f, unit bearing (feature.h:41) */
  double dist;  /* m_pt_ref[i] or -1 */
} hso_ref_feat;

/* Inputs of makeDepthRef for one feature (CoarseTracker.cpp:214-236). */
typedef struct hso_depth_ref_in {
  int32_t has_point;   /* (*it_ft)->point != NULL */
  int32_t host_pose;   /* index into the pose table: point->hostFeature_->frame->T_f_w_ */
  double host_f[3];    /* point->hostFeature_->f */
  double idist;        /* point->idist_ */
} hso_depth_ref_in;

/* dist_out[i] = | T_ref_w * T_host_w^-1 * (host_f / idist) | or -1. */
int hso_gpu_make_depth_ref(hso_gpu_ctx* ctx, const hso_depth_ref_in* in, int n,
                           const hso_se3* poses_f_w, int n_poses,
                           const hso_se3* T_ref_w, double* dist_out);

typedef struct hso_track_params {
  int32_t inverse_composition; /* CoarseTracker ctor arg 1 */
  int32_t max_level;           /* arg 2 (Config::kltMaxLevel() = 4) */
  int32_t min_level;           /* arg 3 (kltMinLevel()+1 = 1; 0 when relocalising) */
  int32_t n_iter;              /* arg 4 (50; 15 when relocalising) */
} hso_track_params;

typedef struct hso_track_job {
  int64_t ref_frame_id;
  int64_t cur_frame_id;
  const hso_ref_feat* feats; /* host pointer, n entries */
  int32_t n_feats;
  int32_t feats_soa;         /* 0: `feats` are n_feats hso_ref_feat records.  1: `feats` points to the kernel's own layout — six
                                arrays px[0] | px[1] | f[0] | f[1] | f[2] | dist of ((n_feats + 31) & ~31) doubles each (pad entries 0)
                                — which a caller that builds tables for many sequences in parallel writes directly; page-locked
                                tables of consecutive jobs laid out back to back leave in one DMA */
  hso_se3 T_cur_ref;         /* cur.T_f_w_ * ref.T_f_w_^-1 (CoarseTracker.cpp:63) */
  float exposure_rat;        /* cur.integralImage_/ref.integralImage_ (CoarseTracker.cpp:60) */
  float _pad2;
} hso_track_job;

#define HSO_TRACK_MAX_ITER 64
typedef struct hso_track_result {
  hso_se3 T_cur_ref;      /* m_T_cur_ref after the last level */
  float exposure_rat;     /* m_exposure_rat */
  int32_t n_tracked;      /* size_t(float(m_total_terms)/PATCH_AREA) of the last evaluation */
  int32_t n_terms_last, n_saturated_last;
  /* per pyramid level (index = level) */
  int32_t iters[HSO_N_PYR_LEVELS];      /* LM iterations executed */
  int32_t n_eval[HSO_N_PYR_LEVELS];     /* computeResiduals calls (1 + iters) */
  uint64_t accept_mask[HSO_N_PYR_LEVELS]; /* bit i = iteration i accepted */
  float huber[HSO_N_PYR_LEVELS];        /* m_huber_thresh */
  float outlier[HSO_N_PYR_LEVELS];      /* m_outlier_thresh */
  int32_t n_select[HSO_N_PYR_LEVELS];   /* errors.size() in selectRobustFunctionLevel */
  double energy[HSO_N_PYR_LEVELS];      /* final energy_old of the level */
  /* shader-clock cycles the owning workgroup spent per phase, summed over levels:
   * [0] stage image to LDS + precompute reference patches, [1] robust thresholds,
   * [2] residual/Jacobian evaluations incl. reductions, [3] LM solve + SE3 update,
   * [4] whole job; [5..9] inside the evaluations: projection, pixel loop, expansion,
   * wave exchange, workgroup combine.  Filled only when built with -DHSO_PHASE_TIMERS. */
  uint64_t phase_cycles[10];
  int32_t status;                       /* 0 ok */
  int16_t coop_workgroups;              /* workgroups that shared this job (0: the one-workgroup batch shapes) */
  int16_t coop_same_xcd;                /* 1: they all ran on one XCD and exchanged through its L2 (diagnostic; results do not depend on it) */
} hso_track_result;

/* Replaces `CoarseTracker(inverse, max_level, min_level, n_iter, verbose)
 * .run(ref, cur)` (CoarseTracker.cpp:51-208) for a batch of independent
 * (ref, cur) pairs: the whole level loop and Levenberg-Marquardt loop run on
 * the device, one workgroup per pair, no host round trips.  The caller applies
 * the write-back of CoarseTracker.cpp:198-202 (cur.T_f_w_, m_exposure_time).
 * Both frames of every job must be resident (hso_gpu_frame_upload). */
int hso_gpu_coarse_track_batch(hso_gpu_ctx* ctx, const hso_camera* cam,
                               const hso_track_params* params,
                               const hso_track_job* jobs, int n_jobs,
                               hso_track_result* results);

/* Same, split for callers that keep jobs resident and time only the device
 * work: `prepare` uploads the jobs' feature tables, `launch` enqueues the
 * kernel on the context stream (asynchronous), `collect` synchronises and
 * copies the results back.  Every successful `launch` must be followed by its
 * `collect` (or the context's destruction): a launch of the cooperative shape
 * (a batch smaller than the chip, split over workgroups that wait for each other)
 * holds its device's cooperative-launch turn until then, and another context's
 * such launch on the same device waits for the turn. */
int hso_gpu_coarse_track_prepare(hso_gpu_ctx* ctx, const hso_camera* cam,
                                 const hso_track_params* params,
                                 const hso_track_job* jobs, int n_jobs);
int hso_gpu_coarse_track_launch(hso_gpu_ctx* ctx);
int hso_gpu_coarse_track_collect(hso_gpu_ctx* ctx, hso_track_result* results);
/* The read-back in two halves for a caller that steps a resident batch repeatedly: _begin queues the copy of the launch's result
 * records behind it and returns at once, _end waits for the OLDEST read-back in flight (at most two) and delivers its records — the
 * caller enqueues the next step (frame construction, launch, _begin) before it calls _end for this one, so the stream never runs
 * dry between steps.  Batch shapes only (HSO_E_UNSUPPORTED for a cooperative launch: use hso_gpu_coarse_track_collect). */
int hso_gpu_coarse_track_collect_begin(hso_gpu_ctx* ctx);
int hso_gpu_coarse_track_collect_end(hso_gpu_ctx* ctx, hso_track_result* results);

/* One residual + Jacobian + normal-equation evaluation
 * (precomputeReferencePatches :416-497, optional selectRobustFunctionLevel
 * :530-644, computeResiduals :242-414, computeGS :499-525) at one level for one
 * pair — the unit of work of SURVEY.md §8(d) and the per-call parity hook. */
typedef struct hso_eval_out {
  double H[49];        /* row-major 7x7, symmetric */
  double b[7];
  double energy;       /* E / m_total_terms */
  double energy_sum;   /* E */
  int32_t n_terms;     /* m_total_terms */
  int32_t n_saturated; /* m_saturated_terms */
  int32_t n_select;    /* errors.size() seen by selectRobustFunctionLevel (0 if not run) */
  int32_t n_visible;   /* set bits of m_visible_fts */
  float huber, outlier;/* thresholds used */
} hso_eval_out;

/* huber_thresh <= 0 => run selectRobustFunctionLevel(T, exposure_rat) first and
 * use (and report) its thresholds.  ref_patch_out (n*PATCH_AREA floats),
 * visible_out (n bytes) and abs_err_out (n*PATCH_AREA floats, first n_select
 * valid, order unspecified) may be NULL. */
int hso_gpu_tracker_eval(hso_gpu_ctx* ctx, const hso_camera* cam,
                         const hso_track_params* params, const hso_track_job* job,
                         int level, const hso_se3* T_cur_ref, float exposure_rat,
                         float huber_thresh, float outlier_thresh,
                         hso_eval_out* out, float* ref_patch_out,
                         uint8_t* visible_out, float* abs_err_out);

/* ---- Matcher::findMatchDirect + feature_alignment::align1D/align2D,
 *      src/matcher.cpp:46-155,226-238,270-440, src/feature_alignment.cpp:164-308,464-605 ---- */

enum { HSO_FTR_CORNER = 0, HSO_FTR_EDGELET = 1, HSO_FTR_GRADIENT = 2 }; /* Feature::FeatureType, feature.h:37 */

/* One reprojection candidate, flattened by the caller from (Point, ref_ftr_ chosen by
 * Point::getCloseViewObs, src/point.cpp:116-136) exactly as findMatchDirect reads them. */
typedef struct hso_align_job {
  int64_t ref_frame_id;   /* ref_ftr_->frame (resident) */
  int32_t ref_level;      /* ref_ftr_->level */
  int32_t type;           /* ref_ftr_->type: EDGELET -> align1D + checkNormal, else align2D */
  double px_ref[2];       /* ref_ftr_->px */
  double f_ref[3];        /* ref_ftr_->f */
  double depth;           /* 1/pt.idist_ if the reference is the host frame, else |ref.pos - pt.pos| (matcher.cpp:295-306) */
  double grad[2];         /* ref_ftr_->grad */
  hso_se3 T_cur_ref;      /* cur.T_f_w_ * ref.T_f_w_^-1 (matcher.cpp:293) */
  double px_cur[2];       /* in: projected estimate, level-0 pixels */
  float exposure_rat;     /* float(cur.m_exposure_time / ref.m_exposure_time) (matcher.cpp:319) */
  int32_t kf_gap_lt4;     /* cur.keyFrameId_ - ref.keyFrameId_ < 4 (matcher.cpp:317) */
} hso_align_job;

enum {                    /* hso_align_out.stage: the first check that failed, 0 = success */
  HSO_ALIGN_OK = 0, HSO_ALIGN_REF_BORDER = 1, HSO_ALIGN_NOT_CONVERGED = 2, HSO_ALIGN_NORMAL = 3,
  HSO_ALIGN_NCC = 4, HSO_ALIGN_JUMP = 5
};

typedef struct hso_align_out {
  int32_t success;        /* findMatchDirect's return value */
  int32_t stage;
  int32_t search_level;   /* Matcher::search_level_ */
  int32_t iters;          /* LK iterations executed */
  double px_cur[2];       /* refined position, level-0 pixels (px_scaled * 2^search_level) */
  double A_cur_ref[4];    /* Matcher::A_cur_ref_, row-major (needed to rotate edgelet grads, reprojector.cpp:400-406) */
  double h_inv;           /* Matcher::h_inv_ (align1D only) */
  float ncc;              /* the NCC value checkNCC compares with 0.7 */
  float chi2;             /* chi2 of the last LK iteration */
} hso_align_out;

/* All candidates of one current frame in one launch, one wavefront per candidate (64 lanes =
 * the 8x8 patch).  The caller applies Reprojector::reprojectCell's first-success-per-cell
 * rule (src/reprojector.cpp:352-429) to the result array in its own visiting order. */
int hso_gpu_align_batch(hso_gpu_ctx* ctx, const hso_camera* cam, int64_t cur_frame_id,
                        const hso_align_job* jobs, int n_jobs, hso_align_out* out);
/* The candidates of many current frames (independent sequences, one frame size) in one launch:
 * cur_frame_ids[i] is the frame job i is searched in. */
int hso_gpu_align_multi(hso_gpu_ctx* ctx, const hso_camera* cam, const int64_t* cur_frame_ids,
                        const hso_align_job* jobs, int n_jobs, hso_align_out* out);

/* ---- Reprojector::reprojectPoint + the matching of its candidates, src/reprojector.cpp:504-529 and
 *      :352-429: project every map point of the overlapping keyframes
 *      into the current frame, bin it into the reprojection grid, choose its reference observation
 *      (Point::getCloseViewObs, src/point.cpp:116-136) and run findMatchDirect on it — one call,
 *      nothing returns to the host between projection and matching.  The caller keeps what walks
 *      the pointer graph and mutates it: which keyframes overlap (:108-199), the per-cell sort and
 *      visiting order, the first-success-per-cell / maxFts budget and the n_failed_reproj_
 *      bookkeeping (:352-429), all of which only read this call's result arrays. ---- */
typedef struct hso_kf {        /* a keyframe whose points are projected (it must be resident) */
  int64_t frame_id;
  hso_se3 T_f_w;
  double exposure_time;        /* Frame::m_exposure_time */
  int32_t keyframe_id;         /* Frame::keyFrameId_ */
  int32_t pad_;
} hso_kf;

typedef struct hso_obs {       /* one entry of Point::obs_: a keyframe feature observing the point */
  int32_t kf;                  /* index into the hso_kf table */
  int32_t level;               /* Feature::level */
  int32_t type;                /* Feature::type */
  int32_t pad_;
  double px[2], f[3], grad[2];
} hso_obs;

typedef struct hso_map_point {
  double pos[3];               /* Point::pos_ (world) */
  double idist;                /* Point::idist_ in the host frame */
  double host_f[3];            /* hostFeature_->f */
  int32_t host_kf;             /* index of hostFeature_->frame in the hso_kf table */
  int32_t obs_begin, obs_count;/* obs_[0..count) = obs[obs_begin ...], in list order */
  int32_t pad_;                /* sequence maps: the point's state word (HSO_PT_*: quality key (Point::type_ << 4) | Point::ftr_type_,
                                  n_failed_reproj_, n_succeeded_reproj_, isBad_); unused by the value-passing calls */
} hso_map_point;

typedef struct hso_reproj_point {
  int32_t projected;           /* reprojectPoint's return value (:508, :512) */
  int32_t cell;                /* grid cell index k (:517-518), valid when projected */
  double px[2];                /* Candidate::px before refinement */
  int32_t ref_obs;             /* index into obs of the observation getCloseViewObs chose; -1: none within 60 degrees */
  int32_t pad_;
} hso_reproj_point;

/* cell_size, grid_n_cols: Reprojector::initializeGrid (:58-68).  proj_out and match_out hold
 * n_points entries in point order; match_out[i] is findMatchDirect's result for point i (all zero
 * when the point is not projected or has no reference observation: the reference never calls the
 * matcher for the former and returns false at once for the latter, src/matcher.cpp:276-286). */
int hso_gpu_reproject_match(hso_gpu_ctx* ctx, const hso_camera* cam, int64_t cur_frame_id, const hso_se3* T_cur_w,
                            double cur_exposure_time, int cur_keyframe_id, const hso_kf* kfs, int n_kfs,
                            const hso_map_point* points, int n_points, const hso_obs* obs, int n_obs,
                            int cell_size, int grid_n_cols, hso_reproj_point* proj_out, hso_align_out* match_out);

/* The same for the current frames of many sequences in one launch.  Frame f owns the keyframes
 * kfs[kf_begin .. kf_begin + kf_count) and the points points[point_begin .. + point_count); inside
 * those points host_kf and obs[].kf count from the frame's kf_begin (obs_begin stays absolute).
 * Every point must belong to exactly one frame. */
typedef struct hso_reproj_frame {
  int64_t cur_frame_id;
  hso_se3 T_cur_w;
  double cur_exposure_time;
  int32_t cur_keyframe_id;
  int32_t kf_begin, kf_count;
  int32_t point_begin, point_count;
  int32_t pad_;
} hso_reproj_frame;
int hso_gpu_reproject_match_multi(hso_gpu_ctx* ctx, const hso_camera* cam, const hso_reproj_frame* frames, int n_frames,
                                  const hso_kf* kfs, int n_kfs, const hso_map_point* points, int n_points,
                                  const hso_obs* obs, int n_obs, int cell_size, int grid_n_cols,
                                  hso_reproj_point* proj_out, hso_align_out* match_out);

/* one listed point after projection and matching (the record the grid selection reads; recorded runs fetch the examined ones) */
typedef struct hso_match_brief {
  double px[2];                /* Candidate::px, the projected position */
  double px_cur[2];            /* the refined position (valid when success) */
  float grad[2];               /* the new feature's gradient direction: normalised A_cur_ref * ref grad (reprojector.cpp:400-406) */
  int32_t cell;                /* grid cell, -1: reprojectPoint returned false */
  int32_t ref_obs;             /* chosen observation, index into the map's obs table; -1: none within 60 degrees */
  int8_t success, stage, search_level, ref_type;   /* findMatchDirect's result, HSO_ALIGN_* stage, Matcher::search_level_, ref_ftr_->type */
  int32_t pad_;                /* hso_gpu_seq_chain: the point's position in its job's list */
} hso_match_brief;

/* ---- pose_optimizer::optimizeLevenbergMarquardt3rd, src/pose_optimizer.cpp:399-771 ---- */

/* One feature of the frame being optimised, in Frame::fts_ order.  has_point = 0 keeps the
 * slot (point == NULL features are skipped but still count in fts_.size(), :696). */
typedef struct hso_pose_feat {
  int32_t has_point;    /* (*it)->point != NULL */
  int32_t type;         /* Feature::type (HSO_FTR_*): EDGELET uses the 1-D residual grad^T e */
  int32_t level;        /* Feature::level: residual scaled by 1/2^level */
  int32_t temporary;    /* point->type_ == Point::TYPE_TEMPORARY: weight * 0.5 */
  int32_t host_pose;    /* index into poses_f_w: point->hostFeature_->frame->T_f_w_ */
  int32_t _pad;
  double f[3];          /* Feature::f (observation bearing in this frame) */
  double grad[2];       /* Feature::grad */
  double host_f[3];     /* point->hostFeature_->f */
  double idist;         /* point->idist_ */
} hso_pose_feat;

typedef struct hso_pose_job {
  const hso_pose_feat* feats;  /* host pointer */
  int32_t n_feats;             /* frame->fts_.size() */
  int32_t n_poses;
  const hso_se3* poses_f_w;    /* host pointer: T_f_w_ of the host keyframes */
  hso_se3 T_f_w;               /* frame->T_f_w_ (initial) */
  double reproj_thresh;        /* Config::poseOptimThresh() = 2.0 */
  int32_t n_iter;              /* 12, frame_handler_mono.cpp:242 */
  int32_t _pad;
} hso_pose_job;

typedef struct hso_pose_result {
  hso_se3 T_f_w;           /* frame->T_f_w_ */
  double cov[36];          /* frame->Cov_, row-major */
  double estimated_scale;  /* in pixels (x errorMultiplier2) */
  double error_init, error_final;
  float error_in_px;       /* frame->m_error_in_px */
  int32_t num_obs;         /* after outlier removal */
  int32_t n_deleted;       /* features whose point was set to NULL */
  int32_t iters;           /* outer LM iterations entered */
  int32_t n_trials_total;  /* linear solves performed */
  int32_t status;          /* 0 ok, 1 = no residuals (early return, :456) */
} hso_pose_result;

/* Batched over independent frames: one workgroup per job, the whole LM loop on the device.
 * outlier_mask[j] (n_feats bytes each, may be NULL) = 1 where the reference sets
 * feature->point = NULL (:722-748). */
int hso_gpu_pose_optimize_batch(hso_gpu_ctx* ctx, const hso_camera* cam, const hso_pose_job* jobs,
                                int n_jobs, hso_pose_result* results, uint8_t* const* outlier_mask);

/* ---- local bundle adjustment: the Jacobian / Hessian build the reference delegates to g2o.
 *      Edges: EdgeProjectID2UV / EdgeProjectID2UVEdgeLet (include/hso/bundle_adjustment.h:204-404),
 *      created at src/bundle_adjustment.cpp:690-812; accumulation = g2o BlockSolver::buildSystem
 *      -> BaseMultiEdge::constructQuadraticForm (thirdparty/g2o/g2o/core/base_multi_edge.hpp:36-48,
 *      171-222) with RobustKernelHuber (robust_kernel_impl.cpp:78-91) and
 *      robustInformation = rho'(chi2) * Omega (base_edge.h:96-102). ---- */
typedef struct hso_ba_edge {
  int32_t point;      /* vertex 0: inverse-depth point index */
  int32_t host;       /* vertex 1: host keyframe pose index */
  int32_t target;     /* vertex 2: observing keyframe pose index */
  int32_t type;       /* HSO_FTR_EDGELET -> 1-D edge along `normal`, else 2-D edge */
  int32_t level;      /* information = 1 / 4^level (bundle_adjustment.cpp:758,788) */
  int32_t _pad;
  double fH[3];       /* setHostBearing(point->hostFeature_->f) */
  double meas[2];     /* project2d(obs->f); edgelets: meas[0] = grad^T project2d(obs->f) */
  double normal[2];   /* setTargetNormal(obs->grad) (edgelets) */
} hso_ba_edge;

/* Dense outputs, sized by the caller.  Unknown order: points first (1 each), then poses (6 each,
 * g2o SE3Quat tangent order [omega, upsilon]); entries of fixed poses are left zero.
 *   Hpp[n_points], bp[n_points]                      point diagonal and gradient
 *   Hpc[n_points * n_poses * 6]                      point-pose blocks (1x6), row-major
 *   Hcc[n_poses * n_poses * 36]                      pose-pose blocks (6x6, row-major), block (i,j)
 *                                                    filled for i <= j (the upper half, like g2o)
 *   bc[n_poses * 6]
 *   edge_err[n_edges * 2], edge_chi2[n_edges]        _error and chi2() = e^T Omega e per edge
 *   chi2_sum[2]                                      sum of chi2, sum of robustified rho(chi2) */
int hso_gpu_ba_linearize(hso_gpu_ctx* ctx, const hso_se3* poses_f_w, const uint8_t* pose_fixed, int n_poses,
                         const double* idist, int n_points, const hso_ba_edge* edges, int n_edges,
                         double huber_corner, double huber_edge,
                         double* Hpp, double* bp, double* Hpc, double* Hcc, double* bc,
                         double* edge_err, double* edge_chi2, double* chi2_sum);

/* Huber deltas of ba::LocalBundleAdjustment, src/bundle_adjustment.cpp:618-680: per non-host observation
 * e = (project2d(obs->f) - project2d(Tth * fH / idist)) / 2^level; corners collect |e| and edgelets
 * |grad^T e| as floats; delta = 1.4826 * getMedian (upper median, include/hso/vikit/math_utils.h:119-126),
 * with the fallbacks 1 / errorMultiplier2 (corner) and 0.5 / errorMultiplier2 (edgelet) when one class is
 * empty (:664-680; both empty: the reference leaves the deltas uninitialised, here they are returned as 0).
 * obs_uv[2k..2k+1] = project2d(obs->f) of edge k (for a corner edge this equals edges[k].meas; an edgelet's
 * meas only keeps grad^T of it).  error_multiplier2 = center_kf->cam_->errorMultiplier2(). */
int hso_gpu_ba_huber_deltas(hso_gpu_ctx* ctx, const hso_se3* poses_f_w, int n_poses, const double* idist, int n_points,
                            const hso_ba_edge* edges, const double* obs_uv, int n_edges, double error_multiplier2,
                            float* huber_corner, float* huber_edge);
/* The same for the windows of many sequences in one call (one upload, one launch, one read-back): huber_corner / huber_edge of
 * every job are written; a job with n_edges == 0 gets 0 / 0. */
typedef struct hso_ba_deltas_job {
  const hso_se3* poses_f_w; const double* idist; const hso_ba_edge* edges; const double* obs_uv;
  int32_t n_poses, n_points, n_edges;
  float huber_corner, huber_edge;   /* out */
} hso_ba_deltas_job;
int hso_gpu_ba_huber_deltas_multi(hso_gpu_ctx* ctx, hso_ba_deltas_job* jobs, int n_jobs, double error_multiplier2);

/* What runSparseBAOptimizer (src/bundle_adjustment.cpp:351-361) leaves behind. */
typedef struct hso_ba_result {
  double init_chi2;     /* optimizer.activeChi2() before optimize(): sum of e^T Omega e */
  double final_chi2;    /* activeChi2() after optimize(): of the LAST evaluation — a rejected trial's errors when the
                           last LM step was rejected (g2o pops the vertices, not the edge errors) */
  double robust_chi2;   /* currentChi (activeRobustChi2 of the accepted state) at exit */
  double lambda;        /* _currentLambda at exit */
  int32_t iterations;   /* outer iterations executed (SparseOptimizer::optimize's cjIterations) */
  int32_t n_solves;     /* LM trials = linear solves in total */
  int32_t n_accepted;   /* trials with rho > 0 */
  int32_t stop;         /* 0: iteration budget used; 1: Terminate (trials exhausted or rho == 0); 2: nBad >= 3 */
} hso_ba_result;

/* The numeric core of ba::LocalBundleAdjustment (src/bundle_adjustment.cpp:815-823 -> runSparseBAOptimizer :351-361 ->
 * SparseOptimizer::optimize, thirdparty/g2o/g2o/core/sparse_optimizer.cpp:354-420): Levenberg-Marquardt exactly as
 * OptimizationAlgorithmLevenberg::solve drives it (optimization_algorithm_levenberg.cpp:61-164): lambda0 = 1e-5 * max
 * diagonal entry, per trial (H + lambda I) x = b, SE3Quat::exp(dx) * pose / idist += dx (se3quat.h:223-257,
 * bundle_adjustment.h:198-200), rho = (chi - chi_new) / (x^T (lambda x + b) + 1e-3), a good step scales lambda by
 * clamp(1 - (2 rho - 1)^3, 1/3, 2/3), a bad one by ni (ni *= 2), at most 5 trials (setMaxTrialsAfterFailure(5),
 * bundle_adjustment.cpp:571), ORB-SLAM's stop: three iterations in a row that gain < 0.1 % (:154-161).
 * Errors, Jacobians and the robustified blocks come from the device kernels of hso_gpu_ba_linearize; the linear
 * solve eliminates the 1-D inverse-depth unknowns on the device (scalar Schur complement, then back-substitution of the
 * points) and factors the remaining <= 6 * n_free_poses (<= 96) system densely, also on the device — the same solution as
 * the reference's sparse LDL^T of the full system up to rounding.  poses_f_w / idist are updated in place (fixed poses unchanged); edge_chi2_out[n_edges]
 * (may be NULL) = what edge->chi2() returns after optimize(), the input of the culling at :855-892. */
int hso_gpu_ba_optimize(hso_gpu_ctx* ctx, hso_se3* poses_f_w, const uint8_t* pose_fixed, int n_poses, double* idist,
                        int n_points, const hso_ba_edge* edges, int n_edges, double huber_corner, double huber_edge,
                        int n_iter, double* edge_chi2_out, hso_ba_result* result);

/* The local-BA windows of many sequences (one per keyframe event) in one call: the problems advance through the
 * Levenberg loop in lockstep rounds — per round each kind of device work is launched once for all problems that want it, and the
 * accept / reject decision of the round is taken on the device too, so the rounds of a call are enqueued back to back and the
 * caller's thread waits once (it was one round trip per LM step, and before that one per problem and step).  Each problem's
 * arithmetic is exactly that of hso_gpu_ba_optimize. */
typedef struct hso_ba_problem {
  hso_se3* poses_f_w;          /* in / out */
  const uint8_t* pose_fixed;
  double* idist;               /* in / out */
  const hso_ba_edge* edges;
  double* edge_chi2_out;       /* [n_edges] or NULL */
  hso_ba_result* result;
  int32_t n_poses, n_points, n_edges, n_iter;
  double huber_corner, huber_edge;
} hso_ba_problem;
int hso_gpu_ba_optimize_multi(hso_gpu_ctx* ctx, const hso_ba_problem* problems, int n_problems);

/* ba::LocalBundleAdjustment's two device stages in ONE call (src/bundle_adjustment.cpp:618-680 then :815-830): the Huber deltas
 * of every window from its initial state — hso_gpu_ba_huber_deltas' arithmetic, the two medians by an exact radix select on the
 * device — and then hso_gpu_ba_optimize_multi with them.  The windows go up once (the two-call form staged and sent the edge
 * tables twice and read every window's error magnitudes back for the medians).  obs_uv[q] = project2d(obs->f) of window q's
 * edges [2 * n_edges]; problems[q].huber_corner / huber_edge are ignored; huber_out[2 * q], [2 * q + 1] receive the deltas used
 * (corner, edge).  Results are those of the two calls in sequence, bit for bit. */
int hso_gpu_ba_local_multi(hso_gpu_ctx* ctx, const hso_ba_problem* problems, const double* const* obs_uv, int n_problems,
                           double error_multiplier2, float* huber_out);

/* ---- DepthFilter seed observation: DepthFilter::observeDepthRow (src/depth_filter.cpp:580-675),
 *      updateSeed :528-537, computeTau :539-555; Matcher::doLineStereo (src/matcher.cpp:802-1049),
 *      KLTLimited2D/1D :1296-1606, warp::createPatch :159-196, ZMNCC_F
 *      (include/hso/vikit/patch_score.h:268-305), depthFromTriangulation :242-255 ---- */
typedef struct hso_seed {
  int64_t ref_frame_id;   /* seed.ftr->frame (resident) */
  int32_t level;          /* seed.ftr->level */
  int32_t type;           /* seed.ftr->type (HSO_FTR_*) */
  double px[2];           /* seed.ftr->px */
  double f[3];            /* seed.ftr->f */
  double grad[2];         /* seed.ftr->grad */
  hso_se3 T_ref_w;        /* seed.ftr->frame->T_f_w_ */
  double ref_exposure;    /* seed.ftr->frame->m_exposure_time */
  float mu, sigma2;       /* Seed::mu, sigma2 (include/hso/depth_filter.h:54-58) */
  float b;                /* Seed::b (outlier counter, ++ on a failed match) */
  float _pad;
} hso_seed;

typedef struct hso_seed_out {
  float mu, sigma2, b;    /* updated state */
  int32_t result;         /* doLineStereo's code 1 / -1..-4; 0 = not visible in the active frame (:593-606) */
  int32_t is_update;      /* Seed::is_update */
  int32_t is_valid;       /* 0 if z_inv_min was NaN (:618) */
  int32_t search_level;   /* Matcher::search_level_ */
  int32_t epl_start[2], epl_end[2];
  int32_t n_steps;        /* epipolar samples visited */
  double px_cur[2];       /* Matcher::px_cur_ (last_matched_px) */
  double z;               /* triangulated depth (result == 1) */
  float zmncc_best, zmncc_second;
} hso_seed_out;

/* One wavefront per seed (lane = patch pixel).  cur_T_f_w / cur_exposure: the active frame's pose
 * and m_exposure_time; px_error_angle: DepthFilter::px_error_angle_ (:360-366). */
int hso_gpu_seed_observe(hso_gpu_ctx* ctx, const hso_camera* cam, int64_t cur_frame_id, const hso_se3* cur_T_f_w,
                         double cur_exposure, double px_error_angle, const hso_seed* seeds, int n_seeds,
                         hso_seed_out* out);

/* The same for the active frames of many sequences (or the frame queue of one, :208-246) in one
 * launch: seed i is observed in frames[seed_frame[i]]. */
typedef struct hso_seed_frame {
  int64_t frame_id;
  hso_se3 T_f_w;
  double exposure_time;
} hso_seed_frame;
int hso_gpu_seed_observe_multi(hso_gpu_ctx* ctx, const hso_camera* cam, const hso_seed_frame* frames, int n_frames,
                               const int32_t* seed_frame, double px_error_angle, const hso_seed* seeds, int n_seeds,
                               hso_seed_out* out);

/* ---- resident seed tables: the std::list<Seed> of a DepthFilter (include/hso/depth_filter.h:211) kept in HBM between calls.
 *      initializeSeeds (src/depth_filter.cpp:164-205) appends, the erase sites of updateSeeds (:368-401, :423-497) erase, and one
 *      observation of every live seed (observeDepth, :557-675) updates mu / sigma2 / b in place.  Per frame only one hso_seed_frame
 *      per group goes in and one 16-byte hso_seed_brief per slot comes out (the value-passing hso_gpu_seed_observe moves a 216-byte
 *      record in and an 88-byte one out per seed).  A seed's `group` names which entry of the observe call's frames[] it is
 *      observed in: 0 for the table of one sequence; the sequence index when one table serves many.  Slot indices are stable:
 *      an erased slot is skipped, never reused.  Host frames of the seeds must stay resident while their seeds are alive. ---- */
typedef struct hso_seed_brief {
  float mu, sigma2, b;       /* the state after the observation (what the table now holds) */
  int8_t result;             /* doLineStereo's code 1 / -1..-4; 0 = not visible in the active frame, or an erased slot */
  int8_t is_update, is_valid, search_level;
} hso_seed_brief;
int hso_gpu_seed_table_create(hso_gpu_ctx* ctx, int* table_out);
int hso_gpu_seed_table_destroy(hso_gpu_ctx* ctx, int table);
int hso_gpu_seed_table_append(hso_gpu_ctx* ctx, int table, const hso_seed* seeds, const int32_t* group /* n or NULL = all 0 */, int n,
                              int32_t* first_slot);
int hso_gpu_seed_table_erase(hso_gpu_ctx* ctx, int table, const int32_t* slots, int n);
int hso_gpu_seed_table_size(hso_gpu_ctx* ctx, int table, int* n_slots, int* n_live);
/* Erased slots keep their index (and their place in the launch grid and in brief_out) until the table is compacted: the live
 * records move to slots 0 .. n_live - 1 in their order; remap (n_slots entries of before the call, or NULL) receives each old
 * slot's new index, -1 for an erased one.  Returns the new number of slots.  A long sequence erases as many seeds as it starts
 * (convergence, divergence, host keyframe dropped): compact at keyframe rate to keep the table at the live size. */
int hso_gpu_seed_table_compact(hso_gpu_ctx* ctx, int table, int32_t* remap);
/* brief_out: n_slots entries or NULL; full_out: n_slots hso_seed_out records or NULL (epipolar end points, px_cur, z: what a
 * keyframe observation needs for FeatureExtractor::setGridOccpuancy, :669-673) */
int hso_gpu_seed_table_observe(hso_gpu_ctx* ctx, const hso_camera* cam, int table, const hso_seed_frame* frames, int n_frames,
                               double px_error_angle, hso_seed_brief* brief_out, hso_seed_out* full_out);
/* The same with (a) groups that sit a step out: frames[g].frame_id < 0 skips every seed of group g (state untouched, brief all
 * zero); (b) px_out (n_slots x 2 floats, may be NULL): Matcher::px_cur_ of the seeds whose result is 1 — what a keyframe
 * observation feeds to FeatureExtractor::setGridOccpuancy (src/depth_filter.cpp:669-673). */
int hso_gpu_seed_table_observe_groups(hso_gpu_ctx* ctx, const hso_camera* cam, int table, const hso_seed_frame* frames, int n_frames,
                                      double px_error_angle, hso_seed_brief* brief_out, float* px_out, hso_seed_out* full_out);
/* DepthFilter::observeDepthWithPreviousFrameOnce (reference src/depth_filter.cpp:677-726, called from the depth thread's idle loop,
 * :254-263) with Matcher::findEpipolarMatchPrevious (src/matcher.cpp:1051-1293) over a resident table: every live seed hosted in
 * keyframe host_frame_ids[k] observes the EARLIER frame pre_frames[k] (the first of Seed::pre_frames; the caller owns those lists,
 * one per keyframe, and drops the entry afterwards whatever the outcome, :693-724); seeds of keyframes not named sit the call
 * out.  mu / sigma2 are updated in place where the match succeeds; b is never touched.  Per slot (brief_out / full_out sized to
 * the table, may be null): is_update = 1 when the earlier frame saw the point (it then joins Seed::optFrames_P, :702-703), result =
 * 1 matched and updated, -1 epipolar-angle filter or > 100 steps, -4 march rejected, -3 refinement rejected, -2 triangulation; a
 * slot not addressed reports zeros.  The frames must be resident. */
int hso_gpu_seed_table_observe_previous(hso_gpu_ctx* ctx, const hso_camera* cam, int table, const int64_t* host_frame_ids,
                                        const hso_seed_frame* pre_frames, int n, double px_error_angle, hso_seed_brief* brief_out,
                                        hso_seed_out* full_out);
/* The same pass on the depth filter's own stream, overlapped with whatever the caller issues next on the context's stream (the
 * reference runs the depth filter on its own thread beside the tracker, src/depth_filter.cpp:130-162): _begin queues the pass behind
 * everything already issued and returns at once; _end waits for it and delivers the briefs (brief_out: n_brief >= the table's size at
 * _begin, or NULL).  One pass in flight per context.  Until _end the caller must not rely on the table's contents; the library itself
 * waits for the pass before any other seed-table entry point, hso_gpu_frame_release and hso_gpu_synchronize proceed, so nothing it
 * reads can be changed or freed under it.  Results are those of hso_gpu_seed_table_observe_previous. */
int hso_gpu_seed_table_observe_previous_begin(hso_gpu_ctx* ctx, const hso_camera* cam, int table, const int64_t* host_frame_ids,
                                              const hso_seed_frame* pre_frames, int n, double px_error_angle);
int hso_gpu_seed_table_observe_previous_end(hso_gpu_ctx* ctx, int table, hso_seed_brief* brief_out, int n_brief);
/* Local BA moved keyframes (src/bundle_adjustment.cpp:826-834): refresh T_ref_w of every live seed hosted in one of these frames
 * (the reference reads seed.ftr->frame->T_f_w_ at every observation, src/depth_filter.cpp:588) */
int hso_gpu_seed_table_set_host_pose(hso_gpu_ctx* ctx, int table, const int64_t* frame_ids, const hso_se3* T_f_w, int n);
/* the records of slots [first, first + n) as the table holds them now (convergence / activation read them at keyframe rate) */
int hso_gpu_seed_table_read(hso_gpu_ctx* ctx, int table, int first, int n, hso_seed* seeds_out);

/* ---- DepthFilter::activatePoint + seedOptimizer, src/depth_filter.cpp:729-1073 ---- */

#define HSO_ACTIVATE_MAX_TARGETS 64

/* One frame of seed.optFrames_P followed by seed.optFrames_A, in the reference's visiting order. */
typedef struct hso_activate_target {
  int64_t frame_id;       /* resident */
  hso_se3 T_f_w;
  double exposure;        /* m_exposure_time */
} hso_activate_target;

typedef struct hso_activate_out {
  int32_t activated;      /* activatePoint's return value */
  int32_t is_valid;       /* the isValid out-parameter: 1 / 0, or -1 when the reference leaves it untouched */
  int32_t n_targets;      /* targets.size() after the projection test (:741-769) */
  int32_t n_matched;      /* targetResult.size() (:785-822) */
  double dist_mean;       /* distMean (:827) */
  double huber;           /* MAD scale used by seedOptimizer (:884) */
  double opt_id;          /* seed.opt_id */
  double energy;          /* robust energy at opt_id */
  int32_t n_iter;         /* outer LM iterations executed */
  int32_t _pad;
} hso_activate_out;

/* Seeds [i] use targets[target_begin[i] .. target_begin[i+1]) (<= HSO_ACTIVATE_MAX_TARGETS each).
 * n_mean_converge_frame: DepthFilter::nMeanConvergeFrame_.  match_out (optional, one per target):
 * the findMatchSeed result of every visited target (zero where no match was attempted).
 * Kernel 1: one wavefront per (seed, target) runs the projection test, the parallax test and
 * findMatchSeed; kernel 2: one wavefront per seed (lane = target) applies the gates and runs the
 * 1-D LM in fp64, summing over the targets in the reference's order. */
int hso_gpu_seed_activate(hso_gpu_ctx* ctx, const hso_camera* cam, const hso_seed* seeds, int n_seeds,
                          const int32_t* target_begin, const hso_activate_target* targets, int n_mean_converge_frame,
                          hso_activate_out* out, hso_align_out* match_out);
/* The same for the converged seeds of many sequences in one call (seeds and targets name their frames by id, so the tables
 * simply concatenate): n_mean_converge_frame[i] = nMeanConvergeFrame_ of the DepthFilter seed i belongs to. */
int hso_gpu_seed_activate_multi(hso_gpu_ctx* ctx, const hso_camera* cam, const hso_seed* seeds, int n_seeds,
                                const int32_t* target_begin, const hso_activate_target* targets,
                                const int32_t* n_mean_converge_frame, hso_activate_out* out, hso_align_out* match_out);

/* The same with the target frames named once: the seeds of a sequence share their few dozen target frames (Seed::optFrames_P / _A
 * hold the frames since their keyframes), so the call takes the UNIQUE frames (`frames`, n_frames of them) and, per (seed, target)
 * pair, an index into that table (target_frame[target_begin[i] .. target_begin[i+1])) — 4 bytes per pair instead of a 72-byte
 * record, one frame lookup per frame instead of one per pair (77 000 pairs per step of 128 sequences). */
int hso_gpu_seed_activate_frames(hso_gpu_ctx* ctx, const hso_camera* cam, const hso_seed* seeds, int n_seeds, const int32_t* target_begin,
                                 const int32_t* target_frame, const hso_activate_target* frames, int n_frames,
                                 const int32_t* n_mean_converge_frame, hso_activate_out* out);

/* The same for seeds that live in a resident seed table (hso_gpu_seed_table_*): seed i is the record in `slots[i]` as the table holds
 * it — mu / sigma2 / b as its last observation left them, the host keyframe's pose as hso_gpu_seed_table_set_host_pose left it — so
 * 4 bytes per seed cross the bus instead of a 152-byte record (0.8 MB per step of 128 sequences), and the caller does not assemble
 * records it already mirrored.  A dead slot is an error.  Results are those of hso_gpu_seed_activate_frames on
 * hso_gpu_seed_table_read's records of the same slots. */
int hso_gpu_seed_table_activate(hso_gpu_ctx* ctx, const hso_camera* cam, int table, const int32_t* slots, int n_seeds, const int32_t* target_begin,
                                const int32_t* target_frame, const hso_activate_target* frames, int n_frames,
                                const int32_t* n_mean_converge_frame, hso_activate_out* out);

/* The seed branch of Reprojector::reprojectMap (src/reprojector.cpp:309-329): reprojectorSeed (:531-554) — pTarget =
 * (T_cur_w * T_ref_w^-1) * (f / mu), rejected when its z < 0.001 or its truncated pixel lies within 8 px of the border —
 * and Matcher::findMatchSeed (src/matcher.cpp:442-518: parallax test, warp, exposure compensation, align1D / align2D,
 * NCC 0.8) of every seed in one launch.  proj_out[i].projected / px / cell (ref_obs unused), match_out[i] = the
 * findMatchSeed result (zero when not projected or the parallax test failed).  The caller keeps which seeds take part
 * (sigma test, haveReprojected), the per-cell sigma2 order and the first-success-per-cell rule. */
int hso_gpu_seed_reproject_match(hso_gpu_ctx* ctx, const hso_camera* cam, int64_t cur_frame_id, const hso_se3* T_cur_w,
                                 double cur_exposure, const hso_seed* seeds, int n_seeds, int cell_size, int grid_n_cols,
                                 hso_reproj_point* proj_out, hso_align_out* match_out);

/* ---- the grid selection of Reprojector::reprojectMap (remainder), src/reprojector.cpp:253-306
 *      (branch on the number of projected candidates, three passes over the cells in cell_order), reprojectCell :352-429
 *      (pointQualityComparator :333-345 on first visit, every examined candidate erased, first success per cell in passes 1
 *      and 2, up to three in pass 3), reprojectCellAll :556-612.  With every projected candidate already matched
 *      (hso_gpu_reproject_match), the policy is a pure function of the per-candidate (cell, quality, flags), the cell order
 *      and max_fts; this call evaluates it on the device for any number of frames (one workgroup each).
 *   frame_begin[n_frames + 1]  candidate ranges, candidates of a frame in projection order (the order reprojectPoint saw them)
 *   cell[i]                    grid cell (hso_reproj_point.cell)
 *   quality[i]                 (Point::type_ << 4) | Point::ftr_type_ — the comparator's key; higher sorts first, ties keep order
 *   flags[i]                   bit 0: findMatchDirect succeeded; bit 1: the point is TYPE_DELETED (costs a trial, nothing else)
 *   cell_order[n_cells]        the permutation Reprojector::initializeGrid shuffles once (:70-76)
 *   examined_out               per frame, at frame_begin[f]: the candidates examined, in examination order, as
 *                              (index within the frame) | (became a feature ? 0x80000000 : 0) — the caller walks the list and
 *                              applies the reference's bookkeeping (n_failed_reproj_ / n_succeeded_reproj_, new Feature)
 *   counts_out[4 * n_frames]   n_examined (= n_trials_), n_matches_, cell passes run (0 for reprojectCellAll), branch ---- */
int hso_gpu_reproject_select(hso_gpu_ctx* ctx, const int32_t* frame_begin, int n_frames, const int32_t* cell,
                             const uint8_t* quality, const uint8_t* flags, const int32_t* cell_order, int n_cells, int max_fts,
                             int32_t* examined_out, int32_t* counts_out);

/* An examined candidate of the chain in the form a driver applies it (src/reprojector.cpp:366-425): the
 * point (index in the frame's list), whether it became a feature, and the new feature's pixel, level, type and gradient. */
typedef struct hso_frame_match {
  double px_cur[2];
  float grad[2];
  int32_t point;               /* index in the frame's point list (hso_match_brief.pad_) */
  int8_t success, search_level, ref_type, pad_;
} hso_frame_match;

/* ---- sequence maps: the whole map of a sequence resident in HBM, mirrored row for row from the caller's tables and patched in
 *      place.  The reference decides per frame which keyframes'
 *      points it projects and in which order (src/reprojector.cpp:124-202: the covisible keyframes of the last frame, then the
 *      closest ones until ten), appends points between keyframes (converged seeds become candidates, :207-226) and threads every
 *      new keyframe's features into its points' observation lists (Point::addFrameRef, src/point.cpp:78-82: push_front).  Here
 *        - the point table is indexed by the caller's point id, the observation table by its feature id (a keyframe feature IS
 *          the observation row); a point's observations form a linked list through the observation rows
 *          (hso_map_point.obs_begin = first row, obs_count = length, hso_obs.pad_ = next row), so push_front / erase of one
 *          observation patches one or two rows;
 *        - hso_gpu_seqmap_patch scatters changed rows (asynchronous on the context stream; rows past the end grow the tables);
 *        - the keyframes' feature lists (Frame::fts_), the candidate list and the Feature::point / Frame::key_pts_ links are
 *          mirrored too (hso_gpu_seqmap_patch_lists / _patch_links / _set_key_points), so that the per-frame walk of
 *          Reprojector::reprojectMap runs on the device (hso_gpu_seq_chain below). ---- */
int hso_gpu_seqmap_create(hso_gpu_ctx* ctx, int* map_out);
int hso_gpu_seqmap_destroy(hso_gpu_ctx* ctx, int map);
/* the keyframe table (whole table every time: a few dozen rows kept on the host side of the library; poses change with every
 * local BA).  points' host_kf and observations' kf index it; every keyframe must be resident. */
int hso_gpu_seqmap_set_keyframes(hso_gpu_ctx* ctx, int map, const hso_kf* kfs, int n_kfs);
/* row point_ids[i] of the point table = points[i]; row obs_ids[i] of the observation table = obs[i].  Either part may be empty.
 * Asynchronous on the context stream: the rows are copied out of the caller's arrays before the call returns. */
int hso_gpu_seqmap_patch(hso_gpu_ctx* ctx, int map, const int32_t* point_ids, const hso_map_point* points, int n_points,
                         const int32_t* obs_ids, const hso_obs* obs, int n_obs);
/* the patches of many maps (one step of many sequences) in one call: one staging image, one scatter per row kind */
typedef struct hso_seqmap_rows {
  int32_t map, n_points, n_obs, pad_;
  const int32_t* point_ids; const hso_map_point* points;
  const int32_t* obs_ids; const hso_obs* obs;
  const int32_t* obs_point;    /* may be NULL: Feature::point of the patched observation rows (point row, -1: NULL) */
} hso_seqmap_rows;
/* a map appears at most once per call */
int hso_gpu_seqmap_patch_multi(hso_gpu_ctx* ctx, const hso_seqmap_rows* patches, int n_patches);
int hso_gpu_seqmap_size(hso_gpu_ctx* ctx, int map, int* n_kfs, int* n_points, int* n_obs);
/* parity / trace read-back of rows as the device holds them now */
int hso_gpu_seqmap_read(hso_gpu_ctx* ctx, int map, const int32_t* point_ids, int n_points, hso_map_point* points_out,
                        const int32_t* obs_ids, int n_obs, hso_obs* obs_out);
/* ---- the resident per-frame chain ("removes the last per-frame host loop over the
 *      pointer graph and the D2H / H2D of candidates between tracker and alignment").  FrameHandlerMono::processFrame from the motion
 *      prior to the inputs of its keyframe decision (src/frame_handler_mono.cpp:173-291) for the current frames of n sequences, every
 *      table it reads or writes resident in the sequence maps:
 *        1. CoarseTracker::makeDepthRef (src/CoarseTracker.cpp:210-240): the reference frame's feature table (px, f, distance of the
 *           point along the bearing) is built on the device from the features the previous call left for that frame — or, when the
 *           reference is a keyframe, from the keyframe's feature list — and the point rows; CoarseTracker::run; the write-back of
 *           :198-202 (cur.T_f_w_, m_exposure_time);
 *        2. Reprojector::reprojectMap's walk (src/reprojector.cpp:98-254): the covisible keyframes the job names, then the keyframes
 *           that see the frame (Map::getCloseKeyframes, src/map.cpp:193-213: a key point of the keyframe projects into the frame),
 *           nearest first, up to max_kfs; the points of their features once each (Point::last_projected_kf_id_), TYPE_TEMPORARY
 *           skipped; then the candidates; then the temporary points the job lists;
 *        3. projection, reference choice, findMatchDirect, the grid selection, the frame's features, optimizeLevenbergMarquardt3rd
 *           (one launch sequence over the device's own list);
 *        4. the bookkeeping of reprojectCell / reprojectCellAll on the examined candidates (:366-425, :214-222, :247-251):
 *           n_failed_reproj_ / n_succeeded_reproj_, UNKNOWN -> GOOD above 10 successes, deletion above 15 / 30 failures — applied to
 *           the point rows' state words, the changes of kind reported as events; the outlier mask of the pose optimiser applied to the
 *           frame's feature table (feature->point = NULL, src/pose_optimizer.cpp:722-748);
 *        5. the inputs of the decisions that follow: needNewKf's two optical-flow sums over the last keyframe's features
 *           (src/frame_handler_mono.cpp:428-507), createCovisibilityGraph's votes and ranking (:559-647), getSceneDepth /
 *           getSceneDistance (src/frame.cpp:323-366: upper medians and the minimum depth).
 *      Per sequence one hso_seq_job goes in and one hso_seq_result comes back; the caller keeps the keyframe-rate bookkeeping
 *      (promotion, local BA, seeds) and mirrors the events. ---- */

/* hso_map_point.pad_ of a sequence map = the point's state word */
#define HSO_PT_KEY(w)     ((uint32_t)(w) & 0xffu)            /* (Point::type_ << 4) | Point::ftr_type_; type 0 = TYPE_DELETED */
#define HSO_PT_NFAIL(w)   (((uint32_t)(w) >> 8) & 0x3ffu)    /* n_failed_reproj_, saturating at 1023 */
#define HSO_PT_BAD        (1u << 18)                         /* a temporary point given up (isBad_) */
#define HSO_PT_KEEP_NFAIL (1u << 19)                         /* in a PATCHED row only: keep the device's n_failed_reproj_ ... */
#define HSO_PT_NOK(w)     (((uint32_t)(w) >> 20) & 0x7ffu)   /* n_succeeded_reproj_, saturating at 2047 */
#define HSO_PT_KEEP_NOK   (1u << 31)                         /* ... / n_succeeded_reproj_ (the device counts them; the caller owns key and bad flag,
                                                                which it mirrors from the chain's events, and resets a counter by sending it) */
#define HSO_PT_WORD(key, n_fail, bad, n_ok) ((int32_t)(((uint32_t)(key) & 0xffu) | (((uint32_t)(n_fail) & 0x3ffu) << 8) | ((bad) ? HSO_PT_BAD : 0u) | (((uint32_t)(n_ok) & 0x7ffu) << 20)))

enum { HSO_LIST_CANDIDATES = -1 };   /* hso_seqmap_list_patch.list: MapPointCandidates::candidates_ in list order; >= 0: Frame::fts_ of that keyframe row */
typedef struct hso_seqmap_list_patch {
  int32_t map, list;
  int32_t first, n;            /* entries [first, first + n) are written; the list's length becomes first + n */
  const int32_t* ids;          /* feature (= observation) rows of a keyframe's list; point rows of the candidate list */
} hso_seqmap_list_patch;
/* fts_cap: capacity of a keyframe's feature list (its own features + the features of its seeds that became points:
 * max(2000, max_fts) + max_fts + 100 in the reference's configuration).  Once per map, before the first list patch. */
int hso_gpu_seqmap_configure(hso_gpu_ctx* ctx, int map, int fts_cap);
int hso_gpu_seqmap_patch_lists(hso_gpu_ctx* ctx, const hso_seqmap_list_patch* patches, int n_patches);
/* Feature::point of observation rows (the row of the point a keyframe feature observes, -1: none) and Frame::key_pts_ of the
 * keyframes as point rows (5 per keyframe row, -1: none), the two links the chain follows that the tables above do not carry */
int hso_gpu_seqmap_patch_links(hso_gpu_ctx* ctx, int map, const int32_t* obs_ids, const int32_t* obs_point, int n_obs);
int hso_gpu_seqmap_set_key_points(hso_gpu_ctx* ctx, int map, const int32_t* key_points /* 5 * n_kfs */, int n_kfs);

#define HSO_SEQ_MAX_VISIT 24
#define HSO_SEQ_MAX_KFS 2048      /* rows of a sequence map's keyframe table the chain accepts (the reference keeps Config::maxNKfs() = 2000) */
#define HSO_SEQ_MAX_COVIS 8
#define HSO_SEQ_EVENTS 120
enum { HSO_EV_ERASE_POINT = 1,       /* Map::safeDeletePoint (a TYPE_UNKNOWN point failed more than 15 times, reprojector.cpp:376-381) */
       HSO_EV_ERASE_CANDIDATE = 2,   /* MapPointCandidates::deleteCandidatePoint (more than 30 failures, :382-386, :214-222) */
       HSO_EV_TEMP_BAD = 3,          /* a temporary point's isBad_ (:387-390, :247-251) */
       HSO_EV_GOOD = 4 };            /* TYPE_UNKNOWN -> TYPE_GOOD (:412-416) */
enum { HSO_SEQ_DEPTH_STATS = 4,      /* the frame will be a keyframe whatever the flow says (the frame after the initialisation): form depth_median / dist_median / depth_min.
                                        Without the flag they are formed when needNewKf's flow score comes within 10 % of its threshold, else depth_min = -1 */
       HSO_SEQ_NO_TRACK = 1,         /* hso_seq_job.flags: the reference frame has no features: CoarseTracker::run returns 0 at once (CoarseTracker.cpp:53-54) */
       HSO_SEQ_SEED_BRANCH = 2 };    /* the sequence has seeds: with fewer than 100 matches reprojectMap goes on to match them (src/reprojector.cpp:309-329),
                                        the frame's features change and the pose is optimised after that — by the caller: the chain's pose
                                        result is then informative only and its culling is NOT applied to the frame's feature table */

typedef struct hso_seq_job {
  int32_t map;                 /* the sequence map */
  int32_t flags;               /* HSO_SEQ_* */
  int64_t ref_frame_id;        /* the frame the tracker aligns against; resident */
  int64_t cur_frame_id;        /* the new frame; resident */
  hso_se3 T_ref_w;             /* ref.T_f_w_ */
  hso_se3 T_cur_w;             /* cur.T_f_w_ as processFrame sets it before tracking: motion model * last pose (:176) */
  double ref_exposure;         /* ref.m_exposure_time */
  int32_t ref_kf_row;          /* >= 0: the reference frame is this keyframe of the map (its fts list is the feature table);
                                  -1: it is the frame whose features the previous chain call (or hso_gpu_seq_set_frame_features) left */
  int32_t n_ref_feats;         /* ref.fts_.size() (the caller knows it: the previous result's n_feats, or the keyframe list's length) */
  int32_t cur_keyframe_id;     /* cur.keyFrameId_ */
  int32_t last_kf_row;         /* Map::lastKeyframe() (needNewKf's keyframe), -1: skip the flow sums */
  int32_t covis[5];            /* ref.connectedKeyFrames as keyframe rows in list order, -1 padded (reprojector.cpp:124-170) */
  int32_t temps_begin, n_temps;/* this job's slice of the call's `temps` array: the temporary points to list (not bad, positions already placed) */
  float exposure_rat;          /* cur.integralImage_ / ref.integralImage_: the tracker's initial exposure ratio (src/CoarseTracker.cpp:60) */
  int32_t seed_group;          /* the sequence's group in cfg.seed_table, or -1: its seeds are not observed by this call */
  int32_t pad_;
} hso_seq_job;

typedef struct hso_seq_chain_cfg {
  hso_track_params track;      /* CoarseTracker's constructor arguments; every job of a call runs the same mode */
  int32_t cell_size, grid_n_cols, n_cells, max_fts;   /* Reprojector::initializeGrid; Config::maxFts() */
  const int32_t* cell_order;   /* host, n_cells entries */
  int32_t max_kfs;             /* Reprojector::Options::max_n_kfs (10) */
  int32_t pose_n_iter;         /* 12 */
  double pose_reproj_thresh;   /* Config::poseOptimThresh() */
  int32_t quality_min_fts;     /* Config::qualityMinFts(): with fewer matches processFrame gives the frame up before the pose
                                  optimiser's result is used (frame_handler_mono.cpp:224-230) — its culling is then not applied */
  int32_t want_debug;          /* 1: keep the intermediate tables for hso_gpu_debug_fetch (recorded runs) */
  /* DepthFilter::addFrame -> updateSeeds (src/depth_filter.cpp:136-144, 330-509) chained behind the frame for the sequences whose
   * frame is a regular one — at least quality_min_fts matches and inliers, not the seed branch, no keyframe (the chain evaluates
   * needNewKf's score itself: hso_seq_result.make_kf): one observation of every live seed of their groups in the resident table, with
   * the pose the frame ends with and its exposure; the briefs of ALL slots (zero for groups that sat out) come back with the results.
   * seed_table < 0: no observation (the caller observes). */
  int32_t seed_table, n_seed_groups;
  double px_error_angle;       /* DepthFilter::px_error_angle_ */
  hso_seed_brief* seed_brief_out;  /* host, seed_brief_cap entries >= the table's slots */
  int32_t seed_brief_cap, pad_;
} hso_seq_chain_cfg;

typedef struct hso_seq_result {
  hso_track_result track;
  hso_pose_result pose;
  hso_se3 T_tracked;           /* cur.T_f_w_ after CoarseTracker::run (the pose the reprojection used) */
  double exposure;             /* cur.m_exposure_time */
  int32_t counts[4];           /* n_trials_, n_matches_, cell passes, branch (hso_gpu_reproject_select) */
  int32_t n_feats;             /* features of the frame = candidates that became features, in fts_ order */
  int32_t n_listed, n_kf_points, n_candidates;   /* the list: keyframe points | candidates | temporary points */
  int32_t n_visit, visit[HSO_SEQ_MAX_VISIT];     /* keyframe rows whose points were listed, in visiting order */
  float flow_full, flow_shift; /* needNewKf: the two float sums over the last keyframe's features with a point (serial, in list order) */
  int32_t flow_count;
  int32_t n_with_point;        /* features of the frame that still have a point after the pose optimiser's culling */
  int32_t n_covis;             /* keyframes observing at least one of the frame's points */
  int32_t covis[HSO_SEQ_MAX_COVIS], covis_votes[HSO_SEQ_MAX_COVIS];   /* the ranking of createCovisibilityGraph: keyframe rows, best first */
  int32_t covis_best;          /* the keyframe row with the most votes (the fallback when none reaches the bar) */
  int32_t make_kf;             /* needNewKf as the chain evaluated it (frame_handler_mono.cpp:486-506) or HSO_SEQ_DEPTH_STATS: the frame becomes a keyframe */
  int32_t seeds_observed;      /* 1: the seeds of the job's group were observed in this frame (cfg.seed_table) */
  double depth_median, dist_median, depth_min;   /* getSceneDepth / getSceneDistance over the frame's points; depth_min = DBL_MAX when none */
  int32_t n_events;            /* kind changes of points, in the order the reference makes them; > HSO_SEQ_EVENTS: fetch them all with hso_gpu_seq_events */
  int32_t events[HSO_SEQ_EVENTS];   /* (HSO_EV_* << 28) | point row */
} hso_seq_result;

/* temps: the jobs' temporary-point lists back to back (host; may be NULL when n_temps_total == 0) */
int hso_gpu_seq_chain(hso_gpu_ctx* ctx, const hso_camera* cam, const hso_seq_chain_cfg* cfg, const hso_seq_job* jobs, int n_jobs,
                      const int32_t* temps, int n_temps_total, hso_seq_result* results);
/* every event of job `job` of the last chain call (n_events of them) */
int hso_gpu_seq_events(hso_gpu_ctx* ctx, int job, int32_t* events_out, int cap);

/* ---- ba::LocalBundleAdjustment on a sequence map (src/bundle_adjustment.cpp:577-892; the resident form of hso_gpu_ba_local_multi,
 *      SURVEY.md section 8(f)).  The window is assembled on the device from the map's own tables, optimised there and written back
 *      there; the caller names the core keyframes and receives what its mirror of the map needs.  Per job:
 *        1. the graph (:592-812): vertices 0 .. n_core-1 = the core keyframes in the caller's order; the window's points = every point
 *           a feature of a core keyframe's list observes (Feature::point != NULL), each once, ascending by point row (the reference
 *           walks std::set<Point*> in address order; rows give a run-independent one); per point in that order its host keyframe and
 *           then, along the observation list, every observing keyframe that is not the host gets the next vertex number on its first
 *           appearance (fixed, :700-737) and every such observation one edge (hso_ba_edge as the value-passing calls take it:
 *           fH = the point row's host_f, meas = project2d(obs f) — an edgelet's: grad^T of it —, information 1 / 4^level);
 *        2. the Huber deltas of the window's initial state (:618-680) and the Levenberg loop (:815-823): hso_gpu_ba_local_multi's
 *           kernels and driver on the tables of step 1, bit for bit what that call returns for the same tables;
 *        3. the write-back (:826-853): the core keyframes' poses (the library's keyframe table of the map and core_pose), idist_ of the
 *           window's points and pos_ = T_host^-1 * (host_f * (1 / idist)) in the point rows and in point_state;
 *        4. the inputs of the culling (:855-892): the observation rows of the edges whose chi2 exceeds chi2_corner / chi2_edgelet, corner
 *           edges first (edge order), then edgelet edges — the caller removes them from its tables (removePtFrameRef walks the pointer
 *           graph it owns) and patches the rows that changes.
 *      A window without edges or points (status 1) is left alone (the reference would optimise an empty graph). ---- */
#define HSO_SEQ_BA_MAX_CORE 16   /* the dense reduced system holds 16 free poses; the reference's core is Config::coreNKfs() (7) + 2 */
typedef struct hso_seq_ba_job {
  int32_t map;
  int32_t n_core;
  int32_t core[HSO_SEQ_BA_MAX_CORE];    /* keyframe rows of the map, in vertex order */
  uint8_t fixed[HSO_SEQ_BA_MAX_CORE];   /* v->setFixed of a core keyframe (:595) */
  int32_t n_iter;
  int32_t point_cap;                    /* entries point_ids / point_state hold: at least the window's points (<= the sum of the core lists' lengths) */
  int32_t cull_cap;                     /* entries culled holds */
  int32_t pad_;
  int32_t* point_ids;                   /* out: the window's points (rows), ascending */
  double* point_state;                  /* out [4 * point_cap]: idist, pos[3] of every window point after the optimisation (what the rows hold) */
  int32_t* culled;                      /* out: observation rows, in removal order; entries past cull_cap are dropped (n_culled says how many there were) */
} hso_seq_ba_job;
typedef struct hso_seq_ba_result {
  hso_ba_result lm;
  float huber_corner, huber_edge;       /* the deltas used */
  int32_t status;                       /* 0: optimised; 1: no edges or no points, nothing done (point_ids / n_points still returned) */
  int32_t n_poses, n_points, n_edges;
  int32_t n_culled[2];                  /* corner edges, edgelet edges */
  hso_se3 core_pose[HSO_SEQ_BA_MAX_CORE];   /* T_f_w of the core keyframes after the optimisation */
} hso_seq_ba_result;
/* Every map at most once per call; its patches must have been sent.  error_multiplier2 = cam->errorMultiplier2(). */
int hso_gpu_seq_local_ba(hso_gpu_ctx* ctx, const hso_seq_ba_job* jobs, int n_jobs, double error_multiplier2, double chi2_corner,
                         double chi2_edgelet, hso_seq_ba_result* results);

/* A frame's features as the chain keeps them for the sequence's newest frame (Frame::fts_ of a frame that is not a keyframe) */
typedef struct hso_seq_feature {
  double px[2];                /* Feature::px */
  double f[3];                 /* Feature::f */
  float grad[2];               /* Feature::grad (edgelets) */
  int32_t point;               /* Feature::point as a point row, -1: NULL */
  int8_t level, type;          /* Feature::level, Feature::type */
  int16_t pad_;
} hso_seq_feature;
/* read the feature tables the last chain call left for the frames of `maps` (a keyframe promotion needs them on the host):
 * out holds n_maps rows of cap features; n_out[i] = the table's length.  The frame ids must match what the maps hold. */
int hso_gpu_seq_frame_features(hso_gpu_ctx* ctx, const int32_t* maps, const int64_t* frame_ids, int n_maps, hso_seq_feature* out, int cap, int32_t* n_out);
/* replace the table a map holds (the caller changed the frame's features: the seed branch of reprojectMap, a two-view start) */
int hso_gpu_seq_set_frame_features(hso_gpu_ctx* ctx, int map, int64_t frame_id, const hso_seq_feature* feats, int n);
/* recorded runs / parity read-backs of the chain (hso_gpu_seq_debug_*, hso_gpu_debug_fetch, hso_gpu_seqmap_debug_dump) and the developer
 * census: include/hso_gpu_debug.h */

/* ---- FeatureExtractor::fastDetect, src/feature_detection.cpp:518-587 (fastDetectST per level, fastDetect) (SURVEY.md section 8f rank 1,
 *      first stage): FAST-9 corners of pyramid levels 0..n_levels-1 — fast_corner_detect_9_sse2,
 *      fast_corner_score_9, fast_nonmax_3x3 (thirdparty/fast/src) — and hso::shiTomasiScore
 *      (src/vikit/vision.cpp:111-151) of the survivors ---- */
typedef struct hso_corner {
  int16_t x, y;      /* level coordinates (fast::fast_xy); KeyPoint position = (x << L, y << L) */
  int32_t score;     /* fast_corner_score_9 */
  float response;    /* shiTomasiScore, the KeyPoint response */
} hso_corner;

/* Level L's non-max-suppressed corners inside the border (x < border || x > W_L - border ||
 * y < border || y > H_L - border are dropped, :573), in raster order — the order fastDetect pushes
 * them into featurePerLevel_[L], i.e. the candidate index.  out holds n_levels * cap entries (level
 * L at out + L * cap); counts[L] = corners found on level L (if > cap only the first cap are
 * written).  Integer / byte arithmetic throughout: results are bit-identical to the reference
 * library (tests/golden/fast9.json was produced by it). */
int hso_gpu_fast_detect(hso_gpu_ctx* ctx, int64_t frame_id, int n_levels, int threshold, int border,
                        hso_corner* out, int cap, int32_t* counts);
/* The new keyframes of n independent sequences (one frame size) in three launches per level.
 * out: n_frames * n_levels * cap entries, frame-major ((i * n_levels + L) * cap); counts likewise
 * n_frames * n_levels.  cap == 0 (out may be NULL) only counts. */
int hso_gpu_fast_detect_batch(hso_gpu_ctx* ctx, const int64_t* frame_ids, int n_frames, int n_levels, int threshold,
                              int border, hso_corner* out, int cap, int32_t* counts);

/* ---- FeatureExtractor::detect, non-init branch up to the oct-tree: fastDetectMT + edgeLetDetectMT
 *      (src/feature_detection.cpp:408-447, 518-545, 749-830; SURVEY.md section 8f rank 1, second
 *      stage).  Per level L < n_levels (the reference runs 3, one thread each):
 *        1. fastDetectST: the corners of hso_gpu_fast_detect with threshold = min_thresh and
 *           border 8; each one marks haveFeatures_[L][getCellIndex(x, y, L)]
 *           (include/hso/feature_detection.h:295-299 — x is divided by the number of grid rows,
 *           as written there);
 *        2. edgeLetDetectST: cv::Canny(sobelX_[L], sobelY_[L], edges, 31*min_thresh, 70*min_thresh,
 *           L2gradient = true) on the resident Sobel-5 images, then every grid index without a
 *           feature keeps the edge pixel with the largest sqrtf(gx^2 + gy^2) of its cell window
 *           (origin (index % gridCols * g, index / gridRows * g), :769-770, clipped to the 8-pixel
 *           border; first maximum in raster order wins).
 *      Levels with Sobel images only (n_levels <= HSO_N_SOBEL_LEVELS). ---- */
typedef struct hso_edgelet {
  int16_t x, y;      /* level coordinates; KeyPoint position = (x << L, y << L) */
  int16_t gx, gy;    /* Sobel-5 gradient at (x, y): KeyPoint::gx, gy -> Feature::grad = normalised */
  float grad;        /* sqrtf(gx*gx + gy*gy), the KeyPoint response */
} hso_edgelet;

/* corners / corner_counts as in hso_gpu_fast_detect_batch (frame-major, cap entries per level);
 * edgelets likewise ((i * n_levels + L) * edgelet_cap), in grid-index order — the order
 * edgeLetDetectST pushes them into featurePerLevel_[L] after the corners.  Integer arithmetic and
 * one correctly rounded sqrtf: bit-identical to the oracle. */
int hso_gpu_detect_candidates(hso_gpu_ctx* ctx, const int64_t* frame_ids, int n_frames, int n_levels, int min_thresh,
                              hso_corner* corners, int corner_cap, int32_t* corner_counts,
                              hso_edgelet* edgelets, int edgelet_cap, int32_t* edgelet_counts);
/* The same with one minThresh_ per frame (FeatureExtractor::detect takes Frame::gradMean_ as its barrier,
 * src/feature_detection.cpp:408-445: the keyframes of different sequences have different ones): min_thresh[n_frames], each in
 * [0, 255].  One call for the keyframes a bank of sequences takes in a step; every frame's lists equal the single-barrier call's. */
int hso_gpu_detect_candidates_multi(hso_gpu_ctx* ctx, const int64_t* frame_ids, int n_frames, int n_levels, const int32_t* min_thresh,
                                    hso_corner* corners, int corner_cap, int32_t* corner_counts,
                                    hso_edgelet* edgelets, int edgelet_cap, int32_t* edgelet_counts);

/* The initialisation branch of FeatureExtractor::detect (:439-442): fastDetectMT as above, then
 * fillingHole on level 0 (src/feature_detection.cpp:1125-1154) — fast_corner_detect_plain_12,
 * fast_corner_score_12, fast_nonmax_3x3 (thirdparty/fast/src) at barrier max(0.6 * min_thresh, 6),
 * border 8, and a survivor is kept only if its grid index holds no FAST-9 feature or earlier
 * survivor (raster order).  fill: n_frames * fill_cap entries (species kGrad, level 0: position,
 * FAST-12 score, Shi-Tomasi response), fill_counts[n_frames].  Bit-identical to the reference
 * library (tests/golden/fast12.json was produced by it). */
int hso_gpu_detect_candidates_init(hso_gpu_ctx* ctx, const int64_t* frame_ids, int n_frames, int n_levels, int min_thresh,
                                   hso_corner* corners, int corner_cap, int32_t* corner_counts,
                                   hso_corner* fill, int fill_cap, int32_t* fill_counts);

/* ---- FeatureExtractor::computeKeyPointsOctTree, src/feature_detection.cpp:833-1122 (ExtractorNode::DivideNode,
 *      include/hso/feature_detection.h:217-272): the spatial distribution that ends FeatureExtractor::detect
 *      (:449-455).  Host code (sequential refinement over a few thousand candidates), no context needed. ---- */
enum { HSO_KP_CORNER_HIGH = 0, HSO_KP_EDGELET = 1, HSO_KP_GRAD = 2, HSO_KP_OCCUR = 3 };  /* FeatureSpecies, feature_detection.h:180-186 */
typedef struct hso_keypoint {   /* KeyPoint, include/hso/feature_detection.h:188-208 */
  float x, y;          /* level-0 pixels */
  float response;
  int32_t level;
  int32_t species;     /* HSO_KP_* */
  int32_t gx, gy;
} hso_keypoint;

/* keys: allFeturesToDistribute_ in the reference's order — the occupancy keys of the existing
 * features (setExistingFeatures :1169-1177, species HSO_KP_OCCUR) first, then per level the corners
 * followed by the edgelets.  The rectangle is (0, width, 0, height) at the call site; n_features =
 * nFeatures_ (Config::maxFts() + 100, or 2000 while initialising, :382-385).  Returns the number of
 * selected keys (written to out in the order of the reference's node list, at most cap of them) or
 * a negative status.  Where the reference orders equal-sized nodes by their heap addresses
 * (std::sort over (size, pointer) pairs, :974), node creation order is used. */
int hso_gpu_select_octree(const hso_keypoint* keys, int n, int min_x, int max_x, int min_y, int max_y, int n_features,
                             hso_keypoint* out, int cap);

/* ---- two-view initialisation, image side: initialization::trackKlt (src/initialization.cpp:225-300) =
 *      cv::calcOpticalFlowPyrLK(img_prev, img_cur, px_prev, px_cur, status, error, Size(30, 30), 4,
 *      TermCriteria(COUNT + EPS, 30, 0.0001), OPTFLOW_USE_INITIAL_FLOW) and patchCheck (:476-563) per point.  The geometry that
 *      follows (computeInitializeMatrix, :300-385) is host code (hso_amd/host/hso_init.cpp). ---- */
typedef struct hso_klt_params {
  int32_t win_size;          /* 30 (klt_win_size); 3..32 */
  int32_t max_level;         /* 4: the pyramid stops earlier when the next level would not exceed the window (752x480: levels 0..3) */
  int32_t max_iter;          /* 30 (klt_max_iter) */
  int32_t use_initial_flow;  /* 1: px_init is the start (OPTFLOW_USE_INITIAL_FLOW); 0: px_prev is */
  double epsilon;            /* 0.0001 (klt_eps): the stop test is |delta|^2 <= epsilon^2 */
} hso_klt_params;
enum { HSO_KLT_TRACKED = 1,  /* calcOpticalFlowPyrLK's status byte */
       HSO_KLT_PATCH_OK = 2  /* patchCheck: both 8x8 patches inside their images and zero-mean NCC > 0.8 */ };
typedef struct hso_klt_result {
  float px[2];               /* px_cur */
  float ncc;                 /* patchCheck's correlation; -2 when a patch leaves its image */
  int32_t status;            /* HSO_KLT_* bits; trackKlt keeps a point iff both are set */
} hso_klt_result;
/* Both frames resident (same size).  px_prev / px_init: n x 2 floats (host); out: n results (host). */
int hso_gpu_klt_track(hso_gpu_ctx* ctx, int64_t frame_prev, int64_t frame_cur, const float* px_prev, const float* px_init, int n,
                      const hso_klt_params* params, hso_klt_result* out);
int hso_gpu_klt_levels(int width, int height, int win, int max_level);   /* index of the coarsest level the call above uses */
/* static tables of include/hso/CoarseTracker.h:58-120 for a level */
int hso_gpu_tracker_pattern(int max_level, int level, int* patch_area,
                            int* half_patch, int8_t* offsets_xy /* 2*40 */);

#ifdef __cplusplus
}
#endif
#endif /* HSO_GPU_H */
