// hso_tracker_defs.h — what the translation units of the tracker share (hso_tracker.hip: the batch shapes trk1 / trk2 and
// the host side; hso_tracker_coop.hip: the cooperative shape that splits ONE job across workgroups): the static patterns of
// include/hso/CoarseTracker.h, the kernel argument blocks and the per-workgroup scratch layout.
#ifndef HSO_TRACKER_DEFS_H
#define HSO_TRACKER_DEFS_H
#include "hso_ctx.h"
#include "hso_dev_math.h"

#define TRK_MAX_PA 25
#define KEY_INVALID 0xFFFFFFFFu
#define N_RED 38  // 28 H + 7 b + E + n_terms + n_saturated

// include/hso/CoarseTracker.h:58-120 (staticPattern, staticPatternNum, staticPatternPadding)
static constexpr int8_t h_pattern[8][40][2] = {
  { {0,0} },
  { {0,-1}, {-1,0}, {0,0}, {1,0}, {0,1} },
  { {-1,-1}, {-1,0}, {-1,1}, {-1,0}, {0,0}, {0,1}, {1,-1}, {1,0}, {1,1} },
  { {0,-2}, {-1,-1}, {1,-1}, {-2,0}, {0,0}, {2,0}, {-1,1}, {1,1}, {0,2}, {0,-1}, {-1,0}, {1,0}, {0,1} },
  { {0,-2}, {-1,-1}, {1,-1}, {-2,0}, {0,0}, {2,0}, {-1,1}, {1,1}, {0,2}, {-2,-2}, {-2,2}, {2,-2}, {2,2} },
  { {0,-2}, {-1,-1}, {1,-1}, {-2,0}, {0,0}, {2,0}, {-1,1}, {1,1}, {0,2}, {-2,-2}, {-2,2}, {2,-2}, {2,2},
    {-3,-1}, {-3,1}, {3,-1}, {3,1}, {1,-3}, {-1,-3}, {1,3}, {-1,3} },
  { {-2,-2}, {-2,-1}, {-2,0}, {-2,1}, {-2,2}, {-1,-2}, {-1,-1}, {-1,0}, {-1,1}, {-1,2},
    {0,-2}, {0,-1}, {0,0}, {0,1}, {0,2}, {1,-2}, {1,-1}, {1,0}, {1,1}, {1,2},
    {2,-2}, {2,-1}, {2,0}, {2,1}, {2,2} },
  { {-4,-4}, {-4,-2}, {-4,0}, {-4,2}, {-4,4}, {-2,-4}, {-2,-2}, {-2,0}, {-2,2}, {-2,4},
    {0,-4}, {0,-2}, {0,0}, {0,2}, {0,4}, {2,-4}, {2,-2}, {2,0}, {2,2}, {2,4},
    {4,-4}, {4,-2}, {4,0}, {4,2}, {4,4} },
};
static constexpr int h_pattern_num[8] = { 1, 5, 9, 13, 13, 21, 25, 25 };
static constexpr int h_pattern_pad[8] = { 1, 1, 1, 2, 2, 3, 2, 4 };
#define PATTERN_OFFSET 2  // m_pattern_offset, CoarseTracker.h:122

// per pyramid level: geometry, PATCH_AREA, HALF_PATCH_SIZE and the byte offset oy*stride+ox of
// every pattern pixel in that level's image (CoarseTracker.cpp:80-82,337).  Lives in device
// memory: indexing a by-value kernel argument with the run-time level would make the compiler
// copy the whole argument block to scratch and read the camera from there in the hot loops.
struct TrackLevel {
  int w, h;
  uint32_t off;       // byte offset of the level in the frame's pyramid block
  int pa, pad;
  int pi;             // index into the static pattern tables (max_level - level + 2), -1 = none
  int poff[TRK_MAX_PA];
};

// kernel argument by value: scalars only ever indexed with constants => stays in SGPRs
struct TrackConsts {
  hso_camera cam;
  int inverse, max_level, min_level, n_iter;
  int level_first, level_last;  // the levels this launch works through (max_level .. min_level, or a part of it: see track_launch)
  int resume;         // != 0: start from the pose / exposure / bookkeeping an earlier launch left in the result record
  int lds_img_cap;    // bytes of LDS available for the staged level image
  int n_max;          // scratch stride (features)
  int keys_in_memory; // parity hook: leave the |residual| keys in the scratch buffer (abs_err_out)
  const TrackLevel* lv;  // [HSO_N_PYR_LEVELS], device memory
};

struct TrackJobDev {
  const uint8_t* ref_base;
  const uint8_t* cur_base;
  const double* feats;  // SoA [6][n_stride]: px, py, fx, fy, fz, dist
  int n, n_stride;
  hso_se3 T;
  float a;
  int n_total;          // features of the whole job (= n except for the slices of the cooperative shape)
  int coop_K, coop_pad_; // cooperative shape: workgroups that share this job
};

// per-workgroup scratch in global memory (L2 resident): sized for n_max features
struct Scratch {
  float* ref_patch;   // [PA][n_max]  reference intensities (m_ref_patch_cache, pixel-major)
  float* ref_dx;      // [PA][n_max]  inverse-compositional: reference image gradients
  float* ref_dy;
  uint32_t* keys;     // [PA*n_max]   |residual| bit patterns for the robust thresholds
  uint8_t* visible;   // [n_max]      m_visible_fts
};

HSO_HD size_t scratch_bytes(int n_max)
{
  const size_t t = (size_t)TRK_MAX_PA * n_max;
  return ((t * 4 * 4 + n_max + 255) / 256) * 256;
}
HSO_HD Scratch scratch_at(char* base, int n_max)
{
  const size_t t = (size_t)TRK_MAX_PA * n_max;
  Scratch s;
  s.ref_patch = reinterpret_cast<float*>(base);
  s.ref_dx = s.ref_patch + t;
  s.ref_dy = s.ref_dx + t;
  s.keys = reinterpret_cast<uint32_t*>(s.ref_dy + t);
  s.visible = reinterpret_cast<uint8_t*>(s.keys + t);
  return s;
}

// ---- cooperative shape (hso_tracker_coop.hip): several workgroups share one job; see "exchange between the workgroups of
// one job" in hso_tracker_core.h.  One CoopJobState per job in device memory, zeroed by the launch's memset.
#define COOP_KMAX 32            // workgroups per job (the CUs of one XCD)
#define COOP_MAX_JOBS 8         // one job per XCD
#define COOP_REGIONS 32         // histogram / candidate-list regions a job may consume: <= 6 per level x 5 levels
#define COOP_REGION_WORDS 4096
#define COOP_SPIN_LIMIT (1u << 22)
#ifndef COOP_FEATS_PER_WG
#define COOP_FEATS_PER_WG 256   // features per workgroup the host aims for when it picks K: one feature per thread of the
                                // 512-thread shape's busier half.  Measured on 2000 EuRoC features (profiles/r3_latency.txt):
                                // K = 32 / 16 / 8 / 4 workgroups -> 0.534 / 0.495 / 0.487 / 0.497 ms per launch (one workgroup:
                                // 0.832); below ~256 features the exchange costs more than the split saves (200 features:
                                // 0.404 ms on one workgroup, 0.49 on four)
#endif
// the partial sums one workgroup publishes per evaluation, as 32-bit granule values: the 38 doubles of Shared::red
#define COOP_NG (2 * N_RED)
struct CoopRegion {
  unsigned arrive, count, extra, pad_[13];
  unsigned w[COOP_REGION_WORDS];
};
struct CoopJobState {
  unsigned long long gran[2][COOP_KMAX][COOP_NG + 2];   // {tag, value} granules of the partial sums, two alternating sets
  unsigned long long hello[COOP_KMAX];                  // {1, XCC id} of every workgroup (placement census, see coop_hello)
  CoopRegion region[COOP_REGIONS];
};
size_t hso_track_coop_lds_bytes();
int hso_track_coop_img_cap();
// launches the cooperative kernel for n_jobs <= COOP_MAX_JOBS jobs; job j runs on subjobs[j * k_stride + r].coop_K workgroups
// r = 0 .. K_j - 1 (K_j <= k_stride <= CUs per XCD: every workgroup must be resident), `state` holds n_jobs zeroed CoopJobState,
// `fail_flag` a zeroed word.  scatter != 0 places a job's workgroups on consecutive blocks (different XCDs): a test knob for
// the placement-independent transport.
hipError_t hso_track_coop_launch(hipStream_t stream, const TrackConsts& C, const TrackJobDev* subjobs, int n_jobs, int k_stride,
                                 int scatter, CoopJobState* state, unsigned* fail_flag, char* scratch, size_t scratch_stride,
                                 hso_track_result* results);
#endif
