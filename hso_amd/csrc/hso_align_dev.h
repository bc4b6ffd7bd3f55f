// hso_align_dev.h — the reprojection matcher's job record and launcher, shared by the callers that build jobs on the device
// (hso_align.hip: the projection kernels; hso_activate.hip: the seed activation's (seed, target) pairs).
#pragma once
#include "hso_ctx.h"

struct AlignConsts {
  hso_camera cam;
  PyrGeom g;
  float ncc_thresh = 0.7f;   // checkNCC's threshold: 0.7 in findMatchDirect (src/matcher.cpp:364), 0.8 in findMatchSeed (:509)
};

struct AlignJobDev {
  const uint8_t* ref_base;   // null: no job (the output record stays zero)
  const uint8_t* cur_base;   // the frame this candidate is searched in (jobs of many frames share a launch)
  hso_align_job j;
};

// k_align_t<true> over n device-built jobs (null jobs skipped; d_out must have been zeroed), asynchronous on the context's stream
int hso_align_launch_sparse(hso_gpu_ctx* ctx, const hso_camera* cam, const PyrGeom& g, float ncc_thresh, const AlignJobDev* d_jobs, int n, hso_align_out* d_out);
