// hso_pose_dev.h — what hso_select.hip needs of the pose optimiser (hso_pose.hip) to chain it behind the grid selection
// without leaving the device: the job record the kernel reads and a launcher over job records that already live in device
// memory (feature tables built on the device from the selected matches).
#pragma once
#include "hso_ctx.h"

struct PoseJobDev {
  const hso_pose_feat* feats;   // device
  const hso_se3* poses;         // device: T_f_w of the host keyframes
  uint8_t* mask;                // device, may be null: 1 = the feature's point was culled (:698-760)
  int n_feats, n_poses;
  hso_se3 T;
  double reproj_thresh;
  int n_iter, _pad;
};

#define HSO_POSE_MAX_FEATS 4096
#define HSO_POSE_MAX_POSES 128
// one workgroup per job; n_max_feats = an upper bound of the jobs' n_feats (selects the features-per-thread instantiation)
int hso_pose_launch_device(hso_gpu_ctx* ctx, const hso_camera* cam, const PoseJobDev* d_jobs, int n_jobs, int n_max_feats,
                           hso_pose_result* d_results);
