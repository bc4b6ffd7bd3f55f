"""`python -m hso_amd.run_sequence <image folder> <stamp file | None> <camera file> [key=value ...]`

The reference's test_dataset harness (test/test_dataset.cpp:260-335: BenchmarkNode::runFromFolder +
saveResult) without OpenCV: read the calibration (incl. the > 848x800 downscale rule), list the images of the
folder, resize every image to the camera size on the device when the sensor is larger, feed
FrameHandlerMono::addImage in order, write the keyframe trajectory in the reference's format.

Options (the reference's `key=value` style, test_dataset.cpp:66-114):
  start=<i> end=<i>        frame range (test/euroc_batch.sh:9 uses start=50 for MH_01)
  depth0=<file.npy|.f32>   optical-axis depth image of the first frame: the initial map comes from it (hso_vo_set_first_frame)
                           instead of the two-view initialisation (hso_vo_start, the reference's own start)
  max_fts=<n>              Config::maxFts() (200)
  result=<path>            trajectory file (default ./result/KeyFrameTrajectory.txt)
  gt=<trajectory file>     ground truth in the same format: prints the ATE (RMSE after similarity alignment)
  trace=<path>             record every device call of the run (trace_frames=<n>: only of the first n frames)
  times=1                  print per-frame wall time

Layout of a sequence folder = the reference's: <folder>/*.png (or .pgm), one stamp per line in the stamp file."""
import os
import sys
import time

import numpy as np

HAS_TWO_VIEW_INIT = True       # hso_vo_start: KLT + essential matrix / homography (hso_amd/host/hso_init.h)


def load_image(path):
    from . import formats
    return formats.read_pgm(path) if path.lower().endswith(".pgm") else formats.read_png(path)


def main(argv, return_line=False):
    if len(argv) < 3:
        print(__doc__)
        return 2
    folder, stamp_file, cam_file = argv[:3]
    opt = dict(a.split("=", 1) for a in argv[3:] if "=" in a)
    from . import capi, formats, vo
    calib = formats.parse_calibration(cam_file)
    cam = calib["camera"]
    W, H = calib["width"], calib["height"]
    files = formats.list_images(folder) or sorted(os.path.join(folder, n) for n in os.listdir(folder) if n.lower().endswith(".pgm"))
    stamps = formats.read_stamps(stamp_file) if stamp_file not in ("None", "none", "") else None
    start, end = int(opt.get("start", 0)), min(int(opt.get("end", len(files))), len(files))
    d0 = None
    if "depth0" in opt:
        d0 = np.load(opt["depth0"]) if opt["depth0"].endswith(".npy") else np.fromfile(opt["depth0"], np.float32).reshape(H, W)
    odo = vo.VisualOdometry(cam, int(opt.get("max_fts", 200)))
    if "trace" in opt:
        odo.trace(opt["trace"])
    resize_ctx = None

    def prepare(img):
        # ImageReader::readImage: cv::resize to the camera size (src/ImageReader.cpp:79), on the device
        nonlocal resize_ctx
        if img.shape == (H, W):
            return img
        if resize_ctx is None:
            resize_ctx = capi.Context(0)
        resize_ctx.frame_upload_resized(1, img, W, H)
        out = resize_ctx.frame_level(1, 0, W, H)
        resize_ctx.frame_release(1)
        return out

    rows, t_frames, n_fail, n_init = [], [], 0, 0
    if d0 is None:
        odo.start()                                     # vo_->start(), test/test_dataset.cpp:276
    trace_frames = int(opt.get("trace_frames", 0))
    for k, i in enumerate(range(start, end)):
        img = prepare(load_image(files[i]))
        if trace_frames and k == trace_frames:
            odo.trace(None)
        t0 = time.perf_counter()
        if k == 0 and d0 is not None:
            odo.set_first_frame(img, d0, float(i))
            st = odo.status()
        else:
            st = odo.add_image(img, float(i))          # vo_->addImage(image, img_id, &time_stamp)
            if st.stage == 0 and d0 is None:            # the initialisation failed and paused the handler: start over, like a user of the reference would
                odo.start()
            initialising = st.stage in (0, 1, 2) and d0 is None and n_init == k
            n_init += int(initialising)
            n_fail += int(not initialising and (st.result == 2 or st.stage != 3))     # RESULT_FAILURE / not STAGE_DEFAULT_FRAME
        t_frames.append(time.perf_counter() - t0)
        if opt.get("times"):
            print("frame %d  %.2f ms  kf=%d stage=%d obs=%d matches=%d seeds=%d" % (i, 1e3 * t_frames[-1], st.is_keyframe, st.stage,
                                                                                   st.n_inliers, st.n_matches, st.n_seeds))
    for ts, T, fid in odo.keyframes():
        # the reference names a pose by the frame's line of the stamp file; EuRoC images are named by their stamp, so without a
        # stamp file the image's base name serves
        name = stamps[int(ts)] if stamps is not None and int(ts) < len(stamps) else os.path.splitext(os.path.basename(files[int(ts)]))[0]
        rows.append((name, tuple(T.q[:]), tuple(T.t[:])))
    result = opt.get("result", os.path.join("result", "KeyFrameTrajectory.txt"))
    os.makedirs(os.path.dirname(os.path.abspath(result)), exist_ok=True)
    formats.write_trajectory(result, rows)
    line = {"frames": end - start, "keyframes": len(rows), "frames_per_s": (len(t_frames) - 1) / max(sum(t_frames[1:]), 1e-9),
            "tracking_failures": n_fail, "init_frames": n_init, "result": result}
    if "gt" in opt:
        (es, exyz, _), (gs, gxyz, _) = formats.read_trajectory(result), formats.read_trajectory(opt["gt"])
        common = [s_ for s_ in es if s_ in set(gs)]
        if len(common) >= 3:
            rmse, scale, _, _ = formats.ate_rmse(gxyz[[gs.index(s_) for s_ in common]], exyz[[es.index(s_) for s_ in common]])
            line["ate_rmse"], line["ate_scale"], line["ate_keyframes"] = rmse, scale, len(common)
    print(line)
    odo.close()
    return line if return_line else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
