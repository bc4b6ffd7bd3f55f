"""Render a stand-in for a staged data set in the reference's folder layout, to exercise tests/test_dataset_gpu.py without the
download (there is no network here):

  python tests/make_standin_dataset.py euroc <out_dir> [n_frames]   -> <out_dir>/cam0/data/<stamp>.png (752x480) + <out_dir>/gt.txt
  python tests/make_standin_dataset.py tum   <out_dir> [n_frames]   -> <out_dir>/images/<k>.png (1280x1024 sensor size) + <out_dir>/gt.txt

then e.g.  HSO_EUROC_MH01=<out_dir>/cam0/data HSO_EUROC_MH01_GT=<out_dir>/gt.txt python -m pytest tests/test_dataset_gpu.py -m gpu -k euroc
           HSO_TUM_SEQ01=<out_dir>/images HSO_TUM_SEQ01_GT=<out_dir>/gt.txt python -m pytest tests/test_dataset_gpu.py -m gpu -k tum
Synthetic scenes (hso_amd/synth.py) with low-frequency texture; the TUM stand-in is rendered at the camera size 920x736 and
enlarged to the sensor's 1280x1024, so the driver's device-side cv::resize path runs.  Round 3 on one MI355X: euroc 270 frames ->
200 frames from start=50, initialisation 32 frames, 14 keyframes, no failure, ATE 5.1 mm, replay of the first 40 frames' device
calls green; the whole run at 2000 features ATE 3.5 mm; tum 200 frames -> 15 keyframes, no failure, ATE 2.0 mm."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hso_amd import formats, synth  # noqa: E402


def main(kind, out, n):
    bands = ((0.004, 0.05), (0.05, 0.6))
    rows = []
    if kind == "euroc":
        S = synth.sequence(n_frames=n, spec=dict(synth.EUROC, texture_om=bands), step=(0.02, 0.006, 0.004), rot_deg_per_frame=(0.03, -0.06, 0.02))
        folder = os.path.join(out, "cam0", "data")
        os.makedirs(folder, exist_ok=True)
        for k, im in enumerate(S["images"]):
            ts = 1403636579763555584 + 50000000 * k
            formats.write_png(os.path.join(folder, "%019d.png" % ts), im)
            rows.append(("%d" % ts, S["T_f_w"][k][0], S["T_f_w"][k][1]))
    else:
        from scipy.ndimage import zoom
        S = synth.sequence(n_frames=n, spec=dict(synth.TUM_WIDE, texture_om=bands), step=(0.03, 0.009, 0.006), rot_deg_per_frame=(0.03, -0.08, 0.02))
        folder = os.path.join(out, "images")
        os.makedirs(folder, exist_ok=True)
        for k, im in enumerate(S["images"]):
            big = np.clip(zoom(im.astype(np.float32), (1024 / 736, 1280 / 920), order=1), 0, 255).astype(np.uint8)
            formats.write_png(os.path.join(folder, "%05d.png" % k), big)
            rows.append(("%05d" % k, S["T_f_w"][k][0], S["T_f_w"][k][1]))
    formats.write_trajectory(os.path.join(out, "gt.txt"), rows)
    print("wrote %d frames under %s" % (n, out))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 270)
