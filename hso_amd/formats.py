"""The data formats on either side of the hot path, OpenCV-free:

  parse_calibration   the camera files of test/cameras/*.txt as BenchmarkNode::BenchmarkNode reads
                      them (reference test/test_dataset.cpp:133-248), including the > 848x800
                      downscale rule with its operator-precedence quirk
  read_stamps         the four time-stamp line formats of ImageReader (src/ImageReader.cpp:24-66)
  list_images         ImageReader::getDir (:88-125): *.png / *.jpg of a folder, sorted
  read_pgm / read_png 8-bit grayscale images without cv::imread (binary PGM; non-interlaced PNG)
  write_trajectory    BenchmarkNode::saveResult (test/test_dataset.cpp:312-335): one keyframe per
                      line, "stamp tx ty tz qx qy qz qw" of T_f_w^-1 (the TUM trajectory format)
  read_trajectory, ate_rmse   the evaluation side: closed-form alignment (Umeyama) + RMSE of positions
  save_snapshot / load_snapshot   the map state in the C-ABI's table layouts, one .npz

Host-side plumbing: nothing here touches the GPU.
"""
import os
import re
import struct
import zlib

import numpy as np

from .capi import CAM_PINHOLE, CAM_FOV, make_camera

G_MAX_RESOLUTION = 848 * 800          # test/test_dataset.cpp:55


def _f32(x):
    return np.float32(x)


def parse_calibration(path_or_text):
    """-> dict(camera=hso_camera, model, width, height, file_width, file_height, undistort).
    Line 1: "Pinhole fx fy cx cy d0 d1 d2 d3" | "Equidistant ..." (9 tokens) or
    "FOV fx fy cx cy omega" (6 tokens); line 2: "width height"; FOV line 3: "true" = undistort the
    image first.  Values pass through float like the reference's sscanf("%f")."""
    text = open(path_or_text).read() if os.path.exists(str(path_or_text)) else str(path_or_text)
    lines = text.splitlines()
    tok = lines[0].split()
    kind = tok[0][0].lower()
    if len(tok) >= 9 and kind in "pe":
        ic = [_f32(t) for t in tok[1:9]]
    elif len(tok) >= 6 and kind == "f":
        ic = [_f32(t) for t in tok[1:6]]
    else:
        raise ValueError("Camera file error.")
    wh = [_f32(t) for t in lines[1].split()[:2]]
    width_i, height_i = int(wh[0]), int(wh[1])
    scale_intrinsics = kind != "f" or (ic[2] > 1 and ic[3] > 1)    # normalised FOV intrinsics scale with the ctor instead
    if float(wh[0] * wh[1]) > G_MAX_RESOLUTION + 0.00000001:
        resize_rate = np.float64(np.sqrt(wh[0] * wh[1] / _f32(G_MAX_RESOLUTION)))   # float operands, result kept in a double
        width_i, height_i = int(np.float64(wh[0]) / resize_rate), int(np.float64(wh[1]) / resize_rate)
        # sic: wh0*wh1/width_i*height_i = ((wh0*wh1)/width_i)*height_i in float (test_dataset.cpp:167)
        resize_rate = np.float64(np.sqrt(wh[0] * wh[1] / _f32(width_i) * _f32(height_i)))
        if scale_intrinsics:
            ic[0:4] = [_f32(np.float64(v) / resize_rate) for v in ic[0:4]]
    if kind == "e":
        raise NotImplementedError("EquidistantCamera is outside the hot path's camera models (pinhole / radtan / FOV)")
    if kind == "p":
        cam = make_camera(CAM_PINHOLE, width_i, height_i, ic[0], ic[1], ic[2], ic[3], d=[float(v) for v in ic[4:8]] + [0.0])
        undistort = False
    else:
        fx, fy, cx, cy = (float(v) for v in ic[0:4])
        if cx < 1.0 and cy < 1.0:                                  # FOVCamera ctor, src/camera.cpp:142-148
            fx, fy, cx, cy = fx * width_i, fy * height_i, cx * width_i, cy * height_i
        undistort = len(lines) > 2 and lines[2].strip() == "true"
        cam = make_camera(CAM_FOV, width_i, height_i, fx, fy, cx, cy, d=[float(ic[4])], distortion=0 if undistort else 1)
    return dict(camera=cam, model={"p": "Pinhole", "f": "FOV"}[kind], width=width_i, height=height_i,
                file_width=int(wh[0]), file_height=int(wh[1]), undistort=undistort)


def read_stamps(path):
    """ImageReader's stamp file (src/ImageReader.cpp:24-66): each line goes through the reference's own sscanf
    cascade — "%s %f %f %f %f %f %f %f" (8 fields), "%d %s %f" (3), "%d %s" (2), "%s" (1), first that converts fully
    wins — by calling libc's sscanf with those very formats, so every corner of its matching ("123.456" alone is
    id 123 + stamp ".456"; inf / nan / hex floats in the first form) is the reference's.  -> list of stamp strings."""
    import ctypes as C
    libc = C.CDLL("libc.so.6")
    out = []
    for line in open(path, "rb").read().split(b"\n"):
        buf = line[:999]                                   # tr.getline(buf, 1000)
        stamp = C.create_string_buffer(1024)
        f = [C.c_float() for _ in range(7)]
        i = C.c_int()
        if libc.sscanf(buf, b"%s %f %f %f %f %f %f %f", stamp, *[C.byref(v) for v in f]) == 8:
            out.append(stamp.value.decode("latin-1"))
        elif libc.sscanf(buf, b"%d %s %f", C.byref(i), stamp, C.byref(f[0])) == 3:
            out.append(stamp.value.decode("latin-1"))
        elif libc.sscanf(buf, b"%d %s", C.byref(i), stamp) == 2:
            out.append(stamp.value.decode("latin-1"))
        elif libc.sscanf(buf, b"%s", stamp) == 1:
            out.append(stamp.value.decode("latin-1"))
    return out


def list_images(folder):
    return [os.path.join(folder, n) for n in sorted(os.listdir(folder)) if ".png" in n or ".jpg" in n]


def read_pgm(path):
    data = open(path, "rb").read()
    m = re.match(rb"P5\s+(?:#[^\n]*\n\s*)*(\d+)\s+(?:#[^\n]*\n\s*)*(\d+)\s+(?:#[^\n]*\n\s*)*(\d+)\s", data)
    if not m:
        raise ValueError("not a binary PGM: %s" % path)
    w, h, maxval = int(m.group(1)), int(m.group(2)), int(m.group(3))
    if maxval > 255:
        raise ValueError("16-bit PGM is not an 8-bit grayscale image")
    return np.frombuffer(data, np.uint8, w * h, m.end()).reshape(h, w).copy()


def write_pgm(path, img):
    img = np.ascontiguousarray(img, np.uint8)
    with open(path, "wb") as f:
        f.write(b"P5\n%d %d\n255\n" % (img.shape[1], img.shape[0]))
        f.write(img.tobytes())


def read_png(path):
    """8-bit grayscale (colour type 0) or gray+alpha (4, alpha dropped) PNG, non-interlaced."""
    data = open(path, "rb").read()
    if data[:8] != b"\x89PNG\r\n\x1a\n":
        raise ValueError("not a PNG: %s" % path)
    pos, idat, hdr = 8, [], None
    while pos < len(data):
        n, kind = struct.unpack(">I4s", data[pos:pos + 8])
        body = data[pos + 8:pos + 8 + n]
        if kind == b"IHDR":
            hdr = struct.unpack(">IIBBBBB", body)
        elif kind == b"IDAT":
            idat.append(body)
        elif kind == b"IEND":
            break
        pos += 12 + n
    w, h, depth, ctype, _, _, interlace = hdr
    if depth != 8 or ctype not in (0, 4) or interlace:
        raise ValueError("only non-interlaced 8-bit grayscale PNGs are read without a decoder library")
    bpp = 1 if ctype == 0 else 2
    raw = zlib.decompress(b"".join(idat))
    stride = w * bpp
    out = np.zeros((h, stride), np.uint8)
    prev = np.zeros(stride, np.int32)
    for y in range(h):
        ft = raw[y * (stride + 1)]
        line = np.frombuffer(raw, np.uint8, stride, y * (stride + 1) + 1).astype(np.int32)
        if ft == 0:
            cur = line
        elif ft == 2:
            cur = (line + prev) & 255
        else:                       # Sub / Average / Paeth depend on the pixel to the left: sequential
            cur = np.zeros(stride, np.int32)
            for x in range(stride):
                a = cur[x - bpp] if x >= bpp else 0
                b = prev[x]
                c = prev[x - bpp] if x >= bpp else 0
                if ft == 1:
                    p = a
                elif ft == 3:
                    p = (a + b) >> 1
                else:
                    pa, pb, pc = abs(b - c), abs(a - c), abs(a + b - 2 * c)
                    p = a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)
                cur[x] = (line[x] + p) & 255
        out[y] = cur
        prev = cur
    return out[:, ::bpp].copy()


def write_png(path, img):
    """8-bit grayscale PNG (filter 0), for fixtures and round trips."""
    img = np.ascontiguousarray(img, np.uint8)
    h, w = img.shape

    def chunk(kind, body):
        return struct.pack(">I", len(body)) + kind + body + struct.pack(">I", zlib.crc32(kind + body) & 0xffffffff)
    raw = b"".join(b"\x00" + img[y].tobytes() for y in range(h))
    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 0, 0, 0, 0)) +
                chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b""))


def _quat_to_R(q):
    x, y, z, w = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def write_trajectory(path, keyframes):
    """keyframes: iterable of (stamp, q_f_w (x, y, z, w), t_f_w).  Writes T_f_w^-1 per line with
    operator<< default precision (6 significant digits), like saveResult."""
    with open(path, "w") as f:
        for stamp, q, t in keyframes:
            R = _quat_to_R(q)
            c = -R.T @ np.asarray(t, float)                # Tinv.translation()
            qi = (-q[0], -q[1], -q[2], q[3])               # Tinv.unit_quaternion()
            f.write("%s %s\n" % (stamp if isinstance(stamp, str) else "%g" % stamp, " ".join("%g" % v for v in (*c, *qi))))


def read_trajectory(path):
    """-> (stamps as strings, positions [n, 3], quaternions [n, 4] (x, y, z, w))."""
    stamps, rows = [], []
    for line in open(path):
        t = line.split()
        if len(t) >= 8 and not t[0].startswith("#"):
            stamps.append(t[0]); rows.append([float(x) for x in t[1:8]])
    a = np.array(rows, float).reshape(-1, 7)
    return stamps, a[:, 0:3], a[:, 3:7]


def ate_rmse(gt_xyz, est_xyz, with_scale=True):
    """Absolute trajectory error after the closed-form similarity (monocular: with_scale) or rigid
    alignment of est onto gt (Umeyama 1991).  -> (rmse, scale, R, t)."""
    gt, est = np.asarray(gt_xyz, float), np.asarray(est_xyz, float)
    mg, me = gt.mean(0), est.mean(0)
    G, E = gt - mg, est - me
    U, D, Vt = np.linalg.svd(G.T @ E / len(gt))
    S = np.eye(3)
    if np.linalg.det(U) * np.linalg.det(Vt) < 0:
        S[2, 2] = -1
    R = U @ S @ Vt
    s = float(np.trace(np.diag(D) @ S) / (E ** 2).sum() * len(gt)) if with_scale else 1.0
    t = mg - s * R @ me
    err = gt - (s * (R @ est.T).T + t)
    return float(np.sqrt((err ** 2).sum(1).mean())), s, R, t


# ---- state snapshot: the map state the per-frame entry points consume, in
# the C-ABI's own table layouts, so a saved state replays through hso_gpu_reproject_match /
# hso_gpu_seed_observe / hso_gpu_coarse_track_batch without any of the reference's containers.
SNAPSHOT_VERSION = 1


def _npz_path(path):
    """np.savez appends '.npz' to a path without it; save and load agree on the final name."""
    path = os.fspath(path)
    return path if path.endswith(".npz") else path + ".npz"


def save_snapshot(path, camera, keyframes, images, points, observations, seeds=None, meta=None):
    """keyframes: KF_DTYPE array; images: one u8 image per keyframe (level 0; the pyramid is rebuilt
    on upload); points / observations: MAP_POINT_DTYPE / OBS_DTYPE arrays (obs_begin / kf indices as
    in include/hso_gpu.h); seeds: optional ctypes array or list of capi.Seed.  One .npz file."""
    from . import capi
    import ctypes as C
    kfs = np.ascontiguousarray(keyframes, capi.KF_DTYPE)
    if len(images) != len(kfs):
        raise ValueError("one image per keyframe")
    cam = np.frombuffer(bytes(camera), np.uint8).copy()
    arrs = dict(version=np.array([SNAPSHOT_VERSION]), camera=cam, keyframes=kfs,
                points=np.ascontiguousarray(points, capi.MAP_POINT_DTYPE),
                observations=np.ascontiguousarray(observations, capi.OBS_DTYPE),
                meta=np.frombuffer(repr(meta or {}).encode(), np.uint8).copy())
    for k, img in enumerate(images):
        arrs["image_%d" % k] = np.ascontiguousarray(img, np.uint8)
    if seeds is not None:
        n = len(seeds)
        sarr = seeds if isinstance(seeds, C.Array) else (capi.Seed * n)(*seeds)
        arrs["seeds"] = np.frombuffer(bytes(sarr), np.uint8).copy()
    np.savez_compressed(_npz_path(path), **arrs)


def load_snapshot(path):
    """-> dict(camera, keyframes, images, points, observations, seeds (ctypes array or None), meta)."""
    from . import capi
    import ast
    import ctypes as C
    z = np.load(_npz_path(path))
    if int(z["version"][0]) != SNAPSHOT_VERSION:
        raise ValueError("snapshot version %d" % int(z["version"][0]))
    cam = capi.Camera.from_buffer_copy(z["camera"].tobytes())
    kfs = z["keyframes"]
    seeds = None
    if "seeds" in z.files:
        raw = z["seeds"].tobytes()
        seeds = (capi.Seed * (len(raw) // C.sizeof(capi.Seed))).from_buffer_copy(raw)
    return dict(camera=cam, keyframes=kfs, images=[z["image_%d" % k] for k in range(len(kfs))], points=z["points"],
                observations=z["observations"], seeds=seeds, meta=ast.literal_eval(z["meta"].tobytes().decode()))
