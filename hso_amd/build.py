"""Build libhso_gpu.so (hand-written HIP for gfx950) in-tree with hipcc.

`python -m hso_amd.build` or hso_amd.build.build().  hipcc cross-compiles
without a GPU.  -ffp-contract=off: per-term arithmetic must round exactly like
the expressions written in the kernels ("numerics").
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(CSRC, "libhso_gpu.so")
SOURCES = ["hso_ctx.hip", "hso_frame.hip", "hso_tracker.hip", "hso_tracker_coop.hip", "hso_align.hip", "hso_pose.hip", "hso_ba.hip",
           "hso_seed.hip", "hso_activate.hip", "hso_fast.hip", "hso_edgelet.hip", "hso_select.hip", "hso_klt.hip", "hso_octree.cpp"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
         "-fno-fast-math", "-Wall", "-Wno-unused-function"]
# Per-file additions.  The tracker megakernel is compiled without LLVM's SLP vectoriser: the packed fp32 / packed 16-bit
# operations it forms there need register pairs, and forming them cost 55-61 spilled VGPRs at the kernel's 256-register
# budget (3-7 without; k_track 13.9 -> 13.3 ms on 4096 EuRoC pairs).
PER_FILE = {"hso_tracker.hip": ["-fno-slp-vectorize"], "hso_tracker_coop.hip": ["-fno-slp-vectorize"],
            # no spills either way, but the scalar forms are faster here too: k_align_t 596 -> 559 us, k_seed_observe 931 -> 852 us
            # (128 sequences, profiles/collect_stages.sh); k_pose, k_sobel unchanged, the remaining files not measured
            "hso_align.hip": ["-fno-slp-vectorize"], "hso_seed.hip": ["-fno-slp-vectorize"]}
OBJ = os.path.join(HERE, "..", "build", "obj")


def _headers():
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    deps.append(os.path.join(HERE, "..", "include", "hso_gpu.h"))
    return deps


def _host_sources():
    host = os.path.join(HERE, "host")
    return [os.path.join(host, f) for f in sorted(os.listdir(host)) if f.endswith((".cpp", ".h"))] + [os.path.join(HERE, "..", "include", "hso_vo.h")]


def needs_build():
    if not os.path.exists(OUT) or not os.path.exists(os.path.join(HERE, "host", "libhso_host.so")) \
            or not os.path.exists(os.path.join(HERE, "host", "libhso_gather.so")):
        return True
    t = os.path.getmtime(OUT)
    deps = [os.path.join(CSRC, f) for f in SOURCES] + _headers()
    th = os.path.getmtime(os.path.join(HERE, "host", "libhso_host.so"))
    return any(os.path.getmtime(d) > t for d in deps) or any(os.path.getmtime(d) > th for d in _host_sources() + _headers())


def build(force=False, verbose=False, extra=()):
    if not force and not needs_build():
        return OUT
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    extra = list(extra) + os.environ.get("HSO_EXTRA_FLAGS", "").split()
    os.makedirs(OBJ, exist_ok=True)
    newest_header = max(os.path.getmtime(d) for d in _headers())
    objs, cmds = [], []
    for src in SOURCES:
        obj = os.path.join(OBJ, os.path.splitext(src)[0] + ".o")
        objs.append(obj)
        path = os.path.join(CSRC, src)
        # a translation unit is recompiled when it or any header is newer than its object (or flags were added for this run)
        if not force and not extra and os.path.exists(obj) and os.path.getmtime(obj) > max(os.path.getmtime(path), newest_header):
            continue
        cmds.append([hipcc] + FLAGS + PER_FILE.get(src, []) + extra + ["-c", "-o", obj, path])
    # one compiler process per translation unit, all at once (the tracker alone takes about as long as the rest together)
    procs = []
    for cmd in cmds:
        if verbose:
            print(" ".join(cmd))
        procs.append(subprocess.Popen(cmd))
    failed = [cmd for cmd, pr in zip(cmds, procs) if pr.wait() != 0]
    if failed:
        raise subprocess.CalledProcessError(1, failed[0])
    if cmds or not os.path.exists(OUT):
        link = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] + objs
        if verbose:
            print(" ".join(link))
        subprocess.check_call(link)
    build_host(verbose)
    return OUT


def build_host(verbose=False):
    """libhso_host.so: the sequence engine (hso_engine*.cpp; C interface include/hso_vo.h) and the per-call mirror of the
    reference's class surface (hso_host.cpp) + the mirror's test driver: plain g++ against the C-ABI only (the device library is
    found at run time through $ORIGIN)."""
    host = os.path.join(HERE, "host")
    rpath = ["-L" + CSRC, "-lhso_gpu", "-Wl,-rpath,$ORIGIN/../csrc",
             "-Wl,-rpath," + os.path.join(os.environ.get("ROCM_PATH", "/opt/rocm"), "lib")]
    engine = [os.path.join(host, f) for f in ("hso_math.cpp", "hso_init.cpp", "hso_engine.cpp", "hso_engine_step.cpp", "hso_engine_kf.cpp",
                                              "hso_engine_init.cpp", "hso_engine_c.cpp")]
    srcs = engine + [os.path.join(host, "hso_host.cpp")]
    lib = os.path.join(host, "libhso_host.so")
    exe = os.path.join(host, "hso_host_test")
    for cmd in (["g++", "-O2", "-std=c++17", "-Wall", "-fPIC", "-shared", "-pthread"] + srcs + rpath + ["-o", lib],
                ["g++", "-O2", "-std=c++17", "-Wall", "-pthread"] + srcs + [os.path.join(host, "hso_host_test.cpp")] + rpath + ["-o", exe]):
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    # libhso_gather.so: the native result gather (ncclAllGather), on its own so that the engine library has no RCCL dependency
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    cmd = [os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "-O2", "-std=c++17", "-Wall", "-fPIC", "-shared",
           os.path.join(host, "hso_gather.cpp"), "-L" + os.path.join(rocm, "lib"), "-lrccl", "-Wl,-rpath," + os.path.join(rocm, "lib"),
           "-o", os.path.join(host, "libhso_gather.so")]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return exe


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True,
          extra=["-Rpass-analysis=kernel-resource-usage"] if "--usage" in sys.argv else [])
