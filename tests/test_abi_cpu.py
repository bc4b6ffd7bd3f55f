"""CPU tests of the drop-in boundary: libhso_gpu.so loads without a GPU and exports every
symbol include/hso_gpu.h declares; host-only entry points behave; the binding refuses to
run without the HIP library (no CPU fallback)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import ROOT
from hso_amd import capi


@pytest.fixture(scope="module")
def lib():
    from hso_amd import build
    build.build()  # hipcc cross-compiles gfx950 without a GPU; no-op when up to date
    return capi.load()


def declared_symbols(header="hso_gpu.h"):
    src = open(os.path.join(ROOT, "include", header)).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(hso_gpu_\w+)\s*\(", src)))


def test_library_exports_every_declared_symbol(lib):
    names = declared_symbols()
    assert len(names) >= 17
    for n in names:
        assert hasattr(lib, n), "libhso_gpu.so does not export %s" % n
    assert sorted(capi.EXPORTED_SYMBOLS) == names, "capi.EXPORTED_SYMBOLS out of sync with the header"
    # the parity / trace read-backs live in their own header: none of them is declared by the boundary's
    dbg = declared_symbols("hso_gpu_debug.h")
    for n in dbg:
        assert hasattr(lib, n), "libhso_gpu.so does not export %s" % n
    assert sorted(capi.DEBUG_SYMBOLS) == dbg and not set(dbg) & set(names)
    assert not [n for n in names if "debug" in n]


def test_abi_version_and_struct_layout(lib):
    assert lib.hso_gpu_abi_version() == capi.ABI_VERSION == 2
    # POD layouts the header promises (checked against the C compiler's view by the sizes the
    # library itself was built with: a mismatch shows up as corrupted results in the GPU tests)
    assert C.sizeof(capi.Camera) == 16 + 4 * 8 + 5 * 8
    assert C.sizeof(capi.SE3) == 56
    assert C.sizeof(capi.RefFeat) == 48 and capi.REF_FEAT_DTYPE.itemsize == 48
    assert C.sizeof(capi.TrackJob) == 8 + 8 + 8 + 8 + 56 + 8
    assert C.sizeof(capi.EvalOut) == 49 * 8 + 7 * 8 + 16 + 16 + 8
    assert C.sizeof(capi.TrackResult) % 8 == 0


def test_pattern_tables_match_oracle(lib, orc):
    for level in range(5):
        pa, hp = C.c_int(), C.c_int()
        offs = np.zeros((40, 2), np.int8)
        rc = lib.hso_gpu_tracker_pattern(4, level, C.byref(pa), C.byref(hp), offs.ctypes.data_as(C.c_void_p))
        orc_rc, opa, ohp, ooffs = orc.pattern(4, level)
        assert (rc, pa.value, hp.value) == (orc_rc, opa, ohp)
        assert np.array_equal(offs[:opa], ooffs)
    assert lib.hso_gpu_tracker_pattern(4, 9, None, None, None) < 0


def test_create_fails_cleanly_without_gpu(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    h = C.c_void_p()
    assert lib.hso_gpu_create(C.byref(h), 0, None) < 0 and not h.value
    assert lib.hso_gpu_last_error(None) == b"null context"
    assert lib.hso_gpu_frame_release(None, 1) < 0


def test_no_cpu_fallback_when_library_missing(monkeypatch):
    monkeypatch.setattr(capi, "_lib", None)
    monkeypatch.setattr(capi, "LIB_PATH", "/nonexistent/libhso_gpu.so")
    with pytest.raises(capi.HsoGpuError, match="no CPU fallback"):
        capi.Context(0)


def test_product_never_imports_the_oracle():
    """Nothing under hso_amd/ or include/ may reference oracle/ (tests, smoke and bench's
    cpu_baseline leg are the only allowed users)."""
    bad = []
    for base, _, files in os.walk(os.path.join(ROOT, "hso_amd")):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp", ".c")):
                txt = open(os.path.join(base, f), errors="ignore").read()
                if re.search(r"hso_or_|oracle_py|from oracle|import oracle|libhso_oracle", txt):
                    bad.append(os.path.join(base, f))
    assert not bad, bad


def test_host_driver_library_exports_its_c_interface(lib):
    """libhso_host.so (the FrameHandlerMono::addImage pipeline in C++) loads without a GPU, exports every symbol
    include/hso_vo.h declares, and refuses to create a driver without a device."""
    from hso_amd import vo
    src = open(os.path.join(ROOT, "include", "hso_vo.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = sorted(set(re.findall(r"\b(hso_vo_\w+)\s*\(", src)))
    hl = vo.load()
    for n in names:
        assert hasattr(hl, n), n
    assert sorted(vo.EXPORTED_SYMBOLS) == names
    import torch
    if not torch.cuda.is_available():
        h = C.c_void_p()
        cam = capi.make_camera(capi.CAM_PINHOLE, 640, 480, 480, 480, 320, 240)
        assert hl.hso_vo_create(C.byref(h), C.byref(cam), 200, 0) < 0 and not h.value
        assert hl.hso_vo_last_error(None) == b"null handle"


def test_trace_reader_round_trip(tmp_path):
    """The driver's call trace format (hso_amd/host/hso_trace.h) as vo.read_trace parses it."""
    import struct
    from hso_amd import vo
    p = tmp_path / "t.bin"
    rec = struct.pack("<II", 0x52545348, 4) + b"call" + struct.pack("<I", 2)
    rec += struct.pack("<I", 1) + b"x" + struct.pack("<Q", 8) + struct.pack("<d", 2.5)
    rec += struct.pack("<I", 3) + b"tab" + struct.pack("<Q", 3) + b"abc"
    p.write_bytes(rec * 2)
    out = vo.read_trace(str(p))
    assert [n for n, _ in out] == ["call", "call"] and vo.scalar(out[1][1], "x") == 2.5 and out[0][1]["tab"] == b"abc"


def test_graft_entry_build_passes_on_this_tree():
    """the driver's "does it build" check (__graft_entry__.build(): compile what is stale, load every library, check the ABI generation
    and every exported symbol) — run here so that a bumped ABI version or a renamed symbol cannot leave it behind"""
    import __graft_entry__ as entry
    entry.build()
