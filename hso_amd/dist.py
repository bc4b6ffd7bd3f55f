"""Multi-GPU plumbing: independent sequences shard one-per-rank with no data-path
collective; the only exchange is a gather of fixed-size per-frame result records
(BASELINE.json north_star: "RCCL over xGMI used only to gather results").

Backend "nccl" is RCCL on ROCm; the same code runs on "gloo" for the CPU tests."""
import numpy as np


def shard_sequences(n_sequences, rank, world):
    """Contiguous block partition of sequence ids over ranks (remainder to the low ranks)."""
    base, rem = divmod(n_sequences, world)
    lo = rank * base + min(rank, rem)
    return list(range(lo, lo + base + (1 if rank < rem else 0)))


def pack_records(results):
    """[n, 8] float64: quaternion (x,y,z,w), translation, exposure ratio — one row per frame."""
    rec = np.zeros((len(results), 8))
    for i, r in enumerate(results):
        rec[i, :4], rec[i, 4:7], rec[i, 7] = r.T_cur_ref.q[:], r.T_cur_ref.t[:], r.exposure_rat
    return rec


def pack_trajectories(trajs, n_frames):
    """[S, n_frames, 8] float64 for S sequences: per frame quaternion (x,y,z,w), translation, time stamp; NaN rows where a
    sequence is shorter.  trajs[s] = list of (timestamp, (q, t))."""
    out = np.full((len(trajs), n_frames, 8), np.nan)
    for s, tr in enumerate(trajs):
        for k, (ts, (q, t)) in enumerate(tr[:n_frames]):
            out[s, k, :4], out[s, k, 4:7], out[s, k, 7] = q, t, ts
    return out


def gather_records(rec, device=None):
    """all_gather of equally-shaped per-rank record blocks -> [world, n, 8] on every rank."""
    import torch
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return rec[None]
    mine = torch.from_numpy(np.ascontiguousarray(rec))
    if device is not None:
        mine = mine.to(device)
    world = dist.get_world_size()
    # concatenated-along-dim-0 output form: accepted by both RCCL (nccl) and gloo
    out = torch.empty((world * mine.shape[0],) + tuple(mine.shape[1:]), dtype=mine.dtype, device=mine.device)
    dist.all_gather_into_tensor(out, mine)
    return out.cpu().numpy().reshape((world,) + tuple(mine.shape))


def max_over_ranks(value, device=None):
    import torch
    import torch.distributed as dist
    t = torch.tensor([float(value)], dtype=torch.float64, device=device or "cpu")
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


# ---- the native gather (include/hso_vo.h: hso_gather_*, hso_amd/host/libhso_gather.so): ncclAllGather straight from C,
# for harnesses without torch; bench.py keeps torch.distributed for its gathers and can cross-check them with this one
# (`--native-gather 1`).
GATHER_SYMBOLS = ["hso_gather_unique_id", "hso_gather_create", "hso_gather_destroy", "hso_gather_size", "hso_gather_rank",
                  "hso_gather_records", "hso_gather_last_error"]
GATHER_ID_BYTES = 128
_gather_lib = None


def load_gather():
    """dlopen libhso_gather.so (RCCL is resolved through its rpath).  Raises if the library was not built."""
    global _gather_lib
    if _gather_lib is None:
        import ctypes as C
        import os
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "host", "libhso_gather.so")
        if not os.path.exists(path):
            raise RuntimeError("libhso_gather.so is missing: run python -m hso_amd.build")
        lib = C.CDLL(path)
        lib.hso_gather_last_error.restype = C.c_char_p
        lib.hso_gather_unique_id.argtypes = [C.c_void_p]
        lib.hso_gather_create.argtypes = [C.POINTER(C.c_void_p), C.c_void_p, C.c_int, C.c_int, C.c_int]
        lib.hso_gather_destroy.argtypes = [C.c_void_p]
        lib.hso_gather_size.argtypes = [C.c_void_p]
        lib.hso_gather_rank.argtypes = [C.c_void_p]
        lib.hso_gather_records.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        _gather_lib = lib
    return _gather_lib


class NativeGather:
    """One rank's handle on the RCCL communicator.  `uid` = the 128 bytes of rank 0's NativeGather.unique_id(), carried to the
    other ranks by the caller (bench.py: a torch.distributed broadcast; a C++ harness: whatever launched it)."""

    def __init__(self, uid, rank, world, device=0):
        import ctypes as C
        self.lib = load_gather()
        if len(uid) != GATHER_ID_BYTES:
            raise ValueError("communicator id must be %d bytes" % GATHER_ID_BYTES)
        self.h = C.c_void_p()
        buf = (C.c_uint8 * GATHER_ID_BYTES).from_buffer_copy(bytes(uid))
        rc = self.lib.hso_gather_create(C.byref(self.h), buf, rank, world, device)
        if rc != 0:
            raise RuntimeError("hso_gather_create: %d %s" % (rc, self.lib.hso_gather_last_error().decode()))
        self.world, self.rank = world, rank

    @staticmethod
    def unique_id():
        import ctypes as C
        lib = load_gather()
        buf = (C.c_uint8 * GATHER_ID_BYTES)()
        rc = lib.hso_gather_unique_id(buf)
        if rc != 0:
            raise RuntimeError("hso_gather_unique_id: %d %s" % (rc, lib.hso_gather_last_error().decode()))
        return bytes(buf)

    def gather(self, rec):
        """rec [..., 8] float64 (same shape on every rank) -> [world, ...same..., 8]."""
        rec = np.ascontiguousarray(rec, dtype=np.float64)
        if rec.shape[-1] != 8:
            raise ValueError("records are rows of 8 doubles")
        n_rows = rec.size // 8
        out = np.empty((self.world,) + rec.shape)
        rc = self.lib.hso_gather_records(self.h, rec.ctypes.data, n_rows, out.ctypes.data)
        if rc != 0:
            raise RuntimeError("hso_gather_records: %d %s" % (rc, self.lib.hso_gather_last_error().decode()))
        return out

    def close(self):
        if self.h:
            self.lib.hso_gather_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
