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
