"""Sharding of the decode path over the GPUs of a node.

The path has no exchange step: packets of intra-only streams are independent, and a stream with
0xFF "unchanged" blocks is independent of every other stream.  So ranks never trade data; the only
collective is the final reduction of (frames, pixels, mismatches, frames compared) by SUM and elapsed time by MAX,
a few bytes over RCCL (backend "nccl" on ROCm) or gloo in the CPU tests."""
from dataclasses import dataclass


def frames_for_rank(n_frames, rank, world, mode="block"):
    """Frame numbers this rank decodes.  "block": contiguous chunks (keeps packet staging sequential);
    "cyclic": frame k -> rank k mod world (cfg 5's frame-level scatter).  Every frame is owned once."""
    if mode == "cyclic":
        return list(range(rank, n_frames, world))
    per, extra = divmod(n_frames, world)
    lo = rank * per + min(rank, extra)
    return list(range(lo, lo + per + (1 if rank < extra else 0)))


def streams_for_rank(n_streams, rank, world):
    """Stateful streams (skip blocks) shard whole: stream i -> rank i mod world, decoded in order there."""
    return list(range(rank, n_streams, world))


@dataclass
class Report:
    frames: int
    pixels: int
    mismatches: int
    elapsed: float
    checked: int = 0  # frames of this rank compared with the CPU decoder


def reduce_report(local, dist=None, device=None, force=False):
    """SUM of frames/pixels/mismatches and MAX of elapsed over all ranks; identity without dist
    (force: run the collectives even in a group of one, to rehearse the backend)."""
    if dist is None or not dist.is_initialized() or (dist.get_world_size() == 1 and not force):
        return Report(local.frames, local.pixels, local.mismatches, local.elapsed, local.checked)
    import torch
    dev = device if device is not None else "cpu"
    s = torch.tensor([local.frames, local.pixels, local.mismatches, local.checked], dtype=torch.int64, device=dev)
    m = torch.tensor([local.elapsed], dtype=torch.float64, device=dev)
    dist.all_reduce(s, op=dist.ReduceOp.SUM)
    dist.all_reduce(m, op=dist.ReduceOp.MAX)
    return Report(int(s[0].item()), int(s[1].item()), int(s[2].item()), float(m[0].item()), int(s[3].item()))
