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


# ---------------------------------------------------------------------------------------------------------------
# Host side of a rank: the cores next to its GPU.  A rank that copies packets in and pictures out (configs[3]: one 4K
# stream per GPU, host to host) moves ~55 GB/s through pinned memory; on a two-socket node a rank whose thread and
# pinned pages sit on the other socket pays the socket link for every byte.  bench.py calls bind_rank_to_gpu_node()
# before the first GPU call of a rank (pinned allocations made afterwards land on the node by first touch).
# ---------------------------------------------------------------------------------------------------------------
def _read(path):
    try:
        with open(path) as f:
            return f.read().strip()
    except OSError:
        return None


def parse_cpulist(text):
    """'0-3,8,10-11' -> {0, 1, 2, 3, 8, 10, 11}"""
    out = set()
    for part in (text or "").split(","):
        part = part.strip()
        if not part:
            continue
        if "-" in part:
            a, b = part.split("-", 1)
            out.update(range(int(a), int(b) + 1))
        else:
            out.add(int(part))
    return out


def gpu_numa_nodes(sysfs="/sys"):
    """NUMA node of every AMD GPU of the host, in the order the runtime enumerates them by default (PCI bus address):
    [(pci address, node)], node -1 when the platform does not say."""
    import glob
    import os
    found = {}
    for dev in glob.glob(os.path.join(sysfs, "class", "drm", "card*", "device")):
        if os.path.basename(os.path.dirname(dev)).count("-"):  # card0-DP-1 and the like: connectors, not devices
            continue
        if (_read(os.path.join(dev, "vendor")) or "").lower() != "0x1002":
            continue
        real = os.path.realpath(dev)
        node = _read(os.path.join(dev, "numa_node"))
        found[os.path.basename(real)] = int(node) if node not in (None, "") else -1
    return sorted(found.items())


def cpus_for_gpu(gpu_index, sysfs="/sys", allowed=None):
    """The CPUs a rank that drives GPU `gpu_index` should run on: those of the GPU's NUMA node that the process may
    use; None when that cannot be told (no such GPU in sysfs, no node, nothing of the node allowed) — the caller then
    leaves the affinity alone."""
    import os
    gpus = gpu_numa_nodes(sysfs)
    if gpu_index < 0 or gpu_index >= len(gpus):
        return None
    node = gpus[gpu_index][1]
    if node < 0:
        return None
    cpus = parse_cpulist(_read(os.path.join(sysfs, "devices", "system", "node", f"node{node}", "cpulist")))
    if allowed is not None:
        cpus &= set(allowed)
    return sorted(cpus) or None


def bind_rank_to_gpu_node(gpu_index, sysfs="/sys"):
    """Pin this process to the cores of its GPU's NUMA node.  Returns what it did, for the bench line."""
    import os
    try:
        allowed = os.sched_getaffinity(0)
        cpus = cpus_for_gpu(gpu_index, sysfs, allowed)
        if not cpus:
            return {"bound": False, "reason": "no NUMA node known for this GPU"}
        if set(cpus) == set(allowed):
            return {"bound": False, "reason": "the process may only use that node's cores already", "cpus": len(cpus)}
        os.sched_setaffinity(0, cpus)
        return {"bound": True, "node": gpu_numa_nodes(sysfs)[gpu_index][1], "cpus": len(cpus)}
    except OSError as exc:
        return {"bound": False, "reason": str(exc)}
