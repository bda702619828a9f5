"""Multi-rank logic of the path on CPU: world_size-2 gloo processes exercise the same sharding and
reduction code that bench.py runs over RCCL (one process per GPU, no data-path collective)."""
import os
import socket
import subprocess
import sys
import textwrap

from pkg import P, ROOT


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_partitions_cover_every_frame_once():
    sh = __import__("importlib").import_module("gmerlin-avdecoder_amd.shard")
    for n in (0, 1, 7, 64, 257):
        for world in (1, 2, 3, 8):
            for mode in ("block", "cyclic"):
                got = sorted(sum((sh.frames_for_rank(n, r, world, mode) for r in range(world)), []))
                assert got == list(range(n)), (n, world, mode)
            assert sorted(sum((sh.streams_for_rank(n, r, world) for r in range(world)), [])) == list(range(n))
    # block mode keeps every rank's frames contiguous and balanced
    sizes = [len(sh.frames_for_rank(257, r, 8)) for r in range(8)]
    assert max(sizes) - min(sizes) <= 1


WORKER = textwrap.dedent("""
    import importlib, os, sys
    sys.path.insert(0, {root!r})
    import torch.distributed as dist
    sh = importlib.import_module("gmerlin-avdecoder_amd.shard")
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    mine = sh.frames_for_rank(64, rank, world, "cyclic")
    # pretend each frame is 1920x1088 and rank 1 is slower and saw one mismatch
    rep = sh.reduce_report(sh.Report(len(mine), len(mine) * 1920 * 1088, rank, 1.0 + 0.5 * rank), dist)
    dist.barrier()
    if rank == 0:
        print("RESULT", rep.frames, rep.pixels, rep.mismatches, rep.elapsed)
    dist.destroy_process_group()
""")


def test_two_rank_reduction_over_gloo(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER.format(root=ROOT))
    port = free_port()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=180) for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    line = [l for l in outs[0][0].splitlines() if l.startswith("RESULT")][0].split()
    assert int(line[1]) == 64 and int(line[2]) == 64 * 1920 * 1088 and int(line[3]) == 1
    assert abs(float(line[4]) - 1.5) < 1e-9  # MAX over ranks, not the mean


def test_rank_to_numa_node_mapping(tmp_path):
    """a rank runs on the cores of its GPU's NUMA node.  A made-up sysfs tree: four AMD GPUs on two
    nodes (enumerated by PCI address, whatever the card numbers), one card of another vendor, connector entries."""
    import importlib
    shard = importlib.import_module("gmerlin-avdecoder_amd.shard")
    sysfs = tmp_path / "sys"

    def card(name, pci, vendor, node):
        real = sysfs / "devices" / "pci0000:00" / pci
        real.mkdir(parents=True)
        (real / "vendor").write_text(vendor + "\n")
        if node is not None:
            (real / "numa_node").write_text(f"{node}\n")
        d = sysfs / "class" / "drm" / name
        d.mkdir(parents=True)
        (d / "device").symlink_to(real)

    card("card3", "0000:05:00.0", "0x1002", 0)
    card("card0", "0000:85:00.0", "0x1002", 1)
    card("card1", "0000:25:00.0", "0x1002", 0)
    card("card2", "0000:a5:00.0", "0x1002", 1)
    card("card4", "0000:01:00.0", "0x10de", 0)
    (sysfs / "class" / "drm" / "card0-DP-1").mkdir()
    for node, cpus in ((0, "0-3,16-19"), (1, "4-7,20-23")):
        nd = sysfs / "devices" / "system" / "node" / f"node{node}"
        nd.mkdir(parents=True)
        (nd / "cpulist").write_text(cpus + "\n")
    assert shard.gpu_numa_nodes(str(sysfs)) == [("0000:05:00.0", 0), ("0000:25:00.0", 0), ("0000:85:00.0", 1), ("0000:a5:00.0", 1)]
    assert shard.cpus_for_gpu(0, str(sysfs)) == [0, 1, 2, 3, 16, 17, 18, 19]
    assert shard.cpus_for_gpu(3, str(sysfs)) == [4, 5, 6, 7, 20, 21, 22, 23]
    assert shard.cpus_for_gpu(2, str(sysfs), allowed={5, 6, 99}) == [5, 6]
    assert shard.cpus_for_gpu(2, str(sysfs), allowed={0, 1}) is None   # nothing of that node may be used: hands off
    assert shard.cpus_for_gpu(7, str(sysfs)) is None
    assert shard.parse_cpulist("0-2,5, 9-9") == {0, 1, 2, 5, 9}
    r = shard.bind_rank_to_gpu_node(9, str(sysfs))
    assert r["bound"] is False
