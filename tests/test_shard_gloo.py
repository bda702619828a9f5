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
