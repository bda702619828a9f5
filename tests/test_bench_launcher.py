"""bench.py --gpus N starts its own ranks when no launcher did, and refuses to report a run whose world size is not
N (round 1's bench.py silently ran one rank for --gpus 8).  The CPU tests use --selftest-ranks (every rank sends a
dummy shard through the real reduction, gloo, no GPU); the gpu tests run the real workloads with two ranks sharing
the box's one device (MI_RTJ_SHARE_DEVICE=1, reduction over gloo) and the other two bench configurations at small
sizes."""
import json
import os
import subprocess
import sys

import pytest

from pkg import ROOT

BENCH = os.path.join(ROOT, "bench.py")


def run_bench(args, env=None, timeout=600):
    e = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, BENCH] + args, capture_output=True, text=True, env=e, timeout=timeout)


def last_json(r):
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert lines, r.stdout[-2000:] + r.stderr[-3000:]
    return json.loads(lines[-1])


def test_gpus_2_without_a_launcher_starts_two_ranks():
    r = run_bench(["--gpus", "2", "--selftest-ranks"])
    assert r.returncode == 0, r.stderr[-3000:]
    d = last_json(r)
    assert d["n_gpus"] == 2 and d["frames"] == 10 + 11 and abs(d["elapsed"] - 2.0) < 1e-9  # SUM of frames, MAX of elapsed


def test_world_size_that_differs_from_gpus_is_an_error():
    r = run_bench(["--gpus", "2", "--selftest-ranks"], env={"WORLD_SIZE": "1", "RANK": "0"})
    assert r.returncode != 0 and "refusing" in (r.stderr + r.stdout)
    r = run_bench(["--gpus", "1", "--selftest-ranks"], env={"WORLD_SIZE": "2", "RANK": "0", "MASTER_ADDR": "127.0.0.1",
                                                             "MASTER_PORT": "29999"})
    assert r.returncode != 0 and "refusing" in (r.stderr + r.stdout)


SHARE = {"MI_RTJ_DIST_BACKEND": "gloo", "MI_RTJ_SHARE_DEVICE": "1"}


@pytest.mark.gpu
def test_two_ranks_on_one_box_report_two_gpus():
    """the N > 1 line carries its own evidence: every rank compares a sample of its own output with the CPU decoder
    (parity_checked = N x sample, through the path's one reduction), rank 0 times the CPU baseline"""
    r = run_bench(["--gpus", "2", "--frames", "128", "--steps", "3", "--warmup", "1", "--verify-frames", "32",
                   "--cpu-seconds", "1", "--no-stress", "--no-e2e"], env=SHARE)
    assert r.returncode == 0, r.stderr[-3000:]
    d = last_json(r)
    assert d["n_gpus"] == 2 and d["config"]["frames_per_gpu"] == 128 and d["value"] > 0
    assert d["parity_checked"] == 2 * 32 and d["parity_sample_per_rank"] == 32 and d["parity_mismatches"] == 0
    assert d["cpu_baseline"]["value"] > 0 and d["cpu_baseline"]["cores"] == 1
    assert d["cpu_baseline_all_cores"]["value"] > 0
    assert d["roofline"]["frac"] > 0


def test_selftest_reduction_carries_the_frames_compared():
    """(CPU, gloo) the reduction sums what every rank compared"""
    sys.path.insert(0, ROOT)
    import importlib
    shard = importlib.import_module("gmerlin-avdecoder_amd.shard")
    rep = shard.reduce_report(shard.Report(10, 1000, 1, 2.0, 32))
    assert (rep.frames, rep.pixels, rep.mismatches, rep.elapsed, rep.checked) == (10, 1000, 1, 2.0, 32)


@pytest.mark.gpu
def test_one_rank_line_checks_every_sampled_frame_and_carries_the_extras():
    r = run_bench(["--frames", "512", "--steps", "3", "--warmup", "1", "--cpu-seconds", "1", "--verify-frames", "64"])
    assert r.returncode == 0, r.stderr[-3000:]
    d = last_json(r)
    assert d["parity_checked"] == 64 and d["parity_mismatches"] == 0
    assert set(d["by_batch"]) >= {"256", "512"} and d["by_batch"]["256"]["frames_per_s"] > 0
    assert "k_synth" in d["config"]["workload"]
    assert d["cpu_baseline"]["cores"] == 1 and d["cpu_baseline_all_cores"]["cores"] >= 1
    assert d["median_step"]["device_ms"] > 0 and d["roofline"]["frac"] > 0
    assert d["kernels"]["k_decode"]["gbs"] > 0
    for k, v in d["kernels"].items():  # kernels that served an empty to-do list claim no bandwidth
        assert v["gbs"] is None or v["ms"] > 0.05, k
    assert d["end_to_end"].get("fps", 0) > 0 and d["stress_amp64"]["frames_per_s"] > 0


@pytest.mark.gpu
@pytest.mark.parametrize("cfg,extra", [("streams4k", ["--frames", "8", "--steps", "2", "--warmup", "1"]),
                                        ("mixed", ["--frames", "1", "--steps", "2", "--warmup", "1"])])
def test_other_bench_configurations(cfg, extra):
    for gpus in (1, 2):
        r = run_bench(["--gpus", str(gpus), "--config", cfg] + extra, env=SHARE if gpus > 1 else None)
        assert r.returncode == 0, r.stderr[-3000:]
        d = last_json(r)
        assert d["n_gpus"] == gpus and d["parity_mismatches"] == 0 and d["parity_checked"] > 0 and d["value"] > 0
        assert d["roofline"]["frac"] >= 0 and d["roofline"]["kernel"] and d["cpu_baseline"]["value"] > 0
        if cfg == "streams4k":  # the whole first lap of every stream is compared (8 packets here)
            assert d["parity_checked"] == 8 * gpus
        assert cfg.replace("streams4k", "3840x2160").replace("mixed", "mixed") in d["metric"] or cfg == "mixed"
