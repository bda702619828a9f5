#!/usr/bin/env python3
"""bench.py — RTjpeg decode throughput on MI355X (BASELINE.json's metric).

    python bench.py --gpus N --steps K --warmup W [--config frames1080|streams4k|mixed|dv]

With --gpus N > 1 and no WORLD_SIZE in the environment this process starts the N ranks itself (one process per
GPU, `python -m torch.distributed.run`, rendezvous on 127.0.0.1) before anything touches a GPU, passes rank 0's
line through and exits with the ranks' status.  A world size that differs from --gpus is an error, never a
silent one-rank run.

Workloads (`config.workload` in the output names the one that ran):
  frames1080  (default; BASELINE configs[1]) a "step" is one pass of the whole hot path (block-offset index +
              dequant/IDCT/plane scatter) over `--frames` distinct synthetic RTjpeg frames per GPU that are already
              resident in HBM (coded 1920x1088 = display 1080p, YUV420, Q=255, intra only; SURVEY.md §8d cfg 2; content:
              that section's generator, `--content lcg`, the default since round 4).
              Frames, streams and outputs never leave the device inside the timed region.  Weak scaling.
  streams4k   (configs[3]) one 3840x2160 stream WITH unchanged (0xFF) blocks per GPU, decoded in order through a
              pipelined session (mi_rtj_pipe_*, what the frame-owning plugin instance uses): host packets in, host
              pictures out, PCIe both ways inside the timed region.  A step is one pass over the stream's `--frames`
              packets.
  mixed       (configs[4]) 64 intra-only streams of mixed geometry and quality, frames dealt cyclically to the
              ranks, one plan per rank, every frame of every rank compared with the CPU oracle.

  dv          (configs[2]) `--frames` (1024) DV25 525/60 DIF frames of 120,000 bytes resident in HBM -> 720x480 4:1:1
              pictures, one kernel (libmi_dv.so); "parity": "unpinned" — the reference holds no DV pixel decoder, the
              checker is this repository's own statement of the format (bench_dv.py).

One process per GPU.  No data-path collective: RCCL carries the barrier and the final (frames, pixels,
mismatches, max elapsed) reduction only.

Rank 0 prints ONE JSON line: the contract fields plus
  "roofline":      dominant kernel of the path, algorithmic bytes / its HIP-event time vs HBM peak; `traffic` = HBM
                   bytes per launch from the committed PMC profile, only while that profile was taken from the same
                   kernel sources (else null)
  "roofline_valu": the same kernel against the vector-issue roof: instructions per launch (PMC) x measured cost per
                   instruction / measured time
  "cpu_baseline":  the reference's lib/RTjpeg.c (oracle/_ref, kind "reference") or the oracle port, one thread, timed
                   on this box (rank 0's host, whatever N) on a bounded sample of the same packets;
                   "cpu_baseline_all_cores": one decoder per core
  "parity_checked" / "parity_mismatches": frames of this run compared with the CPU decoder: EVERY rank copies a sample
                   of its own output to the host, has its own CPU processes decode the same packets and feeds the counts
                   into the one reduction of the path (parity_checked = N x sample)
  "by_batch":      (N = 1) the same step at 256 / 1024 / 4096 frames per launch next to the default 16384
The process exits non-zero when a compared frame differs.
"""
import argparse
import hashlib
import importlib
import json
import os
import socket
import statistics
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
# what one wave64 vector instruction costs a SIMD while instructions of the slower class are in flight on it, which
# in k_decode is always (tools/ubench/valu_mix*.hip, valu_stagger.hip; profiles/r02/ubench_valu.txt): 1.80 ns
VALU_NS_MIXED = 1.80
VALU_NS_PLAIN = 1.00  # plain 32-bit add / sub / shift / logic (and 16-bit min / max) while no expensive form is in flight
SIMDS = 256 * 4


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", choices=["frames1080", "streams4k", "mixed", "dv"], default="frames1080")
    ap.add_argument("--frames", type=int, default=None, help="frames resident per GPU (frames1080: 16384) / per stream")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1088)
    ap.add_argument("--quality", type=int, default=255)
    ap.add_argument("--amp", type=int, default=8, help="noise amplitude of the synthetic content")
    ap.add_argument("--seed", type=int, default=12345)
    ap.add_argument("--content", choices=["hash", "lcg"], default="lcg",
                    help="noise of the synthetic pictures: lcg (default since round 4) = SURVEY 8d / BASELINE.md section 2's "
                         "linear congruential sequence (k_synth_lcg), hash = k_synth's counter-based hash (rounds 1-3)")
    ap.add_argument("--cpu-seconds", type=float, default=6.0, help="budget of each CPU baseline leg")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU legs (and with them the parity check)")
    ap.add_argument("--no-stress", action="store_true", help="skip the short second measurement on noisy content")
    ap.add_argument("--no-e2e", action="store_true", help="skip the end-to-end (plugin harness) measurement")
    ap.add_argument("--verify-frames", type=int, default=None,
                    help="frames of EACH rank's batch compared with the CPU decoder (default 256 at one GPU, 64 per rank otherwise; at least 32)")
    ap.add_argument("--no-sweep", action="store_true", help="skip the batch-size sweep (by_batch)")
    ap.add_argument("--selftest-ranks", action="store_true",
                    help="no GPU work: every rank reports a dummy shard through the same reduction (launcher test)")
    return ap.parse_args()


# --------------------------------------------------------------------------- rank launcher
def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(a):
    """--gpus N without a launcher around us: start the N ranks here.  Nothing has touched a GPU yet."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run(cmd, env=env)
    raise SystemExit(r.returncode)


# --------------------------------------------------------------------------- CPU side (checker + baseline)
def _cpu_decoder(w, h):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    import rtjlib as R  # the checker; used here only as the measured CPU baseline and the parity check
    if R.have_reference():
        dec = R.RefCodec()
        dec.w, dec.h_ = w, h

        def run(pkt, out):
            dec.L.RTjpeg_decompress(dec.h, R._ptr(pkt), R._planes_arg(out, w, h))
        return "reference", run, dec
    dec = R.OracleDecoder()

    def run(pkt, out):
        dec.decode(pkt, out)
    return "port", run, dec


def _cpu_worker(args):
    """One core: decodes its slice of the sample once (digests for the parity check), then keeps decoding for the
    budget exactly as decode_rtjpeg does it: RTjpeg_decompress into the private frame, then one full-frame copy
    (gavl_video_frame_copy, lib/video_rtjpeg.c:81-82)."""
    import numpy as np
    pkts, w, h, budget_s = args
    kind, run, _keep = _cpu_decoder(w, h)
    fsz = w * h * 3 // 2
    padded = [np.concatenate([p, np.zeros(4096, np.uint8)]) for p in pkts]
    priv, user = np.zeros(fsz, np.uint8), np.zeros(fsz, np.uint8)
    digests = []
    for p in padded:
        run(p, priv)
        digests.append(hashlib.sha256(priv.tobytes()).hexdigest()[:32])
    done, t0 = 0, time.perf_counter()
    while budget_s > 0 and padded:
        for p in padded:
            run(p, priv)
            np.copyto(user, priv)
            done += 1
            if time.perf_counter() - t0 > budget_s:
                break
        if time.perf_counter() - t0 > budget_s:
            break
    dt = time.perf_counter() - t0
    return kind, digests, done, dt


def parity_check(pkts, w, h, gpu_digests, procs):
    """(frames compared, mismatches): this rank's sample decoded by `procs` CPU processes of this rank's own host
    process (one decoder each, no timing), digests against the GPU's."""
    import multiprocessing as mp
    procs = max(1, min(procs, len(pkts)))
    shares = [pkts[i::procs] for i in range(procs)]
    with mp.get_context("spawn").Pool(procs) as pool:
        res = pool.map(_cpu_worker, [(s, w, h, 0.0) for s in shares])
    mism, checked = 0, 0
    for i, (_, digs, _, _) in enumerate(res):
        for j, d in enumerate(digs):
            k = i + j * procs
            checked += 1
            mism += int(d != gpu_digests[k])
    return checked, mism


def cpu_timed_legs(pkts, w, h, budget_s):
    """(cpu_baseline, cpu_baseline_all_cores), rank 0 only, nothing else running on the host: the one-core figure is
    timed on its own, then one decoder per core this process may use."""
    import multiprocessing as mp
    cores = max(1, len(os.sched_getaffinity(0)))
    kind, _, done1, dt1 = _cpu_worker((pkts[: min(len(pkts), 16)], w, h, budget_s))
    one = {"value": round(done1 / dt1, 2), "unit": "frames/s", "cores": 1, "kind": kind,
           "sample": f"{done1} decodes of {min(len(pkts), 16)} of the same {w}x{h} packets, decode + one frame copy, "
                     f"{dt1:.1f} s, host has {os.cpu_count()} logical cores",
           "mpixels_per_s": round(done1 * w * h / dt1 / 1e6, 1)}
    shares = [pkts[i::cores] for i in range(cores)]
    shares = [s_ if s_ else pkts[:1] for s_ in shares]
    with mp.get_context("spawn").Pool(cores) as pool:
        res = pool.map(_cpu_worker, [(s_, w, h, budget_s) for s_ in shares])
    allc = {"value": round(sum(r[2] / r[3] for r in res if r[3] > 0), 1), "unit": "frames/s", "cores": cores,
            "kind": kind, "sample": f"{cores} processes, one decoder each, {budget_s:.0f} s, shares of the same "
                                    f"{len(pkts)} packets"}
    return one, allc


# --------------------------------------------------------------------------- PMC profile of the same sources
def kernel_source_digest():
    """Identifies the kernel sources a PMC profile belongs to (profiles/traffic.json carries the digest it was
    taken with; tools/make_traffic.py writes it)."""
    hsh = hashlib.sha256()
    csrc = os.path.join(ROOT, "gmerlin-avdecoder_amd", "csrc")
    for name in sorted(os.listdir(csrc)):
        if name.endswith((".h", ".hip", ".cpp")) and ("kernels" in name or "idct" in name or
                                                       name in ("rtj_common.h", "rtj_decode_chroma.h")):
            hsh.update(name.encode())
            hsh.update(strip_comments(open(os.path.join(csrc, name), encoding="utf-8").read()).encode())
    # ... and the launch geometry (grids, spans, walkers per launch: what decides how often bytes are fetched), which
    # lives on the host side: plan_launch() of mi_rtjpeg.hip, from its first line to the closing brace in column 0
    host = open(os.path.join(csrc, "mi_rtjpeg.hip"), encoding="utf-8").read()
    a = host.find("int plan_launch(")
    b = host.find("\n}\n", a)
    if a < 0 or b < 0:
        raise SystemExit("bench.py: plan_launch() not found in mi_rtjpeg.hip: the traffic digest cannot be formed")
    hsh.update(b"plan_launch")
    hsh.update(strip_comments(host[a:b]).encode())
    return hsh.hexdigest()[:16]


def strip_comments(text):
    """The code of a source file without // comments, trailing blanks and empty lines: what the digest covers (a
    reworded comment does not make a profile stale)."""
    out = []
    for line in text.split("\n"):
        cut = len(line)
        k = line.find("//")
        while k >= 0:
            if line[:k].count('"') % 2 == 0:  # not inside a string literal
                cut = k
                break
            k = line.find("//", k + 2)
        line = line[:cut].rstrip()
        if line:
            out.append(line)
    return "\n".join(out)


def pmc_profile():
    path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        tj = json.load(open(path))
    except Exception:
        return None
    tj["_current"] = tj.get("source_digest") == kernel_source_digest()
    return tj


# --------------------------------------------------------------------------- workloads
def run_frames(a, dev, rank, n, amp, steps, warmup, barrier, sync_all, timed_unprofiled=False):
    """The resident-batch workload: returns everything rank 0 needs for its line.
    timed_unprofiled (the by_batch sweep): the timed steps run without the per-kernel events — eight event records per
    launch are barriers between kernels that a launch of 256 pictures feels — and a few more steps with them on give the
    kernel times.  The headline run keeps its events inside the timed region, as the contract asks."""
    import numpy as np
    w, h, Q = a.width, a.height, a.quality
    fsz = w * h * 3 // 2
    d_fr = (dev.synth_lcg if a.content == "lcg" else dev.synth)(w, h, rank * n, n, seed=a.seed, amp=amp)
    d_st0 = dev.alloc(dev.encode_bound(w, h, n))  # (the stream's worst-case buffer, 51 GB at the default size: not the encoder's time)
    dev.sync()
    t_enc = time.perf_counter()
    d_st, po, pl = dev.encode(w, h, Q, n, d_fr, d_stream=d_st0)  # the GPU encoder: synchronous, frames and stream resident in HBM
    dev.sync()
    t_enc = time.perf_counter() - t_enc
    dev.free(d_fr)
    hdr0 = dev.d2h(d_st, 12, offset=int(po[0]))
    # MI_RTJ_BENCH_PAD (experiments only): bytes left free behind every picture of the output buffer
    pad = int(os.environ.get("MI_RTJ_BENCH_PAD", "0"))
    oo = np.arange(n, dtype=np.uint64) * np.uint64(fsz + pad)
    d_out = dev.alloc((fsz + pad) * n)
    plan = dev.plan(np.tile(hdr0, (n, 1)), po, pl, oo)
    info = plan.info()
    for _ in range(warmup):
        plan.decode(d_st, d_out)
    dev.sync()
    plan.profile(not timed_unprofiled)
    barrier()
    sync_all()
    import gc  # the interpreter's cyclic collector stays out of the timed region (bench_configs.py: 50 ms pauses)
    gc.collect()
    gc.disable()
    t0 = time.perf_counter()
    for _ in range(steps):
        plan.decode(d_st, d_out)
    sync_all()
    barrier()
    dt = time.perf_counter() - t0
    gc.enable()
    if timed_unprofiled:
        plan.profile(True)
        for _ in range(min(steps, 8)):
            plan.decode(d_st, d_out)
        sync_all()
    ktimes, launches = plan.times()
    step_ms = plan.step_times()
    plan.profile(False)
    return dict(plan=plan, info=info, dt=dt, ktimes=ktimes, launches=launches, step_ms=step_ms, d_st=d_st, d_out=d_out,
                po=po, pl=pl, fsz=fsz, n=n, t_enc=t_enc)


def main():
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        launch_ranks(a)
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
        raise SystemExit(f"bench.py: --gpus {a.gpus} but WORLD_SIZE={world}: refusing to report a run of another size")

    shard = importlib.import_module("gmerlin-avdecoder_amd.shard")
    backend = os.environ.get("MI_RTJ_DIST_BACKEND", "nccl")
    if a.selftest_ranks:  # launcher + reduction only, no GPU: what tests/test_bench_launcher.py runs on a CPU box
        import torch.distributed as dist
        dist.init_process_group("gloo")
        rep = shard.reduce_report(shard.Report(10 + rank, (10 + rank) * 100, 0, 1.0 + rank), dist)
        dist.barrier()
        if rank == 0:
            print(json.dumps({"metric": "selftest", "n_gpus": world, "frames": rep.frames, "elapsed": rep.elapsed}), flush=True)
        dist.destroy_process_group()
        return

    import numpy as np
    import torch
    dist = None
    # MI_RTJ_DIST_BACKEND=gloo + MI_RTJ_SHARE_DEVICE=1 rehearse the multi-rank flow on a one-GPU box
    # (all ranks on device 0, reduction over gloo); the real run is one rank per GPU over RCCL.
    gpu = 0 if os.environ.get("MI_RTJ_SHARE_DEVICE") else local
    # a rank runs next to its GPU: the cores (and, by first touch, the pinned memory) of the GPU's NUMA node — before
    # the first GPU call (matters for the host-to-host workloads on a two-socket node)
    numa = shard.bind_rank_to_gpu_node(gpu) if world > 1 or os.environ.get("MI_RTJ_BIND_NUMA") else {"bound": False, "reason": "one rank"}
    force_dist = bool(os.environ.get("MI_RTJ_FORCE_DIST"))  # the process-group path even with one rank
    if world > 1 or force_dist:
        import torch.distributed as dist
        torch.cuda.set_device(gpu)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", gpu))  # nccl == RCCL on ROCm
        else:
            dist.init_process_group(backend)
    if a.config == "dv":
        def barrier_dv():
            if dist is not None:
                dist.barrier()
        red = f"cuda:{gpu}" if dist is not None and backend == "nccl" else None
        out, mism = importlib.import_module("bench_dv").run(a, rank, world, gpu, dist, red, shard, barrier_dv, force_dist)
        if rank == 0:
            print(json.dumps(out), flush=True)
        if dist is not None:
            dist.destroy_process_group()
        if mism:
            raise SystemExit(3)
        return
    P = importlib.import_module("gmerlin-avdecoder_amd")
    dev = P.MiRtj(gpu)  # raises if the HIP library or the device is missing: no CPU fallback

    def barrier():
        if dist is not None:
            dist.barrier()

    def sync_all():
        dev.sync()
        torch.cuda.synchronize()

    red_dev = f"cuda:{gpu}" if dist is not None and backend == "nccl" else None
    if a.config != "frames1080":
        mod = importlib.import_module("bench_configs")  # next to this file: bench code, not product (it drives the CPU checker)
        out, mism = mod.run(a, dev, P, shard, dist, red_dev, rank, world, barrier, sync_all, force_dist)
        if rank == 0:
            print(json.dumps(out), flush=True)
        dev.close()
        if dist is not None:
            dist.destroy_process_group()
        if mism:
            raise SystemExit(3)
        return

    w, h, Q = a.width, a.height, a.quality
    # 16384 pictures per launch (61 GB of packets and planes, ~110 GB with the index buffers): the launch is long enough
    # that the last, partly filled round of resident waves of each kernel no longer shows
    # (profiles/r02/ab_frames_per_launch.txt: 4096 -> 8192 -> 16384 pictures per launch = 6.64 -> 6.32 -> 6.24 ms per 4096
    # pictures; with the walkers' records interleaved per wave 6.05-6.10 at 16384, ab_walker_record_layout.txt)
    n = a.frames or 16384
    r = run_frames(a, dev, rank, n, a.amp, a.steps, a.warmup, barrier, sync_all)
    plan, info, fsz = r["plan"], r["info"], r["fsz"]

    # parity, on EVERY rank: a sample of this rank's own output goes to the host and is compared with what CPU
    # decoders (processes of this rank, its share of the host's cores) make of the same packets
    pad = int(os.environ.get("MI_RTJ_BENCH_PAD", "0"))
    pkts, checked, my_mism = [], 0, 0
    if not a.no_cpu:
        ns = min(n, max(32, a.verify_frames if a.verify_frames else (256 if world == 1 else 64)))
        pkts = [dev.d2h(r["d_st"], int(r["pl"][i]), offset=int(r["po"][i])) for i in range(ns)]
        gpu_digests = [hashlib.sha256(dev.d2h(r["d_out"], fsz, offset=i * (fsz + pad)).tobytes()).hexdigest()[:32]
                       for i in range(ns)]
        checked, my_mism = parity_check(pkts, w, h, gpu_digests, max(1, len(os.sched_getaffinity(0)) // world))
    # the only collective of the path: SUM(frames, pixels, mismatches, frames compared), MAX(elapsed) — a few bytes over RCCL
    rep = shard.reduce_report(shard.Report(n, n * w * h, my_mism, r["dt"], checked), dist, device=red_dev, force=force_dist)
    tot_frames, dt = rep.frames, rep.elapsed
    mismatches = rep.mismatches
    if rank == 0:
        fps = tot_frames * a.steps / dt
        alg_bytes = info["bytes_in"] + info["bytes_out"]  # packet read once + planes written once
        alg = {"k_index_summarize": info["bytes_in"], "k_index_resolve": 0, "k_index_emit": info["bytes_in"],
               "k_decode": alg_bytes, "k_spec_walk": info["bytes_in"], "k_spec_verify": 0}
        kernels = {}
        for name, ms in r["ktimes"].items():
            per = ms / max(r["launches"], 1)
            if per <= 0:
                continue
            worked = per > 0.05 and alg[name] > 0  # a kernel that returned at once on an empty to-do list moved nothing
            kernels[name] = {"ms": round(per, 4), "alg_bytes": alg[name] if worked else None,
                             "gbs": round(alg[name] / (per * 1e-3) / 1e9, 2) if worked else None}
        dom = max(kernels, key=lambda k: kernels[k]["ms"])
        ach = kernels[dom]["gbs"] or 0.0
        prof = pmc_profile()
        scale = n / prof.get("frames_per_launch", n) if prof else 0.0
        traffic = int(prof[dom] * scale) if prof and prof["_current"] and dom in prof else None
        dec = kernels.get("k_decode", {"gbs": 0.0, "ms": 0.0})
        med = statistics.median(r["step_ms"]) if r["step_ms"] else None
        out = {
            "metric": "RTjpeg 1080p decode frames/sec", "value": round(fps, 1), "unit": "frames/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(dt / a.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "i32", "data": "synthetic",
            "config": {"workload": f"RTjpeg {w}x{h} YUV420 Q={Q} intra-only, {n} distinct frames/GPU resident in HBM "
                                   f"(BASELINE configs[1]); content: gradient + noise amp {a.amp}, seed {a.seed}, " +
                                   ("SURVEY 8d's generator: the noise of ONE linear congruential sequence (s*1664525+1013904223, "
                                    "s>>8) over all frames, device generator k_synth_lcg = tests/rtjlib.py synth_frame_lcg; frames, "
                                    "packets and planes pinned to the reference's own lib/RTjpeg.c (tests/golden/lcg_golden.json)"
                                    if a.content == "lcg" else
                                    "device generator k_synth (= tests/rtjlib.py synth_frame: hash-counter noise, gradient "
                                    "wrapped mod w+h); `--content lcg` runs SURVEY 8d's LCG noise instead (same gradient, "
                                    "packets within 1 % of the same size), digests pinned to the reference encoder on the same frames"),
                       "content": a.content,
                       "frames_per_gpu": n, "avg_packet_bytes": int(info["bytes_in"] // n),
                       "sharding": "frames, no data-path collective"},
            "mpixels_per_s": round(fps * w * h / 1e6, 1),
            "median_step": {"device_ms": round(med, 4) if med else None,
                            "frames_per_s_per_gpu": round(n / (med * 1e-3), 1) if med else None,
                            "note": "median over the timed steps of first-kernel-start to k_decode-end (HIP events, rank 0): the "
                                    "LATENCY of a step.  Batch plans build the index of step k + 1 while step k is being "
                                    "transformed (two block-offset indices per plan), so steps overlap and this is longer than "
                                    "ms_per_step; `value` is the contract's whole-run figure"},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(ach / HBM_PEAK_GBS, 5), "traffic": traffic,
                         "traffic_source": (f"profiles/traffic.json ({prof.get('note', '')})" if traffic is not None else
                                            "none: the committed PMC profile was not taken from these kernel sources"),
                         "alg_bytes_per_launch": kernels[dom]["alg_bytes"], "ms_per_launch": kernels[dom]["ms"],
                         "path_frac": round(alg_bytes * a.steps / dt / 1e9 / HBM_PEAK_GBS, 5),
                         "note": "vector-issue bound, not HBM bound: roofline_valu and DESIGN.md section 5.  frac is the dominant "
                                 "kernel's (k_decode = the transform step of a launch: k_decode_split or the classic form, and "
                                 "k_decode_list, one pair of HIP events around them); path_frac is the whole step's — index and "
                                 "transform — on the same algorithmic bytes"},
            "roofline_decode": {"bound": "hbm", "kernel": "k_decode", "achieved": dec["gbs"], "peak": HBM_PEAK_GBS,
                                "unit": "GB/s", "frac": round((dec["gbs"] or 0.0) / HBM_PEAK_GBS, 5),
                                "ms_per_launch": dec["ms"], "alg_bytes_per_launch": alg_bytes},
            "kernels": kernels,
            "index_mode": os.environ.get("MI_RTJ_INDEX", "parallel"),  # of the exact index; see speculative_index
            "speculative_index": dict(zip(("packets_proven", "stream_chunks"), plan.spec_stats()),
                                      chunks_repaired=getattr(plan, "repaired", 0),
                                      walker_lead_bytes=plan.spec_lead()[0]),  # what the policy chose for the next launch
            "path_gbs": round(alg_bytes * a.steps / dt / 1e9, 2),
            "encoder": {"frames_per_s": round(n / r["t_enc"], 1), "seconds": round(r["t_enc"], 4),
                        "gbs": round((info["bytes_in"] + info["bytes_out"]) / r["t_enc"] / 1e9, 1),
                        "frac_of_hbm_peak": round((info["bytes_in"] + info["bytes_out"]) / r["t_enc"] / 1e9 / HBM_PEAK_GBS, 4),
                        "note": "mi_rtj_encode_frames making this run's packets (intra, byte-exact with the reference encoder: "
                                "tests/test_gpu_parity.py): pictures read once + packets written once over the wall time of the "
                                "call, host side included (one synchronisation at the end); not part of `value`"},
            "host_binding": numa,  # rank 0's: cores of its GPU's NUMA node (gmerlin-avdecoder_amd/shard.py)
        }
        # the vector-issue roof of k_decode: instructions per launch from the PMC profile of these sources
        vi = prof.get("valu_instructions", {}).get("k_decode") if prof and prof["_current"] else None
        if vi and dec["ms"]:
            ni = vi * scale
            out["roofline_valu"] = {
                "bound": "valu-issue", "kernel": "k_decode", "instructions_per_launch": int(ni), "simds": SIMDS,
                "measured_ms": dec["ms"], "ns_per_instruction_per_simd": round(dec["ms"] * 1e6 * SIMDS / ni, 3),
                "ubench_ns_plain": VALU_NS_PLAIN, "ubench_ns_expensive_or_mixed": VALU_NS_MIXED,
                "floor_ms_all_plain": round(ni * VALU_NS_PLAIN / SIMDS * 1e-6, 4),
                "floor_ms_all_mixed": round(ni * VALU_NS_MIXED / SIMDS * 1e-6, 4),
                "frac": round(ni * VALU_NS_PLAIN / SIMDS * 1e-6 / dec["ms"], 4),
                "frac_of_expensive_rate": round(ni * VALU_NS_MIXED / SIMDS * 1e-6 / dec["ms"], 4),
                "note": "wave64 vector instructions (PMC SQ_INSTS_VALU of the same sources) x what one costs a SIMD "
                        "(tools/ubench/valu_*.hip: 1.0 ns for plain add/sub/shift/logic while no wave of the SIMD has an "
                        "expensive form in flight, 1.8 ns for everything else — packed 16-bit, SDWA, mad, perm, compares, "
                        "selects — and for everything once one of those is in flight) / SIMDs.  frac = floor_ms_all_plain / "
                        "measured: the share of the 2.4-cycle issue roof the kernel reaches; frac_of_expensive_rate = "
                        "floor_ms_all_mixed / measured: nearly all of k_decode's instructions are of the expensive class "
                        "(its transform runs two values to a register), so this is the roof it sits under"}
        if not a.no_cpu:
            # the CPU baseline: rank 0's host, for any N (the other ranks wait at the barrier below, the host is idle)
            one, allc = cpu_timed_legs(pkts, w, h, a.cpu_seconds)
            out["cpu_baseline"] = one
            out["cpu_baseline_all_cores"] = allc
            out["parity_checked"] = rep.checked
            out["parity_mismatches"] = rep.mismatches
            out["parity_sample_per_rank"] = checked
            out["speedup_vs_cpu_1core"] = round(fps / one["value"], 1)
            out["speedup_vs_cpu_all_cores"] = round(fps / allc["value"], 1) if allc["value"] else None
        # end to end through the plugin seam (host packets in, host pictures out; SURVEY 8d "End-to-end vs kernel (ii)"):
        # the C wrapper driven by tests/harness/plugin_harness.c in bench mode — never part of `value`
        if world == 1 and not a.no_e2e:
            try:
                e2e = importlib.import_module("tools.e2e_bench")
                r2 = e2e.run(w, h, packets=min(n, 64), repeat=17)  # the first lap is not timed (the session's allocations)
                two = r2.pop("two_streams_two_threads", {})
                two_dflt = r2.pop("two_streams_two_threads_default_queues", {})
                out["end_to_end"] = {"fps": max((v.get("fps", 0) for v in r2.values() if isinstance(v, dict)), default=0),
                                     "two_streams_fps": two.get("fps"),
                                     "two_streams_default_hw_queues_fps": two_dflt.get("fps"),
                                     "pcie_cap_fps": r2["pcie_cap_fps"], "workload": r2.get("workload"), "by_flavour": {k: v for k, v in r2.items() if isinstance(v, dict)},
                                     "note": "fps: ONE stream, one host thread, through csrc/video_rtjpeg_mi355x.c, the best "
                                             "flavour (its default build: frame-owning with packets in flight, pictures leaving on "
                                             "two copy streams); two_streams_fps: two instances on two threads of one process, aggregate, with "
                                             "GPU_MAX_HW_QUEUES=8 (the runtime's default of four hardware queues makes two sessions' "
                                             "eight streams share queues: two_streams_default_hw_queues_fps); "
                                             "the clock starts behind the first lap of the packet list (a session allocates its "
                                             "buffers while its first packets come in: the W untimed steps of this contract); "
                                             "pcie_cap_fps = what the host link gives ONE picture-sized pinned copy at a time "
                                             "(38.4 GB/s for 3.1 MB, tools/pcie_probe.py) / picture bytes — the cap of round 2's "
                                             "sessions, which had one copy out in flight"}
            except Exception as exc:  # the harness is a convenience here, not the measurement
                out["end_to_end"] = {"error": str(exc)[:200]}
        # what a plain streaming copy kernel sustains on this device (read + write), measured now: the second
        # yardstick of SURVEY.md section 8d next to the nominal peak (it overwrites half of the output: after the check)
        half = (fsz * n // 2) & ~15
        out["copy_ceiling_gbs"] = round(dev.copy_ceiling(r["d_out"], r["d_out"] + half, half), 1)

    plan.close()
    dev.free(r["d_st"])
    dev.free(r["d_out"])
    barrier()  # ranks other than 0 wait here while rank 0 times the CPU legs

    # the same step at other launch sizes, fresh frames, never part of `value`
    if world == 1 and rank == 0 and not a.no_sweep:
        sweep = {}
        for nb in (256, 1024, 4096):
            if nb >= n:
                continue
            # enough steps for the steady state to show: plans of these sizes build the index of step k + 1 while step k
            # is transformed, and the first steps of a run have nothing to overlap with (round 3 first timed 3 steps
            # after 2 and reported a 1024-picture launch 13 % below what a run of 30 gives)
            sw_steps = max(8, min(64, 32768 // nb))
            sw = run_frames(a, dev, rank, nb, a.amp, sw_steps, 6, barrier, sync_all, timed_unprofiled=True)
            sweep[str(nb)] = {"frames_per_s": round(nb * sw_steps / sw["dt"], 1), "steps": sw_steps,
                              "steps_overlap": 129 <= nb < 8192 and w * h == 1920 * 1088,  # mi_rtjpeg.hip: plan overlap policy
                              "kernels_ms": {k: round(v / max(sw["launches"], 1), 4) for k, v in sw["ktimes"].items() if v > 0},
                              "index": "speculative" if sw["plan"].spec_stats()[1] else "exact"}
            sw["plan"].close()
            dev.free(sw["d_st"])
            dev.free(sw["d_out"])
        sweep[str(n)] = {"frames_per_s": out["value"], "kernels_ms": {k: v["ms"] for k, v in out["kernels"].items()},
                         "index": "speculative" if out["speculative_index"]["stream_chunks"] else "exact"}
        out["by_batch"] = dict(sweep, note="frames per launch -> whole-step frames/s (`steps` timed steps after 6 warm-up steps; "
                                           "the timed steps of the sweep carry no per-kernel events, kernels_ms come from up to 8 more steps with them; "
                                           "the last entry is the headline run, events inside its timed region).  steps_overlap: plans of that size build the index of "
                                           "step k + 1 while step k is transformed; their kernels share the device, so kernels_ms "
                                           "of those entries are not kernel costs (the headline's are: its kernels run back to back)")

    # SURVEY.md section 8d's stress variant (noise +-64: 2.3 MB packets, nothing for the speculative index to lock on)
    # as a second, short, clearly labelled measurement; never part of `value`
    if world == 1 and rank == 0 and not a.no_stress and a.amp != 64:
        ns_frames = n  # the headline launch size
        # five warm-up launches: the plan's policy needs three to give the speculation up on this content (short lead lost,
        # long lead lost twice: k_spec_policy), after which the walkers return at once for 64 launches — the timed steps
        # are the steady state (round 2 timed one of the three and reported 93.6 K)
        s = run_frames(a, dev, rank, ns_frames, 64, 5, 5, barrier, sync_all)
        out["stress_amp64"] = {"frames_per_s": round(ns_frames * 5 / s["dt"], 1), "frames_per_launch": ns_frames,
                               "avg_packet_bytes": int(s["info"]["bytes_in"] // ns_frames),
                               "kernels_ms": {k: round(v / max(s["launches"], 1), 4) for k, v in s["ktimes"].items() if v > 0},
                               "packets_proven": s["plan"].spec_stats()[0],
                               "speculation_paused_launches_left": s["plan"].spec_lead()[1],
                               "index": "serial walker (k_index_walk_todo, one wave per packet)" if ns_frames >= 4096 else "exact kernels",
                               "decode_form": s["plan"].decode_form()[0],
                               "note": "same code, noise amplitude 64 (SURVEY 8d stress variant), 5 timed steps after 5 warm-up "
                                       "launches: steady state — the speculative index paused by its policy, the packets indexed by the "
                                       "serial walker (launches of 4096 packets and more: its time is in the k_index_summarize slot) or "
                                       "the exact kernels, k_decode in its classic form (decode_form 1); once the host has seen the decode policy "
                                       "in that mode (a pinned word, read without waiting) a plan of this size builds the index of launch "
                                       "k + 1 next to the transform of launch k, as smaller plans always do: kernels_ms are then times of "
                                       "kernels that share the device, not kernel costs (MI_RTJ_OVERLAP=0 runs them back to back: "
                                       "profiles/r04/overlap_at_16384.txt)",
                               "index_overlaps_transform": s["plan"].overlapped()}
        s["plan"].close()
        dev.free(s["d_st"])
        dev.free(s["d_out"])
    if rank == 0:
        print(json.dumps(out), flush=True)
    dev.close()
    if dist is not None:
        dist.destroy_process_group()
    if mismatches:
        raise SystemExit(3)


if __name__ == "__main__":
    main()
