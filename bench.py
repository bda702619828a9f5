#!/usr/bin/env python3
"""bench.py — RTjpeg 1080p decode throughput on MI355X (BASELINE.json's metric).

    python bench.py --gpus N --steps K --warmup W

A "step" is one pass of the whole hot path (block-offset index + dequant/IDCT/plane scatter) over
one batch of `--frames` distinct synthetic RTjpeg frames that are already resident in HBM
(coded 1920x1088 = display 1080p, YUV420, Q=255, intra only; SURVEY.md §8d cfg 2).  Frames,
streams and outputs never leave the device inside the timed region.

One process per GPU (RANK/LOCAL_RANK/WORLD_SIZE from the environment when launched by
torch.distributed.run).  The path shards by frame with no data-path collective (weak scaling: every
rank decodes its own `--frames` frames); RCCL is used only for the barrier and the final
(frames, max elapsed) reduction.

Rank 0 prints ONE JSON line: the contract fields plus
  "roofline":     dominant kernel of the path, algorithmic bytes / its HIP-event time vs HBM peak
  "cpu_baseline": the reference's lib/RTjpeg.c (oracle/_ref, kind "reference") or the oracle port,
                  one thread, timed on this box on a bounded sample of the same packets.
"""
import argparse
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--frames", type=int, default=4096, help="distinct frames resident per GPU")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1088)
    ap.add_argument("--quality", type=int, default=255)
    ap.add_argument("--amp", type=int, default=8, help="noise amplitude of the synthetic content")
    ap.add_argument("--seed", type=int, default=12345)
    ap.add_argument("--cpu-seconds", type=float, default=10.0, help="budget of the CPU baseline leg")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--cpu-all-cores", action="store_true",
                    help="also time one reference decoder per host core of this process's share (extra field)")
    ap.add_argument("--no-verify", action="store_true")
    return ap.parse_args()


def cpu_baseline(pkts, w, h, budget_s, gpu_planes):
    """Times the reference decoder (or the oracle port) on this box's host cores, one thread, exactly
    as decode_rtjpeg calls it: RTjpeg_decompress into the private frame, then one full-frame copy
    (gavl_video_frame_copy, lib/video_rtjpeg.c:81-82)."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import rtjlib as R  # the checker; used here only as the measured CPU baseline
    fsz = w * h * 3 // 2
    if R.have_reference():
        kind, dec = "reference", R.RefCodec()
        dec.w, dec.h_ = w, h
        padded = [np.concatenate([p, np.zeros(4096, np.uint8)]) for p in pkts]

        def run(i, out):
            dec.L.RTjpeg_decompress(dec.h, R._ptr(padded[i]), R._planes_arg(out, w, h))
    else:
        kind, dec = "port", R.OracleDecoder()

        def run(i, out):
            dec.decode(pkts[i], out)
    priv = np.zeros(fsz, np.uint8)
    user = np.zeros(fsz, np.uint8)
    mismatches = 0
    for i, want in gpu_planes.items():  # untimed: the CPU result is also the parity check of this run
        run(i, priv)
        mismatches += int(not np.array_equal(priv, want))
    done, t0 = 0, time.perf_counter()
    while True:
        for i in range(len(pkts)):
            run(i, priv)
            np.copyto(user, priv)
            done += 1
            if time.perf_counter() - t0 > budget_s:
                break
        if time.perf_counter() - t0 > budget_s or done >= 64 * len(pkts):
            break
    dt = time.perf_counter() - t0
    return {"value": round(done / dt, 2), "unit": "frames/s", "cores": 1, "kind": kind,
            "sample": f"{done} frames of the same {w}x{h} packets, decode + one frame copy, {dt:.1f} s, "
                      f"host has {os.cpu_count()} logical cores",
            "mpixels_per_s": round(done * w * h / dt / 1e6, 1)}, mismatches


def _cpu_worker(args):
    pkts, w, h, budget_s = args
    return cpu_baseline(pkts, w, h, budget_s, {})[0]["value"]


def cpu_baseline_all_cores(pkts, w, h, budget_s):
    """One reference decoder per core this process may use (one stream each, as independent bgav instances
    would run): the honest "whole host share" figure of SURVEY.md section 8d."""
    import multiprocessing as mp
    cores = len(os.sched_getaffinity(0))
    with mp.get_context("spawn").Pool(cores) as pool:
        vals = pool.map(_cpu_worker, [(pkts[: min(len(pkts), 8)], w, h, budget_s)] * cores)
    return {"value": round(sum(vals), 1), "unit": "frames/s", "cores": cores, "kind": "reference"
            if os.path.exists(os.path.join(ROOT, "oracle", "_ref", "librtjpeg_ref.so")) else "port",
            "sample": f"{cores} processes, one decoder each, {budget_s:.0f} s, same packets"}


def main():
    a = parse()
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus and world > 1:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")

    import numpy as np
    import torch
    dist = None
    # MI_RTJ_DIST_BACKEND=gloo + MI_RTJ_SHARE_DEVICE=1 rehearse the multi-rank flow on a one-GPU box
    # (all ranks on device 0, reduction over gloo); the real run is one rank per GPU over RCCL.
    backend = os.environ.get("MI_RTJ_DIST_BACKEND", "nccl")
    gpu = 0 if os.environ.get("MI_RTJ_SHARE_DEVICE") else local
    # MI_RTJ_FORCE_DIST=1: take the process-group path even with one rank (rehearses RCCL on a one-GPU box)
    force_dist = bool(os.environ.get("MI_RTJ_FORCE_DIST"))
    if world > 1 or force_dist:
        import torch.distributed as dist
        torch.cuda.set_device(gpu)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", gpu))  # nccl == RCCL on ROCm
        else:
            dist.init_process_group(backend)
    P = importlib.import_module("gmerlin-avdecoder_amd")
    dev = P.MiRtj(gpu)  # raises if the HIP library or the device is missing: no CPU fallback

    w, h, Q, n = a.width, a.height, a.quality, a.frames
    fsz = w * h * 3 // 2
    # ---- untimed: make this rank's frames and streams on the device ----
    first = rank * n
    d_fr = dev.synth(w, h, first, n, seed=a.seed, amp=a.amp)
    d_st, po, pl = dev.encode(w, h, Q, n, d_fr)
    dev.sync()
    dev.free(d_fr)
    hdr0 = dev.d2h(d_st, 12, offset=int(po[0]))
    hdrs = np.tile(hdr0, (n, 1))
    oo = np.arange(n, dtype=np.uint64) * np.uint64(fsz)
    d_out = dev.alloc(fsz * n)
    plan = dev.plan(hdrs, po, pl, oo)
    info = plan.info()

    def barrier():
        if dist is not None:
            dist.barrier()

    for _ in range(a.warmup):
        plan.decode(d_st, d_out)
    dev.sync()
    plan.profile(True)
    barrier()
    torch.cuda.synchronize()
    dev.sync()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        plan.decode(d_st, d_out)
    dev.sync()
    torch.cuda.synchronize()
    barrier()
    dt = time.perf_counter() - t0
    ktimes, launches = plan.times()
    plan.profile(False)

    # the only collective of the path: SUM(frames, pixels, mismatches), MAX(elapsed) — a few bytes over RCCL
    shard = importlib.import_module("gmerlin-avdecoder_amd.shard")
    rep = shard.reduce_report(shard.Report(n, n * w * h, 0, dt), dist,
                              device=f"cuda:{gpu}" if dist is not None and backend == "nccl" else None,
                              force=force_dist)
    tot_frames, dt = rep.frames, rep.elapsed

    if rank == 0:
        fps = tot_frames * a.steps / dt
        # per-launch figures of every kernel of the path (this rank), from HIP events on the launch stream
        alg_bytes = info["bytes_in"] + info["bytes_out"]  # packet read once + planes written once
        alg = {"k_index_summarize": info["bytes_in"], "k_index_resolve": 0, "k_index_emit": info["bytes_in"],
               "k_decode": alg_bytes, "k_spec_walk": info["bytes_in"], "k_spec_verify": 0}
        kernels = {}
        for name, ms in ktimes.items():
            per = ms / max(launches, 1)
            if per <= 0:
                continue
            kernels[name] = {"ms": round(per, 4), "alg_bytes": alg[name],
                             "gbs": round(alg[name] / (per * 1e-3) / 1e9, 2)}
        dom = max(kernels, key=lambda k: kernels[k]["ms"])
        ach = kernels[dom]["gbs"] or 0.0
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))  # PMC bytes per launch of tj["frames_per_launch"] frames; linear in frames
                traffic = int(tj[dom] * n / tj.get("frames_per_launch", 256)) if dom in tj else None
            except Exception:
                traffic = None
        # the north star's kernel of interest, whatever dominates: IDCT + plane scatter
        dec = kernels.get("k_decode", {"gbs": 0.0, "ms": 0.0})
        out = {
            "metric": "RTjpeg 1080p decode frames/sec", "value": round(fps, 1), "unit": "frames/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(dt / a.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "i32", "data": "synthetic",
            "config": {"workload": f"RTjpeg {w}x{h} YUV420 Q={Q} intra-only, {n} distinct frames/GPU resident in HBM "
                                   f"(BASELINE configs[1]); gradient+noise amp {a.amp}, seed {a.seed}",
                       "frames_per_gpu": n, "avg_packet_bytes": int(info["bytes_in"] // n),
                       "sharding": "frames, no data-path collective"},
            "mpixels_per_s": round(fps * w * h / 1e6, 1),
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(ach / HBM_PEAK_GBS, 5), "traffic": traffic,
                         "alg_bytes_per_launch": kernels[dom]["alg_bytes"], "ms_per_launch": kernels[dom]["ms"],
                         "note": "integer-VALU-issue bound, not HBM bound: DESIGN.md section 5"},
            "roofline_decode": {"bound": "hbm", "kernel": "k_decode", "achieved": dec["gbs"], "peak": HBM_PEAK_GBS,
                                "unit": "GB/s", "frac": round((dec["gbs"] or 0.0) / HBM_PEAK_GBS, 5),
                                "ms_per_launch": dec["ms"], "alg_bytes_per_launch": alg_bytes},
            "kernels": kernels,
            "index_mode": os.environ.get("MI_RTJ_INDEX", "parallel"),  # of the exact index; see speculative_index
            "speculative_index": dict(zip(("packets_proven", "stream_chunks"), plan.spec_stats()),
                                      chunks_repaired=getattr(plan, "repaired", 0),
                                      walker_lead_bytes=plan.spec_lead()[0]),  # what the policy chose for the next launch
            "path_gbs": round(alg_bytes * a.steps / dt / 1e9, 2),
        }
        if world == 1 and not a.no_cpu:
            ns = min(n, 64)
            pkts = [dev.d2h(d_st, int(pl[i]), offset=int(po[i])) for i in range(ns)]
            sample = {} if a.no_verify else {i: dev.d2h(d_out, fsz, offset=i * fsz) for i in (0, ns // 2, ns - 1)}
            cb, mism = cpu_baseline(pkts, w, h, a.cpu_seconds, sample)
            out["cpu_baseline"] = cb
            out["parity_mismatches"] = mism
            out["speedup_vs_cpu_1core"] = round(fps / cb["value"], 1)
            if a.cpu_all_cores:
                out["cpu_baseline_all_cores"] = cpu_baseline_all_cores(pkts, w, h, a.cpu_seconds)
        # what a plain streaming copy kernel sustains on this device (read + write), measured now: the second
        # yardstick of SURVEY.md section 8d next to the nominal peak (last: it overwrites half of the output)
        half = (fsz * n // 2) & ~15
        out["copy_ceiling_gbs"] = round(dev.copy_ceiling(d_out, d_out + half, half), 1)
        print(json.dumps(out), flush=True)

    plan.close()
    dev.free(d_st)
    dev.free(d_out)
    dev.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
