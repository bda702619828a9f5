"""bench.py's workloads for BASELINE.json configs[3] and configs[4] (bench code: like bench.py itself it may drive the
CPU checker under tests/ for its parity checks; the product package never does).

Workloads (SURVEY.md section 8d cfg 4 / cfg 5).

streams4k  one 3840x2160 stream WITH unchanged (0xFF) blocks per rank (stream i -> rank i, shard.streams_for_rank),
           decoded in order through a pipelined session (mi_rtj_pipe_*: the entry points the frame-owning plugin
           instance uses): host packets in, host pictures out, PCIe both ways inside the timed region.
mixed      64 intra-only streams of mixed geometry (320x240, 1920x1088, 3840x2160) and quality (64, 128, 255); the
           frames of all streams are dealt cyclically to the ranks (shard.frames_for_rank(..., "cyclic")), each rank
           decodes its share as one plan, resident in HBM, and EVERY frame of every rank is compared with the CPU
           decoder.

Each returns (the JSON object rank 0 prints, mismatches seen by this rank)."""
import hashlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))


def _checker():
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import rtjlib as R  # the CPU checker (oracle / reference build): parity only, never measured here
    return R


def _digest(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()[:32]


def _cpu_digests(args):
    """worker: decode packets in order with one CPU decoder, return the pictures' digests"""
    pkts, = args
    R = _checker()
    dec = R.OracleDecoder()
    out = []
    pic = np.zeros(0, np.uint8)
    for p in pkts:
        w, h = int(p[6]) | (int(p[7]) << 8), int(p[8]) | (int(p[9]) << 8)
        if pic.size != w * h * 3 // 2:
            pic = np.zeros(w * h * 3 // 2, np.uint8)  # the stream's picture: unchanged (0xFF) blocks keep what it holds
        dec.decode(p, pic)
        out.append(_digest(pic))
    return out


HBM_PEAK_GBS = 8000.0  # as bench.py


def _roofline(ktimes, launches, alg_bytes_per_launch, note):
    """the bench contract's roofline object for the dominant kernel of a profiled run"""
    per = {k: v / max(launches, 1) for k, v in ktimes.items() if v > 0}
    dom = max(per, key=per.get)
    ach = alg_bytes_per_launch[dom] / (per[dom] * 1e-3) / 1e9 if alg_bytes_per_launch.get(dom) else 0.0
    return {"bound": "hbm", "kernel": dom, "achieved": round(ach, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(ach / HBM_PEAK_GBS, 5), "traffic": None, "ms_per_launch": round(per[dom], 4),
            "alg_bytes_per_launch": alg_bytes_per_launch.get(dom), "kernels_ms": {k: round(v, 4) for k, v in per.items()},
            "note": note}


def _cpu_timed(args):
    """worker: one CPU decoder (the reference's RTjpeg.c when built, else the oracle port) over the packets in order,
    decode + one frame copy as decode_rtjpeg does, for about `budget` seconds"""
    pkts, budget = args
    import time as _t
    R = _checker()
    if R.have_reference():
        kind, codec = "reference", R.RefCodec()
    else:
        kind, codec = "port", R.OracleDecoder()
    pics = {}
    done, t0 = 0, _t.perf_counter()
    while True:
        for p in pkts:
            w, h = int(p[6]) | (int(p[7]) << 8), int(p[8]) | (int(p[9]) << 8)
            pic = pics.setdefault((w, h), (np.zeros(w * h * 3 // 2, np.uint8), np.zeros(w * h * 3 // 2, np.uint8)))
            pp = np.concatenate([p, np.zeros(4096, np.uint8)])
            if kind == "reference":
                codec.w, codec.h_ = w, h
                codec.L.RTjpeg_decompress(codec.h, R._ptr(pp), R._planes_arg(pic[0], w, h))
            else:
                codec.decode(pp, pic[0])
            np.copyto(pic[1], pic[0])
            done += 1
            if _t.perf_counter() - t0 > budget:
                break
        if _t.perf_counter() - t0 > budget:
            break
    return kind, done, _t.perf_counter() - t0


def _cpu_baseline(pkts, budget, what):
    import multiprocessing as mp
    with mp.get_context("spawn").Pool(1) as pool:  # a process of its own: this one holds the GPU context
        kind, done, dt = pool.map(_cpu_timed, [(pkts, budget)])[0]
    return {"value": round(done / dt, 2), "unit": "frames/s", "cores": 1, "kind": kind,
            "sample": f"{done} decodes ({what}), decode + one frame copy, {dt:.1f} s"}


def run_streams4k(a, dev, shard, dist, red_dev, rank, world, barrier, sync_all, force_dist):
    w, h, Q = 3840, 2160, a.quality
    nf = a.frames or 48
    (stream_id,) = shard.streams_for_rank(world, rank, world)  # as many streams as ranks: stream i on rank i
    d_fr = dev.synth(w, h, stream_id * 1000, nf, seed=a.seed + stream_id, amp=a.amp)
    d_st, po, pl = dev.encode(w, h, Q, nf, d_fr, key_rate=12, lmask=1, cmask=1)
    dev.sync()
    dev.free(d_fr)
    pkts = [dev.d2h(d_st, int(pl[i]), offset=int(po[i])) for i in range(nf)]
    dev.free(d_st)
    # twelve packets in flight, the wrapper's default (pictures leave four at a time): with four in flight (two pairs)
    # the copy engine idles while the caller refills a pair (2,820 against 4,130 pictures per second on one box,
    # profiles/r03/e2e_steady_state.txt)
    pipe = dev.pipe(depth=int(os.environ.get("MI_RTJ_DEPTH_OVERRIDE", "12")), coded_w=w, coded_h=h)

    def lap(check=None):
        got, nxt = 0, 0
        while got < nf:
            while nxt < nf and pipe.room() > 0:
                pipe.submit(pkts[nxt], nxt)
                nxt += 1
            y, u, v, tag = pipe.next()
            if check is not None and tag < len(check):
                check[tag] = _digest(np.concatenate([y, u, v]))
            got += 1

    # the whole first lap is compared: every key period of the stream (key_rate 12: the encoder's reference picture is
    # reset when the key counter wraps, lib/RTjpeg.c:3504-3514) and the pictures after each wrap
    ncheck = nf
    mine = [None] * ncheck
    lap(mine)  # first lap: also the warm-up (the stream starts from a blank picture)
    for _ in range(max(a.warmup - 1, 0)):
        lap()
    barrier()
    sync_all()
    # the interpreter's cyclic collector is kept out of the timed region: a full collection in a process that has
    # imported torch walks millions of objects and took 50 ms out of single laps of 11.6 (profiles/r03/streams4k_laps.txt)
    import gc
    gc.collect()
    gc.disable()
    t0 = time.perf_counter()
    lap_ms = []
    for _ in range(a.steps):
        t1 = time.perf_counter()
        lap()
        lap_ms.append(round((time.perf_counter() - t1) * 1e3, 3))
    sync_all()
    barrier()
    dt = time.perf_counter() - t0
    gc.enable()
    # kernel times: a lap of its own with profiling on (it costs the submitting thread two event records per kernel)
    pipe.profile(True)
    lap()
    ktimes, launches = pipe.times()
    pipe.profile(False)
    pipe.close()
    want = _cpu_digests((pkts[:ncheck],))
    mism = sum(int(x != y) for x, y in zip(mine, want))
    rep = shard.reduce_report(shard.Report(nf, nf * w * h, mism, dt, ncheck), dist, device=red_dev, force=force_dist)
    fps = rep.frames * a.steps / rep.elapsed
    out = {"metric": "RTjpeg 3840x2160 decode frames/sec, in-order streams, host to host", "value": round(fps, 1),
           "unit": "frames/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
           "ms_per_step": round(rep.elapsed / a.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
           "vs_baseline": None, "dtype": "i32", "data": "synthetic",
           "config": {"workload": f"{world} RTjpeg 3840x2160 YUV420 Q={Q} streams with unchanged (0xFF) blocks "
                                  f"(key_rate 12), one per GPU, {nf} packets each, decoded in order through a pipelined "
                                  f"session, host packets in / host pictures out (BASELINE configs[3])",
                       "frames_per_stream": nf, "avg_packet_bytes": int(sum(p.size for p in pkts) // nf),
                       "sharding": "streams, one per GPU, no data-path collective"},
           "mpixels_per_s": round(fps * w * h / 1e6, 1),
           "pcie_cap_frames_per_s_per_gpu": round(54.6e9 / (w * h * 1.5), 0),  # 12.4 MB pinned copies: tools/pcie_probe.py
           "parity_checked": rep.checked, "parity_mismatches": rep.mismatches,
           "lap_ms": lap_ms}  # rank 0's laps of nf pictures, in order (host-side clock)
    if rank == 0:
        bytes_in = sum(int(p.size) - 12 for p in pkts) / nf
        alg = {"k_decode": bytes_in + w * h * 1.5, "k_index_summarize": bytes_in, "k_index_emit": bytes_in}
        out["roofline"] = _roofline(ktimes, launches, alg, "one packet per launch (in-order stream): the path is bound by "
                                    "the host link, not by this kernel; algorithmic bytes = packet + picture")
        if not a.no_cpu:
            out["cpu_baseline"] = _cpu_baseline(pkts, a.cpu_seconds, f"the stream's {nf} 3840x2160 packets in order")
    return out, mism


GEOMS = [(320, 240), (1920, 1088), (3840, 2160)]
QUALS = [64, 128, 255]


def run_mixed(a, dev, shard, dist, red_dev, rank, world, barrier, sync_all, force_dist):
    nstreams, per = 64, a.frames or 4
    total = nstreams * per
    mine = shard.frames_for_rank(total, rank, world, "cyclic")  # frame k = stream k // per, picture k % per
    # make this rank's packets on the device, one (geometry, quality) group at a time, then bring them together
    pkts = {}
    groups = {}
    for k in mine:
        s = k // per
        groups.setdefault((GEOMS[s % 3], QUALS[(s // 3) % 3]), []).append(k)
    for ((w, h), Q), ks in groups.items():
        for k in ks:  # content depends on (stream, picture) only, whatever the rank
            d_fr = dev.synth(w, h, k, 1, seed=a.seed + k // per, amp=a.amp)
            d_st, po, pl = dev.encode(w, h, Q, 1, d_fr)
            dev.sync()
            pkts[k] = dev.d2h(d_st, int(pl[0]), offset=int(po[0]))
            dev.free(d_fr)
            dev.free(d_st)
    order = sorted(pkts)
    plist = [pkts[k] for k in order]
    d_stream, po, pl, hdrs = dev.upload_packets(plist, align=64)
    sizes = [(int(p[6]) | (int(p[7]) << 8)) * (int(p[8]) | (int(p[9]) << 8)) * 3 // 2 for p in plist]
    oo = np.concatenate([[0], np.cumsum([(s + 255) // 256 * 256 for s in sizes])]).astype(np.uint64)
    d_out = dev.alloc(int(oo[-1]))
    plan = dev.plan(hdrs, po, pl, oo[:-1].copy())
    info = plan.info()
    for _ in range(a.warmup):
        plan.decode(d_stream, d_out)
    dev.sync()
    plan.profile(True)
    barrier()
    sync_all()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        plan.decode(d_stream, d_out)
    sync_all()
    barrier()
    dt = time.perf_counter() - t0
    ktimes, launches = plan.times()
    plan.profile(False)
    # every picture of this rank against the CPU decoder (one decoder per picture: the plan applies the header state
    # machine in plan order, and so does a fresh CPU decoder given the same packet alone, since every packet carries
    # a non-zero quality)
    got = [_digest(dev.d2h(d_out, sizes[i], offset=int(oo[i]))) for i in range(len(plist))]
    import multiprocessing as mp
    cores = max(1, min(len(os.sched_getaffinity(0)), 16))
    with mp.get_context("spawn").Pool(cores) as pool:
        res = pool.map(_cpu_digests, [([p],) for p in plist])
    mism = sum(int(g != r[0]) for g, r in zip(got, res))
    pixels = sum(s * 2 // 3 for s in sizes)
    plan.close()
    dev.free(d_stream)
    dev.free(d_out)
    rep = shard.reduce_report(shard.Report(len(plist), pixels, mism, dt, len(plist)), dist, device=red_dev, force=force_dist)
    fps = rep.frames * a.steps / rep.elapsed
    out = {"metric": "RTjpeg decode frames/sec, mixed batch", "value": round(fps, 1), "unit": "frames/s",
           "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(rep.elapsed / a.steps * 1e3, 4),
           "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "i32", "data": "synthetic",
           "config": {"workload": f"{nstreams} intra-only RTjpeg streams x {per} pictures, geometries 320x240 / 1920x1088 / "
                                  f"3840x2160, Q 64 / 128 / 255, frame-level cyclic scatter over {world} GPU(s), one plan per "
                                  f"GPU resident in HBM (BASELINE configs[4])",
                       "frames_total": rep.frames, "sharding": "frames, cyclic, no data-path collective"},
           "mpixels_per_s": round(rep.pixels * a.steps / rep.elapsed / 1e6, 1),
           "parity_checked": rep.checked, "parity_mismatches": rep.mismatches}
    if rank == 0:
        alg = {"k_decode": info["bytes_in"] + info["bytes_out"], "k_index_summarize": info["bytes_in"],
               "k_index_emit": info["bytes_in"], "k_spec_walk": info["bytes_in"]}
        out["roofline"] = _roofline(ktimes, launches, alg, "rank 0's plan; algorithmic bytes = packets + pictures of the plan")
        if not a.no_cpu:
            out["cpu_baseline"] = _cpu_baseline(plist[:24], a.cpu_seconds, "rank 0's first 24 packets of the mix, in plan order")
    return out, mism


def run(a, dev, P, shard, dist, red_dev, rank, world, barrier, sync_all, force_dist):
    fn = run_streams4k if a.config == "streams4k" else run_mixed
    return fn(a, dev, shard, dist, red_dev, rank, world, barrier, sync_all, force_dist)
