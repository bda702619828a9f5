#!/usr/bin/env python3
"""End-to-end (host packets in, host pictures out) frames per second through the plugin seam: the three flavours of
csrc/video_rtjpeg_mi355x.c driven by tests/harness/plugin_harness.c in bench mode.  Prints one JSON object.

    python tools/e2e_bench.py [--width 1920 --height 1088 --packets 64 --repeat 32 --depth 6]
"""
import argparse, importlib, json, os, struct, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(width=1920, height=1088, packets=64, repeat=32, depth=12, quality=255, amp=8, flavours=("", "_nocopy", "_pipe"),
        two_streams=True, warm=None):
    warm = packets if warm is None else warm  # one lap untimed: the decoder allocates its buffers while the first packets come in
    P = importlib.import_module("gmerlin-avdecoder_amd")
    dev = P.MiRtj(0)
    d_fr = dev.synth(width, height, 0, packets, seed=12345, amp=amp)
    d_st, po, pl = dev.encode(width, height, quality, packets, d_fr)
    dev.sync()
    tmp = tempfile.mkdtemp(prefix="mi_rtj_e2e_")
    path = os.path.join(tmp, "p.bin")
    with open(path, "wb") as fh:
        for i in range(packets):
            pkt = dev.d2h(d_st, int(pl[i]), offset=int(po[i]))
            fh.write(struct.pack("<I", pkt.size))
            fh.write(pkt.tobytes())
    dev.free(d_fr); dev.free(d_st); dev.close()
    subprocess.run(["make", "-C", os.path.join(ROOT, "gmerlin-avdecoder_amd", "csrc")], check=True, capture_output=True)
    out = {"workload": f"{packets} RTjpeg {width}x{height} Q={quality} packets x {repeat} laps (the first {warm} pictures untimed), display {width}x{height - 8 if height == 1088 else height}",
           # what the host link gives ONE picture-sized pinned copy at a time (tools/pcie_probe.py, profiles/r02/pcie_probe.json:
           # 38.4 GB/s for 3.1 MB, 54.6 GB/s for 12.4 MB): the copy out is what bounds a session
           "pcie_cap_fps": round((38.4e9 if width * height * 1.5 < 6e6 else 54.6e9) / (width * height * 1.5), 0)}
    for fl in flavours:
        exe = os.path.join(ROOT, "gmerlin-avdecoder_amd", "lib", "plugin_harness" + fl)
        env = dict(os.environ, MI_RTJ_DEPTH=str(depth))
        ih = height - 8 if height == 1088 else height
        r = subprocess.run([exe, path, str(width), str(ih), "/dev/null", f"repeat={repeat}", "bench=1", f"warm={warm}"], capture_output=True, text=True, env=env, timeout=600)
        name = {"": "copy (synchronous, caller's planes; -DMI_RTJ_COPY_MODE)", "_nocopy": "frame-owning (synchronous; -DMI_RTJ_SYNC_NOCOPY)",
                "_pipe": f"frame-owning, {depth} packets in flight (the default build)"}[fl]
        try:
            out[name] = json.loads(r.stdout.strip().splitlines()[-1])
            for ln in r.stderr.splitlines():  # MI_RTJ_PIPE_STATS=1: where the submitting thread's time went
                if ln.startswith('{"pipe_stats"'):
                    out[name]["pipe_stats"] = json.loads(ln)["pipe_stats"]
        except Exception:
            out[name] = {"error": (r.stderr or r.stdout)[-300:]}
        if fl == "_pipe" and two_streams:  # two decoder instances on two threads of one process: the aggregate
            # A session has four HIP streams (kernels, copy in, two for copies out) and the runtime multiplexes all
            # streams of a process onto GPU_MAX_HW_QUEUES hardware queues, four by default: two sessions then share
            # queues, one's copy in waits behind the other's copy out, and the aggregate falls BELOW one session's rate
            # (profiles/r03/e2e_two_streams.txt).  An application with several streams sets GPU_MAX_HW_QUEUES to four per
            # session; both figures are reported.
            for key, extra in (("two_streams_two_threads", {"GPU_MAX_HW_QUEUES": "8"}), ("two_streams_two_threads_default_queues", {})):
                r = subprocess.run([exe, path, str(width), str(ih), "/dev/null", f"repeat={repeat}", "bench=1", "streams=2", f"warm={warm}"],
                                   capture_output=True, text=True, env=dict(env, **extra), timeout=600)
                try:
                    out[key] = json.loads(r.stdout.strip().splitlines()[-1])
                    out[key]["env"] = extra
                except Exception:
                    out[key] = {"error": (r.stderr or r.stdout)[-300:]}
    os.remove(path); os.rmdir(tmp)
    return out


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--width", type=int, default=1920); ap.add_argument("--height", type=int, default=1088)
    ap.add_argument("--packets", type=int, default=64); ap.add_argument("--repeat", type=int, default=32)
    ap.add_argument("--depth", type=int, default=12)
    a = ap.parse_args()
    print(json.dumps(run(a.width, a.height, a.packets, a.repeat, a.depth)))
