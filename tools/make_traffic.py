#!/usr/bin/env python3
"""profiles/traffic.json from a tools/pmc_summary.py summary: HBM bytes and vector instructions per launch of every
kernel, tagged with the digest of the kernel sources they were measured on (bench.py uses the figures only while that
digest matches what it is running).
FETCH_SIZE / WRITE_SIZE are in KiB.  What FETCH_SIZE counts depends on the access shape (MI355X_MICROARCH.md, HBM
section: "other access widths are uncalibrated: calibrate on a known byte count in your own access pattern"):
  * wide coalesced reads (16 bytes per lane: k_decode's stream bytes and block offsets, k_spec_verify, k_index_emit, the
    copy yardstick) are tallied at HALF their bytes: x 2 (checked on k_index_emit in round 1 — 301 MB + 10.7 % halo =
    333 MB, counter 166 MB — and on tools/ubench/lane_line_fetch.hip's k_stream16 in round 4: 4096 MiB read, 2048 counted);
  * the walkers' shape — one lane per 2048-byte chunk, each lane reading its own 128-byte line with dword loads — is
    NOT halved: lane_line_fetch.hip's k_lane_line<0> reads 4224 MiB (4096 + the 33rd dword of every tile) and the
    counter says 3829 MiB: x 1.10.  Round 3 doubled it and reported the walker's traffic as 2.3 x the packets; with the
    calibrated factor it is about 1.3 x.

    python tools/make_traffic.py pmc_summary.json profiles/traffic.json <tag> <frames per launch>"""
import importlib.util, json, os, sys
src, dst, tag, frames = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4])
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("bench", os.path.join(root, "bench.py"))
bench = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bench)
d = json.load(open(src))
out = {"note": "HBM bytes per launch from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes "
               "(tools/pmc_passes.sh), state %s; FETCH_SIZE x 2 for wide coalesced reads (the gfx950 correction), x 1.10 for the "
               "walkers' lane-per-line dword reads (calibrated: tools/ubench/lane_line_fetch.hip)" % tag,
       "source_digest": bench.kernel_source_digest(), "kernels": {}, "valu_instructions": {}}
for k, v in d.items():
    if "SQ_INSTS_VALU" in v:
        out["valu_instructions"][k] = int(v["SQ_INSTS_VALU"])
    if "FETCH_SIZE" not in v or "WRITE_SIZE" not in v:
        continue
    fetch_factor = 1.10 if "k_spec_walk" in k else 2.0
    r, w = int(v["FETCH_SIZE"] * 1024 * fetch_factor), int(v["WRITE_SIZE"] * 1024)
    out["kernels"][k] = {"hbm_read_bytes": r, "hbm_write_bytes": w, "hbm_bytes": r + w}
# the transform step of a batch launch is three kernels since round 4 (k_decode_split, the classic form — idle while the
# policy keeps the split form — and k_decode_list): bench.py's k_decode slot times all three, so their traffic and
# instructions are summed under that name
parts = [k for k in out["kernels"] if k.startswith("k_decode")]
if parts:
    agg = {f: sum(out["kernels"][k][f] for k in parts) for f in ("hbm_read_bytes", "hbm_write_bytes", "hbm_bytes")}
    agg["of"] = parts
    out["kernels"]["k_decode"] = agg
vparts = [k for k in out["valu_instructions"] if k.startswith("k_decode")]
if vparts:
    out["valu_instructions"]["k_decode"] = sum(out["valu_instructions"][k] for k in vparts)
out["frames_per_launch"] = frames  # bench.py scales the figures to its own batch (traffic is linear in frames)
for k, v in out["kernels"].items():
    out[k] = v["hbm_bytes"]
json.dump(out, open(dst, "w"), indent=1)
print("wrote", dst, "digest", out["source_digest"])
