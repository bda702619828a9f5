#!/usr/bin/env python3
"""profiles/traffic.json from a tools/pmc_summary.py summary: HBM bytes and vector instructions per launch of every
kernel, tagged with the digest of the kernel sources they were measured on (bench.py uses the figures only while that
digest matches what it is running).
FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE is doubled (MI355X_MICROARCH.md, HBM section;
checked on k_index_emit, which streams the length table: 301 MB + 10.7 % halo = 333 MB, counter 166 MB).

    python tools/make_traffic.py pmc_summary.json profiles/traffic.json <tag> <frames per launch>"""
import importlib.util, json, os, sys
src, dst, tag, frames = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4])
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("bench", os.path.join(root, "bench.py"))
bench = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bench)
d = json.load(open(src))
out = {"note": "HBM bytes per launch from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes "
               "(tools/pmc_passes.sh), state %s; FETCH_SIZE doubled per the gfx950 correction" % tag,
       "source_digest": bench.kernel_source_digest(), "kernels": {}, "valu_instructions": {}}
for k, v in d.items():
    if "SQ_INSTS_VALU" in v:
        out["valu_instructions"][k] = int(v["SQ_INSTS_VALU"])
    if "FETCH_SIZE" not in v or "WRITE_SIZE" not in v:
        continue
    r, w = int(v["FETCH_SIZE"] * 1024 * 2), int(v["WRITE_SIZE"] * 1024)
    out["kernels"][k] = {"hbm_read_bytes": r, "hbm_write_bytes": w, "hbm_bytes": r + w}
out["frames_per_launch"] = frames  # bench.py scales the figures to its own batch (traffic is linear in frames)
for k, v in out["kernels"].items():
    out[k] = v["hbm_bytes"]
json.dump(out, open(dst, "w"), indent=1)
print("wrote", dst, "digest", out["source_digest"])
