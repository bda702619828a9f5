#!/bin/bash
# trace the copies of a session (copy engine activity) to see why they are slower than back-to-back copies
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3b; mkdir -p $O
python - <<'PY'
import sys, os, struct, importlib
sys.path.insert(0, '.')
P = importlib.import_module("gmerlin-avdecoder_amd")
dev = P.MiRtj(0)
for (w, h, n, name) in ((3840, 2160, 24, "p4k.bin"), (1920, 1088, 64, "p1080.bin")):
    d_fr = dev.synth(w, h, 0, n, seed=12345, amp=8)
    d_st, po, pl = dev.encode(w, h, 255, n, d_fr)
    dev.sync()
    with open("/tmp/" + name, "wb") as fh:
        for i in range(n):
            pkt = dev.d2h(d_st, int(pl[i]), offset=int(po[i]))
            fh.write(struct.pack("<I", pkt.size)); fh.write(pkt.tobytes())
    dev.free(d_fr); dev.free(d_st)
dev.close()
PY
H=gmerlin-avdecoder_amd/lib/plugin_harness_pipe
for cfg in "4k_singles_skip6 3840 2160 /tmp/p4k.bin 6 1" "4k_pairs_full 3840 2160 /tmp/p4k.bin 0 2" "1080_pairs_full 1920 1080 /tmp/p1080.bin 0 2" "1080_singles_skip6 1920 1080 /tmp/p1080.bin 6 1"; do
  set -- $cfg
  export MI_RTJ_EXP_SKIP=$5 MI_RTJ_OUT_GROUP=$6 MI_RTJ_DEPTH=6
  timeout -k 10 120 rocprofv3 --memory-copy-trace --kernel-trace --output-format csv -d $O/tr_$1 -- $H $4 $2 $3 /dev/null repeat=6 bench=1 > $O/tr_$1.log 2>&1; echo "$1 rc=$?"
  tail -1 $O/tr_$1.log
  python - $O/tr_$1 "$1" <<'PY' | tee -a $O/copy_trace.txt
import csv, glob, os, sys
root, label = sys.argv[1], sys.argv[2]
paths = glob.glob(os.path.join(root, "**", "*memory_copy_trace.csv"), recursive=True)
rows = list(csv.DictReader(open(paths[0])))
print(label, "columns:", list(rows[0].keys()))
d2h = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r) for r in rows if "DEVICE_TO_HOST" in r.get("Direction", "").upper() or "D2H" in r.get("Direction", "").upper()]
h2d = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r) for r in rows if "HOST_TO_DEVICE" in r.get("Direction", "").upper()]
for name, v in (("d2h", d2h), ("h2d", h2d)):
    v.sort()
    v = v[len(v)//4:]  # steady state
    if len(v) < 4: continue
    dur = sorted(e - s for s, e, _ in v)
    gap = sorted(v[i+1][0] - v[i][1] for i in range(len(v)-1))
    span = v[-1][1] - v[0][0]
    print(f"  {name}: {len(v)} copies, duration median {dur[len(dur)//2]/1e3:.1f} us (min {dur[0]/1e3:.1f}, max {dur[-1]/1e3:.1f}); gap to the next median {gap[len(gap)//2]/1e3:.1f} us (min {gap[0]/1e3:.1f}, max {gap[-1]/1e3:.1f}); busy {sum(dur)/span*100:.0f} % of {span/1e3:.0f} us")
PY
  rm -rf $O/tr_$1
done
