"""Static resource and instruction-class report of the compiled gfx950 kernels.

    python tools/isa_report.py [--kernel k_decode] [--cflags "..."] [--keep DIR] [--json]

Compiles csrc/mi_rtjpeg.hip with -save-temps (device assembly) and prints, per kernel, what the
code object's metadata says (VGPR / SGPR count, scalar-register spills, LDS bytes, scratch) and a
static census of the instruction stream: vector / scalar / LDS / vector-memory instructions and —
the figure VERDICT r2 asked to track — v_readlane_b32 / v_writelane_b32 (scalar spills are paid in
those).  Numbers are static (program text), not dynamic counts."""
import argparse
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "gmerlin-avdecoder_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def device_asm(cflags, keep=None):
    d = keep or tempfile.mkdtemp(prefix="mirtj_isa_")
    os.makedirs(d, exist_ok=True)
    cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wno-unused-function",
           "-save-temps"] + cflags + ["-o", os.path.join(d, "lib.so"), os.path.join(CSRC, "mi_rtjpeg.hip"),
                                      os.path.join(CSRC, "rtj_tables.cpp")]
    r = subprocess.run(cmd, cwd=d, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(r.stdout + r.stderr)
    for f in os.listdir(d):
        if f.startswith("mi_rtjpeg") and f.endswith("gfx950.s"):
            return os.path.join(d, f)
    raise RuntimeError("no device assembly in " + d)


def demangle(names):
    try:
        r = subprocess.run(["c++filt"] + names, capture_output=True, text=True)
        out = r.stdout.split("\n")
        return {n: (out[i] if i < len(out) and out[i] else n) for i, n in enumerate(names)}
    except OSError:
        return {n: n for n in names}


def parse(path):
    text = open(path).read().split("\n")
    kernels = {}
    cur = None
    for ln in text:
        m = re.match(r"^(_Z\w+):\s", ln)
        if m:
            cur = m.group(1)
            kernels[cur] = {"lines": []}
            continue
        if ln.startswith(".Lfunc_end"):
            cur = None
            continue
        if cur is not None:
            kernels[cur]["lines"].append(ln)
    # metadata
    meta = {}
    name = None
    block = {}
    for ln in text:
        s = ln.strip()
        if s.startswith("- .agpr_count") or s.startswith("- .args"):
            if name:
                meta[name] = block
            name, block = None, {}
        m = re.match(r"^(?:    |  - )\.(\w+):\s+(\S+)", ln)  # kernel-level keys only (argument entries sit deeper)
        if m:
            block[m.group(1)] = m.group(2)
            if m.group(1) == "name":
                name = m.group(2)
    if name:
        meta[name] = block
    rep = {}
    for k, v in kernels.items():
        if k not in meta:
            continue
        c = {"valu": 0, "salu": 0, "lds": 0, "vmem": 0, "smem": 0, "v_readlane": 0, "v_writelane": 0, "waitcnt": 0,
             "s_nop": 0, "branch": 0}
        for ln in v["lines"]:
            s = ln.strip()
            if not s or s.startswith(";") or s.startswith("."):
                continue
            op = s.split()[0]
            if op.endswith(":"):
                continue
            if op.startswith("v_readlane") or op.startswith("v_readfirstlane"):
                c["v_readlane"] += 1
            if op.startswith("v_writelane"):
                c["v_writelane"] += 1
            if op.startswith("v_"):
                c["valu"] += 1
            elif op.startswith("s_waitcnt"):
                c["waitcnt"] += 1
            elif op.startswith("s_nop"):
                c["s_nop"] += 1
            elif op.startswith("s_cbranch") or op.startswith("s_branch"):
                c["branch"] += 1
            elif op.startswith("s_load") or op.startswith("s_buffer_load") or op.startswith("s_store"):
                c["smem"] += 1
            elif op.startswith("s_"):
                c["salu"] += 1
            elif op.startswith("ds_"):
                c["lds"] += 1
            elif op.startswith(("global_", "buffer_", "flat_", "scratch_")):
                c["vmem"] += 1
        m = meta[k]
        rep[k] = {"vgpr": int(m.get("vgpr_count", 0)), "sgpr": int(m.get("sgpr_count", 0)),
                  "sgpr_spill": int(m.get("sgpr_spill_count", 0)), "vgpr_spill": int(m.get("vgpr_spill_count", 0)),
                  "lds_bytes": int(m.get("group_segment_fixed_size", 0)),
                  "scratch_bytes": int(m.get("private_segment_fixed_size", 0)), "static": c}
    return rep


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--kernel", default="")
    ap.add_argument("--cflags", default=os.environ.get("MI_RTJ_CFLAGS", ""))
    ap.add_argument("--keep", default=None)
    ap.add_argument("--json", action="store_true")
    a = ap.parse_args()
    rep = parse(device_asm(a.cflags.split(), a.keep))
    names = demangle(list(rep))
    out = {}
    for k, v in rep.items():
        short = re.sub(r"\(.*", "", names[k]).replace("mirtj::", "").replace("void ", "")
        if a.kernel and a.kernel not in short:
            continue
        out[short] = v
    if a.json:
        print(json.dumps(out, indent=1))
        return
    for k, v in out.items():
        c = v["static"]
        print(f"{k}: vgpr {v['vgpr']} sgpr {v['sgpr']} sgpr_spill {v['sgpr_spill']} vgpr_spill {v['vgpr_spill']} "
              f"lds {v['lds_bytes']} scratch {v['scratch_bytes']} | valu {c['valu']} (readlane {c['v_readlane']} "
              f"writelane {c['v_writelane']}) salu {c['salu']} smem {c['smem']} lds {c['lds']} vmem {c['vmem']} "
              f"waitcnt {c['waitcnt']} s_nop {c['s_nop']} branch {c['branch']}")


if __name__ == "__main__":
    sys.exit(main())
