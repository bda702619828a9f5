#!/usr/bin/env python3
"""Checks the compiled k_decode for the one thing the compiler does not know (csrc/rtj_decode_kernels.h): the
registers that the hand-issued loads of the group loop are filling must not be touched between the loads and the
hand-placed `s_waitcnt vmcnt(8)` / `vmcnt(0)` block behind the row stores, no other vector-memory instruction than the
8-byte row stores (scratch_* spill traffic counts on vmcnt too and is refused) and no compiler-placed vmcnt wait may
sit in between.  Every instantiation of k_decode in the file is checked.

    python tools/check_async_loads.py file.s        (hipcc -save-temps device assembly)

Exit status 0 = clean.  Run by gmerlin-avdecoder_amd/build.py on every library it builds (a finding fails the build)
and by tests/test_abi_cpu.py."""
import re
import sys


def regs_of(tok):
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r"v(\d+)", tok)
    return {int(m.group(1))} if m else set()


def all_vregs(line):
    out = set()
    for tok in re.findall(r"v\[\d+:\d+\]|v\d+", line):
        out |= regs_of(tok)
    return out


def kernels_of(path, stem="k_decode"):
    """(label, first line, end line) of every instantiation of the kernel template `stem` in the file"""
    lines = open(path).read().split("\n")
    out = []
    for i, l in enumerate(lines):
        m = re.match(r"^(_ZN5mirtj\d+" + stem + r"(?:I\w+?E)?E\w*):", l)
        if m:
            end = next(k for k in range(i, len(lines)) if lines[k].startswith(".Lfunc_end"))
            out.append((m.group(1), i, end))
    return lines, out


# the two hand-issued load / wait pairs of the decode kernels, told apart by the comments their asm blocks carry:
# decode_wave's group loop (rtj_decode_kernels.h) and the pooling chroma waves' round loop (rtj_decode_chroma.h)
ROLES = {
    "luma": dict(loads="mirtj luma loads", wait="mirtj luma wait", arms=(0, 8), per_block=8),
    "chroma": dict(loads="mirtj chroma pool loads", wait="mirtj chroma pool wait", arms=(0, 8, 16, 24), per_block=8),
}
# which pairs a kernel must hold
KERNEL_ROLES = (("k_decode_split", ("luma", "chroma")), ("k_decode_list", ("luma",)), ("k_decode", ("luma",)))


def check_all(path, stem=None):
    errs = []
    seen = set()
    for kstem, roles in KERNEL_ROLES:
        if stem and stem != kstem:
            continue
        lines, ks = kernels_of(path, kstem)
        ks = [k for k in ks if k[0] not in seen]
        if not ks:
            errs.append(f"{kstem}: not found in {path}")
        for name, a, b in ks:
            seen.add(name)
            for r in roles:
                errs += check(lines[a:b], name, r)
    return errs


def check(body, kernel, role="luma"):
    R = ROLES[role]
    # the hand-issued block: the ASMSTART region that carries the role's comment (exactly one, inside the loop)
    blocks = []
    i = 0
    while i < len(body):
        if "#ASMSTART" in body[i]:
            j = i
            while "#ASMEND" not in body[j]:
                j += 1
            if any(R["loads"] in t for t in body[i:j]):
                blocks.append((i, j))
            i = j
        i += 1
    if len(blocks) != 1:
        return [f"{kernel}: expected one hand-issued {role} load block, found {len(blocks)}"]
    b0, b1 = blocks[0]
    # the loads take their bases from scalar registers: a vector instruction just in front of the block may have
    # written one (v_readlane_b32 reloading a spilled value), and nobody pads that hazard inside an asm block
    first = next(t.strip() for t in body[b0 + 1:b1] if t.strip())
    if first != "s_nop 4":
        return [f"{kernel}: the hand-issued load block must open with 's_nop 4' (scalar base written by a vector "
                f"instruction, read by a vector-memory instruction: five wait states), found '{first}'"]
    pending = set()
    for t in body[b0:b1]:
        t = t.strip()
        if t.startswith("global_load"):
            pending |= regs_of(t.split()[1].rstrip(","))
    errs = []
    # ---- basic blocks between the loads and the hand-placed wait block, with their store counts and successors ----
    wait_at = None
    for k in range(b1 + 1, len(body)):
        if "#ASMSTART" in body[k]:
            j = k
            while "#ASMEND" not in body[j]:
                j += 1
            txt = body[k:j]
            if any(R["wait"] in x for x in txt):
                arms = sorted(int(m) for x in txt for m in re.findall(r"s_waitcnt vmcnt\((\d+)\)", x))
                if tuple(arms) != R["arms"]:
                    errs.append(f"line {k}: the {role} wait block has the arms vmcnt{arms}, expected {R['arms']}")
                wait_at = k
                break
    if wait_at is None:
        return [f"{kernel}: hand-placed wait block not found behind the loads"]
    blocks_ = []  # (label or None, first line, stores, terminators)
    cur = {"label": None, "start": b1 + 1, "stores": 0, "succ": [], "fall": True}
    inasm = False
    for k in range(b1 + 1, wait_at):
        raw = body[k]
        if "#ASMSTART" in raw:
            inasm = True
        if "#ASMEND" in raw:
            inasm = False
        t = raw.split(";")[0].strip()
        if not t or t.startswith("."):
            if not (t.endswith(":") and t.startswith(".L")):
                continue
        if t.endswith(":") and not inasm:
            blocks_.append(cur)
            cur = {"label": t[:-1], "start": k, "stores": 0, "succ": [], "fall": True}
            continue
        op = t.split()[0]
        if op == "s_waitcnt" and "vmcnt" in t:
            errs.append(f"line {k}: compiler-placed '{t}' while hand-issued loads are pending")
            continue
        if op.startswith("global_store"):
            cur["stores"] += 1
            continue
        if op.startswith(("global_load", "buffer_", "flat_", "global_atomic", "scratch_")):
            errs.append(f"line {k}: vector-memory instruction '{t}' between the loads and their wait")
        if op.startswith("s_cbranch") and not inasm:
            cur["succ"].append(t.split()[1])
            blocks_.append(cur)
            cur = {"label": None, "start": k + 1, "stores": 0, "succ": [], "fall": True}
            continue
        if op == "s_branch" and not inasm:
            cur["succ"].append(t.split()[1])
            cur["fall"] = False
            blocks_.append(cur)
            cur = {"label": None, "start": k + 1, "stores": 0, "succ": [], "fall": True}
            continue
        touched = all_vregs(t) & pending
        if touched:
            errs.append(f"line {k}: '{t}' touches pending v{sorted(touched)}")
    blocks_.append(cur)
    # How many stores lie on a path is decided by wave-uniform branches this script does not interpret (the three
    # transform variants; the compiler also sinks a variant's last store into a shared block), so
    # only what can be told from the text is checked: every store is a row store (8 bytes), and no basic
    # block holds more than a variant's eight.  That each executed path issues the 8 stores the counted wait
    # assumes is what the parity tests show at run time: a wait that is one short hands the parser stale registers.
    per_block = [b["stores"] for b in blocks_]
    if max(per_block) > R["per_block"] or sum(per_block) < 8:
        errs.append(f"{kernel}: stores per basic block {sorted(n for n in per_block if n)}")
    for k in range(b1 + 1, wait_at):
        t = body[k].split(";")[0].strip()
        if t.startswith("global_store") and not t.startswith("global_store_dwordx2"):
            errs.append(f"line {k}: '{t}': only 8-byte stores are expected here")
    return [f"{kernel}: {e}" if not e.startswith(kernel) else e for e in errs]


if __name__ == "__main__":
    e = check_all(sys.argv[1])
    for x in e:
        print("ASYNC-LOAD CHECK:", x)
    print("pending-load check:", "clean" if not e else f"{len(e)} problem(s)")
    sys.exit(1 if e else 0)
