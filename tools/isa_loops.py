#!/usr/bin/env python3
"""Instruction mix of the loops of one kernel in a hipcc -save-temps .s file (gfx950).
usage: tools/isa_loops.py file.s kernel_substring [min_len]
Lists every loop (backward branch) with its length and a histogram by class: plain VALU, packed VALU (v_pk_*), DPP moves,
LDS, VMEM, SALU, waitcnt -- the numbers behind 'instructions per tile / per symbol' in DESIGN.md."""
import re
import sys
from collections import Counter


def classify(m, ops):
    if m.startswith("v_pk_"):
        return "valu_pk"
    if "dpp" in ops or m.endswith("_dpp"):
        return "valu_dpp"
    if m.startswith("v_cmp") or m.startswith("v_cmpx"):
        return "valu_cmp"
    if m.startswith("v_readlane") or m.startswith("v_writelane") or m.startswith("v_readfirstlane"):
        return "valu_lane"
    if m.startswith("v_"):
        return "valu"
    if m.startswith("ds_"):
        return "lds"
    if m.startswith("global_") or m.startswith("buffer_") or m.startswith("flat_") or m.startswith("scratch_"):
        return "vmem"
    if m.startswith("s_waitcnt") or m.startswith("s_nop"):
        return "wait"
    if m.startswith("s_cbranch") or m.startswith("s_branch"):
        return "branch"
    if m.startswith("s_"):
        return "salu"
    return "other"


def main():
    path, key = sys.argv[1], sys.argv[2]
    min_len = int(sys.argv[3]) if len(sys.argv) > 3 else 20
    lines = open(path).read().splitlines()
    start = None
    for i, l in enumerate(lines):
        if re.match(r"^_Z\w*:", l) and key in l:
            start = i
            break
    if start is None:
        sys.exit("kernel not found")
    end = start + 1
    while end < len(lines) and not lines[end].startswith("\t.section") and not lines[end].startswith(".Lfunc_end"):
        end += 1
    body = lines[start:end]
    labels = {}
    insts = []  # (index in body, mnemonic, operands)
    for i, l in enumerate(body):
        m = re.match(r"^(\.LBB\w+):", l)
        if m:
            labels[m.group(1)] = len(insts)
            continue
        m = re.match(r"^\t([a-z_0-9]+)\s*(.*)$", l)
        if m and not m.group(1).startswith("."):
            insts.append((m.group(1), m.group(2)))
    print("%s: %d instructions" % (body[0].rstrip(":"), len(insts)))
    tot = Counter(classify(m, o) for m, o in insts)
    print("  whole kernel:", dict(tot))
    loops = []
    for i, (m, o) in enumerate(insts):
        if m.startswith("s_cbranch") or m == "s_branch":
            tgt = o.strip().split()[-1]
            if tgt in labels and labels[tgt] <= i:
                loops.append((labels[tgt], i))
    for a, b in sorted(loops, key=lambda t: -(t[1] - t[0])):
        if b - a + 1 < min_len:
            continue
        c = Counter(classify(m, o) for m, o in insts[a:b + 1])
        valu = sum(v for k, v in c.items() if k.startswith("valu"))
        print("  loop inst %5d..%5d  len %4d  VALU %4d  %s" % (a, b, b - a + 1, valu, dict(c)))


if __name__ == "__main__":
    main()
