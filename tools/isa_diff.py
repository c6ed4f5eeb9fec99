#!/usr/bin/env python3
"""Which kernels' device code differs between two `hipcc -S --cuda-device-only` listings of kernels.hip (comments and basic-block
numbering normalised)?  tools/isa_diff.py old.s new.s  -- used to show that a commit which adds or changes one kernel leaves the
others, and therefore the measurement set taken before it, untouched."""
import re, sys
def kernels(path):
    out = {}; cur = None; buf = []
    for line in open(path):
        m = re.match(r'^(_ZN4aisk\S+):', line)
        if m: cur = m.group(1); buf = []; continue
        if cur is not None:
            if line.strip().startswith('.end_amdhsa_kernel') or line.startswith('.Lfunc_end'):
                body = [re.sub(r'\.LBB\d+_', '.LBB_', l.split(';')[0].rstrip()) for l in buf]
                out[cur] = "\n".join(l for l in body if l); cur = None
            else: buf.append(line)
    return out
a, b = kernels(sys.argv[1]), kernels(sys.argv[2])
print("%d / %d kernels" % (len(a), len(b)))
print("changed:", [k for k in a if k in b and a[k] != b[k]])
print("new:", [k for k in b if k not in a])
print("gone:", [k for k in a if k not in b])
