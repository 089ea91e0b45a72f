#!/usr/bin/env python3
"""Development aid: per-basic-block opcode histogram of one kernel in a hipcc --save-temps .s file.
usage: isa_blocks.py file.s <mangled-name-substring> [min_instructions]"""
import collections
import re
import sys

path, needle = sys.argv[1], sys.argv[2]
least = int(sys.argv[3]) if len(sys.argv) > 3 else 40
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and needle in l and ":" in l)
end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
blocks, name, cur = [], "entry", []
for l in lines[start + 1:end + 1]:
    t = l.strip()
    if not t or t.startswith(";") or t.startswith("."):
        if t.startswith(".LBB") and t.endswith(":"):
            blocks.append((name, cur)); name, cur = t[:-1], []
        continue
    if re.match(r"^\.?LBB\w+:", t):
        blocks.append((name, cur)); name, cur = t.split(":")[0], []
        continue
    cur.append(t.split()[0])
blocks.append((name, cur))
for name, ops in blocks:
    if len(ops) < least:
        continue
    h = collections.Counter(ops)
    valu = sum(c for o, c in h.items() if o.startswith("v_"))
    print("%s: %d instructions, %d VALU, %d LDS, %d SALU, %d VMEM" % (
        name, len(ops), valu, sum(c for o, c in h.items() if o.startswith("ds_")),
        sum(c for o, c in h.items() if o.startswith("s_")),
        sum(c for o, c in h.items() if o.startswith("global_") or o.startswith("buffer_"))))
    print("   " + ", ".join("%s x%d" % kv for kv in h.most_common(40)))
