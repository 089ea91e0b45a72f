#!/usr/bin/env python3
"""Development aid: opcode histogram of every innermost loop (a label that a later branch of the same kernel jumps
back to, with no other such label in between) of one kernel in a hipcc --save-temps .s file.
usage: isa_loops.py file.s <mangled-name-substring> [--json]"""
import collections
import json
import re
import sys


def loops(path, needle):
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and needle in l and ":" in l)
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
    body = lines[start + 1:end + 1]
    where = {}
    for i, l in enumerate(body):
        m = re.match(r"^(\.LBB\w+):", l.strip())
        if m:
            where[m.group(1)] = i
    found = []
    for i, l in enumerate(body):
        t = l.strip().split()
        if len(t) == 2 and t[0].startswith("s_cbranch") and t[1] in where and where[t[1]] < i:
            found.append((where[t[1]], i, t[1]))
    # (a backward branch between the flow blocks of a scalar if-chain - s_setprio by quarter of the horizon - is not a loop
    # of the arithmetic: ranges without a vector instruction do not count)
    def has_valu(f):
        return any(l.strip().startswith("v_") for l in body[f[0]:f[1] + 1])
    found = [f for f in found if has_valu(f)]
    inner = [f for f in found if not any(o is not f and f[0] <= o[0] and o[1] <= f[1] for o in found)]
    out = []
    for lo, hi, label in inner:
        ops = []
        for l in body[lo:hi + 1]:
            t = l.strip()
            if not t or t.startswith(";") or t.startswith(".") or re.match(r"^\.?LBB\w+:", t):
                continue
            ops.append(t.split()[0])
        out.append((label, collections.Counter(ops)))
    return out


if __name__ == "__main__":
    result = loops(sys.argv[1], sys.argv[2])
    if "--json" in sys.argv:
        print(json.dumps({label: dict(h) for label, h in result}, indent=1))
    else:
        for label, h in result:
            valu = sum(c for o, c in h.items() if o.startswith("v_"))
            print("%s: %d instructions, %d VALU, %d LDS, %d SALU" % (
                label, sum(h.values()), valu, sum(c for o, c in h.items() if o.startswith("ds_")),
                sum(c for o, c in h.items() if o.startswith("s_"))))
            print("   " + ", ".join("%s x%d" % kv for kv in h.most_common(60)))
