#!/usr/bin/env python3
"""Development tool: the VALU opcode mix of the instruction-bound kernels' hot loops, from the compiler's own assembly.

Compiles csrc/acmpc_kernels_temporal.hip and csrc/acmpc_kernels.hip with the library's flags + --save-temps in a scratch
directory, takes the innermost loops of the named kernels (tools/isa_loops.py) and writes
profiles/<tag>_isa_mix.json: per entry the static opcode histogram of ONE trip of the loop (one wave-step), the wave's
candidates per lane, and the sha256 of the sources it was compiled from - bench.py prices the mix with the issue costs of
profiles/<tag>_valu_probe.json (tools/valu_probe.hip) and flags a mix whose sources are not the loaded library's.

usage: python3 tools/isa_mix.py r04        (CPU only: hipcc cross-compiles)"""
import hashlib
import json
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ac-mpc_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from acmpc_amd import _build  # noqa: E402
from isa_loops import loops  # noqa: E402

# entry -> (source, kernel name substring, how to recognise the loop: (opcode, count) pairs that must match)
ENTRIES = {
    "window_2_5": ("acmpc_kernels_temporal.hip", "rollout_kernelILi1ELi1ELi2ELi256ELi1ELi8E", {"ds_read_b128": 16}),
    "window_1_2": ("acmpc_kernels_temporal.hip", "rollout_kernelILi1ELi1ELi2ELi256ELi1ELi8E", {"ds_read_b128": 10}),
    "exhaustive": ("acmpc_kernels_temporal.hip", "rollout_kernelILi1ELi1ELi2ELi256ELi1ELi8E", {"ds_read_b128": 20}),
    # the fused sample + rollout round: its step loop comes in several unrolled pieces of one mix; the largest stands for it
    # (the SQ count per candidate-step also holds the Philox draws and the staging, priced with the same mix)
    "fused_round": ("acmpc_kernels.hip", "rollout_sampled_kernelILi0E", None),
}
CANDIDATES_PER_LANE = {"fused_round": 1}


def source_hash():
    h = hashlib.sha256()
    for name in sorted(_build.SOURCES + _build.HEADERS):
        with open(os.path.join(_build.CSRC_DIR, name), "rb") as handle:
            h.update(handle.read())
    h.update(_build.flag_record().encode())
    return h.hexdigest()


def assembly(source, scratch):
    flags = list(_build.HIPCC_FLAGS) + list(_build.EXTRA_FLAGS.get(source, ()))
    subprocess.run([_build.find_hipcc(), *flags, "-w", "-c", os.path.join(_build.CSRC_DIR, source), "-o",
                    os.path.join(scratch, "unit.o"), "--save-temps"], check=True, cwd=scratch)
    stem = os.path.splitext(source)[0]
    return os.path.join(scratch, stem + "-hip-amdgcn-amd-amdhsa-gfx950.s")


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
    out = {"tool": "tools/isa_mix.py " + tag, "source_sha256": source_hash(), "build_flags": _build.flag_record(),
           "unit": "instructions of ONE trip of the step loop = one wave-step (64 lanes x candidates_per_lane candidates)",
           "entries": {}}
    cache = {}
    with tempfile.TemporaryDirectory() as scratch:
        for entry, (source, kernel, signature) in ENTRIES.items():
            if source not in cache:
                cache[source] = assembly(source, scratch)
            every = loops(cache[source], kernel)
            if signature is None:
                found = [max(every, key=lambda lh: sum(lh[1].values()))]
            else:
                found = [(label, h) for label, h in every if all(h.get(op, 0) == count for op, count in signature.items())]
            if len(found) != 1:
                raise SystemExit("%s: %d loops match %r" % (entry, len(found), signature))
            label, hist = found[0]
            out["entries"][entry] = {"kernel": kernel, "loop": label, "candidates_per_lane": CANDIDATES_PER_LANE.get(entry, 2),
                                     "valu": {op: c for op, c in sorted(hist.items()) if op.startswith("v_")},
                                     "lds": {op: c for op, c in sorted(hist.items()) if op.startswith("ds_")},
                                     "salu": sum(c for op, c in hist.items() if op.startswith("s_")),
                                     "vmem": {op: c for op, c in sorted(hist.items())
                                              if op.startswith("global_") or op.startswith("buffer_")}}
    path = os.path.join(ROOT, "profiles", tag + "_isa_mix.json")
    with open(path, "w") as handle:
        json.dump(out, handle, indent=1)
    for entry, e in out["entries"].items():
        print(entry, e["loop"], "VALU", sum(e["valu"].values()), "LDS", sum(e["lds"].values()), "SALU", e["salu"])


if __name__ == "__main__":
    main()
