"""Builds libacmpc_hip.so for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
from __future__ import annotations

import os
import shutil
import subprocess

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC_DIR = os.path.normpath(os.path.join(PKG_DIR, "..", "csrc"))
LIB_DIR = os.path.join(PKG_DIR, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libacmpc_hip.so")
SOURCES = ("acmpc_kernels.hip", "acmpc_kernels_temporal.hip", "acmpc_capi.hip", "acmpc_prologue.hip", "acmpc_pf.hip",
           "acmpc_speed_profile.cpp", "acmpc_host_path.cpp")
HEADERS = ("acmpc_kernels.h", "acmpc_device.h", "acmpc_frames.h", "acmpc_admm.h", "acmpc_prologue.h", "acmpc_lq.h", "acmpc_lq_box.h",
           os.path.join("..", "..", "include", "acmpc.h"))

# -ffp-contract=off: no IMPLICIT fused multiply-add anywhere; the FMAs of mode T's specification are spelt out
HIPCC_FLAGS = ("--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wall", "-Wextra")
# per-source extras: the mode-T rollout is faster without the SLP vectoriser's v_pk_* re-packing (measured, see the source)
# -mfma on the host-only solver: its three sequential sweeps spell out one fma per step (csrc/acmpc_admm.h), which
# should be the instruction, not a libm call
EXTRA_FLAGS = {"acmpc_kernels_temporal.hip": ("-fno-slp-vectorize",), "acmpc_speed_profile.cpp": ("-mfma",)}


def find_hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: set HIPCC or install ROCm")


FLAGS_PATH = os.path.join(LIB_DIR, "build_flags.txt")


def flag_record() -> str:
    """Everything besides the sources that decides what the library is: the flag sets and a one-off experiment's extras."""
    lines = ["common: " + " ".join(HIPCC_FLAGS)]
    lines += ["%s: %s" % (src, " ".join(EXTRA_FLAGS.get(src, ()))) for src in SOURCES]
    lines.append("experiment: " + " ".join(os.environ.get("ACMPC_HIPCC_EXTRA", "").split()))
    return "\n".join(lines) + "\n"


def is_stale() -> bool:
    """Sources or headers newer than the library, a build script newer than it, or a library built with other flags
    (an A/B build with ACMPC_HIPCC_EXTRA left in the tree must not be what the tests and the bench silently run)."""
    if not os.path.exists(LIB_PATH) or not os.path.exists(FLAGS_PATH):
        return True
    built = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(CSRC_DIR, f) for f in SOURCES + HEADERS]   # (acmpc_kernels_temporal.hip includes acmpc_kernels.hip)
    deps.append(os.path.abspath(__file__))
    if any(os.path.getmtime(d) > built for d in deps):
        return True
    with open(FLAGS_PATH) as handle:
        return handle.read() != flag_record()


def build_library(force: bool = False, verbose: bool = False) -> str:
    """Compile the HIP sources into ac-mpc_amd/acmpc_amd/lib/libacmpc_hip.so; returns its path."""
    if not force and not is_stale():
        return LIB_PATH
    os.makedirs(LIB_DIR, exist_ok=True)
    obj_dir = os.path.join(LIB_DIR, "obj")
    os.makedirs(obj_dir, exist_ok=True)
    hipcc = find_hipcc()

    def run(cmd):
        if verbose:
            print(" ".join(cmd))
        proc = subprocess.run(cmd, capture_output=True, text=True)
        if proc.returncode != 0:
            raise RuntimeError("hipcc failed:\n" + proc.stdout + proc.stderr)

    from concurrent.futures import ThreadPoolExecutor
    objects = [os.path.join(obj_dir, os.path.splitext(src)[0] + ".o") for src in SOURCES]
    experiment = tuple(os.environ.get("ACMPC_HIPCC_EXTRA", "").split())   # flags of a one-off A/B build, all sources
    with ThreadPoolExecutor(max_workers=4) as pool:   # one hipcc per source (they carry different flags), then one link
        list(pool.map(lambda so: run([hipcc, *HIPCC_FLAGS, *EXTRA_FLAGS.get(so[0], ()), *experiment, "-c",
                                      os.path.join(CSRC_DIR, so[0]), "-o", so[1]]), zip(SOURCES, objects)))
    run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *objects, "-o", LIB_PATH])
    with open(FLAGS_PATH, "w") as handle:
        handle.write(flag_record())
    return LIB_PATH


if __name__ == "__main__":
    print(build_library(force=True, verbose=True))
