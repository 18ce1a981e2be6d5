"""Build the HIP extension (libsmarts_mi355x.so) in-tree for gfx950.

    python -m smarts_amd.build            # build if stale
    python -m smarts_amd.build --force

``-ffp-contract=off`` is part of the contract, not a tuning knob: lane ids, event
flags and dones depend on floating-point comparisons that must round like the
reference's Python floats (see smarts_amd/csrc/smx_device.h).
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB_NAME = "libsmarts_mi355x.so"
LIB_PATH = os.path.join(HERE, LIB_NAME)
SOURCES = ["smx_kernels.hip"]
HEADERS = ["smx_device.h", "smx_roadmap.h", "smx_vehicle.h", os.path.join("..", "..", "include", "smx.h")]
ARCH = "gfx950"
FLAGS = ["-O3", "-ffp-contract=off", "-fPIC", "-shared", "-std=c++17", "-fno-fast-math"]


def hipcc_path() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (looked at $HIPCC, /opt/rocm/bin/hipcc, PATH)")


def is_stale() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(CSRC, f) for f in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build_variant(suffix: str, defines) -> str:
    """Developer variants (e.g. ``_prof`` with -DSMX_DEBUG_TIMING); selected with $SMX_LIBRARY."""
    out = os.path.join(HERE, LIB_NAME.replace(".so", f"{suffix}.so"))
    cmd = [hipcc_path(), f"--offload-arch={ARCH}", *FLAGS, *[f"-D{d}" for d in defines],
           *[os.path.join(CSRC, s) for s in SOURCES], "-o", out]
    proc = subprocess.run(cmd, cwd=CSRC, capture_output=True, text=True)
    if proc.returncode != 0:
        raise RuntimeError("hipcc failed:\n" + " ".join(cmd) + "\n" + proc.stdout + proc.stderr)
    return out


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not is_stale():
        return LIB_PATH
    cmd = [hipcc_path(), f"--offload-arch={ARCH}", *FLAGS, *[os.path.join(CSRC, s) for s in SOURCES], "-o", LIB_PATH]
    if verbose:
        cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
    proc = subprocess.run(cmd, cwd=CSRC, capture_output=True, text=True)
    if proc.returncode != 0:
        raise RuntimeError("hipcc failed:\n" + " ".join(cmd) + "\n" + proc.stdout + proc.stderr)
    if verbose:
        sys.stderr.write(proc.stderr)
    return LIB_PATH


if __name__ == "__main__":
    if "--prof" in sys.argv:
        print(build_variant("_prof", ["SMX_DEBUG_TIMING"]))
    else:
        print(build(force="--force" in sys.argv, verbose=True))
