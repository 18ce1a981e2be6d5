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
INCLUDE = os.path.join(HERE, "..", "include")
ARCH = "gfx950"
FLAGS = ["-O3", "-ffp-contract=off", "-fPIC", "-shared", "-std=c++17", "-fno-fast-math"]


def hipcc_path() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (looked at $HIPCC, /opt/rocm/bin/hipcc, PATH)")


def dependencies():
    """Everything the library is compiled from: every file under csrc/ and include/ (a header added
    later is picked up without touching this list) and this recipe itself."""
    import glob

    deps = sorted(glob.glob(os.path.join(CSRC, "*")) + glob.glob(os.path.join(INCLUDE, "*.h")))
    return [d for d in deps if os.path.isfile(d)] + [os.path.abspath(__file__)]


def is_stale(lib_path: str = LIB_PATH) -> bool:
    if not os.path.exists(lib_path):
        return True
    t = os.path.getmtime(lib_path)
    return any(os.path.getmtime(d) > t for d in dependencies())


def _compile(out: str, extra) -> str:
    """hipcc into a temporary file beside `out`, then an atomic rename: a process that loads the library
    while another one rebuilds it never sees a half-written file."""
    tmp = f"{out}.tmp.{os.getpid()}"
    cmd = [hipcc_path(), f"--offload-arch={ARCH}", *FLAGS, *extra, *[os.path.join(CSRC, s) for s in SOURCES], "-o", tmp]
    proc = subprocess.run(cmd, cwd=CSRC, capture_output=True, text=True)
    if proc.returncode != 0:
        if os.path.exists(tmp):
            os.remove(tmp)
        raise RuntimeError("hipcc failed:\n" + " ".join(cmd) + "\n" + proc.stdout + proc.stderr)
    os.replace(tmp, out)
    return proc.stderr


def build_variant(suffix: str, defines) -> str:
    """Developer variants (e.g. ``_prof`` with -DSMX_DEBUG_TIMING); selected with $SMX_LIBRARY."""
    out = os.path.join(HERE, LIB_NAME.replace(".so", f"{suffix}.so"))
    _compile(out, [f"-D{d}" for d in defines])
    return out


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not is_stale():
        return LIB_PATH
    log = _compile(LIB_PATH, ["-Rpass-analysis=kernel-resource-usage"] if verbose else [])
    if verbose:
        sys.stderr.write(log)
    return LIB_PATH


if __name__ == "__main__":
    if "--prof" in sys.argv:
        print(build_variant("_prof", ["SMX_DEBUG_TIMING"]))
    else:
        print(build(force="--force" in sys.argv, verbose=True))
