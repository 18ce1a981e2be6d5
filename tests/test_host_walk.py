"""The device's waypoint-path code, host-compiled under AddressSanitizer + UndefinedBehaviorSanitizer and run
against the reference-generated fixtures (tests/native/host_walk.cpp; smarts_amd/csrc/smx_roadmap.h is the code
under test, unchanged: only ``__device__`` and friends are defined away).

Two things come out of it:
* the walk / interpolation source itself — not only the oracle's restatement of the reference — reproduces the
  reference's waypoint paths (integers exact, float64 to 1e-9) on all 3 x (300/300/200) golden poses but the enumerated tie-sensitive
  ones, and its nearest-lane / on-road answers on every pose, with the sanitizers silent;
* the round-1 "stale next0" form (``-DSMX_WALK_CARRIED_NEXT0``: KnotWalk reading ``next0`` from the lanepoint
  record it carries by value) is just as clean and gives the same bits on the host: no undefined behaviour in
  the source, which leaves device code generation (profiles/r02_next0_isa_diff.txt).
"""
import json
import os
import subprocess
import sys

import pytest

import tie_sensitive

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NATIVE = os.path.join(ROOT, "tests", "native")


def _build(tmp_path, name, defines=()):
    out = str(tmp_path / f"libhost_walk_{name}.so")
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fPIC", "-shared", "-ffp-contract=off", "-fno-fast-math",
           "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-I", os.path.join(NATIVE, "shim"),
           "-I", os.path.join(ROOT, "smarts_amd", "csrc"), *[f"-D{d}" for d in defines],
           os.path.join(NATIVE, "host_walk.cpp"), "-o", out]
    proc = subprocess.run(cmd, capture_output=True, text=True)
    assert proc.returncode == 0, proc.stderr[-2000:]
    return out


def _run(lib, maps):
    asan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(asan):
        pytest.skip("no libasan in this toolchain")
    env = dict(os.environ, LD_PRELOAD=asan, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    proc = subprocess.run([sys.executable, os.path.join(NATIVE, "run_host_walk.py"), lib, *maps], capture_output=True,
                          text=True, env=env, timeout=600)
    assert proc.returncode == 0 and "runtime error" not in proc.stderr and "AddressSanitizer" not in proc.stderr, proc.stderr[-3000:]
    return json.loads(proc.stdout.strip().splitlines()[-1])


@pytest.mark.parametrize("form", ["shipped", "carried_next0"])
def test_device_walk_source_under_sanitizers_matches_the_reference(form, tmp_path):
    lib = _build(tmp_path, form, ("SMX_WALK_CARRIED_NEXT0",) if form == "carried_next0" else ())
    res = _run(lib, ["loop", "4lane", "minicity"])
    for name in ("loop", "4lane", "minicity"):
        for lookahead in (16, 32):
            r = res[f"waypoints_{name}_{lookahead}"]
            assert r["differing"] == tie_sensitive.WAYPOINTS[(name, "empty_route", lookahead)], (name, lookahead, r)
        assert res[f"nearest_{name}"]["differing"] == [], name
        # fixed routes (missions_<map>.npz): seeds along the route + the per-lane route table
        r = res[f"routed_{name}"]
        assert r["differing"] == tie_sensitive.ROUTE_WAYPOINTS[(name, 32)] and r["poses"] > 100, (name, r)
