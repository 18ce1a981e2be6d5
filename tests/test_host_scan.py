"""The scan's seeded searches (smarts_amd/csrc/smx_scan.h: team_nearest10_carried, team_road_facts_seeded, the
grid form of team_lane_heading_at_point), host-compiled with one-lane teams under AddressSanitizer + UBSan and
driven over random drives on the three BASELINE maps (tests/native/host_scan.cpp, run_host_scan.py).

A search that starts from last tick's answers must return exactly what the search from scratch returns at the same
pose — ten nearest lanepoints (indices and squared distances), path seeds, nearest lane, distance, on-road and
corner bits — whatever the step was (forward drives of 0-2 m, sideways drift, jumps of up to 30 m).  The same holds
for the one-lane forms the large launch form runs (facts_one_lane, seeds_one_lane)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NATIVE = os.path.join(ROOT, "tests", "native")


def test_seeded_scan_equals_the_search_from_scratch(tmp_path):
    lib = str(tmp_path / "libhost_scan.so")
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fPIC", "-shared", "-ffp-contract=off", "-fno-fast-math",
           "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-I", os.path.join(NATIVE, "shim"),
           "-I", os.path.join(ROOT, "smarts_amd", "csrc"), os.path.join(NATIVE, "host_scan.cpp"), "-o", lib]
    proc = subprocess.run(cmd, capture_output=True, text=True)
    assert proc.returncode == 0, proc.stderr[-2000:]
    asan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(asan):
        pytest.skip("no libasan in this toolchain")
    env = dict(os.environ, LD_PRELOAD=asan, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    proc = subprocess.run([sys.executable, os.path.join(NATIVE, "run_host_scan.py"), lib, "25", "loop", "4lane", "minicity"],
                          capture_output=True, text=True, env=env, timeout=900)
    assert proc.returncode == 0 and "runtime error" not in proc.stderr and "AddressSanitizer" not in proc.stderr, proc.stderr[-3000:]
    res = json.loads(proc.stdout.strip().splitlines()[-1])
    for name in ("loop", "4lane", "minicity"):
        r = res[name]
        assert r["steps"] >= 2000, r
        assert r["facts_differ"] == [] and r["seeds_differ"] == [] and r["heading_forms_differ"] == [], (name, r)
        assert r["guessed"] > r["steps"] // 2, (name, r)  # the seeded path is the one that ran
        # the one-lane forms of the large launch form (two passes over per-lane candidate lists; path seeds without
        # the ten-nearest list): identical where they serve the vehicle, and they serve most of them
        assert r["one_lane_facts_differ"] == [] and r["one_lane_seeds_differ"] == [], (name, r)
        assert r["one_lane_facts_served"] > 0.8 * r["steps"] and r["one_lane_seeds_served"] > 0.5 * r["steps"], (name, r)  # (minicity: a third of the poses lie inside junctions)
