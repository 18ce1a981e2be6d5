#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection CSVs per kernel: mean counter value per dispatch.
    python tools/pmc_summary.py gpurun_out/pmc_a gpurun_out/pmc_b ...  > profiles/<name>.txt"""
import csv, glob, os, sys
from collections import defaultdict

def kernel_name(raw):
    name = raw.split("(")[0].strip()
    if name.startswith("void "):
        name = name[5:]
    if name.startswith("k_scan_half<"):  # the two halves are different kernels
        return name.split(",")[0] + ">"
    return name.split("<")[0]


tot = defaultdict(lambda: defaultdict(float)); cnt = defaultdict(lambda: defaultdict(int))
for d in sys.argv[1:]:
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(path, newline="") as f:
            for row in csv.DictReader(f):
                k = kernel_name(row["Kernel_Name"])
                if not k.startswith("k_"):
                    continue
                if int(row["Grid_Size"]) < 4096:   # reset-pass stubs at tiny grids are not interesting
                    pass
                tot[k][row["Counter_Name"]] += float(row["Counter_Value"]); cnt[k][row["Counter_Name"]] += 1
for k in sorted(tot):
    print(k)
    for c in sorted(tot[k]):
        print(f"   {c:28s} {tot[k][c] / cnt[k][c]:16.1f}   (n={cnt[k][c]})")
