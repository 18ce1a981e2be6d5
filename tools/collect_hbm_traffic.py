#!/usr/bin/env python3
"""Turn two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE — they do not fit one pass on gfx950,
MI355X_MICROARCH.md "rocprofv3 PMC slots") of ``bench.py`` into profiles/<round>_hbm_traffic.json.

    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_fetch -- python3 bench.py ...
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_write -- python3 bench.py ...
    python tools/collect_hbm_traffic.py --fetch gpurun_out/pmc_fetch --write gpurun_out/pmc_write \
        --workload-key c2:loop:1024x8 --out profiles/r01_hbm_traffic.json

Units and corrections (MI355X_MICROARCH.md §HBM): both counters are reported in KiB; WRITE_SIZE is
exact for wide streaming stores; FETCH_SIZE tallies 128-B requests at 64 B for wide coalesced reads,
so the guide's correction doubles it.  The per-step figure divides the totals of this library's
kernels by the number of k_control dispatches (= smx_step calls).
"""
import argparse
import csv
import glob
import json
import os
from collections import defaultdict


def kernel_name(raw):
    """'void k_control<0>(KernelArgs)' -> 'k_control': templated kernels are reported with their signature."""
    name = raw.split("(")[0].strip()
    if name.startswith("void "):
        name = name[5:]
    return name.split("<")[0]


def read_counters(directory, counter):
    per_kernel = defaultdict(float)
    dispatches = defaultdict(int)
    files = glob.glob(os.path.join(directory, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        raise SystemExit(f"no *counter_collection.csv under {directory}")
    for path in files:
        with open(path, newline="") as f:
            for row in csv.DictReader(f):
                if row.get("Counter_Name") != counter:
                    continue
                name = kernel_name(row["Kernel_Name"])
                per_kernel[name] += float(row["Counter_Value"])
                dispatches[name] += 1
    return per_kernel, dispatches


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--fetch", required=True)
    ap.add_argument("--write", required=True)
    ap.add_argument("--workload-key", required=True)
    ap.add_argument("--out", required=True)
    ap.add_argument("--bench-json", default=None,
                    help="stdout of the profiled bench.py command (its JSON line): the alive agents per tick of that run are kept in the record")
    args = ap.parse_args()
    fetch, fd = read_counters(args.fetch, "FETCH_SIZE")
    write, wd = read_counters(args.write, "WRITE_SIZE")
    ours = [k for k in set(fetch) | set(write) if k.startswith("k_")]
    # one k_commit dispatch per smx_step (the reset pass commits inside k_first)
    steps_f = fd.get("k_commit", 0)
    steps_w = wd.get("k_commit", 0)
    if not steps_f or not steps_w:
        raise SystemExit("k_commit not found in the counter files")
    kernels = {}
    tot_r = tot_w = 0.0
    for k in sorted(ours):
        r = fetch.get(k, 0.0) * 1024.0 / steps_f
        w = write.get(k, 0.0) * 1024.0 / steps_w
        kernels[k] = {"fetch_bytes_per_step_raw": r, "fetch_bytes_per_step_x2": 2 * r, "write_bytes_per_step": w,
                      "dispatches_per_step": fd.get(k, 0) / steps_f}
        tot_r += r
        tot_w += w
    rec = {
        "workload_key": args.workload_key,
        "steps_profiled": {"fetch_pass": steps_f, "write_pass": steps_w},
        "read_bytes_per_step_raw": tot_r,
        "read_bytes_per_step": 2 * tot_r,
        "write_bytes_per_step": tot_w,
        "bytes_per_step": 2 * tot_r + tot_w,
        "kernels": kernels,
        "source": f"{os.path.basename(args.out)}: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), KiB -> bytes, "
                  "FETCH_SIZE doubled per MI355X_MICROARCH.md (upper bound: the doubling is calibrated for wide "
                  "coalesced reads only)",
    }
    if args.bench_json:
        # the counters are those of the agents alive in the profiled ticks (an agent that is gone writes no rows)
        line = json.loads(open(args.bench_json).read().strip().splitlines()[-1])
        alive = line["config"]["alive_agents_per_tick"]
        rec["alive_agents_per_tick"] = alive["mean_rank0"]
        rec["agent_slots"] = alive["of"]
        rec["alive_fraction"] = alive["mean_rank0"] / alive["of"]
        rec["bytes_per_alive_agent_step"] = rec["bytes_per_step"] / alive["mean_rank0"]
        rec["algorithmic_bytes_per_step"] = line["roofline"]["algorithmic_bytes"]["total"]
        rec["profiled_command"] = "bench.py steps %d warmup %d (this run: %.4f ms per tick under the counters)" % (
            line["steps"], line["warmup"], line["ms_per_step"])
    with open(args.out, "w") as f:
        json.dump(rec, f, indent=1)
    print(json.dumps({k: rec[k] for k in ("read_bytes_per_step", "write_bytes_per_step", "bytes_per_step")}))


if __name__ == "__main__":
    main()
