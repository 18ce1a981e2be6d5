"""Developer: timeline of one forked tick from a rocprofv3 --kernel-trace CSV (kernel_trace.csv).
    python tools/dev_timeline.py <kernel_trace.csv> [tick index from the end, default 3]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
k = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", ""), r.get("Stream_Id", r.get("Queue_Id", "?"))) for r in rows]
k.sort()
starts = [i for i, r in enumerate(k) if r[2].startswith("k_alive_list")]
which = int(sys.argv[2]) if len(sys.argv) > 2 else 3
a = starts[-which - 1]; b = starts[-which]
t0 = k[a][0]
print(f"tick of {(k[b][0] - t0) / 1e3:.1f} us (alive_list to alive_list)")
for s, e, name, q in k[a:b]:
    print(f"{(s - t0) / 1e3:8.1f} -> {(e - t0) / 1e3:8.1f}  ({(e - s) / 1e3:7.1f})  q{q:>3s}  {name}")
# average tick over the last 20
d = [(k[starts[i + 1]][0] - k[starts[i]][0]) / 1e3 for i in range(len(starts) - 21, len(starts) - 1)]
print("mean of the last 20 ticks: %.1f us" % (sum(d) / len(d)))
