"""Per-kernel average duration and the idle gap before each kernel, over the last `n` dispatches of a rocprofv3
kernel trace CSV.  Usage: trace_gaps.py <kernel_trace.csv> [n]"""
import csv
import json
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
rows = rows[-n:]
agg, prev_end = {}, None
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].split("(")[0].replace("void ", "")[:60]
    a = agg.setdefault(name, [0, 0.0, 0.0])
    a[0] += 1
    a[1] += (e - s) / 1e3
    if prev_end is not None:
        a[2] += max(0.0, (s - prev_end) / 1e3)
    prev_end = e
span = (int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])) / 1e3
# concurrency: time during which >= 2 kernels are in flight (sweep over start / end events)
events = sorted([(int(r["Start_Timestamp"]), 1) for r in rows] + [(int(r["End_Timestamp"]), -1) for r in rows])
live, last, overlapped = 0, None, 0
for ts, d in events:
    if live >= 2 and last is not None:
        overlapped += ts - last
    live += d
    last = ts
out = {"dispatches": len(rows), "span_us": span, "us_with_two_or_more_kernels_in_flight": overlapped / 1e3,
       "busy_us": sum(a[1] for a in agg.values()), "gap_us": sum(a[2] for a in agg.values()),
       "kernels": {k: {"calls": a[0], "avg_us": a[1] / a[0], "avg_gap_before_us": a[2] / a[0]}
                   for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1])}}
print(json.dumps(out, indent=1))
