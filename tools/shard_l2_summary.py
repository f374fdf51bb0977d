"""Per-kernel L2 picture of a profiled command (tools/shard_l2.sh): average duration, L2 hits / misses / hit rate, L2
requests, fabric-side read and write bytes (FETCH_SIZE KiB x2 — the gfx950 half-count of wide streaming reads,
MI355X_MICROARCH.md "HBM" —, WRITE_SIZE KiB x1).  Prints one JSON document.  Usage: shard_l2_summary.py <dir> <tag>"""
import csv
import glob
import json
import os
import sys


def find(root, pattern):
    hits = sorted(glob.glob(os.path.join(root, "**", pattern), recursive=True))
    return hits[0] if hits else None


def short(name):
    return name.split("(")[0].replace("void ", "")[:110]


def per_kernel(root, sub, counters):
    cc = find(os.path.join(root, sub), "*counter_collection.csv")
    out = {}
    if not cc:
        return out
    for r in csv.DictReader(open(cc)):
        c = r.get("Counter_Name")
        if c not in counters:
            continue
        a = out.setdefault(short(r["Kernel_Name"]), {}).setdefault(c, [0, 0.0])
        a[0] += 1
        a[1] += float(r["Counter_Value"])
    return {k: {c: v[1] / v[0] for c, v in d.items()} | {"dispatches": max(v[0] for v in d.values())} for k, d in out.items()}


def main():
    root, tag = sys.argv[1], sys.argv[2]
    doc = {"tag": tag, "units": "per dispatch averages; bytes corrected as in the module docstring", "kernels": {}}
    stats = find(os.path.join(root, "trace"), "*kernel_stats.csv")
    dur = {}
    if stats:
        for r in csv.DictReader(open(stats)):
            dur[short(r["Name"])] = {"calls": int(r["Calls"]), "avg_us": float(r["AverageNs"]) / 1e3,
                                     "total_ms": float(r["TotalDurationNs"]) / 1e6}
    hit = per_kernel(root, "pmc_hit", ("TCC_HIT_sum", "TCC_MISS_sum"))
    fetch = per_kernel(root, "pmc_fetch", ("FETCH_SIZE",))
    write = per_kernel(root, "pmc_write", ("WRITE_SIZE",))
    req = per_kernel(root, "pmc_req", ("TCC_REQ_sum", "TCC_READ_sum"))
    names = sorted(set(dur) | set(hit), key=lambda k: -dur.get(k, {}).get("total_ms", 0.0))
    for k in names[:40]:
        e = dict(dur.get(k, {}))
        h = hit.get(k)
        if h:
            hh, mm = h.get("TCC_HIT_sum", 0.0), h.get("TCC_MISS_sum", 0.0)
            e.update(l2_hits=hh, l2_misses=mm, l2_hit_rate=(hh / (hh + mm) if hh + mm else None))
        if k in fetch:
            e["fabric_read_bytes"] = 2.0 * 1024.0 * fetch[k]["FETCH_SIZE"]
        if k in write:
            e["fabric_write_bytes"] = 1024.0 * write[k]["WRITE_SIZE"]
        if k in req:
            e["l2_requests"] = req[k].get("TCC_REQ_sum")
            e["l2_read_requests"] = req[k].get("TCC_READ_sum")
        if e.get("avg_us") and "fabric_read_bytes" in e:
            e["fabric_GBps"] = (e["fabric_read_bytes"] + e.get("fabric_write_bytes", 0.0)) / (e["avg_us"] * 1e-6) / 1e9
        doc["kernels"][k] = e
    print(json.dumps(doc, indent=1))


if __name__ == "__main__":
    main()
