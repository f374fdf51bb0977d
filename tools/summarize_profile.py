"""Condense rocprofv3 output (kernel stats + PMC passes) into small files under gpurun_out/prof_<tag>/summary
that are then committed under profiles/.  Usage: summarize_profile.py <prof_dir> <tag>"""
import csv
import glob
import json
import os
import sys


def find(root, pattern):
    hits = sorted(glob.glob(os.path.join(root, "**", pattern), recursive=True))
    return hits[0] if hits else None


def main():
    root, tag = sys.argv[1], sys.argv[2]
    out_dir = os.path.join(root, "summary")
    os.makedirs(out_dir, exist_ok=True)
    summary = {"tag": tag}
    stats = find(os.path.join(root, "trace"), "*kernel_stats.csv")
    if stats:
        rows = list(csv.DictReader(open(stats)))
        with open(os.path.join(out_dir, f"{tag}_kernel_stats.csv"), "w") as f:
            f.write(open(stats).read())
        summary["kernel_stats"] = rows[:25]
        print("== kernel stats (top 15 by total time) ==")
        for r in rows[:15]:
            print({k: r[k] for k in r if k in ("Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs")})
    trace = find(os.path.join(root, "trace"), "*kernel_trace.csv")
    if trace:
        rows = list(csv.DictReader(open(trace)))
        rows.sort(key=lambda r: int(r["Start_Timestamp"]))
        # timeline of the last ~40 dispatches: name, duration, gap to previous
        tl = []
        prev_end = None
        for r in rows[-60:]:
            s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
            tl.append({"kernel": r["Kernel_Name"][:70], "dur_us": (e - s) / 1e3,
                       "gap_us": None if prev_end is None else (s - prev_end) / 1e3})
            prev_end = e
        summary["timeline_tail"] = tl
        busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows[-600:])
        span = int(rows[-1]["End_Timestamp"]) - int(rows[-600]["Start_Timestamp"]) if len(rows) >= 600 else None
        summary["gpu_busy_frac_last600"] = busy / span if span else None
        print("gpu busy fraction over the last 600 dispatches:", summary["gpu_busy_frac_last600"])
        for t in tl[-26:]:
            print(t)
        # Per-trial-step timeline of the TIMED region (steps that end in the device controller kernel): period
        # between consecutive controller kernels, GPU-busy time inside it, and where the idle gaps sit.
        marks = [i for i, r in enumerate(rows) if "norm_finalize_ctrl_kernel" in r["Kernel_Name"]]
        if len(marks) < 10:
            marks = [i for i, r in enumerate(rows) if "norm_finalize_kernel" in r["Kernel_Name"]]
        spans = []
        for a, b in zip(marks[:-1], marks[1:]):
            seg = rows[a:b + 1]
            if not any(r["Kernel_Name"].startswith("Cijk") for r in seg):
                continue      # solver-only passes at the end of bench.py (no func launches)
            period = (int(seg[-1]["End_Timestamp"]) - int(seg[0]["End_Timestamp"])) / 1e3
            busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in seg[1:]) / 1e3
            gaps = []
            for prev, cur in zip(seg[:-1], seg[1:]):
                g = (int(cur["Start_Timestamp"]) - int(prev["End_Timestamp"])) / 1e3
                key = prev["Kernel_Name"].split("(")[0].replace("void ", "")[:48] + " -> " + \
                    cur["Kernel_Name"].split("(")[0].replace("void ", "")[:48]
                gaps.append((key, g))
            spans.append((period, busy, gaps))
        if spans:
            # drop the spans that contain non-step work (the parity solve before the timed loop, allocations)
            mid0 = sorted(p for p, _, _ in spans)[len(spans) // 2]
            spans = [sp for sp in spans if sp[0] <= 2.0 * mid0]
            n = len(spans)
            periods = [sp[0] for sp in spans]
            busys = [sp[1] for sp in spans]
            gap_after = {}
            for _, _, gaps in spans:
                for key, g in gaps:
                    acc = gap_after.setdefault(key, [0, 0.0])
                    acc[0] += 1
                    acc[1] += g
            summary["step_timeline"] = {
                "steps": n, "median_period_us": sorted(periods)[n // 2], "mean_period_us": sum(periods) / n,
                "mean_gpu_busy_us": sum(busys) / n, "mean_idle_us": (sum(periods) - sum(busys)) / n,
                "mean_gap_us_by_boundary": {k: v[1] / v[0] for k, v in sorted(gap_after.items(),
                                                                             key=lambda kv: -kv[1][1])[:16]}}
            print("== per-step timeline (timed region) ==")
            print(json.dumps(summary["step_timeline"], indent=1))
    for name, counter in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
        cc = find(os.path.join(root, name), "*counter_collection.csv")
        if not cc:
            continue
        rows = list(csv.DictReader(open(cc)))
        agg = {}
        for r in rows:
            if r.get("Counter_Name") != counter:
                continue
            k = r["Kernel_Name"]
            a = agg.setdefault(k, [0, 0.0])
            a[0] += 1
            a[1] += float(r["Counter_Value"])
        per = {k: {"dispatches": v[0], "avg_" + counter: v[1] / v[0]} for k, v in agg.items()}
        summary[name] = per
        print(f"== {counter} per kernel (avg per dispatch, raw counter units) ==")
        for k, v in sorted(per.items(), key=lambda kv: -kv[1]["avg_" + counter])[:12]:
            print(k[:80], v)
    # HBM traffic per launch of the tdeq kernels: FETCH_SIZE/WRITE_SIZE are reported in KiB; on gfx950 the
    # read counter tallies the 128-B requests of a 16-B/lane streaming read at 64 B, i.e. reports half the
    # bytes (MI355X_MICROARCH.md, "HBM") -> x2; WRITE_SIZE is used as reported (it matches the one N-sized
    # store of every streaming kernel exactly, which calibrates it for this access pattern).
    if "pmc_fetch" in summary and "pmc_write" in summary:
        hbm = {}
        for k, v in summary["pmc_fetch"].items():
            if "tdeq::" not in k:
                continue
            w = summary["pmc_write"].get(k, {}).get("avg_WRITE_SIZE", 0.0)
            rd = 2.0 * v["avg_FETCH_SIZE"] * 1024.0
            wr = w * 1024.0
            short = k.split("(")[0].replace("void ", "")
            hbm[short] = {"read_bytes": rd, "write_bytes": wr, "hbm_bytes_per_launch": rd + wr,
                          "dispatches": v["dispatches"]}
        summary["hbm_traffic"] = hbm
        json.dump({"tag": tag, "corrections": "FETCH_SIZE KiB x2 (gfx950 half-count), WRITE_SIZE KiB x1",
                   "kernels": hbm}, open(os.path.join(out_dir, f"{tag}_pmc_hbm.json"), "w"), indent=1)
    json.dump(summary, open(os.path.join(out_dir, f"{tag}_summary.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
