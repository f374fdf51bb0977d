"""Roofline table of the tdeq kernels of one profiled run, from the committed summaries under profiles/:

    python tools/roofline_table.py <tag> [--n ELEMENTS --w BYTES_PER_WORD]

Reads profiles/<tag>_kernel_stats.csv (rocprofv3 --kernel-trace --stats: average duration per kernel) and
profiles/<tag>_pmc_hbm.json (HBM bytes per launch from the separate --pmc FETCH_SIZE / WRITE_SIZE passes, corrected as
MI355X_MICROARCH.md prescribes).  Per kernel: average duration, HBM bytes per launch by the counters, achieved
GB/s = counter bytes / duration, fraction of the 8 TB/s peak and — when the state size is given (one size per kernel
name in the run) — the algorithmic bytes (SURVEY.md §8d words x N x w) and traffic / algorithmic.
Writes profiles/<tag>_roofline.json and prints a markdown table."""
import argparse
import csv
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PEAK = 8.0e12


def words(name):
    """Algorithmic words per element of a kernel instantiation (reads + writes), or None."""
    m = re.match(r"tdeq::(\w+)<(?:float|double)(?:, )?([^>]*)>", name)
    if not m:
        return None
    kern, args = m.group(1), [a.strip() for a in m.group(2).split(",") if a.strip()]
    nt = int(args[0]) if args and args[0].lstrip("-").isdigit() else None
    table = {"stage_combine_kernel": lambda: nt + 2, "stage_combine_fill_kernel": lambda: nt + 2,
             "stage_combine_err_kernel": lambda: nt + 3, "stage_combine_sel_kernel": lambda: 3,
             "error_norm_partial_kernel": lambda: nt + 3, "error_norm_kernel": lambda: nt + 2,
             "dense_kernel": lambda: nt + 3, "init_norms_kernel": lambda: 2 if nt == 0 else 3}
    f = table.get(kern)
    return f() if f else None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("tag")
    ap.add_argument("--n", type=int, default=None)
    ap.add_argument("--w", type=int, default=4)
    a = ap.parse_args()
    stats = list(csv.DictReader(open(os.path.join(ROOT, "profiles", f"{a.tag}_kernel_stats.csv"))))
    pmc = json.load(open(os.path.join(ROOT, "profiles", f"{a.tag}_pmc_hbm.json")))["kernels"]
    rows = []
    for r in stats:
        name = r["Name"].split("(")[0].replace("void ", "")
        if not name.startswith("tdeq::"):
            continue
        dur = float(r["AverageNs"]) * 1e-9
        hbm = pmc.get(name, {}).get("hbm_bytes_per_launch")
        row = {"kernel": name, "calls": int(r["Calls"]), "avg_us": dur * 1e6, "percent_of_gpu_time": float(r["Percentage"]),
               "hbm_bytes_per_launch": hbm, "GBps": hbm / dur / 1e9 if hbm else None,
               "frac_of_peak": hbm / dur / PEAK if hbm else None}
        wds = words(name)
        if a.n and wds:
            alg = wds * a.n * a.w
            row.update(algorithmic_bytes=alg, algorithmic_GBps=alg / dur / 1e9, algorithmic_frac=alg / dur / PEAK,
                       traffic_over_algorithmic=(hbm / alg) if hbm else None)
        rows.append(row)
    json.dump({"tag": a.tag, "n": a.n, "w": a.w, "peak_GBps": PEAK / 1e9, "kernels": rows},
              open(os.path.join(ROOT, "profiles", f"{a.tag}_roofline.json"), "w"), indent=1)
    print("| kernel | calls | avg µs | % GPU time | HBM bytes/launch (PMC) | GB/s | frac of 8 TB/s |"
          + (" algorithmic bytes | traffic/alg |" if a.n else ""))
    print("|---|---|---|---|---|---|---|" + ("---|---|" if a.n else ""))
    for r in rows:
        if r["avg_us"] < 8 and not r.get("algorithmic_bytes"):
            continue
        line = "| `{}` | {} | {:.1f} | {:.1f} | {} | {} | {} |".format(
            r["kernel"].replace("tdeq::", ""), r["calls"], r["avg_us"], r["percent_of_gpu_time"],
            "{:.1f} MB".format(r["hbm_bytes_per_launch"] / 1e6) if r["hbm_bytes_per_launch"] else "—",
            "{:.0f}".format(r["GBps"]) if r["GBps"] else "—",
            "{:.2f}".format(r["frac_of_peak"]) if r["frac_of_peak"] else "—")
        if a.n:
            line += " {} | {} |".format(
                "{:.1f} MB".format(r["algorithmic_bytes"] / 1e6) if r.get("algorithmic_bytes") else "—",
                "{:.3f}".format(r["traffic_over_algorithmic"]) if r.get("traffic_over_algorithmic") else "—")
        print(line)


if __name__ == "__main__":
    main()
