#!/usr/bin/env python
"""Cache-policy sweep of the streaming launches IN SITU (r05): TDEQ_COMBINE_POLICY = 0 (default loads / stores),
1 (non-temporal loads), 2 (non-temporal stores), 3 (both), and "auto" (tdeq_abi.hip stream_policy(): non-temporal for
launches whose streams exceed TDEQ_NT_THRESHOLD_MB) on

  cfg2  dopri5 fp32 65536 x 128 trial steps (bench.py's timed region: ms per step + the dominant launch's duration),
  cfg4  dopri8 fp64 16384 x 512 whole odeint (ms, NFE) + its dominant launch (939 MB) in situ and cold,
  cfg3  the full adjoint pass (ms per forward + backward).

The policy is read once per process, so every setting runs in a subprocess of this script.

    python tools/policy_sweep.py > gpurun_out/r05_policy_sweep.json
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import json, os, sys, time
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
sys.argv = ["bench.py"]
import importlib.util, argparse, torch
spec = importlib.util.spec_from_file_location("bench", os.path.join({root!r}, "bench.py"))
b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
out = {{}}
args = argparse.Namespace(gpus=1, steps=100, warmup=20, workload="linear", scaling="weak", no_cpu_baseline=True, no_extras=True)
o, field, y0 = b.run_linear(args, 0, 1, dev)
rf = o["roofline"]
out["cfg2"] = {{"ms_per_step": o["ms_per_step"], "blocks": o["blocks"]["ms_per_step"], "dominant_avg_launch_ms": rf["avg_launch_ms"],
               "dominant_frac": rf["frac"], "dominant_cold_ms": rf["cold"]["avg_launch_ms"], "dominant_cold_frac": rf["cold"]["frac"],
               "solver_only_us_per_step": o["solver_only"]["us_per_step"], "rel_err_vs_reference": o["rel_err_vs_reference"],
               "nfe": o["nfe"]}}
del o, field, y0
torch.cuda.empty_cache()
c4 = b.cfg4_config(dev)
out["cfg4"] = {{"odeint_t01_ms": c4["odeint_t01_ms"], "nfe": c4["nfe"], "rel_err_vs_reference": c4["rel_err_vs_reference"],
               "dominant_avg_launch_ms": c4["roofline"]["avg_launch_ms"], "dominant_frac": c4["roofline"]["frac"],
               "dominant_cold_ms": c4["roofline"]["cold"]["avg_launch_ms"], "dominant_cold_frac": c4["roofline"]["cold"]["frac"]}}
torch.cuda.empty_cache()
a = b.adjoint_pass(1, 0, dev, b.ADJ_BATCH, 3, 1)
out["cfg3"] = {{"ms_per_pass": a["ms_per_pass"], "fwd_ms": a["fwd_ms"], "bwd_ms": a["bwd_ms_incl_allreduce"], "nfe_fwd": a["nfe_fwd"],
               "nfe_bwd": a["nfe_bwd"]}}
print("RESULT " + json.dumps(out))
"""


def main():
    settings = sys.argv[1:] or ["0", "1", "2", "3", "auto"]
    results = {}
    for pol in settings:
        env = dict(os.environ, TDEQ_COMBINE_POLICY=pol.split(":")[0])
        if ":" in pol:          # "auto:128" = size switch at 128 MiB
            env["TDEQ_NT_THRESHOLD_MB"] = pol.split(":")[1]
        r = subprocess.run([sys.executable, "-c", CHILD.format(root=ROOT)], env=env, capture_output=True, text=True,
                           timeout=900)
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("RESULT ")]
        results[pol] = json.loads(line[-1][7:]) if line else {"error": (r.stdout[-500:] + r.stderr[-1500:])}
        sys.stderr.write(f"policy {pol}: {json.dumps(results[pol])[:400]}\n")
    print(json.dumps({"what": "TDEQ_COMBINE_POLICY sweep in situ (tools/policy_sweep.py); bit 0 = non-temporal loads, "
                              "bit 1 = non-temporal stores, auto = by launch size (TDEQ_NT_THRESHOLD_MB, default "
                              "include/tdeq_hip.h TDEQ_NT_THRESHOLD_DEFAULT_MB)",
                      "policies": results}, indent=1))


if __name__ == "__main__":
    main()
