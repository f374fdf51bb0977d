"""Per-kernel durations of one dopri5 trial step of cfg2's field at several shard sizes, captured (hipGraph replay) vs
eager (look-ahead): where a captured step of a MEDIUM state (2-8 M elements) loses to the eager one.  GPU box:
python tools/graph_vs_eager_breakdown.py -> JSON."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

dev = torch.device("cuda:0")
out = {}
for rows in (65536, 32768, 16384, 8192):
    A, y0 = bench.make_problem(dev, rows=slice(0, rows))
    At = A.T.contiguous()
    field = lambda t, y: y @ At
    for name, kw in (("lookahead", dict(lookahead=True)), ("hip_graph", dict(hip_graph=True))):
        from torchdiffeq_amd import solvers
        solvers.adaptive._GRAPH_MODE_MAX_ELEMENTS = solvers.fixed._GRAPH_MODE_MAX_ELEMENTS = 1 << 24          # let the captured path run at every size for this comparison
        solver = bench.make_stepper(field, y0, **kw)
        st = bench.block_stats(bench.time_steps(solver, 100, 30, 1, dev, n_blocks=3), 100)
        solver = bench.make_stepper(field, y0, **kw)
        with torch.no_grad():
            for _ in range(30):
                solver._trial_step()
            bd = bench.kernel_breakdown(solver._trial_step, 50)
            if solver._g is not None:
                torch.cuda.synchronize()
                solver._g.release()
        out[f"{rows}x128 {name}"] = {"ms_per_step": st["median"], "floor_us": bd["floor_us"], "func_us": bd["func_kernel_us"],
                                     "solver_us": bd["solver_kernel_us"], "dispatches": bd["dispatches_per_call"],
                                     "kernels": {k[:70]: round(v["avg_us"], 2) for k, v in bd["top_kernels"].items()}}
        del solver
        torch.cuda.empty_cache()
print(json.dumps(out, indent=1))
