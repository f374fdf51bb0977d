"""r06 ("act once"): the norm launch's reduction chunk at SHARD sizes.  profiles/r06_shard_l2.json showed
the 1/8 shard's `error_norm_partial_kernel` at 6.6 us for 16.8 MB (2.5 TB/s) while the combines next to it move 29 MB in
3.8 us: at 2^20 elements and 2048 elements per workgroup the launch has 512 workgroups — two per CU — and is latency-,
not bandwidth-bound.  Per state size and chunk (TDEQ_CHUNK): the captured dopri5 trial step of cfg2's field and the
per-kernel durations inside it.  Prints one JSON object (-> profiles/r06_chunk_sweep.json)."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from torchdiffeq_amd import solvers  # noqa: E402

solvers.adaptive._GRAPH_MODE_MAX_ELEMENTS = solvers.fixed._GRAPH_MODE_MAX_ELEMENTS = 1 << 24

dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
res = {"field": "dy/dt = A y, dim 128 fp32 (cfg2's field)",
       "unit": "ms per dopri5 trial step (median of 3 blocks of 100); kernel durations in us from torch.profiler"}
for rows in (2048, 8192, 16384, 32768, 65536):
    A, y0 = bench.make_problem(dev, rows=slice(0, rows))
    At = A.T.contiguous()
    entry = {"elements": rows * bench.DIM}
    for chunk in (1024, 2048, 4096):
        os.environ["TDEQ_CHUNK"] = str(chunk)
        for name, kw in (("hip_graph", dict(hip_graph=True)), ("lookahead", dict(lookahead=True, hip_graph=False))):
            field = lambda t, y: y @ At                   # a fresh func object per variant: its own captured step
            solver = bench.make_stepper(field, y0, **kw)
            assert solver.plan.chunk == chunk
            blocks = bench.time_steps(solver, 100, 20, 1, dev, n_blocks=3)
            e = {"ms_per_step": bench.block_stats(blocks, 100)["median"]}
            if name == "hip_graph":
                with torch.no_grad():
                    bd = bench.kernel_breakdown(solver._trial_step, 50)
                e["solver_kernel_us"] = bd["solver_kernel_us"]
                e["norm_kernels_us"] = {k[:60]: round(v["avg_us"], 2) for k, v in bd["top_kernels"].items() if "norm" in k}
            entry[f"chunk{chunk}_{name}"] = e
            if solver._g is not None:
                solver._g.release()
    os.environ.pop("TDEQ_CHUNK", None)
    res[f"{rows}x{bench.DIM}"] = entry
print(json.dumps(res, indent=1))
