"""Trial-step time of the cfg2 field vs state size on one MI355X, on the three step paths (host-driven loop,
look-ahead controller, captured hipGraph): where does a step stop being launch-latency-bound?  Sets
solvers._GRAPH_AUTO_MAX_ELEMENTS (`hip_graph="auto"`).  Prints one JSON object (-> profiles/r02_shard_regime.json)."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from torchdiffeq_amd import solvers  # noqa: E402

solvers.adaptive._GRAPH_MODE_MAX_ELEMENTS = solvers.fixed._GRAPH_MODE_MAX_ELEMENTS = 1 << 24      # measure the captured step beyond its shipped size limit too

dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
res = {"field": "dy/dt = A y, dim 128 fp32 (cfg2's field)", "unit": "ms per dopri5 trial step, median of 3 blocks of 100"}
for rows in (512, 2048, 8192, 16384, 32768, 65536):
    A, y0 = bench.make_problem(dev, rows=slice(0, rows))
    At = A.T.contiguous()
    field = lambda t, y: y @ At
    entry = {"elements": rows * bench.DIM}
    for name, kw in (("host_driven", dict(lookahead=False)), ("lookahead", dict(lookahead=True)),
                     ("hip_graph", dict(hip_graph=True))):
        try:
            solver = bench.make_stepper(field, y0, **kw)
            blocks = bench.time_steps(solver, 100, 20, 1, dev, n_blocks=3)
            entry[name] = bench.block_stats(blocks, 100)["median"]
        except Exception as exc:
            entry[name] = repr(exc)
    res[f"{rows}x{bench.DIM}"] = entry
print(json.dumps(res, indent=1))
