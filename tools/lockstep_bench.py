"""Cost of lock-step sharding per trial step, measured on ONE GPU at world size 1 through RCCL (the collectives
run for real: one all-reduce of 3 doubles per trial step): cfg2's state, three controllers —
    independent      no collective (the default of a sharded solve)
    lockstep_host    TDEQ_LOOKAHEAD=0: sums -> pinned host -> tensor -> all_reduce -> host controller (r01's path)
    lockstep_device  sums stay on the device: finalize -> all_reduce (RCCL) -> tdeq_step_controller -> look-ahead stage
Prints one JSON object (-> profiles/r02_lockstep_bench.json)."""
import json
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from torchdiffeq_amd.misc import OdeFunc, StateLayout, rms_norm  # noqa: E402
from torchdiffeq_amd.solvers import Dopri5Solver  # noqa: E402

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29641")
torch.cuda.set_device(0)
dist.init_process_group(backend="nccl", rank=0, world_size=1)
dev = torch.device("cuda:0")
res = {"world_size": 1, "backend": dist.get_backend(), "unit": "ms per dopri5 trial step (median of 3 blocks of 100)"}
for rows in (65536, 8192):
    A, y0 = bench.make_problem(dev, rows=slice(0, rows))
    At = A.T.contiguous()
    field = lambda t, y: y @ At
    entry = {}
    for name, sync, look in (("independent", None, "1"), ("lockstep_host", dist.group.WORLD, "0"),
                             ("lockstep_device", dist.group.WORLD, "1")):
        os.environ["TDEQ_LOOKAHEAD"] = look
        layout = StateLayout([y0.shape], False)
        func = OdeFunc(field, layout, 1.0, y0.dtype, dev)
        solver = Dopri5Solver(func=func, y0=y0.reshape(-1), rtol=bench.RTOL, atol=bench.ATOL, norm=rms_norm,
                              dist_sync=sync)
        solver._before_integrate([0.0])
        solver._t_end = float("inf")
        blocks = bench.time_steps(solver, 100, 20, 1, dev, n_blocks=3)
        entry[name] = {"ms_per_step": bench.block_stats(blocks, 100)["median"], "lookahead": bool(solver._lookahead),
                       "accepted": solver.n_accepted, "rejected": solver.n_rejected}
    res[f"{rows}x128"] = entry
os.environ.pop("TDEQ_LOOKAHEAD", None)
dist.destroy_process_group()
print(json.dumps(res, indent=1))
