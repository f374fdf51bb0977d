"""300 captured trial steps (hip_graph mode) of the cfg2 field on the 8192 x 128 shard — the command traced by
rocprofv3 to see where a graph-mode step's time goes (kernel durations vs dependency gaps)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
dev = torch.device("cuda:0")
A, y0 = bench.make_problem(dev, rows=slice(0, rows))
At = A.T.contiguous()
solver = bench.make_stepper(lambda t, y: y @ At, y0, hip_graph=True)
blocks = bench.time_steps(solver, 100, 20, 1, dev, n_blocks=3)
print("ms per step", bench.block_stats(blocks, 100))
