"""cProfile of the eager look-ahead trial step on the GPU box's host at a launch-bound state (8192 x 128): where
do the ~230 us of host time per step go?"""
import cProfile
import os
import pstats
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

dev = torch.device("cuda:0")
A, y0 = bench.make_problem(dev, rows=slice(0, 8192))
At = A.T.contiguous()
field = lambda t, y: y @ At
s = bench.make_stepper(field, y0, lookahead=True)
with torch.no_grad():
    for _ in range(100):
        s._trial_step()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(1000):
        s._trial_step()
    torch.cuda.synchronize()
    print("us/step", (time.perf_counter() - t) / 1000 * 1e6)
    # raw cost of the pieces
    y = y0.clone()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(2000):
        y @ At
    torch.cuda.synchronize()
    print("us per bare matmul dispatch (async, includes GPU if GPU-bound)", (time.perf_counter() - t) / 2000 * 1e6)
    t = time.perf_counter()
    for _ in range(2000):
        torch.empty_like(y)
    print("us per empty_like", (time.perf_counter() - t) / 2000 * 1e6)
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(1000):
        s._trial_step()
    pr.disable()
    torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(25)
