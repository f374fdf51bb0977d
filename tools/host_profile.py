#!/usr/bin/env python
"""Where the HOST time of an eager trial step goes (GPU box): cProfile of dopri5 trial steps on a state small enough
(8192 x 128 fp32, or the bf16 headline shape) that the loop is host-bound.

    python tools/host_profile.py [fp32|bf16] [steps]"""
import cProfile
import io
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "fp32"
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
    dev = torch.device("cuda", 0)
    A, y0 = bench.make_problem(dev)
    if which == "bf16":
        A = (A + 0.1 * torch.eye(A.shape[0], device=dev)).to(torch.bfloat16)
        y0 = y0.to(torch.bfloat16)
        kw = dict(rtol=1e-2, atol=1e-3)
    else:
        y0 = y0[:8192].contiguous()
        kw = {}
    At = A.T.contiguous()
    with torch.no_grad():
        solver = bench.make_stepper(lambda t, y: y @ At, y0, **kw)
        for _ in range(20):
            solver._trial_step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            solver._trial_step()
        torch.cuda.synchronize()
        print("ms_per_step", 1e3 * (time.perf_counter() - t0) / steps)
        solver = bench.make_stepper(lambda t, y: y @ At, y0, **kw)
        for _ in range(20):
            solver._trial_step()
        pr = cProfile.Profile()
        pr.enable()
        for _ in range(steps):
            solver._trial_step()
        pr.disable()
        torch.cuda.synchronize()
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28)
    print(s.getvalue()[:6000])
