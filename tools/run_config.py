"""Run one of BASELINE.json's configurations at full size on the GPU a few times (the command profiled by
tools/profile_cmd.sh).  Usage: run_config.py <cfg2|cfg2_shard|cfg4|cfg3|cfg3_shard|cfg5> [reps]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torchdiffeq_amd as tda  # noqa: E402
import _fullsize as fs  # noqa: E402

case = sys.argv[1]
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dev = torch.device("cuda:0")


def linear(B, D, dtype, method, rtol, atol):
    A, y0 = fs.linear_problem(B, D, dtype)
    At, y0 = A.T.contiguous().to(dev), y0.to(dev)
    t = torch.tensor([0.0, 1.0], dtype=dtype, device=dev)
    field = lambda tt, y: y @ At         # ONE func object: a fresh lambda per call would be a first sight for 'auto' every time
    return lambda: tda.odeint(field, y0, t, rtol=rtol, atol=atol, method=method)


def cfg3(rows):
    field, y0 = fs.cfg3_problem(rows)
    field, y0 = field.to(dev), y0.to(dev)
    t = torch.tensor([0.0, 1.0], device=dev)

    def run():
        for p in field.parameters():
            p.grad = None
        x = y0.clone().requires_grad_(True)
        y = tda.odeint_adjoint(field, x, t, rtol=1e-5, atol=1e-7, method="dopri5")
        y[-1].pow(2).sum().backward()
    return run


def cfg5():
    z = fs.load("cfg5")
    cnf = fs.ExampleCNF([z[f"p{i}"] for i in range(6)], trace="closed").to(dev)
    z0, logp0 = fs.cfg5_problem()
    z0, logp0 = z0.to(dev), logp0.to(dev)
    t = torch.tensor([10.0, 0.0], device=dev)

    def run():
        for p in cnf.parameters():
            p.grad = None
        x = z0.clone().requires_grad_(True)
        zt, lp = tda.odeint_adjoint(cnf, (x, logp0), t, atol=1e-5, rtol=1e-5, method="dopri5")
        (lp[-1].mean() - zt[-1].pow(2).sum() / 100).backward()
    return run


fn = {"cfg2": lambda: linear(65536, 128, torch.float32, "dopri5", 1e-7, 1e-9),
      "cfg4": lambda: linear(16384, 512, torch.float64, "dopri8", 1e-9, 1e-11),
      "cfg2_shard": lambda: linear(8192, 128, torch.float32, "dopri5", 1e-7, 1e-9),      # the 1/8 strong-scaling shard
      "cfg3": lambda: cfg3(None), "cfg3_shard": lambda: cfg3(slice(0, 8192)), "cfg5": cfg5}[case]()
no_grad = case in ("cfg2", "cfg4", "cfg2_shard")
ctx = torch.no_grad() if no_grad else torch.enable_grad()
with ctx:
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
print(f"{case}: {1e3 * (time.perf_counter() - t0) / reps:.2f} ms per pass over {reps} passes")
