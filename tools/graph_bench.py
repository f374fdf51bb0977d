"""hipGraph mode (options={'hip_graph': True}) vs the eager path on launch-latency-bound solves (run on the GPU
box): wall time of `odeint`, best of 3, same process; the solutions must be identical."""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torchdiffeq_amd as tda  # noqa: E402
from _cases import PlanarCNF, load  # noqa: E402

dev = torch.device("cuda:0")


def best_of(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    best, out = None, None
    for _ in range(reps):
        t = time.perf_counter()
        out = fn()
        torch.cuda.synchronize()
        d = time.perf_counter() - t
        best = d if best is None else min(best, d)
    return best, out


def main():
    res = {}
    A = torch.tensor([[-0.1, 2.0], [-2.0, -0.1]], device=dev)
    spiral = lambda t, y: (y ** 3) @ A
    y0 = torch.tensor([[2.0, 0.0]], device=dev)
    torch.manual_seed(0)
    mlp = torch.nn.Sequential(torch.nn.Linear(64, 64), torch.nn.Tanh(), torch.nn.Linear(64, 64), torch.nn.Tanh(),
                              torch.nn.Linear(64, 64)).to(dev)
    ymlp = torch.randn(256, 64, device=dev)
    cnf = PlanarCNF(load("cnf.npz"), dev)
    g = torch.Generator().manual_seed(11)
    z0 = torch.randn(32768, 2, generator=g).to(dev)
    cases = {
        "spiral (ode_demo.py), dopri5, 1000 output times": (spiral, y0, torch.linspace(0.0, 25.0, 1000, device=dev),
                                                            "dopri5", dict(rtol=1e-7, atol=1e-9)),
        "spiral, rk4 fixed grid, 1000 points (cfg1)": (spiral, y0, torch.linspace(0.0, 25.0, 1000, device=dev),
                                                      "rk4", {}),
        "MLP 64-64-64 field, 256x64 state, dopri5": (lambda t, y: mlp(y), ymlp, torch.tensor([0.0, 1.0], device=dev),
                                                     "dopri5", dict(rtol=1e-6, atol=1e-8)),
        "CNF forward (cfg5 state: 32768x2 + logp), dopri5": (cnf, (z0, torch.zeros(32768, 1, device=dev)),
                                                            torch.tensor([10.0, 0.0], device=dev), "dopri5",
                                                            dict(rtol=1e-5, atol=1e-5)),
    }
    for name, (f, y, t, method, kw) in cases.items():
        with torch.no_grad():
            te, ye = best_of(lambda: tda.odeint(f, y, t, method=method, **kw))
            tg, yg = best_of(lambda: tda.odeint(f, y, t, method=method, options=dict(hip_graph=True), **kw))
        ye = ye if isinstance(ye, tuple) else (ye,)
        yg = yg if isinstance(yg, tuple) else (yg,)
        res[name] = {"eager_s": te, "hip_graph_s": tg, "speedup": te / tg,
                     "identical": all(torch.equal(a, b) for a, b in zip(ye, yg))}
        print(name, res[name], flush=True)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(res, open(os.path.join(ROOT, "gpurun_out", "graph_bench.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
