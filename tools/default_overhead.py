"""r06: what the built-in default (hip_graph='auto') costs per SOLVE over hip_graph=True once a func's steps are captured —
the cache-key walk over func, and the per-solve re-check (one combine launch, one evaluation of func, one comparison) — and
what it gains over the eager loop, on short solves of a training-loop shape.  Prints one JSON object
(-> profiles/r06_default_overhead.json)."""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.pop("TDEQ_HIP_GRAPH", None)
import torchdiffeq_amd as tda  # noqa: E402

dev = torch.device("cuda:0")
res = {"unit": "ms per odeint call (median of 5 x 40 calls after 6 warm-up calls), dopri5, t = [0, 0.5, 1], rtol 1e-6"}
for rows, d, width in ((64, 8, 32), (4096, 32, 64), (8192, 128, 0)):
    torch.manual_seed(0)
    if width:
        net = torch.nn.Sequential(torch.nn.Linear(d, width), torch.nn.Tanh(), torch.nn.Linear(width, d)).to(dev)
    else:
        net = torch.nn.Linear(d, d, bias=False).to(dev)

    class F(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.net = net

        def forward(self, t, y):
            return self.net(y)
    y0 = torch.randn(rows, d, device=dev)
    t = torch.tensor([0.0, 0.5, 1.0], device=dev)
    entry = {}
    for name, opts in (("eager", dict(hip_graph=False)), ("hip_graph_true", dict(hip_graph=True)), ("default", None)):
        f = F()
        with torch.no_grad():
            for _ in range(6):
                tda.odeint(f, y0, t, method="dopri5", rtol=1e-6, atol=1e-8, options=opts)
            torch.cuda.synchronize()
            blocks = []
            for _ in range(5):
                t0 = time.perf_counter()
                for _ in range(40):
                    tda.odeint(f, y0, t, method="dopri5", rtol=1e-6, atol=1e-8, options=opts)
                torch.cuda.synchronize()
                blocks.append((time.perf_counter() - t0) / 40)
        entry[name] = round(1e3 * sorted(blocks)[2], 4)
    entry["default_minus_true_ms"] = round(entry["default"] - entry["hip_graph_true"], 4)
    res[f"{rows}x{d}" + (f" MLP {width}" if width else " linear")] = entry
print(json.dumps(res, indent=1))
