"""Soak: many thousands of trial steps on the three step paths — device memory must not grow and results must stay
finite (leak / drift check of the host loop, the captured-step cache and the read-back machinery)."""
import gc
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import torchdiffeq_amd as tda  # noqa: E402

dev = torch.device("cuda:0")
A, y0 = bench.make_problem(dev, rows=slice(0, 8192))
At = A.T.contiguous()
field = lambda t, y: y @ At
res = {}
for name, kw in (("host_driven", dict(lookahead=False)), ("lookahead", dict(lookahead=True)), ("hip_graph", dict(hip_graph=True))):
    s = bench.make_stepper(field, y0, **kw)
    with torch.no_grad():
        for _ in range(500):
            s._trial_step()
        torch.cuda.synchronize(); gc.collect()
        m0 = torch.cuda.memory_allocated()
        for _ in range(20000):
            s._trial_step()
        torch.cuda.synchronize(); gc.collect()
        m1 = torch.cuda.memory_allocated()
    res[name] = {"steps": 20000, "bytes_before": m0, "bytes_after": m1, "accepted": s.n_accepted, "rejected": s.n_rejected,
                 "finite": bool(torch.isfinite(s.y1).all()), "t": s.t1}
    if s._g is not None:
        s._g.release()
# repeated full solves (adjoint with captured steps): the cache must stay bounded
net = torch.nn.Sequential(torch.nn.Linear(16, 32), torch.nn.Tanh(), torch.nn.Linear(32, 16)).to(dev)


class F(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.net = net

    def forward(self, t, y):
        return self.net(y)


f = F()
x0 = torch.randn(64, 16, device=dev)
t = torch.tensor([0.0, 0.5, 1.0], device=dev)


def one():
    for p in f.parameters():
        p.grad = None
    x = x0.clone().requires_grad_(True)
    tda.odeint_adjoint(f, x, t, rtol=1e-4, atol=1e-6, options=dict(hip_graph=True))[-1].sum().backward()


for _ in range(5):
    one()
torch.cuda.synchronize(); gc.collect()
m0 = torch.cuda.memory_allocated()
for _ in range(300):
    one()
torch.cuda.synchronize(); gc.collect()
res["adjoint_hip_graph_300_passes"] = {"bytes_before": m0, "bytes_after": torch.cuda.memory_allocated()}
print(json.dumps(res, indent=1))
