"""cProfile of an adjoint backward pass over MANY output intervals (200 time points, small state): the per-interval
host overhead of the backward solve (solver construction, norm plan, initial step selection)."""
import cProfile
import os
import pstats
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torchdiffeq_amd as tda  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
net = torch.nn.Sequential(torch.nn.Linear(16, 64), torch.nn.Tanh(), torch.nn.Linear(64, 16)).to(dev)


class F(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.net = net

    def forward(self, t, y):
        return self.net(y)


f = F()
y0 = torch.randn(64, 16, device=dev)
t = torch.linspace(0, 1, 200, device=dev)


def run():
    for p in f.parameters():
        p.grad = None
    x = y0.clone().requires_grad_(True)
    y = tda.odeint_adjoint(f, x, t, rtol=1e-4, atol=1e-6, method="dopri5")
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    y.pow(2).sum().backward()
    torch.cuda.synchronize()
    return time.perf_counter() - t0


run()
print("backward s:", min(run() for _ in range(3)))
pr = cProfile.Profile()
# the backward solve runs in the autograd engine's own thread: profile it from inside
from torchdiffeq_amd.adjoint import OdeintAdjointMethod  # noqa: E402
_orig = OdeintAdjointMethod.backward


def _profiled(ctx, *g):
    pr.enable()
    try:
        return _orig(ctx, *g)
    finally:
        pr.disable()


OdeintAdjointMethod.backward = staticmethod(_profiled)
run()
pstats.Stats(pr).sort_stats("tottime").print_stats(30)
