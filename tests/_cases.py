"""Shared problem builders for the golden-vector tests (inputs and reference outputs come from
tests/golden/*.npz, produced by tests/golden/make_golden.py running the reference itself)."""
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return np.load(os.path.join(GOLDEN, name))


def T(a, device="cpu"):
    return torch.from_numpy(np.asarray(a)).to(device)


def rel_err(a, b):
    a = torch.as_tensor(a).double().cpu()
    b = torch.as_tensor(b).double().cpu()
    return float((a - b).abs().max() / b.abs().max())


class StatFunc(torch.nn.Module):
    """Wraps f(t, y) counting evaluations and accepted / rejected steps through the public callbacks."""

    def __init__(self, fn):
        super().__init__()
        self.fn = fn
        self.nfe = 0
        self.accept, self.reject, self.steps = [], [], []

    def forward(self, t, y):
        self.nfe += 1
        return self.fn(t, y)

    def callback_step(self, t0, y0, dt):
        self.steps.append(float(dt))

    def callback_accept_step(self, t0, y0, dt):
        self.accept.append(float(dt))

    def callback_reject_step(self, t0, y0, dt):
        self.reject.append(float(dt))


def make_mlp(z, tag, device="cpu"):
    ps = [T(z[f"adj_{tag}_p{i}"]) for i in range(6)]
    d, h = ps[0].shape[1], ps[0].shape[0]
    net = torch.nn.Sequential(torch.nn.Linear(d, h), torch.nn.Tanh(), torch.nn.Linear(h, h), torch.nn.Tanh(),
                              torch.nn.Linear(h, d)).to(ps[0].dtype)
    with torch.no_grad():
        for p, q in zip(net.parameters(), ps):
            p.copy_(q)

    class F(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.net = net

        def forward(self, t, y):
            return self.net(y)

    return F().to(device)


# (golden key prefix, method, solution tolerance vs the reference fp32/fp64 result)
# fp32 at rtol=1e-7 sits on the rounding-noise floor: the reference differs from ITSELF by 3.9e-6 when
# only its CPU thread count changes, so 1e-5 is the meaningful bound there.
SOLVE_CASES = [
    ("cfg2_tight", "dopri5", 1e-5),
    ("cfg2_loose", "dopri5", 2e-6),
    ("cfg2_rev", "dopri5", 2e-6),
]


def linear_case(z, prefix, device="cpu"):
    base = prefix.split("_")[0]
    A, y0 = T(z[f"{base}_A"], device), T(z[f"{base}_y0"], device)
    t = T(z[f"{prefix}_t"], device)
    rtol, atol = [float(v) for v in z[f"{prefix}_tol"]]
    return A, y0, t, rtol, atol
