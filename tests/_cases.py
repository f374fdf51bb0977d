"""Shared problem builders for the golden-vector tests (inputs and reference outputs come from
tests/golden/*.npz, produced by tests/golden/make_golden.py running the reference itself)."""
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    """The arrays of a golden file as a dict (read eagerly: no file handle is left for the GC to warn about)."""
    with np.load(os.path.join(GOLDEN, name)) as z:
        return {k: z[k] for k in z.files}


def T(a, device="cpu"):
    return torch.from_numpy(np.asarray(a)).to(device)


def rel_err(a, b):
    a = torch.as_tensor(a).detach().double().cpu()
    b = torch.as_tensor(b).detach().double().cpu()
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


class PlanarCNF(torch.nn.Module):
    """The continuous normalizing flow of the reference's examples/cnf.py:34-114 (cfg5), restated:
    a hyper-network maps t to (W, B, U); dz/dt = mean_k tanh(z.w_k + b_k) u_k and
    dlogp/dt = -tr(d(dz/dt)/dz).  The trace is written in closed form, tr = mean_k (1-h_k^2)(w_k.u_k),
    instead of the example's per-dimension autograd loop — same function, differentiable, no nested
    autograd.  Parameters are loaded from the golden file (the reference's seeded init)."""

    def __init__(self, z, device="cpu"):
        super().__init__()
        self.fc1 = torch.nn.Linear(1, 32)
        self.fc2 = torch.nn.Linear(32, 32)
        self.fc3 = torch.nn.Linear(32, 3 * 64 * 2 + 64)
        with torch.no_grad():
            for i, p in enumerate(self.parameters()):
                p.copy_(T(z[f"cnf_p{i}"]))
        self.width, self.dim = 64, 2
        self.to(device)

    def forward(self, t, states):
        z, _ = states
        width, dim, block = self.width, self.dim, self.width * self.dim
        p = torch.tanh(self.fc1(t.reshape(1, 1)))
        p = torch.tanh(self.fc2(p))
        p = self.fc3(p).reshape(-1)
        W = p[:block].reshape(width, dim)
        U = p[block:2 * block].reshape(width, dim) * torch.sigmoid(p[2 * block:3 * block].reshape(width, dim))
        Bv = p[3 * block:]
        h = torch.tanh(z @ W.T + Bv)                         # [batch, width]
        dz = (h @ U) / width
        trace = ((1 - h * h) * (W * U).sum(-1)).sum(-1, keepdim=True) / width
        return dz, -trace


# func outputs of the wrong shape (tests/golden/brow.npz `shape_*`: what the reference accepts, per method)
FUNC_SHAPE_CASES = {
    # name: (state shape(s), which slice / view of -y func returns)
    "vec2_first1": ((2,), lambda y: -y[:1]),
    "vec2_0dim": ((2,), lambda y: -y[0]),
    "mat32_row2": ((3, 2), lambda y: -y[0]),
    "mat32_1x2": ((3, 2), lambda y: -y[:1]),
    "mat32_col31": ((3, 2), lambda y: -y[:, :1]),
    "mat32_flat6": ((3, 2), lambda y: -y.reshape(-1)),
    "mat32_T23": ((3, 2), lambda y: -y.T),
    "mat32_lead1": ((3, 2), lambda y: -y[None]),
    "mat32_lead2": ((3, 2), lambda y: -torch.stack([y, y])),
    "zero_dim_to_1": ((), lambda y: -y.reshape(1)),
    "one_to_zero_dim": ((1,), lambda y: -y[0]),
    "tuple_short": (((2,), (3,)), lambda y: (-y[0][:1], -y[1])),
    "tuple_reshaped": (((2, 3), (3,)), lambda y: (-y[0].T, -y[1].reshape(3, 1))),
}
FUNC_SHAPE_METHODS = ("dopri5", "dopri8", "bosh3", "adaptive_heun", "rk4", "euler", "midpoint")


# r06: tuple tolerances whose vector ENTRIES are not tensors — `torch.as_tensor(entry).expand(numel)` in the reference
# (torchdiffeq/_impl/misc.py:113-123) takes lists, tuples and numpy arrays alike.  Inputs of tests/golden/tuple_tol.npz
# (`ttl_*`, generated by make_golden.py from the imported reference) and of the tests that replay them.
TUPLE_TOL_LIST_FORMS = {
    "rtol_list": (([1e-5, 1e-4, 1e-6], 1e-4), (1e-8, 1e-8)),
    "atol_list": ((1e-5, 1e-4), ([1e-8, 1e-7, 1e-9], 1e-8)),
    "both": (([1e-5, 1e-4, 1e-6], [1e-4, 1e-5]), ([1e-8, 1e-7, 1e-9], 1e-8)),
    "numpy_and_tuple": ((np.array([1e-5, 1e-4, 1e-6]), 1e-4), (1e-8, (1e-8, 1e-7))),
}


# r06: small-state dopri8 cases of tests/golden/dopri8_small.npz — name: (field kind, rows, dim, seed, end time)
DOPRI8_SMALL_CASES = {
    "tanh": ("tanh", 4, 3, 41, 1.3),
    "linear_t": ("linear_t", 1, 5, 43, 0.9),
    "cubic": ("cubic", 9, 2, 47, 1.7),
}


def dopri8_small_field(kind, W):
    def f(t, y):
        if kind == "tanh":
            return torch.tanh(y @ W.T) * torch.cos(t)
        if kind == "linear_t":
            return y @ W.T * (1 + 0.3 * t) - 0.2 * y
        return -0.3 * y ** 3 + torch.sin(3 * t) * (y @ W.T)
    return f
