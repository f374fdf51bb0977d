"""BASELINE.json's configurations at full size: the synthetic inputs of SURVEY.md §8(d), rebuilt from their seeds
(identical to tests/golden/make_golden_fullsize.py, which fed them to the reference), and helpers to compare a run of
the HIP path with the reference's results stored in tests/golden/fullsize_*.npz.  Used by the -m gpu tests and by
bench.py (`rel_err_vs_reference`)."""
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(case):
    with np.load(os.path.join(GOLDEN, f"fullsize_{case}.npz")) as z:
        return {k: z[k] for k in z.files}


def linear_problem(B, D, dtype):
    """cfg2 (65536 x 128 fp32) / cfg4 (16384 x 512 fp64): dy/dt = A y with A = skew - 0.1 I."""
    g = torch.Generator().manual_seed(0)
    G = torch.randn(D, D, generator=g, dtype=torch.float64) / D ** 0.5
    A = (0.5 * (G - G.T) - 0.1 * torch.eye(D, dtype=torch.float64)).to(dtype)
    y0 = torch.randn(B, D, generator=g, dtype=torch.float64).to(dtype)
    return A, y0


class MLPField(torch.nn.Module):
    """cfg3's vector field: Linear(64,256)-Tanh-Linear(256,256)-Tanh-Linear(256,64), time-independent."""

    def __init__(self, net):
        super().__init__()
        self.net = net
        self.nfe = 0
        self.counting = True     # False: no per-evaluation side effect — what hip_graph="auto" requires of a func it replays

    def forward(self, t, y):
        if self.counting:
            self.nfe += 1
        return self.net(y)


class F64MLPField(torch.nn.Module):
    """cfg3's MLP with fp32 parameters and fp32 states EVALUATED IN fp64 (inputs and weights cast up, the result
    rounded once): a field whose rounding noise is ~1e-9 of the fp32 one's.  Both the reference and this package are
    run on THIS module for the `*_f64field` fixtures, so the fp32 error estimate of the adjoint's backward solve is
    no longer the field's own noise and the two step sequences can be compared step for step."""

    def __init__(self, net):
        super().__init__()
        self.net = net
        self.nfe = 0

    def forward(self, t, y):
        self.nfe += 1
        h = y.double()
        for layer in self.net:
            if isinstance(layer, torch.nn.Linear):
                h = torch.nn.functional.linear(h, layer.weight.double(), layer.bias.double())
            else:
                h = layer(h)
        return h.to(y.dtype)


def cfg3_problem(rows=None):
    """manual_seed(0), default-initialised layers, then y0 = randn(65536, 64) from the same
    global CPU generator.  Returns (field, y0) on the CPU; `rows` selects a shard of the batch."""
    state = torch.random.get_rng_state()
    try:
        torch.manual_seed(0)
        net = torch.nn.Sequential(torch.nn.Linear(64, 256), torch.nn.Tanh(), torch.nn.Linear(256, 256),
                                  torch.nn.Tanh(), torch.nn.Linear(256, 64))
        y0 = torch.randn(65536, 64)
    finally:
        torch.random.set_rng_state(state)
    if rows is not None:
        y0 = y0[rows].clone()
    return MLPField(net), y0


def cfg5_problem():
    """cfg5's initial samples: z0 = randn(32768, 2) (generator seed 11), logp0 = 0; t: 10 -> 0."""
    g = torch.Generator().manual_seed(11)
    z0 = torch.randn(32768, 2, generator=g, dtype=torch.float64).float()
    return z0, torch.zeros(32768, 1)


class ExampleCNF(torch.nn.Module):
    """The flow of the reference's examples/cnf.py:34-114 restated: a hyper-network maps t to (W, B, U) and
    dz/dt = mean_k tanh(z w_k + b_k) u_k, dlogp/dt = -tr(d(dz/dt)/dz).  `trace` selects how the Jacobian trace is
    formed: "autograd" = the example's exact per-dimension autograd loop (cnf.py:66-74; a nested graph inside func),
    "closed" = the same quantity in closed form, "hutchinson" = the one-probe stochastic estimator e^T (df/dz) e
    with a fixed Rademacher probe (a benchmark-side variant; not in the reference's tree)."""

    def __init__(self, params, trace="autograd", width=64, dim=2, hidden=32, probe_seed=0, f64=False):
        super().__init__()
        self.f64 = f64          # closed-form trace only: evaluate in fp64, round the two outputs once (see F64MLPField)
        self.fc1 = torch.nn.Linear(1, hidden)
        self.fc2 = torch.nn.Linear(hidden, hidden)
        self.fc3 = torch.nn.Linear(hidden, 3 * width * dim + width)
        with torch.no_grad():
            for p, q in zip(self.parameters(), params):
                p.copy_(torch.as_tensor(q))
        self.width, self.dim, self.trace = width, dim, trace
        self.probe_seed, self._probe = probe_seed, None
        self.nfe = 0
        self.counting = True     # False: no per-evaluation side effect (see MLPField)

    def _hyper(self, t):
        width, dim, block = self.width, self.dim, self.width * self.dim
        if self.f64:
            lin = lambda layer, x: torch.nn.functional.linear(x, layer.weight.double(), layer.bias.double())
            p = torch.tanh(lin(self.fc1, t.double().reshape(1, 1)))
            p = lin(self.fc3, torch.tanh(lin(self.fc2, p))).reshape(-1)
        else:
            p = torch.tanh(self.fc1(t.reshape(1, 1)))
            p = self.fc3(torch.tanh(self.fc2(p))).reshape(-1)
        W = p[:block].reshape(width, dim)
        U = p[block:2 * block].reshape(width, dim) * torch.sigmoid(p[2 * block:3 * block].reshape(width, dim))
        return W, U, p[3 * block:]

    def forward(self, t, states):
        if self.counting:
            self.nfe += 1
        z = states[0]
        W, U, b = self._hyper(t)
        if self.trace == "closed":
            out_dtype = z.dtype
            if self.f64:
                z = z.double()
            h = torch.tanh(z @ W.T + b)
            dz = (h @ U) / self.width
            tr = ((1 - h * h) * (W * U).sum(-1)).sum(-1, keepdim=True) / self.width
            return dz.to(out_dtype), (-tr).to(out_dtype)
        with torch.enable_grad():
            if not z.requires_grad:          # forward solve (no-grad mode): a leaf sharing z's storage
                z = z.detach().requires_grad_(True)
            h = torch.tanh(z @ W.T + b)
            dz = (h @ U) / self.width
            if self.trace == "autograd":
                tr = 0.0
                for i in range(self.dim):
                    tr = tr + torch.autograd.grad(dz[:, i].sum(), z, create_graph=True)[0][:, i]
            else:
                if self._probe is None or self._probe.shape != z.shape or self._probe.device != z.device:
                    g = torch.Generator().manual_seed(self.probe_seed)
                    self._probe = (torch.randint(0, 2, z.shape, generator=g).to(z.dtype) * 2 - 1).to(z.device)
                e = self._probe
                tr = (torch.autograd.grad(dz, z, e, create_graph=True)[0] * e).sum(-1)
        return dz, -tr.reshape(-1, 1)


def sample_rel_err(value_rows, ref_rows, ref_absmax):
    """BASELINE.json's rel-err on the stored sample: max|y - y_ref| over the sample rows / max|y_ref| over ALL rows."""
    a = torch.as_tensor(value_rows).detach().double().cpu()
    b = torch.as_tensor(ref_rows).double().cpu()
    return float((a - b).abs().max() / float(ref_absmax))


def steps_match(mine, ref, rel=1e-2):
    """Accepted (or rejected) step sequences [(t0, dt)]: same count, every dt within `rel`."""
    mine, ref = np.asarray(mine, dtype=np.float64).reshape(-1, 2), np.asarray(ref, dtype=np.float64).reshape(-1, 2)
    if mine.shape != ref.shape:
        return False, f"{len(mine)} steps vs the reference's {len(ref)}"
    if len(ref) == 0:
        return True, ""
    dev = np.abs(mine[:, 1] - ref[:, 1]) / np.abs(ref[:, 1])
    return bool(dev.max() <= rel), f"max dt deviation {dev.max():.3e}"


class Recorder:
    """Attach the reference-style step callbacks (forward and adjoint) to a func and count its evaluations."""

    def __init__(self, func):
        self.acc, self.rej, self.acc_adj, self.rej_adj = [], [], [], []
        func.callback_accept_step = lambda t0, y0, dt: self.acc.append((float(t0), float(dt)))
        func.callback_reject_step = lambda t0, y0, dt: self.rej.append((float(t0), float(dt)))
        func.callback_accept_step_adjoint = lambda t0, y0, dt: self.acc_adj.append((float(t0), float(dt)))
        func.callback_reject_step_adjoint = lambda t0, y0, dt: self.rej_adj.append((float(t0), float(dt)))
        self.func = func

    def detach(self):
        for name in ("callback_accept_step", "callback_reject_step", "callback_accept_step_adjoint",
                     "callback_reject_step_adjoint"):
            if name in self.func.__dict__:
                delattr(self.func, name)
