"""Golden vectors of BASELINE.json's configurations at FULL size, produced by running the REFERENCE itself.

Run in the build container only (the reference is mounted read-only at /root/reference and does not exist on
the GPU box); takes several minutes of CPU time (single-threaded, so the reference's fp32 reduction order is
fixed):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_fullsize.py [cfg2 cfg4 cfg3 cfg5 ...]

Inputs are exactly SURVEY.md §8(d)'s synthetic cases (the same seeds bench.py and tests/_fullsize.py use), so the
GPU tests can rebuild them without the fixture.  Stored per case (tests/golden/fullsize_<case>.npz):

  * the reference's accepted / rejected (t0, dt) sequence — recorded through its own `callback_accept_step` /
    `callback_reject_step` hooks (torchdiffeq/_impl/misc.py:313-343) — and its evaluation count;
  * sample rows of the solution (rows 0..31, every 8th row — 12.5 % of the batch —, the last 32 rows) and max|y| over ALL rows, so that
    BASELINE.json's "max rel-err vs reference odeint" can be evaluated on the sample;
  * for the adjoint cases the same for dL/dy0 plus every parameter gradient in full.

Every array is an output of rtqichen/torchdiffeq v0.2.5 (or an input fed to it); nothing of its source is copied.
"""
import os
import sys
import time

import numpy as np
import torch

sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference")

import torchdiffeq  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
torch.set_num_threads(1)


def save(name, **arrays):
    out = {}
    for k, v in arrays.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    np.savez_compressed(os.path.join(HERE, name), **out)
    print(name, {k: v.shape for k, v in out.items()}, flush=True)


def sample_rows(n_rows):
    # every 8th row (12.5 % of the batch, r04; r03: every 64th) + both ends
    idx = set(range(min(32, n_rows))) | set(range(0, n_rows, 8)) | set(range(max(0, n_rows - 32), n_rows))
    return np.array(sorted(idx), dtype=np.int64)


class Recorded(torch.nn.Module):
    """func wrapper: evaluation count + the solver's step decisions, forward and (adjoint) backward."""

    def __init__(self, fn):
        super().__init__()
        self.fn = fn
        self.nfe = 0
        self.acc, self.rej, self.acc_adj, self.rej_adj = [], [], [], []

    def forward(self, t, y):
        self.nfe += 1
        return self.fn(t, y)

    def callback_accept_step(self, t0, y0, dt):
        self.acc.append((float(t0), float(dt)))

    def callback_reject_step(self, t0, y0, dt):
        self.rej.append((float(t0), float(dt)))

    def callback_accept_step_adjoint(self, t0, y0, dt):
        self.acc_adj.append((float(t0), float(dt)))

    def callback_reject_step_adjoint(self, t0, y0, dt):
        self.rej_adj.append((float(t0), float(dt)))


def steps(lst):
    return np.array(lst, dtype=np.float64).reshape(-1, 2)


def linear_problem(B, D, dtype):
    """SURVEY.md §8(d) cfg2 / cfg4 inputs."""
    g = torch.Generator().manual_seed(0)
    G = torch.randn(D, D, generator=g, dtype=torch.float64) / D ** 0.5
    A = (0.5 * (G - G.T) - 0.1 * torch.eye(D, dtype=torch.float64)).to(dtype)
    y0 = torch.randn(B, D, generator=g, dtype=torch.float64).to(dtype)
    return A, y0


def gen_linear(tag, B, D, dtype, method, rtol, atol, options=None):
    A, y0 = linear_problem(B, D, dtype)
    At = A.T.contiguous()
    f = Recorded(lambda t, y: y @ At)
    t = torch.tensor([0.0, 1.0], dtype=torch.float64 if dtype == torch.float64 else torch.float32)
    w = time.perf_counter()
    with torch.no_grad():
        y = torchdiffeq.odeint(f, y0, t, rtol=rtol, atol=atol, method=method, options=options)
    w = time.perf_counter() - w
    idx = sample_rows(B)
    exact = y0.double() @ torch.linalg.matrix_exp(A.double()).T
    save(f"fullsize_{tag}.npz", rows=idx, y_end_rows=y[-1][idx], y_end_absmax=y[-1].abs().max(),
         y0_rows=y0[:4], A_rows=A[:2], nfe=f.nfe, accepted=steps(f.acc), rejected=steps(f.rej),
         tol=np.array([rtol, atol]), wall_s_1thread=w,
         rel_err_vs_expm=float((y[-1].double() - exact).abs().max() / exact.abs().max()))


def gen_cfg3(tag="cfg3", rows=None, f64field=False, f64state=False):
    """manual_seed(0); Sequential(Linear(64,256),Tanh,Linear(256,256),Tanh,Linear(256,64));
    y0 = randn(65536, 64) from the global generator right after the layers; rtol 1e-5, atol 1e-7; loss y[-1]^2 sum."""
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(64, 256), torch.nn.Tanh(), torch.nn.Linear(256, 256), torch.nn.Tanh(),
                              torch.nn.Linear(256, 64))
    y0_all = torch.randn(65536, 64)
    y0 = (y0_all if rows is None else y0_all[rows].clone())
    if f64state:        # the same numbers carried in fp64 (state, parameters, times): no fp32 rounding anywhere
        net, y0 = net.double(), y0.double()
    y0 = y0.requires_grad_(True)
    if f64field:
        # the `*_f64field` companions: the SAME module on both sides (tests/_fullsize.F64MLPField — fp32 parameters and
        # states, evaluated in fp64), so that the backward solve's fp32 error estimate is not the field's own noise
        sys.path.insert(0, os.path.dirname(HERE))
        import _fullsize as fs
        f = Recorded(fs.F64MLPField(net))
    else:
        f = Recorded(lambda t, y: net(y))
    f.net = net          # parameters visible to odeint_adjoint
    t = torch.tensor([0.0, 1.0], dtype=y0.dtype)
    w0 = time.perf_counter()
    y = torchdiffeq.odeint_adjoint(f, y0, t, rtol=1e-5, atol=1e-7, method="dopri5")
    w1 = time.perf_counter()
    nfe_fwd, f.nfe = f.nfe, 0
    y[-1].pow(2).sum().backward()
    w2 = time.perf_counter()
    idx = sample_rows(y0.shape[0])
    arrays = dict(rows=idx, y0_rows=y0[:4], y_end_rows=y[-1][idx], y_end_absmax=y[-1].abs().max(),
                  grad_y0_rows=y0.grad[idx], grad_y0_absmax=y0.grad.abs().max(),
                  nfe_fwd=nfe_fwd, nfe_bwd=f.nfe, accepted=steps(f.acc), rejected=steps(f.rej),
                  accepted_adjoint=steps(f.acc_adj), rejected_adjoint=steps(f.rej_adj),
                  wall_s_1thread=np.array([w1 - w0, w2 - w1]))
    for i, p in enumerate(net.parameters()):
        arrays[f"p{i}"] = p
        arrays[f"grad_p{i}"] = p.grad
    save(f"fullsize_{tag}.npz", **arrays)


def gen_cfg5_f64field():
    """Companion of cfg5 with a noise-free field: tests/_fullsize.ExampleCNF(trace="closed", f64=True) — cfg5's flow
    with the same parameters, evaluated in fp64 — through the reference's odeint_adjoint; same state, times,
    tolerances and loss as cfg5."""
    sys.path.insert(0, os.path.dirname(HERE))
    import _fullsize as fs
    z = fs.load("cfg5")
    func = fs.ExampleCNF([z[f"p{i}"] for i in range(6)], trace="closed", f64=True)
    stats = {"acc": [], "rej": [], "acc_adj": [], "rej_adj": []}
    func.callback_accept_step = lambda t0, y0, dt: stats["acc"].append((float(t0), float(dt)))
    func.callback_reject_step = lambda t0, y0, dt: stats["rej"].append((float(t0), float(dt)))
    func.callback_accept_step_adjoint = lambda t0, y0, dt: stats["acc_adj"].append((float(t0), float(dt)))
    func.callback_reject_step_adjoint = lambda t0, y0, dt: stats["rej_adj"].append((float(t0), float(dt)))
    z0, logp0 = fs.cfg5_problem()
    z0 = z0.requires_grad_(True)
    t = torch.tensor([10.0, 0.0])
    z_t, logp_t = torchdiffeq.odeint_adjoint(func, (z0, logp0), t, atol=1e-5, rtol=1e-5, method="dopri5")
    nfe_fwd, func.nfe = func.nfe, 0
    loss = logp_t[-1].mean() - z_t[-1].pow(2).sum() / 100
    loss.backward()
    idx = sample_rows(z0.shape[0])
    arrays = dict(rows=idx, z_end_rows=z_t[-1][idx], logp_end_rows=logp_t[-1][idx], z_end_absmax=z_t[-1].abs().max(),
                  logp_end_absmax=logp_t[-1].abs().max(), grad_z0_rows=z0.grad[idx], grad_z0_absmax=z0.grad.abs().max(),
                  loss=loss, nfe_fwd=nfe_fwd, nfe_bwd=func.nfe, accepted=steps(stats["acc"]), rejected=steps(stats["rej"]),
                  accepted_adjoint=steps(stats["acc_adj"]), rejected_adjoint=steps(stats["rej_adj"]))
    for i, p in enumerate(func.parameters()):
        arrays[f"grad_p{i}"] = p.grad
    save("fullsize_cfg5_f64field.npz", **arrays)


def gen_cfg5():
    """cfg5: the CNF of examples/cnf.py:34-114 (exact trace by its per-dimension autograd loop) at random init
    (seed 0), state (z[32768,2], logp[32768,1]), t: 10 -> 0, dopri5 + adjoint, rtol = atol = 1e-5;
    loss = mean(logp(t1)) - sum(z(t1)^2)/100 (the reduced-size golden cnf.npz uses the same loss)."""
    import importlib.util
    argv, sys.argv = sys.argv, ["cnf.py"]
    try:
        spec = importlib.util.spec_from_file_location("ref_cnf_example", "/root/reference/examples/cnf.py")
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
    finally:
        sys.argv = argv
    torch.manual_seed(0)
    func = mod.CNF(in_out_dim=2, hidden_dim=32, width=64)
    stats = {"nfe": 0, "acc": [], "rej": [], "acc_adj": [], "rej_adj": []}
    inner_forward = func.forward

    def counted(t, states):
        stats["nfe"] += 1
        return inner_forward(t, states)
    func.forward = counted
    func.callback_accept_step = lambda t0, y0, dt: stats["acc"].append((float(t0), float(dt)))
    func.callback_reject_step = lambda t0, y0, dt: stats["rej"].append((float(t0), float(dt)))
    func.callback_accept_step_adjoint = lambda t0, y0, dt: stats["acc_adj"].append((float(t0), float(dt)))
    func.callback_reject_step_adjoint = lambda t0, y0, dt: stats["rej_adj"].append((float(t0), float(dt)))
    B = 32768
    g = torch.Generator().manual_seed(11)
    z0 = torch.randn(B, 2, generator=g, dtype=torch.float64).float().requires_grad_(True)
    logp0 = torch.zeros(B, 1)
    t = torch.tensor([10.0, 0.0])
    w0 = time.perf_counter()
    z_t, logp_t = torchdiffeq.odeint_adjoint(func, (z0, logp0), t, atol=1e-5, rtol=1e-5, method="dopri5")
    w1 = time.perf_counter()
    nfe_fwd, stats["nfe"] = stats["nfe"], 0
    loss = logp_t[-1].mean() - z_t[-1].pow(2).sum() / 100
    loss.backward()
    w2 = time.perf_counter()
    idx = sample_rows(B)
    arrays = dict(rows=idx, z0_rows=z0[:4], z_end_rows=z_t[-1][idx], logp_end_rows=logp_t[-1][idx],
                  z_end_absmax=z_t[-1].abs().max(), logp_end_absmax=logp_t[-1].abs().max(),
                  grad_z0_rows=z0.grad[idx], grad_z0_absmax=z0.grad.abs().max(), loss=loss,
                  nfe_fwd=nfe_fwd, nfe_bwd=stats["nfe"], accepted=steps(stats["acc"]), rejected=steps(stats["rej"]),
                  accepted_adjoint=steps(stats["acc_adj"]), rejected_adjoint=steps(stats["rej_adj"]),
                  wall_s_1thread=np.array([w1 - w0, w2 - w1]))
    for i, (name, p) in enumerate(func.named_parameters()):
        arrays[f"p{i}"] = p
        arrays[f"grad_p{i}"] = p.grad
    save("fullsize_cfg5.npz", **arrays)


CASES = {
    "cfg2": lambda: gen_linear("cfg2", 65536, 128, torch.float32, "dopri5", 1e-7, 1e-9),
    "cfg2_shard": lambda: gen_linear("cfg2_shard", 8192, 128, torch.float32, "dopri5", 1e-7, 1e-9),
    "cfg4": lambda: gen_linear("cfg4", 16384, 512, torch.float64, "dopri8", 1e-9, 1e-11),
    "cfg3": lambda: gen_cfg3(),
    "cfg3_shard": lambda: gen_cfg3("cfg3_shard", slice(0, 8192)),
    "cfg5": gen_cfg5,
    # exact companions of the noise-limited comparisons (r03): same solver, same inputs, noise removed at its source
    "cfg3_f64field": lambda: gen_cfg3("cfg3_f64field", f64field=True),
    "cfg3_shard_f64field": lambda: gen_cfg3("cfg3_shard_f64field", slice(0, 8192), f64field=True),
    "cfg3_shard_f64state": lambda: gen_cfg3("cfg3_shard_f64state", slice(0, 8192), f64state=True),
    "cfg4_first_step": lambda: gen_linear("cfg4_first_step", 16384, 512, torch.float64, "dopri8", 1e-9, 1e-11,
                                          options=dict(first_step=0.1)),
    "cfg5_f64field": gen_cfg5_f64field,
}

if __name__ == "__main__":
    for name in (sys.argv[1:] or list(CASES)):
        t0 = time.perf_counter()
        CASES[name]()
        print(f"{name}: {time.perf_counter() - t0:.1f} s", flush=True)
