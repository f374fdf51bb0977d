"""Round-3 advisor findings, pinned."""
import warnings

import pytest
import torch

import torchdiffeq_amd as tda


# -- 1. fixed-grid graph mode must not swallow parameter gradients ------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("method", ["rk4", "euler", "midpoint", "heun3"])
def test_fixed_grid_graph_mode_keeps_parameter_gradients(method):
    """Plain `odeint` training (grad mode on, only func's parameters require grad) under hip_graph=True: the replayed
    kernels write raw buffers, so the solve has to take the eager path (decided from what func holds) and the gradients
    must equal the eager ones; under no_grad the same call is captured."""
    torch.manual_seed(0)
    lin = torch.nn.Linear(6, 6).cuda()
    y0 = torch.randn(8, 6, device="cuda")
    t = torch.linspace(0.0, 1.0, 9, device="cuda")
    grads = {}
    for graph in (False, True):
        lin.zero_grad()
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            y = tda.odeint(lambda t_, y_: torch.tanh(lin(y_)), y0, t, method=method, options=dict(hip_graph=graph))
        assert y.requires_grad
        y[-1].pow(2).sum().backward()
        grads[graph] = [p.grad.clone() for p in lin.parameters()]
    for a, b in zip(grads[False], grads[True]):
        assert torch.equal(a, b)
    with torch.no_grad():
        y_eager = tda.odeint(lambda t_, y_: torch.tanh(lin(y_)), y0, t, method=method)
        y_graph = tda.odeint(lambda t_, y_: torch.tanh(lin(y_)), y0, t, method=method, options=dict(hip_graph=True))
    assert torch.equal(y_eager, y_graph)


def test_fixed_grid_graph_capability_is_decided_without_evaluating_func(cpu_backend, monkeypatch):
    """Advisor r04: whether a grad-mode solve may replay captured steps is decided from what func HOLDS (module
    parameters, closure cells, ...), not by a probe evaluation — func is not called (no RNG / counter side effects, no
    extra evaluation), and a func whose parameter dependence is gated by time is still found out."""
    from torchdiffeq_amd import solvers
    from torchdiffeq_amd.misc import OdeFunc, StateLayout
    lin = torch.nn.Linear(3, 3).double()
    frozen = torch.nn.Linear(3, 3).double().requires_grad_(False)
    w = torch.ones(3, dtype=torch.float64, requires_grad=True)
    y0 = torch.ones(3, dtype=torch.float64)
    t = torch.linspace(0.0, 1.0, 4, dtype=torch.float64)
    calls = [0]

    def counted(f):
        def g(t_, y_):
            calls[0] += 1
            return f(t_, y_)
        return g
    cases = [(counted(lambda t_, y_: lin(y_)), False),                                  # a module in a closure cell
             (counted(lambda t_, y_: lin(y_) if float(t_) > 0.5 else -y_), False),      # gated by time: t[0] would not show it
             (counted(lambda t_, y_: y_ * w), False),                                   # a leaf tensor in a closure cell
             (counted(lambda t_, y_: frozen(y_)), True),                                # nothing requires grad: replay is fine
             (counted(lambda t_, y_: -y_), True)]
    for f, capable in cases:
        func = OdeFunc(f, StateLayout([y0.shape], False), 1.0, y0.dtype, y0.device)
        s = solvers.RK4(func=func, y0=y0, atol=1e-9, hip_graph=True)
        monkeypatch.setattr(s, "device", torch.device("cuda"))               # the capability question only
        monkeypatch.setattr(s.kernels, "grid_advance_stages", lambda *a, **k: None, raising=False)
        assert s._graph_capable(t, t) is capable
        with torch.no_grad():
            assert s._graph_capable(t, t) is True
    assert calls[0] == 0


# -- 2. the "running the eager path" warning of fixed-grid solvers: only for an explicit option ---------------------------
def test_fixed_grid_eager_warning_only_for_the_explicit_option(cpu_backend, monkeypatch):
    y0 = torch.ones(3, dtype=torch.float64)
    t = torch.linspace(0.0, 1.0, 4, dtype=torch.float64)
    f = lambda t_, y_: -y_
    monkeypatch.setenv("TDEQ_HIP_GRAPH", "1")
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        tda.odeint(f, y0, t, method="rk4")                       # process-wide default, not capable (CPU): silent
    monkeypatch.delenv("TDEQ_HIP_GRAPH")
    with pytest.warns(UserWarning, match="running the eager path"):
        tda.odeint(f, y0, t, method="rk4", options=dict(hip_graph=True))


# -- 3. proxy check of the captured backward solve -----------------------------------------------------------------------
def test_proxy_check_uses_solver_time_and_a_nonzero_cotangent(cpu_backend):
    """`proxy_is_faithful` gets the forward solve's SOLVER time (for a reversed `t` the user's func must be asked at its
    own +t, not at -t) and probes with a fixed non-zero cotangent, so a loss that ignores the last output (adj_y = 0)
    still compares real VJPs."""
    from torchdiffeq_amd.adjoint import _AugmentedDynamics, _PROXY_CHECKED
    from torchdiffeq_amd.misc import OdeFunc, StateLayout
    seen = []

    class Field(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.w = torch.nn.Parameter(torch.tensor([0.5, -0.25], dtype=torch.float64))
            self.alias = [self.w]                     # reached NOT by attribute lookup: functional_call cannot re-route it

        def forward(self, t, y):
            seen.append(float(t))
            return y * self.alias[0] * torch.sqrt(t)          # undefined (NaN) for t < 0
    f = Field()
    y0 = torch.ones(2, dtype=torch.float64)
    fwd = OdeFunc(f, StateLayout([y0.shape], False), -1.0, y0.dtype, y0.device)      # reversed-time forward solve
    aug_layout = StateLayout([torch.Size(())] + [y0.shape] * 2 + [f.w.shape], True, chunk=fwd.layout.chunk)
    aug = torch.zeros(aug_layout.total, dtype=torch.float64)      # adj_y == 0: the vacuous case
    aug_layout.unpack(aug)[1].copy_(y0)
    dyn = _AugmentedDynamics(fwd, aug_layout, (f.w,), False)
    _PROXY_CHECKED.pop(f, None)
    t_solver = torch.tensor(-2.0, dtype=torch.float64)           # solver time of user time +2
    assert dyn.proxy_is_faithful(t_solver, aug) is False         # the alias is found out even though adj_y is zero
    assert seen and all(s == 2.0 for s in seen)                  # func was asked at +2, where it is defined
    assert float(aug_layout.unpack(aug)[2].abs().sum()) == 0.0   # the caller's state is not touched by the probe


# -- 4. the one known residue of the 0-dim promotion emulation, bounded ---------------------------------------------------
def test_zero_dim_fp32_state_on_an_fp64_grid_residue_is_bounded(dev):
    """A 0-dim fp32 state on an fp64 time grid under dopri5 differs from the reference by ~1.3e-6 relative (the
    reference's 0-dim x 0-dim promotions inside the adaptive step are not emulated — docs/LAB_NOTEBOOK.md §8).  Bounded here so
    that a regression is visible; rk4 / Adams on the same inputs are bit-identical (tests/test_dropin_golden.py)."""
    from _cases import load, rel_err
    ref = load("brow.npz")["zero_dim_f32_on_f64_grid_dopri5"]
    with torch.no_grad():
        y = tda.odeint(lambda t_, y_: -y_ * torch.cos(t_), torch.tensor(1.5, device=dev),
                       torch.linspace(0.0, 2.0, 5, dtype=torch.float64, device=dev), method="dopri5")
    assert y.dtype == torch.float32 and y.shape == (5,)
    assert rel_err(y, ref) < 5e-6
