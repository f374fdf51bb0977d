"""Implicit fixed-grid Runge–Kutta methods of the reference's SOLVERS table (implicit_euler, implicit_midpoint,
trapezoid, radauIIA3, gl4, radauIIA5, gl6, sdirk2, trbdf2; rk_common.py:378-558, fixed_grid_implicit.py) and the
SciPy bridge, against the reference's own outputs (tests/golden/implicit.npz, written by make_golden.gen_implicit).

The reference runs Broyden's method on a DENSE (stages·N)² Jacobian; the product runs the same iteration matrix-free
(torchdiffeq_amd/implicit.py).  Same iterates up to rounding, same stopping test ⇒ same evaluation counts and
solutions equal to rounding level: the tolerances below are 1e-12 (fp64) and 1e-6 (fp32) relative — far inside the
nonlinear solver's own stopping tolerance (1e-8 / 1e-6 on the residual).  `dev`: "cuda" = HIP kernels on the MI355X,
"cpu" = host logic with the oracle substituted for the kernels."""
import warnings

import numpy as np
import pytest
import torch

import torchdiffeq_amd as tda
from torchdiffeq_amd.tableaus import IMPLICIT_TABLEAUS
from _cases import T, load, rel_err

METHODS = ["implicit_euler", "implicit_midpoint", "trapezoid", "radauIIA3", "gl4", "radauIIA5", "gl6", "sdirk2",
           "trbdf2"]


def _field(t, y):
    """Exactly rounded elementwise operations only: the same derivative bits on every CPU and on the GPU."""
    return (1 - t * 0.5) * (y.roll(1, -1) * 0.3 - y * 2.0) - y * y * y * 0.01


class _Count:
    def __init__(self, fn):
        self.fn, self.nfe = fn, 0

    def __call__(self, t, y):
        self.nfe += 1
        return self.fn(t, y)


def test_solver_table_equals_the_reference_table():
    """Every name of the reference's SOLVERS (odeint.py:19-46), in its order."""
    z = load("implicit.npz")
    assert list(tda.SOLVERS) == [str(s) for s in z["solver_names"]]
    assert tda.SOLVERS["fixed_adams"] is tda.SOLVERS["implicit_adams"]


@pytest.mark.parametrize("method", METHODS)
def test_tableaus_bit_identical(method):
    z = load("implicit.npz")
    tab = IMPLICIT_TABLEAUS[method]
    assert np.array_equal(np.array(tab.alpha), z[f"{method}_alpha"])
    assert np.array_equal(np.array([b for row in tab.beta for b in row]), z[f"{method}_beta_flat"])
    assert np.array_equal(np.array(tab.c_sol), z[f"{method}_c_sol"])
    assert tda.SOLVERS[method].order == int(z[f"{method}_order"]) == tab.order


@pytest.mark.parametrize("method", METHODS)
def test_solves_match_the_reference(dev, method):
    z = load("implicit.npz")
    y0, t = T(z["y0"], dev), T(z["t"], dev)
    cases = {
        "grid": (y0, torch.linspace(0, 1, 11, device="cpu").to(dev), {}),
        "step": (y0, t, dict(step_size=0.05)),
        "perturb": (y0, t, dict(step_size=0.05, perturb=True)),
        "cubic": (y0, t, dict(step_size=0.05, interp="cubic")),
        "rev": (y0, torch.tensor([1.0, 0.45, 0.0]), dict(step_size=0.05)),
        "iters2": (y0, t, dict(step_size=0.1, max_iters=2)),
        "f64": (y0.double(), t.double(), dict(step_size=0.05)),
    }
    for tag, (y, tt, opts) in cases.items():
        c = _Count(_field)
        with warnings.catch_warnings(record=True) as w, torch.no_grad():
            warnings.simplefilter("always")
            got = tda.odeint(c, y, tt, method=method, options=opts)
        ref = T(z[f"{method}_{tag}"], dev)
        assert got.shape == ref.shape and got.dtype == ref.dtype
        assert rel_err(got, ref) < (1e-12 if tag == "f64" else 1e-6), tag
        # a residual norm within rounding of the stopping tolerance may cost one more / one fewer iteration
        # (each costs one evaluation per stage); measured: identical counts for every case
        assert abs(c.nfe - int(z[f"{method}_{tag}_nfe"])) <= 2 * len(IMPLICIT_TABLEAUS[method].alpha), tag
        n_warn = sum("did not converge" in str(x.message) for x in w)      # (a ResourceWarning of another test may land here)
        assert n_warn == int(z[f"{method}_{tag}_warnings"]), tag


@pytest.mark.parametrize("method", METHODS)
def test_tuple_state_and_event(dev, method):
    z = load("implicit.npz")
    y0 = T(z["y0"], dev).double()
    ft = lambda t, y: (_field(t, y[0]), -y[1] * y[0][0, :3] * (1 + t))
    yt = (y0, torch.tensor([0.5, 0.25, 1.0], dtype=torch.float64))
    with torch.no_grad():
        out = tda.odeint(ft, yt, torch.linspace(0, 1, 11, dtype=torch.float64, device="cpu").to(dev), method=method)
    for i in range(2):
        assert rel_err(out[i], z[f"{method}_tuple{i}"]) < 1e-12
    fe = lambda t, y: torch.stack([y[1], -y[0]])
    et, ys = tda.odeint_event(fe, torch.tensor([1.0, 0.0], dtype=torch.float64), torch.tensor(0.0, dtype=torch.float64),
                              event_fn=lambda t, y: y[0], method=method, options=dict(step_size=0.05), atol=1e-8)
    assert abs(float(et) - float(z[f"{method}_event_t"])) < 1e-10
    assert rel_err(ys, z[f"{method}_event_y"]) < 1e-10


def test_stiff_problem_where_explicit_euler_blows_up(dev):
    """y' = -200 (y - cos t): explicit euler at h = 0.05 has amplification |1 - 10| per step; the implicit methods
    follow the slow manifold."""
    f = lambda t, y: -200.0 * (y - torch.cos(t))
    y0 = torch.zeros(64, dtype=torch.float64)
    t = torch.linspace(0, 1, 21, dtype=torch.float64)
    with torch.no_grad():
        bad = tda.odeint(f, y0, t, method="euler")
        assert float(bad[-1].abs().max()) > 1e10
        for method, tol in [("implicit_euler", 2e-2), ("trapezoid", 2e-2), ("radauIIA5", 1e-4), ("sdirk2", 2e-2)]:
            y = tda.odeint(f, y0, t, method=method)
            exact = (200.0 ** 2 * torch.cos(t[-1]) + 200.0 * torch.sin(t[-1])) / (1 + 200.0 ** 2)   # slow manifold
            assert float((y[-1] - exact).abs().max()) < tol, method


@pytest.mark.parametrize("method", ["implicit_midpoint", "radauIIA5", "trbdf2"])
def test_batch_scale_state_the_dense_jacobian_could_not_hold(dev, method):
    """2^17 unknowns per stage: the reference's dense Broyden matrix would need (stages·N)² entries (> 1 TB for
    radauIIA5); the matrix-free iteration needs a few vectors.  Closed form of y' = -a y per row."""
    n = 1 << 17
    a = torch.linspace(0.5, 3.0, n, dtype=torch.float64)
    y0 = torch.ones(n, dtype=torch.float64)
    t = torch.linspace(0, 1, 11, dtype=torch.float64)
    with torch.no_grad():
        y = tda.odeint(lambda t_, y_: -a * y_, y0, t, method=method)
    tol = {"implicit_midpoint": 2e-3, "radauIIA5": 1e-7, "trbdf2": 2e-3}[method]
    assert rel_err(y[-1], torch.exp(-a)) < tol


def test_rms_residual_test_for_large_fp32_states(dev):
    """Opt-in `residual_norm="rms"`: at 2^18 fp32 unknowns per stage the reference's absolute 2-norm bound of 1e-6 sits
    below the rounding floor of the residual, so every step would run all max_iters iterations; the RMS form of the
    same bound converges in a few.  Same solution to fp32 rounding level."""
    n = 1 << 18
    a = torch.linspace(0.5, 3.0, n)
    y0 = torch.ones(n)
    t = torch.linspace(0, 1, 5)
    calls = {"rms": 0, "l2": 0}

    def make(tag):
        def f(t_, y_):
            calls[tag] += 1
            return -a * y_
        return f
    with torch.no_grad(), warnings.catch_warnings():
        warnings.simplefilter("ignore")
        y_rms = tda.odeint(make("rms"), y0, t, method="radauIIA3", options=dict(residual_norm="rms"))
        y_l2 = tda.odeint(make("l2"), y0, t, method="radauIIA3", options=dict(max_iters=12))
    assert calls["rms"] < calls["l2"]            # l2 runs into its iteration cap at every step
    assert rel_err(y_rms, y_l2) < 1e-5 and rel_err(y_rms[-1], torch.exp(-a)) < 5e-3       # 4 steps of a 3rd-order method
    with pytest.raises(ValueError, match="residual_norm"):
        tda.odeint(lambda t_, y_: -y_, torch.ones(3), t, method="gl4", options=dict(residual_norm="max"))


def test_adjoint_with_implicit_methods(dev):
    """odeint_adjoint only needs no-grad solves, so the implicit methods serve as forward and backward method."""
    g = torch.Generator(device="cpu").manual_seed(7)
    lin = torch.nn.Linear(3, 3).double().to(dev)
    with torch.no_grad():
        lin.weight.copy_(torch.randn(3, 3, generator=g, dtype=torch.float64, device="cpu") * 0.5)
        lin.bias.copy_(torch.randn(3, generator=g, dtype=torch.float64, device="cpu") * 0.5)
    params = tuple(lin.parameters())
    f = lambda t_, y_: torch.tanh(lin(y_))
    y0 = torch.randn(4, 3, generator=g, dtype=torch.float64, device="cpu").to(dev).requires_grad_(True)
    t = torch.linspace(0, 1, 21, dtype=torch.float64)
    grads = {}
    for method in ["dopri5", "gl4", "sdirk2"]:
        y0.grad = None
        y = tda.odeint_adjoint(f, y0, t, method=method, rtol=1e-9, atol=1e-11, adjoint_params=params)
        y[-1].pow(2).sum().backward()
        grads[method] = y0.grad.clone()
    assert rel_err(grads["gl4"], grads["dopri5"]) < 1e-6
    assert rel_err(grads["sdirk2"], grads["dopri5"]) < 2e-3


@pytest.mark.parametrize("method", METHODS)
def test_backprop_through_the_solver(dev, method):
    """Plain odeint in grad mode: the product attaches the converged stage values to the graph by the implicit
    function theorem (exact gradient of the converged solution); the reference differentiates its unrolled Broyden
    iterations, which stop at a residual of 1e-8 — so the two agree to about that level (measured 1e-10 .. 2e-8)."""
    z = load("implicit.npz")
    lin = torch.nn.Linear(3, 3).double().to(dev)
    with torch.no_grad():
        lin.weight.copy_(T(z[f"{method}_bp_w"], dev))
        lin.bias.copy_(T(z[f"{method}_bp_b"], dev))
    y0 = T(z[f"{method}_bp_y0"], dev).requires_grad_(True)
    t = torch.linspace(0, 1, 6, dtype=torch.float64).requires_grad_(True)
    y = tda.odeint(lambda t_, y_: torch.tanh(lin(y_)) * torch.cos(t_), y0, t, method=method)
    assert rel_err(y.detach(), z[f"{method}_bp_y"]) < 1e-12
    loss = y[-1].pow(2).sum() + y[3].sum()
    g = torch.autograd.grad(loss, [y0, t, lin.weight, lin.bias])
    for name, v in zip(["gy0", "gt", "gw", "gb"], g):
        assert rel_err(v, z[f"{method}_bp_{name}"]) < 2e-7, name


@pytest.mark.parametrize("method", ["implicit_midpoint", "trapezoid", "radauIIA3", "trbdf2"])
def test_gradcheck_nonlinear_field(dev, method):
    """The implicit-function gradient against finite differences of the solve itself, on a field that depends on y
    (the reference's gradcheck, gradient_tests.py:13-23, uses the constant field only)."""
    a = torch.tensor([[-0.3, 0.7], [-0.7, -0.2]], dtype=torch.float64)
    f = lambda t_, y_: torch.tanh(y_ @ a) * (1 + 0.3 * t_)
    y0 = torch.tensor([[0.5, -0.4], [0.2, 0.9]], dtype=torch.float64, requires_grad=True)
    t = torch.tensor([0.0, 0.3, 0.55], dtype=torch.float64, requires_grad=True)
    assert torch.autograd.gradcheck(lambda y_, t_: tda.odeint(f, y_, t_, method=method), (y0, t), atol=1e-5, rtol=1e-4)


@pytest.mark.parametrize("solver", ["LSODA", "RK45"])
def test_scipy_solver(dev, solver):
    """The SciPy bridge: analytic check (the reference's own test, odeint_tests.py:83-96) for a tensor and a tuple state,
    decreasing time, and the no-gradient contract."""
    a = torch.tensor([0.5, 1.0, 2.0], dtype=torch.float64)
    f = lambda t_, y_: -a * y_
    y0 = torch.ones(3, dtype=torch.float64)
    t = torch.linspace(0, 1, 6, dtype=torch.float64)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        y = tda.odeint(f, y0, t, method="scipy_solver", options=dict(solver=solver), rtol=1e-9, atol=1e-11)
        assert y.shape == (6, 3) and y.dtype == torch.float64 and y.device.type == torch.device(dev).type
        assert rel_err(y, torch.exp(-a * t[:, None])) < 1e-6
        yr = tda.odeint(f, y0, -t, method="scipy_solver", options=dict(solver=solver), rtol=1e-9, atol=1e-11)
        assert rel_err(yr, torch.exp(a * t[:, None])) < 1e-6
        yt = tda.odeint(lambda t_, y_: (-a * y_[0], y_[1] * 0.5), (y0, torch.ones(2, 2, dtype=torch.float64)), t, method="scipy_solver",
                        options=dict(solver=solver), rtol=1e-6, atol=1e-8)
        assert yt[1].shape == (6, 2, 2) and yt[1].dtype == torch.float64
        assert rel_err(yt[1][-1], torch.full((2, 2), float(np.exp(0.5)))) < 1e-4
        assert tda.odeint(f, y0, t[:1], method="scipy_solver").shape == (1, 3)
