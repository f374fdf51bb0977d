"""SURVEY.md §8(f) rank 2 — the remaining explicit RK methods of the reference's SOLVERS table
(tsit5, bosh3, fehlberg2, adaptive_heun; euler, midpoint, heun2, heun3) and the fixed-grid options
(`step_size`, `perturb`, `interp='cubic'`) against the reference's own outputs (tests/golden/methods.npz).

Runs through the `dev` fixture like test_parity_golden.py: "cuda" = the product on the MI355X,
"cpu" = the product's host logic with the oracle substituted for the HIP kernels (test-only)."""
import numpy as np
import pytest
import torch

import torchdiffeq_amd as tda
from _cases import StatFunc, T, load, rel_err

ADAPTIVE = ["tsit5", "bosh3", "fehlberg2", "adaptive_heun"]
FIXED = ["euler", "midpoint", "heun2", "heun3", "rk4"]


def test_solver_table_matches_reference_explicit_methods():
    """Same keys, same order as the reference's table for the explicit RK family (odeint.py:19-30)."""
    assert list(tda.SOLVERS)[:11] == ["dopri8", "dopri5", "tsit5", "bosh3", "fehlberg2", "adaptive_heun",
                                      "euler", "midpoint", "heun2", "heun3", "rk4"]
    assert [tda.SOLVERS[m].order for m in ADAPTIVE] == [5, 3, 2, 2]
    assert [tda.SOLVERS[m].order for m in FIXED] == [1, 2, 2, 3, 4]


@pytest.mark.parametrize("method", ADAPTIVE)
def test_adaptive_pairs_fp64_step_sequence(dev, method):
    """fp64, well above the rounding floor: the accept/reject sequence and every dt must match the reference."""
    z = load("methods.npz")
    A, y0, t = T(z["ad_A"], dev), T(z["ad_y0"], dev), T(z["ad_t"], dev)
    rtol, atol = [float(v) for v in z[f"ad_{method}_tol"]]
    f = StatFunc(lambda t_, y_: torch.sin(2 * t_) * (y_ @ A.T) * 2 - 0.5 * y_ ** 3)
    with torch.no_grad():
        y = tda.odeint(f, y0, t, rtol=rtol, atol=atol, method=method)
    assert rel_err(y, z[f"ad_{method}_y"]) < 1e-10
    assert f.nfe == int(z[f"ad_{method}_nfe"])
    assert (len(f.accept), len(f.reject)) == (len(z[f"ad_{method}_accept_dt"]), len(z[f"ad_{method}_reject_dt"]))
    np.testing.assert_allclose(f.accept, z[f"ad_{method}_accept_dt"], rtol=1e-7)
    np.testing.assert_allclose(f.reject, z[f"ad_{method}_reject_dt"], rtol=1e-7)


@pytest.mark.parametrize("method", ADAPTIVE)
def test_adaptive_pairs_fp32_reverse_time(dev, method):
    z = load("methods.npz")
    A, y0 = T(z["ad_A"], dev).float(), T(z["ad_y0"], dev).float()
    f = StatFunc(lambda t_, y_: torch.sin(2 * t_) * (y_ @ A.T) * 2 - 0.1 * y_)
    with torch.no_grad():
        y = tda.odeint(f, y0, torch.tensor([1.0, 0.3, 0.0], dtype=torch.float64), rtol=1e-4, atol=1e-6,
                       method=method)
    assert y.dtype == torch.float32
    # The heuristic first step is ~5x smaller than the tolerance needs, so its fp32 error estimate is rounding
    # noise and the second dt differs from the reference's by ~2 % (measured: dopri5 0.3145 vs
    # 0.3138, tsit5 0.1947 vs 0.1906); both solutions are equally valid at rtol = 1e-4 and agree to ~3e-6.
    assert rel_err(y, z[f"ad32_{method}_y"]) < 1e-5
    assert abs(f.nfe - int(z[f"ad32_{method}_nfe"])) <= tda.SOLVERS[method].tableau.n_stages


@pytest.mark.parametrize("method", FIXED)
def test_fixed_grid_methods(dev, method):
    """grid = t, `step_size` (+ linear interpolation), `perturb`, `interp='cubic'`, decreasing time, fp64:
    no reductions anywhere -> bit-identical to the reference when func is the same torch CPU code."""
    z = load("methods.npz")
    A, y0, t = T(z["fx_A"], dev), T(z["fx_y0"], dev), T(z["fx_t"], dev)
    f = lambda t_, y_: torch.cos(t_) * (y_ @ A.T) - 0.1 * y_
    A64 = A.double()
    f64 = lambda t_, y_: torch.cos(t_) * (y_ @ A64.T) - 0.1 * y_
    with torch.no_grad():
        got = {
            "grid": tda.odeint(f, y0, torch.linspace(0, 1, 9), method=method),
            "step": tda.odeint(f, y0, t, method=method, options=dict(step_size=0.1)),
            "perturb": tda.odeint(f, y0, t, method=method, options=dict(step_size=0.1, perturb=True)),
            "cubic": tda.odeint(f, y0, t, method=method, options=dict(step_size=0.1, interp="cubic")),
            "rev": tda.odeint(f, y0, torch.tensor([1.0, 0.45, 0.0]), method=method,
                              options=dict(step_size=0.125, interp="cubic")),
            "f64": tda.odeint(f64, y0.double(), t.double(), method=method, options=dict(step_size=0.05)),
        }
    for tag, y in got.items():
        ref = T(z[f"fx_{method}_{tag}"], dev)
        assert y.shape == ref.shape and y.dtype == ref.dtype
        if dev == "cpu":
            assert torch.equal(y, ref), tag
        else:       # cos / GEMM of the field are evaluated by the GPU (1-ulp differences vs the CPU's)
            assert rel_err(y, ref) < (1e-12 if tag == "f64" else 2e-6), tag


def test_fixed_grid_unknown_interp_raises(dev):
    with pytest.raises(ValueError, match="Unknown interpolation"):
        with torch.no_grad():
            tda.odeint(lambda t_, y_: -y_, torch.ones(3), torch.tensor([0.0, 1.0]), method="euler",
                       options=dict(interp="quintic"))


def test_adjoint_with_low_order_methods(dev):
    """odeint_adjoint with a fixed-grid forward and an adaptive low-order backward method."""
    torch.manual_seed(4)
    lin = torch.nn.Linear(3, 3).double()
    params = tuple(lin.parameters())
    f = lambda t_, y_: torch.tanh(lin(y_))
    y0 = torch.randn(4, 3, dtype=torch.float64, requires_grad=True)
    t = torch.linspace(0, 1, 41, dtype=torch.float64)
    grads = {}
    for method, amethod, opts, aopts in [("heun3", "heun3", None, None), ("midpoint", "bosh3", {}, {}),
                                         ("dopri5", "tsit5", {}, {})]:
        y0.grad = None
        y = tda.odeint_adjoint(f, y0, t, method=method, adjoint_method=amethod, options=opts, adjoint_options=aopts,
                               rtol=1e-9, atol=1e-11, adjoint_params=params)
        y[-1].pow(2).sum().backward()
        grads[method] = y0.grad.clone()
    assert rel_err(grads["heun3"], grads["dopri5"]) < 1e-4
    assert rel_err(grads["midpoint"], grads["dopri5"]) < 1e-3
