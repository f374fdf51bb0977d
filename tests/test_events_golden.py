"""SURVEY.md §8(f) ranks 3-4 — event handling (`odeint(..., event_fn=)`, `odeint_event`, solver
`integrate_until_event`) and `odeint_dense`, against the reference's outputs (tests/golden/events.npz).

`dev` = "cuda": product on the MI355X; "cpu": the product's host logic over the oracle kernels (test-only)."""
import numpy as np
import pytest
import torch

import torchdiffeq_amd as tda
from _cases import T, load, rel_err

CASES = [("dopri5", {}), ("dopri8", {}), ("tsit5", {}), ("bosh3", {}), ("adaptive_heun", {}),
         ("rk4", dict(step_size=0.01)), ("rk4", dict(step_size=0.01, interp="cubic")),
         ("euler", dict(step_size=0.001)), ("midpoint", dict(step_size=0.01, interp="cubic")),
         ("heun3", dict(step_size=0.02, interp="cubic"))]


def _ev_scalar(t, y):
    return y[0, 0] - 0.5


def _ev_multi(t, y):
    return torch.stack([y[0, 0] + 1.0, y[1, 1] + 0.25, t - 5.0])


@pytest.mark.parametrize("method,opts", CASES, ids=[m + ("_cubic" if o.get("interp") else "") for m, o in CASES])
@pytest.mark.parametrize("ename", ["scalar", "multi"])
@pytest.mark.parametrize("rev", [False, True], ids=["fwd", "rev"])
def test_event_time_and_state(dev, method, opts, ename, rev):
    z = load("events.npz")
    A, y0 = T(z["ev_A"], dev), T(z["ev_y0"], dev)
    t = torch.tensor([0.0, -1.0] if rev else [0.0, 1.0], dtype=torch.float64)
    efn = _ev_scalar if ename == "scalar" else _ev_multi
    with torch.no_grad():
        te, ye = tda.odeint(lambda t_, y_: y_ @ A.T, y0, t, event_fn=efn, method=method, options=opts, rtol=1e-8,
                            atol=1e-9)
    tag = method + ("_cubic" if opts.get("interp") == "cubic" else "")
    key = f"ev_{tag}_{ename}_{'rev' if rev else 'fwd'}"
    assert te.dtype == torch.float64 and te.dim() == 0 and ye.shape == (2, 3, 2)
    assert torch.equal(ye[0], y0)
    if "step_size" in opts:
        # fixed grid: no error estimate anywhere -> same steps and the same bisection sequence as the reference
        assert float(te) == pytest.approx(float(z[key + "_t"]), rel=1e-12, abs=1e-13)
        assert rel_err(ye, z[key + "_y"]) < 1e-11
    else:
        # adaptive: the heuristic first step is so small that its error estimate is fp64 rounding noise, so the
        # second dt differs from the reference's by ~1e-9 relative (measured -5.4e-11 absolute for dopri5; torch.sum
        # over the stage axis is not our left-to-right sum); the bracket [t0, t1] handed to the
        # bisection moves by that much and the event time with it — well inside the bisection tolerance (atol).
        # dopri8 takes ~0.5-long steps, over which the quartic dense output is only ~1e-7 accurate: its event time
        # moves by that much when the (noise-floor dependent, ~1 %) second step size moves the step boundaries.
        tol_t, tol_y = (1e-6, 1e-6) if method == "dopri8" else (2e-9, 1e-8)
        assert float(te) == pytest.approx(float(z[key + "_t"]), abs=tol_t)
        assert rel_err(ye, z[key + "_y"]) < tol_y
    # the event function is (numerically) zero at the returned point
    c = efn(te if not rev else te, ye[-1])
    assert float(c.abs().min()) < 1e-6


def test_event_fp32_and_tuple_state(dev):
    z = load("events.npz")
    A, y0 = T(z["ev_A"], dev), T(z["ev_y0"], dev)
    A32 = A.float()
    with torch.no_grad():
        te, ye = tda.odeint(lambda t_, y_: y_ @ A32.T, y0.float(), torch.tensor([0.0, 1.0]), event_fn=_ev_scalar,
                            method="rk4", options=dict(step_size=0.01), atol=1e-6)
        assert te.dtype == torch.float32
        # fixed grid keeps time in the state dtype: the bisection sequence is reproduced in fp32
        assert float(te) == pytest.approx(float(z["ev32_rk4_t"]), rel=1e-6)
        assert rel_err(ye, z["ev32_rk4_y"]) < 1e-5
        te, ye = tda.odeint(lambda t_, y_: y_ @ A32.T, y0.float(), torch.tensor([0.0, 1.0]), event_fn=_ev_scalar,
                            method="dopri5", rtol=1e-5, atol=1e-6)
        assert float(te) == pytest.approx(float(z["ev32_dopri5_t"]), rel=1e-5)
        assert rel_err(ye, z["ev32_dopri5_y"]) < 1e-5
        te, (ya, yb) = tda.odeint(lambda t_, y_: (y_[0] @ A.T, -y_[1]), (y0, torch.ones(2, dtype=torch.float64)),
                                  torch.tensor([0.0, 1.0], dtype=torch.float64),
                                  event_fn=lambda t_, y_: y_[0][0, 0] - y_[1][0], method="dopri5", rtol=1e-8, atol=1e-9)
    assert float(te) == pytest.approx(float(z["ev_tuple_t"]), abs=2e-9)
    assert rel_err(ya, z["ev_tuple_ya"]) < 1e-8 and rel_err(yb, z["ev_tuple_yb"]) < 1e-8


def test_event_api_errors(dev):
    f = lambda t_, y_: -y_
    y0 = torch.ones(3, dtype=torch.float64)
    with pytest.raises(ValueError, match="len\\(t\\) == 2"):
        tda.odeint(f, y0, torch.tensor([0.0, 1.0, 2.0], dtype=torch.float64), event_fn=lambda t_, y_: y_[0] - 0.5)
    with pytest.raises(AssertionError, match="step_size"):
        with torch.no_grad():
            tda.odeint(f, y0, torch.tensor([0.0, 1.0], dtype=torch.float64), event_fn=lambda t_, y_: y_[0] - 0.5,
                       method="rk4")
    # event already satisfied at t0: returned immediately (rk_common.py:254-255)
    with torch.no_grad():
        te, ye = tda.odeint(f, y0, torch.tensor([0.0, 1.0], dtype=torch.float64), event_fn=lambda t_, y_: y_[0] - 1.0)
    assert float(te) == 0.0 and torch.equal(ye[1], y0)


@pytest.mark.parametrize("rev", [False, True], ids=["fwd", "rev"])
def test_odeint_event_gradients_through_event_time(dev, rev):
    """odeint_event + odeint_adjoint: d(event_t)/d(y0, A, t0) and d(state at the event)/d(...) via the
    implicit-function rerouting (odeint.py:160-231)."""
    z = load("events.npz")

    class F(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.A = torch.nn.Parameter(T(z["ev_A"], dev).clone())

        def forward(self, t, y):
            return y @ self.A.T

    func = F()
    y0 = T(z["ev_y0"], dev).clone().requires_grad_(True)
    t0 = torch.tensor(0.0, dtype=torch.float64, requires_grad=True)
    te, sol = tda.odeint_event(func, y0, t0, event_fn=_ev_scalar, reverse_time=rev, odeint_interface=tda.odeint_adjoint,
                               method="dopri5", rtol=1e-9, atol=1e-10)
    (te * 3.0 + sol[-1].pow(2).sum()).backward()
    tag = "rev" if rev else "fwd"
    assert float(te.detach()) == pytest.approx(float(z[f"oe_{tag}_t"]), abs=1e-9)
    assert rel_err(sol.detach(), z[f"oe_{tag}_sol"]) < 1e-9
    assert rel_err(y0.grad, z[f"oe_{tag}_grad_y0"]) < 1e-7
    assert rel_err(func.A.grad, z[f"oe_{tag}_grad_A"]) < 1e-7
    assert float(t0.grad) == pytest.approx(float(z[f"oe_{tag}_grad_t0"]), rel=1e-7)


def test_odeint_dense(dev):
    z = load("events.npz")
    A, y0 = T(z["ev_A"], dev), T(z["ev_y0"], dev)
    t_eval = z["dense_t_eval"]
    fn = tda.odeint_dense(lambda t_, y_: y_ @ A.T, y0, torch.tensor(0.0, dtype=torch.float64),
                          torch.tensor(3.0, dtype=torch.float64), rtol=1e-6, atol=1e-8, method="dopri5")
    got = torch.stack([fn(torch.tensor(float(te), dtype=torch.float64)) for te in t_eval])
    assert got.shape == (6, 3, 2)
    assert rel_err(got, z["dense_y_eval"]) < 1e-12
    assert fn.interp_coeffs.shape[1:] == (5, 6) and len(fn.times) == fn.interp_coeffs.shape[0] + 1
    A32 = A.float()
    fn = tda.odeint_dense(lambda t_, y_: y_ @ A32.T, y0.float(), torch.tensor(0.0), torch.tensor(3.0), rtol=1e-4,
                          atol=1e-6, method="dopri5")
    got = torch.stack([fn(torch.tensor(float(te))) for te in t_eval])
    assert rel_err(got, z["dense32_y_eval"]) < 1e-5
    with pytest.raises(IndexError):
        fn(torch.tensor(fn.times[-1], dtype=torch.float64))     # the end of the last step indexes past the last interval, as in the reference
    with pytest.raises(AssertionError):
        tda.odeint_dense(lambda t_, y_: -y_, y0, torch.tensor(0.0), torch.tensor(1.0), method="rk4")
