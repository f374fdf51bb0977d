"""SURVEY.md §8(f) rank 1 — backpropagation THROUGH the solver for plain `odeint`: every kernel call is one
autograd node with a hand-written backward (torchdiffeq_amd/autodiff.py, tdeq_scale_many / tdeq_multi_dot).
Gradients wrt y0, the field's parameters and `t` against the reference's autograd-through-eager-ops
(tests/golden/backprop.npz, fp64).

Tolerances: every method computes the same discrete function as the reference, including the reference's
differentiable FIRST step size (`_select_initial_step`, misc.py:36-77, is not under no_grad: dt0 is a function of
y0, f0 and f1 — torchdiffeq_amd.solvers._InitialStepShadow records the same graph) -> 1e-9 for the fixed-grid
methods, 1e-8 for the adaptive ones (measured 1e-14 ... 8e-12 on the CPU host-logic run; the GPU evaluates the field
with its own libm)."""
import numpy as np
import pytest
import torch

import torchdiffeq_amd as tda
from _cases import T, load, rel_err

# dopri8 at rtol 1e-8 takes steps whose error estimate is rounding noise: a field evaluated by the GPU's libm moves
# the step sequence, the solution by 1e-9 and the gradients with it (CPU host-logic run: 8e-12)
CASES = [("dopri5", "dopri5", None, False, 1e-8), ("dopri8", "dopri8", None, False, 1e-6),
         ("tsit5", "tsit5", None, False, 1e-8), ("bosh3", "bosh3", None, False, 1e-8),
         ("fehlberg2", "fehlberg2", None, False, 1e-8), ("adaptive_heun", "adaptive_heun", None, False, 1e-8),
         ("dopri5_rev", "dopri5", None, False, 1e-8), ("dopri5_tuple", "dopri5", None, True, 1e-8),
         ("rk4_grid", "rk4", None, False, 1e-9), ("euler_grid", "euler", None, False, 1e-9),
         ("midpoint_step", "midpoint", dict(step_size=0.1), False, 1e-9),
         ("heun2_perturb", "heun2", dict(step_size=0.1, perturb=True), False, 1e-9),
         ("heun3_cubic", "heun3", dict(step_size=0.1, interp="cubic"), False, 1e-9),
         ("rk4_cubic_rev", "rk4", dict(step_size=0.125, interp="cubic"), False, 1e-9),
         # r04 (found by `tools/fuzz_vs_reference.py hostexact`): a step that ends on a `step_t` / `jump_t` point
         ("dopri5_step_t_first", "dopri5", dict(step_t=[0.013]), False, 1e-9),
         ("dopri5_step_t_mid", "dopri5", dict(step_t=[0.55]), False, 1e-9),
         ("bosh3_jump_t_mid", "bosh3", dict(jump_t=[0.55]), False, 1e-9),
         ("tsit5_tuple_step_jump", "tsit5", dict(step_t=[0.2, 0.8], jump_t=[0.5]), True, 1e-9),
         ("dopri5_rev_step_t", "dopri5", dict(step_t=[0.6]), False, 1e-9),
         ("bosh3_min_step", "bosh3", dict(min_step=0.5), False, 1e-9)]      # heuristic first step clamped: a constant


def _problem(z, dev):
    p = [T(z[k], dev).clone().requires_grad_(True) for k in ("bp_W1", "bp_b1", "bp_W2")]
    y0 = T(z["bp_y0"], dev).clone().requires_grad_(True)
    field = lambda t_, y_: torch.tanh(y_ @ p[0].T + p[1]) @ p[2].T * torch.cos(t_) - 0.1 * y_
    return p, y0, field


@pytest.mark.parametrize("tag,method,opts,tup,tol", CASES, ids=[c[0] for c in CASES])
def test_backprop_through_solver_matches_reference(dev, tag, method, opts, tup, tol):
    z = load("backprop.npz")
    p, y0, field = _problem(z, dev)
    t = T(z[f"bp_{tag}_t"], dev).clone().requires_grad_(True)
    kw = dict(rtol=1e-4, atol=1e-6) if method in ("fehlberg2", "adaptive_heun") else dict(rtol=1e-8, atol=1e-10)
    if tup:
        f = lambda t_, y_: (field(t_, y_[0]), -y_[1] * y_[0].sum(-1, keepdim=True))
        ya, yb = tda.odeint(f, (y0, torch.ones(y0.shape[0], 1, dtype=torch.float64)), t, method=method, options=opts, **kw)
        loss = ya[-1].pow(2).sum() + ya[1].sum() + yb[-1].sum()
        sol = ya
    else:
        sol = tda.odeint(field, y0, t, method=method, options=opts, **kw)
        loss = sol[-1].pow(2).sum() + sol[1].sum()
    assert sol.requires_grad
    loss.backward()
    assert rel_err(sol.detach(), z[f"bp_{tag}_y"]) < max(tol * 1e-2, 1e-11)
    assert rel_err(y0.grad, z[f"bp_{tag}_g_y0"]) < tol
    for g, name in zip(p, ("g_W1", "g_b1", "g_W2")):
        assert rel_err(g.grad, z[f"bp_{tag}_{name}"]) < tol, name
    assert rel_err(t.grad, z[f"bp_{tag}_g_t"]) < tol


@pytest.mark.parametrize("method,opts", [("dopri5", None), ("tsit5", None), ("rk4", None),
                                         ("heun3", dict(step_size=0.07, interp="cubic")),
                                         ("midpoint", dict(step_size=0.07))])
def test_gradcheck_y0_and_t(dev, method, opts):
    """The reference's own gradient test (gradient_tests.py:13-23): gradcheck of odeint wrt (y0, t)."""
    torch.manual_seed(0)
    A = torch.randn(3, 3, dtype=torch.float64) * 0.5
    f = lambda t_, y_: torch.sin(t_) * torch.tanh(y_ @ A.T) - 0.2 * y_
    y0 = torch.randn(2, 3, dtype=torch.float64, requires_grad=True)
    t = torch.tensor([0.0, 0.3, 0.9], dtype=torch.float64, requires_grad=True)
    fn = lambda y0_, t_: tda.odeint(f, y0_, t_, method=method, options=opts, rtol=1e-9, atol=1e-11)
    assert torch.autograd.gradcheck(fn, (y0, t), eps=1e-6, atol=1e-5, rtol=1e-3)


def test_backprop_agrees_with_adjoint(dev):
    """gradient_tests.py:34-87: discretise-then-optimise (this path) vs optimise-then-discretise (adjoint)."""
    z = load("backprop.npz")
    grads = {}
    for name, solve in (("bp", tda.odeint), ("adj", tda.odeint_adjoint)):
        p, y0, field = _problem(z, dev)
        t = torch.tensor([0.0, 0.4, 1.0], dtype=torch.float64, requires_grad=True)
        kw = dict(adjoint_params=tuple(p)) if name == "adj" else {}
        y = solve(field, y0, t, rtol=1e-9, atol=1e-12, **kw)
        (y[-1].pow(2).sum() + y[1].sum()).backward()
        grads[name] = [y0.grad, t.grad] + [q.grad for q in p]
    for a, b in zip(grads["bp"], grads["adj"]):
        assert rel_err(a, b) < 1e-6


def test_no_graph_when_nothing_requires_grad(dev):
    """Grad mode on but nothing differentiable: the plain kernels run (no autograd nodes, rows written in place)."""
    A = torch.randn(3, 3, dtype=torch.float64)
    y = tda.odeint(lambda t_, y_: y_ @ A.T, torch.ones(2, 3, dtype=torch.float64),
                   torch.tensor([0.0, 0.5, 1.0], dtype=torch.float64))
    assert not y.requires_grad and y.grad_fn is None


def test_ode_demo_style_training_step(dev):
    """examples/ode_demo.py:150-170: fit a small net by backprop through dopri5 — the loss must go down."""
    torch.manual_seed(0)
    true_A = torch.tensor([[-0.1, 2.0], [-2.0, -0.1]])
    y0 = torch.tensor([[2.0, 0.0]])
    t = torch.linspace(0.0, 1.0, 6)
    with torch.no_grad():
        target = tda.odeint(lambda t_, y_: (y_ ** 3) @ true_A, y0, t, method="dopri5")
    net = torch.nn.Sequential(torch.nn.Linear(2, 16), torch.nn.Tanh(), torch.nn.Linear(16, 2))
    opt = torch.optim.Adam(net.parameters(), lr=1e-2)
    losses = []
    for _ in range(12):
        opt.zero_grad()
        pred = tda.odeint(lambda t_, y_: net(y_ ** 3), y0, t, method="dopri5", rtol=1e-5, atol=1e-6)
        loss = (pred - target).abs().mean()
        loss.backward()
        opt.step()
        losses.append(loss.item())
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in net.parameters())
    assert losses[-1] < losses[0]
