"""hipGraph mode of the fixed-grid rk4 solver (options={'hip_graph': True}): one captured step replayed per grid
interval must reproduce the eager path bit for bit (same kernels, same operation order, times formed on the device
with the host's rounding sequence)."""
import warnings

import pytest
import torch

import torchdiffeq_amd as tda
from torchdiffeq_amd.misc import check_inputs
from torchdiffeq_amd.odeint import SOLVERS

pytestmark = pytest.mark.gpu


def _spiral():
    A = torch.tensor([[-0.1, 2.0], [-2.0, -0.1]], device="cuda")
    return (lambda t, y: (y ** 3) @ A), torch.tensor([[2.0, 0.0]], device="cuda")


def test_cfg1_graph_equals_eager_and_reference_bits():
    f, y0 = _spiral()
    t = torch.linspace(0.0, 25.0, 1000, device="cuda")
    with torch.no_grad():
        y_eager = tda.odeint(f, y0, t, method="rk4")
        y_graph = tda.odeint(f, y0, t, method="rk4", options=dict(hip_graph=True))
    assert torch.equal(y_graph, y_eager)
    # the reference's CPU result for cfg1
    assert y_graph[-1, 0].tolist() == [-0.4436032772064209, 0.27951884269714355]


@pytest.mark.parametrize("state_dtype", [torch.float32, torch.float64], ids=["f32", "f64"])
@pytest.mark.parametrize("grid_dtype", [torch.float32, torch.float64], ids=["t32", "t64"])
@pytest.mark.parametrize("perturb", [False, True])
@pytest.mark.parametrize("reverse", [False, True])
def test_graph_mode_time_dependent_field(state_dtype, grid_dtype, perturb, reverse):
    g = torch.Generator().manual_seed(4)
    y0 = torch.randn(257, 3, generator=g, dtype=torch.float64).to(state_dtype).cuda()
    w = torch.randn(3, 3, generator=g, dtype=torch.float64).to(state_dtype).cuda() * 0.3
    f = lambda t, y: torch.tanh(y @ w) * torch.cos(3 * t) - 0.1 * y * t
    t = torch.linspace(0.3, 2.1, 37, dtype=grid_dtype, device="cuda")
    if reverse:
        t = t.flip(0)
    opts = dict(perturb=perturb)
    with torch.no_grad():
        y_eager = tda.odeint(f, y0, t, method="rk4", options=dict(opts))
        y_graph = tda.odeint(f, y0, t, method="rk4", options=dict(opts, hip_graph=True))
    assert torch.equal(y_graph, y_eager)


def test_graph_mode_counts_evaluations_and_handles_short_grids():
    f, y0 = _spiral()
    for n_t in (1, 2, 3, 10):
        t = torch.linspace(0.0, 0.5, n_t, device="cuda") if n_t > 1 else torch.tensor([0.0], device="cuda")
        ci = check_inputs(f, y0, t, 1e-7, 1e-9, "rk4", dict(hip_graph=True), None, SOLVERS)
        solver = SOLVERS["rk4"](func=ci.func, y0=ci.y0_flat, rtol=ci.rtol, atol=ci.atol, **ci.options)
        with torch.no_grad():
            y = solver.integrate(ci.t)
            ref = tda.odeint(f, y0, t, method="rk4")
        assert torch.equal(y.view_as(ref), ref)
        assert ci.func.nfe == 4 * (n_t - 1)


def test_graph_mode_falls_back_with_a_warning_when_not_applicable():
    f, y0 = _spiral()
    t = torch.linspace(0.0, 1.0, 5, device="cuda")
    with torch.no_grad():
        ref = tda.odeint(f, y0, t, method="rk4", options=dict(step_size=0.05))
        with warnings.catch_warnings(record=True) as rec:
            warnings.simplefilter("always")
            y = tda.odeint(f, y0, t, method="rk4", options=dict(step_size=0.05, hip_graph=True))
    assert any("hip_graph" in str(w.message) for w in rec)
    assert torch.equal(y, ref)
