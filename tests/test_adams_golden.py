"""Adams multistep methods of the reference's SOLVERS table — `explicit_adams`, `implicit_adams`, `fixed_adams`
(fixed_adams.py:164-228) — against the reference's own outputs (tests/golden/adams.npz, written by
tests/golden/make_golden.py::gen_adams running the reference).

`dev` fixture as in test_methods_golden.py: "cuda" = the product on the MI355X (tdeq_adams_predict /
tdeq_adams_correct), "cpu" = the product's host logic with the oracle substituted for the HIP kernels.  The
methods contain no reduction other than the convergence test, which is an exact census, so with the same torch CPU
code evaluating the field the results must be BIT-IDENTICAL to the reference's; on the GPU the field's cos / sin /
GEMM differ from the CPU's by an ulp, hence a tolerance there."""
import warnings

import numpy as np
import pytest
import torch

import torchdiffeq_amd as tda
from torchdiffeq_amd.tableaus import adams_coefficients
from _cases import T, load, rel_err

METHODS = ["explicit_adams", "implicit_adams"]


def _field(t, y):
    """Exactly rounded elementwise operations only (see make_golden.adams_field): the same bits on any CPU or GPU."""
    return (1 - t * 0.5) * (y.roll(1, -1) * 0.3 - y * 0.2) - y * y * y * 0.01


class _Count:
    def __init__(self, fn):
        self.fn, self.nfe = fn, 0

    def __call__(self, t, y):
        self.nfe += 1
        return self.fn(t, y)


def test_solver_table_has_the_adams_methods():
    names = list(tda.SOLVERS)
    assert names[11:13] == ["explicit_adams", "implicit_adams"] and "fixed_adams" in names
    assert tda.SOLVERS["fixed_adams"] is tda.SOLVERS["implicit_adams"]
    assert issubclass(tda.SOLVERS["explicit_adams"], tda.SOLVERS["implicit_adams"])
    assert tda.SOLVERS["implicit_adams"].order == 4


def test_coefficients_bit_identical_to_the_reference_tables():
    """Generated in exact rational arithmetic here; the reference divides integer tables (fixed_adams.py:10-156)."""
    z = load("adams.npz")
    for k in range(1, 13):
        bash, moulton = adams_coefficients(k)
        assert np.array_equal(np.array(bash), z[f"bashforth_{k}"]), k
        if k > 1:      # the reference's order-1 Moulton entry is 1/11 (a typo it never uses: orders start at 4)
            assert np.array_equal(np.array(moulton), z[f"moulton_{k}"]), k


@pytest.mark.parametrize("method", METHODS)
def test_solves_match_the_reference(dev, method):
    """Bit-identical solutions, evaluation counts (= corrector iterations) and non-convergence warnings, on the
    host-logic path and on the MI355X alike: the field consists of exactly rounded operations, so its values do not
    depend on where it is evaluated and only the solver arithmetic is under test.  (A field with transcendental
    functions or a GEMM differs by an ulp between machines, and the top-order fp32 Adams–Bashforth formula —
    coefficients of magnitude 1e3 — amplifies that to 1e-3 in the solution: the method's own conditioning.)"""
    z = load("adams.npz")
    y0, t = T(z["y0"], dev), T(z["t"], dev)
    cases = {
        "grid": (y0, torch.linspace(0, 1, 41, device="cpu").to(dev), {}),
        "step": (y0, t, dict(step_size=0.02)),
        "perturb": (y0, t, dict(step_size=0.02, perturb=True)),
        "cubic": (y0, t, dict(step_size=0.02, interp="cubic")),
        "rev": (y0, torch.tensor([1.0, 0.45, 0.0]), dict(step_size=0.025, interp="cubic")),
        "order6": (y0, t, dict(step_size=0.02, max_order=6)),
        "iters1": (y0, t, dict(step_size=0.02, max_iters=1)),
        "f64": (y0.double(), t.double(), dict(step_size=0.0125)),
    }
    for tag, (y, tt, opts) in cases.items():
        c = _Count(_field)
        with warnings.catch_warnings(record=True) as w, torch.no_grad():
            warnings.simplefilter("always")
            got = tda.odeint(c, y, tt, method=method, options=opts, rtol=1e-6, atol=1e-8)
        ref = T(z[f"{method}_{tag}"], dev)
        assert got.shape == ref.shape and got.dtype == ref.dtype
        assert torch.equal(got, ref), tag
        assert c.nfe == int(z[f"{method}_{tag}_nfe"]), tag
        n_warn = sum("did not converge" in str(x.message) for x in w)      # (a ResourceWarning of another test may land here)
        assert n_warn == int(z[f"{method}_{tag}_warnings"]), tag


@pytest.mark.parametrize("method", METHODS)
def test_fixed_adams_alias_and_reference_defaults(dev, method):
    """`fixed_adams` is the implicit method; without tolerances odeint passes its own defaults (1e-7, 1e-9)."""
    z = load("adams.npz")
    y0 = T(z["y0"], dev)
    with torch.no_grad():
        a = tda.odeint(_field, y0, torch.linspace(0, 1, 41), method="fixed_adams", rtol=1e-6, atol=1e-8)
        b = tda.odeint(_field, y0, torch.linspace(0, 1, 41), method="implicit_adams", rtol=1e-6, atol=1e-8)
    assert torch.equal(a, b)


@pytest.mark.parametrize("method", METHODS)
def test_tuple_state_with_per_component_tolerances(dev, method):
    z = load("adams.npz")
    y0 = T(z["y0"], dev)
    ft = lambda t, y: (_field(t, y[0]), -y[1] * y[0][0, :3] * (1 + t))
    yt = (y0, torch.tensor([0.5, 0.25, 1.0]))
    with torch.no_grad():
        out = tda.odeint(ft, yt, torch.linspace(0, 1, 31, dtype=torch.float64, device="cpu").to(dev), method=method,
                         rtol=(1e-6, 1e-5), atol=(1e-8, 1e-7))
    for i in range(2):
        ref = T(z[f"{method}_tuple{i}"], dev)
        assert out[i].dtype == ref.dtype and out[i].shape == ref.shape
        assert torch.equal(out[i], ref)


@pytest.mark.parametrize("method", METHODS)
def test_zero_dim_fp32_state_promotion_quirk(dev, method):
    """A 0-dim fp32 state multiplies the reference's 0-dim fp64 coefficients as 0-dim x 0-dim, which promotes to fp64:
    `_dot_product` runs in fp64 and is rounded once (fixed_adams.py:205-214).  Reproduced bit for bit."""
    z = load("adams.npz")
    fs = lambda t, y: (1 - t * 0.5) * (y * -0.7) - y * y * y * 0.01
    with torch.no_grad():
        y = tda.odeint(fs, torch.tensor(1.5), torch.linspace(0, 1, 41, device="cpu").to(dev), method=method,
                       rtol=1e-6, atol=1e-8)
    ref = T(z[f"{method}_zerodim"], dev)
    assert y.shape == ref.shape == (41,) and y.dtype == torch.float32
    assert torch.equal(y, ref)


@pytest.mark.parametrize("method", ["euler", "midpoint", "heun3", "rk4", "explicit_adams", "implicit_adams"])
def test_zero_dim_fp32_state_on_fp64_grid_runs_in_fp64_like_the_reference(dev, method):
    """`dt` (0-dim fp64) x `f` (0-dim fp32) promotes: after the first evaluation the reference's solve is fp64 and only
    the stored rows are fp32 (odeint._zero_dim_promotion).  Bit for bit."""
    z = load("adams.npz")
    fs = lambda t, y: (1 - t * 0.5) * (y * -0.7) - y * y * y * 0.01
    with torch.no_grad():
        y = tda.odeint(fs, torch.tensor(1.5), torch.linspace(0, 1, 21, dtype=torch.float64, device="cpu").to(dev),
                       method=method, rtol=1e-6, atol=1e-8)
    ref = T(z[f"zerodim64_{method}"], dev)
    assert y.dtype == torch.float32 and torch.equal(y, ref)


@pytest.mark.parametrize("method", METHODS)
def test_backprop_through_the_solver(dev, method):
    """Gradients wrt y0, t and the field's parameters equal the reference's autograd-through-eager-ops result."""
    z = load("adams.npz")
    lin = torch.nn.Linear(3, 3).double().to(dev)
    with torch.no_grad():
        lin.weight.copy_(T(z[f"{method}_bp_w"], dev))
        lin.bias.copy_(T(z[f"{method}_bp_b"], dev))
    y0 = T(z[f"{method}_bp_y0"], dev).requires_grad_(True)
    t = torch.linspace(0, 1, 21, dtype=torch.float64).requires_grad_(True)
    y = tda.odeint(lambda t_, y_: torch.tanh(lin(y_)) * torch.cos(t_), y0, t, method=method, rtol=1e-6, atol=1e-8)
    assert rel_err(y.detach(), z[f"{method}_bp_y"]) < 1e-10      # tanh / cos / GEMM: an ulp apart between machines, amplified
    loss = y[-1].pow(2).sum() + y[7].sum()
    g = torch.autograd.grad(loss, [y0, t, lin.weight, lin.bias])
    for name, v in zip(["gy0", "gt", "gw", "gb"], g):
        ref = T(z[f"{method}_bp_{name}"])
        assert float((v.cpu() - ref).abs().max()) < 1e-9 * max(1.0, float(ref.abs().max())), name


@pytest.mark.parametrize("method", METHODS)
def test_event_mode(dev, method):
    z = load("adams.npz")
    fe = lambda t, y: torch.stack([y[1], -y[0]])
    et, ys = tda.odeint_event(fe, torch.tensor([1.0, 0.0], dtype=torch.float64),
                              torch.tensor(0.0, dtype=torch.float64), event_fn=lambda t, y: y[0], method=method,
                              options=dict(step_size=0.01), atol=1e-8)
    assert abs(float(et) - float(z[f"{method}_event_t"])) < 1e-12
    assert rel_err(ys, z[f"{method}_event_y"]) < 1e-12


def test_adjoint_with_adams_methods(dev):
    """odeint_adjoint accepts the Adams methods for the forward and the backward solve."""
    g = torch.Generator(device="cpu").manual_seed(4)       # the same problem on every device
    lin = torch.nn.Linear(3, 3).double().to(dev)
    with torch.no_grad():
        lin.weight.copy_(torch.randn(3, 3, generator=g, dtype=torch.float64, device="cpu") * 0.5)
        lin.bias.copy_(torch.randn(3, generator=g, dtype=torch.float64, device="cpu") * 0.5)
    params = tuple(lin.parameters())
    f = lambda t_, y_: torch.tanh(lin(y_))
    y0 = torch.randn(4, 3, generator=g, dtype=torch.float64, device="cpu").to(dev).requires_grad_(True)
    t = torch.linspace(0, 1, 41, dtype=torch.float64)
    grads = {}
    for method in ["dopri5", "explicit_adams", "implicit_adams"]:
        y0.grad = None
        y = tda.odeint_adjoint(f, y0, t, method=method, rtol=1e-9, atol=1e-11, adjoint_params=params)
        y[-1].pow(2).sum().backward()
        grads[method] = y0.grad.clone()
    # the methods' own accuracy at 40 steps (measured 1.0e-4 / 9.8e-10; the reference gives the same numbers)
    assert rel_err(grads["explicit_adams"], grads["dopri5"]) < 1e-3
    assert rel_err(grads["implicit_adams"], grads["dopri5"]) < 1e-7


def test_option_checks(dev):
    y0, t = torch.ones(3), torch.linspace(0, 1, 5)
    with pytest.raises(AssertionError, match="max_order must be at most"):
        tda.odeint(lambda t_, y_: -y_, y0, t, method="implicit_adams", options=dict(max_order=13))
    with pytest.warns(UserWarning, match="reduces to `rk4`"):
        with torch.no_grad():
            a = tda.odeint(lambda t_, y_: -y_, y0, t, method="implicit_adams", options=dict(max_order=3))
            b = tda.odeint(lambda t_, y_: -y_, y0, t, method="rk4")
    assert torch.equal(a, b)
