"""Corners of the drop-in surface found by a differential run against the imported reference (round 4, build container:
75 API / error cases, same exception classes and results after these fixes).  No GPU and no reference needed here: each
test pins what the reference does, cited."""
import warnings

import numpy as np
import pytest
import torch

import torchdiffeq_amd as tda


class _Field(torch.nn.Module):
    def __init__(self):
        super().__init__()
        torch.manual_seed(0)
        self.lin = torch.nn.Linear(3, 3)

    def forward(self, t, y):
        return torch.tanh(self.lin(y)) * torch.cos(t)


_Y0 = torch.tensor([[0.3, -0.2, 0.5], [1.0, 0.1, -0.7]])
_T = torch.linspace(0, 1, 5)


@pytest.fixture(params=["cpu", pytest.param("cuda", marks=pytest.mark.gpu)])
def on(request):
    """The device of the state: the torch-op host path ("cpu") or the HIP kernels ("cuda") — same behaviour on both."""
    global Y0, T
    Y0, T = _Y0.to(request.param), _T.to(request.param)
    prev = torch.get_default_device()
    torch.set_default_device(request.param)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore" if request.param == "cpu" else "error", tda.HostPathWarning)
        yield request.param
    torch.set_default_device(prev)


def test_step_size_zero_is_torchs_range_error(on):
    # solvers.py:86-96: torch.arange(0, inf) -> RuntimeError
    with pytest.raises(RuntimeError, match="unsupported range"):
        tda.odeint(_Field(), Y0, T, method="rk4", options=dict(step_size=0.0))


def test_norm_returning_more_than_one_element_is_a_runtime_error(on):
    # misc.py:60 `if d0 < 1e-5 or d1 < 1e-5` on a vector
    with pytest.raises(RuntimeError, match="more than one value"):
        tda.odeint(_Field(), Y0, T, method="dopri5", options=dict(norm=lambda x: x))


def test_unknown_adjoint_norm_string_fails_in_the_backward_solve_not_before(on):
    # adjoint.py:271-288: everything but "seminorm" is taken for a callable
    f = _Field()
    y = tda.odeint_adjoint(f, Y0.clone().requires_grad_(True), T, adjoint_options=dict(norm="semi"))
    with pytest.raises(TypeError, match="not callable"):
        y[-1].sum().backward()


def test_tuple_state_of_mixed_dtypes_is_promoted_as_a_whole(on):
    # misc.py:206-207: the components are concatenated -> one dtype for func's inputs and for every output
    seen = []

    def f(t, y):
        seen.append(tuple(c.dtype for c in y))
        return (-y[0], -0.5 * y[1])
    a, b = tda.odeint(f, (Y0, Y0.double()), T, method="dopri5")
    assert a.dtype == b.dtype == torch.float64 and set(seen) == {(torch.float64, torch.float64)}
    a64, b64 = tda.odeint(lambda t, y: (-y[0], -0.5 * y[1]), (Y0.double(), Y0.double()), T, method="dopri5")
    assert torch.equal(a, a64) and torch.equal(b, b64)
    z, w = tda.odeint(lambda t, y: (-y[0], 1j * y[1]), (Y0, Y0.to(torch.complex64)), T, method="rk4")
    assert z.dtype == w.dtype == torch.complex64


def test_func_returning_a_python_number(on):
    # fixed_grid.py: `y0 + dt * 1.0` broadcasts; rk_common.py:69 needs `.shape` -> AttributeError
    y = tda.odeint(lambda t, y: 1.0, Y0, T, method="rk4")
    assert torch.allclose(y[-1], Y0 + 1.0, atol=1e-6)
    with pytest.raises(AttributeError) as exc:
        tda.odeint(lambda t, y: 1.0, Y0, T, method="dopri5")
    assert isinstance(exc.value, TypeError)          # (what this package has raised so far)


def test_integer_state_is_refused_with_the_references_exception_class(on):
    # misc.py:185-196: nextafter is not implemented for integers
    with pytest.raises(NotImplementedError) as exc:
        tda.odeint(lambda t, y: y, torch.tensor([1, 2]), T, method="dopri5")
    assert isinstance(exc.value, TypeError)


@pytest.mark.parametrize("method", ["euler", "midpoint", "rk4"])
def test_integer_state_on_a_fixed_grid_is_refused_where_the_reference_truncates(on, method):
    """Advisor r05: the one documented drop-in DEVIATION for a call the reference completes.  Its fixed-grid methods
    accept an int64 state — every step is promoted to float by `y0 + dt * f` and TRUNCATED back into the int64 solution
    buffer (tests/golden/int_state.npz holds what it returns: e.g. midpoint [4, 8] -> [2, 5] -> [1, 3]) —, its adaptive ones
    raise NotImplementedError (recorded there too).  No ODE has an integer state; this package refuses both with one
    exception that is a TypeError and a NotImplementedError, and says how to convert."""
    from _cases import load
    from torchdiffeq_amd.misc import UnsupportedStateDtype
    z = load("int_state.npz")
    assert z[f"int_{method}_y"].dtype == np.int64 and str(z["int_dopri5_error"]) == "NotImplementedError"
    with pytest.raises(UnsupportedStateDtype, match=r"y0\.float\(\)"):
        tda.odeint(lambda t, y: -0.5 * y, torch.tensor([4, 8]), torch.tensor([0.0, 1.0, 2.0]), method=method)
    # the conversion the message recommends gives the un-truncated solution the reference's arithmetic was heading for
    y = tda.odeint(lambda t, y: -0.5 * y, torch.tensor([4.0, 8.0]), torch.tensor([0.0, 1.0, 2.0]), method=method)
    assert torch.equal(y[1].to(torch.int64), torch.from_numpy(z[f"int_{method}_y"][1]).to(y.device))


def test_zero_tolerances_fail_like_the_reference_without_numpy_warnings(on):
    # rtol = atol = 0: the heuristic divides 0 by 0 — 0-dim tensors do that silently, so do the host scalars
    with warnings.catch_warnings():
        warnings.simplefilter("error", RuntimeWarning)
        with pytest.raises(AssertionError):
            tda.odeint(_Field(), Y0, T, rtol=0.0, atol=0.0, method="dopri5", options=dict(max_num_steps=50))


def test_dense_output_takes_one_time_per_call(on):
    # odeint.py:151-156: a vector of query times indexes the coefficient stack out of bounds -> IndexError
    dense = tda.odeint_dense(lambda t, y: -y, Y0, torch.tensor(0.0), torch.tensor(1.0))
    assert torch.allclose(dense(torch.tensor(0.5)), Y0 * torch.exp(torch.tensor(-0.5)), atol=1e-6)
    with pytest.raises(IndexError):
        dense(torch.tensor([0.25, 0.5]))


def test_nonfinite_state_message_shows_the_state_without_alignment_padding(on):
    # rk_common.py:280: the assertion prints the flat state; a tuple state's flat vector here is padded per component
    with pytest.raises(AssertionError, match="non-finite values in state") as exc:
        tda.odeint(lambda t, y: (-y[0], -y[1]), (torch.tensor([float("inf"), 1.0]), torch.ones(3)), T, method="dopri5",
                   options=dict(first_step=0.1))
    assert "tensor([inf, 1., 1., 1., 1.]" in str(exc.value)          # 5 values, not a chunk-padded vector


def test_cubic_interpolation_evaluates_the_step_end_once_per_output_time(on):
    # solvers.py:119-122: `f1 = self.func(t1, y1)` sits INSIDE the loop over the output times of a step — a counting
    # (or stateful) func sees one call per output time, not one per step (found by tools/fuzz_programs_vs_reference.py)
    calls = []

    def field(t, y):
        calls.append(float(t))
        return -y * (1.0 + t)

    t = torch.tensor([0.0, 0.1, 0.2, 0.25, 0.7, 1.0])
    tda.odeint(field, Y0, t, method="heun2", options=dict(step_size=0.5, interp="cubic"))
    # two steps x two stage evaluations, + f1 for the 3 output times in (0, 0.5] and the 2 in (0.5, 1]
    assert len(calls) == 2 * 2 + 3 + 2
    assert calls == [0.0, 0.5, 0.5, 0.5, 0.5, 0.5, 1.0, 1.0, 1.0]

