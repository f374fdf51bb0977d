"""Differences found by the program-level differential runs of round 4b (tools/fuzz_programs_vs_reference.py,
tools/fuzz_api_programs_vs_reference.py), pinned to outputs of the reference (tests/golden/r4b.npz <- make_golden.py r4b):
16-bit states under every explicit fixed-grid method — also on a 16-bit TIME grid, which numpy cannot hold —
(torchdiffeq/_impl/solvers.py:102-126, rk_common.py:110-157), the shape a tensor state has in the user's callbacks
(misc.py:313-333) and the evaluation sequence of cubic interpolation (solvers.py:119-122)."""
import warnings

import numpy as np
import pytest
import torch

import torchdiffeq_amd as tda
from _cases import load

Z = load("r4b.npz")
LOW = {"bf16": torch.bfloat16, "f16": torch.float16}


@pytest.mark.parametrize("direction", ["fwd", "rev"])
@pytest.mark.parametrize("opts", ["plain", "cubic", "perturb"])
@pytest.mark.parametrize("method", ["euler", "midpoint", "heun2", "heun3", "rk4"])
@pytest.mark.parametrize("tname", ["t32", "tlow"])
@pytest.mark.parametrize("lname", ["bf16", "f16"])
def test_sixteen_bit_states_on_fixed_grids_are_the_reference_bit_for_bit(lname, tname, method, opts, direction):
    """CPU host path (the 16-bit element types have no HIP kernels, docs/LAB_NOTEBOOK.md §10): every stored row equals the
    reference's in every bit, with the same number of evaluations.  heun3's `k * (1/3)` / `k * (2/3)` take the Python
    weight at fp32 (ATen's second-operand rule), which is what this pins; `tlow` = the time grid in the state's own
    16-bit type (dt, the stage times and the interpolation weights are then 16-bit 0-dim tensors in the reference)."""
    ldtype = LOW[lname]
    A = torch.tensor(Z["low_A"]).to(ldtype)
    y0 = torch.tensor(Z["low_y0"]).to(ldtype)
    tv = [0.0, 0.3, 0.55, 1.0] if direction == "fwd" else [1.0, 0.55, 0.3, 0.0]
    t = torch.tensor(tv).to(torch.float32 if tname == "t32" else ldtype)
    options = {"plain": {}, "cubic": dict(step_size=0.13, interp="cubic"), "perturb": dict(perturb=True)}[opts]
    calls = []

    def field(t_, y_):
        calls.append(t_.dtype)
        return y_ @ A.T - y_ * 0.5 * torch.cos(t_).to(y_.dtype)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", tda.HostPathWarning)
        y = tda.odeint(field, y0, t, method=method, options=options)
    key = f"lowgrid_{lname}_{tname}_{method}_{opts}_{direction}"
    assert y.dtype == ldtype and len(calls) == int(Z[key + "_nfe"])
    assert all(d == ldtype for d in calls)              # misc.py:185-187: func sees t in y0.abs().dtype
    assert torch.equal(y.float(), torch.tensor(Z[key + "_y"]))


@pytest.fixture(params=["cpu", pytest.param("cuda", marks=pytest.mark.gpu)])
def on(request):
    with warnings.catch_warnings():
        warnings.simplefilter("ignore" if request.param == "cpu" else "error", tda.HostPathWarning)
        yield request.param


@pytest.mark.parametrize("direction", ["fwd", "rev"])
@pytest.mark.parametrize("method,options", [("dopri5", {}), ("rk4", dict(step_size=0.25))])
def test_callbacks_see_a_tensor_state_in_its_own_shape(on, method, options, direction):
    """misc.py:313-333: only a TUPLE state is re-shaped for the callbacks — a tensor state was never flattened by the
    reference, so `callback_step(t0, y0, dt)` gets y0 as [2, 3], not as the package's flat [6] vector."""
    A = torch.tensor(Z["low_A"], device=on)
    y0 = torch.tensor(Z["low_y0"], device=on)
    seen = []

    class Field(torch.nn.Module):
        def forward(self, t, y):
            return y @ A.T * torch.cos(t)

        def callback_step(self, t0, y_, dt):
            seen.append(("step", tuple(y_.shape), float(t0), float(dt), float(y_.sum())))

        def callback_accept_step(self, t0, y_, dt):
            seen.append(("accept", tuple(y_.shape), float(t0), float(dt), float(y_.sum())))
    tv = [0.0, 0.5, 1.0] if direction == "fwd" else [1.0, 0.5, 0.0]
    with warnings.catch_warnings():
        warnings.filterwarnings("ignore", message="Solver 'rk4' does not support callbacks")
        tda.odeint(Field(), y0, torch.tensor(tv, device=on), method=method, rtol=1e-4, atol=1e-6, options=dict(options))
    key = f"cb_{method}_{direction}"
    assert seen and all(s[1] == (2, 3) for s in seen) and Z[key + "_shape"].tolist() == [[2, 3]] * len(Z[key + "_kind"])
    if on == "cuda" and method == "dopri5" and len(seen) != len(Z[key + "_kind"]):
        return          # (a noise-driven extra trial step on the device: the shapes above are what this test is about)
    assert [s[0] for s in seen] == list(Z[key + "_kind"])
    # values: exact on the CPU host path; on the MI355X the fp32 error ratio carries summation-order noise, which the
    # controller turns into slightly different adaptive step sizes (docs/LAB_NOTEBOOK.md §12) — the fixed grid stays at rounding
    rtol = 1e-12 if on == "cpu" else (0.2 if method == "dopri5" else 2e-6)
    np.testing.assert_allclose(np.array([s[2:] for s in seen]), Z[key + "_vals"], rtol=rtol, atol=1e-6)


def test_cubic_interpolation_calls_are_the_reference_sequence(on):
    y0 = torch.tensor(Z["low_y0"], device=on)
    calls = []

    def counted(t_, y_):
        calls.append(float(t_))
        return -y_ * (1.0 + t_)
    y = tda.odeint(counted, y0, torch.tensor([0.0, 0.1, 0.2, 0.25, 0.7, 1.0], device=on), method="heun2",
                   options=dict(step_size=0.5, interp="cubic"))
    assert calls == Z["cubic_calls"].tolist()
    np.testing.assert_allclose(y.cpu().numpy(), Z["cubic_y"], rtol=0 if on == "cpu" else 1e-6, atol=0 if on == "cpu" else 1e-7)


@pytest.mark.parametrize("nname,norm", [("mixed", None), ("semi", "seminorm")])
def test_adjoint_norm_takes_the_time_component_as_abs_not_as_rms(on, nname, norm):
    """adjoint.py:250, 273: `max(t.abs(), state_norm(y), ...)`.  The scaled time VJP of this field is ~1e29: its square
    is inf in fp32, an rms of the one-element component would make the norm inf and the backward solve's initial step
    0 ("underflow in dt 0.0") where the reference integrates — which the host path did until round 4b."""
    class Steep(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.w = torch.nn.Parameter(torch.tensor([0.5, -0.25], device=on))

        def forward(self, t, y):
            return -y + self.w * torch.sin(t * 1e20) * 1e3
    f = Steep()
    x = torch.tensor([[1.0, 2.0]], device=on, requires_grad=True)
    y = tda.odeint_adjoint(f, x, torch.tensor([0.0, 1e-18], device=on), method="dopri5", rtol=1e-3, atol=1e-6,
                           adjoint_options=dict(norm=norm) if norm else None)
    y[-1].sum().backward()
    tol = dict(rtol=0, atol=0) if on == "cpu" else dict(rtol=1e-5, atol=1e-30)
    np.testing.assert_allclose(y.detach().cpu().numpy(), Z[f"steep_{nname}_y"], **tol)
    np.testing.assert_allclose(x.grad.cpu().numpy(), Z[f"steep_{nname}_gy"], **tol)
    # (dL/dw = 1e-18-sized integral of an oscillating integrand, far below atol: its digits follow the step sequence,
    #  which on the device differs by the usual fp32 noise — what matters there is that the backward solve RUNS)
    np.testing.assert_allclose(f.w.grad.cpu().numpy(), Z[f"steep_{nname}_gw"], **(tol if on == "cpu" else dict(rtol=0.5)))


@pytest.mark.parametrize("sname", ["tensor2d", "tuple"])
def test_grid_constructor_gets_func_and_state_in_the_references_form(on, sname):
    """solvers.py:103 `self.grid_constructor(self.func, self.y0, t)`: y0 is a tensor state in ITS shape, a tuple state —
    also the adjoint's augmented state of the backward solve — as the unpadded concatenation of its components (never
    the package's chunk-padded flat buffer), and `func(t, y0)` works on it.  This grid is refined by the field's size."""
    seen = []
    w = torch.tensor([0.5, -0.3, 0.8], device=on, requires_grad=True)

    def grid(func, y, tt):
        d = func(tt[0], y)
        seen.append((tuple(y.shape), tuple(d.shape), float(d.abs().max())))
        return torch.linspace(float(tt[0]), float(tt[-1]), 2 + int(float(d.abs().max()) * 3), device=on).to(tt)
    if sname == "tensor2d":
        f = lambda t_, y_: -y_ * w * (1 + t_) + torch.sin(y_)       # noqa: E731
        state = torch.tensor([[1.0, 2.0, 3.0], [0.5, 0.1, -1.0]], device=on, requires_grad=True)
    else:
        f = lambda t_, y_: (-y_[0] * w * (1 + t_), torch.sin(y_[1]) - y_[0].sum())     # noqa: E731
        state = (torch.tensor([[1.0, 2.0, 3.0]], device=on, requires_grad=True), torch.tensor([0.5, 0.1], device=on))
    sol = tda.odeint_adjoint(f, state, torch.tensor([0.0, 0.4, 1.0], device=on), method="rk4",
                             options=dict(grid_constructor=grid), adjoint_params=(w,))
    (sol[0] if sname == "tuple" else sol)[-1].sum().backward()
    assert all(s[0] == s[1] for s in seen)
    assert [s[0][0] if len(s[0]) == 1 else -1 for s in seen] == Z[f"grid_{sname}_yshapes"].tolist()
    tol = dict(rtol=0, atol=0) if on == "cpu" else dict(rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose([s[2] for s in seen], Z[f"grid_{sname}_dmax"], **tol)
    np.testing.assert_allclose(w.grad.cpu().numpy(), Z[f"grid_{sname}_gw"], **tol)
    np.testing.assert_allclose((sol[0] if sname == "tuple" else sol).detach().cpu().numpy(), Z[f"grid_{sname}_y"], **tol)


@pytest.mark.parametrize("method", ["bosh3", "dopri5"])
def test_complex_adjoint_is_the_reference_bit_for_bit_on_the_host_path(method):
    """|z| and z / real are not single IEEE operations: ATen's vectorised loop and its scalar tail round them differently,
    and which one an element gets depends on its position in the tensor.  The reference forms the error ratio of the
    backward solve on the concatenated augmented state (rk_common.py:22-27), so must the host path (`_fallback._joint`)."""
    y0 = torch.view_as_complex(torch.tensor(Z["cplx_y0"]))
    A = torch.view_as_complex(torch.tensor(Z["cplx_A"]))
    a_ = A.clone().requires_grad_(True)
    x = y0.clone().requires_grad_(True)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", tda.HostPathWarning)
        y = tda.odeint_adjoint(lambda t_, y_: y_ @ a_ - y_ * 0.5, x, torch.linspace(0.0, 1.0, 4, dtype=torch.float64),
                               method=method, adjoint_params=(a_,))
    y[-1].abs().sum().backward()
    assert torch.equal(torch.view_as_real(y.detach()), torch.tensor(Z[f"cplx_{method}_y"]))
    assert torch.equal(torch.view_as_real(a_.grad), torch.tensor(Z[f"cplx_{method}_gA"]))
    assert torch.equal(torch.view_as_real(x.grad), torch.tensor(Z[f"cplx_{method}_gy"]))


def test_heun3_with_a_non_finite_stage_gives_the_references_rows_on_the_host_path():
    """fixed_grid.py:38-44 multiplies every stage by its tableau weight, the zeros too: `k1 * 0.0` is NaN for an inf k1.
    The host path evaluates that literal expression (the HIP kernels do not read zero-weight terms: inf stays inf there,
    docs/LAB_NOTEBOOK.md §8)."""
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", tda.HostPathWarning)
        y = tda.odeint(lambda t_, y_: torch.where(t_ > 0.4, torch.full_like(y_, float("inf")), -y_),
                       torch.tensor([1.0, 2.0, 0.5], dtype=torch.float64), torch.tensor([0.0, 1.0, 3.0], dtype=torch.float64),
                       method="heun3", options=dict(step_size=0.25))
    ref = torch.tensor(Z["heun3_inf_field_y"])
    assert torch.equal(y.isnan(), ref.isnan()) and torch.equal(y.nan_to_num(), ref.nan_to_num())
    assert bool(ref[-1].isnan().all())


def test_dopri8_blow_up_ends_after_the_references_number_of_evaluations_on_the_host_path():
    """rk_common.py:79-89 multiplies every stage by its weight: near the blow-up of y' = y^2 a dopri8 stage is inf while its
    weight in the error row is 0, `inf * 0 = NaN`, the step size becomes NaN and the solve ends in `underflow in dt 0.0`
    after 470 evaluations.  The host path's rows keep their zero weights (`tableaus.SparseRow.literal`); summed over the
    non-zero stages only the estimate stayed finite and the same assertion came after 1133 (the kernels' behaviour,
    docs/LAB_NOTEBOOK.md §8)."""
    calls = []

    def square(t_, y_):
        calls.append(1)
        return y_[0] * y_[0], -y_[1]
    state = (torch.tensor([1.1101932525634766, 1.453037977218628, 1.1249815225601196]), torch.ones(2))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", tda.HostPathWarning)
        with pytest.raises(AssertionError) as info, torch.no_grad():
            tda.odeint(square, state, torch.tensor([0.0, 1.0, 3.0]), method="dopri8", rtol=1e-3, atol=1e-6,
                       options=dict(max_num_steps=200))
    assert str(info.value) == str(Z["blowup_dopri8_message"]) and len(calls) == int(Z["blowup_dopri8_calls"])


@pytest.mark.parametrize("iface", ["odeint", "odeint_adjoint"])
def test_event_solve_on_a_trajectory_that_turns_nan_returns_the_references_time(on, iface):
    """odeint.py:160-231 / event_handling.py:5-20: a NaN sign "differs" from every sign, so the bisection still returns
    a finite event time for a trajectory that left the finite range (a diverged training run), the state NaN, every
    gradient NaN.  The first-order correction `odeint_event` attaches here must not turn that time into NaN."""
    class NanField(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.w = torch.nn.Parameter(torch.tensor(1.0, device=on))

        def forward(self, t_, y_):
            return torch.where(t_ > 0.55, torch.full_like(y_, float("nan")), -y_ * self.w)
    f = NanField()
    x = torch.tensor([1.0, 2.0], device=on, requires_grad=True)
    et, ys = tda.odeint_event(f, x, torch.tensor(0.0, device=on), event_fn=lambda t_, y_: y_[0] - 0.1, method="rk4",
                              options=dict(step_size=0.1), odeint_interface=getattr(tda, iface), atol=1e-6, rtol=1e-4)
    (et + 0).backward()
    assert bool(torch.isfinite(et.detach()))
    assert float(et.detach()) == pytest.approx(float(Z[f"nan_event_{iface}_t"]), rel=0 if on == "cpu" else 1e-6)
    assert np.array_equal(ys.detach().cpu().numpy(), Z[f"nan_event_{iface}_y"], equal_nan=True)
    assert bool(f.w.grad.isnan()) and bool(np.isnan(Z[f"nan_event_{iface}_gw"]))
    assert bool(x.grad.isnan().all()) and bool(np.isnan(Z[f"nan_event_{iface}_gy"]).all())
