"""Drop-in gaps found by running the reference's OWN test suite against the package (tools/run_reference_tests.py, r03 —
all 22 tests / 1213 subtests pass there on the host path); pinned here to reference outputs (tests/golden/dropin.npz <-
make_golden.py dropin) so they also run on the HIP path:

  * a 0-dim fp32 state under the Adams methods while autograd records (fixed_adams.py:205-216: 0-dim x 0-dim products
    promote to fp64 whether or not something requires grad) — solution bit for bit, gradients, and the event time of the
    diverging explicit-Adams run of event_tests.py:14-49, which rounding decides;
  * what a norm placed in `grad_fn.adjoint_options['norm']` is called with when the forward state is a tuple
    (norm_tests.py:155-191, adjoint.py:243-288): (t, y, adj_y, *adj_params) with y and adj_y FLAT; the auto-built norm is
    callable on exactly that;
  * the SciPy wrapper handing back fewer rows than len(t) when solve_ivp gives up (odeint_tests.py:251-268)."""
import math
import warnings

import pytest
import torch

import torchdiffeq_amd as tda
from torchdiffeq_amd import _fallback
from _cases import T, load


@pytest.fixture(params=["test-backend-or-hip", "host-path"])
def where(request, dev, monkeypatch):
    """`dev` = cpu runs the host logic over the oracle kernels, cuda the HIP kernels; "host-path" (cpu only) removes the
    substitution again so the package's own torch-op path for CPU states runs."""
    if request.param == "host-path":
        if dev != "cpu":
            pytest.skip("the host path is the CPU half")
        monkeypatch.undo()
        torch.set_default_device("cpu")
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", _fallback.HostPathWarning)
        yield dev
    torch.set_default_device(None)


@pytest.mark.parametrize("tag,tdtype", [("t32", torch.float32), ("t64", torch.float64)])
@pytest.mark.parametrize("method", ["explicit_adams", "implicit_adams"])
def test_zero_dim_fp32_adams_with_autograd_recording(where, method, tag, tdtype):
    z = load("dropin.npz")
    key = f"adams0_{method}_{tag}"
    w = torch.tensor(0.7, requires_grad=True)
    x = torch.tensor(1.3, requires_grad=True)
    t = torch.linspace(0.0, 0.5, 26, dtype=tdtype).requires_grad_(True)
    y = tda.odeint(lambda t_, y_: -y_ * w * (1 + t_) + torch.sin(3 * t_), x, t, method=method)
    assert y.dtype == torch.float32 and y.shape == (26,)
    if where == "cpu":
        assert torch.equal(y.detach(), T(z[f"{key}_y"])), float((y.detach() - T(z[f"{key}_y"])).abs().max())
    else:       # the field's sin() is the device's, not the CPU's: last-place differences
        assert torch.allclose(y.detach().cpu(), T(z[f"{key}_y"]), rtol=3e-6, atol=0)
    y[-1].backward()
    for got, name in ((x.grad, "gx"), (w.grad, "gw"), (t.grad, "gt")):
        want = T(z[f"{key}_{name}"])
        assert got.dtype == want.dtype
        # fp32 gradient of 25 steps (explicit Adams: an oscillating, cancelling time gradient of magnitude 1e3)
        assert float((got.cpu() - want).abs().max()) <= 2e-4 * float(want.abs().max()), name


def test_event_time_of_the_diverging_explicit_adams_run(where):
    """event_tests.py:14-49, ode='sine', fp32, explicit_adams: the run is unstable (tolerance 7e-2 in the reference's
    test) and the sign change that ends it is decided by rounding — equal only if every step is."""
    z = load("dropin.npz")
    t_points, sol = T(z["sine_t_points"]), T(z["sine_sol"])

    class Sine(torch.nn.Module):
        def forward(self, t_, y_):
            return 2 * y_ / t_ + t_ ** 4 * torch.sin(2 * t_) - t_ ** 2 + 4 * t_ ** 3
    sol_d = sol.to(where)
    et, ys = tda.odeint(Sine(), sol_d[0], t_points[0:2].to(where), event_fn=lambda t_, y_: torch.sum(y_ - sol_d[2]).real,
                        method="explicit_adams", options={"step_size": 0.01, "interp": "cubic"})
    assert et.dtype == torch.float64
    if where == "cpu":
        assert float(et) == float(z["sine_event_t"]) and torch.equal(ys.detach().cpu(), T(z["sine_event_y"]))
    else:       # sin / pow of the device's libm differ from the CPU's in the last place; the run amplifies that
        assert abs(float(et) - float(t_points[2])) / float(t_points[2]) < 7e-2


@pytest.mark.parametrize("tag", ["default", "seminorm"])
def test_tuple_state_adjoint_norm_calling_convention(where, tag):
    z = load("dropin.npz")
    p1 = T(z["adjnorm_p1"]).to(where).requires_grad_(True)
    p2 = T(z["adjnorm_p2"]).to(where).requires_grad_(True)
    x0 = (torch.tensor(1.0), torch.tensor([[0.5, 0.5], [0.1, 0.1]]))
    kw = dict(adjoint_options=dict(norm="seminorm")) if tag == "seminorm" else {}
    xs = tda.odeint_adjoint(lambda t_, x_: (x_[0] * p2, x_[1] * p1[:4].reshape(2, 2)), x0, torch.tensor([0.0, 1.0]),
                            adjoint_params=(p1, p2), **kw)
    # the reference's test reaches the Function through the views of the tuple output (norm_tests.py:168)
    opts = xs[0].grad_fn.next_functions[0][0].next_functions[0][0].adjoint_options
    auto = opts["norm"]
    seen = []

    def spy(tensors):
        assert isinstance(tensors, tuple)
        t_, y, adj_y, a1, a2 = tensors
        assert t_.shape == () and y.shape == (5,) and adj_y.shape == (5,) and a1.shape == (7,) and a2.shape == ()
        want = max(t_.abs(), y[0].abs(), y[1:].pow(2).mean().sqrt(), adj_y[0].abs(), adj_y[1:].pow(2).mean().sqrt())
        if tag == "default":
            want = max(want, a1.pow(2).mean().sqrt(), a2.abs())
        got = auto(tensors)
        assert isinstance(got, torch.Tensor) and got.shape == ()
        assert float((got - want).abs()) <= 1e-6 * max(1.0, float(want))
        seen.append(float(got))
        return got
    opts["norm"] = spy
    (xs[0].sum() + xs[1].sum()).backward()
    assert len(seen) == len(z[f"adjnorm_{tag}_n_entries"])          # as many norm evaluations as the reference: same steps
    for a, b in zip(seen[:3], z[f"adjnorm_{tag}_first_values"]):
        assert a == pytest.approx(float(b), rel=1e-4)
    assert torch.allclose(p1.grad.cpu(), T(z[f"adjnorm_{tag}_gp1"]), rtol=1e-4, atol=1e-6)
    assert torch.allclose(p2.grad.cpu(), T(z[f"adjnorm_{tag}_gp2"]), rtol=1e-4, atol=1e-6)


def test_user_adjoint_norm_still_sees_the_components_of_a_tuple_state(where):
    """adjoint.py:271-288: the USER's own adjoint norm gets (t, *y, *adj_y, *adj_params); what sits in adjoint_options
    afterwards is the wrapper taking the flat form."""
    p = torch.rand(3, requires_grad=True)
    shapes = []

    def norm(tensors):
        shapes.append([tuple(v.shape) for v in tensors])
        return max(v.abs().max() for v in tensors)
    x0 = (torch.tensor(1.0), torch.tensor([[0.5, 0.5], [0.1, 0.1]]))
    xs = tda.odeint_adjoint(lambda t_, x_: (x_[0] * p[0], x_[1] * p[1]), x0, torch.tensor([0.0, 1.0]),
                            adjoint_params=(p,), adjoint_options=dict(norm=norm))
    wrapper = xs[0].grad_fn.next_functions[0][0].next_functions[0][0].adjoint_options["norm"]
    assert wrapper is not norm
    (xs[0].sum() + xs[1].sum()).backward()
    assert shapes and all(s == [(), (), (2, 2), (), (2, 2), (3,)] for s in shapes)
    flat = (torch.tensor(0.5), torch.arange(5.0), -torch.arange(5.0), torch.ones(3))
    n0 = len(shapes)
    assert float(wrapper(flat)) == 4.0 and len(shapes) == n0 + 1


def test_scipy_wrapper_passes_a_short_solution_through():
    """odeint_tests.py:251-268 with LSODA and min_step = 2: solve_ivp gives up after the first output; the reference
    returns the rows it got (scipy_wrapper.py:43-51, odeint.py:98-101) instead of failing on the reshape.  In a
    subprocess: ODEPACK reports the failure through Fortran's buffered stdout, which would otherwise land after pytest's
    summary line."""
    pytest.importorskip("scipy")
    import os
    import subprocess
    import sys
    code = """
import math, sys, warnings, torch
sys.path.insert(0, %r)
import torchdiffeq_amd as tda
warnings.simplefilter("ignore")
A = torch.tensor([[-0.5, 2.0], [-2.0, -0.5]], dtype=torch.float64) * 4
t = torch.linspace(1.0, 8.0, 10, dtype=torch.float64)
y0 = torch.ones(2, dtype=torch.float64)
full = tda.odeint(lambda t_, y_: y_ @ A.T, y0, t, method="scipy_solver", options=dict(solver="LSODA"))
short = tda.odeint(lambda t_, y_: y_ @ A.T * math.exp(3.0), y0, t, method="scipy_solver",
                   options=dict(solver="LSODA", min_step=2.0, max_step=5.0))
import json
print("SHAPES", json.dumps([list(full.shape), list(short.shape)]))
""" % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("SHAPES")][0]
    import json
    full_shape, short_shape = json.loads(line[len("SHAPES"):])
    assert full_shape == [10, 2]
    assert short_shape[1:] == [2] and 1 <= short_shape[0] <= 10


def test_flat_state_padding_is_zero_so_step_size_gradients_stay_finite(where, monkeypatch):
    """examples/cnf.py without --adjoint (found by tools/run_reference_examples.py): tuple state, y0 without grad, func
    parameters with grad.  The flat state pads every component to a chunk boundary; the kernels stream over the padding
    and the gradient of the first step size is a dot product over the WHOLE flat vector (tdeq_multi_dot) — padding left
    uninitialised by StateLayout.pack turned into NaN parameter gradients whenever the allocator handed back memory
    holding NaNs.  Here every fresh `torch.empty` is poisoned to provoke exactly that."""
    real_empty = torch.empty

    def poisoned_empty(*args, **kw):
        out = real_empty(*args, **kw)
        if out.is_floating_point():
            out.fill_(float("nan"))
        return out
    lin = torch.nn.Linear(3, 3).to(where)
    x = torch.randn(5, 3)
    logp = torch.zeros(5, 1)

    def func(t_, state):
        z, _ = state
        dz = torch.tanh(lin(z)) * torch.cos(t_)
        return dz, dz.sum(1, keepdim=True)
    monkeypatch.setattr(torch, "empty", poisoned_empty)
    z_t, l_t = tda.odeint(func, (x, logp), torch.tensor([1.0, 0.0]), rtol=1e-5, atol=1e-5, method="dopri5")
    monkeypatch.setattr(torch, "empty", real_empty)
    (z_t[-1].pow(2).sum() + l_t[-1].sum()).backward()
    for p in lin.parameters():
        assert torch.isfinite(p.grad).all()


@pytest.mark.parametrize("method", ["dopri5", "dopri8", "bosh3", "rk4", "euler", "implicit_adams"])
def test_empty_states(where, method):
    """Edge case: a state without elements.  Fixed-grid methods and empty COMPONENTS of a tuple behave as in the
    reference (shapes [len(t), 0, ...]); an entirely empty state under an adaptive method returns the empty solution
    here, where the reference fails with 'underflow in dt 0.0' (its RMS norm of nothing is NaN) — docs/LAB_NOTEBOOK.md §8."""
    t = torch.tensor([0.0, 0.5, 1.0])
    with torch.no_grad():
        y = tda.odeint(lambda t_, y_: -y_, torch.empty(0, 3), t, method=method)
        assert y.shape == (3, 0, 3)
        y = tda.odeint_adjoint(lambda t_, y_: -y_, torch.empty(0), t, method=method, adjoint_params=())
        assert y.shape == (3, 0)
        ya, yb = tda.odeint(lambda t_, s: (-s[0], -s[1]), (torch.ones(2), torch.empty(0)), t, method=method)
    assert ya.shape == (3, 2) and yb.shape == (3, 0)
    assert torch.allclose(ya[-1].cpu(), torch.full((2,), 0.36787944, device="cpu"), rtol=0.4 if method == "euler" else 1e-3)


@pytest.mark.parametrize("inner_method", ["rk4", "dopri5"])
def test_nested_solves(where, inner_method):
    """Re-entrancy ('nested odeint calls'): `func` integrates an inner ODE per evaluation — the inner
    solver runs while the outer one has a trial step (and, on the GPU, its look-ahead read-back) in flight.  Closed form
    of the inner solve: z(t + 0.1) = y exp(-(0.1 + 0.1 t + 0.005))."""
    import math
    w = torch.tensor(0.1, dtype=torch.float64, requires_grad=True)
    opts = dict(step_size=0.025) if inner_method == "rk4" else None

    def nested(t_, y):
        z = tda.odeint(lambda s, z_: -z_ * (1 + s), y, torch.stack([t_, t_ + 0.1]), method=inner_method, options=opts,
                       rtol=1e-10, atol=1e-12)[-1]
        return -y + w * z

    def closed(t_, y):
        return -y + w * y * torch.exp(-(0.105 + 0.1 * t_))
    y0 = torch.tensor([1.0, 2.0, 3.0], dtype=torch.float64)
    t = torch.tensor([0.0, 0.5, 1.0], dtype=torch.float64)
    grads = []
    for field in (nested, closed):
        w.grad = None
        y = tda.odeint(field, y0, t, rtol=1e-8, atol=1e-10)
        y[-1].sum().backward()
        grads.append((y.detach(), w.grad.clone()))
    assert torch.allclose(grads[0][0], grads[1][0], rtol=1e-7, atol=1e-9)
    assert torch.allclose(grads[0][1], grads[1][1], rtol=1e-6)
    assert math.isfinite(float(grads[0][1]))


@pytest.mark.gpu
@pytest.mark.parametrize("graph", [False, True], ids=["eager", "hip_graph"])
@pytest.mark.parametrize("own_streams", [False, True], ids=["one-stream", "stream-per-thread"])
def test_concurrent_solves_from_python_threads(own_streams, graph):
    """Four Python threads solve different problems at once (shared current stream, or a stream each): the C-ABI is
    stateless, every solver has its own norm plan / read-back words, the look-ahead controller's polling is per plan —
    results must be the serial ones bit for bit."""
    import threading
    dev = torch.device("cuda:0")
    probs = []
    for i in range(4):
        g = torch.Generator().manual_seed(100 + i)
        A = (torch.randn(16, 16, generator=g, dtype=torch.float64) / 6 - 0.2 * torch.eye(16, dtype=torch.float64))
        dtype = torch.float32 if i % 2 else torch.float64
        probs.append((A.to(dtype).to(dev), torch.randn(256 * (i + 1), 16, generator=g, dtype=torch.float64).to(dtype).to(dev),
                      torch.tensor([0.0, 0.7, 1.5], dtype=dtype, device=dev), ["dopri5", "dopri8", "bosh3", "rk4"][i]))

    fields = [(lambda A: (lambda t_, y: torch.tanh(y @ A.T) * torch.cos(t_)))(p[0]) for p in probs]     # one func object per problem

    def solve(A, y0, t, method, captured=False):
        o = dict(step_size=0.01) if method == "rk4" else {}
        if captured and method != "rk4":        # (the rk4 graph mode steps between the output times only)
            o["hip_graph"] = True               # concurrent stream captures: thread_local capture mode, one graph per func
        field = fields[[id(p[0]) for p in probs].index(id(A))]
        with torch.no_grad():
            return tda.odeint(field, y0, t, method=method, rtol=1e-6, atol=1e-8, options=o or None)
    serial = [solve(*p) for p in probs]
    torch.cuda.synchronize()
    out, errs = [None] * 4, []

    def work(i):
        try:
            ctx = torch.cuda.stream(torch.cuda.Stream(dev)) if own_streams else torch.cuda.device(dev)
            with ctx:
                for _ in range(5):
                    out[i] = solve(*probs[i], captured=graph)
                torch.cuda.current_stream().synchronize()
        except Exception as e:       # noqa: BLE001
            errs.append(repr(e))
    threads = [threading.Thread(target=work, args=(i,)) for i in range(4)]
    [th.start() for th in threads]
    [th.join(120) for th in threads]
    torch.cuda.synchronize()
    assert not errs, errs
    for a, b in zip(serial, out):
        assert b is not None and torch.equal(a, b)


@pytest.mark.parametrize("direction", ["fwd", "rev"])
@pytest.mark.parametrize("method,opts", [("rk4", dict(step_size=0.03, interp="cubic")), ("euler", dict(step_size=0.01, interp="linear")),
                                         ("midpoint", dict(step_size=0.02, interp="cubic")), ("dopri5", {})],
                         ids=["rk4", "euler", "midpoint", "dopri5"])
def test_event_gradients_including_the_start_time(where, method, opts, direction):
    """`odeint_event` gradients wrt the initial state, a parameter and the START time (found by a differential run of
    odeint_event gradients against the reference, r03): the fixed-grid solvers keep t0 in the graph of `t1 = t0 + dt` and of
    the interpolation fraction (solvers.py:130-164), so d(event time)/d t0 is there for them too — it used to come back
    None.  Values from the reference (tests/golden/dropin.npz)."""
    z = load("dropin.npz")
    key = f"evgrad_{method}_{direction}"
    y0 = torch.tensor([1.0, 0.1], dtype=torch.float64, requires_grad=True)
    t0 = torch.tensor(0.3, dtype=torch.float64, requires_grad=True)
    k = torch.tensor(1.0, dtype=torch.float64, requires_grad=True)
    et, ys = tda.odeint_event(lambda t_, y_: torch.stack([y_[1], -y_[0] * k * (1 + 0.5 * t_)]), y0, t0,
                              event_fn=lambda t_, y_: y_[0] - 0.3, method=method, options=dict(opts),
                              reverse_time=direction == "rev", atol=1e-9, rtol=1e-7)
    g = torch.autograd.grad(et * 2.0 + (ys[-1] ** 2).sum(), [y0, t0, k], allow_unused=True)
    tol = 1e-6 if method == "dopri5" else 1e-9
    assert float(et.detach()) == pytest.approx(float(z[f"{key}_t"]), rel=tol)
    assert torch.allclose(ys.detach().cpu(), T(z[f"{key}_y"]), rtol=tol, atol=tol)
    for got, name in zip(g, ("gy0", "gt0", "gk")):
        assert got is not None, name
        assert torch.allclose(got.cpu(), T(z[f"{key}_{name}"]), rtol=10 * tol, atol=10 * tol), name


@pytest.mark.parametrize("method", ["dopri5", "rk4"])
def test_adjoint_callbacks_see_the_reference_tuple_for_a_tuple_state(where, method):
    """adjoint.py:107-114 + misc.py:313-343: `callback_*_adjoint` receive the backward solve's state as
    (t, y, adj_y, *adj_params) with a tuple forward state FLAT (the reference's autograd Function only ever sees the
    flattened state), in un-negated time; forward callbacks see the components."""
    rec = {"fwd": [], "adj": []}

    class F(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.w = torch.nn.Parameter(torch.tensor([0.5, 0.2]))

        def forward(self, t_, st):
            return -st[0] * self.w[0], -st[1] * self.w[1] * torch.cos(t_)

        def callback_step(self, t0, y0, dt):
            rec["fwd"].append((float(t0), [tuple(v.shape) for v in y0]))

        def callback_step_adjoint(self, t0, y0, dt):
            assert isinstance(y0, tuple)
            rec["adj"].append((float(t0), [tuple(v.shape) for v in y0], float(dt)))
    x0 = (torch.tensor(1.0, requires_grad=True), torch.tensor([[0.5, 0.5], [0.1, 0.1]]))
    out = tda.odeint_adjoint(F(), x0, torch.tensor([0.0, 0.5, 1.0]), method=method,
                             options=dict(step_size=0.25) if method == "rk4" else None)
    (out[0][-1] + out[1][-1].sum()).backward()
    assert rec["fwd"][0] == (0.0, [(), (2, 2)])
    assert all(shapes == [(), (5,), (5,), (2,)] for _, shapes, _ in rec["adj"])
    times = [t for t, _, _ in rec["adj"]]
    assert times[0] == 1.0 and all(dt > 0 for _, _, dt in rec["adj"])
    if method == "rk4":
        assert times == [1.0, 0.75, 0.5, 0.25]


@pytest.mark.parametrize("direction", ["fwd", "rev"])
@pytest.mark.parametrize("method", ["euler", "midpoint", "heun3", "rk4", "explicit_adams", "implicit_adams"])
def test_zero_dim_fp32_state_on_an_fp64_grid_with_perturb(where, method, direction):
    """The last piece of the 0-dim promotion artefact (docs/LAB_NOTEBOOK.md §8): with `perturb` the reference perturbs the FIRST
    evaluation time in fp32 (the state is still fp32 there) and every later one in fp64.  Polynomial field (no libm), so
    the MI355X result equals the CPU reference bit for bit as well."""
    z = load("dropin.npz")
    t = torch.linspace(0.1, 0.6, 11, dtype=torch.float64)
    if direction == "rev":
        t = torch.linspace(0.6, 0.1, 11, dtype=torch.float64)
    with torch.no_grad():
        y = tda.odeint(lambda t_, y_: -y_ * (1 + 0.3 * t_) + 0.2 * t_ * t_, torch.tensor(0.7), t, method=method,
                       options=dict(perturb=True, step_size=0.013))
    want = T(z[f"zerodim_perturb_{method}_{direction}"])
    assert y.dtype == torch.float32 and torch.equal(y.cpu(), want), float((y.cpu() - want).abs().max())


@pytest.mark.parametrize("method,step", [("rk4", 0.1), ("heun3", 0.07), ("euler", 0.05)])
def test_second_order_time_gradients_through_cubic_interpolation(where, method, step):
    """Found by a differential run of Hessians against the reference (r03): with `interp='cubic'` the output between two
    grid points is a cubic Hermite polynomial in h = (t - t0) / (t1 - t0); its recorded node carried the first derivative
    of the basis only, so d2/dt2 of anything interpolated was wrong (heun3: 42 %).  Values from the reference."""
    z = load("dropin.npz")
    W = T(z["hesscubic_W"]).to(where).requires_grad_(True)
    x = T(z["hesscubic_x"]).to(where).requires_grad_(True)
    tt = torch.tensor([0.05, 0.43, 0.96], dtype=torch.float64, requires_grad=True)
    y = tda.odeint(lambda t_, y_: torch.tanh(y_ @ W.T) * torch.cos(t_), x, tt, method=method,
                   options=dict(step_size=step, interp="cubic"))
    loss = (y[-1] ** 2).sum() + (y[1] ** 3).sum()
    g1 = torch.autograd.grad(loss, (x, W, tt), create_graph=True)
    g2 = torch.autograd.grad(sum((v ** 2).sum() for v in g1), (x, W, tt))
    for name, got in zip(("gx", "gW", "gt", "hx", "hW", "ht"), list(g1) + list(g2)):
        want = T(z[f"hesscubic_{method}_{name}"])
        assert torch.allclose(got.detach().cpu(), want, rtol=1e-8, atol=1e-10 * float(want.abs().max())), name


def test_per_component_adjoint_tolerances_follow_the_reference_backward_state(where):
    """`adjoint_rtol` / `adjoint_atol` as sequences are per component of the REFERENCE's backward state
    (t, y, adj_y, *adj_params): 3 + P entries, also for a tuple forward state (whose y / adj_y are flat there) — spread over
    this package's 1 + 2 n_y + P segments.  A tuple forward `rtol` is inherited unchanged and fails the length check in the
    backward pass, in the reference as here (adjoint.py:167-170)."""
    z = load("dropin.npz")

    def run(**kw):
        p1 = torch.tensor([0.5, 0.2], dtype=torch.float64, requires_grad=True)
        p2 = torch.tensor(0.3, dtype=torch.float64, requires_grad=True)
        x = torch.tensor([1.0, 2.0, 3.0], dtype=torch.float64, requires_grad=True)
        zz = torch.tensor([[0.5, 0.1]], dtype=torch.float64)
        out = tda.odeint_adjoint(lambda t_, s: (-s[0] * p1[0] * torch.cos(t_) + s[1].sum() * p2, -s[1] * p1[1]), (x, zz),
                                 torch.tensor([0.0, 0.6, 1.0], dtype=torch.float64), adjoint_params=(p1, p2), method="dopri5", **kw)
        (out[0][-1].pow(2).sum() + out[1][-1].sum()).backward()
        return x.grad.cpu(), p1.grad.cpu(), p2.grad.cpu()
    got = run(rtol=1e-6, atol=1e-8, adjoint_rtol=(1e-3, 1e-6, 1e-5, 1e-4, 1e-4), adjoint_atol=(1e-4, 1e-8, 1e-7, 1e-6, 1e-6))
    for g, name in zip(got, ("gx", "gp1", "gp2")):
        assert torch.allclose(g, T(z[f"adjtol_{name}"]), rtol=1e-9, atol=1e-12), name
    with pytest.raises(AssertionError, match="tupled rtol"):        # 1 + 2 n_y + P entries is NOT the convention
        run(rtol=1e-6, atol=1e-8, adjoint_rtol=(1e-3, 1e-6, 1e-6, 1e-5, 1e-5, 1e-4, 1e-4))
    with pytest.raises(AssertionError, match="tupled rtol"):        # inherited tuple rtol: length 2, not 3 + P
        run(rtol=(1e-5, 1e-7), atol=(1e-7, 1e-9))
    run(rtol=(1e-5, 1e-7), atol=(1e-7, 1e-9), adjoint_rtol=1e-6, adjoint_atol=1e-8)


@pytest.mark.parametrize("method", ["dopri5", "bosh3"])
@pytest.mark.parametrize("tag", ["rtolvec", "atollist", "both"])
def test_per_element_tolerances(where, tag, method):
    """Tolerances given PER ELEMENT — a tensor or list that broadcasts against the state (in the reference
    `atol + rtol * max(|y0|, |y1|)` simply broadcasts, misc.py:80-82; the tolerances are fp64 tensors by then,
    rk_common.py:186-187).  They used to be rejected ('tupled rtol must have the same length as the tuple y0').  The
    kernels deliver the raw error / initial-step quantities, scaling and norm run as torch ops in fp64: the reference's
    steps exactly (accepted step sizes, evaluation counts, solution)."""
    z = load("dropin.npz")
    kw = {"rtolvec": dict(rtol=torch.tensor([1e-3, 1e-6, 1e-9], dtype=torch.float64), atol=1e-9),
          "atollist": dict(rtol=1e-6, atol=[1e-3, 1e-6, 1e-9]),
          "both": dict(rtol=torch.tensor([1e-3, 1e-6, 1e-7]), atol=[1e-4, 1e-8, 1e-9])}[tag]
    y0 = torch.tensor([[1.0, 2.0, 3.0], [0.5, 1.0, 1.5]], dtype=torch.float64)
    c = torch.tensor([1.0, 5.0, 0.2], dtype=torch.float64)
    t = torch.tensor([0.0, 0.5, 1.0], dtype=torch.float64)
    accepted = []

    class F(torch.nn.Module):
        nfe = 0

        def forward(self, t_, y):
            self.nfe += 1
            return -y * c * (1 + 0.2 * t_)

        def callback_accept_step(self, t0, y_, dt):
            accepted.append(float(dt))
    f = F()
    with torch.no_grad():
        y = tda.odeint(f, y0, t, method=method, **kw)
    key = f"vectol_{tag}_{method}"
    assert f.nfe == int(z[f"{key}_nfe"])
    assert torch.allclose(torch.tensor(accepted, dtype=torch.float64), torch.as_tensor(z[f"{key}_accept_dt"], dtype=torch.float64), rtol=1e-9)
    assert float((y.cpu() - T(z[f"{key}_y"])).abs().max()) < 1e-11


def test_per_element_tolerances_inside_a_tuple_tolerance_with_gradients(where):
    """misc.py:115-123: an entry of a tuple tolerance may itself be a vector over its component."""
    z = load("dropin.npz")
    c = torch.tensor([1.0, 5.0, 0.2], dtype=torch.float64)
    w = torch.tensor(0.7, dtype=torch.float64, requires_grad=True)
    x = torch.tensor([1.0, 2.0, 3.0], dtype=torch.float64, requires_grad=True)
    out = tda.odeint(lambda t_, s: (-s[0] * c * w, -s[1] * 0.3), (x, torch.ones(2, dtype=torch.float64)),
                     torch.tensor([0.0, 0.5, 1.0], dtype=torch.float64),
                     rtol=(torch.tensor([1e-3, 1e-6, 1e-8], dtype=torch.float64), 1e-5),
                     atol=(1e-9, torch.tensor([1e-7, 1e-9], dtype=torch.float64)))
    out[0][-1].pow(2).sum().backward()
    assert torch.allclose(out[0].detach().cpu(), T(z["vectol_tuple_y"]), rtol=1e-12, atol=1e-14)
    assert torch.allclose(x.grad.cpu(), T(z["vectol_tuple_gx"]), rtol=1e-10) and torch.allclose(w.grad.cpu(), T(z["vectol_tuple_gw"]), rtol=1e-10)
