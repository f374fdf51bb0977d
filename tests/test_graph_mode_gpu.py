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


@pytest.fixture(autouse=True)
def _no_process_wide_graph_default(monkeypatch):
    """These tests choose the step path per call (`options={'hip_graph': ...}`) and assert on cache entries and Python
    call counts: the process-wide default ("auto" since r06; tests/test_graph_default_gpu.py covers it) must not turn
    their eager baselines into captured solves."""
    monkeypatch.setenv("TDEQ_HIP_GRAPH", "0")


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


@pytest.mark.parametrize("method", ["rk4", "euler", "midpoint", "heun2", "heun3"])
@pytest.mark.parametrize("state_dtype", [torch.float32, torch.float64], ids=["f32", "f64"])
@pytest.mark.parametrize("grid_dtype", [torch.float32, torch.float64], ids=["t32", "t64"])
@pytest.mark.parametrize("perturb", [False, True])
@pytest.mark.parametrize("reverse", [False, True])
def test_graph_mode_time_dependent_field(method, state_dtype, grid_dtype, perturb, reverse):
    g = torch.Generator().manual_seed(4)
    y0 = torch.randn(257, 3, generator=g, dtype=torch.float64).to(state_dtype).cuda()
    w = torch.randn(3, 3, generator=g, dtype=torch.float64).to(state_dtype).cuda() * 0.3
    f = lambda t, y: torch.tanh(y @ w) * torch.cos(3 * t) - 0.1 * y * t
    t = torch.linspace(0.3, 2.1, 37, dtype=grid_dtype, device="cuda")
    if reverse:
        t = t.flip(0)
    opts = dict(perturb=perturb)
    with torch.no_grad(), warnings.catch_warnings():
        warnings.simplefilter("error")            # in particular: no "running the eager path" fallback warning
        y_eager = tda.odeint(f, y0, t, method=method, options=dict(opts))
        y_graph = tda.odeint(f, y0, t, method=method, options=dict(opts, hip_graph=True))
    assert torch.equal(y_graph, y_eager)


@pytest.mark.parametrize("method,n_eval", [("euler", 1), ("midpoint", 2), ("heun2", 2), ("heun3", 3), ("rk4", 4)])
def test_graph_mode_of_every_explicit_fixed_grid_method_replays(method, n_eval):
    """r03: the captured step exists for every explicit Runge-Kutta fixed-grid method, not only rk4: same bits as the
    eager path, func runs in Python for the first step and the capture only, the evaluation count is the eager one."""
    A = torch.tensor([[-0.1, 2.0], [-2.0, -0.1]], device="cuda")
    calls = [0]

    def f(t, y):
        calls[0] += 1
        return torch.tanh(y @ A) * torch.cos(t)
    y0 = torch.tensor([[2.0, 0.0]], device="cuda")
    t = torch.linspace(0.0, 5.0, 400, device="cuda")
    with torch.no_grad():
        y_eager = tda.odeint(f, y0, t, method=method)
        assert calls[0] == n_eval * 399
        calls[0] = 0
        ci = check_inputs(f, y0, t, 1e-7, 1e-9, method, dict(hip_graph=True), None, SOLVERS)
        solver = SOLVERS[method](func=ci.func, y0=ci.y0_flat, rtol=ci.rtol, atol=ci.atol, **ci.options)
        y_graph = solver.integrate(ci.t)
    assert torch.isfinite(y_eager).all() and torch.equal(y_graph.view_as(y_eager), y_eager)
    assert calls[0] == 2 * n_eval              # the eager first step + the captured one
    assert ci.func.nfe == n_eval * 399         # ... while the solver's count is the eager path's


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


# ---------------------------------------------------------------------------------------------------------------
# adaptive solvers: one captured trial step
# ---------------------------------------------------------------------------------------------------------------
class _Count:
    def __init__(self, fn):
        self.fn, self.n = fn, 0

    def __call__(self, t, y):
        self.n += 1
        return self.fn(t, y)


def _vdp(t, y):
    x, v = y[..., 0], y[..., 1]
    return torch.stack([v, 3.0 * (1 - x * x) * v - x], dim=-1)


ADAPTIVE = [
    ("dopri5", dict(rtol=1e-6, atol=1e-8), [0.0, 1.5, 4.0]),
    ("dopri5", dict(rtol=1e-5, atol=1e-7, options=dict(first_step=0.9)), [0.0, 3.0, 7.0]),      # rejections
    ("dopri5", dict(rtol=1e-6, atol=1e-8, field="decay"), [2.0, 1.0, -0.5]),                      # decreasing time
    ("dopri8", dict(rtol=1e-8, atol=1e-10), [0.0, 5.0]),
    ("tsit5", dict(rtol=1e-6, atol=1e-8), [0.0, 2.0, 4.0]),                                       # non-FSAL
    ("bosh3", dict(rtol=1e-4, atol=1e-6), [0.0, 4.0]),
    ("adaptive_heun", dict(rtol=1e-3, atol=1e-5), [0.0, 0.7]),
    ("dopri5", dict(rtol=1e-6, atol=1e-8, options=dict(min_step=1e-3, max_step=0.05)), [0.0, 2.0]),
]


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32], ids=["f64", "f32"])
@pytest.mark.parametrize("case", ADAPTIVE, ids=[f"{m}-{i}" for i, (m, _, _) in enumerate(ADAPTIVE)])
def test_adaptive_graph_mode_equals_eager(case, dtype, monkeypatch):
    """Same accepted / rejected sequence (equal NFE) and the same solution as the eager device-controller path
    (look-ahead); with the controller on the device in both, results are bit-identical."""
    method, kw, ts = case
    kw = dict(kw)
    opts = dict(kw.pop("options", {}))
    field = _vdp if kw.pop("field", "vdp") == "vdp" else (lambda t, y: -0.7 * y + torch.sin(2 * t))
    y0 = torch.tensor([[2.0, 0.0], [1.0, -1.0], [0.5, 0.5]], dtype=dtype, device="cuda")
    t = torch.tensor(ts, dtype=torch.float64, device="cuda")
    f_e, f_g = _Count(field), _Count(field)
    with torch.no_grad():
        y_eager = tda.odeint(f_e, y0, t, method=method, options=dict(opts), **kw)
        y_graph = tda.odeint(f_g, y0, t, method=method, options=dict(opts, hip_graph=True), **kw)
    assert torch.equal(y_graph, y_eager)
    assert f_g.n <= f_e.n       # Python-side counter: the graph replays do not call into Python ...
    # ... the solver's own evaluation count does follow them
    ci = check_inputs(field, y0, t, kw["rtol"], kw["atol"], method, dict(opts, hip_graph=True), None, SOLVERS)
    solver = SOLVERS[method](func=ci.func, y0=ci.y0_flat, rtol=ci.rtol, atol=ci.atol, **ci.options)
    with torch.no_grad():
        solver.integrate(ci.t)
    assert ci.func.nfe == f_e.n
    assert solver._g is not None and (solver._g.graphs[0] is not None or solver._g.calls == 1)


def test_adaptive_graph_mode_many_outputs_and_tuple_state():
    lin = torch.nn.Linear(4, 4).double().cuda()

    def f(t, y):
        a, b = y
        return torch.tanh(lin(a)) * torch.cos(t), -b * t

    g = torch.Generator().manual_seed(0)
    y0 = (torch.randn(8, 4, generator=g, dtype=torch.float64).cuda(), torch.randn(3, generator=g, dtype=torch.float64).cuda())
    t = torch.linspace(0.0, 2.0, 41, dtype=torch.float64, device="cuda")
    with torch.no_grad():
        ya, yb = tda.odeint(f, y0, t, rtol=1e-7, atol=1e-9, method="dopri5")
        ga, gb = tda.odeint(f, y0, t, rtol=1e-7, atol=1e-9, method="dopri5", options=dict(hip_graph=True))
    assert torch.equal(ga, ya) and torch.equal(gb, yb)


def _adjoint_case(kind):
    torch.manual_seed(0)
    if kind == "mlp":
        lin1, lin2 = torch.nn.Linear(8, 24).double().cuda(), torch.nn.Linear(24, 8).double().cuda()

        class F(torch.nn.Module):
            def __init__(self):
                super().__init__()
                self.a, self.b = lin1, lin2

            def forward(self, t, y):
                return self.b(torch.tanh(self.a(y))) * torch.cos(t)        # time-dependent: vjp_t is live
        y0 = torch.randn(16, 8, dtype=torch.float64, device="cuda")
        return F(), y0, lambda y: y[-1].pow(2).sum() + y[1].sum()
    from _fullsize import ExampleCNF, load
    z = load("cfg5")
    cnf = ExampleCNF([z[f"p{i}"] for i in range(6)], trace="closed").double().cuda()
    z0 = torch.randn(64, 2, dtype=torch.float64, device="cuda")
    return cnf, (z0, torch.zeros(64, 1, dtype=torch.float64, device="cuda")), \
        lambda out: out[1][-1].mean() - out[0][-1].pow(2).sum() / 100


@pytest.mark.parametrize("kind", ["mlp", "cnf_tuple"])
@pytest.mark.parametrize("reverse", [False, True])
def test_adjoint_backward_solve_captured_equals_eager(kind, reverse):
    """r02: `hip_graph` reaches the adjoint's BACKWARD solve — one trial step of the augmented system (S evaluations
    of func with torch.autograd.grad behind each, the segment packing, the combines, the 9–11-segment norm +
    controller) is captured from inside the autograd engine's thread and replayed.  Same decisions, same gradients
    as the eager backward solve, over several output intervals and across repeated backward passes (the captured
    step is reused), with no warning."""
    from torchdiffeq_amd.solvers import _GraphStep
    tda.clear_graph_cache()
    f, y0, loss_fn = _adjoint_case(kind)
    ts = [0.0, 0.4, 1.0]
    t = torch.tensor(ts[::-1] if reverse else ts, dtype=torch.float64, device="cuda")
    params = list(f.parameters())

    def run(opts, aopts):
        for p in params:
            p.grad = None
        if isinstance(y0, tuple):
            x = tuple(v.clone().requires_grad_(i == 0) for i, v in enumerate(y0))
            leaf = x[0]
        else:
            x = leaf = y0.clone().requires_grad_(True)
        with warnings.catch_warnings(record=True) as rec:
            warnings.simplefilter("always")
            out = tda.odeint_adjoint(f, x, t, rtol=1e-7, atol=1e-9, method="dopri5", options=opts, adjoint_options=aopts)
            loss_fn(out).backward()
        assert not [w for w in rec if "hip_graph" in str(w.message)], [str(w.message) for w in rec]
        return [leaf.grad.clone()] + [p.grad.clone() for p in params]

    eager = run(None, None)
    for rep in range(3):                      # first pass captures, later passes reuse the captured step
        captured = run(dict(hip_graph=True), None)          # inherited by the adjoint options, like every option
        for a, b in zip(captured, eager):
            assert torch.equal(a, b), (rep, float((a - b).abs().max()))
    per_func = _GraphStep._cache.get(f)
    assert per_func is not None and len(per_func) == 2      # one captured step for the forward, one for the backward
    assert all(g.graphs[0] is not None for g in per_func.values())
    explicit = run(None, dict(hip_graph=True))              # backward only
    for a, b in zip(explicit, eager):
        assert torch.equal(a, b)
    tda.clear_graph_cache()


def test_backward_solve_is_captured_for_a_closure_field_with_explicit_adjoint_params():
    """r05: `func` need not be an nn.Module that owns its parameters (torchdiffeq/_impl/adjoint.py:161-164 accepts any
    callable with explicit `adjoint_params`).  A closure over two weight tensors — one of them also reached through a
    list: the captured backward step evaluates func under `adjoint._AliasParams`, which hands leaf aliases of the
    parameters to every torch call, so the VJPs stay off the parameters' own AccumulateGrad nodes.  Gradients are
    bit-identical to the eager backward solve's, and the backward solve really is replayed."""
    from torchdiffeq_amd import _graph
    tda.clear_graph_cache()
    torch.manual_seed(0)
    W1 = (torch.randn(6, 16, dtype=torch.float64, device="cuda") / 3).requires_grad_(True)
    W2 = (torch.randn(16, 6, dtype=torch.float64, device="cuda") / 3).requires_grad_(True)
    held = [W2]

    def field(t, y):
        return torch.tanh(y @ W1) @ held[0] * torch.cos(t)

    y0 = torch.randn(32, 6, dtype=torch.float64, device="cuda")
    t = torch.tensor([0.0, 0.7, 1.5], dtype=torch.float64, device="cuda")
    kw = dict(method="dopri5", rtol=1e-7, atol=1e-9, adjoint_params=(W1, W2))

    def run(adjoint_options):
        W1.grad = W2.grad = None
        x = y0.clone().requires_grad_(True)
        y = tda.odeint_adjoint(field, x, t, adjoint_options=adjoint_options, **kw)
        (y[-1].pow(2).sum() + y[1].sum()).backward()
        return y.detach().clone(), x.grad.clone(), W1.grad.clone(), W2.grad.clone()

    eager = run(None)
    replays = [0]
    real_replay = torch.cuda.CUDAGraph.replay

    def counting_replay(self):
        replays[0] += 1
        return real_replay(self)
    torch.cuda.CUDAGraph.replay = counting_replay
    try:
        with warnings.catch_warnings():
            warnings.simplefilter("error")                   # in particular: no "running it eagerly"
            captured = run(dict(hip_graph=True))
    finally:
        torch.cuda.CUDAGraph.replay = real_replay
    assert replays[0] > 0
    for a, b in zip(captured, eager):
        assert torch.equal(a, b), float((a - b).abs().max())
    assert float(captured[2].abs().max()) > 0 and float(captured[3].abs().max()) > 0
    # a pre-computed VIEW of a parameter cannot be re-routed: found out by the probe, eager with a warning, same numbers
    W2t = W2.t()

    def field_view(t, y):
        return torch.tanh(y @ W1) @ W2t.t() * torch.cos(t)
    W1.grad = W2.grad = None
    x = y0.clone().requires_grad_(True)
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        y = tda.odeint_adjoint(field_view, x, t, adjoint_options=dict(hip_graph=True), **kw)
        (y[-1].pow(2).sum() + y[1].sum()).backward()
    assert any("cannot be re-routed" in str(w.message) for w in rec)
    assert torch.equal(W2.grad, eager[3]) and torch.equal(x.grad, eager[1])
    tda.clear_graph_cache()


def test_uncapturable_func_falls_back_to_eager():
    """A func that synchronises with the host cannot be captured: the solver warns and finishes the solve eagerly,
    with the same result."""
    def f(t, y):
        scale = float(t)              # device -> host read: illegal while a stream is capturing
        return -y * (1.0 + 0.1 * scale)

    y0 = torch.tensor([[1.0, 2.0, 3.0]], dtype=torch.float64, device="cuda")
    t = torch.tensor([0.0, 1.0, 2.0], dtype=torch.float64, device="cuda")
    with torch.no_grad():
        ref = tda.odeint(f, y0, t, method="dopri5", rtol=1e-7, atol=1e-9)
        with warnings.catch_warnings(record=True) as rec:
            warnings.simplefilter("always")
            y = tda.odeint(f, y0, t, method="dopri5", rtol=1e-7, atol=1e-9, options=dict(hip_graph=True))
    assert any("could not be captured" in str(w.message) for w in rec)
    assert torch.equal(y, ref)
    tg = torch.linspace(0.0, 1.0, 9, dtype=torch.float64, device="cuda")
    with torch.no_grad():
        ref4 = tda.odeint(f, y0, tg, method="rk4")
        with warnings.catch_warnings(record=True) as rec:
            warnings.simplefilter("always")
            y4 = tda.odeint(f, y0, tg, method="rk4", options=dict(hip_graph=True))
    assert any("could not be captured" in str(w.message) for w in rec)
    assert torch.equal(y4, ref4)
    # the device is still usable
    assert float((y0 * 2).sum()) == 12.0


def test_auto_mode_remembers_a_func_that_cannot_be_captured():
    """hip_graph='auto' on a func with a host read inside (advisor r04): the first solve that tries the capture records
    the func as refused — ONE warning worded for auto, look-ahead back on for the rest of that solve — and later solves
    of the same func object do not try again (no capture attempt, no warning), all with the eager result."""
    from torchdiffeq_amd.solvers import _GraphStep
    tda.clear_graph_cache()

    class F(torch.nn.Module):
        def forward(self, t, y):
            scale = float(t)              # device -> host read: illegal while a stream is capturing
            return -y * (1.0 + 0.1 * scale)

    f = F()
    y0 = torch.tensor([[1.0, 2.0, 3.0]], dtype=torch.float64, device="cuda")
    t = torch.tensor([0.0, 1.0, 2.0], dtype=torch.float64, device="cuda")
    with torch.no_grad():
        ref = tda.odeint(f, y0, t, method="dopri5", rtol=1e-7, atol=1e-9)
        msgs = []
        for rep in range(4):             # 1st: eager ("later"), 2nd: capture attempt fails -> refused, 3rd/4th: never
            with warnings.catch_warnings(record=True) as rec:
                warnings.simplefilter("always")
                y = tda.odeint(f, y0, t, method="dopri5", rtol=1e-7, atol=1e-9, options=dict(hip_graph="auto"))
            assert torch.equal(y, ref), rep
            msgs.append([str(w.message) for w in rec if "hip_graph" in str(w.message)])
    flat = [m for ms in msgs for m in ms]
    assert len(flat) == 1 and "hip_graph='auto'" in flat[0] and "capturing it failed" in flat[0], msgs
    assert f in _GraphStep._refused and not _GraphStep._cache.get(f)
    assert msgs[2] == [] and msgs[3] == []
    tda.clear_graph_cache()


def test_graph_is_reused_across_solves_of_the_same_func():
    """Second and later solves with the same func object and state layout replay the graph captured by the first
    (no capture, no eager warm-up step) with the new initial state — and see in-place parameter updates."""
    from torchdiffeq_amd.solvers import _GraphStep
    tda.clear_graph_cache()
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(6, 16), torch.nn.Tanh(), torch.nn.Linear(16, 6)).double().cuda()

    class F(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.net = net

        def forward(self, t, y):
            return self.net(y) * torch.cos(t)

    f = F()
    t = torch.tensor([0.0, 0.6, 1.5], dtype=torch.float64, device="cuda")
    kw = dict(method="dopri5", rtol=1e-7, atol=1e-9)
    graphs = []
    for rep in range(4):
        y0 = torch.randn(32, 6, dtype=torch.float64, device="cuda")
        if rep == 2:
            with torch.no_grad():
                net[0].weight.mul_(1.1)          # an optimizer step between solves
        with torch.no_grad():
            ref = tda.odeint(f, y0, t, **kw)
            y = tda.odeint(f, y0, t, options=dict(hip_graph=True), **kw)
        assert torch.equal(y, ref), rep
        per_func = _GraphStep._cache.get(f)
        assert per_func is not None and len(per_func) == 1
        g = next(iter(per_func.values()))
        graphs.append((g, g.graphs[0], g.calls))
        assert not g.in_use
    assert all(a[0] is graphs[0][0] and a[1] is graphs[0][1] for a in graphs)      # one capture, reused
    assert graphs[-1][2] > graphs[0][2]
    # a different state layout gets its own graph
    with torch.no_grad():
        y0 = torch.randn(7, 6, dtype=torch.float64, device="cuda")
        assert torch.equal(tda.odeint(f, y0, t, options=dict(hip_graph=True), **kw), tda.odeint(f, y0, t, **kw))
    assert len(_GraphStep._cache.get(f)) == 2
    tda.clear_graph_cache()
    assert _GraphStep._cache.get(f) is None


def test_graph_cache_notices_reallocated_parameters():
    """In-place parameter updates are seen by the captured graph; parameters that move to new storage (here: a
    dtype round trip) lead to a fresh capture instead of a graph reading freed memory."""
    from torchdiffeq_amd.solvers import _GraphStep
    tda.clear_graph_cache()
    torch.manual_seed(1)

    class F(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.lin = torch.nn.Linear(5, 5).double().cuda()

        def forward(self, t, y):
            return torch.tanh(self.lin(y))

    f = F()
    y0 = torch.randn(9, 5, dtype=torch.float64, device="cuda")
    t = torch.tensor([0.0, 1.0], dtype=torch.float64, device="cuda")
    kw = dict(method="dopri5", rtol=1e-7, atol=1e-9)

    def both():
        with torch.no_grad():
            return tda.odeint(f, y0, t, options=dict(hip_graph=True), **kw), tda.odeint(f, y0, t, **kw)

    yg, ye = both()
    assert torch.equal(yg, ye)
    first = next(iter(_GraphStep._cache.get(f).values()))
    f.float().double()                      # same values, new storages
    with torch.no_grad():
        f.lin.weight.mul_(0.5)
    yg, ye = both()
    assert torch.equal(yg, ye)
    entries = list(_GraphStep._cache.get(f).values())
    assert len(entries) == 2 and entries[-1] is not first
    for _ in range(6):                      # the cache stays bounded
        f.float().double()
        yg, ye = both()
        assert torch.equal(yg, ye)
    assert len(_GraphStep._cache.get(f)) <= _GraphStep._MAX_PER_FUNC
    tda.clear_graph_cache()


def test_graph_cache_notices_rebound_closure_tensors_and_plain_attributes():
    """the captured-step cache must not replay a graph that reads tensors the func no longer uses.  A
    closure whose variable is bound to a NEW tensor, a Module with a plain (unregistered) tensor attribute that is
    re-assigned, and a user-bumped `hip_graph_token` each lead to a fresh capture; in-place updates do not need one."""
    tda.clear_graph_cache()
    torch.manual_seed(2)
    y0 = torch.randn(7, 4, dtype=torch.float64, device="cuda")
    t = torch.tensor([0.0, 1.0], dtype=torch.float64, device="cuda")
    kw = dict(method="dopri5", rtol=1e-7, atol=1e-9)
    box = {"A": torch.randn(4, 4, dtype=torch.float64, device="cuda") * 0.3}

    def make():
        A = box["A"]
        return lambda tt, y: torch.tanh(y @ A)

    class Plain(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.W = box["A"].clone()        # plain attribute: not a Parameter, not a buffer

        def forward(self, tt, y):
            return torch.tanh(y @ self.W)

    mod = Plain()
    with torch.no_grad():
        for step in range(3):
            f = make()
            yg = tda.odeint(f, y0, t, options=dict(hip_graph=True), **kw)
            assert torch.equal(yg, tda.odeint(f, y0, t, **kw))
            ym = tda.odeint(mod, y0, t, options=dict(hip_graph=True), **kw)
            assert torch.equal(ym, tda.odeint(mod, y0, t, **kw))
            box["A"] = torch.randn(4, 4, dtype=torch.float64, device="cuda") * 0.3      # NEW storage
            mod.W = box["A"].clone()
        # in-place update of a held tensor: same storage, the replayed graph sees the new values
        f = make()
        y1 = tda.odeint(f, y0, t, options=dict(hip_graph=True), **kw)
        box["A"].mul_(0.5)
        y2 = tda.odeint(f, y0, t, options=dict(hip_graph=True), **kw)
        assert torch.equal(y2, tda.odeint(f, y0, t, **kw)) and not torch.equal(y1, y2)
        # the token: same object, same storages, but the user says it computes something else now
        mode = {"k": 1.0}

        def g(tt, y):
            return torch.tanh(y @ box["A"]) * mode["k"]
        g.hip_graph_token = 0
        ya = tda.odeint(g, y0, t, options=dict(hip_graph=True), **kw)
        mode["k"] = -1.0
        g.hip_graph_token = 1
        yb = tda.odeint(g, y0, t, options=dict(hip_graph=True), **kw)
        assert torch.equal(yb, tda.odeint(g, y0, t, **kw)) and not torch.equal(ya, yb)
    tda.clear_graph_cache()


# ---------------------------------------------------------------------------------------------------------------------
# hip_graph="auto" (r04): capture lazily, and only what is verifiably safe to replay
# ---------------------------------------------------------------------------------------------------------------------
class _PureField(torch.nn.Module):
    """A pure function of (t, y) and its parameters.  Python calls are counted in a hook's closure — outside the
    module's attributes, so that counting them is not itself a visible side effect; `f.calls()` returns the counter."""

    def __init__(self, d=8):
        super().__init__()
        torch.manual_seed(3)
        self.lin = torch.nn.Linear(d, d).cuda()
        calls = [0]
        self.register_forward_pre_hook(lambda m, a: calls.__setitem__(0, calls[0] + 1))
        self.calls = lambda: calls

    def forward(self, t, y):
        return torch.tanh(self.lin(y)) * torch.cos(t) - 0.3 * y


def _auto_problem(n=64, d=8):
    g = torch.Generator().manual_seed(5)
    return torch.randn(n, d, generator=g).cuda(), torch.tensor([0.0, 0.5, 1.0], device="cuda")


def test_auto_mode_captures_a_pure_func_from_its_second_solve_on():
    from torchdiffeq_amd.solvers import _GraphStep
    y0, t = _auto_problem()
    f = _PureField()
    calls = f.calls()
    with torch.no_grad():
        y_eager = tda.odeint(f, y0, t, method="dopri5", rtol=1e-6, atol=1e-8)
        nfe = calls[0]
        calls[0] = 0
        with warnings.catch_warnings():
            warnings.simplefilter("error")
            y1 = tda.odeint(f, y0, t, method="dopri5", rtol=1e-6, atol=1e-8, options=dict(hip_graph="auto"))
            assert calls[0] == nfe                    # first sight of (func, layout): a short solve stays eager
            calls[0] = 0
            y2 = tda.odeint(f, y0, t, method="dopri5", rtol=1e-6, atol=1e-8, options=dict(hip_graph="auto"))
            second = calls[0]
            calls[0] = 0
            y3 = tda.odeint(f, y0, t, method="dopri5", rtol=1e-6, atol=1e-8, options=dict(hip_graph="auto"))
            third = calls[0]
    assert torch.equal(y1, y_eager) and torch.equal(y2, y_eager) and torch.equal(y3, y_eager)
    # second solve: warm-up step + capture + the probe's eager twin run in Python, everything else is replayed
    assert second < nfe or nfe <= 2 + 6 * 4, (second, nfe)
    assert third <= 2 + 6                                 # third solve: the two initial evaluations + (at most) a side-1 capture
    assert f not in _GraphStep._refused


class _CountingField(torch.nn.Module):
    def __init__(self, d=8):
        super().__init__()
        torch.manual_seed(3)
        self.lin = torch.nn.Linear(d, d).cuda()
        self.nfe = 0

    def forward(self, t, y):
        self.nfe += 1                                   # the classic evaluation counter of the reference's examples
        return torch.tanh(self.lin(y)) * torch.cos(t) - 0.3 * y


def test_auto_mode_refuses_a_func_with_an_evaluation_counter():
    """A captured func runs in Python only while the graph is built: `self.nfe += 1` would stop counting.  "auto" sees
    the attribute change during the first eager step, warns ONCE, and keeps every solve of that func eager — counts and
    solution exactly the eager ones."""
    y0, t = _auto_problem()
    f = _CountingField()
    with torch.no_grad():
        y_eager = tda.odeint(f, y0, t, method="dopri5", rtol=1e-6, atol=1e-8)
        nfe = f.nfe
        counts, ys = [], []
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            for _ in range(3):
                f.nfe = 0
                ys.append(tda.odeint(f, y0, t, method="dopri5", rtol=1e-6, atol=1e-8, options=dict(hip_graph="auto")))
                counts.append(f.nfe)
    assert counts == [nfe, nfe, nfe]
    assert all(torch.equal(y, y_eager) for y in ys)
    msgs = [str(x.message) for x in w if "hip_graph='auto'" in str(x.message)]
    assert len(msgs) == 1 and "changed its own attributes" in msgs[0]


def test_auto_mode_refuses_random_fields():
    y0, t = _auto_problem()
    lin = torch.nn.Linear(8, 8).cuda()
    f = lambda t_, y_: lin(y_) * 0.1 + 1e-3 * torch.rand_like(y_)
    with torch.no_grad(), warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        for _ in range(3):
            tda.odeint(f, y0, t, method="dopri5", rtol=1e-3, atol=1e-4, options=dict(hip_graph="auto"))
    msgs = [str(x.message) for x in w if "hip_graph='auto'" in str(x.message)]
    assert len(msgs) == 1 and "random-number" in msgs[0]


def test_auto_mode_probe_catches_what_the_fingerprint_cannot_see():
    """A func whose value depends on something no attribute, buffer or RNG offset shows — here: whether a stream capture
    is in progress — passes the fingerprint; the replay-vs-eager comparison of the first captured step does not: the
    solve stays correct (the eager results are kept), warns once, and the func is never captured again."""
    y0, t = _auto_problem()
    lin = torch.nn.Linear(8, 8).cuda()

    def f(t_, y_):
        return lin(y_) * (0.2 if torch.cuda.is_current_stream_capturing() else 0.1)
    outs = []
    with torch.no_grad(), warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        for mode in (False, "auto", "auto", "auto"):
            outs.append(tda.odeint(f, y0, t, method="dopri5", rtol=1e-6, atol=1e-8, options=dict(hip_graph=mode)))
    assert all(torch.equal(o, outs[0]) for o in outs[1:])
    msgs = [str(x.message) for x in w if "hip_graph='auto'" in str(x.message)]
    assert len(msgs) == 1 and "bit for bit" in msgs[0]


def test_auto_mode_switches_to_replays_inside_one_long_solve():
    """First sight of a func, but the solve is long: after _AUTO_CAPTURE_AFTER_STEPS eager trial steps the rest is
    replayed — same bits as the eager solve, most evaluations without Python."""
    from torchdiffeq_amd import solvers
    y0, _ = _auto_problem()
    t = torch.linspace(0.0, 40.0, 5, device="cuda")
    f = _PureField()
    calls = f.calls()
    with torch.no_grad():
        y_eager = tda.odeint(f, y0, t, method="dopri5", rtol=1e-7, atol=1e-9)
        nfe = calls[0]
        calls[0] = 0
        with warnings.catch_warnings():
            warnings.simplefilter("error")
            y_auto = tda.odeint(f, y0, t, method="dopri5", rtol=1e-7, atol=1e-9, options=dict(hip_graph="auto"))
    steps = (nfe - 2) // 6
    assert steps > solvers._AUTO_CAPTURE_AFTER_STEPS + 20, steps
    assert torch.equal(y_auto, y_eager)
    assert calls[0] <= 2 + 6 * (solvers._AUTO_CAPTURE_AFTER_STEPS + 4), (calls[0], nfe)


def test_auto_mode_on_fixed_grids():
    """rk4 under "auto": a counting func stays eager (one warning), a pure one is replayed; short grids are not captured."""
    A = torch.tensor([[-0.1, 2.0], [-2.0, -0.1]], device="cuda")
    y0 = torch.tensor([[2.0, 0.0]], device="cuda")
    t = torch.linspace(0.0, 5.0, 200, device="cuda")

    class Counting(torch.nn.Module):
        nfe = 0

        def forward(self, t_, y_):
            self.nfe += 1
            return (y_ ** 3) @ A
    f = Counting()
    with torch.no_grad():
        y_eager = tda.odeint(f, y0, t, method="rk4")
        n = f.nfe
        f.nfe = 0
        with pytest.warns(UserWarning, match="hip_graph='auto'"):
            y_auto = tda.odeint(f, y0, t, method="rk4", options=dict(hip_graph="auto"))
    assert f.nfe == n and torch.equal(y_auto, y_eager)
    calls = [0]
    pure = torch.nn.Identity()
    pure.register_forward_pre_hook(lambda m, a: calls.__setitem__(0, calls[0] + 1))
    g = lambda t_, y_: (pure(y_) ** 3) @ A
    with torch.no_grad(), warnings.catch_warnings():
        warnings.simplefilter("error")
        y_auto = tda.odeint(g, y0, t, method="rk4", options=dict(hip_graph="auto"))
        long_calls, calls[0] = calls[0], 0
        y_short = tda.odeint(g, y0, t[:10], method="rk4", options=dict(hip_graph="auto"))
        short_calls = calls[0]
    assert torch.equal(y_auto, y_eager) and torch.equal(y_short, y_eager[:10])
    assert long_calls == 8 and short_calls == 4 * 9          # replayed (first step + capture) / too short: eager


def test_auto_mode_recaptures_when_a_python_scalar_of_func_changes():
    """A captured graph bakes Python numbers into kernel arguments.  Under "auto" the plain attributes of func are part
    of the cache key, so `f.scale = ...` between two solves gives the new value's solution, not a stale replay."""
    y0, t = _auto_problem()

    class Scaled(torch.nn.Module):
        def __init__(self):
            super().__init__()
            torch.manual_seed(3)
            self.lin = torch.nn.Linear(8, 8).cuda()
            self.scale = 0.5

        def forward(self, t_, y_):
            return torch.tanh(self.lin(y_)) * self.scale
    f = Scaled()
    with torch.no_grad(), warnings.catch_warnings():
        warnings.simplefilter("error")
        for scale in (0.5, 0.5, 0.5, 1.5, 1.5, 1.5):
            f.scale = scale
            y_auto = tda.odeint(f, y0, t, method="dopri5", rtol=1e-6, atol=1e-8, options=dict(hip_graph="auto"))
            y_eager = tda.odeint(f, y0, t, method="dopri5", rtol=1e-6, atol=1e-8)
            assert torch.equal(y_auto, y_eager), scale


def test_auto_mode_never_adds_evaluations_a_counting_func_could_see():
    """`odeint_adjoint(..., options={'hip_graph': 'auto'})` with a field that counts its evaluations (every example of the
    reference does): forward and backward counts of every iteration of a training loop equal the eager ones — in
    particular the backward solve's proxy check (two evaluations of func, adjoint._AugmentedDynamics.proxy_is_faithful)
    is not run while nothing can be captured (first sight / refused func)."""
    y0, t = _auto_problem()
    f = _CountingField()

    def one(options):
        x = y0.clone().requires_grad_(True)
        f.zero_grad()
        f.nfe = 0
        y = tda.odeint_adjoint(f, x, t, method="dopri5", rtol=1e-6, atol=1e-8, options=options)
        fwd, f.nfe = f.nfe, 0
        y[-1].pow(2).sum().backward()
        return fwd, f.nfe, x.grad.clone(), f.lin.weight.grad.clone()
    ref = one(None)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for _ in range(4):
            got = one(dict(hip_graph="auto"))
            assert got[:2] == ref[:2], (got[:2], ref[:2])
            assert torch.equal(got[2], ref[2]) and torch.equal(got[3], ref[3])


def test_auto_mode_adjoint_loop_backward_is_captured_at_its_second_sight():
    """Iteration 1 of a training loop: forward and backward eager (Python calls = the eager count).  Iteration 2: both
    captured (adjoint: proxy check + capture).  Iteration 3: replays only — a handful of Python calls; gradients agree
    with the eager ones to the last bit in every iteration."""
    y0, t = _auto_problem()
    f = _PureField()
    calls = f.calls()

    def one(options):
        x = y0.clone().requires_grad_(True)
        f.zero_grad()
        calls[0] = 0
        y = tda.odeint_adjoint(f, x, t, method="dopri5", rtol=1e-6, atol=1e-8, options=options)
        y[-1].pow(2).sum().backward()
        return calls[0], x.grad.clone(), f.lin.weight.grad.clone()
    n_eager, gx, gw = one(None)
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        counts = []
        for _ in range(4):
            n, gx_a, gw_a = one(dict(hip_graph="auto"))
            counts.append(n)
            assert torch.equal(gx_a, gx) and torch.equal(gw_a, gw)
    assert counts[0] == n_eager, (counts, n_eager)
    assert counts[2] < n_eager // 2 and counts[3] <= counts[2], (counts, n_eager)
