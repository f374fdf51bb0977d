"""Per-element tolerances on the fused norm kernel (r05, tdeq_error_norm_vec): `rtol` / `atol` tensors that broadcast
against the state (torchdiffeq/_impl/misc.py:80-82: plain broadcasting in the reference) go into the error-norm launch
as two more fp64 streams instead of a materialised raw error + ~10 ATen passes in fp64."""
import math
import os
import sys

import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
pytestmark = pytest.mark.gpu

import torchdiffeq_amd as tda  # noqa: E402
from torchdiffeq_amd import _native  # noqa: E402

COEFS = (0.0371, -0.211, 0.5, 1.25, -0.0625, 0.33, 0.9)


@pytest.mark.parametrize("form", ["both", "rtol_only", "atol_only"])
@pytest.mark.parametrize("n", [1, 1031, 5000])
@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
def test_kernel_equals_the_references_broadcast_expression(dtype, n, form):
    """misc.py:81-82 evaluated literally by ATen on the same device — type promotion included (a dimensioned fp64
    tolerance promotes, a 0-dim one does not) — against the kernel's fp64 sum of squares."""
    g = torch.Generator().manual_seed(n)
    r = lambda: torch.randn(n, generator=g, dtype=torch.float64).to(dtype).cuda()
    y0, y1, ks = r(), r(), [r() for _ in range(6)]
    rtol_v = (torch.rand(n, generator=g, dtype=torch.float64) * 1e-3 + 1e-6).cuda()
    atol_v = (torch.rand(n, generator=g, dtype=torch.float64) * 1e-5 + 1e-8).cuda()
    rtol = rtol_v if form != "atol_only" else torch.tensor(3e-4, dtype=torch.float64, device="cuda")
    atol = atol_v if form != "rtol_only" else torch.tensor(2e-6, dtype=torch.float64, device="cuda")
    kern = _native.get_kernels(torch.device("cuda:0"), dtype)
    plan = kern.make_plan([(0, n, 0.0, 1.0)], n, 1024, torch.device("cuda:0"))
    dt = 0.0371
    kern.error_norm_vec(plan, y0, y1, ks, COEFS[:6], dt, rtol if rtol.dim() else float(rtol), atol if atol.dim() else float(atol))
    sumsq, _, bad = kern.read_norms(plan)
    # the reference's expression; the error row in the kernels' order (left to right, every product and sum rounded in T)
    dtT = torch.tensor(dt, dtype=torch.float64).to(dtype)
    err = None
    for k_, c in zip(ks, COEFS):
        p = k_ * (torch.tensor(c, dtype=torch.float64).to(dtype) * dtT).cuda()
        err = p if err is None else err + p
    tol = atol + rtol * torch.max(y0.abs(), y1.abs())
    q = err / tol
    assert q.dtype == torch.float64
    assert sumsq[0] == pytest.approx(float(q.abs().pow(2).sum()), rel=1e-12) and bad == [0.0]


@pytest.mark.parametrize("n_lead", [6, 5, 4])
@pytest.mark.parametrize("n", [1031, 5000])
@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
def test_kernel_continues_a_partial_error_row(dtype, n, n_lead):
    """ABI 20: `err_partial` — the leading run of the error row as tdeq_stage_combine_err sums it — continued by the vector-
    tolerance kernel over the remaining 0 / 1 / 2 stages: the same left-to-right sum, so the per-segment sums equal those of
    the whole-row launch to the last bit, with and without the device controller."""
    g = torch.Generator().manual_seed(n + n_lead)
    r = lambda: torch.randn(n, generator=g, dtype=torch.float64).to(dtype).cuda()
    y0, ks = r(), [r() for _ in range(6)]
    rtol_v = (torch.rand(n, generator=g, dtype=torch.float64) * 1e-3 + 1e-6).cuda()
    atol_v = (torch.rand(n, generator=g, dtype=torch.float64) * 1e-5 + 1e-8).cuda()
    kern = _native.get_kernels(torch.device("cuda:0"), dtype)
    plan = kern.make_plan([(0, n, 0.0, 1.0)], n, 1024, torch.device("cuda:0"))
    dt = 0.0371
    y1, part = torch.empty_like(y0), torch.empty_like(y0)
    kern.stage_combine_err(y1, part, y0, ks[:n_lead], COEFS[:n_lead], COEFS[1:n_lead + 1], dt)
    err_coefs = COEFS[1:7]
    kern.error_norm_vec(plan, y0, y1, ks, err_coefs, dt, rtol_v, atol_v)
    whole = kern.read_norms(plan)
    kern.error_norm_vec(plan, y0, y1, ks[n_lead:], err_coefs[n_lead:], dt, rtol_v, atol_v, partial=part)
    split = kern.read_norms(plan)
    assert split[0] == whole[0] and split[2] == whole[2] == [0.0]
    y1[n // 2] = float("nan")
    kern.error_norm_vec(plan, y0, y1, ks[n_lead:], err_coefs[n_lead:], dt, 3e-4, atol_v, partial=part)
    assert kern.read_norms(plan)[2] == [1.0]


def test_solve_with_vector_tolerances_takes_the_fused_kernel(monkeypatch):
    """A no-grad dopri5 solve with a per-element rtol: every trial step's error ratio comes from tdeq_error_norm_vec[_ctrl] (the
    raw-error route `error_scaled` is never taken), and the steps are those of the torch-op evaluation of the same
    formula (TDEQ-internal switch off) — equal evaluation counts, solutions equal to fp64 rounding."""
    from torchdiffeq_amd import solvers
    g = torch.Generator().manual_seed(0)
    A = (torch.randn(12, 12, generator=g, dtype=torch.float64) / 4 - 0.2 * torch.eye(12, dtype=torch.float64)).cuda()
    y0 = torch.randn(300, 12, generator=g, dtype=torch.float64).cuda()
    t = torch.tensor([0.0, 0.7, 2.0], dtype=torch.float64, device="cuda")
    rtol = torch.logspace(-8, -4, 12, dtype=torch.float64, device="cuda").expand(300, 12)
    kern = _native.get_kernels(torch.device("cuda:0"), torch.float64)
    calls = {"vec": 0, "scaled": 0}
    real_vec, real_scaled, real_ctrl = kern.error_norm_vec, kern.error_scaled, kern.error_norm_vec_ctrl
    monkeypatch.setattr(kern, "error_norm_vec", lambda *a, **k: (calls.__setitem__("vec", calls["vec"] + 1), real_vec(*a, **k))[1])
    monkeypatch.setattr(kern, "error_norm_vec_ctrl", lambda *a, **k: (calls.__setitem__("vec", calls["vec"] + 1), real_ctrl(*a, **k))[1])
    monkeypatch.setattr(kern, "error_scaled", lambda *a, **k: (calls.__setitem__("scaled", calls["scaled"] + 1), real_scaled(*a, **k))[1])
    nfe = [0]

    def f(t_, y_):
        nfe[0] += 1
        return y_ @ A.T * torch.cos(t_)
    with torch.no_grad():
        y = tda.odeint(f, y0, t, rtol=rtol, atol=1e-9, method="dopri5")
    n_fused, nfe[0] = nfe[0], 0
    assert calls["vec"] == (n_fused - 2) // 6 and calls["scaled"] == 0
    # the same solve with the fused kernel switched off: the r04 route (raw error + torch ops in fp64)
    orig_init = solvers.RKAdaptiveStepsizeODESolver.__init__

    def init_without(self, *a, **k):
        orig_init(self, *a, **k)
        self._vec_fused, self._vec_ctrl, self._lookahead = None, False, False
    monkeypatch.setattr(solvers.RKAdaptiveStepsizeODESolver, "__init__", init_without)
    with torch.no_grad():
        y_ref = tda.odeint(f, y0, t, rtol=rtol, atol=1e-9, method="dopri5")
    assert calls["scaled"] > 0 and nfe[0] == n_fused
    assert float((y - y_ref).abs().max()) < 1e-12


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
def test_vector_tolerances_keep_the_look_ahead_controller(monkeypatch, dtype):
    """r05: with per-element tolerances the norm launch's finalize step still runs the step controller on the device
    (tdeq_error_norm_vec_ctrl: ratio in fp64 — the promoted type —, step size and stage times in T) and the next trial
    step's first stage is enqueued ahead.  Same solve with TDEQ_LOOKAHEAD=0 (every decision by the host from the read-back
    sums): equal evaluation counts, accepted / rejected counts, identical solutions (fp32 bit for bit)."""
    from torchdiffeq_amd import solvers
    g = torch.Generator().manual_seed(2)
    A = (torch.randn(12, 12, generator=g, dtype=torch.float64) / 3 - 0.2 * torch.eye(12, dtype=torch.float64)).to(dtype).cuda()
    y0 = torch.randn(500, 12, generator=g, dtype=torch.float64).to(dtype).cuda()
    rtol = torch.logspace(-7, -3, 12, dtype=torch.float64, device="cuda").expand(500, 12)
    atol = torch.full((500, 12), 1e-8, dtype=torch.float64, device="cuda")
    kern = _native.get_kernels(torch.device("cuda:0"), dtype)
    calls = {"vec_ctrl": 0}
    real = kern.error_norm_vec_ctrl
    monkeypatch.setattr(kern, "error_norm_vec_ctrl", lambda *a, **k: (calls.__setitem__("vec_ctrl", calls["vec_ctrl"] + 1), real(*a, **k))[1])
    nfe = [0]

    def f(t_, y_):
        nfe[0] += 1
        return (y_ @ A.T) * (1.5 + torch.sin(4 * t_))
    out = {}
    for t in (torch.tensor([0.0, 0.7, 2.0], dtype=torch.float64, device="cuda"),
              torch.tensor([1.5, 0.2], dtype=torch.float64, device="cuda")):
        for la in ("1", "0"):
            monkeypatch.setenv("TDEQ_LOOKAHEAD", la)
            made = []
            orig = solvers.RKAdaptiveStepsizeODESolver.__init__

            def spy(self, *a, **k):
                orig(self, *a, **k)
                made.append(self)
            monkeypatch.setattr(solvers.RKAdaptiveStepsizeODESolver, "__init__", spy)
            nfe[0] = 0
            before = calls["vec_ctrl"]
            with torch.no_grad():
                y = tda.odeint(f, y0, t, rtol=rtol, atol=atol, method="dopri5", options=dict(first_step=0.8))
            monkeypatch.setattr(solvers.RKAdaptiveStepsizeODESolver, "__init__", orig)
            s = made[-1]
            assert bool(s._lookahead) == (la == "1") and s._vec_fused is not None
            assert (calls["vec_ctrl"] > before) == (la == "1")
            out[la] = (y, nfe[0], s.n_accepted, s.n_rejected)
        assert out["1"][1:] == out["0"][1:], (out["1"][1:], out["0"][1:])
        if dtype == torch.float32:
            assert torch.equal(out["1"][0], out["0"][0])
        else:       # the device's `pow` is within 2 ulp of libm's: step sizes agree to 1e-16, so do fp64 rows
            assert float((out["1"][0] - out["0"][0]).abs().max() / out["0"][0].abs().max()) < 1e-13
        assert out["1"][3] > 0          # the large first step is rejected: that path is compared too


@pytest.mark.parametrize("form", ["rtol_list", "atol_list", "both", "numpy_and_tuple"])
@pytest.mark.parametrize("dname", ["f32", "f64"])
def test_tuple_tolerance_with_list_entries_takes_the_fused_kernel(monkeypatch, dname, form):
    """r06: an ENTRY of a tuple tolerance that is a Python list / tuple / numpy array is a vector over
    its component exactly like a tensor entry (torchdiffeq/_impl/misc.py:113-123: `torch.as_tensor(tol_).expand(...)`):
    the solve runs (r05 raised ValueError from `_per_segment`), every trial step's ratio comes from tdeq_error_norm_vec[_ctrl],
    and the result is the reference's (tests/golden/tuple_tol.npz `ttl_*`, generated from the imported reference)."""
    import numpy as np
    from _cases import TUPLE_TOL_LIST_FORMS, T, load, rel_err
    z = load("tuple_tol.npz")
    dtype = torch.float32 if dname == "f32" else torch.float64
    x0, b0 = T(z[f"ttl_{dname}_x0"], "cuda"), T(z[f"ttl_{dname}_b0"], "cuda")
    w = torch.tensor([1.0, 3.0, 0.3], dtype=dtype, device="cuda")
    t = torch.tensor([0.0, 0.4, 1.1], dtype=dtype, device="cuda")
    rtol, atol = TUPLE_TOL_LIST_FORMS[form]
    kern = _native.get_kernels(torch.device("cuda:0"), dtype)
    calls = {"vec": 0, "scaled": 0}
    real_vec, real_scaled, real_ctrl = kern.error_norm_vec, kern.error_scaled, kern.error_norm_vec_ctrl
    monkeypatch.setattr(kern, "error_norm_vec", lambda *a, **k: (calls.__setitem__("vec", calls["vec"] + 1), real_vec(*a, **k))[1])
    monkeypatch.setattr(kern, "error_norm_vec_ctrl", lambda *a, **k: (calls.__setitem__("vec", calls["vec"] + 1), real_ctrl(*a, **k))[1])
    monkeypatch.setattr(kern, "error_scaled", lambda *a, **k: (calls.__setitem__("scaled", calls["scaled"] + 1), real_scaled(*a, **k))[1])
    nfe, accepted = [0], []

    class F(torch.nn.Module):
        def forward(self, t_, y_):
            nfe[0] += 1
            return -y_[0] * w * (1 + 0.2 * t_) + 0.1 * torch.sin(y_[0]), -0.4 * y_[1]
    with torch.no_grad():
        sa, sb = tda.odeint(F(), (x0, b0), t, rtol=rtol, atol=atol, method="dopri5")
    key = f"ttl_{dname}_{form}"
    assert calls["vec"] == (nfe[0] - 2) // 6 and calls["scaled"] == 0
    tol = 1e-11 if dname == "f64" else 2e-5
    assert rel_err(sa, z[f"{key}_ya"]) < tol and rel_err(sb, z[f"{key}_yb"]) < tol
    if dname == "f64":
        assert nfe[0] == int(z[f"{key}_nfe"])


@pytest.mark.parametrize("form", ["both", "rtol_only", "atol_only"])
@pytest.mark.parametrize("mode", [0, 1])
@pytest.mark.parametrize("n", [1, 1031, 5000])
@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
def test_init_norms_vec_equals_the_references_broadcast_expression(dtype, n, mode, form):
    """r06, tdeq_init_norms_vec (ABI 21): misc.py:50-56,68 evaluated literally by ATen on the same device — `scale = atol +
    |y0| * rtol` with W = fp64 tolerance tensors, type promotion included (a dimensioned fp64 tolerance promotes, a 0-dim one
    does not), `(f1 - f0)` formed in T — against the kernel's fp64 sums of squares."""
    g = torch.Generator().manual_seed(7 * n + mode)
    r = lambda: torch.randn(n, generator=g, dtype=torch.float64).to(dtype).cuda()
    a, b, y = r(), r(), r()
    rtol_v = (torch.rand(n, generator=g, dtype=torch.float64) * 1e-3 + 1e-6).cuda()
    atol_v = (torch.rand(n, generator=g, dtype=torch.float64) * 1e-5 + 1e-8).cuda()
    rtol = rtol_v if form != "atol_only" else torch.tensor(3e-4, dtype=torch.float64, device="cuda")
    atol = atol_v if form != "rtol_only" else torch.tensor(2e-6, dtype=torch.float64, device="cuda")
    kern = _native.get_kernels(torch.device("cuda:0"), dtype)
    plan = kern.make_plan([(0, n, 0.0, 1.0)], n, 1024, torch.device("cuda:0"))
    kern.init_norms_vec(plan, mode, a, b, y, rtol if rtol.dim() else float(rtol), atol if atol.dim() else float(atol))
    s0, s1, bad = kern.read_norms(plan)
    scale = atol + torch.abs(y) * rtol                      # misc.py:50
    assert scale.dtype == torch.float64
    if mode == 0:
        q0, q1 = a / scale, b / scale                       # misc.py:55-56
        assert s0[0] == pytest.approx(float(q0.abs().pow(2).sum()), rel=1e-12)
        assert s1[0] == pytest.approx(float(q1.abs().pow(2).sum()), rel=1e-12)
    else:
        q = (a - b) / scale                                 # misc.py:68
        assert q.dtype == torch.float64
        assert s0[0] == pytest.approx(float(q.abs().pow(2).sum()), rel=1e-12)
    assert bad == [0.0]
    y[n // 2] = float("inf")
    kern.init_norms_vec(plan, mode, a, b, y, rtol if rtol.dim() else float(rtol), atol if atol.dim() else float(atol))
    assert kern.read_norms(plan)[2] == [1.0]


@pytest.mark.parametrize("tuple_state", [False, True], ids=["tensor", "tuple"])
@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
def test_initial_step_with_vector_tolerances_runs_on_the_kernel_and_keeps_its_value(monkeypatch, dtype, tuple_state):
    """The once-per-solve heuristic (misc.py:36-77) with per-element tolerances: three tdeq_init_norms_vec launches, no
    `init_scaled` + torch-op scaling (the r05 route) — and the first step size it returns equals the r05 route's to fp64
    rounding (sums of ≤ 5000 fp64 squares in a different order), the evaluation counts of the whole solve are equal."""
    from torchdiffeq_amd import solvers
    g = torch.Generator().manual_seed(11)
    A = (torch.randn(12, 12, generator=g, dtype=torch.float64) / 4 - 0.2 * torch.eye(12, dtype=torch.float64)).to(dtype).cuda()
    ya = torch.randn(300, 12, generator=g, dtype=torch.float64).to(dtype).cuda()
    yb = torch.randn(7, generator=g, dtype=torch.float64).to(dtype).cuda()
    t = torch.tensor([0.0, 0.7, 2.0], dtype=torch.float64, device="cuda")
    rt = torch.logspace(-7, -4, 12, dtype=torch.float64, device="cuda").expand(300, 12)
    if tuple_state:
        y0, rtol, atol = (ya, yb), (rt.reshape(-1), 1e-5), (1e-9, torch.full((7,), 1e-8, dtype=torch.float64, device="cuda"))
        f = lambda t_, y_: (y_[0] @ A.T * torch.cos(t_), -0.5 * y_[1])
    else:
        y0, rtol, atol = ya, rt, 1e-9
        f = lambda t_, y_: y_ @ A.T * torch.cos(t_)
    kern = _native.get_kernels(torch.device("cuda:0"), dtype)
    calls = {"vec": 0, "scaled": 0}
    real_vec, real_scaled = kern.init_norms_vec, kern.init_scaled
    monkeypatch.setattr(kern, "init_norms_vec", lambda *a, **k: (calls.__setitem__("vec", calls["vec"] + 1), real_vec(*a, **k))[1])
    monkeypatch.setattr(kern, "init_scaled", lambda *a, **k: (calls.__setitem__("scaled", calls["scaled"] + 1), real_scaled(*a, **k))[1])
    made = []
    orig = solvers.RKAdaptiveStepsizeODESolver._select_initial_step

    def spy(self, *a, **k):
        dt0 = orig(self, *a, **k)
        made.append(dt0)
        return dt0
    monkeypatch.setattr(solvers.RKAdaptiveStepsizeODESolver, "_select_initial_step", spy)
    nfe = [0]

    def counted(t_, y_):
        nfe[0] += 1
        return f(t_, y_)
    with torch.no_grad():
        y_new = tda.odeint(counted, y0, t, rtol=rtol, atol=atol, method="dopri5", options=dict(hip_graph=False))
    n_new, nfe[0] = nfe[0], 0
    assert calls == {"vec": 2, "scaled": 0}
    # the r05 route: hide the entry point
    monkeypatch.delattr(type(kern), "init_norms_vec", raising=False)
    monkeypatch.delattr(kern, "init_norms_vec", raising=False)
    assert not hasattr(kern, "init_norms_vec")
    with torch.no_grad():
        y_old = tda.odeint(counted, y0, t, rtol=rtol, atol=atol, method="dopri5", options=dict(hip_graph=False))
    assert calls["scaled"] == 2 and nfe[0] == n_new
    assert made[0] == pytest.approx(made[1], rel=1e-12)
    ya_new, ya_old = (y_new[0], y_old[0]) if tuple_state else (y_new, y_old)
    assert float((ya_new - ya_old).abs().max()) <= (1e-12 if dtype == torch.float64 else 2e-5) * float(ya_old.abs().max())


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
def test_captured_steps_with_vector_tolerances_equal_the_eager_solve(dtype):
    """r06 (ABI 21: `state_in_dev` on tdeq_error_norm_vec_ctrl): per-element tolerances no longer switch captured trial steps
    off.  hip_graph=True against the eager look-ahead solve: identical bits, fewer Python evaluations; a second solve with
    OTHER tolerance values re-uses the captured step (its static tolerance buffers are refreshed) and is again identical."""
    from torchdiffeq_amd import _graph
    g = torch.Generator().manual_seed(2)
    lin = torch.nn.Linear(12, 12).to(dtype).cuda()
    y0 = torch.randn(500, 12, generator=g, dtype=torch.float64).to(dtype).cuda()
    t = torch.tensor([0.0, 0.7, 2.0], dtype=torch.float64, device="cuda")
    calls = [0]
    lin.register_forward_pre_hook(lambda m, a: calls.__setitem__(0, calls[0] + 1))
    f = lambda t_, y_: torch.tanh(lin(y_)) * (1.5 + torch.sin(4 * t_))
    try:
        for scale in (1.0, 0.1):
            rtol = (torch.logspace(-7, -3, 12, dtype=torch.float64, device="cuda") * scale).expand(500, 12)
            atol = torch.full((500, 12), 1e-8 * scale, dtype=torch.float64, device="cuda")
            with torch.no_grad():
                calls[0] = 0
                y_e = tda.odeint(f, y0, t, rtol=rtol, atol=atol, method="dopri5", options=dict(hip_graph=False))
                n_e, calls[0] = calls[0], 0
                y_g = tda.odeint(f, y0, t, rtol=rtol, atol=atol, method="dopri5", options=dict(hip_graph=True))
                n_g = calls[0]
            assert torch.equal(y_g, y_e), scale
            assert n_g < n_e, (n_g, n_e)
        assert f in _graph._GraphStep._cache
    finally:
        _graph.clear_graph_cache()
