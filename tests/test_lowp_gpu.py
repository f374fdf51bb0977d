"""bfloat16 / float16 states on the MI355X: the HIP kernels of csrc/tdeq_kernels_lp.hpp (through the C-ABI, dtype
TDEQ_BF16 / TDEQ_F16) against oracle/lp_kernels.py — the reference's own torch expressions evaluated by ATen's CPU
kernels on reduced-precision tensors (pinned to the reference by tests/test_lowp_oracle.py) — BIT FOR BIT, element by
element; the norm sums to fp64 rounding; and whole solves: no torch-op host path for a reduced-precision `cuda` state."""
import os
import sys
import warnings

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
pytestmark = pytest.mark.gpu

import torchdiffeq_amd as tda  # noqa: E402
from oracle import lp_kernels as olp  # noqa: E402
from torchdiffeq_amd import _fallback, _lowp, _native, tableaus as tb  # noqa: E402

DT = {"bf16": torch.bfloat16, "f16": torch.float16}
SIZES = [1, 7, 8, 1031, 8192 + 3]


def kernels(dtype):
    k = _native.get_kernels(torch.device("cuda:0"), dtype)
    assert isinstance(k, _lowp.LowPrecisionHipKernels) and k.name == "hip-low"
    return k


def draw(n, seed, dtype, count):
    g = torch.Generator().manual_seed(seed)
    return [torch.randn(n, generator=g, dtype=torch.float64).to(dtype) for _ in range(count)]


def dev(ts, offset=0):
    """On the GPU; offset > 0: a view that starts `offset` elements into its buffer (not 16-byte aligned)."""
    out = []
    for t in ts:
        buf = torch.empty(t.numel() + offset, dtype=t.dtype, device="cuda")
        buf[offset:].copy_(t)
        out.append(buf[offset:])
    return out


def same(got_dev, ref_cpu):
    got = got_dev.cpu()
    assert got.dtype == ref_cpu.dtype
    assert torch.equal(got.view(torch.int16), ref_cpu.view(torch.int16)), \
        int((got.view(torch.int16) != ref_cpu.view(torch.int16)).sum())


COEFS = (0.0371, -0.211, 0.5, 1.25, -0.0625, 0.33, 0.9, -0.7, 0.0123, 2.5, -1.0, 0.125, 0.77, -0.31)


@pytest.mark.parametrize("offset", [0, 3])
@pytest.mark.parametrize("n", SIZES)
@pytest.mark.parametrize("low", ["bf16", "f16"])
def test_stage_combines_bit_exact(low, n, offset):
    dtype, k = DT[low], kernels(DT[low])
    y0, *ks = draw(n, 1, dtype, 15)
    y0d, *ksd = dev([y0] + ks, offset)
    for nt in (1, 2, 3, 6, 7, 13, 14):
        out = torch.empty_like(y0d)
        k.stage_combine(out, y0d, ksd[:nt], COEFS[:nt], 0.0371)
        same(out, olp.stage_combine(y0, ks[:nt], COEFS[:nt], 0.0371))
    # first stage + the step's stage times in one launch
    times = torch.empty(6, dtype=dtype, device="cuda")
    vals = [0.3, 0.30742, 0.31113, 0.32968, 0.33297, 0.3371]
    out = torch.empty_like(y0d)
    k.stage_combine_fill(out, y0d, ksd[:1], COEFS[:1], -0.0371, times, vals)
    same(out, olp.stage_combine(y0, ks[:1], COEFS[:1], -0.0371))
    same(times, torch.tensor(vals, dtype=torch.float64).to(dtype))
    # last combine + the error row over the same stages
    out, eo = torch.empty_like(y0d), torch.empty_like(y0d)
    k.stage_combine_err(out, eo, y0d, ksd[:6], COEFS[:6], COEFS[6:12], 0.0371)
    y_ref, e_ref = olp.stage_combine_err(y0, ks[:6], COEFS[:6], COEFS[6:12], 0.0371)
    same(out, y_ref)
    same(eo, e_ref)


@pytest.mark.parametrize("n", [1, 1031, 5000])
@pytest.mark.parametrize("low", ["bf16", "f16"])
def test_error_norm_and_init_norms(low, n):
    dtype, k = DT[low], kernels(DT[low])
    y0, y1, *ks = draw(n, 2, dtype, 9)
    y0d, y1d, *ksd = dev([y0, y1] + ks)
    chunk = 1024
    plan = k.make_plan([(0, n, 1e-2, 1e-3)], n, chunk, torch.device("cuda:0"))
    scaled = torch.empty_like(y0d)
    k.error_norm(plan, y0d, y1d, ksd[:7], COEFS[:7], 0.0371, scaled_out=scaled)
    s0, _, bad = k.read_norms(plan)
    q = olp.error_quotient(y0, y1, ks[:7], COEFS[:7], 0.0371, 1e-2, 1e-3)
    same(scaled, q)
    sumsq, _ = olp.norm_terms(q)
    assert s0[0] == pytest.approx(sumsq, rel=1e-12) and bad == [0.0]
    # the norm value the solver uses = ATen's sqrt(mean(|q|^2)) of the same tensor, in the state's type
    assert plan.rms0[0] == float(q.abs().pow(2).mean().sqrt())
    if n == 1:
        assert plan.abs0[0] == float(q.abs())
    k.error_norm(plan, y0d, y1d, ksd[:7], COEFS[:7], 0.0371)           # without the materialised quotient
    assert k.read_norms(plan)[0][0] == pytest.approx(sumsq, rel=1e-12)
    # initial-step quotients and their norms
    k.init_norms(plan, 0, ksd[0], ksd[1], y0d)
    s0, s1, _ = k.read_norms(plan)
    q0, q1 = olp.init_quotients(0, ks[0], ks[1], y0, 1e-2, 1e-3)
    assert s0[0] == pytest.approx(olp.norm_terms(q0)[0], rel=1e-12) and s1[0] == pytest.approx(olp.norm_terms(q1)[0], rel=1e-12)
    assert plan.rms0[0] == float(q0.abs().pow(2).mean().sqrt()) and plan.rms1[0] == float(q1.abs().pow(2).mean().sqrt())
    k.init_norms(plan, 1, ksd[0], ksd[1], y0d)
    assert k.read_norms(plan)[0][0] == pytest.approx(olp.norm_terms(olp.init_quotients(1, ks[0], ks[1], y0, 1e-2, 1e-3)[0])[0], rel=1e-12)
    o0, o1 = torch.empty_like(y0d), torch.empty_like(y0d)
    k.init_scaled(plan, 0, ksd[0], ksd[1], y0d, o0, o1)
    same(o0, q0)
    same(o1, q1)
    # a non-finite state element is counted, not averaged away
    y_bad = y0.clone()
    y_bad[0] = float("inf")
    k.error_norm(plan, dev([y_bad])[0], y1d, ksd[:7], COEFS[:7], 0.0371)
    assert k.read_norms(plan)[2] == [1.0]


@pytest.mark.parametrize("low", ["bf16", "f16"])
def test_segmented_state_norms(low):
    """Three segments with their own tolerances and padding between them (the adjoint's layout)."""
    dtype, k = DT[low], kernels(DT[low])
    chunk, sizes, tols = 1024, [1, 1500, 37], [(1e-2, 1e-3), (2e-2, 1e-3), (1e-1, 1e-2)]
    offs, total = [], 0
    for s in sizes:
        offs.append(total)
        total += -(-s // chunk) * chunk
    y0, y1, *ks = draw(total, 3, dtype, 8)
    y0d, y1d, *ksd = dev([y0, y1] + ks)
    plan = k.make_plan([(o, s, rt, at) for o, s, (rt, at) in zip(offs, sizes, tols)], total, chunk, torch.device("cuda:0"))
    scaled = torch.empty_like(y0d)
    k.error_norm(plan, y0d, y1d, ksd[:6], COEFS[:6], 0.05, scaled_out=scaled)
    s0, _, _ = k.read_norms(plan)
    got = scaled.cpu()
    for i, (o, s, (rt, at)) in enumerate(zip(offs, sizes, tols)):
        sl = slice(o, o + s)
        q = olp.error_quotient(y0[sl], y1[sl], [kk[sl] for kk in ks[:6]], COEFS[:6], 0.05, rt, at)
        assert torch.equal(got[sl].view(torch.int16), q.view(torch.int16))
        assert s0[i] == pytest.approx(olp.norm_terms(q)[0], rel=1e-12)
        assert plan.rms0[i] == float(q.abs().pow(2).mean().sqrt())
        pad = got[o + s:offs[i + 1] if i + 1 < len(offs) else total]
        assert not pad.numel() or float(pad.abs().max()) == 0.0
    assert plan.abs0[0] == float(got[0].abs())


@pytest.mark.parametrize("n", [1, 8, 1031, 8192 + 3])
@pytest.mark.parametrize("low", ["bf16", "f16"])
def test_dense_output_and_fixed_grid_stages_bit_exact(low, n):
    dtype, k = DT[low], kernels(DT[low])
    y0, y1, *ks = draw(n, 4, dtype, 16)
    y0d, y1d, *ksd = dev([y0, y1] + ks)
    for nt in (3, 6, 10):
        out = torch.empty_like(y0d)
        for x in (0.0, 0.3, 1.0):
            k.dense_eval(out, y0d, y1d, ksd[0], ksd[13], ksd[:nt], COEFS[:nt], 0.0371, x)
            same(out, olp.dense_eval(y0, y1, ks[0], ks[13], ks[:nt], COEFS[:nt], 0.0371, x))
        planes = torch.empty(5, n, dtype=dtype, device="cuda")
        k.interp_fit(planes, y0d, y1d, ksd[0], ksd[13], ksd[:nt], COEFS[:nt], 0.0371)
        for got, ref in zip(planes, olp.quartic(y0, y1, ks[0], ks[13], ks[:nt], COEFS[:nt], 0.0371)):
            same(got, ref)
    xs = [0.05 * i for i in range(1, 12)]
    for m in (2, 4, 5, 11):
        rows = torch.empty(m, n, dtype=dtype, device="cuda")
        k.dense_eval_multi(rows, y0d, y1d, ksd[0], ksd[13], ksd[:6], COEFS[:6], 0.0371, xs[:m])
        for got, x in zip(rows, xs):
            same(got, olp.dense_eval(y0, y1, ks[0], ks[13], ks[:6], COEFS[:6], 0.0371, x))
    out = torch.empty_like(y0d)
    for stage in (1, 2, 3, 4):
        k.rk4_stage(stage, out, y0d, ksd[0], ksd[1], ksd[2] if stage > 2 else None, ksd[3] if stage > 3 else None, 0.025)
        same(out, olp.rk4_stage(stage, y0, ks[0], ks[1], ks[2], ks[3], 0.025))
    k.lerp(out, y0d, y1d, 0.2417)
    same(out, olp.lerp(y0, y1, 0.2417))
    k.fixed_stage(1, out, y0d, ksd[:1], (1 / 3,), 0.025)
    same(out, olp.fixed_stage(1, y0, ks[:1], (1 / 3,), 0.025))
    k.fixed_stage(0, out, y0d, ksd[:3], (0.25, 0.1, 0.75), 0.025)
    same(out, olp.fixed_stage(0, y0, ks[:3], (0.25, 0.1, 0.75), 0.025))
    k.scaled_add(out, y0d, ksd[0], 0.0125)
    same(out, y0 + ks[0] * 0.0125)
    k.weighted_sum(out, ksd[:3], (0.7, -1.3, 2.0))
    same(out, olp.weighted_sum(ks[:3], (0.7, -1.3, 2.0)))


def test_entry_points_outside_the_step_refuse_reduced_precision():
    """Entry points without 16-bit kernels (backward helpers, the fused partial-error forms): TDEQ_EINVAL, not a misread buffer."""
    import ctypes
    lib = _native.load_library()
    y = torch.zeros(64, dtype=torch.bfloat16, device="cuda")
    ptrs = (ctypes.c_void_p * 1)(y.data_ptr())
    w = (ctypes.c_double * 1)(0.5)
    assert lib.tdeq_scale_many(ptrs, y.data_ptr(), w, 1, 64, _native.TDEQ_BF16, None) == -1
    # the captured-step combine exists for 16-bit states, but not its fused partial-error output (a row is rounded once)
    assert lib.tdeq_stage_combine_dev(y.data_ptr(), y.data_ptr(), y.data_ptr(), ptrs, w, w, 1, y.data_ptr(), 64,
                                      _native.TDEQ_F16, None) == -1


@pytest.mark.parametrize("method,kw", [("dopri5", dict(rtol=1e-2, atol=1e-3)), ("dopri8", dict(rtol=1e-2, atol=1e-3)),
                                        ("bosh3", dict(rtol=1e-2, atol=1e-3)), ("rk4", {})])
def test_solves_run_on_the_hip_kernels_and_follow_the_torch_op_path(method, kw):
    """A bf16 `cuda` state: no HostPathWarning (the torch-op path is not taken), the solver's backend is `hip-low`, and
    the solution agrees with the package's own torch-op path forced onto the same device — the arithmetic is the same
    up to the order of a row's float32 accumulation, so the step sequences coincide (equal evaluation counts) and the
    solutions differ by last-place noise."""
    g = torch.Generator().manual_seed(0)
    A = (torch.randn(16, 16, generator=g) / 4 - 0.3 * torch.eye(16)).to(torch.bfloat16).cuda()
    y0 = torch.randn(256, 16, generator=g).to(torch.bfloat16).cuda()
    t = torch.linspace(0.0, 1.0, 5, device="cuda")
    nfe = [0]

    def f(t_, y_):
        nfe[0] += 1
        return (y_ @ A.T) * torch.cos(t_)
    with torch.no_grad(), warnings.catch_warnings():
        warnings.simplefilter("error", _fallback.HostPathWarning)
        y = tda.odeint(f, y0, t, method=method, **kw)
    n_hip, nfe[0] = nfe[0], 0
    low = _fallback.LowPrecisionHostKernels()
    orig = _native.get_kernels
    _native.get_kernels = lambda device, dtype=None: low if dtype == torch.bfloat16 else orig(device, dtype)
    try:
        with torch.no_grad():
            y_host = tda.odeint(f, y0, t, method=method, **kw)
    finally:
        _native.get_kernels = orig
    assert y.dtype == torch.bfloat16 and torch.isfinite(y.float()).all()
    assert n_hip == nfe[0], (n_hip, nfe[0])
    err = float((y.float() - y_host.float()).abs().max() / y_host.float().abs().max())
    assert err < 0.02, err


@pytest.mark.parametrize("method", ["dopri5", "dopri8", "bosh3", "tsit5"])
def test_device_controller_and_look_ahead_take_the_host_loops_steps(monkeypatch, method):
    """The norm launch's finalize step runs the step controller on the device IN THE STATE'S TYPE (ratio = ATen's
    sqrt(mean) sequence in bf16, next step size, the next trial step's stage times with bf16 arithmetic and nextafter)
    and the next first stage is enqueued ahead (tdeq_stage_combine_sel).  Same solve with TDEQ_LOOKAHEAD=0 — every
    decision taken by the host loop from the read-back sums: identical evaluation counts and bit-identical solutions,
    forwards and in decreasing time, including rejected steps."""
    from torchdiffeq_amd import solvers
    g = torch.Generator().manual_seed(1)
    A = (torch.randn(16, 16, generator=g) / 2 - 0.2 * torch.eye(16)).to(torch.bfloat16).cuda()
    y0 = (3 * torch.randn(512, 16, generator=g)).to(torch.bfloat16).cuda()
    nfe = [0]

    def f(t_, y_):
        nfe[0] += 1
        return (y_ @ A.T) * (2 + 2 * torch.sin(3 * t_))
    seen = {}
    for t in (torch.linspace(0.0, 3.0, 7, device="cuda"), torch.linspace(2.0, -1.0, 4, device="cuda")):
        out = {}
        for la in ("1", "0"):
            monkeypatch.setenv("TDEQ_LOOKAHEAD", la)
            nfe[0] = 0
            made = []
            orig = solvers.RKAdaptiveStepsizeODESolver.__init__

            def spy(self, *a, **k):
                orig(self, *a, **k)
                made.append(self)
            monkeypatch.setattr(solvers.RKAdaptiveStepsizeODESolver, "__init__", spy)
            with torch.no_grad():
                y = tda.odeint(f, y0, t, method=method, rtol=3e-2, atol=1e-3, options=dict(first_step=2.0))
            monkeypatch.setattr(solvers.RKAdaptiveStepsizeODESolver, "__init__", orig)
            s = made[-1]
            assert s.kernels.name == "hip-low" and bool(s._lookahead) == (la == "1")
            out[la] = (y, nfe[0], s.n_accepted, s.n_rejected)
        assert out["1"][1:] == out["0"][1:], (out["1"][1:], out["0"][1:])
        assert torch.equal(out["1"][0].view(torch.int16), out["0"][0].view(torch.int16))
        seen[float(t[-1])] = out["1"][1:]
    assert any(v[2] > 0 for v in seen.values())      # the large first step is rejected: that path is compared too


@pytest.mark.parametrize("norm", [None, "seminorm"])
def test_bf16_adjoint_backward_keeps_the_device_controller(monkeypatch, norm):
    """odeint_adjoint on a bf16 state and bf16 parameters — the mainstream reduced-precision use.  The backward solve's
    augmented state [vjp_t | y | adj_y | θ…] has a ONE-element first segment that enters the adjoint norm as |t|
    (adjoint.py:250, 273), not as an rms: the 16-bit device controller takes it that way (`leading_abs`), so the backward
    solve keeps the controller + look-ahead.  Against TDEQ_LOOKAHEAD=0 (host-driven decisions): equal evaluation counts,
    bit-identical solution and gradients."""
    torch.manual_seed(0)
    lin1 = torch.nn.Linear(8, 32).to(torch.bfloat16).cuda()
    lin2 = torch.nn.Linear(32, 8).to(torch.bfloat16).cuda()

    class Field(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.a, self.b, self.nfe = lin1, lin2, 0

        def forward(self, t, y):
            self.nfe += 1
            return self.b(torch.tanh(self.a(y))) * (1 + torch.sin(2 * t))
    field = Field()
    x0 = torch.randn(64, 8, generator=torch.Generator().manual_seed(1)).to(torch.bfloat16).cuda()
    t = torch.tensor([0.0, 0.5, 1.5], device="cuda")
    res = {}
    for la in ("1", "0"):
        monkeypatch.setenv("TDEQ_LOOKAHEAD", la)
        field.zero_grad()
        x = x0.clone().requires_grad_(True)
        field.nfe = 0
        kw = dict(adjoint_options=dict(norm=norm)) if norm else {}
        y = tda.odeint_adjoint(field, x, t, method="dopri5", rtol=2e-2, atol=2e-3, **kw)
        n_fwd, field.nfe = field.nfe, 0
        (y[-1].float().pow(2).sum() + y[1].float().sum()).backward()
        res[la] = (n_fwd, field.nfe, y.detach().clone(), x.grad.clone(), [p.grad.clone() for p in field.parameters()])
    assert res["1"][:2] == res["0"][:2], (res["1"][:2], res["0"][:2])
    assert torch.equal(res["1"][2].view(torch.int16), res["0"][2].view(torch.int16))
    assert torch.equal(res["1"][3].view(torch.int16), res["0"][3].view(torch.int16))
    for a, b in zip(res["1"][4], res["0"][4]):
        assert torch.equal(a.view(torch.int16), b.view(torch.int16)) and float(a.float().abs().max()) > 0


@pytest.mark.parametrize("method", ["dopri5", "bosh3", "tsit5", "dopri8"])
def test_captured_trial_steps_of_a_bf16_state_replay_the_eager_solve(method):
    """`hip_graph=True` for a bf16 state: one trial step — the S evaluations of func, the 16-bit combines reading the step
    size from the device controller's words (tdeq_stage_combine_dev), the whole-row error norm + controller with the state on
    the device — is ONE hipGraph replay.  Same kernels, same decisions: bit-identical to the eager (look-ahead) solve, and
    really replayed."""
    tda.clear_graph_cache()
    g = torch.Generator().manual_seed(4)
    A = (torch.randn(16, 16, generator=g) / 2 - 0.2 * torch.eye(16)).to(torch.bfloat16).cuda()
    y0 = (2 * torch.randn(128, 16, generator=g)).to(torch.bfloat16).cuda()

    def f(t_, y_):
        return (y_ @ A.T) * (2 + 2 * torch.sin(3 * t_))
    replays = [0]
    real = torch.cuda.CUDAGraph.replay

    def counting(self):
        replays[0] += 1
        return real(self)
    for t in (torch.linspace(0.0, 3.0, 7, device="cuda"), torch.linspace(2.0, -1.0, 4, device="cuda")):
        with torch.no_grad():
            eager = tda.odeint(f, y0, t, method=method, rtol=3e-2, atol=1e-3, options=dict(first_step=2.0))
            torch.cuda.CUDAGraph.replay = counting
            try:
                with warnings.catch_warnings():
                    warnings.simplefilter("error")           # in particular: no "running the eager path"
                    captured = tda.odeint(f, y0, t, method=method, rtol=3e-2, atol=1e-3,
                                          options=dict(first_step=2.0, hip_graph=True))
            finally:
                torch.cuda.CUDAGraph.replay = real
        assert torch.equal(captured.view(torch.int16), eager.view(torch.int16))
    assert replays[0] > 4
    tda.clear_graph_cache()
