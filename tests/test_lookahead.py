"""Device-resident step controller + look-ahead first stage (tdeq_error_norm_partial_ctrl / tdeq_stage_combine_sel).

The look-ahead path must be invisible: the same accept/reject sequence, the same number of func evaluations and the
same solution as the host-driven loop (TDEQ_LOOKAHEAD=0).
  * "cpu" runs: host logic over the CPU oracle — both loops use libm's pow, so everything is BIT-identical.
  * "cuda" runs: the device controller uses the GPU's pow, which may differ from libm in the last ulp of dt_next,
    so solutions are compared to 1e-12 (fp64) / 2e-6 (fp32); step counts and NFE must still be equal.
  * kernel parity: the controller kernel vs the oracle's controller fed with the device's own sums.
"""
import math

import numpy as np
import pytest
import torch

import torchdiffeq_amd as tda
from torchdiffeq_amd import _native
from torchdiffeq_amd.tableaus import DOPRI5, DOPRI8, SparseRow

from _cases import StatFunc


class _Counting:
    def __init__(self, fn):
        self.fn, self.nfe = fn, 0

    def __call__(self, t, y):
        self.nfe += 1
        return self.fn(t, y)


def _solve(monkeypatch, lookahead, fn, y0, t, **kw):
    monkeypatch.setenv("TDEQ_LOOKAHEAD", "1" if lookahead else "0")
    f = _Counting(fn)
    with torch.no_grad():
        y = tda.odeint(f, y0, t, **kw)
    return y, f.nfe


def _vdp(mu):
    def f(t, y):
        x, v = y[..., 0], y[..., 1]
        return torch.stack([v, mu * (1 - x * x) * v - x], dim=-1)
    return f


CASES = [
    # (name, field, y0 shape builder, t, kwargs)
    ("dopri5_linear", "lin", [0.0, 0.3, 1.0, 2.5], dict(method="dopri5", rtol=1e-6, atol=1e-8)),
    ("dopri5_reverse", "lin", [2.0, 1.0, -0.5], dict(method="dopri5", rtol=1e-6, atol=1e-8)),
    ("dopri5_rejects", "vdp", [0.0, 3.0, 7.0], dict(method="dopri5", rtol=1e-5, atol=1e-7,
                                                  options=dict(first_step=0.9))),
    ("dopri8", "vdp", [0.0, 5.0], dict(method="dopri8", rtol=1e-8, atol=1e-10)),
    ("tsit5", "vdp", [0.0, 2.0, 4.0], dict(method="tsit5", rtol=1e-6, atol=1e-8)),
    ("bosh3", "vdp", [0.0, 4.0], dict(method="bosh3", rtol=1e-4, atol=1e-6)),
    ("adaptive_heun", "lin", [0.0, 0.5], dict(method="adaptive_heun", rtol=1e-3, atol=1e-5)),
    ("fehlberg2_not_capable", "lin", [0.0, 0.5], dict(method="fehlberg2", rtol=1e-3, atol=1e-5)),
    ("min_max_step", "vdp", [0.0, 3.0], dict(method="dopri5", rtol=1e-6, atol=1e-8,
                                             options=dict(min_step=1e-3, max_step=0.05))),
    ("time_dependent", "tdep", [0.0, 1.0, 2.0], dict(method="dopri5", rtol=1e-7, atol=1e-9)),
]


def _problem(kind, dtype, device):
    g = torch.Generator(device="cpu").manual_seed(3)
    if kind == "lin":
        A = (torch.randn(6, 6, generator=g, dtype=torch.float64, device="cpu") * 0.5 - torch.eye(6, dtype=torch.float64, device="cpu")).to(dtype)
        A = A.to(device)
        y0 = torch.randn(50, 6, generator=g, dtype=torch.float64, device="cpu").to(dtype).to(device)
        return (lambda t, y: y @ A.T), y0
    if kind == "tdep":
        y0 = torch.randn(40, 3, generator=g, dtype=torch.float64, device="cpu").to(dtype).to(device)
        return (lambda t, y: -y * (1 + torch.sin(3 * t)) + torch.cos(t)), y0
    y0 = torch.tensor([[2.0, 0.0], [1.0, -1.0], [0.5, 0.5]], dtype=dtype, device=device)
    return _vdp(3.0), y0


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32], ids=["f64", "f32"])
@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_lookahead_is_invisible(dev, monkeypatch, case, dtype):
    name, kind, ts, kw = case
    fn, y0 = _problem(kind, dtype, dev)
    t = torch.tensor(ts, dtype=torch.float64, device=dev)
    y_on, nfe_on = _solve(monkeypatch, True, fn, y0, t, **kw)
    y_off, nfe_off = _solve(monkeypatch, False, fn, y0, t, **kw)
    assert nfe_on == nfe_off, f"{name}: look-ahead changed the number of func evaluations"
    if dev == "cpu":
        assert torch.equal(y_on, y_off), f"{name}: look-ahead changed the solution"
    else:
        tol = 1e-12 if dtype == torch.float64 else 2e-6
        err = float((y_on - y_off).abs().max() / y_off.abs().max())
        assert err <= tol, f"{name}: {err}"


def test_lookahead_runs_and_keeps_step_sequence(dev, monkeypatch):
    """The look-ahead path is really taken (stage_combine_sel launched), and accepted / rejected step sizes equal
    those of the host-driven loop (callbacks disable look-ahead, so that run is the reference sequence)."""
    fn, y0 = _problem("vdp", torch.float64, dev)
    t = torch.tensor([0.0, 3.0, 6.0], dtype=torch.float64, device=dev)
    kw = dict(method="dopri5", rtol=1e-6, atol=1e-8, options=dict(first_step=0.7))
    stat = StatFunc(fn)
    with torch.no_grad():
        y_cb = tda.odeint(stat, y0, t, **kw)
    assert len(stat.reject) > 0, "the case is meant to exercise rejected steps"

    kern = _native.get_kernels(y0.device)
    calls = {"sel": 0, "ctrl": 0}
    sel, ctrl = kern.stage_combine_sel, kern.error_norm_partial_ctrl
    monkeypatch.setattr(kern, "stage_combine_sel", lambda *a, **k: (calls.__setitem__("sel", calls["sel"] + 1), sel(*a, **k))[1],
                        raising=False)
    monkeypatch.setattr(kern, "error_norm_partial_ctrl",
                        lambda *a, **k: (calls.__setitem__("ctrl", calls["ctrl"] + 1), ctrl(*a, **k))[1], raising=False)
    y_la, nfe = _solve(monkeypatch, True, fn, y0, t, **kw)
    n_trials = len(stat.accept) + len(stat.reject)
    assert calls["ctrl"] == n_trials
    assert 0 < calls["sel"] < n_trials          # every trial but those that may end the solve
    assert nfe == stat.nfe
    if dev == "cpu":
        assert torch.equal(y_la, y_cb)
    else:
        assert float((y_la - y_cb).abs().max()) < 1e-11


def test_lookahead_tuple_state_and_adjoint(dev, monkeypatch):
    """Segmented states (tuple forward state; the adjoint's augmented state with parameter segments)."""
    torch.manual_seed(0)
    lin = torch.nn.Linear(4, 4).double().to(dev)
    y0 = (torch.randn(8, 4, dtype=torch.float64).to(dev), torch.randn(3, dtype=torch.float64).to(dev))

    class F(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.lin = lin

        def forward(self, t, y):
            a, b = y
            return torch.tanh(self.lin(a)), -b * t

    t = torch.tensor([0.0, 0.7, 1.5], dtype=torch.float64, device=dev)
    grads = {}
    for la in (True, False):
        monkeypatch.setenv("TDEQ_LOOKAHEAD", "1" if la else "0")
        f = F()
        f.zero_grad()
        a0 = y0[0].clone().requires_grad_(True)
        ya, yb = tda.odeint_adjoint(f, (a0, y0[1]), t, rtol=1e-7, atol=1e-9, method="dopri5")
        (ya[-1].pow(2).sum() + yb[-1].sum()).backward()
        grads[la] = (ya.detach().clone(), a0.grad.clone(), lin.weight.grad.clone())
    for on, off in zip(grads[True], grads[False]):
        if dev == "cpu":
            assert torch.equal(on, off)
        else:
            assert float((on - off).abs().max() / off.abs().max()) < 1e-11


# ---------------------------------------------------------------------------------------------------------------
# kernel parity on the GPU
# ---------------------------------------------------------------------------------------------------------------
def _ctrl(t0, dt, order, tab, sign=1.0, min_step=0.0, max_step=math.inf, n_norm_seg=1, np_dtype=np.float64):
    c = _native.StepCtrl()
    c.t0, c.dt = t0, dt
    c.safety, c.ifactor, c.dfactor = 0.9, 10.0, 0.2
    c.exponent = 1.0 / order
    c.min_step, c.max_step, c.time_sign = min_step, max_step, sign
    mask = 0
    for i, a in enumerate(tab.alpha):
        c.alpha[i] = float(np_dtype(a))
        if a == 1.0:
            mask |= 1 << i
    c.alpha_is_one, c.n_times, c.n_norm_seg = mask, len(tab.alpha), n_norm_seg
    return c


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.float32, torch.float64], ids=["f32", "f64"])
@pytest.mark.parametrize("n", [5, 4099, (1 << 20) + 3])
@pytest.mark.parametrize("scale", [1e-3, 1.0, 30.0, 0.0, float("nan")], ids=["small", "unit", "large", "zero", "nan"])
def test_controller_kernel_vs_oracle(hip_kernels, oracle_kernels, dtype, n, scale):
    """Sums as tdeq_error_norm_partial (same kernels); accept / ratio / t0' exact; dt_next and the stage times to
    4 ulp of fp64 (the GPU's pow vs libm's); ctrl_dev consistent with dt_next."""
    g = torch.Generator().manual_seed(n)
    y0 = torch.randn(n, generator=g, dtype=torch.float64).to(dtype)
    y1 = y0 + 0.01 * torch.randn(n, generator=g, dtype=torch.float64).to(dtype)
    part = (torch.randn(n, generator=g, dtype=torch.float64) * 1e-7 * scale).to(dtype)
    k6 = torch.randn(n, generator=g, dtype=torch.float64).to(dtype) * (0.0 if scale == 0.0 else 1e-7 * scale)
    if math.isnan(scale):
        part = torch.randn(n, generator=g, dtype=torch.float64).to(dtype)
        part[0] = float("nan")
    chunk = 1024
    seg = [(0, n, 1e-6, 1e-8)]
    plan_d = hip_kernels.make_plan(seg, n, chunk, torch.device("cuda:0"))
    plan_o = oracle_kernels.make_plan(seg, n, chunk, None)
    np_dtype = np.float32 if dtype == torch.float32 else np.float64
    for tab, order, sign, t0, dt, lo, hi in [(DOPRI5, 5, 1.0, 0.37, 0.0123, 0.0, math.inf),
                                             (DOPRI8, 8, -1.0, -2.5, 0.31, 0.0, math.inf),
                                             (DOPRI5, 5, 1.0, 1e3, 0.5, 0.1, 0.4),      # dt > max_step: forced reject
                                             (DOPRI5, 5, 1.0, 0.0, 0.05, 0.05, 1.0)]:   # dt <= min_step: forced accept
        c = _ctrl(t0, dt, order, tab, sign, lo, hi, np_dtype=np_dtype)
        dts = float(np_dtype(dt)) * sign
        tn_d = torch.empty(c.n_times, dtype=dtype, device="cuda")
        yd, y1d, pd, kd = y0.cuda(), y1.cuda(), part.cuda(), k6.cuda()
        hip_kernels.error_norm_partial_ctrl(plan_d, pd, yd, y1d, [kd], [0.025], dts, c, tn_d)
        accept, dt_next, ratio, bad = hip_kernels.read_ctrl(plan_d)
        full = hip_kernels._read_out(plan_d)          # [sumsq | - | nonfinite | accept, dt_next, ratio, t0']
        sums_ctrl, t0_next = full[:1], full[6]
        # same sums as the plain entry point
        hip_kernels.error_norm_partial(plan_d, pd, yd, y1d, [kd], [0.025], dts)
        sums_plain, _, bad_plain = hip_kernels.read_norms(plan_d)
        assert np.array_equal(np.array(sums_ctrl), np.array(sums_plain), equal_nan=True)
        assert bad == bad_plain
        # controller restated on the host from the device's sums
        tn_o = torch.empty(c.n_times, dtype=dtype)
        out_ctrl, ctrl_dev_o = oracle_kernels.step_controller(plan_o, sums_plain, c, tn_o, dtype)
        assert accept == (out_ctrl[0] != 0.0)
        assert np.array_equal([ratio, t0_next], [out_ctrl[2], out_ctrl[3]], equal_nan=True)
        if math.isnan(out_ctrl[1]):
            assert math.isnan(dt_next)
        else:
            assert abs(dt_next - out_ctrl[1]) <= 4 * np.spacing(abs(out_ctrl[1]))
        ctrl_dev = plan_d.ctrl_dev.cpu().tolist()
        assert ctrl_dev[0] == ctrl_dev_o[0]
        ulp = float(np.spacing(np_dtype(abs(ctrl_dev_o[1]))))
        assert abs(ctrl_dev[1] - ctrl_dev_o[1]) <= 4 * ulp     # the GPU's pow vs libm's: a few ulp of fp64
        tol_t = 2 * float(np.spacing(np_dtype(max(abs(t0) + dt * 11, 1e-30))))
        assert float((tn_d.cpu().double() - tn_o.double()).abs().max()) <= tol_t
        if ctrl_dev[1] == ctrl_dev_o[1]:
            assert torch.equal(tn_d.cpu(), tn_o)      # identical dt' => identical stage times


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.float32, torch.float64], ids=["f32", "f64"])
@pytest.mark.parametrize("n", [1, 3, 1024, 65536 + 7])
def test_stage_combine_sel_bit_exact(hip_kernels, oracle_kernels, dtype, n):
    g = torch.Generator().manual_seed(7)
    ts = [torch.randn(n, generator=g, dtype=torch.float64).to(dtype) for _ in range(4)]
    td = [t.cuda() for t in ts]
    plan_d = hip_kernels.make_plan([(0, n, 1e-6, 1e-8)], n, 1024, torch.device("cuda:0"))
    plan_o = oracle_kernels.make_plan([(0, n, 1e-6, 1e-8)], n, 1024, None)
    np_dtype = np.float32 if dtype == torch.float32 else np.float64
    for accept in (0.0, 1.0):
        for dt in (0.0371, -0.25):
            dtT = float(np_dtype(dt))
            plan_d.ctrl_dev[:2].copy_(torch.tensor([accept, dtT], dtype=torch.float64))
            plan_o.ctrl_dev.copy_(torch.tensor([accept, dtT], dtype=torch.float64))
            out_d, out_o = torch.empty_like(td[0]), torch.empty_like(ts[0])
            hip_kernels.stage_combine_sel(out_d, td[0], td[1], td[2], td[3], 0.2, plan_d)
            oracle_kernels.stage_combine_sel(out_o, ts[0], ts[1], ts[2], ts[3], 0.2, plan_o)
            assert torch.equal(out_d.cpu(), out_o)
            # ... and equal to the host-driven stage_combine on the selected pair
            ref = torch.empty_like(td[0])
            ysel, fsel = (td[0], td[1]) if accept else (td[2], td[3])
            hip_kernels.stage_combine(ref, ysel, [fsel], [0.2], dtT)
            assert torch.equal(out_d, ref)


@pytest.mark.gpu
def test_lookahead_under_every_readback_mode(hip_kernels, monkeypatch):
    """The controller's words travel with the norm results in all three read-back modes (poll on pinned words,
    pinned + stream sync, device buffer + copy): same decisions, same solution."""
    fn, y0 = _problem("vdp", torch.float64, "cuda")
    t = torch.tensor([0.0, 2.0, 5.0], dtype=torch.float64, device="cuda")
    monkeypatch.setenv("TDEQ_LOOKAHEAD", "1")
    outs = []
    for mode in ("poll", "pinned", "copy"):
        monkeypatch.setenv("TDEQ_READBACK", mode)
        kern = _native.HipKernels(hip_kernels.lib)
        monkeypatch.setattr(_native, "get_kernels", lambda device, dtype=None, _k=kern: _k)
        f = _Counting(fn)
        with torch.no_grad():
            y = tda.odeint(f, y0, t, method="dopri5", rtol=1e-6, atol=1e-8, options=dict(first_step=0.7))
        outs.append((y.cpu(), f.nfe))
    assert outs[0][1] == outs[1][1] == outs[2][1]
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][0], outs[2][0])


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
@pytest.mark.parametrize("n_seg,skip_tail,plant_nan", [(17, 0, False), (40, 0, False), (40, 37, False), (23, 0, True)])
def test_controller_kernel_many_segments(hip_kernels, oracle_kernels, dtype, n_seg, skip_tail, plant_nan):
    """More segments than the inline table holds (r02, ABI 14): per-segment sums by the parallel finalize launch,
    controller on them.  Sums equal tdeq_error_norm_partial's bit for bit; ratio / accept / t0' equal the oracle's
    controller on those sums (a max over segments: order-independent; NaN wins); seminorm (`n_norm_seg` < n_seg) too."""
    chunk = 1024
    g = torch.Generator().manual_seed(n_seg)
    numels = [int(v) for v in torch.randint(1, 3 * chunk, (n_seg,), generator=g)]
    numels[1] = 5 * chunk + 17                      # one big segment
    offs, off = [], 0
    for m in numels:
        offs.append(off)
        off += -(-m // chunk) * chunk
    total = off
    segs = [(o, m, 1e-6 * (1 + i % 3), 1e-8) for i, (o, m) in enumerate(zip(offs, numels))]
    y0 = torch.randn(total, generator=g, dtype=torch.float64).to(dtype)
    y1 = y0 + 0.01 * torch.randn(total, generator=g, dtype=torch.float64).to(dtype)
    part = (torch.randn(total, generator=g, dtype=torch.float64) * 3e-7).to(dtype)
    k6 = (torch.randn(total, generator=g, dtype=torch.float64) * 1e-7).to(dtype)
    if plant_nan:
        part[offs[20] + 3] = float("nan")
    plan_d = hip_kernels.make_plan(segs, total, chunk, torch.device("cuda:0"))
    plan_o = oracle_kernels.make_plan(segs, total, chunk, None)
    assert plan_d.segs_dev is not None
    np_dtype = np.float32 if dtype == torch.float32 else np.float64
    c = _ctrl(0.37, 0.0123, 5, DOPRI5, 1.0, 0.0, math.inf, np_dtype=np_dtype)
    c.n_norm_seg = n_seg - skip_tail
    dts = float(np_dtype(0.0123))
    tn_d = torch.empty(c.n_times, dtype=dtype, device="cuda")
    yd, y1d, pd, kd = y0.cuda(), y1.cuda(), part.cuda(), k6.cuda()
    hip_kernels.error_norm_partial_ctrl(plan_d, pd, yd, y1d, [kd], [0.025], dts, c, tn_d)
    accept, dt_next, ratio, bad = hip_kernels.read_ctrl(plan_d)
    full = hip_kernels._read_out(plan_d)
    sums_ctrl, t0_next = full[:n_seg], full[3 * n_seg + 3]
    hip_kernels.error_norm_partial(plan_d, pd, yd, y1d, [kd], [0.025], dts)
    sums_plain, _, bad_plain = hip_kernels.read_norms(plan_d)
    assert np.array_equal(np.array(sums_ctrl), np.array(sums_plain), equal_nan=True)
    assert bad == bad_plain          # (the census counts non-finite STATE entries; a NaN error shows up in the sums)
    tn_o = torch.empty(c.n_times, dtype=dtype)
    out_ctrl, ctrl_dev_o = oracle_kernels.step_controller(plan_o, sums_plain, c, tn_o, dtype)
    assert accept == (out_ctrl[0] != 0.0)
    assert np.array_equal([ratio, t0_next], [out_ctrl[2], out_ctrl[3]], equal_nan=True)
    if plant_nan and skip_tail == 0:
        assert math.isnan(ratio) and not accept
    if not math.isnan(out_ctrl[1]):
        assert abs(dt_next - out_ctrl[1]) <= 4 * np.spacing(abs(out_ctrl[1]))
    assert torch.allclose(tn_d.cpu().double(), tn_o.double(), rtol=1e-6 if dtype == torch.float32 else 1e-14, equal_nan=True)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
@pytest.mark.parametrize("n_seg", [1, 3, 20])
def test_step_controller_on_device_sums_equals_fused_controller(hip_kernels, dtype, n_seg):
    """tdeq_step_controller (ABI 15; the lock-step path: sums in device memory, all-reduced there, then the controller)
    at world size 1 — a sum over one rank — must reproduce tdeq_error_norm_partial_ctrl word for word: sums, census,
    accept, dt_next, ratio, t0', ctrl_dev and the next stage times."""
    from torchdiffeq_amd import _native
    chunk = 1024
    g = torch.Generator().manual_seed(100 + n_seg)
    numels = [int(v) for v in torch.randint(1, 4 * chunk, (n_seg,), generator=g)]
    offs, off = [], 0
    for m in numels:
        offs.append(off)
        off += (-(-m // chunk) * chunk) if n_seg > 1 else m
    total = off
    segs = [(o, m, 1e-6, 1e-8) for o, m in zip(offs, numels)]
    dev = torch.device("cuda:0")
    y0 = torch.randn(total, generator=g, dtype=torch.float64).to(dtype).cuda()
    y1 = (y0.cpu().double() + 0.01 * torch.randn(total, generator=g, dtype=torch.float64)).to(dtype).cuda()
    part = (torch.randn(total, generator=g, dtype=torch.float64) * 2e-7).to(dtype).cuda()
    k6 = (torch.randn(total, generator=g, dtype=torch.float64) * 1e-7).to(dtype).cuda()
    np_dtype = np.float32 if dtype == torch.float32 else np.float64
    c = _ctrl(0.37, 0.0123, 5, DOPRI5, 1.0, 0.0, math.inf, np_dtype=np_dtype)
    c.n_norm_seg = n_seg
    dts = float(np_dtype(0.0123))
    plan_a = hip_kernels.make_plan(segs, total, chunk, dev)
    tn_a = torch.empty(c.n_times, dtype=dtype, device=dev)
    hip_kernels.error_norm_partial_ctrl(plan_a, part, y0, y1, [k6], [0.025], dts, c, tn_a)
    ref_words = hip_kernels._read_out(plan_a)
    ref_dev = plan_a.ctrl_dev.cpu().tolist()
    plan_b = hip_kernels.make_plan(segs, total, chunk, dev)                 # host-visible results
    plan_sums = _native.NormPlan(segs, total, chunk, dev, pinned=False)     # sums in device memory
    tn_b = torch.empty(c.n_times, dtype=dtype, device=dev)
    hip_kernels.error_norm_partial(plan_sums, part, y0, y1, [k6], [0.025], dts)
    hip_kernels.step_controller(plan_b, plan_sums, plan_b, c, tn_b, dtype)
    words = hip_kernels._read_out(plan_b)
    n = n_seg
    assert np.array_equal(np.array(words[:n]), np.array(ref_words[:n]))                      # sums
    assert words[2 * n:3 * n] == ref_words[2 * n:3 * n]                                      # census
    assert np.array_equal(np.array(words[3 * n:3 * n + 4]), np.array(ref_words[3 * n:3 * n + 4]), equal_nan=True)
    assert plan_b.ctrl_dev.cpu().tolist() == ref_dev
    assert torch.equal(tn_a, tn_b)


@pytest.mark.parametrize("lookahead", [True, False])
def test_max_num_steps_is_exceeded_after_the_references_number_of_evaluations(cpu_backend, monkeypatch, lookahead):
    """rk_common.py:243-249: the budget is checked at the head of every trial step.  The look-ahead must not have
    evaluated func for a trial step the budget no longer allows: the reference raises after 2 + 20 x 6 = 122 evaluations
    (found by tools/fuzz_api_programs_vs_reference.py, blow-up family under TDEQ_FUZZ_DEVICE=oracle: 123 before r04b).
    Host logic over the oracle — the same `_step_until` / `_adaptive_step` the HIP path runs."""
    monkeypatch.setenv("TDEQ_LOOKAHEAD", "1" if lookahead else "0")
    f = _Counting(lambda t, y: y * y)
    with pytest.raises(AssertionError, match=r"max_num_steps exceeded \(20>=20\)"):
        with torch.no_grad():
            tda.odeint(f, torch.tensor([1.0, 0.7, 1.3], dtype=torch.float64), torch.tensor([0.0, 1.0, 3.0], dtype=torch.float64),
                       method="dopri5", rtol=1e-3, atol=1e-9, options=dict(max_num_steps=20))
    assert f.nfe == 122



@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.float32, torch.float64], ids=["f32", "f64"])
@pytest.mark.parametrize("layout", ["single", "single_unaligned", "segmented"])
@pytest.mark.parametrize("n_terms", [1, 2])
def test_norm_launch_also_writes_the_last_stage_for_captured_steps(hip_kernels, dtype, layout, n_terms):
    """r06, ABI 21 `copy_last_k`: the captured step's norm launch (state in device memory) writes the last remaining stage
    — the step's f1 — into a second buffer, replacing the N-word copy node.  Same sums and controller words as without the
    copy, the buffer an exact copy (alignment padding of a segmented layout included), neighbours untouched."""
    g = torch.Generator().manual_seed(3)
    chunk = 1024
    if layout == "segmented":
        numels = [1, 1500, 4099]
        offs, total = [], 0
        for m in numels:
            offs.append(total)
            total += -(-m // chunk) * chunk
        segs = [(o, m, 1e-6, 1e-8) for o, m in zip(offs, numels)]
    else:
        total = 5003
        segs = [(0, total, 1e-6, 1e-8)]
    pad = 3 if layout == "single_unaligned" else 0          # views at an odd offset: the scalar path

    def buf():
        return torch.randn(total + pad + 8, generator=g, dtype=torch.float64).to(dtype).cuda()[pad:pad + total]
    y0, y1, part = buf(), buf(), buf() * 1e-7
    ks = [buf() * 1e-7 for _ in range(n_terms)]
    y1 = y0 + 0.01 * y1
    plan = hip_kernels.make_plan(segs, total, chunk, torch.device("cuda:0"))
    np_dtype = np.float32 if dtype == torch.float32 else np.float64
    c = _ctrl(0.37, 0.0123, 5, DOPRI5, 1.0, 0.0, math.inf, np_dtype=np_dtype, n_norm_seg=len(segs))
    tn = torch.empty(c.n_times, dtype=dtype, device="cuda")
    coefs = [0.025, -0.0125][:n_terms]

    def run(copy_to):
        plan.ctrl_dev.copy_(torch.tensor([0.0, float(np_dtype(0.0123)), 0.37, 0.0123], dtype=torch.float64))
        hip_kernels.error_norm_partial_ctrl(plan, part, y0, y1, ks, coefs, 0.0, c, tn, state_in_dev=True, copy_last_to=copy_to)
        words = hip_kernels.read_ctrl(plan)
        return words, list(hip_kernels._read_out(plan)), tn.clone(), plan.ctrl_dev.clone()
    ref = run(None)
    whole = torch.full((total + pad + 16,), 7.0, dtype=dtype, device="cuda")
    dst = whole[pad + 8:pad + 8 + total]
    got = run(dst)
    assert got[0] == ref[0] and np.array_equal(np.array(got[1]), np.array(ref[1]), equal_nan=True)
    assert torch.equal(got[2], ref[2]) and torch.equal(got[3], ref[3])
    assert torch.equal(dst, ks[-1])
    assert bool((whole[:pad + 8] == 7.0).all()) and bool((whole[pad + 8 + total:] == 7.0).all())
