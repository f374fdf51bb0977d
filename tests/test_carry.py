"""Carried partial sums (tableaus.carry_plan, tdeq_stage_combine_multi): the planned launches must reproduce the
row-by-row `_runge_kutta_step` (rk_common.py:69-89) BIT FOR BIT — every stage input, the partial embedded error, and
therefore every decision of a solve — while moving fewer words."""
import os

import numpy as np
import pytest
import torch

from torchdiffeq_amd import tableaus as tb


def _rand(n, dtype, seed, special=True):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(n, generator=g, dtype=torch.float64, device="cpu").to(dtype)
    if special and n >= 8:          # signed zeros and an exact cancellation partner: the cases a `0 + x` start would change
        x[0], x[1], x[2] = 0.0, -0.0, -0.0
    return x


@pytest.mark.parametrize("name,words,launches,before", [("dopri5", 35, 7, 37), ("dopri8", 75, 13, 98), ("tsit5", 41, 8, 46)])
def test_plan_word_counts(name, words, launches, before):
    plan = tb.carry_plan(name)
    tab = tb.ADAPTIVE_TABLEAUS[name]
    assert (plan.words, plan.launches) == (words, launches)
    assert tb.row_by_row_words(tab) == before
    rows = tab.beta_rows()
    S = len(rows) if tab.fsal_solution else len(rows) + 1          # launch rows (tsit5: + the c_sol combine)
    # structure: every row's stage input is produced exactly once; the error partial exactly once
    produced = [0]
    for op in plan.ops:
        if op is None:
            continue
        assert len(op.targets) <= tb.MAX_MULTI_OUT and op.targets[0] == op.row
        for t, (coefs, mask, done) in zip(op.targets, op.spec):
            assert mask and mask < (1 << len(op.idx)) and len(coefs) == len(op.idx)
            if done:
                produced.append(t)
    assert sorted(produced) == list(range(S))
    assert sum(1 for op in plan.ops if op is not None and S in op.targets) == 1
    assert tb.carry_plan("bosh3") is None and tb.carry_plan("fehlberg2") is None


def _run_rows(kern, tab, y0, ks, dt, planned):
    """Stage inputs y_1..y_{S-1} [+ the solution y1 of a non-FSAL pair] (given ALL stages up front — the combines are
    linear in them, so parity of the launch forms does not need a func) + the partial error and what is left to the
    norm kernel."""
    rows = tab.beta_rows()
    if not tab.fsal_solution:
        rows = rows + [tb.SparseRow.from_dense(tab.c_sol)]
    S = len(rows)
    err = tb.SparseRow.from_dense(tab.c_error)
    ys = {}
    if not planned:
        for i in range(1, S):
            r = rows[i]
            ys[i] = torch.empty_like(y0)
            if i == S - 1:
                ep = torch.empty_like(y0)
                kern.stage_combine_err(ys[i], ep, y0, [ks[j] for j in r.idx], r.coef, err.coef[:len(r.idx)], dt)
            else:
                kern.stage_combine(ys[i], y0, [ks[j] for j in r.idx], r.coef, dt)
        return ys, ep, err.idx[len(rows[-1].idx):]
    plan = tb.carry_plan(tab.name)
    held = {}
    for i in range(1, S):
        op = plan.ops[i]
        if op is None:
            ys[i] = held.pop(i)
            continue
        outs = [torch.empty_like(y0) for _ in op.targets]
        kern.stage_combine_multi(outs, op.spec, y0, held.pop(i) if op.continues else None, [ks[j] for j in op.idx], dt)
        ys[i] = outs[0]
        for t, b in zip(op.targets[1:], outs[1:]):
            held[t] = b
    ep = held.pop(S)
    assert not held
    return ys, ep, plan.err_idx


def _bits(x):
    return x.cpu().contiguous().view(torch.int32 if x.dtype == torch.float32 else torch.int64)


@pytest.mark.parametrize("name", ["dopri5", "dopri8", "tsit5"])
@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
@pytest.mark.parametrize("n", [1, 7, 1024, 4099])
@pytest.mark.parametrize("dt", [0.37, -0.011])
def test_planned_rows_equal_row_by_row_oracle(oracle_kernels, name, dtype, n, dt):
    tab = tb.ADAPTIVE_TABLEAUS[name]
    y0 = _rand(n, dtype, 1)
    ks = [_rand(n, dtype, 10 + j) for j in range(len(tab.beta) + 1)]
    a, ea, ra = _run_rows(oracle_kernels, tab, y0, ks, dt, planned=False)
    b, eb, rb = _run_rows(oracle_kernels, tab, y0, ks, dt, planned=True)
    for i in a:
        assert torch.equal(_bits(a[i]), _bits(b[i])), (name, i)
    # the error: (partial + remaining stages) must be the same left-to-right sum either way
    err = tb.SparseRow.from_dense(tab.c_error)
    T = np.float32 if dtype == torch.float32 else np.float64

    def finish(ep, rem):
        e = ep.numpy().copy()
        for j in rem:
            c = T(T(dict(zip(err.idx, err.coef))[j]) * T(dt))
            e = (e + ks[j].numpy() * c).astype(T)
        return e
    assert np.array_equal(finish(ea, ra).view(np.int32 if T is np.float32 else np.int64),
                          finish(eb, rb).view(np.int32 if T is np.float32 else np.int64))


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["dopri5", "dopri8", "tsit5"])
@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
@pytest.mark.parametrize("n", [1, 5, 1023, 65536 + 3, 1 << 20])
@pytest.mark.parametrize("dt", [0.37, -0.011])
def test_planned_rows_equal_row_by_row_hip_and_oracle(hip_kernels, oracle_kernels, name, dtype, n, dt):
    tab = tb.ADAPTIVE_TABLEAUS[name]
    y0 = _rand(n, dtype, 1)
    ks = [_rand(n, dtype, 10 + j) for j in range(len(tab.beta) + 1)]
    ref, eref, _ = _run_rows(oracle_kernels, tab, y0, ks, dt, planned=False)
    dev = torch.device("cuda:0")
    y0d, ksd = y0.to(dev), [k.to(dev) for k in ks]
    for planned in (False, True):
        got, ep, _ = _run_rows(hip_kernels, tab, y0d, ksd, dt, planned=planned)
        for i in ref:
            assert torch.equal(_bits(ref[i]), _bits(got[i])), (name, i, planned)
    got, ep, rem = _run_rows(hip_kernels, tab, y0d, ksd, dt, planned=True)
    o, eo, remo = _run_rows(oracle_kernels, tab, y0, ks, dt, planned=True)
    assert rem == remo and torch.equal(_bits(eo), _bits(ep))


@pytest.mark.gpu
def test_multi_unaligned_views_take_the_scalar_path(hip_kernels, oracle_kernels):
    tab = tb.DOPRI8
    n = 3001
    dev = torch.device("cuda:0")
    y0 = _rand(n + 1, torch.float32, 3)
    ks = [_rand(n + 1, torch.float32, 30 + j) for j in range(14)]
    ref, eref, _ = _run_rows(oracle_kernels, tab, y0[1:].clone(), [k[1:].clone() for k in ks], 0.1, planned=True)
    got, eg, _ = _run_rows(hip_kernels, tab, y0.to(dev)[1:], [k.to(dev)[1:] for k in ks], 0.1, planned=True)
    for i in ref:
        assert torch.equal(_bits(ref[i]), _bits(got[i]))
    assert torch.equal(_bits(eref), _bits(eg))


def test_multi_rejects_bad_arguments():
    import ctypes
    from torchdiffeq_amd import _native
    lib = _native.load_library()
    spec = (_native.MultiOut * 1)()
    spec[0].out, spec[0].mask = 16, 0            # empty mask
    k = (ctypes.c_void_p * 1)(16)
    assert lib.tdeq_stage_combine_multi(spec, 1, 16, None, k, 1, 0.1, 8, 0, None) == -1
    spec[0].mask = 0b10                          # bit outside n_terms
    assert lib.tdeq_stage_combine_multi(spec, 1, 16, None, k, 1, 0.1, 8, 0, None) == -1
    spec[0].mask = 1
    assert lib.tdeq_stage_combine_multi(spec, 5, 16, None, k, 1, 0.1, 8, 0, None) == -1     # too many outputs
    assert lib.tdeq_stage_combine_multi(spec, 1, 16, None, k, 1, 0.1, 0, 0, None) == 0      # n = 0: nothing to do


def _solve(method, dtype, device, carry, monkeypatch, rtol, atol, reverse=False):
    import torchdiffeq_amd as tda
    monkeypatch.setenv("TDEQ_CARRY", "1" if carry else "0")      # "1": every tableau with a plan (dopri5 too)
    g = torch.Generator(device="cpu").manual_seed(5)
    D = 24
    G = torch.randn(D, D, generator=g, dtype=torch.float64, device="cpu") / D ** 0.5
    A = (0.5 * (G - G.T) - 0.1 * torch.eye(D, dtype=torch.float64, device="cpu")).to(dtype).to(device)
    y0 = torch.randn(96, D, generator=g, dtype=torch.float64, device="cpu").to(dtype).to(device)
    t = torch.tensor([0.0, 0.3, 0.7, 1.5] if not reverse else [1.5, 0.7, 0.0], dtype=torch.float64, device=device)
    steps = []

    class F(torch.nn.Module):
        def forward(self, t_, y):
            return torch.tanh(y @ A.T) * 0.5 + y @ A.T

        def callback_accept_step(self, t0, y, dt):
            steps.append(("a", float(t0), float(dt)))

        def callback_reject_step(self, t0, y, dt):
            steps.append(("r", float(t0), float(dt)))
    with torch.no_grad():
        y = tda.odeint(F(), y0, t, method=method, rtol=rtol, atol=atol)
    return y, steps


@pytest.mark.parametrize("method,dtype,rtol,atol", [("dopri5", torch.float32, 1e-6, 1e-8), ("dopri5", torch.float64, 1e-9, 1e-11),
                                                    ("dopri8", torch.float64, 1e-10, 1e-12), ("dopri8", torch.float32, 1e-6, 1e-8),
                                                    ("tsit5", torch.float32, 1e-6, 1e-8), ("tsit5", torch.float64, 1e-9, 1e-11)])
@pytest.mark.parametrize("reverse", [False, True])
def test_solve_with_plan_is_bit_identical(dev, monkeypatch, method, dtype, rtol, atol, reverse):
    a, sa = _solve(method, dtype, dev, False, monkeypatch, rtol, atol, reverse)
    b, sb = _solve(method, dtype, dev, True, monkeypatch, rtol, atol, reverse)
    assert sa == sb and len(sa) > 3
    assert torch.equal(_bits(a), _bits(b))


def test_plan_is_used_by_the_no_grad_solve(cpu_backend, monkeypatch):
    calls = []
    orig = cpu_backend.stage_combine_multi
    monkeypatch.setattr(cpu_backend, "stage_combine_multi", lambda *a, **k: (calls.append(1), orig(*a, **k))[1], raising=False)
    _solve("dopri8", torch.float64, "cpu", True, monkeypatch, 1e-8, 1e-10)
    assert calls
    n = len(calls)
    _solve("dopri8", torch.float64, "cpu", False, monkeypatch, 1e-8, 1e-10)
    assert len(calls) == n


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["dopri5", "dopri8", "tsit5"])
@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
@pytest.mark.parametrize("n", [7, 1023, (1 << 17) + 5])
def test_multi_with_the_step_size_on_the_device_equals_host_dt(hip_kernels, name, dtype, n):
    """tdeq_stage_combine_multi_dev (captured steps: dt = ctrl_dev[1], read by the kernel) against the host-dt entry
    point on every planned launch of the tableau: bit-identical outputs."""
    tab = tb.ADAPTIVE_TABLEAUS[name]
    plan = tb.carry_plan(name)
    dev = torch.device("cuda:0")
    T_ = np.float32 if dtype == torch.float32 else np.float64
    dt = -0.0371
    y0 = _rand(n, dtype, 1).to(dev)
    ks = [_rand(n, dtype, 10 + j).to(dev) for j in range(len(tab.beta) + 1)]
    nplan = hip_kernels.make_plan([(0, n, 1e-3, 1e-6)], n, 1024, dev)
    nplan.ctrl_dev.copy_(torch.tensor([1.0, float(T_(dt)), 0.0, abs(dt)], dtype=torch.float64))
    acc = _rand(n, dtype, 99).to(dev)
    checked = 0
    for op in plan.ops:
        if op is None:
            continue
        a = [torch.empty_like(y0) for _ in op.targets]
        b = [torch.empty_like(y0) for _ in op.targets]
        hip_kernels.stage_combine_multi(a, op.spec, y0, acc if op.continues else None, [ks[j] for j in op.idx], dt)
        hip_kernels.stage_combine_multi_dev(b, op.spec, y0, acc if op.continues else None, [ks[j] for j in op.idx], nplan)
        for x, y in zip(a, b):
            assert torch.equal(_bits(x), _bits(y)), (name, op.row)
        checked += 1
    assert checked >= 5
