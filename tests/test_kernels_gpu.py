"""Kernel parity on the MI355X: every C-ABI entry point vs the CPU oracle on the same seeded inputs.

Tolerances: the element arithmetic of both sides is the same sequence of individually rounded T
operations, so elementwise outputs must agree BIT-EXACTLY; reductions accumulate in fp64 with a
different association on the two sides, so sums agree to ~1e-13 relative."""
import math

import numpy as np
import pytest
import torch

from torchdiffeq_amd.tableaus import ADAPTIVE_HEUN, BOSH3, DOPRI5, DOPRI8, TSIT5, SparseRow

pytestmark = pytest.mark.gpu

DTYPES = [torch.float32, torch.float64]
SIZES = [1, 3, 255, 1024, 4099, 65536 + 7, 1 << 20]


def _rand(n, dtype, seed, offset=0):
    g = torch.Generator().manual_seed(seed)
    base = torch.randn(n + offset, generator=g, dtype=torch.float64).to(dtype)
    return base[offset:] if offset else base


def _dev(ts):
    return [t.cuda() for t in ts]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("n", SIZES)
@pytest.mark.parametrize("tab", [DOPRI5, DOPRI8], ids=["dopri5", "dopri8"])
def test_stage_combine_all_rows(hip_kernels, oracle_kernels, dtype, n, tab):
    S = tab.n_stages
    y0 = _rand(n, dtype, 1)
    ks = [_rand(n, dtype, 10 + j) for j in range(S + 1)]
    y0d, ksd = y0.cuda(), _dev(ks)
    for sign in (1.0, -1.0):
        dt = 0.0371 * sign
        for row in tab.beta_rows():
            out_ref = torch.empty_like(y0)
            oracle_kernels.stage_combine(out_ref, y0, [ks[j] for j in row.idx], row.coef, dt)
            out = torch.empty_like(y0d)
            hip_kernels.stage_combine(out, y0d, [ksd[j] for j in row.idx], row.coef, dt)
            assert torch.equal(out.cpu(), out_ref)


@pytest.mark.parametrize("dtype", DTYPES)
def test_stage_combine_unaligned(hip_kernels, oracle_kernels, dtype):
    n = 10007
    row = DOPRI5.beta_rows()[4]
    y0 = _rand(n, dtype, 1, offset=1)          # views at odd element offsets -> scalar path
    ks = [_rand(n, dtype, 10 + j, offset=1 + (j % 3)) for j in range(7)]
    y0c, ksc = y0.contiguous(), [k.contiguous() for k in ks]
    ref = torch.empty_like(y0c)
    oracle_kernels.stage_combine(ref, y0c, [ksc[j] for j in row.idx], row.coef, 0.1)
    base_y = torch.empty(n + 1, dtype=dtype).cuda()
    y0d = base_y[1:]
    y0d.copy_(y0c)
    ksd = []
    for j, k in enumerate(ksc):
        buf = torch.empty(n + 3, dtype=dtype).cuda()
        v = buf[1 + (j % 3):1 + (j % 3) + n]
        v.copy_(k)
        ksd.append(v)
    out = torch.empty(n + 1, dtype=dtype).cuda()[1:]
    hip_kernels.stage_combine(out, y0d, [ksd[j] for j in row.idx], row.coef, 0.1)
    assert torch.equal(out.cpu(), ref)


def _plan_pair(hip_kernels, oracle_kernels, segments, total, chunk):
    return (hip_kernels.make_plan(segments, total, chunk, torch.device("cuda:0")),
            oracle_kernels.make_plan(segments, total, chunk, None))


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("n", SIZES)
@pytest.mark.parametrize("chunk", [1024, 2048, 4096])
def test_error_norm_single_segment(hip_kernels, oracle_kernels, dtype, n, chunk):
    for tab in (DOPRI5, DOPRI8):
        S = tab.n_stages
        err = SparseRow.from_dense(tab.c_error)
        y0, y1 = _rand(n, dtype, 1), _rand(n, dtype, 2)
        ks = [_rand(n, dtype, 10 + j) for j in range(S + 1)]
        segs = [(0, n, 1e-3, 1e-4)]
        pg, pc = _plan_pair(hip_kernels, oracle_kernels, segs, n, chunk)
        oracle_kernels.error_norm(pc, y0, y1, [ks[j] for j in err.idx], err.coef, 0.0371)
        ref, _, bad_ref = oracle_kernels.read_norms(pc)
        ksd = _dev(ks)
        hip_kernels.error_norm(pg, y0.cuda(), y1.cuda(), [ksd[j] for j in err.idx], err.coef, 0.0371)
        got, _, bad = hip_kernels.read_norms(pg)
        assert bad == [0.0] and bad_ref == [0.0]
        assert got[0] == pytest.approx(ref[0], rel=1e-12)


@pytest.mark.parametrize("dtype", DTYPES)
def test_error_norm_segments_and_scaled(hip_kernels, oracle_kernels, dtype):
    chunk = 1024
    numels = [5000, 1, 1024, 77, 3000]
    offs, off = [], 0
    for m in numels:
        offs.append(off)
        off += math.ceil(m / chunk) * chunk
    total = off
    segs = [(o, m, 1e-3 * (i + 1), 1e-5 * (i + 1)) for i, (o, m) in enumerate(zip(offs, numels))]
    err = SparseRow.from_dense(DOPRI5.c_error)
    y0, y1 = _rand(total, dtype, 1), _rand(total, dtype, 2)
    ks = [_rand(total, dtype, 10 + j) for j in range(7)]
    # poison the padding: it must be ignored by the norms
    mask = torch.ones(total, dtype=torch.bool)
    for o, m in zip(offs, numels):
        mask[o:o + m] = False
    for tns in [y0, y1] + ks:
        tns[mask] = float("nan")
    pg, pc = _plan_pair(hip_kernels, oracle_kernels, segs, total, chunk)
    sc_ref = torch.full((total,), 7.0, dtype=dtype)
    oracle_kernels.error_scaled(pc, sc_ref, y0, y1, [ks[j] for j in err.idx], err.coef, -0.02)
    ref, _, bad_ref = oracle_kernels.read_norms(pc)
    ksd = _dev(ks)
    sc = torch.full((total,), 7.0, dtype=dtype).cuda()
    hip_kernels.error_scaled(pg, sc, y0.cuda(), y1.cuda(), [ksd[j] for j in err.idx], err.coef, -0.02)
    got, _, bad = hip_kernels.read_norms(pg)
    assert bad == bad_ref == [0.0] * len(numels)
    assert got == pytest.approx(ref, rel=1e-12)
    assert torch.equal(sc.cpu(), sc_ref)
    # and the plain (non-writing) kernel gives the same sums
    hip_kernels.error_norm(pg, y0.cuda(), y1.cuda(), [ksd[j] for j in err.idx], err.coef, -0.02)
    got2, _, _ = hip_kernels.read_norms(pg)
    assert got2 == got


def test_error_norm_many_segments_device_table(hip_kernels, oracle_kernels):
    chunk, dtype = 1024, torch.float32
    numels = [((i * 37) % 2500) + 1 for i in range(40)]     # > TDEQ_INLINE_SEGMENTS
    offs, off = [], 0
    for m in numels:
        offs.append(off)
        off += math.ceil(m / chunk) * chunk
    total = off
    segs = [(o, m, 1e-3, 1e-6) for o, m in zip(offs, numels)]
    err = SparseRow.from_dense(DOPRI5.c_error)
    y0, y1 = _rand(total, dtype, 1), _rand(total, dtype, 2)
    ks = [_rand(total, dtype, 10 + j) for j in range(7)]
    pg, pc = _plan_pair(hip_kernels, oracle_kernels, segs, total, chunk)
    oracle_kernels.error_norm(pc, y0, y1, [ks[j] for j in err.idx], err.coef, 0.3)
    ref, _, _ = oracle_kernels.read_norms(pc)
    ksd = _dev(ks)
    hip_kernels.error_norm(pg, y0.cuda(), y1.cuda(), [ksd[j] for j in err.idx], err.coef, 0.3)
    got, _, bad = hip_kernels.read_norms(pg)
    assert got == pytest.approx(ref, rel=1e-12)
    assert sum(bad) == 0


def test_error_norm_nonfinite_census(hip_kernels, oracle_kernels):
    n, dtype = 5000, torch.float32
    err = SparseRow.from_dense(DOPRI5.c_error)
    y0, y1 = _rand(n, dtype, 1), _rand(n, dtype, 2)
    y1[17] = float("inf")
    y0[4000] = float("nan")
    ks = [_rand(n, dtype, 10 + j) for j in range(7)]
    pg, pc = _plan_pair(hip_kernels, oracle_kernels, [(0, n, 1e-3, 1e-6)], n, 1024)
    ksd = _dev(ks)
    hip_kernels.error_norm(pg, y0.cuda(), y1.cuda(), [ksd[j] for j in err.idx], err.coef, 0.3)
    got, _, bad = hip_kernels.read_norms(pg)
    assert bad == [2.0]          # the census, not the (fmax-based) ratio, is what flags the state
    pc = oracle_kernels.make_plan([(0, n, 1e-3, 1e-6)], n, 1024, None)
    oracle_kernels.error_norm(pc, y0, y1, [ks[j] for j in err.idx], err.coef, 0.3)
    ref, _, bad_ref = oracle_kernels.read_norms(pc)
    assert bad_ref == [2.0] and got[0] == pytest.approx(ref[0], rel=1e-12)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("n", [1, 1000, 4099, 1 << 20])
def test_init_norms(hip_kernels, oracle_kernels, dtype, n):
    a, b, y = _rand(n, dtype, 1), _rand(n, dtype, 2), _rand(n, dtype, 3)
    pg, pc = _plan_pair(hip_kernels, oracle_kernels, [(0, n, 1e-5, 1e-7)], n, 2048)
    for mode in (0, 1):
        oracle_kernels.init_norms(pc, mode, a, b, y)
        r0, r1, _ = oracle_kernels.read_norms(pc)
        hip_kernels.init_norms(pg, mode, a.cuda(), b.cuda(), y.cuda())
        g0, g1, bad = hip_kernels.read_norms(pg)
        assert g0 == pytest.approx(r0, rel=1e-12)
        if mode == 0:
            assert g1 == pytest.approx(r1, rel=1e-12)
        assert bad == [0.0]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("n", [1, 1001, 4099, 1 << 20])
@pytest.mark.parametrize("tab", [DOPRI5, DOPRI8], ids=["dopri5", "dopri8"])
def test_dense_eval_and_fit(hip_kernels, oracle_kernels, dtype, n, tab):
    S = tab.n_stages
    mid = SparseRow.from_dense(tab.c_mid)
    y0, y1 = _rand(n, dtype, 1), _rand(n, dtype, 2)
    ks = [_rand(n, dtype, 10 + j) for j in range(S + 1)]
    ksd = _dev(ks)
    for dt, x in [(0.05, 0.3), (-0.05, 1.0), (0.2, 0.0)]:
        ref = torch.empty_like(y0)
        oracle_kernels.dense_eval(ref, y0, y1, ks[0], ks[-1], [ks[j] for j in mid.idx], mid.coef, dt, x)
        out = torch.empty_like(y0).cuda()
        hip_kernels.dense_eval(out, y0.cuda(), y1.cuda(), ksd[0], ksd[-1], [ksd[j] for j in mid.idx], mid.coef, dt, x)
        assert torch.equal(out.cpu(), ref)
    cref = torch.empty(5 * n, dtype=dtype)
    oracle_kernels.interp_fit(cref, y0, y1, ks[0], ks[-1], [ks[j] for j in mid.idx], mid.coef, 0.05)
    cout = torch.empty(5 * n, dtype=dtype).cuda()
    hip_kernels.interp_fit(cout, y0.cuda(), y1.cuda(), ksd[0], ksd[-1], [ksd[j] for j in mid.idx], mid.coef, 0.05)
    assert torch.equal(cout.cpu(), cref)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("n", [2, 1001, 4099, 1 << 20])
def test_rk4_and_lerp(hip_kernels, oracle_kernels, dtype, n):
    y0 = _rand(n, dtype, 1)
    k = [_rand(n, dtype, 10 + j) for j in range(4)]
    kd = _dev(k)
    for dt in (0.025, -0.025):
        for stage in (1, 2, 3, 4):
            args = [k[j] if j < stage else None for j in range(4)]
            argsd = [kd[j] if j < stage else None for j in range(4)]
            ref = torch.empty_like(y0)
            oracle_kernels.rk4_stage(stage, ref, y0, *args, dt)
            out = torch.empty_like(y0).cuda()
            hip_kernels.rk4_stage(stage, out, y0.cuda(), *argsd, dt)
            assert torch.equal(out.cpu(), ref)
    ref = torch.empty_like(y0)
    oracle_kernels.lerp(ref, y0, k[0], 0.37)
    out = torch.empty_like(y0).cuda()
    hip_kernels.lerp(out, y0.cuda(), kd[0], 0.37)
    assert torch.equal(out.cpu(), ref)


def test_readback_modes_agree(hip_kernels, monkeypatch):
    """poll (spin on pinned words), pinned (stream sync) and copy (device buffer + memcpy) read-backs
    return the same numbers, repeatedly (the poll sentinel is re-armed per launch)."""
    from torchdiffeq_amd import _native
    n, dtype = 100000, torch.float32
    err = SparseRow.from_dense(DOPRI5.c_error)
    y0, y1 = _rand(n, dtype, 1).cuda(), _rand(n, dtype, 2).cuda()
    ks = _dev([_rand(n, dtype, 10 + j) for j in range(7)])
    res = []
    for mode in ("poll", "pinned", "copy"):
        monkeypatch.setenv("TDEQ_READBACK", mode)
        kern = _native.HipKernels(hip_kernels.lib)
        plan = kern.make_plan([(0, n, 1e-3, 1e-6)], n, 2048, torch.device("cuda:0"))
        for rep in range(3):
            kern.error_norm(plan, y0, y1, [ks[j] for j in err.idx], err.coef, 0.1 * (rep + 1))
            out = kern.read_norms(plan)
            kern.init_norms(plan, 0, y0, y1, y0)
            out2 = kern.read_norms(plan)
        res.append((out, out2))
    assert res[0] == res[1] == res[2]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("n", SIZES)
def test_fixed_stage_and_weighted_sum(hip_kernels, oracle_kernels, dtype, n):
    """tdeq_fixed_stage (rk2/rk3 step forms) and tdeq_weighted_sum (cubic Hermite interpolation): bit-exact."""
    y0 = _rand(n, dtype, 1)
    ks = [_rand(n, dtype, 10 + j) for j in range(8)]
    y0d, ksd = y0.cuda(), _dev(ks)
    for dt in (0.0371, -0.0371):
        for mode, idx, ws in [(1, [0], [1 / 3]), (1, [2], [1.0]), (0, [1], [2 / 3]), (0, [0, 1], [0.5, 0.5]),
                              (0, [0, 2], [0.25, 0.75]), (0, [0, 1, 2], [0.1, 0.2, 0.7]), (0, [0, 1, 2, 3], [0.1, 0.2, 0.3, 0.4])]:
            ref = torch.empty_like(y0)
            oracle_kernels.fixed_stage(mode, ref, y0, [ks[j] for j in idx], ws, dt)
            out = torch.empty_like(y0d)
            hip_kernels.fixed_stage(mode, out, y0d, [ksd[j] for j in idx], ws, dt)
            assert torch.equal(out.cpu(), ref), (mode, idx)
    for nt in range(1, 9):
        ws = [(-1) ** j * (0.3 + j / 7) for j in range(nt)]
        ref = torch.empty_like(y0)
        oracle_kernels.weighted_sum(ref, ks[:nt], ws)
        out = torch.empty_like(y0d)
        hip_kernels.weighted_sum(out, ksd[:nt], ws)
        assert torch.equal(out.cpu(), ref), nt


@pytest.mark.parametrize("dtype", DTYPES)
def test_fixed_stage_unaligned(hip_kernels, oracle_kernels, dtype):
    n = 10007
    y0, k1, k2 = _rand(n, dtype, 1), _rand(n, dtype, 2), _rand(n, dtype, 3)
    ref = torch.empty_like(y0)
    oracle_kernels.fixed_stage(0, ref, y0, [k1, k2], [0.25, 0.75], 0.1)
    bufs = [torch.empty(n + 3, dtype=dtype).cuda() for _ in range(4)]
    views = [b[1 + i % 3:1 + i % 3 + n] for i, b in enumerate(bufs)]
    for v, src in zip(views[1:], (y0, k1, k2)):
        v.copy_(src)
    hip_kernels.fixed_stage(0, views[0], views[1], [views[2], views[3]], [0.25, 0.75], 0.1)
    assert torch.equal(views[0].cpu(), ref)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("n", SIZES)
def test_scale_many_and_multi_dot(hip_kernels, oracle_kernels, dtype, n):
    """Backward helpers of the differentiable odeint: outs[m] = w_m g (bit-exact), <g, x_m> in fp64 (1e-12)."""
    g = _rand(n, dtype, 3)
    xs = [_rand(n, dtype, 20 + j) for j in range(14)]
    gd, xsd = g.cuda(), _dev(xs)
    for nt in (1, 2, 5, 7, 14):
        ws = [(-1) ** j * (0.37 + j / 11) for j in range(nt)]
        ref = [torch.empty_like(g) for _ in range(nt)]
        oracle_kernels.scale_many(ref, g, ws)
        outs = [torch.empty_like(gd) for _ in range(nt)]
        hip_kernels.scale_many(outs, gd, ws)
        for o, r in zip(outs, ref):
            assert torch.equal(o.cpu(), r)
        dref = oracle_kernels.multi_dot(g, xs[:nt])
        d = hip_kernels.multi_dot(gd, xsd[:nt])
        assert d.dtype == torch.float64 and d.shape == (nt,)
        scale = torch.stack([(g.double().abs() * x.double().abs()).sum() for x in xs[:nt]])
        assert float(((d.cpu() - dref).abs() / (scale + 1e-300)).max()) < 1e-13


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("n", [1, 255, 4099, 1 << 20])
def test_stage_combine_fill(hip_kernels, oracle_kernels, dtype, n):
    """First-stage combine fused with the stage-time fill == the two separate calls, bit for bit."""
    y0, f0, k1 = _rand(n, dtype, 1), _rand(n, dtype, 2), _rand(n, dtype, 3)
    vals = [0.1 * j + 1e-9 for j in range(13)]
    for ks, coefs in (([f0], [0.2]), ([f0, k1], [3 / 40, 9 / 40])):
        ref, tref = torch.empty_like(y0), torch.empty(13, dtype=dtype)
        oracle_kernels.stage_combine_fill(ref, y0, ks, coefs, -0.05, tref, vals)
        out, tb = torch.empty_like(y0).cuda(), torch.full((16,), -7.0, dtype=dtype).cuda()
        hip_kernels.stage_combine_fill(out, y0.cuda(), _dev(ks), coefs, -0.05, tb, vals)
        assert torch.equal(out.cpu(), ref)
        assert torch.equal(tb[:13].cpu(), tref) and torch.all(tb[13:] == -7.0)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("n", [1, 255, 4099, 65536 + 7, 1 << 20])
@pytest.mark.parametrize("tab", [DOPRI5, DOPRI8, BOSH3, TSIT5, ADAPTIVE_HEUN], ids=lambda t: t.name)
def test_fused_end_of_step_pair(hip_kernels, oracle_kernels, dtype, n, tab):
    """tdeq_stage_combine_err + tdeq_error_norm_partial == tdeq_stage_combine + tdeq_error_norm:
    elementwise outputs bit for bit, and the SAME fp64 sums (same per-element values, same chunk order)."""
    S = tab.n_stages
    last = tab.beta_rows()[-1] if tab.fsal_solution else SparseRow.from_dense(tab.c_sol)
    err = SparseRow.from_dense(tab.c_error)
    n_lead = len(last.idx)
    assert err.idx[:n_lead] == last.idx
    y0, y1 = _rand(n, dtype, 1), _rand(n, dtype, 2)
    ks = [_rand(n, dtype, 10 + j) for j in range(S + 1)]
    y0d, y1d, ksd = y0.cuda(), y1.cuda(), _dev(ks)
    dt = -0.0371
    # oracle, fused
    out_ref, ep_ref = torch.empty_like(y0), torch.empty_like(y0)
    oracle_kernels.stage_combine_err(out_ref, ep_ref, y0, [ks[j] for j in last.idx], last.coef, err.coef[:n_lead], dt)
    out, ep = torch.empty_like(y0d), torch.empty_like(y0d)
    hip_kernels.stage_combine_err(out, ep, y0d, [ksd[j] for j in last.idx], last.coef, err.coef[:n_lead], dt)
    assert torch.equal(out.cpu(), out_ref) and torch.equal(ep.cpu(), ep_ref)
    plain = torch.empty_like(y0d)
    hip_kernels.stage_combine(plain, y0d, [ksd[j] for j in last.idx], last.coef, dt)
    assert torch.equal(plain, out)
    segs = [(0, n, 1e-3, 1e-4)]
    for chunk in (1024, 2048):
        pg, pc = _plan_pair(hip_kernels, oracle_kernels, segs, n, chunk)
        rest_idx, rest_coef = err.idx[n_lead:], err.coef[n_lead:]
        hip_kernels.error_norm_partial(pg, ep, y0d, y1d, [ksd[j] for j in rest_idx], rest_coef, dt)
        fused, _, bad_f = hip_kernels.read_norms(pg)
        hip_kernels.error_norm(pg, y0d, y1d, [ksd[j] for j in err.idx], err.coef, dt)
        unfused, _, bad_u = hip_kernels.read_norms(pg)
        assert fused == unfused and bad_f == bad_u == [0.0]
        oracle_kernels.error_norm_partial(pc, ep_ref, y0, y1, [ks[j] for j in rest_idx], rest_coef, dt)
        ref, _, _ = oracle_kernels.read_norms(pc)
        assert fused[0] == pytest.approx(ref[0], rel=1e-12)


def test_stage_combine_timed_same_result_and_plausible_time(hip_kernels):
    """Measurement hook: identical output to stage_combine; the dispatch-stamped events give a duration between the
    HBM-roofline time of the launch and the event -> launch -> event bracket around the same kernel."""
    n = 1 << 22
    row = DOPRI5.beta_rows()[4]
    y0 = _rand(n, torch.float32, 1).cuda()
    ks = _dev([_rand(n, torch.float32, 10 + j) for j in range(7)])
    sel = [ks[j] for j in row.idx]
    ref, out = torch.empty_like(y0), torch.empty_like(y0)
    hip_kernels.stage_combine(ref, y0, sel, row.coef, 0.1)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    e1.record()
    for _ in range(3):
        hip_kernels.stage_combine_timed(out, y0, sel, row.coef, 0.1, e0, e1)
    torch.cuda.synchronize()
    assert torch.equal(out, ref)
    ms = e0.elapsed_time(e1)
    b0, b1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    b0.record()
    hip_kernels.stage_combine(out, y0, sel, row.coef, 0.1)
    b1.record()
    torch.cuda.synchronize()
    bracket = b0.elapsed_time(b1)
    # lower bound: the algorithmic bytes at 3x the 8 TB/s HBM peak — at this size (7 x 16 MB) the streams were just
    # written and sit in the 256 MiB Infinity Cache, so the launch may beat the HBM rate (seen: 0.97x the HBM-peak time)
    floor_ms = (len(sel) + 2) * n * 4 / (3 * 8.0e12) * 1e3
    assert floor_ms < ms <= bracket * 1.25, (floor_ms, ms, bracket)      # slack: clocks ramp between the two


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("chunk", [1024, 2048])
def test_pack_segments_bit_exact(hip_kernels, oracle_kernels, dtype, chunk):
    """tdeq_pack_segments vs the oracle and vs the torch ops it replaces (cat / neg / zeros): segments of ragged
    sizes (1 element, sub-chunk, multi-chunk, empty, missing), +-1 scales, unaligned source views, zeroed padding."""
    numels = [1, 5, chunk, 3 * chunk + 17, 0, 700, 2 * chunk]
    scales = [-1.0, 1.0, -1.0, 1.0, 1.0, -1.0, 1.0]
    g = torch.Generator().manual_seed(5)
    srcs = []
    for i, m in enumerate(numels):
        if i == 5:
            srcs.append(None)                                   # absent gradient -> zeros
        elif i == 3:
            srcs.append(torch.randn(m + 1, generator=g, dtype=torch.float64).to(dtype)[1:])   # odd offset: scalar path
        else:
            srcs.append(torch.randn(m, generator=g, dtype=torch.float64).to(dtype))
    starts, off = [], 0
    for m in numels:
        starts.append(off // chunk)
        off += max(1, math.ceil(m / chunk)) * chunk
    ref = torch.full((off,), float("nan"), dtype=dtype)
    oracle_kernels.pack_segments(ref, [None if t is None else t.contiguous() for t in srcs], starts, numels, scales, chunk)
    # the torch ops being replaced
    want = torch.zeros(off, dtype=dtype)
    for t, st, m, sc in zip(srcs, starts, numels, scales):
        if t is not None and m:
            want[st * chunk:st * chunk + m] = t * sc
    assert torch.equal(ref, want)
    dev_srcs = []
    for t in srcs:
        if t is None:
            dev_srcs.append(None)
        elif t.storage_offset():
            buf = torch.empty(t.numel() + 1, dtype=dtype, device="cuda")
            buf[1:].copy_(t)
            dev_srcs.append(buf[1:])
        else:
            dev_srcs.append(t.cuda())
    out = torch.full((off,), float("nan"), dtype=dtype, device="cuda")
    hip_kernels.pack_segments(out, dev_srcs, starts, numels, scales, chunk)
    assert torch.equal(out.cpu(), ref)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("n", [1, 3, 1023, 4099, 65536 + 7])
@pytest.mark.parametrize("tab", [DOPRI5, DOPRI8], ids=["dopri5", "dopri8"])
def test_dense_eval_multi_rows_equal_single_calls(hip_kernels, oracle_kernels, dtype, n, tab):
    """tdeq_dense_eval_multi: every output row bit-identical to tdeq_dense_eval at the same x and to the oracle;
    row strides that keep / break 16-byte alignment; 1..16 outputs."""
    S = tab.n_stages
    y0, y1 = _rand(n, dtype, 1), _rand(n, dtype, 2)
    ks = [_rand(n, dtype, 10 + j) for j in range(S + 1)]
    mid = SparseRow.from_dense(tab.c_mid)
    y0d, y1d, ksd = y0.cuda(), y1.cuda(), _dev(ks)
    for m in (1, 2, 5, 16):
        xs = [float(torch.tensor((q + 0.5) / m, dtype=dtype)) for q in range(m)]
        rows = torch.full((m, n), float("nan"), dtype=dtype, device="cuda")
        hip_kernels.dense_eval_multi(rows, y0d, y1d, ksd[0], ksd[-1], [ksd[j] for j in mid.idx], mid.coef, -0.07, xs)
        ref = torch.empty(m, n, dtype=dtype)
        oracle_kernels.dense_eval_multi(ref, y0, y1, ks[0], ks[-1], [ks[j] for j in mid.idx], mid.coef, -0.07, xs)
        assert torch.equal(rows.cpu(), ref)
        single = torch.empty(n, dtype=dtype, device="cuda")
        for q, x in enumerate(xs):
            hip_kernels.dense_eval(single, y0d, y1d, ksd[0], ksd[-1], [ksd[j] for j in mid.idx], mid.coef, -0.07, x)
            assert torch.equal(rows[q], single)


# ---- Adams–Bashforth(–Moulton): tdeq_adams_predict / tdeq_adams_correct ----------------------------------
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("n", [1, 255, 4099, 65536 + 7, 1 << 20])
def test_adams_predict_bit_exact(hip_kernels, oracle_kernels, dtype, n):
    from torchdiffeq_amd.tableaus import adams_coefficients
    y0 = _rand(n, dtype, 1)
    hist = [_rand(n, dtype, 20 + j) for j in range(14)]
    y0d, histd = y0.cuda(), _dev(hist)
    for order in (1, 3, 4, 7, 11, 14):
        bash, _ = adams_coefficients(order)
        _, moulton = adams_coefficients(order + 1)
        for dt in (0.0371, -0.0123):
            cb = [dt * b for b in bash]
            # explicit: only y0 + dy
            ref = torch.empty_like(y0)
            oracle_kernels.adams_predict(ref, y0, hist[:order], cb)
            out = torch.empty_like(y0d)
            hip_kernels.adams_predict(out, y0d, histd[:order], cb)
            assert torch.equal(out.cpu(), ref)
            # implicit: dy and the corrector's constant part from the same pass
            refs = [torch.empty_like(y0) for _ in range(3)]
            oracle_kernels.adams_predict(refs[0], y0, hist[:order], cb, list(moulton[1:]), dt, dy_out=refs[1],
                                         delta_out=refs[2])
            outs = [torch.empty_like(y0d) for _ in range(3)]
            hip_kernels.adams_predict(outs[0], y0d, histd[:order], cb, list(moulton[1:]), dt, dy_out=outs[1],
                                      delta_out=outs[2])
            for o, r in zip(outs, refs):
                assert torch.equal(o.cpu(), r)


@pytest.mark.parametrize("dtype", DTYPES)
def test_adams_predict_unaligned(hip_kernels, oracle_kernels, dtype):
    n = 10007
    y0 = _rand(n, dtype, 1)
    hist = [_rand(n, dtype, 30 + j) for j in range(5)]
    cb, cm = [0.3, -0.2, 0.1, 0.05, -0.01], [0.5, 0.25, -0.125, 0.0625, 0.03]
    refs = [torch.empty_like(y0) for _ in range(3)]
    oracle_kernels.adams_predict(refs[0], y0, hist, cb, cm, 0.07, dy_out=refs[1], delta_out=refs[2])

    def view(src, off):          # a view at an odd element offset -> scalar path
        buf = torch.empty(n + 3, dtype=dtype).cuda()
        v = buf[off:off + n]
        v.copy_(src)
        return v
    histd = [view(h, 1 + j % 3) for j, h in enumerate(hist)]
    outs = [view(torch.zeros(n, dtype=dtype), 1) for _ in range(3)]
    hip_kernels.adams_predict(outs[0], view(y0, 1), histd, cb, cm, 0.07, dy_out=outs[1], delta_out=outs[2])
    for o, r in zip(outs, refs):
        assert torch.equal(o.cpu(), r)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("n", [1, 255, 4099, 65536 + 7, 1 << 20])
@pytest.mark.parametrize("chunk", [1024, 2048])
def test_adams_correct_single_segment(hip_kernels, oracle_kernels, dtype, n, chunk):
    rtol, atol = 1e-3, 1e-4
    pg, pc = _plan_pair(hip_kernels, oracle_kernels, [(0, n, rtol, atol)], n, chunk)
    y0, f, delta = _rand(n, dtype, 1), _rand(n, dtype, 2), _rand(n, dtype, 3)
    c = 0.0173
    # dy_old close to the new value for most elements, far for a few: a non-trivial census
    dy_ref_full = (f.double() * c + delta.double()).to(dtype)
    noise = _rand(n, dtype, 4)
    dy_old = dy_ref_full + noise * 1e-4 * (noise.abs() > 1.5)
    y_ref, dy_ref = torch.empty_like(y0), torch.empty_like(y0)
    oracle_kernels.adams_correct(pc, dy_ref, dy_old, y_out=y_ref, f=f, delta=delta, y0=y0, c=c)
    cnt_ref, _, bad_ref = oracle_kernels.read_norms(pc)
    y, dy = torch.empty_like(y0).cuda(), torch.empty_like(y0).cuda()
    hip_kernels.adams_correct(pg, dy, dy_old.cuda(), y_out=y, f=f.cuda(), delta=delta.cuda(), y0=y0.cuda(), c=c)
    cnt, _, bad = hip_kernels.read_norms(pg)
    assert torch.equal(dy.cpu(), dy_ref) and torch.equal(y.cpu(), y_ref)
    assert cnt == cnt_ref and bad == bad_ref == [0.0]        # an exact census: no tolerance
    # census-only mode on the same pair
    hip_kernels.adams_correct(pg, dy, dy_old.cuda(), compute=False)
    cnt2, _, _ = hip_kernels.read_norms(pg)
    assert cnt2 == cnt
    # identical iterates -> converged; a NaN anywhere -> not converged (torch.max propagates NaN, NaN < 1 is false)
    hip_kernels.adams_correct(pg, dy, dy, compute=False)
    assert hip_kernels.read_norms(pg)[0] == [0.0]
    dy_nan = dy.clone()
    dy_nan[n // 2] = float("nan")
    hip_kernels.adams_correct(pg, dy_nan, dy, compute=False)
    cnt3, _, bad3 = hip_kernels.read_norms(pg)
    assert cnt3 == [1.0] and bad3 == [1.0]


@pytest.mark.parametrize("dtype", DTYPES)
def test_adams_correct_segments_and_padding(hip_kernels, oracle_kernels, dtype):
    chunk = 1024
    numels = [5000, 1, 1024, 77, 3000]
    offs, off = [], 0
    for m in numels:
        offs.append(off)
        off += math.ceil(m / chunk) * chunk
    total = off
    segs = [(o, m, 10.0 ** -(i + 2), 10.0 ** -(i + 4)) for i, (o, m) in enumerate(zip(offs, numels))]
    pg, pc = _plan_pair(hip_kernels, oracle_kernels, segs, total, chunk)
    y0, f, delta = _rand(total, dtype, 1), _rand(total, dtype, 2), _rand(total, dtype, 3)
    mask = torch.ones(total, dtype=torch.bool)
    for o, m in zip(offs, numels):
        mask[o:o + m] = False
    for tns in (y0, f, delta):
        tns[mask] = 0.0                       # the padding of a segmented state is zero in every input
    c = -0.031
    dy_full = (f.double() * c + delta.double()).to(dtype)
    dy_old = dy_full * (1 + 3e-4 * _rand(total, dtype, 5))
    dy_old[mask] = 0.0
    y_ref, dy_ref = torch.empty_like(y0), torch.empty_like(y0)
    oracle_kernels.adams_correct(pc, dy_ref, dy_old, y_out=y_ref, f=f, delta=delta, y0=y0, c=c)
    cnt_ref, _, _ = oracle_kernels.read_norms(pc)
    y, dy = torch.full_like(y0, 7.0).cuda(), torch.full_like(y0, 7.0).cuda()
    hip_kernels.adams_correct(pg, dy, dy_old.cuda(), y_out=y, f=f.cuda(), delta=delta.cuda(), y0=y0.cuda(), c=c)
    cnt, _, bad = hip_kernels.read_norms(pg)
    assert torch.equal(dy.cpu(), dy_ref) and torch.equal(y.cpu(), y_ref)      # padding written (zeros) as well
    assert cnt == cnt_ref and bad == [0.0] * len(numels)
    assert 0 < sum(cnt) < sum(numels)      # tolerances differ per segment: a mixed census


# ---------------------------------------------------------------------------------------------------
# hipGraph-mode kernels: step size read from device memory
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("n", [5, 4099, (1 << 17) - 3, 1 << 17, (1 << 17) + 5, (1 << 20) + 1])
@pytest.mark.parametrize("offset", [0, 1])
def test_stage_combine_dev_equals_host_dt_kernels(hip_kernels, dtype, n, offset):
    """tdeq_stage_combine_dev (dt = ctrl_dev[1] on the device; run-time-term kernel below 2^17 elements, the
    templated 16-byte-per-lane kernel from there on; `offset` = a view that is not 16-byte aligned -> scalar path)
    must give the bits of tdeq_stage_combine / tdeq_stage_combine_err with the same dt in the kernel arguments —
    which are pinned to the oracle above.  Every dopri5 and dopri8 row."""
    dev = torch.device("cuda:0")
    dt = -0.0371 if offset else 0.0371
    T = np.float32 if dtype == torch.float32 else np.float64
    plan = hip_kernels.make_plan([(0, n, 1e-5, 1e-7)], n, 1024, dev)
    plan.ctrl_dev.copy_(torch.tensor([1.0, float(T(dt)), 0.0, abs(dt)], dtype=torch.float64))
    for tab in (DOPRI5, DOPRI8):
        S = len(tab.alpha)
        y0 = _rand(n, dtype, 1, offset).cuda()
        ks = [_rand(n, dtype, 10 + j, offset).cuda() for j in range(S + 1)]
        c_err = SparseRow.from_dense(tab.c_error)
        for row in tab.beta_rows():
            kk = [ks[j] for j in row.idx]
            ref, out = torch.empty_like(y0), torch.empty_like(y0)
            hip_kernels.stage_combine(ref, y0, kk, row.coef, float(T(dt)))
            hip_kernels.stage_combine_dev(out, None, y0, kk, row.coef, None, plan)
            assert torch.equal(out, ref), (tab.name if hasattr(tab, "name") else S, len(kk))
            ecoef = [0.37 * (j + 1) for j in range(len(kk))]
            ref_e, out_e = torch.empty_like(y0), torch.empty_like(y0)
            hip_kernels.stage_combine_err(ref, ref_e, y0, kk, row.coef, ecoef, float(T(dt)))
            hip_kernels.stage_combine_dev(out, out_e, y0, kk, row.coef, ecoef, plan)
            assert torch.equal(out, ref) and torch.equal(out_e, ref_e)
        del c_err


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("n,offset", [(7, 0), (4099, 0), (4099, 1), ((1 << 18) + 3, 0)])
def test_step_commit(hip_kernels, dtype, n, offset):
    """tdeq_step_commit: on accept (y_prev, f_prev) <- (y_cur, f_cur), (y_cur, f_cur) <- (y1, f1); on reject nothing."""
    dev = torch.device("cuda:0")
    plan = hip_kernels.make_plan([(0, n, 1e-5, 1e-7)], n, 1024, dev)
    bufs = [_rand(n, dtype, 20 + j, offset).cuda() for j in range(6)]
    for accept in (0.0, 1.0):
        y_prev, f_prev, y_cur, f_cur, y1, f1 = [b.clone() for b in bufs]
        if offset:      # clones are aligned: rebuild unaligned views
            y_prev, f_prev, y_cur, f_cur, y1, f1 = [torch.cat([b.new_zeros(1), b])[1:] for b in
                                                    (y_prev, f_prev, y_cur, f_cur, y1, f1)]
        plan.ctrl_dev.copy_(torch.tensor([accept, 0.1, 0.0, 0.1], dtype=torch.float64))
        hip_kernels.step_commit(y_prev, f_prev, y_cur, f_cur, y1, f1, plan)
        exp = [bufs[2], bufs[3], bufs[4], bufs[5]] if accept else [bufs[0], bufs[1], bufs[2], bufs[3]]
        for got, want in zip((y_prev, f_prev, y_cur, f_cur), exp):
            assert torch.equal(got, want)
        assert torch.equal(y1, bufs[4]) and torch.equal(f1, bufs[5])
