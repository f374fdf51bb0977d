"""Pin the CPU oracle (oracle/rk_oracle.c, oracle/reference_solver.py) to the reference's own outputs.

Tolerances
  * elementwise kernels: <= 2 ulp of the state dtype — `torch.sum` over the short stage dimension is
    not a left-to-right sum, so bitwise agreement with the reference is not defined;
    rk4 (no reductions anywhere) must be BIT-EXACT.
  * error ratio: the reference reduces in T with ATen's blocked order, the oracle in fp64: 1e-6 (fp32),
    1e-13 (fp64) relative.
  * whole solves: fp64 1e-12; fp32 per case (noise floor, see _cases.SOLVE_CASES).
"""
import math

import numpy as np
import pytest
import torch

from _cases import SOLVE_CASES, T, load, rel_err
from oracle import reference_solver as orc
from oracle.kernels import OracleKernels

KERN = OracleKernels()
ULP = {"f32": 2.0 ** -23, "f64": 2.0 ** -52}


def _ulp_err(a, b, scale, eps):
    return float(np.max(np.abs(a.astype(np.float64) - b.astype(np.float64)) / (np.abs(scale).astype(np.float64) + 1e-300))) / eps


@pytest.mark.parametrize("method", ["dopri5", "dopri8"])
@pytest.mark.parametrize("dname", ["f32", "f64"])
@pytest.mark.parametrize("ops_cls", [orc.NumpyOps, orc.COps], ids=["numpy", "c"])
def test_rk_step_vectors(method, dname, ops_cls):
    """_runge_kutta_step: stage inputs, y1, y1_error; _compute_error_ratio; _interp_fit/_evaluate."""
    z = load("kernels.npz")
    key = f"{method}_{dname}"
    ops = ops_cls()
    tab = orc.tableau(method)
    y0, k = z[f"{key}_y0"], z[f"{key}_k"]
    t0, dt = z[f"{key}_t0_dt"]
    Tn = y0.dtype.type
    dt_T = Tn(dt)
    eps = ULP[dname]
    # stage inputs (rk_common.py:79) — magnitude scale: |y0| + sum |c_j k_j|
    for i, beta in enumerate(tab.beta):
        idx, coef = orc._nz(beta)
        got = ops.combine(y0, [k[j] for j in idx], coef, dt_T)
        ref = z[f"{key}_stage_inputs"][i]
        scale = np.abs(y0) + sum(abs(c * dt) * np.abs(k[j]) for j, c in zip(idx, coef))
        assert _ulp_err(got, ref, scale, eps) <= 2.0, (i,)
    assert np.array_equal(z[f"{key}_stage_inputs"][-1], z[f"{key}_y1"])   # FSAL: y1 is the last stage input
    # stage times and perturbation flags (rk_common.py:72-78) via a recording solver run
    seen = []
    solver = orc.AdaptiveRK(lambda tt, y: (seen.append(float(tt)), k[len(seen)])[1], y0, tab, 1e-3, 1e-4, ops=ops,
                            first_step=float(dt))
    solver.y1, solver.f1, solver.t0, solver.t1, solver.dt, solver.rec = y0, k[0], float(t0), float(t0), float(dt), None
    solver.adaptive_step()
    # the golden times were recorded BEFORE the reference's _PerturbFunc; apply misc.py:185-196 here
    ref_times = []
    for tt, p in zip(z[f"{key}_stage_times"], z[f"{key}_stage_perturb"]):
        tt = Tn(tt)
        ref_times.append(float(np.nextafter(tt, tt - Tn(1)) if p == 1 else tt))
    assert seen == ref_times
    assert list(z[f"{key}_stage_perturb"]) == [1 if a == 1.0 else 0 for a in tab.alpha]
    # error estimate -> ratio (rk_common.py:89, misc.py:80-82)
    idx, coef = orc._nz(tab.c_error)
    rtol, atol = z[f"{key}_rtol_atol"]
    mean_sq, bad = ops.error_ratio_sq(y0, z[f"{key}_y1"], [k[j] for j in idx], coef, dt_T, [rtol], [atol], [(0, y0.size)])
    ratio = float(Tn(math.sqrt(mean_sq[0])))
    assert not bad
    assert ratio == pytest.approx(float(z[f"{key}_error_ratio"]), rel=1e-6 if dname == "f32" else 1e-13)
    # dense output (rk_common.py:363-369, interp.py)
    idx, coef = orc._nz(tab.c_mid)
    t_eval = float(z[f"{key}_t_eval"])
    x = Tn((t_eval - t0) / ((t0 + dt) - t0))
    got = ops.dense_eval(y0, z[f"{key}_y1"], k[0], k[-1], [k[j] for j in idx], coef, dt_T, x)
    ref = z[f"{key}_y_eval"]
    coeffs = z[f"{key}_interp_coeffs"]
    scale = sum(np.abs(c) for c in coeffs) + 64 * (np.abs(y0) + np.abs(z[f"{key}_y1"]))
    assert _ulp_err(got, ref, scale, eps) <= 2.0


@pytest.mark.parametrize("dname", ["f32", "f64"])
def test_interp_fit_planes(dname):
    z = load("kernels.npz")
    for method in ("dopri5", "dopri8"):
        key = f"{method}_{dname}"
        tab = orc.tableau(method)
        y0, k, y1 = T(z[f"{key}_y0"]), T(z[f"{key}_k"]), T(z[f"{key}_y1"])
        dt = float(z[f"{key}_t0_dt"][1])
        idx, coef = orc._nz(tab.c_mid)
        out = torch.empty(5 * y0.numel(), dtype=y0.dtype)
        KERN.interp_fit(out, y0, y1, k[0].contiguous(), k[-1].contiguous(), [k[j].contiguous() for j in idx], coef, dt)
        ref = z[f"{key}_interp_coeffs"]
        got = out.view(5, -1).numpy()
        assert np.array_equal(got[0], ref[0])                       # e = y0
        scale = np.abs(ref).sum(0) + 64 * (np.abs(z[f"{key}_y0"]) + np.abs(z[f"{key}_y1"]))
        for p in range(1, 5):
            assert _ulp_err(got[p], ref[p], scale, ULP[dname]) <= 2.0, p


@pytest.mark.parametrize("dname", ["f32", "f64"])
def test_rk4_is_bit_exact(dname):
    """rk4_alt_step_func has no reductions: stage inputs and y1 are reproduced bit for bit."""
    z = load("kernels.npz")
    y0, k = z[f"rk4_{dname}_y0"], z[f"rk4_{dname}_k"]
    dt = z[f"rk4_{dname}_t0_dt"][1]
    ops = orc.NumpyOps()
    ref_in = z[f"rk4_{dname}_stage_inputs"]
    assert np.array_equal(ops.rk4_stage(1, y0, k[0], None, None, None, dt), ref_in[0])
    assert np.array_equal(ops.rk4_stage(2, y0, k[0], k[1], None, None, dt), ref_in[1])
    assert np.array_equal(ops.rk4_stage(3, y0, k[0], k[1], k[2], None, dt), ref_in[2])
    assert np.array_equal(ops.rk4_stage(4, y0, k[0], k[1], k[2], k[3], dt), z[f"rk4_{dname}_y1"])
    ty0, tk = T(y0), [T(k[j]).contiguous() for j in range(4)]
    for stage, ref in [(1, ref_in[0]), (2, ref_in[1]), (3, ref_in[2]), (4, z[f"rk4_{dname}_y1"])]:
        out = torch.empty_like(ty0)
        KERN.rk4_stage(stage, out, ty0, *[tk[j] if j < stage else None for j in range(4)], float(dt))
        assert np.array_equal(out.numpy(), ref), stage


def test_c_and_numpy_ops_agree_bitwise():
    z = load("kernels.npz")
    a, b = orc.NumpyOps(), orc.COps()
    for key in ("dopri5_f32", "dopri8_f64"):
        tab = orc.tableau(key.split("_")[0])
        y0, k, y1 = z[f"{key}_y0"], z[f"{key}_k"], z[f"{key}_y1"]
        for beta in tab.beta:
            idx, coef = orc._nz(beta)
            assert np.array_equal(a.combine(y0, [k[j] for j in idx], coef, 0.05), b.combine(y0, [k[j] for j in idx], coef, 0.05))
        idx, coef = orc._nz(tab.c_mid)
        args = (y0, y1, k[0], k[-1], [k[j] for j in idx], coef, 0.05, 0.3)
        assert np.array_equal(a.dense_eval(*args), b.dense_eval(*args))
        idx, coef = orc._nz(tab.c_error)
        ra, _ = a.error_ratio_sq(y0, y1, [k[j] for j in idx], coef, 0.05, [1e-3], [1e-4], [(0, y0.size)])
        rb, _ = b.error_ratio_sq(y0, y1, [k[j] for j in idx], coef, 0.05, [1e-3], [1e-4], [(0, y0.size)])
        assert ra[0] == pytest.approx(rb[0], rel=1e-13)


def test_step_controller_matches_reference():
    """_optimal_step_size (misc.py:85-95) incl. ratio = 0, NaN, inf."""
    from torchdiffeq_amd.solvers import optimal_step_size
    z = load("controller.npz")
    for args, ref in zip(z["optimal_step_in"], z["optimal_step_out"]):
        last, ratio, safety, ifactor, dfactor, order = args
        ratio32 = float(np.float32(ratio))
        got = optimal_step_size(float(last), ratio32, float(safety), float(ifactor), float(dfactor), int(order))
        if math.isnan(ref):
            assert math.isnan(got), args
        else:
            assert got == pytest.approx(float(ref), rel=1e-15), args


def test_oracle_device_controller_matches_reference(oracle_kernels):
    """oracle_step_controller (the CPU twin of tdeq_error_norm_partial_ctrl's controller) against the reference's
    `_optimal_step_size` vectors: one segment of one element whose sum of squares is ratio^2, so the controller's
    sqrt(mean) is the golden error ratio (an fp32 tensor in the reference -> T = fp32 here).  Also the accept rule
    (rk_common.py:324-330) and the next trial's times (rk_common.py:268-275, 72-78)."""
    import ctypes
    from torchdiffeq_amd import _native
    from torchdiffeq_amd.tableaus import DOPRI5
    z = load("controller.npz")
    plan = oracle_kernels.make_plan([(0, 1, 1e-6, 1e-8)], 1, 1024, None)
    for args, ref in zip(z["optimal_step_in"], z["optimal_step_out"]):
        last, ratio, safety, ifactor, dfactor, order = args
        ratio32 = float(np.float32(ratio))
        c = _native.StepCtrl()
        c.t0, c.dt, c.safety, c.ifactor, c.dfactor, c.exponent = 0.25, float(last), safety, ifactor, dfactor, 1.0 / order
        c.min_step, c.max_step, c.time_sign = 0.0, math.inf, 1.0
        mask = 0
        for i, a in enumerate(DOPRI5.alpha):
            c.alpha[i] = float(np.float32(a))
            mask |= (1 << i) if a == 1.0 else 0
        c.alpha_is_one, c.n_times, c.n_norm_seg = mask, len(DOPRI5.alpha), 1
        tn = torch.empty(6, dtype=torch.float32)
        sumsq = ratio32 * ratio32 if math.isfinite(ratio32) else ratio32
        out_ctrl, ctrl_dev = oracle_kernels.step_controller(plan, [sumsq], c, tn, torch.float32)
        accept, dt_next, got_ratio, t0n = out_ctrl
        if math.isnan(ref):
            assert math.isnan(dt_next), args
        else:
            assert dt_next == pytest.approx(float(ref), rel=1e-15), args
        if math.isfinite(ratio32):
            assert got_ratio == ratio32
        assert bool(accept) == (ratio32 <= 1.0)
        assert t0n == (0.25 + float(last) if accept else 0.25)
        assert ctrl_dev[0] == accept
        if math.isfinite(dt_next):
            # next trial: T(t0') + alpha_i * T(dt') in fp32; alpha == 1 -> nextafter(T(t0' + dt'), below)
            t0T, dtT = np.float32(t0n), np.float32(dt_next)
            t1T = np.float32(t0n + dt_next)
            want = [np.nextafter(t1T, t1T - np.float32(1)) if a == 1.0 else t0T + np.float32(a) * dtT
                    for a in DOPRI5.alpha]
            assert tn.numpy().tolist() == [float(w) for w in want], args
            assert ctrl_dev[1] == float(dtT)


@pytest.mark.parametrize("dname", ["f32", "f64"])
def test_initial_step_matches_reference(dname):
    z = load("controller.npz")
    A, y0 = z[f"init_{dname}_A"], z[f"init_{dname}_y0"]
    for order, name in [(4, "dopri5"), (7, "dopri8")]:
        for scale_y in (1.0, 1e-7):
            yy = (y0 * y0.dtype.type(scale_y)).reshape(-1)
            fn = lambda tt, y: (y.reshape(y0.shape) @ A.T).reshape(-1)
            solver = orc.AdaptiveRK(fn, yy, orc.tableau(name), 1e-6, 1e-8)
            h = solver.select_initial_step(0.5, fn(0.5, yy))
            ref = float(z[f"init_{dname}_o{order}_s{scale_y}"])
            assert h == pytest.approx(ref, rel=2e-6 if dname == "f32" else 1e-13), (order, scale_y)


def test_solver_cfg1_spiral_rk4():
    """cfg1: 999 rk4 steps of the spiral.  The solver arithmetic is bit-exact (test_rk4_is_bit_exact and
    tests/test_host_logic_cpu.py::test_cfg1_bit_exact, which uses the same torch func as the reference);
    here the FIELD is numpy (`y**3 @ A` rounds differently from torch), so compare to 1e-5."""
    z = load("solves.npz")
    y = orc.odeint(orc.SpiralField(z["cfg1_A"]), z["cfg1_y0"], z["cfg1_t"], method="rk4")
    assert rel_err(y, z["cfg1_y"]) < 1e-5
    assert z["cfg1_y"][-1, 0].tolist() == [-0.4436032772064209, 0.27951884269714355]


@pytest.mark.parametrize("prefix,method,tol", SOLVE_CASES)
def test_solver_cfg2_reduced(prefix, method, tol):
    z = load("solves.npz")
    rtol, atol = z[f"{prefix}_tol"]
    stats = {}
    y = orc.odeint(orc.LinearField(z["cfg2_A"]), z["cfg2_y0"], z[f"{prefix}_t"], method, rtol, atol, stats=stats)
    assert rel_err(y, z[f"{prefix}_y"]) < tol
    assert stats["nfe"] == int(z[f"{prefix}_nfe"])
    assert stats["n_accept"] == len(z[f"{prefix}_accept_dt"]) and stats["n_reject"] == len(z[f"{prefix}_reject_dt"])
    np.testing.assert_allclose(stats["dts"], z[f"{prefix}_accept_dt"], rtol=5e-2)
    # fp32: the embedded error estimate is a cancellation of O(1) stage values down to ~1e-6, so its
    # last bits (and through ratio**(-1/5) the step sizes, at the 1e-3..1e-2 level) depend on the
    # summation order of the 6-term combine — torch.sum's order is not ours.


def test_solver_cfg4_reduced_dopri8_fp64():
    z = load("solves.npz")
    stats = {}
    y = orc.odeint(orc.LinearField(z["cfg4_A"]), z["cfg4_y0"], z["cfg4_t"], "dopri8", 1e-9, 1e-11, stats=stats)
    # The first dopri8 step (dt0 = 0.034) has an embedded error estimate of ~1e-16*|y|, i.e. pure fp64
    # rounding noise, so the second step size is summation-order dependent (1 % here) and two correct
    # implementations agree only to the solve's own accuracy (~10*rtol), not to fp64 eps.
    assert rel_err(y, z["cfg4_y"]) < 1e-7
    assert stats["nfe"] == int(z["cfg4_nfe"])
    np.testing.assert_allclose(stats["dts"], z["cfg4_accept_dt"], rtol=5e-2)


def test_solver_time_dependent_with_rejections():
    z = load("solves.npz")
    A = z["tdep_A"]

    class Field:
        params = []

        def f(self, t, y):
            return np.sin(3 * t) * (y @ A.T) * 4 - y ** 3

    stats = {}
    y = orc.odeint(Field(), z["tdep_y0"], z["tdep_t"], "dopri5", 1e-8, 1e-10, stats=stats)
    assert rel_err(y, z["tdep_y"]) < 1e-10
    assert stats["n_accept"] == len(z["tdep_accept_dt"]) and stats["n_reject"] == len(z["tdep_reject_dt"])
    assert stats["nfe"] == int(z["tdep_nfe"])


@pytest.mark.parametrize("tag,tol", [("f32", 1e-5), ("f64", 1e-12)])
@pytest.mark.parametrize("norm_tag", ["default", "seminorm"])
def test_adjoint_cfg3_reduced(tag, tol, norm_tag):
    z = load("adjoint.npz")
    rtol, atol = [float(v) for v in z[f"adj_{tag}_tol"]]
    field = orc.MLPField([z[f"adj_{tag}_p{i}"] for i in range(6)])
    t = z[f"adj_{tag}_t"]
    y = orc.odeint(field, z[f"adj_{tag}_y0"], t, "dopri5", rtol, atol)
    grad_y = np.zeros_like(y)
    grad_y[-1] = 2 * y[-1]
    if len(t) > 2:
        grad_y[1:] += 1
    _, g0, gp = orc.adjoint_grads(field, z[f"adj_{tag}_y0"], t, grad_y, "dopri5", rtol, atol,
                                  seminorm=(norm_tag == "seminorm"))
    assert rel_err(y, z[f"adj_{tag}_{norm_tag}_y"]) < tol
    assert rel_err(g0, z[f"adj_{tag}_{norm_tag}_grad_y0"]) < tol
    for i, g in enumerate(gp):
        assert rel_err(g, z[f"adj_{tag}_{norm_tag}_grad_p{i}"]) < tol, i


# ---------------------------------------------------------------------------------------------------
# §8(f) rank 2: the other explicit RK methods (tests/golden/methods.npz)
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("method", ["tsit5", "bosh3", "fehlberg2", "adaptive_heun"])
def test_oracle_adaptive_low_order_pairs(method):
    """fp64 well above the noise floor: same accept / reject sequence and step sizes as the reference."""
    z = load("methods.npz")
    A = z["ad_A"]

    class Field:
        params = []

        def f(self, t, y):
            return np.sin(2 * t) * (y @ A.T) * 2 - 0.5 * y ** 3

    rtol, atol = z[f"ad_{method}_tol"]
    stats = {}
    y = orc.odeint(Field(), z["ad_y0"], z["ad_t"], method, rtol, atol, stats=stats)
    assert rel_err(y, z[f"ad_{method}_y"]) < 1e-10
    assert stats["nfe"] == int(z[f"ad_{method}_nfe"])
    assert (stats["n_accept"], stats["n_reject"]) == (len(z[f"ad_{method}_accept_dt"]), len(z[f"ad_{method}_reject_dt"]))
    np.testing.assert_allclose(stats["dts"], z[f"ad_{method}_accept_dt"], rtol=1e-7)


@pytest.mark.parametrize("method", ["euler", "midpoint", "heun2", "heun3", "rk4"])
def test_oracle_fixed_grid_methods(method):
    """The field is numpy here (cos / GEMM round differently from torch's), so 1e-6 (fp32) / 1e-13 (fp64);
    bit-exactness of the solver arithmetic itself is pinned by tests/test_methods_golden.py[cpu], which runs
    the product's host logic over the C oracle with the reference's own torch field."""
    z = load("methods.npz")
    A, y0, t = z["fx_A"], z["fx_y0"], z["fx_t"]

    class Field:
        params = []

        def __init__(self, A):
            self.A = A

        def f(self, tt, y):
            return (np.cos(tt) * (y @ self.A.T) - y.dtype.type(0.1) * y).astype(y.dtype)

    f32 = Field(A)
    cases = {
        "grid": orc.odeint(f32, y0, np.linspace(0, 1, 9, dtype=np.float32), method),
        "step": orc.odeint(f32, y0, t, method, step_size=0.1),
        "perturb": orc.odeint(f32, y0, t, method, step_size=0.1, perturb=True),
        "cubic": orc.odeint(f32, y0, t, method, step_size=0.1, interp="cubic"),
        "rev": orc.odeint(f32, y0, np.array([1.0, 0.45, 0.0], dtype=np.float32), method, step_size=0.125,
                          interp="cubic"),
        "f64": orc.odeint(Field(A.astype(np.float64)), y0.astype(np.float64), t.astype(np.float64), method,
                          step_size=0.05),
    }
    for tag, y in cases.items():
        assert rel_err(y, z[f"fx_{method}_{tag}"]) < (1e-13 if tag == "f64" else 1e-6), tag


def test_oracle_c_fixed_stage_matches_numpy_bitwise():
    z = load("kernels.npz")
    a = orc.NumpyOps()
    for dname in ("f32", "f64"):
        y0, k = z[f"rk4_{dname}_y0"], z[f"rk4_{dname}_k"]
        ty0, tk = T(y0), [T(k[j]).contiguous() for j in range(4)]
        for mode, idx, ws in [(1, [0], [1 / 3]), (0, [1], [2 / 3]), (0, [0, 2], [0.25, 0.75]), (0, [0, 1, 2, 3], [0.1, 0.2, 0.3, 0.4])]:
            out = torch.empty_like(ty0)
            KERN.fixed_stage(mode, out, ty0, [tk[j] for j in idx], ws, -0.037)
            assert np.array_equal(out.numpy(), a.fixed_stage(mode, y0, [k[j] for j in idx], ws, -0.037))
        out = torch.empty_like(ty0)
        KERN.weighted_sum(out, [ty0] + tk, [0.5, -0.25, 0.125, 3.0, 1 / 3])
        assert np.array_equal(out.numpy(), a.weighted_sum([y0] + [k[j] for j in range(4)], [0.5, -0.25, 0.125, 3.0, 1 / 3]))


def test_oracle_fused_end_of_step_pair_is_the_same_arithmetic():
    """oracle_stage_combine_err / oracle_error_norm_partial == oracle_stage_combine / oracle_error_norm."""
    z = load("kernels.npz")
    for key, method in (("dopri5_f32", "dopri5"), ("dopri8_f64", "dopri8")):
        tab = orc.tableau(method)
        y0, k, y1 = T(z[f"{key}_y0"]), [T(z[f"{key}_k"][j]).contiguous() for j in range(len(tab.alpha) + 1)], T(z[f"{key}_y1"])
        idx, coef = orc._nz(tab.beta[-1])
        eidx, ecoef = orc._nz(tab.c_error)
        assert eidx[:len(idx)] == idx
        out, ep, plain = torch.empty_like(y0), torch.empty_like(y0), torch.empty_like(y0)
        KERN.stage_combine_err(out, ep, y0, [k[j] for j in idx], coef, ecoef[:len(idx)], 0.0371)
        KERN.stage_combine(plain, y0, [k[j] for j in idx], coef, 0.0371)
        assert torch.equal(out, plain)
        plan = KERN.make_plan([(0, y0.numel(), 1e-3, 1e-4)], y0.numel(), 1024, None)
        KERN.error_norm_partial(plan, ep, y0, y1, [k[j] for j in eidx[len(idx):]], ecoef[len(idx):], 0.0371)
        a = KERN.read_norms(plan)[0][0]
        KERN.error_norm(plan, y0, y1, [k[j] for j in eidx], ecoef, 0.0371)
        assert a == KERN.read_norms(plan)[0][0]


@pytest.mark.parametrize("dname", ["f32", "f64"])
def test_eager_torch_port_follows_the_numpy_oracle(dname):
    """oracle/eager_torch_port.py (the "reference-style eager PyTorch" timed by bench.py on the GPU) against the
    numpy oracle on the same trial steps: same accept/reject sequence; step sizes to 3e-4 in fp32 (the embedded
    error is a cancelling sum: torch.sum's association over the stage axis vs the oracle's left-to-right one moves it
    by ~1e-4 relative, and the eager port accumulates its norm in T like the reference) and 1e-11 in fp64."""
    from oracle import eager_torch_port as ep
    dtype = np.float32 if dname == "f32" else np.float64
    rng = np.random.default_rng(3)
    A = (rng.standard_normal((8, 8)) / 3 - 0.3 * np.eye(8)).astype(dtype)
    y0 = rng.standard_normal((20, 8)).astype(dtype)
    field = orc.LinearField(A)
    solver = orc.AdaptiveRK(lambda tt, y: field.f(tt, y.reshape(20, 8)).reshape(-1), y0.reshape(-1),
                            orc.tableau("dopri5"), 1e-5, 1e-7, first_step=0.35)
    solver.before_integrate(0.0)
    At = torch.from_numpy(A.T.copy())
    eager = ep.EagerAdaptiveRK(lambda t, y: y @ At, torch.from_numpy(y0.copy()), 0.0, 0.35, 1e-5, 1e-7)
    for _ in range(12):
        solver.adaptive_step()
        eager.adaptive_step()
        assert (eager.n_accepted, eager.n_rejected) == (solver.n_accept, solver.n_reject)
        assert float(eager.dt) == pytest.approx(solver.dt, rel=3e-4 if dname == "f32" else 1e-11)
        assert float(eager.t) == pytest.approx(solver.t1, rel=3e-4 if dname == "f32" else 1e-12)
    tol = 1e-4 if dname == "f32" else 5e-12     # the states are compared at (slightly) different times in fp32
    assert rel_err(eager.y.numpy().reshape(-1), solver.y1) < tol
    assert solver.n_reject > 0          # the large first step exercises the reject branch


@pytest.mark.parametrize("tag,dtype", [("f32", torch.float32), ("f64", torch.float64)])
def test_oracle_adams_kernels_pinned_to_the_reference_expressions(oracle_kernels, tag, dtype):
    """oracle_adams_predict / oracle_adams_correct against outputs of the reference's own expressions
    (fixed_adams.py:205, :210, :213-215 and `_has_converged` :189-192) on random vectors — golden/adams.npz `kv_*`."""
    from torchdiffeq_amd.tableaus import adams_coefficients
    z = load("adams.npz")
    hist = [h.contiguous() for h in T(z[f"kv_{tag}_hist"])]
    y0, dt = T(z[f"kv_{tag}_y0"]), float(z[f"kv_{tag}_dt"])
    order = len(hist)
    bash, _ = adams_coefficients(order)
    _, moulton = adams_coefficients(order + 1)
    y, dy, delta = torch.empty_like(y0), torch.empty_like(y0), torch.empty_like(y0)
    oracle_kernels.adams_predict(y, y0, hist, [dt * b for b in bash], list(moulton[1:]), dt, dy_out=dy, delta_out=delta)
    assert torch.equal(dy, T(z[f"kv_{tag}_dy"]))
    assert torch.equal(delta, T(z[f"kv_{tag}_delta"]))
    assert torch.equal(y, T(z[f"kv_{tag}_ypred"]))
    y_only = torch.empty_like(y0)
    oracle_kernels.adams_predict(y_only, y0, hist, [dt * b for b in bash])
    assert torch.equal(y_only, y)
    n = y0.numel()
    plan = oracle_kernels.make_plan([(0, n, 1e-3, 1e-4)], n, 1024, None)
    y_new, dy_new = torch.empty_like(y0), torch.empty_like(y0)
    oracle_kernels.adams_correct(plan, dy_new, dy, y_out=y_new, f=T(z[f"kv_{tag}_f"]), delta=delta, y0=y0,
                                 c=dt * moulton[0])
    assert torch.equal(dy_new, T(z[f"kv_{tag}_dy_new"]))
    assert torch.equal(y_new, T(z[f"kv_{tag}_y_new"]))
    converged = lambda: oracle_kernels.read_norms(plan)[0] == [0.0]
    assert converged() == bool(z[f"kv_{tag}_converged_far"])
    oracle_kernels.adams_correct(plan, dy_new, T(z[f"kv_{tag}_dy_close"]), compute=False)
    assert oracle_kernels.read_norms(plan)[0] == [float(z[f"kv_{tag}_violations"])]     # the exact census
    assert converged() == bool(z[f"kv_{tag}_converged_close"])
    oracle_kernels.adams_correct(plan, dy_new, dy_new, compute=False)
    assert converged() == bool(z[f"kv_{tag}_converged_same"]) is True


@pytest.mark.parametrize("tag", ["f32", "f64"])
@pytest.mark.parametrize("method", ["dopri5", "dopri8", "bosh3"])
def test_eager_torch_port_is_the_reference_op_for_op(tag, method):
    """r03: oracle/eager_torch_port.py stands in for "the unmodified reference on cuda"  in bench.py.  Its
    fidelity AS AN OP SEQUENCE is pinned here to the reference itself: on one CPU thread the port must reproduce the
    reference's accepted AND rejected (t0, dt) pairs — fp64 values that have gone through every stage sum, the fp32 /
    fp64 error norm and the controller — BIT FOR BIT (tests/golden/eager_pin.npz <- make_golden.py eager_pin: the
    reference's own callbacks, `first_step` given).  In fp32 the error estimate is rounding-level, so any difference in
    operation order would show."""
    from oracle import eager_torch_port as ep
    z = load("eager_pin.npz")
    key = f"{tag}_{method}"
    A, y0 = T(z[key + "_A"]), T(z[key + "_y0"])
    rtol, atol, first = [float(v) for v in z[key + "_tol"]]
    threads = torch.get_num_threads()
    torch.set_num_threads(1)
    try:
        e = ep.EagerAdaptiveRK(lambda t, y: y @ A.T, y0, 0.0, first, rtol, atol, method)
        acc, rej = [], []
        with torch.no_grad():
            while float(e.t) < 2.0 and len(acc) + len(rej) < 1000:
                t0, dt = float(e.t), float(e.dt)
                (acc if e.adaptive_step() else rej).append((t0, dt))
    finally:
        torch.set_num_threads(threads)
    assert np.array_equal(np.array(acc).reshape(-1, 2), z[key + "_acc"])
    assert np.array_equal(np.array(rej).reshape(-1, 2), z[key + "_rej"])
    assert len(rej) > 0
