"""Generate the golden vectors in tests/golden/*.npz by running the REFERENCE itself.

Run in the build container only (the reference is mounted read-only at /root/reference and does not
exist on the GPU box):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

Every array stored here is an output of rtqichen/torchdiffeq v0.2.5 functions on CPU (or an input fed
to them), so the CPU oracle (oracle/) and the HIP path can be pinned to the reference without the
reference being present.  Nothing from the reference's source is copied — only its numerical outputs.
"""
import math
import os
import sys

import numpy as np
import torch

sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference")

import torchdiffeq  # noqa: E402
from torchdiffeq._impl import adaptive_heun, bosh3, dopri5, dopri8, fehlberg2, interp, misc, rk_common, tsit5  # noqa: E402
from torchdiffeq._impl.misc import Perturb  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
torch.set_num_threads(1)   # one thread: the reference's fp32 reductions depend on the thread count


def save(name, **arrays):
    out = {}
    for k, v in arrays.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    np.savez_compressed(os.path.join(HERE, name), **out)
    print(name, {k: v.shape for k, v in out.items()})


def rand(*shape, seed, dtype=torch.float64):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g, dtype=torch.float64).to(dtype)


# ---------------------------------------------------------------------------------------------------
def gen_tableaus():
    arrays = {}
    for name, tab, mid in [("dopri5", dopri5._DORMAND_PRINCE_SHAMPINE_TABLEAU, dopri5.DPS_C_MID),
                           ("dopri8", dopri8._DOPRI8_TABLEAU, dopri8._C_mid),
                           ("tsit5", tsit5._TSITOURAS_TABLEAU, tsit5.TSIT_C_MID),
                           ("bosh3", bosh3._BOGACKI_SHAMPINE_TABLEAU, bosh3._BS_C_MID),
                           ("fehlberg2", fehlberg2._FEHLBERG2_TABLEAU, fehlberg2._FE_C_MID),
                           ("adaptive_heun", adaptive_heun._ADAPTIVE_HEUN_TABLEAU, adaptive_heun._AH_C_MID)]:
        arrays[f"{name}_alpha"] = tab.alpha
        arrays[f"{name}_beta_flat"] = torch.cat(list(tab.beta))
        arrays[f"{name}_c_sol"] = tab.c_sol
        arrays[f"{name}_c_error"] = tab.c_error
        arrays[f"{name}_c_mid"] = mid
    save("tableaus.npz", **arrays)


class Replay:
    """A `func` that returns pre-drawn stage derivatives and records what the solver passes in."""

    def __init__(self, ks):
        self.ks, self.i, self.seen_y, self.seen_t, self.seen_p = ks, 0, [], [], []

    def __call__(self, t, y, perturb=Perturb.NONE):
        self.seen_t.append(float(t))
        self.seen_y.append(y.clone())
        self.seen_p.append(perturb.value)
        out = self.ks[self.i]
        self.i += 1
        return out


def gen_kernel_vectors():
    """_runge_kutta_step / _compute_error_ratio / _interp_fit / _interp_evaluate on random stages."""
    arrays = {}
    n = 1031
    for mname, solver_cls in [("dopri5", dopri5.Dopri5Solver), ("dopri8", dopri8.Dopri8Solver)]:
        for dname, dtype in [("f32", torch.float32), ("f64", torch.float64)]:
            key = f"{mname}_{dname}"
            y0 = rand(n, seed=1, dtype=dtype)
            S = len(solver_cls.tableau.alpha)
            ks = [rand(n, seed=10 + j, dtype=dtype) for j in range(S + 1)]
            solver = solver_cls(func=None, y0=y0, rtol=1e-3, atol=1e-4, norm=misc._rms_norm)
            t0 = torch.tensor(0.3, dtype=torch.float64)
            dt = torch.tensor(0.0371, dtype=torch.float64)
            replay = Replay(ks[1:])
            y1, f1, y1_error, k = rk_common._runge_kutta_step(replay, y0, ks[0], t0, dt, t0 + dt, solver.tableau)
            ratio = misc._compute_error_ratio(y1_error, solver.rtol, solver.atol, y0, y1, misc._rms_norm)
            coeffs = solver._interp_fit(y0, y1, k, dt)
            t_eval = t0 + 0.3 * dt
            y_eval = interp._interp_evaluate(coeffs, t0, t0 + dt, t_eval)
            arrays[f"{key}_y0"] = y0
            arrays[f"{key}_k"] = torch.stack(ks)
            arrays[f"{key}_t0_dt"] = torch.stack([t0, dt])
            arrays[f"{key}_stage_inputs"] = torch.stack(replay.seen_y)
            arrays[f"{key}_stage_times"] = torch.tensor(replay.seen_t, dtype=torch.float64)
            arrays[f"{key}_stage_perturb"] = torch.tensor(replay.seen_p)
            arrays[f"{key}_y1"] = y1
            arrays[f"{key}_y1_error"] = y1_error
            arrays[f"{key}_error_ratio"] = ratio.to(torch.float64)
            arrays[f"{key}_rtol_atol"] = torch.tensor([1e-3, 1e-4], dtype=torch.float64)
            arrays[f"{key}_interp_coeffs"] = torch.stack(coeffs)
            arrays[f"{key}_t_eval"] = t_eval
            arrays[f"{key}_y_eval"] = y_eval
    # rk4 3/8-rule increments
    for dname, dtype in [("f32", torch.float32), ("f64", torch.float64)]:
        y0 = rand(n, seed=1, dtype=dtype)
        ks = [rand(n, seed=10 + j, dtype=dtype) for j in range(4)]
        replay = Replay(ks[1:])
        t0 = torch.tensor(0.3, dtype=dtype)
        dt = torch.tensor(0.025, dtype=dtype)
        dy = rk_common.rk4_alt_step_func(replay, t0, dt, t0 + dt, y0, f0=ks[0])
        arrays[f"rk4_{dname}_y0"] = y0
        arrays[f"rk4_{dname}_k"] = torch.stack(ks)
        arrays[f"rk4_{dname}_t0_dt"] = torch.stack([t0, dt]).to(torch.float64)
        arrays[f"rk4_{dname}_stage_inputs"] = torch.stack(replay.seen_y)
        arrays[f"rk4_{dname}_stage_times"] = torch.tensor(replay.seen_t, dtype=torch.float64)
        arrays[f"rk4_{dname}_y1"] = y0 + dy
    save("kernels.npz", **arrays)


def gen_controller_vectors():
    """_optimal_step_size and _select_initial_step on scalar / small inputs."""
    cases, outs = [], []
    for last_step in [0.1, 1e-3, 2.5]:
        for ratio in [0.0, 1e-8, 0.3, 0.999, 1.0, 1.7, 50.0, 1e6, float("nan"), float("inf")]:
            for order in [5, 8]:
                args = [last_step, ratio, 0.9, 10.0, 0.2, order]
                r = misc._optimal_step_size(torch.tensor(last_step, dtype=torch.float64),
                                            torch.tensor(ratio, dtype=torch.float32),
                                            torch.tensor(0.9, dtype=torch.float64),
                                            torch.tensor(10.0, dtype=torch.float64),
                                            torch.tensor(0.2, dtype=torch.float64), order)
                cases.append(args)
                outs.append(float(r))
    arrays = {"optimal_step_in": np.array(cases), "optimal_step_out": np.array(outs)}
    # initial step for a linear field
    for dname, dtype in [("f32", torch.float32), ("f64", torch.float64)]:
        A = rand(16, 16, seed=3, dtype=dtype) / 4
        y0 = rand(40, 16, seed=4, dtype=dtype)
        func = misc._PerturbFunc(lambda t, y: y @ A.T)
        for order in [4, 7]:
            for scale_y in [1.0, 1e-7]:
                yy = y0 * scale_y
                h = misc._select_initial_step(func, torch.tensor(0.5, dtype=torch.float64), yy, order,
                                              torch.tensor(1e-6, dtype=torch.float64),
                                              torch.tensor(1e-8, dtype=torch.float64), misc._rms_norm)
                arrays[f"init_{dname}_o{order}_s{scale_y}"] = h
        arrays[f"init_{dname}_A"] = A
        arrays[f"init_{dname}_y0"] = y0
    save("controller.npz", **arrays)


class Counter:
    def __init__(self):
        self.accept, self.reject = [], []


def solve(func, y0, t, **kw):
    """Reference odeint with step statistics (via the reference's own callbacks)."""
    c = Counter()

    class F(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.nfe = 0

        def forward(self, t, y):
            self.nfe += 1
            return func(t, y)

        def callback_accept_step(self, t0, y0, dt):
            c.accept.append(float(dt))

        def callback_reject_step(self, t0, y0, dt):
            c.reject.append(float(dt))

    f = F()
    with torch.no_grad():
        y = torchdiffeq.odeint(f, y0, t, **kw)
    return y, f.nfe, c


def linear_problem(B, D, dtype, seed=0):
    g = torch.Generator().manual_seed(seed)
    G = torch.randn(D, D, generator=g, dtype=torch.float64) / D ** 0.5
    A = 0.5 * (G - G.T) - 0.1 * torch.eye(D, dtype=torch.float64)
    y0 = torch.randn(B, D, generator=g, dtype=torch.float64)
    return A.to(dtype), y0.to(dtype)


def gen_solves():
    arrays = {}
    # cfg1: spiral, rk4 on the output grid (examples/ode_demo.py:29-41 with method='rk4')
    A = torch.tensor([[-0.1, 2.0], [-2.0, -0.1]])
    y0 = torch.tensor([[2.0, 0.0]])
    t = torch.linspace(0.0, 25.0, 1000)
    with torch.no_grad():
        y = torchdiffeq.odeint(lambda t, y: (y ** 3) @ A, y0, t, method="rk4")
    arrays.update(cfg1_A=A, cfg1_y0=y0, cfg1_t=t, cfg1_y=y)

    # cfg2 (reduced batch): dopri5, dy/dt = A y, fp32; reference-default and looser tolerances
    A, y0 = linear_problem(64, 128, torch.float32)
    arrays.update(cfg2_A=A, cfg2_y0=y0)
    for tag, rtol, atol, tt in [("tight", 1e-7, 1e-9, [0.0, 1.0]), ("loose", 1e-5, 1e-7, [0.0, 0.25, 0.5, 1.0]),
                                ("rev", 1e-5, 1e-7, [1.0, 0.4, 0.0])]:
        t = torch.tensor(tt, dtype=torch.float64)
        y, nfe, c = solve(lambda t, y: y @ A.T, y0, t, rtol=rtol, atol=atol, method="dopri5")
        arrays[f"cfg2_{tag}_t"] = t
        arrays[f"cfg2_{tag}_tol"] = np.array([rtol, atol])
        arrays[f"cfg2_{tag}_y"] = y
        arrays[f"cfg2_{tag}_nfe"] = nfe
        arrays[f"cfg2_{tag}_accept_dt"] = np.array(c.accept)
        arrays[f"cfg2_{tag}_reject_dt"] = np.array(c.reject)

    # cfg4 (reduced): dopri8 fp64, rtol 1e-9
    A, y0 = linear_problem(32, 64, torch.float64, seed=1)
    t = torch.tensor([0.0, 0.3, 1.0], dtype=torch.float64)
    y, nfe, c = solve(lambda t, y: y @ A.T, y0, t, rtol=1e-9, atol=1e-11, method="dopri8")
    arrays.update(cfg4_A=A, cfg4_y0=y0, cfg4_t=t, cfg4_y=y, cfg4_nfe=nfe, cfg4_accept_dt=np.array(c.accept),
                  cfg4_reject_dt=np.array(c.reject))

    # dopri5 fp64 with rejections (stiff-ish scaling) and a time-dependent field
    A, y0 = linear_problem(16, 8, torch.float64, seed=2)
    t = torch.tensor([0.0, 2.0, 5.0], dtype=torch.float64)
    y, nfe, c = solve(lambda t, y: torch.sin(3 * t) * (y @ A.T) * 4 - y ** 3, y0, t, rtol=1e-8, atol=1e-10,
                      method="dopri5")
    arrays.update(tdep_A=A, tdep_y0=y0, tdep_t=t, tdep_y=y, tdep_nfe=nfe, tdep_accept_dt=np.array(c.accept),
                  tdep_reject_dt=np.array(c.reject))

    # tuple state (mixed norm)
    A, y0 = linear_problem(30, 8, torch.float32, seed=3)
    ya, yb = y0[:10].clone(), y0[10:].clone()
    t = torch.tensor([0.0, 1.0], dtype=torch.float64)
    with torch.no_grad():
        out = torchdiffeq.odeint(lambda t, y: (y[0] @ A.T, 2 * (y[1] @ A.T)), (ya, yb), t, rtol=1e-6, atol=1e-8)
    arrays.update(tuple_A=A, tuple_ya=ya, tuple_yb=yb, tuple_t=t, tuple_out_a=out[0], tuple_out_b=out[1])

    # options: first_step / step_t / jump_t / max_step (fp64, dopri5)
    A, y0 = linear_problem(8, 8, torch.float64, seed=4)
    t = torch.tensor([0.0, 1.0, 2.0], dtype=torch.float64)
    for tag, opts in [("first_step", dict(first_step=0.01)),
                      ("step_t", dict(step_t=torch.tensor([0.25, 1.5]))),
                      ("jump_t", dict(jump_t=torch.tensor([0.7]))),
                      ("max_step", dict(max_step=0.05)),
                      ("min_step", dict(min_step=0.2))]:
        y, nfe, c = solve(lambda t, y: y @ A.T, y0, t, rtol=1e-6, atol=1e-8, method="dopri5", options=opts)
        arrays[f"opt_{tag}_y"] = y
        arrays[f"opt_{tag}_nfe"] = nfe
        arrays[f"opt_{tag}_accept_dt"] = np.array(c.accept)
        arrays[f"opt_{tag}_reject_dt"] = np.array(c.reject)
    arrays.update(opt_A=A, opt_y0=y0, opt_t=t)

    # rk4 with step_size (linear interpolation of outputs) and perturb
    A, y0 = linear_problem(4, 8, torch.float32, seed=5)
    t = torch.tensor([0.0, 0.33, 1.0])
    with torch.no_grad():
        y = torchdiffeq.odeint(lambda t, y: y @ A.T, y0, t, method="rk4", options=dict(step_size=0.1))
        yp = torchdiffeq.odeint(lambda t, y: torch.cos(t) * (y @ A.T), y0, t, method="rk4",
                                options=dict(step_size=0.1, perturb=True))
    arrays.update(rk4s_A=A, rk4s_y0=y0, rk4s_t=t, rk4s_y=y, rk4s_y_perturb=yp)
    save("solves.npz", **arrays)


def gen_dopri8_small():
    """r06: dopri8 on SMALL states, fp32 at rtol 1e-5 and fp64 at rtol 1e-9, with the accepted
    step sequence stored.  The first step of such a solve is tiny, its 13-stage error row cancels to rounding noise and the
    second step size depends on the association of the row sums (the reference: ATen's `torch.sum` order, rk_common.py:79,
    89; this package: non-zeros left to right) — the one place where "identical results" is a bound, not
    bits.  tests/test_dopri8_small_golden.py asserts the measured bounds so that a regression is visible."""
    from _cases import DOPRI8_SMALL_CASES, dopri8_small_field
    arrays = {}
    for name, (kind, n, d, seed, t1) in DOPRI8_SMALL_CASES.items():
        for dname, dtype, rtol, atol in (("f32", torch.float32, 1e-5, 1e-7), ("f64", torch.float64, 1e-9, 1e-11)):
            W = (rand(d, d, seed=seed) * 0.6).to(dtype)
            y0 = rand(n, d, seed=seed + 1).to(dtype)
            t = torch.tensor([0.0, t1], dtype=dtype)
            y, nfe, c = solve(dopri8_small_field(kind, W), y0, t, rtol=rtol, atol=atol, method="dopri8")
            key = f"d8_{name}_{dname}"
            arrays[f"{key}_W"], arrays[f"{key}_y0"], arrays[f"{key}_t"] = W, y0, t
            arrays[f"{key}_y"], arrays[f"{key}_nfe"] = y, nfe
            arrays[f"{key}_accept_dt"], arrays[f"{key}_reject_dt"] = np.array(c.accept), np.array(c.reject)
    save("dopri8_small.npz", **arrays)


def gen_int_state():
    """What the reference does with an INTEGER state (advisor r05): its fixed-grid methods run — `y0 + dt * f` promotes
    every step to float, the solution buffer keeps y0's dtype, so each output row is TRUNCATED back to int64 —, its adaptive
    methods fail inside `nextafter` (NotImplementedError).  This package refuses integer / bool states for every method
    (`UnsupportedStateDtype`, a TypeError and a NotImplementedError); tests/test_api_corners_r4.py pins the refusal next to
    these recorded results, so the deviation is a decision on record, not an accident."""
    arrays = {}
    y0, t = torch.tensor([4, 8]), torch.tensor([0.0, 1.0, 2.0])
    for m in ("euler", "midpoint", "rk4"):
        y = torchdiffeq.odeint(lambda t_, y_: -0.5 * y_, y0, t, method=m)
        assert y.dtype == torch.int64
        arrays[f"int_{m}_y"] = y
    try:
        torchdiffeq.odeint(lambda t_, y_: -0.5 * y_, y0, t, method="dopri5")
        arrays["int_dopri5_error"] = np.array("none")
    except Exception as exc:      # noqa: BLE001
        arrays["int_dopri5_error"] = np.array(type(exc).__name__)
    save("int_state.npz", **arrays)


def gen_adjoint():
    """cfg3 (reduced): odeint_adjoint through a tanh MLP, loss = sum(y(T)^2) (+ intermediate outputs)."""
    arrays = {}
    for tag, dtype, d, h, B, rtol, atol, tt in [("f32", torch.float32, 16, 32, 96, 1e-5, 1e-7, [0.0, 1.0]),
                                                 ("f64", torch.float64, 8, 16, 40, 1e-8, 1e-10, [0.0, 0.5, 1.0])]:
        torch.manual_seed(0)
        net = torch.nn.Sequential(torch.nn.Linear(d, h), torch.nn.Tanh(), torch.nn.Linear(h, h), torch.nn.Tanh(),
                                  torch.nn.Linear(h, d)).to(dtype)

        class F(torch.nn.Module):
            def __init__(self):
                super().__init__()
                self.net = net

            def forward(self, t, y):
                return self.net(y)

        f = F()
        y0 = rand(B, d, seed=7, dtype=dtype).requires_grad_(True)
        t = torch.tensor(tt, dtype=torch.float64)
        for norm_tag, aopts in [("default", None), ("seminorm", dict(norm="seminorm"))]:
            for p in f.parameters():
                p.grad = None
            y0.grad = None
            y = torchdiffeq.odeint_adjoint(f, y0, t, rtol=rtol, atol=atol, method="dopri5", adjoint_options=aopts)
            loss = y[-1].pow(2).sum() + (y[1:].sum() if len(tt) > 2 else 0.0)
            loss.backward()
            arrays[f"adj_{tag}_{norm_tag}_y"] = y
            arrays[f"adj_{tag}_{norm_tag}_grad_y0"] = y0.grad
            for i, p in enumerate(f.parameters()):
                arrays[f"adj_{tag}_{norm_tag}_grad_p{i}"] = p.grad
        arrays[f"adj_{tag}_y0"] = y0
        arrays[f"adj_{tag}_t"] = t
        arrays[f"adj_{tag}_tol"] = np.array([rtol, atol])
        for i, p in enumerate(f.parameters()):
            arrays[f"adj_{tag}_p{i}"] = p
    save("adjoint.npz", **arrays)


def gen_cnf():
    """cfg5 (reduced): the CNF of examples/cnf.py:34-114 at random init (seed 0), tuple state (z, logp),
    t: 10 -> 0 (decreasing), dopri5 + adjoint, rtol = atol = 1e-5; loss = mean(logp(t1)) - sum(z(t1)^2)/100."""
    import importlib.util
    argv, sys.argv = sys.argv, ["cnf.py"]
    try:
        spec = importlib.util.spec_from_file_location("ref_cnf_example", "/root/reference/examples/cnf.py")
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
    finally:
        sys.argv = argv
    arrays = {}
    torch.manual_seed(0)
    func = mod.CNF(in_out_dim=2, hidden_dim=32, width=64)
    B = 96
    z0 = rand(B, 2, seed=11, dtype=torch.float32).requires_grad_(True)
    logp0 = torch.zeros(B, 1)
    t = torch.tensor([10.0, 0.0])
    z_t, logp_t = torchdiffeq.odeint_adjoint(func, (z0, logp0), t, atol=1e-5, rtol=1e-5, method="dopri5")
    loss = logp_t[-1].mean() - z_t[-1].pow(2).sum() / 100
    loss.backward()
    arrays.update(cnf_z0=z0, cnf_t=t, cnf_z=z_t, cnf_logp=logp_t, cnf_grad_z0=z0.grad)
    for i, (name, p) in enumerate(func.named_parameters()):
        arrays[f"cnf_p{i}"] = p
        arrays[f"cnf_grad_p{i}"] = p.grad
    arrays["cnf_param_names"] = np.array([n for n, _ in func.named_parameters()])
    save("cnf.npz", **arrays)


def gen_methods():
    """§8(f) rank 2: every other explicit RK method of the reference's SOLVERS table.
    Adaptive pairs on a time-dependent nonlinear field (fp64: step sequences must match; fp32: solution),
    fixed-grid methods with step_size / perturb / cubic interpolation (no reductions -> bit-exact)."""
    arrays = {}
    A, y0 = linear_problem(12, 8, torch.float64, seed=6)
    field = lambda t, y: torch.sin(2 * t) * (y @ A.T) * 2 - 0.5 * y ** 3
    t = torch.tensor([0.0, 0.6, 2.0], dtype=torch.float64)
    arrays.update(ad_A=A, ad_y0=y0, ad_t=t)
    for method, rtol, atol in [("tsit5", 1e-8, 1e-10), ("bosh3", 1e-6, 1e-8), ("fehlberg2", 1e-4, 1e-6),
                               ("adaptive_heun", 1e-4, 1e-6)]:
        y, nfe, c = solve(field, y0, t, rtol=rtol, atol=atol, method=method)
        arrays[f"ad_{method}_tol"] = np.array([rtol, atol])
        arrays[f"ad_{method}_y"] = y
        arrays[f"ad_{method}_nfe"] = nfe
        arrays[f"ad_{method}_accept_dt"] = np.array(c.accept)
        arrays[f"ad_{method}_reject_dt"] = np.array(c.reject)
        # fp32 state, decreasing time
        A32, y32 = A.float(), y0.float()
        f32 = lambda t, y: torch.sin(2 * t) * (y @ A32.T) * 2 - 0.1 * y
        y, nfe, c = solve(f32, y32, torch.tensor([1.0, 0.3, 0.0], dtype=torch.float64), rtol=1e-4, atol=1e-6,
                          method=method)
        arrays[f"ad32_{method}_y"] = y
        arrays[f"ad32_{method}_nfe"] = nfe
    # fixed grid
    A, y0 = linear_problem(4, 8, torch.float32, seed=5)
    t = torch.tensor([0.0, 0.33, 0.7, 1.0])
    arrays.update(fx_A=A, fx_y0=y0, fx_t=t)
    with torch.no_grad():
        for method in ["euler", "midpoint", "heun2", "heun3", "rk4"]:
            f = lambda t, y: torch.cos(t) * (y @ A.T) - 0.1 * y
            arrays[f"fx_{method}_grid"] = torchdiffeq.odeint(f, y0, torch.linspace(0, 1, 9), method=method)
            arrays[f"fx_{method}_step"] = torchdiffeq.odeint(f, y0, t, method=method, options=dict(step_size=0.1))
            arrays[f"fx_{method}_perturb"] = torchdiffeq.odeint(f, y0, t, method=method,
                                                                options=dict(step_size=0.1, perturb=True))
            arrays[f"fx_{method}_cubic"] = torchdiffeq.odeint(f, y0, t, method=method,
                                                              options=dict(step_size=0.1, interp="cubic"))
            arrays[f"fx_{method}_rev"] = torchdiffeq.odeint(f, y0, torch.tensor([1.0, 0.45, 0.0]), method=method,
                                                            options=dict(step_size=0.125, interp="cubic"))
            y64 = y0.double()
            A64 = A.double()
            f64 = lambda t, y: torch.cos(t) * (y @ A64.T) - 0.1 * y
            arrays[f"fx_{method}_f64"] = torchdiffeq.odeint(f64, y64, t.double(), method=method,
                                                            options=dict(step_size=0.05))
    save("methods.npz", **arrays)


def gen_events():
    """§8(f) ranks 3-4: odeint(event_fn=...), odeint_event (+ adjoint gradients through the event time) and
    odeint_dense, on a damped rotation dy/dt = A y whose first coordinate crosses fixed levels."""
    arrays = {}
    A = torch.tensor([[-0.1, 2.0], [-2.0, -0.1]], dtype=torch.float64)
    y0 = torch.tensor([[2.0, 0.0], [1.0, 1.0], [0.5, -1.5]], dtype=torch.float64)
    arrays.update(ev_A=A, ev_y0=y0)
    f = lambda t, y: y @ A.T
    ev_scalar = lambda t, y: y[0, 0] - 0.5
    ev_multi = lambda t, y: torch.stack([y[0, 0] + 1.0, y[1, 1] + 0.25, t - 5.0])
    with torch.no_grad():
        for method, opts in [("dopri5", {}), ("dopri8", {}), ("tsit5", {}), ("bosh3", {}), ("adaptive_heun", {}),
                             ("rk4", dict(step_size=0.01)), ("rk4", dict(step_size=0.01, interp="cubic")),
                             ("euler", dict(step_size=0.001)), ("midpoint", dict(step_size=0.01, interp="cubic")),
                             ("heun3", dict(step_size=0.02, interp="cubic"))]:
            tag = method + ("_cubic" if opts.get("interp") == "cubic" else "")
            for ename, efn in [("scalar", ev_scalar), ("multi", ev_multi)]:
                for rev in (False, True):
                    t = torch.tensor([0.0, -1.0] if rev else [0.0, 1.0], dtype=torch.float64)
                    te, ye = torchdiffeq.odeint(f, y0, t, event_fn=efn, method=method, options=opts, rtol=1e-8,
                                                atol=1e-9)
                    arrays[f"ev_{tag}_{ename}_{'rev' if rev else 'fwd'}_t"] = te
                    arrays[f"ev_{tag}_{ename}_{'rev' if rev else 'fwd'}_y"] = ye
        # fp32 state, fixed grid: times are kept in the state dtype (solvers.py:132)
        te, ye = torchdiffeq.odeint(lambda t, y: y @ A.float().T, y0.float(), torch.tensor([0.0, 1.0]),
                                    event_fn=ev_scalar, method="rk4", options=dict(step_size=0.01), atol=1e-6)
        arrays.update(ev32_rk4_t=te, ev32_rk4_y=ye)
        te, ye = torchdiffeq.odeint(lambda t, y: y @ A.float().T, y0.float(), torch.tensor([0.0, 1.0]),
                                    event_fn=ev_scalar, method="dopri5", rtol=1e-5, atol=1e-6)
        arrays.update(ev32_dopri5_t=te, ev32_dopri5_y=ye)
        # tuple state
        te, (ya, yb) = torchdiffeq.odeint(lambda t, y: (y[0] @ A.T, -y[1]), (y0, torch.ones(2, dtype=torch.float64)),
                                          torch.tensor([0.0, 1.0], dtype=torch.float64),
                                          event_fn=lambda t, y: y[0][0, 0] - y[1][0], method="dopri5", rtol=1e-8, atol=1e-9)
        arrays.update(ev_tuple_t=te, ev_tuple_ya=ya, ev_tuple_yb=yb)

    # odeint_event + adjoint: gradient of the event time and of the state at the event
    class F(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.A = torch.nn.Parameter(A.clone())

        def forward(self, t, y):
            return y @ self.A.T

    for rev in (False, True):
        func = F()
        y0g = y0.clone().requires_grad_(True)
        t0 = torch.tensor(0.0, dtype=torch.float64, requires_grad=True)
        te, sol = torchdiffeq.odeint_event(func, y0g, t0, event_fn=ev_scalar, reverse_time=rev,
                                           odeint_interface=torchdiffeq.odeint_adjoint, method="dopri5", rtol=1e-9,
                                           atol=1e-10)
        loss = te * 3.0 + sol[-1].pow(2).sum()
        loss.backward()
        tag = "rev" if rev else "fwd"
        arrays[f"oe_{tag}_t"] = te
        arrays[f"oe_{tag}_sol"] = sol
        arrays[f"oe_{tag}_grad_y0"] = y0g.grad
        arrays[f"oe_{tag}_grad_A"] = func.A.grad
        arrays[f"oe_{tag}_grad_t0"] = t0.grad

    # odeint_dense
    with torch.no_grad():
        fn = torchdiffeq.odeint_dense(f, y0, torch.tensor(0.0, dtype=torch.float64), torch.tensor(3.0, dtype=torch.float64),
                                      rtol=1e-6, atol=1e-8, method="dopri5")
        t_eval = torch.tensor([0.0, 0.01, 0.4, 1.234, 2.0, 2.999], dtype=torch.float64)
        arrays["dense_t_eval"] = t_eval
        arrays["dense_y_eval"] = torch.stack([fn(te_) for te_ in t_eval])
        y32 = y0.float()
        fn = torchdiffeq.odeint_dense(lambda t, y: y @ A.float().T, y32, torch.tensor(0.0), torch.tensor(3.0),
                                      rtol=1e-4, atol=1e-6, method="dopri5")
        arrays["dense32_y_eval"] = torch.stack([fn(te_) for te_ in t_eval])
    save("events.npz", **arrays)


def gen_backprop():
    """§8(f) rank 1: gradients of plain `odeint` (backprop THROUGH the solver) wrt y0, parameters and t,
    for a time-dependent tanh field; fp64 so the comparison is limited by the algorithm, not by rounding."""
    arrays = {}
    torch.manual_seed(5)
    d, h, B = 4, 8, 6
    W1, b1, W2 = torch.randn(h, d, dtype=torch.float64) * 0.6, torch.randn(h, dtype=torch.float64) * 0.1, \
        torch.randn(d, h, dtype=torch.float64) * 0.6
    y0 = rand(B, d, seed=21)
    arrays.update(bp_W1=W1, bp_b1=b1, bp_W2=W2, bp_y0=y0)

    def run(method, tt, options=None, tuple_state=False, rtol=1e-8, atol=1e-10):
        p = [x.clone().requires_grad_(True) for x in (W1, b1, W2)]
        yy = y0.clone().requires_grad_(True)
        t = torch.tensor(tt, dtype=torch.float64, requires_grad=True)
        field = lambda t_, y_: torch.tanh(y_ @ p[0].T + p[1]) @ p[2].T * torch.cos(t_) - 0.1 * y_
        if tuple_state:
            f = lambda t_, y_: (field(t_, y_[0]), -y_[1] * y_[0].sum(-1, keepdim=True))
            ya, yb = torchdiffeq.odeint(f, (yy, torch.ones(B, 1, dtype=torch.float64)), t, method=method,
                                        options=options, rtol=rtol, atol=atol)
            loss = ya[-1].pow(2).sum() + ya[1].sum() + yb[-1].sum()
            sol = ya
        else:
            sol = torchdiffeq.odeint(field, yy, t, method=method, options=options, rtol=rtol, atol=atol)
            loss = sol[-1].pow(2).sum() + sol[1].sum()
        loss.backward()
        return dict(y=sol, g_y0=yy.grad, g_W1=p[0].grad, g_b1=p[1].grad, g_W2=p[2].grad, g_t=t.grad)

    cases = [("dopri5", "dopri5", [0.0, 0.4, 1.0], None, False), ("dopri8", "dopri8", [0.0, 0.4, 1.0], None, False),
             ("tsit5", "tsit5", [0.0, 0.4, 1.0], None, False), ("bosh3", "bosh3", [0.0, 0.4, 1.0], None, False),
             ("fehlberg2", "fehlberg2", [0.0, 0.4, 1.0], None, False),
             ("adaptive_heun", "adaptive_heun", [0.0, 0.4, 1.0], None, False),
             ("dopri5_rev", "dopri5", [1.0, 0.3, 0.0], None, False),
             ("dopri5_tuple", "dopri5", [0.0, 0.4, 1.0], None, True),
             ("rk4_grid", "rk4", [0.0, 0.2, 0.45, 0.7, 1.0], None, False),
             ("euler_grid", "euler", [0.0, 0.2, 0.45, 0.7, 1.0], None, False),
             ("midpoint_step", "midpoint", [0.0, 0.33, 1.0], dict(step_size=0.1), False),
             ("heun2_perturb", "heun2", [0.0, 0.33, 1.0], dict(step_size=0.1, perturb=True), False),
             ("heun3_cubic", "heun3", [0.0, 0.33, 1.0], dict(step_size=0.1, interp="cubic"), False),
             ("rk4_cubic_rev", "rk4", [1.0, 0.45, 0.0], dict(step_size=0.125, interp="cubic"), False),
             # r04: steps cut short at prescribed points — dt = t_point - t0 moves against t0, t1 = t_point with nothing
             # (rk_common.py:296-309): the first-step heuristic's / t[0]'s gradient ends there
             ("dopri5_step_t_first", "dopri5", [0.0, 0.4, 1.0], dict(step_t=[0.013]), False),
             ("dopri5_step_t_mid", "dopri5", [0.0, 0.4, 1.0], dict(step_t=[0.55]), False),
             ("bosh3_jump_t_mid", "bosh3", [0.0, 0.4, 1.0], dict(jump_t=[0.55]), False),
             ("tsit5_tuple_step_jump", "tsit5", [0.0, 0.4, 1.0], dict(step_t=[0.2, 0.8], jump_t=[0.5]), True),
             ("dopri5_rev_step_t", "dopri5", [1.0, 0.3, 0.0], dict(step_t=[0.6]), False),
             # the heuristic first step clamped to min_step: `dt.clamp(...)` is a constant there (rk_common.py:271)
             ("bosh3_min_step", "bosh3", [0.0, 0.4, 1.0], dict(min_step=0.5), False)]
    for tag, method, tt, opts, tup in cases:
        tol = dict(rtol=1e-4, atol=1e-6) if method in ("fehlberg2", "adaptive_heun") else {}
        out = run(method, tt, opts, tup, **tol)
        for k, v in out.items():
            arrays[f"bp_{tag}_{k}"] = v
        arrays[f"bp_{tag}_t"] = np.array(tt)
    save("backprop.npz", **arrays)


sys.path.insert(0, os.path.join(HERE, ".."))
from _cases import TUPLE_TOL_LIST_FORMS  # noqa: E402  (inputs only: the tolerance forms of the r06 list-entry cases)


def gen_tuple_tolerances():
    """Tuple states with PER-COMPONENT tolerances (misc.py:115-123 `_tuple_tol`: rtol / atol become per-element
    vectors, fp32 values widened to fp64, so the reference forms the error ratio in fp64 there)."""
    arrays = {}
    for dname, dtype in (("f32", torch.float32), ("f64", torch.float64)):
        A = (rand(5, 5, seed=21) * 0.4 - 0.3 * torch.eye(5, dtype=torch.float64)).to(dtype)
        ya, yb = rand(30, 5, seed=22).to(dtype), rand(7, seed=23).to(dtype)
        t = torch.tensor([0.0, 0.7, 2.0], dtype=torch.float64)
        count = [0]

        def f(t_, y_):
            count[0] += 1
            return y_[0] @ A.T * torch.cos(t_), -y_[1] * 0.5

        for tag, tt in (("fwd", t), ("rev", t.flip(0))):
            count[0] = 0
            sa, sb = torchdiffeq.odeint(f, (ya, yb), tt, rtol=(1e-5, 1e-3), atol=(1e-7, 1e-4), method="dopri5")
            arrays[f"tt_{dname}_{tag}_ya"], arrays[f"tt_{dname}_{tag}_yb"] = sa, sb
            arrays[f"tt_{dname}_{tag}_nfe"] = count[0]
        arrays[f"tt_{dname}_A"], arrays[f"tt_{dname}_y0a"], arrays[f"tt_{dname}_y0b"] = A, ya, yb
    # r06: an ENTRY of a tuple tolerance may be any array-like `torch.as_tensor` takes (misc.py:121) —
    # a Python list, a tuple of numbers, a numpy array — not only a tensor.  Forms: list entry in rtol / in atol / in
    # both / numpy + tuple entries; fp32 and fp64 states; one odeint_adjoint forward + backward.
    for dname, dtype in (("f32", torch.float32), ("f64", torch.float64)):
        x0, b0 = rand(3, seed=31).to(dtype), rand(2, seed=32).to(dtype)
        w = torch.tensor([1.0, 3.0, 0.3], dtype=dtype)
        t = torch.tensor([0.0, 0.4, 1.1], dtype=dtype)
        arrays[f"ttl_{dname}_x0"], arrays[f"ttl_{dname}_b0"] = x0, b0
        for form, (rtol, atol) in TUPLE_TOL_LIST_FORMS.items():
            count, accepted = [0], []

            class F(torch.nn.Module):
                def forward(self, t_, y_):
                    count[0] += 1
                    return -y_[0] * w * (1 + 0.2 * t_) + 0.1 * torch.sin(y_[0]), -0.4 * y_[1]

                def callback_accept_step(self, t0, y_, dt):
                    accepted.append(float(dt))
            with torch.no_grad():
                sa, sb = torchdiffeq.odeint(F(), (x0, b0), t, rtol=rtol, atol=atol, method="dopri5")
            arrays[f"ttl_{dname}_{form}_ya"], arrays[f"ttl_{dname}_{form}_yb"] = sa, sb
            arrays[f"ttl_{dname}_{form}_nfe"] = count[0]
            arrays[f"ttl_{dname}_{form}_accept_dt"] = np.array(accepted)
    x = rand(3, seed=31).requires_grad_(True)
    wp = torch.tensor([1.0, 3.0, 0.3], dtype=torch.float64, requires_grad=True)
    rtol, atol = TUPLE_TOL_LIST_FORMS["both"]
    out = torchdiffeq.odeint_adjoint(lambda t_, y_: (-y_[0] * wp * (1 + 0.2 * t_) + 0.1 * torch.sin(y_[0]), -0.4 * y_[1]),
                                     (x, rand(2, seed=32)), torch.tensor([0.0, 0.4, 1.1], dtype=torch.float64),
                                     rtol=rtol, atol=atol, adjoint_rtol=1e-8, adjoint_atol=1e-10, adjoint_params=(wp,))
    out[0][-1].pow(2).sum().backward()
    arrays["ttl_adj_ya"], arrays["ttl_adj_gx"], arrays["ttl_adj_gw"] = out[0], x.grad, wp.grad
    save("tuple_tol.npz", **arrays)


def gen_adjoint_time_dependent():
    """Adjoint of a TIME-DEPENDENT field, with default and user-supplied norms.  Two reference behaviours are pinned
    here: (i) vjp_t is formed in every backward evaluation even when `t` does not require grad (the in-place
    `requires_grad_` of adjoint.py:84-88), so it takes part in the backward solve's norms and step sizes;
    (ii) a user forward `norm` is also used by `_select_initial_step` (rk_common.py:217) and, through
    `handle_adjoint_norm_`, on y / adj_y of the backward solve (adjoint.py:247-270)."""
    arrays = {}
    torch.manual_seed(5)
    W = torch.randn(4, 4, dtype=torch.float64) * 0.6
    b = torch.randn(4, dtype=torch.float64) * 0.1
    ya = rand(6, 4, seed=31)
    yb = rand(3, seed=32)
    t = torch.tensor([0.0, 0.6, 1.4], dtype=torch.float64)
    arrays.update(adjt_W=W, adjt_b=b, adjt_ya=ya, adjt_yb=yb, adjt_t=t)
    for tup in (False, True):
        for norm_tag in ("default", "usernorm", "usernorm_semi", "semi"):
            lin = torch.nn.Linear(4, 4).double()
            with torch.no_grad():
                lin.weight.copy_(W)
                lin.bias.copy_(b)
            acc, rej, nfe = [], [], [0]

            class F(torch.nn.Module):
                def __init__(self):
                    super().__init__()
                    self.lin = lin

                def forward(self, t_, y_):
                    nfe[0] += 1
                    if tup:
                        return torch.tanh(self.lin(y_[0])) * torch.cos(t_), -y_[1] * 0.3 * t_
                    return torch.tanh(self.lin(y_)) * torch.cos(t_)

                def callback_accept_step_adjoint(self, t0, y0, dt):
                    acc.append(float(dt))

                def callback_reject_step_adjoint(self, t0, y0, dt):
                    rej.append(float(dt))

            f = F()
            a0 = ya.clone().requires_grad_(True)
            b0 = yb.clone().requires_grad_(True)
            y0 = (a0, b0) if tup else a0
            opts, aopts = None, None
            if norm_tag.startswith("usernorm"):
                opts = dict(norm=(lambda y: max(y[0].abs().max(), y[1].abs().max())) if tup else (lambda y: y.abs().max()))
            if norm_tag.endswith("semi"):
                aopts = dict(norm="seminorm")
            out = torchdiffeq.odeint_adjoint(f, y0, t, rtol=1e-6, atol=1e-8, method="dopri5", options=opts,
                                             adjoint_options=aopts)
            o = out[0] if tup else out
            nfe_fwd = nfe[0]
            (o[-1].pow(2).sum() + o[1].sum() + (out[1][-1].sum() if tup else 0.0)).backward()
            key = f"adjt_{'tup' if tup else 'ten'}_{norm_tag}"
            arrays[f"{key}_y"] = o
            arrays[f"{key}_g_y0"] = a0.grad
            arrays[f"{key}_g_W"] = lin.weight.grad
            arrays[f"{key}_g_b"] = lin.bias.grad
            arrays[f"{key}_nfe"] = np.array([nfe_fwd, nfe[0] - nfe_fwd])
            arrays[f"{key}_acc"] = np.array(acc)
            arrays[f"{key}_rej"] = np.array(rej)
    save("adjoint_tdep.npz", **arrays)


def adams_field(t, y):
    """Only exactly rounded elementwise operations (+, -, * as separate torch ops; roll): the derivative values are
    the same bits on every CPU and on the GPU — no transcendental function or GEMM whose last bit depends on the
    machine — so the solves can be compared for equality wherever the tests run."""
    return (1 - t * 0.5) * (y.roll(1, -1) * 0.3 - y * 0.2) - y * y * y * 0.01


def gen_adams():
    """Widening beyond §8(f): the Adams multistep methods of the reference's SOLVERS table (fixed_adams.py).
    Coefficient tables, solves (grid / step_size / perturb / cubic / reverse / low max_order / one iteration with
    its non-convergence warnings / fp64), NFE counts, a tuple state with per-component tolerances, gradients by
    backprop through the solver and an event solve."""
    import warnings
    from torchdiffeq._impl import fixed_adams
    arrays = {}
    for k in range(1, 13):
        arrays[f"bashforth_{k}"] = fixed_adams._BASHFORTH_DIVISOR[k]
        arrays[f"moulton_{k}"] = fixed_adams._MOULTON_DIVISOR[k]
    A, y0 = linear_problem(4, 8, torch.float32, seed=15)
    t = torch.tensor([0.0, 0.33, 0.7, 1.0])
    arrays.update(A=A, y0=y0, t=t)

    class Count:
        def __init__(self, fn):
            self.fn, self.nfe = fn, 0

        def __call__(self, t, y):
            self.nfe += 1
            return self.fn(t, y)

    with torch.no_grad():
        for method in ["explicit_adams", "implicit_adams"]:
            f = f64 = adams_field
            cases = {
                "grid": (f, y0, torch.linspace(0, 1, 41), {}),
                "step": (f, y0, t, dict(step_size=0.02)),
                "perturb": (f, y0, t, dict(step_size=0.02, perturb=True)),
                "cubic": (f, y0, t, dict(step_size=0.02, interp="cubic")),
                "rev": (f, y0, torch.tensor([1.0, 0.45, 0.0]), dict(step_size=0.025, interp="cubic")),
                "order6": (f, y0, t, dict(step_size=0.02, max_order=6)),
                "iters1": (f, y0, t, dict(step_size=0.02, max_iters=1)),
                "f64": (f64, y0.double(), t.double(), dict(step_size=0.0125)),
            }
            for tag, (fn, y, tt, opts) in cases.items():
                c = Count(fn)
                with warnings.catch_warnings(record=True) as w:
                    warnings.simplefilter("always")
                    arrays[f"{method}_{tag}"] = torchdiffeq.odeint(c, y, tt, method=method, options=opts,
                                                                   rtol=1e-6, atol=1e-8)
                arrays[f"{method}_{tag}_nfe"] = c.nfe
                arrays[f"{method}_{tag}_warnings"] = len(w)
            # tuple state, per-component tolerances, fp64 time grid over an fp32 state
            ft = lambda t, y: (adams_field(t, y[0]), -y[1] * y[0][0, :3] * (1 + t))
            yt = (y0, torch.tensor([0.5, 0.25, 1.0]))
            out = torchdiffeq.odeint(ft, yt, torch.linspace(0, 1, 31, dtype=torch.float64), method=method,
                                     rtol=(1e-6, 1e-5), atol=(1e-8, 1e-7))
            arrays[f"{method}_tuple0"], arrays[f"{method}_tuple1"] = out
            # 0-dim fp32 state: the fp64 coefficient tensors promote the history dot products to fp64
            fs = lambda t, y: (1 - t * 0.5) * (y * -0.7) - y * y * y * 0.01
            arrays[f"{method}_zerodim"] = torchdiffeq.odeint(fs, torch.tensor(1.5), torch.linspace(0, 1, 41),
                                                             method=method, rtol=1e-6, atol=1e-8)
    # 0-dim fp32 state on an fp64 time grid: `dt * f` promotes to fp64 and the rest of the solve runs in fp64
    fs = lambda t, y: (1 - t * 0.5) * (y * -0.7) - y * y * y * 0.01
    with torch.no_grad():
        for method in ["euler", "midpoint", "heun3", "rk4", "explicit_adams", "implicit_adams"]:
            arrays[f"zerodim64_{method}"] = torchdiffeq.odeint(fs, torch.tensor(1.5),
                                                               torch.linspace(0, 1, 21, dtype=torch.float64),
                                                               method=method, rtol=1e-6, atol=1e-8)
    # kernel-level vectors: the reference's own expressions (fixed_adams.py:205, :210, :213-215, :189-192) on random data
    for dtype, tag in [(torch.float32, "f32"), (torch.float64, "f64")]:
        n, order = 777, 7
        hist = [rand(n, seed=30 + j, dtype=dtype) for j in range(order)]
        y0k = rand(n, seed=29, dtype=dtype)
        dt = torch.tensor(0.0371, dtype=dtype)
        solver = fixed_adams.AdamsBashforthMoulton(lambda t, y: y, y0k, rtol=1e-3, atol=1e-4)
        bash, moulton = solver.bashforth[order], solver.moulton[order + 1]
        dy = fixed_adams._dot_product(dt * bash, hist).type_as(y0k)
        delta = dt * fixed_adams._dot_product(moulton[1:], hist).type_as(y0k)
        fnew = rand(n, seed=28, dtype=dtype)
        dy_new = (dt * moulton[0] * fnew).type_as(y0k) + delta
        # a second pair that is close to dy_new, so that the census is mixed
        dy_close = dy_new * (1 + 2e-3 * rand(n, seed=27, dtype=dtype))
        ratio = misc._compute_error_ratio(torch.abs(dy_close - dy_new), solver.rtol, solver.atol, dy_close, dy_new,
                                          misc._linf_norm)
        viol = ((torch.abs(dy_close - dy_new) / (solver.atol + solver.rtol * torch.max(dy_close.abs(), dy_new.abs())))
                >= 1).sum()
        arrays.update({f"kv_{tag}_hist": torch.stack(hist), f"kv_{tag}_y0": y0k, f"kv_{tag}_dt": dt,
                       f"kv_{tag}_dy": dy, f"kv_{tag}_delta": delta, f"kv_{tag}_ypred": y0k + dy,
                       f"kv_{tag}_f": fnew, f"kv_{tag}_dy_new": dy_new, f"kv_{tag}_y_new": y0k + dy_new,
                       f"kv_{tag}_dy_close": dy_close, f"kv_{tag}_ratio": ratio, f"kv_{tag}_violations": viol,
                       f"kv_{tag}_converged_far": solver._has_converged(dy, dy_new),
                       f"kv_{tag}_converged_close": solver._has_converged(dy_close, dy_new),
                       f"kv_{tag}_converged_same": solver._has_converged(dy_new, dy_new)})
    # gradients by backprop through the solver (y0, t and the parameters of the field)
    for method in ["explicit_adams", "implicit_adams"]:
        torch.manual_seed(3)
        lin = torch.nn.Linear(3, 3).double()
        yg = rand(5, 3, seed=16).requires_grad_(True)
        tg = torch.linspace(0, 1, 21, dtype=torch.float64).requires_grad_(True)
        fg = lambda t, y: torch.tanh(lin(y)) * torch.cos(t)
        y = torchdiffeq.odeint(fg, yg, tg, method=method, rtol=1e-6, atol=1e-8)
        loss = y[-1].pow(2).sum() + y[7].sum()
        g = torch.autograd.grad(loss, [yg, tg, lin.weight, lin.bias])
        arrays[f"{method}_bp_w"], arrays[f"{method}_bp_b"] = lin.weight, lin.bias
        arrays[f"{method}_bp_y0"] = yg
        arrays[f"{method}_bp_y"] = y
        for name, v in zip(["gy0", "gt", "gw", "gb"], g):
            arrays[f"{method}_bp_{name}"] = v
        # event: harmonic oscillator crossing y[0] = 0
        fe = lambda t, y: torch.stack([y[1], -y[0]])
        et, ys = torchdiffeq.odeint_event(fe, torch.tensor([1.0, 0.0], dtype=torch.float64),
                                          torch.tensor(0.0, dtype=torch.float64), event_fn=lambda t, y: y[0],
                                          method=method, options=dict(step_size=0.01), atol=1e-8)
        arrays[f"{method}_event_t"], arrays[f"{method}_event_y"] = et, ys
    save("adams.npz", **arrays)


IMPLICIT_METHODS = ["implicit_euler", "implicit_midpoint", "trapezoid", "radauIIA3", "gl4", "radauIIA5", "gl6",
                    "sdirk2", "trbdf2"]


def implicit_field(t, y):
    """Exactly rounded elementwise operations only (see adams_field); moderately stiff linear part."""
    return (1 - t * 0.5) * (y.roll(1, -1) * 0.3 - y * 2.0) - y * y * y * 0.01


def gen_implicit():
    """The implicit fixed-grid RK methods of the reference's SOLVERS table (fixed_grid_implicit.py, Broyden iterations
    on a dense Jacobian in the reference): tableaus, solves with evaluation counts and warning counts, a tuple
    state, an event solve."""
    import warnings
    from torchdiffeq._impl import fixed_grid_implicit as fgi
    from torchdiffeq._impl.odeint import SOLVERS
    arrays = {}
    _, y0 = linear_problem(3, 6, torch.float32, seed=21)
    t = torch.tensor([0.0, 0.33, 0.7, 1.0])
    arrays.update(y0=y0, t=t)
    arrays["solver_names"] = np.array(list(SOLVERS))

    class Count:
        def __init__(self, fn):
            self.fn, self.nfe = fn, 0

        def __call__(self, t, y):
            self.nfe += 1
            return self.fn(t, y)

    with torch.no_grad():
        for method in IMPLICIT_METHODS:
            tab = SOLVERS[method].tableau
            arrays[f"{method}_alpha"] = tab.alpha
            arrays[f"{method}_beta_flat"] = torch.cat([b.reshape(-1) for b in tab.beta])
            arrays[f"{method}_c_sol"] = tab.c_sol
            arrays[f"{method}_order"] = SOLVERS[method].order
            cases = {
                "grid": (y0, torch.linspace(0, 1, 11), {}),
                "step": (y0, t, dict(step_size=0.05)),
                "perturb": (y0, t, dict(step_size=0.05, perturb=True)),
                "cubic": (y0, t, dict(step_size=0.05, interp="cubic")),
                "rev": (y0, torch.tensor([1.0, 0.45, 0.0]), dict(step_size=0.05)),
                "iters2": (y0, t, dict(step_size=0.1, max_iters=2)),
                "f64": (y0.double(), t.double(), dict(step_size=0.05)),
            }
            for tag, (y, tt, opts) in cases.items():
                c = Count(implicit_field)
                with warnings.catch_warnings(record=True) as w:
                    warnings.simplefilter("always")
                    arrays[f"{method}_{tag}"] = torchdiffeq.odeint(c, y, tt, method=method, options=opts)
                arrays[f"{method}_{tag}_nfe"] = c.nfe
                arrays[f"{method}_{tag}_warnings"] = len(w)
            ft = lambda t, y: (implicit_field(t, y[0]), -y[1] * y[0][0, :3] * (1 + t))
            yt = (y0.double(), torch.tensor([0.5, 0.25, 1.0], dtype=torch.float64))
            out = torchdiffeq.odeint(ft, yt, torch.linspace(0, 1, 11, dtype=torch.float64), method=method)
            arrays[f"{method}_tuple0"], arrays[f"{method}_tuple1"] = out
            fe = lambda t, y: torch.stack([y[1], -y[0]])
            et, ys = torchdiffeq.odeint_event(fe, torch.tensor([1.0, 0.0], dtype=torch.float64),
                                              torch.tensor(0.0, dtype=torch.float64), event_fn=lambda t, y: y[0],
                                              method=method, options=dict(step_size=0.05), atol=1e-8)
            arrays[f"{method}_event_t"], arrays[f"{method}_event_y"] = et, ys
    # gradients of plain odeint (the reference backpropagates through its unrolled Broyden iterations)
    for method in IMPLICIT_METHODS:
        torch.manual_seed(3)
        lin = torch.nn.Linear(3, 3).double()
        yg = rand(5, 3, seed=22).requires_grad_(True)
        tg = torch.linspace(0, 1, 6, dtype=torch.float64).requires_grad_(True)
        fg = lambda t, y: torch.tanh(lin(y)) * torch.cos(t)
        y = torchdiffeq.odeint(fg, yg, tg, method=method)
        loss = y[-1].pow(2).sum() + y[3].sum()
        g = torch.autograd.grad(loss, [yg, tg, lin.weight, lin.bias])
        arrays[f"{method}_bp_w"], arrays[f"{method}_bp_b"] = lin.weight, lin.bias
        arrays[f"{method}_bp_y0"] = yg
        arrays[f"{method}_bp_y"] = y
        for name, v in zip(["gy0", "gt", "gw", "gb"], g):
            arrays[f"{method}_bp_{name}"] = v
    save("implicit.npz", **arrays)


def gen_detest():
    """The reference's integration benchmark (tests/DETEST/run.py) as a parity fixture: 24 classic non-stiff problems
    solved over [0, 20] by the reference with dopri5 (tol 1e-3, 1e-6, 1e-9), tsit5 and dopri8 — solutions, evaluation
    counts.  The problem definitions are restated in tests/_detest.py and checked here against the reference's own."""
    sys.path.insert(0, os.path.join(HERE, ".."))
    sys.path.insert(0, "/root/reference/tests/DETEST")
    import _detest
    import detest as ref_detest
    prev = torch.get_default_dtype()
    torch.set_default_dtype(torch.float64)        # the reference's run.py works in double
    arrays = {}
    try:
        for name, (field, y0) in _detest.problems().items():
            rf, rinit, _ = getattr(ref_detest, name)()
            rt0, ry0 = rinit()
            assert torch.equal(ry0.double().reshape(y0.shape), y0), name
            for tt in (0.0, 0.7, 3.1):
                probe = y0 + 0.01 * (tt + 1)
                assert torch.allclose(field(torch.tensor(tt), probe), rf(torch.tensor(tt), probe).double(), rtol=1e-14,
                                      atol=1e-15), name
            t = torch.tensor([0.0, 20.0])
            for method, tol in [("dopri5", 1e-3), ("dopri5", 1e-6), ("dopri5", 1e-9), ("tsit5", 1e-6),
                                ("dopri8", 1e-9)]:
                nfe = [0]

                def f(t_, y_):
                    nfe[0] += 1
                    return field(t_, y_)
                with torch.no_grad():
                    y = torchdiffeq.odeint(f, y0, t, rtol=tol, atol=tol, method=method)[1]
                key = f"{name}_{method}_{tol:g}"
                arrays[key], arrays[key + "_nfe"] = y, nfe[0]
    finally:
        torch.set_default_dtype(prev)
    save("detest.npz", **arrays)


# ---------------------------------------------------------------------------------------------------
def complex_problem(D, dtype, seed=3):
    g = torch.Generator().manual_seed(seed)
    G = torch.randn(D, D, generator=g, dtype=torch.float64) / D ** 0.5
    H = torch.randn(D, D, generator=g, dtype=torch.float64) / D ** 0.5
    A = (0.5 * (G - G.T) - 0.1 * torch.eye(D, dtype=torch.float64)) + 1j * 0.5 * (H + H.T)
    y0 = torch.randn(5, D, generator=g, dtype=torch.float64) + 1j * torch.randn(5, D, generator=g, dtype=torch.float64)
    return A.to(dtype), y0.to(dtype)


def gen_hostpath():
    """States the HIP kernels do not take (r03, torchdiffeq_amd/_fallback.py): complex states (misc.py:185,
    rk_common.py:61) and second-order gradients through plain odeint (rk_common.py:31-40 is a differentiable op)."""
    arrays = {}
    for tag, dtype, tdtype in (("c64", torch.complex64, torch.float32), ("c128", torch.complex128, torch.float64)):
        A, y0 = complex_problem(6, dtype)
        arrays[f"{tag}_A"], arrays[f"{tag}_y0"] = A, y0
        for method, kw in (("dopri5", dict(rtol=1e-5, atol=1e-7)), ("dopri8", dict(rtol=1e-6, atol=1e-8)),
                           ("rk4", dict(options=dict(step_size=0.05))), ("bosh3", dict(rtol=1e-4, atol=1e-6))):
            for dname, t in (("fwd", torch.tensor([0.0, 0.4, 1.0], dtype=tdtype)),
                             ("rev", torch.tensor([1.0, 0.3, 0.0], dtype=tdtype))):
                y, nfe, c = solve(lambda t_, y_: y_ @ A.T, y0, t, method=method, **kw)
                arrays[f"{tag}_{method}_{dname}_t"] = t
                arrays[f"{tag}_{method}_{dname}_y"] = y
                arrays[f"{tag}_{method}_{dname}_nfe"] = nfe
                arrays[f"{tag}_{method}_{dname}_accept_dt"] = np.array(c.accept)
                arrays[f"{tag}_{method}_{dname}_reject_dt"] = np.array(c.reject)
    # gradient of a real loss through a complex solve (plain odeint, autograd through the solver)
    A, y0 = complex_problem(6, torch.complex128)
    y0g = y0.clone().requires_grad_(True)
    y = torchdiffeq.odeint(lambda t_, y_: y_ @ A.T, y0g, torch.tensor([0.0, 1.0], dtype=torch.float64), method="dopri5",
                           rtol=1e-7, atol=1e-9)
    (y[-1].abs() ** 2).sum().backward()
    arrays["c128_grad_y0"] = y0g.grad
    # second-order gradients, real fp64
    g = torch.Generator().manual_seed(9)
    W0 = torch.randn(4, 4, generator=g, dtype=torch.float64) * 0.5
    x0 = torch.randn(3, 4, generator=g, dtype=torch.float64)
    arrays["hess_W"], arrays["hess_y0"] = W0, x0
    for method, kw in (("rk4", dict(options=dict(step_size=0.1))), ("dopri5", dict(rtol=1e-8, atol=1e-10)),
                       ("midpoint", dict(options=dict(step_size=0.1)))):
        W = W0.clone().requires_grad_(True)
        x = x0.clone().requires_grad_(True)
        y = torchdiffeq.odeint(lambda t_, y_: torch.tanh(y_ @ W.T), x, torch.tensor([0.0, 0.5, 1.0], dtype=torch.float64),
                               method=method, **kw)
        loss = (y[-1] ** 2).sum() + (y[1] ** 3).sum()
        gx, gW = torch.autograd.grad(loss, (x, W), create_graph=True)
        second = (gx ** 2).sum() + (gW ** 2).sum()
        hx, hW = torch.autograd.grad(second, (x, W))
        arrays[f"hess_{method}_gx"], arrays[f"hess_{method}_gW"] = gx.detach(), gW.detach()
        arrays[f"hess_{method}_hx"], arrays[f"hess_{method}_hW"] = hx, hW
    # second order with the output TIMES in the graph (dense-output weights are polynomial in the interpolation point)
    for method, kw in (("rk4", dict(options=dict(step_size=0.1))), ("dopri5", dict(rtol=1e-8, atol=1e-10))):
        W = W0.clone().requires_grad_(True)
        x = x0.clone().requires_grad_(True)
        tt = torch.tensor([0.0, 0.43, 1.0], dtype=torch.float64, requires_grad=True)
        y = torchdiffeq.odeint(lambda t_, y_: torch.tanh(y_ @ W.T) * torch.cos(t_), x, tt, method=method, **kw)
        loss = (y[-1] ** 2).sum() + (y[1] ** 3).sum()
        gx, gt = torch.autograd.grad(loss, (x, tt), create_graph=True)
        second = (gx ** 2).sum() + (gt ** 2).sum()
        hx, ht, hW = torch.autograd.grad(second, (x, tt, W))
        arrays[f"hesst_{method}_gt"], arrays[f"hesst_{method}_hx"] = gt.detach(), hx
        arrays[f"hesst_{method}_ht"], arrays[f"hesst_{method}_hW"] = ht, hW
    save("hostpath.npz", **arrays)


def gen_eager_pin():
    """Pins oracle/eager_torch_port.py (bench.py's `reference_style_eager_gpu` comparator) to the
    reference AS AN OP SEQUENCE: the reference's accepted / rejected (t0, dt) pairs with a given first step, on one CPU
    thread.  A port that issues the same ATen ops in the same order reproduces these fp64 values bit for bit."""
    arrays = {}
    for tag, dtype in (("f32", torch.float32), ("f64", torch.float64)):
        for method, rtol, atol, first in (("dopri5", 1e-5, 1e-7, 0.35), ("dopri8", 1e-7, 1e-9, 0.5), ("bosh3", 1e-4, 1e-6, 0.2)):
            A, y0 = linear_problem(20, 8, dtype, seed=5)
            A = A * 3.0
            steps = {"acc": [], "rej": []}

            class F(torch.nn.Module):
                def forward(self, t, y):
                    return y @ A.T

                def callback_accept_step(self, t0, y_, dt):
                    steps["acc"].append((float(t0), float(dt)))

                def callback_reject_step(self, t0, y_, dt):
                    steps["rej"].append((float(t0), float(dt)))
            with torch.no_grad():
                y = torchdiffeq.odeint(F(), y0, torch.tensor([0.0, 2.0], dtype=dtype), rtol=rtol, atol=atol, method=method,
                                       options=dict(first_step=first))
            key = f"{tag}_{method}"
            arrays[f"{key}_A"], arrays[f"{key}_y0"] = A, y0
            arrays[f"{key}_tol"] = np.array([rtol, atol, first])
            arrays[f"{key}_acc"] = np.array(steps["acc"]).reshape(-1, 2)
            arrays[f"{key}_rej"] = np.array(steps["rej"]).reshape(-1, 2)
            arrays[f"{key}_y"] = y
    save("eager_pin.npz", **arrays)


def gen_dropin():
    """Found by running the reference's own test suite against the package (tools/run_reference_tests.py, r03):
    (1) a 0-dim fp32 state under the Adams methods while something requires grad (fixed_adams.py:205-216: the 0-dim x
    0-dim products promote to fp64 whether or not autograd records) — solution, gradients and the event time of the
    unstable explicit-Adams run of event_tests.py:14-49; (2) what a norm placed in `grad_fn.adjoint_options['norm']`
    is called with when the forward state is a tuple (norm_tests.py:155-191): (t, y, adj_y, *adj_params) with y and
    adj_y FLAT."""
    arrays = {}
    for method in ("explicit_adams", "implicit_adams"):
        for tag, tdtype in (("t32", torch.float32), ("t64", torch.float64)):
            w = torch.tensor(0.7, requires_grad=True)
            x = torch.tensor(1.3, requires_grad=True)
            t = torch.linspace(0.0, 0.5, 26, dtype=tdtype).requires_grad_(True)
            y = torchdiffeq.odeint(lambda t_, y_: -y_ * w * (1 + t_) + torch.sin(3 * t_), x, t, method=method)
            y[-1].backward()
            key = f"adams0_{method}_{tag}"
            arrays[f"{key}_y"], arrays[f"{key}_gx"], arrays[f"{key}_gt"], arrays[f"{key}_gw"] = y.detach(), x.grad, t.grad, w.grad
    # event_tests.py: the sine problem (problems.py:27-37), fp32, 0-dim, explicit Adams at step 0.01 (diverging run: the
    # event time is decided by rounding — 2.3979 in the reference)
    sys.path.insert(0, "/root/reference/tests")         # the reference's own problem set (tests/problems.py:27-37)
    from problems import construct_problem
    f, y0, t_points, sol = construct_problem(dtype=torch.float32, device="cpu", ode="sine")
    t_points, sol = t_points.detach(), sol.detach()
    event_fn = lambda t_, y_: torch.sum(y_ - sol[2]).real
    et, ys = torchdiffeq.odeint(f, y0, t_points[0:2], event_fn=event_fn, method="explicit_adams",
                                options={"step_size": 0.01, "interp": "cubic"})
    arrays["sine_t_points"], arrays["sine_sol"] = t_points, sol
    arrays["sine_event_t"], arrays["sine_event_y"] = et.detach(), ys.detach()

    # (2) tuple forward state: the tuples a replaced adjoint norm receives in the backward solve
    g = torch.Generator().manual_seed(3)
    p1 = torch.rand(7, generator=g).requires_grad_(True)
    p2 = torch.rand((), generator=g).requires_grad_(True)
    x0 = (torch.tensor(1.0), torch.tensor([[0.5, 0.5], [0.1, 0.1]]))
    for tag, kw in (("default", {}), ("seminorm", dict(adjoint_options=dict(norm="seminorm")))):
        xs = torchdiffeq.odeint_adjoint(lambda t_, x_: (x_[0] * p2, x_[1] * p1[:4].reshape(2, 2)), x0,
                                        torch.tensor([0.0, 1.0]), adjoint_params=(p1, p2), **kw)
        opts = xs[0].grad_fn.next_functions[0][0].next_functions[0][0].adjoint_options
        auto = opts["norm"]
        seen = []

        def spy(tensors):
            seen.append(([tuple(v.shape) for v in tensors], float(auto(tensors))))
            return auto(tensors)
        opts["norm"] = spy
        (xs[0].sum() + xs[1].sum()).backward()
        arrays[f"adjnorm_{tag}_n_entries"] = np.array([len(sh) for sh, _ in seen])
        arrays[f"adjnorm_{tag}_y_numel"] = np.array([int(np.prod(sh[1])) for sh, _ in seen])
        arrays[f"adjnorm_{tag}_first_values"] = np.array([v for _, v in seen[:3]])
        arrays[f"adjnorm_{tag}_gp1"], arrays[f"adjnorm_{tag}_gp2"] = p1.grad.clone(), p2.grad.clone()
        p1.grad = p2.grad = None
    arrays["adjnorm_p1"], arrays["adjnorm_p2"] = p1.detach(), p2.detach()

    # (3) a USER-DEFINED tableau on the reference's adaptive machinery (rk_common.py:15, :153-211): Cash–Karp 5(4), which
    # is in neither library's table, written the way the reference's tableaus are (a closing stage at t1 whose row is the
    # solution weights); the package's native solver takes the same table (tests/test_plugin_protocol.py)
    from fractions import Fraction as Fr
    alpha = [Fr(1, 5), Fr(3, 10), Fr(3, 5), Fr(1), Fr(7, 8)]
    a = [[Fr(1, 5)], [Fr(3, 40), Fr(9, 40)], [Fr(3, 10), Fr(-9, 10), Fr(6, 5)],
         [Fr(-11, 54), Fr(5, 2), Fr(-70, 27), Fr(35, 27)],
         [Fr(1631, 55296), Fr(175, 512), Fr(575, 13824), Fr(44275, 110592), Fr(253, 4096)]]
    b5 = [Fr(37, 378), 0, Fr(250, 621), Fr(125, 594), 0, Fr(512, 1771)]
    b4 = [Fr(2825, 27648), 0, Fr(18575, 48384), Fr(13525, 55296), Fr(277, 14336), Fr(1, 4)]
    alpha_f = [float(x) for x in alpha] + [1.0]
    beta_f = [[float(x) for x in r] for r in a] + [[float(x) for x in b5]]
    c_sol = [float(x) for x in b5] + [0.0]
    c_err = [float(x - y) for x, y in zip(b5, b4)] + [0.0]
    mid = [x / 2 for x in c_sol]            # cubic Hermite midpoint: (y0 + y1) / 2 + dt / 8 (f0 - f1)
    mid[0] += 0.125
    mid[-1] -= 0.125
    f64 = lambda v: torch.tensor(v, dtype=torch.float64)

    class RefCashKarp(rk_common.RKAdaptiveStepsizeODESolver):
        order = 5
        tableau = rk_common._ButcherTableau(alpha=f64(alpha_f), beta=[f64(r) for r in beta_f], c_sol=f64(c_sol),
                                            c_error=f64(c_err))
    RefCashKarp.mid = f64(mid)
    from torchdiffeq._impl.odeint import SOLVERS as REF_SOLVERS
    REF_SOLVERS["cashkarp"] = RefCashKarp
    A = torch.tensor([[-0.1, 2.0], [-2.0, -0.1]], dtype=torch.float64)
    y0 = torch.tensor([[2.0, 0.0], [1.0, 0.5]], dtype=torch.float64)
    for tag, t in (("fwd", torch.linspace(0, 2, 7, dtype=torch.float64)), ("rev", torch.linspace(2, 0, 7, dtype=torch.float64))):
        y, nfe, c = solve(lambda t_, y_: torch.tanh(y_ @ A) * torch.cos(t_), y0, t, method="cashkarp", rtol=1e-8, atol=1e-10)
        arrays[f"cashkarp_{tag}_t"], arrays[f"cashkarp_{tag}_y"], arrays[f"cashkarp_{tag}_nfe"] = t, y, nfe
        arrays[f"cashkarp_{tag}_accept_dt"] = np.array(c.accept)
    del REF_SOLVERS["cashkarp"]
    arrays["cashkarp_alpha"], arrays["cashkarp_c_sol"], arrays["cashkarp_c_err"], arrays["cashkarp_mid"] = \
        np.array(alpha_f), np.array(c_sol), np.array(c_err), np.array(mid)
    for i, r in enumerate(beta_f):
        arrays[f"cashkarp_beta{i}"] = np.array(r)
    arrays["cashkarp_A"], arrays["cashkarp_y0"] = A, y0

    # (4) odeint_event gradients incl. the START time (odeint.py:160-231, solvers.py:130-164): the fixed-grid solvers form
    # t1 = t0 + dt and the interpolation fraction on the tensor t0, so d(event time)/d t0 exists for them as well
    for method, opts in (("rk4", dict(step_size=0.03, interp="cubic")), ("euler", dict(step_size=0.01, interp="linear")),
                         ("midpoint", dict(step_size=0.02, interp="cubic")), ("dopri5", {})):
        for rev in (False, True):
            y0g = torch.tensor([1.0, 0.1], dtype=torch.float64, requires_grad=True)
            t0g = torch.tensor(0.3, dtype=torch.float64, requires_grad=True)
            kg = torch.tensor(1.0, dtype=torch.float64, requires_grad=True)
            et, ys = torchdiffeq.odeint_event(lambda t_, y_: torch.stack([y_[1], -y_[0] * kg * (1 + 0.5 * t_)]), y0g, t0g,
                                              event_fn=lambda t_, y_: y_[0] - 0.3, method=method, options=dict(opts),
                                              reverse_time=rev, atol=1e-9, rtol=1e-7)
            g = torch.autograd.grad(et * 2.0 + (ys[-1] ** 2).sum(), [y0g, t0g, kg])
            key = f"evgrad_{method}_{'rev' if rev else 'fwd'}"
            arrays[f"{key}_t"], arrays[f"{key}_y"] = et.detach(), ys.detach()
            arrays[f"{key}_gy0"], arrays[f"{key}_gt0"], arrays[f"{key}_gk"] = g
    # (6) second-order gradients with the output TIMES in the graph and CUBIC Hermite interpolation between grid points
    # (solvers.py:166-173): the basis is cubic in h = (t - t0) / (t1 - t0), so the Hessian needs its curvature
    g = torch.Generator().manual_seed(9)
    W0 = torch.randn(3, 3, generator=g, dtype=torch.float64) * 0.5
    x0 = torch.randn(2, 3, generator=g, dtype=torch.float64)
    arrays["hesscubic_W"], arrays["hesscubic_x"] = W0, x0
    for method, step in (("rk4", 0.1), ("heun3", 0.07), ("euler", 0.05)):
        W = W0.clone().requires_grad_(True)
        x = x0.clone().requires_grad_(True)
        tt = torch.tensor([0.05, 0.43, 0.96], dtype=torch.float64, requires_grad=True)
        y = torchdiffeq.odeint(lambda t_, y_: torch.tanh(y_ @ W.T) * torch.cos(t_), x, tt, method=method,
                               options=dict(step_size=step, interp="cubic"))
        loss = (y[-1] ** 2).sum() + (y[1] ** 3).sum()
        g1 = torch.autograd.grad(loss, (x, W, tt), create_graph=True)
        g2 = torch.autograd.grad(sum((v ** 2).sum() for v in g1), (x, W, tt))
        for name, v in zip(("gx", "gW", "gt", "hx", "hW", "ht"), list(g1) + list(g2)):
            arrays[f"hesscubic_{method}_{name}"] = v.detach()

    # (7) per-component tolerances of the BACKWARD solve with a tuple forward state: given for the reference's backward
    # state (t, y, adj_y, *adj_params) — 3 + P entries (adjoint.py:64-65, misc.py:115-123)
    p1 = torch.tensor([0.5, 0.2], dtype=torch.float64, requires_grad=True)
    p2 = torch.tensor(0.3, dtype=torch.float64, requires_grad=True)
    xg = torch.tensor([1.0, 2.0, 3.0], dtype=torch.float64, requires_grad=True)
    zg = torch.tensor([[0.5, 0.1]], dtype=torch.float64)
    out = torchdiffeq.odeint_adjoint(lambda t_, s: (-s[0] * p1[0] * torch.cos(t_) + s[1].sum() * p2, -s[1] * p1[1]), (xg, zg),
                                     torch.tensor([0.0, 0.6, 1.0], dtype=torch.float64), adjoint_params=(p1, p2), method="dopri5",
                                     rtol=1e-6, atol=1e-8, adjoint_rtol=(1e-3, 1e-6, 1e-5, 1e-4, 1e-4),
                                     adjoint_atol=(1e-4, 1e-8, 1e-7, 1e-6, 1e-6))
    (out[0][-1].pow(2).sum() + out[1][-1].sum()).backward()
    arrays["adjtol_gx"], arrays["adjtol_gp1"], arrays["adjtol_gp2"] = xg.grad, p1.grad, p2.grad

    # (8) PER-ELEMENT tolerances: tensors / lists that broadcast against the state (misc.py:80-82 is plain broadcasting;
    # rk_common.py:186-187 makes them fp64), and vector entries of a tuple tolerance (misc.py:115-123)
    yv = torch.tensor([[1.0, 2.0, 3.0], [0.5, 1.0, 1.5]], dtype=torch.float64)
    cv = torch.tensor([1.0, 5.0, 0.2], dtype=torch.float64)
    tv = torch.tensor([0.0, 0.5, 1.0], dtype=torch.float64)
    cases = {"rtolvec": dict(rtol=torch.tensor([1e-3, 1e-6, 1e-9], dtype=torch.float64), atol=1e-9),
             "atollist": dict(rtol=1e-6, atol=[1e-3, 1e-6, 1e-9]),
             "both": dict(rtol=torch.tensor([1e-3, 1e-6, 1e-7]), atol=[1e-4, 1e-8, 1e-9])}
    for tag, kw in cases.items():
        for method in ("dopri5", "bosh3"):
            y, nfe, c = solve(lambda t_, y_: -y_ * cv * (1 + 0.2 * t_), yv, tv, method=method, **kw)
            arrays[f"vectol_{tag}_{method}_y"], arrays[f"vectol_{tag}_{method}_nfe"] = y, nfe
            arrays[f"vectol_{tag}_{method}_accept_dt"] = np.array(c.accept)
    wv = torch.tensor(0.7, dtype=torch.float64, requires_grad=True)
    xv = yv[0].clone().requires_grad_(True)
    out = torchdiffeq.odeint(lambda t_, s: (-s[0] * cv * wv, -s[1] * 0.3), (xv, torch.ones(2, dtype=torch.float64)), tv,
                             rtol=(torch.tensor([1e-3, 1e-6, 1e-8], dtype=torch.float64), 1e-5),
                             atol=(1e-9, torch.tensor([1e-7, 1e-9], dtype=torch.float64)))
    out[0][-1].pow(2).sum().backward()
    arrays["vectol_tuple_y"], arrays["vectol_tuple_gx"], arrays["vectol_tuple_gw"] = out[0].detach(), xv.grad, wv.grad

    # (5) a 0-dim fp32 state on an fp64 grid WITH the perturb option (misc.py:174-197): the first evaluation time is
    # perturbed in fp32 (the state is still fp32 there), every later one in fp64 (0-dim x 0-dim promotion)
    for method in ("euler", "midpoint", "heun3", "rk4", "explicit_adams", "implicit_adams"):
        for tag, tt in (("fwd", torch.linspace(0.1, 0.6, 11, dtype=torch.float64)),
                        ("rev", torch.linspace(0.6, 0.1, 11, dtype=torch.float64))):
            with torch.no_grad():
                y = torchdiffeq.odeint(lambda t_, y_: -y_ * (1 + 0.3 * t_) + 0.2 * t_ * t_, torch.tensor(0.7), tt, method=method,
                                       options=dict(perturb=True, step_size=0.013))
            arrays[f"zerodim_perturb_{method}_{tag}"] = y
    save("dropin.npz", **arrays)


# ---------------------------------------------------------------------------------------------------
sys.path.insert(0, os.path.dirname(HERE))
from _cases import FUNC_SHAPE_CASES, FUNC_SHAPE_METHODS  # noqa: E402  (shared with tests/test_brow_golden.py)


def gen_brow():
    """SURVEY.md §8(b) corners the r03 differential run found (Weak 2): the adaptive solvers' `dtype` option
    (rk_common.py:176-194), states below fp32 (misc.py:185-187, rk_common.py:61-65 — bf16 runs, fp16 underflows) and
    what the reference does with a func output of the wrong shape."""
    arrays = {}
    # ---- `dtype` option: every time-like scalar in promote_types(dtype, y0.abs().dtype) ----
    for sname, sdtype in (("f32", torch.float32), ("f64", torch.float64)):
        A, y0 = linear_problem(16, 8, sdtype, seed=11)
        arrays[f"dt_{sname}_A"], arrays[f"dt_{sname}_y0"] = A, y0
        for method, kw in (("dopri5", dict(rtol=1e-5, atol=1e-7)), ("dopri8", dict(rtol=1e-5, atol=1e-7)),
                           ("bosh3", dict(rtol=1e-4, atol=1e-6))):
            for oname, odtype in (("o32", torch.float32), ("o64", torch.float64), ("o16", torch.float16)):
                for dname, t in (("fwd", torch.linspace(0.0, 1.5, 7)), ("rev", torch.linspace(1.5, 0.0, 7))):
                    y, nfe, c = solve(lambda t_, y_: (y_ @ A.T) * torch.cos(t_), y0, t, method=method,
                                      options=dict(dtype=odtype), **kw)
                    key = f"dt_{sname}_{method}_{oname}_{dname}"
                    arrays[key + "_y"], arrays[key + "_nfe"] = y, nfe
                    arrays[key + "_accept_dt"], arrays[key + "_reject_dt"] = np.array(c.accept), np.array(c.reject)
        # step_t / jump_t / first_step / event time with dtype=float32
        t = torch.linspace(0.0, 1.0, 4)
        opts = dict(dtype=torch.float32, step_t=torch.tensor([0.31, 0.77]), jump_t=torch.tensor([0.5]), first_step=0.013)
        y, nfe, c = solve(lambda t_, y_: (y_ @ A.T) * torch.cos(t_), y0, t, method="dopri5", rtol=1e-5, atol=1e-7, options=opts)
        arrays[f"dt_{sname}_grid_y"], arrays[f"dt_{sname}_grid_nfe"] = y, nfe
        arrays[f"dt_{sname}_grid_accept_dt"] = np.array(c.accept)
        ev_t, ev_y = torchdiffeq.odeint_event(lambda t_, y_: (y_ @ A.T) * torch.cos(t_), y0, torch.tensor(0.0),
                                              event_fn=lambda t_, y_: y_[0, 0] - 0.5 * y0[0, 0], method="dopri5",
                                              rtol=1e-5, atol=1e-7, options=dict(dtype=torch.float32))
        arrays[f"dt_{sname}_event_t"], arrays[f"dt_{sname}_event_y"] = ev_t, ev_y
        arrays[f"dt_{sname}_event_t_is_f32"] = np.array(ev_t.dtype == torch.float32)
    # adjoint with dtype=float32 (adjoint_options inherit it)
    A, y0 = linear_problem(16, 8, torch.float32, seed=11)
    lin = torch.nn.Linear(8, 8, bias=False)
    with torch.no_grad():
        lin.weight.copy_(A)

    class Field(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.lin = lin

        def forward(self, t, y):
            return self.lin(y) * torch.cos(t)
    y0g = y0.clone().requires_grad_(True)
    y = torchdiffeq.odeint_adjoint(Field(), y0g, torch.tensor([0.0, 0.7, 1.5]), method="dopri5", rtol=1e-5, atol=1e-7,
                                   options=dict(dtype=torch.float32))
    y[-1].pow(2).sum().backward()
    arrays["dt_adj_y"], arrays["dt_adj_gy"], arrays["dt_adj_gW"] = y.detach(), y0g.grad, lin.weight.grad

    # ---- states below fp32 ----
    A2 = torch.tensor([[-0.1, 2.0], [-2.0, -0.1]])
    arrays["low_A"] = A2
    y0 = torch.tensor([[2.0, 0.0], [1.0, 1.0], [-0.5, 0.75]])
    arrays["low_y0"] = y0
    for lname, ldtype in (("bf16", torch.bfloat16), ("f16", torch.float16)):
        for method, kw in (("dopri5", dict(rtol=1e-2, atol=1e-3)), ("dopri8", dict(rtol=1e-2, atol=1e-3)),
                           ("bosh3", dict(rtol=1e-2, atol=1e-3)), ("tsit5", dict(rtol=1e-2, atol=1e-3)),
                           ("adaptive_heun", dict(rtol=1e-2, atol=1e-3)), ("rk4", {}),
                           ("rk4_step", dict(options=dict(step_size=0.0625)))):
            for dname, t in (("fwd", torch.linspace(0.0, 1.0, 5)), ("rev", torch.linspace(1.0, 0.0, 5))):
                key = f"low_{lname}_{method}_{dname}"
                try:
                    y, nfe, c = solve(lambda t_, y_: (y_ @ A2.to(y_.dtype).T) * torch.cos(t_), y0.to(ldtype), t,
                                      method=method.split("_step")[0], **kw)
                    arrays[key + "_y"], arrays[key + "_nfe"] = y.float(), nfe
                    arrays[key + "_raises"] = np.array("")
                    assert y.dtype == ldtype
                except AssertionError as exc:
                    arrays[key + "_raises"] = np.array(str(exc))
        # tuple state, explicit dtype below fp32 (W = the state's own type), adjoint
    for method in ("dopri5", "bosh3"):
        ya, yb = y0.to(torch.bfloat16), y0[:, :1].to(torch.bfloat16) * 0.5
        out = torchdiffeq.odeint(lambda t_, y_: ((y_[0] @ A2.to(torch.bfloat16).T), -y_[1] * y_[0][:, :1]), (ya, yb),
                                 torch.linspace(0.0, 1.0, 3), method=method, rtol=1e-2, atol=1e-3)
        arrays[f"low_tuple_{method}_a"], arrays[f"low_tuple_{method}_b"] = out[0].float(), out[1].float()
        y = torchdiffeq.odeint(lambda t_, y_: y_ @ A2.to(torch.bfloat16).T, ya, torch.linspace(0.0, 1.0, 3), method=method,
                               rtol=1e-2, atol=1e-3, options=dict(dtype=torch.bfloat16))
        arrays[f"low_w16_{method}_y"] = y.float()

    # bf16 through odeint_adjoint: forward rows, dL/dy0, dL/dW
    for method in ("dopri5", "bosh3", "rk4"):
        torch.manual_seed(1)
        lin = torch.nn.Linear(4, 4).to(torch.bfloat16)
        x = torch.randn(6, 4).to(torch.bfloat16).requires_grad_(True)
        if method == "dopri5":
            arrays["low_adj_W"], arrays["low_adj_b"], arrays["low_adj_y0"] = lin.weight.detach().float(), lin.bias.detach().float(), x.detach().float()

        class LowField(torch.nn.Module):
            def __init__(self):
                super().__init__()
                self.lin = lin

            def forward(self, t, y):
                return torch.tanh(self.lin(y))
        y = torchdiffeq.odeint_adjoint(LowField(), x, torch.tensor([0.0, 0.5, 1.0]), method=method, rtol=1e-2, atol=1e-3)
        y[-1].float().pow(2).sum().backward()
        arrays[f"low_adj_{method}_y"], arrays[f"low_adj_{method}_gy"] = y.detach().float(), x.grad.float()
        arrays[f"low_adj_{method}_gW"] = lin.weight.grad.float()

    # ---- func outputs of the wrong shape: which (method, case) pairs the reference accepts ----
    table = np.zeros((len(FUNC_SHAPE_METHODS), len(FUNC_SHAPE_CASES)), dtype=np.int8)
    finals = np.zeros(table.shape, dtype=np.float64)
    for i, method in enumerate(FUNC_SHAPE_METHODS):
        for j, (name, (shape, view)) in enumerate(FUNC_SHAPE_CASES.items()):
            if shape and isinstance(shape[0], tuple):
                state = tuple(torch.arange(1, 1 + int(np.prod(s)), dtype=torch.float64).reshape(s) for s in shape)
            else:
                state = torch.arange(1, 1 + int(np.prod(shape)), dtype=torch.float64).reshape(shape)
            try:
                with torch.no_grad():
                    y = torchdiffeq.odeint(lambda t_, y_: view(y_), state, torch.tensor([0.0, 0.5, 1.0], dtype=torch.float64),
                                           method=method, rtol=1e-4, atol=1e-6)
                table[i, j] = 1
                last = y[0][-1] if isinstance(y, tuple) else y[-1]
                finals[i, j] = float(last.reshape(-1)[-1])
            except RuntimeError:
                table[i, j] = 0
    # ---- known residue (docs/LAB_NOTEBOOK.md §8): a 0-dim fp32 state on an fp64 time grid under an adaptive method ----
    y = torchdiffeq.odeint(lambda t_, y_: -y_ * torch.cos(t_), torch.tensor(1.5), torch.linspace(0.0, 2.0, 5, dtype=torch.float64),
                           method="dopri5")
    arrays["zero_dim_f32_on_f64_grid_dopri5"] = y
    arrays["shape_methods"] = np.array(FUNC_SHAPE_METHODS)
    arrays["shape_cases"] = np.array(list(FUNC_SHAPE_CASES))
    arrays["shape_accepts"], arrays["shape_final"] = table, finals
    save("brow.npz", **arrays)


def gen_r4b():
    """Found by the program-level differential runs of round 4b (tools/fuzz_programs_vs_reference.py,
    tools/fuzz_api_programs_vs_reference.py): 16-bit states under EVERY explicit fixed-grid method, also on a 16-bit time
    grid (solvers.py:108-126 with t.dtype bf16 / fp16; rk_common.py:110-157 — Python weights are second operands);
    callbacks see a tensor state in its own shape (misc.py:313-333 wraps only tuple states); cubic interpolation
    evaluates func at the step end once per output time (solvers.py:119-122)."""
    arrays = {}
    A2 = torch.tensor([[-0.1, 2.0, 0.3], [-2.0, -0.1, 0.0], [0.2, 0.1, -0.5]])
    y0 = torch.tensor([[2.0, 0.0, 0.3], [1.0, 1.0, -1.0]])
    arrays["low_A"], arrays["low_y0"] = A2, y0
    for lname, ldtype in (("bf16", torch.bfloat16), ("f16", torch.float16)):
        for tname, tdtype in (("t32", torch.float32), ("tlow", ldtype)):
            for method in ("euler", "midpoint", "heun2", "heun3", "rk4"):
                for oname, opts in (("plain", {}), ("cubic", dict(step_size=0.13, interp="cubic")),
                                    ("perturb", dict(perturb=True))):
                    for dname, tv in (("fwd", [0.0, 0.3, 0.55, 1.0]), ("rev", [1.0, 0.55, 0.3, 0.0])):
                        t = torch.tensor(tv).to(tdtype)
                        Al = A2.to(ldtype)
                        y, nfe, _ = solve(lambda t_, y_: y_ @ Al.T - y_ * 0.5 * torch.cos(t_).to(y_.dtype),
                                          y0.to(ldtype), t, method=method, options=dict(opts))
                        assert y.dtype == ldtype
                        key = f"lowgrid_{lname}_{tname}_{method}_{oname}_{dname}"
                        arrays[key + "_y"], arrays[key + "_nfe"] = y.float(), nfe

    # callback arguments: a [2, 3] tensor state, both directions
    for method, opts in (("dopri5", {}), ("rk4", dict(step_size=0.25))):
        for dname, tv in (("fwd", [0.0, 0.5, 1.0]), ("rev", [1.0, 0.5, 0.0])):
            seen = []

            class CbField(torch.nn.Module):
                def forward(self, t, y):
                    return y @ A2.T * torch.cos(t)

                def callback_step(self, t0, y_, dt):
                    seen.append(("step", tuple(y_.shape), float(t0), float(dt), float(y_.sum())))

                def callback_accept_step(self, t0, y_, dt):
                    seen.append(("accept", tuple(y_.shape), float(t0), float(dt), float(y_.sum())))
            torchdiffeq.odeint(CbField(), y0, torch.tensor(tv), method=method, rtol=1e-4, atol=1e-6, options=dict(opts))
            key = f"cb_{method}_{dname}"
            arrays[key + "_kind"] = np.array([s[0] for s in seen])
            arrays[key + "_shape"] = np.array([s[1] for s in seen])
            arrays[key + "_vals"] = np.array([s[2:] for s in seen])

    # evaluation times of a cubic-interpolated fixed-grid solve with several output times per step
    calls = []

    def counted(t_, y_):
        calls.append(float(t_))
        return -y_ * (1.0 + t_)
    y = torchdiffeq.odeint(counted, y0, torch.tensor([0.0, 0.1, 0.2, 0.25, 0.7, 1.0]), method="heun2",
                           options=dict(step_size=0.5, interp="cubic"))
    arrays["cubic_calls"], arrays["cubic_y"] = np.array(calls), y

    # the adjoint norms take their time component as `t.abs()` (adjoint.py:250, 273), NOT as an rms: a time VJP whose
    # scaled value is 1e29 squares to inf in fp32, and an inf norm would make the backward solve's first step 0
    class Steep(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.w = torch.nn.Parameter(torch.tensor([0.5, -0.25]))

        def forward(self, t, y):
            return -y + self.w * torch.sin(t * 1e20) * 1e3
    for nname, norm in (("mixed", None), ("semi", "seminorm")):
        f = Steep()
        x = torch.tensor([[1.0, 2.0]], requires_grad=True)
        y = torchdiffeq.odeint_adjoint(f, x, torch.tensor([0.0, 1e-18]), method="dopri5", rtol=1e-3, atol=1e-6,
                                       adjoint_options=dict(norm=norm) if norm else None)
        y[-1].sum().backward()
        arrays[f"steep_{nname}_y"], arrays[f"steep_{nname}_gy"], arrays[f"steep_{nname}_gw"] = y.detach(), x.grad, f.w.grad

    # grid_constructor(func, y0, t): y0 in the state's own form — a tensor state in its shape, a tuple state (and the
    # adjoint's augmented one) as the UNPADDED concatenation — and func usable on it (solvers.py:103)
    for sname in ("tensor2d", "tuple"):
        seen = []
        w = torch.tensor([0.5, -0.3, 0.8], requires_grad=True)

        def grid(func, y, tt):
            d = func(tt[0], y)
            seen.append((tuple(y.shape), tuple(d.shape), float(d.abs().max())))
            return torch.linspace(float(tt[0]), float(tt[-1]), 2 + int(float(d.abs().max()) * 3)).to(tt)
        if sname == "tensor2d":
            f = lambda t_, y_: -y_ * w * (1 + t_) + torch.sin(y_)       # noqa: E731
            state = torch.tensor([[1.0, 2.0, 3.0], [0.5, 0.1, -1.0]], requires_grad=True)
        else:
            f = lambda t_, y_: (-y_[0] * w * (1 + t_), torch.sin(y_[1]) - y_[0].sum())     # noqa: E731
            state = (torch.tensor([[1.0, 2.0, 3.0]], requires_grad=True), torch.tensor([0.5, 0.1]))
        sol = torchdiffeq.odeint_adjoint(f, state, torch.tensor([0.0, 0.4, 1.0]), method="rk4",
                                         options=dict(grid_constructor=grid), adjoint_params=(w,))
        (sol[0] if sname == "tuple" else sol)[-1].sum().backward()
        arrays[f"grid_{sname}_yshapes"] = np.array([s[0][0] if len(s[0]) == 1 else -1 for s in seen])
        arrays[f"grid_{sname}_dmax"] = np.array([s[2] for s in seen])
        arrays[f"grid_{sname}_gw"] = w.grad
        arrays[f"grid_{sname}_y"] = (sol[0] if sname == "tuple" else sol).detach()

    # complex128 through odeint_adjoint: the backward solve's error ratio is formed on the concatenated augmented state
    # (|z| and z / real round position-dependently in ATen's vectorised loops) — a case where per-segment evaluation
    # gave another last bit in step 77 of the backward solve
    # (the values of the fuzz case that showed it: tools/fuzz_api_programs_vs_reference.py, seed 32, case 5)
    yc = torch.view_as_complex(torch.tensor(
        [0.9763743281364441, 0.4763857126235962, 0.3485761880874634, 1.0541315078735352, -0.37900879979133606,
         0.24454638361930847, 0.20677243173122406, 0.1580955982208252, 0.5700197219848633, 0.023011041805148125,
         0.09842822700738907, -1.2679963111877441], dtype=torch.float64).reshape(2, 3, 2))
    Ac = torch.view_as_complex(torch.tensor(
        [-1.0376341342926025, -0.14303676784038544, -0.03842546045780182, -0.5508574843406677, 0.22360268235206604,
         -0.5132613182067871, -0.1185542568564415, -0.09384410828351974, -0.07820449024438858, -0.5325219035148621,
         -0.5970337986946106, 0.1508079320192337, 0.046314921230077744, 0.8552271127700806, 0.10754086822271347,
         0.02906198613345623, -0.2525499761104584, 0.6338606476783752], dtype=torch.float64).reshape(3, 3, 2))
    arrays["cplx_y0"], arrays["cplx_A"] = torch.view_as_real(yc), torch.view_as_real(Ac)
    for method in ("bosh3", "dopri5"):
        a_ = Ac.clone().requires_grad_(True)
        x = yc.clone().requires_grad_(True)
        y = torchdiffeq.odeint_adjoint(lambda t_, y_: y_ @ a_ - y_ * 0.5, x, torch.linspace(0.0, 1.0, 4, dtype=torch.float64),
                                       method=method, adjoint_params=(a_,))
        y[-1].abs().sum().backward()
        arrays[f"cplx_{method}_y"] = torch.view_as_real(y.detach())
        arrays[f"cplx_{method}_gA"], arrays[f"cplx_{method}_gy"] = torch.view_as_real(a_.grad), torch.view_as_real(x.grad)

    # non-finite stages under heun3: the reference multiplies EVERY stage by its weight, also the tableau's zeros
    # (fixed_grid.py:38-44: `k1 * 0.0 + k2 * (2/3)`), so an inf stage turns the rows into NaN, not inf
    yb = torchdiffeq.odeint(lambda t_, y_: torch.where(t_ > 0.4, torch.full_like(y_, float("inf")), -y_),
                            torch.tensor([1.0, 2.0, 0.5], dtype=torch.float64), torch.tensor([0.0, 1.0, 3.0], dtype=torch.float64),
                            method="heun3", options=dict(step_size=0.25))
    arrays["heun3_inf_field_y"] = yb

    # ... and in an adaptive method's error row: near the blow-up of y' = y^2 one dopri8 stage is inf while its weight in
    # c_error is 0 -> `inf * 0 = NaN` error estimate -> NaN step size; the solve ends in `underflow in dt 0.0` after 470
    # evaluations (a row sum over the non-zero stages only stays finite and needs 1133)
    calls = []

    def square(t_, y_):
        calls.append(1)
        return y_[0] * y_[0], -y_[1]
    state = (torch.tensor([1.1101932525634766, 1.453037977218628, 1.1249815225601196]), torch.ones(2))
    try:
        with torch.no_grad():
            torchdiffeq.odeint(square, state, torch.tensor([0.0, 1.0, 3.0]), method="dopri8", rtol=1e-3, atol=1e-6,
                               options=dict(max_num_steps=200))
        msg = ""
    except AssertionError as exc:
        msg = str(exc)
    arrays["blowup_dopri8_calls"], arrays["blowup_dopri8_message"] = np.array(len(calls)), np.array(msg)

    # event solve on a trajectory that turns NaN (a diverged training run): the bisection takes the NaN sign for a sign
    # change and still returns a finite time, the state NaN; gradients NaN
    class NanField(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.w = torch.nn.Parameter(torch.tensor(1.0))

        def forward(self, t_, y_):
            return torch.where(t_ > 0.55, torch.full_like(y_, float("nan")), -y_ * self.w)
    for iface in ("odeint", "odeint_adjoint"):
        f = NanField()
        x = torch.tensor([1.0, 2.0], requires_grad=True)
        et, ys = torchdiffeq.odeint_event(f, x, torch.tensor(0.0), event_fn=lambda t_, y_: y_[0] - 0.1, method="rk4",
                                          options=dict(step_size=0.1), odeint_interface=getattr(torchdiffeq, iface),
                                          atol=1e-6, rtol=1e-4)   # (at atol 1e-9 the fp32 bisection ends ON the grid point: finite)
        (et + 0).backward()
        arrays[f"nan_event_{iface}_t"], arrays[f"nan_event_{iface}_y"] = et.detach(), ys.detach()
        arrays[f"nan_event_{iface}_gw"], arrays[f"nan_event_{iface}_gy"] = f.w.grad, x.grad
    save("r4b.npz", **arrays)


def gen_programs():
    """40 training-loop programs of tools/fuzz_programs_vs_reference.py (seed 7) run by the REFERENCE: every logged value
    — solution, loss, each parameter gradient, dL/dy0, dL/dt, the evaluation count after every SGD iteration, event time
    and state — as tests/golden/programs.npz; tests/test_programs_golden.py replays the same programs on the package."""
    sys.path.insert(0, os.path.join(HERE, "..", "..", "tools"))
    argv, sys.argv = sys.argv, ["fuzz_programs_vs_reference.py", "7", "40"]
    try:
        import fuzz_programs_vs_reference as fz
    finally:
        sys.argv = argv
    assert fz.ref is torchdiffeq
    arrays = {}
    for case_no in range(40):
        case = fz.make_case(fz.rng)
        log = fz.run(torchdiffeq, case)
        arrays[f"p{case_no}_n"] = np.array(len(log))
        arrays[f"p{case_no}_desc"] = np.array(f"{case['kind'].__name__} {case['api']} {case['method']} {case['kw']}")
        for i, (name, value) in enumerate(log):
            arrays[f"p{case_no}_{i}_name"] = np.array(name)
            if value is None:
                arrays[f"p{case_no}_{i}_none"] = np.array(1)
            elif torch.is_tensor(value):
                arrays[f"p{case_no}_{i}_val"] = value
            else:
                arrays[f"p{case_no}_{i}_val"] = np.array(value)
    save("programs.npz", **arrays)


if __name__ == "__main__":
    only = sys.argv[1:]
    for name, fn in [("tableaus", gen_tableaus), ("kernels", gen_kernel_vectors), ("controller", gen_controller_vectors),
                     ("solves", gen_solves), ("dopri8_small", gen_dopri8_small), ("int_state", gen_int_state), ("adjoint", gen_adjoint), ("cnf", gen_cnf), ("methods", gen_methods), ("events", gen_events), ("backprop", gen_backprop), ("tuple_tol", gen_tuple_tolerances), ("adjoint_tdep", gen_adjoint_time_dependent),
                     ("adams", gen_adams), ("implicit", gen_implicit), ("detest", gen_detest), ("hostpath", gen_hostpath), ("eager_pin", gen_eager_pin), ("dropin", gen_dropin), ("brow", gen_brow), ("r4b", gen_r4b), ("programs", gen_programs)]:
        if not only or name in only:
            fn()
