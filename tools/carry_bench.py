"""Carried partial sums on / off (TDEQ_CARRY) on the MI355X: ms per trial step in the middle of a long solve, whole
`odeint(t=[0,1])` wall time and the solver kernels alone (events, no func), for cfg2 (dopri5 fp32 65536x128), cfg4
(dopri8 fp64 16384x512) and their 1/8 shards.  Writes gpurun_out/carry_bench.json (copy to profiles/)."""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torchdiffeq_amd as tda  # noqa: E402
import _fullsize as fs  # noqa: E402
from torchdiffeq_amd import tableaus as tb  # noqa: E402
from torchdiffeq_amd.misc import OdeFunc, StateLayout, rms_norm  # noqa: E402
from torchdiffeq_amd.solvers import Dopri5Solver, Dopri8Solver, Tsit5Solver  # noqa: E402

dev = torch.device("cuda:0")


def stepper(cls, field, y0, rtol, atol):
    layout = StateLayout([y0.shape], False)
    func = OdeFunc(field, layout, 1.0, y0.dtype, y0.device)
    s = cls(func=func, y0=y0.reshape(-1), rtol=rtol, atol=atol, norm=rms_norm)
    s._before_integrate([0.0])
    s._t_end = float("inf")
    return s


def ms_per_step(s, steps, warmup, blocks=5):
    with torch.no_grad():
        for _ in range(warmup):
            s._trial_step()
        out = []
        for _ in range(blocks):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                s._trial_step()
            torch.cuda.synchronize()
            out.append(1e3 * (time.perf_counter() - t0) / steps)
    return sorted(out)[len(out) // 2]


def solver_kernels_only(s, reps=30):
    """The step's solver launches back to back on the last step's stage tensors (no func), ms per step."""
    rec, kern = s._dense, s.kernels
    ks, y0 = rec.k, rec.y0
    rows, S = s._beta, len(s._beta)
    plan = s._carry
    dt = rec.dt_signed

    def one():
        held = {}
        o = torch.empty_like(y0)
        kern.stage_combine(o, y0, [ks[0]], rows[0].coef, dt)
        for i in range(1, S):
            op = plan.ops[i] if plan is not None else None
            if plan is not None and op is None:
                held.pop(i)
                continue
            if plan is None or (len(op.targets) == 1 and not op.continues):
                if plan is None and i == S - 1:
                    ep = torch.empty_like(y0)
                    kern.stage_combine_err(torch.empty_like(y0), ep, y0, [ks[j] for j in rows[i].idx], rows[i].coef,
                                           s._fuse[0], dt)
                    held[S] = ep
                else:
                    kern.stage_combine(torch.empty_like(y0), y0, [ks[j] for j in rows[i].idx], rows[i].coef, dt)
                continue
            outs = [torch.empty_like(y0) for _ in op.targets]
            kern.stage_combine_multi(outs, op.spec, y0, held.pop(i) if op.continues else None,
                                     [ks[j] for j in op.idx], dt)
            for t_, b in zip(op.targets[1:], outs[1:]):
                held[t_] = b
        rem = (plan.err_idx, plan.err_coef) if plan is not None else s._fuse[1:]
        kern.error_norm_partial(s.plan, held.pop(S), y0, rec.y1, [ks[j] for j in rem[0]], rem[1], dt)
    for _ in range(3):
        one()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        one()
    e1.record()
    torch.cuda.synchronize()
    s.plan.expect = ()
    return e0.elapsed_time(e1) / reps


def case(name, cls, B, D, dtype, method, rtol, atol, steps):
    A, y0 = fs.linear_problem(B, D, dtype)
    At, y0 = A.T.contiguous().to(dev), y0.to(dev)
    field = lambda t, y: y @ At
    t = torch.tensor([0.0, 1.0], dtype=dtype, device=dev)
    out = {"state": f"{B} x {D} {str(dtype).split('.')[-1]}", "method": method}
    sols = {}
    for carry in ("0", "1"):
        os.environ["TDEQ_CARRY"] = carry
        s = stepper(cls, field, y0, rtol, atol)
        r = {"ms_per_trial_step": ms_per_step(s, steps, max(steps // 5, 3))}
        if s.tableau.fsal_solution:
            r["solver_kernels_only_ms_per_step"] = solver_kernels_only(s)
        with torch.no_grad():
            tda.odeint(field, y0, t, rtol=rtol, atol=atol, method=method)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(3):
                sols[carry] = tda.odeint(field, y0, t, rtol=rtol, atol=atol, method=method)
            torch.cuda.synchronize()
            r["odeint_t01_ms"] = 1e3 * (time.perf_counter() - t0) / 3
        tab = tb.ADAPTIVE_TABLEAUS[method]
        r["words_per_element_and_step"] = tb.carry_plan(method).words if carry == "1" else tb.row_by_row_words(tab)
        out["carry_on" if carry == "1" else "carry_off"] = r
        del s
    out["solutions_bit_identical"] = bool(torch.equal(sols["0"], sols["1"]))
    for key in ("ms_per_trial_step", "solver_kernels_only_ms_per_step", "odeint_t01_ms"):
        if key in out["carry_on"]:
            out["gain_" + key] = 1.0 - out["carry_on"][key] / out["carry_off"][key]
    print(name, json.dumps(out), flush=True)
    return out


CASES = {
    "cfg2": lambda: case("cfg2", Dopri5Solver, 65536, 128, torch.float32, "dopri5", 1e-7, 1e-9, 100),
    "cfg4": lambda: case("cfg4", Dopri8Solver, 16384, 512, torch.float64, "dopri8", 1e-9, 1e-11, 30),
    "cfg2_shard": lambda: case("cfg2_shard", Dopri5Solver, 8192, 128, torch.float32, "dopri5", 1e-7, 1e-9, 100),
    "cfg4_shard": lambda: case("cfg4_shard", Dopri8Solver, 2048, 512, torch.float64, "dopri8", 1e-9, 1e-11, 30),
    "dopri8_f32_cfg2_state": lambda: case("dopri8_f32", Dopri8Solver, 65536, 128, torch.float32, "dopri8", 1e-7, 1e-9, 30),
    "tsit5_cfg2_state": lambda: case("tsit5_f32", Tsit5Solver, 65536, 128, torch.float32, "tsit5", 1e-7, 1e-9, 60),
    "tsit5_f64_cfg4_state": lambda: case("tsit5_f64", Tsit5Solver, 16384, 512, torch.float64, "tsit5", 1e-9, 1e-11, 30),
}
names = sys.argv[1:] or list(CASES)
res = {name: CASES[name]() for name in names}
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
tag = "" if not sys.argv[1:] else "_" + "_".join(names)
json.dump(res, open(os.path.join(ROOT, "gpurun_out", f"carry_bench{tag}.json"), "w"), indent=1)
