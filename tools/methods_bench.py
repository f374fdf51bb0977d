"""Wall-clock of the methods beyond the explicit-RK hot path at the benchmark state (65536 x 128 fp32, the linear
field of BASELINE.json configs[1]) on one MI355X: the Adams multistep methods and the implicit RK methods (matrix-free
Broyden).  For each: time per grid step, evaluations per step and the solver's own share (total minus func time
measured separately).  The reference cannot run the implicit methods at this size at all (dense (stages*N)^2
Jacobian); it runs the Adams methods through ~2*order+12 eager ops per step.
Prints one JSON object; committed as profiles/<tag>_methods_bench.json."""
import json
import os
import sys
import time
import warnings

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torchdiffeq_amd as tda  # noqa: E402

dev = torch.device("cuda:0")
B, D = 65536, 128
g = torch.Generator().manual_seed(0)
G = torch.randn(D, D, generator=g, dtype=torch.float64) / D ** 0.5
A = (0.5 * (G - G.T) - 0.1 * torch.eye(D, dtype=torch.float64)).float().to(dev)
y0 = torch.randn(B, D, generator=g, dtype=torch.float64).float().to(dev)
At = A.T.contiguous()


class Counting:
    def __init__(self):
        self.nfe = 0

    def __call__(self, t, y):
        self.nfe += 1
        return y @ At


def timed(fn, reps=2):
    fn()
    torch.cuda.synchronize()
    best = None
    for _ in range(reps):
        t = time.perf_counter()
        out = fn()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t
        best = dt if best is None else min(best, dt)
    return best, out


# func alone
f = Counting()
tt = torch.zeros((), device=dev)
w_f, _ = timed(lambda: [f(tt, y0) for _ in range(200)])
t_func = w_f / 200
res = {"state": [B, D], "dtype": "f32", "func_us": t_func * 1e6}
n_steps = int(os.environ.get("TDEQ_BENCH_STEPS", "8"))
t = torch.linspace(0.0, 1.0, n_steps + 1, device=dev)
exact = y0.double() @ torch.linalg.matrix_exp(A.double()).T
with torch.no_grad(), warnings.catch_warnings():
    warnings.simplefilter("ignore")
    # implicit RK: the reference's absolute 2-norm bound (1e-6) is below the fp32 rounding floor of a 8.4M x stages
    # residual, so every step would run all 100 Broyden iterations (measured: 135-350 ms per step); the opt-in RMS
    # form of the same bound is what is timed here
    rms = dict(residual_norm="rms")
    for method, kw, extra in [("rk4", {}, {}), ("explicit_adams", {}, {}),
                              ("implicit_adams", dict(rtol=1e-5, atol=1e-7), {}),
                              ("implicit_euler", {}, rms), ("implicit_midpoint", {}, rms), ("trapezoid", {}, rms),
                              ("radauIIA3", {}, rms), ("gl4", {}, rms), ("radauIIA5", {}, rms), ("gl6", {}, rms),
                              ("sdirk2", {}, rms), ("trbdf2", {}, rms)]:
        if len(sys.argv) > 1 and method not in sys.argv[1:]:
            continue
        f = Counting()
        w, y = timed(lambda: tda.odeint(f, y0, t[[0, -1]], method=method,
                                        options=dict(step_size=1.0 / n_steps, **extra), **kw))
        nfe = f.nfe // 3
        err = float((y[-1].double() - exact).abs().max() / exact.abs().max())
        res[method] = {"wall_ms": w * 1e3, "ms_per_step": w * 1e3 / n_steps, "nfe_per_step": nfe / n_steps,
                       "solver_share_ms_per_step": (w - nfe * t_func) * 1e3 / n_steps, "rel_err_vs_expm": err}
        print(method, res[method], flush=True)
print(json.dumps(res))
