"""Golden vectors for bfloat16 / float16 STATES: outputs of the REFERENCE's own step functions on reduced-precision
tensors (CPU), stored as float32 (exact for both types).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_lowp.py      (build container only: /root/reference)

-> tests/golden/lowp_kernels.npz.  Pins oracle/lp_kernels.py (tests/test_lowp_oracle.py), which in turn is what the HIP
kernels of csrc/tdeq_kernels_lp.hpp are compared with bit for bit on the GPU (tests/test_lowp_gpu.py).
"""
import os
import sys

import numpy as np
import torch

sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference")

from torchdiffeq._impl import bosh3, dopri5, dopri8, interp, misc, rk_common, tsit5  # noqa: E402
from torchdiffeq._impl.misc import Perturb  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
torch.set_num_threads(1)


def rand(*shape, seed, dtype):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g, dtype=torch.float64).to(dtype)


class Replay:
    """A `func` that returns pre-drawn stage derivatives and records the stage inputs the solver builds."""

    def __init__(self, ks):
        self.ks, self.i, self.seen_y, self.seen_t = ks, 0, [], []

    def __call__(self, t, y, perturb=Perturb.NONE):
        self.seen_t.append(float(t))
        self.seen_y.append(y.clone())
        out = self.ks[self.i]
        self.i += 1
        return out


def main():
    arrays = {}
    n = 1031
    f32 = lambda t: t.to(torch.float32)
    for dname, dtype in [("bf16", torch.bfloat16), ("f16", torch.float16)]:
        for mname, solver_cls in [("dopri5", dopri5.Dopri5Solver), ("dopri8", dopri8.Dopri8Solver),
                                  ("tsit5", tsit5.Tsit5Solver), ("bosh3", bosh3.Bosh3Solver)]:
            key = f"{mname}_{dname}"
            y0 = rand(n, seed=1, dtype=dtype)
            S = len(solver_cls.tableau.alpha)
            ks = [rand(n, seed=10 + j, dtype=dtype) for j in range(S + 1)]
            rtol, atol = 1e-2, 1e-3
            solver = solver_cls(func=None, y0=y0, rtol=rtol, atol=atol, norm=misc._rms_norm)
            t0 = torch.tensor(0.3, dtype=torch.float64)
            dt = torch.tensor(0.0371, dtype=torch.float64)
            replay = Replay(ks[1:])
            y1, f1, y1_error, k = rk_common._runge_kutta_step(replay, y0, ks[0], t0, dt, t0 + dt, solver.tableau)
            tol = solver.atol + solver.rtol * torch.max(y0.abs(), y1.abs())        # misc.py:81
            quotient = y1_error / tol                                                # misc.py:82 (before the norm)
            ratio = misc._compute_error_ratio(y1_error, solver.rtol, solver.atol, y0, y1, misc._rms_norm)
            coeffs = solver._interp_fit(y0, y1, k, dt)
            t_evals = [t0 + 0.3 * dt, t0 + 0.9 * dt]
            y_evals = [interp._interp_evaluate(coeffs, t0, t0 + dt, te) for te in t_evals]
            assert y1.dtype == dtype and quotient.dtype == dtype and coeffs[2].dtype == dtype
            arrays[f"{key}_y0"] = f32(y0)
            arrays[f"{key}_k"] = f32(torch.stack(ks))
            arrays[f"{key}_t0_dt"] = torch.stack([t0, dt])
            arrays[f"{key}_stage_inputs"] = f32(torch.stack(replay.seen_y))
            arrays[f"{key}_y1"] = f32(y1)
            arrays[f"{key}_y1_error"] = f32(y1_error)
            arrays[f"{key}_quotient"] = f32(quotient)
            arrays[f"{key}_error_ratio"] = ratio.to(torch.float64)
            arrays[f"{key}_rtol_atol"] = torch.tensor([rtol, atol], dtype=torch.float64)
            arrays[f"{key}_interp_coeffs"] = f32(torch.stack(coeffs))
            arrays[f"{key}_x_evals"] = torch.tensor([0.3, 0.9], dtype=torch.float64)
            arrays[f"{key}_t_evals"] = torch.stack(t_evals)
            arrays[f"{key}_y_evals"] = f32(torch.stack(y_evals))
        # rk4 3/8 rule: the grid (t0, dt) stays in t.dtype = float32 (solvers.py:102-128)
        y0 = rand(n, seed=1, dtype=dtype)
        ks = [rand(n, seed=10 + j, dtype=dtype) for j in range(4)]
        replay = Replay(ks[1:])
        t0 = torch.tensor(0.3, dtype=torch.float32)
        dt = torch.tensor(0.025, dtype=torch.float32)
        dy = rk_common.rk4_alt_step_func(replay, t0, dt, t0 + dt, y0, f0=ks[0])
        arrays[f"rk4_{dname}_y0"] = f32(y0)
        arrays[f"rk4_{dname}_k"] = f32(torch.stack(ks))
        arrays[f"rk4_{dname}_dt"] = dt.to(torch.float64)
        arrays[f"rk4_{dname}_stage_inputs"] = f32(torch.stack(replay.seen_y))
        arrays[f"rk4_{dname}_y1"] = f32(y0 + dy)
        # linear interpolation between grid points (solvers.py:175-181)
        ya, yb = rand(n, seed=31, dtype=dtype), rand(n, seed=32, dtype=dtype)
        ta, tb, tq = (torch.tensor(v, dtype=torch.float32) for v in (0.25, 0.5, 0.3125))
        slope = (tq - ta) / (tb - ta)
        arrays[f"lerp_{dname}_ya_yb"] = f32(torch.stack([ya, yb]))
        arrays[f"lerp_{dname}_slope"] = slope.to(torch.float64)
        arrays[f"lerp_{dname}_out"] = f32(ya + slope * (yb - ya))
        # initial-step quotients (misc.py:50-66) — d0, d1 and the scale they share
        A = rand(16, 16, seed=3, dtype=dtype) / 4
        yy = rand(40, 16, seed=4, dtype=dtype)
        f0 = yy @ A.T
        rtol_t, atol_t = torch.tensor(1e-2, dtype=torch.float64), torch.tensor(1e-3, dtype=torch.float64)
        scale = atol_t + torch.abs(yy) * rtol_t
        assert scale.dtype == dtype
        arrays[f"init_{dname}_y0_f0"] = f32(torch.stack([yy, f0]))
        arrays[f"init_{dname}_q0"] = f32(yy / scale)
        arrays[f"init_{dname}_q1"] = f32(f0 / scale)
        arrays[f"init_{dname}_d0_d1"] = torch.stack([misc._rms_norm(yy / scale), misc._rms_norm(f0 / scale)]).to(torch.float64)
    out = {k: (v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)) for k, v in arrays.items()}
    np.savez_compressed(os.path.join(HERE, "lowp_kernels.npz"), **out)
    print("lowp_kernels.npz", len(out), "arrays")


if __name__ == "__main__":
    main()
