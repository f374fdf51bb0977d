"""Eager-PyTorch restatement of the reference's adaptive RK trial step — what rtqichen/torchdiffeq v0.2.5 executes
when its state lives on a GPU: a chain of stock ATen tensor ops with the time-like scalars kept as 0-dim DEVICE
tensors, so that every Python `if` / `assert` / `max` on them is a device->host synchronisation.

TEST INFRASTRUCTURE ONLY (like the rest of oracle/): used by tests/ to pin it to the numpy oracle and by bench.py's
`reference_style_eager_gpu` leg, which times it on the same MI355X next to the HIP path.  The reference itself cannot
travel to the GPU box, so this port stands in for "the reference on cuda" of SURVEY.md §8(d); it follows the
reference's op sequence function by function (paths relative to torchdiffeq/_impl/):

    rk_step            rk_common.py:43-90    stage-minor k[*shape, S+1]; k[..., :i+1] * (beta_i * dt) summed over the
                                             stage axis; FSAL shortcut or the c_sol combine; error = k @ (c_error*dt)
    error_ratio        misc.py:80-82         atol + rtol * max(|y0|, |y1|); rms norm (misc.py:22-23)
    optimal_step_size  misc.py:85-95         under no_grad, tensors all the way
    interp_fit         interp.py:1-22 via rk_common.py:363-369   (computed on every accepted step, as the reference does)
    adaptive_step      rk_common.py:266-361  guards (isfinite(y0).all(), dt underflow), accept / reject bookkeeping

Tableau coefficients are the reference's own fp64 values from tests/golden/tableaus.npz, cast to the state dtype on
the state's device once (rk_common.py:201-205)."""
from __future__ import annotations

import torch

from .reference_solver import tableau as _ref_tableau


class EagerTableau:
    def __init__(self, name: str, dtype, device):
        ref = _ref_tableau(name)
        cast = lambda a: torch.as_tensor(a, dtype=dtype, device=device)
        self.order = ref.order
        self.alpha = cast(ref.alpha)
        self.beta = [cast(b) for b in ref.beta]
        self.c_sol = cast(ref.c_sol)
        self.c_error = cast(ref.c_error)
        self.c_mid = cast(ref.c_mid)


def rms_norm(x):                                             # misc.py:22-23
    return x.abs().pow(2).mean().sqrt()


def rk_step(func, y0, f0, t0, dt, t1, tab: EagerTableau):    # rk_common.py:43-90
    t_dtype = y0.abs().dtype
    t0, dt, t1 = t0.to(t_dtype), dt.to(t_dtype), t1.to(t_dtype)
    k = torch.empty(*f0.shape, len(tab.alpha) + 1, dtype=y0.dtype, device=y0.device)
    k[..., 0] = f0
    for i, (alpha_i, beta_i) in enumerate(zip(tab.alpha, tab.beta)):
        if alpha_i == 1.0:                                   # a sync per stage, as in the reference
            ti = t1
        else:
            ti = t0 + alpha_i * dt
        yi = y0 + torch.sum(k[..., :i + 1] * (beta_i * dt), dim=-1).view_as(f0)
        k[..., i + 1] = func(ti, yi)
    if not (tab.c_sol[-1] == 0 and (tab.c_sol[:-1] == tab.beta[-1]).all()):     # two syncs per step
        yi = y0 + torch.sum(k * (dt * tab.c_sol), dim=-1).view_as(f0)
    y1 = yi
    f1 = k[..., -1]
    y1_error = torch.sum(k * (dt * tab.c_error), dim=-1)
    return y1, f1, y1_error, k


def error_ratio(err, rtol, atol, y0, y1):                    # misc.py:80-82
    tol = atol + rtol * torch.max(y0.abs(), y1.abs())
    return rms_norm(err / tol).abs()


@torch.no_grad()
def optimal_step_size(last_step, ratio, safety, ifactor, dfactor, order):       # misc.py:85-95
    if ratio == 0:
        return last_step * ifactor
    if ratio < 1:
        dfactor = torch.ones((), dtype=last_step.dtype, device=last_step.device)
    ratio = ratio.type_as(last_step)
    exponent = torch.tensor(order, dtype=last_step.dtype, device=last_step.device).reciprocal()
    factor = torch.min(ifactor, torch.max(safety / ratio ** exponent, dfactor))
    return last_step * factor


def interp_fit(y0, y1, k, dt, tab: EagerTableau):            # rk_common.py:363-369 + interp.py:1-22
    dt = dt.type_as(y0)
    y_mid = y0 + k.matmul(dt * tab.c_mid).view_as(y0)
    f0, f1 = k[..., 0], k[..., -1]
    a = 2 * dt * (f1 - f0) - 8 * (y1 + y0) + 16 * y_mid
    b = dt * (5 * f0 - 3 * f1) + 18 * y0 + 14 * y1 - 32 * y_mid
    c = dt * (f1 - 4 * f0) - 11 * y0 - 5 * y1 + 16 * y_mid
    d = dt * f0
    return [y0, d, c, b, a]


class EagerAdaptiveRK:
    """State + one trial step per call (rk_common.py:161-361), all scalars 0-dim fp64 tensors on the state's device."""

    def __init__(self, func, y0, t0: float, first_step: float, rtol: float, atol: float, method: str = "dopri5"):
        dev = y0.device
        scalar = lambda v: torch.as_tensor(v, dtype=torch.float64, device=dev)
        self.func = func
        self.tab = EagerTableau(method, y0.dtype, dev)
        self.rtol, self.atol = scalar(rtol), scalar(atol)
        self.safety, self.ifactor, self.dfactor = scalar(0.9), scalar(10.0), scalar(0.2)
        self.min_step, self.max_step = scalar(0.0), scalar(float("inf"))
        self.y, self.t, self.dt = y0, scalar(t0), scalar(first_step)
        self.f = func(self.t.to(y0.dtype), y0)
        self.interp_coeff = [y0] * 5
        self.n_accepted = self.n_rejected = 0

    def adaptive_step(self):                                 # rk_common.py:266-361
        y0, f0, t0, dt = self.y, self.f, self.t, self.dt
        if not torch.isfinite(dt):
            dt = self.min_step
        dt = dt.clamp(self.min_step, self.max_step)
        t1 = t0 + dt
        assert t0 + dt > t0, "underflow in dt {}".format(dt.item())
        assert torch.isfinite(y0).all(), "non-finite values in state `y`"
        y1, f1, y1_error, k = rk_step(self.func, y0, f0, t0, dt, t1, self.tab)
        ratio = error_ratio(y1_error, self.rtol, self.atol, y0, y1)
        accept = ratio <= 1
        if dt > self.max_step:
            accept = False
        if dt <= self.min_step:
            accept = True
        if accept:
            self.interp_coeff = interp_fit(y0, y1, k, dt, self.tab)
            self.y, self.f, self.t = y1, f1, t1
            self.n_accepted += 1
        else:
            self.n_rejected += 1
        dt_next = optimal_step_size(dt, ratio, self.safety, self.ifactor, self.dfactor, self.tab.order)
        self.dt = dt_next.clamp(self.min_step, self.max_step)
        return bool(accept)
