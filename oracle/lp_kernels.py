"""CPU restatement of the reference's Runge–Kutta step arithmetic for bfloat16 / float16 STATES.

TEST INFRASTRUCTURE ONLY (like rk_oracle.c): the parity oracle of torchdiffeq_amd/csrc/tdeq_kernels_lp.hpp.  Only tests/
may import it; the product never does.

The reference integrates a reduced-precision state with ordinary ATen ops on tensors of that type (every time-like
scalar is cast to `y0.abs().dtype`: torchdiffeq/_impl/rk_common.py:61-65, misc.py:185-187), so the restatement IS the
reference's own torch expressions, evaluated by ATen's CPU kernels on bf16 / fp16 tensors — each function cites the lines
it repeats (paths relative to torchdiffeq/_impl/).  Two deliberate differences, the same the fp32 / fp64 oracle has
(docs/LAB_NOTEBOOK.md §8): a tableau row is summed over its NON-ZERO weights, left to right (`_row_sum`: products rounded to the
state's type, accumulated in float32, rounded once — what `torch.sum(k * c, dim=-1)` does for a reduced-precision
tensor, with the order fixed), and norms are reported as fp64 sums of the rounded squares.

Pinning: tests/test_lowp_oracle.py checks these functions against vectors the imported reference produced on bf16 / fp16
inputs (tests/golden/make_golden_lowp.py -> tests/golden/lowp_kernels.npz).
"""
from __future__ import annotations

from typing import List, Sequence, Tuple

import torch


def _t(x: float, dtype) -> torch.Tensor:
    """A host number as the reference holds it next to a state of `dtype`: a 0-dim tensor of that type."""
    return torch.tensor(float(x), dtype=torch.float64).to(torch.float32).to(dtype)


def _row_sum(ks: Sequence[torch.Tensor], coefs: Sequence[float], dt: float) -> torch.Tensor:
    """`torch.sum(k[..., :n] * (beta_i * dt), dim=-1)` (rk_common.py:79, 89, 366) over the non-zero weights: the
    tableau row is cast to the state's type (rk_common.py:201-205), `beta_i * dt` is a product of two tensors of that
    type, `k * c` rounds every product, the sum accumulates them in float32 and rounds once."""
    dtype = ks[0].dtype
    dtT = _t(dt, dtype)
    acc = None
    for k, c in zip(ks, coefs):
        cT = _t(c, dtype) * dtT                       # fl_T(fl_T(c) * fl_T(dt))
        p = (k * cT).float()                          # the rounded product, widened exactly
        acc = p if acc is None else acc + p           # float32 accumulation, left to right
    return acc.to(dtype)


def stage_combine(y0, ks, coefs, dt):
    """yi = y0 + sum(k * (beta_i * dt))        rk_common.py:79 (and :83-85; misc.py:65 with one term)"""
    return y0 + _row_sum(ks, coefs, dt)


def stage_combine_err(y0, ks, coefs, err_coefs, dt):
    """(y1, the error row over the same stages)        rk_common.py:79 / :89"""
    return y0 + _row_sum(ks, coefs, dt), _row_sum(ks, err_coefs, dt)


def error_quotient(y0, y1, ks, coefs, dt, rtol, atol):
    """y1_error / (atol + rtol * max(|y0|, |y1|))        rk_common.py:89, misc.py:80-82.  rtol / atol are 0-dim tensors
    (rk_common.py:186-187) and FIRST operands here: ATen casts them to the state's type."""
    err = _row_sum(ks, coefs, dt)
    dtype = y0.dtype
    tol = _t(atol, dtype) + _t(rtol, dtype) * torch.max(y0.abs(), y1.abs())
    return err / tol


def norm_terms(q) -> Tuple[float, float]:
    """(sum of fl(|q|^2), sum of |q|) in fp64: the terms of `sqrt(mean(|x|^2))` (misc.py:22-23) before the mean."""
    a = q.abs()
    return float(a.pow(2).double().sum()), float(a.double().sum())


def init_quotients(mode, a, b, y, rtol, atol):
    """misc.py:50-66: scale = atol + |y0| * rtol with rtol the SECOND operand (a 0-dim fp64 tensor next to a
    reduced-precision tensor: ATen's CPU kernel takes it at float32), atol cast to the state's type;
    mode 0: (y0 / scale, f0 / scale); mode 1: (f1 - f0) / scale."""
    rt, at = torch.tensor(float(rtol), dtype=torch.float64), _t(atol, y.dtype)
    scale = at + y.abs() * rt
    assert scale.dtype == y.dtype
    if mode == 0:
        return a / scale, b / scale
    return (a - b) / scale, None


def quartic(y0, y1, f0, f1, ks, coefs, dt):
    """rk_common.py:363-369 + interp.py:17-21, literally, with dt a 0-dim tensor of the state's type."""
    dtype = y0.dtype
    dtT = _t(dt, dtype)
    y_mid = y0 + _row_sum(ks, coefs, dt)
    a = 2 * dtT * (f1 - f0) - 8 * (y1 + y0) + 16 * y_mid
    b = dtT * (5 * f0 - 3 * f1) + 18 * y0 + 14 * y1 - 32 * y_mid
    c = dtT * (f1 - 4 * f0) - 11 * y0 - 5 * y1 + 16 * y_mid
    d = dtT * f0
    e = y0
    return [e, d, c, b, a]


def dense_eval(y0, y1, f0, f1, ks, coefs, dt, x):
    """interp.py:38-48: x cast to the state's type, `total = coefficients[0] + x * coefficients[1]`, then
    `x_power = x_power * x; total = total + x_power * coefficient` for the rest."""
    co = quartic(y0, y1, f0, f1, ks, coefs, dt)
    xT = _t(x, y0.dtype)
    total = co[0] + xT * co[1]
    x_power = xT
    for coefficient in co[2:]:
        x_power = x_power * xT
        total = total + x_power * coefficient
    return total


def _time(x: float) -> torch.Tensor:
    """A time-like scalar of a FIXED-GRID solver: a 0-dim tensor in `t.dtype` (solvers.py:102-128 never casts the grid
    to the state's type) — next to a reduced-precision tensor ATen rounds it to that type as a first operand and takes
    it at float32 as a second operand."""
    return torch.tensor(float(x), dtype=torch.float64)


def rk4_stage(stage, y0, k1, k2, k3, k4, dt):
    """rk_common.py:110-118 (3/8 rule), the stage inputs and the step: dt a 0-dim tensor in the grid's type,
    `_one_third` / 0.125 / 3 Python numbers."""
    dtT = _time(dt)
    third = 1.0 / 3.0
    if stage == 1:
        return y0 + dtT * k1 * third
    if stage == 2:
        return y0 + dtT * (k2 - k1 * third)
    if stage == 3:
        return y0 + dtT * (k1 - k2 + k3)
    return y0 + (k1 + 3 * (k2 + k3) + k4) * dtT * 0.125


def lerp(y0, y1, slope):
    """solvers.py:175-181 `_linear_interp`: y0 + slope * (y1 - y0), slope a 0-dim tensor in the grid's type."""
    return y0 + _time(slope) * (y1 - y0)


def fixed_stage(mode, y0, ks, ws, dt):
    """rk_common.py:121-157 / fixed_grid.py: mode 1 = y0 + dt * k0 * w0; mode 0 = y0 + dt * (k0 * w0 + k1 * w1 ...)
    (dt a 0-dim tensor in the grid's type, the weights Python numbers, the sum a chain of elementwise additions)."""
    dtT = _time(dt)
    if mode == 1:
        return y0 + dtT * ks[0] * ws[0]
    acc = ks[0] * ws[0]
    for k, w in zip(ks[1:], ws[1:]):
        acc = acc + k * w
    return y0 + dtT * acc


def weighted_sum(xs, ws):
    """x0 * w0 + x1 * w1 + ... with the weights 0-dim tensors of the state's type (the Adams / backward helpers)."""
    acc = xs[0] * _t(ws[0], xs[0].dtype)
    for x, w in zip(xs[1:], ws[1:]):
        acc = acc + x * _t(w, x.dtype)
    return acc
