"""Host drivers of the explicit Runge–Kutta solvers (dopri5, dopri8, rk4) over the HIP kernels.

The accept/reject loop stays on the host (one per process / shard): per trial step it issues S
`stage_combine` launches interleaved with the user's `func`, one fused `error_norm` launch, reads back
n_seg doubles, and runs the step controller in Python doubles — instead of the reference's ≈220 eager
ops and ≈19 device->host syncs per trial step.  Control flow and numerics follow

  RKAdaptiveStepsizeODESolver   torchdiffeq/_impl/rk_common.py:161-369
  _runge_kutta_step             rk_common.py:43-90
  _select_initial_step / _compute_error_ratio / _optimal_step_size   misc.py:36-95
  _interp_fit / _interp_evaluate  interp.py:1-48 (fused, evaluated lazily: only for requested outputs)
  FixedGridODESolver / RK4      solvers.py:52-181, fixed_grid.py:24-29, rk_common.py:110-118

with time-like scalars (t0, t1, dt, rtol, ...) as host doubles instead of 0-dim device tensors.
"""
from __future__ import annotations

import bisect
import math
from typing import List, Optional, Sequence

import numpy as np
import torch

from . import _native
from .misc import (BuiltinNorm, OdeFunc, Perturb, StateLayout, find_event, handle_unused_kwargs, rms_norm)
from .misc import _null_callback as _null
from .tableaus import ADAPTIVE_HEUN, BOSH3, DOPRI5, DOPRI8, FEHLBERG2, TSIT5, SparseRow, Tableau


def _nan_max(a: float, b: float) -> float:
    """torch.max semantics: NaN propagates."""
    if math.isnan(a) or math.isnan(b):
        return math.nan
    return max(a, b)


def _nan_min(a: float, b: float) -> float:
    if math.isnan(a) or math.isnan(b):
        return math.nan
    return min(a, b)


def _clamp(x: float, lo: float, hi: float) -> float:
    """torch.clamp semantics for host doubles (NaN stays NaN)."""
    if math.isnan(x):
        return x
    return min(max(x, lo), hi)


def _as_float(x) -> float:
    if isinstance(x, torch.Tensor):
        return float(x.item())
    return float(x)


def optimal_step_size(last_step: float, error_ratio: float, safety: float, ifactor: float,
                      dfactor: float, order: int) -> float:
    """Next step size — the reference's I-controller (misc.py:85-95) in host doubles."""
    if error_ratio == 0:
        return last_step * ifactor
    if error_ratio < 1:
        dfactor = 1.0
    exponent = 1.0 / order
    try:
        scaled = safety / error_ratio ** exponent
    except (OverflowError, ZeroDivisionError):
        scaled = math.inf
    factor = _nan_min(ifactor, _nan_max(scaled, dfactor))
    return last_step * factor


class _DenseRecord:
    """Data of the last accepted step, kept for lazy dense output (rk_common.py:363-369)."""
    __slots__ = ("y0", "y1", "k", "dt_signed", "t0", "t1")


class RKAdaptiveStepsizeODESolver:
    """Adaptive embedded RK pair driven from the host; subclasses set `order` and `tableau`."""
    order: int
    tableau: Tableau

    def __init__(self, func: OdeFunc, y0: torch.Tensor, rtol, atol, min_step=0, max_step=float("inf"),
                 first_step=None, step_t=None, jump_t=None, safety=0.9, ifactor=10.0, dfactor=0.2,
                 max_num_steps=2 ** 31 - 1, dtype=torch.float64, norm=None, **unused_kwargs):
        handle_unused_kwargs(self, unused_kwargs)
        del unused_kwargs
        if not isinstance(func, OdeFunc):
            raise TypeError("solver classes of torchdiffeq_amd take the wrapped func built by check_inputs")
        self.func = func
        self.y0 = y0
        self.layout: StateLayout = func.layout
        self.state_dtype = y0.dtype
        self.np_dtype = np.float32 if y0.dtype == torch.float32 else np.float64
        self.dtype = torch.promote_types(dtype, y0.dtype)   # accepted for API parity; host math is fp64
        self.norm = rms_norm if norm is None else norm
        self.rtol, self.atol = rtol, atol
        self.min_step = _as_float(min_step)
        self.max_step = _as_float(max_step)
        self.first_step = None if first_step is None else _as_float(first_step)
        self.safety = _as_float(safety)
        self.ifactor = _as_float(ifactor)
        self.dfactor = _as_float(dfactor)
        self.max_num_steps = int(_as_float(max_num_steps))
        self.step_t = None if step_t is None else torch.as_tensor(step_t, dtype=torch.float64).reshape(-1).tolist()
        self.jump_t = None if jump_t is None else torch.as_tensor(jump_t, dtype=torch.float64).reshape(-1).tolist()

        self.kernels = _native.get_kernels(y0.device)
        self.plan = self.kernels.make_plan(self.layout.segments(rtol, atol), self.layout.total,
                                           self.layout.chunk, y0.device)
        tab = self.tableau
        self._beta = tab.beta_rows()
        self._c_err = SparseRow.from_dense(tab.c_error)
        self._c_mid = SparseRow.from_dense(tab.c_mid)
        self._c_sol = SparseRow.from_dense(tab.c_sol)
        # stage abscissae rounded to the state dtype, as the reference's tableau cast (rk_common.py:201)
        self._alpha = [self.np_dtype(a) for a in tab.alpha]
        self._alpha_is_one = [a == 1.0 for a in tab.alpha]
        self.n_accepted = 0
        self.n_rejected = 0

    @classmethod
    def valid_callbacks(cls):
        return {"callback_step", "callback_accept_step", "callback_reject_step"}

    # -- norms -------------------------------------------------------------------------------------
    def _segment_norm(self, sumsq: Sequence[float], bad: Sequence[float]):
        """max over the selected segments of sqrt(mean), rounded to the state dtype (misc.py:22-33)."""
        numels = self.plan.numels
        n = len(numels)
        if isinstance(self.norm, BuiltinNorm) and self.norm.n_skip_tail:
            n -= self.norm.n_skip_tail
        val = 0.0
        for s in range(n):
            if numels[s] == 0:
                continue
            val = _nan_max(val, math.sqrt(sumsq[s] / numels[s]))
        with np.errstate(over="ignore"):
            return float(self.np_dtype(val))

    def _time_tensor(self, value: float) -> torch.Tensor:
        return torch.tensor(value, dtype=torch.float64, device=self.y0.device)

    # -- integrate ---------------------------------------------------------------------------------
    def integrate(self, t: torch.Tensor) -> torch.Tensor:
        """solution[len(t), total] with solution[0] = y0 (solvers.py:28-35)."""
        t_host = t.detach().to(torch.float64).cpu().tolist()
        solution = torch.empty(len(t_host), self.layout.total, dtype=self.y0.dtype, device=self.y0.device)
        solution[0].copy_(self.y0)
        self._before_integrate(t_host)
        for i in range(1, len(t_host)):
            self._advance(t_host[i], solution[i])
        return solution

    def integrate_dense(self, t: torch.Tensor):
        """Integrate over [t[0], t[-1]] keeping the dense output of EVERY accepted step (odeint.py:124-147):
        returns (times, coeffs) with `times` the n_steps + 1 accepted step boundaries (host doubles) and
        `coeffs[n_steps, 5, total]` the quartic coefficients [e, d, c, b, a] (`tdeq_interp_fit`)."""
        t_host = t.detach().to(torch.float64).cpu().tolist()
        self._before_integrate(t_host)
        times, planes = [self.t0], []
        mid = self._c_mid
        for next_t in t_host[1:]:
            n_steps = 0
            while next_t > self.t1:
                assert n_steps < self.max_num_steps, \
                    "max_num_steps exceeded ({}>={})".format(n_steps, self.max_num_steps)
                accepted_before = self.n_accepted
                self._adaptive_step()
                n_steps += 1
                if self.n_accepted != accepted_before:
                    rec = self._dense
                    buf = torch.empty(5, self.layout.total, dtype=self.y0.dtype, device=self.y0.device)
                    self.kernels.interp_fit(buf, rec.y0, rec.y1, rec.k[0], rec.k[-1], [rec.k[j] for j in mid.idx],
                                            mid.coef, rec.dt_signed)
                    times.append(rec.t1)
                    planes.append(buf)
        coeffs = torch.stack(planes) if planes else torch.empty(0, 5, self.layout.total, dtype=self.y0.dtype,
                                                                 device=self.y0.device)
        return times, coeffs

    def integrate_until_event(self, t0: torch.Tensor, event_fn):
        """(event_t, solution[2, total]): step until `event_fn(t, y)` changes sign, then bisect on the last
        step's dense output (solvers.py:44-49, rk_common.py:252-264, event_handling.py:5-20)."""
        self._before_integrate([float(t0.detach())])
        event_time, y1 = self._advance_until_event(event_fn)
        solution = torch.stack([self.y0, y1], dim=0)
        return self._time_tensor(float(event_time)), solution

    def _advance_until_event(self, event_fn):
        ev = lambda: event_fn(self._time_tensor(self.t1), self.y1)
        if ev() == 0:
            return self.t1, self.y1
        n_steps = 0
        sign0 = float(torch.sign(ev()))
        while sign0 == float(torch.sign(ev())):
            assert n_steps < self.max_num_steps, \
                "max_num_steps exceeded ({}>={})".format(n_steps, self.max_num_steps)
            self._adaptive_step()
            n_steps += 1

        def interp_fn(t):
            out = torch.empty_like(self.y0)
            self._interp_evaluate(float(t), out)
            return out

        atol = self.atol
        if isinstance(atol, torch.Tensor):
            atol = atol.min().item()
        elif not isinstance(atol, (int, float)):
            atol = min(float(a) for a in atol)
        return find_event(interp_fn, sign0, self.t0, self.t1, event_fn, float(atol), self._time_tensor)

    def _before_integrate(self, t_host: List[float]) -> None:
        t0 = t_host[0]
        f0 = self.func.eval(t0, self.y0)
        if self.first_step is None:
            first_step = self._select_initial_step(t0, self.y0, f0)
        else:
            first_step = self.first_step
            # no initial-step heuristic -> still take the non-finite census of y0 (rk_common.py:287)
            self.kernels.init_norms(self.plan, 1, f0, f0, self.y0)
            _, _, bad = self.kernels.read_norms(self.plan)
            self._y_nonfinite = any(b != 0 for b in bad)
        self.y1, self.f1 = self.y0, f0
        self.t0, self.t1, self.dt = t0, t0, first_step
        self._dense: Optional[_DenseRecord] = None

        step_t = [] if self.step_t is None else sorted(v for v in self.step_t if v >= t0)
        jump_t = [] if self.jump_t is None else sorted(v for v in self.jump_t if v >= t0)
        both = step_t + jump_t
        if len(set(both)) != len(both):
            raise ValueError("`step_t` and `jump_t` must not have any repeated elements between them.")
        self._step_t, self._jump_t = step_t, jump_t
        self.next_step_index = min(bisect.bisect(step_t, t0), len(step_t) - 1)
        self.next_jump_index = min(bisect.bisect(jump_t, t0), len(jump_t) - 1)

    def _select_initial_step(self, t0: float, y0: torch.Tensor, f0: torch.Tensor) -> float:
        """Hairer II.4 starting step (misc.py:36-77), scalars in the state precision T."""
        T = self.np_dtype
        kern, plan = self.kernels, self.plan
        order = self.order - 1   # the reference passes `self.order - 1` (rk_common.py:217)
        kern.init_norms(plan, 0, y0, f0, y0)
        s0, s1, bad = kern.read_norms(plan)
        self._y_nonfinite = any(b != 0 for b in bad)
        d0 = T(self._segment_norm(s0, bad))
        d1 = T(self._segment_norm(s1, bad))
        if d0 < 1e-5 or d1 < 1e-5:
            h0 = T(1e-6)
        else:
            h0 = T(T(0.01) * d0) / d1
        h0 = abs(h0)
        y1 = torch.empty_like(y0)
        kern.stage_combine(y1, y0, [f0], [1.0], float(h0) * self.func.sign)
        f1 = self.func.eval(t0 + float(h0), y1)
        kern.init_norms(plan, 1, f1, f0, y0)
        s2, _, bad = kern.read_norms(plan)
        with np.errstate(all="ignore"):
            d2 = abs(T(self._segment_norm(s2, bad)) / h0)
            if d1 <= 1e-15 and d2 <= 1e-15:
                h1 = max(T(1e-6), T(h0 * T(1e-3)))
            else:
                h1 = T(T(0.01) / max(d1, d2)) ** T(1.0 / float(order + 1))
            h1 = abs(h1)
            return float(min(T(100) * h0, h1))

    def _advance(self, next_t: float, out: torch.Tensor) -> None:
        """Step until next_t is inside the last accepted step, then write y(next_t) into `out`."""
        n_steps = 0
        while next_t > self.t1:
            assert n_steps < self.max_num_steps, \
                "max_num_steps exceeded ({}>={})".format(n_steps, self.max_num_steps)
            self._adaptive_step()
            n_steps += 1
        self._interp_evaluate(next_t, out)

    def _interp_evaluate(self, t: float, out: torch.Tensor) -> None:
        """Fused `_interp_fit` + `_interp_evaluate` (rk_common.py:363-369, interp.py:25-48)."""
        rec = self._dense
        assert rec is not None and rec.t0 <= t <= rec.t1, \
            "invalid interpolation, fails `t0 <= t <= t1`: {}, {}, {}".format(self.t0, t, self.t1)
        x = float(self.np_dtype((t - rec.t0) / (rec.t1 - rec.t0)))
        mid = self._c_mid
        self.kernels.dense_eval(out, rec.y0, rec.y1, rec.k[0], rec.k[-1], [rec.k[j] for j in mid.idx],
                                mid.coef, rec.dt_signed, x)

    def _adaptive_step(self) -> None:
        """One trial step (rk_common.py:266-361)."""
        func, kern, T = self.func, self.kernels, self.np_dtype
        y0, f0, t0, dt = self.y1, self.f1, self.t1, self.dt
        if not math.isfinite(dt):
            dt = self.min_step
        dt = _clamp(dt, self.min_step, self.max_step)
        if func.callback_step is not _null:
            func.callback_step(self._time_tensor(t0), y0, self._time_tensor(dt))
        t1 = t0 + dt
        assert t0 + dt > t0, "underflow in dt {}".format(dt)
        assert not self._y_nonfinite, "non-finite values in state `y`: {}".format(y0)

        on_step_t = False
        if len(self._step_t):
            next_step_t = self._step_t[self.next_step_index]
            on_step_t = t0 < next_step_t < t0 + dt
            if on_step_t:
                t1 = next_step_t
                dt = t1 - t0
        on_jump_t = False
        if len(self._jump_t):
            next_jump_t = self._jump_t[self.next_jump_index]
            on_jump_t = t0 < next_jump_t < t0 + dt
            if on_jump_t:
                on_step_t = False
                t1 = next_jump_t
                dt = t1 - t0

        # ---- Runge–Kutta stages (rk_common.py:43-90); times in the state precision T ----
        t0_T, dt_T, t1_T = T(t0), T(dt), T(t1)
        dt_signed = float(dt_T) * func.sign
        stage_times = func.time_tensors(kern, [
            (t1_T, Perturb.PREV) if self._alpha_is_one[i] else (t0_T + self._alpha[i] * dt_T, Perturb.NONE)
            for i in range(len(self._beta))])
        k: List[torch.Tensor] = [f0]
        yi = y0
        for i, row in enumerate(self._beta):
            yi = torch.empty_like(y0)
            kern.stage_combine(yi, y0, [k[j] for j in row.idx], row.coef, dt_signed)
            k.append(func.eval_at(stage_times[i], yi))
        if self.tableau.fsal_solution:
            y1 = yi
        else:
            y1 = torch.empty_like(y0)
            kern.stage_combine(y1, y0, [k[j] for j in self._c_sol.idx], self._c_sol.coef, dt_signed)
        f1 = k[-1]

        # ---- error ratio (misc.py:80-82) ----
        err = self._c_err
        if isinstance(self.norm, BuiltinNorm):
            kern.error_norm(self.plan, y0, y1, [k[j] for j in err.idx], err.coef, dt_signed)
            sumsq, _, bad = kern.read_norms(self.plan)
            error_ratio = self._segment_norm(sumsq, bad)
            y1_nonfinite = any(b != 0 for b in bad)
        else:
            error_ratio, y1_nonfinite = self._user_norm_ratio(y0, y1, k, dt_signed)
        accept_step = error_ratio <= 1
        if dt > self.max_step:
            accept_step = False
        if dt <= self.min_step:
            accept_step = True

        # ---- update state (rk_common.py:335-361) ----
        if accept_step:
            if func.callback_accept_step is not _null:
                func.callback_accept_step(self._time_tensor(t0), y0, self._time_tensor(dt))
            rec = _DenseRecord()
            rec.y0, rec.y1, rec.k, rec.dt_signed, rec.t0, rec.t1 = y0, y1, k, dt_signed, t0, t1
            self._dense = rec
            if on_step_t and self.next_step_index != len(self._step_t) - 1:
                self.next_step_index += 1
            if on_jump_t:
                if self.next_jump_index != len(self._jump_t) - 1:
                    self.next_jump_index += 1
                f1 = func.eval(t1, y1, Perturb.NEXT)
            self.y1, self.f1, self.t0, self.t1 = y1, f1, t0, t1
            self._y_nonfinite = y1_nonfinite
            self.n_accepted += 1
        else:
            if func.callback_reject_step is not _null:
                func.callback_reject_step(self._time_tensor(t0), y0, self._time_tensor(dt))
            self.t0 = t0   # (y, f, t1) unchanged: the step is retried from t0 with a smaller dt
            self.n_rejected += 1
        dt_next = optimal_step_size(dt, error_ratio, self.safety, self.ifactor, self.dfactor, self.order)
        self.dt = _clamp(dt_next, self.min_step, self.max_step)

    def _user_norm_ratio(self, y0, y1, k, dt_signed):
        """User-supplied `norm` callable (misc.py:80-82 with a custom norm): the kernel materialises
        err/tol (padding zero-filled) and the user's own function reduces it."""
        err = self._c_err
        scaled = torch.empty_like(y0)
        self.kernels.error_scaled(self.plan, scaled, y0, y1, [k[j] for j in err.idx], err.coef, dt_signed)
        _, _, bad = self.kernels.read_norms(self.plan)
        ratio = self.norm(scaled)
        ratio = abs(float(ratio))
        return ratio, any(b != 0 for b in bad)


class Dopri5Solver(RKAdaptiveStepsizeODESolver):
    """Dormand–Prince 5(4): 6 evaluations per step, 7 stage slots (dopri5.py:33-36)."""
    order = 5
    tableau = DOPRI5


class Dopri8Solver(RKAdaptiveStepsizeODESolver):
    """Prince–Dormand 8(7): 13 evaluations per step, 14 stage slots (dopri8.py:73-76)."""
    order = 8
    tableau = DOPRI8


class Tsit5Solver(RKAdaptiveStepsizeODESolver):
    """Tsitouras 5(4): 6 evaluations per step + a 7-term solution combine (tsit5.py:79-82)."""
    order = 5
    tableau = TSIT5


class Bosh3Solver(RKAdaptiveStepsizeODESolver):
    """Bogacki–Shampine 3(2), FSAL (bosh3.py:19-22)."""
    order = 3
    tableau = BOSH3


class Fehlberg2(RKAdaptiveStepsizeODESolver):
    """Fehlberg 2(1) (fehlberg2.py:19-22)."""
    order = 2
    tableau = FEHLBERG2


class AdaptiveHeunSolver(RKAdaptiveStepsizeODESolver):
    """Heun–Euler 2(1) (adaptive_heun.py:22-25)."""
    order = 2
    tableau = ADAPTIVE_HEUN


# ---------------------------------------------------------------------------------------------------
# Fixed grid
# ---------------------------------------------------------------------------------------------------
class FixedGridODESolver(object):
    """Fixed-grid explicit RK driver (solvers.py:52-181): grid from `t`, `step_size` or `grid_constructor`;
    outputs by linear (default) or cubic Hermite interpolation between grid points.  Time-like scalars
    keep `t.dtype` (no fp64 promotion in the fixed-grid path).  Subclasses implement `_step`."""
    order: int

    def __init__(self, func: OdeFunc, y0: torch.Tensor, step_size=None, grid_constructor=None,
                 interp="linear", perturb=False, **unused_kwargs):
        self.atol = unused_kwargs.pop("atol")
        unused_kwargs.pop("rtol", None)
        unused_kwargs.pop("norm", None)
        handle_unused_kwargs(self, unused_kwargs)
        del unused_kwargs
        if not isinstance(func, OdeFunc):
            raise TypeError("solver classes of torchdiffeq_amd take the wrapped func built by check_inputs")
        self.func = func
        self.y0 = y0
        self.layout = func.layout
        self.dtype = y0.dtype
        self.device = y0.device
        self.step_size = step_size
        self.interp = interp
        self.perturb = perturb
        self.kernels = _native.get_kernels(y0.device)
        if step_size is None:
            if grid_constructor is None:
                self.grid_constructor = lambda f, y0, t: t
            else:
                self.grid_constructor = grid_constructor
        else:
            if grid_constructor is None:
                self.grid_constructor = self._grid_constructor_from_step_size(step_size)
            else:
                raise ValueError("step_size and grid_constructor are mutually exclusive arguments.")

    @classmethod
    def valid_callbacks(cls):
        return {"callback_step"}

    @staticmethod
    def _grid_constructor_from_step_size(step_size):
        def _grid_constructor(func, y0, t):
            start_time = t[0]
            end_time = t[-1]
            niters = torch.ceil((end_time - start_time) / step_size + 1).item()
            t_infer = torch.arange(0, niters, dtype=t.dtype, device=t.device) * step_size + start_time
            t_infer[-1] = t[-1]
            return t_infer
        return _grid_constructor

    # -- one step ------------------------------------------------------------------------------------
    def _step(self, t0, dt, t1, y0: torch.Tensor, y1: torch.Tensor):
        """Write y(t1) into `y1`; return f0 = func(t0, y0).  t0 and t1 are numpy scalars of the grid's dtype;
        `dt` is one too in `integrate`, and the Python float `step_size` in `integrate_until_event`."""
        raise NotImplementedError

    @staticmethod
    def _tmul(scalar, dt, c: float):
        """`dt * c` as the reference forms it: a 0-dim tensor dt times a Python float is rounded in the
        grid dtype with c rounded first; a Python-float dt (event mode, solvers.py:134) multiplies in double
        and is rounded when it meets the time tensor."""
        if isinstance(dt, float):
            return scalar(dt * c)
        return scalar(dt * scalar(c))

    def _first_perturb(self) -> Perturb:
        return Perturb.NEXT if self.perturb else Perturb.NONE

    def _last_perturb(self) -> Perturb:
        return Perturb.PREV if self.perturb else Perturb.NONE

    # -- integrate -----------------------------------------------------------------------------------
    def integrate(self, t: torch.Tensor) -> torch.Tensor:
        func, kern = self.func, self.kernels
        time_grid = self.grid_constructor(func, self.y0, t)
        assert time_grid[0] == t[0] and time_grid[-1] == t[-1]
        if self.interp not in ("linear", "cubic"):
            raise ValueError(f"Unknown interpolation method {self.interp}")
        # host copies, in the grid's own dtype (dt = t1 - t0 is formed in t.dtype: solvers.py:112)
        grid = time_grid.detach().cpu().numpy()
        tt = t.detach().cpu().numpy()
        scalar = grid.dtype.type
        linear = self.interp == "linear"

        solution = torch.empty(len(tt), self.layout.total, dtype=self.dtype, device=self.device)
        solution[0].copy_(self.y0)
        has_cb = func.callback_step is not _null
        j = 1
        y0 = self.y0
        for t0, t1 in zip(grid[:-1], grid[1:]):
            dt = scalar(t1 - t0)
            if has_cb:
                func.callback_step(torch.tensor(t0, device=self.device), y0, torch.tensor(dt, device=self.device))
            # y1 goes straight into the output row when the grid point is an output time
            if linear and j < len(tt) and t1 == tt[j]:
                y1 = solution[j]
            else:
                y1 = torch.empty_like(y0)
            f0 = self._step(t0, dt, t1, y0, y1)

            f1 = None
            while j < len(tt) and t1 >= tt[j]:
                if linear:
                    if tt[j] == t1:
                        if y1.data_ptr() != solution[j].data_ptr():
                            solution[j].copy_(y1)
                    elif tt[j] == t0:
                        solution[j].copy_(y0)
                    else:
                        slope = scalar(scalar(tt[j] - t0) / scalar(t1 - t0))
                        kern.lerp(solution[j], y0, y1, float(slope))
                else:
                    if f1 is None:
                        f1 = func.eval(t1, y1)                   # solvers.py:121, once per grid interval hit
                    self._cubic_hermite_interp(solution[j], scalar, t0, y0, f0, t1, y1, f1, tt[j])
                j += 1
            y0 = y1
        return solution

    def integrate_until_event(self, t0: torch.Tensor, event_fn):
        """Fixed steps of `step_size` until the event function changes sign, then bisection on the linear /
        cubic interpolant of that step (solvers.py:129-164).  Times are kept in the state dtype (:132)."""
        assert self.step_size is not None, \
            "Event handling for fixed step solvers currently requires `step_size` to be provided in options."
        func, kern = self.func, self.kernels
        scalar = func.np_dtype
        time_tensor = lambda v: torch.tensor(float(v), dtype=self.dtype, device=self.device)
        t0 = scalar(float(t0.detach()))
        y0 = self.y0
        dt = float(self.step_size)
        if self.interp not in ("linear", "cubic"):
            raise ValueError(f"Unknown interpolation method {self.interp}")

        sign0 = float(torch.sign(event_fn(time_tensor(t0), y0)))
        max_itrs = 20000
        itr = 0
        while True:
            itr += 1
            t1 = scalar(t0 + scalar(dt))
            y1 = torch.empty_like(y0)
            f0 = self._step(t0, dt, t1, y0, y1)
            sign1 = float(torch.sign(event_fn(time_tensor(t1), y1)))
            if sign0 != sign1:
                if self.interp == "linear":
                    def interp_fn(t, t0=t0, t1=t1, y0=y0, y1=y1):
                        if t == t0:
                            return y0
                        if t == t1:
                            return y1
                        out = torch.empty_like(y0)
                        kern.lerp(out, y0, y1, float(scalar(scalar(t - t0) / scalar(t1 - t0))))
                        return out
                else:
                    f1 = func.eval(t1, y1)

                    def interp_fn(t, t0=t0, t1=t1, y0=y0, y1=y1, f0=f0, f1=f1):
                        out = torch.empty_like(y0)
                        self._cubic_hermite_interp(out, scalar, t0, y0, f0, t1, y1, f1, t)
                        return out
                event_time, y1 = find_event(interp_fn, sign0, t0, t1, event_fn, float(self.atol), time_tensor,
                                            scalar=scalar)
                break
            else:
                t0, y0 = t1, y1
            if itr >= max_itrs:
                raise RuntimeError(f"Reached maximum number of iterations {max_itrs}.")
        solution = torch.stack([self.y0, y1], dim=0)
        return time_tensor(event_time), solution

    def _cubic_hermite_interp(self, out, scalar, t0, y0, f0, t1, y1, f1, t) -> None:
        """solvers.py:166-173; the basis values are scalars of t.dtype formed on the host."""
        one, two, three = scalar(1), scalar(2), scalar(3)
        h = scalar(scalar(t - t0) / scalar(t1 - t0))
        omh = scalar(one - h)
        h00 = scalar(scalar(scalar(one + scalar(two * h)) * omh) * omh)
        h10 = scalar(scalar(h * omh) * omh)
        hh = scalar(h * h)
        h01 = scalar(hh * scalar(three - scalar(two * h)))
        h11 = scalar(hh * scalar(h - one))
        dt = scalar(t1 - t0)
        sign = scalar(self.func.sign)       # f0 / f1 are raw func outputs: fold the time sign into their weights
        self.kernels.weighted_sum(out, [y0, f0, y1, f1],
                                  [float(h00), float(scalar(h10 * dt) * sign), float(h01), float(scalar(h11 * dt) * sign)])


class Euler(FixedGridODESolver):
    """Forward Euler (fixed_grid.py:6-11): dy = dt * f0."""
    order = 1

    def _step(self, t0, dt, t1, y0, y1):
        func = self.func
        f0 = func.eval(t0, y0, self._first_perturb())
        self.kernels.stage_combine(y1, y0, [f0], [1.0], float(dt) * func.sign)
        return f0


class Midpoint(FixedGridODESolver):
    """Explicit midpoint (fixed_grid.py:14-21): y_mid = y0 + f0*(dt/2); dy = dt * f(t0 + dt/2, y_mid)."""
    order = 2

    def _step(self, t0, dt, t1, y0, y1):
        func, kern = self.func, self.kernels
        scalar = type(t0)
        dts = float(dt) * func.sign
        half_dt = self._tmul(scalar, dt, 0.5)
        f0 = func.eval(t0, y0, self._first_perturb())
        y_mid = torch.empty_like(y0)
        kern.stage_combine(y_mid, y0, [f0], [0.5], dts)
        k2 = func.eval(scalar(t0 + half_dt), y_mid)
        kern.stage_combine(y1, y0, [k2], [1.0], dts)
        return f0


class Heun2(FixedGridODESolver):
    """Heun's 2nd-order method through the reference's rk2 step (fixed_grid.py:49-60, rk_common.py:142-157)."""
    order = 2

    def _step(self, t0, dt, t1, y0, y1):
        func, kern = self.func, self.kernels
        scalar = type(t0)
        dts = float(dt) * func.sign
        k1 = func.eval(t0, y0, self._first_perturb())
        ya = torch.empty_like(y0)
        kern.fixed_stage(1, ya, y0, [k1], [1.0], dts)
        k2 = func.eval(scalar(t0 + self._tmul(scalar, dt, 1.0)), ya, self._last_perturb())
        kern.fixed_stage(0, y1, y0, [k1, k2], [0.5, 0.5], dts)
        return k1


class Heun3(FixedGridODESolver):
    """Heun's 3rd-order method through the reference's rk3 step (fixed_grid.py:32-46, rk_common.py:121-140)."""
    order = 3

    def _step(self, t0, dt, t1, y0, y1):
        func, kern = self.func, self.kernels
        scalar = type(t0)
        dts = float(dt) * func.sign
        third, two_thirds = 1 / 3, 2 / 3
        k1 = func.eval(t0, y0, self._first_perturb())
        ya = torch.empty_like(y0)
        kern.fixed_stage(1, ya, y0, [k1], [third], dts)
        k2 = func.eval(scalar(t0 + self._tmul(scalar, dt, third)), ya)
        yb = torch.empty_like(y0)
        kern.fixed_stage(0, yb, y0, [k2], [two_thirds], dts)            # k1's weight is a structural zero
        k3 = func.eval(scalar(t0 + self._tmul(scalar, dt, two_thirds)), yb)
        kern.fixed_stage(0, y1, y0, [k1, k3], [1 / 4, 3 / 4], dts)      # k2's weight is a structural zero
        return k1


class RK4(FixedGridODESolver):
    """Fixed-grid 4th-order RK, 3/8 rule (fixed_grid.py:24-29 -> rk_common.py:110-118)."""
    order = 4

    def _step(self, t0, dt, t1, y0, y1):
        func, kern = self.func, self.kernels
        scalar = type(t0)
        third, two_thirds = 1 / 3, 2 / 3
        dts = float(dt) * func.sign
        ts = func.time_tensors(kern, [(t0, self._first_perturb()),
                                      (scalar(t0 + self._tmul(scalar, dt, third)), Perturb.NONE),
                                      (scalar(t0 + self._tmul(scalar, dt, two_thirds)), Perturb.NONE),
                                      (t1, self._last_perturb())])
        k1 = func.eval_at(ts[0], y0)
        ya = torch.empty_like(y0)
        kern.rk4_stage(1, ya, y0, k1, None, None, None, dts)
        k2 = func.eval_at(ts[1], ya)
        yb = torch.empty_like(y0)
        kern.rk4_stage(2, yb, y0, k1, k2, None, None, dts)
        k3 = func.eval_at(ts[2], yb)
        yc = torch.empty_like(y0)
        kern.rk4_stage(3, yc, y0, k1, k2, k3, None, dts)
        k4 = func.eval_at(ts[3], yc)
        kern.rk4_stage(4, y1, y0, k1, k2, k3, k4, dts)
        return k1


SOLVER_CLASSES = {"dopri8": Dopri8Solver, "dopri5": Dopri5Solver, "tsit5": Tsit5Solver, "bosh3": Bosh3Solver,
                  "fehlberg2": Fehlberg2, "adaptive_heun": AdaptiveHeunSolver, "euler": Euler,
                  "midpoint": Midpoint, "heun2": Heun2, "heun3": Heun3, "rk4": RK4}
