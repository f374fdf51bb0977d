"""Event mode (torchdiffeq/_impl/rk_common.py:252-264, solvers.py:44-49, 130-166, event_handling.py:5-20) as mixins of the
two solver families: step until `event_fn(t, y)` changes sign, then bisect on the dense output of the last step."""
from __future__ import annotations

import bisect  # noqa: F401
import collections  # noqa: F401
import math  # noqa: F401
import os  # noqa: F401
import warnings  # noqa: F401
from typing import List, Optional, Sequence  # noqa: F401

import numpy as np  # noqa: F401
import torch

from .. import _native
# captured trial steps, their cache and the "auto" policy live in _graph.py; the size limits and step thresholds are READ
# here (tools patch `solvers._GRAPH_MODE_MAX_ELEMENTS` to measure beyond the shipped limit)
from .._graph import (_AUTO_CAPTURE_AFTER_STEPS, _AUTO_MIN_GRID_STEPS, _GRAPH_AUTO_MAX_ELEMENTS,  # noqa: F401
                     _GRAPH_MODE_MAX_ELEMENTS, _CaptureFailed, _DtCell, _GraphStep, _capture, _graph_request,
                     _held_tensor_ptrs, _reusable_across_solves, _scalar_state, _side_effect_fingerprint, _side_stream,
                     _request_is_explicit, _stream_is_capturing, clear_graph_cache)
from .._scalars import is_low, power, rdiv, scalar_type  # noqa: F401
from ..autodiff import Ops, stitch  # noqa: F401
from ..misc import (BuiltinNorm, OdeFunc, Perturb, StateLayout, component_norm, find_event, handle_unused_kwargs, rms_norm,  # noqa: F401
                   vector_tolerances)
from ..misc import _null_callback as _null
from ..tableaus import (ADAPTIVE_HEUN, ADAPTIVE_TABLEAUS, BOSH3, CARRY_DEFAULT_ON, DOPRI5, DOPRI8, FEHLBERG2, TSIT5, SparseRow, Tableau,  # noqa: F401
                       adams_coefficients, carry_plan)
from ._common import _nan_max, _nan_min, _clamp, _norm_value, _as_float, optimal_step_size, optimal_step_size_in, _StepShadow, _NoShadow, _NO_SHADOW  # noqa: F401


class AdaptiveEvents:
    """`integrate_until_event` of the adaptive Runge–Kutta solvers (mixed into RKAdaptiveStepsizeODESolver)."""

    @_native.on_state_device
    def integrate_until_event(self, t0: torch.Tensor, event_fn):
        """(event_t, solution[2, total]): step until `event_fn(t, y)` changes sign, then bisect on the last
        step's dense output (solvers.py:44-49, rk_common.py:252-264, event_handling.py:5-20)."""
        self._set_time_anchor(t0.reshape(-1))
        self._before_integrate([float(t0.detach().to(self.dtype))])
        event_time, y1 = self._advance_until_event(event_fn)
        solution = torch.stack([self.y0, y1], dim=0)
        return self._time_tensor(float(event_time)), solution

    def _advance_until_event(self, event_fn):
        ev = lambda: event_fn(self._time_tensor(self.t1), self.y1)
        if ev() == 0:
            return self.t1, self.y1
        n_steps = 0
        sign0 = float(torch.sign(ev()).detach())
        while sign0 == float(torch.sign(ev()).detach()):
            assert n_steps < self.max_num_steps, \
                "max_num_steps exceeded ({}>={})".format(n_steps, self.max_num_steps)
            self._adaptive_step()
            n_steps += 1

        def interp_fn(t):
            return self._interp_evaluate(float(t))

        atol = self.atol
        if isinstance(atol, torch.Tensor):
            atol = atol.min().item()
        elif not isinstance(atol, (int, float)):
            atol = min(float(a) for a in atol)
        return find_event(interp_fn, sign0, self.t0, self.t1, event_fn, self._w(float(atol)), self._time_tensor,
                          scalar=self._W)


class FixedGridEvents:
    """`integrate_until_event` of the fixed-grid solvers (mixed into FixedGridODESolver)."""

    @_native.on_state_device
    def integrate_until_event(self, t0: torch.Tensor, event_fn):
        """Fixed steps of `step_size` until the event function changes sign, then bisection on the linear /
        cubic interpolant of that step (solvers.py:129-164).  Times are kept in the state dtype (:132).
        When the start time requires grad its gradient is carried by step shadows, as in `integrate`: the reference
        forms `t1 = t0 + dt` and the interpolation fraction `(t - t0) / (t1 - t0)` on the tensor `t0` itself, so the
        state at the (detached) event time depends on it — which is what `odeint_event` turns into d(event time)/d t0."""
        assert self.step_size is not None, \
            "Event handling for fixed step solvers currently requires `step_size` to be provided in options."
        func, ops = self.func, self.ops
        scalar = func.np_dtype
        time_tensor = lambda v: torch.tensor(float(v), dtype=func.time_dtype, device=self.device)     # solvers.py:132
        start = t0 if (torch.is_grad_enabled() and torch.is_tensor(t0) and t0.requires_grad) else None
        t0 = scalar(float(t0.detach()))
        t_first = float(t0)
        y0 = self.y0
        dt = float(self.step_size)
        if self.interp not in ("linear", "cubic"):
            raise ValueError(f"Unknown interpolation method {self.interp}")

        def shadow(ta, tb):
            if start is None:
                return _NO_SHADOW
            return _StepShadow(start + (float(ta) - t_first), start + (float(tb) - t_first), func.sign)

        sign0 = float(torch.sign(event_fn(time_tensor(t0), y0)).detach())
        step_budget = 20000
        for _ in range(step_budget):
            t1 = scalar(t0 + scalar(dt))
            sh = shadow(t0, t1)
            y1, f0 = self._step(t0, dt, t1, y0, None, sh)
            sign1 = float(torch.sign(event_fn(time_tensor(t1), y1)).detach())
            if sign0 != sign1:
                if self.interp == "linear":
                    def interp_fn(t, t0=t0, t1=t1, y0=y0, y1=y1, sh=sh):
                        if t == t0:
                            return y0
                        if t == t1:
                            return y1
                        return ops.lerp(y0, y1, float(scalar(scalar(t - t0) / scalar(t1 - t0))), sh.fraction(None, t))
                else:
                    f1 = func.eval(t1, y1, shadow=sh.time(1.0))

                    def interp_fn(t, t0=t0, t1=t1, y0=y0, y1=y1, f0=f0, f1=f1, sh=sh):
                        return self._cubic_hermite_interp(scalar, t0, y0, f0, t1, y1, f1, t, sh, None)
                event_time, y1 = find_event(interp_fn, sign0, t0, t1, event_fn, float(self.atol), time_tensor,
                                            scalar=scalar)
                return time_tensor(event_time), torch.stack([self.y0, y1], dim=0)
            t0, y0 = t1, y1
        raise RuntimeError(f"Reached maximum number of iterations {step_budget}.")

    def _cubic_hermite_interp(self, scalar, t0, y0, f0, t1, y1, f1, t, sh, t_shadow, out=None) -> torch.Tensor:
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
        ws = [float(h00), float(scalar(h10 * dt) * sign), float(h01), float(scalar(h11 * dt) * sign)]
        scalars, w_fn = (), None
        if sh is not _NO_SHADOW:
            hf, dtf, sg = float(h), float(dt), float(sign)
            d_h = [-6 * hf * (1 - hf), (1 - hf) * (1 - 3 * hf) * dtf * sg, 6 * hf * (1 - hf),
                   (3 * hf * hf - 2 * hf) * dtf * sg]
            d_dt = [0.0, float(h10) * sg, 0.0, float(h11) * sg]
            scalars = [(sh.fraction(t_shadow, t), d_h), (sh.width(), d_dt)]
            device = y0.device

            def w_fn(live):
                """The basis as torch expressions of (h, dt) — cubic in h, so second-order time gradients need its
                curvature (values from the host scalars, gradients through the shadows)."""
                h_s, dt_s = live
                h_t = torch.full((), hf, dtype=torch.float64, device=device)
                dt_t = torch.full((), dtf, dtype=torch.float64, device=device)
                if h_s is not None:
                    h_t = h_t + (h_s - h_s.detach()).double()
                if dt_s is not None:
                    dt_t = dt_t + (dt_s - dt_s.detach()).double()
                omh_t = 1 - h_t
                b00, b10 = (1 + 2 * h_t) * omh_t * omh_t, h_t * omh_t * omh_t
                b01, b11 = h_t * h_t * (3 - 2 * h_t), h_t * h_t * (h_t - 1)
                d00, d10 = -6 * h_t * omh_t, omh_t * (1 - 3 * h_t)
                d01, d11 = 6 * h_t * omh_t, 3 * h_t * h_t - 2 * h_t
                zero = torch.zeros((), dtype=torch.float64, device=device)
                return ([b00, b10 * dt_t * sg, b01, b11 * dt_t * sg],
                        [[d00, d10 * dt_t * sg, d01, d11 * dt_t * sg], [zero, b10 * sg, zero, b11 * sg]])
        return self.ops.weighted_sum([y0, f0, y1, f1], ws, scalars, out=out, w_fn=w_fn)
