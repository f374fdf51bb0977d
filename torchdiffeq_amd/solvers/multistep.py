"""Adams–Bashforth(–Moulton) on a fixed grid (torchdiffeq/_impl/fixed_adams.py:164-228).  SURVEY.md §2 marks the multistep
methods out of scope for the hot path: frozen since r03, kept so that `method='explicit_adams' / 'implicit_adams'` callers
find them."""
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
from .adaptive import _LockStep
from .fixed import FixedGridODESolver, _rk4_38_step


_ADAMS_MIN_ORDER = 4
_ADAMS_MAX_ORDER = 12
_ADAMS_MAX_ITERS = 4


class AdamsBashforthMoulton(FixedGridODESolver):
    """`implicit_adams` / `fixed_adams` (fixed_adams.py:164-223): variable-order (up to `max_order`) Adams–Bashforth
    predictor and, with `implicit=True`, an Adams–Moulton corrector solved by at most `max_iters` fixed-point
    iterations; the first steps — until three past derivatives exist — are 3/8-rule RK4 steps.

    The history `prev_f` is a deque of SEPARATE contiguous func outputs (newest first), read once per step by
    tdeq_adams_predict (predictor sum, the corrector's constant part and y0 + dy in one pass: order+1 reads, 1 or 3
    writes); each corrector iteration is ONE tdeq_adams_correct launch (new dy, next evaluation point and the
    convergence census of `_has_converged`), with one polled read-back per iteration — the reference spends ~2·order
    + 12 eager ops and a host sync there.  The method's quirks are kept: the corrected derivative never replaces
    the predictor's in the history (`_update_history(t0, f)` finds `prev_t == t0`, :222), and a step whose iteration
    did not converge warns and drops the OLDEST derivative (:219-221)."""
    order = 4

    def __init__(self, func, y0, rtol=1e-3, atol=1e-4, implicit=True, max_iters=_ADAMS_MAX_ITERS,
                 max_order=_ADAMS_MAX_ORDER, dist_sync=None, **kwargs):
        super().__init__(func, y0, rtol=rtol, atol=atol, **kwargs)
        self.max_order = self._checked_max_order(max_order)
        self.implicit, self.max_iters = implicit, max_iters
        self.rtol, self.atol = rtol, atol           # the corrector's convergence test (`_converged`)
        # past derivatives, newest first, and the time the newest one belongs to (`_update_history`)
        self.prev_t, self.prev_f = None, collections.deque(maxlen=self.max_order - 1)
        self._sync = _LockStep(dist_sync) if dist_sync is not None else None
        self._plan = None
        # A 0-dim fp32 state meets the reference's fp64 coefficient tensors as 0-dim x 0-dim, which PyTorch promotes
        # to fp64 (a dimensioned fp32 tensor would stay fp32): products and sums of `_dot_product` run in fp64 and are
        # rounded once by `.type_as(y0)` — see _step_zero_dim.
        self._zero_dim_f32 = (not self.layout.is_tuple and tuple(self.layout.shapes[0]) == ()
                              and y0.dtype in (torch.float32, torch.complex64))        # complex64 promotes to complex128 alike

    @staticmethod
    def _checked_max_order(max_order) -> int:
        """The option's two documented reactions (fixed_adams.py:170-172): orders beyond the coefficient table are
        refused, orders below the multistep minimum only ever take the RK4 start-up steps."""
        assert max_order <= _ADAMS_MAX_ORDER, "max_order must be at most {}".format(_ADAMS_MAX_ORDER)
        if max_order < _ADAMS_MIN_ORDER:
            warnings.warn("max_order is below {}, so the solver reduces to `rk4`.".format(_ADAMS_MIN_ORDER))
        return int(max_order)

    def _step_zero_dim(self, t1, y0, f0, hist, order, dt64, sh):
        """The step for a 0-dim fp32 state with the reference's type promotion (fixed_adams.py:205-216): the history
        dot products and `dt * m0 * f` are formed in fp64 (0-dim fp64 coefficient x 0-dim fp32 derivative promotes) and
        rounded to fp32 once.  Same kernels, on fp64 copies of the one-element tensors; through `ops`, so the step is
        recorded for autograd when something requires grad (func's parameters, y0, t)."""
        func, ops = self.func, self.ops
        sign = func.sign
        dsh = sh.dt_signed()
        wrt_dt = lambda dw: [(dsh, list(dw))] if dsh is not None else []
        bash, _ = adams_coefficients(order)
        low, wide = y0.dtype, (torch.float64 if y0.dtype == torch.float32 else torch.complex128)
        h64 = [h.to(wide) for h in hist]
        dot64 = lambda coefs, sc=(): ops._long_sum(h64, list(coefs), list(sc)).to(low)     # left to right in fp64, one rounding
        add = lambda a, b: ops.weighted_sum([a, b], [1.0, 1.0])

        dy = dot64([dt64 * b * sign for b in bash], wrt_dt(bash))
        y = add(y0, dy)
        if not self.implicit:
            return y, f0
        _, moulton = adams_coefficients(order + 1)
        delta = ops.weighted_sum([dot64(moulton[1:])], [dt64 * sign], wrt_dt([1.0]))
        if self._plan is None:
            self._plan = self.kernels.make_plan(self.layout.segments(self.rtol, self.atol), self.layout.total,
                                                self.layout.chunk, self.device)
        c = dt64 * moulton[0] * sign
        last = self._last_perturb()
        converged = False
        for _ in range(self.max_iters):
            f = func.eval(t1, y, last, shadow=sh.time(1.0))
            dy_new = add(ops.weighted_sum([f.to(wide)], [c], wrt_dt([moulton[0]])).to(low), delta)
            y = add(y0, dy_new)
            self.kernels.adams_correct(self._plan, dy_new.detach(), dy.detach(), compute=False)
            dy = dy_new
            converged = self._converged()
            if converged:
                break
        if not converged:
            warnings.warn("Functional iteration did not converge. Solution may be incorrect.")
            self.prev_f.pop()
        return y, f0

    def _update_history(self, t, f) -> None:
        if self.prev_t is None or self.prev_t != t:
            self.prev_f.appendleft(f)
            self.prev_t = t

    def _converged(self) -> bool:
        """`_has_converged` (fixed_adams.py:189-192) from the census of the last tdeq_adams_correct launch."""
        counts, _, _ = self.kernels.read_norms(self._plan)
        if self._sync is not None:      # sharded batch in lock step: every rank iterates until all have converged
            counts = self._sync._allreduce(list(counts), self.device)
        return not any(c != 0.0 for c in counts)

    def _step(self, t0, dt, t1, y0, y1_out, sh):
        func, ops = self.func, self.ops
        f0 = func.eval(t0, y0, self._first_perturb(), shadow=sh.time(0.0))
        self._update_history(t0, f0)
        order = min(len(self.prev_f), self.max_order - 1)
        if order < _ADAMS_MIN_ORDER - 1:
            y1, _ = _rk4_38_step(self, t0, dt, t1, y0, self.prev_f[0], y1_out, sh)
            return y1, f0
        sign = func.sign
        dt64 = float(dt)
        bash, _ = adams_coefficients(order)
        hist = [self.prev_f[j] for j in range(order)]
        if self._zero_dim_f32:
            return self._step_zero_dim(t1, y0, f0, hist, order, dt64, sh)
        cb = [dt64 * b * sign for b in bash]            # `dt * bashforth_coeffs` in fp64 (:205); the sign is exact
        dsh = sh.dt_signed()
        if not self.implicit:
            y1, _, _ = ops.adams_predict(y0, hist, cb, None, 0.0, dsh, list(bash), out=y1_out)
            return y1, f0
        _, moulton = adams_coefficients(order + 1)
        y, dy, delta = ops.adams_predict(y0, hist, cb, list(moulton[1:]), dt64 * sign, dsh, list(bash))
        if self._plan is None:
            self._plan = self.kernels.make_plan(self.layout.segments(self.rtol, self.atol), self.layout.total,
                                                self.layout.chunk, self.device)
        c = dt64 * moulton[0] * sign                      # `dt * moulton_coeffs[0]`: 0-dim fp32/fp64 x fp64 -> fp64 (:214)
        last = self._last_perturb()
        converged = False
        for _ in range(self.max_iters):
            f = func.eval(t1, y, last, shadow=sh.time(1.0))
            y, dy = ops.adams_correct(self._plan, y0, f, delta, dy, c, dsh, moulton[0])
            converged = self._converged()
            if converged:
                break
        if not converged:
            warnings.warn("Functional iteration did not converge. Solution may be incorrect.")
            self.prev_f.pop()
        self._update_history(t0, f)
        return y, f0


class AdamsBashforth(AdamsBashforthMoulton):
    """`explicit_adams` (fixed_adams.py:226-228)."""

    def __init__(self, func, y0, **kwargs):
        super().__init__(func, y0, implicit=False, **kwargs)
