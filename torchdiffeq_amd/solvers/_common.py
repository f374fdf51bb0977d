"""Host scalars of the step controller (misc.py:85-95 as doubles) and the autograd shadows of a fixed-grid step — shared by
the solver modules of this package."""
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


def _norm_value(x) -> float:
    """|value| of what a user's `norm` callable returned.  More than one element is the reference's error: its next
    statement compares the result (`d0 < 1e-5`, `error_ratio <= 1`: misc.py:60, rk_common.py:303)."""
    if isinstance(x, torch.Tensor) and x.numel() != 1:
        raise RuntimeError("Boolean value of Tensor with more than one value is ambiguous (the `norm` callable must "
                           "return a scalar)")
    return abs(float(x))


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


@np.errstate(all="ignore")     # host scalars follow IEEE silently, as 0-dim tensors do
def optimal_step_size_in(W, last_step, error_ratio, safety, ifactor, dfactor, order) -> float:
    """The same controller with every operation rounded in the host scalar type W (misc.py:85-95 on 0-dim tensors of
    the solver option `dtype`, rk_common.py:176-194) — for W other than fp64."""
    with np.errstate(all="ignore"):
        last_step, ratio = W(last_step), W(error_ratio)
        if ratio == 0:
            return float(last_step * W(ifactor))
        floor = W(1.0) if ratio < 1 else W(dfactor)
        exponent = W(1.0) / W(order)                  # torch.tensor(order, dtype).reciprocal()
        scaled = W(safety) / ratio ** exponent
        factor = _nan_min(W(ifactor), _nan_max(scaled, floor))
        return float(last_step * factor)


class _StepShadow:
    """Autograd shadows of one fixed-grid step's time scalars (solver time): t0, t1 are entries of the
    time grid tensor (whose graph leads back to `t`); host scalars give the values, these only the gradient."""
    __slots__ = ("t0", "t1", "sign")

    def __init__(self, t0, t1, sign):
        self.t0, self.t1, self.sign = t0, t1, sign

    def width(self):
        """dt in solver time."""
        return self.t1 - self.t0

    def dt_signed(self):
        """The scalar handed to the kernels as `dt` (time sign folded in)."""
        return (self.t1 - self.t0) * self.sign

    def time(self, c: float):
        """User time of the stage at t0 + c dt."""
        return (self.t0 + (self.t1 - self.t0) * c) * self.sign

    def fraction(self, t_shadow, t_const=None):
        """(t - t0) / (t1 - t0) for an output time t (`t_const`: the value of a time that is not in the graph — it
        matters as soon as the step width itself carries a gradient)."""
        if t_shadow is not None:
            num = t_shadow - self.t0
        else:
            num = -self.t0 if t_const is None else float(t_const) - self.t0
        return num / (self.t1 - self.t0)


class _NoShadow:
    def width(self):
        return None

    def dt_signed(self):
        return None

    def time(self, c):
        return None

    def fraction(self, t_shadow, t_const=None):
        return None


_NO_SHADOW = _NoShadow()
