"""Fixed-grid solvers: the output loop with linear interpolation, euler / midpoint / heun2 / heun3 / rk4 steps, captured grid
steps (torchdiffeq/_impl/solvers.py:52-181, fixed_grid.py, rk_common.py:94-157)."""
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
from .events import FixedGridEvents


def _host_times(times: torch.Tensor):
    """A time tensor as host scalars that round like its 0-dim elements (`_scalars`): a numpy array for fp32 / fp64
    (its scalar type does IEEE arithmetic in the type itself), a list of `BFloat16Scalar` / `Float16Scalar` for the
    16-bit types numpy cannot hold or would not round like ATen.  Returns (sequence, scalar type)."""
    low = scalar_type(times.dtype) if times.dtype in (torch.bfloat16, torch.float16) else None
    if low is None:
        host = times.detach().cpu().numpy()
        return host, host.dtype.type
    return [low(v) for v in times.detach().float().cpu().tolist()], low


def _uniform_grid(t: torch.Tensor, step_size) -> torch.Tensor:
    """Points t[0] + i·step_size covering [t[0], t[-1]], the last one moved onto t[-1] exactly.  Formed with tensor
    arithmetic in t.dtype on t.device — the point count ceil(span / step_size + 1) and every grid value must round as
    the reference's do (solvers.py:86-96; on a ROCm device a tensor divided by a host scalar is a multiplication by
    its reciprocal, which host arithmetic would not reproduce), and the grid keeps the autograd graph of `t`."""
    first, last = t[0], t[-1]
    count = float(torch.ceil((last - first) / step_size + 1).detach())
    if not math.isfinite(count):
        torch.arange(0, count)          # step_size 0 / nan: torch's own RuntimeError ("unsupported range: 0 -> inf")
    count = int(count)
    grid = torch.arange(count, dtype=t.dtype, device=t.device) * step_size + first
    grid[-1] = last
    return grid


class FixedGridODESolver(FixedGridEvents):
    """Fixed-grid explicit RK driver (solvers.py:52-181): grid from `t`, `step_size` or `grid_constructor`;
    outputs by linear (default) or cubic Hermite interpolation between grid points.  Time-like scalars
    keep `t.dtype` (no fp64 promotion in the fixed-grid path).  Subclasses implement `_step`."""
    order: int
    flat_state_native = True

    def __init__(self, func: OdeFunc, y0: torch.Tensor, step_size=None, grid_constructor=None,
                 interp="linear", perturb=False, hip_graph=None, **unused_kwargs):
        self.atol = unused_kwargs.pop("atol")
        # `hip_graph=True` (an extension, not a reference option): replay one captured hipGraph per grid interval
        # instead of launching a step's kernels one by one — see RK4._integrate_graph.  "auto": where it applies
        # (rk4, small states), without the warning otherwise.
        self.hip_graph, self._graph_auto = _graph_request(hip_graph)
        self._graph_explicit = hip_graph is not None
        self._graph_warn = _request_is_explicit(hip_graph)       # refusals warn only where captured steps were asked for
        unused_kwargs.pop("rtol", None)
        unused_kwargs.pop("norm", None)
        unused_kwargs.pop("dist_sync", None)          # fixed grids are in lock step by construction
        unused_kwargs.pop("dist_replicated", None)
        handle_unused_kwargs(self, unused_kwargs)
        del unused_kwargs
        if not isinstance(func, OdeFunc):
            raise TypeError("solver classes of torchdiffeq_amd take the wrapped func built by check_inputs")
        if step_size is not None and grid_constructor is not None:
            raise ValueError("step_size and grid_constructor are mutually exclusive arguments.")
        self.func, self.y0, self.layout = func, y0, func.layout
        self.dtype, self.device = y0.dtype, y0.device
        self.kernels = _native.get_kernels(y0.device, y0.dtype)
        self.ops = Ops(self.kernels, func.np_dtype)
        self.interp, self.perturb = interp, perturb
        # where the grid comes from: a user callable, a uniform spacing, or — neither given — the output times
        self.step_size = step_size
        self._user_grid = grid_constructor

    @classmethod
    def valid_callbacks(cls):
        return {"callback_step"}

    def _time_grid(self, t: torch.Tensor) -> torch.Tensor:
        """The integration grid for output times `t` (solvers.py:70-96): the user's `grid_constructor(func, y0, t)`,
        else the uniform `step_size` grid, else `t` itself (the same tensor object: that is what graph mode tests)."""
        if self._user_grid is not None:
            return self._user_grid(*self._reference_view(), t)
        if self.step_size is None:
            return t
        return _uniform_grid(t, self.step_size)

    def _reference_view(self):
        """(func, y0) as the reference hands them to a user's `grid_constructor(func, y0, t)` (solvers.py:103): a tensor
        state in ITS shape, a tuple state — also the adjoint's augmented one — as the plain concatenation of its
        components (misc.py:206-209), not this package's chunk-padded flat buffer; `func(t, y)` maps such a state to its
        derivative in the same form (t in solver time, like every call of the reference's wrapped func)."""
        lay, func = self.layout, self.func
        if not lay.is_tuple:
            shape = lay.shapes[0]
            return (lambda t, y, **kw: func(t, y.reshape(-1), **kw).view(shape)), self.y0.view(shape)

        def joined(flat):
            return torch.cat([c.reshape(-1) for c in lay.unpack(flat)]) if lay.n_seg else flat

        def on_joined(t, y, **kw):
            parts, off = [], 0
            for n, shape in zip(lay.numels, lay.shapes):
                parts.append(y[off:off + n].view(shape))
                off += n
            return joined(func(t, lay.pack(parts, dtype=self.dtype), **kw))
        return on_joined, joined(self.y0)

    # -- one step ------------------------------------------------------------------------------------
    def _step(self, t0, dt, t1, y0: torch.Tensor, y1_out: Optional[torch.Tensor], sh: "_StepShadow"):
        """Return (y(t1), f0 = func(t0, y0)); y(t1) is written into `y1_out` when given (no-grad callers).
        t0 and t1 are numpy scalars of the grid's dtype; `dt` is one too in `integrate`, and the Python float
        `step_size` in `integrate_until_event`.  `sh` carries the autograd shadows of t0 / dt when the grid
        requires grad."""
        raise NotImplementedError

    @staticmethod
    def _tmul(scalar, dt, c: float):
        """`dt * c` as the reference forms it: a 0-dim tensor dt times a Python float is rounded in the
        grid dtype with c rounded first; a Python-float dt (event mode, solvers.py:134) multiplies in double
        and is rounded when it meets the time tensor."""
        if is_low(type(dt)):
            return dt * c                   # a 16-bit 0-dim tensor times a Python number: the number at fp32, one rounding
        if isinstance(dt, float):
            return scalar(dt * c)
        return scalar(dt * scalar(c))

    def _first_perturb(self) -> Perturb:
        return Perturb.NEXT if self.perturb else Perturb.NONE

    def _last_perturb(self) -> Perturb:
        return Perturb.PREV if self.perturb else Perturb.NONE

    # -- integrate -----------------------------------------------------------------------------------
    @_native.on_state_device
    def integrate(self, t: torch.Tensor) -> torch.Tensor:
        func, ops = self.func, self.ops
        time_grid = self._time_grid(t)
        assert time_grid[0] == t[0] and time_grid[-1] == t[-1]
        if self.interp not in ("linear", "cubic"):
            raise ValueError(f"Unknown interpolation method {self.interp}")
        if self.hip_graph:
            if self._graph_capable(t, time_grid) and not (self._graph_auto and
                                                          self.layout.total > _GRAPH_AUTO_MAX_ELEMENTS):
                solution = self._integrate_graph(t)
                if solution is not None:
                    return solution
                # (None: the first evaluations showed func's outputs to be part of an autograd graph the static look at
                #  func had not found — the solve starts over on the eager, differentiable path below)
            elif not self._graph_auto and self._graph_explicit:
                # (asked for by option; a process-wide TDEQ_HIP_GRAPH=1 default applies where it can and stays silent)
                warnings.warn("{}: hip_graph=True needs an explicit Runge-Kutta fixed-grid method (euler, midpoint, "
                              "heun2, heun3, rk4), the output times as the grid, linear interpolation, no callback, no "
                              "autograd graph and a ROCm device; running the eager path".format(self.__class__.__name__))
        # host copies, in the grid's own dtype (dt = t1 - t0 is formed in t.dtype: solvers.py:112)
        grid, scalar = _host_times(time_grid)
        tt, _ = _host_times(t)
        linear = self.interp == "linear"
        grad_mode = torch.is_grad_enabled()
        time_grad = grad_mode and (time_grid.requires_grad or t.requires_grad)
        sign = func.sign

        rows: List[Optional[torch.Tensor]] = [self.y0] + [None] * (len(tt) - 1)
        solution = None
        has_cb = func.callback_step is not _null
        j = 1
        y0 = self.y0
        for n, (t0, t1) in enumerate(zip(grid[:-1], grid[1:])):
            dt = scalar(t1 - t0)
            if has_cb:
                func.callback_step(torch.tensor(t0, dtype=time_grid.dtype, device=self.device), y0,
                                   torch.tensor(dt, dtype=time_grid.dtype, device=self.device))
            sh = _StepShadow(time_grid[n], time_grid[n + 1], sign) if time_grad else _NO_SHADOW
            # Without a graph, y1 goes straight into the output row when the grid point is an output time.
            differentiable = grad_mode and (time_grad or y0.requires_grad)
            y1_out = None
            if not differentiable:
                if solution is None:
                    solution = torch.empty(len(tt), self.layout.total, dtype=self.dtype, device=self.device)
                if linear and j < len(tt) and t1 == tt[j]:
                    y1_out = solution[j]
            y1, f0 = self._step(t0, dt, t1, y0, y1_out, sh)
            differentiable = differentiable or (grad_mode and y1.requires_grad)

            while j < len(tt) and t1 >= tt[j]:
                tj_shadow = t[j] if time_grad else None
                if linear:
                    if tt[j] == t1:
                        rows[j] = y1
                    elif tt[j] == t0:
                        rows[j] = y0
                    else:
                        slope = scalar(scalar(tt[j] - t0) / scalar(t1 - t0))
                        rows[j] = ops.lerp(y0, y1, float(slope), sh.fraction(tj_shadow),
                                           out=None if differentiable or solution is None else solution[j])
                else:
                    # solvers.py:121: evaluated anew for EVERY output time inside the step — a counting or stateful
                    # func sees the reference's calls, and each output row hangs on its own graph node
                    f1 = func.eval(t1, y1, shadow=sh.time(1.0))
                    rows[j] = self._cubic_hermite_interp(scalar, t0, y0, f0, t1, y1, f1, tt[j], sh, tj_shadow,
                                                         out=None if differentiable or solution is None
                                                         else solution[j])
                j += 1
            y0 = y1
        if any(r.requires_grad for r in rows) and grad_mode:
            return torch.stack(rows, dim=0)
        if solution is None:
            solution = torch.empty(len(tt), self.layout.total, dtype=self.dtype, device=self.device)
        for i, r in enumerate(rows):
            if r.data_ptr() != solution[i].data_ptr():
                solution[i].copy_(r)
        return solution

    # -- hipGraph mode ----------------------------------------------------------------------------------------
    _graph_times = None          # per method: ((fraction of dt, mode bits), ...) of its stage times — see _integrate_graph

    def _graph_step(self, ts, y_cur, dt_dev, ctrl):
        """The method's step on device-resident step data: evaluations at the 0-dim tensors `ts`, stage kernels that
        read the step size from `dt_dev` (`ctrl.ctrl_dev[1]`); returns y(t1)."""
        raise NotImplementedError

    def _graph_capable(self, t: torch.Tensor, time_grid: torch.Tensor) -> bool:
        if not (self._graph_times is not None and time_grid is t and self.interp == "linear"
                and self.func.callback_step is _null and self.device.type == "cuda"
                and hasattr(self.kernels, "grid_advance_stages")):
            return False
        if _stream_is_capturing():
            return False            # the caller is capturing a graph of its own around this solve: no nested capture
        if not torch.is_grad_enabled():
            return True
        if t.requires_grad or self.y0.requires_grad:
            return False
        # grad mode with neither y0 nor t in the graph: func's own parameters may still be (plain `odeint` training).
        # The replayed kernels write into raw buffers — a solution without an autograd graph — so such a solve has to
        # take the eager path.  Decided from what func HOLDS (module parameters / buffers, closure cells, globals its body
        # names, attributes of a callable object: `_graph._held_tensors`), not by evaluating it: a probe evaluation would
        # be visible to the user (RNG / dropout state, counters, one more evaluation) and would look at t[0] only.
        from .._graph import _holds_a_tensor_that_requires_grad
        return not _holds_a_tensor_that_requires_grad(self.func.base_func)

    def _integrate_graph(self, t: torch.Tensor) -> torch.Tensor:
        """`integrate` for small states, where a step costs launch latency, not bandwidth: ONE hipGraph — the method's
        evaluations of `func`, its stage kernels with the step size read from device memory, tdeq_grid_commit (y1 ->
        output row and next state) and tdeq_grid_advance_stages (next step's dt and stage times, formed on the device
        with the host's rounding sequence) — is captured once and replayed per grid interval.  Same kernels and
        operation order as the eager path, so the solution is bit-identical.  euler, midpoint, heun2, heun3, rk4 (r03:
        the method is data — `_graph_times` — plus its `_graph_step`).  `func` must be capturable (static shapes, no
        host synchronisation, no Python side effects it relies on: it runs only for the first step and once more
        during capture)."""
        func, kern = self.func, self.kernels
        n_t = len(t)
        solution = torch.empty(n_t, self.layout.total, dtype=self.dtype, device=self.device)
        solution[0].copy_(self.y0)
        if n_t == 1:
            return solution
        grid = t.detach().contiguous()
        y_cur = self.y0.clone()
        counter = torch.full((), -1, dtype=torch.int64, device=self.device)
        fracs, modes = [f for f, _ in self._graph_times], [m for _, m in self._graph_times]
        n_eval = len(fracs)
        times = torch.empty(n_eval, dtype=func.time_dtype, device=self.device)
        # {unused, sign * dt}: the layout tdeq_stage_combine_dev reads its step size from (a norm plan's ctrl_dev)
        ctrl = _DtCell(torch.zeros(2, dtype=torch.float64, device=self.device))
        dt_dev = ctrl.ctrl_dev[1:]
        kern.grid_advance_stages(grid, counter, self.perturb, func.sign, fracs, modes, times, dt_dev)      # step 0
        ts = times.unbind(0)

        def step():
            y1 = self._graph_step(ts, y_cur, dt_dev, ctrl)
            kern.grid_commit(solution, y_cur, y1, counter)
            kern.grid_advance_stages(grid, counter, self.perturb, func.sign, fracs, modes, times, dt_dev)

        # the first step runs eagerly on a side stream (library / allocator warm-up before capture) ...
        current = torch.cuda.current_stream(self.device)
        side = _side_stream(self.device)
        side.wait_stream(current)
        auto = self._graph_auto
        before = _side_effect_fingerprint(func.base_func, self.device) if auto else None
        func.grad_output_seen = False
        nfe_first = func.nfe
        with torch.cuda.stream(side):
            step()
        current.wait_stream(side)
        if func.grad_output_seen and torch.is_grad_enabled():
            # Dynamic guard (advisor r05): func's output requires grad although nothing `_held_tensors` could see does — a
            # parameter behind a property, a C-extension object, a weak reference ...  The kernels of this path write raw
            # buffers, the solution would carry no graph and the parameter gradients would be lost silently.  The step
            # just taken is discarded; `integrate` runs the eager path, which records one node per kernel call.  (Its
            # evaluations have happened: a func that counts its own calls sees `n_eval` more than the reference's.)
            func.nfe = nfe_first
            return None
        if auto and n_t > 2:
            # "auto": replay only what is safe and worth it — a func whose evaluation visibly changed its own state (a
            # counter, a cache, random numbers) is not captured; nor is a grid too short to pay for the capture
            base = func.base_func
            reason = None
            try:
                reason = _GraphStep._refused.get(base)
            except TypeError:
                pass
            if reason is None and _side_effect_fingerprint(base, self.device) != before:
                reason = ("evaluating it changed its own attributes, buffers or the device's random-number state (an "
                          "evaluation counter, a cache, dropout ...), which a replay would not repeat")
                try:
                    _GraphStep._refused[base] = reason
                except TypeError:
                    pass
                if self._graph_warn:
                    warnings.warn("hip_graph='auto': {} is not captured into a hipGraph — {}; its solves run on the eager "
                                  "path (pass hip_graph=True to capture it regardless)".format(type(base).__name__, reason))
            if reason is not None or n_t - 2 < _AUTO_MIN_GRID_STEPS:
                for _ in range(n_t - 2):
                    step()
                return solution
        if n_t > 2:
            # ... the others are replays of one captured step
            graph = torch.cuda.CUDAGraph()
            nfe_before = func.nfe
            try:
                with _capture(graph):
                    step()
            except Exception as exc:      # func is not capturable: nothing has run, the same body works eagerly
                func.nfe = nfe_before
                if auto:
                    # remembered per func object: the next solve of a training loop does not try again
                    try:
                        _GraphStep._refused[func.base_func] = "capturing it failed ({!r})".format(exc)
                    except TypeError:
                        pass
                if self._graph_warn:
                    warnings.warn("hip_graph=True: func could not be captured into a hipGraph ({!r}); continuing with "
                                  "the eager path".format(exc))
                for _ in range(n_t - 2):
                    step()
                return solution
            func.nfe = nfe_before
            for _ in range(n_t - 2):
                graph.replay()
            func.nfe += n_eval * (n_t - 2)
            # the graph and its private memory pool go away with this frame: let the replays finish first
            current.synchronize()
        return solution


class Euler(FixedGridODESolver):
    """Forward Euler (fixed_grid.py:6-11): dy = dt * f0."""
    order = 1

    def _step(self, t0, dt, t1, y0, y1_out, sh):
        func = self.func
        f0 = func.eval(t0, y0, self._first_perturb(), shadow=sh.time(0.0))
        y1 = self.ops.combine(y0, [f0], [1.0], float(dt) * func.sign, sh.dt_signed(), out=y1_out)
        return y1, f0

    _graph_times = ((0.0, 2),)                                           # t0 (NEXT under `perturb`)

    def _graph_step(self, ts, y_cur, dt_dev, ctrl):
        f0 = self.func.eval_at(ts[0], y_cur)
        y1 = torch.empty_like(y_cur)
        self.kernels.stage_combine_dev(y1, None, y_cur, [f0], (1.0,), None, ctrl)
        return y1


class Midpoint(FixedGridODESolver):
    """Explicit midpoint (fixed_grid.py:14-21): y_mid = y0 + f0*(dt/2); dy = dt * f(t0 + dt/2, y_mid)."""
    order = 2

    def _step(self, t0, dt, t1, y0, y1_out, sh):
        func, ops = self.func, self.ops
        scalar = type(t0)
        dts = float(dt) * func.sign
        half_dt = self._tmul(scalar, dt, 0.5)
        f0 = func.eval(t0, y0, self._first_perturb(), shadow=sh.time(0.0))
        if is_low(func.np_dtype) and not (torch.is_grad_enabled() and (y0.requires_grad or f0.requires_grad)):
            # 16-bit states: `f0 * half_dt` takes the scalar at fp32 (ATen's second-operand rule), not rounded to the state
            y_mid = torch.empty_like(y0)
            self.kernels.scaled_add(y_mid, y0, f0, float(half_dt) * func.sign)
        else:
            y_mid = ops.combine(y0, [f0], [0.5], dts, sh.dt_signed())
        k2 = func.eval(scalar(t0 + half_dt), y_mid, shadow=sh.time(0.5))
        y1 = ops.combine(y0, [k2], [1.0], dts, sh.dt_signed(), out=y1_out)
        return y1, f0

    _graph_times = ((0.0, 2), (0.5, 0))                                  # t0 (NEXT), t0 + dt/2

    def _graph_step(self, ts, y_cur, dt_dev, ctrl):
        func, kern = self.func, self.kernels
        f0 = func.eval_at(ts[0], y_cur)
        y_mid = torch.empty_like(y_cur)
        kern.stage_combine_dev(y_mid, None, y_cur, [f0], (0.5,), None, ctrl)
        k2 = func.eval_at(ts[1], y_mid)
        y1 = torch.empty_like(y_cur)
        kern.stage_combine_dev(y1, None, y_cur, [k2], (1.0,), None, ctrl)
        return y1


class Heun2(FixedGridODESolver):
    """Heun's 2nd-order method through the reference's rk2 step (fixed_grid.py:49-60, rk_common.py:142-157)."""
    order = 2

    def _step(self, t0, dt, t1, y0, y1_out, sh):
        func, ops = self.func, self.ops
        scalar = type(t0)
        dts = float(dt) * func.sign
        k1 = func.eval(t0, y0, self._first_perturb(), shadow=sh.time(0.0))
        ya = ops.fixed_stage(1, y0, [k1], [1.0], dts, sh.dt_signed())
        k2 = func.eval(scalar(t0 + self._tmul(scalar, dt, 1.0)), ya, self._last_perturb(), shadow=sh.time(1.0))
        y1 = ops.fixed_stage(0, y0, [k1, k2], [0.5, 0.5], dts, sh.dt_signed(), out=y1_out)
        return y1, k1

    _graph_times = ((0.0, 2), (1.0, 4))                                  # t0 (NEXT), t0 + dt*1.0 (PREV) — not t1 itself

    def _graph_step(self, ts, y_cur, dt_dev, ctrl):
        func, kern = self.func, self.kernels
        k1 = func.eval_at(ts[0], y_cur)
        ya = torch.empty_like(y_cur)
        kern.fixed_stage_dev(1, ya, y_cur, [k1], (1.0,), dt_dev)
        k2 = func.eval_at(ts[1], ya)
        y1 = torch.empty_like(y_cur)
        kern.fixed_stage_dev(0, y1, y_cur, [k1, k2], (0.5, 0.5), dt_dev)
        return y1


class Heun3(FixedGridODESolver):
    """Heun's 3rd-order method through the reference's rk3 step (fixed_grid.py:32-46, rk_common.py:121-140)."""
    order = 3

    def _step(self, t0, dt, t1, y0, y1_out, sh):
        func, ops = self.func, self.ops
        scalar = type(t0)
        dts = float(dt) * func.sign
        third, two_thirds = 1 / 3, 2 / 3
        k1 = func.eval(t0, y0, self._first_perturb(), shadow=sh.time(0.0))
        ya = ops.fixed_stage(1, y0, [k1], [third], dts, sh.dt_signed())
        k2 = func.eval(scalar(t0 + self._tmul(scalar, dt, third)), ya, shadow=sh.time(third))
        # The tableau's structural zeros (k1 in the third stage, k2 in the result): the kernels do not read a term whose
        # weight is zero; the torch-op host path evaluates the reference's literal `k1 * 0.0 + k2 * (2/3)`
        # (fixed_grid.py:38-44) — the same number for finite stages, and NaN instead of inf once a stage is non-finite
        literal = getattr(self.kernels, "literal_row_sums", False)
        yb = ops.fixed_stage(0, y0, *(([k1, k2], [0.0, two_thirds]) if literal else ([k2], [two_thirds])),
                             dts, sh.dt_signed())
        k3 = func.eval(scalar(t0 + self._tmul(scalar, dt, two_thirds)), yb, shadow=sh.time(two_thirds))
        y1 = ops.fixed_stage(0, y0, *(([k1, k2, k3], [1 / 4, 0.0, 3 / 4]) if literal else ([k1, k3], [1 / 4, 3 / 4])),
                             dts, sh.dt_signed(), out=y1_out)
        return y1, k1

    _graph_times = ((0.0, 2), (1 / 3, 0), (2 / 3, 0))                    # t0 (NEXT), t0 + dt/3, t0 + 2dt/3

    def _graph_step(self, ts, y_cur, dt_dev, ctrl):
        func, kern = self.func, self.kernels
        third, two_thirds = 1 / 3, 2 / 3
        k1 = func.eval_at(ts[0], y_cur)
        ya = torch.empty_like(y_cur)
        kern.fixed_stage_dev(1, ya, y_cur, [k1], (third,), dt_dev)
        k2 = func.eval_at(ts[1], ya)
        yb = torch.empty_like(y_cur)
        kern.fixed_stage_dev(0, yb, y_cur, [k2], (two_thirds,), dt_dev)
        k3 = func.eval_at(ts[2], yb)
        y1 = torch.empty_like(y_cur)
        kern.fixed_stage_dev(0, y1, y_cur, [k1, k3], (1 / 4, 3 / 4), dt_dev)
        return y1


def _rk4_38_step(solver, t0, dt, t1, y0, k1, y1_out, sh):
    """One 3/8-rule step (rk_common.py:110-118); `k1` = func(t0, y0) when the caller already has it (the Adams
    methods' start-up steps, fixed_adams.py:200), else it is evaluated here.  Returns (y1, k1)."""
    func, ops = solver.func, solver.ops
    scalar = type(t0)
    third, two_thirds = 1 / 3, 2 / 3
    dts = float(dt) * func.sign
    stages = [(scalar(t0 + solver._tmul(scalar, dt, third)), Perturb.NONE),
              (scalar(t0 + solver._tmul(scalar, dt, two_thirds)), Perturb.NONE),
              (t1, solver._last_perturb())]
    shadows = [sh.time(third), sh.time(two_thirds), sh.time(1.0)]
    if k1 is None:
        stages.insert(0, (t0, solver._first_perturb()))
        shadows.insert(0, sh.time(0.0))
    ts = func.time_tensors(solver.kernels, stages, shadows=shadows)
    dsh = sh.dt_signed()
    if k1 is None:
        k1 = func.eval_at(ts[0], y0)
        ts = ts[1:]
    ya = ops.rk4_stage(1, y0, k1, None, None, None, dts, dsh)
    k2 = func.eval_at(ts[0], ya)
    yb = ops.rk4_stage(2, y0, k1, k2, None, None, dts, dsh)
    k3 = func.eval_at(ts[1], yb)
    yc = ops.rk4_stage(3, y0, k1, k2, k3, None, dts, dsh)
    k4 = func.eval_at(ts[2], yc)
    y1 = ops.rk4_stage(4, y0, k1, k2, k3, k4, dts, dsh, out=y1_out)
    return y1, k1


class RK4(FixedGridODESolver):
    """Fixed-grid 4th-order RK, 3/8 rule (fixed_grid.py:24-29 -> rk_common.py:110-118)."""
    order = 4

    def _step(self, t0, dt, t1, y0, y1_out, sh):
        return _rk4_38_step(self, t0, dt, t1, y0, None, y1_out, sh)

    # -- hipGraph mode (FixedGridODESolver._integrate_graph) ------------------------------------------------
    _graph_times = ((0.0, 2), (1 / 3, 0), (2 / 3, 0), (0.0, 1 | 4))     # t0 (NEXT), t0 + dt/3, t0 + 2dt/3, t1 (PREV)

    def _graph_step(self, ts, y_cur, dt_dev, ctrl):
        func, kern = self.func, self.kernels
        k1 = func.eval_at(ts[0], y_cur)
        ya = torch.empty_like(y_cur)
        kern.rk4_stage_dev(1, ya, y_cur, k1, None, None, None, dt_dev)
        k2 = func.eval_at(ts[1], ya)
        yb = torch.empty_like(y_cur)
        kern.rk4_stage_dev(2, yb, y_cur, k1, k2, None, None, dt_dev)
        k3 = func.eval_at(ts[2], yb)
        yc = torch.empty_like(y_cur)
        kern.rk4_stage_dev(3, yc, y_cur, k1, k2, k3, None, dt_dev)
        k4 = func.eval_at(ts[3], yc)
        y1 = torch.empty_like(y_cur)
        kern.rk4_stage_dev(4, y1, y_cur, k1, k2, k3, k4, dt_dev)
        return y1


# ---------------------------------------------------------------------------------------------------
# Adams–Bashforth(–Moulton) multistep methods on a fixed grid
# ---------------------------------------------------------------------------------------------------
