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
  AdamsBashforth(Moulton)       fixed_adams.py:164-228

with time-like scalars (t0, t1, dt, rtol, ...) as host doubles instead of 0-dim device tensors.
"""
from __future__ import annotations

import bisect
import collections
import math
import os
import warnings
from typing import List, Optional, Sequence

import numpy as np
import torch

from . import _native
# captured trial steps, their cache and the "auto" policy live in _graph.py; the size limits and step thresholds are READ
# here (tools patch `solvers._GRAPH_MODE_MAX_ELEMENTS` to measure beyond the shipped limit)
from ._graph import (_AUTO_CAPTURE_AFTER_STEPS, _AUTO_MIN_GRID_STEPS, _GRAPH_AUTO_MAX_ELEMENTS,  # noqa: F401
                     _GRAPH_MODE_MAX_ELEMENTS, _CaptureFailed, _DtCell, _GraphStep, _capture, _graph_request,
                     _held_tensor_ptrs, _reusable_across_solves, _scalar_state, _side_effect_fingerprint, _side_stream,
                     clear_graph_cache)
from ._scalars import is_low, power, rdiv, scalar_type
from .autodiff import Ops, stitch
from .misc import (BuiltinNorm, OdeFunc, Perturb, StateLayout, component_norm, find_event, handle_unused_kwargs, rms_norm,
                   vector_tolerances)
from .misc import _null_callback as _null
from .tableaus import (ADAPTIVE_HEUN, ADAPTIVE_TABLEAUS, BOSH3, CARRY_DEFAULT_ON, DOPRI5, DOPRI8, FEHLBERG2, TSIT5, SparseRow, Tableau,
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


class _LockStep:
    """Cross-rank reduction of the norm sums for a batch-sharded solve whose shards must take identical steps
    (SURVEY.md §8e "exact mode").  One all-reduce of 3·n_seg doubles per norm evaluation; segments that hold
    replicated data (the adjoint's time / parameter adjoints once they are all-reduced per evaluation) are counted
    from rank 0 only.  Nothing of this exists in the reference."""

    def __init__(self, group, replicated=()):
        import torch.distributed as dist
        self.dist = dist
        self.group = None if group is True else group
        self.replicated = sorted(set(int(i) for i in replicated))
        self.rank = dist.get_rank(self.group)
        backend = dist.get_backend(self.group)
        self.on_device = backend == "nccl"      # RCCL reduces device buffers; gloo host buffers

    def _allreduce(self, values: List[float], device) -> List[float]:
        v = torch.tensor(values, dtype=torch.float64, device=device if self.on_device else "cpu")
        self.dist.all_reduce(v, op=self.dist.ReduceOp.SUM, group=self.group)
        return v.tolist()

    def global_numels(self, numels: Sequence[int], device) -> List[int]:
        mine = [0 if (i in self.replicated and self.rank != 0) else int(n) for i, n in enumerate(numels)]
        return [int(round(x)) for x in self._allreduce([float(m) for m in mine], device)]

    def reduce_device(self, buf: torch.Tensor, n: int) -> None:
        """In-place all-reduce of a norm plan's DEVICE result words [n sums | n (second sum) | n non-finite counters]
        — RCCL over xGMI on the buffer the finalize kernel wrote, no host round trip; replicated segments are taken
        from rank 0 only."""
        if self.rank != 0 and self.replicated:
            if getattr(self, "_zero_idx", None) is None or self._zero_idx.device != buf.device:
                idx = [q * n + s for s in self.replicated for q in range(3)]
                self._zero_idx = torch.tensor(idx, dtype=torch.int64, device=buf.device)
            buf.index_fill_(0, self._zero_idx, 0.0)
        self.dist.all_reduce(buf[:3 * n], op=self.dist.ReduceOp.SUM, group=self.group)

    def reduce(self, s0, s1, bad, device):
        n = len(s0)
        if self.rank != 0:
            s0, s1, bad = list(s0), list(s1), list(bad)
            for i in self.replicated:
                s0[i] = s1[i] = bad[i] = 0.0
        # entries a launch did not write (second sum of a one-sum launch) may hold stale values: harmless, unused
        out = self._allreduce(list(s0) + list(s1) + list(bad), device)
        return out[:n], out[n:2 * n], out[2 * n:]




class _InitialStepShadow:
    """Autograd graph of the starting step size (misc.py:36-77), for backprop through the solver.  The VALUES of the
    heuristic come from the norm kernels (host scalars, `_select_initial_step`); this records the same formulas with
    torch ops on the live tensors, taking the branches the host took, so that the first step size carries the
    gradient the reference's does (its heuristic is not under no_grad).  Segment-wise: scale_s = atol_s + |y0|·rtol_s,
    norm = max over segments of sqrt(mean(x²)) (misc.py:22-33)."""

    def __init__(self, solver, y0, f0, h0_value: float, h0_is_const: bool):
        s = self.s = solver
        lay = s.layout
        self.segs = [(off, n, rt, at) for off, n, rt, at in lay.segments(*s._seg_tol) if n > 0]
        self.y0, self.f0 = y0, f0
        if s._vec_tol is not None:      # per-element tolerances
            part = lambda v, off, n: v if v.dim() == 0 else v[off:off + n]
            self.scale = [part(s._vec_tol[1], off, n) + y0[off:off + n].abs() * part(s._vec_tol[0], off, n)
                          for off, n, _, _ in self.segs]
        else:
            self.scale = [at + y0[off:off + n].abs() * rt for off, n, rt, at in self.segs]
        self.d0 = self._norm(y0)
        self.d1 = self._norm(f0)
        if h0_is_const:
            self.h0 = torch.full((), h0_value, dtype=s.func.time_dtype, device=y0.device)
        else:
            self.h0 = stitch(torch.full((), h0_value, dtype=s.func.time_dtype, device=y0.device),
                             (0.01 * self.d0 / self.d1).abs())
        # y1 = y0 + h0 * f0 in solver time (f0 is the raw func output: the time sign rides on h0)
        self.y1 = y0 + (self.h0 * s.func.sign) * f0

    def _norm(self, x, diff=None):
        vals = []
        for (off, n, _, _), sc in zip(self.segs, self.scale):
            v = x[off:off + n] if diff is None else x[off:off + n] - diff[off:off + n]
            vals.append((v / sc).abs().pow(2).mean().sqrt())
        return max(vals) if vals else torch.zeros((), dtype=x.dtype, device=x.device)

    def time_shadow(self, anchor, sign: float):
        """User-time graph of t0 + h0 for the heuristic's own evaluation of func."""
        base = self.h0 if anchor is None else anchor + self.h0
        return base * sign

    def finish(self, f1, h1_is_floor: bool, d1_is_max: bool, h0_branch: bool, order: int, value: float):
        """dt0 = min(100·h0, h1) with the branches the host took; returns a 0-dim fp64 tensor whose value is the
        host's first step and whose graph is the heuristic's."""
        d2 = (self._norm(f1, diff=self.f0) / self.h0).abs()
        if h1_is_floor:
            h1 = self.h0 * 1e-3       # (the constant 1e-6 floor carries no gradient either way)
        else:
            h1 = (0.01 / (self.d1 if d1_is_max else d2)) ** (1.0 / float(order + 1))
        h1 = h1.abs()
        dt0 = (100 * self.h0) if h0_branch else h1
        dt0 = dt0.to(torch.float64)
        if not dt0.requires_grad:
            return None
        return stitch(torch.full((), value, dtype=torch.float64, device=dt0.device), dt0)


class _DenseRecord:
    """Data of the last accepted step, kept for lazy dense output (rk_common.py:363-369)."""
    __slots__ = ("y0", "y1", "k", "dt_signed", "t0", "t1", "dt_shadow", "anchor")

    def __init__(self):
        self.dt_shadow = None     # graph of the step size (only the first step's, see _initial_step_shadow)
        self.anchor = None        # time anchor (graph of the step's start time) when the step was taken


class RKAdaptiveStepsizeODESolver:
    """Adaptive embedded RK pair driven from the host; subclasses set `order` and `tableau`."""
    order: int
    tableau: Tableau
    flat_state_native = True         # takes the package's padded flat state / BuiltinNorm / per-segment tolerances (odeint.py)
    func_output_numel_must_match = True     # misc.OdeFunc._conform: no silent expansion of a too-small func output

    def __init__(self, func: OdeFunc, y0: torch.Tensor, rtol, atol, min_step=0, max_step=float("inf"),
                 first_step=None, step_t=None, jump_t=None, safety=0.9, ifactor=10.0, dfactor=0.2,
                 max_num_steps=2 ** 31 - 1, dtype=torch.float64, norm=None, dist_sync=None, dist_replicated=(),
                 hip_graph=None, **unused_kwargs):
        handle_unused_kwargs(self, unused_kwargs)
        del unused_kwargs
        if not isinstance(func, OdeFunc):
            raise TypeError("solver classes of torchdiffeq_amd take the wrapped func built by check_inputs")
        self.func = func
        self.y0 = y0
        # Lock-step mode of a batch-sharded solve (torchdiffeq_amd.dist): the per-segment error sums of all ranks
        # are added (one small all-reduce per norm evaluation), so every shard takes the steps of the whole-batch
        # solve.  `dist_replicated` = segments whose content is identical on every rank (counted once).
        self._sync = _LockStep(dist_sync, dist_replicated) if dist_sync is not None else None
        self.layout: StateLayout = func.layout
        self.state_dtype = y0.dtype
        self.np_dtype = func.np_dtype        # T = y0.abs().dtype (real also for complex states: rk_common.py:61)
        # `dtype` (rk_common.py:176-194): the type W of every time-like scalar — t, t0, t1, dt, the controller's
        # constants —, promoted with T = y0.abs().dtype.  They are host numbers here; `_w` rounds one to W after each
        # operation (identity for the default fp64: Python floats are W).  fp64 -> fp32 double rounding is innocuous for
        # + - * /, so `_w(a op b)` on W-valued doubles is the W operation.
        self.dtype = torch.promote_types(dtype, func.time_dtype)
        self._W = scalar_type(self.dtype)
        self._wide = self._W is np.float64
        self.norm = rms_norm if norm is None else norm
        if not self._wide:
            # `torch.as_tensor(rtol, dtype=W)` (rk_common.py:186-187): the tolerances are W numbers before anything uses them
            in_w = lambda tol: tol.to(self.dtype) if isinstance(tol, torch.Tensor) else (
                type(tol)(in_w(v) for v in tol) if isinstance(tol, (list, tuple)) else self._w(float(tol)))
            rtol, atol = in_w(rtol), in_w(atol)
        self.rtol, self.atol = rtol, atol
        # Per-element tolerances (a tensor / list broadcasting against the state — plain broadcasting in the reference,
        # misc.py:80-82): the kernels take one (rtol, atol) per segment, so they are asked for the RAW error and initial-
        # step quantities (tolerances 0 and 1) and the per-element scaling and the norm run as torch ops in fp64, which is
        # also the reference's precision for this case.  Routed like a user norm: host-driven steps, no captured graphs.
        self.kernels = _native.get_kernels(y0.device, y0.dtype)
        self._vec_tol = vector_tolerances(rtol, atol, self.layout, y0.device, self.dtype,
                                          tuple_entries_too=getattr(self.kernels, "literal_norms", False)
                                          and dist_sync is None)
        self._vec_fused = None
        if self._vec_tol is not None:
            rtol, atol = 0.0, 1.0
            if isinstance(self.norm, BuiltinNorm):
                # r05: the per-step error norm stays ONE fused launch (tdeq_error_norm_vec: the tolerance vectors are two more
                # fp64 streams of the same kernel) for real fp32 / fp64 states with fp64 tolerances; the once-per-solve
                # initial-step norms and everything else keep the callable below
                if hasattr(self.kernels, "error_norm_vec") and self._wide and dist_sync is None \
                        and y0.dtype in (torch.float32, torch.float64):
                    self._vec_fused = tuple(v.contiguous() if v.dim() else float(v) for v in self._vec_tol) \
                        + (self.norm.n_skip_tail,)
                self.norm = component_norm(self.layout, self.norm.n_skip_tail)
        self._seg_tol = (rtol, atol)
        w = self._w
        self.min_step = w(_as_float(min_step))
        self.max_step = w(_as_float(max_step))
        self.first_step = None if first_step is None else w(_as_float(first_step))
        self.safety = w(_as_float(safety))
        self.ifactor = w(_as_float(ifactor))
        self.dfactor = w(_as_float(dfactor))
        self.max_num_steps = int(_as_float(max_num_steps))
        self.step_t = None if step_t is None else torch.as_tensor(step_t, dtype=self.dtype).reshape(-1).tolist()
        self.jump_t = None if jump_t is None else torch.as_tensor(jump_t, dtype=self.dtype).reshape(-1).tolist()

        self.ops = Ops(self.kernels, self.np_dtype)      # elementwise kernels, differentiable when grad is needed
        self.plan = self.kernels.make_plan(self.layout.segments(rtol, atol), self.layout.total,
                                           self.layout.chunk, y0.device)
        # element counts behind the per-segment sums (global counts in lock-step mode)
        self._numels = list(self.plan.numels) if self._sync is None else self._sync.global_numels(self.plan.numels,
                                                                                                  y0.device)
        if self._sync is not None and not isinstance(self.norm, BuiltinNorm):
            raise NotImplementedError("lock-step sharded solves need a builtin norm (a user norm callable reduces "
                                      "only this rank's rows)")
        self._anchor = None        # t[0] (solver time) when `t` requires grad: every step time moves with it
        self._t_grad = False       # `t` requires grad (output times carry their own gradient)
        tab = self.tableau
        # rows without their zero weights for the kernels; the torch-op host path multiplies every slot like the reference
        literal_rows = bool(getattr(self.kernels, "literal_row_sums", False))
        row = SparseRow.literal if literal_rows else SparseRow.from_dense
        self._beta = tab.beta_rows(literal_rows)
        self._c_err, self._c_mid, self._c_sol = row(tab.c_error), row(tab.c_mid), row(tab.c_sol)
        # stage abscissae rounded to the state dtype, as the reference's tableau cast (rk_common.py:201)
        self._alpha = [self.np_dtype(a) for a in tab.alpha]
        self._alpha_is_one = [a == 1.0 for a in tab.alpha]
        # End-of-step fusion (tdeq_stage_combine_err + tdeq_error_norm_partial): the step's last combine — the last
        # stage row of an FSAL pair, else the c_sol combine — also emits the partial embedded error over its own
        # stages.  Bit-identical only if those stages are a leading run of the error row's non-zeros.
        last = self._beta[-1] if tab.fsal_solution else self._c_sol
        n_lead = len(last.idx)
        self._fuse = None
        # (not on the torch-op host path, which hands every row to ATen's `torch.sum` whole — `literal_row_sums`,
        # _fallback.py: splitting a row re-associates it, and for bf16 / fp16 states would round it twice)
        # (nor for reduced-precision states on the HIP kernels — `split_row_sums` False: a row is rounded ONCE)
        if self._c_err.idx[:n_lead] == last.idx and len(self._c_err.idx) - n_lead <= 2 \
                and not getattr(self.kernels, "literal_row_sums", False) \
                and getattr(self.kernels, "split_row_sums", True):
            self._fuse = (self._c_err.coef[:n_lead], self._c_err.idx[n_lead:], self._c_err.coef[n_lead:])
        # Carried partial sums (tableaus.carry_plan / tdeq_stage_combine_multi): fewer bytes per step for the same bits.
        # TDEQ_CARRY: unset / "auto" = the tableaus where it is a measured gain (CARRY_DEFAULT_ON); "1" = every
        # tableau that has a plan; "0" = row-by-row launches.
        self._carry = None
        carry_env = os.environ.get("TDEQ_CARRY", "auto").lower()
        if self._fuse is not None and carry_env != "0" \
                and (carry_env == "1" or self.layout.total >= CARRY_DEFAULT_ON.get(tab.name, float("inf"))) \
                and hasattr(self.kernels, "stage_combine_multi") and ADAPTIVE_TABLEAUS.get(tab.name) is tab:
            self._carry = carry_plan(tab.name)
        self.n_accepted = 0
        self.n_rejected = 0
        # Device-resident controller + look-ahead first stage (tdeq_error_norm_partial_ctrl / tdeq_stage_combine_sel):
        # the loop stays here, but the scalar decision of a trial step is also taken on the device so that the next
        # trial step's first stage and func evaluation are enqueued before the decision has been read back.
        n_norm_seg = self.layout.n_seg - (self.norm.n_skip_tail if isinstance(self.norm, BuiltinNorm) else 0)
        # (lock-step sharding: only when the collective runs on device buffers — backend nccl = RCCL —, where the sums
        # are all-reduced between the norm's finalize and the controller kernel without leaving the GPU)
        sync_dev = self._sync is not None and self._sync.on_device and y0.device.type == "cuda" \
            and hasattr(self.kernels, "step_controller")
        # (reduced-precision states on the HIP kernels — `whole_row_controller`: the controller launch takes the WHOLE error
        #  row instead of continuing a fused partial sum; builtin norms without the adjoint's |t| component, stage times in
        #  the state's type)
        self._whole_row_ctrl = bool(getattr(self.kernels, "whole_row_controller", False)) and self._fuse is None \
            and not getattr(self.norm, "leading_scalar", False) and func.time_dtype == y0.dtype and dist_sync is None
        device_ctrl = (getattr(self.kernels, "device_controller", True)
                       and (self._fuse is not None or self._whole_row_ctrl) and isinstance(self.norm, BuiltinNorm)
                       and len(self._beta) <= _native.TDEQ_MAX_STAGE_TIMES and n_norm_seg >= 0
                       and self.step_t is None and self.jump_t is None and (self._sync is None or sync_dev)
                       and self._wide)          # the device controller computes in fp64: W = fp64 only
        self._plan_dev = self._plan_glob = None
        if device_ctrl and self._sync is not None:
            segs = self.layout.segments(rtol, atol)
            self._plan_dev = _native.NormPlan(segs, self.layout.total, self.layout.chunk, y0.device, pinned=False)
            self._plan_glob = self.kernels.make_plan([(off, gn, rt, at) for (off, _, rt, at), gn in
                                                      zip(segs, self._numels)], self.layout.total,
                                                     self.layout.chunk, y0.device)
        self._lookahead = device_ctrl and os.environ.get("TDEQ_LOOKAHEAD", "1") != "0"
        # `hip_graph=True` (an extension, not a reference option): one captured hipGraph per trial step, see _GraphStep
        # "auto" = only where it pays (states up to _GRAPH_AUTO_MAX_ELEMENTS) and silently; never the built-in default,
        # because a captured func runs in Python only while the graph is being built: per-evaluation Python side
        # effects (an evaluation counter, data-dependent branches) are not replayed — the user has to vouch for that
        wanted, auto = _graph_request(hip_graph)
        self._graph_auto = auto
        self._auto = None           # auto mode: this solve's policy ("now" / "later" / "never", _GraphStep.auto_policy)
        self._auto_steps = 0
        self._hold_pre = False      # auto mode: the eager step before the switch to replays enqueues no look-ahead stage
        self.hip_graph = wanted and device_ctrl and self._sync is None and y0.device.type == "cuda" \
            and hasattr(self.kernels, "stage_combine_dev") \
            and self.layout.total <= (_GRAPH_AUTO_MAX_ELEMENTS if auto else _GRAPH_MODE_MAX_ELEMENTS)
        if wanted and not auto and not self.hip_graph and hip_graph is not None:
            # (asked for by option; a process-wide TDEQ_HIP_GRAPH=1 default applies where it can and stays silent)
            warnings.warn("{}: hip_graph=True needs a builtin norm, no step_t / jump_t, a tableau with a fused error "
                          "combine, a ROCm device, no lock-step sharding (dist_sync: a per-step collective does not "
                          "belong in a captured graph) and a state of at most {} elements (larger states are "
                          "bandwidth-bound: the eager path with its unrolled kernels is the fast one); running the "
                          "eager path".format(self.__class__.__name__, _GRAPH_MODE_MAX_ELEMENTS))
        self._g = None
        self._dt_shadow = None      # autograd graph of the current step size (the first, heuristic one only)
        if device_ctrl:
            c = _native.StepCtrl()
            c.safety, c.ifactor, c.dfactor = self.safety, self.ifactor, self.dfactor
            c.exponent = 1.0 / self.order
            c.min_step, c.max_step = self.min_step, self.max_step
            c.time_sign = func.sign
            mask = 0
            for i, a in enumerate(self._alpha):
                c.alpha[i] = float(a)
                if self._alpha_is_one[i]:
                    mask |= 1 << i
            c.alpha_is_one = mask
            c.n_times = len(self._alpha)
            c.n_norm_seg = n_norm_seg
            self._ctrl = c
        self._max_rows = _native.TDEQ_MAX_DENSE_OUTPUTS if os.environ.get("TDEQ_DENSE_MULTI", "1") != "0" else 1
        self._t_end = -math.inf     # last output time of the running `integrate` (look-ahead only before it)
        self._pre = None            # (stage input, stage times, k_1) of the trial step enqueued ahead
        self._last_trial = False    # (_step_until) this trial step is the last one the max_num_steps budget allows

    @classmethod
    def valid_callbacks(cls):
        return {"callback_step", "callback_accept_step", "callback_reject_step"}

    # -- norms -------------------------------------------------------------------------------------
    @np.errstate(all="ignore")     # host scalars follow IEEE silently, as 0-dim tensors do
    def _segment_norm(self, sumsq: Sequence[float], bad: Sequence[float], which: int = 0):
        """max over the selected segments of sqrt(mean), rounded to the state dtype (misc.py:22-33).  `which`: the first /
        second sum of the last norm launch — only needed on the torch-op host path, whose plan also holds the
        reference's own norm values (`literal_norms`: ATen's `abs().pow(2).mean().sqrt()` per segment) and they are the
        ones used there."""
        numels = self._numels
        n = len(numels)
        if isinstance(self.norm, BuiltinNorm) and self.norm.n_skip_tail:
            n -= self.norm.n_skip_tail
        literal = None
        if self._sync is None and getattr(self.kernels, "literal_norms", False):
            literal = list(self.plan.rms1 if which else self.plan.rms0)
            if getattr(self.norm, "leading_scalar", False) and numels[0] == 1:
                # adjoint.py:250, 273: `max(t.abs(), ...)` — the time component is not squared
                literal[0] = (self.plan.abs1 if which else self.plan.abs0)[0]
        val = 0.0
        for s in range(n):
            if numels[s] == 0:
                continue
            val = _nan_max(val, literal[s] if literal is not None else math.sqrt(sumsq[s] / numels[s]))
        with np.errstate(over="ignore"):
            return float(self.np_dtype(val))

    def _read_norms(self):
        """Results of the last norm launch; in lock-step mode summed over the ranks of the process group."""
        s0, s1, bad = self.kernels.read_norms(self.plan)
        if self._sync is not None:
            s0, s1, bad = self._sync.reduce(s0, s1, bad, self.y0.device)
        return s0, s1, bad

    def _w(self, x: float) -> float:
        """`x` rounded to the time dtype W (see __init__)."""
        return x if self._wide else float(self._W(x))

    def _time_tensor(self, value: float) -> torch.Tensor:
        return torch.tensor(value, dtype=self.dtype, device=self.y0.device)

    # -- integrate ---------------------------------------------------------------------------------
    @_native.on_state_device
    def integrate(self, t: torch.Tensor) -> torch.Tensor:
        """solution[len(t), total] with solution[0] = y0 (solvers.py:28-35)."""
        t_host = t.detach().to(self.dtype).cpu().tolist()
        self._set_time_anchor(t)
        self._before_integrate(t_host)
        self._t_end = t_host[-1]
        if self._differentiable():
            # backprop through the solver: rows are autograd nodes, assembled by a differentiable stack
            rows = [self.y0]
            for i in range(1, len(t_host)):
                rows.append(self._advance(t_host[i], None, t[i] if self._t_grad else None))
            return torch.stack(rows, dim=0)
        solution = torch.empty(len(t_host), self.layout.total, dtype=self.y0.dtype, device=self.y0.device)
        solution[0].copy_(self.y0)
        i, n_t = 1, len(t_host)
        try:
            while i < n_t:
                self._step_until(t_host[i])
                # every output time inside the step just accepted is an interpolation of that step
                # (rk_common.py:243-250): one launch per <= 16 of them (tdeq_dense_eval_multi), not one per time
                j = i + 1
                while j < n_t and t_host[j] <= self.t1 and j - i < self._max_rows:
                    j += 1
                self._interp_evaluate_rows(t_host[i:j], solution[i:j])
                i = j
        finally:
            if self._g is not None:
                # hipGraph mode: the step's static buffers are about to be re-armed by the next solve that takes
                # this (cached) graph — let the kernels that still read them finish, then hand the graph back
                torch.cuda.current_stream(self.y0.device).synchronize()
                self._g.release()
        return solution

    def _set_time_anchor(self, t: torch.Tensor) -> None:
        self._t_grad = torch.is_grad_enabled() and t.requires_grad
        # (the anchor can be lost on the way — a step ending on a `step_t` / `jump_t` point starts the next one at a
        # constant —, the output times keep their own gradient: `_t_grad`)
        self._anchor = t[0] if self._t_grad else None
        self.func.set_time_anchor(self._anchor)

    def _differentiable(self) -> bool:
        """True when the solution must carry an autograd graph (grad mode on and y0, `t` or func's output —
        i.e. its parameters — require grad)."""
        return torch.is_grad_enabled() and (self.y0.requires_grad or self._anchor is not None or
                                            self.f1.requires_grad)

    @_native.on_state_device
    def integrate_dense(self, t: torch.Tensor):
        """Integrate over [t[0], t[-1]] keeping the dense output of EVERY accepted step (odeint.py:124-147):
        returns (times, coeffs) with `times` the n_steps + 1 accepted step boundaries (host doubles) and
        `coeffs[n_steps, 5, total]` the quartic coefficients [e, d, c, b, a] (`tdeq_interp_fit`)."""
        t_host = t.detach().to(self.dtype).cpu().tolist()
        self._set_time_anchor(t.detach())
        self._before_integrate(t_host)
        self._t_end = t_host[-1]
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

    def _before_integrate(self, t_host: List[float]) -> None:
        t0 = t_host[0]
        self._dt_shadow = None
        f0 = self.func.eval(t0, self.y0)
        if self.first_step is None:
            first_step = self._select_initial_step(t0, self.y0, f0)
        else:
            first_step = self.first_step
            # no initial-step heuristic -> still take the non-finite census of y0 (rk_common.py:287)
            self.kernels.init_norms(self.plan, 1, f0.detach(), f0.detach(), self.y0.detach())
            _, _, bad = self._read_norms()
            self._y_nonfinite = any(b != 0 for b in bad)
        self.y1, self.f1 = self.y0, f0
        self.t0, self.t1, self.dt = t0, t0, first_step
        self._dense: Optional[_DenseRecord] = None
        self._t_end, self._pre = -math.inf, None    # event mode / direct stepping: no look-ahead
        self._g = None
        self._auto, self._auto_steps, self._hold_pre = None, 0, False
        if self.first_step is not None:
            self._dt_shadow = None

        step_t = [] if self.step_t is None else sorted(v for v in self.step_t if v >= t0)
        jump_t = [] if self.jump_t is None else sorted(v for v in self.jump_t if v >= t0)
        both = step_t + jump_t
        if len(set(both)) != len(both):
            raise ValueError("`step_t` and `jump_t` must not have any repeated elements between them.")
        self._step_t, self._jump_t = step_t, jump_t
        self.next_step_index = min(bisect.bisect(step_t, t0), len(step_t) - 1)
        self.next_jump_index = min(bisect.bisect(jump_t, t0), len(jump_t) - 1)

    @np.errstate(all="ignore")     # host scalars follow IEEE silently, as 0-dim tensors do
    def _select_initial_step(self, t0: float, y0: torch.Tensor, f0: torch.Tensor) -> float:
        """Hairer II.4 starting step (misc.py:36-77), scalars in the state precision T."""
        T = self.np_dtype
        # per-element tolerances are W tensors: the heuristic's norms and everything formed from them promote to W
        # (misc.py:50-77 on tensors), only the constant 1e-6 floors stay in the state's type
        S = self._W if self._vec_tol is not None and not is_low(self._W) else T
        kern, plan = self.kernels, self.plan
        order = self.order - 1   # the reference passes `self.order - 1` (rk_common.py:217)
        # Values come from the kernels on detached data; when the solve is differentiated, the SAME formulas are
        # recorded a second time with torch ops (`_initial_step_shadow`) only to carry the gradient: the reference's
        # heuristic is not under no_grad, so its first step size is a differentiable function of y0, f0 and f1.
        shadow_on = self.first_step is None and torch.is_grad_enabled() and \
            (y0.requires_grad or f0.requires_grad or self._anchor is not None)
        y0_g, f0_g = y0, f0
        y0, f0 = y0.detach(), f0.detach()
        user_norm = not isinstance(self.norm, BuiltinNorm)
        kern.init_norms(plan, 0, y0, f0, y0)
        s0, s1, bad = self._read_norms()
        self._y_nonfinite = any(b != 0 for b in bad)
        if user_norm:
            # the reference hands ITS norm to the heuristic (rk_common.py:217): materialise the quotients and let the
            # user's callable reduce them
            q0, q1 = torch.empty_like(y0), torch.empty_like(y0)
            kern.init_scaled(plan, 0, y0, f0, y0, q0, q1)
            if self._vec_tol is not None:
                vec_scale = self._vec_tol[1] + y0.abs() * self._vec_tol[0]      # misc.py:50, promoted as there
                q0, q1 = q0 / vec_scale, q1 / vec_scale
            with torch.no_grad():
                d0, d1 = S(_norm_value(self.norm(q0))), S(_norm_value(self.norm(q1)))
        else:
            d0 = T(self._segment_norm(s0, bad))
            d1 = T(self._segment_norm(s1, bad, which=1))
        # Scalar arithmetic below: each operation as ATen rounds it for 0-dim tensors of type T with Python numbers
        # mixed in (_scalars.py) — `0.01 / x` is reciprocal-then-multiply, `x ** e` is raised in double for fp32.
        if d0 < 1e-5 or d1 < 1e-5:
            h0 = T(1e-6)
        else:
            h0 = 0.01 * d0 / d1
        h0 = abs(h0)
        if shadow_on:
            sh = _InitialStepShadow(self, y0_g, f0_g, float(h0), bool(d0 < 1e-5 or d1 < 1e-5))
            f1_g = self.func.eval(self._w(t0 + float(h0)), sh.y1, shadow=sh.time_shadow(self._anchor, self.func.sign))
            f1 = f1_g.detach()
        else:
            y1 = torch.empty_like(y0)
            kern.stage_combine(y1, y0, [f0], [1.0], float(h0) * self.func.sign)
            with torch.no_grad():
                f1 = self.func.eval(self._w(t0 + float(h0)), y1)
        if user_norm:
            q0 = torch.empty_like(y0)
            kern.init_scaled(plan, 1, f1, f0, y0, q0)
            if self._vec_tol is not None:
                q0 = q0 / vec_scale
            with torch.no_grad():
                d2_num = S(_norm_value(self.norm(q0)))
        else:
            kern.init_norms(plan, 1, f1, f0, y0)
            s2, _, bad = self._read_norms()
            d2_num = T(self._segment_norm(s2, bad))
        with np.errstate(all="ignore"):
            d2 = abs(d2_num / h0)
            if d1 <= 1e-15 and d2 <= 1e-15:
                h1 = max(T(1e-6), h0 * 1e-3)
            else:
                h1 = power(rdiv(0.01, max(d1, d2)), 1.0 / float(order + 1))
            h1 = abs(h1)
            first_step = float(min(100 * h0, h1))
        if shadow_on:
            self._dt_shadow = sh.finish(f1_g, bool(d1 <= 1e-15 and d2 <= 1e-15), bool(d1 >= d2),
                                        bool(T(100) * h0 <= h1), order, first_step)
        return first_step

    def _step_until(self, next_t: float) -> None:
        """Trial steps until next_t is inside the last accepted step (rk_common.py:243-249)."""
        n_steps = 0
        while next_t > self.t1:
            assert n_steps < self.max_num_steps, \
                "max_num_steps exceeded ({}>={})".format(n_steps, self.max_num_steps)
            # (the last trial this budget allows enqueues no look-ahead evaluation: if it does not reach next_t, the
            #  assertion above comes next, after the reference's number of evaluations)
            self._last_trial = n_steps + 1 >= self.max_num_steps
            self._trial_step()
            n_steps += 1

    def _trial_step(self) -> None:
        """One trial step by the path this solve runs on: a hipGraph replay (hip_graph mode, small states) or the
        eager launch sequence."""
        if self.hip_graph and self._graph_step_ok() and self._graph_now():
            try:
                self._graph_trial_step()
            except _CaptureFailed as exc:
                # a failed capture executes nothing: the static buffers still hold the current state, continue
                # with the eager path from it.  The half-built captured step never goes back to the per-func cache:
                # "auto" remembers the func as unfit (one warning per func object, no retry on the next solve of a
                # training loop); hip_graph=True drops the entry so that the next solve starts a fresh capture.
                g, self._g = self._g, None
                if self._graph_auto and g is not None:
                    g.refuse(self, "capturing it failed ({}): a host synchronisation, .item() or data-dependent "
                                   "Python branch inside func".format(exc))
                else:
                    if g is not None:
                        g.evict(self)
                    warnings.warn("hip_graph=True: func could not be captured into a hipGraph ({}); continuing with "
                                  "the eager path".format(exc))
                self.hip_graph = False
                self._hold_pre = False
                self._adaptive_step()
        else:
            self._adaptive_step()

    def _unpadded(self, flat: torch.Tensor) -> torch.Tensor:
        """The state as the reference's flat vector (components back to back, no alignment padding) — error messages."""
        if self.layout.n_seg == 1:
            return flat
        return torch.cat([c.reshape(-1) for c in self.layout.unpack(flat)])

    def _graph_now(self) -> bool:
        """Whether THIS trial step goes through the captured-step path.  Always, once a solve is on it or when
        `hip_graph=True` was asked for; under "auto" the first capture is put off as _GraphStep.auto_policy says."""
        if self._g is not None or not self._graph_auto:
            return True
        if self._auto is None:
            self._auto = _GraphStep.auto_policy(self, getattr(self, "_auto_seen_before", False))
        if self._auto == "never":
            self.hip_graph = False
            return False
        if self._auto == "later":
            self._auto_steps += 1
            if self._auto_steps <= _AUTO_CAPTURE_AFTER_STEPS:
                # (the last eager step enqueues no look-ahead stage, so that the replays start from a clean state)
                self._hold_pre = self._auto_steps == _AUTO_CAPTURE_AFTER_STEPS
                return False
        return self._pre is None

    def _advance(self, next_t: float, out: Optional[torch.Tensor], t_shadow=None) -> torch.Tensor:
        """Step until next_t is inside the last accepted step, then return y(next_t) (written into `out` if
        given).  `t_shadow` = the entry of `t` this output belongs to, when `t` requires grad."""
        self._step_until(next_t)
        return self._interp_evaluate(next_t, out, t_shadow)

    @np.errstate(all="ignore")     # host scalars follow IEEE silently, as 0-dim tensors do
    def _interp_fraction(self, rec, t: float) -> float:
        """x = (t - t0) / (t1 - t0) formed in W, then cast to T (interp.py:39-40)."""
        w = self._w
        return float(self.np_dtype(w(w(t - rec.t0) / w(rec.t1 - rec.t0))))

    def _interp_evaluate_rows(self, times: Sequence[float], rows: torch.Tensor) -> None:
        """y(t) for several output times inside the last accepted step, written to the rows of `rows` (a slice of
        the solution tensor) by one launch; no autograd graph (the differentiable path evaluates row by row)."""
        rec = self._dense
        if len(times) == 1:
            self._interp_evaluate(times[0], rows[0])
            return
        xs = []
        for t in times:
            assert rec is not None and rec.t0 <= t <= rec.t1, \
                "invalid interpolation, fails `t0 <= t <= t1`: {}, {}, {}".format(self.t0, t, self.t1)
            xs.append(self._interp_fraction(rec, t))
        mid = self._c_mid
        self.kernels.dense_eval_multi(rows, rec.y0, rec.y1, rec.k[0], rec.k[-1], [rec.k[j] for j in mid.idx],
                                      mid.coef, rec.dt_signed, xs)

    def _interp_evaluate(self, t: float, out: Optional[torch.Tensor] = None, t_shadow=None) -> torch.Tensor:
        """Fused `_interp_fit` + `_interp_evaluate` (rk_common.py:363-369, interp.py:25-48)."""
        rec = self._dense
        assert rec is not None and rec.t0 <= t <= rec.t1, \
            "invalid interpolation, fails `t0 <= t <= t1`: {}, {}, {}".format(self.t0, t, self.t1)
        x = self._interp_fraction(rec, t)
        x_shadow = None
        if rec.dt_shadow is not None:
            # first step of a differentiated solve: x = (t - t0) / dt0 with dt0 a function of (y0, f0, f1) too
            width = stitch(torch.full((), rec.t1 - rec.t0, dtype=torch.float64, device=rec.y0.device),
                           rec.dt_shadow * self.func.sign)
            num = torch.full((), t - rec.t0, dtype=torch.float64, device=rec.y0.device)
            if t_shadow is not None:
                num = num + (t_shadow - t_shadow.detach())
            if rec.anchor is not None:
                num = num - (rec.anchor - rec.anchor.detach())
            x_shadow = num / width
        elif rec.anchor is not None or t_shadow is not None:
            # x = (t - t0_step) / (t1_step - t0_step): the step boundaries move with the anchor (if any), the width is a
            # constant
            x_shadow = ((t_shadow if t_shadow is not None else 0.0) - (rec.anchor if rec.anchor is not None else 0.0)) \
                / (rec.t1 - rec.t0)
        mid = self._c_mid
        return self.ops.dense_eval(rec.y0, rec.y1, rec.k, mid.idx, mid.coef, rec.dt_signed, x,
                                   dt_shadow=rec.dt_shadow, x_shadow=x_shadow, out=out)

    def _adaptive_step(self) -> None:
        """One trial step (rk_common.py:266-361)."""
        func, kern, T = self.func, self.kernels, self.np_dtype
        y0, f0, t0, dt = self.y1, self.f1, self.t1, self.dt
        if not math.isfinite(dt):
            dt = self.min_step
        if self._dt_shadow is not None and not self.min_step <= dt <= self.max_step:
            self._dt_shadow = None          # `dt.clamp(min_step, max_step)` (rk_common.py:271): a constant outside the range
        dt = _clamp(dt, self.min_step, self.max_step)
        if func.callback_step is not _null:
            func.callback_step(self._time_tensor(t0), y0, self._time_tensor(dt))
        w = self._w
        t1 = w(t0 + dt)
        assert t1 > t0, "underflow in dt {}".format(dt)
        assert not self._y_nonfinite, "non-finite values in state `y`: {}".format(self._unpadded(y0))

        on_step_t = False
        if len(self._step_t):
            next_step_t = self._step_t[self.next_step_index]
            on_step_t = t0 < next_step_t < w(t0 + dt)
            if on_step_t:
                t1 = next_step_t
                dt = w(t1 - t0)
        on_jump_t = False
        if len(self._jump_t):
            next_jump_t = self._jump_t[self.next_jump_index]
            on_jump_t = t0 < next_jump_t < w(t0 + dt)
            if on_jump_t:
                on_step_t = False
                t1 = next_jump_t
                dt = w(t1 - t0)

        # ---- Runge–Kutta stages (rk_common.py:43-90); times in the state precision T ----
        t0_T, dt_T, t1_T = T(t0), T(dt), T(t1)
        dt_signed = float(dt_T) * func.sign
        ops = self.ops
        row0 = self._beta[0]
        plain = not (torch.is_grad_enabled() and (y0.requires_grad or f0.requires_grad or self._anchor is not None))
        # graph of this trial's step size: the first, heuristic one has one (the controller is under no_grad) — and a
        # step cut short at a `step_t` / `jump_t` point: dt = t_point - t0 (rk_common.py:296-309) moves AGAINST whatever t0
        # moves with (the first step size, t[0] when `t` requires grad), and t1 = t_point with nothing any more
        clipped = on_step_t or on_jump_t
        if plain:
            dsh = None
        elif clipped:
            dsh = None if self._anchor is None else -self._anchor
        else:
            dsh = self._dt_shadow
        self._dt_shadow = None
        dsh_signed = None if dsh is None else dsh * func.sign
        lookahead = (self._lookahead and plain and func.callback_step is _null
                     and func.callback_accept_step is _null and func.callback_reject_step is _null)
        pre, self._pre = self._pre, None
        if lookahead and pre is not None:
            # this trial step's first stage was enqueued by the previous one (tdeq_stage_combine_sel on the pair and
            # the step size the device controller chose) together with its func evaluation
            yi, stage_times, k1 = pre
        else:
            times = [(t1_T, Perturb.PREV) if self._alpha_is_one[i] else (t0_T + self._alpha[i] * dt_T, Perturb.NONE)
                     for i in range(len(self._beta))]
            if len(times) <= 16 and plain:
                # one launch: first stage input + the step's stage times (tdeq_stage_combine_fill)
                yi = torch.empty_like(y0)
                tbuf = torch.empty(len(times), dtype=func.time_dtype, device=y0.device)
                kern.stage_combine_fill(yi, y0, [f0], row0.coef, dt_signed, tbuf,
                                        [func.user_time(t, p) for t, p in times])
                stage_times = tbuf.unbind(0)
            else:
                yi = ops.combine(y0, [f0], row0.coef, dt_signed, dsh_signed)
                shadows = None
                if dsh is not None:
                    # t_i = t0 + alpha_i dt0 (t1 = t0 + dt0 for alpha_i = 1), in user time
                    base = self._anchor
                    shadows = []
                    for i in range(len(self._beta)):
                        inc = dsh if self._alpha_is_one[i] else float(self._alpha[i]) * dsh
                        shadows.append((inc if base is None else base + inc) * func.sign)
                stage_times = func.time_tensors(kern, times, shadows=shadows)
            k1 = func.eval_at(stage_times[0], yi)
        k: List[torch.Tensor] = [f0, k1]
        n_rows = len(self._beta)
        fsal = self.tableau.fsal_solution
        builtin_norm = isinstance(self.norm, BuiltinNorm)
        err_partial = None
        err_rem = self._fuse[1:] if self._fuse is not None else None     # (stages, weights) left to the norm kernel
        nograd = not torch.is_grad_enabled()      # no-grad solves (the adjoint's two solves, inference): no graph checks
        carry = self._carry if (builtin_norm and (nograd or (plain and not k1.requires_grad))) else None
        y1_planned = None
        if carry is not None:
            # planned launches (tableaus.carry_plan): a launch may also emit the left-to-right prefixes of later rows'
            # sums (continued by those rows: fewer bytes) or a later stage input that needs no newer stage; launch row
            # n_rows of a pair whose solution is not its last stage input is the c_sol combine (no evaluation follows).
            # Every stage input is bit-identical to the row-by-row launches below.
            held, R = {}, len(carry.ops)
            for i in range(1, R):
                op = carry.ops[i]
                row = self._beta[i] if i < n_rows else self._c_sol
                if op is None:
                    yi = held.pop(i)                      # finished by an earlier launch
                elif len(op.targets) == 1 and not op.continues:
                    yi = torch.empty_like(y0)
                    kern.stage_combine(yi, y0, [k[j] for j in row.idx], row.coef, dt_signed)
                elif op.targets == (i, R) and i == R - 1 and not op.continues and op.idx == row.idx:
                    yi, held[R] = torch.empty_like(y0), torch.empty_like(y0)      # the end-of-step pair as before
                    kern.stage_combine_err(yi, held[R], y0, [k[j] for j in row.idx], row.coef, self._fuse[0], dt_signed)
                else:
                    outs = [torch.empty_like(y0) for _ in op.targets]
                    kern.stage_combine_multi(outs, op.spec, y0, held.pop(i) if op.continues else None,
                                             [k[j] for j in op.idx], dt_signed)
                    yi = outs[0]
                    for tgt, buf in zip(op.targets[1:], outs[1:]):
                        held[tgt] = buf
                if i < n_rows:
                    k.append(func.eval_at(stage_times[i], yi))
                else:
                    y1_planned = yi
            err_partial, err_rem = held.pop(R), (carry.err_idx, carry.err_coef)
        for i in range(1, n_rows if carry is None else 0):
            row = self._beta[i]
            if i == n_rows - 1 and fsal and self._fuse is not None and builtin_norm and \
                    (nograd or not (y0.requires_grad or k[-1].requires_grad)):
                yi, err_partial = torch.empty_like(y0), torch.empty_like(y0)
                kern.stage_combine_err(yi, err_partial, y0, [k[j] for j in row.idx], row.coef, self._fuse[0], dt_signed)
            elif nograd:
                yi = torch.empty_like(y0)
                kern.stage_combine(yi, y0, [k[j] for j in row.idx], row.coef, dt_signed)
            else:
                yi = ops.combine(y0, [k[j] for j in row.idx], row.coef, dt_signed, dsh_signed)
            k.append(func.eval_at(stage_times[i], yi))
        if fsal:
            y1 = yi
        elif y1_planned is not None:
            y1 = y1_planned
        elif self._fuse is not None and builtin_norm and \
                not (torch.is_grad_enabled() and (y0.requires_grad or k[-1].requires_grad)):
            sol = self._c_sol
            y1, err_partial = torch.empty_like(y0), torch.empty_like(y0)
            kern.stage_combine_err(y1, err_partial, y0, [k[j] for j in sol.idx], sol.coef, self._fuse[0], dt_signed)
        else:
            y1 = ops.combine(y0, [k[j] for j in self._c_sol.idx], self._c_sol.coef, dt_signed, dsh_signed)
        f1 = k[-1]

        # ---- error ratio (misc.py:80-82) ----
        err = self._c_err
        use_ctrl = lookahead and (err_partial is not None or (self._whole_row_ctrl and builtin_norm))
        if use_ctrl:
            ctrl = self._ctrl
            ctrl.t0, ctrl.dt = t0, dt
            tnext = torch.empty(ctrl.n_times, dtype=func.time_dtype, device=y0.device)
            if err_partial is None:
                kern.error_norm_ctrl(self.plan, y0, y1, [k[j] for j in err.idx], err.coef, dt_signed, ctrl, tnext)
            elif self._sync is None:
                kern.error_norm_partial_ctrl(self.plan, err_partial, y0, y1, [k[j] for j in err_rem[0]],
                                             err_rem[1], dt_signed, ctrl, tnext)
            else:
                # lock step: this rank's sums -> device buffer, all-reduce over the ranks on the device, the
                # controller on the global sums (global element counts): every rank takes the whole-batch decision
                kern.error_norm_partial(self._plan_dev, err_partial, y0, y1, [k[j] for j in err_rem[0]],
                                        err_rem[1], dt_signed)
                self._sync.reduce_device(self._plan_dev.out, self.plan.n_seg)
                kern.step_controller(self.plan, self._plan_dev, self._plan_glob, ctrl, tnext, y0.dtype)
            if t1 < self._t_end and not self._hold_pre and not self._last_trial:
                # accepted or rejected, another trial step follows: enqueue its first stage and func evaluation now
                yi_n = torch.empty_like(y0)
                kern.stage_combine_sel(yi_n, y1, f1, y0, f0, row0.coef[0], self.plan)
                tn = tnext.unbind(0)
                self._pre = (yi_n, tn, func.eval_at(tn[0], yi_n))
            accept_dev, dt_next_dev, error_ratio, bad = kern.read_ctrl(self.plan)
            y1_nonfinite = any(b != 0 for b in bad)
        elif err_partial is not None:
            kern.error_norm_partial(self.plan, err_partial, y0, y1, [k[j] for j in err_rem[0]], err_rem[1],
                                    dt_signed)
            sumsq, _, bad = self._read_norms()
            error_ratio = self._segment_norm(sumsq, bad)
            y1_nonfinite = any(b != 0 for b in bad)
        elif builtin_norm:
            kern.error_norm(self.plan, y0, y1, [k[j] for j in err.idx], err.coef, dt_signed)
            sumsq, _, bad = self._read_norms()
            error_ratio = self._segment_norm(sumsq, bad)
            y1_nonfinite = any(b != 0 for b in bad)
        else:
            error_ratio, y1_nonfinite = self._user_norm_ratio(y0, y1, k, dt_signed)
        if use_ctrl:
            accept_step = accept_dev      # the device's decision is the one its look-ahead stage was built on
        else:
            # rk_common.py:324-332: a step at the floor is always taken, one above the ceiling never, else the error decides
            accept_step = bool(dt <= self.min_step or (error_ratio <= 1 and not dt > self.max_step))

        # ---- update state (rk_common.py:335-361) ----
        if accept_step:
            if func.callback_accept_step is not _null:
                func.callback_accept_step(self._time_tensor(t0), y0, self._time_tensor(dt))
            rec = _DenseRecord()
            rec.y0, rec.y1, rec.k, rec.dt_signed, rec.t0, rec.t1 = y0, y1, k, dt_signed, t0, t1
            rec.dt_shadow, rec.anchor = dsh_signed, self._anchor
            self._dense = rec
            if dsh is not None:
                # every later time of the solve is t0 + dt0 + constants: it moves with the first step size — until a
                # step ends on a prescribed point, a constant
                self._anchor = None if clipped else (dsh if self._anchor is None else self._anchor + dsh)
                func.set_time_anchor(self._anchor)
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
        if use_ctrl:
            self.dt = dt_next_dev         # already clamped (tdeq_error_norm_partial_ctrl)
        else:
            if self._wide:
                dt_next = optimal_step_size(dt, error_ratio, self.safety, self.ifactor, self.dfactor, self.order)
            else:
                dt_next = optimal_step_size_in(self._W, dt, error_ratio, self.safety, self.ifactor, self.dfactor,
                                               self.order)
            self.dt = _clamp(dt_next, self.min_step, self.max_step)

    # -- hipGraph mode -----------------------------------------------------------------------------------
    def _graph_step_ok(self) -> bool:
        func = self.func
        return (func.callback_step is _null and func.callback_accept_step is _null
                and func.callback_reject_step is _null
                and not (torch.is_grad_enabled() and (self.y1.requires_grad or self.f1.requires_grad
                                                      or self._anchor is not None or self._t_grad)))

    def _graph_trial_step(self) -> None:
        """One trial step as ONE hipGraph replay (`options={'hip_graph': True}`; small states, where a step costs
        launch latency).  The graph holds the S evaluations of `func`, the stage combines reading the step size
        from device memory (tdeq_stage_combine_dev) and the error norm + device controller (state_in_dev); the host
        replays the graph of the current side (see _GraphStep: two graphs over ping-pong state buffers), reads the
        controller's words, flips the side when the step was accepted and keeps its own mirror of (t0, dt) — identical
        doubles — for the output loop.  Same kernels' arithmetic and decisions as the eager path."""
        func, kern, T = self.func, self.kernels, self.np_dtype
        t0, dt = self.t1, self.dt
        if not math.isfinite(dt):
            dt = self.min_step
        dt = _clamp(dt, self.min_step, self.max_step)
        t1 = t0 + dt
        assert t0 + dt > t0, "underflow in dt {}".format(dt)
        assert not self._y_nonfinite, "non-finite values in state `y`: {}".format(self._unpadded(self.y1))
        g = self._g
        if g is None:
            g = self._g = _GraphStep.acquire(self, t0, dt)
        side = g.side
        g.run(self)
        accept_step, dt_next, _ratio, bad = g.take_words(kern, self.plan)
        dt_signed = float(T(dt)) * func.sign
        if accept_step:
            k = g.k[side]
            rec = _DenseRecord()
            if g.eager:
                # warm-up step on transient buffers: keep private copies of the pair it started from (the static pair
                # is about to receive its end state)
                rec.y0, rec.k = g.y[0].clone(), [g.f0.clone()] + k[1:]
                rec.y1 = g.y[1].clone()
                self.y1, self.f1 = g.y[0], g.f0
            else:
                rec.y0, rec.y1, rec.k = g.y[side], g.y[1 - side], k
                self.y1, self.f1 = g.y[1 - side], (k[-1] if side == 0 else g.f0)
            rec.dt_signed, rec.t0, rec.t1 = dt_signed, t0, t1
            self._dense = rec
            g.accepted(self)
            self.t0, self.t1 = t0, t1
            self._y_nonfinite = any(b != 0 for b in bad)
            self.n_accepted += 1
        else:
            self.t0 = t0
            self.n_rejected += 1
        self.dt = dt_next
        if g.refused is not None:
            # auto mode: the step just taken stands (it was evaluated eagerly), the rest of the solve runs on the eager
            # path from the state it left in g's buffers — which stay this solve's own (g is not handed back for reuse)
            self.hip_graph = False
            self._g = None
            self._hold_pre = False

    def _user_norm_ratio(self, y0, y1, k, dt_signed):
        """User-supplied `norm` callable (misc.py:80-82 with a custom norm): the kernel materialises
        err/tol (padding zero-filled) and the user's own function reduces it."""
        err = self._c_err
        y0, y1 = y0.detach(), y1.detach()
        if self._vec_fused is not None:
            # per-element tolerances under the built-in norm: err / tol and the per-segment sums in one launch, in the
            # reference's promoted precision (fp64); max over the components of sqrt(mean) as misc.py:22-33
            rtol_v, atol_v, n_skip = self._vec_fused
            self.kernels.error_norm_vec(self.plan, y0, y1, [k[j].detach() for j in err.idx], err.coef, dt_signed,
                                        rtol_v, atol_v)
            sumsq, _, bad = self.kernels.read_norms(self.plan)
            ratio = 0.0
            for s_, n_ in list(zip(sumsq, self._numels))[:len(self._numels) - n_skip]:
                if n_:
                    ratio = _nan_max(ratio, math.sqrt(s_ / n_))
            return ratio, any(b != 0 for b in bad)
        scaled = torch.empty_like(y0)
        self.kernels.error_scaled(self.plan, scaled, y0, y1, [k[j].detach() for j in err.idx], err.coef, dt_signed)
        _, _, bad = self.kernels.read_norms(self.plan)
        with torch.no_grad():
            if self._vec_tol is not None:       # `scaled` is the raw error estimate here (segment tolerances 0 / 1)
                scaled = scaled / (self._vec_tol[1] + self._vec_tol[0] * torch.maximum(y0.abs(), y1.abs()))
            ratio = self.norm(scaled)
        ratio = _norm_value(ratio)
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


class FixedGridODESolver(object):
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
                return self._integrate_graph(t)
            if not self._graph_auto and self._graph_explicit:
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
        if not torch.is_grad_enabled():
            return True
        if t.requires_grad or self.y0.requires_grad:
            return False
        # grad mode with neither y0 nor t in the graph: func's own parameters may still be (plain `odeint` training).
        # The replayed kernels write into raw buffers — a solution without an autograd graph — so such a solve has to
        # take the eager path.  One probe evaluation tells (as RKAdaptiveStepsizeODESolver._graph_step_ok reads f1);
        # it is not counted.
        nfe = self.func.nfe
        probe = self.func.eval(float(t[0].detach()), self.y0, self._first_perturb())
        self.func.nfe = nfe
        return not probe.requires_grad

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
        with torch.cuda.stream(side):
            step()
        current.wait_stream(side)
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


SOLVER_CLASSES = {"dopri8": Dopri8Solver, "dopri5": Dopri5Solver, "tsit5": Tsit5Solver, "bosh3": Bosh3Solver,
                  "fehlberg2": Fehlberg2, "adaptive_heun": AdaptiveHeunSolver, "euler": Euler,
                  "midpoint": Midpoint, "heun2": Heun2, "heun3": Heun3, "rk4": RK4,
                  "explicit_adams": AdamsBashforth, "implicit_adams": AdamsBashforthMoulton,
                  "fixed_adams": AdamsBashforthMoulton}
